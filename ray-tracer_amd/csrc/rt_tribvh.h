// rt_tribvh.h — hierarchy over the scene's triangles.
//
// The reference's triangle test (maths.cpp:64-119) computes
//     t = (N.orig + d) / (N.dir),   d = N.p1          (maths.cpp:77-78)
// so the point P = orig + t dir it reports satisfies N.(P - p1) = 2 N.orig: it
// lies on the triangle's plane pushed along N by 2 N.orig, and the three inside
// tests (maths.cpp:92-116) look at P only through its projection along N.  Put
// differently: the ray hits triangle i exactly when the ray with the SAME
// direction from the mirrored origin
//     o'_i = orig - 2 (N_i.orig) N_i
// hits it in the textbook sense, and then P = Q + 2 (N_i.orig) N_i for the
// textbook hit point Q.  |o'_i| = |orig|, and o'_i moves by at most
// 2 |orig| s (1 + 2s) when N_i moves by s.
//
// A node therefore stores, per child, a bounding sphere of the member
// triangles, a mean unit normal and the largest distance s of a member normal
// from it: a ray can only hit a member if the line from the mean mirrored origin
// passes within  R + 2 |orig| s (1 + 2s)  (+ rounding terms) of the centre.  For
// rays that start near the world origin (the eye) this is an ordinary tight
// hierarchy; for rays that start far from it the cull loosens with the spread of
// normals inside a node, and degrades to "test everything" for a node of
// random normals, but never drops a triangle the reference would report.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

#include "rt_bvh.h"

namespace rt {

// One node = one 128-byte line.  A million triangles make ~50 MB of nodes, far
// more than the caches hold, and every lane of a wave reads a different node:
// the walk is bound by the lines it pulls from HBM, so the mean normal is
// packed to 3 x 16 bits and the spread to a half float (both rounded so that the
// bound only grows).
struct alignas(128) TriNode {
  float4 s[BVH_WIDTH];     // centre.xyz, radius (covers the member triangles and the
                           // builder's per-triangle allowances)
  uint2 ns[BVH_WIDTH];     // x: mean normal n.x, n.y as signed 16-bit fractions of
                           //    32767 (low, high); y: n.z (low), spread as half (high)
  int32_t link[BVH_WIDTH]; // as BvhNode: >= 0 child node, < 0 leaf ~(slot << 3),
                           // BVH_EMPTY none
  int32_t pad[4];
};

// A leaf's triangle: what maths.cpp:64-119 reads, in one 64-byte half line.
struct alignas(64) TriLeaf {
  float4 n_d;  // N.xyz, d = N.p1 (as TriRec)
  float4 a;    // p1.xyz, p2.x
  float4 b;    // p2.yz, p3.xy
  float4 c;    // p3.z, bits of the triangle's index, 0, 0
};

struct TriBvhHost {
  std::vector<TriNode> nodes;
  std::vector<int32_t> leaf_idx; // triangle index of each leaf slot
  std::vector<int32_t> loose;    // triangles every ray tests directly (ill-shaped or
                                 // out-of-range; normally none)
  bool valid = false;
};

// tris: 9*n floats (p1, p2, p3), the reference's layout (main.cpp:270-305).
// Triangles whose normal (maths.cpp:39-46, evaluated in float) is zero or not
// finite can never be reported as hit and are left out altogether.
void tri_bvh_build(const float *tris, int n, TriBvhHost &out);

} // namespace rt
