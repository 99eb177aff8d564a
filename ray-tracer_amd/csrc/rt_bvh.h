// rt_bvh.h — hierarchy of boxes over the scene's spheres.
//
// Used for rays that do not travel in coherent bundles (bounce rays, bulk
// check_intersection queries): each lane walks the tree for its own ray with a
// small stack and runs the reference's arithmetic only on the spheres of the
// leaves its line can touch.  Which spheres are tested changes, the result does
// not: the nearest-hit rule "smallest |P|, lowest index on ties"
// (renderer.cpp:168-177) does not depend on the order of the tests.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <vector>

namespace rt {

// One inner node holds the bounding boxes of up to four children in 64 bytes,
// half a cache line.  A walk reads one node per visit and lane, every lane a
// different one, and the vector L1 serves such scattered 16-byte reads at about
// one lane per cycle: the bytes of a node, not the arithmetic on them, set the
// pace of a walk (128-byte nodes kept the texture path 70 % busy and the vector
// ALUs 30 %).  So the child boxes are stored as 8-bit offsets from a per-node
// origin in units of a per-axis power of two, rounded outwards: a decoded box
// contains the exact one, which is all the walk's exits need (rt_device.h,
// bvh_visit); the slack is at most 1/255 of the node's extent per side, small
// beside the allowance the exits already make.  Boxes, not bounding spheres:
// the reference's scenes are sheets of spheres, whose clusters are flat; a line
// touches 1.7 times fewer boxes than bounding spheres on the 100 x 100 grid.
// link >= 0: index of the child node.  link < 0: leaf, ~link = (slot << 3) into
// the leaf arrays (one sphere per leaf).  link = BVH_EMPTY: no such child.
constexpr int BVH_WIDTH = 4;
struct alignas(64) BvhNode {
  float org[3];            // origin of the node's grid
  uint32_t exps;           // byte k: biased exponent e of axis k's unit 2^(e - 127)
  int32_t link[BVH_WIDTH];
  uint32_t qlo[3];         // byte c of qlo[k]: low corner of child c on axis k, in units
  uint32_t qhi[3];         // the same for the high corner
  uint32_t near_bits[2];   // upper 16 bits of four floats (children 0,1 | 2,3): a lower
                           // bound of the distance from the world origin to the child
};
static_assert(sizeof(BvhNode) == 64, "one node is half a line");
constexpr int32_t BVH_EMPTY = 0x7fffffff;
constexpr int BVH_STACK = 48;

// A child's box as the walk sees it.  Host (builder, checker) and device decode
// with the same correctly rounded operations, so the builder verifies exactly
// the numbers the walk will use.
struct BvhBox {
  float lo[3], hi[3], near;
};
__host__ __device__ inline float bvh_unit(uint32_t exps, int axis) {
  union {
    uint32_t u;
    float f;
  } v;
  v.u = ((exps >> (8 * axis)) & 0xffu) << 23;
  return v.f;
}
__host__ __device__ inline float bvh_near(const uint32_t near_bits[2], int c) {
  union {
    uint32_t u;
    float f;
  } v;
  const uint32_t w = near_bits[c >> 1];
  v.u = (c & 1) ? (w & 0xffff0000u) : (w << 16);
  return v.f;
}
__host__ __device__ inline BvhBox bvh_child_box(const BvhNode &N, int c) {
  BvhBox B;
  for (int k = 0; k < 3; k++) {
    const float u = bvh_unit(N.exps, k);
    B.lo[k] = fmaf((float)((N.qlo[k] >> (8 * c)) & 0xffu), u, N.org[k]);
    B.hi[k] = fmaf((float)((N.qhi[k] >> (8 * c)) & 0xffu), u, N.org[k]);
  }
  B.near = bvh_near(N.near_bits, c);
  return B;
}

struct BvhHost {
  std::vector<BvhNode> nodes;
  std::vector<int32_t> leaf_idx; // sphere index of each leaf slot
  int depth = 0;                 // levels of inner nodes on the longest path
  bool valid = false;            // false: non-finite or unrepresentable spheres, or a
                                 // tree deeper than the walks' stack: do not use
};

// spheres: 3*n floats, radius: n floats.
void bvh_build(const float *spheres, const float *radius, int n, BvhHost &out);

} // namespace rt
