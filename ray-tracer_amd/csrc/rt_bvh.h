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
#include <vector>

namespace rt {

// One inner node holds the bounding boxes of up to four children in one
// 128-byte line, so a visit costs one load and the walk is half as deep as a
// binary tree's (the walk is bound by instruction issue and by the chain of
// dependent node loads).  Boxes, not bounding spheres: the reference's scenes
// are sheets of spheres, whose clusters are flat; a line touches 1.7 times
// fewer boxes than bounding spheres on the 100 x 100 grid (10.8 against 18.6
// node visits per bounce ray).  link >= 0: index of the child node.  link < 0:
// leaf, ~link = (slot << 3) into the leaf arrays (one sphere per leaf).
constexpr int BVH_WIDTH = 4;
struct alignas(128) BvhNode {
  // child c: a[c] = (lo.x, lo.y, lo.z, hi.x), b[c] = (hi.y, hi.z, near, bits of link).
  // [lo, hi] covers every member sphere entirely; near = distance from the world
  // origin to the box, rounded down (no point of the child is closer than this);
  // link = BVH_EMPTY: no such child.
  float4 a[BVH_WIDTH];
  float4 b[BVH_WIDTH];
};
constexpr int32_t BVH_EMPTY = 0x7fffffff;
constexpr int BVH_STACK = 48;

struct BvhHost {
  std::vector<BvhNode> nodes;
  std::vector<int32_t> leaf_idx; // sphere index of each leaf slot
  bool valid = false;            // false: scene has non-finite spheres, do not use
};

// spheres: 3*n floats, radius: n floats.
void bvh_build(const float *spheres, const float *radius, int n, BvhHost &out);

} // namespace rt
