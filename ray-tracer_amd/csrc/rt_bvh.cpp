// rt_bvh.cpp — host-side construction of the sphere hierarchy (two rounds of
// median splits on the longest axis of the centres' bounding box per node: up
// to four children; one sphere per leaf; children bounded by boxes).
#include "rt_bvh.h"

#include <algorithm>
#include <cmath>
#include <numeric>

namespace rt {
namespace {

struct Builder {
  const float *c;
  const float *r;
  std::vector<int32_t> order;
  BvhHost *out;

  // Smallest float >= x / largest float <= x.
  static float round_up(double x) {
    float f = (float)x;
    if ((double)f < x)
      f = std::nextafter(f, INFINITY);
    return f;
  }
  static float round_down(double x) {
    float f = (float)x;
    if ((double)f > x)
      f = std::nextafter(f, -INFINITY);
    return f;
  }

  struct Box {
    float lo[3], hi[3];
  };

  // Box of order[b, e) covering every member sphere entirely, rounded outwards;
  // axis = longest axis of the centres' box.
  Box bound(int b, int e, int &axis) const {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    double clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = b; k < e; k++) {
      const double rad = std::fabs((double)r[order[k]]) * (1.0 + 1e-6);
      for (int d = 0; d < 3; d++) {
        const double v = c[3 * order[k] + d];
        clo[d] = std::min(clo[d], v);
        chi[d] = std::max(chi[d], v);
        lo[d] = std::min(lo[d], v - rad);
        hi[d] = std::max(hi[d], v + rad);
      }
    }
    axis = 0;
    for (int d = 1; d < 3; d++)
      if (chi[d] - clo[d] > chi[axis] - clo[axis])
        axis = d;
    Box B;
    for (int d = 0; d < 3; d++) {
      B.lo[d] = round_down(lo[d]);
      B.hi[d] = round_up(hi[d]);
    }
    return B;
  }

  // distance from the world origin to the box, rounded down
  static float near_bound(const Box &B) {
    double s2 = 0;
    for (int d = 0; d < 3; d++) {
      const double g = B.lo[d] > 0 ? B.lo[d] : (B.hi[d] < 0 ? -(double)B.hi[d] : 0.0);
      s2 += g * g;
    }
    const double dist = std::sqrt(s2);
    return round_down(dist * (1.0 - 1e-6));
  }

  void split(int b, int e, int &m) {
    int axis;
    (void)bound(b, e, axis);
    m = b + (e - b) / 2;
    std::nth_element(order.begin() + b, order.begin() + m, order.begin() + e,
                     [&](int32_t x, int32_t y) {
                       float vx = c[3 * x + axis], vy = c[3 * y + axis];
                       return vx < vy || (vx == vy && x < y);
                     });
  }

  static void store(BvhNode &N, int k, const Box &B, int32_t link, bool present) {
    union {
      int32_t i;
      float f;
    } bits;
    bits.i = link;
    if (present) {
      N.a[k] = make_float4(B.lo[0], B.lo[1], B.lo[2], B.hi[0]);
      N.b[k] = make_float4(B.hi[1], B.hi[2], near_bound(B), bits.f);
    } else {
      N.a[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      N.b[k] = make_float4(0.f, 0.f, INFINITY, bits.f);
    }
  }

  // Returns the link for order[b, e) and its box.
  int32_t build(int b, int e, Box &box) {
    int axis;
    box = bound(b, e, axis);
    const int n = e - b;
    if (n == 1) {
      const int slot = (int)out->leaf_idx.size();
      out->leaf_idx.push_back(order[b]);
      return ~(slot << 3);
    }
    // two median splits: up to four parts
    int m;
    split(b, e, m);
    int parts = 0;
    int lo[4], hi[4];
    const int halves[2][2] = {{b, m}, {m, e}};
    for (int h = 0; h < 2; h++) {
      const int hb = halves[h][0], he = halves[h][1];
      if (he - hb >= 2) {
        int mm;
        split(hb, he, mm);
        lo[parts] = hb;
        hi[parts++] = mm;
        lo[parts] = mm;
        hi[parts++] = he;
      } else if (he - hb == 1) {
        lo[parts] = hb;
        hi[parts++] = he;
      }
    }
    const int me = (int)out->nodes.size();
    out->nodes.push_back(BvhNode{});
    Box cs[BVH_WIDTH];
    int32_t cl[BVH_WIDTH];
    for (int k = 0; k < BVH_WIDTH; k++)
      cl[k] = k < parts ? build(lo[k], hi[k], cs[k]) : BVH_EMPTY;
    BvhNode &N = out->nodes[me];
    for (int k = 0; k < BVH_WIDTH; k++)
      store(N, k, cs[k], cl[k], k < parts);
    return me;
  }
};

} // namespace

void bvh_build(const float *spheres, const float *radius, int n, BvhHost &out) {
  out.nodes.clear();
  out.leaf_idx.clear();
  out.valid = false;
  if (n <= 0)
    return;
  for (int i = 0; i < n; i++)
    if (!std::isfinite(spheres[3 * i]) || !std::isfinite(spheres[3 * i + 1]) ||
        !std::isfinite(spheres[3 * i + 2]) || !std::isfinite(radius[i]) ||
        std::fabs(spheres[3 * i]) > 1e18f || std::fabs(spheres[3 * i + 1]) > 1e18f ||
        std::fabs(spheres[3 * i + 2]) > 1e18f || std::fabs(radius[i]) > 1e18f)
      return;
  Builder B;
  B.c = spheres;
  B.r = radius;
  B.order.resize(n);
  std::iota(B.order.begin(), B.order.end(), 0);
  B.out = &out;
  out.leaf_idx.reserve(n);
  Builder::Box root;
  // node 0 is always the root; a one-sphere scene still gets a node
  if (n == 1) {
    out.nodes.push_back(BvhNode{});
    int32_t l = B.build(0, 1, root);
    BvhNode &N = out.nodes[0];
    for (int k = 0; k < BVH_WIDTH; k++)
      Builder::store(N, k, root, k == 0 ? l : BVH_EMPTY, k == 0);
  } else {
    int32_t l = B.build(0, n, root);
    (void)l; // == 0: the first node pushed
  }
  out.valid = true;
}

} // namespace rt
