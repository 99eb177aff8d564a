// rt_bvh.cpp — host-side construction of the bounding-sphere hierarchy
// (two rounds of median splits on the longest axis of the centres' bounding
// box per node: up to four children; one sphere per leaf).
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

  // Smallest float >= x.
  static float round_up(double x) {
    float f = (float)x;
    if ((double)f < x)
      f = std::nextafter(f, INFINITY);
    return f;
  }

  // Bounding sphere of order[b, e): centre of the centres' box, radius covering
  // every member sphere entirely, rounded outwards.
  float4 bound(int b, int e, int &axis) const {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = b; k < e; k++)
      for (int d = 0; d < 3; d++) {
        double v = c[3 * order[k] + d];
        lo[d] = std::min(lo[d], v);
        hi[d] = std::max(hi[d], v);
      }
    double mid[3];
    axis = 0;
    for (int d = 0; d < 3; d++) {
      mid[d] = 0.5 * (lo[d] + hi[d]);
      if (hi[d] - lo[d] > hi[axis] - lo[axis])
        axis = d;
    }
    float cf[3] = {(float)mid[0], (float)mid[1], (float)mid[2]};
    double R = 0;
    for (int k = b; k < e; k++) {
      const float *p = c + 3 * order[k];
      double dx = p[0] - (double)cf[0], dy = p[1] - (double)cf[1], dz = p[2] - (double)cf[2];
      R = std::max(R, std::sqrt(dx * dx + dy * dy + dz * dz) + std::fabs((double)r[order[k]]));
    }
    return make_float4(cf[0], cf[1], cf[2], round_up(R * (1.0 + 1e-6)));
  }

  static float near_bound(float4 s) {
    double d = std::sqrt((double)s.x * s.x + (double)s.y * s.y + (double)s.z * s.z) - (double)s.w;
    float f = (float)(d * (d > 0 ? 1.0 - 1e-6 : 1.0 + 1e-6));
    if ((double)f > d)
      f = std::nextafter(f, -INFINITY);
    return f;
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

  // Returns the link for order[b, e) and its bounding sphere.
  int32_t build(int b, int e, float4 &sphere) {
    int axis;
    sphere = bound(b, e, axis);
    const int n = e - b;
    if (n == 1) {
      const int slot = (int)out->leaf_idx.size();
      out->leaf_idx.push_back(order[b]);
      return ~(slot << 3);
    }
    // two median splits: up to four parts
    int cut[5] = {b, b, b, b, e};
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
    (void)cut;
    const int me = (int)out->nodes.size();
    out->nodes.push_back(BvhNode{});
    float4 cs[BVH_WIDTH];
    int32_t cl[BVH_WIDTH];
    for (int k = 0; k < BVH_WIDTH; k++) {
      if (k < parts) {
        cl[k] = build(lo[k], hi[k], cs[k]);
      } else {
        cl[k] = BVH_EMPTY;
        cs[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    BvhNode &N = out->nodes[me];
    for (int k = 0; k < BVH_WIDTH; k++) {
      N.s[k] = cs[k];
      N.link[k] = cl[k];
      N.near[k] = k < parts ? near_bound(cs[k]) : INFINITY;
    }
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
  float4 root;
  // node 0 is always the root; a one-sphere scene still gets a node
  if (n == 1) {
    out.nodes.push_back(BvhNode{});
    int32_t l = B.build(0, 1, root);
    BvhNode &N = out.nodes[0];
    for (int k = 0; k < BVH_WIDTH; k++) {
      N.s[k] = root;
      N.link[k] = k == 0 ? l : BVH_EMPTY;
      N.near[k] = k == 0 ? -INFINITY : INFINITY;
    }
  } else {
    int32_t l = B.build(0, n, root);
    (void)l; // == 0: the first node pushed
  }
  out.valid = true;
}

} // namespace rt
