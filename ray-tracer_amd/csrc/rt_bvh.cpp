// rt_bvh.cpp — host-side construction of the sphere hierarchy (two rounds of
// median splits on the longest axis of the centres' bounding box per node: up
// to four children; one sphere per leaf; children bounded by boxes).
#include "rt_bvh.h"

#include <algorithm>
#include <cmath>
#include <cstring>
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

  // distance from the world origin to a box, rounded down to a float whose low 16
  // bits are zero
  static uint32_t near_bits16(const float lo[3], const float hi[3]) {
    double s2 = 0;
    for (int d = 0; d < 3; d++) {
      const double g = lo[d] > 0 ? lo[d] : (hi[d] < 0 ? -(double)hi[d] : 0.0);
      s2 += g * g;
    }
    const float f = round_down(std::sqrt(s2) * (1.0 - 1e-6));
    uint32_t u;
    memcpy(&u, &f, 4);
    return u >> 16; // truncation of a non-negative float rounds down
  }

  // Packs the boxes of a node's children.  The grid origin is the low corner of
  // their union; per axis the unit is the smallest power of two with which 255
  // steps reach the high corner; every corner is rounded outwards and then
  // checked on the decoded value (the device's own arithmetic).  Returns false
  // if some box cannot be represented (ranges beyond what a float exponent
  // byte holds), in which case the scene gets no hierarchy.
  static bool pack(BvhNode &N, const Box *B, const int32_t *link, int parts) {
    memset(&N, 0, sizeof N);
    for (int k = 0; k < BVH_WIDTH; k++)
      N.link[k] = k < parts ? link[k] : BVH_EMPTY;
    for (int d = 0; d < 3; d++) {
      float lo = INFINITY, hi = -INFINITY;
      for (int k = 0; k < parts; k++) {
        lo = std::min(lo, B[k].lo[d]);
        hi = std::max(hi, B[k].hi[d]);
      }
      if (!(lo <= hi) || !std::isfinite(lo) || !std::isfinite(hi))
        return false;
      N.org[d] = lo;
      int e = 0;
      (void)std::frexp(((double)hi - (double)lo) / 255.0, &e); // extent / 255 < 2^e
      e = std::max(e, -120);
      for (;; e++) {
        if (e + 127 > 254)
          return false;
        const float u = std::ldexp(1.0f, e);
        bool ok = true;
        uint32_t qlo = 0, qhi = 0;
        for (int k = 0; k < parts && ok; k++) {
          long ql = (long)std::floor(((double)B[k].lo[d] - lo) / u);
          ql = std::max(0l, std::min(255l, ql));
          while (ql > 0 && fmaf((float)ql, u, lo) > B[k].lo[d])
            ql--;
          long qh = (long)std::ceil(((double)B[k].hi[d] - lo) / u);
          qh = std::max(0l, qh);
          while (qh <= 255 && fmaf((float)qh, u, lo) < B[k].hi[d])
            qh++;
          ok = qh <= 255 && fmaf((float)ql, u, lo) <= B[k].lo[d];
          qlo |= (uint32_t)ql << (8 * k);
          qhi |= (uint32_t)(qh & 255) << (8 * k);
        }
        if (ok) {
          N.exps |= (uint32_t)(e + 127) << (8 * d);
          N.qlo[d] = qlo;
          N.qhi[d] = qhi;
          break;
        }
      }
    }
    for (int k = 0; k < parts; k++) {
      const BvhBox D = bvh_child_box(N, k);
      N.near_bits[k >> 1] |= near_bits16(D.lo, D.hi) << (16 * (k & 1));
    }
    return true;
  }

  // Returns the link for order[b, e) and its box.
  bool failed = false;
  int depth_seen = 0;
  int32_t build(int b, int e, Box &box, int depth = 1) {
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
      cl[k] = k < parts ? build(lo[k], hi[k], cs[k], depth + 1) : BVH_EMPTY;
    if (!pack(out->nodes[me], cs, cl, parts))
      failed = true;
    depth_seen = std::max(depth_seen, depth);
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
    if (!Builder::pack(out.nodes[0], &root, &l, 1))
      B.failed = true;
    B.depth_seen = 1;
  } else {
    int32_t l = B.build(0, n, root);
    (void)l; // == 0: the first node pushed
  }
  out.depth = B.depth_seen;
  // A walk keeps the node it visits next in a register and stacks at most three
  // siblings per level: a deeper tree than the stack allows is not used at all
  // (the walks would otherwise have to drop nodes).
  out.valid = !B.failed && (BVH_WIDTH - 1) * B.depth_seen <= BVH_STACK;
}

} // namespace rt
