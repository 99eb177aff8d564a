// rt_bvh.cpp — host-side construction of the sphere hierarchy: up to four
// children per node, one sphere per leaf, children bounded by boxes.  A node's
// members are cut into parts by sweeps along the three axes that minimise
// area(left) * count(left) + area(right) * count(right) (the expected number of
// box tests for lines through the node), under the one constraint that no part
// holds more members than a subtree of the remaining levels can: the tree is
// never deeper than the least depth ceil(log4 n) a four-way tree of n leaves
// can have, which is what sizes the walks' stacks.  Which tree is built changes
// how many boxes a walk tests, never what it reports.
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

  // Box of the members in [b, e) covering every sphere entirely, rounded outwards.
  Box bound(int b, int e) const {
    const std::vector<int32_t> &order = by_axis[0];
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = b; k < e; k++) {
      const double rad = std::fabs((double)r[order[k]]) * (1.0 + 1e-6);
      for (int d = 0; d < 3; d++) {
        const double v = c[3 * order[k] + d];
        lo[d] = std::min(lo[d], v - rad);
        hi[d] = std::max(hi[d], v + rad);
      }
    }
    Box B;
    for (int d = 0; d < 3; d++) {
      B.lo[d] = round_down(lo[d]);
      B.hi[d] = round_up(hi[d]);
    }
    return B;
  }

  // The members in three orders, by centre along x, y and z (ties by index).  A
  // range [b, e) holds the same members in all three, each range sorted by its
  // own axis; order (= by x) is the one the boxes and leaves are read from.
  std::vector<int32_t> by_axis[3];
  std::vector<double> ilo, ihi;   // the members' own boxes, 3 doubles each
  std::vector<double> right_area; // scratch of a sweep
  std::vector<uint8_t> goes_left;
  std::vector<int32_t> moved;

  struct Sweep {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    void add(const double *l, const double *h) {
      for (int d = 0; d < 3; d++) {
        lo[d] = std::min(lo[d], l[d]);
        hi[d] = std::max(hi[d], h[d]);
      }
    }
    double area() const {
      const double x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
      return x * y + y * z + x * z;
    }
  };

  void prepare(int n) {
    ilo.resize(3 * (size_t)n);
    ihi.resize(3 * (size_t)n);
    for (int i = 0; i < n; i++) {
      const double rad = std::fabs((double)r[i]);
      for (int d = 0; d < 3; d++) {
        ilo[3 * (size_t)i + d] = (double)c[3 * i + d] - rad;
        ihi[3 * (size_t)i + d] = (double)c[3 * i + d] + rad;
      }
    }
    for (int a = 0; a < 3; a++) {
      by_axis[a].resize(n);
      std::iota(by_axis[a].begin(), by_axis[a].end(), 0);
      std::sort(by_axis[a].begin(), by_axis[a].end(), [&](int32_t x, int32_t y) {
        const float vx = c[3 * x + a], vy = c[3 * y + a];
        return vx < vy || (vx == vy && x < y);
      });
    }
    right_area.resize((size_t)n + 1);
    goes_left.assign(n, 0);
    moved.resize(n);
  }

  double area_of(int b, int e) const {
    Sweep S;
    for (int k = b; k < e; k++)
      S.add(&ilo[3 * (size_t)by_axis[0][k]], &ihi[3 * (size_t)by_axis[0][k]]);
    return S.area();
  }

  // Cuts [b, e) where the cost is least among the cuts that leave kmin .. kmax
  // members on the left (1 <= kmin <= kmax < e - b); returns the first index of
  // the right part.  Equal costs: lowest axis, then fewest members on the left.
  int cut(int b, int e, int kmin, int kmax) {
    const int n = e - b;
    double best = INFINITY;
    int best_axis = 0, best_k = kmin;
    for (int a = 0; a < 3; a++) {
      const int32_t *o = by_axis[a].data() + b;
      Sweep R;
      for (int k = n - 1; k >= kmin; k--) {
        R.add(&ilo[3 * (size_t)o[k]], &ihi[3 * (size_t)o[k]]);
        right_area[k] = R.area();
      }
      Sweep L;
      for (int k = 1; k <= kmax; k++) {
        L.add(&ilo[3 * (size_t)o[k - 1]], &ihi[3 * (size_t)o[k - 1]]);
        if (k < kmin)
          continue;
        const double cost = L.area() * k + right_area[k] * (n - k);
        if (cost < best) {
          best = cost;
          best_axis = a;
          best_k = k;
        }
      }
    }
    const int32_t *o = by_axis[best_axis].data() + b;
    for (int k = 0; k < n; k++)
      goes_left[o[k]] = k < best_k;
    for (int a = 0; a < 3; a++) {
      if (a == best_axis)
        continue;
      int32_t *q = by_axis[a].data() + b;
      int l = 0, rr = 0;
      for (int k = 0; k < n; k++)
        if (goes_left[q[k]])
          q[l++] = q[k]; // l <= k: never overwrites an entry not yet read
        else
          moved[rr++] = q[k];
      std::copy(moved.begin(), moved.begin() + rr, q + l);
    }
    return b + best_k;
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

  // Returns the link for the members in [b, e) and their box.
  bool failed = false;
  int depth_seen = 0;
  // levels: inner-node levels this subtree may use (4^levels >= e - b).
  int32_t build(int b, int e, Box &box, int levels, int depth = 1) {
    const std::vector<int32_t> &order = by_axis[0];
    box = bound(b, e);
    const int n = e - b;
    if (n == 1) {
      const int slot = (int)out->leaf_idx.size();
      out->leaf_idx.push_back(order[b]);
      return ~(slot << 3);
    }
    int parts = 0;
    int lo[BVH_WIDTH + 1];
    if (n <= BVH_WIDTH) {
      for (; parts < n; parts++)
        lo[parts] = b + parts;
    } else {
      long cap = 1; // members a child's subtree can hold
      for (int k = 1; k < levels; k++)
        cap *= BVH_WIDTH;
      // first cut: each side must fit two children
      lo[0] = b;
      lo[1] = cut(b, e, (int)std::max(1l, n - 2 * cap), (int)std::min((long)n - 1, 2 * cap));
      parts = 2;
      lo[2] = e;
      // two more cuts: a part too large for one child first, else the part with
      // the largest box
      while (parts < BVH_WIDTH) {
        int pick = -1, others_too_large = 0;
        for (int k = 0; k < parts; k++)
          if (lo[k + 1] - lo[k] > cap) {
            if (pick < 0)
              pick = k;
            else
              others_too_large++;
          }
        if (pick < 0) {
          double widest = -1.0;
          for (int k = 0; k < parts; k++)
            if (lo[k + 1] - lo[k] > 1) {
              const double a = area_of(lo[k], lo[k + 1]);
              if (a > widest) {
                widest = a;
                pick = k;
              }
            }
        }
        if (pick < 0)
          break;
        const long cnt = lo[pick + 1] - lo[pick];
        // a half may exceed one child's capacity only if a cut is left over for it
        const long larger = (BVH_WIDTH - (parts + 1)) - others_too_large >= 1 ? 2 * cap : cap;
        const int m = cut(lo[pick], lo[pick + 1], (int)std::max(1l, cnt - larger),
                          (int)std::min(cnt - 1, larger));
        for (int k = parts; k > pick; k--)
          lo[k + 1] = lo[k];
        lo[pick + 1] = m;
        parts++;
      }
    }
    lo[parts] = e;
    const int me = (int)out->nodes.size();
    out->nodes.push_back(BvhNode{});
    Box cs[BVH_WIDTH];
    int32_t cl[BVH_WIDTH];
    for (int k = 0; k < BVH_WIDTH; k++)
      cl[k] = k < parts ? build(lo[k], lo[k + 1], cs[k], levels - 1, depth + 1) : BVH_EMPTY;
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
  B.prepare(n);
  B.out = &out;
  int levels = 1; // least depth of a four-way tree over n leaves
  for (long reach = BVH_WIDTH; reach < n; reach *= BVH_WIDTH)
    levels++;
  out.leaf_idx.reserve(n);
  Builder::Box root;
  // node 0 is always the root; a one-sphere scene still gets a node
  if (n == 1) {
    out.nodes.push_back(BvhNode{});
    int32_t l = B.build(0, 1, root, levels);
    if (!Builder::pack(out.nodes[0], &root, &l, 1))
      B.failed = true;
    B.depth_seen = 1;
  } else {
    int32_t l = B.build(0, n, root, levels);
    (void)l; // == 0: the first node pushed
  }
  out.depth = B.depth_seen;
  // A walk keeps the node it visits next in a register and stacks at most three
  // siblings per level: a deeper tree than the stack allows is not used at all
  // (the walks would otherwise have to drop nodes).
  out.valid = !B.failed && (BVH_WIDTH - 1) * B.depth_seen <= BVH_STACK;
}

} // namespace rt
