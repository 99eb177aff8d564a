// rt_tribvh.cpp — host-side construction of the triangle hierarchy (rt_tribvh.h).
//
// All bounds are formed in double from the float scene data and rounded
// outwards.  The per-triangle unit normal is evaluated in float by the very
// operations of maths.cpp:39-52 (this file is compiled without contraction), so
// it is the normal the device tests with; the allowances below do not depend on
// that being bit-exact.
//
// A node's triangles are cut into up to four parts by sweeps along the three
// axes (centroid order) that minimise size(left) * count(left) + size(right) *
// count(right), under the one constraint that no part holds more triangles than
// a subtree of the remaining levels can: the tree has the least depth
// ceil(log4 n) of a four-way tree.  Which tree is built changes how many nodes
// a walk visits, never what it reports.
#include "rt_tribvh.h"

#include <algorithm>
#include <cmath>
#include <future>
#include <numeric>

namespace rt {
namespace {

float round_up(double x) {
  float f = (float)x;
  if ((double)f < x)
    f = std::nextafter(f, INFINITY);
  return f;
}

// maths.cpp:39-52 + 27-37 in float
bool float_normal(const float *p1, const float *p2, const float *p3, float N[3]) {
  float ax = p2[0] - p1[0], ay = p2[1] - p1[1], az = p2[2] - p1[2];
  float bx = p3[0] - p1[0], by = p3[1] - p1[1], bz = p3[2] - p1[2];
  float nx = ay * bz - az * by;
  float ny = az * bx - ax * bz;
  float nz = ax * by - ay * bx;
  float l = sqrtf(nx * nx + ny * ny + nz * nz);
  if (l != 0.0f) {
    nx /= l;
    ny /= l;
    nz /= l;
  }
  N[0] = nx;
  N[1] = ny;
  N[2] = nz;
  return std::isfinite(nx) && std::isfinite(ny) && std::isfinite(nz) &&
         !(nx == 0.0f && ny == 0.0f && nz == 0.0f);
}

struct Tri {
  int32_t index;
  double n[3];   // unit normal (the float normal, renormalised in double)
  double c[3];   // centroid, split key
  double extra;  // how far the corners leave the plane N.(x - p1) = 0
};

// What every builder of one tree shares: the triangles (as positions in `tri`)
// in three orders, by centroid along x, y and z (ties by position).  A range
// [b, e) holds the same triangles in all three, each range sorted by its own
// axis; by_axis[0] is the one bounds and leaves are read from.
struct Orders {
  std::vector<int32_t> by_axis[3];
  std::vector<double> lo, hi;      // the triangles' own boxes, 3 doubles each
  std::vector<uint8_t> goes_left;  // by position in `tri`
};

struct Sweep {
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  void add(const double *l, const double *h) {
    for (int d = 0; d < 3; d++) {
      lo[d] = std::min(lo[d], l[d]);
      hi[d] = std::max(hi[d], h[d]);
    }
  }
  // What a line through the node pays for the part: the children are bounded
  // by spheres around their boxes' centres, and a line meets a sphere with a
  // probability proportional to its radius squared.  (With the box's area
  // instead, the 1 M-triangle 8K frame takes 64.9 ms against 61.4.)
  double size() const {
    const double x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
    return x * x + y * y + z * z;
  }
};

// Builds the subtree of a range of the orders into `nodes` / `leaves`, numbering
// both from wherever those vectors stand.  Several builders may work on disjoint
// ranges of the same orders at once, each into vectors of its own.
struct Builder {
  const float *v;              // 9 floats per triangle
  const std::vector<Tri> &tri;
  Orders &ord;
  std::vector<int32_t> &order; // = ord.by_axis[0]: positions in `tri`
  std::vector<TriNode> *nodes;
  std::vector<int32_t> *leaves;
  std::vector<double> right_size; // scratch of a sweep
  std::vector<int32_t> moved;
  Builder(const float *v_, const std::vector<Tri> &t, Orders &o, std::vector<TriNode> *n,
          std::vector<int32_t> *l)
      : v(v_), tri(t), ord(o), order(o.by_axis[0]), nodes(n), leaves(l) {}

  struct Bound {
    float4 s;
    uint2 ns;
  };

  // smallest half float >= x (x >= 0), as bits
  static uint32_t half_up(double x) {
    if (!(x > 0))
      return 0;
    if (x > 65504.0)
      return 0x7c00; // +inf
    int e;
    double m = std::frexp(x, &e); // x = m 2^e, m in [0.5, 1)
    uint32_t bits;
    if (e < -13) { // subnormal: units of 2^-24
      bits = (uint32_t)std::ceil(x * 16777216.0);
    } else {
      // 1.f x 2^(e-1): 10 fraction bits
      uint32_t f = (uint32_t)std::ceil((m * 2.0 - 1.0) * 1024.0);
      bits = ((uint32_t)(e - 1 + 15) << 10) + f; // a carry out of f moves to the exponent
    }
    return bits;
  }

  Bound bound(int b, int e) const {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    double sum[3] = {0, 0, 0};
    const Tri &first = tri[order[b]];
    for (int k = b; k < e; k++) {
      const Tri &T = tri[order[k]];
      const float *p = v + 9 * (size_t)T.index;
      for (int c = 0; c < 3; c++)
        for (int d = 0; d < 3; d++) {
          lo[d] = std::min(lo[d], (double)p[3 * c + d]);
          hi[d] = std::max(hi[d], (double)p[3 * c + d]);
        }
      // N and -N mirror the origin to the same point: fold to one side
      const double sg = T.n[0] * first.n[0] + T.n[1] * first.n[1] + T.n[2] * first.n[2] < 0 ? -1 : 1;
      for (int d = 0; d < 3; d++)
        sum[d] += sg * T.n[d];
    }
    const float cf[3] = {(float)(0.5 * (lo[0] + hi[0])), (float)(0.5 * (lo[1] + hi[1])),
                         (float)(0.5 * (lo[2] + hi[2]))};
    double sl = std::sqrt(sum[0] * sum[0] + sum[1] * sum[1] + sum[2] * sum[2]);
    double mean[3];
    if (sl > 1e-6 * (e - b))
      for (int d = 0; d < 3; d++)
        mean[d] = sum[d] / sl;
    else
      for (int d = 0; d < 3; d++)
        mean[d] = first.n[d];
    // the normal as the device will unpack it: 16-bit fractions, then its unit vector
    int32_t nq[3];
    double u[3], ul = 0;
    for (int d = 0; d < 3; d++) {
      nq[d] = (int32_t)std::lrint(mean[d] * 32767.0);
      u[d] = (double)(float)nq[d] * (double)(1.0f / 32767.0f);
      ul += u[d] * u[d];
    }
    ul = std::sqrt(ul);
    double R = 0, s = 0;
    for (int k = b; k < e; k++) {
      const Tri &T = tri[order[k]];
      const float *p = v + 9 * (size_t)T.index;
      for (int c = 0; c < 3; c++) {
        const double dx = p[3 * c] - (double)cf[0], dy = p[3 * c + 1] - (double)cf[1],
                     dz = p[3 * c + 2] - (double)cf[2];
        R = std::max(R, std::sqrt(dx * dx + dy * dy + dz * dz) + 1.01 * T.extra);
      }
      double dm = 0, dp = 0;
      for (int d = 0; d < 3; d++) {
        const double w = u[d] / ul;
        dm += (T.n[d] - w) * (T.n[d] - w);
        dp += (T.n[d] + w) * (T.n[d] + w);
      }
      s = std::max(s, std::sqrt(std::min(dm, dp)));
    }
    // the unpacked normal is not exactly unit: mirroring with it moves the image
    // of the origin by 2 (|u|^2 - 1) |o|, the same as 2 | |u| - 1 | more spread
    s += 2.0 * std::fabs(ul - 1.0) + 1e-6;
    Bound B;
    B.s = make_float4(cf[0], cf[1], cf[2], round_up(R * (1.0 + 1e-6)));
    B.ns.x = ((uint32_t)nq[0] & 0xffffu) | (((uint32_t)nq[1] & 0xffffu) << 16);
    B.ns.y = ((uint32_t)nq[2] & 0xffffu) | (half_up(s) << 16);
    return B;
  }

  double size_of(int b, int e) const {
    Sweep S;
    for (int k = b; k < e; k++)
      S.add(&ord.lo[3 * (size_t)order[k]], &ord.hi[3 * (size_t)order[k]]);
    return S.size();
  }

  // Cuts [b, e) where size(left) * count(left) + size(right) * count(right) is
  // least among the cuts that leave kmin .. kmax triangles on the left
  // (1 <= kmin <= kmax < e - b); returns the first index of the right part.
  // Equal costs: lowest axis, then fewest triangles on the left.
  int cut(int b, int e, int kmin, int kmax) {
    const int n = e - b;
    if ((int)right_size.size() < n + 1) {
      right_size.resize((size_t)n + 1);
      moved.resize(n);
    }
    double best = INFINITY;
    int best_axis = 0, best_k = kmin;
    for (int a = 0; a < 3; a++) {
      const int32_t *o = ord.by_axis[a].data() + b;
      Sweep R;
      for (int k = n - 1; k >= kmin; k--) {
        R.add(&ord.lo[3 * (size_t)o[k]], &ord.hi[3 * (size_t)o[k]]);
        right_size[k] = R.size();
      }
      Sweep L;
      for (int k = 1; k <= kmax; k++) {
        L.add(&ord.lo[3 * (size_t)o[k - 1]], &ord.hi[3 * (size_t)o[k - 1]]);
        if (k < kmin)
          continue;
        const double cost = L.size() * k + right_size[k] * (n - k);
        if (cost < best) {
          best = cost;
          best_axis = a;
          best_k = k;
        }
      }
    }
    const int32_t *o = ord.by_axis[best_axis].data() + b;
    for (int k = 0; k < n; k++)
      ord.goes_left[o[k]] = k < best_k;
    for (int a = 0; a < 3; a++) {
      if (a == best_axis)
        continue;
      int32_t *q = ord.by_axis[a].data() + b;
      int l = 0, r = 0;
      for (int k = 0; k < n; k++)
        if (ord.goes_left[q[k]])
          q[l++] = q[k]; // l <= k: never overwrites an entry not yet read
        else
          moved[r++] = q[k];
      std::copy(moved.begin(), moved.begin() + r, q + l);
    }
    return b + best_k;
  }

  // par: levels of the recursion whose children are built by tasks of their own
  // (ranges above 32768 triangles only).  The tasks build into private vectors,
  // which are then appended in child order with their links shifted: exactly the
  // numbering of the sequential depth-first build.
  // levels: levels of nodes this subtree may use (4^levels >= e - b).
  int32_t build(int b, int e, Bound &me_bound, int levels, int par = 0) {
    me_bound = bound(b, e);
    const int n = e - b;
    if (n == 1) {
      const int slot = (int)leaves->size();
      leaves->push_back(tri[order[b]].index);
      return ~(slot << 3);
    }
    int parts = 0, lo[BVH_WIDTH + 1];
    if (n <= BVH_WIDTH) {
      for (; parts < n; parts++)
        lo[parts] = b + parts;
    } else {
      long cap = 1; // triangles a child's subtree can hold
      for (int k = 1; k < levels; k++)
        cap *= BVH_WIDTH;
      // first cut: each side must fit two children
      lo[0] = b;
      lo[1] = cut(b, e, (int)std::max(1l, n - 2 * cap), (int)std::min((long)n - 1, 2 * cap));
      lo[2] = e;
      parts = 2;
      // two more cuts: a part too large for one child first, else the largest part
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
              const double a = size_of(lo[k], lo[k + 1]);
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
    const int me = (int)nodes->size();
    nodes->push_back(TriNode{});
    Bound cb[BVH_WIDTH];
    int32_t cl[BVH_WIDTH];
    for (int k = parts; k < BVH_WIDTH; k++) {
      cl[k] = BVH_EMPTY;
      cb[k].s = make_float4(0.f, 0.f, 0.f, 0.f);
      cb[k].ns = make_uint2(0u, 0u);
    }
    if (par > 0 && e - b > 32768) {
      std::vector<TriNode> sub_nodes[BVH_WIDTH];
      std::vector<int32_t> sub_leaves[BVH_WIDTH];
      std::future<int32_t> done[BVH_WIDTH];
      for (int k = 0; k < parts; k++)
        done[k] = std::async(std::launch::async, [&, k]() {
          Builder sub(v, tri, ord, &sub_nodes[k], &sub_leaves[k]);
          return sub.build(lo[k], lo[k + 1], cb[k], levels - 1, par - 1);
        });
      for (int k = 0; k < parts; k++) {
        int32_t link = done[k].get();
        const int node_off = (int)nodes->size(), leaf_off = (int)leaves->size();
        auto shift = [&](int32_t l) -> int32_t {
          if (l == BVH_EMPTY)
            return l;
          return l >= 0 ? l + node_off : ~((((~l) >> 3) + leaf_off) << 3);
        };
        for (TriNode &N : sub_nodes[k])
          for (int c = 0; c < BVH_WIDTH; c++)
            N.link[c] = shift(N.link[c]);
        cl[k] = shift(link);
        nodes->insert(nodes->end(), sub_nodes[k].begin(), sub_nodes[k].end());
        leaves->insert(leaves->end(), sub_leaves[k].begin(), sub_leaves[k].end());
      }
    } else {
      for (int k = 0; k < parts; k++)
        cl[k] = build(lo[k], lo[k + 1], cb[k], levels - 1, 0);
    }
    TriNode &N = (*nodes)[me];
    for (int k = 0; k < BVH_WIDTH; k++) {
      N.s[k] = cb[k].s;
      N.ns[k] = cb[k].ns;
      N.link[k] = cl[k];
    }
    return me;
  }
};

} // namespace

void tri_bvh_build(const float *tris, int n, TriBvhHost &out) {
  out.nodes.clear();
  out.leaf_idx.clear();
  out.loose.clear();
  out.valid = false;
  if (n <= 0)
    return;
  std::vector<Tri> fit;
  fit.reserve(n);
  for (int i = 0; i < n; i++) {
    const float *p = tris + 9 * (size_t)i;
    float Nf[3];
    if (!float_normal(p, p + 3, p + 6, Nf))
      continue; // can never be reported as hit
    bool big = false;
    for (int k = 0; k < 9; k++)
      big |= !(std::fabs(p[k]) <= 1e18f);
    // shape checks, in double on the float data
    Tri T;
    T.index = i;
    const double nl = std::sqrt((double)Nf[0] * Nf[0] + (double)Nf[1] * Nf[1] + (double)Nf[2] * Nf[2]);
    for (int d = 0; d < 3; d++) {
      T.n[d] = Nf[d] / nl;
      T.c[d] = ((double)p[d] + p[3 + d] + p[6 + d]) / 3.0;
    }
    // the float edge vectors the inside tests use (maths.cpp:92,101,110)
    const float ef[3][3] = {{p[3] - p[0], p[4] - p[1], p[5] - p[2]},
                            {p[6] - p[3], p[7] - p[4], p[8] - p[5]},
                            {p[0] - p[6], p[1] - p[7], p[2] - p[8]}};
    double el[3], diam = 0;
    bool ok = !big;
    for (int k = 0; k < 3 && ok; k++) {
      el[k] = std::sqrt((double)ef[k][0] * ef[k][0] + (double)ef[k][1] * ef[k][1] +
                        (double)ef[k][2] * ef[k][2]);
      diam = std::max(diam, el[k]);
      if (!(el[k] > 0))
        ok = false;
    }
    if (ok) {
      // N must be (nearly) perpendicular to every edge, or the prism the inside
      // tests describe is not the triangle's own
      for (int k = 0; k < 3; k++) {
        const double ne = std::fabs(T.n[0] * ef[k][0] + T.n[1] * ef[k][1] + T.n[2] * ef[k][2]);
        if (!(ne <= 0.4 * el[k]))
          ok = false;
      }
      const double a1 = std::fabs(T.n[0] * ((double)p[3] - p[0]) + T.n[1] * ((double)p[4] - p[1]) +
                                  T.n[2] * ((double)p[5] - p[2]));
      const double a2 = std::fabs(T.n[0] * ((double)p[6] - p[0]) + T.n[1] * ((double)p[7] - p[1]) +
                                  T.n[2] * ((double)p[8] - p[2]));
      T.extra = std::max(a1, a2);
      if (!(T.extra <= 0.01 * diam))
        ok = false;
      // sharpest corner: a rounding allowance d on each edge reaches d / sin(angle/2)
      // beyond the corner; the device's allowance assumes a factor <= 64
      for (int k = 0; k < 3 && ok; k++) {
        const int j = (k + 2) % 3; // edges k and j meet at corner k: e_k and -e_j
        const double cosang = -((double)ef[k][0] * ef[j][0] + (double)ef[k][1] * ef[j][1] +
                                (double)ef[k][2] * ef[j][2]) /
                              (el[k] * el[j]);
        const double half_sin = std::sqrt(std::max(0.0, (1.0 - cosang) * 0.5));
        if (!(half_sin * 64.0 >= 1.0))
          ok = false;
      }
    }
    if (!ok) {
      out.loose.push_back(i);
      continue;
    }
    fit.push_back(T);
  }
  const int m = (int)fit.size();
  out.valid = true;
  if (m == 0)
    return; // everything is loose or unhittable: no nodes
  // The tree is as deep as the shallowest four-way tree over m leaves.  A walk
  // keeps the node it visits next in a register and stacks at most three
  // siblings per level: a tree deeper than the stack allows is not built (the
  // triangles are then looped over).
  int depth = 1;
  for (long reach = BVH_WIDTH; reach < m; reach *= BVH_WIDTH)
    depth++;
  if ((BVH_WIDTH - 1) * depth > BVH_STACK) {
    out.valid = false;
    return;
  }
  Orders ord;
  ord.lo.resize(3 * (size_t)m);
  ord.hi.resize(3 * (size_t)m);
  ord.goes_left.assign(m, 0);
  for (int i = 0; i < m; i++) {
    const float *p = tris + 9 * (size_t)fit[i].index;
    for (int d = 0; d < 3; d++) {
      ord.lo[3 * (size_t)i + d] = std::min({p[d], p[3 + d], p[6 + d]});
      ord.hi[3 * (size_t)i + d] = std::max({p[d], p[3 + d], p[6 + d]});
    }
  }
  {
    std::future<void> sorted[3];
    for (int a = 0; a < 3; a++)
      sorted[a] = std::async(std::launch::async, [&, a]() {
        std::vector<int32_t> &o = ord.by_axis[a];
        o.resize(m);
        std::iota(o.begin(), o.end(), 0);
        std::sort(o.begin(), o.end(), [&](int32_t x, int32_t y) {
          const double vx = fit[x].c[a], vy = fit[y].c[a];
          return vx < vy || (vx == vy && x < y);
        });
      });
    for (int a = 0; a < 3; a++)
      sorted[a].get();
  }
  out.leaf_idx.reserve(m);
  Builder B(tris, fit, ord, &out.nodes, &out.leaf_idx);
  Builder::Bound root;
  if (m == 1) {
    out.nodes.push_back(TriNode{});
    const int32_t l = B.build(0, 1, root, depth);
    TriNode &N = out.nodes[0];
    for (int k = 0; k < BVH_WIDTH; k++) {
      N.s[k] = root.s;
      N.ns[k] = root.ns;
      N.link[k] = k == 0 ? l : BVH_EMPTY;
    }
  } else {
    (void)B.build(0, m, root, depth, 2); // node 0 is the root; up to 16 tasks
  }
}

} // namespace rt
