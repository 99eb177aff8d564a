// rt_tribvh.cpp — host-side construction of the triangle hierarchy (rt_tribvh.h).
//
// All bounds are formed in double from the float scene data and rounded
// outwards.  The per-triangle unit normal is evaluated in float by the very
// operations of maths.cpp:39-52 (this file is compiled without contraction), so
// it is the normal the device tests with; the allowances below do not depend on
// that being bit-exact.
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

// Builds the subtree of a range of `order` into `nodes` / `leaves`, numbering both
// from wherever those vectors stand.  Several builders may work on disjoint
// ranges of the same `order` at once, each into vectors of its own.
struct Builder {
  const float *v;              // 9 floats per triangle
  const std::vector<Tri> &tri;
  std::vector<int32_t> &order; // positions in `tri`
  std::vector<TriNode> *nodes;
  std::vector<int32_t> *leaves;
  Builder(const float *v_, const std::vector<Tri> &t, std::vector<int32_t> &o,
          std::vector<TriNode> *n, std::vector<int32_t> *l)
      : v(v_), tri(t), order(o), nodes(n), leaves(l) {}

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

  Bound bound(int b, int e, int &axis) const {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    double clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
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
      for (int d = 0; d < 3; d++) {
        clo[d] = std::min(clo[d], T.c[d]);
        chi[d] = std::max(chi[d], T.c[d]);
      }
      // N and -N mirror the origin to the same point: fold to one side
      const double sg = T.n[0] * first.n[0] + T.n[1] * first.n[1] + T.n[2] * first.n[2] < 0 ? -1 : 1;
      for (int d = 0; d < 3; d++)
        sum[d] += sg * T.n[d];
    }
    axis = 0;
    for (int d = 1; d < 3; d++)
      if (chi[d] - clo[d] > chi[axis] - clo[axis])
        axis = d;
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

  // longest axis of the centroids' box of order[b, e): all a split needs
  int split_axis(int b, int e) const {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = b; k < e; k++) {
      const Tri &T = tri[order[k]];
      for (int d = 0; d < 3; d++) {
        lo[d] = std::min(lo[d], T.c[d]);
        hi[d] = std::max(hi[d], T.c[d]);
      }
    }
    int axis = 0;
    for (int d = 1; d < 3; d++)
      if (hi[d] - lo[d] > hi[axis] - lo[axis])
        axis = d;
    return axis;
  }

  void split(int b, int e, int &m) {
    const int axis = split_axis(b, e);
    m = b + (e - b) / 2;
    std::nth_element(order.begin() + b, order.begin() + m, order.begin() + e,
                     [&](int32_t x, int32_t y) {
                       const double vx = tri[x].c[axis], vy = tri[y].c[axis];
                       return vx < vy || (vx == vy && x < y);
                     });
  }

  // par: levels of the recursion whose children are built by tasks of their own
  // (ranges above 32768 triangles only).  The tasks build into private vectors,
  // which are then appended in child order with their links shifted: exactly the
  // numbering of the sequential depth-first build.
  int32_t build(int b, int e, Bound &me_bound, int par = 0) {
    int axis;
    me_bound = bound(b, e, axis);
    if (e - b == 1) {
      const int slot = (int)leaves->size();
      leaves->push_back(tri[order[b]].index);
      return ~(slot << 3);
    }
    int m;
    split(b, e, m);
    int parts = 0, lo[4], hi[4];
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
          Builder sub(v, tri, order, &sub_nodes[k], &sub_leaves[k]);
          return sub.build(lo[k], hi[k], cb[k], par - 1);
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
        cl[k] = build(lo[k], hi[k], cb[k], 0);
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
  std::vector<int32_t> order;
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
  // Median splits quarter a range per level.  A walk keeps the node it visits
  // next in a register and stacks at most three siblings per level: a tree deeper
  // than the stack allows is not built (the triangles are then looped over).
  int depth = 0;
  for (long cnt = m; cnt > 1; depth++)
    cnt = ((cnt + 1) / 2 + 1) / 2;
  if ((BVH_WIDTH - 1) * std::max(depth, 1) > BVH_STACK) {
    out.valid = false;
    return;
  }
  order.resize(m);
  std::iota(order.begin(), order.end(), 0);
  out.leaf_idx.reserve(m);
  Builder B(tris, fit, order, &out.nodes, &out.leaf_idx);
  Builder::Bound root;
  if (m == 1) {
    out.nodes.push_back(TriNode{});
    const int32_t l = B.build(0, 1, root);
    TriNode &N = out.nodes[0];
    for (int k = 0; k < BVH_WIDTH; k++) {
      N.s[k] = root.s;
      N.ns[k] = root.ns;
      N.link[k] = k == 0 ? l : BVH_EMPTY;
    }
  } else {
    (void)B.build(0, m, root, 2); // node 0 is the root; up to 16 tasks
  }
}

} // namespace rt
