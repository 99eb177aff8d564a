// rt_device.h — device-side scene layout and the per-ray math of the path.
//
// Everything here must reproduce the reference's UNFUSED binary32/binary64
// arithmetic in source order (this translation unit is compiled with
// -ffp-contract=off; never -ffast-math).  Comments cite the reference line each
// expression follows.  Where work is hoisted (per-ray `a`, per-sphere r*r, the
// triangle normal) the hoisted value is computed by the same operations in the
// same order, so the bits are the same.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_bvh.h"
#include "rt_tribvh.h"

namespace rt {

// ---- layout in HBM ----------------------------------------------------------
// Spheres, two 16-byte records per sphere (structure of arrays):
//   geo[i]  = (cx, cy, cz, r*r)                      any ray
//   prim[i] = (Lx, Ly, Lz, c) with L = eye - centre,  rays leaving the eye
//             c = L.L - r*r  (maths.cpp:126,129)
// Both arrays are padded to a multiple of SPH_UNROLL with records that can
// never pass the discriminant test.
// Triangles, seven float4 per triangle (TriRec): unit normal N and d = N.p1
// (maths.cpp:69,77), the corners, the three edge vectors (maths.cpp:92,101,110)
// and N.eye for rays leaving the eye (maths.cpp:78).
struct TriRec {
  float4 n_d;   // N.xyz, d
  float4 p1_no; // p1.xyz, N.eye
  float4 p2;    // p2.xyz
  float4 p3;    // p3.xyz
  float4 e0;    // p2 - p1
  float4 e1;    // p3 - p2
  float4 e2;    // p1 - p3
};

constexpr int SPH_UNROLL = 4;

struct SceneDev {
  const float4 *sph_geo;
  const float4 *sph_prim;
  const TriRec *tri;
  const uchar4 *col_tri;
  const uchar4 *col_sph;
  const float4 *mat_tri; // (kr, kt, ior, 0) or nullptr
  const float4 *mat_sph;
  const float4 *lights;  // xyz
  int t_size, s_size, s_pad, l_size;
  // hierarchy of boxes over the spheres (rt_bvh.h); bvh == nullptr: none
  const BvhNode *bvh;
  const float4 *leaf_geo;    // sph_geo gathered into leaf order
  const int32_t *leaf_idx;   // original sphere index of each leaf slot
  int bvh_lds;               // the tree fits a lane's LDS stack (LDS_STACK_ROWS)
  const float4 *leaf_mat;    // per leaf slot: (kr, kt, ior, bits of the 8-bit colour)
  // hierarchy over the triangles (rt_tribvh.h); tbvh == nullptr: none, loop
  const TriNode *tbvh;
  const TriLeaf *tleaf;      // the triangles in leaf order, with their indices
  const int32_t *tloose;     // triangles outside the hierarchy, tested by every ray
  int n_tleaf, n_tloose;
  // Spheres by screen cell for the rays that leave the eye (k_bin_spheres), or
  // nullptr: cell (x, y) of BIN_PX x BIN_PX pixels has the record
  // bins[(y * bins_x + x) * BIN_WORDS ...] = {n, n sphere indices in rising order};
  // n < 0: no list for this cell (too many spheres, or no cone could be formed).
  const int32_t *bins;
  int bins_x;
  // Spheres by direction as seen from a light, for the shadow rays that aim at it
  // (k_bin_light), or nullptr.  The directions from the light are the cells of a cube:
  // six faces of LCELL_R x LCELL_R cells; cell c of light l has the record
  // lcells[(l * LCELL_COUNT + c) * LCELL_WORDS ...] = {n, n sphere indices}, n < 0: no
  // list.  A list serves rays that start within lcell_reach[l] of the light; lights
  // from lcell_lights on have no lists.
  const int32_t *lcells;
  int lcell_lights;
  float lcell_reach[4];
};
constexpr int LCELL_MAX_LIGHTS = 4;
// 64 / 128 / 192 / 256 / 512 cells along a face: the depth-1 4K frame of 10 000 spheres takes
// 0.286 / 0.271 / 0.269 / 0.279 / 0.316 ms (fewer, longer lists against more cells per tile).
constexpr int LCELL_R = 128;
constexpr int LCELL_COUNT = 6 * LCELL_R * LCELL_R;
constexpr int LCELL_WORDS = 32;
constexpr int LCELL_SUPER_R = 16; // the lists are made in two levels (k_bin_light)
constexpr int LCELL_SUPER_WORDS = 256;
static_assert(LCELL_R % LCELL_SUPER_R == 0, "a cell has one parent");
// cells of 8 / 16 / 32 pixels: 0.283 / 0.277 / 0.281 ms for that frame
constexpr int BIN_PX = 16;
constexpr int BIN_WORDS = 64;
// A list is read four entries at a time: one scalar load for the indices, then four
// independent ones for the spheres (entry by entry the two loads of every sphere wait
// for each other: depth-1 4K frame of 10 000 spheres 0.271 ms, in fours 0.263, in
// eights 0.295, more of the last group then being filler).  A list's capacity
// leaves room for the last group, and the makers fill a list up to a multiple of the
// group with index 0.
constexpr int LIST_GROUP = 4;
constexpr int BIN_CAP = BIN_WORDS - LIST_GROUP;
constexpr int BIN_SUPER_PX = 128;     // the lists are made in two levels (k_bin_spheres)
constexpr int BIN_SUPER_WORDS = 1024;
constexpr int BIN_SUPER_SLICES = 8;   // lists per first-level cell, one per range of spheres
static_assert(BIN_SUPER_PX % BIN_PX == 0, "a cell has one parent");

struct CameraDev {
  double tan_half; // tan(fov*0.5*M_PI/180), evaluated on the host (renderer.cpp:35)
  float aspect;    // (float)w/(float)h (renderer.cpp:11)
  float ox, oy, oz;
  float plane_z;
};

#define RT_INF __builtin_huge_valf()

// ---- maths.cpp helpers ------------------------------------------------------
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx,
                                      float by, float bz) {
  return ax * bx + ay * by + az * bz; // maths.cpp:54-56, left to right
}

__device__ __forceinline__ float length3(float x, float y, float z) {
  return sqrtf(x * x + y * y + z * z); // maths.cpp:23-25
}

__device__ __forceinline__ void normalize3(float &x, float &y, float &z) {
  float l = length3(x, y, z); // maths.cpp:27-37
  if (l == 0.0f)
    return;
  x /= l;
  y /= l;
  z /= l;
}

// maths.cpp:163-168
__device__ __forceinline__ void reflect3(float dx, float dy, float dz, float nx,
                                         float ny, float nz, float &rx,
                                         float &ry, float &rz) {
  float k = -2.0f * dot3(dx, dy, dz, nx, ny, nz);
  rx = nx * k + dx;
  ry = ny * k + dy;
  rz = nz * k + dz;
}

// Unit normal of a triangle exactly as maths.cpp:39-46 (sub, sub, cross,
// normalize).  Used once per triangle at scene upload.
__device__ __forceinline__ void tri_normal(const float *p1, const float *p2,
                                           const float *p3, float &nx,
                                           float &ny, float &nz) {
  float ax = p2[0] - p1[0], ay = p2[1] - p1[1], az = p2[2] - p1[2];
  float bx = p3[0] - p1[0], by = p3[1] - p1[1], bz = p3[2] - p1[2];
  nx = ay * bz - az * by; // maths.cpp:48-52
  ny = az * bx - ax * bz;
  nz = ax * by - ay * bx;
  normalize3(nx, ny, nz);
}

// ---- ray / sphere, the part after the discriminant screen --------------------
// maths.cpp:132-158 given a, b, c of maths.cpp:127-129 and the two products
// bb = b*b, ac = (4*a)*c of maths.cpp:132.  Returns true and P on a hit.
__device__ __forceinline__ bool sphere_finish(float a, float b, float c,
                                              float bb, float ac, float ox,
                                              float oy, float oz, float dx,
                                              float dy, float dz, float &Px,
                                              float &Py, float &Pz) {
  const float eps = 0.001f; // (float)0.001, maths.cpp:124
  float discr = bb - ac;
  float t0, t1;
  if (discr < 0) {
    return false;
  } else if (discr == 0) {
    t0 = t1 = (float)(-0.5 * (double)b / (double)a); // maths.cpp:136, double
  } else {
    // maths.cpp:138-141.  -0.5 * x is an exact scaling, so the double product
    // rounded to float equals the float product.
    float sq = sqrtf(discr);
    float q = (b > eps) ? -0.5f * (b + sq) : -0.5f * (b - sq);
    t0 = q / a;
    t1 = c / q;
  }
  if (t0 >= t1) { // maths.cpp:144-145
    float tmp = t1;
    t1 = t0;
    t0 = tmp;
  }
  if (t0 <= eps) { // maths.cpp:147-152
    t0 = t1;
    if (t0 <= eps)
      return false;
  }
  Px = ox + dx * t0; // maths.cpp:154-158
  Py = oy + dy * t0;
  Pz = oz + dz * t0;
  return true;
}

// ---- ray / triangle from the precomputed record -------------------------------
// maths.cpp:64-119.  `no` is N.orig (maths.cpp:78).
__device__ __forceinline__ bool triangle_test(const TriRec &T, float no,
                                              float ox, float oy, float oz,
                                              float dx, float dy, float dz,
                                              float &Px, float &Py, float &Pz) {
  const float eps = 0.0001f; // (float)0.0001, maths.cpp:66
  const float nx = T.n_d.x, ny = T.n_d.y, nz = T.n_d.z;
  float nd = dot3(nx, ny, nz, dx, dy, dz);
  if (fabsf(nd) < eps)
    return false;
  float t = (no + T.n_d.w) / nd; // maths.cpp:78 (sic: N.orig + d)
  if (t < 0)
    return false;
  Px = ox + dx * t; // maths.cpp:83-87
  Py = oy + dy * t;
  Pz = oz + dz * t;
  {
    float vx = Px - T.p1_no.x, vy = Py - T.p1_no.y, vz = Pz - T.p1_no.z;
    float cx = T.e0.y * vz - T.e0.z * vy; // maths.cpp:95
    float cy = T.e0.z * vx - T.e0.x * vz;
    float cz = T.e0.x * vy - T.e0.y * vx;
    if (dot3(nx, ny, nz, cx, cy, cz) < 0)
      return false;
  }
  {
    float vx = Px - T.p2.x, vy = Py - T.p2.y, vz = Pz - T.p2.z;
    float cx = T.e1.y * vz - T.e1.z * vy; // maths.cpp:104
    float cy = T.e1.z * vx - T.e1.x * vz;
    float cz = T.e1.x * vy - T.e1.y * vx;
    if (dot3(nx, ny, nz, cx, cy, cz) < 0)
      return false;
  }
  {
    float vx = Px - T.p3.x, vy = Py - T.p3.y, vz = Pz - T.p3.z;
    float cx = T.e2.y * vz - T.e2.z * vy; // maths.cpp:113
    float cy = T.e2.z * vx - T.e2.x * vz;
    float cz = T.e2.x * vy - T.e2.y * vx;
    if (dot3(nx, ny, nz, cx, cy, cz) < 0)
      return false;
  }
  return true;
}

// ---- ray / sphere, the whole of maths.cpp:121-161 ------------------------------
// (cx,cy,cz,rr) is the sphere record; a = dir.dir and a4 = 4*a are per ray.
__device__ __forceinline__ bool sphere_exact(float cx, float cy, float cz,
                                             float rr, float a, float a4,
                                             float ox, float oy, float oz,
                                             float dx, float dy, float dz,
                                             float &Px, float &Py, float &Pz) {
  float Lx = ox - cx, Ly = oy - cy, Lz = oz - cz;  // maths.cpp:126
  float b = 2.0f * dot3(dx, dy, dz, Lx, Ly, Lz);   // maths.cpp:128
  float c = dot3(Lx, Ly, Lz, Lx, Ly, Lz) - rr;     // maths.cpp:129
  float bb = b * b, ac = a4 * c;                   // maths.cpp:132
  if (bb < ac)                                     // discr < 0, maths.cpp:133
    return false;
  return sphere_finish(a, b, c, bb, ac, ox, oy, oz, dx, dy, dz, Px, Py, Pz);
}

// ---- wave-level helpers ----------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float uniform(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ float lane_value(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// ---- the beam screen ---------------------------------------------------------------
// The rays of one wave are coherent: the 64 primary rays of an 8x8 tile leave the
// eye inside a cone a few milliradians wide, and the 64 shadow rays towards one
// light all pass (to rounding) through the light.  Before the per-ray tests each
// wave therefore SCREENS the spheres 64 at a time, one sphere per lane, against a
// cone that contains all its rays' lines; only spheres the cone can touch are
// broadcast to the whole wave for the reference's per-ray arithmetic, in index
// order, so results (including ties) are the same as testing every sphere.
//
// The screen must never drop a sphere whose discriminant test
// (maths.cpp:132-133, evaluated in float) could pass for some ray.  Bound used:
// with L = o - c and |d|^2 = a, the float test passing implies
//     dist(c, line)^2  <=  r^2 + 21 u (|L|^2 + r^2),     u = 2^-24
// (3-term dot products contribute <= 3u relative to |d||L|, the remaining
// products/sums u each).  The screen uses eps = 2^-16 = 256u in place of 21u and,
// for rays that only approximately pass through the apex X (shadow rays:
// d = normalize(X - P) in float, angular error < 4u), widens the radius by
// delta = 2^-20 * max|X - P|.  Its own float arithmetic is made one-sided:
//   - sin(alpha) is inflated (x(1+2^-8) + 2^-30 on the square, x(1+2^-18) +
//     2^-20 on the root), cos(alpha) deflated by (1-2^-18);
//   - |M|^2 is deflated by (1-2^-19) before p^2 is taken from it;
//   - every comparison is written so that NaN/inf means "keep".
// Rays whose a = d.d is 0 or NaN can never produce an accepted hit in the
// reference (t becomes inf/NaN, |P| is then not below +inf) and are left out.
struct Beam {
  float Xx, Xy, Xz; // apex
  float Ux, Uy, Uz; // unit axis
  float sin_a, cos_a;
  float K;     // 2 eps D^2 + absolute slack, D = max distance ray origin - apex
  float delta; // radius widening for lines that miss the apex by rounding
  bool narrow; // false: do not screen, test everything
};

#define RT_SCREEN_EPS 1.52587890625e-05f /* 2^-16 */

// delta = D * delta_scale with delta_scale = 2^-20: every line passes through the
// apex up to the rounding of its direction.
__device__ __forceinline__ Beam make_beam(bool lv, float a, float dx, float dy,
                                          float dz, float Xx, float Xy, float Xz,
                                          float Dk, float delta_scale) {
  Beam B;
  B.Xx = Xx;
  B.Xy = Xy;
  B.Xz = Xz;
  const bool sane = a > 9.094947e-13f && a < 1.0995116e12f; // 2^-40 .. 2^40
  float sx = wave_sum(lv ? dx : 0.f), sy = wave_sum(lv ? dy : 0.f),
        sz = wave_sum(lv ? dz : 0.f);
  float n2 = fmaf(sz, sz, fmaf(sy, sy, sx * sx));
  float inv = 1.0f / sqrtf(n2);
  float Ux = uniform(sx * inv), Uy = uniform(sy * inv), Uz = uniform(sz * inv);
  float cx = Uy * dz - Uz * dy, cy = Uz * dx - Ux * dz, cz = Ux * dy - Uy * dx;
  float s2k = fmaf(cz, cz, fmaf(cy, cy, cx * cx)) / a;
  float along = fmaf(Uz, dz, fmaf(Uy, dy, Ux * dx));
  bool bad = lv && !(sane && along > 0.f && s2k >= 0.f);
  float s2 = uniform(wave_max(lv ? s2k : 0.f));
  s2 = fmaf(s2, 1.00390625f, 9.313225746154785e-10f); // (1+2^-8), 2^-30
  float D = uniform(wave_max(lv ? Dk : 0.f));
  B.Ux = Ux;
  B.Uy = Uy;
  B.Uz = Uz;
  B.sin_a = fmaf(sqrtf(s2), 1.000003814697265625f, 9.5367431640625e-07f);
  B.cos_a = sqrtf(1.0f - s2) * 0.999996185302734375f;
  B.K = fmaf(2.0f * RT_SCREEN_EPS * D, D, 8.271806125530277e-25f); // + 2^-80
  B.delta = D * delta_scale;
  B.narrow = !__any(bad) && s2 < 0.25f && n2 > 0.f && D < 1.8446744e19f;
  return B;
}

// true = the sphere may be hit by some ray of the beam (or nothing can be proved)
__device__ __forceinline__ bool beam_keeps(const Beam &B, float4 g) {
  float Mx = g.x - B.Xx, My = g.y - B.Xy, Mz = g.z - B.Xz;
  float m2 = fmaf(Mz, Mz, fmaf(My, My, Mx * Mx));
  float p = fmaf(B.Uz, Mz, fmaf(B.Uy, My, B.Ux * Mx));
  float q2 = fmaf(-p, p, m2 * 0.99999809265136719f); // (1-2^-19)
  float q = __builtin_amdgcn_sqrtf(fmaxf(q2, 0.f));
  float lhs = fmaf(q, B.cos_a, -(fabsf(p) * B.sin_a)); // <= |M| sin(theta-alpha)
  float reff2 = fmaf(g.w, 1.0f + RT_SCREEN_EPS, fmaf(m2, 2.0f * RT_SCREEN_EPS, B.K));
  float reff = fmaf(__builtin_amdgcn_sqrtf(reff2), 1.00000095367431640625f, B.delta);
  return !(lhs > reff);
}

// ---- a lane's stack of nodes still to visit --------------------------------------
// A walk stacks at most three siblings per level of the tree (the child it
// visits next stays in a register), and the builders refuse trees deeper than
// BVH_STACK / 3 levels, so BVH_STACK entries always suffice.
//
// Two homes for it.  PrivateStack: an array in private memory (scratch), 192 bytes
// per lane, for trees of any admitted depth.  In LDS (kernels of 256 threads): word
// (row * 256 + thread), so the 64 lanes of a wave hit 64 different banks whatever
// row each of them is at; LDS_STACK_ROWS rows serve trees of up to
// LDS_WALK_MAX_DEPTH levels of inner nodes (three rows per level below which there
// are inner nodes, and one row more that takes the stores that do not count:
// stack_children), 65 536 spheres when balanced.  An entry is
// (16 bits of ordering key) << 16 | node index, hence at most 65 535 nodes; the host
// sets SceneDev::bvh_lds only for such trees.  Scratch stores and loads go through
// the same queue as the walk's node loads and delay them; the LDS form also lets
// the visit be written without branches (below).
constexpr int LDS_STACK_ROWS = 22;
constexpr int LDS_WALK_MAX_DEPTH = 8;
constexpr size_t LDS_WALK_MAX_NODES = 65535;
static_assert(3 * (LDS_WALK_MAX_DEPTH - 1) + 1 <= LDS_STACK_ROWS,
              "three rows per level but the last, and the row of the stores that do not count");
// The stack interface of the walks that push and pop by branches (k_walk, the
// triangle walks): LdsStack keeps the first LDS_STACK_ROWS entries in the block's
// LDS (lds = this thread's column) and only what a deeper walk stacks beyond them
// in private memory; PrivateStack keeps everything there.  The triangle walk of
// the 1 M-triangle frame wrote 3.5 GB of stack per bounce level through scratch;
// with LdsStack its bounce levels take 6 % less time.
struct LdsStack {
  int *lds;
  int spill[BVH_STACK - LDS_STACK_ROWS];
  __device__ __forceinline__ void push(int &sp, int v) {
    if (sp < LDS_STACK_ROWS)
      lds[sp * 256] = v;
    else if (sp < BVH_STACK)
      spill[sp - LDS_STACK_ROWS] = v;
    sp += sp < BVH_STACK ? 1 : 0;
  }
  __device__ __forceinline__ int pop(int &sp) {
    --sp;
    return sp < LDS_STACK_ROWS ? lds[sp * 256] : spill[sp - LDS_STACK_ROWS];
  }
};
struct PrivateStack {
  int slot[BVH_STACK];
  __device__ __forceinline__ void push(int &sp, int v) {
    if (sp < BVH_STACK)
      slot[sp++] = v;
  }
  __device__ __forceinline__ int pop(int &sp) { return slot[--sp]; }
};

// ---- check_intersection (renderer.cpp:145-193) ---------------------------------
// One lane = one ray; the primitive being tested is the same for every lane of
// the wave, so its record is read through wave-uniform (scalar) loads or a lane
// broadcast and the per-lane work is the reference's arithmetic only.
//
// FROM_EYE: the ray starts at the camera eye, so L, c (spheres) and N.orig
//           (triangles) come precomputed from the scene records.
// ANY_HIT : only "is anything hit" is wanted (shadow rays, renderer.cpp:119);
//           the loop may stop as soon as every live lane has a hit, which
//           cannot change the answer.
// BEAM    : RT_BEAM_NONE = every ray against every sphere.  RT_BEAM_APEX = the
//           wave's rays share the apex (Xx,Xy,Xz) (the eye, or the light a shadow
//           ray aims at) and are screened as a bundle; Dk is the lane's distance
//           from its ray origin to the apex.  RT_ACCEL_BVH = no bundle: each lane
//           walks the sphere hierarchy for its own ray (bvh_visit).
//
// Nearest is by |P| from the WORLD origin with strict '>' (lowest index wins in
// a class); between classes a triangle wins ties ('<=', renderer.cpp:181).  A
// candidate whose |P| is not below +inf is never accepted (renderer.cpp:159,170
// with the FLT_MAX start value), which also keeps NaNs out.
struct Hit {
  int kind;  // 0 none, 1 triangle, 2 sphere
  int index; // class-local
  float Px, Py, Pz;
};

struct ScreenStats { // filled only by the verification build of the kernels
  unsigned long long *dropped_but_candidate;
  unsigned long long *kept;
  unsigned long long *walk; // [0] rays walked, [1] node bounds tested, [2] leaf spheres tested
  unsigned long long *twalk = nullptr; // the same three for the triangle hierarchy
};

#define RT_BEAM_NONE 0
#define RT_BEAM_APEX 1
#define RT_ACCEL_BVH 3

// One node visit of the sphere walk: which of the node's (up to four) children
// can hold a sphere the line {o + t d} touches?  A child must be kept whenever
// the float discriminant test (maths.cpp:132-133) can pass for a member sphere
// (centre c, radius r, entirely inside the child's box [lo, hi]).  From the
// bound above,
//   dist(c, line) <= r sqrt(1+64u) + sqrt(64u) |o - c|,  sqrt(64u) = 2^-9,
// and with S1 := the sum over the axes of max(|lo - o|, |hi - o|) >= |o - c| + r,
// the line passes within 0.002 S1 of the box, i.e. through the box grown by
// m = 0.0021 S1 on every side.  S1 is taken from the node's whole grid box
// G = [org, org + 255 unit], which contains every child box, so it is at least
// each child's own S1 and the allowances below only grow.
//
// Slab arithmetic.  A child corner is org + q unit (q = 0..255); per axis
//     t_lo = fma(q_lo, U, A_lo),  U = unit * i,  A_lo = ((org - o) - m) * i
//     t_hi = fma(q_hi, U, A_hi),                 A_hi = ((org - o) + m) * i
// are the parameters at which the line crosses the grown box's faces (i = 1/d on
// that axis).  Against exact arithmetic each carries at most 2^-20 S1 |i| of
// rounding (four roundings of numbers no larger than 2 S1 |i|), a hundredth of
// the extra 1e-4 S1 |i| that m allows on that axis.  The child is kept unless the
// entry parameter exceeds the exit parameter by more than 2^-20 of their
// magnitudes.  |i| is capped at 2^60 (sph_walk_ray): with |d| >= 2^-20 (the walks
// are only used then) a direction component below 2^-60 moves the line by less
// than 2^-39 S1 across any box in reach, so the cap only turns infinite slabs
// into astronomically wide ones, rejects where the origin lies outside the slab
// by more than m, and keeps 0 x inf out of the arithmetic.  Whatever NaN or
// inf - inf is left (overflowing coordinates) answers "keep".
//
// Two more ways out, both one-sided, with mu = 0.0045 S1 > 2^-8 (|o - c| + r):
//  - the grown box lies behind the ray origin by more than mu: its exit
//    parameter is below -mu / |d| (ida >= 1/|d|).  Every member's closest
//    approach and roots lie inside the box's parameter range, so d.L > mu |d|
//    and |L|^2 - r^2 > mu^2, margins thousands of times their rounding:
//    maths.cpp:138-152 computes two negative roots and reports no hit;
//  - every point of the box is farther from the world origin than `bound`
//    (the best |P| so far): near > bound + mu.  A reported P is within mu of the
//    sphere it belongs to even when the roots are ill-conditioned (the
//    discriminant carries an absolute error <= 17u (2|d||L|)^2, i.e. <= 1e-3 |L|
//    on the roots), so such a child cannot supply a nearer hit.  Ties are kept:
//    the comparison is strict.
// Two ordering keys come back: entry_key, the parameter at which the line enters
// the grown box (any-hit walks: the nearer an occluder is to the ray's origin, the
// sooner it ends the walk), and near_key, the child's distance bound from the
// world origin (nearest-hit walks: visiting the nearest child first tightens
// `bound` soonest).
struct SphWalkRay {
  float ix, iy, iz, ida;
};
__device__ __forceinline__ float capped_inverse(float d) {
  // the hardware reciprocal (within one ulp of 1/d: as if the line were turned by
  // 2^-23 rad, a ten-thousandth of what m allows); 1/0 = inf and huge quotients
  // become 2^60 with the sign of d
  return copysignf(fminf(fabsf(__builtin_amdgcn_rcpf(d)), 1152921504606846976.0f), d);
}
__device__ __forceinline__ SphWalkRay sph_walk_ray(float dx, float dy, float dz, float a) {
  SphWalkRay r;
  r.ix = capped_inverse(dx);
  r.iy = capped_inverse(dy);
  r.iz = capped_inverse(dz);
  r.ida = __builtin_amdgcn_rsqf(a) * 1.001f;
  return r;
}

__device__ __forceinline__ void bvh_visit2(const BvhNode &N, float ox, float oy, float oz,
                                           float ix, float iy, float iz, float ida, float bound,
                                           bool touch[BVH_WIDTH], float near_key[BVH_WIDTH],
                                           float entry_key[BVH_WIDTH]) {
  const float ux = bvh_unit(N.exps, 0), uy = bvh_unit(N.exps, 1), uz = bvh_unit(N.exps, 2);
  const float bx = N.org[0] - ox, by = N.org[1] - oy, bz = N.org[2] - oz;
  const float gx = fmaf(255.0f, ux, bx), gy = fmaf(255.0f, uy, by), gz = fmaf(255.0f, uz, bz);
  const float S1 = fmaxf(fabsf(bx), fabsf(gx)) + fmaxf(fabsf(by), fabsf(gy)) +
                   fmaxf(fabsf(bz), fabsf(gz));
  const float m = S1 * 0.0021f, mu = S1 * 0.0045f;
  const float Ux = ux * ix, Uy = uy * iy, Uz = uz * iz;
  // Per axis the line enters through the low face when it runs upwards and through
  // the high face otherwise (i is never zero: capped_inverse), so the faces are
  // picked by the sign of i once per node instead of by a min and a max per child:
  // with s = m carrying the sign of i, An = ((org - o) - s) * i belongs to the
  // entry face and Af = ((org - o) + s) * i to the exit face, the same numbers as
  // A_lo and A_hi above.  fma is monotonic in each argument, so t_n <= t_f
  // wherever the box is not inverted.
  const bool upx = ix >= 0.f, upy = iy >= 0.f, upz = iz >= 0.f;
  const float sx = copysignf(m, ix), sy = copysignf(m, iy), sz = copysignf(m, iz);
  const float Anx = (bx - sx) * ix, Afx = (bx + sx) * ix;
  const float Any = (by - sy) * iy, Afy = (by + sy) * iy;
  const float Anz = (bz - sz) * iz, Afz = (bz + sz) * iz;
  const uint32_t qnx = upx ? N.qlo[0] : N.qhi[0], qfx = upx ? N.qhi[0] : N.qlo[0];
  const uint32_t qny = upy ? N.qlo[1] : N.qhi[1], qfy = upy ? N.qhi[1] : N.qlo[1];
  const uint32_t qnz = upz ? N.qlo[2] : N.qhi[2], qfz = upz ? N.qhi[2] : N.qlo[2];
  const float behind_below = -(mu * ida);
  const float far_above = bound + mu;
#pragma unroll
  for (int c = 0; c < BVH_WIDTH; c++) {
    const int sh = 8 * c;
    const float tnx = fmaf((float)((qnx >> sh) & 0xffu), Ux, Anx);
    const float tny = fmaf((float)((qny >> sh) & 0xffu), Uy, Any);
    const float tnz = fmaf((float)((qnz >> sh) & 0xffu), Uz, Anz);
    const float tfx = fmaf((float)((qfx >> sh) & 0xffu), Ux, Afx);
    const float tfy = fmaf((float)((qfy >> sh) & 0xffu), Uy, Afy);
    const float tfz = fmaf((float)((qfz >> sh) & 0xffu), Uz, Afz);
    const float tmin = fmaxf(fmaxf(tnx, tny), tnz);
    const float tmax = fminf(fminf(tfx, tfy), tfz);
    const bool miss = (tmin - tmax) > 9.5367431640625e-07f * (fabsf(tmin) + fabsf(tmax));
    const bool behind = tmax < behind_below;
    const float near = bvh_near(N.near_bits, c);
    const bool far = near > far_above;
    touch[c] = N.link[c] != BVH_EMPTY && !miss && !behind && !far;
    near_key[c] = near;
    entry_key[c] = tmin;
  }
}

template <bool ANY_HIT>
__device__ __forceinline__ void bvh_visit(const BvhNode &N, float ox, float oy, float oz,
                                          float ix, float iy, float iz, float ida, float bound,
                                          bool touch[BVH_WIDTH], float key[BVH_WIDTH]) {
  float other[BVH_WIDTH];
  if (ANY_HIT)
    bvh_visit2(N, ox, oy, oz, ix, iy, iz, ida, bound, touch, other, key);
  else
    bvh_visit2(N, ox, oy, oz, ix, iy, iz, ida, bound, touch, key, other);
}

constexpr uint32_t NO_ENTRY = 0xffffffffu;

// What a visit does with the touched inner children, written without branches: each
// becomes an entry (key, node), integer min / max bring the one with the smallest
// key to the front, the others go on the lane's LDS stack by unconditional
// stores whose row only advances for a real entry (a store that does not count is
// overwritten by the next one), and the nearest one, or failing that the top of the
// stack, is returned (NO_ENTRY: the walk is over).  by_entry: order by the
// parameter at which the line enters the child (any-hit walks: the nearer an
// occluder is to the ray's origin, the sooner it ends the walk); otherwise by the
// child's distance bound from the world origin (it tightens `bound` soonest).
// Sixteen bits of either are plenty for an order.  The branching form of this, a
// dozen exec-mask branches per child, was more than half of a visit's instructions.
__device__ __forceinline__ uint32_t stack_children(const BvhNode &N, const bool touch[BVH_WIDTH],
                                                   const float entry_key[BVH_WIDTH],
                                                   bool by_entry, uint32_t *stack_l, int &sp) {
  uint32_t e[BVH_WIDTH];
#pragma unroll
  for (int c = 0; c < BVH_WIDTH; c++) {
    const int32_t link = N.link[c];
    const uint32_t nk =
        (c & 1) ? (N.near_bits[c >> 1] & 0xffff0000u) : (N.near_bits[c >> 1] << 16);
    const uint32_t ek = __float_as_uint(fmaxf(entry_key[c], 0.f)) & 0xffff0000u;
    e[c] = (touch[c] && link >= 0) ? ((by_entry ? ek : nk) | (uint32_t)link) : NO_ENTRY;
  }
  // the smallest entry to the front (three steps of integer min / max); the other
  // three go on the stack as they come (ordering them all, two steps more, was
  // measured half a percent slower)
  {
    const uint32_t m01 = e[0] < e[1] ? e[0] : e[1], x01 = e[0] < e[1] ? e[1] : e[0];
    const uint32_t m23 = e[2] < e[3] ? e[2] : e[3], x23 = e[2] < e[3] ? e[3] : e[2];
    e[0] = m01 < m23 ? m01 : m23;
    e[1] = m01 < m23 ? m23 : m01;
    e[2] = x01;
    e[3] = x23;
  }
#pragma unroll
  for (int c = BVH_WIDTH - 1; c >= 1; c--) {
    stack_l[sp * 256] = e[c];
    sp += e[c] != NO_ENTRY ? 1 : 0;
  }
  uint32_t next = e[0];
  if (next == NO_ENTRY && sp > 0)
    next = stack_l[--sp * 256];
  return next;
}

// ---- the triangle hierarchy (rt_tribvh.h) ------------------------------------------
// Can the ray hit a triangle of the child (s, n, spread)?  Must say yes whenever
// the float test of maths.cpp:64-119 can report a hit for a member triangle i.
// Such a hit has P~ = o + t~ d (float) passing the three inside tests with
// t~ >= 0.  Write Q~ = P~ - 2 (N_i.o) N_i.  Then
//   - Q~ is within e1 = 7u|o| + 2u(|M| + R) of the line from the mirrored origin
//     o'_i = o - 2 (N_i.o) N_i along d (rounding of P~, maths.cpp:83-87);
//   - Q~ projects along N_i into the triangle grown by at most
//     11u * 64 * (2|o| + diam): each inside test is a float evaluation of
//     (N x e_k).(P~ - p_k) with error <= 10u |e_k||P~ - p_k| (maths.cpp:92-116),
//     and the builder only admits corners no sharper than a growth factor of 64;
//   - N_i.(Q~ - p1) = N_i.o + t~ (N_i.d) - N_i.p1 deviates from 0 by at most
//     lam = 8u (1 + |d| / |nd|) (|o| + |p1|): 4u from the sums of maths.cpp:77-78,
//     3u |d| / |nd| from the rounding of nd = N.d (maths.cpp:69), 1u from the
//     division; |nd| >= max(|n.d| - spread |d|, 0.99e-4) by the parallel test
//     (maths.cpp:72); plus the builder's allowance for corners off the plane,
//     which is inside R.
// So Q~ lies within R + lam + 2^-13 (|o| + R) + e1 of the centre, and o'_i within
// |o| (K - 2^-12) of the mean mirrored origin q = o - 2 (n.o) n (K = n.w).  With
// |o| + |p1| <= 2|o| + |M| + R =: S,
//   Rtot = R (1 + 2^-12) + |o| K + 2e-5 |M| + 2^-21 S (1 + |d| / ndmin)
// bounds the distance from the centre to the line {q + t d} for every possible
// hit, t~ |d|^2 <= (C - q).d + Rtot |d| (so a child entirely behind q cannot be
// hit: maths.cpp:80 rejects t < 0), and |P~| >= |C + (o - q)| - Rtot.  NaN or inf
// anywhere answers yes.  `ol` >= |o|, `da` >= |d|.
// NEAREST false (any-hit walks: no best |P| to prune against, no visiting order to
// choose): the last exit is not evaluated and near_out is 0.
template <bool NEAREST = true>
__device__ __forceinline__ bool tri_touches(float4 s, uint2 ns, float ox, float oy, float oz,
                                            float dx, float dy, float dz, float a, float da,
                                            float ol, float bound, float &near_out) {
  // unpack: the normal's 16-bit fractions (its length is within 5e-5 of 1, which
  // the builder has added to the spread), the spread as a half float
  float4 n;
  n.x = (float)(int16_t)(ns.x & 0xffffu) * (1.0f / 32767.0f);
  n.y = (float)(int16_t)(ns.x >> 16) * (1.0f / 32767.0f);
  n.z = (float)(int16_t)(ns.y & 0xffffu) * (1.0f / 32767.0f);
  union {
    unsigned short u;
    _Float16 h;
  } cv;
  cv.u = (unsigned short)(ns.y >> 16);
  const float sp = (float)cv.h;
  n.w = fmaf(2.0f * sp, fmaf(2.0f, sp, 1.0f), 0.000244140625f) * 1.000001f; // K
  const float spread = sp + 3.814697265625e-06f;                            // + 2^-18
  const float no2 = 2.0f * fmaf(n.z, oz, fmaf(n.y, oy, n.x * ox));
  const float qx = fmaf(-no2, n.x, ox), qy = fmaf(-no2, n.y, oy), qz = fmaf(-no2, n.z, oz);
  const float Mx = s.x - qx, My = s.y - qy, Mz = s.z - qz;
  const float m2 = fmaf(Mz, Mz, fmaf(My, My, Mx * Mx));
  const float md = fmaf(dz, Mz, fmaf(dy, My, dx * Mx));
  const float mlen = __builtin_amdgcn_sqrtf(m2) * 1.000001f;
  const float nbd = fabsf(fmaf(n.z, dz, fmaf(n.y, dy, n.x * dx)));
  const float ndmin = fmaxf(fmaf(-spread, da, nbd * 0.9999f) * 0.999999f, 9.9e-5f);
  const float S = fmaf(2.0f, ol, mlen + s.w);
  const float lam = (S * 4.76837158203125e-07f) *
                    fmaf(da * 1.00001f, __builtin_amdgcn_rcpf(ndmin), 1.0f); // 2^-21
  const float Rtot = fmaf(s.w, 1.000245f, fmaf(ol, n.w, fmaf(mlen, 2e-5f, lam)));
  const float lhs = fmaf(-md, md, (a * m2) * 0.99999809265136719f);
  const float rhs = (a * Rtot) * (Rtot * 1.000001f);
  const bool behind = fmaf(da, Rtot * 1.0001f, md) < 0.f;
  bool far = false;
  near_out = 0.f;
  if (NEAREST) {
    const float fx = Mx + ox, fy = My + oy, fz = Mz + oz; // C + (o - q)
    const float flen = __builtin_amdgcn_sqrtf(fmaf(fz, fz, fmaf(fy, fy, fx * fx)));
    near_out = fmaf(flen, 0.999998f, -(Rtot * 1.000002f));
    far = near_out > bound;
  }
  return !(lhs > rhs) && !behind && !far;
}

// One exact triangle test (maths.cpp:64-119) folded into the running nearest
// triangle hit: smallest |P|, lowest index on ties, the order-free form of
// renderer.cpp:156-166.  Returns true when the hit was accepted.
__device__ __forceinline__ bool tri_fold(const TriRec &T, int idx, float ox, float oy, float oz,
                                         float dx, float dy, float dz, float &len_t, int &idx_t,
                                         float &tPx, float &tPy, float &tPz) {
  const float no = dot3(T.n_d.x, T.n_d.y, T.n_d.z, ox, oy, oz); // maths.cpp:78
  float Px, Py, Pz;
  if (!triangle_test(T, no, ox, oy, oz, dx, dy, dz, Px, Py, Pz))
    return false;
  const float len = length3(Px, Py, Pz);
  if (len < RT_INF && (len < len_t || (len == len_t && idx < idx_t))) {
    len_t = len;
    idx_t = idx;
    tPx = Px;
    tPy = Py;
    tPz = Pz;
    return true;
  }
  return false;
}

// The same for a leaf of the hierarchy: the edge vectors are the float
// differences k_prep_scene stores in a TriRec (maths.cpp:92,101,110).
__device__ __forceinline__ bool tri_fold_leaf(const TriLeaf &L, float ox, float oy, float oz,
                                              float dx, float dy, float dz, float &len_t,
                                              int &idx_t, float &tPx, float &tPy, float &tPz) {
  TriRec T;
  T.n_d = L.n_d;
  T.p1_no = make_float4(L.a.x, L.a.y, L.a.z, 0.f);
  T.p2 = make_float4(L.a.w, L.b.x, L.b.y, 0.f);
  T.p3 = make_float4(L.b.z, L.b.w, L.c.x, 0.f);
  T.e0 = make_float4(T.p2.x - T.p1_no.x, T.p2.y - T.p1_no.y, T.p2.z - T.p1_no.z, 0.f);
  T.e1 = make_float4(T.p3.x - T.p2.x, T.p3.y - T.p2.y, T.p3.z - T.p2.z, 0.f);
  T.e2 = make_float4(T.p1_no.x - T.p3.x, T.p1_no.y - T.p3.y, T.p1_no.z - T.p3.z, 0.f);
  return tri_fold(T, __float_as_int(L.c.y), ox, oy, oz, dx, dy, dz, len_t, idx_t, tPx, tPy, tPz);
}

// Per-ray constants of the triangle walk.  `usable` false: the ray's numbers
// are outside the range the bounds were derived for and every triangle must
// be tested; `possible` false: no triangle can be reported at all (d = 0 gives
// nd = 0, a NaN in d or a non-finite origin makes |P| NaN or inf, and
// renderer.cpp:159 never accepts those).
struct TriWalkRay {
  float a, da, ol;
  bool possible, usable;
};
__device__ __forceinline__ TriWalkRay tri_walk_ray(float ox, float oy, float oz, float dx,
                                                   float dy, float dz) {
  TriWalkRay r;
  r.a = dot3(dx, dy, dz, dx, dy, dz);
  const float o2 = fmaf(oz, oz, fmaf(oy, oy, ox * ox));
  r.da = __builtin_amdgcn_sqrtf(r.a) * 1.001f;
  r.ol = __builtin_amdgcn_sqrtf(o2) * 1.001f + 1e-30f;
  r.possible = r.a > 0.f && o2 < RT_INF;
  r.usable = r.a > 9.094947e-13f && r.a < 1.0995116e12f && o2 < 1e36f;
  return r;
}

// Every triangle for one lane's ray through the hierarchy (same result as the
// loop of renderer.cpp:156-166).
// `stack`: LdsStack or PrivateStack (above).
template <bool ANY_HIT, bool VERIFY, typename Stack>
__device__ __forceinline__ void walk_triangles(const SceneDev &sc, Stack &stack, float ox, float oy,
                                               float oz, float dx, float dy, float dz,
                                               float &len_t, int &idx_t, float &tPx, float &tPy,
                                               float &tPz, float outer_bound = RT_INF,
                                               unsigned long long *counters = nullptr) {
  const TriWalkRay r = tri_walk_ray(ox, oy, oz, dx, dy, dz);
  if (!r.possible)
    return;
  for (int k = 0; k < sc.n_tloose; k++) {
    const int i = sc.tloose[k];
    if (tri_fold(sc.tri[i], i, ox, oy, oz, dx, dy, dz, len_t, idx_t, tPx, tPy, tPz) && ANY_HIT)
      return;
  }
  if (sc.n_tleaf == 0)
    return;
  if (!r.usable) {
    for (int slot = 0; slot < sc.n_tleaf; slot++)
      if (tri_fold_leaf(sc.tleaf[slot], ox, oy, oz, dx, dy, dz, len_t, idx_t, tPx, tPy, tPz) &&
          ANY_HIT)
        return;
    return;
  }
  int sp = 0, node = 0;
  // outer_bound: a |P| beyond which a triangle hit cannot matter to the caller
  float bound = ANY_HIT ? RT_INF : fminf(len_t, outer_bound);
  unsigned n_bounds = 0, n_leaf = 0;
  while (true) {
    const TriNode N = sc.tbvh[node];
    int next = -1;
    float next_near = RT_INF;
    n_bounds += BVH_WIDTH;
#pragma unroll
    for (int c = 0; c < BVH_WIDTH; c++) {
      const int32_t link = N.link[c];
      float near;
      if (link == BVH_EMPTY ||
          !tri_touches<!ANY_HIT>(N.s[c], N.ns[c], ox, oy, oz, dx, dy, dz, r.a, r.da, r.ol, bound,
                                 near))
        continue;
      if (link >= 0) {
        int push = link;
        if (near < next_near) {
          push = next;
          next = link;
          next_near = near;
        }
        if (push >= 0)
          stack.push(sp, push);
      } else {
        const int slot = (~link) >> 3;
        n_leaf++;
        if (tri_fold_leaf(sc.tleaf[slot], ox, oy, oz, dx, dy, dz, len_t, idx_t, tPx, tPy, tPz)) {
          if (ANY_HIT)
            next = -1, sp = 0; // nothing more to find
          else
            bound = fminf(bound, len_t);
        }
      }
    }
    if (ANY_HIT && idx_t >= 0)
      break;
    if (next >= 0)
      node = next;
    else if (sp > 0)
      node = stack.pop(sp);
    else
      break;
  }
  if (VERIFY && counters) {
    atomicAdd(counters + 0, 1ull);
    atomicAdd(counters + 1, (unsigned long long)n_bounds);
    atomicAdd(counters + 2, (unsigned long long)n_leaf);
  }
}

// The sphere walk of one lane's ray with the lane's stack in LDS (stack_l = this
// thread's column, rows LDS_STACK_ROWS apart by 256 words; sc.bvh_lds must hold).
// The same visits, exits and order-free nearest-hit rule as the walk with a
// private stack in trace<> below; the visit itself is branch-free (stack_children)
// and the touched leaves are tested right away, each lane's next one per round
// (the rays of a wave that comes here travel together, so the rounds are dense and
// `bound` is tightened at once).  `bound`: a |P| beyond which a hit cannot matter.
template <bool ANY_HIT>
__device__ __forceinline__ void walk_spheres_lds(const SceneDev &sc, uint32_t *stack_l, bool go,
                                                 float a, float a4, float ox, float oy, float oz,
                                                 float dx, float dy, float dz, float bound,
                                                 float &len_s, int &idx_s, float &sPx, float &sPy,
                                                 float &sPz, unsigned &n_nodes, unsigned &n_leaf) {
  const SphWalkRay wr = sph_walk_ray(dx, dy, dz, a);
  int sp = 0, node = 0;
  while (__any(go)) {
    if (go) {
      const BvhNode N = sc.bvh[node];
      bool touch[BVH_WIDTH];
      float near_key[BVH_WIDTH], entry_key[BVH_WIDTH];
      bvh_visit2(N, ox, oy, oz, wr.ix, wr.iy, wr.iz, wr.ida, bound, touch, near_key, entry_key);
      n_nodes += BVH_WIDTH;
      const uint32_t next = stack_children(N, touch, entry_key, ANY_HIT, stack_l, sp);
      unsigned lm = 0;
#pragma unroll
      for (int c = 0; c < BVH_WIDTH; c++)
        lm |= ((touch[c] && N.link[c] < 0) ? 1u : 0u) << c;
      while (__any(lm != 0)) {
        if (lm != 0) {
          const unsigned c = (unsigned)__builtin_ctz(lm);
          lm &= lm - 1;
          const int32_t link = c == 0 ? N.link[0] : c == 1 ? N.link[1] : c == 2 ? N.link[2] : N.link[3];
          const int slot = (~link) >> 3;
          n_leaf += 1;
          const float4 g = sc.leaf_geo[slot];
          const int idx = sc.leaf_idx[slot];
          float Px, Py, Pz;
          if (sphere_exact(g.x, g.y, g.z, g.w, a, a4, ox, oy, oz, dx, dy, dz, Px, Py, Pz)) {
            const float len = length3(Px, Py, Pz);
            // renderer.cpp:170 scans in index order with a strict '>': the smallest
            // |P| wins and, among equals, the lowest index
            if (len < RT_INF && (len < len_s || (len == len_s && idx < idx_s))) {
              len_s = len;
              idx_s = idx;
              sPx = Px;
              sPy = Py;
              sPz = Pz;
              if (ANY_HIT)
                lm = 0;
              else
                bound = fminf(bound, len);
            }
          }
        }
      }
      if (next != NO_ENTRY && !(ANY_HIT && idx_s >= 0))
        node = (int)(next & 0xffffu);
      else
        go = false;
    }
  }
}

// The cell (of an R x R x 6 cube around the light) of the direction v from the light.
// Face = the largest component and its sign; the other two, divided by it, run over
// [-1, 1].  A direction within an ulp of a cell's edge may be given to the neighbour:
// the cells' cones are inflated by a hundred times that (make_beam).
template <int R>
__device__ __forceinline__ int light_cell(float vx, float vy, float vz) {
  const float ax = fabsf(vx), ay = fabsf(vy), az = fabsf(vz);
  int face;
  float m, u, w;
  if (ax >= ay && ax >= az) {
    face = vx < 0.f ? 1 : 0;
    m = ax;
    u = vy;
    w = vz;
  } else if (ay >= az) {
    face = vy < 0.f ? 3 : 2;
    m = ay;
    u = vx;
    w = vz;
  } else {
    face = vz < 0.f ? 5 : 4;
    m = az;
    u = vx;
    w = vy;
  }
  const float s = (0.5f * R) / m;
  int cu = (int)fmaf(u, s, 0.5f * R), cw = (int)fmaf(w, s, 0.5f * R);
  cu = cu < 0 ? 0 : (cu > R - 1 ? R - 1 : cu);
  cw = cw < 0 ? 0 : (cw > R - 1 ? R - 1 : cw);
  return (face * R + cw) * R + cu;
}

// TRI_TREE false: the caller knows that the scene has no triangle hierarchy, and the
// kernel then carries no triangle walk (its code and stack cost the sphere-only
// primary pass 3 % even when it never runs).
template <bool FROM_EYE, bool ANY_HIT, int BEAM, bool VERIFY = false, bool TRI_TREE = true>
__device__ __forceinline__ Hit trace(const SceneDev &sc, bool live, float ox,
                                     float oy, float oz, float dx, float dy,
                                     float dz, float Xx = 0.f, float Xy = 0.f,
                                     float Xz = 0.f, float Dk = 0.f,
                                     ScreenStats st = ScreenStats{nullptr, nullptr, nullptr},
                                     uint32_t *lds_stack = nullptr,
                                     const int32_t *bin_rec = nullptr, int cell_light = -1) {
  float len_t = RT_INF, len_s = RT_INF;
  int idx_t = -1, idx_s = -1;
  float tPx = 0, tPy = 0, tPz = 0, sPx = 0, sPy = 0, sPz = 0;

  // ---- triangles, renderer.cpp:156-166 ----
  // With a triangle hierarchy the triangles are walked AFTER the spheres (below):
  // the sphere walk is several times cheaper and its hit bounds the triangle walk.
  // (The plain-loop build carries no walk code: it is the brute-force yardstick.)
  const bool tri_walk = TRI_TREE && BEAM != RT_BEAM_NONE && sc.tbvh != nullptr;
  if (!tri_walk) {
    for (int i = 0; i < sc.t_size; i++) {
      const TriRec &T = sc.tri[i];
      float no = FROM_EYE ? T.p1_no.w
                          : dot3(T.n_d.x, T.n_d.y, T.n_d.z, ox, oy, oz);
      float Px, Py, Pz;
      if (live && triangle_test(T, no, ox, oy, oz, dx, dy, dz, Px, Py, Pz)) {
        float len = length3(Px, Py, Pz);
        if (len_t > len) {
          len_t = len;
          idx_t = i;
          tPx = Px;
          tPy = Py;
          tPz = Pz;
        }
      }
      if (ANY_HIT && !__any(live && idx_t < 0))
        break;
    }
  }

  // ---- spheres, renderer.cpp:168-177 ----
  const float a = dot3(dx, dy, dz, dx, dy, dz); // maths.cpp:127
  const float a4 = 4.0f * a;                    // maths.cpp:132, (4*a)*c
  bool seek = live && !(ANY_HIT && idx_t >= 0);

  bool screened = false;
  if (BEAM == RT_BEAM_APEX) {
    // rays with a == 0 or NaN never yield an accepted sphere hit (see above)
    const bool lv = seek && (a > 0.f);
    if (__any(lv)) {
      const float dscale = 9.5367431640625e-07f; // 2^-20
      const Beam B = make_beam(lv, a, dx, dy, dz, Xx, Xy, Xz, Dk, dscale);
      if (B.narrow) {
        screened = true;
        const int lane = threadIdx.x & 63;
        bool want = lv;
        for (int base = 0; base < sc.s_size; base += 64) {
          const int j = base + lane;
          const float4 g = sc.sph_geo[j < sc.s_size ? j : sc.s_size - 1];
          const bool keep = (j < sc.s_size) && beam_keeps(B, g);
          unsigned long long mask = __ballot(keep);
          if (VERIFY) {
            // count spheres the screen dropped although some ray's discriminant
            // test passes: must stay zero
            unsigned long long dropped = __ballot((j < sc.s_size) && !keep);
            while (dropped) {
              const int k = __builtin_ctzll(dropped);
              dropped &= dropped - 1;
              float cx = lane_value(g.x, k), cy = lane_value(g.y, k),
                    cz = lane_value(g.z, k), rr = lane_value(g.w, k);
              float Lx = ox - cx, Ly = oy - cy, Lz = oz - cz;
              float b = 2.0f * dot3(dx, dy, dz, Lx, Ly, Lz);
              float c = dot3(Lx, Ly, Lz, Lx, Ly, Lz) - rr;
              bool cand = lv && !(b * b < a4 * c);
              unsigned long long viol = __ballot(cand);
              if (viol && lane == 0)
                atomicAdd(st.dropped_but_candidate, 1ull);
            }
            if (lane == 0)
              atomicAdd(st.kept, (unsigned long long)__builtin_popcountll(mask));
          }
          while (mask) {
            const int k = __builtin_ctzll(mask);
            mask &= mask - 1;
            float cx = lane_value(g.x, k), cy = lane_value(g.y, k),
                  cz = lane_value(g.z, k), rr = lane_value(g.w, k);
            float Px, Py, Pz;
            if (want && sphere_exact(cx, cy, cz, rr, a, a4, ox, oy, oz, dx, dy,
                                     dz, Px, Py, Pz)) {
              float len = length3(Px, Py, Pz);
              if (len_s > len) { // renderer.cpp:170, strict
                len_s = len;
                idx_s = base + k;
                sPx = Px;
                sPy = Py;
                sPz = Pz;
                if (ANY_HIT)
                  want = false;
              }
            }
          }
          if (ANY_HIT && !__any(want))
            break;
        }
      }
    } else {
      screened = true; // no lane can get a sphere hit
    }
  }

  if (BEAM == RT_ACCEL_BVH && FROM_EYE && !ANY_HIT && !VERIFY && bin_rec != nullptr) {
    // Rays from the eye through one screen cell: the cell's list holds every sphere
    // that the discriminant test (maths.cpp:132-133) can pass for any of them
    // (k_bin_spheres), in rising index order, so the loop below is the plain loop of
    // renderer.cpp:168-177 without the spheres that cannot matter.  The list and
    // the records are the same for the whole wave: scalar loads.
    const int n = bin_rec[0];
    if (n >= 0) {
      screened = true;
      for (int k0 = 0; k0 < n; k0 += LIST_GROUP) {
        int idx[LIST_GROUP];
        float4 s[LIST_GROUP];
#pragma unroll
        for (int u = 0; u < LIST_GROUP; u++)
          idx[u] = bin_rec[1 + k0 + u];
#pragma unroll
        for (int u = 0; u < LIST_GROUP; u++)
          s[u] = sc.sph_prim[idx[u]];
#pragma unroll
        for (int u = 0; u < LIST_GROUP; u++) {
          const float b = 2.0f * dot3(dx, dy, dz, s[u].x, s[u].y, s[u].z); // maths.cpp:128
          const float bb = b * b, ac = a4 * s[u].w;                        // maths.cpp:132
          if (seek && k0 + u < n && !(bb < ac)) {
            float Px, Py, Pz;
            if (sphere_finish(a, b, s[u].w, bb, ac, ox, oy, oz, dx, dy, dz, Px, Py, Pz)) {
              const float len = length3(Px, Py, Pz);
              if (len_s > len) { // renderer.cpp:170, strict
                len_s = len;
                idx_s = idx[u];
                sPx = Px;
                sPy = Py;
                sPz = Pz;
              }
            }
          }
        }
      }
    }
  }

  if (BEAM == RT_ACCEL_BVH && !FROM_EYE && ANY_HIT && !VERIFY && cell_light >= 0 &&
      sc.lcells != nullptr) {
    // Shadow rays towards light `cell_light` (the apex X): the list of the cell that
    // holds the direction from the light to a ray's origin has every sphere the ray's
    // discriminant test can pass for (k_bin_light), provided the origin is within the
    // lists' reach.  The wave runs the reference's test on the lists of all the cells
    // its lanes fall into, every lane on every sphere (a sphere that is not in a lane's
    // own list cannot be hit by it, and the test says so).  A cell without a list or
    // an origin out of reach sends the lanes still seeking to the walk below.
    const bool lv = seek && (a > 0.f);
    if (!__any(lv && !(Dk <= sc.lcell_reach[cell_light]))) {
      const int32_t *const cells = sc.lcells + (size_t)cell_light * LCELL_COUNT * LCELL_WORDS;
      const int cell = lv ? light_cell<LCELL_R>(ox - Xx, oy - Xy, oz - Xz) : -1;
      bool want = lv, complete = true;
      unsigned long long todo = __ballot(lv);
      while (todo != 0 && complete) {
        const int c = __builtin_amdgcn_readlane(cell, __builtin_ctzll(todo));
        todo &= ~__ballot(cell == c);
        const int32_t *const rec = cells + (size_t)c * LCELL_WORDS;
        const int n = rec[0];
        complete = n >= 0;
        for (int k0 = 0; k0 < n && __any(want); k0 += LIST_GROUP) {
          int idx[LIST_GROUP];
          float4 g[LIST_GROUP];
#pragma unroll
          for (int u = 0; u < LIST_GROUP; u++)
            idx[u] = rec[1 + k0 + u];
#pragma unroll
          for (int u = 0; u < LIST_GROUP; u++)
            g[u] = sc.sph_geo[idx[u]];
#pragma unroll
          for (int u = 0; u < LIST_GROUP; u++) {
            float Px, Py, Pz;
            if (want && k0 + u < n &&
                sphere_exact(g[u].x, g[u].y, g[u].z, g[u].w, a, a4, ox, oy, oz, dx, dy, dz, Px, Py,
                             Pz)) {
              const float len = length3(Px, Py, Pz);
              if (len_s > len) { // renderer.cpp:170: a |P| that is not below +inf is no hit
                len_s = len;
                idx_s = idx[u];
                sPx = Px;
                sPy = Py;
                sPz = Pz;
                want = false;
              }
            }
          }
        }
      }
      if (complete)
        screened = true;
      else
        seek = seek && idx_s < 0;
    }
  }

  if (BEAM == RT_ACCEL_BVH && sc.bvh != nullptr && !screened) {
    // Per-lane walk.  Rays with a == 0 / NaN never yield an accepted sphere hit;
    // an a far outside [2^-40, 2^40] sends the whole wave to the plain loop.
    const bool sane = a > 9.094947e-13f && a < 1.0995116e12f;
    if (!__any(seek && a > 0.f && !sane)) {
      screened = true;
      if (lds_stack != nullptr && sc.bvh_lds) {
        // a sphere can only win against the nearest triangle by being strictly
        // nearer (renderer.cpp:181)
        unsigned n_nodes = 0, n_leaf = 0;
        walk_spheres_lds<ANY_HIT>(sc, lds_stack, seek && a > 0.f, a, a4, ox, oy, oz, dx, dy, dz,
                                  ANY_HIT ? RT_INF : len_t, len_s, idx_s, sPx, sPy, sPz, n_nodes,
                                  n_leaf);
        if (VERIFY && st.walk && seek && a > 0.f) {
          atomicAdd(st.walk + 0, 1ull);
          atomicAdd(st.walk + 1, (unsigned long long)n_nodes);
          atomicAdd(st.walk + 2, (unsigned long long)n_leaf);
        }
      } else if (seek && a > 0.f) {
        PrivateStack stack;
        int sp = 0;
        int node = 0;
        bool done = false;
        unsigned n_nodes = 0, n_leaf = 0;
        const SphWalkRay wr = sph_walk_ray(dx, dy, dz, a);
        // a sphere can only win against the nearest triangle by being strictly
        // nearer (renderer.cpp:181)
        float bound = ANY_HIT ? RT_INF : len_t;
        while (!done) {
          const BvhNode N = sc.bvh[node];
          n_nodes += BVH_WIDTH;
          int next = -1;
          float next_near = RT_INF;
          bool touch[BVH_WIDTH];
          float keys[BVH_WIDTH];
          bvh_visit<false>(N, ox, oy, oz, wr.ix, wr.iy, wr.iz, wr.ida, bound, touch, keys);
#pragma unroll
          for (int c = 0; c < BVH_WIDTH; c++) {
            const int32_t link = N.link[c];
            if (!touch[c])
              continue;
            if (link >= 0) {
              // keep the child nearest to the world origin for the next visit (it
              // tightens `bound`), stack the others
              const float key = keys[c];
              int push = link;
              if (key < next_near) {
                push = next;
                next = link;
                next_near = key;
              }
              if (push >= 0)
                stack.push(sp, push);
            } else {
              const int slot = (~link) >> 3;
              n_leaf += 1;
              const float4 g = sc.leaf_geo[slot];
              float Px, Py, Pz;
              if (sphere_exact(g.x, g.y, g.z, g.w, a, a4, ox, oy, oz, dx, dy, dz, Px, Py,
                               Pz)) {
                const float len = length3(Px, Py, Pz);
                const int idx = sc.leaf_idx[slot];
                // renderer.cpp:170 scans in index order with a strict '>':
                // the smallest |P| wins and, among equals, the lowest index
                if (len < RT_INF && (len < len_s || (len == len_s && idx < idx_s))) {
                  len_s = len;
                  idx_s = idx;
                  sPx = Px;
                  sPy = Py;
                  sPz = Pz;
                  if (ANY_HIT)
                    done = true;
                  else
                    bound = fminf(bound, len);
                }
              }
            }
          }
          if (!done) {
            if (next >= 0)
              node = next;
            else if (sp > 0)
              node = stack.pop(sp);
            else
              done = true;
          }
        }
        if (VERIFY && st.walk) {
          atomicAdd(st.walk + 0, 1ull);
          atomicAdd(st.walk + 1, (unsigned long long)n_nodes);
          atomicAdd(st.walk + 2, (unsigned long long)n_leaf);
        }
      }
    }
  }

  if (!screened && (!ANY_HIT || __any(seek))) {
    const float4 *__restrict__ rec = FROM_EYE ? sc.sph_prim : sc.sph_geo;
    for (int i = 0; i < sc.s_pad; i += SPH_UNROLL) {
      float b[SPH_UNROLL], c[SPH_UNROLL], bb[SPH_UNROLL], ac[SPH_UNROLL];
      unsigned long long cand = 0;
#pragma unroll
      for (int u = 0; u < SPH_UNROLL; u++) {
        const float4 s = rec[i + u];
        float Lx, Ly, Lz;
        if (FROM_EYE) {
          Lx = s.x;
          Ly = s.y;
          Lz = s.z;
          c[u] = s.w;
        } else {
          Lx = ox - s.x; // maths.cpp:126
          Ly = oy - s.y;
          Lz = oz - s.z;
          c[u] = dot3(Lx, Ly, Lz, Lx, Ly, Lz) - s.w; // maths.cpp:129
        }
        b[u] = 2.0f * dot3(dx, dy, dz, Lx, Ly, Lz); // maths.cpp:128
        bb[u] = b[u] * b[u];
        ac[u] = a4 * c[u];
        // maths.cpp:132-133: discr = bb - ac is negative exactly when bb < ac
        // (a float difference has the sign of the exact difference).
        cand |= __ballot(!(bb[u] < ac[u]));
      }
      if (cand) {
#pragma unroll
        for (int u = 0; u < SPH_UNROLL; u++) {
          if (seek && !(bb[u] < ac[u]) && (i + u) < sc.s_size) {
            float Px, Py, Pz;
            if (sphere_finish(a, b[u], c[u], bb[u], ac[u], ox, oy, oz, dx, dy,
                              dz, Px, Py, Pz)) {
              float len = length3(Px, Py, Pz);
              if (len_s > len) { // renderer.cpp:170, strict
                len_s = len;
                idx_s = i + u;
                sPx = Px;
                sPy = Py;
                sPz = Pz;
                if (ANY_HIT)
                  seek = false;
              }
            }
          }
        }
        if (ANY_HIT && !__any(seek))
          break;
      }
    }
  }

  if (tri_walk && live && !(ANY_HIT && idx_s >= 0)) {
    // a triangle only matters at |P| <= the nearest sphere's (renderer.cpp:181);
    // the walk's prune comparison is strict, so a tie survives
    // (the sphere walk above is over: its LDS column, if there is one, is free)
    if (BEAM == RT_ACCEL_BVH) { // every such caller passes its LDS column
      LdsStack stack;
      stack.lds = (int *)lds_stack;
      walk_triangles<ANY_HIT, VERIFY>(sc, stack, ox, oy, oz, dx, dy, dz, len_t, idx_t, tPx, tPy,
                                      tPz, len_s, st.twalk);
    } else {
      PrivateStack stack;
      walk_triangles<ANY_HIT, VERIFY>(sc, stack, ox, oy, oz, dx, dy, dz, len_t, idx_t, tPx, tPy,
                                      tPz, len_s, st.twalk);
    }
  }

  // ---- renderer.cpp:179-192 ----
  Hit h;
  h.kind = 0;
  h.index = -1;
  h.Px = h.Py = h.Pz = 0;
  if (idx_t < 0 && idx_s < 0)
    return h;
  if (len_t <= len_s) {
    h.kind = 1;
    h.index = idx_t;
    h.Px = tPx;
    h.Py = tPy;
    h.Pz = tPz;
  } else if (len_t > len_s) {
    h.kind = 2;
    h.index = idx_s;
    h.Px = sPx;
    h.Py = sPy;
    h.Pz = sPz;
  }
  return h;
}

} // namespace rt
