// rt_kernels.hip — the HIP kernels of the trace/shade path (gfx950, wave64).
//
// Compiled with -ffp-contract=off: the reference image is defined by unfused
// IEEE arithmetic, and a single contracted multiply-add flips hit ids.
//
// Kernels
//   k_prep_scene      once per scene: flat arrays -> sphere/triangle records
//   k_prep_eye        once per eye position: eye-relative sphere/triangle terms
//   k_gather_leaves   once per scene: sphere records in hierarchy leaf order
//   k_render          one lane per pixel, one 8x8 pixel tile per wave:
//                     ray generation, nearest hit, Phong + shadow rays, pixel
//                     write (renderer.cpp:28-140), first-bounce rays queued
//   k_walk, k_shade   one bounce level as three launches (rt_walk.h)
//   k_bounce          one bounce level in one launch (no hierarchy / self-check)
//   k_compose         adds the bounce hits' integer light to the framebuffer
//   k_check           one lane per caller-supplied ray (renderer.cpp:145-193)
#include "rt_device.h"
#include "rt_kernels.h"

namespace rt {

// ---------------------------------------------------------------------------
// Scene preparation
// ---------------------------------------------------------------------------
__global__ void k_prep_scene(PrepArgs a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.s_pad) {
    float4 g;
    if (i < a.s_size) {
      float r = a.radius[i];
      g = make_float4(a.spheres[3 * i], a.spheres[3 * i + 1],
                      a.spheres[3 * i + 2], r * r); // maths.cpp:129 radius*radius
      uchar4 c;
      c.x = a.color_sphere[3 * i];
      c.y = a.color_sphere[3 * i + 1];
      c.z = a.color_sphere[3 * i + 2];
      c.w = 0;
      a.col_sph[i] = c;
    } else {
      // padding: c = L.L - (-inf) = +inf, so bb < (4a)*c always holds
      g = make_float4(0.f, 0.f, 0.f, -RT_INF);
    }
    a.sph_geo[i] = g;
  }
  if (i < a.t_size) {
    const float *p1 = a.tris + 9 * i, *p2 = p1 + 3, *p3 = p1 + 6;
    float nx, ny, nz;
    tri_normal(p1, p2, p3, nx, ny, nz);
    TriRec T;
    T.n_d = make_float4(nx, ny, nz, dot3(nx, ny, nz, p1[0], p1[1], p1[2]));
    T.p1_no = make_float4(p1[0], p1[1], p1[2], 0.f);
    T.p2 = make_float4(p2[0], p2[1], p2[2], 0.f);
    T.p3 = make_float4(p3[0], p3[1], p3[2], 0.f);
    T.e0 = make_float4(p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2], 0.f);
    T.e1 = make_float4(p3[0] - p2[0], p3[1] - p2[1], p3[2] - p2[2], 0.f);
    T.e2 = make_float4(p1[0] - p3[0], p1[1] - p3[1], p1[2] - p3[2], 0.f);
    a.tri[i] = T;
    uchar4 c;
    c.x = a.color_tri[3 * i];
    c.y = a.color_tri[3 * i + 1];
    c.z = a.color_tri[3 * i + 2];
    c.w = 0;
    a.col_tri[i] = c;
  }
}

__global__ void k_prep_eye(EyeArgs a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.s_pad) {
    float4 g = a.sph_geo[i];
    float4 p;
    if (i < a.s_size) {
      float Lx = a.ox - g.x, Ly = a.oy - g.y, Lz = a.oz - g.z; // maths.cpp:126
      p = make_float4(Lx, Ly, Lz, dot3(Lx, Ly, Lz, Lx, Ly, Lz) - g.w);
    } else {
      p = make_float4(0.f, 0.f, 0.f, RT_INF);
    }
    a.sph_prim[i] = p;
  }
  if (i < a.t_size) {
    float4 n = a.tri[i].n_d;
    a.tri[i].p1_no.w = dot3(n.x, n.y, n.z, a.ox, a.oy, a.oz); // maths.cpp:78
  }
}

// ---------------------------------------------------------------------------
// Shading of one hit (renderer.cpp:62-128)
// ---------------------------------------------------------------------------
// renderer.cpp:121-126 with clamp<uint16_t> (maths.hpp:26-28)
__device__ __forceinline__ uint32_t accumulate_u8(uint32_t old, float add) {
  uint32_t v = (uint32_t)((float)old + add) & 0xffffu;
  return v > 255u ? 255u : v;
}

// lang^20 in double: renderer.cpp:106 calls pow(double, 20.0).  Five exact-order
// double multiplies stay within a few ulp(double) of it, far below the float
// rounding that follows.
__device__ __forceinline__ float pow20(float lang) {
  double x = (double)lang;
  double x2 = x * x;
  double x4 = x2 * x2;
  double x5 = x4 * x;
  double x10 = x5 * x5;
  return (float)(x10 * x10);
}

struct Surface {
  float nx, ny, nz;
  uint32_t c0, c1, c2;
};

__device__ __forceinline__ Surface surface_at(const SceneDev &sc, const Hit &h) {
  Surface s;
  uchar4 col;
  if (h.kind == 1) { // renderer.cpp:66-70: the unit normal, normalised again (:76)
    float4 n = sc.tri[h.index].n_d;
    s.nx = n.x;
    s.ny = n.y;
    s.nz = n.z;
    col = sc.col_tri[h.index];
  } else if (h.kind == 3) { // a sphere named by its leaf slot (k_trace)
    const float4 g = sc.leaf_geo[h.index];
    s.nx = h.Px - g.x;
    s.ny = h.Py - g.y;
    s.nz = h.Pz - g.z;
    const uint32_t bits = __float_as_uint(sc.leaf_mat[h.index].w);
    col = make_uchar4(bits & 0xffu, (bits >> 8) & 0xffu, (bits >> 16) & 0xffu, 0);
  } else { // renderer.cpp:71-74
    float4 g = sc.sph_geo[h.index];
    s.nx = h.Px - g.x;
    s.ny = h.Py - g.y;
    s.nz = h.Pz - g.z;
    col = sc.col_sph[h.index];
  }
  normalize3(s.nx, s.ny, s.nz); // renderer.cpp:76
  s.c0 = col.x;
  s.c1 = col.y;
  s.c2 = col.z;
  return s;
}

// One light, renderer.cpp:85-115.  Returns angle; fills fColor and the unit
// direction from P to the light.
__device__ __forceinline__ float phong_light(const Surface &s, const Hit &h,
                                             float dx, float dy, float dz,
                                             float4 light, float fColor[3],
                                             float &lx, float &ly, float &lz) {
  lx = light.x - h.Px;
  ly = light.y - h.Py;
  lz = light.z - h.Pz;
  normalize3(lx, ly, lz);
  float angle = dot3(s.nx, s.ny, s.nz, lx, ly, lz);
  angle = angle < 0 ? 0 : angle;
  float mx = 0.f - lx, my = 0.f - ly, mz = 0.f - lz; // renderer.cpp:97-99
  float rx, ry, rz;
  reflect3(mx, my, mz, s.nx, s.ny, s.nz, rx, ry, rz);
  float vx = 0.f - dx, vy = 0.f - dy, vz = 0.f - dz; // renderer.cpp:102-103
  float lang = dot3(rx, ry, rz, vx, vy, vz);
  lang = lang < 0 ? 0 : lang;
  float sp = pow20(lang);
  const uint32_t col[3] = {s.c0, s.c1, s.c2};
#pragma unroll
  for (int k = 0; k < 3; k++) {
    uint32_t spec = (uint32_t)((float)col[k] * sp) & 0xffu;            // :108-110
    fColor[k] = (float)((double)spec * 0.3 + (double)col[k] * 0.7);    // :113-115
  }
  return angle;
}

// ---------------------------------------------------------------------------
// Material bounces (an extension; the reference has none)
// ---------------------------------------------------------------------------
// A hit on a primitive with material (kr, kt, ior) at path weight w:
//   - its Phong light is weighted by wl = w * ((1 - kr) - kt); below the primary
//     hit a light whose three integer contributions are all zero casts no shadow ray
//   - if depth < max_depth and w*kr >= 2^-8: a mirror ray from P along
//     normalize(reflect(dir, n)), weight w*kr
//   - if depth < max_depth and w*kt >= 2^-8: a refracted ray (Snell; total
//     internal reflection falls back to the mirror direction), weight w*kt
// The primary hit accumulates into the pixel as renderer.cpp:121-126 does.
// Every later hit adds trunc(fColor*angle*wl) as an integer and the pixel is the
// saturating sum, so the order in which bounce rays are processed cannot change
// the image: they are appended to a ray queue in HBM and each bounce level runs
// as its own launch with one queued ray per lane (all lanes busy, whatever the
// depth of the individual paths).  Anything lighter than 2^-8 cannot reach one
// LSB, so dropping it is exact.
#define RT_MIN_WEIGHT 0.00390625f

__device__ __forceinline__ float4 material_at(const SceneDev &sc, const Hit &h) {
  if (h.kind == 1)
    return sc.mat_tri ? sc.mat_tri[h.index] : make_float4(0.f, 0.f, 0.f, 0.f);
  if (h.kind == 3)
    return sc.leaf_mat[h.index];
  return sc.mat_sph ? sc.mat_sph[h.index] : make_float4(0.f, 0.f, 0.f, 0.f);
}

__device__ __forceinline__ void refract3(float dx, float dy, float dz, float nx,
                                         float ny, float nz, float ior, float &rx,
                                         float &ry, float &rz) {
  float cosi = -dot3(dx, dy, dz, nx, ny, nz);
  float mx = nx, my = ny, mz = nz, eta;
  if (cosi < 0) { // leaving the medium
    mx = -nx;
    my = -ny;
    mz = -nz;
    cosi = -cosi;
    eta = ior;
  } else {
    eta = 1.0f / ior;
  }
  float k = 1.0f - (eta * eta) * (1.0f - cosi * cosi);
  if (k < 0) {
    reflect3(dx, dy, dz, nx, ny, nz, rx, ry, rz);
  } else {
    float f = eta * cosi - sqrtf(k);
    rx = eta * dx + f * mx;
    ry = eta * dy + f * my;
    rz = eta * dz + f * mz;
  }
  normalize3(rx, ry, rz);
}

// The segment a block appends to and consumes from.
__device__ __forceinline__ uint32_t my_segment() {
  return (blockIdx.y * gridDim.x + blockIdx.x) % QUEUE_SEGMENTS;
}

// Wave-aggregated append: one atomic per wave, lanes write consecutive slots of
// the block's segment.  Lanes that find the segment full move on to the next
// one; only when every segment is full is the ray dropped and `overflow` set.
__device__ __forceinline__ void queue_push_raw(const RayQueue &q, bool want, float4 a,
                                               float4 b) {
  const int lane = threadIdx.x & 63;
  uint32_t seg = my_segment();
  for (int attempt = 0; attempt < QUEUE_SEGMENTS; attempt++) {
    const unsigned long long m = __ballot(want);
    if (!m)
      return;
    const int leader = __builtin_ctzll(m);
    uint32_t base = 0;
    if (lane == leader)
      base = atomicAdd(q.count + seg * CTL_WORDS, (uint32_t)__builtin_popcountll(m));
    base = __builtin_amdgcn_readlane(base, leader);
    if (want) {
      const uint32_t slot = base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
      if (slot < q.seg_capacity) {
        const size_t g = (size_t)seg * q.seg_capacity + slot;
        q.items[2 * g] = a;
        q.items[2 * g + 1] = b;
        want = false;
      }
    }
    seg = (seg + 1) % QUEUE_SEGMENTS;
  }
  if (want)
    *q.overflow = 1u;
}

__device__ __forceinline__ void queue_push(const RayQueue &q, bool want, float ox,
                                           float oy, float oz, float dx, float dy,
                                           float dz, float w, uint32_t pixel,
                                           uint32_t depth) {
  queue_push_raw(q, want, make_float4(ox, oy, oz, w),
                 make_float4(dx, dy, dz, __uint_as_float((pixel << 6) | depth)));
}

// Shade the hits of one wave (renderer.cpp:62-128) and queue their bounces.
// PRIMARY: the hit of the pixel's own ray; light is accumulated into fb0..2 with
// the reference's float form.  Otherwise light is added to the pixel's integer
// side sums.  Returns the number of shadow rays this lane traced.
template <bool PRIMARY, bool MATERIALS, int BEAM, bool VERIFY, bool TRI_TREE = true>
__device__ __forceinline__ uint32_t
shade_hits(const SceneDev &sc, const BounceDev &bd, bool hit, const Hit &h,
           float dx, float dy, float dz, float w, uint32_t pixel, uint32_t depth,
           uint32_t &fb0, uint32_t &fb1, uint32_t &fb2, ScreenStats st,
           uint32_t *lds_stack = nullptr) {
  uint32_t n_rays = 0;
  if (!__any(hit))
    return 0;
  Surface s;
  float kr = 0.f, kt = 0.f, ior = 1.f, wl = w;
  if (hit) {
    s = surface_at(sc, h);
    if (MATERIALS) {
      float4 m = material_at(sc, h);
      kr = m.x;
      kt = m.y;
      ior = m.z > 0.f ? m.z : 1.0f;
      wl = w * ((1.0f - kr) - kt);
    }
  }
  const bool lit = hit && (PRIMARY || wl >= RT_MIN_WEIGHT);
  if (__any(lit)) {
    for (int l = 0; l < sc.l_size; l++) { // renderer.cpp:82
      float fColor[3] = {0, 0, 0};
      float lx = 0, ly = 0, lz = 0, angle = 0, dist = 0;
      const float4 light = sc.lights[l];
      if (lit) {
        angle = phong_light(s, h, dx, dy, dz, light, fColor, lx, ly, lz);
        dist = length3(light.x - h.Px, light.y - h.Py, light.z - h.Pz);
      }
      float a0 = fColor[0] * angle, a1 = fColor[1] * angle, a2 = fColor[2] * angle;
      if (MATERIALS) {
        a0 *= wl;
        a1 *= wl;
        a2 *= wl;
      }
      // Below the primary hit a light that cannot add one unit to any channel casts
      // no shadow ray (nothing would be added whatever the ray found); the primary
      // hit always casts it, as renderer.cpp:119 does.
      const bool cast = lit && (PRIMARY || ((uint32_t)a0 | (uint32_t)a1 | (uint32_t)a2) != 0);
      // renderer.cpp:119: shadow ray from exactly P, no distance limit.  All
      // shadow rays of the wave aim at the light: that is the beam's apex.
      Hit sh = trace<false, true, BEAM, VERIFY, TRI_TREE>(sc, cast, h.Px, h.Py, h.Pz, lx, ly,
                                                lz, light.x, light.y, light.z,
                                                dist, st, lds_stack, nullptr,
                                                l < sc.lcell_lights ? l : -1);
      if (cast) {
        n_rays++;
        if (sh.kind == 0) {
          if (PRIMARY) {
            fb0 = accumulate_u8(fb0, a0);
            fb1 = accumulate_u8(fb1, a1);
            fb2 = accumulate_u8(fb2, a2);
          } else {
            const unsigned long long c0 = (uint32_t)a0, c1 = (uint32_t)a1, c2 = (uint32_t)a2;
            const unsigned long long v = c0 | (c1 << 21) | (c2 << 42);
            if (v)
              atomicAdd(bd.extra + pixel, v);
          }
        }
      }
    }
  }
  if (MATERIALS) {
    const bool deeper = hit && (int)depth < bd.max_depth;
    const bool want_t = deeper && kt > 0.f && (w * kt >= RT_MIN_WEIGHT);
    const bool want_r = deeper && kr > 0.f && (w * kr >= RT_MIN_WEIGHT);
    if (__any(want_t)) {
      float rx = 0, ry = 0, rz = 0;
      if (want_t)
        refract3(dx, dy, dz, s.nx, s.ny, s.nz, ior, rx, ry, rz);
      queue_push(bd.out, want_t, h.Px, h.Py, h.Pz, rx, ry, rz, w * kt, pixel, depth + 1);
    }
    if (__any(want_r)) {
      float rx = 0, ry = 0, rz = 0;
      if (want_r) {
        reflect3(dx, dy, dz, s.nx, s.ny, s.nz, rx, ry, rz);
        normalize3(rx, ry, rz);
      }
      queue_push(bd.out, want_r, h.Px, h.Py, h.Pz, rx, ry, rz, w * kr, pixel, depth + 1);
    }
  }
  return n_rays;
}

// Rays traced by this wave, added to the frame's count: a wave sum, then one atomic
// per wave on one of RAY_PARTS partial counters (64 bytes apart, chosen by block).
// Atomics on one address retire one after the other, about 11 ns each: the 129 600
// waves of a 4K primary pass adding to a single word took 1.4 ms, longer than the
// pass itself.  k_sum_rays folds the partial counters into the caller's word.
__device__ __forceinline__ void add_ray_count(uint64_t *parts, uint32_t n) {
  if (!parts)
    return;
  uint32_t v = n;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0 && v) {
    const uint32_t part = (blockIdx.y * gridDim.x + blockIdx.x) % RAY_PARTS;
    atomicAdd(reinterpret_cast<unsigned long long *>(parts) + part * RAY_PART_STRIDE,
              (unsigned long long)v);
  }
}

// One block of RAY_PARTS threads: total += the partial counters, which are cleared
// for the next frame.
__global__ void k_sum_rays(uint64_t *parts, uint64_t *total) {
  unsigned long long *p = reinterpret_cast<unsigned long long *>(parts) + threadIdx.x * RAY_PART_STRIDE;
  unsigned long long v = *p;
  *p = 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0 && v)
    atomicAdd(reinterpret_cast<unsigned long long *>(total), v);
}

// ---------------------------------------------------------------------------
// k_render
// ---------------------------------------------------------------------------
// Work decomposition: a wave owns an 8x8 pixel tile (lane = 8*row + col inside
// the tile), a 256-thread block owns four horizontally adjacent tiles, so the
// block's pixel writes cover 128 contiguous bytes per row.  blockIdx.y counts
// the tile rows this call owns; bands of band_rows rows are dealt round-robin
// over band_count callers (row sharding, renderer.cpp:27-28).
// Five waves per SIMD (96 registers, a hundred bytes more scratch than at 123
// registers and four waves): the pass waits on dependent node reads, and the fifth
// wave is worth more than the spills cost (depth-1 4K frame 1.01 -> 0.96 ms).
template <int ACCEL, bool VERIFY, bool MATERIALS, bool TRI_TREE>
__global__ __launch_bounds__(256, 5) void k_render(SceneDev sc, CameraDev cam,
                                                FrameDev f, BounceDev bd,
                                                ScreenStats st) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int tiles_per_band = f.band_rows >> 3;
  const int band_local = blockIdx.y / tiles_per_band;
  const int tile_in_band = blockIdx.y - band_local * tiles_per_band;
  const int band = band_local * f.band_count + f.band_index;
  const int i = band * f.band_rows + tile_in_band * 8 + (lane >> 3);
  const int j = (blockIdx.x * 4 + wave) * 8 + (lane & 7);
  const bool inside = (i < f.h) && (j < f.w);

  // ---- ray generation, renderer.cpp:34-44 (double, rounded once) ----
  float px = (float)((2 * ((j + 0.5) / (double)(float)f.w) - 1) * cam.tan_half *
                     (double)cam.aspect);
  float py = (float)((1 - 2 * ((i + 0.5) / (double)(float)f.h)) * cam.tan_half);
  float dx = px - cam.ox, dy = py - cam.oy, dz = cam.plane_z - cam.oz;
  normalize3(dx, dy, dz);

  constexpr int BM = ACCEL;
  // the lanes' stacks of the sphere walk (rt_device.h, walk_spheres_lds)
  __shared__ uint32_t walk_words[ACCEL == RT_ACCEL_BVH ? LDS_STACK_ROWS * 256 : 1];
  uint32_t *const lds_stack = ACCEL == RT_ACCEL_BVH ? walk_words + threadIdx.x : nullptr;
  // the screen cell of the wave's tile (a tile lies in one cell)
  const int32_t *bin_rec = nullptr;
  if (ACCEL == RT_ACCEL_BVH && sc.bins != nullptr) {
    const int ti = band * f.band_rows + tile_in_band * 8, tj = (blockIdx.x * 4 + wave) * 8;
    if (ti < f.h && tj < f.w)
      bin_rec = sc.bins + (size_t)__builtin_amdgcn_readfirstlane(
                              (ti / BIN_PX) * sc.bins_x + tj / BIN_PX) * BIN_WORDS;
  }
  Hit h = trace<true, false, BM, VERIFY, TRI_TREE>(sc, inside, cam.ox, cam.oy, cam.oz, dx,
                                         dy, dz, cam.ox, cam.oy, cam.oz, 0.f, st, lds_stack,
                                         bin_rec);

  uint32_t fb0 = 0, fb1 = 0, fb2 = 0; // renderer.cpp:50-52
  uint32_t n_rays = inside ? 1u : 0u;
  const bool hit = inside && h.kind != 0;
  const size_t p = (size_t)f.w * (inside ? i : 0) + (inside ? j : 0);

  if (f.unlit) {
    // renderer.cpp:130-137: clamp<int>(0 + colour, 0, 255) is the colour itself
    if (hit) {
      const uchar4 col = h.kind == 1 ? sc.col_tri[h.index] : sc.col_sph[h.index];
      fb0 = col.x;
      fb1 = col.y;
      fb2 = col.z;
    }
  } else {
    n_rays += shade_hits<true, MATERIALS, BM, VERIFY, TRI_TREE>(sc, bd, hit, h, dx, dy, dz, 1.0f,
                                                      (uint32_t)p, 1u, fb0, fb1, fb2, st,
                                                      lds_stack);
  }

  if (inside) {
    uchar4 o;
    o.x = (unsigned char)fb0;
    o.y = (unsigned char)fb1;
    o.z = (unsigned char)fb2;
    o.w = 255; // renderer.cpp:53
    reinterpret_cast<uchar4 *>(f.fb)[p] = o;
    if (f.ids)
      f.ids[p] = h.kind == 0 ? -1 : (h.kind == 1 ? h.index : sc.t_size + h.index);
  }
  add_ray_count(f.ray_count, n_rays);
}

// ---------------------------------------------------------------------------
// k_bin_spheres — spheres by screen cell, for the rays that leave the eye
// ---------------------------------------------------------------------------
// One wave per cell of PX x PX pixels.  The rays of the cell's four corner
// pixels are generated exactly as k_render generates them; the pixels of the cell
// lie in the convex hull of the corners on the image plane and the set of points
// of a plane seen from the eye within an angle of an axis is convex, so the cone of
// make_beam around the four corner rays (inflated there by far more than the
// rounding of a direction) holds every ray of the cell.  The spheres are screened
// against it 64 at a time with the test of the beam screen (beam_keeps: never drops
// a sphere whose discriminant test can pass for a line of the cone through the
// apex) and the kept ones listed in rising index order.
// Two levels, so that a change of camera costs a few microseconds: cells of
// BIN_SUPER_PX pixels screen all the spheres (FROM_LIST false), the cells of BIN_PX
// pixels then only their parent's list (a cell's rays are among its parent's, so
// the parent's list holds every sphere one of them can hit, and the cell's own
// screen drops none of those).  A parent without a list leaves its children
// without one: those tiles walk the hierarchy.
template <int PX, int WORDS, bool FROM_LIST>
__global__ __launch_bounds__(64) void k_bin_spheres(SceneDev sc, CameraDev cam, int w, int h,
                                                    int32_t *out, int out_x,
                                                    const int32_t *parent, int parent_x) {
  const int lane = threadIdx.x;
  const int j0 = blockIdx.x * PX, i0 = blockIdx.y * PX;
  const int j1 = j0 + PX - 1 < w ? j0 + PX - 1 : w - 1;
  const int i1 = i0 + PX - 1 < h ? i0 + PX - 1 : h - 1;
  const int i = (lane & 2) ? i1 : i0, j = (lane & 1) ? j1 : j0;
  // renderer.cpp:34-44 as in k_render
  float px = (float)((2 * ((j + 0.5) / (double)(float)w) - 1) * cam.tan_half * (double)cam.aspect);
  float py = (float)((1 - 2 * ((i + 0.5) / (double)(float)h)) * cam.tan_half);
  float dx = px - cam.ox, dy = py - cam.oy, dz = cam.plane_z - cam.oz;
  normalize3(dx, dy, dz);
  const bool lv = lane < 4;
  const float a = dot3(dx, dy, dz, dx, dy, dz);
  const Beam B = make_beam(lv, a, dx, dy, dz, cam.ox, cam.oy, cam.oz, 0.f, 9.5367431640625e-07f);
  // The first level cuts the spheres into BIN_SUPER_SLICES ranges, one wave and one
  // list per range (blockIdx.z), so that a change of camera is not 157 dependent
  // rounds of 64 spheres for a few hundred waves; the second level reads a parent's
  // lists one after the other, which keeps the indices in rising order.
  int32_t *const rec = out + ((size_t)blockIdx.y * out_x + blockIdx.x) * WORDS +
                       (FROM_LIST ? 0 : blockIdx.z * (WORDS / BIN_SUPER_SLICES));
  constexpr int CAP = (FROM_LIST ? WORDS : WORDS / BIN_SUPER_SLICES) - LIST_GROUP;
  if (!B.narrow) {
    if (lane == 0)
      rec[0] = -1;
    return;
  }
  int n = 0;
  const int per_slice = (sc.s_size + BIN_SUPER_SLICES - 1) / BIN_SUPER_SLICES;
  for (int z = 0; z < (FROM_LIST ? BIN_SUPER_SLICES : 1); z++) {
    const int32_t *src = nullptr;
    int first = 0, n_src;
    if (FROM_LIST) {
      src = parent + ((size_t)(i0 / BIN_SUPER_PX) * parent_x + j0 / BIN_SUPER_PX) * BIN_SUPER_WORDS +
            z * (BIN_SUPER_WORDS / BIN_SUPER_SLICES);
      n_src = src[0];
      if (n_src < 0) {
        if (lane == 0)
          rec[0] = -1;
        return;
      }
    } else {
      first = blockIdx.z * per_slice;
      n_src = sc.s_size - first < per_slice ? sc.s_size - first : per_slice;
    }
    for (int base = 0; base < n_src; base += 64) {
      const int k = base + lane;
      int idx = first + (k < n_src ? k : n_src - 1);
      if (FROM_LIST)
        idx = src[1 + idx];
      const float4 g = sc.sph_geo[idx];
      const bool keep = k < n_src && beam_keeps(B, g);
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const int pos = n + __builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (pos < CAP)
          rec[1 + pos] = idx;
      }
      n += __builtin_popcountll(m);
    }
  }
  if (n <= CAP && lane < LIST_GROUP && n + lane < ((n + LIST_GROUP - 1) & ~(LIST_GROUP - 1)))
    rec[1 + n + lane] = 0; // the readers take LIST_GROUP entries at a time
  if (lane == 0)
    rec[0] = n <= CAP ? n : -1;
}

// ---------------------------------------------------------------------------
// k_bin_light — spheres by direction from a light, for the shadow rays aimed at it
// ---------------------------------------------------------------------------
// One wave per cell of an R x R x 6 cube of directions around the light X (grid
// R x R x 6).  A cell is a rectangle on a face of the cube, so the cone of
// make_beam around its four corner directions holds every direction of the cell
// (as for k_bin_spheres); the lines it stands for are the shadow rays
// {P + t d}, d = normalize(X - P) in float, that start within `reach` of X in that
// cell: make_beam's slack K and widening delta are taken for that distance.
// Two levels as in k_bin_spheres.
template <int R, int WORDS, bool FROM_LIST>
__global__ __launch_bounds__(64) void k_bin_light(SceneDev sc, float Xx, float Xy, float Xz,
                                                  float reach, int32_t *out,
                                                  const int32_t *parent) {
  const int lane = threadIdx.x;
  const int cu = blockIdx.x, cw = blockIdx.y, face = blockIdx.z;
  const float u = ((lane & 1) ? cu + 1 : cu) * (2.0f / R) - 1.0f;
  const float w = ((lane & 2) ? cw + 1 : cw) * (2.0f / R) - 1.0f;
  const float m = (face & 1) ? -1.0f : 1.0f;
  float dx = face < 2 ? m : u, dy = face < 2 ? u : (face < 4 ? m : w), dz = face < 4 ? w : m;
  normalize3(dx, dy, dz);
  const bool lv = lane < 4;
  const float a = dot3(dx, dy, dz, dx, dy, dz);
  const Beam B = make_beam(lv, a, dx, dy, dz, Xx, Xy, Xz, reach, 9.5367431640625e-07f);
  int32_t *const rec = out + ((size_t)(face * R + cw) * R + cu) * WORDS;
  const int32_t *src = nullptr;
  int n_src = sc.s_size;
  if (FROM_LIST) {
    constexpr int F = LCELL_R / LCELL_SUPER_R;
    src = parent + ((size_t)(face * LCELL_SUPER_R + cw / F) * LCELL_SUPER_R + cu / F) * LCELL_SUPER_WORDS;
    n_src = src[0];
  }
  if (!B.narrow || n_src < 0) {
    if (lane == 0)
      rec[0] = -1;
    return;
  }
  int n = 0;
  for (int base = 0; base < n_src; base += 64) {
    const int k = base + lane;
    int idx = k < n_src ? k : n_src - 1;
    if (FROM_LIST)
      idx = src[1 + idx];
    const float4 g = sc.sph_geo[idx];
    const bool keep = k < n_src && beam_keeps(B, g);
    const unsigned long long mk = __ballot(keep);
    if (keep) {
      const int pos = n + __builtin_popcountll(mk & ((1ull << lane) - 1ull));
      if (pos < WORDS - LIST_GROUP)
        rec[1 + pos] = idx;
    }
    n += __builtin_popcountll(mk);
  }
  if (n <= WORDS - LIST_GROUP && lane < LIST_GROUP && n + lane < ((n + LIST_GROUP - 1) & ~(LIST_GROUP - 1)))
    rec[1 + n + lane] = 0; // the readers take LIST_GROUP entries at a time
  if (lane == 0)
    rec[0] = n <= WORDS - LIST_GROUP ? n : -1;
}

// ---------------------------------------------------------------------------
// k_verify_lists — do the sphere lists hold every sphere a ray could hit?
// ---------------------------------------------------------------------------
// One thread per pixel, every sphere of the scene for every ray (slow: a diagnostic).
// For the pixel's ray from the eye, and for the shadow rays of its hit towards the
// lights that have lists: a sphere for which the float discriminant test
// (maths.cpp:132-133) passes must be in the list of the ray's cell.  The lists are
// in rising index order, so a pointer into the list walks along with the loop over
// the spheres.  out[0] rays from the eye whose cell has a list, out[1] (ray, sphere)
// pairs of those with a passing test and the sphere not listed - must be 0,
// out[2] shadow rays whose cell has a list and whose origin is within reach,
// out[3] as out[1] for those.
__global__ __launch_bounds__(256) void k_verify_lists(SceneDev sc, CameraDev cam, int w, int h,
                                                      unsigned long long *out) {
  const int j = blockIdx.x * 16 + (threadIdx.x & 15), i = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (i >= h || j >= w)
    return;
  float px = (float)((2 * ((j + 0.5) / (double)(float)w) - 1) * cam.tan_half * (double)cam.aspect);
  float py = (float)((1 - 2 * ((i + 0.5) / (double)(float)h)) * cam.tan_half);
  float dx = px - cam.ox, dy = py - cam.oy, dz = cam.plane_z - cam.oz;
  normalize3(dx, dy, dz);
  const float a = dot3(dx, dy, dz, dx, dy, dz), a4 = 4.0f * a;
  unsigned long long served = 0, missing = 0, s_served = 0, s_missing = 0;
  if (sc.bins != nullptr) {
    const int32_t *rec = sc.bins + ((size_t)(i / BIN_PX) * sc.bins_x + j / BIN_PX) * BIN_WORDS;
    const int n = rec[0];
    if (n >= 0) {
      served = 1;
      int p = 0;
      for (int k = 0; k < sc.s_size; k++) {
        const float4 s = sc.sph_prim[k];
        const float b = 2.0f * dot3(dx, dy, dz, s.x, s.y, s.z);
        const bool pass = !(b * b < a4 * s.w);
        while (p < n && rec[1 + p] < k)
          p++;
        if (pass && !(p < n && rec[1 + p] == k))
          missing++;
      }
    }
  }
  // the pixel's hit, by the plain loops, and its shadow rays (renderer.cpp:82-119)
  const Hit hit = trace<true, false, RT_BEAM_NONE>(sc, true, cam.ox, cam.oy, cam.oz, dx, dy, dz);
  if (hit.kind != 0 && sc.lcells != nullptr) {
    const Surface sf = surface_at(sc, hit);
    for (int l = 0; l < sc.lcell_lights; l++) {
      const float4 light = sc.lights[l];
      float fColor[3], lx, ly, lz;
      (void)phong_light(sf, hit, dx, dy, dz, light, fColor, lx, ly, lz);
      const float dist = length3(light.x - hit.Px, light.y - hit.Py, light.z - hit.Pz);
      const float sa = dot3(lx, ly, lz, lx, ly, lz);
      if (!(sa > 0.f) || !(dist <= sc.lcell_reach[l]))
        continue;
      const int cell = light_cell<LCELL_R>(hit.Px - light.x, hit.Py - light.y, hit.Pz - light.z);
      const int32_t *rec = sc.lcells + ((size_t)l * LCELL_COUNT + cell) * LCELL_WORDS;
      const int n = rec[0];
      if (n < 0)
        continue;
      s_served++;
      int p = 0;
      for (int k = 0; k < sc.s_size; k++) {
        const float4 g = sc.sph_geo[k];
        const float Lx = hit.Px - g.x, Ly = hit.Py - g.y, Lz = hit.Pz - g.z; // maths.cpp:126
        const float b = 2.0f * dot3(lx, ly, lz, Lx, Ly, Lz);                 // :128
        const float c = dot3(Lx, Ly, Lz, Lx, Ly, Lz) - g.w;                  // :129
        const bool pass = !(b * b < (4.0f * sa) * c);
        while (p < n && rec[1 + p] < k)
          p++;
        if (pass && !(p < n && rec[1 + p] == k))
          s_missing++;
      }
    }
  }
  if (served)
    atomicAdd(out + 0, served);
  if (missing)
    atomicAdd(out + 1, missing);
  if (s_served)
    atomicAdd(out + 2, s_served);
  if (s_missing)
    atomicAdd(out + 3, s_missing);
}

// ---------------------------------------------------------------------------
// k_bounce — one bounce level: one queued ray per lane
// ---------------------------------------------------------------------------
// The grid is fixed; waves stride over the queue, whose length is read from
// device memory (the host never waits for it).
template <bool BEAM, bool VERIFY>
__global__ __launch_bounds__(256) void k_bounce(SceneDev sc, BounceDev bd,
                                                uint64_t *ray_count, ScreenStats st) {
  // this block works on segment (block % n_seg) together with the other blocks
  // of the same residue
  const uint32_t seg = my_segment();
  uint32_t n_in = bd.in.count[seg * CTL_WORDS];
  if (n_in > bd.in.seg_capacity)
    n_in = bd.in.seg_capacity;
  const uint32_t sib = gridDim.x / QUEUE_SEGMENTS; // blocks per segment (grid is a multiple)
  const uint32_t stride = sib * blockDim.x;
  const uint32_t first = (blockIdx.x / QUEUE_SEGMENTS) * blockDim.x + (threadIdx.x & ~63u);
  // Bounce rays of one wave share neither an origin nor a direction: every
  // lane walks the sphere hierarchy for its own ray, shadow rays included.
  constexpr int BM = BEAM ? RT_ACCEL_BVH : RT_BEAM_NONE;
  constexpr int BS = BEAM ? RT_ACCEL_BVH : RT_BEAM_NONE;
  __shared__ uint32_t walk_words[BEAM ? LDS_STACK_ROWS * 256 : 1];
  uint32_t *const lds_stack = BEAM ? walk_words + threadIdx.x : nullptr;
  uint32_t n_rays = 0;
  for (uint32_t base = first; base < n_in; base += stride) {
    const uint32_t r = base + (threadIdx.x & 63);
    const bool live = r < n_in;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (live) {
      const size_t g = (size_t)seg * bd.in.seg_capacity + r;
      a = bd.in.items[2 * g];
      b = bd.in.items[2 * g + 1];
    }
    const uint32_t tag = __float_as_uint(b.w);
    const uint32_t pixel = tag >> 6, depth = tag & 63u;
    Hit h = trace<false, false, BM, VERIFY>(sc, live, a.x, a.y, a.z, b.x, b.y, b.z, 0.f,
                                            0.f, 0.f, 0.f, st, lds_stack);
    if (live)
      n_rays++;
    const bool hit = live && h.kind != 0;
    uint32_t u0 = 0, u1 = 0, u2 = 0;
    ScreenStats st2 = st;
    if (st2.walk)
      st2.walk += 3; // shadow-ray walks are counted apart from nearest-hit walks
    if (st2.twalk)
      st2.twalk += 3;
    n_rays += shade_hits<false, true, BS, VERIFY>(sc, bd, hit, h, b.x, b.y, b.z, a.w, pixel,
                                                  depth, u0, u1, u2, st2, lds_stack);
  }
  add_ray_count(ray_count, n_rays);
}

// Pixel = saturating sum of the primary hit's light (already in fb) and the
// integer side sums of all bounce hits; the side sums are cleared for reuse.
// One thread per pixel of the bands this frame owns (bands of band_rows rows, band
// b belongs to share b % band_count): the other ranks' rows have no sums here.
// The first block also folds the frame's partial ray counters into the caller's word,
// as k_sum_rays does (parts may be null; RAY_PARTS threads per block).
__global__ void k_compose(uint8_t *fb, unsigned long long *extra, int w, int h, int band_rows,
                          int band_index, int band_count, int owned_rows, uint64_t *parts,
                          uint64_t *total) {
  if (parts != nullptr && blockIdx.x == 0 && blockIdx.y == 0) {
    unsigned long long *p =
        reinterpret_cast<unsigned long long *>(parts) + threadIdx.x * RAY_PART_STRIDE;
    unsigned long long v = *p;
    *p = 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
      v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0 && v)
      atomicAdd(reinterpret_cast<unsigned long long *>(total), v);
  }
  const int x = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (x >= w)
    return;
  for (int r_owned = (int)blockIdx.y; r_owned < owned_rows; r_owned += (int)gridDim.y) {
    const int row = ((r_owned / band_rows) * band_count + band_index) * band_rows + r_owned % band_rows;
    if (row >= h)
      return;
    const size_t p = (size_t)row * (size_t)w + (size_t)x;
    const unsigned long long e = extra[p];
    if (e == 0)
      continue;
    const uint32_t e0 = (uint32_t)(e & 0x1fffffu), e1 = (uint32_t)((e >> 21) & 0x1fffffu),
                   e2 = (uint32_t)((e >> 42) & 0x1fffffu);
    uchar4 c = reinterpret_cast<uchar4 *>(fb)[p];
    uint32_t v0 = c.x + e0, v1 = c.y + e1, v2 = c.z + e2;
    c.x = (unsigned char)(v0 > 255u ? 255u : v0);
    c.y = (unsigned char)(v1 > 255u ? 255u : v1);
    c.z = (unsigned char)(v2 > 255u ? 255u : v2);
    reinterpret_cast<uchar4 *>(fb)[p] = c;
    extra[p] = 0;
  }
}

} // namespace rt
#include "rt_trace.h"
namespace rt {

// ---------------------------------------------------------------------------
// k_check — bulk check_intersection (renderer.hpp:21-23)
// ---------------------------------------------------------------------------
template <bool ACCEL>
__global__ __launch_bounds__(256) void k_check(SceneDev sc, CheckArgs a) {
  long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  bool live = r < a.n;
  long rr = live ? r : 0;
  float ox = a.orig[3 * rr], oy = a.orig[3 * rr + 1], oz = a.orig[3 * rr + 2];
  float dx = a.dir[3 * rr], dy = a.dir[3 * rr + 1], dz = a.dir[3 * rr + 2];
  __shared__ uint32_t walk_words[ACCEL ? LDS_STACK_ROWS * 256 : 1];
  Hit h = trace<false, false, ACCEL ? RT_ACCEL_BVH : RT_BEAM_NONE>(
      sc, live, ox, oy, oz, dx, dy, dz, 0.f, 0.f, 0.f, 0.f, ScreenStats{nullptr, nullptr, nullptr},
      ACCEL ? walk_words + threadIdx.x : nullptr);
  if (live) {
    a.kind[r] = h.kind;
    a.index[r] = h.index;
    a.P[3 * r] = h.Px;
    a.P[3 * r + 1] = h.Py;
    a.P[3 * r + 2] = h.Pz;
  }
}

// ---------------------------------------------------------------------------
// k_specular — the specular term of renderer.cpp:106-110 on caller-supplied
// values, one per lane: strength = pow(lang, 20) as the shading computes it
// (pow20 above), specular = (unsigned char)(colour * strength).  Exists so that
// this one expression, whose libm call the device replaces by five multiplies,
// can be checked value by value.
// ---------------------------------------------------------------------------
__global__ void k_specular(long n, const float *lang, const uint8_t *color, float *strength,
                           uint8_t *specular) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const float s = pow20(lang[i]);
  strength[i] = s;
  specular[i] = (uint8_t)((uint32_t)((float)color[i] * s) & 0xffu); // renderer.cpp:108-110
}

// ---------------------------------------------------------------------------
// Launchers (host)
// ---------------------------------------------------------------------------
hipError_t launch_specular(long n, const float *lang, const uint8_t *color, float *strength,
                           uint8_t *specular, hipStream_t st) {
  if (n <= 0)
    return hipSuccess;
  hipLaunchKernelGGL(k_specular, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, lang,
                     color, strength, specular);
  return hipGetLastError();
}

hipError_t launch_prep_scene(const PrepArgs &a, hipStream_t st) {
  int n = a.s_pad > a.t_size ? a.s_pad : a.t_size;
  if (n <= 0)
    return hipSuccess;
  hipLaunchKernelGGL(k_prep_scene, dim3((n + 255) / 256), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_prep_eye(const EyeArgs &a, hipStream_t st) {
  int n = a.s_pad > a.t_size ? a.s_pad : a.t_size;
  if (n <= 0)
    return hipSuccess;
  hipLaunchKernelGGL(k_prep_eye, dim3((n + 255) / 256), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_bin_spheres(const SceneDev &sc, const CameraDev &cam, int w, int h, int32_t *bins,
                              int32_t *super, hipStream_t st) {
  if (w <= 0 || h <= 0 || sc.s_size <= 0 || !bins || !super)
    return hipErrorInvalidValue;
  const int bx = (w + BIN_PX - 1) / BIN_PX, by = (h + BIN_PX - 1) / BIN_PX;
  const int sx = (w + BIN_SUPER_PX - 1) / BIN_SUPER_PX, sy = (h + BIN_SUPER_PX - 1) / BIN_SUPER_PX;
  hipLaunchKernelGGL((k_bin_spheres<BIN_SUPER_PX, BIN_SUPER_WORDS, false>),
                     dim3(sx, sy, BIN_SUPER_SLICES), dim3(64), 0, st, sc, cam, w, h, super, sx,
                     (const int32_t *)nullptr, 0);
  hipLaunchKernelGGL((k_bin_spheres<BIN_PX, BIN_WORDS, true>), dim3(bx, by), dim3(64), 0, st, sc,
                     cam, w, h, bins, bx, (const int32_t *)super, sx);
  return hipGetLastError();
}

hipError_t launch_bin_light(const SceneDev &sc, float Xx, float Xy, float Xz, float reach,
                            int32_t *cells, int32_t *super, hipStream_t st) {
  if (sc.s_size <= 0 || !cells || !super || !(reach > 0.f))
    return hipErrorInvalidValue;
  hipLaunchKernelGGL((k_bin_light<LCELL_SUPER_R, LCELL_SUPER_WORDS, false>),
                     dim3(LCELL_SUPER_R, LCELL_SUPER_R, 6), dim3(64), 0, st, sc, Xx, Xy, Xz, reach,
                     super, (const int32_t *)nullptr);
  hipLaunchKernelGGL((k_bin_light<LCELL_R, LCELL_WORDS, true>), dim3(LCELL_R, LCELL_R, 6), dim3(64),
                     0, st, sc, Xx, Xy, Xz, reach, cells, (const int32_t *)super);
  return hipGetLastError();
}

hipError_t launch_verify_lists(const SceneDev &sc, const CameraDev &cam, int w, int h,
                               unsigned long long *out, hipStream_t st) {
  if (w <= 0 || h <= 0 || !out)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_verify_lists, dim3((w + 15) / 16, (h + 15) / 16), dim3(256), 0, st, sc, cam, w,
                     h, out);
  return hipGetLastError();
}

hipError_t launch_render(const SceneDev &sc, const CameraDev &cam,
                         const FrameDev &f, const BounceDev &bd,
                         int owned_tile_rows, int accel,
                         unsigned long long *verify_stats, hipStream_t st) {
  if (owned_tile_rows <= 0)
    return hipSuccess;
  dim3 grid((f.w + 31) / 32, owned_tile_rows);
  ScreenStats ss{nullptr, nullptr, nullptr};
  const bool mats = bd.max_depth > 1;
#define RT_LAUNCH(A, V, M)                                                               \
  do {                                                                                   \
    if (sc.tbvh != nullptr)                                                              \
      hipLaunchKernelGGL((k_render<A, V, M, true>), grid, dim3(256), 0, st, sc, cam, f, bd, ss);  \
    else                                                                                 \
      hipLaunchKernelGGL((k_render<A, V, M, false>), grid, dim3(256), 0, st, sc, cam, f, bd, ss); \
  } while (0)
  if (verify_stats) {
    ss.dropped_but_candidate = verify_stats;
    ss.kept = verify_stats + 1;
    if (mats)
      RT_LAUNCH(RT_BEAM_APEX, true, true);
    else
      RT_LAUNCH(RT_BEAM_APEX, true, false);
  } else if (accel == RT_BEAM_APEX) {
    if (mats)
      RT_LAUNCH(RT_BEAM_APEX, false, true);
    else
      RT_LAUNCH(RT_BEAM_APEX, false, false);
  } else if (accel == RT_ACCEL_BVH && sc.bvh) {
    if (mats)
      RT_LAUNCH(RT_ACCEL_BVH, false, true);
    else
      RT_LAUNCH(RT_ACCEL_BVH, false, false);
  } else {
    if (mats)
      RT_LAUNCH(RT_BEAM_NONE, false, true);
    else
      RT_LAUNCH(RT_BEAM_NONE, false, false);
  }
#undef RT_LAUNCH
  return hipGetLastError();
}

hipError_t launch_bounce(const SceneDev &sc, const BounceDev &bd, bool beam,
                         uint64_t *ray_count, unsigned long long *verify_stats,
                         int blocks, hipStream_t st) {
  ScreenStats ss{nullptr, nullptr, nullptr};
  if (verify_stats) {
    ss.dropped_but_candidate = verify_stats;
    ss.kept = verify_stats + 1;
    ss.walk = verify_stats + 2;
    ss.twalk = verify_stats + 8;
    hipLaunchKernelGGL((k_bounce<true, true>), dim3(blocks), dim3(256), 0, st, sc, bd,
                       ray_count, ss);
  } else if (beam) {
    hipLaunchKernelGGL((k_bounce<true, false>), dim3(blocks), dim3(256), 0, st, sc, bd,
                       ray_count, ss);
  } else {
    hipLaunchKernelGGL((k_bounce<false, false>), dim3(blocks), dim3(256), 0, st, sc, bd,
                       ray_count, ss);
  }
  return hipGetLastError();
}

hipError_t launch_walk(const SceneDev &sc, const WalkArgs &w, bool any_hit, bool triangles,
                       int blocks, hipStream_t st) {
  if (triangles && sc.tbvh == nullptr)
    return hipErrorInvalidValue;
  if (any_hit && triangles)
    hipLaunchKernelGGL((k_walk<true, true>), dim3(blocks), dim3(256), 0, st, sc, w);
  else if (any_hit)
    hipLaunchKernelGGL((k_walk<true, false>), dim3(blocks), dim3(256), 0, st, sc, w);
  else if (triangles)
    hipLaunchKernelGGL((k_walk<false, true>), dim3(blocks), dim3(256), 0, st, sc, w);
  else
    hipLaunchKernelGGL((k_walk<false, false>), dim3(blocks), dim3(256), 0, st, sc, w);
  return hipGetLastError();
}

hipError_t launch_trace(const SceneDev &sc, const TraceArgs &t, int bvh_depth, int blocks,
                        hipStream_t st) {
  if (sc.bvh == nullptr || sc.tbvh != nullptr || blocks <= 0 || bvh_depth < 1 ||
      bvh_depth > LDS_WALK_MAX_DEPTH)
    return hipErrorInvalidValue;
  // the fewest stack rows that serve the tree: the rest of the column parks leaves
  if (bvh_depth <= 6)
    hipLaunchKernelGGL(k_trace<trace_stack_rows(6)>, dim3(blocks), dim3(256), 0, st, sc, t);
  else if (bvh_depth == 7)
    hipLaunchKernelGGL(k_trace<trace_stack_rows(7)>, dim3(blocks), dim3(256), 0, st, sc, t);
  else
    hipLaunchKernelGGL(k_trace<trace_stack_rows(8)>, dim3(blocks), dim3(256), 0, st, sc, t);
  return hipGetLastError();
}

int trace_resident_blocks() {
  int dev = 0, per_cu = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace<LDS_STACK_ROWS>, 256, 0) !=
          hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      per_cu <= 0 || cus <= 0)
    return 1024;
  return per_cu * cus;
}

int walk_resident_blocks() {
  int dev = 0, per_cu = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_walk<false, true>, 256, 0) !=
          hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      per_cu <= 0 || cus <= 0)
    return 1024;
  const int blocks = per_cu * cus / QUEUE_SEGMENTS * QUEUE_SEGMENTS;
  return blocks > 0 ? blocks : QUEUE_SEGMENTS;
}

hipError_t launch_shade(const SceneDev &sc, const ShadeArgs &s, int blocks, hipStream_t st) {
  hipLaunchKernelGGL(k_shade, dim3(blocks), dim3(256), 0, st, sc, s);
  return hipGetLastError();
}

#ifdef RT_STAMPS
// Diagnostic build only: reads and clears the stamp counters of rt_walk.h.
extern "C" int rt_debug_stamps(unsigned long long *out16) {
  unsigned long long zero[16] = {};
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamps), sizeof zero) != hipSuccess)
    return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
// The same for the stamp counters of k_trace (rt_trace.h).
extern "C" int rt_debug_trace_stamps(unsigned long long *out24) {
  unsigned long long zero[24] = {};
  zero[19] = ~0ull; // a minimum
  if (hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_tstamps), sizeof zero) != hipSuccess)
    return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_tstamps), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
// Reads and clears the service-phase timeline of the sampled waves of k_trace
// (rt_trace.h, g_tlog): TLOG_WAVES x TLOG_RECS x 4 words.
extern "C" int rt_debug_trace_log(uint32_t *out, int *n_waves, int *n_recs) {
  *n_waves = TLOG_WAVES;
  *n_recs = TLOG_RECS;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tlog), sizeof(uint32_t) * TLOG_WAVES * TLOG_RECS * 4) !=
      hipSuccess)
    return -1;
  void *p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_tlog)) != hipSuccess)
    return -1;
  return hipMemset(p, 0, sizeof(uint32_t) * TLOG_WAVES * TLOG_RECS * 4) == hipSuccess ? 0 : -1;
}
#endif

hipError_t launch_sum_rays(uint64_t *parts, uint64_t *total, hipStream_t st) {
  hipLaunchKernelGGL(k_sum_rays, dim3(1), dim3(RAY_PARTS), 0, st, parts, total);
  return hipGetLastError();
}

static_assert(RAY_PARTS == 256, "k_compose sums the partial counters with one block of 256");
hipError_t launch_compose(uint8_t *fb, unsigned long long *extra, int w, int h, int band_rows,
                          int band_index, int band_count, uint64_t *parts, uint64_t *total,
                          hipStream_t st) {
  if (w <= 0 || h <= 0 || band_rows <= 0 || band_count <= 0 || band_index < 0 ||
      band_index >= band_count)
    return hipErrorInvalidValue;
  const int n_bands = (h + band_rows - 1) / band_rows;
  const int owned = n_bands > band_index ? (n_bands - band_index + band_count - 1) / band_count : 0;
  const long owned_rows = (long)owned * band_rows;
  if (owned_rows == 0)
    return parts ? launch_sum_rays(parts, total, st) : hipSuccess;
  const dim3 grid((unsigned)((w + 255) / 256), (unsigned)(owned_rows < 65535 ? owned_rows : 65535));
  hipLaunchKernelGGL(k_compose, grid, dim3(256), 0, st, fb, extra, w, h, band_rows, band_index,
                     band_count, (int)owned_rows, parts, total);
  return hipGetLastError();
}

hipError_t launch_check(const SceneDev &sc, const CheckArgs &a, hipStream_t st) {
  if (a.n <= 0)
    return hipSuccess;
  const dim3 grid((unsigned)((a.n + 255) / 256));
  if (sc.bvh || sc.tbvh)
    hipLaunchKernelGGL((k_check<true>), grid, dim3(256), 0, st, sc, a);
  else
    hipLaunchKernelGGL((k_check<false>), grid, dim3(256), 0, st, sc, a);
  return hipGetLastError();
}

// Sphere records in leaf order: the geometry for the walks, and for the kernel
// that names a hit sphere by its leaf slot (k_trace) material and colour in one
// record, so that shading such a hit costs two scattered reads instead of three.
__global__ void k_gather_leaves(const float4 *sph_geo, const uchar4 *col_sph,
                                const float4 *mat_sph, const int32_t *leaf_idx, float4 *leaf_geo,
                                float4 *leaf_mat, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const int32_t k = leaf_idx[i];
  leaf_geo[i] = sph_geo[k];
  const uchar4 c = col_sph[k];
  float4 m = mat_sph ? mat_sph[k] : make_float4(0.f, 0.f, 0.f, 0.f);
  m.w = __uint_as_float((uint32_t)c.x | ((uint32_t)c.y << 8) | ((uint32_t)c.z << 16));
  leaf_mat[i] = m;
}

__global__ void k_gather_tri_leaves(const TriRec *tri, const int32_t *leaf_idx, TriLeaf *leaf,
                                    int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const int32_t t = leaf_idx[i];
  const TriRec T = tri[t];
  TriLeaf L;
  L.n_d = T.n_d;
  L.a = make_float4(T.p1_no.x, T.p1_no.y, T.p1_no.z, T.p2.x);
  L.b = make_float4(T.p2.y, T.p2.z, T.p3.x, T.p3.y);
  L.c = make_float4(T.p3.z, __int_as_float(t), 0.f, 0.f);
  leaf[i] = L;
}

hipError_t launch_gather_tri_leaves(const TriRec *tri, const int32_t *leaf_idx, TriLeaf *leaf,
                                    int n, hipStream_t st) {
  if (n <= 0)
    return hipSuccess;
  hipLaunchKernelGGL(k_gather_tri_leaves, dim3((n + 255) / 256), dim3(256), 0, st, tri, leaf_idx,
                     leaf, n);
  return hipGetLastError();
}

hipError_t launch_gather_leaves(const float4 *sph_geo, const uchar4 *col_sph,
                                const float4 *mat_sph, const int32_t *leaf_idx, float4 *leaf_geo,
                                float4 *leaf_mat, int n, hipStream_t st) {
  if (n <= 0)
    return hipSuccess;
  hipLaunchKernelGGL(k_gather_leaves, dim3((n + 255) / 256), dim3(256), 0, st, sph_geo, col_sph,
                     mat_sph, leaf_idx, leaf_geo, leaf_mat, n);
  return hipGetLastError();
}

} // namespace rt
