// rt_kernels.hip — the HIP kernels of the trace/shade path (gfx950, wave64).
//
// Compiled with -ffp-contract=off: the reference image is defined by unfused
// IEEE arithmetic, and a single contracted multiply-add flips hit ids.
//
// Kernels
//   k_prep_scene      once per scene: flat arrays -> sphere/triangle records
//   k_prep_eye        once per eye position: eye-relative sphere/triangle terms
//   k_render          one lane per pixel, one 8x8 pixel tile per wave:
//                     ray generation, nearest hit, Phong + shadow rays, pixel
//                     write (renderer.cpp:28-140)
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
// k_render
// ---------------------------------------------------------------------------
// Work decomposition: a wave owns an 8x8 pixel tile (lane = 8*row + col inside
// the tile), a 256-thread block owns four horizontally adjacent tiles, so the
// block's pixel writes cover 128 contiguous bytes per row.  blockIdx.y counts
// the tile rows this call owns; bands of band_rows rows are dealt round-robin
// over band_count callers (row sharding, renderer.cpp:27-28).
template <bool BEAM, bool VERIFY>
__global__ __launch_bounds__(256) void k_render(SceneDev sc, CameraDev cam,
                                                FrameDev f, ScreenStats st) {
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

  Hit h = trace<true, false, BEAM, VERIFY>(sc, inside, cam.ox, cam.oy, cam.oz,
                                           dx, dy, dz, cam.ox, cam.oy, cam.oz,
                                           0.f, st);

  uint32_t fb0 = 0, fb1 = 0, fb2 = 0; // renderer.cpp:50-52
  uint32_t n_rays = inside ? 1u : 0u;
  const bool hit = inside && h.kind != 0;

  if (__any(hit)) {
    Surface s;
    if (hit)
      s = surface_at(sc, h);
    for (int l = 0; l < sc.l_size; l++) { // renderer.cpp:82
      float fColor[3] = {0, 0, 0};
      float lx = 0, ly = 0, lz = 0, angle = 0, dist = 0;
      const float4 light = sc.lights[l];
      if (hit) {
        angle = phong_light(s, h, dx, dy, dz, light, fColor, lx, ly, lz);
        dist = length3(light.x - h.Px, light.y - h.Py, light.z - h.Pz);
      }
      // renderer.cpp:119: shadow ray from exactly P, no distance limit.  All
      // shadow rays of the wave aim at the light: that is the beam's apex.
      Hit sh = trace<false, true, BEAM, VERIFY>(sc, hit, h.Px, h.Py, h.Pz, lx, ly,
                                                lz, light.x, light.y, light.z,
                                                dist, st);
      if (hit) {
        n_rays++;
        if (sh.kind == 0) {
          fb0 = accumulate_u8(fb0, fColor[0] * angle);
          fb1 = accumulate_u8(fb1, fColor[1] * angle);
          fb2 = accumulate_u8(fb2, fColor[2] * angle);
        }
      }
    }
  }

  if (inside) {
    size_t p = (size_t)f.w * i + j;
    uchar4 o;
    o.x = (unsigned char)fb0;
    o.y = (unsigned char)fb1;
    o.z = (unsigned char)fb2;
    o.w = 255; // renderer.cpp:53
    reinterpret_cast<uchar4 *>(f.fb)[p] = o;
    if (f.ids)
      f.ids[p] = h.kind == 0 ? -1 : (h.kind == 1 ? h.index : sc.t_size + h.index);
  }

  if (f.ray_count) {
    // wave sum, then one atomic per wave
    uint32_t v = n_rays;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
      v += __shfl_down(v, off, 64);
    if (lane == 0 && v)
      atomicAdd(reinterpret_cast<unsigned long long *>(f.ray_count),
                (unsigned long long)v);
  }
}

// ---------------------------------------------------------------------------
// k_check — bulk check_intersection (renderer.hpp:21-23)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_check(SceneDev sc, CheckArgs a) {
  long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  bool live = r < a.n;
  long rr = live ? r : 0;
  float ox = a.orig[3 * rr], oy = a.orig[3 * rr + 1], oz = a.orig[3 * rr + 2];
  float dx = a.dir[3 * rr], dy = a.dir[3 * rr + 1], dz = a.dir[3 * rr + 2];
  Hit h = trace<false, false, false>(sc, live, ox, oy, oz, dx, dy, dz);
  if (live) {
    a.kind[r] = h.kind;
    a.index[r] = h.index;
    a.P[3 * r] = h.Px;
    a.P[3 * r + 1] = h.Py;
    a.P[3 * r + 2] = h.Pz;
  }
}

// ---------------------------------------------------------------------------
// Launchers (host)
// ---------------------------------------------------------------------------
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

hipError_t launch_render(const SceneDev &sc, const CameraDev &cam,
                         const FrameDev &f, int owned_tile_rows, bool beam,
                         unsigned long long *verify_stats, hipStream_t st) {
  if (owned_tile_rows <= 0)
    return hipSuccess;
  dim3 grid((f.w + 31) / 32, owned_tile_rows);
  ScreenStats ss{nullptr, nullptr};
  if (verify_stats) {
    ss.dropped_but_candidate = verify_stats;
    ss.kept = verify_stats + 1;
    hipLaunchKernelGGL((k_render<true, true>), grid, dim3(256), 0, st, sc, cam, f, ss);
  } else if (beam) {
    hipLaunchKernelGGL((k_render<true, false>), grid, dim3(256), 0, st, sc, cam, f, ss);
  } else {
    hipLaunchKernelGGL((k_render<false, false>), grid, dim3(256), 0, st, sc, cam, f, ss);
  }
  return hipGetLastError();
}

hipError_t launch_check(const SceneDev &sc, const CheckArgs &a, hipStream_t st) {
  if (a.n <= 0)
    return hipSuccess;
  hipLaunchKernelGGL(k_check, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0,
                     st, sc, a);
  return hipGetLastError();
}

} // namespace rt
