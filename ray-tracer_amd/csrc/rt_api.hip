// rt_api.hip — the C ABI of include/rt_amd.h on top of the kernels.
//
// Host-side only: argument checks, HBM residency of the scene (the explicit
// form of the `map(to:)` clause at renderer.cpp:13-21), launch geometry and the
// copy-back of the framebuffer (`map(from:)`, renderer.cpp:22-25).  There is no
// CPU rendering path in this library: if HIP is unusable every entry point
// fails with an error code.
#include "../../include/rt_amd.h"
#include "rt_kernels.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <algorithm>
#include <utility>
#include <vector>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace {

thread_local char g_err[512] = "";
int g_screen = 1; // rt_set_screen

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                          \
  do {                                                                         \
    hipError_t e_ = (expr);                                                    \
    if (e_ != hipSuccess)                                                      \
      return fail(RT_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                  \
                  hipGetErrorString(e_), __FILE__, __LINE__);                  \
  } while (0)

template <typename T> struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  hipError_t alloc(size_t count) {
    release();
    n = count;
    if (count == 0)
      return hipSuccess;
    return hipMalloc((void **)&p, count * sizeof(T));
  }
  void release() {
    if (p)
      (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  ~DevBuf() { release(); }
};

} // namespace

// A frame with bounces can be rendered as several independent sub-frames
// (alternate row bands), each a chain of dependent launches on its own stream
// with its own queues (rt_set_lanes).  Measured on the 10k-sphere 4K depth-8
// frame the chains do run concurrently (kernel trace) but the walk kernels are
// bound by instruction issue, not by idle tails, so nothing is gained: 8.78 ms
// with 1 lane, 8.80 with 2, 9.92 with 3, 9.69 with 4; 1 is the default.
constexpr int kMaxLanes = 4;
int g_lanes = 1;
int g_side_stream = 1; // rt_set_side_stream

struct BounceLane {
  DevBuf<float4> queue[2];
  // control words.  A "row" is one counter per queue segment (QUEUE_SEGMENTS x
  // CTL_WORDS uint32).  Per level: rows {queue length, sphere-walk cursor,
  // triangle-walk cursor}, then per group of lights {shadow items, sphere-walk
  // cursor, triangle-walk cursor}; the very last word is the overflow flag
  DevBuf<uint32_t> counts;
  int ctl_rows = 0; // rows per level
  DevBuf<float4> hit_rec, tri_hit_rec, shadow_items[2]; // shadow items: by level parity
  DevBuf<int32_t> hit_idx, tri_hit_idx;
  size_t queue_capacity = 0;
  // The shadow walks of level L and the nearest-hit walks of level L+1 both
  // depend only on k_shade of level L, so the shadow walks run on a side stream.
  hipStream_t main = nullptr; // lanes >= 1; lane 0 runs on the caller's stream
  hipStream_t side = nullptr;
  hipEvent_t ev_fork[2] = {nullptr, nullptr}, ev_join[2] = {nullptr, nullptr}, ev_end = nullptr;

  ~BounceLane() {
    for (int k = 0; k < 2; k++) {
      if (ev_fork[k])
        (void)hipEventDestroy(ev_fork[k]);
      if (ev_join[k])
        (void)hipEventDestroy(ev_join[k]);
    }
    if (ev_end)
      (void)hipEventDestroy(ev_end);
    if (side)
      (void)hipStreamDestroy(side);
    if (main)
      (void)hipStreamDestroy(main);
  }
};

struct rt_scene {
  int device = 0;
  int t_size = 0, s_size = 0, s_pad = 0, l_size = 0;
  DevBuf<float4> sph_geo, sph_prim, lights, mat_tri, mat_sph;
  DevBuf<rt::TriRec> tri;
  DevBuf<uchar4> col_tri, col_sph;
  DevBuf<rt::BvhNode> bvh;
  DevBuf<float4> leaf_geo;
  DevBuf<int32_t> leaf_idx;
  bool has_bvh = false;
  DevBuf<rt::TriNode> tbvh;
  DevBuf<rt::TriLeaf> tleaf;
  DevBuf<int32_t> tloose;
  bool has_tbvh = false;
  bool has_mat_tri = false, has_mat_sph = false;
  bool eye_valid = false;
  float eye[3] = {0, 0, 0};
  // bounce machinery, allocated on the first max_depth > 1 render
  BounceLane lanes[kMaxLanes];
  // queue capacity per level: this many rays per owned pixel (+ 65536).  No level
  // can hold more than 256 per pixel: every queued ray weighs >= 2^-8 and the
  // weights of one pixel's rays at one level sum to at most 1.
  int rays_per_pixel = 4;
  DevBuf<unsigned long long> extra; // per pixel, 3 x 21 bits
  size_t extra_pixels = 0;
  hipEvent_t ev_begin = nullptr;

  ~rt_scene() {
    if (ev_begin)
      (void)hipEventDestroy(ev_begin);
  }

  rt::SceneDev dev() const {
    rt::SceneDev d;
    d.sph_geo = sph_geo.p;
    d.sph_prim = sph_prim.p;
    d.tri = tri.p;
    d.col_tri = col_tri.p;
    d.col_sph = col_sph.p;
    d.mat_tri = has_mat_tri ? mat_tri.p : nullptr;
    d.mat_sph = has_mat_sph ? mat_sph.p : nullptr;
    d.lights = lights.p;
    d.t_size = t_size;
    d.s_size = s_size;
    d.s_pad = s_pad;
    d.l_size = l_size;
    const bool use_bvh = has_bvh && g_screen != 0;
    d.bvh = use_bvh ? bvh.p : nullptr;
    d.leaf_geo = use_bvh ? leaf_geo.p : nullptr;
    d.leaf_idx = use_bvh ? leaf_idx.p : nullptr;
    const bool use_tbvh = has_tbvh && g_screen != 0;
    d.tbvh = use_tbvh ? tbvh.p : nullptr;
    d.tleaf = use_tbvh ? tleaf.p : nullptr;
    d.tloose = use_tbvh ? tloose.p : nullptr;
    d.n_tleaf = use_tbvh ? (int)tleaf.n : 0;
    d.n_tloose = use_tbvh ? (int)tloose.n : 0;
    return d;
  }
};

namespace {

int check_scene_desc(const scene_desc *s) {
  if (!s)
    return fail(RT_ERR_BAD_ARG, "scene is NULL");
  if (s->t_size < 0 || s->s_size < 0 || s->l_size < 0)
    return fail(RT_ERR_BAD_ARG, "negative primitive or light count");
  if (s->t_size > 0 && (!s->tris || !s->color_tri))
    return fail(RT_ERR_BAD_ARG, "t_size > 0 but tris/color_tri is NULL");
  if (s->s_size > 0 && (!s->spheres || !s->radius || !s->color_sphere))
    return fail(RT_ERR_BAD_ARG, "s_size > 0 but spheres/radius/color_sphere is NULL");
  if (s->l_size > 0 && !s->lights)
    return fail(RT_ERR_BAD_ARG, "l_size > 0 but lights is NULL");
  const rt_material *mats[2] = {s->mat_tri, s->mat_sphere};
  const int counts[2] = {s->t_size, s->s_size};
  for (int k = 0; k < 2; k++)
    for (int i = 0; mats[k] && i < counts[k]; i++) {
      const rt_material &m = mats[k][i];
      if (!(m.reflectivity >= 0.f) || !(m.transparency >= 0.f) ||
          !(m.reflectivity + m.transparency <= 1.0f))
        return fail(RT_ERR_BAD_ARG,
                    "material %d of %s: weights must be >= 0 and sum to <= 1", i,
                    k ? "spheres" : "triangles");
    }
  return RT_OK;
}

int ensure_device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(RT_ERR_NO_DEVICE, "no HIP device: %s",
                e == hipSuccess ? "count is 0" : hipGetErrorString(e));
  return RT_OK;
}

template <typename T>
int upload(DevBuf<T> &dst, const void *src, size_t count) {
  HIP_TRY(dst.alloc(count));
  if (count)
    HIP_TRY(hipMemcpy(dst.p, src, count * sizeof(T), hipMemcpyHostToDevice));
  return RT_OK;
}

// Fewer triangles than this are looped over by every ray (renderer.cpp:156-166
// as written); from here on the rays walk a hierarchy (rt_tribvh.h).
constexpr int kTriTreeMin = 16;

int scene_build(rt_scene *S, const scene_desc *s) {
  HIP_TRY(hipGetDevice(&S->device));
  S->t_size = s->t_size;
  S->s_size = s->s_size;
  S->l_size = s->l_size;
  S->s_pad = (s->s_size + rt::SPH_UNROLL - 1) / rt::SPH_UNROLL * rt::SPH_UNROLL;

  // staging copies of the flat arrays; freed when this function returns
  DevBuf<float> tris, spheres, radius;
  DevBuf<uint8_t> ctri, csph;
  int rc;
  if ((rc = upload(tris, s->tris, (size_t)9 * s->t_size)) ||
      (rc = upload(ctri, s->color_tri, (size_t)3 * s->t_size)) ||
      (rc = upload(spheres, s->spheres, (size_t)3 * s->s_size)) ||
      (rc = upload(radius, s->radius, (size_t)s->s_size)) ||
      (rc = upload(csph, s->color_sphere, (size_t)3 * s->s_size)))
    return rc;

  HIP_TRY(S->sph_geo.alloc(S->s_pad));
  HIP_TRY(S->sph_prim.alloc(S->s_pad));
  HIP_TRY(S->tri.alloc(S->t_size));
  HIP_TRY(S->col_tri.alloc(S->t_size));
  HIP_TRY(S->col_sph.alloc(S->s_size));

  // lights as float4
  {
    float4 *tmp = new (std::nothrow) float4[S->l_size > 0 ? S->l_size : 1];
    if (!tmp)
      return fail(RT_ERR_HIP, "out of host memory");
    for (int l = 0; l < S->l_size; l++)
      tmp[l] = make_float4(s->lights[3 * l], s->lights[3 * l + 1],
                           s->lights[3 * l + 2], 0.f);
    rc = upload(S->lights, tmp, (size_t)S->l_size);
    delete[] tmp;
    if (rc)
      return rc;
  }
  static_assert(sizeof(rt_material) == sizeof(float4), "material is one float4");
  S->has_mat_tri = s->mat_tri != nullptr && s->t_size > 0;
  S->has_mat_sph = s->mat_sphere != nullptr && s->s_size > 0;
  if (S->has_mat_tri && (rc = upload(S->mat_tri, s->mat_tri, (size_t)s->t_size)))
    return rc;
  if (S->has_mat_sph && (rc = upload(S->mat_sph, s->mat_sphere, (size_t)s->s_size)))
    return rc;

  rt::PrepArgs a;
  a.tris = tris.p;
  a.color_tri = ctri.p;
  a.spheres = spheres.p;
  a.radius = radius.p;
  a.color_sphere = csph.p;
  a.t_size = S->t_size;
  a.s_size = S->s_size;
  a.s_pad = S->s_pad;
  a.sph_geo = S->sph_geo.p;
  a.tri = S->tri.p;
  a.col_tri = S->col_tri.p;
  a.col_sph = S->col_sph.p;
  HIP_TRY(rt::launch_prep_scene(a, nullptr));

  // sphere hierarchy for rays that do not come in bundles
  {
    rt::BvhHost H;
    rt::bvh_build(s->spheres, s->radius, s->s_size, H);
    S->has_bvh = H.valid;
    if (H.valid) {
      if ((rc = upload(S->bvh, H.nodes.data(), H.nodes.size())) ||
          (rc = upload(S->leaf_idx, H.leaf_idx.data(), H.leaf_idx.size())))
        return rc;
      HIP_TRY(S->leaf_geo.alloc(H.leaf_idx.size()));
      HIP_TRY(rt::launch_gather_leaves(S->sph_geo.p, S->leaf_idx.p, S->leaf_geo.p,
                                       (int)H.leaf_idx.size(), nullptr));
    }
  }
  // triangle hierarchy, once there are enough triangles for it to pay
  if (s->t_size >= kTriTreeMin) {
    rt::TriBvhHost H;
    rt::tri_bvh_build(s->tris, s->t_size, H);
    S->has_tbvh = H.valid && !H.nodes.empty();
    if (S->has_tbvh) {
      DevBuf<int32_t> order; // leaf slot -> triangle; only the gather needs it
      if ((rc = upload(S->tbvh, H.nodes.data(), H.nodes.size())) ||
          (rc = upload(order, H.leaf_idx.data(), H.leaf_idx.size())) ||
          (rc = upload(S->tloose, H.loose.data(), H.loose.size())))
        return rc;
      HIP_TRY(S->tleaf.alloc(H.leaf_idx.size()));
      HIP_TRY(rt::launch_gather_tri_leaves(S->tri.p, order.p, S->tleaf.p,
                                           (int)H.leaf_idx.size(), nullptr));
      HIP_TRY(hipStreamSynchronize(nullptr)); // `order` dies here
    }
  }
  HIP_TRY(hipStreamSynchronize(nullptr)); // staging buffers die here
  S->eye_valid = false;
  return RT_OK;
}

// Eye-relative records are rebuilt only when the eye moves.
int scene_set_eye(rt_scene *S, const camera *cam, hipStream_t st) {
  if (S->eye_valid && memcmp(S->eye, cam->origin, sizeof S->eye) == 0)
    return RT_OK;
  rt::EyeArgs a;
  a.sph_geo = S->sph_geo.p;
  a.sph_prim = S->sph_prim.p;
  a.tri = S->tri.p;
  a.t_size = S->t_size;
  a.s_size = S->s_size;
  a.s_pad = S->s_pad;
  a.ox = cam->origin[0];
  a.oy = cam->origin[1];
  a.oz = cam->origin[2];
  HIP_TRY(rt::launch_prep_eye(a, st));
  memcpy(S->eye, cam->origin, sizeof S->eye);
  S->eye_valid = true;
  return RT_OK;
}

constexpr int kMaxDepth = 63;       // depth is stored in 6 bits of the ray tag
constexpr int kBounceBlocks = 2048; // 8 blocks of 256 lanes per CU

constexpr int kLightGroup = 4; // lights whose shadow rays share one pass
constexpr int kLevels = 66; // levels 0..65 (depth <= 63, +1 for the last spawn target)
constexpr size_t kRowWords = (size_t)rt::QUEUE_SEGMENTS * rt::CTL_WORDS;

int ensure_bounce_buffers(rt_scene *S, BounceLane *B, bool own_stream, size_t frame_pixels,
                          size_t owned_pixels) {
  size_t want = (size_t)S->rays_per_pixel * owned_pixels + 65536;
  if (want > 0x0ffffff0ull)
    want = 0x0ffffff0ull;
  want = (want + rt::QUEUE_SEGMENTS - 1) / rt::QUEUE_SEGMENTS * rt::QUEUE_SEGMENTS;
  if (B->queue_capacity < want) {
    for (int q = 0; q < 2; q++)
      HIP_TRY(B->queue[q].alloc(2 * want));
    HIP_TRY(B->hit_rec.alloc(want));
    HIP_TRY(B->hit_idx.alloc(want));
    if (S->has_tbvh) {
      HIP_TRY(B->tri_hit_rec.alloc(want));
      HIP_TRY(B->tri_hit_idx.alloc(want));
    }
    const int per_ray = S->l_size < kLightGroup ? (S->l_size > 0 ? S->l_size : 1) : kLightGroup;
    for (int k = 0; k < 2; k++)
      HIP_TRY(B->shadow_items[k].alloc(2 * want * per_ray));
    B->queue_capacity = want;
  }
  if (!B->side) {
    HIP_TRY(hipStreamCreateWithFlags(&B->side, hipStreamNonBlocking));
    for (int k = 0; k < 2; k++) {
      HIP_TRY(hipEventCreateWithFlags(&B->ev_fork[k], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&B->ev_join[k], hipEventDisableTiming));
    }
    HIP_TRY(hipEventCreateWithFlags(&B->ev_end, hipEventDisableTiming));
  }
  if (own_stream && !B->main)
    HIP_TRY(hipStreamCreateWithFlags(&B->main, hipStreamNonBlocking));
  if (!B->counts.p) {
    B->ctl_rows = 3 + 3 * ((S->l_size + kLightGroup - 1) / kLightGroup + 1);
    HIP_TRY(B->counts.alloc((size_t)kLevels * B->ctl_rows * kRowWords + 1));
  }
  if (S->extra_pixels < frame_pixels) {
    HIP_TRY(S->extra.alloc(frame_pixels));
    HIP_TRY(hipMemset(S->extra.p, 0, frame_pixels * sizeof(unsigned long long)));
    S->extra_pixels = frame_pixels;
  }
  return RT_OK;
}

uint32_t *ctl(BounceLane *B, int level, int row) {
  return B->counts.p + ((size_t)level * B->ctl_rows + row) * kRowWords;
}
uint32_t *overflow_flag(BounceLane *B) {
  return B->counts.p + (size_t)kLevels * B->ctl_rows * kRowWords;
}

rt::RayQueue queue_for_level(BounceLane *B, int level) {
  rt::RayQueue q;
  q.items = B->queue[level & 1].p;
  q.count = ctl(B, level, 0);
  q.seg_capacity = (uint32_t)(B->queue_capacity / rt::QUEUE_SEGMENTS);
  q.overflow = overflow_flag(B);
  return q;
}

rt::CameraDev camera_dev(const camera *cam, int w, int h) {
  rt::CameraDev c;
  c.tan_half = tan(cam->fov_deg * 0.5 * M_PI / 180); // renderer.cpp:35
  c.aspect = (float)w / (float)h;                     // renderer.cpp:11
  c.ox = cam->origin[0];
  c.oy = cam->origin[1];
  c.oz = cam->origin[2];
  c.plane_z = cam->plane_z;
  return c;
}

} // namespace

extern "C" {

void rt_camera_default(camera *cam) {
  if (!cam)
    return;
  cam->fov_deg = 90;     // main.cpp:88
  cam->origin[0] = 0.0f; // renderer.cpp:39
  cam->origin[1] = 0.0f;
  cam->origin[2] = 1.0f;
  cam->plane_z = -1.0f;  // renderer.cpp:38
}

const char *rt_last_error(void) { return g_err; }

const char *rt_version(void) { return "rt_amd 0.1 (gfx950)"; }

int rt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

int rt_scene_create(const scene_desc *scene, rt_scene **out) {
  if (!out)
    return fail(RT_ERR_BAD_ARG, "out is NULL");
  *out = nullptr;
  int rc;
  if ((rc = check_scene_desc(scene)) || (rc = ensure_device()))
    return rc;
  rt_scene *S = new (std::nothrow) rt_scene;
  if (!S)
    return fail(RT_ERR_HIP, "out of host memory");
  rc = scene_build(S, scene);
  if (rc) {
    delete S;
    return rc;
  }
  *out = S;
  return RT_OK;
}

void rt_scene_destroy(rt_scene *scene) { delete scene; }

// One sub-frame: lane `lane` of `n_lanes`.  Lane 0 runs on the caller's stream,
// the others on their own, started after everything the caller queued before.
static int render_lane(rt_scene *S, int lane, int n_lanes, const camera *cam, const rt_frame *fr,
                       int band_rows, int band_count, bool bounces, hipStream_t caller,
                       uint64_t *dev_ray_count, uint64_t *dev_verify) {
  BounceLane *B = &S->lanes[lane];
  const int sub_index = fr->band_index + lane * band_count;
  const int sub_count = band_count * n_lanes;
  rt::FrameDev f;
  f.fb = fr->dev_fb;
  f.ids = fr->dev_ids;
  f.ray_count = dev_ray_count;
  f.w = fr->w;
  f.h = fr->h;
  f.max_depth = fr->max_depth;
  f.band_rows = band_rows;
  f.band_index = sub_index;
  f.band_count = sub_count;
  int bands_total = (fr->h + band_rows - 1) / band_rows;
  int owned = bands_total > sub_index ? (bands_total - sub_index + sub_count - 1) / sub_count : 0;
  const int owned_tile_rows = owned * (band_rows / 8);
  rt::BounceDev bd;
  memset(&bd, 0, sizeof bd);
  bd.max_depth = 1;
  hipStream_t st = caller;
  int rc;
  if (bounces) {
    rc = ensure_bounce_buffers(S, B, lane > 0, (size_t)fr->w * fr->h,
                               (size_t)owned_tile_rows * 8 * fr->w);
    if (rc)
      return rc;
    if (lane > 0) {
      st = B->main;
      HIP_TRY(hipStreamWaitEvent(st, S->ev_begin, 0));
    }
    // only the levels this frame can reach, and the overflow flag
    HIP_TRY(hipMemsetAsync(B->counts.p, 0,
                           (size_t)(fr->max_depth + 2) * B->ctl_rows * kRowWords * sizeof(uint32_t),
                           st));
    HIP_TRY(hipMemsetAsync(overflow_flag(B), 0, sizeof(uint32_t), st));
    bd.max_depth = fr->max_depth;
    bd.extra = S->extra.p;
    bd.out = queue_for_level(B, 2);
  }
  const rt::SceneDev sd = S->dev();
  // 0 plain loop, 1 automatic (hierarchy when the scene has one, else the beam
  // screen), 2 beam screen, 3 hierarchy
  int accel = RT_BEAM_NONE;
  if (g_screen == 2 || ((g_screen == 1 || g_screen == 3) && !S->has_bvh))
    accel = RT_BEAM_APEX;
  else if (g_screen == 1 || g_screen == 3)
    accel = RT_ACCEL_BVH;
  HIP_TRY(rt::launch_render(sd, camera_dev(cam, fr->w, fr->h), f, bd, owned_tile_rows,
                            accel, (unsigned long long *)dev_verify, st));
  if (bounces) {
    // With a sphere hierarchy each level runs as walk / shade / shadow-walk
    // launches with dynamic ray fetch (rt_walk.h); otherwise, or in the
    // self-checking build, as one k_bounce launch per level.
    const bool stream_levels = (sd.bvh != nullptr || sd.tbvh != nullptr) && !dev_verify;
    const bool tri_walks = sd.tbvh != nullptr;
    bool joined[2] = {false, false};
    for (int level = 2; level <= fr->max_depth; level++) {
      bd.in = queue_for_level(B, level);
      bd.out = queue_for_level(B, level + 1);
      if (!stream_levels) {
        HIP_TRY(rt::launch_bounce(sd, bd, g_screen != 0, dev_ray_count,
                                  (unsigned long long *)dev_verify, kBounceBlocks, st));
        continue;
      }
      rt::HitList hl, tl;
      hl.rec = B->hit_rec.p;
      hl.idx = B->hit_idx.p;
      tl.rec = tri_walks ? B->tri_hit_rec.p : nullptr;
      tl.idx = tri_walks ? B->tri_hit_idx.p : nullptr;
      rt::WalkArgs wa;
      memset(&wa, 0, sizeof wa);
      wa.fetch_at = rt::WALK_FETCH_AT;
      wa.in = bd.in;
      wa.extra = S->extra.p;
      // Each ray is counted once, by the first walk it goes through.  The sphere
      // walk goes first: it is several times cheaper per ray (a 320 KB tree in L2, 41
      // instructions per child against ~90), and the |P| of its hit then bounds the
      // triangle walk (a triangle only matters at |P| <= the sphere's, renderer.cpp:181).
      wa.cursor = ctl(B, level, 1);
      wa.hits = hl;
      wa.ray_count = dev_ray_count;
      HIP_TRY(rt::launch_walk(sd, wa, false, false, kBounceBlocks, st));
      if (tri_walks) {
        wa.cursor = ctl(B, level, 2);
        wa.hits = tl;
        wa.prior = hl;
        wa.ray_count = nullptr;
        HIP_TRY(rt::launch_walk(sd, wa, false, true, kBounceBlocks, st));
      }
      rt::ShadeArgs sa;
      memset(&sa, 0, sizeof sa);
      sa.rays = bd.in;
      sa.hits = hl;
      sa.tri_hits = tl;
      sa.max_depth = fr->max_depth;
      sa.next = bd.out;
      sa.ray_count = dev_ray_count;
      const int n_groups = S->l_size > 0 ? (S->l_size + kLightGroup - 1) / kLightGroup : 1;
      const int per_ray = S->l_size < kLightGroup ? (S->l_size > 0 ? S->l_size : 1) : kLightGroup;
      // one group of lights: its shadow walks go to the side stream; more groups
      // share the item buffer and stay in order on the caller's stream
      const bool fork = n_groups == 1 && S->l_size > 0 && g_side_stream != 0;
      const int par = level & 1;
      for (int g = 0; g < n_groups; g++) {
        sa.light_begin = g * kLightGroup;
        sa.light_end = S->l_size < (g + 1) * kLightGroup ? S->l_size : (g + 1) * kLightGroup;
        sa.spawn = g == 0;
        sa.shadow.items = B->shadow_items[fork ? par : 0].p;
        sa.shadow.count = ctl(B, level, 3 + 3 * g);
        sa.shadow.seg_capacity = bd.in.seg_capacity * per_ray;
        sa.shadow.overflow = overflow_flag(B);
        if (fork && joined[par]) // the shadow walk two levels up has read this buffer
          HIP_TRY(hipStreamWaitEvent(st, B->ev_join[par], 0));
        HIP_TRY(rt::launch_shade(sd, sa, kBounceBlocks, st));
        if (S->l_size > 0) {
          hipStream_t ws_stream = st;
          if (fork) {
            HIP_TRY(hipEventRecord(B->ev_fork[par], st));
            HIP_TRY(hipStreamWaitEvent(B->side, B->ev_fork[par], 0));
            ws_stream = B->side;
          }
          rt::WalkArgs ws;
          memset(&ws, 0, sizeof ws);
          ws.fetch_at = rt::WALK_FETCH_AT;
          ws.in = sa.shadow;
          ws.extra = S->extra.p;
          // spheres first (the cheaper walk); only what they leave unoccluded is
          // worth a triangle walk
          ws.cursor = ctl(B, level, 4 + 3 * g);
          ws.ray_count = dev_ray_count;
          ws.last_stage = tri_walks ? 0 : 1;
          HIP_TRY(rt::launch_walk(sd, ws, true, false, kBounceBlocks, ws_stream));
          if (tri_walks) {
            ws.cursor = ctl(B, level, 5 + 3 * g);
            ws.ray_count = nullptr;
            ws.last_stage = 1;
            HIP_TRY(rt::launch_walk(sd, ws, true, true, kBounceBlocks, ws_stream));
          }
          if (fork) {
            HIP_TRY(hipEventRecord(B->ev_join[par], B->side));
            joined[par] = true;
          }
        }
      }
    }
    for (int k = 0; k < 2; k++)
      if (joined[k])
        HIP_TRY(hipStreamWaitEvent(st, B->ev_join[k], 0));
    if (lane > 0)
      HIP_TRY(hipEventRecord(B->ev_end, st));
  }
  return RT_OK;
}

static int render_device_impl(rt_scene *S, const camera *cam, const rt_frame *fr,
                              uint64_t *dev_ray_count, uint64_t *dev_verify) {
  if (!S || !cam || !fr)
    return fail(RT_ERR_BAD_ARG, "scene, camera or frame is NULL");
  if (!fr->dev_fb)
    return fail(RT_ERR_BAD_ARG, "frame.dev_fb is NULL");
  if (fr->w <= 0 || fr->h <= 0)
    return fail(RT_ERR_BAD_ARG, "frame size %dx%d", fr->w, fr->h);
  if (fr->max_depth < 1)
    return fail(RT_ERR_BAD_ARG, "max_depth %d < 1", fr->max_depth);
  int band_count = fr->band_count < 1 ? 1 : fr->band_count;
  int band_rows = band_count == 1 ? 8 : fr->band_rows;
  if (band_rows <= 0 || (band_rows & 7))
    return fail(RT_ERR_BAD_ARG, "band_rows %d must be a positive multiple of 8",
                band_rows);
  if (fr->band_index < 0 || fr->band_index >= band_count)
    return fail(RT_ERR_BAD_ARG, "band_index %d outside [0,%d)", fr->band_index,
                band_count);
  if (fr->max_depth > kMaxDepth)
    return fail(RT_ERR_BAD_ARG, "max_depth %d > %d", fr->max_depth, kMaxDepth);
  if (fr->max_depth > 1 && S->l_size > 8000)
    return fail(RT_ERR_UNSUPPORTED, "bounces with more than 8000 lights (21-bit light sums)");
  hipStream_t st = (hipStream_t)fr->stream;
  HIP_TRY(hipSetDevice(S->device));
  int rc = scene_set_eye(S, cam, st);
  if (rc)
    return rc;
  // Bounces exist only where a material asks for them.
  const bool bounces = fr->max_depth > 1 && (S->has_mat_tri || S->has_mat_sph);
  int n_lanes = bounces && !dev_verify ? g_lanes : 1;
  if (n_lanes < 1)
    n_lanes = 1;
  if (n_lanes > kMaxLanes)
    n_lanes = kMaxLanes;
  if (n_lanes > 1) {
    if (!S->ev_begin)
      HIP_TRY(hipEventCreateWithFlags(&S->ev_begin, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(S->ev_begin, st));
  }
  for (int k = 0; k < n_lanes; k++) {
    // lane k renders bands band_index + k * band_count, stepping n_lanes * band_count
    rc = render_lane(S, k, n_lanes, cam, fr, band_rows, band_count, bounces, st, dev_ray_count,
                     dev_verify);
    if (rc)
      return rc;
  }
  if (bounces) {
    for (int k = 1; k < n_lanes; k++)
      HIP_TRY(hipStreamWaitEvent(st, S->lanes[k].ev_end, 0));
    HIP_TRY(rt::launch_compose(fr->dev_fb, S->extra.p, (size_t)fr->w * fr->h, st));
  }
  return RT_OK;
}

int rt_render_device(rt_scene *S, const camera *cam, const rt_frame *fr,
                     uint64_t *dev_ray_count) {
  return render_device_impl(S, cam, fr, dev_ray_count, nullptr);
}

int rt_render_device_selfcheck(rt_scene *S, const camera *cam, const rt_frame *fr,
                               uint64_t *dev_ray_count, uint64_t *dev_stats) {
  if (!dev_stats)
    return fail(RT_ERR_BAD_ARG, "dev_stats is NULL");
  return render_device_impl(S, cam, fr, dev_ray_count, dev_stats);
}

int rt_scene_queue_overflowed(rt_scene *S, int *overflowed) {
  if (!S || !overflowed)
    return fail(RT_ERR_BAD_ARG, "scene or result is NULL");
  *overflowed = 0;
  HIP_TRY(hipSetDevice(S->device));
  for (int k = 0; k < kMaxLanes; k++) {
    if (!S->lanes[k].counts.p)
      continue;
    uint32_t flag = 0;
    HIP_TRY(hipMemcpy(&flag, overflow_flag(&S->lanes[k]), sizeof flag, hipMemcpyDeviceToHost));
    *overflowed |= flag != 0;
  }
  return RT_OK;
}

int rt_scene_set_rays_per_pixel(rt_scene *S, int n) {
  if (!S)
    return fail(RT_ERR_BAD_ARG, "scene is NULL");
  if (n < 1 || n > 256)
    return fail(RT_ERR_BAD_ARG, "rays per pixel %d outside 1..256", n);
  S->rays_per_pixel = n;
  return RT_OK;
}

// Host-only self-check of the two hierarchy builders (no HIP call): walks each
// tree and counts violated invariants in double precision.
int rt_host_check_hierarchies(const scene_desc *s, int64_t *out) {
  int rc = check_scene_desc(s);
  if (rc)
    return rc;
  if (!out)
    return fail(RT_ERR_BAD_ARG, "out is NULL");
  for (int k = 0; k < 8; k++)
    out[k] = 0;
  // ---- spheres: boxes ----
  {
    rt::BvhHost H;
    rt::bvh_build(s->spheres, s->radius, s->s_size, H);
    out[0] = H.valid ? (int64_t)H.nodes.size() : -1;
    if (H.valid) {
      std::vector<int> seen(s->s_size, 0);
      struct Item {
        int node;
        std::vector<std::pair<float4, float4>> boxes; // ancestors' child entries
      };
      std::vector<Item> todo;
      todo.push_back(Item{0, {}});
      while (!todo.empty()) {
        Item it = todo.back();
        todo.pop_back();
        const rt::BvhNode &N = H.nodes[it.node];
        for (int c = 0; c < rt::BVH_WIDTH; c++) {
          int32_t link;
          memcpy(&link, &N.b[c].w, 4);
          if (link == rt::BVH_EMPTY)
            continue;
          auto boxes = it.boxes;
          boxes.push_back({N.a[c], N.b[c]});
          if (link >= 0) {
            todo.push_back(Item{link, boxes});
            continue;
          }
          const int idx = H.leaf_idx[(~link) >> 3];
          seen[idx]++;
          const double r = std::fabs((double)s->radius[idx]);
          for (auto &bx : boxes) {
            const double lo[3] = {bx.first.x, bx.first.y, bx.first.z};
            const double hi[3] = {bx.first.w, bx.second.x, bx.second.y};
            double d2 = 0;
            for (int d = 0; d < 3; d++) {
              const double cc = s->spheres[3 * idx + d];
              if (cc - r < lo[d] || cc + r > hi[d])
                out[1]++; // sphere sticks out of an ancestor box
              const double g = lo[d] > 0 ? lo[d] : (hi[d] < 0 ? -hi[d] : 0.0);
              d2 += g * g;
            }
            if ((double)bx.second.z > std::sqrt(d2))
              out[2]++; // distance bound above the true distance
          }
        }
      }
      for (int i = 0; i < s->s_size; i++)
        if (seen[i] != 1)
          out[3]++; // sphere missing or duplicated
    }
  }
  // ---- triangles: spheres + normal cones ----
  {
    rt::TriBvhHost H;
    rt::tri_bvh_build(s->tris, s->t_size, H);
    out[4] = H.valid ? (int64_t)H.nodes.size() : -1;
    out[5] = (int64_t)H.loose.size();
    if (H.valid && !H.nodes.empty()) {
      std::vector<int> seen(s->t_size, 0);
      for (int i : H.loose)
        seen[i]++;
      struct Item {
        int node;
        std::vector<std::pair<float4, uint2>> bounds;
      };
      std::vector<Item> todo;
      todo.push_back(Item{0, {}});
      while (!todo.empty()) {
        Item it = todo.back();
        todo.pop_back();
        const rt::TriNode &N = H.nodes[it.node];
        for (int c = 0; c < rt::BVH_WIDTH; c++) {
          const int32_t link = N.link[c];
          if (link == rt::BVH_EMPTY)
            continue;
          auto bounds = it.bounds;
          bounds.push_back({N.s[c], N.ns[c]});
          if (link >= 0) {
            todo.push_back(Item{link, bounds});
            continue;
          }
          const int idx = H.leaf_idx[(~link) >> 3];
          seen[idx]++;
          const float *p = s->tris + 9 * (size_t)idx;
          // the triangle's unit normal, in double
          const double ax = (double)p[3] - p[0], ay = (double)p[4] - p[1], az = (double)p[5] - p[2];
          const double bx = (double)p[6] - p[0], by = (double)p[7] - p[1], bz = (double)p[8] - p[2];
          double n[3] = {ay * bz - az * by, az * bx - ax * bz, ax * by - ay * bx};
          const double nl = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
          for (auto &bd : bounds) {
            for (int k = 0; k < 3; k++) {
              const double dx = p[3 * k] - (double)bd.first.x, dy = p[3 * k + 1] - (double)bd.first.y,
                           dz = p[3 * k + 2] - (double)bd.first.z;
              if (std::sqrt(dx * dx + dy * dy + dz * dz) > (double)bd.first.w)
                out[6]++; // corner outside an ancestor sphere
            }
            // unpack the node normal and spread as the device does
            const double u[3] = {(double)(int16_t)(bd.second.x & 0xffffu) / 32767.0,
                                 (double)(int16_t)(bd.second.x >> 16) / 32767.0,
                                 (double)(int16_t)(bd.second.y & 0xffffu) / 32767.0};
            const uint32_t hb = bd.second.y >> 16;
            const int he = (hb >> 10) & 31, hm = hb & 1023;
            const double spread = he == 0 ? hm * std::ldexp(1.0, -24)
                                          : (he == 31 ? INFINITY : (1.0 + hm / 1024.0) * std::ldexp(1.0, he - 15));
            if (nl > 0) {
              const double ul = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
              double dm = 0, dp = 0;
              for (int k = 0; k < 3; k++) {
                dm += (n[k] / nl - u[k] / ul) * (n[k] / nl - u[k] / ul);
                dp += (n[k] / nl + u[k] / ul) * (n[k] / nl + u[k] / ul);
              }
              if (std::sqrt(std::min(dm, dp)) > spread + 1e-5)
                out[7]++; // normal outside an ancestor cone (1e-5: float normal vs this double one)
            }
          }
        }
      }
      for (int i = 0; i < s->t_size; i++)
        if (seen[i] > 1)
          out[3]++; // triangle in two places
    }
  }
  return RT_OK;
}

void rt_set_side_stream(int on) { g_side_stream = on != 0; }
int rt_get_side_stream(void) { return g_side_stream; }
void rt_set_lanes(int n) { g_lanes = n < 1 ? 1 : (n > kMaxLanes ? kMaxLanes : n); }
int rt_get_lanes(void) { return g_lanes; }
void rt_set_screen(int enabled) { g_screen = enabled; }
int rt_get_screen(void) { return g_screen; }

int gpu_render_ids(const scene_desc *scene, const camera *cam, uint8_t *rgb,
                   int32_t *hit_ids, uint64_t *n_rays, int w, int h,
                   int max_depth) {
  if (!cam || !rgb)
    return fail(RT_ERR_BAD_ARG, "camera or output buffer is NULL");
  if (w <= 0 || h <= 0)
    return fail(RT_ERR_BAD_ARG, "frame size %dx%d", w, h);
  rt_scene *S = nullptr;
  int rc = rt_scene_create(scene, &S);
  if (rc)
    return rc;
  size_t npx = (size_t)w * h;
  DevBuf<uint8_t> fb;
  DevBuf<int32_t> ids;
  DevBuf<uint64_t> cnt;
  auto body = [&]() -> int {
    HIP_TRY(fb.alloc(npx * 4));
    if (hit_ids)
      HIP_TRY(ids.alloc(npx));
    HIP_TRY(cnt.alloc(1));
    HIP_TRY(hipMemset(cnt.p, 0, sizeof(uint64_t)));
    rt_frame fr;
    memset(&fr, 0, sizeof fr);
    fr.dev_fb = fb.p;
    fr.dev_ids = ids.p;
    fr.w = w;
    fr.h = h;
    fr.max_depth = max_depth;
    fr.band_rows = 8;
    fr.band_index = 0;
    fr.band_count = 1;
    fr.stream = nullptr;
    // render; if a bounce queue overflowed, again with eight times the capacity
    // (at 256 rays per pixel it cannot overflow)
    for (;;) {
      HIP_TRY(hipMemset(cnt.p, 0, sizeof(uint64_t)));
      int r = rt_render_device(S, cam, &fr, cnt.p);
      if (r)
        return r;
      HIP_TRY(hipStreamSynchronize(nullptr));
      int over = 0;
      if ((r = rt_scene_queue_overflowed(S, &over)))
        return r;
      if (!over)
        break;
      if (S->rays_per_pixel >= 256)
        return fail(RT_ERR_CAPACITY, "bounce-ray queue overflowed; image is incomplete");
      S->rays_per_pixel = S->rays_per_pixel * 8 > 256 ? 256 : S->rays_per_pixel * 8;
    }
    HIP_TRY(hipMemcpy(rgb, fb.p, npx * 4, hipMemcpyDeviceToHost));
    if (hit_ids)
      HIP_TRY(hipMemcpy(hit_ids, ids.p, npx * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (n_rays)
      HIP_TRY(hipMemcpy(n_rays, cnt.p, sizeof(uint64_t), hipMemcpyDeviceToHost));
    return RT_OK;
  };
  rc = body();
  rt_scene_destroy(S);
  return rc;
}

int gpu_render(const scene_desc *scene, const camera *cam, uint8_t *rgb, int w,
               int h, int max_depth) {
  return gpu_render_ids(scene, cam, rgb, nullptr, nullptr, w, h, max_depth);
}

int gpu_check_intersection(rt_scene *S, int64_t n, const float *orig,
                           const float *dir, int32_t *kind, int32_t *index,
                           float *P) {
  if (!S)
    return fail(RT_ERR_BAD_ARG, "scene is NULL");
  if (n < 0)
    return fail(RT_ERR_BAD_ARG, "n < 0");
  if (n == 0)
    return RT_OK;
  if (!orig || !dir || !kind || !index || !P)
    return fail(RT_ERR_BAD_ARG, "NULL ray or result array");
  HIP_TRY(hipSetDevice(S->device));
  DevBuf<float> d_o, d_d, d_P;
  DevBuf<int32_t> d_k, d_i;
  int rc;
  if ((rc = upload(d_o, orig, (size_t)3 * n)) || (rc = upload(d_d, dir, (size_t)3 * n)))
    return rc;
  HIP_TRY(d_P.alloc((size_t)3 * n));
  HIP_TRY(d_k.alloc((size_t)n));
  HIP_TRY(d_i.alloc((size_t)n));
  rt::CheckArgs a;
  a.n = (long)n;
  a.orig = d_o.p;
  a.dir = d_d.p;
  a.kind = d_k.p;
  a.index = d_i.p;
  a.P = d_P.p;
  HIP_TRY(rt::launch_check(S->dev(), a, nullptr));
  HIP_TRY(hipMemcpy(kind, d_k.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(index, d_i.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(P, d_P.p, (size_t)3 * n * sizeof(float), hipMemcpyDeviceToHost));
  return RT_OK;
}

} // extern "C"
