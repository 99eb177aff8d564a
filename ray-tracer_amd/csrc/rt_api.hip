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
#include <cstdlib>
#include <cstring>
#include <new>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <utility>
#include <vector>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

// The bounce machinery of rt_render_device_noalloc: part of the code object, one
// copy per GPU.  Host code never uses this name as an address; it asks the
// runtime for the array's address on each device once (noalloc_scratch below).
static __device__ __align__(256) unsigned char
    scratch[RT_NOALLOC_SCRATCH_BYTES];

namespace {

thread_local char g_err[512] = "";
std::atomic<int> g_default_screen{1}; // rt_set_screen: mode of scenes created later

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

constexpr int kMaxDepth = 63;       // depth is stored in 6 bits of the ray tag
constexpr int kBounceBlocks = 2048; // 8 blocks of 256 lanes per CU
constexpr int kLightGroup = 4;      // lights whose shadow rays share one pass
constexpr size_t kRowWords = (size_t)rt::QUEUE_SEGMENTS * rt::CTL_WORDS;
// A bounce ray carries its pixel in the upper 26 bits of a 32-bit tag
// (pixel << 6 | depth, rt_kernels.hip queue_push).
constexpr size_t kMaxBouncePixels = (size_t)1 << 26;
constexpr size_t kSmallSharePixels = 1500000; // see bounce_needs
constexpr size_t kCheckPinnedRays = 256;      // see gpu_check_intersection

// Measurement knobs of the one-launch bounce kernel, read once from the environment
// (tools/ only; the defaults are what ships): RT_AMD_TRACE_SMALL_GRID = percent of the
// resident grid that a small share of a frame runs on (bounce_needs).
static int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}
static const int g_trace_small_grid_pct = env_int("RT_AMD_TRACE_SMALL_GRID", 50);
// RT_AMD_TRACE_SMALL_PIXELS = owned pixels below which a share counts as small
static const int g_trace_small_pixels = env_int("RT_AMD_TRACE_SMALL_PIXELS", (int)kSmallSharePixels);
// RT_AMD_PRIMARY_BINS = 0: the rays from the eye walk the sphere hierarchy like all others
// (measurement only; what ships lists the spheres by screen cell, k_bin_spheres).
static const int g_primary_bins = env_int("RT_AMD_PRIMARY_BINS", 1);
// Scenes with fewer spheres get no lists by screen cell or by light direction: their
// tree is a handful of nodes.  RT_AMD_LIST_MIN_SPHERES lowers that for tests.
static const int kBinMinSpheres = env_int("RT_AMD_LIST_MIN_SPHERES", 64);
// RT_AMD_LIGHT_CELLS = 0: the primary hits' shadow rays walk the hierarchy too.
static const int g_light_cells = env_int("RT_AMD_LIGHT_CELLS", 1);
// RT_AMD_TRACE_FETCH_AT = idle lanes at which a wave of that kernel stops to refill
static const int g_trace_fetch_at = env_int("RT_AMD_TRACE_FETCH_AT", rt::TRACE_FETCH_AT);
// the same for the launch-per-level walks: nearest-hit and shadow walks
static const int g_walk_fetch_at = env_int("RT_AMD_WALK_FETCH_AT", rt::WALK_FETCH_AT);
static const int g_walk_fetch_at_shadow = env_int("RT_AMD_WALK_FETCH_AT_SHADOW", rt::WALK_FETCH_AT);

// What the bounce levels of one frame need besides the scene: ray queues, hit
// lists, shadow items, control words and the per-pixel integer sums.  All of it
// is carved out of ONE region of device memory, so that the same launch code
// runs on a region the scene allocated and on a region a caller provides.
struct BounceNeeds {
  size_t capacity = 0;     // rays per level (a multiple of QUEUE_SEGMENTS)
  int per_ray = 1;         // shadow items per ray and pass
  bool tri = false;        // hit lists of the triangle walk as well
  int levels = 0;          // control rows for levels 0 .. levels-1
  int ctl_rows = 0;        // control rows per level
  size_t frame_pixels = 0; // per-pixel sums for the whole frame
  // One launch for all levels (rt_trace.h) instead of three per level: only the
  // first-bounce queue is frame-wide, the rest is per wave.
  bool trace = false;
  size_t trace_waves = 0;  // waves of the k_trace grid
  uint32_t stack_cap = 0;  // work items per wave

  bool covers(const BounceNeeds &o) const {
    return trace == o.trace && trace_waves >= o.trace_waves && stack_cap >= o.stack_cap &&
           capacity >= o.capacity && per_ray >= o.per_ray && (tri || !o.tri) &&
           levels >= o.levels && ctl_rows >= o.ctl_rows && frame_pixels >= o.frame_pixels;
  }
  void grow_to(const BounceNeeds &o) {
    if (trace != o.trace) { // the other kind of machinery: only the pixel sums carry over
      const size_t px = std::max(frame_pixels, o.frame_pixels);
      *this = o;
      frame_pixels = px;
      return;
    }
    trace_waves = std::max(trace_waves, o.trace_waves);
    stack_cap = std::max(stack_cap, o.stack_cap);
    capacity = std::max(capacity, o.capacity);
    per_ray = std::max(per_ray, o.per_ray);
    tri = tri || o.tri;
    levels = std::max(levels, o.levels);
    ctl_rows = std::max(ctl_rows, o.ctl_rows);
    frame_pixels = std::max(frame_pixels, o.frame_pixels);
  }
};

struct BounceArena {
  BounceNeeds have;
  float4 *queue[2] = {nullptr, nullptr};
  // control words.  A "row" is one counter per queue segment (QUEUE_SEGMENTS x
  // CTL_WORDS uint32).  Per level: rows {queue length, sphere-walk cursor,
  // triangle-walk cursor}, then per group of lights {shadow items, sphere-walk
  // cursor, triangle-walk cursor}; the very last word is the overflow flag
  uint32_t *counts = nullptr;
  float4 *hit_rec = nullptr, *tri_hit_rec = nullptr;
  float4 *shadow_items[2] = {nullptr, nullptr}; // by level parity
  int32_t *hit_idx = nullptr, *tri_hit_idx = nullptr;
  unsigned long long *extra = nullptr; // per pixel, 3 x 21 bits
  float4 *wave_stack = nullptr, *wave_hits = nullptr, *wave_save = nullptr; // k_trace
  size_t bytes = 0;

  // Places the regions for `n` behind `base` (256-byte aligned pieces) and
  // returns the bytes used; base == nullptr only measures.
  size_t carve(unsigned char *base, const BounceNeeds &n) {
    size_t off = 0;
    auto take = [&](size_t bytes_wanted) {
      unsigned char *p = base ? base + off : nullptr;
      off += (bytes_wanted + 255) & ~(size_t)255;
      return p;
    };
    const size_t cap = n.capacity;
    const size_t lvl = n.trace ? 0 : cap; // what only the launch-per-level path needs
    queue[0] = (float4 *)take(2 * cap * sizeof(float4));
    queue[1] = (float4 *)take(2 * lvl * sizeof(float4));
    hit_rec = (float4 *)take(lvl * sizeof(float4));
    hit_idx = (int32_t *)take(lvl * sizeof(int32_t));
    tri_hit_rec = (float4 *)take(n.tri ? lvl * sizeof(float4) : 0);
    tri_hit_idx = (int32_t *)take(n.tri ? lvl * sizeof(int32_t) : 0);
    for (int k = 0; k < 2; k++)
      shadow_items[k] = (float4 *)take(2 * lvl * (size_t)n.per_ray * sizeof(float4));
    wave_stack = (float4 *)take(n.trace_waves * n.stack_cap * 2 * sizeof(float4));
    wave_hits = (float4 *)take(n.trace_waves * rt::TRACE_HIT_CAP * 3 * sizeof(float4));
    wave_save = (float4 *)take(n.trace_waves * 64 * rt::TRACE_SAVE_WORDS * sizeof(float));
    counts = (uint32_t *)take(((size_t)n.levels * n.ctl_rows * kRowWords + 1) * sizeof(uint32_t));
    extra = (unsigned long long *)take(n.frame_pixels * sizeof(unsigned long long));
    if (base) {
      have = n;
      bytes = off;
    }
    return off;
  }
  uint32_t *ctl(int level, int row) const {
    return counts + ((size_t)level * have.ctl_rows + row) * kRowWords;
  }
  uint32_t *overflow_flag() const {
    return counts + (size_t)have.levels * have.ctl_rows * kRowWords;
  }
  rt::RayQueue queue_for_level(int level) const {
    rt::RayQueue q;
    q.items = queue[level & 1];
    q.count = ctl(level, 0);
    q.seg_capacity = (uint32_t)(have.capacity / rt::QUEUE_SEGMENTS);
    q.overflow = overflow_flag();
    return q;
  }
};

} // namespace

struct rt_scene {
  int device = 0;
  int t_size = 0, s_size = 0, s_pad = 0, l_size = 0;
  DevBuf<float4> sph_geo, sph_prim, lights, mat_tri, mat_sph;
  DevBuf<rt::TriRec> tri;
  DevBuf<uchar4> col_tri, col_sph;
  DevBuf<rt::BvhNode> bvh;
  DevBuf<float4> leaf_geo, leaf_mat;
  DevBuf<int32_t> leaf_idx;
  DevBuf<uint64_t> ray_parts; // partial ray counters of the frame in flight (zero between frames)
  bool has_bvh = false;
  int bvh_depth = 0;     // levels of inner nodes
  size_t bvh_nodes = 0;
  DevBuf<rt::TriNode> tbvh;
  DevBuf<rt::TriLeaf> tleaf;
  DevBuf<int32_t> tloose;
  bool has_tbvh = false;
  size_t tbvh_nodes = 0;
  bool has_mat_tri = false, has_mat_sph = false;
  int screen = 1;      // rt_scene_set_screen
  int side_stream = 1; // rt_scene_set_side_stream
  int one_launch = 1;  // rt_scene_set_one_launch
  int trace_blocks = 1024; // resident blocks of k_trace on this device
  int walk_blocks = 1024;  // of k_walk
  // Eye-relative records (k_prep_eye) are rebuilt only when the eye moves.  The
  // rebuild runs on the stream of the frame that asks for it; ev_eye lets frames
  // on other streams wait for it, ev_frame marks the end of the last frame's
  // launches so that a rebuild does not overtake a frame still reading the old
  // records.
  bool eye_valid = false;
  float eye[3] = {0, 0, 0};
  // Spheres by screen cell for the rays from the eye (k_bin_spheres): valid for one
  // camera and frame size, rebuilt with the eye-relative records' protocol when
  // either changes.  bins_use: this frame's launches read them.
  DevBuf<int32_t> bins, bins_super; // bins_super: the first of the two levels they are made in
  // Spheres by direction from the first lights (k_bin_light), made with the scene
  DevBuf<int32_t> lcells;
  int lcell_lights = 0;
  float lcell_reach[rt::LCELL_MAX_LIGHTS] = {0, 0, 0, 0};
  bool bins_valid = false, bins_use = false;
  int bins_w = 0, bins_h = 0;
  camera bins_cam = {};
  hipEvent_t ev_eye = nullptr, ev_frame = nullptr;
  bool frame_recorded = false;
  // bounce machinery, allocated on the first frame with bounces
  DevBuf<unsigned char> arena_mem;
  BounceArena arena;
  size_t arena_limit = 0; // rt_scene_set_arena_limit: 0 = none
  // queue capacity per level: this many rays per owned pixel (+ 65536).  No level
  // can hold more than 256 per pixel: every queued ray weighs >= 2^-8 and the
  // weights of one pixel's rays at one level sum to at most 1.
  int rays_per_pixel = 4;
  hipStream_t side = nullptr;
  hipEvent_t ev_fork[2] = {nullptr, nullptr}, ev_join[2] = {nullptr, nullptr};
  // rt_scene_time_kernels: events around the stages of the last frame
  bool timing = false;
  hipEvent_t ev_t[4] = {nullptr, nullptr, nullptr, nullptr};
  int timed_stages = 0;
  // staging for gpu_check_intersection (grow-only); for a few rays a block of
  // pinned host memory that the kernel reads and writes directly
  DevBuf<float> chk_in, chk_P;
  DevBuf<int32_t> chk_out;
  float *chk_pin = nullptr, *chk_pin_dev = nullptr;

  ~rt_scene() {
    for (hipEvent_t e : {ev_eye, ev_frame, ev_fork[0], ev_fork[1], ev_join[0], ev_join[1], ev_t[0],
                         ev_t[1], ev_t[2], ev_t[3]})
      if (e)
        (void)hipEventDestroy(e);
    if (side)
      (void)hipStreamDestroy(side);
    if (chk_pin)
      (void)hipHostFree(chk_pin);
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
    const bool use_bvh = has_bvh && screen != 0;
    d.bvh = use_bvh ? bvh.p : nullptr;
    d.leaf_geo = use_bvh ? leaf_geo.p : nullptr;
    d.leaf_idx = use_bvh ? leaf_idx.p : nullptr;
    d.leaf_mat = use_bvh ? leaf_mat.p : nullptr;
    d.bvh_lds = use_bvh && bvh_depth <= rt::LDS_WALK_MAX_DEPTH && bvh_nodes <= rt::LDS_WALK_MAX_NODES;
    const bool use_tbvh = has_tbvh && screen != 0;
    d.tbvh = use_tbvh ? tbvh.p : nullptr;
    d.tleaf = use_tbvh ? tleaf.p : nullptr;
    d.tloose = use_tbvh ? tloose.p : nullptr;
    d.n_tleaf = use_tbvh ? (int)tleaf.n : 0;
    d.n_tloose = use_tbvh ? (int)tloose.n : 0;
    d.bins = bins_use && use_bvh ? bins.p : nullptr;
    d.bins_x = (bins_w + rt::BIN_PX - 1) / rt::BIN_PX;
    const bool use_cells = use_bvh && lcell_lights > 0 && screen == 1;
    d.lcells = use_cells ? lcells.p : nullptr;
    d.lcell_lights = use_cells ? lcell_lights : 0;
    for (int k = 0; k < rt::LCELL_MAX_LIGHTS; k++)
      d.lcell_reach[k] = lcell_reach[k];
    return d;
  }
};

namespace {

// need_colors false: a geometry-only scene (check_intersection), whose colour
// arrays may be NULL and are then taken as zeros.
int check_scene_desc(const scene_desc *s, bool need_colors = true) {
  if (!s)
    return fail(RT_ERR_BAD_ARG, "scene is NULL");
  if (s->t_size < 0 || s->s_size < 0 || s->l_size < 0)
    return fail(RT_ERR_BAD_ARG, "negative primitive or light count");
  if (s->t_size > 0 && (!s->tris || (need_colors && !s->color_tri)))
    return fail(RT_ERR_BAD_ARG, "t_size > 0 but tris/color_tri is NULL");
  if (s->s_size > 0 && (!s->spheres || !s->radius || (need_colors && !s->color_sphere)))
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

// src == nullptr uploads zeros.
template <typename T>
int upload(DevBuf<T> &dst, const void *src, size_t count) {
  HIP_TRY(dst.alloc(count));
  if (count && src)
    HIP_TRY(hipMemcpy(dst.p, src, count * sizeof(T), hipMemcpyHostToDevice));
  else if (count)
    HIP_TRY(hipMemset(dst.p, 0, count * sizeof(T)));
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
    S->bvh_depth = H.depth;
    S->bvh_nodes = H.nodes.size();
    if (H.valid) {
      if ((rc = upload(S->bvh, H.nodes.data(), H.nodes.size())) ||
          (rc = upload(S->leaf_idx, H.leaf_idx.data(), H.leaf_idx.size())))
        return rc;
      HIP_TRY(S->leaf_geo.alloc(H.leaf_idx.size()));
      HIP_TRY(S->leaf_mat.alloc(H.leaf_idx.size()));
      HIP_TRY(rt::launch_gather_leaves(S->sph_geo.p, S->col_sph.p,
                                       S->has_mat_sph ? S->mat_sph.p : nullptr, S->leaf_idx.p,
                                       S->leaf_geo.p, S->leaf_mat.p, (int)H.leaf_idx.size(),
                                       nullptr));
    }
  }
  // triangle hierarchy, once there are enough triangles for it to pay
  if (s->t_size >= kTriTreeMin) {
    rt::TriBvhHost H;
    rt::tri_bvh_build(s->tris, s->t_size, H);
    S->has_tbvh = H.valid && !H.nodes.empty();
    S->tbvh_nodes = H.nodes.size();
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
  // spheres by direction from the first lights, for the shadow rays of the primary hits
  if (g_light_cells && S->has_bvh && s->s_size >= kBinMinSpheres && s->l_size > 0) {
    const int nl = s->l_size < rt::LCELL_MAX_LIGHTS ? s->l_size : rt::LCELL_MAX_LIGHTS;
    DevBuf<int32_t> super;
    if (S->lcells.alloc((size_t)nl * rt::LCELL_COUNT * rt::LCELL_WORDS) == hipSuccess &&
        super.alloc((size_t)6 * rt::LCELL_SUPER_R * rt::LCELL_SUPER_R * rt::LCELL_SUPER_WORDS) ==
            hipSuccess) {
      const rt::SceneDev sd = S->dev();
      for (int l = 0; l < nl; l++) {
        const float *X = s->lights + 3 * l;
        // how far from the light a point of a sphere's surface can be, with room for
        // the rounding of a hit point
        double reach = 0;
        for (int i = 0; i < s->s_size; i++) {
          const double dx = (double)s->spheres[3 * i] - X[0], dy = (double)s->spheres[3 * i + 1] - X[1],
                       dz = (double)s->spheres[3 * i + 2] - X[2];
          reach = std::max(reach, std::sqrt(dx * dx + dy * dy + dz * dz) + std::fabs((double)s->radius[i]));
        }
        reach = reach * 1.01 + 0.01;
        if (!(reach < 1e18) || !std::isfinite(X[0]) || !std::isfinite(X[1]) || !std::isfinite(X[2]))
          break;
        S->lcell_reach[l] = (float)reach;
        HIP_TRY(rt::launch_bin_light(sd, X[0], X[1], X[2], (float)reach,
                                     S->lcells.p + (size_t)l * rt::LCELL_COUNT * rt::LCELL_WORDS,
                                     super.p, nullptr));
        S->lcell_lights = l + 1;
      }
      HIP_TRY(hipStreamSynchronize(nullptr)); // `super` dies here
    } else {
      (void)hipGetLastError();
      S->lcells.release(); // no lists: the shadow rays walk the hierarchy
    }
  }
  HIP_TRY(hipStreamSynchronize(nullptr)); // staging buffers die here
  S->eye_valid = false;
  S->screen = g_default_screen.load();
  S->trace_blocks = rt::trace_resident_blocks();
  S->walk_blocks = rt::walk_resident_blocks();
  HIP_TRY(S->ray_parts.alloc((size_t)rt::RAY_PARTS * rt::RAY_PART_STRIDE));
  HIP_TRY(hipMemset(S->ray_parts.p, 0, (size_t)rt::RAY_PARTS * rt::RAY_PART_STRIDE * sizeof(uint64_t)));
  HIP_TRY(hipEventCreateWithFlags(&S->ev_eye, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&S->ev_frame, hipEventDisableTiming));
  return RT_OK;
}

rt::CameraDev camera_dev(const camera *cam, int w, int h);

// Eye-relative records are rebuilt only when the eye moves, the spheres' lists by
// screen cell when the camera or the frame size changes.  Frames of one scene may
// come in on different streams: a rebuild waits for the last frame's launches (they
// read the old records), and every frame waits for the rebuild.
// may_allocate false (the entry point that allocates nothing): lists that exist
// for this view are used, none are made.
int scene_set_eye(rt_scene *S, const camera *cam, const rt_frame *fr, hipStream_t st,
                  bool may_allocate) {
  const bool eye_same = S->eye_valid && memcmp(S->eye, cam->origin, sizeof S->eye) == 0;
  const bool want_bins = g_primary_bins && S->has_bvh && S->screen == 1 && S->s_size >= kBinMinSpheres;
  const bool bins_same = S->bins_valid && S->bins_w == fr->w && S->bins_h == fr->h &&
                         memcmp(&S->bins_cam, cam, sizeof *cam) == 0;
  S->bins_use = want_bins && bins_same;
  const bool build_bins = want_bins && !bins_same && may_allocate;
  if (eye_same && !build_bins) {
    HIP_TRY(hipStreamWaitEvent(st, S->ev_eye, 0)); // no-op on the stream that built them
    return RT_OK;
  }
  if (S->frame_recorded)
    HIP_TRY(hipStreamWaitEvent(st, S->ev_frame, 0));
  if (!eye_same) {
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
  }
  if (build_bins) {
    const int bx = (fr->w + rt::BIN_PX - 1) / rt::BIN_PX, by = (fr->h + rt::BIN_PX - 1) / rt::BIN_PX;
    const int sx = (fr->w + rt::BIN_SUPER_PX - 1) / rt::BIN_SUPER_PX,
              sy = (fr->h + rt::BIN_SUPER_PX - 1) / rt::BIN_SUPER_PX;
    const size_t words = (size_t)bx * by * rt::BIN_WORDS;
    const size_t super_words = (size_t)sx * sy * rt::BIN_SUPER_WORDS;
    S->bins_valid = false;
    if (S->bins.n < words || S->bins_super.n < super_words) {
      HIP_TRY(hipDeviceSynchronize()); // nothing may still read the old lists
      if (S->bins.alloc(words) != hipSuccess || S->bins_super.alloc(super_words) != hipSuccess) {
        (void)hipGetLastError();
        S->bins.release(); // no lists: the walks serve the rays from the eye
        S->bins_super.release();
      }
    }
    if (S->bins.p) {
      HIP_TRY(rt::launch_bin_spheres(S->dev(), camera_dev(cam, fr->w, fr->h), fr->w, fr->h,
                                     S->bins.p, S->bins_super.p, st));
      S->bins_valid = true;
      S->bins_w = fr->w;
      S->bins_h = fr->h;
      S->bins_cam = *cam;
      S->bins_use = true;
    }
  }
  HIP_TRY(hipEventRecord(S->ev_eye, st));
  return RT_OK;
}

// Does a frame of this scene run its bounce levels as one launch (rt_trace.h)?
// That kernel walks the sphere hierarchy (a lane's stack of nodes in LDS, sized
// for trees of LDS_WALK_MAX_DEPTH levels) and loops over the triangles; scenes
// with a triangle hierarchy, scenes without any hierarchy or with a deeper one,
// more lights than one group, and the largest queue setting (which must not be
// able to overflow) take the launch-per-level path.
bool bvh_fits_lds(const rt_scene *S) {
  return S->has_bvh && S->bvh_depth <= rt::LDS_WALK_MAX_DEPTH &&
         S->bvh_nodes <= rt::LDS_WALK_MAX_NODES;
}
bool uses_trace(const rt_scene *S) {
  return S->has_bvh && S->screen != 0 && !S->has_tbvh && S->l_size <= kLightGroup &&
         bvh_fits_lds(S) &&
         S->rays_per_pixel < 256 && S->one_launch;
}

// What a frame's bounce levels need (see BounceNeeds).  max_bytes > 0: a fixed
// region; the grid of the one-launch kernel is cut down until it fits.
BounceNeeds bounce_needs(const rt_scene *S, int max_depth, size_t frame_pixels,
                         size_t owned_pixels, size_t max_bytes = 0) {
  BounceNeeds n;
  if (uses_trace(S)) {
    // a primary hit spawns at most two rays (mirror and refracted)
    const size_t want = 2 * owned_pixels + 65536;
    n.trace = true;
    n.capacity = (want + rt::QUEUE_SEGMENTS - 1) / rt::QUEUE_SEGMENTS * rt::QUEUE_SEGMENTS;
    n.levels = 3; // only level 2 is queued
    n.ctl_rows = 2;
    n.frame_pixels = frame_pixels;
    // a lane's item leaves at most 2 + lights items behind, and a wave works
    // depth-first: 64 x (2 + lights) items per level is what a stack holds at worst
    // when every batch is of one level; batches mix levels, hence the margin and
    // the overflow flag
    const int lights = S->l_size > 0 ? S->l_size : 1;
    n.stack_cap = (uint32_t)(64 * (2 + lights) * (max_depth + 2) *
                             std::max(1, S->rays_per_pixel / 4));
    // A small share of a frame (one rank's bands of an N-way split) runs on half the
    // resident grid: its waves then get enough rays to keep their lanes busy, and
    // the other half of the chip is free for the frames beside it (1/8 share with
    // four frames in flight 1.12 -> 1.02 ms; a whole frame on half the grid loses).
    int blocks = S->trace_blocks;
    if (owned_pixels < (size_t)g_trace_small_pixels && blocks >= 128)
      blocks = blocks * g_trace_small_grid_pct / 100;
    n.trace_waves = (size_t)blocks * 4;
    if (max_bytes) {
      BounceArena probe;
      while (blocks > 64 && probe.carve(nullptr, n) > max_bytes) {
        blocks -= 64;
        n.trace_waves = (size_t)blocks * 4;
      }
    }
    return n;
  }
  size_t want = (size_t)S->rays_per_pixel * owned_pixels + 65536;
  if (want > 0x0ffffff0ull)
    want = 0x0ffffff0ull;
  n.capacity = (want + rt::QUEUE_SEGMENTS - 1) / rt::QUEUE_SEGMENTS * rt::QUEUE_SEGMENTS;
  n.per_ray = S->l_size < kLightGroup ? (S->l_size > 0 ? S->l_size : 1) : kLightGroup;
  n.tri = S->has_tbvh;
  n.levels = max_depth + 2; // levels 0 .. max_depth + 1 (the last spawn target)
  n.ctl_rows = 3 + 3 * ((S->l_size + kLightGroup - 1) / kLightGroup + 1);
  n.frame_pixels = frame_pixels;
  return n;
}

// The scene's own arena: grown (never shrunk) to what the frame needs.  The old
// region is given back before the new one is asked for (both side by side may not
// fit), and the scene forgets it first: when the allocation fails the scene has no
// arena at all, and the next frame, perhaps a smaller one, starts from nothing
// instead of carving freed memory.
int ensure_arena(rt_scene *S, const BounceNeeds &need) {
  if (!S->arena.counts || !S->arena.have.covers(need)) {
    BounceNeeds n = S->arena.have;
    n.grow_to(need);
    BounceArena probe;
    const size_t bytes = probe.carve(nullptr, n);
    HIP_TRY(hipDeviceSynchronize()); // nothing may still use the old region
    S->arena = BounceArena();
    S->arena_mem.release();
    if (S->arena_limit && bytes > S->arena_limit)
      return fail(RT_ERR_CAPACITY,
                  "the frame's bounce machinery needs %zu bytes, the scene's limit is %zu "
                  "(rt_scene_set_arena_limit)",
                  bytes, S->arena_limit);
    const hipError_t e = S->arena_mem.alloc(bytes);
    if (e != hipSuccess) {
      (void)hipGetLastError(); // the failure is reported here, not by a later launch
      S->arena_mem.release();
      return fail(e == hipErrorOutOfMemory ? RT_ERR_CAPACITY : RT_ERR_HIP,
                  "hipMalloc of %zu bytes for the bounce machinery failed: %s", bytes,
                  hipGetErrorString(e));
    }
    BounceArena fresh;
    fresh.carve(S->arena_mem.p, n);
    // the per-pixel sums start at zero and k_compose leaves them at zero
    HIP_TRY(hipMemset(fresh.extra, 0, n.frame_pixels * sizeof(unsigned long long)));
    S->arena = fresh;
  }
  if (!S->side) {
    HIP_TRY(hipStreamCreateWithFlags(&S->side, hipStreamNonBlocking));
    for (int k = 0; k < 2; k++) {
      HIP_TRY(hipEventCreateWithFlags(&S->ev_fork[k], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&S->ev_join[k], hipEventDisableTiming));
    }
  }
  return RT_OK;
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

static int scene_create(const scene_desc *scene, bool need_colors, rt_scene **out) {
  if (!out)
    return fail(RT_ERR_BAD_ARG, "out is NULL");
  *out = nullptr;
  int rc;
  if ((rc = check_scene_desc(scene, need_colors)) || (rc = ensure_device()))
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

int rt_scene_create(const scene_desc *scene, rt_scene **out) {
  return scene_create(scene, true, out);
}

void rt_scene_destroy(rt_scene *scene) {
  if (!scene)
    return;
  int cur = 0;
  const bool have = hipGetDevice(&cur) == hipSuccess;
  (void)hipSetDevice(scene->device);
  delete scene;
  if (have)
    (void)hipSetDevice(cur);
}

// What of a frame this call renders.
struct FramePlan {
  int band_rows = 8, band_count = 1;
  int owned_tile_rows = 0;
  bool bounces = false;
  bool unlit = false;
  int trace_blocks = 0; // grid of the one-launch bounce kernel for this frame (bounce_needs)
};

static int plan_frame(const rt_scene *S, const camera *cam, const rt_frame *fr, FramePlan &P) {
  if (!S || !cam || !fr)
    return fail(RT_ERR_BAD_ARG, "scene, camera or frame is NULL");
  if (!fr->dev_fb)
    return fail(RT_ERR_BAD_ARG, "frame.dev_fb is NULL");
  if (fr->w <= 0 || fr->h <= 0)
    return fail(RT_ERR_BAD_ARG, "frame size %dx%d", fr->w, fr->h);
  if (fr->max_depth < 1)
    return fail(RT_ERR_BAD_ARG, "max_depth %d < 1", fr->max_depth);
  if (fr->flags & ~RT_FRAME_UNLIT)
    return fail(RT_ERR_BAD_ARG, "unknown frame flags 0x%x", fr->flags);
  P.band_count = fr->band_count < 1 ? 1 : fr->band_count;
  P.band_rows = P.band_count == 1 ? 8 : fr->band_rows;
  if (P.band_rows <= 0 || (P.band_rows & 7))
    return fail(RT_ERR_BAD_ARG, "band_rows %d must be a positive multiple of 8", P.band_rows);
  if (fr->band_index < 0 || fr->band_index >= P.band_count)
    return fail(RT_ERR_BAD_ARG, "band_index %d outside [0,%d)", fr->band_index, P.band_count);
  if (fr->max_depth > kMaxDepth)
    return fail(RT_ERR_BAD_ARG, "max_depth %d > %d", fr->max_depth, kMaxDepth);
  P.unlit = (fr->flags & RT_FRAME_UNLIT) != 0;
  // Bounces exist only where a material asks for them.
  P.bounces = !P.unlit && fr->max_depth > 1 && (S->has_mat_tri || S->has_mat_sph);
  if (P.bounces && S->l_size > 8000)
    return fail(RT_ERR_UNSUPPORTED, "bounces with more than 8000 lights (21-bit light sums)");
  if (P.bounces && (size_t)fr->w * fr->h > kMaxBouncePixels)
    return fail(RT_ERR_UNSUPPORTED,
                "bounces on a frame of more than 2^26 pixels (a bounce ray's tag holds a "
                "26-bit pixel index)");
  const int bands_total = (fr->h + P.band_rows - 1) / P.band_rows;
  const int owned = bands_total > fr->band_index
                        ? (bands_total - fr->band_index + P.band_count - 1) / P.band_count
                        : 0;
  P.owned_tile_rows = owned * (P.band_rows / 8);
  return RT_OK;
}

// Enqueues the launches of one frame on the caller's stream.  A: the bounce
// machinery (unused without bounces); side: a stream for the shadow walks, or
// null to keep everything on the caller's stream.
static int render_frame_launches(rt_scene *S, const BounceArena *A, hipStream_t side,
                                 const camera *cam, const rt_frame *fr, const FramePlan &P,
                                 uint64_t *dev_ray_count, uint64_t *dev_verify,
                                 uint64_t *ray_total, bool *summed);

// The kernels count rays in the scene's partial counters; the caller's word gets
// their sum at the end of the frame.
static int render_frame(rt_scene *S, const BounceArena *A, hipStream_t side, const camera *cam,
                        const rt_frame *fr, const FramePlan &P, uint64_t *dev_ray_count,
                        uint64_t *dev_verify) {
  bool summed = false; // the last kernel of a frame with bounces does the sum itself
  const int rc = render_frame_launches(S, A, side, cam, fr, P,
                                       dev_ray_count ? S->ray_parts.p : nullptr, dev_verify,
                                       dev_ray_count, &summed);
  if (rc != RT_OK && dev_ray_count) {
    // a frame that stopped half way (a launch failed) leaves what its first kernels
    // counted in the partial counters: they must be zero when the next frame starts
    (void)hipMemsetAsync(S->ray_parts.p, 0,
                         (size_t)rt::RAY_PARTS * rt::RAY_PART_STRIDE * sizeof(uint64_t),
                         (hipStream_t)fr->stream);
    (void)hipGetLastError();
  }
  if (rc != RT_OK || !dev_ray_count || summed)
    return rc;
  HIP_TRY(rt::launch_sum_rays(S->ray_parts.p, dev_ray_count, (hipStream_t)fr->stream));
  return RT_OK;
}

static int render_frame_launches(rt_scene *S, const BounceArena *A, hipStream_t side,
                                 const camera *cam, const rt_frame *fr, const FramePlan &P,
                                 uint64_t *dev_ray_count, uint64_t *dev_verify,
                                 uint64_t *ray_total, bool *summed) {
  hipStream_t st = (hipStream_t)fr->stream;
  rt::FrameDev f;
  f.fb = fr->dev_fb;
  f.ids = fr->dev_ids;
  f.ray_count = dev_ray_count;
  f.w = fr->w;
  f.h = fr->h;
  f.max_depth = fr->max_depth;
  f.band_rows = P.band_rows;
  f.band_index = fr->band_index;
  f.band_count = P.band_count;
  f.unlit = P.unlit ? 1 : 0;
  rt::BounceDev bd;
  memset(&bd, 0, sizeof bd);
  bd.max_depth = 1;
  if (P.bounces) {
    // only the levels this frame can reach, and the overflow flag
    const int used_levels = A->have.trace ? 3 : fr->max_depth + 2;
    const size_t row_bytes = (size_t)used_levels * A->have.ctl_rows * kRowWords * sizeof(uint32_t);
    if (used_levels == A->have.levels) {
      // the flag is the word behind the last row (and the region is padded to 256 bytes)
      HIP_TRY(hipMemsetAsync(A->counts, 0, row_bytes + 16, st));
    } else {
      HIP_TRY(hipMemsetAsync(A->counts, 0, row_bytes, st));
      HIP_TRY(hipMemsetAsync(A->overflow_flag(), 0, sizeof(uint32_t), st));
    }
    bd.max_depth = fr->max_depth;
    bd.extra = A->extra;
    bd.out = A->queue_for_level(2);
  }
  const rt::SceneDev sd = S->dev();
  // 0 plain loop, 1 automatic (hierarchy when the scene has one, else the beam
  // screen), 2 beam screen, 3 hierarchy
  int accel = RT_BEAM_NONE;
  if (S->screen == 2 || ((S->screen == 1 || S->screen == 3) && !S->has_bvh))
    accel = RT_BEAM_APEX;
  else if (S->screen == 1 || S->screen == 3)
    accel = RT_ACCEL_BVH;
  S->timed_stages = 0;
  if (S->timing)
    HIP_TRY(hipEventRecord(S->ev_t[0], st));
  HIP_TRY(rt::launch_render(sd, camera_dev(cam, fr->w, fr->h), f, bd, P.owned_tile_rows, accel,
                            (unsigned long long *)dev_verify, st));
  if (S->timing) {
    HIP_TRY(hipEventRecord(S->ev_t[1], st));
    S->timed_stages = 1;
  }
  if (!P.bounces)
    return RT_OK;
  // With a hierarchy each level runs as walk / shade / shadow-walk launches with
  // dynamic ray fetch (rt_walk.h); otherwise, or in the self-checking build, as
  // one k_bounce launch per level.
  const bool stream_levels = (sd.bvh != nullptr || sd.tbvh != nullptr) && !dev_verify;
  const bool tri_walks = sd.tbvh != nullptr;
  if (A->have.trace && !dev_verify) {
    // every level in one launch: the waves keep what their rays spawn (rt_trace.h)
    rt::TraceArgs ta;
    memset(&ta, 0, sizeof ta);
    ta.in = A->queue_for_level(2);
    ta.cursor = A->ctl(2, 1);
    ta.stack = A->wave_stack;
    ta.hits = A->wave_hits;
    ta.save = A->wave_save;
    ta.stack_cap = A->have.stack_cap;
    ta.hit_cap = rt::TRACE_HIT_CAP;
    ta.extra = A->extra;
    ta.ray_count = dev_ray_count;
    ta.overflow = A->overflow_flag();
    ta.max_depth = fr->max_depth;
    ta.fetch_at = g_trace_fetch_at;
    HIP_TRY(rt::launch_trace(sd, ta, S->bvh_depth, P.trace_blocks, st)); // <= A->have.trace_waves / 4
    if (S->timing)
      HIP_TRY(hipEventRecord(S->ev_t[2], st));
    HIP_TRY(rt::launch_compose(fr->dev_fb, A->extra, fr->w, fr->h, P.band_rows,
                               P.band_count == 1 ? 0 : fr->band_index, P.band_count,
                               dev_ray_count, dev_ray_count ? ray_total : nullptr, st));
    *summed = dev_ray_count != nullptr;
    if (S->timing) {
      HIP_TRY(hipEventRecord(S->ev_t[3], st));
      S->timed_stages = 3;
    }
    return RT_OK;
  }
  bool joined[2] = {false, false};
  for (int level = 2; level <= fr->max_depth; level++) {
    bd.in = A->queue_for_level(level);
    bd.out = A->queue_for_level(level + 1);
    if (!stream_levels) {
      HIP_TRY(rt::launch_bounce(sd, bd, S->screen != 0, dev_ray_count,
                                (unsigned long long *)dev_verify, kBounceBlocks, st));
      continue;
    }
    rt::HitList hl, tl;
    hl.rec = A->hit_rec;
    hl.idx = A->hit_idx;
    tl.rec = tri_walks ? A->tri_hit_rec : nullptr;
    tl.idx = tri_walks ? A->tri_hit_idx : nullptr;
    rt::WalkArgs wa;
    memset(&wa, 0, sizeof wa);
    wa.fetch_at = g_walk_fetch_at;
    wa.in = bd.in;
    wa.extra = A->extra;
    // Each ray is counted once, by the first walk it goes through.  The sphere
    // walk goes first: it is several times cheaper per ray, and the |P| of its hit
    // then bounds the triangle walk (a triangle only matters at |P| <= the
    // sphere's, renderer.cpp:181).
    wa.cursor = A->ctl(level, 1);
    wa.hits = hl;
    wa.ray_count = dev_ray_count;
    HIP_TRY(rt::launch_walk(sd, wa, false, false, S->walk_blocks, st));
    if (tri_walks) {
      wa.cursor = A->ctl(level, 2);
      wa.hits = tl;
      wa.prior = hl;
      wa.ray_count = nullptr;
      HIP_TRY(rt::launch_walk(sd, wa, false, true, S->walk_blocks, st));
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
    // one group of lights: its shadow walks go to the side stream; more groups
    // share the item buffer and stay in order on the caller's stream
    const bool fork = n_groups == 1 && S->l_size > 0 && side != nullptr;
    const int par = level & 1;
    for (int g = 0; g < n_groups; g++) {
      sa.light_begin = g * kLightGroup;
      sa.light_end = S->l_size < (g + 1) * kLightGroup ? S->l_size : (g + 1) * kLightGroup;
      sa.spawn = g == 0;
      sa.shadow.items = A->shadow_items[fork ? par : 0];
      sa.shadow.count = A->ctl(level, 3 + 3 * g);
      sa.shadow.seg_capacity = bd.in.seg_capacity * A->have.per_ray;
      sa.shadow.overflow = A->overflow_flag();
      if (fork && joined[par]) // the shadow walk two levels up has read this buffer
        HIP_TRY(hipStreamWaitEvent(st, S->ev_join[par], 0));
      HIP_TRY(rt::launch_shade(sd, sa, kBounceBlocks, st));
      if (S->l_size > 0) {
        hipStream_t ws_stream = st;
        if (fork) {
          HIP_TRY(hipEventRecord(S->ev_fork[par], st));
          HIP_TRY(hipStreamWaitEvent(side, S->ev_fork[par], 0));
          ws_stream = side;
        }
        rt::WalkArgs ws;
        memset(&ws, 0, sizeof ws);
        ws.fetch_at = g_walk_fetch_at_shadow;
        ws.in = sa.shadow;
        ws.extra = A->extra;
        // spheres first (the cheaper walk); only what they leave unoccluded is
        // worth a triangle walk
        ws.cursor = A->ctl(level, 4 + 3 * g);
        ws.ray_count = dev_ray_count;
        ws.last_stage = tri_walks ? 0 : 1;
        HIP_TRY(rt::launch_walk(sd, ws, true, false, S->walk_blocks, ws_stream));
        if (tri_walks) {
          ws.cursor = A->ctl(level, 5 + 3 * g);
          ws.ray_count = nullptr;
          ws.last_stage = 1;
          HIP_TRY(rt::launch_walk(sd, ws, true, true, S->walk_blocks, ws_stream));
        }
        if (fork) {
          HIP_TRY(hipEventRecord(S->ev_join[par], side));
          joined[par] = true;
        }
      }
    }
  }
  for (int k = 0; k < 2; k++)
    if (joined[k])
      HIP_TRY(hipStreamWaitEvent(st, S->ev_join[k], 0));
  if (S->timing)
    HIP_TRY(hipEventRecord(S->ev_t[2], st));
  HIP_TRY(rt::launch_compose(fr->dev_fb, A->extra, fr->w, fr->h, P.band_rows,
                               P.band_count == 1 ? 0 : fr->band_index, P.band_count,
                               dev_ray_count, dev_ray_count ? ray_total : nullptr, st));
  *summed = dev_ray_count != nullptr;
  if (S->timing) {
    HIP_TRY(hipEventRecord(S->ev_t[3], st));
    S->timed_stages = 3;
  }
  return RT_OK;
}

static int render_device_impl(rt_scene *S, const camera *cam, const rt_frame *fr,
                              uint64_t *dev_ray_count, uint64_t *dev_verify) {
  FramePlan P;
  int rc = plan_frame(S, cam, fr, P);
  if (rc)
    return rc;
  hipStream_t st = (hipStream_t)fr->stream;
  HIP_TRY(hipSetDevice(S->device));
  if (P.bounces) {
    const int one_launch = S->one_launch;
    if (dev_verify)
      S->one_launch = 0; // the self-checking kernels exist for the launch-per-level path
    const BounceNeeds need = bounce_needs(S, fr->max_depth, (size_t)fr->w * fr->h,
                                          (size_t)P.owned_tile_rows * 8 * fr->w);
    S->one_launch = one_launch;
    P.trace_blocks = (int)(need.trace_waves / 4);
    if ((rc = ensure_arena(S, need)))
      return rc;
  }
  if ((rc = scene_set_eye(S, cam, fr, st, true)))
    return rc;
  rc = render_frame(S, &S->arena, S->side_stream ? S->side : nullptr, cam, fr, P, dev_ray_count,
                    dev_verify);
  if (rc)
    return rc;
  HIP_TRY(hipEventRecord(S->ev_frame, st));
  S->frame_recorded = true;
  return RT_OK;
}

int rt_render_device(rt_scene *S, const camera *cam, const rt_frame *fr,
                     uint64_t *dev_ray_count) {
  return render_device_impl(S, cam, fr, dev_ray_count, nullptr);
}

int rt_verify_sphere_lists(rt_scene *S, const camera *cam, int w, int h, uint64_t out[4]) {
  if (!S || !cam || !out)
    return fail(RT_ERR_BAD_ARG, "scene, camera or out is NULL");
  if (w <= 0 || h <= 0)
    return fail(RT_ERR_BAD_ARG, "frame size %d x %d", w, h);
  for (int k = 0; k < 4; k++)
    out[k] = 0;
  HIP_TRY(hipSetDevice(S->device));
  rt_frame fr;
  memset(&fr, 0, sizeof fr);
  fr.w = w;
  fr.h = h;
  int rc = scene_set_eye(S, cam, &fr, nullptr, true); // the view's lists, made if need be
  if (rc)
    return rc;
  DevBuf<unsigned long long> dev;
  HIP_TRY(dev.alloc(4));
  HIP_TRY(hipMemsetAsync(dev.p, 0, 4 * sizeof(unsigned long long), nullptr));
  HIP_TRY(rt::launch_verify_lists(S->dev(), camera_dev(cam, w, h), w, h, dev.p, nullptr));
  HIP_TRY(hipMemcpy(out, dev.p, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipEventRecord(S->ev_frame, nullptr));
  S->frame_recorded = true;
  return RT_OK;
}

int rt_render_device_selfcheck(rt_scene *S, const camera *cam, const rt_frame *fr,
                               uint64_t *dev_ray_count, uint64_t *dev_stats) {
  if (!dev_stats)
    return fail(RT_ERR_BAD_ARG, "dev_stats is NULL");
  return render_device_impl(S, cam, fr, dev_ray_count, dev_stats);
}

// ---- the entry point that allocates nothing -----------------------------------
static std::mutex g_scratch_mutex;
static unsigned char *g_scratch_ptr[64]; // per device, filled on first use

// Address of `scratch` on `device` (which must be the current device).
static int noalloc_scratch(int device, unsigned char **out) {
  if (device < 0 || device >= 64)
    return fail(RT_ERR_UNSUPPORTED, "device %d: rt_render_device_noalloc serves devices 0..63",
                device);
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[device]) {
    void *p = nullptr;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p)
      return fail(RT_ERR_HIP, "hipGetSymbolAddress returned a null address");
    g_scratch_ptr[device] = (unsigned char *)p;
    if (getenv("RT_AMD_DEBUG"))
      fprintf(stderr, "rt_amd: fixed array of device %d at %p (%zu bytes)\n", device, p,
              (size_t)RT_NOALLOC_SCRATCH_BYTES);
  }
  *out = g_scratch_ptr[device];
  return RT_OK;
}

int rt_noalloc_scratch_needed(rt_scene *S, const rt_frame *fr, uint64_t *bytes) {
  if (!bytes)
    return fail(RT_ERR_BAD_ARG, "bytes is NULL");
  *bytes = 0;
  camera cam;
  rt_camera_default(&cam);
  FramePlan P;
  if (!fr)
    return fail(RT_ERR_BAD_ARG, "frame is NULL");
  rt_frame sized = *fr; // only the sizes matter here: no framebuffer needed
  if (!sized.dev_fb)
    sized.dev_fb = reinterpret_cast<uint8_t *>(uintptr_t(256));
  int rc = plan_frame(S, &cam, &sized, P);
  if (rc)
    return rc;
  if (P.bounces) {
    BounceArena probe;
    *bytes = probe.carve(nullptr, bounce_needs(S, fr->max_depth, (size_t)fr->w * fr->h,
                                               (size_t)P.owned_tile_rows * 8 * fr->w,
                                               RT_NOALLOC_SCRATCH_BYTES));
  }
  return RT_OK;
}

int rt_render_device_noalloc(rt_scene *S, const camera *cam, const rt_frame *fr,
                             uint64_t *dev_ray_count, uint32_t *dev_overflow) {
  FramePlan P;
  int rc = plan_frame(S, cam, fr, P);
  if (rc)
    return rc;
  hipStream_t st = (hipStream_t)fr->stream;
  HIP_TRY(hipSetDevice(S->device));
  BounceArena A;
  if (P.bounces) {
    const BounceNeeds need =
        bounce_needs(S, fr->max_depth, (size_t)fr->w * fr->h,
                     (size_t)P.owned_tile_rows * 8 * fr->w, RT_NOALLOC_SCRATCH_BYTES);
    P.trace_blocks = (int)(need.trace_waves / 4);
    const size_t bytes = A.carve(nullptr, need);
    if (bytes > (size_t)RT_NOALLOC_SCRATCH_BYTES)
      return fail(RT_ERR_CAPACITY,
                  "the frame's bounce machinery needs %zu bytes, the fixed array holds %zu",
                  bytes, (size_t)RT_NOALLOC_SCRATCH_BYTES);
    unsigned char *base = nullptr;
    if ((rc = noalloc_scratch(S->device, &base)))
      return rc;
    A.carve(base, need);
    // the array keeps the last call's contents: the per-pixel sums must start at
    // zero (render_frame clears the control words and the overflow flag itself)
    HIP_TRY(hipMemsetAsync(A.extra, 0, need.frame_pixels * sizeof(unsigned long long), st));
  }
  if ((rc = scene_set_eye(S, cam, fr, st, false)))
    return rc;
  if ((rc = render_frame(S, &A, nullptr, cam, fr, P, dev_ray_count, nullptr)))
    return rc;
  if (dev_overflow) {
    if (P.bounces)
      HIP_TRY(hipMemcpyAsync(dev_overflow, A.overflow_flag(), sizeof(uint32_t),
                             hipMemcpyDeviceToDevice, st));
    else
      HIP_TRY(hipMemsetAsync(dev_overflow, 0, sizeof(uint32_t), st));
  }
  HIP_TRY(hipEventRecord(S->ev_frame, st));
  S->frame_recorded = true;
  return RT_OK;
}

int rt_scene_queue_overflowed(rt_scene *S, int *overflowed) {
  if (!S || !overflowed)
    return fail(RT_ERR_BAD_ARG, "scene or result is NULL");
  *overflowed = 0;
  HIP_TRY(hipSetDevice(S->device));
  if (S->arena.counts) {
    uint32_t flag = 0;
    HIP_TRY(hipMemcpy(&flag, S->arena.overflow_flag(), sizeof flag, hipMemcpyDeviceToHost));
    *overflowed = flag != 0;
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

int rt_scene_set_arena_limit(rt_scene *S, uint64_t max_bytes) {
  if (!S)
    return fail(RT_ERR_BAD_ARG, "scene is NULL");
  S->arena_limit = (size_t)max_bytes;
  return RT_OK;
}

int rt_scene_set_screen(rt_scene *S, int mode) {
  if (!S)
    return fail(RT_ERR_BAD_ARG, "scene is NULL");
  if (mode < 0 || mode > 3)
    return fail(RT_ERR_BAD_ARG, "screen mode %d outside 0..3", mode);
  S->screen = mode;
  return RT_OK;
}

int rt_scene_time_kernels(rt_scene *S, int on) {
  if (!S)
    return fail(RT_ERR_BAD_ARG, "scene is NULL");
  HIP_TRY(hipSetDevice(S->device));
  if (on && !S->ev_t[0])
    for (hipEvent_t &e : S->ev_t)
      HIP_TRY(hipEventCreate(&e));
  S->timing = on != 0;
  S->timed_stages = 0;
  return RT_OK;
}

int rt_scene_kernel_times(rt_scene *S, float ms[3]) {
  if (!S || !ms)
    return fail(RT_ERR_BAD_ARG, "scene or result is NULL");
  if (!S->timing || S->timed_stages == 0)
    return fail(RT_ERR_BAD_ARG, "no timed frame (rt_scene_time_kernels, then render)");
  HIP_TRY(hipSetDevice(S->device));
  ms[0] = ms[1] = ms[2] = 0.f;
  HIP_TRY(hipEventSynchronize(S->ev_t[S->timed_stages]));
  for (int k = 0; k < S->timed_stages; k++)
    HIP_TRY(hipEventElapsedTime(&ms[k], S->ev_t[k], S->ev_t[k + 1]));
  return RT_OK;
}

int rt_scene_describe(const rt_scene *S, int64_t info[8]) {
  if (!S || !info)
    return fail(RT_ERR_BAD_ARG, "scene or info is NULL");
  for (int k = 0; k < 8; k++)
    info[k] = 0;
  info[0] = S->has_bvh ? (int64_t)S->bvh_nodes : 0;
  info[1] = S->has_bvh ? S->bvh_depth : 0;
  info[2] = bvh_fits_lds(S) ? 1 : 0;
  info[3] = uses_trace(S) ? 1 : 0;
  info[4] = S->has_tbvh ? (int64_t)S->tbvh_nodes : 0;
  // the sphere lists (waits for the device and copies them: an inspection call)
  info[5] = S->lcell_lights;
  info[6] = info[7] = -1;
  if (S->bins_valid || S->lcell_lights > 0) {
    HIP_TRY(hipSetDevice(S->device));
    HIP_TRY(hipDeviceSynchronize());
    std::vector<int32_t> host;
    if (S->bins_valid) {
      const size_t cells = (size_t)((S->bins_w + rt::BIN_PX - 1) / rt::BIN_PX) *
                           ((S->bins_h + rt::BIN_PX - 1) / rt::BIN_PX);
      host.resize(cells * rt::BIN_WORDS);
      HIP_TRY(hipMemcpy(host.data(), S->bins.p, host.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
      info[6] = 0;
      for (size_t c = 0; c < cells; c++)
        info[6] += host[c * rt::BIN_WORDS] >= 0;
    }
    if (S->lcell_lights > 0) {
      const size_t cells = (size_t)S->lcell_lights * rt::LCELL_COUNT;
      host.resize(cells * rt::LCELL_WORDS);
      HIP_TRY(hipMemcpy(host.data(), S->lcells.p, host.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
      info[7] = 0;
      for (size_t c = 0; c < cells; c++)
        info[7] += host[c * rt::LCELL_WORDS] < 0;
    }
  }
  return RT_OK;
}

int rt_scene_set_one_launch(rt_scene *S, int on) {
  if (!S)
    return fail(RT_ERR_BAD_ARG, "scene is NULL");
  S->one_launch = on != 0;
  return RT_OK;
}

int rt_scene_set_side_stream(rt_scene *S, int on) {
  if (!S)
    return fail(RT_ERR_BAD_ARG, "scene is NULL");
  S->side_stream = on != 0;
  return RT_OK;
}

void rt_set_screen(int mode) { g_default_screen.store(mode < 0 || mode > 3 ? 1 : mode); }
int rt_get_screen(void) { return g_default_screen.load(); }

int rt_band_plan(int h, int band_rows, int band_index, int band_count, int32_t *row_begin,
                 int32_t *row_end, int max_bands) {
  if (h <= 0 || band_rows <= 0 || (band_rows & 7) || band_count < 1 || band_index < 0 ||
      band_index >= band_count || max_bands < 0 || (max_bands > 0 && (!row_begin || !row_end)))
    return fail(RT_ERR_BAD_ARG, "rt_band_plan: bad arguments");
  const int bands_total = (h + band_rows - 1) / band_rows;
  int n = 0;
  for (int b = band_index; b < bands_total; b += band_count, n++)
    if (n < max_bands) {
      row_begin[n] = b * band_rows;
      row_end[n] = std::min(h, (b + 1) * band_rows);
    }
  return n;
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
        std::vector<rt::BvhBox> boxes; // ancestors' child entries, as the walk decodes them
      };
      std::vector<Item> todo;
      todo.push_back(Item{0, {}});
      size_t levels = 0;
      while (!todo.empty()) {
        Item it = todo.back();
        todo.pop_back();
        const rt::BvhNode &N = H.nodes[it.node];
        for (int c = 0; c < rt::BVH_WIDTH; c++) {
          const int32_t link = N.link[c];
          if (link == rt::BVH_EMPTY)
            continue;
          auto boxes = it.boxes;
          boxes.push_back(rt::bvh_child_box(N, c));
          levels = std::max(levels, boxes.size());
          if (link >= 0) {
            todo.push_back(Item{link, boxes});
            continue;
          }
          const int idx = H.leaf_idx[(~link) >> 3];
          seen[idx]++;
          const double r = std::fabs((double)s->radius[idx]);
          for (auto &bx : boxes) {
            const double lo[3] = {bx.lo[0], bx.lo[1], bx.lo[2]};
            const double hi[3] = {bx.hi[0], bx.hi[1], bx.hi[2]};
            double d2 = 0;
            for (int d = 0; d < 3; d++) {
              const double cc = s->spheres[3 * idx + d];
              if (cc - r < lo[d] || cc + r > hi[d])
                out[1]++; // sphere sticks out of an ancestor box
              const double g = lo[d] > 0 ? lo[d] : (hi[d] < 0 ? -hi[d] : 0.0);
              d2 += g * g;
            }
            if ((double)bx.near > std::sqrt(d2))
              out[2]++; // distance bound above the true distance
          }
        }
      }
      for (int i = 0; i < s->s_size; i++)
        if (seen[i] != 1)
          out[3]++; // sphere missing or duplicated
      size_t least = 1; // levels of the shallowest four-way tree over s_size leaves
      for (long reach = rt::BVH_WIDTH; reach < s->s_size; reach *= rt::BVH_WIDTH)
        least++;
      if (levels != least || (int)levels != H.depth)
        out[3]++; // deeper than the walks' stacks are sized for, or misreported
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
      size_t levels = 0;
      long in_tree = 0;
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
          levels = std::max(levels, bounds.size());
          if (link >= 0) {
            todo.push_back(Item{link, bounds});
            continue;
          }
          const int idx = H.leaf_idx[(~link) >> 3];
          in_tree++;
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
      size_t least = 1; // levels of the shallowest four-way tree over the leaves
      for (long reach = rt::BVH_WIDTH; reach < in_tree; reach *= rt::BVH_WIDTH)
        least++;
      if (levels != least)
        out[3]++; // deeper than the walks' stacks are sized for
    }
  }
  return RT_OK;
}

// ---- scenes kept between one-call renders -------------------------------------
// gpu_render* receive host arrays every time.  The device copy of the last scene
// (per device) is kept and reused when the next call brings the same pointers,
// sizes and bytes; two 64-bit hashes, one of the geometry and one of everything
// else, decide the last point (a million triangles hash in a few ms, their
// hierarchy takes most of a second to build).  The hashes are not cryptographic: they catch
// arrays edited in place between calls, not an adversary.  A host that never
// edits its arrays in place can switch the hashing off (rt_cache_trust_pointers).
} // extern "C"

namespace {

constexpr int kMaxCachedDevices = 16;
constexpr int kFrameChunks = 4; // render_cached: row chunks whose copies back overlap the render

struct SceneKey {
  scene_desc d;
  uint64_t geo_hash = 0;  // tris, spheres, radius
  uint64_t rest_hash = 0; // colours, lights, materials
  bool operator==(const SceneKey &o) const {
    return memcmp(&d, &o.d, sizeof d) == 0 && geo_hash == o.geo_hash && rest_hash == o.rest_hash;
  }
  // the same primitives, whatever their colours, lights and materials: all that
  // check_intersection (renderer.hpp:21-23) looks at
  bool same_geometry(const SceneKey &o) const {
    return d.tris == o.d.tris && d.t_size == o.d.t_size && d.spheres == o.d.spheres &&
           d.radius == o.d.radius && d.s_size == o.d.s_size && geo_hash == o.geo_hash;
  }
};

std::atomic<int> g_trust_pointers{0}; // rt_cache_trust_pointers

uint64_t hash_bytes(uint64_t h, const void *p, size_t n) {
  // four interleaved multiply-xor lanes over 8-byte words, then the tail
  const unsigned char *b = (const unsigned char *)p;
  uint64_t l[4] = {h ^ 0x9e3779b97f4a7c15ull, h + 0xc2b2ae3d27d4eb4full, ~h, h * 0xff51afd7ed558ccdull};
  size_t i = 0;
  for (; i + 32 <= n; i += 32)
    for (int k = 0; k < 4; k++) {
      uint64_t w;
      memcpy(&w, b + i + 8 * k, 8);
      l[k] = (l[k] ^ w) * 0x100000001b3ull;
      l[k] ^= l[k] >> 29;
    }
  uint64_t r = (l[0] * 31 + l[1]) * 31 + (l[2] * 31 + l[3]);
  for (; i < n; i++)
    r = (r ^ b[i]) * 0x100000001b3ull;
  return r ^ (uint64_t)n;
}

// geometry_only: the caller is check_intersection; nothing but the geometry is
// looked at, so nothing else is hashed (one ray per call must stay cheap).
SceneKey scene_key(const scene_desc *s, bool geometry_only = false) {
  SceneKey k;
  memset(&k.d, 0, sizeof k.d); // padding bytes compare equal
  k.d.tris = s->tris;
  k.d.color_tri = s->color_tri;
  k.d.t_size = s->t_size;
  k.d.spheres = s->spheres;
  k.d.radius = s->radius;
  k.d.color_sphere = s->color_sphere;
  k.d.s_size = s->s_size;
  k.d.lights = s->lights;
  k.d.l_size = s->l_size;
  k.d.mat_tri = s->mat_tri;
  k.d.mat_sphere = s->mat_sphere;
  if (g_trust_pointers.load())
    return k;
  uint64_t h = 0x6a09e667f3bcc908ull;
  h = hash_bytes(h, s->tris, sizeof(float) * 9 * (size_t)s->t_size);
  h = hash_bytes(h, s->spheres, sizeof(float) * 3 * (size_t)s->s_size);
  h = hash_bytes(h, s->radius, sizeof(float) * (size_t)s->s_size);
  k.geo_hash = h;
  if (geometry_only)
    return k;
  h = 0xbb67ae8584caa73bull;
  if (s->color_tri)
    h = hash_bytes(h, s->color_tri, 3 * (size_t)s->t_size);
  if (s->color_sphere)
    h = hash_bytes(h, s->color_sphere, 3 * (size_t)s->s_size);
  h = hash_bytes(h, s->lights, sizeof(float) * 3 * (size_t)s->l_size);
  if (s->mat_tri)
    h = hash_bytes(h, s->mat_tri, sizeof(rt_material) * (size_t)s->t_size);
  if (s->mat_sphere)
    h = hash_bytes(h, s->mat_sphere, sizeof(rt_material) * (size_t)s->s_size);
  k.rest_hash = h;
  return k;
}

// One kept scene.  Slot d belongs to device d for the one-device entry points;
// gpu_render_multi uses slots 0 .. n_devices-1 (which are devices 0 .. n_devices-1
// unless several slots are mapped onto one device for a rehearsal, see there).
struct CachedScene {
  std::mutex mutex;       // held by the call that renders from this slot
  rt_scene *scene = nullptr;
  int device = -1;        // the device the scene and the buffers below live on
  SceneKey key;
  // frame buffers of the one-call entry points, grow-only
  DevBuf<uint8_t> fb;
  DevBuf<int32_t> ids;
  DevBuf<uint64_t> cnt;
  hipStream_t stream = nullptr; // this slot's stream (gpu_render_multi, chunked frames)
  hipStream_t copy_stream = nullptr;          // chunked frames: the copies back
  hipEvent_t ev_chunk[kFrameChunks] = {};     // chunk k rendered

  // Everything this slot holds, on the device it was made on.
  void drop() {
    if (!scene && !fb.p && !ids.p && !cnt.p && !stream && !copy_stream)
      return;
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    if (device >= 0)
      (void)hipSetDevice(device);
    rt_scene_destroy(scene);
    scene = nullptr;
    fb.release();
    ids.release();
    cnt.release();
    if (stream)
      (void)hipStreamDestroy(stream);
    stream = nullptr;
    if (copy_stream)
      (void)hipStreamDestroy(copy_stream);
    copy_stream = nullptr;
    for (hipEvent_t &e : ev_chunk) {
      if (e)
        (void)hipEventDestroy(e);
      e = nullptr;
    }
    device = -1;
    if (have)
      (void)hipSetDevice(cur);
  }
};
CachedScene g_cache[kMaxCachedDevices];
std::atomic<uint64_t> g_cache_builds{0}; // scenes built by cached_scene (for tests)

// The resident scene for `s` in slot C on the current device, built if the kept
// one is a different scene.  Call with C.mutex held.  need_colors false: the
// caller is check_intersection, which any kept scene of the same geometry serves.
int cached_scene(CachedScene &C, const scene_desc *s, const SceneKey &key, bool need_colors = true) {
  int rc;
  if ((rc = check_scene_desc(s, need_colors)) || (rc = ensure_device()))
    return rc;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  const int mode = g_default_screen.load();
  const bool usable = C.scene && C.device == dev &&
                      (need_colors ? C.key == key : C.key.same_geometry(key));
  if (!usable) {
    C.drop(); // the buffers too when the slot moves to another device
    C.device = dev;
    if ((rc = scene_create(s, need_colors, &C.scene)))
      return rc;
    C.key = key;
    g_cache_builds++;
  }
  C.scene->screen = mode; // the one-call entry points follow rt_set_screen
  return RT_OK;
}

// One frame on the current device from the kept scene into the kept buffers
// (launches only; the caller synchronises).
int cached_frame(CachedScene &C, const camera *cam, bool want_ids, int w, int h, int max_depth,
                 int flags, int band_rows, int band_index, int band_count, hipStream_t st,
                 bool reset_count = true) {
  const size_t npx = (size_t)w * h;
  if (C.fb.n < npx * 4)
    HIP_TRY(C.fb.alloc(npx * 4));
  if (want_ids && C.ids.n < npx)
    HIP_TRY(C.ids.alloc(npx));
  if (!C.cnt.p)
    HIP_TRY(C.cnt.alloc(1));
  if (reset_count)
    HIP_TRY(hipMemsetAsync(C.cnt.p, 0, sizeof(uint64_t), st));
  rt_frame fr;
  memset(&fr, 0, sizeof fr);
  fr.dev_fb = C.fb.p;
  fr.dev_ids = want_ids ? C.ids.p : nullptr;
  fr.w = w;
  fr.h = h;
  fr.max_depth = max_depth;
  fr.band_rows = band_rows;
  fr.band_index = band_index;
  fr.band_count = band_count;
  fr.flags = flags;
  fr.stream = st;
  return rt_render_device(C.scene, cam, &fr, C.cnt.p);
}

// A frame of a one-call entry point dropped bounce rays, or its bounce machinery
// could not be allocated: what to change before rendering it again.  The
// one-launch path keeps a stack per resident wave, which grows with the queue
// setting times five thousand waves; the launch-per-level path has frame-wide
// queues only, and at 256 rays per pixel they cannot overflow.  So: leave the
// one-launch path first, then enlarge the queues.  False: nothing left to try.
bool next_attempt(rt_scene *S) {
  if (uses_trace(S)) {
    S->one_launch = 0;
    return true;
  }
  if (S->rays_per_pixel >= 256)
    return false;
  S->rays_per_pixel = S->rays_per_pixel * 8 > 256 ? 256 : S->rays_per_pixel * 8;
  return true;
}

int render_cached(const scene_desc *scene, const camera *cam, uint8_t *rgb, int32_t *hit_ids,
                  uint64_t *n_rays, int w, int h, int max_depth, int flags) {
  if (!cam || !rgb)
    return fail(RT_ERR_BAD_ARG, "camera or output buffer is NULL");
  if (w <= 0 || h <= 0)
    return fail(RT_ERR_BAD_ARG, "frame size %dx%d", w, h);
  int rc = check_scene_desc(scene);
  if (rc || (rc = ensure_device()))
    return rc;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxCachedDevices)
    return fail(RT_ERR_UNSUPPORTED, "device %d: only devices 0..%d keep a scene", dev,
                kMaxCachedDevices - 1);
  const SceneKey key = scene_key(scene);
  CachedScene &C = g_cache[dev];
  // one slot, one call at a time; calls on other devices do not wait for this one
  std::lock_guard<std::mutex> lock(C.mutex);
  if ((rc = cached_scene(C, scene, key)))
    return rc;
  rt_scene *S = C.scene;
  const size_t npx = (size_t)w * h;
  const bool bounces =
      !(flags & RT_FRAME_UNLIT) && max_depth > 1 && (S->has_mat_tri || S->has_mat_sph);
  if (!bounces && h >= 16 * kFrameChunks) {
    // A frame without bounce levels (what render() of the reference asks for) is
    // finished row by row: it is rendered as kFrameChunks launches over
    // contiguous blocks of rows, and the copy back of a block runs, on a second
    // stream, while the next blocks render.  The copy takes longer than the
    // render (pageable memory: 1.9 ms against 1.0 ms for 3840 x 2160), so the
    // call then lasts about as long as the copy alone.
    if (!C.stream)
      HIP_TRY(hipStreamCreateWithFlags(&C.stream, hipStreamNonBlocking));
    if (!C.copy_stream)
      HIP_TRY(hipStreamCreateWithFlags(&C.copy_stream, hipStreamNonBlocking));
    for (hipEvent_t &e : C.ev_chunk)
      if (!e)
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const int band_rows = ((h + kFrameChunks - 1) / kFrameChunks + 7) & ~7;
    const int chunks = (h + band_rows - 1) / band_rows; // <= kFrameChunks; chunk k = band k
    for (int k = 0; k < chunks; k++) {
      if ((rc = cached_frame(C, cam, hit_ids != nullptr, w, h, max_depth, flags, band_rows, k,
                             chunks, C.stream, k == 0)))
        return rc;
      HIP_TRY(hipEventRecord(C.ev_chunk[k], C.stream));
    }
    for (int k = 0; k < chunks; k++) {
      const size_t r0 = (size_t)k * band_rows, r1 = std::min((size_t)h, r0 + band_rows);
      const size_t off = r0 * w, cnt = (r1 - r0) * w;
      HIP_TRY(hipStreamWaitEvent(C.copy_stream, C.ev_chunk[k], 0));
      HIP_TRY(hipMemcpyAsync(rgb + 4 * off, C.fb.p + 4 * off, 4 * cnt, hipMemcpyDeviceToHost,
                             C.copy_stream));
      if (hit_ids)
        HIP_TRY(hipMemcpyAsync(hit_ids + off, C.ids.p + off, cnt * sizeof(int32_t),
                               hipMemcpyDeviceToHost, C.copy_stream));
    }
    HIP_TRY(hipStreamSynchronize(C.copy_stream));
    HIP_TRY(hipStreamSynchronize(C.stream));
    if (n_rays)
      HIP_TRY(hipMemcpy(n_rays, C.cnt.p, sizeof(uint64_t), hipMemcpyDeviceToHost));
    return RT_OK;
  }
  // render; if bounce rays were dropped or the machinery does not fit, again
  // another way (next_attempt)
  for (;;) {
    rc = cached_frame(C, cam, hit_ids != nullptr, w, h, max_depth, flags, 8, 0, 1, nullptr);
    if (rc == RT_ERR_CAPACITY && next_attempt(S))
      continue;
    if (rc)
      return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));
    int over = 0;
    if ((rc = rt_scene_queue_overflowed(S, &over)))
      return rc;
    if (!over)
      break;
    if (!next_attempt(S))
      return fail(RT_ERR_CAPACITY, "bounce-ray queue overflowed; image is incomplete");
  }
  // (a buffer from rt_alloc_host_buffer takes this copy at the full rate of the
  // link; ordinary pageable memory goes through the runtime's staging)
  HIP_TRY(hipMemcpy(rgb, C.fb.p, npx * 4, hipMemcpyDeviceToHost));
  if (hit_ids)
    HIP_TRY(hipMemcpy(hit_ids, C.ids.p, npx * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (n_rays)
    HIP_TRY(hipMemcpy(n_rays, C.cnt.p, sizeof(uint64_t), hipMemcpyDeviceToHost));
  return RT_OK;
}

// gpu_render_multi, rehearsal only: RT_AMD_LOGICAL_DEVICES=n lets the call take up
// to n "devices" on a machine with fewer GPUs; slot d then lives on GPU d modulo
// the GPUs present, with a kept scene, frame buffers and stream of its own, and
// the band copies, the overflow retry and the ray-count sum run as they would
// between n GPUs.  Unset (the default), a slot is a GPU.
int logical_devices() {
  const char *e = getenv("RT_AMD_LOGICAL_DEVICES");
  if (!e)
    return 0;
  const int n = atoi(e);
  return n > 0 ? n : 0;
}

} // namespace

extern "C" {

void rt_cache_clear(void) {
  for (int d = 0; d < kMaxCachedDevices; d++) {
    std::lock_guard<std::mutex> lock(g_cache[d].mutex);
    g_cache[d].drop();
  }
}

// Scenes the one-call entry points have built so far (diagnostic, for tests of
// the residency rule; not part of the documented interface).
uint64_t rt_cache_builds(void) { return g_cache_builds.load(); }

void rt_cache_trust_pointers(int on) { g_trust_pointers.store(on != 0); }

int rt_alloc_host_buffer(size_t bytes, void **out) {
  if (!out || bytes == 0)
    return fail(RT_ERR_BAD_ARG, "rt_alloc_host_buffer: null result pointer or zero size");
  *out = nullptr;
  int rc = ensure_device();
  if (rc)
    return rc;
  void *p = nullptr;
  const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(RT_ERR_HIP, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  }
  *out = p;
  return RT_OK;
}

void rt_free_host_buffer(void *p) {
  if (p)
    (void)hipHostFree(p);
}

int gpu_render_ids(const scene_desc *scene, const camera *cam, uint8_t *rgb,
                   int32_t *hit_ids, uint64_t *n_rays, int w, int h,
                   int max_depth) {
  return render_cached(scene, cam, rgb, hit_ids, n_rays, w, h, max_depth, 0);
}

int gpu_render(const scene_desc *scene, const camera *cam, uint8_t *rgb, int w,
               int h, int max_depth) {
  return render_cached(scene, cam, rgb, nullptr, nullptr, w, h, max_depth, 0);
}

int gpu_render_unlit(const scene_desc *scene, const camera *cam, uint8_t *rgb, int w, int h) {
  return render_cached(scene, cam, rgb, nullptr, nullptr, w, h, 1, RT_FRAME_UNLIT);
}

int gpu_render_multi(const scene_desc *scene, const camera *cam, uint8_t *rgb,
                     int32_t *hit_ids, uint64_t *n_rays, int w, int h, int max_depth,
                     int n_devices, int band_rows) {
  if (!cam || !rgb)
    return fail(RT_ERR_BAD_ARG, "camera or output buffer is NULL");
  if (w <= 0 || h <= 0)
    return fail(RT_ERR_BAD_ARG, "frame size %dx%d", w, h);
  if (band_rows <= 0 || (band_rows & 7))
    return fail(RT_ERR_BAD_ARG, "band_rows %d must be a positive multiple of 8", band_rows);
  int rc = check_scene_desc(scene);
  if (rc)
    return rc;
  int visible = 0;
  if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0)
    return fail(RT_ERR_NO_DEVICE, "no HIP device");
  const int slots = std::min(std::max(visible, logical_devices()), kMaxCachedDevices);
  if (n_devices < 1 || n_devices > slots)
    return fail(RT_ERR_BAD_ARG, "n_devices %d outside 1..%d", n_devices, slots);
  auto gpu_of = [&](int d) { return d % visible; };
  int home = 0;
  HIP_TRY(hipGetDevice(&home));
  const SceneKey key = scene_key(scene);
  // the slots are taken in rising order by every caller, so two calls cannot wait
  // for each other
  std::unique_lock<std::mutex> locks[kMaxCachedDevices];
  for (int d = 0; d < n_devices; d++)
    locks[d] = std::unique_lock<std::mutex>(g_cache[d].mutex);
  const size_t npx = (size_t)w * h;
  const int bands_total = (h + band_rows - 1) / band_rows;
  CachedScene *C[kMaxCachedDevices] = {};
  auto body = [&]() -> int {
    // the scene on every device (the builds of different devices run one after
    // the other on this thread; kept scenes cost nothing)
    for (int d = 0; d < n_devices; d++) {
      HIP_TRY(hipSetDevice(gpu_of(d)));
      C[d] = &g_cache[d];
      int r = cached_scene(*C[d], scene, key);
      if (r)
        return r;
      if (!C[d]->stream)
        HIP_TRY(hipStreamCreateWithFlags(&C[d]->stream, hipStreamNonBlocking));
      if (gpu_of(d) != gpu_of(0)) {
        int can = 0;
        HIP_TRY(hipDeviceCanAccessPeer(&can, gpu_of(0), gpu_of(d)));
        if (can) { // device 0 reads its peers' bands directly over xGMI
          hipError_t e = hipSetDevice(gpu_of(0)) == hipSuccess
                             ? hipDeviceEnablePeerAccess(gpu_of(d), 0)
                             : hipErrorInvalidDevice;
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
            return fail(RT_ERR_HIP, "hipDeviceEnablePeerAccess(%d): %s", gpu_of(d),
                        hipGetErrorString(e));
          (void)hipGetLastError();
        }
      }
    }
    for (;;) {
      // every device renders its bands on its own stream
      bool short_of_memory = false;
      for (int d = 0; d < n_devices; d++) {
        HIP_TRY(hipSetDevice(gpu_of(d)));
        int r = cached_frame(*C[d], cam, hit_ids != nullptr, w, h, max_depth, 0, band_rows, d,
                             n_devices, C[d]->stream);
        if (r == RT_ERR_CAPACITY)
          short_of_memory = true;
        else if (r)
          return r;
      }
      // the bands travel to device 0: one peer copy per band, on the owner's
      // stream (behind its render), so the copies of different devices overlap
      for (int d = 1; d < n_devices && !short_of_memory; d++) {
        HIP_TRY(hipSetDevice(gpu_of(d)));
        for (int b = d; b < bands_total; b += n_devices) {
          const size_t r0 = (size_t)b * band_rows;
          const size_t r1 = std::min((size_t)h, r0 + band_rows);
          const size_t off = r0 * w, cnt = (r1 - r0) * w;
          HIP_TRY(hipMemcpyPeerAsync(C[0]->fb.p + 4 * off, gpu_of(0), C[d]->fb.p + 4 * off,
                                     gpu_of(d), 4 * cnt, C[d]->stream));
          if (hit_ids)
            HIP_TRY(hipMemcpyPeerAsync(C[0]->ids.p + off, gpu_of(0), C[d]->ids.p + off, gpu_of(d),
                                       sizeof(int32_t) * cnt, C[d]->stream));
        }
      }
      bool over = short_of_memory;
      for (int d = 0; d < n_devices; d++) {
        HIP_TRY(hipSetDevice(gpu_of(d)));
        HIP_TRY(hipStreamSynchronize(C[d]->stream));
        int o = 0;
        int r = rt_scene_queue_overflowed(C[d]->scene, &o);
        if (r)
          return r;
        over = over || o;
      }
      if (!over)
        break;
      // every device renders the frame again the same other way
      bool more = true;
      for (int d = 0; d < n_devices; d++)
        more = next_attempt(C[d]->scene) && more;
      if (!more)
        return fail(RT_ERR_CAPACITY, "bounce-ray queue overflowed; image is incomplete");
    }
    HIP_TRY(hipSetDevice(gpu_of(0)));
    HIP_TRY(hipMemcpy(rgb, C[0]->fb.p, npx * 4, hipMemcpyDeviceToHost));
    if (hit_ids)
      HIP_TRY(hipMemcpy(hit_ids, C[0]->ids.p, npx * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (n_rays) {
      *n_rays = 0;
      for (int d = 0; d < n_devices; d++) {
        uint64_t v = 0;
        HIP_TRY(hipSetDevice(gpu_of(d)));
        HIP_TRY(hipMemcpy(&v, C[d]->cnt.p, sizeof v, hipMemcpyDeviceToHost));
        *n_rays += v;
      }
    }
    return RT_OK;
  };
  rc = body();
  (void)hipSetDevice(home);
  return rc;
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
  const size_t un = (size_t)n;
  rt::CheckArgs a;
  a.n = (long)n;
  if (un <= kCheckPinnedRays) {
    // A few rays (the reference's callers ask for one per call): the kernel reads
    // the rays from, and writes the answers to, a small block of pinned host
    // memory that the scene keeps mapped into the device, so the call is a launch
    // and a wait; five small copies cost several times that.
    if (!S->chk_pin) {
      HIP_TRY(hipHostMalloc((void **)&S->chk_pin, kCheckPinnedRays * 11 * sizeof(float),
                            hipHostMallocMapped));
      HIP_TRY(hipHostGetDevicePointer((void **)&S->chk_pin_dev, S->chk_pin, 0));
    }
    float *h = S->chk_pin, *d = S->chk_pin_dev;
    memcpy(h, orig, 3 * un * sizeof(float));
    memcpy(h + 3 * un, dir, 3 * un * sizeof(float));
    a.orig = d;
    a.dir = d + 3 * un;
    a.P = d + 6 * un;
    a.kind = (int32_t *)(d + 9 * un);
    a.index = (int32_t *)(d + 10 * un);
    HIP_TRY(rt::launch_check(S->dev(), a, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    memcpy(P, h + 6 * un, 3 * un * sizeof(float));
    memcpy(kind, h + 9 * un, un * sizeof(int32_t));
    memcpy(index, h + 10 * un, un * sizeof(int32_t));
    return RT_OK;
  }
  // staging kept by the scene (grow-only): origins then directions; kinds then
  // indices
  if (S->chk_in.n < 6 * un) {
    HIP_TRY(S->chk_in.alloc(6 * un));
    HIP_TRY(S->chk_P.alloc(3 * un));
    HIP_TRY(S->chk_out.alloc(2 * un));
  }
  HIP_TRY(hipMemcpy(S->chk_in.p, orig, 3 * un * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(S->chk_in.p + 3 * un, dir, 3 * un * sizeof(float), hipMemcpyHostToDevice));
  a.orig = S->chk_in.p;
  a.dir = S->chk_in.p + 3 * un;
  a.kind = S->chk_out.p;
  a.index = S->chk_out.p + un;
  a.P = S->chk_P.p;
  HIP_TRY(rt::launch_check(S->dev(), a, nullptr));
  HIP_TRY(hipMemcpy(kind, a.kind, un * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(index, a.index, un * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(P, a.P, 3 * un * sizeof(float), hipMemcpyDeviceToHost));
  return RT_OK;
}

// renderer.cpp:106-110 for n (lang, colour) pairs: host arrays in, host arrays out.
int gpu_specular_terms(int64_t n, const float *lang, const uint8_t *color, float *strength,
                       uint8_t *specular) {
  if (n < 0)
    return fail(RT_ERR_BAD_ARG, "n < 0");
  if (n == 0)
    return RT_OK;
  if (!lang || !color || !strength || !specular)
    return fail(RT_ERR_BAD_ARG, "NULL input or result array");
  int rc = ensure_device();
  if (rc)
    return rc;
  const size_t un = (size_t)n;
  DevBuf<float> d_lang, d_str;
  DevBuf<uint8_t> d_col, d_spec;
  HIP_TRY(d_lang.alloc(un));
  HIP_TRY(d_str.alloc(un));
  HIP_TRY(d_col.alloc(un));
  HIP_TRY(d_spec.alloc(un));
  HIP_TRY(hipMemcpy(d_lang.p, lang, un * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_col.p, color, un, hipMemcpyHostToDevice));
  HIP_TRY(rt::launch_specular((long)n, d_lang.p, d_col.p, d_str.p, d_spec.p, nullptr));
  HIP_TRY(hipMemcpy(strength, d_str.p, un * sizeof(float), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(specular, d_spec.p, un, hipMemcpyDeviceToHost));
  return RT_OK;
}

// One ray through the scene that gpu_render* would keep for these arrays
// (renderer.hpp:21-23 as the reference's own callers use it: the same arrays,
// one ray per call).  The first call uploads; later calls with unchanged
// geometry only hash it (rt_cache_trust_pointers: not even that) and launch.
// A scene that render() left resident serves these calls as it is, and the other
// way round a render() after them rebuilds once (it needs colours and lights).
int gpu_check_intersection_arrays(const scene_desc *scene, const float *orig, const float *dir,
                                  int32_t *kind, int32_t *index, float *P) {
  int rc = check_scene_desc(scene, false);
  if (rc || (rc = ensure_device()))
    return rc;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxCachedDevices)
    return fail(RT_ERR_UNSUPPORTED, "device %d: only devices 0..%d keep a scene", dev,
                kMaxCachedDevices - 1);
  const SceneKey key = scene_key(scene, true);
  CachedScene &C = g_cache[dev];
  std::lock_guard<std::mutex> lock(C.mutex);
  if ((rc = cached_scene(C, scene, key, false)))
    return rc;
  return gpu_check_intersection(C.scene, 1, orig, dir, kind, index, P);
}

} // extern "C"
