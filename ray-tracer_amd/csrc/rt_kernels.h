// rt_kernels.h — argument blocks and host launchers of the kernels in
// rt_kernels.hip.  Internal to the library (not part of the C ABI).
#pragma once
#include "rt_device.h"

namespace rt {

struct PrepArgs {
  // inputs: the reference's flat arrays, already in device memory
  const float *tris;
  const uint8_t *color_tri;
  const float *spheres;
  const float *radius;
  const uint8_t *color_sphere;
  int t_size, s_size, s_pad;
  // outputs
  float4 *sph_geo;
  TriRec *tri;
  uchar4 *col_tri;
  uchar4 *col_sph;
};

struct EyeArgs {
  const float4 *sph_geo;
  float4 *sph_prim;
  TriRec *tri;
  int t_size, s_size, s_pad;
  float ox, oy, oz;
};

// Rays are counted in RAY_PARTS partial counters, RAY_PART_STRIDE words apart (one
// per 64-byte line), that k_sum_rays adds to the caller's word at the end of a
// frame (rt_kernels.hip, add_ray_count).  Every `ray_count` below points to them.
constexpr uint32_t RAY_PARTS = 256;
constexpr uint32_t RAY_PART_STRIDE = 8;

struct FrameDev {
  uint8_t *fb;   // 4 bytes per pixel, whole frame
  int32_t *ids;  // or nullptr
  uint64_t *ray_count; // the partial counters, or nullptr
  int w, h;
  int max_depth;
  int band_rows, band_index, band_count;
  int unlit; // 1: the reference's -DUNLIT=1 build (renderer.cpp:130-137)
};

// Bounce-ray queue in HBM: two float4 per ray,
//   (origin.xyz, weight) and (dir.xyz, bits of (pixel << 6 | depth)).
// The queue is cut into n_seg equal segments, each with its own length counter
// 256 bytes from the next: every wave of a block appends to (and later consumes
// from) segment  block % n_seg.  One shared counter hands out only ~90 M slots/s
// on this chip (returning atomics on one address serialise at ~11 ns); spread
// over 64+ addresses the same atomics run at > 4 G/s (tools/atomic_rate.hip).
constexpr int QUEUE_SEGMENTS = 128;
constexpr int CTL_WORDS = 64; // uint32 words between two segments' counters

struct RayQueue {
  float4 *items;      // segment s holds items [s * seg_capacity, (s+1) * seg_capacity)
  uint32_t *count;    // count[s * CTL_WORDS]: rays appended to segment s (may exceed
                      // seg_capacity: the excess is dropped and `overflow` is set)
  uint32_t seg_capacity;
  uint32_t *overflow; // set to 1 when a ray had to be dropped
};

// Nearest sphere hit of every ray of a level, at the ray's own queue slot.
struct HitList {
  float4 *rec;  // P.xyz, |P|
  int32_t *idx; // sphere index, -1 = no sphere hit
};

struct BounceDev {
  RayQueue in;     // rays to trace at this level (unused by k_render)
  RayQueue out;    // rays spawned for the next level
  unsigned long long *extra; // per pixel: integer light of bounce hits, 3 x 21 bits
  int max_depth;   // 1 = no bounces (all other fields may be null)
};

// Bounce levels as separate launches (rt_walk.h).
// Refill threshold: 4..48 measured on the 10k-sphere depth-8 frame, 11.98 ms at 4,
// 10.72 at 16, 10.53 at 24, 10.66 at 48.
constexpr int WALK_FETCH_AT = 24;
struct WalkArgs {
  RayQueue in;        // items to walk: (origin, tagA bits), (direction, tagB bits)
  uint32_t *cursor;   // cursor[s * CTL_WORDS]: next item of segment s to hand out
  HitList hits;       // nearest-hit walks: where hits go
  HitList prior;      // triangle nearest-hit walks: the rays' sphere hits (walked first), or null
  unsigned long long *extra; // shadow walks: tagA = pixel, tagB = packed integer
                             // light, added to the pixel if unoccluded
  uint64_t *ray_count;
  int last_stage;     // shadow walks: 1 = add the light if unoccluded, 0 = cancel occluded items
  int fetch_at;       // refill a wave when this many of its lanes have nothing to do
};

// All bounce levels in one launch (rt_trace.h).
struct TraceArgs {
  RayQueue in;        // first-bounce rays of the frame (written by k_render)
  uint32_t *cursor;   // cursor[s * CTL_WORDS]: next item of segment s to hand out
  float4 *stack;      // per wave: stack_cap work items of two float4
  float4 *hits;       // per wave: hit_cap hit records of three float4
  float4 *save;       // per wave: 64 lanes x TRACE_SAVE_WORDS words of walk state
  uint32_t stack_cap, hit_cap;
  unsigned long long *extra; // per-pixel integer light sums
  uint64_t *ray_count;
  uint32_t *overflow; // set to 1 when a wave's stack was too small
  int max_depth;
  int fetch_at;       // refill a wave when this many of its lanes have nothing to do
};
// Refill threshold of k_trace.  With nine parking rows 24 and 40 were level (4.94 and
// 4.97 ms on the bench frame then); with twelve (rt_trace.h: the rows a 7-level tree's
// stack leaves free) a lane holds more touched leaves before it forces a work-off, and
// longer stretches of visits between service phases pay: 16 / 24 / 32 / 40 / 48 / 56
// idle lanes give 3.96 / 3.84 / 3.80 / 3.75 / 3.76 / 3.84 ms.
constexpr int TRACE_FETCH_AT = 40;
constexpr uint32_t TRACE_SAVE_WORDS = 20;
constexpr uint32_t TRACE_HIT_CAP = 128; // < 64 waiting + <= 64 handed in at once

struct ShadeArgs {
  RayQueue rays;     // the level's rays
  HitList hits;      // their nearest sphere hits (k_walk)
  HitList tri_hits;  // their nearest triangle hits (k_walk over the triangle hierarchy)
  int light_begin, light_end; // lights evaluated in this launch
  int spawn;         // also queue the next level's rays
  int max_depth;
  RayQueue shadow;   // shadow-ray items that no triangle blocks
  RayQueue next;     // next level's rays
  uint64_t *ray_count;
};

struct CheckArgs {
  long n;
  const float *orig;
  const float *dir;
  int32_t *kind;
  int32_t *index;
  float *P;
};

hipError_t launch_prep_scene(const PrepArgs &a, hipStream_t st);
hipError_t launch_prep_eye(const EyeArgs &a, hipStream_t st);
// Lists the spheres by screen cell of BIN_PX x BIN_PX pixels for the rays from the eye
// of a w x h frame: bins holds ceil(w / BIN_PX) * ceil(h / BIN_PX) records of BIN_WORDS
// words (rt_device.h, SceneDev::bins).
// `super`: ceil(w / BIN_SUPER_PX) * ceil(h / BIN_SUPER_PX) records of BIN_SUPER_WORDS words
// for the first of the two levels.
hipError_t launch_bin_spheres(const SceneDev &sc, const CameraDev &cam, int w, int h, int32_t *bins,
                              int32_t *super, hipStream_t st);
// Lists the spheres by direction from the light X for shadow rays that start within
// `reach` of it: cells holds LCELL_COUNT records of LCELL_WORDS words (rt_device.h,
// SceneDev::lcells), super 6 * LCELL_SUPER_R^2 records of LCELL_SUPER_WORDS words for
// the first of the two levels.
hipError_t launch_bin_light(const SceneDev &sc, float Xx, float Xy, float Xz, float reach,
                            int32_t *cells, int32_t *super, hipStream_t st);
// Diagnostic: every sphere against every pixel ray and primary shadow ray of a w x h
// frame; out (device, 4 x uint64, zeroed by the caller) as described at k_verify_lists.
hipError_t launch_verify_lists(const SceneDev &sc, const CameraDev &cam, int w, int h,
                               unsigned long long *out, hipStream_t st);
// accel: RT_BEAM_NONE (plain loop), RT_BEAM_APEX (screen spheres per wave
// before the per-ray tests) or RT_ACCEL_BVH (every lane walks the sphere
// hierarchy); same results, different amounts of work.  verify_stats: device uint64[2] {spheres dropped although a ray's
// discriminant test passes (must stay 0), spheres kept}; selects a slow
// self-checking build of the kernel.
hipError_t launch_render(const SceneDev &sc, const CameraDev &cam,
                         const FrameDev &f, const BounceDev &bd,
                         int owned_tile_rows, int accel,
                         unsigned long long *verify_stats, hipStream_t st);
// verify_stats (device uint64[16] or null): [0..1] as launch_render, [2..4] rays /
// node bounds / leaf spheres of nearest-hit walks, [5..7] the same for shadow walks,
// [8..13] the same six numbers for the triangle hierarchy.
hipError_t launch_bounce(const SceneDev &sc, const BounceDev &bd, bool beam,
                         uint64_t *ray_count, unsigned long long *verify_stats,
                         int blocks, hipStream_t st);
hipError_t launch_walk(const SceneDev &sc, const WalkArgs &w, bool any_hit, bool triangles,
                       int blocks, hipStream_t st);
// bvh_depth: levels of inner nodes of the scene's sphere tree (<= LDS_WALK_MAX_DEPTH)
hipError_t launch_trace(const SceneDev &sc, const TraceArgs &t, int bvh_depth, int blocks,
                        hipStream_t st);
// Blocks of k_trace that can be resident at once on the current device (what a
// launch should use: a block that waits for a free slot finds nothing left to do).
int trace_resident_blocks();
// The same for k_walk, as a multiple of QUEUE_SEGMENTS (a walking wave takes items
// from every segment, so more blocks than fit at once only shorten each wave's share).
int walk_resident_blocks();
hipError_t launch_shade(const SceneDev &sc, const ShadeArgs &s, int blocks, hipStream_t st);
// over the pixels of the bands the frame owns (band b belongs to share b % band_count)
// parts / total (or null): the first block also does what launch_sum_rays does
hipError_t launch_compose(uint8_t *fb, unsigned long long *extra, int w, int h, int band_rows,
                          int band_index, int band_count, uint64_t *parts, uint64_t *total,
                          hipStream_t st);
hipError_t launch_sum_rays(uint64_t *parts, uint64_t *total, hipStream_t st);
hipError_t launch_gather_leaves(const float4 *sph_geo, const uchar4 *col_sph,
                                const float4 *mat_sph, const int32_t *leaf_idx, float4 *leaf_geo,
                                float4 *leaf_mat, int n, hipStream_t st);
hipError_t launch_gather_tri_leaves(const TriRec *tri, const int32_t *leaf_idx, TriLeaf *leaf,
                                    int n, hipStream_t st);
hipError_t launch_check(const SceneDev &sc, const CheckArgs &a, hipStream_t st);
// renderer.cpp:106-110 on n caller-supplied (lang, colour) pairs, device arrays
hipError_t launch_specular(long n, const float *lang, const uint8_t *color, float *strength,
                           uint8_t *specular, hipStream_t st);

} // namespace rt
