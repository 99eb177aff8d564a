/*
 * rt_amd.h — C ABI of the MI355X trace/shade path.
 *
 * This library replaces exactly one thing in matheusmortatti/Ray-Tracer: the
 * body of render() (renderer.cpp:7-141) together with check_intersection()
 * (renderer.cpp:145-193) and the math it calls (maths.cpp:5-168).  Everything
 * the reference keeps on the host (scene file parsing, the sphere-grid
 * generator, the viewer, the PPM dump) calls in through the functions below.
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no C++/torch types cross the boundary
 *   - every function returns RT_OK (0) or a negative rt_status;
 *     rt_last_error() gives the text of the last failure on this thread
 *   - "host" pointers are ordinary process memory, "dev" pointers are HIP
 *     device memory on the device the rt_scene was created on
 *   - framebuffer layout is the reference's (renderer.cpp:47-53): row-major,
 *     top row first, 4 bytes per pixel; bytes 0..2 are the accumulated colour
 *     in scene-file channel order, byte 3 is 255
 *   - hit-id buffer: one int32 per pixel; -1 = miss, [0,t_size) = triangle
 *     index, t_size + [0,s_size) = sphere index, i.e. the position of the
 *     primitive in the order check_intersection tests them
 *     (renderer.cpp:156,168)
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rt_status {
  RT_OK = 0,
  RT_ERR_BAD_ARG = -1,      /* null pointer, negative size, w/h <= 0 ... */
  RT_ERR_NO_DEVICE = -2,    /* no HIP device visible */
  RT_ERR_HIP = -3,          /* a HIP runtime call failed */
  RT_ERR_CAPACITY = -4,     /* a fixed-size buffer is too small for the request */
  RT_ERR_UNSUPPORTED = -5
} rt_status;

/*
 * Per-primitive material.  NOT in the reference: there a primitive carries
 * only an 8-bit colour (main.cpp:287-289, 297-299) and every surface is
 * Phong with KS 0.3 / KD 0.7 / exponent 20 (renderer.cpp:3-5).  Materials are
 * read only when max_depth > 1; a scene whose material pointers are NULL, or
 * whose materials are all zero, renders exactly as the reference does at any
 * max_depth, and max_depth = 1 renders exactly as the reference whatever the
 * materials are.
 */
typedef struct rt_material {
  float reflectivity; /* weight of the mirror bounce, 0..1 */
  float transparency; /* weight of the refracted bounce, 0..1 */
  float ior;          /* index of refraction (> 0) for the refracted bounce */
  float reserved;
} rt_material;

/*
 * The eleven scene arguments of render() (renderer.hpp:16-18), as one struct.
 * Layouts are the reference's flat arrays (built at main.cpp:270-305 and
 * main.cpp:377-392).  The library only reads these.
 */
typedef struct scene_desc {
  const float *tris;           /* 9*t_size : p1.xyz p2.xyz p3.xyz per triangle */
  const uint8_t *color_tri;    /* 3*t_size */
  int32_t t_size;
  const float *spheres;        /* 3*s_size : centre.xyz */
  const float *radius;         /* s_size */
  const uint8_t *color_sphere; /* 3*s_size */
  int32_t s_size;
  const float *lights;         /* 3*l_size : position.xyz */
  int32_t l_size;
  const rt_material *mat_tri;    /* t_size entries or NULL (extension) */
  const rt_material *mat_sphere; /* s_size entries or NULL (extension) */
} scene_desc;

/*
 * The reference has no camera object: eye (0,0,1) (renderer.cpp:39), image
 * plane z=-1 (renderer.cpp:38), fov an int in degrees (main.cpp:88).
 * rt_camera_default() fills exactly those values.
 */
typedef struct camera {
  int32_t fov_deg;
  float origin[3];
  float plane_z;
} camera;

void rt_camera_default(camera *cam);

/* ---- one-call entry point ------------------------------------------------
 * Replaces render() (renderer.hpp:16-18; called at main.cpp:104-106).
 * Blocking.  Uploads the scene, renders w x h pixels on the current HIP
 * device and copies the 4-byte/pixel framebuffer into host memory `rgb`
 * (4*w*h bytes).  max_depth = 1 is the reference's behaviour; larger values
 * enable the material bounces described at rt_material.                      */
int gpu_render(const scene_desc *scene, const camera *cam, uint8_t *rgb, int w,
               int h, int max_depth);

/* As gpu_render, and also returns the per-pixel hit-id buffer (host, w*h
 * int32, may be NULL) and the number of rays traced (may be NULL).  The id
 * buffer is what calling check_intersection (renderer.hpp:21-23) with each
 * pixel's primary ray yields.                                                */
int gpu_render_ids(const scene_desc *scene, const camera *cam, uint8_t *rgb,
                   int32_t *hit_ids, uint64_t *n_rays, int w, int h,
                   int max_depth);

/* As gpu_render for the reference built with -DUNLIT=1 (renderer.cpp:78,
 * 130-137): flat colour of the hit primitive, no lights.                     */
int gpu_render_unlit(const scene_desc *scene, const camera *cam, uint8_t *rgb,
                     int w, int h);

/* gpu_render, gpu_render_ids, gpu_render_unlit and gpu_render_multi keep the
 * device copy of the last scene they were given (per device): a later call with
 * the same array pointers, sizes and contents (all bytes are hashed) skips the
 * upload and the hierarchy builds; anything else replaces the kept scene.
 * rt_cache_clear releases it.  The `map(to:)` of renderer.cpp:13-21 happens once
 * per distinct scene, not once per frame.                                    */
void rt_cache_clear(void);
/* How many scenes those entry points have built so far in this process
 * (diagnostic: lets a host or a test see that a call reused the kept scene).  */
uint64_t rt_cache_builds(void);
/* The content hashes above (64-bit, not cryptographic: they notice arrays that
 * were edited in place between calls, they do not resist an adversary) cost a
 * pass over every array per call: 36 MB for a million triangles.  A host that
 * never edits a scene's arrays in place (the reference's main.cpp builds them
 * once, main.cpp:98-101, and only reads them afterwards) may say so: with
 * on != 0 the same pointers and sizes ARE the same scene and nothing is hashed.
 * Process-wide; no counterpart in the reference.                              */
void rt_cache_trust_pointers(int on);

/* The one-call entry points end with a device-to-host copy of the frame into
 * `rgb` (and `hit_ids`).  Into ordinary pageable memory (new[], malloc: what the
 * reference allocates at main.cpp:91) that copy goes through the runtime's
 * staging buffers, 1.9 ms for a 3840 x 2160 frame, twice the render of a depth-1
 * frame; into page-locked memory it runs at the rate of the PCIe link, 0.6 ms.
 * rt_alloc_host_buffer returns such memory (hipHostMalloc) for a host that is
 * willing to allocate its frame buffer through the library, one line changed in
 * the reference's main.cpp; rt_free_host_buffer releases it.  (Pinning a buffer the
 * host already has, hipHostRegister, is refused by the driver stack this library
 * was developed on, so no entry point offers it.)  Either kind of buffer is
 * accepted everywhere; nothing else changes.  No counterpart in the reference,
 * whose frame buffer is written by host threads.                               */
int rt_alloc_host_buffer(size_t bytes, void **out);
void rt_free_host_buffer(void *p);

/* ---- resident scene + device-side rendering -------------------------------
 * The `map(to: ...)` clause of renderer.cpp:13-21 as an explicit object: the
 * flat arrays are uploaded once, re-laid-out for the kernels, and stay in HBM
 * until rt_scene_destroy.                                                    */
typedef struct rt_scene rt_scene;

int rt_scene_create(const scene_desc *scene, rt_scene **out);
void rt_scene_destroy(rt_scene *scene);

typedef struct rt_frame {
  uint8_t *dev_fb;      /* device, 4*w*h bytes, whole frame; required */
  int32_t *dev_ids;     /* device, w*h int32, or NULL */
  int32_t w, h;
  int32_t max_depth;    /* 1..63 */
  /* Row-band sharding (renderer.cpp:27-28 parallelises rows; here bands of
   * band_rows rows are dealt round-robin): this call renders bands b with
   * b % band_count == band_index.  band_count = 1 renders the whole frame.   */
  int32_t band_rows;    /* rows per band, > 0 (ignored when band_count = 1) */
  int32_t band_index;
  int32_t band_count;
  int32_t flags;        /* 0 or RT_FRAME_UNLIT */
  void *stream;         /* hipStream_t or NULL for the default stream */
} rt_frame;

/* rt_frame.flags: the reference built with -DUNLIT=1 (renderer.cpp:78,130-137;
 * CMakeLists.txt:36): a hit pixel gets its primitive's colour, no lights, no
 * shadow rays, no bounces (max_depth is ignored).                             */
#define RT_FRAME_UNLIT 1

/* Enqueues the render of `frame` on frame->stream and returns without
 * waiting.  Pixels outside this call's bands are not touched.
 * dev_ray_count (device uint64, or NULL) is incremented by the rays traced.  */
int rt_render_device(rt_scene *scene, const camera *cam, const rt_frame *frame,
                     uint64_t *dev_ray_count);

/* Bounce rays wait in a queue in HBM sized at 4 rays per pixel of the call (see
 * rt_scene_set_rays_per_pixel).  If a frame spawns more, the excess is dropped
 * and this reports 1 after the frame (blocking read): the image is incomplete.
 * gpu_render / gpu_render_ids check it themselves and render again: first with
 * a launch per bounce level instead of one launch for all (rt_scene_set_one_launch
 * 0: the one-launch form keeps a stack per resident wave, which can overflow
 * whatever the queue setting and whose size grows with it five thousand-fold),
 * then with eight times the queue capacity per attempt; they return
 * RT_ERR_CAPACITY only if the largest queue overflows too, which the weight
 * cut-off makes impossible.  A caller of rt_render_device has the same two
 * remedies: rt_scene_set_one_launch(scene, 0) and rt_scene_set_rays_per_pixel. */
int rt_scene_queue_overflowed(rt_scene *scene, int *overflowed);

/* Capacity of the bounce-ray queues of later rt_render_device calls: n rays per
 * owned pixel and bounce level (1..256, default 4; + 65536).  256 can never
 * overflow: every queued ray carries a weight of at least 2^-8 and the weights
 * of a pixel's rays at one level sum to at most 1.  A caller that finds
 * rt_scene_queue_overflowed set raises it and renders the frame again;
 * gpu_render / gpu_render_ids do exactly that by themselves (x8 per attempt).
 * No counterpart in the reference.                                            */
int rt_scene_set_rays_per_pixel(rt_scene *scene, int n);

/* Upper limit, in bytes, of the device memory a scene may take for the bounce
 * machinery of its frames (ray queues, per-wave stacks, per-pixel sums; it grows
 * with the frame size, max_depth, the number of lights and the rays-per-pixel
 * setting).  0 = no limit (default).  A frame that would need more fails with
 * RT_ERR_CAPACITY before anything is allocated, exactly as when the device is
 * out of memory; in both cases the scene is left without any machinery and the
 * next frame starts afresh.  gpu_render* react to RT_ERR_CAPACITY by rendering
 * the frame with a launch per bounce level, whose needs are smaller, before they
 * give up.  No counterpart in the reference.                                  */
int rt_scene_set_arena_limit(rt_scene *scene, uint64_t max_bytes);

/* ---- work avoidance (no counterpart in the reference) ----------------------
 * The reference tests every ray against every primitive (renderer.cpp:156-177).
 * The library gets the same answers with far fewer tests, in four ways:
 *   - beam screen: a wave's rays that share an apex (primary rays: the eye;
 *     shadow rays: the light) are bounded by a cone, the spheres are screened
 *     64 at a time (one per lane) against that cone, and the reference's per-ray
 *     arithmetic runs only on spheres the cone can touch;
 *   - sphere hierarchy: a tree of boxes built at rt_scene_create; each
 *     lane walks it for its own ray (bounce rays, bulk queries, and in mode 3
 *     also primary and shadow rays);
 *   - sphere lists (scenes with 64 spheres or more, mode 1): per cell of 16 x 16
 *     pixels the spheres that the cone of the cell's rays from the eye can touch,
 *     made anew when the camera or the frame size changes (two small kernels); and
 *     per direction from each of the first four lights (a cube of 6 x 128 x 128
 *     cells, 12.6 MB per light, made at rt_scene_create) the spheres that a shadow
 *     ray towards that light can touch.  Both by the beam screen's test.  The rays
 *     of a tile then run the reference's loop over a list of a handful of spheres
 *     instead of walking the tree;
 *   - triangle hierarchy: scenes with 16 or more triangles get a tree over the
 *     triangles as well, with bounds that follow the reference's own triangle
 *     arithmetic (maths.cpp:77-78: the test is the textbook one seen from the ray
 *     origin mirrored in the plane through the world origin with the triangle's
 *     normal).
 * All are conservative by construction, so images, ids and ray counts are
 * identical in every mode.  mode: 0 = plain loop, every ray against every
 * primitive (used to measure the brute-force roofline); 1 = automatic (default:
 * the lists for primary rays and their shadow rays, the hierarchies for all
 * other rays); 2 = beam screen for primary/shadow rays; 3 = hierarchy for all rays.
 * The mode belongs to a scene: rt_scene_set_screen changes it for later frames
 * of that scene; rt_set_screen sets the mode that scenes created afterwards
 * start with (gpu_render* create theirs inside the call).                     */
void rt_set_screen(int mode);
int rt_get_screen(void);
int rt_scene_set_screen(rt_scene *scene, int mode);

/* Within one frame the shadow walks of a bounce level run on a side stream owned
 * by the scene, beside the next level's nearest-hit walks (default, on: one frame
 * at a time is 10 % faster that way).  A caller that keeps several frames in
 * flight on streams of its own should turn it off for those scenes: the device
 * has 4 hardware queues, and streams that share one serialise (3 frames in flight
 * on one MI355X: 7.8 ms per depth-8 frame without, 8.1 ms with).  Results do not
 * depend on it.  No counterpart in the reference.                              */
int rt_scene_set_side_stream(rt_scene *scene, int on);

/* Bounce levels (max_depth > 1) run in one of two ways.  One launch for all
 * levels (default, on): every wave keeps the rays its rays spawn on a stack of
 * its own, so a frame is three launches whatever its depth.  One launch per stage
 * and level (off): walk / shade / shadow walk with the whole frame's rays queued
 * in HBM between them.  Scenes with a triangle hierarchy (16 or more triangles),
 * without a sphere hierarchy, with more than four lights or with 256 rays per
 * pixel always take the second way.  Results do not depend on it.  No
 * counterpart in the reference.                                                */
int rt_scene_set_one_launch(rt_scene *scene, int on);

/* Stage times of a scene's frames, measured with HIP events on the frame's own
 * stream (no counterpart in the reference).  After rt_scene_time_kernels(scene,
 * 1) every rt_render_device call records events around its stages;
 * rt_scene_kernel_times waits for the last frame and returns, in milliseconds,
 * ms[0] the primary pass (k_render: ray generation, nearest hit, Phong and shadow
 * rays of renderer.cpp:28-140), ms[1] all bounce levels (k_trace, or the
 * launch-per-level chain; 0 without bounces), ms[2] the final sum into the
 * framebuffer.                                                               */
int rt_scene_time_kernels(rt_scene *scene, int on);
int rt_scene_kernel_times(rt_scene *scene, float ms[3]);

/* Which machinery a scene's frames use (no counterpart in the reference; for
 * tests and for hosts that want to know what they pay for).  info[0] nodes of
 * the sphere hierarchy (0: none), info[1] its levels of inner nodes, info[2] 1 if
 * the walks keep their stacks in LDS (trees of at most 8 levels and 65 535
 * nodes; deeper ones use per-lane stacks in scratch memory), info[3] 1 if a frame
 * with bounces runs its bounce levels as one launch (rt_scene_set_one_launch and
 * the conditions listed there), info[4] nodes of the triangle hierarchy (0:
 * none), info[5] lights that have sphere lists by direction, info[6] screen cells
 * that have a sphere list for the view of the last frame (-1: no lists were made),
 * info[7] direction cells of those lights that have no list (-1: no lists).
 * info[6..7] wait for the device and copy the lists to the host.                */
int rt_scene_describe(const rt_scene *scene, int64_t info[8]);

/* As rt_render_device, with a slow self-checking build of the kernel that
 * re-tests every sphere the screen dropped.  dev_stats (device, 16 x uint64,
 * zeroed by the caller) receives
 *   [0] wave-level drops for which some ray's discriminant test
 *       (maths.cpp:132-133) passes - must be 0
 *   [1] spheres the screen kept
 *   [2..4] bounce rays walked through the sphere hierarchy, node bounds tested,
 *          leaf spheres tested;  [5..7] the same for the bounce hits' shadow rays
 *   [8..13] the same six numbers for walks through the triangle hierarchy
 *   [14..15] unused                                                            */
int rt_render_device_selfcheck(rt_scene *scene, const camera *cam,
                               const rt_frame *frame, uint64_t *dev_ray_count,
                               uint64_t *dev_stats);

/* Self-check of the sphere lists of the primary pass (rt_scene_set_screen): for
 * every pixel ray of a w x h frame of this camera, and for every shadow ray of
 * the pixels' hits towards a light that has lists, every sphere of the scene is
 * put through the discriminant test of maths.cpp:132-133; a sphere that passes
 * must be in the list of the ray's cell.  Slow (pixels x spheres); blocking.
 * out[0] pixel rays whose screen cell has a list, out[1] passing spheres missing
 * from those lists - must be 0, out[2] shadow rays served by a list, out[3]
 * passing spheres missing from those - must be 0.  No counterpart in the
 * reference.                                                                  */
int rt_verify_sphere_lists(rt_scene *scene, const camera *cam, int w, int h, uint64_t out[4]);

/* Host-only self-check of the hierarchy builders (no GPU needed): builds the
 * sphere and the triangle hierarchy for the scene and counts violated
 * invariants.  out (8 x int64): [0] sphere nodes (-1: no hierarchy), [1] spheres
 * sticking out of an ancestor box, [2] distance bounds above the true distance,
 * [3] primitives missing from or duplicated in a hierarchy, plus 1 for each
 * hierarchy that is not exactly ceil(log4 leaves) levels deep, [4] triangle nodes
 * (-1: none), [5] triangles kept outside the hierarchy, [6] triangle corners
 * outside an ancestor sphere, [7] triangle normals outside an ancestor cone.
 * [1..3] and [6..7] must be 0.  No counterpart in the reference.             */
int rt_host_check_hierarchies(const scene_desc *scene, int64_t *out);

/* Bulk form of check_intersection (renderer.hpp:21-23): n rays, host arrays.
 * orig/dir are 3*n floats; kind (0 none / 1 triangle / 2 sphere), index
 * (class-local, untouched semantics -> -1 on miss) are n int32; P is 3*n
 * floats (hit point, unspecified on miss).                                   */
int gpu_check_intersection(rt_scene *scene, int64_t n, const float *orig,
                           const float *dir, int32_t *kind, int32_t *index,
                           float *P);

/* The specular term of renderer.cpp:106-110 on n caller-supplied values (host
 * arrays): strength[i] = (float)pow((double)lang[i], 20.0) as the device's
 * shading computes it (five double multiplies instead of libm's pow) and
 * specular[i] = (unsigned char)(color[i] * strength[i]).  For known-answer tests
 * of that one substitution; rendering does not go through this entry point.   */
int gpu_specular_terms(int64_t n, const float *lang, const uint8_t *color,
                       float *strength, uint8_t *specular);

/* check_intersection as the reference's own callers use it (renderer.hpp:21-23:
 * the scene arrays and ONE ray per call): the scene is kept resident exactly as
 * gpu_render keeps it, so only the first call for a set of arrays uploads and
 * builds; a scene that gpu_render (or the shim's render()) left resident serves
 * these calls as it is when the triangles, spheres and radii are the same, so a
 * host that alternates render() and check_intersection() builds once.  Only the
 * geometry is hashed per call (nothing with rt_cache_trust_pointers).
 * Colours play no part: scene->color_tri / color_sphere may be NULL.
 * kind, index, P: one element (three floats for P), semantics as above.      */
int gpu_check_intersection_arrays(const scene_desc *scene, const float *orig,
                                  const float *dir, int32_t *kind,
                                  int32_t *index, float *P);

/* ---- several GPUs of one process -------------------------------------------
 * The OpenMP offload contract of renderer.cpp:13-30 (scene `map(to:)` every
 * worker, rows in parallel, framebuffer rows `map(from:)` back) on n_devices
 * GPUs driven by the calling thread: the scene is made resident on devices
 * 0..n_devices-1 (kept as gpu_render keeps it), bands of band_rows rows
 * (a positive multiple of 8) are dealt round-robin (band b -> device
 * b % n_devices), every device renders its bands on a stream of its own, the
 * bands travel to device 0 peer to peer over xGMI (hipMemcpyPeerAsync), and
 * device 0 hands the whole frame to `rgb` / `hit_ids` (host; hit_ids and n_rays
 * may be NULL).  Blocking, like render().  n_devices = 1 is gpu_render_ids.  */
int gpu_render_multi(const scene_desc *scene, const camera *cam, uint8_t *rgb,
                     int32_t *hit_ids, uint64_t *n_rays, int w, int h,
                     int max_depth, int n_devices, int band_rows);

/* The band bookkeeping of the sharded entry points, for hosts that move the
 * bands themselves: rows [row_begin[k], row_end[k]) for k < return value are the
 * bands of a frame of h rows that share `band_index` of `band_count` owns
 * (band b covers rows b*band_rows ...; b % band_count == band_index).  Writes at
 * most `max_bands` entries; returns the number of owned bands, or a negative
 * rt_status.  Pure host arithmetic (no GPU needed).                          */
int rt_band_plan(int h, int band_rows, int band_index, int band_count,
                 int32_t *row_begin, int32_t *row_end, int max_bands);

/* ---- for callers that may not allocate --------------------------------------
 * No counterpart in the reference.  rt_render_device allocates the bounce
 * machinery of a scene (ray queues, hit lists, counters, per-pixel sums) with
 * hipMalloc on the first frame that needs it, and creates a side stream.  Code
 * under HIP graph capture, tight serving loops and harnesses that forbid
 * allocation cannot have that: this variant renders the same frame (same bytes,
 * ids and ray count) and calls neither hipMalloc / hipFree nor any other
 * allocating routine.  Its bounce machinery lives in a fixed array of
 * RT_NOALLOC_SCRATCH_BYTES bytes that is part of the library's code object (one
 * copy per GPU, held until the process ends); everything runs on frame->stream.
 * A frame whose machinery does not fit returns RT_ERR_CAPACITY and renders
 * nothing (rt_noalloc_scratch_needed tells in advance): with the default 4 rays
 * per pixel, one light and max_depth 8 a 256 x 256 frame needs 52 MB; frames
 * without bounces (max_depth 1, no materials, RT_FRAME_UNLIT) need none and
 * are never refused.  All calls on one GPU share the array: calls that overlap
 * in time on one GPU race, also from different scenes or streams; the caller
 * serialises them.  rt_scene_queue_overflowed does not see this variant's
 * frames: pass dev_overflow (device uint32, or NULL), which the frame sets to
 * 0 or 1.
 * Under stream capture: the FIRST call on a device (it asks the runtime for the
 * array's address, which loads the code object) and the first call after the
 * camera's origin changed (it rebuilds the scene's eye-relative records and
 * waits on events of earlier frames) must have run outside the capture; later
 * calls only enqueue kernels, memsets and one device-to-device copy of four
 * bytes on frame->stream.                                                    */
#define RT_NOALLOC_SCRATCH_BYTES (64u << 20)
int rt_render_device_noalloc(rt_scene *scene, const camera *cam,
                             const rt_frame *frame, uint64_t *dev_ray_count,
                             uint32_t *dev_overflow);
/* Bytes of that array the frame would use (0 = no bounce machinery needed).
 * Only frame->w, h, max_depth, flags and the band fields are read; dev_fb may
 * be NULL here.                                                              */
int rt_noalloc_scratch_needed(rt_scene *scene, const rt_frame *frame,
                              uint64_t *bytes);

/* ---- misc ---------------------------------------------------------------- */
const char *rt_last_error(void);
int rt_device_count(void);
const char *rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
