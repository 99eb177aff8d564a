// rt_trace.h — all bounce levels of a frame in ONE launch.
//
// k_walk / k_shade (rt_walk.h) run a bounce level as three launches with the
// rays of the whole frame queued in HBM between them: 21 dependent launches for
// a depth-8 frame, each ending with the few longest walks of its level while the
// rest of the chip idles, and 6 GB of queue traffic.  Here a wave keeps what its
// rays spawn to itself.  Every wave owns a small stack of work items in device
// memory; it takes first-bounce rays from the frame's queue (written by
// k_render), and everything those rays lead to - shadow rays, mirror and
// refracted rays, down to the last level - is pushed on and popped off its own
// stack.  Nothing is handed from one wave to another inside the launch, so no
// level ever waits for the slowest ray of the frame, and the items stay in L2.
//
// Per lane the work is the walk of k_walk (same arithmetic, same order-free
// nearest-hit rule, same parked leaves); a lane that finishes takes the next
// item.  Rays that found their nearest sphere are appended to a per-wave list of
// hits, and whenever 64 of them are waiting the whole wave shades them together
// (dense lanes, like k_shade): the triangles (renderer.cpp:156-166; this kernel
// is used for scenes whose triangles are looped over), the choice between
// triangle and sphere (renderer.cpp:179-192), Phong per light with the shadow
// ray's triangle test, then shadow items and the next level's rays go on the
// stack.  Lanes that are in the middle of a walk at that moment put their walk
// state (20 words) into a save area of the wave and take it back afterwards, so
// the walk's registers are free while the wave shades and the kernel's register
// budget is the larger of the two phases, not their sum.
//
// Results cannot depend on any of this scheduling: the nearest-hit rule is order
// independent and bounce light is a commutative integer sum (rt_kernels.hip).
#pragma once
#include "rt_walk.h"

namespace rt {

// One work item = two float4, as in the frame's queue:
//   ray    : (origin.xyz, weight > 0)              (dir.xyz, bits of pixel << 6 | depth)
//   shadow : (origin.xyz, bits of pixel | 2^31)    (dir.xyz, bits of the packed light)
// The sign bit of the fourth word tells them apart (a weight is positive, a pixel
// index is below 2^26).
#define RT_ITEM_SHADOW 0x80000000u
static_assert(QUEUE_SEGMENTS % 64 == 0, "k_trace looks for work in 64 segments at a time");

// The wave's stack of items: push for the lanes with `want`, consecutive slots in
// lane order.  Returns the new height.  An item that does not fit is dropped
// and `overflow` is set (the host then renders the frame again another way).
__device__ __forceinline__ uint32_t stack_push(float4 *stack, uint32_t sp, uint32_t cap,
                                               uint32_t *overflow, bool want, float4 a,
                                               float4 b) {
  const unsigned long long m = __ballot(want);
  if (!m)
    return sp;
  const int lane = threadIdx.x & 63;
  const uint32_t slot = sp + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
  if (want) {
    if (slot < cap) {
      stack[2 * (size_t)slot] = a;
      stack[2 * (size_t)slot + 1] = b;
    } else {
      *overflow = 1u;
    }
  }
  const uint32_t top = sp + (uint32_t)__builtin_popcountll(m);
  return top < cap ? top : cap;
}

struct ShadeResult {
  uint32_t sp;     // stack height after the pushes
  uint32_t n_rays; // shadow rays this lane finished here (blocked by a triangle)
};

// Shades n (<= 64) hit records, one per lane: records hits[3 * (first + lane)].
// A record is the ray's item followed by (P.xyz, bits of the sphere's leaf slot or -1).
__device__ __forceinline__ ShadeResult trace_shade(const SceneDev &sc, float4 *stack,
                                                   uint32_t sp, uint32_t cap,
                                                   uint32_t *overflow, const float4 *hits,
                                                   uint32_t first, uint32_t n, int max_depth) {
  const uint32_t lane = threadIdx.x & 63;
  const bool live = lane < n;
  ShadeResult out;
  out.n_rays = 0;
  Hit h;
  h.kind = 0;
  h.index = -1;
  h.Px = h.Py = h.Pz = 0;
  float dx = 0, dy = 0, dz = 0, wgt = 0;
  uint32_t pixel = 0, depth = 0;
  if (live) {
    const float4 *rec = hits + 3 * (size_t)(first + lane);
    const float4 qa = rec[0], qb = rec[1], sh = rec[2];
    const float ox = qa.x, oy = qa.y, oz = qa.z;
    wgt = qa.w;
    dx = qb.x;
    dy = qb.y;
    dz = qb.z;
    const uint32_t tag = __float_as_uint(qb.w);
    pixel = tag >> 6;
    depth = tag & 63u;
    const int slot_s = __float_as_int(sh.w); // the sphere's leaf slot, or -1
    // |P| as the walk computed it: the same three products and sums (maths.cpp:23-25)
    const float len_s = slot_s >= 0 ? length3(sh.x, sh.y, sh.z) : RT_INF;
    float len_t = RT_INF, tPx = 0, tPy = 0, tPz = 0;
    int idx_t = -1;
    lane_triangles(sc, false, ox, oy, oz, dx, dy, dz, len_t, idx_t, tPx, tPy, tPz);
    if (idx_t >= 0 || slot_s >= 0) { // renderer.cpp:179-192
      if (len_t <= len_s) {
        h.kind = 1;
        h.index = idx_t;
        h.Px = tPx;
        h.Py = tPy;
        h.Pz = tPz;
      } else if (len_t > len_s) {
        h.kind = 3; // a sphere by its leaf slot (surface_at, material_at)
        h.index = slot_s;
        h.Px = sh.x;
        h.Py = sh.y;
        h.Pz = sh.z;
      }
    }
  }
  const bool hit = live && h.kind != 0;
  if (!__any(hit)) {
    out.sp = sp;
    return out;
  }
  Surface sf;
  float kr = 0.f, kt = 0.f, ior = 1.f, wl = 0.f;
  if (hit) {
    sf = surface_at(sc, h);
    const float4 m = material_at(sc, h);
    kr = m.x;
    kt = m.y;
    ior = m.z > 0.f ? m.z : 1.0f;
    wl = wgt * ((1.0f - kr) - kt);
  }
  const bool lit = hit && wl >= RT_MIN_WEIGHT;
  if (__any(lit)) {
    for (int l = 0; l < sc.l_size; l++) { // renderer.cpp:82
      float fColor[3] = {0, 0, 0};
      float lx = 0, ly = 0, lz = 0, angle = 0;
      if (lit)
        angle = phong_light(sf, h, dx, dy, dz, sc.lights[l], fColor, lx, ly, lz);
      const uint32_t c0 = (uint32_t)((fColor[0] * angle) * wl),
                     c1 = (uint32_t)((fColor[1] * angle) * wl),
                     c2 = (uint32_t)((fColor[2] * angle) * wl);
      const uint32_t packed = (c0 & 0x3ffu) | ((c1 & 0x3ffu) << 10) | ((c2 & 0x3ffu) << 20);
      // a light that cannot add one unit to any channel casts no shadow ray: whatever
      // the ray found, nothing would be added (on the bench frame that is three in
      // five of the bounce hits' lights: surfaces that face away from the light)
      const bool cast = lit && packed != 0;
      bool blocked = false;
      if (cast) {
        // renderer.cpp:119: the shadow ray starts exactly at P; its triangles here
        float len_t = RT_INF, tx, ty, tz;
        int idx_t = -1;
        lane_triangles(sc, true, h.Px, h.Py, h.Pz, lx, ly, lz, len_t, idx_t, tx, ty, tz);
        blocked = idx_t >= 0;
        if (blocked)
          out.n_rays++; // traced to the end here; the others are counted by the walk
      }
      sp = stack_push(stack, sp, cap, overflow, cast && !blocked,
                      make_float4(h.Px, h.Py, h.Pz, __uint_as_float(pixel | RT_ITEM_SHADOW)),
                      make_float4(lx, ly, lz, __uint_as_float(packed)));
    }
  }
  const bool deeper = hit && (int)depth < max_depth;
  const bool want_t = deeper && kt > 0.f && (wgt * kt >= RT_MIN_WEIGHT);
  const bool want_r = deeper && kr > 0.f && (wgt * kr >= RT_MIN_WEIGHT);
  if (__any(want_t)) {
    float rx = 0, ry = 0, rz = 0;
    if (want_t)
      refract3(dx, dy, dz, sf.nx, sf.ny, sf.nz, ior, rx, ry, rz);
    sp = stack_push(stack, sp, cap, overflow, want_t, make_float4(h.Px, h.Py, h.Pz, wgt * kt),
                    make_float4(rx, ry, rz, __uint_as_float((pixel << 6) | (depth + 1))));
  }
  if (__any(want_r)) {
    float rx = 0, ry = 0, rz = 0;
    if (want_r) {
      reflect3(dx, dy, dz, sf.nx, sf.ny, sf.nz, rx, ry, rz);
      normalize3(rx, ry, rz);
    }
    sp = stack_push(stack, sp, cap, overflow, want_r, make_float4(h.Px, h.Py, h.Pz, wgt * kr),
                    make_float4(rx, ry, rz, __uint_as_float((pixel << 6) | (depth + 1))));
  }
  out.sp = sp;
  return out;
}

// ---- the lane's two small arrays in LDS -------------------------------------------
// Word (row * 256 + thread).  Rows 0 .. LDS_STACK_ROWS-1 are the lane's stack of
// nodes still to visit (rt_device.h, stack_children), the next TRACE_PARK_ROWS its
// parked leaves; five blocks of 31 KB share a CU's LDS (of five blocks of 32 KB
// the hardware runs only four).  The host sends trees too deep for the LDS stack
// to the launch-per-level kernels.
//
// The visit is written without branches: every child is parked by an
// unconditional store whose row only advances when the child is a touched leaf, and
// the touched inner children are ordered and pushed the same way.  A walk's pace is
// set by how many instructions a wave issues per visit, and the branching form (per
// child a dozen exec-mask branches, each a handful of scalar instructions and a
// wait for them) was more than twice as many.  For the stores to stay in range a
// lane never enters a visit with more than TRACE_PARK_FULL parked leaves: the wave
// works the parked leaves off first (all lanes together, as at every refill).
#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES 5
#endif
// The column has TRACE_ROWS rows in all.  How many of them the stack needs depends on
// the scene's tree (three per level of inner nodes but the last, and the row of the
// stores that do not count), and every row it does not need is one more parked leaf
// before a lane forces a work-off: the kernel exists for stacks of 22, 19 and 16 rows
// (trees of 8, 7 and at most 6 levels), and launch_trace picks by the tree's depth.
// The bench scene's tree has 7 levels: 12 parking rows instead of 9 took its bounce
// stage from 3.91 to 3.83 ms.
constexpr int TRACE_ROWS = 31;
constexpr int trace_stack_rows(int depth) { return 3 * (depth - 1) + 1; }
static_assert(trace_stack_rows(LDS_WALK_MAX_DEPTH) == LDS_STACK_ROWS, "the deepest admitted tree");
// A walk steps aside for a batch of shading into its own column (k_trace): sixteen
// rows from its stack height on, and the column's last row.
constexpr int TRACE_SAVE_MAX_SP = TRACE_ROWS - 1 - 16;
static_assert(TRACE_ROWS * 1024 < 32768,
              "five blocks per CU: five times 32 KB, all of a CU's LDS, are not admitted");

#ifdef RT_STAMPS
// Diagnostic build only (make stamps): where a wave of k_trace spends its cycles.
// [0] wave cycles, then cycles in [1] node visits, [2] parked leaves, [3] handing in,
// [4] shading (with saving and restoring the walks), [5] fetching; counts of
// [6] visits, [7] service phases, [8] batches of four screens, [9] rounds of exact
// leaf tests, [10] batches shaded, [12] waves, [13] service phases forced by a
// lane's full row of parked leaves; lanes summed over [11] the visits (lanes
// walking), [14] the exact-test rounds (lanes testing), [15] the screen batches
// (lanes with a parked leaf in the batch); [16] the longest wave's cycles; [17] the
// end time of the last wave and [18] the sum of the waves' end times, both in
// cycles after the first wave's start ([19])
__device__ unsigned long long g_tstamps[24];
// A timeline of every TLOG_EVERY-th wave (tools/trace_timeline.py): one record per
// service phase, {cycles since the wave started / 16, lanes walking at entry, items on
// the wave's stack, hits waiting | drained << 16 | visits since the last phase << 20}
constexpr int TLOG_WAVES = 32, TLOG_RECS = 1024, TLOG_EVERY = 160;
__device__ uint32_t g_tlog[TLOG_WAVES][TLOG_RECS][4];
#define RT_TSTAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define RT_TCOUNT(k, v) tst[k] += (v)
#else
#define RT_TSTAMP(var)
#define RT_TCOUNT(k, v)
#endif

// k_trace — see the head of this file.  Grid: any number of 256-thread blocks;
// wave w of the grid owns stack[w * stack_cap ...] and hits[w * hit_cap ...].
// TRACE_STACK_ROWS: rows of a lane's column that hold its stack of nodes (the scene's
// tree must not be deeper than they serve); the others hold its parked leaves.
template <int TRACE_STACK_ROWS>
__global__ __launch_bounds__(256, RT_TRACE_WAVES) void k_trace(SceneDev sc, TraceArgs t) {
  constexpr int TRACE_PARK_ROWS = TRACE_ROWS - TRACE_STACK_ROWS;
  constexpr int TRACE_PARK_FULL = TRACE_PARK_ROWS - BVH_WIDTH;
  const int lane = threadIdx.x & 63;
  // The block starts on segment (block % QUEUE_SEGMENTS) of the frame's queue and
  // moves on to the other segments when that one is used up (any wave may take
  // from any segment: the cursors are atomics, the lengths were written by
  // k_render before this launch).  Without this a wave whose segment held little
  // or cheap work is done long before the others: the mean wave lived 70 % of the
  // longest one.
  uint32_t seg = my_segment();
  uint32_t n_in = t.in.count[seg * CTL_WORDS];
  if (n_in > t.in.seg_capacity)
    n_in = t.in.seg_capacity;
  size_t seg_base = (size_t)seg * t.in.seg_capacity;
  uint32_t *cursor = t.cursor + seg * CTL_WORDS;
  const size_t wave_id = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  float4 *const stack = t.stack + 2 * wave_id * t.stack_cap;
  float4 *const hits = t.hits + 3 * wave_id * t.hit_cap;
  float4 *const save = t.save + TRACE_SAVE_WORDS / 4 * 64 * wave_id + lane; // [k * 64 + lane]
  uint32_t sp_w = 0;   // items on the wave's stack (the same in every lane)
  uint32_t n_hits = 0; // hit records waiting to be shaded (the same in every lane)

  enum { EMPTY = 0, BUSY = 1, DRAIN = 2 }; // DRAIN: walk over, parked leaves left
  int state = EMPTY;
  bool shadow = false;
  float4 qa = make_float4(0.f, 0.f, 0.f, 0.f), qb = qa; // the lane's item
  float a = 0, a4 = 0, ix = 0, iy = 0, iz = 0, ida = 0;
  float len_s = RT_INF, bound = RT_INF;
  // the nearest sphere so far is kept by its leaf slot: the walk never needs its
  // index in the scene's arrays, except to settle a tie in |P| (renderer.cpp:170:
  // the lower index wins), and a scattered read per accepted hit is saved
  int slot_s = -1;
  float sPx = 0, sPy = 0, sPz = 0;
  __shared__ uint32_t lds_words[TRACE_ROWS * 256];
  uint32_t *const stack_l = lds_words + threadIdx.x; // stack_l[k * 256]: k-th stacked node
  uint32_t *const park = stack_l + TRACE_STACK_ROWS * 256; // park[k * 256]: k-th parked leaf
  int sp = 0, node = 0;
  int n_park = 0;
  uint32_t n_rays = 0;
  bool drained = false; // the frame's queue has nothing left for this block

#ifdef RT_STAMPS
  unsigned long long tst[24] = {};
  float dbg_near = 0.f;
  unsigned long long tlog_visits = 0;
#endif
  RT_TSTAMP(t_begin);

  while (true) {
    const int n_free = __builtin_popcountll(__ballot(state != BUSY));
    const bool more = sp_w > 0 || !drained;
    if (n_free == 64 || (more && n_free >= t.fetch_at) ||
        __any(state == BUSY && n_park > TRACE_PARK_FULL)) {
      RT_TSTAMP(t_s0);
      RT_TCOUNT(7, 1);
#ifdef RT_STAMPS
      if (wave_id % TLOG_EVERY == 0 && wave_id / TLOG_EVERY < TLOG_WAVES && tst[7] <= TLOG_RECS &&
          lane == 0) {
        uint32_t *r = g_tlog[wave_id / TLOG_EVERY][tst[7] - 1];
        r[0] = (uint32_t)((t_s0 - t_begin) >> 4);
        r[1] = 64u - (uint32_t)n_free;
        r[2] = sp_w;
        r[3] = n_hits | ((drained ? 1u : 0u) << 16) | ((uint32_t)(tst[6] - tlog_visits) << 20);
      }
      tlog_visits = tst[6];
#endif
      RT_TCOUNT(13, (n_free == 64 || (more && n_free >= t.fetch_at)) ? 0 : 1);
      // ---- work off parked leaves, all lanes together: the cheap screen on every
      // parked leaf first, then the full test only as often as the busiest lane
      // has passers ----
      {
        // Four parked leaves at a time: their rows, then their records and indices,
        // then the arithmetic, so that a lane waits once per four leaves.  Every leaf
        // gets the discriminant screen of maths.cpp:132-133 (most fail it); the rest
        // of the test runs, on the same b and c, only while some lane has a passer.
        // A passer with c > 0 and b > eps is dropped as well: maths.cpp:136-152 then
        // computes q < 0 and two roots <= 0 (or, with discr == 0, -0.5 b / a < 0) and
        // reports no hit.
        for (int k0 = 0; __any(k0 < n_park); k0 += 4) {
          RT_TCOUNT(8, 1);
          RT_TCOUNT(15, __builtin_popcountll(__ballot(k0 < n_park)));
          uint32_t slot[4];
          float4 g[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int k = k0 + j < TRACE_PARK_ROWS ? k0 + j : TRACE_PARK_ROWS - 1;
            slot[j] = park[k * 256];
          }
#pragma unroll
          for (int j = 0; j < 4; j++) {
            slot[j] = k0 + j < n_park ? (~slot[j]) >> 3 : 0u;
            g[j] = sc.leaf_geo[slot[j]];
          }
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const float Lx = qa.x - g[j].x, Ly = qa.y - g[j].y, Lz = qa.z - g[j].z; // maths.cpp:126
            const float b = 2.0f * dot3(qb.x, qb.y, qb.z, Lx, Ly, Lz);               // :128
            const float c = dot3(Lx, Ly, Lz, Lx, Ly, Lz) - g[j].w;                   // :129
            const float bb = b * b, ac = a4 * c;                                     // :132
            const bool may = (k0 + j < n_park) & !(bb < ac) & !((c > 0.f) & (b > 0.001f)) &
                             !(shadow & (slot_s >= 0));
            if (__any(may)) {
              RT_TCOUNT(9, 1);
              RT_TCOUNT(14, __builtin_popcountll(__ballot(may)));
              float Px, Py, Pz;
              if (may && sphere_finish(a, b, c, bb, ac, qa.x, qa.y, qa.z, qb.x, qb.y, qb.z, Px, Py,
                                       Pz)) {
                const float len = length3(Px, Py, Pz);
                // smallest |P|, lowest index on ties (renderer.cpp:168-177, order-free)
                if (len < RT_INF &&
                    (len < len_s ||
                     (len == len_s && sc.leaf_idx[slot[j]] < sc.leaf_idx[slot_s]))) {
                  len_s = len;
                  slot_s = (int)slot[j];
                  sPx = Px;
                  sPy = Py;
                  sPz = Pz;
                  if (shadow)
                    state = DRAIN; // occluded: nothing more to find
                  else
                    bound = fminf(bound, len);
                }
              }
            }
          }
        }
        n_park = 0;
      }
      RT_TSTAMP(t_s1);
      RT_TCOUNT(2, t_s1 - t_s0);
      // ---- hand in finished items ----
      {
        const bool done = state == DRAIN;
        if (done && shadow && slot_s < 0) // unoccluded: renderer.cpp:119-127 as an integer sum
          add_light(t.extra, __float_as_uint(qa.w) & ~RT_ITEM_SHADOW, __float_as_uint(qb.w));
        const bool rec = done && !shadow;
        const unsigned long long m = __ballot(rec);
        if (rec) {
          float4 *r = hits + 3 * (size_t)(n_hits + (uint32_t)__builtin_popcountll(
                                                       m & ((1ull << lane) - 1ull)));
          r[0] = qa;
          r[1] = qb;
          r[2] = make_float4(sPx, sPy, sPz, __int_as_float(slot_s));
        }
        n_hits += (uint32_t)__builtin_popcountll(m);
        if (done)
          state = EMPTY;
      }
      RT_TSTAMP(t_s2);
      RT_TCOUNT(3, t_s2 - t_s1);
      // ---- shade the waiting hits, 64 at a time (the last few when nothing else is left) ----
      if (n_hits >= 64 || (n_hits > 0 && sp_w == 0 && drained)) {
        // Walks in progress step aside.  At this point a lane's parking rows are
        // empty (worked off above) and so are the rows of its column above its stack
        // of nodes: rows sp .. TRACE_ROWS - 1, one stretch of
        // the column.  The walk state, sixteen words, goes into the first sixteen of
        // them, and the stack height itself, with the lane's state, into the last row
        // of the column.  Through device memory (five float4 per lane out and back
        // around every batch) the saved walks were a third of the kernel's traffic.
        // A lane whose stack leaves fewer than seventeen rows (trees near the depth
        // limit) sends the whole wave through the save area in device memory.
        const bool save_in_lds = !__any(state == BUSY && sp > TRACE_SAVE_MAX_SP);
        if (save_in_lds) {
          uint32_t *const sv = stack_l + sp * 256;
          sv[0 * 256] = __float_as_uint(qa.x);
          sv[1 * 256] = __float_as_uint(qa.y);
          sv[2 * 256] = __float_as_uint(qa.z);
          sv[3 * 256] = __float_as_uint(qa.w);
          sv[4 * 256] = __float_as_uint(qb.x);
          sv[5 * 256] = __float_as_uint(qb.y);
          sv[6 * 256] = __float_as_uint(qb.z);
          sv[7 * 256] = __float_as_uint(qb.w);
          sv[8 * 256] = __float_as_uint(len_s);
          sv[9 * 256] = __float_as_uint(bound);
          sv[10 * 256] = (uint32_t)slot_s;
          sv[11 * 256] = __float_as_uint(sPx);
          sv[12 * 256] = __float_as_uint(sPy);
          sv[13 * 256] = __float_as_uint(sPz);
          sv[14 * 256] = (uint32_t)node;
          sv[15 * 256] = n_rays;
          park[(TRACE_PARK_ROWS - 1) * 256] = (uint32_t)sp | ((uint32_t)state << 8);
        } else {
          save[0 * 64] = qa;
          save[1 * 64] = qb;
          save[2 * 64] = make_float4(len_s, bound, __int_as_float(slot_s), sPx);
          save[3 * 64] = make_float4(sPy, sPz, __int_as_float(sp), __int_as_float(node));
          save[4 * 64] = make_float4(__int_as_float(state << 8), __uint_as_float(n_rays), 0.f, 0.f);
        }
        uint32_t shade_rays = 0;
        while (n_hits >= 64 || (n_hits > 0 && sp_w == 0 && drained)) {
          const uint32_t n = n_hits < 64 ? n_hits : 64;
          n_hits -= n;
          RT_TCOUNT(10, 1);
          const ShadeResult r =
              trace_shade(sc, stack, sp_w, t.stack_cap, t.overflow, hits, n_hits, n, t.max_depth);
          sp_w = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.sp);
          shade_rays += r.n_rays;
        }
        // nothing computed before this point may be carried in registers past it:
        // the walk state below is what the loads return
        asm volatile("" ::: "memory");
        if (save_in_lds) {
          const uint32_t packed = park[(TRACE_PARK_ROWS - 1) * 256];
          sp = (int)(packed & 0xffu);
          state = (int)(packed >> 8);
          const uint32_t *const sv = stack_l + sp * 256;
          qa = make_float4(__uint_as_float(sv[0 * 256]), __uint_as_float(sv[1 * 256]),
                           __uint_as_float(sv[2 * 256]), __uint_as_float(sv[3 * 256]));
          qb = make_float4(__uint_as_float(sv[4 * 256]), __uint_as_float(sv[5 * 256]),
                           __uint_as_float(sv[6 * 256]), __uint_as_float(sv[7 * 256]));
          len_s = __uint_as_float(sv[8 * 256]);
          bound = __uint_as_float(sv[9 * 256]);
          slot_s = (int)sv[10 * 256];
          sPx = __uint_as_float(sv[11 * 256]);
          sPy = __uint_as_float(sv[12 * 256]);
          sPz = __uint_as_float(sv[13 * 256]);
          node = (int)sv[14 * 256];
          n_rays = sv[15 * 256] + shade_rays;
        } else {
          const float4 s2 = save[2 * 64], s3 = save[3 * 64], s5 = save[4 * 64];
          qa = save[0 * 64];
          qb = save[1 * 64];
          len_s = s2.x;
          bound = s2.y;
          slot_s = __float_as_int(s2.z);
          sPx = s2.w;
          sPy = s3.x;
          sPz = s3.y;
          sp = __float_as_int(s3.z);
          node = __float_as_int(s3.w);
          state = (__float_as_int(s5.x) >> 8) & 0xff;
          n_rays = __float_as_uint(s5.y) + shade_rays;
        }
        {
          n_park = 0; // worked off before the batch
          // a walking lane's item says which kind it is (a lane without an item gets
          // the flag with its next one)
          shadow = (__float_as_uint(qa.w) & RT_ITEM_SHADOW) != 0;
          // per-ray constants of the walk, by the operations that made them at the fetch
          a = dot3(qb.x, qb.y, qb.z, qb.x, qb.y, qb.z);
          a4 = 4.0f * a;
          const SphWalkRay wr = sph_walk_ray(qb.x, qb.y, qb.z, a);
          ix = wr.ix;
          iy = wr.iy;
          iz = wr.iz;
          ida = wr.ida;
        }
      }
      RT_TSTAMP(t_s3);
      RT_TCOUNT(4, t_s3 - t_s2);
      // ---- fetch: the wave's own stack first, then the frame's queue ----
      {
        unsigned long long e = __ballot(state == EMPTY);
        int n_e = __builtin_popcountll(e);
        bool got = false;
        if (n_e && sp_w) {
          const uint32_t k = sp_w < (uint32_t)n_e ? sp_w : (uint32_t)n_e;
          const uint32_t mine = (uint32_t)__builtin_popcountll(e & ((1ull << lane) - 1ull));
          if (state == EMPTY && mine < k) {
            const size_t s = (size_t)(sp_w - 1 - mine);
            qa = stack[2 * s];
            qb = stack[2 * s + 1];
            got = true;
          }
          sp_w -= k;
        }
        e = __ballot(state == EMPTY && !got);
        n_e = __builtin_popcountll(e);
        if (n_e && !drained) {
          const int leader = __builtin_ctzll(e);
          uint32_t base = 0;
          if (lane == leader)
            base = atomicAdd(cursor, (uint32_t)n_e);
          base = __builtin_amdgcn_readlane(base, leader);
          if (state == EMPTY && !got) {
            const uint32_t r = base + (uint32_t)__builtin_popcountll(e & ((1ull << lane) - 1ull));
            if (r < n_in) {
              qa = t.in.items[2 * (seg_base + r)];
              qb = t.in.items[2 * (seg_base + r) + 1];
              got = true;
            }
          }
          if (base + (uint32_t)n_e >= n_in) {
            // This segment is used up: on to the next one that still has rays to
            // hand out (lanes it could not serve are served at the next refill), or
            // nothing is left.  The lanes look at 64 segments at a time; a cursor
            // read here may be behind, which only costs a useless attempt later.
            // (every wave starts its search somewhere else: waves that all went to the
            // next segment would use it up together and move on as a crowd, all on
            // one cursor)
            bool found = false;
            const uint32_t hop = 1u + (uint32_t)(wave_id * 37u) % (QUEUE_SEGMENTS - 1);
            for (uint32_t round = 0; round < QUEUE_SEGMENTS / 64 && !found; round++) {
              const uint32_t s = (seg + hop + round * 64 + (uint32_t)lane) % QUEUE_SEGMENTS;
              uint32_t n = t.in.count[s * CTL_WORDS];
              if (n > t.in.seg_capacity)
                n = t.in.seg_capacity;
              const uint32_t c = __hip_atomic_load(t.cursor + s * CTL_WORDS, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
              const unsigned long long m = __ballot(c < n);
              if (m != 0) {
                const int first = __builtin_ctzll(m);
                seg = (uint32_t)__builtin_amdgcn_readlane((int)s, first);
                n_in = (uint32_t)__builtin_amdgcn_readlane((int)n, first);
                found = true;
              }
            }
            if (found) {
              seg_base = (size_t)seg * t.in.seg_capacity;
              cursor = t.cursor + seg * CTL_WORDS;
            } else {
              drained = true;
            }
          }
        }
        if (got) {
          shadow = (__float_as_uint(qa.w) & RT_ITEM_SHADOW) != 0;
          n_rays++;
          len_s = RT_INF;
          slot_s = -1;
          bound = RT_INF;
          sp = 0;
          node = 0;
          n_park = 0;
          state = BUSY;
          a = dot3(qb.x, qb.y, qb.z, qb.x, qb.y, qb.z); // maths.cpp:127
          a4 = 4.0f * a;
          const SphWalkRay wr = sph_walk_ray(qb.x, qb.y, qb.z, a);
          ix = wr.ix;
          iy = wr.iy;
          iz = wr.iz;
          ida = wr.ida;
          const bool sane = a > 9.094947e-13f && a < 1.0995116e12f;
          if (!(a > 0.f) || sc.s_size == 0) {
            state = DRAIN; // no sphere can be hit (rt_device.h, beam screen notes)
          } else if (!sane) {
            // out-of-range direction length: this lane tests every sphere (every
            // sphere has a leaf slot)
            for (int k = 0; k < sc.s_size; k++) {
              const float4 g = sc.leaf_geo[k];
              float Px, Py, Pz;
              if (sphere_exact(g.x, g.y, g.z, g.w, a, a4, qa.x, qa.y, qa.z, qb.x, qb.y, qb.z, Px,
                               Py, Pz)) {
                const float len = length3(Px, Py, Pz);
                if (len < RT_INF &&
                    (len < len_s || (len == len_s && sc.leaf_idx[k] < sc.leaf_idx[slot_s]))) {
                  len_s = len;
                  slot_s = k;
                  sPx = Px;
                  sPy = Py;
                  sPz = Pz;
                  if (shadow)
                    break;
                }
              }
            }
            state = DRAIN;
          }
        }
      }
#ifdef RT_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      tst[5] += __builtin_amdgcn_s_memtime() - t_s3;
#endif
      if (!__any(state != EMPTY) && sp_w == 0 && n_hits == 0 && drained)
        break;
      continue;
    }
    RT_TSTAMP(t_v0);
    RT_TCOUNT(6, 1);
    RT_TCOUNT(11, __builtin_popcountll(__ballot(state == BUSY)));
#ifdef RT_STAMPS
    tst[22] += __builtin_popcountll(__ballot(state == BUSY && !shadow));
    tst[23] += __builtin_popcountll(__ballot(state == BUSY && !shadow && dbg_near > bound * 1.01f + 0.01f));
#endif

    if (state == BUSY) {
      // ---- one node visit: the bounds of its (up to four) children ----
      bool touch[BVH_WIDTH];
      float near_key[BVH_WIDTH], entry_key[BVH_WIDTH];
      const BvhNode N = sc.bvh[node];
      bvh_visit2(N, qa.x, qa.y, qa.z, ix, iy, iz, ida, bound, touch, near_key, entry_key);
#pragma unroll
      for (int c = 0; c < BVH_WIDTH; c++) {
        // a touched leaf is parked
        const int32_t link = N.link[c];
        park[n_park * 256] = (uint32_t)link; // the leaf's slot is (~link) >> 3
        n_park += (touch[c] && link < 0) ? 1 : 0;
      }
      // the touched inner children: the nearest is visited next, the others stacked
      // (ordered by the children's distance bounds from the world origin for shadow
      // rays as well: ordering those by where the line enters a child, which ends an
      // occluded walk sooner, was measured 1 % slower than not computing that key)
      const uint32_t next = stack_children(N, touch, entry_key, false, stack_l, sp);
#ifdef RT_STAMPS
      dbg_near = __uint_as_float(next & 0xffff0000u);
#endif
      if (next != NO_ENTRY)
        node = (int)(next & 0xffffu);
      else
        state = DRAIN;
    }
#ifdef RT_STAMPS
    tst[1] += __builtin_amdgcn_s_memtime() - t_v0;
#endif
  }
#ifdef RT_STAMPS
  const unsigned long long t_end = __builtin_amdgcn_s_memtime();
  tst[0] = t_end - t_begin;
  tst[12] = 1;
  if (lane == 0) {
    for (int k = 0; k < 16; k++)
      atomicAdd(&g_tstamps[k], tst[k]);
    atomicMax(&g_tstamps[16], tst[0]);
    atomicMax(&g_tstamps[17], t_end);
    atomicAdd(&g_tstamps[18], t_end & 0xffffffffffull);
    atomicMin(&g_tstamps[19], t_begin);
    atomicAdd(&g_tstamps[22], tst[22]);
    atomicAdd(&g_tstamps[23], tst[23]);
    atomicAdd(&g_tstamps[20], tst[6] == 0 ? 1ull : 0ull);            // waves that never visited a node
    atomicAdd(&g_tstamps[21], tst[0] * 2 < 20000000ull ? 1ull : 0ull); // waves under 10 M cycles
  }
#endif
  add_ray_count(t.ray_count, n_rays);
}

} // namespace rt
