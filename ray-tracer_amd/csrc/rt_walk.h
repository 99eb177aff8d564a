// rt_walk.h — bounce levels as a stream of small kernels with dynamic ray fetch.
//
// A bounce ray's walk through the sphere hierarchy takes anywhere from a dozen
// to several hundred node visits (a ray skimming along a plane of spheres
// touches all of them), so a wave that owns 64 fixed rays spends most of its
// time with a few lanes working (measured: 8-19 of 64 lanes active).  Here a
// lane that has finished its ray fetches the next one from the queue instead of
// waiting for its neighbours, and the three phases of a bounce level run as
// separate launches so that every launch starts with dense lanes:
//
//   k_walk<false>   queue of rays      -> nearest hit per ray (compact hit list)
//   k_shade         hit list           -> per light: shadow-ray items; next level's rays
//   k_walk<true>    shadow-ray items   -> integer light added to the pixel if unoccluded
//
// k_walk exists per class of primitive.  A scene with a triangle hierarchy runs
// the triangle walk before the sphere walk in both places; with only a handful
// of triangles k_shade loops over them instead.
//
// The arithmetic per ray and per primitive is the same trace described in
// rt_device.h; only the scheduling differs.  Results cannot depend on it: the
// nearest-hit rule is order independent (smallest |P|, lowest index on ties) and
// bounce light is a commutative integer sum.
#pragma once
#include "rt_device.h"
#include "rt_kernels.h"

namespace rt {


// Tests every triangle for one lane's ray (renderer.cpp:156-166).
__device__ __forceinline__ void lane_triangles(const SceneDev &sc, bool any_hit, float ox,
                                               float oy, float oz, float dx, float dy,
                                               float dz, float &len_t, int &idx_t, float &tPx,
                                               float &tPy, float &tPz) {
  for (int i = 0; i < sc.t_size; i++) {
    const TriRec &T = sc.tri[i];
    float no = dot3(T.n_d.x, T.n_d.y, T.n_d.z, ox, oy, oz);
    float Px, Py, Pz;
    if (triangle_test(T, no, ox, oy, oz, dx, dy, dz, Px, Py, Pz)) {
      float len = length3(Px, Py, Pz);
      if (len_t > len) { // renderer.cpp:159, strict
        len_t = len;
        idx_t = i;
        tPx = Px;
        tPy = Py;
        tPz = Pz;
        if (any_hit)
          return;
      }
    }
  }
}

// An unoccluded bounce hit's light: three integers, 10 bits apiece in `packed`,
// added to the pixel's side sums with ONE 64-bit atomic (21 bits per channel;
// the weights of a pixel's rays sum to at most 1 per light, so a channel's sum
// stays below 255 x lights + rays, far from 2^21).  Scattered atomics leave L2
// as 64-byte requests each: three per ray were 3.8 GB of HBM writes per frame.
__device__ __forceinline__ void add_light(unsigned long long *extra, uint32_t pixel,
                                          uint32_t packed) {
  const unsigned long long c0 = packed & 0x3ffu, c1 = (packed >> 10) & 0x3ffu,
                           c2 = (packed >> 20) & 0x3ffu;
  const unsigned long long v = c0 | (c1 << 21) | (c2 << 42);
  if (v)
    atomicAdd(extra + pixel, v);
}

// One exact sphere test for a lane's ray against leaf slot `slot`, folded into
// the lane's running nearest sphere hit (smallest |P|, lowest index on ties:
// the order-free form of renderer.cpp:168-177).  Returns true on an accepted hit.
__device__ __forceinline__ bool leaf_test(const SceneDev &sc, int slot, float a, float a4,
                                          float ox, float oy, float oz, float dx, float dy,
                                          float dz, float &len_s, int &idx_s, float &sPx,
                                          float &sPy, float &sPz) {
  const float4 g = sc.leaf_geo[slot];
  float Px, Py, Pz;
  if (!sphere_exact(g.x, g.y, g.z, g.w, a, a4, ox, oy, oz, dx, dy, dz, Px, Py, Pz))
    return false;
  const float len = length3(Px, Py, Pz);
  const int idx = sc.leaf_idx[slot];
  if (len < RT_INF && (len < len_s || (len == len_s && idx < idx_s))) {
    len_s = len;
    idx_s = idx;
    sPx = Px;
    sPy = Py;
    sPz = Pz;
    return true;
  }
  return false;
}

// Lanes of one wave hand words to each other through LDS: the hardware runs a
// wave's LDS instructions in order, but a thread never reads what it wrote itself
// here, so nothing tells the compiler to keep the order; this does (and waits).
#define RT_WAVE_SYNC_LDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
constexpr int PARK_SLOTS = 8; // parked leaves per lane (rt_walk.h, k_walk)
constexpr int WALK_SHARE_AT = 16; // idle lanes at which walks in progress are shared out
static_assert(QUEUE_SEGMENTS % 64 == 0, "a wave looks for work in 64 segments at a time");

// The discriminant screen of maths.cpp:132-133 for leaf slot `slot`: can the
// exact test pass at all?  (The same b, c, products and comparison as
// sphere_exact; a lane whose screen fails need not run the rest.)
__device__ __forceinline__ bool leaf_screen(float4 g, float a4, float ox, float oy, float oz,
                                            float dx, float dy, float dz) {
  const float Lx = ox - g.x, Ly = oy - g.y, Lz = oz - g.z;
  const float b = 2.0f * dot3(dx, dy, dz, Lx, Ly, Lz);
  const float c = dot3(Lx, Ly, Lz, Lx, Ly, Lz) - g.w;
  return !(b * b < a4 * c);
}
__device__ __forceinline__ bool leaf_screen(const SceneDev &sc, int slot, float a4, float ox,
                                            float oy, float oz, float dx, float dy, float dz) {
  return leaf_screen(sc.leaf_geo[slot], a4, ox, oy, oz, dx, dy, dz);
}

// k_walk — each lane walks one hierarchy (TRI: the triangles', else the
// spheres') for one queue item at a time and fetches the next item when it is
// done.
//   ANY_HIT = false: item = bounce ray; the nearest hit of this class (P, |P|,
//                    index or -1) is written to hits[slot of the ray]
//   ANY_HIT = true : item = shadow ray (tagA = pixel, tagB = packed light).
//                    Last stage: the light is added to the pixel if nothing is hit.
//                    Earlier stage (another class of primitive follows): an occluded
//                    item is cancelled in place (direction 0, light 0), so that the
//                    walk which follows finds nothing to do and nothing to add.
// Leaves are single primitives.  Running the exact test the moment a leaf is
// touched would make every iteration of the wave pay for it with one or two lanes
// active, so a lane parks up to PARK_SLOTS touched leaves (in LDS, word
// slot * 256 + thread: conflict-free whatever the lanes' counts are) and the whole
// wave works them off together whenever it stops to refill.  With four slots in
// registers 29 % of the visits still had some lane test a leaf on the spot, a
// couple of hundred instructions for one or two lanes each time.
// (80 VGPRs, 6 waves per SIMD for the sphere walk.  Forcing 7 or 8 waves makes it
// spill 8 or 16 registers and costs more than the occupancy brings: 8.44 and
// 9.42 ms per depth-8 frame against 8.17.)
#ifdef RT_STAMPS
// Diagnostic build only (make stamps): where a walking wave's cycles go.
// [0] wave cycles in k_walk<false,false>, [1] in service phases, [2] from a node
// load's issue to its data, [3] rest of the visits, [4] service phases, [5] visits,
// [6] waves, [7] of the service time: waiting for the cursor atomic and the items,
// [8] visits in which some lane tested a leaf on the spot (four already parked),
// [9] lanes with a ray, summed over the visits, [10] of the service phases: testing parked leaves
__device__ unsigned long long g_stamps[16];
#define RT_STAMP() __builtin_amdgcn_s_memtime()
#ifdef RT_STAMPS_TRI // the same for the nearest-hit walk over the triangles
constexpr bool STAMPS_OF_TRI = true;
#else
constexpr bool STAMPS_OF_TRI = false;
#endif
#endif

template <bool ANY_HIT, bool TRI>
__global__ __launch_bounds__(256, 5) void k_walk(SceneDev sc, WalkArgs w) {
  const int lane = threadIdx.x & 63;
#ifdef RT_STAMPS
  unsigned long long st_serv = 0, st_load = 0, st_visit = 0, st_fetch = 0, n_serv = 0, n_visit = 0;
  unsigned long long n_over = 0, n_busy = 0, st_work = 0;
  const unsigned long long st_begin = RT_STAMP();
#endif
  // The block starts on segment (block % n_seg) of the queue and moves on to the
  // other segments when that one is used up (the cursors are atomics, the lengths
  // were written before this launch): a wave that stayed with one segment spent
  // the end of every launch on the longest walks of that segment's last rays, and
  // the nearest-hit walk over a million triangles ran with 25 of 64 lanes walking.
  uint32_t seg = my_segment();
  uint32_t n_in = w.in.count[seg * CTL_WORDS];
  if (n_in > w.in.seg_capacity)
    n_in = w.in.seg_capacity;
  size_t seg_base = (size_t)seg * w.in.seg_capacity;
  uint32_t *cursor = w.cursor + seg * CTL_WORDS;

  // DRAIN: walk over, parked leaves left.  OVER: those tested too; the lane waits
  // for the lanes that took over parts of its walk, if any.  REPORT: see below.
  enum { EMPTY = 0, BUSY = 1, DRAIN = 2, OVER = 3, REPORT = 4 };
  int state = EMPTY;
  // Sharing a walk.  Once the queue is used up, a lane with nothing to do takes the
  // oldest entry off the stack of a lane that is still walking (the subtree nearest
  // the root) and walks it for that lane's ray: `owner` is the lane it took it from,
  // `n_help` counts the lanes that have taken from this one and not yet reported.
  // A helper's nearest hit is folded into its owner's by the same order-free rule
  // (smallest |P|, lowest index on ties), so the result is that of the undivided
  // walk; its bound is the owner's at the time, which can only be looser.  Without
  // this the end of every launch belonged to the few longest walks of the level: a
  // grazing ray over a million-triangle height field visits hundreds of nodes, and
  // the waves of the nearest-hit triangle walk had 25 of 64 lanes walking.
  int owner = -1, n_help = 0;
  __shared__ int pair_words[256]; // per wave: the lanes to take from, by rank
  uint32_t ray = 0, tagA = 0, tagB = 0; // ray: the item's slot in the whole queue
  float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, a = 0, a4 = 0, da = 0, ol = 0;
  float ix = 0, iy = 0, iz = 0; // spheres: 1/d per axis; `da` then holds >= 1/|d|
  float len_s = RT_INF, bound = RT_INF;
  int idx_s = -1;
  float sPx = 0, sPy = 0, sPz = 0;
  __shared__ int stack_words[LDS_STACK_ROWS * 256];
  LdsStack stack;
  stack.lds = stack_words + threadIdx.x;
  int sp = 0, node = 0;
  __shared__ int park_words[PARK_SLOTS * 256];
  int *const park = park_words + threadIdx.x; // park[k * 256]: this lane's k-th parked leaf
  int n_park = 0;
  uint32_t n_rays = 0;
  bool drained = false;

  // exact test of leaf `slot` for this lane's ray, folded into its nearest hit
  auto leaf = [&](int slot) -> bool {
    if (TRI)
      return tri_fold_leaf(sc.tleaf[slot], ox, oy, oz, dx, dy, dz, len_s, idx_s, sPx, sPy, sPz);
    return leaf_test(sc, slot, a, a4, ox, oy, oz, dx, dy, dz, len_s, idx_s, sPx, sPy, sPz);
  };

  while (true) {
    const int n_free = __builtin_popcountll(__ballot(state != BUSY));
    if (n_free == 64 || (!drained && n_free >= w.fetch_at) ||
        (drained && __builtin_popcountll(__ballot(state == EMPTY || state == DRAIN)) >= WALK_SHARE_AT &&
         __any(state == BUSY && sp >= 1))) {
#ifdef RT_STAMPS
      const unsigned long long st_a = RT_STAMP();
      unsigned long long st_f0 = st_a, st_f1 = st_a;
#endif
      // ---- work off parked leaves, all lanes together ----
      if (!TRI) {
        // spheres: first the cheap screen on every parked leaf (most fail it), then
        // the full test only as often as the busiest lane has passers
        unsigned pass = 0;
        for (int k = 0; __any(k < n_park); k++)
          if (k < n_park && leaf_screen(sc, park[k * 256], a4, ox, oy, oz, dx, dy, dz))
            pass |= 1u << k;
        n_park = 0;
        while (__any(pass != 0)) {
          if (pass != 0) {
            const int slot = park[(__builtin_ctz(pass)) * 256];
            pass &= pass - 1;
            if (!(ANY_HIT && idx_s >= 0) && leaf(slot)) {
              if (ANY_HIT)
                state = DRAIN; // occluded: nothing more to find
              else
                bound = fminf(bound, len_s);
            }
          }
        }
      }
      while (__any(n_park > 0)) {
        if (n_park > 0) {
          const int slot = park[--n_park * 256];
          if (!(ANY_HIT && idx_s >= 0) && leaf(slot)) {
            if (ANY_HIT)
              state = DRAIN; // occluded: nothing more to find
            else
              bound = fminf(bound, len_s);
          }
        }
      }
#ifdef RT_STAMPS
      st_work += RT_STAMP() - st_a;
#endif
      // ---- walks that are over: helpers report to the lane whose ray they share ----
      if (state == DRAIN)
        state = OVER;
      while (true) {
        if (state == OVER && n_help == 0 && owner >= 0)
          state = REPORT;
        const unsigned long long m = __ballot(state == REPORT);
        if (!m)
          break;
        for (unsigned long long rest = m; rest; rest &= rest - 1) {
          const int h = __builtin_ctzll(rest);
          const int o = __builtin_amdgcn_readlane(owner, h);
          const float hl = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(len_s), h));
          const int hi = __builtin_amdgcn_readlane(idx_s, h);
          const float hx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(sPx), h)),
                      hy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(sPy), h)),
                      hz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(sPz), h));
          if (lane == o) {
            if (hi >= 0 && (idx_s < 0 || hl < len_s || (hl == len_s && hi < idx_s))) {
              len_s = hl;
              idx_s = hi;
              sPx = hx;
              sPy = hy;
              sPz = hz;
            }
            n_help--;
          }
        }
        if (state == REPORT) {
          state = EMPTY;
          owner = -1;
        }
      }
      // ---- hand in finished items ----
      if (state == OVER && n_help == 0) {
        if (ANY_HIT) {
          if (!w.last_stage) {
            if (idx_s >= 0) // blocked: cancel the item for the walk that follows
              w.in.items[2 * (size_t)ray + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
          } else if (idx_s < 0) { // unoccluded: renderer.cpp:119-127, as an integer sum
            add_light(w.extra, tagA, tagB);
          }
        } else {
          w.hits.rec[ray] = make_float4(sPx, sPy, sPz, len_s);
          w.hits.idx[ray] = idx_s;
        }
        state = EMPTY;
      }
      // ---- fetch ----
      if (!drained) {
        const unsigned long long e = __ballot(state == EMPTY);
        const int n_e = __builtin_popcountll(e);
        if (n_e) {
          const int leader = __builtin_ctzll(e);
          uint32_t base = 0;
#ifdef RT_STAMPS
          st_f0 = RT_STAMP();
#endif
          if (lane == leader)
            base = atomicAdd(cursor, (uint32_t)n_e);
          base = __builtin_amdgcn_readlane(base, leader);
          const size_t here = seg_base;
          const uint32_t n_here = n_in;
          if (base + (uint32_t)n_e >= n_in) {
            // This segment is used up: on to the next one that still has items to
            // hand out (lanes it could not serve are served at the next refill), or
            // nothing is left.  The lanes look at 64 segments at a time; a cursor
            // read here may be behind, which only costs a useless attempt later.
            // (every wave starts its search somewhere else: waves that all went to the
            // next segment would use that one up together and move on as a crowd, all
            // on one cursor)
            bool found = false;
            const uint32_t hop = 1u + (uint32_t)((blockIdx.x * 4u + (threadIdx.x >> 6)) * 37u) %
                                          (QUEUE_SEGMENTS - 1);
            for (uint32_t round = 0; round < QUEUE_SEGMENTS / 64 && !found; round++) {
              const uint32_t s = (seg + hop + round * 64 + (uint32_t)lane) % QUEUE_SEGMENTS;
              uint32_t n = w.in.count[s * CTL_WORDS];
              if (n > w.in.seg_capacity)
                n = w.in.seg_capacity;
              const uint32_t c = __hip_atomic_load(w.cursor + s * CTL_WORDS, __ATOMIC_RELAXED,
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
              seg_base = (size_t)seg * w.in.seg_capacity;
              cursor = w.cursor + seg * CTL_WORDS;
            } else {
              drained = true;
            }
          }
          if (state == EMPTY) {
            const uint32_t r = base + (uint32_t)__builtin_popcountll(e & ((1ull << lane) - 1ull));
            if (r < n_here) {
              const size_t seg_base = here; // of the segment this item came from
              const float4 qa = w.in.items[2 * (seg_base + r)],
                           qb = w.in.items[2 * (seg_base + r) + 1];
#ifdef RT_STAMPS
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              st_f1 = RT_STAMP();
#endif
              ray = (uint32_t)(seg_base + r);
              ox = qa.x;
              oy = qa.y;
              oz = qa.z;
              tagA = __float_as_uint(qa.w);
              dx = qb.x;
              dy = qb.y;
              dz = qb.z;
              tagB = __float_as_uint(qb.w);
              n_rays++;
              len_s = RT_INF;
              idx_s = -1;
              bound = RT_INF;
              sp = 0;
              node = 0;
              n_park = 0;
              state = BUSY;
              if (TRI) {
                const TriWalkRay tr = tri_walk_ray(ox, oy, oz, dx, dy, dz);
                a = tr.a;
                da = tr.da;
                ol = tr.ol;
                if (!tr.possible) {
                  state = DRAIN;
                } else {
                  bool found = false;
                  for (int k = 0; k < sc.n_tloose && !(ANY_HIT && found); k++) {
                    const int i = sc.tloose[k];
                    found |= tri_fold(sc.tri[i], i, ox, oy, oz, dx, dy, dz, len_s, idx_s, sPx, sPy,
                                      sPz);
                  }
                  if (!tr.usable) // out-of-range numbers: this lane tests every triangle
                    for (int slot = 0; slot < sc.n_tleaf && !(ANY_HIT && found); slot++)
                      found |= leaf(slot);
                  if (!tr.usable || sc.n_tleaf == 0 || (ANY_HIT && found))
                    state = DRAIN;
                  else if (!ANY_HIT) {
                    bound = len_s;
                    // a triangle only matters if it is at least as near as the ray's
                    // nearest sphere (renderer.cpp:181; the prune comparison is strict,
                    // so a tie survives)
                    if (w.prior.idx != nullptr && w.prior.idx[seg_base + r] >= 0)
                      bound = fminf(bound, w.prior.rec[seg_base + r].w);
                  }
                }
              } else {
                a = dot3(dx, dy, dz, dx, dy, dz); // maths.cpp:127
                a4 = 4.0f * a;
                const SphWalkRay wr = sph_walk_ray(dx, dy, dz, a);
                ix = wr.ix;
                iy = wr.iy;
                iz = wr.iz;
                da = wr.ida;
                // (Bounding the walk by the looped triangles' nearest hit as well was
                // measured: the tests run on the few lanes that are refilling and cost
                // more than the pruning saves, 9.27 ms against 8.76 ms per frame.)
                const bool sane = a > 9.094947e-13f && a < 1.0995116e12f;
                if (!(a > 0.f) || sc.s_size == 0) {
                  state = DRAIN; // no sphere can be hit (rt_device.h, beam screen notes)
                } else if (!sane || sc.bvh == nullptr) {
                  // out-of-range direction length: this lane tests every sphere
                  for (int i = 0; i < sc.s_size; i++) {
                    const float4 g = sc.sph_geo[i];
                    float Px, Py, Pz;
                    if (sphere_exact(g.x, g.y, g.z, g.w, a, a4, ox, oy, oz, dx, dy, dz, Px, Py,
                                     Pz)) {
                      const float len = length3(Px, Py, Pz);
                      if (len_s > len) {
                        len_s = len;
                        idx_s = i;
                        sPx = Px;
                        sPy = Py;
                        sPz = Pz;
                        if (ANY_HIT)
                          break;
                      }
                    }
                  }
                  state = DRAIN;
                }
              }
            }
          }
        }
      }
      // ---- nothing left to fetch: idle lanes take over parts of the walks in progress ----
      if (drained) {
        const unsigned long long th = __ballot(state == EMPTY);
        const unsigned long long vi = __ballot(state == BUSY && sp >= 1);
        if (th != 0 && vi != 0) {
          const int wave_base = (int)(threadIdx.x & ~63u);
          const unsigned long long lt = (1ull << lane) - 1ull;
          const int my_v = __builtin_popcountll(vi & lt), my_t = __builtin_popcountll(th & lt);
          const bool robbed = state == BUSY && sp >= 1 && my_v < __builtin_popcountll(th);
          const bool thief = state == EMPTY && my_t < __builtin_popcountll(vi);
          if (robbed)
            pair_words[wave_base + my_v] = lane;
          RT_WAVE_SYNC_LDS();
          int v = lane;
          if (thief)
            v = pair_words[wave_base + my_t];
          const float vox = __shfl(ox, v), voy = __shfl(oy, v), voz = __shfl(oz, v);
          const float vdx = __shfl(dx, v), vdy = __shfl(dy, v), vdz = __shfl(dz, v);
          const float va = __shfl(a, v), va4 = __shfl(a4, v), vda = __shfl(da, v), vol = __shfl(ol, v);
          const float vix = __shfl(ix, v), viy = __shfl(iy, v), viz = __shfl(iz, v);
          const float vbound = __shfl(bound, v);
          int taken = 0;
          if (thief)
            taken = stack_words[wave_base + v]; // the oldest entry of that lane's stack
          RT_WAVE_SYNC_LDS();
          if (robbed) {
            const int top = stack.pop(sp);
            if (sp > 0)
              stack.lds[0] = top;
            n_help++;
          }
          if (thief) {
            ox = vox;
            oy = voy;
            oz = voz;
            dx = vdx;
            dy = vdy;
            dz = vdz;
            a = va;
            a4 = va4;
            da = vda;
            ol = vol;
            ix = vix;
            iy = viy;
            iz = viz;
            bound = vbound;
            owner = v;
            n_help = 0;
            len_s = RT_INF;
            idx_s = -1;
            sp = 0;
            n_park = 0;
            node = taken;
            state = BUSY;
          }
        }
      }
#ifdef RT_STAMPS
      st_f1 = __builtin_amdgcn_readfirstlane((int)(st_f1 - st_f0)); // any fetching lane's wait
      st_fetch += st_f1 & 0xffffffffull;
      st_serv += RT_STAMP() - st_a;
      n_serv++;
#endif
      if (!__any(state != EMPTY))
        break;
      continue;
    }

#ifdef RT_STAMPS
    const unsigned long long st_c = RT_STAMP();
    unsigned long long st_d = st_c;
    int st_over = 0;
    n_busy += __builtin_popcountll(__ballot(state == BUSY));
#endif
    if (state == BUSY) {
      // ---- one node visit: the bounds of its (up to four) children ----
      int next = -1;
      float next_near = RT_INF;
      int32_t links[BVH_WIDTH];
      float nears[BVH_WIDTH];
      bool touch[BVH_WIDTH];
      if (TRI) {
        const TriNode N = sc.tbvh[node];
#ifdef RT_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st_d = RT_STAMP();
#endif
#pragma unroll
        for (int c = 0; c < BVH_WIDTH; c++) {
          links[c] = N.link[c];
          nears[c] = RT_INF;
          touch[c] = links[c] != BVH_EMPTY &&
                     tri_touches<!ANY_HIT>(N.s[c], N.ns[c], ox, oy, oz, dx, dy, dz, a, da, ol,
                                           bound, nears[c]);
        }
      } else {
        const BvhNode N = sc.bvh[node];
#ifdef RT_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st_d = RT_STAMP();
#endif
#pragma unroll
        for (int c = 0; c < BVH_WIDTH; c++)
          links[c] = N.link[c];
        // any hit: first entered, first visited; nearest: nearest to the world origin first
        bvh_visit<ANY_HIT>(N, ox, oy, oz, ix, iy, iz, da, bound, touch, nears);
      }
#pragma unroll
      for (int c = 0; c < BVH_WIDTH; c++) {
        const int32_t link = links[c];
        if (touch[c] && link < 0) {
          // a touched leaf is parked (or, with every slot taken, tested now)
          const int slot = (~link) >> 3;
          if (n_park < PARK_SLOTS) {
            park[n_park++ * 256] = slot;
          } else {
#ifdef RT_STAMPS
            st_over = 1;
#endif
            if (leaf(slot)) {
              if (!ANY_HIT)
                bound = fminf(bound, len_s);
            }
          }
        } else if (touch[c]) {
          // visit the child nearest to the world origin next (it tightens
          // `bound`), stack the others
          int push = link;
          if (nears[c] < next_near) {
            push = next;
            next = link;
            next_near = nears[c];
          }
          if (push >= 0)
            stack.push(sp, push);
        }
      }
      if (next < 0 && sp > 0)
        next = stack.pop(sp);
      if (next >= 0 && !(ANY_HIT && idx_s >= 0))
        node = next;
      else
        state = DRAIN;
    }
#ifdef RT_STAMPS
    {
      // a lane that took the branch has the load's stamp; the others did not load
      const unsigned long long now = RT_STAMP();
      const unsigned long long any_d = __builtin_amdgcn_readfirstlane((int)(st_d - st_c));
      st_load += any_d & 0xffffffffull;
      st_visit += now - st_c;
      n_visit++;
      if (__any(st_over))
        n_over++;
    }
#endif
  }
#ifdef RT_STAMPS
  if (!ANY_HIT && TRI == STAMPS_OF_TRI && lane == 0) {
    atomicAdd(&g_stamps[0], RT_STAMP() - st_begin);
    atomicAdd(&g_stamps[1], st_serv);
    atomicAdd(&g_stamps[2], st_load);
    atomicAdd(&g_stamps[3], st_visit - st_load);
    atomicAdd(&g_stamps[4], n_serv);
    atomicAdd(&g_stamps[5], n_visit);
    atomicAdd(&g_stamps[6], 1ull);
    atomicAdd(&g_stamps[7], st_fetch);
    atomicAdd(&g_stamps[8], n_over);
    atomicAdd(&g_stamps[9], n_busy);
    atomicAdd(&g_stamps[10], st_work);
  }
#endif
  add_ray_count(w.ray_count, n_rays);
}

// k_shade — one ray of the level per lane: the triangles (renderer.cpp:156-166),
// the choice between nearest triangle and nearest sphere (renderer.cpp:179-192),
// then shading: per light the Phong term and the shadow ray, whose triangle test
// also runs here so that only rays a triangle does not already block go on to
// the sphere walk; finally the next level's rays.
__global__ __launch_bounds__(256) void k_shade(SceneDev sc, ShadeArgs s) {
  const uint32_t seg = my_segment();
  uint32_t n = s.rays.count[seg * CTL_WORDS];
  if (n > s.rays.seg_capacity)
    n = s.rays.seg_capacity;
  const size_t seg_base = (size_t)seg * s.rays.seg_capacity;
  const uint32_t stride = (gridDim.x / QUEUE_SEGMENTS) * blockDim.x;
  uint32_t n_rays = 0;
  for (uint32_t base = (blockIdx.x / QUEUE_SEGMENTS) * blockDim.x + (threadIdx.x & ~63u); base < n;
       base += stride) {
    const uint32_t i = base + (threadIdx.x & 63);
    const bool live = i < n;
    Hit h;
    h.kind = 0;
    h.index = -1;
    h.Px = h.Py = h.Pz = 0;
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, wgt = 0;
    uint32_t pixel = 0, depth = 0;
    if (live) {
      const float4 qa = s.rays.items[2 * (seg_base + i)], qb = s.rays.items[2 * (seg_base + i) + 1];
      ox = qa.x;
      oy = qa.y;
      oz = qa.z;
      wgt = qa.w;
      dx = qb.x;
      dy = qb.y;
      dz = qb.z;
      const uint32_t tag = __float_as_uint(qb.w);
      pixel = tag >> 6;
      depth = tag & 63u;
      const float4 sh = s.hits.rec[seg_base + i];
      const int idx_s = s.hits.idx[seg_base + i];
      const float len_s = idx_s >= 0 ? sh.w : RT_INF;
      float len_t = RT_INF, tPx = 0, tPy = 0, tPz = 0;
      int idx_t = -1;
      if (sc.tbvh != nullptr) { // found by k_walk<false, true>
        idx_t = s.tri_hits.idx[seg_base + i];
        if (idx_t >= 0) {
          const float4 th = s.tri_hits.rec[seg_base + i];
          tPx = th.x;
          tPy = th.y;
          tPz = th.z;
          len_t = th.w;
        }
      } else {
        lane_triangles(sc, false, ox, oy, oz, dx, dy, dz, len_t, idx_t, tPx, tPy, tPz);
      }
      if (idx_t >= 0 || idx_s >= 0) { // renderer.cpp:179-192
        if (len_t <= len_s) {
          h.kind = 1;
          h.index = idx_t;
          h.Px = tPx;
          h.Py = tPy;
          h.Pz = tPz;
        } else if (len_t > len_s) {
          h.kind = 2;
          h.index = idx_s;
          h.Px = sh.x;
          h.Py = sh.y;
          h.Pz = sh.z;
        }
      }
    }
    const bool hit = live && h.kind != 0;
    if (!__any(hit))
      continue;
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
      for (int l = s.light_begin; l < s.light_end; l++) {
        float fColor[3] = {0, 0, 0};
        float lx = 0, ly = 0, lz = 0, angle = 0;
        if (lit)
          angle = phong_light(sf, h, dx, dy, dz, sc.lights[l], fColor, lx, ly, lz);
        const uint32_t c0 = (uint32_t)((fColor[0] * angle) * wl),
                       c1 = (uint32_t)((fColor[1] * angle) * wl),
                       c2 = (uint32_t)((fColor[2] * angle) * wl);
        // the shadow item: tagA = pixel, tagB = the three integer contributions
        // (each <= 255), 10 bits apiece
        const uint32_t packed = (c0 & 0x3ffu) | ((c1 & 0x3ffu) << 10) | ((c2 & 0x3ffu) << 20);
        // a light that cannot add one unit to any channel casts no shadow ray
        const bool cast = lit && packed != 0;
        bool blocked = false;
        // renderer.cpp:119: the shadow ray starts exactly at P; its triangles
        // here, unless a hierarchy walk over the triangles follows
        if (cast && sc.tbvh == nullptr) {
          float len_t = RT_INF, tx, ty, tz;
          int idx_t = -1;
          lane_triangles(sc, true, h.Px, h.Py, h.Pz, lx, ly, lz, len_t, idx_t, tx, ty, tz);
          blocked = idx_t >= 0;
          if (blocked)
            n_rays++; // traced to the end here; the others are counted by the walk
        }
        queue_push_raw(s.shadow, cast && !blocked,
                       make_float4(h.Px, h.Py, h.Pz, __uint_as_float(pixel)),
                       make_float4(lx, ly, lz, __uint_as_float(packed)));
      }
    }
    if (s.spawn) {
      const bool deeper = hit && (int)depth < s.max_depth;
      const bool want_t = deeper && kt > 0.f && (wgt * kt >= RT_MIN_WEIGHT);
      const bool want_r = deeper && kr > 0.f && (wgt * kr >= RT_MIN_WEIGHT);
      if (__any(want_t)) {
        float rx = 0, ry = 0, rz = 0;
        if (want_t)
          refract3(dx, dy, dz, sf.nx, sf.ny, sf.nz, ior, rx, ry, rz);
        queue_push(s.next, want_t, h.Px, h.Py, h.Pz, rx, ry, rz, wgt * kt, pixel, depth + 1);
      }
      if (__any(want_r)) {
        float rx = 0, ry = 0, rz = 0;
        if (want_r) {
          reflect3(dx, dy, dz, sf.nx, sf.ny, sf.nz, rx, ry, rz);
          normalize3(rx, ry, rz);
        }
        queue_push(s.next, want_r, h.Px, h.Py, h.Pz, rx, ry, rz, wgt * kr, pixel, depth + 1);
      }
    }
  }
  add_ray_count(s.ray_count, n_rays);
}

} // namespace rt
