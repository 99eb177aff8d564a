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
// The arithmetic per ray and per primitive is the same trace described in
// rt_device.h; only the scheduling differs.  Results cannot depend on it: the
// nearest-hit rule is order independent (smallest |P|, lowest index on ties) and
// bounce light is a commutative integer sum.
#pragma once
#include "rt_device.h"
#include "rt_kernels.h"

namespace rt {

constexpr int WALK_FETCH_AT = 16; // refill when this many lanes have nothing to do

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

// An unoccluded bounce hit's light: three integers, 10 bits apiece in `packed`.
__device__ __forceinline__ void add_light(uint32_t *extra, uint32_t pixel, uint32_t packed) {
  const uint32_t c0 = packed & 0x3ffu, c1 = (packed >> 10) & 0x3ffu, c2 = (packed >> 20) & 0x3ffu;
  if (c0)
    atomicAdd(extra + 3 * (size_t)pixel + 0, c0);
  if (c1)
    atomicAdd(extra + 3 * (size_t)pixel + 1, c1);
  if (c2)
    atomicAdd(extra + 3 * (size_t)pixel + 2, c2);
}

template <bool ANY_HIT>
__global__ __launch_bounds__(256) void k_walk(SceneDev sc, WalkArgs w) {
  const int lane = threadIdx.x & 63;
  // this block serves segment (block % n_seg) of the queue together with the
  // other blocks of that residue; they share the segment's cursor
  const uint32_t seg = my_segment();
  uint32_t n_in = w.in.count[seg * CTL_WORDS];
  if (n_in > w.in.seg_capacity)
    n_in = w.in.seg_capacity;
  const size_t seg_base = (size_t)seg * w.in.seg_capacity;
  uint32_t *const cursor = w.cursor + seg * CTL_WORDS;
  uint32_t *const n_hits = ANY_HIT ? nullptr : w.hits.count + seg * CTL_WORDS;
  const size_t hit_base = ANY_HIT ? 0 : (size_t)seg * w.hits.seg_capacity;

  // lane state
  enum { EMPTY = 0, BUSY = 1, FINISHED = 2 };
  int state = EMPTY;
  uint32_t ray = 0, tagA = 0, tagB = 0;
  float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, a = 0, a4 = 0, da = 0;
  float len_t = RT_INF, len_s = RT_INF, bound = RT_INF;
  int idx_t = -1, idx_s = -1;
  float tPx = 0, tPy = 0, tPz = 0, sPx = 0, sPy = 0, sPz = 0;
  int stack[BVH_STACK];
  int sp = 0, node = 0;
  uint32_t n_rays = 0;
  bool drained = false;

  while (true) {
    const unsigned long long not_busy = __ballot(state != BUSY);
    const int n_free = __builtin_popcountll(not_busy);
    if (n_free == 64 || (!drained && n_free >= WALK_FETCH_AT)) {
      // ---- hand in finished rays ----
      if (!ANY_HIT) {
        // renderer.cpp:179-192: nothing / triangle (wins ties) / sphere
        int kind = 0, index = -1;
        float Px = 0, Py = 0, Pz = 0;
        if (state == FINISHED && (idx_t >= 0 || idx_s >= 0)) {
          if (len_t <= len_s) {
            kind = 1;
            index = idx_t;
            Px = tPx;
            Py = tPy;
            Pz = tPz;
          } else if (len_t > len_s) {
            kind = 2;
            index = idx_s;
            Px = sPx;
            Py = sPy;
            Pz = sPz;
          }
        }
        const bool push = kind != 0;
        const unsigned long long m = __ballot(push);
        if (m) {
          const int leader = __builtin_ctzll(m);
          uint32_t base = 0;
          if (lane == leader)
            base = atomicAdd(n_hits, (uint32_t)__builtin_popcountll(m));
          base = __builtin_amdgcn_readlane(base, leader);
          if (push) {
            const uint32_t slot =
                base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
            // a segment of hits is as long as its segment of rays: it cannot overflow
            w.hits.rec[hit_base + slot] =
                make_float4(Px, Py, Pz, __uint_as_float(((uint32_t)kind << 28) | (uint32_t)index));
            w.hits.ray[hit_base + slot] = (uint32_t)(seg_base + ray);
          }
        }
      }
      if (state == FINISHED)
        state = EMPTY;
      // ---- fetch new rays ----
      if (!drained) {
        const unsigned long long e = __ballot(state == EMPTY);
        const int n_e = __builtin_popcountll(e);
        const int leader = __builtin_ctzll(e);
        uint32_t base = 0;
        if (lane == leader)
          base = atomicAdd(cursor, (uint32_t)n_e);
        base = __builtin_amdgcn_readlane(base, leader);
        if (base + (uint32_t)n_e >= n_in)
          drained = true;
        if (state == EMPTY) {
          const uint32_t r = base + (uint32_t)__builtin_popcountll(e & ((1ull << lane) - 1ull));
          if (r < n_in) {
            const float4 qa = w.in.items[2 * (seg_base + r)],
                         qb = w.in.items[2 * (seg_base + r) + 1];
            ray = r;
            ox = qa.x;
            oy = qa.y;
            oz = qa.z;
            tagA = __float_as_uint(qa.w);
            dx = qb.x;
            dy = qb.y;
            dz = qb.z;
            tagB = __float_as_uint(qb.w);
            n_rays++;
            len_t = len_s = RT_INF;
            idx_t = idx_s = -1;
            lane_triangles(sc, ANY_HIT, ox, oy, oz, dx, dy, dz, len_t, idx_t, tPx, tPy, tPz);
            a = dot3(dx, dy, dz, dx, dy, dz); // maths.cpp:127
            a4 = 4.0f * a;
            da = sqrtf(a) * 1.000244140625f;
            bound = ANY_HIT ? RT_INF : len_t;
            sp = 0;
            node = 0;
            state = BUSY;
            const bool sane = a > 9.094947e-13f && a < 1.0995116e12f;
            if ((ANY_HIT && idx_t >= 0) || !(a > 0.f) || sc.s_size == 0) {
              state = FINISHED; // occluded by a triangle / no sphere can be hit
            } else if (!sane || sc.bvh == nullptr) {
              // out-of-range direction length: this lane tests every sphere
              for (int i = 0; i < sc.s_size && state == BUSY; i++) {
                const float4 g = sc.sph_geo[i];
                float Px, Py, Pz;
                if (sphere_exact(g.x, g.y, g.z, g.w, a, a4, ox, oy, oz, dx, dy, dz, Px, Py, Pz)) {
                  const float len = length3(Px, Py, Pz);
                  if (len_s > len) {
                    len_s = len;
                    idx_s = i;
                    sPx = Px;
                    sPy = Py;
                    sPz = Pz;
                    if (ANY_HIT)
                      state = FINISHED;
                  }
                }
              }
              state = FINISHED;
            }
          }
        }
      }
      if (ANY_HIT && state == FINISHED) {
        // shadow ray done before walking: occluded by a triangle, or unoccluded
        // with no sphere able to block it
        if (idx_t < 0 && idx_s < 0) {
          add_light(w.extra, tagA, tagB);
        }
        state = EMPTY;
      }
      if (!__any(state != EMPTY))
        break;
      continue;
    }

    if (state == BUSY) {
      // ---- one node visit (see trace<..., RT_ACCEL_BVH> in rt_device.h) ----
      const BvhNode N = sc.bvh[node];
      int next = -1;
      bool done = false;
      const bool swap = N.near1 < N.near0;
#pragma unroll
      for (int cc = 0; cc < 2; cc++) {
        const bool c = (cc != 0) != swap;
        const float4 s = c ? N.s1 : N.s0;
        const float near = c ? N.near1 : N.near0;
        const int32_t link = c ? N.link1 : N.link0;
        if (link == BVH_EMPTY || !bvh_touches(s, near, ox, oy, oz, dx, dy, dz, a, da, bound))
          continue;
        if (link >= 0) {
          if (next < 0)
            next = link;
          else if (sp < BVH_STACK)
            stack[sp++] = link;
        } else {
          const int v = ~link, first = v >> 3, count = (v & 7) + 1;
          for (int k = 0; k < count; k++) {
            const float4 g = sc.leaf_geo[first + k];
            float Px, Py, Pz;
            if (sphere_exact(g.x, g.y, g.z, g.w, a, a4, ox, oy, oz, dx, dy, dz, Px, Py, Pz)) {
              const float len = length3(Px, Py, Pz);
              const int idx = sc.leaf_idx[first + k];
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
      }
      if (!done) {
        if (next >= 0)
          node = next;
        else if (sp > 0)
          node = stack[--sp];
        else
          done = true;
      }
      if (done) {
        if (ANY_HIT) {
          if (idx_s < 0) { // unoccluded: renderer.cpp:119-127, as an integer sum
            add_light(w.extra, tagA, tagB);
          }
          state = EMPTY;
        } else {
          state = FINISHED;
        }
      }
    }
  }
  add_ray_count(w.ray_count, n_rays);
}

// One hit of a bounce ray per lane: surface, material, then for light `light`
// the Phong term as a shadow-ray item; with spawn set, the next level's rays.
__global__ __launch_bounds__(256) void k_shade(SceneDev sc, ShadeArgs s) {
  const uint32_t seg = my_segment();
  uint32_t n = s.hits.count[seg * CTL_WORDS];
  if (n > s.hits.seg_capacity)
    n = s.hits.seg_capacity;
  const size_t hit_base = (size_t)seg * s.hits.seg_capacity;
  const uint32_t stride = (gridDim.x / QUEUE_SEGMENTS) * blockDim.x;
  for (uint32_t base = (blockIdx.x / QUEUE_SEGMENTS) * blockDim.x + (threadIdx.x & ~63u); base < n;
       base += stride) {
    const uint32_t i = base + (threadIdx.x & 63);
    const bool live = i < n;
    Hit h;
    h.kind = 0;
    h.index = -1;
    h.Px = h.Py = h.Pz = 0;
    float dx = 0, dy = 0, dz = 0, wgt = 0;
    uint32_t pixel = 0, depth = 0;
    if (live) {
      const float4 hr = s.hits.rec[hit_base + i];
      const uint32_t r = s.hits.ray[hit_base + i];
      const float4 qa = s.rays.items[2 * (size_t)r], qb = s.rays.items[2 * (size_t)r + 1];
      const uint32_t ki = __float_as_uint(hr.w);
      h.kind = (int)(ki >> 28);
      h.index = (int)(ki & 0x0fffffffu);
      h.Px = hr.x;
      h.Py = hr.y;
      h.Pz = hr.z;
      dx = qb.x;
      dy = qb.y;
      dz = qb.z;
      wgt = qa.w;
      const uint32_t tag = __float_as_uint(qb.w);
      pixel = tag >> 6;
      depth = tag & 63u;
    }
    Surface sf;
    float kr = 0.f, kt = 0.f, ior = 1.f, wl = 0.f;
    if (live) {
      sf = surface_at(sc, h);
      const float4 m = material_at(sc, h);
      kr = m.x;
      kt = m.y;
      ior = m.z > 0.f ? m.z : 1.0f;
      wl = wgt * ((1.0f - kr) - kt);
    }
    const bool lit = live && wl >= RT_MIN_WEIGHT;
    if (s.light >= 0 && __any(lit)) {
      float fColor[3] = {0, 0, 0};
      float lx = 0, ly = 0, lz = 0, angle = 0;
      if (lit)
        angle = phong_light(sf, h, dx, dy, dz, sc.lights[s.light], fColor, lx, ly, lz);
      const uint32_t c0 = (uint32_t)((fColor[0] * angle) * wl), c1 = (uint32_t)((fColor[1] * angle) * wl),
                     c2 = (uint32_t)((fColor[2] * angle) * wl);
      // the shadow item: origin exactly P (renderer.cpp:119), tagA = pixel,
      // tagB = the three integer contributions (each <= 255), 10 bits apiece
      const uint32_t packed = (c0 & 0x3ffu) | ((c1 & 0x3ffu) << 10) | ((c2 & 0x3ffu) << 20);
      queue_push_raw(s.shadow, lit, make_float4(h.Px, h.Py, h.Pz, __uint_as_float(pixel)),
                     make_float4(lx, ly, lz, __uint_as_float(packed)));
    }
    if (s.spawn) {
      const bool deeper = live && (int)depth < s.max_depth;
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
}

} // namespace rt
