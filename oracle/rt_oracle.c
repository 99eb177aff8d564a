/*
 * rt_oracle.c — CPU restatement of the reference's trace/shade path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker the HIP path is compared
 * against (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  The
 * shipped library never links, loads or calls it.
 *
 * It restates, in plain C with run-time image size, what
 * /root/reference/renderer.cpp and /root/reference/maths.cpp compute; every
 * function names the reference lines it follows.  It must be compiled with
 * -ffp-contract=off and without -ffast-math: the reference image is a function
 * of unfused IEEE-754 binary32/binary64 arithmetic in source order.
 *
 * Pinning: tests/test_oracle_pin.py checks this file bit-for-bit against the
 * reference itself (oracle/_ref/libref.so, built from the reference sources by
 * oracle/Makefile) and against the fixtures in tests/golden/ generated from
 * that build by oracle/gen_golden.py.  That covers max_depth = 1.  For
 * max_depth > 1 (material bounces, an extension the reference does not have)
 * this file is the definition and PARITY IS UNPINNED.
 *
 * C/C++ difference that matters: the reference is C++ where sqrt(float) and
 * fabs(float) are the float overloads; here sqrtf/fabsf are spelled out.
 */
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---- maths.cpp ------------------------------------------------------------ */

/* maths.cpp:23-25 */
float oracle_length(const float v[3]) {
  return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
}

/* maths.cpp:27-37 — divides per component; zero vector left alone */
void oracle_normalize(float v[3]) {
  float l = oracle_length(v);
  if (l == 0.0)
    return;
  v[0] /= l;
  v[1] /= l;
  v[2] /= l;
}

/* maths.cpp:54-56 */
float oracle_dot(const float a[3], const float b[3]) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

/* maths.cpp:48-52 */
static void cross3(const float a[3], const float b[3], float r[3]) {
  r[0] = a[1] * b[2] - a[2] * b[1];
  r[1] = a[2] * b[0] - a[0] * b[2];
  r[2] = a[0] * b[1] - a[1] * b[0];
}

static void sub3(const float a[3], const float b[3], float r[3]) {
  r[0] = a[0] - b[0]; /* maths.cpp:17-21 */
  r[1] = a[1] - b[1];
  r[2] = a[2] - b[2];
}

/* maths.cpp:39-46 */
void oracle_normal(const float p1[3], const float p2[3], const float p3[3],
                   float res[3]) {
  float a[3], b[3];
  sub3(p2, p1, a);
  sub3(p3, p1, b);
  cross3(a, b, res);
  oracle_normalize(res);
}

/* maths.cpp:163-168 : refl = n * (-2 * (d.n)) + d */
void oracle_reflect(const float dir[3], const float n[3], float refl[3]) {
  float dot = oracle_dot(dir, n);
  float k = -2 * dot;
  refl[0] = n[0] * k + dir[0];
  refl[1] = n[1] * k + dir[1];
  refl[2] = n[2] * k + dir[2];
}

/* maths.cpp:64-119.  Note the sign convention of t at :78 (N.orig + d, not
 * d - N.orig) and that P is written before the edge tests (:87). */
int oracle_ray_triangle(const float orig[3], const float dir[3],
                        const float p1[3], const float p2[3],
                        const float p3[3], float P[3]) {
  const float eps = 0.0001;
  float N[3];
  oracle_normal(p1, p2, p3, N);
  float nd = oracle_dot(N, dir);
  if (fabsf(nd) < eps)
    return 0;
  float d = oracle_dot(N, p1);
  float t = (oracle_dot(N, orig) + d) / nd;
  if (t < 0)
    return 0;
  P[0] = orig[0] + dir[0] * t;
  P[1] = orig[1] + dir[1] * t;
  P[2] = orig[2] + dir[2] * t;

  const float *corner[4] = {p1, p2, p3, p1};
  for (int k = 0; k < 3; k++) { /* :91-116, three identical edge tests */
    float edge[3], vp[3], C[3];
    sub3(corner[k + 1], corner[k], edge);
    sub3(P, corner[k], vp);
    cross3(edge, vp, C);
    if (oracle_dot(N, C) < 0)
      return 0;
  }
  return 1;
}

/* maths.cpp:121-161.  The "-0.5 * ..." products are double (the literal is a
 * double) and are rounded to float on assignment. */
int oracle_ray_sphere(const float orig[3], const float dir[3],
                      const float centre[3], float radius, float P[3]) {
  const float eps = 0.001;
  float t0, t1;
  float L[3];
  sub3(orig, centre, L);
  float a = oracle_dot(dir, dir);
  float b = 2 * oracle_dot(dir, L);
  float c = oracle_dot(L, L) - radius * radius;
  float discr = b * b - 4 * a * c;
  if (discr < 0) {
    return 0;
  } else if (discr == 0) {
    t0 = t1 = (float)(-0.5 * (double)b / (double)a);
  } else {
    float q = (b > eps) ? (float)(-0.5 * (double)(b + sqrtf(discr)))
                        : (float)(-0.5 * (double)(b - sqrtf(discr)));
    t0 = q / a;
    t1 = c / q;
  }
  if (t0 >= t1) {
    float tmp = t1;
    t1 = t0;
    t0 = tmp;
  }
  if (t0 <= eps) {
    t0 = t1;
    if (t0 <= eps)
      return 0;
  }
  P[0] = orig[0] + dir[0] * t0;
  P[1] = orig[1] + dir[1] * t0;
  P[2] = orig[2] + dir[2] * t0;
  return 1;
}

/* ---- renderer.cpp:145-193 -------------------------------------------------
 * Nearest hit by |P| measured from the WORLD origin, strict '>' so the lowest
 * index wins ties inside a class, '<=' so a triangle wins a tie with a sphere.
 * *index is left untouched on a miss, and P is scratch on a miss. */
int oracle_check_intersection(const float *tris, int t_size,
                              const float *spheres, const float *radius,
                              int s_size, float P[3], int *index,
                              const float orig[3], const float dir[3]) {
  int index_tri = 0, index_s = 0;
  int t_hit = 0, s_hit = 0;
  float p_tri[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  float p_sph[3] = {FLT_MAX, FLT_MAX, FLT_MAX};

  for (int i = 0; i < t_size; i++) {
    if (oracle_ray_triangle(orig, dir, tris + i * 9, tris + i * 9 + 3,
                            tris + i * 9 + 6, P)) {
      if (oracle_length(p_tri) > oracle_length(P)) {
        index_tri = i;
        t_hit = 1;
        memcpy(p_tri, P, sizeof p_tri);
      }
    }
  }
  for (int i = 0; i < s_size; i++) {
    if (oracle_ray_sphere(orig, dir, spheres + i * 3, radius[i], P)) {
      if (oracle_length(p_sph) > oracle_length(P)) {
        index_s = i;
        s_hit = 1;
        memcpy(p_sph, P, sizeof p_sph);
      }
    }
  }
  if (!t_hit && !s_hit)
    return 0;
  if (oracle_length(p_tri) <= oracle_length(p_sph)) {
    memcpy(P, p_tri, sizeof p_tri);
    *index = index_tri;
    return 1;
  }
  if (oracle_length(p_tri) > oracle_length(p_sph)) {
    memcpy(P, p_sph, sizeof p_sph);
    *index = index_s;
    return 2;
  }
  return 0; /* NaN lengths, renderer.cpp:192 */
}

/* n independent queries; index[i] is preset to -1 so "untouched" shows. */
void oracle_check_intersection_batch(const scene_desc *sc, long n,
                                     const float *orig, const float *dir,
                                     int32_t *kind, int32_t *index, float *P) {
#pragma omp parallel for schedule(dynamic, 64)
  for (long i = 0; i < n; i++) {
    int idx = -1;
    kind[i] = oracle_check_intersection(sc->tris, sc->t_size, sc->spheres,
                                        sc->radius, sc->s_size, P + 3 * i, &idx,
                                        orig + 3 * i, dir + 3 * i);
    index[i] = idx;
  }
}

/* ---- renderer.cpp:11,34-44 -------------------------------------------------
 * px,py are evaluated in double (j + 0.5 is double) and rounded once to float;
 * the ray goes from the eye through (px,py,plane_z) in WORLD space. */
void oracle_primary_ray(const camera *cam, int w, int h, int i, int j,
                        float orig[3], float dir[3]) {
  float aspect = (float)w / (float)h;
  float px = (2 * ((j + 0.5) / (float)w) - 1) *
             tan(cam->fov_deg * 0.5 * M_PI / 180) * aspect;
  float py = (1 - 2 * ((i + 0.5) / (float)h)) *
             tan(cam->fov_deg * 0.5 * M_PI / 180);
  float rayP[3] = {px, py, cam->plane_z};
  orig[0] = cam->origin[0];
  orig[1] = cam->origin[1];
  orig[2] = cam->origin[2];
  sub3(rayP, orig, dir);
  oracle_normalize(dir);
}

/* renderer.cpp:121-126 with clamp<uint16_t> of maths.hpp:26-28 */
static uint8_t accumulate_u8(uint8_t old, float add) {
  uint16_t v = (uint16_t)(old + add);
  return (uint8_t)(v > 255 ? 255 : v);
}

/* ---- shading of one hit, renderer.cpp:62-128 -------------------------------
 * kind/index/P come from check_intersection with ray direction `dir`.
 * Outputs the unit normal, and for every unshadowed light calls sink().      */
typedef struct hit_info {
  int kind, index;
  float P[3], n[3];
  uint8_t color[3];
} hit_info;

static void hit_normal_color(const scene_desc *sc, hit_info *hi) {
  if (hi->kind == 1) { /* renderer.cpp:66-70 */
    const float *t = sc->tris + hi->index * 9;
    oracle_normal(t, t + 3, t + 6, hi->n);
    memcpy(hi->color, sc->color_tri + hi->index * 3, 3);
  } else { /* renderer.cpp:71-74 */
    sub3(hi->P, sc->spheres + hi->index * 3, hi->n);
    memcpy(hi->color, sc->color_sphere + hi->index * 3, 3);
  }
  oracle_normalize(hi->n); /* renderer.cpp:76 */
}

/* The specular term on its own, renderer.cpp:106-110 (known-answer door; the
 * same two statements as in phong_light below). */
void oracle_specular_batch(long n, const float *lang, const uint8_t *color,
                           float *strength, uint8_t *specular) {
  for (long i = 0; i < n; i++) {
    float s = (float)pow((double)lang[i], 20.0); /* renderer.cpp:106 */
    strength[i] = s;
    specular[i] = (uint8_t)(color[i] * s);       /* :108-110 */
  }
}

/* One light's Phong term, renderer.cpp:85-115.  Returns angle, fills fColor
 * and the unit direction to the light. */
static float phong_light(const hit_info *hi, const float dir[3],
                         const float light[3], float fColor[3],
                         float rayDir[3]) {
  sub3(light, hi->P, rayDir);
  oracle_normalize(rayDir);
  float angle = oracle_dot(hi->n, rayDir);
  angle = angle < 0 ? 0 : angle;

  float zero[3] = {0, 0, 0};
  float ml[3], r[3], ndir[3];
  sub3(zero, rayDir, ml);
  oracle_reflect(ml, hi->n, r);
  sub3(zero, dir, ndir);
  float lang = oracle_dot(r, ndir);
  lang = lang < 0 ? 0 : lang;
  float s = (float)pow((double)lang, 20.0); /* renderer.cpp:106 */

  for (int k = 0; k < 3; k++) {
    uint8_t spec = (uint8_t)(hi->color[k] * s);            /* :108-110 */
    fColor[k] = (float)(spec * 0.3 + hi->color[k] * 0.7);  /* :113-115 */
  }
  return angle;
}

/* ---- the bounce extension (NOT in the reference; parity unpinned) ----------
 * Active only when max_depth > 1.  Then a hit on a primitive with material
 * (kr, kt, ior) at path weight w:
 *   - local Phong light is weighted by wl = w * ((1 - kr) - kt); below the
 *     primary hit a light whose three integer contributions trunc(fColor*angle*wl)
 *     are all zero casts no shadow ray (the ray could not change the pixel)
 *   - if depth < max_depth and w*kr >= 2^-8: a mirror ray from P along
 *     normalize(reflect(dir, n)) with weight w*kr
 *   - if depth < max_depth and w*kt >= 2^-8: a refracted ray (Snell, ior;
 *     total internal reflection falls back to the mirror direction) with
 *     weight w*kt
 * The primary hit accumulates into the pixel exactly as renderer.cpp:121-126.
 * Every later hit adds trunc(fColor*angle*wl) as an integer; the pixel is the
 * saturating sum, which does not depend on the order rays are processed in.
 * Anything lighter than 2^-8 cannot reach 1 LSB, so dropping it is exact.
 * Secondary rays start exactly at P like the reference's shadow rays
 * (renderer.cpp:119), relying on the same epsilons for self-hits.           */
#define RT_MIN_WEIGHT 0.00390625f

typedef struct ray_item {
  float o[3], d[3], w;
  int depth;
} ray_item;

static const rt_material zero_mat = {0, 0, 0, 0};

static const rt_material *hit_material(const scene_desc *sc, int kind,
                                       int index) {
  if (kind == 1)
    return sc->mat_tri ? sc->mat_tri + index : &zero_mat;
  return sc->mat_sphere ? sc->mat_sphere + index : &zero_mat;
}

void oracle_refract(const float dir[3], const float n[3], float ior,
                    float out[3]) {
  float cosi = -oracle_dot(dir, n);
  float nn[3] = {n[0], n[1], n[2]};
  float eta;
  if (cosi < 0) { /* leaving the medium */
    nn[0] = -n[0];
    nn[1] = -n[1];
    nn[2] = -n[2];
    cosi = -cosi;
    eta = ior;
  } else {
    eta = 1.0f / ior;
  }
  float k = 1.0f - (eta * eta) * (1.0f - cosi * cosi);
  if (k < 0) {
    oracle_reflect(dir, n, out);
  } else {
    float f = eta * cosi - sqrtf(k);
    out[0] = eta * dir[0] + f * nn[0];
    out[1] = eta * dir[1] + f * nn[1];
    out[2] = eta * dir[2] + f * nn[2];
  }
  oracle_normalize(out);
}

static uint64_t shade_pixel(const scene_desc *sc, const camera *cam, int w,
                            int h, int i, int j, int max_depth, uint8_t *px,
                            int32_t *id_out) {
  uint64_t rays = 0;
  uint32_t extra[3] = {0, 0, 0};
  ray_item stack[64];
  int sp = 0;

  px[0] = px[1] = px[2] = 0; /* renderer.cpp:50-53 */
  px[3] = (uint8_t)-1;

  oracle_primary_ray(cam, w, h, i, j, stack[0].o, stack[0].d);
  stack[0].w = 1.0f;
  stack[0].depth = 1;
  sp = 1;

  while (sp > 0) {
    ray_item r = stack[--sp];
    hit_info hi;
    int index = -1;
    hi.kind = oracle_check_intersection(sc->tris, sc->t_size, sc->spheres,
                                        sc->radius, sc->s_size, hi.P, &index,
                                        r.o, r.d);
    hi.index = index;
    rays++;
    if (r.depth == 1 && id_out)
      *id_out = hi.kind == 0 ? -1
                             : (hi.kind == 1 ? index : sc->t_size + index);
    if (hi.kind == 0)
      continue;
    hit_normal_color(sc, &hi);
    if (max_depth == 0) { /* the UNLIT build, renderer.cpp:130-137 */
      for (int k = 0; k < 3; k++) {
        int v = px[k] + hi.color[k];
        px[k] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
      }
      continue;
    }

    /* max_depth = 1 is the reference: materials play no part at all */
    const rt_material *m =
        max_depth > 1 ? hit_material(sc, hi.kind, hi.index) : &zero_mat;
    float kr = m->reflectivity, kt = m->transparency;
    float wl = r.w * ((1.0f - kr) - kt);

    if (r.depth == 1 || wl >= RT_MIN_WEIGHT) {
      for (int l = 0; l < sc->l_size; l++) { /* renderer.cpp:82 */
        float fColor[3], rayDir[3], P1[3];
        int dummy;
        float angle = phong_light(&hi, r.d, sc->lights + l * 3, fColor, rayDir);
        /* extension: below the primary hit a light that cannot add one unit to
         * any channel casts no shadow ray (nothing could be added whatever the
         * ray finds); the primary hit always casts it, renderer.cpp:119 */
        if (r.depth > 1 && (uint32_t)((fColor[0] * angle) * wl) == 0 &&
            (uint32_t)((fColor[1] * angle) * wl) == 0 &&
            (uint32_t)((fColor[2] * angle) * wl) == 0)
          continue;
        rays++;
        if (oracle_check_intersection(sc->tris, sc->t_size, sc->spheres,
                                      sc->radius, sc->s_size, P1, &dummy, hi.P,
                                      rayDir) == 0) { /* renderer.cpp:119 */
          for (int k = 0; k < 3; k++) {
            float add = (fColor[k] * angle) * wl;
            if (r.depth == 1)
              px[k] = accumulate_u8(px[k], add);
            else
              extra[k] += (uint32_t)add;
          }
        }
      }
    }

    if (r.depth < max_depth && sp + 2 <= 64) {
      if (kt > 0 && r.w * kt >= RT_MIN_WEIGHT) {
        ray_item *c = &stack[sp++];
        memcpy(c->o, hi.P, sizeof c->o);
        oracle_refract(r.d, hi.n, m->ior > 0 ? m->ior : 1.0f, c->d);
        c->w = r.w * kt;
        c->depth = r.depth + 1;
      }
      if (kr > 0 && r.w * kr >= RT_MIN_WEIGHT) {
        ray_item *c = &stack[sp++];
        memcpy(c->o, hi.P, sizeof c->o);
        oracle_reflect(r.d, hi.n, c->d);
        oracle_normalize(c->d);
        c->w = r.w * kr;
        c->depth = r.depth + 1;
      }
    }
  }
  for (int k = 0; k < 3; k++) {
    uint32_t v = px[k] + extra[k];
    px[k] = (uint8_t)(v > 255 ? 255 : v);
  }
  return rays;
}

/* renderer.cpp:27-31: rows in parallel, dynamic schedule, columns serial.
 * Renders rows row_begin, row_begin+row_step, ... < row_end of a w x h frame
 * into fb (whole-frame addressing).  Returns the rays traced. */
uint64_t oracle_render(const scene_desc *sc, const camera *cam, uint8_t *fb,
                       int32_t *ids, int w, int h, int max_depth,
                       int row_begin, int row_end, int row_step) {
  uint64_t total = 0;
  if (row_step < 1)
    row_step = 1;
  int nrows = row_end > row_begin ? (row_end - row_begin + row_step - 1) / row_step : 0;
#pragma omp parallel for schedule(dynamic) reduction(+ : total)
  for (int r = 0; r < nrows; r++) {
    int i = row_begin + r * row_step;
    for (int j = 0; j < w; j++) {
      size_t p = (size_t)w * i + j;
      total += shade_pixel(sc, cam, w, h, i, j, max_depth, fb + 4 * p,
                           ids ? ids + p : NULL);
    }
  }
  return total;
}

/* The reference compiled with -DUNLIT=1 (renderer.cpp:78,130-137): the hit
 * primitive's colour, no lights, no shadow rays.  Same row selection as
 * oracle_render. */
uint64_t oracle_render_unlit(const scene_desc *sc, const camera *cam, uint8_t *fb,
                             int32_t *ids, int w, int h, int row_begin, int row_end,
                             int row_step) {
  return oracle_render(sc, cam, fb, ids, w, h, 0, row_begin, row_end, row_step);
}

/* The same per-pixel work for a list of n pixels (ij = row, column pairs) of a
 * w x h frame: out_rgba gets 4 bytes and out_ids (may be NULL) one id per listed
 * pixel.  Lets a sample of a frame be checked when the whole frame would take
 * hours on a CPU.  Returns the rays traced. */
uint64_t oracle_render_pixels(const scene_desc *sc, const camera *cam, uint8_t *out_rgba,
                              int32_t *out_ids, int w, int h, int max_depth, int64_t n,
                              const int32_t *ij) {
  uint64_t total = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : total)
  for (int64_t k = 0; k < n; k++)
    total += shade_pixel(sc, cam, w, h, ij[2 * k], ij[2 * k + 1], max_depth, out_rgba + 4 * k,
                         out_ids ? out_ids + k : NULL);
  return total;
}
