// ref_harness.cpp — C-callable doorway into the REAL reference code.
//
// TEST INFRASTRUCTURE ONLY.  oracle/Makefile compiles this file together with
// the reference's own maths.cpp / renderer.cpp / main.cpp, taken where they lie
// under /root/reference (never copied), into oracle/_ref/libref.so.  Nothing
// here computes anything itself except the per-pixel primary ray used to ask
// the reference's check_intersection which primitive a pixel sees
// (renderer.cpp:34-44 builds that ray inline, so there is no function to call).
//
// The prototypes come from the reference's own headers via -I/root/reference.
#include "renderer.hpp" // reference header: render, check_intersection, CANVAS_*
#include "maths.hpp"    // reference header: vector helpers and the two ray tests

#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>

// Defined (non-static) in the reference's main.cpp:215 and main.cpp:358.
void ReadSceneFile(std::string path, float **tris, unsigned char **t_colors,
                   float **spheres, float **radius, unsigned char **s_colors,
                   float **lights, int &t_size, int &s_size, int &l_size);
int init_spheres(float **spheres, float **radius, unsigned char **colors,
                 int row, int col);

extern "C" {

int ref_canvas_width() { return CANVAS_WIDTH; }
int ref_canvas_height() { return CANVAS_HEIGHT; }

// The reference's render(), untouched (renderer.cpp:7).
void ref_render(unsigned char *fb, int fov, float *tris,
                unsigned char *color_tri, int t_size, float *spheres,
                float *radius, unsigned char *color_sphere, int s_size,
                float *lights, int l_size) {
  render(fb, fov, tris, color_tri, t_size, spheres, radius, color_sphere,
         s_size, lights, l_size);
}

// n independent calls of the reference's check_intersection (renderer.cpp:145).
// index[i] is preset to -1 so "left untouched" is visible.
void ref_check_intersection_batch(long n, float *tris, int t_size,
                                  float *spheres, float *radius, int s_size,
                                  float *orig, float *dir, int *kind,
                                  int *index, float *P) {
#pragma omp parallel for schedule(dynamic, 64)
  for (long i = 0; i < n; i++) {
    index[i] = -1;
    kind[i] = check_intersection(tris, t_size, spheres, radius, s_size,
                                 P + 3 * i, index + i, orig + 3 * i,
                                 dir + 3 * i);
  }
}

// Hit-id buffer at the reference's compiled-in resolution: the primary ray of
// renderer.cpp:34-44 (same expressions, same types) fed to check_intersection.
void ref_hit_ids(int fov, float *tris, int t_size, float *spheres,
                 float *radius, int s_size, int *ids) {
  float aspectRatio = (float)CANVAS_WIDTH / (float)CANVAS_HEIGHT;
#pragma omp parallel for schedule(dynamic)
  for (int i = 0; i < CANVAS_HEIGHT; i++) {
    for (int j = 0; j < CANVAS_WIDTH; j++) {
      float px = (2 * ((j + 0.5) / (float)CANVAS_WIDTH) - 1) *
                 tan(fov * 0.5 * M_PI / 180) * aspectRatio;
      float py = (1 - 2 * ((i + 0.5) / (float)CANVAS_HEIGHT)) *
                 tan(fov * 0.5 * M_PI / 180);
      float rayP[3] = {px, py, -1.0};
      float orig[3] = {0.0, 0.0, 1.0};
      float dir[3];
      sub_vec(rayP, orig, dir);
      normalize(dir);
      float P[3];
      int index = -1;
      int k = check_intersection(tris, t_size, spheres, radius, s_size, P,
                                 &index, orig, dir);
      ids[(long)CANVAS_WIDTH * i + j] =
          k == 0 ? -1 : (k == 1 ? index : t_size + index);
    }
  }
}

// Batched known-answer doors onto the reference's math (maths.cpp).
void ref_ray_sphere_batch(long n, float *orig, float *dir, float *centre,
                          float *radius, int *hit, float *P) {
  for (long i = 0; i < n; i++)
    hit[i] = raySphereIntersects(orig + 3 * i, dir + 3 * i, centre + 3 * i,
                                 radius[i], P + 3 * i);
}

void ref_ray_triangle_batch(long n, float *orig, float *dir, float *tri,
                            int *hit, float *P) {
  for (long i = 0; i < n; i++)
    hit[i] = rayTriangleIntersects(orig + 3 * i, dir + 3 * i, tri + 9 * i,
                                   tri + 9 * i + 3, tri + 9 * i + 6, P + 3 * i);
}

void ref_normalize_batch(long n, float *v) {
  for (long i = 0; i < n; i++)
    normalize(v + 3 * i);
}

void ref_normal_batch(long n, float *tri, float *out) {
  for (long i = 0; i < n; i++)
    normal(tri + 9 * i, tri + 9 * i + 3, tri + 9 * i + 6, out + 3 * i);
}

void ref_reflect_batch(long n, float *dir, float *nrm, float *out) {
  for (long i = 0; i < n; i++)
    reflect(dir + 3 * i, nrm + 3 * i, out + 3 * i);
}

void ref_length_batch(long n, float *v, float *out) {
  for (long i = 0; i < n; i++)
    out[i] = length(v + 3 * i);
}

// The specular term of renderer.cpp:106-110.  The reference has no function for
// it (the statements sit inside render()'s light loop), so the two statements are
// restated here, and what these vectors pin is the value the reference's build
// gets from its C library for pow(double, 20.0) at the given arguments, followed
// by the reference's conversions: compiled by the same compiler with the same
// flags and linked against the same libm as renderer.cpp beside it.
void ref_specular_batch(long n, float *lang, unsigned char *color,
                        float *strength, unsigned char *specular) {
  for (long i = 0; i < n; i++) {
    float specular_strength = pow(lang[i], 20); // renderer.cpp:106 (float, int)
    strength[i] = specular_strength;
    specular[i] = color[i] * specular_strength; // renderer.cpp:108-110
  }
}

// The reference's grid generator (main.cpp:358-395).  Caller provides arrays of
// row*col*3, row*col, row*col*3 elements.
int ref_init_spheres(int row, int col, float *spheres, float *radius,
                     unsigned char *colors) {
  float *s, *r;
  unsigned char *c;
  int n = init_spheres(&s, &r, &c, row, col);
  memcpy(spheres, s, sizeof(float) * 3 * n);
  memcpy(radius, r, sizeof(float) * n);
  memcpy(colors, c, 3 * n);
  delete[] s;
  delete[] r;
  delete[] c;
  return n;
}

// The reference's scene-file loader (main.cpp:215-307).  Its behaviour on
// blank lines is undefined (main.cpp:232-237 reads an uninitialised `type`), and
// what it returns can change from one call to the next, so it is called ONCE
// per ref_read_scene and the arrays are kept until ref_read_scene_copy.
static float *g_t, *g_s, *g_r, *g_l;
static unsigned char *g_ct, *g_cs;
static int g_nt, g_ns, g_nl;

int ref_read_scene(const char *path, int *t_size, int *s_size, int *l_size) {
  delete[] g_t;
  delete[] g_ct;
  delete[] g_s;
  delete[] g_r;
  delete[] g_cs;
  delete[] g_l;
  ReadSceneFile(path, &g_t, &g_ct, &g_s, &g_r, &g_cs, &g_l, g_nt, g_ns, g_nl);
  *t_size = g_nt;
  *s_size = g_ns;
  *l_size = g_nl;
  return 0;
}

void ref_read_scene_copy(float *tris, unsigned char *color_tri, float *spheres,
                         float *radius, unsigned char *color_sphere,
                         float *lights) {
  memcpy(tris, g_t, sizeof(float) * 9 * g_nt);
  memcpy(color_tri, g_ct, 3 * g_nt);
  memcpy(spheres, g_s, sizeof(float) * 3 * g_ns);
  memcpy(radius, g_r, sizeof(float) * g_ns);
  memcpy(color_sphere, g_cs, 3 * g_ns);
  memcpy(lights, g_l, sizeof(float) * 3 * g_nl);
}

} // extern "C"
