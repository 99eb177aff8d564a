/*
 * rt_host.h — host-side helpers around the GPU path: the reference's scene
 * sources restated with defined behaviour.  Plain C ABI, no GPU needed.
 *
 *   rt_host_load_scene   scene-file reader   (ReadSceneFile, main.cpp:215-307;
 *                                             format scene.txt:1-3)
 *   rt_host_sphere_grid  grid generator      (init_spheres, main.cpp:358-395)
 *   rt_host_write_ppm    P3 writer           (main.cpp:155-173)
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_host_scene {
  float *tris;             /* 9*t_size */
  uint8_t *color_tri;      /* 3*t_size */
  int32_t t_size;
  float *spheres;          /* 3*s_size */
  float *radius;           /* s_size */
  uint8_t *color_sphere;   /* 3*s_size */
  int32_t s_size;
  float *lights;           /* 3*l_size */
  int32_t l_size;
} rt_host_scene;

/* Blank lines.  The reference reads an uninitialised `type` on a blank line
 * (main.cpp:232-237); what usually happens is that the previous line's type is
 * reused and a primitive of unread values is appended.
 *   RT_BLANK_SKIP     blank lines are ignored (defined behaviour, default)
 *   RT_BLANK_PHANTOM  a blank line appends an all-zero primitive of the type of
 *                     the last parsed line, which reproduces the reference's
 *                     usual counts (8 triangles for the shipped scene.txt);
 *                     all-zero triangles are degenerate and can never be hit.
 *                     A file that ends in a newline gets one more such primitive:
 *                     the reference's `while (!file.eof())` (main.cpp:228) reads
 *                     an empty line after it                                  */
enum { RT_BLANK_SKIP = 0, RT_BLANK_PHANTOM = 1 };

/* Returns 0, or -1 (cannot open), -2 (malformed line; *bad_line = 1-based line
 * number).  Lines whose first character is '/' are comments (main.cpp:232).  */
int rt_host_load_scene(const char *path, int blank_mode, rt_host_scene *out,
                       int *bad_line);
void rt_host_free_scene(rt_host_scene *scene);

/* Replaces the spheres of `scene` by a row x col grid of unit spheres at
 * ((2j-col), (2i-row), -50), colour (0,255,0), as main.cpp:101 does.           */
int rt_host_sphere_grid(rt_host_scene *scene, int row, int col);

/* "P3\nW H 255\n", then per row "b2 b1 b0 " per pixel (bytes 2,1,0 of the
 * 4-byte pixel, main.cpp:165-169) and a newline.                              */
int rt_host_write_ppm(const char *path, const uint8_t *fb, int w, int h);

#ifdef __cplusplus
}
#endif
#endif
