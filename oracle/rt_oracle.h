/*
 * rt_oracle.h — entry points of the CPU checker (see rt_oracle.c).
 * TEST INFRASTRUCTURE ONLY: nothing in the shipped library includes this.
 * scene_desc / camera / rt_material are the public ABI structs.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "../include/rt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

float oracle_length(const float v[3]);
void oracle_normalize(float v[3]);
float oracle_dot(const float a[3], const float b[3]);
void oracle_normal(const float p1[3], const float p2[3], const float p3[3],
                   float res[3]);
void oracle_reflect(const float dir[3], const float n[3], float refl[3]);
void oracle_refract(const float dir[3], const float n[3], float ior,
                    float out[3]);
int oracle_ray_triangle(const float orig[3], const float dir[3],
                        const float p1[3], const float p2[3],
                        const float p3[3], float P[3]);
int oracle_ray_sphere(const float orig[3], const float dir[3],
                      const float centre[3], float radius, float P[3]);
int oracle_check_intersection(const float *tris, int t_size,
                              const float *spheres, const float *radius,
                              int s_size, float P[3], int *index,
                              const float orig[3], const float dir[3]);
void oracle_check_intersection_batch(const scene_desc *sc, long n,
                                     const float *orig, const float *dir,
                                     int32_t *kind, int32_t *index, float *P);
void oracle_specular_batch(long n, const float *lang, const uint8_t *color,
                           float *strength, uint8_t *specular);
void oracle_primary_ray(const camera *cam, int w, int h, int i, int j,
                        float orig[3], float dir[3]);
uint64_t oracle_render(const scene_desc *sc, const camera *cam, uint8_t *fb,
                       int32_t *ids, int w, int h, int max_depth,
                       int row_begin, int row_end, int row_step);
uint64_t oracle_render_unlit(const scene_desc *sc, const camera *cam, uint8_t *fb,
                             int32_t *ids, int w, int h, int row_begin, int row_end,
                             int row_step);
uint64_t oracle_render_pixels(const scene_desc *sc, const camera *cam, uint8_t *out_rgba,
                              int32_t *out_ids, int w, int h, int max_depth, int64_t n,
                              const int32_t *ij);

#ifdef __cplusplus
}
#endif
#endif
