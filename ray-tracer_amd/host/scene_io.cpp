// scene_io.cpp — the reference's scene sources with defined behaviour
// (see include/rt_host.h).  Host only.
#include "../../include/rt_host.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace {

template <typename T> T *copy_out(const std::vector<T> &v) {
  T *p = (T *)malloc(sizeof(T) * (v.empty() ? 1 : v.size()));
  if (p && !v.empty())
    memcpy(p, v.data(), sizeof(T) * v.size());
  return p;
}

} // namespace

extern "C" {

int rt_host_load_scene(const char *path, int blank_mode, rt_host_scene *out,
                       int *bad_line) {
  if (!path || !out)
    return -1;
  memset(out, 0, sizeof *out);
  std::ifstream file(path);
  if (!file.is_open())
    return -1;
  std::vector<float> vt, vs, vl; // 12 / 7 / 3 values per record, as main.cpp:225-227
  std::string line;
  int lineno = 0;
  char last_type = 0;
  bool last_line_ended = false; // the last line read ended in '\n'
  auto phantom = [&]() {
    if (last_type == 's')
      vs.insert(vs.end(), 7, 0.0f);
    else if (last_type == 't')
      vt.insert(vt.end(), 12, 0.0f);
    else if (last_type == 'l')
      vl.insert(vl.end(), 3, 0.0f);
  };
  while (std::getline(file, line)) {
    lineno++;
    last_line_ended = !file.eof();
    if (!line.empty() && line[0] == '/') // main.cpp:232
      continue;
    std::istringstream stream(line);
    char type = 0;
    if (!(stream >> type)) { // blank line
      if (blank_mode == RT_BLANK_PHANTOM)
        phantom();
      continue;
    }
    int want = type == 's' ? 7 : type == 't' ? 12 : type == 'l' ? 3 : 0; // main.cpp:242-263
    std::vector<float> &dst = type == 's' ? vs : type == 't' ? vt : vl;
    if (want)
      last_type = type;
    for (int i = 0; i < want; i++) {
      float val;
      if (!(stream >> val)) {
        if (bad_line)
          *bad_line = lineno;
        return -2;
      }
      dst.push_back(val);
    }
  }
  // The reference loops `while (!file.eof())` (main.cpp:228): after a file whose last
  // byte is '\n' it reads one more, empty, line and appends one more primitive of
  // the last type.
  if (blank_mode == RT_BLANK_PHANTOM && lineno > 0 && last_line_ended)
    phantom();
  const int nt = (int)(vt.size() / 12), ns = (int)(vs.size() / 7), nl = (int)(vl.size() / 3);
  std::vector<float> tris(9 * (size_t)nt), sph(3 * (size_t)ns), rad(ns), lights(3 * (size_t)nl);
  std::vector<uint8_t> ct(3 * (size_t)nt), cs(3 * (size_t)ns);
  for (int i = 0; i < nt; i++) { // main.cpp:277-290
    for (int k = 0; k < 9; k++)
      tris[9 * i + k] = vt[12 * i + k];
    for (int k = 0; k < 3; k++)
      ct[3 * i + k] = (unsigned char)vt[12 * i + 9 + k];
  }
  for (int i = 0; i < ns; i++) { // main.cpp:292-300: radius first, then the centre
    rad[i] = vs[7 * i];
    for (int k = 0; k < 3; k++) {
      sph[3 * i + k] = vs[7 * i + 1 + k];
      cs[3 * i + k] = (unsigned char)vs[7 * i + 4 + k];
    }
  }
  for (int i = 0; i < 3 * nl; i++) // main.cpp:302-306
    lights[i] = vl[i];
  out->tris = copy_out(tris);
  out->color_tri = copy_out(ct);
  out->t_size = nt;
  out->spheres = copy_out(sph);
  out->radius = copy_out(rad);
  out->color_sphere = copy_out(cs);
  out->s_size = ns;
  out->lights = copy_out(lights);
  out->l_size = nl;
  return 0;
}

void rt_host_free_scene(rt_host_scene *s) {
  if (!s)
    return;
  free(s->tris);
  free(s->color_tri);
  free(s->spheres);
  free(s->radius);
  free(s->color_sphere);
  free(s->lights);
  memset(s, 0, sizeof *s);
}

int rt_host_sphere_grid(rt_host_scene *s, int row, int col) {
  if (!s || row < 0 || col < 0)
    return -1;
  const size_t n = (size_t)row * col;
  float *sph = (float *)malloc(sizeof(float) * 3 * (n ? n : 1));
  float *rad = (float *)malloc(sizeof(float) * (n ? n : 1));
  uint8_t *colr = (uint8_t *)malloc(3 * (n ? n : 1));
  if (!sph || !rad || !colr) {
    free(sph);
    free(rad);
    free(colr);
    return -1;
  }
  const float r = 1.0f; // main.cpp:360
  for (int i = 0; i < row; i++)
    for (int j = 0; j < col; j++) { // main.cpp:381-392
      const size_t k = (size_t)i * col + j;
      sph[3 * k + 0] = (j * 2 - col) * r;
      sph[3 * k + 1] = (i * 2 - row) * r;
      sph[3 * k + 2] = -50;
      rad[k] = r;
      colr[3 * k + 0] = 0;
      colr[3 * k + 1] = 255;
      colr[3 * k + 2] = 0;
    }
  free(s->spheres);
  free(s->radius);
  free(s->color_sphere);
  s->spheres = sph;
  s->radius = rad;
  s->color_sphere = colr;
  s->s_size = (int32_t)n;
  return 0;
}

int rt_host_write_ppm(const char *path, const uint8_t *fb, int w, int h) {
  if (!path || !fb || w <= 0 || h <= 0)
    return -1;
  FILE *f = fopen(path, "w");
  if (!f)
    return -1;
  fprintf(f, "P3\n%d %d %d\n", w, h, 255); // main.cpp:159-160
  std::string row;
  row.reserve((size_t)w * 12 + 2);
  char buf[16];
  for (int i = 0; i < h; i++) {
    row.clear();
    for (int j = 0; j < w; j++) { // main.cpp:163-170: bytes 2, 1, 0
      const uint8_t *p = fb + ((size_t)w * i + j) * 4;
      int n = snprintf(buf, sizeof buf, "%d %d %d ", p[2], p[1], p[0]);
      row.append(buf, n);
    }
    row.push_back('\n'); // main.cpp:171
    fwrite(row.data(), 1, row.size(), f);
  }
  return fclose(f) == 0 ? 0 : -1;
}

} // extern "C"
