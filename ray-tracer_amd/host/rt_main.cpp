// rt_main.cpp — headless driver with the reference's command line
// (main.cpp:61-101,155-173) on top of the GPU path:
//   -f <file>  scene file (default scene.txt)      -row N / -col N  sphere grid (default 10)
//   -n         no display: write image.ppm          -h               help
// As in the reference the spheres of the scene file are replaced by the grid
// (main.cpp:98-101).  Additions, all optional: -w/-H frame size (default the
// reference's 2560x1440, renderer.hpp:10-11), -depth D, -o <ppm path>,
// -keep-spheres (use the file's spheres), -phantoms (reference-compatible blank
// lines), -repeat N (call gpu_render N times and report each; the first call of
// a process also pays for the HIP context and code-object load), -pageable (keep
// the frame buffer in ordinary memory, as the reference's new[] at main.cpp:91
// does, instead of taking it from rt_alloc_host_buffer).  There is no viewer here: without -n the frame is rendered and dropped.
#include "../../include/rt_amd.h"
#include "../../include/rt_host.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static const char *opt(int argc, char **argv, const char *name) {
  for (int i = 1; i + 1 < argc; i++)
    if (!strcmp(argv[i], name))
      return argv[i + 1];
  return nullptr;
}
static bool flag(int argc, char **argv, const char *name) {
  for (int i = 1; i < argc; i++)
    if (!strcmp(argv[i], name))
      return true;
  return false;
}

int main(int argc, char **argv) {
  if (flag(argc, argv, "-h")) {
    printf("Ray-Tracer (MI355X path)\n"
           "-f <filename> : Input file to configurate the scene (default: scene.txt)\n"
           "-row          : Row number of the sphere matrix (default: 10)\n"
           "-col          : Column number of the sphere matrix (default: 10)\n"
           "-n            : No display (writes image.ppm)\n"
           "-w/-H/-depth/-o/-keep-spheres/-phantoms : see rt_main.cpp\n"
           "-h            : Print this message\n");
    return 0;
  }
  const char *file = opt(argc, argv, "-f") ? opt(argc, argv, "-f") : "scene.txt";
  const int col = opt(argc, argv, "-col") ? atoi(opt(argc, argv, "-col")) : 10;
  const int row = opt(argc, argv, "-row") ? atoi(opt(argc, argv, "-row")) : 10;
  const int w = opt(argc, argv, "-w") ? atoi(opt(argc, argv, "-w")) : 2560;
  const int h = opt(argc, argv, "-H") ? atoi(opt(argc, argv, "-H")) : 1440;
  const int depth = opt(argc, argv, "-depth") ? atoi(opt(argc, argv, "-depth")) : 1;
  const char *out = opt(argc, argv, "-o") ? opt(argc, argv, "-o") : "image.ppm";
  const int repeat = opt(argc, argv, "-repeat") ? atoi(opt(argc, argv, "-repeat")) : 1;

  rt_host_scene hs;
  int bad = 0;
  int rc = rt_host_load_scene(file, flag(argc, argv, "-phantoms") ? RT_BLANK_PHANTOM : RT_BLANK_SKIP,
                              &hs, &bad);
  if (rc == -1) {
    fprintf(stderr, "Could not open file '%s'\n", file); // main.cpp:222
    memset(&hs, 0, sizeof hs);                           // the reference carries on with an empty scene
  } else if (rc == -2) {
    fprintf(stderr, "%s:%d: malformed line\n", file, bad);
    return 2;
  }
  if (!flag(argc, argv, "-keep-spheres"))
    rt_host_sphere_grid(&hs, row, col); // main.cpp:101

  scene_desc s;
  memset(&s, 0, sizeof s);
  s.tris = hs.tris;
  s.color_tri = hs.color_tri;
  s.t_size = hs.t_size;
  s.spheres = hs.spheres;
  s.radius = hs.radius;
  s.color_sphere = hs.color_sphere;
  s.s_size = hs.s_size;
  s.lights = hs.lights;
  s.l_size = hs.l_size;
  camera cam;
  rt_camera_default(&cam); // fov 90, main.cpp:88

  // the copy back into page-locked memory runs at the rate of the link
  const size_t fb_bytes = (size_t)4 * w * h;
  std::vector<uint8_t> pageable;
  void *locked = nullptr;
  if (flag(argc, argv, "-pageable") || rt_alloc_host_buffer(fb_bytes, &locked) != RT_OK) {
    locked = nullptr;
    pageable.resize(fb_bytes);
  }
  uint8_t *const fb = locked ? (uint8_t *)locked : pageable.data();
  const bool pinned = locked != nullptr;
  uint64_t rays = 0;
  for (int it = 0; it < (repeat > 0 ? repeat : 1); it++) {
    auto t0 = std::chrono::steady_clock::now();
    rc = gpu_render_ids(&s, &cam, fb, nullptr, &rays, w, h, depth);
    double ms =
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rc != RT_OK) {
      fprintf(stderr, "gpu_render failed (%d): %s\n", rc, rt_last_error());
      rt_free_host_buffer(locked);
      rt_host_free_scene(&hs);
      return 1;
    }
    fprintf(stderr, "%dx%d, %d triangles, %d spheres, %d lights, depth %d: %llu rays in %.1f ms "
                    "(upload + render + copy-back%s)\n",
            w, h, s.t_size, s.s_size, s.l_size, depth, (unsigned long long)rays, ms,
            pinned ? ", page-locked frame buffer" : "");
  }
  if (flag(argc, argv, "-n") && rt_host_write_ppm(out, fb, w, h) != 0) { // main.cpp:155-173
    fprintf(stderr, "cannot write %s\n", out);
    rt_free_host_buffer(locked);
    rt_host_free_scene(&hs);
    return 1;
  }
  rt_free_host_buffer(locked);
  rt_host_free_scene(&hs);
  return 0;
}
