// tools/host_register_probe.hip — does hipHostRegister work on this machine, and
// at what rate do device-to-host copies run into pageable, registered and
// hipHostMalloc'd memory?  hipcc -O2 --offload-arch=gfx950 -o build/host_register_probe tools/host_register_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/resource.h>

static double copy_ms(void *dst, const void *src, size_t n) {
  hipMemcpy(dst, src, n, hipMemcpyDeviceToHost);
  auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < 5; k++)
    hipMemcpy(dst, src, n, hipMemcpyDeviceToHost);
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 5;
}

int main() {
  rlimit rl;
  getrlimit(RLIMIT_MEMLOCK, &rl);
  printf("RLIMIT_MEMLOCK soft %llu hard %llu\n", (unsigned long long)rl.rlim_cur,
         (unsigned long long)rl.rlim_max);
  const size_t n = (size_t)3840 * 2160 * 4;
  void *d = nullptr;
  if (hipMalloc(&d, n) != hipSuccess)
    return 1;
  hipMemset(d, 1, n);
  void *pageable = malloc(n + 64);
  memset(pageable, 0, n + 64);
  printf("pageable: %.3f ms per 33 MB copy\n", copy_ms(pageable, d, n));
  void *aligned = nullptr;
  posix_memalign(&aligned, 4096, n);
  memset(aligned, 0, n);
  const unsigned flags[] = {hipHostRegisterDefault, hipHostRegisterPortable, hipHostRegisterMapped};
  const char *names[] = {"Default", "Portable", "Mapped"};
  for (int f = 0; f < 3; f++) {
    hipError_t e = hipHostRegister(aligned, n, flags[f]);
    printf("hipHostRegister(page-aligned, %s): %s\n", names[f], hipGetErrorString(e));
    if (e == hipSuccess) {
      printf("  registered: %.3f ms per copy\n", copy_ms(aligned, d, n));
      hipHostUnregister(aligned);
    }
    (void)hipGetLastError();
    e = hipHostRegister((char *)pageable + 48, n, flags[f]);
    printf("hipHostRegister(unaligned, %s): %s\n", names[f], hipGetErrorString(e));
    if (e == hipSuccess)
      hipHostUnregister((char *)pageable + 48);
    (void)hipGetLastError();
  }
  hipError_t e = hipHostRegister(aligned, 4096, hipHostRegisterDefault);
  printf("hipHostRegister(one page): %s\n", hipGetErrorString(e));
  (void)hipGetLastError();
  void *pinned = nullptr;
  e = hipHostMalloc(&pinned, n, hipHostMallocDefault);
  printf("hipHostMalloc: %s\n", hipGetErrorString(e));
  if (e == hipSuccess) {
    printf("  hipHostMalloc'd: %.3f ms per copy\n", copy_ms(pinned, d, n));
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < 5; k++)
      memcpy(pageable, pinned, n);
    printf("  memcpy pinned -> pageable, one thread: %.3f ms\n",
           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 5);
  }
  return 0;
}
