// Lab harness (not product): dcp_unwarp_image_f32 on a HOST frame from a plain C++ process -- pageable source, destination from
// hipHostMalloc ("malloc") or mmap + hipHostRegister ("reg" / "regshared", what the Python pool hands out); source pages plain or
// MADV_HUGEPAGE (numpy's); argv: <malloc|reg|regshared> <host_bands or 0> <plain|huge> [direct].  A C++ process runs on /opt/rocm's HIP
// runtime, the Python front end on the one bundled with PyTorch whenever torch is installed (discorpy_amd/_ffi.py).
//   hipcc -O2 tools/attic/ubench_host/host_call.cpp -Iinclude -Ldiscorpy_amd/lib -ldiscorpy_hip -Wl,-rpath,$PWD/discorpy_amd/lib -o tools/attic/ubench_host/host_call
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "discorpy_hip.h"
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const bool reg = argc > 1 && !strncmp(argv[1], "reg", 3);
  const bool shared = argc > 1 && !strcmp(argv[1], "regshared");      // Python's mmap.mmap(-1, n) is MAP_SHARED | MAP_ANONYMOUS
  const bool huge = argc > 3 && !strcmp(argv[3], "huge");             // numpy asks for transparent huge pages under its large arrays
  const int bands = argc > 2 ? atoi(argv[2]) : 0;
  const int64_t H = 4096, W = 4096;
  const size_t n = (size_t)H * W * 4;
  float* src = (float*)aligned_alloc(2u << 20, n);
  if (huge) madvise(src, n, MADV_HUGEPAGE);
  for (size_t i = 0; i < n / 4; ++i) src[i] = (float)(i % 977) * 0.001f;
  float* dst = nullptr;
  if (reg) {
    dst = (float*)mmap(nullptr, n, PROT_READ | PROT_WRITE, (shared ? MAP_SHARED : MAP_PRIVATE) | MAP_ANONYMOUS, -1, 0);
    memset(dst, 0, n);
    if (hipHostRegister(dst, n, hipHostRegisterDefault) != hipSuccess) return 1;
  } else {
    if (hipHostMalloc((void**)&dst, n, hipHostMallocDefault) != hipSuccess) return 1;
    memset(dst, 0, n);
  }
  const double fact[5] = {1.00227490554, -9.360115380562501e-06, 8.784366093749999e-09, -4.79328802218628e-12, 7.714082828693389e-16};
  if (bands) dcp_set_option("host_bands", bands);
  if (argc > 4 && !strcmp(argv[4], "direct")) dcp_set_option("host_direct", 2);
  std::vector<double> ts;
  for (int rep = 0; rep < 7; ++rep) {
    const double t0 = now();
    const int rc = dcp_unwarp_image_f32(src, dst, H, W, W, 1, 1883.8169650464, 1478.6964217312, fact, 5, 1, 1, DCP_BLEND_SCIPY, DCP_MEM_HOST, -1, nullptr);
    ts.push_back(now() - t0);
    if (rc) { fprintf(stderr, "rc %d: %s\n", rc, dcp_last_error()); return 1; }
  }
  std::sort(ts.begin(), ts.end());
  printf("%s%s destination, %s source, bands option %d: median %.3f ms  min %.3f  max %.3f  (%s)\n", reg ? "registered" : "hipHostMalloc", shared ? " MAP_SHARED" : "", huge ? "MADV_HUGEPAGE" : "plain", bands, ts[ts.size() / 2] * 1e3, ts.front() * 1e3,
         ts.back() * 1e3, dcp_debug_last_kernel());
  return 0;
}
