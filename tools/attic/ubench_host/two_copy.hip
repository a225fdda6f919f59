// Microbenchmark (lab, not product): the two-copy host scheme enqueued from ONE thread -- a pageable 64 MiB frame goes up in B bands
// (linear copies, stream 1), a device kernel per band behind it, an event, the band's download into PINNED host memory on stream 2.
//   hipcc -O2 --offload-arch=gfx950 tools/attic/ubench_host/two_copy.hip -o /tmp/two_copy && /tmp/two_copy
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void copy_kernel(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
#include <sys/mman.h>
int main(int argc, char** argv) {
  const bool reg = argc > 1 && !strcmp(argv[1], "reg");       // the destination as the Python pool makes it: mmap + hipHostRegister
  const size_t n = 64u << 20;
  std::vector<char*> pages;
  for (int i = 0; i < 12; ++i) { char* p = (char*)aligned_alloc(4096, n); memset(p, 1 + i, n); pages.push_back(p); }
  char *pin = nullptr, *d0 = nullptr, *d1 = nullptr;
  if (reg) {
    pin = (char*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    memset(pin, 3, n);
    CK(hipHostRegister(pin, n, hipHostRegisterDefault));
    printf("destination: mmap + hipHostRegister\n");
  } else {
    CK(hipHostMalloc((void**)&pin, n, hipHostMallocDefault));
    memset(pin, 3, n);
  }
  CK(hipMalloc((void**)&d0, n));
  CK(hipMalloc((void**)&d1, n));
  hipStream_t s_up, s_down, s_k;
  CK(hipStreamCreateWithFlags(&s_k, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s_down, hipStreamNonBlocking));
  std::vector<hipEvent_t> ev(64), ev2(64);
  for (auto& e : ev2) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (int three = 0; three < 2; ++three)
  for (int fresh = 0; fresh < 2; ++fresh)
    for (int B : {1, 6, 12}) {
      std::vector<double> ts;
      for (int rep = 0; rep < 11; ++rep) {
        const char* src = pages[fresh ? (size_t)rep % pages.size() : 0];
        const double t0 = now();
        for (int k = 0; k < B; ++k) {
          const size_t o = n / B / 16 * 16 * k, len = (k == B - 1) ? n - o : n / B / 16 * 16;
          CK(hipMemcpyAsync(d0 + o, src + o, len, hipMemcpyHostToDevice, s_up));
          if (three) {          // uploads alone on their stream, kernels on a third
            CK(hipEventRecord(ev2[k], s_up));
            CK(hipStreamWaitEvent(s_k, ev2[k], 0));
            hipLaunchKernelGGL(copy_kernel, dim3(1024), dim3(256), 0, s_k, (const float4*)(d0 + o), (float4*)(d1 + o), len / 16);
            CK(hipEventRecord(ev[k], s_k));
          } else {
            hipLaunchKernelGGL(copy_kernel, dim3(1024), dim3(256), 0, s_up, (const float4*)(d0 + o), (float4*)(d1 + o), len / 16);
            CK(hipEventRecord(ev[k], s_up));
          }
          CK(hipStreamWaitEvent(s_down, ev[k], 0));
          CK(hipMemcpyAsync(pin + o, d1 + o, len, hipMemcpyDeviceToHost, s_down));
        }
        CK(hipStreamSynchronize(s_up));
        CK(hipStreamSynchronize(s_k));
        CK(hipStreamSynchronize(s_down));
        ts.push_back(now() - t0);
      }
      std::sort(ts.begin(), ts.end());
      printf("%s %s pageable source, %2d bands   median %.3f ms  min %.3f  max %.3f\n", three ? "three streams (upload | kernel | download)" : "two streams (upload, kernel | download)  ", fresh ? "a different" : "the same   ", B,
             ts[ts.size() / 2] * 1e3, ts.front() * 1e3, ts.back() * 1e3);
    }
  return 0;
}
