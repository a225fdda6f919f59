// Microbenchmark (lab, not product): the direct-write host scheme in isolation -- a pageable 64 MiB frame goes up in B bands on one
// stream (linear copies), a kernel on a second stream, behind an event per band, copies the band into pinned host memory (the shader's
// stores cross PCIe while the copy engine uploads).  What does the link give this scheme, by band count?
//   hipcc -O2 --offload-arch=gfx950 tools/attic/ubench_host/direct_write.hip -o /tmp/direct_write && /tmp/direct_write
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
int main() {
  const size_t n = 64u << 20;
  char* page = (char*)aligned_alloc(4096, n);
  memset(page, 1, n);
  char *pin = nullptr, *pin_src = nullptr, *d0 = nullptr;
  CK(hipHostMalloc((void**)&pin, n, hipHostMallocDefault));
  CK(hipHostMalloc((void**)&pin_src, n, hipHostMallocDefault));
  memset(pin, 3, n);
  memset(pin_src, 4, n);
  CK(hipMalloc((void**)&d0, n));
  hipStream_t s_up, s_k;
  CK(hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s_k, hipStreamNonBlocking));
  std::vector<hipEvent_t> ev(64);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  auto timeit = [&](const char* what, auto&& fn) {
    std::vector<double> ts;
    for (int rep = 0; rep < 9; ++rep) {
      const double t0 = now();
      fn();
      CK(hipStreamSynchronize(s_up));
      CK(hipStreamSynchronize(s_k));
      ts.push_back(now() - t0);
    }
    std::sort(ts.begin(), ts.end());
    printf("%-64s median %.3f ms  min %.3f\n", what, ts[ts.size() / 2] * 1e3, ts.front() * 1e3);
  };
  timeit("upload alone (pageable, one copy)", [&] { CK(hipMemcpyAsync(d0, page, n, hipMemcpyHostToDevice, s_up)); });
  timeit("kernel device -> pinned host alone (64 MiB, 2048 workgroups)", [&] { hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, s_k, (const float4*)d0, (float4*)pin, n / 16); });
  timeit("kernel device -> pinned host alone (64 MiB, 256 workgroups)", [&] { hipLaunchKernelGGL(copy_kernel, dim3(256), dim3(256), 0, s_k, (const float4*)d0, (float4*)pin, n / 16); });
  for (int src_kind = 0; src_kind < 2; ++src_kind)
    for (int B : {1, 3, 6, 12, 24}) {
      char what[128];
      snprintf(what, sizeof(what), "%s source, %2d bands: upload | event | kernel writes pinned", src_kind ? "pinned  " : "pageable", B);
      const char* src = src_kind ? pin_src : page;
      timeit(what, [&] {
        for (int k = 0; k < B; ++k) {
          const size_t o = n / B / 16 * 16 * k, len = (k == B - 1) ? n - o : n / B / 16 * 16;
          CK(hipMemcpyAsync(d0 + o, src + o, len, hipMemcpyHostToDevice, s_up));
          CK(hipEventRecord(ev[k], s_up));
          CK(hipStreamWaitEvent(s_k, ev[k], 0));
          hipLaunchKernelGGL(copy_kernel, dim3(1024), dim3(256), 0, s_k, (const float4*)(d0 + o), (float4*)(pin + o), len / 16);
        }
      });
    }
  return 0;
}
