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
// the frame kernels' store shape: 4 bytes per lane, a wave writes 256 contiguous bytes of one row, 16 rows per wave
__global__ void copy_kernel_b32(const float* __restrict__ in, float* __restrict__ out, size_t n, int W) {
  const size_t tiles_x = W / 64;
  for (size_t t = blockIdx.x * 4 + (threadIdx.x >> 6); t * 16 * 64 < n; t += (size_t)gridDim.x * 4) {
    const size_t tx = t % tiles_x, ty = t / tiles_x;
    for (int k = 0; k < 16; ++k) {
      const size_t i = (ty * 16 + k) * W + tx * 64 + (threadIdx.x & 63);
      if (i < n) __builtin_nontemporal_store(in[i], &out[i]);
    }
  }
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
  timeit("kernel device -> pinned host alone, 4-byte nt stores in 64 x 16 wave tiles", [&] { hipLaunchKernelGGL(copy_kernel_b32, dim3(2048), dim3(256), 0, s_k, (const float*)d0, (float*)pin, n / 4, 4096); });
  for (int B : {6, 12}) {
    char what[128];
    snprintf(what, sizeof(what), "pageable source, %2d bands, 4-byte nt stores in wave tiles", B);
    timeit(what, [&] {
      for (int k = 0; k < B; ++k) {
        const size_t o = n / B / 65536 * 65536 * k, len = (k == B - 1) ? n - o : n / B / 65536 * 65536;
        CK(hipMemcpyAsync(d0 + o, page + o, len, hipMemcpyHostToDevice, s_up));
        CK(hipEventRecord(ev[k], s_up));
        CK(hipStreamWaitEvent(s_k, ev[k], 0));
        hipLaunchKernelGGL(copy_kernel_b32, dim3(1024), dim3(256), 0, s_k, (const float*)(d0 + o), (float*)(pin + o), len / 4, 4096);
      }
    });
  }
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
