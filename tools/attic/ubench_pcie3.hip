// Why do H2D and D2H stop overlapping in the streamed stack path?  Same as ubench_pcie2 "up || down" with
// (a) a kernel on the upload stream between copies, (b) the kernel on a third stream, (c) 2-D uploads.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(float* p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0f; }
int main() {
  const size_t cb = 2560u * 2560u * 4u, nc = 16;
  void *d0, *d1; CK(hipMalloc(&d0, cb)); CK(hipMalloc(&d1, cb));
  char* up = (char*)aligned_alloc(4096, cb * nc); memset(up, 1, cb * nc);
  char* dn = (char*)aligned_alloc(4096, cb * nc); memset(dn, 2, cb * nc);
  hipStream_t s0, s1, s2;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const char* names[] = {"plain 1-D", "kernel on the upload stream", "kernel on a third stream", "2-D upload (height 1)", "2-D upload + kernel on upload stream",
                         "kernel on third stream, device-wide sync after it"};
  for (int mode = 0; mode < 6; ++mode)
    for (int it = 0; it < 3; ++it) {
      double t = now();
      std::thread th([&] { for (size_t c = 0; c < nc; ++c) { CK(hipMemcpyAsync(dn + c * cb, d1, cb, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); } });
      for (size_t c = 0; c < nc; ++c) {
        if (mode == 3 || mode == 4) CK(hipMemcpy2DAsync(d0, cb, up + c * cb, cb, cb, 1, hipMemcpyHostToDevice, s0));
        else CK(hipMemcpyAsync(d0, up + c * cb, cb, hipMemcpyHostToDevice, s0));
        if (mode == 1 || mode == 4) hipLaunchKernelGGL(touch, dim3(6400), dim3(256), 0, s0, (float*)d0, cb / 4);
        CK(hipStreamSynchronize(s0));
        if (mode == 2 || mode == 5) { hipLaunchKernelGGL(touch, dim3(6400), dim3(256), 0, s2, (float*)d0, cb / 4); CK(hipStreamSynchronize(s2)); }
      }
      th.join();
      t = now() - t;
      printf("%-52s %.2f ms\n", names[mode], t * 1e3);
    }
  return 0;
}
