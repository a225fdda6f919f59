// Two host threads, pageable memory: 16 x 26 MB chunks up while 16 x 26 MB chunks go down, each chunk at
// its own host address (a stack streamed through the GPU), with and without hipHostRegister.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const size_t cb = 2560u * 2560u * 4u, nc = 16;
  void *d0, *d1; CK(hipMalloc(&d0, cb)); CK(hipMalloc(&d1, cb));
  char* up = (char*)aligned_alloc(4096, cb * nc); memset(up, 1, cb * nc);
  char* dn = (char*)aligned_alloc(4096, cb * nc); memset(dn, 2, cb * nc);
  for (int flags = 0; flags < 2; ++flags) {
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, flags ? hipStreamNonBlocking : hipStreamDefault));
    CK(hipStreamCreateWithFlags(&s1, flags ? hipStreamNonBlocking : hipStreamDefault));
    for (int mode = 0; mode < 4; ++mode) {
      for (int it = 0; it < 3; ++it) {
        double t = now();
        if (mode == 3) { CK(hipHostRegister(up, cb * nc, 0)); CK(hipHostRegister(dn, cb * nc, 0)); }
        std::thread th;
        if (mode != 1) th = std::thread([&] { for (size_t c = 0; c < nc; ++c) { CK(hipMemcpyAsync(dn + c * cb, d1, cb, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); } });
        if (mode != 2) for (size_t c = 0; c < nc; ++c) { CK(hipMemcpyAsync(d0, up + c * cb, cb, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0)); }
        if (th.joinable()) th.join();
        if (mode == 3) { CK(hipHostUnregister(up)); CK(hipHostUnregister(dn)); }
        t = now() - t;
        printf("%s streams, %s: %.2f ms\n", flags ? "non-blocking" : "default", mode == 0 ? "up || down" : mode == 1 ? "up only" : mode == 2 ? "down only" : "register + up || down + unregister", t * 1e3);
      }
    }
  }
  return 0;
}
