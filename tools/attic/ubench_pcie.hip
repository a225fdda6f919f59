// Host<->device transfer rates for one 4096^2 float32 frame (64 MiB): pageable, registered, pinned,
// chunked, and both directions at once.  Informs the DCP_MEM_HOST staging of api_image.cpp / api_stack.cpp.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_pcie.hip -o /tmp/ubench_pcie && /tmp/ubench_pcie
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t n = 64u << 20;
  void *d0, *d1;
  CK(hipMalloc(&d0, n)); CK(hipMalloc(&d1, n));
  char* pg = (char*)aligned_alloc(4096, n); memset(pg, 1, n);
  char* pg2 = (char*)aligned_alloc(4096, n); memset(pg2, 2, n);
  char *pin, *pin2;
  CK(hipHostMalloc((void**)&pin, n, hipHostMallocDefault)); memset(pin, 3, n);
  CK(hipHostMalloc((void**)&pin2, n, hipHostMallocDefault)); memset(pin2, 4, n);
  hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
  auto rep = [&](const char* name, auto&& f, int reps = 5) {
    f(); CK(hipDeviceSynchronize());
    double best = 1e9;
    for (int r = 0; r < reps; ++r) { double t = now(); f(); CK(hipDeviceSynchronize()); t = now() - t; if (t < best) best = t; }
    printf("%-52s %8.3f ms  %6.1f GB/s (64 MiB)\n", name, best * 1e3, n / best / 1e9);
  };
  rep("H2D pageable hipMemcpy", [&] { CK(hipMemcpy(d0, pg, n, hipMemcpyHostToDevice)); });
  rep("D2H pageable hipMemcpy", [&] { CK(hipMemcpy(pg2, d1, n, hipMemcpyDeviceToHost)); });
  rep("H2D pinned hipMemcpyAsync", [&] { CK(hipMemcpyAsync(d0, pin, n, hipMemcpyHostToDevice, s0)); });
  rep("D2H pinned hipMemcpyAsync", [&] { CK(hipMemcpyAsync(pin2, d1, n, hipMemcpyDeviceToHost, s1)); });
  rep("H2D + D2H pinned, two streams", [&] { CK(hipMemcpyAsync(d0, pin, n, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(pin2, d1, n, hipMemcpyDeviceToHost, s1)); });
  rep("H2D + D2H pageable, two streams (async calls)", [&] { CK(hipMemcpyAsync(d0, pg, n, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(pg2, d1, n, hipMemcpyDeviceToHost, s1)); });
  rep("memcpy pageable -> pinned (1 thread)", [&] { memcpy(pin, pg, n); });
  rep("hipHostRegister + Unregister 64 MiB", [&] { CK(hipHostRegister(pg, n, hipHostRegisterDefault)); CK(hipHostUnregister(pg)); });
  rep("register + H2D + unregister", [&] { CK(hipHostRegister(pg, n, hipHostRegisterDefault)); CK(hipMemcpyAsync(d0, pg, n, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0)); CK(hipHostUnregister(pg)); });
  CK(hipHostRegister(pg, n, hipHostRegisterDefault)); CK(hipHostRegister(pg2, n, hipHostRegisterDefault));
  rep("H2D registered", [&] { CK(hipMemcpyAsync(d0, pg, n, hipMemcpyHostToDevice, s0)); });
  rep("H2D + D2H registered, two streams", [&] { CK(hipMemcpyAsync(d0, pg, n, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(pg2, d1, n, hipMemcpyDeviceToHost, s1)); });
  for (int chunks : {4, 16}) {
    char name[96]; snprintf(name, sizeof name, "H2D pinned in %d chunks, one stream", chunks);
    rep(name, [&] { for (int c = 0; c < chunks; ++c) CK(hipMemcpyAsync((char*)d0 + c * (n / chunks), pin + c * (n / chunks), n / chunks, hipMemcpyHostToDevice, s0)); });
    snprintf(name, sizeof name, "H2D pageable in %d chunks via pinned bounce x2", chunks);
    rep(name, [&] {
      // double-buffered: CPU memcpy of chunk c+1 into the other half of a pinned buffer while chunk c is in flight
      const size_t cs = n / chunks; hipEvent_t ev[2]; CK(hipEventCreate(&ev[0])); CK(hipEventCreate(&ev[1]));
      for (int c = 0; c < chunks; ++c) {
        char* b = pin + (c & 1) * cs;
        if (c >= 2) CK(hipEventSynchronize(ev[c & 1]));
        memcpy(b, pg2 + c * cs, cs);
        CK(hipMemcpyAsync((char*)d0 + c * cs, b, cs, hipMemcpyHostToDevice, s0));
        CK(hipEventRecord(ev[c & 1], s0));
      }
      CK(hipEventDestroy(ev[0])); CK(hipEventDestroy(ev[1]));
    });
  }
  // a chunk of a (D, H, W) stack: 33 bands of 70 rows x 2560 floats, one projection (2560 x 2560 floats) apart
  {
    const size_t wbytes = 70u * 2560u * 4u, pitch = 2560u * 2560u * 4u, hgt = 33;
    char* big = (char*)aligned_alloc(4096, pitch * hgt); memset(big, 7, pitch * hgt);
    auto rep2 = [&](const char* name, auto&& f) {
      f(); CK(hipDeviceSynchronize());
      double best = 1e9;
      for (int r = 0; r < 5; ++r) { double t = now(); f(); CK(hipDeviceSynchronize()); t = now() - t; if (t < best) best = t; }
      printf("%-52s %8.3f ms  %6.1f GB/s (%.1f MB)\n", name, best * 1e3, wbytes * hgt / best / 1e9, wbytes * hgt / 1e6);
    };
    rep2("H2D pageable hipMemcpy2DAsync 33 x 717 KB bands", [&] { CK(hipMemcpy2DAsync(d0, wbytes, big, pitch, wbytes, hgt, hipMemcpyHostToDevice, s0)); });
    rep2("H2D pageable 33 x hipMemcpyAsync of 717 KB", [&] { for (size_t i = 0; i < hgt; ++i) CK(hipMemcpyAsync((char*)d0 + i * wbytes, big + i * pitch, wbytes, hipMemcpyHostToDevice, s0)); });
    rep2("H2D pageable one hipMemcpyAsync of the same bytes", [&] { CK(hipMemcpyAsync(d0, big, wbytes * hgt, hipMemcpyHostToDevice, s0)); });
    // the streamed stack path: 8 chunks up (2-D) in this thread while another thread copies 8 chunks down
    char* dn = (char*)aligned_alloc(4096, wbytes * hgt * 8); memset(dn, 1, wbytes * hgt * 8);
    for (int it = 0; it < 9; ++it) { const int mode = it % 3;
      double t = now();
      std::thread th([&] { for (int c = 0; c < 8; ++c) { CK(hipMemcpyAsync(dn + c * wbytes * hgt, d1, wbytes * hgt, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); } });
      for (int c = 0; c < 8; ++c) {
        if (mode == 0) CK(hipMemcpy2DAsync(d0, wbytes, big, pitch, wbytes, hgt, hipMemcpyHostToDevice, s0));
        else if (mode == 1) CK(hipMemcpyAsync(d0, big, wbytes * hgt, hipMemcpyHostToDevice, s0));
        else for (size_t i = 0; i < hgt; ++i) CK(hipMemcpyAsync((char*)d0 + i * wbytes, big + i * pitch, wbytes, hipMemcpyHostToDevice, s0));
        CK(hipStreamSynchronize(s0));
      }
      th.join();
      t = now() - t;
      printf("8 chunks up (%s) || 8 chunks down, two threads: %.3f ms (%.1f MB each way)\n", mode == 0 ? "2-D" : mode == 1 ? "1-D" : "33 x 1-D", t * 1e3, wbytes * hgt * 8 / 1e6);
    }
    free(dn);
    free(big);
  }
  // first-time registration of buffers never seen by the runtime, and unregistration
  for (int k = 0; k < 4; ++k) {
    char* f = (char*)aligned_alloc(4096, n); memset(f, k, n);
    double t0 = now(); CK(hipHostRegister(f, n, hipHostRegisterDefault)); double t1 = now();
    CK(hipMemcpyAsync(d0, f, n, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0)); double t2 = now();
    CK(hipHostUnregister(f)); double t3 = now();
    printf("fresh buffer %d: register %.3f ms, H2D %.3f ms, unregister %.3f ms\n", k, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
    free(f);
  }
  // two host threads, blocking pageable copies in opposite directions
  {
    char* a = (char*)aligned_alloc(4096, n); memset(a, 5, n);
    char* b = (char*)aligned_alloc(4096, n); memset(b, 6, n);
    rep("H2D + D2H pageable, two host threads", [&] {
      std::thread t([&] { CK(hipMemcpyAsync(b, d1, n, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); });
      CK(hipMemcpyAsync(d0, a, n, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0));
      t.join();
    });
  }
  return 0;
}
