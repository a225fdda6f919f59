// Microbenchmark (lab, not product): how fast does a PAGEABLE 64 MiB host frame reach the device?
//   (a) hipMemcpyAsync straight from the pageable buffer (the runtime pins or stages it itself),
//   (b) from a registered buffer (the PCIe floor),
//   (c) staged by T host threads through a pinned buffer in chunks of C KiB, one hipMemcpyAsync per chunk,
// each alone and with a 64 MiB download from the device into pinned memory running on a second stream.
//   hipcc -O2 -pthread tools/attic/ubench_host/staged_upload.cpp -o /tmp/staged_upload && /tmp/staged_upload
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const size_t n = 64u << 20;
  char* page = (char*)aligned_alloc(4096, n);
  memset(page, 1, n);
  char *pin_up = nullptr, *pin_down = nullptr, *d0 = nullptr, *d1 = nullptr;
  CK(hipHostMalloc((void**)&pin_up, n, hipHostMallocDefault));
  CK(hipHostMalloc((void**)&pin_down, n, hipHostMallocDefault));
  memset(pin_up, 2, n);
  memset(pin_down, 3, n);
  CK(hipMalloc((void**)&d0, n));
  CK(hipMalloc((void**)&d1, n));
  hipStream_t s_up, s_down;
  CK(hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s_down, hipStreamNonBlocking));
  auto with_download = [&](bool dl, auto&& up) {
    double best = 1e30;
    std::vector<double> ts;
    for (int rep = 0; rep < 9; ++rep) {
      const double t0 = now();
      if (dl) CK(hipMemcpyAsync(pin_down, d1, n, hipMemcpyDeviceToHost, s_down));
      up();
      CK(hipStreamSynchronize(s_up));
      if (dl) CK(hipStreamSynchronize(s_down));
      ts.push_back(now() - t0);
    }
    std::sort(ts.begin(), ts.end());
    best = ts[ts.size() / 2];
    return best * 1e3;
  };
  for (int dl = 0; dl < 2; ++dl) {
    printf("---- %s\n", dl ? "with a 64 MiB download to pinned memory on a second stream" : "upload alone");
    printf("pageable, one hipMemcpyAsync            %.3f ms\n", with_download(dl, [&] { CK(hipMemcpyAsync(d0, page, n, hipMemcpyHostToDevice, s_up)); }));
    printf("pinned, one hipMemcpyAsync              %.3f ms\n", with_download(dl, [&] { CK(hipMemcpyAsync(d0, pin_up, n, hipMemcpyHostToDevice, s_up)); }));
    for (int T : {1, 2, 4, 6, 8, 12}) {
      for (size_t ckb : {512, 1024, 2048, 4096}) {
        const size_t C = ckb << 10, nchunk = n / C;
        const double t = with_download(dl, [&] {
          std::atomic<size_t> next{0};
          std::vector<std::thread> th;
          for (int t_ = 0; t_ < T; ++t_)
            th.emplace_back([&] {
              for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= nchunk) break;
                memcpy(pin_up + i * C, page + i * C, C);
                CK(hipMemcpyAsync(d0 + i * C, pin_up + i * C, C, hipMemcpyHostToDevice, s_up));
              }
            });
          for (auto& x : th) x.join();
        });
        printf("staged: %2d threads, %4zu KiB chunks       %.3f ms\n", T, ckb, t);
      }
    }
    // host memcpy alone (no device): what the threads can move
    for (int T : {1, 2, 4, 8}) {
      double best = 1e30;
      for (int rep = 0; rep < 5; ++rep) {
        const double t0 = now();
        std::vector<std::thread> th;
        for (int t_ = 0; t_ < T; ++t_) th.emplace_back([&, t_] { memcpy(pin_up + n / T * t_, page + n / T * t_, n / T); });
        for (auto& x : th) x.join();
        best = std::min(best, now() - t0);
      }
      printf("host memcpy pageable -> pinned, %d threads  %.3f ms (%.1f GB/s)\n", T, best * 1e3, n / best / 1e9);
    }
  }
  return 0;
}
