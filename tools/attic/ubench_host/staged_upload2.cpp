// Microbenchmark (lab, not product), second part: the upload of a FRESH pageable 64 MiB frame (a buffer the runtime has not seen:
// its pin cache cannot help), whole or in 24 bands, 1-D and 2-D copies, against staging through pinned memory by host threads.
//   hipcc -O2 -pthread tools/attic/ubench_host/staged_upload2.cpp -o /tmp/staged_upload2 && /tmp/staged_upload2
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
  const size_t n = 64u << 20, row = 16384, rows = n / row;
  const int NB = 12;
  std::vector<char*> fresh;
  auto get_fresh = [&]() { char* p = (char*)aligned_alloc(4096, n); memset(p, 1, n); fresh.push_back(p); return p; };
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
  CK(hipMemcpyAsync(d0, pin_up, n, hipMemcpyHostToDevice, s_up));
  CK(hipStreamSynchronize(s_up));
  auto run = [&](const char* what, bool dl, bool fresh_each, auto&& up) {
    std::vector<double> ts;
    char* same = get_fresh();
    for (int rep = 0; rep < 7; ++rep) {
      char* src = fresh_each ? get_fresh() : same;
      const double t0 = now();
      if (dl) CK(hipMemcpyAsync(pin_down, d1, n, hipMemcpyDeviceToHost, s_down));
      up(src);
      CK(hipStreamSynchronize(s_up));
      if (dl) CK(hipStreamSynchronize(s_down));
      ts.push_back(now() - t0);
    }
    std::sort(ts.begin(), ts.end());
    printf("%-58s %s %s  median %.3f ms  min %.3f  max %.3f\n", what, fresh_each ? "fresh buffer" : "same buffer ", dl ? "+download" : "alone    ", ts[ts.size() / 2] * 1e3,
           ts.front() * 1e3, ts.back() * 1e3);
    for (char* p : fresh) free(p);
    fresh.clear();
  };
  for (int dl = 0; dl < 2; ++dl)
    for (int fe = 0; fe < 2; ++fe) {
      run("one hipMemcpyAsync", dl, fe, [&](char* src) { CK(hipMemcpyAsync(d0, src, n, hipMemcpyHostToDevice, s_up)); });
      run("12 bands, hipMemcpyAsync", dl, fe, [&](char* src) {
        for (int k = 0; k < NB; ++k) CK(hipMemcpyAsync(d0 + n / NB * k, src + n / NB * k, n / NB, hipMemcpyHostToDevice, s_up));
      });
      run("12 bands, hipMemcpy2DAsync (pitch = width)", dl, fe, [&](char* src) {
        for (int k = 0; k < NB; ++k)
          CK(hipMemcpy2DAsync(d0 + n / NB * k, row, src + n / NB * k, row, row, rows / NB, hipMemcpyHostToDevice, s_up));
      });
      for (int T : {2, 4, 8})
        for (size_t cmb : {4, 8}) {
          char what[96];
          snprintf(what, sizeof(what), "staged through pinned memory: %d threads, %zu MiB chunks", T, cmb);
          const size_t C = cmb << 20, nchunk = n / C;
          run(what, dl, fe, [&](char* src) {
            std::atomic<size_t> next{0};
            std::vector<std::thread> th;
            for (int t_ = 0; t_ < T; ++t_)
              th.emplace_back([&] {
                for (;;) {
                  const size_t i = next.fetch_add(1);
                  if (i >= nchunk) break;
                  memcpy(pin_up + i * C, src + i * C, C);
                  CK(hipMemcpyAsync(d0 + i * C, pin_up + i * C, C, hipMemcpyHostToDevice, s_up));
                }
              });
            for (auto& x : th) x.join();
          });
        }
      // hipHostRegister + copy + unregister
      run("hipHostRegister + one copy + hipHostUnregister", dl, fe, [&](char* src) {
        CK(hipHostRegister(src, n, hipHostRegisterDefault));
        CK(hipMemcpyAsync(d0, src, n, hipMemcpyHostToDevice, s_up));
        CK(hipStreamSynchronize(s_up));
        CK(hipHostUnregister(src));
      });
    }
  return 0;
}
