// Cost of wave-wide gather shapes through the vector L1 (TCP) on gfx950: which per-lane access
// width / stride / alignment the bilinear gather should use.  Data stays L2/L1 resident.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_gather tools/ubench_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int ITERS = 512;

// MODE: 0 dword stride4 | 1 dwordx2 stride8 aligned | 2 dwordx2 stride4 (overlap) base%8==0 | 3 same base%8==4
//       4 dwordx4 stride16 | 5 dwordx4 stride8 overlap | 6 two dwords (x, x+1) stride4 | 7 dwordx3 stride8
//       8 dwordx2 stride4 with a row jump every 16 lanes (distorted footprint) | 9 dwordx4 stride 4 (overlap 4x)
template <int MODE>
__global__ void __launch_bounds__(256) k_gather(const float* __restrict__ src, float* out, unsigned bytes, int row_bytes, int rows) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)bytes, 0x00020000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // each workgroup walks its own 1 KB-wide column of the window, top to bottom, like a remap tile
  unsigned base = (blockIdx.x % 16) * 1024 + wave * 256 + (blockIdx.x / 16) * (unsigned)row_bytes * (unsigned)rows;
  unsigned off;
  if (MODE == 0 || MODE == 6 || MODE == 2 || MODE == 9) off = base + lane * 4;
  else if (MODE == 3) off = base + 4 + lane * 4;
  else if (MODE == 1 || MODE == 5 || MODE == 7) off = base + lane * 8;
  else if (MODE == 4) off = base + lane * 16;
  else off = base + lane * 4 + (lane >> 4) * row_bytes;   // MODE 8
  float acc = 0.f;
  int row = 0;
  for (int i = 0; i < ITERS; ++i) {
    unsigned o = off + row * row_bytes;
    if (MODE == 0) { acc += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, o, 0, 0)); }
    else if (MODE == 6) { acc += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, o, 0, 0)) + __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, o + 4, 0, 0)); }
    else if (MODE == 1 || MODE == 2 || MODE == 3 || MODE == 8) { u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, o, 0, 0); acc += __uint_as_float(v.x) + __uint_as_float(v.y); }
    else if (MODE == 7) { u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(r, o, 0, 0); acc += __uint_as_float(v.x) + __uint_as_float(v.y) + __uint_as_float(v.z); }
    else { u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, 0); acc += __uint_as_float(v.x) + __uint_as_float(v.y) + __uint_as_float(v.z) + __uint_as_float(v.w); }
    row = row + 1; if (row >= rows) row = 0;
  }
  if (acc == 123.456f) out[threadIdx.x] = acc;
}

template <typename F> static float time_ms(F&& f, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int r = 0; r < reps; ++r) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}

int main(int argc, char** argv) {
  // rows walked per workgroup: 8 -> the working set stays in L1; 512 (= ITERS) -> every row is
  // new: 2048 workgroups x 512 rows x 1 KB = 1 GiB streamed from HBM once
  const int row_bytes = 16384, rows = argc > 1 ? atoi(argv[1]) : 8;
  const int blocks = 256 * 8;
  size_t bytes64 = (size_t)row_bytes * rows * (blocks / 16) + (size_t)row_bytes * 8;
  if (bytes64 > 0xffff0000ull) { printf("window too large\n"); return 1; }
  unsigned bytes = (unsigned)bytes64;
  float *src, *out; CK(hipMalloc(&src, bytes)); CK(hipMalloc(&out, 4096)); CK(hipMemset(src, 0, bytes));
  printf("rows per workgroup %d, window %.1f MB\n", rows, bytes / 1e6);
  const char* names[] = {"dword   stride 4  (coalesced 256 B)", "dwordx2 stride 8  (coalesced 512 B)", "dwordx2 stride 4  overlap, base%8=0",
                         "dwordx2 stride 4  overlap, base%8=4", "dwordx4 stride 16 (coalesced 1 KB)", "dwordx4 stride 8  overlap",
                         "2x dword (x, x+1) stride 4", "dwordx3 stride 8  overlap", "dwordx2 stride 4 + row jump per 16 lanes", "dwordx4 stride 4 overlap"};
#define RUN(M) { float ms = time_ms([&]{ hipLaunchKernelGGL(k_gather<M>, dim3(blocks), dim3(256), 0, 0, src, out, bytes, row_bytes, rows); }, 5); \
    double wave_instr_per_cu = (double)ITERS * 8 * 4 * ((M) == 6 ? 2 : 1); \
    printf("mode %d %-42s %.3f ms -> %.1f cycles/wave-load/CU @2.4GHz\n", M, names[M], ms, ms * 1e-3 * 2.4e9 / wave_instr_per_cu); }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9)
  return 0;
}
