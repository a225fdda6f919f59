// Vector-L1 cost of the bilinear access pattern without the arithmetic: every wave walks down a
// 64-pixel-wide column of a row-major image, per output row it needs source rows r and r+1 (two taps
// each) and stores one value per lane.  Variants differ in how the taps are fetched.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// MODE 0: 2 x dwordx2 (rows r, r+1)            -- what remap_tile_kernel does
// MODE 1: 4 x dword                             -- same taps as single dwords
// MODE 2: 1 x dwordx2 (row r+1), row r reused from registers
// MODE 3: 2 x dword (row r+1: x, x+1), row r reused
// MODE 4: 1 x dword (row r+1, x), x+1 taken from the neighbour lane (DPP), row r reused
// MODE 5: store only
template <int MODE>
__global__ void __launch_bounds__(256) k_walk(const float* __restrict__ src, float* __restrict__ dst, unsigned bytes,
                                              int W, int rows) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)bytes, 0x00020000);
  const int tiles_x = W / 256;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const unsigned x = tx * 256 + threadIdx.x;
  const unsigned rb = (unsigned)W * 4u;
  unsigned off = ((unsigned)ty * rows * W + x) * 4u + 4u;   // +4: taps are 4-byte, not 8-byte, aligned
  float* out = dst + (size_t)ty * rows * W + x;
  u32x2 prev = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
  for (int k = 0; k < rows; ++k) {
    float acc;
    if (MODE == 0) {
      u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
      u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(r, off, rb, 0);
      acc = __uint_as_float(a.x) + __uint_as_float(a.y) + __uint_as_float(b.x) + __uint_as_float(b.y);
    } else if (MODE == 1) {
      acc = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0)) +
            __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off + 4, 0, 0)) +
            __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, rb, 0)) +
            __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off + 4, rb, 0));
    } else if (MODE == 2) {
      u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(r, off, rb, 0);
      acc = __uint_as_float(prev.x) + __uint_as_float(prev.y) + __uint_as_float(b.x) + __uint_as_float(b.y);
      prev = b;
    } else if (MODE == 3) {
      u32x2 b;
      b.x = __builtin_amdgcn_raw_buffer_load_b32(r, off, rb, 0);
      b.y = __builtin_amdgcn_raw_buffer_load_b32(r, off + 4, rb, 0);
      acc = __uint_as_float(prev.x) + __uint_as_float(prev.y) + __uint_as_float(b.x) + __uint_as_float(b.y);
      prev = b;
    } else if (MODE == 4) {
      u32x2 b;
      b.x = __builtin_amdgcn_raw_buffer_load_b32(r, off, rb, 0);
      b.y = (unsigned)__builtin_amdgcn_update_dpp((int)b.x, (int)b.x, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
      acc = __uint_as_float(prev.x) + __uint_as_float(prev.y) + __uint_as_float(b.x) + __uint_as_float(b.y);
      prev = b;
    } else {
      acc = (float)k;
    }
    out[(size_t)k * W] = acc;
    off += rb;
  }
}

template <typename F> static float time_ms(F&& f, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int r = 0; r < reps; ++r) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}

int main(int argc, char** argv) {
  const int W = 4096, H = 4096, rows = argc > 1 ? atoi(argv[1]) : 16;
  const int NR = 12;                                  // ring of frames > infinity cache
  size_t fb = (size_t)W * H * 4;
  float *src[NR], *dst[NR];
  for (int i = 0; i < NR; ++i) { CK(hipMalloc(&src[i], fb + 65536)); CK(hipMalloc(&dst[i], fb)); CK(hipMemset(src[i], 0, fb + 65536)); }
  const int blocks = (W / 256) * (H / rows);
  const char* names[] = {"2 x dwordx2 (rows r, r+1)", "4 x dword", "1 x dwordx2 + register reuse of row r", "2 x dword + register reuse",
                         "1 x dword + DPP neighbour + register reuse", "store only"};
  printf("4096^2 frame, %d rows per workgroup walk, ring of %d frames\n", rows, NR);
#define RUN(M) { int it = 0; float ms = time_ms([&]{ hipLaunchKernelGGL(k_walk<M>, dim3(blocks), dim3(256), 0, 0, src[it % NR], dst[it % NR], (unsigned)fb + 4096u, W, rows); ++it; }, 48); \
    printf("mode %d %-44s %.2f us per frame  (%.2f TB/s of 8 B/pixel)\n", M, names[M], ms * 1e3, 8.0 * W * H / ms / 1e9); }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
  return 0;
}
