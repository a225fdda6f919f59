// What this memory system gives pure reads, pure writes and a 1 : 1 mix, by bytes per lane and store policy (MI355X, gfx950):
// the ceilings the unwarp kernels are measured against.  2 GiB per stream, one element per thread.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_rw tools/ubench_rw.hip && /tmp/ubench_rw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <typename E, bool NT>
__global__ void __launch_bounds__(256) write_k(E* __restrict__ d, size_t n, E v) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    if constexpr (NT) __builtin_nontemporal_store(v, d + i);
    else d[i] = v;
  }
}
template <typename E>
__global__ void __launch_bounds__(256) read_k(const E* __restrict__ s, size_t n, E* sink) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const E v = s[i];
    if (((const unsigned char*)&v)[0] == 123 && ((const unsigned char*)&v)[1] == 77) *sink = v;     // (never true on zeroed memory)
  }
}
template <typename E, bool NT>
__global__ void __launch_bounds__(256) copy_k(const E* __restrict__ s, E* __restrict__ d, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const E v = s[i];
    if constexpr (NT) __builtin_nontemporal_store(v, d + i);
    else d[i] = v;
  }
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// the stack kernel's store pattern: a workgroup owns a 128 x 32 tile of a (D, H, W) float32 volume and walks `dc` projections; a wave
// owns 64 columns x 16 rows of it, a lane PX adjacent pixels of 16 / PX rows -- stores of 4 PX bytes, nothing else
template <int PX, int TW = 128>
__global__ void __launch_bounds__(256) tile_store_k(float* __restrict__ out, int H, int W, int D, int dc) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int WX = TW / 64;                      // waves side by side; the tile is TW x (64 / WX) rows
  const int x0 = blockIdx.x * TW + (wave % WX) * 64 + (lane % (64 / PX)) * PX;
  const int y0 = blockIdx.y * (64 / WX) + (wave / WX) * 16 + (lane / (64 / PX)) * (16 / PX);
  const int d0 = blockIdx.z * dc;
  typedef float vec __attribute__((ext_vector_type(PX)));
  for (int d = d0; d < d0 + dc && d < D; ++d) {
    float* o = out + ((size_t)d * H + y0) * W + x0;
#pragma unroll
    for (int r = 0; r < 16 / PX; ++r) {
      vec v;
      for (int p = 0; p < PX; ++p) v[p] = (float)(d + r + p);
      __builtin_nontemporal_store(v, (vec*)(o + (size_t)r * W));
    }
  }
}

// the same bytes as rows of LONG segments: a lane owns 4 adjacent pixels of 4 rows, a wave 256 pixels x 4 rows, the four waves of a
// workgroup WX side by side and 4 / WX on top of each other: segments of 1 KB x WX
template <int WX>
__global__ void __launch_bounds__(256) wide_store_k(float* __restrict__ out, int H, int W, int D, int dc) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int x0 = blockIdx.x * (256 * WX) + (wave % WX) * 256 + lane * 4;
  const int y0 = blockIdx.y * (16 / WX) + (wave / WX) * 4;
  const int d0 = blockIdx.z * dc;
  typedef float vec __attribute__((ext_vector_type(4)));
  if (x0 >= W) return;
  for (int d = d0; d < d0 + dc && d < D; ++d) {
    float* o = out + ((size_t)d * H + y0) * W + x0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      vec v = {(float)d, (float)r, 1.f, 2.f};
      __builtin_nontemporal_store(v, (vec*)(o + (size_t)r * W));
    }
  }
}

template <typename F>
static double time_us(F launch, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int r = 0; r < reps; ++r) launch();
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1000.0 / reps;
}

int main() {
  const size_t bytes = (size_t)2 << 30;
  void *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
  void* sink; CK(hipMalloc(&sink, 64));
#define GRID(E) dim3((unsigned)(bytes / sizeof(E) / 256))
#define REPORT(name, moved, us) printf("%-34s %8.1f us  %6.2f TB/s (%.2f of 8)\n", name, us, (double)(moved) / us / 1e6, (double)(moved) / us / 1e6 / 8.0)
  double us;
  us = time_us([&] { hipLaunchKernelGGL((write_k<unsigned short, true>), GRID(unsigned short), dim3(256), 0, 0, (unsigned short*)b, bytes / 2, (unsigned short)1); }, 5); REPORT("write  2 B per lane, nt", bytes, us);
  us = time_us([&] { hipLaunchKernelGGL((write_k<uint32_t, true>), GRID(uint32_t), dim3(256), 0, 0, (uint32_t*)b, bytes / 4, 1u); }, 5); REPORT("write  4 B per lane, nt", bytes, us);
  us = time_us([&] { hipLaunchKernelGGL((write_k<uint32_t, false>), GRID(uint32_t), dim3(256), 0, 0, (uint32_t*)b, bytes / 4, 1u); }, 5); REPORT("write  4 B per lane, plain", bytes, us);
  us = time_us([&] { hipLaunchKernelGGL((write_k<u32x2, true>), GRID(u32x2), dim3(256), 0, 0, (u32x2*)b, bytes / 8, u32x2{1, 2}); }, 5); REPORT("write  8 B per lane, nt", bytes, us);
  us = time_us([&] { hipLaunchKernelGGL((write_k<u32x4, true>), GRID(u32x4), dim3(256), 0, 0, (u32x4*)b, bytes / 16, u32x4{1, 2, 3, 4}); }, 5); REPORT("write 16 B per lane, nt", bytes, us);
  us = time_us([&] { hipLaunchKernelGGL((write_k<u32x4, false>), GRID(u32x4), dim3(256), 0, 0, (u32x4*)b, bytes / 16, u32x4{1, 2, 3, 4}); }, 5); REPORT("write 16 B per lane, plain", bytes, us);
  us = time_us([&] { hipLaunchKernelGGL((read_k<uint32_t>), GRID(uint32_t), dim3(256), 0, 0, (const uint32_t*)a, bytes / 4, (uint32_t*)sink); }, 5); REPORT("read   4 B per lane", bytes, us);
  us = time_us([&] { hipLaunchKernelGGL((read_k<u32x4>), GRID(u32x4), dim3(256), 0, 0, (const u32x4*)a, bytes / 16, (u32x4*)sink); }, 5); REPORT("read  16 B per lane", bytes, us);
  us = time_us([&] { hipLaunchKernelGGL((copy_k<unsigned short, true>), GRID(unsigned short), dim3(256), 0, 0, (const unsigned short*)a, (unsigned short*)b, bytes / 2); }, 5); REPORT("copy   2 B per lane, nt store", 2 * bytes, us);
  us = time_us([&] { hipLaunchKernelGGL((copy_k<uint32_t, true>), GRID(uint32_t), dim3(256), 0, 0, (const uint32_t*)a, (uint32_t*)b, bytes / 4); }, 5); REPORT("copy   4 B per lane, nt store", 2 * bytes, us);
  us = time_us([&] { hipLaunchKernelGGL((copy_k<uint32_t, false>), GRID(uint32_t), dim3(256), 0, 0, (const uint32_t*)a, (uint32_t*)b, bytes / 4); }, 5); REPORT("copy   4 B per lane, plain store", 2 * bytes, us);
  us = time_us([&] { hipLaunchKernelGGL((copy_k<u32x4, true>), GRID(u32x4), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, bytes / 16); }, 5); REPORT("copy  16 B per lane, nt store", 2 * bytes, us);
  us = time_us([&] { hipLaunchKernelGGL((copy_k<u32x4, false>), GRID(u32x4), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, bytes / 16); }, 5); REPORT("copy  16 B per lane, plain store", 2 * bytes, us);
  {
    const int H = 2560, W = 2560, D = 80, dc = 8;      // 2.1 GB
    const size_t vb = (size_t)D * H * W * 4;
    float* vol; CK(hipMalloc(&vol, vb));
    const dim3 grid(W / 128, H / 32, D / dc);
    us = time_us([&] { hipLaunchKernelGGL((tile_store_k<1>), grid, dim3(256), 0, 0, vol, H, W, D, dc); }, 5); REPORT("stack tiles, 16 x  4 B stores / lane", vb, us);
    us = time_us([&] { hipLaunchKernelGGL((tile_store_k<2>), grid, dim3(256), 0, 0, vol, H, W, D, dc); }, 5); REPORT("stack tiles,  8 x  8 B stores / lane", vb, us);
    us = time_us([&] { hipLaunchKernelGGL((tile_store_k<4>), grid, dim3(256), 0, 0, vol, H, W, D, dc); }, 5); REPORT("stack tiles,  4 x 16 B stores / lane", vb, us);
    us = time_us([&] { hipLaunchKernelGGL((tile_store_k<1, 64>), dim3(W / 64, H / 64, D / dc), dim3(256), 0, 0, vol, H, W, D, dc); }, 5); REPORT("stack tiles  64 x 64, 4 B stores", vb, us);
    us = time_us([&] { hipLaunchKernelGGL((tile_store_k<1, 256>), dim3(W / 256, H / 16, D / dc), dim3(256), 0, 0, vol, H, W, D, dc); }, 5); REPORT("stack tiles 256 x 16, 4 B stores", vb, us);
    us = time_us([&] { hipLaunchKernelGGL((tile_store_k<4, 256>), dim3(W / 256, H / 16, D / dc), dim3(256), 0, 0, vol, H, W, D, dc); }, 5); REPORT("stack tiles 256 x 16, 16 B stores", vb, us);
    us = time_us([&] { hipLaunchKernelGGL((tile_store_k<1>), dim3(W / 128, H / 32, D), dim3(256), 0, 0, vol, H, W, D, 1); }, 5); REPORT("stack tiles, ONE projection per workgroup", vb, us);
    us = time_us([&] { hipLaunchKernelGGL((tile_store_k<1>), dim3(W / 128, H / 32, D / 2), dim3(256), 0, 0, vol, H, W, D, 2); }, 5); REPORT("stack tiles, two projections per workgroup", vb, us);
    us = time_us([&] { hipLaunchKernelGGL((tile_store_k<1>), dim3(W / 128, H / 32, D / 40), dim3(256), 0, 0, vol, H, W, D, 40); }, 5); REPORT("stack tiles, 40 projections per workgroup", vb, us);
    us = time_us([&] { hipLaunchKernelGGL((wide_store_k<1>), dim3(W / 256, H / 16, D / dc), dim3(256), 0, 0, vol, H, W, D, dc); }, 5); REPORT("tiles  256 x 16, 1 KB row segments", vb, us);
    us = time_us([&] { hipLaunchKernelGGL((wide_store_k<2>), dim3(W / 512, H / 8, D / dc), dim3(256), 0, 0, vol, H, W, D, dc); }, 5); REPORT("tiles  512 x  8, 2 KB row segments", vb, us);
    us = time_us([&] { hipLaunchKernelGGL((wide_store_k<4>), dim3((W + 1023) / 1024, H / 4, D / dc), dim3(256), 0, 0, vol, H, W, D, dc); }, 5); REPORT("tiles 1024 x  4, 4 KB row segments", vb, us);
    CK(hipFree(vol));
  }
  us = time_us([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); }, 5); REPORT("hipMemcpyAsync device to device", 2 * bytes, us);
  return 0;
}
