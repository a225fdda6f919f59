// Microbenchmarks that set the budget for the unwarp kernels on gfx950:
//  (1) VALU issue cost of the fp64 / conversion instructions the coordinate
//      polynomial needs, (2) achievable HBM copy bandwidth for 4 B/lane and
//      16 B/lane streams, (3) correct rounding of the hand-rolled fp64 sqrt.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench tools/ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 2048;
constexpr int CHAINS = 8;

#define OPKERNEL0(NAME, T, ASMSTR)                                                \
__global__ void __launch_bounds__(256) k_##NAME(double* out, double seed) {      \
  T a[CHAINS]; for (int c = 0; c < CHAINS; ++c) a[c] = (T)seed + (T)c;           \
  for (int i = 0; i < ITERS; ++i) {                                              \
    _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) asm volatile(ASMSTR : "+v"(a[c])); \
  }                                                                              \
  double s = 0; for (int c = 0; c < CHAINS; ++c) s += (double)a[c];              \
  if (s == 12345.678) out[threadIdx.x] = s;                                      \
}
#define OPKERNEL2(NAME, T, ASMSTR)                                                \
__global__ void __launch_bounds__(256) k_##NAME(double* out, double seed) {      \
  T a[CHAINS]; T b = (T)seed; T cc = (T)(seed * 0.5);                            \
  for (int c = 0; c < CHAINS; ++c) a[c] = (T)seed + (T)c;                        \
  for (int i = 0; i < ITERS; ++i) {                                              \
    _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) asm volatile(ASMSTR : "+v"(a[c]) : "v"(b), "v"(cc)); \
  }                                                                              \
  double s = 0; for (int c = 0; c < CHAINS; ++c) s += (double)a[c];              \
  if (s == 12345.678) out[threadIdx.x] = s;                                      \
}

OPKERNEL2(fma_f64, double, "v_fma_f64 %0, %0, %1, %2")
OPKERNEL2(add_f64, double, "v_add_f64 %0, %0, %1")
OPKERNEL2(mul_f64, double, "v_mul_f64 %0, %0, %1")
OPKERNEL2(max_f64, double, "v_max_f64 %0, %0, %1")
OPKERNEL0(rsq_f64, double, "v_rsq_f64 %0, %0")
OPKERNEL0(rcp_f64, double, "v_rcp_f64 %0, %0")
OPKERNEL0(sqrt_f64, double, "v_sqrt_f64 %0, %0")
OPKERNEL0(floor_f64, double, "v_floor_f64 %0, %0")
OPKERNEL2(fma_f32, float, "v_fma_f32 %0, %0, %1, %2")
OPKERNEL0(rsq_f32, float, "v_rsq_f32 %0, %0")
OPKERNEL0(floor_f32, float, "v_floor_f32 %0, %0")
OPKERNEL2(med3_f32, float, "v_med3_f32 %0, %0, %1, %2")
OPKERNEL2(mad_u32, unsigned, "v_mad_u32_u24 %0, %0, %1, %2")

// conversions: chain through a pair so each op depends on the previous one in its chain
__global__ void __launch_bounds__(256) k_cvt_f32_f64(double* out, double seed) {
  double a[CHAINS]; float f[CHAINS];
  for (int c = 0; c < CHAINS; ++c) a[c] = seed + c;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[c]) : "v"(a[c]));
  }
  double s = 0; for (int c = 0; c < CHAINS; ++c) s += f[c];
  if (s == 12345.678) out[threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_cvt_f64_f32(double* out, double seed) {
  double a[CHAINS]; float f[CHAINS];
  for (int c = 0; c < CHAINS; ++c) f[c] = (float)seed + c;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[c]) : "v"(f[c]));
  }
  double s = 0; for (int c = 0; c < CHAINS; ++c) s += a[c];
  if (s == 12345.678) out[threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_cvt_f64_i32(double* out, double seed) {
  double a[CHAINS]; int f[CHAINS];
  for (int c = 0; c < CHAINS; ++c) f[c] = (int)seed + c;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[c]) : "v"(f[c]));
  }
  double s = 0; for (int c = 0; c < CHAINS; ++c) s += a[c];
  if (s == 12345.678) out[threadIdx.x] = s;
}

template <typename T>
__global__ void __launch_bounds__(256) k_copy(const T* __restrict__ src, T* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = src[i];
}
template <typename T>
__global__ void __launch_bounds__(256) k_copy_nt(const T* __restrict__ src, T* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(&src[i]), &dst[i]);
}
template <typename T>
__global__ void __launch_bounds__(256) k_write(T* __restrict__ dst, size_t n, T v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = v;
}
__global__ void __launch_bounds__(256) k_read(const float4* __restrict__ src, float* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  float acc = 0;
  for (; i < n; i += stride) { float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 12345.678f) dst[0] = acc;
}

// hand-rolled correctly-rounded-candidate sqrt (no range scaling; 0 handled by select)
__device__ inline double sqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return x == 0.0 ? 0.0 : g;
}
__device__ inline double sqrt_nr1(double x) {  // one correction round fewer
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return x == 0.0 ? 0.0 : g;
}
__device__ inline double sqrt_f32seed(double x) {  // f32 rsq seed (~22 bits) + 2 coupled rounds + 1 fixup
  float xf = (float)x;
  double y = (double)__builtin_amdgcn_rsqf(xf);
  double g = x * y;
  double h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return x == 0.0 ? 0.0 : g;
}
__global__ void k_sqrt_check(const double* in, double* o_nr, double* o_nr1, double* o_f32, double* o_lib, double* o_rsq, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    double x = in[i];
    o_nr[i] = sqrt_nr(x); o_nr1[i] = sqrt_nr1(x); o_f32[i] = sqrt_f32seed(x); o_lib[i] = sqrt(x);
    o_rsq[i] = __builtin_amdgcn_rsq(x);
  }
}

template <typename F>
static float time_ms(F&& f, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz memclk %d kHz\n", p.name, p.multiProcessorCount, p.clockRate, p.memoryClockRate);
  double* dout; CK(hipMalloc(&dout, 4096));
  const int blocks = p.multiProcessorCount * 8;  // 8 blocks x 4 waves = 32 waves/CU = 8 waves/SIMD
  const double wave_instr_per_simd = (double)ITERS * CHAINS * 8;  // 8 waves per SIMD each ITERS*CHAINS
#define RUNOP(NAME) { float ms = time_ms([&]{ hipLaunchKernelGGL(k_##NAME, dim3(blocks), dim3(256), 0, 0, dout, 1.5); }, 5); \
    double cyc = ms * 1e-3 * 2.4e9 / wave_instr_per_simd; \
    printf("%-14s %8.3f ms  -> %.2f cyc/wave-instr/SIMD @2.4GHz (%.1f Gops/s/lane-total)\n", #NAME, ms, cyc, wave_instr_per_simd*4*p.multiProcessorCount*64/ms/1e6); }
  RUNOP(fma_f64) RUNOP(add_f64) RUNOP(mul_f64) RUNOP(max_f64) RUNOP(rsq_f64) RUNOP(rcp_f64) RUNOP(sqrt_f64) RUNOP(floor_f64)
  RUNOP(fma_f32) RUNOP(rsq_f32) RUNOP(floor_f32) RUNOP(med3_f32) RUNOP(mad_u32) RUNOP(cvt_f32_f64) RUNOP(cvt_f64_f32) RUNOP(cvt_f64_i32)

  // bandwidth
  size_t bytes = (size_t)2 << 30;
  void *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
  for (int gb : {2048, 4096, 8192, 16384, 65536}) {
    float ms;
    ms = time_ms([&]{ hipLaunchKernelGGL(k_copy<float>, dim3(gb), dim3(256), 0, 0, (const float*)a, (float*)b, bytes/4); }, 5);
    printf("copy f32   grid %6d: %.3f ms %.1f GB/s\n", gb, ms, 2.0*bytes/ms/1e6);
    ms = time_ms([&]{ hipLaunchKernelGGL(k_copy<float4>, dim3(gb), dim3(256), 0, 0, (const float4*)a, (float4*)b, bytes/16); }, 5);
    printf("copy f32x4 grid %6d: %.3f ms %.1f GB/s\n", gb, ms, 2.0*bytes/ms/1e6);
    ms = time_ms([&]{ hipLaunchKernelGGL(k_copy_nt<float>, dim3(gb), dim3(256), 0, 0, (const float*)a, (float*)b, bytes/4); }, 5);
    printf("copyNT f32 grid %6d: %.3f ms %.1f GB/s\n", gb, ms, 2.0*bytes/ms/1e6);
  }
  {
    float ms = time_ms([&]{ hipLaunchKernelGGL(k_write<float>, dim3(8192), dim3(256), 0, 0, (float*)b, bytes/4, 1.0f); }, 5);
    printf("write f32: %.3f ms %.1f GB/s\n", ms, 1.0*bytes/ms/1e6);
    ms = time_ms([&]{ hipLaunchKernelGGL(k_write<float4>, dim3(8192), dim3(256), 0, 0, (float4*)b, bytes/16, make_float4(1,2,3,4)); }, 5);
    printf("write f32x4: %.3f ms %.1f GB/s\n", ms, 1.0*bytes/ms/1e6);
    ms = time_ms([&]{ hipLaunchKernelGGL(k_read, dim3(8192), dim3(256), 0, 0, (const float4*)a, (float*)b, bytes/16); }, 5);
    printf("read f32x4: %.3f ms %.1f GB/s\n", ms, 1.0*bytes/ms/1e6);
  }
  // small working set (fits the 256 MiB infinity cache): 64 MiB in + 64 MiB out, like one 4096^2 frame
  {
    size_t fb = (size_t)64 << 20;
    float ms = time_ms([&]{ hipLaunchKernelGGL(k_copy<float4>, dim3(8192), dim3(256), 0, 0, (const float4*)a, (float4*)b, fb/16); }, 20);
    printf("copy f32x4 64MiB frame resident in MALL: %.4f ms %.1f GB/s\n", ms, 2.0*fb/ms/1e6);
    ms = time_ms([&]{ hipLaunchKernelGGL(k_copy<float>, dim3(16384), dim3(256), 0, 0, (const float*)a, (float*)b, fb/4); }, 20);
    printf("copy f32 64MiB frame resident in MALL: %.4f ms %.1f GB/s\n", ms, 2.0*fb/ms/1e6);
  }

  // sqrt correctness: r2 = xu^2 + yu^2 style inputs and random doubles over many magnitudes
  size_t n = (size_t)1 << 24;
  std::vector<double> h(n);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (size_t i = 0; i < n; ++i) {
    if (i % 4 == 0) { double xu = (double)(rnd() % 8192) - 1883.8169650464; double yu = (double)(rnd() % 8192) - 1478.6964217312; h[i] = xu*xu + yu*yu; }
    else if (i % 4 == 1) { h[i] = ldexp(1.0 + (double)(rnd() >> 11) * 0x1p-53, (int)(rnd() % 120) - 60); }
    else if (i % 4 == 2) { double q = (double)(rnd() % 100000000) ; h[i] = q * q + (double)((int)(rnd()%3) - 1); if (h[i] < 0) h[i] = 0; }
    else { h[i] = (double)(rnd() >> 11) * 0x1p-53 * 7e7; }
  }
  h[0] = 0.0; h[1] = 1.0; h[2] = 4.0; h[3] = 2.0; h[5] = 5e-324; h[6] = 1e-300; h[7] = 1e300;
  double *din, *o1, *o2, *o3, *o4, *o5;
  CK(hipMalloc(&din, n*8)); CK(hipMalloc(&o1, n*8)); CK(hipMalloc(&o2, n*8)); CK(hipMalloc(&o3, n*8)); CK(hipMalloc(&o4, n*8)); CK(hipMalloc(&o5, n*8));
  CK(hipMemcpy(din, h.data(), n*8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_sqrt_check, dim3((n+255)/256), dim3(256), 0, 0, din, o1, o2, o3, o4, o5, n);
  CK(hipDeviceSynchronize());
  std::vector<double> r1(n), r2(n), r3(n), r4(n), r5(n);
  CK(hipMemcpy(r1.data(), o1, n*8, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), o2, n*8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(r3.data(), o3, n*8, hipMemcpyDeviceToHost)); CK(hipMemcpy(r4.data(), o4, n*8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(r5.data(), o5, n*8, hipMemcpyDeviceToHost));
  size_t bad1=0,bad2=0,bad3=0,bad4=0; double maxrsq = 0;
  for (size_t i = 8; i < n; ++i) {
    double ref = sqrt(h[i]);
    if (r1[i] != ref) { if (bad1 < 3) printf("nr  mismatch x=%a got %a ref %a\n", h[i], r1[i], ref); bad1++; }
    if (r2[i] != ref) bad2++;
    if (r3[i] != ref) { if (bad3 < 3) printf("f32 mismatch x=%a got %a ref %a\n", h[i], r3[i], ref); bad3++; }
    if (r4[i] != ref) bad4++;
    if (h[i] > 0) { double e = fabs(r5[i] * ref - 1.0); if (e > maxrsq) maxrsq = e; }
  }
  printf("sqrt check over %zu values: nr(2 fixups) bad=%zu, nr1(1 fixup) bad=%zu, f32seed bad=%zu, libsqrt bad=%zu, rsq_f64 max rel err=%.3e\n", n-8, bad1, bad2, bad3, bad4, maxrsq);
  for (int i = 0; i < 8; ++i) printf("special x=%g nr=%g nr1=%g f32=%g lib=%g ref=%g\n", h[i], r1[i], r2[i], r3[i], r4[i], sqrt(h[i]));
  return 0;
}
