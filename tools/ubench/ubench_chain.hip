// ubench_chain.hip -- latency of DEPENDENT float64 operations on gfx950, by waves per SIMD: what bounds the recursive B-spline
// prefilter (t = x lam + z t: one v_mul_f64 and one v_add_f64 on the critical path per sample, two waves per SIMD in
// spline_prefilter2d_kernel).  Every wave runs ONE chain of N dependent (mul, add) pairs -- or, with ILP = 2, two independent chains
// interleaved -- and the launch holds exactly W waves per SIMD (one round, no LDS).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_chain.hip -o /tmp/ubench_chain && /tmp/ubench_chain
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ILP, bool FMA>
__global__ void __launch_bounds__(64) chain_kernel(double* out, int n, double z, double x) {
  double t0 = threadIdx.x * 1e-3, t1 = t0 + 0.5;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if constexpr (FMA) {
        asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(t0) : "v"(z), "v"(x));
        if constexpr (ILP == 2) asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(t1) : "v"(z), "v"(x));
      } else {
        asm volatile("v_mul_f64 %0, %1, %0\n\tv_add_f64 %0, %0, %2" : "+v"(t0) : "v"(z), "v"(x));
        if constexpr (ILP == 2) asm volatile("v_mul_f64 %0, %1, %0\n\tv_add_f64 %0, %0, %2" : "+v"(t1) : "v"(z), "v"(x));
      }
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = t0 + t1;
}

template <int ILP, bool FMA>
static void run(const char* what, int waves_per_simd, int ncu, double* out) {
  const int n = 4096;                                   // x 16 steps
  const int blocks = ncu * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((chain_kernel<ILP, FMA>), dim3(blocks), dim3(64), 0, 0, out, n, -0.2679491924311227, 0.3);
  hipEventRecord(e0, 0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((chain_kernel<ILP, FMA>), dim3(blocks), dim3(64), 0, 0, out, n, -0.2679491924311227, 0.3);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double steps = (double)n * 16.0;                // recursion steps per chain
  const double ns_step = ms * 1e6 / reps / steps;
  printf("%-34s %d wave(s) per SIMD, %d chain(s) per wave: %7.2f ns per step of one chain, %6.2f ns per step and SIMD\n", what, waves_per_simd, ILP, ns_step,
         ns_step / (waves_per_simd * ILP));
}

int main() {
  int ncu = 256;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  double* out;
  hipMalloc(&out, (size_t)ncu * 4 * 16 * 64 * sizeof(double));
  for (int w : {1, 2, 3, 4, 6, 8}) run<1, false>("t = z t + x as v_mul_f64, v_add_f64", w, ncu, out);
  for (int w : {1, 2, 4}) run<2, false>("the same, two chains interleaved", w, ncu, out);
  for (int w : {1, 2, 4, 8}) run<1, true>("t = fma(z, t, x)", w, ncu, out);
  hipFree(out);
  return 0;
}
