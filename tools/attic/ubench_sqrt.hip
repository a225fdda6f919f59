// ubench_sqrt.hip -- how often does the kernels' sqrt (dcp_device.h: v_rsq_f64, one coupled Newton step, one residual correction)
// differ from the correctly rounded result, with and without the refinement of h = 1 / (2 sqrt x)?  Operands: r^2 of a frame,
// uniform in (0, 3.4e7).      hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/ubench_sqrt.hip -o tools/ubench_sqrt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ double splitmix(uint64_t& s) {
  s += 0x9E3779B97F4A7C15ull;
  uint64_t z = s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

template <bool REFINE_H>
__global__ void check(unsigned long long* bad, int iters, uint64_t seed) {
  uint64_t s = seed + (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x100000001B3ull;
  unsigned long long nbad = 0;
  for (int i = 0; i < iters; ++i) {
    const double x = 1e-3 + splitmix(s) * 3.4e7;
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    if (REFINE_H) h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    if (g != __builtin_sqrt(x)) ++nbad;
  }
  if (nbad) atomicAdd(bad, nbad);
}

int main() {
  unsigned long long *bad, h;
  hipMalloc(&bad, 8);
  const int blocks = 2048, iters = 8192;
  for (int refine = 1; refine >= 0; --refine) {
    hipMemset(bad, 0, 8);
    if (refine) hipLaunchKernelGGL(check<true>, dim3(blocks), dim3(256), 0, 0, bad, iters, 777ull);
    else hipLaunchKernelGGL(check<false>, dim3(blocks), dim3(256), 0, 0, bad, iters, 777ull);
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    printf("h %s: %llu of %.3g results differ from the correctly rounded sqrt\n", refine ? "refined  " : "unrefined", h, (double)blocks * 256 * iters);
  }
  return 0;
}
