// ubench_div.hip -- is nx / den from ONE refined reciprocal correctly rounded?  v_rcp_f64 + `steps` Newton steps, then the
// Markstein correction per quotient (dcp_device.h: div2_rn), against the IEEE division, on random operands of the magnitudes
// the homography of postprocessing.py:453-455 produces (numerators up to ~1e4, denominators around 1).
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/ubench_div.hip -o tools/ubench_div && tools/ubench_div
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

template <int STEPS>
__global__ void check(unsigned long long* bad, unsigned long long* worst_ulp, int iters, uint64_t seed) {
  uint64_t s = seed + (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x100000001B3ull;
  unsigned long long nbad = 0;
  for (int i = 0; i < iters; ++i) {
    const double den = 0.5 + 1.5 * splitmix(s);                 // c7 x + c8 y + 1 over a frame: around 1
    const double nx = (splitmix(s) - 0.1) * 9000.0;
    double r = __builtin_amdgcn_rcp(den);
#pragma unroll
    for (int k = 0; k < STEPS; ++k) {
      const double e = __builtin_fma(-den, r, 1.0);
      r = __builtin_fma(r, e, r);
    }
    double q = nx * r;
    const double t = __builtin_fma(-den, q, nx);
    q = __builtin_fma(t, r, q);
    const double want = nx / den;
    if (q != want) ++nbad;
  }
  if (nbad) atomicAdd(bad, nbad);
}

int main() {
  unsigned long long *bad, h[2];
  hipMalloc(&bad, 16);
  for (int steps = 1; steps <= 2; ++steps) {
    hipMemset(bad, 0, 16);
    const int blocks = 2048, iters = 4096;
    if (steps == 1) hipLaunchKernelGGL(check<1>, dim3(blocks), dim3(256), 0, 0, bad, bad + 1, iters, 12345ull);
    else hipLaunchKernelGGL(check<2>, dim3(blocks), dim3(256), 0, 0, bad, bad + 1, iters, 12345ull);
    hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    printf("%d Newton step(s): %llu quotients differ from the IEEE division out of %.3g\n", steps, h[0], (double)blocks * 256 * iters);
  }
  return 0;
}
