// Known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// (MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of a wide coalesced read; other widths
// are uncalibrated).  Each kernel moves exactly 1 GiB in and 1 GiB out.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) calib_copy_dword(const float* __restrict__ s, float* __restrict__ d, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) d[i] = s[i];
}
__global__ void __launch_bounds__(256) calib_copy_dwordx4(const float4* __restrict__ s, float4* __restrict__ d, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) d[i] = s[i];
}
// the unwarp gather's shape: 8-byte loads at 4-byte lane stride (every element read twice), 4-byte stores
__global__ void __launch_bounds__(256) calib_gather_dwordx2(const float* __restrict__ s, float* __restrict__ d, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i + 1 < n) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(s + (size_t)blockIdx.x * 256), 0, 1028, 0x00020000);
    u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, threadIdx.x * 4, 0, 0);
    d[i] = __uint_as_float(v.x) + __uint_as_float(v.y);
  }
}
int main() {
  size_t bytes = (size_t)1 << 30, n = bytes / 4;
  float *a, *b; CK(hipMalloc(&a, bytes + 64)); CK(hipMalloc(&b, bytes + 64));
  CK(hipMemset(a, 0, bytes + 64)); CK(hipMemset(b, 0, bytes + 64));
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(calib_copy_dword, dim3((unsigned)(n / 256)), dim3(256), 0, 0, a, b, n);
    hipLaunchKernelGGL(calib_copy_dwordx4, dim3((unsigned)(n / 4 / 256)), dim3(256), 0, 0, (const float4*)a, (float4*)b, n / 4);
    hipLaunchKernelGGL(calib_gather_dwordx2, dim3((unsigned)(n / 256)), dim3(256), 0, 0, a, b, n);
  }
  CK(hipDeviceSynchronize());
  printf("calibration kernels done: %zu bytes read and written per launch\n", bytes);
  return 0;
}
