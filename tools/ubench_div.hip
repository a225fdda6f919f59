// Correct rounding of the shared-reciprocal double division used by the perspective map:
// qx = nx/den, qy = ny/den with ONE refined reciprocal.  Compared bit for bit with the host's IEEE
// division on random operands in (and well beyond) the homography's range.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__device__ inline void div2_rn(double nx, double ny, double den, double* qx, double* qy) {
  double r = __builtin_amdgcn_rcp(den);             // ~2^-24
  double e = __builtin_fma(-den, r, 1.0);
  r = __builtin_fma(r, e, r);                        // ~2^-48
  e = __builtin_fma(-den, r, 1.0);
  r = __builtin_fma(r, e, r);                        // <= 1 ulp
  double q = nx * r;
  double t = __builtin_fma(-den, q, nx);
  *qx = __builtin_fma(t, r, q);
  q = ny * r;
  t = __builtin_fma(-den, q, ny);
  *qy = __builtin_fma(t, r, q);
}
__global__ void k(const double* nx, const double* ny, const double* den, double* qx, double* qy, double* lx, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { div2_rn(nx[i], ny[i], den[i], &qx[i], &qy[i]); lx[i] = nx[i] / den[i]; }
}
int main() {
  size_t n = (size_t)1 << 24;
  std::vector<double> a(n), b(n), d(n);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  auto u = [&]() { return (double)(rnd() >> 11) * 0x1p-53; };
  for (size_t i = 0; i < n; ++i) {
    int kind = i % 4;
    if (kind == 0) {           // homography-like: den ~ 1 +- 0.2, numerators up to image size
      d[i] = 1.0 + (u() - 0.5) * 0.4; a[i] = (u() - 0.1) * 9000.0; b[i] = (u() - 0.1) * 9000.0;
    } else if (kind == 1) {    // integer-ish grid points through a real homography
      double x = (double)(rnd() % 8192), y = (double)(rnd() % 8192);
      d[i] = (-8.075209829141167e-06 * x + -1.0417072082535193e-05 * y) + 1.0;
      a[i] = (0.9450284704184375 * x + -0.019662775048787898 * y) + 55.99511925916719;
      b[i] = (-0.01478311636447244 * x + 0.9403850653789713 * y) + 45.65706672670265;
    } else if (kind == 2) {    // wide dynamic range
      d[i] = ldexp(1.0 + u(), (int)(rnd() % 60) - 30) * ((rnd() & 1) ? 1 : -1);
      a[i] = ldexp(1.0 + u(), (int)(rnd() % 80) - 40) * ((rnd() & 1) ? 1 : -1);
      b[i] = ldexp(1.0 + u(), (int)(rnd() % 80) - 40);
    } else {                   // exact and near-exact quotients
      double q = (double)(rnd() % 1000000) / 64.0; d[i] = 1.0 + (double)(rnd() % 4096) / 4096.0;
      a[i] = q * d[i]; b[i] = nextafter(a[i], 1e300);
    }
  }
  a[0] = 0.0; b[0] = 1.0; d[0] = 1.0; a[1] = 5.0; b[1] = -5.0; d[1] = 1.0; a[2] = 1.0; b[2] = 3.0; d[2] = 3.0;
  double *da, *db, *dd, *qx, *qy, *lx;
  for (double** p : {&da, &db, &dd, &qx, &qy, &lx}) CK(hipMalloc(p, n * 8));
  CK(hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dd, d.data(), n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, da, db, dd, qx, qy, lx, n);
  CK(hipDeviceSynchronize());
  std::vector<double> rx(n), ry(n), rl(n);
  CK(hipMemcpy(rx.data(), qx, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(ry.data(), qy, n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(rl.data(), lx, n * 8, hipMemcpyDeviceToHost));
  size_t badx = 0, bady = 0, badl = 0;
  for (size_t i = 0; i < n; ++i) {
    double ex = a[i] / d[i], ey = b[i] / d[i];
    if (rx[i] != ex) { if (badx < 5) printf("x mismatch kind %zu: %a / %a = %a got %a\n", i % 4, a[i], d[i], ex, rx[i]); badx++; }
    if (ry[i] != ey) { if (bady < 5) printf("y mismatch kind %zu: %a / %a = %a got %a\n", i % 4, b[i], d[i], ey, ry[i]); bady++; }
    if (rl[i] != ex) badl++;
  }
  printf("shared-reciprocal division over %zu triples: x mismatches %zu, y mismatches %zu; compiler division mismatches %zu\n", n, badx, bady, badl);
  return 0;
}
