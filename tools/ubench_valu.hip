// ubench_valu.hip -- issue cost of single gfx950 VALU / LDS instructions, relative to v_fma_f32 (2 cycles per wave64 on a SIMD-32,
// MI355X_MICROARCH.md): every SIMD runs 8 waves that issue long runs of ONE instruction on four independent register sets.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_valu.hip -o tools/ubench_valu && tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define REP4(X) X(0) X(1) X(2) X(3)
#define REP16(X) REP4(X) REP4(X) REP4(X) REP4(X)

#define KERNEL(NAME, DECL, BODY, SINK)                                         \
  __global__ void __launch_bounds__(256) NAME(double* out, int iters) {        \
    DECL                                                                       \
    for (int i = 0; i < iters; ++i) {                                          \
      REP16(BODY)                                                              \
    }                                                                          \
    out[blockIdx.x * 256 + threadIdx.x] = SINK;                                \
  }

#define DECL_D double d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, e0 = 1.0000001, e1 = 0.5; \
  unsigned u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7; float f0 = u0, f1 = u1, f2 = u2, f3 = u3; int sh = threadIdx.x & 31;
#define SINK_ALL (d0 + d1 + d2 + d3 + (double)(u0 + u1 + u2 + u3) + (double)(f0 + f1 + f2 + f3))

#define B_FMA32(n) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f##n) : "v"(f0));
#define B_FMA64(n) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d##n) : "v"(e0));
#define B_ADD64(n) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d##n) : "v"(e0));
#define B_MUL64(n) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d##n) : "v"(e0));
#define B_CVT_F64_U32(n) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d##n) : "v"(u##n));
#define B_CVT_F64_I32(n) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d##n) : "v"(u##n));
#define B_CVT_U32_F64(n) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(u##n) : "v"(d##n));
#define B_CVT_F64_F32(n) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d##n) : "v"(f##n));
#define B_CVT_F32_F64(n) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f##n) : "v"(d##n));
#define B_CVT_F32_U32(n) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f##n) : "v"(u##n));
#define B_CVT_F32_UB1(n) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(f##n) : "v"(u##n));
#define B_FLOOR64(n) asm volatile("v_floor_f64 %0, %1" : "=v"(d##n) : "v"(d##n));
#define B_FRACT32(n) asm volatile("v_fract_f32 %0, %1" : "=v"(f##n) : "v"(f##n));
#define B_ALIGNBIT(n) asm volatile("v_alignbit_b32 %0, %1, %2, %3" : "=v"(u##n) : "v"(u##n), "v"(u0), "v"(sh));
#define B_SUB_SDWA(n) asm volatile("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0" : "=v"(u##n) : "v"(u##n), "v"(u0));
#define B_AND(n) asm volatile("v_and_b32 %0, 0xffff, %1" : "=v"(u##n) : "v"(u##n));
#define B_RSQ64(n) asm volatile("v_rsq_f64 %0, %1" : "=v"(d##n) : "v"(d##n));
#define B_RCP64(n) asm volatile("v_rcp_f64 %0, %1" : "=v"(d##n) : "v"(d##n));
#define B_MIN_U32(n) asm volatile("v_min_u32 %0, %1, %2" : "=v"(u##n) : "v"(u##n), "v"(u0));
#define B_CVT_I32_F32(n) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u##n) : "v"(f##n));
#define B_MED3(n) asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(f##n) : "v"(f##n), "v"(f0), "v"(f1));
#define B_MAD_U24(n) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(u##n) : "v"(u##n), "v"(u0), "v"(u1));
#define B_PK_FMA32(n) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(d##n) : "v"(e0));

KERNEL(k_fma32, DECL_D, B_FMA32, SINK_ALL)
KERNEL(k_fma64, DECL_D, B_FMA64, SINK_ALL)
KERNEL(k_add64, DECL_D, B_ADD64, SINK_ALL)
KERNEL(k_mul64, DECL_D, B_MUL64, SINK_ALL)
KERNEL(k_cvt_f64_u32, DECL_D, B_CVT_F64_U32, SINK_ALL)
KERNEL(k_cvt_f64_i32, DECL_D, B_CVT_F64_I32, SINK_ALL)
KERNEL(k_cvt_u32_f64, DECL_D, B_CVT_U32_F64, SINK_ALL)
KERNEL(k_cvt_f64_f32, DECL_D, B_CVT_F64_F32, SINK_ALL)
KERNEL(k_cvt_f32_f64, DECL_D, B_CVT_F32_F64, SINK_ALL)
KERNEL(k_cvt_f32_u32, DECL_D, B_CVT_F32_U32, SINK_ALL)
KERNEL(k_cvt_f32_ubyte1, DECL_D, B_CVT_F32_UB1, SINK_ALL)
KERNEL(k_floor64, DECL_D, B_FLOOR64, SINK_ALL)
KERNEL(k_fract32, DECL_D, B_FRACT32, SINK_ALL)
KERNEL(k_alignbit, DECL_D, B_ALIGNBIT, SINK_ALL)
KERNEL(k_sub_sdwa, DECL_D, B_SUB_SDWA, SINK_ALL)
KERNEL(k_and, DECL_D, B_AND, SINK_ALL)
KERNEL(k_rsq64, DECL_D, B_RSQ64, SINK_ALL)
KERNEL(k_rcp64, DECL_D, B_RCP64, SINK_ALL)
KERNEL(k_min_u32, DECL_D, B_MIN_U32, SINK_ALL)
KERNEL(k_cvt_i32_f32, DECL_D, B_CVT_I32_F32, SINK_ALL)
KERNEL(k_med3_f32, DECL_D, B_MED3, SINK_ALL)
KERNEL(k_mad_u24, DECL_D, B_MAD_U24, SINK_ALL)
KERNEL(k_pk_fma32, DECL_D, B_PK_FMA32, SINK_ALL)

typedef void (*kern_t)(double*, int);
struct Case { const char* name; kern_t fn; };

int main() {
  const int blocks = 256 * 8, iters = 2000;       // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  double* out;
  if (hipMalloc(&out, sizeof(double) * blocks * 256) != hipSuccess) { printf("no device\n"); return 1; }
  Case cases[] = {{"v_fma_f32", k_fma32}, {"v_fma_f64", k_fma64}, {"v_add_f64", k_add64}, {"v_mul_f64", k_mul64},
                  {"v_cvt_f64_u32", k_cvt_f64_u32}, {"v_cvt_f64_i32", k_cvt_f64_i32}, {"v_cvt_u32_f64", k_cvt_u32_f64},
                  {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_cvt_f32_f64", k_cvt_f32_f64}, {"v_cvt_f32_u32", k_cvt_f32_u32},
                  {"v_cvt_f32_ubyte1", k_cvt_f32_ubyte1}, {"v_floor_f64", k_floor64}, {"v_fract_f32", k_fract32},
                  {"v_alignbit_b32", k_alignbit}, {"v_sub_u32_sdwa", k_sub_sdwa}, {"v_and_b32", k_and}, {"v_rsq_f64", k_rsq64},
                  {"v_rcp_f64", k_rcp64}, {"v_min_u32", k_min_u32}, {"v_cvt_i32_f32", k_cvt_i32_f32}, {"v_med3_f32", k_med3_f32},
                  {"v_mad_u32_u24", k_mad_u24}, {"v_pk_fma_f32", k_pk_fma32}};
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  double base = 0;
  for (int pass = 0; pass < 2; ++pass)
    for (auto& c : cases) {
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, out, iters);
      hipEventRecord(e0, 0);
      for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double per = ms / 5.0;
      if (!strcmp(c.name, "v_fma_f32")) base = per;
      if (pass == 1) printf("%-18s %8.3f ms   %5.2f cycles per wave-instruction (v_fma_f32 = 2)\n", c.name, per, 2.0 * per / base);
    }
  return 0;
}
