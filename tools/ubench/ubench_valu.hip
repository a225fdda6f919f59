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

#define B_ADD_U32(n) asm volatile("v_add_u32 %0, %1, %2" : "=v"(u##n) : "v"(u##n), "v"(u0));
#define B_LSHL(n) asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(u##n) : "v"(u##n));
#define B_LSHL_OR(n) asm volatile("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(u##n) : "v"(u##n), "v"(u0));
#define B_LSHL_ADD(n) asm volatile("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(u##n) : "v"(u##n), "v"(u0));
#define B_ADD3(n) asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(u##n) : "v"(u##n), "v"(u0), "v"(u1));
#define B_MUL32(n) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(f##n) : "v"(f##n), "v"(f0));
#define B_ADD32(n) asm volatile("v_add_f32 %0, %1, %2" : "=v"(f##n) : "v"(f##n), "v"(f0));
#define B_FMAC32(n) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f##n) : "v"(f0), "v"(f1));
#define B_MOV(n) asm volatile("v_mov_b32 %0, %1" : "=v"(u##n) : "v"(u0));
#define B_CNDMASK(n) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u##n) : "v"(u##n), "v"(u0) : "vcc");
#define B_BFE(n) asm volatile("v_bfe_u32 %0, %1, 16, 16" : "=v"(u##n) : "v"(u##n));
#define B_MUL_U24(n) asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(u##n) : "v"(u##n), "v"(u0));
#define B_MUL_LO(n) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(u##n) : "v"(u##n), "v"(u0));
#define B_DOT2(n) asm volatile("v_dot2_u32_u16 %0, %1, %2, %3" : "=v"(u##n) : "v"(u##n), "v"(u0), "v"(u1));
#define B_PERM(n) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u##n) : "v"(u##n), "v"(u0), "v"(u1));
#define B_MAD_U64(n) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d##n) : "v"(u0), "v"(u1) : "vcc");
#define B_CVT_U32_F32(n) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(u##n) : "v"(f##n));
#define B_CVT_F32_I32(n) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f##n) : "v"(u##n));
#define B_PK_ADD32(n) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d##n) : "v"(e0));
#define B_PK_MUL32(n) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d##n) : "v"(e0));
#define B_MAX32(n) asm volatile("v_max_f32 %0, %1, %2" : "=v"(f##n) : "v"(f##n), "v"(f0));
#define B_MIN_I32(n) asm volatile("v_min_i32 %0, %1, %2" : "=v"(u##n) : "v"(u##n), "v"(u0));
#define B_SUB32(n) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(f##n) : "v"(f##n), "v"(f0));
#define B_FLOOR32(n) asm volatile("v_floor_f32 %0, %1" : "=v"(f##n) : "v"(f##n));
#define B_LSHR(n) asm volatile("v_lshrrev_b32 %0, 15, %1" : "=v"(u##n) : "v"(u##n));
#define B_XOR(n) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(u##n) : "v"(u##n), "v"(u0));
#define B_FMA32_3(n) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f##n) : "v"(f##n), "v"(f0), "v"(f1));
#define B_MAD_I24(n) asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(u##n) : "v"(u##n), "v"(u0), "v"(u1));
#define B_SUB_U32(n) asm volatile("v_sub_u32 %0, %1, %2" : "=v"(u##n) : "v"(u##n), "v"(u0));

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

KERNEL(k2_0, DECL_D, B_ADD_U32, SINK_ALL)
KERNEL(k2_1, DECL_D, B_SUB_U32, SINK_ALL)
KERNEL(k2_2, DECL_D, B_LSHL, SINK_ALL)
KERNEL(k2_3, DECL_D, B_LSHR, SINK_ALL)
KERNEL(k2_4, DECL_D, B_XOR, SINK_ALL)
KERNEL(k2_5, DECL_D, B_LSHL_OR, SINK_ALL)
KERNEL(k2_6, DECL_D, B_LSHL_ADD, SINK_ALL)
KERNEL(k2_7, DECL_D, B_ADD3, SINK_ALL)
KERNEL(k2_8, DECL_D, B_MUL32, SINK_ALL)
KERNEL(k2_9, DECL_D, B_ADD32, SINK_ALL)
KERNEL(k2_10, DECL_D, B_SUB32, SINK_ALL)
KERNEL(k2_11, DECL_D, B_MAX32, SINK_ALL)
KERNEL(k2_12, DECL_D, B_FLOOR32, SINK_ALL)
KERNEL(k2_13, DECL_D, B_FMAC32, SINK_ALL)
KERNEL(k2_14, DECL_D, B_FMA32_3, SINK_ALL)
KERNEL(k2_15, DECL_D, B_MOV, SINK_ALL)
KERNEL(k2_16, DECL_D, B_CNDMASK, SINK_ALL)
KERNEL(k2_17, DECL_D, B_BFE, SINK_ALL)
KERNEL(k2_18, DECL_D, B_MUL_U24, SINK_ALL)
KERNEL(k2_19, DECL_D, B_MAD_I24, SINK_ALL)
KERNEL(k2_20, DECL_D, B_MUL_LO, SINK_ALL)
KERNEL(k2_21, DECL_D, B_DOT2, SINK_ALL)
KERNEL(k2_22, DECL_D, B_PERM, SINK_ALL)
KERNEL(k2_23, DECL_D, B_MAD_U64, SINK_ALL)
KERNEL(k2_24, DECL_D, B_CVT_U32_F32, SINK_ALL)
KERNEL(k2_25, DECL_D, B_CVT_F32_I32, SINK_ALL)
KERNEL(k2_26, DECL_D, B_PK_ADD32, SINK_ALL)
KERNEL(k2_27, DECL_D, B_PK_MUL32, SINK_ALL)
KERNEL(k2_28, DECL_D, B_MIN_I32, SINK_ALL)

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
                  {"v_mad_u32_u24", k_mad_u24}, {"v_pk_fma_f32", k_pk_fma32},
                  {"v_add_u32", k2_0}, {"v_sub_u32", k2_1}, {"v_lshlrev_b32", k2_2}, {"v_lshrrev_b32", k2_3}, {"v_xor_b32", k2_4}, {"v_lshl_or_b32", k2_5}, {"v_lshl_add_u32", k2_6}, {"v_add3_u32", k2_7}, {"v_mul_f32", k2_8}, {"v_add_f32", k2_9}, {"v_sub_f32", k2_10}, {"v_max_f32", k2_11}, {"v_floor_f32", k2_12}, {"v_fmac_f32", k2_13}, {"v_fma_f32_3src", k2_14}, {"v_mov_b32", k2_15}, {"v_cndmask_b32", k2_16}, {"v_bfe_u32", k2_17}, {"v_mul_u32_u24", k2_18}, {"v_mad_i32_i24", k2_19}, {"v_mul_lo_u32", k2_20}, {"v_dot2_u32_u16", k2_21}, {"v_perm_b32", k2_22}, {"v_mad_u64_u32", k2_23}, {"v_cvt_u32_f32", k2_24}, {"v_cvt_f32_i32", k2_25}, {"v_pk_add_f32", k2_26}, {"v_pk_mul_f32", k2_27}, {"v_min_i32", k2_28}};
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
