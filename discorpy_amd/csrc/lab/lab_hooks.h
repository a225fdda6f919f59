// lab/lab_hooks.h -- bodies of the hooks of dcp_lab.h (builds with -DDCP_LAB only: `make lab`).  Timing instruments, no product code.
#pragma once

// ---- per-wave phase timestamps of the frame kernels (tools/trace_k1.py): twelve 64-bit slots per wave, the first 65536 waves of a launch.
// slots 0..6: s_memtime at the phase boundaries marked DCP_TRACE(n) in the kernels; 7: XCC id << 32 | HW_ID; 8 / 9: s_memrealtime at
// the wave's start / end.  DCP_TRACE_WAVE_BEGIN declares `trace_id` / `trace_on` in the kernel's scope (lane must be in scope).
__device__ unsigned long long g_trace[65536 * 12];
#define DCP_TRACE(slot)                                                                                         \
  do {                                                                                                          \
    if (trace_on) {                                                                                             \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                               \
      if (lane == 0) g_trace[trace_id * 12 + (slot)] = t_;                                                      \
    }                                                                                                           \
  } while (0)
#define DCP_TRACE_WAVE_BEGIN(id)                                                                                \
  const unsigned trace_id = (id);                                                                               \
  const bool trace_on = trace_id < 65536u;                                                                      \
  DCP_TRACE(0);                                                                                                 \
  if (trace_on && lane == 0) {                                                                                  \
    unsigned hwid_, xcc_;                                                                                       \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid_));                                         \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                                         \
    g_trace[trace_id * 12 + 7] = ((unsigned long long)xcc_ << 32) | hwid_;                                      \
    g_trace[trace_id * 12 + 8] = __builtin_amdgcn_s_memrealtime();                                              \
  }
#define DCP_TRACE_WAVE_END()                                                                                    \
  do {                                                                                                          \
    if (trace_on && lane == 0) g_trace[trace_id * 12 + 9] = __builtin_amdgcn_s_memrealtime();                   \
  } while (0)
#define DCP_LAB_HOST_DEFINITIONS_UNWARP                                                                         \
  extern "C" __attribute__((visibility("default"))) int dcp_experiment_read_trace(unsigned long long* out, int nwaves) { \
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * 12 * (size_t)nwaves); \
  }

// ---- per-tile phase timestamps of spline_tile_filter_kernel (tools/trace_tf.py): thread 0's view and the recursion's (thread 512)
#define DCP_LAB_DEFINITIONS_SPLINE                                                                              \
  __device__ unsigned long long g_tf_trace[2][4096][8];                                                         \
  __device__ unsigned long long g_tf_trace_r[2][4096][8];                                                       \
  __device__ unsigned long long g_pf2d_trace[1024][8][8];                                                       \
  extern "C" __attribute__((visibility("default"))) int dcp_experiment_read_pf2d_trace(unsigned long long* out) { \
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pf2d_trace), sizeof(g_pf2d_trace));                       \
  }                                                                                                             \
  extern "C" __attribute__((visibility("default"))) int dcp_experiment_read_tf_trace(unsigned long long* out) { \
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tf_trace), sizeof(g_tf_trace));                           \
  }                                                                                                             \
  extern "C" __attribute__((visibility("default"))) int dcp_experiment_read_tf_trace_r(unsigned long long* out) { \
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tf_trace_r), sizeof(g_tf_trace_r));                       \
  }
// ---- per-step phase timestamps of spline_prefilter2d_kernel (tools/trace_pf2d.py): wave 0 of the first 1024 workgroups, eight steps
#define PF2D_TRACE(slot)                                                                                        \
  do {                                                                                                          \
    const int st_ = (r0 - Y0) / R;                                                                              \
    if (threadIdx.x == 0 && blockIdx.x < 1024 && st_ < 8) g_pf2d_trace[blockIdx.x][st_][slot] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define TF_TRACE(slot)                                                                                          \
  do {                                                                                                          \
    if (threadIdx.x == 0 && tile < 4096) g_tf_trace[AXIS][tile][slot] = __builtin_amdgcn_s_memtime();           \
  } while (0)
#define TF_TRACE_R(slot)                                                                                        \
  do {                                                                                                          \
    if (threadIdx.x == 512 && tile < 4096) g_tf_trace_r[AXIS][tile][slot] = __builtin_amdgcn_s_memtime();       \
  } while (0)
