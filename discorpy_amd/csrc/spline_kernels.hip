// spline_kernels.hip -- spline orders 2..5 of the unwarp path: what
// scipy.ndimage.map_coordinates does for order >= 2, which the reference reaches through the
// `order` argument of unwarp_image_backward / correct_perspective_image
// (discorpy/post/postprocessing.py:111,147,462,491; order=3 in examples/readthedocs_demo/demo_07.py:60).
//
//   spline_expand_kernel   image -> float64 plane, padded by 12 for 'nearest' / 'grid-constant' (an unpadded
//                          float32 image is read by the first causal pass directly instead)
//   spline_causal_kernel / spline_anticausal_kernel / spline_transpose_kernel
//                          recursive B-spline prefilter, chunked along the line (see below)
//   spline_remap_kernel    (order+1)^2-tap gather at the radial / perspective / explicit coordinates
//
// The arithmetic is operation for operation that of the spline section of oracle/unwarp_oracle.c
// (same expressions, same order, no contraction; pow(z, n) is evaluated on the host and passed in),
// so GPU and oracle agree bit for bit; the oracle is within one float32 ulp of scipy.
// Lines longer than 256 samples are filtered in overlapping chunks: there GPU and oracle agree to
// ~1e-24 relative in the coefficients rather than bit for bit.  The gather reads its taps
// straight from global memory.
#include "dcp_internal.h"
#include "dcp_device.h"
#include "dcp_lab.h"
#include <cstdio>
#include <type_traits>

#ifndef DCP_SPLINE_OUT_AUX
#define DCP_SPLINE_OUT_AUX 2   // cache-policy bits of spline_wg_kernel's result store: 2 = nt (0: plain, A/B)
#endif

namespace dcp {

constexpr int kSplBlock = 256;
int g_spline_wg = 1;             // 0: the (order + 1)^2 taps always gathered from global memory (option spline_wg)
void set_spline_wg(int v) { g_spline_wg = v; }
int get_spline_wg() { return g_spline_wg; }
int g_spline_tiled = 1;          // option spline_tiled (A/B runs and tests): 0 the chunked passes + transposes; 1 the fastest kernels (one pole, float32 source: both axes
                                 // in one launch, spline_prefilter2d_kernel; otherwise as 6); 2 the LDS tile kernel on both axes; 3 as 6 with the unstaged column
                                 // stream kernel; 4 tile kernel down the columns, cross-lane scan along the rows; 6 the two launches of rounds 3-5 (the LDS-staged
                                 // register column pass, then the register row pass behind an odd-pitch LDS staging)
void set_spline_tiled(int v) { g_spline_tiled = v; }
int g_pf2d_chunk = 0;            // option pf2d_chunk (A/B runs): rows per chunk of spline_prefilter2d_kernel, 0 = chosen from the plane and the chip
void set_pf2d_chunk(int v) { g_pf2d_chunk = v; }
int get_pf2d_chunk() { return g_pf2d_chunk; }
int g_pf2d_xcd = 1;              // option pf2d_xcd (A/B runs): 0 = tiles in plain row-major launch order (neighbouring stripes on different XCDs)
void set_pf2d_xcd(int v) { g_pf2d_xcd = v; }
int get_pf2d_xcd() { return g_pf2d_xcd; }
int g_pf2d_two_pole = 1;         // option pf2d_two_pole (A/B runs): 0 = orders 4 / 5 keep the tile filter, one launch per axis (rounds 2-5)
void set_pf2d_two_pole(int v) { g_pf2d_two_pole = v; }
int get_pf2d_two_pole() { return g_pf2d_two_pole; }
int g_spline_xcd = 1;            // option spline_xcd (A/B runs): 0 = spline_wg_kernel's tiles in plain row-major launch order
void set_spline_xcd(int v) { g_spline_xcd = v; }
int get_spline_xcd() { return g_spline_xcd; }
int get_spline_tiled() { return g_spline_tiled; }

__global__ void __launch_bounds__(kSplBlock) spline_expand_kernel(const SplineArgs a) {
  const int64_t i = (int64_t)blockIdx.x * kSplBlock + threadIdx.x;
  if (i >= (int64_t)a.Hp * a.Wp) return;
  const int y = (int)(i / a.Wp), x = (int)(i - (int64_t)y * a.Wp);
  int sy = y - a.pad, sx = x - a.pad;
  double v;
  if (a.mode == kModeGridConstant && (sy < 0 || sy >= a.H || sx < 0 || sx >= a.W)) {
    v = 0.0;
  } else {
    sy = sy < 0 ? 0 : (sy > a.H - 1 ? a.H - 1 : sy);
    sx = sx < 0 ? 0 : (sx > a.W - 1 ? a.W - 1 : sx);
    v = load_any(a.src, a.src_dtype, (size_t)sy * a.src_stride + (size_t)sx * a.src_cstride);
  }
  a.coef[i] = v;
}

// ---- recursive prefilter --------------------------------------------------------------------
// A line of n samples is cut into chunks of kChunk; thread (line, chunk) restarts the recursion
// kWarm samples before its chunk from a zero state.  The recursions forget like |z|^k with
// |z| <= 0.431, so after kWarm = 64 steps the restart error is below 4e-24 of the signal -- far
// under one float64 ulp -- while every chunk but the first/last runs independently: n/kChunk
// times more parallelism than one thread per line.  Lines of at most kChunk samples take a
// single chunk and are then exactly the oracle's serial recursion.  Passes are out of place
// (in -> out); lines run along axis 0 of a row-major (n x nlines) plane, i.e. lanes walk down
// columns and every access is coalesced; the row pass is a column pass on the transposed plane.
constexpr int kChunk = 256;
constexpr int kWarm = 64;
constexpr int kHorizon = 64;
#ifndef DCP_SPL_BATCH
#define DCP_SPL_BATCH 16
#endif
constexpr int kBatch = DCP_SPL_BATCH;   // loads issued ahead of the recursion     // terms kept of the exact initial sums (SPL_HORIZON in the oracle)

struct FilterPass {
  const double* in;
  double* out;
  int32_t n;             // samples per line
  int32_t nlines;
  int32_t kind;          // SplineFilterKind
  double z, zpow;        // pole; z^n (reflect) / z^(n-1) (mirror)
  double lam;            // gain applied to the input of the first causal pass, 1.0 afterwards
};

// FROM_SRC: the first causal pass (axis 0, first pole) reads the caller's float32 image directly instead of
// a float64 copy of it (no expansion pass).  Other element types and the two padded modes are expanded first.
template <bool FROM_SRC>
__global__ void __launch_bounds__(kSplBlock) spline_causal_kernel(const FilterPass f, const SplineArgs a) {
  const int line = blockIdx.x * kSplBlock + threadIdx.x;
  if (line >= f.nlines) return;
  const int64_t s = f.nlines;                         // stride between samples of a line
  const double* __restrict__ in = f.in + line;
  double* __restrict__ out = f.out + line;
  // sample i of this line; FROM_SRC is used for unpadded modes only, so image row i is sample i
  const float* __restrict__ col = FROM_SRC ? (const float*)a.src + (size_t)line * a.src_cstride : nullptr;
  const int64_t cstep = a.src_stride;
  auto ld = [&](int i) -> double {
    if constexpr (FROM_SRC) return (double)col[(int64_t)i * cstep];
    else return in[(int64_t)i * s];
  };
  const int n = f.n;
  const int c0 = blockIdx.y * kChunk;                 // first sample this thread writes
  const int c1 = min(n, c0 + kChunk);
  const double z = f.z, lam = f.lam;
  if (n < 2) {                                        // scipy leaves a 1-sample line untouched
    if (c0 == 0 && n == 1) out[0] = ld(0);
    return;
  }
  double t;
  int i;
  if (c0 - kWarm <= 0) {
    // exact start of the line (the expressions of spline_filter_line() in the oracle)
    const double x0 = ld(0) * lam;
    if (f.kind == kSplReflect) {
      double z_i = z;
      const double z_n = f.zpow;
      double acc = x0 + z_n * (ld(n - 1) * lam);
      const int m = min(n - 1, kHorizon);
      for (int k = 1; k <= m; ++k) {
        // (scipy accumulates this sum in c[0], so its last term reads the partial sum where the formula wants sample 0:
        // see spline_filter_line() in the oracle)
        const double far = n - 1 - k == 0 ? acc : ld(n - 1 - k) * lam;
        acc += z_i * (ld(k) * lam + z_n * far);
        z_i *= z;
      }
      t = acc * z / (1.0 - z_i * z_i) + x0;
    } else if (f.kind == kSplMirror) {
      double z_i = z;
      const double z_n_1 = f.zpow;
      double acc = x0 + z_n_1 * (ld(n - 1) * lam);
      const int m = min(n - 2, kHorizon);
      for (int k = 1; k <= m; ++k) {
        acc += z_i * (ld(k) * lam + z_n_1 * (ld(n - 1 - k) * lam));
        z_i *= z;
      }
      t = acc / (1.0 - z_n_1 * z_n_1);
    } else {
      double z_i = z, acc = x0;
      const int m = min(n - 1, kHorizon);
      for (int k = 0; k < m; ++k) {
        acc += z_i * (ld(n - 1 - k) * lam);
        z_i *= z;
      }
      t = acc / (1.0 - z_i);
    }
    if (c0 == 0) out[0] = t;
    i = 1;
  } else {
    t = 0.0;
    i = c0 - kWarm;
  }
  for (; i < c0; ++i) t = ld(i) * lam + z * t;          // warm-up, nothing stored
  // kBatch loads in flight per thread: one wave per SIMD is all the parallelism a 4096 x 4096 plane
  // offers (lines x chunks), so the latency has to be covered inside the thread
  for (; i + kBatch <= c1; i += kBatch) {
    double v[kBatch];
#pragma unroll
    for (int j = 0; j < kBatch; ++j) v[j] = ld(i + j);
#pragma unroll
    for (int j = 0; j < kBatch; ++j) {
      t = v[j] * lam + z * t;
      out[(int64_t)(i + j) * s] = t;
    }
  }
  for (; i < c1; ++i) {
    t = ld(i) * lam + z * t;
    out[(int64_t)i * s] = t;
  }
}

__global__ void __launch_bounds__(kSplBlock) spline_anticausal_kernel(const FilterPass f) {
  const int line = blockIdx.x * kSplBlock + threadIdx.x;
  if (line >= f.nlines) return;
  const int64_t s = f.nlines;
  const double* __restrict__ in = f.in + line;        // output of the causal pass
  double* __restrict__ out = f.out + line;
  const int n = f.n;
  const int c0 = blockIdx.y * kChunk;
  const int c1 = min(n, c0 + kChunk);                 // this thread writes samples c1-1 down to c0
  const double z = f.z;
  if (n < 2) {
    if (c0 == 0 && n == 1) out[0] = in[0];
    return;
  }
  double t;
  int i;
  if (c1 + kWarm >= n) {
    // exact end of the line
    if (f.kind == kSplReflect) {
      t = in[(int64_t)(n - 1) * s] * (z / (z - 1.0));
    } else if (f.kind == kSplMirror) {
      t = (z / (z * z - 1.0)) * (in[(int64_t)(n - 1) * s] + z * in[(int64_t)(n - 2) * s]);
    } else {
      double z_i = z, acc = in[(int64_t)(n - 1) * s];
      const int m = min(n - 1, kHorizon);
      for (int k = 0; k < m; ++k) {
        acc += z_i * in[(int64_t)k * s];
        z_i *= z;
      }
      t = acc * z / (z_i - 1.0);
    }
    if (c1 == n) out[(int64_t)(n - 1) * s] = t;
    i = n - 2;
  } else {
    t = 0.0;
    i = c1 + kWarm - 1;
  }
  for (; i >= c1; --i) t = z * (t - in[(int64_t)i * s]);              // warm-up
  for (; i - kBatch + 1 >= c0; i -= kBatch) {
    double v[kBatch];
#pragma unroll
    for (int j = 0; j < kBatch; ++j) v[j] = in[(int64_t)(i - j) * s];
#pragma unroll
    for (int j = 0; j < kBatch; ++j) {
      t = z * (t - v[j]);
      out[(int64_t)(i - j) * s] = t;
    }
  }
  for (; i >= c0; --i) {
    t = z * (t - in[(int64_t)i * s]);
    out[(int64_t)i * s] = t;
  }
}

// ---- the prefilter of one axis in ONE pass over the plane ---------------------------------------
// A workgroup stages kTfLines lines x up to kTfSamples samples (its core of `core` samples and `halo` samples on both
// sides) in LDS as float64, runs every pole's causal and anti-causal recursion there (lane = line, wave 0), and writes
// the core back: the plane is read once and written once per axis instead of four times plus two transposes.  The halo
// plays kWarm's part -- a recursion restarted from a zero state forgets like |z|^k; the host sizes it so that every pole
// has decayed below 2^-64 of the signal before the core -- and a tile that reaches the start (end) of the line uses the
// exact initial sums of the oracle.  AXIS 0: lines are columns (lane = column: every global and LDS access is
// contiguous across the wave); AXIS 1: lines are rows (loads and stores run along the row, the recursion walks LDS rows
// of pitch kTfSamples + 1).  Used for the reflect / mirror boundary kinds on lines long enough that z^n underflows to zero
// (the far end of the line then does not enter the initial sums); everything else takes the chunked passes above.
constexpr int kTfLines = 64;
constexpr int kTfSamples = 256;        // samples per tile incl. halos: 128 KB of LDS, one workgroup per CU (two-pole orders 4 and 5)
// (152-sample tiles -- 76 KB, two workgroups per CU, one moving data while the other's wave 0 recurses -- were slower:
// cubic 4096^2 0.396 ms against 0.351, the shorter core pays the halo more often than the overlap returns)

struct TileFilter {
  const void* in;
  double* out;
  int64_t in_ls, in_ss;      // input strides in elements: between lines, between samples of a line
  int64_t out_ls, out_ss;
  int32_t n, nlines, kind, npoles, halo, core;
  int32_t hp[2];             // per pole: samples after which a recursion restarted from zero has decayed below 2^-64 (halo = their sum)
  double z[2];
  double lam;
};

#ifndef DCP_TF_RW
#define DCP_TF_RW 16                 // waves that recurse (segments of 256 / DCP_TF_RW samples)
#endif
constexpr int kTfBlock = 1024;       // 16 waves move the tile (16 rows of loads in flight each); wave 0 runs the recursions
constexpr int kTfWaves = kTfBlock / 64;

DCP_LAB_DEFINITIONS_SPLINE

// Workgroup barrier that orders LDS traffic only (__syncthreads() also drains the wave's global memory operations; the tile
// lives in LDS and the prefetched values in registers the compiler tracks itself, so the barriers here need lgkmcnt only and
// the stores of one tile / the loads of the next stay in flight across them).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int AXIS, bool IN_F32, int SAMPLES>
__global__ void __launch_bounds__(kTfBlock) spline_tile_filter_kernel(const TileFilter f) {
  extern __shared__ double s_t[];
  constexpr int kTfPitch1 = SAMPLES + 1;
  // (the wave index is wave-uniform by construction; said explicitly, or every segment bound derived from it lives in VGPRs and
  // the recursion loops run under per-lane predicates)
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
  // The workgroup is persistent (one per CU: the tile takes most of the LDS) and walks tiles blockIdx.x, + gridDim.x, ...;
  // the 16 loads per lane of tile i + 1 are issued into registers before wave 0 starts the recursions of tile i and are
  // written to LDS when tile i has been stored, so the global reads run under the recursions.
  constexpr int NL = AXIS == 0 ? (SAMPLES + kTfWaves - 1) / kTfWaves : (kTfLines / kTfWaves) * ((SAMPLES + 63) / 64);
  constexpr int LW = kTfLines / kTfWaves;                     // AXIS 1: 4 lines per wave, segments of 64 samples each
  constexpr int NS = (SAMPLES + 63) / 64;
  const int tiles_l = (f.nlines + kTfLines - 1) / kTfLines, tiles_c = (f.n + f.core - 1) / f.core;
  const int ntiles = tiles_l * tiles_c;
  struct Geo {
    int l0, g0, gs, ge, R;
  };
  auto geo_of = [&](int t) -> Geo {
    Geo g;
    const int bc = t / tiles_l;
    g.l0 = (t - bc * tiles_l) * kTfLines;
    g.g0 = bc * f.core;                                       // first sample the tile writes
    g.gs = max(0, g.g0 - f.halo);
    g.ge = min(f.n, g.g0 + f.core + f.halo);
    g.R = g.ge - g.gs;
    return g;
  };
  // (raw loaded values, every address valid: a conversion or a select here would make the compiler wait for the data
  // where the loads are issued instead of where they are used)
  using RawT = typename std::conditional<IN_F32, float, double>::type;
  RawT pre[NL];
  const RawT* const src_raw = (const RawT*)f.in;
  auto issue_loads = [&](const Geo& g) {
    if constexpr (AXIS == 0) {
      const int line = min(g.l0 + lane, f.nlines - 1);        // lanes past the last line repeat it (never stored)
      const int64_t base = (int64_t)line * f.in_ls + (int64_t)g.gs * f.in_ss;
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const int r = min(wave + kTfWaves * j, g.R - 1);
        pre[j] = src_raw[base + (int64_t)r * f.in_ss];
      }
    } else {
#pragma unroll
      for (int q = 0; q < LW; ++q) {
        const int line = min(g.l0 + wave * LW + q, f.nlines - 1);
        const int64_t base = (int64_t)line * f.in_ls + (int64_t)g.gs * f.in_ss;
#pragma unroll
        for (int j = 0; j < NS; ++j) pre[q * NS + j] = src_raw[base + (int64_t)min(lane + 64 * j, g.R - 1) * f.in_ss];
      }
    }
  };
  auto commit = [&](const Geo& g) {
    if constexpr (AXIS == 0) {
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const int r = wave + kTfWaves * j;
        if (r < g.R) s_t[r * kTfLines + lane] = (double)pre[j];
      }
    } else {
#pragma unroll
      for (int q = 0; q < LW; ++q)
#pragma unroll
        for (int j = 0; j < NS; ++j)
          if (lane + 64 * j < g.R) s_t[(wave * LW + q) * kTfPitch1 + lane + 64 * j] = (double)pre[q * NS + j];
    }
  };
  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  Geo cur = geo_of(tile);
  issue_loads(cur);
  for (;;) {
  TF_TRACE(0);
  commit(cur);
  lds_barrier();
  TF_TRACE(1);
  const int l0 = cur.l0, g0 = cur.g0, gs = cur.gs, ge = cur.ge, R = cur.R;
  const int next = tile + (int)gridDim.x;
  Geo nxt = cur;
  if (next < ntiles) {
    nxt = geo_of(next);
    issue_loads(nxt);
  }
  TF_TRACE(2);
  // ---- the recursions, lane = line.  EVERY wave owns a segment of SEG consecutive samples of the tile's 64 lines: it first
  // runs the recursion, without storing, over the `hp` samples in front of its segment from a zero state (|z|^hp <= 2^-64: the
  // restart the tile's own halo already relies on -- or from the exact initial sum when that range reaches the start of the
  // line), then, after a barrier (nobody has overwritten an input yet), through its segment in place.  16 waves x (hp + 16)
  // dependent steps instead of one wave x 256: the recursion of a tile is ~5 times shorter and no wave idles through it.
  {
    double* a = AXIS == 0 ? s_t + lane : s_t + lane * kTfPitch1;
    constexpr int S = AXIS == 0 ? kTfLines : 1;               // LDS distance between consecutive samples of a line
    constexpr int SEG = (SAMPLES + DCP_TF_RW - 1) / DCP_TF_RW;
    const int s0 = wave * SEG, e0 = min(R, s0 + SEG);         // this wave's segment [s0, e0); empty when s0 >= R
    const bool mine = wave < DCP_TF_RW && s0 < R;
    for (int p = 0; p < f.npoles; ++p) {
      const double z = f.z[p], lam = p == 0 ? f.lam : 1.0;
      const int hp = f.hp[p];
      TF_TRACE_R(0);
      // -- causal: the state in front of the segment
      double t = 0.0;
      bool first_exact = false;                               // t is the exact value of sample 0 (wave-uniform)
      if (mine) {
        int i = max(0, s0 - hp);
        if (i == 0 && gs == 0) {
          // exact start of the line, z^n == 0 (the expressions of spline_filter_line() in the oracle with that factor dropped)
          const double x0 = a[0] * lam;
          double z_i = z, acc = x0;
          if (f.kind == kSplReflect) {
            const int m = min(f.n - 1, kHorizon);
            for (int k = 1; k <= m; ++k) {
              acc += z_i * (a[k * S] * lam);
              z_i *= z;
            }
            t = acc * z / (1.0 - z_i * z_i) + x0;
          } else {
            const int m = min(f.n - 2, kHorizon);
            for (int k = 1; k <= m; ++k) {
              acc += z_i * (a[k * S] * lam);
              z_i *= z;
            }
            t = acc;
          }
          first_exact = true;
          i = 1;
        }
        for (; i + 8 <= s0; i += 8) {
          double v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = a[(i + j) * S];
#pragma unroll
          for (int j = 0; j < 8; ++j) t = v[j] * lam + z * t;
        }
        for (; i < s0; ++i) t = a[i * S] * lam + z * t;
      }
      TF_TRACE(6);
      TF_TRACE_R(1);
      lds_barrier();
      TF_TRACE(7);
      TF_TRACE_R(2);
      double cs[SEG];                                         // the segment's causal values stay in registers for its anti-causal pass
      if (mine) {
#pragma unroll
        for (int j = 0; j < SEG; ++j) cs[j] = s0 + j < e0 ? a[(s0 + j) * S] : 0.0;
#pragma unroll
        for (int j = 0; j < SEG; ++j) {
          if (s0 + j < e0) {
            if (!(first_exact && s0 + j == 0)) t = cs[j] * lam + z * t;
            cs[j] = t;
            a[(s0 + j) * S] = t;                              // (the neighbours' anti-causal warm-up reads it)
          }
        }
      }
      TF_TRACE_R(3);
      lds_barrier();
      TF_TRACE_R(4);
      // -- anti-causal: the state behind the segment, from the causal values of the following samples
      t = 0.0;
      bool last_exact = false;                                // t is the exact value of sample R - 1
      if (mine) {
        int i = min(R, e0 + hp) - 1;
        if (i == R - 1 && ge == f.n) {
          if (f.kind == kSplReflect) t = a[(R - 1) * S] * (z / (z - 1.0));
          else t = (z / (z * z - 1.0)) * (a[(R - 1) * S] + z * a[(R - 2) * S]);
          last_exact = true;
          i = R - 2;
        }
        for (; i - 7 >= e0; i -= 8) {
          double v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = a[(i - j) * S];
#pragma unroll
          for (int j = 0; j < 8; ++j) t = z * (t - v[j]);
        }
        for (; i >= e0; --i) t = z * (t - a[i * S]);
      }
      TF_TRACE_R(5);
      lds_barrier();
      TF_TRACE_R(6);
      if (mine) {
#pragma unroll
        for (int j = SEG - 1; j >= 0; --j) {
          if (s0 + j < e0) {
            if (!(last_exact && s0 + j == R - 1)) t = z * (t - cs[j]);
            a[(s0 + j) * S] = t;
          }
        }
      }
      TF_TRACE_R(7);
      lds_barrier();
    }
  }
  lds_barrier();
  TF_TRACE(3);
  // ---- LDS -> global: the core
  const int c_lo = g0 - gs, c_n = min(f.core, f.n - g0);
  if constexpr (AXIS == 0) {
    if (l0 + lane < f.nlines) {
      double* o = f.out + (int64_t)(l0 + lane) * f.out_ls + (int64_t)g0 * f.out_ss;
      for (int r = wave; r < c_n; r += kTfWaves) o[(int64_t)r * f.out_ss] = s_t[(c_lo + r) * kTfLines + lane];
    }
  } else {
    for (int li = wave; li < kTfLines; li += kTfWaves) {
      if (l0 + li >= f.nlines) break;
      double* o = f.out + (int64_t)(l0 + li) * f.out_ls + (int64_t)g0 * f.out_ss;
      for (int sI = lane; sI < c_n; sI += 64) o[(int64_t)sI * f.out_ss] = s_t[li * kTfPitch1 + c_lo + sI];
    }
  }
  TF_TRACE(4);
  if (next >= ntiles) break;
  lds_barrier();                                              // the tile has been read out: LDS is free for the next one
  TF_TRACE(5);
  tile = next;
  cur = nxt;
  }
}

// ---- the column pass (axis 0) of a float32 image WITHOUT LDS and without barriers --------------------------------------
// spline_tile_filter_kernel moves a 128 KB tile through LDS in synchronised phases (load, causal warm-up, barrier, causal,
// barrier, anti-causal warm-up, barrier, ...): one workgroup per CU, the recursion stages bound by LDS writes.  Down the
// columns nothing has to be exchanged between lanes: lane = column, so a row of the plane is one contiguous 256-byte load
// (float32 in) / 512-byte store (float64 out) per wave.  Here a wave owns 64 columns x kCsSeg rows and every lane keeps its
// column's values in REGISTERS: the causal recursion runs over the HP rows in front of the segment from a zero state (or from
// the exact initial sum at the top of the column), through the segment and HP rows past it; the anti-causal recursion comes
// back over those HP rows (from zero, or from the exact end value) and through the segment, storing as it goes.  The HP rows
// on both sides are re-read by the neighbouring segments -- 3.1 loads per sample, of which the L2 / Infinity Cache serve 2.1
// (the float32 source of a 4096^2 frame is 67 MB) -- in exchange for no LDS traffic, no barrier and 8 independent waves per
// CU instead of one lock-step workgroup.  Single-pole orders (2, 3), reflect / mirror kinds on columns long enough that
// z^n underflows (the conditions of the tile kernel); same arithmetic per sample as the tile kernel and the oracle
// (t = x lam + z t; t = z (t - c)), the same 2^-64 restarts.
#ifndef DCP_CS_SEG
#define DCP_CS_SEG 32
#endif
constexpr int kCsSeg = DCP_CS_SEG;
#ifndef DCP_CS_K
#define DCP_CS_K 1                   // consecutive segments a wave walks: the causal state carries over, only the anti-causal pass restarts
#endif
constexpr int kCsK = DCP_CS_K;
// A wave walks kCsK consecutive segments of kCsSeg rows down its 64 columns: the causal recursion runs on without a restart
// (its state is exact), only the anti-causal pass of every segment restarts HP rows further down -- from values the causal
// pass has already produced and that the next segment reuses.  Rows read per output row: (K SEG + 2 HP) / (K SEG) = 1.5 for
// K = 4 (3.1 for K = 1): the L1-miss stream, which bounds this kernel, is 0.77 of what one segment per wave moves.
#ifndef DCP_CS_WAVES
#define DCP_CS_WAVES 2
#endif
template <int HP>
__global__ void __launch_bounds__(256, DCP_CS_WAVES) spline_col_stream_kernel(const TileFilter f) {
  constexpr int SEG = kCsSeg, K = kCsK, NC = K * SEG + HP;   // causal values of the wave's rows and of the HP rows behind them
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
  const int col = blockIdx.x * 64 + lane;
  const int n = f.n;
  const int R0 = __builtin_amdgcn_readfirstlane(((int)blockIdx.y * 4 + wave) * (K * SEG));
  if (R0 >= n) return;
  const float* __restrict__ src = (const float*)f.in + (int64_t)min(col, f.nlines - 1) * f.in_ls;
  const int64_t ss = f.in_ss;
  const double z = f.z[0], lam = f.lam;
  double* __restrict__ out = f.out + (int64_t)col * f.out_ls;
  const bool store_lane = col < f.nlines;
  // the first SEG + HP rows, all in flight while the warm-up runs (rows past the end repeat the last one and are not used)
  float pre[SEG + HP];
#pragma unroll
  for (int j = 0; j < SEG + HP; ++j) pre[j] = src[(int64_t)min(R0 + j, n - 1) * ss];
  // ---- causal: the state in front of the wave's rows
  double t = 0.0;
  int i = max(0, R0 - HP);
  bool first_exact = false;                                  // t is the exact value of sample 0 (wave-uniform)
  if (i == 0) {
    // exact start of the line, z^n == 0 (spline_filter_line() of the oracle with that factor dropped; as the tile kernel)
    const double x0 = (double)src[0] * lam;
    double z_i = z, acc = x0;
    if (f.kind == kSplReflect) {
      const int m = min(n - 1, kHorizon);
      for (int k = 1; k <= m; ++k) {
        acc += z_i * ((double)src[(int64_t)k * ss] * lam);
        z_i *= z;
      }
      t = acc * z / (1.0 - z_i * z_i) + x0;
    } else {
      const int m = min(n - 2, kHorizon);
      for (int k = 1; k <= m; ++k) {
        acc += z_i * ((double)src[(int64_t)k * ss] * lam);
        z_i *= z;
      }
      t = acc;
    }
    first_exact = true;
    i = 1;
  }
  for (; i + 8 <= R0; i += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[(int64_t)(i + j) * ss];
#pragma unroll
    for (int j = 0; j < 8; ++j) t = (double)v[j] * lam + z * t;
  }
  for (; i < R0; ++i) t = (double)src[(int64_t)i * ss] * lam + z * t;
  const double c_before = t;                                 // causal value of row R0 - 1 (the mirror end formula may need it)
  double C[NC];                                              // (indices are compile-time constants: registers, live ~SEG + HP at a time)
  double tc = t;                                             // the causal state
#pragma unroll
  for (int j = 0; j < SEG + HP; ++j) {
    C[j] = 0.0;
    if (R0 + j < n) {
      if (!(first_exact && R0 + j == 0)) tc = (double)pre[j] * lam + z * tc;
      C[j] = tc;
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int r0 = R0 + k * SEG;
    if (r0 < n) {                                            // (wave-uniform)
      const int r1 = min(n, r0 + SEG), a1 = min(n, r1 + HP);
      // the next segment's new rows, in flight under this segment's anti-causal pass
      float nx[SEG];
      if (k + 1 < K) {
#pragma unroll
        for (int j = 0; j < SEG; ++j) nx[j] = src[(int64_t)min(r0 + SEG + HP + j, n - 1) * ss];
      }
      // ---- anti-causal: back over the HP rows behind the segment (from zero, or from the exact end value), then through the
      // segment, storing
      double ta = 0.0;
#pragma unroll
      for (int j = (k + 1) * SEG + HP - 1; j >= k * SEG; --j) {
        const int row = R0 + j;
        if (row < a1) {
          if (a1 == n && row == n - 1) {
            if (f.kind == kSplReflect) ta = C[j] * (z / (z - 1.0));
            else ta = (z / (z * z - 1.0)) * (C[j] + z * (j > 0 ? C[j > 0 ? j - 1 : 0] : c_before));
          } else {
            ta = z * (ta - C[j]);
          }
          if (row < r1 && store_lane) out[(int64_t)row * f.out_ss] = ta;
        }
      }
      if (k + 1 < K) {
#pragma unroll
        for (int j = 0; j < SEG; ++j) {
          const int jj = (k + 1) * SEG + HP + j;
          C[jj < NC ? jj : 0] = 0.0;
          if (R0 + jj < n) {
            tc = (double)nx[j] * lam + z * tc;
            C[jj < NC ? jj : 0] = tc;
          }
        }
      }
    }
  }
}

// ---- the same column pass with the float32 rows of a workgroup staged in LDS ONCE -----------------------------------------
// spline_col_stream_kernel's four waves read overlapping rows (each wave its segment and HP rows on both sides: 3.1 loads per
// sample), and what bounds it is the rate at which a CU takes streamed data in (no-store ablation: 38 us for 67 MB of source).
// Here the workgroup copies the rows its four segments need -- 4 SEG + 2 HP rows of 64 float32 columns, <= 49 KB -- into LDS
// with 16-byte LDS-DMA loads (one row = 256 contiguous bytes = a quarter of a load), one barrier, and every lane then reads
// its column out of LDS (lane = column: conflict-free): 1.5 loads per sample from the L2, the rest from LDS.  The recursions
// run in registers exactly as in the stream kernel; the float64 results go straight to global memory, 512 bytes per wave and
// row (plain stores: the row pass that follows finds them in the Infinity Cache).
template <int HP>
__global__ void __launch_bounds__(256, 3) spline_col_lds_kernel(const TileFilter f, const uint32_t src_bytes) {
  constexpr int SEG = 32, J = SEG + HP, ROWS = 4 * SEG + 2 * HP;
  __shared__ __attribute__((aligned(16))) float s_in[ROWS * 64];
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
  const int col0 = blockIdx.x * 64, col = col0 + lane;
  const int n = f.n;
  const int R0 = (int)blockIdx.y * (4 * SEG);                 // first row the workgroup writes
  const int base = max(0, R0 - HP), top = min(n, R0 + 4 * SEG + HP);   // rows [base, top) are staged
  // ---- fill: row (base + 4 q + lane / 16), 16-byte chunk (lane % 16), q = wave, wave + 4, ...
  {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)f.in, 0, (int)src_bytes, 0x00020000);
    const uint32_t rstep = (uint32_t)f.in_ss * 4u;
    const uint32_t off0 = (uint32_t)(base + (lane >> 4)) * rstep + (uint32_t)col0 * 4u + (uint32_t)(lane & 15) * 16u;
    const int ngroups = (top - base + 3) >> 2;
#pragma unroll
    for (int q = 0; q < (ROWS + 15) / 16; ++q) {
      const int g = q * 4 + wave;
      // (rows past `top` are inside the image or past the descriptor's extent -- zeros --, and the slab has room for whole groups)
      if (g < ngroups) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(s_in + g * 256), 16, off0 + (uint32_t)g * 4u * rstep, 0, 0, 0);
    }
  }
  const int r0 = __builtin_amdgcn_readfirstlane(R0 + wave * SEG);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (r0 >= n) return;
  const int r1 = min(n, r0 + SEG), a1 = min(n, r1 + HP);
  const float* a = s_in + lane - base * 64;                   // a[row * 64]: sample `row` of this lane's column
  const double z = f.z[0], lam = f.lam;
  // ---- causal: the state in front of the segment
  double t = 0.0;
  int i = max(0, r0 - HP);
  bool first_exact = false;                                  // t is the exact value of sample 0 (wave-uniform)
  if (i == 0) {
    // exact start of the line, z^n == 0 (spline_filter_line() of the oracle with that factor dropped; rows 0 .. 64 are staged:
    // this is the workgroup at the top of the column)
    const double x0 = (double)a[0] * lam;
    double z_i = z, acc = x0;
    if (f.kind == kSplReflect) {
      const int m = min(n - 1, kHorizon);
      for (int k = 1; k <= m; ++k) {
        acc += z_i * ((double)a[k * 64] * lam);
        z_i *= z;
      }
      t = acc * z / (1.0 - z_i * z_i) + x0;
    } else {
      const int m = min(n - 2, kHorizon);
      for (int k = 1; k <= m; ++k) {
        acc += z_i * ((double)a[k * 64] * lam);
        z_i *= z;
      }
      t = acc;
    }
    first_exact = true;
    i = 1;
  }
  for (; i + 8 <= r0; i += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = a[(i + j) * 64];
#pragma unroll
    for (int j = 0; j < 8; ++j) t = (double)v[j] * lam + z * t;
  }
  for (; i < r0; ++i) t = (double)a[i * 64] * lam + z * t;
  const double c_before = t;                                 // causal value of row r0 - 1 (the mirror end formula may need it)
  // ---- causal through the segment and the HP rows behind it, kept in registers
  double cs[J];
  const float* as = a + r0 * 64;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    cs[j] = 0.0;
    if (r0 + j < a1) {
      if (!(first_exact && r0 + j == 0)) t = (double)as[j * 64] * lam + z * t;
      cs[j] = t;
    }
  }
  // ---- anti-causal: back over the rows behind the segment (from zero, or from the exact end value), then through the segment
  double* __restrict__ out = f.out + (int64_t)col * f.out_ls;
  const bool store_lane = col < f.nlines;
  t = 0.0;
#pragma unroll
  for (int j = J - 1; j >= 0; --j) {
    const int row = r0 + j;
    if (row < a1) {
      if (a1 == n && row == n - 1) {
        if (f.kind == kSplReflect) t = cs[j] * (z / (z - 1.0));
        else t = (z / (z * z - 1.0)) * (cs[j] + z * (j > 0 ? cs[j > 0 ? j - 1 : 0] : c_before));
      } else {
        t = z * (t - cs[j]);
      }
      if (row < r1 && store_lane) out[(int64_t)row * f.out_ss] = t;
    }
  }
}

// ---- the row pass (axis 1) with the column pass's recipe: stage once, recurse in registers ---------------------------------
// Along a row the recursion of a line runs through CONSECUTIVE addresses, so lane = line needs the tile transposed: a workgroup
// stages 16 rows x 384 float64 samples in LDS at an ODD pitch (385 doubles: the 16 lanes of an LDS pass then hit 32 distinct
// banks whatever sample they read).  16-byte LDS-DMA cannot write an odd pitch; 4-byte LDS-DMA can: one load = 256 contiguous
// bytes of a row = 32 samples, twelve per row, 48 per wave.  After ONE barrier thread (row, segment) -- 16 rows x 16 segments of SEG
// outputs -- runs its causal recursion from HP samples in front of its segment through SEG + HP samples in registers and the
// anti-causal one back (the restarts and exact end formulas of the other kernels), a second barrier (everybody has read its
// inputs), the SEG results go into the tile in place, a third barrier, and the core leaves as contiguous 512-byte stores.
// Three barriers per tile instead of 2 + 4 per pole, a 49 KB tile (three workgroups per CU, not one), no LDS writes inside the
// recursion.  One pole (orders 2, 3), reflect / mirror kinds, z^n underflowed -- the tile kernel's conditions.
#ifndef DCP_RL_ROWS
#define DCP_RL_ROWS 16               // rows of a tile: 16 x 384 samples (32 x 192 moves 1.6 samples per output instead of 1.26)
#endif
constexpr int kRlRows = DCP_RL_ROWS, kRlWidth = 6144 / kRlRows, kRlPitch = kRlWidth + 1, kRlSegs = 256 / kRlRows;
static_assert(kRlRows == 16 || kRlRows == 32, "an LDS pass of 16 lanes must stay inside one segment");
template <int HP>
struct RowLds {
  static constexpr int SEG = (kRlWidth - 2 * HP) / kRlSegs;    // outputs per thread
  static constexpr int CORE = kRlSegs * SEG;                   // outputs per tile and row
};

template <int HP>
__global__ void __launch_bounds__(256, 3) spline_row_lds_kernel(const TileFilter f, const uint32_t in_bytes) {
  using G = RowLds<HP>;
  constexpr int SEG = G::SEG, CORE = G::CORE, J = SEG + HP;
  __shared__ __attribute__((aligned(16))) double s_t[kRlRows * kRlPitch];
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
  const int n = f.n;
  const int l0 = (int)blockIdx.x * kRlRows;                    // first line (row of the plane) of the tile
  const int g0 = (int)blockIdx.y * CORE;                       // first sample the tile writes
  const int base = max(0, g0 - HP);                            // first sample staged
  // ---- fill: rows wave, wave + 4, ...; six 256-byte chunks each (samples past the row's end belong to the next row or lie
  // past the descriptor's extent -- zeros --, and are never used)
  {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)f.in, 0, (int)in_bytes, 0x00020000);
#pragma unroll
    for (int q = 0; q < kRlRows / 4; ++q) {
      const int rr = q * 4 + wave;
      const uint32_t row_off = (uint32_t)min(l0 + rr, f.nlines - 1) * (uint32_t)f.in_ls * 8u + (uint32_t)base * 8u + (uint32_t)lane * 4u;
#pragma unroll
      for (int c = 0; c < kRlWidth / 32; ++c)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(s_t + rr * kRlPitch + c * 32), 4, row_off + (uint32_t)c * 256u, 0, 0, 0);
    }
  }
  const int row_l = (int)threadIdx.x % kRlRows, seg = (int)threadIdx.x / kRlRows;   // 16 consecutive lanes = 16 rows of one segment: conflict-free LDS passes
  const int s0 = g0 + seg * SEG;                               // this thread's outputs [s0, s1)
  const double* a = s_t + row_l * kRlPitch - base;             // a[i]: sample i of this thread's row
  const double z = f.z[0], lam = f.lam;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  double cs[J];
  const bool interior = g0 - HP >= 0 && g0 + CORE + HP <= n;   // (workgroup-uniform) no line end in reach: every thread does the same
  if (interior) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < HP; ++i) t = a[s0 - HP + i] * lam + z * t;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      t = a[s0 + j] * lam + z * t;
      cs[j] = t;
    }
    t = 0.0;
#pragma unroll
    for (int j = J - 1; j >= 0; --j) {
      t = z * (t - cs[j]);
      cs[j] = t;
    }
  } else {
    // a tile at the start or the end of the lines (2 of ~35 tile columns): the same recursions with every bound per thread
    const int s1 = min(min(g0 + CORE, n), s0 + SEG), a1 = min(n, s1 + HP);
    double t = 0.0;
    int i = max(0, s0 - HP);
    bool first_exact = false;
    if (i == 0 && s0 < n) {
      const double x0 = a[0] * lam;
      double z_i = z, acc = x0;
      const int m = min(f.kind == kSplReflect ? n - 1 : n - 2, kHorizon);
      for (int k = 1; k <= m; ++k) {
        acc += z_i * (a[k] * lam);
        z_i *= z;
      }
      t = f.kind == kSplReflect ? acc * z / (1.0 - z_i * z_i) + x0 : acc;
      first_exact = true;
      i = 1;
    }
    for (; i < s0; ++i) t = a[i] * lam + z * t;
    const double c_before = t;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      cs[j] = 0.0;
      if (s0 + j < a1) {
        if (!(first_exact && s0 + j == 0)) t = a[s0 + j] * lam + z * t;
        cs[j] = t;
      }
    }
    t = 0.0;
#pragma unroll
    for (int j = J - 1; j >= 0; --j) {
      const int idx = s0 + j;
      if (idx < a1) {
        if (a1 == n && idx == n - 1) {
          if (f.kind == kSplReflect) t = cs[j] * (z / (z - 1.0));
          else t = (z / (z * z - 1.0)) * (cs[j] + z * (j > 0 ? cs[j > 0 ? j - 1 : 0] : c_before));
        } else {
          t = z * (t - cs[j]);
        }
        cs[j] = t;
      }
    }
  }
  __syncthreads();                                             // every thread has read its inputs: the results may overwrite them
  {
    double* w = s_t + row_l * kRlPitch - base + s0;
#pragma unroll
    for (int j = 0; j < SEG; ++j)
      if (s0 + j < n) w[j] = cs[j];
  }
  __syncthreads();
  // ---- the core, 512 contiguous bytes per wave and store
  const int c_n = min(CORE, n - g0);
#pragma unroll
  for (int q = 0; q < kRlRows / 4; ++q) {
    const int rr = q * 4 + wave;
    if (l0 + rr >= f.nlines) break;
    double* o = f.out + (int64_t)(l0 + rr) * f.out_ls + g0;
    const double* r = s_t + rr * kRlPitch + (g0 - base);
    for (int idx = lane; idx < c_n; idx += 64) o[idx] = r[idx];
  }
}

// ---- BOTH axes of the one-pole prefilter in ONE pass over the plane: float32 image -> final coefficient plane -----------------
// The column pass writes a float64 plane (8 B per sample) that the row pass reads back and writes again: 12 + 16 of the spline
// path's 40 bytes per pixel.  Here a workgroup owns a STRIPE of 252 columns -- CORE columns it writes and HP on both sides -- and
// walks down it in steps of R = 32 rows: thread = column runs the column recursions in registers as spline_col_stream_kernel
// does (the causal state carries from step to step, only the anti-causal pass restarts HP rows below the step from values the
// next step reuses), but its R results go into an LDS tile of 32 x 252 float64 at an odd pitch instead of to memory.  After one
// barrier thread (row, segment) -- 32 rows x 8 segments of SEG columns -- runs the row recursions over SEG + 2 HP samples of its
// row as spline_row_lds_kernel does, the results go back into the tile in place and leave as contiguous row pieces.  The
// column-filtered plane never exists in memory: 4 B read (each source row once per stripe: 252 / CORE = 1.37 loads per sample, the
// neighbour's share out of the L2) + 8 B written per sample.  The rows of the next step are loaded into registers in front of
// the row pass and arrive under it.  Vertically the stripe is cut into chunks of `chunk_rows` rows (a multiple of R) so that the
// launch fills the chip; a chunk pays HP rows of causal warm-up above and HP rows of look-ahead below.
// No line end is special here: rows and columns outside the plane are READ AT THEIR MIRROR IMAGES (half-sample symmetric for the
// reflect kind, whole-sample for mirror), which is the extension scipy's exact initial values stand for once z^n has underflowed
// (the condition of every one-pass kernel) -- a recursion started from zero HP samples outside the plane reaches the edge with
// the exact value up to |z|^HP <= 2^-64 of the signal, the error every restart inside the plane has too.  Same arithmetic per
// sample as the two kernels it replaces (t = x lam + z t; t = z (t - c)); single coefficients may differ in the last bit as they
// do between any two of the prefilter paths.
#ifndef DCP_PF2D_WAVES
#define DCP_PF2D_WAVES 2
#endif
template <int HP, int RR = 32>
struct Pf2d {
  static constexpr int R = RR;                           // rows per step (32; 16 for the long horizons of the two-pole orders' first pole)
  static constexpr int NCOL = 252;                       // columns of a stripe incl. both halos (threads 252..255 idle in the column pass)
  static constexpr int PITCH = 253;                      // odd: the rows of one column fall into distinct bank pairs; 32 x 253 x 8 B = 63.25 KB
  static constexpr int SEGS = 256 / R;                   // row-pass segments per row
#ifndef DCP_PF2D_XCH_MIN
#define DCP_PF2D_XCH_MIN 40
#endif
  static constexpr bool XCH = HP >= DCP_PF2D_XCH_MIN;     // (see below)
  static constexpr int SEGC = XCH ? (NCOL - HP) / SEGS : 0;                  // XCH: causal values per thread, 26 (HP = 44) / 25 (HP = 48)
  static constexpr int SEG0 = (NCOL - 2 * HP) / SEGS;
  // outputs per row-pass thread: 23 (HP = 34) / 25 (HP = 26); XCH: the anti-causal warm-ups must find causal values, CORE + HP <= SEGS SEGC
  static constexpr int SEG = XCH && (SEGS * SEGC - HP) / SEGS < SEG0 ? (SEGS * SEGC - HP) / SEGS : SEG0;      // XCH: 20 (HP = 44) / 19 (HP = 48)
  static constexpr int CORE = SEGS * SEG;                // columns a stripe writes: 184 / 200 (XCH: 160 / 152)
  // Long horizons (the first pole of orders 4 / 5): a row-pass thread that keeps the causal values of its segment AND of the HP
  // samples behind it in registers (J = SEG + HP doubles, 3 HP + 2 SEG dependent steps) leaves room for 16-row steps only.  XCH: the
  // threads EXCHANGE the causal values through the tile instead -- causal pass over SEGC columns each (core + right halo, warm-up HP),
  // barrier, write, barrier, anti-causal pass over SEG outputs each with its warm-up read from the neighbours' causal values:
  // 2 HP + SEGC + SEG steps, SEGC live doubles, two barriers more per step -- and 32-row steps again.
  static_assert(!XCH || (HP + SEGS * SEGC <= NCOL && CORE + HP <= SEGS * SEGC), "the causal segments end inside the stripe and cover the anti-causal warm-ups");
};
// Orders 4 and 5 (two poles) as TWO passes of this kernel, one pole each (round 6): scipy filters every line with pole 1 and then
// with pole 2, axis by axis -- P2y P1y P2x P1x; the four operators commute (the two axes act on different indices, and the two
// poles of one axis are functions of the same mirrored shift), so P2y P2x (P1y P1x image) is the same plane up to the rounding of
// float64 sums.  Pass 1 reads the image (pole 1: |z| = 0.36 / 0.43 -- restart horizons of 44 / 48 samples, the row pass exchanging its
// causal values through the tile: XCH above) and writes a float64 plane, pass 2 reads that plane (ST = kF64; pole 2:
// |z| = 0.014 / 0.043, horizon 16) and writes the coefficient plane.  Each pass carries its own gain (1 - z)(1 - 1/z).
constexpr int kF64Src = 100;                             // ST of the second pass (not an element type of the ABI)

// PAD: the plane is the image with `pad` samples added on every side ('nearest': the edge sample repeated, 'grid-constant':
// zeros -- what spline_expand_kernel would write into a float64 copy first); the recursions run over the padded plane, the
// loads clamp into the image or return zero.
struct Pf2dPad {
  int32_t pad, Hs, Ws, zero_outside;      // samples added per side; rows / columns of the source image; 1: zeros outside it
};

// ST: element type of the source image (kF32, or the integer types cameras deliver -- kU8 / kU16 / kI16: every one exact in float64,
// read as raw bits and converted where a row is used)
template <int HP, bool PAD, int ST, int RR>
__device__ __forceinline__ void pf2d_body(const TileFilter& f, const Pf2dPad& pd, const uint32_t src_bytes, const uint32_t out_bytes, const int stripe, const int Y0,
                                          const int Yend, double* s_t) {
  using G = Pf2d<HP, RR>;
  constexpr int R = G::R, NCOL = G::NCOL, PITCH = G::PITCH, SEG = G::SEG, CORE = G::CORE, NC = R + HP, J = SEG + HP;
  const int tid = (int)threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int H = f.n, W = f.nlines;                             // rows, columns of the plane
  const int g0 = stripe * CORE;                                // first column the stripe writes
  const double z = f.z[0], lam = f.lam;
  const int sym = f.kind == kSplReflect ? 1 : 0;               // index i < 0 reads -i - sym, i >= n reads 2 n - 2 + sym - i
  // ---- column pass: thread = column g0 - HP + tid of the (mirrored) plane
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)f.in, 0, (int)src_bytes, 0x00020000);
  int gc = g0 - HP + min(tid, NCOL - 1);
  gc = gc < 0 ? -gc - sym : gc;
  gc = gc >= W ? 2 * W - 2 + sym - gc : gc;
  gc = max(0, min(gc, W - 1));                                 // (a stripe that ends far past the plane: those columns are never used)
  bool col_zero = false;                                       // PAD, zeros outside: this thread's column lies in the padding
  if constexpr (PAD) {
    gc -= pd.pad;
    col_zero = pd.zero_outside && (gc < 0 || gc >= pd.Ws);
    gc = max(0, min(gc, pd.Ws - 1));
  }
  constexpr uint32_t ES = ST == kF64Src ? 8u : (ST == kF32 ? 4u : (ST == kU8 ? 1u : 2u));
  static_assert(ST == kF32 || ST == kU8 || ST == kU16 || ST == kI16 || ST == kF64Src, "source element types of the one-launch prefilter");
  static_assert(!(PAD && ST == kF64Src), "the second pass of a two-pole order reads a plane that is padded already");
  typedef unsigned int u32x2_raw __attribute__((ext_vector_type(2)));
  using raw_t = typename std::conditional<ST == kF64Src, unsigned long long, uint32_t>::type;      // what a prefetch register holds
  const uint32_t voff = (uint32_t)gc * (uint32_t)f.in_ls * ES;
  const uint32_t rstep = (uint32_t)f.in_ss * ES;
  auto ld_raw = [&](uint32_t so) -> raw_t {                    // the element's bits, zero-extended
    if constexpr (ST == kF64Src) {
      const u32x2_raw v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, so, 0);
      return (unsigned long long)v.x | ((unsigned long long)v.y << 32);
    } else if constexpr (ST == kF32) return __builtin_amdgcn_raw_buffer_load_b32(rs, voff, so, 0);
    else if constexpr (ST == kU8) return (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rs, voff, so, 0);
    else return (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rs, voff, so, 0);
  };
  // (row is wave-uniform: the row offset travels in an SGPR.  Only the steps at the top and the bottom of the plane mirror their
  // rows -- a dozen scalar instructions per load, and a wave issues one instruction per four cycles whatever its kind)
  auto ld = [&](int row, auto mirrored) -> raw_t {
    int r = __builtin_amdgcn_readfirstlane(row);
    if constexpr (decltype(mirrored)::value) {
      r = r < 0 ? -r - sym : r;
      r = r >= H ? 2 * H - 2 + sym - r : r;
      r = max(0, min(r, H - 1));
    }
    if constexpr (PAD) {
      r -= pd.pad;                                             // row of the source image
      if constexpr (decltype(mirrored)::value) {               // (plain loads stay inside the image: no clamp, no zero rows)
        if (pd.zero_outside && (r < 0 || r >= pd.Hs)) return (raw_t)0;  // (the bits of 0.0f and of the integer 0)
        r = max(0, min(r, pd.Hs - 1));
      }
    }
    return ld_raw((uint32_t)r * rstep);
  };
  // (a column in zero padding: applied where a loaded value is USED -- a select right behind the load would make the wave wait for
  // every prefetched row at once: 91 us instead of 51 per 4096^2 plane)
  auto val = [&](raw_t bits) -> double {
    const raw_t b = (PAD && col_zero) ? (raw_t)0 : bits;
    if constexpr (ST == kF64Src) return __longlong_as_double((long long)b);
    else if constexpr (ST == kF32) return (double)__uint_as_float((uint32_t)b);
    else if constexpr (ST == kI16) return (double)(int)(short)b;
    else return (double)b;
  };
  // rows [plain_lo, plain_hi) of the plane are read as they are: no mirror, and with PAD inside the image
  const int plain_lo = PAD ? pd.pad : 0, plain_hi = PAD ? H - pd.pad : H;
  int r0 = Y0;                                                 // first row of the current step
  double tc = 0.0;                                             // the causal state
  double C[NC];                                                // causal values of rows r0 .. r0 + R + HP - 1
  {
    raw_t pre[HP], pre2[NC];
    if (r0 - HP >= plain_lo && r0 + NC <= plain_hi) {
#pragma unroll
      for (int j = 0; j < HP; ++j) pre[j] = ld(r0 - HP + j, std::false_type{});
#pragma unroll
      for (int j = 0; j < NC; ++j) pre2[j] = ld(r0 + j, std::false_type{});
    } else {
#pragma unroll
      for (int j = 0; j < HP; ++j) pre[j] = ld(r0 - HP + j, std::true_type{});
#pragma unroll
      for (int j = 0; j < NC; ++j) pre2[j] = ld(r0 + j, std::true_type{});
    }
#pragma unroll
    for (int j = 0; j < HP; ++j) tc = val(pre[j]) * lam + z * tc;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      tc = val(pre2[j]) * lam + z * tc;
      C[j] = tc;
    }
  }
  // ---- row pass geometry: thread (row_l, seg)
  const int row_l = tid % R, seg = tid / R;                    // 32 consecutive lanes = the 32 rows of one segment: conflict-free LDS passes
  const double* a = s_t + row_l * PITCH + seg * SEG;           // a[i]: column-filtered sample g0 - HP + seg SEG + i of this thread's row
  // (threads 252..255 have no column of their own: they recurse on a copy of column NCOL - 1 and all four write the tile's PAD column
  // -- index NCOL, the 253rd double of a row, which exists only to make the pitch odd and which nothing ever reads: a write sink, four
  // same-address stores of dead values per row, kept unpredicated so that the column pass stays branch-free)
  double* const colw = s_t + min(tid, NCOL);
  const int c_n = min(CORE, W - g0);
  const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)f.out, 0, (int)out_bytes, 0x00020000);
  const uint32_t ovoff = (uint32_t)(g0 + lane) * 8u, orow = (uint32_t)f.out_ls * 8u;
  for (;;) {
    PF2D_TRACE(0);
    // ---- anti-causal down the columns: back over the HP rows below the step from zero, then through the step into the tile
    {
      double ta = 0.0;
#pragma unroll
      for (int j = NC - 1; j >= 0; --j) {
        ta = z * (ta - C[j]);
        if (j < R) colw[j * PITCH] = ta;
      }
    }
    // the next step's new rows, in flight under the row pass (the chunk's last step, where nobody uses them, loads one cached row
    // R times instead: registers that are defined on one path only cost the register allocation a dozen spills)
    const bool more = r0 + R < Yend;                           // (workgroup-uniform)
    raw_t nx[R];
    auto issue_next_rows = [&]() {
      if (!more) {
#pragma unroll
        for (int j = 0; j < R; ++j) nx[j] = ld(min(r0, H - 1), std::true_type{});
      } else if (r0 + NC >= plain_lo && r0 + NC + R <= plain_hi) {
#pragma unroll
        for (int j = 0; j < R; ++j) nx[j] = ld(r0 + NC + j, std::false_type{});
      } else {
#pragma unroll
        for (int j = 0; j < R; ++j) nx[j] = ld(r0 + NC + j, std::true_type{});
      }
    };
    // (XCH: the loads go out behind the causal row pass instead -- its SEGC causal values and the column window's carried part would
    // not fit beside 32 prefetch registers; they still arrive under the anti-causal pass, the write-back and the store-out)
    if constexpr (!G::XCH) issue_next_rows();
    PF2D_TRACE(1);
    lds_barrier();                                             // the tile is complete
    PF2D_TRACE(2);
    // ---- row pass: causal from HP samples in front of the segment through SEG + HP samples, anti-causal back
    double cs[G::XCH ? SEG : J];
    if constexpr (G::XCH) {
      constexpr int SEGC = G::SEGC;
      {
        const double* ac = s_t + row_l * PITCH + seg * SEGC;   // warm-up over [seg SEGC, + HP), causal values of [HP + seg SEGC, + SEGC)
        double cc[SEGC];
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < HP; ++i) t = ac[i] * lam + z * t;
#pragma unroll
        for (int j = 0; j < SEGC; ++j) {
          t = ac[HP + j] * lam + z * t;
          cc[j] = t;
        }
        lds_barrier();                                         // every thread has read its column-filtered inputs
        double* w = s_t + row_l * PITCH + HP + seg * SEGC;
#pragma unroll
        for (int j = 0; j < SEGC; ++j) w[j] = cc[j];
      }
      issue_next_rows();
      lds_barrier();                                           // the causal values of the core and of the right halo are in the tile
      {
        const double* cr = s_t + row_l * PITCH + HP + seg * SEG;   // this thread's SEG outputs, then HP causal values to warm up on
        double t = 0.0;
#pragma unroll
        for (int i = SEG + HP - 1; i >= SEG; --i) t = z * (t - cr[i]);
#pragma unroll
        for (int j = SEG - 1; j >= 0; --j) {
          t = z * (t - cr[j]);
          cs[j] = t;
        }
      }
    } else {
      double t = 0.0;
#pragma unroll
      for (int i = 0; i < HP; ++i) t = a[i] * lam + z * t;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        t = a[HP + j] * lam + z * t;
        cs[j] = t;
      }
      t = 0.0;
#pragma unroll
      for (int j = J - 1; j >= 0; --j) {
        t = z * (t - cs[j]);
        cs[j] = t;
      }
    }
    PF2D_TRACE(3);
    lds_barrier();                                             // every thread has read its inputs: the results may overwrite them
    {
      double* w = s_t + row_l * PITCH + seg * SEG + HP;
#pragma unroll
      for (int j = 0; j < SEG; ++j) w[j] = cs[j];
    }
    lds_barrier();
    PF2D_TRACE(4);
    // ---- the core of the step, contiguous row pieces (8 bytes per lane: the rows dealt to the lanes as one list of pairs -- 16-byte
    // stores, a wave's kilobyte straddling two row pieces -- measured 3 % slower, profiles/round_5/r05p_ab_store16.txt)
#pragma unroll
    for (int q = 0; q < R / 4; ++q) {
      const int rr = q * 4 + wave;
      if (r0 + rr >= Yend) break;
      const uint32_t so = (uint32_t)(r0 + rr) * orow;          // (wave-uniform: an SGPR)
      const double* r = s_t + rr * PITCH + HP + lane;
#pragma unroll
      for (int k = 0; k < (CORE + 63) / 64; ++k) {
        if (lane + 64 * k < c_n) {
          typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
          const unsigned long long b = (unsigned long long)__double_as_longlong(r[64 * k]);
          const u32x2_t pk = {(uint32_t)b, (uint32_t)(b >> 32)};
          __builtin_amdgcn_raw_buffer_store_b64(pk, ors, ovoff + 512u * k, so, 0);
        }
      }
    }
    PF2D_TRACE(5);
    if (!more) break;
    // ---- the next step: the window moves down R rows, the causal recursion runs on through the new rows
    r0 += R;
#pragma unroll
    for (int j = 0; j < HP; ++j) C[j] = C[j + R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      tc = val(nx[j]) * lam + z * tc;
      C[HP + j] = tc;
    }
    r0 -= R;
    PF2D_TRACE(6);
    lds_barrier();                                             // the tile has been read out: the next step may overwrite it
    PF2D_TRACE(7);
    r0 += R;
  }
}

template <int HP, bool PAD = false, int ST = kF32, int RR = 32>
__global__ void __launch_bounds__(256, DCP_PF2D_WAVES) spline_prefilter2d_kernel(const TileFilter f, const Pf2dPad pd, const uint32_t src_bytes, const uint32_t out_bytes,
                                                                                 const int chunk_rows, const int stripes, const int chunks, const int xcd_order) {
  using G = Pf2d<HP, RR>;
  __shared__ double s_t[G::R * G::PITCH];
  // Workgroups go round-robin to the 8 XCDs (blockIdx.x & 7), each with its own L2: XCD k takes a contiguous run of tiles in
  // row-major order (stripe fastest), so that the stripes that share 2 HP columns -- neighbours that read the same source rows
  // at about the same time -- meet in one L2 (FETCH_SIZE x 2: 162 -> 93 MB per 4096^2 frame, profiles/round_5/r05k_pmc_spline*.txt)
  const int ntiles = stripes * chunks;
  const int xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
  const int q = ntiles >> 3, rem = ntiles & 7;
  const int t = xcd_order ? xcd * q + min(xcd, rem) + j : (int)blockIdx.x;
  if (xcd_order && j >= q + (xcd < rem ? 1 : 0)) return;        // (the grid is rounded up to a multiple of 8)
  const int chunk = t / stripes, stripe = t - chunk * stripes;
  const int Y0 = chunk * chunk_rows;                           // first row the chunk writes
  const int Yend = min(f.n, Y0 + chunk_rows);
  pf2d_body<HP, PAD, ST, RR>(f, pd, src_bytes, out_bytes, stripe, Y0, Yend, s_t);
}

// ---- the row pass (axis 1) as a cross-lane scan: no LDS, no barrier, every access a contiguous 512 bytes ---------------
// Along a row the recursion runs ACROSS the lanes of a wave.  c_i = x_i + z c_(i-1) over the 64 samples of a block is an
// inclusive scan with multiplier z -- six steps v_i += z^d v_(i-d), d = 1, 2, 4, .., 32 (the neighbour's value through the LDS
// crossbar, ds_bpermute: no memory) -- plus z^(i+1) times the value carried in from the block before; the anti-causal pass is the
// same scan mirrored.  A wave owns one row and kRsNB consecutive blocks: it starts one block early from a zero carry (64 >= hp
// samples: the 2^-64 restart of the other kernels; or from the exact initial value at the start of the row), keeps the causal
// values of its blocks in registers (one double per lane and block), reads one block past its own as the anti-causal warm-up
// (or takes the exact end value) and comes back storing.  (kRsNB + 2) / kRsNB = 1.25 loads per sample, ~0.8 instructions per
// sample, 32 768 independent waves for a 4096^2 plane.  The sums associate differently from the sequential recursion
// (t = x lam + z t), so a coefficient may differ from the tile kernel's / the oracle's in its last bits -- as it does at every
// restart of those kernels; the float32 result of the interpolation flips with probability ~1e-8 per pixel (tests: <= 4 pixels
// of a frame against the plain kernels, <= 8 against the oracle, as before).  One pole (orders 2, 3), reflect / mirror kinds
// on rows long enough that z^n underflows.
constexpr int kRsNB = 8;
__device__ __forceinline__ double lane_read(double v, int src_lane) {       // v of lane `src_lane` (any lane pattern; LDS crossbar)
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_uniform(double v, int src_lane) {    // v of lane `src_lane` as a scalar
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src_lane), __builtin_amdgcn_readlane(__double2loint(v), src_lane));
}
// v through a DPP lane pattern (VALU, no LDS); lanes the pattern does not reach read 0 (bound_ctrl: no `old` operand to initialise)
template <int CTRL>
__device__ __forceinline__ double dpp_read(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

__global__ void __launch_bounds__(256) spline_row_scan_kernel(const TileFilter f) {
  constexpr int NB = kRsNB;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
  const int row = (int)blockIdx.y * 4 + wave;
  const int n = f.n, nbt = (n + 63) >> 6;                      // blocks of the row
  const int b0 = (int)blockIdx.x * NB;
  if (row >= f.nlines || b0 >= nbt) return;
  const int b1 = min(nbt, b0 + NB);                            // the wave writes blocks [b0, b1); b1 < nbt: block b1 is its look-ahead
  const double* __restrict__ in = (const double*)f.in + (int64_t)row * f.in_ls;
  double* __restrict__ out = f.out + (int64_t)row * f.out_ls;
  const double z = f.z[0], lam = f.lam;
  // blocks b0 - 1 .. b0 + NB, all in flight (samples outside the row read as 0)
  double xs[NB + 2];
#pragma unroll
  for (int q = 0; q < NB + 2; ++q) {
    const int idx = (b0 - 1 + q) * 64 + lane;
    xs[q] = (idx >= 0 && idx < n && b0 - 1 + q <= b1) ? in[idx] : 0.0;
  }
  // z^1, z^2, z^4, .. z^32 (uniform) and the per-lane powers of the scan
  double zp[6];
  zp[0] = z;
#pragma unroll
  for (int k = 1; k < 6; ++k) zp[k] = zp[k - 1] * zp[k - 1];
  const double z64 = zp[5] * zp[5];
  auto zpow = [&](int e) {                                     // z^e, 0 <= e <= 64
    double r = (e & 64) ? z64 : 1.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) r = (e >> k) & 1 ? r * zp[k] : r;
    return r;
  };
  // (the two row_bcast steps reach every row; the rows that must not take the value multiply it by 0)
  const double z_up = zpow(lane + 1), z_l = zpow(lane), z_r16 = (lane & 16) ? zpow((lane & 15) + 1) : 0.0, z_r32 = (lane & 32) ? zpow((lane & 31) + 1) : 0.0;
  // inclusive scan with multiplier z over the 64 lanes: s_i = sum_(k <= i) z^(i - k) v_k.  Four Kogge-Stone steps inside the rows
  // of 16 lanes (DPP row_shr), then lane 15 of rows 0 / 2 into rows 1 / 3 and lane 31 into rows 2 and 3 (DPP row_bcast)
  auto wscan = [&](double v) {
    v = __builtin_fma(zp[0], dpp_read<0x111>(v), v);
    v = __builtin_fma(zp[1], dpp_read<0x112>(v), v);
    v = __builtin_fma(zp[2], dpp_read<0x114>(v), v);
    v = __builtin_fma(zp[3], dpp_read<0x118>(v), v);
    v = __builtin_fma(z_r16, dpp_read<0x142>(v), v);
    v = __builtin_fma(z_r32, dpp_read<0x143>(v), v);
    return v;
  };
  // ---- causal
  double c[NB + 2];
  double carry = 0.0;
#pragma unroll
  for (int q = 0; q < NB + 2; ++q) {
    c[q] = 0.0;
    const int bq = b0 - 1 + q;
    if (bq < 0 || bq > b1 || bq >= nbt) continue;              // (wave-uniform)
    double v = xs[q] * lam;
    if (bq == 0) {
      // exact value of sample 0 with z^n dropped: S = sum_k z^k x_k lam over the first samples (z^64 ~ 1e-37: one block holds
      // everything that can reach a double); mirror: c_0 = S, reflect: c_0 = S z + x_0 lam
      double s_ = v * z_l;
      if (f.kind != kSplReflect && lane > n - 2) s_ = 0.0;     // (the mirror sum runs to n - 2)
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) s_ += lane_read(s_, lane ^ d);
      const double x0 = wave_uniform(v, 0);
      const double c0 = f.kind == kSplReflect ? s_ * z + x0 : s_;
      v = lane == 0 ? c0 : v;
    }
    v = wscan(v);
    v = __builtin_fma(z_up, carry, v);
    carry = wave_uniform(v, 63);
    c[q] = v;
  }
  // ---- anti-causal, on the REVERSED lanes of a block (lane j <-> sample 63 - j: y_i = z (y_(i+1) - c_i) becomes the same forward
  // scan), storing the wave's own blocks -- a wave's store is still one contiguous 512 bytes
  carry = 0.0;
  const double kend_r = z / (z - 1.0), kend_m = z / (z * z - 1.0);
#pragma unroll
  for (int q = NB + 1; q >= 1; --q) {
    const int bq = b0 - 1 + q;
    if (bq > b1 || bq >= nbt) continue;
    const int idx = bq * 64 + 63 - lane;
    const double cr = lane_read(c[q], 63 - lane);
    double u = idx < n ? -z * cr : 0.0;
    if (bq == nbt - 1) {
      // exact value of the last sample (sample L of this block)
      const int L = (n - 1) & 63;
      const double cl = wave_uniform(c[q], L);
      double ye;
      if (f.kind == kSplReflect) {
        ye = cl * kend_r;
      } else {
        const double cm = L > 0 ? wave_uniform(c[q], L > 0 ? L - 1 : 0) : wave_uniform(c[q - 1], 63);
        ye = kend_m * (cl + z * cm);
      }
      u = lane == 63 - L ? ye : u;
    }
    u = wscan(u);
    u = __builtin_fma(z_up, carry, u);
    carry = wave_uniform(u, 63);
    if (bq < b1 && idx < n) out[idx] = u;
  }
}

// (rows x cols) -> (cols x rows), 32 x 32 tiles through LDS
__global__ void __launch_bounds__(kSplBlock) spline_transpose_kernel(const double* in, double* out, int rows, int cols) {
  __shared__ double tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 x 8
  for (int j = ty; j < 32; j += 8)
    if (by + j < rows && bx + tx < cols) tile[j][tx] = in[(size_t)(by + j) * cols + (bx + tx)];
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (bx + j < cols && by + tx < rows) out[(size_t)(bx + j) * rows + (by + tx)] = tile[tx][j];
}

// x / D, correctly rounded, for the constants of the weight polynomials: q = x * RN(1/D), one exact residual, one
// correction (Markstein) -- three instructions where the compiler's IEEE division takes about twenty, and the weights
// need up to eight divisions per pixel.  |x| is O(1) here: nothing leaves the normal range.
template <int D>
__device__ __forceinline__ double div_c(double x) {
  constexpr double r = 1.0 / (double)D;
  const double q = x * r;
  const double e = __builtin_fma(-(double)D, q, x);
  return __builtin_fma(e, r, q);
}

// centred B-spline weights (the expressions of spline_weights() in the oracle); returns the first tap
// FAST (the factorised gather only, where the sum is not scipy's to the last bit anyway): the cubic weights as fused polynomials,
// y^2 (y / 2 - 1) + 2 / 3 and z^3 / 6 -- 15 operations per axis instead of 27; each weight within one float64 ulp of scipy's form.
template <int ORDER, bool FAST = false>
__device__ __forceinline__ int spline_weights(double x, double* w) {
  double s;
  if constexpr (ORDER & 1) s = __builtin_floor(x);
  else s = __builtin_floor(x + 0.5);
  const double t = x - s;
  const int start = (int)s - ORDER / 2;
  double y = t, z = 1.0 - t, t2;
  if constexpr (ORDER == 3 && FAST) {
    w[1] = __builtin_fma(y * y, __builtin_fma(y, 0.5, -1.0), 2.0 / 3.0);
    w[2] = __builtin_fma(z * z, __builtin_fma(z, 0.5, -1.0), 2.0 / 3.0);
    w[0] = (z * z) * (z * (1.0 / 6.0));
    w[3] = 1.0 - w[0] - w[1] - w[2];
  } else if constexpr (ORDER == 4 && FAST) {
    // (round 6) the quartic and quintic weights as fused Horner chains in the same variables as scipy's expressions: 21 / 29 operations
    // per axis instead of 37 / 53, every weight within a few float64 ulps of scipy's form (the constants 1/6, 1/24, 1/120 rounded once)
    t2 = t * t;
    w[2] = __builtin_fma(t2, __builtin_fma(t2, 0.25, -0.625), 115.0 / 192.0);
    y = 1.0 + t;
    z = 1.0 - t;
    w[1] = __builtin_fma(y, __builtin_fma(y, __builtin_fma(y, __builtin_fma(y, -1.0 / 6.0, 5.0 / 6.0), -1.25), 5.0 / 24.0), 55.0 / 96.0);
    w[3] = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, -1.0 / 6.0, 5.0 / 6.0), -1.25), 5.0 / 24.0), 55.0 / 96.0);
    y = 0.5 - t;
    y *= y;
    w[0] = (y * y) * (1.0 / 24.0);
    w[4] = 1.0 - w[0] - w[1] - w[2] - w[3];
  } else if constexpr (ORDER == 5 && FAST) {
    t2 = y * y;
    w[2] = __builtin_fma(t2, __builtin_fma(t2, __builtin_fma(y, -1.0 / 12.0, 0.25), -0.5), 0.55);
    t2 = z * z;
    w[3] = __builtin_fma(t2, __builtin_fma(t2, __builtin_fma(z, -1.0 / 12.0, 0.25), -0.5), 0.55);
    w[0] = (t2 * t2) * (z * (1.0 / 120.0));
    y += 1.0;
    w[1] = __builtin_fma(y, __builtin_fma(y, __builtin_fma(y, __builtin_fma(y, __builtin_fma(y, 1.0 / 24.0, -0.375), 1.25), -1.75), 0.625), 0.425);
    y = z + 1.0;
    w[4] = __builtin_fma(y, __builtin_fma(y, __builtin_fma(y, __builtin_fma(y, __builtin_fma(y, 1.0 / 24.0, -0.375), 1.25), -1.75), 0.625), 0.425);
    w[5] = 1.0 - w[0] - w[1] - w[2] - w[3] - w[4];
  } else if constexpr (ORDER == 2) {
    w[1] = 0.75 - t * t;
    y = 0.5 + t;
    w[2] = 0.5 * y * y;
    w[0] = 1.0 - w[1] - w[2];
  } else if constexpr (ORDER == 3) {
    w[1] = div_c<6>(y * y * (y - 2.0) * 3.0 + 4.0);
    w[2] = div_c<6>(z * z * (z - 2.0) * 3.0 + 4.0);
    w[0] = div_c<6>(z * z * z);
    w[3] = 1.0 - w[0] - w[1] - w[2];
  } else if constexpr (ORDER == 4) {
    t2 = t * t;
    w[2] = t2 * (t2 * 0.25 - 0.625) + 115.0 / 192.0;
    y = 1.0 + t;
    z = 1.0 - t;
    w[1] = y * (y * (div_c<6>(y * (5.0 - y)) - 1.25) + 5.0 / 24.0) + 55.0 / 96.0;
    w[3] = z * (z * (div_c<6>(z * (5.0 - z)) - 1.25) + 5.0 / 24.0) + 55.0 / 96.0;
    y = 0.5 - t;
    y *= y;
    w[0] = div_c<24>(y * y);
    w[4] = 1.0 - w[0] - w[1] - w[2] - w[3];
  } else {
    t2 = y * y;
    w[2] = t2 * (t2 * (0.25 - div_c<12>(y)) - 0.5) + 0.55;
    t2 = z * z;
    w[3] = t2 * (t2 * (0.25 - div_c<12>(z)) - 0.5) + 0.55;
    y += 1.0;
    w[1] = y * (y * (y * (y * (div_c<24>(y) - 0.375) + 1.25) - 1.75) + 0.625) + 0.425;
    y = z + 1.0;
    w[4] = y * (y * (y * (y * (div_c<24>(y) - 0.375) + 1.25) - 1.75) + 0.625) + 0.425;
    t2 = z * z;
    w[0] = div_c<120>(t2 * t2 * z);
    w[5] = 1.0 - w[0] - w[1] - w[2] - w[3] - w[4];
  }
  return start;
}

__device__ __forceinline__ int spline_fold(int i, int n, int mode) {
  if (i >= 0 && i < n) return i;
  if (mode == kModeReflect || mode == kModeGridMirror) {
    const int s2 = 2 * n;
    i %= s2;
    if (i < 0) i += s2;
    return i < n ? i : s2 - 1 - i;
  }
  if (mode == kModeGridWrap) {
    i %= n;
    return i < 0 ? i + n : i;
  }
  if (mode == kModeNearest || mode == kModeGridConstant) return i < 0 ? 0 : n - 1;
  if (n == 1) return 0;
  const int s2 = 2 * n - 2;
  i %= s2;
  if (i < 0) i += s2;
  return i < n ? i : s2 - i;
}

// MAPKIND 0 radial, 1 perspective, 2 explicit coordinates (dst[i] for point i), 3 fused perspective -> radial
template <int MAPKIND, int ORDER>
__global__ void __launch_bounds__(kSplBlock) spline_remap_kernel(const SplineArgs a, const MapArgs map, const CoordArgs ca,
                                                                void* dst) {
  int64_t i;
  int x = 0, y = 0;
  if constexpr (MAPKIND == 2) {
    i = (int64_t)blockIdx.x * kSplBlock + threadIdx.x;
    if (i >= ca.npts) return;
  } else {       // blockIdx.y walks the rows (no 64-bit division)
    x = blockIdx.x * kSplBlock + (int)threadIdx.x;
    y = blockIdx.y + blockIdx.z * 65535;
    if (x >= a.W || y >= a.H) return;
    i = (int64_t)y * a.W + x;
  }
  double yc, xc;   // float32-rounded (or caller-supplied) coordinates in the unpadded image
  if constexpr (MAPKIND == 2) {
    if (ca.is_f64) {
      yc = ((const double*)ca.ycoord)[i];
      xc = ((const double*)ca.xcoord)[i];
    } else {
      yc = (double)((const float*)ca.ycoord)[i];
      xc = (double)((const float*)ca.xcoord)[i];
    }
    // a caller's coordinate outside the image: what scipy does before it evaluates the spline (spline_map_point() in the
    // oracle) -- 'constant' returns cval = 0, 'nearest' and 'grid-constant' evaluate where the coordinate is (taps outside
    // the padded plane clamp to its edge / read 0), every other mode moves it into the extended image
    if (!(yc >= 0.0 && yc <= (double)(a.H - 1) && xc >= 0.0 && xc <= (double)(a.W - 1)) && a.mode != kModeNearest &&
        a.mode != kModeGridConstant) {
      yc = mc_map_coordinate(yc, a.H, a.mode);
      xc = mc_map_coordinate(xc, a.W, a.mode);
      if (a.mode == kModeConstant && (yc <= -1.0 || xc <= -1.0)) {
        store_any(dst, a.dst_dtype, (size_t)i, 0.0);
        return;
      }
    }
    // (tap indices are 32-bit: a coordinate a billion samples out already has every tap outside the plane)
    yc = yc < -1.0e9 ? -1.0e9 : (yc > 1.0e9 ? 1.0e9 : yc);
    xc = xc < -1.0e9 ? -1.0e9 : (xc > 1.0e9 ? 1.0e9 : xc);
  } else {
    const float wmaxf = (float)(a.W - 1), hmaxf = (float)(a.H - 1);
    double xd, yd;
    pixel_coord<MAPKIND == 0 ? kRadial : MAPKIND == 1 ? kPersp : kFused>(map, (double)x, (double)y, wmaxf, hmaxf, &xd, &yd);
    xc = (double)round_clip_f32(xd, wmaxf);
    yc = (double)round_clip_f32(yd, hmaxf);
  }
  double wy[6], wx[6];
  const int sy = spline_weights<ORDER>(yc + (double)a.pad, wy);
  const int sx = spline_weights<ORDER>(xc + (double)a.pad, wx);
  int ix[ORDER + 1];
#pragma unroll
  for (int k = 0; k <= ORDER; ++k) ix[k] = spline_fold(sx + k, a.Wp, a.mode);
  double t = 0.0;
  if (MAPKIND == 2 && a.mode == kModeGridConstant) {
    // explicit coordinates under 'grid-constant' may put taps outside the padded plane: those read cval = 0
#pragma unroll
    for (int j = 0; j <= ORDER; ++j) {
      const bool yout = sy + j < 0 || sy + j >= a.Hp;
      const double* row = a.coef + (size_t)spline_fold(sy + j, a.Hp, a.mode) * (size_t)a.Wp;
#pragma unroll
      for (int k = 0; k <= ORDER; ++k) t += (((yout || sx + k < 0 || sx + k >= a.Wp) ? 0.0 : row[ix[k]]) * wy[j]) * wx[k];
    }
  } else {
#pragma unroll
    for (int j = 0; j <= ORDER; ++j) {
      const double* row = a.coef + (size_t)spline_fold(sy + j, a.Hp, a.mode) * (size_t)a.Wp;
#pragma unroll
      for (int k = 0; k <= ORDER; ++k) t += (row[ix[k]] * wy[j]) * wx[k];
    }
  }
  store_any(dst, a.dst_dtype, (size_t)i, t);
}

// ---- the gather out of LDS ---------------------------------------------------------------------
// spline_remap_kernel's arithmetic with remap_wg_kernel's data path (unwarp_kernels.hip): a workgroup owns a 128 x 32
// tile of output pixels (four waves, 2 x 2 sub-tiles of 64 x 16), evaluates the map at the tile's four corner pixels,
// and -- under the host's tile-deviation certificate (MapArgs::tile_dev_ok >= 2: every coordinate of the tile within
// 0.90 px of the bilinear interpolant of the corners) -- copies the box of float64 coefficients that holds every tap of
// every pixel into LDS by LDS-DMA, the loads going out between the coordinate rows of phase 1.  The (order + 1)^2
// taps of a pixel are then LDS reads instead of global gathers.  A tile whose box does not fit the slab or reaches
// over the edge of the coefficient plane (there the taps fold according to the boundary mode) takes the global
// gather of spline_remap_kernel for all its pixels.  Radial and perspective maps.
constexpr int kSwTW = 128, kSwTH = 32;             // workgroup tile
constexpr int kSwBoxW = 144, kSwBoxH = 45;         // slab: 144 x 45 float64 = 51 840 B; with the row tables three workgroups per CU
constexpr int kSwCH = kSwBoxW * 8 / 16;            // 16-byte chunks per slab row
constexpr int kSwNJ = (kSwBoxH * kSwCH + 255) / 256;   // loads per wave that cover the slab: 13

// NF: length of the radial polynomial (5: coefficients from the kernel arguments, shorter vectors padded with zeros by the
// launcher -- fma(r2, 0, a) = a exactly; -1: any length, coefficients staged in LDS).  Phase 1 is remap_wg_kernel's: a row table
// per wave and the hoisted evaluation map_coord (dcp_device.h) -- round 2 walked the coefficient vector with one scalar load and
// one wait per coefficient and pixel row.
template <int KIND, int ORDER, int NF, bool EXACT>
__global__ void __launch_bounds__(256, ORDER >= 5 ? 2 : 3) spline_wg_kernel(const SplineArgs a, const MapArgs map, void* dst) {
  constexpr int RW = KIND == kRadial ? 2 : 4;
  __shared__ __attribute__((aligned(16))) unsigned char s_box[kSwBoxH * kSwBoxW * 8];
  __shared__ double s_row[4][16][RW];                                // one row table per wave: no barrier before it is read
  __shared__ double s_coef[NF < 0 ? kMaxFact : 1];
  static_assert(sizeof(s_box) + sizeof(s_row) + sizeof(s_coef) <= 160 * 1024 / 3, "three workgroups per CU");
  typedef __attribute__((address_space(3))) void* lds_ptr;
  constexpr int PB = kSwBoxW * 8;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = (int)threadIdx.x & 63;
  const int wx = wave & 1, wy = wave >> 1;
  // tile order: workgroups go round-robin to the 8 XCDs (blockIdx.x & 7: the grid's x extent is a multiple of 8), each with its own
  // L2 -- XCD s owns a run of neighbouring tile columns and sweeps it row by row, so the box columns and rows that neighbouring
  // tiles share are fetched into one L2 (remap_wg_kernel's order; a.xcd_remap = 0: plain row-major, A/B)
  int tx = blockIdx.x;
  const int ty = blockIdx.y;
  if (a.xcd_remap) {
    const int tiles_x = (a.W + kSwTW - 1) / kSwTW;
    const int s_ = (int)blockIdx.x & 7, c_ = (int)blockIdx.x >> 3;
    const int wq = tiles_x >> 3, wr = tiles_x & 7;
    if (c_ >= wq + (s_ < wr ? 1 : 0)) return;                // (workgroup-uniform, before any barrier)
    tx = s_ * wq + min(s_, wr) + c_;
  }
  const int y0 = __builtin_amdgcn_readfirstlane(ty * kSwTH + wy * 16);
  const int x = tx * kSwTW + wx * 64 + lane;
  const float wmaxf = (float)(a.W - 1), hmaxf = (float)(a.H - 1);
  // ---- corner pixels (lanes 0..3) -> hull of their taps' base positions in the padded plane
  int cx0, cx1, cy0, cy1;
  {
    const double X = (double)min(tx * kSwTW + (lane & 1) * (kSwTW - 1), a.W - 1);
    const double Y = (double)min(ty * kSwTH + ((lane >> 1) & 1) * (kSwTH - 1), a.H - 1);
    double xd, yd;
    corner_coord<KIND, NF>(map, X, Y, &xd, &yd);
    const int cxi = (int)round_clip_f32(xd, wmaxf) + a.pad, cyi = (int)round_clip_f32(yd, hmaxf) + a.pad;
    const int xa = __builtin_amdgcn_readlane(cxi, 0), xb = __builtin_amdgcn_readlane(cxi, 1);
    const int xc_ = __builtin_amdgcn_readlane(cxi, 2), xd_ = __builtin_amdgcn_readlane(cxi, 3);
    const int ya = __builtin_amdgcn_readlane(cyi, 0), yb = __builtin_amdgcn_readlane(cyi, 1);
    const int yc_ = __builtin_amdgcn_readlane(cyi, 2), yd_ = __builtin_amdgcn_readlane(cyi, 3);
    cx0 = min(min(xa, xb), min(xc_, xd_));
    cx1 = max(max(xa, xb), max(xc_, xd_));
    cy0 = min(min(ya, yb), min(yc_, yd_));
    cy1 = max(max(ya, yb), max(yc_, yd_));
  }
  // floor(coordinate) lies in [c0 - 1, c1 + 1]; odd orders tap floor - ORDER/2 .. + ORDER, even orders
  // floor(coordinate + 0.5) - ORDER/2 .. + ORDER: all inside [c0 - 1 - ORDER/2, c1 + 2 + (ORDER + 1)/2]
  const int bx0 = cx0 - 1 - ORDER / 2, bx1 = cx1 + 2 + (ORDER + 1) / 2;
  const int by0 = cy0 - 1 - ORDER / 2, by1 = cy1 + 2 + (ORDER + 1) / 2;
  const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
  // staged: the box fits the slab and lies inside the plane (no tap folds); workgroup-uniform
  const bool staged = bw <= kSwBoxW && bh <= kSwBoxH && bx0 >= 0 && by0 >= 0 && bx1 <= a.Wp - 1 && by1 <= a.Hp - 1;
  // the fill's descriptor ends with the box's last row: a chunk of a later row is out of range (zeros, no memory access), which
  // replaces a per-lane row test in every load (remap_wg_kernel)
  const uint32_t rstep = (uint32_t)a.Wp * 8u;
  const unsigned long long plane_bytes = (unsigned long long)a.Hp * rstep, rows_end = (unsigned long long)(by1 + 1) * rstep;
  const __amdgpu_buffer_rsrc_t src_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.coef, 0, (int)(uint32_t)(staged && rows_end < plane_bytes ? rows_end : plane_bytes), 0x00020000);
  const int fc = wave * 64 + lane;
  const int crow0 = fc / kSwCH;
  const int c160 = fc - crow0 * kSwCH;
  const uint32_t off0 = ((uint32_t)by0 * (uint32_t)a.Wp + (uint32_t)bx0) * 8u + (uint32_t)crow0 * rstep + (uint32_t)c160 * 16u;
  const int nchunk = bh * kSwCH;
  auto issue_fill = [&](auto jc) {
    constexpr int j = decltype(jc)::value;
    if constexpr (j < kSwNJ) {
      if (staged && (j * 4 + wave) * 64 < nchunk) {
        constexpr int qrow = (256 * j) / kSwCH, rem = (256 * j) % kSwCH;
        const bool wrap = c160 >= kSwCH - rem;
        const uint32_t step_nowrap = (uint32_t)qrow * rstep + (uint32_t)rem * 16u, step_wrap = step_nowrap + rstep - (uint32_t)PB;
        // The slab holds kSwBoxH * kSwCH = 3240 chunks, the kSwNJ = 13 loads of the four waves 3328: the LAST load of a box of full
        // height would write its trailing lanes -- zeros, their rows are past the descriptor's extent -- behind the slab, into the row
        // tables that follow it in LDS (found in round 4 by the large-frame campaign: tiles with the tallest boxes came out with the
        // coordinates of some rows zeroed, differently from run to run).  Those lanes are masked; every other load lies inside.
        if constexpr ((j * 4 + 4) * 64 > kSwBoxH * kSwCH) {
          if ((j * 4 + wave) * 64 + lane < kSwBoxH * kSwCH)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rsrc, (lds_ptr)(s_box + (j * 4 + wave) * 1024), 16, off0 + (wrap ? step_wrap : step_nowrap), 0, 0, 0);
        } else {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rsrc, (lds_ptr)(s_box + (j * 4 + wave) * 1024), 16, off0 + (wrap ? step_wrap : step_nowrap), 0, 0, 0);
        }
      }
    }
  };
  // ---- row table of this wave's 16 rows (lanes 0..15; same-wave LDS traffic is ordered, no barrier)
  if (lane < 16) fill_row<KIND, RW>(map, s_row[wave], lane, (double)min(y0 + lane, a.H - 1));
  if constexpr (NF < 0 && KIND != kPersp) {
    if ((int)threadIdx.x < map.nfact) s_coef[threadIdx.x] = map.fact[threadIdx.x];
    __syncthreads();
  }
  // ---- phase 1: the float32 coordinates of this wave's 16 rows, a load going out in front of each of the first 13
  const int rows = __builtin_amdgcn_readfirstlane(max(0, min(16, a.H - y0)));
  const ColCtx col = make_col<KIND, NF>(map, min(x, a.W - 1));
  float xf[16], yf[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (k == 0) issue_fill(std::integral_constant<int, 0>{});
    if (k == 1) issue_fill(std::integral_constant<int, 1>{});
    if (k == 2) issue_fill(std::integral_constant<int, 2>{});
    if (k == 3) issue_fill(std::integral_constant<int, 3>{});
    if (k == 4) issue_fill(std::integral_constant<int, 4>{});
    if (k == 5) issue_fill(std::integral_constant<int, 5>{});
    if (k == 6) issue_fill(std::integral_constant<int, 6>{});
    if (k == 7) issue_fill(std::integral_constant<int, 7>{});
    if (k == 8) issue_fill(std::integral_constant<int, 8>{});
    if (k == 9) issue_fill(std::integral_constant<int, 9>{});
    if (k == 10) issue_fill(std::integral_constant<int, 10>{});
    if (k == 11) issue_fill(std::integral_constant<int, 11>{});
    if (k == 12) issue_fill(std::integral_constant<int, 12>{});
    static_assert(kSwNJ <= 13, "one load per coordinate row");
    double xd, yd;
    map_coord<KIND, NF, RW>(map, s_row[wave], s_coef, col, k, wmaxf, hmaxf, &xd, &yd);
    xf[k] = round_clip_f32(xd, wmaxf);
    yf[k] = round_clip_f32(yd, hmaxf);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (rows == 0 || x >= a.W) return;
  // ---- phase 2
  const double padd = (double)a.pad;
  if (staged) {
    const int org = by0 * PB + bx0 * 8;
    auto value = [&](int k) -> double {
      double wyv[6], wxv[6];
      int sy, sx;
      if constexpr (EXACT) {
        sy = spline_weights<ORDER>((double)yf[k] + padd, wyv);
        sx = spline_weights<ORDER>((double)xf[k] + padd, wxv);
      } else {
        sy = spline_weights<ORDER, true>((double)yf[k] + padd, wyv);
        sx = spline_weights<ORDER, true>((double)xf[k] + padd, wxv);
      }
      DCP_BOUNDS(sy * PB + sx * 8 - org, ORDER * PB + (ORDER + 1) * 8, sizeof(s_box), 7);
      const unsigned char* base = s_box + (sy * PB + sx * 8 - org);
      double t = 0.0;
      if constexpr (EXACT) {             // scipy's order: t += (c * wy) * wx, tap by tap
#pragma unroll
        for (int j = 0; j <= ORDER; ++j) {
          const double* row = (const double*)(base + j * PB);
#pragma unroll
          for (int q = 0; q <= ORDER; ++q) t += (row[q] * wyv[j]) * wxv[q];
        }
      } else {                           // factorised: sum_j wy_j (sum_q c_jq wx_q), fused -- (ORDER + 1)(ORDER + 2) operations instead of 3 (ORDER + 1)^2
#pragma unroll
        for (int j = 0; j <= ORDER; ++j) {
          const double* row = (const double*)(base + j * PB);
          double r = row[0] * wxv[0];
#pragma unroll
          for (int q = 1; q <= ORDER; ++q) r = __builtin_fma(row[q], wxv[q], r);
          t = j == 0 ? r * wyv[0] : __builtin_fma(r, wyv[j], t);
        }
      }
      return t;
    };
    if (a.dst_dtype == kF32 && (uint64_t)a.H * (uint64_t)a.W * 4u < (1ull << 32)) {
      // float32 results (the common case) through a buffer descriptor: the row offset is a scalar, no 64-bit address per store
      const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)((uint32_t)a.H * (uint32_t)a.W * 4u), 0x00020000);
      const uint32_t xoff = ((uint32_t)y0 * (uint32_t)a.W + (uint32_t)x) * 4u, row_bytes = (uint32_t)a.W * 4u;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (k >= rows) continue;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)value(k)), drs, xoff, (uint32_t)k * row_bytes, DCP_SPLINE_OUT_AUX);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (k >= rows) continue;
        store_any(dst, a.dst_dtype, (size_t)(y0 + k) * (size_t)a.W + (size_t)x, value(k));
      }
    }
  } else {
    // (rare: a box that reaches over the plane's edge or does not fit.  The coordinates are evaluated again in a rolled loop --
    // the same values -- instead of sixteen copies of the folding gather)
#pragma unroll 1
    for (int k = 0; k < rows; ++k) {
      double xd, yd;
      map_coord<KIND, NF, RW>(map, s_row[wave], s_coef, col, k, wmaxf, hmaxf, &xd, &yd);
      double wyv[6], wxv[6];
      const int sy = spline_weights<ORDER>((double)round_clip_f32(yd, hmaxf) + padd, wyv);
      const int sx = spline_weights<ORDER>((double)round_clip_f32(xd, wmaxf) + padd, wxv);
      int ix[ORDER + 1];
#pragma unroll
      for (int q = 0; q <= ORDER; ++q) ix[q] = spline_fold(sx + q, a.Wp, a.mode);
      double t = 0.0;
#pragma unroll
      for (int j = 0; j <= ORDER; ++j) {
        const double* row = a.coef + (size_t)spline_fold(sy + j, a.Hp, a.mode) * (size_t)a.Wp;
#pragma unroll
        for (int q = 0; q <= ORDER; ++q) t += (row[ix[q]] * wyv[j]) * wxv[q];
      }
      store_any(dst, a.dst_dtype, (size_t)(y0 + k) * (size_t)a.W + (size_t)x, t);
    }
  }
}

template <int KIND, int NF>
static hipError_t launch_spline_wg_nf(const SplineArgs& a, const MapArgs& map, void* dst, hipStream_t stream) {
  const unsigned tiles_x = (unsigned)((a.W + kSwTW - 1) / kSwTW);
  const dim3 grid(a.xcd_remap ? ((tiles_x + 7u) / 8u) * 8u : tiles_x, (unsigned)((a.H + kSwTH - 1) / kSwTH));
#define DCP_SWG(ORD)                                                                                                        \
  if (a.exact_sum) hipLaunchKernelGGL((spline_wg_kernel<KIND, ORD, NF, true>), grid, dim3(256), 0, stream, a, map, dst);    \
  else hipLaunchKernelGGL((spline_wg_kernel<KIND, ORD, NF, false>), grid, dim3(256), 0, stream, a, map, dst)
  switch (a.order) {
    case 2: DCP_SWG(2); break;
    case 3: DCP_SWG(3); break;
    case 4: DCP_SWG(4); break;
    default: DCP_SWG(5); break;
  }
#undef DCP_SWG
  return hipGetLastError();
}

template <int KIND>
static hipError_t launch_spline_wg(const SplineArgs& a_in, const MapArgs& map_in, void* dst, hipStream_t stream) {
  SplineArgs a = a_in;
  a.xcd_remap = g_spline_xcd;
  if constexpr (KIND == kPersp) {
    return launch_spline_wg_nf<KIND, 0>(a, map_in, dst, stream);          // (no polynomial)
  } else {
    if (map_in.nfact > 5) return launch_spline_wg_nf<KIND, -1>(a, map_in, dst, stream);
    MapArgs map = map_in;                                                 // <= 5 coefficients: the NF = 5 instantiation, zero-padded
    for (int i = map.nfact < 0 ? 0 : map.nfact; i < 5; ++i) map.fact[i] = 0.0;
    map.nfact = 5;
    return launch_spline_wg_nf<KIND, 5>(a, map, dst, stream);
  }
}

template <int MAPKIND>
static hipError_t launch_remap_order(const SplineArgs& a, const MapArgs& map, const CoordArgs& ca, void* dst,
                                     int64_t total, hipStream_t stream) {
  const dim3 grid = MAPKIND == 2 ? dim3((unsigned)((total + kSplBlock - 1) / kSplBlock))
                                 : dim3((unsigned)((a.W + kSplBlock - 1) / kSplBlock), (unsigned)(a.H < 65535 ? a.H : 65535),
                                        (unsigned)((a.H + 65534) / 65535));
  switch (a.order) {
    case 2: hipLaunchKernelGGL((spline_remap_kernel<MAPKIND, 2>), grid, dim3(kSplBlock), 0, stream, a, map, ca, dst); break;
    case 3: hipLaunchKernelGGL((spline_remap_kernel<MAPKIND, 3>), grid, dim3(kSplBlock), 0, stream, a, map, ca, dst); break;
    case 4: hipLaunchKernelGGL((spline_remap_kernel<MAPKIND, 4>), grid, dim3(kSplBlock), 0, stream, a, map, ca, dst); break;
    default: hipLaunchKernelGGL((spline_remap_kernel<MAPKIND, 5>), grid, dim3(kSplBlock), 0, stream, a, map, ca, dst); break;
  }
  return hipGetLastError();
}

DCP_DEFINE_BOUNDS_READER(read_bounds_spline)

hipError_t launch_spline(const SplineArgs& a, int map_kind, const MapArgs& map, const CoordArgs& ca, void* dst,
                         hipStream_t stream) {
  // prefilter: axis 0 on the (Hp x Wp) plane, transpose, axis 1 as axis 0 of the (Wp x Hp) plane,
  // transpose back.  Two planes ping-pong: a.coef (A) and a.scratch (B); the result ends in A.
  double lam = 1.0;
  for (int p = 0; p < a.npoles; ++p) lam *= (1.0 - a.poles[p]) * (1.0 - 1.0 / a.poles[p]);
  const bool direct = a.src_dtype == kF32 && a.pad == 0;
  bool expanded = false;
  auto expand = [&]() {                     // the image as a float64 plane (padded modes, element types other than float32), once
    if (direct || expanded) return;
    const int64_t plane = (int64_t)a.Hp * a.Wp;
    hipLaunchKernelGGL(spline_expand_kernel, dim3((unsigned)((plane + kSplBlock - 1) / kSplBlock)), dim3(kSplBlock), 0,
                       stream, a);
    expanded = true;
  };
  // the one-pass tiles when both axes qualify (reflect / mirror kind, z^n underflowed to zero for every pole)
  bool tiled = (a.filter_kind == kSplReflect || a.filter_kind == kSplMirror) && a.Hp >= kTfSamples && a.Wp >= kTfSamples && g_spline_tiled;
  int halo = 0, hp[2] = {0, 0};
  for (int p = 0; p < a.npoles; ++p) {
    tiled = tiled && a.zpow[0][p] == 0.0 && a.zpow[1][p] == 0.0;
    hp[p] = (int)ceil(-64.0 * 0.6931471805599453 / log(fabs(a.poles[p])));      // |z|^h <= 2^-64
    halo += hp[p];
  }
  const int samples = kTfSamples;
  bool col_stream = false, col_lds = false, row_scan = false, row_lds = false, fused2d = false, two_pass = false;
  if (tiled && samples - 2 * halo >= 32) {
    TileFilter f;
    f.kind = a.filter_kind;
    f.npoles = a.npoles;
    f.z[0] = a.poles[0];
    f.z[1] = a.npoles > 1 ? a.poles[1] : 0.0;
    f.lam = lam;
    f.halo = halo;
    f.hp[0] = hp[0];
    f.hp[1] = hp[1];
    f.core = samples - 2 * halo;
    auto launch = [&](auto axis, auto in_f32, auto smp, const dim3& grid) {
      constexpr int AX = decltype(axis)::value, SM = decltype(smp)::value;
      constexpr bool F32 = decltype(in_f32)::value;
      const size_t lds = AX == 0 ? (size_t)SM * kTfLines * sizeof(double) : (size_t)kTfLines * (SM + 1) * sizeof(double);
      int ncu = 0, dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 1) ncu = 256;
      // more than 64 KB of dynamic LDS has to be allowed once per kernel and device (one flag array per instantiation of the
      // lambda; setting it twice from two threads is harmless)
      static bool attr_set[64] = {};
      if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        (void)hipFuncSetAttribute((const void*)spline_tile_filter_kernel<AX, F32, SM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
      }
      const unsigned ntiles = grid.x * grid.y;
      hipLaunchKernelGGL((spline_tile_filter_kernel<AX, F32, SM>), dim3(ntiles < (unsigned)ncu ? ntiles : (unsigned)ncu), dim3(kTfBlock), lds, stream, f);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using SL = std::integral_constant<int, kTfSamples>;
    // axis 0: lines are the columns of the (Hp x Wp) plane; source image (or its expanded copy A) -> B
    f.n = a.Hp;
    f.nlines = a.Wp;
    f.out = a.scratch;
    f.out_ls = 1;
    f.out_ss = a.Wp;
    dim3 grid((unsigned)((f.nlines + kTfLines - 1) / kTfLines), (unsigned)((f.n + f.core - 1) / f.core));
    // (columns of an interleaved image -- a channel of an (H, W, C) array -- are read in place: 4-byte loads at the pixel pitch)
    const double ext_src = ((double)(a.H - 1) * (double)a.src_stride + (double)(a.W - 1) * (double)a.src_cstride + 1.0) * (double)elem_size(a.src_dtype);
    const double ext_plane = (double)a.Hp * (double)a.Wp * 8.0;
    // (the padded modes too: 'nearest' and 'grid-constant' filter the image with 12 samples added per side, which the loads of this
    // kernel produce by clamping / zeroing instead of reading a float64 copy that spline_expand_kernel would have to write first)
    // Element types: float32 and the integer types cameras deliver (uint8, uint16, int16), read in place -- the others take the float64 copy.
    const bool pf2d_type = a.src_dtype == kF32 || a.src_dtype == kU8 || a.src_dtype == kU16 || a.src_dtype == kI16;
    const bool padded_f32 = pf2d_type && a.pad > 0 && (a.mode == kModeNearest || a.mode == kModeGridConstant) && a.H > 0 && a.W > 0;
    const bool pf2d_ok = ((pf2d_type && a.pad == 0) || padded_f32) && g_spline_tiled == 1 && a.src_cstride >= 1 && a.src_stride >= 0 && ext_src < 4294967000.0 &&
                         ext_plane < 4294967000.0;
    // one launch of spline_prefilter2d_kernel<HPV, PADV, STV, RV>: `tf` names the source (pointer, strides), the pole, its gain and the result plane
    auto launch_pf2d = [&](auto hpv, auto padv, auto stv, auto rv, const TileFilter& tf, const Pf2dPad& pd, double src_extent) {
      constexpr int HPV = decltype(hpv)::value, STV = decltype(stv)::value, RV = decltype(rv)::value;
      constexpr bool PADV = decltype(padv)::value;
      const int core = Pf2d<HPV, RV>::CORE;
      const int stripes = (a.Wp + core - 1) / core;
      int ncu = 0, dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 1) ncu = 256;
      // chunks of rows: one resident round of workgroups (DCP_PF2D_WAVES per CU) where the plane is tall enough, never under 64 rows
      int chunk = g_pf2d_chunk;
      if (chunk <= 0) {
        const int want = (DCP_PF2D_WAVES * ncu) / stripes;                       // chunks per stripe that fill the chip once
        chunk = want > 0 ? (a.Hp + want - 1) / want : a.Hp;
      }
      chunk = ((chunk + 31) / 32) * 32;
      if (chunk < 64) chunk = 64;
      const int chunks = (a.Hp + chunk - 1) / chunk;
      const int xcd_order = g_pf2d_xcd;
      const dim3 g5((unsigned)(xcd_order ? ((stripes * chunks + 7) / 8) * 8 : stripes * chunks));
      hipLaunchKernelGGL((spline_prefilter2d_kernel<HPV, PADV, STV, RV>), g5, dim3(256), 0, stream, tf, pd, (uint32_t)src_extent, (uint32_t)ext_plane, chunk, stripes,
                         chunks, xcd_order);
    };
    // ... the same with the source element type and the padding resolved at run time
    auto launch_pf2d_src = [&](auto hpv, auto rv, const TileFilter& tf, const Pf2dPad& pd) {
      using T = std::true_type;
      using Fa = std::false_type;
#define DCP_PF2D_T(STV)                                                                                       \
  if (padded_f32) launch_pf2d(hpv, T{}, std::integral_constant<int, STV>{}, rv, tf, pd, ext_src);             \
  else launch_pf2d(hpv, Fa{}, std::integral_constant<int, STV>{}, rv, tf, pd, ext_src)
      switch (a.src_dtype) {
        case kU8: DCP_PF2D_T(kU8); break;
        case kU16: DCP_PF2D_T(kU16); break;
        case kI16: DCP_PF2D_T(kI16); break;
        default: DCP_PF2D_T(kF32); break;
      }
#undef DCP_PF2D_T
    };
    Pf2dPad pd;
    pd.pad = a.pad;
    pd.Hs = a.H;
    pd.Ws = a.W;
    pd.zero_outside = a.mode == kModeGridConstant ? 1 : 0;
    using I32c = std::integral_constant<int, 32>;
    if (pf2d_ok && a.npoles == 1 && (hp[0] == 26 || hp[0] == 34)) {
      // float32 source, one pole: both axes in one pass, the column-filtered plane stays in LDS (spline_prefilter2d_kernel)
      f.in = a.src;
      f.in_ls = a.src_cstride;
      f.in_ss = a.src_stride;
      f.out = a.coef;
      f.out_ls = a.Wp;
      f.out_ss = 1;
      if (hp[0] == 34) launch_pf2d_src(std::integral_constant<int, 34>{}, I32c{}, f, pd);
      else launch_pf2d_src(std::integral_constant<int, 26>{}, I32c{}, f, pd);
      fused2d = true;
    } else if (pf2d_ok && a.npoles == 2 && (hp[0] == 44 || hp[0] == 53) && hp[1] <= 16 && g_pf2d_two_pole) {
      // two poles (orders 4, 5): one pass of the same kernel per pole -- image -> scratch plane (pole 1, 16-row steps), scratch plane ->
      // coefficient plane (pole 2) -- see Pf2d
      TileFilter p1 = f;
      p1.npoles = 1;
      p1.z[0] = a.poles[0];
      p1.lam = (1.0 - a.poles[0]) * (1.0 - 1.0 / a.poles[0]);
      p1.in = a.src;
      p1.in_ls = a.src_cstride;
      p1.in_ss = a.src_stride;
      p1.out = a.scratch;
      p1.out_ls = a.Wp;
      p1.out_ss = 1;
      // (order 5's first pole restarts 48 samples out, not 53: |z|^48 = 2^-58.4 of the signal, 2 % of ONE float64 rounding error where
      // every sample of the plane carries dozens -- the five samples keep the 80-row column window inside 256 VGPRs without scratch)
      if (hp[0] == 44) launch_pf2d_src(std::integral_constant<int, 44>{}, I32c{}, p1, pd);
      else launch_pf2d_src(std::integral_constant<int, 48>{}, I32c{}, p1, pd);
      TileFilter p2 = p1;
      p2.z[0] = a.poles[1];
      p2.lam = (1.0 - a.poles[1]) * (1.0 - 1.0 / a.poles[1]);
      p2.in = a.scratch;
      p2.in_ls = 1;
      p2.in_ss = a.Wp;
      p2.out = a.coef;
      Pf2dPad none = pd;
      none.pad = 0;
      none.Hs = a.Hp;
      none.Ws = a.Wp;
      none.zero_outside = 0;
      launch_pf2d(std::integral_constant<int, 16>{}, std::false_type{}, std::integral_constant<int, kF64Src>{}, I32c{}, p2, none, ext_plane);
      fused2d = true;
      two_pass = true;
    } else if (direct && a.npoles == 1 && g_spline_tiled != 2 && g_spline_tiled != 4 && (hp[0] == 26 || hp[0] == 34)) {
      // float32 source, one pole (orders 2 and 3): the register-streaming column pass, no LDS
      f.in = a.src;
      f.in_ls = a.src_cstride;
      f.in_ss = a.src_stride;
      const dim3 g2((unsigned)((f.nlines + 63) / 64), (unsigned)((f.n + 4 * kCsK * kCsSeg - 1) / (4 * kCsK * kCsSeg)));
      // (the LDS-staged form needs unit column stride, 16-byte loads inside a 32-bit extent; g_spline_tiled = 3 selects the
      // unstaged stream kernel for A/B runs)
      const double ext = ((double)(a.H - 1) * (double)a.src_stride + (double)a.W) * 4.0;
      if (a.src_cstride == 1 && ext < 4294967000.0 && g_spline_tiled != 3) {
        const dim3 g3((unsigned)((f.nlines + 63) / 64), (unsigned)((f.n + 127) / 128));
        if (hp[0] == 26) hipLaunchKernelGGL((spline_col_lds_kernel<26>), g3, dim3(256), 0, stream, f, (uint32_t)ext);
        else hipLaunchKernelGGL((spline_col_lds_kernel<34>), g3, dim3(256), 0, stream, f, (uint32_t)ext);
        col_lds = true;
      } else if (hp[0] == 26) {
        hipLaunchKernelGGL((spline_col_stream_kernel<26>), g2, dim3(256), 0, stream, f);
      } else {
        hipLaunchKernelGGL((spline_col_stream_kernel<34>), g2, dim3(256), 0, stream, f);
      }
      col_stream = true;
    } else if (direct) {
      f.in = a.src;
      f.in_ls = a.src_cstride;
      f.in_ss = a.src_stride;
      launch(I0{}, std::true_type{}, SL{}, grid);
    } else {
      expand();
      f.in = a.coef;
      f.in_ls = 1;
      f.in_ss = a.Wp;
      launch(I0{}, std::false_type{}, SL{}, grid);
    }
    if (!fused2d) {
    // axis 1: lines are the rows; B -> A
    f.n = a.Wp;
    f.nlines = a.Hp;
    f.in = a.scratch;
    f.in_ls = a.Wp;
    f.in_ss = 1;
    f.out = a.coef;
    f.out_ls = a.Wp;
    f.out_ss = 1;
    grid = dim3((unsigned)((f.nlines + kTfLines - 1) / kTfLines), (unsigned)((f.n + f.core - 1) / f.core));
    const double ext1 = (double)a.Hp * (double)a.Wp * 8.0;
    if (a.npoles == 1 && (g_spline_tiled == 1 || g_spline_tiled == 5 || g_spline_tiled == 6) && ext1 < 4294967000.0 && (hp[0] == 26 || hp[0] == 34)) {
      // one pole: staged once at an odd pitch, recursions in registers (spline_row_lds_kernel)
      const int core = hp[0] == 34 ? RowLds<34>::CORE : RowLds<26>::CORE;
      const dim3 g4((unsigned)((f.nlines + kRlRows - 1) / kRlRows), (unsigned)((f.n + core - 1) / core));
      if (hp[0] == 34) hipLaunchKernelGGL((spline_row_lds_kernel<34>), g4, dim3(256), 0, stream, f, (uint32_t)ext1);
      else hipLaunchKernelGGL((spline_row_lds_kernel<26>), g4, dim3(256), 0, stream, f, (uint32_t)ext1);
      row_lds = true;
    } else if (a.npoles == 1 && g_spline_tiled == 4 && f.nlines <= 4 * 65535) {
      // one pole, option spline_tiled = 4 only: the cross-lane scan along the rows (no LDS) -- measured at 84 us per 4096^2 plane
      // against the tile kernel's 72 (VALU-bound: ~50 instructions per 64 samples and scan step), kept for A/B runs
      const int nbt = (f.n + 63) / 64;
      hipLaunchKernelGGL(spline_row_scan_kernel, dim3((unsigned)((nbt + kRsNB - 1) / kRsNB), (unsigned)((f.nlines + 3) / 4)), dim3(256), 0, stream, f);
      row_scan = true;
    } else {
      launch(I1{}, std::false_type{}, SL{}, grid);
    }
    }
  } else {
    tiled = false;
  }
  auto filter_axis = [&](double* A, double* B, int n, int nlines, int axis, bool from_src) {
    const dim3 grid((unsigned)((nlines + kSplBlock - 1) / kSplBlock), (unsigned)((n + kChunk - 1) / kChunk));
    for (int p = 0; p < a.npoles; ++p) {
      FilterPass f;
      f.n = n;
      f.nlines = nlines;
      f.kind = a.filter_kind;
      f.z = a.poles[p];
      f.zpow = a.zpow[axis][p];
      f.lam = p == 0 ? lam : 1.0;
      f.in = A;
      f.out = B;
      if (from_src && p == 0) hipLaunchKernelGGL(spline_causal_kernel<true>, grid, dim3(kSplBlock), 0, stream, f, a);
      else hipLaunchKernelGGL(spline_causal_kernel<false>, grid, dim3(kSplBlock), 0, stream, f, a);
      f.in = B;
      f.out = A;
      f.lam = 1.0;
      hipLaunchKernelGGL(spline_anticausal_kernel, grid, dim3(kSplBlock), 0, stream, f);
    }
  };
  if (!tiled) {
    expand();
    filter_axis(a.coef, a.scratch, a.Hp, a.Wp, 0, direct);   // float32: the first pass reads the image itself
    hipLaunchKernelGGL(spline_transpose_kernel, dim3((unsigned)((a.Wp + 31) / 32), (unsigned)((a.Hp + 31) / 32)),
                       dim3(kSplBlock), 0, stream, (const double*)a.coef, a.scratch, a.Hp, a.Wp);
    filter_axis(a.scratch, a.coef, a.Wp, a.Hp, 1, false);
    hipLaunchKernelGGL(spline_transpose_kernel, dim3((unsigned)((a.Hp + 31) / 32), (unsigned)((a.Wp + 31) / 32)),
                       dim3(kSplBlock), 0, stream, (const double*)a.scratch, a.coef, a.Wp, a.Hp);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const int64_t total = map_kind == 2 ? ca.npts : (int64_t)a.H * a.W;
  if (total == 0) return hipSuccess;
  // certified radial / perspective maps on frames of at least one workgroup tile: the taps out of LDS
  const bool wg = (map_kind == 0 || map_kind == 1) && map.tile_dev_ok >= 2 && g_spline_wg && a.H >= kSwTH && a.W >= kSwTW &&
                  (int64_t)a.Hp * a.Wp * 8 < ((int64_t)1 << 32) && a.Hp < 65535 * kSwTH;
  {
    char name[160];
    const char* colk = col_lds ? "spline_col_lds_kernel" : col_stream ? "spline_col_stream_kernel" : "spline_tile_filter_kernel";
    const char* rowk = row_scan ? "spline_row_scan_kernel" : row_lds ? "spline_row_lds_kernel" : "spline_tile_filter_kernel";
    if (fused2d) snprintf(name, sizeof(name), "spline_prefilter2d_kernel%s + %s<order=%d>", two_pass ? " x 2" : "", wg ? "spline_wg_kernel" : "spline_remap_kernel", a.order);
    else if (!tiled) snprintf(name, sizeof(name), "spline_causal / anticausal / transpose kernels + %s<order=%d>", wg ? "spline_wg_kernel" : "spline_remap_kernel", a.order);
    else if (!col_stream && !row_scan && !row_lds) snprintf(name, sizeof(name), "spline_tile_filter_kernel x 2 + %s<order=%d>", wg ? "spline_wg_kernel" : "spline_remap_kernel", a.order);
    else snprintf(name, sizeof(name), "%s + %s + %s<order=%d>", colk, rowk, wg ? "spline_wg_kernel" : "spline_remap_kernel", a.order);
    set_last_kernel_name(name);
  }
  if (wg) {
    if (map_kind == 0) return launch_spline_wg<kRadial>(a, map, dst, stream);
    return launch_spline_wg<kPersp>(a, map, dst, stream);
  }
  if (map_kind == 0) return launch_remap_order<0>(a, map, ca, dst, total, stream);
  if (map_kind == 1) return launch_remap_order<1>(a, map, ca, dst, total, stream);
  if (map_kind == 3) return launch_remap_order<3>(a, map, ca, dst, total, stream);
  return launch_remap_order<2>(a, map, ca, dst, total, stream);
}

}  // namespace dcp
