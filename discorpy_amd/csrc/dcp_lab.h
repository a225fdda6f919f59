// dcp_lab.h -- the measurement lab's hooks in the kernel sources.  In the product build (the default: `make`) every hook expands to
// nothing, and the shipped translation units carry no experiment code.  `make lab` (-DDCP_LAB, objects under ../lib/lab/) takes
// the hooks' bodies from lab/lab_hooks.h: per-wave phase timestamps of remap_lds_kernel / remap_wg_kernel (tools/trace_k1.py)
// and per-tile ones of spline_tile_filter_kernel (tools/trace_tf.py), per-step ones of spline_prefilter2d_kernel (tools/trace_pf2d.py).  The ablation builds of rounds 1-4 (fill through VGPRs, no
// fill, no wait for the fill, no stores) are not kept in the sources: their results are in profiles/LAB_NOTEBOOK.md, their code at
// the commit named there (tools/variant_from_git.sh builds a library from any revision's kernels).
#pragma once
#ifdef DCP_LAB
#include "lab/lab_hooks.h"
#else
#define DCP_TRACE(slot) do { } while (0)
#define DCP_TRACE_WAVE_BEGIN(id) do { } while (0)
#define DCP_TRACE_WAVE_END() do { } while (0)
#define DCP_LAB_HOST_DEFINITIONS_UNWARP
#define TF_TRACE(slot) do { } while (0)
#define TF_TRACE_R(slot) do { } while (0)
#define PF2D_TRACE(slot) do { } while (0)
#define DCP_LAB_DEFINITIONS_SPLINE
#endif
