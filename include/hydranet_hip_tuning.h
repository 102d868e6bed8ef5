/* hydranet_hip_tuning.h -- entry points that exist ONLY in the tuning build of the library (libhydranet_hip_tuning.so, compiled with
 * -DHN_TUNING: HN_TUNING=1 in the environment of multitask_hydranet_amd/_lib.py, used by tools/ only).  They switch heuristics, select
 * template variants that the product never launches (ring / 256 x 256 weight-gradient GEMMs, the 32-channel ring form of the direct 3x3
 * convolution, the register-staged operand-transform GEMM loader) and arm the kernels' ablation bits and s_memtime stamps.
 * libhydranet_hip.so has none of these symbols, no mutable global state, and none of those kernels. */
#pragma once
#include "hydranet_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* tuning hook for tools/: force the cout tile (16/32/64/128) and LDS ring depth (2..4) of later hn_conv_gemm_nt launches; 0 = automatic */
int hn_debug_nt_config(int bc, int r);
/* tools/ A/B hook: 1 = software-pipelined direct 3x3 kernel (one workgroup per CU, weight ring of 3 + 2 patch buffers), 0 (default, faster
 * on every measured shape) = the two-workgroups-per-CU double-buffer form */
int hn_debug_direct_pipe(int on);
int hn_debug_tn_config(int bc, int bn, int splits);
/* tools/ sweep hook: heuristic constants (0 TN split target, 1 TN minimum rows per split, 2 / 3 fused-BatchNorm row-block targets, 4 / 5 pixel
 * thresholds of the 64x64 GEMM tile); the defaults are the shipped heuristics */
int hn_debug_knob(int id, long value);


#ifdef __cplusplus
}
#endif
