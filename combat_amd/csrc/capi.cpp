// Library identity for the combat_hip C ABI (include/combat_hip.h).
#include "combat_hip.h"

// 3: combat_pack_desc.row_scale, fused normalisation entries, combat_relu_mask, tile ids 10-15, conv / wgrad workspaces
#define COMBAT_ABI_VERSION 12   // 12: combat_comm_* / combat_allreduce (RCCL for non-PyTorch hosts); 11: combat_head_fwd_bwd, combat_head_bwd_weights; 10: combat_wgrad_args.reduce_first (a weight gradient folds its predecessor's slabs first; deterministic reductions); 9: combat_conv_args.pro_act_dst (in-LDS prologue of the DMA-staged 3x3 kernel); 8: combat_conv_args.src2 (shortcut input gradient as second reduction source), tile 18; 7: COMBAT_STATS_PER_WORKGROUP; 6: combat_plan_* (C-side replay), tile 17; 4: WaNet entry points, tile 16, large-image augment / DCT; 5: combat_conv_gemm_pair, combat_log_terms

extern "C" const char *combat_version(void) { return "combat_hip gfx950 abi12"; }
extern "C" int combat_abi_version(void) { return COMBAT_ABI_VERSION; }
