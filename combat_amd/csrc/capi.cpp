// Library identity for the combat_hip C ABI (include/combat_hip.h).
#include "combat_hip.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

// 3: combat_pack_desc.row_scale, fused normalisation entries, combat_relu_mask, tile ids 10-15, conv / wgrad workspaces
#define COMBAT_ABI_VERSION 13   // 13: combat_set_deterministic / combat_get_deterministic (every parameter-gradient reduction and the augmentation adjoint without order-dependent fp32 atomics); 12: combat_comm_* / combat_allreduce (RCCL for non-PyTorch hosts); 11: combat_head_fwd_bwd, combat_head_bwd_weights; 10: combat_wgrad_args.reduce_first (a weight gradient folds its predecessor's slabs first; deterministic reductions); 9: combat_conv_args.pro_act_dst (in-LDS prologue of the DMA-staged 3x3 kernel); 8: combat_conv_args.src2 (shortcut input gradient as second reduction source), tile 18; 7: COMBAT_STATS_PER_WORKGROUP; 6: combat_plan_* (C-side replay), tile 17; 4: WaNet entry points, tile 16, large-image augment / DCT; 5: combat_conv_gemm_pair, combat_log_terms

extern "C" const char *combat_version(void) { return "combat_hip gfx950 abi13"; }
extern "C" int combat_abi_version(void) { return COMBAT_ABI_VERSION; }

// Deterministic mode: -1 = not set yet (the COMBAT_DETERMINISTIC environment variable decides at first use)
static std::atomic<int> g_deterministic{-1};
bool combat_deterministic() {
    int v = g_deterministic.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("COMBAT_DETERMINISTIC");
        v = e && e[0] == '1';
        g_deterministic.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}
extern "C" void combat_set_deterministic(int on) { g_deterministic.store(on ? 1 : 0, std::memory_order_relaxed); }
extern "C" int combat_get_deterministic(void) { return combat_deterministic() ? 1 : 0; }

// Scratch owned by the library, one region per stream (launches of one stream run one after the other, so a two-stage
// reduction may leave its partial sums there between its two launches): combat_colsum, combat_head_bwd*.  Allocated at a
// stream's first use (never during a replay's steady state) and kept for the life of the process.
float *combat_stream_scratch(void *stream, size_t bytes) {
    constexpr size_t kBytes = 2u << 20;
    if (bytes > kBytes) return nullptr;
    static std::mutex mu;
    static std::unordered_map<void *, float *> regions;
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    void *key = reinterpret_cast<void *>(reinterpret_cast<uintptr_t>(stream) ^ (static_cast<uintptr_t>(dev + 1) << 56));
    auto it = regions.find(key);
    if (it != regions.end()) return it->second;
    void *ptr = nullptr;
    if (hipMalloc(&ptr, kBytes) != hipSuccess) return nullptr;
    regions.emplace(key, static_cast<float *>(ptr));
    return static_cast<float *>(ptr);
}
