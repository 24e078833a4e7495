// Fused epilogue of the DMA-staged convolution kernels (conv3x3_dma.hip, conv_gather_dma.hip): device
// code only (it uses the buffer-resource type, which the host pass does not know; include it from
// device function templates).
//
// A wave owns 32 pixels x BN channels (TileCfg<128, BN, 4>).  It transposes its accumulators through a
// private LDS image (no workgroup barrier) so that every global access is 16 bytes of 8 consecutive
// channels and NC lanes cover one pixel's contiguous bytes.  Residual / mask operands and the
// (scale, shift) table pair come through buffer loads (rows outside the tensor and absent tensors use
// an out-of-range offset / an empty resource: loads give zeros, stores are dropped), issued by
// epi_fetch() some steps before the main loop ends so that their latency is covered by the remaining
// MFMA steps; the fixed instruction count (EpiCfg::NPF) lets the main loop's counted vmcnt waits
// include them.  Semantics = conv_common.hpp's conv_epilogue (bias, add_pre, mask, add_post,
// statistics) plus the activated second output.
#pragma once
#include "conv_common.hpp"

constexpr unsigned kDmaOob = 0x40000000u;  // buffer offset beyond every tensor on this path: reads as zeros

typedef __attribute__((address_space(3))) float lds_f32_t;   // (an LDS pointer the compiler KNOWS is one: ds_read, not flat_load)
__device__ __forceinline__ void load8f_lds(const lds_f32_t *p, float (&v)[8]) {
    typedef __attribute__((address_space(3))) f32x4_t lds_f32x4_t;
    const f32x4_t a = *reinterpret_cast<const lds_f32x4_t *>(p);
    const f32x4_t b = *reinterpret_cast<const lds_f32x4_t *>(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}

template <typename T>
struct EpiCfg {
    static constexpr int NC = T::NC;              // 16-byte chunks per dst row of the tile
    static constexpr int EQ = 32 * NC / 64;       // (row, chunk) items per lane
    static constexpr int NPF = 3 * EQ + 4;        // operand + table fetches of epi_fetch()
    static constexpr int EPW = 32 * T::EPS > 64 * 20 ? 32 * T::EPS : 64 * 20;   // floats per wave: tile image / statistics scratch
    static constexpr int LDS_BYTES = 4 * 4 * EPW;
};

template <typename T>
struct EpiRegs {   // plain data only
    unsigned evoff[EpiCfg<T>::EQ];                // byte offset of this lane's items in the dst-shaped tensors (or kDmaOob)
    u32x4_t e_pre[EpiCfg<T>::EQ], e_x[EpiCfg<T>::EQ], e_post[EpiCfg<T>::EQ];
    f32x4_t t_sc[2], t_sh[2];                     // mask tables (input-gradient passes) or activation-output tables (forward)
};

// evoff[q] = row_off(tile row) * 2 + channel bytes; row_off(row) = element index of the row's dst pixel * K, or -1
template <typename T, typename RowOff>
__device__ __forceinline__ void epi_init(EpiRegs<T> &r, int lane, int wid, int n0, RowOff row_off) {
    constexpr int NC = EpiCfg<T>::NC;
    const int ec = lane % NC;
#pragma unroll
    for (int q = 0; q < EpiCfg<T>::EQ; ++q) {
        const long e = row_off(wid * 32 + (q * 64 + lane) / NC);
        r.evoff[q] = e >= 0 ? (unsigned)((e + n0 + ec * 8) * 2) : kDmaOob;
    }
}

template <typename T>
__device__ __forceinline__ void epi_fetch(EpiRegs<T> &r, const combat_conv_args &a, unsigned dst_bytes, int lane, int n0) {
    const unsigned tab_bytes = (unsigned)a.K * 4u;
    const __amdgpu_buffer_rsrc_t r_pre = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.add_pre), 0, a.add_pre ? dst_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.mask_x), 0, a.mask_x ? dst_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_post = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.add_post), 0, a.add_post ? dst_bytes : 0u, 0x00020000);
    // one (scale, shift) table pair travels with the operand prefetch: the mask tables or the
    // activation-output tables -- a launch never has both (the kernels' applicable())
    const float *tab_sc = a.mask_x ? a.mask_scale : a.act_scale, *tab_sh = a.mask_x ? a.mask_shift : a.act_shift;
    const __amdgpu_buffer_rsrc_t r_sc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tab_sc), 0, tab_sc ? tab_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_sh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tab_sh), 0, tab_sh ? tab_bytes : 0u, 0x00020000);
#pragma unroll
    for (int q = 0; q < EpiCfg<T>::EQ; ++q) {
        r.e_pre[q] = __builtin_amdgcn_raw_buffer_load_b128(r_pre, r.evoff[q], 0, 0);
        r.e_x[q] = __builtin_amdgcn_raw_buffer_load_b128(r_x, r.evoff[q], 0, 0);
        r.e_post[q] = __builtin_amdgcn_raw_buffer_load_b128(r_post, r.evoff[q], 0, 0);
    }
    const int tb = (n0 + (lane % EpiCfg<T>::NC) * 8) * 4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        r.t_sc[h] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r_sc, tb + 16 * h, 0, 0));
        r.t_sh[h] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r_sh, tb + 16 * h, 0, 0));
    }
}

// Make the compiler wait for epi_fetch()'s loads HERE (an empty asm that "uses" every fetched register).  Its wait
// counts track only VGPR-destination loads: a use of these registers placed after younger LDS-DMA instructions gets
// a vmcnt(0..3) that also waits for those DMAs.  Called before such DMAs are issued, the wait costs nothing (the
// fetches are a main loop old) and later uses need none.
template <typename T>
__device__ __forceinline__ void epi_touch(EpiRegs<T> &r) {
#pragma unroll
    for (int q = 0; q < EpiCfg<T>::EQ; ++q) asm volatile("" : "+v"(r.e_pre[q]), "+v"(r.e_x[q]), "+v"(r.e_post[q]));
    asm volatile("" : "+v"(r.t_sc[0]), "+v"(r.t_sc[1]), "+v"(r.t_sh[0]), "+v"(r.t_sh[1]));
}

// ---- epilogue flavours.  epi_finish's feature tests (which operand tensors exist, which mask form, which
// statistics) are uniform per launch.  As run-time flags they cost every launch the VALU work of every
// feature -- 1157 vector instructions per wave for a 32 x 64 tile, 4.8 k cycles of a 15 k-cycle workgroup
// (in-kernel stamps), more than the nine MFMA taps of a 64-channel layer -- because a select is cheaper to
// emit than a branch.  The launches of the step use a handful of combinations; those are compiled as
// specialised bodies (features as template constants: absent ones vanish) behind ONE uniform switch on
// `flavour`, anything else takes the generic body (FL < 0: all features tested at run time).
enum : int {
    EF_PRE = 1,          // add_pre tensor
    EF_POST = 2,         // add_post tensor
    EF_MASK = 4,         // mask_x tensor
    EF_MASK_TAB = 8,     // kept-test on fma(x, scale, shift) (else on x itself: activated tensors / table-free masks)
    EF_MUL_SCALE = 16,   // kept elements are multiplied by the table's scale (BatchNorm backward)
    EF_PER_IMAGE = 32,   // mask / statistics tables per (image, channel): InstanceNorm
    EF_ACT = 64,         // activated second output
    EF_STATS1 = 128,     // sums / second moments of the stored values
    EF_STATS2 = 256,     // sums / dz * xhat products (norm backward)
    EF_BIAS = 512,
    EF_NODST = 1024,     // no raw output tensor
};

// feature bits of a launch (host and device)
__host__ __device__ inline int epi_flags_of(const combat_conv_args &a) {
    int f = 0;
    if (a.add_pre) f |= EF_PRE;
    if (a.add_post) f |= EF_POST;
    if (a.mask_x) {
        f |= EF_MASK;
        if (a.mask_scale && !a.mask_activated) f |= EF_MASK_TAB;
        if (a.mask_mul_scale && a.mask_scale) f |= EF_MUL_SCALE;
        if (a.mask_group_stride) f |= EF_PER_IMAGE;
    }
    if (a.act_dst) f |= EF_ACT;
    if ((a.stats_kind & 3) == 1) f |= EF_STATS1;   // (bit 2 of stats_kind = one row per workgroup: a run-time matter)
    if ((a.stats_kind & 3) == 2) f |= EF_STATS2;
    if (a.bias) f |= EF_BIAS;
    if (!a.dst) f |= EF_NODST;
    return f;
}

// The specialised combinations (what the alternated step launches most, engine.py).  Only the KIND of epilogue is
// a compile-time constant; which residual / bias operands exist and whether the raw tensor is stored stay uniform
// run-time branches inside a body (kEpiRuntime bits): every extra body raises the kernel's register allocation
// (the allocator sees one function: 180 registers with the generic body alone, 223 with three more, spills with 17).
constexpr int kEpiRuntime = EF_PRE | EF_POST | EF_BIAS | EF_NODST;
constexpr int kEpiFlavours[] = {
    0,                                        // 0 plain store (+ residuals)
    EF_ACT,                                   // 1 eval forward: (raw +) activated output
    EF_STATS1,                                // 2 train forward: statistics of the stored values
    EF_MASK | EF_MUL_SCALE,                   // 3 eval backward: activated mask x BatchNorm scale
    EF_MASK | EF_MASK_TAB | EF_STATS2,        // 4 train backward (BatchNorm): mask from the raw tensor + norm-backward sums
};
constexpr int kNumEpiFlavours = sizeof(kEpiFlavours) / sizeof(int);

// index into kEpiFlavours, or -1 (generic body)
inline int epi_flavour_of(const combat_conv_args &a) {
    const int f = epi_flags_of(a) & ~kEpiRuntime;
    for (int i = 0; i < kNumEpiFlavours; ++i)
        if (kEpiFlavours[i] == f) return i;
    return -1;
}

// Call after a workgroup barrier that follows the last LDS read of the main loop (the wave images
// overlay the operand images).  stats_row = this wave's row of the statistics array (< 0: the wave lies
// entirely outside the tensor and owns no row); ragged = the wave
// may own rows outside the tensor (their statistics count as zeros).
// FL >= 0: the launch's feature bits, known at compile time; FL < 0: read from the arguments.
struct EpiNoHook {
    __device__ __forceinline__ void operator()() const {}
};

// pre_store(): called once, after the last use of the fetched operands and before the first global store (a
// persistent caller waits for its in-flight LDS-DMA there: behind the stores, the in-order counter could only be
// waited down together with them)
//
// Statistics rows.  stats_kind 1 / 2: one row per wave (stats_row).  With COMBAT_STATS_PER_WORKGROUP set and
// wg_waves > 0 (the caller's promise that all wg_waves waves of the workgroup make this call, with
// stats_row = first row of the workgroup + wid): the waves' sums meet in LDS behind one workgroup barrier and wave 0
// writes row stats_row / wg_waves, the waves added in index order -- a quarter of the rows for the consumer to reduce
// (what decides whether a 32 x 32 or 16 x 16 layer's BatchNorm needs a norm_stage1 launch first).  stat_sink (a
// persistent caller that owns no barrier here): receives this call's sums instead, `tot[v]` = sum of value
// (lane / NC) * VPL + v of channel chunk lane % NC, and writes its one row itself.
template <typename T, int FL, typename PreStore = EpiNoHook, typename StatSink = EpiNoHook>
__device__ __forceinline__ void epi_finish_fl(EpiRegs<T> &er, unsigned char *smem, const f32x4_t (&acc)[T::FN][T::FM],
                                              const combat_conv_args &a, unsigned dst_bytes, int lane, int wid, int n0,
                                              long stats_row, bool ragged, int PQ, const lds_f32_t *lds_tabs = nullptr,
                                              PreStore pre_store = PreStore(), int wg_waves = 0,
                                              StatSink stat_sink = StatSink()) {
    // lds_tabs: the per-channel tables this body would fetch from global memory (bias, xh_rstd, xh_mean: K floats
    // each, in that order; absent ones as zeros), staged in LDS by a persistent caller -- a global load issued here,
    // behind that caller's in-flight LDS-DMA, could only be waited for together with the DMA (in-order vmcnt)
    constexpr int NC = EpiCfg<T>::NC, EQ = EpiCfg<T>::EQ, EPW = EpiCfg<T>::EPW;
    constexpr bool RT = FL < 0;
    const int fl = RT ? epi_flags_of(a) : (FL | (epi_flags_of(a) & kEpiRuntime));
    const bool has_pre = fl & EF_PRE, has_post = fl & EF_POST, has_mask = fl & EF_MASK, mask_tab = fl & EF_MASK_TAB;
    const bool mul_scale = fl & EF_MUL_SCALE, per_image = fl & EF_PER_IMAGE, has_act = fl & EF_ACT;
    const bool kind1 = fl & EF_STATS1, kind2 = fl & EF_STATS2, has_bias = fl & EF_BIAS, has_dst = !(fl & EF_NODST);
    const int fr = lane & 15, fq = lane >> 4, ec = lane % NC, K = a.K;
    const __amdgpu_buffer_rsrc_t r_dst = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, a.dst ? dst_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_act = __builtin_amdgcn_make_buffer_rsrc(a.act_dst, 0, a.act_dst ? dst_bytes : 0u, 0x00020000);
    float *ep = reinterpret_cast<float *>(smem) + wid * EPW;
#pragma unroll
    for (int i = 0; i < T::FN; ++i)
#pragma unroll
        for (int j = 0; j < T::FM; ++j)
            *reinterpret_cast<f32x4_t *>(ep + (j * 16 + fr) * T::EPS + i * 16 + fq * 4) = acc[i][j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // a wave's LDS accesses execute in order

    const int n = n0 + ec * 8;
    // per-channel tables (mask tables in a backward launch, activation-output tables in a forward one: the
    // pair that travelled with the operand prefetch)
    float tsc[8], tsh[8], bias8[8], hrs[8], hmn[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        tsc[e] = er.t_sc[e >> 2][e & 3];
        tsh[e] = er.t_sh[e >> 2][e & 3];
        bias8[e] = hrs[e] = hmn[e] = 0.f;
    }
    if (lds_tabs) {
        if (has_bias) load8f_lds(lds_tabs + n, bias8);
        if (kind2 && !per_image) {
            load8f_lds(lds_tabs + K + n, hrs);
            load8f_lds(lds_tabs + 2 * K + n, hmn);
        }
    } else {
        if (has_bias) load8f(a.bias + n, bias8);   // the rarer tables are not worth registers during the main loop
        if (kind2 && !per_image) {
            load8f(a.xh_rstd + n, hrs);
            load8f(a.xh_mean + n, hmn);
        }
    }
    const float mslope = a.mask_slope, aslope = a.act_slope;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    u32x4_t packed[EQ], packed_act[EQ];
#pragma unroll
    for (int q = 0; q < EQ; ++q) {
        const int r = (q * 64 + lane) / NC;
        float v[8], t[8], xm[8];
        load8f(ep + r * T::EPS + ec * 8, v);
        if (has_pre) {
            unpack8v(er.e_pre[q], t);
            if (has_bias) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bias8[e] + t[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += t[e];
            }
        } else if (has_bias) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bias8[e];
        }
        if (has_mask) {
            unpack8v(er.e_x[q], xm);
            if (per_image && er.evoff[q] != kDmaOob) {   // InstanceNorm: tables of this row's image (L2-resident)
                const long g = (long)((er.evoff[q] >> 1) / (unsigned)K / (unsigned)PQ) * a.mask_group_stride + n;
                if (mask_tab || mul_scale) {
                    load8f(a.mask_scale + g, tsc);
                    load8f(a.mask_shift + g, tsh);
                }
                if (kind2) {
                    load8f(a.xh_rstd + g, hrs);
                    load8f(a.xh_mean + g, hmn);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float test = mask_tab ? fmaf(xm[e], tsc[e], tsh[e]) : xm[e];
                const float kept = mul_scale ? tsc[e] : 1.f;
                v[e] *= test > 0.f ? kept : kept * mslope;
            }
        }
        if (has_post) {
            unpack8v(er.e_post[q], t);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += t[e];
        }
        packed[q] = pack8v(v);
        float vr[8];
        if (has_act || kind1 || kind2) unpack8v(packed[q], vr);   // the stored (rounded) values
        if (has_act) {   // the next layer's (eval BatchNorm + ReLU) prologue, applied to the stored value
            float y[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float qa = fmaf(vr[e], tsc[e], tsh[e]);
                y[e] = qa > 0.f ? qa : qa * aslope;
            }
            packed_act[q] = pack8v(y);
        }
        if (kind1 || kind2) {
            if (ragged && er.evoff[q] == kDmaOob) {   // rows beyond the last image count as zeros
#pragma unroll
                for (int e = 0; e < 8; ++e) vr[e] = 0.f;
            }
            if (kind2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s1[e] += vr[e];
                    s2[e] = fmaf(vr[e], (xm[e] - hmn[e]) * hrs[e], s2[e]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s1[e] += vr[e];
                    s2[e] = fmaf(vr[e], vr[e], s2[e]);
                }
            }
        }
    }
    pre_store();
    if (has_dst) {
#pragma unroll
        for (int q = 0; q < EQ; ++q) __builtin_amdgcn_raw_buffer_store_b128(packed[q], r_dst, er.evoff[q], 0, 0);
    }
    if (has_act) {
#pragma unroll
        for (int q = 0; q < EQ; ++q) __builtin_amdgcn_raw_buffer_store_b128(packed_act[q], r_act, er.evoff[q], 0, 0);
    }
    if (kind1 || kind2) {
        constexpr bool kSink = !__is_same(StatSink, EpiNoHook);
        const bool per_wg = (kSink || wg_waves > 0) && (a.stats_kind & COMBAT_STATS_PER_WORKGROUP);
        constexpr int VPL = NC * 16 / 64;            // values per lane (1, 2 or 4)
        const int vg = (lane / NC) * VPL;
        float tot[VPL];
#pragma unroll
        for (int u = 0; u < VPL; ++u) tot[u] = 0.f;
        if (stats_row >= 0) {
            // the wave's 32 pixels.  Lanes with equal chunk differ by multiples of NC:
            // one pass through the wave's LDS image ([lane][16] partial sums; lane -> (chunk, value pair))
            // instead of log2(64 / NC) dependent cross-lane shuffles of 16 values each.
            float *sp = ep + lane * 20;   // 16 values + pad: 80-byte pitch keeps the 16-byte stores conflict-free
            *reinterpret_cast<f32x4_t *>(sp) = f32x4_t{s1[0], s1[1], s1[2], s1[3]};
            *reinterpret_cast<f32x4_t *>(sp + 4) = f32x4_t{s1[4], s1[5], s1[6], s1[7]};
            *reinterpret_cast<f32x4_t *>(sp + 8) = f32x4_t{s2[0], s2[1], s2[2], s2[3]};
            *reinterpret_cast<f32x4_t *>(sp + 12) = f32x4_t{s2[4], s2[5], s2[6], s2[7]};
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // NC chunks x 16 values = NC * 16 sums; lane handles chunk (lane % NC), values vg .. vg + VPL - 1
#pragma unroll
            for (int src = 0; src < 64 / NC; ++src) {
                const float *rp = ep + (src * NC + ec) * 20 + vg;
#pragma unroll
                for (int u = 0; u < VPL; ++u) tot[u] += rp[u];
            }
        }
        // value index v: 0..7 -> sum, 8..15 -> second moment, of channel n + (v & 7)
        if (!per_wg) {
            if (stats_row >= 0) {
                float *orow = a.stats + ((size_t)stats_row * 2 + (vg >> 3)) * K + n + (vg & 7);
#pragma unroll
                for (int u = 0; u < VPL; ++u) orow[u] = tot[u];
            }
        } else if constexpr (kSink) {
            stat_sink(tot);
        } else {
            float *slot = ep + lane * VPL;   // (this wave's image again: its reads above are complete, LDS runs in order)
#pragma unroll
            for (int u = 0; u < VPL; ++u) slot[u] = tot[u];
            __syncthreads();
            if (wid == 0) {
                float sum[VPL];
#pragma unroll
                for (int u = 0; u < VPL; ++u) sum[u] = 0.f;
                for (int w = 0; w < wg_waves; ++w)
#pragma unroll
                    for (int u = 0; u < VPL; ++u) sum[u] += slot[w * EPW + u];
                float *orow = a.stats + ((size_t)(stats_row / wg_waves) * 2 + (vg >> 3)) * K + n + (vg & 7);
#pragma unroll
                for (int u = 0; u < VPL; ++u) orow[u] = sum[u];
            }
        }
    }
}

// dispatch on the launch's flavour (uniform): one specialised body runs
template <typename T>
__device__ __forceinline__ void epi_finish(EpiRegs<T> &er, unsigned char *smem, const f32x4_t (&acc)[T::FN][T::FM],
                                           const combat_conv_args &a, unsigned dst_bytes, int lane, int wid, int n0,
                                           long stats_row, bool ragged, int PQ, int flavour, int wg_waves = 0) {
#define COMBAT_EPI_CASE(i)                                                                                      \
    case i:                                                                                                     \
        epi_finish_fl<T, kEpiFlavours[i]>(er, smem, acc, a, dst_bytes, lane, wid, n0, stats_row, ragged, PQ,   \
                                          nullptr, EpiNoHook(), wg_waves);                                      \
        break;
    switch (flavour) {
        COMBAT_EPI_CASE(0) COMBAT_EPI_CASE(1) COMBAT_EPI_CASE(2) COMBAT_EPI_CASE(3) COMBAT_EPI_CASE(4)
    default:
        epi_finish_fl<T, -1>(er, smem, acc, a, dst_bytes, lane, wid, n0, stats_row, ragged, PQ, nullptr, EpiNoHook(), wg_waves);
    }
#undef COMBAT_EPI_CASE
    static_assert(kNumEpiFlavours == 5, "one case per flavour");
}
