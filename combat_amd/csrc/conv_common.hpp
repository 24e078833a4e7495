// Shared pieces of the convolution kernels: tile geometry and the fused epilogue.
#pragma once
#include "common.hpp"
#include <type_traits>

// halo kernel (conv3x3.hip) entry points used by the dispatcher in conv_gemm.hip
int conv3x3_pick(const combat_conv_args *a);
int conv3x3_stats_layout(const combat_conv_args *a, int tile, int *rows, int *rows_per_image);
int conv3x3_launch(const combat_conv_args *a, int tile, hipStream_t st);
// DMA-staged variant for prologue-free convolutions (conv3x3_dma.hip), reached through the three above
int conv3x3d_pick(const combat_conv_args *a);
int conv3x3d_stats_layout(const combat_conv_args *a, int tile, int *rows, int *rows_per_image);
int conv3x3d_launch(const combat_conv_args *a, int tile, hipStream_t st);
// DMA-staged gather-GEMM for the remaining prologue-free convolutions (conv_gather_dma.hip)
int conv_gather_dma_bn(const combat_conv_args *a);                 // 0 (not applicable) / 64 / 32 channels per tile
bool conv_gather_dma_parity_split(const combat_conv_args *a);      // statistics rows not image-aligned
bool conv_c8_ok(const combat_conv_args *a);
int conv_c8_launch(const combat_conv_args *a, hipStream_t st);
bool conv_k8_ok(const combat_conv_args *a);      // conv_k8.hip: eight OUTPUT channels
int conv_k8_launch(const combat_conv_args *a, hipStream_t st);
int conv_gather_dma_launch(const combat_conv_args *a, hipStream_t st);
int conv_gather_dma_pair_launch(const combat_conv_args *a, const combat_conv_args *b, hipStream_t st);   // 1: not groupable
long conv_gather_dma_workspace(const combat_conv_args *a);        // scratch bytes a split reduction would use
// halo weight-gradient kernel (conv_wgrad3x3.hip): 0 launched, 1 not applicable, <0 error
int conv_wgrad3x3_try(const combat_wgrad_args *a, hipStream_t st);
int conv_wgrad3x3_dma_try(const combat_wgrad_args *a, hipStream_t st);   // same, for inputs without a prologue
long conv_wgrad3x3_dma_workspace(const combat_wgrad_args *a);             // scratch bytes it can use (0: not applicable)
int conv_wgrad3x3_dma_reduce(const combat_wgrad_args *a, hipStream_t st);  // the reduction a defer_reduce launch left out

template <int BM, int BN, int WGM_ = 0>
struct TileCfg {
    static constexpr int WGM = WGM_ ? WGM_ : ((BN == 16) ? 4 : 2);
    static constexpr int WGN = 4 / WGM;
    static constexpr int WM = BM / WGM;
    static constexpr int WN = BN / WGN;
    static constexpr int FM = WM / 16;
    static constexpr int FN = WN / 16;
    static constexpr int A_ITERS = BM / 32;
    static constexpr int B_ITERS = (BN + 31) / 32;
    static constexpr int EPS = BN + 4;  // fp32 epilogue row stride
    static constexpr int STAGE_BYTES = 2 * (BM + BN) * 128;
    static constexpr int EP_BYTES = BM * EPS * 4;
    static constexpr int SMEM = STAGE_BYTES > EP_BYTES ? STAGE_BYTES : EP_BYTES;
    static constexpr int NC = BN / 8;             // 16-byte chunks per dst row
    static constexpr int RPT = BM * NC / 256;     // dst rows per thread in the epilogue
    static constexpr int SG = BM / 4;             // rows covered by one wave = statistics granule
};

// Accumulators (lane: 4 consecutive dst channels of one pixel) -> fp32 LDS image of the tile ->
// row-major fused tail: every global access is 16 bytes of 8 consecutive channels of one row.
// row_to_m(row) gives the dst pixel index of a tile row or -1; part_row is the statistics row this
// wave writes (or -1).
template <typename T, typename RowMap>
__device__ __forceinline__ void conv_epilogue(unsigned char *smem, const f32x4_t (&acc)[T::FN][T::FM],
                                              const combat_conv_args &a, int n0, int PQ, RowMap row_to_m,
                                              int part_row) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wave_m = wid % T::WGM, wave_n = wid / T::WGM;
    float *ep = reinterpret_cast<float *>(smem);
    {
        const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
        for (int i = 0; i < T::FN; ++i)
#pragma unroll
            for (int j = 0; j < T::FM; ++j) {
                const int n = wave_n * T::WN + i * 16 + fq * 4;
                const int row = wave_m * T::WM + j * 16 + fr;
                *reinterpret_cast<f32x4_t *>(ep + row * T::EPS + n) = acc[i][j];
            }
    }
    __syncthreads();

    const int K = a.K;
    const int cc = tid % T::NC, rgrp = tid / T::NC;
    const int n = n0 + cc * 8;
    const bool n_ok = n < K;
    float bias8[8];
    if (a.bias && n_ok) load8f(a.bias + n, bias8);
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    __bf16 *__restrict__ dst = reinterpret_cast<__bf16 *>(a.dst);
    __bf16 *__restrict__ act_dst = reinterpret_cast<__bf16 *>(a.act_dst);
    float asc[8], ash[8];
    if (act_dst && n_ok) {
        load8f(a.act_scale + n, asc);
        load8f(a.act_shift + n, ash);
    }

    // Per-channel tables are loop invariant when they are not per-image (BatchNorm): hoist them;
    // two instantiations so that no register array is conditionally rewritten (scratch).
    auto tail = [&](auto shared_tag) {
    constexpr bool tab_shared = decltype(shared_tag)::value;
    float msc[8], msh[8], hrs[8], hmn[8];
    if (a.mask_x && n_ok && tab_shared) {
        if (a.mask_scale) {
            load8f(a.mask_scale + n, msc);
            load8f(a.mask_shift + n, msh);
        }
        if ((a.stats_kind & 3) == 2) {
            load8f(a.xh_rstd + n, hrs);
            load8f(a.xh_mean + n, hmn);
        }
    }
    // Rows are processed RB at a time: all their global loads are issued before any is consumed.
    constexpr int RB = T::RPT < 4 ? T::RPT : 4;
#pragma unroll
    for (int pb = 0; pb < T::RPT; pb += RB) {
        int mrow[RB];
        uint4 l_pre[RB], l_x[RB], l_post[RB];
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int row = rgrp * T::RPT + pb + q;
            const int m = n_ok ? row_to_m(row) : -1;
            mrow[q] = m;
            if (m >= 0) {
                const size_t off = (size_t)m * K + n;
                if (a.add_pre) l_pre[q] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const __bf16 *>(a.add_pre) + off);
                if (a.mask_x) l_x[q] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const __bf16 *>(a.mask_x) + off);
                if (a.add_post) l_post[q] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const __bf16 *>(a.add_post) + off);
            }
        }
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int m = mrow[q];
            if (m < 0) continue;
            const int row = rgrp * T::RPT + pb + q;
            float v[8];
            load8f(ep + row * T::EPS + cc * 8, v);
            const size_t off = (size_t)m * K + n;
            if (a.bias) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bias8[e];
            }
            if (a.add_pre) {
                float t[8];
                unpack8(l_pre[q], t);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += t[e];
            }
            float xm[8];
            if (a.mask_x) {
                unpack8(l_x[q], xm);
                if constexpr (!tab_shared) {
                    const int g = (m / PQ) * a.mask_group_stride;
                    if (a.mask_scale) {
                        load8f(a.mask_scale + g + n, msc);
                        load8f(a.mask_shift + g + n, msh);
                    }
                    if ((a.stats_kind & 3) == 2) {
                        load8f(a.xh_rstd + g + n, hrs);
                        load8f(a.xh_mean + g + n, hmn);
                    }
                }
                if (a.mask_scale) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float qv = a.mask_activated ? xm[e] : fmaf(xm[e], msc[e], msh[e]);
                        float d = qv > 0.f ? 1.f : a.mask_slope;
                        if (a.mask_mul_scale) d *= msc[e];
                        v[e] *= d;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= xm[e] > 0.f ? 1.f : a.mask_slope;
                }
            }
            if (a.add_post) {
                float t[8];
                unpack8(l_post[q], t);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += t[e];
            }
            if (a.tanh_out) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
            }
            const uint4 packed = pack8(v);
            if (dst) *reinterpret_cast<uint4 *>(dst + off) = packed;
            if (act_dst) {   // the next layer's (eval BatchNorm + ReLU) prologue, applied to the stored value
                float t[8];
                unpack8(packed, t);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float q = fmaf(t[e], asc[e], ash[e]);
                    t[e] = q > 0.f ? q : q * a.act_slope;
                }
                *reinterpret_cast<uint4 *>(act_dst + off) = pack8(t);
            }
            if (a.stats_kind) {
                float vr[8];
                unpack8(packed, vr);
                if ((a.stats_kind & 3) == 1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        s1[e] += vr[e];
                        s2[e] = fmaf(vr[e], vr[e], s2[e]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        s1[e] += vr[e];
                        s2[e] = fmaf(vr[e], (xm[e] - hmn[e]) * hrs[e], s2[e]);
                    }
                }
            }
        }
    }
    };
    if (a.mask_group_stride == 0) tail(std::true_type{}); else tail(std::false_type{});

    if (a.stats_kind) {
        // lanes of one wave with equal cc differ by multiples of NC
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int o = T::NC; o < 64; o <<= 1) {
                s1[e] += __shfl_xor(s1[e], o);
                s2[e] += __shfl_xor(s2[e], o);
            }
        }
        if (lane < T::NC && n_ok && part_row >= 0) {
            float *o1 = a.stats + ((size_t)part_row * 2) * K + n;
            float *o2 = o1 + K;
            *reinterpret_cast<float4 *>(o1) = make_float4(s1[0], s1[1], s1[2], s1[3]);
            *reinterpret_cast<float4 *>(o1 + 4) = make_float4(s1[4], s1[5], s1[6], s1[7]);
            *reinterpret_cast<float4 *>(o2) = make_float4(s2[0], s2[1], s2[2], s2[3]);
            *reinterpret_cast<float4 *>(o2 + 4) = make_float4(s2[4], s2[5], s2[6], s2[7]);
        }
    }
}
