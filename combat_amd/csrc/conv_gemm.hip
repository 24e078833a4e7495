// Convolution forward / input-gradient as a gather-GEMM on bf16 MFMA (gfx950).
//
//   dst[m][n] = epilogue( sum_{tap,c} prologue(src[pix(m,tap)][c]) * wpack[n][tap*C + c] )
//
// One 256-thread workgroup (4 waves) owns a BM x BN tile of dst (BM destination pixels, BN
// destination channels) and walks the reduction in steps of 64.  Both operands are staged
// through registers into XOR-swizzled LDS images ([row][64 k] bf16, 128-B rows, slot ^= (row>>1)&7),
// double buffered: the loads of step t+1 are issued before the MFMAs of step t and written to LDS
// after them, one barrier per step.  The gather (zero padding, stride, transposed addressing for
// dgrad) and the BatchNorm/InstanceNorm + ReLU/LeakyReLU prologue are applied on the way from
// registers to LDS, so a normalised activation tensor never exists in HBM.
//
// MFMA operand assignment: A = weights (16 dst channels x 32 k), B = pixels (32 k x 16 pixels),
// so a lane's 4 accumulator registers are 4 consecutive dst channels of one pixel.  The epilogue
// goes through an fp32 LDS image of the tile so that every global access of the fused element-wise
// tail (bias, gradient accumulation, activation mask, residual, tanh, store, norm statistics)
// is a 16-byte access of 8 consecutive channels of one row.
//
// Replaces (see include/combat_hip.h): nn.Conv2d forward and its input gradient at
// classifier_models/preact_resnet.py:21,23,27-29,77, classifier_models/resnet.py:20,22,27-30,72,
// networks/models.py:275-314, defenses/frequency_based/model.py:13-39, together with the
// normalisation/activation/residual element-wise ops around them.
#include "common.hpp"

namespace {

struct ConvParams {
    combat_conv_args a;
    int M;        // N*P*Q
    int PQ;
    int ntaps;    // R*S
    int c_shift;  // log2(C)
    int s_shift;  // log2(stride)
    int nkt;      // reduction steps of 64
    int tiles_m, tiles_n;
};

template <int BM, int BN>
struct TileCfg {
    static constexpr int WGM = (BN == 16) ? 4 : 2;
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

__device__ __forceinline__ int lds_off(int row, int kchunk) { return row * 128 + ((kchunk ^ ((row >> 1) & 7)) << 4); }

template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvParams p) {
    using T = TileCfg<BM, BN>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const combat_conv_args &a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wave_m = wid % T::WGM, wave_n = wid / T::WGM;

    // XCD-aware, bijective remap: blocks b and b+8 share an XCD, so give each XCD a contiguous
    // run of tiles; dst-channel tiles of one pixel tile are neighbours and re-read src through L2.
    int tile_m, tile_n;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        tile_n = swz % p.tiles_n;
        tile_m = swz / p.tiles_n;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int C = a.C, H = a.H, W = a.W;
    const __bf16 *__restrict__ src = reinterpret_cast<const __bf16 *>(a.src);
    const __bf16 *__restrict__ wp = reinterpret_cast<const __bf16 *>(a.wpack);

    // ---- per-thread gather bookkeeping: this thread always stages the same rows / k-chunk
    const int a_chunk = tid & 7, a_row0 = tid >> 3;
    int a_pix[T::A_ITERS], a_by[T::A_ITERS], a_bx[T::A_ITERS], a_g[T::A_ITERS];
#pragma unroll
    for (int i = 0; i < T::A_ITERS; ++i) {
        const int m = m0 + a_row0 + 32 * i;
        if (m < p.M) {
            const int img = m / p.PQ, rem = m - img * p.PQ;
            const int oy = rem / a.Q, ox = rem - oy * a.Q;
            a_pix[i] = img * H * W;
            a_g[i] = img * a.pro_group_stride;
            if (a.mode == 0) {
                a_by[i] = oy * a.stride - a.pad;
                a_bx[i] = ox * a.stride - a.pad;
            } else {
                a_by[i] = oy + a.pad;
                a_bx[i] = ox + a.pad;
            }
        } else {
            a_pix[i] = -1;
            a_by[i] = a_bx[i] = a_g[i] = 0;
        }
    }
    const bool b_active = (BN >= 32) || (tid < BN * 8);
    const __bf16 *b_ptr = wp + (size_t)(n0 + a_row0) * a.kpad + a_chunk * 8;

    uint4 ra[T::A_ITERS], rb[T::B_ITERS];
    unsigned ra_valid = 0;
    float pscale[8], pshift[8];  // BatchNorm-style prologue (group stride 0): prefetched with the tile
    const bool pro_affine = a.pro_scale != nullptr;
    const bool pro_shared = pro_affine && a.pro_group_stride == 0;

    auto load_tile = [&](int kt) {
        const int kbase = kt * 64 + a_chunk * 8;
        const int tap = kbase >> p.c_shift, ci = kbase & (C - 1);
        const int r = (a.S == 3) ? ((tap * 11) >> 5) : tap;
        const int s = tap - r * a.S;
        const bool tap_ok = tap < p.ntaps;
        ra_valid = 0;
#pragma unroll
        for (int i = 0; i < T::A_ITERS; ++i) {
            int iy, ix;
            bool v = tap_ok && a_pix[i] >= 0;
            if (a.mode == 0) {
                iy = a_by[i] + r;
                ix = a_bx[i] + s;
                v = v && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            } else {
                const int ty = a_by[i] - r, tx = a_bx[i] - s;
                v = v && ty >= 0 && tx >= 0 && (((ty | tx) & (a.stride - 1)) == 0);
                iy = ty >> p.s_shift;
                ix = tx >> p.s_shift;
                v = v && iy < H && ix < W;
            }
            uint4 val = make_uint4(0, 0, 0, 0);
            if (v) {
                val = *reinterpret_cast<const uint4 *>(src + ((size_t)(a_pix[i] + iy * W + ix) * C + ci));
                ra_valid |= 1u << i;
            }
            ra[i] = val;
        }
        if (b_active) {
#pragma unroll
            for (int j = 0; j < T::B_ITERS; ++j)
                rb[j] = *reinterpret_cast<const uint4 *>(b_ptr + (size_t)(32 * j) * a.kpad + kt * 64);
        }
        if (pro_shared && tap_ok) {
            load8f(a.pro_scale + ci, pscale);
            load8f(a.pro_shift + ci, pshift);
        }
    };

    auto store_tile = [&](int kt, int buf) {
        unsigned char *al = smem + buf * (BM * 128);
        unsigned char *bl = smem + 2 * (BM * 128) + buf * (BN * 128);
        const int ci = (kt * 64 + a_chunk * 8) & (C - 1);
#pragma unroll
        for (int i = 0; i < T::A_ITERS; ++i) {
            uint4 val = ra[i];
            if ((pro_affine || a.pro_act) && ((ra_valid >> i) & 1u)) {
                float v[8];
                unpack8(val, v);
                if (pro_affine) {
                    if (!pro_shared) {
                        load8f(a.pro_scale + a_g[i] + ci, pscale);
                        load8f(a.pro_shift + a_g[i] + ci, pshift);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], pscale[e], pshift[e]);
                }
                if (a.pro_act) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.pro_slope;
                }
                val = pack8(v);
            }
            const int row = a_row0 + 32 * i;
            *reinterpret_cast<uint4 *>(al + lds_off(row, a_chunk)) = val;
        }
        if (b_active) {
#pragma unroll
            for (int j = 0; j < T::B_ITERS; ++j) {
                const int row = a_row0 + 32 * j;
                *reinterpret_cast<uint4 *>(bl + lds_off(row, a_chunk)) = rb[j];
            }
        }
    };

    f32x4_t acc[T::FN][T::FM];
#pragma unroll
    for (int i = 0; i < T::FN; ++i)
#pragma unroll
        for (int j = 0; j < T::FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        const unsigned char *al = smem + buf * (BM * 128);
        const unsigned char *bl = smem + 2 * (BM * 128) + buf * (BN * 128);
        const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t pix[T::FM], wts[T::FN];
#pragma unroll
            for (int j = 0; j < T::FM; ++j)
                pix[j] = *reinterpret_cast<const bf16x8_t *>(al + lds_off(wave_m * T::WM + j * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int i = 0; i < T::FN; ++i)
                wts[i] = *reinterpret_cast<const bf16x8_t *>(bl + lds_off(wave_n * T::WN + i * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int i = 0; i < T::FN; ++i)
#pragma unroll
                for (int j = 0; j < T::FM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wts[i], pix[j], acc[i][j], 0, 0, 0);
        }
    };

    // ---- main loop
    load_tile(0);
    store_tile(0, 0);
    __syncthreads();
    for (int kt = 0; kt < p.nkt; ++kt) {
        const bool more = kt + 1 < p.nkt;
        if (more) load_tile(kt + 1);
        compute(kt & 1);
        if (more) store_tile(kt + 1, (kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: accumulators -> fp32 LDS image -> row-major fused tail
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

#pragma unroll
    for (int pr = 0; pr < T::RPT; ++pr) {
        const int row = rgrp * T::RPT + pr;
        const int m = m0 + row;
        if (m >= p.M || !n_ok) continue;
        float v[8];
        load8f(ep + row * T::EPS + cc * 8, v);
        const size_t off = (size_t)m * K + n;
        if (a.bias) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bias8[e];
        }
        if (a.add_pre) {
            float t[8];
            unpack8(*reinterpret_cast<const uint4 *>(reinterpret_cast<const __bf16 *>(a.add_pre) + off), t);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += t[e];
        }
        float xm[8];
        int g = 0;
        if (a.mask_x) {
            unpack8(*reinterpret_cast<const uint4 *>(reinterpret_cast<const __bf16 *>(a.mask_x) + off), xm);
            g = (m / p.PQ) * a.mask_group_stride;
            if (a.mask_scale) {
                float sc[8], sh[8];
                load8f(a.mask_scale + g + n, sc);
                load8f(a.mask_shift + g + n, sh);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float q = fmaf(xm[e], sc[e], sh[e]);
                    float d = q > 0.f ? 1.f : a.mask_slope;
                    if (a.mask_mul_scale) d *= sc[e];
                    v[e] *= d;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= xm[e] > 0.f ? 1.f : a.mask_slope;
            }
        }
        if (a.add_post) {
            float t[8];
            unpack8(*reinterpret_cast<const uint4 *>(reinterpret_cast<const __bf16 *>(a.add_post) + off), t);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += t[e];
        }
        if (a.tanh_out) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
        }
        const uint4 packed = pack8(v);
        *reinterpret_cast<uint4 *>(dst + off) = packed;
        if (a.stats_kind) {
            float vr[8];
            unpack8(packed, vr);
            if (a.stats_kind == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s1[e] += vr[e];
                    s2[e] = fmaf(vr[e], vr[e], s2[e]);
                }
            } else {
                float hr[8], hm[8];
                load8f(a.xh_rstd + g + n, hr);
                load8f(a.xh_mean + g + n, hm);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s1[e] += vr[e];
                    s2[e] = fmaf(vr[e], (xm[e] - hm[e]) * hr[e], s2[e]);
                }
            }
        }
    }
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
        const int gi = m0 / T::SG + wid;
        if (lane < T::NC && n_ok && gi * T::SG < p.M) {
            float *o1 = a.stats + ((size_t)gi * 2) * K + n;
            float *o2 = o1 + K;
            *reinterpret_cast<float4 *>(o1) = make_float4(s1[0], s1[1], s1[2], s1[3]);
            *reinterpret_cast<float4 *>(o1 + 4) = make_float4(s1[4], s1[5], s1[6], s1[7]);
            *reinterpret_cast<float4 *>(o2) = make_float4(s2[0], s2[1], s2[2], s2[3]);
            *reinterpret_cast<float4 *>(o2 + 4) = make_float4(s2[4], s2[5], s2[6], s2[7]);
        }
    }
}

template <int BM, int BN>
int launch(const ConvParams &p, hipStream_t st) {
    using T = TileCfg<BM, BN>;
    static bool attr_set = false;
    auto kern = conv_gemm_kernel<BM, BN>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                T::SMEM) != hipSuccess)
            return COMBAT_ELAUNCH;
        attr_set = true;
    }
    ConvParams q = p;
    q.tiles_m = (p.M + BM - 1) / BM;
    q.tiles_n = (p.a.K + BN - 1) / BN;
    if (q.tiles_n * BN > p.a.rows_pad) return COMBAT_EINVAL;
    hipLaunchKernelGGL(kern, dim3(q.tiles_m * q.tiles_n), dim3(256), T::SMEM, st, q);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

int pick_tile(const combat_conv_args *a) {
    if (a->tile) return a->tile;
    const long M = (long)a->N * a->P * a->Q;
    if (a->K <= 16) return COMBAT_TILE_128x16;
    if (a->K % 128 != 0) return (M / 128) * ((a->K + 63) / 64) >= 256 ? COMBAT_TILE_128x64 : COMBAT_TILE_64x64;
    if ((M / 128) * (a->K / 128) >= 256) return COMBAT_TILE_128x128;
    if ((M / 64) * (a->K / 128) >= 256) return COMBAT_TILE_64x128;
    return COMBAT_TILE_64x64;
}

}  // namespace

extern "C" int combat_conv_pick_tile(const combat_conv_args *a) { return a ? pick_tile(a) : COMBAT_EINVAL; }

extern "C" int combat_conv_stats_granule(int tile) {
    switch (tile) {
        case COMBAT_TILE_128x128:
        case COMBAT_TILE_128x64:
        case COMBAT_TILE_128x16: return 32;
        case COMBAT_TILE_64x64:
        case COMBAT_TILE_64x128: return 16;
        default: return COMBAT_EINVAL;
    }
}

extern "C" int combat_conv_gemm(const combat_conv_args *a, void *stream) {
    if (!a || !a->src || !a->wpack || !a->dst) return COMBAT_EINVAL;
    if (a->N <= 0 || a->H <= 0 || a->W <= 0 || a->P <= 0 || a->Q <= 0) return COMBAT_EINVAL;
    if (a->C < 8 || a->K < 8 || (a->K & 7)) return COMBAT_EINVAL;
    if (a->R != a->S || (a->R != 1 && a->R != 3)) return COMBAT_EINVAL;
    if (a->stride != 1 && a->stride != 2) return COMBAT_EINVAL;
    if (a->mode != 0 && a->mode != 1) return COMBAT_EINVAL;
    if (a->kpad <= 0 || (a->kpad & 63) || a->kpad < a->R * a->S * a->C) return COMBAT_EINVAL;
    if ((a->pro_scale == nullptr) != (a->pro_shift == nullptr)) return COMBAT_EINVAL;
    if (a->mask_scale && !a->mask_shift) return COMBAT_EINVAL;
    if (a->mask_mul_scale && !a->mask_scale) return COMBAT_EINVAL;
    if (a->stats_kind < 0 || a->stats_kind > 2 || (a->stats_kind && !a->stats)) return COMBAT_EINVAL;
    if (a->stats_kind == 2 && (!a->mask_x || !a->xh_mean || !a->xh_rstd)) return COMBAT_EINVAL;
    ConvParams p;
    p.a = *a;
    p.c_shift = ilog2_exact(a->C);
    p.s_shift = ilog2_exact(a->stride);
    if (p.c_shift < 0) return COMBAT_EINVAL;  // channel counts on this path are powers of two
    p.PQ = a->P * a->Q;
    const long M = (long)a->N * p.PQ;
    if (M > 0x7fffffffL / 8) return COMBAT_EINVAL;
    p.M = (int)M;
    p.ntaps = a->R * a->S;
    p.nkt = a->kpad / 64;
    // only the steps that contain real taps are walked
    const int need = (p.ntaps * a->C + 63) / 64;
    if (need < p.nkt) p.nkt = need;
    p.tiles_m = p.tiles_n = 0;
    hipStream_t st = as_stream(stream);
    switch (pick_tile(a)) {
        case COMBAT_TILE_128x128: return launch<128, 128>(p, st);
        case COMBAT_TILE_128x64: return launch<128, 64>(p, st);
        case COMBAT_TILE_64x64: return launch<64, 64>(p, st);
        case COMBAT_TILE_128x16: return launch<128, 16>(p, st);
        case COMBAT_TILE_64x128: return launch<64, 128>(p, st);
        default: return COMBAT_EINVAL;
    }
}
