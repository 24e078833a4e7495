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
#include "conv_common.hpp"
#include "plan.hpp"
#include <stdlib.h>

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
    // stride-2 input-gradient passes: destination pixels are enumerated parity-class-major
    // [(oy & 1, ox & 1)][image][oy >> 1][ox >> 1], so that a tile's pixels all see the same 1, 2 or 4
    // filter taps (of 9) and the others are skipped instead of being gathered as zeros
    int psplit;   // 0 / 1
    int mq;       // pixels per parity class (M / 4)
    int cpt;      // 64-channel chunks per tap (C / 64)
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
    // parity-class-major pixel order (psplit): class of this tile, its valid taps (4 bits each) and count
    int pcls = 0, taplist = 0, ntap_valid = 0;
    if (p.psplit) {
        pcls = m0 / p.mq;
        const int py = pcls >> 1, px = pcls & 1;   // tap r is valid iff (oy + pad - r) is even
        const int r0 = (py + a.pad) & 1, s0 = (px + a.pad) & 1;
        for (int r = r0; r < a.R; r += 2)
            for (int sx = s0; sx < a.S; sx += 2) taplist |= (r * a.S + sx) << (4 * ntap_valid++);
    }
    auto decode = [&](int m, int &img, int &oy, int &ox) {
        if (p.psplit) {
            const int rem = m - pcls * p.mq, q4 = p.PQ >> 2, hq = a.Q >> 1;
            img = rem / q4;
            const int r2 = rem - img * q4, yy = r2 / hq;
            oy = 2 * yy + (pcls >> 1);
            ox = 2 * (r2 - yy * hq) + (pcls & 1);
        } else {
            img = m / p.PQ;
            const int rem = m - img * p.PQ;
            oy = rem / a.Q;
            ox = rem - oy * a.Q;
        }
    };
    int a_pix[T::A_ITERS], a_by[T::A_ITERS], a_bx[T::A_ITERS], a_g[T::A_ITERS];
#pragma unroll
    for (int i = 0; i < T::A_ITERS; ++i) {
        const int m = m0 + a_row0 + 32 * i;
        if (m < p.M) {
            int img, oy, ox;
            decode(m, img, oy, ox);
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

    uint4 ra[T::A_ITERS];
    u32x4_t rb[T::B_ITERS];
    unsigned ra_valid = 0;
    float pscale[8], pshift[8];  // BatchNorm-style prologue (group stride 0): prefetched with the tile
    const bool pro_affine = a.pro_scale != nullptr;
    const bool pro_shared = pro_affine && a.pro_group_stride == 0;

    // reduction step kt -> (filter tap, first channel of this thread's chunk, k offset of the weight step)
    auto step_of = [&](int kt, int &tap, int &ci, int &wk) {
        if (p.psplit) {
            const int j = kt / p.cpt, cc = kt - j * p.cpt;
            tap = (taplist >> (4 * j)) & 15;
            ci = cc * 64 + a_chunk * 8;
            wk = tap * C + cc * 64;
        } else {
            const int kbase = kt * 64 + a_chunk * 8;
            tap = kbase >> p.c_shift;
            ci = kbase & (C - 1);
            wk = kt * 64;
        }
    };
    const int nkt = p.psplit ? ntap_valid * p.cpt : p.nkt;

    auto load_tile = [&](int kt) {
        int tap, ci, wk;
        step_of(kt, tap, ci, wk);
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
                rb[j] = *reinterpret_cast<const u32x4_t *>(b_ptr + (size_t)(32 * j) * a.kpad + wk);
        }
        if (pro_shared && tap_ok) {
            load8f(a.pro_scale + ci, pscale);
            load8f(a.pro_shift + ci, pshift);
        }
    };

    auto store_tile = [&](int kt, int buf) {
        unsigned char *al = smem + buf * (BM * 128);
        unsigned char *bl = smem + 2 * (BM * 128) + buf * (BN * 128);
        int tap_, ci, wk_;
        step_of(kt, tap_, ci, wk_);
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
                *reinterpret_cast<u32x4_t *>(bl + lds_off(row, a_chunk)) = rb[j];
            }
        }
    };

    f32x4_t acc[T::FN][T::FM];
#pragma unroll
    for (int i = 0; i < T::FN; ++i)
#pragma unroll
        for (int j = 0; j < T::FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // all fragments of the 64-deep step are fetched before its MFMAs and the order is pinned: the
    // compiler otherwise interleaves one ds_read + s_waitcnt lgkmcnt(0) per few MFMAs (an exposed LDS
    // round trip each time, see conv3x3.hip)
    auto compute = [&](int buf) {
        const unsigned char *al = smem + buf * (BM * 128);
        const unsigned char *bl = smem + 2 * (BM * 128) + buf * (BN * 128);
        const int fr = lane & 15, fq = lane >> 4;
        bf16x8_t pix[2][T::FM], wts[2][T::FN];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int j = 0; j < T::FM; ++j)
                pix[ks][j] = *reinterpret_cast<const bf16x8_t *>(al + lds_off(wave_m * T::WM + j * 16 + fr, ks * 4 + fq));
#pragma unroll
            for (int i = 0; i < T::FN; ++i)
                wts[ks][i] = *reinterpret_cast<const bf16x8_t *>(bl + lds_off(wave_n * T::WN + i * 16 + fr, ks * 4 + fq));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < T::FN; ++i)
#pragma unroll
                for (int j = 0; j < T::FM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wts[ks][i], pix[ks][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- main loop
    load_tile(0);
    store_tile(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + 1 < nkt;
        if (more) load_tile(kt + 1);
        compute(kt & 1);
        if (more) store_tile(kt + 1, (kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue (shared with the halo kernel)
    {
        const int gi = m0 / T::SG + wid;
        conv_epilogue<T>(smem, acc, a, n0, p.PQ,
                         [&](int row) {
                             const int m = m0 + row;
                             if (m >= p.M) return -1;
                             if (!p.psplit) return m;
                             int img, oy, ox;
                             decode(m, img, oy, ox);
                             return (img * a.P + oy) * a.Q + ox;
                         },
                         gi * T::SG < p.M ? gi : -1);
    }
}

// stride-2 input-gradient passes whose statistics (if any) are per-channel: pixels may be enumerated
// parity-class-major (the statistics rows are then not image-aligned)
bool parity_split(const combat_conv_args &a) {
    return a.mode == 1 && a.stride == 2 && (a.P & 1) == 0 && (a.Q & 1) == 0 && (a.C & 63) == 0 && a.R <= 3 &&
           (a.stats_kind == 0 || a.mask_group_stride == 0) && a.pro_group_stride == 0;
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
    q.psplit = parity_split(p.a) && (p.M / 4) % BM == 0;
    q.mq = p.M / 4;
    q.cpt = p.a.C / 64;
    if (q.tiles_n * BN > p.a.rows_pad) return COMBAT_EINVAL;
    COMBAT_LAUNCH(kern, dim3(q.tiles_m * q.tiles_n), dim3(256), T::SMEM, st, q);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

int pick_tile(const combat_conv_args *a) {
    if (a->src2) {   // a second reduction source: the gathered-DMA kernel or nothing
        if (a->tile && a->tile != COMBAT_TILE_G128x64 && a->tile != COMBAT_TILE_G128x32) return 0;
        const int bn = conv_gather_dma_bn(a);
        return !bn ? 0 : (bn == 64 ? COMBAT_TILE_G128x64 : COMBAT_TILE_G128x32);
    }
    if (a->tile == 0 && a->workspace && conv_gather_dma_workspace(a) > 0 &&
        a->workspace_bytes >= conv_gather_dma_workspace(a))   // skinny layer: split reduction beats any single-workgroup tile
        return conv_gather_dma_bn(a) == 64 ? COMBAT_TILE_G128x64 : COMBAT_TILE_G128x32;
    if (const int halo = conv3x3_pick(a)) return halo;   // 3x3 / stride 1 with the patch held in LDS
    if ((a->tile >= COMBAT_TILE_H256x64 && a->tile < COMBAT_TILE_G128x64) || a->tile == COMBAT_TILE_D256x64 ||
        a->tile == COMBAT_TILE_D256W64 || a->tile == COMBAT_TILE_S128x64)
        return 0;   // a 3x3 tile was forced but does not apply
    if ((a->tile == 0 || a->tile == COMBAT_TILE_C8) && conv_c8_ok(a)) return COMBAT_TILE_C8;
    if (a->tile == COMBAT_TILE_C8) return 0;
    if ((a->tile == COMBAT_TILE_K8 || (a->tile == 0 && !getenv("COMBAT_NO_K8"))) && conv_k8_ok(a)) return COMBAT_TILE_K8;
    if (a->tile == COMBAT_TILE_K8) return 0;
    if (a->tile == 0 || a->tile >= COMBAT_TILE_G128x64) {   // prologue-free: operands by DMA
        const int bn = conv_gather_dma_bn(a);
        if (bn) return bn == 64 ? COMBAT_TILE_G128x64 : COMBAT_TILE_G128x32;
        if (a->tile) return 0;
    }
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

extern "C" int64_t combat_conv_workspace_bytes(const combat_conv_args *a) {
    if (!a || (a->tile && a->tile != COMBAT_TILE_G128x64 && a->tile != COMBAT_TILE_G128x32)) return 0;
    return (int64_t)conv_gather_dma_workspace(a);   // (with it, a skinny 3x3 layer is routed to the gather kernel)
}

extern "C" int combat_conv_stats_granule(int tile) {
    switch (tile) {
        case COMBAT_TILE_H256x64: return 64;
        case COMBAT_TILE_128x128:
        case COMBAT_TILE_128x64:
        case COMBAT_TILE_128x16:
        case COMBAT_TILE_H128x128:
        case COMBAT_TILE_H128x64:
        case COMBAT_TILE_D128x64:
        case COMBAT_TILE_D128x32:
        case COMBAT_TILE_G128x64:
        case COMBAT_TILE_G128x32:
        case COMBAT_TILE_C8:
        case COMBAT_TILE_K8:
        case COMBAT_TILE_D256W64:
        case COMBAT_TILE_S128x64:
        case COMBAT_TILE_D256x64: return 32;
        case COMBAT_TILE_64x64:
        case COMBAT_TILE_64x128:
        case COMBAT_TILE_H64x64: return 16;
        default: return COMBAT_EINVAL;
    }
}

extern "C" int combat_conv_stats_layout(const combat_conv_args *a, int32_t *rows, int32_t *rows_per_image) {
    if (!a || !rows || !rows_per_image) return COMBAT_EINVAL;
    const int tile = pick_tile(a);
    if ((tile >= COMBAT_TILE_H256x64 && tile < COMBAT_TILE_G128x64) || tile == COMBAT_TILE_D256x64 || tile == COMBAT_TILE_D256W64 ||
        tile == COMBAT_TILE_S128x64)
        return conv3x3_stats_layout(a, tile, rows, rows_per_image);
    int gran = combat_conv_stats_granule(tile);
    if (gran <= 0) return COMBAT_EINVAL;
    if ((a->stats_kind & COMBAT_STATS_PER_WORKGROUP) &&
        (tile == COMBAT_TILE_G128x64 || tile == COMBAT_TILE_G128x32 || tile == COMBAT_TILE_C8))
        gran = 128;   // the DMA kernels' workgroup tile
    const long M = (long)a->N * a->P * a->Q, PQ = (long)a->P * a->Q;
    *rows = (int)((M + gran - 1) / gran);
    *rows_per_image = (PQ % gran == 0) ? (int)(PQ / gran) : 0;
    const int bm = (tile == COMBAT_TILE_64x64 || tile == COMBAT_TILE_64x128) ? 64 : 128;
    if (tile == COMBAT_TILE_C8) {
        // pixels in order: rows_per_image as computed
    } else if (tile == COMBAT_TILE_G128x64 || tile == COMBAT_TILE_G128x32) {
        if (conv_gather_dma_parity_split(a)) *rows_per_image = 0;
    } else if (parity_split(*a) && (M / 4) % bm == 0) {
        *rows_per_image = 0;   // parity-class-major pixel order
    }
    return COMBAT_OK;
}

// argument checks shared by combat_conv_gemm and combat_conv_gemm_pair (the pair promises results equal to two single
// calls: it must also reject what they reject, before any kernel dereferences a shape)
static int validate_conv_args(const combat_conv_args *a) {
    if (!a || !a->src || !a->wpack || (!a->dst && !a->act_dst)) return COMBAT_EINVAL;
    if (a->act_dst && (!a->act_scale || !a->act_shift)) return COMBAT_EINVAL;
    if (a->N <= 0 || a->H <= 0 || a->W <= 0 || a->P <= 0 || a->Q <= 0) return COMBAT_EINVAL;
    if (a->C < 8 || a->K < 8 || (a->K & 7)) return COMBAT_EINVAL;
    if (a->R != a->S || (a->R != 1 && a->R != 3)) return COMBAT_EINVAL;
    if (a->stride != 1 && a->stride != 2) return COMBAT_EINVAL;
    if (a->mode != 0 && a->mode != 1) return COMBAT_EINVAL;
    if (a->kpad <= 0 || (a->kpad & 63) || a->kpad < a->R * a->S * a->C) return COMBAT_EINVAL;
    if ((a->pro_scale == nullptr) != (a->pro_shift == nullptr)) return COMBAT_EINVAL;
    if (a->mask_scale && !a->mask_shift) return COMBAT_EINVAL;
    if (a->mask_mul_scale && !a->mask_scale) return COMBAT_EINVAL;
    if ((a->src2 == nullptr) != (a->wpack2 == nullptr)) return COMBAT_EINVAL;
    const int skind = a->stats_kind & 3;
    if (a->stats_kind < 0 || (a->stats_kind & ~(3 | COMBAT_STATS_PER_WORKGROUP)) || skind == 3 || (!skind && a->stats_kind) ||
        (skind && !a->stats))
        return COMBAT_EINVAL;
    if (skind == 2 && (!a->mask_x || !a->xh_mean || !a->xh_rstd)) return COMBAT_EINVAL;
    if (ilog2_exact(a->C) < 0) return COMBAT_EINVAL;   // channel counts on this path are powers of two
    if ((long)a->N * a->P * a->Q > 0x7fffffffL / 8) return COMBAT_EINVAL;
    return COMBAT_OK;
}

extern "C" int combat_conv_gemm(const combat_conv_args *a, void *stream);

// Two convolutions with no data dependence between them (neither reads what the other writes): one launch when both take
// the gather kernel with the same channel tile and unsplit reductions, otherwise a, then b.
extern "C" int combat_conv_gemm_pair(const combat_conv_args *a, const combat_conv_args *b, void *stream) {
    COMBAT_PLAN_HOOK(combat_conv_gemm_pair, a, b);
    if (validate_conv_args(a) != COMBAT_OK || validate_conv_args(b) != COMBAT_OK) return COMBAT_EINVAL;
    const int ta = pick_tile(a), tb = pick_tile(b);
    const bool gather = (ta == COMBAT_TILE_G128x64 || ta == COMBAT_TILE_G128x32) && ta == tb && !a->src2 && !b->src2;
    if (gather) {
        const int rc = conv_gather_dma_pair_launch(a, b, as_stream(stream));
        if (rc != 1) return rc;
    }
    const int rc = combat_conv_gemm(a, stream);
    return rc ? rc : combat_conv_gemm(b, stream);
}

extern "C" int combat_conv_gemm(const combat_conv_args *a, void *stream) {
    COMBAT_PLAN_HOOK(combat_conv_gemm, a);
    if (validate_conv_args(a) != COMBAT_OK) return COMBAT_EINVAL;
    ConvParams p;
    p.a = *a;
    p.c_shift = ilog2_exact(a->C);
    p.s_shift = ilog2_exact(a->stride);
    p.PQ = a->P * a->Q;
    p.M = (int)((long)a->N * p.PQ);
    p.ntaps = a->R * a->S;
    p.nkt = a->kpad / 64;
    // only the steps that contain real taps are walked
    const int need = (p.ntaps * a->C + 63) / 64;
    if (need < p.nkt) p.nkt = need;
    p.tiles_m = p.tiles_n = 0;
    hipStream_t st = as_stream(stream);
    const int tile = pick_tile(a);
    // (the prologue's second output exists on the DMA-staged 3x3 kernel only: refuse rather than silently not write it)
    if (a->pro_act_dst && tile != COMBAT_TILE_D128x64 && tile != COMBAT_TILE_D128x32) return COMBAT_EINVAL;
    if (tile == COMBAT_TILE_C8) return conv_c8_launch(a, st);
    if (tile == COMBAT_TILE_K8) return conv_k8_launch(a, st);
    if (tile == COMBAT_TILE_G128x64 || tile == COMBAT_TILE_G128x32) return conv_gather_dma_launch(a, st);
    if (tile == COMBAT_TILE_D256W64 || tile == COMBAT_TILE_S128x64) return conv3x3_launch(a, tile, st);
    if (tile >= COMBAT_TILE_H256x64) return conv3x3_launch(a, tile, st);
    switch (tile) {
        case COMBAT_TILE_128x128: return launch<128, 128>(p, st);
        case COMBAT_TILE_128x64: return launch<128, 64>(p, st);
        case COMBAT_TILE_64x64: return launch<64, 64>(p, st);
        case COMBAT_TILE_128x16: return launch<128, 16>(p, st);
        case COMBAT_TILE_64x128: return launch<64, 128>(p, st);
        default: return COMBAT_EINVAL;
    }
}
