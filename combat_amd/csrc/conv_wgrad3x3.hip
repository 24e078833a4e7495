// Weight gradient of a 3x3 / stride-1 / pad-1 convolution with the input patch held in LDS:
//
//   dW[n][tap][c] += sum_{pixels of this workgroup's tiles} dy[pix][n] * prologue(x[pix + tap][c])
//
// The generic kernel (conv_wgrad.hip) gives every (tap, pixel range) its own workgroup, so each input
// element is fetched and normalised nine times and the fp32-atomic traffic (tiles x taps x ranges) made
// it atomic-bound.  Here one workgroup owns a 64(n) x 64(c) tile of dW for ALL nine taps (144
// accumulator registers per lane) and walks 64-pixel patches (TI images x TH x TW): per patch the dy
// tile and the (TH+2)x(TW+2) halo patch of x are staged ONCE (prologue applied once per element,
// double buffered against the MFMAs), and the nine taps read the same x image at nine byte offsets
// through ds_read_b64_tr_b16 (pixel-major operands, transposed on the way to the MFMA).  The number
// of pixel ranges is chosen against an atomic-byte budget (the chip sustains ~1.3 TB/s of fp32 atomics).
#include "conv_common.hpp"

namespace {

constexpr int kRow = 160;   // LDS bytes per pixel row (64 bf16 + pad): conflict-light tr reads, 16-B aligned
constexpr int kMaxHP = 256;
constexpr int kHIT = kMaxHP * 8 / 256;

struct W3Params {
    combat_wgrad_args a;
    int TW, TH, TI, HW, HH, HP, tw_shift, th_shift;
    int tiles_x, tiles_y, ntiles, tiles_k, tiles_c, split, per;
};

__device__ __forceinline__ s16x4_t tr16(const unsigned char *p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4_t *)(reinterpret_cast<uintptr_t>(p)));
}

__device__ __forceinline__ bf16x8_t join(const s16x4_t lo, const s16x4_t hi) {
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

__global__ __launch_bounds__(256, 1) void conv_wgrad3x3_kernel(const W3Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const combat_wgrad_args &a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wave_k = wid & 1, wave_c = wid >> 1;          // 2 x 2 waves over (n, c); wave tile 32 x 32
    int bid = blockIdx.x;
    const int tile_c = bid % p.tiles_c; bid /= p.tiles_c;
    const int tile_k = bid % p.tiles_k;
    const int sp = bid / p.tiles_k;
    const int k0 = tile_k * 64, c0 = tile_c * 64;
    const int t_begin = sp * p.per;
    int t_end = t_begin + p.per;
    if (t_end > p.ntiles) t_end = p.ntiles;
    if (t_begin >= t_end) return;

    const __bf16 *__restrict__ src = reinterpret_cast<const __bf16 *>(a.src);
    const __bf16 *__restrict__ dy = reinterpret_cast<const __bf16 *>(a.dy);
    const int C = a.C, K = a.K, H = a.H, W = a.W;
    const bool pro_affine = a.pro_scale != nullptr;
    const bool tab_uniform = a.pro_group_stride == 0 || p.TI == 1;
    const int xbytes = (p.HP * kRow + 15) & ~15;
    const int stage_bytes = 64 * kRow + xbytes;             // [dy tile | x halo patch]

    // ---- per-lane fragment addresses.  Transposing read: lane 4q+pp of a 16-lane group addresses pixel
    // row q (4 rows per read), channels 4pp..4pp+3, and receives the 4 pixels of channel (lane & 15).
    const int q = (lane & 15) >> 2, pp = lane & 3, fq = lane >> 4;
    int dyb[2][2], xb[2][2];                                // [k-step][lo/hi 4-pixel group]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
            const int pk = ks * 32 + fq * 8 + q + 4 * hi;   // pixel of the 64-pixel tile
            dyb[ks][hi] = pk * kRow + (wave_k * 32 + pp * 4) * 2;
            const int tx = pk & (p.TW - 1), ty = (pk >> p.tw_shift) & (p.TH - 1), ti = pk >> (p.tw_shift + p.th_shift);
            xb[ks][hi] = 64 * kRow + ((ti * p.HH + ty) * p.HW + tx) * kRow + (wave_c * 32 + pp * 4) * 2;
        }

    f32x4_t acc[9][2][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- halo decode (ti, hy, hx) of the chunks this thread stages: the same for every tile, so the
    // two runtime integer divisions per chunk are paid once per kernel, not once per tile
    int hdec[kHIT];
#pragma unroll
    for (int it = 0; it < kHIT; ++it) {
        const int hp = (tid + 256 * it) >> 3;
        const int hx = hp % p.HW, tt = hp / p.HW;
        hdec[it] = hx | ((tt % p.HH) << 8) | ((tt / p.HH) << 16);
    }

    // ---- staging registers of one pixel tile
    u32x4_t rdy[2], rx[kHIT];
    int gofs[kHIT];
    float psc[8], psh[8];
    const int xtotal = p.HP * 8;

    auto issue = [&](int t) __attribute__((always_inline)) {
        const int tx_ = t % p.tiles_x, ty_ = (t / p.tiles_x) % p.tiles_y, ig = t / (p.tiles_x * p.tiles_y);
        const int img0 = ig * p.TI, oy0 = ty_ * p.TH, ox0 = tx_ * p.TW;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + 256 * it, px = idx >> 3, ch = idx & 7;
            const int tx = px & (p.TW - 1), ty = (px >> p.tw_shift) & (p.TH - 1), img = img0 + (px >> (p.tw_shift + p.th_shift));
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (img < a.N)
                v = *reinterpret_cast<const u32x4_t *>(dy + ((size_t)(img * H + oy0 + ty) * W + ox0 + tx) * K + k0 + ch * 8);
            rdy[it] = v;
        }
#pragma unroll
        for (int it = 0; it < kHIT; ++it) {
            const int idx = tid + 256 * it;
            u32x4_t v = {0u, 0u, 0u, 0u};
            gofs[it] = -1;
            if (idx < xtotal) {
                const int ch = idx & 7;
                const int hx = hdec[it] & 255, hy = (hdec[it] >> 8) & 255, ti = hdec[it] >> 16;
                const int img = img0 + ti, iy = oy0 + hy - 1, ix = ox0 + hx - 1;
                if (img < a.N && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                    v = *reinterpret_cast<const u32x4_t *>(src + ((size_t)(img * H + iy) * W + ix) * C + c0 + ch * 8);
                    gofs[it] = img * a.pro_group_stride + c0 + ch * 8;
                }
            }
            rx[it] = v;
        }
        if (pro_affine && tab_uniform) {
            const int g0 = (img0 < a.N ? img0 : 0) * a.pro_group_stride + c0 + (tid & 7) * 8;
            load8f(a.pro_scale + g0, psc);
            load8f(a.pro_shift + g0, psh);
        }
    };
    auto commit = [&](unsigned char *st) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + 256 * it;
            *reinterpret_cast<u32x4_t *>(st + (idx >> 3) * kRow + (idx & 7) * 16) = rdy[it];
        }
#pragma unroll
        for (int it = 0; it < kHIT; ++it) {
            const int idx = tid + 256 * it;
            if (idx < xtotal) {
                u32x4_t val = rx[it];
                if ((pro_affine || a.pro_act) && gofs[it] >= 0) {
                    float v[8];
                    unpack8v(val, v);
                    if (pro_affine) {
                        if (tab_uniform) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], psc[e], psh[e]);
                        } else {
                            float sc[8], sh[8];
                            load8f(a.pro_scale + gofs[it], sc);
                            load8f(a.pro_shift + gofs[it], sh);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
                        }
                    }
                    if (a.pro_act) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.pro_slope;
                    }
                    val = pack8v(v);
                }
                *reinterpret_cast<u32x4_t *>(st + 64 * kRow + (idx >> 3) * kRow + (idx & 7) * 16) = val;
            }
        }
    };

    // ---- MFMAs of one staged tile: 9 taps x 2 k-steps x (2 x 2) fragments.  dy fragments of both
    // k-steps are read once per tile; the x fragments of tap t+1 (both k-steps, 8 transposing reads)
    // are in flight while the 8 MFMAs of tap t run -- 128 MFMA cycles, an LDS round trip.
    auto x_frags = [&](bf16x8_t (&fx)[2][2], const unsigned char *st, int tap) __attribute__((always_inline)) {
        const int r = (tap * 11) >> 5, s = tap - 3 * r;
        const int toff = (r * p.HW + s) * kRow;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                fx[ks][j] = join(tr16(st + xb[ks][0] + toff + j * 32), tr16(st + xb[ks][1] + toff + j * 32));
        __builtin_amdgcn_sched_barrier(0);
    };
    auto compute = [&](const unsigned char *st) __attribute__((always_inline)) {
        bf16x8_t fk[2][2], fx0[2][2], fx1[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) fk[ks][i] = join(tr16(st + dyb[ks][0] + i * 32), tr16(st + dyb[ks][1] + i * 32));
        x_frags(fx0, st, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if ((tap & 1) == 0) {
                if (tap < 8) x_frags(fx1, st, tap + 1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[tap][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[ks][i], fx0[ks][j], acc[tap][i][j], 0, 0, 0);
            } else {
                if (tap < 8) x_frags(fx0, st, tap + 1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[tap][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[ks][i], fx1[ks][j], acc[tap][i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    issue(t_begin);
    commit(smem);
    __syncthreads();
    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        const bool more = t + 1 < t_end;
        if (more) issue(t + 1);
        compute(smem + buf * stage_bytes);
        if (more) commit(smem + (buf ^ 1) * stage_bytes);
        __syncthreads();
    }

    // ---- nine [64 n][64 c] partials -> fp32 LDS image -> atomics in 256-byte runs along c
    float *ep = reinterpret_cast<float *>(smem);
    constexpr int EPS = 68;
    const int fr = lane & 15;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = wave_c * 32 + j * 16 + fr;
                const int row = wave_k * 32 + i * 16 + fq * 4;
                f32x4_t v;
                switch (tap) {   // acc is indexed statically (a runtime index would put it in scratch)
                    case 0: v = acc[0][i][j]; break;
                    case 1: v = acc[1][i][j]; break;
                    case 2: v = acc[2][i][j]; break;
                    case 3: v = acc[3][i][j]; break;
                    case 4: v = acc[4][i][j]; break;
                    case 5: v = acc[5][i][j]; break;
                    case 6: v = acc[6][i][j]; break;
                    case 7: v = acc[7][i][j]; break;
                    default: v = acc[8][i][j]; break;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) ep[(row + e) * EPS + col] = v[e];
            }
        __syncthreads();
        for (int idx = tid; idx < 64 * 64; idx += 256) {
            const int row = idx >> 6, col = idx & 63;
            const int n = k0 + row, c = c0 + col;
            if (n < a.k_real && c < a.c_real) atomicAdd(a.dw + ((size_t)n * 9 + tap) * a.c_real + c, ep[row * EPS + col]);
        }
        __syncthreads();
    }
}

bool w3_geometry(const combat_wgrad_args *a, W3Params &p, int &smem) {
    if (!(a->R == 3 && a->S == 3 && a->stride == 1 && a->pad == 1 && a->P == a->H && a->Q == a->W)) return false;
    if (a->C < 64 || (a->C & 63) || a->K < 64 || (a->K & 63) || a->c_real != a->C) return false;
    const int W = a->W, H = a->H;
    const int TW = W < 16 ? W : 16;
    const int TH = H < 64 / TW ? H : 64 / TW;
    const int TI = 64 / (TW * TH);
    if (TI < 1 || TW * TH * TI != 64) return false;
    p.a = *a;
    p.TW = TW; p.TH = TH; p.TI = TI;
    p.HW = TW + 2; p.HH = TH + 2; p.HP = TI * p.HH * p.HW;
    p.tw_shift = ilog2_exact(TW); p.th_shift = ilog2_exact(TH);
    if (p.tw_shift < 0 || p.th_shift < 0 || W % TW || H % TH || p.HP > kMaxHP) return false;
    p.tiles_x = W / TW; p.tiles_y = H / TH;
    p.ntiles = p.tiles_x * p.tiles_y * ((a->N + TI - 1) / TI);
    p.tiles_k = a->K / 64; p.tiles_c = a->C / 64;
    const int stage = 64 * kRow + ((p.HP * kRow + 15) & ~15);
    const int ep = 64 * 68 * 4;
    smem = 2 * stage > ep ? 2 * stage : ep;
    return smem <= 150 * 1024;
}

}  // namespace

// returns COMBAT_OK if launched, 1 if this kernel does not apply (caller falls back), <0 on error
int conv_wgrad3x3_try(const combat_wgrad_args *a, hipStream_t st) {
    W3Params p;
    int smem;
    if (!w3_geometry(a, p, smem)) return 1;
    const int base = p.tiles_k * p.tiles_c;
    int split = a->split;
    if (split <= 0) {
        // one workgroup per (tile, range).  Budget: <= ~24 MB of fp32 atomics (147 KB per workgroup),
        // <= one workgroup per CU, >= 4 pixel tiles per workgroup.
        const int by_atomics = 164 / base > 0 ? 164 / base : 1;
        const int by_cus = (256 + base - 1) / base;
        split = by_atomics < by_cus ? by_atomics : by_cus;
        const int by_work = (p.ntiles + 3) / 4;
        if (split > by_work) split = by_work;
        if (split < 1) split = 1;
    }
    if (split > p.ntiles) split = p.ntiles;
    p.per = (p.ntiles + split - 1) / split;
    p.split = (p.ntiles + p.per - 1) / p.per;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wgrad3x3_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
            return COMBAT_ELAUNCH;
        attr = true;
    }
    COMBAT_LAUNCH(conv_wgrad3x3_kernel, dim3(base * p.split), dim3(256), smem, st, p);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}
