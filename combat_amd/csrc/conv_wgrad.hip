// Convolution weight gradient on bf16 MFMA (gfx950):
//
//   dW[n][tap][c] += sum_{m in pixel range} dy[m][n] * prologue(src[pix(m,tap)][c])
//
// GEMM view: rows = dy channels (n), cols = src channels (c), reduction = pixels.  Both operands
// live in HBM pixel-major (NHWC), i.e. with the reduction index as the SLOW dimension, so the
// 64-pixel tiles are staged into LDS exactly as loaded ([pixel][channel] rows) and the MFMA
// fragments are fetched with ds_read_b64_tr_b16, whose 4x16 transposing read hands each lane
// 4 consecutive pixels of one channel -- no shuffles, no transposed copy.
// One workgroup = one (n-tile, c-tile, tap, pixel range); partial sums over pixel ranges are
// combined with fp32 atomics on the [K][taps][c] gradient (contiguous 256-B runs per wave).
//
// Replaces the weight-gradient half of autograd's conv backward for the lines listed in
// conv_gemm.hip; the prologue recomputes the normalised/activated conv input from the saved
// pre-normalisation tensor, as the forward kernel does.
#include "conv_common.hpp"
#include "plan.hpp"

namespace {

struct WgradParams {
    combat_wgrad_args a;
    int M, PQ, ntaps, tiles_k, tiles_c, split, pix_per_split;
    float *ws;   // C = 8 kernel: per-workgroup partial gradients (plain stores + a reduction launch instead of atomics)
};

template <int BMC, int BNC>
struct WCfg {
    static constexpr int WGK = (BNC == 16) ? 4 : (BMC == 16 ? 1 : 2);
    static constexpr int WGC = 4 / WGK;
    static constexpr int WK = BMC / WGK;
    static constexpr int WC = BNC / WGC;
    static constexpr int FK = WK / 16;
    static constexpr int FC = WC / 16;
    static constexpr int SK = (BMC + 16) * 2;  // LDS row strides in bytes
    static constexpr int SC = (BNC + 16) * 2;
    static constexpr int CHK = BMC / 8;
    static constexpr int CHC = BNC / 8;
    static constexpr int ITK = (64 * CHK + 255) / 256;
    static constexpr int ITC = (64 * CHC + 255) / 256;
    static constexpr int BUF = 64 * (SK + SC);
    static constexpr int EPS = BNC + 4;
    static constexpr int EP_BYTES = BMC * EPS * 4;
    static constexpr int SMEM = (2 * BUF > EP_BYTES) ? 2 * BUF : EP_BYTES;
};

__device__ __forceinline__ s16x4_t lds_tr16(const unsigned char *p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4_t *)(reinterpret_cast<uintptr_t>(p)));
}

template <int BMC, int BNC>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
    using T = WCfg<BMC, BNC>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const combat_wgrad_args &a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wave_k = wid % T::WGK, wave_c = wid / T::WGK;

    int bid = blockIdx.x;
    const int tile_c = bid % p.tiles_c; bid /= p.tiles_c;
    const int tile_k = bid % p.tiles_k; bid /= p.tiles_k;
    const int tap = bid % p.ntaps;
    const int sp = bid / p.ntaps;
    const int k0 = tile_k * BMC, c0 = tile_c * BNC;
    const int r = tap / a.S, s = tap - r * a.S;
    const int m_begin = sp * p.pix_per_split;
    int m_end = m_begin + p.pix_per_split;
    if (m_end > p.M) m_end = p.M;
    const int nkt = (m_end - m_begin + 63) / 64;
    if (nkt <= 0) return;

    const __bf16 *__restrict__ src = reinterpret_cast<const __bf16 *>(a.src);
    const __bf16 *__restrict__ dy = reinterpret_cast<const __bf16 *>(a.dy);
    const int C = a.C, K = a.K, H = a.H, W = a.W;
    const bool pro_affine = a.pro_scale != nullptr;

    u32x4_t rk[T::ITK];
    uint4 rc[T::ITC];
    unsigned vc = 0;
    int gc[T::ITC];

    auto load_tile = [&](int kt) {
        const int mt = m_begin + kt * 64;
#pragma unroll
        for (int i = 0; i < T::ITK; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / T::CHK, ch = idx % T::CHK;
            const int m = mt + row, n = k0 + ch * 8;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (idx < 64 * T::CHK && m < m_end && n < K) v = *reinterpret_cast<const u32x4_t *>(dy + (size_t)m * K + n);
            rk[i] = v;
        }
        vc = 0;
#pragma unroll
        for (int i = 0; i < T::ITC; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / T::CHC, ch = idx % T::CHC;
            const int m = mt + row, c = c0 + ch * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            gc[i] = 0;
            if (idx < 64 * T::CHC && m < m_end && c < C) {
                const int img = m / p.PQ, rem = m - img * p.PQ;
                const int oy = rem / a.Q, ox = rem - oy * a.Q;
                const int iy = oy * a.stride - a.pad + r, ix = ox * a.stride - a.pad + s;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                    v = *reinterpret_cast<const uint4 *>(src + ((size_t)(img * H + iy) * W + ix) * C + c);
                    vc |= 1u << i;
                    gc[i] = img * a.pro_group_stride;
                }
            }
            rc[i] = v;
        }
    };

    auto store_tile = [&](int buf) {
        unsigned char *kl = smem + buf * T::BUF;
        unsigned char *cl = kl + 64 * T::SK;
#pragma unroll
        for (int i = 0; i < T::ITK; ++i) {
            const int idx = tid + 256 * i;
            if (idx < 64 * T::CHK) {
                const int row = idx / T::CHK, ch = idx % T::CHK;
                *reinterpret_cast<u32x4_t *>(kl + row * T::SK + ch * 16) = rk[i];
            }
        }
#pragma unroll
        for (int i = 0; i < T::ITC; ++i) {
            const int idx = tid + 256 * i;
            if (idx < 64 * T::CHC) {
                const int row = idx / T::CHC, ch = idx % T::CHC;
                uint4 val = rc[i];
                if ((pro_affine || a.pro_act) && ((vc >> i) & 1u)) {
                    float v[8];
                    unpack8(val, v);
                    if (pro_affine) {
                        float sc[8], sh[8];
                        load8f(a.pro_scale + gc[i] + c0 + ch * 8, sc);
                        load8f(a.pro_shift + gc[i] + c0 + ch * 8, sh);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
                    }
                    if (a.pro_act) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.pro_slope;
                    }
                    val = pack8(v);
                }
                *reinterpret_cast<uint4 *>(cl + row * T::SC + ch * 16) = val;
            }
        }
    };

    f32x4_t acc[T::FK][T::FC];
#pragma unroll
    for (int i = 0; i < T::FK; ++i)
#pragma unroll
        for (int j = 0; j < T::FC; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        const unsigned char *kl = smem + buf * T::BUF;
        const unsigned char *cl = kl + 64 * T::SK;
        // transposing read: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a
        // 4(pixel) x 16(channel) block and receives the 4 pixels of channel (lane & 15).
        // All fragments of the 64-pixel step first, then the MFMAs (order pinned, see conv3x3.hip).
        const int q = (lane & 15) >> 2, pp = lane & 3, fq = lane >> 4;
        typedef __attribute__((ext_vector_type(8))) short s16x8_t;
        bf16x8_t fk[2][T::FK], fc[2][T::FC];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int prow = ks * 32 + fq * 8 + q;
#pragma unroll
            for (int i = 0; i < T::FK; ++i) {
                const unsigned char *b = kl + prow * T::SK + (wave_k * T::WK + i * 16 + pp * 4) * 2;
                const s16x4_t lo = lds_tr16(b), hi = lds_tr16(b + 4 * T::SK);
                const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                fk[ks][i] = __builtin_bit_cast(bf16x8_t, v);
            }
#pragma unroll
            for (int j = 0; j < T::FC; ++j) {
                const unsigned char *b = cl + prow * T::SC + (wave_c * T::WC + j * 16 + pp * 4) * 2;
                const s16x4_t lo = lds_tr16(b), hi = lds_tr16(b + 4 * T::SC);
                const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                fc[ks][j] = __builtin_bit_cast(bf16x8_t, v);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < T::FK; ++i)
#pragma unroll
                for (int j = 0; j < T::FC; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[ks][i], fc[ks][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + 1 < nkt;
        if (more) load_tile(kt + 1);
        compute(kt & 1);
        if (more) store_tile((kt + 1) & 1);
        __syncthreads();
    }

    // accumulators (lane: 4 consecutive dy channels of one src channel) -> fp32 LDS image
    float *ep = reinterpret_cast<float *>(smem);
    {
        const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
        for (int i = 0; i < T::FK; ++i)
#pragma unroll
            for (int j = 0; j < T::FC; ++j) {
                const int col = wave_c * T::WC + j * 16 + fr;
                const int row = wave_k * T::WK + i * 16 + fq * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) ep[(row + e) * T::EPS + col] = acc[i][j][e];
            }
    }
    __syncthreads();
    if (p.ws) {   // deterministic mode: this pixel range's tile as a slab; wgrad_slab_reduce_kernel sums the ranges in order
        float *slab = p.ws + ((((size_t)sp * p.ntaps + tap) * p.tiles_k + tile_k) * p.tiles_c + tile_c) * (size_t)(BMC * BNC);
        for (int idx = tid; idx < BMC * BNC; idx += 256) {
            const int row = idx / BNC, col = idx - row * BNC;
            slab[idx] = ep[row * T::EPS + col];
        }
        return;
    }
    const int creal = a.c_real;
    for (int idx = tid; idx < BMC * BNC; idx += 256) {
        const int row = idx / BNC, col = idx - row * BNC;
        const int n = k0 + row;
        int c = c0 + col;
        if (n >= a.k_real || c >= C) continue;
        if (creal < C) {  // hi/lo image channels fold onto the real ones
            if (c >= 2 * creal) continue;
            if (c >= creal) c -= creal;
        }
        atomicAdd(a.dw + ((size_t)n * p.ntaps + tap) * creal + c, ep[row * T::EPS + col]);
    }
}

// ------------------------------------------------------------------ C = 8 inputs, all nine taps per workgroup
// The generic kernel gives every filter tap its own workgroups, so a layer re-reads dy nine times and a
// 64-pixel step feeds two MFMAs per wave: for the 8-channel inputs (classifier stems, the generator's first
// convolution; they are the LAST weight gradients of their backward passes, fully exposed) that was 46-59 us.
// Here the "column" operand of a step is the pixel's im2col row [9 taps x 8 channels (+ 8 zeros)] = 80 columns,
// gathered by nine 16-byte loads per pixel from the (cache-resident, 2 MB) input, so dy is staged once and a
// step feeds ten MFMAs per wave.  64 dy channels per workgroup (four waves x 16), pixel ranges over the grid,
// fp32 atomics into dW as the generic kernel.
__global__ __launch_bounds__(256) void conv_wgrad_c8_kernel(const WgradParams p) {
    constexpr int BMC = 64, NCOL = 80, SK = (BMC + 16) * 2, SC = (NCOL + 16) * 2, BUF = 64 * (SK + SC);
    constexpr int CHK = BMC / 8, CHC = NCOL / 8, ITK = (64 * CHK + 255) / 256, ITC = (64 * CHC + 255) / 256;
    constexpr int FC = NCOL / 16, EPS = NCOL + 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const combat_wgrad_args &a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int bid = blockIdx.x;
    const int tile_k = bid % p.tiles_k;
    const int sp = bid / p.tiles_k;
    const int k0 = tile_k * BMC;
    const int m_begin = sp * p.pix_per_split;
    int m_end = m_begin + p.pix_per_split;
    if (m_end > p.M) m_end = p.M;
    const int nkt = (m_end - m_begin + 63) / 64;
    if (nkt <= 0) return;
    const __bf16 *__restrict__ src = reinterpret_cast<const __bf16 *>(a.src);
    const __bf16 *__restrict__ dy = reinterpret_cast<const __bf16 *>(a.dy);
    const int K = a.K, H = a.H, W = a.W;
    u32x4_t rk[ITK];
    uint4 rc[ITC];
    auto load_tile = [&](int kt) {
        const int mt = m_begin + kt * 64;
#pragma unroll
        for (int i = 0; i < ITK; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / CHK, ch = idx % CHK;
            const int m = mt + row;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (idx < 64 * CHK && m < m_end) v = *reinterpret_cast<const u32x4_t *>(dy + (size_t)m * K + k0 + ch * 8);
            rk[i] = v;
        }
#pragma unroll
        for (int i = 0; i < ITC; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / CHC, tap = idx % CHC;      // chunk = filter tap (9: zero padding)
            const int m = mt + row;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < 64 * CHC && m < m_end && tap < 9) {
                const int img = m / p.PQ, rem = m - img * p.PQ;
                const int oy = rem / a.Q, ox = rem - oy * a.Q;
                const int r = (tap * 11) >> 5, s = tap - r * 3;
                const int iy = oy * a.stride - a.pad + r, ix = ox * a.stride - a.pad + s;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                    v = *reinterpret_cast<const uint4 *>(src + ((size_t)(img * H + iy) * W + ix) * 8);
            }
            rc[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char *kl = smem + buf * BUF;
        unsigned char *cl = kl + 64 * SK;
#pragma unroll
        for (int i = 0; i < ITK; ++i) {
            const int idx = tid + 256 * i;
            if (idx < 64 * CHK) *reinterpret_cast<u32x4_t *>(kl + (idx / CHK) * SK + (idx % CHK) * 16) = rk[i];
        }
#pragma unroll
        for (int i = 0; i < ITC; ++i) {
            const int idx = tid + 256 * i;
            if (idx < 64 * CHC) *reinterpret_cast<uint4 *>(cl + (idx / CHC) * SC + (idx % CHC) * 16) = rc[i];
        }
    };
    f32x4_t acc[FC];
#pragma unroll
    for (int j = 0; j < FC; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) {
        const unsigned char *kl = smem + buf * BUF;
        const unsigned char *cl = kl + 64 * SK;
        const int q = (lane & 15) >> 2, pp = lane & 3, fq = lane >> 4;
        typedef __attribute__((ext_vector_type(8))) short s16x8_t;
        bf16x8_t fk[2], fc[2][FC];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int prow = ks * 32 + fq * 8 + q;
            {
                const unsigned char *b = kl + prow * SK + (wid * 16 + pp * 4) * 2;
                const s16x4_t lo = lds_tr16(b), hi = lds_tr16(b + 4 * SK);
                const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                fk[ks] = __builtin_bit_cast(bf16x8_t, v);
            }
#pragma unroll
            for (int j = 0; j < FC; ++j) {
                const unsigned char *b = cl + prow * SC + (j * 16 + pp * 4) * 2;
                const s16x4_t lo = lds_tr16(b), hi = lds_tr16(b + 4 * SC);
                const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                fc[ks][j] = __builtin_bit_cast(bf16x8_t, v);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < FC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[ks], fc[ks][j], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + 1 < nkt;
        if (more) load_tile(kt + 1);
        compute(kt & 1);
        if (more) store_tile((kt + 1) & 1);
        __syncthreads();
    }
    float *ep = reinterpret_cast<float *>(smem);
    {
        const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
        for (int j = 0; j < FC; ++j) {
            const int col = j * 16 + fr, row = wid * 16 + fq * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) ep[(row + e) * EPS + col] = acc[j][e];
        }
    }
    __syncthreads();
    const int creal = a.c_real;
    if (p.ws) {
        // dW of these layers is a few thousand floats: hundreds of workgroups adding into it atomically
        // serialise on the same addresses (that, not the arithmetic, was the generic kernel's 58 us).
        // Each workgroup stores its folded partial [64][9][creal]; conv_wgrad_c8_reduce_kernel sums them.
        float *slab = p.ws + (size_t)blockIdx.x * (BMC * 9 * creal);
        for (int idx = tid; idx < BMC * 9 * creal; idx += 256) {
            const int row = idx / (9 * creal), rem = idx - row * 9 * creal;
            const int tap = rem / creal, c = rem - tap * creal;
            float v = ep[row * EPS + tap * 8 + c];
            if (creal < 8 && c + creal < 8) v += ep[row * EPS + tap * 8 + c + creal];   // hi/lo image channels fold
            slab[idx] = v;
        }
        return;
    }
    for (int idx = tid; idx < BMC * 72; idx += 256) {
        const int row = idx / 72, col = idx - row * 72;
        const int n = k0 + row, tap = col >> 3;
        int c = col & 7;
        if (n >= a.k_real) continue;
        if (creal < 8) {  // hi/lo image channels fold onto the real ones
            if (c >= 2 * creal) continue;
            if (c >= creal) c -= creal;
        }
        atomicAdd(a.dw + ((size_t)n * 9 + tap) * creal + c, ep[row * EPS + col]);
    }
}

// dw[(k0 + row)][tap][c] += sum over pixel ranges of the slabs [range][k tile][64][9][creal]
__global__ __launch_bounds__(256) void conv_wgrad_c8_reduce_kernel(const float *__restrict__ ws, float *__restrict__ dw,
                                                                   int tiles_k, int split, int per_tile, int k_real,
                                                                   int row_elems, float *__restrict__ part) {
    // blockIdx.z = one of gridDim.z interleaved groups of pixel ranges (a thread walking all of them alone is a
    // chain of `split` dependent round trips); the groups meet in dw with a handful of atomics per element
    const int e = blockIdx.x * 256 + threadIdx.x, tile_k = blockIdx.y;
    if (e >= per_tile) return;
    const int n = tile_k * 64 + e / row_elems;
    if (n >= k_real) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const int G = gridDim.z;
    int sp = blockIdx.z;
    for (; sp + 3 * G < split; sp += 4 * G) {
        s0 += ws[((size_t)sp * tiles_k + tile_k) * per_tile + e];
        s1 += ws[((size_t)(sp + G) * tiles_k + tile_k) * per_tile + e];
        s2 += ws[((size_t)(sp + 2 * G) * tiles_k + tile_k) * per_tile + e];
        s3 += ws[((size_t)(sp + 3 * G) * tiles_k + tile_k) * per_tile + e];
    }
    for (; sp < split; sp += G) s0 += ws[((size_t)sp * tiles_k + tile_k) * per_tile + e];
    const float s = (s0 + s1) + (s2 + s3);
    if (part) part[((size_t)blockIdx.z * tiles_k + tile_k) * per_tile + e] = s;   // deterministic mode: a second pass adds the groups in order
    else atomicAdd(dw + (size_t)tile_k * per_tile + e, s);
}

int launch_c8(WgradParams p, hipStream_t st) {
    constexpr int SMEM = 2 * 64 * ((64 + 16) * 2 + (80 + 16) * 2);   // two stages; the fp32 epilogue image (64 x 84 x 4) fits
    static_assert(SMEM >= 64 * 84 * 4, "epilogue image");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wgrad_c8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess)
            return COMBAT_ELAUNCH;
        attr_set = true;
    }
    p.tiles_k = p.a.K / 64;
    p.tiles_c = 1;
    int split = p.a.split;
    const int ktiles = (p.M + 63) / 64;
    if (split <= 0) {
        split = (256 + p.tiles_k - 1) / p.tiles_k;
        const int max_split = (ktiles + 3) / 4;
        if (split > max_split) split = max_split;
        if (split < 1) split = 1;
    }
    if (split > ktiles) split = ktiles;
    p.pix_per_split = ((ktiles + split - 1) / split) * 64;
    p.split = (p.M + p.pix_per_split - 1) / p.pix_per_split;
    const int per_tile = 64 * 9 * p.a.c_real;
    const long need = (long)p.tiles_k * p.split * per_tile * 4;
    p.ws = (p.split > 1 && p.a.workspace && p.a.workspace_bytes >= need) ? reinterpret_cast<float *>(p.a.workspace) : nullptr;
    if (combat_deterministic() && p.split > 1 && !p.ws) return COMBAT_EINVAL;
    COMBAT_LAUNCH(conv_wgrad_c8_kernel, dim3((unsigned)(p.tiles_k * p.split)), dim3(256), SMEM, st, p);
    CB_LAUNCH_CHECK();
    if (p.ws) {
        const int groups = p.split >= 32 ? 16 : 1;
        // deterministic mode: the 16 groups' sums go behind the slabs and a second pass (one group: one add per element)
        // adds them in order; without room for that, one group walks every range
        float *part = nullptr;
        if (combat_deterministic() && groups > 1) {
            const long extra = (long)groups * p.tiles_k * per_tile * 4;
            if (p.a.workspace_bytes >= need + extra) part = p.ws + need / 4;
        }
        const int g1 = combat_deterministic() && !part ? 1 : groups;
        COMBAT_LAUNCH(conv_wgrad_c8_reduce_kernel, dim3((per_tile + 255) / 256, p.tiles_k, g1), dim3(256), 0, st, (const float *)p.ws,
                           p.a.dw, p.tiles_k, p.split, per_tile, p.a.k_real, 9 * p.a.c_real, part);
        CB_LAUNCH_CHECK();
        if (part) {
            COMBAT_LAUNCH(conv_wgrad_c8_reduce_kernel, dim3((per_tile + 255) / 256, p.tiles_k, 1), dim3(256), 0, st, (const float *)part,
                               p.a.dw, p.tiles_k, groups, per_tile, p.a.k_real, 9 * p.a.c_real, (float *)nullptr);
            CB_LAUNCH_CHECK();
        }
    }
    return COMBAT_OK;
}

// Deterministic mode: dw[n][tap][c] += sum over the pixel ranges (ascending) of the slabs conv_wgrad_kernel left --
// one thread per gradient element, sole owner of it; the hi / lo image channels (c and c + c_real of an 8-channel
// input, c_real < C) fold onto the real one here.
__global__ __launch_bounds__(256) void wgrad_slab_reduce_kernel(const float *__restrict__ ws, float *__restrict__ dw, int split,
                                                                int ntaps, int tiles_k, int tiles_c, int bmc, int bnc, int k_real,
                                                                int c_real, int C) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)k_real * ntaps * c_real) return;
    const int c = (int)(e % c_real), tap = (int)((e / c_real) % ntaps), n = (int)(e / ((long)c_real * ntaps));
    const int tile_k = n / bmc, row = n - tile_k * bmc;
    const size_t tile_elems = (size_t)bmc * bnc, sstride = (size_t)ntaps * tiles_k * tiles_c * tile_elems;
    float total = 0.f;
    for (int half = 0; half < 2; ++half) {
        const int cs = c + half * c_real;
        if (half && (c_real >= C || cs >= C || cs >= 2 * c_real)) break;
        const int tile_c = cs / bnc, col = cs - tile_c * bnc;
        const float *src = ws + (((size_t)tap * tiles_k + tile_k) * tiles_c + tile_c) * tile_elems + (size_t)row * bnc + col;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int sp = 0;
        for (; sp + 4 <= split; sp += 4) {      // four loads in flight; the order of the additions is fixed
            s0 += src[(size_t)sp * sstride];
            s1 += src[(size_t)(sp + 1) * sstride];
            s2 += src[(size_t)(sp + 2) * sstride];
            s3 += src[(size_t)(sp + 3) * sstride];
        }
        for (; sp < split; ++sp) s0 += src[(size_t)sp * sstride];
        total += (s0 + s1) + (s2 + s3);
    }
    dw[e] += total;
}

template <int BMC, int BNC>
int64_t generic_slab_bytes(const combat_wgrad_args &a, int split) {
    const long tiles_k = (a.K + BMC - 1) / BMC, tiles_c = (a.C + BNC - 1) / BNC;
    return (int64_t)split * a.R * a.S * tiles_k * tiles_c * BMC * BNC * 4;
}

template <int BMC, int BNC>
int launch(WgradParams p, hipStream_t st) {
    using T = WCfg<BMC, BNC>;
    static bool attr_set = false;
    auto kern = conv_wgrad_kernel<BMC, BNC>;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                T::SMEM) != hipSuccess)
            return COMBAT_ELAUNCH;
        attr_set = true;
    }
    p.tiles_k = (p.a.K + BMC - 1) / BMC;
    p.tiles_c = (p.a.C + BNC - 1) / BNC;
    int split = p.a.split;
    const int ktiles = (p.M + 63) / 64;
    if (split <= 0) {
        const int base = p.tiles_k * p.tiles_c * p.ntaps;
        // ~one workgroup per CU: the launch runs beside the input-gradient chain (auxiliary stream), and fewer,
        // longer workgroups mean fewer fp32 atomics (whole step: 256 -> 5.02 ms, 1024 -> 5.06, 128 -> 5.08, 64 -> 5.41)
        // (K <= 16 rows -- the generator's 3-channel output: a block is a sliver of work, 2304 of them halve the time)
        split = ((BMC == 16 ? 2304 : 256) + base - 1) / base;
        const int max_split = (ktiles + 3) / 4;  // at least 4 reduction steps per block
        if (split > max_split) split = max_split;
        if (split < 1) split = 1;
    }
    if (split > ktiles) split = ktiles;
    p.pix_per_split = ((ktiles + split - 1) / split) * 64;
    p.split = (p.M + p.pix_per_split - 1) / p.pix_per_split;
    const long blocks = (long)p.tiles_k * p.tiles_c * p.ntaps * p.split;
    p.ws = nullptr;
    if (combat_deterministic() && p.split > 1) {   // (one pixel range: one atomic add per element, nothing to order)
        const int64_t need = generic_slab_bytes<BMC, BNC>(p.a, p.split);
        if (!p.a.workspace || p.a.workspace_bytes < need) return COMBAT_EINVAL;
        p.ws = reinterpret_cast<float *>(p.a.workspace);
    }
    COMBAT_LAUNCH(kern, dim3((unsigned)blocks), dim3(256), T::SMEM, st, p);
    CB_LAUNCH_CHECK();
    if (p.ws) {
        const long elems = (long)p.a.k_real * p.ntaps * p.a.c_real;
        COMBAT_LAUNCH(wgrad_slab_reduce_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, st, (const float *)p.ws, p.a.dw,
                      p.split, p.ntaps, p.tiles_k, p.tiles_c, BMC, BNC, p.a.k_real, p.a.c_real, p.a.C);
        CB_LAUNCH_CHECK();
    }
    return COMBAT_OK;
}

// pixel ranges the generic kernel would use (the rule of launch<>, without launching)
template <int BMC, int BNC>
int generic_split(const combat_wgrad_args &a) {
    const long M = (long)a.N * a.P * a.Q;
    const int ktiles = (int)((M + 63) / 64), ntaps = a.R * a.S;
    const int tiles_k = (a.K + BMC - 1) / BMC, tiles_c = (a.C + BNC - 1) / BNC;
    int split = a.split > 0 ? a.split : 0;
    if (split <= 0) {
        const int base = tiles_k * tiles_c * ntaps;
        split = ((BMC == 16 ? 2304 : 256) + base - 1) / base;
        const int max_split = (ktiles + 3) / 4;
        if (split > max_split) split = max_split;
        if (split < 1) split = 1;
    }
    if (split > ktiles) split = ktiles;
    const int pix_per_split = ((ktiles + split - 1) / split) * 64;
    return (int)((M + pix_per_split - 1) / pix_per_split);
}

bool c8_applies(const combat_wgrad_args *a) {
    return a->split >= 0 && a->C == 8 && a->R == 3 && (a->K & 63) == 0 && !a->pro_scale && !a->pro_act;
}

int64_t c8_workspace(const combat_wgrad_args &a) {   // launch_c8's rule
    const long M = (long)a.N * a.P * a.Q;
    const int ktiles = (int)((M + 63) / 64), tiles_k = a.K / 64;
    int split = a.split;
    if (split <= 0) {
        split = (256 + tiles_k - 1) / tiles_k;
        const int max_split = (ktiles + 3) / 4;
        if (split > max_split) split = max_split;
        if (split < 1) split = 1;
    }
    if (split > ktiles) split = ktiles;
    const int pix_per_split = ((ktiles + split - 1) / split) * 64;
    const long nsplit = (M + pix_per_split - 1) / pix_per_split;
    if (nsplit <= 1) return 0;
    const int64_t per_tile = 64 * 9 * a.c_real * 4;
    return tiles_k * nsplit * per_tile + (nsplit >= 32 ? 16 * tiles_k * per_tile : 0);   // slabs + the 16 groups' sums
}

int64_t generic_workspace(const combat_wgrad_args &a) {
    if (a.C <= 16) return generic_slab_bytes<64, 16>(a, generic_split<64, 16>(a));
    if (a.K <= 16) return generic_slab_bytes<16, 64>(a, generic_split<16, 64>(a));
    if (a.C % 128 == 0 && a.K % 128 == 0) return generic_slab_bytes<128, 128>(a, generic_split<128, 128>(a));
    return generic_slab_bytes<64, 64>(a, generic_split<64, 64>(a));
}

}  // namespace

extern "C" int64_t combat_conv_wgrad_workspace_bytes(const combat_wgrad_args *a) {
    if (!a) return 0;
    const int64_t dma = a->split >= 0 ? (int64_t)conv_wgrad3x3_dma_workspace(a) : 0;
    if (dma > 0 || !combat_deterministic()) return dma;
    // deterministic mode: the generic kernel's slabs (layers the DMA-staged 3x3 kernel does not take; the 8-channel
    // kernel sizes its own slabs inside whatever workspace it is given)
    return c8_applies(a) ? c8_workspace(*a) : generic_workspace(*a);
}

static int validate_wgrad_args(const combat_wgrad_args *a) {
    if (!a || !a->src || !a->dy || !a->dw) return COMBAT_EINVAL;
    if (a->N <= 0 || a->H <= 0 || a->W <= 0 || a->P <= 0 || a->Q <= 0) return COMBAT_EINVAL;
    if (a->C < 8 || (a->C & 7) || a->K < 8 || (a->K & 7)) return COMBAT_EINVAL;
    if (a->R != a->S || (a->R != 1 && a->R != 3)) return COMBAT_EINVAL;
    if (a->stride != 1 && a->stride != 2) return COMBAT_EINVAL;
    if (a->c_real <= 0 || a->c_real > a->C || a->k_real <= 0 || a->k_real > a->K) return COMBAT_EINVAL;
    if ((a->pro_scale == nullptr) != (a->pro_shift == nullptr)) return COMBAT_EINVAL;
    if ((long)a->N * a->P * a->Q > 0x7fffffffL / 8) return COMBAT_EINVAL;
    return COMBAT_OK;
}

extern "C" int combat_conv_wgrad_reduce(const combat_wgrad_args *a, void *stream) {
    COMBAT_PLAN_HOOK(combat_conv_wgrad_reduce, a);
    if (validate_wgrad_args(a) != COMBAT_OK) return COMBAT_EINVAL;
    if (a->split < 0 || !a->defer_reduce) return COMBAT_OK;       // (the launch did its own reduction)
    return conv_wgrad3x3_dma_reduce(a, as_stream(stream));
}

extern "C" int combat_conv_wgrad(const combat_wgrad_args *a, void *stream) {
    COMBAT_PLAN_HOOK(combat_conv_wgrad, a);
    if (validate_wgrad_args(a) != COMBAT_OK) return COMBAT_EINVAL;
    WgradParams p;
    p.a = *a;
    p.ws = nullptr;
    p.PQ = a->P * a->Q;
    p.M = (int)((long)a->N * p.PQ);
    p.ntaps = a->R * a->S;
    hipStream_t st = as_stream(stream);
    if (a->split >= 0) {   // split < 0 forces the generic kernel (tests)
        int rc = conv_wgrad3x3_dma_try(a, st);   // prologue-free input: both operands by LDS-DMA (takes reduce_first along)
        if (rc <= 0) return rc;
    }
    if (a->reduce_first) {   // the kernels below have no reduce-first prologue: that launch's reduction as a launch of its own
        const int rc = combat_conv_wgrad_reduce(a->reduce_first, stream);
        if (rc != COMBAT_OK) return rc;
    }
    if (a->split >= 0 && !combat_deterministic()) {   // (register-staged 3x3: its pixel ranges meet through fp32 atomics only)
        const int rc = conv_wgrad3x3_try(a, st);
        if (rc <= 0) return rc;
    }
    if (a->split < 0) p.a.split = 0;
    if (c8_applies(a)) return launch_c8(p, st);
    if (a->C <= 16) return launch<64, 16>(p, st);
    if (a->K <= 16) return launch<16, 64>(p, st);
    if (a->C % 128 == 0 && a->K % 128 == 0) return launch<128, 128>(p, st);
    return launch<64, 64>(p, st);
}
