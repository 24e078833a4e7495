// 3x3 / stride-1 / pad-1 convolution (forward and input-gradient) with the input patch held in LDS.
//
// The generic gather kernel (conv_gemm.hip) re-fetches -- and re-normalises -- every input element
// once per filter tap, which makes its main loop VALU-bound.  Here a workgroup owns a spatial patch
// of BM output pixels (TI images x TH x TW) and BN output channels and, per 64-channel chunk of the
// input, stages the (TH+2) x (TW+2) halo patch ONCE: global -> registers -> BatchNorm/InstanceNorm +
// (Leaky)ReLU prologue -> LDS.  The nine taps are then nine MFMA passes whose pixel operand is the
// same LDS image read at nine constant byte offsets; only the weight tile changes per tap (double
// buffered through registers).  LDS rows are 160 bytes (128 + 32 pad): conflict-free ds_read_b128
// for 16 consecutive pixels, and a tap is a plain address add (no swizzle to recompute).
//
// dgrad of such a convolution is the same kernel with the tap offset mirrored (2-r, 2-s) and the
// transposed weight pack.  Everything after the accumulators is the shared fused epilogue.
#include "conv_common.hpp"

// In-kernel phase stamps (profiling builds only: -DCOMBAT_STAMPS).  Thread 0 of every workgroup
// writes the shader clock at phase boundaries to a buffer registered with combat_debug_set_stamps.
#ifdef COMBAT_STAMPS
__device__ unsigned long long *g_stamps;
extern "C" int combat_debug_set_stamps(void *p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p)); }
#define STAMP(k)                                                                                         \
    do {                                                                                                 \
        if (threadIdx.x == 0 && g_stamps) {                                                              \
            g_stamps[blockIdx.x * 16 + (k)] = __builtin_readcyclecounter();                              \
            g_stamps[blockIdx.x * 16 + 8 + (k)] = wall_clock64();                                        \
        }                                                                                                \
    } while (0)
#else
#define STAMP(k)
#endif

namespace {

constexpr int kRow = 160;  // bytes per LDS pixel / weight row (64 bf16 + pad)

struct HaloParams {
    combat_conv_args a;
    int TW, TH, TI, HW, HH, HP;
    int tw_shift, th_shift;
    int tiles_x, tiles_y, tiles_m, tiles_n;
    int PQ, nchunks;
};

template <int BM>
struct HaloMax {  // largest halo patch (pixels) a BM-pixel tile may need
    static constexpr int HP = BM == 256 ? 400 : (BM == 128 ? 288 : 256);
    static constexpr int ITERS = (HP * 8 + 255) / 256;
};

// ===== variant 1: one tap per weight stage, one halo image (large-M tiles; small LDS footprint) =====
// weight tile of one (tap, channel chunk): global -> registers ... registers -> LDS.  Free functions
// taking the register array by reference (capturing lambdas made the compiler keep it in scratch).
template <int N>
__device__ __forceinline__ void halo1_load_w(u32x4_t (&rw)[N], const __bf16 *w_ptr, int kpad, int koff) {
#pragma unroll
    for (int j = 0; j < N; ++j) rw[j] = *reinterpret_cast<const u32x4_t *>(w_ptr + (size_t)(32 * j) * kpad + koff);
}

template <int N>
__device__ __forceinline__ void halo1_store_w(const u32x4_t (&rw)[N], unsigned char *b, int w_row0, int w_chunk) {
#pragma unroll
    for (int j = 0; j < N; ++j) *reinterpret_cast<u32x4_t *>(b + (w_row0 + 32 * j) * kRow + w_chunk * 16) = rw[j];
}

template <int BM, int BN, int WGM, int MINW>
__global__ __launch_bounds__(256, MINW) void conv3x3_halo1_kernel(const HaloParams p) {
    using T = TileCfg<BM, BN, WGM>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const combat_conv_args &a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wave_m = wid % T::WGM, wave_n = wid / T::WGM;

    int tile_m, tile_n;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        tile_n = swz % p.tiles_n;
        tile_m = swz / p.tiles_n;
    }
    const int n0 = tile_n * BN;
    const int tx_ = tile_m % p.tiles_x, ty_ = (tile_m / p.tiles_x) % p.tiles_y, ig = tile_m / (p.tiles_x * p.tiles_y);
    const int img0 = ig * p.TI, oy0 = ty_ * p.TH, ox0 = tx_ * p.TW;
    const int C = a.C, H = a.H, W = a.W;
    const __bf16 *__restrict__ src = reinterpret_cast<const __bf16 *>(a.src);
    const __bf16 *__restrict__ wp = reinterpret_cast<const __bf16 *>(a.wpack);
    unsigned char *halo = smem;
    unsigned char *wl = smem + ((p.HP * kRow + 15) & ~15);
    const bool pro_affine = a.pro_scale != nullptr;

    // per-lane LDS byte offsets of this wave's pixel fragments (tap (0,0) of the halo patch)
    int hbase[T::FM];
#pragma unroll
    for (int j = 0; j < T::FM; ++j) {
        const int pj = wave_m * T::WM + j * 16 + (lane & 15);
        const int tx = pj & (p.TW - 1), ty = (pj >> p.tw_shift) & (p.TH - 1), ti = pj >> (p.tw_shift + p.th_shift);
        hbase[j] = ((ti * p.HH + ty) * p.HW + tx) * kRow + (lane >> 4) * 16;
    }
    const int w_chunk = tid & 7, w_row0 = tid >> 3;
    const __bf16 *w_ptr = wp + (size_t)(n0 + w_row0) * a.kpad + w_chunk * 8;
    const int wfrag = (wave_n * T::WN + (lane & 15)) * kRow + (lane >> 4) * 16;

    f32x4_t acc[T::FN][T::FM];
#pragma unroll
    for (int i = 0; i < T::FN; ++i)
#pragma unroll
        for (int j = 0; j < T::FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // Weight tiles are prefetched TWO taps ahead through two register sets (rwa: even steps, rwb: odd
    // steps of the linearised (chunk, tap) sequence), so a load has two MFMA phases to land before its
    // registers are written to the LDS double buffer.
    u32x4_t rwa[T::B_ITERS], rwb[T::B_ITERS];
    const int nsteps = p.nchunks * 9;
    auto koff_of = [&](int g) {  // reduction offset of linear step g = chunk * 9 + tap
        const int cc = g / 9, tap = g - cc * 9;
        return tap * C + cc * 64;
    };
    halo1_load_w(rwa, w_ptr, a.kpad, koff_of(0));
    if (nsteps > 1) halo1_load_w(rwb, w_ptr, a.kpad, koff_of(1));

    // ---- software-pipelined tap loop -----------------------------------------------------------
    // A tap is two k-steps of (FM pixel + FN weight) fragments -> FM*FN MFMAs each.  Fragments live
    // in two register sets: while the MFMAs of one k-step run, the ds_reads of the next are in
    // flight.  sched_barrier pins "issue reads | MFMAs | (waits are inserted at first use)": left
    // alone the compiler emits one ds_read + s_waitcnt lgkmcnt(0) per 4 MFMAs, an exposed LDS round
    // trip every 64 MFMA cycles, which at 1-2 waves per SIMD made a tap cost ~3000 cycles.
    bf16x8_t fp0[T::FM], fw0[T::FN], fp1[T::FM], fw1[T::FN];
    auto tap_off = [&](int tap) {
        const int r = (tap * 11) >> 5, s = tap - 3 * r;
        return (a.mode == 0 ? (r * p.HW + s) : ((2 - r) * p.HW + (2 - s))) * kRow;
    };
    auto read_frags = [&](bf16x8_t (&fp)[T::FM], bf16x8_t (&fw)[T::FN], int tap, int ks, int buf) {
        const int toff = tap_off(tap) + ks * 64;
        const unsigned char *wb = wl + buf * (BN * kRow) + wfrag + ks * 64;
#pragma unroll
        for (int j = 0; j < T::FM; ++j) fp[j] = *reinterpret_cast<const bf16x8_t *>(halo + hbase[j] + toff);
#pragma unroll
        for (int i = 0; i < T::FN; ++i) fw[i] = *reinterpret_cast<const bf16x8_t *>(wb + i * 16 * kRow);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mfma_frags = [&](const bf16x8_t (&fp)[T::FM], const bf16x8_t (&fw)[T::FN]) {
#pragma unroll
        for (int i = 0; i < T::FN; ++i)
#pragma unroll
            for (int j = 0; j < T::FM; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fp[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // one chunk = 9 taps; `b0` = LDS weight buffer of the chunk's first step (its parity); the
    // register set holding step g+1 is rwb when g is even, rwa when g is odd
    auto run_chunk = [&](auto parity_tag, int g0) {
        constexpr int B0 = decltype(parity_tag)::value;
        read_frags(fp0, fw0, 0, 0, B0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int g = g0 + tap;
            const int buf = (B0 + tap) & 1;
            // phase A: k-step 0 of this tap; fetch k-step 1
            read_frags(fp1, fw1, tap, 1, buf);
            mfma_frags(fp0, fw0);
            // weights of the next step go to the other buffer (free since the previous tap's barrier)
            if (g + 1 < nsteps) {
                if (buf == 0) halo1_store_w(rwb, wl + BN * kRow, w_row0, w_chunk);
                else halo1_store_w(rwa, wl, w_row0, w_chunk);
            }
            if (g + 2 < nsteps) {
                if (buf == 0) halo1_load_w(rwa, w_ptr, a.kpad, koff_of(g + 2));
                else halo1_load_w(rwb, w_ptr, a.kpad, koff_of(g + 2));
            }
            __syncthreads();
            // phase B: k-step 1; fetch k-step 0 of the next tap (same chunk only: the halo changes)
            if (tap < 8) read_frags(fp0, fw0, tap + 1, 0, buf ^ 1);
            mfma_frags(fp1, fw1);
        }
    };

    STAMP(0);
    int g = 0;  // linear step; LDS weight buffer of step g is g & 1, its registers rwa (even) / rwb (odd)
    for (int cc = 0; cc < p.nchunks; ++cc) {
        // ---- stage the halo patch of this channel chunk (prologue applied once per element)
        {
            uint4 rh[HaloMax<BM>::ITERS];
            int gofs[HaloMax<BM>::ITERS];
            const int total = p.HP * 8;
#pragma unroll
            for (int it = 0; it < HaloMax<BM>::ITERS; ++it) {
                const int idx = tid + 256 * it;
                uint4 v = make_uint4(0, 0, 0, 0);
                gofs[it] = -1;
                if (idx < total) {
                    const int hp = idx >> 3, ch = idx & 7;
                    const int hx = hp % p.HW, t = hp / p.HW;
                    const int hy = t % p.HH, ti = t / p.HH;
                    const int img = img0 + ti, iy = oy0 + hy - 1, ix = ox0 + hx - 1;
                    if (img < a.N && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                        v = *reinterpret_cast<const uint4 *>(src + ((size_t)(img * H + iy) * W + ix) * C + cc * 64 + ch * 8);
                        gofs[it] = img * a.pro_group_stride + cc * 64 + ch * 8;
                    }
                }
                rh[it] = v;
            }
            // the (scale, shift) pair depends on the channel chunk (tid & 7, the same in every
            // iteration) and, for InstanceNorm, on the image: loop invariant unless a tile spans images
            auto stage = [&](auto uniform_tag) {
            constexpr bool tab_uniform = decltype(uniform_tag)::value;
            float sc[8], sh[8];
            if (pro_affine && tab_uniform) {
                const int g0 = (img0 < a.N ? img0 : 0) * a.pro_group_stride + cc * 64 + (tid & 7) * 8;
                load8f(a.pro_scale + g0, sc);
                load8f(a.pro_shift + g0, sh);
            }
#pragma unroll
            for (int it = 0; it < HaloMax<BM>::ITERS; ++it) {
                const int idx = tid + 256 * it;
                if (idx < total) {
                    uint4 val = rh[it];
                    if ((pro_affine || a.pro_act) && gofs[it] >= 0) {
                        float v[8];
                        unpack8(val, v);
                        if (pro_affine) {
                            if constexpr (!tab_uniform) {
                                load8f(a.pro_scale + gofs[it], sc);
                                load8f(a.pro_shift + gofs[it], sh);
                            }
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
                        }
                        if (a.pro_act) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.pro_slope;
                        }
                        val = pack8(v);
                    }
                    *reinterpret_cast<uint4 *>(halo + (idx >> 3) * kRow + (idx & 7) * 16) = val;
                }
            }
            };
            if (a.pro_group_stride == 0 || p.TI == 1) stage(std::true_type{}); else stage(std::false_type{});
        }
        if (cc == 0) halo1_store_w(rwa, wl, w_row0, w_chunk);   // step 0 -> buffer 0 (later chunks: done in the tap loop)
        __syncthreads();
        if (cc == 0) STAMP(1);
        // ---- nine taps (parity of the first step alternates between chunks: two instantiations)
        if ((g & 1) == 0) run_chunk(std::integral_constant<int, 0>{}, g);
        else run_chunk(std::integral_constant<int, 1>{}, g);
        g += 9;
        __syncthreads();   // every wave is done with this chunk's halo image before it is restaged
        if (cc == 0) STAMP(2);
    }
    STAMP(3);

    conv_epilogue<T>(smem, acc, a, n0, p.PQ,
                     [&](int row) {
                         const int tx = row & (p.TW - 1), ty = (row >> p.tw_shift) & (p.TH - 1);
                         const int img = img0 + (row >> (p.tw_shift + p.th_shift));
                         return img < a.N ? (img * H + oy0 + ty) * W + ox0 + tx : -1;
                     },
                     tile_m * 4 + wid);
    STAMP(4);
}

// ===== variant 2: TPS taps per weight stage, HB halo images (deep pipeline for small-M tiles) =====
// weight tiles: global -> registers ... registers -> LDS.  Free functions taking the register array
// by reference; the array type is a native vector (see common.hpp: HIP's uint4 struct would leave it in scratch).
template <int TPS, int B>
__device__ __forceinline__ void halo_load_w(u32x4_t (&rw)[TPS * B], const __bf16 *w_ptr, int kpad, int C, int tap0, int cc) {
#pragma unroll
    for (int t = 0; t < TPS; ++t)
#pragma unroll
        for (int j = 0; j < B; ++j)
            rw[t * B + j] = *reinterpret_cast<const u32x4_t *>(w_ptr + (size_t)(32 * j) * kpad + (tap0 + t) * C + cc * 64);
}

template <int TPS, int B, int BN>
__device__ __forceinline__ void halo_store_w(const u32x4_t (&rw)[TPS * B], unsigned char *buf, int w_row0, int w_chunk) {
#pragma unroll
    for (int t = 0; t < TPS; ++t)
#pragma unroll
        for (int j = 0; j < B; ++j)
            *reinterpret_cast<u32x4_t *>(buf + t * (BN * kRow) + (w_row0 + 32 * j) * kRow + w_chunk * 16) = rw[t * B + j];
}

// TPS = filter taps per weight stage (1 or 3); HB = halo images in LDS (1 or 2).
// MINW = waves per SIMD the register allocation must allow (2 for the large-M tile, whose workgroups
// overlap each other's staging/epilogue phases; 1 for the deep-pipelined small-M tile).
template <int BM, int BN, int WGM, int TPS, int HB, int MINW>
__global__ __launch_bounds__(256, MINW) void conv3x3_halo_kernel(const HaloParams p) {
    using T = TileCfg<BM, BN, WGM>;
    constexpr int NST = 9 / TPS;            // weight stages per channel chunk (odd: parity alternates per chunk)
    constexpr int WST = TPS * BN * kRow;    // bytes of one weight stage buffer
    constexpr int HIT = HaloMax<BM>::ITERS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const combat_conv_args &a = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wave_m = wid % T::WGM, wave_n = wid / T::WGM;

    int tile_m, tile_n;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        tile_n = swz % p.tiles_n;
        tile_m = swz / p.tiles_n;
    }
    const int n0 = tile_n * BN;
    const int tx_ = tile_m % p.tiles_x, ty_ = (tile_m / p.tiles_x) % p.tiles_y, ig = tile_m / (p.tiles_x * p.tiles_y);
    const int img0 = ig * p.TI, oy0 = ty_ * p.TH, ox0 = tx_ * p.TW;
    const int C = a.C, H = a.H, W = a.W;
    const __bf16 *__restrict__ src = reinterpret_cast<const __bf16 *>(a.src);
    const __bf16 *__restrict__ wp = reinterpret_cast<const __bf16 *>(a.wpack);
    const int halo_bytes = (p.HP * kRow + 15) & ~15;
    unsigned char *halo = smem;
    unsigned char *wl = smem + HB * halo_bytes;
    const bool pro_affine = a.pro_scale != nullptr;

    // per-lane LDS byte offsets of this wave's pixel fragments (tap (0,0) of the halo patch)
    int hbase[T::FM];
#pragma unroll
    for (int j = 0; j < T::FM; ++j) {
        const int pj = wave_m * T::WM + j * 16 + (lane & 15);
        const int tx = pj & (p.TW - 1), ty = (pj >> p.tw_shift) & (p.TH - 1), ti = pj >> (p.tw_shift + p.th_shift);
        hbase[j] = ((ti * p.HH + ty) * p.HW + tx) * kRow + (lane >> 4) * 16;
    }
    const int w_chunk = tid & 7, w_row0 = tid >> 3;
    const __bf16 *w_ptr = wp + (size_t)(n0 + w_row0) * a.kpad + w_chunk * 8;
    const int wfrag = (wave_n * T::WN + (lane & 15)) * kRow + (lane >> 4) * 16;

    f32x4_t acc[T::FN][T::FM];
#pragma unroll
    for (int i = 0; i < T::FN; ++i)
#pragma unroll
        for (int j = 0; j < T::FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- weight stages: stage q = chunk q / NST, taps [(q % NST) * TPS, +TPS).  Two register sets
    // (rwa: even stages, rwb: odd stages) keep two stages in flight ahead of the LDS double buffer.
    u32x4_t rwa[TPS * T::B_ITERS], rwb[TPS * T::B_ITERS];
    const int nstages = p.nchunks * NST;
    auto load_stage = [&](u32x4_t (&rw)[TPS * T::B_ITERS], int q) __attribute__((always_inline)) {
        const int cc = q / NST, st = q - cc * NST;
        halo_load_w<TPS, T::B_ITERS>(rw, w_ptr, a.kpad, C, st * TPS, cc);
    };

    // ---- halo decode (ti, hy, hx) of the chunks this thread stages: identical for every channel chunk,
    // so the two runtime integer divisions per chunk are paid once
    int hdec[HIT];
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
        const int hp = (tid + 256 * it) >> 3;
        const int hx = hp % p.HW, tt = hp / p.HW;
        hdec[it] = hx | ((tt % p.HH) << 8) | ((tt / p.HH) << 16);
    }

    // ---- halo patch of one channel chunk: issue (global -> registers) ... commit (prologue -> LDS)
    u32x4_t rh[HIT];
    int gofs[HIT];
    float psc[8], psh[8];
    const int total = p.HP * 8;
    const bool tab_uniform = a.pro_group_stride == 0 || p.TI == 1;
    auto halo_issue = [&](int cc) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < HIT; ++it) {
            const int idx = tid + 256 * it;
            u32x4_t v = {0u, 0u, 0u, 0u};
            gofs[it] = -1;
            if (idx < total) {
                const int ch = idx & 7;
                const int hx = hdec[it] & 255, hy = (hdec[it] >> 8) & 255, ti = hdec[it] >> 16;
                const int img = img0 + ti, iy = oy0 + hy - 1, ix = ox0 + hx - 1;
                if (img < a.N && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                    v = *reinterpret_cast<const u32x4_t *>(src + ((size_t)(img * H + iy) * W + ix) * C + cc * 64 + ch * 8);
                    gofs[it] = img * a.pro_group_stride + cc * 64 + ch * 8;
                }
            }
            rh[it] = v;
        }
        // the (scale, shift) pair depends on the channel chunk (tid & 7, the same in every iteration)
        // and, for InstanceNorm, on the image: loop invariant unless a tile spans several images
        if (pro_affine && tab_uniform) {
            const int g0 = (img0 < a.N ? img0 : 0) * a.pro_group_stride + cc * 64 + (tid & 7) * 8;
            load8f(a.pro_scale + g0, psc);
            load8f(a.pro_shift + g0, psh);
        }
    };
    auto halo_commit = [&](unsigned char *dstl) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < HIT; ++it) {
            const int idx = tid + 256 * it;
            if (idx < total) {
                u32x4_t val = rh[it];
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
                *reinterpret_cast<u32x4_t *>(dstl + (idx >> 3) * kRow + (idx & 7) * 16) = val;
            }
        }
    };

    // ---- software-pipelined MFMA loop ---------------------------------------------------------
    // A tap is two k-steps of (FM pixel + FN weight) fragments -> FM*FN MFMAs each.  Fragments live in
    // two register sets: while the MFMAs of one k-step run, the ds_reads of the next are in flight.
    // sched_barrier pins "issue reads | MFMAs"; left alone the compiler emits one ds_read +
    // s_waitcnt lgkmcnt(0) per 4 MFMAs, an exposed LDS round trip every 64 MFMA cycles, which at
    // 1-2 waves per SIMD made a tap cost ~3000 cycles.
    bf16x8_t fp0[T::FM], fw0[T::FN], fp1[T::FM], fw1[T::FN];
    auto tap_off = [&](int tap) __attribute__((always_inline)) {
        const int r = (tap * 11) >> 5, s = tap - 3 * r;
        return (a.mode == 0 ? (r * p.HW + s) : ((2 - r) * p.HW + (2 - s))) * kRow;
    };
    auto read_frags = [&](bf16x8_t (&fp)[T::FM], bf16x8_t (&fw)[T::FN], int tap, int ks, int wbuf, int hbuf) __attribute__((always_inline)) {
        const unsigned char *hb = halo + hbuf * halo_bytes + tap_off(tap) + ks * 64;
        const unsigned char *wb = wl + wbuf * WST + (tap % TPS) * (BN * kRow) + wfrag + ks * 64;
#pragma unroll
        for (int j = 0; j < T::FM; ++j) fp[j] = *reinterpret_cast<const bf16x8_t *>(hb + hbase[j]);
#pragma unroll
        for (int i = 0; i < T::FN; ++i) fw[i] = *reinterpret_cast<const bf16x8_t *>(wb + i * 16 * kRow);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mfma_frags = [&](const bf16x8_t (&fp)[T::FM], const bf16x8_t (&fw)[T::FN]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < T::FN; ++i)
#pragma unroll
            for (int j = 0; j < T::FM; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fp[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    // one channel chunk; B0 = parity of the chunk (= parity of its first stage, NST being odd)
    auto run_chunk = [&](auto parity_tag, int cc) __attribute__((always_inline)) {
        constexpr int B0 = decltype(parity_tag)::value;
        constexpr int HCUR = HB == 2 ? B0 : 0;
        const int q0 = cc * NST;
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const int q = q0 + st;
            constexpr int dummy = 0;
            (void)dummy;
            const int wbuf = (B0 + st) & 1;
            const bool chunk_end = st == NST - 1;
#pragma unroll
            for (int s = 0; s < 2 * TPS; ++s) {
                const int tap = st * TPS + s / 2;
                if (s + 1 < 2 * TPS) {   // inside the stage: next k-step comes from the same buffers
                    const int ntap = st * TPS + (s + 1) / 2, nks = (s + 1) & 1;
                    if ((s & 1) == 0) { read_frags(fp1, fw1, ntap, nks, wbuf, HCUR); mfma_frags(fp0, fw0); }
                    else { read_frags(fp0, fw0, ntap, nks, wbuf, HCUR); mfma_frags(fp1, fw1); }
                    (void)tap;
                    continue;
                }
                // ---- last k-step of the stage (s is odd: current fragments are fp1/fw1)
                if (q + 1 < nstages) {   // weights of the next stage -> the other LDS buffer, then refill
                    if (wbuf == 0) halo_store_w<TPS, T::B_ITERS, BN>(rwb, wl + WST, w_row0, w_chunk);
                    else halo_store_w<TPS, T::B_ITERS, BN>(rwa, wl, w_row0, w_chunk);
                }
                if (q + 3 < nstages) {
                    if (wbuf == 0) load_stage(rwb, q + 3);
                    else load_stage(rwa, q + 3);
                }
                if constexpr (HB == 2) {
                    if (chunk_end && cc + 1 < p.nchunks) halo_commit(halo + (HCUR ^ 1) * halo_bytes);
                    __syncthreads();
                    if (chunk_end && cc + 2 < p.nchunks) halo_issue(cc + 2);
                    if (q + 1 < nstages)
                        read_frags(fp0, fw0, chunk_end ? 0 : (st + 1) * TPS, 0, wbuf ^ 1, chunk_end ? (HCUR ^ 1) : HCUR);
                    mfma_frags(fp1, fw1);
                } else {
                    __syncthreads();
                    if (!chunk_end) {
                        read_frags(fp0, fw0, (st + 1) * TPS, 0, wbuf ^ 1, 0);
                        mfma_frags(fp1, fw1);
                    } else {
                        mfma_frags(fp1, fw1);
                        if (cc + 1 < p.nchunks) {   // single halo image: drain, restage, refill
                            halo_issue(cc + 1);
                            halo_commit(halo);
                            __syncthreads();
                            read_frags(fp0, fw0, 0, 0, wbuf ^ 1, 0);
                        }
                    }
                }
            }
        }
    };

    // ---- prologue: chunk 0 halo + stages 0 (-> LDS) and 1, 2 (-> registers)
    STAMP(0);
    halo_issue(0);
    load_stage(rwa, 0);
    if (nstages > 1) load_stage(rwb, 1);
    halo_commit(halo);
    halo_store_w<TPS, T::B_ITERS, BN>(rwa, wl, w_row0, w_chunk);
    __syncthreads();
    STAMP(1);
    if (HB == 2 && p.nchunks > 1) halo_issue(1);
    if (nstages > 2) load_stage(rwa, 2);
    read_frags(fp0, fw0, 0, 0, 0, 0);
    for (int cc = 0; cc < p.nchunks; ++cc) {
        if ((cc & 1) == 0) run_chunk(std::integral_constant<int, 0>{}, cc);
        else run_chunk(std::integral_constant<int, 1>{}, cc);
        if (cc == 0) STAMP(2);
    }
    __syncthreads();   // every wave is done with the LDS images before the epilogue overlays them
    STAMP(3);

    conv_epilogue<T>(smem, acc, a, n0, p.PQ,
                     [&](int row) {
                         const int tx = row & (p.TW - 1), ty = (row >> p.tw_shift) & (p.TH - 1);
                         const int img = img0 + (row >> (p.tw_shift + p.th_shift));
                         return img < a.N ? (img * H + oy0 + ty) * W + ox0 + tx : -1;
                     },
                     tile_m * 4 + wid);
    STAMP(4);
}

bool geometry(const combat_conv_args *a, int BM, int BN, HaloParams &p, int &smem, int TPS = 1, int HB = 1) {
    const int W = a->W, H = a->H;
    int TW = W < 16 ? W : 16;
    int TH = H < BM / TW ? H : BM / TW;
    int TI = BM / (TW * TH);
    if (TI < 1 || TW * TH * TI != BM) return false;
    p.a = *a;
    p.TW = TW; p.TH = TH; p.TI = TI;
    p.HW = TW + 2; p.HH = TH + 2; p.HP = TI * p.HH * p.HW;
    p.tw_shift = ilog2_exact(TW); p.th_shift = ilog2_exact(TH);
    if (p.tw_shift < 0 || p.th_shift < 0 || W % TW || H % TH) return false;
    const int hpmax = BM == 256 ? 400 : (BM == 128 ? 288 : 256);
    if (p.HP > hpmax) return false;
    p.tiles_x = W / TW; p.tiles_y = H / TH;
    p.tiles_m = p.tiles_x * p.tiles_y * ((a->N + TI - 1) / TI);
    p.tiles_n = (a->K + BN - 1) / BN;
    if (p.tiles_n * BN > a->rows_pad) return false;
    p.PQ = H * W;
    p.nchunks = a->C / 64;
    const int stage = HB * ((p.HP * kRow + 15) & ~15) + 2 * TPS * BN * kRow;
    const int ep = BM * (BN + 4) * 4;
    smem = stage > ep ? stage : ep;
    return smem <= 150 * 1024;
}

template <int BM, int BN, int WGM, int MINW>
int launch_halo1(const HaloParams &p, int smem, hipStream_t st) {
    auto kern = conv3x3_halo1_kernel<BM, BN, WGM, MINW>;
    static int attr_bytes = 0;
    if (smem > attr_bytes) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                150 * 1024) != hipSuccess)
            return COMBAT_ELAUNCH;
        attr_bytes = 150 * 1024;
    }
    COMBAT_LAUNCH(kern, dim3(p.tiles_m * p.tiles_n), dim3(256), smem, st, p);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

template <int BM, int BN, int WGM, int TPS, int HB, int MINW>
int launch_halo(const HaloParams &p, int smem, hipStream_t st) {
    auto kern = conv3x3_halo_kernel<BM, BN, WGM, TPS, HB, MINW>;
    static int attr_bytes = 0;
    if (smem > attr_bytes) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                150 * 1024) != hipSuccess)
            return COMBAT_ELAUNCH;
        attr_bytes = 150 * 1024;
    }
    COMBAT_LAUNCH(kern, dim3(p.tiles_m * p.tiles_n), dim3(256), smem, st, p);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

struct Cand {
    int tile, BM, BN;
};
const Cand kCands[] = {{COMBAT_TILE_H256x64, 256, 64}, {COMBAT_TILE_H128x128, 128, 128},
                       {COMBAT_TILE_H128x64, 128, 64}, {COMBAT_TILE_H64x64, 64, 64}};

bool applicable(const combat_conv_args *a) {
    return a->R == 3 && a->S == 3 && a->stride == 1 && a->pad == 1 && a->P == a->H && a->Q == a->W &&
           a->C >= 64 && (a->C & 63) == 0 && a->K >= 64 && (a->K & 63) == 0 && a->kpad >= 9 * a->C;
}

}  // namespace

// tile id (COMBAT_TILE_H*) the halo kernel would use for these args, or 0 if it does not apply
int conv3x3_pick(const combat_conv_args *a) {
    if (const int d = conv3x3d_pick(a)) return d;
    if (a->tile >= COMBAT_TILE_D128x64) return 0;   // (the gather tiles 12/13 are not 3x3 tiles: the caller tries them next)
    if (!applicable(a)) return 0;
    if (a->tile) {
        for (const Cand &c : kCands)
            if (c.tile == a->tile) {
                HaloParams p;
                int smem;
                return geometry(a, c.BM, c.BN, p, smem) ? c.tile : 0;
            }
        return 0;
    }
    // Measured on MI355X (profiles/r01_b_tile_sweep.txt): the 128x64 tile wins while it yields >= 512
    // workgroups (2 per CU), the 64x64 tile below that; the 4-wave-wide tiles (256x64, 128x128) need
    // > 300 VGPRs once the tap loop is software-pipelined and are kept for explicit requests only.
    HaloParams p;
    int smem;
    const bool ok8 = geometry(a, 128, 64, p, smem);
    const long blocks8 = ok8 ? (long)p.tiles_m * p.tiles_n : 0;
    if (ok8 && blocks8 >= 512) return COMBAT_TILE_H128x64;
    if (geometry(a, 64, 64, p, smem)) return COMBAT_TILE_H64x64;
    return ok8 ? COMBAT_TILE_H128x64 : 0;
}

int conv3x3_stats_layout(const combat_conv_args *a, int tile, int *rows, int *rows_per_image) {
    if (tile >= COMBAT_TILE_D128x64) return conv3x3d_stats_layout(a, tile, rows, rows_per_image);
    for (const Cand &c : kCands)
        if (c.tile == tile) {
            HaloParams p;
            int smem;
            if (!geometry(a, c.BM, c.BN, p, smem)) return COMBAT_EINVAL;
            *rows = p.tiles_m * 4;
            const int part = c.BM / 4;  // rows of one wave
            *rows_per_image = ((p.TW * p.TH) % part == 0) ? (a->H * a->W) / part : 0;
            return COMBAT_OK;
        }
    return COMBAT_EINVAL;
}

int conv3x3_launch(const combat_conv_args *a, int tile, hipStream_t st) {
    if (tile >= COMBAT_TILE_D128x64) return conv3x3d_launch(a, tile, st);
    HaloParams p;
    int smem;
    switch (tile) {
        case COMBAT_TILE_H256x64:
            if (!geometry(a, 256, 64, p, smem)) return COMBAT_EINVAL;
            return launch_halo1<256, 64, 4, 1>(p, smem, st);
        case COMBAT_TILE_H128x128:
            if (!geometry(a, 128, 128, p, smem)) return COMBAT_EINVAL;
            return launch_halo1<128, 128, 2, 1>(p, smem, st);
        case COMBAT_TILE_H128x64:
            // large-M layers: small LDS footprint, several workgroups per CU overlap each other's phases
            if (!geometry(a, 128, 64, p, smem)) return COMBAT_EINVAL;
            return launch_halo1<128, 64, 2, 1>(p, smem, st);
        case COMBAT_TILE_H64x64:
            // small-M layers (one workgroup per CU, many channel chunks): weights staged three taps at a
            // time and the next chunk's halo patch prefetched into a second LDS image
            if (!geometry(a, 64, 64, p, smem, 3, 2)) return COMBAT_EINVAL;
            return launch_halo<64, 64, 2, 3, 2, 1>(p, smem, st);
        default: return COMBAT_EINVAL;
    }
}
