// Strided / 1x1 / 3x3 convolutions (forward and input-gradient) WITHOUT an input prologue as a gather-GEMM
// whose operands are both moved global -> LDS by the DMA path.
//
// conv_gemm.hip stages its operands through registers (it has to: the prologue touches every element)
// with one 64-deep step of prefetch, and at the one or two workgroups per CU these layers give it, a
// step costs a full global-load round trip (~1.2 us).  Prologue-free launches (eval-mode forwards whose
// producer wrote the activation, every input-gradient pass, train-mode forwards over a materialised
// activation) need no VALU work on the way in: here a reduction step (filter tap, 64-channel chunk) is
//   * 128 gathered pixel rows x 128 B: one 16-byte DMA per lane with its own source address -- stride,
//     padding and the transposed addressing of the input gradient live in that address; rows whose tap
//     falls outside the image use an out-of-range buffer offset, which lands in LDS as zeros;
//   * BN weight rows x 128 B;
// into a three-stage LDS ring, two steps ahead of the MFMAs; one counted s_waitcnt vmcnt + s_barrier
// per step.  LDS rows are unpadded (a DMA instruction writes 64 lanes x 16 B contiguously); the eight
// 16-byte slots of a row are rotated by (row & 6), which keeps ds_read_b128 fragment reads conflict
// free.  Stride-2 input gradients walk destination pixels parity-class-major (as conv_gemm.hip), so a
// tile's pixels share their 1, 2 or 4 reachable taps and the others are skipped.
// The epilogue is the shared DMA-kernel epilogue (conv_dma_epilogue.hpp).
#include "conv_dma_epilogue.hpp"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;

struct GatherParams {
    combat_conv_args a;
    int M, PQ, ntaps, cpt;       // dst pixels, P*Q, R*S, 64-channel chunks per tap
    int tiles_m, tiles_n;
    int m_fastest;               // order of an XCD's contiguous tile range (per-XCD L2: input / 8 + weights, or input + weights / 8)
    int psplit, mq;              // parity-class-major pixel order (stride-2 dgrad), pixels per class
    int s_shift;
    int pq_shift, q_shift;       // log2(P*Q), log2(Q) when both are powers of two (and, with psplit, Q >= 2, P*Q >= 4), else -1
    unsigned src_bytes, w_bytes, dst_bytes;
    int splits;                  // > 1: the reduction steps of a tile are divided among `splits` workgroups that
    float *ws;                   //      write fp32 slabs [split][tile][128][BN] here; conv_gather_finish_kernel combines
    int flavour;                 // epilogue specialisation (conv_dma_epilogue.hpp), -1: generic
    int block_base;              // first workgroup of this problem in the launch (0 unless it is the second of a pair)
    int w_prefetch;              // warm the XCD's L2 with this workgroup's weight rows at kernel start (conv3x3_dma.hip)
    int cpt2;                    // > 0: 64-channel chunks of the second reduction source (a.src2 / a.wpack2: the shortcut's
    unsigned w2_bytes;           //      input gradient as extra steps on the centre tap's pixels), bytes of its operand
    unsigned long long *stamps;  // profiling builds only (-DCOMBAT_STAMPS): per workgroup, cycles per phase
};

#ifdef COMBAT_STAMPS
static unsigned long long *g_stamps_gather_host = nullptr;
extern "C" int combat_debug_set_stamps_gather(void *p) { g_stamps_gather_host = (unsigned long long *)p; return 0; }
#define GCLK() __builtin_readcyclecounter()
#else
#define GCLK() 0ull
#endif

template <int N>
__device__ __forceinline__ void wait_vm_lgkm0_bar() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// FINISH = false: the convolution (whole, or one slab of a split reduction).  FINISH = true: the second
// launch of a split reduction: sum the slabs into the accumulators and run the fused epilogue.
// NS = stages of the LDS ring (steps are issued NS - 1 ahead); 3 is what launches use (see launch()).
// SRC2: the launch has a second reduction source (combat_conv_args.src2); a separate instantiation, because its two
// extra buffer resources and operand offsets push the kernel over its 104 scalar registers (spills inside the main
// loop: 30 -> 45 us on a stride-2 input gradient when every launch carried them).
template <int BN, bool FINISH, int NS, bool SRC2 = false>
__device__ __forceinline__ void conv_gather_dma_body(const GatherParams &p) {
    constexpr int BM = 128;
    using T = TileCfg<BM, BN, 4>;           // four waves along the pixels: a wave owns 32 pixels x all BN channels
    using EC = EpiCfg<T>;
    constexpr int WPW = BN / 32;            // weight DMA pieces per wave and step
    constexpr int PBYTES = BM * 128, WBYTES = BN * 128, SBYTES = PBYTES + WBYTES;
    constexpr int NDMA = 4 + WPW;           // DMA instructions per wave and step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const combat_conv_args &a = p.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

    int tile_m, tile_n, sp, tile;
    int pf_rank = 0, pf_size = 1;   // this workgroup's place among the workgroups of its XCD that share its weight rows
    {
        const int ntile = p.tiles_m * p.tiles_n;
        const int wg_index = blockIdx.x - (unsigned)p.block_base;   // (no cast of blockIdx.x itself: the host pass cannot convert it)
        const int nb = ntile, bid = wg_index % ntile;
        sp = wg_index / ntile;                           // slab of a split reduction (0 otherwise)
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, cnt = q + (xcd < r ? 1 : 0);
        tile = base + idx;
        if (p.m_fastest) {   // weights outweigh the input: an XCD's contiguous tiles share few weight rows, all pixels
            tile_m = tile % p.tiles_m;
            tile_n = tile / p.tiles_m;
            const int lo = base > tile_n * p.tiles_m ? base : tile_n * p.tiles_m;
            const int hi = base + cnt < (tile_n + 1) * p.tiles_m ? base + cnt : (tile_n + 1) * p.tiles_m;
            pf_rank = tile - lo;
            pf_size = hi - lo;
        } else {
            tile_n = tile % p.tiles_n;
            tile_m = tile / p.tiles_n;
            const int first = base + (tile_n - base % p.tiles_n + p.tiles_n) % p.tiles_n;
            pf_rank = (tile - first) / p.tiles_n;
            pf_size = (base + cnt - 1 - first) / p.tiles_n + 1;
        }
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int C = a.C, H = a.H, W = a.W;
#ifdef COMBAT_STAMPS
    const unsigned long long c_start = GCLK(), w_start = wall_clock64();
#endif

    // parity-class-major pixel order (psplit): class of this tile, its reachable taps (4 bits each)
    int pcls = 0, taplist = 0, ntap = p.ntaps;
    if (p.psplit) {
        pcls = m0 / p.mq;
        const int py = pcls >> 1, px = pcls & 1;   // tap r is reachable iff (oy + pad - r) is even
        ntap = 0;
        for (int r = (py + a.pad) & 1; r < a.R; r += 2)
            for (int sx = (px + a.pad) & 1; sx < a.S; sx += 2) taplist |= (r * a.S + sx) << (4 * ntap++);
    }
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.src), 0, p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.wpack), 0, p.w_bytes, 0x00020000);
    using std::integral_constant;
    // second source (a residual block's 1x1 / stride-2 shortcut, input gradient): its pixels are those of the 3x3's
    // centre tap, so its chunks are cpt2 more reduction steps "tap index ntap" -- in parity-class-major order only the
    // (even, even) class reaches them, otherwise the centre tap's validity bit decides per pixel
    const bool has2 = SRC2 && p.cpt2 > 0 && (!p.psplit || pcls == 0);
    const int nsteps_all = ntap * p.cpt + (has2 ? p.cpt2 : 0);
    int g_lo = 0, nsteps = nsteps_all;
    if (p.splits > 1) {
        const int per = (nsteps_all + p.splits - 1) / p.splits;
        g_lo = sp * per;
        nsteps = nsteps_all - g_lo < per ? nsteps_all - g_lo : per;
        if (nsteps < 0) nsteps = 0;
    }
    // ---- steps are issued strictly in order: (tap index, channel chunk) of the next one as running counters
    // (a division per step was a tenth of it)
    int jn = g_lo / p.cpt, ccn = g_lo - jn * p.cpt;
    auto advance = [&](int &j_, int &cc) __attribute__((always_inline)) {
        if (++cc == p.cpt) {
            cc = 0;
            ++j_;
        }
    };
    auto tap_of = [&](int j_) __attribute__((always_inline)) {
        if (SRC2 && j_ >= ntap) return 4;      // the second source's steps: addressed as the centre tap
        return p.psplit ? (taplist >> (4 * j_)) & 15 : j_;
    };
    unsigned wvoff[WPW], wvoff2[WPW];
#pragma unroll
    for (int j = 0; j < WPW; ++j) {
        const int n = (wid + 4 * j) * 8 + (lane >> 3), slot = lane & 7;
        wvoff[j] = (unsigned)(((n0 + n) * a.kpad + ((slot - (n & 6)) & 7) * 8) * 2);
        wvoff2[j] = SRC2 ? (unsigned)(((n0 + n) * a.kpad2 + ((slot - (n & 6)) & 7) * 8) * 2) : 0u;
    }
    auto issue_w = [&](int j_, int cc, auto stage_tag) __attribute__((always_inline)) {
        constexpr int sbase = decltype(stage_tag)::value * SBYTES;
        if (SRC2 && j_ >= ntap) {      // (uniform) the shortcut's operand: one tap, k = channel.  (Its resource is built
            // here, per step: scalar registers are what this kernel is short of.)
            const __amdgpu_buffer_rsrc_t wrsrc2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.wpack2), 0, p.w2_bytes, 0x00020000);
            const int soff = cc * 64 * 2;
#pragma unroll
            for (int j = 0; j < WPW; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc2, (lds_void_t *)(smem + sbase + PBYTES + (wid + 4 * j) * 1024), 16,
                                                         wvoff2[j], soff, 0, 0);
            return;
        }
        const int soff = (tap_of(j_) * C + cc * 64) * 2;
#pragma unroll
        for (int j = 0; j < WPW; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_void_t *)(smem + sbase + PBYTES + (wid + 4 * j) * 1024), 16,
                                                     wvoff[j], soff, 0, 0);
    };
    // dispatch on a uniform stage number (one taken branch)
    auto at_stage = [&](int stage, auto &&f) __attribute__((always_inline)) {
        if (stage == 0) f(integral_constant<int, 0>{});
        else if (stage == 1) f(integral_constant<int, 1>{});
        else if (stage == 2 || NS == 3) f(integral_constant<int, 2>{});
        else if (stage == 3 || NS == 4) f(integral_constant<int, (NS > 3 ? 3 : 0)>{});
        else if (stage == 4 || NS == 5) f(integral_constant<int, (NS > 4 ? 4 : 0)>{});
        else f(integral_constant<int, (NS > 5 ? 5 : 0)>{});
    };
    // The weights of the first NS - 1 steps go out before anything per-pixel is computed: their round trip
    // overlaps the ~4000 cycles of row decoding below (the rows follow as soon as their offsets exist).
    if (!FINISH) {
        int jw = jn, cw = ccn;
#pragma unroll
        for (int q = 0; q < NS - 1; ++q)
            if (q < nsteps) {
                at_stage(q, [&](auto st) __attribute__((always_inline)) { issue_w(jw, cw, st); });
                advance(jw, cw);
            }
    }

    // ---- weight rows of this workgroup's reduction steps -> the XCD's L2, now (see conv3x3_dma.hip: the ring asks
    // for a step's weights two steps ahead, a fraction of an HBM round trip; in the step the weights are cold).
    // A step's weights are one 128-byte line per row; lines [g_lo, g_lo + nsteps) of each of the BN rows, divided
    // among the workgroups of this XCD that share them.
    if (!FINISH && p.w_prefetch && !p.psplit && nsteps > 0) {
        const int nlines = BN * nsteps;
        unsigned char *dummy = smem + NS * SBYTES;
        for (int j = tid; pf_rank + pf_size * (j - lane) < nlines; j += 256) {
            const int l = pf_rank + pf_size * j;
            const int row = l / nsteps, col = l - row * nsteps;
            const unsigned off = l < nlines ? (unsigned)((n0 + row) * a.kpad * 2 + (g_lo + col) * 128) : kDmaOob;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_void_t *)dummy, 16, off, 0, 0, 0);
        }
    }

    // destination pixel m -> (image, y, x).  Shifts when P*Q and Q are powers of two (every CIFAR / CelebA
    // shape): eight emulated 32-bit divisions per lane were a third of the cycles in front of the main loop.
    auto decode = [&](int m, int &img, int &oy, int &ox) {
        if (p.psplit) {
            const int rem = m - pcls * p.mq, q4 = p.PQ >> 2, hq = a.Q >> 1;
            int r2, yy;
            if (p.pq_shift >= 0) {
                img = rem >> (p.pq_shift - 2);
                r2 = rem & (q4 - 1);
                yy = r2 >> (p.q_shift - 1);
            } else {
                img = rem / q4;
                r2 = rem - img * q4;
                yy = r2 / hq;
            }
            oy = 2 * yy + (pcls >> 1);
            ox = 2 * (r2 - yy * hq) + (pcls & 1);
        } else if (p.pq_shift >= 0) {
            img = m >> p.pq_shift;
            const int rem = m & (p.PQ - 1);
            oy = rem >> p.q_shift;
            ox = rem & (a.Q - 1);
        } else {
            img = m / p.PQ;
            const int rem = m - img * p.PQ;
            oy = rem / a.Q;
            ox = rem - oy * a.Q;
        }
    };

    // ---- this lane's four gathered rows (one per pixel DMA piece).  A step's source address is
    //   pixoff[j] + (uniform offset of the step's tap and channel chunk)   if the tap reaches an input pixel,
    // so everything per-lane is computed once: the row's byte offset for tap (0, 0) -- virtual: it may lie
    // outside the tensor, only sums with a valid tap's offset are used -- and one validity bit per tap
    // (padding, stride parity of the input gradient, rows beyond M).  A step then costs 3 VALU
    // instructions per piece; evaluating the bounds per step made address arithmetic half of a step.
    int pixoff[4], vmask[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wid + 4 * j) * 8 + (lane >> 3), slot = lane & 7;
        const int m = m0 + row;
        const int ch = ((slot - (row & 6)) & 7) * 16;   // bytes inside the 64-channel chunk (slot rotation)
        pixoff[j] = 0;
        vmask[j] = 0;
        if (m < p.M) {
            int img, oy, ox;
            decode(m, img, oy, ox);
            const int by = a.mode == 0 ? oy * a.stride - a.pad : oy + a.pad;
            const int bx = a.mode == 0 ? ox * a.stride - a.pad : ox + a.pad;
            const int sh = a.mode == 0 ? 0 : p.s_shift;
            pixoff[j] = ((img * H * W + (by >> sh) * W + (bx >> sh)) * C) * 2 + ch;
            // tap (r, s) is valid iff row r and column s are: 3 + 3 tests, then the outer product of the bits
            int rv = 0, cv = 0;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                bool vr, vc;
                if (a.mode == 0) {
                    vr = (unsigned)(by + t) < (unsigned)H;
                    vc = (unsigned)(bx + t) < (unsigned)W;
                } else {
                    const int ty = by - t, tx = bx - t;
                    vr = ty >= 0 && ((ty & (a.stride - 1)) == 0) && (ty >> sh) < H;
                    vc = tx >= 0 && ((tx & (a.stride - 1)) == 0) && (tx >> sh) < W;
                }
                rv |= (int)(vr && t < a.R) << t;
                cv |= (int)(vc && t < a.S) << t;
            }
            if (a.S == 3) vmask[j] = ((rv & 1) ? cv : 0) | ((rv & 2) ? cv << 3 : 0) | ((rv & 4) ? cv << 6 : 0);
            else vmask[j] = rv & cv & 1;     // 1x1
        }
        asm volatile("" : "+v"(pixoff[j]), "+v"(vmask[j]));
    }
#ifdef COMBAT_STAMPS
    const unsigned long long c_pix = GCLK();
#endif
    // gathered pixel rows of reduction step (tap index j_, channel chunk cc) into ring stage `stage`
    auto issue_a = [&](int j_, int cc, auto stage_tag) __attribute__((always_inline)) {
        constexpr int sbase = decltype(stage_tag)::value * SBYTES;
        const int tap = tap_of(j_);
        const int r = (a.S == 3) ? ((tap * 11) >> 5) : tap, s = tap - r * a.S;
        const int toff = a.mode == 0 ? ((r * W + s) * C + cc * 64) * 2
                                     : (cc * 64 - ((r >> p.s_shift) * W + (s >> p.s_shift)) * C) * 2;
        if (SRC2 && j_ >= ntap) {      // (uniform) the second source: same pixels as the centre tap, its own tensor
            const __amdgpu_buffer_rsrc_t srsrc2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.src2), 0, p.src_bytes, 0x00020000);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned off = ((vmask[j] >> tap) & 1) ? (unsigned)(pixoff[j] + toff) : kDmaOob;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srsrc2, (lds_void_t *)(smem + sbase + (wid + 4 * j) * 1024), 16, off, 0, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned off = ((vmask[j] >> tap) & 1) ? (unsigned)(pixoff[j] + toff) : kDmaOob;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srsrc, (lds_void_t *)(smem + sbase + (wid + 4 * j) * 1024), 16, off, 0, 0, 0);
        }
    };
    // a step = its pixel rows, then its weights; the counters walk on
    auto issue = [&](auto stage_tag) __attribute__((always_inline)) {
        issue_a(jn, ccn, stage_tag);
        issue_w(jn, ccn, stage_tag);
        advance(jn, ccn);
    };

    // ---- fragment read offsets: 16 consecutive rows from a multiple of 16 -> rotation key = lane & 6
    int fa[2];
    {
        const int row = lane & 15, s0 = ((lane >> 4) + (row & 6)) & 7;
        fa[0] = row * 128 + s0 * 16;
        fa[1] = row * 128 + (s0 ^ 4) * 16;
    }
    f32x4_t acc[T::FN][T::FM];
#pragma unroll
    for (int i = 0; i < T::FN; ++i)
#pragma unroll
        for (int j = 0; j < T::FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    unsigned long long t_mid = 0, ph[5] = {0, 0, 0, 0, 0};
    auto compute = [&](auto stage_tag) __attribute__((always_inline)) {
        const unsigned char *pb = smem + decltype(stage_tag)::value * SBYTES + wid * 32 * 128;
        const unsigned char *wb = smem + decltype(stage_tag)::value * SBYTES + PBYTES;
        bf16x8_t fp[2][T::FM], fw[2][T::FN];
        // both k-steps' fragment reads are issued up front, in k-step order: the MFMAs of the first wait for
        // its six or four reads only, the second k-step's reads land behind them
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int j = 0; j < T::FM; ++j) fp[ks][j] = *reinterpret_cast<const bf16x8_t *>(pb + fa[ks] + j * 2048);
#pragma unroll
            for (int i = 0; i < T::FN; ++i) fw[ks][i] = *reinterpret_cast<const bf16x8_t *>(wb + fa[ks] + i * 2048);
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef COMBAT_STAMPS
        t_mid = GCLK();
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < T::FN; ++i)
#pragma unroll
                for (int j = 0; j < T::FM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ks][i], fp[ks][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- ring of three stages, steps issued two ahead; unrolled over the stages so that every LDS
    // address is "register + immediate".  A launch with zero reachable steps (1x1 stride-2 input gradient,
    // odd pixel classes) still runs the epilogue on zero accumulators.
    const int fr = lane & 15, fq = lane >> 4;
    const size_t slab = (size_t)BM * BN;        // floats per (split, tile)
    const bool whole = p.splits <= 1;           // this workgroup also runs the epilogue
    constexpr int PF = 3;                       // epilogue fetch this many steps before the end
    if (!FINISH) {   // ... and the pixel rows of those steps; the epilogue rows are decoded while all of it flies
#pragma unroll
        for (int q = 0; q < NS - 1; ++q)
            if (q < nsteps) {
                at_stage(q, [&](auto st) __attribute__((always_inline)) { issue_a(jn, ccn, st); });
                advance(jn, ccn);
            }
    }
#ifdef COMBAT_STAMPS
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long c_iss = GCLK();
    __builtin_amdgcn_sched_barrier(0);
#endif
    // ---- fused epilogue operands: fetched PF steps before the end
    EpiRegs<T> epi;
    epi_init<T>(epi, lane, wid, n0, [&](int row) -> long {
        const int m = m0 + row;
        if (m >= p.M) return -1;
        if (!p.psplit) return (long)m * a.K;
        int img, oy, ox;
        decode(m, img, oy, ox);
        return (long)((img * a.P + oy) * a.Q + ox) * a.K;
    });

#ifdef COMBAT_STAMPS
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long c_epi = GCLK();
    __builtin_amdgcn_sched_barrier(0);
#endif
    if (FINISH) {
        // ---- second launch of a split reduction: accumulators = sum of the slabs (fragment layout), epilogue
        epi_fetch<T>(epi, a, p.dst_bytes, lane, n0);
#pragma unroll
        for (int i = 0; i < T::FN; ++i)
#pragma unroll
            for (int j = 0; j < T::FM; ++j) {
                const float *src = p.ws + (size_t)tile * slab + (wid * 32 + j * 16 + fr) * BN + i * 16 + fq * 4;
                f32x4_t v = {0.f, 0.f, 0.f, 0.f};
                for (int q = 0; q < p.splits; ++q)
                    v += *reinterpret_cast<const f32x4_t *>(src + (size_t)q * p.tiles_m * p.tiles_n * slab);
                acc[i][j] = v;
            }
        epi_finish<T>(epi, smem, acc, a, p.dst_bytes, lane, wid, n0, m0 + wid * 32 < p.M ? (long)(m0 / 32) + wid : -1L,
                      m0 + BM > p.M, p.PQ, p.flavour, 4);
        return;
    }
    // wait until step `g + 1` has landed while the `k` younger steps issued after it stay in flight
    auto wait_next = [&](int k) __attribute__((always_inline)) {
        if (NS > 5 && k >= 4) wait_vm_lgkm0_bar<4 * NDMA>();
        else if (NS > 4 && k == 3) wait_vm_lgkm0_bar<3 * NDMA>();
        else if (NS > 3 && k == 2) wait_vm_lgkm0_bar<2 * NDMA>();
        else if (k == 1) wait_vm_lgkm0_bar<NDMA>();
        else wait_vm_lgkm0_bar<0>();
    };
    if (whole && nsteps <= PF) epi_fetch<T>(epi, a, p.dst_bytes, lane, n0);
    bool fetched = !whole || nsteps <= PF;      // (a slab never fetches: `fetched` then only selects the plain waits)
    const bool drain = whole;                   // after the epilogue fetch every wait drains the queue
    {   // step 0 = the first weights (issued long ago) + the first pixel rows; younger: 4 row pieces per later step
        const int young = (nsteps < NS - 1 ? nsteps : NS - 1) - 1;
        if (drain && fetched || young <= 0) wait_vm_lgkm0_bar<0>();
        else if (young == 1) wait_vm_lgkm0_bar<4>();
        else if (NS > 3 && young == 2) wait_vm_lgkm0_bar<8>();
        else if (NS > 4 && young == 3) wait_vm_lgkm0_bar<12>();
        else wait_vm_lgkm0_bar<(NS > 5 ? 16 : 0)>();
    }
    auto body = [&](auto stage_tag, int g) __attribute__((always_inline)) {
        constexpr int stage = decltype(stage_tag)::value, nstage = (stage + NS - 1) % NS;
        const bool ahead = g + NS - 1 < nsteps;
        const unsigned long long c0 = GCLK();
        if (ahead) issue(integral_constant<int, nstage>{});
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long c1 = GCLK();
        const bool fetch_now = whole && !fetched && g + PF >= nsteps;
        if (fetch_now) {
            epi_fetch<T>(epi, a, p.dst_bytes, lane, n0);
            fetched = true;
        }
        compute(integral_constant<int, stage>{});
        // step g + 1 must have landed; younger: the steps issued after it and, if issued after step g + 1's
        // operands, the epilogue fetches.  Keep it simple and exact: drain everything whenever the
        // epilogue fetch is in flight (at most PF waits per launch).
        int young = (g + NS - 1 < nsteps ? g + NS - 1 : nsteps - 1) - (g + 1);
        const unsigned long long c3 = GCLK();
        wait_next(drain && fetched ? 0 : (young > 0 ? young : 0));
#ifdef COMBAT_STAMPS
        const unsigned long long c4 = GCLK();
        ph[0] += c1 - c0; ph[1] += t_mid - c1; ph[2] += c3 - t_mid; ph[3] += c4 - c3; ph[4] += 1;
#else
        (void)c0; (void)c1; (void)c3;
#endif
    };
#ifdef COMBAT_STAMPS
    const unsigned long long c_loop_start = GCLK();
#endif
    for (int g = 0; g < nsteps; g += NS) {
        body(integral_constant<int, 0>{}, g);
        if (g + 1 < nsteps) body(integral_constant<int, 1>{}, g + 1);
        if (g + 2 < nsteps) body(integral_constant<int, 2>{}, g + 2);
        if (NS > 3 && g + 3 < nsteps) body(integral_constant<int, (NS > 3 ? 3 : 0)>{}, g + 3);
        if (NS > 4 && g + 4 < nsteps) body(integral_constant<int, (NS > 4 ? 4 : 0)>{}, g + 4);
        if (NS > 5 && g + 5 < nsteps) body(integral_constant<int, (NS > 5 ? 5 : 0)>{}, g + 5);
    }
    (void)EC::NPF;
#ifdef COMBAT_STAMPS
    const unsigned long long c_loop_end = GCLK();
    if (threadIdx.x == 0 && p.stamps) {
        for (int q = 0; q < 5; ++q) p.stamps[blockIdx.x * 16 + q] = ph[q];
        p.stamps[blockIdx.x * 16 + 5] = c_loop_start - c_start;
        p.stamps[blockIdx.x * 16 + 6] = c_loop_end - c_loop_start;
        p.stamps[blockIdx.x * 16 + 7] = wall_clock64() - w_start;
        p.stamps[blockIdx.x * 16 + 8] = w_start;
        p.stamps[blockIdx.x * 16 + 11] = c_pix - c_start;
        p.stamps[blockIdx.x * 16 + 12] = c_iss - c_pix;
        p.stamps[blockIdx.x * 16 + 13] = c_epi - c_iss;
        p.stamps[blockIdx.x * 16 + 14] = c_loop_start - c_epi;
    }
#endif
    if (!whole) {   // this workgroup's slab, straight from the fragments (16 bytes per lane, 64 per row and quad)
        float *dst = p.ws + ((size_t)sp * p.tiles_m * p.tiles_n + tile) * slab;
#pragma unroll
        for (int i = 0; i < T::FN; ++i)
#pragma unroll
            for (int j = 0; j < T::FM; ++j)
                *reinterpret_cast<f32x4_t *>(dst + (wid * 32 + j * 16 + fr) * BN + i * 16 + fq * 4) = acc[i][j];
        return;
    }
    epi_finish<T>(epi, smem, acc, a, p.dst_bytes, lane, wid, n0, m0 + wid * 32 < p.M ? (long)(m0 / 32) + wid : -1L,
                  m0 + BM > p.M, p.PQ, p.flavour, 4);
#ifdef COMBAT_STAMPS
    if (threadIdx.x == 0 && p.stamps) {
        p.stamps[blockIdx.x * 16 + 9] = GCLK() - c_loop_end;
        p.stamps[blockIdx.x * 16 + 10] = wall_clock64();
    }
#endif
}

// (the host pass cannot instantiate the FINISH = false body -- device-only builtins in its signature-dependent
// lambdas -- so the kernels below call it in the device pass only; the host pass needs just their stubs)
#if defined(__HIP_DEVICE_COMPILE__)
#define COMBAT_GATHER_BODY(BN, NS, P) conv_gather_dma_body<BN, false, NS>(P)
#define COMBAT_GATHER_BODY2(BN, NS, P) conv_gather_dma_body<BN, false, NS, true>(P)
#else
#define COMBAT_GATHER_BODY(BN, NS, P) ((void)(P))
#define COMBAT_GATHER_BODY2(BN, NS, P) ((void)(P))
#endif

struct GatherPair {
    GatherParams p[2];
    int n0;
};

// ONE problem per launch: the parameter block's fields sit at fixed kernel-argument offsets, so the compiler keeps what
// the main loop needs in scalar registers.  (Round 2 ran every launch through the pair kernel below, whose body took
// `blockIdx.x >= n0 ? p[1] : p[0]` as a REFERENCE: a run-time pointer into the kernel-argument segment -- the compiler
// then re-loaded fields from memory inside the loop, 8 dependent `s_load_dword` + `s_waitcnt lgkmcnt(0)` per
// reduction step (271 scalar loads in the kernel against ~17 in its siblings), each of which also drains the wave's
// LDS reads: most of the "450-600 cycles of DMA issue per step" of DESIGN.md section 3.)
template <int BN, int NS>
__global__ __launch_bounds__(256, NS > 3 ? 1 : 2) void conv_gather_dma_kernel(const GatherParams p) {
    COMBAT_GATHER_BODY(BN, NS, p);
}
template <int BN, int NS>     // ... with a second reduction source
__global__ __launch_bounds__(256, NS > 3 ? 1 : 2) void conv_gather_dma_src2_kernel(const GatherParams p) {
    COMBAT_GATHER_BODY2(BN, NS, p);
}

// Workgroups [0, n0) run problem 0, the rest problem 1.  Two independent
// convolutions in ONE launch: a residual block's stride-2 first convolution and its 1x1 shortcut (preact_resnet.py:33-36,
// resnet.py:24-31: both read the same activated tensor) -- the shortcut's 0.6 GFLOP ride along in slots the 3x3 launch
// leaves idle instead of paying a dependent 12-17 us launch of their own.  The body is instantiated once per problem
// behind a uniform branch, so that each copy addresses its parameters statically (see above).
template <int BN, int NS>
__global__ __launch_bounds__(256, NS > 3 ? 1 : 2) void conv_gather_dma_pair_kernel(const GatherPair pp) {
    // (the second problem starts at a multiple of eight, so that its `index & 7` is again the blocks' XCD group: the
    // up-to-seven workgroups in between have nothing to do)
    if (blockIdx.x >= (unsigned)pp.n0) {
        if (blockIdx.x < (unsigned)pp.p[1].block_base) return;
        COMBAT_GATHER_BODY(BN, NS, pp.p[1]);
    } else {
        COMBAT_GATHER_BODY(BN, NS, pp.p[0]);
    }
}
template <int BN>
__global__ __launch_bounds__(256, 2) void conv_gather_finish_kernel(const GatherParams p) { conv_gather_dma_body<BN, true, 3>(p); }

// ------------------------------------------------------------------ 8-channel inputs (images, 3-channel gradients)
// A 3x3 convolution over C = 8 (the c8 image layout; the gradient w.r.t. a 3-channel output): the whole
// reduction is 9 taps x 8 channels = 72 <= 96 = three MFMA k-steps, and one k-block of an MFMA operand
// (8 consecutive k of one row) is exactly one tap's eight channels of one pixel = ONE 16-byte load.  So
// there is no staging at all: a wave's 32 pixels x 12 tap slots are six buffer loads per lane straight
// into the B operands (padding / absent taps: out-of-range offset = zeros), the 64 x 96 weight slice is
// twelve more, 24 MFMAs, and the shared fused epilogue.  Replaces the gather-per-tap GEMM for the stem of
// the classifiers, the generator's first convolution and the input gradient of its last one (24-39 us ->
// what the epilogue costs).
template <int BN>
__device__ __forceinline__ void conv_c8_body(const GatherParams &p) {
    constexpr int BM = 128;
    using T = TileCfg<BM, BN, 4>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const combat_conv_args &a = p.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile_m, tile_n;
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        tile_n = tile % p.tiles_n;
        tile_m = tile / p.tiles_n;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int H = a.H, W = a.W;
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.src), 0, p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.wpack), 0, p.w_bytes, 0x00020000);
    auto decode = [&](int m, int &img, int &oy, int &ox) {
        if (p.pq_shift >= 0) {
            img = m >> p.pq_shift;
            const int rem = m & (p.PQ - 1);
            oy = rem >> p.q_shift;
            ox = rem & (a.Q - 1);
        } else {
            img = m / p.PQ;
            const int rem = m - img * p.PQ;
            oy = rem / a.Q;
            ox = rem - oy * a.Q;
        }
    };
    const int fr = lane & 15, kb = lane >> 4;
    // weights: row n0 + i * 16 + fr, k = ks * 32 + kb * 8 (k = tap * 8 + channel; zero beyond tap 8)
    u32x4_t fw[3][T::FN], fp[3][T::FM];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int i = 0; i < T::FN; ++i)
            fw[ks][i] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (unsigned)(((n0 + i * 16 + fr) * a.kpad + ks * 32 + kb * 8) * 2), 0, 0);
    // pixels: fragment j row fr = tile pixel wid * 32 + j * 16 + fr; k-block kb of k-step ks = tap ks * 4 + kb
#pragma unroll
    for (int j = 0; j < T::FM; ++j) {
        const int m = m0 + wid * 32 + j * 16 + fr;
        int img = 0, oy = 0, ox = 0;
        if (m < p.M) decode(m, img, oy, ox);
        const int by = a.mode == 0 ? oy * a.stride - a.pad : oy + a.pad;
        const int bx = a.mode == 0 ? ox * a.stride - a.pad : ox + a.pad;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int tap = ks * 4 + kb, tr = (tap * 11) >> 5, ts = tap - tr * 3;
            const int iy = a.mode == 0 ? by + tr : by - tr, ix = a.mode == 0 ? bx + ts : bx - ts;
            const bool v = m < p.M && tap < 9 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            fp[ks][j] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, v ? (unsigned)(((img * H + iy) * W + ix) * 16) : kDmaOob, 0, 0);
        }
    }
    EpiRegs<T> epi;
    epi_init<T>(epi, lane, wid, n0, [&](int row) -> long {
        const int m = m0 + row;
        return m < p.M ? (long)m * a.K : -1;
    });
    epi_fetch<T>(epi, a, p.dst_bytes, lane, n0);
    f32x4_t acc[T::FN][T::FM];
#pragma unroll
    for (int i = 0; i < T::FN; ++i)
#pragma unroll
        for (int j = 0; j < T::FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int i = 0; i < T::FN; ++i)
#pragma unroll
            for (int j = 0; j < T::FM; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fw[ks][i]),
                                                                    __builtin_bit_cast(bf16x8_t, fp[ks][j]), acc[i][j], 0, 0, 0);
    epi_finish<T>(epi, smem, acc, a, p.dst_bytes, lane, wid, n0, m0 + wid * 32 < p.M ? (long)(m0 / 32) + wid : -1L,
                  m0 + BM > p.M, p.PQ, p.flavour, 4);
}

#ifndef COMBAT_C8_WAVES
#define COMBAT_C8_WAVES 2
#endif
__global__ __launch_bounds__(256, COMBAT_C8_WAVES) void conv_c8_kernel(const GatherParams p) { conv_c8_body<64>(p); }

// Workgroups per tile for a launch with `tiles` tiles and `nsteps` reduction steps: split the reduction
// only when the tiles alone leave most of the chip idle (skinny layers: 2x2 / 4x4 feature maps).
int pick_splits(long tiles, int nsteps) {
    if (tiles >= 96 || nsteps < 8) return 1;
    int s = (int)(512 / tiles);   // two workgroups per CU: one alone waits ~2000 cycles per gathered stage
    if (s > 16) s = 16;
    if (s > nsteps / 4) s = nsteps / 4;
    return s < 2 ? 1 : s;
}

bool psplit_ok(const combat_conv_args &a) {
    return a.mode == 1 && a.stride == 2 && (a.P & 1) == 0 && (a.Q & 1) == 0 &&
           (a.stats_kind == 0 || a.mask_group_stride == 0);
}

template <int BN>
void fill(const combat_conv_args *a, GatherParams &p) {
    p.a = *a;
    p.PQ = a->P * a->Q;
    p.M = a->N * p.PQ;
    p.ntaps = a->R * a->S;
    p.cpt = a->C / 64;
    p.tiles_m = (p.M + 127) / 128;
    p.tiles_n = a->K / BN;
    p.m_fastest = (long)a->N * a->H * a->W * a->C * 2 < (long)a->K * a->C * p.ntaps * 2;
    p.psplit = psplit_ok(*a) && (p.M / 4) % 128 == 0;
    p.mq = p.M / 4;
    p.s_shift = a->stride == 2 ? 1 : 0;
    p.pq_shift = p.q_shift = -1;
    if ((p.PQ & (p.PQ - 1)) == 0 && (a->Q & (a->Q - 1)) == 0 && a->Q >= 2 && p.PQ >= 4) {
        p.pq_shift = __builtin_ctz(p.PQ);
        p.q_shift = __builtin_ctz(a->Q);
    }
    p.src_bytes = (unsigned)((long)a->N * a->H * a->W * a->C * 2);
    p.w_bytes = (unsigned)((long)a->rows_pad * a->kpad * 2);
    p.dst_bytes = (unsigned)((long)p.M * a->K * 2);
    const int nsteps = p.ntaps * p.cpt;
    p.splits = p.psplit ? 1 : pick_splits((long)p.tiles_m * p.tiles_n, nsteps);
    const long need = (long)p.splits * p.tiles_m * p.tiles_n * 128 * BN * 4;
    if (p.splits > 1 && (!a->workspace || a->workspace_bytes < need)) p.splits = 1;
    p.ws = p.splits > 1 ? reinterpret_cast<float *>(a->workspace) : nullptr;
    p.flavour = epi_flavour_of(*a);
    p.block_base = 0;
    p.w_prefetch = (long)a->C * p.ntaps >= 1024 && !getenv("COMBAT_NO_WPREFETCH");   // >= 16 lines per weight row
    p.cpt2 = a->src2 ? p.cpt : 0;      // (the second source has the first one's shape: conv_gather_dma_src2_ok)
    p.w2_bytes = a->src2 ? (unsigned)((long)a->rows_pad2 * a->kpad2 * 2) : 0u;
#ifdef COMBAT_STAMPS
    p.stamps = g_stamps_gather_host;
#else
    p.stamps = nullptr;
#endif
}

template <int BN>
int launch(const combat_conv_args *a, hipStream_t st) {
    GatherParams p;          // (filled per call, passed by value: no host state shared between callers)
    fill<BN>(a, p);
    constexpr int stage = 128 * 128 + BN * 128;
    constexpr int ep = EpiCfg<TileCfg<128, BN, 4>>::LDS_BYTES;
    constexpr int smem = (3 * stage > ep ? 3 * stage : ep) + 1024;   // (+ the weight prefetch's scratch KB)
    // (A six-stage ring for launches of at most one workgroup per CU -- ring depth is a template parameter --
    // changed nothing on the kernel alone and cost the step 1.5 %: a workgroup holding 144 KB of LDS keeps the
    // other streams' workgroups off its CU.  What bounds a lone workgroup is in-order issue, not DMA latency.)
    auto kern = a->src2 ? conv_gather_dma_src2_kernel<BN, 3> : conv_gather_dma_kernel<BN, 3>;
    auto fin = conv_gather_finish_kernel<BN>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv_gather_dma_kernel<BN, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(conv_gather_dma_src2_kernel<BN, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(fin), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return COMBAT_ELAUNCH;
        attr_set = true;
    }
    const int tiles = p.tiles_m * p.tiles_n;
    COMBAT_LAUNCH(kern, dim3(tiles * (p.splits > 1 ? p.splits : 1)), dim3(256), smem, st, p);
    CB_LAUNCH_CHECK();
    if (p.splits > 1) {
        COMBAT_LAUNCH(fin, dim3(tiles), dim3(256), ep, st, p);
        CB_LAUNCH_CHECK();
    }
    return COMBAT_OK;
}

}  // namespace

// the C = 8 kernel applies: 3x3, 64-channel output tiles, forward (stride 1 / 2) or stride-1 input gradient
bool conv_c8_ok(const combat_conv_args *a) {
    if (a->C != 8 || a->R != 3 || a->S != 3 || (a->K & 63) || a->kpad < 96) return false;
    if (a->pro_scale || a->pro_act || a->tanh_out || (a->mask_x && a->act_dst)) return false;
    if (!(a->mode == 0 && (a->stride == 1 || a->stride == 2)) && !(a->mode == 1 && a->stride == 1)) return false;
    const long big = 0x40000000L;
    return (long)a->N * a->H * a->W * 16 < big && (long)a->rows_pad * a->kpad * 2 < big &&
           (long)a->N * a->P * a->Q * a->K * 2 < big && a->rows_pad >= a->K;
}

int conv_c8_launch(const combat_conv_args *a, hipStream_t st) {
    if (!conv_c8_ok(a)) return COMBAT_EINVAL;
    GatherParams p;
    p.a = *a;
    p.PQ = a->P * a->Q;
    p.M = a->N * p.PQ;
    p.ntaps = 9;
    p.cpt = 1;
    p.tiles_m = (p.M + 127) / 128;
    p.tiles_n = a->K / 64;
    p.m_fastest = p.psplit = 0;
    p.mq = 0;
    p.s_shift = 0;
    p.pq_shift = p.q_shift = -1;
    if ((p.PQ & (p.PQ - 1)) == 0 && (a->Q & (a->Q - 1)) == 0 && a->Q >= 2 && p.PQ >= 4) {
        p.pq_shift = __builtin_ctz(p.PQ);
        p.q_shift = __builtin_ctz(a->Q);
    }
    p.src_bytes = (unsigned)((long)a->N * a->H * a->W * 16);
    p.w_bytes = (unsigned)((long)a->rows_pad * a->kpad * 2);
    p.dst_bytes = (unsigned)((long)p.M * a->K * 2);
    p.splits = 1;
    p.ws = nullptr;
    p.flavour = epi_flavour_of(*a);
    p.w_prefetch = 0;
    p.stamps = nullptr;
    constexpr int smem = EpiCfg<TileCfg<128, 64, 4>>::LDS_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv_c8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return COMBAT_ELAUNCH;
        attr_set = true;
    }
    COMBAT_LAUNCH(conv_c8_kernel, dim3(p.tiles_m * p.tiles_n), dim3(256), smem, st, p);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

bool conv_gather_dma_parity_split(const combat_conv_args *a);

// 0 / BN (64 or 32) this kernel would use for these args
// the optional second reduction source (combat_conv_args.src2): what the kernel's extra steps assume
bool conv_gather_dma_src2_ok(const combat_conv_args *a) {
    if (!a->src2) return true;
    return a->wpack2 && a->mode == 1 && a->R == 3 && a->S == 3 && a->stride == 2 && a->pad == 1 && a->kpad2 >= a->C &&
           (a->kpad2 & 63) == 0 && a->rows_pad2 >= a->K && (long)a->rows_pad2 * a->kpad2 * 2 < 0x40000000L;
}

int conv_gather_dma_bn(const combat_conv_args *a) {
    if (!conv_gather_dma_src2_ok(a)) return 0;
    if (a->pro_scale || a->pro_act || a->tanh_out) return 0;
    if (a->mask_x && a->act_dst) return 0;
    if (a->R != a->S || (a->R != 1 && a->R != 3) || (a->stride != 1 && a->stride != 2)) return 0;
    if (a->C < 64 || (a->C & 63) || a->kpad < a->R * a->S * a->C) return 0;
    const long big = 0x40000000L;
    if ((long)a->N * a->H * a->W * a->C * 2 >= big || (long)a->rows_pad * a->kpad * 2 >= big ||
        (long)a->N * a->P * a->Q * a->K * 2 >= big)
        return 0;
    const long tiles_m = ((long)a->N * a->P * a->Q + 127) / 128;
    if (a->K % 64 == 0 && tiles_m * (a->K / 64) >= 192) return 64;
    if (a->K % 32 == 0) return 32;
    return a->K % 64 == 0 ? 64 : 0;
}

// scratch bytes a launch for these args can use (0: none): slabs of a split reduction
long conv_gather_dma_workspace(const combat_conv_args *a) {
    const int bn = conv_gather_dma_bn(a);
    if (!bn || conv_gather_dma_parity_split(a)) return 0;
    const long tiles = (((long)a->N * a->P * a->Q + 127) / 128) * (a->K / bn);
    const int s = pick_splits(tiles, a->R * a->S * (a->C / 64));
    return s > 1 ? (long)s * tiles * 128 * bn * 4 : 0;
}

bool conv_gather_dma_parity_split(const combat_conv_args *a) {
    return psplit_ok(*a) && ((long)a->N * a->P * a->Q / 4) % 128 == 0;
}

namespace {
template <int BN>
int launch_pair(const combat_conv_args *a, const combat_conv_args *b, hipStream_t st) {
    GatherPair pp;             // (2 x ~400 B of kernel arguments; filled per call, passed by value)
    fill<BN>(a, pp.p[0]);
    fill<BN>(b, pp.p[1]);
    if (pp.p[0].splits > 1 || pp.p[1].splits > 1) return 1;   // split reductions have a finish launch each: not grouped
    constexpr int stage = 128 * 128 + BN * 128;
    constexpr int ep = EpiCfg<TileCfg<128, BN, 4>>::LDS_BYTES;
    constexpr int smem = (3 * stage > ep ? 3 * stage : ep) + 1024;
    auto kern = conv_gather_dma_pair_kernel<BN, 3>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return COMBAT_ELAUNCH;
        attr_set = true;
    }
    // (tried: the two problems' workgroups alternating in groups of eight, so that the short one runs beside the long
    // one from the start instead of as a tail -- same launch time, same step)
    const int n0 = pp.p[0].tiles_m * pp.p[0].tiles_n, n1 = pp.p[1].tiles_m * pp.p[1].tiles_n;
    const int base1 = (n0 + 7) & ~7;
    pp.n0 = n0;
    pp.p[1].block_base = base1;
    COMBAT_LAUNCH(kern, dim3(base1 + n1), dim3(256), smem, st, pp);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}
}  // namespace

// both convolutions in one launch: COMBAT_OK; 1: not groupable (the caller launches them one after the other)
int conv_gather_dma_pair_launch(const combat_conv_args *a, const combat_conv_args *b, hipStream_t st) {
    const int bn = conv_gather_dma_bn(a);
    if (!bn || bn != conv_gather_dma_bn(b)) return 1;
    if (a->tile || b->tile) return 1;
    return bn == 64 ? launch_pair<64>(a, b, st) : launch_pair<32>(a, b, st);
}

int conv_gather_dma_launch(const combat_conv_args *a, hipStream_t st) {
    const int bn = conv_gather_dma_bn(a);
    if (bn == 64) return launch<64>(a, st);
    if (bn == 32) return launch<32>(a, st);
    return COMBAT_EINVAL;
}
