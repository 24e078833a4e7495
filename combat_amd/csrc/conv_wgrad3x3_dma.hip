// Weight gradient of a 3x3 / stride-1 / pad-1 convolution whose input needs NO prologue (it is already
// the activated tensor), with both operands moved global -> LDS by the DMA path:
//
//   dW[n][tap][c] += sum_{pixels of this workgroup's tiles} dy[pix][n] * x[pix + tap][c]
//
// Same decomposition as conv_wgrad3x3.hip -- one workgroup owns a 64(n) x 64(c) tile of dW for all nine
// taps and walks 64-pixel patches -- but a patch costs no VALU staging work: the dy tile and the
// (TH+2) x (TW+2) halo patch of x are DMA'd (buffer_load ... lds, out-of-range offsets as the zero
// padding) into a double-buffered stage while the MFMAs of the previous patch run; the only
// synchronisation per patch is one s_waitcnt vmcnt(0) + s_barrier.  The register-staged kernel spent
// ~600 VALU instructions per patch and thread on address arithmetic and the BatchNorm/ReLU prologue
// against 72 MFMAs.
//
// LDS rows are 128 bytes (an LDS-DMA instruction writes 64 lanes x 16 B contiguously: no padding);
// the four 32-byte channel pairs of a row sit at position (pair ^ key(row)), with key chosen per
// geometry so that every ds_read_b64_tr_b16 (4 pixel rows x 32 B per 16 lanes, two groups of 8 rows per
// 32 lanes) is bank-conflict-free.  Eight waves (two per SIMD, sharing one LDS stage): wave (wave_k,
// wave_c) owns dst channels {16 wave_k + [0,16)} u {32 + 16 wave_k + [0,16)} -- its second dy fragment
// is the first one's address ^ 64 -- and src channels 16 wave_c + [0,16), for all nine taps (72
// accumulator registers).
#include "conv_common.hpp"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
constexpr unsigned kOob = 0x40000000u;

struct W3dParams {
    combat_wgrad_args a;
    int TW, TH, TI, HWP, HH, HROWS, hpw, tw_shift, th_shift;
    int tiles_x, tiles_y, ntiles, tiles_k, tiles_c, split, per;
    unsigned x_bytes, dy_bytes;
    float *ws;   // partial-sum slabs [split][tiles_k][tiles_c][9][64][64] or null (atomics)
    // the partial sums an EARLIER launch left behind (combat_wgrad_args.reduce_first), folded into its dw by this
    // launch's workgroups before they start on their own patches: see reduce_slabs()
    const float *red_ws;
    float *red_dw;
    int red_base, red_tiles_c, red_split, red_k_real, red_c_real, red_pad;
    unsigned long long *stamps;   // profiling builds only (-DCOMBAT_STAMPS): cycles per phase, per workgroup
};

#ifdef COMBAT_STAMPS
static unsigned long long *g_stamps_wgrad_host = nullptr;
extern "C" int combat_debug_set_stamps_wgrad(void *p) { g_stamps_wgrad_host = (unsigned long long *)p; return 0; }
#define WCLK() __builtin_readcyclecounter()
#else
#define WCLK() 0ull
#endif

typedef __attribute__((address_space(3))) unsigned char lds_u8;
// p: an LDS pointer held in a VGPR plus a compile-time byte offset -> "ds_read_b64_tr_b16 v, v offset:imm"
__device__ __forceinline__ s16x4_t tr16d(const lds_u8 *p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t *)(const_cast<lds_u8 *>(p)));
}

__device__ __forceinline__ bf16x8_t join8(const s16x4_t lo, const s16x4_t hi) {
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

// position key of a halo row (line, column) / of a dy row (pixel index): see the file comment
__device__ __forceinline__ int xkey(int TW, int line, int hx) {
    return ((hx >> 1) & 1) | ((TW == 4 ? (line >> 1) & 1 : ((hx >> 3) + line) & 1) << 1);
}
__device__ __forceinline__ int dkey(int pk) { return ((pk >> 1) & 1) | (((pk >> 3) & 1) << 1); }

template <int N>
__device__ __forceinline__ void wait_vm_lgkm0_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// dw += sum over the pixel ranges' slabs, by the calling workgroup for ITS share of the register quads, without
// atomics: the quads of the [base][9][1024] partial-tile space (fragment order, as the kernel below stores them) are
// dealt to the workgroups in contiguous runs; inside a workgroup thread (slab lane sl, quad ql) adds slabs sl, sl + SL,
// ... in that order, the SL partial sums meet in LDS and are added in lane order by the thread that owns the quad's four
// dw elements.  Every element is written by exactly one thread and every sum has a fixed order: the weight gradient
// is bit-reproducible.  NT threads; `scratch`: NT x 64 bytes of LDS.
// Quad f4 of a tap = [dy fragment i][wave][lane]: rows (i * 2 + (wave & 1)) * 16 + (lane >> 4) * 4 + 0..3 of column
// (wave >> 1) * 16 + (lane & 15).
template <int NT>
__device__ __forceinline__ void reduce_slabs(const float *__restrict__ ws, float *__restrict__ dw, int base, int tiles_c, int split,
                                             int k_real, int c_real, unsigned char *scratch, int block, int nblocks) {
    constexpr int U = 4;                                           // quads per thread and pass: 4 x 4 loads in flight
    const int tid = threadIdx.x;
    const long e4 = (long)base * 9 * 1024;
    const long per = (e4 + nblocks - 1) / nblocks;
    const long q0 = (long)block * per, q1 = q0 + per < e4 ? q0 + per : e4;
    // slab lanes: as many as still let ONE pass cover the workgroup's quads (passes are serial, each a memory round trip)
    int SL = 1;
    while (SL * 2 <= split && SL < 32 && (long)(NT / (SL * 2)) * U >= per) SL *= 2;
    const int QP = NT / SL, sl = tid / QP, ql = tid - sl * QP;
    const int nsl = (split - sl + SL - 1) / SL;                     // slabs of this lane: sl, sl + SL, ...
    f32x4_t *scr = reinterpret_cast<f32x4_t *>(scratch);
    const size_t sstride = (size_t)base * 9 * 4096;
    for (long qb = q0; qb < q1; qb += (long)QP * U) {
        f32x4_t acc[U];
        long qq[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            qq[u] = qb + (long)u * QP + ql;
        }
        for (int i = 0; i < nsl; i += 4) {
            f32x4_t v[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = qq[u] < q1 && i + j < nsl;
                    const float *src = ws + (ok ? qq[u] : q0) * 4 + (size_t)(ok ? sl + (i + j) * SL : 0) * sstride;
                    const f32x4_t x = *reinterpret_cast<const f32x4_t *>(src);
                    v[u][j] = ok ? x : f32x4_t{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
        }
        if (SL > 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) scr[tid * U + u] = acc[u];
            __syncthreads();
            if (sl == 0)
                for (int l = 1; l < SL; ++l)
#pragma unroll
                    for (int u = 0; u < U; ++u) acc[u] += scr[(l * QP + ql) * U + u];
        }
        if (sl == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long q = qq[u];
                if (q >= q1) continue;
                const int f4 = (int)(q & 1023), tap = (int)((q >> 10) % 9), tile = (int)(q / (9 * 1024));
                const int i = f4 >> 9, wave = (f4 >> 6) & 7, lane = f4 & 63;
                const int row = (i * 2 + (wave & 1)) * 16 + (lane >> 4) * 4, col = (wave >> 1) * 16 + (lane & 15);
                const int tile_c = tile % tiles_c, tile_k = tile / tiles_c;
                const int n = tile_k * 64 + row, c = tile_c * 64 + col;
                if (c < c_real) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < k_real) dw[((size_t)(n + e) * 9 + tap) * c_real + c] += acc[u][e];
                }
            }
        }
        if (SL > 1) __syncthreads();
    }
}

// HPW = x halo pieces (8 rows each) per wave (compile time: the counted vmcnt waits need the DMA count
// of a stage), NS = stages in the LDS ring (NS - 1 patches are in flight ahead of the one being consumed)
// Where a patch's 2 600 cycles go (per-wave in-kernel stamps, 16 patches per workgroup): all eight waves issue their
// DMA pieces (400), then compute -- the older wave of each SIMD finishes its 36 MFMAs in ~1 500 cycles, the younger one
// ~400 later (oldest-first arbitration) -- then everyone meets at the barrier: 1 152 cycles of MFMA per SIMD in
// 2 600.  Without the DMA (ablation build) a patch takes 1 660.  Tried: the two waves of a SIMD half a patch apart (waves
// 0-3 issue patch i + 2 then compute patch i while waves 4-7 compute patch i - 1 then issue; four-stage ring): the
// loop 44.8 k -> 38.2 k cycles, the launch and the step unchanged -- the 96-KB ring and 226 registers keep the
// input-gradient chain's workgroups off the CU (72 KB and 162 registers leave them room), and inside the step that
// sharing, not this loop, sets the pace.
// RED: the launch first folds an earlier launch's slabs (reduce_first): its own instantiation, so that the default
// kernels keep their scalar-register budget (with the fold compiled in they spilled nine SGPRs)
template <int HPW, int NS, bool RED>
__device__ __forceinline__ void wgrad3x3_dma_body(const W3dParams &p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const combat_wgrad_args &a = p.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_k = wid & 1, wave_c = wid >> 1;          // 2 x 4 waves over (n, c)
    // ---- first: this workgroup's share of the PREVIOUS weight gradient's slab reduction (same queue: that launch has
    // completed).  It used to be a launch of its own behind every weight gradient -- 24 per step, ~11 us each on the
    // auxiliary queue; here it is ~1/128 of the slabs per workgroup (147 KB, mostly from L2 / the Infinity Cache)
    // before the first patch is asked for.
    if (RED && p.red_ws) {
        reduce_slabs<512>(p.red_ws, p.red_dw, p.red_base, p.red_tiles_c, p.red_split, p.red_k_real, p.red_c_real, smem, blockIdx.x, gridDim.x);
        __syncthreads();       // (the scratch is the first DMA stage)
    }
    int bid = blockIdx.x;
    const int tile_c = bid % p.tiles_c; bid /= p.tiles_c;
    const int tile_k = bid % p.tiles_k;
    const int sp = bid / p.tiles_k;
    const int k0 = tile_k * 64, c0 = tile_c * 64;
    const int t_begin = sp * p.per;
    int t_end = t_begin + p.per;
    if (t_end > p.ntiles) t_end = p.ntiles;
    if (t_begin >= t_end) return;

    const int C = a.C, K = a.K, H = a.H, W = a.W;
    // LDS: [x halo patches of the NS slots | dy tiles of the NS slots].  Grouped by operand so that the slot offset of a
    // fragment read folds into its 16-bit immediate: four 16-KB x slots span exactly 64 KB -> ONE set of address registers.
    constexpr int XBYTES = HPW * 8192, DYBYTES = 64 * 128, DYBASE = NS * XBYTES;
    constexpr int NDMA = 1 + HPW;                           // DMA instructions per wave and stage
    // The LDS-DMA is issued from inline asm (resource words in SGPRs, LDS address through M0).  With the builtin, the
    // compiler guards the first transposing LDS read behind a DMA it knows of with `s_waitcnt vmcnt(0)`: every patch's
    // MFMAs then waited for the patch issued just before them -- two patches ahead in the ring -- to LAND, i.e. the ring
    // never overlapped anything (2 600 cycles per patch for 1 152 of MFMA).  Unknown to the compiler, the DMA is ordered
    // by the counted waits below alone, as designed.  (COMBAT_WGRAD_BUILTIN_DMA: the old form, for A/B timing.)
    auto rsrc_words = [&](const void *ptr, unsigned bytes) __attribute__((always_inline)) {
        const unsigned long base = (unsigned long)ptr;
        u32x4_t w;
        w[0] = __builtin_amdgcn_readfirstlane((unsigned)base);
        w[1] = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32) & 0xffffu);
        w[2] = __builtin_amdgcn_readfirstlane(bytes);
        w[3] = 0x00020000u;
        return w;
    };
    const u32x4_t xrs = rsrc_words(a.src, p.x_bytes), dyrs = rsrc_words(a.dy, p.dy_bytes);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lds_void_t *)smem) + wid * 1024;
#ifdef COMBAT_WGRAD_BUILTIN_DMA
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.src), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dyrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.dy), 0, p.dy_bytes, 0x00020000);
#define COMBAT_W3_DMA(rs_builtin, rs_words, lds_off, voff) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_builtin, (lds_void_t *)(smem + (lds_off) + wid * 1024), 16, voff, 0, 0, 0)
#else
#define COMBAT_W3_DMA(rs_builtin, rs_words, lds_off, voff) \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds0_l + (unsigned)(lds_off)), "v"(voff), "s"(rs_words##_l) : "memory", "m0")
#endif

    // ---- DMA bookkeeping that does not depend on the patch: this lane's row of every piece it issues
    // (decoded position for the border tests, byte offset relative to the patch origin with the channel
    // chunk its LDS slot must receive)
    int hdec[HPW], hrel[HPW];   // x pieces: hx | hy << 8 | ti << 16, or -1 (row outside the halo patch)
#pragma unroll
    for (int j = 0; j < HPW; ++j) {
        const int row = (wid + 8 * j) * 8 + (lane >> 3), slot = lane & 7;
        const int hx = row % p.HWP, line = row / p.HWP;
        const int hy = line % p.HH, ti = line / p.HH;
        const int chunk = (((slot >> 1) ^ xkey(p.TW, line, hx)) << 1) | (slot & 1);
        hdec[j] = row < p.HROWS ? (hx | (hy << 8) | (ti << 16)) : -1;
        hrel[j] = (((ti * H + hy - 1) * W + hx - 1) * C + c0 + chunk * 8) * 2;
        asm volatile("" : "+v"(hdec[j]), "+v"(hrel[j]));
    }
    int dti[1], drel[1];        // dy piece: image of the patch, byte offset relative to the patch origin
#pragma unroll
    for (int j = 0; j < 1; ++j) {
        const int pk = wid * 8 + (lane >> 3), slot = lane & 7;
        const int tx = pk & (p.TW - 1), ty = (pk >> p.tw_shift) & (p.TH - 1), ti = pk >> (p.tw_shift + p.th_shift);
        const int chunk = (((slot >> 1) ^ dkey(pk)) << 1) | (slot & 1);
        dti[j] = ti;
        drel[j] = (((ti * H + ty) * W + tx) * K + k0 + chunk * 8) * 2;
        asm volatile("" : "+v"(dti[j]), "+v"(drel[j]));
    }
    // A patch's DMA = its origin (wave-uniform: image group, tile row, tile column as running counters -- the three
    // divisions of the patch index were 300 of a patch's 2800 cycles) + NDMA pieces.
    int nt_x, nt_y, nt_g;                                     // tile coordinates of the NEXT patch to issue
    {
        const int t = t_begin;
        nt_x = t % p.tiles_x;
        nt_y = (t / p.tiles_x) % p.tiles_y;
        nt_g = t / (p.tiles_x * p.tiles_y);
    }
    int is_img0 = 0, is_oy0 = 0, is_ox0 = 0, is_xorg = 0, is_dorg = 0;
    auto issue_prep = [&]() __attribute__((always_inline)) {  // origin of the next patch; advances the counters
        is_img0 = nt_g * p.TI;
        is_oy0 = nt_y * p.TH;
        is_ox0 = nt_x * p.TW;
        const int origin = (is_img0 * H + is_oy0) * W + is_ox0;
        is_xorg = origin * C * 2;
        is_dorg = origin * K * 2;
        if (++nt_x == p.tiles_x) {
            nt_x = 0;
            if (++nt_y == p.tiles_y) {
                nt_y = 0;
                ++nt_g;
            }
        }
    };
    auto issue_piece = [&](auto j_tag, auto slot_tag) __attribute__((always_inline)) {
        constexpr int j = decltype(j_tag)::value;
        constexpr int slot_i = decltype(slot_tag)::value;
        // (named copies: an asm operand inside a generic lambda does not capture the enclosing function's variable)
        const u32x4_t xrs_l = xrs, dyrs_l = dyrs;
        const unsigned lds0_l = lds0;
        (void)xrs_l; (void)dyrs_l; (void)lds0_l;
        if constexpr (j == 0) {
            const unsigned off = is_img0 + dti[0] < a.N ? (unsigned)(is_dorg + drel[0]) : kOob;
            COMBAT_W3_DMA(dyrsrc, dyrs, DYBASE + slot_i * DYBYTES, off);
        } else if constexpr (j <= HPW) {
            const int d = hdec[j - 1];
            const bool ok = d >= 0 && is_img0 + (d >> 16) < a.N && (unsigned)(is_oy0 + ((d >> 8) & 255) - 1) < (unsigned)H &&
                            (unsigned)(is_ox0 + (d & 255) - 1) < (unsigned)W;
            const unsigned off = ok ? (unsigned)(is_xorg + hrel[j - 1]) : kOob;
            COMBAT_W3_DMA(xrsrc, xrs, slot_i * XBYTES + 8 * (j - 1) * 1024, off);
        }
    };
    auto issue = [&](auto slot_tag) __attribute__((always_inline)) {   // a whole patch at once (the first NS - 1)
        using std::integral_constant;
        issue_prep();
        issue_piece(integral_constant<int, 0>{}, slot_tag);
        issue_piece(integral_constant<int, 1>{}, slot_tag);
        issue_piece(integral_constant<int, 2>{}, slot_tag);
        issue_piece(integral_constant<int, 3>{}, slot_tag);
        issue_piece(integral_constant<int, 4>{}, slot_tag);
    };

    // ---- per-lane fragment addresses (relative to the stage).  Transposing read: lane 4q+pp of a
    // 16-lane group addresses pixel row q (4 rows per read), channels 4pp..4pp+3 of the fragment's 16,
    // and receives the 4 pixels of channel (lane & 15).
    const int q = (lane & 15) >> 2, pp = lane & 3, fq = lane >> 4;
    // LDS read addresses as pointers in VGPRs: a slot's stage base is a compile-time offset that folds into the read's
    // 16-bit immediate, so one register set serves SPS = 64 KB / stage slots (the compiler, left alone, kept one set of
    // 44 address registers PER SLOT: 132 of the kernel's 250 VGPRs at three slots, spills at four)
    constexpr int SPS = 65536 / XBYTES, NSETS = (NS + SPS - 1) / SPS;   // x slots per address-register set
    const lds_u8 *const smem_l = (const lds_u8 *)smem;
    const lds_u8 *dyp[2][2][2], *xp[NSETS][2][2][9];   // [k-step][lo/hi 4-pixel group][dy fragment] / [set][k-step][lo/hi][tap]
    int dya[2][2], xa[2][2][9];                             // [k-step][lo/hi 4-pixel group]([tap]); dy fragment 1 = ^ 64
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
            const int pk = ks * 32 + fq * 8 + q + 4 * hi;   // pixel of the 64-pixel tile
            dya[ks][hi] = pk * 128 + ((wave_k ^ dkey(pk)) & 3) * 32 + pp * 8;
            const int tx = pk & (p.TW - 1), ty = (pk >> p.tw_shift) & (p.TH - 1), ti = pk >> (p.tw_shift + p.th_shift);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int r = (tap * 11) >> 5, s = tap - 3 * r;
                const int line = ti * p.HH + ty + r, hx = tx + s;
                xa[ks][hi][tap] = (line * p.HWP + hx) * 128 + ((wave_c ^ xkey(p.TW, line, hx)) & 3) * 32 + pp * 8;
                // pin the value: left alone the compiler re-derives all 36 addresses inside the patch loop
                // (~460 VALU instructions per patch) instead of keeping them in registers
#pragma unroll
                for (int set = 0; set < NSETS; ++set) {
                    xp[set][ks][hi][tap] = smem_l + set * SPS * XBYTES + xa[ks][hi][tap];
                    asm volatile("" : "+v"(xp[set][ks][hi][tap]));
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                dyp[ks][hi][i] = smem_l + DYBASE + (dya[ks][hi] ^ (i * 64));
                asm volatile("" : "+v"(dyp[ks][hi][i]));
            }
        }

    f32x4_t acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[t][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- MFMAs of one staged patch: 9 taps x 2 k-steps x 2 dy fragments.  The dy fragments of both
    // k-steps are read once per patch; the x fragments of tap t+1 are in flight while tap t's MFMAs run.
    auto x_frags = [&](bf16x8_t (&fx)[2], auto slot_tag, auto tap_tag) __attribute__((always_inline)) {
        constexpr int tap = decltype(tap_tag)::value, slot = decltype(slot_tag)::value;
        constexpr int set = slot / SPS, imm = (slot % SPS) * XBYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fx[ks] = join8(tr16d(xp[set][ks][0][tap] + imm), tr16d(xp[set][ks][1][tap] + imm));
#ifdef COMBAT_NO_INTERLEAVE
        __builtin_amdgcn_sched_barrier(0);
#endif
    };
    auto mfma_tap = [&](auto tap_tag, const bf16x8_t (&fk)[2][2], const bf16x8_t (&fx)[2]) __attribute__((always_inline)) {
        constexpr int tap = decltype(tap_tag)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[tap][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[ks][i], fx[ks], acc[tap][i], 0, 0, 0);
#ifndef COMBAT_NO_INTERLEAVE
        // the next tap's four transposing reads (issued just before, into the other register pair) go one per MFMA
        // gap: an MFMA holds the SIMD's vector issue for 8 of its 16 cycles (conv3x3_dma.hip, mfma_frags)
        if (tap < 8) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
    };
    auto compute = [&](auto slot_tag, auto nslot_tag, bool ahead) __attribute__((always_inline)) {
        constexpr int slot = decltype(slot_tag)::value, imm = slot * DYBYTES;
        bf16x8_t fk[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                fk[ks][i] = join8(tr16d(dyp[ks][0][i] + imm), tr16d(dyp[ks][1][i] + imm));
        using std::integral_constant;
        // tap t: fragments of tap t + 1 -> other register set | MFMAs of tap t
        // (tried: one DMA piece of patch i + NS - 1 behind every second tap block instead of all of them in front of
        // the patch -- in-kernel stamps: compute 1452 -> 2513 cycles per patch for 524 cycles of issue removed, a DMA
        // issue between two MFMA blocks stalls both waves of the SIMD; fragments three taps ahead instead of one --
        // compute 1506 -> 1420, but the 24 extra registers made the compiler re-derive the DMA offsets per patch:
        // issue 408 -> 2576)
        (void)nslot_tag;
        (void)ahead;
        bf16x8_t fx0[2], fx1[2];
        x_frags(fx0, slot_tag, integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
#define COMBAT_W3_TAP(t, cur, nxt)                                                   \
        if (t < 8) x_frags(nxt, slot_tag, integral_constant<int, (t < 8 ? t + 1 : 8)>{});  \
        mfma_tap(integral_constant<int, t>{}, fk, cur);
        COMBAT_W3_TAP(0, fx0, fx1) COMBAT_W3_TAP(1, fx1, fx0) COMBAT_W3_TAP(2, fx0, fx1) COMBAT_W3_TAP(3, fx1, fx0)
        COMBAT_W3_TAP(4, fx0, fx1) COMBAT_W3_TAP(5, fx1, fx0) COMBAT_W3_TAP(6, fx0, fx1) COMBAT_W3_TAP(7, fx1, fx0)
        COMBAT_W3_TAP(8, fx0, fx1)
#undef COMBAT_W3_TAP
    };
    static_assert(HPW <= 4, "issue() spells out 1 + HPW <= 5 pieces");

    // ---- ring of NS stages: patch i of this workgroup lives in slot i % NS; NS - 1 patches are DMA'd
    // ahead.  The loop is unrolled over the slots so that every LDS address is "register + immediate".
    // At the top of iteration i every wave has passed the barrier that followed compute(i - 1), so
    // slot (i - 1) % NS is free for patch i + NS - 1.  The wait after compute(i) needs patch i + 1 only:
    // everything younger (NS - 2 patches) may stay in flight.
    const int npatch = t_end - t_begin;
    using std::integral_constant;
    unsigned long long ph[4] = {0, 0, 0, 0};
    unsigned long long c_loop0 = 0;
    {
    issue(integral_constant<int, 0>{});
    if (NS > 2 && npatch > 1) issue(integral_constant<int, 1 % NS>{});
    if (NS > 2 && npatch > 1) wait_vm_lgkm0_barrier<NDMA>(); else wait_vm_lgkm0_barrier<0>();
    c_loop0 = WCLK();
    auto body = [&](auto slot_tag, int i) __attribute__((always_inline)) {
        constexpr int slot = decltype(slot_tag)::value, nslot = (slot + NS - 1) % NS;
        const bool ahead = i + NS - 1 < npatch;
        const unsigned long long c0 = WCLK();
        if (ahead) issue(integral_constant<int, nslot>{});
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long c1 = WCLK();
        compute(integral_constant<int, slot>{}, integral_constant<int, nslot>{}, ahead);
        const unsigned long long c2 = WCLK();
        if (ahead) wait_vm_lgkm0_barrier<(NS - 2) * NDMA>(); else wait_vm_lgkm0_barrier<0>();
#ifdef COMBAT_STAMPS
        const unsigned long long c3 = WCLK();
        ph[0] += c1 - c0; ph[1] += c2 - c1; ph[2] += c3 - c2; ph[3] += 1;
#else
        (void)c0; (void)c1; (void)c2;
#endif
    };
    for (int i = 0; i < npatch; i += NS) {
        body(integral_constant<int, 0>{}, i);
        if (NS > 1 && i + 1 < npatch) body(integral_constant<int, 1 % NS>{}, i + 1);
        if (NS > 2 && i + 2 < npatch) body(integral_constant<int, 2 % NS>{}, i + 2);
    }
    }
    static_assert(NS == 2 || NS == 3, "the patch loop is unrolled for rings of two or three stages");
#ifdef COMBAT_STAMPS
    const unsigned long long c_loop1 = WCLK();
#endif

    // ---- nine [64 n][64 c] partials.  With a workspace: this workgroup's slab in FRAGMENT order -- per tap
    // [dy fragment i][wave][lane] x 4 floats, i.e. every accumulator register quad as one 16-byte store, 1 KB per
    // wave instruction, no LDS pass and no barrier (the row-major form cost 9 x (LDS image + two barriers) = 9.5 k of a
    // workgroup's 57 k cycles); wgrad3x3_reduce_kernel maps a quad back to its (row, column).
    if (p.ws) {
        float *slab = p.ws + (size_t)blockIdx.x * 9 * 4096;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                *reinterpret_cast<f32x4_t *>(slab + tap * 4096 + ((i * 8 + wid) * 64 + lane) * 4) = acc[tap][i];
    } else {
        // without one: fp32 LDS image per tap -> atomics in 256-byte runs along c
        float *ep = reinterpret_cast<float *>(smem);
        constexpr int EPS = 68;
        const int fr = lane & 15;
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int col = wave_c * 16 + fr;
                const int row = (i * 2 + wave_k) * 16 + fq * 4;
                f32x4_t v;
                switch (tap) {   // acc is indexed statically (a runtime index would put it in scratch)
                    case 0: v = acc[0][i]; break;
                    case 1: v = acc[1][i]; break;
                    case 2: v = acc[2][i]; break;
                    case 3: v = acc[3][i]; break;
                    case 4: v = acc[4][i]; break;
                    case 5: v = acc[5][i]; break;
                    case 6: v = acc[6][i]; break;
                    case 7: v = acc[7][i]; break;
                    default: v = acc[8][i]; break;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) ep[(row + e) * EPS + col] = v[e];
            }
            __syncthreads();
            for (int idx = tid; idx < 64 * 64; idx += 512) {
                const int row = idx >> 6, col = idx & 63;
                const int n = k0 + row, c = c0 + col;
                if (n < a.k_real && c < a.c_real) atomicAdd(a.dw + ((size_t)n * 9 + tap) * a.c_real + c, ep[row * EPS + col]);
            }
            __syncthreads();
        }
    }
#ifdef COMBAT_STAMPS
    if (lane == 0 && p.stamps) {     // one record per wave
        unsigned long long *o = p.stamps + (blockIdx.x * 8 + wid) * 8;
        o[0] = ph[0]; o[1] = ph[1]; o[2] = ph[2]; o[3] = ph[3];
        o[4] = c_loop1 - c_loop0; o[5] = WCLK() - c_loop1; o[6] = c_loop0;
    }
#endif
}

// dw += sum over the pixel ranges' slabs.  grid.x: register quads of the [tiles][9][1024] partial tile space in the
// kernel's fragment order (256 threads each), grid.y: groups of ranges (each adds its share with one fp32 atomic
// per element -- a few per element instead of `split`).  Quad f4 of a tap = [dy fragment i][wave][lane]:
// rows (i * 2 + (wave & 1)) * 16 + (lane >> 4) * 4 + 0..3 of column (wave >> 1) * 16 + (lane & 15).
__global__ __launch_bounds__(256) void wgrad3x3_reduce_kernel(const float *__restrict__ ws, float *__restrict__ dw,
                                                              int base, int tiles_c, int split, int k_real, int c_real) {
    const long e4 = (long)blockIdx.x * 256 + threadIdx.x;          // quad index inside [base][9][1024]
    if (e4 >= (long)base * 9 * 1024) return;
    const int f4 = (int)(e4 & 1023), tap = (int)((e4 >> 10) % 9), tile = (int)(e4 / (9 * 1024));
    const int per = (split + gridDim.y - 1) / gridDim.y;
    const int s0 = blockIdx.y * per, s1 = s0 + per < split ? s0 + per : split;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    const float *src = ws + ((size_t)tile * 9 + tap) * 4096 + f4 * 4;
    const size_t sstride = (size_t)base * 9 * 4096;
    int s = s0;
    for (; s + 4 <= s1; s += 4) {   // four independent loads in flight
        const f32x4_t v0 = *reinterpret_cast<const f32x4_t *>(src + (size_t)s * sstride);
        const f32x4_t v1 = *reinterpret_cast<const f32x4_t *>(src + (size_t)(s + 1) * sstride);
        const f32x4_t v2 = *reinterpret_cast<const f32x4_t *>(src + (size_t)(s + 2) * sstride);
        const f32x4_t v3 = *reinterpret_cast<const f32x4_t *>(src + (size_t)(s + 3) * sstride);
        acc += (v0 + v1) + (v2 + v3);
    }
    for (; s < s1; ++s) acc += *reinterpret_cast<const f32x4_t *>(src + (size_t)s * sstride);
    const int i = f4 >> 9, wave = (f4 >> 6) & 7, lane = f4 & 63;
    const int row = (i * 2 + (wave & 1)) * 16 + (lane >> 4) * 4, col = (wave >> 1) * 16 + (lane & 15);
    const int tile_c = tile % tiles_c, tile_k = tile / tiles_c;
    const int n = tile_k * 64 + row, c = tile_c * 64 + col;
    if (c < c_real && s0 < s1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (n + e >= k_real) break;
            float *o = dw + ((size_t)(n + e) * 9 + tap) * c_real + c;
            if (gridDim.y == 1) *o += acc[e];     // sole owner of these elements: no atomics
            else atomicAdd(o, acc[e]);
        }
    }
}

// stand-alone form of reduce_slabs (the last weight gradient of a queue has no successor to ride with; launches
// without defer_reduce): 256 workgroups of 256 threads
__global__ __launch_bounds__(256) void wgrad3x3_reduce_det_kernel(const float *__restrict__ ws, float *__restrict__ dw, int base,
                                                                  int tiles_c, int split, int k_real, int c_real) {
    __shared__ __attribute__((aligned(16))) unsigned char scratch[256 * 64];
    reduce_slabs<256>(ws, dw, base, tiles_c, split, k_real, c_real, scratch, blockIdx.x, gridDim.x);
}

// (the body is a device function: the host pass cannot see the buffer-resource type it uses)
template <int HPW, int NS, bool RED>
__global__ __launch_bounds__(512, 1) void conv_wgrad3x3_dma_kernel(const W3dParams p) { wgrad3x3_dma_body<HPW, NS, RED>(p); }

template <int HPW, int NS, bool RED>
int launch_w3d_red(const W3dParams &p, int blocks, hipStream_t st) {
    constexpr int stage = 8192 + HPW * 8192;
    constexpr int ep = 64 * 68 * 4;
    constexpr int smem = NS * stage > ep ? NS * stage : ep;
    auto kern = conv_wgrad3x3_dma_kernel<HPW, NS, RED>;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return COMBAT_ELAUNCH;
        attr = true;
    }
    COMBAT_LAUNCH(kern, dim3(blocks), dim3(512), smem, st, p);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

template <int HPW, int NS>
int launch_w3d(const W3dParams &p, int blocks, hipStream_t st) {
    return p.red_ws ? launch_w3d_red<HPW, NS, true>(p, blocks, st) : launch_w3d_red<HPW, NS, false>(p, blocks, st);
}

bool w3d_geometry(const combat_wgrad_args *a, W3dParams &p, int &smem) {
    if (!(a->R == 3 && a->S == 3 && a->stride == 1 && a->pad == 1 && a->P == a->H && a->Q == a->W)) return false;
    if (a->pro_scale || a->pro_act) return false;   // operands go straight to LDS
    if (a->C < 64 || (a->C & 63) || a->K < 64 || (a->K & 63) || a->c_real != a->C) return false;
    const int W = a->W, H = a->H;
    const int TW = W < 16 ? W : 16;
    const int TH = H < 64 / TW ? H : 64 / TW;
    const int TI = 64 / (TW * TH);
    if (TW < 2 || TI < 1 || TW * TH * TI != 64) return false;
    p.a = *a;
    p.TW = TW; p.TH = TH; p.TI = TI;
    p.HWP = TW + 2; p.HH = TH + 2; p.HROWS = TI * p.HH * p.HWP;
    p.hpw = (p.HROWS + 63) / 64;   // 8-row pieces per wave, eight waves
    p.tw_shift = ilog2_exact(TW); p.th_shift = ilog2_exact(TH);
    if (p.tw_shift < 0 || p.th_shift < 0 || W % TW || H % TH) return false;
    const long xb = (long)a->N * H * W * a->C * 2, db = (long)a->N * H * W * a->K * 2;
    if (xb >= (long)kOob || db >= (long)kOob) return false;
    p.x_bytes = (unsigned)xb; p.dy_bytes = (unsigned)db;
    p.tiles_x = W / TW; p.tiles_y = H / TH;
    p.ntiles = p.tiles_x * p.tiles_y * ((a->N + TI - 1) / TI);
    p.tiles_k = a->K / 64; p.tiles_c = a->C / 64;
    smem = 0;
    return p.hpw >= 2 && p.hpw <= 4;
}

// number of pixel ranges: one workgroup per (tile, range), >= 4 patches each.  With a workspace the
// partial sums cost plain stores (147 KB per workgroup, read back by the reduction launch) and the
// launch is sized for HALF the CUs: weight gradients run on an auxiliary stream beside the
// input-gradient chain, and 128 long-running 512-thread workgroups leave that chain room while
// halving the slab traffic (measured on the whole step: 96-128 workgroups 5.07 ms, 192: 5.18,
// 256: 5.30, 384: 5.81, 64: 5.14, 32: 5.69).  Without a workspace: fp32 atomics against a ~24 MB budget.
int pick_split(const combat_wgrad_args *a, const W3dParams &p, bool with_ws) {
    const int base = p.tiles_k * p.tiles_c;
    int split = a->split;
    if (split <= 0) {
        static const int ws_wgs = getenv("COMBAT_WGRAD_WGS") ? atoi(getenv("COMBAT_WGRAD_WGS")) : 128;   // (experiments)
        const int by_cus = ((with_ws ? ws_wgs : 256) + base - 1) / base;
        const int by_atomics = 164 / base > 0 ? 164 / base : 1;
        split = with_ws || by_cus < by_atomics ? by_cus : by_atomics;
        const int by_work = (p.ntiles + 3) / 4;
        if (split > by_work) split = by_work;
        if (split < 1) split = 1;
    }
    if (split > p.ntiles) split = p.ntiles;
    const int per = (p.ntiles + split - 1) / split;
    return (p.ntiles + per - 1) / per;
}

}  // namespace

long conv_wgrad3x3_dma_workspace(const combat_wgrad_args *a) {
    W3dParams p;
    int smem;
    if (!w3d_geometry(a, p, smem)) return 0;
    const int split = pick_split(a, p, true);
    return split > 1 ? (long)split * p.tiles_k * p.tiles_c * 9 * 4096 * 4 : 0;
}

namespace {
// geometry, pixel ranges and workspace use of a launch: ONE place, so that the deferred reduction walks exactly the
// slabs its kernel launch wrote
bool w3d_plan(const combat_wgrad_args *a, W3dParams &p) {
    int smem;
    if (!w3d_geometry(a, p, smem)) return false;
    const int base = p.tiles_k * p.tiles_c;
    int split = pick_split(a, p, true);
    const bool use_ws = split > 1 && a->workspace && a->workspace_bytes >= (long)split * base * 9 * 4096 * 4;
    if (!use_ws) split = pick_split(a, p, false);
    p.ws = use_ws ? reinterpret_cast<float *>(a->workspace) : nullptr;
    p.per = (p.ntiles + split - 1) / split;
    p.split = (p.ntiles + p.per - 1) / p.per;
    return true;
}

int launch_reduce(const combat_wgrad_args *a, const W3dParams &p, hipStream_t st) {
    const int base = p.tiles_k * p.tiles_c;
    // Default: the round-3 reduction (ranges dealt to <= 14 groups of workgroups, one fp32 atomic per element and group):
    // 4-8 us back to back.  COMBAT_WGRAD_DET_REDUCE=1: reduce_slabs() as a launch of its own -- fixed summation order, no
    // atomics, bit-reproducible -- 9-18 us (tools/wgrad_chain_bench.py): a workgroup owns whole elements, so the slabs
    // of an element cannot be spread over the chip.
    static const bool env_det = getenv("COMBAT_WGRAD_DET_REDUCE") != nullptr;
    const bool det = combat_deterministic();
    // deterministic mode: with >= 4 tiles (144 workgroups of 256 quads) ONE group of ranges fills enough of the chip
    // and owns its elements (`dw += sum`, ranges in ascending order, no atomics): the round-3 kernel as it is; the
    // 64-channel layers (one tile, up to 256 ranges) take reduce_slabs(), which spreads an element's slabs over lanes
    const bool atomics = !env_det && (!det || base >= 4);
    if (atomics) {
        // enough workgroups to fill the chip; each group of ranges costs one fp32 atomic per element
        const long e4 = (long)base * 9 * 1024;
        int groups = (int)(512 / ((e4 + 255) / 256));
        if (groups < 1 || det) groups = 1;
        if (groups > p.split) groups = p.split;
        COMBAT_LAUNCH(wgrad3x3_reduce_kernel, dim3((unsigned)((e4 + 255) / 256), groups), dim3(256), 0, st, p.ws, a->dw, base,
                           p.tiles_c, p.split, a->k_real, a->c_real);
    } else {
        COMBAT_LAUNCH(wgrad3x3_reduce_det_kernel, dim3(256), dim3(256), 0, st, (const float *)p.ws, a->dw, base, p.tiles_c, p.split,
                           a->k_real, a->c_real);
    }
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

// `prev`: a launch recorded with defer_reduce whose slabs this launch folds into prev->dw first (reduce_first)
bool fill_reduce_first(const combat_wgrad_args *prev, W3dParams &p) {
    p.red_ws = nullptr;
    p.red_dw = nullptr;
    p.red_base = p.red_tiles_c = p.red_split = p.red_k_real = p.red_c_real = p.red_pad = 0;
    if (!prev) return true;
    W3dParams q;
    if (!w3d_plan(prev, q)) return true;     // (not a launch of this kernel: the generic kernels reduce their own slabs)
    if (!q.ws) return true;                  // (that launch needed no reduction)
    p.red_ws = q.ws;
    p.red_dw = prev->dw;
    p.red_base = q.tiles_k * q.tiles_c;
    p.red_tiles_c = q.tiles_c;
    p.red_split = q.split;
    p.red_k_real = prev->k_real;
    p.red_c_real = prev->c_real;
    return true;
}
}  // namespace

// returns COMBAT_OK if launched, 1 if this kernel does not apply (caller falls back), <0 on error
int conv_wgrad3x3_dma_try(const combat_wgrad_args *a, hipStream_t st) {
    W3dParams p;
    if (!w3d_plan(a, p)) return 1;
    // deterministic mode: several pixel ranges meet through slabs (the kernel's other epilogue adds every range's tile
    // to dw with fp32 atomics)
    if (combat_deterministic() && !p.ws && p.split > 1) return COMBAT_EINVAL;
    if (!fill_reduce_first(a->reduce_first, p)) return COMBAT_EINVAL;
#ifdef COMBAT_STAMPS
    p.stamps = g_stamps_wgrad_host;
#else
    p.stamps = nullptr;
#endif
    const int blocks = p.tiles_k * p.tiles_c * p.split;
    int rc;
    if (p.hpw == 2) rc = launch_w3d<2, 3>(p, blocks, st);        // 3 x 24 KB
    else if (p.hpw == 3) rc = launch_w3d<3, 2>(p, blocks, st);   // 2 x 32 KB
    else rc = launch_w3d<4, 2>(p, blocks, st);                   // 2 x 40 KB
    if (rc != COMBAT_OK || !p.ws || a->defer_reduce) return rc;
    return launch_reduce(a, p, st);
}

// the reduction a defer_reduce launch left out: COMBAT_OK (also when there is nothing to reduce), <0 on error
int conv_wgrad3x3_dma_reduce(const combat_wgrad_args *a, hipStream_t st) {
    W3dParams p;
    if (!w3d_plan(a, p) || !p.ws) return COMBAT_OK;
    return launch_reduce(a, p, st);
}
