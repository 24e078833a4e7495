// 3x3 / stride-1 / pad-1 convolution (forward and input-gradient) WITHOUT an input prologue, with both
// operands moved global -> LDS by the DMA path (buffer_load ... lds, 16 bytes per lane).
//
// The halo kernels of conv3x3.hip stage their operands through registers because the BatchNorm /
// InstanceNorm prologue has to touch every input element.  Convolutions whose input is already in
// its final form (every input-gradient pass; forwards whose producer applied the activation) need no
// VALU work on the way in, so here no operand byte ever visits a VGPR before its MFMA fragment read:
//   * per 64-channel chunk the (TH+2) x (TW+2) halo patch of the workgroup's 128 pixels is DMA'd
//     once; out-of-image pixels use an out-of-range buffer offset, which the hardware returns -- and
//     writes to LDS -- as zeros (the padding);
//   * per (chunk, tap) step the BN x 64 weight tile is DMA'd into a three-deep ring, two steps ahead
//     of its use; the only synchronisation per step is one counted s_waitcnt vmcnt + one s_barrier.
// An LDS-DMA instruction writes 64 lanes x 16 B contiguously, so rows cannot be padded; instead the
// eight 16-byte slots of every 128-byte row are ROTATED by (row & 6): slot = (chunk + (row & 6)) & 7,
// applied on the global-address side.  For any 16 consecutive rows this makes ds_read_b128 fragment
// reads bank-conflict-free (the same property the 160-byte row pitch gives the halo kernels).  The
// halo row pitch is a multiple of 8 pixels, so the rotation depends on the tap's column offset only
// and every fragment read is "precomputed VGPR + immediate".
//
// The fused epilogue (bias, residuals, activation mask, normalisation statistics: the same semantics
// as conv_common.hpp's) is wave-private: each wave transposes its 32 pixels x BN channels through LDS
// and its operand fetches are issued four steps before the main loop ends.
#include "conv_dma_epilogue.hpp"
#include <stdlib.h>

#ifdef COMBAT_STAMPS   // in-kernel phase stamps of profiling builds: the buffer pointer travels as a kernel argument
static unsigned long long *g_stamps_dma_host = nullptr;
extern "C" int combat_debug_set_stamps_dma(void *p) { g_stamps_dma_host = (unsigned long long *)p; return 0; }
#define DSTAMP(k)                                                                                        \
    do {                                                                                                 \
        if (threadIdx.x == 0 && p.stamps) p.stamps[blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); \
        if (threadIdx.x == 0 && p.stamps && ((k) == 0 || (k) == 4)) p.stamps[blockIdx.x * 16 + 8 + (k)] = wall_clock64(); \
    } while (0)
#else
#define DSTAMP(k)
#endif

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;

// Ablation switches for profiling builds (results are garbage; tools/conv_bench.py timings only):
//   COMBAT_ABL_NOMFMA  the MFMA blocks become one VALU op per fragment pair (fragment reads stay alive)
//   COMBAT_ABL_NODMA   no weight / halo DMA after the first halo patch (the counted waits shrink accordingly)
//   COMBAT_ABL_NOREAD  no LDS fragment reads (the MFMAs run on whatever the registers hold)
//   COMBAT_ABL_NOEPI   no epilogue (operand fetches still issued; nothing is stored)
#if defined(COMBAT_ABL_NODMA)
#define ABL_DMA(x) 0
#else
#define ABL_DMA(x) (x)
#endif

constexpr unsigned kOob = kDmaOob;

struct DmaParams {
    combat_conv_args a;
    int tiles_x, tiles_y, tiles_m, tiles_n;
    int m_fastest;               // order of an XCD's contiguous tile range (each XCD has its own L2)
    int PQ, nchunks;
    unsigned src_bytes, w_bytes;
    int flavour;                  // epilogue specialisation (conv_dma_epilogue.hpp), -1: generic
    int w_prefetch;               // warm the XCD's L2 with this channel tile's weight rows at kernel start (see the body)
    int per;                      // weight-stationary kernel: tiles per workgroup
    unsigned long long *stamps;   // profiling builds only
};

// geometry class of a tile of RPW * NW pixels (NW waves of RPW = 32 or 64 pixels), keyed by the tile width
template <int TW, int NW, int RPW = 32>
struct DGeo {
    static constexpr int BM = RPW * NW;
    static constexpr int TH = TW == 16 ? BM / 16 : (TW == 8 ? 8 : 4);
    static constexpr int TI = BM / (TW * TH);
    static constexpr int HH = TH + 2;
    // halo row pitch in pixels.  The slot rotation is keyed on the halo COLUMN (hx & 6), so a tap's row
    // offset (a multiple of the pitch) never changes it; the 16 pixels of a fragment lie in one halo
    // row (TW 16) or in rows whose equal columns fall into the same rotation class (TW 8), which keeps
    // ds_read_b128 conflict-free for any pitch.  TW 4 (pitch 8): a fragment is four rows of the same four
    // columns, so the key is (column bit 1, row parity) there -- with the column key alone every fragment read
    // was a 2-way bank conflict (LdsBankConflict 0.50 in the PMC pass, 0.03-0.06 on the other geometries).
    static constexpr int HWP = TW == 4 ? 8 : TW + 2;
    // WHOLE (4 x 4 and 8 x 8 maps): a tile is TI whole images, so its halo is nothing but padding, and the padding is
    // not DMA'd.  The LDS image holds each image's TW * TH pixels back to back (pitch TW), one image row of zeros
    // above the first image and below every image (the filter rows that leave the image read those), and a zone of
    // zeros behind the last one which the fragment reads of a filter COLUMN that leaves the image are pointed at
    // (fixed per lane and column offset, see `pa`; three rows one filter-row stride apart).  The zeros are written
    // once, when the kernel starts; the DMA pieces -- 8 rows = TW 8: one image row, TW 4: two -- carry data only:
    // 4 per wave and chunk where the padded form (rows of pitch 8 / 10 with the padding fetched as out-of-range
    // loads) issued 12 / 7, and 22 KB per image where that took 48 / 28 (TW 4: 109 KB per workgroup = one per CU and
    // no other queue's convolution beside it).  Measured against the padded form on one box: 512 -> 512 at 4 x 4
    // 16.3 -> 14.6 us back to back, 22.5 -> 20.7 cold; 256 -> 256 at 8 x 8 unchanged (14.6); the step 3.727 -> 3.699 ms.
    static constexpr bool WHOLE = TW == 4 || TW == 8;
    static constexpr int IPITCH = TW * TH + TW;                  // WHOLE: LDS rows from one image to the next
    static constexpr int ZONE = (TW + TI * IPITCH) * 128;        // WHOLE: byte offset of the zone of zeros
    static constexpr int HROWS = WHOLE ? BM : TI * HH * HWP;     // LDS rows (pixels) the DMA fills in one halo image
    static constexpr int HPW = (HROWS + 8 * NW - 1) / (8 * NW);  // 1-KiB DMA pieces (8 rows) per wave
    // (every wave issues the same number of pieces)
    static constexpr int HBYTES = WHOLE ? (ZONE + (2 * TW + 1) * 128 + 1023) / 1024 * 1024 : HPW * NW * 1024;
    // LDS row of row `sub` (0..7) of DMA piece k
    static __device__ __forceinline__ int piece_row(int k, int sub) {
        if constexpr (WHOLE) {
            const int d = k * 8 + sub;                           // data row: image * (TW * TH) + y * TW + x
            return TW + (d / (TW * TH)) * IPITCH + d % (TW * TH);
        } else {
            return k * 8 + sub;
        }
    }
    // LDS row -> (image of the tile, halo row, halo column); halo coordinates = pixel coordinates + 1
    static __device__ __forceinline__ void halo_pos(int row, int &ti, int &hy, int &hx) {
        if constexpr (WHOLE) {
            const int r = row - TW;
            ti = r / IPITCH;
            const int w = r % IPITCH;
            hy = w / TW + 1, hx = w % TW + 1;
        } else {
            hx = row % HWP;
            const int t = row / HWP;
            hy = t % HH, ti = t / HH;
        }
    }
    static constexpr int TW_SHIFT = TW == 16 ? 4 : (TW == 8 ? 3 : 2);
    static constexpr int TH_SHIFT = TH == 32 ? 5 : (TH == 16 ? 4 : (TH == 8 ? 3 : 2));
};

template <int VM>
__device__ __forceinline__ void wait_vm_lgkm0() {   // counted DMA wait + all of this wave's LDS reads back
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(VM) : "memory");
}

__device__ __forceinline__ void block_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// HB = halo images in LDS: 1 for single-chunk layers (C = 64), 2 otherwise (the next chunk's patch is
// DMA'd while the current one is consumed)
// NW = waves per workgroup (4: 128-pixel tiles; 8: 256-pixel tiles, which halve the weight DMA per MFMA)
// RPW = pixels per wave.  32: a wave's register tile is 32 pixels x BN channels.  64: 64 x BN -- per MFMA half the
//       weight-fragment LDS reads and half the weight DMA of the 32-pixel form (LDS traffic per FLOP is set by the
//       wave tile: (4 + 4) instead of (2 + 4) fragment reads per 16 MFMAs), and a 256-pixel workgroup of four waves:
//       512 workgroups for a 64 -> 64 layer on 32 x 32 maps = ONE round at two per CU, where the 128-pixel tiles need
//       1.33 rounds at three.  Its epilogue runs as two 32-pixel halves with operand fetches issued after the loop.
// PRO: the input's per-channel normalisation + activation (combat_conv_args.pro_*: train-mode BatchNorm + ReLU in front
//       of the convolution, preact_resnet.py:32,35) applied IN PLACE IN LDS after a halo patch has landed -- the
//       operands still arrive by DMA, the raw tensor is read once, and no activated copy has to be materialised by a
//       launch of its own in front of this one (combat_norm_act_fused: 25 dependent launches of a step).  Each wave
//       transforms the pieces it DMA'd itself (its own counted wait covers them: no extra barrier), one 16-byte slot
//       per lane: lane -> (row = piece row, channel group g = lane & 7), so a lane's eight (scale, shift) pairs are
//       fixed per chunk; out-of-image rows stay the zeros the DMA wrote (the padding of the ACTIVATED tensor).  The
//       next chunk's patch is transformed during taps 3..7 of the current one (it has landed by the counted wait of
//       tap 3 and is first read after the barrier of tap 8); only the first chunk's pass is exposed.  The workgroups
//       of channel tile 0 also write the activated interior of their patch to pro_act_dst (read back from LDS during
//       the same taps): the tensor the weight gradient of this layer reads.
// FLX:  >= 0: only epilogue flavour kEpiFlavours[FLX] is compiled into the kernel (the prologue's registers do not fit
//       beside five epilogue bodies); -2: the run-time switch over all of them
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {      // f(integral_constant<int, I>) ... f(integral_constant<int, N - 1>)
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int HPW_>
constexpr int pro_slot_pieces(int u) {   // pieces handled in the slot behind the barrier of tap u (u = 3..7)
    constexpr int PPP = (HPW_ + 4) / 5;
    if (u < 3 || u > 7) return 0;
    const int rem = HPW_ - (u - 3) * PPP;
    return rem <= 0 ? 0 : (rem < PPP ? rem : PPP);
}

template <int BN, int TW, int HB, int NW, int RPW, bool PRO = false, int FLX = -2>
__device__ __forceinline__ void conv3x3_dma_body(const DmaParams &p) {
    using T = TileCfg<4 * RPW, BN, 4>;  // shape of ONE wave's share: RPW pixels x all BN channels (whatever NW is)
    using TE = TileCfg<128, BN, 4>;     // the epilogue's unit: 32 pixels x BN channels
    using G = DGeo<TW, NW, RPW>;
    constexpr bool WIDE = RPW == 64;
    constexpr int WPW = BN / (8 * NW);      // weight DMA pieces per wave and step
    constexpr int WBYTES = BN * 128;        // one ring slot
    constexpr int HBYTES = G::HBYTES;
    constexpr int HPW = G::HPW;
    static_assert(!PRO || (NW == 4 && RPW == 32), "the LDS prologue exists for the 128-pixel tiles");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const combat_conv_args &a = p.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // PRO: the (scale, shift) table is fetched ahead of every DMA and parked in LDS behind the weight ring
    f32x4_t pt_sc = f32x4_t{0.f, 0.f, 0.f, 0.f}, pt_sh = pt_sc;
    if constexpr (PRO) {
        if (tid * 4 < a.C) {
            pt_sc = *reinterpret_cast<const f32x4_t *>(a.pro_scale + tid * 4);
            pt_sh = *reinterpret_cast<const f32x4_t *>(a.pro_shift + tid * 4);
        }
    }
#ifdef COMBAT_STAMPS
    const unsigned long long c_entry = __builtin_readcyclecounter();
    __builtin_amdgcn_sched_barrier(0);
#endif

    int tile_m, tile_n;
    int pf_rank = 0, pf_size = 1;   // this workgroup's place among the workgroups of its XCD that share its channel tile
    {
        const int nb = gridDim.x, bid = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, cnt = q + (xcd < r ? 1 : 0);
        const int swz = base + idx;
        if (p.m_fastest) {   // weights outweigh the input: an XCD's tiles share few weight rows, all pixels
            tile_m = swz % p.tiles_m;
            tile_n = swz / p.tiles_m;
            const int lo = base > tile_n * p.tiles_m ? base : tile_n * p.tiles_m;
            const int hi = base + cnt < (tile_n + 1) * p.tiles_m ? base + cnt : (tile_n + 1) * p.tiles_m;
            pf_rank = swz - lo;
            pf_size = hi - lo;
        } else {
            tile_n = swz % p.tiles_n;
            tile_m = swz / p.tiles_n;
            const int first = base + (tile_n - base % p.tiles_n + p.tiles_n) % p.tiles_n;
            pf_rank = (swz - first) / p.tiles_n;
            pf_size = (base + cnt - 1 - first) / p.tiles_n + 1;
        }
    }
    const int n0 = tile_n * BN;
    const int tx_ = tile_m % p.tiles_x, ty_ = (tile_m / p.tiles_x) % p.tiles_y, ig = tile_m / (p.tiles_x * p.tiles_y);
    const int img0 = ig * G::TI, oy0 = ty_ * G::TH, ox0 = tx_ * TW;
    const int C = a.C, H = a.H, W = a.W;
    unsigned char *halo = smem;
    unsigned char *wring = smem + HB * HBYTES;
    if constexpr (G::WHOLE) {   // the zeros of each halo image (no DMA ever writes them; first read after the first barrier)
        constexpr int NPAD = G::TI + 1, PADU = TW * 8, ZU = (2 * TW + 1) * 8;   // 16-byte units: one padding row group / the zone
        for (int u = tid; u < HB * (NPAD * PADU + ZU); u += 64 * NW) {
            const int hb_ = u / (NPAD * PADU + ZU), v = u % (NPAD * PADU + ZU);
            const int off = v < NPAD * PADU ? (v / PADU) * G::IPITCH * 128 + (v % PADU) * 16 : G::ZONE + (v - NPAD * PADU) * 16;
            *reinterpret_cast<u32x4_t *>(halo + hb_ * HBYTES + off) = u32x4_t{0u, 0u, 0u, 0u};
        }
    }

    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.src), 0, p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.wpack), 0, p.w_bytes, 0x00020000);

    unsigned wvoff[WPW];
#pragma unroll
    for (int j = 0; j < WPW; ++j) {
        const int n = (wid + NW * j) * 8 + (lane >> 3), slot = lane & 7;
        const int chunk = (slot - (n & 6)) & 7;
        wvoff[j] = (unsigned)(((n0 + n) * a.kpad + chunk * 8) * 2);
    }
    // weights of loop position t of chunk cc (dgrad walks the filter taps mirrored: tap 8 - t)
    auto issue_w = [&](int t, int cc, int slot) __attribute__((always_inline)) {
        const int tap = a.mode == 0 ? t : 8 - t;
        const int soff = (tap * C + cc * 64) * 2;
#pragma unroll
        for (int j = 0; j < ABL_DMA(WPW); ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_void_t *)(wring + slot * WBYTES + (wid + NW * j) * 1024), 16,
                                                     wvoff[j], soff, 0, 0);
    };

    // The first three weight tiles go out before anything else is computed: their round trip (and the
    // halo patch's, issued as soon as its per-lane offsets exist) overlaps the rest of the set-up --
    // ~3000 cycles of address arithmetic in a kernel of 15-25 k.
    DSTAMP(0);
    issue_w(0, 0, 0);
    issue_w(1, 0, 1);
    issue_w(2, 0, 2);
    // ---- weight rows -> this XCD's L2, now.  The loop asks for a tap's weight tile three taps (~1000 cycles) ahead;
    // in the step the weights come from HBM (400 MB of other tensors pass between two uses of a layer), a round trip
    // several times that, and a 512-channel layer spends 72 steps waiting for it: 20 us with the weights in cache,
    // 33 us cold (tools/conv_bench.py).  Every 128-byte line of the channel tile's rows is touched once here by a
    // 16-byte LDS-DMA into a scratch KB nobody reads -- no registers, no waits; the lines are divided among the
    // workgroups of this XCD that share the channel tile -- and has arrived by the time the halo patch has.
    if (p.w_prefetch) {
        const int lpr = (9 * C * 2) >> 7;                  // 128-byte lines per weight row (C is a multiple of 64)
        const int nlines = BN * lpr;
        unsigned char *dummy = smem + HB * HBYTES + 3 * WBYTES;
        for (int j = tid; pf_rank + pf_size * (j - lane) < nlines; j += 64 * NW) {
            const int l = pf_rank + pf_size * j;
            const int row = l / lpr, col = l - row * lpr;
            const unsigned off = l < nlines ? (unsigned)((n0 + row) * a.kpad * 2 + col * 128) : kOob;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_void_t *)dummy, 16, off, 0, 0, 0);
        }
    }
    // ---- per-lane DMA source offsets of the halo patch (bytes): the chunk term is added through the scalar offset
    unsigned hvoff[HPW];
#pragma unroll
    for (int j = 0; j < HPW; ++j) {
        const int row = G::piece_row(wid + NW * j, lane >> 3), slot = lane & 7;
        int ti, hy, hx;
        G::halo_pos(row, ti, hy, hx);
        // slot rotation of this halo position (see the fragment reads): by column; on 4-wide maps, where a
        // fragment's 16 pixels are four rows of the same four columns, by column bit 1 and row parity
        const int rot = TW == 4 ? ((hx & 2) + ((hy & 1) << 2)) : (hx & 6);
        const int chunk = (slot - rot) & 7;
        const int img = img0 + ti, iy = oy0 + hy - 1, ix = ox0 + hx - 1;
        const bool ok = (G::WHOLE || (row < G::HROWS && hx < TW + 2)) && img < a.N && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        hvoff[j] = ok ? (unsigned)((((img * H + iy) * W + ix) * C + chunk * 8) * 2) : kOob;
    }
    auto issue_h = [&](int cc, int hbuf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < (cc == 0 ? HPW : ABL_DMA(HPW)); ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srsrc, (lds_void_t *)(halo + hbuf * HBYTES + G::piece_row(wid + NW * j, 0) * 128), 16,
                                                     hvoff[j], cc * 128, 0, 0);
    };
    issue_h(0, 0);
    // ---- PRO: this lane's slot of every piece it transforms (same rows as its DMA pieces; channel group g = lane & 7
    // sits in LDS slot (g + rotation of the row) & 7), which of them lie inside the image, and where the interior
    // ones go in pro_act_dst (channel tile 0 only; everything else gets an out-of-range offset: the store is dropped)
    constexpr int kTA = PRO ? HPW : 1;
    int taddr[kTA];
    unsigned aoff[kTA], tvalid = 0;
    float *const ptab = reinterpret_cast<float *>(smem + HB * HBYTES + 3 * WBYTES + 1024);
    const float pslope = a.pro_slope;
    __amdgpu_buffer_rsrc_t r_pact = srsrc;
    if constexpr (PRO) {
        r_pact = __builtin_amdgcn_make_buffer_rsrc(a.pro_act_dst, 0, a.pro_act_dst ? p.src_bytes : 0u, 0x00020000);
        const int g = lane & 7;
#pragma unroll
        for (int j = 0; j < HPW; ++j) {
            const int row = G::piece_row(wid + NW * j, lane >> 3);
            int ti, hy, hx;
            G::halo_pos(row, ti, hy, hx);
            const int rot = TW == 4 ? ((hx & 2) + ((hy & 1) << 2)) : (hx & 6);
            taddr[j] = row * 128 + ((g + rot) & 7) * 16;
            const int img = img0 + ti, iy = oy0 + hy - 1, ix = ox0 + hx - 1;
            const bool ok = (G::WHOLE || (row < G::HROWS && hx < TW + 2)) && img < a.N && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            tvalid |= ok ? (1u << j) : 0u;
            const bool interior = ok && hx >= 1 && hx <= TW && hy >= 1 && hy <= G::TH && tile_n == 0;
            aoff[j] = interior ? (unsigned)((((img * H + iy) * W + ix) * C + g * 8) * 2) : kOob;
        }
        if (tid * 4 < C) {
            *reinterpret_cast<f32x4_t *>(ptab + tid * 4) = pt_sc;
            *reinterpret_cast<f32x4_t *>(ptab + C + tid * 4) = pt_sh;
        }
    }
    auto pro_tab = [&](int cc, float (&sc)[8], float (&sh)[8]) __attribute__((always_inline)) {
        const lds_f32_t *tab = (const lds_f32_t *)(smem + HB * HBYTES + 3 * WBYTES + 1024);
        load8f_lds(tab + cc * 64 + (lane & 7) * 8, sc);
        load8f_lds(tab + C + cc * 64 + (lane & 7) * 8, sh);
    };
    // act = lrelu(x * scale + shift): the arithmetic of norm_act_fused_kernel (norm.hip), bit for bit
    // (two steps, so that a caller can issue the reads of several pieces before the first write: a write in between
    // would keep the later reads behind it -- the compiler cannot tell the slots apart -- one LDS round trip per piece)
    auto pro_read = [&](auto j_tag, const unsigned char *hbase) __attribute__((always_inline)) {
        return *reinterpret_cast<const u32x4_t *>(hbase + taddr[decltype(j_tag)::value]);
    };
    auto pro_xform = [&](auto j_tag, const u32x4_t v, const float (&sc)[8], const float (&sh)[8], unsigned char *hbase) __attribute__((always_inline)) {
        constexpr int j = decltype(j_tag)::value;
        u32x4_t *q = reinterpret_cast<u32x4_t *>(hbase + taddr[j]);
        float f[8];
        unpack8v(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float y = fmaf(f[e], sc[e], sh[e]);
            f[e] = y > 0.f ? y : y * pslope;
        }
        const u32x4_t o = pack8v(f);
        const bool ok = (tvalid >> j) & 1u;
        u32x4_t w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = ok ? o[e] : v[e];     // (out-of-image rows: the zeros the DMA wrote)
        *q = w;
    };
    auto pro_side = [&](auto j_tag, int cc, const unsigned char *hbase) __attribute__((always_inline)) {
        constexpr int j = decltype(j_tag)::value;
        const u32x4_t v = *reinterpret_cast<const u32x4_t *>(hbase + taddr[j]);
        // (the chunk's byte offset goes into the VECTOR offset, not the scalar one: with an SGPR soffset the compiler
        // sees no hazard in overwriting a 16-byte store's data registers right behind it -- GCNHazardRecognizer's
        // rule for MUBUF stores wider than 8 bytes -- and on gfx950 the store then read registers the next piece's
        // arithmetic had already reused: ~3 % wrong values in lanes 12-15 / 28-31 / 44-47 / 60-63, different ones
        // from launch to launch.  With an immediate soffset the wait state is inserted.)
        __builtin_amdgcn_raw_buffer_store_b128(v, r_pact, aoff[j] + (unsigned)(cc * 128), 0, 0);
    };

    // ---- per-lane fragment read offsets: pixel fragment j at column offset dx, k-step ks
    int pa[2][T::FM][3];
#pragma unroll
    for (int j = 0; j < T::FM; ++j) {
        const int pj = wid * RPW + j * 16 + (lane & 15);
        const int tx = pj & (TW - 1), ty = (pj >> G::TW_SHIFT) & (G::TH - 1), ti = pj >> (G::TW_SHIFT + G::TH_SHIFT);
        // (WHOLE: the row above the pixel's, one column to the left -- filter row 0 of the first image reads the
        // padding row in front of it; a column offset that leaves the image reads the zone of zeros)
        const int r0 = G::WHOLE ? ti * G::IPITCH + ty * TW + tx - 1 : (ti * G::HH + ty) * G::HWP + tx;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int rot = TW == 4 ? (((tx + dx) & 2) + ((ty & 1) << 2)) : ((tx + dx) & 6);   // (TW 4: for filter row 0)
            const int s0 = ((lane >> 4) + rot) & 7;
            const bool in = !G::WHOLE || (tx + dx >= 1 && tx + dx <= TW);
            pa[0][j][dx] = in ? (r0 + dx) * 128 + s0 * 16 : G::ZONE;
            pa[1][j][dx] = in ? (r0 + dx) * 128 + (s0 ^ 4) * 16 : G::ZONE;
        }
    }
    int wa[2];
    {
        const int n = lane & 15;
        const int s0 = ((lane >> 4) + (n & 6)) & 7;
        wa[0] = n * 128 + s0 * 16;
        wa[1] = n * 128 + (s0 ^ 4) * 16;
    }

    f32x4_t acc[T::FN][T::FM];
#pragma unroll
    for (int i = 0; i < T::FN; ++i)
#pragma unroll
        for (int j = 0; j < T::FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragments of one k-step (32 channels) of loop position t: FM pixel + FN weight reads
    auto read_frags = [&](bf16x8_t (&fp)[T::FM], bf16x8_t (&fw)[T::FN], auto t_tag, auto ks_tag, auto hbuf_tag) __attribute__((always_inline)) {
        constexpr int t = decltype(t_tag)::value, ks = decltype(ks_tag)::value, hbuf = decltype(hbuf_tag)::value;
        constexpr int dy = t / 3, dx = t % 3;
        const unsigned char *hb = halo + hbuf * HBYTES + dy * (G::WHOLE ? TW : G::HWP) * 128;
        const unsigned char *wb = wring + (t % 3) * WBYTES;
        // TW 4: an odd filter row flips the row parity of the rotation key = the other k-step's slot (s0 ^ 4)
        constexpr int kk = (TW == 4 && (dy & 1)) ? (ks ^ 1) : ks;
#ifndef COMBAT_ABL_NOREAD
#pragma unroll
        for (int j = 0; j < T::FM; ++j) fp[j] = *reinterpret_cast<const bf16x8_t *>(hb + pa[kk][j][dx]);
#pragma unroll
        for (int i = 0; i < T::FN; ++i) fw[i] = *reinterpret_cast<const bf16x8_t *>(wb + wa[ks] + i * 2048);
#else
#pragma unroll
        for (int j = 0; j < T::FM; ++j) asm volatile("" : "+v"(fp[j]));
#pragma unroll
        for (int i = 0; i < T::FN; ++i) asm volatile("" : "+v"(fw[i]));
#endif
    };
    // with_reads: the FM + FN fragment reads issued just before (into the other register set) are spread one per
    // MFMA gap -- an MFMA holds the SIMD's vector issue for 8 of its 16 cycles, so a read issued inside a gap is nearly
    // free, while a block of reads in front of the MFMA block costs its full issue time (COMBAT_NO_INTERLEAVE: the
    // block form, for A/B timing).  Measured (tools/conv_bench.py, back to back / cold): 64-channel tiles 14.6 -> 14.0 /
    // 20.0 -> 19.3 us on the 128-channel 16 x 16 layer; 32-channel tiles (as many reads as MFMAs: nothing to hide
    // them behind) 15.6 -> 16.3 us on the 512-channel 4 x 4 layer, so those keep the block form.
    constexpr bool kInterleave = T::FM * T::FN > T::FM + T::FN;
    auto mfma_frags = [&](const bf16x8_t (&fp)[T::FM], const bf16x8_t (&fw)[T::FN], bool with_reads) __attribute__((always_inline)) {
#ifdef COMBAT_NO_INTERLEAVE
        __builtin_amdgcn_sched_barrier(0);
#else
        if (!kInterleave) __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int i = 0; i < T::FN; ++i)
#pragma unroll
            for (int j = 0; j < T::FM; ++j)
#ifndef COMBAT_ABL_NOMFMA
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fp[j], acc[i][j], 0, 0, 0);
#else
                acc[i][j][0] += (float)fw[i][0] * (float)fp[j][0];
#endif
#if !defined(COMBAT_NO_INTERLEAVE) && !defined(COMBAT_ABL_NOREAD) && !defined(COMBAT_ABL_NOMFMA)
        if (with_reads && kInterleave) {
            constexpr int NR = T::FM + T::FN, NM = T::FM * T::FN, NP = NR < NM ? NR : NM;
#pragma unroll
            for (int r = 0; r < NP; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one LDS read
            }
            if (NM > NP) __builtin_amdgcn_sched_group_barrier(0x008, NM - NP, 0);
        }
#else
        (void)with_reads;
#endif
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- fused epilogue.  A wave owns 32 pixels x BN channels; it transposes its accumulators through a
    // private LDS image (no workgroup barrier) so that every global access is 16 bytes of 8 consecutive
    // channels and 8 lanes cover one pixel's 128 contiguous bytes.  Residual / mask operands and the
    // per-channel tables come through buffer loads (rows beyond the last image and absent tensors use
    // an out-of-range offset / empty resource: loads give zeros, stores are dropped); they are issued
    // in the middle of the main loop's last chunk, so that their latency is covered by the remaining
    // MFMA steps -- the counted waits of those steps include them.
    // ---- fused epilogue (conv_dma_epilogue.hpp): operand fetches are issued PF_T taps before the loop ends
    using EC = EpiCfg<TE>;
    constexpr int NPF = WIDE ? 0 : EC::NPF;
    const unsigned dst_bytes = (unsigned)(a.N * p.PQ) * (unsigned)a.K * 2u;
    auto tile_row_off = [&](int row) -> long {   // element offset of tile pixel `row` in the dst-shaped tensors, or -1
        const int tx = row & (TW - 1), ty = (row >> G::TW_SHIFT) & (G::TH - 1);
        const int img = img0 + (row >> (G::TW_SHIFT + G::TH_SHIFT));
        return img < a.N ? (long)((img * H + oy0 + ty) * W + ox0 + tx) * a.K : -1;
    };
    EpiRegs<TE> epi;
    if (!WIDE) epi_init<TE>(epi, lane, wid, n0, tile_row_off);
    const bool ragged = G::TI > 1 && img0 + G::TI > a.N;
    auto epilogue_fetch = [&]() __attribute__((always_inline)) {
        if (!WIDE) epi_fetch<TE>(epi, a, dst_bytes, lane, n0);
    };

    // ---- main loop, software pipelined.  Loop position t of a chunk = filter tap; its weights live in
    // ring slot t % 3 and are DMA'd three positions ahead.  A position is two k-steps; its fragments
    // are read one k-step ahead into the other register set:
    //     read(t, k1) -> B | MFMA(A)  [t, k0] | wait W(t+1), barrier, DMA W(t+3) | read(t+1, k0) -> A | MFMA(B)  [t, k1]
    // so every LDS read has an MFMA block to land behind, and the one barrier per position sits between
    // two MFMA blocks.  At the barrier every wave has all its reads of slot t % 3 back (lgkmcnt(0)), so the
    // DMA of position t + 3 may overwrite it.  Counted vmcnt: what may stay in flight at the wait of
    // position t is everything issued after W(t+1): W(t+2) and, if issued one or two positions ago,
    // the next chunk's halo patch (issued at t = 0) / the epilogue operands (issued at t = PF_T).
    constexpr int PF_T = 4;
    const int nchunks = p.nchunks;
    bf16x8_t fpA[T::FM], fwA[T::FN], fpB[T::FM], fwB[T::FN];
#ifdef COMBAT_ABL_NOREAD
#pragma unroll
    for (int j = 0; j < T::FM; ++j) fpA[j] = fpB[j] = bf16x8_t{};
#pragma unroll
    for (int i = 0; i < T::FN; ++i) fwA[i] = fwB[i] = bf16x8_t{};
#endif
    static_assert(T::FM == RPW / 16 && T::FN == BN / 16, "wave tile");
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
#ifdef COMBAT_STAMPS
    if (threadIdx.x == 0 && p.stamps) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        p.stamps[blockIdx.x * 16 + 15] = ((unsigned long long)xcc << 32) | hw;
        p.stamps[blockIdx.x * 16 + 14] = c_entry;
    }
#endif
    wait_vm_lgkm0<0>();      // the halo patch was issued last
    block_barrier();
    DSTAMP(1);
    if constexpr (PRO) {     // the first chunk's patch: every wave its own pieces, then one more barrier (the only exposed pass)
        float sc0[8], sh0[8];
        pro_tab(0, sc0, sh0);
        u32x4_t raw0[HPW];
        static_for<0, HPW>([&](auto j_tag) __attribute__((always_inline)) { raw0[decltype(j_tag)::value] = pro_read(j_tag, halo); });
        static_for<0, HPW>([&](auto j_tag) __attribute__((always_inline)) { pro_xform(j_tag, raw0[decltype(j_tag)::value], sc0, sh0, halo); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        block_barrier();
    }
    read_frags(fpA, fwA, I0{}, I0{}, I0{});
    __builtin_amdgcn_sched_barrier(0);

    // PRO, behind the barrier of tap t = 3..7: this wave's pieces [first, first + count) -- the CURRENT chunk's go to
    // pro_act_dst (read back from LDS: the same schedule for every chunk), the NEXT chunk's patch (landed: the counted wait of tap 3 covers this wave's own halo
    // pieces) is transformed; its table registers are loaded in the slot of tap 3 and live to the slot of tap 7
    float nsc[8], nsh[8];
    auto pro_slot = [&](auto t_tag, auto last_tag, auto hbuf_tag, int cc) __attribute__((always_inline)) {
        constexpr int t = decltype(t_tag)::value, hbuf = decltype(hbuf_tag)::value;
        constexpr bool last = decltype(last_tag)::value;
        constexpr int PPP = (HPW + 4) / 5, first = (t - 3) * PPP, count = pro_slot_pieces<HPW>(t);
        if constexpr (count > 0) {
            unsigned char *cur = halo + hbuf * HBYTES, *nxt = halo + (HB == 2 ? (hbuf ^ 1) : 0) * HBYTES;
            if (!last && t == 3) pro_tab(cc + 1, nsc, nsh);
            u32x4_t rawn[count];
            if constexpr (!last)
                static_for<0, count>([&](auto i_tag) __attribute__((always_inline)) {
                    rawn[decltype(i_tag)::value] = pro_read(std::integral_constant<int, first + decltype(i_tag)::value>{}, nxt);
                });
            static_for<first, first + count>([&](auto j_tag) __attribute__((always_inline)) { pro_side(j_tag, cc, cur); });
            if constexpr (!last)
                static_for<0, count>([&](auto i_tag) __attribute__((always_inline)) {
                    pro_xform(std::integral_constant<int, first + decltype(i_tag)::value>{}, rawn[decltype(i_tag)::value], nsc, nsh, nxt);
                });
        }
    };
    auto chunk = [&](auto last_tag, auto hbuf_tag, int cc) __attribute__((always_inline)) {
        constexpr bool last = decltype(last_tag)::value;
        constexpr int hbuf = decltype(hbuf_tag)::value;
        using HBUF = std::integral_constant<int, hbuf>;
        using HNEXT = std::integral_constant<int, HB == 2 ? (hbuf ^ 1) : 0>;
        // (counted waits: at the wait of tap t every LOAD issued behind W(t + 1) may stay in flight -- W(t + 2), and
        // what the slots of taps t - 2 and t - 1 issued after their weight tile: the next patch, the epilogue
        // operands.  PRO's pro_act_dst STORES are deliberately not in the count: loads complete in order among
        // themselves, stores among themselves, but a store may be acknowledged before an older load has landed -- with
        // the stores added to the allowance, the wait let taps 6..8 read weight tiles that were still in flight
        // (wrong outputs on ~half the elements of some launches, right ones on others: found with the engine's step
        // test, not with the stand-alone comparison).  With at most (younger loads) operations allowed in flight,
        // W(t + 1) has landed whatever the stores do; the price is that the few stores of the previous slot must have
        // been acknowledged by then, a position later.)
#define COMBAT_DMA_POS(t)                                                                                    \
    {                                                                                                        \
        using TT = std::integral_constant<int, t>;                                                           \
        using TN = std::integral_constant<int, (t + 1) % 9>;                                                 \
        read_frags(fpB, fwB, TT{}, I1{}, HBUF{});                                                            \
        mfma_frags(fpA, fwA, true);                                                                          \
        if (!(last && t == 8)) {                                                                             \
            constexpr int n_w = (last && t >= 7) ? 0 : ABL_DMA(WPW);                                         \
            constexpr int n_h = (!last && (t == 1 || t == 2)) ? ABL_DMA(HPW) : 0;                            \
            constexpr int n_e = (last && (t == PF_T + 1 || t == PF_T + 2)) ? NPF : 0;                        \
            wait_vm_lgkm0<n_w + n_h + n_e>();                                                                \
            block_barrier();                                                                                 \
            if (t + 3 < 9) issue_w(t + 3, cc, t % 3);                                                        \
            else if (!last) issue_w(t + 3 - 9, cc + 1, t % 3);                                               \
            if (t == 0 && !last) issue_h(cc + 1, HNEXT::value);                                              \
            if (last && t == PF_T) epilogue_fetch();                                                         \
            if constexpr (PRO) pro_slot(TT{}, last_tag, hbuf_tag, cc);                                       \
            __builtin_amdgcn_sched_barrier(0);                                                               \
            if (t < 8) read_frags(fpA, fwA, TN{}, I0{}, HBUF{});                                             \
            else read_frags(fpA, fwA, TN{}, I0{}, HNEXT{});                                                  \
        }                                                                                                    \
        mfma_frags(fpB, fwB, !(last && t == 8));                                                             \
    }
        COMBAT_DMA_POS(0) COMBAT_DMA_POS(1) COMBAT_DMA_POS(2) COMBAT_DMA_POS(3) COMBAT_DMA_POS(4)
        COMBAT_DMA_POS(5) COMBAT_DMA_POS(6) COMBAT_DMA_POS(7) COMBAT_DMA_POS(8)
#undef COMBAT_DMA_POS
        if (cc == 0) DSTAMP(2);
    };
    if (HB == 1) {
        chunk(std::true_type{}, I0{}, 0);
    } else {
        int cc = 0;
        for (; cc + 2 < nchunks; cc += 2) {
            chunk(std::false_type{}, I0{}, cc);
            chunk(std::false_type{}, I1{}, cc + 1);
        }
        if (cc + 2 == nchunks) {
            chunk(std::false_type{}, I0{}, cc);
            chunk(std::true_type{}, I1{}, cc + 1);
        } else {
            chunk(std::true_type{}, I0{}, cc);
        }
    }
    DSTAMP(3);

    // ---- epilogue: accumulators -> this wave's fp32 LDS image (the operand images are dead once every
    // wave has passed the barrier) -> row-major items
    if constexpr (!WIDE) {
        // One epilogue flavour per kernel (FLX): the operands that flavour never reads would be dead loads, the
        // compiler would drop them, and the main loop's counted waits -- which allow EXACTLY epi_fetch()'s NPF
        // loads in flight at taps PF_T + 1 / + 2 -- would let the weight tiles of taps 6 / 7 be read before they
        // have landed (seen only with cold caches, inside the step: isolated launches passed 2 800 times).  Every
        // fetched register is therefore "used" here, as in the kernels that carry all flavours.
        if constexpr (FLX >= 0) epi_touch<TE>(epi);
        block_barrier();
#ifdef COMBAT_ABL_NOEPI     // (ablation: one store per lane keeps the accumulators alive)
        {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < T::FN; ++i)
#pragma unroll
                for (int j = 0; j < T::FM; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
            if (s == 12345.678f) reinterpret_cast<float *>(const_cast<void *>(a.dst))[lane] = s;
        }
#else
        if constexpr (FLX >= 0)
            epi_finish_fl<TE, kEpiFlavours[FLX < 0 ? 0 : FLX]>(epi, smem, acc, a, dst_bytes, lane, wid, n0, (long)tile_m * NW + wid, ragged,
                                                                 p.PQ, nullptr, EpiNoHook(), NW);
        else
            epi_finish<TE>(epi, smem, acc, a, dst_bytes, lane, wid, n0, (long)tile_m * NW + wid, ragged, p.PQ, p.flavour, NW);
#endif
    } else {
        // two 32-pixel halves, one after the other (both halves' operand sets at once would be 136 registers on
        // top of the 64 accumulators); the first half's fetches go out before the barrier that frees the operand images
        epi_init<TE>(epi, lane, wid, n0, [&](int row) -> long { return tile_row_off(wid * 64 + (row - wid * 32)); });
        epi_fetch<TE>(epi, a, dst_bytes, lane, n0);
        block_barrier();
        f32x4_t half[T::FN][2];
#pragma unroll
        for (int i = 0; i < T::FN; ++i) {
            half[i][0] = acc[i][0];
            half[i][1] = acc[i][1];
        }
        epi_finish<TE>(epi, smem, half, a, dst_bytes, lane, wid, n0, ((long)tile_m * NW + wid) * 2, ragged, p.PQ, p.flavour);
        epi_init<TE>(epi, lane, wid, n0, [&](int row) -> long { return tile_row_off(wid * 64 + 32 + (row - wid * 32)); });
        epi_fetch<TE>(epi, a, dst_bytes, lane, n0);
#pragma unroll
        for (int i = 0; i < T::FN; ++i) {
            half[i][0] = acc[i][2];
            half[i][1] = acc[i][3];
        }
        epi_finish<TE>(epi, smem, half, a, dst_bytes, lane, wid, n0, ((long)tile_m * NW + wid) * 2 + 1, ragged, p.PQ, p.flavour);
    }
    DSTAMP(4);
}

// (the body is a device function: the host pass cannot see the buffer-resource type it uses)
template <int BN, int TW, int HB, int NW, int RPW>
__global__ __launch_bounds__(64 * NW, NW == 8 || (RPW == 64 && HB == 2) ? 1 : 2)
void conv3x3_dma_kernel(const DmaParams p) {
    conv3x3_dma_body<BN, TW, HB, NW, RPW>(p);
}

// the same with the input's normalisation + activation applied in LDS (PRO above), one epilogue flavour per instantiation
template <int BN, int TW, int HB, int FLX>
__global__ __launch_bounds__(256, 2)
void conv3x3_dma_pro_kernel(const DmaParams p) {
    conv3x3_dma_body<BN, TW, HB, 4, 32, true, FLX>(p);
}

// ---------------------------------------------------------------------------------------------------------
// Weight-stationary, persistent variant for C = K = 64 on maps at least 16 wide (PreActResNet18 layer1, the
// generator's 32 x 32 layers): tile COMBAT_TILE_S128x64.
//
// The layer's whole filter bank -- 9 taps x 64 x 64 bf16 = 72 KB -- fits in LDS, so a workgroup loads it ONCE and
// walks a contiguous range of 128-pixel tiles; the per-tap weight ring of the kernel above, its DMA issue and its
// barrier per tap disappear, and a tile's main loop is nothing but fragment reads and MFMAs.  These layers move
// 33-67 MB for 9.7 GFLOP: they are bound by HBM and by what a launch cannot overlap with itself when all its
// workgroups run one round in lockstep (loads first, MFMAs, stores last -- DESIGN.md section 5).  Here one 512-thread
// workgroup per CU runs two groups of four waves half a tile apart ("ping-pong"):
//
//     half-period h:   group A: MFMAs of its tile      |  group B: epilogue of its previous tile (stores), halo DMA
//     half-period h+1: group A: epilogue, halo DMA ... |  group B: MFMAs                          of its next tile
//
// so each SIMD always has one wave on the matrix pipe and one on the VALU / memory side, the next tile's halo patch
// lands while the other group computes, and output stores drain under MFMAs.  One barrier per half-period.  The two
// groups never run their epilogues at the same time, so they share ONE set of wave transposition images:
// 72 KB weights + 2 x 24 KB halo patches + 34 KB epilogue = 154 KB of the CU's 160.
// Tile geometry, statistics rows and the fused epilogue are exactly COMBAT_TILE_D128x64's (16 x 8 pixels of one
// image, one statistics row per wave), so the two kernels are interchangeable launch by launch.
template <int FLI>   // FLI: index into kEpiFlavours (one epilogue body per kernel instantiation), -1: the generic body
__device__ __forceinline__ void conv3x3_ws_body(const DmaParams &p) {
    constexpr int FL = FLI < 0 ? -1 : kEpiFlavours[FLI < 0 ? 0 : FLI];
    constexpr int BN = 64, TW = 16;
    using T = TileCfg<128, BN, 4>;
    using G = DGeo<TW, 4, 32>;
    using EC = EpiCfg<T>;
    constexpr int WBYTES = 9 * BN * 128, HBYTES = G::HBYTES, HPW = G::HPW, EPI_OFF = WBYTES + 2 * HBYTES;
    constexpr int TAB_OFF = EPI_OFF + EC::LDS_BYTES;    // bias | xh_rstd | xh_mean, 64 floats each (see epi_finish_fl)
    static_assert(G::TI == 1 && G::TH == 8 && G::HWP == 18, "one 16 x 8 patch of one image per tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const combat_conv_args &a = p.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2, gw = wid & 3;
    const int C = a.C, H = a.H, W = a.W;
    unsigned char *const wlds = smem;
    unsigned char *const halo = smem + WBYTES + grp * HBYTES;

#ifdef COMBAT_STAMPS     // per wave: 40 cycle stamps (tools/stamps_ws.py)
#define WS_STAMP(i)                                                                                               \
    do {                                                                                                          \
        if (lane == 0 && p.stamps && (i) < 40) p.stamps[(blockIdx.x * 8 + wid) * 40 + (i)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define WS_STAMP(i)
#endif
    WS_STAMP(0);
    const int first = blockIdx.x * p.per, left = p.tiles_m - first;
    const int cnt = left < p.per ? left : p.per;            // tiles of this workgroup (>= 1)
    const int ng = (cnt - grp + 1) >> 1;                      // ... of this group: first + grp, first + grp + 2, ...
    constexpr int SVPL = EC::NC * 16 / 64;                  // statistics values per lane (epi_finish_fl)
    float stat_run[SVPL];                                     // this wave's sums over all its tiles (per-workgroup rows)
#pragma unroll
    for (int u = 0; u < SVPL; ++u) stat_run[u] = 0.f;

    // Every LDS-DMA of this kernel is issued from inline asm (see the halo patch below for why): resources as four
    // scalar words, the LDS address through M0.
    auto rsrc_words = [&](const void *ptr, unsigned bytes) __attribute__((always_inline)) {
        const unsigned long base = (unsigned long)ptr;
        u32x4_t w;
        w[0] = __builtin_amdgcn_readfirstlane((unsigned)base);
        w[1] = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32) & 0xffffu);
        w[2] = __builtin_amdgcn_readfirstlane(ptr ? bytes : 0u);      // (absent tensor: every load reads zeros)
        w[3] = 0x00020000u;
        return w;
    };
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(lds_void_t *)smem);
#define COMBAT_DMA16(rs, lds_addr, voff)                                                         \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs) : "memory", "m0")
#define COMBAT_DMA4(rs, lds_addr, voff)                                                          \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs) : "memory", "m0")
    // ---- the filter bank, once: wave w brings rows 8 w .. 8 w + 7 of every tap (dgrad walks the taps mirrored)
    {
        const u32x4_t wrs = rsrc_words(a.wpack, p.w_bytes);
        const int n = wid * 8 + (lane >> 3), slot = lane & 7;
        const int chunk = (slot - (n & 6)) & 7;
        const unsigned voff = (unsigned)((n * a.kpad + chunk * 8) * 2);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int tap = a.mode == 0 ? t : 8 - t;
            COMBAT_DMA16(wrs, lds0 + t * (BN * 128) + wid * 1024, voff + (unsigned)(tap * C * 2));
        }
    }
    // ---- per-channel tables of the epilogue, once: 64 floats each = one 4-byte DMA per table (wave 0)
    if (wid == 0) {
        COMBAT_DMA4(rsrc_words(a.bias, BN * 4), lds0 + TAB_OFF, (unsigned)(lane * 4));
        COMBAT_DMA4(rsrc_words(a.xh_rstd, BN * 4), lds0 + TAB_OFF + BN * 4, (unsigned)(lane * 4));
        COMBAT_DMA4(rsrc_words(a.xh_mean, BN * 4), lds0 + TAB_OFF + 2 * BN * 4, (unsigned)(lane * 4));
    }
    // ---- halo DMA bookkeeping that does not depend on the tile: decoded position + byte offset relative to the
    // tile's first pixel, with the channel chunk this lane's LDS slot must receive (rotation by halo column)
    int hdec[HPW], hrel[HPW];
#pragma unroll
    for (int j = 0; j < HPW; ++j) {
        const int row = (gw + 4 * j) * 8 + (lane >> 3), slot = lane & 7;
        const int hx = row % G::HWP, hy = row / G::HWP;
        const int chunk = (slot - (hx & 6)) & 7;
        hdec[j] = (row < G::HROWS && hx < TW + 2) ? (hx | (hy << 8)) : -1;
        hrel[j] = (((hy - 1) * W + hx - 1) * C + chunk * 8) * 2;
    }
    // tile coordinates of this group's current tile (tile index, column / row of tiles, image): divided out once,
    // then stepped by two tiles with carries (the divisions were ~300 cycles of every half-period)
    int tm_cur = first + grp, c_tx = 0, c_ty = 0, t_img = 0, t_oy = 0, t_ox = 0;
    {
        const int r = tm_cur / p.tiles_x;
        c_tx = tm_cur - r * p.tiles_x;
        t_img = r / p.tiles_y;
        c_ty = r - t_img * p.tiles_y;
        t_ox = c_tx * TW;
        t_oy = c_ty * G::TH;
    }
    auto next_tile = [&]() __attribute__((always_inline)) {
        tm_cur += 2;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (++c_tx == p.tiles_x) {
                c_tx = 0;
                if (++c_ty == p.tiles_y) {
                    c_ty = 0;
                    ++t_img;
                }
            }
        }
        t_ox = c_tx * TW;
        t_oy = c_ty * G::TH;
    };
    // The DMA is issued from inline asm because the compiler guards the first LDS read behind an LDS-DMA it knows of
    // with s_waitcnt vmcnt(0) -- here that would park the epilogue (whose transposition image is LDS) until the next
    // tile's patch has landed, the very latency this schedule hides.  Unknown to the compiler, a DMA only makes its
    // own counted waits conservative (the counter is in order); the explicit waits below order the patch's readers.
    const u32x4_t srs = rsrc_words(a.src, p.src_bytes);
    const unsigned halo_lds = lds0 + WBYTES + grp * HBYTES + gw * 1024;
    auto issue_halo = [&]() __attribute__((always_inline)) {          // of the tile tile_coords() was last called for
        const int org = ((t_img * H + t_oy) * W + t_ox) * C * 2;
#pragma unroll
        for (int j = 0; j < HPW; ++j) {
            const int d = hdec[j];
            const bool ok = d >= 0 && (unsigned)(t_oy + (d >> 8) - 1) < (unsigned)H && (unsigned)(t_ox + (d & 255) - 1) < (unsigned)W;
            const unsigned voff = ok ? (unsigned)(org + hrel[j]) : kOob;
            COMBAT_DMA16(srs, halo_lds + j * 4096, voff);
        }
    };
#undef COMBAT_DMA4

    // ---- per-lane fragment read offsets (as in the kernel above, a group's wave in place of the workgroup's)
    int pa[2][T::FM][3];
#pragma unroll
    for (int j = 0; j < T::FM; ++j) {
        const int pj = gw * 32 + j * 16 + (lane & 15);
        const int tx = pj & (TW - 1), ty = pj >> 4;
        const int r0 = ty * G::HWP + tx;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int s0 = ((lane >> 4) + ((tx + dx) & 6)) & 7;
            pa[0][j][dx] = (r0 + dx) * 128 + s0 * 16;
            pa[1][j][dx] = (r0 + dx) * 128 + (s0 ^ 4) * 16;
        }
    }
    int wa[2];
    {
        const int n = lane & 15;
        const int s0 = ((lane >> 4) + (n & 6)) & 7;
        wa[0] = n * 128 + s0 * 16;
        wa[1] = n * 128 + (s0 ^ 4) * 16;
    }

    const unsigned dst_bytes = (unsigned)(a.N * p.PQ) * (unsigned)a.K * 2u;
    // byte offset of this lane's epilogue items relative to the tile's first pixel (every tile lies inside the tensor)
    unsigned erel[EC::EQ];
#pragma unroll
    for (int q = 0; q < EC::EQ; ++q) {
        const int row = gw * 32 + (q * 64 + lane) / EC::NC;
        erel[q] = (unsigned)((((row >> 4) * W + (row & (TW - 1))) * a.K + (lane % EC::NC) * 8) * 2);
    }
    constexpr bool kUsesMask = FL < 0 || (FL & EF_MASK), kUsesTabs = FL < 0 || (FL & (EF_MASK | EF_ACT));
    // operand fetches of the current tile: only what this flavour can read, and only the tensors that exist
    auto ws_fetch = [&](EpiRegs<T> &r) __attribute__((always_inline)) {
        const unsigned org = (unsigned)(((t_img * H + t_oy) * W + t_ox) * a.K * 2);
#pragma unroll
        for (int q = 0; q < EC::EQ; ++q) r.evoff[q] = org + erel[q];
        if (a.add_pre) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.add_pre), 0, dst_bytes, 0x00020000);
#pragma unroll
            for (int q = 0; q < EC::EQ; ++q) r.e_pre[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, r.evoff[q], 0, 0);
        }
        if constexpr (kUsesMask) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.mask_x), 0, a.mask_x ? dst_bytes : 0u, 0x00020000);
#pragma unroll
            for (int q = 0; q < EC::EQ; ++q) r.e_x[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, r.evoff[q], 0, 0);
        }
        if (a.add_post) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.add_post), 0, dst_bytes, 0x00020000);
#pragma unroll
            for (int q = 0; q < EC::EQ; ++q) r.e_post[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, r.evoff[q], 0, 0);
        }
    };
    auto ws_touch = [&](EpiRegs<T> &r) __attribute__((always_inline)) {   // (see epi_touch)
#pragma unroll
        for (int q = 0; q < EC::EQ; ++q) {
            asm volatile("" : "+v"(r.e_pre[q]), "+v"(r.e_post[q]));
            if constexpr (kUsesMask) asm volatile("" : "+v"(r.e_x[q]));
        }
    };
    EpiRegs<T> epi;
    if constexpr (kUsesTabs) {   // the (scale, shift) table pair of this flavour: per channel, the same for every tile
        // (buffer loads: an absent table is an empty resource and reads zeros -- no select on the loaded value, which
        // would make the compiler wait for the load, and with it for the weight DMA in front of it, right here)
        const float *tab_sc = a.mask_x ? a.mask_scale : a.act_scale, *tab_sh = a.mask_x ? a.mask_shift : a.act_shift;
        const __amdgpu_buffer_rsrc_t r_sc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tab_sc), 0, tab_sc ? BN * 4u : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_sh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tab_sh), 0, tab_sh ? BN * 4u : 0u, 0x00020000);
        const int tb = (lane % EC::NC) * 32;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            epi.t_sc[h2] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r_sc, tb + 16 * h2, 0, 0));
            epi.t_sh[h2] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r_sh, tb + 16 * h2, 0, 0));
        }
    } else {
        epi.t_sc[0] = epi.t_sc[1] = epi.t_sh[0] = epi.t_sh[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    f32x4_t acc[T::FN][T::FM];

    // One k-step = FM + FN fragment reads for the NEXT k-step and FM * FN MFMAs of the current one, interleaved one
    // read per MFMA gap: an MFMA holds the SIMD's vector issue for 8 of its 16 cycles, so a ds_read_b128 issued inside
    // the gap is (nearly) free, while a block of six reads in front of the MFMA block costs its full issue time.
    auto read_frags = [&](bf16x8_t (&fp)[T::FM], bf16x8_t (&fw)[T::FN], auto t_tag, auto ks_tag) __attribute__((always_inline)) {
        constexpr int t = decltype(t_tag)::value, ks = decltype(ks_tag)::value;
        constexpr int dy = t / 3, dx = t % 3;
        const unsigned char *hb = halo + dy * G::HWP * 128;
        const unsigned char *wb = wlds + t * (BN * 128);
#pragma unroll
        for (int j = 0; j < T::FM; ++j) fp[j] = *reinterpret_cast<const bf16x8_t *>(hb + pa[ks][j][dx]);
#pragma unroll
        for (int i = 0; i < T::FN; ++i) fw[i] = *reinterpret_cast<const bf16x8_t *>(wb + wa[ks] + i * 2048);
    };
    auto mfma_frags = [&](const bf16x8_t (&fp)[T::FM], const bf16x8_t (&fw)[T::FN], bool with_reads) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < T::FN; ++i)
#pragma unroll
            for (int j = 0; j < T::FM; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fp[j], acc[i][j], 0, 0, 0);
        if (with_reads) {
#pragma unroll
            for (int r = 0; r < T::FM + T::FN; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one LDS read
            }
            __builtin_amdgcn_sched_group_barrier(0x008, T::FM * T::FN - (T::FM + T::FN), 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // ---- prologue: the filter bank and this group's first halo patch must have landed before anyone reads LDS; the
    // first tile's epilogue operands are fetched behind that wait and land under its MFMAs
    if (ng > 0) issue_halo();
    WS_STAMP(1);
    wait_vm_lgkm0<0>();
    block_barrier();
    // ---- group B runs half a period behind group A: one barrier interval = one half-period, in which one group
    // computes a tile and the other finishes its previous one.  Both groups run the SAME straight-line loop body
    // (compute, barrier, finish + prefetch, barrier), so the loop-carried epilogue operands stay in their registers.
    WS_STAMP(2);
    if (grp == 1) block_barrier();
    WS_STAMP(3);
    for (int k = 0; k < ng; ++k) {
        // this tile's epilogue operands: fetched now, landing under its MFMAs, consumed in the same iteration (not
        // carried around the loop: the compiler copied loop-carried load results at the latch, behind a full wait)
        ws_fetch(epi);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < T::FN; ++i)
#pragma unroll
            for (int j = 0; j < T::FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        {
            bf16x8_t fpA[T::FM], fwA[T::FN], fpB[T::FM], fwB[T::FN];
            __builtin_amdgcn_s_setprio(1);      // the matrix pipe's wave goes first; its SIMD partner is in an epilogue
            read_frags(fpA, fwA, I0{}, I0{});
            __builtin_amdgcn_sched_barrier(0);
#define COMBAT_WS_POS(t)                                                                       \
    {                                                                                          \
        read_frags(fpB, fwB, std::integral_constant<int, t>{}, I1{});                          \
        mfma_frags(fpA, fwA, true);                                                            \
        if (t < 8) read_frags(fpA, fwA, std::integral_constant<int, (t + 1) % 9>{}, I0{});     \
        mfma_frags(fpB, fwB, t < 8);                                                           \
    }
            COMBAT_WS_POS(0) COMBAT_WS_POS(1) COMBAT_WS_POS(2) COMBAT_WS_POS(3) COMBAT_WS_POS(4)
            COMBAT_WS_POS(5) COMBAT_WS_POS(6) COMBAT_WS_POS(7) COMBAT_WS_POS(8)
#undef COMBAT_WS_POS
            __builtin_amdgcn_s_setprio(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's fragment reads are back: its patch may be overwritten
        WS_STAMP(4 + 6 * k);
        block_barrier();
        WS_STAMP(5 + 6 * k);
        const int tm_done = tm_cur;
        const bool more = k + 1 < ng;
        ws_touch(epi);       // (the operand fetches have landed: say so before the DMA below enters the wait counts)
        if (more) {          // the patch of this group's next tile lands while the other group computes
            next_tile();
            issue_halo();
        }
        WS_STAMP(6 + 6 * k);
        // (the wait for the next patch sits in front of this tile's stores: nothing else is in flight there)
        epi_finish_fl<T, FL>(epi, smem + EPI_OFF, acc, a, dst_bytes, lane, gw, 0, (long)tm_done * 4 + gw, false, p.PQ,
                             (const lds_f32_t *)(smem + TAB_OFF), [&]() __attribute__((always_inline)) {
                                 __builtin_amdgcn_sched_barrier(0);
                                 WS_STAMP(7 + 6 * k);
                                 wait_vm_lgkm0<0>();
                                 WS_STAMP(8 + 6 * k);
                                 __builtin_amdgcn_sched_barrier(0);
                             }, 0, [&](const float (&tot)[SVPL]) __attribute__((always_inline)) {
#pragma unroll
                                 for (int u = 0; u < SVPL; ++u) stat_run[u] += tot[u];
                             });
        WS_STAMP(9 + 6 * k);
        block_barrier();
    }
    // every wave executes the same number of barriers: A ran 1 + 2 ceil(cnt / 2), B 2 + 2 floor(cnt / 2)
    if ((cnt & 1) == grp) block_barrier();
    // ---- one statistics row per workgroup (COMBAT_STATS_PER_WORKGROUP): the eight waves' running sums, added in wave order
    if ((FL < 0 ? (a.stats_kind & 3) != 0 : (FL & (EF_STATS1 | EF_STATS2)) != 0) && (a.stats_kind & COMBAT_STATS_PER_WORKGROUP)) {
        float *slots = reinterpret_cast<float *>(smem + EPI_OFF);     // (every epilogue is behind the barriers above)
#pragma unroll
        for (int u = 0; u < SVPL; ++u) slots[wid * (64 * SVPL) + lane * SVPL + u] = stat_run[u];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        block_barrier();
        if (wid == 0) {
            float sum[SVPL];
#pragma unroll
            for (int u = 0; u < SVPL; ++u) sum[u] = 0.f;
            for (int w = 0; w < 8; ++w)
#pragma unroll
                for (int u = 0; u < SVPL; ++u) sum[u] += slots[w * (64 * SVPL) + lane * SVPL + u];
            const int vg = (lane / EC::NC) * SVPL, n = (lane % EC::NC) * 8;
            float *orow = a.stats + ((size_t)blockIdx.x * 2 + (vg >> 3)) * a.K + n + (vg & 7);
#pragma unroll
            for (int u = 0; u < SVPL; ++u) orow[u] = sum[u];
        }
    }
#undef WS_STAMP
}

template <int FLI>
__global__ __launch_bounds__(512, 2) void conv3x3_ws_kernel(const DmaParams p) { conv3x3_ws_body<FLI>(p); }

int geo_tw(const combat_conv_args *a) {
    if (a->W >= 16) return (a->W % 16 == 0 && a->H % 8 == 0) ? 16 : 0;
    if (a->W == 8) return a->H == 8 ? 8 : 0;
    if (a->W == 4) return a->H == 4 ? 4 : 0;
    return 0;
}

// the in-LDS prologue (PRO in the kernel body): per-channel scale / shift + (leaky) ReLU in front of a forward
// convolution whose epilogue is one of the two flavours a train-mode network uses behind it (plain / statistics)
bool pro_ok(const combat_conv_args *a) {
    static const bool off = getenv("COMBAT_NO_DMA_PRO") != nullptr;     // A/B: such launches fall back to the halo kernels
    if (off) return false;
    if (!a->pro_scale || !a->pro_shift || !a->pro_act || a->pro_group_stride != 0 || a->mode != 0) return false;
    if (a->C > 512) return false;                                        // (the table's LDS: 2 x 512 floats)
    const int fl = epi_flavour_of(*a);
    return fl == 0 || fl == 2;
}

bool applicable(const combat_conv_args *a, int BN) {
    if (!(a->R == 3 && a->S == 3 && a->stride == 1 && a->pad == 1 && a->P == a->H && a->Q == a->W)) return false;
    if ((a->pro_scale || a->pro_act || a->pro_act_dst) && !pro_ok(a)) return false;   // (only the in-LDS prologue below)
    if (a->tanh_out) return false;   // (stays with the halo kernels)
    if (a->mask_x && a->act_dst) return false;   // one table pair travels with the operand prefetch
    if (a->C < 64 || (a->C & 63) || a->K % BN || a->kpad < 9 * a->C) return false;
    if ((long)a->N * a->H * a->W * a->C * 2 >= (long)kOob) return false;
    if ((long)a->rows_pad * a->kpad * 2 >= (long)kOob) return false;
    if ((long)a->N * a->H * a->W * a->K * 2 >= (long)kOob) return false;
    return geo_tw(a) != 0;
}

template <int TW, int NW, int RPW>
void fill(const combat_conv_args *a, int BN, DmaParams &p) {
    using G = DGeo<TW, NW, RPW>;
    p.a = *a;
    p.tiles_x = a->W / TW;
    p.tiles_y = a->H / G::TH;
    p.tiles_m = p.tiles_x * p.tiles_y * ((a->N + G::TI - 1) / G::TI);
    p.tiles_n = a->K / BN;
    {   // per-XCD L2 footprint: (input / 8 + all weights) with channel tiles fastest, (input + weights / 8) otherwise
        const long in_b = (long)a->N * a->H * a->W * a->C * 2, w_b = (long)a->K * a->C * 18;
        p.m_fastest = in_b < w_b;   // (512-channel 4x4 layers: 19.3 -> 16.7 us)
    }
    p.PQ = a->H * a->W;
    p.nchunks = a->C / 64;
    p.src_bytes = (unsigned)((long)a->N * a->H * a->W * a->C * 2);
    p.w_bytes = (unsigned)((long)a->rows_pad * a->kpad * 2);
    p.flavour = epi_flavour_of(*a);
    // (worth it where a channel tile's weight rows are many lines per workgroup step: >= 256 channels)
    p.w_prefetch = a->C >= 256 && !getenv("COMBAT_NO_WPREFETCH");
#ifdef COMBAT_STAMPS
    p.stamps = g_stamps_dma_host;
#else
    p.stamps = nullptr;
#endif
}

template <int BN, int TW, int HB, int NW, int RPW>
int launch_hb(const DmaParams &p, hipStream_t st) {
    using G = DGeo<TW, NW, RPW>;
    constexpr int stage = HB * G::HBYTES + 3 * BN * 128 + 1024;   // (+ the weight prefetch's scratch KB)
    constexpr int ep = EpiCfg<TileCfg<128, BN, 4>>::LDS_BYTES / 4 * NW;
    constexpr int smem0 = stage > ep ? stage : ep;
    static const int exp_pad = getenv("COMBAT_EXP_LDS") ? atoi(getenv("COMBAT_EXP_LDS")) : 0;   // EXPERIMENT: force a footprint
    const int smem = exp_pad > smem0 && (BN == 32 || getenv("COMBAT_EXP_LDS_ALL")) ? exp_pad : smem0;
    auto kern = conv3x3_dma_kernel<BN, TW, HB, NW, RPW>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return COMBAT_ELAUNCH;
        attr_set = true;
    }
    COMBAT_LAUNCH(kern, dim3(p.tiles_m * p.tiles_n), dim3(64 * NW), smem, st, p);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

template <int BN, int TW, int HB, int FLX>
int launch_pro_fl(const DmaParams &p, hipStream_t st) {
    using G = DGeo<TW, 4, 32>;
    constexpr int stage = HB * G::HBYTES + 3 * BN * 128 + 1024 + 2 * 512 * 4;   // (+ the prologue's table)
    constexpr int ep = EpiCfg<TileCfg<128, BN, 4>>::LDS_BYTES;
    constexpr int smem = stage > ep ? stage : ep;
    auto kern = conv3x3_dma_pro_kernel<BN, TW, HB, FLX>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return COMBAT_ELAUNCH;
        attr_set = true;
    }
    COMBAT_LAUNCH(kern, dim3(p.tiles_m * p.tiles_n), dim3(256), smem, st, p);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

template <int BN, int TW>
int launch_pro(const DmaParams &p, hipStream_t st) {
    if (p.flavour == 0) return p.nchunks == 1 ? launch_pro_fl<BN, TW, 1, 0>(p, st) : launch_pro_fl<BN, TW, 2, 0>(p, st);
    if (p.flavour == 2) return p.nchunks == 1 ? launch_pro_fl<BN, TW, 1, 2>(p, st) : launch_pro_fl<BN, TW, 2, 2>(p, st);
    return COMBAT_EINVAL;
}

template <int BN, int TW, int NW, int RPW = 32>
int launch(const combat_conv_args *a, hipStream_t st) {
    DmaParams p;
    fill<TW, NW, RPW>(a, BN, p);
    if (a->pro_scale) {
        if constexpr (NW == 4 && RPW == 32) return launch_pro<BN, TW>(p, st);
        else return COMBAT_EINVAL;
    }
    return p.nchunks == 1 ? launch_hb<BN, TW, 1, NW, RPW>(p, st) : launch_hb<BN, TW, 2, NW, RPW>(p, st);
}

int tile_bn(int tile) {
    return tile == COMBAT_TILE_D128x64 || tile == COMBAT_TILE_D256x64 || tile == COMBAT_TILE_D256W64 ? 64
                                                                                                 : (tile == COMBAT_TILE_D128x32 ? 32 : 0);
}
int tile_bm(int tile) { return tile == COMBAT_TILE_D256x64 || tile == COMBAT_TILE_D256W64 ? 256 : 128; }

// tile height for a tile of bm pixels (0: the geometry does not exist)
int geo_th(const combat_conv_args *a, int bm) {
    const int tw = geo_tw(a);
    if (!tw || (tw == 4 && bm == 256)) return 0;   // 16 images of 4x4 per tile: two halo images would not fit LDS
    if (tw == 16 && bm == 256 && a->H % 16 != 0) return 0;
    const int th = tw == 16 ? bm / 16 : (tw == 8 ? 8 : 4);
    return a->H % th == 0 ? th : 0;
}

int tiles_m_of(const combat_conv_args *a, int bm = 128) {
    const int tw = geo_tw(a), th = geo_th(a, bm), ti = bm / (tw * th);
    return (a->W / tw) * (a->H / th) * ((a->N + ti - 1) / ti);
}

// weight-stationary persistent kernel: the whole filter bank in LDS (C = K = 64), 16-wide tiles, and at least two
// tiles per CU (with one, there is nothing to overlap: the ring kernel's three co-resident workgroups do better)
constexpr int kWsSmem = 9 * 64 * 128 + 2 * DGeo<16, 4, 32>::HBYTES + EpiCfg<TileCfg<128, 64, 4>>::LDS_BYTES + 3 * 64 * 4;
bool ws_applicable(const combat_conv_args *a) {
    if (a->pro_scale) return false;   // (the in-LDS prologue lives in the ring kernel)
    if (a->C != 64 || a->K != 64 || !applicable(a, 64) || geo_tw(a) != 16 || a->H % 8) return false;
    if (epi_flavour_of(*a) < 0) return false;   // (the generic epilogue body does not fit beside the persistent state: 44 spills)
    return (long)tiles_m_of(a) >= 2 * 256 || a->tile == COMBAT_TILE_S128x64;
}

template <int FLI>
int launch_ws_fl(const DmaParams &p, int blocks, hipStream_t st) {
    auto kern = conv3x3_ws_kernel<FLI>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kWsSmem) != hipSuccess)
            return COMBAT_ELAUNCH;
        attr_set = true;
    }
    COMBAT_LAUNCH(kern, dim3(blocks), dim3(512), kWsSmem, st, p);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

// tiles per persistent workgroup / workgroups of the weight-stationary launch
void ws_grid(int tiles_m, int &per, int &blocks) {
    const int cus = 256;
    per = (tiles_m + cus - 1) / cus;
    if (per < 1) per = 1;
    blocks = (tiles_m + per - 1) / per;
}

int launch_ws(const combat_conv_args *a, hipStream_t st) {
    DmaParams p;
    fill<16, 4, 32>(a, 64, p);
    int blocks;
    ws_grid(p.tiles_m, p.per, blocks);
    switch (p.flavour) {
        case 0: return launch_ws_fl<0>(p, blocks, st);
        case 1: return launch_ws_fl<1>(p, blocks, st);
        case 2: return launch_ws_fl<2>(p, blocks, st);
        case 3: return launch_ws_fl<3>(p, blocks, st);
        case 4: return launch_ws_fl<4>(p, blocks, st);
        default: return COMBAT_EINVAL;
    }
    static_assert(kNumEpiFlavours == 5, "one case per flavour");
}

}  // namespace

// tile id the DMA kernel would use for these args (an explicit a->tile is honoured), or 0.
// Measured on MI355X (profiles/r01_c_dma_tile_sweep.txt): wherever it applies it beats the
// register-staged halo tiles (1.3-1.9x on the PreActResNet18 layer shapes).
int conv3x3d_pick(const combat_conv_args *a) {
    if (a->tile == COMBAT_TILE_S128x64) return ws_applicable(a) ? a->tile : 0;
    if (a->tile) {
        const int bn = tile_bn(a->tile);
        if (a->pro_scale && tile_bm(a->tile) != 128) return 0;
        if (a->tile == COMBAT_TILE_D256W64 && geo_tw(a) != 16) return 0;
        return bn && applicable(a, bn) && geo_th(a, tile_bm(a->tile)) ? a->tile : 0;
    }
    if (!applicable(a, 64)) return applicable(a, 32) ? COMBAT_TILE_D128x32 : 0;
    if (ws_applicable(a) && !getenv("COMBAT_NO_WS")) return COMBAT_TILE_S128x64;
    // skinny layers: 32-channel tiles double the workgroup count when 64-channel tiles give no more than
    // one workgroup per CU (a workgroup alone on a CU issues in order: DMA pieces ~100 cycles each, then
    // fragment reads, then MFMAs; a second one fills those gaps: 5-6 % on the 256-tile shapes).  (256-pixel tiles -- eight waves, half the weight DMA per MFMA -- exist for explicit requests
    // only: measured 0-15 % slower than three co-resident 128-pixel workgroups on every layer shape.)
    // (4-pixel-wide maps kept 64-channel tiles at 193-256 workgroups while their padded halo images allowed one
    // workgroup per CU whatever BN was; with the whole-image form the two tie in the step)
    const long wgs64 = tiles_m_of(a) * (a->K / 64);
    return wgs64 <= 256 ? COMBAT_TILE_D128x32 : COMBAT_TILE_D128x64;
}

int conv3x3d_stats_layout(const combat_conv_args *a, int tile, int *rows, int *rows_per_image) {
    const bool per_wg = a->stats_kind & COMBAT_STATS_PER_WORKGROUP;
    if (tile == COMBAT_TILE_S128x64) {
        if (per_wg) {   // one row per persistent workgroup: whatever tiles it was dealt
            if (!ws_applicable(a)) return COMBAT_EINVAL;
            int per;
            ws_grid(tiles_m_of(a), per, *rows);
            *rows_per_image = 0;
            return COMBAT_OK;
        }
        tile = COMBAT_TILE_D128x64;     // same tiles, same rows
    }
    const int bn = tile_bn(tile);
    if (!bn || !applicable(a, bn)) return COMBAT_EINVAL;
    const int tw = geo_tw(a), bm = tile_bm(tile);
    const int th = geo_th(a, bm);
    if (!th) return COMBAT_EINVAL;
    if (per_wg && tile != COMBAT_TILE_D256W64) {   // one row per tile (the wide-wave tile keeps its per-wave rows)
        *rows = tiles_m_of(a, bm);
        *rows_per_image = tw * th == bm ? (a->H * a->W) / bm : 0;
        return COMBAT_OK;
    }
    *rows = tiles_m_of(a, bm) * (bm / 32);   // one row per wave = 32 pixels
    *rows_per_image = ((tw * th) % 32 == 0) ? (a->H * a->W) / 32 : 0;
    return COMBAT_OK;
}

int conv3x3d_launch(const combat_conv_args *a, int tile, hipStream_t st) {
    if (tile == COMBAT_TILE_S128x64) return ws_applicable(a) ? launch_ws(a, st) : COMBAT_EINVAL;
    const int bn = tile_bn(tile);
    if (!bn || !applicable(a, bn)) return COMBAT_EINVAL;
    const int tw = geo_tw(a);
    if (!geo_th(a, tile_bm(tile))) return COMBAT_EINVAL;
    if (tile == COMBAT_TILE_D256x64) {
        if (tw == 16) return launch<64, 16, 8>(a, st);
        return launch<64, 8, 8>(a, st);
    }
    if (tile == COMBAT_TILE_D256W64) return tw == 16 ? launch<64, 16, 4, 64>(a, st) : COMBAT_EINVAL;
    if (bn == 64) {
        if (tw == 16) return launch<64, 16, 4>(a, st);
        if (tw == 8) return launch<64, 8, 4>(a, st);
        return launch<64, 4, 4>(a, st);
    }
    if (tw == 16) return launch<32, 16, 4>(a, st);
    if (tw == 8) return launch<32, 8, 4>(a, st);
    return launch<32, 4, 4>(a, st);
}
