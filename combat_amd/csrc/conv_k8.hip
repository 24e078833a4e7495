// 3x3 / stride 1 / pad 1 convolutions with EIGHT output channels (the c8 image layout): the generator's output layer
// (models.py UnetGenerator: 64 -> 3 behind InstanceNorm + LeakyReLU, tanh) and the input gradient of a classifier's stem
// (64-channel dY -> the 3-channel image gradient).  0.15 GFLOP over a 16.8-MB input at 128 images of 32 x 32: HBM work.
// The generic gather GEMM (conv_gemm_kernel<128, 16>) spent 28-36 us on it -- it fetches and normalises every input
// element once per tap through registers, for a 16-wide output tile.  Here:
//   * a workgroup owns 16 x 8 pixels of one image; the 18 x 10 halo patch of a 64-channel chunk goes through registers
//     ONCE (prologue applied there: per-channel or per-(image, channel) scale / shift + LeakyReLU, padding stays zero)
//     into LDS, 16-byte slots rotated by the patch column so that a fragment read is conflict-free;
//   * the weights of a chunk are 18 MFMA A-fragments (rows = output channels, 8 real of 16) held in registers;
//   * a wave computes its 32 pixels as 2 x 18 MFMA 16x16x32, each fed by ONE ds_read_b128 of shifted patch pixels;
//   * lanes 0..31 hold the eight channels of the wave's pixels: bias, tanh, two 8-byte stores per pixel.
// Input gradients (mode 1) are the same convolution over dY with the taps mirrored (the dgrad operand layout of
// combat_pack_weights has k = tap * channels(dY) + n with rows = channels(dX): the forward layout of that convolution).
#include "conv_common.hpp"
#include "plan.hpp"

namespace {

struct K8Params {
    combat_conv_args a;
    int cin;                  // reduction channels (the source tensor's; in mode 1 the source is dY)
    int tiles_x, tiles_y;     // 16 x 8-pixel tiles per image
};

constexpr int kPatchPx = 18 * 10;
constexpr int kK8Smem = kPatchPx * 128;   // one 64-channel chunk of the patch

template <bool PRO>
__global__ __launch_bounds__(256, 2) void conv_k8_kernel(const K8Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const combat_conv_args &a = p.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W, cin = p.cin;
    const int tx = blockIdx.x % p.tiles_x, ty = (blockIdx.x / p.tiles_x) % p.tiles_y, img = blockIdx.x / (p.tiles_x * p.tiles_y);
    const __bf16 *__restrict__ src = reinterpret_cast<const __bf16 *>(a.src);
    const __bf16 *__restrict__ wp = reinterpret_cast<const __bf16 *>(a.wpack);
    const int fr = lane & 15, fq = lane >> 4;
    f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};

    for (int cc = 0; cc < cin; cc += 64) {
        // ---- this chunk's weights: fragment (tap, half) = rows fr (zero beyond the eight real ones), k-block fq
        bf16x8_t fw[9][2];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int tap = a.mode == 0 ? t : 8 - t;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x4_t v = {0u, 0u, 0u, 0u};
                if (fr < 8) v = *reinterpret_cast<const u32x4_t *>(wp + (size_t)fr * a.kpad + tap * cin + cc + h * 32 + fq * 8);
                fw[t][h] = __builtin_bit_cast(bf16x8_t, v);
            }
        }
        // ---- the halo patch of this chunk, through registers (all loads of a thread first, then the prologue)
        constexpr int NI = (kPatchPx * 8 + 255) / 256;
        u32x4_t ld[NI];
        int chan[NI];
        bool ok[NI];
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int item = q * 256 + tid, px = item >> 3, slot = item & 7;
            const int hy = px / 18, hx = px - hy * 18;
            const int gy = ty * 8 + hy - 1, gx = tx * 16 + hx - 1;
            chan[q] = cc + (((slot - hx) & 7) << 3);
            ok[q] = px < kPatchPx && gy >= 0 && gy < H && gx >= 0 && gx < W;
            ld[q] = u32x4_t{0u, 0u, 0u, 0u};
            if (ok[q]) ld[q] = *reinterpret_cast<const u32x4_t *>(src + (((size_t)img * H + gy) * W + gx) * cin + chan[q]);
        }
        if (cc) __syncthreads();      // the previous chunk's fragment reads are done
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int item = q * 256 + tid;
            if (PRO && ok[q]) {
                float v[8], sc[8], sh[8];
                unpack8v(ld[q], v);
                const size_t g = (size_t)img * a.pro_group_stride + chan[q];
                if (a.pro_scale) {
                    load8f(a.pro_scale + g, sc);
                    load8f(a.pro_shift + g, sh);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
                }
                if (a.pro_act) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.pro_slope;
                }
                ld[q] = pack8v(v);
            }
            if (item < kPatchPx * 8) *reinterpret_cast<u32x4_t *>(smem + item * 16) = ld[q];
        }
        __syncthreads();
        // ---- 2 pixel fragments (tile rows 2 wid, 2 wid + 1) x 9 taps x 2 halves
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3, dx = t % 3;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int hx = fr + dx, px = (2 * wid + j + dy) * 18 + hx;
                    const int slot = (h * 4 + fq + hx) & 7;
                    const bf16x8_t fp = *reinterpret_cast<const bf16x8_t *>(smem + (px * 8 + slot) * 16);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[t][h], fp, acc[j], 0, 0, 0);
                }
            }
        }
    }
    // ---- lanes 0..31: channels 4 fq .. 4 fq + 3 of pixel column fr
    if (fq < 2) {
        float b[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = a.bias[fq * 4 + i];
        }
        __bf16 *__restrict__ dst = reinterpret_cast<__bf16 *>(a.dst);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gy = ty * 8 + 2 * wid + j, gx = tx * 16 + fr;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = acc[j][i] + b[i];
                if (a.tanh_out) v[i] = tanhf(v[i]);
            }
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
            const bf16x4_t o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            *reinterpret_cast<bf16x4_t *>(dst + (((size_t)img * H + gy) * W + gx) * 8 + fq * 4) = o;
        }
    }
}

}  // namespace

// what the kernel covers: eight output channels, a reduction over 64-channel chunks, stride 1, whole 16 x 8 tiles, no
// residual / mask / statistics / second output (those launches stay with the general kernels)
bool conv_k8_ok(const combat_conv_args *a) {
    if (a->R != 3 || a->S != 3 || a->stride != 1 || a->pad != 1 || a->P != a->H || a->Q != a->W) return false;
    const int cout = a->K, cin = a->C;      // (both modes describe src -> dst: in mode 1 src is dY, dst dX)
    if (cout != 8 || cin < 64 || (cin & 63) || a->rows_pad < 8 || a->kpad < 9 * cin) return false;
    if ((a->W & 15) || (a->H & 7) || !a->dst) return false;
    if (a->add_pre || a->add_post || a->mask_x || a->stats_kind || a->act_dst) return false;
    if (a->mode == 1 && (a->pro_scale || a->pro_act)) return false;
    if (a->pro_scale && a->pro_group_stride != 0 && a->pro_group_stride != cin) return false;
    return true;
}

int conv_k8_launch(const combat_conv_args *a, hipStream_t st) {
    if (!conv_k8_ok(a)) return COMBAT_EINVAL;
    K8Params p;
    p.a = *a;
    p.cin = a->C;
    p.tiles_x = a->W / 16;
    p.tiles_y = a->H / 8;
    const dim3 grid((unsigned)(a->N * p.tiles_x * p.tiles_y));
    if (a->pro_scale || a->pro_act) COMBAT_LAUNCH(conv_k8_kernel<true>, grid, dim3(256), kK8Smem, st, p);
    else COMBAT_LAUNCH(conv_k8_kernel<false>, grid, dim3(256), kK8Smem, st, p);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}
