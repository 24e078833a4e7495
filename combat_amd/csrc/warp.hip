// WaNet trigger (reference train_generator_wanet.py:151-157, 196-202, 212; GridGenerator networks/models.py:344-385).
//
// The reference's GridGenerator pools an affine-free InstanceNorm output -- whose spatial mean is 0 by construction --
// so its result is tanh(fc2(lrelu(fc1.bias))) for every input (pinned against the reference module in
// tests/test_oracle_golden.py): a learned constant [2][S][S] warp field.  What remains on the device:
//   combat_grid_head_fwd   field = tanh(W2 lrelu(b1) + b2)
//   combat_wanet_grid      noise_grid = U field U^T (bicubic, align_corners=True, as the [H][S] matrix U of
//                          F.interpolate), grid = clamp(identity (1 - r) + noise_grid r, -1, 1)        [H][H][2]
//   combat_warp_fwd        out[n] = grid_sample(x[index[n]], grid)   bilinear, zeros padding, align_corners=True
//   combat_warp_bwd        sum over images and channels of d_out * d(sample)/d(grid), per pixel, in `groups` partial
//                          buffers (image ranges; deterministic, no atomics)
//   combat_wanet_field_bwd partial buffers -> d_grid -> clamp mask, rescale, + L2 term -> U^T . U -> tanh' ->
//                          gradients of fc2.weight, fc2.bias, fc1.bias (fc1.weight and the encoder receive exactly 0)
// Everything here is latency-bound glue around 8 numbers; the images are fp32 [N][3][H][H] planes as in trigger.hip.
#include "common.hpp"
#include "plan.hpp"

namespace {

constexpr int kMaxField = 32;   // 2 * S * S <= 32  (S <= 4)
constexpr int kMaxNf = 256;

__global__ __launch_bounds__(64) void grid_head_fwd_kernel(const float *__restrict__ b1, const float *__restrict__ w2,
                                                           const float *__restrict__ b2, int nf, int nout,
                                                           float *__restrict__ field) {
    const int o = threadIdx.x;
    if (o >= nout) return;
    float z = b2[o];
    for (int k = 0; k < nf; ++k) {
        const float h = b1[k] > 0.f ? b1[k] : 0.2f * b1[k];
        z = fmaf(w2[o * nf + k], h, z);
    }
    field[o] = tanhf(z);
}

// one thread per pixel (i, j): noise_grid[i][j][c] = sum_ab U[i][a] U[j][b] field[c][a][b]
__global__ __launch_bounds__(256) void wanet_grid_kernel(const float *__restrict__ field, const float *__restrict__ U, int S,
                                                         int H, float rescale, float *__restrict__ noise_grid,
                                                         float *__restrict__ grid) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= H * H) return;
    const int i = pix / H, j = pix - i * H;
    const float ax = -1.f + 2.f * (float)j / (float)(H - 1), ay = -1.f + 2.f * (float)i / (float)(H - 1);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        float v = 0.f;
        for (int a = 0; a < S; ++a) {
            float r = 0.f;
            for (int b = 0; b < S; ++b) r = fmaf(U[j * S + b], field[(c * S + a) * S + b], r);
            v = fmaf(U[i * S + a], r, v);
        }
        noise_grid[pix * 2 + c] = v;
        // identity grid: torch.linspace(-1, 1, H) -- [..., 0] the column (x), [..., 1] the row (y)
        const float raw = (c == 0 ? ax : ay) * (1.f - rescale) + v * rescale;
        grid[pix * 2 + c] = fminf(fmaxf(raw, -1.f), 1.f);
    }
}

struct Taps {
    int x0, y0;
    float tx, ty;
};

__device__ __forceinline__ Taps taps_of(float gx, float gy, int H) {
    const float ix = (gx + 1.f) * 0.5f * (float)(H - 1), iy = (gy + 1.f) * 0.5f * (float)(H - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    return Taps{(int)fx, (int)fy, ix - fx, iy - fy};
}

__device__ __forceinline__ float tap(const float *__restrict__ p, int x, int y, int H) {
    return ((unsigned)x < (unsigned)H && (unsigned)y < (unsigned)H) ? p[y * H + x] : 0.f;
}

// grid: [H][H][2] shared by the batch (grid_stride 0) or one per image (grid_stride H*H*2)
__global__ __launch_bounds__(256) void warp_fwd_kernel(const float *__restrict__ x, const int *__restrict__ index,
                                                       const float *__restrict__ grid, long grid_stride, int H,
                                                       float *__restrict__ out) {
    const int pix = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
    if (pix >= H * H) return;
    const float *g = grid + (long)n * grid_stride + pix * 2;
    const Taps t = taps_of(g[0], g[1], H);
    const int src = index ? index[n] : n;
    const float w00 = (1.f - t.tx) * (1.f - t.ty), w10 = t.tx * (1.f - t.ty), w01 = (1.f - t.tx) * t.ty, w11 = t.tx * t.ty;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float *p = x + ((long)src * 3 + c) * H * H;
        const float v = w00 * tap(p, t.x0, t.y0, H) + w10 * tap(p, t.x0 + 1, t.y0, H) + w01 * tap(p, t.x0, t.y0 + 1, H) +
                        w11 * tap(p, t.x0 + 1, t.y0 + 1, H);
        out[((long)n * 3 + c) * H * H + pix] = v;
    }
}

// partial[g][pix][2] = sum over the images of group g and the 3 channels of (d_out + d_out2) * d(sample)/d(grid)
__global__ __launch_bounds__(256) void warp_bwd_kernel(const float *__restrict__ x, const float *__restrict__ d_out,
                                                       const float *__restrict__ d_out2, const float *__restrict__ grid,
                                                       long grid_stride, int n_img, int per_group, int H,
                                                       float *__restrict__ partial) {
    const int pix = blockIdx.x * 256 + threadIdx.x, grp = blockIdx.y;
    if (pix >= H * H) return;
    const int n0 = grp * per_group, n1 = n0 + per_group < n_img ? n0 + per_group : n_img;
    float agx = 0.f, agy = 0.f;
    for (int n = n0; n < n1; ++n) {
        const float *g = grid + (long)n * grid_stride + pix * 2;
        const Taps t = taps_of(g[0], g[1], H);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const long o = ((long)n * 3 + c) * H * H;
            const float *p = x + o;
            float d = d_out[o + pix];
            if (d_out2) d += d_out2[o + pix];
            const float v00 = tap(p, t.x0, t.y0, H), v10 = tap(p, t.x0 + 1, t.y0, H), v01 = tap(p, t.x0, t.y0 + 1, H),
                        v11 = tap(p, t.x0 + 1, t.y0 + 1, H);
            agx = fmaf(d, (1.f - t.ty) * (v10 - v00) + t.ty * (v11 - v01), agx);
            agy = fmaf(d, (1.f - t.tx) * (v01 - v00) + t.tx * (v11 - v10), agy);
        }
    }
    const float k = 0.5f * (float)(H - 1);     // d(ix) / d(gx), align_corners=True
    partial[((long)grp * H * H + pix) * 2] = agx * k;
    partial[((long)grp * H * H + pix) * 2 + 1] = agy * k;
}

// d_x[n][c][tap] += weight(tap) * d_out[n][c][pix]: the transpose of warp_fwd_kernel (fp32 atomics into a zeroed buffer;
// the training path never needs it -- the images are data -- it completes grid_sample's backward for callers that do)
__global__ __launch_bounds__(256) void warp_bwd_input_kernel(const float *__restrict__ d_out, const float *__restrict__ grid,
                                                             long grid_stride, int H, float *__restrict__ d_x) {
    const int pix = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
    if (pix >= H * H) return;
    const float *g = grid + (long)n * grid_stride + pix * 2;
    const Taps t = taps_of(g[0], g[1], H);
    const float w[4] = {(1.f - t.tx) * (1.f - t.ty), t.tx * (1.f - t.ty), (1.f - t.tx) * t.ty, t.tx * t.ty};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const long o = ((long)n * 3 + c) * H * H;
        const float d = d_out[o + pix];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int xx = t.x0 + (q & 1), yy = t.y0 + (q >> 1);
            if ((unsigned)xx < (unsigned)H && (unsigned)yy < (unsigned)H) atomicAdd(d_x + o + yy * H + xx, w[q] * d);
        }
    }
}

struct FieldBwdArgs {
    const float *partial, *noise_grid, *U, *field, *b1, *w2;
    int groups, S, H, nf, nout;
    float rescale, l2_scale;
    float *d_b1, *d_w2, *d_b2, *d_field;
};

template <int S>
__global__ __launch_bounds__(256) void wanet_field_bwd_kernel(const FieldBwdArgs a) {
    constexpr int NOUT = 2 * S * S;
    __shared__ float red[256];
    __shared__ float dfield[NOUT], dz[NOUT], hval[kMaxNf];
    const int tid = threadIdx.x, H = a.H, HH = H * H;
    float acc[NOUT];
#pragma unroll
    for (int q = 0; q < NOUT; ++q) acc[q] = 0.f;
    const float l2k = a.l2_scale * 2.f / (float)(HH * 2);   // d/dv of l2_scale * mean_{H,H,2}(v^2) (the batch holds B equal copies)
    for (int pix = tid; pix < HH; pix += 256) {
        const int i = pix / H, j = pix - i * H;
        const float ax = -1.f + 2.f * (float)j / (float)(H - 1), ay = -1.f + 2.f * (float)i / (float)(H - 1);
        float ui[S], uj[S];
#pragma unroll
        for (int q = 0; q < S; ++q) {
            ui[q] = a.U[i * S + q];
            uj[q] = a.U[j * S + q];
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float dg = 0.f;
            for (int g = 0; g < a.groups; ++g) dg += a.partial[((long)g * HH + pix) * 2 + c];
            const float v = a.noise_grid[pix * 2 + c];
            const float raw = (c == 0 ? ax : ay) * (1.f - a.rescale) + v * a.rescale;
            const float dv = (raw >= -1.f && raw <= 1.f ? dg * a.rescale : 0.f) + l2k * v;   // torch.clamp passes the bounds
#pragma unroll
            for (int aa = 0; aa < S; ++aa)
#pragma unroll
                for (int b = 0; b < S; ++b) acc[(c * S + aa) * S + b] = fmaf(ui[aa] * uj[b], dv, acc[(c * S + aa) * S + b]);
        }
    }
#pragma unroll
    for (int q = 0; q < NOUT; ++q) {      // block sums, fixed order
        red[tid] = acc[q];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        if (tid == 0) dfield[q] = red[0];
        __syncthreads();
    }
    if (tid < NOUT) {
        const float f = a.field[tid];
        dz[tid] = dfield[tid] * (1.f - f * f);
        if (a.d_field) a.d_field[tid] = dfield[tid];
        a.d_b2[tid] = dz[tid];
    }
    for (int k = tid; k < a.nf; k += 256) hval[k] = a.b1[k] > 0.f ? a.b1[k] : 0.2f * a.b1[k];
    __syncthreads();
    for (int e = tid; e < NOUT * a.nf; e += 256) a.d_w2[e] = dz[e / a.nf] * hval[e % a.nf];
    for (int k = tid; k < a.nf; k += 256) {
        float dh = 0.f;
        for (int o = 0; o < NOUT; ++o) dh = fmaf(a.w2[o * a.nf + k], dz[o], dh);
        a.d_b1[k] = dh * (a.b1[k] > 0.f ? 1.f : 0.2f);
    }
}

}  // namespace

extern "C" int combat_grid_head_fwd(const float *fc1_bias, const float *fc2_weight, const float *fc2_bias, int32_t nf,
                                    int32_t nout, float *field, void *stream) {
    COMBAT_PLAN_HOOK(combat_grid_head_fwd, fc1_bias, fc2_weight, fc2_bias, nf, nout, field);
    if (!fc1_bias || !fc2_weight || !fc2_bias || !field || nf <= 0 || nf > kMaxNf || nout <= 0 || nout > kMaxField) return COMBAT_EINVAL;
    COMBAT_LAUNCH(grid_head_fwd_kernel, dim3(1), dim3(64), 0, as_stream(stream), fc1_bias, fc2_weight, fc2_bias, nf, nout, field);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_wanet_grid(const float *field, const float *U, int32_t S, int32_t H, float rescale, float *noise_grid,
                                 float *grid, void *stream) {
    COMBAT_PLAN_HOOK(combat_wanet_grid, field, U, S, H, rescale, noise_grid, grid);
    if (!field || !U || !noise_grid || !grid || S < 1 || 2 * S * S > kMaxField || H < 2) return COMBAT_EINVAL;
    COMBAT_LAUNCH(wanet_grid_kernel, dim3((H * H + 255) / 256), dim3(256), 0, as_stream(stream), field, U, S, H, rescale,
                       noise_grid, grid);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_warp_fwd(const float *x, const int32_t *src_index, const float *grid, int32_t per_image_grid, int32_t n,
                               int32_t H, float *out, void *stream) {
    COMBAT_PLAN_HOOK(combat_warp_fwd, x, src_index, grid, per_image_grid, n, H, out);
    if (!x || !grid || !out || n < 0 || H < 2) return COMBAT_EINVAL;
    if (n == 0) return COMBAT_OK;
    COMBAT_LAUNCH(warp_fwd_kernel, dim3((H * H + 255) / 256, n), dim3(256), 0, as_stream(stream), x, src_index, grid,
                       per_image_grid ? (long)H * H * 2 : 0L, H, out);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_warp_bwd(const float *x, const float *d_out, const float *d_out2, const float *grid,
                               int32_t per_image_grid, int32_t n, int32_t H, int32_t groups, float *partial, void *stream) {
    COMBAT_PLAN_HOOK(combat_warp_bwd, x, d_out, d_out2, grid, per_image_grid, n, H, groups, partial);
    if (!x || !d_out || !grid || !partial || n <= 0 || H < 2 || groups < 1) return COMBAT_EINVAL;
    const int per = (n + groups - 1) / groups;
    COMBAT_LAUNCH(warp_bwd_kernel, dim3((H * H + 255) / 256, groups), dim3(256), 0, as_stream(stream), x, d_out, d_out2, grid,
                       per_image_grid ? (long)H * H * 2 : 0L, n, per, H, partial);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_warp_bwd_input(const float *d_out, const float *grid, int32_t per_image_grid, int32_t n, int32_t H,
                                     float *d_x, void *stream) {
    COMBAT_PLAN_HOOK(combat_warp_bwd_input, d_out, grid, per_image_grid, n, H, d_x);
    if (!d_out || !grid || !d_x || n < 0 || H < 2) return COMBAT_EINVAL;
    if (n == 0) return COMBAT_OK;
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(d_x, 0, (size_t)n * 3 * H * H * sizeof(float), st) != hipSuccess) return COMBAT_ELAUNCH;
    COMBAT_LAUNCH(warp_bwd_input_kernel, dim3((H * H + 255) / 256, n), dim3(256), 0, st, d_out, grid,
                       per_image_grid ? (long)H * H * 2 : 0L, H, d_x);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_wanet_field_bwd(const float *partial, int32_t groups, const float *noise_grid, const float *U, int32_t S,
                                      int32_t H, float rescale, float l2_scale, const float *field, const float *fc1_bias,
                                      const float *fc2_weight, int32_t nf, float *d_fc1_bias, float *d_fc2_weight,
                                      float *d_fc2_bias, float *d_field, void *stream) {
    COMBAT_PLAN_HOOK(combat_wanet_field_bwd, partial, groups, noise_grid, U, S, H, rescale, l2_scale, field, fc1_bias, fc2_weight, nf, d_fc1_bias, d_fc2_weight, d_fc2_bias, d_field);
    if (!partial || !noise_grid || !U || !field || !fc1_bias || !fc2_weight || !d_fc1_bias || !d_fc2_weight || !d_fc2_bias)
        return COMBAT_EINVAL;
    if (S < 1 || 2 * S * S > kMaxField || H < 2 || groups < 1 || nf <= 0 || nf > kMaxNf) return COMBAT_EINVAL;
    FieldBwdArgs a{partial, noise_grid, U, field, fc1_bias, fc2_weight, groups, S, H, nf, 2 * S * S,
                   rescale, l2_scale, d_fc1_bias, d_fc2_weight, d_fc2_bias, d_field};
    hipStream_t st = as_stream(stream);
    switch (S) {
        case 1: COMBAT_LAUNCH(wanet_field_bwd_kernel<1>, dim3(1), dim3(256), 0, st, a); break;
        case 2: COMBAT_LAUNCH(wanet_field_bwd_kernel<2>, dim3(1), dim3(256), 0, st, a); break;
        case 3: COMBAT_LAUNCH(wanet_field_bwd_kernel<3>, dim3(1), dim3(256), 0, st, a); break;
        default: COMBAT_LAUNCH(wanet_field_bwd_kernel<4>, dim3(1), dim3(256), 0, st, a); break;
    }
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}
