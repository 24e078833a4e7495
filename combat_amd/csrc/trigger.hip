// Trigger synthesis, on-device augmentation and the detector's DCT input: the image-sized
// (B x 3 x hw x hw) part of the step.  One workgroup per image; everything is staged in LDS as
// fp32 planes and expressed with dense hw x hw matrix products (low-pass P, reflect-padded blur
// Kb, DCT D), which makes every backward the same kernel with transposed factors.
//
//   out = Kb * clamp(x + rate * (P * noise * P^T), -1, 1) * Kb^T
//
// Replaces: low_freq (train_generator.py:47-55 + utils/dct.py:13-111), torch.clamp mix (:192,:225),
// T.GaussianBlur (:165,:194,:226), MSELoss term (:234), PostTensorTransform (utils/dataloader.py:45-60,
// kornia), dct_2d(byte()) (train_generator.py:245).
#include "common.hpp"

namespace {

// C[i][j] = sum_k A[i][k] * B[k][j]  (tb: use B[j][k]; ta: use A[k][i]); all hw x hw fp32 in LDS
template <bool TA, bool TB>
__device__ __forceinline__ void mm(const float *A, const float *B, float *Cm, int hw, int tid, int nthr) {
    for (int o = tid; o < hw * hw; o += nthr) {
        const int i = o / hw, j = o - i * hw;
        float s = 0.f;
        for (int k = 0; k < hw; ++k) {
            const float av = TA ? A[k * hw + i] : A[i * hw + k];
            const float bv = TB ? B[j * hw + k] : B[k * hw + j];
            s = fmaf(av, bv, s);
        }
        Cm[o] = s;
    }
}

// reflect-padded 3-tap blur as a matrix: out[i] = k0*in[r(i-1)] + k1*in[i] + k2*in[r(i+1)]
__device__ __forceinline__ void build_blur(float *Kb, const float *k1, int hw, int tid, int nthr) {
    for (int o = tid; o < hw * hw; o += nthr) Kb[o] = 0.f;
    __syncthreads();
    for (int i = tid; i < hw; i += nthr) {
        const int im = i > 0 ? i - 1 : 1, ip = i + 1 < hw ? i + 1 : hw - 2;
        Kb[i * hw + im] += k1[0];
        Kb[i * hw + i] += k1[1];
        Kb[i * hw + ip] += k1[2];
    }
    __syncthreads();
}

__device__ __forceinline__ uint4 hilo_px(float r, float g, float b) {
    const float hr = round_bf16(r), hg = round_bf16(g), hb = round_bf16(b);
    uint4 u;
    u.x = pack_bf16x2(hr, hg);
    u.y = pack_bf16x2(hb, r - hr);
    u.z = pack_bf16x2(g - hg, b - hb);
    u.w = 0;
    return u;
}

// dynamic LDS: P, Kb, two scratch planes A/B, three result planes R (7 * hw*hw floats)
__global__ __launch_bounds__(256) void trigger_fwd_kernel(const float *__restrict__ x, const __bf16 *__restrict__ noise,
                                                          const float *__restrict__ P, const float *__restrict__ k1,
                                                          float rate, int hw, float *__restrict__ out,
                                                          uint4 *__restrict__ out_c8, float *__restrict__ mse) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int hw2 = hw * hw, tid = threadIdx.x, img = blockIdx.x;
    float *Pm = sm, *Kb = sm + hw2, *A = sm + 2 * hw2, *B = sm + 3 * hw2, *R = sm + 4 * hw2;
    for (int o = tid; o < hw2; o += 256) Pm[o] = P[o];
    build_blur(Kb, k1, hw, tid, 256);
    const float *xi = x + (long)img * 3 * hw2;
    for (int c = 0; c < 3; ++c) {
        for (int o = tid; o < hw2; o += 256)
            A[o] = (float)noise[((long)img * hw2 + o) * 8 + c];
        __syncthreads();
        mm<false, false>(Pm, A, B, hw, tid, 256);  // P * N
        __syncthreads();
        mm<false, true>(B, Pm, A, hw, tid, 256);   // (P N) * P^T
        __syncthreads();
        for (int o = tid; o < hw2; o += 256) A[o] = fminf(fmaxf(fmaf(A[o], rate, xi[c * hw2 + o]), -1.f), 1.f);
        __syncthreads();
        mm<false, false>(Kb, A, B, hw, tid, 256);  // Kb * bd
        __syncthreads();
        mm<false, true>(B, Kb, R + c * hw2, hw, tid, 256);  // ... * Kb^T
        __syncthreads();
    }
    float se = 0.f;
    for (int o = tid; o < 3 * hw2; o += 256) {
        const float v = R[o];
        out[(long)img * 3 * hw2 + o] = v;
        const float d = v - xi[o];
        se = fmaf(d, d, se);
    }
    if (out_c8)
        for (int o = tid; o < hw2; o += 256) out_c8[(long)img * hw2 + o] = hilo_px(R[o], R[hw2 + o], R[2 * hw2 + o]);
    if (mse) {
        A[tid] = se;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) A[tid] += A[tid + s];
            __syncthreads();
        }
        if (tid == 0) mse[img] = A[0];
    }
}

// d_noise = rate * P * ( clampmask .* (Kb^T * (d_out + 2*l2*(out-x)) * Kb) ) * P      (P symmetric)
// dynamic LDS: P, Kb, scratch A/B/G, results R (8 * hw*hw floats)
__global__ __launch_bounds__(256) void trigger_bwd_kernel(const float *__restrict__ x, const __bf16 *__restrict__ noise,
                                                          const float *__restrict__ P, const float *__restrict__ k1,
                                                          float rate, int hw, const float *__restrict__ d_out,
                                                          const float *__restrict__ outp, float l2_scale,
                                                          int pre_tanh, uint4 *__restrict__ d_noise) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int hw2 = hw * hw, tid = threadIdx.x, img = blockIdx.x;
    float *Pm = sm, *Kb = sm + hw2, *A = sm + 2 * hw2, *B = sm + 3 * hw2, *G = sm + 4 * hw2, *R = sm + 5 * hw2;
    for (int o = tid; o < hw2; o += 256) Pm[o] = P[o];
    build_blur(Kb, k1, hw, tid, 256);
    const float *xi = x + (long)img * 3 * hw2;
    for (int c = 0; c < 3; ++c) {
        for (int o = tid; o < hw2; o += 256) {
            A[o] = (float)noise[((long)img * hw2 + o) * 8 + c];
            const long go = ((long)img * 3 + c) * hw2 + o;
            float g = d_out ? d_out[go] : 0.f;
            if (l2_scale != 0.f) g = fmaf(2.f * l2_scale, outp[go] - xi[c * hw2 + o], g);
            G[o] = g;
        }
        __syncthreads();
        mm<false, false>(Pm, A, B, hw, tid, 256);
        __syncthreads();
        mm<false, true>(B, Pm, A, hw, tid, 256);   // A = P N P^T  (pre-clamp value needs x + rate*A)
        __syncthreads();
        mm<true, false>(Kb, G, B, hw, tid, 256);   // B = Kb^T g
        __syncthreads();
        mm<false, false>(B, Kb, G, hw, tid, 256);  // G = Kb^T g Kb
        __syncthreads();
        for (int o = tid; o < hw2; o += 256) {
            const float v = fmaf(A[o], rate, xi[c * hw2 + o]);
            G[o] = (v >= -1.f && v <= 1.f) ? G[o] * rate : 0.f;
        }
        __syncthreads();
        mm<false, false>(Pm, G, B, hw, tid, 256);
        __syncthreads();
        mm<false, false>(B, Pm, R + c * hw2, hw, tid, 256);  // P g P
        __syncthreads();
    }
    for (int o = tid; o < hw2; o += 256) {
        uint4 u;
        if (pre_tanh) {  // noise = tanh(z): hand back the gradient w.r.t. z
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float t = (float)noise[((long)img * hw2 + o) * 8 + c];
                R[c * hw2 + o] *= 1.f - t * t;
            }
        }
        u.x = pack_bf16x2(R[o], R[hw2 + o]);
        u.y = pack_bf16x2(R[2 * hw2 + o], 0.f);
        u.z = 0;
        u.w = 0;
        d_noise[(long)img * hw2 + o] = u;
    }
}

// ------------------------------------------------------------------ augmentation
struct AugGeom {
    float ca, sa, cx, cy;
    int ox, oy, flip, rot;
};

__device__ __forceinline__ AugGeom aug_geom(const float *params, int img, int hw) {
    AugGeom g;
    g.ox = g.oy = g.flip = g.rot = 0;
    g.ca = 1.f;
    g.sa = 0.f;
    g.cx = g.cy = 0.5f * (hw - 1);
    if (params) {
        const float *q = params + 4 * img;
        g.ox = (int)q[0];
        g.oy = (int)q[1];
        g.rot = q[2] != 0.f;
        g.ca = cosf(q[2]);
        g.sa = sinf(q[2]);
        g.flip = q[3] != 0.f;
    }
    return g;
}

// value of the cropped image Ic(u, v) = I(u + ox, v + oy) inside the hw window, zero outside
__device__ __forceinline__ float crop_at(const float *plane, int hw, const AugGeom &g, int u, int v) {
    if ((unsigned)u >= (unsigned)hw || (unsigned)v >= (unsigned)hw) return 0.f;
    const int sx = u + g.ox, sy = v + g.oy;
    if ((unsigned)sx >= (unsigned)hw || (unsigned)sy >= (unsigned)hw) return 0.f;
    return plane[sy * hw + sx];
}

__global__ __launch_bounds__(256) void augment_fwd_kernel(const float *__restrict__ x, const int *__restrict__ index,
                                                          const float *__restrict__ params, int hw,
                                                          uint4 *__restrict__ out_c8, float *__restrict__ out_f32) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int hw2 = hw * hw, tid = threadIdx.x, img = blockIdx.x;
    const int srci = index ? index[img] : img;
    for (int o = tid; o < 3 * hw2; o += 256) sm[o] = x[(long)srci * 3 * hw2 + o];
    __syncthreads();
    const AugGeom g = aug_geom(params, img, hw);
    for (int o = tid; o < hw2; o += 256) {
        const int yo = o / hw, xo0 = o - yo * hw;
        const int xo = g.flip ? hw - 1 - xo0 : xo0;  // flip is the last stage
        float v[3];
        if (g.rot) {
            const float dx = xo - g.cx, dy = yo - g.cy;
            const float sx = g.ca * dx - g.sa * dy + g.cx, sy = g.sa * dx + g.ca * dy + g.cy;
            const float fx = floorf(sx), fy = floorf(sy);
            const int x0 = (int)fx, y0 = (int)fy;
            const float ax = sx - fx, ay = sy - fy;
            for (int c = 0; c < 3; ++c) {
                const float *pl = sm + c * hw2;
                v[c] = (1.f - ay) * ((1.f - ax) * crop_at(pl, hw, g, x0, y0) + ax * crop_at(pl, hw, g, x0 + 1, y0)) +
                       ay * ((1.f - ax) * crop_at(pl, hw, g, x0, y0 + 1) + ax * crop_at(pl, hw, g, x0 + 1, y0 + 1));
            }
        } else {
            for (int c = 0; c < 3; ++c) v[c] = crop_at(sm + c * hw2, hw, g, xo, yo);
        }
        out_c8[(long)img * hw2 + o] = hilo_px(v[0], v[1], v[2]);
        if (out_f32)
            for (int c = 0; c < 3; ++c) out_f32[((long)img * 3 + c) * hw2 + o] = v[c];
    }
}

__device__ __forceinline__ void crop_scatter(float *plane, int hw, const AugGeom &g, int u, int v, float val) {
    if ((unsigned)u >= (unsigned)hw || (unsigned)v >= (unsigned)hw) return;
    const int sx = u + g.ox, sy = v + g.oy;
    if ((unsigned)sx >= (unsigned)hw || (unsigned)sy >= (unsigned)hw) return;
    atomicAdd(plane + sy * hw + sx, val);  // LDS atomic
}

__global__ __launch_bounds__(256) void augment_bwd_kernel(const __bf16 *__restrict__ d_c8, int cch,
                                                          const float *__restrict__ params, int hw,
                                                          float *__restrict__ d_x, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int hw2 = hw * hw, tid = threadIdx.x, img = blockIdx.x;
    for (int o = tid; o < 3 * hw2; o += 256) sm[o] = accumulate ? d_x[(long)img * 3 * hw2 + o] : 0.f;
    __syncthreads();
    const AugGeom g = aug_geom(params, img, hw);
    for (int o = tid; o < hw2; o += 256) {
        const int yo = o / hw, xo0 = o - yo * hw;
        const int xo = g.flip ? hw - 1 - xo0 : xo0;
        const __bf16 *gp = d_c8 + ((long)img * hw2 + o) * cch;
        const float gv[3] = {(float)gp[0], (float)gp[1], (float)gp[2]};
        if (g.rot) {
            const float dx = xo - g.cx, dy = yo - g.cy;
            const float sx = g.ca * dx - g.sa * dy + g.cx, sy = g.sa * dx + g.ca * dy + g.cy;
            const float fx = floorf(sx), fy = floorf(sy);
            const int x0 = (int)fx, y0 = (int)fy;
            const float ax = sx - fx, ay = sy - fy;
            for (int c = 0; c < 3; ++c) {
                float *pl = sm + c * hw2;
                crop_scatter(pl, hw, g, x0, y0, (1.f - ay) * (1.f - ax) * gv[c]);
                crop_scatter(pl, hw, g, x0 + 1, y0, (1.f - ay) * ax * gv[c]);
                crop_scatter(pl, hw, g, x0, y0 + 1, ay * (1.f - ax) * gv[c]);
                crop_scatter(pl, hw, g, x0 + 1, y0 + 1, ay * ax * gv[c]);
            }
        } else {
            for (int c = 0; c < 3; ++c) crop_scatter(sm + c * hw2, hw, g, xo, yo, gv[c]);
        }
    }
    __syncthreads();
    for (int o = tid; o < 3 * hw2; o += 256) d_x[(long)img * 3 * hw2 + o] = sm[o];
}

// ------------------------------------------------------------------ DCT of the uint8-truncated image
__global__ __launch_bounds__(256) void dct_u8_kernel(const float *__restrict__ x, const float *__restrict__ D, int hw,
                                                     uint4 *__restrict__ out_c8) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int hw2 = hw * hw, tid = threadIdx.x, img = blockIdx.x;
    float *Dm = sm, *A = sm + hw2, *B = sm + 4 * hw2;
    for (int o = tid; o < hw2; o += 256) Dm[o] = D[o];
    for (int o = tid; o < 3 * hw2; o += 256) {
        const float q = (x[(long)img * 3 * hw2 + o] + 1.f) / 2.f * 255.f;
        A[o] = (float)(unsigned char)(int)q;  // .byte(): truncate toward zero, wrap mod 256
    }
    __syncthreads();
    for (int c = 0; c < 3; ++c) mm<false, false>(Dm, A + c * hw2, B + c * hw2, hw, tid, 256);
    __syncthreads();
    for (int c = 0; c < 3; ++c) mm<false, true>(B + c * hw2, Dm, A + c * hw2, hw, tid, 256);
    __syncthreads();
    for (int o = tid; o < hw2; o += 256) out_c8[(long)img * hw2 + o] = hilo_px(A[o], A[hw2 + o], A[2 * hw2 + o]);
}

template <typename K>
int set_smem(K kern, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               bytes) == hipSuccess
               ? COMBAT_OK
               : COMBAT_ELAUNCH;
}

}  // namespace

extern "C" int combat_trigger_fwd(const float *x, const void *noise, const float *P, const float *k1,
                                  float noise_rate, int32_t n, int32_t hw, float *out, void *out_c8,
                                  float *mse_partial, void *stream) {
    if (!x || !noise || !P || !k1 || !out || n < 0 || hw < 16 || hw > 64) return COMBAT_EINVAL;
    if (n == 0) return COMBAT_OK;
    const int bytes = 7 * hw * hw * 4;
    if (set_smem(trigger_fwd_kernel, bytes)) return COMBAT_ELAUNCH;
    hipLaunchKernelGGL(trigger_fwd_kernel, dim3(n), dim3(256), bytes, as_stream(stream), x,
                       reinterpret_cast<const __bf16 *>(noise), P, k1, noise_rate, hw, out,
                       reinterpret_cast<uint4 *>(out_c8), mse_partial);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_trigger_bwd(const float *x, const void *noise, const float *P, const float *k1,
                                  float noise_rate, int32_t n, int32_t hw, const float *d_out, const float *out,
                                  float l2_scale, int32_t pre_tanh, void *d_noise, void *stream) {
    if (!x || !noise || !P || !k1 || !d_noise || n < 0 || hw < 16 || hw > 64) return COMBAT_EINVAL;
    if (l2_scale != 0.f && !out) return COMBAT_EINVAL;
    if (n == 0) return COMBAT_OK;
    const int bytes = 8 * hw * hw * 4;
    if (set_smem(trigger_bwd_kernel, bytes)) return COMBAT_ELAUNCH;
    hipLaunchKernelGGL(trigger_bwd_kernel, dim3(n), dim3(256), bytes, as_stream(stream), x,
                       reinterpret_cast<const __bf16 *>(noise), P, k1, noise_rate, hw, d_out, out, l2_scale,
                       pre_tanh, reinterpret_cast<uint4 *>(d_noise));
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_augment_fwd(const float *x, const int32_t *src_index, const float *params, int32_t n,
                                  int32_t hw, void *out_c8, float *out_f32, void *stream) {
    if (!x || !out_c8 || n < 0 || hw < 2 || hw > 96) return COMBAT_EINVAL;
    if (n == 0) return COMBAT_OK;
    const int bytes = 3 * hw * hw * 4;
    if (set_smem(augment_fwd_kernel, bytes)) return COMBAT_ELAUNCH;
    hipLaunchKernelGGL(augment_fwd_kernel, dim3(n), dim3(256), bytes, as_stream(stream), x, src_index, params, hw,
                       reinterpret_cast<uint4 *>(out_c8), out_f32);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_augment_bwd(const void *d_c8, int32_t c8_channels, const float *params, int32_t n, int32_t hw,
                                  float *d_x, int32_t accumulate, void *stream) {
    if (!d_c8 || !d_x || c8_channels < 3 || n < 0 || hw < 2 || hw > 96) return COMBAT_EINVAL;
    if (n == 0) return COMBAT_OK;
    const int bytes = 3 * hw * hw * 4;
    if (set_smem(augment_bwd_kernel, bytes)) return COMBAT_ELAUNCH;
    hipLaunchKernelGGL(augment_bwd_kernel, dim3(n), dim3(256), bytes, as_stream(stream),
                       reinterpret_cast<const __bf16 *>(d_c8), c8_channels, params, hw, d_x, accumulate);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_dct_u8(const float *x, const float *D, int32_t n, int32_t hw, void *out_c8, void *stream) {
    if (!x || !D || !out_c8 || n < 0 || hw < 2 || hw > 64) return COMBAT_EINVAL;
    if (n == 0) return COMBAT_OK;
    const int bytes = 7 * hw * hw * 4;
    if (set_smem(dct_u8_kernel, bytes)) return COMBAT_ELAUNCH;
    hipLaunchKernelGGL(dct_u8_kernel, dim3(n), dim3(256), bytes, as_stream(stream), x, D, hw,
                       reinterpret_cast<uint4 *>(out_c8));
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}
