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
#include "plan.hpp"

namespace {

// ---- dense hw x hw fp32 products in LDS, 256 threads.  A thread owns 4 consecutive outputs of one row:
// per k one (broadcast) scalar of the left operand and one float4 of the right operand feed 4 FMAs.
// The left operand has row pitch LP (hw + 1 keeps the 8 rows a wave touches on distinct banks), the
// right operand pitch hw; the output pitch OP is chosen by what the result is used as next.  The
// k-order is ascending with one fma per term, as a plain dot-product loop.
__device__ __forceinline__ void mm4(const float *__restrict__ L, int LP, const float *__restrict__ Rm,
                                    float *__restrict__ O, int OP, int hw, int tid) {
    const int q = hw >> 2;
    for (int o = tid; o < hw * q; o += 256) {
        const int i = o / q, j0 = (o - i * q) << 2;
        f32x4_t s = {0.f, 0.f, 0.f, 0.f};
        const float *lp = L + i * LP;
        for (int k = 0; k < hw; ++k) {
            const float a = lp[k];
            const f32x4_t b = *reinterpret_cast<const f32x4_t *>(Rm + k * hw + j0);
            s[0] = fmaf(a, b[0], s[0]);
            s[1] = fmaf(a, b[1], s[1]);
            s[2] = fmaf(a, b[2], s[2]);
            s[3] = fmaf(a, b[3], s[3]);
        }
        float *op = O + i * OP + j0;
        op[0] = s[0]; op[1] = s[1]; op[2] = s[2]; op[3] = s[3];
    }
}

// entry (i, k) of the reflect-padded 3-tap blur matrix: out[i] = k0*in[r(i-1)] + k1*in[i] + k2*in[r(i+1)]
__device__ __forceinline__ float blur_coef(const float (&kk)[3], int hw, int i, int k) {
    const int im = i > 0 ? i - 1 : 1, ip = i + 1 < hw ? i + 1 : hw - 2;
    float c = 0.f;
    if (k == im) c += kk[0];
    if (k == i) c += kk[1];
    if (k == ip) c += kk[2];
    return c;
}

// Kb * X (ROWS) or X * Kb^T (!ROWS), or with the transposed matrix (TR): three-term stencils evaluated in
// ascending index order, i.e. the dense product's fma chain without its zero terms.
template <bool ROWS, bool TR>
__device__ __forceinline__ void blur_pass(const float *__restrict__ X, float *__restrict__ O, const float (&kk)[3],
                                          int hw, int tid) {
    for (int o = tid; o < hw * hw; o += 256) {
        const int i = o / hw, j = o - i * hw;
        const int t = ROWS ? i : j;            // the blurred index
        float s = 0.f;
        for (int k = t > 0 ? t - 1 : 0; k <= (t + 1 < hw ? t + 1 : hw - 1); ++k) {
            const float c = TR ? blur_coef(kk, hw, k, t) : blur_coef(kk, hw, t, k);
            s = fmaf(c, ROWS ? X[k * hw + j] : X[i * hw + k], s);
        }
        O[o] = s;
    }
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

__device__ __forceinline__ void store_hilo(__bf16 *px8, int c, float v) {   // hi/lo halves of channel c of a c8 pixel
    const float h = round_bf16(v);
    px8[c] = (__bf16)h;
    px8[3 + c] = (__bf16)(v - h);
    if (c == 0) *reinterpret_cast<unsigned int *>(px8 + 6) = 0u;
}

// One workgroup per (channel, image): 3n workgroups.  LDS (floats): Pl [hw][hw+1], Pr [hw][hw], A [hw][hw],
// B [hw][hw+1].  P is symmetric (D^T diag(mask) D), so P N P^T = (P N) P needs no transposed operand.
__global__ __launch_bounds__(256) void trigger_fwd_kernel(const float *__restrict__ x, const __bf16 *__restrict__ noise,
                                                          const float *__restrict__ P, const float *__restrict__ k1,
                                                          float rate, int hw, const int *__restrict__ src_index,
                                                          float *__restrict__ out,
                                                          __bf16 *__restrict__ out_c8, float *__restrict__ mse) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int hw2 = hw * hw, lp = hw + 1, tid = threadIdx.x, c = blockIdx.x, img = blockIdx.y;
    const long src = src_index ? src_index[img] : img;   // row of x / noise this output image is made from
    float *Pl = sm, *Pr = Pl + hw * lp, *A = Pr + hw2, *B = A + hw2;
    const float kk[3] = {k1[0], k1[1], k1[2]};
    const float *xi = x + (src * 3 + c) * hw2;
    for (int o = tid; o < hw2; o += 256) {
        const float pv = P[o];
        Pr[o] = pv;
        Pl[(o / hw) * lp + (o % hw)] = pv;
        A[o] = (float)noise[(src * hw2 + o) * 8 + c];
    }
    __syncthreads();
    mm4(Pl, lp, A, B, lp, hw, tid);            // B = P N
    __syncthreads();
    mm4(B, lp, Pr, A, hw, hw, tid);            // A = (P N) P
    __syncthreads();
    for (int o = tid; o < hw2; o += 256) A[o] = fminf(fmaxf(fmaf(A[o], rate, xi[o]), -1.f), 1.f);
    __syncthreads();
    blur_pass<true, false>(A, B, kk, hw, tid);   // Kb * bd
    __syncthreads();
    blur_pass<false, false>(B, A, kk, hw, tid);  // ... * Kb^T
    __syncthreads();
    float se = 0.f;
    for (int o = tid; o < hw2; o += 256) {
        const float v = A[o];
        out[((long)img * 3 + c) * hw2 + o] = v;
        const float d = v - xi[o];
        se = fmaf(d, d, se);
        if (out_c8) store_hilo(out_c8 + ((long)img * hw2 + o) * 8, c, v);
    }
    if (mse) {
        __syncthreads();
        B[tid] = se;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) B[tid] += B[tid + s];
            __syncthreads();
        }
        if (tid == 0) mse[img * 3 + c] = B[0];
    }
}

// d_noise = rate * P * ( clampmask .* (Kb^T * (d_out + 2*l2*(out-x)) * Kb) ) * P      (P symmetric)
// LDS (floats): Pl, Pr, A, B as above + G [hw][hw]
__global__ __launch_bounds__(256) void trigger_bwd_kernel(const float *__restrict__ x, const __bf16 *__restrict__ noise,
                                                          const float *__restrict__ P, const float *__restrict__ k1,
                                                          float rate, int hw, const float *__restrict__ d_out,
                                                          const float *__restrict__ d_out2,
                                                          const float *__restrict__ outp, float l2_scale,
                                                          int pre_tanh, __bf16 *__restrict__ d_noise) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int hw2 = hw * hw, lp = hw + 1, tid = threadIdx.x, c = blockIdx.x, img = blockIdx.y;
    float *Pl = sm, *Pr = Pl + hw * lp, *A = Pr + hw2, *B = A + hw2, *G = B + hw * lp;
    const float kk[3] = {k1[0], k1[1], k1[2]};
    const float *xi = x + ((long)img * 3 + c) * hw2;
    for (int o = tid; o < hw2; o += 256) {
        const float pv = P[o];
        Pr[o] = pv;
        Pl[(o / hw) * lp + (o % hw)] = pv;
        A[o] = (float)noise[((long)img * hw2 + o) * 8 + c];
        const long go = ((long)img * 3 + c) * hw2 + o;
        float g = d_out ? d_out[go] : 0.f;
        if (d_out2) g += d_out2[go];      // a second gradient of the same tensor (another classifier's share)
        if (l2_scale != 0.f) g = fmaf(2.f * l2_scale, outp[go] - xi[o], g);
        G[o] = g;
    }
    __syncthreads();
    mm4(Pl, lp, A, B, lp, hw, tid);
    __syncthreads();
    mm4(B, lp, Pr, A, hw, hw, tid);              // A = P N P  (pre-clamp value is x + rate*A)
    __syncthreads();
    blur_pass<true, true>(G, B, kk, hw, tid);    // B = Kb^T g      (pitch hw: a stencil operand)
    __syncthreads();
    blur_pass<false, true>(B, G, kk, hw, tid);   // G = Kb^T g Kb
    __syncthreads();
    for (int o = tid; o < hw2; o += 256) {
        const float v = fmaf(A[o], rate, xi[o]);
        G[o] = (v >= -1.f && v <= 1.f) ? G[o] * rate : 0.f;
    }
    __syncthreads();
    mm4(Pl, lp, G, B, lp, hw, tid);              // B = P g
    __syncthreads();
    mm4(B, lp, Pr, A, hw, hw, tid);              // A = P g P
    __syncthreads();
    for (int o = tid; o < hw2; o += 256) {
        __bf16 *px = d_noise + ((long)img * hw2 + o) * 8;
        float r = A[o];
        if (pre_tanh) {  // noise = tanh(z): hand back the gradient w.r.t. z
            const float t = (float)noise[((long)img * hw2 + o) * 8 + c];
            r *= 1.f - t * t;
        }
        px[c] = (__bf16)r;
        if (c == 0) {   // channels 3..7 of the c8 pixel carry no gradient
            px[3] = (__bf16)0.f;
            *reinterpret_cast<uint2 *>(px + 4) = make_uint2(0u, 0u);
        }
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

// Images too large for one workgroup's LDS (hw > 96: ImageNet-10's 224 x 224): the same geometry, one thread per
// output pixel, gathering from global memory.
__global__ __launch_bounds__(256) void augment_fwd_big_kernel(const float *__restrict__ x, const int *__restrict__ index,
                                                              const float *__restrict__ params, int hw,
                                                              uint4 *__restrict__ out_c8, float *__restrict__ out_f32) {
    const int hw2 = hw * hw, img = blockIdx.y, o = blockIdx.x * 256 + threadIdx.x;
    if (o >= hw2) return;
    const float *src = x + (long)(index ? index[img] : img) * 3 * hw2;
    const AugGeom g = aug_geom(params, img, hw);
    const int yo = o / hw, xo0 = o - yo * hw;
    const int xo = g.flip ? hw - 1 - xo0 : xo0;
    float v[3];
    if (g.rot) {
        const float dx = xo - g.cx, dy = yo - g.cy;
        const float sx = g.ca * dx - g.sa * dy + g.cx, sy = g.sa * dx + g.ca * dy + g.cy;
        const float fx = floorf(sx), fy = floorf(sy);
        const int x0 = (int)fx, y0 = (int)fy;
        const float ax = sx - fx, ay = sy - fy;
        for (int c = 0; c < 3; ++c) {
            const float *pl = src + c * hw2;
            v[c] = (1.f - ay) * ((1.f - ax) * crop_at(pl, hw, g, x0, y0) + ax * crop_at(pl, hw, g, x0 + 1, y0)) +
                   ay * ((1.f - ax) * crop_at(pl, hw, g, x0, y0 + 1) + ax * crop_at(pl, hw, g, x0 + 1, y0 + 1));
        }
    } else {
        for (int c = 0; c < 3; ++c) v[c] = crop_at(src + c * hw2, hw, g, xo, yo);
    }
    out_c8[(long)img * hw2 + o] = hilo_px(v[0], v[1], v[2]);
    if (out_f32)
        for (int c = 0; c < 3; ++c) out_f32[((long)img * 3 + c) * hw2 + o] = v[c];
}

__device__ __forceinline__ void crop_scatter(float *plane, int hw, const AugGeom &g, int u, int v, float val) {
    if ((unsigned)u >= (unsigned)hw || (unsigned)v >= (unsigned)hw) return;
    const int sx = u + g.ox, sy = v + g.oy;
    if ((unsigned)sx >= (unsigned)hw || (unsigned)sy >= (unsigned)hw) return;
    atomicAdd(plane + sy * hw + sx, val);  // LDS atomic (global in the big-image variant)
}

// The adjoint as a GATHER: no atomics, a fixed summation order (the scatter form -- every output pixel adding its four
// weighted shares to an LDS image with ds_add_f32 -- cost the same 15 us and summed in whatever order the waves ran:
// the generator's gradient differed in the last bits from run to run).  The gradient image is
// staged in LDS; every source pixel (X, Y) -- crop-space pixel (u, v) = (X - ox, Y - oy) -- visits the output pixels
// whose bilinear footprint can contain (u, v): the rotation maps output pixel o to s = R (o - c) + c and (u, v) is one
// of its four taps iff s lies in [u - 1, u + 1) x [v - 1, v + 1), so o lies within (|cos| + |sin|) <= 1.415 of
// q = R^T ((u, v) - c) + c in either axis: the 6 x 6 window [floor(q) - 2, floor(q) + 3] holds them all with 0.58
// pixels to spare.  Each candidate recomputes s, the tap corner and the weights with the forward kernel's own
// expressions (the same products the scatter adds), and the matches are summed in window order.
__global__ __launch_bounds__(256) void augment_bwd_gather_kernel(const __bf16 *__restrict__ d_c8, int cch,
                                                                 const float *__restrict__ params, int hw,
                                                                 float *__restrict__ d_x, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int hw2 = hw * hw, tid = threadIdx.x, img = blockIdx.x;
    for (int o = tid; o < hw2; o += 256) {
        const __bf16 *gp = d_c8 + ((long)img * hw2 + o) * cch;
        sm[o] = (float)gp[0], sm[hw2 + o] = (float)gp[1], sm[2 * hw2 + o] = (float)gp[2];
    }
    __syncthreads();
    const AugGeom g = aug_geom(params, img, hw);
    for (int o = tid; o < hw2; o += 256) {
        const int Y = o / hw, X = o - Y * hw;
        const int u = X - g.ox, v = Y - g.oy;
        float acc[3] = {0.f, 0.f, 0.f};
        if ((unsigned)u < (unsigned)hw && (unsigned)v < (unsigned)hw) {
            if (g.rot) {
                const float qx = g.ca * (u - g.cx) + g.sa * (v - g.cy) + g.cx, qy = -g.sa * (u - g.cx) + g.ca * (v - g.cy) + g.cy;
                const int bx = (int)floorf(qx) - 2, by = (int)floorf(qy) - 2;
                for (int jy = 0; jy < 6; ++jy) {
                    const int yo = by + jy;
                    if ((unsigned)yo >= (unsigned)hw) continue;
                    for (int jx = 0; jx < 6; ++jx) {
                        const int xo = bx + jx;
                        if ((unsigned)xo >= (unsigned)hw) continue;
                        const float dx = xo - g.cx, dy = yo - g.cy;
                        const float sx = g.ca * dx - g.sa * dy + g.cx, sy = g.sa * dx + g.ca * dy + g.cy;
                        const float fx = floorf(sx), fy = floorf(sy);
                        const int x0 = (int)fx, y0 = (int)fy;
                        if ((u != x0 && u != x0 + 1) || (v != y0 && v != y0 + 1)) continue;
                        const float ax = sx - fx, ay = sy - fy;
                        const float w = (v == y0 ? 1.f - ay : ay) * (u == x0 ? 1.f - ax : ax);
                        const int m = yo * hw + (g.flip ? hw - 1 - xo : xo);
#pragma unroll
                        for (int c = 0; c < 3; ++c) acc[c] += w * sm[c * hw2 + m];
                    }
                }
            } else {
                const int m = v * hw + (g.flip ? hw - 1 - u : u);
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] = sm[c * hw2 + m];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float *d = d_x + ((long)img * 3 + c) * hw2 + o;
            *d = accumulate ? *d + acc[c] : acc[c];
        }
    }
}

// hw > 96: scatter with global fp32 atomics into d_x (zeroed by the launcher unless `accumulate`)
__global__ __launch_bounds__(256) void augment_bwd_big_kernel(const __bf16 *__restrict__ d_c8, int cch,
                                                              const float *__restrict__ params, int hw,
                                                              float *__restrict__ d_x) {
    const int hw2 = hw * hw, img = blockIdx.y, o = blockIdx.x * 256 + threadIdx.x;
    if (o >= hw2) return;
    const AugGeom g = aug_geom(params, img, hw);
    const int yo = o / hw, xo0 = o - yo * hw;
    const int xo = g.flip ? hw - 1 - xo0 : xo0;
    const __bf16 *gp = d_c8 + ((long)img * hw2 + o) * cch;
    const float gv[3] = {(float)gp[0], (float)gp[1], (float)gp[2]};
    float *dst = d_x + (long)img * 3 * hw2;
    if (g.rot) {
        const float dx = xo - g.cx, dy = yo - g.cy;
        const float sx = g.ca * dx - g.sa * dy + g.cx, sy = g.sa * dx + g.ca * dy + g.cy;
        const float fx = floorf(sx), fy = floorf(sy);
        const int x0 = (int)fx, y0 = (int)fy;
        const float ax = sx - fx, ay = sy - fy;
        for (int c = 0; c < 3; ++c) {
            float *pl = dst + c * hw2;
            crop_scatter(pl, hw, g, x0, y0, (1.f - ay) * (1.f - ax) * gv[c]);
            crop_scatter(pl, hw, g, x0 + 1, y0, (1.f - ay) * ax * gv[c]);
            crop_scatter(pl, hw, g, x0, y0 + 1, ay * (1.f - ax) * gv[c]);
            crop_scatter(pl, hw, g, x0 + 1, y0 + 1, ay * ax * gv[c]);
        }
    } else {
        for (int c = 0; c < 3; ++c) crop_scatter(dst + c * hw2, hw, g, xo, yo, gv[c]);
    }
}

// ------------------------------------------------------------------ DCT of the uint8-truncated image
// one workgroup per (channel, image); LDS (floats): Dl [hw][hw+1] = D, Dr [hw][hw] = D^T, A, B [hw][hw+1]
__global__ __launch_bounds__(256) void dct_u8_kernel(const float *__restrict__ x, const float *__restrict__ D, int hw,
                                                     __bf16 *__restrict__ out_c8) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int hw2 = hw * hw, lp = hw + 1, tid = threadIdx.x, c = blockIdx.x, img = blockIdx.y;
    float *Dl = sm, *Dr = Dl + hw * lp, *A = Dr + hw2, *B = A + hw2;
    for (int o = tid; o < hw2; o += 256) {
        const int i = o / hw, j = o - i * hw;
        const float dv = D[o];
        Dl[i * lp + j] = dv;
        Dr[j * hw + i] = dv;
        const float q = (x[((long)img * 3 + c) * hw2 + o] + 1.f) / 2.f * 255.f;
        A[o] = (float)(unsigned char)(int)q;  // .byte(): truncate toward zero, wrap mod 256
    }
    __syncthreads();
    mm4(Dl, lp, A, B, lp, hw, tid);     // B = D q
    __syncthreads();
    mm4(B, lp, Dr, A, hw, hw, tid);     // A = D q D^T
    __syncthreads();
    for (int o = tid; o < hw2; o += 256) store_hilo(out_c8 + ((long)img * hw2 + o) * 8, c, A[o]);
}

// hw > 64 (224): one workgroup per (16-row strip, channel, image).  T = D[strip] q in LDS, then Y[strip] = T D^T.
// LDS: Ds [16][hw] (the strip of D), Ts [16][hw].  Metric only (the detector's accuracy), ~4 GFLOP fp32 per 32 images.
__global__ __launch_bounds__(256) void dct_u8_big_kernel(const float *__restrict__ x, const float *__restrict__ D, int hw,
                                                         __bf16 *__restrict__ out_c8) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int hw2 = hw * hw, tid = threadIdx.x, r0 = blockIdx.x * 16, c = blockIdx.y, img = blockIdx.z;
    float *Ds = sm, *Ts = sm + 16 * hw;
    for (int o = tid; o < 16 * hw; o += 256) Ds[o] = D[r0 * hw + o];
    __syncthreads();
    const float *plane = x + ((long)img * 3 + c) * hw2;
    for (int j = tid; j < hw; j += 256) {     // column j of the strip: T[r][j] = sum_k D[r0 + r][k] q[k][j]
        float acc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int k = 0; k < hw; ++k) {
            const float q = (float)(unsigned char)(int)((plane[k * hw + j] + 1.f) / 2.f * 255.f);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = fmaf(Ds[r * hw + k], q, acc[r]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Ts[r * hw + j] = acc[r];
    }
    __syncthreads();
    for (int j = tid; j < hw; j += 256) {     // Y[r][j] = sum_k T[r][k] D[j][k]
        float acc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float *dj = D + (long)j * hw;
        for (int k = 0; k < hw; ++k) {
            const float d = dj[k];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = fmaf(Ts[r * hw + k], d, acc[r]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) store_hilo(out_c8 + ((long)img * hw2 + (r0 + r) * hw + j) * 8, c, acc[r]);
    }
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
                                  float noise_rate, int32_t n, int32_t hw, const int32_t *src_index, float *out,
                                  void *out_c8, float *mse_partial, void *stream) {
    COMBAT_PLAN_HOOK(combat_trigger_fwd, x, noise, P, k1, noise_rate, n, hw, src_index, out, out_c8, mse_partial);
    if (!x || !noise || !P || !k1 || !out || n < 0 || hw < 16 || hw > 64 || (hw & 3)) return COMBAT_EINVAL;
    if (n == 0) return COMBAT_OK;
    const int bytes = (4 * hw * hw + 2 * hw) * 4;
    if (set_smem(trigger_fwd_kernel, bytes)) return COMBAT_ELAUNCH;
    COMBAT_LAUNCH(trigger_fwd_kernel, dim3(3, n), dim3(256), bytes, as_stream(stream), x,
                       reinterpret_cast<const __bf16 *>(noise), P, k1, noise_rate, hw, src_index, out,
                       reinterpret_cast<__bf16 *>(out_c8), mse_partial);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_trigger_bwd(const float *x, const void *noise, const float *P, const float *k1,
                                  float noise_rate, int32_t n, int32_t hw, const float *d_out, const float *d_out2,
                                  const float *out, float l2_scale, int32_t pre_tanh, void *d_noise, void *stream) {
    COMBAT_PLAN_HOOK(combat_trigger_bwd, x, noise, P, k1, noise_rate, n, hw, d_out, d_out2, out, l2_scale, pre_tanh, d_noise);
    if (!x || !noise || !P || !k1 || !d_noise || n < 0 || hw < 16 || hw > 64 || (hw & 3)) return COMBAT_EINVAL;
    if (l2_scale != 0.f && !out) return COMBAT_EINVAL;
    if (n == 0) return COMBAT_OK;
    const int bytes = (5 * hw * hw + 2 * hw) * 4;
    if (set_smem(trigger_bwd_kernel, bytes)) return COMBAT_ELAUNCH;
    COMBAT_LAUNCH(trigger_bwd_kernel, dim3(3, n), dim3(256), bytes, as_stream(stream), x,
                       reinterpret_cast<const __bf16 *>(noise), P, k1, noise_rate, hw, d_out, d_out2, out, l2_scale,
                       pre_tanh, reinterpret_cast<__bf16 *>(d_noise));
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_augment_fwd(const float *x, const int32_t *src_index, const float *params, int32_t n,
                                  int32_t hw, void *out_c8, float *out_f32, void *stream) {
    COMBAT_PLAN_HOOK(combat_augment_fwd, x, src_index, params, n, hw, out_c8, out_f32);
    if (!x || !out_c8 || n < 0 || hw < 2 || hw > 1024) return COMBAT_EINVAL;
    if (n == 0) return COMBAT_OK;
    if (hw > 96) {
        COMBAT_LAUNCH(augment_fwd_big_kernel, dim3((hw * hw + 255) / 256, n), dim3(256), 0, as_stream(stream), x, src_index,
                           params, hw, reinterpret_cast<uint4 *>(out_c8), out_f32);
        CB_LAUNCH_CHECK();
        return COMBAT_OK;
    }
    const int bytes = 3 * hw * hw * 4;
    if (set_smem(augment_fwd_kernel, bytes)) return COMBAT_ELAUNCH;
    COMBAT_LAUNCH(augment_fwd_kernel, dim3(n), dim3(256), bytes, as_stream(stream), x, src_index, params, hw,
                       reinterpret_cast<uint4 *>(out_c8), out_f32);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_augment_bwd(const void *d_c8, int32_t c8_channels, const float *params, int32_t n, int32_t hw,
                                  float *d_x, int32_t accumulate, void *stream) {
    COMBAT_PLAN_HOOK(combat_augment_bwd, d_c8, c8_channels, params, n, hw, d_x, accumulate);
    if (!d_c8 || !d_x || c8_channels < 3 || n < 0 || hw < 2 || hw > 1024) return COMBAT_EINVAL;
    if (n == 0) return COMBAT_OK;
    if (hw > 96) {
        hipStream_t st = as_stream(stream);
        if (!accumulate && hipMemsetAsync(d_x, 0, (size_t)n * 3 * hw * hw * sizeof(float), st) != hipSuccess) return COMBAT_ELAUNCH;
        COMBAT_LAUNCH(augment_bwd_big_kernel, dim3((hw * hw + 255) / 256, n), dim3(256), 0, st,
                           reinterpret_cast<const __bf16 *>(d_c8), c8_channels, params, hw, d_x);
        CB_LAUNCH_CHECK();
        return COMBAT_OK;
    }
    const int bytes = 3 * hw * hw * 4;
    if (set_smem(augment_bwd_gather_kernel, bytes)) return COMBAT_ELAUNCH;
    COMBAT_LAUNCH(augment_bwd_gather_kernel, dim3(n), dim3(256), bytes, as_stream(stream),
                       reinterpret_cast<const __bf16 *>(d_c8), c8_channels, params, hw, d_x, accumulate);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_dct_u8(const float *x, const float *D, int32_t n, int32_t hw, void *out_c8, void *stream) {
    COMBAT_PLAN_HOOK(combat_dct_u8, x, D, n, hw, out_c8);
    if (!x || !D || !out_c8 || n < 0 || hw < 4 || hw > 256 || (hw & 3) || (hw > 64 && (hw & 15))) return COMBAT_EINVAL;
    if (n == 0) return COMBAT_OK;
    if (hw > 64) {
        COMBAT_LAUNCH(dct_u8_big_kernel, dim3(hw / 16, 3, n), dim3(256), 2 * 16 * hw * 4, as_stream(stream), x, D, hw,
                           reinterpret_cast<__bf16 *>(out_c8));
        CB_LAUNCH_CHECK();
        return COMBAT_OK;
    }
    const int bytes = (4 * hw * hw + 2 * hw) * 4;
    if (set_smem(dct_u8_kernel, bytes)) return COMBAT_ELAUNCH;
    COMBAT_LAUNCH(dct_u8_kernel, dim3(3, n), dim3(256), bytes, as_stream(stream), x, D, hw,
                       reinterpret_cast<__bf16 *>(out_c8));
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}
