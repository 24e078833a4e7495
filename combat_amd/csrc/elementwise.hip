// Data-movement and element-wise kernels around the convolutions: weight packing, image layout
// changes, the UNet decoder's norm+skip+upsample glue, multi-tensor Nesterov SGD, pooling.
// All HBM-bound; every global access is a 16-byte (8 x bf16 / 4 x fp32) vector where the layout
// allows it.
#include "common.hpp"
#include "plan.hpp"

namespace {

// ------------------------------------------------------------------ weight packing
__global__ __launch_bounds__(256) void pack_weights_kernel(const float *__restrict__ w, int K, int taps, int creal,
                                                           int C, int dup, __bf16 *__restrict__ wf, int rows_f,
                                                           int kpad_f, __bf16 *__restrict__ wd, int rows_d,
                                                           int kpad_d, int Kc) {
    const long nf = (long)rows_f * kpad_f;
    const long nd = wd ? (long)rows_d * kpad_d : 0;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < nf + nd; t += (long)gridDim.x * blockDim.x) {
        if (t < nf) {
            const int n = (int)(t / kpad_f), k = (int)(t - (long)n * kpad_f);
            const int tap = k / C, c = k - tap * C;
            float v = 0.f;
            if (n < K && tap < taps) {
                int cm = -1;
                if (c < creal) cm = c;
                else if (dup && c < 2 * creal) cm = c - creal;
                if (cm >= 0) v = w[((long)n * taps + tap) * creal + cm];
            }
            wf[t] = (__bf16)v;
        } else {
            const long u = t - nf;
            const int c = (int)(u / kpad_d), k = (int)(u - (long)c * kpad_d);
            const int tap = k / Kc, n = k - tap * Kc;
            float v = 0.f;
            if (c < creal && tap < taps && n < K) v = w[((long)n * taps + tap) * creal + c];
            wd[u] = (__bf16)v;
        }
    }
}

// every convolution of a network: blockIdx.y = descriptor.  The forward layout keeps the master's
// channel order (coalesced both ways); the input-gradient layout swaps the roles of dst channel and
// src channel, so it goes through 32 x 32 LDS tiles: 128-byte reads along c, 64-byte writes along n.
__global__ __launch_bounds__(256) void pack_weights_batch_kernel(const combat_pack_desc *__restrict__ descs) {
    __shared__ float tile[32][33];
    const combat_pack_desc d = descs[blockIdx.y];
    const float *__restrict__ w = d.w;
    __bf16 *__restrict__ wf = reinterpret_cast<__bf16 *>(d.wf);
    __bf16 *__restrict__ wd = reinterpret_cast<__bf16 *>(d.wd);
    const int K = d.K, taps = d.taps, creal = d.c_real, C = d.C, Kc = (d.K + 7) & ~7;
    const float *__restrict__ rs = d.row_scale;
    const long nf = (long)d.rows_pad_f * d.kpad_f;
    if (creal == C && !d.dup_hilo && d.kpad_f == taps * C) {
        // forward layout == the master's layout: a streaming fp32 -> bf16 conversion, 8 elements per thread
        const long nreal = (long)K * d.kpad_f;   // rows >= K are zero padding
        for (long t8 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; t8 < nf; t8 += (long)gridDim.x * blockDim.x * 8) {
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (t8 < nreal) {
                load8f(w + t8, v);
                if (rs) {
                    const float f = rs[t8 / d.kpad_f];   // kpad is a multiple of 8: the eight share a row
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= f;
                }
            }
            *reinterpret_cast<uint4 *>(wf + t8) = pack8(v);
        }
    } else {
        for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < nf; t += (long)gridDim.x * blockDim.x) {
            const int n = (int)(t / d.kpad_f), k = (int)(t - (long)n * d.kpad_f);
            const int tap = k / C, c = k - tap * C;
            float v = 0.f;
            if (n < K && tap < taps) {
                int cm = -1;
                if (c < creal) cm = c;
                else if (d.dup_hilo && c < 2 * creal) cm = c - creal;
                if (cm >= 0) v = w[((long)n * taps + tap) * creal + cm] * (rs ? rs[n] : 1.f);
            }
            wf[t] = (__bf16)v;
        }
    }
    if (!wd) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int tiles_k = d.kpad_d / 32, tiles_c = (d.rows_pad_d + 31) / 32;   // kpad is a multiple of 64
    for (int tl = blockIdx.x; tl < tiles_k * tiles_c; tl += gridDim.x) {
        const int c0 = (tl / tiles_k) * 32, k0 = (tl % tiles_k) * 32;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int kl = ty + 8 * it, k = k0 + kl, tap = k / Kc, n = k - tap * Kc, c = c0 + tx;
            tile[kl][tx] = (c < creal && tap < taps && n < K) ? w[((long)n * taps + tap) * creal + c] * (rs ? rs[n] : 1.f) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int cl = ty + 8 * it, c = c0 + cl;
            if (c < d.rows_pad_d) wd[(long)c * d.kpad_d + k0 + tx] = (__bf16)tile[tx][cl];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ image layout
__device__ __forceinline__ uint4 hilo_pixel(float r, float g, float b) {
    const float hr = round_bf16(r), hg = round_bf16(g), hb = round_bf16(b);
    uint4 u;
    u.x = pack_bf16x2(hr, hg);
    u.y = pack_bf16x2(hb, r - hr);
    u.z = pack_bf16x2(g - hg, b - hb);
    u.w = 0;
    return u;
}

__global__ __launch_bounds__(256) void image_to_c8_kernel(const float *__restrict__ x, int n, int hw2,
                                                          uint4 *__restrict__ out) {
    const long total = (long)n * hw2;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long img = t / hw2, px = t - img * hw2;
        const float *p = x + img * 3 * hw2 + px;
        out[t] = hilo_pixel(p[0], p[hw2], p[2 * hw2]);
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const __bf16 *__restrict__ x, int n, int hw2, int C, int c,
                                                           float *__restrict__ out) {
    const long total = (long)n * c * hw2;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long img = t / ((long)c * hw2);
        const long rem = t - img * c * hw2;
        const int ch = (int)(rem / hw2);
        const long px = rem - (long)ch * hw2;
        out[t] = (float)x[(img * hw2 + px) * C + ch];
    }
}

__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float *__restrict__ x, int n, int c, int hw2, int C,
                                                           __bf16 *__restrict__ out) {
    const long total = (long)n * hw2 * C;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long pix = t / C;
        const int ch = (int)(t - pix * C);
        const long img = pix / hw2, px = pix - img * hw2;
        out[t] = (__bf16)(ch < c ? x[(img * c + ch) * hw2 + px] : 0.f);
    }
}

// ------------------------------------------------------------------ column sums
__global__ __launch_bounds__(256) void colsum_kernel(const __bf16 *__restrict__ x, long rows, int C, int c_out,
                                                     float *__restrict__ out, float *__restrict__ part) {
    // one block per 8-channel chunk and row group; deterministic tree over the block
    __shared__ float sh[256][8];
    const int c = blockIdx.x * 8;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    for (long r = (long)blockIdx.y * 256 + threadIdx.x; r < rows; r += 256L * gridDim.y) {
        float v[8];
        unpack8(*reinterpret_cast<const uint4 *>(x + r * C + c), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) sh[threadIdx.x][e] = s[e];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sh[threadIdx.x][e] += sh[threadIdx.x + o][e];
        }
        __syncthreads();
    }
    // part != nullptr: this row group's sums as row blockIdx.y of part[gridDim.y][C] (colsum_finish_kernel adds the
    // groups in order); else (one row group) straight into out
    if (threadIdx.x < 8) {
        if (part) part[(long)blockIdx.y * C + c + threadIdx.x] = sh[0][threadIdx.x];
        else if (c + (int)threadIdx.x < c_out) out[c + threadIdx.x] = sh[0][threadIdx.x];
    }
}

__global__ __launch_bounds__(256) void colsum_finish_kernel(const float *__restrict__ part, int groups, int C, int c_out,
                                                            float *__restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= c_out) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int y = 0;
    for (; y + 4 <= groups; y += 4) {
        s0 += part[(long)y * C + c];
        s1 += part[(long)(y + 1) * C + c];
        s2 += part[(long)(y + 2) * C + c];
        s3 += part[(long)(y + 3) * C + c];
    }
    for (; y < groups; ++y) s0 += part[(long)y * C + c];
    out[c] = (s0 + s1) + (s2 + s3);
}

// ------------------------------------------------------------------ max pool / ELU+affine
__global__ __launch_bounds__(256) void maxpool2_kernel(const __bf16 *__restrict__ x, int n, int h, int w, int C,
                                                       __bf16 *__restrict__ out) {
    const int nch = C >> 3, ho = h >> 1, wo = w >> 1;
    const long total = (long)n * ho * wo * nch;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(t % nch);
        long p = t / nch;
        const int ox = (int)(p % wo);
        p /= wo;
        const int oy = (int)(p % ho);
        const long img = p / ho;
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                float v[8];
                unpack8(*reinterpret_cast<const uint4 *>(x + ((img * h + 2 * oy + dy) * w + 2 * ox + dx) * C + ch * 8), v);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], v[e]);
            }
        *reinterpret_cast<uint4 *>(out + ((img * ho + oy) * wo + ox) * C + ch * 8) = pack8(m);
    }
}

__global__ __launch_bounds__(256) void elu_affine_kernel(const __bf16 *__restrict__ x, long rows, int C,
                                                         const float *__restrict__ scale,
                                                         const float *__restrict__ shift, __bf16 *__restrict__ out) {
    const int nch = C >> 3;
    const long total = rows * nch;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int c = (int)(t % nch) * 8;
        float v[8], sc[8], sh[8];
        unpack8(*reinterpret_cast<const uint4 *>(x + t * 8), v);
        load8f(scale + c, sc);
        load8f(shift + c, sh);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float el = v[e] > 0.f ? v[e] : expm1f(v[e]);
            v[e] = fmaf(el, sc[e], sh[e]);
        }
        *reinterpret_cast<uint4 *>(out + t * 8) = pack8(v);
    }
}

// out = lrelu(x * scale[g][c] + shift[g][c], slope), g = row / group_rows (group_rows 0: one group):
// the activation a train-mode convolution's prologue would compute, as a tensor (same fma, same rounding)
__global__ __launch_bounds__(256) void affine_act_kernel(const __bf16 *__restrict__ x, long rows, int C,
                                                         const float *__restrict__ scale,
                                                         const float *__restrict__ shift, long group_rows,
                                                         float slope, __bf16 *__restrict__ out) {
    const int nch = C >> 3;
    const long total = rows * nch;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long row = t / nch;
        const int c = (int)(t - row * nch) * 8;
        float v[8];
        unpack8(*reinterpret_cast<const uint4 *>(x + t * 8), v);
        if (scale) {
            const long g = group_rows > 0 ? (row / group_rows) * C : 0;
            float sc[8], sh[8];
            load8f(scale + g + c, sc);
            load8f(shift + g + c, sh);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
        *reinterpret_cast<uint4 *>(out + t * 8) = pack8(v);
    }
}

// ------------------------------------------------------------------ UNet decoder glue
// u = y*sy+ty (+ lrelu(s*ss+ts)); out = lrelu(up2x(u)), align_corners=False:
//   out[2i]   = .25*u[max(i-1,0)] + .75*u[i];   out[2i+1] = .75*u[i] + .25*u[min(i+1,H-1)]
__device__ __forceinline__ void up_taps(int o, int n, int &i0, int &i1, float &w0, float &w1) {
    const int i = o >> 1;
    if (o & 1) {
        i0 = i;
        i1 = i + 1 < n ? i + 1 : n - 1;
        w0 = 0.75f;
        w1 = 0.25f;
    } else {
        i0 = i > 0 ? i - 1 : 0;
        i1 = i;
        w0 = 0.25f;
        w1 = 0.75f;
    }
}

struct UpArgs {
    const __bf16 *y, *s;
    const float *sy, *ty, *ss, *ts;
    int N, H, W, C;
    __bf16 *out;
};

__device__ __forceinline__ void u_value(const UpArgs &a, long img, int iy, int ix, int c, float (&u)[8]) {
    const long off = ((img * a.H + iy) * a.W + ix) * a.C + c;
    float v[8], sc[8], sh[8];
    unpack8(*reinterpret_cast<const uint4 *>(a.y + off), v);
    load8f(a.sy + img * a.C + c, sc);
    load8f(a.ty + img * a.C + c, sh);
#pragma unroll
    for (int e = 0; e < 8; ++e) u[e] = fmaf(v[e], sc[e], sh[e]);
    if (a.s) {
        unpack8(*reinterpret_cast<const uint4 *>(a.s + off), v);
        load8f(a.ss + img * a.C + c, sc);
        load8f(a.ts + img * a.C + c, sh);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float q = fmaf(v[e], sc[e], sh[e]);
            u[e] += q > 0.f ? q : 0.2f * q;
        }
    }
}

__global__ __launch_bounds__(256) void unet_up_fwd_kernel(const UpArgs a) {
    const int nch = a.C >> 3, Ho = 2 * a.H, Wo = 2 * a.W;
    const long total = (long)a.N * Ho * Wo * nch;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int c = (int)(t % nch) * 8;
        long p = t / nch;
        const int ox = (int)(p % Wo);
        p /= Wo;
        const int oy = (int)(p % Ho);
        const long img = p / Ho;
        int y0, y1, x0, x1;
        float wy0, wy1, wx0, wx1;
        up_taps(oy, a.H, y0, y1, wy0, wy1);
        up_taps(ox, a.W, x0, x1, wx0, wx1);
        float u00[8], u01[8], u10[8], u11[8], o[8];
        u_value(a, img, y0, x0, c, u00);
        u_value(a, img, y0, x1, c, u01);
        u_value(a, img, y1, x0, c, u10);
        u_value(a, img, y1, x1, c, u11);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = wy0 * (wx0 * u00[e] + wx1 * u01[e]) + wy1 * (wx0 * u10[e] + wx1 * u11[e]);
            o[e] = v > 0.f ? v : 0.2f * v;
        }
        *reinterpret_cast<uint4 *>(a.out + ((img * Ho + oy) * Wo + ox) * a.C + c) = pack8(o);
    }
}

// du[i] = sum over the (<=3 per axis) output rows that read u[i], of weight * d_out*lrelu'(out)
__global__ __launch_bounds__(256) void unet_up_bwd_kernel(const __bf16 *__restrict__ d_out,
                                                          const __bf16 *__restrict__ out, int N, int H, int W, int C,
                                                          __bf16 *__restrict__ du) {
    const int nch = C >> 3, Ho = 2 * H, Wo = 2 * W;
    const long total = (long)N * H * W * nch;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int c = (int)(t % nch) * 8;
        long p = t / nch;
        const int ix = (int)(p % W);
        p /= W;
        const int iy = (int)(p % H);
        const long img = p / H;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int oy = 2 * iy - 1; oy <= 2 * iy + 2; ++oy) {
            if (oy < 0 || oy >= Ho) continue;
            int y0, y1;
            float wy0, wy1;
            up_taps(oy, H, y0, y1, wy0, wy1);
            const float wy = (y0 == iy ? wy0 : 0.f) + (y1 == iy ? wy1 : 0.f);
            if (wy == 0.f) continue;
            for (int ox = 2 * ix - 1; ox <= 2 * ix + 2; ++ox) {
                if (ox < 0 || ox >= Wo) continue;
                int x0, x1;
                float wx0, wx1;
                up_taps(ox, W, x0, x1, wx0, wx1);
                const float wx = (x0 == ix ? wx0 : 0.f) + (x1 == ix ? wx1 : 0.f);
                if (wx == 0.f) continue;
                const long off = ((img * Ho + oy) * Wo + ox) * C + c;
                float g[8], o[8];
                unpack8(*reinterpret_cast<const uint4 *>(d_out + off), g);
                unpack8(*reinterpret_cast<const uint4 *>(out + off), o);
                const float wgt = wy * wx;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(wgt * (o[e] > 0.f ? 1.f : 0.2f), g[e], acc[e]);
            }
        }
        *reinterpret_cast<uint4 *>(du + ((img * H + iy) * W + ix) * C + c) = pack8(acc);
    }
}

// ------------------------------------------------------------------ SGD
__global__ __launch_bounds__(256) void sgd_kernel(void *const *__restrict__ ptrs, const int64_t *__restrict__ sizes,
                                                  float lr, float mu, float wd, float gscale, int first) {
    const int t = blockIdx.y;
    float *__restrict__ p = reinterpret_cast<float *>(ptrs[3 * t]);
    const float *__restrict__ g = reinterpret_cast<const float *>(ptrs[3 * t + 1]);
    float *__restrict__ b = reinterpret_cast<float *>(ptrs[3 * t + 2]);
    const long n = sizes[t];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float pv = p[i];
        const float gv = fmaf(wd, pv, g[i] * gscale);
        const float bv = first ? gv : fmaf(mu, b[i], gv);
        b[i] = bv;
        p[i] = pv - lr * fmaf(mu, bv, gv);
    }
}

// ------------------------------------------------------------------ fp32 linear on NHWC features
__global__ __launch_bounds__(256) void linear_nhwc_kernel(const __bf16 *__restrict__ x, int h, int w, int C,
                                                          const float *__restrict__ Wt, const float *__restrict__ b,
                                                          int classes, float *__restrict__ logits) {
    // block per sample; feature index of torch's flatten = (c*h + y)*w + xx
    __shared__ float red[256];
    const int img = blockIdx.x;
    const int in = C * h * w;
    for (int j = 0; j < classes; ++j) {
        float s = 0.f;
        for (int i = threadIdx.x; i < in; i += 256) {
            const int c = i / (h * w), r = i - c * h * w;
            s = fmaf((float)x[((long)img * h * w + r) * C + c], Wt[(long)j * in + i], s);
        }
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) logits[(long)img * classes + j] = red[0] + b[j];
        __syncthreads();
    }
}

inline unsigned grid_for(long total, int cap = 4096) {
    long b = (total + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}


// ------------------------------------------------------------------ logged-only terms of the alternated step
// One launch for what the reference computes with a dozen ATen calls per step (train_generator.py:234-247):
//   acc[0] += sum(mse_partial) / (n 3 H W)                                   loss_l2 (from the trigger kernel's partials)
//   acc[1] += mean(dH(pad(x)) - dH(pad(xb)))^2 + mean(dW(pad(x)) - dW(pad(xb)))^2   loss_grad_l2, pad = F.pad(., (1, 1, 2, 1)):
//             one zero column left and right, two zero rows above, one below; both differences are linear, so they are
//             taken of delta = x - xb
//   hits   += #{images : logits[i][1] > logits[i][0]}                           detector says "backdoor" (argmax == 1)
// Workgroup = one (image, channel) plane; fp64 accumulators, one atomic per workgroup and term.
__global__ __launch_bounds__(256) void log_terms_kernel(const float *__restrict__ x, const float *__restrict__ xb,
                                                        const float *__restrict__ mse_partial, int n_partial, int n, int hw,
                                                        const float *__restrict__ logits, int n_logits,
                                                        double *__restrict__ acc, double *__restrict__ hits, double *__restrict__ part) {
    __shared__ float red[2][256];
    const int tid = threadIdx.x, plane = blockIdx.x;
    const float *px = x + (long)plane * hw * hw, *pb = xb + (long)plane * hw * hw;
    auto delta = [&](int r, int c) -> float {       // padded coordinates: rows 0..hw+2, columns 0..hw+1
        const int y = r - 2, xx = c - 1;
        return ((unsigned)y < (unsigned)hw && (unsigned)xx < (unsigned)hw) ? px[y * hw + xx] - pb[y * hw + xx] : 0.f;
    };
    float sh = 0.f, sw = 0.f;
    const int PH = hw + 3, PW = hw + 2;
    for (int i = tid; i < (PH - 1) * PW; i += 256) {          // differences along H: (PH - 1) x PW values
        const int r = i / PW, c = i - r * PW;
        const float d = delta(r + 1, c) - delta(r, c);
        sh = fmaf(d, d, sh);
    }
    for (int i = tid; i < PH * (PW - 1); i += 256) {          // along W: PH x (PW - 1)
        const int r = i / (PW - 1), c = i - r * (PW - 1);
        const float d = delta(r, c + 1) - delta(r, c);
        sw = fmaf(d, d, sw);
    }
    red[0][tid] = sh;
    red[1][tid] = sw;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            red[0][tid] += red[0][tid + s];
            red[1][tid] += red[1][tid + s];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double planes = (double)n * 3.0;
        const double v = (double)red[0][0] / (planes * (PH - 1) * PW) + (double)red[1][0] / (planes * PH * (PW - 1));
        if (part) part[plane] = v;      // deterministic mode: log_terms_finish_kernel adds the planes in order
        else atomicAdd(acc + 1, v);
    }
    if (plane == 0) {
        if (mse_partial) {
            double s = 0.0;
            for (int i = tid; i < n_partial; i += 256) s += (double)mse_partial[i];
            __shared__ double rd[256];
            rd[tid] = s;
            __syncthreads();
            for (int k = 128; k > 0; k >>= 1) {
                if (tid < k) rd[tid] += rd[tid + k];
                __syncthreads();
            }
            if (tid == 0) atomicAdd(acc, rd[0] / ((double)n * 3.0 * hw * hw));
        }
        if (logits && hits) {
            int h = 0;
            for (int i = tid; i < n_logits; i += 256) h += logits[2 * i + 1] > logits[2 * i] ? 1 : 0;
            __shared__ int rh[256];
            rh[tid] = h;
            __syncthreads();
            for (int k = 128; k > 0; k >>= 1) {
                if (tid < k) rh[tid] += rh[tid + k];
                __syncthreads();
            }
            if (tid == 0) atomicAdd(hits, (double)rh[0]);
        }
    }
}

__global__ void log_terms_finish_kernel(const double *__restrict__ part, int planes, double *__restrict__ acc) {
    double s = 0.0;
    for (int i = 0; i < planes; ++i) s += part[i];
    acc[1] += s;
}

}  // namespace

extern "C" int combat_pack_weights(const float *w, int32_t K, int32_t taps, int32_t c_real, int32_t C,
                                   int32_t dup_hilo, void *wf, int32_t rows_pad_f, int32_t kpad_f, void *wd,
                                   int32_t rows_pad_d, int32_t kpad_d, void *stream) {
    COMBAT_PLAN_HOOK(combat_pack_weights, w, K, taps, c_real, C, dup_hilo, wf, rows_pad_f, kpad_f, wd, rows_pad_d, kpad_d);
    if (!w || !wf || K <= 0 || taps <= 0 || c_real <= 0 || C < c_real || (C & 7)) return COMBAT_EINVAL;
    if (rows_pad_f < K || kpad_f < taps * C || (kpad_f & 63)) return COMBAT_EINVAL;
    const int Kc = (K + 7) & ~7;
    if (wd && (rows_pad_d < C || kpad_d < taps * Kc || (kpad_d & 63))) return COMBAT_EINVAL;
    const long total = (long)rows_pad_f * kpad_f + (wd ? (long)rows_pad_d * kpad_d : 0);
    COMBAT_LAUNCH(pack_weights_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), w, K, taps, c_real,
                       C, dup_hilo, reinterpret_cast<__bf16 *>(wf), rows_pad_f, kpad_f, reinterpret_cast<__bf16 *>(wd),
                       rows_pad_d, kpad_d, Kc);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_pack_weights_batch(const combat_pack_desc *descs, int32_t n, void *stream) {
    COMBAT_PLAN_HOOK(combat_pack_weights_batch, descs, n);
    if (!descs || n < 0) return COMBAT_EINVAL;
    if (n == 0) return COMBAT_OK;
    COMBAT_LAUNCH(pack_weights_batch_kernel, dim3(512, (unsigned)n), dim3(256), 0, as_stream(stream), descs);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_image_to_c8(const float *x, int32_t n, int32_t hw, void *out_c8, void *stream) {
    COMBAT_PLAN_HOOK(combat_image_to_c8, x, n, hw, out_c8);
    if (!x || !out_c8 || n < 0 || hw <= 0) return COMBAT_EINVAL;
    if (n == 0) return COMBAT_OK;
    COMBAT_LAUNCH(image_to_c8_kernel, dim3(grid_for((long)n * hw * hw)), dim3(256), 0, as_stream(stream), x, n,
                       hw * hw, reinterpret_cast<uint4 *>(out_c8));
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_nhwc_to_nchw_f32(const void *x, int32_t n, int32_t h, int32_t w, int32_t C, int32_t c,
                                       float *out, void *stream) {
    COMBAT_PLAN_HOOK(combat_nhwc_to_nchw_f32, x, n, h, w, C, c, out);
    if (!x || !out || n < 0 || h <= 0 || w <= 0 || c <= 0 || c > C) return COMBAT_EINVAL;
    if (n == 0) return COMBAT_OK;
    COMBAT_LAUNCH(nhwc_to_nchw_kernel, dim3(grid_for((long)n * c * h * w)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const __bf16 *>(x), n, h * w, C, c, out);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_nchw_to_nhwc_bf16(const float *x, int32_t n, int32_t c, int32_t h, int32_t w, int32_t C,
                                        void *out, void *stream) {
    COMBAT_PLAN_HOOK(combat_nchw_to_nhwc_bf16, x, n, c, h, w, C, out);
    if (!x || !out || n < 0 || h <= 0 || w <= 0 || c <= 0 || c > C || (C & 7)) return COMBAT_EINVAL;
    if (n == 0) return COMBAT_OK;
    COMBAT_LAUNCH(nchw_to_nhwc_kernel, dim3(grid_for((long)n * h * w * C)), dim3(256), 0, as_stream(stream), x, n,
                       c, h * w, C, reinterpret_cast<__bf16 *>(out));
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_memset_zero(void *ptr, int64_t bytes, void *stream) {
    COMBAT_PLAN_HOOK(combat_memset_zero, ptr, bytes);
    if (!ptr || bytes < 0) return COMBAT_EINVAL;
    if (bytes == 0) return COMBAT_OK;
    return hipMemsetAsync(ptr, 0, (size_t)bytes, as_stream(stream)) == hipSuccess ? COMBAT_OK : COMBAT_ELAUNCH;
}

// ------------------------------------------------------------------ the step's small copies, one launch
// Up to three byte copies in ONE kernel launch (grid.y = copy): the step table (from PINNED host memory, read through its
// device mapping), the batch, a re-packed bias.  As three asynchronous copies they cost the critical queue 18 us of
// copy kernels plus 27 us of idle time in front of them at every step boundary (tools/timeline.py).
struct Copy3Args {
    void *dst[3];
    const void *src[3];
    long bytes[3];
};
__global__ __launch_bounds__(256) void copy3_kernel(const Copy3Args a) {
    const int k = blockIdx.y;
    const long n = a.bytes[k];
    unsigned char *__restrict__ d = reinterpret_cast<unsigned char *>(a.dst[k]);
    const unsigned char *__restrict__ s = reinterpret_cast<const unsigned char *>(a.src[k]);
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x, step = (long)gridDim.x * blockDim.x;
    if ((((unsigned long)d | (unsigned long)s) & 15) == 0) {
        const long n16 = n >> 4;
        for (long i = t; i < n16; i += step) reinterpret_cast<uint4 *>(d)[i] = reinterpret_cast<const uint4 *>(s)[i];
        for (long i = (n16 << 4) + t; i < n; i += step) d[i] = s[i];
    } else {
        for (long i = t; i < n; i += step) d[i] = s[i];
    }
}

extern "C" int combat_copy3(void *dst0, const void *src0, int64_t bytes0, void *dst1, const void *src1, int64_t bytes1,
                            void *dst2, const void *src2, int64_t bytes2, void *stream) {
    COMBAT_PLAN_HOOK(combat_copy3, dst0, src0, bytes0, dst1, src1, bytes1, dst2, src2, bytes2);
    Copy3Args a;
    void *dst[3] = {dst0, dst1, dst2};
    const void *src[3] = {src0, src1, src2};
    const int64_t bytes[3] = {bytes0, bytes1, bytes2};
    hipStream_t st = as_stream(stream);
    int n = 0;
    long most = 0;
    for (int k = 0; k < 3; ++k) {
        if (bytes[k] < 0 || (bytes[k] > 0 && (!dst[k] || !src[k]))) return COMBAT_EINVAL;
        if (bytes[k] == 0) continue;
        // a host source must be reachable from the device (pinned memory is mapped); anything else: an ordinary copy
        hipPointerAttribute_t at;
        const void *dev_src = src[k];
        if (hipPointerGetAttributes(&at, dst[k]) != hipSuccess || (at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeManaged) ||
            hipPointerGetAttributes(&at, src[k]) != hipSuccess) {      // (a destination the kernel may not write: ordinary copy)
            (void)hipGetLastError();
            if (hipMemcpyAsync(dst[k], src[k], (size_t)bytes[k], hipMemcpyDefault, st) != hipSuccess) return COMBAT_ELAUNCH;
            continue;
        }
        if (at.type == hipMemoryTypeHost && at.devicePointer) {
            dev_src = at.devicePointer;
        } else if (at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeManaged) {
            // pageable host memory ("unregistered": the query succeeds for it), anything unknown: never hand it to a kernel
            if (hipMemcpyAsync(dst[k], src[k], (size_t)bytes[k], hipMemcpyDefault, st) != hipSuccess) return COMBAT_ELAUNCH;
            continue;
        }
        a.dst[n] = dst[k];
        a.src[n] = dev_src;
        a.bytes[n] = bytes[k];
        if (bytes[k] > most) most = bytes[k];
        ++n;
    }
    if (!n) return COMBAT_OK;
    long bx = (most / 16 + 255) / 256;
    if (bx < 1) bx = 1;
    if (bx > 512) bx = 512;
    COMBAT_LAUNCH(copy3_kernel, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, st, a);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_log_terms(const float *x, const float *xb, const float *mse_partial, int32_t n, int32_t hw,
                                const float *detector_logits, double *acc2, double *hits, void *stream) {
    COMBAT_PLAN_HOOK(combat_log_terms, x, xb, mse_partial, n, hw, detector_logits, acc2, hits);
    if (!x || !xb || !acc2 || n <= 0 || hw < 2) return COMBAT_EINVAL;
    double *part = combat_deterministic() ? reinterpret_cast<double *>(combat_stream_scratch(stream, (size_t)3 * n * sizeof(double))) : nullptr;
    COMBAT_LAUNCH(log_terms_kernel, dim3(3 * n), dim3(256), 0, as_stream(stream), x, xb, mse_partial, 3 * n, n, hw,
                       detector_logits, n, acc2, hits, part);
    CB_LAUNCH_CHECK();
    if (part) {
        COMBAT_LAUNCH(log_terms_finish_kernel, dim3(1), dim3(1), 0, as_stream(stream), (const double *)part, 3 * n, acc2);
        CB_LAUNCH_CHECK();
    }
    return COMBAT_OK;
}

extern "C" int combat_colsum(const void *x, int64_t rows, int32_t C, int32_t c_out, float *out, void *stream) {
    COMBAT_PLAN_HOOK(combat_colsum, x, rows, C, c_out, out);
    if (!x || !out || rows <= 0 || C <= 0 || (C & 7) || c_out <= 0 || c_out > C) return COMBAT_EINVAL;
    hipStream_t st = as_stream(stream);
    long ysplit = (rows + 2047) / 2048;   // ~8 rows per thread
    if (ysplit > 512) ysplit = 512;
    // the row groups' sums meet in a second launch, in a fixed order (they used to meet in `out` through fp32 atomics:
    // a bias gradient that differed in the last bits from run to run); library-owned scratch of this stream
    float *part = ysplit > 1 ? combat_stream_scratch(stream, (size_t)ysplit * C * sizeof(float)) : nullptr;
    if (ysplit > 1 && !part) ysplit = 1;
    COMBAT_LAUNCH(colsum_kernel, dim3((c_out + 7) / 8, (unsigned)ysplit), dim3(256), 0, st,
                       reinterpret_cast<const __bf16 *>(x), (long)rows, C, c_out, out, part);
    CB_LAUNCH_CHECK();
    if (part) {
        COMBAT_LAUNCH(colsum_finish_kernel, dim3((c_out + 255) / 256), dim3(256), 0, st, (const float *)part, (int)ysplit, C, c_out, out);
        CB_LAUNCH_CHECK();
    }
    return COMBAT_OK;
}

extern "C" int combat_maxpool2(const void *x, int32_t n, int32_t h, int32_t w, int32_t C, void *out, void *stream) {
    COMBAT_PLAN_HOOK(combat_maxpool2, x, n, h, w, C, out);
    if (!x || !out || n <= 0 || h <= 0 || w <= 0 || (h & 1) || (w & 1) || (C & 7)) return COMBAT_EINVAL;
    COMBAT_LAUNCH(maxpool2_kernel, dim3(grid_for((long)n * (h / 2) * (w / 2) * (C / 8))), dim3(256), 0,
                       as_stream(stream), reinterpret_cast<const __bf16 *>(x), n, h, w, C,
                       reinterpret_cast<__bf16 *>(out));
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_elu_affine(const void *x, int64_t rows, int32_t C, const float *scale, const float *shift,
                                 void *out, void *stream) {
    COMBAT_PLAN_HOOK(combat_elu_affine, x, rows, C, scale, shift, out);
    if (!x || !out || !scale || !shift || rows <= 0 || C <= 0 || (C & 7)) return COMBAT_EINVAL;
    COMBAT_LAUNCH(elu_affine_kernel, dim3(grid_for(rows * (C / 8))), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const __bf16 *>(x), (long)rows, C, scale, shift,
                       reinterpret_cast<__bf16 *>(out));
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

namespace {
// out = act > 0 ? g : 0  (gradient through the ReLU that produced `act`), 8 bf16 per thread
__global__ __launch_bounds__(256) void relu_mask_kernel(const uint4 *__restrict__ g, const uint4 *__restrict__ act,
                                                        long n8, uint4 *__restrict__ out) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n8; t += (long)gridDim.x * blockDim.x) {
        float gv[8], av[8];
        unpack8(g[t], gv);
        unpack8(act[t], av);
#pragma unroll
        for (int e = 0; e < 8; ++e) gv[e] = av[e] > 0.f ? gv[e] : 0.f;
        out[t] = pack8(gv);
    }
}
}  // namespace

extern "C" int combat_relu_mask(const void *g, const void *act, int64_t elements, void *out, void *stream) {
    COMBAT_PLAN_HOOK(combat_relu_mask, g, act, elements, out);
    if (!g || !act || !out || elements <= 0 || (elements & 7)) return COMBAT_EINVAL;
    COMBAT_LAUNCH(relu_mask_kernel, dim3(grid_for(elements / 8, 8192)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const uint4 *>(g), reinterpret_cast<const uint4 *>(act), (long)(elements / 8),
                       reinterpret_cast<uint4 *>(out));
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_affine_act(const void *x, int64_t rows, int32_t C, const float *scale, const float *shift,
                                 int64_t group_rows, float slope, void *out, void *stream) {
    COMBAT_PLAN_HOOK(combat_affine_act, x, rows, C, scale, shift, group_rows, slope, out);
    if (!x || !out || rows <= 0 || C <= 0 || (C & 7) || group_rows < 0) return COMBAT_EINVAL;
    if ((scale == nullptr) != (shift == nullptr)) return COMBAT_EINVAL;
    COMBAT_LAUNCH(affine_act_kernel, dim3(grid_for(rows * (C / 8), 8192)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const __bf16 *>(x), (long)rows, C, scale, shift, (long)group_rows, slope,
                       reinterpret_cast<__bf16 *>(out));
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_unet_up_fwd(const void *y, const float *sy, const float *ty, const void *s, const float *ss,
                                  const float *ts, int32_t N, int32_t H, int32_t W, int32_t C, void *out,
                                  void *stream) {
    COMBAT_PLAN_HOOK(combat_unet_up_fwd, y, sy, ty, s, ss, ts, N, H, W, C, out);
    if (!y || !sy || !ty || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return COMBAT_EINVAL;
    if (s && (!ss || !ts)) return COMBAT_EINVAL;
    UpArgs a{reinterpret_cast<const __bf16 *>(y), reinterpret_cast<const __bf16 *>(s), sy, ty, ss, ts, N, H, W, C,
             reinterpret_cast<__bf16 *>(out)};
    COMBAT_LAUNCH(unet_up_fwd_kernel, dim3(grid_for((long)N * 4 * H * W * (C / 8), 8192)), dim3(256), 0,
                       as_stream(stream), a);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_unet_up_bwd(const void *d_out, const void *out, int32_t N, int32_t H, int32_t W, int32_t C,
                                  void *du, void *stream) {
    COMBAT_PLAN_HOOK(combat_unet_up_bwd, d_out, out, N, H, W, C, du);
    if (!d_out || !out || !du || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return COMBAT_EINVAL;
    COMBAT_LAUNCH(unet_up_bwd_kernel, dim3(grid_for((long)N * H * W * (C / 8), 8192)), dim3(256), 0,
                       as_stream(stream), reinterpret_cast<const __bf16 *>(d_out), reinterpret_cast<const __bf16 *>(out),
                       N, H, W, C, reinterpret_cast<__bf16 *>(du));
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_sgd_nesterov(const void *ptrs, const int64_t *sizes, int32_t count, int64_t max_size, float lr,
                                   float momentum, float weight_decay, float grad_scale, int32_t first_step,
                                   void *stream) {
    COMBAT_PLAN_HOOK(combat_sgd_nesterov, ptrs, sizes, count, max_size, lr, momentum, weight_decay, grad_scale, first_step);
    if (!ptrs || !sizes || count <= 0 || max_size <= 0) return COMBAT_EINVAL;
    long bx = (max_size + 255) / 256;
    if (bx > 4096) bx = 4096;
    COMBAT_LAUNCH(sgd_kernel, dim3((unsigned)bx, (unsigned)count), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<void *const *>(ptrs), sizes, lr, momentum, weight_decay, grad_scale, first_step);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_linear_nhwc(const void *x, int32_t n, int32_t h, int32_t w, int32_t C, const float *Wt,
                                  const float *b, int32_t classes, float *logits, void *stream) {
    COMBAT_PLAN_HOOK(combat_linear_nhwc, x, n, h, w, C, Wt, b, classes, logits);
    if (!x || !Wt || !b || !logits || n <= 0 || h <= 0 || w <= 0 || C <= 0 || classes <= 0) return COMBAT_EINVAL;
    COMBAT_LAUNCH(linear_nhwc_kernel, dim3(n), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const __bf16 *>(x), h, w, C, Wt, b, classes, logits);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}
