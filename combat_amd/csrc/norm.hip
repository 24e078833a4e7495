// Normalisation statistics and backward for BatchNorm2d (train / eval) and InstanceNorm2d.
// All kernels here are HBM/latency-bound reductions over the [rows][2][C] partial sums that the
// convolution epilogues emit (one row per wave of a tile), or short element-wise passes.
//
// Replaces: nn.BatchNorm2d forward statistics + running-stat update and backward
// (classifier_models/preact_resnet.py:20,22,33,36; resnet.py:21,23,29), nn.InstanceNorm2d forward
// statistics and backward (networks/models.py:278-313).
#include "common.hpp"

namespace {

constexpr int kStageRows = 64;  // rows of the second-stage array per group

// Sum partial rows [r0, r1) of group g for 64 channels; thread (cx = tid&63, ry = tid>>6).
__device__ __forceinline__ void reduce_rows(const float *__restrict__ part, int C, int c, long row0, int nrows,
                                            int ry, double &s1, double &s2) {
    s1 = 0.0;
    s2 = 0.0;
    if (c >= C) return;
    for (int r = ry; r < nrows; r += 4) {
        const float *p = part + ((row0 + r) * 2) * (long)C + c;
        s1 += (double)p[0];
        s2 += (double)p[C];
    }
}

__device__ __forceinline__ void block_combine(double &s1, double &s2, int tid) {
    __shared__ double sh[2][256];
    sh[0][tid] = s1;
    sh[1][tid] = s2;
    __syncthreads();
    if (tid < 64) {
        s1 = sh[0][tid] + sh[0][tid + 64] + sh[0][tid + 128] + sh[0][tid + 192];
        s2 = sh[1][tid] + sh[1][tid + 64] + sh[1][tid + 128] + sh[1][tid + 192];
    }
}

// stage 1: [groups][rows][2][C] -> [groups][kStageRows][2][C]
__global__ __launch_bounds__(256) void norm_stage1_kernel(const float *__restrict__ part, int rows_per_group, int C,
                                                          float *__restrict__ out) {
    const int tid = threadIdx.x, cx = tid & 63, ry = tid >> 6;
    const int c = blockIdx.x * 64 + cx, g = blockIdx.y, sl = blockIdx.z;
    const int per = (rows_per_group + kStageRows - 1) / kStageRows;
    const int r0 = sl * per;
    int n = rows_per_group - r0;
    if (n > per) n = per;
    double s1, s2;
    reduce_rows(part, C, c, (long)g * rows_per_group + r0, n > 0 ? n : 0, ry, s1, s2);
    block_combine(s1, s2, tid);
    if (tid < 64 && c < C) {
        float *o = out + (((long)g * kStageRows + sl) * 2) * C + c;
        o[0] = (float)s1;
        o[C] = (float)s2;
    }
}

struct FinalizeArgs {
    const float *part;
    int rows_per_group, C;
    float count, eps;
    const float *gamma, *beta;
    float *mean, *rstd, *scale, *shift, *running_mean, *running_var;
    float momentum;
    int64_t *nbt;
};

__global__ __launch_bounds__(256) void norm_finalize_kernel(const FinalizeArgs a) {
    const int tid = threadIdx.x, cx = tid & 63, ry = tid >> 6;
    const int c = blockIdx.x * 64 + cx, g = blockIdx.y;
    double s1, s2;
    reduce_rows(a.part, a.C, c, (long)g * a.rows_per_group, a.rows_per_group, ry, s1, s2);
    block_combine(s1, s2, tid);
    if (tid < 64 && c < a.C) {
        const double mean = s1 / a.count;
        double var = s2 / a.count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)a.eps);
        const long o = (long)g * a.C + c;
        const double gm = a.gamma ? (double)a.gamma[c] : 1.0;
        const double bt = a.beta ? (double)a.beta[c] : 0.0;
        if (a.mean) a.mean[o] = (float)mean;
        if (a.rstd) a.rstd[o] = (float)rstd;
        if (a.scale) a.scale[o] = (float)(gm * rstd);
        if (a.shift) a.shift[o] = (float)(bt - mean * gm * rstd);
        if (a.running_mean) {
            const double unb = a.count > 1.f ? var * a.count / (a.count - 1.0) : var;
            a.running_mean[c] = (float)((1.0 - a.momentum) * a.running_mean[c] + a.momentum * mean);
            a.running_var[c] = (float)((1.0 - a.momentum) * a.running_var[c] + a.momentum * unb);
        }
    }
    if (a.nbt && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *a.nbt += 1;
}

struct BwdFinalizeArgs {
    const float *part;
    int rows_per_group, C;
    float count;
    const float *gamma, *mean, *rstd;
    float *ca, *cb, *cc, *dgamma, *dbeta;
};

__global__ __launch_bounds__(256) void norm_bwd_finalize_kernel(const BwdFinalizeArgs a) {
    const int tid = threadIdx.x, cx = tid & 63, ry = tid >> 6;
    const int c = blockIdx.x * 64 + cx, g = blockIdx.y;
    double s1, s2;
    reduce_rows(a.part, a.C, c, (long)g * a.rows_per_group, a.rows_per_group, ry, s1, s2);
    block_combine(s1, s2, tid);
    if (tid < 64 && c < a.C) {
        const long o = (long)g * a.C + c;
        const double gm = a.gamma ? (double)a.gamma[c] : 1.0;
        const double mean = a.mean[o], rstd = a.rstd[o];
        const double m1 = s1 / a.count, m2 = s2 / a.count;
        a.ca[o] = (float)(gm * rstd);
        a.cb[o] = (float)(-gm * rstd * rstd * m2);
        a.cc[o] = (float)(gm * rstd * (mean * rstd * m2 - m1));
        if (a.dgamma) a.dgamma[c] = (float)s2;
        if (a.dbeta) a.dbeta[c] = (float)s1;
    }
}

__global__ void bn_eval_fold_kernel(const float *gamma, const float *beta, const float *rm, const float *rv,
                                    float eps, int C, float *scale, float *shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = s;
    shift[c] = beta[c] - rm[c] * s;
}

__global__ void bn_eval_fold_batch_kernel(const combat_bn_desc *__restrict__ descs, float eps) {
    const combat_bn_desc d = descs[blockIdx.y];
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < d.C; c += gridDim.x * blockDim.x) {
        const float s = d.gamma[c] / sqrtf(d.running_var[c] + eps);
        d.scale[c] = s;
        d.shift[c] = d.beta[c] - d.running_mean[c] * s;
    }
}

// one thread per (part, 8-channel chunk); a part is a run of <= 64 rows (a whole small
// InstanceNorm group, or a 32-row slab of a larger one -- same layout as the conv epilogue's)
__global__ __launch_bounds__(256) void group_stats_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ dz,
                                                          int groups, int rows_per_group, int C, int parts_per_image,
                                                          const float *__restrict__ xh_mean,
                                                          const float *__restrict__ xh_rstd,
                                                          float *__restrict__ part) {
    const int nch = C >> 3;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)groups * nch) return;
    const int g = (int)(t / nch), c = (int)(t % nch) * 8;
    float s1[8], s2[8], hs[8], hh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    if (dz) {
        const long img = parts_per_image > 0 ? g / parts_per_image : 0;
        load8f(xh_rstd + img * C + c, hs);
        load8f(xh_mean + img * C + c, hh);
    }
    for (int r = 0; r < rows_per_group; ++r) {
        const long off = ((long)g * rows_per_group + r) * C + c;
        float xv[8];
        unpack8(*reinterpret_cast<const uint4 *>(x + off), xv);
        if (dz) {
            float dv[8];
            unpack8(*reinterpret_cast<const uint4 *>(dz + off), dv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1[e] += dv[e];
                s2[e] = fmaf(dv[e], (xv[e] - hh[e]) * hs[e], s2[e]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1[e] += xv[e];
                s2[e] = fmaf(xv[e], xv[e], s2[e]);
            }
        }
    }
    float *o = part + ((long)g * 2) * C + c;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o[e] = s1[e];
        o[C + e] = s2[e];
    }
}

__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const __bf16 *__restrict__ dz, const __bf16 *__restrict__ x,
                                                             const __bf16 *__restrict__ add, __bf16 *__restrict__ dx,
                                                             long rows, int C, int rows_per_group, int grouped,
                                                             const float *__restrict__ ca, const float *__restrict__ cb,
                                                             const float *__restrict__ cc) {
    const int nch = C >> 3;
    const long total = rows * nch;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long row = t / nch;
        const int c = (int)(t - row * nch) * 8;
        const long go = grouped ? (row / rows_per_group) * C : 0;
        float a[8], b[8], k[8], dv[8], xv[8];
        load8f(ca + go + c, a);
        load8f(cb + go + c, b);
        load8f(cc + go + c, k);
        const long off = row * C + c;
        unpack8(*reinterpret_cast<const uint4 *>(dz + off), dv);
        unpack8(*reinterpret_cast<const uint4 *>(x + off), xv);
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaf(a[e], dv[e], fmaf(b[e], xv[e], k[e]));
        if (add) {
            float ad[8];
            unpack8(*reinterpret_cast<const uint4 *>(add + off), ad);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += ad[e];
        }
        *reinterpret_cast<uint4 *>(dx + off) = pack8(o);
    }
}

// returns the array the finalize kernel should read (possibly the stage-1 output)
const float *maybe_stage1(const float *partials, int groups, int &rows_per_group, int C, float *scratch,
                          int64_t scratch_bytes, hipStream_t st, int &err) {
    err = COMBAT_OK;
    if (rows_per_group <= 2 * kStageRows) return partials;
    const int64_t need = (int64_t)groups * kStageRows * 2 * C * sizeof(float);
    if (!scratch || scratch_bytes < need) {
        err = COMBAT_EINVAL;
        return nullptr;
    }
    hipLaunchKernelGGL(norm_stage1_kernel, dim3((C + 63) / 64, groups, kStageRows), dim3(256), 0, st, partials,
                       rows_per_group, C, scratch);
    if (hipGetLastError() != hipSuccess) {
        err = COMBAT_ELAUNCH;
        return nullptr;
    }
    rows_per_group = kStageRows;
    return scratch;
}

}  // namespace

extern "C" int64_t combat_norm_scratch_bytes(int32_t groups, int32_t C) {
    return (int64_t)groups * kStageRows * 2 * C * sizeof(float);
}

extern "C" int combat_norm_finalize(const float *partials, int32_t groups, int32_t rows_per_group, int32_t C,
                                    float count, float eps, const float *gamma, const float *beta, float *mean,
                                    float *rstd, float *scale, float *shift, float *running_mean,
                                    float *running_var, float momentum, int64_t *num_batches_tracked,
                                    float *scratch, int64_t scratch_bytes, void *stream) {
    if (!partials || groups <= 0 || rows_per_group <= 0 || C <= 0 || count <= 0.f) return COMBAT_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return COMBAT_EINVAL;
    if (running_mean && groups != 1) return COMBAT_EINVAL;
    hipStream_t st = as_stream(stream);
    int err;
    int rpg = rows_per_group;
    const float *src = maybe_stage1(partials, groups, rpg, C, scratch, scratch_bytes, st, err);
    if (err) return err;
    FinalizeArgs a{src, rpg, C, count, eps, gamma, beta, mean, rstd, scale, shift, running_mean, running_var,
                   momentum, num_batches_tracked};
    hipLaunchKernelGGL(norm_finalize_kernel, dim3((C + 63) / 64, groups), dim3(256), 0, st, a);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_norm_bwd_finalize(const float *partials, int32_t groups, int32_t rows_per_group, int32_t C,
                                        float count, const float *gamma, const float *mean, const float *rstd,
                                        float *ca, float *cb, float *cc, float *dgamma, float *dbeta,
                                        float *scratch, int64_t scratch_bytes, void *stream) {
    if (!partials || !mean || !rstd || !ca || !cb || !cc) return COMBAT_EINVAL;
    if (groups <= 0 || rows_per_group <= 0 || C <= 0 || count <= 0.f) return COMBAT_EINVAL;
    if ((dgamma || dbeta) && groups != 1) return COMBAT_EINVAL;
    hipStream_t st = as_stream(stream);
    int err;
    int rpg = rows_per_group;
    const float *src = maybe_stage1(partials, groups, rpg, C, scratch, scratch_bytes, st, err);
    if (err) return err;
    BwdFinalizeArgs a{src, rpg, C, count, gamma, mean, rstd, ca, cb, cc, dgamma, dbeta};
    hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3((C + 63) / 64, groups), dim3(256), 0, st, a);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_bn_eval_fold(const float *gamma, const float *beta, const float *running_mean,
                                   const float *running_var, float eps, int32_t C, float *scale, float *shift,
                                   void *stream) {
    if (!gamma || !beta || !running_mean || !running_var || !scale || !shift || C <= 0) return COMBAT_EINVAL;
    hipLaunchKernelGGL(bn_eval_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), gamma, beta,
                       running_mean, running_var, eps, C, scale, shift);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_bn_eval_fold_batch(const combat_bn_desc *descs, int32_t n, float eps, void *stream) {
    if (!descs || n < 0) return COMBAT_EINVAL;
    if (n == 0) return COMBAT_OK;
    hipLaunchKernelGGL(bn_eval_fold_batch_kernel, dim3(2, (unsigned)n), dim3(256), 0, as_stream(stream), descs, eps);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_group_stats(const void *x, int32_t groups, int32_t rows_per_group, int32_t C,
                                  float *partials, void *stream) {
    if (!x || !partials || groups <= 0 || rows_per_group <= 0 || C <= 0 || (C & 7)) return COMBAT_EINVAL;
    const long t = (long)groups * (C >> 3);
    hipLaunchKernelGGL(group_stats_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const __bf16 *>(x), (const __bf16 *)nullptr, groups, rows_per_group, C, 0,
                       (const float *)nullptr, (const float *)nullptr, partials);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_group_stats_bwd(const void *dz, const void *x, int32_t groups, int32_t rows_per_group,
                                      int32_t C, int32_t parts_per_image, const float *xh_mean,
                                      const float *xh_rstd, float *partials, void *stream) {
    if (!dz || !x || !partials || !xh_mean || !xh_rstd || parts_per_image < 0) return COMBAT_EINVAL;
    if (groups <= 0 || rows_per_group <= 0 || C <= 0 || (C & 7)) return COMBAT_EINVAL;
    const long t = (long)groups * (C >> 3);
    hipLaunchKernelGGL(group_stats_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const __bf16 *>(x), reinterpret_cast<const __bf16 *>(dz), groups,
                       rows_per_group, C, parts_per_image, xh_mean, xh_rstd, partials);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_norm_bwd_apply(const void *dz, const void *x, const void *add, void *dx, int64_t rows,
                                     int32_t C, int32_t rows_per_group, int32_t grouped, const float *ca,
                                     const float *cb, const float *cc, void *stream) {
    if (!dz || !x || !dx || !ca || !cb || !cc || rows <= 0 || C <= 0 || (C & 7)) return COMBAT_EINVAL;
    if (grouped && rows_per_group <= 0) return COMBAT_EINVAL;
    const long total = rows * (C >> 3);
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(norm_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const __bf16 *>(dz), reinterpret_cast<const __bf16 *>(x),
                       reinterpret_cast<const __bf16 *>(add), reinterpret_cast<__bf16 *>(dx), (long)rows, C,
                       rows_per_group, grouped, ca, cb, cc);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}
