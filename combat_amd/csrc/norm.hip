// Normalisation statistics and backward for BatchNorm2d (train / eval) and InstanceNorm2d.
// All kernels here are HBM/latency-bound reductions over the [rows][2][C] partial sums that the
// convolution epilogues emit (one row per wave of a tile), or short element-wise passes.
//
// Replaces: nn.BatchNorm2d forward statistics + running-stat update and backward
// (classifier_models/preact_resnet.py:20,22,33,36; resnet.py:21,23,29), nn.InstanceNorm2d forward
// statistics and backward (networks/models.py:278-313).
#include "common.hpp"
#include "plan.hpp"

namespace {

constexpr int kStageRows = 64;  // rows of the second-stage array per group

// Sum partial rows [r0, r1) of group g for 64 channels; thread (cx = tid&63, ry = tid>>6).
__device__ __forceinline__ void reduce_rows(const float *__restrict__ part, int C, int c, long row0, int nrows,
                                            int ry, double &s1, double &s2) {
    s1 = 0.0;
    s2 = 0.0;
    if (c >= C) return;
    // eight rows' loads in flight per thread (a plain loop waited out one L2 round trip per row: 64 of them for the
    // 256 rows a batch-statistics layer leaves)
    for (int r = ry; r < nrows; r += 32) {
        float u[8], v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int rr = r + 4 * q < nrows ? r + 4 * q : r;
            const float *p = part + ((row0 + rr) * 2) * (long)C + c;
            u[q] = p[0];
            v[q] = p[C];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (r + 4 * q < nrows) {
                s1 += (double)u[q];
                s2 += (double)v[q];
            }
        }
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

// ------------------------------------------------------------------ fused statistics -> coefficients -> apply
// One launch instead of finalize + element-wise pass (and, for small InstanceNorm groups, the separate
// statistics pass): a workgroup owns (64 channels, one pixel chunk of one group); it first reduces the
// group's partial rows for its channels -- every chunk of a group repeats that, <= 256 rows of 512 B --
// or, with no partial rows, the group's pixels themselves (the group is then a single chunk), turns the
// sums into coefficients with the formulas of the stand-alone finalize kernels, and applies them to its
// pixels.  Chunk 0 publishes the per-channel results.  Thread = (8-channel octet co, pixel lane pl of 32).
// Wave-level sums of the 32 pixel lanes, one LDS row per wave; then thread ch < 64 owns channel ch of the
// workgroup's 64: it adds the four waves (fixed order) and does the fp64 finalisation ONCE per channel --
// with every lane finalising its own eight channels the div / sqrt chains were half of the launch.
__device__ __forceinline__ void fused_combine(double (&s1)[8], double (&s2)[8], int tid, double (*sh)[8][16]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int m = 8; m < 64; m <<= 1) {
            s1[e] += __shfl_xor(s1[e], m, 64);
            s2[e] += __shfl_xor(s2[e], m, 64);
        }
    }
    const int lane = tid & 63, w = tid >> 6, co = tid & 7;
    if (lane < 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sh[w][co][e] = s1[e];
            sh[w][co][8 + e] = s2[e];
        }
    }
    __syncthreads();
}

// rows pl, pl + 32, ... of a group's partial-sum array ([rows][2][C], `part` at the group's row 0, this thread's eight
// channels) added in fp64, FOUR rows' loads (eight 32-byte loads) in flight: as a plain loop every row waited out its own
// L2 round trip -- eight in a row for the 256 rows a batch-statistics layer leaves.  Ascending row order as before.
// Called BEFORE the tensor prefetch of the fused kernels (round 3 tried it behind the prefetch: the registers of both
// at once cost a wave per SIMD and the step 1 %).
__device__ __forceinline__ void fused_reduce_rows(const float *__restrict__ part, int C, int rpg, int pl, double (&s1)[8], double (&s2)[8]) {
    for (int r = pl; r < rpg; r += 128) {
        float u[4][8], v[4][8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rr = r + 32 * q < rpg ? r + 32 * q : r;
            const float *p = part + ((long)rr * 2) * C;
            load8f(p, u[q]);
            load8f(p + C, v[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (r + 32 * q < rpg) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s1[e] += (double)u[q][e];
                    s2[e] += (double)v[q][e];
                }
            }
        }
    }
}

__device__ __forceinline__ void fused_channel_sums(double (*sh)[8][16], int ch, double &t1, double &t2) {
    const int co = ch >> 3, e = ch & 7;
    t1 = sh[0][co][e] + sh[1][co][e] + sh[2][co][e] + sh[3][co][e];
    t2 = sh[0][co][8 + e] + sh[1][co][8 + e] + sh[2][co][8 + e] + sh[3][co][8 + e];
}

struct FusedFwdArgs {
    const __bf16 *x;
    __bf16 *act;
    const float *part;
    int rpg, C;
    long pxg;        // pixels per group
    int chunk_px;
    float count, eps, slope;
    const float *gamma, *beta;
    float *mean, *rstd, *scale, *shift, *running_mean, *running_var;
    float momentum;
    int64_t *nbt;
    const __bf16 *add;                     // optional residual: act = lrelu(x*scale + shift + (add*add_scale + add_shift))
    const float *add_scale, *add_shift;    // per channel (NULL: 1 / 0): the shortcut's own, already finalised BatchNorm
};

__global__ __launch_bounds__(256) void norm_act_fused_kernel(const FusedFwdArgs a) {
    const int tid = threadIdx.x, co = tid & 7, pl = tid >> 3;
    const int c = blockIdx.x * 64 + co * 8, chunk = blockIdx.y, g = blockIdx.z;
    const bool live = c < a.C;
    // The first pass of this workgroup's chunk is loaded BEFORE the statistics are reduced: the two do not depend on
    // each other, and in a launch whose workgroups all run at once (<= 1024 of them) the row reduction's round trip and
    // the tensor's would otherwise follow one another -- 2 of a 10-us launch.
    constexpr int NB = 8;
    const long i0 = (long)chunk * a.chunk_px;
    long i1 = i0 + a.chunk_px;
    if (i1 > a.pxg) i1 = a.pxg;
    u32x4_t lx[NB], la[NB];
    auto load_pass = [&](long ib) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const long i = ib + 32 * q;
            const long off = ((long)g * a.pxg + (i < i1 ? i : i1 - 1)) * a.C + c;
            lx[q] = *reinterpret_cast<const u32x4_t *>(a.x + off);
            if (a.add) la[q] = *reinterpret_cast<const u32x4_t *>(a.add + off);
        }
    };
    double s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.0;
    if (live && a.part) fused_reduce_rows(a.part + ((long)g * a.rpg * 2) * a.C + c, a.C, a.rpg, pl, s1, s2);
    __builtin_amdgcn_sched_barrier(0);      // (the row loads' registers are dead before the prefetch's are allocated)
    if (live && i0 + pl < i1) load_pass(i0 + pl);
    if (live) {
        if (!a.part) {
            float f1[8], f2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f1[e] = f2[e] = 0.f;
            for (long i = pl; i < a.pxg; i += 32) {
                float v[8];
                unpack8(*reinterpret_cast<const uint4 *>(a.x + ((long)g * a.pxg + i) * a.C + c), v);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    f1[e] += v[e];
                    f2[e] = fmaf(v[e], v[e], f2[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1[e] = (double)f1[e];
                s2[e] = (double)f2[e];
            }
        }
    }
    __shared__ double sums[4][8][16];
    __shared__ float coef[2][64];
    fused_combine(s1, s2, tid, sums);
    if (tid < 64) {
        const int cc = blockIdx.x * 64 + tid;
        double t1, t2;
        fused_channel_sums(sums, tid, t1, t2);
        const bool ok = cc < a.C;
        const double mean = t1 / a.count;
        double var = t2 / a.count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)a.eps);
        const double gm = (a.gamma && ok) ? (double)a.gamma[cc] : 1.0;
        const double bt = (a.beta && ok) ? (double)a.beta[cc] : 0.0;
        const float fsc = (float)(gm * rstd), fsh = (float)(bt - mean * gm * rstd);
        coef[0][tid] = fsc;
        coef[1][tid] = fsh;
        if (ok && chunk == 0) {
            const long o = (long)g * a.C + cc;
            if (a.mean) a.mean[o] = (float)mean;
            if (a.rstd) a.rstd[o] = (float)rstd;
            if (a.scale) a.scale[o] = fsc;
            if (a.shift) a.shift[o] = fsh;
            if (a.running_mean) {
                const double unb = a.count > 1.f ? var * a.count / (a.count - 1.0) : var;
                a.running_mean[cc] = (float)((1.0 - a.momentum) * a.running_mean[cc] + a.momentum * mean);
                a.running_var[cc] = (float)((1.0 - a.momentum) * a.running_var[cc] + a.momentum * unb);
            }
        }
    }
    if (a.nbt && blockIdx.x == 0 && chunk == 0 && g == 0 && tid == 0) *a.nbt += 1;
    __syncthreads();
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = coef[0][co * 8 + e];
        sh[e] = coef[1][co * 8 + e];
    }
    if (!live) return;
    float asc[8], ash[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        asc[e] = 1.f;
        ash[e] = 0.f;
    }
    if (a.add && a.add_scale) {
        load8f(a.add_scale + c, asc);
        load8f(a.add_shift + c, ash);
    }
    // Eight pixels per thread and pass, ALL their loads issued before the first is consumed: written as a plain
    // load -> arithmetic -> store loop every iteration waited out its own load (x and act may alias for all the
    // compiler knows, so it keeps the order) -- eight exposed round trips to HBM per thread for a 256-pixel chunk,
    // most of the launch's 10 us.
    for (long ib = i0 + pl; ib < i1; ib += 32 * NB) {
        if (ib != i0 + pl) load_pass(ib);
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const long i = ib + 32 * q;
            if (i >= i1) break;
            const long off = ((long)g * a.pxg + i) * a.C + c;
            float v[8];
            unpack8v(lx[q], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
            if (a.add) {
                float r[8];
                unpack8v(la[q], r);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += fmaf(r[e], asc[e], ash[e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.slope;
            *reinterpret_cast<u32x4_t *>(a.act + off) = pack8v(v);
        }
    }
}

struct FusedBwdArgs {
    const __bf16 *dz, *x, *add;
    __bf16 *dx;
    const float *part;
    int rpg, C;
    long pxg;
    int chunk_px;
    float count;
    const float *gamma, *mean, *rstd;
    float *dgamma, *dbeta;
};

__global__ __launch_bounds__(256) void norm_bwd_fused_kernel(const FusedBwdArgs a) {
    const int tid = threadIdx.x, co = tid & 7, pl = tid >> 3;
    const int c = blockIdx.x * 64 + co * 8, chunk = blockIdx.y, g = blockIdx.z;
    const bool live = c < a.C;
    constexpr int NB = 8;   // (first pass loaded before the statistics, all of a pass's loads before its first use: see norm_act_fused_kernel)
    const long i0 = (long)chunk * a.chunk_px;
    long i1 = i0 + a.chunk_px;
    if (i1 > a.pxg) i1 = a.pxg;
    u32x4_t ld[NB], lx[NB], la[NB];
    auto load_pass = [&](long ib) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const long i = ib + 32 * q;
            const long off = ((long)g * a.pxg + (i < i1 ? i : i1 - 1)) * a.C + c;
            ld[q] = *reinterpret_cast<const u32x4_t *>(a.dz + off);
            lx[q] = *reinterpret_cast<const u32x4_t *>(a.x + off);
            if (a.add) la[q] = *reinterpret_cast<const u32x4_t *>(a.add + off);
        }
    };
    double s1[8], s2[8];
    float mu[8], rs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        s1[e] = s2[e] = 0.0;
        mu[e] = 0.f;
        rs[e] = 1.f;
    }
    if (live && a.part) fused_reduce_rows(a.part + ((long)g * a.rpg * 2) * a.C + c, a.C, a.rpg, pl, s1, s2);
    __builtin_amdgcn_sched_barrier(0);      // (the row loads' registers are dead before the prefetch's are allocated)
    if (live && i0 + pl < i1) load_pass(i0 + pl);
    if (live) {
        load8f(a.mean + (long)g * a.C + c, mu);
        load8f(a.rstd + (long)g * a.C + c, rs);
        if (!a.part) {
            float f1[8], f2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f1[e] = f2[e] = 0.f;
            for (long i = pl; i < a.pxg; i += 32) {
                const long off = ((long)g * a.pxg + i) * a.C + c;
                float dv[8], xv[8];
                unpack8(*reinterpret_cast<const uint4 *>(a.dz + off), dv);
                unpack8(*reinterpret_cast<const uint4 *>(a.x + off), xv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    f1[e] += dv[e];
                    f2[e] = fmaf(dv[e], (xv[e] - mu[e]) * rs[e], f2[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1[e] = (double)f1[e];
                s2[e] = (double)f2[e];
            }
        }
    }
    __shared__ double sums[4][8][16];
    __shared__ float coef[3][64];
    fused_combine(s1, s2, tid, sums);
    if (tid < 64) {
        const int cc = blockIdx.x * 64 + tid;
        double t1, t2;
        fused_channel_sums(sums, tid, t1, t2);
        const bool ok = cc < a.C;
        const double gm = (a.gamma && ok) ? (double)a.gamma[cc] : 1.0;
        const double mean = ok ? (double)a.mean[(long)g * a.C + cc] : 0.0, rstd = ok ? (double)a.rstd[(long)g * a.C + cc] : 1.0;
        const double m1 = t1 / a.count, m2 = t2 / a.count;
        coef[0][tid] = (float)(gm * rstd);
        coef[1][tid] = (float)(-gm * rstd * rstd * m2);
        coef[2][tid] = (float)(gm * rstd * (mean * rstd * m2 - m1));
        if (ok && chunk == 0) {
            if (a.dgamma) a.dgamma[cc] = (float)t2;
            if (a.dbeta) a.dbeta[cc] = (float)t1;
        }
    }
    __syncthreads();
    float ka[8], kb[8], kc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ka[e] = coef[0][co * 8 + e];
        kb[e] = coef[1][co * 8 + e];
        kc[e] = coef[2][co * 8 + e];
    }
    if (!live) return;
    for (long ib = i0 + pl; ib < i1; ib += 32 * NB) {
        if (ib != i0 + pl) load_pass(ib);
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const long i = ib + 32 * q;
            if (i >= i1) break;
            const long off = ((long)g * a.pxg + i) * a.C + c;
            float dv[8], xv[8], o[8];
            unpack8v(ld[q], dv);
            unpack8v(lx[q], xv);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf(ka[e], dv[e], fmaf(kb[e], xv[e], kc[e]));
            if (a.add) {
                float ad[8];
                unpack8v(la[q], ad);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += ad[e];
            }
            *reinterpret_cast<u32x4_t *>(a.dx + off) = pack8v(o);
        }
    }
}

// ------------------------------------------------------------------ InstanceNorm finalisation + UNet decoder glue
// combat_norm_finalize + combat_unet_up_fwd in one launch for the decoder inputs whose InstanceNorm has no
// activation tensor of its own (the normalised map is only ever consumed through the bilinear upsample):
//   out = lrelu_0.2( up2x( IN(y) [+ lrelu_0.2(s * ss + ts)] ) ),  IN statistics from the partial rows / from y.
// Workgroup = (64 channels, a band of output rows, one image); it finalises its image's 64 channels as the
// fused kernels above do (bands repeat that: <= 32 rows), band 0 publishes mean / rstd / scale / shift.
struct UpFusedArgs {
    const __bf16 *y, *s;
    __bf16 *out;
    const float *part;
    int rpg, C, H, W, band;      // band = output rows per workgroup
    float eps;
    const float *ss, *ts;
    float *mean, *rstd, *scale, *shift;
};

__device__ __forceinline__ void up_taps2(int o, int n, int &i0, int &i1, float &w0, float &w1) {
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

__global__ __launch_bounds__(256) void unet_up_fused_kernel(const UpFusedArgs a) {
    const int tid = threadIdx.x, co = tid & 7, pl = tid >> 3;
    const int c = blockIdx.x * 64 + co * 8, band = blockIdx.y, g = blockIdx.z;
    const bool live = c < a.C;
    const long pxg = (long)a.H * a.W;
    double s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.0;
    if (live) {
        if (a.part) {
            for (int r = pl; r < a.rpg; r += 32) {
                const float *p = a.part + (((long)g * a.rpg + r) * 2) * a.C + c;
                float u[8], v[8];
                load8f(p, u);
                load8f(p + a.C, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s1[e] += (double)u[e];
                    s2[e] += (double)v[e];
                }
            }
        } else {
            float f1[8], f2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f1[e] = f2[e] = 0.f;
            for (long i = pl; i < pxg; i += 32) {
                float v[8];
                unpack8(*reinterpret_cast<const uint4 *>(a.y + ((long)g * pxg + i) * a.C + c), v);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    f1[e] += v[e];
                    f2[e] = fmaf(v[e], v[e], f2[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1[e] = (double)f1[e];
                s2[e] = (double)f2[e];
            }
        }
    }
    __shared__ double sums[4][8][16];
    __shared__ float coef[2][64];
    fused_combine(s1, s2, tid, sums);
    if (tid < 64) {
        const int cc = blockIdx.x * 64 + tid;
        double t1, t2;
        fused_channel_sums(sums, tid, t1, t2);
        const double cnt = (double)pxg, mean = t1 / cnt;
        double var = t2 / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)a.eps);
        const float fsc = (float)rstd, fsh = (float)(-mean * rstd);
        coef[0][tid] = fsc;
        coef[1][tid] = fsh;
        if (cc < a.C && band == 0) {
            const long o = (long)g * a.C + cc;
            if (a.mean) a.mean[o] = (float)mean;
            if (a.rstd) a.rstd[o] = (float)rstd;
            if (a.scale) a.scale[o] = fsc;
            if (a.shift) a.shift[o] = fsh;
        }
    }
    __syncthreads();
    if (!live) return;
    float sy[8], ty[8], ss[8], ts[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sy[e] = coef[0][co * 8 + e];
        ty[e] = coef[1][co * 8 + e];
        ss[e] = ts[e] = 0.f;
    }
    if (a.s) {
        load8f(a.ss + (long)g * a.C + c, ss);
        load8f(a.ts + (long)g * a.C + c, ts);
    }
    const int Ho = 2 * a.H, Wo = 2 * a.W;
    const int oy0 = band * a.band, oy1 = oy0 + a.band < Ho ? oy0 + a.band : Ho;
    auto u_at = [&](int iy, int ix, float (&u)[8]) {
        const long off = (((long)g * a.H + iy) * a.W + ix) * a.C + c;
        float v[8];
        unpack8(*reinterpret_cast<const uint4 *>(a.y + off), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) u[e] = fmaf(v[e], sy[e], ty[e]);
        if (a.s) {
            unpack8(*reinterpret_cast<const uint4 *>(a.s + off), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float q = fmaf(v[e], ss[e], ts[e]);
                u[e] += q > 0.f ? q : 0.2f * q;
            }
        }
    };
    for (int p = oy0 * Wo + pl; p < oy1 * Wo; p += 32) {
        const int oy = p / Wo, ox = p - oy * Wo;
        int y0, y1, x0, x1;
        float wy0, wy1, wx0, wx1;
        up_taps2(oy, a.H, y0, y1, wy0, wy1);
        up_taps2(ox, a.W, x0, x1, wx0, wx1);
        float u00[8], u01[8], u10[8], u11[8], o[8];
        u_at(y0, x0, u00);
        u_at(y0, x1, u01);
        u_at(y1, x0, u10);
        u_at(y1, x1, u11);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = wy0 * (wx0 * u00[e] + wx1 * u01[e]) + wy1 * (wx0 * u10[e] + wx1 * u11[e]);
            o[e] = v > 0.f ? v : 0.2f * v;
        }
        *reinterpret_cast<uint4 *>(a.out + (((long)g * Ho + oy) * Wo + ox) * a.C + c) = pack8(o);
    }
}

// Backward of the above: du = adjoint_up(d_out * lrelu'(out)) (stored: the encoder skip paths add it), then the
// InstanceNorm backward of du w.r.t. y, whose two sums run over the whole image -- so a workgroup owns
// (64 channels, one image): pass 1 computes, stores and sums du, pass 2 re-reads its own du and applies
// dx = ca*du + cb*y + cc.  Replaces combat_unet_up_bwd + combat_norm_bwd_fused (sums from the tensors).
// weight with which input row i of n enters output row 2 i - 1 + k (k = 0..3) of the 2x bilinear upsample (up_taps2)
__device__ __forceinline__ float up_adjoint_weight(int i, int n, int k) {
    if (k == 0) return i >= 1 ? 0.25f : 0.f;
    if (k == 1) return i == 0 ? 1.f : 0.75f;
    if (k == 2) return i == n - 1 ? 1.f : 0.75f;
    return i + 1 < n ? 0.25f : 0.f;
}

struct UpBwdFusedArgs {
    const __bf16 *d_out, *out, *y;
    __bf16 *du, *dx;
    int C, H, W;
    const float *mean, *rstd;
};

__global__ __launch_bounds__(256) void unet_up_bwd_fused_kernel(const UpBwdFusedArgs a) {
    const int tid = threadIdx.x, co = tid & 7, pl = tid >> 3;
    const int c = blockIdx.x * 64 + co * 8, g = blockIdx.y;
    const bool live = c < a.C;
    const int H = a.H, W = a.W, Ho = 2 * H, Wo = 2 * W;
    const long pxg = (long)H * W;
    float mu[8], rs[8], f1[8], f2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        mu[e] = f1[e] = f2[e] = 0.f;
        rs[e] = 1.f;
    }
    if (live) {
        load8f(a.mean + (long)g * a.C + c, mu);
        load8f(a.rstd + (long)g * a.C + c, rs);
        for (long i = pl; i < pxg; i += 32) {
            const int iy = (int)(i / W), ix = (int)(i - (long)iy * W);
            float acc[8], wx[4];
            int oxc[4];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            // input row i enters output rows 2 i - 1 .. 2 i + 2 with 0.25 / 0.75 / 0.75 / 0.25 (the clamped edge rows
            // take both weights: up_taps2).  The eight loads of an output row go out before the first is used: as a
            // doubly nested loop with `continue`s every one of the sixteen taps waited out its own two loads -- 32
            // dependent round trips per lane on an 8 x 8 map, most of this launch's 25 us.  Same taps, same order.
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ox = 2 * ix - 1 + k;
                wx[k] = up_adjoint_weight(ix, W, k);
                oxc[k] = ox < 0 ? 0 : (ox >= Wo ? Wo - 1 : ox);
            }
#pragma unroll
            for (int ky = 0; ky < 4; ++ky) {
                const int oy = 2 * iy - 1 + ky;
                const float wy = up_adjoint_weight(iy, H, ky);
                const int oyc = oy < 0 ? 0 : (oy >= Ho ? Ho - 1 : oy);
                uint4 gq[4], oq[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const long off = (((long)g * Ho + oyc) * Wo + oxc[k]) * a.C + c;
                    gq[k] = *reinterpret_cast<const uint4 *>(a.d_out + off);
                    oq[k] = *reinterpret_cast<const uint4 *>(a.out + off);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float wgt = wy * wx[k];
                    if (wgt == 0.f) continue;
                    float gv[8], o[8];
                    unpack8(gq[k], gv);
                    unpack8(oq[k], o);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] = fmaf(wgt * (o[e] > 0.f ? 1.f : 0.2f), gv[e], acc[e]);
                }
            }
            const long off = ((long)g * pxg + i) * a.C + c;
            const uint4 packed = pack8(acc);
            *reinterpret_cast<uint4 *>(a.du + off) = packed;
            float dv[8], xv[8];
            unpack8(packed, dv);               // the sums see the stored (bf16) values, as the two-launch form did
            unpack8(*reinterpret_cast<const uint4 *>(a.y + off), xv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                f1[e] += dv[e];
                f2[e] = fmaf(dv[e], (xv[e] - mu[e]) * rs[e], f2[e]);
            }
        }
    }
    double s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        s1[e] = (double)f1[e];
        s2[e] = (double)f2[e];
    }
    __shared__ double sums[4][8][16];
    __shared__ float coef[3][64];
    fused_combine(s1, s2, tid, sums);
    if (tid < 64) {
        const int cc = blockIdx.x * 64 + tid;
        double t1, t2;
        fused_channel_sums(sums, tid, t1, t2);
        const bool ok = cc < a.C;
        const double mean = ok ? (double)a.mean[(long)g * a.C + cc] : 0.0, rstd = ok ? (double)a.rstd[(long)g * a.C + cc] : 1.0;
        const double m1 = t1 / (double)pxg, m2 = t2 / (double)pxg;
        coef[0][tid] = (float)rstd;
        coef[1][tid] = (float)(-rstd * rstd * m2);
        coef[2][tid] = (float)(rstd * (mean * rstd * m2 - m1));
    }
    __syncthreads();
    if (!live) return;
    float ka[8], kb[8], kc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ka[e] = coef[0][co * 8 + e];
        kb[e] = coef[1][co * 8 + e];
        kc[e] = coef[2][co * 8 + e];
    }
    for (long i = pl; i < pxg; i += 32) {   // (each lane re-reads the du it stored itself)
        const long off = ((long)g * pxg + i) * a.C + c;
        float dv[8], xv[8], o[8];
        unpack8(*reinterpret_cast<const uint4 *>(a.du + off), dv);
        unpack8(*reinterpret_cast<const uint4 *>(a.y + off), xv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaf(ka[e], dv[e], fmaf(kb[e], xv[e], kc[e]));
        *reinterpret_cast<uint4 *>(a.dx + off) = pack8(o);
    }
}

constexpr int kFusedMaxRows = 256;     // partial rows a fused workgroup reduces itself (more: stage 1 first)
constexpr long kFusedMaxDirect = 1024;  // pixels per group the fused kernels reduce without partial rows

// pixel chunk per workgroup: >= 256 pixels (8 per lane, so the repeated row reduction stays small beside
// it) and no more than ~1024 workgroups
int fused_chunk(long pxg, int groups, int C) {
    const long wg_c = (C + 63) / 64;
    long chunk = 256;
    while ((pxg + chunk - 1) / chunk * groups * wg_c > 1024) chunk *= 2;
    return (int)chunk;
}

// returns the array the finalize kernel should read (possibly the stage-1 output)
const float *maybe_stage1(const float *partials, int groups, int &rows_per_group, int C, float *scratch,
                          int64_t scratch_bytes, hipStream_t st, int &err) {
    err = COMBAT_OK;
    if (rows_per_group <= 4 * kStageRows) return partials;     // (<= 256 rows: one launch, eight loads in flight per thread)
    const int64_t need = (int64_t)groups * kStageRows * 2 * C * sizeof(float);
    if (!scratch || scratch_bytes < need) {
        err = COMBAT_EINVAL;
        return nullptr;
    }
    COMBAT_LAUNCH(norm_stage1_kernel, dim3((C + 63) / 64, groups, kStageRows), dim3(256), 0, st, partials,
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
    COMBAT_PLAN_HOOK(combat_norm_finalize, partials, groups, rows_per_group, C, count, eps, gamma, beta, mean, rstd, scale, shift, running_mean, running_var, momentum, num_batches_tracked, scratch, scratch_bytes);
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
    COMBAT_LAUNCH(norm_finalize_kernel, dim3((C + 63) / 64, groups), dim3(256), 0, st, a);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_norm_bwd_finalize(const float *partials, int32_t groups, int32_t rows_per_group, int32_t C,
                                        float count, const float *gamma, const float *mean, const float *rstd,
                                        float *ca, float *cb, float *cc, float *dgamma, float *dbeta,
                                        float *scratch, int64_t scratch_bytes, void *stream) {
    COMBAT_PLAN_HOOK(combat_norm_bwd_finalize, partials, groups, rows_per_group, C, count, gamma, mean, rstd, ca, cb, cc, dgamma, dbeta, scratch, scratch_bytes);
    if (!partials || !mean || !rstd || !ca || !cb || !cc) return COMBAT_EINVAL;
    if (groups <= 0 || rows_per_group <= 0 || C <= 0 || count <= 0.f) return COMBAT_EINVAL;
    if ((dgamma || dbeta) && groups != 1) return COMBAT_EINVAL;
    hipStream_t st = as_stream(stream);
    int err;
    int rpg = rows_per_group;
    const float *src = maybe_stage1(partials, groups, rpg, C, scratch, scratch_bytes, st, err);
    if (err) return err;
    BwdFinalizeArgs a{src, rpg, C, count, gamma, mean, rstd, ca, cb, cc, dgamma, dbeta};
    COMBAT_LAUNCH(norm_bwd_finalize_kernel, dim3((C + 63) / 64, groups), dim3(256), 0, st, a);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

static int norm_act_fused_launch(const void *x, const float *partials, int32_t groups, int32_t rows_per_group,
                                 int64_t px_per_group, int32_t C, float eps, float slope, const float *gamma,
                                 const float *beta, float *mean, float *rstd, float *scale, float *shift,
                                 float *running_mean, float *running_var, float momentum,
                                 int64_t *num_batches_tracked, float *scratch, int64_t scratch_bytes, const void *add,
                                 const float *add_scale, const float *add_shift, void *act, void *stream) {
    if (!x || !act || groups <= 0 || px_per_group <= 0 || C <= 0 || (C & 7)) return COMBAT_EINVAL;
    if (partials ? rows_per_group <= 0 : px_per_group > kFusedMaxDirect) return COMBAT_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return COMBAT_EINVAL;
    if (running_mean && groups != 1) return COMBAT_EINVAL;
    if ((add_scale == nullptr) != (add_shift == nullptr) || (add_scale && (!add || groups != 1))) return COMBAT_EINVAL;
    hipStream_t st = as_stream(stream);
    int rpg = partials ? rows_per_group : 0;
    const float *src = partials;
    if (partials && rpg > kFusedMaxRows) {
        int err;
        src = maybe_stage1(partials, groups, rpg, C, scratch, scratch_bytes, st, err);
        if (err) return err;
    }
    const int chunk = partials ? fused_chunk(px_per_group, groups, C) : (int)px_per_group;
    FusedFwdArgs a{reinterpret_cast<const __bf16 *>(x), reinterpret_cast<__bf16 *>(act), src, rpg, C,
                   (long)px_per_group, chunk, (float)px_per_group, eps, slope, gamma, beta, mean, rstd, scale, shift,
                   running_mean, running_var, momentum, num_batches_tracked, reinterpret_cast<const __bf16 *>(add),
                   add_scale, add_shift};
    COMBAT_LAUNCH(norm_act_fused_kernel,
                       dim3((C + 63) / 64, (unsigned)((px_per_group + chunk - 1) / chunk), groups), dim3(256), 0, st, a);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_norm_act_fused(const void *x, const float *partials, int32_t groups, int32_t rows_per_group,
                                     int64_t px_per_group, int32_t C, float eps, float slope, const float *gamma,
                                     const float *beta, float *mean, float *rstd, float *scale, float *shift,
                                     float *running_mean, float *running_var, float momentum,
                                     int64_t *num_batches_tracked, float *scratch, int64_t scratch_bytes, void *act,
                                     void *stream) {
    COMBAT_PLAN_HOOK(combat_norm_act_fused, x, partials, groups, rows_per_group, px_per_group, C, eps, slope, gamma, beta, mean, rstd, scale, shift, running_mean, running_var, momentum, num_batches_tracked, scratch, scratch_bytes, act);
    return norm_act_fused_launch(x, partials, groups, rows_per_group, px_per_group, C, eps, slope, gamma, beta, mean, rstd,
                                 scale, shift, running_mean, running_var, momentum, num_batches_tracked, scratch,
                                 scratch_bytes, nullptr, nullptr, nullptr, act, stream);
}

extern "C" int combat_norm_add_act_fused(const void *x, const float *partials, int32_t groups, int32_t rows_per_group,
                                         int64_t px_per_group, int32_t C, float eps, float slope, const float *gamma,
                                         const float *beta, float *mean, float *rstd, float *scale, float *shift,
                                         float *running_mean, float *running_var, float momentum,
                                         int64_t *num_batches_tracked, float *scratch, int64_t scratch_bytes,
                                         const void *add, const float *add_scale, const float *add_shift, void *act,
                                         void *stream) {
    COMBAT_PLAN_HOOK(combat_norm_add_act_fused, x, partials, groups, rows_per_group, px_per_group, C, eps, slope, gamma, beta, mean, rstd, scale, shift, running_mean, running_var, momentum, num_batches_tracked, scratch, scratch_bytes, add, add_scale, add_shift, act);
    if (!add) return COMBAT_EINVAL;
    return norm_act_fused_launch(x, partials, groups, rows_per_group, px_per_group, C, eps, slope, gamma, beta, mean, rstd,
                                 scale, shift, running_mean, running_var, momentum, num_batches_tracked, scratch,
                                 scratch_bytes, add, add_scale, add_shift, act, stream);
}

extern "C" int combat_unet_up_fused(const void *y, const float *partials, int32_t rows_per_group, const void *s,
                                    const float *ss, const float *ts, int32_t N, int32_t H, int32_t W, int32_t C,
                                    float eps, float *mean, float *rstd, float *scale, float *shift, void *out,
                                    void *stream) {
    COMBAT_PLAN_HOOK(combat_unet_up_fused, y, partials, rows_per_group, s, ss, ts, N, H, W, C, eps, mean, rstd, scale, shift, out);
    if (!y || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return COMBAT_EINVAL;
    if (s && (!ss || !ts)) return COMBAT_EINVAL;
    if (partials ? (rows_per_group <= 0 || rows_per_group > kFusedMaxRows) : (long)H * W > kFusedMaxDirect) return COMBAT_EINVAL;
    // bands of output rows: enough workgroups to fill the chip, at least 32 output pixels each
    const int wg_c = (C + 63) / 64, Ho = 2 * H, Wo = 2 * W;
    int bands = 1;
    while ((long)N * wg_c * bands < 512 && bands * 2 <= Ho && (Ho / (bands * 2)) * Wo >= 32) bands *= 2;
    const int band = (Ho + bands - 1) / bands;
    UpFusedArgs a{reinterpret_cast<const __bf16 *>(y), reinterpret_cast<const __bf16 *>(s), reinterpret_cast<__bf16 *>(out),
                  partials, partials ? rows_per_group : 0, C, H, W, band, eps, ss, ts, mean, rstd, scale, shift};
    COMBAT_LAUNCH(unet_up_fused_kernel, dim3(wg_c, (Ho + band - 1) / band, N), dim3(256), 0, as_stream(stream), a);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_unet_up_bwd_fused(const void *d_out, const void *out, const void *y, const float *mean,
                                        const float *rstd, int32_t N, int32_t H, int32_t W, int32_t C, void *du,
                                        void *dx, void *stream) {
    COMBAT_PLAN_HOOK(combat_unet_up_bwd_fused, d_out, out, y, mean, rstd, N, H, W, C, du, dx);
    if (!d_out || !out || !y || !mean || !rstd || !du || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7))
        return COMBAT_EINVAL;
    if ((long)H * W > kFusedMaxDirect) return COMBAT_EINVAL;
    UpBwdFusedArgs a{reinterpret_cast<const __bf16 *>(d_out), reinterpret_cast<const __bf16 *>(out),
                     reinterpret_cast<const __bf16 *>(y), reinterpret_cast<__bf16 *>(du), reinterpret_cast<__bf16 *>(dx),
                     C, H, W, mean, rstd};
    COMBAT_LAUNCH(unet_up_bwd_fused_kernel, dim3((C + 63) / 64, N), dim3(256), 0, as_stream(stream), a);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_norm_bwd_fused(const void *dz, const void *x, const void *add, const float *partials,
                                     int32_t groups, int32_t rows_per_group, int64_t px_per_group, int32_t C,
                                     const float *gamma, const float *mean, const float *rstd, float *dgamma,
                                     float *dbeta, float *scratch, int64_t scratch_bytes, void *dx, void *stream) {
    COMBAT_PLAN_HOOK(combat_norm_bwd_fused, dz, x, add, partials, groups, rows_per_group, px_per_group, C, gamma, mean, rstd, dgamma, dbeta, scratch, scratch_bytes, dx);
    if (!dz || !x || !dx || !mean || !rstd || groups <= 0 || px_per_group <= 0 || C <= 0 || (C & 7))
        return COMBAT_EINVAL;
    if (partials ? rows_per_group <= 0 : px_per_group > kFusedMaxDirect) return COMBAT_EINVAL;
    if ((dgamma || dbeta) && groups != 1) return COMBAT_EINVAL;
    hipStream_t st = as_stream(stream);
    int rpg = partials ? rows_per_group : 0;
    const float *src = partials;
    if (partials && rpg > kFusedMaxRows) {
        int err;
        src = maybe_stage1(partials, groups, rpg, C, scratch, scratch_bytes, st, err);
        if (err) return err;
    }
    const int chunk = partials ? fused_chunk(px_per_group, groups, C) : (int)px_per_group;
    FusedBwdArgs a{reinterpret_cast<const __bf16 *>(dz), reinterpret_cast<const __bf16 *>(x),
                   reinterpret_cast<const __bf16 *>(add), reinterpret_cast<__bf16 *>(dx), src, rpg, C,
                   (long)px_per_group, chunk, (float)px_per_group, gamma, mean, rstd, dgamma, dbeta};
    COMBAT_LAUNCH(norm_bwd_fused_kernel,
                       dim3((C + 63) / 64, (unsigned)((px_per_group + chunk - 1) / chunk), groups), dim3(256), 0, st, a);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_bn_eval_fold(const float *gamma, const float *beta, const float *running_mean,
                                   const float *running_var, float eps, int32_t C, float *scale, float *shift,
                                   void *stream) {
    COMBAT_PLAN_HOOK(combat_bn_eval_fold, gamma, beta, running_mean, running_var, eps, C, scale, shift);
    if (!gamma || !beta || !running_mean || !running_var || !scale || !shift || C <= 0) return COMBAT_EINVAL;
    COMBAT_LAUNCH(bn_eval_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), gamma, beta,
                       running_mean, running_var, eps, C, scale, shift);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_bn_eval_fold_batch(const combat_bn_desc *descs, int32_t n, float eps, void *stream) {
    COMBAT_PLAN_HOOK(combat_bn_eval_fold_batch, descs, n, eps);
    if (!descs || n < 0) return COMBAT_EINVAL;
    if (n == 0) return COMBAT_OK;
    COMBAT_LAUNCH(bn_eval_fold_batch_kernel, dim3(2, (unsigned)n), dim3(256), 0, as_stream(stream), descs, eps);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_group_stats(const void *x, int32_t groups, int32_t rows_per_group, int32_t C,
                                  float *partials, void *stream) {
    COMBAT_PLAN_HOOK(combat_group_stats, x, groups, rows_per_group, C, partials);
    if (!x || !partials || groups <= 0 || rows_per_group <= 0 || C <= 0 || (C & 7)) return COMBAT_EINVAL;
    const long t = (long)groups * (C >> 3);
    COMBAT_LAUNCH(group_stats_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const __bf16 *>(x), (const __bf16 *)nullptr, groups, rows_per_group, C, 0,
                       (const float *)nullptr, (const float *)nullptr, partials);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_group_stats_bwd(const void *dz, const void *x, int32_t groups, int32_t rows_per_group,
                                      int32_t C, int32_t parts_per_image, const float *xh_mean,
                                      const float *xh_rstd, float *partials, void *stream) {
    COMBAT_PLAN_HOOK(combat_group_stats_bwd, dz, x, groups, rows_per_group, C, parts_per_image, xh_mean, xh_rstd, partials);
    if (!dz || !x || !partials || !xh_mean || !xh_rstd || parts_per_image < 0) return COMBAT_EINVAL;
    if (groups <= 0 || rows_per_group <= 0 || C <= 0 || (C & 7)) return COMBAT_EINVAL;
    const long t = (long)groups * (C >> 3);
    COMBAT_LAUNCH(group_stats_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const __bf16 *>(x), reinterpret_cast<const __bf16 *>(dz), groups,
                       rows_per_group, C, parts_per_image, xh_mean, xh_rstd, partials);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}

extern "C" int combat_norm_bwd_apply(const void *dz, const void *x, const void *add, void *dx, int64_t rows,
                                     int32_t C, int32_t rows_per_group, int32_t grouped, const float *ca,
                                     const float *cb, const float *cc, void *stream) {
    COMBAT_PLAN_HOOK(combat_norm_bwd_apply, dz, x, add, dx, rows, C, rows_per_group, grouped, ca, cb, cc);
    if (!dz || !x || !dx || !ca || !cb || !cc || rows <= 0 || C <= 0 || (C & 7)) return COMBAT_EINVAL;
    if (grouped && rows_per_group <= 0) return COMBAT_EINVAL;
    const long total = rows * (C >> 3);
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    COMBAT_LAUNCH(norm_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const __bf16 *>(dz), reinterpret_cast<const __bf16 *>(x),
                       reinterpret_cast<const __bf16 *>(add), reinterpret_cast<__bf16 *>(dx), (long)rows, C,
                       rows_per_group, grouped, ca, cb, cc);
    CB_LAUNCH_CHECK();
    return COMBAT_OK;
}
