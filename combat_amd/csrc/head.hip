// Classifier head: avg_pool(4) -> flatten -> Linear -> CrossEntropyLoss, forward and backward.
// Tiny (B x 512 features), latency-bound; one workgroup per sample, fp32 throughout.
//
// Replaces: F.avg_pool2d / nn.AvgPool2d(4), view, nn.Linear (preact_resnet.py:82-83,99-101;
// resnet.py:93-96) and nn.CrossEntropyLoss + argmax accuracy counters (train_generator.py:162,
// 207,231,251,262-267).
#include "common.hpp"
#include "plan.hpp"

namespace {

constexpr int kMaxClasses = 16;

struct HeadFwdArgs {
    const __bf16 *feat;
    int hw, C, classes;
    const float *W, *b;
    const int64_t *targets, *targets2;
    float loss_weight;
    int n;
    float *pooled, *logits, *loss_sum;
    int *correct, *correct2;
    float *loss_part;   // deterministic mode: image i's share of the loss goes to loss_part[i]; head_loss_finish_kernel adds them in order
};

__global__ __launch_bounds__(256) void head_fwd_kernel(const HeadFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // pooled features [in]
    __shared__ float lg[kMaxClasses];
    const int tid = threadIdx.x, img = blockIdx.x;
    const int ph = a.hw / 4, in = a.C * ph * ph;
    // pooled feature index as torch's NCHW flatten: (c*ph + py)*ph + px
    for (int i = tid; i < in; i += 256) {
        const int c = i / (ph * ph), r = i - c * ph * ph;
        const int py = r / ph, px = r - py * ph;
        float s = 0.f;
        for (int dy = 0; dy < 4; ++dy)
            for (int dx = 0; dx < 4; ++dx)
                s += (float)a.feat[(((long)img * a.hw + py * 4 + dy) * a.hw + px * 4 + dx) * a.C + c];
        s *= (1.f / 16.f);
        sm[i] = s;
        if (a.pooled) a.pooled[(long)img * in + i] = s;
    }
    __syncthreads();
    // one class per wave at a time: lane-strided dot product + in-wave butterfly (no workgroup barriers)
    {
        const int lane = tid & 63, wv = tid >> 6;
        for (int j = wv; j < a.classes; j += 4) {
            float s = 0.f;
            for (int i = lane; i < in; i += 64) s = fmaf(sm[i], a.W[(long)j * in + i], s);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) {
                lg[j] = s + a.b[j];
                a.logits[(long)img * a.classes + j] = lg[j];
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        float mx = lg[0];
        int am = 0;
        for (int j = 1; j < a.classes; ++j)
            if (lg[j] > mx) {
                mx = lg[j];
                am = j;
            }
        if (a.targets) {
            float se = 0.f;
            for (int j = 0; j < a.classes; ++j) se += expf(lg[j] - mx);
            const int t = (int)a.targets[img];
            const float li = logf(se) + mx - lg[t];
            if (a.loss_part) a.loss_part[img] = a.loss_weight * li / (float)a.n;
            else if (a.loss_sum) atomicAdd(a.loss_sum, a.loss_weight * li / (float)a.n);
            if (a.correct && am == t) atomicAdd(a.correct, 1);
        }
        if (a.targets2 && a.correct2 && am == (int)a.targets2[img]) atomicAdd(a.correct2, 1);
    }
}

// head_fwd_kernel + head_bwd_feat_kernel in ONE launch (round 4): the forward has everything the feature gradient needs
// (logits in LDS, the label), and on the critical queue the pair was two dependent ~10-us launches with a 6.5-us bubble
// between them -- the end of every classifier forward pass whose loss is differentiated.  Same arithmetic, same order.
__global__ __launch_bounds__(256) void head_fwd_bwd_kernel(const HeadFwdArgs a, float *__restrict__ dlogits, __bf16 *__restrict__ d_feat) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // pooled features [in], then their gradient
    __shared__ float lg[kMaxClasses], dl[kMaxClasses];
    const int tid = threadIdx.x, img = blockIdx.x;
    const int ph = a.hw / 4, in = a.C * ph * ph;
    for (int i = tid; i < in; i += 256) {
        const int c = i / (ph * ph), r = i - c * ph * ph;
        const int py = r / ph, px = r - py * ph;
        float s = 0.f;
        for (int dy = 0; dy < 4; ++dy)
            for (int dx = 0; dx < 4; ++dx)
                s += (float)a.feat[(((long)img * a.hw + py * 4 + dy) * a.hw + px * 4 + dx) * a.C + c];
        s *= (1.f / 16.f);
        sm[i] = s;
        if (a.pooled) a.pooled[(long)img * in + i] = s;
    }
    __syncthreads();
    {
        const int lane = tid & 63, wv = tid >> 6;
        for (int j = wv; j < a.classes; j += 4) {
            float s = 0.f;
            for (int i = lane; i < in; i += 64) s = fmaf(sm[i], a.W[(long)j * in + i], s);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) {
                lg[j] = s + a.b[j];
                a.logits[(long)img * a.classes + j] = lg[j];
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        float mx = lg[0];
        int am = 0;
        for (int j = 1; j < a.classes; ++j)
            if (lg[j] > mx) {
                mx = lg[j];
                am = j;
            }
        float se = 0.f;
        for (int j = 0; j < a.classes; ++j) se += expf(lg[j] - mx);
        const int t = (int)a.targets[img];
        const float li = logf(se) + mx - lg[t];
        if (a.loss_part) a.loss_part[img] = a.loss_weight * li / (float)a.n;
        else if (a.loss_sum) atomicAdd(a.loss_sum, a.loss_weight * li / (float)a.n);
        if (a.correct && am == t) atomicAdd(a.correct, 1);
        if (a.targets2 && a.correct2 && am == (int)a.targets2[img]) atomicAdd(a.correct2, 1);
        for (int j = 0; j < a.classes; ++j) {       // (head_bwd_feat_kernel's expression, from the same logits)
            const float v = a.loss_weight / (float)a.n * (expf(lg[j] - mx) / se - (j == t ? 1.f : 0.f));
            dl[j] = v;
            dlogits[(long)img * a.classes + j] = v;
        }
    }
    __syncthreads();
    if (!d_feat) return;
    for (int i = tid; i < in; i += 256) {
        float s = 0.f;
        for (int j = 0; j < a.classes; ++j) s = fmaf(dl[j], a.W[(long)j * in + i], s);
        sm[i] = s * (1.f / 16.f);
    }
    __syncthreads();
    const int nch = a.C >> 3, hw = a.hw;
    for (int t = tid; t < hw * hw * nch; t += 256) {
        const int ch = (t % nch) * 8, px = t / nch;
        const int y = px / hw, x = px - y * hw;
        const int py = y >> 2, pxx = x >> 2;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = sm[((ch + e) * ph + py) * ph + pxx];
        *reinterpret_cast<uint4 *>(d_feat + (((long)img * hw + y) * hw + x) * a.C + ch) = pack8(o);
    }
}

// dlogits[s][j] = w/n * (softmax - onehot); d_feat = dlogits W / 16 broadcast over each 4x4 window
__global__ __launch_bounds__(256) void head_bwd_feat_kernel(const float *__restrict__ logits,
                                                            const int64_t *__restrict__ targets, float loss_weight,
                                                            int n, int hw, int C, int classes,
                                                            const float *__restrict__ W, float *__restrict__ dlogits,
                                                            __bf16 *__restrict__ d_feat) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];  // gradient of the pooled features [in]
    __shared__ float dl[kMaxClasses];
    const int tid = threadIdx.x, img = blockIdx.x;
    if (tid == 0) {
        const float *lg = logits + (long)img * classes;
        float mx = lg[0];
        for (int j = 1; j < classes; ++j) mx = fmaxf(mx, lg[j]);
        float se = 0.f;
        for (int j = 0; j < classes; ++j) se += expf(lg[j] - mx);
        const int t = (int)targets[img];
        for (int j = 0; j < classes; ++j) {
            const float v = loss_weight / (float)n * (expf(lg[j] - mx) / se - (j == t ? 1.f : 0.f));
            dl[j] = v;
            dlogits[(long)img * classes + j] = v;
        }
    }
    __syncthreads();
    if (!d_feat) return;
    const int ph = hw / 4, in = C * ph * ph;
    // gradient of every pooled feature (coalesced reads of W along i), then broadcast over its 4x4 window
    for (int i = tid; i < in; i += 256) {
        float s = 0.f;
        for (int j = 0; j < classes; ++j) s = fmaf(dl[j], W[(long)j * in + i], s);
        dsm[i] = s * (1.f / 16.f);
    }
    __syncthreads();
    const int nch = C >> 3;
    for (int t = tid; t < hw * hw * nch; t += 256) {
        const int ch = (t % nch) * 8, px = t / nch;
        const int y = px / hw, x = px - y * hw;
        const int py = y >> 2, pxx = x >> 2;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = dsm[((ch + e) * ph + py) * ph + pxx];
        *reinterpret_cast<uint4 *>(d_feat + (((long)img * hw + y) * hw + x) * C + ch) = pack8(o);
    }
}

// dW[j][i] += sum_s dlogits[s][j] * pooled[s][i];  db[j] += sum_s dlogits[s][j]   (blockIdx.z: sample range; the
// gradient buffer is zeroed at the start of every backward pass).  part != nullptr: a range's sums go to row blockIdx.z
// of part[gridDim.z][classes * in + classes] and head_bwd_w_finish_kernel adds the ranges in order -- the ranges used
// to meet in dW through fp32 atomics; else (one range) one add per element.
__global__ __launch_bounds__(256) void head_bwd_w_kernel(const float *__restrict__ dlogits,
                                                         const float *__restrict__ pooled, int n, int in, int classes,
                                                         float *__restrict__ dW, float *__restrict__ db, float *__restrict__ part) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int j = blockIdx.y;
    const int per = (n + gridDim.z - 1) / gridDim.z;
    const int s0 = blockIdx.z * per, s1 = s0 + per < n ? s0 + per : n;
    float *prow = part ? part + (long)blockIdx.z * ((long)classes * in + classes) : nullptr;
    if (i < in) {
        float s = 0.f;
        for (int smp = s0; smp < s1; ++smp) s = fmaf(dlogits[(long)smp * classes + j], pooled[(long)smp * in + i], s);
        if (prow) prow[(long)j * in + i] = s;
        else dW[(long)j * in + i] += s;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int smp = s0; smp < s1; ++smp) s += dlogits[(long)smp * classes + j];
        if (prow) prow[(long)classes * in + j] = s;
        else db[j] += s;
    }
}

__global__ __launch_bounds__(256) void head_bwd_w_finish_kernel(const float *__restrict__ part, int ranges, int in, int classes,
                                                                float *__restrict__ dW, float *__restrict__ db) {
    const long total = (long)classes * in + classes, e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    float s = 0.f;
    for (int z = 0; z < ranges; ++z) s += part[(long)z * total + e];
    if (e < (long)classes * in) dW[e] += s;
    else db[e - (long)classes * in] += s;
}

// deterministic mode: the running loss sum takes the images' shares in image order (they used to arrive as n float atomics)
__global__ void head_loss_finish_kernel(const float *__restrict__ part, int n, float *__restrict__ loss_sum) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += part[i];
    *loss_sum += s;
}

// scratch for the per-image loss shares (deterministic mode with a loss accumulator and labels), or nullptr
float *head_loss_part(const float *loss_sum, const int64_t *targets, int n, void *stream) {
    if (!loss_sum || !targets || !combat_deterministic()) return nullptr;
    return combat_stream_scratch(stream, (size_t)n * sizeof(float));
}

int launch_head_bwd_w(const float *dlogits, const float *pooled, int n, int in, int classes, float *dW, float *db, void *stream) {
    constexpr int kRanges = 8;
    float *part = combat_stream_scratch(stream, (size_t)kRanges * ((size_t)classes * in + classes) * sizeof(float));
    hipStream_t st = as_stream(stream);
    COMBAT_LAUNCH(head_bwd_w_kernel, dim3((in + 255) / 256, classes, part ? kRanges : 1), dim3(256), 0, st, dlogits, pooled, n, in,
                       classes, dW, db, part);
    CB_LAUNCH_CHECK();
    if (part) {
        const long total = (long)classes * in + classes;
        COMBAT_LAUNCH(head_bwd_w_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const float *)part, kRanges, in,
                           classes, dW, db);
        CB_LAUNCH_CHECK();
    }
    return COMBAT_OK;
}

}  // namespace

extern "C" int combat_head_fwd(const void *feat, int32_t n, int32_t hw, int32_t C, const float *W, const float *b,
                               int32_t classes, const int64_t *targets, float loss_weight, float *pooled,
                               float *logits, float *loss_sum, int32_t *correct, const int64_t *targets2,
                               int32_t *correct2, void *stream) {
    COMBAT_PLAN_HOOK(combat_head_fwd, feat, n, hw, C, W, b, classes, targets, loss_weight, pooled, logits, loss_sum, correct, targets2, correct2);
    if (!feat || !W || !b || !logits || n <= 0 || hw < 4 || (hw & 3) || C <= 0 || (C & 7)) return COMBAT_EINVAL;
    if (classes <= 0 || classes > kMaxClasses) return COMBAT_EINVAL;
    const int in = C * (hw / 4) * (hw / 4);
    const int bytes = in * 4;
    if (bytes > 150 * 1024) return COMBAT_EINVAL;
    HeadFwdArgs a{reinterpret_cast<const __bf16 *>(feat), hw, C, classes, W, b, targets, targets2, loss_weight, n,
                  pooled, logits, loss_sum, correct, correct2, head_loss_part(loss_sum, targets, n, stream)};
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(head_fwd_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
            return COMBAT_ELAUNCH;
        attr = true;
    }
    COMBAT_LAUNCH(head_fwd_kernel, dim3(n), dim3(256), bytes, as_stream(stream), a);
    CB_LAUNCH_CHECK();
    if (a.loss_part) {
        COMBAT_LAUNCH(head_loss_finish_kernel, dim3(1), dim3(1), 0, as_stream(stream), (const float *)a.loss_part, n, loss_sum);
        CB_LAUNCH_CHECK();
    }
    return COMBAT_OK;
}

extern "C" int combat_head_bwd(const float *pooled, int32_t n, int32_t hw, int32_t C, const float *W,
                               int32_t classes, const float *logits, const int64_t *targets, float loss_weight,
                               float *dlogits, void *d_feat, float *dW, float *db, void *stream) {
    COMBAT_PLAN_HOOK(combat_head_bwd, pooled, n, hw, C, W, classes, logits, targets, loss_weight, dlogits, d_feat, dW, db);
    if (!W || !logits || !targets || !dlogits || n <= 0 || hw < 4 || (hw & 3) || C <= 0 || (C & 7))
        return COMBAT_EINVAL;
    if (classes <= 0 || classes > kMaxClasses) return COMBAT_EINVAL;
    if (dW && (!db || !pooled)) return COMBAT_EINVAL;
    hipStream_t st = as_stream(stream);
    const int in_ = C * (hw / 4) * (hw / 4);
    if (in_ * 4 > 150 * 1024) return COMBAT_EINVAL;
    static bool attr = false;
    if (!attr) {     // 224 x 224 inputs pool to 7 x 7 x 512 features: 98 KB of LDS
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(head_bwd_feat_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
            return COMBAT_ELAUNCH;
        attr = true;
    }
    COMBAT_LAUNCH(head_bwd_feat_kernel, dim3(n), dim3(256), in_ * 4, st, logits, targets, loss_weight, n, hw, C,
                       classes, W, dlogits, reinterpret_cast<__bf16 *>(d_feat));
    CB_LAUNCH_CHECK();
    if (dW) {
        const int in = C * (hw / 4) * (hw / 4);
        const int rc = launch_head_bwd_w(dlogits, pooled, n, in, classes, dW, db, stream);
        if (rc != COMBAT_OK) return rc;
    }
    return COMBAT_OK;
}

extern "C" int combat_head_fwd_bwd(const void *feat, int32_t n, int32_t hw, int32_t C, const float *W, const float *b,
                                   int32_t classes, const int64_t *targets, float loss_weight, float *pooled,
                                   float *logits, float *loss_sum, int32_t *correct, const int64_t *targets2,
                                   int32_t *correct2, float *dlogits, void *d_feat, void *stream) {
    COMBAT_PLAN_HOOK(combat_head_fwd_bwd, feat, n, hw, C, W, b, classes, targets, loss_weight, pooled, logits, loss_sum, correct, targets2, correct2, dlogits, d_feat);
    if (!feat || !W || !b || !logits || !targets || !dlogits || n <= 0 || hw < 4 || (hw & 3) || C <= 0 || (C & 7)) return COMBAT_EINVAL;
    if (classes <= 0 || classes > kMaxClasses) return COMBAT_EINVAL;
    const int in = C * (hw / 4) * (hw / 4);
    const int bytes = in * 4;
    if (bytes > 150 * 1024) return COMBAT_EINVAL;
    HeadFwdArgs a{reinterpret_cast<const __bf16 *>(feat), hw, C, classes, W, b, targets, targets2, loss_weight, n,
                  pooled, logits, loss_sum, correct, correct2, head_loss_part(loss_sum, targets, n, stream)};
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(head_fwd_bwd_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
            return COMBAT_ELAUNCH;
        attr = true;
    }
    COMBAT_LAUNCH(head_fwd_bwd_kernel, dim3(n), dim3(256), bytes, as_stream(stream), a, dlogits, reinterpret_cast<__bf16 *>(d_feat));
    CB_LAUNCH_CHECK();
    if (a.loss_part) {
        COMBAT_LAUNCH(head_loss_finish_kernel, dim3(1), dim3(1), 0, as_stream(stream), (const float *)a.loss_part, n, loss_sum);
        CB_LAUNCH_CHECK();
    }
    return COMBAT_OK;
}

extern "C" int combat_head_bwd_weights(const float *dlogits, const float *pooled, int32_t n, int32_t hw, int32_t C,
                                       int32_t classes, float *dW, float *db, void *stream) {
    COMBAT_PLAN_HOOK(combat_head_bwd_weights, dlogits, pooled, n, hw, C, classes, dW, db);
    if (!dlogits || !pooled || !dW || !db || n <= 0 || hw < 4 || (hw & 3) || C <= 0 || classes <= 0 || classes > kMaxClasses)
        return COMBAT_EINVAL;
    const int in = C * (hw / 4) * (hw / 4);
    return launch_head_bwd_w(dlogits, pooled, n, in, classes, dW, db, stream);
}
