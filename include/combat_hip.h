/*
 * combat_hip.h -- C ABI of the MI355X (gfx950) kernels behind COMBAT's alternated
 * generator/surrogate training step.
 *
 * The reference (VinAIResearch/COMBAT) has no native code and no FFI: every operator on its
 * hot path is an ATen call made from Python.  Each entry point below therefore
 * replaces the ATen operator(s) issued at the cited reference lines; INTEGRATION.md shows the
 * ctypes stub a maintainer of the reference would add to call them.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless named host_*;
 *   - the caller owns all memory; kernels allocate nothing and keep no state;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all launches are
 *     asynchronous and graph-capturable (no sync, no malloc inside);
 *   - return 0 on success, COMBAT_EINVAL (-1) for an unsupported shape/argument (nothing is
 *     launched), COMBAT_ELAUNCH (-2) if the HIP launch itself failed;
 *   - activations: NHWC, bf16, channel count a multiple of 8 ("c8"); images that enter a
 *     network are NHWC c8 with channels 0..2 = bf16(x) and 3..5 = bf16(x - bf16(x)) (hi/lo
 *     split, so the first convolution sees ~16 mantissa bits), 6..7 = 0;
 *   - packed weights: bf16 [rows_pad][kpad], k = tap * C + c, zero padded
 *     (rows_pad % 128 == 0 or == 16, kpad % 64 == 0);
 *   - fp32 master weights / gradients: [Cout][R][S][Cin] physical order, i.e. a torch OIHW
 *     tensor in channels_last memory format (values and state_dict layout unchanged);
 *   - threading: ONE launching thread and ONE device per process (the one-rank-per-GPU model of
 *     INTEGRATION.md).  Entry points keep no kernel-argument state between calls, but the
 *     "dynamic LDS size set" flags of the kernels are per process (not per device), and plan
 *     recording (combat_plan_record) arms the calling thread only.
 */
#ifndef COMBAT_HIP_H
#define COMBAT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COMBAT_OK 0
#define COMBAT_EINVAL (-1)
#define COMBAT_ELAUNCH (-2)

/* library identity: "combat_hip gfx950 <abi version>" */
const char *combat_version(void);
int combat_abi_version(void);

/* Deterministic mode (process-wide; default: the COMBAT_DETERMINISTIC environment variable, "1" = on).
 * The reference sets neither seeds nor cudnn flags (no manual_seed / cudnn.deterministic anywhere in it):
 * on a GPU it runs with whatever order cuDNN's kernels sum in; on the CPU its modules are reproducible
 * for a fixed thread count (tests/golden/make_golden.py relies on that).
 * Off: partial sums over pixel ranges meet in the gradient buffers through a few fp32 atomics per
 * element (weight gradients of the stride-2 / 1x1 / 8-channel kernels, the 3x3 slab reduction, the
 * head's and the bias gradients) and the augmentation adjoint scatters with LDS atomics: results
 * differ in the last bits from run to run.  On: every one of those reductions has a fixed summation
 * order (slabs + an ordered reduction launch; one sample range; a gathering adjoint) -- two runs on
 * the same inputs give bit-identical parameters; costs a few launches per backward pass (DESIGN
 * section 5).  Weight-gradient launches then REQUIRE the workspace combat_conv_wgrad_workspace_bytes
 * asks for (COMBAT_EINVAL without it).  The logged sums follow: the images' loss shares
 * (combat_head_fwd*) and the planes' gradient-L2 terms (combat_log_terms) are added in index order by a
 * one-thread launch, so every metric is reproducible too.  Not covered: combat_warp_bwd,
 * combat_augment_bwd on images wider than 96 pixels (global scatter). */
void combat_set_deterministic(int on);
int combat_get_deterministic(void);

/* ------------------------------------------------------------------------------------------
 * Convolution as gather-GEMM on MFMA (v_mfma_f32_16x16x32_bf16), fused prologue/epilogue.
 *
 *   dst[m][n] = epilogue( sum_{tap,c} prologue(src[pix(m,tap)][c]) * wpack[n][tap*C + c] )
 *
 * mode 0 (forward)  replaces nn.Conv2d.forward:  preact_resnet.py:21,23,27-29,77;
 *                   resnet.py:20,22,27-30,72; networks/models.py:275-314; frequency model.py:13-39
 *         fused prologue replaces the BatchNorm2d/InstanceNorm2d + ReLU/LeakyReLU that feed
 *         the convolution (preact_resnet.py:33,36; networks/models.py:322-340);
 *         fused epilogue replaces bias add, `out += shortcut` (preact_resnet.py:39), tanh
 *         (models.py:340) and the batch statistics of the following norm layer.
 * mode 1 (dgrad)    replaces the input-gradient half of conv backward (autograd of the same
 *         lines); fused epilogue replaces the ReLU/LeakyReLU mask, the eval-BatchNorm scale,
 *         gradient accumulation over branches and the two reductions of norm backward.
 * ------------------------------------------------------------------------------------------ */
typedef struct combat_conv_args {
    /* geometry: src is N x H x W x C, dst is N x P x Q x K (both NHWC bf16, C,K % 8 == 0) */
    int32_t N, H, W, C;
    int32_t P, Q, K;
    int32_t R, S, stride, pad;   /* R==S in {1,3}; stride in {1,2} */
    int32_t mode;                /* 0 forward, 1 dgrad (src = dY, dst = dX) */
    const void *src;             /* bf16 */
    const void *wpack;           /* bf16 [rows_pad][kpad] (rows = K) */
    int32_t kpad;                /* padded reduction length of wpack, multiple of 64 */
    int32_t rows_pad;            /* padded row count of wpack */
    void *dst;                   /* bf16 (may be NULL when act_dst is given) */
    /* prologue on src: v = src*scale[g][c] + shift[g][c] (if pro_scale), then
       v = v > 0 ? v : v*pro_slope (if pro_act); padding stays exactly 0.
       g = image index * pro_group_stride / C  (stride 0: per-channel, BatchNorm;
       stride C: per-(image,channel), InstanceNorm) */
    const float *pro_scale, *pro_shift;
    int32_t pro_group_stride;
    int32_t pro_act;
    float pro_slope;
    /* epilogue, applied in this order on the fp32 accumulator v of dst[m][n]: */
    const float *bias;           /* v += bias[n]                                  (may be NULL) */
    const void *add_pre;         /* v += add_pre[m][n]   bf16, dst-shaped         (may be NULL) */
    const void *mask_x;          /* bf16 dst-shaped: q = mask_x*mask_scale[g][n] + mask_shift[g][n];
                                    v *= (q > 0 ? 1 : mask_slope)                 (may be NULL) */
    const float *mask_scale, *mask_shift;   /* NULL scale => q = mask_x */
    int32_t mask_activated;      /* mask_x already holds the activation (act_dst of the forward pass):
                                    q = mask_x; mask_scale is then only the mask_mul_scale factor */
    int32_t mask_group_stride;
    float mask_slope;
    int32_t mask_mul_scale;      /* also v *= mask_scale[g][n] (eval-mode BatchNorm backward) */
    const void *add_post;        /* v += add_post[m][n]  bf16, dst-shaped         (may be NULL) */
    int32_t tanh_out;            /* v = tanh(v) */
    /* statistics of the stored (bf16-rounded) values, per granule of `stats_granule` rows:
       stats_kind 1: (sum v, sum v*v);  2: (sum v, sum v*xh) with xh = (mask_x - xh_mean)*xh_rstd.
       layout fp32 [rows][2][K]; one row per wave of a tile (combat_conv_stats_layout).
       | COMBAT_STATS_PER_WORKGROUP: the DMA kernels write one row per workgroup instead (the waves' sums added
       in wave order; the weight-stationary kernel: one row per persistent workgroup) -- for statistics over the
       whole batch, whose consumer then has a quarter (or less) of the rows to reduce; rows no longer align with
       images unless combat_conv_stats_layout says so.  Kernels without that form ignore the bit; the layout
       call reports what the launch will write. */
    int32_t stats_kind;
    float *stats;
    const float *xh_mean, *xh_rstd;         /* group stride = mask_group_stride */
    /* second output (may be NULL): what the NEXT layer's prologue would compute from the stored value
       y = bf16(v), so that the next convolution needs no prologue (eval-mode BatchNorm + ReLU, whose
       scale/shift are known before this layer runs: preact_resnet.py:33,36):
       act_dst[m][n] = bf16(lrelu(y * act_scale[n] + act_shift[n], act_slope)) */
    void *act_dst;
    const float *act_scale, *act_shift;
    float act_slope;
    /* optional scratch (may be NULL): skinny layers (few tiles, long reductions) divide the reduction among
       several workgroups per tile, which write fp32 slabs here and are combined by a second launch; needs
       combat_conv_workspace_bytes(a) bytes, used only by this launch */
    void *workspace;
    int64_t workspace_bytes;
    int32_t tile;                /* 0 = auto; else a COMBAT_TILE_* value */
    /* second reduction source (may be NULL; mode 1 with R = S = 3, stride 2, pad 1, on the gathered-DMA kernel only):
       dst also receives the input gradient of a 1x1 / stride-2 / pad-0 convolution over src2 -- a tensor of src's
       shape -- with operand wpack2 [rows_pad2][kpad2] (rows = K, k = channel of src2): a residual block's shortcut
       (preact_resnet.py:27-36, resnet.py:24-31).  Its gradient lands exactly on the pixels of the 3x3 convolution's
       centre tap, so it rides along as C / 64 more reduction steps: one launch instead of two and no intermediate
       tensor; the epilogue (mask, statistics, residuals) then sees the sum, as it did through add_pre. */
    const void *src2, *wpack2;
    int32_t kpad2, rows_pad2;
    /* second output of the PROLOGUE (may be NULL; forward 3x3 / stride-1 launches with a per-channel prologue on the
       DMA-staged kernel, which applies the prologue in LDS after the operand has landed): the activated input
       pro_act_dst[m][c] = bf16(lrelu(src[m][c] * pro_scale[c] + pro_shift[c], pro_slope)), src-shaped -- the tensor
       the weight gradient of this layer reads, written by the launch that computes it anyway instead of by a
       normalisation launch of its own in front of it (train-mode relu(bn(x)): preact_resnet.py:32,35).  Launches that
       cannot honour it return COMBAT_EINVAL (combat_conv_pick_tile says which kernel a launch takes). */
    void *pro_act_dst;
} combat_conv_args;

#define COMBAT_STATS_PER_WORKGROUP 4
#define COMBAT_TILE_128x128 1
#define COMBAT_TILE_128x64 2
#define COMBAT_TILE_64x64 3
#define COMBAT_TILE_128x16 4
#define COMBAT_TILE_64x128 5
/* 3x3 / stride-1 / pad-1 convolutions with C, K multiples of 64 run with the input patch (plus
 * halo) staged once in LDS and the nine taps as shifted LDS reads; tile = pixels x channels */
#define COMBAT_TILE_H256x64 6
#define COMBAT_TILE_H128x128 7
#define COMBAT_TILE_H128x64 8
#define COMBAT_TILE_H64x64 9
/* the same convolutions when they have NO input prologue, per-channel (not per-image) mask tables
 * and no tanh: both operands go global -> LDS by DMA (buffer_load ... lds), 128 pixels x 64
 * channels per workgroup (conv3x3_dma.hip) */
#define COMBAT_TILE_D128x64 10
#define COMBAT_TILE_D128x32 11   /* skinny layers: twice the workgroups */
/* every other prologue-free convolution with C % 64 == 0 (strided, 1x1, input gradients): gathered pixel
 * rows and weight rows by DMA into a three-stage LDS ring (conv_gather_dma.hip) */
#define COMBAT_TILE_G128x64 12
#define COMBAT_TILE_G128x32 13
#define COMBAT_TILE_D256x64 14   /* conv3x3_dma with 256-pixel tiles (eight waves): large layers */
#define COMBAT_TILE_C8 15        /* 3x3 over C = 8 (c8 images, 3-channel gradients): operands straight into the MFMA registers */
#define COMBAT_TILE_S128x64 17   /* conv3x3 for C = K = 64, weight-stationary + persistent: the filter bank stays in LDS, 128-pixel tiles (statistics rows as D128x64) */
#define COMBAT_TILE_K8 18        /* 3x3 stride 1 with eight OUTPUT channels (generator output layer, stem input gradient): patch through registers with the prologue, weights in registers (conv_k8.hip) */
#define COMBAT_TILE_D256W64 16   /* conv3x3_dma, 256-pixel tiles as four waves of 64 pixels x 64 channels (16-wide maps, >= 16 rows) */

int combat_conv_gemm(const combat_conv_args *a, void *stream);
/* Two convolutions that do not depend on each other, in stream order a then b -- as ONE launch where both take the
 * gathered-DMA kernel with the same channel tile (a residual block's stride-2 first convolution and its 1x1 shortcut over
 * the same activated input: preact_resnet.py:33-36, resnet.py:24-31).  Results equal two combat_conv_gemm calls. */
int combat_conv_gemm_pair(const combat_conv_args *a, const combat_conv_args *b, void *stream);
/* scratch bytes the launch for these args can use (0: none) */
int64_t combat_conv_workspace_bytes(const combat_conv_args *a);
/* tile the launcher would pick for these args (a->tile honoured) and its stats granule */
int combat_conv_pick_tile(const combat_conv_args *a);
int combat_conv_stats_granule(int tile);
/* shape of the statistics array this launch would write: `rows` rows of [2][K]; if every row
 * lies inside one image, rows_per_image > 0 rows belong to each image in image order (what
 * InstanceNorm needs), else 0 */
int combat_conv_stats_layout(const combat_conv_args *a, int32_t *rows, int32_t *rows_per_image);

/* ------------------------------------------------------------------------------------------
 * Weight gradient: dW[n][tap][c] += sum_m dy[m][n] * prologue(src[pix(m,tap)][c])
 * replaces the weight-gradient half of conv backward for the same reference lines.
 * dw is fp32 [k_real][R*S][c_real] and must be zeroed by the caller (split-K partial sums are
 * accumulated with fp32 atomics).  c_real < C folds the hi/lo image channels: channel c of
 * src contributes to dw channel c % c_real for c < 2*c_real, others are dropped.
 * ------------------------------------------------------------------------------------------ */
typedef struct combat_wgrad_args {
    int32_t N, H, W, C;          /* src (conv input) */
    int32_t P, Q, K;             /* dy  (conv output gradient) */
    int32_t R, S, stride, pad;
    const void *src;             /* bf16 NHWC */
    const void *dy;              /* bf16 NHWC */
    float *dw;                   /* fp32 [k_real][R*S][c_real] */
    int32_t k_real;              /* real output channels (dy channels >= k_real are padding) */
    int32_t c_real;
    const float *pro_scale, *pro_shift;
    int32_t pro_group_stride;
    int32_t pro_act;
    float pro_slope;
    int32_t split;               /* number of pixel ranges: 0 = auto, > 0 explicit, < 0 = auto with the
                                    generic per-tap kernel forced (3x3/s1 layers otherwise use the
                                    LDS-patch kernel that computes all nine taps per workgroup) */
    void *workspace;             /* optional scratch (may be NULL): with at least
                                    combat_conv_wgrad_workspace_bytes(a) bytes the pixel ranges' partial
                                    sums are combined by plain stores + one reduction launch instead
                                    of fp32 atomics (the chip sustains ~1.3 TB/s of those) */
    int64_t workspace_bytes;
    int32_t defer_reduce;        /* 1: a launch that leaves partial sums in `workspace` does NOT combine them; the caller
                                    issues combat_conv_wgrad_reduce(a) later -- on any stream ordered after this launch,
                                    with the workspace untouched in between -- so that a chain of weight gradients is not
                                    a chain of (kernel, reduction) pairs.  Launches that need no reduction ignore it. */
    int32_t reserved;
    const struct combat_wgrad_args *reduce_first;
                                 /* NULL, or the arguments of an EARLIER launch (HOST pointer, read at launch time)
                                    that ran with defer_reduce = 1 on the same stream and has not been reduced yet: this
                                    launch's workgroups first add that launch's partial sums into ITS dw (each workgroup
                                    a share, fixed summation order, no atomics), then do their own work -- a chain of
                                    weight gradients then carries its reductions along instead of alternating with ~10-us
                                    reduction launches; only the chain's last one needs combat_conv_wgrad_reduce.  The
                                    two launches need disjoint workspaces.  A launch that cannot take it along (kernels
                                    other than the DMA-staged 3x3 one) reduces it in a launch of its own first: the
                                    results are the same either way. */
} combat_wgrad_args;

int combat_conv_wgrad(const combat_wgrad_args *a, void *stream);
/* dw += the partial sums a combat_conv_wgrad(a) launch with defer_reduce = 1 left in a->workspace (a no-op returning
 * COMBAT_OK where that launch needed no reduction).  `a` must be the same arguments. */
int combat_conv_wgrad_reduce(const combat_wgrad_args *a, void *stream);
/* scratch bytes the launch for these args can use (0: none) */
int64_t combat_conv_wgrad_workspace_bytes(const combat_wgrad_args *a);

/* fp32 [K][taps][c_real] master weights -> bf16 packed operands.
 * wf: forward  [rows_pad(K)][kpad_f], k = tap*C + c   (C = padded channel count, hi/lo dup if dup_hilo)
 * wd: dgrad    [rows_pad(C)][kpad_d], k = tap*K + n   (may be NULL)
 * replaces nothing in the reference (layout preparation for the two kernels above). */
int combat_pack_weights(const float *w, int32_t K, int32_t taps, int32_t c_real, int32_t C, int32_t dup_hilo,
                        void *wf, int32_t rows_pad_f, int32_t kpad_f,
                        void *wd, int32_t rows_pad_d, int32_t kpad_d, void *stream);
/* every convolution of a network in ONE launch (after each optimizer step all of them are stale):
 * `descs` is an array of n descriptors in DEVICE memory, fields as the arguments above */
typedef struct combat_pack_desc {
    const float *w;
    void *wf, *wd;
    int32_t K, taps, c_real, C, dup_hilo, rows_pad_f, kpad_f, rows_pad_d, kpad_d;
    int32_t reserved;
    const float *row_scale;   /* NULL, or [K]: output channel n is packed as row_scale[n] * w[n] -- an eval-mode
                               * BatchNorm that FOLLOWS the convolution folded into its weights (post-activation
                               * ResNet blocks, classifier_models/resnet.py:27-33: relu(bn2(conv2(.)) + shortcut)) */
} combat_pack_desc;
int combat_pack_weights_batch(const combat_pack_desc *descs, int32_t n, void *stream);

/* ------------------------------------------------------------------------------------------
 * Normalisation statistics (BatchNorm2d train: preact_resnet.py:20,22; InstanceNorm2d:
 * networks/models.py:278-313).  `partials` is the [rows][2][C] array written by the conv
 * epilogue (or by combat_group_stats); group g owns rows [g*rows_per_group, (g+1)*rows_per_group).
 * Outputs per (g, c): mean, rstd = 1/sqrt(var_biased + eps), and the conv-prologue pair
 * scale = gamma*rstd, shift = beta - mean*scale (gamma/beta NULL => 1/0).
 * If running_mean/var are given (groups must be 1) they are updated in place with `momentum`
 * and the unbiased variance, as nn.BatchNorm2d does; num_batches_tracked (int64) is bumped.
 * ------------------------------------------------------------------------------------------ */
int combat_norm_finalize(const float *partials, int32_t groups, int32_t rows_per_group, int32_t C,
                         float count, float eps, const float *gamma, const float *beta,
                         float *mean, float *rstd, float *scale, float *shift,
                         float *running_mean, float *running_var, float momentum,
                         int64_t *num_batches_tracked, float *scratch, int64_t scratch_bytes, void *stream);
/* rows_per_group > 256 is reduced in two launches through `scratch`
 * (>= combat_norm_scratch_bytes(groups, C) bytes); smaller problems ignore it (may be NULL). */
int64_t combat_norm_scratch_bytes(int32_t groups, int32_t C);

/* eval-mode BatchNorm folded to scale/shift from running statistics (preact_resnet.py:33,36 under
 * netC.eval(), train_generator.py:217) */
int combat_bn_eval_fold(const float *gamma, const float *beta, const float *running_mean,
                        const float *running_var, float eps, int32_t C, float *scale, float *shift,
                        void *stream);
/* all BatchNorm layers of a network in one launch; `descs`: n descriptors in DEVICE memory */
typedef struct combat_bn_desc {
    const float *gamma, *beta, *running_mean, *running_var;
    float *scale, *shift;
    int32_t C, reserved;
} combat_bn_desc;
int combat_bn_eval_fold_batch(const combat_bn_desc *descs, int32_t n, float eps, void *stream);

/* column sums of x (and x*x) over runs of rows_per_group rows, for tensors whose normalisation
 * groups do not align with a conv tile granule (or whose producer is not a conv):
 * x bf16 [groups*rows_per_group][C] -> partials fp32 [groups][2][C]   (rows_per_group <= 64) */
int combat_group_stats(const void *x, int32_t groups, int32_t rows_per_group, int32_t C,
                       float *partials, void *stream);

/* Norm backward, reduction half.  partials: [rows][2][C] = (sum dz, sum dz*xhat) per granule.
 * Produces the coefficients of  dx = ca*dz + cb*x + cc  per (g, c):
 *   ca = gamma*rstd, cb = -gamma*rstd^2*m2, cc = gamma*rstd*(mean*rstd*m2 - m1),
 *   m1 = sum(dz)/count, m2 = sum(dz*xhat)/count;
 * and (if dgamma/dbeta given, groups == 1)  dgamma = sum dz*xhat, dbeta = sum dz. */
int combat_norm_bwd_finalize(const float *partials, int32_t groups, int32_t rows_per_group, int32_t C,
                             float count, const float *gamma, const float *mean, const float *rstd,
                             float *ca, float *cb, float *cc, float *dgamma, float *dbeta,
                             float *scratch, int64_t scratch_bytes, void *stream);

/* Norm backward, apply half: dx = ca[g][c]*dz + cb[g][c]*x + cc[g][c] (+ add), bf16 in/out,
 * rows = groups_rows_total, group of row r = r / rows_per_group (stride 0 => single group). */
int combat_norm_bwd_apply(const void *dz, const void *x, const void *add, void *dx, int64_t rows,
                          int32_t C, int32_t rows_per_group, int32_t grouped,
                          const float *ca, const float *cb, const float *cc, void *stream);

/* Fused forms of the two chains above, one launch each (the step is bound by the number of dependent
 * ~6 us launches, not by bytes):
 *   combat_norm_act_fused  = combat_norm_finalize + combat_affine_act:  act = lrelu(x*scale + shift, slope)
 *                            with (mean, rstd, scale, shift, running stats) published as by the finalize;
 *   combat_norm_bwd_fused  = combat_norm_bwd_finalize + combat_norm_bwd_apply (ca/cb/cc stay in registers).
 * x / dz / add / act / dx: bf16 [groups][px_per_group][C]; count = px_per_group.  partials as for the
 * stand-alone calls ([groups*rows_per_group][2][C]; more than 256 rows per group are pre-reduced into
 * `scratch`).  partials == NULL (px_per_group <= 1024): the sums are taken from the tensors themselves,
 * which also replaces combat_group_stats / combat_group_stats_bwd for small InstanceNorm groups.
 * Same formulas and fp64 finalisation as the stand-alone kernels (nn.BatchNorm2d / nn.InstanceNorm2d,
 * classifier_models/preact_resnet.py:20-36, networks/models.py:278-313). */
int combat_norm_act_fused(const void *x, const float *partials, int32_t groups, int32_t rows_per_group,
                          int64_t px_per_group, int32_t C, float eps, float slope, const float *gamma,
                          const float *beta, float *mean, float *rstd, float *scale, float *shift,
                          float *running_mean, float *running_var, float momentum, int64_t *num_batches_tracked,
                          float *scratch, int64_t scratch_bytes, void *act, void *stream);
/* combat_norm_act_fused with a residual operand (post-activation ResNet blocks in train mode,
 * classifier_models/resnet.py:27-33): act = lrelu(x*scale + shift + (add*add_scale[c] + add_shift[c]), slope);
 * add_scale / add_shift NULL: the residual is added as is (identity shortcut); given: the shortcut's own
 * BatchNorm, finalised before this call (groups must be 1). */
int combat_norm_add_act_fused(const void *x, const float *partials, int32_t groups, int32_t rows_per_group,
                              int64_t px_per_group, int32_t C, float eps, float slope, const float *gamma,
                              const float *beta, float *mean, float *rstd, float *scale, float *shift,
                              float *running_mean, float *running_var, float momentum,
                              int64_t *num_batches_tracked, float *scratch, int64_t scratch_bytes, const void *add,
                              const float *add_scale, const float *add_shift, void *act, void *stream);
int combat_norm_bwd_fused(const void *dz, const void *x, const void *add, const float *partials, int32_t groups,
                          int32_t rows_per_group, int64_t px_per_group, int32_t C, const float *gamma,
                          const float *mean, const float *rstd, float *dgamma, float *dbeta, float *scratch,
                          int64_t scratch_bytes, void *dx, void *stream);

/* per-part (sum dz, sum dz*xhat), same layout as combat_group_stats:
 * xhat = (x - xh_mean[i][c]) * xh_rstd[i][c], i = part / parts_per_image (0 => i = 0) */
int combat_group_stats_bwd(const void *dz, const void *x, int32_t groups, int32_t rows_per_group,
                           int32_t C, int32_t parts_per_image, const float *xh_mean, const float *xh_rstd,
                           float *partials, void *stream);

/* ------------------------------------------------------------------------------------------
 * UNet decoder glue (networks/models.py:329-339):
 *   out = LeakyReLU_0.2( bilinear_up2x( y*sy[g][c] + ty[g][c]  [+ LeakyReLU(s*ss[g][c] + ts[g][c])] ) )
 * y, s: bf16 [N][H][W][C]; out: bf16 [N][2H][2W][C]; `u` (may be NULL) receives the pre-upsample
 * sum (bf16, [N][H][W][C]) when the caller needs it.  align_corners=False, as nn.Upsample.
 * The backward takes d_out and returns du = adjoint_up( d_out * lrelu'(out) ).
 * ------------------------------------------------------------------------------------------ */
int combat_unet_up_fwd(const void *y, const float *sy, const float *ty, const void *s, const float *ss,
                       const float *ts, int32_t N, int32_t H, int32_t W, int32_t C, void *out, void *stream);
int combat_unet_up_bwd(const void *d_out, const void *out, int32_t N, int32_t H, int32_t W, int32_t C,
                       void *du, void *stream);
/* combat_unet_up_bwd + the InstanceNorm backward of its result w.r.t. y (combat_norm_bwd_fused with the sums
 * taken from the tensors) in one launch: du = adjoint_up(d_out * lrelu'(out)) is stored (the encoder skip paths
 * add it), dx = ca*du + cb*y + cc with the coefficients of y's InstanceNorm (mean / rstd [N][C]).  H*W <= 1024. */
int combat_unet_up_bwd_fused(const void *d_out, const void *out, const void *y, const float *mean, const float *rstd,
                             int32_t N, int32_t H, int32_t W, int32_t C, void *du, void *dx, void *stream);
/* combat_norm_finalize (InstanceNorm, no affine) + combat_unet_up_fwd in one launch, for decoder inputs whose
 * normalised map is consumed only through the upsample: sy / ty are computed from y's partial rows
 * ([N * rows_per_group][2][C], rows_per_group <= 256) or, partials == NULL (H*W <= 1024), from y itself, and
 * published with mean / rstd for the backward pass. */
int combat_unet_up_fused(const void *y, const float *partials, int32_t rows_per_group, const void *s,
                         const float *ss, const float *ts, int32_t N, int32_t H, int32_t W, int32_t C, float eps,
                         float *mean, float *rstd, float *scale, float *shift, void *out, void *stream);
/* same sum without upsampling (level 0: u1 = IN(upconv1_0) + f0 feeds up() only, but the
 * encoder skip needs d(skip) = du * lrelu'(skip pre-activation) -- handled by conv epilogues) */

/* ------------------------------------------------------------------------------------------
 * Trigger: low_freq -> clamp-mix -> Gaussian blur (train_generator.py:47-55, 190-194, 224-226)
 *   lf  = P * noise * P^T           per image, per channel (P fp32 [hw][hw], see oracle.lowpass_matrix)
 *   bd  = clamp(x + lf*noise_rate, -1, 1)
 *   out = blur3x3(bd; k1[3] normalised 1-D kernel, reflect padding)
 * noise: bf16 NHWC c8 (channels 0..2) -- the generator's tanh output; x, out: fp32 NCHW [n][3][hw][hw].
 * out_c8 (may be NULL): the same result as the NHWC c8 hi/lo image the classifier stem reads.
 * mse_partial (may be NULL): fp32 [3n] per-(image, channel) sum (out - x)^2 (MSELoss, train_generator.py:234).
 * Backward: d_noise (bf16 NHWC c8) from d_out (fp32 NCHW) [+ 2*l2_scale*(out-x) MSE term];
 * pre_tanh != 0 multiplies by (1 - noise^2), i.e. returns the gradient w.r.t. the generator's
 * pre-tanh output (networks/models.py:340).
 * ------------------------------------------------------------------------------------------ */
int combat_trigger_fwd(const float *x, const void *noise, const float *P, const float *k1, float noise_rate,
                       int32_t n, int32_t hw, const int32_t *src_index /* NULL, or output image i is made from row
                       src_index[i] of x and noise: the poisoned sub-batch, train_generator.py:186-192 */,
                       float *out, void *out_c8, float *mse_partial, void *stream);
int combat_trigger_bwd(const float *x, const void *noise, const float *P, const float *k1, float noise_rate,
                       int32_t n, int32_t hw, const float *d_out, const float *d_out2 /* NULL, or added to d_out */,
                       const float *out, float l2_scale, int32_t pre_tanh, void *d_noise, void *stream);

/* ------------------------------------------------------------------------------------------
 * PostTensorTransform (utils/dataloader.py:45-60): per-sample crop(pad, integer offset) ->
 * rotation(bilinear, zeros, about the centre) -> horizontal flip, in one gather.
 * params: fp32 [n][4] = (crop_dx - pad, crop_dy - pad, angle_radians, flip) ; NULL = identity.
 * src_index (may be NULL): int32 [n] gather of the batch (train_generator.py:195 reorder).
 * x fp32 NCHW [*][3][hw][hw] -> out_c8 bf16 NHWC c8 hi/lo (and out_f32 NCHW if not NULL).
 * Backward: d_c8 bf16 NHWC c8 (channels 0..2 = gradient w.r.t. the image) -> d_x fp32 NCHW
 * (overwritten, or added to when accumulate != 0: two classifiers read the same image).
 * ------------------------------------------------------------------------------------------ */
int combat_augment_fwd(const float *x, const int32_t *src_index, const float *params, int32_t n, int32_t hw,
                       void *out_c8, float *out_f32, void *stream);
int combat_augment_bwd(const void *d_c8, int32_t c8_channels, const float *params, int32_t n, int32_t hw,
                       float *d_x, int32_t accumulate, void *stream);

/* ------------------------------------------------------------------------------------------
 * Classifier head: avg_pool(4) -> flatten -> Linear -> CrossEntropyLoss(mean)
 * (preact_resnet.py:99-101, resnet.py:93-96, train_generator.py:162,207,231,251).
 * feat bf16 [n][hw][hw][C] (hw = 4*ph), pooled features ordered (c, py, px) as torch's
 * NCHW flatten, written to `pooled` fp32 [n][in] (may be NULL).  Outputs logits fp32 [n][classes];
 * loss_sum += weight * sum_i CE_i / n (atomic); correct += #argmax==targets (int32 atomic);
 * correct2 += #argmax==targets2 (both optional; classes <= 16).
 * Backward: dlogits fp32 [n][classes] = loss_weight/n * (softmax - onehot);
 * d_feat bf16 [n][hw][hw][C] = dlogits W / 16 (may be NULL); and (if dW != NULL)
 * dW fp32 [classes][in] += dlogits^T pooled, db fp32 [classes] += column sums (accumulated INTO the
 * buffers: zero them first).  Eight sample ranges are summed first and a second launch adds them in
 * order (no atomics: the same bits every run); their partial sums live in scratch the library owns --
 * 2 MB per stream, hipMalloc'ed at the stream's first such call and kept (combat_colsum uses it too).
 * ------------------------------------------------------------------------------------------ */
int combat_head_fwd(const void *feat, int32_t n, int32_t hw, int32_t C, const float *W, const float *b,
                    int32_t classes, const int64_t *targets, float loss_weight, float *pooled, float *logits,
                    float *loss_sum, int32_t *correct, const int64_t *targets2, int32_t *correct2,
                    void *stream);
int combat_head_bwd(const float *pooled, int32_t n, int32_t hw, int32_t C, const float *W, int32_t classes,
                    const float *logits, const int64_t *targets, float loss_weight, float *dlogits,
                    void *d_feat, float *dW, float *db, void *stream);
/* combat_head_fwd + the feature half of combat_head_bwd (dlogits, d_feat; d_feat may be NULL) in ONE launch, for a
 * forward pass whose loss is differentiated right away (train_generator.py:207-208, 231, 251-254): same values as the
 * two calls.  combat_head_bwd_weights is then the other half of combat_head_bwd (dW += dlogits^T pooled, db += column
 * sums): nothing on the critical chain reads it, so it can run beside the input-gradient launches. */
int combat_head_fwd_bwd(const void *feat, int32_t n, int32_t hw, int32_t C, const float *W, const float *b,
                        int32_t classes, const int64_t *targets, float loss_weight, float *pooled, float *logits,
                        float *loss_sum, int32_t *correct, const int64_t *targets2, int32_t *correct2,
                        float *dlogits, void *d_feat, void *stream);
int combat_head_bwd_weights(const float *dlogits, const float *pooled, int32_t n, int32_t hw, int32_t C,
                            int32_t classes, float *dW, float *db, void *stream);

/* ------------------------------------------------------------------------------------------
 * SGD(momentum, weight_decay, nesterov) over a list of tensors (train_generator.py:123,125,212,255;
 * torch.optim.SGD semantics: g += wd*p; buf = first ? g : mu*buf + g; p -= lr*(g + mu*buf)).
 * ptrs: DEVICE array of 3*count pointers (param, grad, buf triples); sizes: DEVICE int64[count].
 * grad_scale multiplies the gradient first (1/world_size after an all-reduce sum).
 * ------------------------------------------------------------------------------------------ */
int combat_sgd_nesterov(const void *ptrs, const int64_t *sizes, int32_t count, int64_t max_size, float lr,
                        float momentum, float weight_decay, float grad_scale, int32_t first_step,
                        void *stream);

/* ------------------------------------------------------------------------------------------
 * Small data-movement kernels
 * ------------------------------------------------------------------------------------------ */
/* fp32 NCHW [n][c][h][w] -> bf16 NHWC c8 hi/lo image (c == 3) */
int combat_image_to_c8(const float *x, int32_t n, int32_t hw, void *out_c8, void *stream);
/* bf16 NHWC [rows][C] channels [0, c) -> fp32 NCHW [n][c][h][w] */
int combat_nhwc_to_nchw_f32(const void *x, int32_t n, int32_t h, int32_t w, int32_t C, int32_t c, float *out,
                            void *stream);
/* fp32 NCHW -> bf16 NHWC with C padded to a multiple of 8 */
int combat_nchw_to_nhwc_bf16(const float *x, int32_t n, int32_t c, int32_t h, int32_t w, int32_t C, void *out,
                             void *stream);
/* out = act > 0 ? g : 0 over `elements` bf16 (a multiple of 8): the gradient through the final ReLU of a
 * post-activation block whose consumer is not a convolution (the head; classifier_models/resnet.py:33,99) */
int combat_relu_mask(const void *g, const void *act, int64_t elements, void *out, void *stream);
/* hipMemsetAsync(ptr, 0, bytes): gradient buffers are accumulated into and must start at zero */
int combat_memset_zero(void *ptr, int64_t bytes, void *stream);
/* up to three byte copies (bytes == 0: skipped) as ONE kernel launch on `stream`; a source in pinned host memory is read
 * through its device mapping, an unmapped / unknown one falls back to hipMemcpyAsync.  The alternated step's table, batch
 * and re-packed bias at a step boundary (train_generator.py:169-171: inputs.to(device) and the per-step draws). */
int combat_copy3(void *dst0, const void *src0, int64_t bytes0, void *dst1, const void *src1, int64_t bytes1,
                 void *dst2, const void *src2, int64_t bytes2, void *stream);
/* column sums of a bf16 [rows][C] tensor into fp32 out[c_out] (overwritten): conv bias gradient.  Two launches: row
 * groups, then their sums in a fixed order (library-owned per-stream scratch, see combat_head_bwd). */
int combat_colsum(const void *x, int64_t rows, int32_t C, int32_t c_out, float *out, void *stream);
/* The logged-only terms of one step in one launch (train_generator.py:234-247): acc2[0] += MSE(inputs_bd, inputs) from the
 * trigger kernel's per-plane partial sums (mse_partial [3n], may be NULL), acc2[1] += loss_grad_l2 = MSE of the H- and
 * W-differences of F.pad(x, (1, 1, 2, 1)) and F.pad(xb, ...), hits += #{i : argmax(detector_logits[i]) == 1} ([n][2], may be
 * NULL).  x, xb: fp32 [n][3][hw][hw]; acc2, hits: fp64 device accumulators. */
int combat_log_terms(const float *x, const float *xb, const float *mse_partial, int32_t n, int32_t hw,
                     const float *detector_logits, double *acc2, double *hits, void *stream);
/* 2x2 max pool, bf16 NHWC (frequency model.py:21,32,43) */
int combat_maxpool2(const void *x, int32_t n, int32_t h, int32_t w, int32_t C, void *out, void *stream);
/* y = BN_eval(ELU(x)) elementwise per channel, bf16 in/out (frequency model.py:14-16) */
int combat_elu_affine(const void *x, int64_t rows, int32_t C, const float *scale, const float *shift, void *out,
                      void *stream);
/* out = lrelu(x * scale[g][c] + shift[g][c], slope) as a bf16 tensor, g = row / group_rows (0: one
 * group = BatchNorm; H*W: per image = InstanceNorm); scale/shift NULL: the activation alone.  The
 * train-mode normalisation + activation (preact_resnet.py:33,36; networks/models.py:322-340) that the
 * convolution prologues otherwise recompute per use: materialised once, the 3x3 forward and the
 * weight-gradient passes then take both operands by LDS-DMA. */
int combat_affine_act(const void *x, int64_t rows, int32_t C, const float *scale, const float *shift,
                      int64_t group_rows, float slope, void *out, void *stream);
/* DCT-II of the truncated 0..255 image (train_generator.py:245): x fp32 NCHW in [-1,1] ->
 * out bf16 NHWC c8 hi/lo of D*q*D^T, q = trunc((x+1)/2*255); D fp32 [hw][hw] */
int combat_dct_u8(const float *x, const float *D, int32_t n, int32_t hw, void *out_c8, void *stream);
/* plain fp32 linear head without pooling/loss: logits = flatten_chw(x) W^T + b  (frequency model.py:46-47) */
int combat_linear_nhwc(const void *x, int32_t n, int32_t h, int32_t w, int32_t C, const float *W, const float *b,
                       int32_t classes, float *logits, void *stream);

/* ---- WaNet trigger (train_generator_wanet.py:151-157, 196-202, 212; GridGenerator networks/models.py:344-385).
 * The reference's GridGenerator pools an affine-free InstanceNorm output (spatial mean exactly 0), so its output is
 * tanh(fc2(lrelu(fc1.bias))) for every input: a constant [2][S][S] field (tests/test_oracle_golden.py pins this
 * against the reference module).  Images are fp32 [N][3][H][H] planes; grid / noise_grid are fp32 [H][H][2]
 * ([...][0] = x), shared by the batch (per_image_grid 0) or one per image (1). */
/* field[2*S*S] = tanh(fc2_weight[nout][nf] . lrelu_0.2(fc1_bias) + fc2_bias)   (models.py:379-384 with f = 0) */
int combat_grid_head_fwd(const float *fc1_bias, const float *fc2_weight, const float *fc2_bias, int32_t nf, int32_t nout,
                         float *field, void *stream);
/* noise_grid = U field U^T (U = the [H][S] matrix of F.upsample(bicubic, align_corners=True), :152);
 * grid = clamp(identity_grid (1 - rescale) + noise_grid rescale, -1, 1)  (:155-156, identity_grid :560-562) */
int combat_wanet_grid(const float *field, const float *U, int32_t S, int32_t H, float rescale, float *noise_grid, float *grid,
                      void *stream);
/* out[i] = F.grid_sample(x[src_index ? src_index[i] : i], grid, align_corners=True)  bilinear, zeros padding (:157, :202) */
int combat_warp_fwd(const float *x, const int32_t *src_index, const float *grid, int32_t per_image_grid, int32_t n, int32_t H,
                    float *out, void *stream);
/* partial[g][H][H][2] = sum over the images of range g and the channels of (d_out + d_out2) * d grid_sample / d grid */
int combat_warp_bwd(const float *x, const float *d_out, const float *d_out2, const float *grid, int32_t per_image_grid,
                    int32_t n, int32_t H, int32_t groups, float *partial, void *stream);
/* d_x = d grid_sample / d input applied to d_out (d_x is overwritten).  Not on the training path (the images carry no
 * gradient, :196-202); completes F.grid_sample's backward. */
int combat_warp_bwd_input(const float *d_out, const float *grid, int32_t per_image_grid, int32_t n, int32_t H, float *d_x,
                          void *stream);
/* backward of combat_wanet_grid + combat_grid_head_fwd for a batch-shared grid, plus the gradient of
 * l2_scale * MSE(noise_grid, 0) (:212, :229): writes d fc1.bias [nf], d fc2.weight [nout][nf], d fc2.bias [nout]
 * (fc1.weight and the encoder receive exactly 0) and optionally d_field [nout] */
int combat_wanet_field_bwd(const float *partial, int32_t groups, const float *noise_grid, const float *U, int32_t S, int32_t H,
                           float rescale, float l2_scale, const float *field, const float *fc1_bias, const float *fc2_weight,
                           int32_t nf, float *d_fc1_bias, float *d_fc2_weight, float *d_fc2_bias, float *d_field,
                           void *stream);

/* ---- Plan replay -------------------------------------------------------------------------------------
 * The reference executes its step as a Python loop over ATen operators (train_generator.py:170-290, one host call
 * per operator).  Here a network pass is a fixed sequence of the entry points above; a combat_plan holds such a
 * sequence with its arguments and replays it from C: one foreign call per pass, hand-off events created once.
 *
 * Recording: combat_plan_record(plan, queue) arms the calling thread; the NEXT stream-taking entry point it calls
 * is captured (arguments by value, pointers as pointers: structs such as combat_conv_args must outlive the plan and
 * may be edited between replays) instead of launched, and returns COMBAT_OK.  queue < 0: replayed on the plan's own
 * stream; queue >= 0: replayed on auxiliary stream (queue % n_aux), ordered after every call recorded before it,
 * concurrent with the calls after it (weight gradients beside the input-gradient chain).
 * combat_plan_run replays calls [begin, end) and returns the first non-zero status (combat_plan_failed_call tells
 * which call); combat_plan_join orders `stream` after the auxiliary work issued so far (before a gradient
 * all-reduce, and at the end of a plan).  n_aux == 0 replays everything in line on `stream`.
 * Threading: one thread records / replays a given plan at a time (the single launching thread per rank that every
 * entry point of this library assumes: kernel-argument scratch and attribute flags are per process, not per thread). */
typedef struct combat_plan combat_plan;
combat_plan *combat_plan_create(void);
void combat_plan_destroy(combat_plan *plan);
int combat_plan_record(combat_plan *plan, int32_t queue);
/* the call recorded last additionally waits (on its own stream) for the completion of recorded call `call_index`
 * (0-based, earlier, on any queue): a reduction on the plan's stream behind a weight gradient on an auxiliary one */
int combat_plan_set_after(combat_plan *plan, int32_t call_index);
int combat_plan_record_cancel(void);   /* disarm (the call made was not a capturable entry point); 1 if it was armed */
int32_t combat_plan_size(const combat_plan *plan);
int combat_plan_run(combat_plan *plan, int32_t begin, int32_t end, void *stream, void *const *aux_streams, int32_t n_aux);
int combat_plan_join(combat_plan *plan, void *stream, void *const *aux_streams, int32_t n_aux);
int32_t combat_plan_failed_call(const combat_plan *plan);

/* ------------------------------------------------------------------------------------------
 * Gradient exchange of the data-parallel step (no counterpart in the reference, which has no distributed
 * code): a thin wrapper over RCCL for hosts that are not PyTorch -- the repo's own path uses torch.distributed's "nccl"
 * backend (= RCCL) on the same flat fp32 gradient buffers.  One communicator per rank-process / GPU: rank 0 calls
 * combat_comm_unique_id and ships the 128 bytes to the other ranks out of band, every rank calls combat_comm_init_rank;
 * combat_allreduce sums `count` elements of `buf` in place over the ranks, on `stream` (netC's gradients after Phase C,
 * netG's after Phase G: train_generator.py:208-212, :253-255 with the exchange in front of the optimiser step).
 * RCCL is looked up at first use (dlsym among the loaded libraries, else dlopen("librccl.so")): COMBAT_ELAUNCH if absent.
 * ------------------------------------------------------------------------------------------ */
#define COMBAT_COMM_UNIQUE_ID_BYTES 128
#define COMBAT_DTYPE_F32 0
#define COMBAT_DTYPE_BF16 1
int combat_comm_unique_id(void *out128);
int combat_comm_init_rank(void **comm, int32_t nranks, const void *unique_id128, int32_t rank);
int combat_comm_destroy(void *comm);
int combat_allreduce(void *buf, int64_t count, int32_t dtype, void *comm, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* COMBAT_HIP_H */
