/*
 * zsv_hip.h -- C ABI of libzsv_hip.so: the MI355X (gfx950) kernels behind the
 * 3D-conv video hot path of damien911224/ZeroShotVideoClassification.
 *
 * The reference has no native code and no FFI of its own: every entry point below
 * replaces an ATen operator that the reference's Python reaches through torch.nn
 * (SURVEY.md section 2a / 8b).  Each declaration cites the reference call site whose
 * arithmetic it supplies.  Conventions:
 *
 *   - all tensors are contiguous fp32, channel-major "NCS": (N, C, T, H, W) with
 *     S = T*H*W; pointers are device pointers owned by the caller;
 *   - kernels never allocate: workspaces are passed in (query the size first);
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on it;
 *   - return value: 0 = OK, otherwise a ZSV_E_* code (zsv_status_string() names it);
 *     nothing throws across this boundary;
 *   - re-entrant and thread-safe (backward is called from autograd's worker thread).
 */
#ifndef ZSV_HIP_H
#define ZSV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZSV_OK 0
#define ZSV_E_BAD_SHAPE 1      /* inconsistent or unsupported geometry            */
#define ZSV_E_NULL 2           /* required pointer is NULL                        */
#define ZSV_E_WORKSPACE 3      /* workspace smaller than the *_workspace_bytes()  */
#define ZSV_E_TOO_LARGE 4      /* a tensor has >= 2^31 elements                   */
#define ZSV_E_LAUNCH 5         /* hipGetLastError() after the launch was not OK   */
#define ZSV_E_UNSUPPORTED 6    /* valid but not implemented (e.g. overlapping pool) */

/* Geometry of one Conv3d call site.  Replaces the arguments of
 * nn.Conv3d / F.conv3d at resnet.py:40-52 (Conv2Plus1D spatial 1xkxk and temporal
 * tx1x1), resnet.py:23-30 (3x3x3), resnet.py:63-70 (1x3x3), resnet.py:170,181,184
 * (stems), resnet.py:270 (1x1x1 strided shortcut) and network.py:102-117 (C3D).
 * Dilation is 1 and groups is 1 everywhere in the reference. */
typedef struct zsv_conv_desc {
    int32_t N, Cin, Ti, Hi, Wi;    /* input  (N, Cin, Ti, Hi, Wi)                 */
    int32_t Cout, To, Ho, Wo;      /* output (N, Cout, To, Ho, Wo)                */
    int32_t kT, kH, kW;            /* weight (Cout, Cin, kT, kH, kW)              */
    int32_t sT, sH, sW;            /* stride                                      */
    int32_t pT, pH, pW;            /* zero padding                                */
} zsv_conv_desc;

const char* zsv_status_string(int status);
/* build identification: "zsv_hip gfx950 <date>" */
const char* zsv_version(void);
/* The ZSV_* debug / A-B environment switches are snapshotted when the library is loaded; the launch
 * path never calls getenv().  After changing one of them in a live process call this (no launch in flight on another
 * thread); returns the number of switches that are set.  Not needed in normal use. */
int32_t zsv_reload_knobs(void);

/* ---- convolution (aten::conv3d and its autograd formulas) ------------------- */
/* y = conv3d(x, w) (+ bias[c]) (relu optional, for network.py:147-162 `relu(conv(x))`).
 * `workspace` holds the weights re-packed tap-major for the fast kernel (query the size;
 * 0 means the call needs none).  It must be 16-byte aligned. */
size_t zsv_conv3d_fwd_workspace_bytes(const zsv_conv_desc* d);
/* Same, and the epilogue also emits BatchNorm partial statistics of y: bn_partials[0..Cout*tiles)
 * = per (channel, column tile) sums, then Cout*tiles sums of squares, tiles =
 * zsv_conv3d_fwd_stat_tiles(d, y) (0 = this geometry cannot; then call the plain form).  Only
 * without bias / ReLU.  Feeds zsv_bn_fwd_train_stats, which then skips its pass over y. */
int32_t zsv_conv3d_fwd_stat_tiles(const zsv_conv_desc* d, const float* y);
int zsv_conv3d_fwd_stats(const zsv_conv_desc* d, const float* x, const float* w, const float* bias,
                         float* y, int fuse_relu, float* bn_partials, int32_t stat_tiles, void* workspace,
                         size_t workspace_bytes, void* stream);
/* Inference forward with the block tail fused: y = relu?(conv3d(x, w) + bias + residual) -- with the
 * eval-mode BatchNorm folded into (w, bias) this is Conv3d -> BatchNorm3d -> `out += residual` -> ReLU
 * (resnet.py:97,110-111) in one pass.  Only where zsv_conv3d_fwd_add_supported(d) != 0 (tap kernel, no
 * split-K); `residual` has y's shape.  zsv_conv3d_fwd_full is the common form of the three entry points. */
int32_t zsv_conv3d_fwd_add_supported(const zsv_conv_desc* d);
int zsv_conv3d_fwd_add(const zsv_conv_desc* d, const float* x, const float* w, const float* bias,
                       const float* residual, float* y, int fuse_relu, void* workspace, size_t workspace_bytes,
                       void* stream);
int zsv_conv3d_fwd_full(const zsv_conv_desc* d, const float* x, const float* w, const float* bias,
                        const float* residual, float* y, int fuse_relu, float* bn_partials, int32_t stat_tiles,
                        void* workspace, size_t workspace_bytes, void* stream);
int zsv_conv3d_fwd(const zsv_conv_desc* d, const float* x, const float* w, const float* bias,
                   float* y, int fuse_relu, void* workspace, size_t workspace_bytes, void* stream);
/* dx = conv3d_input_grad(dy, w): what autograd runs for every conv but the first.  Strided
 * convolutions are solved per residue class of input voxels (no multiplies by inserted zeros). */
size_t zsv_conv3d_dgrad_workspace_bytes(const zsv_conv_desc* d);
int zsv_conv3d_dgrad(const zsv_conv_desc* d, const float* dy, const float* w, float* dx,
                     void* workspace, size_t workspace_bytes, void* stream);
/* dx = conv3d_input_grad(dy, w) + add: the gradient arriving over an identity shortcut
 * (`out += residual`, resnet.py:108-110, residual = the convolution's own input) is added in the
 * epilogue instead of by a separate pass.  Only where zsv_conv3d_dgrad_add_supported(d) != 0 (stride 1,
 * no split-K); `add` has dx's shape; add == NULL = zsv_conv3d_dgrad. */
int32_t zsv_conv3d_dgrad_add_supported(const zsv_conv_desc* d);
int zsv_conv3d_dgrad_add(const zsv_conv_desc* d, const float* dy, const float* w, const float* add, float* dx,
                         void* workspace, size_t workspace_bytes, void* stream);
/* dx = conv3d_input_grad(dy, w) + the gradient of the block's strided 1x1x1 shortcut convolution, which reads the same input
 * (`downsample`, resnet.py:240-246, next to the strided first convolution of the block, resnet.py:40-45).  `sub` is that gradient
 * in COMPACT form [N][Cin][ceil(Ti/st)][Hi/sh][Wi/sw] -- the input gradient of the 1x1x1 convolution taken at stride 1 over its
 * own output voxels -- and is added where dx[.., st*a, sh*b, sw*c] is produced: no zero-filled full-size tensor, no separate add
 * over the block input (aten's autograd: two full-size gradients summed).  Only where
 * zsv_conv3d_dgrad_add_strided_supported(d, st, sh, sw) != 0 (the merged stride-(1,2,2) kernel; st = 1 or 2, sh = sw = 2);
 * workspace as zsv_conv3d_dgrad. */
int32_t zsv_conv3d_dgrad_add_strided_supported(const zsv_conv_desc* d, int32_t st, int32_t sh, int32_t sw);
int zsv_conv3d_dgrad_add_strided(const zsv_conv_desc* d, const float* dy, const float* w, const float* sub, int32_t st, int32_t sh,
                                 int32_t sw, float* dx, void* workspace, size_t workspace_bytes, void* stream);
/* dw = conv3d_weight_grad(x, dy).  Deterministic: position range is cut into a fixed
 * number of slices, each slice writes a partial slab into `workspace`, a second kernel
 * sums the slabs in slice order. */
size_t zsv_conv3d_wgrad_workspace_bytes(const zsv_conv_desc* d);
int zsv_conv3d_wgrad(const zsv_conv_desc* d, const float* x, const float* dy, float* dw,
                     void* workspace, size_t workspace_bytes, void* stream);
/* The tap-validity table the stride-1 weight-gradient kernels read depends on the geometry only (input extents, kernel,
 * padding): zsv_conv3d_wgrad rebuilds it on every call; a caller that keeps it -- zsv_conv3d_wgrad_mask_bytes(d) bytes
 * (0: this geometry's kernel reads no table), filled once by zsv_conv3d_wgrad_mask -- passes it to zsv_conv3d_wgrad_masked
 * and saves a launch per call (mask == NULL behaves like zsv_conv3d_wgrad).  Same results bit for bit. */
size_t zsv_conv3d_wgrad_mask_bytes(const zsv_conv_desc* d);
int zsv_conv3d_wgrad_mask(const zsv_conv_desc* d, void* mask, void* stream);
int zsv_conv3d_wgrad_masked(const zsv_conv_desc* d, const float* x, const float* dy, float* dw, void* workspace,
                            size_t workspace_bytes, const void* mask, void* stream);

/* ---- convolution fed by a BatchNorm + ReLU that is never written out ------------------------------ */
/* Conv2Plus1D runs `Conv3d(1x3x3) -> BatchNorm3d -> ReLU -> Conv3d(3x1x1)` (resnet.py:40-52).  Instead of one HBM pass
 * that normalises the 636 MB mid tensor and a second that reads it back, the temporal convolution (forward and weight
 * gradient) reads the RAW spatial output x and applies relu(x * scale[c] + shift[c]) on its way into the MFMA -- the same
 * fmaf / max as the BatchNorm apply pass: results are bit-identical to the unfused sequence.
 *   zsv_bn_fwd_train_coeffs : training-mode BatchNorm up to, but without, the normalise pass: batch statistics (from the
 *       producing convolution's epilogue partials when given), save_mean / save_invstd, running statistics, and
 *       coef = [2][coef_pitch] (scale row, shift row; coef_pitch >= C, a multiple of 16, tail zeroed; 16-byte aligned).
 *   zsv_conv3d_pre_supported: 1 when both zsv_conv3d_fwd_pre and zsv_conv3d_wgrad_pre can run this geometry.
 *   zsv_conv3d_fwd_pre / zsv_conv3d_wgrad_pre: zsv_conv3d_fwd_stats / zsv_conv3d_wgrad with x read through the affine + ReLU
 *       (same workspace sizes as their plain forms).  The input gradient is zsv_conv3d_dgrad as usual (it is the gradient
 *       w.r.t. the virtual activation), followed by zsv_bn_bwd(relu_mode = 2), which recomputes the ReLU mask from x. */
int zsv_bn_fwd_train_coeffs(const float* x, int32_t N, int32_t C, int32_t S, const float* gamma, const float* beta,
                            float* save_mean, float* save_invstd, float* running_mean, float* running_var, float momentum,
                            float eps, const float* conv_partials, int32_t stat_tiles, float* coef, int32_t coef_pitch,
                            void* workspace, size_t workspace_bytes, void* stream);
int32_t zsv_conv3d_pre_supported(const zsv_conv_desc* d);
int zsv_conv3d_fwd_pre(const zsv_conv_desc* d, const float* x, const float* pre_coef, int32_t coef_pitch, const float* w,
                       float* y, float* bn_partials, int32_t stat_tiles, void* workspace, size_t workspace_bytes,
                       void* stream);
int zsv_conv3d_wgrad_pre(const zsv_conv_desc* d, const float* x, const float* pre_coef, int32_t coef_pitch,
                         const float* dy, float* dw, void* workspace, size_t workspace_bytes, void* stream);
/* db[c] = sum over (n, s) of dy  (bias gradient of C3D's convs, network.py:102-117,
 * and of nn.Linear when S == 1). */
size_t zsv_channel_sum_workspace_bytes(int32_t N, int32_t C, int32_t S);
int zsv_channel_sum(const float* dy, int32_t N, int32_t C, int32_t S, float* db,
                    void* workspace, size_t workspace_bytes, void* stream);

/* ---- BatchNorm3d (+ fused residual add / ReLU) ------------------------------- */
/* Training-mode forward of nn.BatchNorm3d (resnet.py:48,95,97,183,186,272) followed,
 * optionally, by `out += residual` (resnet.py:110) and ReLU (resnet.py:49,95,111):
 *   mean/var over (N, S) per channel (biased var for normalising),
 *   running_mean/var updated in place with `momentum` (unbiased var), like torch;
 *   y = relu?((x - mean) * invstd * gamma + beta + residual?)
 * save_mean / save_invstd (C floats each) are kept for backward. */
size_t zsv_bn_workspace_bytes(int32_t N, int32_t C, int32_t S);
int zsv_bn_fwd_train(const float* x, int32_t N, int32_t C, int32_t S, const float* gamma,
                     const float* beta, const float* residual, int fuse_relu, float* y,
                     float* save_mean, float* save_invstd, float* running_mean,
                     float* running_var, float momentum, float eps, void* workspace,
                     size_t workspace_bytes, void* stream);
/* Same with the batch statistics taken from the producing convolution's epilogue partials
 * (zsv_conv3d_fwd_stats) instead of a pass over x. */
int zsv_bn_fwd_train_stats(const float* x, int32_t N, int32_t C, int32_t S, const float* gamma,
                           const float* beta, const float* residual, int fuse_relu, float* y,
                           float* save_mean, float* save_invstd, float* running_mean,
                           float* running_var, float momentum, float eps, const float* conv_partials,
                           int32_t stat_tiles, void* workspace, size_t workspace_bytes, void* stream);
/* Eval-mode forward (model.eval(), main.py:229): uses the running statistics. */
int zsv_bn_fwd_eval(const float* x, int32_t N, int32_t C, int32_t S, const float* gamma,
                    const float* beta, const float* running_mean, const float* running_var,
                    const float* residual, int fuse_relu, float eps, float* y, void* workspace,
                    size_t workspace_bytes, void* stream);
/* Backward of the fused op.  fuse_relu: 0 = none; 1 = ReLU mask from the saved OUTPUT `y`
 * (cf. nn.ReLU(inplace=True), which also keeps only its output); 2 = ReLU mask recomputed from x
 * with the forward's exact fma (x*scale + shift > 0) -- valid when the forward had no residual
 * input, `y` may then be NULL and one tensor read is saved per pass.  Writes dx, dgamma, dbeta
 * and, when d_residual != NULL, the gradient flowing into the residual branch (= masked dy). */
int zsv_bn_bwd(const float* dy, const float* x, const float* y, int32_t N, int32_t C, int32_t S,
               const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
               int fuse_relu, float* dx, float* d_residual, float* dgamma, float* dbeta,
               void* workspace, size_t workspace_bytes, void* stream);

/* ---- elementwise -------------------------------------------------------------- */
/* nn.ReLU / F.relu (resnet.py:49; network.py:147-166,614). */
int zsv_relu_fwd(const float* x, float* y, int64_t n, void* stream);
/* dx = dy * (y > 0), y = saved output. */
int zsv_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream);
/* dx = dy * (y > 0) AND db[c] = sum over (n, s) of that, in one pass (backward of C3D's fused `relu(conv(x) + bias)`,
 * network.py:147-162); (N, C, S) tensors; workspace as zsv_channel_sum. */
int zsv_relu_bwd_bias(const float* dy, const float* y, float* dx, int32_t N, int32_t C, int32_t S, float* db,
                      void* workspace, size_t workspace_bytes, void* stream);
/* out = relu(a + b): `out += residual; relu(out)` (resnet.py:110-111) when no BN is fused. */
int zsv_add_relu_fwd(const float* a, const float* b, float* y, int64_t n, void* stream);

/* ---- pooling -------------------------------------------------------------------- */
/* torch.mean(f, dim=(2,3,4)) (network.py:595) / AdaptiveAvgPool3d(1) (resnet.py:251). */
int zsv_meanpool_fwd(const float* x, int32_t N, int32_t C, int32_t S, float* y, void* stream);
int zsv_meanpool_bwd(const float* dy, int32_t N, int32_t C, int32_t S, float* dx, void* stream);
/* nn.MaxPool3d with kernel == stride (network.py:103-118); padding is -inf padding.
 * `argmax` (int32, one per output element) holds the flat (t*H+h)*W+w input index. */
int zsv_maxpool3d_fwd(const float* x, int32_t N, int32_t C, int32_t Ti, int32_t Hi, int32_t Wi,
                      int32_t kT, int32_t kH, int32_t kW, int32_t pT, int32_t pH, int32_t pW,
                      int32_t To, int32_t Ho, int32_t Wo, float* y, int32_t* argmax, void* stream);
int zsv_maxpool3d_bwd(const float* dy, const int32_t* argmax, int32_t N, int32_t C, int32_t Ti,
                      int32_t Hi, int32_t Wi, int32_t kT, int32_t kH, int32_t kW, int32_t pT,
                      int32_t pH, int32_t pW, int32_t To, int32_t Ho, int32_t Wo, float* dx,
                      void* stream);

/* ---- dense head ----------------------------------------------------------------- */
/* nn.Linear (network.py:611-616 MLP, :120,132 fc6/regressor): y = x W^T + b, optional ReLU.
 * x (rows, in), w (out, in), y (rows, out).  Implemented on the same MFMA GEMM core. */
size_t zsv_linear_fwd_workspace_bytes(int32_t rows, int32_t in_features, int32_t out_features);
int zsv_linear_fwd(const float* x, const float* w, const float* bias, float* y, int32_t rows,
                   int32_t in_features, int32_t out_features, int fuse_relu, void* workspace,
                   size_t workspace_bytes, void* stream);
size_t zsv_linear_dgrad_workspace_bytes(int32_t rows, int32_t in_features, int32_t out_features);
int zsv_linear_dgrad(const float* dy, const float* w, float* dx, int32_t rows, int32_t in_features,
                     int32_t out_features, void* workspace, size_t workspace_bytes, void* stream);
size_t zsv_linear_wgrad_workspace_bytes(int32_t rows, int32_t in_features, int32_t out_features);
int zsv_linear_wgrad(const float* x, const float* dy, float* dw, int32_t rows,
                     int32_t in_features, int32_t out_features, void* workspace,
                     size_t workspace_bytes, void* stream);

/* ---- cosine nearest classes (evaluate / train accuracy) -------------------------------------- */
/* cdist(embed, class_embed, 'cosine').argsort(1)[:, :k] of compute_accuracy (main.py:316-325; k = 5 and
 * k = 1) and of the per-step train accuracy (main.py:182-185).  embed (rows, dim) and class_embed
 * (n_classes, dim) fp32, not necessarily normalised.  Distances 1 - <u,v>/(|u||v|) are formed in double
 * precision on the matrix core (v_mfma_f64_16x16x4_f64), like scipy's double cdist; out_index (rows, k)
 * int32 = the k nearest classes, ties by lower index (stable argsort); out_dist (rows, k) double may be
 * NULL.  workspace = the (rows, n_classes rounded up to 16) double distance matrix. */
size_t zsv_cosine_topk_workspace_bytes(int32_t rows, int32_t n_classes);
int zsv_cosine_topk(const float* embed, const float* class_embed, int32_t rows, int32_t dim, int32_t n_classes,
                    int32_t k, int32_t* out_index, double* out_dist, void* workspace, size_t workspace_bytes,
                    void* stream);

/* ---- optimiser ------------------------------------------------------------------ */
/* One torch.optim.Adam step (main.py:131; betas (0.9, 0.999), eps 1e-8, no weight decay,
 * no amsgrad) over a flat fp32 buffer: p, g, exp_avg, exp_avg_sq of n elements.
 * `step` is the 1-based step count (bias correction). */
int zsv_adam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, int64_t n,
                  float lr, float beta1, float beta2, float eps, int32_t step, void* stream);

/* ---- bf16 inference path (BASELINE config 5: 32-frame bf16 eval, main.py:224-313) ------------ */
/* Forward-only convolution with eval-mode BatchNorm folded in:
 *     y = relu?( conv3d(x, w * scale[cout]) + shift[cout] (+ residual) )
 * = Conv3d -> BatchNorm3d.eval() -> ReLU (resnet.py:40-52,94-98) and `out += residual; relu`
 * (resnet.py:110-111); scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale.
 * bf16 x bf16 products accumulate in fp32 (v_mfma_f32_16x16x32_bf16).
 *
 * Activations are channels-last bf16: [N][T][H][W][Cp] with Cp = zsv_bf16_channel_pitch(C)
 * (C rounded up to 32; pad channels are written as zero).  A clip (Cin <= 4) is the exception: it is
 * stored [N][T][Hi][Wi][4] with the H/W zero border already materialised (zsv_clip_to_bf16), so its
 * conv desc carries the padded Hi/Wi and pH = pW = 0, and needs (Wo-1)*sW + 8 <= Wi.
 * `blob` = the packed bf16 weights followed by the fp32 shifts (zsv_conv3d_bf16_blob_bytes), built
 * once per layer by zsv_conv3d_bf16_pack (scale / shift may be NULL = 1 / 0: plain conv or bias).
 * `residual` (may be NULL) and y have the output's layout. */
int32_t zsv_bf16_channel_pitch(int32_t channels);
size_t zsv_conv3d_bf16_blob_bytes(const zsv_conv_desc* d);
int zsv_conv3d_bf16_pack(const zsv_conv_desc* d, const float* w, const float* scale, const float* shift,
                         void* blob, void* stream);
/* The blob of the INPUT-GRADIENT problem of a convolution (bf16 training step, amp.py): `d` describes that problem -- a
 * stride-1 convolution of the (zero-interleaved) output gradient, d->Cin = the forward's Cout, d->Cout = the forward's Cin,
 * padding k-1-p -- and `w_fwd` is the FORWARD weight (forward Cout, forward Cin, kT, kH, kW) as the module holds it: channel
 * roles are swapped and the taps flipped while packing (aten::convolution_backward's input gradient, resnet.py:40-52). */
int zsv_conv3d_bf16_pack_dgrad(const zsv_conv_desc* d, const float* w_fwd, void* blob, void* stream);
int zsv_conv3d_bf16_fwd(const zsv_conv_desc* d, const void* x, const void* blob, const void* residual,
                        int fuse_relu, void* y, void* stream);
/* The same forward in front of a training-mode BatchNorm (no residual, no ReLU) with the BatchNorm's batch statistics taken in the
 * epilogue: per column tile the per-channel sum and sum of squares of the values AS STORED (rounded to bf16: what a pass over y would
 * read) go to bn_partials[row][0 / 1][Cp] (Cp = Cout rounded up to 32), `*rows` rows of them; zsv_conv3d_bf16_stat_rows(d) is an
 * upper bound for the caller's allocation.  zsv_bn_cl_fwd_train_stats then skips its statistics pass (resnet.py:46-47,96-97 under
 * main.py:172's autocast). */
int32_t zsv_conv3d_bf16_stat_rows(const zsv_conv_desc* d);
int zsv_conv3d_bf16_fwd_stats(const zsv_conv_desc* d, const void* x, const void* blob, void* y, float* bn_partials,
                              int32_t rows_capacity, int32_t* rows, void* stream);
/* (N, C<=4, T, H, W) fp32 clip -> [N][T][Hp][Wp][4] bf16, the frame placed at (padH, padW) inside a
 * zero border; Hp >= H + padH, Wp >= W + padW. */
int zsv_clip_to_bf16(const float* x, int32_t N, int32_t C, int32_t T, int32_t H, int32_t W, int32_t padH,
                     int32_t padW, int32_t Hp, int32_t Wp, void* out, void* stream);
/* nn.MaxPool3d with kernel == stride (network.py:148-163: C3D's five pools) on channels-last bf16 [N][T][H][W][Cp]; padding is
 * -inf padding (2 * pad <= kernel).  The bf16 evaluation engine of network.C3D (inference.Bf16EngineC3D). */
int zsv_maxpool3d_bf16(const void* x, int32_t N, int32_t C, int32_t Ti, int32_t Hi, int32_t Wi, int32_t kT, int32_t kH, int32_t kW,
                       int32_t pT, int32_t pH, int32_t pW, int32_t To, int32_t Ho, int32_t Wo, void* y, void* stream);
/* [N][S][Cp] bf16 -> (N, C) fp32 mean over the S voxels (resnet.py:251-254 avgpool + flatten). */
int zsv_meanpool_bf16(const void* x, int32_t N, int32_t S, int32_t C, float* out, void* stream);

/* ---- bf16 TRAINING step: the reference's mixed-precision step (main.py:172 `with autocast():`, main.py:137,195-203
 * GradScaler) on channels-last bf16 activations [R = N*T*H*W][Cp] (the layout of the bf16 convolution above).  Under
 * autocast aten::batch_norm keeps fp32 statistics / affine parameters on a reduced-precision input and writes the reduced
 * precision; `out += residual; relu` (resnet.py:110-111) are element-wise passes in it.
 * zsv_bn_cl_fwd_train: train-mode BatchNorm3d (resnet.py:42,50,96,97,184,272) of the raw convolution output z: batch
 *   statistics (fp32 / fp64 accumulation), running statistics updated as torch does (unbiased variance, `momentum`),
 *   y = relu?(gamma*(z-mean)*invstd + beta (+ residual)) rounded once to bf16; save_mean / save_invstd (C floats) for backward.
 * zsv_bn_cl_bwd: g = dy * (y > 0) when relu_mask (y = the forward's output), dgamma / dbeta (C floats, may be NULL),
 *   dz (bf16) = the BatchNorm input gradient; g_out (may be NULL) receives g itself -- the gradient of a residual branch.
 *   save_coef (may be NULL): 2 * Cp floats, the forward's scale a and shift b per (padded) channel.  Passed back as `fwd_coef`
 *   with y == NULL (no residual in the forward) the backward recomputes the ReLU mask as fma(z, a, b) > 0 -- exactly what the
 *   forward rounded to y -- and does not read the saved output at all (4 of its 14 bytes per element).
 * workspace: zsv_bn_cl_workspace_bytes(R, C) bytes (0 = unsupported shape). */
size_t zsv_bn_cl_workspace_bytes(int64_t R, int32_t C);
int zsv_bn_cl_fwd_train(const void* z, const void* residual, int64_t R, int32_t C, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum, float eps, int fuse_relu, void* y,
                        float* save_mean, float* save_invstd, float* save_coef, void* workspace, size_t workspace_bytes, void* stream);
/* The same with the batch statistics taken by the producing convolution's epilogue (zsv_conv3d_bf16_fwd_stats: `conv_rows` rows of
 * [2][Cp] partial sums): the statistics pass over z is skipped. */
int zsv_bn_cl_fwd_train_stats(const void* z, const void* residual, int64_t R, int32_t C, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps, int fuse_relu, void* y,
                              float* save_mean, float* save_invstd, float* save_coef, const float* conv_partials, int32_t conv_rows,
                              void* workspace, size_t workspace_bytes, void* stream);
int zsv_bn_cl_bwd(const void* dy, const void* y, const void* z, int64_t R, int32_t C, const float* gamma, const float* save_mean,
                  const float* save_invstd, const float* fwd_coef, int relu_mask, void* dz, void* g_out, float* dgamma, float* dbeta,
                  void* workspace, size_t workspace_bytes, void* stream);
/* Weight gradient of a convolution from channels-last bf16 operands (x [N][Ti][Hi][Wi][CinP], dz [N][To][Ho][Wo][CoutP]) with
 * fp32 accumulation: dw (Cout, Cin, kT, kH, kW) fp32 -- aten::convolution_backward's weight gradient under autocast
 * (resnet.py:40-52 1x3x3 / 3x1x1 incl. the strided ones, resnet.py:23-30 3x3x3, resnet.py:270 1x1x1).  The workspace holds
 * per-slice partial sums; zsv_conv3d_bf16_wgrad_workspace_bytes() == 0 means "not this kernel's geometry" (the clip
 * convolution, other kernel shapes, tensors of 4 GiB and more): the caller converts the operands (below) and uses
 * zsv_conv3d_wgrad. */
size_t zsv_conv3d_bf16_wgrad_workspace_bytes(const zsv_conv_desc* d);
int zsv_conv3d_bf16_wgrad(const zsv_conv_desc* d, const void* x, const void* dz, float* dw, void* workspace, size_t workspace_bytes,
                          void* stream);
/* C3D's half of the mixed-precision step (network.py:147-163 under autocast: `relu(conv(x) + bias)` and MaxPool3d backward):
 * zsv_maxpool3d_bf16_bwd: gradient of zsv_maxpool3d_bf16 -- dy goes to each window's FIRST maximum in (t, h, w) order per channel
 *   (aten::max_pool3d_with_indices' argmax, recomputed from the saved input x), zero elsewhere;
 * zsv_relu_bias_bwd_cl: g = dy * (y > 0) in bf16 and dbias[c] = sum of g over the R rows (fp32 / fp64 accumulation); workspace =
 *   zsv_bn_cl_workspace_bytes(R, C). */
int zsv_maxpool3d_bf16_bwd(const void* dy, const void* x, int32_t N, int32_t C, int32_t Ti, int32_t Hi, int32_t Wi, int32_t kT, int32_t kH,
                           int32_t kW, int32_t pT, int32_t pH, int32_t pW, int32_t To, int32_t Ho, int32_t Wo, void* dx, void* stream);
int zsv_relu_bias_bwd_cl(const void* dy, const void* y, int64_t R, int32_t C, void* g_out, float* dbias, void* workspace,
                         size_t workspace_bytes, void* stream);
/* layout converters between the two activation layouts: [N][S][Cp] bf16 <-> (N, C, S) fp32 (C > 4; pad channels read as /
 * written with zero): they hand a bf16 tensor to the fp32 NCDHW kernels (weight gradients of the mixed-precision step). */
int zsv_cl_bf16_to_ncs_f32(const void* x, int32_t N, int32_t S, int32_t C, float* out, void* stream);
int zsv_ncs_f32_to_cl_bf16(const float* x, int32_t N, int32_t S, int32_t C, void* out, void* stream);
/* gradient of zsv_meanpool_bf16 (network.py:595 under autocast): dx[n][s][c] = dpooled[n][c] / S in bf16 */
int zsv_meanpool_bf16_bwd(const float* dpooled, int32_t N, int32_t S, int32_t C, void* dx, void* stream);

/* ---- clip pre-processing ------------------------------------------------------ */
/* The reference's transform chain (auxiliary/transforms.py:41-56): (u8/255 - 1)/2 and THWC->CTHW
 * (:116-117), bilinear resize of the short side to 128 with align_corners=False (:99-107), a
 * crop x crop window at (top, left) of the resized frame (:80-97,132-158) and an optional horizontal
 * flip (:188-195), fused into one pass.  frames_u8: (N, T, Hin, Win, 3) uint8;
 * crop_flip_params_device: N x {top, left, flip} int32 on the device; out: (N, 3, T, crop, crop)
 * fp32.  Hres/Wres = floor(Hin*scale), floor(Win*scale); inv_scale = (float)(1/scale). */
int zsv_clip_transform(const uint8_t* frames_u8, int32_t N, int32_t T, int32_t Hin, int32_t Win,
                       int32_t Hres, int32_t Wres, float inv_scale, int32_t crop,
                       const int32_t* crop_flip_params_device, float* out, void* stream);

/* The same update for every parameter tensor of a model in ONE launch (the reference's
 * optimizer.step(), main.py:200, is ~113 tensors).  `table_device` is a device array of `count`
 * descriptors sorted by first_chunk; a chunk is 4096 elements; first_chunk = running sum of
 * ceil(n / 4096) over the preceding tensors; total_chunks = that sum over all tensors. */
typedef struct zsv_adam_tensor {
    float* p;
    const float* g;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t n;
    int64_t first_chunk;
} zsv_adam_tensor;
int zsv_adam_multi(const zsv_adam_tensor* table_device, int32_t count, int64_t total_chunks, float lr,
                   float beta1, float beta2, float eps, int32_t step, void* stream);

/* ---- loss scaling: torch.cuda.amp.GradScaler on the device (main.py:137,195-203) ------------------ */
/* The reference trains with `scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()`.
 * The state lives in device memory so a step never waits for the host:
 *   scale          current loss scale (GradScaler default 65536)
 *   growth_tracker consecutive steps without a non-finite gradient
 *   found_inf      set by zsv_grad_check_multi, consumed by zsv_adam_multi_scaled, cleared by zsv_scaler_update
 *   steps_done     optimizer steps actually taken (skipped steps do not count): the Adam bias correction uses it */
typedef struct zsv_scaler_state {
    float scale;
    int32_t growth_tracker;
    int32_t found_inf;
    int32_t steps_done;
} zsv_scaler_state;
/* found_inf |= any(!isfinite(g)) over the gradients named by an Adam table (the `g` / `n` / `first_chunk` fields;
 * typically the flat all-reduce buckets of the data-parallel exchange): _amp_foreach_non_finite_check_and_unscale_
 * without the write-back -- the unscale itself happens inside zsv_adam_multi_scaled. */
int zsv_grad_check_multi(const zsv_adam_tensor* table_device, int32_t count, int64_t total_chunks,
                         zsv_scaler_state* state_device, void* stream);
/* zsv_adam_multi with g * (1 / scale) as the gradient, skipped entirely when found_inf is set (scaler.step,
 * main.py:200); step number = steps_done + 1 read on the device. */
int zsv_adam_multi_scaled(const zsv_adam_tensor* table_device, int32_t count, int64_t total_chunks, float lr,
                          float beta1, float beta2, float eps, const zsv_scaler_state* state_device, void* stream);
/* scaler.update() (main.py:203): found_inf ? scale *= backoff, tracker = 0 : (++tracker == interval ? scale *= growth,
 * tracker = 0); steps_done += !found_inf; found_inf = 0.  GradScaler defaults: 2.0, 0.5, 2000. */
int zsv_scaler_update(zsv_scaler_state* state_device, float growth_factor, float backoff_factor,
                      int32_t growth_interval, void* stream);

/* ---- weight panels packed ahead of the call ---------------------------------------------------------------------------
 * Every forward / dgrad entry point above first re-lays its weights out (a "panel": the direct kernel's [block][tap][16][m]
 * image, the Winograd kernels' transformed weights, the stride-2 dgrad's tap-major image) in a small launch of its own: 76
 * launches and 1.1-1.3 ms per R(2+1)D-18 training step (profiles/r03_pack_launches_cost.txt), although the weights change once
 * per step.  A caller that knows when they change can keep the panels: query the panel size, ask for the job that fills it,
 * run ALL jobs of a step in one zsv_pack_multi launch, and call the *_panel forms, which skip the pack launch.
 * `direction` 0 = forward, 1 = input gradient; `extras` != 0 = the forward call carries a bias / ReLU / residual / statistics
 * (some geometries then take another kernel and so another panel).  *panel_bytes = 0: this geometry has no single panel
 * (3-channel stems, class-by-class strided gradients): use the ordinary entry points.  The job holds raw pointers (w, panel):
 * it stays valid while those allocations do.  Panels depend on the ZSV_* switches: re-query after zsv_reload_knobs(). */
typedef struct zsv_pack_job {
    int32_t kind;            /* 0 direct kernel, 1 Winograd F(2,3), 2 Winograd F(4,3), 3 stride-2 input gradient */
    int32_t reserved;
    int64_t total;           /* elements of the panel */
    int64_t first_block;     /* filled by the caller: running sum of ceil(total / 1024) over the preceding jobs of the table */
    const float* w;
    float* out;
    int64_t l[2];            /* kind-specific strides */
    int32_t i[20];           /* kind-specific integers */
} zsv_pack_job;
int zsv_conv3d_panel_query(const zsv_conv_desc* d, int32_t direction, int32_t extras, size_t* panel_bytes);
int zsv_conv3d_panel_job(const zsv_conv_desc* d, int32_t direction, int32_t extras, const float* w, void* panel,
                         size_t panel_bytes, zsv_pack_job* job);
/* one launch for `count` jobs (device array sorted by first_block; total_blocks = sum of ceil(total / 1024)) */
int zsv_pack_multi(const zsv_pack_job* jobs_device, int32_t count, int64_t total_blocks, void* stream);
/* the entry points above with the panel supplied (packed by zsv_pack_multi from this call's job): no pack launch */
int zsv_conv3d_fwd_full_panel(const zsv_conv_desc* d, const float* x, const float* w, const float* bias,
                              const float* residual, float* y, int fuse_relu, float* bn_partials, int32_t stat_tiles,
                              void* workspace, size_t workspace_bytes, void* stream, const void* panel, size_t panel_bytes);
int zsv_conv3d_fwd_pre_panel(const zsv_conv_desc* d, const float* x, const float* pre_coef, int32_t coef_pitch,
                             const float* w, float* y, float* bn_partials, int32_t stat_tiles, void* workspace,
                             size_t workspace_bytes, void* stream, const void* panel, size_t panel_bytes);
int zsv_conv3d_dgrad_add_panel(const zsv_conv_desc* d, const float* dy, const float* w, const float* add, float* dx,
                               void* workspace, size_t workspace_bytes, void* stream, const void* panel, size_t panel_bytes);
int zsv_conv3d_dgrad_add_strided_panel(const zsv_conv_desc* d, const float* dy, const float* w, const float* sub, int32_t st,
                                       int32_t sh, int32_t sw, float* dx, void* workspace, size_t workspace_bytes,
                                       void* stream, const void* panel, size_t panel_bytes);

#ifdef __cplusplus
}
#endif
#endif /* ZSV_HIP_H */
