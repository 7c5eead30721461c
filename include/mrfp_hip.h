/*
 * mrfp_hip.h -- C ABI of libmrfp_hip.so: the MI355X (gfx950) kernels behind the MRFP+ training
 * hot path (reference airl-iisc/MRFP: deepv3.py:152-367 and its callees).
 *
 * The reference has no native code and no FFI: its hot path is torch.nn modules calling
 * ATen/cuDNN.  Each entry point below names the reference call site (file:line) whose
 * arithmetic it replaces.  The Python side (mrfp_amd/ops.py) binds these with ctypes; see
 * INTEGRATION.md for the binding stub and the nn.Module drop-in surface.
 *
 * Conventions
 *  - plain pointers and sizes only; no torch types.  All tensor pointers are DEVICE pointers.
 *  - activations are NHWC ("channels-last"): x[b][h][w][c], dense, dtype MRFP_F32, MRFP_BF16 or MRFP_F16.
 *    Statistics, coefficients, losses and weights' master copies are always fp32.
 *  - the caller allocates every output and every workspace; nothing is allocated, freed or
 *    synchronised inside; every launch goes to `stream` (a hipStream_t passed as void*).
 *  - return 0 on success, negative on error; the message is mrfp_last_error() (thread-local).
 *  - "geometry" arguments shared by the row kernels:
 *      B, Ho, Wo, C  : logical (destination) tensor [B,Ho,Wo,C]
 *      Hs, Ws        : source tensor [B,Hs,Ws,C] when the op reads through a nearest-neighbour
 *                      resize (HRFP, reference deepv3.py:320-327); Hs==Ho, Ws==Wo otherwise
 *      tabH, tabW    : int32 device tables, tabH[oh] = source row of destination row oh
 *                      (NULL = identity).  Built on the host with ATen's float32 rule.
 */
#ifndef MRFP_HIP_H
#define MRFP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { MRFP_F32 = 0, MRFP_BF16 = 1, MRFP_F16 = 2 } mrfp_dtype;

int mrfp_version(void);
const char* mrfp_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Per-channel statistics over NHWC rows.  Replaces the reductions inside F.batch_norm
 * (reference mynn.py:19-25 Norm2d), nn.InstanceNorm2d (reference Resnet.py:176-178, 534-536),
 * feat.mean((2,3)) of NP+ (reference deepv3.py:269) and AdaptiveAvgPool2d(1) (deepv3.py:109).
 *
 * Workspace layout: float ws[B][nslab][2][C] with nslab = mrfp_stats_nslab(B, Ho);
 * ws[b][s][0][c] = partial sum, ws[b][s][1][c] = partial sum of squares (fwd), or
 * partial sum of dy' and of dy'*(x-mean) (bwd).
 * ------------------------------------------------------------------------------------------- */
int64_t mrfp_stats_nslab(int64_t B, int64_t Ho);

int mrfp_stats_fwd(const void* x, int dtype, int64_t B, int64_t Ho, int64_t Wo, int64_t C,
                   int64_t Hs, int64_t Ws, const int32_t* tabH, const int32_t* tabW,
                   float* ws, void* stream);

/* ReLU mask of the backward pass: dy' = dy * (y > 0) when y != NULL (mask from the forward output);
 * else dy' = dy * (x*fA + fS > 0) when fA/fS != NULL (mask recomputed from x with the forward apply
 * coefficients -- one tensor less to read); else dy' = dy.
 * mean, fA, fS: float [G][C] (G = B if per_image else 1); mean NULL = 0. */
int mrfp_stats_bwd(const void* dy, const void* x, const void* y, const float* mean,
                   const float* fA, const float* fS, int per_image, int dtype, int64_t B, int64_t Ho, int64_t Wo, int64_t C,
                   int64_t Hs, int64_t Ws, const int32_t* tabH, const int32_t* tabW,
                   float* ws, void* stream);
/* The same with the ReLU gate read from a SIGN MASK (1 bit per element, bit i of byte e/8 = element e of the dense [B,H,W,C]
 * tensor, written by mrfp_affine_fwd_relu_mask) instead of y: the backward of a residual block's BatchNorm -> add -> ReLU tail
 * (reference Resnet.py:202-225) needs only the sign of its output.  16-bit activations, C % 8 == 0, identity geometry. */
int mrfp_stats_bwd_mask(const void* dy, const void* x, const void* mask, const float* mean, int per_image, int dtype, int64_t B,
                        int64_t H, int64_t W, int64_t C, float* ws, void* stream);

/* BatchNorm (train): statistics over B*Ho*Wo; biased var for normalisation, unbiased for the
 * running update (momentum).  Outputs mean[C], invstd[C] and the apply coefficients
 * A[c] = w*invstd, S[c] = b - mean*A.  running_* may be NULL. */
int mrfp_bn_finalize(const float* ws, int64_t B, int64_t nslab, int64_t count, int64_t C,
                     const float* weight, const float* bias, float eps, float momentum,
                     float* running_mean, float* running_var,
                     float* mean, float* invstd, float* A, float* S, void* stream);
/* BatchNorm (eval): coefficients from the running statistics. */
int mrfp_bn_eval_coef(int64_t C, const float* weight, const float* bias, const float* running_mean,
                      const float* running_var, float eps, float* A, float* S, void* stream);
/* backward: dweight[C], dbias[C] and the input-gradient coefficients
 * dx = P*dy' + Q*x + R  (per channel). */
int mrfp_bn_bwd_finalize(const float* ws, int64_t B, int64_t nslab, int64_t count, int64_t C,
                         const float* weight, const float* mean, const float* invstd,
                         float* dweight, float* dbias, float* P, float* Q, float* R, void* stream);

/* InstanceNorm2d(affine): statistics per (b,c) over Ho*Wo, biased variance.  weight/bias may be
 * NULL (affine=False, reference instance_whitening.py:5-16).  Outputs are [B][C]. */
int mrfp_in_finalize(const float* ws, int64_t B, int64_t nslab, int64_t count, int64_t C,
                     const float* weight, const float* bias, float eps,
                     float* mean, float* invstd, float* A, float* S, void* stream);
/* dweight/dbias are accumulated over b inside (written, not added). */
int mrfp_in_bwd_finalize(const float* ws, int64_t B, int64_t nslab, int64_t count, int64_t C,
                         const float* weight, const float* mean, const float* invstd,
                         float* dweight, float* dbias, float* P, float* Q, float* R, void* stream);

/* NP+ (reference deepv3.py:268-277).  alpha, beta_noise: the two normal draws, float [B][C].
 * fwd: mu[B][C] = plane means; sigma[C] = unbiased std over the batch; scale = 1.5*sigma/max(sigma);
 *      y = A*x + S with A = alpha, S = (1 + beta_noise*scale - alpha)*mu.
 * bwd: from G[b][c] = sum_hw dy (ws from mrfp_stats_bwd with y=NULL, mean=NULL) produces the
 *      per-plane additive constant K so that dx = alpha*dy + K  (gradients flow through mu,
 *      sigma and the max exactly as autograd does for the reference expression). */
int mrfp_np_finalize(const float* ws, int64_t B, int64_t nslab, int64_t count, int64_t C,
                     const float* alpha, const float* beta_noise,
                     float* mu, float* sigma, float* A, float* S, void* stream);
int mrfp_np_bwd_finalize(const float* ws, int64_t B, int64_t nslab, int64_t count, int64_t C,
                         const float* alpha, const float* beta_noise, const float* mu,
                         const float* sigma, float* Gtmp /* scratch [B][C] */, float* K, void* stream);

/* Plane means only (AdaptiveAvgPool2d(1), reference deepv3.py:109,117): out[B][C] (dtype);
 * tmp: fp32 scratch [B][C].  Its backward (broadcast of g/(H*W)) is mrfp_affine_fwd with x = NULL. */
int mrfp_mean_finalize(const float* ws, int64_t B, int64_t nslab, int64_t count, int64_t C,
                       float* tmp, void* out, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Apply kernels.
 * fwd:  y[b,oh,ow,c] = act( x[b,tabH[oh],tabW[ow],c] * A[g,c] + S[g,c] + res[b,oh,ow,c] )
 *       g = b if coef_per_image else 0;  res may be NULL;  relu = 0/1.
 *       A == NULL means A = 1 (pure broadcast add of S), x == NULL means x = 0.
 * bwd:  dy' = dy * (y > 0) if y != NULL, else dy * (x*fA + fS > 0) if fA != NULL (see mrfp_stats_bwd)
 *       dx[b,ih,iw,c] = sum over destinations mapped to (ih,iw) of P*dy'  + n*(Q*x + R)
 *       (n = number of such destinations; invH/invW give their half-open ranges:
 *        destination rows [invH[2*ih], invH[2*ih+1]) map to source row ih; NULL = identity)
 *       dres (optional, same geometry as dy) receives dy'.
 * ------------------------------------------------------------------------------------------- */
int mrfp_affine_fwd(const void* x, const void* res, void* y, int dtype,
                    int64_t B, int64_t Ho, int64_t Wo, int64_t C, int64_t Hs, int64_t Ws,
                    const int32_t* tabH, const int32_t* tabW,
                    const float* A, const float* S, int coef_per_image, int relu, void* stream);

/* mrfp_affine_fwd (identity geometry) that also writes the per-workgroup partial sums of its STORED output, float ws[B][nslab][2][C]
 * with nslab = mrfp_stats_nslab(B, H) -- bit for bit the rows mrfp_stats_fwd(y) would produce -- for a per-image normalisation of y that
 * follows at once: the InstanceNorm behind a residual tail (reference Resnet.py:218-225) or NP+ behind an InstanceNorm
 * (deepv3.py:333-335): feed ws to mrfp_in_finalize / mrfp_np_finalize and skip their statistics pass. */
int mrfp_affine_fwd_stats(const void* x, const void* res, void* y, int dtype, int64_t B, int64_t H, int64_t W, int64_t C,
                          const float* A, const float* S, int coef_per_image, int relu, float* ws, void* stream);
int mrfp_affine_bwd(const void* dy, const void* x, const void* y, void* dx, void* dres, int dtype,
                    int64_t B, int64_t Ho, int64_t Wo, int64_t C, int64_t Hs, int64_t Ws,
                    const int32_t* invH, const int32_t* invW,
                    const float* P, const float* Q, const float* R,
                    const float* fA, const float* fS, int coef_per_image, void* stream);
/* y = relu(x*A + S + res) AND its sign mask (see mrfp_stats_bwd_mask); the matching backward apply reads the mask where
 * mrfp_affine_bwd reads y (dres = the masked gradient, for the skip connection). */
int mrfp_affine_fwd_relu_mask(const void* x, const void* res, void* y, void* mask, int dtype, int64_t B, int64_t H, int64_t W,
                              int64_t C, const float* A, const float* S, int coef_per_image, void* stream);
int mrfp_affine_bwd_mask(const void* dy, const void* x, const void* mask, void* dx, void* dres, int dtype, int64_t B, int64_t H,
                         int64_t W, int64_t C, const float* P, const float* Q, const float* R, int coef_per_image, void* stream);

/* dst[p][c0_dst + c] = src[p][c0_src + c], c < C, over npix pixels of NHWC tensors with channel pitches ld_src / ld_dst:
 * torch.cat((...), 1) of reference deepv3.py:125, 353 (one call per input) and its backward (one call per slice). */
int mrfp_copy_channels(const void* src, void* dst, int dtype, int64_t npix, int64_t C, int64_t ld_src, int64_t c0_src,
                       int64_t ld_dst, int64_t c0_dst, void* stream);

/* y = a + b, elementwise over n elements (torch.add of reference deepv3.py:330, 357). */
int mrfp_add(const void* a, const void* b, void* y, int dtype, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Bilinear resize, align_corners=True (reference mynn.py:114-119 Upsample).
 * fwd: y[B,Ho,Wo,C] = bilinear(x[B,Hi,Wi,0:C]) (+ addend[B,Ho,Wo,C] if not NULL; deepv3.py:356-357)
 * bwd: dx[B,Hi,Wi,0:C] = adjoint applied to dy[B,Ho,Wo,C]  (gather form, no atomics)
 * ld_in: channel pitch of the low-resolution tensor (>= C; the 19-class logits are kept in a
 *        32-channel padded buffer at low resolution so that the final 1x1 conv stays chunk-aligned).
 * ------------------------------------------------------------------------------------------- */
int mrfp_bilinear_fwd(const void* x, const void* addend, void* y, int dtype,
                      int64_t B, int64_t Hi, int64_t Wi, int64_t Ho, int64_t Wo, int64_t C, int64_t ld_in,
                      void* stream);
int mrfp_bilinear_bwd(const void* dy, void* dx, int dtype,
                      int64_t B, int64_t Hi, int64_t Wi, int64_t Ho, int64_t Wo, int64_t C, int64_t ld_in,
                      void* stream);
/* The same with the output (forward) / the incoming gradient (backward) being a block of C channels inside a wider NHWC tensor
 * with ld_out / ld_dy channels per pixel (y / dy point at the block's first channel): Upsample() writing straight into its
 * slot of a torch.cat(dim=1) buffer and reading its slice of that buffer's gradient (reference deepv3.py:349-353). */
int mrfp_bilinear_fwd_into(const void* x, void* y, int dtype, int64_t B, int64_t Hi, int64_t Wi, int64_t Ho, int64_t Wo,
                           int64_t C, int64_t ld_in, int64_t ld_out, void* stream);
int mrfp_bilinear_bwd_from(const void* dy, void* dx, int dtype, int64_t B, int64_t Hi, int64_t Wi, int64_t Ho, int64_t Wo,
                           int64_t C, int64_t ld_in, int64_t ld_dy, void* stream);

/* ---------------------------------------------------------------------------------------------
 * MaxPool2d(kernel 3, stride 2, padding 1) (reference Resnet.py:551, deepv3.py:315).
 * y[B,Ho,Wo,C], Ho = (H-1)/2+1; idx[B,Ho,Wo,C] (uint8) stores the window position 0..8 of the
 * first maximum.  bwd (gather form): dx[B,H,W,C] = sum of dy over windows whose arg-max is this pixel.
 * ------------------------------------------------------------------------------------------- */
int mrfp_maxpool_fwd(const void* x, void* y, uint8_t* idx, int dtype,
                     int64_t B, int64_t H, int64_t W, int64_t C, void* stream);
int mrfp_maxpool_bwd(const void* dy, const uint8_t* idx, void* dx, int dtype,
                     int64_t B, int64_t H, int64_t W, int64_t C, void* stream);

/* The stem's  norm -> ReLU -> maxpool  (reference Resnet.py:549-551, 413-421; deepv3.py:309-315) without the normalised tensor in
 * memory.  maxpool_affine_fwd pools relu(x*A + S) (A, S[C] or [B,C] as mrfp_affine_fwd; every window value is rounded to the
 * activation type before the comparison, so y and idx are those of mrfp_affine_fwd followed by mrfp_maxpool_fwd).  The two
 * backward passes of the normalisation take the POOLED gradient dy[B,Ho,Wo,C] + idx and x (the normalisation's input):
 * pool_norm_bwd_stats writes the partial sums mrfp_stats_bwd would write for the un-pooled gradient (ws: [B][mrfp_stats_nslab(B,H)]
 * [2][C] floats; ReLU gate (x*fA + fS) > 0), pool_norm_bwd_apply writes dx = P*d + Q*x + R as mrfp_affine_bwd does. */
int mrfp_maxpool_affine_fwd(const void* x, const float* A, const float* S, int coef_per_image, int relu, void* y, uint8_t* idx,
                            int dtype, int64_t B, int64_t H, int64_t W, int64_t C, void* stream);
int mrfp_pool_norm_bwd_stats(const void* dy, const uint8_t* idx, const void* x, const float* mean, const float* fA,
                             const float* fS, int coef_per_image, int relu, float* ws, int dtype,
                             int64_t B, int64_t H, int64_t W, int64_t C, void* stream);
int mrfp_pool_norm_bwd_apply(const void* dy, const uint8_t* idx, const void* x, const float* P, const float* Q, const float* R,
                             const float* fA, const float* fS, int coef_per_image, int relu, void* dx, int dtype,
                             int64_t B, int64_t H, int64_t W, int64_t C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * CrossEntropyLoss(ignore_index=255), mean over valid pixels (reference main.py:822,
 * deepv3.py:363).  logits [B,H,W,C] NHWC (dtype), target int64 [B,H,W].
 * ws: float [2*nblk] with nblk = mrfp_ce_nblocks(B*H*W).
 * fwd writes loss[0] = mean NLL, loss[1] = number of valid pixels (fp32).
 * bwd: dlogits = (softmax - onehot) * gscale[0] / nvalid  for valid pixels, 0 otherwise.
 * ------------------------------------------------------------------------------------------- */
int64_t mrfp_ce_nblocks(int64_t npix);
int mrfp_ce_fwd(const void* logits, const int64_t* target, int dtype, int64_t npix, int64_t C,
                int64_t ignore_index, float* ws, float* loss, void* stream);
int mrfp_ce_bwd(const void* logits, const int64_t* target, const float* loss, const float* gscale,
                void* dlogits, int dtype, int64_t npix, int64_t C, int64_t ignore_index, void* stream);

/* Fused bilinear upsample (align_corners) of the channel-padded low-resolution class scores P[B,Hi,Wi,ld]
 * + cross entropy at [B,H,W] (reference deepv3.py:361-365: in training only the scalar loss is needed, so the
 * full-resolution logits are never written).  ws / loss as mrfp_ce_fwd (npix = B*H*W).
 * bwd writes d(logits)[B,H,W,Cd] (Cd = C rounded up to a 16-byte chunk, pad channels zero) for mrfp_bilinear_bwd. */
int mrfp_upsample_ce_fwd(const void* P, int64_t ld, const int64_t* target, int dtype, int64_t B, int64_t Hi, int64_t Wi,
                         int64_t H, int64_t W, int64_t C, int64_t ignore_index, float* ws, float* loss, void* stream);
int mrfp_upsample_ce_bwd(const void* P, int64_t ld, const int64_t* target, const float* loss, const float* gscale,
                         void* dlogits, int64_t Cd, int dtype, int64_t B, int64_t Hi, int64_t Wi, int64_t H, int64_t W,
                         int64_t C, int64_t ignore_index, void* stream);

/* Eval: argmax over classes + 19x19 confusion histogram on the device (reference main.py:898-909,
 * metrics.py:122-126): hist[num_classes*gt + pred] += 1 for gt in [0,num_classes).  hist: int64
 * [C*C], accumulated (not cleared).  pred (optional, uint8 [npix]) receives the arg-max. */
int mrfp_argmax_hist(const void* logits, const int64_t* target, int dtype, int64_t npix, int64_t C,
                     int64_t* hist, uint8_t* pred, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Convolution on the matrix cores (implicit GEMM, NHWC).  Replaces every nn.Conv2d of the hot
 * path (reference Resnet.py:156-161, deepv3.py:96-112, 200-237) and its autograd backward.
 *
 * mrfp_pack_weight: OIHW fp32 master [N][C][R][S] -> forward pack wf[Npad][R][S][Cpad] and
 *   dgrad pack wd[C][R][S][Npad] (taps flipped), both in `dtype`, pad entries zero.
 * mrfp_conv_fwd:  y[B,Ho,Wo,0:N] (pitch ldy) = conv(x[B,H,W,C], wpack) + bias
 *   output position o reads source position o*stride - pad + r*dil; with sstride > 1 the tap
 *   exists only where that position is a multiple of sstride (then index = position / sstride):
 *   this is the dgrad of a strided convolution, run on dy with the wd pack.
 *   addend (optional, same shape / pitch / dtype as y) is added in the epilogue: the gradient of a skip
 *   connection is accumulated by the dgrad launch itself instead of a separate elementwise pass.
 *   C*sizeof(dtype) must be a multiple of 16 (pad the channels).
 * mrfp_conv_wgrad: dw[N][Ctrue][R][S] (fp32, OIHW) = sum over pixels of dy[.,n] * x[tap(.),c];
 *   ws: mrfp_conv_wgrad_ws_bytes(M = B*Ho*Wo, N, Q = R*S*C) bytes of scratch (split-K slabs,
 *   summed in a fixed order: bitwise reproducible).
 * mrfp_nchw_to_nhwc_pad: network input NCHW fp32 -> NHWC `dtype` with zero pad channels.
 * ------------------------------------------------------------------------------------------- */
int mrfp_pack_weight(const float* w, void* wf, void* wd, int dtype, int64_t N, int64_t C, int64_t R, int64_t S,
                     int64_t Npad, int64_t Cpad, void* stream);
/* All packs of a model in one launch (after the optimizer step rewrote the fp32 masters).  jobs: device array of
 *   struct { const float* w; void* wf; void* wd; int32_t N, C, R, S, Npad, Cpad; }   (48 bytes, both packs required)
 * prefix: device int64[njobs + 1], exclusive prefix sum of the jobs' brick counts ceil(Npad/64) * ceil(Cpad/bc), bc = 8 (64 for pointwise filters)
 * (one workgroup packs a brick of 64 output x 8 input channels x R x S -- 64 x 64 for pointwise filters -- through LDS); total = prefix[njobs].  Every job must have R*S <= 9. */
int mrfp_pack_weights_batched(const void* jobs, const int64_t* prefix, int64_t njobs, int64_t total, int dtype, void* stream);
int mrfp_conv_fwd(const void* x, const void* wpack, const float* bias, void* y, int dtype,
                  int64_t B, int64_t H, int64_t W, int64_t C, int64_t N, int64_t ldy, int64_t R, int64_t S,
                  int64_t Ho, int64_t Wo, int64_t stride, int64_t pad_h, int64_t pad_w, int64_t dil,
                  int64_t sstride, const void* addend, float* colstats, void* stream);
/* colstats (optional): the epilogue also writes per-channel partial sums of the STORED output,
 * float [nblk][2][ldy] (sum, sum of squares per row block), nblk = mrfp_conv_stats_blocks(...): the
 * BatchNorm statistics pass over the conv output disappears (feed it to mrfp_bn_finalize with B = 1,
 * nslab = nblk, count = B*Ho*Wo). */
int64_t mrfp_conv_stats_blocks(int dtype, int64_t B, int64_t H, int64_t W, int64_t C, int64_t N, int64_t R, int64_t S, int64_t Ho,
                               int64_t Wo, int64_t stride, int64_t pad_h, int64_t pad_w, int64_t dil, int64_t sstride);
/* rows of the [M][N] output one of those row blocks covers (block r = rows [r * rb, (r + 1) * rb)): when rb divides Ho*Wo no block
 * straddles an image and the blocks of image b are the per-image partial sums nn.InstanceNorm2d needs (reference Resnet.py:176-178,
 * 534-536: InstanceNorm after the stem convolutions) -- feed them to mrfp_in_finalize with nslab = Ho*Wo / rb. */
int64_t mrfp_conv_stats_block_rows(int dtype, int64_t B, int64_t H, int64_t W, int64_t C, int64_t N, int64_t R, int64_t S, int64_t Ho,
                                   int64_t Wo, int64_t stride, int64_t pad_h, int64_t pad_w, int64_t dil, int64_t sstride);
/* takes the geometry arguments of the mrfp_conv_fwd call it describes: the kernel the launch runs on (tile shape, the pointwise
 * kernels, the row-reuse 3x3 kernels) -- and with it the number of row blocks -- depends on all of them. */
/* The K-loop gathers through 32-bit buffer-descriptor offsets, so one launch reads at most 3.75 GB of input; a larger
 * activation (BASELINE.json configs[4] at 16 images per GPU: 16 x 256 x 512 x 1024 bf16 = 4.3 GB) is walked in batch ranges
 * by mrfp_conv_fwd / mrfp_conv_wgrad themselves (one image must stay below the limit).  Returns 1 when B images of
 * image_bytes = H*W*C*sizeof(dtype) run as ONE launch -- only then are the fused per-row-block statistics available. */
int mrfp_conv_single_launch(int64_t B, int64_t image_bytes);
/* colstats must hold mrfp_conv_stats_rows(nblk) rows of 2*ldy floats; large launches fold their row blocks into
 * 64 groups appended behind them: hand rows [final_first, final_first + final_count) to mrfp_bn_finalize. */
int64_t mrfp_conv_stats_rows(int64_t nblk);
int64_t mrfp_conv_stats_final_first(int64_t nblk);
int64_t mrfp_conv_stats_final_count(int64_t nblk);
/* mrfp_conv_fwd whose fused statistics count output row m (pixel b, oh, ow) `rowweight[m]` times: the statistics of the
 * nearest-neighbour RESIZED output (rowweight = the pixel's multiplicity in the resize, 0..255) -- the HRFP stages
 * conv -> F.interpolate(nearest) -> BatchNorm(train) (reference deepv3.py:320-327) without a statistics pass over the
 * resized tensor.  rowweight: 4-byte aligned, B*Ho*Wo bytes + 256 readable bytes of padding.  Not for pointwise convolutions. */
int mrfp_conv_fwd_wstats(const void* x, const void* wpack, const float* bias, void* y, int dtype,
                         int64_t B, int64_t H, int64_t W, int64_t C, int64_t N, int64_t ldy, int64_t R, int64_t S,
                         int64_t Ho, int64_t Wo, int64_t stride, int64_t pad_h, int64_t pad_w, int64_t dil,
                         const uint8_t* rowweight, float* colstats, void* stream);
/* mrfp_conv_fwd with a GATED addend: y = conv(x) + (addend where its gate bit is set, else 0).  addend_mask holds one bit per
 * element of the dense [M][N] addend (bit e & 7 of byte e >> 3, e = m*N + n) -- the sign mask mrfp_affine_fwd_relu_mask wrote for
 * the residual tail whose incoming gradient `addend` is: the skip-connection gradient dy * [y > 0] (reference: autograd of
 * `out += residual; out = relu(out)`, Resnet.py:214-225) is then never materialised.  16-bit activations, N % 8 == 0, ldy == N. */
int mrfp_conv_fwd_gated(const void* x, const void* wpack, const float* bias, void* y, int dtype,
                        int64_t B, int64_t H, int64_t W, int64_t C, int64_t N, int64_t ldy, int64_t R, int64_t S,
                        int64_t Ho, int64_t Wo, int64_t stride, int64_t pad_h, int64_t pad_w, int64_t dil,
                        int64_t sstride, const void* addend, const void* addend_mask, void* stream);
/* Diagnostic (no reference counterpart): in a library built with -DMRFP_CLOCK_STAMP=1 (tools/clock_stamp.py; never the product
 * build, where this returns -1) every convolution workgroup records d(s_memtime) and d(s_memrealtime) around its main loop; out
 * receives n pairs {shader cycles, 100 MHz ticks} of the LAST launch of `family` (0: conv_igemm, 1: pointwise, 2: wgrad).
 * In-kernel clock = cycles / ticks x 100 MHz (MI355X_MICROARCH.md, DVFS give-back).
 * family 3 (round 6, tools/phase_stamp.py): n pairs {cycles, samples}, eight per (workgroup, wave 0 / wave 4) of the two-group long-K
 * pointwise kernel -- the cycles that wave spent in each phase of its K loop (csrc/conv_pwk.hip: PhaseClock). */
int mrfp_debug_clock_stamps(int family, uint64_t* out, int64_t n);
int64_t mrfp_conv_wgrad_ws_bytes(int64_t M, int64_t N, int64_t Q);
int mrfp_conv_wgrad(const void* x, const void* dy, float* dw, void* ws, int dtype,
                    int64_t B, int64_t H, int64_t W, int64_t C, int64_t Ctrue, int64_t N, int64_t ldn,
                    int64_t R, int64_t S, int64_t Ho, int64_t Wo, int64_t stride, int64_t pad_h, int64_t pad_w,
                    int64_t dil, void* stream);
/* GROUPED weight gradients: `count` (<= mrfp_conv_wgrad_group_max()) problems of ONE geometry -- the repeated blocks of a ResNet
 * stage (reference network/Resnet.py:579-585 _make_layer builds blocks-1 identical Bottlenecks; autograd of Resnet.py:202-216
 * produces their weight gradients one by one) -- in ONE launch + ONE slab reduction: count x the output tiles fill the chip with
 * 2-3 K' splits per problem instead of 21-32.  xs / dys / dws: HOST arrays of `count` device pointers (x[g], dy[g] as in
 * mrfp_conv_wgrad; dws[g] = that problem's OIHW fp32 gradient); ws: mrfp_conv_wgrad_grouped_ws_bytes(M, N, Q, count) bytes.
 * Every problem's result equals a mrfp_conv_wgrad_grouped call of the same count (fixed summation order: bitwise reproducible);
 * it differs from the single launch's only in the association of the K' splits.  Each activation must fit one 3.75 GB range. */
int64_t mrfp_conv_wgrad_group_max(void);
int64_t mrfp_conv_wgrad_grouped_ws_bytes(int64_t M, int64_t N, int64_t Q, int64_t count);
int mrfp_conv_wgrad_grouped(const void* const* xs, const void* const* dys, float* const* dws, int64_t count, void* ws, int dtype,
                            int64_t B, int64_t H, int64_t W, int64_t C, int64_t Ctrue, int64_t N, int64_t ldn,
                            int64_t R, int64_t S, int64_t Ho, int64_t Wo, int64_t stride, int64_t pad_h, int64_t pad_w,
                            int64_t dil, void* stream);
int mrfp_nchw_to_nhwc_pad(const float* x, void* y, int dtype, int64_t B, int64_t C, int64_t H, int64_t W,
                          int64_t Cpad, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused SGD(momentum, weight decay) over a flat fp32 arena (reference main.py:826-839, 863-864):
 *   g' = g*gscale + wd*p ; m = g' (first) | momentum*m + g' ; p -= lr*m.   n % 4 == 0.
 * ------------------------------------------------------------------------------------------- */
int mrfp_sgd_step(float* p, const float* g, float* m, int64_t n, float lr, float momentum,
                  float weight_decay, float gscale, int first, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Group whitening passes (groups of 16 channels; reference network/sync_switchwhiten.py:20-26, 161-170:
 * in_data.mean(-1), bmm(in_data, in_data^T) per group; :217 bmm(wm, in_data); network/instance_whitening.py):
 *   mrfp_group_moments: M[b,g,i,j] = sum_p a[b,p,16g+i] * b[b,p,16g+j]  (fp32 [B,C/16,16,16]) and, when sum_a != NULL,
 *     sum_a[b,c] = sum_p a[b,p,c] -- one read of a and of b (a == b: second moments of the forward pass; a = dy,
 *     b = x: gradient of the whitening matrix and of the shift).  ws: mrfp_group_moments_ws_bytes() bytes of scratch
 *     (per-workgroup fp32 partials, combined in fp64 in a fixed order).
 *   mrfp_group_apply:   y[b,p,16g+i] = sum_j Wm[b,g,i,j] x[b,p,16g+j] (+ sum_j Vm[b,g,i,j] z[b,p,16g+j]) + shift[b,16g+i]
 *     (z, Vm both NULL or both given; shift optional) -- forward: the folded whitening matrix and offset; backward:
 *     dx = Wm^T dy + (dM + dM^T) x + dmean/HW in one pass.
 * a, b, x, z, y: [B,HW,C] activations (dtype); C % 16 == 0, C <= 1024.
 * ------------------------------------------------------------------------------------------- */
int64_t mrfp_group_moments_ws_bytes(int64_t B, int64_t HW, int64_t C);
int mrfp_group_moments(const void* a, const void* b, float* M, float* sum_a, void* ws, int dtype, int64_t B, int64_t HW,
                       int64_t C, void* stream);
int mrfp_group_apply(const void* x, const float* Wm, const void* z, const float* Vm, const float* shift, void* y, int dtype,
                     int64_t B, int64_t HW, int64_t C, void* stream);
/* Inverse square root of n 16x16 covariance matrices by T <= 8 Newton-Schulz steps (reference
 * sync_switchwhiten.py:206-215: P <- 1.5 P - 0.5 P^3 (S/tr S), wm = P sqrt(1/tr S)), and its backward (recomputes the
 * chain; dcov from dwm).  cov, wm, dwm, dcov: fp32 [n,16,16]. */
int mrfp_group_isqrt_fwd(const float* cov, float* wm, int64_t n, int T, void* stream);
int mrfp_group_isqrt_bwd(const float* cov, const float* dwm, float* dcov, int64_t n, int T, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fourier amplitude perturbation (north_star extension; no function of this kind exists in the
 * reference's model -- nearest arithmetic: dataloaders.py:24-79; semantics are build-defined):
 *   F = rfft2(x[b,:,:,c]); ratio = band ? ((1-lam)|F| + lam|F_partner|)/|F| : 1; y = irfft2(F*ratio)
 *   band = (min(kh,H-kh)^2 + kw^2 <= radius^2), complemented when high != 0; partner = perm[b].
 * Ws = mrfp_fourier_stored_bins(H, W, radius, high) is the number of bins along W the scratch spectra hold: W/2+1 in
 * general; floor(radius)+1 for a LOW band on the register path, where the ratio differs from 1 only in the columns
 * kw <= radius, so only those columns are transformed and y = x + irfft2(F*(ratio-1)) (same arithmetic, 2.5x less
 * HBM traffic).  S, S3: scratch spectra, float2 [B,H,Ws,C] each (mrfp_fourier_spectrum_bytes() is the upper bound);
 * ratio (optional, float [B,H,Ws,C]) is written (load_ratio = 0) or read (load_ratio = 1: the backward pass applies
 * the same detached ratio to the gradient; pass the forward call's radius and high, they fix Ws).
 * twH / twW: float2 tables exp(-2 pi i t/N), t < N, for N = H and N = W.
 * H, W of the form 2^a 3^b, <= 512, W even; C % 16 == 0.
 * ------------------------------------------------------------------------------------------- */
int64_t mrfp_fourier_spectrum_bytes(int64_t B, int64_t H, int64_t W, int64_t C);
int64_t mrfp_fourier_stored_bins(int64_t H, int64_t W, float radius, int high);
int mrfp_fourier_mix(const void* x, void* y, const int64_t* perm, void* S, void* S3, float* ratio, int load_ratio,
                     const void* twH, const void* twW, int dtype, int64_t B, int64_t H, int64_t W, int64_t C,
                     float radius, float lam, int high, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The training input pipeline (transform_tr), bit-exact with the PIL calls of the reference (main.py:409-419
 * transform_tr: dataloaders.py:139-150 flip, 398-435 RandomSizeAndCrop = img.resize(BICUBIC) / mask.resize(NEAREST),
 * 257-337 RandomCrop with ImageOps.expand padding, 118-136 ToTensor).  Pillow resizes 8-bit images in two separable
 * fixed-point passes: out = clip8((2^21 + sum_t in[lo + t] * coefs[o][t]) >> 22), horizontal first.
 *   mrfp_resample_u8: one pass over a uint8 [Hin,Win,C] image; bounds [out][2] = (lo, count), coefs [out][ksize] int32
 *     (host-built as Pillow builds them, mrfp_amd/input_pipeline.py); vertical = 0: Hout == Hin, flip != 0 reads the
 *     source mirrored (the reference flips before it scales); vertical = 1: Wout == Win.
 *   mrfp_input_assemble: pad (image 0 / label `ignore`) + crop + ToTensor.  img: scaled uint8 [Hs,Ws,3]; lab: ORIGINAL
 *     uint8 label map [Hl,Wl]; ytab [Hs], xtab [Ws]: Pillow's nearest-neighbour source indices of the scaled label;
 *     the scaled image sits at (pad_x, pad_y) of the padded one, the crop starts at (x1, y1); out_img float [3,Hc,Wc]
 *     (values 0..255, no /255, as dataloaders.py:128-133), out_lab int64 [Hc,Wc]; with out_u8 != NULL the image goes
 *     there as uint8 [Hc,Wc,3] instead (the blur passes run before ToTensor = mrfp_u8hwc_to_f32chw).
 *   mrfp_box_blur3_u8: one pass of ImageFilter.GaussianBlur(radius < 1) (dataloaders.py:168-177; Pillow BoxBlur.c):
 *     out = (in*ww + (left + right)*fw + 2^23) >> 24, edges replicated; GaussianBlur = 3 horizontal + 3 vertical passes;
 *     ww, fw from the radius as Pillow derives them (mrfp_amd/input_pipeline.py::_blur_weights).
 * ------------------------------------------------------------------------------------------- */
int mrfp_resample_u8(const void* src, void* dst, int64_t Hin, int64_t Win, int64_t Hout, int64_t Wout, int64_t C,
                     const int32_t* bounds, const int32_t* coefs, int ksize, int vertical, int flip, void* stream);
int mrfp_input_assemble(const void* img, const void* lab, const int32_t* ytab, const int32_t* xtab, int64_t Hs, int64_t Ws,
                        int64_t Hl, int64_t Wl, int flip, int pad_x, int pad_y, int x1, int y1, int64_t Hc, int64_t Wc, int ignore,
                        float* out_img, void* out_u8, int64_t* out_lab, void* stream);
int mrfp_box_blur3_u8(const void* src, void* dst, int64_t H, int64_t W, int64_t C, int64_t ww, int64_t fw, int vertical,
                      void* stream);
int mrfp_u8hwc_to_f32chw(const void* src, float* dst, int64_t H, int64_t W, void* stream);
/* One ColorJitter step on a uint8 [npix,3] image (dataloaders.py:491-660; Pillow ImageEnhance = Image.blend with a degenerate
 * image, Convert.c rgb2hsv / hsv2rgb): op 0 brightness, 1 contrast (needs ws: 16 bytes, the L mean is reduced on the device),
 * 2 saturation, 3 hue (shift = uint8(hue_factor * 255), factor unused).  Byte-exact with PIL. */
int mrfp_jitter_u8(const void* src, void* dst, int64_t npix, int op, float factor, int shift, void* ws, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Whitening-loss setup (host-side, no kernel; csrc/hostmath.hip).
 *   mrfp_kmeans1d: globally optimal k-means of n doubles into k clusters -- what `kmeans1d.cluster(var_flatten,
 *     self.clusters)` does at reference network/cov_settings.py:57 (kmeans1d is an un-vendored PyPI dependency; its published
 *     dynamic programme over the sorted values is restated).  labels [n] int32 in input order, clusters numbered by
 *     ascending centroid; centroids [k] (k is clamped to n).  Host pointers.
 * ------------------------------------------------------------------------------------------- */
int mrfp_kmeans1d(const double* x, int64_t n, int k, int32_t* labels, double* centroids);

#ifdef __cplusplus
}
#endif
#endif /* MRFP_HIP_H */
