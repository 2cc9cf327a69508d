/* vvae_hip.h -- C ABI of libvvae_hip.so: the MI355X (gfx950) kernels of the video-VAE hot path.
 *
 * The reference (floatingtrees/video-VAE) has no native code and no FFI: every op below replaces an
 * XLA-lowered Flax/JAX call.  Each entry point cites the reference call site it stands in for
 * (paths relative to the reference checkout).  INTEGRATION.md shows the Python-side binding.
 *
 * Conventions
 *   - plain pointers + sizes only; every buffer is BORROWED device memory owned by the caller;
 *   - activations are channels-last (n,t,h,w,c) seen as rows of `C` channels with row pitch `ld*`
 *     (in elements, >= C), so a channel slice of a wider buffer is a valid operand;
 *   - `dtype`: VVAE_DT_F32 (0) or VVAE_DT_BF16 (1) is the STORAGE type of activations; parameters and
 *     parameter gradients are always fp32 in Flax layout (Conv kernel (kt,kh,kw,Cin,Cout));
 *   - `stream` is a hipStream_t; everything is enqueued asynchronously on it, nothing synchronises;
 *   - return value: 0 on success, a hipError_t value, or VVAE_ERR_* (>= 1000);
 *   - process-global mutable state is limited to test / tuning hooks, none of which the product path calls:
 *     vvae_conv3d_force_generic, vvae_conv3d_roll_config, vvae_conv3d_wgrad_config, vvae_layernorm_config, vvae_layernorm_fwd_mode,
 *     vvae_gemm_tn_use_big_tiles, vvae_gemm_nt_stagger, vvae_gemm_nt_persistent, vvae_gemm_nt_prefetch, vvae_gemm_nt_prefetch_mask, vvae_linear_residual_algo, vvae_temporal_attn_mfma_enable, vvae_temporal_attn_mfma32_enable (each documented at its declaration), plus one cache: vvae_linear_residual_bf16 / vvae_linear_residual_wt_bf16 keep the
 *     hipBLASLt handle and the solution the library's heuristic chose per (shape, pitches) behind a mutex; everything else is a pure function of its arguments.
 */
#ifndef VVAE_HIP_H
#define VVAE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VVAE_DT_F32 0
#define VVAE_DT_BF16 1
#define VVAE_ERR_BAD_ARG 1001
#define VVAE_ERR_WORKSPACE 1002
#define VVAE_ERR_LIBRARY 1003

/* ---- Conv3d, SAME padding, stride 1, odd kernel: nnx.Conv at train/unet.py:13-21 (3x3x3 ConvBlock3D),
 *      :111-113 (3x7x7 patch_mixer), :144-153 (1x1x1 final_conv) and their autodiff (dgrad, wgrad). ---- */
size_t vvae_conv3d_workspace_bytes(int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, int dtype,
                                   int which /* 0 fwd, 1 dgrad, 2 wgrad */);
int vvae_conv3d_fwd(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy,
                    int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, int dtype,
                    void* ws, size_t ws_bytes, void* stream);
int vvae_conv3d_dgrad(const void* dy, int lddy, const float* w, void* dx, int lddx,
                      int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, int dtype,
                      void* ws, size_t ws_bytes, void* stream);
int vvae_conv3d_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw, float* dbias,
                      int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, int dtype,
                      void* ws, size_t ws_bytes, void* stream);
void vvae_conv3d_force_generic(int on);   /* test hook: 1 = bypass the bf16 fast path */
/* the any-shape fp32-matrix-core path, exported so tests can cross-check the fast path against it */
int vvae_conv3d_fwd_generic(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy,
                            int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, int dtype, void* stream);
int vvae_conv3d_dgrad_generic(const void* dy, int lddy, const float* w, void* dx, int lddx,
                              int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, int dtype, void* stream);
/* no atomics: one fp32 partial dW per voxel chunk in ws (vvae_conv3d_wgrad_generic_ws_bytes), folded in index order */
size_t vvae_conv3d_wgrad_generic_ws_bytes(int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw);
int vvae_conv3d_wgrad_generic(const void* x, int ldx, const void* dy, int lddy, float* dw, float* dbias,
                              int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, int dtype,
                              void* ws, size_t ws_bytes, void* stream);
int vvae_conv3d_deep_config(int on);               /* test/tuning hook: deep (K-split waves) rolling fwd/dgrad kernel on/off (off: per-frame kernel) */
int vvae_conv3d_roll_config(int on, int tchunk);   /* test/tuning hook: rolling time-column fwd/dgrad kernel on/off, frames per workgroup */
int vvae_conv3d_wgrad_config(int cob16, int blocks); /* tuning hook: 16 output channels per wgrad workgroup (default 0), persistent grid size (0 = per-config default) */
int vvae_conv3d_bf16_supported(int Cin, int Cout, int kt, int kh, int kw, int ld_in, int ld_out, int which, int flags);
size_t vvae_conv3d_bf16_ws_bytes(int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, int which);
/* flags (pack and forward calls of one layer must agree): bit 0 = input-gradient form; bits 8-15 = how many of the layer's K channels
   (Cin forward, Cout for the input gradient) are real when tensor and weights are zero-padded to 16 (0 = all).  The 3x7x7 patch mixer
   with 12 real channels (train/unet.py:111-113) then multiplies K = 3*7*12 = 252 instead of 336; other values are accepted and ignored. */
int vvae_conv3d_pack_bf16(const float* w, void* ws, size_t ws_bytes, int Cin, int Cout, int kt, int kh, int kw,
                          int flags, void* stream);
/* n <= 64 packings in one launch (every conv layer of the UNet, forward + input-gradient forms: the weights change once per
   optimizer step); host arrays of device pointers / ints, kt = 3, kw = kh in {3, 7}. */
int vvae_conv3d_pack_grouped_bf16(const float* const* w, void* const* ws, const size_t* ws_bytes, const int* Cin, const int* Cout,
                                  const int* kh, const int* flags, int n, void* stream);
int vvae_conv3d_fwd_bf16(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy,
                         int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, int flags,
                         int prepacked, void* ws, size_t ws_bytes, void* stream);
/* Forward Conv3d that also emits the GroupNorm statistics of its (rounded) output -- ConvBlock3D's conv + norm statistics in one
   pass (reference train/unet.py:13-23).  vvae_conv3d_gn_blocks: rows per sample of the partial buffer (0 = layer not eligible,
   use vvae_gn_stats); gn_part: N * blocks * groups * 2 floats, consumed by vvae_gn_finalize. */
int vvae_conv3d_gn_blocks(int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, int ld_in, int ld_out, int groups);
int vvae_conv3d_fwd_bf16_gn(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy,
                            int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw,
                            int prepacked, void* ws, size_t ws_bytes, float* gn_part, int groups, void* stream);
size_t vvae_conv3d_wgrad_bf16_ws_bytes(int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw);
int vvae_conv3d_wgrad_bf16(const void* x, int ldx, const void* dy, int lddy, float* dw, float* dbias,
                           int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw,
                           void* ws, size_t ws_bytes, void* stream);
/* The decoder's concat([up, skip], channels) + conv1 (reference train/unet.py:79-81) at 16 + 16 channels WITHOUT the joint tensor: the
   conv reads its input channels from two dense tensors (x: the first c_split, x2: the rest), its input gradient writes two dense
   tensors (y, y2) and its weight gradient stages X from both.  (A producer that fills a 32-byte channel half of 64-byte voxels
   runs at a third of the HBM rate; see tools/convt_pitch_probe.py.)  which: 0 forward (y2 = NULL; gn_part optional, as in
   vvae_conv3d_fwd_bf16_gn), 1 input gradient (x = dY, x2 = NULL).  ws: packed weights (vvae_conv3d_pack_bf16 / _grouped). */
int vvae_conv3d_cat2_supported(int Cin, int Cout, int c_split, int kt, int kh, int kw);
int vvae_conv3d_fwd_bf16_cat2(const void* x, int ldx, const void* x2, int ldx2, const float* bias, void* y, int ldy, void* y2, int ldy2,
                              int c_split, int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, int which,
                              const void* ws, size_t ws_bytes, float* gn_part, int groups, void* stream);
int vvae_conv3d_wgrad_bf16_cat2(const void* x, int ldx, const void* x2, int ldx2, int c_split, const void* dy, int lddy, float* dw,
                                float* dbias, int N, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw, void* ws,
                                size_t ws_bytes, void* stream);
/* out[c] = sum over V rows of x[v][c] (bias gradients) */
int vvae_colsum_blocks(long V);   /* rows of vvae_colsum's partial buffer */
/* out[c] = sum over V rows of x[v][c]; part: fp32 scratch, vvae_colsum_blocks(V) x C floats; two stages, no atomics: bitwise reproducible */
int vvae_colsum(const void* x, int ld, long V, int C, float* out, float* part, int dtype, void* stream);

/* ---- the rl flavour's latent gate (train/rl_model.py:136-145): pair doubling + Bernoulli frame masks + fill (1 - mask) + z mask, one launch each way.
 *      z fp32 (B2/2, T, per), prob fp32 (B2/2, T), u fp32 (B2, T), fill fp32 (LD) -> comp bf16 (B2, T, per), mask fp32 (B2, T);
 *      bwd: dcomp bf16 -> dz fp32 (B2/2, T, per) and vvae_rl_gate_blocks(...) partial rows of d fill (LD floats each).  per = hw * LD. ---- */
int vvae_rl_gate_ok(int B2, int T, long per, int LD);
int vvae_rl_gate_blocks(int B2, int T, long per);
int vvae_rl_gate_fwd(const float* z, const float* prob, const float* u, const float* fill, void* comp, float* mask, int B2, int T, long per,
                     int LD, void* stream);
int vvae_rl_gate_bwd(const void* dcomp, const float* mask, float* dz, float* part, int B2, int T, long per, int LD, void* stream);

/* ---- zero-pad (unpad = 0) / cut back (unpad = 1) the last two dims of n <= 8 small contiguous fp32 tensors in one launch: the UNet's
 *      12-channel weights on the 16-channel matrix-core kernels (train/unet.py:98-104).  pad: src (rows, s0, s1) -> dst (rows, d0, d1);
 *      unpad: src (rows, d0, d1) -> dst (rows, s0, s1).  Host arrays of device pointers / ints. ---- */
int vvae_pad_last2_grouped(const float* const* src, float* const* dst, const long* rows, const int* s0, const int* s1, const int* d0,
                           const int* d1, int n, int unpad, void* stream);

/* ---- 1x1x1 convolutions onto 3 channels as HBM streams (UNet.final_conv train/unet.py:144-153,188; the PatchUnEmbedding
 *      down-projection train/layers.py:60-79).  Reached through vvae_conv3d_{fwd,dgrad,wgrad}; V = voxels, Cin in {12, 16}. ---- */
int vvae_conv_pointwise_supported(int Cin, int Cout, int kt, int kh, int kw, int ldx, int dtype, const void* x);
size_t vvae_conv_pointwise_ws_bytes(long V, int Cin, int Cout);
int vvae_conv_pointwise_fwd(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy, long V, int Cin, int Cout,
                            int dtype, void* stream);
/* y = addend + conv(x) + bias (addend: V rows of Cout channels, pitch ldadd, or NULL): the decoder's coarse + UNet(features), train/model.py:97 */
int vvae_conv_pointwise_fwd_add(const void* x, int ldx, const float* w, const float* bias, const void* addend, int ldadd, void* y, int ldy,
                                long V, int Cin, int Cout, int dtype, void* stream);
int vvae_conv_pointwise_dgrad(const void* dy, int lddy, const float* w, void* dx, int lddx, long V, int Cin, int Cout, int dtype,
                              void* stream);
/* dx = addend + dy W^T (addend: V rows of Cin channels, pitch ldadd, or NULL): the gradient of the tensor's other consumer joins inside the launch */
int vvae_conv_pointwise_dgrad_add(const void* dy, int lddy, const float* w, const void* addend, int ldadd, void* dx, int lddx, long V, int Cin,
                                  int Cout, int dtype, void* stream);
int vvae_conv_pointwise_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw, float* dbias, long V, int Cin, int Cout,
                              int dtype, void* ws, size_t ws_bytes, void* stream);

/* ---- dst_i (cols_i, rows_i) = src_i (rows_i, cols_i)^T for n <= 64 contiguous bf16 matrices (dims multiples of 64) in one launch: the
 *      (out, in) shadows of the Linear kernels that vvae_gemm_nt_bf16 multiplies in the forward pass, refreshed once per optimizer step. ---- */
int vvae_transpose_grouped_bf16(const void* const* src, void* const* dst, const int* rows, const int* cols, int n, void* stream);

/* The scalar end of the recon + KL loss, value and gradients in one launch (reference train/legacy/training_loop_adversarial.py:100-124:
 * selection density against 1 / max_compression_rate with magnified negatives, MSE + gamma1 selection + gamma2 KL).
 * mse_ps fp32 (B, mse_cols), kl_ps fp32 (B, kl_cols): the per-sample MSE / KL terms as >= 1 partial sums each (summed here in index order: the
 * per-workgroup partials of vvae_masked_mse_mae_fwd, the per-frame partials of vvae_encoder_head_fwd, or one column); selection, mask fp32 (B, T)
 * contiguous.  out fp32 [5] = loss, MSE, selection_loss, kl_loss, mean kept-frame density.  grads fp32 [2 B + B T] = d loss / d (per-sample MSE) |
 * d loss / d (per-sample KL) | d loss / d selection.  B <= 1024. */
int vvae_loss_tail_plain(const float* mse_ps, int mse_cols, const float* kl_ps, int kl_cols, const float* selection, const float* mask, int B, int T,
                         float max_compression_rate, float magnify_negatives_rate, float gamma1, float gamma2, float* out,
                         float* grads, void* stream);

/* The same for the pair / REINFORCE loss of the rl flavour (train/rl_nonadversarial.py:130-186; samples 2k, 2k + 1 are a pair): mse, mae fp32
 * (B2, cols) partial sums, perc fp32 [B2] or NULL, kl fp32 [B2], sel (probabilities) / act (sampled actions) / mask fp32 (B2, T).  out fp32 [9] = loss,
 * MSE, perceptual, selection_loss, kl_loss, mean density, mean trajectory probability, rl_loss, MAE.  grads fp32 [4 B2 + B2 T] = d loss / d mse |
 * mae | perc | kl (per sample) | sel.  kl: (B2, kl_cols) partial sums of the per-sample term (kl_cols = 1: the term itself).  B2 even, <= 1024. */
int vvae_loss_tail_rl(const float* mse, const float* mae, int cols, const float* perc, const float* kl, int kl_cols, const float* sel, const float* act,
                      const float* mask, int B2, int T, float max_compression_rate, float magnify_negatives_rate, float gamma1,
                      float gamma2, float gamma3, float gamma4, float rl_loss_weight, float* out, float* grads, void* stream);

/* ---- The encoder's heads and the latent gate of the model.py flavour in train mode, one launch each way (reference train/model.py:53-59,
 *      121-133; train/layers.py:226-252): log_var = log(softplus(v)), selection logits = Linear(hw -> 1)(Linear(ld -> 1)(mean)) + 1,
 *      sel = round(sigmoid(logits + log(u / (1 - u)))) with the straight-through gradient, z = mean + eps exp(log_var / 2),
 *      comp = fill (1 - sel) + z sel, and the per-frame share of the KL term.  One workgroup per frame; bf16 tensors, fp32 parameters.
 *      mean, v bf16 (B, T, HW, LD) contiguous; w1 (LD), b1 (1), w2 (HW), b2 (1), fill (LD) fp32; u fp32 (B T); eps fp32 (B, T, HW, LD);
 *      mask fp32 rows of T, mask_pitch elements apart per sample.  fwd -> logvar, comp bf16; sel, y (noisy logit) fp32 (B T); s1 fp32 (B T, HW);
 *      kl_frame fp32 (B T).  bwd: dcomp bf16 / dsel fp32 (B T) / gkl fp32 at [b gkl_pitch_b + t gkl_pitch_t] / dlv_ext bf16, each may be NULL
 *      -> dmean, dv bf16; one partial row per frame of part1 (B T, LD) = dW1, part2 (B T, HW) = dW2, part3 (B T, LD) = d fill and
 *      partb (2, B T, 4) = [db1 0 0 0] rows, then [db2 0 0 0] rows.  HW % 4 == 0, LD % 8 == 0; mask_pitch 0 = one mask row for all samples. ---- */
int vvae_encoder_head_ok(int B, int T, int HW, int LD);
/* attribution hook (tests / tools): bit 0 keeps d logits, bit 1 keeps d s1 in fp32 instead of the backward's bf16 rounding points; 0 = shipped */
int vvae_encoder_head_debug(int flags);
int vvae_encoder_head_fwd(const void* mean, const void* v, const float* w1, const float* b1, const float* w2, const float* b2,
                          const float* u, const float* eps, const float* mask, long mask_pitch, const float* fill, void* logvar,
                          void* comp, float* sel, float* y, float* s1, float* kl_frame, int B, int T, int HW, int LD, void* stream);
int vvae_encoder_head_bwd(const void* mean, const void* v, const void* logvar, const float* eps, const float* mask, long mask_pitch, const float* fill,
                          const float* w1, const float* w2, const float* y, const float* s1, const float* sel, const void* dcomp,
                          const float* dsel, const float* gkl, long gkl_pitch_b, long gkl_pitch_t, const void* dlv_ext, void* dmean, void* dv,
                          float* part1, float* part2, float* part3, float* partb, int B, int T, int HW, int LD, void* stream);

/* The rl flavour's counterpart (reference train/rl_model.py:50-60,119-147): the selection is the probability sigmoid(logits); every clip is doubled into a
 * pair (samples 2k, 2k + 1 of the outputs) whose members draw their own Bernoulli frame mask u2 < probability and gate the shared latent with it.  u2 fp32
 * (2B T); logvar2, mean2, comp2 bf16 (2B, T, HW, LD); prob fp32 (2B T), pair-doubled; mask2 fp32 (2B T); kl_frame2 fp32 (2B, T).  bwd: dcomp2 bf16 (2B, T, HW, LD), dprob2 fp32
 * (2B T) (the gradient at the pair-doubled probability), gkl at [i gkl_pitch_b + t gkl_pitch_t] for i < 2B, dlv_ext bf16 (B, T, HW, LD): each may be NULL
 * -> dmean, dv bf16 (B, T, HW, LD) and the partial rows of vvae_encoder_head_bwd. */
int vvae_encoder_head_rl_fwd(const void* mean, const void* v, const float* w1, const float* b1, const float* w2, const float* b2,
                             const float* u2, const float* eps, const float* mask, long mask_pitch, const float* fill, void* logvar2,
                             void* mean2, void* comp2, float* prob, float* mask2, float* y, float* s1, float* kl_frame2, int B, int T, int HW,
                             int LD, void* stream);
int vvae_encoder_head_rl_bwd(const void* mean, const void* v, const void* logvar2, const float* eps, const float* mask, long mask_pitch,
                             const float* fill, const float* w1, const float* w2, const float* y, const float* s1, const float* mask2,
                             const void* dcomp2, const float* dprob2, const float* gkl, long gkl_pitch_b, long gkl_pitch_t, const void* dlv_ext,
                             void* dmean, void* dv, float* part1, float* part2, float* part3, float* partb, int B, int T, int HW, int LD,
                             void* stream);

/* ---- y = silu(x) over a contiguous bf16 tensor of n elements (n % 8 == 0): the activation between the MLP's two Linear layers
 *      (train/layers.py:186-189). ---- */
int vvae_silu_bf16(const void* x, void* y, long n, void* stream);

/* ---- Linear + bias + residual in one library product: y (M,N) = x (M,K) w (K,N) + bias (N) + res (M,N), bf16, fp32 accumulation.
 *      The Linear that closes an attention / MLP branch followed by `x = x + branch` (train/layers.py:212-221, 151, 189): hipBLASLt
 *      reads the residual stream as its C operand (beta = 1, C != D), so the add costs no pass of its own.  bias: NULL / bf16 / fp32
 *      (bias_dtype); res: NULL = plain Linear; ws: device scratch for the library (16-byte aligned; 0 allowed). ---- */
int vvae_linear_residual_bf16(const void* x, int ldx, const void* w, int ldw, const void* bias, int bias_dtype, const void* res, int ldr,
                              void* y, int ldy, int M, int N, int K, void* ws, size_t ws_bytes, void* stream);
/* The same product with the weight given as its (N, K) row-major transpose, pitch ldwt >= K (the optimizer's second bf16 shadow). */
int vvae_linear_residual_wt_bf16(const void* x, int ldx, const void* wt, int ldwt, const void* bias, int bias_dtype, const void* res, int ldr,
                                 void* y, int ldy, int M, int N, int K, void* ws, size_t ws_bytes, void* stream);
/* Test / tuning hook: new plans take the idx-th solution of the library's ranked list (0 = default). */
int vvae_linear_residual_algo(int idx);

/* ---- PatchUnEmbedding's "b t (h w) (p1 p2 cu) -> b t (h p1) (w p2) cu" (train/layers.py:48) fused with the zero padding of the
 *      channel axis from cu to c (the multiple of 16 the conv kernels take), and its transpose.  frames = b*t; bf16; cu, c % 4 == 0. ---- */
int vvae_unpatch_pad_fwd(const void* x, void* y, long frames, int h, int w, int p, int cu, int c, int dtype, void* stream);
int vvae_unpatch_pad_bwd(const void* g, void* gx, long frames, int h, int w, int p, int cu, int c, int dtype, void* stream);

/* ---- GroupNorm(G, eps) + SiLU: nnx.GroupNorm + nnx.silu at train/unet.py:22-23,28-29.
 *      sums: fp64 [N][G][2] (sum, sum of squares) produced by vvae_gn_stats; S = voxels per sample. ---- */
size_t vvae_gn_part_floats(int N, long S, int C);   /* fp32 scratch floats for `part` below (per-workgroup partial sums) */
int vvae_gn_stats(const void* x, int ldx, int N, long S, int C, int G, double* sums, float* part, int dtype, void* stream);
int vvae_gn_finalize(const float* part, int N, int nblk, int G, double* sums, void* stream);   /* second half of vvae_gn_stats for partials written by vvae_conv3d_fwd_bf16_gn */
int vvae_gn_silu_fwd(const void* x, int ldx, void* y, int ldy, const double* sums, const float* gamma, const float* beta,
                     int N, long S, int C, int G, float eps, int dtype, void* stream);
/* silu(GroupNorm(x)) and its (1,2,2) max-pool in one launch (conv2 -> GN -> SiLU -> max_pool of an encoder level, train/unet.py:44-51):
 * y (N,T,H,W,C) pitch ldy, pool (N,T,H/2,W/2,C) pitch ldp; `sums` as for vvae_gn_silu_fwd. */
int vvae_gn_silu_pool_supported(int H, int W, int C, int G, int ldx, int ldy, int ldp, int dtype);
int vvae_gn_silu_pool_fwd(const void* x, int ldx, void* y, int ldy, void* pool, int ldp, const double* sums, const float* gamma,
                          const float* beta, int N, int T, int H, int W, int C, int G, float eps, int dtype, void* stream);
int vvae_gn_silu_bwd(const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx, const double* sums,
                     const float* gamma, const float* beta, double* csum /* fp64 [N][C][2] scratch */, float* part,
                     float* dgamma, float* dbeta, int N, long S, int C, int G, float eps, int dtype, void* stream);

/* ---- max-pool (1,2,2)/(1,2,2): nnx.max_pool at train/unet.py:50.  NT = n*t planes; H, W = input size.
 *      bwd: dx = (dskip ? dskip : 0) + scatter(dpool) to the first arg-max of each window. ---- */
int vvae_maxpool_1x2x2_fwd(const void* x, int ldx, void* y, int ldy, int NT, int H, int W, int C, int dtype, void* stream);
int vvae_maxpool_1x2x2_bwd(const void* x, int ldx, const void* dpool, int lddp, const void* dskip, int ldds,
                           void* dx, int lddx, int NT, int H, int W, int C, int dtype, void* stream);

/* ---- ConvTranspose kernel (1,2,2) strides (1,2,2): nnx.ConvTranspose at train/unet.py:61-69,78.
 *      out[2i+d] = x[i] * K[1-d] per spatial axis (kernel NOT flipped).  H, W = input (low) resolution. ---- */
int vvae_convt_1x2x2_fwd(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy,
                         int NT, int H, int W, int Cin, int Cout, int dtype, void* stream);
int vvae_convt_1x2x2_dgrad(const void* dy, int lddy, const float* w, void* dx, int lddx,
                           int NT, int H, int W, int Cin, int Cout, int dtype, void* stream);
size_t vvae_convt_1x2x2_wgrad_ws_bytes(int NT, int H, int W, int Cin, int Cout);   /* per-chunk fp32 partials, folded in index order (no atomics) */
int vvae_convt_1x2x2_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw,
                           int NT, int H, int W, int Cin, int Cout, int dtype, void* ws, size_t ws_bytes, void* stream);
/* bf16 MFMA path (weights in registers, no LDS) for the UNet decoder shapes 128->64, 64->32, 32->16:
 * dgrad = 0: x -> y (+bias); dgrad = 1: "x" is dy (2H x 2W, Cout), "y" is dx (H x W, Cin).  ws: packed weights. */
/* bf16 matrix-core weight + bias gradient of the same layer (128->64, 64->32, 32->16): deterministic slab reduction. */
int vvae_convt_wgrad_bf16_supported(int Cin, int Cout, int ldx, int lddy);
size_t vvae_convt_wgrad_bf16_ws_bytes(int NT, int H, int W, int Cin, int Cout);
int vvae_convt_1x2x2_wgrad_bf16(const void* x, int ldx, const void* dy, int lddy, float* dw, float* dbias, int NT, int H, int W,
                                int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
int vvae_convt_bf16_supported(int Cin, int Cout, int ld_in, int ld_out);
size_t vvae_convt_bf16_ws_bytes(int Cin, int Cout);
int vvae_convt_1x2x2_bf16(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy,
                          int NT, int H, int W, int Cin, int Cout, int dgrad, void* ws, size_t ws_bytes, void* stream);
/* dgrad bit 8 (0x100): ws already holds that direction's packed weights, written by this call for n <= 16 kernels in one launch
 * (host arrays of device pointers / ints; ws[i] >= vvae_convt_bf16_ws_bytes(Cin[i], Cout[i]) bytes): once per optimizer step. */
int vvae_convt_pack_grouped_bf16(const float* const* w, void* const* ws, const int* Cin, const int* Cout, const int* dgrad, int n, void* stream);

/* ---- temporal attention core: q_norm/k_norm + RoPE + masked softmax(QK^T/sqrt(D))V,
 *      train/layers.py:159-170 (called from FactoredAttention, layers.py:212-213).
 *      qkv (A,T,3*heads*D) pitch ld; mask uint8 (ceil(A/mask_div), T) 1 = attend, or NULL. ---- */
int vvae_temporal_attn_fwd(const void* qkv, int ld, void* out, int ldo, const float* q_scale, const float* k_scale,
                           const float* cos_table, const float* sin_table, const uint8_t* mask, int mask_div,
                           int A, int T, int heads, int D, float eps, int dtype, void* stream);
size_t vvae_temporal_attn_bwd_ws_bytes(int A, int heads, int D);   /* partial rows of the q/k-norm scale gradients (no atomics) */
int vvae_temporal_attn_bwd(const void* qkv, int ld, const void* dout, int lddo, void* dqkv, int lddq,
                           const float* q_scale, const float* k_scale, const float* cos_table, const float* sin_table,
                           const uint8_t* mask, int mask_div, float* dq_scale, float* dk_scale,
                           int A, int T, int heads, int D, float eps, int dtype, void* ws, size_t ws_bytes, void* stream);

/* lane-per-frame form of the same core for head_dim D in {8,16,32,64} (the production path): forward also writes the
 * row log-sum-exp `lse` (A*heads, T) fp32; backward consumes (out, lse) and writes per-workgroup partials of the
 * cos_table / sin_table of the _fast, qk_prep and spatial entry points: fp32 values ALREADY rounded to the activation dtype.
 * q_norm / k_norm scale gradients: dscale_part (vvae_temporal_attn_fast_blocks(...), 2*D) fp32, summed by the caller.
 * inner = 1: sequences contiguous (A,T,C).  inner = hw: tensors are (b,t,hw,C), sequence a = b*hw+i strides over frames
 * (no transpose copies around the temporal half of FactoredAttention). */
int vvae_temporal_attn_fast_supported(int T, int D, int ld, int ldo, int dtype);
int vvae_temporal_attn_fast_blocks(int A, int T, int heads, int D, int dtype);
int vvae_temporal_attn_mfma_enable(int on);   /* test hook: 0 = keep bf16 / T = 16 / head_dim 64 on the VALU kernels (default 1: matrix cores) */
int vvae_temporal_attn_mfma32_enable(int on);   /* test hook: 0 = T = 32 / 64 temporal attention on the VALU kernels again */
int vvae_temporal_attn_fwd_fast(const void* qkv, int ld, void* out, int ldo, float* lse, const float* q_scale,
                                const float* k_scale, const float* cos_table, const float* sin_table, const uint8_t* mask,
                                int mask_div, int inner, int A, int T, int heads, int D, float eps, int dtype, void* stream);
int vvae_temporal_attn_bwd_fast(const void* qkv, int ld, const void* out, int ldo, const void* dout, int lddo, const float* lse,
                                void* dqkv, int lddq, const float* q_scale, const float* k_scale, const float* cos_table,
                                const float* sin_table, const uint8_t* mask, int mask_div, int inner, float* dscale_part,
                                int A, int T, int heads, int D, float eps, int dtype, void* stream);

/* ---- q/k-norm + RoPE prep around a library attention core (spatial half of FactoredAttention, train/layers.py:153-170,217-221).
 *      fwd: qkv (tokens, 3*heads*D) -> out (tokens, 2*heads*D) = [rope(q_norm(q)) | rope(k_norm(k))]; RoPE position = token % S.
 *      bwd: (dq', dk', dv) with element strides (token, head) -> dqkv (tokens, 3*heads*D), all three sections, one launch;
 *           scale-gradient partials part (vvae_qk_prep_blocks(...), 2, D) fp32: [dq_scale | dk_scale], summed by the caller. ---- */
int vvae_qk_prep_supported(int D, int dtype);
int vvae_qk_prep_blocks(long tokens, int heads, int D);
int vvae_qk_prep_fwd(const void* qkv, int ld, void* out, int ldo, const float* q_scale, const float* k_scale,
                     const float* cos_table, const float* sin_table, long tokens, int S, int heads, int D, float eps,
                     int dtype, void* stream);
int vvae_qk_prep_bwd(const void* qkv, int ld, const void* dq, long dq_ts, long dq_hs, const void* dk, long dk_ts, long dk_hs,
                     const void* dv, long dv_ts, long dv_hs, void* dqkv, int lddq, const float* q_scale, const float* k_scale,
                     const float* cos_table, const float* sin_table, float* part, long tokens, int S, int heads, int D,
                     float eps, int dtype, void* stream);

/* ---- fused spatial attention (sequence = h*w patches of a frame, train/layers.py:153-170,217-221): q/k-norm + RoPE +
 *      softmax(QK^T/sqrt(D))V in one kernel per direction; bf16, head_dim 64, S % 32 == 0, S <= 256, no mask.
 *      qkv (A*S, 3*heads*D) row pitch ld; out (A*S, heads*D) row pitch ldo; lse2 fp32 (A*heads, S) (base-2 log-sum-exp). ---- */
int vvae_spatial_attn_supported(int S, int D, int dtype);
int vvae_spatial_attn_fwd(const void* qkv, int ld, void* out, int ldo, float* lse2, const float* q_scale, const float* k_scale,
                          const float* cos_table, const float* sin_table, int A, int S, int heads, int D, float eps, int dtype,
                          void* stream);
/* part: fp32 (A*heads, 2, D) per-(sequence, head) partials of [dq_scale | dk_scale]; dqkv fully written (q, k, v sections). */
int vvae_spatial_attn_bwd(const void* qkv, int ld, const void* out, int ldo, const void* dout, int lddo, const float* lse2,
                          void* dqkv, int lddq, const float* q_scale, const float* k_scale, const float* cos_table,
                          const float* sin_table, float* part, int A, int S, int heads, int D, float eps, int dtype, void* stream);

/* ---- LayerNorm(eps, fast variance, fp32 stats): nnx.LayerNorm at train/layers.py:17,152,155-156,178.
 *      x row r at x + (r / inner) * outer_pitch + (r % inner) * inner_pitch (elements); y, dy, dx contiguous (rows, C).
 *      bwd writes per-workgroup partials part (vvae_layernorm_bwd_blocks(...), 2, C): [sum dy*xhat | sum dy]. ---- */
int vvae_layernorm_supported(int C, int dtype);
int vvae_layernorm_bwd_blocks(long rows, int C, int dtype);
int vvae_delay_us(int us, void* stream);   /* measurement aid: occupy the stream for ~us microseconds (1..1000), see ops.KernelTimer */
int vvae_layernorm_fwd_config(int fwd_cap);   /* tuning hook: workgroups of the bf16 forward kernel, default 1024 */
int vvae_gn_config(int stream_blocks);         /* tuning hook: workgroups of the GroupNorm forward / backward-apply passes, default 4096 */
int vvae_layernorm_config(int bwd_cap);   /* tuning hook: workgroups (= partial rows) of the backward kernel, default 384 (8 waves each) */
int vvae_layernorm_fwd_mode(int late_stage);   /* test hook: 1 = round-1 forward variant (affine parked behind the first rows' loads, raw s_barrier) */
int vvae_layernorm_fwd(const void* x, void* y, const float* gamma, const float* beta, float* mean, float* rstd,
                       const void* addend, void* xsum, long rows, int C, int inner, long outer_pitch, long inner_pitch, float eps,
                       int dtype, void* stream);   /* addend/xsum non-NULL: normalise round(x + addend), write the sum to xsum */
int vvae_layernorm_bwd(const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd, const void* dres,
                       void* dx, float* part, long rows, int C, int inner, long outer_pitch, long inner_pitch, int dtype,
                       void* stream);

/* ---- reparameterise + KL: train/model.py:124-128, train/rl_nonadversarial.py:146-147. ---- */
size_t vvae_loss_part_floats(int B, long M);   /* fp32 scratch floats for `part` below (M = elements per sample): per-workgroup
                                                  partial sums, folded in fixed order -- no float atomics, bitwise reproducible */
int vvae_reparam_kl_fwd(const void* mean, const void* logvar, const float* eps, const float* mask, float* z, float* kl,
                        float* part, int B, int T, long per, int dtype, void* stream);
int vvae_reparam_kl_bwd(const void* mean, const void* logvar, const float* eps, const float* mask, const float* dz,
                        const float* gkl, void* dmean, void* dlogvar, int B, int T, long per, int dtype, void* stream);

/* ---- masked MSE / MAE: train/rl_nonadversarial.py:114-121.  video sample = b / video_div (pair doubling).  mse = mae = NULL: no fold, the
 *      caller reads part = [mse (B, chunks) | mae (B, chunks)], chunks = vvae_loss_part_floats / (2 B). ---- */
int vvae_masked_mse_mae_fwd(const void* video, const void* recon, const float* mask, float* mse, float* mae,
                            float* part, int B, int T, long P, int video_div, int dtype, void* stream);
int vvae_masked_mse_mae_bwd(const void* video, const void* recon, const float* mask, const float* gmse, const float* gmae,
                            void* drecon, int B, int T, long P, int video_div, int dtype, void* stream);

/* ---- weight-gradient GEMM of the dense layers (nnx.Linear under autodiff, train/layers.py:15,142-151,179-189):
 *      C[M][N] fp32 = sum_k A[k][m] * B[k][n], db[n] = sum_k B[k][n]; A (K,M), B (K,N) bf16 token-major. ---- */
int vvae_gemm_tn_supported(int M, int N, int K, int lda, int ldb);
size_t vvae_gemm_tn_ws_bytes(int M, int N, int K);
int vvae_gemm_tn_bf16(const void* A, int lda, const void* B, int ldb, float* C, float* db, int M, int N, int K,
                      void* ws, size_t ws_bytes, void* stream);
/* Grouped form: n <= 64 weight gradients that share K in one launch, one whole-K 256x256 tile per workgroup (no split-K slabs,
 * no reduction pass): C_i (M_i, N_i) fp32 contiguous = A_i^T B_i, db_i (N_i) or NULL; M_i, N_i % 256 == 0, K % 32 == 0.  The arrays
 * are host arrays of device pointers / ints. */
int vvae_gemm_tn_grouped_bf16(const void* const* A, const int* lda, const void* const* B, const int* ldb, float* const* C,
                              float* const* db, const int* M, const int* N, int n, int K, void* stream);
int vvae_gemm_tn_use_big_tiles(int on);   /* test hook: 0 = 128x128 kernel for every shape */

/* ---- optimiser: optax.chain(clip_by_global_norm, adam) at train/rl_nonadversarial.py:248-251. ---- */
/*      The global norm is reduced without atomics: vvae_sqnorm_partials writes vvae_sqnorm_blocks(n) fp64 partial sums of squares,
 *      every workgroup of vvae_adam_clip_step folds them in one fixed order (so every rank of a data-parallel job, holding the
 *      same all-reduced gradient, applies the bitwise-same clip factor) and the total lands in *gnorm_sq_out. */
int vvae_sqnorm_blocks(long n);
int vvae_sqnorm_partials(const float* g, long n, double* part, void* stream);
int vvae_adam_clip_step(float* p, const float* g, float* m, float* v, void* p_bf16, long n, const double* gnorm_part, int nparts,
                        double* gnorm_sq_out, float gscale, float max_norm, float lr, float b1, float b2, float eps, long count,
                        void* stream);
int vvae_cast_f32_to_bf16(const float* x, void* y, long n, void* stream);
/* ---- dense-layer GEMM, both operands K-contiguous: C (M,N) bf16 = epi(A (M,K) . B (N,K)^T + bias), fp32 accumulation.
 *      Linear forward (B = transposed bf16 weight shadow) and input gradient (B = the weight itself) of nnx.Linear at
 *      train/layers.py:15,142-151,179-189.  epi 0 none; 1 + res (residual add); 2 SiLU, rounded pre-activation -> C2;
 *      3 * silu'(res) (res = saved pre-activation).  The elementwise tail acts on the bf16-rounded linear output. ---- */
int vvae_gemm_nt_supported(int M, int N, int K, int lda, int ldb, int ldc);
int vvae_gemm_nt_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias, const void* res,
                      int ldr, void* C2, int ldc2, int epi, int M, int N, int K, void* stream);
/* The same product on the second kernel form (csrc/gemm_pp.hip: all LDS-DMA staging issued from the read phases -- the token operand through a
 * ring of three slots by waves 4-7, the weight operand through two by waves 0-3 -- so that the MFMA phases are bare MFMAs; one stream of k-tiles over
 * the tiles a workgroup walks; a per-wave epilogue under the partner wave's MFMA phase).  Arguments and results as vvae_gemm_nt_bf16 (bit for
 * bit); additionally K >= 128 and N <= 1536 (N a multiple of 192) or 2048 (of 128): the bias vector of the launch is staged in LDS. */
int vvae_gemm_pp_supported(int M, int N, int K, int lda, int ldb, int ldc);
int vvae_gemm_pp_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias, const void* res,
                      int ldr, void* C2, int ldc2, int epi, int M, int N, int K, void* stream);
/* Timing-only hook of the -DPP_ABLATION build of vvae_gemm_pp_bf16 (tools/pp_ablation.py): bit 0 no DMA, bit 1 no fragment reads, bit 2 no MFMAs
 * in the plain product's main loop (wrong results); the shipped library ignores it. */
int vvae_gemm_pp_ablate(int bits);
/* Tuning hook of vvae_gemm_pp_bf16: 1 (default) = the last epilogue of a launch goes through the idle operand rings, all units of a wave at once; 0 = unit by unit. */
int vvae_gemm_pp_final_ring(int on);
/* Test / tuning hook: start-time stagger between the two workgroup cohorts of vvae_gemm_nt_bf16, in units of 2048 cycles (default 0 = off: a gain on
 * back-to-back copies of one product, none inside the train step). */
int vvae_gemm_nt_stagger(int units);
/* Test / tuning hook: 0 = one tile per workgroup, 1 = persistent workgroups over tiles b, b + 256, ... (default). */
int vvae_gemm_nt_persistent(int on);
/* Test / tuning hook: L2 prefetch distance (k-tiles of 64) of the token panel in vvae_gemm_nt_bf16 (default 3, 0 = off). */
int vvae_gemm_nt_prefetch(int dist);
/* Test / tuning hook: bit e set = epilogue kind e of vvae_gemm_nt_bf16 prefetches (default 5: plain and SiLU pair). */
int vvae_gemm_nt_prefetch_mask(int mask);

/* out[c] = sum_r part[r][c] in fixed order: folds the per-workgroup partial rows of the backward kernels (cols % 4 == 0). */
int vvae_sum_rows(const float* part, int rows, int cols, float* out, void* stream);
/* n <= 64 partial buffers folded in one launch: part[i] fp32 (rows[i], cols[i]); columns [0, n0[i]) -> d0[i], the rest -> d1[i] (or NULL). */
int vvae_fold_rows_grouped(const void* const* part, float* const* d0, float* const* d1, const int* rows, const int* cols,
                           const int* n0, int n, void* stream);
/* n <= 64 contiguous fp32 ranges copied in one launch (gradients landing in the optimizer's flat buffer; replaces the
   per-parameter copies of optax's tree_map update path, reference train/rl_nonadversarial.py:233-236) */
int vvae_copy_grouped(const float* const* src, float* const* dst, const long* count, int n, void* stream);

/* Test hook: y[i] = x[i ^ o] within each group of 64 floats (o in {1,2,4,8,16,32}, n % 64 == 0), yd likewise in fp64 (scaled by
   1.000000001): the VALU-only lane exchange (DPP + v_permlane16/32_swap) every wave-level reduction of this library uses. */
int vvae_selftest_xor_lane(const float* x, float* y, double* yd, int n, int o, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VVAE_HIP_H */
