/* C ABI of libdua_hip.so -- the MI355X (gfx950) kernels behind the Diff-UNet hot path.
 *
 * The reference (aarchiiive/diff-unet-amos) is pure Python: its "FFI" for this path is the set
 * of torch.nn / torch elementwise calls listed below.  Each entry point names the reference
 * call site it replaces (file:line relative to the reference tree).  All entry points:
 *   - take raw device pointers, sizes and a hipStream_t (as void*); no torch types;
 *   - enqueue work on that stream and return immediately (no allocation, no synchronisation),
 *     so a caller may capture them into a hipGraph;
 *   - return 0 on success, a positive hipError_t, or DUA_ERR_ARG for a rejected argument.
 *
 * Activation layout everywhere: channels-last [N][D][H][W][Cstride]; a call addresses channels
 * [C_off, C_off + C) of the buffer.  dtype selects the element type of activations and packed
 * weights: DUA_F16 (fp16 operands, fp32 accumulate on MFMA 32x32x16) or DUA_F32 (exact fp32 on
 * MFMA 32x32x2).  Statistics, scale/shift vectors, biases and sampler state are always fp32.
 */
#ifndef DUA_HIP_H
#define DUA_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define DUA_F32 0
#define DUA_F16 1
#define DUA_ERR_ARG (-22)

/* ---- fused producer normalisation --------------------------------------------------------------
 * A convolution stores its RAW output (conv + bias) and accumulates per-(n, c) sums of it into
 * out_stats = dua_stat_word [N][8][4][c_pad] (8 replica rows x 4 words x channels; the caller zeroes it before the
 * producing launch).  The words of a channel are (sum x: integer part, sum x: fraction * 2^44, sum x^2: integer part,
 * sum x^2: fraction * 2^44), added with 64-bit INTEGER atomics: integer addition is associative, so the sums -- and every
 * scale / shift derived from them -- do not depend on the order in which workgroups arrive (two launches on the same
 * inputs agree bit for bit).  Whoever consumes the raw tensor passes this descriptor and applies
 *   y = LeakyReLU(x * scale + shift) + add,  scale = gamma / sqrt(var + eps), shift = beta - mean * scale
 * (biased variance) while staging its input: InstanceNorm3d(affine) -> Dropout(0) -> LeakyReLU of MONAI's
 * ADN (models/basic_unet/denoiser.py:206-207, models/diff_unet.py:34-35) and the temb add of
 * TwoConv.forward (denoiser.py:65).  stats == NULL means "input is already materialised". */
typedef long long dua_stat_word;
typedef struct {
  const dua_stat_word* stats; /* producer's sums, [N][8][4][c_pad] */
  const float* gamma;        /* producer's InstanceNorm weight [C] */
  const float* beta;         /* producer's InstanceNorm bias [C] */
  const float* add;          /* optional fp32 [N][add_stride] bias added after the activation, or NULL */
  int add_stride;            /* 0 = C */
  int c_pad;                 /* channel stride of stats */
  long long count;           /* D*H*W of the producer's output: voxels per (sample, channel), as an INTEGER -- the kernels divide
                                the sums by it in double.  (Up to ABI 7 this was float 1/count: float(1/884736) is 3e-8 off, which
                                is 0.3 % of the variance of a channel whose mean is 100 standard deviations.)  Must be > 0. */
  float eps;                 /* 1e-5 */
  float slope;               /* LeakyReLU negative slope, 0 <= slope <= 1 (the fp16 kernels apply it as max(t, slope t)) */
} dua_in_norm;

/* ---- 3x3x3 convolution --------------------------------------------------------------------
 * Replaces nn.Conv3d(k3,s1,p1,bias) built by MONAI Convolution at
 * models/basic_unet/denoiser.py:56-59 and models/basic_unet/pretrained/basic_unet.py:60-63. */
typedef struct {
  int dtype;
  int N, D, H, W;
  int Cin, Cin_stride, Cin_off;    /* all multiples of 8 */
  int Cout, Cout_stride, Cout_off; /* all multiples of 8 */
  int tap_channel_plus1;           /* 0 = ordinary convolution.  k > 0 (DUA_F16, Cin <= 32, no fused
                                      producer): packed input channel k-1 (0 or 16) is the LAST real channel (Cin = k-1 + 8,
                                      channels behind it are zero padding); its 27 taps are contracted as two 16-wide k-steps
                                      instead of 27 padded ones.  Needs weights from dua_pack_conv3_weights_tap. */
  int background;                  /* 1 = this launch runs on a second stream UNDER a chain of small
                                      launches that the caller is waiting for: it asks for LDS it does not use, so that one of
                                      its workgroups fits a CU instead of two and the chain's workgroups always find free
                                      registers and LDS.  Same result, longer launch.  0 = ordinary launch. */
  int layout;                      /* bit 0 (DUA_IN_BLOCKED): x is stored in 16-channel blocks, bit 1 (DUA_OUT_BLOCKED): y is.
                                      A blocked buffer of Cstride channels (a multiple of 16) holds sample n as
                                      [Cstride / 16][voxels][16] instead of [voxels][Cstride]; channel offsets must be multiples
                                      of 16.  It exists for the wide-tile convolution (csrc/conv3d_wide.hip), which walks its
                                      input 16 channels at a time: from channels-last rows every pass touches every 128-byte
                                      voxel line again (2.9x the algorithmic bytes fetched, 32 cache lines per wave request).
                                      Only the kernels named by dua_conv3d_k3_kernel_kind / dua_deconv_k2s2_kernel_kind below
                                      read or write it; any other combination is rejected with DUA_ERR_ARG.  0 = channels-last. */
  int policy;                      /* 0 = the launcher's automatic choice of kernel form.  Non-zero values select a form by hand --
                                      for the kernel tests (every form is exercised on shapes the automatic choice would not
                                      give it) and same-process A/B timing; results are the same.  dua_conv3d_k3_fwd: low byte
                                      2 = 4x8x8 tiles without split-K, 3 = 2x8x8 tiles (slab form), 6 = automatic without the
                                      kd-plane / LDS-DMA form of the small layers, 7 = automatic without the wide-tile form, 8 / 9 = the wide-tile
                                      form with persistent workgroups (9: staggered start; both measured slower, kept for A/B);
                                      bit 8 (DUA_POLICY_NO_FINISH) = skip the split-K finish kernel (timing the main kernel alone:
                                      outputs are then NOT valid).  dua_deconv_k2s2_fwd: 6 = the one-tap-per-workgroup kernel
                                      for every shape that has it, 256-voxel all-taps tiles.  dua_conv3d_k3_wgrad: bit 0 = plain
                                      block order, bits 1-4 = workgroups per CU over the launch, bit 5 / 6 = the 6-wave forms,
                                      bit 7 = tiles through registers, bit 8 = the first form (a workgroup per kd plane of taps)
                                      instead of the fetch-once form (fp16).  A per-call field: the library keeps no mutable option
                                      state.  Any other value is rejected with DUA_ERR_ARG. */
} dua_conv3_desc;
#define DUA_POLICY_NO_FINISH 256
#define DUA_IN_BLOCKED 1
#define DUA_OUT_BLOCKED 2

/* Which kernel a launch described by d takes (the launchers' own policy, exported so that a caller that lays buffers out in
 * 16-channel blocks decides with the rule the launcher uses).  fused = the call will pass a producer descriptor.
 * dua_conv3d_k3_kernel_kind: 0 = conv3d_k3_v2_kernel (channels-last only), 1 = conv3d_k3_first_kernel (output may be
 * blocked), 2 = conv3d_k3_wide_kernel (input and output may be blocked).  dua_deconv_k2s2_kernel_kind: 0 = a kernel that
 * writes channels-last only, 2 = deconv_k2s2_alltaps_kernel (output may be blocked). */
int dua_conv3d_k3_kernel_kind(const dua_conv3_desc* d, int fused, int has_workspace);
int dua_deconv_k2s2_kernel_kind(const dua_conv3_desc* d);

/* w_packed: from dua_pack_conv3_weights.  bias_padded: fp32, at least Cout entries (a buffer padded to
 * ceil(Cout/64)*64 works, entries behind Cout are never read).  in: NULL or the
 * producer descriptor of x.  y: raw output.  out_stats: dua_stat_word [N][8][4][ceil(Cout/64)*64], pre-zeroed.
 * workspace (may be NULL): scratch for split-K on layers too small to fill 256 CUs (<= 24^3): the K range
 * (Cin chunk x kd) is divided over workgroups, fp32 partial tiles land in the workspace and a finish kernel
 * sums them, adds the bias and takes the statistics.  dua_conv3d_k3_workspace gives the bytes that enables it. */
long dua_conv3d_k3_workspace(const dua_conv3_desc* d);
int dua_conv3d_k3_fwd(const dua_conv3_desc* d, const void* x, const void* w_packed, const float* bias_padded,
                      const dua_in_norm* in, void* y, dua_stat_word* out_stats, void* workspace, long workspace_bytes,
                      void* stream);

/* ---- UpCat's first convolution with the transposed convolution folded in ------------------------------------
 * Replaces, in ONE launch, UpCat.forward up to the first convolution (models/basic_unet/denoiser.py:172-194, 56-59):
 *     x_0 = ConvTranspose3d(k2, s2)(act(norm(u)))      (MONAI UpSample "deconv", denoiser.py:161-170)
 *     y   = Conv3d(k3, p1)(cat([x_e, x_0], 1))          (TwoConv.conv_0.conv)
 * Nothing non-linear sits between the two layers, so the upsampled half of the convolution is a 2x2x2-parent contraction
 * per output parity with composed weights W'[phi][delta] = sum over taps of Wc_up[tap] Wd[child] (8 parents x Cu channels
 * instead of 27 taps x Cmid channels: 3.4x fewer multiply-adds on that half, and the transposed convolution's launch, its
 * output write and the re-read of it are gone); the transposed convolution's bias becomes one of 27 per-border-class bias
 * vectors (taps that fall outside the volume see no bias).  Same function of the same parameters; the composed weights are
 * formed in fp32 and rounded once to fp16.  D, H, W: OUTPUT extents (multiples of 8); u is the coarse tensor
 * [N][D/2][H/2][W/2][Cu_stride] (channels-last; u_in = the producer descriptor of a RAW convolution output, or NULL when u is already
 * an activation);
 * xskip holds x_e on the fine grid (channels-last or 16-channel blocks).  DUA_F16 only; Cskip % 16 == 0, Cu % 64 == 0,
 * Cu <= 256.  dua_upconv_k3_supported: 1 when the descriptor can be launched (callers fall back to
 * dua_deconv_k2s2_fwd + dua_conv3d_k3_fwd otherwise). */
typedef struct {
  int dtype;
  int N, D, H, W;                       /* output extents */
  int Cskip, Cskip_stride, Cskip_off;   /* x_e: channels [Cskip_off, Cskip_off + Cskip) of the fine-grid buffer */
  int Cu, Cu_stride, Cu_off;            /* u: channels of the coarse buffer */
  int Cout, Cout_stride, Cout_off;
  int layout;                           /* DUA_IN_BLOCKED: xskip in 16-channel blocks; DUA_OUT_BLOCKED: y in 16-channel blocks */
} dua_upconv_desc;
int dua_upconv_k3_supported(const dua_upconv_desc* d);
/* wc: Conv3d weight fp32 [Cout][Cskip + Cmid][3][3][3]; up_first = 0: input channels [0, Cskip) are x_e and [Cskip, Cskip + Cmid)
 * the upsampled half (torch.cat([x_e, x_0]), models/basic_unet/denoiser.py:190); up_first = 1: the upsampled half comes first
 * (torch.cat((out, skip)), MONAI UnetrUpBlock as used by models/swin_unetr/denoiser.py:388-397).  bc: its bias [Cout] or NULL.
 * wd: ConvTranspose3d weight fp32 [Cu][Cmid][2][2][2] (rows of zero-padding channels of u: zeros), bd its bias [Cmid] or NULL.
 * Writes the composed weights (fp16, the order the kernel streams them) and bias_table fp32 [27][ceil(Cout/64)*64] (class =
 * (cd*3 + ch)*3 + cw, 0 = low border, 1 = interior, 2 = high border).  Returns the bytes of wu_packed (query with wu_packed ==
 * NULL).  The skip half's weights are packed by dua_pack_conv3_weights(dtype, Cout, Cskip + Cmid, Cskip, wc, in_perm, ...) with
 * in_perm = NULL (up_first = 0) or the channels [Cmid, Cmid + Cskip) (up_first = 1). */
long dua_pack_upconv_weights(int dtype, int Cout, int Cskip, int Cmid, int Cu, int up_first, const float* wc, const float* bc,
                             const float* wd, const float* bd, void* wu_packed, float* bias_table, void* stream);
int dua_upconv_k3_fwd(const dua_upconv_desc* d, const void* xskip, const void* u, const dua_in_norm* u_in, const void* w_skip_packed,
                      const void* wu_packed, const float* bias_table, void* y, dua_stat_word* out_stats, void* stream);

/* Weight gradient of the same convolution (backward of train.py:258-268 through denoiser.py:56-59):
 *   dw[co][ci][kd][kh][kw] += sum over (n, voxel) of dy[n, v, co] * x[n, v + tap - 1, ci]
 * d describes the FORWARD convolution (x: Cin channels at Cin_off of a Cin_stride buffer; dy: Cout channels at
 * Cout_off of a Cout_stride buffer, same N/D/H/W).  dw: fp32 [Cout][Cin_src][27] in the reference's layout,
 * ACCUMULATED into with fp32 atomics (zero it first).  in_perm (device int[ceil(Cin/64)*64]) or NULL maps a packed
 * input channel of x to its source channel in dw (negative = padding), as in dua_pack_conv3_weights.
 * workspace (dua_conv3d_k3_wgrad_workspace bytes; may be NULL): per-workgroup partial sums, reduced by a second
 * kernel -- without it every workgroup adds into dw with fp32 atomics (correct, much slower). */
long dua_conv3d_k3_wgrad_workspace(const dua_conv3_desc* d);
/* Data gradient of the same convolution: dx = dua_conv3d_k3_fwd(dy, W') with W'[ci][co][tap] = W[co][ci][26-tap].
 * This packs W' straight from the forward weights w (fp32 [Cout][Cin][27]) for a dy buffer of Cout_packed (>= Cout)
 * channels; same return convention as dua_pack_conv3_weights (bytes; query with w_packed == NULL). */
long dua_pack_conv3_weights_dgrad(int dtype, int Cout, int Cin, int Cout_packed, const float* w, void* w_packed,
                                  void* stream);
int dua_conv3d_k3_wgrad(const dua_conv3_desc* d, const void* x, const void* dy, float* dw, int Cin_src,
                        const int* in_perm, void* workspace, long workspace_bytes, void* stream);

/* Backward of a = LeakyReLU(InstanceNorm3d_affine(y)) [+ add] [+ emb] (denoiser.py:63-67,206-207 under
 * train.py:258-268).  y (raw) and its forward statistics (in->stats) are what the forward kept.
 *   reduce: sums[N][8][in->c_pad][4] (fp64, pre-zeroed) += (sum dA, sum dZ, sum dZ*zhat, unused) per replica row;
 *           d add[n,c] = sum over replicas of [0]; d beta[c] = sum_n [1]; d gamma[c] = sum_n [2]
 *   apply : dY = gamma*rstd*(dZ - S1/V - zhat*S2/V), written to channels [out_off, out_off+C) of dY's buffer. */
typedef struct {
  int dtype;
  int N;
  long voxels;                 /* D*H*W */
  int C;                       /* multiple of 8, <= 1024 */
  int da_stride, da_off;
  int raw_stride, raw_off;
  int out_stride, out_off;     /* apply only */
} dua_norm_bwd_desc;
int dua_instnorm_bwd_reduce(const dua_norm_bwd_desc* d, const void* dA, const void* raw, const dua_in_norm* in,
                            double* sums, void* stream);
/* dgamma, dbeta (fp32 [C], ZEROED by the caller, accumulated over the samples) and dadd (fp32 [N][C], written) are the
 * parameter gradients of the layer -- d gamma, d beta of InstanceNorm3d(affine) and the gradient of the timestep-embedding
 * add -- emitted by the same launch; each may be NULL. */
int dua_instnorm_bwd_apply(const dua_norm_bwd_desc* d, const void* dA, const void* raw, const dua_in_norm* in,
                           const double* sums, void* dY, float* dgamma, float* dbeta, float* dadd, void* stream);

/* Training: the data gradient of a Conv3d (dua_conv3d_k3_fwd on dy with the weights flipped and transposed -- d, dy, w_packed,
 * bias_padded as there; dx = its output) whose input was a = LeakyReLU(InstanceNorm3d(raw)) of ANOTHER layer: dx is that layer's
 * dA, and the launch adds the three sums of ITS InstanceNorm backward (dua_instnorm_bwd_reduce: sum dA, sum dZ, sum dZ * zhat per
 * (n, c), on the rounded dA it stores) to `sums` (zeroed by the caller) instead of accumulating statistics of dx -- one pass
 * over dA and raw less per layer pair (train.py:258-268 through denoiser.py:56-67).  raw: that layer's convolution output,
 * channels [raw_off, raw_off + Cout) of a raw_stride buffer; raw_in: its forward statistics, gamma, beta, count, eps, slope.
 * Only the wide-tile form has this epilogue: dua_conv3d_k3_dgrad_reduce_supported(d) == 1 (fp16, channels-last, >= 1024 tiles,
 * extents multiples of 8), else DUA_ERR_ARG -- the caller keeps dua_conv3d_k3_fwd + dua_instnorm_bwd_reduce. */
int dua_conv3d_k3_dgrad_reduce_supported(const dua_conv3_desc* d);
int dua_conv3d_k3_dgrad_reduce(const dua_conv3_desc* d, const void* dy, const void* w_packed, const float* bias_padded, void* dx,
                               const void* raw, int raw_stride, int raw_off, const dua_in_norm* raw_in, double* sums, void* stream);

/* The 1x1x1 residual branch of MONAI's UnetResBlock over torch.cat((ConvTranspose3d_k2s2(lo), skip), 1) -- the decoder blocks of
 * models/swin_unetr/denoiser.py:388-397 (UnetrUpBlock: transp_conv, cat, UnetResBlock whose conv3 is the 1x1x1 branch; no bias in
 * either layer) -- as ONE launch:  res[o] = (W3_up Wd[child(o)]^T) lo[parent(o)] + W3_skip skip[o],  plus this layer's InstanceNorm
 * sums.  d describes the transposed convolution as dua_deconv_k2s2_fwd does (N/D/H/W and Cin* of lo; Cout* of res on the 2D x 2H x 2W
 * grid), with w_packed = dua_pack_deconv_weights of the COMPOSED weights [Cin][Cout][8]; xskip = Cs channels at Cs_off of a
 * Cs_stride buffer on the fine grid; ws_packed = dua_pack_deconv_weights(Cs, Cout) of W3_skip^T repeated over the 8 taps (tap 0
 * is read).  DUA_F16, channels-last, Cin <= 128, Cs <= 128.  dua_deconv_k2s2_res_supported: 1 when it can be launched. */
int dua_deconv_k2s2_res_supported(const dua_conv3_desc* d, int Cs);
int dua_deconv_k2s2_res_fwd(const dua_conv3_desc* d, const void* lo, const void* w_packed, const void* xskip, int Cs, int Cs_stride,
                            int Cs_off, const void* ws_packed, void* y, dua_stat_word* stats, void* stream);

/* Backward of nn.MaxPool3d(2) (denoiser.py:100,106) fused with the sum of x_l's two gradient paths:
 *   out[v] = dA[v] (or 0 when dA is NULL) + (v is the arg-max of its 2x2x2 window ? dP[window] : 0),
 * arg-max recomputed from the stored activation (first maximum in d,h,w scan order, as torch).  D, H, W = the
 * un-pooled extent (even); act/dA are channel slices [off, off+C) of strided buffers, dP/out start at channel 0. */
int dua_maxpool2_bwd_add(int dtype, int N, int D, int H, int W, int C, const void* act, int act_stride, int act_off,
                         const void* dA, int da_stride, int da_off, const void* dP, int dp_stride, void* out,
                         int out_stride, void* stream);

/* Backward of ConvTranspose3d(k2, s2) (denoiser.py:161-170) for the training step.  d describes the FORWARD op (as
 * dua_deconv_k2s2_fwd): x = Cin channels at Cin_off of a Cin_stride buffer, N/D/H/W its extent; dy = Cout channels at
 * Cout_off of a Cout_stride buffer with extent 2D x 2H x 2W (e.g. the gradient of the concat buffer, read in place).
 *   dx (or NULL): data gradient, written to the same slice geometry as x; needs w_packed_dgrad from
 *                 dua_pack_deconv_weights_dgrad (w = fp32 [Cin][Cout][8], the reference's layout)
 *   dw (or NULL): fp32 [Cin][Cout][8], ACCUMULATED into; needs x and a workspace of dua_deconv_k2s2_bwd_workspace bytes
 * (the bias gradient is a plain column sum of dy and is left to the caller). */
long dua_pack_deconv_weights_dgrad(int dtype, int Cin, int Cout, const float* w, void* w_packed, void* stream);
long dua_deconv_k2s2_bwd_workspace(const dua_conv3_desc* d);
int dua_deconv_k2s2_bwd(const dua_conv3_desc* d, const void* x, const void* dy, const void* w_packed_dgrad, void* dx,
                        float* dw, void* workspace, long workspace_bytes, void* stream);

/* final_conv (1x1x1, denoiser.py:282,311) for the training step, on a materialised activation u (channels-last
 * [voxels][u_stride], first C channels; C <= 64, multiple of 8) with W fp32 [K][C], b fp32 [K], K <= 16 classes:
 *   fwd: logits[v][k] = b[k] + sum_c u[v][c] W[k][c]
 *   bwd: du[v][c] = sum_k dlogits[v][k] W[k][c];  dW[k][c] += sum_v dlogits[v][k] u[v][c];  db[k] += sum_v dlogits[v][k]
 * dW, db are ACCUMULATED into (zero them first).  workspace (dua_head_bwd_workspace bytes; may be NULL): per-block
 * partial sums reduced by a second kernel; without it ~1000 blocks add into ~1000 addresses with fp32 atomics
 * (correct, ~10x slower).  voxels counts all samples of the batch. */
int dua_head_fwd(int dtype, long voxels, int C, int K, const void* u, int u_stride, const float* W, const float* b,
                 void* logits, int logits_stride, void* stream);
long dua_head_bwd_workspace(long voxels);
int dua_head_bwd(int dtype, long voxels, int C, int K, const void* dlogits, int dlogits_stride, const void* u,
                 int u_stride, const float* W, void* du, int du_stride, float* dW, float* db, void* workspace,
                 long workspace_bytes, void* stream);

/* Loss of the training step and its gradient (losses/loss.py:25-86 for the names "mse", "bce", "dice"; MONAI
 * DiceLoss(sigmoid=True) defaults):  mse = mean((sigmoid(p)-y)^2), bce = mean(BCEWithLogits(p,y)),
 * dice = mean_{n,c}(1 - (2I+e)/(S+Y+e)); the diffusion configs combine all three by "sum".
 * logits: channels-last [N][voxels][logits_stride] (first C used); labels: fp32 NCDHW [N][C][voxels].
 * reduce: sums (fp64 [N*C*4 + 2], pre-zeroed) += per (n,c) (I = sum s*y, S = sum s, Y = sum y, -), then
 *         (sum (s-y)^2, sum of BCE terms); the caller forms the selected terms and their combination from them.
 * grad  : dlogits = *gscale * (w_mse d mse/dp + w_bce d bce/dp + w_dice d dice/dp)  (gscale: device fp32 scalar or
 *         NULL = 1 -- it carries the loss scale and the derivative of the "mean" / "log" combine; w_*: 1 for the names
 *         in use, 0 otherwise), same layout as logits. */
int dua_seg_loss_reduce(int dtype, int N, int C, long voxels, const void* logits, int logits_stride, const float* labels,
                        double* sums, void* stream);
int dua_seg_loss_grad(int dtype, int N, int C, long voxels, const void* logits, int logits_stride, const float* labels,
                      const double* sums, const float* gscale, float w_mse, float w_bce, float w_dice, void* dlogits,
                      int dlogits_stride, void* stream);

/* ---- the rest of the training step (train.py:214-268 around the network): everything that is not a convolution --------
 * One launch each where torch needed five to twelve (profiles/r4_train_step_timeline_before.txt: 172 torch / hipBLASLt
 * launches, 1.5 ms of a 15.6 ms step).
 *
 * dua_stats_channel_sums: out[c] = fp32( sum over samples of the decoded "sum x" word pair of channel c ), c < C -- the bias
 *   gradient of a layer from the statistics rows dua_instnorm_stats accumulated over its output gradient.
 * dua_seg_loss_finish: the scalar tail of losses/loss.py:64-86 from the sums of dua_seg_loss_reduce: terms
 *   mse = sums[-2]/M, bce = sums[-1]/M, dice = mean_{n,c}(1 - (2I + 1e-5)/(S + Y + 1e-5)), M = N*C*voxels; the selected terms
 *   (use_* = 0/1) are added; combine 0 "sum" (or a single term), 1 "mean", 2 "log" (log(1 + total)).  loss[0] = L,
 *   dcomb[0] = dL/d(total) (1, 1/count, 1/(1 + total)); double arithmetic, fp32 results.
 * dua_q_sample_affine: out = sqrt_ab[t[n]] * (a * src + b) + sqrt_1m_ab[t[n]] * eps  (train.py:258-262: x_start = label * 2 - 1
 *   followed by q_sample, gaussian_diffusion.py:214-231) with sched = fp32 [T][2] (sqrt(alphas_cumprod), sqrt(1 - alphas_cumprod))
 *   and t = int64 [N] on the device: a * src + b is rounded to fp32 before the multiply, as the two torch passes it replaces. */
int dua_stats_channel_sums(int N, int C, int c_pad, const dua_stat_word* stats, float* out, void* stream);
int dua_seg_loss_finish(int N, int C, long voxels, int use_mse, int use_bce, int use_dice, int combine, const double* sums,
                        float* loss, float* dcomb, void* stream);
int dua_q_sample_affine(int N, long per_sample, const float* src, float a, float b, const float* eps, const float* sched, int T,
                        const long long* t, float* out, void* stream);

/* Timestep embedding under training (models/diffusion/utils.py:5-54, denoiser.py:51-52,65): for sample n
 *   e = [sin | cos](t[n] * freqs), z1 = W0 e + b0, h1 = swish(z1), z2 = W1 h1 + b1, s = swish(z2), add_b = Wp_b s + bp_b
 * for every TwoConv block b (its temb_proj).  ``add`` / ``dadd`` are BLOCK-MAJOR: block b's [N][cout_b] rows are contiguous
 * at float offset N * (cout_0 + .. + cout_{b-1}) -- each block reads / writes an ordinary dense [N][cout] tensor.
 * fwd: writes add and saved = fp32 [N][2*half + 4*hidden] (e, z1, h1, z2, s).
 * bwd: from dadd and saved, every parameter gradient in three launches: dw0 [hidden][2*half], db0, dw1 [hidden][hidden], db1 and
 *      blocks->dw[b] [cout_b][hidden], blocks->db[b] [cout_b] (all WRITTEN, summed over the samples in a fixed order);
 *      scratch = fp32 [N * hidden * (1 + ceil(P / 64) + hidden / 64)], P = sum of cout.  hidden: 256 or 512; 2*half <= 1024; N <= 64; sum of cout <= 4096; weights 16-byte aligned. */
#define DUA_TEMB_MAX_BLOCKS 16
typedef struct {
  int nblocks;
  int cout[DUA_TEMB_MAX_BLOCKS];
  const float* w[DUA_TEMB_MAX_BLOCKS];   /* temb_proj.weight [cout][hidden] */
  const float* b[DUA_TEMB_MAX_BLOCKS];   /* temb_proj.bias [cout] (fwd) */
  float* dw[DUA_TEMB_MAX_BLOCKS];        /* bwd */
  float* db[DUA_TEMB_MAX_BLOCKS];        /* bwd */
} dua_temb_blocks;
int dua_temb_train_fwd(int N, const long long* t, const float* freqs, int half_dim, int hidden, const float* w0, const float* b0,
                       const float* w1, const float* b1, const dua_temb_blocks* blocks, float* add, float* saved, void* stream);
int dua_temb_train_bwd(int N, int half_dim, int hidden, const float* w1, const dua_temb_blocks* blocks, const float* dadd,
                       const float* saved, float* scratch, float* dw0, float* db0, float* dw1, float* db1, void* stream);

/* AdamW over a list of fp32 tensors (train.py:121-126: torch.optim.AdamW(lr, weight_decay), betas (0.9, 0.999), eps 1e-8) with
 * the dynamic loss scaling of torch.cuda.amp around it (train.py:264-268), device-resident so that a captured step replays it:
 *   dua_grads_nonfinite: *found_inf = 1.0f if any gradient element is Inf / NaN (never written otherwise).
 *   dua_adamw_step: skipped entirely when found_inf && *found_inf != 0.  g' = g * (1 / *grad_scale) (grad_scale NULL: 1);
 *     p -= lr*wd*p;  m += (1 - b1)(g' - m);  v = b2 v + (1 - b2) g'^2;  p -= (lr / (1 - b1^k)) * m / (sqrt(v)/sqrt(1 - b2^k) + eps),
 *     k = *step + 1 (step: device int32, the number of updates applied so far; NOT advanced here -- one list may take several
 *     calls).  lr_dev (device fp32) overrides lr when non-NULL.  store_grad != 0 writes g' back (eager steps whose caller
 *     reads the unscaled gradients).  A list is <= DUA_ADAMW_MAX_TENSORS entries, passed BY VALUE (no device-side table to keep alive).
 *   dua_adamw_advance: end of a step -- found_inf == 0: ++*step, ++*growth, and when *growth == interval: *scale *= growth_factor,
 *     *growth = 0;  found_inf != 0: *scale *= backoff, *growth = 0;  then *seen = the flag (1 / 0) and *found_inf = 0
 *     (torch._amp_update_scale_ semantics; scale / growth / found_inf / seen may be NULL). */
#define DUA_ADAMW_MAX_TENSORS 64
typedef struct {
  int count;
  long numel[DUA_ADAMW_MAX_TENSORS];
  float* p[DUA_ADAMW_MAX_TENSORS];
  float* g[DUA_ADAMW_MAX_TENSORS];
  float* m[DUA_ADAMW_MAX_TENSORS];
  float* v[DUA_ADAMW_MAX_TENSORS];
} dua_adamw_list;
int dua_grads_nonfinite(const dua_adamw_list* list, float* found_inf, void* stream);
int dua_adamw_step(const dua_adamw_list* list, float lr, const float* lr_dev, float beta1, float beta2, float eps,
                   float weight_decay, const float* grad_scale, const float* found_inf, const int* step, int store_grad,
                   void* stream);
int dua_adamw_advance(int* step, float* found_inf, float* scale, int* growth, float growth_factor, float backoff, int interval,
                      float* seen, void* stream);

/* ---- library state ---------------------------------------------------------------------------------------------
 * dua_abi_version(): DUA_ABI_VERSION of the library that is loaded.  It changes whenever a struct of this header, the size
 * or layout of a buffer a caller allocates (e.g. the statistics words of dua_in_norm) or a signature changes; a binding
 * built against another header must refuse to run (diff_unet_amos_amd/_native.py does).
 * dua_prepare(): once per device and process, raises the dynamic-LDS limit (hipFuncSetAttribute) of every kernel of the
 * library that needs more than 64 KB and caches the device's compute-unit count.  Every launcher runs it on first use;
 * callers that capture launches into a hipGraph, or launch from several threads (torch's autograd runs backward() on a
 * worker thread), call it once up front, on the device they will use, so that no such call happens inside a capture.
 * Thread-safe.  Returns 0, DUA_ERR_ARG (no current device) or a hipError_t.
 * dua_prepared_kernels(): how many kernels dua_prepare() configures (tests). */
#define DUA_ABI_VERSION 8
int dua_abi_version(void);
int dua_prepare(void);
int dua_prepared_kernels(void);

/* Measurement aid (bench.py roofline.measured_mfma_ceiling): `workgroups` x 4 waves (one per SIMD at one workgroup per CU)
 * issue iters * 16 back-to-back v_mfma_f32_32x32x16_f16 (32 768 FLOP each) on random register operands.  stamps (or NULL):
 * 2 words per workgroup = (s_memtime cycles, s_memrealtime 100 MHz ticks) spent in the loop.  sink: any device float. */
int dua_mfma_probe(int workgroups, int iters, float* sink, unsigned long long* stamps, void* stream);

/* Measurement aid (tools/ubench_chain.py): one launch of a normalise -> compute -> accumulate-statistics chain reduced to
 * its dependent memory round trips.  mode 0 empty, 1 load/store, 2 + workgroup reduction and 64 system-scope atomics,
 * 3 + a read of the words the previous launch's atomics wrote.  in/out: workgroups * 256 floats; words: 64 x 8 bytes. */
int dua_chain_probe(int mode, int workgroups, const float* in, float* out, unsigned long long* words, void* stream);

/* Packs nn.Conv3d weight fp32[Cout][Cin_src][3][3][3] into the kernel's slab order
 * [cout_tile][chunk][kd][kh*3+kw][k-group][64][16 B].  in_perm (device int32[Cin_packed], may be
 * NULL = identity) gives for each packed input channel its source channel, or -1 for a zero pad
 * channel.  Returns bytes needed when w_packed is NULL. */
long dua_pack_conv3_weights(int dtype, int Cout, int Cin_src, int Cin_packed, const float* w, const int* in_perm,
                            void* w_packed, void* stream);

/* The same packing for the single-channel tap form of dua_conv3_desc.tap_channel_plus1 (DUA_F16, Cin_packed <= 32): the
 * slab weights of packed channel tap_channel are zeroed and a block [cout_tile][4 groups of 8 taps][64 couts][8] (taps
 * 27..31 zero) holding the 27 weights of SOURCE channel tap_src_channel per output channel is appended.  Returns bytes
 * needed when w_packed is NULL. */
long dua_pack_conv3_weights_tap(int dtype, int Cout, int Cin_src, int Cin_packed, int tap_channel, int tap_src_channel,
                                const float* w, const int* in_perm, void* w_packed, void* stream);

/* ---- InstanceNorm3d statistics -> per-(n,c) scale/shift (inspection / tests) -----------------------
 * The same arithmetic every consumer runs in its preamble, written out: scale, shift = fp32 [N][C]. */
int dua_instnorm_finalize(int N, int C, const dua_in_norm* in, float* scale, float* shift, void* stream);

/* ---- materialise: x_i = LeakyReLU(IN(raw)) + embeddings[i], and its MaxPool3d(2) --------------
 * Replaces the normalise/activate half of MONAI ADN for tensors with several consumers, the
 * skip-feature add at models/basic_unet/denoiser.py:300-304, nn.MaxPool3d(2) at
 * denoiser.py:100,106 (pretrained/basic_unet.py:99), and the skip half of torch.cat at
 * denoiser.py:190 (out is a channel slice of the concat buffer). */
typedef struct {
  int dtype;
  int N, D, H, W, C;
  int raw_stride;               /* channel stride of raw (offset 0) */
  int emb_stride;               /* channel stride of emb (offset 0), ignored when emb is NULL */
  int out_stride, out_off;
  int pool_stride;              /* channel stride of pooled (offset 0); D,H,W must be even */
  int out_blocked;              /* 1 = `out` is stored in 16-channel blocks (dua_conv3_desc.layout; out_stride, out_off
                                   multiples of 16); raw, emb and pooled are always channels-last */
} dua_materialize_desc;

int dua_materialize(const dua_materialize_desc* d, const void* raw, const dua_in_norm* in,
                    const void* emb, void* out, void* pooled, void* stream);

/* ---- ConvTranspose3d(k2, s2, bias) --------------------------------------------------------
 * Replaces MONAI UpSample(mode="deconv") = nn.ConvTranspose3d at
 * models/basic_unet/denoiser.py:161-170, writing into the "upsampled" channel slice of the concat
 * buffer (torch.cat at denoiser.py:190).  d->D/H/W are the INPUT extents; y has 2D x 2H x 2W. */
int dua_deconv_k2s2_fwd(const dua_conv3_desc* d, const void* x, const void* w_packed, const float* bias_padded,
                        const dua_in_norm* in, void* y, void* stream);

/* ---- diffusion elementwise arithmetic (guided_diffusion/gaussian_diffusion.py) ----------------
 * Tensors are contiguous fp32 with the batch outermost (any layout inside a sample).
 * q_sample (:187-205): coef = fp32[N][2] = {sqrt_alphas_cumprod[t], sqrt_one_minus_alphas_cumprod[t]}. */
int dua_q_sample(int N, long per_sample, const float* x0, const float* eps, const float* coef, float* out, void* stream);

#define DUA_MODE_LOGITS 0
#define DUA_MODE_DDPM 1   /* p_sample, :395-439   coef row {c1, c2, 1[t!=0]*exp(.5*logvar), -, -, -, -, -} */
#define DUA_MODE_DDIM 2   /* ddim_sample, :537-586 coef row {sqrt_recip, sqrt_recipm1, sqrt(acp_prev),
                             sqrt(1-acp_prev-sigma^2), 1[t!=0]*sigma, -, -, -} */
/* One reverse step given the model output: x_out = update(clamp(model_out,-1,1), x, eps);
 * xstart_out (may be NULL) = clamp(model_out); xstart_sum (may be NULL) += clamp(model_out)
 * (models/diffusion/diffusion.py:94-98).  coef = fp32[N][8]. */
int dua_sampler_step(int mode, int N, long per_sample, const float* model_out, const float* x, const float* eps,
                     const float* coef, float* x_out, float* xstart_out, float* xstart_sum, void* stream);

/* ---- fused denoiser tail --------------------------------------------------------------------
 * InstanceNorm+LeakyReLU of the last decoder block -> final_conv 1x1x1
 * (models/basic_unet/denoiser.py:282,311) -> sampler update -> running sum of x0^ -> x_{t-1} written
 * into the next step's denoiser input slice (torch.cat([image, x]) at denoiser.py:298, in place). */
typedef struct {
  int dtype;
  int N;
  long voxels;
  int K, raw_stride;       /* channels of the last decoder block, channel stride of raw */
  int C, CX;               /* classes; channel stride of the fp32 sampler state (8/16/24/32) */
  int mode;                /* DUA_MODE_* */
  int xin_stride;          /* channel stride of xin (x_{t-1} goes to channels [0, C)) */
  unsigned long long seed; /* Philox key when noise == NULL and seed_dev == NULL */
  const unsigned long long* seed_dev; /* optional DEVICE word holding the Philox key: lets a captured graph draw a
                                         fresh noise field per replay (the host rewrites the word, not the graph) */
} dua_tail_desc;

/* wf: fp32[C][K], bf: fp32[C].  coef: fp32[N][8] on device.  x_state: fp32[N][voxels][CX] in/out.
 * noise: fp32 NCDHW [N][C][voxels] or NULL (in-kernel Philox4x32-10 + Box-Muller, counter =
 * (element, *step_word)).  xin, xstart_sum ([N][voxels][CX]), logits and xstart (NCDHW fp32) may be
 * NULL.  In DUA_MODE_LOGITS only logits is written.
 * in == NULL (or in->stats == NULL): raw is an already materialised activation and goes to the 1x1x1 convolution as it
 * is -- the tail of SwinUNETRDenoiser.forward (`out` UnetOutBlock, models/swin_unetr/denoiser.py:399-400). */
int dua_final_conv_sampler(const dua_tail_desc* d, const void* raw, const dua_in_norm* in,
                           const float* wf, const float* bf, const float* coef, float* x_state, const float* noise,
                           const int* step_word, void* xin, float* xstart_sum, float* logits, float* xstart,
                           void* stream);

/* The same tail fed by the last UnetResBlock of SwinUNETRDenoiser (decoder1's conv_block) BEFORE its output is
 * materialised: the 1x1x1 convolution's input is assembled in registers as
 *   LeakyReLU(norm(raw) + norm_res(res)) + ra * (1 - sigmoid(ra))
 * -- norm2(conv2) + norm3(conv3) and the activation of UnetResBlock.forward (models/swin_unetr/blocks.py:306-316), plus the
 * reverse-attention term of the skip (models/swin_unetr/denoiser.py:397,405-408) -- with the arithmetic order of
 * dua_residual_norm_act, so both routes give the same bits.  DUA_F16, CX == 16, d->K in {32, 64} = channels rounded up to 32
 * (wf is [C][K] with zero columns behind `channels`; raw / res / ra_src may have channel strides below K). */
typedef struct {
  const void* res;         /* the shortcut branch (conv3 output), channels [0, channels) */
  int res_stride;
  dua_in_norm res_norm;    /* its InstanceNorm (norm3); stats must be set */
  const void* ra_src;      /* NULL or the tensor whose channels [ra_off, ra_off + channels) enter x * (1 - sigmoid(x)) */
  int ra_stride, ra_off;
  int channels;            /* real channels of raw / res / ra_src (multiple of 8, <= d->K) */
} dua_tail_residual;

int dua_final_conv_sampler_res(const dua_tail_desc* d, const void* raw, const dua_in_norm* in, const dua_tail_residual* r,
                               const float* wf, const float* bf, const float* coef, float* x_state, const float* noise,
                               const int* step_word, void* xin, float* xstart_sum, float* logits, float* xstart,
                               void* stream);

/* ---- timestep embedding ---------------------------------------------------------------------
 * table[i][:] = concat over TwoConv blocks of temb_proj(swish(TimeStepEmbedder(timesteps[i])))
 * (models/diffusion/utils.py:6-54; models/basic_unet/denoiser.py:51-52,65).  freqs = the
 * exp(-ln(1e4) j/(half-1)) vector, w0 [hidden][2*half], w1 [hidden][hidden], w_cat [P][hidden]. */
int dua_temb_table(int count, const int* timesteps, const float* freqs, int half_dim, int hidden, const float* w0,
                   const float* b0, const float* w1, const float* b1, const float* w_cat, const float* b_cat, int P,
                   float* table, void* stream);

/* Start of one denoiser evaluation: copy the embedding row(s) and sampler coefficients of the
 * current step into the fixed buffers the other kernels read.  Either rows_per_sample
 * (int32[N], training / "denoise") or (row_of_step[nsteps], *counter) (sampling loops: uses step
 * k = *counter, writes step_word[0] = k, then *counter = k + 1) selects the rows.  Replaces the
 * per-step host work at gaussian_diffusion.py:523,703 and respace.py:123-129.
 * Every device-side index is range-checked against table_rows / nsteps: an offender is clamped (no wild
 * read) and *err_word (may be NULL) is set to 1 for the host to inspect. */
int dua_step_begin(int N, int P, const float* table, int table_rows, const int* rows_per_sample, const int* row_of_step,
                   int nsteps, const float* coef_table, int* counter, float* cur_add, float* cur_coef, int* step_word,
                   int* err_word, void* stream);
/* The same, and the same launch also zeroes `clear_bytes` (a multiple of 16; `clear` 16-byte aligned) at `clear`: the
 * statistics arena of the evaluation that starts here (what the per-step .zero_() / memset node did). */
int dua_step_begin_clear(int N, int P, const float* table, int table_rows, const int* rows_per_sample, const int* row_of_step,
                         int nsteps, const float* coef_table, int* counter, float* cur_add, float* cur_coef, int* step_word,
                         int* err_word, void* clear, long clear_bytes, void* stream);

/* ---- one denoiser evaluation as ONE entry point -----------------------------------------------------------------
 * BasicUNetRDenoiser.forward (models/basic_unet/denoiser.py:284-312) + the sampler update that consumes it
 * (guided_diffusion/gaussian_diffusion.py:395-439 / 537-586; models/diffusion/diffusion.py:94-98), i.e. the loop body
 * of p_sample_loop_progressive / ddim_sample_loop_progressive (gaussian_diffusion.py:487-535, 667-716) for the HIP
 * denoiser: step begin -> zero the statistics arena -> the fixed sequence of convolution / materialise /
 * transposed-convolution launches -> fused final_conv + sampler tail.  The caller describes the sequence once (the
 * buffers are resident, only the step's rows / noise / outputs change) and may capture the call into a hipGraph and
 * replay it per step.  Enqueues on `stream`, never allocates, never synchronises; the first failing launch's code is
 * returned. */
#define DUA_OP_CONV3 1        /* dua_conv3d_k3_fwd(conv, x, w, bias, norm?, y, stats, workspace) */
#define DUA_OP_MATERIALIZE 2  /* dua_materialize(mat, raw = x, norm, emb, out = y, pooled) */
#define DUA_OP_DECONV 3       /* dua_deconv_k2s2_fwd(conv, x, w, bias, norm?, y) */
#define DUA_OP_UPCONV 4       /* dua_upconv_k3_fwd(up, xskip = x, u, norm, w, wu, bias (= bias_table), y, stats) */
typedef struct {
  int kind;                /* DUA_OP_* */
  int has_norm;            /* norm below describes the producer of x (fused InstanceNorm + LeakyReLU + add) */
  dua_conv3_desc conv;
  dua_materialize_desc mat;
  dua_in_norm norm;
  const void* x;           /* input (raw tensor for MATERIALIZE) */
  const void* w;           /* packed weights (CONV3 / DECONV) */
  const float* bias;       /* padded bias (CONV3 / DECONV) */
  void* y;                 /* output */
  dua_stat_word* stats;    /* CONV3: this layer's statistics rows inside the arena */
  const void* emb;         /* MATERIALIZE: encoder feature map added after the activation, or NULL */
  void* pooled;            /* MATERIALIZE: MaxPool3d(2) output, or NULL */
  dua_upconv_desc up;      /* UPCONV */
  const void* u;           /* UPCONV: coarse input (norm describes ITS producer) */
  const void* wu;          /* UPCONV: composed weights of the upsampled half */
} dua_step_op;

typedef struct {
  int N, P;                         /* batch; length of one timestep-embedding row */
  /* step begin (dua_step_begin) */
  const float* temb_table; int table_rows;
  const int* rows_per_sample;       /* "denoise": one row per sample ... */
  const int* row_of_step; int nsteps; const float* coef_table; int* counter;   /* ... or the sampling loop's tables */
  float* cur_add; float* cur_coef; int* step_word; int* err_word;
  /* statistics arena of all CONV3 ops, zeroed at the start of the evaluation */
  dua_stat_word* stat_arena; long stat_bytes;
  /* the launch sequence */
  const dua_step_op* ops; int n_ops;
  void* workspace; long workspace_bytes;     /* split-K scratch shared by the CONV3 ops */
  /* tail (dua_final_conv_sampler) */
  dua_tail_desc tail; const void* tail_raw; dua_in_norm tail_norm; const float* wf; const float* bf;
  float* x_state; const float* noise; void* xin; float* xstart_sum; float* logits; float* xstart;
} dua_denoiser_plan;

int dua_denoiser_step(const dua_denoiser_plan* plan, void* stream);

/* ---- windowed multi-head self-attention (DiffSwinUNETR, BASELINE config 5) -------------------------------------------
 * The core of WindowAttention.forward (models/swin_unetr/attention.py:97-120) between its two Linear layers:
 *   out[w, q, h*16 + d] = sum_k softmax_k( scale * <Q[w,h,q,:], K[w,h,k,:]> + bias[h,q,k] + mask[w % wpi, q,k] ) V[w,h,k,d]
 * qkv: [windows][tokens][3][heads][16] (the qkv Linear's output as it stands), element type = dtype; out:
 * [windows][tokens][heads*16].  bias_t: fp32 [heads][tokens(key)][tokens(query)] = relative_position_bias_table gathered
 * by relative_position_index (attention.py:103-106), TRANSPOSED; mask_t: fp32 [windows_per_image][key][query] from
 * compute_mask (attention.py:123-160; 0 / -100), transposed, or NULL for unshifted blocks.  tokens <= 352, head
 * dimension 16 (feature_size 48).  fp16 MFMA operands, fp32 softmax and accumulation.
 * region_ids (or NULL): the same mask in the form compute_mask derives it from -- uint8 [windows_per_image][tokens], the
 * shift region (0..26) of every token; the kernel adds -100 where query and key regions differ.  Give one of the two.
 * bias_table (or NULL): the relative-position bias in the form the reference stores it, transposed to
 * fp32 [heads][(2 grid_d - 1)(2 grid_h - 1)(2 grid_w - 1)] (attention.py:49-54 relative_position_bias_table; grid = the
 * window the index was built for, (7, 7, 7), also when the window itself is clipped); the kernel evaluates
 * relative_position_index (attention.py:56-73) from the token coordinates.  Takes precedence over bias_t. */
int dua_window_attention_fwd(int dtype, int windows, int tokens, int heads, int windows_per_image, const void* qkv,
                             const float* bias_t, const float* mask_t, const unsigned char* region_ids,
                             const float* bias_table, int grid_d, int grid_h, int grid_w, float scale, void* out,
                             void* stream);

/* PatchMerging.forward up to the reduction Linear (models/swin_unetr/patch.py:44-61; legacy != 0: the 3-D gather of
 * :70-91 with its duplicated corners): x [B][D][H][W][C] -> out [B][ceil(D/2)][ceil(H/2)][ceil(W/2)][8C] = LayerNorm_8C(gather),
 * odd extents zero padded.  x: the fp32 token stream; y (or NULL): the last block's MLP output in `dtype`, added on the fly.
 * gamma, beta: fp32 [8C]. */
int dua_patch_merge_norm(int dtype, int B, int D, int H, int W, int C, int legacy, const float* x, const void* y,
                         const float* gamma, const float* beta, float eps, void* out, void* stream);

/* Tail of UnetResBlock.forward (models/swin_unetr/blocks.py:308-316): out = LeakyReLU(IN(raw) + residual), raw = conv2's raw
 * output with its statistics in `in` (no add), residual = res as it is (res_in == NULL) or IN(res) with res_in (conv3 +
 * norm3).  Then the adds SwinUNETRDenoiser.forward applies to a block's output (swin_unetr/denoiser.py:370-399):
 * + post_add (embeddings[k]; NULL = none) and + ra_src * (1 - sigmoid(ra_src)) (reverse_attention of the skip, :405-408;
 * NULL = none).  Channels-last slices like everywhere else; C a multiple of 8. */
int dua_residual_norm_act(int dtype, int N, long voxels, int C, const void* raw, int raw_stride, const dua_in_norm* in,
                          const void* res, int res_stride, const dua_in_norm* res_in, void* out, int out_stride, int out_off,
                          float slope, const void* post_add, int post_stride, int post_off, const void* ra_src,
                          int ra_stride, int ra_off, int background, void* stream);
/* background (here and in dua_token_linear_desc): 1 = the launch runs on a second stream UNDER a chain of small launches of
 * another stream (like dua_conv3_desc.background): one workgroup per CU walks the data, so that the chain's launches find
 * free slots and memory bandwidth at once; the launch itself takes longer. */

/* ---- Swin token stream (models/swin_unetr/transformer.py) ------------------------------------
 * The residual stream x of a stage is fp32 [B][D][H][W][C]; GEMM / convolution operands are written in `dtype`.
 * C in {48, 96, 192, 384, 768} (feature_size 48). */
typedef struct dua_window_geom {
  int B, D, H, W, C;
  int wd, wh, ww;     /* window, already clipped to the map (attention.py:225-251 get_window_size) */
  int sd, sh, sw;     /* shift; 0 on clipped axes and in unshifted blocks */
} dua_window_geom;

/* SwinTransformerBlock.forward_part1 up to the attention (transformer.py:378-417): [x += y (the previous block's MLP
 * output, transformer.py:477-480; NULL = none)] -> norm1 -> zero pad to a window multiple -> roll(-shift) ->
 * window_partition.  out: [B * windows][tokens][C]. */
int dua_window_gather_norm(int dtype, const dua_window_geom* geom, float* x, const void* y, const float* gamma,
                           const float* beta, float eps, void* out, void* stream);
/* The way back (transformer.py:417-431, 475-476, 433): x += crop(roll(+shift)(window_reverse(yw)));  out = norm2(x)
 * ([B][D][H][W][C] in `dtype`) for the MLP. */
int dua_window_scatter_add_norm(int dtype, const dua_window_geom* geom, float* x, const void* yw, const float* gamma,
                                const float* beta, float eps, void* out, void* stream);
/* Between stages (transformer.py:277-312): x = y + tadd[b] (t_proj[i](swish(t)), fp32 [B][tadd_stride]; NULL = none; x may
 * be NULL after the last stage), out = layer_norm(x) without affine (proj_out, :253-268) + emb (the encoder's feature map,
 * swin_unetr/denoiser.py:367-368; NULL = none), written to channels [out_off, out_off + C) of a channels-last buffer. */
int dua_stage_out(int dtype, int B, long tokens_per_sample, int C, const void* y, const float* tadd, int tadd_stride,
                  float eps, const void* emb, float* x, void* out, int out_stride, int out_off, void* stream);
/* PatchEmbed (Conv3d k = s = 2, bias; transformer.py:185-191) on a channels-last input slice [B][D][H][W][Cin_stride]
 * (first Cin_packed channels), fused with the stage-0 adds above.  w_packed: fp32 [8 taps (kd, kh, kw)][Cin_packed][E],
 * E = 48. */
int dua_patch_embed(int dtype, int B, int D, int H, int W, int Cin_stride, int Cin_packed, int E, const void* in,
                    const float* w_packed, const float* bias, const float* tadd, int tadd_stride, float eps, const void* emb,
                    float* x, void* out, int out_stride, int out_off, void* stream);
/* Sum / sum of squares per (n, c) of a channels-last slice, accumulated into a dua_in_norm statistics buffer (zeroed by
 * the caller): the statistics of UnetResBlock's 1x1x1 conv3 (blocks.py:286-296), whose GEMM is a library call. */
int dua_instnorm_stats(int dtype, int N, long voxels, int C, const void* x, int x_stride, int x_off, dua_stat_word* stats,
                       int c_pad, void* stream);
/* Token GEMM with a fused epilogue for the tall / skinny Linear layers of the fine Swin stages and the 1x1x1 conv3 of
 * UnetResBlock: out[token][n] = sum_k A[token][k] W[n][k] (+ bias[n]), fp16 operands (A: [samples * M][lda], W: the
 * nn.Linear weight [N][K] as it is), fp32 accumulation.  K <= 384, N <= 192 per call (callers split wider layers by rows of
 * W).  Modes:
 *   PLAIN    out[token][out_off + n] (fp16, row stride ldc)                              qkv, proj, reduction
 *   GELU     the same after exact GELU                                                   MLPBlock linear1 + act
 *   STATS    PLAIN without bias, and stats[sample][replica][4][c] += (sum, sum^2) of the ROUNDED outputs (N <= 64; M = voxels
 *            per sample, `samples` grid rows)                                            conv3 + norm3 statistics
 *   RESIDUAL x[token][n] += out + bias on the fp32 stream                                x + mlp(norm2(x)), transformer.py:477-480
 *   SCATTER  token = window order: x[voxel(token)] += out + bias, ln_out[voxel] = LayerNorm(x)*gamma+beta; padding tokens
 *            dropped (window_reverse, roll back, crop, shortcut add, norm2: transformer.py:417-434, 475-476) */
#define DUA_TOKLIN_PLAIN 0
#define DUA_TOKLIN_GELU 1
#define DUA_TOKLIN_STATS 2
#define DUA_TOKLIN_RESIDUAL 3
#define DUA_TOKLIN_SCATTER 4
typedef struct dua_token_linear_desc {
  const void* A; int lda; long M; int K, N;
  const void* W; const float* bias;
  int mode, samples;
  void* out; int ldc, out_off;
  float* x;
  dua_stat_word* stats; int c_pad;
  dua_window_geom geom; const float* gamma; const float* beta; float eps; void* ln_out;
  int background;          /* dua_token_linear only: n > 0 = at most n workgroups per CU (see dua_residual_norm_act) */
} dua_token_linear_desc;
int dua_token_linear(const dua_token_linear_desc* d, void* stream);
/* The same contraction for the COARSE Swin stages and the wide 1x1x1 convolutions (stages 1-3: qkv / proj / linear1 / linear2 /
 * reduction, attention.py:97-120, transformer.py:433-435, patch.py:89-92; conv3 of UnetResBlock, blocks.py:311-314) as a
 * tiled MFMA GEMM that streams BOTH operands (any K and N that are multiples of 8; M <= 4 M tokens): modes PLAIN, GELU
 * (out, ldc, out_off as above) and RESIDUAL (x [M][N] fp32 += result).  samples, stats, geom, gamma, beta, ln_out unused.
 * workspace (may be NULL): fp32 scratch that lets the launcher divide K over workgroups for the small-token layers (a few
 * dozen tiles walking thousands of K) and finish with a second launch; dua_token_gemm_workspace gives the bytes (0: the
 * shape is not split). */
long dua_token_gemm_workspace(long M, int K, int N);
int dua_token_gemm(const dua_token_linear_desc* d, void* workspace, long workspace_bytes, void* stream);

/* The MLP of a Swin block in one launch (fp16 operands, C = 48 or 96, hidden = 4 C): x[token] += linear2(GELU(linear1(ln2[token])))
 * on the fp32 stream (MONAI MLPBlock; transformer.py:376,433-434,477-480).  ln2: fp16 [tokens][C] (norm2 of the stream, from
 * dua_token_linear SCATTER / dua_window_scatter_add_norm); W1 [4C][C], W2 [C][4C]: the nn.Linear weights as they are (fp16);
 * b1 [4C], b2 [C]: fp32.  The hidden activation stays in registers (linear1's accumulator is linear2's operand). */
int dua_swin_mlp(long tokens, int C, const void* ln2, const void* W1, const float* b1, const void* W2, const float* b2, float* x,
                 void* stream);

/* Exact (erf) GELU in place between the two MLP GEMMs (MONAI MLPBlock act "GELU"). */
int dua_gelu(int dtype, long elems, void* x, void* stream);

/* ---- layout / packing at the API boundary -------------------------------------------------- */
/* nn.Linear on fp32 rows, the parity plan of the Swin path (attention.py:97-120, transformer.py:433-435, patch.py:89-92,
 * blocks.py:311-314): out[m][n] = sum_k A[m][k] W[n][k] + bias[n] (bias may be NULL), then the exact GELU when gelu != 0.
 * A: M rows of lda floats (first K used), W: [N][K] dense, out: M rows of ldc floats.  K and lda multiples of 4, A and W
 * 16-byte aligned.  Exact-fp32 MFMA; no library GEMM is launched. */
int dua_linear_f32(long M, int K, int N, const float* A, long lda, const float* W, const float* bias, float* out, long ldc,
                   int gelu, void* stream);

/* Many 3x3x3 weight tensors in one launch (a training step repacks every layer twice: forward and data-gradient layout).
 * items: HOST array of count <= 64 entries.  kind 0 = dua_pack_conv3_weights(F16, Cout, Cin, packed, w, NULL, out),
 * kind 1 = dua_pack_conv3_weights_dgrad(F16, Cout, Cin, packed, w, out); fp16 only, Cin % 4 == 0, w and out 16-byte aligned,
 * out sized by the per-layer call with w_packed == NULL.  Anything else: DUA_ERR_ARG (pack that layer on its own). */
typedef struct {
  int kind;
  int Cout, Cin;
  int packed;              /* kind 0: Cin_packed (channels of the input buffer); kind 1: Cout_packed (channels of the dy buffer) */
  const float* w;
  void* out;
} dua_pack_item;
int dua_pack_conv3_weights_batch(int dtype, int count, const dua_pack_item* items, void* stream);

/* nn.ConvTranspose3d weight fp32[Cin][Cout][2][2][2] -> [tap][cout_tile][chunk][k-group][64][16 B].
 * Returns bytes needed when w_packed is NULL. */
long dua_pack_deconv_weights(int dtype, int Cin, int Cout, const float* w, void* w_packed, void* stream);

/* NCDHW fp32 [N][C][voxels] -> channels-last slice; channels [C, C_fill) of the slice are zeroed. */
int dua_to_channels_last(int dtype, int N, int C, long voxels, const float* src, void* dst, int Cstride, int C_off,
                         int C_fill, void* stream);
int dua_from_channels_last(int dtype, int N, int C, long voxels, const void* src, int Cstride, int C_off, float* dst,
                           void* stream);
/* Whole rows: dst[n][v][0 .. Cstride) = (src0 channels | src1 channels | zeros), src1 may be NULL with C1 = 0 -- the network
 * input torch.cat((image, x), dim=1) (denoiser.py:298) in one pass of 16-byte stores.  dst dense (row = Cstride elements, at
 * most 64 bytes, a multiple of 16), 16-byte aligned. */
int dua_to_channels_last_rows(int dtype, int N, int C0, const float* src0, int C1, const float* src1, long voxels, void* dst,
                              int Cstride, void* stream);

#ifdef __cplusplus
}
#endif
#endif
