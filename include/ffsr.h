/*
 * libffsr_hip.so -- C ABI of the MI355X (gfx950) kernels behind the FreqFusionSR x4 inference hot path.
 *
 * The reference (Nikhil-AI-Labs/Image-Super-Resolution) is 100 % Python and dispatches every operator to ATen;
 * its only native seam is mamba_ssm's selective_scan_fn (mambair_arch.py:276, :356-362).  The "FFI" a maintainer
 * binds is therefore ctypes: each entry point below replaces the ATen / mamba_ssm call
 * sequence of one reference function, cited as file:line relative to /root/reference.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 data owned by the caller; nothing is allocated or freed inside;
 *    `stream` is a hipStream_t (NULL = default stream); calls are asynchronous and thread-safe per stream.
 *  - activations are channels-last: a tensor [B, H, W, C] is a matrix of B*H*W pixel rows with a row stride
 *    `ld*` in floats (ld >= C, ld % 4 == 0 for the vectorised paths; padded channels must hold finite values,
 *    zero where a consumer contracts over them).
 *  - return value: 0 = launched, FFSR_EINVAL (-1) = rejected argument, FFSR_ELAUNCH (-2) = HIP launch error.
 *  - activation codes: 0 none, 1 GELU(erf), 2 ReLU, 3 LeakyReLU(slope), 4 sigmoid, 5 SiLU.
 */
#ifndef FFSR_H
#define FFSR_H

#ifdef __cplusplus
extern "C" {
#endif

#define FFSR_OK 0
#define FFSR_EINVAL (-1)
#define FFSR_ELAUNCH (-2)

/* Implicit-GEMM convolution / linear layer on the f32 MFMA.
 *   out[pix, n] = res[pix, n] * rvec[n] * rscale + act(sum_k A[pix, k] * wgt[n, k] + bias[n]) * cvec[n] * cscale
 * in: [B,H,W,ldi] (Cin channels used, Cin % 4 == 0); wgt: [N, KH*KW*Cin] (tap-major, channel-minor);
 * bias/res/cvec/rvec/akscale may be NULL.  akscale [B, Cin] (1x1 only) scales the A operand per (batch, k)
 * with akrows pixels per batch.  shuffle = 2 fuses nn.PixelShuffle(2) into the store (out/res are then
 * [B, 2Ho, 2Wo, ld] with N/4 channels).  tile_hint 0 = automatic.
 * Replaces: nn.Linear / nn.Conv2d everywhere on the path, e.g. drct_arch.py:166,168,83-85 (qkv, proj, Mlp),
 * :601-620 (Upsample conv + PixelShuffle), nafnet_arch.py:110-131 (conv1/3/4/5, SCA scale via akscale) and
 * :181-187 (ups), mambair_arch.py:238,279 (in/out_proj), :346 (x_proj, all 4 directions in one GEMM),
 * enhanced_fusion_v2.py:569-576 (refine stack), hierarchical_fusion.py:90-124, edge_enhancement.py:102-177. */
int ffsr_conv2d_f32(const float* in, const float* wgt, const float* bias, float* out, const float* res,
                    const float* cvec, const float* rvec, const float* akscale, int B, int H, int W, int Cin, int ldi,
                    int N, int ldo, int ldr, int KH, int KW, int stride, int pad_h, int pad_w, int act, float slope,
                    float cscale, float rscale, int shuffle, int akrows, int tile_hint, void* stream);

/* Arithmetic of the split-bf16 kernels (ffsr_conv2d_bf16x3, ffsr_conv2d_planes, ffsr_tok_*): terms = 3 (default) evaluates each
 * fp32 product as hi*hi + hi*lo + lo*hi on the bf16 MFMA; terms = 1 is PLAIN bf16 -- operands rounded to bf16, one MFMA per
 * product, fp32 accumulate (the reference's GPU route runs under torch autocast, models/team29_FreqFusionSR/io.py:263; BASELINE
 * config 2 is quoted in bf16).  ~2e-2 max-abs on NAFNet alone: a precision option (FFSR_GEMM_MODE=bf16), never the default.
 * Process-wide; set it before the first launch of a run. */
int ffsr_set_gemm_terms(int terms);

/* Same operator with the products evaluated as 3-term split-bf16 MFMAs (hi*hi + hi*lo + lo*hi, fp32 accumulate;
 * ~1e-5 relative per product).  wgt_hi / wgt_lo: bf16 planes [n_rows_padded, ldw] of the weight matrix, pre-split at
 * pack time (hi = bf16(w), lo = bf16(w - hi)) and zero padded to n_rows_padded % bn == 0 rows and ldw % 32 == 0
 * columns (n_rows_padded % 128 == 0); zeros: >= 64 bytes of zeros (read for padding taps / rows); bn = column tile,
 * 32 (N <= 32), 64 or 128.  shuffle = 3 (bn 128, act none, N % 64 == 0) fuses NAFNet's SimpleGate (nafnet_arch.py:21-24,
 * :127-129) into the store: out[pix, j] = res * rvec * rscale + x1[j] * x2[j] * cvec[j] * cscale with N / 2 output
 * columns, where the weight / bias rows are packed so that GEMM column 64 b + i is x1[32 b + i] and 64 b + 32 + i is
 * x2[32 b + i] (i < 32; x1 / x2 = the two channel halves of the convolution output). */
int ffsr_conv2d_bf16x3(const float* in, const void* wgt_hi, const void* wgt_lo, int ldw, int n_rows_padded,
                       const float* zeros, const float* bias, float* out, const float* res, const float* cvec,
                       const float* rvec, const float* akscale, int B, int H, int W, int Cin, int ldi, int N, int ldo,
                       int ldr, int KH, int KW, int stride, int pad_h, int pad_w, int act, float slope, float cscale,
                       float rscale, int shuffle, int akrows, int bn, void* stream);

/* The same split-bf16 operator with the A operand PRE-SPLIT in HBM: a_hi / a_lo are bf16 planes [B*H*W, Cp]
 * (hi = bf16(x), lo = bf16(x - hi); Cp % 32 == 0, channels >= the true Cin are zero) -- the same bytes as fp32.  Both
 * operands then go global -> LDS by LDS-DMA with no vector-ALU work in the main loop.  Weights: bf16 planes
 * [n_rows_padded, KH*KW*Cp] (tap-major, Cp channels per tap, zero padded; n_rows_padded % bn == 0).  Outputs: fp32
 * `out` (may be NULL) and / or bf16 planes out_hi / out_lo [M, ldp] (may be NULL; ldp = N rounded up to 32, columns
 * >= N are written as zeros) for a following ffsr_conv2d_planes.  act: 0 none, 1 GELU, 2 ReLU, 3 LeakyReLU only;
 * act | 0x100 (both outputs given, no res / cvec / cscale): `out` receives the PRE-activation z, the planes act(z) -- the
 * training step keeps z for the activation's backward and hands act(z) to the next layer without an fp32 round trip.
 * bm x bn = 128 / 256 rows x 64 / 128 / 192 / 256 columns tile (bm 0 = 128); stages = LDS pipeline depth (0 = default:
 * 2; 2 or 3), or 4 = the tap-strip variant for 3x3 / stride 1 / pad 1 convolutions (bm 128, bn 64 or 128 only): one
 * staged strip of A rows serves the three horizontal taps.  Replaces the same reference calls as
 * ffsr_conv2d_f32 wherever the producer of the input can emit planes (LayerNorm, attention, gates, a previous GEMM). */
int ffsr_conv2d_planes(const void* a_hi, const void* a_lo, int Cp, const void* wgt_hi, const void* wgt_lo,
                       int n_rows_padded, const void* zeros, const float* bias, float* out, const float* res,
                       const float* cvec, const float* rvec, void* out_hi, void* out_lo, int ldp, int B, int H, int W,
                       int N, int ldo, int ldr, int KH, int KW, int stride, int pad_h, int pad_w, int act, float slope,
                       float cscale, float rscale, int bm, int bn, int stages, void* stream);

/* Token-stationary fused chain: per token row x [K] (fp32, row stride ldx)
 *     h   = act1( W1 . pre(x) + b1 )              pre = identity or the normalisation (x - mean) * rstd of nn.LayerNorm
 *                                                 (its affine part is folded into W1 / b1 by the packer)
 *     y   = (W2 . h + b2) * cvec * cscale + res * rvec * rscale
 *     out = g2 ? LayerNorm_N(y) * g2 + be2 + res2 : y         -> fp32 `out` and / or bf16 hi / lo planes [M, ldp]
 * in ONE kernel: a wave keeps its 16 tokens' rows in registers as MFMA operands, the hidden layer h never leaves the
 * registers, W1 / W2 stream through LDS in fragment-major order (packed by image-super-resolution_amd/ops.py::pack_tok_chain:
 * w1 = bf16 [steps][G][ceil(K/32)][2 planes][64 lanes][8], b1 = fp32 [steps * G * 16] in the same tile order,
 * w2 = bf16 [steps][ceil(N/16)][2][64][8] with the k slots in accumulator order).  Products are 3-term split-bf16 MFMAs.
 * mode 0 (G = 2): act1 = exact-erf GELU (erfc by Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7), 32 hidden features per step:
 *   the Swin / GRL Mlp with its LayerNorm and residual -- drct_arch.py:77-95 + :405-407 (x + mlp(norm2(x))), GRL's post-norm
 *   form mixed_attn_block_efficient.py:543-554 (x + norm2(mlp(x))), swin_v1_block.py Mlp.
 * mode 1 (G = 4): SimpleGate, h = (W1a x + b1a) * (W1b x + b1b), 32 gated features per step: NAFBlock's second half
 *   y + gamma * conv5(SimpleGate(conv4(norm2(y)))), nafnet_arch.py:125-131.
 * K, N % 4 == 0; supported (ceil(K/32), ceil(N/16)): (4,8) (6,12) (7,14) (8,16) (9,18) (10,20) for mode 0, (2,4) (4,8) for mode 1.
 * waves: 8, 11 or 12 waves of 16 tokens per workgroup (0 = 8). */
int ffsr_tok_chain_f32(const float* x, int ldx, const void* w1, const float* b1, const void* w2, const float* b2,
                       const float* cvec, const float* res, int ldr, const float* rvec, const float* g2,
                       const float* be2, const float* res2, int ldr2, float* out, int ldo, void* out_hi, void* out_lo,
                       int ldp, long long M, int K, int N, int steps, int mode, int pre_ln, float eps1, float eps2,
                       float cscale, float rscale, int waves, void* stream);

/* ffsr_tok_chain_f32 followed, inside the same kernel, by a third linear layer applied to the chain's output row y while it
 * is still in registers:   out3 = act3(W3 y + b3) * cscale3 + res3 * rscale3   ([M, ldo3], N3 columns).
 * `out` may be NULL: y itself then never reaches HBM.  w3 / b3 in the ffsr_tok_gemm_f32 packing (pack_tok_gemm); act3: none /
 * ReLU / LeakyReLU.  Requires ceil(K/32) == ceil(N/32).  Replaces, per block of DRCT's residual dense group, the Swin block's
 * x + mlp(norm2(x)) AND the 1x1 "adjust" convolution that consumes it (drct_arch.py:292-301: adjust1..4 + LeakyReLU(0.2) write
 * the 32 new channels of the dense concatenation; adjust5 * 0.2 + x closes the group). */
int ffsr_tok_chain_tail_f32(const float* x, int ldx, const void* w1, const float* b1, const void* w2, const float* b2,
                            const float* res, int ldr, float* out, int ldo, long long M, int K, int N, int steps, int mode,
                            int pre_ln, float eps1, const void* w3, const float* b3, const float* res3, int ldr3, float* out3,
                            int ldo3, int N3, int act3, float slope3, float cscale3, float rscale3, int waves, void* stream);

/* ffsr_tok_chain_f32 (mode 0) with a HEAD: a K -> K linear layer in front of the chain, inside the same kernel.  The rows read
 * from HBM are the head's input a [M, lda]; the chain's input row
 *     x1 = LN0?(W0 a + b0) + hres + hres2 * hvec2[row / rows_per_batch]      (LN0 = LayerNorm(g0, be0, eps0) when g0 != NULL)
 * exists only in registers.  Without a post-LN (g2 == NULL):  y = x1 + W2 gelu(W1 norm?(x1) + b1) + b2    (Swin block:
 * x1 = x + proj(attention), drct_arch.py:400-407 -- hres = x); with one:  y = LN2(W2 gelu(W1 x1 + b1) + b2) + x1   (GRL block:
 * x1 = x + norm1(proj(attention)) + conv_branch * channel_attention, mixed_attn_block_efficient.py:536-554 -- hres = x,
 * hres2 = the CAB convolution output, hvec2 = its per-image channel-attention scale [batches, K]).  y -> out / planes as in
 * ffsr_tok_chain_f32; the optional tail (w3 != NULL) as in ffsr_tok_chain_tail_f32.  w0 / b0 in the ffsr_tok_gemm_f32 packing.
 * ascale [batches, K] (or NULL): the head's input row is first scaled per image and channel; mode 1 = the gated chain:
 * NAFBlock  y = x + beta * conv3(gated * sca);  out = y + gamma * conv5(SimpleGate(conv4(norm2(y))))   (nafnet_arch.py:122-131;
 * beta / gamma folded into W0 / W2 by the packer, hres = x, ascale = the channel attention sca, K = 64 or 128).
 * Replaces three launches (proj GEMM, residual / LayerNorm kernel, MLP chain) and the two HBM round trips between them. */
int ffsr_tok_head_chain_f32(const float* a, int lda, const float* ascale, const void* w0, const float* b0, const float* g0,
                            const float* be0, float eps0, const float* hres, int ldhr, const float* hres2, int ldhr2,
                            const float* hvec2, int rows_per_batch, const void* w1, const float* b1, const void* w2,
                            const float* b2, const float* g2, const float* be2, float eps2, float* out, int ldo, void* out_hi,
                            void* out_lo, int ldp, long long M, int K, int steps, int mode, int pre_ln, float eps1,
                            const void* w3, const float* b3, const float* res3, int ldr3, float* out3, int ldo3, int N3,
                            int act3, float slope3, float cscale3, float rscale3, int waves, void* stream);

/* Token-stationary projection with a full-row epilogue (and MambaIR's out_norm / gate as its prologue), one kernel:
 *     a   = x                                                       (xdirs == 1)
 *         = ((x[0] + x[2 xstride]) + x[xstride]) + x[3 xstride]     (xdirs == 4: the four scan directions, mambair_arch.py:381)
 *     a   = LN(a; pg, pb, peps)   when pg != NULL;     a *= silu(z)   when z != NULL          (out_norm, y * F.silu(z))
 *     y   = (W0 a + b0) * cvec * cscale + res * rvec * rscale                                 (out_proj; x * skip_scale + ...)
 *     g2 != NULL:  n = LN(y; g2, be2, eps2);  planes <- n;  out <- y if out_pre_ln else n      (ln_2 feeding the conv branch)
 *     g2 == NULL:  out / planes <- y
 * w0 / b0 in the ffsr_tok_gemm_f32 packing.  Supported: K = 321..384 -> N = 161..192 (MambaIR d_inner 360 -> 180).  Replaces
 * ffsr_mamba_norm_gate_planes_f32 + the out_proj GEMM + the skip scale_add + the ln_2 LayerNorm of a VSS block
 * (mambair_arch.py:381-386, :417-419) and the HBM round trips of the gated tensor and of y between them. */
int ffsr_tok_proj_f32(const float* x, long long xstride, int xdirs, int ldx, const float* z, int ldz, const float* pg,
                      const float* pb, float peps, const void* w0, const float* b0, const float* cvec, const float* res,
                      int ldr, const float* rvec, const float* g2, const float* be2, float eps2, float* out, int ldo,
                      int out_pre_ln, void* out_hi, void* out_lo, int ldp, long long M, int K, int N, float cscale,
                      float rscale, int waves, void* stream);

/* Token-stationary single GEMM with the producer fused in front: per token row x [K] (fp32, row stride ldx)
 *     out = act( W1 . pre(x) + b1 ) * cvec * cscale       -> fp32 `out` [M, ldo] and / or bf16 hi / lo planes [M, ldp]
 * pre = identity or nn.LayerNorm's normalisation (affine part folded into W1 / b1 by the packer).  Same kernel family as
 * ffsr_tok_chain_f32 (rows in registers as MFMA operands, W1 streamed fragment-major through LDS by LDS-DMA, persistent
 * workgroups with the next tile's rows in flight, 3-term split-bf16 products), 32 output features per step, stored as they
 * are produced.  w1 = bf16 [ceil(N/32)][2][ceil(K/32)][2 planes][64][8] with the rows in the lane-column order, b1 = fp32
 * [ceil(N/32) * 32] in the same order (image-super-resolution_amd/ops.py::pack_tok_gemm).  act: none / GELU / ReLU / LeakyReLU.
 * K, N % 4 == 0; ceil(K/32) in {2, 4, 6, 7, 8, 9, 10, 12}.  Replaces LayerNorm + nn.Linear pairs whose LayerNorm output has no
 * other consumer: norm1 + qkv of the Swin blocks (drct_arch.py:385-388 + :166), ln_1 + in_proj of the VSS blocks
 * (mambair_arch.py:417 + :238), norm1 + conv1 of the NAFBlocks (nafnet_arch.py:113-115), and the dense-block 1x1 "adjust"
 * convolutions + LeakyReLU (drct_arch.py:292-301). */
int ffsr_tok_gemm_f32(const float* x, int ldx, const void* w1, const float* b1, const float* cvec, float* out, int ldo,
                      void* out_hi, void* out_lo, int ldp, long long M, int K, int N, int pre_ln, float eps1, int act,
                      float slope, float cscale, int waves, void* stream);

/* fp32 [M, C] (row stride ldx) -> bf16 hi / lo planes [M, ldp] (ldp % 32 == 0, columns C..ldp-1 zero). */
int ffsr_split_planes(const float* x, int ldx, void* hi, void* lo, int ldp, long long M, int C, void* stream);

/* out = LayerNorm_C(x) * gamma + beta (+ res1) (+ res2); biased variance, rows of C <= 1024.
 * Replaces nn.LayerNorm (drct_arch.py:385, grl mixed_attn_block_efficient.py:543-554 incl. the post-norm residual
 * sums, mambair_arch.py:417-419) and NAFNet's LayerNorm2d (nafnet_arch.py:26-44). */
int ffsr_layernorm_f32(const float* x, int ldx, const float* gamma, const float* beta, float eps, float* out, int ldo,
                       const float* res1, int ldr1, const float* res2, int ldr2, int M, int C, void* stream);

/* ffsr_layernorm_f32 that can also (or only) emit the result as bf16 hi / lo planes [M, ldp] (ldp = C rounded up to
 * 32, pad columns zero) for ffsr_conv2d_planes: out may be NULL when out_hi / out_lo are given.  Plane output needs
 * the vectorised path (C % 4 == 0, 16-byte aligned rows).  res2_vec [B, C] (optional) scales res2 per (batch, channel)
 * with rows_per_batch rows per batch: out = LN(x) + res1 + res2 * res2_vec[batch] -- GRL's x + LN(attn(x)) + CAB(x)
 * with the RCAN channel attention of CAB folded in (mixed_attn_block_efficient.py:543-554, mixed_attn_block.py:942-983). */
int ffsr_layernorm_planes_f32(const float* x, int ldx, const float* gamma, const float* beta, float eps, float* out,
                              int ldo, void* out_hi, void* out_lo, int ldp, const float* res1, int ldr1,
                              const float* res2, int ldr2, const float* res2_vec, int rows_per_batch, int M, int C,
                              void* stream);

/* out = clamp(act(x * pre) * alpha * cscale[n] + beta + cbias[n], lo, hi) (cscale, cbias [C] optional; clamp only if
 * do_clamp).  Covers eval-mode BatchNorm (large_kernel_attention.py:143), mean shifts, clamps, activations. */
int ffsr_unary_f32(const float* x, int ldx, float* out, int ldo, long long M, int C, int act, float slope, float pre,
                   float alpha, float beta, const float* cscale, const float* cbias, int do_clamp, float lo, float hi,
                   void* stream);

/* out = alpha * a * avec[n] + beta * b * bvec[m / rows_per_batch, n]  (avec [C], b, bvec [B, C] optional).
 * Replaces the residual / skip-scale / channel-attention combinations, e.g. mambair_arch.py:418-419,
 * ChannelAttention (mambair_arch.py:20-38, grl mixed_attn_block.py:942-961). */
int ffsr_scale_add_f32(const float* a, int lda, const float* avec, const float* b, int ldb, const float* bvec,
                       int rows_per_batch, float* out, int ldo, long long M, int C, float alpha, float beta, void* stream);

/* out = alpha * a * b' + gamma * c;  b' = b[m, n] (bmode 0) or b[m * ldb] broadcast over channels (bmode 1).
 * Replaces SimpleGate (nafnet_arch.py:47-55), LKA / spatial gates (large_kernel_attention.py:105,
 * hierarchical_fusion.py:42, edge_enhancement.py:88). */
int ffsr_mul_add_f32(const float* a, int lda, const float* b, int ldb, int bmode, const float* c, int ldc, float* out,
                     int ldo, long long M, int C, float alpha, float gamma, void* stream);

/* out[b, c] = mean over the R rows of batch b (nn.AdaptiveAvgPool2d(1)); part: scratch [B, nchunk, C]. */
int ffsr_colmean_f32(const float* x, int ldx, float* out, float* part, int B, int R, int C, int nchunk, void* stream);

/* RCAN channel attention of CAB in two launches: att[b, c] = sigmoid(W2 relu(W1 mean_b + b1) + b2), mean_b = spatial mean of
 * the R rows of batch b (mambair_arch.py:20-38, grl mixed_attn_block.py:942-961: AdaptiveAvgPool2d(1), Conv2d(C, C/r, 1),
 * ReLU, Conv2d(C/r, C, 1), Sigmoid).  w1 [sq, ldw1], w2 [C, ldw2] (nn.Conv2d 1x1 weights as matrices), C <= 1024, sq <= 64;
 * part: scratch [B, nchunk, C]. */
int ffsr_channel_attention_f32(const float* x, int ldx, float* part, int nchunk, const float* w1, int ldw1, const float* b1,
                               const float* w2, int ldw2, const float* b2, float* out, int B, int R, int C, int sq, void* stream);

/* Depthwise KHxKW convolution, stride 1, zero padding; w tap-major [KH*KW, C]; act fused.
 * Replaces mambair_arch.py:239-247,378 (dw3x3 + SiLU) and the LKA chain large_kernel_attention.py:92-99. */
int ffsr_dwconv2d_f32(const float* in, int ldi, const float* w, const float* bias, float* out, int ldo, int B, int H,
                      int W, int C, int KH, int KW, int pad_h, int pad_w, int act, void* stream);

/* NAFBlock middle: t = dw3x3(in [..,2C]); out[.., c] = t[c] * t[c + C]; pooled[b, c] = spatial mean of out.
 * Replaces nafnet_arch.py:115-117 (conv2, SimpleGate, AdaptiveAvgPool2d).  part: scratch [B, nchunk, C]. */
int ffsr_dw3x3_gate_pool_f32(const float* in, int ldi, const float* w, const float* bias, float* out, int ldo,
                             float* pooled, float* part, int B, int H, int W, int C, int nchunk, void* stream);

/* F.interpolate(mode='bilinear', align_corners=False, size=(Ho, Wo)); out = (accumulate ? out : 0) + mul * value. */
int ffsr_bilinear_f32(const float* in, int ldi, float* out, int ldo, int B, int Hi, int Wi, int Ho, int Wo, int C,
                      float mul, int accumulate, void* stream);
/* F.interpolate(scale_factor=scale, mode='bicubic', align_corners=False) (A = -0.75): nafnet/__init__.py:128-133. */
int ffsr_bicubic_up_f32(const float* in, int ldi, float* out, int ldo, int B, int H, int W, int C, int scale,
                        void* stream);
/* F.avg_pool2d(x, 2, 2): grl mixed_attn_block.py:714-737 (anchors), edge_enhancement.py:196. */
int ffsr_avgpool2_f32(const float* in, int ldi, float* out, int ldo, int B, int H, int W, int C, void* stream);

/* Host-boundary pixel helpers of models/team29_FreqFusionSR/io.py: _uint2tensor4 :100 (uint8 HWC -> float / 255),
 * _tensor2uint :107 (clamp, * 255, round-half-even, uint8), _pad16 :71 (reflect, right/bottom), _unpad :81 and the
 * feature crops :234,245,268 (optionally with the experts' clamp(0,1), expert_loader.py:441,457). */
int ffsr_u8_to_f32(const unsigned char* in, float* out, int ldo, long long M, int C, void* stream);
int ffsr_f32_to_u8(const float* in, int ldi, unsigned char* out, long long M, int C, void* stream);
int ffsr_pad_reflect_f32(const float* in, int ldi, float* out, int ldo, int B, int H, int W, int Hp, int Wp, int C,
                         void* stream);
int ffsr_crop_f32(const float* in, int ldi, float* out, int ldo, int B, int H, int W, int Ho, int Wo, int C,
                  int do_clamp, void* stream);

/* out[b,i,j,:] (+)= scale * in[b, ay_i*i + ay_j*j + cy, ax_i*i + ax_j*j + cx, :] -- the hflip / rot90 index maps of the
 * 8x geometric self-ensemble (scripts/extract_test_tta_cache.py:97-104; generate_fast_submission.py:55-61,250). */
int ffsr_dihedral_f32(const float* in, int ldi, float* out, int ldo, int B, int Hi, int Wi, int Ho, int Wo, int C,
                      int ay_i, int ay_j, int cy, int ax_i, int ax_j, int cx, float scale, int accumulate, void* stream);

/* DRCT (shifted) window attention, fused: softmax(q k^T * scale + bias (+ shift mask)) v.
 * qkv [B*H*W, ldq]: q | k | v, each [heads][C/heads]; bias = relative_position_bias_table [(2ws-1)^2, heads] (ws = 16); roll / window
 * partition / reverse / mask folded into addressing.  variant: 0 = split-bf16 MFMA kernel, 128 queries per workgroup (4: 256 queries per workgroup; both
 * contractions as 3-term bf16 products, fp32 accumulate, ~1e-5 relative: the arithmetic of ffsr_conv2d_bf16x3),
 * 3 (or 1) = exact f32-MFMA kernel with 128 queries per workgroup, 2 = exact kernel with 256 queries per workgroup.
 * Replaces drct_arch.py:175-206 and :376-414. */
int ffsr_window_attn_f32(const float* qkv, int ldq, const float* bias, float* out, int ldo, int B, int H, int W, int C,
                         int heads, int ws, int shift, float scale, int variant, void* stream);

/* GRL 8x8 cosine window attention (mixed_attn_block_efficient.py:77-94,128-165) on the columns
 * [col0, col0 + 3*heads*hd) of qkv; biasT [heads, 64 keys, 64 queries] = 16*sigmoid(CPB-MLP) transposed;
 * logit [heads] = exp(min(logit_scale, ln 100)).  Output written at columns ocol0 + head*hd. */
int ffsr_grl_window_attn_f32(const float* qkv, int ldq, int col0, const float* biasT, const float* logit, float* out,
                             int ldo, int ocol0, int B, int H, int W, int heads, int hd, int shift, void* stream);

/* GRL anchored stripe attention, both hops fused (mixed_attn_block_efficient.py:215-270); anchor [B,H/2,W/2,lda]. */
int ffsr_grl_stripe_attn_f32(const float* qkv, int ldq, int col0, const float* anchor, int lda, const float* bias1T,
                             const float* bias2T, const float* logit1, const float* logit2, float* out, int ldo,
                             int ocol0, int B, int H, int W, int heads, int hd, void* stream);

/* nn.MultiheadAttention core on S sequences of T tokens (T = 9 bands or 4 experts), heads of 16:
 * qkv [S*T, 3E] -> out [S*T, E].  Replaces large_kernel_attention.py:222-233, 385-396 (between in/out proj).
 * p_drop = 0: eval mode.  p_drop in (0, 1) (training; the reference's two attentions use dropout = 0.1, :196, :298):
 * attention-probability dropout with a counter-based mask -- element (sequence, head, query, key) of the draw `seed` is kept
 * iff splitmix64(seed, element index) >> 32 >= p_drop * 2^32, kept probabilities are scaled by 1 / (1 - p_drop).  The mask is
 * a pure function of (seed, index): ffsr_pixel_mha_bwd_f32 regenerates it from the same seed.  (Not torch's Philox stream:
 * bit-parity with the reference's dropout mask is unpinned by construction.) */
int ffsr_pixel_mha_f32(const float* qkv, int ldq, float* out, int ldo, long long S, int T, int E, int heads, float p_drop,
                       long long seed, void* stream);

/* Four-direction selective scan with the dt projection fused (mamba_ssm selective_scan_fn as called at
 * mambair_arch.py:339-369 incl. the direction gather :343-344 and inverse scatter :365-369).
 * u [B, L, ldu]; xdbl [B, L, ldx] = per direction k: dt(R) | B(16) | C(16) at column k*(R+32);
 * dtw [4, Dm, R]; dtb, Dv [4, Dm]; A [4, Dm, 16] (= -exp(A_logs)); y [4][B, L, ldy] per-direction outputs in
 * pixel order; hstate/decay: scratch [B, 4, ceil(L/chunk), Dm, 16] each. */
int ffsr_selective_scan4_f32(const float* u, int ldu, const float* xdbl, int ldx, const float* dtw, const float* dtb,
                             const float* A, const float* Dv, float* y, int ldy, float* hstate, float* decay, int B,
                             int H, int W, int Dm, int R, int d_state, int chunk, void* stream);
/* The same scan with the four per-direction outputs folded into TWO planes: y [2][B, L, ldy]; the directions 0 and 1 are scanned
 * first and write plane 0 / 1, then 2 and 3 are scanned and ADD their outputs to plane 0 / 1 (two half-size launches of each pass
 * in stream order; the scan is ALU-bound, the read-modify-write rides in its memory slack).  Halves what the consumer
 * (ffsr_mamba_norm_gate_pairs_f32) has to read back.  hstate / decay: scratch [B, 2, ceil(L/chunk), Dm, 16] each. */
int ffsr_selective_scan4_pairs_f32(const float* u, int ldu, const float* xdbl, int ldx, const float* dtw, const float* dtb,
                                   const float* A, const float* Dv, float* y, int ldy, float* hstate, float* decay, int B,
                                   int H, int W, int Dm, int R, int d_state, int chunk, void* stream);
/* out = LayerNorm(y0 + y1 + y2 + y3) * silu(z): mambair_arch.py:380-384.  ystride = elements between directions. */
int ffsr_mamba_norm_gate_f32(const float* y, long long ystride, int ldy, const float* z, int ldz, const float* gamma,
                             const float* beta, float eps, float* out, int ldo, int M, int C, void* stream);
/* out / planes = LayerNorm(p0 + p1) * silu(z) over the two pair planes of ffsr_selective_scan4_pairs_f32 (p0 = y0 + y2,
 * p1 = y1 + y3); arguments as ffsr_mamba_norm_gate_planes_f32. */
int ffsr_mamba_norm_gate_pairs_f32(const float* y, long long ystride, int ldy, const float* z, int ldz, const float* gamma,
                                   const float* beta, float eps, float* out, int ldo, void* out_hi, void* out_lo, int ldp,
                                   int M, int C, void* stream);

/* ffsr_mamba_norm_gate_f32 that can also (or only) emit the result as bf16 hi / lo planes [M, ldp] for the out_proj
 * GEMM (ffsr_conv2d_planes); out may be NULL when the planes are given. */
int ffsr_mamba_norm_gate_planes_f32(const float* y, long long ystride, int ldy, const float* z, int ldz,
                                    const float* gamma, const float* beta, float eps, float* out, int ldo, void* out_hi,
                                    void* out_lo, int ldp, int M, int C, void* stream);

/* Phase-2 band split into bands[pixel][9][4] (ldb = 36): multi_domain_frequency.py:146-196 (DCT), :251-299 (DWT,
 * sub [B, Hd, Wd, 16] then upsampled with ffsr_bilinear_f32), :352-385 (FFT, as separable dense DFTs;
 * twW/twH = (cos, sin)(2 pi j / n) tables, mask [H, W/2+1], work = 10 * B*3*H*(W/2+1) floats). */
int ffsr_dct_bands_f32(const float* img, int ldi, const float* D, const float* masks, const float* scale, float* bands,
                       int ldb, int B, int H, int W, void* stream);
int ffsr_dwt_db4_f32(const float* img, int ldi, const float* lo, const float* hi, float* sub, int B, int H, int W,
                     void* stream);
int ffsr_fft_bands_f32(const float* img, int ldi, const float* twW, const float* twH, const float* mask,
                       const float* scale, float* work, float* bands, int ldb, int B, int H, int W, void* stream);

/* gates = sigmoid(T * (raw - (0.7 - 0.5 * diff))) / clamp(sum_e gates + 1e-8, min 0.3): DynamicExpertSelector.forward,
 * enhanced_fusion_v2.py:462-465.  raw [M, 4], diff [M, 1], temperature = device scalar. */
int ffsr_selector_gates_f32(const float* raw, int ldr, const float* diff, int ldd, const float* temperature,
                            float* gates, int ldg, long long M, void* stream);
/* Phase-4 tail (large_kernel_attention.py:410-427): t_lr [B,h,w,ldt] = modulation[i].0 (1x1 128->32) applied at LR
 * (hoisted in front of the bilinear upsample -- both linear); per HR pixel: mod = sigmoid(W2 gelu(bilinear(t_lr)) + b2),
 * out[.., 0..2] = clamp(img * (1 + 0.2 * (mod - 0.5)), 0, 1).  w2 [3, 32], b2 [3]. */
int ffsr_modulate_f32(const float* t_lr, int ldt, const float* w2, const float* b2, const float* img, int ldi,
                      float* out, int ldo, int B, int h, int w, int Hh, int Wh, void* stream);
/* Phases 5b + 5c + 6 fused at HR (enhanced_fusion_v2.py:735-774): enh [.., 12] = 4 enhanced expert images,
 * hier [.., 3] = sigmoid output of the hierarchical fusion, routing [B,h,w,ldr] = routing_lr,
 * fw = freq_weight_conv packed [W1 16x3 | b1 16 | W2 4x16 | b2 4], gates [B,h,w,4], diff [B,h,w,1] at LR.
 * out [.., 4] = (1 - bw) * (0.7 hier + 0.3 freq_fused) + bw * dynamic_fused, channel 3 zeroed.
 * Improvement switches (io.py:186-193 / enhanced_fusion_v2.py:729-774): fw == NULL (multi_resolution_fusion off) -> `hier`
 * is simple_fusion's output and is taken as the fused image; gates == diff == NULL (dynamic_expert_selection off) -> no
 * dynamic blend.  `routing` gives the LR size and may be the LR image itself. */
int ffsr_fusion_route_f32(const float* enh, int lde, const float* hier, int ldh, const float* routing, int ldr,
                          const float* fw, const float* gates, int ldg, const float* diff, int ldd, float* out, int ldo,
                          int B, int h, int w, int Hh, int Wh, void* stream);
/* out = clamp(clamp(sr + gate * strength * edge, 0, 1) + rscale * bilinear(lr), 0, 1): edge_enhancement.py:261-262 and
 * enhanced_fusion_v2.py:788-795.  gate [.., 1] already sigmoid-ed; strength, rscale = device scalars.
 * edge == NULL (edge_enhancement off, :784-786): out = clamp(sr + rscale * bilinear(lr), 0, 1). */
int ffsr_edge_final_f32(const float* sr, int lds, const float* edge, int lde, const float* gate, int ldg,
                        const float* strength, const float* lr, int ldl, const float* rscale, float* out, int ldo, int B,
                        int h, int w, int Hh, int Wh, void* stream);

/* 3x3 / stride 1 / pad 1 convolution with N <= min(4, Cin / 4) output channels and Cin in {8, 16, 32, 64, 128} (wgt [N, 9 * Cin] tap-major
 * fp32, the layout of ffsr_conv2d_f32): out = res * rscale + act(conv + bias) * cscale, exact fp32 FMA.  The RGB / gate heads at
 * HR resolution -- conv_last of the expert tails (drct_arch.py:789, grl_arch.py:517, mambair_arch.py:669), NAFNet's ending
 * (nafnet_arch.py:190), refine.10 (enhanced_fusion_v2.py:576), to_rgb.2 (hierarchical_fusion.py:124), fusion.2 / edge_gate.2 /
 * attn.2 (edge_enhancement.py:88,177), difficulty_net.4 (enhanced_fusion_v2.py:436) -- as the streaming reduction they are
 * instead of a 32-column MFMA tile. */
int ffsr_conv3x3_thin_f32(const float* in, int ldi, const float* wgt, const float* bias, float* out, int ldo, const float* res,
                          int ldr, int B, int H, int W, int Cin, int N, int act, float slope, float cscale, float rscale,
                          void* stream);

/* SSIM map from Gaussian-filtered moments (src/utils/metrics.py:129-186, calculate_ssim_torch); one channel, stride ld. */
int ffsr_ssim_map_f32(const float* mu1, const float* mu2, const float* e11, const float* e22, const float* e12, int ld,
                      float* out, int ldo, long long M, void* stream);

/* uint8 PSNR / SSIM of the evaluation script, utils/utils_image.py:148-189 (cal_psnr_ssim).
 * ffsr_u8_planes: img [H, W, 3] uint8 RGB -> planes [P, H - 2 crop, W - 2 crop] uint8 of the border-cropped window
 * (:162-164): P = 1, the luma of cv2.cvtColor(COLOR_RGB2YCrCb) for 8-bit images (OpenCV's fixed point
 * (4899 R + 9617 G + 1868 B + 2^13) >> 14, :168-169) when y_channel, else P = 3 planes R, G, B (:171-172).
 * ffsr_psnr_ssim_u8: out[0] = mean squared error over the P planes (exact integer sum; PSNR = 20 log10(255 / sqrt(mse)),
 * :175-179), out[1] = skimage.metrics.structural_similarity(data_range=255[, channel_axis=2]) (:183-187): 7x7 uniform
 * window, sample covariance, mean of the map over the pixels whose window lies inside the plane, mean over planes.
 * out / partial are device doubles (partial: 2 * n_partial of scratch). */
int ffsr_u8_planes(const unsigned char* img, int H, int W, int crop, int y_channel, unsigned char* planes, void* stream);
int ffsr_psnr_ssim_u8(const unsigned char* pa, const unsigned char* pb, int P, int Hc, int Wc, double* partial,
                      int n_partial, double* out, void* stream);

/* ---- loss and optimiser side of the cached-feature training step (the backward kernels of the fusion
 * phases follow further down).  Flat fp32 buffers, deterministic two-stage reductions; `partial` is caller-owned scratch of at least
 * 1024 floats (n_partial says how many).
 *
 * loss[0] = loss_scale * mean |clamp(sr, 0, 1) - hr| over M rows x C columns; grad (may be NULL) = d loss / d sr =
 * loss_scale / (M C) * sign(clamp(sr) - hr) where 0 <= sr <= 1, else 0.  Replaces train.py:326-336 (clamp, L1Loss
 * perceptual_loss.py:68-100, the division by accumulation_steps = loss_scale) and their autograd backward. */
int ffsr_l1_clamp_loss_f32(const float* sr, int ldsr, const float* hr, int ldhr, float* grad, int ldg, float* partial,
                           int n_partial, float* loss, long long M, int C, float loss_scale, void* stream);
/* out[0] = sum x[i]^2 (the squared total norm of torch.nn.utils.clip_grad_norm_, train.py:347-352). */
int ffsr_sumsq_f32(const float* x, long long n, float* partial, int n_partial, float* out, void* stream);
/* One fused pass over the flat parameter buffer: gradient clipping (coef = min(1, max_norm / (sqrt(grad_sumsq[0]) + 1e-6));
 * grad_sumsq NULL or max_norm <= 0: none), torch.optim.AdamW's update for step number `step` (>= 1; decoupled weight
 * decay, bias corrections computed on the host in double) and EMAModel.update (ema may be NULL):
 * train.py:347-359, checkpoint_manager.py:349-356. */
int ffsr_adamw_ema_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema, long long n,
                       const float* grad_sumsq, float max_norm, float lr, float beta1, float beta2, float eps,
                       float weight_decay, int step, float ema_decay, void* stream);

/* ================= backward pass of the fusion network (what loss.backward() runs for
 * CompleteEnhancedFusionSR.forward_with_precomputed in train.py:323-336).  Each entry point replaces the autograd node(s)
 * of the cited forward call.  Parameter gradients are ACCUMULATED (dst += value); all reductions are deterministic
 * two-stage sums over caller-owned scratch. */

/* p[0..n) = 0 (hipMemsetAsync): the zero_grad() of train.py:355 and zero-padded channel buffers. */
int ffsr_zero_f32(float* p, long long n, void* stream);

/* nn.Conv2d weight [N, Cin, KH, KW] -> the packed operand of ffsr_conv2d_f32 (dst_f32 [rows, KH*KW*Cp]) and / or of
 * ffsr_conv2d_bf16x3 (bf16 hi / lo planes [rows_pad, ldw], zero padded).  transpose = 0: the forward operator (rows = N,
 * Cp >= Cin channels per tap); transpose = 1: the input-gradient operator of the stride-1 "same" convolution (rows = Cin,
 * Cp >= N channels per tap, taps flipped) -- autograd's conv2d backward-data as a forward convolution of dY.
 * The weights change every optimiser step, so this runs per step instead of once at load. */
int ffsr_pack_conv_f32(const float* w, int N, int Cin, int KH, int KW, int transpose, float* dst_f32, int Cp, void* dst_hi,
                       void* dst_lo, int rows_pad, int ldw, void* stream);
/* depthwise weight [C, 1, KH, KW] -> tap-major [KH*KW, C] (ffsr_dwconv2d_f32's layout); flip = 1: input-gradient operator. */
int ffsr_pack_dwconv_f32(const float* w, int C, int KH, int KW, int flip, float* dst, void* stream);

/* Weight (and bias) gradient of a stride-1 convolution / linear layer on the f32 MFMA:
 * dw [N, Cin, KH, KW] += sum_pix dy[pix, n] * x[pix + (ky - pad_h, kx - pad_w), c];  dbias [N] += sum_pix dy[pix, n]
 * (dbias NULL = no bias).  partial: scratch of partial_floats floats (>= KH*KW*N*Cin + N; more = more pixel splits in flight).  Replaces autograd's conv2d / linear backward-weight for
 * every nn.Conv2d / nn.Linear of the fusion net (e.g. enhanced_fusion_v2.py:569-576, large_kernel_attention.py:134-138). */
int ffsr_conv_wgrad_f32(const float* x, int ldx, const float* dy, int ldy, float* dw, float* dbias, float* partial,
                        long long partial_floats, int B, int H, int W, int Cin, int N, int KH, int KW, int pad_h, int pad_w,
                        void* stream);
/* The same contract with split-bf16 products (hi*hi + hi*lo + lo*hi on the bf16 MFMA, fp32 accumulate: the arithmetic of
 * ffsr_conv2d_bf16x3) where a kernel for the shape exists -- 3-wide layers (KW = 3, pad_w = 1) with N and Cin multiples of 128,
 * i.e. the refine stack's 128 -> 128 convolutions at HR (enhanced_fusion_v2.py:569-576), both operands read through gfx950's
 * transposing LDS read; every other shape runs ffsr_conv_wgrad_f32's exact fp32 kernels. */
int ffsr_conv_wgrad_bf16x3(const float* x, int ldx, const float* dy, int ldy, float* dw, float* dbias, float* partial,
                           long long partial_floats, int B, int H, int W, int Cin, int N, int KH, int KW, int pad_h, int pad_w,
                           void* stream);
/* ffsr_conv_wgrad_bf16x3 with X given as the bf16 hi / lo planes [B*H*W, ldp] that ffsr_conv2d_planes consumed in the forward
 * pass (ldp % 32 == 0, pad channels zero): no fp32 copy of the activation has to exist (the refine stack's GELU outputs,
 * enhanced_fusion_v2.py:569-576, live as planes only).  dY: the fp32 map dy, or (dy NULL) the planes dy_hi / dy_lo [B*H*W, ldq]
 * written by ffsr_act_bwd_planes_f32.  Shapes: the ones the bf16 kernel takes (see above); FFSR_EINVAL else. */
int ffsr_conv_wgrad_bf16x3_planes(const void* x_hi, const void* x_lo, int ldp, const float* dy, int ldy, const void* dy_hi,
                                  const void* dy_lo, int ldq, float* dw, float* dbias, float* partial, long long partial_floats,
                                  int B, int H, int W, int Cin, int N, int KH, int KW, int pad_h, int pad_w, void* stream);
/* Depthwise (groups = C) weight gradient for the kernel shapes 5x5, 1x21, 21x1 (large_kernel_attention.py:58-76) and 3x3.
 * partial: nchunk * KH*KW * C floats. */
int ffsr_dwconv_wgrad_f32(const float* x, int ldx, const float* dy, int ldy, float* dw, float* partial, int nchunk, int B, int H,
                          int W, int C, int KH, int KW, int pad_h, int pad_w, void* stream);

/* dx = (accumulate ? dx : 0) + alpha * dy * act'(ref); ref = the activation's input (from_output 0) or its output
 * (from_output 1; ReLU / LeakyReLU / sigmoid only).  act codes as in the forward library, plus 6 = clamp(., 0, 1)
 * (edge_enhancement.py:260; gradient passes on the closed interval like torch.clamp) and 7 = clamp(., min=slope)
 * (multi_domain_frequency.py:374). */
int ffsr_act_bwd_f32(const float* dy, int ldy, const float* ref, int ldr, float* dx, int ldx, long long M, int C, int act,
                     float slope, int from_output, float alpha, int accumulate, void* stream);
/* The same (no accumulation) with the result written as bf16 hi / lo planes [M, ldp] (C % 32 == 0, ldp == C): the gradient
 * entering a wide layer goes to ffsr_conv2d_planes (input gradient) and ffsr_conv_wgrad_bf16x3_planes without an fp32 copy. */
int ffsr_act_bwd_planes_f32(const float* dy, int ldy, const float* ref, int ldr, void* out_hi, void* out_lo, int ldp, long long M,
                            int C, int act, float slope, int from_output, float alpha, void* stream);
/* out = alpha * sa[0] * a + beta * sb[0] * b  (sa / sb: learnable DEVICE scalars or NULL = 1; b optional): the residual
 * scalings scale1 / scale2 (large_kernel_attention.py:143-148), residual_weight_*, ResBlock.scale, edge_strength,
 * residual_scale, the band scales of multi_domain_frequency.py:192-194,297,385 -- read on the device, no host sync. */
int ffsr_axpby_dev_f32(const float* a, int lda, const float* sa, float alpha, const float* b, int ldb, const float* sb,
                       float beta, float* out, int ldo, long long M, int C, void* stream);
/* out[0] (+)= scale * sum_{m, c} a * (b ? b : 1): gradient of a learnable scalar.  partial: n_partial doubles. */
int ffsr_dot_acc_f32(const float* a, int lda, const float* b, int ldb, long long M, int C, double* partial, int n_partial,
                     float* out, float scale, int accumulate, void* stream);
/* out[c * ostride] (+)= scale * sum_m a[m, c] * (b ? b[m, c] : 1): bias gradients and per-channel scale gradients.
 * partial: nchunk * C floats. */
int ffsr_coldot_acc_f32(const float* a, int lda, const float* b, int ldb, long long M, int C, float* partial, int nchunk,
                        float* out, int ostride, float scale, int accumulate, void* stream);
/* out[m * ldo] (+)= alpha * sum_c a[m, c] * b[m, c]: gradient of a per-pixel gate that was broadcast over the channels
 * (hierarchical_fusion.py:42, edge_enhancement.py:88). */
int ffsr_rowdot_f32(const float* a, int lda, const float* b, int ldb, float* out, int ldo, long long M, int C, float alpha,
                    int accumulate, void* stream);

/* nn.BatchNorm2d in TRAIN mode (large_kernel_attention.py:84,128,131 under model.train()): batch statistics over the M rows,
 * stat [2, C] <- (mean, rstd), scale_shift [2, C] <- the fused affine y = x * scale + shift (apply with ffsr_unary_f32),
 * running_mean / running_var updated in place (momentum, unbiased variance; NULL = skip).
 * partial: nchunk * C floats, sums: 2 * C floats of scratch. */
int ffsr_bn_train_stats_f32(const float* x, int ldx, long long M, int C, const float* gamma, const float* beta, float eps,
                            float momentum, float* partial, int nchunk, float* sums, float* stat, float* scale_shift,
                            float* run_mean, float* run_var, void* stream);
/* its backward: dx (may alias dy), dgamma / dbeta += .  coef: 3 * C floats of scratch. */
int ffsr_bn_train_bwd_f32(const float* x, int ldx, const float* dy, int ldy, float* dx, int lddx, long long M, int C,
                          const float* gamma, const float* stat, float* partial, int nchunk, float* sums, float* coef,
                          float* dgamma, float* dbeta, void* stream);
/* nn.LayerNorm backward (C <= 256; large_kernel_attention.py:190,288-289): dx, dgamma / dbeta +=.
 * partial: 2 * nblock * C floats. */
int ffsr_layernorm_bwd_f32(const float* x, int ldx, const float* gamma, float eps, const float* dy, int ldy, float* dx, int lddx,
                           float* partial, int nblock, float* dgamma, float* dbeta, long long M, int C, void* stream);

/* Adjoints of ffsr_bilinear_f32 / ffsr_avgpool2_f32 (gather form): din (+)= mul * A^T dout. */
int ffsr_bilinear_bwd_f32(const float* dout, int ldo, float* din, int ldi, int B, int Hi, int Wi, int Ho, int Wo, int C, float mul,
                          int accumulate, void* stream);
int ffsr_avgpool2_bwd_f32(const float* dout, int ldo, float* din, int ldi, int B, int H, int W, int C, int accumulate,
                          void* stream);

/* Backward of ffsr_pixel_mha_f32: dqkv [S*T, 3E]; p_drop / seed = the forward call's (the dropout mask is regenerated).
 * scratch: 2 * S * heads * T * T floats. */
int ffsr_pixel_mha_bwd_f32(const float* qkv, int ldq, const float* dout, int ldo, float* dqkv, int lddq, float* scratch,
                           float p_drop, long long seed, long long S, int T, int E, int heads, void* stream);

/* softmax over the C <= 8 channels of every row and its backward (F.softmax(freq_logits, dim=1), enhanced_fusion_v2.py:743). */
int ffsr_softmax_c_f32(const float* x, int ldx, float* y, int ldy, long long M, int C, void* stream);
int ffsr_softmax_c_bwd_f32(const float* y, int ldy, const float* dy, int lddy, float* dx, int ldx, long long M, int C,
                           void* stream);
/* out[c] = sum_e x[3 e + c] * g[e] / (normalize ? sum_e g[e] + 1e-8 : 1), 4 experts x 3 channels: the frequency-guided
 * sum (enhanced_fusion_v2.py:744-747) and the gated mean of the dynamic selection (:761-768); backward: dx (+)=, dg =. */
int ffsr_expert_sum_f32(const float* x, int ldx, const float* g, int ldg, float* out, int ldo, long long M, int normalize,
                        void* stream);
int ffsr_expert_sum_bwd_f32(const float* x, int ldx, const float* g, int ldg, const float* dy, int lddy, float* dx, int lddx,
                            float* dg, int lddg, long long M, int normalize, int accumulate_dx, void* stream);
/* Backward of ffsr_selector_gates_f32: draw [M, 4], ddiff [M, 1], dtemperature[0] +=.  partial: n_partial doubles. */
int ffsr_selector_gates_bwd_f32(const float* raw, int ldr, const float* diff, int ldd, const float* temperature,
                                const float* dgates, int ldg, float* draw, int lddr, float* ddiff, int lddd, double* partial,
                                int n_partial, float* dtemperature, long long M, void* stream);

/* FFT band split, backward w.r.t. the learnable mask (multi_domain_frequency.py:366-383):
 * ffsr_rfft2_ortho_f32: spec [B*3, H, W/2+1] complex = torch.fft.rfft2(img, norm="ortho") (work: as many float2).
 * ffsr_fft_mask_grad_f32: dmask [H, W/2+1] = w_k * sum_{b,c} Re(X conj(Ghat)) with X = xlo + xhi (the masked spectra the
 * forward ffsr_fft_bands_f32 left in its work buffer at float offsets 2 n and 4 n, n = B*3*H*(W/2+1)) and
 * Ghat = rfft2_ortho(dL/dlow - dL/dhigh). */
int ffsr_rfft2_ortho_f32(const float* img, int ldi, const float* twW, const float* twH, float* work, float* spec, int B, int H,
                         int W, void* stream);
int ffsr_fft_mask_grad_f32(const float* xlo, const float* xhi, const float* ghat, float* dmask, int B, int H, int W,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FFSR_H */
