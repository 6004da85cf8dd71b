/* libsrhip -- C ABI of the MI355X (gfx950) SemiReward hot path.
 *
 * The reference (Westlake-AI/SemiReward) is 100 % Python on ATen; it has no FFI of its own.  The
 * "interface each entry point replaces" is therefore the ATen op sequence at the cited reference
 * call site (paths relative to the reference root; K-numbers = SURVEY.md section 2c).
 *
 * Conventions
 *   - plain C: raw DEVICE pointers + sizes, no torch types.  Buffers are borrowed for the call.
 *   - every function enqueues on `stream` (a hipStream_t passed as void*; NULL = default stream) and
 *     returns immediately:  0 = ok, -1 = invalid argument, <= -2 = -(2 + hipError_t) launch failure.
 *     Nothing throws across the boundary; nothing allocates; no host synchronisation.
 *   - bf16 buffers are raw uint16 bit patterns (void*), row-major, 16-byte aligned.
 *   - int64 index tensors keep the reference's dtype (torch.long) so no host-side conversion is needed.
 *   - single thread per process / one process per GPU, like the reference (train.py:344).
 */
#ifndef SRHIP_H
#define SRHIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------
 * Dense contractions (MFMA).  C[M,N] (+)= A[M,K] . B[N,K]^T, bf16 in, fp32 accumulate.
 * Replaces nn.Linear forward/backward of the backbone:
 *   qkv / proj   semilearn/nets/vit/vit.py:93-98, :105          (K3, K5)
 *   fc1 / fc2    semilearn/nets/vit/vit.py:69-75                (K6)
 * Requirements: K % 32 == 0, N % 4 == 0, lda/ldb % 8 == 0, ldc % 4 == 0, 16-byte aligned bases.
 */
enum {
  SRHIP_EPI_BF16 = 0,       /* C(bf16) = acc + bias                                              */
  SRHIP_EPI_GELU_BF16 = 1,  /* C(bf16) = gelu_erf(acc + bias); aux_out(bf16) = acc + bias if set */
  SRHIP_EPI_RESID_F32 = 2,  /* C(f32) = R + row_scale[m / rows_per_sample] * (acc + bias), R = aux_in (f32, ldaux) if set else C
                               (DropPath + residual add, vit.py:164-165)                            */
  SRHIP_EPI_DGELU_BF16 = 3, /* C(bf16) = acc * gelu_erf'(aux_in)                                 */
  SRHIP_EPI_F32 = 4         /* C(f32) = alpha * acc + beta * C                                   */
};
int srhip_gemm_nt(int epilogue, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                  const float* bias, const float* row_scale, int rows_per_sample, const void* aux_in, void* aux_out,
                  int ldaux, float alpha, float beta, void* stream);
/* The tile kernel srhip_gemm_nt picks for a product (host logic only, no launch): the 128 x 128 LDS-DMA ring kernel, the 64 x 64 deep-ring kernel
 * for launches that would leave most CUs idle, or one of the persistent 256-row kernels.  Exported so that a test can pin the decision per shape family
 * (the nn.Linear products of vit.py:93-98,105,69-75 at ViT-S width and of the HF encoders behind bert.py:34 / wave2vecv2.py:44 at D = 768). */
#define SRHIP_GEMM_PLAN_TILE128 0
#define SRHIP_GEMM_PLAN_SMALL64 1
#define SRHIP_GEMM_PLAN_BIG256 2
#define SRHIP_GEMM_PLAN_BIG128 3
#define SRHIP_GEMM_PLAN_BIG2WG 4
#define SRHIP_GEMM_PLAN_PP256 5      /* 256 x 256 x 64 tiles, two wave groups half a phase apart (K % 64 == 0, operands < 2 GiB); else BIG256 = its lockstep predecessor */
int srhip_gemm_nt_plan(int epilogue, int M, int N, int K, float beta);
/* Launches of fewer than n 128 x 128 tiles go to the 64 x 64 deep-ring kernel (default 256 = one round of the chip; K >= 768 products at N >= 768
 * excepted).  The small tiles are the latency choice for a chain of dependent launches that has the chip to itself; a training step whose
 * row-streaming inference launches own most CUs meanwhile sets a low n (fewer, fatter workgroups).  n < 0: query only.  Returns the previous value.
 * Process-wide (one process per GPU, train.py:344), not thread-safe. */
int srhip_gemm_small_max_grid(int n);


/* SRHIP_EPI_RESID_F32 with nn.Dropout on the branch: C(f32)[M,N] = resid (f32, ldresid; NULL: C) + dropout(acc + bias) -- the
 * ``LayerNorm(x + dropout(dense(.)))`` of BertSelfOutput / BertOutput (reached from semilearn/nets/bert/bert.py:34) before the LayerNorm.
 * Mask element index i = m * N + n (ldc == N and N even required); decision per element pair i >> 1 as described at srhip_attn_masked_fwd. */
int srhip_gemm_nt_resid_dropout(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K,
                                const float* bias, const float* resid, int ldresid, unsigned drop_key, unsigned drop_thresh,
                                float drop_scale, void* stream);
/* The same product with the residual taken as LayerNorm(C) of the PRE-LayerNorm sums C holds (in place):
 *   C = (C - ln_mean[m]) * ln_rstd[m] * ln_gamma[n] + ln_beta[n] + dropout(A B^T + bias)
 * -- the post-LN sub-layer chain of the HF encoders (BertSelfOutput / BertOutput behind bert.py:34, Wav2Vec2EncoderLayer behind
 * wave2vecv2.py:44) for the rows without a backward: srhip_postln_fwd(x = NULL) then writes only the bf16 operand and the statistics. */
int srhip_gemm_nt_resid_ln_dropout(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K,
                                   const float* bias, const float* ln_mean, const float* ln_rstd, const float* ln_gamma,
                                   const float* ln_beta, unsigned drop_key, unsigned drop_thresh, float drop_scale, void* stream);

/* Grouped variant for the fp32 weight-gradient products (dW = dY^T X of every block, reference: autograd of the same
 * nn.Linear call sites): C_p = alpha * A_p . B_p^T + beta * C_p for n_problems independent products in ONE launch.
 * desc_dev: DEVICE array; tile_start = running sum of ceil(M/128)*ceil(N/128) over the preceding problems;
 * total_tiles = that sum over all problems.  Same operand requirements as srhip_gemm_nt (K % 32 == 0 ...). */
typedef struct srhip_group_desc {
  const void* A; const void* B; float* C;
  int M, N, K, lda, ldb, ldc, tile_start, pad0, pad1, pad2;
} srhip_group_desc;                      /* 64 bytes */
int srhip_gemm_nt_grouped_f32(const srhip_group_desc* desc_dev, int n_problems, int total_tiles, float alpha, float beta,
                              void* stream);
/* The same launch over 128 x 64 tiles (tile_start / total_tiles count ceil(M / 128) * ceil(N / 64) per problem): products whose N is at most a few
 * dozen columns -- the grouped positional convolution of Wav2Vec2 / HuBERT (16 groups x 48 output channels, K = 6144; the HF
 * Wav2Vec2PositionalConvEmbedding behind wave2vecv2.py:44), forward and input gradient. */
int srhip_gemm_nt_grouped_n64_f32(const srhip_group_desc* desc_dev, int n_problems, int total_tiles, float alpha, float beta,
                                  void* stream);

/* Grouped weight-gradient products from ROW-MAJOR operands (no transposes): for each problem
 *   C[M,N] = alpha * A^T . B + beta * C,  A bf16 [K, M] (lda), B bf16 [K, N] (ldb), C fp32 [M, N] (ldc);  dbias[M] += colsum(A)
 * i.e. dW = dY^T X and db = sum_tokens dY of an nn.Linear (autograd of vit.py:69-75, :95-112) with K = tokens.
 * K is arbitrary (rows past K read as zero); M % 8 == 0, N % 8 == 0, lda/ldb % 8 == 0, operands 16-byte aligned.
 * tile_start / total_tiles as in srhip_group_desc.  dbias may be NULL.
 * flags & SRHIP_TN_ATOMIC: the entry is one K slice of a product the caller split along the token axis (A, B point at the slice's first
 * row, K = its rows): C += alpha * A^T B and dbias += column sums through fp32 atomic adds; beta is ignored for the entry (the sum over
 * the slices is then order-dependent in the last bits, like any split-K reduction). */
#define SRHIP_TN_ATOMIC 1
/* flags & SRHIP_TN_OVERWRITE: the entry is a token slice with a slab of its OWN (C, dbias point into scratch): C = alpha * A^T B and dbias =
 * column sums are written, not accumulated (beta ignored); srhip_slab_reduce_f32 then adds the slabs of a product into its gradient.  An fp32
 * atomic add is a memory transaction of its own on this chip: with 14 M of them per WideResNet backward the sliced weight gradients spent more time
 * in their epilogues than in their products (profiles/r06_wrn_dw_slabs_ab.txt). */
#define SRHIP_TN_OVERWRITE 2
typedef struct srhip_group_tn_desc {
  const void* A; const void* B; float* C; float* dbias;
  int M, N, K, lda, ldb, ldc, tile_start, flags;
} srhip_group_tn_desc;                   /* 64 bytes */
int srhip_gemm_tn_grouped_f32(const srhip_group_tn_desc* desc_dev, int n_problems, int total_tiles, float alpha, float beta,
                              void* stream);
/* dst[i] += sum_{s < n_slabs} src[s * stride + i], i < count, for every entry (16-byte aligned dst / src, stride in elements); block_start = first
 * workgroup of the entry, 1024 elements per workgroup; total_blocks = their sum. */
typedef struct srhip_slab_desc {
  float* dst; const float* src; long long stride;
  int count, n_slabs, block_start, pad0;
} srhip_slab_desc;                       /* 40 bytes */
int srhip_slab_reduce_f32(const srhip_slab_desc* desc_dev, int n, int total_blocks, void* stream);
/* The same products through 256 x 256 output tiles (persistent workgroups, 64-token K-tiles, two wave groups half a phase apart): the kernel for
 * the wide layers' weight gradients (D = 768: autograd of the HF encoder Linears behind semilearn/nets/bert/bert.py:34 and
 * wave2vecv2/wave2vecv2.py:44; 8192 tokens per step) and for ViT-S tables whose problems 256-tiles cover well.  Same descriptor, same
 * arithmetic (bf16 operands, fp32 accumulation, dbias += column sums of A) -- but tile_start / total_tiles count 256 x 256 tiles:
 * ceil(M / 256) * ceil(N / 256) per entry; K >= 1 for every entry (the refill cursor runs one K-tile ahead of the multiplying waves).  The walk is static (workgroup w takes tiles w, w + 256, ...): the caller balances a short last
 * round by handing the last entries over as SRHIP_TN_ATOMIC token slices (ops.make_group_tn_desc(tile=256) does). */
int srhip_gemm_tn_grouped_pp_f32(const srhip_group_tn_desc* desc_dev, int n_problems, int total_tiles, float alpha, float beta,
                                 void* stream);

/* Fused attention, head_dim 64.  qkv bf16 [B*N, 3*H*64] as written by the qkv Linear; out bf16 [B*N, H*64];
 * lse fp32 [B,H,N] (NULL when no backward is needed).  Replaces vit.py:100-104 (K4).  N <= 512. */
int srhip_attn_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, float scale, void* stream);
/* dqkv bf16 [B*N, 3*H*64]; delta_ws fp32 [B,H,N] scratch.  Autograd backward of vit.py:100-104.  N <= 512. */
int srhip_attn_bwd(const void* qkv, const void* out, const void* d_out, const float* lse, void* dqkv, float* delta_ws,
                   int B, int N, int H, float scale, void* stream);

/* Padding-aware variants for the BERT / Wav2Vec2 encoders (semilearn/nets/bert/bert.py:34 -> transformers BertSelfAttention;
 * wave2vecv2/wave2vecv2.py:44 -> Wav2Vec2Attention): key_len int32 [B] = number of valid (non-padding) keys of every sequence of a
 * right-padded batch (NULL: all N), equivalent to the additive -inf attention mask; every query row is computed.  drop_*: train-mode
 * nn.Dropout on the probabilities from the counter-based generator of all dropout sites: ONE hash h = fmix32(pair * 0x9E3779B1 + drop_key)
 * decides TWO neighbouring elements -- the even one is kept iff (h & 0xFFFF) >= drop_thresh >> 16, the odd one iff (h >> 16) is; kept values
 * * drop_scale; drop_thresh == 0: none (the hash, two quarter-rate multiplies, was 74 % of this forward's vector-ALU work at N = 512;
 * oracle/bert_ref.keep_mask restates it).  At THIS site the pairs are taken within a query row: pair = ((b*H + h)*N + q) * ceil(N / 2) +
 * (key >> 1); at every other site (srhip_gemm_nt_resid_dropout, embeddings, post-LN, pooling) pair = i >> 1 of the row-major element index i.
 * N <= 512 forward AND backward (N > 288: the forward walks two query tiles per wave through 128-key blocks, attn_fwd_pair_kernel; the V
 * fragments of the dQ pass come from L2 instead of LDS).  B*H*N*N < 2^32. */
int srhip_attn_masked_fwd(const void* qkv, void* out, float* lse, const int* key_len, int B, int N, int H, float scale,
                          unsigned drop_key, unsigned drop_thresh, float drop_scale, void* stream);
int srhip_attn_masked_bwd(const void* qkv, const void* out, const void* d_out, const float* lse, void* dqkv, float* delta_ws,
                          const int* key_len, int B, int N, int H, float scale, unsigned drop_key, unsigned drop_thresh,
                          float drop_scale, void* stream);

/* nn.LayerNorm over the last dim (vit.py:135,150,268; eps 1e-6 :222) (K2).  x fp32 [M,D] -> out bf16 [M,D];
 * mean/rstd fp32 [M] are written when non-NULL (needed by the backward).  D in {128, 384, 768}. */
int srhip_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, void* out, float* mean,
                        float* rstd, int M, int D, void* stream);
/* dx (fp32) += LN'(dy bf16); dgamma/dbeta (fp32) += column sums (atomic). */
int srhip_layernorm_bwd(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                        float* dx, float* dgamma, float* dbeta, int M, int D, void* stream);
/* ... and also out_bf16[m, :] = bf16(row_scale[m / rows_per_sample] * dx[m, :]) of the UPDATED dx: the DropPath-scaled output gradient of the
 * next branch (vit.py:164-165 backward), i.e. srhip_layernorm_bwd + srhip_cast_scale_rows in one launch.  row_scale NULL = 1. */
int srhip_layernorm_bwd_cast(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* dx, float* dgamma,
                             float* dbeta, void* out_bf16, const float* row_scale, int rows_per_sample, int M, int D, void* stream);
/* The same with the column sums spread over n_rep partial copies, part fp32 [n_rep][2][D] (copy = workgroup % n_rep; [.][0] dgamma, [.][1]
 * dbeta): the 2 * D words of one LayerNorm are otherwise hit by every workgroup of the launch, and same-address device-scope atomics
 * serialise (measured: 11 of 19 us at M = 4112).  out_bf16 may be NULL.  srhip_ln_grad_reduce adds the copies of n_ln LayerNorms
 * (part [n_ln][n_rep][2][D]) to their dgamma / dbeta (torch.autograd's accumulation into .grad of vit.py:135,150) and clears them:
 * one launch per step, after the last LayerNorm backward. */
int srhip_layernorm_bwd_part(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* dx, float* part,
                             int n_rep, void* out_bf16, const float* row_scale, int rows_per_sample, int M, int D, void* stream);
typedef struct srhip_ln_reduce_desc { float* dgamma; float* dbeta; } srhip_ln_reduce_desc;   /* 16 bytes */
int srhip_ln_grad_reduce(const srhip_ln_reduce_desc* desc_dev, float* part, int n_ln, int n_rep, int D, void* stream);

/* Fused MLP half of a transformer block on the fp32 residual stream (x -> x_out, both [M, D]; x_out may equal x):
 *   x_out = x + row_scale[m / rows_per_sample] * ( fc2( GELU( fc1( LayerNorm(x) ) ) ) + b2 )
 * = Block.forward's second residual (vit.py:165) with Mlp.forward (vit.py:69-75), norm2 (vit.py:150) and DropPath
 * (row_scale, NULL = 1).  W1 bf16 [Hd, D], W2 bf16 [D, Hd], biases / LN affine fp32.  The hidden activation stays in registers.
 * save_rows > 0: the first save_rows rows carry a gradient (they lead a mixed batch); for them the kernel also writes what the
 * hand-written backward needs -- norm2 output (save_ln2 bf16 [save_rows, D]), fc1 pre-activation and GELU output (save_pre,
 * save_h bf16 [save_rows, Hd]) and the LayerNorm statistics (save_mean, save_rstd fp32 [save_rows]) -- exactly the values
 * layernorm_fwd + gemm_nt(GELU, aux_out) would have stored.
 * D == 384 (ViT-S width), Hd % 128 == 0, Hd <= 4096.  Same rounding points as layernorm_fwd + gemm_nt(GELU) + gemm_nt(RESID). */
int srhip_mlp_fused(const float* x, float* x_out, const float* ln_gamma, const float* ln_beta, float eps, const void* W1,
                    const float* b1, const void* W2, const float* b2, const float* row_scale, int rows_per_sample, int save_rows,
                    void* save_ln2, void* save_pre, void* save_h, float* save_mean, float* save_rstd, int M, int D, int Hd,
                    void* stream);
/* The same with the attention output projection and the first residual of the block in front (rows without a backward):
 *   x1 = x + row_scale1 * (ao Wp^T + bp);  x_out = x1 + row_scale2 * (fc2(GELU(fc1(LayerNorm(x1)))) + b2)
 * -- vit.py:105-106 (proj) + :163 (drop_path1, residual) + :165 in ONE launch; replaces srhip_gemm_nt(EPI_RESID_F32) + srhip_mlp_fused.
 * ao bf16 [M, D] (attention output, heads concatenated), Wp bf16 [D, D]; x_out may alias x.  The residual stream makes no round trip inside
 * the launch (x is the start value of the accumulators, x1 stays in them), so the DropPath factors act on the bf16 B operands of the products:
 * row_scale2 on the GELU output before its (single) rounding; row_scale1 on ao -- ao_scaled != 0 says the launch that wrote ao applied it
 * before rounding (srhip_attn_block_fused out_scale: one rounding, as in the unfused path), 0 has it applied here (a second rounding of ao).
 * ln_next (bf16 [M, D], may be NULL): LayerNorm(x_out) with (next_gamma, next_beta) = the NEXT block's norm1 -- the operand of its
 * srhip_attn_block_fused -- written by the same launch (replaces that block's srhip_layernorm_fwd). */
/* GELU inside srhip_mlp_fused_proj: x * Phi(x) with Phi = 1/2 + xc R(xc^2), xc = x clamped to +-4.252893, R a degree-8 polynomial (no
 * v_rcp / v_exp: the kernel's main loop is bound by the vector ALU).  |result - exact-erf GELU| <= max(7e-5, 5e-6 |x|), below half a bf16
 * quantum of the result wherever |GELU| >= 0.03; the result is rounded to bf16 right after.  srhip_gelu_eval evaluates both forms of the
 * library (y_erf: the 1.5e-7 form every other path uses) on n values. */
int srhip_gelu_eval(const float* x, float* y_erf, float* y_poly, int n, void* stream);
int srhip_mlp_fused_proj(const float* x, float* x_out, const void* ao, const void* Wp, const float* bp, const float* row_scale1,
                         int ao_scaled, const float* ln_gamma, const float* ln_beta, float eps, const void* W1, const float* b1, const void* W2,
                         const float* b2, const float* row_scale2, int rows_per_sample, void* ln_next, const float* next_gamma,
                         const float* next_beta, int M, int D, int Hd, void* stream);

/* Fused qkv projection + attention of a ViT block for rows without a backward: ao = softmax(q k^T * scale) v with [q | k | v] = xn Wqkv^T +
 * bqkv -- Attention.forward up to the output projection (semilearn/nets/vit/vit.py:93-104) on the norm1 output (:163) in ONE launch, one
 * workgroup per image; replaces srhip_gemm_nt(qkv) + srhip_attn_fwd for inference rows (the [M, 3D] qkv activation never reaches HBM).
 * xn_bf16 [B*N, D]: the rows normalised by srhip_layernorm_fwd; Wqkv bf16 [3D, D] (q | k | v rows, head-major); out bf16 [B*N, D].
 * Built for D = 384, H = 6 and N in {257, 197} (ViT-S/2 at 32x32, ViT-S/16 at 224x224): srhip_attn_block_supported() tells, anything else
 * is an argument error.  N = 257 = 16 token tiles + one token: qkv_extra bf16 [B, 3D] = q | k | v of token 256 of every image (one
 * srhip_gemm_nt over the B rows xn_bf16 + 256 * D with lda = N * D) must be given; ignored for N = 197.
 * out_scale fp32 [B] or NULL: factor on every output row of image b, applied before the bf16 rounding -- the DropPath factor of the attention
 * branch (vit.py:163) when the consumer is srhip_mlp_fused_proj(ao_scaled = 1). */
int srhip_attn_block_supported(int N, int D, int H);
int srhip_attn_block_fused(const void* xn_bf16, const void* Wqkv, const float* bqkv, const void* qkv_extra, void* out, const float* out_scale,
                           int B, int N, int D, int H, float scale, void* stream);

/* PatchEmbed conv (kernel = stride = ps) + cls token + pos_embed (vit.py:39-44, :277-280) (K1).
 * img fp32 [*, C, HW, HW]; img_index int32 [B] maps batch row -> image (NULL = identity; lets the K+1 passes of
 * one SemiReward step share one copy of the images); x fp32 [B, N, D], N = (HW/ps)^2 + 1.  C*ps*ps <= 64. */
int srhip_patch_embed_fwd(const float* img, const int* img_index, const float* Wp, const float* bp, const float* cls,
                          const float* pos, float* x, int B, int C, int HW, int ps, int D, void* stream);
/* all outputs are accumulated (+=). */
int srhip_patch_embed_bwd(const float* dx, const float* img, const int* img_index, float* dWp, float* dbp, float* dcls,
                          float* dpos, int B, int C, int HW, int ps, int D, void* stream);
/* The same without atomics on the filter: per-(token chunk, image) partial sums in ws (srhip_patch_embed_bwd_ws_floats(...) floats), folded
 * by a second launch in a fixed order (deterministic; the one-launch form ends every workgroup in D * (K + 1) same-address atomics). */
long srhip_patch_embed_bwd_ws_floats(int B, int C, int HW, int ps, int D);
int srhip_patch_embed_bwd_ws(const float* dx, const float* img, const int* img_index, float* dWp, float* dbp, float* dcls,
                             float* dpos, float* ws, int B, int C, int HW, int ps, int D, void* stream);

/* Large-patch PatchEmbed (C * ps * ps > 64, e.g. ViT-S/16 at 224 x 224, vit.py:358-371): the conv of vit.py:39-44 as a GEMM.
 *   patch_im2col : col[b * Np + p][(c,i,j)] = img[img_index[b]][c][py*ps+i][px*ps+j] as bf16 (ps even); then
 *                  srhip_gemm_nt(EPI_F32): tok[B*Np, D] = col . Wp^T with Wp = patch_embed.proj.weight viewed [D, C*ps*ps];
 *   patch_assemble: x[b,0,:] = cls + pos[0], x[b,1+p,:] = tok[b*Np+p,:] + bp + pos[1+p]                       (vit.py:277-280)
 *   patch_grad_operands (backward): dpos += sum_b dx[b,:,:], dcls += sum_b dx[b,0,:], and the patch-token rows of dx as bf16
 *                  [B*Np, D] -- the A operand of srhip_gemm_tn_grouped_f32 with B = col: dWp += dx_tok^T col, dbp += colsum dx_tok. */
int srhip_patch_im2col(const float* img, const int* img_index, void* out, int B, int C, int HW, int ps, void* stream);
int srhip_patch_assemble(const float* tok, const float* bp, const float* cls, const float* pos, float* x, int B, int Np, int D,
                         void* stream);
int srhip_patch_grad_operands(const float* dx, void* dx_tok_bf16, float* dpos, float* dcls, int B, int Np, int D, void* stream);

/* Final norm on the cls token, global_pool='token', classifier head (vit.py:282, :296-305) (K7).
 * feat fp32 [B,D], logits fp32 [B,C]; xhat [B,D] / rstd [B] saved for the backward when non-NULL. */
int srhip_cls_head_fwd(const float* x, const float* gamma, const float* beta, float eps, const float* Wh, const float* bh,
                       float* feat, float* logits, float* xhat, float* rstd, int B, int N, int D, int C, void* stream);
/* The same, ALSO (or only: feat / logits may then be NULL) writing image b's outputs to row out_rows[b] of feat_all [*, D] / logits_all [*, C]:
 * the step's buffers over all (pass, image) rows, filled by the launch trains directly instead of by an index_copy_ each. */
int srhip_cls_head_fwd_scatter(const float* x, const float* gamma, const float* beta, float eps, const float* Wh, const float* bh,
                               float* feat, float* logits, float* xhat, float* rstd, float* feat_all, float* logits_all,
                               const long long* out_rows, int B, int N, int D, int C, void* stream);
/* dx[b,0,:] = ...(rows other than the cls row are left untouched: zero dx first); dWh/dbh/dgamma/dbeta +=.
 * dx == NULL: only the head weight / bias gradients (they sum over ALL images of the backward); dWh == NULL: only dx and the final-norm affine
 * gradients (atomic adds) of the B images handed in -- the two halves of a backward whose images are back-propagated in separate row chains. */
int srhip_cls_head_bwd(const float* dlogits, const float* Wh, const float* gamma, const float* feat, const float* xhat,
                       const float* rstd, float* dx, float* dWh, float* dbh, float* dgamma, float* dbeta, int B, int N,
                       int D, int C, void* stream);

/* glue feeding the GEMM: fp32 -> bf16 with optional per-sample (DropPath) scale; transposes with zero padding,
 * optional exact-erf GELU, optional column sums (bias gradients, atomic +=); flat cast. */
int srhip_cast_scale_rows(const float* x, const float* scale, int rows_per_sample, void* out, long M, int D, void* stream);
int srhip_transpose_to_bf16(const void* in, int in_is_f32, int ld_in, void* out, int ld_out, int M, int Mp, int C,
                            int apply_gelu, float* colsum, void* stream);
/* Batched srhip_transpose_to_bf16 (no column sums): one launch over all 64x64 tiles of n transposes.
 * tile_start = running sum of ceil(Mp/64) * (C/64) over the preceding entries. */
typedef struct srhip_transpose_desc {
  const void* in; void* out;
  int in_is_f32, ld_in, ld_out, M, Mp, C, apply_gelu, tile_start;
  long pad;
} srhip_transpose_desc;                  /* 56 bytes */
int srhip_transpose_batched(const srhip_transpose_desc* desc_dev, int n, int total_tiles, void* stream);
int srhip_cast_f32_bf16(const float* x, void* out, long n, void* stream);
/* timm DropPath (vit.py:148,161): out[depth,2,B] = Bernoulli(1-p_l)/(1-p_l), counter-based RNG. */
int srhip_droppath_fill(float* out, const float* probs, int depth, int B, unsigned long long seed, void* stream);
/* The same draw, but only (and in the order of) the columns cols[0 .. n_cols): out [depth, 2, n_cols], out[l, j, q] = table[l, j, cols[q]].
 * One launch lays the table out in the order of the step's launch trains (gradient rows | read rows | deferred rows), each of which then
 * takes a contiguous column slice -- instead of an index_select per train on the step's critical path. */
int srhip_droppath_fill_cols(float* out, const float* probs, const long long* cols, int depth, int B, int n_cols, unsigned long long seed,
                             void* stream);

/* ---------------------------------------------------------------------------------------------
 * Score filter (K8, K9, K11, K12, K13).
 * srhip_row_max: softmax (compute_prob, semilearn/core/algorithmbase.py:332-333) + torch.max / argmax
 *   (semilearn/algorithms/hooks/pseudo_label.py:40, srflexmatch/utils.py:50).  in fp32 [B,C] = logits
 *   (in_is_probs = 0) or probabilities (1); probs_out optional. */
int srhip_row_max(const float* in, int in_is_probs, float* probs_out, float* max_probs, long long* max_idx, int B, int C,
                  void* stream);
/* The same on rows read in place from a [groups, rows, C] buffer: input row r = in + (r / rows_per_group) * group_stride +
 * (r % rows_per_group) * C (elements) -- the weak rows of every pass inside the step's logits (srflexmatch.py:75-104 takes them pass by pass
 * out of separate forward calls); outputs dense. */
int srhip_row_max_strided(const float* in, int in_is_probs, float* probs_out, float* max_probs, long long* max_idx, int B, int C,
                          int rows_per_group, long long group_stride, void* stream);
/* FlexMatchThresholdingHook.masking + update (semilearn/algorithms/srflexmatch/utils.py:24-63).  State on device:
 * selected_label int64 [ulb_dest_len] (-1 = unused), classwise_acc fp32 [C], hist int32 [C+1] (bin C = unused).
 * idx_ulb must be unique within the batch (reference sampler property). */
int srhip_flexmatch_mask(const float* max_probs, const long long* max_idx, const long long* idx_ulb, float p_cutoff,
                         long long* selected_label, int* hist, float* classwise_acc, float* mask, int B, int C,
                         int ulb_dest_len, int thresh_warmup, void* stream);
/* n_pass consecutive masking calls on the same idx_ulb (the data_generator passes of one step, srflexmatch.py:75-104) in one launch, in
 * order: max_probs / max_idx / mask are [n_pass, B]; identical to n_pass srhip_flexmatch_mask calls. */
int srhip_flexmatch_mask_passes(const float* max_probs, const long long* max_idx, const long long* idx_ulb, float p_cutoff,
                                long long* selected_label, int* hist, float* classwise_acc, float* mask, int n_pass, int B, int C,
                                int ulb_dest_len, int thresh_warmup, void* stream);
/* bit 0: an idx_ulb entry outside [0, ulb_dest_len) reached srhip_flexmatch_mask[_passes] since the last reset (the reference raises
 * IndexError from the index_put at srflexmatch/utils.py:59; here the entry is skipped).  SYNCHRONISES the stream. */
int srhip_index_error(int* bits_out, int reset, void* stream);
int srhip_flexmatch_rebuild_hist(const long long* selected_label, int* hist, int ulb_dest_len, int C, void* stream);
/* FixedThresholdingHook.masking (semilearn/algorithms/hooks/masking.py:42-57). */
int srhip_fixed_mask(const float* max_probs, float p_cutoff, float* mask, int B, void* stream);
/* FreeMatchThresholdingHook (semilearn/algorithms/freematch/utils.py:24-66), split so that data-parallel ranks can all-reduce
 * the sufficient statistics instead of all-gathering probabilities:
 *   stats : colsum[c] = sum_i probs[i,c], hist[c] = #{i : argmax_i = c} of the LOCAL batch;
 *   update: time_p / p_model / label_hist EMA from the GLOBAL statistics (maxp_all [n_all] = gathered max-probs, colsum, hist),
 *           then mask[i] = max_probs[i] >= time_p * p_model[idx_i] / max(p_model) for the local rows.  n_all <= 1024. */
int srhip_freematch_stats(const float* probs, const long long* max_idx, float* colsum, float* hist, int B, int C, void* stream);
int srhip_freematch_update(const float* maxp_all, int n_all, const float* colsum, const float* hist, const float* max_probs,
                           const long long* max_idx, float* time_p, float* p_model, float* label_hist, float* mask, int B, int C,
                           float momentum, float one_minus_momentum, int use_quantile, int clip_thresh, void* stream);
/* SoftMatch.  Distributed ranks all-reduce the column sums / gather the max-probs (as for FreeMatch) and call these.
 * distalign: DistAlignEMAHook.dist_align (semilearn/algorithms/hooks/dist_align.py:26-56): p_model <- EMA of colsum_ulb / n_ulb (first call:
 *   plain mean; *inited is a device flag, 0 before the first call), p_target <- EMA of colsum_lb / n_lb when colsum_lb != NULL
 *   (p_target_type 'model'); aligned[i,:] = probs[i,:] * (p_target + 1e-6) / (p_model + 1e-6), renormalised; max_probs / max_idx of it.
 * softmatch_mask: SoftMatchWeightingHook.update + masking (semilearn/algorithms/srsoftmatch/utils.py:32-76, per_class False): mu_var[0..1]
 *   <- EMA of mean / unbiased variance of maxp_all [n_all] (.item() double arithmetic reproduced), mask[i] = exp(-clamp(p_i - mu, max 0)^2 /
 *   (2 var / n_sigma^2)). */
int srhip_distalign(const float* probs, const float* colsum_ulb, int n_ulb, const float* colsum_lb, int n_lb, float* p_model,
                    float* p_target, int* inited, double momentum, float* aligned, float* max_probs, long long* max_idx, int B, int C,
                    void* stream);
int srhip_softmatch_mask(const float* maxp_all, int n_all, const float* max_probs, float* mu_var, double momentum, int n_sigma, float* mask,
                         int B, void* stream);
/* entropy_loss (semilearn/algorithms/srfreematch/srfreematch.py:16-44) forward + analytic backward on the masked rows of the strong
 * logits; loss = 0 and no gradient when the mask is empty (:216-219).  ws: B*C floats.  accumulate != 0: dlogits += . */
int srhip_freematch_entropy(const float* logits, const float* mask, const float* p_model, const float* label_hist, float grad_scale,
                            float* loss_out, float* dlogits, float* ws, int B, int C, int accumulate, void* stream);
/* mask2 = (reward >= reward.mean()) per independent group of B rows (srflexmatch.py:100-101) (K11).
 * mean_in (optional, [groups]) overrides the local mean: data-parallel global-threshold extension. */
int srhip_reward_mask2(const float* reward, float* mask2, float* mean_out, const float* mean_in, int groups, int B,
                       void* stream);
/* ce_loss / consistency_loss forward + analytic backward (semilearn/core/criterions/cross_entropy.py:11-31,
 * consistency.py:38-45): loss = mean_B(nll*mask*mask2); dlogits = grad_scale*(softmax-onehot)*mask*mask2/B. */
int srhip_masked_ce(const float* logits, const long long* targets, const float* mask, const float* mask2, float grad_scale,
                    float* loss_out, float* dlogits, int B, int C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Rewarder / Generator (K10, K14, K15).  params: flat fp32 block in named_parameters() order.
 * Rewarder.forward (semilearn/algorithms/semireward/semireward.py:52-72) for G independent groups of B rows
 * (each group has its own softmax over its 2B rows): feats [G*B,F], labels int64 [G*B] -> reward [G*B].
 * ws: fp32 scratch of srhip_rewarder_ws_floats(G,B) floats; save_for_bwd needs G == 1. */
long srhip_rewarder_param_count(int F, int L);
long srhip_rewarder_ws_floats(int G, int B);
long srhip_generator_param_count(int F);
/* params_t: transposed copies of the 2-D weights, streamed coalesced by the forward kernels; size
 * srhip_rewarder_t_floats(F) / srhip_generator_t_floats(F); refresh with *_prepare after every parameter change. */
long srhip_rewarder_t_floats(int F);
long srhip_generator_t_floats(int F);
int srhip_rewarder_prepare(const float* params, float* params_t, int F, int L, void* stream);
int srhip_generator_prepare(const float* params, float* params_t, int F, void* stream);
int srhip_rewarder_fwd(const float* params, const float* params_t, const float* feats, const long long* labels, float* reward,
                       float* ws, int G, int B, int F, int L, int save_for_bwd, void* stream);
/* The same with the feature rows of group g at feats + g * feat_group_stride (elements; >= B * F): the weak-row block of every pass read in
 * place from the step's [passes, batch, F] feature buffer.  Both entries run as ONE launch when a group is one row tile (B <= 8, the
 * reference batch), two otherwise (the batch softmax over a group's 2B attention logits, semireward.py:60-62, is the only dependency
 * between workgroups).  max_reward_inout (device scalar or NULL; G == 1 and B <= 8 only): *max_reward = max(*max_reward, mean(reward)) in
 * the same launch -- `reward.mean()` and the running maximum of srflexmatch.py:166-170. */
int srhip_rewarder_fwd_strided(const float* params, const float* params_t, const float* feats, long long feat_group_stride,
                               const long long* labels, float* reward, float* ws, float* max_reward_inout, int G, int B, int F, int L,
                               int save_for_bwd, void* stream);
/* Gradient of MSE(r,1) + MSE(r,target) w.r.t. every rewarder parameter (srflexmatch.py:183-190 / :198-205);
 * grads is overwritten; losses[0..1] = (generator_loss, rewarder_loss) when non-NULL. */
int srhip_rewarder_bwd(const float* params, const float* feats, const long long* labels, const float* target, float* ws,
                       float* grads, float* losses, int B, int F, int L, void* stream);
/* Generator.forward (semireward.py:21-24) and the .long() cast (srflexmatch.py:158-159). */
int srhip_generator_fwd(const float* params, const float* params_t, const float* x, float* out, long long* label, int B, int F,
                        void* stream);
/* (cosine_similarity_n(one_hot, one_hot)+1)/2 (semireward.py:130-139, srflexmatch.py:180-182): 1.0 / 0.5. */
int srhip_sr_target(const long long* gen, const long long* ref, float* target, int B, int num_classes, void* stream);
/* Label-range errors of the launches above since the last reset.  The reference raises from nn.Embedding / F.one_hot when a (generated or
 * pseudo) label is outside [0, label_dim) (semireward.py:57, srflexmatch.py:180-181); the kernels stay memory safe (row 0 is read, the
 * gradient scatter is skipped), set a bit in a device word and the host raises when it reads it here (SYNCHRONISES the stream):
 * bit 0 embedding lookup, bit 1 embedding-gradient scatter, bit 2 generator output NaN / not representable as int64, bit 3 a label outside
 * [0, num_classes) in srhip_sr_target (F.one_hot; num_classes <= 0 disables that check). */
int srhip_label_error(int* bits_out, int reset, void* stream);
/* torch.optim.Adam step on a flat block (srflexmatch.py:54, :192-193). */
/* "_dyn" entry points: the scalars that change from step to step -- the scheduler's lr factor and Adam's bias corrections (param_update.py:33-40,
 * build.py:227-251), the DropPath seed (vit.py:148,161) -- are read from DEVICE memory written by the host before the step, so that the launches
 * of a training step can be captured once in a HIP graph and replayed (semireward_amd/core/stepgraph.py).  Same arithmetic as the by-value forms.
 *   srhip_adamw_flat_dyn: dyn fp32 [3] = {lr_factor, 1 - beta1^step, sqrt(1 - beta2^step)};  srhip_adam_flat_dyn: dyn fp32 [2] = the two corrections;
 *   srhip_droppath_fill_cols_dyn: seed = *seed_dev + seed_offset. */
int srhip_adam_bias_corrections(float beta1, float beta2, int step, float* out2_host);   /* host helper: the two fp32 values of a dyn block */
int srhip_adamw_flat_dyn(float* p, float* g, float* m, float* v, void* p_bf16, float* ema, const int* chunk_table, int n_chunks,
                         const float* lr_t, const float* wd_t, const float* dyn, float beta1, float beta2, float eps, double ema_m,
                         float grad_scale, const float* clip_coef, int zero_grad, void* stream);
int srhip_adam_flat_dyn(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps, const float* dyn,
                        void* stream);
int srhip_droppath_fill_cols_dyn(float* out, const float* probs, const long long* cols, int depth, int B, int n_cols,
                                 const unsigned long long* seed_dev, unsigned long long seed_offset, void* stream);
int srhip_adam_flat(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                    int step, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Backbone optimizer (K16, K17): ParamUpdateHook (semilearn/core/hooks/param_update.py:33-40) with the AdamW
 * param groups of semilearn/core/utils/build.py:193-224 + semilearn/nets/utils.py:143-204, fused with the bf16
 * operand refresh, EMA.update (semilearn/core/utils/misc.py:152-155) and model.zero_grad().
 * chunk_table int32 [n_chunks][4] = (offset, length, tensor id, 0); lr_t / wd_t fp32 per tensor;
 * grad_scale multiplies g first (1/world_size after a data-parallel SUM all-reduce); clip_coef (device, 1 float, may be NULL) multiplies
 * it once more: the clip_grad_norm_ coefficient of this step.  ema (may be NULL) <- (1 - ema_m) * p_new + ema_m * ema, op for op as
 * semilearn/core/utils/misc.py:152-155 (ema_m is a double because the reference forms 1 - decay in double before rounding). */
int srhip_adamw_flat(float* p, float* g, float* m, float* v, void* p_bf16, float* ema, const int* chunk_table, int n_chunks,
                     const float* lr_t, const float* wd_t, float lr_factor, float beta1, float beta2, float eps, int step,
                     double ema_m, float grad_scale, const float* clip_coef, int zero_grad, void* stream);
/* torch.nn.utils.clip_grad_norm_(model.parameters(), clip_grad) of ParamUpdateHook (semilearn/core/hooks/param_update.py:34-35) on the flat
 * gradient block: coef_out[0] = min(1, max_norm / (|| pre_scale * g ||_2 + 1e-6)), coef_out[1] = that norm; ws = srhip_clip_grad_ws_floats()
 * floats.  The gradients are NOT rewritten: the optimizer launch takes coef_out as ``clip_coef``. */
int srhip_clip_grad_ws_floats(void);
int srhip_clip_grad_coef(const float* g, long long n, float pre_scale, float max_norm, float* ws, float* coef_out, void* stream);

/* ---- post-LN transformer encoder glue (BERT / Wav2Vec2 backbones; semilearn/nets/bert/bert.py, wave2vecv2/wave2vecv2.py and the HF modules
 * they call).  D in {128, 384, 768}; dropout arguments as in srhip_gemm_nt_resid_dropout (one 32-bit decision per element) with element index = row * D + column.
 *   embed_ln_fwd : BertEmbeddings -- x = dropout(LayerNorm(word[ids[seq][p]] + pos[p] + type0)) for row (b, p) of a [B, L] batch; ids int64
 *                  [*, ld_ids], seq_index int32 [B] picks the row of ids (NULL: identity); x fp32 and bf16 [B*L, D]; mean/rstd [B*L] optional
 *   embed_ln_bwd : its backward from dy = d/dx: dword[id] += (rows with id == pad_id excluded: nn.Embedding(padding_idx)), dpos[p] +=,
 *                  dtype0 +=, dgamma +=, dbeta +=  (atomic)
 *   postln_fwd   : x = LayerNorm(y) as fp32 (next residual) and bf16 (next GEMM operand); x may alias y
 *   postln_bwd   : dy = d/d(LayerNorm output) -> dx = d/dy fp32 (may alias dy) and dx_bf16 = dropout-masked dx (the gradient of the branch that
 *                  was added under dropout: operand of its dX / dW products); dgamma +=, dbeta += (atomic)
 *   meanpool_fwd : feat[b] = mean over ALL L rows of dropout(x[b])  (bert.py:36-37, padding included);  meanpool_bwd: dx = its adjoint.
 *                  seq_len int32 [B] (NULL: L): the padded length of every sequence's OWN batch when differently padded batches (x_lb, x_ulb_w,
 *                  x_ulb_s of a use_cat=False config) share one launch padded to L -- rows >= seq_len[b] are filler: not averaged, zero gradient
 *   gelu_f32 / gelu_bwd_f32 : nn.GELU() between the classifier Linears (bert.py:16-20; the Linears are srhip_fc_fwd / srhip_fc_bwd)
 *   mask_lengths : key_len[b] = sum(attention_mask[b, :]) for the right-padded batches of nlp_collactor.py:63-69 */
int srhip_embed_ln_fwd(const long long* ids, int ld_ids, const int* seq_index, const float* word, const float* pos, const float* type0,
                       const float* gamma, const float* beta, float eps, float* x, void* x_bf16, float* mean, float* rstd, int B, int L, int D,
                       unsigned drop_key, unsigned drop_thresh, float drop_scale, void* stream);
int srhip_embed_ln_bwd(const float* dy, const long long* ids, int ld_ids, const int* seq_index, const float* word, const float* pos,
                       const float* type0, const float* mean, const float* rstd, const float* gamma, float* dword, float* dpos, float* dtype0,
                       float* dgamma, float* dbeta, int B, int L, int D, int pad_id, unsigned drop_key, unsigned drop_thresh, float drop_scale,
                       void* stream);
int srhip_postln_fwd(const float* y, const float* gamma, const float* beta, float eps, float* x, void* x_bf16, float* mean, float* rstd, int M,
                     int D, void* stream);
int srhip_postln_bwd(const float* dy, const float* y, const float* mean, const float* rstd, const float* gamma, float* dx, void* dx_bf16,
                     float* dgamma, float* dbeta, int M, int D, unsigned drop_key, unsigned drop_thresh, float drop_scale, void* stream);
/* postln_bwd with dgamma / dbeta spread over n_rep partial copies part fp32 [n_rep][2][D] (see srhip_layernorm_bwd_part; folded by
 * srhip_ln_grad_reduce after the last layer). */
int srhip_postln_bwd_part(const float* dy, const float* y, const float* mean, const float* rstd, const float* gamma, float* dx, void* dx_bf16,
                          float* part, int n_rep, int M, int D, unsigned drop_key, unsigned drop_thresh, float drop_scale, void* stream);
int srhip_meanpool_fwd(const float* x, float* feat, const int* seq_len, int B, int L, int D, unsigned drop_key, unsigned drop_thresh,
                       float drop_scale, void* stream);
int srhip_meanpool_bwd(const float* dfeat, float* dx, const int* seq_len, int B, int L, int D, unsigned drop_key, unsigned drop_thresh,
                       float drop_scale, void* stream);
int srhip_gelu_f32(const float* pre, float* out, long n, void* stream);
int srhip_gelu_bwd_f32(const float* dout, const float* pre, float* dpre, long n, void* stream);
int srhip_mask_lengths(const long long* mask, int ld, int* key_len, int B, int L, void* stream);
/* out(bf16)[i] = dropout'(x[i]): the gradient of a branch that sat under nn.Dropout, cast for its dX / dW products (element index i) */
int srhip_dropout_cast(const float* x, void* out_bf16, long n, unsigned drop_key, unsigned drop_thresh, float drop_scale, void* stream);

/* gemm_nt with nn.Dropout in the epilogue (generator / index as srhip_attn_masked_fwd, element index m * N + n, ldc == N):
 *   SRHIP_EPI_GELU_BF16  : C = dropout(GELU(acc + bias)), aux_out = acc + bias        (Wav2Vec2FeedForward.intermediate_dropout)
 *   SRHIP_EPI_DGELU_BF16 : C = dropout'(acc) * GELU'(aux_in)                           (its adjoint)
 *   SRHIP_EPI_RESID_F32  : as srhip_gemm_nt_resid_dropout */
int srhip_gemm_nt_dropout(int epilogue, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const float* bias,
                          const void* aux_in, void* aux_out, int ldaux, unsigned drop_key, unsigned drop_thresh, float drop_scale, void* stream);

/* ---- Wav2Vec2 front end (semilearn/nets/wave2vecv2/wave2vecv2.py:44 -> transformers Wav2Vec2Model: feature encoder, feature projection,
 * SpecAugment, positional conv embedding).  Activations are channel-last [clip, frame, channel] with a per-layer frame pitch P_l (P_{l-1} =
 * stride_l * P_l, P_l >= T_l + 1): the conv layers 1.. are srhip_gemm_nt products whose A operand is the previous activation read with
 * lda = stride * C (overlapping rows) against the tap-major filter; rows >= T_l are filler (finite forward, zero in every gradient).
 *   w2v_conv0            : layer 0 (1 input channel) + GroupNorm(groups == channels) + GELU.  mode 0: statistics into ws (2 doubles per (clip,
 *                          channel), zeroed by the caller); 1: out bf16 [B*P0, C]; 2: backward statistics into ws2 from dY (bf16, d/d out),
 *                          dgamma +=, dbeta +=; 3: dW0 [C, k] +=
 *   w2v_conv_weight_prep : Conv1d filter fp32 [Cout, Cin, k] -> bf16 Wr [Cout, k*Cin] (tap-major) and WrT [k*Cin, Cout]; wgrad_add: the inverse
 *                          permutation, dW += sum of the n_part partial products dWr [n_part, Cout, k*Cin] (the weight-gradient product is
 *                          split over the frames to fill the chip)
 *   w2v_col2im_dgelu     : adjoint of the overlapping-row read: dpre_prev = GELU'(pre_prev) * fold(dcol)   (pre_prev NULL: no GELU factor)
 *   w2v_featln_fwd/bwd   : feature_projection.layer_norm on the bf16 conv output; the backward also applies GELU'(pre) of the last conv layer
 *   w2v_spec_mask_fwd/bwd: masked frames <- masked_spec_embed; backward: dx (+= add, the positional-conv input gradient with pitch Padd),
 *                          masked rows -> dembed, filler rows -> 0
 *   w2v_pos_stage        : group-major zero-padded bf16 copy [groups][rows_total][D/groups] of fp32 rows (frame t lands on row clip*Pp + t +
 *                          pad_left): operand of the grouped positional conv (srhip_gemm_nt_grouped_f32, lda = D/groups)
 *   w2v_weightnorm_prep  : weight_norm(dim=2) filter -> bf16 operands Wf [groups][cg][k][cg] and the tap-reversed transpose Wb; norms [k]
 *   w2v_weightnorm_bwd   : dv +=, dg += from dWf (fp32, Wf layout)
 *   w2v_pos_finish_fwd   : x0 = dropout(LayerNorm(x + GELU(conv + bias))) fp32 + bf16 (filler rows zero); saves y, mean, rstd
 *   w2v_pos_finish_bwd   : dx0 -> dy (in place), dconv = dy * GELU'(conv + bias) fp32 [B*P, D]; dgamma +=, dbeta += */
int srhip_w2v_conv0(int mode, const float* wave, const float* W0, const float* gamma, const float* beta, double* ws, double* ws2, void* out_bf16,
                    const void* dY, float* dW0, float* dgamma, float* dbeta, int B, int S, int T0, int P0, int C, int k, int stride, float eps,
                    void* stream);
int srhip_w2v_conv_weight_prep(const float* W, void* Wr, void* WrT, int Cout, int Cin, int k, void* stream);
int srhip_w2v_conv_wgrad_add(const float* dWr, float* dW, int Cout, int Cin, int k, int n_part, void* stream);
int srhip_w2v_col2im_dgelu(const void* dcol, const void* pre_prev, void* out, int B, int Pl, int Pprev, int C, int k, int stride, void* stream);
int srhip_w2v_featln_fwd(const void* x, const float* gamma, const float* beta, float eps, void* out, float* mean, float* rstd, int B, int T, int P,
                         int C, void* stream);
int srhip_w2v_featln_bwd(const void* dy, const void* x, const void* pre, const float* mean, const float* rstd, const float* gamma, void* dpre,
                         float* dgamma, float* dbeta, int B, int T, int P, int C, void* stream);
int srhip_w2v_spec_mask_fwd(float* x, const unsigned char* mask, const float* embed, long M, int D, void* stream);
int srhip_w2v_spec_mask_bwd(float* dx, const float* add, const unsigned char* mask, float* dembed, int B, int T, int P, int Padd, int D, void* stream);
int srhip_w2v_pos_stage(const float* src, void* out, int B, int T, int P, int Pp, int D, int groups, int pad_left, long rows_total, void* stream);
int srhip_w2v_weightnorm_prep(const float* v, const float* g, float* norms, void* Wf, void* Wb, int D, int groups, int k, void* stream);
int srhip_w2v_weightnorm_bwd(const float* dWf, const float* v, const float* g, const float* norms, float* dv, float* dg, int D, int groups, int k,
                             void* stream);
int srhip_w2v_pos_finish_fwd(const float* x, const float* conv, const float* conv_bias, const float* gamma, const float* beta, float eps, float* x0,
                             void* x0_bf16, float* ysave, float* mean, float* rstd, int B, int T, int P, int Pp, int D, unsigned drop_key,
                             unsigned drop_thresh, float drop_scale, void* stream);
int srhip_w2v_pos_finish_bwd(float* dx0, const float* ysave, const float* conv, const float* conv_bias, const float* mean, const float* rstd,
                             const float* gamma, float* dconv, float* dgamma, float* dbeta, int B, int T, int P, int Pp, int D, unsigned drop_key,
                             unsigned drop_thresh, float drop_scale, void* stream);

/* ---- device-side augmentation: semilearn/datasets/cv_datasets/cifar.py:34-49 transform_weak / transform_strong and
 * semilearn/datasets/augmentation/randaugment.py:16-196, one launch per batch.  src uint8 [n_src, H0, W0, 3]; per output image b a parameter
 * block ip[b*64..] (int32) / dp[b*32..] (float64):
 *   ip[0..2] crop offset (row, column) in the reflect-padded image and flip flag; ip[3] number of RandAugment ops (<= 4);
 *   ip[4..7] Cutout rectangle x0, y0, x1, y1, both ends inclusive (x0 < 0: none); ip[8] source image index;
 *   op k at ip[16 + 12k]: [0] op id in augment_list() order, [1] 1 = pure translation (float64 walk: dp xo, yo, step x, step y at [1..4]),
 *   [2..7] the 16.16 fixed-point coefficients a0, a1, a2', a3, a4, a5' of Pillow's nearest-neighbour affine walk, [8] Posterize mask;
 *   dp[8k] the magnitude.
 * scratch: 2 * S * S * 3 bytes per image.  out fp32 [B, 3, S, S] = (x / 255 - mean) / std; out_u8 (optional) uint8 [B, S, S, 3] before that.
 * mean3 / std3: HOST pointers to 3 floats.  Bit-exact with Pillow (12.2.0) for every op. */
int srhip_augment(const unsigned char* src, int n_src, int H0, int W0, int B, int S, int pad, const int* ip, const double* dp,
                  unsigned char* scratch, float* out, unsigned char* out_u8, const float* mean3, const float* std3, void* stream);

/* ---- WideResNet building blocks (classic_cv backbone, semilearn/nets/wrn/wrn.py; BASELINE.json configs[0], parity configuration) ----
 * Feature maps are NHWC = row-major [rows = B*H*W, C].  conv = im2col (bf16) + srhip_gemm_nt; dW = srhip_gemm_tn_grouped_f32(dY, col);
 * dX = srhip_gemm_nt(dY, W^T) + col2im.
 *   nchw_to_nhwc_bf16 : the input batch [B,C,H,W] fp32 -> bf16 [B,H,W,C]
 *   im2col            : col[(b,yo,xo)][(i*k + j)*C + c] = act[b][yo*s+i-p][xo*s+j-p][c]  (k in {1,3}, p = k/2, zero fill, Kpad % 32 == 0;
 *                       column order == Conv2d weight.flatten(1), wrn.py:33-43)
 *   col2im            : the adjoint gather: dact (= | +=) sum of the dcol entries that read each input pixel
 *   conv_weight_prep  : W fp32 [Cout, C, k, k] -> bf16 [Cout, Kpad] in col's tap-major K order and its transpose bf16 [Kpad, Cout];
 *   add_unpad         : the inverse for the gradient: dW[Cout, C, k, k] += dWpad[Cout, Kpad]
 *   bn_fwd            : nn.BatchNorm2d + LeakyReLU(slope) (wrn.py:32-38, :104-105).  training != 0: statistics of THIS batch (saved in
 *                       save_mean / save_invstd), running_mean / running_var moved with ``momentum`` (unbiased variance) unless
 *                       update_running == 0 (Bn_Controller.freeze_bn, core/utils/misc.py:105-129); training == 0: running statistics.
 *                       Outputs: act_bf16 and/or act_f32 (either may be NULL).  256 % C == 0.  ws: srhip_bn_ws_doubles() doubles of scratch the
 *                       caller ZEROES ONCE (every launch leaves its accumulator copies and arrival counter at zero: no memset per call).
 *   bn_bwd            : dx = resid (or 0) + BN'(LeakyReLU'(dact)); dgamma += , dbeta += .
 *   avgpool_fwd/bwd   : F.adaptive_avg_pool2d(.,1) (wrn.py:121);  fc_fwd/bwd: the classifier Linear (wrn.py:106, :126)
 *   sgd_flat          : torch.optim.SGD(momentum, nesterov=True) on a flat block (core/utils/build.py:193-224, optim 'SGD'); chunk table =
 *                       {int64 end, float weight_decay, float pad} per parameter; optional EMA shadow; first_step: buf = d. */
int srhip_nchw_to_nhwc_bf16(const float* img, void* out, int B, int C, int H, int W, void* stream);
int srhip_im2col(const void* act, void* col, int B, int H, int W, int C, int ksize, int stride, int Kpad, void* stream);
/* im2col_bn : col = im2col(bf16(f(x))) from the fp32 tensor in front of the BatchNorm, f = LeakyReLU(BatchNorm(x; mean, invstd)) (mode 0) or
 * identity (mode 2) -- the filter-gradient operand of the backward without a materialised activation.  C % 8 == 0, C <= 256. */
int srhip_im2col_bn(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope, int mode,
                    void* col, int B, int H, int W, int C, int ksize, int stride, int Kpad, void* stream);
int srhip_col2im(const float* dcol, float* dact, int B, int H, int W, int C, int ksize, int stride, int Kpad, int accumulate, void* stream);
int srhip_conv_weight_prep(const float* Wf, void* Wb, void* WbT, int Cout, int C, int ksize, int Kpad, void* stream);
int srhip_add_unpad(const float* src, float* dst, int Cout, int C, int ksize, int Kpad, void* stream);
/* conv_weight_prep / add_unpad for every convolution of a network in ONE launch each: entry p covers [start_p, start_{p+1}) of a flat index
 * space of `total` elements (prep: Cout * Kpad per entry, a = W fp32, b = Wb, c = WbT; unpad: Cout * C * kk per entry, a = dWpad, b = dW). */
typedef struct srhip_conv_desc {
  const void* a; void* b; void* c;
  int Cout, C, kk, Kpad;
  long long start;
} srhip_conv_desc;                       /* 48 bytes */
int srhip_conv_weight_prep_grouped(const srhip_conv_desc* desc_dev, int n, long long total, void* stream);
int srhip_add_unpad_grouped(const srhip_conv_desc* desc_dev, int n, long long total, void* stream);
/* conv_weight_flip_grouped : the filters of the INPUT-gradient convolutions (stride-1 3x3 layers): dX = srhip_wrn_conv_bn(dY, raw mode, W') with
 * W' bf16 [Cin, Kpad], W'[ci][(8 - t) * Cout + co] = W[co][ci][t] -- the adjoint as one more implicit-GEMM launch instead of GEMM + col2im.
 * Entry: a = W fp32 [Cout, C, 3, 3], b = W', kk = 9, Kpad = round32(9 * Cout); Cin * Kpad elements per entry. */
int srhip_conv_weight_flip_grouped(const srhip_conv_desc* desc_dev, int n, long long total, void* stream);
long long srhip_bn_ws_doubles(void);
/* The same BasicBlock as ONE launch per convolution (csrc/wrn_conv.hip; wrn.py:41-60):
 *   wrn_conv_bn : y fp32 [B*Ho*Wo, Cout] = conv_{k,stride,pad k/2}( f(xin) ) (+ resid), xin fp32 NHWC [B,H,W,Cin] read in place (implicit GEMM);
 *                 f by in_mode: 0 = LeakyReLU(BatchNorm(x; in_mean, in_isd = invstd)), 1 = the same from running statistics (in_isd = running
 *                 VAR, in_eps), 2 = identity (the raw-x path of wrn.py:50), 3 = LeakyReLU(BatchNorm(x)) with the batch statistics folded
 *                 from in_acc (the accumulator the launch that produced xin filled).  pub_mean != NULL (with in_acc, any mode): workgroup
 *                 (0,0) publishes mean / invstd of that BatchNorm (the backward reads them) and moves its running statistics when
 *                 update_running (momentum; unbiased variance).  Wb = conv_weight_prep's bf16 [Cout, Kpad].  acc_out != NULL: per-channel
 *                 sum / sum of squares of y are ADDED into it (srhip_bn_acc_doubles(Cout) doubles, zeroed by the caller before the forward)
 *                 for the BatchNorm that reads y next.  stat_ranks: ranks whose sums in_acc holds (1, or the world size under SyncBatchNorm).
 *                 Cin a power of two in [8,128]; Cout in {16, 32, 64} or a multiple of 64.
 *   bn_stats    : the statistics of a tensor no wrn_conv_bn produced: mean / invstd / running update of x fp32 [rows, C] (ws as for bn_fwd).
 *   bn_act      : act bf16 [rows, C] = f(x) by modes 0-2 -- the backward's im2col operand, recomputed instead of stored. */
int srhip_wrn_conv_supported(int Cin, int Cout, int ksize);
/* Diagnostic (no reference counterpart): the kernel template the last srhip_wrn_conv_bn[_passes] call of this process launched -- 100 + 10 NT + PG
 * for wrn_conv_tile_kernel<NT, PG> (input block in LDS), 10 NT + KS for wrn_conv_kernel<NT, KS>; 0 before the first call.  bench.py names the
 * kernel of the WideResNet leg's roofline object with it. */
int srhip_wrn_conv_last_plan(void);
long long srhip_bn_acc_doubles(int C);
int srhip_wrn_conv_bn(const float* xin, int in_mode, const float* in_mean, const float* in_isd, const double* in_acc, const float* in_gamma,
                      const float* in_beta, float in_eps, float slope, float* pub_mean, float* pub_invstd, float* running_mean,
                      float* running_var, float momentum, int update_running, const void* Wb, const float* resid, float* y, int B, int H,
                      int W, int Cin, int Cout, int ksize, int stride, int Kpad, double* acc_out, int stat_ranks, void* stream);
/* The same launch for `passes` independent forwards of B images each (pass-major tensors: xin [passes, B, H, W, Cin], y / resid [passes,
 * B*Ho*Wo, Cout], in_acc / acc_out [passes, srhip_bn_acc_doubles(C)], pub_mean / pub_invstd [passes, Cin]): every pass is its own BatchNorm
 * statistics group, exactly as separate model() calls are -- the K + 1 forwards of x_ulb_w per step of SRPseudoLabel.data_generator under
 * Bn_Controller.freeze_bn (semilearn/algorithms/srpseudolabel/srpseudolabel.py:59-90) share one launch per convolution instead of K + 1.
 * in_mode 2 or 3 only (handed-in statistics would be one group's); update_running moves the running statistics from pass 0 only. */
int srhip_wrn_conv_bn_passes(const float* xin, int in_mode, const float* in_mean, const float* in_isd, const double* in_acc, const float* in_gamma,
                             const float* in_beta, float in_eps, float slope, float* pub_mean, float* pub_invstd, float* running_mean,
                             float* running_var, float momentum, int update_running, const void* Wb, const float* resid, float* y, int B, int H,
                             int W, int Cin, int Cout, int ksize, int stride, int Kpad, double* acc_out, int stat_ranks, int passes, void* stream);
/* wrn_head : the network's tail (wrn.py:119-126) in one launch: feat [B, C] = mean over the HW2 pixels of LeakyReLU(BatchNorm(x)), logits [B, K] =
 * feat Wc^T + bc; in_mode 0 / 1 / 3 and the publishing arguments as for wrn_conv_bn (workgroup 0 publishes).  C <= 256. */
int srhip_wrn_head(const float* x, int in_mode, const float* in_mean, const float* in_isd, const double* in_acc, const float* gamma,
                   const float* beta, float eps, float slope, float* pub_mean, float* pub_invstd, float* running_mean, float* running_var,
                   float momentum, int update_running, const float* Wc, const float* bc, float* feat, float* logits, int B, int HW2, int C,
                   int K, int stat_ranks, void* stream);
/* ... and for `passes` forwards of B images (x [passes * B, HW2, C]; in_acc / published statistics per pass; in_mode 3). */
int srhip_wrn_head_passes(const float* x, int in_mode, const float* in_mean, const float* in_isd, const double* in_acc, const float* gamma,
                   const float* beta, float eps, float slope, float* pub_mean, float* pub_invstd, float* running_mean, float* running_var,
                   float momentum, int update_running, const float* Wc, const float* bc, float* feat, float* logits, int B, int HW2, int C,
                   int K, int stat_ranks, int passes, void* stream);
int srhip_bn_stats(const float* x, float eps, float momentum, int update_running, float* running_mean, float* running_var, float* out_mean,
                   float* out_invstd, double* ws, int rows, int C, void* stream);
int srhip_bn_act(const float* x, const float* mean, const float* invstd_or_var, const float* gamma, const float* beta, float eps, float slope,
                 int mode, void* act_bf16, float* act_f32, int rows, int C, void* stream);
/* SyncBatchNorm (the reference converts the WideResNet's BatchNorms under DDP: core/utils/misc.py:55): the statistics are sums, so the ranks
 * exchange the ACCUMULATOR between the launch that fills it and the launch that folds it (wrn_conv_bn: stat_ranks = world size scales the row
 * count; the caller all-reduces acc in between), and the backward's two column sums between its reduce and apply halves:
 *   bn_accumulate : sums of x fp32 [rows, C] ADDED into acc (srhip_bn_acc_doubles(C) doubles) -- the statistics of a tensor no wrn_conv_bn produced
 *   bn_fold       : acc -> mean / invstd (+ running update with momentum, unbiased variance) and / or the 2C totals; rows_total = rows of all ranks
 *   bn_bwd_reduce : the backward's sums (dy', dy' * xhat) of this rank into ws[0..2C) (ws as for bn_fwd)
 *   bn_bwd_apply  : dx from `totals` over rows_total rows; d(gamma) / d(beta) += local_totals (NULL: totals) -- per-rank, as torch's
 *                   batch_norm_backward_reduce / _elemt split; dx_bf16 (may be NULL): a bf16 copy of dx, the operand of the convolution
 *                   backward that reads it next.  srhip_bn_bwd = reduce + apply on one rank. */
int srhip_bn_accumulate(const float* x, double* acc, int rows, int C, void* stream);
int srhip_bn_fold(const double* acc, double rows_total, float eps, float momentum, int update_running, float* running_mean, float* running_var,
                  float* out_mean, float* out_invstd, double* totals, int C, void* stream);
int srhip_bn_bwd_reduce(const float* dact, const float* x, const float* save_mean, const float* save_invstd, const float* gamma, const float* beta,
                        float slope, double* ws, int rows, int C, void* stream);
int srhip_bn_bwd_apply(const float* dact, const float* x, const float* save_mean, const float* save_invstd, const float* gamma, const float* beta,
                       float slope, const float* resid, float* dx, float* dgamma, float* dbeta, const double* totals, const double* local_totals,
                       double rows_total, void* dx_bf16, int rows, int C, void* stream);
int srhip_bn_fwd(const float* x, const float* gamma, const float* beta, float eps, float slope, float momentum, int training,
                 int update_running, float* running_mean, float* running_var, float* save_mean, float* save_invstd, void* act_bf16,
                 float* act_f32, double* ws, int rows, int C, void* stream);
int srhip_bn_bwd(const float* dact, const float* x, const float* save_mean, const float* save_invstd, const float* gamma, const float* beta,
                 float slope, const float* resid, float* dx, float* dgamma, float* dbeta, double* ws, int rows, int C, void* stream);
int srhip_avgpool_fwd(const float* act, float* feat, int B, int HW2, int C, void* stream);
int srhip_avgpool_bwd(const float* dfeat, float* dact, int B, int HW2, int C, void* stream);
int srhip_fc_fwd(const float* feat, const float* Wc, const float* bc, float* logits, int B, int F, int K, void* stream);
int srhip_fc_bwd(const float* dlogits, const float* feat, const float* Wc, float* dfeat, float* dWc, float* dbc, int B, int F, int K,
                 void* stream);
int srhip_sgd_flat(float* p, float* g, float* buf, float* ema, const void* chunk_table, int nchunks, long long n, float lr, float momentum,
                   float grad_scale, const float* clip_coef, double ema_m, int first_step, int zero_grad, void* stream);

/* Kernel-execution timing for the benchmark's roofline object (no counterpart in the reference, which times whole iterations:
 * semilearn/core/hooks/timer.py).  While enabled, every launch of this library is issued with a (start, stop) event pair bound to the
 * dispatch itself, so srhip_prof_elapsed_ms returns what `rocprofv3 --kernel-trace` reports for the same launches: execution time
 * without dispatch latency or event packets.  Entry points are numbered in launch order since the last srhip_prof_enable(1).
 * These three calls are host-side only (no stream argument); srhip_prof_elapsed_ms needs a synchronised device. */
int srhip_prof_enable(int on);                                   /* returns the number of launches recorded so far */
int srhip_prof_count(void);
int srhip_prof_elapsed_ms(int first, int last, float* ms_sum);   /* HOST pointer: sum over the launches [first, last) */

#ifdef __cplusplus
}
#endif
#endif
