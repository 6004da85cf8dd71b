// Post-LN transformer encoder glue (BERT / Wav2Vec2 backbones of the usb_nlp / usb_audio configs): everything around the GEMMs and
// the attention kernel of one encoder layer, plus the embedding front and the mean-pool + MLP head of the reference classifiers.
//
// Replaces (reference -> third-party HF modules it calls, transformers >= 4.30):
//   semilearn/nets/bert/bert.py:34      BertModel: BertEmbeddings (word + position + token_type -> LayerNorm -> Dropout),
//                                       BertSelfOutput / BertOutput (LayerNorm(x + Dropout(dense(.))), eps 1e-12)
//   semilearn/nets/bert/bert.py:36-37   Dropout(0.1) on last_hidden_state, mean over ALL positions (padding included)
//   semilearn/nets/bert/bert.py:16-20   classifier = Linear, GELU, Linear (GELU between the two srhip_fc_* launches)
// HBM-bound row kernels: one 64-lane wave per token row, the row lives in registers (D = 128 * NV floats, NV float2 per lane),
// statistics by wave shuffles, 8-byte coalesced accesses; column sums (dgamma, dbeta, token-type gradient) are reduced per 32-row
// workgroup in LDS before ONE atomic per column.  Dropout masks come from the counter-based generator in common.h (drop_keep) and
// are regenerated, not stored.
#include "common.h"
#include "srhip.h"

namespace {

struct Drop { uint32_t key, thresh; float scale; };

__device__ __forceinline__ float2 drop2(float2 v, uint32_t idx, const Drop& d) {
  if (d.thresh) {
    bool kx_, ky_;
    drop_keep2(idx, d.key, d.thresh, kx_, ky_);
    v.x = kx_ ? v.x * d.scale : 0.f;
    v.y = ky_ ? v.y * d.scale : 0.f;
  }
  return v;
}

// ---- BertEmbeddings forward --------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void embed_ln_fwd_kernel(const long long* __restrict__ ids, int ld_ids, const int* __restrict__ seq_index,
                                                          const float* __restrict__ word, const float* __restrict__ pos,
                                                          const float* __restrict__ type0, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, float* __restrict__ x,
                                                          bf16_t* __restrict__ xb, float* __restrict__ mean, float* __restrict__ rstd,
                                                          int B, int L, Drop dr) {
  constexpr int D = NV * 128;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B * L) return;
  const int b = row / L, p = row - b * L;
  const long long id = ids[(size_t)(seq_index ? seq_index[b] : b) * ld_ids + p];
  const float2* wr = reinterpret_cast<const float2*>(word + (size_t)id * D);
  const float2* pr = reinterpret_cast<const float2*>(pos + (size_t)p * D);
  const float2* tr = reinterpret_cast<const float2*>(type0);
  float2 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float2 a = wr[i * 64 + lane], c = pr[i * 64 + lane], t = tr[i * 64 + lane];
    v[i] = make_float2(a.x + c.x + t.x, a.y + c.y + t.y);
    s += v[i].x + v[i].y;
  }
  const float mu = wave_sum(s) * (1.0f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) { const float a = v[i].x - mu, c = v[i].y - mu; q += a * a + c * c; }
  const float rs = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
  float2* xr = reinterpret_cast<float2*>(x + (size_t)row * D);
  uint32_t* br = reinterpret_cast<uint32_t*>(xb + (size_t)row * D);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float2 g = reinterpret_cast<const float2*>(gamma)[i * 64 + lane], c = reinterpret_cast<const float2*>(beta)[i * 64 + lane];
    float2 o = make_float2((v[i].x - mu) * rs * g.x + c.x, (v[i].y - mu) * rs * g.y + c.y);
    o = drop2(o, (uint32_t)row * D + 2 * (i * 64 + lane), dr);
    xr[i * 64 + lane] = o;
    br[i * 64 + lane] = pack_bf2(o.x, o.y);
  }
  if (mean && lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// ---- BertEmbeddings backward: dropout' -> LayerNorm' -> scatter into the three tables --------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void embed_ln_bwd_kernel(const float* __restrict__ dy, const long long* __restrict__ ids, int ld_ids,
                                                          const int* __restrict__ seq_index, const float* __restrict__ word,
                                                          const float* __restrict__ pos, const float* __restrict__ type0,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, float* __restrict__ dword,
                                                          float* __restrict__ dpos, float* __restrict__ dtype0,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int L, int pad_id,
                                                          Drop dr) {
  constexpr int D = NV * 128;
  __shared__ float red[3][4][D];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, M = B * L;
  float2 ag[NV], ab[NV], at[NV], g[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    ag[i] = ab[i] = at[i] = make_float2(0.f, 0.f);
    g[i] = reinterpret_cast<const float2*>(gamma)[i * 64 + lane];
  }
  for (int rr = 0; rr < 8; ++rr) {
    const int row = blockIdx.x * 32 + wave * 8 + rr;
    if (row >= M) break;
    const int b = row / L, p = row - b * L;
    const long long id = ids[(size_t)(seq_index ? seq_index[b] : b) * ld_ids + p];
    const float2* wr = reinterpret_cast<const float2*>(word + (size_t)id * D);
    const float2* pr = reinterpret_cast<const float2*>(pos + (size_t)p * D);
    const float2* tr = reinterpret_cast<const float2*>(type0);
    const float2* dr_ = reinterpret_cast<const float2*>(dy + (size_t)row * D);
    const float mu = mean[row], rs = rstd[row];
    float2 xh[NV], dh[NV];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float2 a = wr[i * 64 + lane], c = pr[i * 64 + lane], t = tr[i * 64 + lane];
      xh[i] = make_float2((a.x + c.x + t.x - mu) * rs, (a.y + c.y + t.y - mu) * rs);
      const float2 d = drop2(dr_[i * 64 + lane], (uint32_t)row * D + 2 * (i * 64 + lane), dr);
      ag[i].x += d.x * xh[i].x; ag[i].y += d.y * xh[i].y;
      ab[i].x += d.x; ab[i].y += d.y;
      dh[i] = make_float2(d.x * g[i].x, d.y * g[i].y);
      c1 += dh[i].x + dh[i].y;
      c2 += dh[i].x * xh[i].x + dh[i].y * xh[i].y;
    }
    c1 = wave_sum(c1) * (1.0f / D);
    c2 = wave_sum(c2) * (1.0f / D);
    float* wg = dword + (size_t)id * D;
    float* pg = dpos + (size_t)p * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float ex = rs * (dh[i].x - c1 - xh[i].x * c2), ey = rs * (dh[i].y - c1 - xh[i].y * c2);
      const int c = 2 * (i * 64 + lane);
      if (id != pad_id) { atomicAdd(wg + c, ex); atomicAdd(wg + c + 1, ey); }      // nn.Embedding(padding_idx): no gradient for [PAD]
      atomicAdd(pg + c, ex); atomicAdd(pg + c + 1, ey);
      at[i].x += ex; at[i].y += ey;
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 2 * (i * 64 + lane);
    red[0][wave][c] = ag[i].x; red[0][wave][c + 1] = ag[i].y;
    red[1][wave][c] = ab[i].x; red[1][wave][c + 1] = ab[i].y;
    red[2][wave][c] = at[i].x; red[2][wave][c + 1] = at[i].y;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    atomicAdd(dgamma + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
    atomicAdd(dbeta + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
    atomicAdd(dtype0 + c, red[2][0][c] + red[2][1][c] + red[2][2][c] + red[2][3][c]);
  }
}

// ---- post-LN forward: x = LayerNorm(y) in fp32 (residual stream of the next sub-layer) and bf16 (its GEMM operand) ------------
// (two rows per wave, every load requested before the first reduction: see ln_fwd_kernel in vit_ops.hip; x may alias y -- a wave reads
// both of its rows before it writes either)
template <int NV>
__global__ __launch_bounds__(256) void postln_fwd_kernel(const float* __restrict__ y, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, float* __restrict__ x,
                                                        bf16_t* __restrict__ xb, float* __restrict__ mean, float* __restrict__ rstd, int M) {
  constexpr int D = NV * 128, RW = 2;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RW, lane = threadIdx.x & 63;
  if (row0 >= M) return;
  float2 v[RW][NV], g[NV], c[NV];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const float2* yr = reinterpret_cast<const float2*>(y + (size_t)min(row0 + r, M - 1) * D);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[r][i] = yr[i * 64 + lane];
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    g[i] = reinterpret_cast<const float2*>(gamma)[i * 64 + lane];
    c[i] = reinterpret_cast<const float2*>(beta)[i * 64 + lane];
  }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int row = row0 + r;
    if (row >= M) break;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[r][i].x + v[r][i].y;
    const float mu = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) { const float a = v[r][i].x - mu, e = v[r][i].y - mu; q += a * a + e * e; }
    const float rs = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
    float2* xr = reinterpret_cast<float2*>(x + (size_t)row * D);
    uint32_t* br = reinterpret_cast<uint32_t*>(xb + (size_t)row * D);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float2 o = make_float2((v[r][i].x - mu) * rs * g[i].x + c[i].x, (v[r][i].y - mu) * rs * g[i].y + c[i].y);
      if (x) xr[i * 64 + lane] = o;                          // x == NULL: the consumer applies this LayerNorm to the residual it reads (srhip_gemm_nt_resid_ln_dropout)
      br[i * 64 + lane] = pack_bf2(o.x, o.y);
    }
    if (mean && lane == 0) { mean[row] = mu; rstd[row] = rs; }
  }
}

// ---- post-LN backward: dy = d/d(LayerNorm output) -> dx = d/d(y) in fp32 (the residual path) and, masked by the dropout that sat on
// the branch (y = x_prev + dropout(branch)), in bf16 (operand of the branch's dX and dW products) ------------------------------
template <int NV, int RPW>
__global__ __launch_bounds__(256) void postln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, float* __restrict__ dx, bf16_t* __restrict__ dxb,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta, int M, Drop dr,
                                                        float* __restrict__ part, int n_rep) {
  constexpr int D = NV * 128;
  __shared__ float red[2][4][D];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float2 ag[NV], ab[NV], g[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    ag[i] = ab[i] = make_float2(0.f, 0.f);
    g[i] = reinterpret_cast<const float2*>(gamma)[i * 64 + lane];
  }
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = blockIdx.x * (4 * RPW) + wave * RPW + rr;
    if (row >= M) break;
    const float mu = mean[row], rs = rstd[row];
    const float2* yr = reinterpret_cast<const float2*>(y + (size_t)row * D);
    const float2* dr_ = reinterpret_cast<const float2*>(dy + (size_t)row * D);
    float2 xh[NV], dh[NV];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float2 a = yr[i * 64 + lane], d = dr_[i * 64 + lane];
      xh[i] = make_float2((a.x - mu) * rs, (a.y - mu) * rs);
      ag[i].x += d.x * xh[i].x; ag[i].y += d.y * xh[i].y;
      ab[i].x += d.x; ab[i].y += d.y;
      dh[i] = make_float2(d.x * g[i].x, d.y * g[i].y);
      c1 += dh[i].x + dh[i].y;
      c2 += dh[i].x * xh[i].x + dh[i].y * xh[i].y;
    }
    c1 = wave_sum(c1) * (1.0f / D);
    c2 = wave_sum(c2) * (1.0f / D);
    float2* xr = reinterpret_cast<float2*>(dx + (size_t)row * D);
    uint32_t* br = reinterpret_cast<uint32_t*>(dxb + (size_t)row * D);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float2 e = make_float2(rs * (dh[i].x - c1 - xh[i].x * c2), rs * (dh[i].y - c1 - xh[i].y * c2));
      xr[i * 64 + lane] = e;
      const float2 m = drop2(e, (uint32_t)row * D + 2 * (i * 64 + lane), dr);
      br[i * 64 + lane] = pack_bf2(m.x, m.y);
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 2 * (i * 64 + lane);
    red[0][wave][c] = ag[i].x; red[0][wave][c + 1] = ag[i].y;
    red[1][wave][c] = ab[i].x; red[1][wave][c + 1] = ab[i].y;
  }
  __syncthreads();
  if (n_rep > 0) { dgamma = part + (size_t)(blockIdx.x % n_rep) * 2 * D; dbeta = dgamma + D; }     // partial copies: see ln_bwd_kernel (vit_ops.hip)
  for (int c = threadIdx.x; c < D; c += 256) {
    atomicAdd(dgamma + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
    atomicAdd(dbeta + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
  }
}

// ---- head: feat[b] = mean over ALL L positions of dropout(x[b])   (bert.py:36-37) -----------------------------------------
// seq_len (optional): logical padded length of every sequence when batches padded to different lengths share one launch (rows
// [seq_len[b], L) exist only as filler: excluded from the mean, zero gradient).
__global__ __launch_bounds__(256) void meanpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ feat, const int* __restrict__ seq_len,
                                                          int L, int D, Drop dr) {
  __shared__ float part[4][64];
  const int b = blockIdx.x, d = blockIdx.y * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  const int Ls = seq_len ? seq_len[b] : L;
  // four rows in flight per wave (one load per trip left the 512-token rows of a sequence a latency chain: 145 us per launch for 107 MB)
  float s = 0.f, sa[4] = {0.f, 0.f, 0.f, 0.f};
  int p = w;
  for (; p + 12 < Ls; p += 16) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t i = ((size_t)b * L + p + 4 * u) * D + d;
      const float v = x[i];
      sa[u] += (!dr.thresh || drop_keep((uint32_t)i, dr.key, dr.thresh)) ? v : 0.f;
    }
  }
  for (; p < Ls; p += 4) {
    const size_t i = ((size_t)b * L + p) * D + d;
    const float v = x[i];
    s += (!dr.thresh || drop_keep((uint32_t)i, dr.key, dr.thresh)) ? v : 0.f;
  }
  s += (sa[0] + sa[1]) + (sa[2] + sa[3]);
  part[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0) feat[(size_t)b * D + d] = (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]) * (dr.scale / Ls);
}
__global__ __launch_bounds__(256) void meanpool_bwd_kernel(const float* __restrict__ dfeat, float* __restrict__ dx, const int* __restrict__ seq_len,
                                                          int L, int D, long n, Drop dr) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int d = (int)(i % D);
  const long b = i / ((long)L * D);
  const int p = (int)((i / D) % L), Ls = seq_len ? seq_len[b] : L;
  const bool keep = p < Ls && (!dr.thresh || drop_keep((uint32_t)i, dr.key, dr.thresh));
  dx[i] = keep ? dfeat[b * D + d] * (dr.scale / Ls) : 0.f;
}

// nn.GELU() (exact erf) between the two classifier Linears, fp32
__global__ __launch_bounds__(256) void gelu_f32_kernel(const float* __restrict__ pre, float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const float v = pre[i]; out[i] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
}
__global__ __launch_bounds__(256) void gelu_bwd_f32_kernel(const float* __restrict__ dout, const float* __restrict__ pre, float* __restrict__ dpre, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const float v = pre[i];
    dpre[i] = dout[i] * (0.5f * (1.0f + erff(v * 0.70710678118654752440f)) + v * 0.39894228040143267794f * expf(-0.5f * v * v));
  }
}

// bf16 cast of a gradient with the adjoint of a dropout site folded in (element index = linear index)
__global__ __launch_bounds__(256) void dropout_cast_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, long n, Drop dr) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= n) return;
  const float2 v = drop2(*reinterpret_cast<const float2*>(x + i), (uint32_t)i, dr);
  *reinterpret_cast<uint32_t*>(out + i) = pack_bf2(v.x, v.y);
}

// key lengths of a right-padded batch: klen[b] = sum(attention_mask[b, :])   (nlp_collactor.py:63-69 pads on the right)
__global__ __launch_bounds__(64) void mask_len_kernel(const long long* __restrict__ mask, int ld, int* __restrict__ klen, int L) {
  float s = 0.f;
  for (int p = threadIdx.x; p < L; p += 64) s += (float)(mask[(size_t)blockIdx.x * ld + p] != 0);
  s = wave_sum(s);
  if (threadIdx.x == 0) klen[blockIdx.x] = (int)s;
}

#define DISPATCH_NV(D, CALL)              \
  if ((D) == 128) { CALL(1); }            \
  else if ((D) == 384) { CALL(3); }       \
  else if ((D) == 768) { CALL(6); }       \
  else return SR_EINVAL;

}  // namespace

extern "C" int srhip_embed_ln_fwd(const long long* ids, int ld_ids, const int* seq_index, const float* word, const float* pos,
                                  const float* type0, const float* gamma, const float* beta, float eps, float* x, void* x_bf16, float* mean,
                                  float* rstd, int B, int L, int D, unsigned drop_key, unsigned drop_thresh, float drop_scale, void* stream) {
  if (!ids || !word || !pos || !type0 || !x || !x_bf16 || B <= 0 || L <= 0 || ((mean == nullptr) != (rstd == nullptr))) return SR_EINVAL;
  const Drop dr{drop_key, drop_thresh, drop_scale};
  hipStream_t s = (hipStream_t)stream;
#define CALL(NV) SR_LAUNCH(embed_ln_fwd_kernel<NV>, dim3(cdiv((long)B * L, 4)), dim3(256), 0, s, ids, ld_ids, seq_index, word, pos, type0, \
                                    gamma, beta, eps, x, (bf16_t*)x_bf16, mean, rstd, B, L, dr)
  DISPATCH_NV(D, CALL)
#undef CALL
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_embed_ln_bwd(const float* dy, const long long* ids, int ld_ids, const int* seq_index, const float* word, const float* pos,
                                  const float* type0, const float* mean, const float* rstd, const float* gamma, float* dword, float* dpos,
                                  float* dtype0, float* dgamma, float* dbeta, int B, int L, int D, int pad_id, unsigned drop_key,
                                  unsigned drop_thresh, float drop_scale, void* stream) {
  if (!dy || !ids || !mean || !rstd || !dword || !dpos || !dtype0 || !dgamma || !dbeta || B <= 0 || L <= 0) return SR_EINVAL;
  const Drop dr{drop_key, drop_thresh, drop_scale};
  hipStream_t s = (hipStream_t)stream;
#define CALL(NV) SR_LAUNCH(embed_ln_bwd_kernel<NV>, dim3(cdiv((long)B * L, 32)), dim3(256), 0, s, dy, ids, ld_ids, seq_index, word, pos, \
                                    type0, mean, rstd, gamma, dword, dpos, dtype0, dgamma, dbeta, B, L, pad_id, dr)
  DISPATCH_NV(D, CALL)
#undef CALL
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_postln_fwd(const float* y, const float* gamma, const float* beta, float eps, float* x, void* x_bf16, float* mean,
                                float* rstd, int M, int D, void* stream) {
  if (!y || !x_bf16 || M <= 0 || ((mean == nullptr) != (rstd == nullptr)) || (!x && !mean)) return SR_EINVAL;      // x == NULL: statistics required
  hipStream_t s = (hipStream_t)stream;
#define CALL(NV) SR_LAUNCH(postln_fwd_kernel<NV>, dim3(cdiv(M, 8)), dim3(256), 0, s, y, gamma, beta, eps, x, (bf16_t*)x_bf16, mean, rstd, M)
  DISPATCH_NV(D, CALL)
#undef CALL
  SR_CHECK_LAUNCH();
  return SR_OK;
}

static int postln_bwd_impl(const float* dy, const float* y, const float* mean, const float* rstd, const float* gamma, float* dx,
                           void* dx_bf16, float* dgamma, float* dbeta, float* part, int n_rep, int M, int D, unsigned drop_key,
                           unsigned drop_thresh, float drop_scale, void* stream) {
  if (!dy || !y || !mean || !rstd || !dx || !dx_bf16 || M <= 0) return SR_EINVAL;
  const Drop dr{drop_key, drop_thresh, drop_scale};
  hipStream_t s = (hipStream_t)stream;
  const bool small = M < 16384;       // few rows: 8 instead of 32 per workgroup (see ln_bwd_kernel)
#define CALL(NV)                                                                                                                                    \
  do {                                                                                                                                              \
    if (small) SR_LAUNCH((postln_bwd_kernel<NV, 2>), dim3(cdiv(M, 8)), dim3(256), 0, s, dy, y, mean, rstd, gamma, dx, (bf16_t*)dx_bf16,     \
                                  dgamma, dbeta, M, dr, part, n_rep);                                                                               \
    else SR_LAUNCH((postln_bwd_kernel<NV, 8>), dim3(cdiv(M, 32)), dim3(256), 0, s, dy, y, mean, rstd, gamma, dx, (bf16_t*)dx_bf16, dgamma, \
                            dbeta, M, dr, part, n_rep);                                                                                             \
  } while (0)
  DISPATCH_NV(D, CALL)
#undef CALL
  SR_CHECK_LAUNCH();
  return SR_OK;
}
extern "C" int srhip_postln_bwd(const float* dy, const float* y, const float* mean, const float* rstd, const float* gamma, float* dx,
                                void* dx_bf16, float* dgamma, float* dbeta, int M, int D, unsigned drop_key, unsigned drop_thresh,
                                float drop_scale, void* stream) {
  if (!dgamma || !dbeta) return SR_EINVAL;
  return postln_bwd_impl(dy, y, mean, rstd, gamma, dx, dx_bf16, dgamma, dbeta, nullptr, 0, M, D, drop_key, drop_thresh, drop_scale, stream);
}
extern "C" int srhip_postln_bwd_part(const float* dy, const float* y, const float* mean, const float* rstd, const float* gamma, float* dx,
                                     void* dx_bf16, float* part, int n_rep, int M, int D, unsigned drop_key, unsigned drop_thresh,
                                     float drop_scale, void* stream) {
  if (!part || n_rep <= 0) return SR_EINVAL;
  return postln_bwd_impl(dy, y, mean, rstd, gamma, dx, dx_bf16, nullptr, nullptr, part, n_rep, M, D, drop_key, drop_thresh, drop_scale, stream);
}

extern "C" int srhip_meanpool_fwd(const float* x, float* feat, const int* seq_len, int B, int L, int D, unsigned drop_key, unsigned drop_thresh,
                                  float drop_scale, void* stream) {
  if (!x || !feat || B <= 0 || L <= 0 || D % 64 || (long)B * L * D >= (1L << 32)) return SR_EINVAL;
  const Drop dr{drop_key, drop_thresh, drop_thresh ? drop_scale : 1.0f};
  SR_LAUNCH(meanpool_fwd_kernel, dim3(B, D / 64), dim3(256), 0, (hipStream_t)stream, x, feat, seq_len, L, D, dr);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
extern "C" int srhip_meanpool_bwd(const float* dfeat, float* dx, const int* seq_len, int B, int L, int D, unsigned drop_key, unsigned drop_thresh,
                                  float drop_scale, void* stream) {
  const long n = (long)B * L * D;
  if (!dfeat || !dx || B <= 0 || L <= 0 || D <= 0 || n >= (1L << 32)) return SR_EINVAL;
  const Drop dr{drop_key, drop_thresh, drop_thresh ? drop_scale : 1.0f};
  SR_LAUNCH(meanpool_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dfeat, dx, seq_len, L, D, n, dr);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_gelu_f32(const float* pre, float* out, long n, void* stream) {
  if (!pre || !out || n <= 0) return SR_EINVAL;
  SR_LAUNCH(gelu_f32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, pre, out, n);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
extern "C" int srhip_gelu_bwd_f32(const float* dout, const float* pre, float* dpre, long n, void* stream) {
  if (!dout || !pre || !dpre || n <= 0) return SR_EINVAL;
  SR_LAUNCH(gelu_bwd_f32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dout, pre, dpre, n);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_dropout_cast(const float* x, void* out_bf16, long n, unsigned drop_key, unsigned drop_thresh, float drop_scale, void* stream) {
  if (!x || !out_bf16 || n <= 0 || (n & 1) || n >= (1L << 32)) return SR_EINVAL;
  const Drop dr{drop_key, drop_thresh, drop_scale};
  SR_LAUNCH(dropout_cast_kernel, dim3(cdiv(n / 2, 256)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)out_bf16, n, dr);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_mask_lengths(const long long* mask, int ld, int* klen, int B, int L, void* stream) {
  if (!mask || !klen || B <= 0 || L <= 0) return SR_EINVAL;
  SR_LAUNCH(mask_len_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, mask, ld, klen, L);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
