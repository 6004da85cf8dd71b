// Wav2Vec2 front end (usb_audio backbone): everything between the raw waveform and the post-LN encoder layers that the BERT path
// already provides (enc_ops.hip, attention.hip, gemm.hip).
//
// Replaces (reference -> third-party HF modules it calls, transformers >= 4.30):
//   semilearn/nets/wave2vecv2/wave2vecv2.py:44  Wav2Vec2Model: Wav2Vec2FeatureEncoder (7 x Conv1d, GroupNorm after the first, GELU),
//                                               Wav2Vec2FeatureProjection, SpecAugment (_mask_hidden_states), Wav2Vec2PositionalConvEmbedding
// Layout: channel-last activations [clip, frame, channel] with a per-layer frame PITCH P_l >= T_l + 1 chosen so that
// P_{l-1} = stride_l * P_l: the rows (clip, t) of a k-tap stride-s Conv1d input window are then ONE contiguous run of k*C values at
// row (clip * P_{l-1} + s*t) -- the im2col matrix of the layers 1..6 is the activation itself read with lda = s*C (overlapping rows),
// so those convolutions are plain srhip_gemm_nt launches with the GELU epilogue and no unfold kernel.  Rows t >= T_l are filler: finite
// values forward, exact zeros in every gradient buffer.  The grouped positional convolution uses the same trick on a group-major,
// zero-padded staging copy [group][clip * Pp + frame][C / groups].
// Layer 0 (1 input channel, k = 10) is a direct kernel: 20 flop per output, recomputed instead of stored (its pre-GroupNorm output would
// be 26 MB per 4-second clip in fp32).
#include <stdlib.h>

#include "common.h"
#include "srhip.h"

namespace {

constexpr int TCH = 128;   // frames per workgroup of the layer-0 kernels
constexpr int K0MAX = 16;

// nn.GELU() ("exact erf") and its derivative through the Abramowitz-Stegun erf of common.h (|err| <= 1.5e-7, one v_exp + one v_rcp, no
// branches) -- the form the GEMM epilogues of the same encoder use; the OCML erff / expf pair was ~50 of the 80 instructions per output of the
// layer-0 backward passes.
__device__ __forceinline__ float gelu_exact(float v) { return gelu_erf(v); }
__device__ __forceinline__ float gelu_exact_grad(float v) { return gelu_erf_grad(v); }

// stage the waveform segment of TCH frames (stride s, k taps) of clip b into LDS
__device__ __forceinline__ void stage_wave(float* seg, const float* wave, int S, int b, int t0, int k, int s) {
  const int n = s * (TCH - 1) + k;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int p = t0 * s + i;
    seg[i] = p < S ? wave[(size_t)b * S + p] : 0.f;
  }
}

// MODE 0: statistics (sum, sum of squares of the conv output over time) -> ws[(b*C + c)*2 + {0,1}] (double atomics)
// MODE 1: apply: out = GELU(GroupNorm(conv)) as bf16 [B*P0, C]; filler rows zero
// MODE 2: backward statistics from dY: ws2[(b*C+c)*2 + {0,1}] = sum dxhat, sum dxhat * xhat;  dgamma +=, dbeta +=
// MODE 3: backward weights: dW0[c][j] += sum_t dconv * wave[s*t + j]
template <int MODE>
__global__ __launch_bounds__(256) void conv0_kernel(const float* __restrict__ wave, const float* __restrict__ W0, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, double* __restrict__ ws, double* __restrict__ ws2,
                                                   bf16_t* __restrict__ out, const bf16_t* __restrict__ dY, float* __restrict__ dW0,
                                                   float* __restrict__ dgamma, float* __restrict__ dbeta, int S, int T0, int P0, int C, int k,
                                                   int s, float eps) {
  extern __shared__ float seg[];
  const int b = blockIdx.x, t0 = blockIdx.y * TCH;
  stage_wave(seg, wave, S, b, t0, k, s);
  __syncthreads();
  const int tend = min(TCH, (MODE == 1 ? P0 : T0) - t0);
  for (int c = threadIdx.x; c < C; c += 256) {
    float w[K0MAX];
#pragma unroll
    for (int j = 0; j < K0MAX; ++j) w[j] = j < k ? W0[c * k + j] : 0.f;
    float mean = 0.f, rstd = 0.f, g = 0.f, be = 0.f, a1 = 0.f, a2 = 0.f;
    if (MODE >= 1) {
      const double m = ws[((size_t)b * C + c) * 2] / T0, q = ws[((size_t)b * C + c) * 2 + 1] / T0;
      mean = (float)m; rstd = rsqrtf((float)(q - m * m) + eps);
      g = gamma[c]; be = beta[c];
    }
    if (MODE == 3) { a1 = (float)(ws2[((size_t)b * C + c) * 2] / T0); a2 = (float)(ws2[((size_t)b * C + c) * 2 + 1] / T0); }
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    float dw[K0MAX];
#pragma unroll
    for (int j = 0; j < K0MAX; ++j) dw[j] = 0.f;
    for (int tt = 0; tt < tend; ++tt) {
      const int t = t0 + tt;
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < K0MAX; ++j) if (j < k) v = fmaf(w[j], seg[tt * s + j], v);
      if (MODE == 0) { acc0 += v; acc1 += v * v; }
      if (MODE == 1) {
        const float y = t < T0 ? gelu_erf((v - mean) * rstd * g + be) : 0.f;      // A-S erf of common.h (|err| < 1.5e-7, output is bf16): erff tripled this pass
        out[((size_t)b * P0 + t) * C + c] = f2bf(y);
      }
      if (MODE >= 2) {
        const float xh = (v - mean) * rstd, dgn = bf2f(dY[((size_t)b * P0 + t) * C + c]) * gelu_exact_grad(xh * g + be);
        const float dxh = dgn * g;
        if (MODE == 2) { acc0 += dxh; acc1 += dxh * xh; acc2 += dgn * xh; acc3 += dgn; }
        if (MODE == 3) {
          const float dc = rstd * (dxh - a1 - xh * a2);
#pragma unroll
          for (int j = 0; j < K0MAX; ++j) if (j < k) dw[j] = fmaf(dc, seg[tt * s + j], dw[j]);
        }
      }
    }
    if (MODE == 0) { atomicAdd(&ws[((size_t)b * C + c) * 2], (double)acc0); atomicAdd(&ws[((size_t)b * C + c) * 2 + 1], (double)acc1); }
    if (MODE == 2) {
      atomicAdd(&ws2[((size_t)b * C + c) * 2], (double)acc0); atomicAdd(&ws2[((size_t)b * C + c) * 2 + 1], (double)acc1);
      atomicAdd(dgamma + c, acc2); atomicAdd(dbeta + c, acc3);
    }
    if (MODE == 3) {
#pragma unroll
      for (int j = 0; j < K0MAX; ++j) if (j < k) atomicAdd(dW0 + c * k + j, dw[j]);
    }
  }
}

// The same four passes for the shape every usb_audio config has (k = 10, stride 5), four frames per iteration.  In the generic kernel a thread
// reads one waveform sample per multiply-add from LDS (ds_read_b32, every lane the same address): 10 LDS instructions per frame and channel,
// 108 k of them per CU -- the LDS issue rate, not the arithmetic, bounded the statistics pass (380 us for 12 arithmetic instructions per
// output).  Four consecutive frames read the 25 samples [5 t, 5 t + 25) as seven 16-byte LDS loads (t a multiple of 4: 80-byte steps) and keep
// them in registers: 1.75 LDS instructions per frame instead of 10.
template <int MODE>
__global__ __launch_bounds__(256) void conv0_k10s5_kernel(const float* __restrict__ wave, const float* __restrict__ W0, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, double* __restrict__ ws, double* __restrict__ ws2,
                                                         bf16_t* __restrict__ out, const bf16_t* __restrict__ dY, float* __restrict__ dW0,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, int S, int T0, int P0, int C,
                                                         float eps, int nsub) {
  constexpr int K = 10, ST = 5, FR = 4, MAXC = 2;                    // C <= 512: a thread owns channels threadIdx.x and threadIdx.x + 256
  extern __shared__ __attribute__((aligned(16))) float seg[];
  const int b = blockIdx.x, tlim = MODE == 1 ? P0 : T0;
  float w[MAXC][K], mean[MAXC], rstd[MAXC], g[MAXC], be[MAXC], a1[MAXC], a2[MAXC];
  float acc0[MAXC], acc1[MAXC], acc2[MAXC], acc3[MAXC], dw[MODE == 3 ? MAXC : 1][K];
#pragma unroll
  for (int q = 0; q < MAXC; ++q) {
    const int c = threadIdx.x + 256 * q;
    acc0[q] = acc1[q] = acc2[q] = acc3[q] = 0.f;
    mean[q] = rstd[q] = g[q] = be[q] = a1[q] = a2[q] = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) { w[q][j] = c < C ? W0[c * K + j] : 0.f; if (MODE == 3) dw[q][j] = 0.f; }
    if (c < C && MODE >= 1) {
      const double m = ws[((size_t)b * C + c) * 2] / T0, qq = ws[((size_t)b * C + c) * 2 + 1] / T0;
      mean[q] = (float)m; rstd[q] = rsqrtf((float)(qq - m * m) + eps);
      g[q] = gamma[c]; be[q] = beta[c];
    }
    if (c < C && MODE == 3) { a1[q] = (float)(ws2[((size_t)b * C + c) * 2] / T0); a2[q] = (float)(ws2[((size_t)b * C + c) * 2 + 1] / T0); }
  }
  // the reduction passes walk nsub segments per workgroup and add their sums once: the filter gradient alone was 13.8 M float atomics onto
  // 5120 addresses (160 cache lines) -- at one atomic per clock and L2 channel that is 0.4 ms, twice the pass's arithmetic
  for (int sub = 0; sub < nsub; ++sub) {
    const int t0 = (blockIdx.y * nsub + sub) * TCH;
    if (t0 >= tlim) break;                                  // (uniform over the workgroup)
    if (sub) __syncthreads();                               // the previous segment's readers are done with seg
    stage_wave(seg, wave, S, b, t0, K, ST);
    __syncthreads();
    const int tend = min(TCH, tlim - t0);
#pragma unroll
    for (int q = 0; q < MAXC; ++q) {
      const int c = threadIdx.x + 256 * q;
      if (c < C) {
        for (int tb = 0; tb < tend; tb += FR) {
          float x[28];
#pragma unroll
          for (int i = 0; i < 7; ++i) {
            const float4 v4 = *reinterpret_cast<const float4*>(seg + tb * ST + 4 * i);
            x[4 * i] = v4.x; x[4 * i + 1] = v4.y; x[4 * i + 2] = v4.z; x[4 * i + 3] = v4.w;
          }
#pragma unroll
          for (int f = 0; f < FR; ++f) {
            const int tt = tb + f, t = t0 + tt;
            if (tt >= tend) break;
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < K; ++j) v = fmaf(w[q][j], x[f * ST + j], v);        // (the generic kernel's order of the multiply-adds)
            if (MODE == 0) { acc0[q] += v; acc1[q] += v * v; }
            if (MODE == 1) {
              const float y = t < T0 ? gelu_erf((v - mean[q]) * rstd[q] * g[q] + be[q]) : 0.f;
              out[((size_t)b * P0 + t) * C + c] = f2bf(y);
            }
            if (MODE >= 2) {
              const float xh = (v - mean[q]) * rstd[q], dgn = bf2f(dY[((size_t)b * P0 + t) * C + c]) * gelu_exact_grad(xh * g[q] + be[q]);
              const float dxh = dgn * g[q];
              if (MODE == 2) { acc0[q] += dxh; acc1[q] += dxh * xh; acc2[q] += dgn * xh; acc3[q] += dgn; }
              if (MODE == 3) {
                const float dc = rstd[q] * (dxh - a1[q] - xh * a2[q]);
#pragma unroll
                for (int j = 0; j < K; ++j) dw[q][j] = fmaf(dc, x[f * ST + j], dw[q][j]);
              }
            }
          }
        }
      }
    }
  }
  // hardware fp atomics (global_atomic_add_f32 / _f64)
#pragma unroll
  for (int q = 0; q < MAXC; ++q) {
    const int c = threadIdx.x + 256 * q;
    if (c >= C) continue;
    if (MODE == 0) { unsafeAtomicAdd(&ws[((size_t)b * C + c) * 2], (double)acc0[q]); unsafeAtomicAdd(&ws[((size_t)b * C + c) * 2 + 1], (double)acc1[q]); }
    if (MODE == 2) {
      unsafeAtomicAdd(&ws2[((size_t)b * C + c) * 2], (double)acc0[q]); unsafeAtomicAdd(&ws2[((size_t)b * C + c) * 2 + 1], (double)acc1[q]);
      unsafeAtomicAdd(dgamma + c, acc2[q]); unsafeAtomicAdd(dbeta + c, acc3[q]);
    }
    if (MODE == 3) {
#pragma unroll
      for (int j = 0; j < K; ++j) unsafeAtomicAdd(dW0 + c * K + j, dw[q][j]);
    }
  }
}

// Conv1d filter [Cout, Cin, k] fp32 -> tap-major bf16 operands: Wr [Cout, k*Cin] (forward / dW layout) and WrT [k*Cin, Cout] (dX)
__global__ __launch_bounds__(256) void conv_w_prep_kernel(const float* __restrict__ W, bf16_t* __restrict__ Wr, bf16_t* __restrict__ WrT, int Cout,
                                                         int Cin, int k) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x, n = (long)Cout * Cin * k;
  if (i >= n) return;
  const int j = (int)(i % k), ci = (int)((i / k) % Cin), co = (int)(i / ((long)k * Cin));
  const bf16_t v = f2bf(W[i]);
  Wr[(size_t)co * k * Cin + (size_t)j * Cin + ci] = v;
  WrT[((size_t)j * Cin + ci) * Cout + co] = v;
}
// n_part partial products (split over the token dimension: a 512 x 1536 weight gradient is 48 tiles, far too few for 256 CUs when the
// reduction runs over 100 000 frames) are summed on the way
__global__ __launch_bounds__(256) void conv_wgrad_add_kernel(const float* __restrict__ dWr, float* __restrict__ dW, int Cout, int Cin, int k,
                                                            int n_part) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x, n = (long)Cout * Cin * k;
  if (i >= n) return;
  const int j = (int)(i % k), ci = (int)((i / k) % Cin), co = (int)(i / ((long)k * Cin));
  const size_t src = (size_t)co * k * Cin + (size_t)j * Cin + ci;
  float a = 0.f;
  for (int s = 0; s < n_part; ++s) a += dWr[(size_t)s * n + src];
  dW[i] += a;
}

// adjoint of the overlapping-row read of a stride-s k-tap conv + (optionally) the GELU of the layer below:
//   dpre_prev[clip, tau, c] = gelu'(pre_prev) * sum_{j : (tau - j) % s == 0, 0 <= (tau - j)/s < Pl} dcol[clip, (tau - j)/s, j, c]
// Eight channels (16 bytes) per thread: one element per thread made this a kernel of 2-byte accesses and 64-bit divisions per element (161 us per
// launch at 1.6 TB/s, 3.7 % of the HuBERT leg); C % 8 == 0 (the conv width is 512).
__global__ __launch_bounds__(256) void col2im_dgelu_kernel(const bf16_t* __restrict__ dcol, const bf16_t* __restrict__ pre_prev,
                                                          bf16_t* __restrict__ out, int Pl, int Pprev, int C, int k, int s, long n8) {
  const long i8 = (long)blockIdx.x * 256 + threadIdx.x;           // index of the 8-channel group
  if (i8 >= n8) return;
  const int C8 = C >> 3;
  const int c = (int)(i8 % C8) * 8;
  const long row = i8 / C8;
  const int tau = (int)(row % Pprev);
  const long clip = row / Pprev;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < k; ++j) {
    const int u = tau - j;
    if (u < 0 || (u % s)) continue;
    const int t = u / s;
    if (t >= Pl) continue;
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(dcol + ((size_t)(clip * Pl + t) * k + j) * C + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[2 * e] += __uint_as_float(v[e] << 16); a[2 * e + 1] += __uint_as_float(v[e] & 0xffff0000u); }
  }
  if (pre_prev) {
    const u32x4_t pv = *reinterpret_cast<const u32x4_t*>(pre_prev + i8 * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[2 * e] *= gelu_exact_grad(__uint_as_float(pv[e] << 16));
      a[2 * e + 1] *= gelu_exact_grad(__uint_as_float(pv[e] & 0xffff0000u));
    }
  }
  *reinterpret_cast<u32x4_t*>(out + i8 * 8) = u32x4_t{pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]), pack_bf2(a[4], a[5]), pack_bf2(a[6], a[7])};
}

// SpecAugment: masked frames are replaced by the learned embedding (forward, in place); their gradient goes to the embedding (backward);
// filler frames (t >= T) get zero gradient.  ``add`` (optional, pitch Padd): the positional-conv input gradient, summed in first.
__global__ __launch_bounds__(256) void spec_mask_fwd_kernel(float* __restrict__ x, const unsigned char* __restrict__ mask, const float* __restrict__ embed,
                                                           int D, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (mask[i / D]) x[i] = embed[i % D];
}
__global__ __launch_bounds__(256) void spec_mask_bwd_kernel(float* __restrict__ dx, const float* __restrict__ add, const unsigned char* __restrict__ mask,
                                                           float* __restrict__ dembed, int T, int P, int Padd, int D) {
  // grid = (D / 64, clips); block 256 = 4 frame lanes x 64 columns
  __shared__ float part[4][64];
  const int d = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6, clip = blockIdx.y;
  float acc = 0.f;
  for (int t = w; t < P; t += 4) {
    const size_t i = ((size_t)clip * P + t) * D + d;
    float v = t < T ? dx[i] + (add ? add[((size_t)clip * Padd + t) * D + d] : 0.f) : 0.f;
    if (t < T && mask && mask[(size_t)clip * P + t]) { acc += v; v = 0.f; }
    dx[i] = v;
  }
  part[w][threadIdx.x & 63] = acc;
  __syncthreads();
  if (w == 0 && mask) atomicAdd(dembed + d, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// group-major zero-padded staging copy for the grouped positional conv: out[g][clip * Pp + u][cg] = src[clip, u - pad_left, g*cg + c]
// (0 outside [0, T)); src fp32 with frame pitch P (forward: hidden states) or Psrc rows of a gradient buffer.
// (eight channels per thread: 32 bytes read, 16 written; cg % 8 == 0)
__global__ __launch_bounds__(256) void pos_stage_kernel(const float* __restrict__ src, bf16_t* __restrict__ out, int B, int T, int P, int Pp, int D,
                                                       int cg, int pad_left, long rows_total, long n8) {
  const long i8 = (long)blockIdx.x * 256 + threadIdx.x;
  if (i8 >= n8) return;
  const int cg8 = cg >> 3;
  const int c = (int)(i8 % cg8) * 8;
  const long r = (i8 / cg8) % rows_total;
  const int g = (int)(i8 / ((long)cg8 * rows_total));
  const long clip = r / Pp;
  const int t = (int)(r % Pp) - pad_left;
  u32x4_t o = {0u, 0u, 0u, 0u};
  if (clip < B && t >= 0 && t < T) {                         // (rows past the last clip are slack: zeros)
    const float* sp = src + ((size_t)clip * P + t) * D + g * cg + c;
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(sp), b = *reinterpret_cast<const f32x4_t*>(sp + 4);
    o = u32x4_t{pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]), pack_bf2(b[0], b[1]), pack_bf2(b[2], b[3])};
  }
  *reinterpret_cast<u32x4_t*>(out + i8 * 8) = o;
}

// weight_norm(dim=2) of the positional conv filter: w[co][ci][j] = g[j] * v[co][ci][j] / ||v[:, :, j]||.
// pass 1: norms[j];  pass 2: bf16 operands  Wf[grp][co_l][j][ci] (forward B operand [N = cg, K = k*cg], also the dW layout) and
// Wb[grp][ci][j'][co_l] = w[.., k-1-j'] (B operand of the input-gradient correlation).
__global__ __launch_bounds__(256) void wn_norm_kernel(const float* __restrict__ v, float* __restrict__ norms, int n_per_tap, int k) {
  const int j = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < n_per_tap; i += 256) { const float a = v[(size_t)i * k + j]; s += a * a; }
  __shared__ float red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) norms[j] = sqrtf(red[0] + red[1] + red[2] + red[3]);
}
__global__ __launch_bounds__(256) void wn_prep_kernel(const float* __restrict__ v, const float* __restrict__ gw, const float* __restrict__ norms,
                                                     bf16_t* __restrict__ Wf, bf16_t* __restrict__ Wb, int D, int cg, int k) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x, n = (long)D * cg * k;
  if (i >= n) return;
  const int j = (int)(i % k), ci = (int)((i / k) % cg), co = (int)(i / ((long)k * cg));
  const int grp = co / cg, col = co % cg;
  const bf16_t w = f2bf(gw[j] * v[i] / norms[j]);
  Wf[(((size_t)grp * cg + col) * k + j) * cg + ci] = w;
  Wb[(((size_t)grp * cg + ci) * k + (k - 1 - j)) * cg + col] = w;
}
// backward: dWf fp32 [grp][co_l][j][ci] -> dv += g/||v|| (dW - v * (sum dW.v)/||v||^2), dg[j] += (sum dW.v)/||v||     (grid = k)
__global__ __launch_bounds__(256) void wn_bwd_kernel(const float* __restrict__ dWf, const float* __restrict__ v, const float* __restrict__ gw,
                                                    const float* __restrict__ norms, float* __restrict__ dv, float* __restrict__ dg, int D, int cg,
                                                    int k) {
  const int j = blockIdx.x, n = D * cg;
  __shared__ float red[4];
  __shared__ float dot_s;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int co = i / cg, ci = i % cg;
    s += dWf[(((size_t)(co / cg) * cg + co % cg) * k + j) * cg + ci] * v[(size_t)i * k + j];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { dot_s = red[0] + red[1] + red[2] + red[3]; dg[j] += dot_s / norms[j]; }
  __syncthreads();
  const float dot = dot_s, nn = norms[j], sc = gw[j] / nn;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int co = i / cg, ci = i % cg;
    const float dw = dWf[(((size_t)(co / cg) * cg + co % cg) * k + j) * cg + ci];
    dv[(size_t)i * k + j] += sc * (dw - v[(size_t)i * k + j] * dot / (nn * nn));
  }
}

// The same backward in three coalesced passes (256 % k == 0, k * (cg + 1) floats of LDS).  The kernel above is one workgroup per tap that walks
// v / dv at a stride of k floats (4-byte accesses 512 bytes apart) twice: 284 us per step for 19 MB.  Here a workgroup owns ONE output channel:
// its dWf block [k][cg] goes through LDS so that dWf, v and dv are all read and written along their contiguous axes.
//   pass 1 (grid = D):  partial[co][j] = sum_ci dWf[co][j][ci] * v[co][ci][j]
//   pass 2 (1 workgroup): dot[j] = sum_co partial[co][j] (fixed order: deterministic);  dg[j] += dot[j] / norms[j]
//   pass 3 (grid = D):  dv[co][ci][j] += g[j] / norms[j] * (dWf[co][j][ci] - v[co][ci][j] * dot[j] / norms[j]^2)
__global__ __launch_bounds__(256) void wn_bwd_dot_kernel(const float* __restrict__ dWf, const float* __restrict__ v, float* __restrict__ partial, int cg, int k) {
  extern __shared__ float wt[];                                   // [k][cg + 1] + 256 floats of reduction space
  float* red = wt + k * (cg + 1);
  const int co = blockIdx.x, n = cg * k, tid = threadIdx.x;
  for (int e = tid; e < n; e += 256) wt[(e / cg) * (cg + 1) + e % cg] = dWf[(size_t)co * n + e];          // dWf block is [j][ci]
  __syncthreads();
  const int j = tid % k;                                          // fixed per thread: 256 % k == 0
  float s = 0.f;
  for (int e = tid; e < n; e += 256) s += wt[j * (cg + 1) + e / k] * v[(size_t)co * n + e];                // v block is [ci][j]
  red[tid] = s;
  __syncthreads();
  if (tid < k) {
    float a = 0.f;
    for (int q = 0; q < 256 / k; ++q) a += red[tid + q * k];
    partial[(size_t)co * k + tid] = a;
  }
}
__global__ __launch_bounds__(1024) void wn_bwd_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ norms, float* __restrict__ dot,
                                                            float* __restrict__ dg, int D, int k) {
  // one workgroup of 1024 threads = (1024 / k) channel groups x k taps, four independent partial sums per thread (one chain of D / nq dependent
  // loads per thread made this pass 86 us); the order of the sum is fixed
  __shared__ float red[1024];
  const int tid = threadIdx.x, j = tid % k, q0 = tid / k, nq = 1024 / k;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int co = q0;
  for (; co + 3 * nq < D; co += 4 * nq) {
    s0 += partial[(size_t)co * k + j];
    s1 += partial[(size_t)(co + nq) * k + j];
    s2 += partial[(size_t)(co + 2 * nq) * k + j];
    s3 += partial[(size_t)(co + 3 * nq) * k + j];
  }
  for (; co < D; co += nq) s0 += partial[(size_t)co * k + j];
  red[tid] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (tid < k) {
    float a = 0.f;
    for (int q = 0; q < nq; ++q) a += red[tid + q * k];
    dot[tid] = a;
    dg[tid] += a / norms[tid];
  }
}
__global__ __launch_bounds__(256) void wn_bwd_apply_kernel(const float* __restrict__ dWf, const float* __restrict__ v, const float* __restrict__ gw,
                                                          const float* __restrict__ norms, const float* __restrict__ dot, float* __restrict__ dv,
                                                          int cg, int k) {
  extern __shared__ float wt[];
  const int co = blockIdx.x, n = cg * k, tid = threadIdx.x;
  for (int e = tid; e < n; e += 256) wt[(e / cg) * (cg + 1) + e % cg] = dWf[(size_t)co * n + e];
  __syncthreads();
  const int j = tid % k;
  const float nn = norms[j], sc = gw[j] / nn, dn = dot[j] / (nn * nn);
  for (int e = tid; e < n; e += 256) {
    const size_t i = (size_t)co * n + e;
    dv[i] += sc * (wt[j * (cg + 1) + e / k] - v[i] * dn);
  }
}

// encoder input: y = x + GELU(conv + bias);  x0 = dropout(LayerNorm(y))  -- one wave per frame, rows (clip, t) with pitch P; the conv
// rows have pitch Pp.  Filler frames produce zeros.  Saves y and the statistics for the backward.
struct Drop { uint32_t key, thresh; float scale; };
template <int NV>
__global__ __launch_bounds__(256) void pos_finish_fwd_kernel(const float* __restrict__ x, const float* __restrict__ conv, const float* __restrict__ cbias,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            float* __restrict__ x0, bf16_t* __restrict__ x0b, float* __restrict__ ysave,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int T, int P, int Pp, int M, Drop dr) {
  constexpr int D = NV * 128;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const int clip = row / P, t = row - clip * P;
  float2* xr = reinterpret_cast<float2*>(x0 + (size_t)row * D);
  uint32_t* br = reinterpret_cast<uint32_t*>(x0b + (size_t)row * D);
  if (t >= T) {
#pragma unroll
    for (int i = 0; i < NV; ++i) { xr[i * 64 + lane] = make_float2(0.f, 0.f); br[i * 64 + lane] = 0u; }
    if (ysave) {
#pragma unroll
      for (int i = 0; i < NV; ++i) reinterpret_cast<float2*>(ysave + (size_t)row * D)[i * 64 + lane] = make_float2(0.f, 0.f);
    }
    if (mean && lane == 0) { mean[row] = 0.f; rstd[row] = 0.f; }
    return;
  }
  const float2* xi = reinterpret_cast<const float2*>(x + (size_t)row * D);
  const float2* cr = reinterpret_cast<const float2*>(conv + ((size_t)clip * Pp + t) * D);
  float2 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float2 a = xi[i * 64 + lane], c = cr[i * 64 + lane], bb = reinterpret_cast<const float2*>(cbias)[i * 64 + lane];
    v[i] = make_float2(a.x + gelu_exact(c.x + bb.x), a.y + gelu_exact(c.y + bb.y));
    s += v[i].x + v[i].y;
  }
  const float mu = wave_sum(s) * (1.0f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) { const float a = v[i].x - mu, c = v[i].y - mu; q += a * a + c * c; }
  const float rs = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float2 g = reinterpret_cast<const float2*>(gamma)[i * 64 + lane], c = reinterpret_cast<const float2*>(beta)[i * 64 + lane];
    float2 o = make_float2((v[i].x - mu) * rs * g.x + c.x, (v[i].y - mu) * rs * g.y + c.y);
    if (dr.thresh) {
      const uint32_t idx = (uint32_t)row * D + 2 * (i * 64 + lane);
      bool kx_, ky_;
      drop_keep2(idx, dr.key, dr.thresh, kx_, ky_);
      o.x = kx_ ? o.x * dr.scale : 0.f;
      o.y = ky_ ? o.y * dr.scale : 0.f;
    }
    xr[i * 64 + lane] = o;
    br[i * 64 + lane] = pack_bf2(o.x, o.y);
    if (ysave) reinterpret_cast<float2*>(ysave + (size_t)row * D)[i * 64 + lane] = v[i];
  }
  if (mean && lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// backward of the above: dx0 -> (dropout', LayerNorm') -> dy;  dx = dy (residual path, fp32, in place over dx0);
// dconv = dy * gelu'(conv + bias) as fp32 [M, D] with pitch P (staged group-major by pos_stage_kernel afterwards); dgamma/dbeta atomics
template <int NV>
__global__ __launch_bounds__(256) void pos_finish_bwd_kernel(float* __restrict__ dx0, const float* __restrict__ ysave, const float* __restrict__ conv,
                                                            const float* __restrict__ cbias, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            float* __restrict__ dconv, float* __restrict__ dgamma, float* __restrict__ dbeta, int T,
                                                            int P, int Pp, int M, Drop dr) {
  constexpr int D = NV * 128;
  __shared__ float red[2][4][D];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float2 ag[NV], ab[NV], g[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    ag[i] = ab[i] = make_float2(0.f, 0.f);
    g[i] = reinterpret_cast<const float2*>(gamma)[i * 64 + lane];
  }
  for (int rr = 0; rr < 8; ++rr) {
    const int row = blockIdx.x * 32 + wave * 8 + rr;
    if (row >= M) break;
    const int clip = row / P, t = row - clip * P;
    float2* dr_ = reinterpret_cast<float2*>(dx0 + (size_t)row * D);
    float2* dc = reinterpret_cast<float2*>(dconv + (size_t)row * D);
    if (t >= T) {
#pragma unroll
      for (int i = 0; i < NV; ++i) { dr_[i * 64 + lane] = make_float2(0.f, 0.f); dc[i * 64 + lane] = make_float2(0.f, 0.f); }
      continue;
    }
    const float mu = mean[row], rs = rstd[row];
    const float2* yr = reinterpret_cast<const float2*>(ysave + (size_t)row * D);
    const float2* cr = reinterpret_cast<const float2*>(conv + ((size_t)clip * Pp + t) * D);
    float2 xh[NV], dh[NV];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float2 a = yr[i * 64 + lane];
      float2 d = dr_[i * 64 + lane];
      if (dr.thresh) {
        const uint32_t idx = (uint32_t)row * D + 2 * (i * 64 + lane);
        bool kx_, ky_;
        drop_keep2(idx, dr.key, dr.thresh, kx_, ky_);
        d.x = kx_ ? d.x * dr.scale : 0.f;
        d.y = ky_ ? d.y * dr.scale : 0.f;
      }
      xh[i] = make_float2((a.x - mu) * rs, (a.y - mu) * rs);
      ag[i].x += d.x * xh[i].x; ag[i].y += d.y * xh[i].y;
      ab[i].x += d.x; ab[i].y += d.y;
      dh[i] = make_float2(d.x * g[i].x, d.y * g[i].y);
      c1 += dh[i].x + dh[i].y;
      c2 += dh[i].x * xh[i].x + dh[i].y * xh[i].y;
    }
    c1 = wave_sum(c1) * (1.0f / D);
    c2 = wave_sum(c2) * (1.0f / D);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float2 e = make_float2(rs * (dh[i].x - c1 - xh[i].x * c2), rs * (dh[i].y - c1 - xh[i].y * c2));
      const float2 c = cr[i * 64 + lane], bb = reinterpret_cast<const float2*>(cbias)[i * 64 + lane];
      dr_[i * 64 + lane] = e;
      dc[i * 64 + lane] = make_float2(e.x * gelu_exact_grad(c.x + bb.x), e.y * gelu_exact_grad(c.y + bb.y));
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 2 * (i * 64 + lane);
    red[0][wave][c] = ag[i].x; red[0][wave][c + 1] = ag[i].y;
    red[1][wave][c] = ab[i].x; red[1][wave][c + 1] = ab[i].y;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    atomicAdd(dgamma + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
    atomicAdd(dbeta + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
  }
}

// Wav2Vec2FeatureProjection.layer_norm on the bf16 output of the last conv layer (rows (clip, t), pitch P; filler rows -> zeros).
template <int NV>
__global__ __launch_bounds__(256) void featln_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, bf16_t* __restrict__ out, float* __restrict__ mean, float* __restrict__ rstd, int T,
                                                        int P, int M) {
  constexpr int D = NV * 128;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  uint32_t* orow = reinterpret_cast<uint32_t*>(out + (size_t)row * D);
  if (row % P >= T) {
#pragma unroll
    for (int i = 0; i < NV; ++i) orow[i * 64 + lane] = 0u;
    if (mean && lane == 0) { mean[row] = 0.f; rstd[row] = 0.f; }
    return;
  }
  const uint32_t* xr = reinterpret_cast<const uint32_t*>(x + (size_t)row * D);
  float2 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const uint32_t u = xr[i * 64 + lane];
    v[i] = make_float2(bf2f((bf16_t)(u & 0xffff)), bf2f((bf16_t)(u >> 16)));
    s += v[i].x + v[i].y;
  }
  const float mu = wave_sum(s) * (1.0f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) { const float a = v[i].x - mu, c = v[i].y - mu; q += a * a + c * c; }
  const float rs = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float2 g = reinterpret_cast<const float2*>(gamma)[i * 64 + lane], c = reinterpret_cast<const float2*>(beta)[i * 64 + lane];
    orow[i * 64 + lane] = pack_bf2((v[i].x - mu) * rs * g.x + c.x, (v[i].y - mu) * rs * g.y + c.y);
  }
  if (mean && lane == 0) { mean[row] = mu; rstd[row] = rs; }
}
// its backward fused with the GELU of the last conv layer: dpre = LN'(dy) * gelu'(pre)  (bf16; filler rows zero); dgamma/dbeta atomics
template <int NV>
__global__ __launch_bounds__(256) void featln_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const bf16_t* __restrict__ pre,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                        bf16_t* __restrict__ dpre, float* __restrict__ dgamma, float* __restrict__ dbeta, int T, int P,
                                                        int M) {
  constexpr int D = NV * 128;
  __shared__ float red[2][4][D];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float2 ag[NV], ab[NV], g[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    ag[i] = ab[i] = make_float2(0.f, 0.f);
    g[i] = reinterpret_cast<const float2*>(gamma)[i * 64 + lane];
  }
  for (int rr = 0; rr < 8; ++rr) {
    const int row = blockIdx.x * 32 + wave * 8 + rr;
    if (row >= M) break;
    uint32_t* orow = reinterpret_cast<uint32_t*>(dpre + (size_t)row * D);
    if (row % P >= T) {
#pragma unroll
      for (int i = 0; i < NV; ++i) orow[i * 64 + lane] = 0u;
      continue;
    }
    const float mu = mean[row], rs = rstd[row];
    const uint32_t* xr = reinterpret_cast<const uint32_t*>(x + (size_t)row * D);
    const uint32_t* dr = reinterpret_cast<const uint32_t*>(dy + (size_t)row * D);
    const uint32_t* pr = reinterpret_cast<const uint32_t*>(pre + (size_t)row * D);
    float2 xh[NV], dh[NV];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const uint32_t u = xr[i * 64 + lane], d2 = dr[i * 64 + lane];
      const float d0 = bf2f((bf16_t)(d2 & 0xffff)), d1 = bf2f((bf16_t)(d2 >> 16));
      xh[i] = make_float2((bf2f((bf16_t)(u & 0xffff)) - mu) * rs, (bf2f((bf16_t)(u >> 16)) - mu) * rs);
      ag[i].x += d0 * xh[i].x; ag[i].y += d1 * xh[i].y;
      ab[i].x += d0; ab[i].y += d1;
      dh[i] = make_float2(d0 * g[i].x, d1 * g[i].y);
      c1 += dh[i].x + dh[i].y;
      c2 += dh[i].x * xh[i].x + dh[i].y * xh[i].y;
    }
    c1 = wave_sum(c1) * (1.0f / D);
    c2 = wave_sum(c2) * (1.0f / D);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const uint32_t pu = pr[i * 64 + lane];
      const float e0 = rs * (dh[i].x - c1 - xh[i].x * c2) * gelu_exact_grad(bf2f((bf16_t)(pu & 0xffff)));
      const float e1 = rs * (dh[i].y - c1 - xh[i].y * c2) * gelu_exact_grad(bf2f((bf16_t)(pu >> 16)));
      orow[i * 64 + lane] = pack_bf2(e0, e1);
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 2 * (i * 64 + lane);
    red[0][wave][c] = ag[i].x; red[0][wave][c + 1] = ag[i].y;
    red[1][wave][c] = ab[i].x; red[1][wave][c + 1] = ab[i].y;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    atomicAdd(dgamma + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
    atomicAdd(dbeta + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
  }
}

}  // namespace

#define W2V_LAUNCH1D(kern, n, ...) SR_LAUNCH(kern, dim3(cdiv((n), 256)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__)

extern "C" int srhip_w2v_conv0(int mode, const float* wave, const float* W0, const float* gamma, const float* beta, double* ws, double* ws2,
                               void* out_bf16, const void* dY, float* dW0, float* dgamma, float* dbeta, int B, int S, int T0, int P0, int C, int k,
                               int stride, float eps, void* stream) {
  if (!wave || !W0 || !ws || B <= 0 || k > K0MAX || C <= 0 || T0 <= 0 || P0 < T0 || mode < 0 || mode > 3) return SR_EINVAL;
  const dim3 grid(B, cdiv(mode == 1 ? P0 : T0, TCH)), block(256);
  const size_t sm = (size_t)(stride * (TCH - 1) + k) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
#define C0(MODE) SR_LAUNCH(conv0_kernel<MODE>, grid, block, sm, s, wave, W0, gamma, beta, ws, ws2, (bf16_t*)out_bf16, (const bf16_t*)dY, dW0, \
                                    dgamma, dbeta, S, T0, P0, C, k, stride, eps)
  static const bool generic_only = SR_TUNE_ENV("SRHIP_W2V_CONV0_GENERIC") != nullptr;            // tuning: the generic kernel for every shape
  if (k == 10 && stride == 5 && !generic_only && C <= 512) {
    static const int env_nsub = SR_TUNE_ENV("SRHIP_W2V_CONV0_NSUB") ? atoi(SR_TUNE_ENV("SRHIP_W2V_CONV0_NSUB")) : 0;
    // segments per workgroup (measured per pass, 27 clips: statistics 139 / 154 / 172 us at 1 / 2 / 4, backward statistics 181 / 256 / 454,
    // filter gradient 544 / 395 / 455): only the filter gradient, 10 atomics per channel and workgroup, gains from fewer workgroups
    const int nsub = mode == 1 ? 1 : (env_nsub > 0 ? env_nsub : (mode == 3 ? 2 : 1));
    const dim3 gridf(B, cdiv(mode == 1 ? P0 : T0, TCH * nsub));
    const size_t smf = sm + 16 * sizeof(float);             // the last 16-byte read of a segment may reach 3 samples past it (never used)
#define C0F(MODE) SR_LAUNCH(conv0_k10s5_kernel<MODE>, gridf, block, smf, s, wave, W0, gamma, beta, ws, ws2, (bf16_t*)out_bf16, \
                                     (const bf16_t*)dY, dW0, dgamma, dbeta, S, T0, P0, C, eps, nsub)
    if (mode == 0) C0F(0); else if (mode == 1) C0F(1); else if (mode == 2) C0F(2); else C0F(3);
#undef C0F
  } else {
    if (mode == 0) C0(0); else if (mode == 1) C0(1); else if (mode == 2) C0(2); else C0(3);
  }
#undef C0
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_w2v_conv_weight_prep(const float* W, void* Wr, void* WrT, int Cout, int Cin, int k, void* stream) {
  if (!W || !Wr || !WrT || Cout <= 0 || Cin <= 0 || k <= 0) return SR_EINVAL;
  W2V_LAUNCH1D(conv_w_prep_kernel, (long)Cout * Cin * k, W, (bf16_t*)Wr, (bf16_t*)WrT, Cout, Cin, k);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
extern "C" int srhip_w2v_conv_wgrad_add(const float* dWr, float* dW, int Cout, int Cin, int k, int n_part, void* stream) {
  if (!dWr || !dW || Cout <= 0 || Cin <= 0 || k <= 0 || n_part <= 0) return SR_EINVAL;
  W2V_LAUNCH1D(conv_wgrad_add_kernel, (long)Cout * Cin * k, dWr, dW, Cout, Cin, k, n_part);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
extern "C" int srhip_w2v_col2im_dgelu(const void* dcol, const void* pre_prev, void* out, int B, int Pl, int Pprev, int C, int k, int stride,
                                      void* stream) {
  if (!dcol || !out || B <= 0 || Pl <= 0 || Pprev <= 0 || C <= 0 || (C & 7) || (((uintptr_t)dcol | (uintptr_t)pre_prev | (uintptr_t)out) & 15)) return SR_EINVAL;
  const long n = (long)B * Pprev * (C / 8);
  W2V_LAUNCH1D(col2im_dgelu_kernel, n, (const bf16_t*)dcol, (const bf16_t*)pre_prev, (bf16_t*)out, Pl, Pprev, C, k, stride, n);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
extern "C" int srhip_w2v_spec_mask_fwd(float* x, const unsigned char* mask, const float* embed, long M, int D, void* stream) {
  if (!x || !mask || !embed || M <= 0 || D <= 0) return SR_EINVAL;
  W2V_LAUNCH1D(spec_mask_fwd_kernel, M * D, x, mask, embed, D, M * D);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
extern "C" int srhip_w2v_spec_mask_bwd(float* dx, const float* add, const unsigned char* mask, float* dembed, int B, int T, int P, int Padd, int D,
                                       void* stream) {
  if (!dx || B <= 0 || D % 64 || (mask && !dembed)) return SR_EINVAL;
  SR_LAUNCH(spec_mask_bwd_kernel, dim3(D / 64, B), dim3(256), 0, (hipStream_t)stream, dx, add, mask, dembed, T, P, Padd, D);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
extern "C" int srhip_w2v_pos_stage(const float* src, void* out, int B, int T, int P, int Pp, int D, int groups, int pad_left, long rows_total,
                                   void* stream) {
  if (!src || !out || B <= 0 || D % groups || ((D / groups) & 7) || rows_total < (long)B * Pp || (((uintptr_t)src | (uintptr_t)out) & 15)) return SR_EINVAL;
  const long n = rows_total * (D / 8);
  W2V_LAUNCH1D(pos_stage_kernel, n, src, (bf16_t*)out, B, T, P, Pp, D, D / groups, pad_left, rows_total, n);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
extern "C" int srhip_w2v_weightnorm_prep(const float* v, const float* g, float* norms, void* Wf, void* Wb, int D, int groups, int k, void* stream) {
  if (!v || !g || !norms || !Wf || !Wb || D % groups) return SR_EINVAL;
  const int cg = D / groups;
  SR_LAUNCH(wn_norm_kernel, dim3(k), dim3(256), 0, (hipStream_t)stream, v, norms, D * cg, k);
  SR_CHECK_LAUNCH();
  W2V_LAUNCH1D(wn_prep_kernel, (long)D * cg * k, v, g, norms, (bf16_t*)Wf, (bf16_t*)Wb, D, cg, k);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
extern "C" int srhip_w2v_weightnorm_bwd(const float* dWf, const float* v, const float* g, const float* norms, float* dv, float* dg, int D, int groups,
                                        int k, void* stream) {
  if (!dWf || !v || !g || !norms || !dv || !dg || D % groups) return SR_EINVAL;
  const int cg = D / groups;
  const size_t lds = ((size_t)k * (cg + 1) + 256) * sizeof(float);
  if (k > 0 && 256 % k == 0 && lds <= 64 * 1024) {
    // library-owned scratch for the per-channel partial sums and the per-tap dots ((D + 1) * k floats; one backward at a time per process)
    static float* scratch = nullptr;
    static size_t scratch_n = 0;
    const size_t need = ((size_t)D + 1) * k;
    if (need > scratch_n) {
      if (scratch) (void)hipFree(scratch);
      if (hipMalloc(&scratch, need * sizeof(float)) != hipSuccess) { scratch = nullptr; scratch_n = 0; return SR_ELAUNCH; }
      scratch_n = need;
    }
    float* partial = scratch;
    float* dot = scratch + (size_t)D * k;
    hipStream_t st = (hipStream_t)stream;
    if (lds > 48 * 1024) {
      (void)hipFuncSetAttribute((const void*)wn_bwd_dot_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void*)wn_bwd_apply_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    SR_LAUNCH(wn_bwd_dot_kernel, dim3(D), dim3(256), lds, st, dWf, v, partial, cg, k);
    SR_LAUNCH(wn_bwd_reduce_kernel, dim3(1), dim3(1024), 0, st, partial, norms, dot, dg, D, k);
    SR_LAUNCH(wn_bwd_apply_kernel, dim3(D), dim3(256), lds, st, dWf, v, g, norms, dot, dv, cg, k);
    SR_CHECK_LAUNCH();
    return SR_OK;
  }
  SR_LAUNCH(wn_bwd_kernel, dim3(k), dim3(256), 0, (hipStream_t)stream, dWf, v, g, norms, dv, dg, D, cg, k);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

#define W2V_NV(D, CALL)                   \
  if ((D) == 128) { CALL(1); }            \
  else if ((D) == 384) { CALL(3); }       \
  else if ((D) == 768) { CALL(6); }       \
  else return SR_EINVAL;

extern "C" int srhip_w2v_pos_finish_fwd(const float* x, const float* conv, const float* conv_bias, const float* gamma, const float* beta, float eps,
                                        float* x0, void* x0_bf16, float* ysave, float* mean, float* rstd, int B, int T, int P, int Pp, int D,
                                        unsigned drop_key, unsigned drop_thresh, float drop_scale, void* stream) {
  if (!x || !conv || !x0 || !x0_bf16 || B <= 0 || T <= 0 || P < T || Pp < T) return SR_EINVAL;
  const Drop dr{drop_key, drop_thresh, drop_scale};
  const int M = B * P;
#define CALL(NV) SR_LAUNCH(pos_finish_fwd_kernel<NV>, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, conv, conv_bias, gamma, beta, eps, \
                                    x0, (bf16_t*)x0_bf16, ysave, mean, rstd, T, P, Pp, M, dr)
  W2V_NV(D, CALL)
#undef CALL
  SR_CHECK_LAUNCH();
  return SR_OK;
}
extern "C" int srhip_w2v_pos_finish_bwd(float* dx0, const float* ysave, const float* conv, const float* conv_bias, const float* mean,
                                        const float* rstd, const float* gamma, float* dconv, float* dgamma, float* dbeta, int B, int T, int P,
                                        int Pp, int D, unsigned drop_key, unsigned drop_thresh, float drop_scale, void* stream) {
  if (!dx0 || !ysave || !conv || !mean || !rstd || !dconv || !dgamma || !dbeta || B <= 0) return SR_EINVAL;
  const Drop dr{drop_key, drop_thresh, drop_scale};
  const int M = B * P;
#define CALL(NV) SR_LAUNCH(pos_finish_bwd_kernel<NV>, dim3(cdiv(M, 32)), dim3(256), 0, (hipStream_t)stream, dx0, ysave, conv, conv_bias, mean, \
                                    rstd, gamma, dconv, dgamma, dbeta, T, P, Pp, M, dr)
  W2V_NV(D, CALL)
#undef CALL
  SR_CHECK_LAUNCH();
  return SR_OK;
}

#define W2V_NVC(D, CALL)                  \
  if ((D) == 128) { CALL(1); }            \
  else if ((D) == 256) { CALL(2); }       \
  else if ((D) == 512) { CALL(4); }       \
  else if ((D) == 768) { CALL(6); }       \
  else return SR_EINVAL;

extern "C" int srhip_w2v_featln_fwd(const void* x, const float* gamma, const float* beta, float eps, void* out, float* mean, float* rstd, int B, int T,
                                    int P, int C, void* stream) {
  if (!x || !out || B <= 0 || T <= 0 || P < T || ((mean == nullptr) != (rstd == nullptr))) return SR_EINVAL;
  const int M = B * P;
#define CALL(NV) SR_LAUNCH(featln_fwd_kernel<NV>, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, gamma, beta, eps, \
                                    (bf16_t*)out, mean, rstd, T, P, M)
  W2V_NVC(C, CALL)
#undef CALL
  SR_CHECK_LAUNCH();
  return SR_OK;
}
extern "C" int srhip_w2v_featln_bwd(const void* dy, const void* x, const void* pre, const float* mean, const float* rstd, const float* gamma, void* dpre,
                                    float* dgamma, float* dbeta, int B, int T, int P, int C, void* stream) {
  if (!dy || !x || !pre || !mean || !rstd || !dpre || !dgamma || !dbeta || B <= 0 || T <= 0 || P < T) return SR_EINVAL;
  const int M = B * P;
#define CALL(NV) SR_LAUNCH(featln_bwd_kernel<NV>, dim3(cdiv(M, 32)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x, \
                                    (const bf16_t*)pre, mean, rstd, gamma, (bf16_t*)dpre, dgamma, dbeta, T, P, M)
  W2V_NVC(C, CALL)
#undef CALL
  SR_CHECK_LAUNCH();
  return SR_OK;
}
