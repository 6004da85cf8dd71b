// HBM-bound backbone kernels: LayerNorm fwd/bwd, patch embedding fwd/bwd, cls-pool + head fwd/bwd,
// casts / transposes feeding the MFMA GEMM, DropPath mask generation.
//
// Reference call sites:
//   K1  PatchEmbed + cls + pos     semilearn/nets/vit/vit.py:39-44, :277-280
//   K2  nn.LayerNorm(eps=1e-6)     vit.py:135,150,268 (eps :222)
//   K7  x[:,0] -> head             vit.py:296-305
//   DropPath                       vit.py:148,161 (timm.models.layers.DropPath)
// All of these move a few bytes per FLOP, so the design rules are: one wave per row with fp32
// statistics via wave shuffles, 8-16 B per lane coalesced accesses, and fusing the dtype cast
// (fp32 residual stream -> bf16 GEMM operand) into the producing kernel.
#include "common.h"
#include "srhip.h"

namespace {

// ---------------------------------------------------------------------------------------------
// LayerNorm forward: x fp32 [M, D] -> out bf16 [M, D]; D = 128 * NV.  One wave per row.
// Two rows per wave, all loads of both rows (and the affine) requested before the first reduction: a wave that handles one row is a chain of
// exposed round trips (row, then gamma / beta) and the launch is bounded by wave turnover.  Lane -> column map and reduction order per row
// are unchanged (bit-identical results).
template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float eps, bf16_t* __restrict__ out,
                                                    float* __restrict__ mean, float* __restrict__ rstd, int M) {
  constexpr int D = NV * 128, RW = 2;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RW, lane = threadIdx.x & 63;
  if (row0 >= M) return;
  float2 v[RW][NV], g[NV], b[NV];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const float2* xr = reinterpret_cast<const float2*>(x + (size_t)min(row0 + r, M - 1) * D);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[r][i] = xr[i * 64 + lane];
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    g[i] = reinterpret_cast<const float2*>(gamma)[i * 64 + lane];
    b[i] = reinterpret_cast<const float2*>(beta)[i * 64 + lane];
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
    for (int i = 0; i < NV; ++i) { const float a = v[r][i].x - mu, c = v[r][i].y - mu; q += a * a + c * c; }
    const float rs = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
    uint32_t* orow = reinterpret_cast<uint32_t*>(out + (size_t)row * D);
#pragma unroll
    for (int i = 0; i < NV; ++i)
      orow[i * 64 + lane] = pack_bf2((v[r][i].x - mu) * rs * g[i].x + b[i].x, (v[r][i].y - mu) * rs * g[i].y + b[i].y);
    if (mean && lane == 0) { mean[row] = mu; rstd[row] = rs; }
  }
}

// LayerNorm backward: dx (fp32, +=) and dgamma/dbeta (fp32, atomic +=).  4 * RPW rows per workgroup: 32 for the large launches (fewer
// atomics), 8 for the 4112-row launches of the gradient images (129 workgroups walking 8 dependent rows each left half the chip idle: 21 us).
template <int NV, int RPW>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ x,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    const float* __restrict__ gamma, float* __restrict__ dx,
                                                    float* __restrict__ dgamma, float* __restrict__ dbeta, int M,
                                                    bf16_t* __restrict__ out_bf16, const float* __restrict__ row_scale, int rows_per_sample,
                                                    float* __restrict__ part, int n_rep) {
  constexpr int D = NV * 128;
  __shared__ float red[2][4][D];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float2 ag[NV], ab[NV], g[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    ag[i] = make_float2(0.f, 0.f); ab[i] = make_float2(0.f, 0.f);
    g[i] = reinterpret_cast<const float2*>(gamma)[i * 64 + lane];
  }
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = blockIdx.x * (4 * RPW) + wave * RPW + rr;
    if (row >= M) break;
    const float mu = mean[row], rs = rstd[row];
    const float2* xr = reinterpret_cast<const float2*>(x + (size_t)row * D);
    const uint32_t* dr = reinterpret_cast<const uint32_t*>(dy + (size_t)row * D);
    float2 xh[NV], gy[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float2 xv = xr[i * 64 + lane];
      const uint32_t d2 = dr[i * 64 + lane];
      const float d0 = bf2f((bf16_t)(d2 & 0xffff)), d1 = bf2f((bf16_t)(d2 >> 16));
      xh[i] = make_float2((xv.x - mu) * rs, (xv.y - mu) * rs);
      gy[i] = make_float2(d0 * g[i].x, d1 * g[i].y);
      s1 += gy[i].x + gy[i].y;
      s2 += gy[i].x * xh[i].x + gy[i].y * xh[i].y;
      ag[i].x += d0 * xh[i].x; ag[i].y += d1 * xh[i].y;
      ab[i].x += d0; ab[i].y += d1;
    }
    const float c1 = wave_sum(s1) * (1.0f / D), c2 = wave_sum(s2) * (1.0f / D);
    float2* dxr = reinterpret_cast<float2*>(dx + (size_t)row * D);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float2 o = dxr[i * 64 + lane];
      o.x += rs * (gy[i].x - c1 - xh[i].x * c2);
      o.y += rs * (gy[i].y - c1 - xh[i].y * c2);
      dxr[i * 64 + lane] = o;
      // the updated gradient of the residual stream, DropPath-scaled and cast: the dY operand of the next branch's products
      // (was a separate cast_scale_rows launch after every LayerNorm backward: 24 latency-bound launches per step)
      if (out_bf16) {
        const float sc = row_scale ? row_scale[row / rows_per_sample] : 1.0f;
        reinterpret_cast<uint32_t*>(out_bf16 + (size_t)row * D)[i * 64 + lane] = pack_bf2(o.x * sc, o.y * sc);
      }
    }
  }
  // 514 workgroups adding into the same 2 * D words serialise at the memory side (device-scope atomics of 8 XCDs: 11 of the 19 us of a
  // 4112-row launch): with replicas, workgroup w adds into copy w % n_rep of [n_rep][2][D]; srhip_ln_grad_reduce folds them once per step.
  if (n_rep > 0) { dgamma = part + (size_t)(blockIdx.x % n_rep) * 2 * D; dbeta = dgamma + D; }
  if (!dgamma) return;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 2;
    red[0][wave][c] = ag[i].x; red[0][wave][c + 1] = ag[i].y;
    red[1][wave][c] = ab[i].x; red[1][wave][c + 1] = ab[i].y;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    atomicAdd(dgamma + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
    atomicAdd(dbeta + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
  }
}

// ---------------------------------------------------------------------------------------------
// Patch embedding (small patches: K = C*p*p <= 48 is an HBM-bound VALU op, not a GEMM).
// x[b, 0, :] = cls + pos[0];  x[b, 1+p, :] = patch(b,p) . Wp^T + bp + pos[1+p].   blockDim = D.
constexpr int PE_TOK = 32;
__global__ void patch_embed_fwd_kernel(const float* __restrict__ img, const int* __restrict__ img_index,
                                       const float* __restrict__ Wp, const float* __restrict__ bp,
                                       const float* __restrict__ cls, const float* __restrict__ pos,
                                       float* __restrict__ x, int C, int HW, int ps, int D) {
  extern __shared__ __attribute__((aligned(16))) float patch[];     // [PE_TOK][K]
  const int gw = HW / ps, N = gw * gw + 1, K = C * ps * ps;
  const int b = blockIdx.y, t0 = blockIdx.x * PE_TOK, d = threadIdx.x;
  const int bi = img_index ? img_index[b] : b;
  const float* im = img + (size_t)bi * C * HW * HW;
  for (int e = threadIdx.x; e < PE_TOK * K; e += blockDim.x) {
    const int tt = e / K, k = e % K, t = t0 + tt;
    float v = 0.f;
    if (t >= 1 && t < N) {
      const int p = t - 1, py = p / gw, px = p % gw;
      const int c = k / (ps * ps), i = (k / ps) % ps, j = k % ps;
      v = im[((size_t)c * HW + py * ps + i) * HW + px * ps + j];
    }
    patch[e] = v;
  }
  __syncthreads();
  const float* w = Wp + (size_t)d * K;
  const float bias = bp[d];
  const int nt = min(PE_TOK, N - t0);
  if (K <= 16 && (K & 3) == 0) {
    // K = 12 (3 x 2 x 2): the filter row lives in registers and the position rows of 8 tokens are requested together -- the token loop was
    // a chain of 32 dependent L2 round trips per thread (35-50 us for a launch whose output is 6 us of HBM time).  Same summation order.
    f32x4_t wr[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) wr[k] = (4 * k < K) ? reinterpret_cast<const f32x4_t*>(w)[k] : f32x4_t{0.f, 0.f, 0.f, 0.f};
    const float cl = cls[d];
    for (int tb = 0; tb < nt; tb += 8) {
      float pv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) pv[i] = pos[(size_t)min(t0 + tb + i, N - 1) * D + d];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int tt = tb + i, t = t0 + tt;
        if (tt < nt) {
          float acc;
          if (t == 0) {
            acc = cl;
          } else {
            acc = 0.f;
            const f32x4_t* p4 = reinterpret_cast<const f32x4_t*>(patch + tt * K);
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (4 * k < K) {
                const f32x4_t a = p4[k];
                acc += a[0] * wr[k][0]; acc += a[1] * wr[k][1]; acc += a[2] * wr[k][2]; acc += a[3] * wr[k][3];
              }
            acc += bias;
          }
          x[((size_t)b * N + t) * D + d] = acc + pv[i];
        }
      }
    }
    return;
  }
  for (int tt = 0; tt < nt; ++tt) {
    const int t = t0 + tt;
    float acc;
    if (t == 0) {
      acc = cls[d];
    } else {
      acc = 0.f;
      for (int k = 0; k < K; ++k) acc += patch[tt * K + k] * w[k];
      acc += bias;
    }
    x[((size_t)b * N + t) * D + d] = acc + pos[(size_t)t * D + d];
  }
}

// Two-stage, atomic-free form of the patch-embed weight gradient (srhip_patch_embed_bwd_ws): stage 1 = the sums of one (32-token chunk, image)
// workgroup into ws[wg][k][d] (k = K: the bias column); stage 2 = dWp[d][k] += sum over the workgroups.  The one-stage kernel ends every
// workgroup in D * (K + 1) same-address atomics: 80-100 us on the critical path of the backward for 19 MFLOP.
__global__ void patch_embed_bwd_part_kernel(const float* __restrict__ dx, const float* __restrict__ img, const int* __restrict__ img_index,
                                            float* __restrict__ ws, int C, int HW, int ps, int D) {
  extern __shared__ __attribute__((aligned(16))) float patch[];     // [PE_TOK][K]
  const int gw = HW / ps, N = gw * gw + 1, K = C * ps * ps;
  const int b = blockIdx.y, t0 = 1 + blockIdx.x * PE_TOK, d = threadIdx.x;
  const int bi = img_index ? img_index[b] : b;
  const float* im = img + (size_t)bi * C * HW * HW;
  const int nt = min(PE_TOK, N - t0);
  for (int e = threadIdx.x; e < nt * K; e += blockDim.x) {
    const int tt = e / K, k = e % K, p = t0 + tt - 1, py = p / gw, px = p % gw;
    const int c = k / (ps * ps), i = (k / ps) % ps, j = k % ps;
    patch[e] = im[((size_t)c * HW + py * ps + i) * HW + px * ps + j];
  }
  __syncthreads();
  const float* g0 = dx + ((size_t)b * N + t0) * D + d;
  float* out = ws + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (K + 1) * D + d;
  float accb = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {         // K is small; register-block 16 taps at a time
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    for (int tb = 0; tb < nt; tb += 8) {       // 8 gradient rows in flight (the loop is otherwise a chain of L2 round trips)
      float g[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) g[i] = (tb + i < nt) ? g0[(size_t)(tb + i) * D] : 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int tt = min(tb + i, nt - 1);
        if (k0 == 0) accb += g[i];
#pragma unroll
        for (int k = 0; k < 16; ++k)
          if (k0 + k < K) acc[k] += g[i] * patch[tt * K + k0 + k];
      }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (k0 + k < K) out[(size_t)(k0 + k) * D] = acc[k];
  }
  out[(size_t)K * D] = accb;
}
__global__ __launch_bounds__(256) void patch_embed_bwd_fold_kernel(const float* __restrict__ ws, float* __restrict__ dWp, float* __restrict__ dbp,
                                                                  int n_wg, int K, int D) {
  const int i = blockIdx.x * 256 + threadIdx.x;          // i = k * D + d
  if (i >= (K + 1) * D) return;
  const int k = i / D, d = i % D;
  float a = 0.f;
#pragma unroll 8
  for (int w = 0; w < n_wg; ++w) a += ws[(size_t)w * (K + 1) * D + i];
  if (k < K) dWp[(size_t)d * K + k] += a; else dbp[d] += a;
}

// dpos[t,d] += sum_b dx[b,t,d]; dcls[d] += sum_b dx[b,0,d].  grid = N, block = D.
__global__ void patch_embed_bwd_pos_kernel(const float* __restrict__ dx, float* __restrict__ dpos, float* __restrict__ dcls,
                                           int B, int N, int D) {
  const int t = blockIdx.x, d = threadIdx.x;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += dx[((size_t)b * N + t) * D + d];
  dpos[(size_t)t * D + d] += s;
  if (t == 0) dcls[d] += s;
}

// dWp[d,k] += sum_{b,p} dx[b,1+p,d] * patch[b,p,k]; dbp[d] += sum dx.   grid = (chunks, B), block = D.
// 128 tokens per workgroup: every workgroup ends in D * K same-address atomics, which serialise across workgroups (32 tokens per
// workgroup = 128 workgroups: 80 us for 19 MFLOP).
constexpr int PE_TOK_BWD = 128;
__global__ void patch_embed_bwd_w_kernel(const float* __restrict__ dx, const float* __restrict__ img,
                                         const int* __restrict__ img_index, float* __restrict__ dWp,
                                         float* __restrict__ dbp, int C, int HW, int ps, int D) {
  extern __shared__ __attribute__((aligned(16))) float patch[];     // [PE_TOK_BWD][K]
  const int gw = HW / ps, N = gw * gw + 1, K = C * ps * ps;
  const int b = blockIdx.y, t0 = 1 + blockIdx.x * PE_TOK_BWD, d = threadIdx.x;
  const int bi = img_index ? img_index[b] : b;
  const float* im = img + (size_t)bi * C * HW * HW;
  const int nt = min(PE_TOK_BWD, N - t0);
  for (int e = threadIdx.x; e < nt * K; e += blockDim.x) {
    const int tt = e / K, k = e % K, p = t0 + tt - 1, py = p / gw, px = p % gw;
    const int c = k / (ps * ps), i = (k / ps) % ps, j = k % ps;
    patch[e] = im[((size_t)c * HW + py * ps + i) * HW + px * ps + j];
  }
  __syncthreads();
  const float* g0 = dx + ((size_t)b * N + t0) * D + d;
  float accb = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {         // K is small; register-block 16 taps at a time
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
#pragma unroll 4
    for (int tt = 0; tt < nt; ++tt) {
      const float g = g0[(size_t)tt * D];
      if (k0 == 0) accb += g;
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (k0 + k < K) acc[k] += g * patch[tt * K + k0 + k];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (k0 + k < K) atomicAdd(dWp + (size_t)d * K + k0 + k, acc[k]);
  }
  atomicAdd(dbp + d, accb);
}

// ---------------------------------------------------------------------------------------------
// cls pooling + final LayerNorm + classifier head (fp32 throughout: tiny, and keeps logits tight).
// grid = B, block = 256.
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void cls_head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          const float* __restrict__ Wh, const float* __restrict__ bh,
                                                          float* __restrict__ feat, float* __restrict__ logits,
                                                          float* __restrict__ xhat, float* __restrict__ rstd,
                                                          float* __restrict__ feat_all, float* __restrict__ logits_all,
                                                          const long long* __restrict__ out_rows, int N, int D, int C) {
  extern __shared__ float f[];        // [D] + 4
  float* sh = f + D;
  const int b = blockIdx.x;
  // feat / logits [B, .] (may be NULL) are the launch's own dense outputs (the backward reads feat); feat_all / logits_all are the step's
  // [all (pass, image) rows, .] buffers, where image b of this launch is row out_rows[b]: written here instead of by an index_copy_ each
  const size_t ob = out_rows ? (size_t)out_rows[b] : (size_t)b;
  const float* xr = x + (size_t)b * N * D;     // token 0 of image b
  // the cls row stays in registers (D <= 1024: at most 4 values per thread): ONE global round trip instead of three dependent passes
  float xv[4], gv[4], bv[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int d = threadIdx.x + 256 * i;
    xv[i] = d < D ? xr[d] : 0.f;
    gv[i] = d < D ? gamma[d] : 0.f;
    bv[i] = d < D ? beta[d] : 0.f;
    s += xv[i];
  }
  const float mu = block_sum(s, sh) / D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) { const float a = (threadIdx.x + 256 * i < D) ? xv[i] - mu : 0.f; q += a * a; }
  const float rs = rsqrtf(block_sum(q, sh) / D + eps);
  // grid.y workgroups share an image: each normalises the cls row itself (384 floats) and takes every grid.y-th group of 4 classes
  // (one image per workgroup left 25 serial wave reductions per wave on 200 of 256 CUs: 63 us for 15 MFLOP)
  const bool first = blockIdx.y == 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int d = threadIdx.x + 256 * i;
    if (d < D) {
      const float xh = (xv[i] - mu) * rs, v = xh * gv[i] + bv[i];
      f[d] = v;
      if (first) {
        if (feat) feat[(size_t)b * D + d] = v;
        if (feat_all) feat_all[ob * D + d] = v;
        if (xhat) xhat[(size_t)b * D + d] = xh;
      }
    }
  }
  if (first && rstd && threadIdx.x == 0) rstd[b] = rs;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // two classes per iteration: the weight rows of the second travel while the first is reduced
  for (int c = blockIdx.y * 4 + wave; c < C; c += 8 * gridDim.y) {
    const int c2 = c + 4 * gridDim.y;
    const float* w = Wh + (size_t)c * D;
    const float* w2 = Wh + (size_t)min(c2, C - 1) * D;
    float a = 0.f, a2 = 0.f;
    for (int d = lane; d < D; d += 64) { const float fv = f[d]; a += fv * w[d]; a2 += fv * w2[d]; }
    a = wave_sum(a); a2 = wave_sum(a2);
    if (lane == 0) {
      if (logits) {
        logits[(size_t)b * C + c] = a + bh[c];
        if (c2 < C) logits[(size_t)b * C + c2] = a2 + bh[c2];
      }
      if (logits_all) {
        logits_all[ob * C + c] = a + bh[c];
        if (c2 < C) logits_all[ob * C + c2] = a2 + bh[c2];
      }
    }
  }
}

// backward A: grid = B.  dfeat = dlogits . Wh ; LN backward of the cls row -> dx[b, 0, :] (=), dgamma/dbeta (atomic +=)
__global__ __launch_bounds__(256) void cls_head_bwd_x_kernel(const float* __restrict__ dlogits, const float* __restrict__ Wh,
                                                            const float* __restrict__ gamma, const float* __restrict__ xhat,
                                                            const float* __restrict__ rstd, float* __restrict__ dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int N, int D, int C) {
  extern __shared__ float sm[];       // dl[C] + 4
  float* dl = sm;
  float* sh = sm + C;
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) dl[c] = dlogits[(size_t)b * C + c];
  __syncthreads();
  float gy[4], xh[4], df[4];
  float s1 = 0.f, s2 = 0.f;
  int n = 0;
  for (int d = threadIdx.x; d < D; d += 256, ++n) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) a += dl[c] * Wh[(size_t)c * D + d];
    df[n] = a; xh[n] = xhat[(size_t)b * D + d]; gy[n] = a * gamma[d];
    s1 += gy[n]; s2 += gy[n] * xh[n];
  }
  const float c1 = block_sum(s1, sh) / D, c2 = block_sum(s2, sh) / D, rs = rstd[b];
  n = 0;
  for (int d = threadIdx.x; d < D; d += 256, ++n) {
    dx[(size_t)b * N * D + d] = rs * (gy[n] - c1 - xh[n] * c2);
    atomicAdd(dgamma + d, df[n] * xh[n]);
    atomicAdd(dbeta + d, df[n]);
  }
}

// backward B: grid = C.  dWh[c,:] += sum_b dlogits[b,c] * feat[b,:]; dbh[c] += sum_b dlogits[b,c]
__global__ __launch_bounds__(256) void cls_head_bwd_w_kernel(const float* __restrict__ dlogits, const float* __restrict__ feat,
                                                            float* __restrict__ dWh, float* __restrict__ dbh, int B, int D, int C) {
  const int c = blockIdx.x;
  for (int d = threadIdx.x; d < D; d += 256) {
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += dlogits[(size_t)b * C + c] * feat[(size_t)b * D + d];
    dWh[(size_t)c * D + d] += a;
  }
  if (threadIdx.x == 0) {
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += dlogits[(size_t)b * C + c];
    dbh[c] += a;
  }
}

// ---------------------------------------------------------------------------------------------
// fp32 [M, D] (x optional per-sample scale) -> bf16.  8 elements per thread.
__global__ void cast_scale_rows_kernel(const float* __restrict__ x, const float* __restrict__ scale, int rows_per_sample,
                                       bf16_t* __restrict__ out, size_t n8, int D) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
  float s = 1.0f;
  if (scale) s = scale[(i * 8 / D) / rows_per_sample];
  uint4 o = {pack_bf2(a.x * s, a.y * s), pack_bf2(a.z * s, a.w * s), pack_bf2(b.x * s, b.y * s), pack_bf2(b.z * s, b.w * s)};
  reinterpret_cast<uint4*>(out)[i] = o;
}

// Transpose in [M, C] (bf16 or fp32, row stride ld_in) -> out bf16 [C, Mp] (row stride ld_out), zero-filling
// columns m in [M, Mp).  Optional exact-erf GELU on the fly, optional column sums (bias gradients, atomic +=).
// 64 x 64 tile through LDS.
template <typename TIN, bool GELU>
__device__ __forceinline__ void transpose_tile(const TIN* __restrict__ in, int ld_in, bf16_t* __restrict__ out, int ld_out,
                                               int M, int Mp, float* __restrict__ colsum, int m0, int c0, float (*tile)[65]) {
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63, m = m0 + r;
    float v = 0.f;
    if (m < M) {
      if constexpr (sizeof(TIN) == 2) v = bf2f(in[(size_t)m * ld_in + c0 + c]);
      else v = (float)in[(size_t)m * ld_in + c0 + c];
      if (GELU) v = gelu_erf(v);
    }
    tile[r][c] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 32; e += 256) {
    const int c = e >> 5, rp = (e & 31) * 2, m = m0 + rp;
    if (m < Mp) *reinterpret_cast<uint32_t*>(out + (size_t)(c0 + c) * ld_out + m) = pack_bf2(tile[rp][c], tile[rp + 1][c]);
  }
  if (colsum && threadIdx.x < 64) {
    float s = 0.f;
    for (int r = 0; r < 64; ++r) s += tile[r][threadIdx.x];
    atomicAdd(colsum + c0 + threadIdx.x, s);
  }
}

// grid = (ceil(Mp/64), C/64)
template <typename TIN, bool GELU>
__global__ __launch_bounds__(256) void transpose_kernel(const TIN* __restrict__ in, int ld_in, bf16_t* __restrict__ out, int ld_out,
                                                       int M, int Mp, float* __restrict__ colsum) {
  __shared__ float tile[64][65];
  transpose_tile<TIN, GELU>(in, ld_in, out, ld_out, M, Mp, colsum, blockIdx.x * 64, blockIdx.y * 64, tile);
}

// Batched form: ONE launch over the 64x64 tiles of many independent transposes (the 4*depth weight copies refreshed
// after every optimizer step, and the 4*depth saved-activation transposes at the start of a backward).
__global__ __launch_bounds__(256) void transpose_batched_kernel(const srhip_transpose_desc* __restrict__ desc, int n) {
  __shared__ float tile[64][65];
  const int t = blockIdx.x;
  int p = 0;
  {  // binary search: the last entry whose tile_start <= t (a token-sliced table has hundreds of entries; a linear walk of dependent scalar
     // loads cost a workgroup as much as its product)
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (t >= desc[mid].tile_start) lo = mid; else hi = mid - 1; }
    p = lo;
  }
  const srhip_transpose_desc d = desc[p];
  const int local = t - d.tile_start, tm = (d.Mp + 63) / 64;
  const int m0 = (local % tm) * 64, c0 = (local / tm) * 64;
  if (d.in_is_f32) transpose_tile<float, false>((const float*)d.in, d.ld_in, (bf16_t*)d.out, d.ld_out, d.M, d.Mp, nullptr, m0, c0, tile);
  else if (d.apply_gelu) transpose_tile<bf16_t, true>((const bf16_t*)d.in, d.ld_in, (bf16_t*)d.out, d.ld_out, d.M, d.Mp, nullptr, m0, c0, tile);
  else transpose_tile<bf16_t, false>((const bf16_t*)d.in, d.ld_in, (bf16_t*)d.out, d.ld_out, d.M, d.Mp, nullptr, m0, c0, tile);
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, size_t n) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    const float4 a = *reinterpret_cast<const float4*>(x + i);
    uint2 o = {pack_bf2(a.x, a.y), pack_bf2(a.z, a.w)};
    *reinterpret_cast<uint2*>(out + i) = o;
  } else {
    for (size_t j = i; j < n; ++j) out[j] = f2bf(x[j]);
  }
}

// DropPath per-sample scales: out[l, j, b] = Bernoulli(1 - p_l) / (1 - p_l), counter-based hash RNG.
// cols != NULL: out is [depth, 2, n] and column q of it is column cols[q] of the [depth, 2, B] table (any selection / order of the columns of ONE
// draw, produced directly: the step's launch trains each take a contiguous slice instead of an index_select each)
__global__ void droppath_fill_kernel(float* __restrict__ out, const float* __restrict__ probs, int depth, int B,
                                     unsigned long long seed, const long long* __restrict__ cols, int n,
                                     const unsigned long long* __restrict__ seed_dev) {
  if (seed_dev) seed += seed_dev[0];                  // srhip_droppath_fill_cols_dyn: the step's seed from device memory (HIP graph replay) + a static offset
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= depth * 2 * n) return;
  long long col = o % n;
  if (cols) { col = cols[col]; if (col < 0 || col >= B) col = 0; }
  const int i = (o / n) * B + (int)col;
  const float p = probs[i / (2 * B)];
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
  const float keep = 1.0f - p;
  out[o] = (p <= 0.f) ? 1.0f : (u < keep ? 1.0f / keep : 0.0f);
}

}  // namespace

// =============================================================================================
// ---------------------------------------------------------------------------------------------
// Large-patch PatchEmbed (C * ps^2 > 64, e.g. ViT-S/16 at 224x224: K = 768): the conv with kernel = stride = ps is a GEMM of
// the unfolded patches with the [D, C*ps*ps] filter (vit.py:39-44).  im2col -> srhip_gemm_nt -> assemble (+ bias, pos, cls).
// out[b * Np + p][(c, i, j)] = img[idx[b]][c][py * ps + i][px * ps + j]  (bf16), (c, i, j) minor order == Conv2d weight.flatten(1)
__global__ __launch_bounds__(256) void patch_im2col_kernel(const float* __restrict__ img, const int* __restrict__ img_index,
                                                          bf16_t* __restrict__ out, int C, int HW, int ps) {
  const int gw = HW / ps, Np = gw * gw, K = C * ps * ps;
  const int p = blockIdx.x, b = blockIdx.y, py = p / gw, px = p % gw;
  const int bi = img_index ? img_index[b] : b;
  const float* im = img + (size_t)bi * C * HW * HW;
  bf16_t* o = out + ((size_t)b * Np + p) * K;
  for (int e = 2 * threadIdx.x; e < K; e += 512) {            // ps is even: the pair (j, j+1) stays inside one image row
    const int c = e / (ps * ps), i = (e / ps) % ps, j = e % ps;
    const float2 v = *reinterpret_cast<const float2*>(im + ((size_t)c * HW + py * ps + i) * HW + px * ps + j);
    *reinterpret_cast<uint32_t*>(o + e) = pack_bf2(v.x, v.y);
  }
}

// x[b, 0, :] = cls + pos[0];  x[b, 1 + p, :] = tok[b * Np + p, :] + bias + pos[1 + p]      (vit.py:277-280)
__global__ void patch_assemble_kernel(const float* __restrict__ tok, const float* __restrict__ bp, const float* __restrict__ cls,
                                      const float* __restrict__ pos, float* __restrict__ x, int Np, int D) {
  const int t = blockIdx.x, b = blockIdx.y, N = Np + 1;
  float* xr = x + ((size_t)b * N + t) * D;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    const float v = (t == 0) ? cls[d] : tok[((size_t)b * Np + t - 1) * D + d] + bp[d];
    xr[d] = v + pos[(size_t)t * D + d];
  }
}

// backward operand: the patch-token rows of dx (fp32 [B, N, D], cls row skipped) as bf16 [B * Np, D]
__global__ void patch_gather_grad_kernel(const float* __restrict__ dx, bf16_t* __restrict__ out, int Np, int D) {
  const int p = blockIdx.x, b = blockIdx.y;
  const float* src = dx + ((size_t)b * (Np + 1) + 1 + p) * D;
  bf16_t* o = out + ((size_t)b * Np + p) * D;
  for (int d = 2 * threadIdx.x; d < D; d += 2 * blockDim.x) *reinterpret_cast<uint32_t*>(o + d) = pack_bf2(src[d], src[d + 1]);
}

extern "C" int srhip_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, void* out,
                                   float* mean, float* rstd, int M, int D, void* stream) {
  if (M <= 0 || (D != 128 && D != 384 && D != 512 && D != 768) || ((mean == nullptr) != (rstd == nullptr))) return SR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(cdiv(M, 8)), block(256);
  if (D == 128) SR_LAUNCH(ln_fwd_kernel<1>, grid, block, 0, s, x, gamma, beta, eps, (bf16_t*)out, mean, rstd, M);
  else if (D == 512) SR_LAUNCH(ln_fwd_kernel<4>, grid, block, 0, s, x, gamma, beta, eps, (bf16_t*)out, mean, rstd, M);
  else if (D == 384) SR_LAUNCH(ln_fwd_kernel<3>, grid, block, 0, s, x, gamma, beta, eps, (bf16_t*)out, mean, rstd, M);
  else SR_LAUNCH(ln_fwd_kernel<6>, grid, block, 0, s, x, gamma, beta, eps, (bf16_t*)out, mean, rstd, M);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

static int layernorm_bwd_impl(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* dx, float* dgamma,
                              float* dbeta, void* out_bf16, const float* row_scale, int rows_per_sample, int M, int D, void* stream,
                              float* part = nullptr, int n_rep = 0) {
  if (M <= 0 || (D != 128 && D != 384 && D != 512 && D != 768) || (row_scale && rows_per_sample <= 0)) return SR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  static const int rpw_env = SR_TUNE_ENV("SRHIP_LNB_RPW") ? atoi(SR_TUNE_ENV("SRHIP_LNB_RPW")) : 0;
  const bool small = rpw_env ? rpw_env == 2 : M < 16384;
  dim3 grid(cdiv(M, small ? 8 : 32)), block(256);
  const int rps = rows_per_sample > 0 ? rows_per_sample : 1;
#define LNB(NV)                                                                                                                                   \
  do {                                                                                                                                            \
    if (small) SR_LAUNCH((ln_bwd_kernel<NV, 2>), grid, block, 0, s, (const bf16_t*)dy, x, mean, rstd, gamma, dx, dgamma, dbeta, M,        \
                                  (bf16_t*)out_bf16, row_scale, rps, part, n_rep);                                                                             \
    else SR_LAUNCH((ln_bwd_kernel<NV, 8>), grid, block, 0, s, (const bf16_t*)dy, x, mean, rstd, gamma, dx, dgamma, dbeta, M,              \
                            (bf16_t*)out_bf16, row_scale, rps, part, n_rep);                                                                                   \
  } while (0)
  if (D == 128) LNB(1); else if (D == 512) LNB(4); else if (D == 384) LNB(3); else LNB(6);
#undef LNB
  SR_CHECK_LAUNCH();
  return SR_OK;
}
extern "C" int srhip_layernorm_bwd(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                   float* dx, float* dgamma, float* dbeta, int M, int D, void* stream) {
  return layernorm_bwd_impl(dy, x, mean, rstd, gamma, dx, dgamma, dbeta, nullptr, nullptr, 0, M, D, stream);
}
extern "C" int srhip_layernorm_bwd_cast(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* dx,
                                        float* dgamma, float* dbeta, void* out_bf16, const float* row_scale, int rows_per_sample, int M, int D,
                                        void* stream) {
  if (!out_bf16) return SR_EINVAL;
  return layernorm_bwd_impl(dy, x, mean, rstd, gamma, dx, dgamma, dbeta, out_bf16, row_scale, rows_per_sample, M, D, stream);
}
extern "C" int srhip_layernorm_bwd_part(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* dx,
                                        float* part, int n_rep, void* out_bf16, const float* row_scale, int rows_per_sample, int M, int D,
                                        void* stream) {
  if (!part || n_rep <= 0) return SR_EINVAL;
  return layernorm_bwd_impl(dy, x, mean, rstd, gamma, dx, nullptr, nullptr, out_bf16, row_scale, rows_per_sample, M, D, stream, part, n_rep);
}

namespace {
// dgamma / dbeta of n_ln LayerNorms += the sum of their n_rep partial copies ([n_ln][n_rep][2][D]); the copies are cleared for the next step.
__global__ __launch_bounds__(256) void ln_grad_reduce_kernel(const srhip_ln_reduce_desc* __restrict__ desc, float* __restrict__ part, int n_rep,
                                                            int D) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= 2 * D) return;
  float* p = part + (size_t)blockIdx.y * n_rep * 2 * D + c;
  float acc = 0.f;
  for (int r = 0; r < n_rep; ++r) { acc += p[(size_t)r * 2 * D]; p[(size_t)r * 2 * D] = 0.f; }
  const srhip_ln_reduce_desc d = desc[blockIdx.y];
  float* dst = c < D ? d.dgamma + c : d.dbeta + (c - D);
  *dst += acc;
}
}  // namespace
extern "C" int srhip_ln_grad_reduce(const srhip_ln_reduce_desc* desc_dev, float* part, int n_ln, int n_rep, int D, void* stream) {
  if (!desc_dev || !part || n_ln <= 0 || n_rep <= 0 || D <= 0) return SR_EINVAL;
  SR_LAUNCH(ln_grad_reduce_kernel, dim3(cdiv(2 * D, 256), n_ln), dim3(256), 0, (hipStream_t)stream, desc_dev, part, n_rep, D);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_patch_embed_fwd(const float* img, const int* img_index, const float* Wp, const float* bp, const float* cls,
                                     const float* pos, float* x, int B, int C, int HW, int ps, int D, void* stream) {
  if (B <= 0 || HW % ps || D % 64 || D > 1024 || C * ps * ps > 64) return SR_EINVAL;
  const int gw = HW / ps, N = gw * gw + 1, K = C * ps * ps;
  SR_LAUNCH(patch_embed_fwd_kernel, dim3(cdiv(N, PE_TOK), B), dim3(D), PE_TOK * K * sizeof(float), (hipStream_t)stream,
                     img, img_index, Wp, bp, cls, pos, x, C, HW, ps, D);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_patch_embed_bwd(const float* dx, const float* img, const int* img_index, float* dWp, float* dbp,
                                     float* dcls, float* dpos, int B, int C, int HW, int ps, int D, void* stream) {
  if (B <= 0 || HW % ps || D % 64 || D > 1024 || C * ps * ps > 64) return SR_EINVAL;
  const int gw = HW / ps, N = gw * gw + 1, K = C * ps * ps;
  hipStream_t s = (hipStream_t)stream;
  SR_LAUNCH(patch_embed_bwd_pos_kernel, dim3(N), dim3(D), 0, s, dx, dpos, dcls, B, N, D);
  SR_CHECK_LAUNCH();
  SR_LAUNCH(patch_embed_bwd_w_kernel, dim3(cdiv(N - 1, PE_TOK_BWD), B), dim3(D), PE_TOK_BWD * K * sizeof(float), s, dx, img,
                     img_index, dWp, dbp, C, HW, ps, D);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" long srhip_patch_embed_bwd_ws_floats(int B, int C, int HW, int ps, int D) {
  if (B <= 0 || ps <= 0 || HW % ps) return -1;
  const int gw = HW / ps, K = C * ps * ps;
  return (long)cdiv(gw * gw, PE_TOK) * B * (K + 1) * D;
}
extern "C" int srhip_patch_embed_bwd_ws(const float* dx, const float* img, const int* img_index, float* dWp, float* dbp, float* dcls,
                                        float* dpos, float* ws, int B, int C, int HW, int ps, int D, void* stream) {
  if (B <= 0 || HW % ps || D % 64 || D > 1024 || C * ps * ps > 64 || !ws) return SR_EINVAL;
  const int gw = HW / ps, N = gw * gw + 1, K = C * ps * ps, nch = cdiv(N - 1, PE_TOK);
  hipStream_t s = (hipStream_t)stream;
  SR_LAUNCH(patch_embed_bwd_pos_kernel, dim3(N), dim3(D), 0, s, dx, dpos, dcls, B, N, D);
  SR_CHECK_LAUNCH();
  SR_LAUNCH(patch_embed_bwd_part_kernel, dim3(nch, B), dim3(D), PE_TOK * K * sizeof(float), s, dx, img, img_index, ws, C, HW, ps, D);
  SR_CHECK_LAUNCH();
  SR_LAUNCH(patch_embed_bwd_fold_kernel, dim3(cdiv((K + 1) * D, 256)), dim3(256), 0, s, ws, dWp, dbp, nch * B, K, D);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_cls_head_fwd(const float* x, const float* gamma, const float* beta, float eps, const float* Wh,
                                  const float* bh, float* feat, float* logits, float* xhat, float* rstd, int B, int N, int D,
                                  int C, void* stream) {
  return srhip_cls_head_fwd_scatter(x, gamma, beta, eps, Wh, bh, feat, logits, xhat, rstd, nullptr, nullptr, nullptr, B, N, D, C, stream);
}
extern "C" int srhip_cls_head_fwd_scatter(const float* x, const float* gamma, const float* beta, float eps, const float* Wh, const float* bh,
                                          float* feat, float* logits, float* xhat, float* rstd, float* feat_all, float* logits_all,
                                          const long long* out_rows, int B, int N, int D, int C, void* stream) {
  if (B <= 0 || D > 1024 || C <= 0) return SR_EINVAL;
  if ((!feat || !logits) && (!feat_all || !logits_all)) return SR_EINVAL;       // some complete (feat, logits) destination
  if ((feat_all || logits_all) && !out_rows) return SR_EINVAL;
  SR_LAUNCH(cls_head_fwd_kernel, dim3(B, C >= 32 ? 4 : 1), dim3(256), (D + 4) * sizeof(float), (hipStream_t)stream, x, gamma, beta, eps,
                     Wh, bh, feat, logits, xhat, rstd, feat_all, logits_all, out_rows, N, D, C);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_cls_head_bwd(const float* dlogits, const float* Wh, const float* gamma, const float* feat,
                                  const float* xhat, const float* rstd, float* dx, float* dWh, float* dbh, float* dgamma,
                                  float* dbeta, int B, int N, int D, int C, void* stream) {
  if (B <= 0 || D > 1024 || C <= 0 || (!dx && !dWh)) return SR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (dx) {               // per-image half: dx of the cls rows, final-norm affine gradients (atomic adds)
    SR_LAUNCH(cls_head_bwd_x_kernel, dim3(B), dim3(256), (C + 4) * sizeof(float), s, dlogits, Wh, gamma, xhat, rstd, dx,
                       dgamma, dbeta, N, D, C);
    SR_CHECK_LAUNCH();
  }
  if (dWh) {              // the half that sums over all images: head weight / bias gradients (plain +=: one launch per backward)
    SR_LAUNCH(cls_head_bwd_w_kernel, dim3(C), dim3(256), 0, s, dlogits, feat, dWh, dbh, B, D, C);
    SR_CHECK_LAUNCH();
  }
  return SR_OK;
}

extern "C" int srhip_cast_scale_rows(const float* x, const float* scale, int rows_per_sample, void* out, long M, int D,
                                     void* stream) {
  if (M <= 0 || D % 8 || (scale && rows_per_sample <= 0)) return SR_EINVAL;
  const size_t n8 = (size_t)M * D / 8;
  SR_LAUNCH(cast_scale_rows_kernel, dim3(cdiv(n8, 256)), dim3(256), 0, (hipStream_t)stream, x, scale,
                     rows_per_sample > 0 ? rows_per_sample : 1, (bf16_t*)out, n8, D);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_transpose_to_bf16(const void* in, int in_is_f32, int ld_in, void* out, int ld_out, int M, int Mp, int C,
                                       int apply_gelu, float* colsum, void* stream) {
  if (M <= 0 || Mp < M || (Mp % 2) || (C % 64) || ld_out < Mp || (ld_out % 2)) return SR_EINVAL;
  dim3 grid(cdiv(Mp, 64), C / 64), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (in_is_f32) {
    if (apply_gelu) return SR_EINVAL;
    SR_LAUNCH((transpose_kernel<float, false>), grid, block, 0, s, (const float*)in, ld_in, (bf16_t*)out, ld_out, M, Mp, colsum);
  } else if (apply_gelu) {
    SR_LAUNCH((transpose_kernel<bf16_t, true>), grid, block, 0, s, (const bf16_t*)in, ld_in, (bf16_t*)out, ld_out, M, Mp, colsum);
  } else {
    SR_LAUNCH((transpose_kernel<bf16_t, false>), grid, block, 0, s, (const bf16_t*)in, ld_in, (bf16_t*)out, ld_out, M, Mp, colsum);
  }
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_cast_f32_bf16(const float* x, void* out, long n, void* stream) {
  if (n <= 0) return SR_EINVAL;
  SR_LAUNCH(cast_f32_bf16_kernel, dim3(cdiv(cdiv(n, 4), 256)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)out, (size_t)n);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_droppath_fill(float* out, const float* probs, int depth, int B, unsigned long long seed, void* stream) {
  if (depth <= 0 || B <= 0) return SR_EINVAL;
  SR_LAUNCH(droppath_fill_kernel, dim3(cdiv((long)depth * 2 * B, 256)), dim3(256), 0, (hipStream_t)stream, out, probs, depth, B, seed,
                     (const long long*)nullptr, B, (const unsigned long long*)nullptr);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
extern "C" int srhip_droppath_fill_cols(float* out, const float* probs, const long long* cols, int depth, int B, int n_cols,
                                        unsigned long long seed, void* stream) {
  if (depth <= 0 || B <= 0 || n_cols <= 0 || !cols) return SR_EINVAL;
  SR_LAUNCH(droppath_fill_kernel, dim3(cdiv((long)depth * 2 * n_cols, 256)), dim3(256), 0, (hipStream_t)stream, out, probs, depth, B, seed,
                     cols, n_cols, (const unsigned long long*)nullptr);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
// seed = *seed_dev + seed_offset: the draw of a captured step follows the counter the host writes before every replay (cols may be NULL: all B columns)
extern "C" int srhip_droppath_fill_cols_dyn(float* out, const float* probs, const long long* cols, int depth, int B, int n_cols,
                                            const unsigned long long* seed_dev, unsigned long long seed_offset, void* stream) {
  if (depth <= 0 || B <= 0 || n_cols <= 0 || !seed_dev) return SR_EINVAL;
  SR_LAUNCH(droppath_fill_kernel, dim3(cdiv((long)depth * 2 * n_cols, 256)), dim3(256), 0, (hipStream_t)stream, out, probs, depth, B, seed_offset,
                     cols, cols ? n_cols : B, seed_dev);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_transpose_batched(const srhip_transpose_desc* desc_dev, int n, int total_tiles, void* stream) {
  if (!desc_dev || n <= 0 || total_tiles <= 0) return SR_EINVAL;
  SR_LAUNCH(transpose_batched_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, desc_dev, n);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_patch_im2col(const float* img, const int* img_index, void* out, int B, int C, int HW, int ps, void* stream) {
  if (!img || !out || B <= 0 || C <= 0 || ps <= 0 || (ps & 1) || HW % ps) return SR_EINVAL;
  const int gw = HW / ps;
  SR_LAUNCH(patch_im2col_kernel, dim3(gw * gw, B), dim3(256), 0, (hipStream_t)stream, img, img_index, (bf16_t*)out, C, HW, ps);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_patch_assemble(const float* tok, const float* bp, const float* cls, const float* pos, float* x, int B, int Np,
                                    int D, void* stream) {
  if (!tok || !bp || !cls || !pos || !x || B <= 0 || Np <= 0 || D <= 0) return SR_EINVAL;
  SR_LAUNCH(patch_assemble_kernel, dim3(Np + 1, B), dim3(D < 256 ? 64 : 256), 0, (hipStream_t)stream, tok, bp, cls, pos, x, Np, D);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_patch_grad_operands(const float* dx, void* dx_tok_bf16, float* dpos, float* dcls, int B, int Np, int D,
                                         void* stream) {
  if (!dx || !dx_tok_bf16 || !dpos || !dcls || B <= 0 || Np <= 0 || D <= 0 || (D & 1) || D > 1024) return SR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  SR_LAUNCH(patch_embed_bwd_pos_kernel, dim3(Np + 1), dim3(D), 0, s, dx, dpos, dcls, B, Np + 1, D);
  SR_CHECK_LAUNCH();
  SR_LAUNCH(patch_gather_grad_kernel, dim3(Np, B), dim3(D < 512 ? 64 : 256), 0, s, dx, (bf16_t*)dx_tok_bf16, Np, D);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
