// WideResNet (classic_cv backbone, semilearn/nets/wrn/wrn.py) building blocks for gfx950.
// Activations are NHWC: a feature map is a row-major matrix [rows = B*H*W, C], so BatchNorm is a column statistic, a convolution is
// im2col (bf16) + srhip_gemm_nt, its weight gradient is srhip_gemm_tn_grouped_f32 on (dY, col) and its input gradient is a GEMM with
// the transposed filter followed by col2im.  This is the CPU-reference parity configuration (BASELINE.json configs[0]), not the
// throughput path: kernels are kept simple and HBM-bound.
#include "../../include/srhip.h"
#include "common.h"
#include "wrn_bn.h"

namespace {

// img fp32 [B, C, H, W] -> bf16 [B, H, W, C]
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int C, int HW2, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // over B*HW2*C
  if (i >= n) return;
  const int c = (int)(i % C);
  const size_t p = i / C;                                               // b * HW2 + pixel
  const size_t b = p / HW2, px = p % HW2;
  out[i] = f2bf(img[(b * C + c) * HW2 + px]);
}

// col[(b, yo, xo)][(i * k + j) * C + c] = act[b][yo*s + i - pad][xo*s + j - pad][c]  (0 outside the image, 0 for columns >= C*k*k).
// The K axis is TAP-major: the C channels of one filter tap are contiguous in col exactly as they are in the NHWC activation, so a thread
// moves V = 8 channels (16 bytes) per load / store and consecutive threads write consecutive 16-byte pieces of a col row.  The filter
// matrix is permuted to the same order once per step (conv_weight_prep) and its gradient back (add_unpad).
template <int V>
__global__ __launch_bounds__(256) void im2col_kernel(const bf16_t* __restrict__ act, bf16_t* __restrict__ col, int H, int W, int C, int ks,
                                                    int stride, int Ho, int Wo, int Kpad, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;          // over rows * (Kpad / V)
  if (idx >= total) return;
  const int kv = Kpad / V;
  const size_t row = idx / kv;                                          // b * Ho * Wo + yo * Wo + xo
  const int k = (int)(idx % kv) * V, K = C * ks * ks, pad = ks >> 1;
  const int b = (int)(row / (Ho * Wo)), r = (int)(row % (Ho * Wo)), yo = r / Wo, xo = r % Wo;
  if (V == 8) {
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (k < K) {
      const int t = k / C, c = k % C, y = yo * stride + t / ks - pad, x = xo * stride + t % ks - pad;
      if (y >= 0 && y < H && x >= 0 && x < W) v = *reinterpret_cast<const u32x4_t*>(act + (((size_t)b * H + y) * W + x) * C + c);
    }
    *reinterpret_cast<u32x4_t*>(col + row * Kpad + k) = v;
  } else {
    bf16_t v = 0;
    if (k < K) {
      const int t = k / C, c = k % C, y = yo * stride + t / ks - pad, x = xo * stride + t % ks - pad;
      if (y >= 0 && y < H && x >= 0 && x < W) v = act[(((size_t)b * H + y) * W + x) * C + c];
    }
    col[row * Kpad + k] = v;
  }
}

// The same col straight from the fp32 tensor in FRONT of the BatchNorm: col = im2col(bf16(LeakyReLU(BN(x)))) (mode 0: mean / invstd; mode 2:
// bf16(x)), bn_apply_kernel's arithmetic per element -- the backward's filter-gradient operand without materialising the activation first
// (a bn_act launch + 12 MB per convolution).  C % 8 == 0, C <= 256.
__global__ __launch_bounds__(256) void im2col_bn_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ isd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float slope, int mode,
                                                       bf16_t* __restrict__ col, int H, int W, int C, int ks, int stride, int Ho, int Wo,
                                                       int Kpad, size_t total) {
  __shared__ float prm[4][256];
  if (mode != 2) {
    for (int c = threadIdx.x; c < C; c += 256) { prm[0][c] = mean[c]; prm[1][c] = isd[c]; prm[2][c] = gamma[c]; prm[3][c] = beta[c]; }
    __syncthreads();
  }
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;          // over rows * (Kpad / 8)
  if (idx >= total) return;
  const int kv = Kpad / 8;
  const size_t row = idx / kv;
  const int k = (int)(idx % kv) * 8, K = C * ks * ks, pad = ks >> 1;
  const int b = (int)(row / (Ho * Wo)), r = (int)(row % (Ho * Wo)), yo = r / Wo, xo = r % Wo;
  u32x4_t o = {0u, 0u, 0u, 0u};
  if (k < K) {
    const int t = k / C, c = k % C, y = yo * stride + t / ks - pad, xx = xo * stride + t % ks - pad;
    if (y >= 0 && y < H && xx >= 0 && xx < W) {
      const float4* p = reinterpret_cast<const float4*>(x + (((size_t)b * H + y) * W + xx) * C + c);
      const float4 v0 = p[0], v1 = p[1];
      float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      if (mode != 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float yv = prm[2][c + e] * ((v[e] - prm[0][c + e]) * prm[1][c + e]) + prm[3][c + e];
          v[e] = yv > 0.f ? yv : slope * yv;
        }
      }
      o = u32x4_t{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
    }
  }
  *reinterpret_cast<u32x4_t*>(col + row * Kpad + k) = o;
}

// dact[b][y][x][c] (=|+=) sum over the (<= k*k) output positions that read this input pixel of dcol[(b,yo,xo)][(i*k + j) * C + c];
// a thread owns V channels of one input pixel (V = 4: 16-byte loads of dcol, consecutive threads consecutive channels)
template <int V>
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcol, float* __restrict__ dact, int H, int W, int C, int ks,
                                                    int stride, int Ho, int Wo, int Kpad, int accumulate, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;          // over B*H*W * (C / V)
  if (idx >= total) return;
  const int cv = C / V, c = (int)(idx % cv) * V, pad = ks >> 1;
  const size_t pix = idx / cv;                                          // b * H * W + y * W + x
  const int b = (int)(pix / (H * W)), r = (int)(pix % (H * W)), y = r / W, x = r % W;
  float s[V];
#pragma unroll
  for (int v = 0; v < V; ++v) s[v] = 0.f;
  for (int i = 0; i < ks; ++i) {
    const int ty = y + pad - i;
    if (ty < 0 || ty % stride) continue;
    const int yo = ty / stride;
    if (yo >= Ho) continue;
    for (int j = 0; j < ks; ++j) {
      const int tx = x + pad - j;
      if (tx < 0 || tx % stride) continue;
      const int xo = tx / stride;
      if (xo >= Wo) continue;
      const float* src = dcol + (((size_t)b * Ho + yo) * Wo + xo) * Kpad + (i * ks + j) * C + c;
      if (V == 4) {
        const float4 t = *reinterpret_cast<const float4*>(src);
        s[0] += t.x; s[1] += t.y; s[2] += t.z; s[3] += t.w;
      } else {
        s[0] += src[0];
      }
    }
  }
  float* d = dact + pix * C + c;
  if (V == 4) {
    float4 o = {s[0], s[1], s[2], s[3]};
    if (accumulate) { const float4 p = *reinterpret_cast<const float4*>(d); o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
    *reinterpret_cast<float4*>(d) = o;
  } else {
    *d = accumulate ? *d + s[0] : s[0];
  }
}

// W fp32 [Cout, C, k, k] -> Wb bf16 [Cout, Kpad] with the K axis tap-major ((i*k + j) * C + c, zero padded) and WbT bf16 [Kpad, Cout]
__global__ void conv_weight_prep_kernel(const float* __restrict__ Wf, bf16_t* __restrict__ Wb, bf16_t* __restrict__ WbT, int Cout, int C, int kk,
                                        int Kpad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Cout * Kpad) return;
  const int o = i / Kpad, k = i % Kpad, K = C * kk;
  const bf16_t v = k < K ? f2bf(Wf[(size_t)o * K + (k % C) * kk + k / C]) : (bf16_t)0;
  Wb[i] = v;
  WbT[(size_t)k * Cout + o] = v;
}

// dW (fp32 [Cout, C, k, k], +=) <- dWpad [Cout, Kpad] (tap-major K axis)
__global__ void add_unpad_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int C, int kk, int Kpad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int K = C * kk;
  if (i >= Cout * K) return;
  const int o = i / K, e = i % K, c = e / kk, t = e % kk;
  dst[i] += src[(size_t)o * Kpad + t * C + c];
}

// The same two passes for ALL convolutions of a network in one launch each (28 launches of 3-4 us per step and direction were each a
// dependent link of a latency-bound chain): entry p covers elements [start_p, start_{p+1}) of the flat index space.
__global__ __launch_bounds__(256) void conv_weight_prep_grouped_kernel(const srhip_conv_desc* __restrict__ d, int n, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int lo = 0, hi = n - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (i >= d[mid].start) lo = mid; else hi = mid - 1; }
  const srhip_conv_desc e = d[lo];
  const int j = (int)(i - e.start), o = j / e.Kpad, k = j % e.Kpad, K = e.C * e.kk;
  const bf16_t v = k < K ? f2bf(((const float*)e.a)[(size_t)o * K + (k % e.C) * e.kk + k / e.C]) : (bf16_t)0;
  ((bf16_t*)e.b)[j] = v;
  ((bf16_t*)e.c)[(size_t)k * e.Cout + o] = v;
}
// The filter of the INPUT-gradient convolution of a stride-1 3x3 layer: dX = conv(dY, W') with W'[ci][(8 - t) * Cout + co] = W[co][ci][t]
// (taps rotated by 180 degrees, channels swapped), bf16, K axis padded to Kpad = round32(9 * Cout); e.C = Cin, e.Cout = Cout, e.kk = 9.
__global__ __launch_bounds__(256) void conv_weight_flip_grouped_kernel(const srhip_conv_desc* __restrict__ d, int n, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int lo = 0, hi = n - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (i >= d[mid].start) lo = mid; else hi = mid - 1; }
  const srhip_conv_desc e = d[lo];
  const int j = (int)(i - e.start), ci = j / e.Kpad, k = j % e.Kpad, K = e.Cout * e.kk;
  bf16_t v = 0;
  if (k < K) {
    const int tf = k / e.Cout, co = k % e.Cout, t = e.kk - 1 - tf;
    v = f2bf(((const float*)e.a)[((size_t)co * e.C + ci) * e.kk + t]);
  }
  ((bf16_t*)e.b)[j] = v;
}
__global__ __launch_bounds__(256) void add_unpad_grouped_kernel(const srhip_conv_desc* __restrict__ d, int n, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int lo = 0, hi = n - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (i >= d[mid].start) lo = mid; else hi = mid - 1; }
  const srhip_conv_desc e = d[lo];
  const int j = (int)(i - e.start), K = e.C * e.kk, o = j / K, q = j % K, c = q / e.kk, t = q % e.kk;
  ((float*)e.b)[j] += ((const float*)e.a)[(size_t)o * e.Kpad + t * e.C + c];
}

// ---- BatchNorm over the rows of x fp32 [rows, C] --------------------------------------------------------------------------------
// column sums in double: ws[0..C) = sum, ws[C..2C) = sum of squares (forward) / sum dy', sum dy' * xhat (backward).
// A workgroup covers rows_per_block rows; a thread owns V consecutive channels (V = 4: one 16-byte load per row) of every (256 / (C / V))-th
// row, four rows requested before the first is used; the per-thread partials meet in LDS and the workgroup adds its 2C sums into an
// accumulator copy; the last workgroup folds the copies (wrn_bn.h) and, for srhip_bn_stats, derives mean / invstd / running statistics.
template <bool BWD, int V>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dact, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float slope, double* __restrict__ ws, int rows, int C,
                                                       int rows_per_block, const BnFinal fin, double* __restrict__ acc_only) {
  __shared__ double red[2][256 * V];
  __shared__ double red2c[512];
  __shared__ int is_last;
  const int cl = C / V;                                   // threads along a row
  const int c0 = (threadIdx.x % cl) * V, lane_row = threadIdx.x / cl, rpb = 256 / cl;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  double s1[V], s2[V];
  float mu[V], is[V], g[V], bt[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    s1[v] = 0.0; s2[v] = 0.0;
    if (BWD) { mu[v] = mean[c0 + v]; is[v] = invstd[c0 + v]; g[v] = gamma[c0 + v]; bt[v] = beta[c0 + v]; }
  }
  constexpr int U = 4;                                    // rows in flight per thread
  for (int rb = r0 + lane_row; rb < r1; rb += U * rpb) {
    float xv[U][V], dv[U][V];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = rb + u * rpb;
#pragma unroll
      for (int v = 0; v < V; ++v) { xv[u][v] = 0.f; dv[u][v] = 0.f; }
      if (r < r1) {
        if (V == 4) {
          const float4 t = *reinterpret_cast<const float4*>(x + (size_t)r * C + c0);
          xv[u][0] = t.x; xv[u][1] = t.y; xv[u][2] = t.z; xv[u][3] = t.w;
          if (BWD) {
            const float4 w = *reinterpret_cast<const float4*>(dact + (size_t)r * C + c0);
            dv[u][0] = w.x; dv[u][1] = w.y; dv[u][2] = w.z; dv[u][3] = w.w;
          }
        } else {
          xv[u][0] = x[(size_t)r * C + c0];
          if (BWD) dv[u][0] = dact[(size_t)r * C + c0];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (rb + u * rpb >= r1) continue;
#pragma unroll
      for (int v = 0; v < V; ++v) {
        if (!BWD) {
          s1[v] += (double)xv[u][v]; s2[v] += (double)xv[u][v] * (double)xv[u][v];
        } else {
          const float xh = (xv[u][v] - mu[v]) * is[v], yv = g[v] * xh + bt[v];
          const float dy = dv[u][v] * (yv > 0.f ? 1.0f : slope);
          s1[v] += (double)dy; s2[v] += (double)dy * (double)xh;
        }
      }
    }
  }
#pragma unroll
  for (int v = 0; v < V; ++v) { red[0][lane_row * C + c0 + v] = s1[v]; red[1][lane_row * C + c0 + v] = s2[v]; }
  __syncthreads();
  for (int o = threadIdx.x; o < 2 * C; o += 256) {
    const int which = o / C, c = o % C;
    double t = 0.0;
    for (int j = 0; j < rpb; ++j) t += red[which][j * C + c];
    red2c[o] = t;
  }
  __syncthreads();
  if (acc_only) {              // the caller folds (after an all-reduce over the ranks, or in the prologue of the launch that reads x)
    double* acc = acc_only + (size_t)(blockIdx.x % BN_COPIES) * 2 * C;
    for (int o = threadIdx.x; o < 2 * C; o += 256) unsafeAtomicAdd(acc + o, red2c[o]);
    return;
  }
  double* acc = ws + BN_WS_ACC + (size_t)(blockIdx.x % BN_COPIES) * 512;
  for (int o = threadIdx.x; o < 2 * C; o += 256) unsafeAtomicAdd(acc + o, red2c[o]);
  if (bn_arrive_and_fold(ws, red2c, C, gridDim.x, blockIdx.x, &is_last)) bn_finalize(red2c, C, rows, fin);
}

// the copies of an accumulator [BN_COPIES][2C] -> totals (optional, 2C doubles) and mean / invstd / running statistics (one workgroup)
__global__ __launch_bounds__(256) void bn_fold_kernel(const double* __restrict__ acc, double* __restrict__ totals, int rows, int C,
                                                     const BnFinal fin) {
  __shared__ double red2c[512];
  for (int o = threadIdx.x; o < 2 * C; o += 256) {
    double u[BN_COPIES], t = 0.0;
#pragma unroll
    for (int q = 0; q < BN_COPIES; ++q) u[q] = acc[(size_t)q * 2 * C + o];
#pragma unroll
    for (int q = 0; q < BN_COPIES; ++q) t += u[q];
    red2c[o] = t;
    if (totals) totals[o] = t;
  }
  __syncthreads();
  bn_finalize(red2c, C, rows, fin);
}

// rows per workgroup of bn_reduce_kernel: a multiple of the rows one pass of the workgroup covers, ~512 workgroups, at most 1024 rows
static inline int bn_rows_per_block(int rows, int C, int V) {
  const int pass = 256 / (C / V);
  int rpb = ((rows / 512 + pass - 1) / pass) * pass;
  if (rpb < pass) rpb = pass;
  if (rpb > 1024) rpb = 1024;
  return rpb;
}

// training forward: statistics of this batch -> (save_mean, save_invstd), optional running update (unbiased variance), activation
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const double* __restrict__ ws, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps, float slope, float momentum, int update_running,
                                                      float* __restrict__ running_mean, float* __restrict__ running_var,
                                                      float* __restrict__ save_mean, float* __restrict__ save_invstd, int use_running,
                                                      bf16_t* __restrict__ act, float* __restrict__ act_f32, int rows, int C) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)rows * C) return;
  const int c = (int)(i % C);
  float mu, var;
  if (use_running) {
    mu = running_mean[c]; var = running_var[c];
  } else {
    const double m = ws[c] / rows, v = ws[C + c] / rows - m * m;
    mu = (float)m; var = (float)(v > 0.0 ? v : 0.0);
  }
  const float is = 1.0f / sqrtf(var + eps);
  if (!use_running && i < (size_t)C) {            // the first C threads of the grid own one channel each
    save_mean[c] = mu; save_invstd[c] = is;
    if (update_running) {
      const double m = ws[c] / rows, v = ws[C + c] / rows - m * m;
      const float unb = (float)((v > 0.0 ? v : 0.0) * ((double)rows / (double)(rows > 1 ? rows - 1 : 1)));
      running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mu;
      running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unb;
    }
  }
  const float yv = gamma[c] * ((x[i] - mu) * is) + beta[c];
  const float a = yv > 0.f ? yv : slope * yv;
  if (act) act[i] = f2bf(a);
  if (act_f32) act_f32[i] = a;
}

// dx = resid + gamma * invstd * (dy' - mean(dy') - xhat * mean(dy' * xhat));  the first C threads add dgamma / dbeta
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dact, const double* __restrict__ ws,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float slope,
                                                          const float* __restrict__ resid, float* __restrict__ dx, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, int rows, int C, const double* __restrict__ wl,
                                                          double rows_total, bf16_t* __restrict__ dx_bf16) {
  // ws: the sums over ALL rows the statistics were taken over (rows_total of them: this rank's, or every rank's under SyncBatchNorm);
  // wl: this rank's own sums -- d(gamma), d(beta) are per-rank quantities (the data-parallel exchange averages them afterwards)
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)rows * C) return;
  const int c = (int)(i % C);
  const float mu = mean[c], is = invstd[c], g = gamma[c];
  const float xh = (x[i] - mu) * is, yv = g * xh + beta[c];
  const float dy = dact[i] * (yv > 0.f ? 1.0f : slope);
  const float m1 = (float)(ws[c] / rows_total), m2 = (float)(ws[C + c] / rows_total);
  const float v = g * is * (dy - m1 - xh * m2);
  const float o = resid ? resid[i] + v : v;
  dx[i] = o;
  if (dx_bf16) dx_bf16[i] = f2bf(o);                 // the GEMM operand of the convolution backward that reads dx next (saves its cast launch)
  if (i < (size_t)C) { dbeta[c] += (float)wl[c]; dgamma[c] += (float)wl[C + c]; }
}

// feat[b][c] = mean over the HW2 pixels of act_f32[b][.][c];  backward: dact[b][p][c] = dfeat[b][c] / HW2
__global__ void avgpool_fwd_kernel(const float* __restrict__ a, float* __restrict__ feat, int HW2, int C) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int p = 0; p < HW2; ++p) s += a[((size_t)b * HW2 + p) * C + c];
    feat[(size_t)b * C + c] = s / HW2;
  }
}
__global__ void avgpool_bwd_kernel(const float* __restrict__ dfeat, float* __restrict__ da, int HW2, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // over B*HW2*C
  const int c = (int)(i % C);
  const size_t b = i / ((size_t)HW2 * C);
  da[i] = dfeat[b * C + c] / HW2;
}

// classifier: logits[b][k] = feat[b] . Wc[k] + bc[k].  One wave per output, lanes along F (coalesced rows of Wc); grid = (B, ceil(K / 4)).
__global__ __launch_bounds__(256) void fc_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ Wc, const float* __restrict__ bc,
                                                    float* __restrict__ logits, int F, int K) {
  const int b = blockIdx.x, k = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (k >= K) return;
  float s = 0.f;
  for (int f = lane; f < F; f += 64) s += feat[(size_t)b * F + f] * Wc[(size_t)k * F + f];
  s = wave_sum(s);
  if (lane == 0) logits[(size_t)b * K + k] = s + bc[k];
}
// dfeat[b][f] = sum_k dlogits[b][k] Wc[k][f]        (grid = (B, ceil(F / 128)); 512 threads = 4 k-groups x 128 columns, rows of Wc read coalesced)
// The k loop is a chain of loads: with one accumulator per thread the 768 x 768 hidden layer of the encoders' classifier (bert.py:17-20) took
// 183 us per launch for 1.2 MFLOP; four k-groups with four independent partial sums each keep 16 rows of Wc in flight per column.
__global__ __launch_bounds__(512) void fc_bwd_x_kernel(const float* __restrict__ dl, const float* __restrict__ Wc, float* __restrict__ dfeat, int F, int K) {
  __shared__ float part[3][128];
  const int b = blockIdx.x, fl = threadIdx.x & 127, kg = threadIdx.x >> 7, f = blockIdx.y * 128 + fl;
  const int kq = (K + 3) / 4, k0 = kg * kq, k1 = min(K, k0 + kq);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (f < F) {
    const float* d = dl + (size_t)b * K;
    int k = k0;
    for (; k + 3 < k1; k += 4) {
      s0 += d[k] * Wc[(size_t)k * F + f];
      s1 += d[k + 1] * Wc[(size_t)(k + 1) * F + f];
      s2 += d[k + 2] * Wc[(size_t)(k + 2) * F + f];
      s3 += d[k + 3] * Wc[(size_t)(k + 3) * F + f];
    }
    for (; k < k1; ++k) s0 += d[k] * Wc[(size_t)k * F + f];
  }
  const float s = (s0 + s1) + (s2 + s3);
  if (kg > 0) part[kg - 1][fl] = s;
  __syncthreads();
  if (kg == 0 && f < F) dfeat[(size_t)b * F + f] = ((s + part[0][fl]) + (part[1][fl] + part[2][fl]));
}
// dWc[k][f] += sum_b dlogits[b][k] feat[b][f];  dbc[k] += sum_b dlogits[b][k]      (grid = K)
__global__ void fc_bwd_w_kernel(const float* __restrict__ dl, const float* __restrict__ feat, float* __restrict__ dWc, float* __restrict__ dbc, int B,
                                int F, int K) {
  const int k = blockIdx.x;
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dl[(size_t)b * K + k] * feat[(size_t)b * F + f];
    dWc[(size_t)k * F + f] += s;
  }
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dl[(size_t)b * K + k];
    dbc[k] += s;
  }
}

// SGD with Nesterov momentum and weight decay on a flat fp32 block; per-chunk weight-decay flag (table of chunk ends).
struct SgdChunk { long long end; float wd; float pad; };
__global__ __launch_bounds__(256) void sgd_flat_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ buf, float* __restrict__ ema,
                                                      const SgdChunk* __restrict__ table, int nchunks, long long n, float lr, float momentum,
                                                      float grad_scale, const float* __restrict__ clip_coef, float ema_m, float ema_1m, int first,
                                                      int zero_grad) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int lo = 0, hi = nchunks - 1;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (i < table[mid].end) hi = mid; else lo = mid + 1; }
  const float wd = table[lo].wd;
  float d = g[i] * (clip_coef ? grad_scale * clip_coef[0] : grad_scale);
  const float pv = p[i];
  if (wd != 0.f) d += wd * pv;
  const float b = first ? d : momentum * buf[i] + d;
  buf[i] = b;
  const float np = pv - lr * (d + momentum * b);
  p[i] = np;
  if (ema) ema[i] = __fadd_rn(__fmul_rn(ema_1m, np), __fmul_rn(ema_m, ema[i]));      // misc.py:154, op for op
  if (zero_grad) g[i] = 0.f;
}

}  // namespace

#define LAUNCH1D(kern, n, ...) SR_LAUNCH(kern, dim3(cdiv((long)(n), 256)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__)

extern "C" int srhip_nchw_to_nhwc_bf16(const float* img, void* out, int B, int C, int H, int W, void* stream) {
  if (!img || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0) return SR_EINVAL;
  const long n = (long)B * C * H * W;
  SR_LAUNCH(nchw_to_nhwc_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, img, (bf16_t*)out, C, H * W, (size_t)n);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_im2col(const void* act, void* col, int B, int H, int W, int C, int ksize, int stride, int Kpad, void* stream) {
  if (!act || !col || B <= 0 || (ksize != 1 && ksize != 3) || stride <= 0 || Kpad < C * ksize * ksize || (Kpad % 32)) return SR_EINVAL;
  const int pad = ksize >> 1, Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
  const size_t rows = (size_t)B * Ho * Wo;
  if (C % 8 == 0) {
    const size_t total = rows * (Kpad / 8);
    SR_LAUNCH(im2col_kernel<8>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)act, (bf16_t*)col, H, W, C,
                       ksize, stride, Ho, Wo, Kpad, total);
  } else {
    const size_t total = rows * Kpad;
    SR_LAUNCH(im2col_kernel<1>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)act, (bf16_t*)col, H, W, C,
                       ksize, stride, Ho, Wo, Kpad, total);
  }
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_im2col_bn(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                               int mode, void* col, int B, int H, int W, int C, int ksize, int stride, int Kpad, void* stream) {
  if (!x || !col || B <= 0 || (ksize != 1 && ksize != 3) || stride <= 0 || Kpad < C * ksize * ksize || (Kpad % 32) || (C % 8) || C > 256 ||
      (mode != 0 && mode != 2))
    return SR_EINVAL;
  if (mode == 0 && (!mean || !invstd || !gamma || !beta)) return SR_EINVAL;
  const int pad = ksize >> 1, Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
  const size_t total = (size_t)B * Ho * Wo * (Kpad / 8);
  SR_LAUNCH(im2col_bn_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, mean, invstd, gamma, beta, slope, mode,
                     (bf16_t*)col, H, W, C, ksize, stride, Ho, Wo, Kpad, total);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_col2im(const float* dcol, float* dact, int B, int H, int W, int C, int ksize, int stride, int Kpad, int accumulate,
                            void* stream) {
  if (!dcol || !dact || B <= 0 || (ksize != 1 && ksize != 3) || stride <= 0 || Kpad < C * ksize * ksize) return SR_EINVAL;
  const int pad = ksize >> 1, Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
  const size_t pix = (size_t)B * H * W;
  if (C % 4 == 0 && Kpad % 4 == 0) {
    const size_t total = pix * (C / 4);
    SR_LAUNCH(col2im_kernel<4>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dcol, dact, H, W, C, ksize, stride, Ho, Wo,
                       Kpad, accumulate, total);
  } else {
    const size_t total = pix * C;
    SR_LAUNCH(col2im_kernel<1>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dcol, dact, H, W, C, ksize, stride, Ho, Wo,
                       Kpad, accumulate, total);
  }
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_conv_weight_prep(const float* Wf, void* Wb, void* WbT, int Cout, int C, int ksize, int Kpad, void* stream) {
  if (!Wf || !Wb || !WbT || Cout <= 0 || C <= 0 || ksize <= 0 || Kpad < C * ksize * ksize) return SR_EINVAL;
  LAUNCH1D(conv_weight_prep_kernel, (long)Cout * Kpad, Wf, (bf16_t*)Wb, (bf16_t*)WbT, Cout, C, ksize * ksize, Kpad);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_conv_weight_prep_grouped(const srhip_conv_desc* desc_dev, int n, long long total, void* stream) {
  if (!desc_dev || n <= 0 || total <= 0) return SR_EINVAL;
  LAUNCH1D(conv_weight_prep_grouped_kernel, total, desc_dev, n, total);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
extern "C" int srhip_conv_weight_flip_grouped(const srhip_conv_desc* desc_dev, int n, long long total, void* stream) {
  if (!desc_dev || n <= 0 || total <= 0) return SR_EINVAL;
  LAUNCH1D(conv_weight_flip_grouped_kernel, total, desc_dev, n, total);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
extern "C" int srhip_add_unpad_grouped(const srhip_conv_desc* desc_dev, int n, long long total, void* stream) {
  if (!desc_dev || n <= 0 || total <= 0) return SR_EINVAL;
  LAUNCH1D(add_unpad_grouped_kernel, total, desc_dev, n, total);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_add_unpad(const float* src, float* dst, int Cout, int C, int ksize, int Kpad, void* stream) {
  if (!src || !dst || Cout <= 0 || C <= 0 || ksize <= 0 || Kpad < C * ksize * ksize) return SR_EINVAL;
  LAUNCH1D(add_unpad_kernel, (long)Cout * C * ksize * ksize, src, dst, Cout, C, ksize * ksize, Kpad);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

static void launch_bn_stats(const float* x, double* ws, int rows, int C, const BnFinal& fin, hipStream_t s, double* acc_only = nullptr) {
  if (C % 4 == 0) {
    const int rpb = bn_rows_per_block(rows, C, 4);
    SR_LAUNCH((bn_reduce_kernel<false, 4>), dim3(cdiv(rows, rpb)), dim3(256), 0, s, x, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f,
                       ws, rows, C, rpb, fin, acc_only);
  } else {
    const int rpb = bn_rows_per_block(rows, C, 1);
    SR_LAUNCH((bn_reduce_kernel<false, 1>), dim3(cdiv(rows, rpb)), dim3(256), 0, s, x, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f,
                       ws, rows, C, rpb, fin, acc_only);
  }
}

extern "C" long long srhip_bn_ws_doubles(void) { return (long long)BN_WS_ACC + (long long)BN_COPIES * 512; }

extern "C" int srhip_bn_fwd(const float* x, const float* gamma, const float* beta, float eps, float slope, float momentum, int training,
                            int update_running, float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                            void* act_bf16, float* act_f32, double* ws, int rows, int C, void* stream) {
  if (!x || !gamma || !beta || !running_mean || !running_var || rows <= 0 || C <= 0 || C > 256 || (256 % C)) return SR_EINVAL;
  if (training && (!save_mean || !save_invstd || !ws)) return SR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (training) {
    launch_bn_stats(x, ws, rows, C, BnFinal{}, s);
    SR_CHECK_LAUNCH();
  }
  SR_LAUNCH(bn_apply_kernel, dim3(cdiv((long)rows * C, 256)), dim3(256), 0, s, x, ws, gamma, beta, eps, slope, momentum, update_running,
                     running_mean, running_var, save_mean, save_invstd, !training, (bf16_t*)act_bf16, act_f32, rows, C);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_bn_stats(const float* x, float eps, float momentum, int update_running, float* running_mean, float* running_var,
                              float* out_mean, float* out_invstd, double* ws, int rows, int C, void* stream) {
  if (!x || !out_mean || !out_invstd || !ws || rows <= 0 || C <= 0 || C > 256 || (256 % C)) return SR_EINVAL;
  if (update_running && (!running_mean || !running_var)) return SR_EINVAL;
  BnFinal fin;
  fin.out_mean = out_mean; fin.out_invstd = out_invstd; fin.running_mean = running_mean; fin.running_var = running_var;
  fin.momentum = momentum; fin.update_running = update_running; fin.eps = eps;
  launch_bn_stats(x, ws, rows, C, fin, (hipStream_t)stream);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_bn_accumulate(const float* x, double* acc, int rows, int C, void* stream) {
  if (!x || !acc || rows <= 0 || C <= 0 || C > 256 || (256 % C)) return SR_EINVAL;
  launch_bn_stats(x, nullptr, rows, C, BnFinal{}, (hipStream_t)stream, acc);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_bn_fold(const double* acc, double rows_total, float eps, float momentum, int update_running, float* running_mean,
                             float* running_var, float* out_mean, float* out_invstd, double* totals, int C, void* stream) {
  if (!acc || rows_total < 1.0 || C <= 0 || C > 256 || (!out_mean && !totals) || (out_mean && !out_invstd)) return SR_EINVAL;
  if (update_running && (!running_mean || !running_var)) return SR_EINVAL;
  BnFinal fin;
  fin.out_mean = out_mean; fin.out_invstd = out_invstd; fin.running_mean = running_mean; fin.running_var = running_var;
  fin.momentum = momentum; fin.update_running = update_running; fin.eps = eps;
  SR_LAUNCH(bn_fold_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, acc, totals, (int)rows_total, C, fin);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_bn_bwd_reduce(const float* dact, const float* x, const float* save_mean, const float* save_invstd, const float* gamma,
                                   const float* beta, float slope, double* ws, int rows, int C, void* stream) {
  if (!dact || !x || !save_mean || !save_invstd || !gamma || !beta || !ws || rows <= 0 || C <= 0 || C > 256 || (256 % C)) return SR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (C % 4 == 0) {
    const int rpb = bn_rows_per_block(rows, C, 4);
    SR_LAUNCH((bn_reduce_kernel<true, 4>), dim3(cdiv(rows, rpb)), dim3(256), 0, s, x, dact, save_mean, save_invstd, gamma, beta, slope, ws,
                       rows, C, rpb, BnFinal{}, nullptr);
  } else {
    const int rpb = bn_rows_per_block(rows, C, 1);
    SR_LAUNCH((bn_reduce_kernel<true, 1>), dim3(cdiv(rows, rpb)), dim3(256), 0, s, x, dact, save_mean, save_invstd, gamma, beta, slope, ws,
                       rows, C, rpb, BnFinal{}, nullptr);
  }
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_bn_bwd_apply(const float* dact, const float* x, const float* save_mean, const float* save_invstd, const float* gamma,
                                  const float* beta, float slope, const float* resid, float* dx, float* dgamma, float* dbeta,
                                  const double* totals, const double* local_totals, double rows_total, void* dx_bf16, int rows, int C,
                                  void* stream) {
  if (!dact || !x || !save_mean || !save_invstd || !gamma || !beta || !dx || !dgamma || !dbeta || !totals || rows <= 0 || C <= 0 ||
      rows_total < rows)
    return SR_EINVAL;
  SR_LAUNCH(bn_bwd_apply_kernel, dim3(cdiv((long)rows * C, 256)), dim3(256), 0, (hipStream_t)stream, x, dact, totals, save_mean,
                     save_invstd, gamma, beta, slope, resid, dx, dgamma, dbeta, rows, C, local_totals ? local_totals : totals, rows_total,
                     (bf16_t*)dx_bf16);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_bn_bwd(const float* dact, const float* x, const float* save_mean, const float* save_invstd, const float* gamma,
                            const float* beta, float slope, const float* resid, float* dx, float* dgamma, float* dbeta, double* ws, int rows,
                            int C, void* stream) {
  const int rc = srhip_bn_bwd_reduce(dact, x, save_mean, save_invstd, gamma, beta, slope, ws, rows, C, stream);
  if (rc != SR_OK) return rc;
  return srhip_bn_bwd_apply(dact, x, save_mean, save_invstd, gamma, beta, slope, resid, dx, dgamma, dbeta, ws, nullptr, (double)rows, nullptr,
                            rows, C, stream);
}

extern "C" int srhip_avgpool_fwd(const float* act, float* feat, int B, int HW2, int C, void* stream) {
  if (!act || !feat || B <= 0 || HW2 <= 0 || C <= 0) return SR_EINVAL;
  SR_LAUNCH(avgpool_fwd_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, act, feat, HW2, C);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
extern "C" int srhip_avgpool_bwd(const float* dfeat, float* dact, int B, int HW2, int C, void* stream) {
  if (!dfeat || !dact || B <= 0 || HW2 <= 0 || C <= 0 || ((long)B * HW2 * C) % 64) return SR_EINVAL;
  SR_LAUNCH(avgpool_bwd_kernel, dim3((long)B * HW2 * C / 64), dim3(64), 0, (hipStream_t)stream, dfeat, dact, HW2, C);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_fc_fwd(const float* feat, const float* Wc, const float* bc, float* logits, int B, int F, int K, void* stream) {
  if (!feat || !Wc || !bc || !logits || B <= 0 || F <= 0 || K <= 0) return SR_EINVAL;
  SR_LAUNCH(fc_fwd_kernel, dim3(B, cdiv(K, 4)), dim3(256), 0, (hipStream_t)stream, feat, Wc, bc, logits, F, K);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
extern "C" int srhip_fc_bwd(const float* dlogits, const float* feat, const float* Wc, float* dfeat, float* dWc, float* dbc, int B, int F, int K,
                            void* stream) {
  if (!dlogits || !feat || !Wc || !dfeat || !dWc || !dbc || B <= 0 || F <= 0 || K <= 0) return SR_EINVAL;
  SR_LAUNCH(fc_bwd_x_kernel, dim3(B, cdiv(F, 128)), dim3(512), 0, (hipStream_t)stream, dlogits, Wc, dfeat, F, K);
  SR_CHECK_LAUNCH();
  SR_LAUNCH(fc_bwd_w_kernel, dim3(K), dim3(128), 0, (hipStream_t)stream, dlogits, feat, dWc, dbc, B, F, K);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_sgd_flat(float* p, float* g, float* buf, float* ema, const void* chunk_table, int nchunks, long long n, float lr,
                              float momentum, float grad_scale, const float* clip_coef, double ema_m, int first_step, int zero_grad,
                              void* stream) {
  if (!p || !g || !buf || !chunk_table || nchunks <= 0 || n <= 0) return SR_EINVAL;
  LAUNCH1D(sgd_flat_kernel, n, p, g, buf, ema, (const SgdChunk*)chunk_table, nchunks, n, lr, momentum, grad_scale, clip_coef, (float)ema_m,
           (float)(1.0 - ema_m), first_step, zero_grad);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
