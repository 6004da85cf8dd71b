// SemiReward Rewarder / Generator: forward, hand-written backward, Adam.
//
// Reference (math restated in SURVEY.md Appendix D):
//   Rewarder.forward   semilearn/algorithms/semireward/semireward.py:52-72
//   Generator.forward  semilearn/algorithms/semireward/semireward.py:21-24  (+ .long() srflexmatch.py:158-159)
//   SR update block    semilearn/algorithms/srflexmatch/srflexmatch.py:180-208  (two MSE losses, two backward(),
//                      Adam(lr=sr_lr) -- the generator's step is a no-op: its graph is cut by .long())
//
// Everything is fp32 (136 962 parameters, batches of 8..256 rows): latency-bound.  Forward design: a tile of
// 8 rows per workgroup sits transposed in LDS, every thread owns one output column and streams the TRANSPOSED
// weight matrix coalesced from L2 (8 FMAs per 4-byte load, no cross-lane reduction, loads independent of each
// other so they pipeline) -- the first version used one wave-reduction per output and was 25x slower because
// 449 dependent L2 round trips were serialised.  ALL independent groups (the K passes of one training step, each
// with its own softmax over its 2B rows) go through ONE launch pair.
// Parameter block: flat fp32 in the reference's named_parameters() order (offsets in RewOff).
#include "common.h"
#include "srhip.h"

namespace {

constexpr int E = 128;  // embedding / hidden width fixed by the reference (semireward.py:34,37,41)

struct RewOff {
  int Wf, bf, gf, bef, Emb, gl, bl, wa, ba, W1, b1, W2, b2, W3, b3, w4, b4, total;
  __host__ __device__ RewOff(int F, int L) {
    int o = 0;
    Wf = o; o += E * F; bf = o; o += E; gf = o; o += E; bef = o; o += E;
    Emb = o; o += L * E; gl = o; o += E; bl = o; o += E;
    wa = o; o += E; ba = o; o += 1;
    W1 = o; o += 256 * E; b1 = o; o += 256; W2 = o; o += E * 256; b2 = o; o += E;
    W3 = o; o += 64 * E; b3 = o; o += 64; w4 = o; o += 64; b4 = o; o += 1;
    total = o;
  }
};

// workspace layout (floats) for one call with G groups of B rows; R = G*B
struct RewWs {
  size_t z, slog, alpha, ctx, xhat, rstd, u, m1, m2, f1, r, dlogit, df1, dm2, dm1, du, dz, total;
  __host__ __device__ RewWs(size_t G, size_t B) {
    const size_t R = G * B;
    size_t o = 0;
    z = o; o += 2 * R * E; slog = o; o += 2 * R; alpha = o; o += 2 * R; ctx = o; o += G * E;
    xhat = o; o += 2 * R * E; rstd = o; o += 2 * R;
    u = o; o += R * E; m1 = o; o += R * 256; m2 = o; o += R * E; f1 = o; o += R * 64; r = o; o += R;
    dlogit = o; o += R; df1 = o; o += R * 64; dm2 = o; o += R * E; dm1 = o; o += R * 256; du = o; o += R * E;
    dz = o; o += 2 * R * E;
    total = o;
  }
};

// LDS hand-off between lanes of ONE wave: DS ops of a wave retire in issue order, so only the compiler
// has to be kept from moving accesses across the hand-off point.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// 128-wide LayerNorm of the two values per lane (j = lane, lane + 64); eps 1e-5 (nn.LayerNorm default)
__device__ __forceinline__ void wave_ln128(float v0, float v1, const float* g, const float* b, int lane, float& o0, float& o1,
                                           float& xh0, float& xh1, float& rs) {
  const float mu = wave_sum(v0 + v1) * (1.0f / E);
  const float a = v0 - mu, c = v1 - mu;
  rs = 1.0f / sqrtf(wave_sum(a * a + c * c) * (1.0f / E) + 1e-5f);
  xh0 = a * rs; xh1 = c * rs;
  o0 = xh0 * g[lane] + b[lane];
  o1 = xh1 * g[lane + 64] + b[lane + 64];
}

// ---- transposed-weight block (built by srhip_rewarder_prepare / srhip_generator_prepare) -------------------
// The forward streams W^T[k][j] so that consecutive threads (outputs j) read consecutive addresses while a
// tile of 8 input rows sits in LDS: 8 FMAs per 4-byte weight load, no cross-lane reductions at all.
struct RewTOff {
  int WfT, W1T, W2T, W3T, total;
  __host__ __device__ RewTOff(int F) {
    int o = 0;
    WfT = o; o += F * E; W1T = o; o += E * 256; W2T = o; o += 256 * E; W3T = o; o += E * 64;
    total = o;
  }
};
struct GenTOff {
  int W1T, W2T, W3T, total;
  __host__ __device__ GenTOff(int F) {
    int o = 0;
    W1T = o; o += F * 256; W2T = o; o += 256 * 128; W3T = o; o += 128 * 64;
    total = o;
  }
};

constexpr int RT = 8;   // rows per workgroup tile

// yT[j][r] = act(b[j] + sum_k xT[k][r] * WT[k*J + j]), r < 8.  256 threads; the k-range is split over 256/J
// thread groups and combined through `scratch` (>= 256*8 floats).  xT / yT: LDS, [*][8] row-tile-transposed.
template <int J, int ACT>
__device__ __forceinline__ void tile_linear(const float* __restrict__ WT, const float* __restrict__ b, const float* xT, float* yT,
                                            float* scratch, int K) {
  constexpr int NP = 256 / J;
  const int t = threadIdx.x, j = t % J, part = t / J;
  const int kc = (K + NP - 1) / NP, k0 = part * kc, k1 = min(K, k0 + kc);
  float acc[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) acc[r] = 0.f;
  // 32 weight loads in flight per thread: the k loop is a chain of L2 round trips (~0.7 us each) -- with 4 per trip the three layers of the
  // score kernel took 72 trips = 51 us for 64 rows; same summation order, so the results do not change.
  constexpr int CH = 32;
  for (int kb = k0; kb < k1; kb += CH) {
    float w[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) w[c] = (kb + c < k1) ? WT[(size_t)(kb + c) * J + j] : 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int k = min(kb + c, k1 - 1);           // (past the end: weight 0 on a valid LDS row)
      const float4 xa = *reinterpret_cast<const float4*>(xT + k * RT), xb = *reinterpret_cast<const float4*>(xT + k * RT + 4);
      acc[0] += w[c] * xa.x; acc[1] += w[c] * xa.y; acc[2] += w[c] * xa.z; acc[3] += w[c] * xa.w;
      acc[4] += w[c] * xb.x; acc[5] += w[c] * xb.y; acc[6] += w[c] * xb.z; acc[7] += w[c] * xb.w;
    }
  }
  if (NP > 1) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RT; ++r) scratch[(part * J + j) * RT + r] = acc[r];
    __syncthreads();
    if (part == 0) {
      for (int q = 1; q < NP; ++q)
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r] += scratch[(q * J + j) * RT + r];
    }
  }
  if (part == 0) {
    const float bj = b[j];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      const float v = acc[r] + bj;
      yT[j * RT + r] = (ACT == 1) ? (v > 0.f || v != v ? v : 0.f) : v;      // nn.ReLU: NaN stays NaN (fmaxf would turn it into 0)
    }
  }
  __syncthreads();
}

__global__ void transpose_small_kernel(const float* __restrict__ W, float* __restrict__ WT, int J, int K) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;       // W [J][K] -> WT [K][J]
  if (i < J * K) { const int j = i / K, k = i % K; WT[(size_t)k * J + j] = W[i]; }
}

// Label range errors (nn.Embedding / F.one_hot raise in the reference for an index outside [0, label_dim), semireward.py:57, srflexmatch.py:
// 180-181).  A kernel cannot raise: it records the event in this word and stays memory safe (embedding row 0 is read, the scatter is skipped);
// the host fetches the word with srhip_label_error() at its next synchronisation point and raises there.
// bit 0: embedding lookup out of range, bit 1: embedding-gradient scatter out of range, bit 2: generator output not representable (NaN / >= 2^63),
// bit 3: one_hot class out of range in the SR target
__device__ int srhip_label_err;

// Kernel 1: a tile of 8 feature rows AND the matching 8 label rows of one group.
// feature_fc (F->128) through tile_linear, then one wave per 2 rows for the two LayerNorms and the attention logit.
// grid = (ceil(B/8), G), block 256, dyn LDS = (F*8 + 128*8 + 256*8) floats.
// feat_gs: elements between the first feature rows of consecutive groups (B * F when the groups are stacked densely; the weak-row block of every
// pass inside the step's [passes, batch, F] feature buffer otherwise -- read in place instead of from a gathered copy)
__device__ __forceinline__ void rew_embed_body(const float* __restrict__ P, const float* __restrict__ PT,
                                               const float* __restrict__ feats, const long long* __restrict__ labels,
                                               float* __restrict__ ws, int G, int B, int F, int L, int save, long long feat_gs, float* sm) {
  const RewOff o(F, L);
  const RewTOff ot(F);
  const RewWs w(G, B);
  float* xT = sm;                  // [F][8]
  float* hT = xT + F * RT;         // [128][8]
  float* scratch = hT + E * RT;    // [256*8]
  const int grp = blockIdx.y, r0 = blockIdx.x * RT;
  for (int e = threadIdx.x; e < F * RT; e += 256) {
    const int r = e / F, k = e % F;
    xT[k * RT + r] = (r0 + r < B) ? feats[(size_t)grp * feat_gs + (size_t)(r0 + r) * F + k] : 0.f;
  }
  __syncthreads();
  tile_linear<E, 0>(PT + ot.WfT, P + o.bf, xT, hT, scratch, F);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int q = wave; q < 2 * RT; q += 4) {           // q < 8: feature row q ; q >= 8: label row q - 8
    const int r = q & (RT - 1), row = r0 + r;
    if (row >= B) continue;
    float v0, v1;
    const float *gam, *bet;
    if (q < RT) {
      v0 = hT[lane * RT + r]; v1 = hT[(lane + 64) * RT + r];
      gam = P + o.gf; bet = P + o.bef;
    } else {
      long long y = labels[(size_t)grp * B + row];
      if (y < 0 || y >= L) { if (lane == 0) atomicOr(&srhip_label_err, 1); y = 0; }
      const float* e = P + o.Emb + (size_t)y * E;
      v0 = e[lane]; v1 = e[lane + 64];
      gam = P + o.gl; bet = P + o.bl;
    }
    float z0, z1, xh0, xh1, rs;
    wave_ln128(v0, v1, gam, bet, lane, z0, z1, xh0, xh1, rs);
    const size_t zr = (size_t)grp * 2 * B + (q < RT ? row : B + row);
    ws[w.z + zr * E + lane] = z0;
    ws[w.z + zr * E + lane + 64] = z1;
    const float s = wave_sum(z0 * P[o.wa + lane] + z1 * P[o.wa + lane + 64]) + P[o.ba];
    if (lane == 0) ws[w.slog + zr] = s;
    if (save) {
      ws[w.xhat + zr * E + lane] = xh0;
      ws[w.xhat + zr * E + lane + 64] = xh1;
      if (lane == 0) ws[w.rstd + zr] = rs;
    }
  }
}

__global__ __launch_bounds__(256) void rew_embed_kernel(const float* __restrict__ P, const float* __restrict__ PT,
                                                       const float* __restrict__ feats, const long long* __restrict__ labels,
                                                       float* __restrict__ ws, int G, int B, int F, int L, int save, long long feat_gs) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  rew_embed_body(P, PT, feats, labels, ws, G, B, F, L, save, feat_gs, sm);
}

// Kernel 2: softmax over the group's 2B logits, context vector, then the MLP/FFN head for a tile of 8 rows.
// grid = (ceil(B/8), G), block 256.  Every workgroup re-derives the group context (2B x 128 reads) instead of
// paying a third launch.
constexpr int SCORE_LDS_FLOATS = E + 8 + E * RT + 256 * RT + E * RT + 64 * RT + 256 * RT;
__device__ __forceinline__ void rew_score_body(const float* __restrict__ P, const float* __restrict__ PT,
                                               float* __restrict__ ws, float* __restrict__ reward,
                                               int G, int B, int F, int L, int save, float* lds, float* __restrict__ max_reward = nullptr) {
  const RewOff o(F, L);
  const RewTOff ot(F);
  const RewWs w(G, B);
  float* uT = lds;                 // 16-byte aligned tiles first
  float* m1T = uT + E * RT;
  float* m2T = m1T + 256 * RT;
  float* f1T = m2T + E * RT;
  float* scratch = f1T + 64 * RT;
  float* ctx = scratch + 256 * RT;
  float* red = ctx + E;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = blockIdx.y, r0 = blockIdx.x * RT;
  const float* sl = ws + w.slog + (size_t)grp * 2 * B;
  const float* z = ws + w.z + (size_t)grp * 2 * B * E;
  // --- softmax statistics over 2B rows (fixed reduction order)
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < 2 * B; i += 256) mx = fmaxf(mx, sl[i]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float se = 0.f;
  for (int i = threadIdx.x; i < 2 * B; i += 256) se += expf(sl[i] - mx);
  se = wave_sum(se);
  if (lane == 0) red[4 + wave] = se;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  // --- ctx[j] = sum_i alpha_i z[i][j]; threads 0..127 own a column, 128..255 help on the odd rows
  {
    const int j = threadIdx.x & (E - 1), half = threadIdx.x >> 7;
    float a = 0.f;
#pragma unroll 8
    for (int i = half; i < 2 * B; i += 2) a += expf(sl[i] - mx) * inv * z[(size_t)i * E + j];
    if (half == 1) scratch[j] = a;
    __syncthreads();
    if (half == 0) ctx[j] = a + scratch[j];
    __syncthreads();
  }
  if (save && blockIdx.x == 0) {
    for (int i = threadIdx.x; i < 2 * B; i += 256) ws[w.alpha + (size_t)grp * 2 * B + i] = expf(sl[i] - mx) * inv;
    if (threadIdx.x < E) ws[w.ctx + (size_t)grp * E + threadIdx.x] = ctx[threadIdx.x];
  }
  // --- u = e + ctx for the tile's rows (label rows live at z[B + row])
  for (int e = threadIdx.x; e < E * RT; e += 256) {
    const int r = e / E, k = e % E, row = r0 + r;
    uT[k * RT + r] = row < B ? z[(size_t)(B + row) * E + k] + ctx[k] : 0.f;
  }
  __syncthreads();
  tile_linear<256, 1>(PT + ot.W1T, P + o.b1, uT, m1T, scratch, E);
  tile_linear<E, 0>(PT + ot.W2T, P + o.b2, m1T, m2T, scratch, 256);
  tile_linear<64, 1>(PT + ot.W3T, P + o.b3, m2T, f1T, scratch, E);
  for (int r = wave; r < RT; r += 4) {
    const int row = r0 + r;
    if (row >= B) continue;
    const float lg = wave_sum(f1T[lane * RT + r] * P[o.w4 + lane]) + P[o.b4];
    const float rr = 1.0f / (1.0f + expf(-lg));
    const size_t gr = (size_t)grp * B + row;
    if (lane == 0) reward[gr] = rr;
    if (save) {
      ws[w.u + gr * E + lane] = uT[lane * RT + r]; ws[w.u + gr * E + lane + 64] = uT[(lane + 64) * RT + r];
      for (int j = lane; j < 256; j += 64) ws[w.m1 + gr * 256 + j] = m1T[j * RT + r];
      ws[w.m2 + gr * E + lane] = m2T[lane * RT + r]; ws[w.m2 + gr * E + lane + 64] = m2T[(lane + 64) * RT + r];
      ws[w.f1 + gr * 64 + lane] = f1T[lane * RT + r];
      if (lane == 0) ws[w.r + gr] = rr;
    }
  }
  // max_reward = max(max_reward, mean(reward)) of a single-tile, single-group scoring call (srflexmatch.py:166-170: `reward.mean()` and the
  // running maximum the stage-2 update compares against) in the same launch: rows summed in index order by one thread
  if (max_reward && G == 1 && gridDim.x == 1) {
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x == 0) {
      float s = 0.f;
      for (int i = 0; i < B; ++i) s += reward[i];
      *max_reward = fmaxf(*max_reward, s / (float)B);
    }
  }
}

__global__ __launch_bounds__(256) void rew_score_kernel(const float* __restrict__ P, const float* __restrict__ PT,
                                                       float* __restrict__ ws, float* __restrict__ reward,
                                                       int G, int B, int F, int L, int save) {
  __shared__ __attribute__((aligned(16))) float lds[SCORE_LDS_FLOATS];
  rew_score_body(P, PT, ws, reward, G, B, F, L, save, lds);
}

// Both kernels as ONE launch when a group is a single row tile (B <= 8: the reference batch, uratio 1 with 8 unlabeled images per pass):
// the only cross-workgroup dependency -- the batch softmax over the group's 2B attention logits (semireward.py:60-62) -- is then inside the
// workgroup, and the rows' z / logit vectors travel through the workspace between two workgroup barriers instead of between two launches.
__global__ __launch_bounds__(256) void rew_fused_kernel(const float* __restrict__ P, const float* __restrict__ PT,
                                                       const float* __restrict__ feats, const long long* __restrict__ labels,
                                                       float* __restrict__ ws, float* __restrict__ reward, int G, int B, int F, int L,
                                                       int save, long long feat_gs, float* __restrict__ max_reward) {
  extern __shared__ __attribute__((aligned(16))) float sm[];      // max(embed tiles, score tiles)
  rew_embed_body(P, PT, feats, labels, ws, G, B, F, L, save, feat_gs, sm);
  __threadfence_block();
  __syncthreads();                 // the workspace rows of this group (z, logits) are visible to the whole workgroup
  rew_score_body(P, PT, ws, reward, G, B, F, L, save, sm, max_reward);
}

// ------------------------------------------------------------------------------------------------
// Backward (single group, G = 1).  Step A, one wave per row: dL/dreward -> dlogit, df1, dm2, dm1, du.
//   dL/dr_b = (2/B) * [(r_b - 1) + (r_b - t_b)]          (MSE(r,1) + MSE(r,t), both into the rewarder)
__global__ __launch_bounds__(256) void rew_bwd_rows_kernel(const float* __restrict__ P, float* __restrict__ ws,
                                                          const float* __restrict__ target, float* __restrict__ losses,
                                                          int B, int F, int L) {
  const RewOff o(F, L);
  const RewWs w(1, B);
  __shared__ float buf[4][64 + E + 256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, row = blockIdx.x * 4 + wave;
  if (row >= B) return;
  float* df1 = buf[wave];
  float* dm2 = df1 + 64;
  float* dm1 = dm2 + E;
  const float r = ws[w.r + row], t = target[row];
  const float dr = (2.0f / (float)B) * ((r - 1.0f) + (r - t));
  const float dlg = dr * r * (1.0f - r);
  if (lane == 0) ws[w.dlogit + row] = dlg;
  // df1 = dlg * w4 * relu'(f1)
  {
    const float f = ws[w.f1 + (size_t)row * 64 + lane];
    const float d = f > 0.f ? dlg * P[o.w4 + lane] : 0.f;
    df1[lane] = d;
    ws[w.df1 + (size_t)row * 64 + lane] = d;
  }
  wave_lds_sync();
  // dm2[k] = sum_j df1[j] W3[j,k]
  for (int k = lane; k < E; k += 64) {
    float a = 0.f;
    for (int j = 0; j < 64; ++j) a += df1[j] * P[o.W3 + (size_t)j * E + k];
    dm2[k] = a;
    ws[w.dm2 + (size_t)row * E + k] = a;
  }
  wave_lds_sync();
  // dm1[k] = relu'(m1[k]) * sum_j dm2[j] W2[j,k]
  for (int k = lane; k < 256; k += 64) {
    float a = 0.f;
    for (int j = 0; j < E; ++j) a += dm2[j] * P[o.W2 + (size_t)j * 256 + k];
    a = ws[w.m1 + (size_t)row * 256 + k] > 0.f ? a : 0.f;
    dm1[k] = a;
    ws[w.dm1 + (size_t)row * 256 + k] = a;
  }
  wave_lds_sync();
  // du[k] = sum_j dm1[j] W1[j,k]
  for (int k = lane; k < E; k += 64) {
    float a = 0.f;
    for (int j = 0; j < 256; ++j) a += dm1[j] * P[o.W1 + (size_t)j * E + k];
    ws[w.du + (size_t)row * E + k] = a;
  }
  (void)losses;
}

// Step B, ONE workgroup of 128 threads (thread j owns column j of the 128-wide vectors):
//   dc = sum_b du_b ; ds_i = alpha_i * ((z_i - c) . dc) ; dz_i = alpha_i * dc + ds_i * wa (+ du for label rows)
//   d(wa) = sum_i ds_i z_i ;  d(ba) = 0 exactly (cancels in the softmax)
//   then LayerNorm backward of every row -> d(pre) written IN PLACE over dz, d(gamma/beta) for both norms,
//   and the MSE losses.
__global__ __launch_bounds__(128) void rew_bwd_ctx_kernel(const float* __restrict__ P, float* __restrict__ ws, float* __restrict__ G_,
                                                         const float* __restrict__ target, float* __restrict__ losses,
                                                         int B, int F, int L) {
  const RewOff o(F, L);
  const RewWs w(1, B);
  __shared__ float sh[2];
  const int j = threadIdx.x, lane = j & 63, wv = j >> 6;
  auto bsum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    return sh[0] + sh[1];
  };
  float dc = 0.f;
  for (int b = 0; b < B; ++b) dc += ws[w.du + (size_t)b * E + j];
  const float c = ws[w.ctx + j], wa = P[o.wa + j];
  float dwa = 0.f, dgf = 0.f, dbf = 0.f, dgl = 0.f, dbl = 0.f;
  for (int i = 0; i < 2 * B; ++i) {
    const float zi = ws[w.z + (size_t)i * E + j], al = ws[w.alpha + i];
    const float ds = al * bsum((zi - c) * dc);
    dwa += ds * zi;
    float dz = al * dc + ds * wa;
    if (i >= B) dz += ws[w.du + (size_t)(i - B) * E + j];
    // LayerNorm backward of row i (gamma of the matching norm)
    const float gam = i < B ? P[o.gf + j] : P[o.gl + j];
    const float xh = ws[w.xhat + (size_t)i * E + j], rs = ws[w.rstd + i];
    const float gy = dz * gam;
    const float c1 = bsum(gy) * (1.0f / E), c2 = bsum(gy * xh) * (1.0f / E);
    ws[w.dz + (size_t)i * E + j] = rs * (gy - c1 - xh * c2);
    if (i < B) { dgf += dz * xh; dbf += dz; } else { dgl += dz * xh; dbl += dz; }
  }
  G_[o.wa + j] = dwa;
  if (j == 0) G_[o.ba] = 0.f;
  G_[o.gf + j] = dgf; G_[o.bef + j] = dbf; G_[o.gl + j] = dgl; G_[o.bl + j] = dbl;
  if (losses) {
    float lg = 0.f, lr = 0.f;
    for (int b = j; b < B; b += 128) {
      const float r = ws[w.r + b], t = target[b];
      lg += (r - 1.0f) * (r - 1.0f);
      lr += (r - t) * (r - t);
    }
    lg = bsum(lg); lr = bsum(lr);
    if (j == 0) { losses[0] = lg / (float)B; losses[1] = lr / (float)B; }
  }
}

// Step C: dW[j,k] = sum_b dY[b,j] X[b,k] (+ db[j] = sum_b dY[b,j]).  grid = J, block = 256 over k.
__global__ void small_dw_kernel(const float* __restrict__ dY, int ldy, const float* __restrict__ X, int ldx, float* __restrict__ dW,
                                float* __restrict__ db, int B, int K) {
  const int j = blockIdx.x;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += dY[(size_t)b * ldy + j] * X[(size_t)b * ldx + k];
    dW[(size_t)j * K + k] = a;
  }
  if (db && threadIdx.x == 0) {
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += dY[(size_t)b * ldy + j];
    db[j] = a;
  }
}

// Step D: label-embedding gradient, dEmb[y_b] += d(e_pre)_b  (duplicates -> atomics).  grid = B, block = 128
__global__ void rew_emb_scatter_kernel(const float* __restrict__ dpre, const long long* __restrict__ labels, float* __restrict__ dEmb, int L) {
  const int b = blockIdx.x;
  const long long y = labels[b];
  if (y < 0 || y >= L) { if (threadIdx.x == 0) atomicOr(&srhip_label_err, 2); return; }
  atomicAdd(dEmb + (size_t)y * E + threadIdx.x, dpre[(size_t)b * E + threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// Generator: relu(L4(relu(L3(relu(L2(relu(L1 x))))))) -> (float, int64 label).  Tile of 8 rows per workgroup.
// dyn LDS = F*8 floats + the fixed buffers below.
__global__ __launch_bounds__(256) void generator_kernel(const float* __restrict__ P, const float* __restrict__ PT,
                                                       const float* __restrict__ x, float* __restrict__ out,
                                                       long long* __restrict__ label, int B, int F) {
  const GenTOff ot(F);
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  float* xT = gsm;                         // [F][8]
  float* h1 = xT + F * RT;                 // [256][8]
  float* h2 = h1 + 256 * RT;               // [128][8]
  float* h3 = h2 + 128 * RT;               // [64][8]
  float* scratch = h3 + 64 * RT;           // [256*8]
  const int r0 = blockIdx.x * RT;
  for (int e = threadIdx.x; e < F * RT; e += 256) {
    const int r = e / F, k = e % F;
    xT[k * RT + r] = (r0 + r < B) ? x[(size_t)(r0 + r) * F + k] : 0.f;
  }
  __syncthreads();
  int off = 0;
  off += 256 * F; const float* b1 = P + off; off += 256;
  off += 128 * 256; const float* b2 = P + off; off += 128;
  off += 64 * 128; const float* b3 = P + off; off += 64;
  const float* W4 = P + off; off += 64; const float* b4 = P + off;
  tile_linear<256, 1>(PT + ot.W1T, b1, xT, h1, scratch, F);
  tile_linear<128, 1>(PT + ot.W2T, b2, h1, h2, scratch, 256);
  tile_linear<64, 1>(PT + ot.W3T, b3, h2, h3, scratch, 128);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int r = wave; r < RT; r += 4) {
    const int row = r0 + r;
    if (row >= B) continue;
    const float pre = wave_sum(h3[lane * RT + r] * W4[lane]) + b4[0];
    const float v = pre != pre ? pre : fmaxf(pre, 0.f);            // torch.relu propagates NaN (fmaxf would return 0)
    if (lane == 0) {
      out[row] = v;
      // .long(): truncation toward zero; NaN / out-of-range conversions are undefined in C and garbage in torch -> -1 + error flag (the
      // reference then fails in F.one_hot / nn.Embedding)
      const bool ok = v < 9.2e18f;                                 // false for NaN and +inf too; v >= 0 after the ReLU
      if (!ok) atomicOr(&srhip_label_err, 4);
      label[row] = ok ? (long long)v : -1;
    }
  }
}

// target_b = 1.0 if gen_b == ref_b else 0.5   ( (cos(one_hot, one_hot) + 1) / 2, srflexmatch.py:180-182 )
__global__ void sr_target_kernel(const long long* __restrict__ gen, const long long* __restrict__ ref, float* __restrict__ target, int B,
                                 int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const long long a = gen[i], b = ref[i];
  if (C > 0 && (a < 0 || a >= C || b < 0 || b >= C)) atomicOr(&srhip_label_err, 8);      // F.one_hot(., num_classes) raises
  target[i] = a == b ? 1.0f : 0.5f;
}

// torch.optim.Adam, flat fp32 block (betas 0.9/0.999, eps 1e-8, no weight decay)
__global__ void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int n,
                                 float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt, const float* __restrict__ dyn) {
  if (dyn) { bc1 = dyn[0]; bc2_sqrt = dyn[1]; }        // srhip_adam_flat_dyn: the bias corrections of this step from device memory (HIP graph replay)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  const float mi = b1 * m[i] + (1.0f - b1) * gi;
  const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
  m[i] = mi; v[i] = vi;
  p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
}

}  // namespace

extern "C" long srhip_rewarder_param_count(int F, int L) { return RewOff(F, L).total; }
extern "C" long srhip_rewarder_ws_floats(int G, int B) { return (long)RewWs(G, B).total; }
extern "C" long srhip_generator_param_count(int F) { return 256L * F + 256 + 128 * 256 + 128 + 64 * 128 + 64 + 64 + 1; }
extern "C" long srhip_rewarder_t_floats(int F) { return RewTOff(F).total; }
extern "C" long srhip_generator_t_floats(int F) { return GenTOff(F).total; }

static void launch_tr(const float* W, float* WT, int J, int K, hipStream_t s) {
  SR_LAUNCH(transpose_small_kernel, dim3(cdiv((long)J * K, 256)), dim3(256), 0, s, W, WT, J, K);
}

extern "C" int srhip_rewarder_prepare(const float* params, float* params_t, int F, int L, void* stream) {
  if (F <= 0 || F > 1024 || L <= 0) return SR_EINVAL;
  const RewOff o(F, L);
  const RewTOff ot(F);
  hipStream_t s = (hipStream_t)stream;
  launch_tr(params + o.Wf, params_t + ot.WfT, E, F, s);
  launch_tr(params + o.W1, params_t + ot.W1T, 256, E, s);
  launch_tr(params + o.W2, params_t + ot.W2T, E, 256, s);
  launch_tr(params + o.W3, params_t + ot.W3T, 64, E, s);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_generator_prepare(const float* params, float* params_t, int F, void* stream) {
  if (F <= 0 || F > 1024) return SR_EINVAL;
  const GenTOff ot(F);
  hipStream_t s = (hipStream_t)stream;
  const float* W1 = params;
  const float* W2 = W1 + 256 * F + 256;
  const float* W3 = W2 + 128 * 256 + 128;
  launch_tr(W1, params_t + ot.W1T, 256, F, s);
  launch_tr(W2, params_t + ot.W2T, 128, 256, s);
  launch_tr(W3, params_t + ot.W3T, 64, 128, s);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_rewarder_fwd(const float* params, const float* params_t, const float* feats, const long long* labels,
                                  float* reward, float* ws, int G, int B, int F, int L, int save_for_bwd, void* stream) {
  return srhip_rewarder_fwd_strided(params, params_t, feats, (long long)B * F, labels, reward, ws, nullptr, G, B, F, L, save_for_bwd, stream);
}
extern "C" int srhip_rewarder_fwd_strided(const float* params, const float* params_t, const float* feats, long long feat_group_stride,
                                          const long long* labels, float* reward, float* ws, float* max_reward_inout, int G, int B, int F,
                                          int L, int save_for_bwd, void* stream) {
  if (G <= 0 || B <= 0 || F <= 0 || F > 1024 || L <= 0 || (save_for_bwd && G != 1) || !params_t) return SR_EINVAL;
  if (feat_group_stride < (long long)B * F) return SR_EINVAL;
  if (max_reward_inout && (G != 1 || B > RT)) return SR_EINVAL;            // the running maximum rides in the one-launch form only
  hipStream_t s = (hipStream_t)stream;
  const size_t sm1 = ((size_t)F * RT + E * RT + 256 * RT) * sizeof(float);
  static const bool two = SR_TUNE_ENV("SRHIP_REWARDER_TWO_LAUNCHES") != nullptr;
  if (B <= RT && (!two || max_reward_inout)) {
    const size_t smf = sm1 > SCORE_LDS_FLOATS * sizeof(float) ? sm1 : SCORE_LDS_FLOATS * sizeof(float);
    SR_LAUNCH(rew_fused_kernel, dim3(1, G), dim3(256), smf, s, params, params_t, feats, labels, ws, reward, G, B, F, L, save_for_bwd,
                       feat_group_stride, max_reward_inout);
    SR_CHECK_LAUNCH();
    return SR_OK;
  }
  SR_LAUNCH(rew_embed_kernel, dim3(cdiv(B, RT), G), dim3(256), sm1, s, params, params_t, feats, labels, ws, G, B, F, L, save_for_bwd,
                     feat_group_stride);
  SR_CHECK_LAUNCH();
  SR_LAUNCH(rew_score_kernel, dim3(cdiv(B, RT), G), dim3(256), 0, s, params, params_t, ws, reward, G, B, F, L, save_for_bwd);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

// grads: flat block, same layout as params, fully overwritten (== zero_grad + the two backward() calls).
extern "C" int srhip_rewarder_bwd(const float* params, const float* feats, const long long* labels, const float* target, float* ws,
                                  float* grads, float* losses, int B, int F, int L, void* stream) {
  if (B <= 0 || F <= 0 || F > 1024 || L <= 0) return SR_EINVAL;
  const RewOff o(F, L);
  const RewWs w(1, B);
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(grads + o.Emb, 0, (size_t)L * E * sizeof(float), s) != hipSuccess) return SR_ELAUNCH;
  SR_LAUNCH(rew_bwd_rows_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, params, ws, target, losses, B, F, L);
  SR_CHECK_LAUNCH();
  SR_LAUNCH(rew_bwd_ctx_kernel, dim3(1), dim3(128), 0, s, params, ws, grads, target, losses, B, F, L);
  SR_CHECK_LAUNCH();
  // weight gradients of the per-row head
  SR_LAUNCH(small_dw_kernel, dim3(1), dim3(64), 0, s, ws + w.dlogit, 1, ws + w.f1, 64, grads + o.w4, grads + o.b4, B, 64);
  SR_LAUNCH(small_dw_kernel, dim3(64), dim3(128), 0, s, ws + w.df1, 64, ws + w.m2, E, grads + o.W3, grads + o.b3, B, E);
  SR_LAUNCH(small_dw_kernel, dim3(E), dim3(256), 0, s, ws + w.dm2, E, ws + w.m1, 256, grads + o.W2, grads + o.b2, B, 256);
  SR_LAUNCH(small_dw_kernel, dim3(256), dim3(128), 0, s, ws + w.dm1, 256, ws + w.u, E, grads + o.W1, grads + o.b1, B, E);
  // feature_fc: d(pre) of the B feature rows sits in dz rows [0,B); label rows [B,2B) feed the embedding
  SR_LAUNCH(small_dw_kernel, dim3(E), dim3(256), 0, s, ws + w.dz, E, feats, F, grads + o.Wf, grads + o.bf, B, F);
  SR_LAUNCH(rew_emb_scatter_kernel, dim3(B), dim3(E), 0, s, ws + w.dz + (size_t)B * E, labels, grads + o.Emb, L);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_generator_fwd(const float* params, const float* params_t, const float* x, float* out, long long* label, int B,
                                   int F, void* stream) {
  if (B <= 0 || F <= 0 || F > 1024 || !params_t) return SR_EINVAL;
  const size_t sm = ((size_t)F * RT + (256 + 128 + 64 + 256) * RT) * sizeof(float);
  SR_LAUNCH(generator_kernel, dim3(cdiv(B, RT)), dim3(256), sm, (hipStream_t)stream, params, params_t, x, out, label, B, F);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_label_error(int* bits_out, int reset, void* stream) {
  if (!bits_out) return SR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemcpyFromSymbolAsync(bits_out, HIP_SYMBOL(srhip_label_err), sizeof(int), 0, hipMemcpyDeviceToHost, s) != hipSuccess) return SR_ELAUNCH;
  if (hipStreamSynchronize(s) != hipSuccess) return SR_ELAUNCH;
  if (reset && *bits_out) {
    const int z = 0;
    if (hipMemcpyToSymbolAsync(HIP_SYMBOL(srhip_label_err), &z, sizeof(int), 0, hipMemcpyHostToDevice, s) != hipSuccess) return SR_ELAUNCH;
    if (hipStreamSynchronize(s) != hipSuccess) return SR_ELAUNCH;
  }
  return SR_OK;
}

extern "C" int srhip_sr_target(const long long* gen, const long long* ref, float* target, int B, int num_classes, void* stream) {
  if (B <= 0) return SR_EINVAL;
  SR_LAUNCH(sr_target_kernel, dim3(cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, gen, ref, target, B, num_classes);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_adam_flat(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                               int step, void* stream) {
  if (n <= 0 || step <= 0) return SR_EINVAL;
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
  SR_LAUNCH(adam_flat_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (int)n, lr, beta1, beta2, eps, bc1, bc2s,
                     (const float*)nullptr);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
extern "C" int srhip_adam_flat_dyn(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                                   const float* dyn, void* stream) {
  if (n <= 0 || !dyn) return SR_EINVAL;
  SR_LAUNCH(adam_flat_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (int)n, lr, beta1, beta2, eps, 1.f, 1.f, dyn);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
