// The pseudo-label "score filter": softmax / argmax / confidence thresholds / FlexMatch state /
// reward-mean mask / masked cross-entropy -- everything between the backbone logits and the loss.
//
// Reference call sites:
//   compute_prob (softmax)            semilearn/core/algorithmbase.py:332-333
//   PseudoLabelingHook (argmax)       semilearn/algorithms/hooks/pseudo_label.py:40
//   FixedThresholdingHook.masking     semilearn/algorithms/hooks/masking.py:42-57
//   FlexMatchThresholdingHook         semilearn/algorithms/srflexmatch/utils.py:24-63
//   mask2 = reward >= reward.mean()   semilearn/algorithms/srflexmatch/srflexmatch.py:100-101
//   ce_loss / consistency_loss        semilearn/core/criterions/cross_entropy.py:11-31, consistency.py:13-45
//
// These are latency-bound integer/compare kernels on [Bu, C] with Bu = 8..4096: one wave per row
// with shuffle reductions, all state on device, no host round trip.  The reference rebuilds the
// class histogram on the HOST from a 50 000-entry tolist() at every masking call; here the
// histogram hist[C+1] (last bin = the "-1 / unused" bucket) is maintained incrementally by the
// scatter itself (old label -> new label), which is bit-identical to a recount.
// Bit-exactness: the threshold p_cutoff * (acc / (2 - acc)) is evaluated op by op
// in fp32 with contraction disabled; classwise_acc = (float)((double)cnt / (double)max).
#include <stdlib.h>

#include "common.h"
#include "srhip.h"

#pragma clang fp contract(off)

namespace {

// wave-level (value, index) max with first-index tie break (torch.max / argmax semantics)
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(i, o, 64);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
}

// One wave per row.  in_is_probs = 0: logits -> softmax (fp32) ; 1: rows are already probabilities.
__global__ __launch_bounds__(256) void row_max_kernel(const float* __restrict__ in, int in_is_probs, float* __restrict__ probs_out,
                                                     float* __restrict__ max_probs, long long* __restrict__ max_idx, int B, int C,
                                                     int rows_per_group, long long group_stride) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B) return;
  // rows_per_group > 0: input row r lives at in + (r / rows_per_group) * group_stride + (r % rows_per_group) * C (a column block of a
  // [groups, rows, C] buffer: the weak rows of every pass, read in place); outputs are dense
  const float* r = rows_per_group > 0 ? in + (size_t)(row / rows_per_group) * group_stride + (size_t)(row % rows_per_group) * C
                                      : in + (size_t)row * C;
  float inv = 1.0f, mx = 0.f;
  if (!in_is_probs) {
    mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, r[c]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(r[c] - mx);
    inv = 1.0f / wave_sum(s);
  }
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    const float p = in_is_probs ? r[c] : expf(r[c] - mx) * inv;
    if (probs_out) probs_out[(size_t)row * C + c] = p;
    if (p > bv) { bv = p; bi = c; }
  }
  wave_argmax(bv, bi);
  if (lane == 0) { max_probs[row] = bv; max_idx[row] = bi; }
}

// Index errors: an idx_ulb entry outside [0, ulb_dest_len) raises IndexError in the reference (utils.py:59 index_put); here the entry is
// skipped (no out-of-bounds write) and bit 0 of this word is set; the host reads it with srhip_index_error() and raises.
__device__ int srhip_index_err;

// FlexMatch masking + state update.  ONE workgroup (the step is sequential by definition).
//   mask[i]   = max_p[i] >= p_cutoff * (acc[idx] / (2 - acc[idx]))          (utils.py:53, BEFORE the update)
//   select[i] = max_p[i] >= p_cutoff ; selected_label[idx_ulb[i]] = max_idx[i]   (utils.py:56-60)
//   hist bookkeeping, then classwise_acc update                                (utils.py:24-35)
// n_pass > 1: the passes of one SemiReward step (srflexmatch.py:75-104 calls masking once per data_generator pass, on the SAME idx_ulb) in
// ONE launch, in order -- pass p reads the state pass p-1 left.
//
// The whole state a launch touches lives in LDS between the passes: hist[C+1], classwise_acc[C] and the B entries selected_label[idx_ulb[i]]
// (the passes share idx_ulb).  Global memory is read once at the start (two dependent round trips: idx_ulb -> selected_label) and written
// once at the end; a pass is three workgroup barriers.  (The first version kept the state in global memory behind device-scope atomics
// and two __threadfence() per pass: 9 us per pass, 81 us for the 9 passes of a step that move 7 KB.)
// Duplicate indices inside one batch (the sampler's concatenated permutations can meet): all copies share one LDS slot (rep = first copy) and
// of the copies selected in a pass the LAST one writes, which is what the reference's CPU index_put does; hist then stays equal to a recount.
__global__ __launch_bounds__(256) void flexmatch_mask_lds_kernel(const float* __restrict__ max_probs, const long long* __restrict__ max_idx,
                                                                const long long* __restrict__ idx_ulb, float p_cutoff,
                                                                long long* __restrict__ selected_label, int* __restrict__ hist,
                                                                float* __restrict__ classwise_acc, float* __restrict__ mask,
                                                                int B, int C, int ulb_dest_len, int thresh_warmup, int n_pass) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fm_raw[];
  int* s_hist = reinterpret_cast<int*>(fm_raw);              // [C + 1]
  float* s_acc = reinterpret_cast<float*>(s_hist + C + 1);   // [C]
  int* s_j = reinterpret_cast<int*>(s_acc + C);              // [B] idx_ulb (or -1: out of range)
  int* s_rep = s_j + B;                                      // [B] first i' with the same index
  int* s_next = s_rep + B;                                   // [B] next i' > i with the same index, or -1
  int* s_sel = s_next + B;                                   // [B] current selected_label of the index (valid at rep)
  int* s_sel0 = s_sel + B;                                   // [B] its value at launch start
  int* s_flag = s_sel0 + B;                                  // [B] select of the running pass
  __shared__ int smax[2];
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int c = tid; c <= C; c += nt) s_hist[c] = hist[c];
  for (int c = tid; c < C; c += nt) s_acc[c] = classwise_acc[c];
  for (int i = tid; i < B; i += nt) {
    long long j = idx_ulb ? idx_ulb[i] : -1;
    if (j < 0 || j >= ulb_dest_len) { atomicOr(&srhip_index_err, 1); j = -1; }
    s_j[i] = (int)j;
    const int v = j >= 0 ? (int)selected_label[j] : -1;
    s_sel[i] = v; s_sel0[i] = v;
  }
  __syncthreads();
  for (int i = tid; i < B; i += nt) {
    const int j = s_j[i];
    int rep = i, nx = -1;
    if (j >= 0) {
      for (int k = 0; k < i; ++k) if (s_j[k] == j) { rep = k; break; }
      for (int k = i + 1; k < B; ++k) if (s_j[k] == j) { nx = k; break; }
    }
    s_rep[i] = rep; s_next[i] = nx;
  }
  __syncthreads();
  for (int p = 0; p < n_pass; ++p) {
    const float* mpp = max_probs + (size_t)p * B;
    const long long* mip = max_idx + (size_t)p * B;
    float* mk = mask + (size_t)p * B;
    for (int i = tid; i < B; i += nt) {
      const float mp = mpp[i];
      const int cls = (int)mip[i];
      const float acc = s_acc[cls];
      const float den = 2.0f - acc;
      const float rat = acc / den;
      const float thr = p_cutoff * rat;
      mk[i] = mp >= thr ? 1.0f : 0.0f;
      s_flag[i] = (mp >= p_cutoff && s_j[i] >= 0) ? 1 : 0;
    }
    if (tid < 2) smax[tid] = 0;
    __syncthreads();
    for (int i = tid; i < B; i += nt) {
      if (!s_flag[i]) continue;
      bool last = true;
      for (int k = s_next[i]; k >= 0; k = s_next[k]) if (s_flag[k]) { last = false; break; }
      if (!last) continue;
      const int cls = (int)mip[i], r = s_rep[i], old = s_sel[r];
      if (old != cls) {
        s_sel[r] = cls;
        atomicSub(&s_hist[old < 0 || old >= C ? C : old], 1);
        atomicAdd(&s_hist[cls], 1);
      }
    }
    __syncthreads();
    int m = 0;
    for (int c = tid; c < C; c += nt) m = max(m, s_hist[c]);
    if (m > 0) atomicMax(&smax[0], m);
    __syncthreads();
    const int max_cls = smax[0], max_all = max(max_cls, s_hist[C]);
    if (max_all < ulb_dest_len) {
      const int den = thresh_warmup ? max_all : max_cls;
      // den == 0 (thresh_warmup False, no entry of the table holds a class in [0, C)) cannot coexist with max_all < ulb_dest_len for a table of
      // -1 / class entries -- the reference's update() is skipped there too (srflexmatch/utils.py:27) -- but a table loaded with foreign
      // values could get here: classwise_acc then stays as it is instead of becoming 0 / 0
      if (den != 0)
        for (int c = tid; c < C; c += nt) s_acc[c] = (float)((double)s_hist[c] / (double)den);
    }
    __syncthreads();                  // the next pass reads s_acc / s_sel / s_hist; smax is reset after this point
  }
  for (int c = tid; c <= C; c += nt) hist[c] = s_hist[c];
  for (int c = tid; c < C; c += nt) classwise_acc[c] = s_acc[c];
  for (int i = tid; i < B; i += nt)
    if (s_j[i] >= 0 && s_rep[i] == i && s_sel[i] != s_sel0[i]) selected_label[s_j[i]] = (long long)s_sel[i];
}

// General path (state too large for LDS: C > 8192 or B > 4096): state in global memory behind device-scope atomics, two fences per pass.
__global__ __launch_bounds__(256) void flexmatch_mask_kernel(const float* __restrict__ max_probs, const long long* __restrict__ max_idx,
                                                            const long long* __restrict__ idx_ulb, float p_cutoff,
                                                            long long* __restrict__ selected_label, int* __restrict__ hist,
                                                            float* __restrict__ classwise_acc, float* __restrict__ mask,
                                                            int B, int C, int ulb_dest_len, int thresh_warmup, int n_pass) {
  __shared__ int smax[2];
  for (int p = 0; p < n_pass; ++p) {
    const float* mpp = max_probs + (size_t)p * B;
    const long long* mip = max_idx + (size_t)p * B;
    float* mk = mask + (size_t)p * B;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
      const float mp = mpp[i];
      const int cls = (int)mip[i];
      const float acc = __hip_atomic_load(classwise_acc + cls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float den = 2.0f - acc;
      const float rat = acc / den;
      const float thr = p_cutoff * rat;
      mk[i] = mp >= thr ? 1.0f : 0.0f;
      if (mp >= p_cutoff) {
        const long long j = idx_ulb[i];
        if (j < 0 || j >= ulb_dest_len) { atomicOr(&srhip_index_err, 1); continue; }
        // exchange, not load + store: two rows with the same index both see the true previous label, so hist stays a recount
        const long long old = (long long)atomicExch(reinterpret_cast<unsigned long long*>(selected_label + j), (unsigned long long)(long long)cls);
        if (old != cls) {
          atomicSub(hist + (old < 0 || old >= C ? C : (int)old), 1);
          atomicAdd(hist + cls, 1);
        }
      }
    }
    if (threadIdx.x < 2) smax[threadIdx.x] = 0;
    __threadfence();
    __syncthreads();
    int m = 0;
    for (int c = threadIdx.x; c < C; c += blockDim.x) m = max(m, __hip_atomic_load(hist + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    atomicMax(&smax[0], m);
    __syncthreads();
    if (threadIdx.x == 0) smax[1] = max(smax[0], __hip_atomic_load(hist + C, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    __syncthreads();
    const int max_all = smax[1];
    if (max_all < ulb_dest_len) {
      const int den = thresh_warmup ? max_all : smax[0];
      if (den != 0)                                                                   // (see the LDS kernel)
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int cnt = __hip_atomic_load(hist + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(classwise_acc + c, (float)((double)cnt / (double)den), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __threadfence();
    __syncthreads();                  // the next pass reads classwise_acc / selected_label / hist; smax is reset after this point
  }
}

// Rebuild hist[C+1] from selected_label (after construction / checkpoint load).
__global__ void flexmatch_hist_kernel(const long long* __restrict__ selected_label, int* __restrict__ hist, int n, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long v = selected_label[i];
  atomicAdd(hist + (v < 0 || v >= C ? C : (int)v), 1);
}

__global__ void fixed_mask_kernel(const float* __restrict__ max_probs, float p_cutoff, float* __restrict__ mask, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) mask[i] = max_probs[i] >= p_cutoff ? 1.0f : 0.0f;
}

// mask2[g, i] = reward[g, i] >= mean_i(reward[g, :]);  one workgroup per independent group g.
// Mean: fp32, fixed order.  B <= 64: left-to-right (== torch CPU for the reference batch of 8);
// larger B: per-lane strided partials then a fixed shuffle tree (deterministic).
__global__ __launch_bounds__(64) void reward_mask2_kernel(const float* __restrict__ reward, float* __restrict__ mask2,
                                                         float* __restrict__ mean_out, const float* __restrict__ mean_in, int B) {
  const float* r = reward + (size_t)blockIdx.x * B;
  const int lane = threadIdx.x;
  float s;
  if (mean_in) {                      // externally supplied threshold (data-parallel global mean extension)
    s = mean_in[blockIdx.x] * (float)B;
  } else if (B <= 64) {
    s = 0.f;
    for (int i = 0; i < B; ++i) s += r[i];
  } else {
    float p = 0.f;
    for (int i = lane; i < B; i += 64) p += r[i];
    s = wave_sum(p);
  }
  const float mean = mean_in ? mean_in[blockIdx.x] : s / (float)B;
  for (int i = lane; i < B; i += 64) mask2[(size_t)blockIdx.x * B + i] = r[i] >= mean ? 1.0f : 0.0f;
  if (mean_out && lane == 0) mean_out[blockIdx.x] = mean;
}

// Masked cross-entropy forward + analytic backward.  ONE workgroup (B is the per-GPU batch).
//   loss = coef_out * mean_i( nll_i * mask_i * mask2_i )         (consistency.py:38-45: mean over ALL rows)
//   dlogits[i, c] = grad_scale * (softmax_ic - [c == y_i]) * mask_i * mask2_i / B
__global__ __launch_bounds__(256) void masked_ce_kernel(const float* __restrict__ logits, const long long* __restrict__ targets,
                                                       const float* __restrict__ mask, const float* __restrict__ mask2,
                                                       float grad_scale, float* __restrict__ loss_out, float* __restrict__ dlogits,
                                                       int B, int C) {
  __shared__ float rowloss[1024];
  __shared__ float part[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int row = wave; row < B; row += 4) {
    const float* r = logits + (size_t)row * C;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, r[c]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(r[c] - mx);
    s = wave_sum(s);
    const int y = (int)targets[row];
    float w = 1.0f;
    if (mask) w *= mask[row];
    if (mask2) w *= mask2[row];
    const float nll = (mx + logf(s)) - r[y];
    if (dlogits) {
      const float k = grad_scale * w / (float)B, inv = 1.0f / s;
      for (int c = lane; c < C; c += 64) dlogits[(size_t)row * C + c] = k * (expf(r[c] - mx) * inv - (c == y ? 1.0f : 0.0f));
    }
    if (lane == 0) {
      if (B <= 1024) rowloss[row] = nll * w; else acc += nll * w;
    }
  }
  __syncthreads();
  if (B <= 1024) {              // fixed left-to-right order -> run-to-run identical
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int i = 0; i < B; ++i) t += rowloss[i];
      loss_out[0] = t / (float)B;
    }
  } else {
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss_out[0] = (part[0] + part[1] + part[2] + part[3]) / (float)B;
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// FreeMatch self-adaptive thresholding (semilearn/algorithms/freematch/utils.py:24-66) and the fairness ("entropy") loss
// (semilearn/algorithms/srfreematch/srfreematch.py:16-44).  All arithmetic fp32, op by op in the reference's order
// (contraction is off in this file) so the EMA state tracks the reference to the last bits.

// column sums of the probabilities and the predicted-class histogram of the LOCAL batch (the data-parallel path all-reduces
// these sufficient statistics instead of all-gathering probs).  grid = ceil(C/256).
__global__ void freematch_stats_kernel(const float* __restrict__ probs, const long long* __restrict__ max_idx, float* __restrict__ colsum,
                                       float* __restrict__ hist, int B, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f, h = 0.f;
  for (int i = 0; i < B; ++i) {
    s += probs[(size_t)i * C + c];
    h += (max_idx[i] == c) ? 1.0f : 0.0f;
  }
  colsum[c] = s;
  hist[c] = h;
}

// ONE workgroup: time_p / p_model / label_hist EMA update from the (global) statistics, then the mask of the LOCAL rows.
__global__ __launch_bounds__(256) void freematch_update_kernel(const float* __restrict__ maxp_all, int n_all, const float* __restrict__ colsum,
                                                              const float* __restrict__ hist, const float* __restrict__ max_probs,
                                                              const long long* __restrict__ max_idx, float* __restrict__ time_p,
                                                              float* __restrict__ p_model, float* __restrict__ label_hist,
                                                              float* __restrict__ mask, int B, int C, float m, float one_minus_m,
                                                              int use_quantile, int clip_thresh) {
  __shared__ float srt[1024];
  __shared__ float sh[8];
  const int tid = threadIdx.x;
  // --- time_p
  if (use_quantile) {          // torch.quantile(x, 0.8), 'linear': sort, rank = 0.8f*(n-1), lerp exactly as ATen does
    for (int i = tid; i < n_all; i += 256) {
      const float v = maxp_all[i];
      int r = 0;
      for (int j = 0; j < n_all; ++j) { const float u = maxp_all[j]; r += (u < v || (u == v && j < i)) ? 1 : 0; }
      srt[r] = v;
    }
    __syncthreads();
    if (tid == 0) {
      const float rank = 0.8f * (float)(n_all - 1);
      const float lo = floorf(rank), w = rank - lo;
      const float a = srt[(int)lo], b = srt[(int)ceilf(rank)], d = b - a;
      sh[0] = (w < 0.5f) ? a + w * d : b - d * (1.0f - w);
    }
  } else if (tid == 0) {
    float s = 0.f;
    for (int i = 0; i < n_all; ++i) s += maxp_all[i];
    sh[0] = s / (float)n_all;
  }
  __syncthreads();
  if (tid == 0) {
    float tp = time_p[0] * m + one_minus_m * sh[0];
    if (clip_thresh) tp = fminf(fmaxf(tp, 0.0f), 0.95f);
    time_p[0] = tp;
    sh[1] = tp;
  }
  // --- p_model, label_hist
  float hs = 0.f;
  for (int c = tid; c < C; c += 256) hs += hist[c];
  hs = wave_sum(hs);
  if ((tid & 63) == 0) sh[2 + (tid >> 6)] = hs;
  __syncthreads();
  const float hsum = sh[2] + sh[3] + sh[4] + sh[5];
  float pmax = 0.f;
  for (int c = tid; c < C; c += 256) {
    const float pm = p_model[c] * m + one_minus_m * (colsum[c] / (float)n_all);
    p_model[c] = pm;
    label_hist[c] = label_hist[c] * m + one_minus_m * (hist[c] / hsum);
    pmax = fmaxf(pmax, pm);
  }
  pmax = wave_max(pmax);
  __syncthreads();
  if ((tid & 63) == 0) sh[2 + (tid >> 6)] = pmax;
  __syncthreads();
  pmax = fmaxf(fmaxf(sh[2], sh[3]), fmaxf(sh[4], sh[5]));
  const float tp = sh[1];
  __threadfence_block();
  for (int i = tid; i < B; i += 256) {
    const float mod = p_model[(int)max_idx[i]] / pmax;
    mask[i] = max_probs[i] >= tp * mod ? 1.0f : 0.0f;
  }
}

// Fairness loss + analytic gradient.  ONE workgroup.  ws: B*C floats (softmax of the strong logits).
//   S = {i : mask_i = 1};  mean_c = mean_{i in S} p_ic;  hist_c = share of S predicted c;  a_c = 1/hist_c (0 where hist_c = 0)
//   q = normalise(mean * a);  w = normalise(p_model / label_hist);  loss = sum_c w_c log(q_c + 1e-12)
//   dlogits_ij (+)= grad_scale * p_ij * (g_j - sum_k p_ik g_k) / |S|,   g_c = a_c * (w_c/(q_c+eps) - sum_k w_k q_k/(q_k+eps)) / Z
__global__ __launch_bounds__(256) void freematch_entropy_kernel(const float* __restrict__ logits, const float* __restrict__ mask,
                                                               const float* __restrict__ p_model, const float* __restrict__ label_hist,
                                                               float grad_scale, float* __restrict__ loss_out, float* __restrict__ dlogits,
                                                               float* __restrict__ ws, int B, int C, int accumulate) {
  __shared__ float g[2048];
  __shared__ int pred[1024];
  __shared__ float sh[8];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  auto bsum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
  };
  for (int row = wave; row < B; row += 4) {
    const float* r = logits + (size_t)row * C;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, r[c]);
    mx = wave_max(mx);
    float sm = 0.f;
    for (int c = lane; c < C; c += 64) sm += expf(r[c] - mx);
    const float inv = 1.0f / wave_sum(sm);
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
      const float p = expf(r[c] - mx) * inv;
      ws[(size_t)row * C + c] = p;
      if (p > bv) { bv = p; bi = c; }
    }
    wave_argmax(bv, bi);
    if (lane == 0) pred[row] = bi;
  }
  float ns = 0.f;
  for (int i = tid; i < B; i += 256) ns += mask[i] != 0.f ? 1.0f : 0.0f;
  const float nsel = bsum(ns);
  if (nsel == 0.f) {                          // reference: `if mask.sum() > 0 ... else ent_loss = 0.0` (srfreematch.py:216-219)
    if (tid == 0) loss_out[0] = 0.f;
    if (!accumulate)
      for (int e = tid; e < B * C; e += 256) dlogits[e] = 0.f;
    return;
  }
  float zu = 0.f, zw = 0.f;
  for (int c = tid; c < C; c += 256) {
    float mean = 0.f, h = 0.f;
    for (int i = 0; i < B; ++i)
      if (mask[i] != 0.f) { mean += ws[(size_t)i * C + c]; h += (pred[i] == c) ? 1.0f : 0.0f; }
    mean = mean / nsel; h = h / nsel;
    const float a = h > 0.f ? 1.0f / h : 0.0f;
    const float lh = label_hist[c];
    g[c] = mean * a;                                   // u_c (overwritten by g_c below)
    zu += mean * a;
    zw += p_model[c] * (lh > 0.f ? 1.0f / lh : 0.0f);
  }
  const float Z = bsum(zu), W = bsum(zw);
  float ls = 0.f, cr = 0.f;
  for (int c = tid; c < C; c += 256) {
    const float lh = label_hist[c];
    const float w = p_model[c] * (lh > 0.f ? 1.0f / lh : 0.0f) / W;
    const float q = g[c] / Z;
    ls += w * logf(q + 1e-12f);
    cr += w * q / (q + 1e-12f);
  }
  const float loss = bsum(ls), cross = bsum(cr);
  for (int c = tid; c < C; c += 256) {
    const float lh = label_hist[c];
    const float w = p_model[c] * (lh > 0.f ? 1.0f / lh : 0.0f) / W;
    const float u = g[c], q = u / Z;
    // a_c = u_c / mean_c is not kept: recompute it from the histogram share
    float h = 0.f;
    for (int i = 0; i < B; ++i) h += (mask[i] != 0.f && pred[i] == c) ? 1.0f : 0.0f;
    const float a = h > 0.f ? nsel / h : 0.0f;
    g[c] = a * (w / (q + 1e-12f) - cross) / Z;
  }
  __syncthreads();
  if (tid == 0) loss_out[0] = loss;
  for (int row = wave; row < B; row += 4) {
    float* d = dlogits + (size_t)row * C;
    if (mask[row] == 0.f) {
      if (!accumulate) for (int c = lane; c < C; c += 64) d[c] = 0.f;
      continue;
    }
    const float* p = ws + (size_t)row * C;
    float dot = 0.f;
    for (int c = lane; c < C; c += 64) dot += p[c] * g[c];
    dot = wave_sum(dot);
    const float k = grad_scale / nsel;
    for (int c = lane; c < C; c += 64) {
      const float v = k * p[c] * (g[c] - dot);
      d[c] = accumulate ? d[c] + v : v;
    }
  }
}

}  // namespace

extern "C" int srhip_row_max(const float* in, int in_is_probs, float* probs_out, float* max_probs, long long* max_idx, int B,
                             int C, void* stream) {
  if (B <= 0 || C <= 0) return SR_EINVAL;
  SR_LAUNCH(row_max_kernel, dim3(cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, in, in_is_probs, probs_out, max_probs, max_idx, B, C, 0,
                     0LL);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
extern "C" int srhip_row_max_strided(const float* in, int in_is_probs, float* probs_out, float* max_probs, long long* max_idx, int B, int C,
                                     int rows_per_group, long long group_stride, void* stream) {
  if (B <= 0 || C <= 0 || rows_per_group <= 0 || group_stride < (long long)rows_per_group * C) return SR_EINVAL;
  SR_LAUNCH(row_max_kernel, dim3(cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, in, in_is_probs, probs_out, max_probs, max_idx, B, C,
                     rows_per_group, group_stride);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_flexmatch_mask(const float* max_probs, const long long* max_idx, const long long* idx_ulb, float p_cutoff,
                                    long long* selected_label, int* hist, float* classwise_acc, float* mask, int B, int C,
                                    int ulb_dest_len, int thresh_warmup, void* stream) {
  return srhip_flexmatch_mask_passes(max_probs, max_idx, idx_ulb, p_cutoff, selected_label, hist, classwise_acc, mask, 1, B, C, ulb_dest_len,
                                     thresh_warmup, stream);
}
extern "C" int srhip_flexmatch_mask_passes(const float* max_probs, const long long* max_idx, const long long* idx_ulb, float p_cutoff,
                                           long long* selected_label, int* hist, float* classwise_acc, float* mask, int n_pass, int B, int C,
                                           int ulb_dest_len, int thresh_warmup, void* stream) {
  if (n_pass <= 0 || B <= 0 || C <= 0 || ulb_dest_len <= 0 || !idx_ulb) return SR_EINVAL;
  // state-in-LDS path only when its request fits the 160 KiB of a CU (minus the kernel's static LDS) AND the attribute call accepts it;
  // anything larger takes the general global-memory kernel instead of failing the launch
  const size_t smem = (size_t)(2 * C + 1) * 4 + (size_t)B * 6 * 4;
  bool lds_path = C <= 8192 && B <= 4096 && smem <= 160 * 1024 - 4096 && !getenv("SRHIP_FLEXMATCH_GENERAL");
  if (lds_path && smem > 48 * 1024)
    lds_path = hipFuncSetAttribute((const void*)flexmatch_mask_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) == hipSuccess;
  if (lds_path) {
    SR_LAUNCH(flexmatch_mask_lds_kernel, dim3(1), dim3(256), smem, (hipStream_t)stream, max_probs, max_idx, idx_ulb, p_cutoff,
                       selected_label, hist, classwise_acc, mask, B, C, ulb_dest_len, thresh_warmup, n_pass);
  } else {
    SR_LAUNCH(flexmatch_mask_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, max_probs, max_idx, idx_ulb, p_cutoff,
                       selected_label, hist, classwise_acc, mask, B, C, ulb_dest_len, thresh_warmup, n_pass);
  }
  SR_CHECK_LAUNCH();
  return SR_OK;
}
extern "C" int srhip_index_error(int* bits_out, int reset, void* stream) {
  if (!bits_out) return SR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemcpyFromSymbolAsync(bits_out, HIP_SYMBOL(srhip_index_err), sizeof(int), 0, hipMemcpyDeviceToHost, s) != hipSuccess) return SR_ELAUNCH;
  if (hipStreamSynchronize(s) != hipSuccess) return SR_ELAUNCH;
  if (reset && *bits_out) {
    const int z = 0;
    if (hipMemcpyToSymbolAsync(HIP_SYMBOL(srhip_index_err), &z, sizeof(int), 0, hipMemcpyHostToDevice, s) != hipSuccess) return SR_ELAUNCH;
    if (hipStreamSynchronize(s) != hipSuccess) return SR_ELAUNCH;
  }
  return SR_OK;
}

extern "C" int srhip_flexmatch_rebuild_hist(const long long* selected_label, int* hist, int ulb_dest_len, int C, void* stream) {
  if (ulb_dest_len <= 0 || C <= 0) return SR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(hist, 0, (size_t)(C + 1) * sizeof(int), s) != hipSuccess) return SR_ELAUNCH;
  SR_LAUNCH(flexmatch_hist_kernel, dim3(cdiv(ulb_dest_len, 256)), dim3(256), 0, s, selected_label, hist, ulb_dest_len, C);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_fixed_mask(const float* max_probs, float p_cutoff, float* mask, int B, void* stream) {
  if (B <= 0) return SR_EINVAL;
  SR_LAUNCH(fixed_mask_kernel, dim3(cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, max_probs, p_cutoff, mask, B);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_reward_mask2(const float* reward, float* mask2, float* mean_out, const float* mean_in, int groups, int B,
                                  void* stream) {
  if (groups <= 0 || B <= 0) return SR_EINVAL;
  SR_LAUNCH(reward_mask2_kernel, dim3(groups), dim3(64), 0, (hipStream_t)stream, reward, mask2, mean_out, mean_in, B);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_masked_ce(const float* logits, const long long* targets, const float* mask, const float* mask2,
                               float grad_scale, float* loss_out, float* dlogits, int B, int C, void* stream) {
  if (B <= 0 || C <= 0) return SR_EINVAL;
  SR_LAUNCH(masked_ce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, targets, mask, mask2, grad_scale,
                     loss_out, dlogits, B, C);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

// ---------------------------------------------------------------------------------------------
// SoftMatch (a12): DistAlignEMAHook.dist_align (semilearn/algorithms/hooks/dist_align.py:26-56) and
// SoftMatchWeightingHook.update/masking (semilearn/algorithms/srsoftmatch/utils.py:32-76, per_class = False).
// ONE workgroup each: B rows of C probabilities (B <= a few hundred per GPU).

// p_model (and p_target when colsum_lb != NULL) EMA from the column sums of the GLOBAL batch, then for the LOCAL rows
// aligned = probs * (p_target + 1e-6) / (p_model + 1e-6), renormalised, and its row max / argmax.
__global__ __launch_bounds__(256) void distalign_kernel(const float* __restrict__ probs, const float* __restrict__ colsum_ulb, int n_ulb,
                                                       const float* __restrict__ colsum_lb, int n_lb, float* __restrict__ p_model,
                                                       float* __restrict__ p_target, int* __restrict__ inited, float m, float one_minus_m,
                                                       float* __restrict__ aligned, float* __restrict__ max_probs,
                                                       long long* __restrict__ max_idx, int B, int C) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool first = (*inited == 0);
  for (int c = tid; c < C; c += 256) {
    const float mean_u = colsum_ulb[c] / (float)n_ulb;                                  // torch.mean(probs_x_ulb, dim=0)
    p_model[c] = first ? mean_u : p_model[c] * m + mean_u * one_minus_m;                // :49-52
    if (colsum_lb) p_target[c] = p_target[c] * m + (colsum_lb[c] / (float)n_lb) * one_minus_m;   // :54-56
  }
  __syncthreads();                            // p_model / p_target of this launch are visible to the whole workgroup (one WG)
  if (tid == 0) *inited = 1;
  for (int i = wave; i < B; i += 4) {         // one wave per row
    const float* pr = probs + (size_t)i * C;
    float* ar = aligned + (size_t)i * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float a = pr[c] * (p_target[c] + 1e-6f) / (p_model[c] + 1e-6f);             // :31
      ar[c] = a;
      s += a;
    }
    s = wave_sum(s);
    float best = -1.0f;
    int bi = 0;
    for (int c = lane; c < C; c += 64) {
      const float a = ar[c] / s;                                                        // :32
      ar[c] = a;
      if (a > best) { best = a; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {         // max with first-index tie break (torch.max)
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { max_probs[i] = best; max_idx[i] = bi; }
  }
}

// mu / var EMA of the max-probs of the GLOBAL batch (mean, unbiased variance), then the truncated-Gaussian weight of the LOCAL rows.
// The reference takes the batch statistics through .item(): (1 - m) * stat is a python double product that is rounded to fp32
// when it meets the fp32 state (utils.py:40-41); m * state is an fp32 product.
__global__ __launch_bounds__(256) void softmatch_mask_kernel(const float* __restrict__ maxp_all, int n_all, const float* __restrict__ max_probs,
                                                            float* __restrict__ mu_var, double m, int n_sigma, float* __restrict__ mask, int B) {
  __shared__ double red[4];
  __shared__ float st[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double s = 0.0;
  for (int i = tid; i < n_all; i += 256) s += (double)maxp_all[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float mean = (float)((red[0] + red[1] + red[2] + red[3]) / (double)n_all);      // torch.mean (fp32 result)
  __syncthreads();
  double q = 0.0;
  for (int i = tid; i < n_all; i += 256) { const double d = (double)maxp_all[i] - (double)mean; q += d * d; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  if (lane == 0) red[wave] = q;
  __syncthreads();
  if (tid == 0) {
    const float var = (float)((red[0] + red[1] + red[2] + red[3]) / (double)(n_all - 1));   // torch.var(unbiased=True); n_all = 1 -> nan as torch
    const float mf = (float)m;
    st[0] = mf * mu_var[0] + (float)((1.0 - m) * (double)mean);                          // :40
    st[1] = mf * mu_var[1] + (float)((1.0 - m) * (double)var);                           // :41
    mu_var[0] = st[0];
    mu_var[1] = st[1];
  }
  __syncthreads();
  const float mu = st[0], den = (2.0f * st[1]) / (float)(n_sigma * n_sigma);
  for (int i = tid; i < B; i += 256) {
    const float d = fminf(max_probs[i] - mu, 0.0f);                                     // torch.clamp(max=0.0)
    mask[i] = expf(-((d * d) / den));                                                   // :75
  }
}

extern "C" int srhip_freematch_stats(const float* probs, const long long* max_idx, float* colsum, float* hist, int B, int C, void* stream) {
  if (B <= 0 || C <= 0) return SR_EINVAL;
  SR_LAUNCH(freematch_stats_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, probs, max_idx, colsum, hist, B, C);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_freematch_update(const float* maxp_all, int n_all, const float* colsum, const float* hist, const float* max_probs,
                                      const long long* max_idx, float* time_p, float* p_model, float* label_hist, float* mask, int B,
                                      int C, float momentum, float one_minus_momentum, int use_quantile, int clip_thresh, void* stream) {
  if (B <= 0 || C <= 0 || n_all <= 0 || n_all > 1024) return SR_EINVAL;
  SR_LAUNCH(freematch_update_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, maxp_all, n_all, colsum, hist, max_probs, max_idx,
                     time_p, p_model, label_hist, mask, B, C, momentum, one_minus_momentum, use_quantile, clip_thresh);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_freematch_entropy(const float* logits, const float* mask, const float* p_model, const float* label_hist,
                                       float grad_scale, float* loss_out, float* dlogits, float* ws, int B, int C, int accumulate,
                                       void* stream) {
  if (B <= 0 || B > 1024 || C <= 0 || C > 2048) return SR_EINVAL;
  SR_LAUNCH(freematch_entropy_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, mask, p_model, label_hist, grad_scale,
                     loss_out, dlogits, ws, B, C, accumulate);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_distalign(const float* probs, const float* colsum_ulb, int n_ulb, const float* colsum_lb, int n_lb, float* p_model,
                               float* p_target, int* inited, double momentum, float* aligned, float* max_probs, long long* max_idx,
                               int B, int C, void* stream) {
  if (!probs || !colsum_ulb || !p_model || !p_target || !inited || !aligned || !max_probs || !max_idx) return SR_EINVAL;
  if (B <= 0 || C <= 0 || n_ulb <= 0 || (colsum_lb && n_lb <= 0)) return SR_EINVAL;
  SR_LAUNCH(distalign_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, probs, colsum_ulb, n_ulb, colsum_lb, n_lb, p_model,
                     p_target, inited, (float)momentum, (float)(1.0 - momentum), aligned, max_probs, max_idx, B, C);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_softmatch_mask(const float* maxp_all, int n_all, const float* max_probs, float* mu_var, double momentum, int n_sigma,
                                    float* mask, int B, void* stream) {
  if (!maxp_all || !max_probs || !mu_var || !mask || n_all <= 0 || B <= 0 || n_sigma <= 0) return SR_EINVAL;
  SR_LAUNCH(softmatch_mask_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, maxp_all, n_all, max_probs, mu_var, momentum, n_sigma,
                     mask, B);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
