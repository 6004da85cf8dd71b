// Fused multi-tensor AdamW over the flat fp32 parameter block (+ bf16 shadow copy, + EMA, + zero_grad).
//
// Reference:
//   ParamUpdateHook.after_train_step  semilearn/core/hooks/param_update.py:33-40  (step, scheduler, zero_grad)
//   get_optimizer / layer decay       semilearn/core/utils/build.py:193-224, semilearn/nets/utils.py:143-204
//   EMA.update                        semilearn/core/utils/misc.py:152-155 (+ two load_state_dict, core/hooks/ema.py:20-24)
// The reference walks 152 tensors in 28 param groups from Python and then copies every parameter three
// more times for the EMA hook.  Here the 21.4 M parameters live in ONE flat block; one launch reads
// p,g,m,v and writes p,m,v (28 B/param), the bf16 GEMM operand copy (2 B) and optionally the EMA shadow,
// and clears the gradient.  Per-tensor lr / weight-decay come from a chunk table built once on the host:
// chunk -> (offset, length, tensor id); a workgroup never straddles two tensors.  HBM-bound by design.
#include "common.h"
#include "srhip.h"

namespace {

__global__ __launch_bounds__(256) void adamw_flat_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, bf16_t* __restrict__ pb, float* __restrict__ ema,
                                                        const int4* __restrict__ chunks, const float* __restrict__ lr_t,
                                                        const float* __restrict__ wd_t, float lr_factor, float b1, float b2,
                                                        float eps, float bc1, float bc2_sqrt, float ema_m, float ema_1m, float grad_scale,
                                                        const float* __restrict__ clip_coef, int zero_grad, const float* __restrict__ dyn) {
  // dyn != NULL (srhip_adamw_flat_dyn): this step's (lr_factor, 1 - beta1^t, sqrt(1 - beta2^t)) come from device memory, so that a launch
  // captured in a HIP graph serves every step
  if (dyn) { lr_factor = dyn[0]; bc1 = dyn[1]; bc2_sqrt = dyn[2]; }
  const int4 ck = chunks[blockIdx.x];                 // x = offset, y = length, z = tensor id
  const float lr = lr_t[ck.z] * lr_factor, wd = wd_t[ck.z];
  const float decay = 1.0f - lr * wd, step = lr / bc1;
  if (clip_coef) grad_scale *= clip_coef[0];          // clip_grad_norm_ coefficient of this step (srhip_clip_grad_coef), stays on the device
  for (int i = threadIdx.x; i < ck.y; i += 256) {
    const size_t o = (size_t)ck.x + i;
    const float gi = g[o] * grad_scale;
    float pi = p[o] * decay;
    const float mi = b1 * m[o] + (1.0f - b1) * gi;
    const float vi = b2 * v[o] + (1.0f - b2) * gi * gi;
    pi -= step * mi / (sqrtf(vi) / bc2_sqrt + eps);
    p[o] = pi; m[o] = mi; v[o] = vi;
    if (pb) pb[o] = f2bf(pi);
    // misc.py:154 ``(1.0 - decay) * param + decay * shadow``: 1 - decay is rounded from double, three separately rounded fp32 ops
    if (ema) ema[o] = __fadd_rn(__fmul_rn(ema_1m, pi), __fmul_rn(ema_m, ema[o]));
    if (zero_grad) g[o] = 0.f;
  }
}

// torch.nn.utils.clip_grad_norm_(parameters, max_norm) (param_update.py:34-35): total = || pre_scale * g ||_2 over the whole flat block,
// coef = min(1, max_norm / (total + 1e-6)).  Two deterministic stages (fixed partition, fixed tree): partial sums of squares per
// workgroup, then one workgroup folds them in double and writes the coefficient.  The optimizer launch multiplies by it.
constexpr int CLIP_WG = 1024;
__global__ __launch_bounds__(256) void sumsq_part_kernel(const float* __restrict__ g, size_t n, float* __restrict__ part) {
  const size_t per = (n + CLIP_WG - 1) / CLIP_WG, lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  float s = 0.f;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) { const float v = g[i]; s = fmaf(v, v, s); }
  s = wave_sum(s);
  __shared__ float sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* __restrict__ part, float pre_scale, float max_norm, float* __restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < CLIP_WG; i += 256) s += (double)part[i];
  __shared__ double sh[256];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) {
    const float total = pre_scale * (float)sqrt(sh[0]);
    const float c = max_norm / (total + 1e-6f);
    out[0] = c < 1.0f ? c : 1.0f;         // a non-finite total gives NaN here and NaN parameters, as error_if_nonfinite=False does in torch
    out[1] = total;
  }
}

}  // namespace

extern "C" int srhip_clip_grad_ws_floats(void) { return CLIP_WG; }
extern "C" int srhip_clip_grad_coef(const float* g, long long n, float pre_scale, float max_norm, float* ws, float* coef_out, void* stream) {
  if (!g || n <= 0 || !ws || !coef_out || !(max_norm > 0.f)) return SR_EINVAL;
  SR_LAUNCH(sumsq_part_kernel, dim3(CLIP_WG), dim3(256), 0, (hipStream_t)stream, g, (size_t)n, ws);
  SR_LAUNCH(clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, pre_scale, max_norm, coef_out);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_adamw_flat(float* p, float* g, float* m, float* v, void* p_bf16, float* ema, const int* chunk_table,
                                int n_chunks, const float* lr_t, const float* wd_t, float lr_factor, float beta1, float beta2,
                                float eps, int step, double ema_m, float grad_scale, const float* clip_coef, int zero_grad, void* stream) {
  if (n_chunks <= 0 || step <= 0) return SR_EINVAL;
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
  SR_LAUNCH(adamw_flat_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)p_bf16, ema,
                     (const int4*)chunk_table, lr_t, wd_t, lr_factor, beta1, beta2, eps, bc1, bc2s, (float)ema_m, (float)(1.0 - ema_m), grad_scale, clip_coef, zero_grad,
                     (const float*)nullptr);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
// (1 - beta1^step, sqrt(1 - beta2^step)) exactly as the by-value launches compute them (host function, fp32 powf): what the host writes into the
// dyn block of a step, so that the graph-replayed step and the eager step agree bit for bit
extern "C" int srhip_adam_bias_corrections(float beta1, float beta2, int step, float* out2) {
  if (!out2 || step <= 0) return SR_EINVAL;
  out2[0] = 1.0f - powf(beta1, (float)step);
  out2[1] = sqrtf(1.0f - powf(beta2, (float)step));
  return SR_OK;
}
// The same launch with the per-step scalars in DEVICE memory: dyn = {lr_factor, 1 - beta1^step, sqrt(1 - beta2^step)} (fp32, written by the host
// before the step: semireward_amd/core/stepgraph.py).  Everything else is step-invariant, so the launch can sit in a captured HIP graph.
extern "C" int srhip_adamw_flat_dyn(float* p, float* g, float* m, float* v, void* p_bf16, float* ema, const int* chunk_table,
                                    int n_chunks, const float* lr_t, const float* wd_t, const float* dyn, float beta1, float beta2,
                                    float eps, double ema_m, float grad_scale, const float* clip_coef, int zero_grad, void* stream) {
  if (n_chunks <= 0 || !dyn) return SR_EINVAL;
  SR_LAUNCH(adamw_flat_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)p_bf16, ema,
                     (const int4*)chunk_table, lr_t, wd_t, 0.f, beta1, beta2, eps, 1.f, 1.f, (float)ema_m, (float)(1.0 - ema_m), grad_scale, clip_coef, zero_grad, dyn);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

