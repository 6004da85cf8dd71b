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
                                                        float eps, float bc1, float bc2_sqrt, float ema_m, float grad_scale, int zero_grad) {
  const int4 ck = chunks[blockIdx.x];                 // x = offset, y = length, z = tensor id
  const float lr = lr_t[ck.z] * lr_factor, wd = wd_t[ck.z];
  const float decay = 1.0f - lr * wd, step = lr / bc1;
  for (int i = threadIdx.x; i < ck.y; i += 256) {
    const size_t o = (size_t)ck.x + i;
    const float gi = g[o] * grad_scale;
    float pi = p[o] * decay;
    const float mi = b1 * m[o] + (1.0f - b1) * gi;
    const float vi = b2 * v[o] + (1.0f - b2) * gi * gi;
    pi -= step * mi / (sqrtf(vi) / bc2_sqrt + eps);
    p[o] = pi; m[o] = mi; v[o] = vi;
    if (pb) pb[o] = f2bf(pi);
    if (ema) ema[o] = (1.0f - ema_m) * pi + ema_m * ema[o];
    if (zero_grad) g[o] = 0.f;
  }
}

}  // namespace

extern "C" int srhip_adamw_flat(float* p, float* g, float* m, float* v, void* p_bf16, float* ema, const int* chunk_table,
                                int n_chunks, const float* lr_t, const float* wd_t, float lr_factor, float beta1, float beta2,
                                float eps, int step, float ema_m, float grad_scale, int zero_grad, void* stream) {
  if (n_chunks <= 0 || step <= 0) return SR_EINVAL;
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adamw_flat_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)p_bf16, ema,
                     (const int4*)chunk_table, lr_t, wd_t, lr_factor, beta1, beta2, eps, bc1, bc2s, ema_m, grad_scale, zero_grad);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

