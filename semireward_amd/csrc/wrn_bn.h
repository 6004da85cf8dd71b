// BatchNorm batch statistics shared by the statistics pass (wrn_ops.hip) and the convolution epilogue (wrn_conv.hip).
//
// ws layout (doubles): [0, 512) totals (sum | sum of squares, 2C values) | [512] arrival counter (low 32 bits) | [520, 520 + 16 * 512)
// accumulator copies.  A producer workgroup adds its 2C partial sums with hardware fp64 atomics into copy (workgroup % BN_COPIES) -- one copy
// for everybody was 512 workgroups queueing on four cache lines, 10 of the pass's 13 us -- and arrives; the workgroup that arrives last folds
// the copies into the totals and leaves copies and counter at ZERO (no memset launch per BatchNorm: the caller zeroes ws once).
// (fp64 sums are order-dependent in the last bit of a double, 29 bits below the float the statistics are rounded to.)
#pragma once
#include "common.h"

namespace {

constexpr int BN_COPIES = 16, BN_WS_COUNTER = 512, BN_WS_ACC = 520;

struct BnFinal {                 // what the last workgroup derives from the totals (out_mean == NULL: nothing)
  float* out_mean; float* out_invstd; float* running_mean; float* running_var;
  float momentum; int update_running; float eps;
};

// Call after this workgroup's atomic adds into its accumulator copy.  Returns true in the last workgroup of the launch (all threads), with
// the totals in ws[0..2C) and in red2c (LDS, >= 2C doubles).
__device__ __forceinline__ bool bn_arrive_and_fold(double* __restrict__ ws, double* red2c, int C, unsigned n_workgroups, unsigned my_index,
                                                   int* is_last_lds) {
  (void)my_index;
  // The adds must be performed before the arrival is counted.  They are device-scope atomics (performed at the memory side, not held dirty
  // in this XCD's L2), so waiting for their acknowledgement is enough: a workgroup-scope release is that s_waitcnt and nothing else.
  // __threadfence() here is a device-scope release = buffer_wbl2, a walk over the XCD's 4 MB L2 (~1 us, serialised per XCD): 64 workgroups
  // per XCD each paying for one made the statistics pass 35 us instead of 11.  (MI355X_MICROARCH.md, inter-workgroup visibility: 8-byte
  // device-scope atomics on both sides are a valid hand-off without fences.)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (explicit: the fence above may lower to nothing)
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* counter = reinterpret_cast<unsigned*>(ws + BN_WS_COUNTER);
    const unsigned seen = atomicAdd(counter, 1u);
    *is_last_lds = seen == n_workgroups - 1;
    if (seen == n_workgroups - 1) *counter = 0u;            // (everybody has arrived: nobody touches it again in this launch)
  }
  __syncthreads();
  if (!*is_last_lds) return false;
  // (acquire side: device-scope atomic loads read at the memory side; nothing of this data is in a cache of this workgroup)
  for (int o = threadIdx.x; o < 2 * C; o += blockDim.x) {
    double u[BN_COPIES], t = 0.0;
#pragma unroll
    for (int q = 0; q < BN_COPIES; ++q) u[q] = __hip_atomic_load(ws + BN_WS_ACC + (size_t)q * 512 + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int q = 0; q < BN_COPIES; ++q) {
      t += u[q];
      __hip_atomic_store(ws + BN_WS_ACC + (size_t)q * 512 + o, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ws[o] = t;
    red2c[o] = t;
  }
  __syncthreads();
  return true;
}

// mean / invstd of the batch (biased variance) and the running update (unbiased variance), as bn_apply_kernel computes them
__device__ __forceinline__ void bn_finalize(const double* red2c, int C, int rows, const BnFinal& f) {
  if (!f.out_mean) return;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const double m = red2c[c] / rows, v = red2c[C + c] / rows - m * m;
    const float mu = (float)m, var = (float)(v > 0.0 ? v : 0.0);
    f.out_mean[c] = mu;
    f.out_invstd[c] = 1.0f / sqrtf(var + f.eps);
    if (f.update_running) {
      const float unb = (float)((v > 0.0 ? v : 0.0) * ((double)rows / (double)(rows > 1 ? rows - 1 : 1)));
      f.running_mean[c] = (1.0f - f.momentum) * f.running_mean[c] + f.momentum * mu;
      f.running_var[c] = (1.0f - f.momentum) * f.running_var[c] + f.momentum * unb;
    }
  }
}

}  // namespace
