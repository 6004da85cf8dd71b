// HIP streams confined to a subset of the compute units (hipExtStreamCreateWithCUMask), for the step's second stream: the workgroups of the
// row-streaming inference kernels own a CU each (147 KB of LDS, every VGPR) for ~100 us, so a launch train of them that may land on EVERY CU
// makes each small launch of the step's critical chain wait for one of them to retire.  Confined to 192 of the 256 CUs (24 per XCD) they leave
// 8 CUs per XCD that the chain finds free at any time.  (semilearn has no counterpart: one CUDA stream, train.py / algorithmbase.py:train.)
#include <hip/hip_runtime.h>

#include "common.h"

namespace {
__global__ void cu_probe_kernel(int* out, int spin) {
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[2 * blockIdx.x] = (int)(xcc & 15);
    out[2 * blockIdx.x + 1] = (int)hw;
  }
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}      // keep the CU busy so that the launch spreads over every CU it may use
}
}  // namespace

// cu_mask: bit i of the little-endian bit string = compute unit i in the driver's numbering (MI355X: bit i belongs to XCD i % 8, the bits of
// one XCD spread over its shader engines); words = number of 32-bit words.  *stream_out receives a hipStream_t.
extern "C" int srhip_stream_create_cu_mask(const unsigned* cu_mask, int words, void** stream_out) {
  if (!cu_mask || words <= 0 || !stream_out) return SR_EINVAL;
  hipStream_t s = nullptr;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)words, cu_mask) != hipSuccess) { (void)hipGetLastError(); return SR_EINVAL; }
  *stream_out = (void*)s;
  return SR_OK;
}

extern "C" int srhip_stream_destroy(void* stream) {
  if (!stream) return SR_EINVAL;
  return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? SR_OK : SR_EINVAL;
}

// Diagnostic: n workgroups of one wave, each spinning ``spin_ticks`` of the 100 MHz wall clock; out[2 i] = XCD, out[2 i + 1] = HW_ID register
// (cu_id bits 11:8, sh_id 12, se_id 15:13) of workgroup i.  Shows which CUs a (masked) stream dispatches to.
extern "C" int srhip_cu_probe(int* out, int n, int spin_ticks, void* stream) {
  if (!out || n <= 0) return SR_EINVAL;
  hipLaunchKernelGGL(cu_probe_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, out, spin_ticks);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
