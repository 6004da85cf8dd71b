// Probe of v_permlane16_swap lane semantics on gfx950 (hipcc --offload-arch=gfx950 -O3 tools/pl_probe.hip -o tools/pl_probe): prints, per lane,
// the two results of __builtin_amdgcn_permlane16_swap(a = lane, b = 1000 + lane).  Result on MI355X: r[0] = {a(0-15), b(0-15), a(32-47), b(32-47)},
// r[1] = {a(16-31), b(16-31), a(48-63), b(48-63)} -- what the widened GEMM epilogue stores rely on (csrc/gemm.hip: store_quad_pair).
#include <hip/hip_runtime.h>
__global__ void k(unsigned* out) {
  unsigned a = threadIdx.x, b = 1000 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[threadIdx.x] = r[0];
  out[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 128 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; ++i) printf("%d:%u/%u ", i, h[i], h[64 + i]);
  printf("\n");
  return 0;
}
