// Probe of ds_read_b64_tr_b16 semantics on gfx950 (not part of the product library).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
__global__ void k(int mode, int rs, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x, l15 = l & 15, g = l >> 4;
  int el;
  if (mode == 0) el = 4 * l;                                        // linear
  else el = (8 * g + (l15 >> 2)) * rs + 4 * (l15 & 3);              // [4 keys][16 cols] block per 16-lane group, row stride rs
  const unsigned addr = (unsigned)(size_t)(&lds[0]) + el * 2;
  u32x2_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[4 * l + 0] = v[0] & 0xffff; out[4 * l + 1] = v[0] >> 16; out[4 * l + 2] = v[1] & 0xffff; out[4 * l + 3] = v[1] >> 16;
}
int main() {
  unsigned short* d; hipMalloc(&d, 512);
  unsigned short h[256];
  for (int mode = 0; mode < 2; ++mode) {
    const int rs = 64;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, mode, rs, d);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("mode %d (rs=%d): lane: 4 returned element indices", mode, rs);
    if (mode) printf(" as (row,col)");
    printf("\n");
    for (int l = 0; l < 64; ++l) {
      printf("  l%02d:", l);
      for (int j = 0; j < 4; ++j) { if (mode) printf(" (%d,%d)", h[4*l+j] / rs, h[4*l+j] % rs); else printf(" %4d", h[4*l+j]); }
      if ((l & 3) == 3) printf("\n");
    }
  }
  return 0;
}
