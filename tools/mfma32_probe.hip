// Standalone probe: operand and result layout of v_mfma_f32_32x32x16_bf16 on gfx950 (for a 4-wave x 32-row shape of the fused MLP kernel).
//   claim checked: A operand lane L supplies row L % 32, k = 8 (L / 32) .. + 7; B operand lane L supplies column L % 32, same k slots;
//                  result register r of lane L = D[8 (r / 4) + 4 (L / 32) + r % 4][L % 32].
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__global__ void k(float* out, int mode) {
  const int L = threadIdx.x, rc = L % 32, kh = L / 32;
  bf16x8_t a, b;
  for (int e = 0; e < 8; ++e) {
    const int kk = 8 * kh + e;                      // k index this slot is claimed to be
    // mode 0: A[i][k] = (k == 3) * i, B[k][j] = (k == 3)      -> D[i][j] = i
    // mode 1: A[i][k] = (k == 11),    B[k][j] = (k == 11) * j -> D[i][j] = j
    // mode 2: A[i][k] = k + 1 (i == 5 only), B[k][j] = (j == 7 && k == 9) -> D[5][7] = 10, everything else 0  (k slot order of both operands)
    float av, bv;
    if (mode == 0) { av = kk == 3 ? (float)rc : 0.f; bv = kk == 3 ? 1.f : 0.f; }
    else if (mode == 1) { av = kk == 11 ? 1.f : 0.f; bv = kk == 11 ? (float)rc : 0.f; }
    else { av = rc == 5 ? (float)(kk + 1) : 0.f; bv = (rc == 7 && kk == 9) ? 1.f : 0.f; }
    a[e] = (__bf16)av; b[e] = (__bf16)bv;
  }
  f32x16_t c;
  for (int e = 0; e < 16; ++e) c[e] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int e = 0; e < 16; ++e) out[L * 16 + e] = c[e];
}

int main() {
  float* d; hipMalloc(&d, 64 * 16 * 4);
  float h[64 * 16];
  int bad = 0;
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int L = 0; L < 64; ++L)
      for (int r = 0; r < 16; ++r) {
        const int i = 8 * (r / 4) + 4 * (L / 32) + r % 4, j = L % 32;
        const float want = mode == 0 ? (float)i : mode == 1 ? (float)j : ((i == 5 && j == 7) ? 10.f : 0.f);
        if (h[L * 16 + r] != want) { if (bad < 8) printf("mode %d lane %d reg %d: got %g want %g\n", mode, L, r, h[L * 16 + r], want); ++bad; }
      }
  }
  printf(bad ? "layout claim WRONG (%d mismatches)\n" : "layout claim holds: A lane L = row L%%32, k 8(L/32)..+7; B lane L = column L%%32, same k; D reg r of lane L = [8(r/4) + 4(L/32) + r%%4][L%%32]  (%d mismatches)\n", bad);
  return 0;
}
