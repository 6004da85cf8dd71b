// Standalone probe (not part of the product library): can ds_read_b128 traffic and MFMA issue overlap on a gfx950 CU, and
// at what rates?  One workgroup per CU (512 threads = 2 waves per SIMD unless noted), `iters` iterations of
//   R fragments read from LDS (16 B per lane, conflict-free image) and/or 8 MFMAs 16x16x32 bf16.
// MODE 0: reads only (consumed by a cheap VALU op).      MODE 1: MFMAs only (operands in registers).
// MODE 2: reads + MFMAs, MFMAs consume the fragments read in the PREVIOUS iteration (software pipelined).
// MODE 3: reads + MFMAs, MFMA operands independent of the reads (pure issue/port interference).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

__device__ __forceinline__ int swz(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

template <int MODE, int R, int NT>
__global__ __launch_bounds__(NT) void k(int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, l15 = lane & 15, lg = lane >> 4;
  for (int i = threadIdx.x; i < 32768; i += NT) ((float*)lds)[i] = (float)i;
  __syncthreads();
  const int fo = (l15 * 32 + ((lg ^ swz(l15)) << 3)) * 2;       // bytes, [rows][32 bf16] image as in gemm.hip
  f32x4_t acc[8];
  for (int t = 0; t < 8; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  u32x4_t fa[8], fb[8];
  for (int t = 0; t < 8; ++t) { fa[t] = u32x4_t{1u, 2u, 3u, (unsigned)lane}; fb[t] = fa[t]; }
  const u32x4_t xb = u32x4_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  unsigned vs = 0;
  for (int it = 0; it < iters; ++it) {
    const char* st = lds + ((it & 7) * 8192);
    if (MODE == 0) {
#pragma unroll
      for (int t = 0; t < R; ++t) fa[t & 7] = *(const u32x4_t*)(st + fo + (t & 7) * 1024);
#pragma unroll
      for (int t = 0; t < (R < 8 ? R : 8); ++t) vs += fa[t][0];
    } else if (MODE == 1) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[t]), __builtin_bit_cast(bf16x8_t, xb), acc[t], 0, 0, 0);
    } else if (MODE == 2) {
#pragma unroll
      for (int t = 0; t < R; ++t) fb[t & 7] = *(const u32x4_t*)(st + fo + (t & 7) * 1024);
#pragma unroll
      for (int t = 0; t < 8; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[t]), __builtin_bit_cast(bf16x8_t, xb), acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 8; ++t) fa[t] = fb[t];
    } else {
#pragma unroll
      for (int t = 0; t < R; ++t) fb[t & 7] = *(const u32x4_t*)(st + fo + (t & 7) * 1024);
#pragma unroll
      for (int t = 0; t < 8; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[t]), __builtin_bit_cast(bf16x8_t, xb), acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < (R < 8 ? R : 8); ++t) vs += fb[t][0];
    }
  }
  float s = (float)vs;
  for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][3];
  if (s == 123.456f) sink[0] = s;
}

template <int MODE, int R, int NT>
static void run(const char* name, int iters) {
  float* sink; hipMalloc(&sink, 4);
  auto kern = k<MODE, R, NT>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(kern, dim3(256), dim3(NT), 131072, 0, iters, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(kern, dim3(256), dim3(NT), 131072, 0, iters, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us = ms * 1e3, waves = NT / 64;
  const double rd = (MODE == 1) ? 0 : (double)iters * R * 1024 * waves;            // LDS bytes per CU
  const double fl = (MODE == 0) ? 0 : (double)iters * 8 * 16384 * waves;           // flop per CU
  printf("%-34s %8.1f us | %7.1f ns/iter | LDS %6.1f B/clk/CU (@2.4GHz) | MFMA %6.1f TF/s chip\n", name, us, us * 1e3 / iters,
         rd / (us * 1e-6) / 2.4e9, fl * 256 / (us * 1e-6) / 1e12);
  hipFree(sink);
}

int main() {
  const int it = 20000;
  run<0, 8, 512>("reads only, 8 waves, 8 frag/it", it);
  run<0, 8, 256>("reads only, 4 waves, 8 frag/it", it);
  run<1, 8, 512>("mfma only, 8 waves", it);
  run<1, 8, 256>("mfma only, 4 waves", it);
  run<2, 8, 512>("pipelined 8 frag : 8 mfma, 8 waves", it);
  run<2, 4, 512>("pipelined 4 frag : 8 mfma, 8 waves", it);
  run<2, 2, 512>("pipelined 2 frag : 8 mfma, 8 waves", it);
  run<2, 8, 256>("pipelined 8 frag : 8 mfma, 4 waves", it);
  run<2, 4, 256>("pipelined 4 frag : 8 mfma, 4 waves", it);
  run<3, 8, 512>("independent 8 frag : 8 mfma, 8 waves", it);
  run<3, 4, 512>("independent 4 frag : 8 mfma, 8 waves", it);
  run<2, 8, 1024>("pipelined 8 frag : 8 mfma, 16 waves", it);
  run<0, 8, 1024>("reads only, 16 waves", it);
  return 0;
}
