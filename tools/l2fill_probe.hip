// Standalone probe of the global->LDS fill path on MI355X by SOURCE SIZE (L2-resident 1-3 MiB vs 8 / 64 MiB) and by whether all
// workgroups walk the same stream (as the weight tiles of the fused MLP kernel) -- not part of the product library.
// Every workgroup streams `steps` stages of `stage_bytes` from an L2-resident source through an LDS ring.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
template <int N_> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

// MODE 0: LDS-DMA.  MODE 1: global_load_dwordx4 -> VGPR -> ds_write_b128 (register ring of depth PD).
// SEG = contiguous bytes per row segment (64 / 128 / 1024); rows are `row_stride` bytes apart.
// Each wave moves IPW KiB per step; NS ring stages, PD = NS-1 steps in flight.
template <int MODE, int SEG, int IPW, int NS>
__global__ __launch_bounds__(512) void fill_kernel(const char* __restrict__ src, size_t src_bytes, int row_stride, int steps, float* sink, int same_stream) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int PD = NS - 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int stage_bytes = nw * IPW * 1024;
  constexpr int LPR = SEG / 16;                       // lanes per row segment
  const size_t wg_off = same_stream ? 0 : ((size_t)blockIdx.x * 7919 * 4096) % (src_bytes / 2);
  auto src_of = [&](int step, int i) {
    const int rows_per_inst = 64 / LPR;
    const size_t row = ((size_t)step * nw * IPW + wave * IPW + i) * rows_per_inst + lane / LPR;
    return src + (wg_off + row * row_stride + (lane % LPR) * 16) % (src_bytes - 4096);
  };
  float acc = 0.f;
  if (MODE == 0) {
    auto issue = [&](int step, int st) {
#pragma unroll
      for (int i = 0; i < IPW; ++i)
        __builtin_amdgcn_global_load_lds((gbl_void*)src_of(step, i), (lds_void*)(lds + st * stage_bytes + (wave * IPW + i) * 1024), 16, 0, 0);
    };
    for (int p = 0; p < PD && p < steps; ++p) issue(p, p);
    for (int s = 0; s < steps; ++s) {
      if (steps - 1 - s >= PD - 1) wait_vm<(PD - 1) * IPW>(); else wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      if (s + PD < steps) issue(s + PD, (s + PD) % NS);
      acc += *(const float*)(lds + (s % NS) * stage_bytes + threadIdx.x * 16);
    }
  } else {
    u32x4_t r[PD][IPW];
    auto ld = [&](int step, int slot) {
#pragma unroll
      for (int i = 0; i < IPW; ++i) r[slot][i] = *(const u32x4_t*)src_of(step, i);
    };
#pragma unroll
    for (int p = 0; p < PD; ++p) ld(p, p);
    for (int s0 = 0; s0 < steps; s0 += PD) {
#pragma unroll
      for (int p = 0; p < PD; ++p) {
        const int s = s0 + p;
#pragma unroll
        for (int i = 0; i < IPW; ++i) *(u32x4_t*)(lds + (s & 1) * stage_bytes + (wave * IPW + i) * 1024 + lane * 16) = r[p][i];
        if (s + PD < steps) ld(s + PD, p);
        __syncthreads();
        acc += *(const float*)(lds + (s & 1) * stage_bytes + threadIdx.x * 16);
      }
    }
  }
  if (acc == 1234.5f) sink[0] = acc;
}

static int g_same = 0;
template <int MODE, int SEG, int IPW, int NS>
void run(const char* name, int threads, const char* src, size_t src_bytes, int row_stride, float* sink, int wgs_per_cu, int cus = 256) {
  const int nw = threads / 64, steps = 400, grid = cus * wgs_per_cu;
  const size_t lds = (size_t)(MODE == 0 ? NS : 2) * nw * IPW * 1024;
  auto k = fill_kernel<MODE, SEG, IPW, NS>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, 0, src, src_bytes, row_stride, steps, sink, g_same);
    hipEventRecord(b); hipEventSynchronize(b);
  }
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)grid * steps * nw * IPW * 1024;
  printf("%-34s thr=%3d wg/cu=%d seg=%4d KiB/wave/step=%d stages=%d lds=%3zuK : %7.1f us  %6.2f TB/s  %5.1f GB/s/CU  %4.1f B/clk/CU\n", name, threads,
         wgs_per_cu, SEG, IPW, NS, lds / 1024, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / cus, bytes / ms / 1e6 / cus / 2.4);
  if (hipGetLastError() != hipSuccess) printf("  (launch error)\n");
}


int main(int argc, char** argv) {
  float* sink; hipMalloc(&sink, 64);
  const size_t sizes[] = {1u << 20, 2u << 20, 3u << 20, 8u << 20, 64u << 20};
  for (int same = 1; same >= 0; --same) {
    g_same = same;
    for (size_t sb : sizes) {
      char* src; hipMalloc(&src, sb); hipMemset(src, 1, sb);
      printf("---- source %zu MiB, %s\n", sb >> 20, same ? "all workgroups read the SAME stream (weights)" : "every workgroup its own offset");
      run<0, 128, 4, 4>("dma seg128 4KiB/wave ring4 8w", 512, src, sb, 768, sink, 1);
      run<0, 1024, 4, 4>("dma seg1024 4KiB/wave ring4 8w", 512, src, sb, 1024, sink, 1);
      run<0, 64, 4, 4>("dma seg64 4KiB/wave ring4 8w", 512, src, sb, 64, sink, 1);
      run<0, 128, 2, 8>("dma seg128 2KiB/wave ring8 8w", 512, src, sb, 768, sink, 1);
      // the row pitches of the fused MLP kernel's ring stages: fc1 weight rows (64 B pieces, 768 B apart), fc2 weight rows (64 B, 3072 B apart)
      run<0, 64, 4, 4>("dma seg64 pitch 768 (W1 stage)", 512, src, sb, 768, sink, 1);
      run<0, 64, 4, 4>("dma seg64 pitch 3072 (W2 stage)", 512, src, sb, 3072, sink, 1);
      run<0, 64, 4, 4>("dma seg64 pitch 3072 on 32 WGs", 512, src, sb, 3072, sink, 1, 32);
      run<0, 64, 1, 16>("dma seg64 pitch 3072 1KiB/wave ring16", 512, src, sb, 3072, sink, 1);
      run<0, 64, 1, 16>("dma seg64 pitch 64 1KiB/wave ring16", 512, src, sb, 64, sink, 1);
      run<0, 128, 4, 4>("dma 8w ring4 on 32 WGs", 512, src, sb, 768, sink, 1, 32);
      run<1, 128, 4, 5>("reg seg128 4KiB/wave pd4 8w", 512, src, sb, 768, sink, 1);
      run<1, 128, 4, 5>("reg seg128 4KiB/wave pd4 8w x2", 512, src, sb, 768, sink, 2);
      hipFree(src);
    }
  }
  return 0;
}
