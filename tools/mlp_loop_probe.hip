// Standalone probe (not part of the product library): the main loop of the fused MLP kernel as a skeleton -- a 16-slot LDS-DMA ring of 8 KB weight
// stages (group syncs of 4 stages: counted vmcnt + s_barrier + refill), one LDS fragment read per MFMA (one half-stage ahead), a filler of
// dependent packed-fp32 operations standing in for the GELU -- in two shapes of the same 128-row tile:
//   A: 8 waves x 16 rows, v_mfma_f32_16x16x32_bf16 (two waves per SIMD, 256 VGPRs)   -- the kernel as built
//   B: 4 waves x 32 rows, v_mfma_f32_32x32x16_bf16 (one wave per SIMD, up to 512 registers): half the LDS reads and half the issue slots per MAC
// Prints time per 12-stage "chunk" (96 KB of weights, 96 MFMAs per wave) with every CU busy.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
template <int N_> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

constexpr int NS = 16, GS = 4, NG = NS / GS, ST = 8192;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ f32x2_t gelu_poly2(f32x2_t v) {                 // as in the library (common.h)
  constexpr float XS = 4.252893f;
  const f32x2_t xc = {__builtin_amdgcn_fmed3f(v[0], -XS, XS), __builtin_amdgcn_fmed3f(v[1], -XS, XS)};
  const f32x2_t t = xc * xc;
  auto sp = [](float c) { return f32x2_t{c, c}; };
  f32x2_t p = sp(5.564872638e-11f);
  p = __builtin_elementwise_fma(p, t, sp(-5.327768675e-09f));
  p = __builtin_elementwise_fma(p, t, sp(2.255431416e-07f));
  p = __builtin_elementwise_fma(p, t, sp(-5.626433893e-06f));
  p = __builtin_elementwise_fma(p, t, sp(9.341875929e-05f));
  p = __builtin_elementwise_fma(p, t, sp(-1.108561217e-03f));
  p = __builtin_elementwise_fma(p, t, sp(9.815971766e-03f));
  p = __builtin_elementwise_fma(p, t, sp(-6.634449185e-02f));
  p = __builtin_elementwise_fma(p, t, sp(3.989023390e-01f));
  const f32x2_t phi = __builtin_elementwise_fma(p, xc, sp(0.5f));
  return v * phi;
}          // ring slots, stages per group, groups, bytes per stage

// BIG = false: shape A (blockDim 512), BIG = true: shape B (blockDim 256).  VF = filler operations (packed fma) per stage and wave.
template <bool BIG, int VF, bool DMA, bool RD, int VAR = 0>
__global__ __launch_bounds__(BIG ? 256 : 512) void probe(const char* __restrict__ w, size_t wbytes, int chunks, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int NW = BIG ? 4 : 8, IPS = 8 / NW;                  // waves, DMA instructions per wave and stage
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < NS * ST / 4; i += blockDim.x) ((float*)lds)[i] = 1.0f;
  __syncthreads();
  using acc_t = typename std::conditional<BIG, f32x16_t, f32x4_t>::type;
  constexpr int NACC = BIG ? 14 : 28;                            // output tiles + the hidden chunk's tiles
  acc_t acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t) for (int e = 0; e < (BIG ? 16 : 4); ++e) acc[t][e] = 0.f;
  u32x4_t xb = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u + (unsigned)lane};
  f32x2_t fill[4] = {{1.f, 2.f}, {3.f, 4.f}, {5.f, 6.f}, {7.f, 8.f}};
  const int nst = chunks * 12;
  // as in the kernel: buffer addressing = resource (SGPRs) + per-lane byte offset (one VGPR, loop-invariant) + uniform byte offset (SGPR)
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(w), 0, (int)wbytes, 0x00020000);
  const int lo = lane * 16;
  // VAR 6 (shape A): the source pattern of the real kernel's stages -- a wave instruction fetches 16 rows x 64 B, rows `pitch` bytes apart
  const int lo_str = (16 * wave + (lane >> 2)) * 768 + (lane & 3) * 16;
  auto issue = [&](int s) __attribute__((always_inline)) {
    if (!DMA) return;
    if constexpr (VAR == 6) {
      const unsigned so = (((unsigned)s % 12u) * 64u + ((unsigned)s / 12u) * 128u * 768u) & (unsigned)(wbytes / 2 - 1);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(lds + (s & (NS - 1)) * ST + wave * 1024), 16, lo_str, (int)so, 0, 0);
      return;
    }
#pragma unroll
    for (int i = 0; i < IPS; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(lds + (s & (NS - 1)) * ST + (wave * IPS + i) * 1024), 16, lo,
                                               (int)((((unsigned)s * ST) & (unsigned)(wbytes / 2 - 1)) + (wave * IPS + i) * 1024), 0, 0);
  };
  constexpr int PD = (NG - 2) * GS;
  for (int p = 0; p < PD; ++p) issue(p);
  u32x4_t fa[VAR == 7 ? 3 : 2][4];
  auto rd = [&](int s, int h) __attribute__((always_inline)) {                                   // the 4 fragments of half-stage (s, h)
    const char* st = lds + (s & (NS - 1)) * ST + lane * 16;
#pragma unroll
    for (int t = 0; t < 4; ++t) fa[h][t] = RD ? *(const u32x4_t*)(st + (4 * h + t) * 1024) : u32x4_t{(unsigned)s, 1u, 2u, (unsigned)t};
  };
  // VAR 5 (shape B only): the dependency structure of the real kernel -- fc1 stages chain on acc[0 / 1], their GELU (gelu_poly2 on pairs read
  // from the accumulators, packed to bf16) produces the B fragments of the fc2 stages, pairs placed 2,1,1,2,1,1 behind stages 3-5 / 6-8
  u32x4_t hfr[2][2] = {{xb, xb}, {xb, xb}};
  auto mm5 = [&](int j, int h) __attribute__((always_inline)) {
    if constexpr (BIG) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (j < 6) acc[j / 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[h][t]), __builtin_bit_cast(bf16x8_t, xb), acc[j / 3], 0, 0, 0);
        else {
          const int u = (j - 6) / 3, th = (j - 6) % 3, a_ = 2 + th * 4 + 2 * h + (t >> 1);
          acc[a_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[h][t]), __builtin_bit_cast(bf16x8_t, hfr[u][t & 1]), acc[a_], 0, 0, 0);
        }
      }
      if (j >= 3 && j <= 8) {
        const int u = (j - 3) / 3, qq = 2 * ((j - 3) % 3) + h;
        const int first[7] = {0, 2, 3, 4, 6, 7, 8};
        for (int p = first[qq]; p < first[qq + 1]; ++p) {
          const f32x2_t g2 = gelu_poly2(f32x2_t{acc[u][2 * p], acc[u][2 * p + 1]});
          hfr[u][p >> 2][p & 3] = pack_bf2(g2[0], g2[1]);
          if (p == 7) for (int e = 0; e < 16; ++e) acc[u][e] = 0.25f;
        }
      }
    }
  };
  auto mm = [&](int j, int h) __attribute__((always_inline)) {       // j: stage inside the chunk (compile-time after unrolling)
    if constexpr (VAR == 5) { mm5(j, h); return; }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int a_ = VAR == 4 ? (j < 6 ? (j / 3) : 2 + ((j - 6) % 3) * 4 + t) % NACC : (j * 8 + 4 * h + t) % NACC;   // VAR 4: the 8 MFMAs of a GEMM1 stage chain on ONE accumulator
      if constexpr (BIG) acc[a_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[h][t]), __builtin_bit_cast(bf16x8_t, xb), acc[a_], 0, 0, 0);
      else acc[a_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[h][t]), __builtin_bit_cast(bf16x8_t, xb), acc[a_], 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < VF / 2; ++v) fill[v & 3] = __builtin_elementwise_fma(fill[v & 3], fill[(v + 1) & 3], f32x2_t{0.5f, 0.25f});
  };
  // group sync before the first read of stage s (s % GS == 0)
  auto sync = [&](int s) {
    if (DMA) wait_vm<GS*(NG - 3) * IPS>();
    __builtin_amdgcn_s_barrier();
    for (int i = 0; i < GS; ++i) issue(s + PD + i);
  };
  sync(0);
  rd(0, 0);
  // VAR 7 (shape A): fragments requested TWO half-stages ahead (three buffers)
  if constexpr (VAR == 7) {
    rd(0, 1);                                  // buffer 1 <- (stage 0, half 1); buffer 0 already holds (0, 0)
    for (int s0 = 0; s0 < nst; s0 += 12) {
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const int s = s0 + j;
        // half-stage index hs = 2 j + h uses buffer hs % 3; request hs + 2
        {
          const int hs = 2 * j;                 // (j, 0): request (j + 1, 0) into buffer (hs + 2) % 3
          if ((j + 1) % GS == 0) sync(s + 1);
          const char* st = lds + ((s + 1) & (NS - 1)) * ST + lane * 16;
#pragma unroll
          for (int t = 0; t < 4; ++t) fa[(hs + 2) % 3][t] = *(const u32x4_t*)(st + t * 1024);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int a_ = (j * 8 + t) % NACC;
            acc[a_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[hs % 3][t]), __builtin_bit_cast(bf16x8_t, xb), acc[a_], 0, 0, 0);
          }
#pragma unroll
          for (int v = 0; v < VF / 2; ++v) fill[v & 3] = __builtin_elementwise_fma(fill[v & 3], fill[(v + 1) & 3], f32x2_t{0.5f, 0.25f});
          __builtin_amdgcn_sched_barrier(0);
        }
        {
          const int hs = 2 * j + 1;             // (j, 1): request (j + 1, 1) into buffer (hs + 2) % 3
          const char* st = lds + ((s + 1) & (NS - 1)) * ST + lane * 16;
#pragma unroll
          for (int t = 0; t < 4; ++t) fa[(hs + 2) % 3][t] = *(const u32x4_t*)(st + (4 + t) * 1024);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int a_ = (j * 8 + 4 + t) % NACC;
            acc[a_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[hs % 3][t]), __builtin_bit_cast(bf16x8_t, xb), acc[a_], 0, 0, 0);
          }
#pragma unroll
          for (int v = 0; v < VF / 2; ++v) fill[v & 3] = __builtin_elementwise_fma(fill[v & 3], fill[(v + 1) & 3], f32x2_t{0.5f, 0.25f});
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  } else
  // VAR 0: as the kernel (a scheduling barrier after every half-stage).  1: no scheduling barriers at all.  2: s_setprio 1 around the MFMAs.
  // 3: barriers only at the group syncs.
  for (int s0 = 0; s0 < nst; s0 += 12) {
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int s = s0 + j;
      rd(s, 1);
      if (VAR == 2) __builtin_amdgcn_s_setprio(1);
      mm(j, 0);
      if (VAR == 2) __builtin_amdgcn_s_setprio(0);
      if (VAR == 0 || VAR == 2) __builtin_amdgcn_sched_barrier(0);
      if ((j + 1) % GS == 0) { if (VAR == 3) __builtin_amdgcn_sched_barrier(0); sync(s + 1); }
      rd(s + 1, 0);
      if (VAR == 2) __builtin_amdgcn_s_setprio(1);
      mm(j, 1);
      if (VAR == 2) __builtin_amdgcn_s_setprio(0);
      if (VAR == 0 || VAR == 2) __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = fill[0][0] + fill[1][1] + fill[2][0] + fill[3][1] + (float)hfr[0][0][0] + (float)hfr[1][1][3];
#pragma unroll
  for (int t = 0; t < NACC; ++t) r += acc[t][0] + acc[t][3];
  if (r == 123.456f) sink[0] = r;
}

template <bool BIG, int VF, bool DMA, bool RD, int VAR = 0>
static void run(const char* name, const char* w, size_t wb, int wgs) {
  float* sink; hipMalloc(&sink, 4);
  auto kern = probe<BIG, VF, DMA, RD, VAR>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, NS * ST);
  const int chunks = 24 * 20;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(BIG ? 256 : 512), NS * ST, 0, w, wb, chunks, sink);
    hipEventRecord(b); hipEventSynchronize(b);
  }
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us_chunk = ms * 1e3 / chunks, fl = 96.0 * (BIG ? 4 : 8) * (BIG ? 32768 : 16384);
  printf("%-58s %4d WGs: %6.3f us per chunk (24 chunks = %5.1f us) | %5.1f TF/s per CU-set, %4.0f %% of the bf16 MFMA peak per CU\n", name, wgs, us_chunk,
         24 * us_chunk, fl * wgs / (us_chunk * 1e-6) / 1e12, 100.0 * fl / (us_chunk * 1e-6) / (2.5e15 / 256));
  if (hipGetLastError() != hipSuccess) printf("   (launch error)\n");
  hipFree(sink);
}

int main() {
  const size_t wb = 4u << 20;          // (the probe walks the first half: 2 MiB, L2-resident like the 2.65 MB of a block's weights)
  char* w; hipMalloc(&w, wb); hipMemset(w, 0x11, wb);
  for (int wgs : {211}) {
    run<false, 12, true, true>("A 8 waves 16x16x32, GELU filler 12 pk/stage", w, wb, wgs);
    run<false, 12, true, true, 1>("A, no scheduling barriers", w, wb, wgs);
    run<false, 12, true, true, 2>("A, s_setprio 1 around the MFMAs", w, wb, wgs);
    run<false, 12, true, true, 3>("A, scheduling barriers at group syncs only", w, wb, wgs);
    run<false, 12, true, true, 6>("A, stage sources as 16 rows x 64 B pieces (pitch 768)", w, wb, wgs);
    run<false, 0, true, true, 6>("A, the same without filler", w, wb, wgs);
    run<false, 12, true, true, 7>("A, fragments two half-stages ahead", w, wb, wgs);
    run<false, 0, true, true>("A, no filler", w, wb, wgs);
    run<false, 12, false, true>("A, no DMA", w, wb, wgs);
    run<false, 0, false, false>("A, MFMA + barriers only", w, wb, wgs);
    run<true, 24, true, true>("B 4 waves 32x32x16, GELU filler 24 pk/stage", w, wb, wgs);
    run<true, 24, true, true, 4>("B, GEMM1 stages as dependent chains on one accumulator", w, wb, wgs);
    run<true, 0, true, true, 5>("B, real dependency structure + gelu_poly2 pairs", w, wb, wgs);
    run<true, 0, true, true>("B, no filler", w, wb, wgs);
    run<true, 24, false, true>("B, no DMA", w, wb, wgs);
    run<true, 0, false, false>("B, MFMA + barriers only", w, wb, wgs);
  }
  return 0;
}
