// LayerNorm fused into the Linear that consumes it, for the rows WITH a backward (gfx950, embed_dim 384):
//     ln  = LayerNorm(x) (bf16, kept for the backward together with mean / rstd)
//     out = ln W^T + b                       (qkv: vit.py:93-98 on norm1, vit.py:163)
//     out = GELU(ln W^T + b), pre = ln W^T + b   (fc1 + act: vit.py:69-72 on norm2, vit.py:165)
// reference: semilearn/nets/vit/vit.py:163,165 (Block.forward: norm1 -> attn, norm2 -> mlp) with the Linear of :93 / :69.
//
// Why: the 16 gradient-carrying images of a SemiReward step (4 112 rows) run their forward as a chain of small launches beside the inference
// rows' launch train -- 7 per block, and what the losses wait for (profiles/r06_phases.txt: the chain ends 0.2 ms after the read train).  Each
// launch of the chain costs its latency, not its work; srhip_layernorm_fwd + srhip_gemm_nt are two of them, twice per block.  Here the workgroup
// normalises its rows in registers, straight into MFMA fragments (the prologue of mlp_fused.hip), and streams the weight block through the
// LDS-DMA ring of mlp_fused.hip's projection phase: 5 launches per block instead of 7.
//
// Structure: one workgroup = 128 rows x 384 output features (8 waves x 16 rows, 96 accumulator registers), grid (row tiles, N / 384).  The
// weight block [384, 384] streams as 36 stages of [128 features x 32 k] (8 KiB) through a 16-slot ring, one LDS-DMA instruction per wave and
// stage, issued one stage at a time behind the stage's first MFMA half, group barrier every 4 stages (counted vmcnt).  The workgroups of
// column block 0 write the normalised rows and their statistics.
// Rounding points are the ones of the two launches it replaces (ln bf16, fp32 accumulation in k order, bias in fp32, outputs bf16).
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "../../include/srhip.h"
#include "common.h"

namespace {

constexpr int D_ = 384;                   // embed dim = K of the product
constexpr int BK = 32;
constexpr int RT = 128;                   // weight rows per ring tile
constexpr int TILE_EL = RT * BK;          // 8 KiB
constexpr int NS = 16, GS = 4, NG = NS / GS, PD = (NG - 2) * GS;
constexpr int FBM = 128;                  // rows per workgroup
constexpr int NB = 384;                   // output features per workgroup
constexpr int KS = D_ / BK;               // 12 k-steps
constexpr int NV = NB / RT;               // 3 thirds
constexpr int NSTG = NV * KS;             // 36 stages
constexpr int NT = NB / 16;               // 24 accumulator tiles

__device__ __forceinline__ int swz(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }   // as in gemm.hip / mlp_fused.hip
typedef __attribute__((address_space(3))) void lds_void;
template <int N_>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

struct LgArgs {
  const float* x;                         // [M, 384] residual stream (LayerNorm source)
  const float *gamma, *beta, *bias;
  const bf16_t* W;                        // [N, 384] bf16
  bf16_t* out;                            // [M, N]
  bf16_t* aux;                            // GELU: pre-activation [M, N], or NULL
  bf16_t* ln;                             // [M, 384] normalised rows, or NULL
  float *mean, *rstd;                     // [M], or NULL
  float eps;
  int M, N;
};

template <int EPI>                        // 0: out = acc + b; 1: out = GELU(acc + b), aux = acc + b
__global__ __launch_bounds__(512, 2) void ln_gemm_kernel(LgArgs a) {
  extern __shared__ __attribute__((aligned(16))) bf16_t sm[];
  float* sgam = reinterpret_cast<float*>(sm + NS * TILE_EL);
  float* sbet = sgam + D_;
  float* sbias = sbet + D_;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;
  const int nb = blockIdx.y;
  for (int i = tid; i < D_; i += 512) { sgam[i] = a.gamma[i]; sbet[i] = a.beta[i]; sbias[i] = a.bias ? a.bias[nb * NB + i] : 0.f; }

  // ---- producer (mlp_fused.hip, projection phase): stage s = 12 v + j is the LDS tile  r -> W[384 nb + 128 v + r][32 j ..]
  const int pr = 16 * wave + (lane >> 2);
  const int psl = ((lane & 3) ^ swz(pr)) * 8;
  const int lop = (pr * D_ + psl) * 2;
  const int pdst = 16 * wave * BK;
  const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.W + (size_t)nb * NB * D_), 0, NB * D_ * 2, 0x00020000);
  constexpr int OOB = 0x7ffffff0;
  auto issue = [&](auto sc) __attribute__((always_inline)) {       // stages past the end: out-of-range no-ops that still count in vmcnt
    constexpr int s = decltype(sc)::value, v = s / KS, j = s % KS;
    lds_void* dst = (lds_void*)(sm + (s & (NS - 1)) * TILE_EL + pdst);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, dst, 16, s < NSTG ? lop : OOB, s < NSTG ? (v * RT * D_ + j * BK) * 2 : 0, 0, 0);
  };
  // the first stages travel while the rows are normalised (the ring is not read before the first group sync)
  static_for<PD>([&](auto sc) __attribute__((always_inline)) { issue(sc); });

  const int m = blockIdx.x * FBM + wave * 16 + l15;
  const int mc = min(m, a.M - 1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // sgam / sbet / sbias are written; the LDS-DMA loads in flight are not drained
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // ---- LayerNorm of the wave's 16 rows, straight into MFMA B-fragments: lane holds x[m][32 k + 8 lg .. + 7], k = 0..11
  u32x4_t xn[KS];
  {
    const float* xr = a.x + (size_t)mc * D_ + 8 * lg;
    f32x4_t v[2 * KS];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      v[2 * k] = *reinterpret_cast<const f32x4_t*>(xr + 32 * k);
      v[2 * k + 1] = *reinterpret_cast<const f32x4_t*>(xr + 32 * k + 4);
    }
#pragma unroll
    for (int k = 0; k < 2 * KS; ++k) s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mu = s * (1.0f / D_);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 2 * KS; ++k) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[k][e] - mu; q += d * d; }
    }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    const float rs = rsqrtf(q * (1.0f / D_) + a.eps);
    if (nb == 0 && a.mean && m < a.M && lg == 0) { a.mean[m] = mu; a.rstd[m] = rs; }
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      const f32x4_t g0 = *reinterpret_cast<const f32x4_t*>(sgam + 32 * k + 8 * lg);
      const f32x4_t g1 = *reinterpret_cast<const f32x4_t*>(sgam + 32 * k + 8 * lg + 4);
      const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(sbet + 32 * k + 8 * lg);
      const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(sbet + 32 * k + 8 * lg + 4);
      const f32x4_t p = v[2 * k], r = v[2 * k + 1];
      xn[k] = u32x4_t{pack_bf2((p[0] - mu) * rs * g0[0] + b0[0], (p[1] - mu) * rs * g0[1] + b0[1]),
                      pack_bf2((p[2] - mu) * rs * g0[2] + b0[2], (p[3] - mu) * rs * g0[3] + b0[3]),
                      pack_bf2((r[0] - mu) * rs * g1[0] + b1[0], (r[1] - mu) * rs * g1[1] + b1[1]),
                      pack_bf2((r[2] - mu) * rs * g1[2] + b1[2], (r[3] - mu) * rs * g1[3] + b1[3])};
      if (k & 1) __builtin_amdgcn_sched_barrier(0);     // keeps the scheduler from hoisting all 48 affine reads (spills)
    }
  }
  if (nb == 0 && a.ln && m < a.M) {
    bf16_t* lr = a.ln + (size_t)m * D_ + 8 * lg;
#pragma unroll
    for (int k = 0; k < KS; ++k) *reinterpret_cast<u32x4_t*>(lr + 32 * k) = xn[k];
  }
  __builtin_amdgcn_sched_barrier(0);       // the accumulators must not become live across the LayerNorm above
  f32x4_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = *reinterpret_cast<const f32x4_t*>(sbias + 16 * t + 4 * lg);      // start value = bias

  // ---- consumer: row l15 of 16-row tile t of a ring tile: + t * 16 * BK
  const int fo2 = l15 * BK + ((lg ^ swz(l15)) << 3);
  u32x4_t fa0[4], fa1[4];
  auto read_half = [&](auto sc, auto hc) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value, h = decltype(hc)::value;
    u32x4_t(&fa)[4] = h ? fa1 : fa0;
    const bf16_t* st = sm + (s & (NS - 1)) * TILE_EL;
#pragma unroll
    for (int t = 0; t < 4; ++t) fa[t] = *reinterpret_cast<const u32x4_t*>(st + fo2 + (4 * h + t) * 16 * BK);
  };
  // group sync before the first read of the group that starts at stage s: own DMA parts of the group have landed (the NG - 3 younger groups
  // may be in flight: with one stage asked for per stage multiplied, stages up to s - 1 + PD are issued), barrier (everybody's parts have)
  auto sync_group = [&]() __attribute__((always_inline)) {
    wait_vm<GS*(NG - 3)>();
    __builtin_amdgcn_s_barrier();
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;
  sync_group();
  read_half(std::integral_constant<int, 0>{}, H0{});
  static_for<NSTG>([&](auto sc) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value, v = s / KS, j = s % KS;
    read_half(sc, H1{});
#pragma unroll
    for (int t = 0; t < 4; ++t)
      acc[v * 8 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa0[t]), __builtin_bit_cast(bf16x8_t, xn[j]), acc[v * 8 + t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    issue(std::integral_constant<int, s + PD>{});          // its slot held stage s + PD - NS, consumed before the last group barrier
    if constexpr (s + 1 < NSTG) {
      if constexpr ((s + 1) % GS == 0) sync_group();
      read_half(std::integral_constant<int, s + 1>{}, H0{});
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
      acc[v * 8 + 4 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa1[t]), __builtin_bit_cast(bf16x8_t, xn[j]), acc[v * 8 + 4 + t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  });

  // ---- epilogue: lane holds y[m][384 nb + 16 t + 4 lg + r]
  if (m < a.M) {
    const size_t ro = (size_t)m * a.N + (size_t)nb * NB + 4 * lg;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const f32x4_t v = acc[t];
      if constexpr (EPI == 1) {
        if (a.aux) *reinterpret_cast<u32x2_t*>(a.aux + ro + 16 * t) = u32x2_t{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        *reinterpret_cast<u32x2_t*>(a.out + ro + 16 * t) = u32x2_t{pack_bf2(gelu_erf(v[0]), gelu_erf(v[1])), pack_bf2(gelu_erf(v[2]), gelu_erf(v[3]))};
      } else {
        *reinterpret_cast<u32x2_t*>(a.out + ro + 16 * t) = u32x2_t{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
      }
    }
  }
}

}  // namespace

extern "C" int srhip_ln_gemm_supported(int D, int N) { return (D == D_ && N > 0 && N % NB == 0) ? 1 : 0; }

// ln = LayerNorm(x; gamma, beta, eps) (bf16 [M, 384], mean / rstd [M]; any of the three outputs may be NULL),
// out = ln W^T + bias (epilogue SRHIP_EPI_BF16) or out = GELU(ln W^T + bias), aux = ln W^T + bias (SRHIP_EPI_GELU_BF16; aux may be NULL).
extern "C" int srhip_ln_gemm(int epilogue, const float* x, const float* gamma, const float* beta, float eps, const void* W, const float* bias,
                             void* out, void* aux_out, void* ln_out, float* mean, float* rstd, int M, int N, int D, void* stream) {
  if (!x || !gamma || !beta || !W || !out || M <= 0) return SR_EINVAL;
  if (!srhip_ln_gemm_supported(D, N)) return SR_EINVAL;
  if (epilogue != SRHIP_EPI_BF16 && epilogue != SRHIP_EPI_GELU_BF16) return SR_EINVAL;
  if ((mean == nullptr) != (rstd == nullptr)) return SR_EINVAL;
  if (((uintptr_t)x | (uintptr_t)W | (uintptr_t)out | (uintptr_t)aux_out | (uintptr_t)ln_out | (uintptr_t)gamma | (uintptr_t)beta) & 15) return SR_EINVAL;
  LgArgs a;
  a.x = x; a.gamma = gamma; a.beta = beta; a.bias = bias; a.W = (const bf16_t*)W; a.out = (bf16_t*)out; a.aux = (bf16_t*)aux_out;
  a.ln = (bf16_t*)ln_out; a.mean = mean; a.rstd = rstd; a.eps = eps; a.M = M; a.N = N;
  const size_t smem = (size_t)NS * TILE_EL * sizeof(bf16_t) + (size_t)(3 * D_) * sizeof(float);
  void (*kern)(LgArgs) = epilogue == SRHIP_EPI_GELU_BF16 ? ln_gemm_kernel<1> : ln_gemm_kernel<0>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  SR_LAUNCH(kern, dim3(cdiv(M, FBM), N / NB), dim3(512), smem, (hipStream_t)stream, a);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
