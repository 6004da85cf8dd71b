// bf16 MFMA GEMM, "NT" form:  C[M,N] (+)= A[M,K] . B[N,K]^T   (both operands K-contiguous)
//
// Replaces the ATen addmm / matmul call sites of the reference backbone:
//   qkv / proj Linear  semilearn/nets/vit/vit.py:93-98,105      (K3, K5 in SURVEY.md 2c)
//   fc1 / GELU / fc2   semilearn/nets/vit/vit.py:69-75          (K6)
// and their autograd backward (dX = dY.W, dW = dY^T.X) -- the host side supplies W^T copies and
// transposed activations so that every product is an NT product.
//
// CDNA4 design: 128x128x64 tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 bf16 tiles,
// fp32 accumulate.  The MFMA a-operand is fed from the B matrix (rows n) and the b-operand from
// the A matrix (rows m), so each lane ends up with 4 CONSECUTIVE n for one m: the epilogue
// stores 8 B (bf16) / 16 B (fp32) contiguous per lane into the row-major C.
// LDS: double-buffered [128][64] bf16 tiles, 16-B slots XOR-swizzled by (row & 7) so that the
// ds_read_b128 fragment reads of 16 different rows spread over the 64 banks.
// Global->LDS is register-staged (global_load_dwordx4 issued before the MFMA block of the
// current tile, ds_write_b128 after it): one barrier per K-step.
// Block->tile mapping is XCD-aware (consecutive N-tiles of one A row-panel share an XCD L2).
#include "common.h"
#include "srhip.h"

namespace {

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const float* bias;
  const float* row_scale;
  const bf16_t* aux_in;
  bf16_t* aux_out;
  int M, N, K, lda, ldb, ldc, ldaux, rows_per_sample;
  float alpha, beta;
};

constexpr int BM = 128, BN = 128, BK = 64;

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) bf16_t smem[2][2][BM * BK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lg = lane >> 4;
  const int ntn = (g.N + BN - 1) / BN;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (wg / ntn) * BM, n0 = (wg % ntn) * BN;

  u32x4_t ra[4], rb[4];
  size_t goa[4], gob[4];
  int soff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + i * 256, row = c >> 3, slot = c & 7;
    const int gm = min(m0 + row, g.M - 1), gn = min(n0 + row, g.N - 1);
    goa[i] = (size_t)gm * g.lda + slot * 8;
    gob[i] = (size_t)gn * g.ldb + slot * 8;
    soff[i] = row * BK + ((slot ^ (row & 7)) << 3);
  }
#define GLOAD(k0)                                                          \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                          \
    ra[i] = *reinterpret_cast<const u32x4_t*>(g.A + goa[i] + (k0));          \
    rb[i] = *reinterpret_cast<const u32x4_t*>(g.B + gob[i] + (k0));          \
  }
#define SSTORE(buf)                                                        \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                          \
    *reinterpret_cast<u32x4_t*>(&smem[buf][0][soff[i]]) = ra[i];             \
    *reinterpret_cast<u32x4_t*>(&smem[buf][1][soff[i]]) = rb[i];             \
  }

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK;
  GLOAD(0)
  SSTORE(0)
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) { GLOAD((kt + 1) * BK) }
    const bf16_t* As = smem[buf][0];
    const bf16_t* Bs = smem[buf][1];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      s16x8_t fa[4], fb[4];
      const int slot = kk * 4 + lg;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int rn = wn * 64 + t * 16 + l15;
        fa[t] = *reinterpret_cast<const s16x8_t*>(&Bs[rn * BK + ((slot ^ (rn & 7)) << 3)]);
        const int rm = wm * 64 + t * 16 + l15;
        fb[t] = *reinterpret_cast<const s16x8_t*>(&As[rm * BK + ((slot ^ (rm & 7)) << 3)]);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(bf16x8_t, fa[nt]), __builtin_bit_cast(bf16x8_t, fb[mt]), acc[nt][mt], 0, 0, 0);
    }
    if (kt + 1 < nk) { SSTORE(buf ^ 1) }
    __syncthreads();
  }

#undef GLOAD
#undef SSTORE
  // ---- epilogue: lane holds C[m][n .. n+3], m = tile row (lane&15), n = 4*(lane>>4) + r
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + wm * 64 + mt * 16 + l15;
    if (m >= g.M) continue;
    float rs = 1.0f;
    if (EPI == SRHIP_EPI_RESID_F32 && g.row_scale) rs = g.row_scale[m / g.rows_per_sample];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + wn * 64 + nt * 16 + lg * 4;
      if (n >= g.N) continue;
      float v[4] = {acc[nt][mt][0], acc[nt][mt][1], acc[nt][mt][2], acc[nt][mt][3]};
      if (g.bias) {
        const float4 b4 = *reinterpret_cast<const float4*>(g.bias + n);
        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
      }
      const size_t off = (size_t)m * g.ldc + n;
      if (EPI == SRHIP_EPI_BF16) {
        uint2 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(g.C) + off) = o;
      } else if (EPI == SRHIP_EPI_GELU_BF16) {
        if (g.aux_out) {
          uint2 p = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
          *reinterpret_cast<uint2*>(g.aux_out + (size_t)m * g.ldaux + n) = p;
        }
        uint2 o = {pack_bf2(gelu_erf(v[0]), gelu_erf(v[1])), pack_bf2(gelu_erf(v[2]), gelu_erf(v[3]))};
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(g.C) + off) = o;
      } else if (EPI == SRHIP_EPI_RESID_F32) {
        // residual source: C itself (in place) or, when the pre-block stream is kept for the backward,
        // a separate fp32 buffer passed as aux_in (leading dimension ldaux)
        float4* cp = reinterpret_cast<float4*>(reinterpret_cast<float*>(g.C) + off);
        float4 x = g.aux_in ? *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(g.aux_in) + (size_t)m * g.ldaux + n)
                            : *cp;
        x.x += rs * v[0]; x.y += rs * v[1]; x.z += rs * v[2]; x.w += rs * v[3];
        *cp = x;
      } else if (EPI == SRHIP_EPI_DGELU_BF16) {
        const uint2 p = *reinterpret_cast<const uint2*>(g.aux_in + (size_t)m * g.ldaux + n);
        const float p0 = bf2f((bf16_t)(p.x & 0xffff)), p1 = bf2f((bf16_t)(p.x >> 16));
        const float p2 = bf2f((bf16_t)(p.y & 0xffff)), p3 = bf2f((bf16_t)(p.y >> 16));
        uint2 o = {pack_bf2(v[0] * gelu_erf_grad(p0), v[1] * gelu_erf_grad(p1)),
                   pack_bf2(v[2] * gelu_erf_grad(p2), v[3] * gelu_erf_grad(p3))};
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(g.C) + off) = o;
      } else {  // SRHIP_EPI_F32: C = alpha*acc + beta*C
        float4* cp = reinterpret_cast<float4*>(reinterpret_cast<float*>(g.C) + off);
        float4 x = {g.alpha * v[0], g.alpha * v[1], g.alpha * v[2], g.alpha * v[3]};
        if (g.beta != 0.0f) {
          const float4 c = *cp;
          x.x += g.beta * c.x; x.y += g.beta * c.y; x.z += g.beta * c.z; x.w += g.beta * c.w;
        }
        *cp = x;
      }
    }
  }
}

}  // namespace

extern "C" int srhip_gemm_nt(int epilogue, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                             int M, int N, int K, const float* bias, const float* row_scale, int rows_per_sample,
                             const void* aux_in, void* aux_out, int ldaux, float alpha, float beta, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (K % BK) || (N % 4) || (lda % 8) || (ldb % 8) || (ldc % 4)) return SR_EINVAL;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return SR_EINVAL;
  if (epilogue == SRHIP_EPI_DGELU_BF16 && !aux_in) return SR_EINVAL;
  if (row_scale && rows_per_sample <= 0) return SR_EINVAL;
  GemmArgs g;
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = C; g.bias = bias; g.row_scale = row_scale;
  g.aux_in = (const bf16_t*)aux_in; g.aux_out = (bf16_t*)aux_out;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = ldaux;
  g.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1; g.alpha = alpha; g.beta = beta;
  const int grid = cdiv(M, BM) * cdiv(N, BN);
  hipStream_t s = (hipStream_t)stream;
  switch (epilogue) {
    case SRHIP_EPI_BF16: hipLaunchKernelGGL(gemm_nt_kernel<SRHIP_EPI_BF16>, dim3(grid), dim3(256), 0, s, g); break;
    case SRHIP_EPI_GELU_BF16: hipLaunchKernelGGL(gemm_nt_kernel<SRHIP_EPI_GELU_BF16>, dim3(grid), dim3(256), 0, s, g); break;
    case SRHIP_EPI_RESID_F32: hipLaunchKernelGGL(gemm_nt_kernel<SRHIP_EPI_RESID_F32>, dim3(grid), dim3(256), 0, s, g); break;
    case SRHIP_EPI_DGELU_BF16: hipLaunchKernelGGL(gemm_nt_kernel<SRHIP_EPI_DGELU_BF16>, dim3(grid), dim3(256), 0, s, g); break;
    case SRHIP_EPI_F32: hipLaunchKernelGGL(gemm_nt_kernel<SRHIP_EPI_F32>, dim3(grid), dim3(256), 0, s, g); break;
    default: return SR_EINVAL;
  }
  SR_CHECK_LAUNCH();
  return SR_OK;
}
