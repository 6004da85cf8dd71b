// Grouped weight-gradient products straight from ROW-MAJOR operands (gfx950):
//     dW_p[M, N] = beta * dW_p + alpha * A_p^T . B_p,   A_p = dY [K tokens, M], B_p = X [K tokens, N]  (bf16, fp32 accumulate)
//     dbias_p[M] += column sums of A_p                                             (optional)
// = the autograd backward of every nn.Linear on the path (vit.py:69-75 Mlp, :95-112 Attention qkv/proj) for all layers in ONE
// launch.  The reduction index (tokens) is the ROW index of both operands, so an NT kernel needs both of them transposed
// first (8 transposes per block, 0.7 ms per step at the reference batch).  Here the tiles are staged as they lie in memory,
// [32 tokens x 128 features], and the MFMA fragments are gathered with ds_read_b64_tr_b16 (CDNA4 LDS transpose read).
//
// Fragment gather.  A 16-lane group passing 16 addresses of 4 contiguous bf16 receives, in lane i, element (i & 3) of the
// chunks of lanes i>>2 + 4j (j = 0..3): with lane (l15) pointing at row k0 + (l15 >> 2), columns c0 + 4 (l15 & 3) lane l15
// gets the 4 keys k0..k0+3 of column c0 + l15 (probed on hardware, tools/tr_probe.hip).  Two such reads (k0 = 4g and 16 + 4g
// for lane group g) fill the 8 k-slots of a 16x16x32 operand; both operands use the same slot -> k map, which is all the
// MFMA needs.
// Bank conflicts.  The 32 lanes of a half-wave read 8 rows x 32 B; rows are 256 B apart = the same banks.  The image is
// therefore rotated by 32 B per row (16-B chunk p of row r holds source chunk (p - 2r) mod 16, applied on the LDS-DMA source
// address): rows 4g..4g+3 of groups g = 0, 1 land on 8 distinct 8-bank segments.
#include <stdlib.h>

#include "../../include/srhip.h"
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int TILE = BK * BM;            // 4096 bf16 = 8 KiB: [32 tokens][128 features]
constexpr int STAGE = 2 * TILE;          // A tile then B tile
constexpr int NS = 3, PD = NS - 1;

typedef __attribute__((address_space(3))) void lds_void;
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;
// the builtin (not inline asm): the compiler then counts the read in lgkmcnt and allocates the destination pair itself
__device__ __forceinline__ u32x2_t ds_read_tr16(const bf16_t* p) {
  return __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p));
}

__global__ __launch_bounds__(256, 3) void gemm_tn_grouped_f32_kernel(const srhip_group_tn_desc* __restrict__ desc, int n_problems,
                                                                     float alpha, float beta) {
  __shared__ __attribute__((aligned(16))) bf16_t smem[NS * STAGE];
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  int p = 0;
  while (p + 1 < n_problems && tile >= desc[p + 1].tile_start) ++p;
  const srhip_group_tn_desc d = desc[p];
  const int local = tile - d.tile_start, ntn = (d.N + BN - 1) / BN;
  const int m0 = (local / ntn) * BM, n0 = (local % ntn) * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lg = lane >> 4;
  const int nk = (d.K + BK - 1) / BK;

  // ---- LDS-DMA: a wave instruction fills 4 rows x 256 B; lane l -> row (l >> 4), physical chunk p = l & 15 <- source chunk
  // (p - 2 row) & 15.  Rows >= K and bytes past the operand are buffer-out-of-range: they read as zero, which is exactly the
  // padding the reduction needs.  Wave w stages rows 8w .. 8w+7 of both tiles (2 + 2 instructions per stage).
  // (ld < row width = overlapping rows, the unfolded operand of a strided Conv1d read in place: the extent is then (K-1)*ld + width; the
  // rows >= K of such an operand are real memory, finite by the caller's contract, and meet the zero rows of the other operand)
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(d.A), 0, max(d.K * d.lda, (d.K - 1) * d.lda + d.M) * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(d.B), 0, max(d.K * d.ldb, (d.K - 1) * d.ldb + d.N) * 2, 0x00020000);
  int va[2], vb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 8 * wave + 4 * i + (lane >> 4);
    const int ch = ((lane & 15) - 2 * row) & 15;
    va[i] = (row * d.lda + 8 * ch) * 2;
    vb[i] = (row * d.ldb + 8 * ch) * 2;
  }
  const int sa0 = m0 * 2, sb0 = n0 * 2, sak = BK * d.lda * 2, sbk = BK * d.ldb * 2;
#define ISSUE(kt_, st_)                                                                                                  \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                        \
    bf16_t* da = smem + (st_) * STAGE + (8 * wave + 4 * i) * BM;                                                         \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)da, 16, va[i], sa0 + (kt_) * sak, 0, 0);                      \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(da + TILE), 16, vb[i], sb0 + (kt_) * sbk, 0, 0);             \
  }

  // ---- fragment offsets (elements): row r0 = 4 g + (l15 >> 2) (second read: + 16 rows = + 16 * 128 elements, same rotation
  // because 2 * 16 = 0 mod 16); column c = 64 w + 16 t + 4 (l15 & 3): chunk q = c >> 3, half = (c >> 2) & 1
  const int r0 = 4 * lg + (l15 >> 2);
  int fo_a[4], fo_b[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int cn = wn * 64 + t * 16 + 4 * (l15 & 3), cm = wm * 64 + t * 16 + 4 * (l15 & 3);
    fo_a[t] = TILE + r0 * BM + ((((cn >> 3) + 2 * r0) & 15) << 3) + (cn & 4);    // MFMA a-operand <- B matrix columns (n)
    fo_b[t] = r0 * BM + ((((cm >> 3) + 2 * r0) & 15) << 3) + (cm & 4);           // MFMA b-operand <- A matrix columns (m)
  }
  f32x4_t acc[4][4], accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    accb[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  const bool want_bias = d.dbias != nullptr && n0 == 0 && wn == 0;      // one column of workgroups, the waves that own m
  const u32x4_t ones = u32x4_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};

#pragma unroll
  for (int q = 0; q < PD; ++q)
    if (q < nk) { ISSUE(q, q) }
  for (int kt = 0; kt < nk; ++kt) {
    const int rem = nk - 1 - kt;
    if (PD >= 3 && rem >= 2) WAIT_VM(8); else if (PD >= 2 && rem >= 1) WAIT_VM(4); else WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    if (kt + PD < nk) { ISSUE(kt + PD, (kt + PD) % NS) }
    const bf16_t* st = smem + (kt % NS) * STAGE;
    u32x4_t fa[4], fb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const u32x2_t a0 = ds_read_tr16(st + fo_a[t]), a1 = ds_read_tr16(st + fo_a[t] + 16 * BM);
      const u32x2_t b0 = ds_read_tr16(st + fo_b[t]), b1 = ds_read_tr16(st + fo_b[t] + 16 * BM);
      fa[t] = u32x4_t{a0[0], a0[1], a1[0], a1[1]};
      fb[t] = u32x4_t{b0[0], b0[1], b1[0], b1[1]};
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[nt]), __builtin_bit_cast(bf16x8_t, fb[mt]),
                                                              acc[nt][mt], 0, 0, 0);
    if (want_bias) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        accb[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ones), __builtin_bit_cast(bf16x8_t, fb[mt]),
                                                           accb[mt], 0, 0, 0);
    }
  }
#undef ISSUE

  const bool atomic = (d.flags & SRHIP_TN_ATOMIC) != 0;
  // ---- epilogue: lane holds C[m][n .. n+3], m = tile row (lane & 15), n = 4 (lane >> 4) + r
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + wm * 64 + mt * 16 + l15;
    if (m >= d.M) continue;
    if (want_bias && lg == 0) {                                          // every row of the ones-product equals the column sum
      if (atomic) unsafeAtomicAdd(d.dbias + m, accb[mt][0]); else d.dbias[m] += accb[mt][0];
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + wn * 64 + nt * 16 + lg * 4;
      if (n >= d.N) continue;
      f32x4_t* cp = reinterpret_cast<f32x4_t*>(d.C + (size_t)m * d.ldc + n);
      f32x4_t x = {alpha * acc[nt][mt][0], alpha * acc[nt][mt][1], alpha * acc[nt][mt][2], alpha * acc[nt][mt][3]};
      if (atomic) {                     // one K slice of a split problem: the slices meet in C through the hardware fp32 atomic add
        float* cf = d.C + (size_t)m * d.ldc + n;
        unsafeAtomicAdd(cf, x[0]); unsafeAtomicAdd(cf + 1, x[1]); unsafeAtomicAdd(cf + 2, x[2]); unsafeAtomicAdd(cf + 3, x[3]);
        continue;
      }
      if (beta != 0.0f) {
        const f32x4_t c = *cp;
        x[0] += beta * c[0]; x[1] += beta * c[1]; x[2] += beta * c[2]; x[3] += beta * c[3];
      }
      *cp = x;
    }
  }
}

}  // namespace

extern "C" int srhip_gemm_tn_grouped_f32(const srhip_group_tn_desc* desc_dev, int n_problems, int total_tiles, float alpha,
                                         float beta, void* stream) {
  if (!desc_dev || n_problems <= 0 || n_problems > 4096 || total_tiles <= 0) return SR_EINVAL;
  SR_LAUNCH(gemm_tn_grouped_f32_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, desc_dev, n_problems, alpha,
                     beta);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
