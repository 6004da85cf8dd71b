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

#include <type_traits>
#include <utility>

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
  {  // binary search: the last entry whose tile_start <= tile (a token-sliced table has hundreds of entries; a linear walk of dependent scalar
     // loads cost a workgroup as much as its product)
    int lo = 0, hi = n_problems - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tile >= desc[mid].tile_start) lo = mid; else hi = mid - 1; }
    p = lo;
  }
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
  const bool overwrite = (d.flags & SRHIP_TN_OVERWRITE) != 0;        // a token slice with a slab of its own: C = alpha * acc, dbias = sums
  // ---- epilogue: lane holds C[m][n .. n+3], m = tile row (lane & 15), n = 4 (lane >> 4) + r
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + wm * 64 + mt * 16 + l15;
    if (m >= d.M) continue;
    if (want_bias && lg == 0) {                                          // every row of the ones-product equals the column sum
      if (atomic) unsafeAtomicAdd(d.dbias + m, accb[mt][0]); else if (overwrite) d.dbias[m] = accb[mt][0]; else d.dbias[m] += accb[mt][0];
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
      if (beta != 0.0f && !overwrite) {
        const f32x4_t c = *cp;
        x[0] += beta * c[0]; x[1] += beta * c[1]; x[2] += beta * c[2]; x[3] += beta * c[3];
      }
      *cp = x;
    }
  }
}


// ================================================================================================================================
// 256 x 256 output tiles, persistent, two wave groups half a phase apart (the schedule of gemm.hip's gemm_pp_kernel) for the weight gradients
// of the wide layers (D = 768: BERT / Wav2Vec2 / HuBERT, tokens K = 8192 per step; also ViT-S where 256-tiles cover enough of the problem).
// The 128 x 128 kernel above streams 128 bytes per 8192 multiply-adds of a K step -- 64 flop per byte filled into LDS, against the ~25 B/clk a CU
// fills at: 0.39 of the matrix peak at best, 0.14-0.23 measured (306 us per ViT-S step, 336 us per BERT layer).  A 256 x 256 tile has twice the
// flop per filled byte and one workgroup per CU keeps 128 x 64 accumulators per wave.
//   * LDS ring of NSLOT 16-KiB half-tiles [64 tokens][128 features] in the order A0, B0, B1, A1 per 64-token K-tile: A0 / A1 = the features
//     every wave multiplies in its upper / lower quadrant (rows 64 h .. of both 128-row halves of the tile), B0 / B1 likewise 32 h .. of every
//     64-column strip.  A half-tile is staged as it lies in memory (token-major), rotated by two 16-byte chunks per token row as above, and the
//     fragments are gathered with ds_read_b64_tr_b16.
//   * waves 2 (m) x 4 (n); the row groups wr = 0 / 1 alternate between "read fragments + issue the refill" and "16 MFMAs"; phases per K-tile:
//       p0: read B0 (4 fragments) + A0 (8)  q(0,0)     p1: read B1 (4)  q(0,1)     p2: read A1 (8)  q(1,1)     p3: --  q(1,0)
//     wait / slot arithmetic exactly as in gemm_pp_kernel (phase k issues half-tile k + NSLOT - 2 and waits until <= NSLOT - 4 are in flight).
//   * the walk is persistent over the tiles of ALL entries of the table (an entry = one Linear's dW, or one token slice of it flagged
//     SRHIP_TN_ATOMIC); the refill cursor runs across tile and entry boundaries.
//   * dbias: wave (wr, wc) sums row tile wc of each half -- two extra MFMAs per half and K-tile against a ones operand, on the tiles of column 0
//     only; the wave's row tiles are taken in the order (mt + wc) & 3 so that "its" tile is always fragment set 0 (no register indexing).
constexpr int PBK = 64;                   // tokens per K-tile
constexpr int PH_EL = PBK * 128;          // one half-tile: [64 tokens][128 features]
constexpr int PSLOT = 10;                 // 160 KiB: the whole LDS of a CU
constexpr int PLEAD = PSLOT - 2;          // phase k issues half-tile k + PLEAD
constexpr int PINFL = 2 * (PLEAD - 2);    // LDS-DMA instructions that may stay in flight behind a phase's wait
constexpr int PGROUP_M = 8;

template <int N_>
__device__ __forceinline__ void wait_vm_n() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

// tile index -> (row tile, column tile) in bands of PGROUP_M row tiles walked column-major (as gemm.hip tile_mn)
__device__ __forceinline__ void pp_tile_mn(int t, int ntm, int ntn, int& tm, int& tn) {
  const int band = t / (PGROUP_M * ntn), r = t - band * PGROUP_M * ntn;
  const int rows = min(PGROUP_M, ntm - band * PGROUP_M);
  tm = band * PGROUP_M + r % rows;
  tn = r / rows;
}

__global__ __launch_bounds__(512, 1) void gemm_tn_pp_kernel(const srhip_group_tn_desc* __restrict__ desc, int n_problems, int total_tiles,
                                                            float alpha, float beta) {
  extern __shared__ __attribute__((aligned(16))) bf16_t psm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l15 = lane & 15, lg = lane >> 4;
  const int first = xcd_remap(blockIdx.x, gridDim.x);
  if (first >= total_tiles) return;
  const int my_tiles = (total_tiles - first + gridDim.x - 1) / gridDim.x;

  // ---- producer: a wave instruction moves one 1-KiB piece = 4 token rows x 256 B; every wave owns pieces wave and wave + 8 (rows + 32) of each
  // half-tile.  Lane -> row 4 wave + (lane >> 4), physical chunk lane & 15 <- logical chunk (p - 2 row) & 15 of the half-tile's 128 features.
  const int prow = 4 * wave + (lane >> 4);
  const int pls = ((lane & 15) - 2 * prow) & 15;
  const int pfa = 8 * pls + (pls >= 8 ? 64 : 0);                // A half h: feature m0 + 64 h + pfa
  const int pfb = 64 * (pls >> 2) + 8 * (pls & 3);              // B half h: feature n0 + 32 h + pfb
  constexpr int OOB = 0x7ffffff0;                               // past num_records: moves nothing, still counts in vmcnt
  int c_tile = first, c_kt = 0, c_slot = 0, c_e = 0, c_nk = 1, c_soa = 0, c_sob = 0, c_ska = 0, c_skb = 0;
  const void* c_pa = desc[0].A;
  const void* c_pb = desc[0].B;
  int c_na = 0, c_nb = 0;
  int va[2][2], vb[2][2];                                       // [half][piece] byte offsets of the lane
  auto set_cur = [&]() {
    if (c_tile >= total_tiles) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { va[i >> 1][i & 1] = OOB; vb[i >> 1][i & 1] = OOB; }
      c_nk = 1 << 28;
      return;
    }
    while (c_e + 1 < n_problems && c_tile >= desc[c_e + 1].tile_start) ++c_e;
    const srhip_group_tn_desc d = desc[c_e];
    const int ntm = (d.M + 255) >> 8, ntn = (d.N + 255) >> 8;
    int tm_, tn_;
    pp_tile_mn(c_tile - d.tile_start, ntm, ntn, tm_, tn_);
    c_pa = d.A; c_pb = d.B;
    c_na = max(d.K * d.lda, (d.K - 1) * d.lda + d.M) * 2;
    c_nb = max(d.K * d.ldb, (d.K - 1) * d.ldb + d.N) * 2;
    c_nk = (d.K + PBK - 1) / PBK;
    c_soa = tm_ * 512; c_sob = tn_ * 512;                       // bytes: 256 features
    c_ska = PBK * d.lda * 2; c_skb = PBK * d.ldb * 2;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        va[h][i] = ((prow + 32 * i) * d.lda + pfa + 64 * h) * 2;
        vb[h][i] = ((prow + 32 * i) * d.ldb + pfb + 32 * h) * 2;
      }
  };
  auto issue = [&](auto kc) __attribute__((always_inline)) {      // kind: 0 = A0, 1 = B0, 2 = B1, 3 = A1
    constexpr int kind = decltype(kc)::value;
    constexpr bool isA = kind == 0 || kind == 3;
    bf16_t* dst = psm + c_slot * PH_EL + wave * 512;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(isA ? c_pa : c_pb), 0, isA ? c_na : c_nb, 0x00020000);
    const int so = isA ? c_soa + c_kt * c_ska : c_sob + c_kt * c_skb;
    const int (&vo)[2] = kind == 0 ? va[0] : (kind == 3 ? va[1] : (kind == 1 ? vb[0] : vb[1]));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, vo[0], so, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + 8 * 512), 16, vo[1], so, 0, 0);
    c_slot = c_slot + 1 == PSLOT ? 0 : c_slot + 1;
    if constexpr (kind == 3) {
      if (++c_kt >= c_nk) { c_kt = 0; c_tile += gridDim.x; set_cur(); }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  set_cur();
  issue(I0{}); issue(I1{}); issue(I2{}); issue(I3{});
  issue(I0{}); issue(I1{}); issue(I2{}); issue(I3{});
  static_assert(PLEAD == 8, "prologue issues two K-tiles");

  // ---- consumer fragment offsets (elements) inside a half-tile: tokens 4 lg + (l15 >> 2) (+ 16), features 16 ct + 4 (l15 & 3) .. + 3
  const int r0 = 4 * lg + (l15 >> 2);
  int foA[4], foB[2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int c = (wr * 4 + ((mt + wc) & 3)) * 16 + 4 * (l15 & 3);
    foA[mt] = r0 * 128 + ((((c >> 3) + 2 * r0) & 15) << 3) + (c & 4);
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int c = (wc * 2 + nt) * 16 + 4 * (l15 & 3);
    foB[nt] = r0 * 128 + ((((c >> 3) + 2 * r0) & 15) << 3) + (c & 4);
  }
  f32x4_t acc[4][8];                     // [column tile][row tile 4 h + mt] of the wave's 128 x 64
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  f32x4_t accb[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
  const u32x4_t ones = u32x4_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  u32x4_t fa[4][2], fb0[2][2], fb1[2][2];

  // fragment (16 features x 32 tokens of k-step ks) of a half-tile: two transpose reads
  auto rdf = [&](const bf16_t* st, int fo, int ks) __attribute__((always_inline)) {
    const u32x2_t x0 = ds_read_tr16(st + fo + ks * 32 * 128), x1 = ds_read_tr16(st + fo + ks * 32 * 128 + 16 * 128);
    return u32x4_t{x0[0], x0[1], x1[0], x1[1]};
  };
  bool do_bias = false;
  auto quadrant = [&](auto hc, auto jc, u32x4_t (&fb)[2][2]) __attribute__((always_inline)) {
    constexpr int h = decltype(hc)::value, j = decltype(jc)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          acc[2 * j + nt][4 * h + mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fb[nt][ks]), __builtin_bit_cast(bf16x8_t, fa[mt][ks]),
                                                                                acc[2 * j + nt][4 * h + mt], 0, 0, 0);
    if constexpr (h == j) {               // the phase right behind the read of A_h: column sums of the wave's own row tile
      if (do_bias) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          accb[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ones), __builtin_bit_cast(bf16x8_t, fa[0][ks]), accb[h], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
#define TPP_SYNC(wait_)                                                  \
  if (wait_) wait_vm_n<PINFL>();                                         \
  __builtin_amdgcn_s_barrier();                                          \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     \
  __builtin_amdgcn_sched_barrier(0);
#define TPP_END()                                                        \
  __builtin_amdgcn_sched_barrier(0);                                     \
  __builtin_amdgcn_s_barrier();                                          \
  __builtin_amdgcn_sched_barrier(0);
  int r_base = 0;                        // ring slot of A0 of the K-tile being multiplied
  auto ktile = [&](int kt) __attribute__((always_inline)) {
    const bool w0 = 4 * kt >= PLEAD - 2, w1 = 4 * kt + 1 >= PLEAD - 2, w2 = 4 * kt + 2 >= PLEAD - 2, w3 = 4 * kt + 3 >= PLEAD - 2;
    int sl[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { const int v = r_base + c; sl[c] = (v < PSLOT ? v : v - PSLOT) * PH_EL; }
    // ---- p0
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fb0[nt][ks] = rdf(psm + sl[1], foB[nt], ks);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fa[mt][ks] = rdf(psm + sl[0], foA[mt], ks);
    issue(I0{});
    TPP_SYNC(w0)
    quadrant(I0{}, I0{}, fb0);
    TPP_END()
    // ---- p1
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fb1[nt][ks] = rdf(psm + sl[2], foB[nt], ks);
    issue(I1{});
    TPP_SYNC(w1)
    quadrant(I0{}, I1{}, fb1);
    TPP_END()
    // ---- p2
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fa[mt][ks] = rdf(psm + sl[3], foA[mt], ks);
    issue(I2{});
    TPP_SYNC(w2)
    quadrant(I1{}, I1{}, fb1);
    TPP_END()
    // ---- p3
    issue(I3{});
    TPP_SYNC(w3)
    quadrant(I1{}, I0{}, fb0);
    TPP_END()
    r_base = r_base + 4 >= PSLOT ? r_base + 4 - PSLOT : r_base + 4;
  };

  WAIT_VM(0);
  __builtin_amdgcn_s_barrier();
  int ct = first, e_e = 0;
  for (int t = 0; t < my_tiles; ++t, ct += gridDim.x) {
    while (e_e + 1 < n_problems && ct >= desc[e_e + 1].tile_start) ++e_e;
    const srhip_group_tn_desc d = desc[e_e];
    const int ntm = (d.M + 255) >> 8, ntn = (d.N + 255) >> 8, nk = (d.K + PBK - 1) / PBK;
    int tm_, tn_;
    pp_tile_mn(ct - d.tile_start, ntm, ntn, tm_, tn_);
    do_bias = d.dbias != nullptr && tn_ == 0;
    if (wr == 1) __builtin_amdgcn_s_barrier();        // the lower row group runs one barrier behind
    for (int kt = 0; kt < nk; ++kt) ktile(kt);
    // ---- epilogue: lane holds C[m][n .. n + 3], m = row l15 of a row tile, n = 4 lg of a column tile
    const int mb = tm_ * 256 + wr * 128, nb = tn_ * 256 + wc * 64;
    const bool atomic = (d.flags & SRHIP_TN_ATOMIC) != 0, overwrite = (d.flags & SRHIP_TN_OVERWRITE) != 0;
    if (wr == 0) __builtin_amdgcn_s_barrier();
    WAIT_VM(0);
    __builtin_amdgcn_sched_barrier(0);
    if (do_bias) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int m = mb + h * 64 + wc * 16 + l15;
        if (lg == 0 && m < d.M) {
          const float v = accb[h][0];
          if (atomic) unsafeAtomicAdd(d.dbias + m, v); else if (overwrite) d.dbias[m] = v; else d.dbias[m] += v;
        }
        accb[h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    }
    const bool rd_c = !atomic && !overwrite && beta != 0.0f;
#pragma unroll
    for (int hq = 0; hq < 4; ++hq) {       // two row tiles at a time: their 8 quads of C are requested together
      f32x4_t res[2][4];
      if (rd_c) {
#pragma unroll
        for (int mq = 0; mq < 2; ++mq) {
          const int m = min(mb + (hq >> 1) * 64 + ((2 * (hq & 1) + mq + wc) & 3) * 16 + l15, d.M - 1);
#pragma unroll
          for (int i = 0; i < 4; ++i) res[mq][i] = *reinterpret_cast<const f32x4_t*>(d.C + (size_t)m * d.ldc + min(nb + i * 16 + lg * 4, d.N - 4));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mq = 0; mq < 2; ++mq) {
        const int mt = 2 * hq + mq, m = mb + (hq >> 1) * 64 + ((2 * (hq & 1) + mq + wc) & 3) * 16 + l15;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int n = nb + i * 16 + lg * 4;
          f32x4_t x = {alpha * acc[i][mt][0], alpha * acc[i][mt][1], alpha * acc[i][mt][2], alpha * acc[i][mt][3]};
          acc[i][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
          if (m < d.M && n < d.N) {
            float* cf = d.C + (size_t)m * d.ldc + n;
            if (atomic) {
              unsafeAtomicAdd(cf, x[0]); unsafeAtomicAdd(cf + 1, x[1]); unsafeAtomicAdd(cf + 2, x[2]); unsafeAtomicAdd(cf + 3, x[3]);
            } else {
              if (rd_c) { x[0] += beta * res[mq][i][0]; x[1] += beta * res[mq][i][1]; x[2] += beta * res[mq][i][2]; x[3] += beta * res[mq][i][3]; }
              *reinterpret_cast<f32x4_t*>(cf) = x;
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#undef TPP_SYNC
#undef TPP_END
}

// dst[i] += sum over the slabs of src[s * stride + i]: the second phase of a token-sliced product whose slices wrote slabs of their own
// (SRHIP_TN_OVERWRITE) instead of meeting in C through fp32 atomics.  One workgroup = 1024 elements of one entry.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const srhip_slab_desc* __restrict__ desc, int n) {
  const int b = blockIdx.x;
  int p = 0;
  {  // binary search: the last entry whose block_start <= b (a token-sliced table has hundreds of entries; a linear walk of dependent scalar
     // loads cost a workgroup as much as its product)
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (b >= desc[mid].block_start) lo = mid; else hi = mid - 1; }
    p = lo;
  }
  const srhip_slab_desc d = desc[p];
  const int i = ((b - d.block_start) * 256 + threadIdx.x) * 4;
  if (i >= d.count) return;
  if (i + 3 < d.count && (d.stride & 3) == 0) {
    f32x4_t a = *reinterpret_cast<const f32x4_t*>(d.dst + i);
    for (int s_ = 0; s_ < d.n_slabs; ++s_) {
      const f32x4_t v = *reinterpret_cast<const f32x4_t*>(d.src + (size_t)s_ * d.stride + i);
      a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
    }
    *reinterpret_cast<f32x4_t*>(d.dst + i) = a;
  } else {
    for (int j = i; j < min(i + 4, d.count); ++j) {
      float a = d.dst[j];
      for (int s_ = 0; s_ < d.n_slabs; ++s_) a += d.src[(size_t)s_ * d.stride + j];
      d.dst[j] = a;
    }
  }
}

}  // namespace

extern "C" int srhip_slab_reduce_f32(const srhip_slab_desc* desc_dev, int n, int total_blocks, void* stream) {
  if (!desc_dev || n <= 0 || total_blocks <= 0) return SR_EINVAL;
  SR_LAUNCH(slab_reduce_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, desc_dev, n);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_gemm_tn_grouped_pp_f32(const srhip_group_tn_desc* desc_dev, int n_problems, int total_tiles, float alpha,
                                            float beta, void* stream) {
  if (!desc_dev || n_problems <= 0 || n_problems > 4096 || total_tiles <= 0) return SR_EINVAL;
  constexpr size_t sm = (size_t)PSLOT * PH_EL * sizeof(bf16_t);
  (void)hipFuncSetAttribute((const void*)gemm_tn_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  SR_LAUNCH(gemm_tn_pp_kernel, dim3(min(total_tiles, 256)), dim3(512), sm, (hipStream_t)stream, desc_dev, n_problems, total_tiles, alpha, beta);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_gemm_tn_grouped_f32(const srhip_group_tn_desc* desc_dev, int n_problems, int total_tiles, float alpha,
                                         float beta, void* stream) {
  if (!desc_dev || n_problems <= 0 || n_problems > 4096 || total_tiles <= 0) return SR_EINVAL;
  SR_LAUNCH(gemm_tn_grouped_f32_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, desc_dev, n_problems, alpha,
                     beta);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
