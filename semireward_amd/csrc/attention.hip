// Fused multi-head self-attention (head_dim = 64), forward and backward, bf16 MFMA.
//
// Replaces reference semilearn/nets/vit/vit.py:100-104 (K4 in SURVEY.md 2c):
//     attn = softmax((q @ k^T) * scale);  x = (attn @ v).transpose(1,2).reshape(B,N,C)
// The reference materialises attn[B,H,N,N] in fp32 (38 MB / layer at Bt=24); here one
// workgroup owns one (image, head): K and V^T of the head live in LDS, every wave walks
// 16-query tiles, keeps the whole (padded) score row-set in registers, does the softmax with
// two cross-lane shuffles, and feeds P straight back into the PV MFMA.  Nothing N x N ever
// touches HBM.  Sequence lengths of the reference are short and fixed (257/197/199/<=512).
//
// MFMA layout facts used (16x16x32 bf16; lane l: l15 = l&15, g = l>>4):
//   a-operand: row l15 of the 16 x 32 "A" tile, k-slots 8g..8g+7    (8 bf16 = 16 B)
//   b-operand: col l15 of the 32 x 16 "B" tile, k-slots 8g..8g+7
//   result   : D[4g + r][l15], r = 0..3
// All products below are arranged so that (i) both operands are contiguous along the
// reduction index, and (ii) the score tile S^T[key][query] comes out with the QUERY on l15:
// the softmax statistics of a lane then belong to one query, and the 4 result registers of a
// score tile (4 consecutive keys) are directly k-slots of the next MFMA's b-operand.
// qkv layout: [B*N, 3*D] row-major (q | k | v, head-major inside each third) exactly as the
// qkv Linear writes it (vit.py:93-98) -- no permute kernel.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "srhip.h"

namespace {

constexpr int HD = 64;        // head dim
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ f32x4_t mfma16(s16x8_t a, s16x8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

__device__ __forceinline__ s16x8_t ld16(const bf16_t* p) { return *reinterpret_cast<const s16x8_t*>(p); }

__device__ __forceinline__ s16x8_t pack8(const float* lo, const float* hi) {
  uint4 r = {pack_bf2(lo[0], lo[1]), pack_bf2(lo[2], lo[3]), pack_bf2(hi[0], hi[1]), pack_bf2(hi[2], hi[3])};
  return __builtin_bit_cast(s16x8_t, r);
}

// Staging helpers.  Trip counts are compile-time (NP) and every global load of a helper is issued before the
// first LDS write, so a workgroup pays ONE memory round trip per operand instead of one per loop iteration
// (the first version serialised ~18 dependent L2/HBM round trips per workgroup: 60 us of latency for 2 us of MFMA).
// Rows [N, NP) are zero-filled.
// ---- conflict-free LDS images (PMC of the first layout, [key][72] rows and [d][NP + 8] transposes: 47 % of the LDS-array cycles were bank conflicts,
// and the two 8-byte V^T reads of an operand were fused into ds_read2_b64 = 8 cycles instead of 4 for one ds_read_b128) ----
// K image: [key][64] with 128-B rows; the 16-B chunk c of row r sits at chunk c ^ ((r >> 1) & 7).  A row covers half of the 64
// banks (even rows 0-31, odd rows 32-63), so the 16 lanes of every ds_read_b128 service group (MI355X_MICROARCH.md, LDS) hit 16
// distinct (row parity, chunk) pairs -- checked for all four groups and for both halves (chunk g and g + 4) of an operand.
template <int NP, int NT>
__device__ __forceinline__ void stage_rows_swz(bf16_t* dst, const bf16_t* src, int ld, int N, int tid) {
  constexpr int IT = (NP * 8 + NT - 1) / NT;
  u32x4_t v[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c = tid + i * NT, row = c >> 3, slot = c & 7;
    v[i] = u32x4_t{0u, 0u, 0u, 0u};
    if (c < NP * 8 && row < N) v[i] = *reinterpret_cast<const u32x4_t*>(src + (size_t)row * ld + slot * 8);
  }
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c = tid + i * NT, row = c >> 3, slot = c & 7;
    if (c < NP * 8) *reinterpret_cast<u32x4_t*>(dst + row * HD + ((slot ^ ((row >> 1) & 7)) << 3)) = v[i];
  }
}

// V^T image: dst[d][pos(key)], pitch TP.  Inside every group of 32 keys the order is permuted so that the 8 k-slots an MFMA lane
// group g needs (keys 4g..4g+3 of the even 16-key tile, then of the odd one -- exactly the registers P comes out in) are ONE
// 16-B chunk:  pos = 32 u + 8 ((kk & 15) >> 2) + 4 (kk >> 4) + (kk & 3), kk = key - 32 u;  chunk index ^= PI[(d >> 2) & 3] as in
// gemm.hip (rows d and d + 4 share banks when TP / 2 is 16 or 48 mod 64).
constexpr int vt_pitch(int NP) { return ((NP / 2) % 64 == 16 || (NP / 2) % 64 == 48) ? NP : NP + 32; }
__device__ __forceinline__ int swz4(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }
template <int NP, int NT>
__device__ __forceinline__ void stage_transposed_perm(bf16_t* dst, const bf16_t* src, int ld, int N, int tid) {
  constexpr int TP = vt_pitch(NP), PAIRS = NP / 2, IT = (PAIRS * 8 + NT - 1) / NT;
  u32x4_t va[IT], vb[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c = tid + i * NT, pr = c % PAIRS, slot = c / PAIRS, r0 = 2 * pr;
    va[i] = u32x4_t{0u, 0u, 0u, 0u};
    vb[i] = u32x4_t{0u, 0u, 0u, 0u};
    if (c < PAIRS * 8) {
      if (r0 < N) va[i] = *reinterpret_cast<const u32x4_t*>(src + (size_t)r0 * ld + slot * 8);
      if (r0 + 1 < N) vb[i] = *reinterpret_cast<const u32x4_t*>(src + (size_t)(r0 + 1) * ld + slot * 8);
    }
  }
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c = tid + i * NT, pr = c % PAIRS, slot = c / PAIRS, r0 = 2 * pr;
    if (c < PAIRS * 8) {
      const int kk = r0 & 31, chunk = (kk & 15) >> 2, within = ((kk >> 4) << 2) + (kk & 3);     // r0 even: the pair stays adjacent
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t lo = (va[i][j] & 0xffffu) | (vb[i][j] << 16);
        const uint32_t hi = (va[i][j] >> 16) | (vb[i][j] & 0xffff0000u);
        const int d0 = slot * 8 + 2 * j, d1 = d0 + 1;
        *reinterpret_cast<uint32_t*>(dst + d0 * TP + (r0 & ~31) + ((chunk ^ swz4(d0)) << 3) + within) = lo;
        *reinterpret_cast<uint32_t*>(dst + d1 * TP + (r0 & ~31) + ((chunk ^ swz4(d1)) << 3) + within) = hi;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward: grid = B*H workgroups of 512 threads (8 waves share the head's K / V^T: two workgroups per CU -> 4 waves per
// SIMD, which is what hides the per-query-tile dependency chain ds_read -> MFMA -> max -> exp -> MFMA).
// NKT = number of 16-key tiles (even), NP = 16*NKT.
#ifdef SRHIP_TUNING
__device__ long long srhip_attn_dbg[4 * 8192];
#define DBG_T(i) do { if (threadIdx.x == 0) srhip_attn_dbg[4 * ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) + (i)] = wall_clock64(); } while (0)
#else
#define DBG_T(i) do { } while (0)
#endif
constexpr int FWD_NT = 512, FWD_NW = FWD_NT / 64;
// dispatch_nkt instantiates NKT in {2, 8, 14, 18, 32} and picks the smallest >= ceil(N / 32) * 2: key tiles below the previous
// size are always complete
constexpr int nkt_lo(int nkt) { return nkt == 32 ? 18 : nkt == 18 ? 14 : nkt == 14 ? 8 : nkt == 8 ? 2 : 0; }
// VAR = true (BERT / Wav2Vec2 encoders): per-sequence key length (right-padded batch: keys >= key_len[b] are masked exactly like the
// additive -inf mask of the reference, every QUERY row is still computed -- the classifier averages over padded positions too) and
// train-mode dropout on the probabilities (counter-based, common.h:drop_pair_hash: one hash per two neighbouring keys of a row).
struct AttnVar { const int* key_len; uint32_t drop_key, drop_thresh; float drop_scale; };
template <int NKT, bool VAR>
__global__ __launch_bounds__(FWD_NT, 2) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                      float* __restrict__ lse, int N, int H, float scale, AttnVar av) {
  constexpr int NP = NKT * 16, TP = vt_pitch(NP);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem_raw);   // [NP][64], chunk-swizzled (stage_rows_swz)
  bf16_t* Vt = Ks + NP * HD;                          // [64][TP], key-permuted + chunk-swizzled (stage_transposed_perm)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / H, h = blockIdx.x % H, D = H * HD, ld = 3 * D;
  const bf16_t* base = qkv + (size_t)b * N * ld + h * HD;
  const float sc2 = scale * LOG2E;
  const int nqt = (N + 15) >> 4;
  DBG_T(0);
  const int klen = (VAR && av.key_len) ? __builtin_amdgcn_readfirstlane(av.key_len[b]) : N;
  const int kc = g ^ ((l15 >> 1) & 7);                // chunk of k-slots 8g.. of this lane's key row; the other half is kc ^ 4
  const int kof0 = l15 * HD + (kc << 3), kof1 = l15 * HD + ((kc ^ 4) << 3);
  const int vof = l15 * TP + ((g ^ swz4(l15)) << 3);
  // N = 257 is 16 full query tiles + ONE row: the wave that owns tile 16 works 3 tiles, the others 2.  Rotating the tile -> wave
  // map with the workgroup index spreads those third tiles over the SIMDs of a CU (two workgroups are resident per CU).
  const int wv = (wave + blockIdx.x) & (FWD_NW - 1);
  // first query tile's operands are requested before the staging traffic so they arrive under it
  s16x8_t qn0, qn1;
  {
    const int qc = min(wv * 16 + l15, N - 1);     // (wv < FWD_NW <= nqt for every supported N >= 113)
    const bf16_t* qp = base + (size_t)qc * ld + g * 8;
    qn0 = ld16(qp); qn1 = ld16(qp + 32);
  }
  stage_rows_swz<NP, FWD_NT>(Ks, base + D, ld, N, tid);
  stage_transposed_perm<NP, FWD_NT>(Vt, base + 2 * D, ld, N, tid);
  __syncthreads();
  DBG_T(1);
  for (int qt = wv; qt < nqt; qt += FWD_NW) {
    const int q = qt * 16 + l15;
    const s16x8_t q0 = qn0, q1 = qn1;
    if (qt + FWD_NW < nqt) {          // prefetch the next tile's Q fragments (hidden behind this tile's MFMAs)
      const int qc = min((qt + FWD_NW) * 16 + l15, N - 1);
      const bf16_t* qp = base + (size_t)qc * ld + g * 8;
      qn0 = ld16(qp); qn1 = ld16(qp + 32);
    }
    // raw scores stay unscaled in registers; p = 2^(s*c - max*c) is ONE fma + v_exp per element (c = scale*log2e > 0,
    // so max commutes with the scaling).  Only key tiles that straddle N need the per-element mask (wave-uniform test).
    f32x4_t s[NKT];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      f32x4_t a = {0.f, 0.f, 0.f, 0.f};
      a = mfma16(ld16(Ks + t * 16 * HD + kof0), q0, a);
      a = mfma16(ld16(Ks + t * 16 * HD + kof1), q1, a);
      // Only the key tiles past the previous dispatch size can straddle N (this instantiation serves 16 * nkt_lo < N <= NP): the
      // others are full by construction.  Testing every tile against the runtime N cost 144 v_cndmask + 94 v_readlane per
      // query tile (the 72 compare results were spilled from SGPR pairs to VGPR lanes) -- 40 % of the loop's VALU work.
      if (VAR) {
        if (t * 16 + 16 > klen) {
#pragma unroll
          for (int r = 0; r < 4; ++r) a[r] = (t * 16 + g * 4 + r < klen) ? a[r] : -INFINITY;
        }
      } else if (t >= nkt_lo(NKT) && t * 16 + 16 > N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = (t * 16 + g * 4 + r < N) ? a[r] : -INFINITY;
      }
      mx = fmaxf(mx, fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])));
      s[t] = a;
      // keep hipcc from hoisting all 36 fragment reads (144 VGPRs) to the top of the unrolled loop: that pushed the
      // kernel to 414 registers = 1 wave/SIMD = 1 workgroup/CU
      if ((t & (NKT >= 32 ? 3 : 1)) == (NKT >= 32 ? 3 : 1)) __builtin_amdgcn_sched_barrier(0);      // (N > 288: one workgroup per CU anyway, 256 registers)
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float nmx = -mx * sc2;
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = fast_exp2(fmaf(s[t][r], sc2, nmx));
        s[t][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    if (VAR && av.drop_thresh) {                     // nn.Dropout on the normalised probabilities: the row sum is the pre-dropout one
      // this lane's keys t * 16 + 4 g + r: pairs (r = 0, 1), (r = 2, 3) = pair indices 8 t + 2 g + j of the row
      const uint32_t h0 = ((((uint32_t)blockIdx.x * (uint32_t)N + (uint32_t)q) * (uint32_t)((N + 1) >> 1)) + (uint32_t)(g * 2)) * 0x9E3779B1u + av.drop_key;
#pragma unroll
      for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const uint32_t hh = fmix32(h0 + (uint32_t)(t * 8 + j) * 0x9E3779B1u);
          s[t][2 * j] = drop_pair_keep(hh, 0, av.drop_thresh) ? s[t][2 * j] : 0.f;
          s[t][2 * j + 1] = drop_pair_keep(hh, 1, av.drop_thresh) ? s[t][2 * j + 1] : 0.f;
        }
    }
    f32x4_t o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NKT / 2; ++u) {
      float lo[4] = {s[2 * u][0], s[2 * u][1], s[2 * u][2], s[2 * u][3]};
      float hi[4] = {s[2 * u + 1][0], s[2 * u + 1][1], s[2 * u + 1][2], s[2 * u + 1][3]};
      const s16x8_t pb = pack8(lo, hi);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        o[dt] = mfma16(ld16(Vt + dt * 16 * TP + 32 * u + vof), pb, o[dt]);
      }
      if (NKT < 32 || (u & 1)) __builtin_amdgcn_sched_barrier(0);
    }
    if (q < N) {
      const float inv = ((VAR && av.drop_thresh) ? av.drop_scale : 1.0f) / sum;
      bf16_t* op = out + ((size_t)b * N + q) * D + h * HD + g * 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        uint2 v = {pack_bf2(o[dt][0] * inv, o[dt][1] * inv), pack_bf2(o[dt][2] * inv, o[dt][3] * inv)};
        *reinterpret_cast<uint2*>(op + dt * 16) = v;
      }
      if (lse && g == 0) lse[((size_t)b * H + h) * N + q] = (mx * sc2 + log2f(sum)) * LN2;
    }
  }
#ifdef SRHIP_TUNING
  __syncthreads();
  DBG_T(2);
#endif
}

// ------------------------------------------------------------------------------------------------
// forward for long sequences (288 < N <= 512: the BERT legs).  Same workgroup = (sequence, head), same LDS images of K and V^T (128 KB at
// N = 512: one workgroup per CU, two waves per SIMD) -- what changes is the wave's walk.  attn_fwd_kernel holds the whole score row set of ONE
// query tile in registers (128 VGPRs at 512 keys) and runs  QK^T MFMAs -> softmax VALU -> PV MFMAs  strictly in sequence: with two waves per
// SIMD the matrix pipe idles under the softmax and the vector ALU under the products (227 TF/s at L = 512 with dropout, 0.09 of the peak; the
// largest non-GEMM share of the BERT step).  Here a wave carries TWO query tiles through the keys together, in key blocks of 128 with a running
// maximum (online softmax): two independent dependency chains per wave -- the products of one tile issue under the exponentials of the other --
// every K / V^T fragment read from LDS feeds both tiles, and the score registers are 2 x 32 instead of 128.  The rounding points are those of
// the whole-row kernel (probabilities bf16, fp32 statistics and accumulation); the running maximum only moves WHEN the rescaling happens.
// lse, the key-length mask and the dropout decision of an element (its global index) are unchanged, so attn_bwd_kernel pairs with either forward.
constexpr int PBT = 8;                     // key tiles per block (128 keys)
template <bool VAR>
__global__ __launch_bounds__(FWD_NT, 2) void attn_fwd_pair_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                 float* __restrict__ lse, int N, int H, float scale, AttnVar av) {
  constexpr int NKT = 32, NP = NKT * 16, TP = vt_pitch(NP);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem_raw);
  bf16_t* Vt = Ks + NP * HD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / H, h = blockIdx.x % H, D = H * HD, ld = 3 * D;
  const bf16_t* base = qkv + (size_t)b * N * ld + h * HD;
  const float sc2 = scale * LOG2E;
  const int nqt = (N + 15) >> 4;
  const int klen = (VAR && av.key_len) ? __builtin_amdgcn_readfirstlane(av.key_len[b]) : N;
  const int nblk = (min(klen, N) + PBT * 16 - 1) / (PBT * 16);      // key blocks with at least one valid key (klen >= 1)
  const int kc = g ^ ((l15 >> 1) & 7);
  const int kof0 = l15 * HD + (kc << 3), kof1 = l15 * HD + ((kc ^ 4) << 3);
  const int vof = l15 * TP + ((g ^ swz4(l15)) << 3);
  const int wv = (wave + blockIdx.x) & (FWD_NW - 1);
  stage_rows_swz<NP, FWD_NT>(Ks, base + D, ld, N, tid);
  stage_transposed_perm<NP, FWD_NT>(Vt, base + 2 * D, ld, N, tid);
  __syncthreads();
  // tiles wv, wv + 8 | wv + 16, wv + 24: a pair per round (a tile past the last one is computed on clamped rows and not stored)
  for (int qa = wv; qa < nqt; qa += 2 * FWD_NW) {
    const int qb = qa + FWD_NW;
    const int qrow[2] = {qa * 16 + l15, qb * 16 + l15};
    s16x8_t q0[2], q1[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const bf16_t* qp = base + (size_t)min(qrow[x], N - 1) * ld + g * 8;
      q0[x] = ld16(qp); q1[x] = ld16(qp + 32);
    }
    f32x4_t o[2][4];
    float mx[2] = {-INFINITY, -INFINITY}, sum[2] = {0.f, 0.f};
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[x][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int kb = 0; kb < nblk; ++kb) {
      const bf16_t* Kb = Ks + kb * PBT * 16 * HD;
      f32x4_t s[2][PBT];
      float bm[2] = {-INFINITY, -INFINITY};
#pragma unroll
      for (int t = 0; t < PBT; ++t) {
        const s16x8_t k0 = ld16(Kb + t * 16 * HD + kof0), k1 = ld16(Kb + t * 16 * HD + kof1);
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          f32x4_t a = {0.f, 0.f, 0.f, 0.f};
          a = mfma16(k0, q0[x], a);
          a = mfma16(k1, q1[x], a);
          s[x][t] = a;
        }
        if ((kb * PBT + t) * 16 + 16 > klen) {             // wave-uniform: only the tile that straddles the key length (and those behind it)
          const int k0i = (kb * PBT + t) * 16 + g * 4;
#pragma unroll
          for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[x][t][r] = (k0i + r < klen) ? s[x][t][r] : -INFINITY;
        }
#pragma unroll
        for (int x = 0; x < 2; ++x) bm[x] = fmaxf(bm[x], fmaxf(fmaxf(s[x][t][0], s[x][t][1]), fmaxf(s[x][t][2], s[x][t][3])));
        if ((t & 1) == 1) __builtin_amdgcn_sched_barrier(0);      // (keeps the fragment reads of the whole block from being hoisted: registers)
      }
      float nm[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        float m = bm[x];
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        const float mn = fmaxf(mx[x], m);                   // finite: every block walked holds a valid key
        const float alpha = fast_exp2((mx[x] - mn) * sc2);  // first block: exp2(-inf) = 0 on zero accumulators
        mx[x] = mn;
        nm[x] = -mn * sc2;
        sum[x] *= alpha;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[x][dt][r] *= alpha;
      }
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        float ps = 0.f;
#pragma unroll
        for (int t = 0; t < PBT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = fast_exp2(fmaf(s[x][t][r], sc2, nm[x]));
            s[x][t][r] = p;
            ps += p;
          }
        sum[x] += ps;
        if (VAR && av.drop_thresh) {                         // the element's pair hash, as in attn_fwd_kernel / attn_bwd_kernel
          const uint32_t h0 = ((((uint32_t)blockIdx.x * (uint32_t)N + (uint32_t)qrow[x]) * (uint32_t)((N + 1) >> 1)) + (uint32_t)(g * 2)) * 0x9E3779B1u +
                              av.drop_key;
#pragma unroll
          for (int t = 0; t < PBT; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const uint32_t hh = fmix32(h0 + (uint32_t)((kb * PBT + t) * 8 + j) * 0x9E3779B1u);
              s[x][t][2 * j] = drop_pair_keep(hh, 0, av.drop_thresh) ? s[x][t][2 * j] : 0.f;
              s[x][t][2 * j + 1] = drop_pair_keep(hh, 1, av.drop_thresh) ? s[x][t][2 * j + 1] : 0.f;
            }
        }
      }
#pragma unroll
      for (int u = 0; u < PBT / 2; ++u) {
        s16x8_t pb[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          float lo[4] = {s[x][2 * u][0], s[x][2 * u][1], s[x][2 * u][2], s[x][2 * u][3]};
          float hi[4] = {s[x][2 * u + 1][0], s[x][2 * u + 1][1], s[x][2 * u + 1][2], s[x][2 * u + 1][3]};
          pb[x] = pack8(lo, hi);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const s16x8_t vf = ld16(Vt + dt * 16 * TP + 32 * (kb * (PBT / 2) + u) + vof);
          o[0][dt] = mfma16(vf, pb[0], o[0][dt]);
          o[1][dt] = mfma16(vf, pb[1], o[1][dt]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      float sm_ = sum[x];
      sm_ += __shfl_xor(sm_, 16, 64);
      sm_ += __shfl_xor(sm_, 32, 64);
      const int q = qrow[x];
      if (q < N) {
        const float inv = ((VAR && av.drop_thresh) ? av.drop_scale : 1.0f) / sm_;
        bf16_t* op = out + ((size_t)b * N + q) * D + h * HD + g * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          uint2 v = {pack_bf2(o[x][dt][0] * inv, o[x][dt][1] * inv), pack_bf2(o[x][dt][2] * inv, o[x][dt][3] * inv)};
          *reinterpret_cast<uint2*>(op + dt * 16) = v;
        }
        if (lse && g == 0) lse[((size_t)b * H + h) * N + q] = (mx[x] * sc2 + log2f(sm_)) * LN2;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward, part 1: dQ.  One workgroup per (image, head); wave walks 16-query tiles.
//   S^T, dP^T (query on l15) per key-tile pair -> P, dS (bf16) -> dQ^T[d][q] += K^T[d][keys] . dS^T[keys][q]
// delta[q] = rowsum(dO[q] * O[q]) is computed in-kernel from the saved forward output.
// grid = (B*H, QS): blockIdx.y takes every QS-th group of BWD_NW query tiles (the backward batch is small -- 16 images --
// so one workgroup per head would leave 60 % of the CUs idle).
constexpr int BWD_NT = 512, BWD_NW = BWD_NT / 64;

// VG = true (N > 288: K, V and K^T images no longer fit the 160 KB of LDS together): the V row fragments -- plain 16-byte row reads, the
// A operand of dP = V . dO^T -- come straight from global memory / L2, software-pipelined one key-tile pair ahead.
template <int NKT, bool VAR, bool VG>
__device__ __forceinline__ void attn_bwd_dq_body(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o_fwd,
                                                         const bf16_t* __restrict__ d_out, const float* __restrict__ lse,
                                                         bf16_t* __restrict__ dqkv, float* __restrict__ delta,
                                                         int N, int H, float scale, AttnVar av) {
  constexpr int NP = NKT * 16, TP = vt_pitch(NP);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem_raw);   // [NP][64]  chunk-swizzled (stage_rows_swz)
  bf16_t* Vs = Ks + NP * HD;                          // [NP][64]  chunk-swizzled (absent when VG)
  bf16_t* Kt = Vs + (VG ? 0 : NP * HD);               // [64][TP]  key-permuted + chunk-swizzled (stage_transposed_perm)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int kc_ = g ^ ((l15 >> 1) & 7);
  const int kof0 = l15 * HD + (kc_ << 3), kof1 = l15 * HD + ((kc_ ^ 4) << 3);
  const int vof = l15 * TP + ((g ^ swz4(l15)) << 3);
  const int b = blockIdx.x / H, h = blockIdx.x % H, D = H * HD, ld = 3 * D;
  const bf16_t* base = qkv + (size_t)b * N * ld + h * HD;
  const float sc2 = scale * LOG2E;
  const int nqt = (N + 15) >> 4;
  s16x8_t nq0, nq1, nd0, nd1, no0, no1;
  float nlse;
  auto fetch = [&](int qt_) {
    const int qc = min(qt_ * 16 + l15, N - 1);
    const bf16_t* qp = base + (size_t)qc * ld + g * 8;
    const bf16_t* dop = d_out + ((size_t)b * N + qc) * D + h * HD + g * 8;
    const bf16_t* op = o_fwd + ((size_t)b * N + qc) * D + h * HD + g * 8;
    nq0 = ld16(qp); nq1 = ld16(qp + 32);
    nd0 = ld16(dop); nd1 = ld16(dop + 32);
    no0 = ld16(op); no1 = ld16(op + 32);
    nlse = lse[((size_t)b * H + h) * N + qc];
  };
  const int qstride = BWD_NW * gridDim.y, qfirst = blockIdx.y * BWD_NW + wave;
  const int klen = (VAR && av.key_len) ? __builtin_amdgcn_readfirstlane(av.key_len[b]) : N;
  fetch(qfirst);
  stage_rows_swz<NP, BWD_NT>(Ks, base + D, ld, N, tid);
  if (!VG) stage_rows_swz<NP, BWD_NT>(Vs, base + 2 * D, ld, N, tid);
  stage_transposed_perm<NP, BWD_NT>(Kt, base + D, ld, N, tid);
  __syncthreads();
  DBG_T(1);
  s16x8_t vg[2][2];                                   // VG: V fragments of key-tile pair u (rows >= N are clamped: their p is 0)
  auto vfetch = [&](int u_) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const bf16_t* vp = base + 2 * D + (size_t)min((2 * u_ + e) * 16 + l15, N - 1) * ld + g * 8;
      vg[e][0] = ld16(vp); vg[e][1] = ld16(vp + 32);
    }
  };
  for (int qt = qfirst; qt < nqt; qt += qstride) {
    const int q = qt * 16 + l15;
    const s16x8_t q0 = nq0, q1 = nq1, do0 = nd0, do1 = nd1;
    // delta[q]: this lane holds d-slots 8g..8g+7 and 32+8g.. of row q
    float dl = 0.f;
    {
      const s16x8_t o0 = no0, o1 = no1;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        dl += bf2f((bf16_t)do0[j]) * bf2f((bf16_t)o0[j]) + bf2f((bf16_t)do1[j]) * bf2f((bf16_t)o1[j]);
      dl += __shfl_xor(dl, 16, 64);
      dl += __shfl_xor(dl, 32, 64);
    }
    const float lse2 = nlse * LOG2E;
    if (qt + qstride < nqt) fetch(qt + qstride);      // next tile's operands travel under this tile's MFMAs
    if (q < N && g == 0) delta[((size_t)b * H + h) * N + q] = dl;
    f32x4_t dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (VG) vfetch(0);
    const uint32_t drow = (uint32_t)blockIdx.x * (uint32_t)N + (uint32_t)q;          // dropout: row of the probability matrix
#pragma unroll 1
    for (int u = 0; u < NKT / 2; ++u) {
      float ds[2][4];
      s16x8_t cv[2][2];
      if (VG) {
#pragma unroll
        for (int e = 0; e < 2; ++e) { cv[e][0] = vg[e][0]; cv[e][1] = vg[e][1]; }
        if (u + 1 < NKT / 2) vfetch(u + 1);
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int t = 2 * u + e;
        f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        s = mfma16(ld16(Ks + t * 16 * HD + kof0), q0, s);
        s = mfma16(ld16(Ks + t * 16 * HD + kof1), q1, s);
        dp = mfma16(VG ? cv[e][0] : ld16(Vs + t * 16 * HD + kof0), do0, dp);
        dp = mfma16(VG ? cv[e][1] : ld16(Vs + t * 16 * HD + kof1), do1, dp);
        if (VAR) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = t * 16 + g * 4 + r;
            const float p = key < klen ? fast_exp2(s[r] * sc2 - lse2) : 0.f;
            float dpv = dp[r];
            if (av.drop_thresh) dpv = drop_keep_attn(drow, (uint32_t)N, (uint32_t)key, av.drop_key, av.drop_thresh) ? dpv * av.drop_scale : 0.f;
            ds[e][r] = p * (dpv - dl) * scale;
          }
        } else if (t >= nkt_lo(NKT)) {                 // (wave-uniform) only these key tiles can hold padded keys, see attn_fwd_kernel
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = t * 16 + g * 4 + r;
            const float p = key < N ? fast_exp2(s[r] * sc2 - lse2) : 0.f;
            ds[e][r] = p * (dp[r] - dl) * scale;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) ds[e][r] = fast_exp2(s[r] * sc2 - lse2) * (dp[r] - dl) * scale;
        }
      }
      const s16x8_t dsb = pack8(ds[0], ds[1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dq[dt] = mfma16(ld16(Kt + dt * 16 * TP + 32 * u + vof), dsb, dq[dt]);
      }
    }
    if (q < N) {
      bf16_t* dqp = dqkv + ((size_t)b * N + q) * ld + h * HD + g * 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        uint2 v = {pack_bf2(dq[dt][0], dq[dt][1]), pack_bf2(dq[dt][2], dq[dt][3])};
        *reinterpret_cast<uint2*>(dqp + dt * 16) = v;
      }
    }
  }
}

// backward, part 2: dK, dV.  One workgroup per (image, head); each wave owns 16-key tiles and
// walks query-tile pairs:  S[q][key], dP[q][key] with the KEY on l15 ->
//   dV^T[d][key] += dO^T[d][q] . P[q][key]      dK^T[d][key] += Q^T[d][q] . dS[q][key]
// QG = true (N > 288: the four images no longer fit the LDS): the (q, dO) ROW fragments come from L2 instead, two query-tile pairs in flight.
template <int NKT, bool VAR, bool QG>
__device__ __forceinline__ void attn_bwd_dkv_body(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o_fwd,
                                                  const bf16_t* __restrict__ d_out, const float* __restrict__ lse,
                                                  bf16_t* __restrict__ dqkv, int N, int H, float scale, AttnVar av) {
  constexpr int NP = NKT * 16, TP = vt_pitch(NP);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* Qt = reinterpret_cast<bf16_t*>(smem_raw);   // [64][TP]  query-permuted + chunk-swizzled (stage_transposed_perm)
  bf16_t* dOt = Qt + 64 * TP;                         // [64][TP]
  bf16_t* Qs = dOt + 64 * TP;                         // [NP][64]  chunk-swizzled rows (stage_rows_swz): the a-operands of S and dP
  bf16_t* dOs = Qs + (QG ? 0 : NP * HD);              // [NP][64]  (both absent when QG)
  float* lse_s = reinterpret_cast<float*>(dOs + (QG ? 0 : NP * HD));   // [NP]  (already * log2e)
  float* dl_s = lse_s + NP;                                            // [NP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / H, h = blockIdx.x % H, D = H * HD, ld = 3 * D;
  const bf16_t* base = qkv + (size_t)b * N * ld + h * HD;
  const bf16_t* dobase = d_out + (size_t)b * N * D + h * HD;
  stage_transposed_perm<NP, BWD_NT>(Qt, base, ld, N, tid);
  stage_transposed_perm<NP, BWD_NT>(dOt, dobase, D, N, tid);
  if (!QG) {
    // Every wave walks ALL query tiles for its key tile: read from L2 that is 8 waves x 8 KB per step through the 64 B/clk vector-memory
    // path of the CU (measured with wall-clock stamps: the loop of this pass took 38 us against 13 us for the dQ pass, which reads its
    // fragments from LDS).  Rows >= N are zero.
    stage_rows_swz<NP, BWD_NT>(Qs, base, ld, N, tid);
    stage_rows_swz<NP, BWD_NT>(dOs, dobase, D, N, tid);
  }
  const int kc_ = g ^ ((l15 >> 1) & 7);
  const int kof0 = l15 * HD + (kc_ << 3), kof1 = l15 * HD + ((kc_ ^ 4) << 3);
  const int vof = (threadIdx.x & 15) * TP + (((threadIdx.x >> 4 & 3) ^ swz4(threadIdx.x & 15)) << 3);
  // delta[q] = rowsum(dO[q] * O[q]) is recomputed here (64 MACs per query from rows that are L2-hot) instead of read from the dQ pass: the
  // two passes then have no dependency and share ONE launch (blockIdx.z picks the role) -- one launch less per layer on a latency-bound
  // chain, and twice the workgroups for a backward batch that does not fill the chip.
  for (int i = tid; i < NP; i += BWD_NT) {
    lse_s[i] = i < N ? lse[((size_t)b * H + h) * N + i] * LOG2E : INFINITY;
    float dl = 0.f;
    if (i < N) {
      const bf16_t* dop = dobase + (size_t)i * D;
      const bf16_t* op = o_fwd + ((size_t)b * N + i) * D + h * HD;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const s16x8_t a = ld16(dop + c * 8), o8 = ld16(op + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) dl += bf2f((bf16_t)a[j]) * bf2f((bf16_t)o8[j]);
      }
    }
    dl_s[i] = dl;
  }
  __syncthreads();
  DBG_T(1);
  const float sc2 = scale * LOG2E;
  const int klen = (VAR && av.key_len) ? __builtin_amdgcn_readfirstlane(av.key_len[b]) : N;
  for (int kt = blockIdx.y * BWD_NW + wave; kt < NKT; kt += BWD_NW * gridDim.y) {
    const int key = kt * 16 + l15, kc = min(key, N - 1);
    if (kt * 16 >= N) break;
    const bf16_t* kp = base + D + (size_t)kc * ld + g * 8;
    const bf16_t* vp = base + 2 * D + (size_t)kc * ld + g * 8;
    const s16x8_t k0 = ld16(kp), k1 = ld16(kp + 32), v0 = ld16(vp), v1 = ld16(vp + 32);
    f32x4_t dv[4], dk[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dv[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dk[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    // (q, dO) row fragments of query-tile pair u come straight from L2 (~1 us away when the chip is busy, an iteration is ~0.3 us):
    // TWO pairs are in flight -- buffers A (even u) and B (odd u) -- so a pair is requested two iterations before it is used
    // (one pair ahead: 51.6 us for the 16 gradient images, the loop sat on vmcnt every iteration).
    s16x8_t fqa[2][2], fda[2][2], fqb[2][2], fdb[2][2];
    auto fetch = [&](s16x8_t (&fq)[2][2], s16x8_t (&fd)[2][2], int u_) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int qc = min((2 * u_ + e) * 16 + l15, N - 1);
        const bf16_t* qp = base + (size_t)qc * ld + g * 8;
        const bf16_t* dop = dobase + (size_t)qc * D + g * 8;
        fq[e][0] = ld16(qp); fq[e][1] = ld16(qp + 32);
        fd[e][0] = ld16(dop); fd[e][1] = ld16(dop + 32);
      }
    };
    auto body = [&](const s16x8_t (&cq)[2][2], const s16x8_t (&cd)[2][2], int u) {
      float pp[2][4], ds[2][4];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        s = mfma16(cq[e][0], k0, s);            // a-operand rows = queries, b-operand cols = keys
        s = mfma16(cq[e][1], k1, s);
        dp = mfma16(cd[e][0], v0, dp);
        dp = mfma16(cd[e][1], v1, dp);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qq = (2 * u + e) * 16 + g * 4 + r;      // result row = query
          // padded queries: lse_s = +inf -> p = 0 exactly.  Padded KEY columns (this lane's key >= N) may hold anything: a column
          // of P / dS only feeds the dK / dV rows of that key, which are never stored.
          float p = fast_exp2(s[r] * sc2 - lse_s[qq]);
          float pk = p, dpv = dp[r];
          if (VAR) {                              // masked keys (rows [klen, N) ARE stored: padded positions have dK = dV = 0)
            p = key < klen ? p : 0.f;
            pk = p;
            if (av.drop_thresh) {
              const bool keep = drop_keep_attn((uint32_t)blockIdx.x * (uint32_t)N + (uint32_t)qq, (uint32_t)N, (uint32_t)key, av.drop_key, av.drop_thresh);
              pk = keep ? p * av.drop_scale : 0.f;
              dpv = keep ? dpv * av.drop_scale : 0.f;
            }
          }
          pp[e][r] = pk;
          ds[e][r] = p * (dpv - dl_s[qq]) * scale;
        }
      }
      const s16x8_t pb = pack8(pp[0], pp[1]), dsb = pack8(ds[0], ds[1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int off = dt * 16 * TP + 32 * u + vof;
        dv[dt] = mfma16(ld16(dOt + off), pb, dv[dt]);
        dk[dt] = mfma16(ld16(Qt + off), dsb, dk[dt]);
      }
    };
    constexpr int NU = NKT / 2;
    if (!QG) {
#pragma unroll 1
      for (int u = 0; u < NU; ++u) {
        s16x8_t cq[2][2], cd[2][2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int ro = (2 * u + e) * 16 * HD;
          cq[e][0] = ld16(Qs + ro + kof0); cq[e][1] = ld16(Qs + ro + kof1);
          cd[e][0] = ld16(dOs + ro + kof0); cd[e][1] = ld16(dOs + ro + kof1);
        }
        body(cq, cd, u);
      }
    } else {
      fetch(fqa, fda, 0);
      if (NU > 1) fetch(fqb, fdb, 1);
#pragma unroll 1
      for (int u = 0; u < NU; u += 2) {
        s16x8_t cq[2][2], cd[2][2];
#pragma unroll
        for (int e = 0; e < 2; ++e) { cq[e][0] = fqa[e][0]; cq[e][1] = fqa[e][1]; cd[e][0] = fda[e][0]; cd[e][1] = fda[e][1]; }
        if (u + 2 < NU) fetch(fqa, fda, u + 2);
        body(cq, cd, u);
        if (u + 1 < NU) {
#pragma unroll
          for (int e = 0; e < 2; ++e) { cq[e][0] = fqb[e][0]; cq[e][1] = fqb[e][1]; cd[e][0] = fdb[e][0]; cd[e][1] = fdb[e][1]; }
          if (u + 3 < NU) fetch(fqb, fdb, u + 3);
          body(cq, cd, u + 1);
        }
      }
    }
    if (key < N) {
      bf16_t* dkp = dqkv + ((size_t)b * N + key) * ld + D + h * HD + g * 4;
      bf16_t* dvp = dkp + D;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        uint2 a = {pack_bf2(dk[dt][0], dk[dt][1]), pack_bf2(dk[dt][2], dk[dt][3])};
        uint2 c = {pack_bf2(dv[dt][0], dv[dt][1]), pack_bf2(dv[dt][2], dv[dt][3])};
        *reinterpret_cast<uint2*>(dkp + dt * 16) = a;
        *reinterpret_cast<uint2*>(dvp + dt * 16) = c;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The LDS-resident backward (N <= 288: every instantiation of the ViT / Wav2Vec2 shapes), two tiles per wave.
//
// Why.  A backward workgroup is a chain of dependent steps per tile -- ds_read -> MFMA -> v_exp / scale -> pack -> MFMA -- walked by two
// waves per SIMD (the four 37-KB images of a role leave room for ONE workgroup per CU): in-kernel stamps put an iteration of that
// chain at ~1100 clocks for 128 clocks of MFMA issue.  The workgroup owns its CU outright, so its CU time is what the gradient rows'
// backward costs the step while the deferred inference rows fill the other CUs (bench.py other_kernels: 58 us per launch beside them,
// 31 us alone -- three rounds of 192 whole-CU workgroups on the ~64 CUs left).  The registers a second resident workgroup would need are
// free (129 of 256 in use), so each wave carries TWO independent tiles through the chain at once: twice the instruction-level
// parallelism under every latency, every K / V (dQ role) or Q / dO (dK / dV role) fragment read from LDS once for both tiles, and the
// wave that owns the odd 17th tile of N = 257 needs two rounds instead of three.  delta = rowsum(dO * O) is computed once per workgroup
// in the staging phase (both roles), so a tile's operands are only its q / dO fragments.
// VG (N > 288: K, V and K^T images exceed the LDS together): the V row fragments -- the A operand of dP = V . dO^T, plain 16-byte row reads that
// BOTH query tiles of the pair use -- come from global memory / L2, requested one key-tile pair ahead.
template <int NKT, bool VAR, bool VG = false>
__device__ __forceinline__ void attn_bwd_dq_pair(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o_fwd,
                                                 const bf16_t* __restrict__ d_out, const float* __restrict__ lse,
                                                 bf16_t* __restrict__ dqkv, float* __restrict__ delta, int N, int H, float scale, AttnVar av) {
  constexpr int NP = NKT * 16, TP = vt_pitch(NP);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem_raw);   // [NP][64]  chunk-swizzled (stage_rows_swz)
  bf16_t* Vs = Ks + NP * HD;                          // [NP][64]  (absent when VG)
  bf16_t* Kt = Vs + (VG ? 0 : NP * HD);               // [64][TP]  key-permuted + chunk-swizzled (stage_transposed_perm)
  float* lse_s = reinterpret_cast<float*>(Kt + 64 * TP);   // [NP]  (* log2e; +inf for padded queries)
  float* dl_s = lse_s + NP;                                // [NP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int kc_ = g ^ ((l15 >> 1) & 7);
  const int kof0 = l15 * HD + (kc_ << 3), kof1 = l15 * HD + ((kc_ ^ 4) << 3);
  const int vof = l15 * TP + ((g ^ swz4(l15)) << 3);
  const int b = blockIdx.x / H, h = blockIdx.x % H, D = H * HD, ld = 3 * D;
  const bf16_t* base = qkv + (size_t)b * N * ld + h * HD;
  const bf16_t* dobase = d_out + (size_t)b * N * D + h * HD;
  const float sc2 = scale * LOG2E;
  const int nqt = (N + 15) >> 4;
  const int qstride = BWD_NW * gridDim.y, qfirst = blockIdx.y * BWD_NW + wave;
  const int klen = (VAR && av.key_len) ? __builtin_amdgcn_readfirstlane(av.key_len[b]) : N;
  // the q / dO fragments of a tile pair (rows clamped: a tile past the end is computed on row N - 1 and never stored)
  s16x8_t nq[2][2], nd[2][2];
  auto fetch = [&](int qa) {
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int qc = min((qa + x * qstride) * 16 + l15, N - 1);
      const bf16_t* qp = base + (size_t)qc * ld + g * 8;
      const bf16_t* dop = dobase + (size_t)qc * D + g * 8;
      nq[x][0] = ld16(qp); nq[x][1] = ld16(qp + 32);
      nd[x][0] = ld16(dop); nd[x][1] = ld16(dop + 32);
    }
  };
  fetch(qfirst);
  stage_rows_swz<NP, BWD_NT>(Ks, base + D, ld, N, tid);
  if (!VG) stage_rows_swz<NP, BWD_NT>(Vs, base + 2 * D, ld, N, tid);
  stage_transposed_perm<NP, BWD_NT>(Kt, base + D, ld, N, tid);
  s16x8_t vg[2][2];                                   // VG: V fragments of key-tile pair u (rows >= N are clamped: their p is 0)
  auto vfetch = [&](int u_) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const bf16_t* vp = base + 2 * D + (size_t)min((2 * u_ + e) * 16 + l15, N - 1) * ld + g * 8;
      vg[e][0] = ld16(vp); vg[e][1] = ld16(vp + 32);
    }
  };
  for (int i = tid; i < NP; i += BWD_NT) {
    float dl = 0.f;
    if (i < N) {
      const bf16_t* dop = dobase + (size_t)i * D;
      const bf16_t* op = o_fwd + ((size_t)b * N + i) * D + h * HD;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const s16x8_t a = ld16(dop + c * 8), o8 = ld16(op + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) dl += bf2f((bf16_t)a[j]) * bf2f((bf16_t)o8[j]);
      }
      if (blockIdx.y == 0) delta[((size_t)b * H + h) * N + i] = dl;
    }
    lse_s[i] = i < N ? lse[((size_t)b * H + h) * N + i] * LOG2E : INFINITY;
    dl_s[i] = dl;
  }
  __syncthreads();
  DBG_T(1);
  for (int qa = qfirst; qa < nqt; qa += 2 * qstride) {
    s16x8_t q0[2], q1[2], do0[2], do1[2];
    float lse2[2], dl[2];
    int q[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      q0[x] = nq[x][0]; q1[x] = nq[x][1]; do0[x] = nd[x][0]; do1[x] = nd[x][1];
      q[x] = (qa + x * qstride) * 16 + l15;
      const int qs = min(q[x], NP - 1);
      lse2[x] = lse_s[qs]; dl[x] = dl_s[qs];
    }
    if (qa + 2 * qstride < nqt) fetch(qa + 2 * qstride);      // the next pair's operands travel under this pair's MFMAs
    f32x4_t dq[2][4];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dq[x][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (VG) vfetch(0);
#pragma unroll 1
    for (int u = 0; u < NKT / 2; ++u) {
      float ds[2][2][4];
      s16x8_t cv[2][2];
      if (VG) {
#pragma unroll
        for (int e = 0; e < 2; ++e) { cv[e][0] = vg[e][0]; cv[e][1] = vg[e][1]; }
        if (u + 1 < NKT / 2) vfetch(u + 1);
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int t = 2 * u + e;
        const s16x8_t ka = ld16(Ks + t * 16 * HD + kof0), kb = ld16(Ks + t * 16 * HD + kof1);
        const s16x8_t va = VG ? cv[e][0] : ld16(Vs + t * 16 * HD + kof0), vb = VG ? cv[e][1] : ld16(Vs + t * 16 * HD + kof1);
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          s = mfma16(ka, q0[x], s);
          s = mfma16(kb, q1[x], s);
          dp = mfma16(va, do0[x], dp);
          dp = mfma16(vb, do1[x], dp);
          if (VAR) {
            const uint32_t drow = (uint32_t)blockIdx.x * (uint32_t)N + (uint32_t)min(q[x], N - 1);
            // this lane's four keys are two pairs of the row: one hash each (common.h drop_pair_hash), as in the forward
            uint32_t hh[2] = {0u, 0u};
            if (av.drop_thresh) {
#pragma unroll
              for (int j = 0; j < 2; ++j) hh[j] = drop_pair_hash(drow, (uint32_t)((N + 1) >> 1), (uint32_t)(t * 8 + g * 2 + j), av.drop_key);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = t * 16 + g * 4 + r;
              const float p = key < klen ? fast_exp2(s[r] * sc2 - lse2[x]) : 0.f;
              float dpv = dp[r];
              if (av.drop_thresh) dpv = drop_pair_keep(hh[r >> 1], r & 1, av.drop_thresh) ? dpv * av.drop_scale : 0.f;
              ds[x][e][r] = p * (dpv - dl[x]) * scale;
            }
          } else if (t >= nkt_lo(NKT)) {                 // (wave-uniform) only these key tiles can hold padded keys, see attn_fwd_kernel
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = t * 16 + g * 4 + r;
              const float p = key < N ? fast_exp2(s[r] * sc2 - lse2[x]) : 0.f;
              ds[x][e][r] = p * (dp[r] - dl[x]) * scale;
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) ds[x][e][r] = fast_exp2(s[r] * sc2 - lse2[x]) * (dp[r] - dl[x]) * scale;
          }
        }
      }
      const s16x8_t dsa = pack8(ds[0][0], ds[0][1]), dsb = pack8(ds[1][0], ds[1][1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const s16x8_t kt = ld16(Kt + dt * 16 * TP + 32 * u + vof);
        dq[0][dt] = mfma16(kt, dsa, dq[0][dt]);
        dq[1][dt] = mfma16(kt, dsb, dq[1][dt]);
      }
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      if (q[x] < N) {
        bf16_t* dqp = dqkv + ((size_t)b * N + q[x]) * ld + h * HD + g * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          uint2 v = {pack_bf2(dq[x][dt][0], dq[x][dt][1]), pack_bf2(dq[x][dt][2], dq[x][dt][3])};
          *reinterpret_cast<uint2*>(dqp + dt * 16) = v;
        }
      }
    }
  }
}

// dK / dV role of the same: each wave owns 16-key tiles, two at a time, and walks the query-tile pairs; S[q][key], dP[q][key] with the KEY
// on l15 ->  dV^T[d][key] += dO^T[d][q] . P[q][key],  dK^T[d][key] += Q^T[d][q] . dS[q][key].
template <int NKT, bool VAR>
__device__ __forceinline__ void attn_bwd_dkv_pair(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o_fwd,
                                                  const bf16_t* __restrict__ d_out, const float* __restrict__ lse,
                                                  bf16_t* __restrict__ dqkv, int N, int H, float scale, AttnVar av) {
  constexpr int NP = NKT * 16, TP = vt_pitch(NP);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* Qt = reinterpret_cast<bf16_t*>(smem_raw);   // [64][TP]  query-permuted + chunk-swizzled (stage_transposed_perm)
  bf16_t* dOt = Qt + 64 * TP;                         // [64][TP]
  bf16_t* Qs = dOt + 64 * TP;                         // [NP][64]  chunk-swizzled rows (stage_rows_swz): the a-operands of S and dP
  bf16_t* dOs = Qs + NP * HD;                         // [NP][64]
  float* lse_s = reinterpret_cast<float*>(dOs + NP * HD);   // [NP]  (already * log2e)
  float* dl_s = lse_s + NP;                                 // [NP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / H, h = blockIdx.x % H, D = H * HD, ld = 3 * D;
  const bf16_t* base = qkv + (size_t)b * N * ld + h * HD;
  const bf16_t* dobase = d_out + (size_t)b * N * D + h * HD;
  const int kstride = BWD_NW * gridDim.y, kfirst = blockIdx.y * BWD_NW + wave;
  // the k / v fragments of the wave's first tile pair are requested before the staging traffic (keys clamped like the queries above)
  s16x8_t kf[2][2], vf[2][2];
  auto kfetch = [&](int ka) {
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int kc = min((ka + x * kstride) * 16 + l15, N - 1);
      const bf16_t* kp = base + D + (size_t)kc * ld + g * 8;
      const bf16_t* vp = base + 2 * D + (size_t)kc * ld + g * 8;
      kf[x][0] = ld16(kp); kf[x][1] = ld16(kp + 32); vf[x][0] = ld16(vp); vf[x][1] = ld16(vp + 32);
    }
  };
  kfetch(kfirst);
  stage_transposed_perm<NP, BWD_NT>(Qt, base, ld, N, tid);
  stage_transposed_perm<NP, BWD_NT>(dOt, dobase, D, N, tid);
  stage_rows_swz<NP, BWD_NT>(Qs, base, ld, N, tid);
  stage_rows_swz<NP, BWD_NT>(dOs, dobase, D, N, tid);
  const int kc_ = g ^ ((l15 >> 1) & 7);
  const int kof0 = l15 * HD + (kc_ << 3), kof1 = l15 * HD + ((kc_ ^ 4) << 3);
  const int vof = (threadIdx.x & 15) * TP + (((threadIdx.x >> 4 & 3) ^ swz4(threadIdx.x & 15)) << 3);
  for (int i = tid; i < NP; i += BWD_NT) {
    lse_s[i] = i < N ? lse[((size_t)b * H + h) * N + i] * LOG2E : INFINITY;
    float dl = 0.f;
    if (i < N) {
      const bf16_t* dop = dobase + (size_t)i * D;
      const bf16_t* op = o_fwd + ((size_t)b * N + i) * D + h * HD;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const s16x8_t a = ld16(dop + c * 8), o8 = ld16(op + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) dl += bf2f((bf16_t)a[j]) * bf2f((bf16_t)o8[j]);
      }
    }
    dl_s[i] = dl;
  }
  __syncthreads();
  DBG_T(1);
  const float sc2 = scale * LOG2E;
  const int klen = (VAR && av.key_len) ? __builtin_amdgcn_readfirstlane(av.key_len[b]) : N;
  for (int ka = kfirst; ka * 16 < N; ka += 2 * kstride) {
    int key[2];
    s16x8_t k0[2], k1[2], v0[2], v1[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) { key[x] = (ka + x * kstride) * 16 + l15; k0[x] = kf[x][0]; k1[x] = kf[x][1]; v0[x] = vf[x][0]; v1[x] = vf[x][1]; }
    if ((ka + 2 * kstride) * 16 < N) kfetch(ka + 2 * kstride);
    f32x4_t dv[2][4], dk[2][4];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) { dv[x][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dk[x][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
    for (int u = 0; u < NKT / 2; ++u) {
      float pp[2][2][4], ds[2][2][4];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int ro = (2 * u + e) * 16 * HD;
        const s16x8_t cq0 = ld16(Qs + ro + kof0), cq1 = ld16(Qs + ro + kof1), cd0 = ld16(dOs + ro + kof0), cd1 = ld16(dOs + ro + kof1);
        const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(lse_s + (2 * u + e) * 16 + g * 4);      // result rows = queries 4 g .. 4 g + 3 of the tile
        const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(dl_s + (2 * u + e) * 16 + g * 4);
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          s = mfma16(cq0, k0[x], s);            // a-operand rows = queries, b-operand cols = keys
          s = mfma16(cq1, k1[x], s);
          dp = mfma16(cd0, v0[x], dp);
          dp = mfma16(cd1, v1[x], dp);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // padded queries: lse_s = +inf -> p = 0 exactly.  Padded KEY columns (this lane's key >= N) may hold anything: a column
            // of P / dS only feeds the dK / dV rows of that key, which are never stored.
            float p = fast_exp2(s[r] * sc2 - l4[r]);
            float pk = p, dpv = dp[r];
            if (VAR) {                              // masked keys (rows [klen, N) ARE stored: padded positions have dK = dV = 0)
              const int qq = (2 * u + e) * 16 + g * 4 + r;
              p = key[x] < klen ? p : 0.f;
              pk = p;
              if (av.drop_thresh) {
                const bool keep = drop_keep_attn((uint32_t)blockIdx.x * (uint32_t)N + (uint32_t)qq, (uint32_t)N, (uint32_t)min(key[x], N - 1), av.drop_key,
                                                 av.drop_thresh);
                pk = keep ? p * av.drop_scale : 0.f;
                dpv = keep ? dpv * av.drop_scale : 0.f;
              }
            }
            pp[x][e][r] = pk;
            ds[x][e][r] = p * (dpv - d4[r]) * scale;
          }
        }
      }
      const s16x8_t pba = pack8(pp[0][0], pp[0][1]), pbb = pack8(pp[1][0], pp[1][1]);
      const s16x8_t dsa = pack8(ds[0][0], ds[0][1]), dsb = pack8(ds[1][0], ds[1][1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int off = dt * 16 * TP + 32 * u + vof;
        const s16x8_t dot = ld16(dOt + off), qt = ld16(Qt + off);
        dv[0][dt] = mfma16(dot, pba, dv[0][dt]);
        dk[0][dt] = mfma16(qt, dsa, dk[0][dt]);
        dv[1][dt] = mfma16(dot, pbb, dv[1][dt]);
        dk[1][dt] = mfma16(qt, dsb, dk[1][dt]);
      }
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      if (key[x] < N) {
        bf16_t* dkp = dqkv + ((size_t)b * N + key[x]) * ld + D + h * HD + g * 4;
        bf16_t* dvp = dkp + D;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          uint2 a = {pack_bf2(dk[x][dt][0], dk[x][dt][1]), pack_bf2(dk[x][dt][2], dk[x][dt][3])};
          uint2 c = {pack_bf2(dv[x][dt][0], dv[x][dt][1]), pack_bf2(dv[x][dt][2], dv[x][dt][3])};
          *reinterpret_cast<uint2*>(dkp + dt * 16) = a;
          *reinterpret_cast<uint2*>(dvp + dt * 16) = c;
        }
      }
    }
  }
}

template <int NKT, bool VAR, bool VG>
__global__ __launch_bounds__(BWD_NT, 2) void attn_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o_fwd,
                                                      const bf16_t* __restrict__ d_out, const float* __restrict__ lse,
                                                      bf16_t* __restrict__ dqkv, float* __restrict__ delta, int N, int H, float scale, AttnVar av) {
  DBG_T(0);
  if constexpr (!VG) {                     // K, V, Q, dO images in LDS: two tiles per wave
    if (blockIdx.z == 0) attn_bwd_dq_pair<NKT, VAR>(qkv, o_fwd, d_out, lse, dqkv, delta, N, H, scale, av);
    else attn_bwd_dkv_pair<NKT, VAR>(qkv, o_fwd, d_out, lse, dqkv, N, H, scale, av);
  } else {                                 // N > 288: the V / (q, dO) row fragments come from L2; the dQ role still walks two query tiles per wave
    if (blockIdx.z == 0) attn_bwd_dq_pair<NKT, VAR, true>(qkv, o_fwd, d_out, lse, dqkv, delta, N, H, scale, av);
    else attn_bwd_dkv_body<NKT, VAR, VG>(qkv, o_fwd, d_out, lse, dqkv, N, H, scale, av);
  }
  __syncthreads();
  DBG_T(2);
}

template <typename F>
int dispatch_nkt(int N, F&& f) {
  const int nkt = 2 * ((N + 31) / 32);
  if (nkt <= 2) return f(std::integral_constant<int, 2>());
  if (nkt <= 8) return f(std::integral_constant<int, 8>());
  if (nkt <= 14) return f(std::integral_constant<int, 14>());
  if (nkt <= 18) return f(std::integral_constant<int, 18>());
  if (nkt <= 32) return f(std::integral_constant<int, 32>());
  return SR_EINVAL;
}

template <bool VAR>
int attn_fwd_launch(const void* qkv, void* out, float* lse, int B, int N, int H, float scale, AttnVar av, void* stream) {
  if (B <= 0 || N <= 0 || H <= 0 || N > 512 || (long)B * H * N * N >= (1L << 32)) return SR_EINVAL;
  return dispatch_nkt(N, [&](auto nk) -> int {
    constexpr int NKT = decltype(nk)::value, NP = NKT * 16;
    const size_t sm = (size_t)NP * HD * 2 + (size_t)64 * vt_pitch(NP) * 2;
    if constexpr (NKT == 32) {                // 288 < N <= 512: two query tiles per wave through 128-key blocks (attn_fwd_pair_kernel)
      static const bool whole_row = SR_TUNE_ENV("SRHIP_ATTN_FWD_WHOLE_ROW") != nullptr;   // tuning build: the whole-row kernel at every length (A/B)
      if (!whole_row) {
        auto kp = attn_fwd_pair_kernel<VAR>;
        (void)hipFuncSetAttribute((const void*)kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        SR_LAUNCH(kp, dim3(B * H), dim3(FWD_NT), sm, (hipStream_t)stream, (const bf16_t*)qkv, (bf16_t*)out, lse, N, H, scale, av);
        SR_CHECK_LAUNCH();
        return SR_OK;
      }
    }
    auto kern = attn_fwd_kernel<NKT, VAR>;
    if (sm > 48 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    SR_LAUNCH(kern, dim3(B * H), dim3(FWD_NT), sm, (hipStream_t)stream, (const bf16_t*)qkv, (bf16_t*)out, lse, N, H, scale, av);
    SR_CHECK_LAUNCH();
    return SR_OK;
  });
}

template <bool VAR>
int attn_bwd_launch(const void* qkv, const void* out, const void* d_out, const float* lse, void* dqkv, float* delta_ws, int B, int N, int H,
                    float scale, AttnVar av, void* stream) {
  if (B <= 0 || N <= 0 || H <= 0 || N > 512 || !lse || !delta_ws || (long)B * H * N * N >= (1L << 32)) return SR_EINVAL;
  return dispatch_nkt(N, [&](auto nk) -> int {
    constexpr int NKT = decltype(nk)::value, NP = NKT * 16;
    constexpr bool VG = NKT > 18;           // K + V + K^T images exceed the LDS: V fragments from L2
    const size_t sm1 = (size_t)(VG ? 1 : 2) * NP * HD * 2 + (size_t)64 * vt_pitch(NP) * 2 + (size_t)2 * NP * 4;
    const size_t sm2 = (size_t)2 * 64 * vt_pitch(NP) * 2 + (VG ? 0 : (size_t)2 * NP * HD * 2) + (size_t)2 * NP * 4;
    if (sm1 > 160 * 1024 || sm2 > 160 * 1024) return SR_EINVAL;
    auto kern = attn_bwd_kernel<NKT, VAR, VG>;
    const size_t sm = sm1 > sm2 ? sm1 : sm2;
    if (sm > 48 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    // split query / key tiles over extra workgroups until the grid covers the chip (each split re-stages K/V or Q/dO)
    const int nt16 = (N + 15) / 16;
    // (measured at the reference batch, 96 (image, head) pairs x 2 roles: no split 1087 img/s, split 2 1072, split 3 1065 while the deferred
    // rows share the chip -- every split re-stages K / V / Q / dO; alone on the chip (K = 0 regime) split 3 wins by 1 %)
    int split = 1;
    while (B * H * split * 2 < 128 && split * BWD_NW < nt16) ++split;
    static const char* force = SR_TUNE_ENV("SRHIP_ATTN_BWD_SPLIT");
    if (force && atoi(force) > 0) split = atoi(force);
    SR_LAUNCH(kern, dim3(B * H, split, 2), dim3(BWD_NT), sm, (hipStream_t)stream, (const bf16_t*)qkv, (const bf16_t*)out,
                       (const bf16_t*)d_out, lse, (bf16_t*)dqkv, delta_ws, N, H, scale, av);
    SR_CHECK_LAUNCH();
    return SR_OK;
  });
}

}  // namespace

#ifdef SRHIP_TUNING
extern "C" int srhip_attn_debug(long long* out_host, int n) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(srhip_attn_dbg), (size_t)n * sizeof(long long)) == hipSuccess ? SR_OK : SR_EINVAL;
}
#endif
extern "C" int srhip_attn_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, float scale, void* stream) {
  return attn_fwd_launch<false>(qkv, out, lse, B, N, H, scale, AttnVar{nullptr, 0u, 0u, 1.0f}, stream);
}
extern "C" int srhip_attn_bwd(const void* qkv, const void* out, const void* d_out, const float* lse, void* dqkv,
                              float* delta_ws, int B, int N, int H, float scale, void* stream) {
  return attn_bwd_launch<false>(qkv, out, d_out, lse, dqkv, delta_ws, B, N, H, scale, AttnVar{nullptr, 0u, 0u, 1.0f}, stream);
}
extern "C" int srhip_attn_masked_fwd(const void* qkv, void* out, float* lse, const int* key_len, int B, int N, int H, float scale,
                                     unsigned drop_key, unsigned drop_thresh, float drop_scale, void* stream) {
  return attn_fwd_launch<true>(qkv, out, lse, B, N, H, scale, AttnVar{key_len, drop_key, drop_thresh, drop_scale}, stream);
}
extern "C" int srhip_attn_masked_bwd(const void* qkv, const void* out, const void* d_out, const float* lse, void* dqkv, float* delta_ws,
                                     const int* key_len, int B, int N, int H, float scale, unsigned drop_key, unsigned drop_thresh,
                                     float drop_scale, void* stream) {
  return attn_bwd_launch<true>(qkv, out, d_out, lse, dqkv, delta_ws, B, N, H, scale, AttnVar{key_len, drop_key, drop_thresh, drop_scale}, stream);
}
