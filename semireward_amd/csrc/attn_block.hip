// Fused qkv projection + attention of a ViT block for rows WITHOUT a backward (gfx950, embed_dim 384 = 6 heads of 64):
//     ao = softmax( q k^T * scale ) v      with   [q | k | v] = xn W_qkv^T + b_qkv,   xn = LayerNorm(x) (bf16, from srhip_layernorm_fwd)
// reference: semilearn/nets/vit/vit.py:93-104 (Attention.forward: qkv Linear + reshape, scaled dot product, softmax, attn @ v) on the
// output of norm1 (:163).
//
// Why: as separate launches the [M, 3D] qkv activation makes a round trip through HBM (157 MB written + read per layer at M = 51400: it
// bounds the attention kernel at 60 % of its HBM floor and makes the qkv product store-bound).  Of the 216 images of a SemiReward step 200
// are forwarded without a backward: for them HBM sees the normalised rows in (bf16) and the attention output (bf16).
//
// Structure: ONE workgroup (8 waves) = ONE image, heads in sequence.
//   * every wave owns two 16-token tiles (tokens 16 w .. and 128 + 16 w ..) end to end: their normalised rows sit in registers as bf16
//     MFMA fragments (96 VGPRs), which are BOTH the B operand of  K^T / Q^T tiles = W . xn^T  and the A operand of  V tile = xn . W^T.
//   * the head's weights (3 x [64, 384] bf16 = 144 KB) stream through an 8-slot LDS-DMA ring of [64 features x 64 k] tiles (8 KB, one
//     buffer_load ... lds per wave and stage), chunk-swizzled so every ds_read_b128 fragment read is bank-conflict free; for K and Q the
//     DMA permutes the feature rows inside each 32-block so that two accumulator tiles of a lane hold 8 consecutive features of its token:
//     K leaves the registers as one 16-byte row chunk of the LDS image, Q IS the B fragment of the score product -- no shuffles.
//   * K image [key][64] and V^T image [64][key'] of the head live in LDS in the layouts of attention.hip (conflict-free, key-permuted V^T
//     so that the probability registers feed the PV product directly); scores stay in registers, four key tiles at a time with a running
//     maximum.
//   * N = 257 = 16 tiles + ONE token.  Carrying that token through the projection loops (a broadcast operand tile on two of the eight
//     waves per matrix) put three differently specialised copies of every loop into the kernel and 87 spilled registers whose reloads
//     (s_waitcnt vmcnt(0)) drained the DMA ring; its q | k | v row is instead computed by one small GEMM over the B last tokens before
//     this launch (qkv_extra) and enters here as LDS rows: key / value 256 of every head and one extra query tile on wave h % 8.
//   * LayerNorm stays a separate launch: fused in (fp32 rows -> statistics -> fragments) it needs 96 + 48 + 48 registers at its peak
//     next to everything else that is live and spilled 240 registers; measured, not guessed.
// Rounding points are the ones of the unfused path (xn bf16, q / k / v bf16, probabilities bf16, fp32 accumulation and softmax).
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "../../include/srhip.h"
#include "common.h"

namespace {

constexpr int HD = 64;                    // head dim
constexpr int DM = 384;                   // embed dim
constexpr int NH = DM / HD;               // 6 heads
constexpr int KS = DM / 32;               // 12 k-steps of 32
constexpr int SK = 64;                    // k per ring stage
constexpr int ST_EL = 64 * SK;            // stage tile: 64 feature rows x 64 k = 4096 bf16 = 8 KiB
constexpr int NS = 8;                     // ring slots
constexpr int GS = 2;                     // stages per group sync
constexpr int NG = NS / GS;               // 4 groups: 1 being read, 1 cooling down, 2 in flight
constexpr int SPM = DM / SK;              // 6 stages per matrix
constexpr int SPH = 3 * SPM;              // 18 stages per head (K, V, Q)
constexpr int NSTG = NH * SPH;            // 108 stages per image
constexpr float LOG2E = 1.4426950408889634f;

typedef __attribute__((address_space(3))) void lds_void;
template <int N_>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }
// workgroup barrier that also publishes this wave's ds_writes (K / V^T images) -- without draining the LDS-DMA loads in flight,
// which __syncthreads() (s_waitcnt vmcnt(0)) would
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// lane id recomputed where it is used (volatile: the compiler must not keep a copy -- and the addresses derived from it -- alive across the
// projection loops, where every register counts: the spilled copies were reloaded behind s_waitcnt vmcnt(0), draining the DMA ring)
__device__ __forceinline__ int fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// max of three without the canonicalising v_max_f32 x, x, x that fmaxf() / __builtin_fmaxf() put in front of every operand (an MFMA result
// is not known to be canonical to the compiler): 30 instead of 8 instructions per chunk of 16 scores in the attention phase.  The median of
// {a, b, +inf} is max(a, b) and v_med3_f32 takes its operands as they are.  (Inline asm is not an option: the MFMA -> VALU read hazard is
// software-managed and the hazard recogniser does not look inside asm statements -- stale accumulator reads.)
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
  return __builtin_amdgcn_fmed3f(__builtin_amdgcn_fmed3f(a, b, INFINITY), c, INFINITY);
}
// all-reduce over the four 16-lane rows of a wave (the four key groups of one query) with the gfx950 row-swap instructions instead of two
// ds_bpermute round trips: permlane16_swap(x, x) = {rows 0 0 2 2, rows 1 1 3 3}, permlane32_swap(x, x) = {lo lo, hi hi} (tools/pl_probe.hip)
__device__ __forceinline__ float rows_max(float x) {
  const unsigned u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned v = __float_as_uint(__builtin_amdgcn_fmed3f(__uint_as_float(a[0]), __uint_as_float(a[1]), INFINITY));
  const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return __builtin_amdgcn_fmed3f(__uint_as_float(b[0]), __uint_as_float(b[1]), INFINITY);
}
__device__ __forceinline__ float rows_sum(float x) {
  const unsigned u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned v = __float_as_uint(__uint_as_float(a[0]) + __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

__device__ __forceinline__ f32x4_t mfma16(u32x4_t a, u32x4_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x4_t ldf(const bf16_t* p) { return *reinterpret_cast<const u32x4_t*>(p); }
__device__ __forceinline__ u32x4_t pack8v(f32x4_t lo, f32x4_t hi) {
  return u32x4_t{pack_bf2(lo[0], lo[1]), pack_bf2(lo[2], lo[3]), pack_bf2(hi[0], hi[1]), pack_bf2(hi[2], hi[3])};
}
constexpr int vt_pitch(int NP) { return ((NP / 2) % 64 == 16 || (NP / 2) % 64 == 48) ? NP : NP + 32; }     // as in attention.hip
__device__ __forceinline__ int swz4(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }
constexpr int nkt_of(int N) { return 2 * ((N + 31) / 32); }

template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

struct AbArgs {
  const bf16_t* xnb;       // [B * N, 384] LayerNorm output in bf16 (srhip_layernorm_fwd)
  const bf16_t* qx;        // [B, 1152] q | k | v of token 256 of every image (N = 257 only)
  const float* bqkv;
  const bf16_t* W;         // [1152, 384] bf16: q | k | v rows, head-major inside each third (vit.py:93-98)
  bf16_t* out;             // [B * N, 384] attention output, heads concatenated (vit.py:104 transpose + reshape)
  const float* osc;        // [B] factor on the image's output rows (DropPath of the branch, vit.py:163), applied before the bf16 rounding; or NULL
  float scale;
  int B;
};

// DBG (tuning builds only, SRHIP_AB_DEBUG): 1 = no attention phase, 2 = no projection MFMAs, 4 = no DMA / no vmcnt waits, 8 = no pass for the
// 257th query, 16 = no v_exp (wrong results by design)
template <int N, int DBG>
__global__ __launch_bounds__(512, 2) void attn_block_kernel(AbArgs a) {
  constexpr int NKT = nkt_of(N), NP = NKT * 16, TP = vt_pitch(NP);
  constexpr int NQT = (N + 15) / 16;
  constexpr bool EXTRA = NQT == 17;                  // N = 257: tile 16 holds exactly one token
  static_assert(NQT <= 16 || (EXTRA && N == 257), "two token tiles per wave (+ one extra token)");
  constexpr int NKT_LO = (N - 1) / 16;               // key tiles below this index are complete
  extern __shared__ __attribute__((aligned(16))) bf16_t sm[];
  bf16_t* ring = sm;                                 // [NS][64][64]
  bf16_t* Ks = ring + NS * ST_EL;                    // [NP][64]   chunk c of row r at c ^ ((r >> 1) & 7)
  bf16_t* Vt = Ks + NP * HD;                         // [64][TP]   key-permuted, chunk ^= swz4(d)
  bf16_t* sx = Vt + HD * TP;                         // [1152] q | k | v row of the extra token (N = 257)
  float* sbias = reinterpret_cast<float*>(sx + 3 * DM);   // [1152]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int img = blockIdx.x;
  const float osc = a.osc ? a.osc[img] : 1.0f;       // uniform: one scalar load
  // heads of this workgroup: the grid is (images, head groups) -- a launch of few images gives every image to 2 or 3 workgroups (each
  // re-reads the image's normalised rows: 197 KB from L2) so that its latency is that of 3 or 2 heads instead of 6
  const int hpw = NH / gridDim.y, hbeg = blockIdx.y * hpw, hend = hbeg + hpw;
  const size_t row0 = (size_t)img * N;

  // ---- ring producer: LDS row rho = 8 wave + (lane >> 3), chunk position pc = lane & 7 holds source chunk pc ^ ((rho >> 1) & 7) of
  // feature row perm(rho) (K, Q: inside every 32-block, LDS rows 0-15 = features 8 (i >> 2) + (i & 3), rows 16-31 = the same + 4) or rho (V)
  const int rho = 8 * wave + (lane >> 3);
  const int srcc = (lane & 7) ^ ((rho >> 1) & 7);
  const int fperm = (rho & 32) + 8 * ((rho & 15) >> 2) + 4 * ((rho >> 4) & 1) + (rho & 3);
  const int vo_kq = (fperm * DM + srcc * 8) * 2, vo_v = (rho * DM + srcc * 8) * 2;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.W), 0, 3 * DM * DM * 2, 0x00020000);
  constexpr int OOB = 0x7ffffff0;
  // stage r of head h (r = 0 .. 17 compile-time: 6 K stages, 6 V, 6 Q; r >= 18 rolls into head h + 1): one DMA instruction per wave;
  // beyond the last head it is a no-op that still counts in vmcnt (uniform wait arithmetic)
  auto issue = [&](auto rc, int h) __attribute__((always_inline)) {
    constexpr int r0 = decltype(rc)::value, r = r0 % SPH, m = r / SPM, kc = r % SPM;      // m: 0 K, 1 V, 2 Q
    const int hh = h + r0 / SPH;
    const bool valid = hh < hend;
    constexpr int brow = m == 0 ? DM : m == 1 ? 2 * DM : 0;
    lds_void* dst = (lds_void*)(ring + (r0 % NS >= 0 ? ((hh * SPH + r) & (NS - 1)) : 0) * ST_EL + wave * 8 * SK);
    if constexpr ((DBG & 4) != 0) return;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, dst, 16, valid ? (m == 1 ? vo_v : vo_kq) : OOB,
                                             valid ? ((brow + hh * HD) * DM + kc * SK) * 2 : 0, 0, 0);
  };
  // the first groups travel while the prologue normalises the tokens
  issue(std::integral_constant<int, 0>{}, hbeg); issue(std::integral_constant<int, 1>{}, hbeg);
  issue(std::integral_constant<int, 2>{}, hbeg); issue(std::integral_constant<int, 3>{}, hbeg);
  static_assert((NG - 2) * GS == 4, "prologue issues the first two groups");

  for (int i = tid; i < 3 * DM; i += 512) sbias[i] = a.bqkv[i];
  if (EXTRA && tid < 3 * DM / 8) reinterpret_cast<u32x4_t*>(sx)[tid] = ldf(a.qx + (size_t)img * 3 * DM + 8 * tid);
  // pad rows of the K image and the whole V^T image start finite (masked scores multiply them by exactly 0)
  for (int i = tid; i < (NP - N) * HD / 8; i += 512) reinterpret_cast<u32x4_t*>(Ks + N * HD)[i] = u32x4_t{0u, 0u, 0u, 0u};
  for (int i = tid; i < HD * TP / 8; i += 512) reinterpret_cast<u32x4_t*>(Vt)[i] = u32x4_t{0u, 0u, 0u, 0u};
  __syncthreads();

  // ---- the wave's token tiles as MFMA fragments: lane holds xn[token l15][32 k + 8 g .. + 7], k = 0 .. 11 (12 sixteen-byte loads per tile)
  u32x4_t xn[2][KS];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const bool have = tt == 0 || NQT - (EXTRA ? 1 : 0) == 16 || wave + 8 < NQT - (EXTRA ? 1 : 0);
    const int tok = min(16 * (wave + 8 * tt) + l15, N - 1);
    const bf16_t* xr = a.xnb + (row0 + tok) * DM + 8 * g;
#pragma unroll
    for (int k = 0; k < KS; ++k) xn[tt][k] = have ? ldf(xr + 32 * k) : u32x4_t{0u, 0u, 0u, 0u};
  }

  // ---- fragment offsets (elements) inside a ring tile: row 16 t + l15, k-step s of the stage: chunk (4 s + g) ^ ((l15 >> 1) & 7)
  const int kc_ = g ^ ((l15 >> 1) & 7);
  const int fo0 = l15 * SK + (kc_ << 3), fo1 = l15 * SK + ((kc_ ^ 4) << 3);
  // attention operand offsets (attention.hip): K rows / V^T rows
  const int kof0 = l15 * HD + (kc_ << 3), kof1 = l15 * HD + ((kc_ ^ 4) << 3);
  const int vof = l15 * TP + ((g ^ swz4(l15)) << 3);
  const float sc2 = a.scale * LOG2E;

  // group sync before the first read of the group that starts at stage r of head h: own DMA parts of the group have landed (the younger
  // group may still be in flight), barrier (everybody's parts have, and everybody is done with the group consumed two syncs ago), refill it
  auto sync_group = [&](auto rc, int h) __attribute__((always_inline)) {
    constexpr int r = decltype(rc)::value;
    if constexpr ((DBG & 4) == 0) wait_vm<GS*(NG - 3)>();
    __builtin_amdgcn_s_barrier();
    if constexpr ((DBG & 32) == 0) {
      issue(std::integral_constant<int, r + (NG - 2) * GS>{}, h);
      issue(std::integral_constant<int, r + (NG - 2) * GS + 1>{}, h);
    }
    static_assert(GS == 2, "two stages per group");
  };
  // DBG & 32 ("spread"): the refills are issued one per stage behind the stage's first unit of MFMAs ("stage s asks for stage s + 4") instead of
  // two in a row behind the group barrier, where all eight waves -- both waves of every SIMD -- block in the DMA issue together (mlp_fused.hip)
  constexpr bool ALL2 = NQT - (EXTRA ? 1 : 0) == 16;          // every wave has two full tiles (N = 257)
  const bool tile1 = ALL2 || wave + 8 < NQT - (EXTRA ? 1 : 0);   // wave-uniform: the second tile exists

  u32x4_t qf[2][2];                                    // Q fragments of the wave's two tiles: [tile][d half]
  // One projection (M: 0 K, 1 V, 2 Q) of head h for the wave's two token tiles.  Every index inside is a compile-time constant: the loop body is
  // 16 ds_read_b128 + 16 (18) MFMAs per stage without a branch.  Both token tiles are always multiplied (a wave without a second tile
  // carries zeros: it would wait at the group barrier otherwise).
  auto project = [&](auto mc, int h) __attribute__((always_inline)) {
    constexpr int M = decltype(mc)::value;
    f32x4_t acc[2][4];
    const float* bb = sbias + (M == 0 ? DM : M == 1 ? 2 * DM : 0) + h * HD;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4_t b4;
      if constexpr (M == 1) { const float b = bb[16 * t + l15]; b4 = f32x4_t{b, b, b, b}; }          // V: D[token][feature l15]
      else b4 = *reinterpret_cast<const f32x4_t*>(bb + 32 * (t >> 1) + 8 * g + 4 * (t & 1));         // K, Q: D[feature 4g + r][token]
      acc[0][t] = b4; acc[1][t] = b4;
    }
    // Software pipeline inside the wave in HALF k-steps (unit u = 2 j + hf: feature tiles 2 hf, 2 hf + 1 of k-step j): the two fragments of
    // unit u + 1 are requested before the four MFMAs of unit u (two fragment sets of two = 16 registers); at a group boundary the sync
    // (wait for the DMA, barrier, refill) and the first reads of the new group come BEFORE the MFMAs of the old group's last unit, whose
    // fragments are already in registers.  The extra token's fragment rides with the unit that multiplies it (one buffer suffices: its
    // next read is issued after that unit's MFMAs).
    u32x4_t fa[2][2];
    auto rd = [&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value, j = u >> 1, hf = u & 1, stg = j >> 1, sh = j & 1;
      const bf16_t* tile = ring + ((h * SPH + M * SPM + stg) & (NS - 1)) * ST_EL;
      fa[u & 1][0] = ldf(tile + (2 * hf) * 16 * SK + (sh ? fo1 : fo0));
      fa[u & 1][1] = ldf(tile + (2 * hf + 1) * 16 * SK + (sh ? fo1 : fo0));
    };
    auto mm = [&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value, j = u >> 1, hf = u & 1;
      u32x4_t(&fr)[2] = fa[u & 1];
      if constexpr ((DBG & 2) != 0) {
        asm volatile("" ::"v"(fr[0]), "v"(fr[1]));
      } else if constexpr (M != 1) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[0][2 * hf + t] = mfma16(fr[t], xn[0][j], acc[0][2 * hf + t]);
          acc[1][2 * hf + t] = mfma16(fr[t], xn[1][j], acc[1][2 * hf + t]);
        }
      } else {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[0][2 * hf + t] = mfma16(xn[0][j], fr[t], acc[0][2 * hf + t]);
          acc[1][2 * hf + t] = mfma16(xn[1][j], fr[t], acc[1][2 * hf + t]);
        }
      }
    };
    sync_group(std::integral_constant<int, M * SPM>{}, h);
    rd(std::integral_constant<int, 0>{});
    static_for<2 * KS>([&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value;
      if constexpr (u + 1 < 2 * KS) {
        if constexpr ((u + 1) % (4 * GS) == 0) sync_group(std::integral_constant<int, M * SPM + (u + 1) / 4>{}, h);
        rd(std::integral_constant<int, u + 1>{});
      }
      __builtin_amdgcn_sched_barrier(0);
      mm(uc);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((DBG & 32) != 0 && u % 4 == 0) issue(std::integral_constant<int, M * SPM + u / 4 + (NG - 2) * GS>{}, h);
    });
    // ---- results of the matrix leave the accumulators
    if constexpr (M == 0) {                            // K image: row = key, 16-byte chunk 4 blk + g holds features 32 blk + 8 g .. + 7
      // (store addresses from a lane id read HERE: computed from l15 / g they are head-invariant, get hoisted out of the head loop and held
      // in four registers across the attention phase -- one of them spilled, and its reload drained the weight ring once per head)
      const int lnk = fresh_lane(), l15k = lnk & 15, gk = lnk >> 4;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        if (tt == 1 && !tile1) break;
        const int key = 16 * (wave + 8 * tt) + l15k;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
          *reinterpret_cast<u32x4_t*>(Ks + key * HD + (((4 * blk + gk) ^ ((key >> 1) & 7)) << 3)) = pack8v(acc[tt][2 * blk], acc[tt][2 * blk + 1]);
      }
      if (EXTRA && wave == 0) {                        // key 256 of this head from the pre-computed row ((256 >> 1) & 7 = 0: chunks in place)
        const int ln_ = fresh_lane();
        if (ln_ < 8) *reinterpret_cast<u32x4_t*>(Ks + 256 * HD + (ln_ << 3)) = ldf(sx + DM + h * HD + 8 * ln_);
      }
    } else if constexpr (M == 1) {                     // V^T image: lane holds keys 4 g .. 4 g + 3 of its tile for feature 16 t + l15
      const int lnv = fresh_lane(), l15v = lnv & 15, gv = lnv >> 4;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        if (tt == 1 && !tile1) break;
        const int T = wave + 8 * tt, u = T >> 1, e = T & 1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int d = 16 * t + l15v;
          *reinterpret_cast<u32x2_t*>(Vt + d * TP + 32 * u + ((gv ^ swz4(d)) << 3) + 4 * e) =
              u32x2_t{pack_bf2(acc[tt][t][0], acc[tt][t][1]), pack_bf2(acc[tt][t][2], acc[tt][t][3])};
        }
      }
      if (EXTRA && wave == 1) {                        // value 256: tile 16 -> u = 8, e = 0, position 32 u + ((0 ^ swz4(d)) << 3), d = lane
        const int ln_ = fresh_lane();
        Vt[ln_ * TP + 256 + (swz4(ln_) << 3)] = sx[2 * DM + h * HD + ln_];
      }
    } else {                                           // Q: the accumulators ARE the B fragments of S^T = K Q^T
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        qf[tt][0] = pack8v(acc[tt][0], acc[tt][1]);
        qf[tt][1] = pack8v(acc[tt][2], acc[tt][3]);
      }
    }
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  using C2 = std::integral_constant<int, 2>;

#pragma unroll 1
  for (int h = hbeg; h < hend; ++h) {
    // =============== projections of this head: K, V, Q (6 stages each) ===============
    project(C0{}, h);
    project(C1{}, h);
    project(C2{}, h);
    lds_barrier();                                     // K / V^T images of this head are complete

    // =============== attention of this head: the wave's query tiles against all keys ===============
    const int nq = (DBG & 1) ? 0 : (tile1 ? 2 : 1) + ((EXTRA && (DBG & 8) == 0 && (h & 7) == wave) ? 1 : 0);
#pragma unroll 1
    for (int qi = 0; qi < nq; ++qi) {
      const bool ext = EXTRA && qi == (tile1 ? 2 : 1);
      // the tile's Q fragments are always qf[0] (the next tile's move up at the end of the iteration): no selected copy held next to both
      int qtok;
      if (ext) { qf[0][0] = ldf(sx + h * HD + 8 * g); qf[0][1] = ldf(sx + h * HD + 32 + 8 * g); qtok = 256 + l15; }
      else qtok = 16 * (wave + 8 * qi) + l15;
      const u32x4_t q0 = qf[0][0], q1 = qf[0][1];
      // Scores in chunks of 4 key tiles with a running maximum (16 score registers instead of 72 next to the 96 of the token fragments):
      // p = 2^(s c - m c) against the maximum so far; partial output and row sum are rescaled by 2^((m_old - m_new) c) whenever a later
      // chunk raises it.  Same rounding points as attention.hip (p bf16, fp32 sums).
      constexpr int CHT = 4, NCH = (NKT + CHT - 1) / CHT;
      f32x4_t o[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      // The phase is bound by the vector ALU, not by the matrix pipe (per lane and head 136 scores: fma + v_exp_f32 at quarter rate + sum +
      // max): everything around the exponential is kept to the instruction minimum -- v_max3_f32 without the canonicalising v_max x, x that
      // fmaxf() costs per operand, the scale-and-shift and the row sums as 2-wide packed fp32 operations (two partial sums per lane), and no
      // work at all on key positions that are padding at compile time (N = 257: one register of tile 16, nothing of tile 17).
      float mrun = -INFINITY;
      f32x2_t psum = {0.f, 0.f};
      static_for<NCH>([&](auto cc) __attribute__((always_inline)) {
        constexpr int ch = decltype(cc)::value, t0 = ch * CHT, nt = (NKT - t0 < CHT) ? NKT - t0 : CHT;
        static_assert(nt % 2 == 0, "the PV product consumes key-tile pairs");
        f32x4_t s[nt];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < nt; ++i) {
          const int t = t0 + i;
          const int nvalid = N - 16 * t;                // keys of this tile that exist (compile-time after unrolling)
          if (nvalid <= 0) continue;                    // all padding: no scores, p = 0 below
          f32x4_t c = {0.f, 0.f, 0.f, 0.f};
          c = mfma16(ldf(Ks + t * 16 * HD + kof0), q0, c);
          c = mfma16(ldf(Ks + t * 16 * HD + kof1), q1, c);
          if (nvalid >= 16) {
            mx = max3_raw(mx, c[0], c[1]);
            mx = max3_raw(mx, c[2], c[3]);
          } else {                                      // key 16 t + 4 g + r: register r can only be valid while r < nvalid
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (r < nvalid) { c[r] = (4 * g + r < nvalid) ? c[r] : -INFINITY; mx = max3_raw(mx, c[r], c[r]); }
            }
          }
          s[i] = c;
          if ((i & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        mx = rows_max(mx);
        const float mnew = max3_raw(mrun, mx, mx);      // (finite from the first chunk on: key 0 is never masked)
        const float nmx = -mnew * sc2;
        if constexpr (ch > 0) {
          const float corr = fast_exp2(fmaf(mrun, sc2, nmx));
          psum *= corr;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) o[dt] *= corr;
        }
        mrun = mnew;
        const f32x4_t sc4 = {sc2, sc2, sc2, sc2}, nm4 = {nmx, nmx, nmx, nmx};
#pragma unroll
        for (int i = 0; i < nt; ++i) {
          const int nvalid = N - 16 * (t0 + i);
          if (nvalid >= 16) {
            const f32x4_t e = __builtin_elementwise_fma(s[i], sc4, nm4);
            const f32x4_t p = (DBG & 16) ? e : f32x4_t{fast_exp2(e[0]), fast_exp2(e[1]), fast_exp2(e[2]), fast_exp2(e[3])};
            s[i] = p;
            psum += f32x2_t{p[0], p[1]};
            psum += f32x2_t{p[2], p[3]};
          } else {
            f32x4_t p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (r < nvalid) { p[r] = fast_exp2(fmaf(s[i][r], sc2, nmx)); psum[r & 1] += p[r]; }
            s[i] = p;
          }
        }
#pragma unroll
        for (int u = 0; u < nt / 2; ++u) {
          const u32x4_t pb = pack8v(s[2 * u], s[2 * u + 1]);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) o[dt] = mfma16(ldf(Vt + dt * 16 * TP + 32 * (t0 / 2 + u) + vof), pb, o[dt]);
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      float sum = psum[0] + psum[1];
      sum = rows_sum(sum);
      if (qtok < N && (!ext || l15 == 0)) {
        const float inv = osc / sum;
        bf16_t* op = a.out + (row0 + qtok) * DM + h * HD + g * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          uint2 v = {pack_bf2(o[dt][0] * inv, o[dt][1] * inv), pack_bf2(o[dt][2] * inv, o[dt][3] * inv)};
          *reinterpret_cast<uint2*>(op + dt * 16) = v;
        }
      }
      qf[0][0] = qf[1][0]; qf[0][1] = qf[1][1];
    }
    // (no barrier here: the next head's first image write comes after three group syncs, which every wave reaches after its attention)
  }
}

template <int N, int DBG = 0>
int launch(const AbArgs& a, hipStream_t s) {
  constexpr int NKT = nkt_of(N), NP = NKT * 16, TP = vt_pitch(NP);
  const size_t smem = (size_t)(NS * ST_EL + NP * HD + HD * TP + 3 * DM) * sizeof(bf16_t) + (size_t)(3 * DM) * sizeof(float);
  auto kern = attn_block_kernel<N, DBG>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  // head groups per image: as many (1, 2, 3 or 6) as keep the launch within one workgroup per CU (SRHIP_AB_SPLIT overrides)
  int split = 1;
  for (int c : {2, 3, 6}) if (a.B * c <= 256) split = c;
  if (const char* e = SR_TUNE_ENV("SRHIP_AB_SPLIT")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 3 || v == 6) split = v; }
  SR_LAUNCH(kern, dim3(a.B, split), dim3(512), smem, s, a);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

}  // namespace

extern "C" int srhip_attn_block_supported(int N, int D, int H) { return (D == DM && H == NH && (N == 257 || N == 197)) ? 1 : 0; }

extern "C" int srhip_attn_block_fused(const void* xn_bf16, const void* Wqkv, const float* bqkv, const void* qkv_extra, void* out,
                                      const float* out_scale, int B, int N, int D, int H, float scale, void* stream) {
  if (!xn_bf16 || !Wqkv || !bqkv || !out || B <= 0) return SR_EINVAL;
  if (!srhip_attn_block_supported(N, D, H)) return SR_EINVAL;
  if (N == 257 && !qkv_extra) return SR_EINVAL;
  if (((uintptr_t)xn_bf16 | (uintptr_t)Wqkv | (uintptr_t)out | (uintptr_t)bqkv | (uintptr_t)qkv_extra) & 15) return SR_EINVAL;
  AbArgs a;
  a.xnb = (const bf16_t*)xn_bf16; a.qx = (const bf16_t*)qkv_extra; a.bqkv = bqkv; a.W = (const bf16_t*)Wqkv; a.out = (bf16_t*)out;
  a.scale = scale; a.B = B; a.osc = out_scale;
  hipStream_t s = (hipStream_t)stream;
#ifdef SRHIP_TUNING
  switch (SR_TUNE_ENV("SRHIP_AB_DEBUG") ? atoi(SR_TUNE_ENV("SRHIP_AB_DEBUG")) : 0) {
    case 1: return N == 257 ? launch<257, 1>(a, s) : launch<197, 1>(a, s);
    case 2: return N == 257 ? launch<257, 2>(a, s) : launch<197, 2>(a, s);
    case 3: return N == 257 ? launch<257, 3>(a, s) : launch<197, 3>(a, s);
    case 4: return N == 257 ? launch<257, 4>(a, s) : launch<197, 4>(a, s);
    case 6: return N == 257 ? launch<257, 6>(a, s) : launch<197, 6>(a, s);
    case 7: return N == 257 ? launch<257, 7>(a, s) : launch<197, 7>(a, s);
    case 8: return N == 257 ? launch<257, 8>(a, s) : launch<197, 8>(a, s);
    case 16: return N == 257 ? launch<257, 16>(a, s) : launch<197, 16>(a, s);
    default: break;
  }
#endif
  // (A/B on one box: 41.5 -> 40.8 us per 105 images at N = 197, no difference at N = 257 -- unlike the fused MLP launch, where it is 5 %)
  static const int spread = SR_TUNE_ENV("SRHIP_ATTN_SPREAD") ? atoi(SR_TUNE_ENV("SRHIP_ATTN_SPREAD")) : 1;
  if (spread) return N == 257 ? launch<257, 32>(a, s) : launch<197, 32>(a, s);
  return N == 257 ? launch<257>(a, s) : launch<197>(a, s);
}
