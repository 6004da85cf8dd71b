// Attention projection + first residual + LayerNorm + fc1 + GELU + fc2 + second residual (+ the next block's norm1) of an inference row
// in ONE launch, as a PRODUCER / CONSUMER split of the workgroup (gfx950).  Same contract as srhip_mlp_fused_proj (mlp_fused.hip):
//     x1 = x + rs1 * (ao Wp^T + bp)                         reference: semilearn/nets/vit/vit.py:105-106 (Attention.proj), :163 (drop_path1)
//     y  = x1 + rs2 * (fc2(GELU(fc1(LayerNorm(x1)))) + b2)  reference: vit.py:69-75 (Mlp.forward), :165 (drop_path2)
//     ln_next = LayerNorm(y) with the next block's norm1     reference: vit.py:163 (norm1 of the following Block)
//
// Why a second structure.  mlp_fused_kernel gives every one of its 8 waves 16 rows end to end (LayerNorm'ed rows in registers, both products
// chained in registers).  With 16 rows per wave every 16x16x32 MFMA needs its own 1 KiB weight fragment from LDS: at the matrix-pipe peak that
// is exactly the 256 B/clk the LDS delivers, and the two waves of a SIMD run the same code in lockstep, so fragment reads, the GELU's
// vector-ALU work and the MFMAs of a pair add up instead of overlapping (measured: matrix pipe 27 % busy, profiles/r02_pmc_dominant_kernels.txt).
// Here a workgroup is 4 PRODUCER waves (P_i: fc1 + GELU of rows 32 i .. 32 i + 31) and 4 CONSUMER waves (C_i: projection, fc2, both
// LayerNorms, the epilogue, same rows), all on v_mfma_f32_32x32x16_bf16:
//   * 32 rows per wave: a 1 KiB weight fragment feeds a 32x32x16 product = twice the flops per LDS byte and per issued instruction;
//   * a SIMD hosts one P and one C wave (waves w and w + 4 share a SIMD): different instruction streams, so the GELU (vector ALU) of the
//     producer runs beside the consumer's MFMAs instead of beside its own partner's GELU;
//   * the hidden activation crosses from P_i to C_i through 2 x 8 KiB of LDS in MFMA B-fragment order: the producer's result registers 8 s .. 8 s + 7
//     of a 32-hidden tile ARE, after GELU and bf16 packing, the consumer's B fragment of k-step s (tile row i = 8 q + 4 b + t is fed with hidden
//     unit 16 (q / 2) + 8 b + 4 (q % 2) + t; tools/mfma32_probe.hip pins the 32x32x16 operand layout), so one ds_write_b128 / ds_read_b128 at
//     16 * lane on either side, conflict free.
// Weights arrive through a 12-slot ring of 8 KiB stages filled by LDS-DMA from a PACKED image (srhip_mlp_ps_pack): every stage is 8 fragments
// in consumption order, every fragment 1 KiB in lane order, so a DMA instruction reads 1 KiB of contiguous memory and a consumer lane reads its
// 16 bytes at 16 * lane (no swizzle, no bank conflict).  The attention output rows take the same road (one 8 KiB stage per 32 input columns).
//
// Schedule of one 128-row tile (group = 2 ring stages, one s_barrier per group, the DMA 5 groups ahead; the stream is issued by the
// producer waves during the projection and by the consumer waves during the MLP -- the role with spare issue slots in that phase):
//   projection   12 half-phases x [ao | Wp0] [Wp1 | Wp2]   C: 24 MFMAs per half-phase on x (the accumulators start at x + rs1 bp); P: idle
//   LayerNorm    C normalises x1 from its accumulators and hands the bf16 rows to P through the exchange buffers (6 rounds of 16 KiB)
//   MLP          2 nch + 2 half-phases x 3 groups [W1 (tile h, k-third) | W2 (tile h - 2, output third)]:
//                P: 24 MFMAs into tile h + GELU / pack / LDS write of tile h - 1;  C: 24 MFMAs on tile h - 2 (k-slice of 32 hidden units)
//   epilogue     C: y (fp32) and LayerNorm(y) (bf16) leave from the accumulator layout
#include <stdlib.h>

#include <utility>

#include "../../include/srhip.h"
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int PS_D = 384;                 // model width the kernel is built for
constexpr int FRB = 1024;                 // bytes of a fragment: 32 rows x 16 k bf16, lane l = (row l % 32, k 8 (l / 32) .. + 7)
constexpr int STB = 8 * FRB;              // bytes of a ring stage
constexpr int NSL = 12;                   // ring slots = 6 groups of 2 stages: one period of the schedule (2 MLP half-phases, 3 projection half-phases),
                                          // so every slot index in the loops below is a compile-time constant (ds_read offsets are immediates)
constexpr int NGR = NSL / 2;              // ring groups
constexpr int RING_B = NSL * STB;         // 96 KiB
constexpr int EXCH_B = 2 * STB;           // two hidden-tile buffers (4 row groups x 2 k-steps x 1 KiB each)
constexpr int PROJ_HP = PS_D / 32;        // 12 projection half-phases (32 input columns each)
constexpr int PROJ_G = 2 * PROJ_HP;       // 24 groups
constexpr int AHEAD = NGR - 1;            // groups of DMA issued ahead of the group being multiplied (5: 4 in flight + 1 landed)

template <int N_>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <class F, int... Is>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ f32x16_t mfma32(u32x4_t a, u32x4_t b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// sum over the two 32-lane halves of a wave (a row's 384 columns live in lanes j and j + 32)
__device__ __forceinline__ float halves_sum(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// hidden unit (inside its 32-unit tile) that fc1 tile row i computes: rows 8 s .. 8 s + 7 of a lane's result registers become k = 8 b .. 8 b + 7
// of k-step s of the fc2 B fragment (see the header)
__host__ __device__ __forceinline__ int hperm(int i) {
  const int q = i >> 3, b = (i >> 2) & 1, t = i & 3;
  return 16 * (q >> 1) + 8 * b + 4 * (q & 1) + t;
}

struct PsArgs {
  const float* x;
  float* xo;
  const bf16_t* ao;
  const bf16_t* pk;        // packed weights (srhip_mlp_ps_pack)
  const float *bp, *rs1, *gamma, *beta, *b1, *b2, *rs2;
  bf16_t* ln_next;
  const float *gamma_n, *beta_n;
  float eps;
  int ao_scaled, M, Hd, rows_per_sample;
};

// ---- packed image: stages in the order the ring consumes them ----------------------------------------------------------------
//   [0, 36)                  projection: stage 3 hp + p = Wp rows 32 (4 p + tt) .., columns 16 (2 hp + kk) ..; fragment f = 4 kk + tt
//   [36, 36 + 6 nch)         fc1: stage 3 T + kt = W1 rows 32 T + hperm(.), columns 16 (8 kt + f) ..; fragment f = k-step inside the third
//   [36 + 6 nch, 36 + 12 nch) fc2: stage 3 T + p = W2 rows 32 (4 p + tt) .., columns 32 T + 16 kk ..; fragment f = 4 kk + tt
// blockIdx.y = transformer block: offs (may be NULL: one block, the three pointers as given) holds per block the element offsets of
// (Wp, W1, W2) from ``Wp`` -- the bf16 copy of a backbone's flat parameter block -- and the packed images follow each other
__global__ void ps_pack_kernel(const bf16_t* __restrict__ Wp, const bf16_t* __restrict__ W1, const bf16_t* __restrict__ W2,
                               u32x4_t* __restrict__ out, const long long* __restrict__ offs, int Hd, int npieces) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npieces) return;
  if (offs) {
    const long long* o = offs + 3 * blockIdx.y;
    W1 = Wp + o[1]; W2 = Wp + o[2]; Wp = Wp + o[0];
    out += (size_t)blockIdx.y * npieces;
  }
  const int lane = i & 63, f = (i >> 6) & 7, s = i >> 9;
  const int j = lane & 31, hf = lane >> 5;
  const int nch = Hd / 64;
  const bf16_t* src;
  if (s < 3 * PROJ_HP) {
    const int hp = s / 3, p = s % 3, kk = f >> 2, tt = f & 3;
    src = Wp + (size_t)(32 * (4 * p + tt) + j) * PS_D + 16 * (2 * hp + kk) + 8 * hf;
  } else if (s < 3 * PROJ_HP + 6 * nch) {
    const int n = s - 3 * PROJ_HP, T = n / 3, kt = n % 3;
    src = W1 + (size_t)(32 * T + hperm(j)) * PS_D + 16 * (8 * kt + f) + 8 * hf;
  } else {
    const int m = s - 3 * PROJ_HP - 6 * nch, T = m / 3, p = m % 3, kk = f >> 2, tt = f & 3;
    src = W2 + (size_t)(32 * (4 * p + tt) + j) * Hd + 32 * T + 16 * kk + 8 * hf;
  }
  out[i] = *reinterpret_cast<const u32x4_t*>(src);
}

#ifdef SRHIP_TUNING
__device__ long long srhip_ps_dbg[8 * 1024];
#define PDBG_T(i) do { if (threadIdx.x == 256 && blockIdx.x < 1024) srhip_ps_dbg[8 * blockIdx.x + (i)] = wall_clock64(); } while (0)
#else
#define PDBG_T(i) do { } while (0)
#endif

// DBG (tuning builds): 1 = no GELU arithmetic, 2 = no DMA / no vmcnt waits, 4 = no fragment reads, 8 = no MFMA
// The two roles are two separate loops over the tiles (the branch is OUTSIDE the tile loop): their register sets -- 96 LayerNorm'ed fragment
// registers + 32 fc1 accumulators for the producer, 192 output accumulators for the consumer -- never coexist in a wave, and the allocator only
// sees that when no code path carries both.  Both loops execute the same sequence of barriers per tile by construction (tile start, the
// group syncs, the 12 LayerNorm hand-off barriers).
template <int DBG>
__global__ __launch_bounds__(512, 2) void mlp_ps_kernel(PsArgs a) {
  constexpr int D_ = PS_D;
  constexpr int KS = D_ / 16;             // 24 k-steps over the width
  constexpr int NT = D_ / 32;             // 12 output tiles of 32 columns
  static_assert(PROJ_G % NGR == 0 && PROJ_HP % 3 == 0, "the projection section is a whole number of ring periods");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ring = smem;
  unsigned char* exch = smem + RING_B;
  float* sb1 = reinterpret_cast<float*>(smem + RING_B + EXCH_B);
  float* sb2 = sb1 + a.Hd;
  float* sbp = sb2 + D_;
  float* sgam = sbp + D_;
  float* sbet = sgam + D_;
  float* sgn = sbet + D_;
  float* sbn = sgn + D_;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hf = lane >> 5;
  const bool isP = wave < 4;
  const int rg = wave & 3;                // row group of the wave: rows 32 rg .. 32 rg + 31 of the tile
  const int nch = a.Hd / 64;
  const int NH = 2 * nch;                 // hidden tiles of 32 units
  for (int i = tid; i < a.Hd; i += 512) sb1[i] = a.b1[i];
  for (int i = tid; i < D_; i += 512) { sb2[i] = a.b2[i]; sbp[i] = a.bp[i]; sgam[i] = a.gamma[i]; sbet[i] = a.beta[i]; }
  if (a.ln_next) for (int i = tid; i < D_; i += 512) { sgn[i] = a.gamma_n[i]; sbn[i] = a.beta_n[i]; }

  // ---- DMA: wave w moves fragment w of every stage (one instruction, 64 lanes x 16 B)
  const int pk_stages = 3 * PROJ_HP + 12 * nch;
  const __amdgpu_buffer_rsrc_t rpk = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.pk), 0, pk_stages * STB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rao = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.ao), 0, (int)min((long long)a.M * D_ * 2, 0x7fffffffLL), 0x00020000);
  // a descriptor with no records: a DMA through it moves nothing but still counts in vmcnt (groups past either end of a section)
  const __amdgpu_buffer_rsrc_t rnull = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.pk), 0, 0, 0x00020000);
  const int lfo = lane * 16;              // a lane's 16 bytes inside a fragment: the ONE long-lived address register of the loops (LDS reads take
                                          // it plus an immediate, the weight DMA takes it as its per-lane offset with wave * FRB in the scalar offset)
#define LRG (lfo + 2 * rg * FRB)          /* ... inside the wave's pair of B fragments (exchange buffers, attention-output stages): re-made per use */
  const int ntiles = (a.M + 127) / 128;
  int ao_lane = 0;                        // producer, per tile: byte offset of the lane's 16 bytes inside the attention-output rows its wave moves

  // ---- who issues the DMA.  An LDS-DMA instruction blocks its wave for ~100+ cycles, and when all eight waves issue right behind a barrier
  // both waves of every SIMD are stuck in it together: the matrix pipe idles ~250 cycles per group (first version: 80 us per launch with the
  // MFMAs, reads and DMAs knocked out).  So the stream is issued by ONE role at a time, the other one computes meanwhile:
  //   projection groups (and the initial fill): the PRODUCER waves, which have nothing else to do there -- 4 instructions per group behind the barrier;
  //   MLP groups: the CONSUMER waves, one instruction behind each of the four MFMA pairs of a group -- beside the producer's GELU + MFMAs.
  // A wave of the issuing role moves pieces 2 r and 2 r + 1 (r = its row group) of both stages of a group: part = 2 * stage + which.
  auto issue_part = [&](auto rqc, auto partc, int gi) __attribute__((always_inline)) {
    constexpr int RQ = decltype(rqc)::value, part = decltype(partc)::value, st = part >> 1, slot = 2 * RQ + st;
    if (DBG & 2) return;
    const int piece = 2 * rg + (part & 1);
    lds_void* dst = (lds_void*)(ring + slot * STB + piece * FRB);
    if (gi < PROJ_G) {
      const int hp = gi >> 1;
      // attention-output stage of half-phase hp: fragment (i, kk) = rows 32 i .. of the tile, columns 32 hp + 16 kk ..: the wave's two pieces are
      // its own row group, kk = 0 / 1 (32 bytes apart)
      if constexpr ((RQ & 1) == 0 && st == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rao, dst, 16, ao_lane, 64 * hp + 32 * (part & 1), 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rpk, dst, 16, lfo, (3 * hp + ((RQ & 1) ? 1 + st : 0)) * STB + piece * FRB, 0, 0);   // [ao | Wp0] [Wp1 | Wp2]
    } else {
      const int gm = gi - PROJ_G, h = gm / 3, kt = gm - 3 * h;
      const bool valid = st == 0 ? h < NH : (h >= 2 && h < NH + 2);       // positions of the schedule that carry nothing: no-op DMAs (vmcnt stays uniform)
      const int stage = st == 0 ? 3 * PROJ_HP + 3 * h + kt : 3 * PROJ_HP + 6 * nch + 3 * (h - 2) + kt;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(valid ? rpk : rnull, dst, 16, lfo, valid ? stage * STB + piece * FRB : 0, 0, 0);
    }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  using P3 = std::integral_constant<int, 3>;
  auto issue_group = [&](auto rqc, int gi) __attribute__((always_inline)) {
    issue_part(rqc, P0{}, gi); issue_part(rqc, P1{}, gi); issue_part(rqc, P2{}, gi); issue_part(rqc, P3{}, gi);
  };
  // Group protocol (G = group about to be multiplied, ring group RQ = G % NGR): behind the barrier group G + 1 has landed for everybody -- the
  // issuing role waited for its own pieces (counted vmcnt: the AHEAD - 2 younger groups stay in flight), the barrier covers the others' -- and the
  // slots of group G - 1, consumed by every wave before it arrived, take group G + AHEAD.
  constexpr int VM_STEADY = 4 * (AHEAD - 2);
#define RQ_OF(RQ, D) std::integral_constant<int, ((RQ) + (D)) % NGR>{}
#define FRAG(SLOT, F) (((DBG & 4) != 0) ? u32x4_t{(unsigned)(SLOT), 1u, 2u, (unsigned)(F)} \
                                        : *reinterpret_cast<const u32x4_t*>(ring + (SLOT) * STB + (F) * FRB + lfo))

  if (isP) {
    // ======================================================================= PRODUCER ====================================================
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const int rowc = min(tile * 128 + 32 * rg + j, a.M - 1);
      float rs2v = 1.0f;
      if (a.rs2) rs2v = a.rs2[rowc / a.rows_per_sample];
      {
        const int ao_row = min(tile * 128 + 32 * rg + j, a.M - 1);
        ao_lane = (ao_row * D_ + 8 * hf) * 2;
      }
      __syncthreads();                     // the previous tile is done with the ring, the exchange buffers (and the bias tables are written)
      sfor<AHEAD>([&](auto gc) __attribute__((always_inline)) { issue_group(gc, decltype(gc)::value); });
      if constexpr ((DBG & 2) == 0) wait_vm<4 * (AHEAD - 1)>();       // own pieces of group 0
      __builtin_amdgcn_s_barrier();
      // projection groups: wait for the own pieces of group G + 1, barrier, issue group G + AHEAD while it is a projection group (the consumer
      // takes over with the first MLP group: the last AHEAD iterations only drain, with the exact count of what is still in flight)
      for (int u3 = 0; u3 < PROJ_G / NGR - 1; ++u3)
        sfor<NGR>([&](auto rqc) __attribute__((always_inline)) {
          if constexpr ((DBG & 2) == 0) wait_vm<VM_STEADY>();
          __builtin_amdgcn_s_barrier();
          issue_group(RQ_OF(decltype(rqc)::value, AHEAD), NGR * u3 + decltype(rqc)::value + AHEAD);
        });
      sfor<NGR>([&](auto rqc) __attribute__((always_inline)) {
        constexpr int RQ = decltype(rqc)::value, G = PROJ_G - NGR + RQ;                 // group G + 1 must have landed; issued: up to PROJ_G - 1
        constexpr int infl = (PROJ_G - 1) - (G + 1);                                     // own groups younger than G + 1
        if constexpr ((DBG & 2) == 0) wait_vm<4 * (infl < 0 ? 0 : infl > AHEAD - 2 ? AHEAD - 2 : infl)>();
        __builtin_amdgcn_s_barrier();
        if constexpr (G + AHEAD < PROJ_G) issue_group(RQ_OF(RQ, AHEAD), G + AHEAD);
      });
      u32x4_t fa0[4], fa1[4];
      // ---- LayerNorm'ed rows from the consumer: 6 rounds of 4 k-steps through the exchange buffers
      u32x4_t xn[KS];
#pragma unroll
      for (int r = 0; r < KS / 4; ++r) {
        __builtin_amdgcn_s_barrier();      // round r is written
#pragma unroll
        for (int e = 0; e < 4; ++e) xn[4 * r + e] = *reinterpret_cast<const u32x4_t*>(exch + (4 * rg + e) * FRB + lfo);
        wait_lgkm0();
        __builtin_amdgcn_s_barrier();      // ... and read
      }
      // first quad of the first fc1 stage: group PROJ_G (ring group 0), issued by the consumer waves behind the projection, landed for everybody with
      // the last hand-off round (the consumer waited for its pieces in front of that barrier)
#pragma unroll
      for (int f = 0; f < 4; ++f) fa0[f] = FRAG(0, f);
      f32x16_t acc1[2];
      u32x4_t hq;
      auto acc_start = [&](auto pc, int T) __attribute__((always_inline)) {       // fc1 bias of tile T in the result layout
        constexpr int p = decltype(pc)::value;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(sb1 + 32 * T + 16 * s + 8 * hf);
          const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(sb1 + 32 * T + 16 * s + 8 * hf + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { acc1[p][8 * s + e] = lo[e]; acc1[p][8 * s + 4 + e] = hi[e]; }
        }
      };
      // GELU (gelu_poly2 of common.h, the same operations in the same order) + DropPath factor + bf16 pack of the 8 result registers 8 s .. 8 s + 7
      // of the tile in acc1[p] = the fc2 B fragment of k-step s.  One k-step = FOUR independent chains of packed fp32 operations, cut into four
      // slices that ride between the MFMA pairs of a group: a single chain (12 dependent packed operations) runs at its latency, not at
      // the vector ALU's rate -- measured: the loop skeleton with one pair in flight took 1.1 us per half-phase for 8 pairs.
      f32x2_t gx[4], gt[4], gp[4];
      auto sp = [](float c) { return f32x2_t{c, c}; };
      auto gelu_slice = [&](auto pc, auto sc, auto stc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value, s = decltype(sc)::value, st = decltype(stc)::value;
        constexpr float XS = 4.252893f;
        if constexpr ((DBG & 1) != 0) {
          if constexpr (st == 3) {
#pragma unroll
            for (int e = 0; e < 4; ++e) hq[e] = pack_bf2(acc1[p][8 * s + 2 * e] * rs2v, acc1[p][8 * s + 2 * e + 1] * rs2v);
          }
        } else if constexpr (st == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            gx[e] = f32x2_t{__builtin_amdgcn_fmed3f(acc1[p][8 * s + 2 * e], -XS, XS), __builtin_amdgcn_fmed3f(acc1[p][8 * s + 2 * e + 1], -XS, XS)};
            gt[e] = gx[e] * gx[e];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) gp[e] = __builtin_elementwise_fma(sp(5.564872638e-11f), gt[e], sp(-5.327768675e-09f));
#pragma unroll
          for (int e = 0; e < 4; ++e) gp[e] = __builtin_elementwise_fma(gp[e], gt[e], sp(2.255431416e-07f));
        } else if constexpr (st == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) gp[e] = __builtin_elementwise_fma(gp[e], gt[e], sp(-5.626433893e-06f));
#pragma unroll
          for (int e = 0; e < 4; ++e) gp[e] = __builtin_elementwise_fma(gp[e], gt[e], sp(9.341875929e-05f));
#pragma unroll
          for (int e = 0; e < 4; ++e) gp[e] = __builtin_elementwise_fma(gp[e], gt[e], sp(-1.108561217e-03f));
        } else if constexpr (st == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) gp[e] = __builtin_elementwise_fma(gp[e], gt[e], sp(9.815971766e-03f));
#pragma unroll
          for (int e = 0; e < 4; ++e) gp[e] = __builtin_elementwise_fma(gp[e], gt[e], sp(-6.634449185e-02f));
#pragma unroll
          for (int e = 0; e < 4; ++e) gp[e] = __builtin_elementwise_fma(gp[e], gt[e], sp(3.989023390e-01f));
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x2_t phi = __builtin_elementwise_fma(gp[e], gx[e], sp(0.5f));
            const f32x2_t g2 = (f32x2_t{acc1[p][8 * s + 2 * e], acc1[p][8 * s + 2 * e + 1]} * phi) * f32x2_t{rs2v, rs2v};
            hq[e] = pack_bf2(g2[0], g2[1]);
          }
        }
      };
      acc_start(std::integral_constant<int, 0>{}, 0);
      // half-phase h, parity p = h & 1.  ACT: the MFMAs of tile h into acc1[p] (h < NH); GEL: GELU of tile h - 1 from acc1[1 - p] in the first
      // two groups (1 <= h <= NH); NEXT: tile h + 1 exists (its bias start, its first weight quad).  All three are compile-time: the steady loop
      // has no branch.
      auto half_phase = [&](auto pc, auto actc, auto gelc, auto nextc, int h) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        constexpr bool ACT = decltype(actc)::value, GEL = decltype(gelc)::value, NEXT = decltype(nextc)::value;
        using Q = std::integral_constant<int, 1 - p>;
        sfor<3>([&](auto ktc) __attribute__((always_inline)) {
          constexpr int kt = decltype(ktc)::value;
          constexpr int RQ = 3 * p + kt, slot = 2 * RQ, nslot = 2 * ((RQ + 1) % NGR);
          constexpr bool G = GEL && kt < 2;
          __builtin_amdgcn_s_barrier();    // (the consumer waves issue and wait for the MLP groups)
          if constexpr (ACT) {
#pragma unroll
            for (int f = 0; f < 4; ++f) fa1[f] = FRAG(slot, 4 + f);
          }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (ACT && (DBG & 8) == 0) {
            acc1[p] = mfma32(fa0[0], xn[8 * kt + 0], acc1[p]);
            acc1[p] = mfma32(fa0[1], xn[8 * kt + 1], acc1[p]);
          }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (G) gelu_slice(Q{}, ktc, std::integral_constant<int, 0>{});
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (ACT && (DBG & 8) == 0) {
            acc1[p] = mfma32(fa0[2], xn[8 * kt + 2], acc1[p]);
            acc1[p] = mfma32(fa0[3], xn[8 * kt + 3], acc1[p]);
          }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (G) gelu_slice(Q{}, ktc, std::integral_constant<int, 1>{});
          if constexpr ((kt < 2) ? ACT : NEXT) {
#pragma unroll
            for (int f = 0; f < 4; ++f) fa0[f] = FRAG(nslot, f);
          }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (ACT && (DBG & 8) == 0) {
            acc1[p] = mfma32(fa1[0], xn[8 * kt + 4], acc1[p]);
            acc1[p] = mfma32(fa1[1], xn[8 * kt + 5], acc1[p]);
          }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (G) gelu_slice(Q{}, ktc, std::integral_constant<int, 2>{});
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (ACT && (DBG & 8) == 0) {
            acc1[p] = mfma32(fa1[2], xn[8 * kt + 6], acc1[p]);
            acc1[p] = mfma32(fa1[3], xn[8 * kt + 7], acc1[p]);
          }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (G) {
            gelu_slice(Q{}, ktc, std::integral_constant<int, 3>{});
            // hidden tile h - 1 (buffer (h - 1) & 1 = 1 - p), k-step kt, rows of this wave: the consumer's B fragment, in lane order
            *reinterpret_cast<u32x4_t*>(exch + (1 - p) * STB + kt * FRB + LRG) = hq;
            if constexpr (kt == 1) wait_lgkm0();   // both writes of the tile are complete before the consumer reads it (behind the next barrier)
          }
          if constexpr (kt == 2 && NEXT) acc_start(Q{}, h + 1);    // (the GELU of the tile that lived there is over: groups 0 and 1)
        });
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using Y = std::true_type;
      using N = std::false_type;
      // tiles 0 .. NH - 1 (NH = 2 nch >= 4): first pair (no GELU behind tile 0), steady pairs, last pair (GELU of the last tile, then idle)
      half_phase(I0{}, Y{}, N{}, Y{}, 0);
      half_phase(I1{}, Y{}, Y{}, Y{}, 1);
      for (int h2 = 1; h2 < nch - 1; ++h2) {
        half_phase(I0{}, Y{}, Y{}, Y{}, 2 * h2);
        half_phase(I1{}, Y{}, Y{}, Y{}, 2 * h2 + 1);
      }
      half_phase(I0{}, Y{}, Y{}, Y{}, NH - 2);
      half_phase(I1{}, Y{}, Y{}, N{}, NH - 1);
      half_phase(I0{}, N{}, Y{}, N{}, NH);
      half_phase(I1{}, N{}, N{}, N{}, NH + 1);
#ifdef SRHIP_TUNING
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
#endif
    }
  } else {
    // ======================================================================= CONSUMER ====================================================
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const int rowc = min(tile * 128 + 32 * rg + j, a.M - 1);     // rows past M: the clamped row M - 1 is recomputed and rewritten with identical values
      float rs1v = 1.0f, rs2v = 1.0f;
      if (a.rs1) rs1v = a.rs1[rowc / a.rows_per_sample];
      if (a.rs2) rs2v = a.rs2[rowc / a.rows_per_sample];
      // the rows' x is the START VALUE of the output accumulators (register 4 q + t of tile To = column 32 To + 8 q + 4 hf + t of row j);
      // requested before the weight stream so that one wait covers both
      f32x16_t acc2[NT];
      {
        const float* xr = a.x + (size_t)rowc * D_ + 4 * hf;
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4_t v = *reinterpret_cast<const f32x4_t*>(xr + 32 * T + 8 * q);
            acc2[T][4 * q] = v[0]; acc2[T][4 * q + 1] = v[1]; acc2[T][4 * q + 2] = v[2]; acc2[T][4 * q + 3] = v[3];
          }
      }
      PDBG_T(0);
      __syncthreads();                     // the previous tile is done with the ring, the exchange buffers (and the bias tables are written)
      __builtin_amdgcn_s_barrier();        // group 0 has landed (the producer waves issued and waited)
#pragma unroll
      for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(sbp + 32 * T + 8 * q + 4 * hf);
#pragma unroll
          for (int t = 0; t < 4; ++t) acc2[T][4 * q + t] += rs1v * bb[t];
        }
      // (row_scale1 is already ON the attention output -- srhip_attn_block_fused out_scale -- and only scales the projection bias here)
      // Register budget of the consumer: 192 accumulators + THREE buffers of two weight fragments (24) + one pair of B fragments (8).  A product
      // stage of 8 fragments is 4 "pairs" (pair q: k-step kk = q / 2, output tiles 4 p3 + 2 (q % 2), + 1); the pipeline requests pair i + 2 into
      // buffer (i + 2) % 3 in front of the two MFMAs of pair i: every fragment is asked for 4 MFMAs (>= 128 cycles) ahead of its use, and a
      // half-phase of 12 pairs brings the rotation back to buffer 0.  (Two buffers of four -- 32 registers -- left the allocator a dozen short:
      // it spilled addresses into the loops, and every reload drains the DMA ring.)  The next B fragments go into hb[kk] right behind the last
      // MFMA that reads the old ones (pairs 9 and 11), 4+ MFMAs ahead of their first use.
      u32x4_t fb[3][2], hb[2];
      // requests of pair q of the stage in ring slot SLOT into buffer B; the two MFMAs of pair q (stage index p3 = which third of the outputs)
#define CREAD(B, SLOT, Q) { fb[B][0] = FRAG(SLOT, 4 * ((Q) >> 1) + 2 * ((Q) & 1)); fb[B][1] = FRAG(SLOT, 4 * ((Q) >> 1) + 2 * ((Q) & 1) + 1); }
      auto cmma = [&](auto bc, auto qc, auto p3c) __attribute__((always_inline)) {
        constexpr int B = decltype(bc)::value, q = decltype(qc)::value, p3 = decltype(p3c)::value, kk = q >> 1, t0 = 4 * p3 + 2 * (q & 1);
        __builtin_amdgcn_sched_barrier(0);      // (the requests above stay in front of these MFMAs: left alone, the scheduler sinks every read
        if constexpr ((DBG & 8) == 0) {         //  to one MFMA before its use, and a 32-cycle MFMA does not cover a ~100-cycle LDS read)
          acc2[t0] = mfma32(fb[B][0], hb[kk], acc2[t0]);
          acc2[t0 + 1] = mfma32(fb[B][1], hb[kk], acc2[t0 + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      // one half-phase = 12 pairs over three stages in slots S0, S1, S2; SYNC(i) runs in front of pair i (i = 0, 4, 8); the look-ahead of the last
      // two pairs goes to the NEXT half-phase's first stage (slot SN) and its B fragments (NEXTB(kk))
#define CHALF(S0, S1, S2, SN, SYNC, NEXTB, ISSUE)                                                                                   \
      sfor<12>([&](auto ic) __attribute__((always_inline)) {                                                                 \
        constexpr int i = decltype(ic)::value, st = i / 4, q = i % 4, n = i + 2, nst = (n % 12) / 4, nq = n % 4;             \
        constexpr int nslot = n >= 12 ? (SN) : nst == 0 ? (S0) : nst == 1 ? (S1) : (S2);                                      \
        if constexpr (q == 0) { SYNC(st); }                                                                                  \
        CREAD(n % 3, nslot, nq)                                                                                              \
        cmma(std::integral_constant<int, i % 3>{}, std::integral_constant<int, q>{}, std::integral_constant<int, st>{});      \
        ISSUE(st, q);                                                                                                        \
        if constexpr (i == 9) { NEXTB(0); }                                                                                  \
        if constexpr (i == 11) { NEXTB(1); }                                                                                 \
      });
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) hb[kk] = *reinterpret_cast<const u32x4_t*>(ring + kk * FRB + LRG);     // ao stage of half-phase 0: slot 0
      CREAD(0, 1, 0)
      CREAD(1, 1, 1)
      PDBG_T(1);
      // ---- projection: half-phase hp = groups 2 hp ([ao | Wp0]) and 2 hp + 1 ([Wp1 | Wp2]); 3 half-phases = one ring period.  (Behind the last
      // half-phase the look-ahead reads MLP stages that nothing multiplies.)
      for (int u3 = 0; u3 < PROJ_HP / 3; ++u3)
        sfor<3>([&](auto uc) __attribute__((always_inline)) {
          constexpr int u = decltype(uc)::value, sa = 4 * u, sn = (4 * u + 4) % NSL;      // slots sa (ao), sa + 1 (Wp0), sa + 2 (Wp1), sa + 3 (Wp2)
          const int hp = 3 * u3 + u;
#define PSYNC(st) if constexpr ((st) < 2) __builtin_amdgcn_s_barrier()                  /* (the producer waves issue and wait for the projection groups) */
#define PNEXTB(kk) hb[kk] = *reinterpret_cast<const u32x4_t*>(ring + sn * STB + (kk) * FRB + LRG)
#define PISSUE(st, q)
          CHALF(sa + 1, sa + 2, sa + 3, sn + 1, PSYNC, PNEXTB, PISSUE)
#undef PSYNC
#undef PNEXTB
#undef PISSUE
        });
      PDBG_T(4);
      // the consumer takes the DMA stream over: the first AHEAD MLP groups go into the ring groups the projection has left (all but the last one
      // were free a barrier ago; the last projection group -- ring group NGR - 1 -- is not among them), and land under the LayerNorm below
      sfor<AHEAD>([&](auto gc) __attribute__((always_inline)) { issue_group(gc, PROJ_G + decltype(gc)::value); });
      // ---- x1 = the accumulators; LayerNorm (vit.py:165 norm2) from the accumulator layout; bf16 rows to the producer in ITS B-fragment order:
      // columns 32 To + 8 q + 4 hf .. + 3 of row j -> fragment k-step 2 To + q / 2, lane (j, q % 2), bytes 8 hf .. + 7
      {
        float sum = 0.f;
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
          for (int e = 0; e < 16; ++e) sum += acc2[T][e];
        const float mu = halves_sum(sum) * (1.0f / D_);
        float qv = 0.f;
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
          for (int e = 0; e < 16; ++e) { const float d = acc2[T][e] - mu; qv += d * d; }
        const float rstd = rsqrtf(halves_sum(qv) * (1.0f / D_) + a.eps);
        int ko = 4 * hf;                   // (re-made opaque every round: the 96 affine reads / 96 packed pairs of all rounds must not be computed up
                                           // front -- that is ~190 live registers next to the 192 accumulators = spills)
#pragma unroll
        for (int r = 0; r < KS / 4; ++r) {
          float mur = mu, rsr = rstd;
          asm volatile("" : "+v"(ko), "+v"(mur), "+v"(rsr));
#pragma unroll
          for (int t2 = 0; t2 < 2; ++t2) {
            const int T = 2 * r + t2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4_t g4 = *reinterpret_cast<const f32x4_t*>(sgam + 32 * T + 8 * q + ko);
              const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(sbet + 32 * T + 8 * q + ko);
              u32x2_t pk2;
              pk2[0] = pack_bf2((acc2[T][4 * q] - mur) * rsr * g4[0] + b4[0], (acc2[T][4 * q + 1] - mur) * rsr * g4[1] + b4[1]);
              pk2[1] = pack_bf2((acc2[T][4 * q + 2] - mur) * rsr * g4[2] + b4[2], (acc2[T][4 * q + 3] - mur) * rsr * g4[3] + b4[3]);
              *reinterpret_cast<u32x2_t*>(exch + (4 * rg + 2 * t2 + (q >> 1)) * FRB + 16 * (32 * (q & 1) + j) + 8 * hf) = pk2;
            }
          }
          wait_lgkm0();
          if constexpr ((DBG & 2) == 0) { if (r == KS / 4 - 1) wait_vm<4 * (AHEAD - 1)>(); }     // own pieces of the first MLP group: the producer reads it behind this barrier
          __builtin_amdgcn_s_barrier();    // round r is written
          __builtin_amdgcn_s_barrier();    // ... and read
        }
      }
      PDBG_T(5);
      // fc2 accumulates on x1 + rs2 * b2
#pragma unroll
      for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(sb2 + 32 * T + 8 * q + 4 * hf);
#pragma unroll
          for (int t = 0; t < 4; ++t) acc2[T][4 * q + t] += rs2v * bb[t];
        }
      // ---- MLP: half-phase h multiplies hidden tile h - 2 (the k-slice 32 (h - 2) .. + 31 of fc2) into all 12 output tiles.  Its two B
      // fragments were written by the producer in half-phase h - 1, groups 0 and 1, and are requested behind pairs 9 and 11 of that half-phase
      // (group 2).  Half-phases 0 and 1 are syncs only (the producer is two tiles ahead); the steady loop has no branch (the look-ahead behind
      // the last half-phase reads stale LDS that nothing multiplies).
      sfor<NGR>([&](auto rqc) __attribute__((always_inline)) {
        if constexpr ((DBG & 2) == 0) wait_vm<VM_STEADY>();
        __builtin_amdgcn_s_barrier();
        issue_group(RQ_OF(decltype(rqc)::value, AHEAD), PROJ_G + decltype(rqc)::value + AHEAD);
      });
      hb[0] = *reinterpret_cast<const u32x4_t*>(exch + LRG);              // hidden tile 0 (buffer 0), complete since group 1 of half-phase 1
      hb[1] = *reinterpret_cast<const u32x4_t*>(exch + FRB + LRG);
      CREAD(0, 1, 0)                                                       // first stage of half-phase 2: ring group 0, slot 1
      CREAD(1, 1, 1)
      for (int h2 = 1; h2 < nch + 1; ++h2)
        sfor<2>([&](auto pc) __attribute__((always_inline)) {
          constexpr int p = decltype(pc)::value;
          const int h = 2 * h2 + p;
#define MSYNC(st) { if constexpr ((DBG & 2) == 0) wait_vm<VM_STEADY>(); __builtin_amdgcn_s_barrier(); }
#define MNEXTB(kk) hb[kk] = *reinterpret_cast<const u32x4_t*>(exch + (1 - p) * STB + (kk) * FRB + LRG)     /* tile h - 1: buffer (h - 1) & 1 */
#define MISSUE(st, q) issue_part(RQ_OF(3 * p + (st), AHEAD), std::integral_constant<int, (q)>{}, PROJ_G + 3 * h + (st) + AHEAD)
          CHALF(6 * p + 1, 6 * p + 3, 6 * p + 5, (6 * p + 7) % NSL, MSYNC, MNEXTB, MISSUE)
#undef MSYNC
#undef MNEXTB
#undef MISSUE
        });
#undef CHALF
#undef CREAD
      PDBG_T(2);
      // ---- epilogue: y (fp32) and, optionally, LayerNorm(y) with the next block's norm1 (bf16)
      float* xw = a.xo + (size_t)rowc * D_ + 4 * hf;
#pragma unroll
      for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4_t*>(xw + 32 * T + 8 * q) = f32x4_t{acc2[T][4 * q], acc2[T][4 * q + 1], acc2[T][4 * q + 2], acc2[T][4 * q + 3]};
      if (a.ln_next) {                      // wave-uniform
        float sum = 0.f;
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
          for (int e = 0; e < 16; ++e) sum += acc2[T][e];
        const float mu = halves_sum(sum) * (1.0f / D_);
        float qv = 0.f;
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
          for (int e = 0; e < 16; ++e) { const float d = acc2[T][e] - mu; qv += d * d; }
        const float rstd = rsqrtf(halves_sum(qv) * (1.0f / D_) + a.eps);
        bf16_t* lw = a.ln_next + (size_t)rowc * D_ + 8 * hf;
        int ko = 4 * hf;
#pragma unroll
        for (int T = 0; T < NT; ++T) {
          float mur = mu, rsr = rstd;
          asm volatile("" : "+v"(ko), "+v"(mur), "+v"(rsr));
#pragma unroll
          for (int qp = 0; qp < 2; ++qp) {          // q = 2 qp (-> lane half 0 keeps it) and 2 qp + 1 (-> lane half 1)
            unsigned pk[2][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int q = 2 * qp + e;
              const f32x4_t g4 = *reinterpret_cast<const f32x4_t*>(sgn + 32 * T + 8 * q + ko);
              const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(sbn + 32 * T + 8 * q + ko);
              pk[e][0] = pack_bf2((acc2[T][4 * q] - mur) * rsr * g4[0] + b4[0], (acc2[T][4 * q + 1] - mur) * rsr * g4[1] + b4[1]);
              pk[e][1] = pack_bf2((acc2[T][4 * q + 2] - mur) * rsr * g4[2] + b4[2], (acc2[T][4 * q + 3] - mur) * rsr * g4[3] + b4[3]);
            }
            // permlane32_swap(a, b) = {a.lo b.lo, a.hi b.hi}: lane half 0 ends up with columns 8 q .. + 7 of q = 2 qp, half 1 with those of q = 2 qp + 1
            const auto w0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
            const auto w1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
            *reinterpret_cast<u32x4_t*>(lw + 32 * T + 16 * qp) = u32x4_t{w0[0], w1[0], w0[1], w1[1]};
          }
        }
      }
#ifdef SRHIP_TUNING
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      PDBG_T(3);
#endif
    }
  }
#undef FRAG
#undef LRG
#undef RQ_OF
}


}  // namespace

#ifdef SRHIP_TUNING
extern "C" int srhip_mlp_ps_debug(long long* out_host, int n) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(srhip_ps_dbg), (size_t)n * sizeof(long long)) == hipSuccess ? SR_OK : SR_EINVAL;
}
#endif

extern "C" long long srhip_mlp_ps_pack_bytes(int D, int Hd) {
  if (D != PS_D || Hd < 128 || (Hd % 64) || Hd > 4096) return SR_EINVAL;
  return (long long)(3 * PROJ_HP + 12 * (Hd / 64)) * STB;
}

extern "C" int srhip_mlp_ps_pack(const void* Wp, const void* W1, const void* W2, void* packed, int D, int Hd, void* stream) {
  if (!Wp || !W1 || !W2 || !packed) return SR_EINVAL;
  if (D != PS_D || Hd < 128 || (Hd % 64) || Hd > 4096) return SR_EINVAL;
  if (((uintptr_t)Wp | (uintptr_t)W1 | (uintptr_t)W2 | (uintptr_t)packed) & 15) return SR_EINVAL;
  const int npieces = (3 * PROJ_HP + 12 * (Hd / 64)) * 8 * 64;
  hipLaunchKernelGGL(ps_pack_kernel, dim3(cdiv(npieces, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)Wp, (const bf16_t*)W1,
                     (const bf16_t*)W2, (u32x4_t*)packed, (const long long*)nullptr, Hd, npieces);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_mlp_ps_pack_blocks(const void* flat_bf16, const long long* offsets, int n_blocks, void* packed, int D, int Hd, void* stream) {
  if (!flat_bf16 || !offsets || !packed || n_blocks <= 0 || n_blocks > 65535) return SR_EINVAL;
  if (D != PS_D || Hd < 128 || (Hd % 64) || Hd > 4096) return SR_EINVAL;
  if (((uintptr_t)flat_bf16 | (uintptr_t)packed) & 15) return SR_EINVAL;
  const int npieces = (3 * PROJ_HP + 12 * (Hd / 64)) * 8 * 64;
  hipLaunchKernelGGL(ps_pack_kernel, dim3(cdiv(npieces, 256), n_blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)flat_bf16,
                     (const bf16_t*)nullptr, (const bf16_t*)nullptr, (u32x4_t*)packed, offsets, Hd, npieces);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_mlp_ps_proj(const float* x, float* x_out, const void* ao, const void* packed, const float* bp, const float* row_scale1,
                                 int ao_scaled, const float* ln_gamma, const float* ln_beta, float eps, const float* b1, const float* b2,
                                 const float* row_scale2, int rows_per_sample, void* ln_next, const float* next_gamma, const float* next_beta,
                                 int M, int D, int Hd, void* stream) {
  if (!x || !x_out || !ao || !packed || !bp || !ln_gamma || !ln_beta || !b1 || !b2 || M <= 0) return SR_EINVAL;
  if (ln_next && (!next_gamma || !next_beta || ((uintptr_t)ln_next & 15))) return SR_EINVAL;
  if (D != PS_D || Hd < 128 || (Hd % 64) || Hd > 1536 || (long long)M * D * 2 > 0x7fffffffLL) return SR_EINVAL;
  if ((row_scale1 || row_scale2) && rows_per_sample <= 0) return SR_EINVAL;
  if (row_scale1 && !ao_scaled) return SR_EINVAL;       // the factor must ride on ao (one rounding, as in the unfused path): srhip_attn_block_fused out_scale
  if (((uintptr_t)x | (uintptr_t)x_out | (uintptr_t)ao | (uintptr_t)packed) & 15) return SR_EINVAL;
  PsArgs a;
  a.x = x; a.xo = x_out; a.ao = (const bf16_t*)ao; a.pk = (const bf16_t*)packed; a.bp = bp; a.rs1 = row_scale1; a.gamma = ln_gamma;
  a.beta = ln_beta; a.b1 = b1; a.b2 = b2; a.rs2 = row_scale2; a.ln_next = (bf16_t*)ln_next; a.gamma_n = next_gamma; a.beta_n = next_beta;
  a.eps = eps; a.ao_scaled = ao_scaled; a.M = M; a.Hd = Hd; a.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1;
  const size_t smem = (size_t)RING_B + EXCH_B + (size_t)(Hd + 6 * D) * sizeof(float);
  void (*kern)(PsArgs) = mlp_ps_kernel<0>;
#ifdef SRHIP_TUNING
  switch (getenv("SRHIP_PS_DEBUG") ? atoi(getenv("SRHIP_PS_DEBUG")) : 0) {
    case 1: kern = mlp_ps_kernel<1>; break;
    case 2: kern = mlp_ps_kernel<2>; break;
    case 8: kern = mlp_ps_kernel<8>; break;
    case 9: kern = mlp_ps_kernel<9>; break;
    case 10: kern = mlp_ps_kernel<10>; break;
    case 14: kern = mlp_ps_kernel<14>; break;
    case 15: kern = mlp_ps_kernel<15>; break;
    default: break;
  }
#endif
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return SR_ELAUNCH;
  const int ntiles = cdiv(M, 128);
  hipLaunchKernelGGL(kern, dim3(min(ntiles, 256)), dim3(512), smem, (hipStream_t)stream, a);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
