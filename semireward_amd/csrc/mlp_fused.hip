// Fused inference MLP half of a transformer block (gfx950):
//     x <- x + row_scale * ( fc2( GELU( fc1( LayerNorm(x) ) ) ) + b2 )            reference: semilearn/nets/vit/vit.py:165
// (Block.forward, second residual: x + drop_path2(ls2(mlp(norm2(x))))), Mlp.forward vit.py:69-75.
//
// Why: at the reference batch 200 of the 216 images of a step are forwarded WITHOUT a backward (the K weak/strong passes that
// only feed the rewarder / the masks).  As three launches (LN, fc1+GELU, fc2+residual) the [M, 4D] hidden activation makes a
// round trip through HBM (2 x 158 MB per layer at M = 51400) and LN re-reads x; fused, HBM sees x in, x out.
//
// Structure (one workgroup = 128 rows, 8 waves x 16 rows, ONE workgroup per CU, 254 VGPRs):
//   * every wave owns 16 rows end to end: it normalises them in registers (bf16 MFMA B-fragments, 48 VGPRs), so neither x
//     nor the hidden activation ever sits in LDS -- all 128 KiB of the ring hold WEIGHT tiles.
//   * weights stream through a 16-slot LDS-DMA ring of 8 KiB tiles ([128 rows x 32 k] bf16, buffer_load ... lds); per
//     64-wide hidden chunk c, 12 stages of 8 MFMAs per wave:
//       stages 0-5  (GEMM1): W1 rows of hidden pair-group u = j/3 (32 hidden) x k-steps 4(j%3) .. +3
//       stages 6-11 (GEMM2): W2 rows 128 th .. (th = j%3) x hidden 64 c + 32 u .., u = (j-6)/3
//     so the accumulators of group u = 0 are complete after stage 2 and its GELU (VALU) has stages 3-5 to hide under; u = 1
//     is finished by stage 5 and needed by stage 9.
//   * GEMM1: D1 = mfma(W1 frag, xn frag): lane (m = lane&15, g = lane>>4) receives rows 4g..4g+3 of the A tile.  A-tile row i
//     of tile P is fed with hidden unit 8 (i>>2) + (i&3), tile Q with 8 (i>>2) + 4 + (i&3) (just the LDS row each lane
//     reads), so after GELU + bias + bf16 pack the lane holds hidden 8g .. 8g+7 of the group: exactly the B fragment of
//     k-step u of GEMM2 -- the chained-MFMA trick, the hidden activation never leaves registers.
//   * software pipeline inside every wave (fragments of stage s+1 requested before the MFMAs of stage s, two fragment sets),
//     one barrier per GS stages, no branches in the loop body (out-of-range ring refills are buffer-OOB no-ops that keep the
//     vmcnt arithmetic uniform).
//   * epilogue: re-read x (fp32), add, store.
// Measured alternatives (tools/microbench.py mlp, M = 51400): 4 waves x 32 rows (one wave per SIMD, every fragment feeding
// two MFMAs, 512 registers) 357 us vs 248 us for this layout -- a lone wave per SIMD stalls at every s_waitcnt; lockstep
// stages without the in-wave prefetch 330 us; GELU in one lump per chunk +10 us.  PMC of this kernel: MFMA pipe 19 % busy, LDS
// array 21 %, VALU 18 %, waves 30 % of their time in s_waitcnt: latency-, not throughput-bound at two waves per SIMD.
// Rounding points are the ones of the unfused path (xn bf16, GELU output bf16, fp32 accumulation), so the two paths agree to
// fp32 summation order.
#include <stdlib.h>

#include <utility>

#include "../../include/srhip.h"
#include "common.h"

namespace {

constexpr int BK = 32;
constexpr int RT = 128;                   // rows (weights) per ring tile
constexpr int TILE_EL = RT * BK;          // 4096 bf16 = 8 KiB
constexpr int NS = 16;                    // ring slots
constexpr int FBM = 128;                  // x rows per workgroup
constexpr int CH = 64;                    // hidden units per chunk (acc1 = 4 tiles: 16 VGPRs; 128 would not fit 256 VGPRs)

__device__ __forceinline__ int swz(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }   // as in gemm.hip

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
template <int N_>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(<N-1>) -- the stage index must be a constant expression
// (s_waitcnt immediates, accumulator indices)
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

struct MlpArgs {
  const float* x;          // input residual stream (LayerNorm source and residual term)
  float* xo;               // output residual stream (== x for the in-place call)
  // activations kept for the backward of the first ``save_rows`` rows (the gradient-carrying images of a mixed batch), or NULL
  bf16_t *s_ln2, *s_pre, *s_h;     // [save_rows, D], [save_rows, Hd], [save_rows, Hd]
  float *s_mean, *s_rstd;          // [save_rows]
  int save_rows;
  const float *gamma, *beta, *b1, *b2, *row_scale;
  const bf16_t *W1, *W2;
  // PROJ variant (rows without a backward): the attention output projection + first residual of the block run in this launch too,
  //   x <- x + row_scale1 * (ao Wp^T + bp)      (vit.py:163 after Attention.forward :105-106), then the MLP half on the result
  const bf16_t *ao, *Wp;   // [M, D] attention output (heads concatenated), [D, D]
  const float *bp, *row_scale1;
  int ao_scaled;           // the producer of ao (srhip_attn_block_fused out_scale) already applied row_scale1 to it
  // PROJ variant, optional: LayerNorm of the OUTPUT rows with the next block's norm1 affine, written as bf16 [M, D] -- the operand of the next
  // block's qkv projection (srhip_attn_block_fused), which saves that block's srhip_layernorm_fwd launch and its read of the residual stream
  bf16_t* ln_next;
  const float *gamma_n, *beta_n;
  float eps;
  int M, Hd, rows_per_sample;
};

// DBG (tuning builds only, SRHIP_MLP_DEBUG): 1 = no GELU, 2 = no DMA / no vmcnt waits, 4 = no ds_reads, 8 = no MFMA
#ifdef SRHIP_TUNING
__device__ long long srhip_mlp_dbg[8 * 1024];
#define MDBG_T(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) srhip_mlp_dbg[8 * blockIdx.x + (i)] = wall_clock64(); } while (0)
#else
#define MDBG_T(i) do { } while (0)
#endif
// SPREAD: the ring refills are issued one per stage behind the stage's first MFMA half ("stage j asks for stage j + PD") instead of GS in a row
// right behind the group barrier.  An LDS-DMA instruction blocks its wave for ~100 cycles, and behind a barrier all eight waves -- both waves of
// every SIMD -- sit in that block together while the matrix pipe idles (measured on the producer / consumer kernel, profiles/r03_mlp_ps_*).
template <int D_, int DBG, int GS, bool PROJ = false, int SPREAD = 0>
__global__ __launch_bounds__(512, 2) void mlp_fused_kernel(MlpArgs a) {
  constexpr int KS1 = D_ / BK;            // 12 k-steps of GEMM1
  constexpr int KQ = 4;                   // k-steps per GEMM1 stage
  constexpr int G1S = 2 * (KS1 / KQ);     // 6 GEMM1 stages per chunk: 2 pair-groups x 3 k-ranges
  constexpr int TH = D_ / RT;             // 3 output thirds of GEMM2
  constexpr int NT2 = D_ / 16;            // 24 output tiles per wave
  constexpr int G2S = (CH / BK) * TH;     // 6 GEMM2 stages per chunk: W2[128 outputs] x 32 hidden
  constexpr int SPC = G1S + G2S;          // ring stages per hidden chunk (12)
  constexpr int NG = NS / GS;             // groups in the ring: 1 being read, 1 cooling down, NG - 2 in flight
  constexpr int PD = (NG - 2) * GS;       // stages in flight after a refill
  static_assert(KS1 % KQ == 0 && D_ % RT == 0 && SPC % GS == 0 && NS % GS == 0 && NG >= 3, "layout");
  extern __shared__ __attribute__((aligned(16))) bf16_t sm[];
  float* sb1 = reinterpret_cast<float*>(sm + NS * TILE_EL);
  float* sb2 = sb1 + a.Hd;
  float* sgam = sb2 + D_;                 // LayerNorm affine, read per k-step in the prologue: 48 16-byte reads per lane that came from
  float* sbet = sgam + D_;                // L1 / L2 in groups of 8 behind a scheduling barrier = six exposed round trips per tile
  float* sbp = sbet + D_;                 // PROJ: bias of the attention projection
  float* sgn = sbp + D_;                  // PROJ + ln_next: the next block's norm1 affine
  float* sbn = sgn + D_;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // SGPR: LDS-DMA destinations (M0) stay scalar arithmetic
  const int l15 = lane & 15, lg = lane >> 4;
  const int nch = a.Hd / CH;
  for (int i = tid; i < a.Hd; i += 512) sb1[i] = a.b1[i];
  for (int i = tid; i < D_; i += 512) { sb2[i] = a.b2[i]; sgam[i] = a.gamma[i]; sbet[i] = a.beta[i]; }
  if constexpr (PROJ) {
    for (int i = tid; i < D_; i += 512) sbp[i] = a.bp[i];
    if (a.ln_next) for (int i = tid; i < D_; i += 512) { sgn[i] = a.gamma_n[i]; sbn[i] = a.beta_n[i]; }
  }
  __syncthreads();

  // ---- producer: one LDS-DMA instruction (16 LDS rows x 64 B) per wave and stage.  Buffer addressing: (SGPR resource
  // descriptor) + (32-bit per-lane byte offset) + (uniform byte offset in an SGPR); num_records = the weight's size, so an
  // offset of 2^31 is out of range: the load moves nothing but still counts in vmcnt.
  const int pr = 16 * wave + (lane >> 2);                        // LDS row of this lane's 16 B
  const int psl = ((lane & 3) ^ swz(pr)) * 8;
  // GEMM1 stage (u, kt): LDS row 32 s + h = W1[64 c + 32 u + h][32 (4 kt + s) ..],  s = k-step inside the stage, h < 32
  // GEMM2 stage (u, th): LDS row r = W2[128 th + r][64 c + 32 u ..]
  const int lo1 = ((pr & 31) * D_ + (pr >> 5) * BK + psl) * 2;
  const int lo2 = (pr * a.Hd + psl) * 2;
  const int pdst = 16 * wave * BK;
  const int wbytes = a.Hd * D_ * 2;
  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.W1), 0, wbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.W2), 0, wbytes, 0x00020000);
  constexpr int OOB = 0x7ffffff0;
  // PROJ: the ring sequence starts with VOFF = 3 "virtual chunks" of SPC stages that carry the projection weight Wp: stage j of virtual
  // chunk v is the LDS tile  r -> Wp[128 v + r][32 j ..]  (the shape of a GEMM2 stage with row pitch D); chunk indices below are virtual
  // (MLP chunk c = virtual chunk c + VOFF) so that slot arithmetic and the look-ahead of the refills run straight through the seam
  constexpr int VOFF = PROJ ? TH : 0;
  const int nchv = nch + VOFF;
  const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(PROJ ? a.Wp : a.W2), 0, D_ * D_ * 2, 0x00020000);
  const int lop = (pr * D_ + psl) * 2;
  auto issue = [&](int cv, int j, int slot) {                // stage j of virtual chunk cv (j is a compile-time constant at every call)
    if (DBG & 2) return;
    lds_void* dst = (lds_void*)(sm + slot * TILE_EL + pdst);
    const bool valid = cv < nchv;
    if (PROJ && cv < VOFF) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, dst, 16, lop, (cv * RT * D_ + j * BK) * 2, 0, 0);
      return;
    }
    const int c = cv - VOFF;
    if (j < G1S) {
      const int u = j / (KS1 / KQ), kt = j % (KS1 / KQ);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, dst, 16, valid ? lo1 : OOB, ((c * CH + 32 * u) * D_ + kt * KQ * BK) * 2, 0, 0);
    } else {
      const int jj = j - G1S, u = jj / TH, th = jj % TH;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, dst, 16, valid ? lo2 : OOB, (th * RT * a.Hd + c * CH + u * BK) * 2, 0, 0);
    }
  };

  // ---- consumer fragment offsets (elements) inside a ring tile
  // GEMM2 (and any plain tile): row l15 of 16-row tile t: + t * 16 * BK
  const int fo2 = l15 * BK + ((lg ^ swz(l15)) << 3);
  // GEMM1: A-tile row i = l15 <- hidden 8 (i>>2) + (i&3) (+4 for tile Q) of the 32-row sub-tile of k-step s: + s * 32 * BK
  const int h1 = 8 * (l15 >> 2) + (l15 & 3);
  const int fo1p = h1 * BK + ((lg ^ swz(h1)) << 3);
  const int fo1q = (h1 + 4) * BK + ((lg ^ swz(h1 + 4)) << 3);

  const int ntiles = (a.M + FBM - 1) / FBM;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m = tile * FBM + wave * 16 + l15;
    const int mc = min(m, a.M - 1);
    MDBG_T(0);
    // ---- LayerNorm of the wave's 16 rows, straight into MFMA B-fragments: lane holds x[m][32 s + 8 g .. +7], s = 0..11
    u32x4_t xn[KS1];
    if constexpr (!PROJ) {
      const float* xr = a.x + (size_t)mc * D_ + 8 * lg;
      f32x4_t v[2 * KS1];
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < KS1; ++k) {
        v[2 * k] = *reinterpret_cast<const f32x4_t*>(xr + 32 * k);
        v[2 * k + 1] = *reinterpret_cast<const f32x4_t*>(xr + 32 * k + 4);
      }
#pragma unroll
      for (int k = 0; k < 2 * KS1; ++k) s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      const float mu = s * (1.0f / D_);
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < 2 * KS1; ++k) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[k][e] - mu; q += d * d; }
      }
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      const float rs = rsqrtf(q * (1.0f / D_) + a.eps);
      if (a.save_rows > 0 && m < a.save_rows && lg == 0) { a.s_mean[m] = mu; a.s_rstd[m] = rs; }
#pragma unroll
      for (int k = 0; k < KS1; ++k) {
        const f32x4_t g0 = *reinterpret_cast<const f32x4_t*>(sgam + 32 * k + 8 * lg);
        const f32x4_t g1 = *reinterpret_cast<const f32x4_t*>(sgam + 32 * k + 8 * lg + 4);
        const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(sbet + 32 * k + 8 * lg);
        const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(sbet + 32 * k + 8 * lg + 4);
        const f32x4_t p = v[2 * k], r = v[2 * k + 1];
        xn[k] = u32x4_t{pack_bf2((p[0] - mu) * rs * g0[0] + b0[0], (p[1] - mu) * rs * g0[1] + b0[1]),
                        pack_bf2((p[2] - mu) * rs * g0[2] + b0[2], (p[3] - mu) * rs * g0[3] + b0[3]),
                        pack_bf2((r[0] - mu) * rs * g1[0] + b1[0], (r[1] - mu) * rs * g1[1] + b1[1]),
                        pack_bf2((r[2] - mu) * rs * g1[2] + b1[2], (r[3] - mu) * rs * g1[3] + b1[3])};
        if (k & 1) __builtin_amdgcn_sched_barrier(0);     // keep the scheduler from hoisting all 48 affine loads (spills)
      }
    }
    const bool save_tile = !PROJ && a.save_rows > 0 && tile * FBM + wave * 16 < a.save_rows;      // wave-uniform
    if (save_tile && m < a.save_rows) {                                                     // norm2 output of the gradient rows
      bf16_t* lr = a.s_ln2 + (size_t)m * D_ + 8 * lg;
#pragma unroll
      for (int k = 0; k < KS1; ++k) *reinterpret_cast<u32x4_t*>(lr + 32 * k) = xn[k];
    }
    // PROJ: the attention output rows of the wave as MFMA B-fragments (lane holds ao[m][32 s + 8 g .. + 7])
    u32x4_t aof[PROJ ? KS1 : 1];
    if constexpr (PROJ) {
      const bf16_t* ar = a.ao + (size_t)mc * D_ + 8 * lg;
#pragma unroll
      for (int k = 0; k < KS1; ++k) aof[k] = *reinterpret_cast<const u32x4_t*>(ar + 32 * k);
    }
    // PROJ: the residual stream never makes a round trip through memory inside the launch.  The rows' x enters as the START VALUE of the
    // output accumulators (lane (l15 = row, lg) holds columns 16 t + 4 lg + r), the projection accumulates on top of it, and the same
    // registers, then holding x1, are the start value of fc2.  The per-row drop-path factors (vit.py:163, :165; 0 or 1 / keep_prob) ride on
    // the B operands of the two products -- the bf16 attention-output fragments and the bf16 GELU output -- since a B column is one row:
    // a factor of 1 (block 0, eval) is exact, 0 leaves x untouched, any other means bf16(rs * value) instead of rs * bf16(value): for GELU
    // the ONE rounding of the operand either way; for ao a second one unless the attention launch applied rs1 before ITS rounding
    // (ao_scaled: srhip_attn_block_fused out_scale).
    float rs1v = 1.0f, rs2v = 1.0f;
    f32x4_t acc2[NT2];
    if constexpr (PROJ) {
      if (a.row_scale1) rs1v = a.row_scale1[mc / a.rows_per_sample];
      if (a.row_scale) rs2v = a.row_scale[mc / a.rows_per_sample];
      const float* xr = a.x + (size_t)mc * D_ + 4 * lg;
#pragma unroll
      for (int t = 0; t < NT2; ++t) acc2[t] = *reinterpret_cast<const f32x4_t*>(xr + 16 * t);
      if (rs1v != 1.0f && !a.ao_scaled) {
#pragma unroll
        for (int k = 0; k < KS1; ++k)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            aof[k][e] = pack_bf2(rs1v * __builtin_bit_cast(float, aof[k][e] << 16), rs1v * __builtin_bit_cast(float, aof[k][e] & 0xffff0000u));
      }
    }
    __syncthreads();                       // previous tile's ring reads are over (and sb1/sb2 are written)
    MDBG_T(1);
#pragma unroll
    for (int p = 0; p < PD; ++p) issue(p / SPC, p % SPC, p);   // groups 0 .. NG-3

    __builtin_amdgcn_sched_barrier(0);     // the accumulators must not become live (zeroed early) across the LayerNorm above
    if constexpr (!PROJ) {
#pragma unroll
      for (int t = 0; t < NT2; ++t) acc2[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    } else {                               // + rs1 * bp while the first ring stages are in flight
#pragma unroll
      for (int t = 0; t < NT2; ++t) {
        const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(sbp + 16 * t + 4 * lg);
        acc2[t][0] += rs1v * bb[0]; acc2[t][1] += rs1v * bb[1]; acc2[t][2] += rs1v * bb[2]; acc2[t][3] += rs1v * bb[3];
      }
    }
    f32x4_t acc1[4];                       // [2 u + {P, Q}], started at the fc1 bias of the tile's hidden units
    u32x4_t hf[2];
    u32x4_t fa0[4], fa1[4];                // A fragments of half-stage h of stage j in fa[h]: 4 fragments = 4 MFMAs

    // Group sync before the first read of group (c * SPC + j) / GS: own DMA parts of the group have landed (counted vmcnt:
    // the NG - 3 younger groups may still be in flight), barrier (everybody's parts have), then refill the group that was
    // consumed two syncs ago.
    auto sync_group = [&](auto jc, int c) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      static_assert(j % GS == 0, "group start");
      if constexpr ((DBG & 2) == 0) wait_vm<GS*(NG - 3)>();
      __builtin_amdgcn_s_barrier();
      if constexpr (!SPREAD) {
#pragma unroll
        for (int i = 0; i < GS; ++i) {
          const int jn = (j + PD + i) % SPC, cn = c + (j + PD + i) / SPC;
          issue(cn, jn, (c * SPC + j + PD + i) & (NS - 1));
        }
      }
    };
    // SPREAD: stage j of (virtual) chunk c asks for stage j + PD (its slot held stage j + PD - NS, consumed before the last group barrier)
    auto issue_ahead = [&](auto jc, int c, bool late) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      // (tried: the SIMD partners -- waves 4-7 -- issuing behind the stage's second MFMA half instead: 30 spilled registers, not built)
      if constexpr (SPREAD == 1) { if (!late) issue(c + (j + PD) / SPC, (j + PD) % SPC, (c * SPC + j + PD) & (NS - 1)); }
    };
    // half-stage (j, h): GEMM1: k-steps 2h, 2h+1 of the stage x tiles P, Q; GEMM2: output tiles 4h .. 4h+3 of the third
    auto read_half = [&](auto jc, auto hc, int c) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value, h = decltype(hc)::value;
      u32x4_t(&fa)[4] = h ? fa1 : fa0;
      const bf16_t* st = sm + ((c * SPC + j) & (NS - 1)) * TILE_EL;
      if constexpr ((DBG & 4) != 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) fa[t] = u32x4_t{(unsigned)(c + t), 1u, 2u, (unsigned)j};
      } else if constexpr (j < G1S) {
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
          fa[2 * s_] = *reinterpret_cast<const u32x4_t*>(st + fo1p + (2 * h + s_) * 32 * BK);
          fa[2 * s_ + 1] = *reinterpret_cast<const u32x4_t*>(st + fo1q + (2 * h + s_) * 32 * BK);
        }
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) fa[t] = *reinterpret_cast<const u32x4_t*>(st + fo2 + (4 * h + t) * 16 * BK);
      }
    };
    // bias + GELU + bf16 pack, ONE element at a time so that it can be threaded between MFMAs: element e = 4 X + r of
    // pair-group u is hidden unit 64 c + 32 u + 8 g + e (X = 0: tile P, 1: tile Q; r = accumulator register).
    float gcarry = 0.f, pcarry = 0.f;
    // accumulator start = fc1 bias: lane (g) holds hidden 64 c + 32 u + 8 g + 4 X + r in register r of tile X
    auto acc1_start = [&](auto uc, auto xc, int c) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value, X = decltype(xc)::value;
      acc1[2 * u + X] = *reinterpret_cast<const f32x4_t*>(sb1 + min(c - VOFF, nch - 1) * CH + 32 * u + 8 * lg + 4 * X);
    };
    auto gelu_elem = [&](auto uc, auto ec, int c) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value, e = decltype(ec)::value, X = e >> 2, r = e & 3;
      const float v = acc1[2 * u + X][r];
      if constexpr (PROJ && (DBG & 1) == 0) {
        // rows without a backward: the two values of a bf16 pair together, packed fp32 math, no transcendentals (gelu_poly2); the row's
        // drop-path factor (lane column of the B fragment) rides on the result before its rounding
        if constexpr ((e & 1) == 1) {
          const f32x2_t g2 = gelu_poly2(f32x2_t{acc1[2 * u + X][r - 1], v}) * f32x2_t{rs2v, rs2v};
          hf[u][e >> 1] = pack_bf2(g2[0], g2[1]);
        }
      } else {
      float gv;
      if constexpr ((DBG & 1) != 0) gv = v; else gv = gelu_erf(v);
      if constexpr (PROJ) gv *= rs2v;          // drop-path factor of the row (lane column of the B fragment)
      if constexpr ((e & 1) == 0) { gcarry = gv; pcarry = v; } else hf[u][e >> 1] = pack_bf2(gcarry, gv);
      }
      if constexpr ((e & 1) == 1) {
        // gradient rows: keep the fc1 pre-activation and the GELU output (bf16, as the unfused path saves them); rare (8 % of the
        // tiles) and conservative for the counted vmcnt waits (extra younger stores can only make a wait longer)
        if (save_tile && m < a.save_rows) {
          const size_t o = (size_t)m * a.Hd + (c - VOFF) * CH + 32 * u + 8 * lg + (e - 1);
          *reinterpret_cast<uint32_t*>(a.s_pre + o) = pack_bf2(pcarry, v);
          *reinterpret_cast<uint32_t*>(a.s_h + o) = hf[u][e >> 1];
        }
      }
      if constexpr (r == 3) acc1_start(uc, std::integral_constant<int, X>{}, c + 1);     // restart for the next chunk
    };
    // Which GELU elements ride behind half-stage (j, h): group 0 (complete after stage 2) behind stages 3-5, group 1
    // (complete after stage 5, needed by stage 9) behind stages 6-8; 2,1,1,2,1,1 elements per half-stage.
    auto gelu_slot = [&](auto jc, auto hc, auto pc, int c) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value, h = decltype(hc)::value, part = decltype(pc)::value;
      if constexpr (j >= 3 && j <= 8 && (DBG & 8) == 0) {
        constexpr int u = (j - 3) / 3, q = 2 * ((j - 3) % 3) + h;           // q = 0..5
        constexpr int first[7] = {0, 2, 3, 4, 6, 7, 8};
        constexpr int e0 = first[q], n = first[q + 1] - first[q];
        if constexpr (part < n) gelu_elem(std::integral_constant<int, u>{}, std::integral_constant<int, e0 + part>{}, c);
      }
    };
    auto mfma_one = [&](auto jc, auto hc, auto tc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value, h = decltype(hc)::value, t = decltype(tc)::value;
      u32x4_t(&fa)[4] = h ? fa1 : fa0;
      if constexpr ((DBG & 8) != 0) {
        asm volatile("" ::"v"(fa[t]));
        if constexpr (j == G1S - 1) { hf[0] = fa[0]; hf[1] = fa[1]; }
      } else if constexpr (j < G1S) {
        constexpr int u = j / (KS1 / KQ), kt = j % (KS1 / KQ);
        acc1[2 * u + (t & 1)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
            __builtin_bit_cast(bf16x8_t, fa[t]), __builtin_bit_cast(bf16x8_t, xn[kt * KQ + 2 * h + (t >> 1)]), acc1[2 * u + (t & 1)],
            0, 0, 0);
      } else {
        constexpr int u = (j - G1S) / TH, th = (j - G1S) % TH;
        acc2[th * 8 + 4 * h + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
            __builtin_bit_cast(bf16x8_t, fa[t]), __builtin_bit_cast(bf16x8_t, hf[u]), acc2[th * 8 + 4 * h + t], 0, 0, 0);
      }
    };
    // half-stage (j, h): 4 MFMAs with up to two GELU elements threaded between them (VALU work in the shadow of the matrix
    // pipe; in one lump it left the pipe idle ~25 % of the time because both waves of a SIMD reach it together)
    auto mfma_half = [&](auto jc, auto hc, int c) __attribute__((always_inline)) {     // c = chunk the stage belongs to
      using T0 = std::integral_constant<int, 0>;
      using T1 = std::integral_constant<int, 1>;
      using T2 = std::integral_constant<int, 2>;
      using T3 = std::integral_constant<int, 3>;
      mfma_one(jc, hc, T0{});
      mfma_one(jc, hc, T1{});
      __builtin_amdgcn_sched_barrier(0);
      gelu_slot(jc, hc, T0{}, c);
      __builtin_amdgcn_sched_barrier(0);
      mfma_one(jc, hc, T2{});
      mfma_one(jc, hc, T3{});
      __builtin_amdgcn_sched_barrier(0);
      gelu_slot(jc, hc, T1{}, c);
      __builtin_amdgcn_sched_barrier(0);
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    if constexpr (PROJ) {
      // ---- attention output projection: VOFF virtual chunks of SPC stages, stage (v, j) = Wp rows 128 v .. x k-step j, the GEMM2 stage
      // shape: 8 MFMAs acc2[8 v + t] += Wp tile t . ao fragment j, same software pipeline as below
      auto read_half_p = [&](auto jc, auto hc, int cv) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value, h = decltype(hc)::value;
        u32x4_t(&fa)[4] = h ? fa1 : fa0;
        const bf16_t* st = sm + ((cv * SPC + j) & (NS - 1)) * TILE_EL;
#pragma unroll
        for (int t = 0; t < 4; ++t) fa[t] = *reinterpret_cast<const u32x4_t*>(st + fo2 + (4 * h + t) * 16 * BK);
      };
      sync_group(H0{}, 0);
      read_half_p(H0{}, H0{}, 0);
      static_for<VOFF>([&](auto vc) __attribute__((always_inline)) {
        constexpr int v = decltype(vc)::value;
        static_for<SPC>([&](auto jc) __attribute__((always_inline)) {
          constexpr int j = decltype(jc)::value;
          constexpr int jn = (j + 1) % SPC, vn = v + (j + 1) / SPC;
          read_half_p(jc, H1{}, v);
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc2[v * 8 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa0[t]), __builtin_bit_cast(bf16x8_t, aof[j]),
                                                                       acc2[v * 8 + t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          issue_ahead(jc, v, false);
          if constexpr (jn % GS == 0) sync_group(std::integral_constant<int, jn>{}, vn);
          if constexpr (vn == VOFF) read_half(std::integral_constant<int, 0>{}, H0{}, vn);       // the seam: first MLP stage (GEMM1 fragments)
          else read_half_p(std::integral_constant<int, jn>{}, H0{}, vn);
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc2[v * 8 + 4 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa1[t]), __builtin_bit_cast(bf16x8_t, aof[j]),
                                                                           acc2[v * 8 + 4 + t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          issue_ahead(jc, v, true);
        });
      });
      MDBG_T(4);
      // ---- first residual + LayerNorm, from the accumulator layout: lane (l15 = row, lg) holds columns 16 t + 4 lg + r of its row.
      //   acc2 = x + rs1 * (ao Wp^T + bp) = x1 (vit.py:163) stays in the accumulators as the start value of fc2 and is normalised from
      //   there; the bf16 pairs are moved into the B-fragment layout (lane holds columns 32 s + 8 lg .. + 7) by two row
      //   swaps per register: for the tile pair (2 s, 2 s + 1)
      //   permlane32_swap(P, Q) = {P0 P1 Q0 Q1, P2 P3 Q2 Q3} (rows = lane groups), then permlane16_swap of those two
      //   = {P0 P2 Q0 Q2, P1 P3 Q1 Q3}: exactly columns 8 lg .. + 3 and 8 lg + 4 .. + 7 of the k-step for lane group lg.
      {
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT2; ++t) sum += (acc2[t][0] + acc2[t][1]) + (acc2[t][2] + acc2[t][3]);
        const float mu = rows_sum4(sum) * (1.0f / D_);
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < NT2; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float d = acc2[t][e] - mu; q += d * d; }
        const float rs = rsqrtf(rows_sum4(q) * (1.0f / D_) + a.eps);
        // (the 48 affine reads must not all be issued up front: 192 live registers = spills whose reloads stall the weight ring; the offset
        // of every k-step is made to depend on the fragment the previous one produced)
        int ko = 4 * lg;
        asm volatile("" : "+v"(ko) : "v"(rs));
#pragma unroll
        for (int sI = 0; sI < KS1; ++sI) {
          if (sI > 0) asm volatile("" : "+v"(ko) : "v"(xn[sI - 1][3]));
          unsigned pk[2][2];                         // [tile of the pair][dword]
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int t = 2 * sI + e;
            const f32x4_t g4 = *reinterpret_cast<const f32x4_t*>(sgam + 16 * t + ko);
            const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(sbet + 16 * t + ko);
            pk[e][0] = pack_bf2((acc2[t][0] - mu) * rs * g4[0] + b4[0], (acc2[t][1] - mu) * rs * g4[1] + b4[1]);
            pk[e][1] = pack_bf2((acc2[t][2] - mu) * rs * g4[2] + b4[2], (acc2[t][3] - mu) * rs * g4[3] + b4[3]);
          }
          unsigned out4[4];
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const auto w32 = __builtin_amdgcn_permlane32_swap(pk[0][d], pk[1][d], false, false);
            const auto w16 = __builtin_amdgcn_permlane16_swap(w32[0], w32[1], false, false);
            out4[d] = w16[0];                        // columns 8 lg + 2 d, + 1
            out4[2 + d] = w16[1];                    // columns 8 lg + 4 + 2 d, + 1
          }
          xn[sI] = u32x4_t{out4[0], out4[1], out4[2], out4[3]};
          if (sI & 1) __builtin_amdgcn_sched_barrier(0);
        }
      }
      MDBG_T(5);
      // fc2 accumulates on x1 + rs2 * b2: the block's output needs no epilogue arithmetic
#pragma unroll
      for (int t = 0; t < NT2; ++t) {
        const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(sb2 + 16 * t + 4 * lg);
        acc2[t][0] += rs2v * bb[0]; acc2[t][1] += rs2v * bb[1]; acc2[t][2] += rs2v * bb[2]; acc2[t][3] += rs2v * bb[3];
      }
    }
    {
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      acc1_start(I0{}, I0{}, VOFF); acc1_start(I0{}, I1{}, VOFF); acc1_start(I1{}, I0{}, VOFF); acc1_start(I1{}, I1{}, VOFF);
    }
    if constexpr (!PROJ) {
      sync_group(H0{}, 0);
      read_half(H0{}, H0{}, 0);
    }
    for (int c = VOFF; c < nchv; ++c)
      static_for<SPC>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        constexpr int jn = (j + 1) % SPC;
        const int cn = c + (j + 1) / SPC;
        read_half(jc, H1{}, c);
        mfma_half(jc, H0{}, c);
        issue_ahead(jc, c, false);
        if constexpr (jn % GS == 0) sync_group(std::integral_constant<int, jn>{}, cn);
        read_half(std::integral_constant<int, jn>{}, H0{}, cn);  // (one stage past the end: a stale slot, never multiplied)
        mfma_half(jc, H1{}, c);
        issue_ahead(jc, c, true);
      });
    MDBG_T(2);
    // ---- epilogue: lane holds y[m][16 t + 4 g + r]
    if (PROJ && a.ln_next) {                // wave-uniform: the rows leave as fp32 (residual stream) AND normalised for the next block
      float* xw = a.xo + (size_t)mc * D_ + 4 * lg;
      float sum = 0.f;
#pragma unroll
      for (int t = 0; t < NT2; ++t) {
        const f32x4_t xv = acc2[t];
        if (m < a.M) *reinterpret_cast<f32x4_t*>(xw + 16 * t) = xv;
        sum += (xv[0] + xv[1]) + (xv[2] + xv[3]);
      }
      const float mu = rows_sum4(sum) * (1.0f / D_);
      float q = 0.f;
#pragma unroll
      for (int t = 0; t < NT2; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = acc2[t][e] - mu; q += d * d; }
      const float rs = rsqrtf(rows_sum4(q) * (1.0f / D_) + a.eps);
      bf16_t* lw = a.ln_next + (size_t)mc * D_ + 8 * lg;
      int ko = 4 * lg;
      asm volatile("" : "+v"(ko) : "v"(rs));
      unsigned chain = 0;
#pragma unroll
      for (int sI = 0; sI < KS1; ++sI) {     // same re-layout as after the projection: 16 bytes per lane and k-step leave
        if (sI > 0) asm volatile("" : "+v"(ko) : "v"(chain));
        unsigned pk[2][2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int t = 2 * sI + e;
          const f32x4_t g4 = *reinterpret_cast<const f32x4_t*>(sgn + 16 * t + ko);
          const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(sbn + 16 * t + ko);
          pk[e][0] = pack_bf2((acc2[t][0] - mu) * rs * g4[0] + b4[0], (acc2[t][1] - mu) * rs * g4[1] + b4[1]);
          pk[e][1] = pack_bf2((acc2[t][2] - mu) * rs * g4[2] + b4[2], (acc2[t][3] - mu) * rs * g4[3] + b4[3]);
        }
        unsigned out4[4];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const auto w32 = __builtin_amdgcn_permlane32_swap(pk[0][d], pk[1][d], false, false);
          const auto w16 = __builtin_amdgcn_permlane16_swap(w32[0], w32[1], false, false);
          out4[d] = w16[0];
          out4[2 + d] = w16[1];
        }
        chain = out4[3];
        if (m < a.M) *reinterpret_cast<u32x4_t*>(lw + 32 * sI) = u32x4_t{out4[0], out4[1], out4[2], out4[3]};
      }
    } else if (PROJ) {
      float* xw = a.xo + (size_t)mc * D_ + 4 * lg;
      if (m < a.M) {
#pragma unroll
        for (int t = 0; t < NT2; ++t) *reinterpret_cast<f32x4_t*>(xw + 16 * t) = acc2[t];
      }
    } else if (m < a.M) {
      const float rsc = a.row_scale ? a.row_scale[m / a.rows_per_sample] : 1.0f;
      const float* xr = a.x + (size_t)m * D_ + 4 * lg;
      float* xw = a.xo + (size_t)m * D_ + 4 * lg;
#pragma unroll
      for (int t = 0; t < NT2; ++t) {
        const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(sb2 + 16 * t + 4 * lg);
        f32x4_t xv = *reinterpret_cast<const f32x4_t*>(xr + 16 * t);
        xv[0] += rsc * (acc2[t][0] + bb[0]); xv[1] += rsc * (acc2[t][1] + bb[1]);
        xv[2] += rsc * (acc2[t][2] + bb[2]); xv[3] += rsc * (acc2[t][3] + bb[3]);
        *reinterpret_cast<f32x4_t*>(xw + 16 * t) = xv;
      }
    }
#ifdef SRHIP_TUNING
    __builtin_amdgcn_s_waitcnt(0);          // stores issued AND acknowledged
    __syncthreads();
    MDBG_T(3);
#endif
  }
}


__global__ void gelu_eval_kernel(const float* __restrict__ x, float* __restrict__ y_erf, float* __restrict__ y_poly, int n) {
  const int i = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  const float a = x[i], b = i + 1 < n ? x[i + 1] : 0.f;
  const f32x2_t g = gelu_poly2(f32x2_t{a, b});
  y_erf[i] = gelu_erf(a); y_poly[i] = g[0];
  if (i + 1 < n) { y_erf[i + 1] = gelu_erf(b); y_poly[i + 1] = g[1]; }
}

}  // namespace

#ifdef SRHIP_TUNING
extern "C" int srhip_mlp_debug(long long* out_host, int n) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(srhip_mlp_dbg), (size_t)n * sizeof(long long)) == hipSuccess ? SR_OK : SR_EINVAL;
}
#endif
// The two GELU evaluations of the library on n values: y_erf = gelu_erf (exact-erf GELU to 1.5e-7, every path with a backward and every unfused
// epilogue), y_poly = gelu_poly2 (the transcendental-free form inside srhip_mlp_fused_proj) -- so that a test can pin the distance between them.
extern "C" int srhip_gelu_eval(const float* x, float* y_erf, float* y_poly, int n, void* stream) {
  if (!x || !y_erf || !y_poly || n <= 0) return SR_EINVAL;
  SR_LAUNCH(gelu_eval_kernel, dim3(cdiv((n + 1) / 2, 256)), dim3(256), 0, (hipStream_t)stream, x, y_erf, y_poly, n);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
extern "C" int srhip_mlp_fused(const float* x, float* x_out, const float* ln_gamma, const float* ln_beta, float eps, const void* W1,
                               const float* b1, const void* W2, const float* b2, const float* row_scale, int rows_per_sample,
                               int save_rows, void* save_ln2, void* save_pre, void* save_h, float* save_mean, float* save_rstd,
                               int M, int D, int Hd, void* stream) {
  if (!x || !x_out || !ln_gamma || !ln_beta || !W1 || !b1 || !W2 || !b2 || M <= 0) return SR_EINVAL;
  if (D != 384 || Hd < 128 || (Hd % RT) || Hd > 4096) return SR_EINVAL;          // ViT-S width; hidden in 128-wide chunks
  if (row_scale && rows_per_sample <= 0) return SR_EINVAL;
  if (save_rows < 0 || save_rows > M || (save_rows > 0 && (!save_ln2 || !save_pre || !save_h || !save_mean || !save_rstd))) return SR_EINVAL;
  if (((uintptr_t)x | (uintptr_t)x_out | (uintptr_t)W1 | (uintptr_t)W2 | (uintptr_t)ln_gamma | (uintptr_t)ln_beta | (uintptr_t)save_ln2 |
       (uintptr_t)save_pre | (uintptr_t)save_h) & 15)
    return SR_EINVAL;
  MlpArgs a;
  a.x = x; a.xo = x_out; a.gamma = ln_gamma; a.beta = ln_beta; a.b1 = b1; a.b2 = b2; a.row_scale = row_scale;
  a.s_ln2 = (bf16_t*)save_ln2; a.s_pre = (bf16_t*)save_pre; a.s_h = (bf16_t*)save_h; a.s_mean = save_mean; a.s_rstd = save_rstd;
  a.save_rows = save_rows;
  a.W1 = (const bf16_t*)W1; a.W2 = (const bf16_t*)W2; a.eps = eps; a.M = M; a.Hd = Hd;
  a.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1;
  a.ao = nullptr; a.Wp = nullptr; a.bp = nullptr; a.row_scale1 = nullptr; a.ao_scaled = 0; a.ln_next = nullptr; a.gamma_n = a.beta_n = nullptr;
  const size_t smem = (size_t)NS * TILE_EL * sizeof(bf16_t) + (size_t)(Hd + 6 * D) * sizeof(float);
  void (*kern)(MlpArgs) = mlp_fused_kernel<384, 0, 4>;
#ifdef SRHIP_TUNING
  switch (SR_TUNE_ENV("SRHIP_MLP_DEBUG") ? atoi(SR_TUNE_ENV("SRHIP_MLP_DEBUG")) : 0) {
    case 1: kern = mlp_fused_kernel<384, 1, 4>; break;
    case 2: kern = mlp_fused_kernel<384, 2, 4>; break;
    case 3: kern = mlp_fused_kernel<384, 3, 4>; break;
    case 7: kern = mlp_fused_kernel<384, 7, 4>; break;
    case 11: kern = mlp_fused_kernel<384, 11, 4>; break;
    case 15: kern = mlp_fused_kernel<384, 15, 4>; break;
    case 100: kern = mlp_fused_kernel<384, 0, 2>; break;
    case 101: kern = mlp_fused_kernel<384, 0, 1>; break;
    default: break;
  }
#endif
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const int ntiles = cdiv(M, FBM);
  SR_LAUNCH(kern, dim3(min(ntiles, 256)), dim3(512), smem, (hipStream_t)stream, a);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

// Attention output projection + first residual + the whole MLP half of a block in ONE launch (rows without a backward):
//   x1 = x + row_scale1 * (ao Wp^T + bp);   x_out = x1 + row_scale2 * (fc2(GELU(fc1(LayerNorm(x1)))) + b2)
// (vit.py:163 tail: proj :105-106 + drop_path1 + residual, and :165).  Saves the proj GEMM launch and two reads + a write of the residual stream:
// x enters as the start value of the accumulators and x1 stays there.  ao_scaled != 0: ao already carries row_scale1 (only bp is scaled here).
// ln_next != NULL: also LayerNorm(x_out) with (next_gamma, next_beta) -- the next block's norm1 (vit.py:163) -- as bf16 [M, D].
extern "C" int srhip_mlp_fused_proj(const float* x, float* x_out, const void* ao, const void* Wp, const float* bp, const float* row_scale1,
                                    int ao_scaled, const float* ln_gamma, const float* ln_beta, float eps, const void* W1, const float* b1, const void* W2,
                                    const float* b2, const float* row_scale2, int rows_per_sample, void* ln_next, const float* next_gamma,
                                    const float* next_beta, int M, int D, int Hd, void* stream) {
  if (!x || !x_out || !ao || !Wp || !bp || !ln_gamma || !ln_beta || !W1 || !b1 || !W2 || !b2 || M <= 0) return SR_EINVAL;
  if (ln_next && (!next_gamma || !next_beta || ((uintptr_t)ln_next & 15))) return SR_EINVAL;
  if (D != 384 || Hd < 128 || (Hd % RT) || Hd > 4096) return SR_EINVAL;
  if ((row_scale1 || row_scale2) && rows_per_sample <= 0) return SR_EINVAL;
  if (((uintptr_t)x | (uintptr_t)x_out | (uintptr_t)ao | (uintptr_t)Wp | (uintptr_t)W1 | (uintptr_t)W2 | (uintptr_t)ln_gamma | (uintptr_t)ln_beta) & 15)
    return SR_EINVAL;
  MlpArgs a;
  a.x = x; a.xo = x_out; a.gamma = ln_gamma; a.beta = ln_beta; a.b1 = b1; a.b2 = b2; a.row_scale = row_scale2;
  a.s_ln2 = a.s_pre = a.s_h = nullptr; a.s_mean = a.s_rstd = nullptr; a.save_rows = 0;
  a.W1 = (const bf16_t*)W1; a.W2 = (const bf16_t*)W2; a.eps = eps; a.M = M; a.Hd = Hd;
  a.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1;
  a.ao = (const bf16_t*)ao; a.Wp = (const bf16_t*)Wp; a.bp = bp; a.row_scale1 = row_scale1; a.ao_scaled = ao_scaled;
  a.ln_next = (bf16_t*)ln_next; a.gamma_n = next_gamma; a.beta_n = next_beta;
  const size_t smem = (size_t)NS * TILE_EL * sizeof(bf16_t) + (size_t)(Hd + 6 * D) * sizeof(float);
  // SRHIP_MLP_SPREAD=0: all GS refills of a group right behind its barrier (the round-2 schedule; A/B on one box: 104 -> 99.5 us per 105-image
  // launch, 1547 -> 1574 img/s on the step)
  static const int spread = SR_TUNE_ENV("SRHIP_MLP_SPREAD") ? atoi(SR_TUNE_ENV("SRHIP_MLP_SPREAD")) : 1;
  void (*kern)(MlpArgs) = spread ? mlp_fused_kernel<384, 0, 4, true, 1> : mlp_fused_kernel<384, 0, 4, true, 0>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const int ntiles = cdiv(M, FBM);
  SR_LAUNCH(kern, dim3(min(ntiles, 256)), dim3(512), smem, (hipStream_t)stream, a);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
