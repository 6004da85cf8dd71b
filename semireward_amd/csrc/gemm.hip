// bf16 MFMA GEMM, "NT" form:  C[M,N] (+)= A[M,K] . B[N,K]^T   (both operands K-contiguous)
//
// Replaces the ATen addmm / matmul call sites of the reference backbone:
//   qkv / proj Linear  semilearn/nets/vit/vit.py:93-98,105      (K3, K5 in SURVEY.md 2c)
//   fc1 / GELU / fc2   semilearn/nets/vit/vit.py:69-75          (K6)
// and their autograd backward (dX = dY.W, dW = dY^T.X) -- the host side supplies W^T copies and
// transposed activations so that every product is an NT product.
//
// CDNA4 design: 128x128x32 tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 bf16 tiles,
// fp32 accumulate.  The MFMA a-operand is fed from the B matrix (rows n) and the b-operand from
// the A matrix (rows m), so each lane ends up with 4 CONSECUTIVE n for one m: the epilogue
// stores 8 B (bf16) / 16 B (fp32) contiguous per lane into the row-major C.
// LDS: 4-stage ring of [128][32] bf16 operand tiles (64 KiB -> 2 workgroups / CU) filled by LDS-DMA
// (global_load_lds_dwordx4): three K-tiles are always in flight ACROSS the per-step raw s_barrier, retired with
// counted s_waitcnt vmcnt(8/4/0) -- no register staging, no ds_write pass.  16-B slots are permuted per row
// (on the source address, as LDS-DMA writes lane-linear) so every ds_read_b128 lane group is conflict-free.
// Block->tile mapping is XCD-aware (consecutive N-tiles of one A row-panel share an XCD L2).
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <utility>

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
  int M, N, K, lda, ldb, ldc, ldaux, rows_per_sample, ksplit_tiles, debug;
  int wide_store;                      // EPI_BF16: 16-byte stores after a lane-group exchange (N % 64 == 0, ldc % 8 == 0)
  float alpha, beta;
  uint32_t drop_key, drop_thresh;      // RESID epilogue: nn.Dropout on (acc + bias) before the residual add (drop_thresh == 0: none)
  float drop_scale;                    // 1 / (1 - p)
  // RESID epilogue, optional: the residual that is read is the PRE-LayerNorm sum of the sub-layer before; its LayerNorm (post-LN encoders) is
  // applied on the way in -- (y - mean[m]) * rstd[m] * gamma[n] + beta[n] -- so that the LayerNorm launch need not write its fp32 output at all
  const float *ln_mean, *ln_rstd, *ln_gamma, *ln_beta;
};

// Internal epilogue id: SRHIP_EPI_RESID_F32 whose residual still owes its LayerNorm (srhip_gemm_nt_resid_ln_dropout).  An instantiation of its
// own, so that the plain residual kernels keep their register allocation (a run-time switch inside them spilled the 128 x 128 and the persistent
// kernel: tests/test_cpu_abi_and_host.py pins their budgets).
constexpr int EPI_RESID_LN = 5;
template <int E>
constexpr bool is_resid = E == SRHIP_EPI_RESID_F32 || E == EPI_RESID_LN;

// the residual quad of row m (valid or clamped), columns n .. n + 3 (clamped to N - 4 by the caller where it matters)
template <int EPI>
__device__ __forceinline__ f32x4_t ln_resid(const GemmArgs& g, int m, int n, f32x4_t x) {
  if constexpr (EPI == EPI_RESID_LN) {
    const float mu = g.ln_mean[m], rs = g.ln_rstd[m];
    const f32x4_t ga = *reinterpret_cast<const f32x4_t*>(g.ln_gamma + n), be = *reinterpret_cast<const f32x4_t*>(g.ln_beta + n);
    x[0] = (x[0] - mu) * rs * ga[0] + be[0]; x[1] = (x[1] - mu) * rs * ga[1] + be[1];
    x[2] = (x[2] - mu) * rs * ga[2] + be[2]; x[3] = (x[3] - mu) * rs * ga[3] + be[3];
  }
  return x;
}

constexpr int BM = 128, BN = 128, BK = 32, NS = 3, PD = NS - 1;
constexpr int TILE = BM * BK;            // elements of one operand tile of one stage
constexpr int STAGE = 2 * TILE;          // A tile then B tile

// 16-B slot swizzle inside a 64-B (32 x bf16) LDS row: physical slot = logical slot ^ PI(row), PI = [0,2,3,1][(row>>2)&3].
// ds_read_b128 is serviced in 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS): with this
// permutation the 16 (row, slot) pairs of every group fall on 16 distinct 4-bank slots -> conflict-free fragment reads.
__device__ __forceinline__ int swz(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <int N_>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

// Epilogue of one lane-quad: v[0..3] = C[m][n..n+3] accumulators (bias not yet added).
template <int EPI>
__device__ __forceinline__ void epi_store(const GemmArgs& g, int m, int n, float (&v)[4], float rs, bool atomic_f32) {
  if (g.debug & 1) {                      // tuning only (SRHIP_DEBUG=1): no epilogue memory traffic
    if (v[0] == 123456.75f) reinterpret_cast<float*>(g.C)[0] = v[1] + v[2] + v[3];
    return;
  }
  if (g.bias) {
    const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(g.bias + n);
    v[0] += b4[0]; v[1] += b4[1]; v[2] += b4[2]; v[3] += b4[3];
  }
  const size_t off = (size_t)m * g.ldc + n;
  if (EPI == SRHIP_EPI_BF16) {
    u32x2_t o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
    *reinterpret_cast<u32x2_t*>(reinterpret_cast<bf16_t*>(g.C) + off) = o;
  } else if (EPI == SRHIP_EPI_GELU_BF16) {
    if (g.aux_out) {
      u32x2_t p = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
      *reinterpret_cast<u32x2_t*>(g.aux_out + (size_t)m * g.ldaux + n) = p;
    }
    float h[4] = {gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3])};
    if (g.drop_thresh) {                 // (uniform) Wav2Vec2FeedForward.intermediate_dropout: dropout(GELU(dense(x))); aux_out keeps the pre-activation
      const uint32_t i0 = (uint32_t)m * (uint32_t)g.N + (uint32_t)n;
      bool dk4[4];
      drop_keep4(i0, g.drop_key, g.drop_thresh, dk4);
#pragma unroll
      for (int r = 0; r < 4; ++r) h[r] = dk4[r] ? h[r] * g.drop_scale : 0.f;
    }
    u32x2_t o = {pack_bf2(h[0], h[1]), pack_bf2(h[2], h[3])};
    *reinterpret_cast<u32x2_t*>(reinterpret_cast<bf16_t*>(g.C) + off) = o;
  } else if (is_resid<EPI>) {
    // residual source: C itself (in place) or, when the pre-block stream is kept for the backward, aux_in (fp32, ldaux)
    f32x4_t* cp = reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(g.C) + off);
    f32x4_t x = g.aux_in ? *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(g.aux_in) + (size_t)m * g.ldaux + n) : *cp;
    x = ln_resid<EPI>(g, m, n, x);
    if (g.drop_thresh) {                 // (uniform) BertSelfOutput / BertOutput: LayerNorm(x + dropout(dense(.)))
      const uint32_t i0 = (uint32_t)m * (uint32_t)g.N + (uint32_t)n;
      bool dk4[4];
      drop_keep4(i0, g.drop_key, g.drop_thresh, dk4);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = dk4[r] ? v[r] * g.drop_scale : 0.f;
    }
    x[0] += rs * v[0]; x[1] += rs * v[1]; x[2] += rs * v[2]; x[3] += rs * v[3];
    *cp = x;
  } else if (EPI == SRHIP_EPI_DGELU_BF16) {
    const u32x2_t p = *reinterpret_cast<const u32x2_t*>(g.aux_in + (size_t)m * g.ldaux + n);
    const float p0 = bf2f((bf16_t)(p[0] & 0xffff)), p1 = bf2f((bf16_t)(p[0] >> 16));
    const float p2 = bf2f((bf16_t)(p[1] & 0xffff)), p3 = bf2f((bf16_t)(p[1] >> 16));
    if (g.drop_thresh) {                 // adjoint of the dropout that followed the GELU
      const uint32_t i0 = (uint32_t)m * (uint32_t)g.N + (uint32_t)n;
      bool dk4[4];
      drop_keep4(i0, g.drop_key, g.drop_thresh, dk4);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = dk4[r] ? v[r] * g.drop_scale : 0.f;
    }
    u32x2_t o = {pack_bf2(v[0] * gelu_erf_grad(p0), v[1] * gelu_erf_grad(p1)),
                 pack_bf2(v[2] * gelu_erf_grad(p2), v[3] * gelu_erf_grad(p3))};
    *reinterpret_cast<u32x2_t*>(reinterpret_cast<bf16_t*>(g.C) + off) = o;
  } else if (atomic_f32) {  // SRHIP_EPI_F32 with split-K: C += alpha*acc (beta == 1 by contract)
    float* cp = reinterpret_cast<float*>(g.C) + off;
#pragma unroll
    for (int r = 0; r < 4; ++r) atomicAdd(cp + r, g.alpha * v[r]);
  } else {  // SRHIP_EPI_F32: C = alpha*acc + beta*C
    f32x4_t* cp = reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(g.C) + off);
    f32x4_t x = {g.alpha * v[0], g.alpha * v[1], g.alpha * v[2], g.alpha * v[3]};
    if (g.beta != 0.0f) {
      const f32x4_t c = *cp;
      x[0] += g.beta * c[0]; x[1] += g.beta * c[1]; x[2] += g.beta * c[2]; x[3] += g.beta * c[3];
    }
    *cp = x;
  }
}

// The bf16 epilogues as a value: the packed output quad C[m][n..n+3] of one lane (bias, GELU / GELU', dropout applied; the GELU
// pre-activation copy is stored on the way).  m is a valid row (callers clamp it for lanes past M; such lanes only take part in the exchange).
template <int EPI>
__device__ __forceinline__ u32x2_t epi_quad_bf16(const GemmArgs& g, int m, int n, float (&v)[4], bool row_ok, const f32x4_t* bpre = nullptr) {
  if (g.bias) {
    const f32x4_t b4 = bpre ? *bpre : *reinterpret_cast<const f32x4_t*>(g.bias + n);
    v[0] += b4[0]; v[1] += b4[1]; v[2] += b4[2]; v[3] += b4[3];
  }
  if (EPI == SRHIP_EPI_GELU_BF16) {
    if (g.aux_out && row_ok) {
      u32x2_t p = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
      *reinterpret_cast<u32x2_t*>(g.aux_out + (size_t)m * g.ldaux + n) = p;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
  } else if (EPI == SRHIP_EPI_DGELU_BF16) {
    const u32x2_t p = *reinterpret_cast<const u32x2_t*>(g.aux_in + (size_t)m * g.ldaux + n);
    const float pv[4] = {bf2f((bf16_t)(p[0] & 0xffff)), bf2f((bf16_t)(p[0] >> 16)), bf2f((bf16_t)(p[1] & 0xffff)), bf2f((bf16_t)(p[1] >> 16))};
    if (g.drop_thresh) {
      const uint32_t i0 = (uint32_t)m * (uint32_t)g.N + (uint32_t)n;
      bool dk4[4];
      drop_keep4(i0, g.drop_key, g.drop_thresh, dk4);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = dk4[r] ? v[r] * g.drop_scale : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] *= gelu_erf_grad(pv[r]);
  }
  if (EPI == SRHIP_EPI_GELU_BF16 && g.drop_thresh) {
    const uint32_t i0 = (uint32_t)m * (uint32_t)g.N + (uint32_t)n;
    bool dk4[4];
    drop_keep4(i0, g.drop_key, g.drop_thresh, dk4);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = dk4[r] ? v[r] * g.drop_scale : 0.f;
  }
  return u32x2_t{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
}

// Widened bf16 stores.  A lane's quad is 8 bytes, so the plain epilogue issues one store of 16 rows x 32 B per 16x16 tile, and the stamps of
// the tuning build put 5 us of a 12-us 128x128 tile there -- store ISSUE, not bandwidth.  v_permlane16_swap trades the quads of two
// neighbouring 16-column tiles (a, b) between lane groups 1 <-> 0 and 3 <-> 2: afterwards groups 0 / 2 hold columns 0-7 / 8-15 of tile a and
// groups 1 / 3 the same of tile b -- ONE 16-byte store per lane and tile pair (lane semantics probed on hardware).  Every lane executes the
// exchange (it is wave-wide); only the store is predicated on the row.  nb = first column of tile a.
__device__ __forceinline__ void store_quad_pair(bf16_t* C, int ldc, int m, bool row_ok, int nb, int lg, u32x2_t qa, u32x2_t qb) {
  const auto rx = __builtin_amdgcn_permlane16_swap(qa[0], qb[0], false, false);
  const auto ry = __builtin_amdgcn_permlane16_swap(qa[1], qb[1], false, false);
  const int col = nb + (lg & 1) * 16 + (lg >> 1) * 8;
  if (row_ok) *reinterpret_cast<u32x4_t*>(C + (size_t)m * ldc + col) = u32x4_t{rx[0], ry[0], rx[1], ry[1]};
}

#ifdef SRHIP_TUNING
__device__ long long srhip_gemm_dbg[4 * 8192];
#define GDBG_T(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) srhip_gemm_dbg[4 * blockIdx.x + (i)] = wall_clock64(); } while (0)
#define PDBG_T(i) do { if (threadIdx.x == 0 && blockIdx.x < 2048) srhip_gemm_dbg[16 * blockIdx.x + (i)] = wall_clock64(); } while (0)
#else
#define GDBG_T(i) do { } while (0)
#define PDBG_T(i) do { } while (0)
#endif
// One 128x128 output tile, K-tiles [kt0, kt1).  smem: NS * STAGE elements (the kernel's ONE __shared__ object: a second
// one makes hipcc drain vmcnt before every ds_read of an LDS-DMA pipeline).
template <int EPI>
__device__ __forceinline__ void gemm_tile_body(const GemmArgs& g, bf16_t* smem, int m0, int n0, int kt0, int kt1, bool atomic_f32) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lg = lane >> 4;

  // ---- LDS-DMA staging (global_load_lds, 16 B/lane): a wave-instruction fills 16 rows x 64 B = 1 KiB, lane l lands on
  // row l>>2, physical slot l&3; the swizzle therefore goes on the per-lane SOURCE address (guide rule 21).
  // Every wave issues 2 instructions for the A tile and 2 for the B tile of a stage: 4 per stage -> counted vmcnt below.
  size_t goa[2], gob[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 32 * wave + 16 * i + (lane >> 2);
    const int s = (lane & 3) ^ swz(r);
    goa[i] = (size_t)min(m0 + r, g.M - 1) * g.lda + s * 8;
    gob[i] = (size_t)min(n0 + r, g.N - 1) * g.ldb + s * 8;
  }
#define ISSUE(kt_, st_)                                                                                          \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                \
    bf16_t* da = smem + (st_) * STAGE + (32 * wave + 16 * i) * BK;                                               \
    __builtin_amdgcn_global_load_lds((gbl_void*)(g.A + goa[i] + (size_t)(kt_) * BK), (lds_void*)da, 16, 0, 0);   \
    __builtin_amdgcn_global_load_lds((gbl_void*)(g.B + gob[i] + (size_t)(kt_) * BK), (lds_void*)(da + TILE), 16, 0, 0); \
  }

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = kt1 - kt0;
  if (nk <= 0) return;
  // bias quads of the wave's four column tiles (they depend on n only): requested before the K loop instead of 16 times in the epilogue
  f32x4_t bq[4];
  constexpr bool BF16_OUT = EPI == SRHIP_EPI_BF16 || EPI == SRHIP_EPI_GELU_BF16 || EPI == SRHIP_EPI_DGELU_BF16;
  if (BF16_OUT && g.wide_store && g.bias) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bq[nt] = *reinterpret_cast<const f32x4_t*>(g.bias + n0 + wn * 64 + nt * 16 + lg * 4);
  }
  // RESID: the residual tile (fp32, 64 KiB per workgroup) is requested BEFORE the K loop so that its HBM latency hides under
  // the MFMA work; the epilogue then only adds and stores.  (The loads are older than every LDS-DMA op, so the counted
  // vmcnt waits below also cover them.)  SRHIP_DEBUG=2 (tuning) loads it in the epilogue instead: measured 1196 vs 1208 img/s with three
  // workgroups per CU, so the prefetch stays.
  f32x4_t res[4][4];
  // (a residual that still owes its LayerNorm is read in the epilogue instead: the prefetched form has no registers for it)
  const bool res_early = EPI == SRHIP_EPI_RESID_F32 && !(g.debug & 3);
  if (res_early) {
    const float* src = g.aux_in ? reinterpret_cast<const float*>(g.aux_in) : reinterpret_cast<const float*>(g.C);
    const int lds_ = g.aux_in ? g.ldaux : g.ldc;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int m = min(m0 + wm * 64 + mt * 16 + l15, g.M - 1);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int n = min(n0 + wn * 64 + nt * 16 + lg * 4, g.N - 4);
        res[nt][mt] = *reinterpret_cast<const f32x4_t*>(src + (size_t)m * lds_ + n);
      }
    }
  }
  // fragment read offsets (elements) inside a tile: row r, logical slot lg
  int fo_a[4], fo_b[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int rn = wn * 64 + t * 16 + l15, rm = wm * 64 + t * 16 + l15;
    fo_a[t] = TILE + rn * BK + ((lg ^ swz(rn)) << 3);     // MFMA a-operand <- B matrix rows (n)
    fo_b[t] = rm * BK + ((lg ^ swz(rm)) << 3);            // MFMA b-operand <- A matrix rows (m)
  }
  // prologue: PD tiles in flight
#pragma unroll
  for (int p = 0; p < PD; ++p)
    if (p < nk) { ISSUE(kt0 + p, p) }
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed once at most min(rem, PD-1) younger tiles (4 DMA ops each) are still outstanding
    const int rem = nk - 1 - kt;
    if (PD >= 3 && rem >= 2) WAIT_VM(8); else if (PD >= 2 && rem >= 1) WAIT_VM(4); else WAIT_VM(0);
    __builtin_amdgcn_s_barrier();       // every wave's share of tile kt is in LDS; stage (kt-1)%NS is no longer read
    if (kt + PD < nk) { ISSUE(kt0 + kt + PD, (kt + PD) % NS) }
    const bf16_t* st = smem + (kt % NS) * STAGE;
    s16x8_t fa[4], fb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      fa[t] = *reinterpret_cast<const s16x8_t*>(st + fo_a[t]);
      fb[t] = *reinterpret_cast<const s16x8_t*>(st + fo_b[t]);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
            __builtin_bit_cast(bf16x8_t, fa[nt]), __builtin_bit_cast(bf16x8_t, fb[mt]), acc[nt][mt], 0, 0, 0);
  }
#undef ISSUE
  GDBG_T(1);

  // ---- epilogue: lane holds C[m][n .. n+3], m = tile row (lane&15), n = 4*(lane>>4) + r
  if ((EPI == SRHIP_EPI_BF16 || EPI == SRHIP_EPI_GELU_BF16 || EPI == SRHIP_EPI_DGELU_BF16) && g.wide_store) {
    bf16_t* Cb = reinterpret_cast<bf16_t*>(g.C);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int mr = m0 + wm * 64 + mt * 16 + l15, m = min(mr, g.M - 1);
#pragma unroll
      for (int np = 0; np < 2; ++np) {
        u32x2_t q[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int nt = 2 * np + e;
          float v[4] = {acc[nt][mt][0], acc[nt][mt][1], acc[nt][mt][2], acc[nt][mt][3]};
          q[e] = epi_quad_bf16<EPI>(g, m, n0 + wn * 64 + nt * 16 + lg * 4, v, mr < g.M, &bq[nt]);
        }
        store_quad_pair(Cb, g.ldc, m, mr < g.M, n0 + wn * 64 + np * 32, lg, q[0], q[1]);
      }
    }
    return;
  }
  if (is_resid<EPI> && g.bias) {      // (the fragment registers of the K loop are free by now: no higher register peak)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bq[nt] = *reinterpret_cast<const f32x4_t*>(g.bias + min(n0 + wn * 64 + nt * 16 + lg * 4, g.N - 4));
  }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + wm * 64 + mt * 16 + l15;
    if (m >= g.M) continue;
    float rs = 1.0f;
    if (is_resid<EPI> && g.row_scale) rs = g.row_scale[m / g.rows_per_sample];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + wn * 64 + nt * 16 + lg * 4;
      if (n >= g.N) continue;
      float v[4] = {acc[nt][mt][0], acc[nt][mt][1], acc[nt][mt][2], acc[nt][mt][3]};
      if (res_early) {
        f32x4_t x = res[nt][mt];
        if (g.bias) {
          const f32x4_t b4 = bq[nt];
          v[0] += b4[0]; v[1] += b4[1]; v[2] += b4[2]; v[3] += b4[3];
        }
        if (g.drop_thresh) {
          const uint32_t i0 = (uint32_t)m * (uint32_t)g.N + (uint32_t)n;
          bool dk4[4];
          drop_keep4(i0, g.drop_key, g.drop_thresh, dk4);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = dk4[r] ? v[r] * g.drop_scale : 0.f;
        }
        x[0] += rs * v[0]; x[1] += rs * v[1]; x[2] += rs * v[2]; x[3] += rs * v[3];
        *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(g.C) + (size_t)m * g.ldc + n) = x;
      } else {
        epi_store<EPI>(g, m, n, v, rs, atomic_f32);
      }
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(256, 3) void gemm_nt_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) bf16_t smem[NS * STAGE];
  const int ntn = (g.N + BN - 1) / BN;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  // split-K (EPI_F32 only): blockIdx.y owns k-tiles [kt0, kt1); partial sums are combined with fp32 atomics
  const int kt0 = (int)blockIdx.y * g.ksplit_tiles, kt1 = min(g.K / BK, kt0 + g.ksplit_tiles);
  GDBG_T(0);
  gemm_tile_body<EPI>(g, smem, (wg / ntn) * BM, (wg % ntn) * BN, kt0, kt1, gridDim.y > 1);
  __syncthreads();
  GDBG_T(2);
}

// Small-problem kernel: 64x64 output tile, 8-stage LDS-DMA ring (7 K-tiles = 56 KiB in flight per workgroup).
// The backward of the 16 gradient-carrying images is made of M = 4112-row products: on 128x128 tiles they give 99-400
// workgroups whose K loop is a chain of ~2 us global->LDS round trips (3 in flight): 15-22 us for 3-7 GFLOP.  Four times as
// many workgroups with more than twice the bytes in flight each turn that into one latency + a short MFMA tail.
constexpr int SBM = 64, SNS = 5, SPD = SNS - 1, STILE = SBM * BK, SSTAGE = 2 * STILE;

template <int EPI>
__global__ __launch_bounds__(256, 4) void gemm_small_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) bf16_t smem[SNS * SSTAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lg = lane >> 4;
  const int ntn = (g.N + SBM - 1) / SBM;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (wg / ntn) * SBM, n0 = (wg % ntn) * SBM;
  const int nk = g.K / BK;

  // one A and one B LDS-DMA instruction (16 rows x 64 B) per wave and stage
  const int r = 16 * wave + (lane >> 2);
  const int sl = ((lane & 3) ^ swz(r)) * 8;
  const bf16_t* ga = g.A + (size_t)min(m0 + r, g.M - 1) * g.lda + sl;
  const bf16_t* gb = g.B + (size_t)min(n0 + r, g.N - 1) * g.ldb + sl;
  bf16_t* dst = smem + 16 * wave * BK;
#define SISSUE(kt_, st_)                                                                                               \
  {                                                                                                                    \
    __builtin_amdgcn_global_load_lds((gbl_void*)(ga + (size_t)(kt_) * BK), (lds_void*)(dst + (st_) * SSTAGE), 16, 0, 0);          \
    __builtin_amdgcn_global_load_lds((gbl_void*)(gb + (size_t)(kt_) * BK), (lds_void*)(dst + (st_) * SSTAGE + STILE), 16, 0, 0);  \
  }
  int fo_a[2], fo_b[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int rn = wn * 32 + t * 16 + l15, rm = wm * 32 + t * 16 + l15;
    fo_a[t] = STILE + rn * BK + ((lg ^ swz(rn)) << 3);
    fo_b[t] = rm * BK + ((lg ^ swz(rm)) << 3);
  }
  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < SPD; ++p)
    if (p < nk) SISSUE(p, p)
  for (int kt = 0; kt < nk; ++kt) {
    const int rem = nk - 1 - kt;          // tile kt has landed once <= 2*min(rem, SPD-1) younger DMA ops are outstanding
    switch (min(rem, SPD - 1)) {
      case 6: WAIT_VM(12); break;
      case 5: WAIT_VM(10); break;
      case 4: WAIT_VM(8); break;
      case 3: WAIT_VM(6); break;
      case 2: WAIT_VM(4); break;
      case 1: WAIT_VM(2); break;
      default: WAIT_VM(0); break;
    }
    __builtin_amdgcn_s_barrier();
    if (kt + SPD < nk) SISSUE(kt + SPD, (kt + SPD) % SNS)
    const bf16_t* st = smem + (kt % SNS) * SSTAGE;
    s16x8_t fa[2], fb[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      fa[t] = *reinterpret_cast<const s16x8_t*>(st + fo_a[t]);
      fb[t] = *reinterpret_cast<const s16x8_t*>(st + fo_b[t]);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
            __builtin_bit_cast(bf16x8_t, fa[nt]), __builtin_bit_cast(bf16x8_t, fb[mt]), acc[nt][mt], 0, 0, 0);
  }
#undef SISSUE
  if ((EPI == SRHIP_EPI_BF16 || EPI == SRHIP_EPI_GELU_BF16 || EPI == SRHIP_EPI_DGELU_BF16) && g.wide_store) {
    bf16_t* Cb = reinterpret_cast<bf16_t*>(g.C);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int mr = m0 + wm * 32 + mt * 16 + l15, m = min(mr, g.M - 1);
      u32x2_t q[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float v[4] = {acc[e][mt][0], acc[e][mt][1], acc[e][mt][2], acc[e][mt][3]};
        q[e] = epi_quad_bf16<EPI>(g, m, n0 + wn * 32 + e * 16 + lg * 4, v, mr < g.M);
      }
      store_quad_pair(Cb, g.ldc, m, mr < g.M, n0 + wn * 32, lg, q[0], q[1]);
    }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int m = m0 + wm * 32 + mt * 16 + l15;
    if (m >= g.M) continue;
    float rs = 1.0f;
    if (is_resid<EPI> && g.row_scale) rs = g.row_scale[m / g.rows_per_sample];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int n = n0 + wn * 32 + nt * 16 + lg * 4;
      if (n >= g.N) continue;
      float v[4] = {acc[nt][mt][0], acc[nt][mt][1], acc[nt][mt][2], acc[nt][mt][3]};
      epi_store<EPI>(g, m, n, v, rs, false);
    }
  }
}

// Grouped fp32-accumulating launch: ONE grid over the 128x128 tiles of up to 64 independent products.  Used for the
// weight gradients dW_l = dY_l^T X_l of all transformer blocks (48 products of 9..36 tiles each for ViT-S): launched one
// by one they fill 4-14 % of the chip (63 us each, 3 ms per step); grouped they are one 1296-tile launch.
__global__ __launch_bounds__(256, 2) void gemm_grouped_f32_kernel(const srhip_group_desc* __restrict__ desc, int n_problems,
                                                                  float alpha, float beta) {
  __shared__ __attribute__((aligned(16))) bf16_t smem[NS * STAGE];
  const int tile = xcd_remap(blockIdx.x, gridDim.x);      // tiles of one product (shared operands) stay on one XCD's L2
  int p = 0;
  {  // binary search: the last entry whose tile_start <= tile (a token-sliced table has hundreds of entries; a linear walk of dependent scalar
     // loads cost a workgroup as much as its product)
    int lo = 0, hi = n_problems - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tile >= desc[mid].tile_start) lo = mid; else hi = mid - 1; }
    p = lo;
  }
  const srhip_group_desc d = desc[p];
  GemmArgs g;
  g.A = (const bf16_t*)d.A; g.B = (const bf16_t*)d.B; g.C = d.C; g.bias = nullptr; g.row_scale = nullptr;
  g.aux_in = nullptr; g.aux_out = nullptr;
  g.M = d.M; g.N = d.N; g.K = d.K; g.lda = d.lda; g.ldb = d.ldb; g.ldc = d.ldc; g.ldaux = 0; g.rows_per_sample = 1;
  g.ksplit_tiles = d.K / BK; g.debug = 0; g.alpha = alpha; g.beta = beta; g.drop_key = 0u; g.drop_thresh = 0u; g.drop_scale = 1.0f;
  g.ln_mean = g.ln_rstd = g.ln_gamma = g.ln_beta = nullptr;
  const int local = tile - d.tile_start, ntn = (d.N + BN - 1) / BN;
  gemm_tile_body<SRHIP_EPI_F32>(g, smem, (local / ntn) * BM, (local % ntn) * BN, 0, d.K / BK, false);
}

// The same grouped launch for products with N <= 64 per column tile (the grouped positional convolution of Wav2Vec2 / HuBERT: 16 groups of 48
// output channels over K = 128 taps x 48 channels = 6144; HF Wav2Vec2PositionalConvEmbedding behind wave2vecv2.py:44).  On the 128 x 128 tile
// 62 % of the B tile that is filled and of the MFMAs that are issued belong to columns that do not exist (118 GFLOP per launch ran at the MFMA
// rate of 1000 TF/s for 375 TF/s of product).  Here the tile is 128 x 64: four waves along m with 32 rows x 64 columns each (8 MFMAs per
// k-step instead of 16), a 12-KiB stage (A 128 x 32, B 64 x 32), three LDS-DMA instructions per wave and stage, four workgroups per CU.
constexpr int NB64 = 64, TILE_B64 = NB64 * BK, STAGE64 = TILE + TILE_B64;
// ... and when the A operand is a SLIDING WINDOW (lda < K: row m of the operand starts lda elements behind row m - 1 -- the unfolded input of a
// convolution read in place), the 128 rows x K elements of a tile are (127 lda + K) distinct elements: 24 KiB for the positional convolution
// against 1.5 MiB streamed k-step by k-step.  The window is staged ONCE per tile and the A fragments are read from it at their sliding offsets
// (16-byte aligned: lda % 8 == 0); only the weights stream (4 KiB per k-step through a 4-stage ring).
constexpr int WIN_EL = 12 * 1024;         // window capacity in elements (24 KiB: 127 x 48 + 6144 = 12 240 for the positional convolution)
constexpr int NSW = 4, PDW = NSW - 1;     // ring of B tiles in the window mode: 24 + 16 KiB per workgroup, four workgroups per CU
__global__ __launch_bounds__(256, 4) void gemm_grouped_n64_f32_kernel(const srhip_group_desc* __restrict__ desc, int n_problems, float alpha, float beta) {
  __shared__ __attribute__((aligned(16))) bf16_t smem[NS * STAGE64 > WIN_EL + NSW * TILE_B64 ? NS * STAGE64 : WIN_EL + NSW * TILE_B64];
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  int p = 0;
  {
    int lo = 0, hi = n_problems - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tile >= desc[mid].tile_start) lo = mid; else hi = mid - 1; }
    p = lo;
  }
  const srhip_group_desc d = desc[p];
  const bf16_t* A = (const bf16_t*)d.A;
  const bf16_t* B = (const bf16_t*)d.B;
  const int local = tile - d.tile_start, ntn = (d.N + NB64 - 1) / NB64;
  const int m0 = (local / ntn) * BM, n0 = (local % ntn) * NB64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int nk = d.K / BK;
  if (nk <= 0) return;
  f32x4_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  size_t gob;
  {
    const int r = 16 * wave + (lane >> 2);
    gob = (size_t)min(n0 + r, d.N - 1) * d.ldb + (((lane & 3) ^ swz(r)) << 3);
  }
  int fo_a[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { const int rn = t * 16 + l15; fo_a[t] = rn * BK + ((lg ^ swz(rn)) << 3); }      // MFMA a-operand <- B matrix rows (n), inside a B tile
  const int span = (BM - 1) * d.lda + d.K;                       // distinct elements of the tile's 128 operand rows
  const bool window = d.lda < d.K && span <= WIN_EL && (d.lda & 7) == 0;        // (uniform)
  if (window) {
    // ---- the window: elements [m0 lda, m0 lda + span) of A, clipped to the operand's extent (rows >= M of a last tile are never stored)
    bf16_t* win = smem;
    bf16_t* ring = smem + WIN_EL;
    const long extent = (long)(d.M - 1) * d.lda + d.K;
    const long w0 = (long)m0 * d.lda;
    for (int c = tid * 8; c < span; c += 256 * 8) {              // 16 bytes per lane, lane-linear in LDS
      if (w0 + c + 8 <= extent) __builtin_amdgcn_global_load_lds((gbl_void*)(A + w0 + c), (lds_void*)(win + (c & ~511) + 0), 16, 0, 0);
    }
    // (global_load_lds writes lane-linear from the wave's base address: the base above is the wave's chunk start -- c & ~511 is uniform per wave
    // because a wave covers 64 x 8 = 512 consecutive elements)
#define ISSUEW(kt_, st_) __builtin_amdgcn_global_load_lds((gbl_void*)(B + gob + (size_t)(kt_) * BK), (lds_void*)(ring + (st_) * TILE_B64 + 16 * wave * BK), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < PDW; ++q)
      if (q < nk) { ISSUEW(q, q) }
    const int fw0 = (wave * 32 + l15) * d.lda + 8 * lg, fw1 = fw0 + 16 * d.lda;
    for (int kt = 0; kt < nk; ++kt) {
      const int rem = nk - 1 - kt;                               // one LDS-DMA instruction per wave and stage: up to PDW - 1 younger stages in flight
      if (rem >= 2) WAIT_VM(2); else if (rem == 1) WAIT_VM(1); else WAIT_VM(0);
      __builtin_amdgcn_s_barrier();
      if (kt + PDW < nk) { ISSUEW(kt + PDW, (kt + PDW) % NSW) }
      const bf16_t* st = ring + (kt % NSW) * TILE_B64;
      s16x8_t fa[4], fb[2];
#pragma unroll
      for (int t = 0; t < 4; ++t) fa[t] = *reinterpret_cast<const s16x8_t*>(st + fo_a[t]);
      fb[0] = *reinterpret_cast<const s16x8_t*>(win + fw0 + kt * BK);
      fb[1] = *reinterpret_cast<const s16x8_t*>(win + fw1 + kt * BK);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[nt]), __builtin_bit_cast(bf16x8_t, fb[mt]), acc[nt][mt], 0, 0, 0);
    }
#undef ISSUEW
    static_assert(PDW == 3, "the counted waits above are written for three stages in flight");
  } else {
    size_t goa[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 32 * wave + 16 * i + (lane >> 2);
      goa[i] = (size_t)min(m0 + r, d.M - 1) * d.lda + (((lane & 3) ^ swz(r)) << 3);
    }
#define ISSUE64(kt_, st_)                                                                                                          \
  {                                                                                                                                 \
    bf16_t* da = smem + (st_) * STAGE64 + 32 * wave * BK;                                                                           \
    __builtin_amdgcn_global_load_lds((gbl_void*)(A + goa[0] + (size_t)(kt_) * BK), (lds_void*)da, 16, 0, 0);                       \
    __builtin_amdgcn_global_load_lds((gbl_void*)(A + goa[1] + (size_t)(kt_) * BK), (lds_void*)(da + 16 * BK), 16, 0, 0);           \
    __builtin_amdgcn_global_load_lds((gbl_void*)(B + gob + (size_t)(kt_) * BK), (lds_void*)(smem + (st_) * STAGE64 + TILE + 16 * wave * BK), 16, 0, 0); \
  }
    int fo_b[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { const int rm = wave * 32 + t * 16 + l15; fo_b[t] = rm * BK + ((lg ^ swz(rm)) << 3); }  // MFMA b-operand <- A matrix rows (m)
#pragma unroll
    for (int q = 0; q < PD; ++q)
      if (q < nk) ISSUE64(q, q)
    for (int kt = 0; kt < nk; ++kt) {
      if (nk - 1 - kt >= 1) WAIT_VM(3); else WAIT_VM(0);       // three LDS-DMA instructions per wave and stage; PD - 1 = 1 younger stage may be in flight
      __builtin_amdgcn_s_barrier();
      if (kt + PD < nk) ISSUE64(kt + PD, (kt + PD) % NS)
      const bf16_t* st = smem + (kt % NS) * STAGE64;
      s16x8_t fa[4], fb[2];
#pragma unroll
      for (int t = 0; t < 4; ++t) fa[t] = *reinterpret_cast<const s16x8_t*>(st + TILE + fo_a[t]);
#pragma unroll
      for (int t = 0; t < 2; ++t) fb[t] = *reinterpret_cast<const s16x8_t*>(st + fo_b[t]);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[nt]), __builtin_bit_cast(bf16x8_t, fb[mt]), acc[nt][mt], 0, 0, 0);
    }
#undef ISSUE64
    static_assert(PD == 2 && NS == 3, "the counted wait above is written for two stages in flight");
  }
  // ---- epilogue: lane holds C[m][n .. n + 3], m = row l15 of row tile mt, n = 4 lg of column tile nt
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int m = m0 + wave * 32 + mt * 16 + l15;
    if (m >= d.M) continue;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + nt * 16 + lg * 4;
      if (n >= d.N) continue;
      f32x4_t* cp = reinterpret_cast<f32x4_t*>(d.C + (size_t)m * d.ldc + n);
      f32x4_t x = {alpha * acc[nt][mt][0], alpha * acc[nt][mt][1], alpha * acc[nt][mt][2], alpha * acc[nt][mt][3]};
      if (beta != 0.0f) {
        const f32x4_t c = *cp;
        x[0] += beta * c[0]; x[1] += beta * c[1]; x[2] += beta * c[2]; x[3] += beta * c[3];
      }
      *cp = x;
    }
  }
}

// =================================================================================================
// Large-problem kernel: 256 x (128|256) CU-level tile, 8 waves, persistent over tiles.
//
// Measured on MI355X (profiles/r01_c_*): a 16 KiB K-step of the 128x128 kernel costs ~0.9 us whatever the staging
// method, because global->LDS latency is ~2 us under load and a CU can only keep (LDS ring) bytes in flight: ~36 GB/s
// per CU.  L2 hit rate is 90 % and HBM fetch is the A panel once -- it is a latency x capacity limit, not bandwidth.
// Throughput is therefore (flop per in-flight byte) x (ring bytes / latency): this kernel doubles the first factor
// (256x128 -> 85 flop/B, 256x256 -> 128 flop/B vs 64) and gives the ring 120-128 KiB instead of 2 x 48 KiB, and it
// never drains the pipeline between tiles: a workgroup walks tiles  blockIdx.x, +gridDim.x, ...  with the LDS-DMA
// prefetch cursor running PDG steps ahead of the MFMA cursor across tile boundaries.
// Waves: 4 (m) x 2 (n); wave tile 64 x (16*NTW); NTW = 4 -> BN = 128, NTW = 8 -> BN = 256.
constexpr int GBM = 256;
// LDS-DMA ring depth of the persistent kernel (stages of (256 + BN) x 32 bf16).  256 x 256: 4 stages = 128 KB.  A fifth stage (the whole 160 KB of
// a CU's LDS, a third more bytes in flight) was measured neutral on every shape of the legs (profiles/r05_gemm_ring_depth.txt: 13952 x 3072 x 768
// 748 vs 749 TF/s, 8192^3 1136 vs 1152): the K loop is not waiting for operand arrival at this depth.  Nor for the refill's issue: the two waves
// of a SIMD taking refill and MFMAs in opposite order (one blocked in its LDS-DMA instructions while the other multiplies) was neutral too
// (profiles/r05_gemm_stagger.txt).
#ifndef SRHIP_BIG_NSTG
#define SRHIP_BIG_NSTG 4
#endif
constexpr int big_stages(int ntw, int wn) { return wn == 1 ? 3 : (ntw == 4 ? 5 : SRHIP_BIG_NSTG); }

// tile index -> (row tile, column tile) in bands of GROUP_M row tiles walked column-major: the 32 consecutive indices an XCD works on at a
// time (xcd_remap) are then 8 x 4 tiles -- 8 A panels + 4 B panels per k step from HBM / MALL instead of 1 + 32 for a row of tiles
constexpr int GROUP_M = 8;
__device__ __forceinline__ void tile_mn(int t, int ntm, int ntn, int& tm, int& tn) {
  const int band = t / (GROUP_M * ntn), r = t - band * GROUP_M * ntn;
  const int rows = min(GROUP_M, ntm - band * GROUP_M);
  tm = band * GROUP_M + r % rows;
  tn = r / rows;
}

// WN = waves along n (2: 8 waves, one workgroup per CU; 1: 4 waves with wave tile 64 x (16*NTW), TWO workgroups per CU whose
// K-loop and epilogue phases drift apart and overlap).
template <int EPI, int NTW, int WN>
__global__ __launch_bounds__(256 * WN, 2) void gemm_big_kernel(GemmArgs g) {
  const bool g_grouped = (g.debug & 64) == 0;          // SRHIP_DEBUG=64: row-major tile walk (tuning)
  constexpr int NWV = 4 * WN;                        // waves per workgroup (4 along m)
  constexpr int GBN = 16 * NTW * WN;
  constexpr int A_EL = GBM * BK, B_EL = GBN * BK, STG = A_EL + B_EL;
  constexpr int NSTG = big_stages(NTW, WN), PDG = NSTG - 1;
  constexpr int AI = 16 / NWV, BI = (GBN / 16) / NWV; // LDS-DMA instructions (16 rows each) per wave for the A / B sub-tile
  constexpr int NI = AI + BI;
  extern __shared__ __attribute__((aligned(16))) bf16_t gsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l15 = lane & 15, lg = lane >> 4;
  const int ntn = (g.N + GBN - 1) / GBN, ntm = (g.M + GBM - 1) / GBM, ntiles = ntm * ntn;
  const int nk = g.K / BK;
  // XCD-aware start tile: workgroups b, b+8, b+16.. share an XCD (private L2), so each XCD walks a CONTIGUOUS chunk of the
  // tile list (tiles of one A row-panel are neighbours).  Without it the 5-6 N-tiles of a panel ran on different XCDs and
  // the A operand was fetched 6x from HBM/MALL (PMC: 249 MB per qkv launch for a 39.5 MB operand).
  const int first = xcd_remap(blockIdx.x, gridDim.x);
  if (first >= ntiles) return;
  const int my_tiles = (ntiles - first + gridDim.x - 1) / gridDim.x;
  const int steps = my_tiles * nk;

  // ---- producer cursor (LDS-DMA); lane -> (row, physical slot) as in the 128x128 kernel
  int pt = first, pk = 0;
  int ra[AI], rb[BI], sa[AI], sb[BI];
#pragma unroll
  for (int i = 0; i < AI; ++i) { ra[i] = (wave * AI + i) * 16 + (lane >> 2); sa[i] = ((lane & 3) ^ swz(ra[i])) * 8; }
#pragma unroll
  for (int i = 0; i < BI; ++i) { rb[i] = (wave * BI + i) * 16 + (lane >> 2); sb[i] = ((lane & 3) ^ swz(rb[i])) * 8; }
  const bf16_t* pa[AI];
  const bf16_t* pb[BI];
  auto set_tile = [&](int t) {
    int tm_, tn_;
    if (g_grouped) tile_mn(t, ntm, ntn, tm_, tn_); else { tm_ = t / ntn; tn_ = t % ntn; }
    const int m0 = tm_ * GBM, n0 = tn_ * GBN;
#pragma unroll
    for (int i = 0; i < AI; ++i) pa[i] = g.A + (size_t)min(m0 + ra[i], g.M - 1) * g.lda + sa[i];
#pragma unroll
    for (int i = 0; i < BI; ++i) pb[i] = g.B + (size_t)min(n0 + rb[i], g.N - 1) * g.ldb + sb[i];
  };
  set_tile(pt);
  auto issue = [&](int stage) {
    bf16_t* da = gsm + stage * STG + (wave * AI * 16) * BK;
    bf16_t* db = gsm + stage * STG + A_EL + (wave * BI * 16) * BK;
    const int ko = pk * BK;
#pragma unroll
    for (int i = 0; i < AI; ++i) __builtin_amdgcn_global_load_lds((gbl_void*)(pa[i] + ko), (lds_void*)(da + i * 16 * BK), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < BI; ++i) __builtin_amdgcn_global_load_lds((gbl_void*)(pb[i] + ko), (lds_void*)(db + i * 16 * BK), 16, 0, 0);
    if (++pk == nk) { pk = 0; pt += gridDim.x; if (pt < ntiles) set_tile(pt); }
  };
#pragma unroll
  for (int p = 0; p < PDG; ++p)
    if (p < steps) issue(p);

  int fo_a[NTW], fo_b[4];
#pragma unroll
  for (int t = 0; t < NTW; ++t) { const int rn = wn * (16 * NTW) + t * 16 + l15; fo_a[t] = A_EL + rn * BK + ((lg ^ swz(rn)) << 3); }
#pragma unroll
  for (int t = 0; t < 4; ++t) { const int rm = wm * 64 + t * 16 + l15; fo_b[t] = rm * BK + ((lg ^ swz(rm)) << 3); }
  f32x4_t acc[NTW][4];
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  int ct = first, ck = 0;
  for (int s = 0; s < steps; ++s) {
    const int rem = steps - 1 - s;        // step s landed once <= min(rem, PDG-1) younger stages (NI ops each) are in flight
    if (rem >= PDG - 1) wait_vm<(PDG - 1) * NI>();
    else if (rem == 2) wait_vm<2 * NI>();
    else if (rem == 1) wait_vm<NI>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (s + PDG < steps) issue((s + PDG) % NSTG);
    const bf16_t* st = gsm + (s % NSTG) * STG;
    s16x8_t fa[NTW], fb[4];
#pragma unroll
    for (int t = 0; t < NTW; ++t) fa[t] = *reinterpret_cast<const s16x8_t*>(st + fo_a[t]);
#pragma unroll
    for (int t = 0; t < 4; ++t) fb[t] = *reinterpret_cast<const s16x8_t*>(st + fo_b[t]);
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[i]), __builtin_bit_cast(bf16x8_t, fb[j]),
                                                            acc[i][j], 0, 0, 0);
    if (++ck == nk) {
      int tm_, tn_;
      if (g_grouped) tile_mn(ct, ntm, ntn, tm_, tn_); else { tm_ = ct / ntn; tn_ = ct % ntn; }
      const int m0 = tm_ * GBM, n0 = tn_ * GBN;
      if ((EPI == SRHIP_EPI_BF16 || EPI == SRHIP_EPI_GELU_BF16 || EPI == SRHIP_EPI_DGELU_BF16) && g.wide_store &&
          n0 + GBN <= g.N) {                      // (wave-uniform: full column tiles only; the ragged last one takes the plain path)
        bf16_t* Cb = reinterpret_cast<bf16_t*>(g.C);
        f32x4_t bq[NTW];                             // bias quads: once per column tile, not once per (row tile, column tile)
        if (g.bias) {
#pragma unroll
          for (int i = 0; i < NTW; ++i) bq[i] = *reinterpret_cast<const f32x4_t*>(g.bias + n0 + wn * (16 * NTW) + i * 16 + lg * 4);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int mr = m0 + wm * 64 + mt * 16 + l15, m = min(mr, g.M - 1);
#pragma unroll
          for (int np = 0; np < NTW / 2; ++np) {
            u32x2_t q[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int i = 2 * np + e;
              float v[4] = {acc[i][mt][0], acc[i][mt][1], acc[i][mt][2], acc[i][mt][3]};
              acc[i][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
              q[e] = epi_quad_bf16<EPI>(g, m, n0 + wn * (16 * NTW) + i * 16 + lg * 4, v, mr < g.M, &bq[i]);
            }
            store_quad_pair(Cb, g.ldc, m, mr < g.M, n0 + wn * (16 * NTW) + np * 32, lg, q[0], q[1]);
          }
        }
      } else {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wm * 64 + mt * 16 + l15;
        float rs = 1.0f;
        if (is_resid<EPI> && g.row_scale && m < g.M) rs = g.row_scale[m / g.rows_per_sample];
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
          const int n = n0 + wn * (16 * NTW) + i * 16 + lg * 4;
          float v[4] = {acc[i][mt][0], acc[i][mt][1], acc[i][mt][2], acc[i][mt][3]};
          acc[i][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
          if (m < g.M && n < g.N) epi_store<EPI>(g, m, n, v, rs, false);
        }
      }
      }
      ck = 0;
      ct += gridDim.x;
    }
  }
}

template <int EPI, int NTW, int WN>
static void launch_big_t(const GemmArgs& g, int grid_cap, hipStream_t s) {
  constexpr int GBN = 16 * NTW * WN;
  constexpr int NSTG = big_stages(NTW, WN);
  constexpr size_t sm = (size_t)NSTG * (GBM + GBN) * BK * sizeof(bf16_t);
  const int tiles = cdiv(g.M, GBM) * cdiv(g.N, GBN);
  auto kern = gemm_big_kernel<EPI, NTW, WN>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  SR_LAUNCH(kern, dim3(min(tiles, grid_cap)), dim3(256 * WN), sm, s, g);
}

// =================================================================================================
// 256 x 256 x 64 kernel, two wave groups half a phase apart (K % 64 == 0; operands < 2 GiB so that buffer addressing reaches them).
//
// The persistent kernel above runs all eight waves in lockstep: barrier, refill, 12 fragment reads, 32 MFMAs -- both waves of a SIMD read
// together and multiply together, and the matrix pipe idles through every read phase (1070-1130 TF/s at 8192^3 against the vendor library's
// 1510-1600).  Here the waves are 2 (m) x 4 (n) with a 128 x 64 wave tile, and the row groups wr = 0 / 1 -- one wave of each on every SIMD --
// alternate: between two consecutive workgroup barriers one group issues its fragment reads and LDS-DMA refills while the other runs 16 MFMAs
// (one 64 x 32 quadrant of its tile over a 64-deep K-tile), then they swap.  A K-tile is four such phases per group:
//     p0: read A0 (8 fragments) + B0 (4)   q(0,0)        p1: read B1 (4)   q(0,1)        p2: read A1 (8)   q(1,1)        p3: --   q(1,0) (B0 kept)
// LDS: a ring of NSLOT 16-KiB half-tiles [128 rows][64 k] (A0 / A1 = the rows every wave multiplies in its upper / lower quadrants, B0 / B1
// likewise for columns), refilled one per phase in the fixed order  q = 4 s + {A0, B0, B1, A1}  of K-tile s, slot q % NSLOT.  Phase k = 4 s + p
// issues q = k + NSLOT - 2: its slot held q - NSLOT = k - 2, last read in phase k - 2 or k - 3 -- two phases back at least, the write-after-read
// distance two staggered groups need -- and then waits until at most NSLOT - 4 half-tiles are in flight, i.e. q <= k + 2 has landed: what phase
// k + 1 reads, one barrier later.  The walk over output tiles is persistent: the refill cursor runs across tile boundaries, so the first
// K-tiles of the next output tile land during the epilogue.
// Stores count in vmcnt like loads, so a counted wait behind an epilogue would drain the stores: every output tile ends with vmcnt(0) BEFORE its
// stores (the half-tiles ahead have had a phase or more to land) and the first NSLOT - 4 phases of the next tile wait for nothing.
// Both groups take their epilogue between the same two barriers (the stagger is closed at the end of a tile and reopened at the start of the
// next: one extra barrier each), otherwise each group's stores would hold the other one at a barrier in turn.
// (A stream-K walk of this kernel -- equal K-tile ranges per workgroup, fp32 slabs + flags for the shared tiles -- was built and measured: the
// 256-KiB slab round trip costs a workgroup 20-30 us, more than the idle CUs of a partial last round; profiles/r05_gemm_stream_k.txt.)
// 16-byte chunks of a 128-byte LDS row are permuted by (row >> 1) & 7 (on the source address, LDS-DMA writes lane-linear): the 16 lanes of a
// ds_read_b128 service group (8 rows at chunk c, 8 at c ^ 1) then fall on 16 distinct 16-byte bank groups.
constexpr int PBK = 64;
constexpr int PH_EL = 128 * PBK;          // one half-tile

template <int EPI, int NSLOT>
__global__ __launch_bounds__(512, 1) void gemm_pp_kernel(GemmArgs g) {
  constexpr int LEAD = NSLOT - 2;         // phase k issues half-tile k + LEAD
  constexpr int INFL = 2 * (LEAD - 2);    // LDS-DMA instructions that may stay in flight behind a phase's wait
  static_assert(NSLOT == 8 || NSLOT == 10, "ring");
  extern __shared__ __attribute__((aligned(16))) bf16_t psm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l15 = lane & 15, lg = lane >> 4;
  const int ntn = (g.N + 255) / 256, ntm = (g.M + 255) / 256, ntiles = ntm * ntn;
  const int nk = g.K / PBK;
  const int first = xcd_remap(blockIdx.x, gridDim.x);
  if (first >= ntiles) return;
  const int my_tiles = (ntiles - first + gridDim.x - 1) / gridDim.x;

  // ---- producer: a wave-instruction moves one 1-KiB piece = 8 rows x 128 B; every wave owns pieces wave and wave + 8 of each half-tile
  const int prho = wave * 8 + (lane >> 3);                      // row of the lane inside the half-tile (piece 1: + 64)
  const int pch = ((lane & 7) ^ ((prho >> 1) & 7)) * 8;         // logical chunk the lane's physical slot holds
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(g.A), 0, (int)(((size_t)(g.M - 1) * g.lda + g.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(g.B), 0, (int)(((size_t)(g.N - 1) * g.ldb + g.K) * 2), 0x00020000);
  constexpr int OOB = 0x7ffffff0;                               // past num_records: the load moves nothing and still counts in vmcnt
  // refill cursor: K-tile (c_tile, c_kt) whose half-tiles are being issued, byte offsets of the lane's two pieces per kind, next ring slot
  int c_tile = first, c_kt = 0, c_slot = 0;
  int va[2][2], vb[2][2];                                       // [half][piece]
  auto set_cur = [&]() {
    if (c_tile >= ntiles) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { va[i >> 1][i & 1] = OOB; vb[i >> 1][i & 1] = OOB; }
      return;
    }
    int tm_, tn_;
    tile_mn(c_tile, ntm, ntn, tm_, tn_);
    const int m0 = tm_ * 256, n0 = tn_ * 256;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        va[h][i] = (min(m0 + i * 128 + h * 64 + prho, g.M - 1) * g.lda + pch) * 2;
        vb[h][i] = (min(n0 + ((prho >> 5) + 2 * i) * 64 + h * 32 + (prho & 31), g.N - 1) * g.ldb + pch) * 2;
      }
  };
  auto issue = [&](auto kc) __attribute__((always_inline)) {      // kind: 0 = A0, 1 = B0, 2 = B1, 3 = A1
    constexpr int kind = decltype(kc)::value;
    bf16_t* dst = psm + c_slot * PH_EL + wave * 512;
    const int ko = c_kt * (PBK * 2);
    const int (&vo)[2] = kind == 0 ? va[0] : (kind == 3 ? va[1] : (kind == 1 ? vb[0] : vb[1]));
    const __amdgpu_buffer_rsrc_t& rs = (kind == 0 || kind == 3) ? rsa : rsb;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, vo[0], ko, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + 8 * 512), 16, vo[1], ko, 0, 0);
    c_slot = c_slot + 1 == NSLOT ? 0 : c_slot + 1;
    if constexpr (kind == 3) {
      if (++c_kt == nk) { c_kt = 0; c_tile += gridDim.x; set_cur(); }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  set_cur();
  issue(I0{}); issue(I1{}); issue(I2{}); issue(I3{});
  issue(I0{}); issue(I1{});
  if constexpr (LEAD == 8) { issue(I2{}); issue(I3{}); }

  // ---- consumer fragment offsets (elements) inside a half-tile: row l15 of a 16-row tile, k-step ks
  int foA[2], foB[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int ch = ((4 * ks + lg) ^ (l15 >> 1)) << 3;
    foA[ks] = (wr * 64 + l15) * PBK + ch;
    foB[ks] = (wc * 32 + l15) * PBK + ch;
  }
  f32x4_t acc[4][8];                     // [column tile nt][row tile mt] of the wave's 128 x 64
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  s16x8_t fa[4][2], fb0[2][2], fb1[2][2];

  auto rd = [&](const bf16_t* p) __attribute__((always_inline)) { return *reinterpret_cast<const s16x8_t*>(p); };
  // 16 MFMAs of quadrant (h, j): acc[2 j + nt][4 h + mt] += B fragment (nt, ks) x A fragment (mt, ks)
  auto quadrant = [&](auto hc, auto jc, s16x8_t (&fb)[2][2]) __attribute__((always_inline)) {
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
    __builtin_amdgcn_s_setprio(0);
  };
#define PP_SYNC(wait_)                                                   \
  if (wait_) wait_vm<INFL>();                                            \
  __builtin_amdgcn_s_barrier();                                          \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     \
  __builtin_amdgcn_sched_barrier(0);
#define PP_END()                                                         \
  __builtin_amdgcn_sched_barrier(0);                                     \
  __builtin_amdgcn_s_barrier();                                          \
  __builtin_amdgcn_sched_barrier(0);
  int r_base = 0;                        // ring slot of A0 of the K-tile being multiplied
  // kt = K-tile inside the output tile: phases 4 kt + p < LEAD - 2 follow the vmcnt(0) of the previous epilogue (or of the prologue)
  auto ktile = [&](int kt) __attribute__((always_inline)) {
    const bool w01 = 4 * kt >= LEAD - 2 - 1, w23 = 4 * kt + 2 >= LEAD - 2;          // LEAD 6: kt >= 1 both; LEAD 8: p0/p1 kt >= 2, p2/p3 kt >= 1
    const bool w0 = 4 * kt >= LEAD - 2, w1 = 4 * kt + 1 >= LEAD - 2, w2 = w23, w3 = 4 * kt + 3 >= LEAD - 2;
    (void)w01;
    int sl[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { const int v = r_base + c; sl[c] = (NSLOT == 8 || v < NSLOT ? v : v - NSLOT) * PH_EL; }
    // ---- p0
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fb0[nt][ks] = rd(psm + sl[1] + foB[ks] + nt * 16 * PBK);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fa[mt][ks] = rd(psm + sl[0] + foA[ks] + mt * 16 * PBK);
    issue(std::integral_constant<int, (LEAD + 0) & 3>{});
    PP_SYNC(w0)
    quadrant(I0{}, I0{}, fb0);
    PP_END()
    // ---- p1
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fb1[nt][ks] = rd(psm + sl[2] + foB[ks] + nt * 16 * PBK);
    issue(std::integral_constant<int, (LEAD + 1) & 3>{});
    PP_SYNC(w1)
    quadrant(I0{}, I1{}, fb1);
    PP_END()
    // ---- p2
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fa[mt][ks] = rd(psm + sl[3] + foA[ks] + mt * 16 * PBK);
    issue(std::integral_constant<int, (LEAD + 2) & 3>{});
    PP_SYNC(w2)
    quadrant(I1{}, I1{}, fb1);
    PP_END()
    // ---- p3
    issue(std::integral_constant<int, (LEAD + 3) & 3>{});
    PP_SYNC(w3)
    quadrant(I1{}, I0{}, fb0);
    PP_END()
    r_base = r_base + 4 >= NSLOT ? r_base + 4 - NSLOT : r_base + 4;
  };

  PDBG_T(12);
  WAIT_VM(0);
  __builtin_amdgcn_s_barrier();
  PDBG_T(13);
  int ct = first;
  constexpr bool BF16_OUT = EPI == SRHIP_EPI_BF16 || EPI == SRHIP_EPI_GELU_BF16 || EPI == SRHIP_EPI_DGELU_BF16;
  for (int t = 0; t < my_tiles; ++t, ct += gridDim.x) {
    PDBG_T(min(t, 1) * 6 + 0);
    if (wr == 1) __builtin_amdgcn_s_barrier();        // the lower row group runs one barrier behind
    for (int kt = 0; kt < nk; ++kt) ktile(kt);
    PDBG_T(min(t, 1) * 6 + 1);
    // ---- epilogue: lane holds C[m][n .. n + 3], m = row l15 of row tile mt, n = 4 lg of column tile nt
    int tm_, tn_;
    tile_mn(ct, ntm, ntn, tm_, tn_);
    const int m0 = tm_ * 256 + wr * 128, n0 = tn_ * 256 + wc * 64;
    // what the epilogue reads first is requested ahead of the closing barrier and the vmcnt(0): the fragment registers are free by now
    f32x4_t bq[4];
    if (g.bias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) bq[i] = *reinterpret_cast<const f32x4_t*>(g.bias + min(n0 + i * 16 + lg * 4, g.N - 4));
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    WAIT_VM(0);
    __builtin_amdgcn_sched_barrier(0);
    PDBG_T(min(t, 1) * 6 + 2);
    if (BF16_OUT && g.wide_store && tn_ * 256 + 256 <= g.N) {
      bf16_t* Cb = reinterpret_cast<bf16_t*>(g.C);
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        const int mr = m0 + mt * 16 + l15, m = min(mr, g.M - 1);
#pragma unroll
        for (int np = 0; np < 2; ++np) {
          u32x2_t q[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int i = 2 * np + e;
            float v[4] = {acc[i][mt][0], acc[i][mt][1], acc[i][mt][2], acc[i][mt][3]};
            acc[i][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            q[e] = epi_quad_bf16<EPI>(g, m, n0 + i * 16 + lg * 4, v, mr < g.M, &bq[i]);
          }
          store_quad_pair(Cb, g.ldc, m, mr < g.M, n0 + np * 32, lg, q[0], q[1]);
        }
      }
    } else if (is_resid<EPI>) {
      // x += rs * (acc + bias): the 16 residual quads of four row tiles are requested together (in place C is read and written through the same
      // pointer, so left to itself every load waits for the store before it) -- two exposed round trips per tile instead of 32
      const float* src = g.aux_in ? reinterpret_cast<const float*>(g.aux_in) : reinterpret_cast<const float*>(g.C);
      const int lds_ = g.aux_in ? g.ldaux : g.ldc;
      constexpr int RB = EPI == EPI_RESID_LN ? 2 : 4;       // row tiles per batch (the LayerNorm of the residual needs the registers of two)
#pragma unroll
      for (int hh = 0; hh < 8 / RB; ++hh) {
        f32x4_t res[RB][4];
#pragma unroll
        for (int mq = 0; mq < RB; ++mq) {
          const int m = min(m0 + (RB * hh + mq) * 16 + l15, g.M - 1);
#pragma unroll
          for (int i = 0; i < 4; ++i) res[mq][i] = *reinterpret_cast<const f32x4_t*>(src + (size_t)m * lds_ + min(n0 + i * 16 + lg * 4, g.N - 4));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mq = 0; mq < RB; ++mq) {
          const int mt = RB * hh + mq, m = m0 + mt * 16 + l15;
          float rs = 1.0f;
          if (g.row_scale && m < g.M) rs = g.row_scale[m / g.rows_per_sample];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int n = n0 + i * 16 + lg * 4;
            float v[4] = {acc[i][mt][0], acc[i][mt][1], acc[i][mt][2], acc[i][mt][3]};
            acc[i][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (g.bias) { v[0] += bq[i][0]; v[1] += bq[i][1]; v[2] += bq[i][2]; v[3] += bq[i][3]; }
            if (g.drop_thresh) {
              const uint32_t i0 = (uint32_t)m * (uint32_t)g.N + (uint32_t)n;
              bool dk4[4];
              drop_keep4(i0, g.drop_key, g.drop_thresh, dk4);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = dk4[r] ? v[r] * g.drop_scale : 0.f;
            }
            f32x4_t x = ln_resid<EPI>(g, min(m, g.M - 1), min(n, g.N - 4), res[mq][i]);
            x[0] += rs * v[0]; x[1] += rs * v[1]; x[2] += rs * v[2]; x[3] += rs * v[3];
            if (m < g.M && n < g.N) *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(g.C) + (size_t)m * g.ldc + n) = x;
          }
        }
      }
    } else {
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        const int m = m0 + mt * 16 + l15;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int n = n0 + i * 16 + lg * 4;
          float v[4] = {acc[i][mt][0], acc[i][mt][1], acc[i][mt][2], acc[i][mt][3]};
          acc[i][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
          if (m < g.M && n < g.N) epi_store<EPI>(g, m, n, v, 1.0f, false);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    PDBG_T(min(t, 1) * 6 + 3);
  }
#undef PP_SYNC
#undef PP_END
}

template <int EPI, int NSLOT>
static void launch_pp_t(const GemmArgs& g, hipStream_t s) {
  constexpr size_t sm = (size_t)NSLOT * PH_EL * sizeof(bf16_t);
  const int tiles = cdiv(g.M, 256) * cdiv(g.N, 256);
  auto kern = gemm_pp_kernel<EPI, NSLOT>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  SR_LAUNCH(kern, dim3(min(tiles, 256)), dim3(512), sm, s, g);
}
template <int EPI>
static void launch_pp(const GemmArgs& g, hipStream_t s) {
  static const char* mode = getenv("SRHIP_GEMM");
  if (mode && !strcmp(mode, "big256r8")) launch_pp_t<EPI, 8>(g, s); else launch_pp_t<EPI, 10>(g, s);
}

// variant: 0 = 256x256 / 8 waves / 1 WG per CU, 1 = 256x128 / 8 waves, 2 = 256x128 / 4 waves / 2 WGs per CU
template <int EPI>
static void launch_big(const GemmArgs& g, int variant, hipStream_t s) {
  if (variant == 0) launch_big_t<EPI, 8, 2>(g, 256, s);
  else if (variant == 1) launch_big_t<EPI, 4, 2>(g, 256, s);
  else launch_big_t<EPI, 8, 1>(g, 512, s);
}

}  // namespace

static int g_small_max_grid = 256;      // srhip_gemm_small_max_grid (one process per GPU, one launching thread: a plain int)

// Which kernel a product goes to (the decision of srhip_gemm_nt, also exported as srhip_gemm_nt_plan so that a test can pin it: a rule written for
// one family of shapes has caught another before -- DESIGN 6f).  Returns SRHIP_GEMM_PLAN_*; *splits = K splits of the 128 x 128 kernel.
static int gemm_plan(int epilogue, int M, int N, int K, float beta, int* splits_out) {
  const int grid = cdiv(M, BM) * cdiv(N, BN);
  // split-K: weight-gradient products (small M x N, long K = tokens) would otherwise fill a few dozen of the 256 CUs.
  // Only the accumulating fp32 epilogue (beta == 1) can be split; ~512 workgroups are targeted.
  int splits = 1;
  const int nkt = K / BK;
  if (epilogue == SRHIP_EPI_F32 && beta == 1.0f && grid < 128 && nkt >= 16) splits = min(min(cdiv(256, grid), nkt / 8), 32);
  splits = cdiv(nkt, cdiv(nkt, splits));
  if (splits_out) *splits_out = splits;
  // large problems go to the persistent 256-row kernel (tuning switches: SRHIP_GEMM=tile|big128|big256)
  static const char* mode = getenv("SRHIP_GEMM");
  const bool force_tile = mode && mode[0] == 't';
  // measured (tools/microbench.py, M = 51400): N >= 1024 -> 256x256 persistent kernel wins (fc1 152 -> 144 us, 8192^3
  // 775 -> 1064 TF); N = 384 products are epilogue/HBM bound and slightly better on the 128x128 kernel (2 WGs/CU).
  // The persistent kernel runs 256 workgroups over 256x256 tiles: below ~3 full rounds of tiles its round quantisation costs more than the
  // larger tile saves (the split launches of a training step: 32 639 x 1152 = 640 tiles = 2.5 -> 3 rounds; on the 128x128 kernel 2313 tiles
  // over 512 slots; measured 1029 -> 1065 img/s).  SRHIP_BIG_MIN_ROUNDS overrides the threshold for tuning.
  static const double big_min_rounds = SR_TUNE_ENV("SRHIP_BIG_MIN_ROUNDS") ? atof(SR_TUNE_ENV("SRHIP_BIG_MIN_ROUNDS")) : 3.0;
  // (N = 512 conv layers of the Wav2Vec2 feature encoder, K = 1024 / 1536 over 10^5..10^6 frames: +5 % clips/s on the persistent kernel)
  // K >= 768 (the D = 768 legs: BERT / Wav2Vec2 / HuBERT): the K loop is long enough that the 256-row tile pays from ~0.5 rounds of tiles on (12288 x 768 x 3072, 0.56 rounds: 98 -> 76 us, tools/gemm_dispatch_probe.py), also
  // at N = 768 (tools/gemm_modes_probe.py, standalone TF/s default -> this rule: BERT qkv 13952 x 2304 x 768 664 -> 800, fc1 615 -> 735, fc2 13952 x
  // 768 x 3072 630 -> 680, Wav2Vec2 fc1 5373 x 3072 x 768 617 -> 775); the 3-round threshold above is for the epilogue-heavy K = 384 products
  const bool big_k = K >= 768;
  const bool force_big = mode && mode[0] == 'b' && strcmp(mode, "bigold") != 0;        // "bigold": the lockstep kernel where the plan says 256 x 256, nothing forced
  const bool want_big = N >= 1024 || (N >= 512 && K >= 1024 && M >= 65536) || (big_k && N >= 768 && M >= 8192) || force_big;
  const double big_rounds = (double)cdiv(M, 256) * cdiv(N, 256) / 256.0;
  const double min_rounds = (big_k && !SR_TUNE_ENV("SRHIP_BIG_MIN_ROUNDS")) ? 0.5 : big_min_rounds;
  if (!force_tile && want_big && epilogue != SRHIP_EPI_F32 && M >= 4 * GBM && (big_rounds >= min_rounds || force_big)) {
    if (mode && !strcmp(mode, "big128")) return SRHIP_GEMM_PLAN_BIG128;
    if (mode && !strcmp(mode, "big2wg")) return SRHIP_GEMM_PLAN_BIG2WG;
    // the two-wave-group kernel whenever its 64-deep K-tiles and 31-bit buffer offsets fit (contiguous operands assumed here; the launch
    // re-checks with the real leading dimensions).  SRHIP_GEMM=bigold pins the lockstep kernel
    const bool pp = (K % 64) == 0 && (size_t)M * K * 2 < (1ull << 31) && (size_t)N * K * 2 < (1ull << 31) && !(mode && !strncmp(mode, "bigold", 6));
    return pp ? SRHIP_GEMM_PLAN_PP256 : SRHIP_GEMM_PLAN_BIG256;
  }
  // under-filled launches (less than one round of 2 workgroups per CU on 128x128 tiles) -> 64x64 tiles, deep ring
  // (the 64x64 kernel is for short K loops: at K >= 768 a launch of < 256 128x128 tiles is still faster on those tiles -- Wav2Vec2 fc2 5373 x 768 x
  // 3072: 405 -> 607 TF/s, BERT gradient-row fc2 4096 x 768 x 3072: 394 -> 470)
  // ... but only for the D = 768 widths: the ViT-S gradient-row products with a long K (fc2 4112 x 384 x 1536, the dX products with K = 1152 / 1536)
  // have N = 384 = 3 column tiles of 128 -- 99 workgroups -- and take 2.5 x as long there (12.8 -> 31 us, measured in the step)
  const bool force_small = mode && mode[0] == 's';
  // The threshold is a run-time setting (srhip_gemm_small_max_grid): 64 x 64 tiles are the LATENCY choice -- 390 instead of 99 workgroups for a
  // 4112 x 384 product, 12.8 instead of 31 us with the chip to itself (the K = 0 regime: 3.21 vs 3.57 ms per step) -- but while the row-streaming
  // launches of the deferred rows own most CUs, fewer and fatter workgroups win (K = 8 headline: 4.86 vs 4.95 ms, ViT-S/16@224 4.07 vs 4.13;
  // DESIGN 6f), so the step sets it per regime.  SRHIP_SMALL_MAX_GRID pins it for tuning.
  static const int small_env = SR_TUNE_ENV("SRHIP_SMALL_MAX_GRID") ? atoi(SR_TUNE_ENV("SRHIP_SMALL_MAX_GRID")) : -1;
  const int small_max_grid = small_env >= 0 ? small_env : g_small_max_grid;
  if (((grid < small_max_grid && !(big_k && N >= 768)) || force_small) && !force_tile && splits == 1 && epilogue != SRHIP_EPI_F32)
    return SRHIP_GEMM_PLAN_SMALL64;
  return SRHIP_GEMM_PLAN_TILE128;
}

extern "C" int srhip_gemm_small_max_grid(int n) {
  const int prev = g_small_max_grid;
  if (n >= 0) g_small_max_grid = n;
  return prev;
}

extern "C" int srhip_gemm_nt_plan(int epilogue, int M, int N, int K, float beta) {
  if (M <= 0 || N <= 0 || K <= 0 || (K % BK)) return SR_EINVAL;
  return gemm_plan(epilogue, M, N, K, beta, nullptr);
}

static int gemm_nt_impl(int epilogue, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                        int M, int N, int K, const float* bias, const float* row_scale, int rows_per_sample,
                        const void* aux_in, void* aux_out, int ldaux, float alpha, float beta, uint32_t drop_key, uint32_t drop_thresh,
                        float drop_scale, void* stream, const float* const* ln = nullptr) {
  if (M <= 0 || N <= 0 || K <= 0 || (K % BK) || (N % 4) || (lda % 8) || (ldb % 8) || (ldc % 4)) return SR_EINVAL;   // BK = 32
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return SR_EINVAL;
  if (epilogue == SRHIP_EPI_DGELU_BF16 && !aux_in) return SR_EINVAL;
  if (row_scale && rows_per_sample <= 0) return SR_EINVAL;
  GemmArgs g;
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = C; g.bias = bias; g.row_scale = row_scale;
  g.aux_in = (const bf16_t*)aux_in; g.aux_out = (bf16_t*)aux_out;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = ldaux;
  g.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1; g.alpha = alpha; g.beta = beta;
  g.drop_key = drop_key; g.drop_thresh = drop_thresh; g.drop_scale = drop_scale;
  g.ln_mean = ln ? ln[0] : nullptr; g.ln_rstd = ln ? ln[1] : nullptr; g.ln_gamma = ln ? ln[2] : nullptr; g.ln_beta = ln ? ln[3] : nullptr;
  static const int dbg = SR_TUNE_ENV("SRHIP_DEBUG") ? atoi(SR_TUNE_ENV("SRHIP_DEBUG")) : 0;
  g.debug = dbg;
  static const bool no_wide = SR_TUNE_ENV("SRHIP_NO_WIDE_STORE") != nullptr;
  // (N % 128 == 0: no ragged column tile in the 128- and 64-column kernels; the persistent kernel checks its own last tile)
  g.wide_store = (epilogue == SRHIP_EPI_BF16 || epilogue == SRHIP_EPI_GELU_BF16 || epilogue == SRHIP_EPI_DGELU_BF16) && !no_wide && !(dbg & 1) &&
                 (N % 128) == 0 && (ldc % 8) == 0 && (epilogue != SRHIP_EPI_GELU_BF16 || !aux_out || (ldaux % 4) == 0);
  const int grid = cdiv(M, BM) * cdiv(N, BN);
  hipStream_t s = (hipStream_t)stream;
  int splits = 1;
  int plan = gemm_plan(epilogue, M, N, K, beta, &splits);
  // the residual that owes its LayerNorm exists for the 64 x 64, 128 x 128 and two-wave-group kernels: where the plan says a lockstep 256-row
  // kernel (operands past 2 GiB, or a pinned test mode) it takes the 128 x 128 tiles instead
  if (ln && (plan == SRHIP_GEMM_PLAN_BIG256 || plan == SRHIP_GEMM_PLAN_BIG128 || plan == SRHIP_GEMM_PLAN_BIG2WG ||
             (plan == SRHIP_GEMM_PLAN_PP256 && !(((size_t)(M - 1) * lda + K) * 2 < (1ull << 31) && ((size_t)(N - 1) * ldb + K) * 2 < (1ull << 31)))))
    plan = SRHIP_GEMM_PLAN_TILE128;
  const int nkt = K / BK;
  g.ksplit_tiles = cdiv(nkt, splits);
  const dim3 grid3(grid, splits);
  if (plan == SRHIP_GEMM_PLAN_PP256 || plan == SRHIP_GEMM_PLAN_BIG256 || plan == SRHIP_GEMM_PLAN_BIG128 || plan == SRHIP_GEMM_PLAN_BIG2WG) {
    const int variant = (plan == SRHIP_GEMM_PLAN_BIG256 || plan == SRHIP_GEMM_PLAN_PP256) ? 0 : (plan == SRHIP_GEMM_PLAN_BIG128 ? 1 : 2);
    // 256 x 256 tiles: the two-wave-group kernel unless the real leading dimensions push an operand past 2 GiB
    const bool pp_ok = plan == SRHIP_GEMM_PLAN_PP256 && ((size_t)(M - 1) * lda + K) * 2 < (1ull << 31) && ((size_t)(N - 1) * ldb + K) * 2 < (1ull << 31);
    if (variant == 0 && pp_ok) {
      switch (epilogue) {
        case SRHIP_EPI_BF16: launch_pp<SRHIP_EPI_BF16>(g, s); break;
        case SRHIP_EPI_GELU_BF16: launch_pp<SRHIP_EPI_GELU_BF16>(g, s); break;
        case SRHIP_EPI_RESID_F32: if (ln) launch_pp<EPI_RESID_LN>(g, s); else launch_pp<SRHIP_EPI_RESID_F32>(g, s); break;
        case SRHIP_EPI_DGELU_BF16: launch_pp<SRHIP_EPI_DGELU_BF16>(g, s); break;
        default: return SR_EINVAL;
      }
      SR_CHECK_LAUNCH();
      return SR_OK;
    }
    switch (epilogue) {
      case SRHIP_EPI_BF16: launch_big<SRHIP_EPI_BF16>(g, variant, s); break;
      case SRHIP_EPI_GELU_BF16: launch_big<SRHIP_EPI_GELU_BF16>(g, variant, s); break;
      case SRHIP_EPI_RESID_F32: if (ln) return SR_EINVAL; launch_big<SRHIP_EPI_RESID_F32>(g, variant, s); break;
      case SRHIP_EPI_DGELU_BF16: launch_big<SRHIP_EPI_DGELU_BF16>(g, variant, s); break;
      default: return SR_EINVAL;
    }
    SR_CHECK_LAUNCH();
    return SR_OK;
  }
  if (plan == SRHIP_GEMM_PLAN_SMALL64) {
    const dim3 gs(cdiv(M, SBM) * cdiv(N, SBM));
    switch (epilogue) {
      case SRHIP_EPI_BF16: SR_LAUNCH(gemm_small_kernel<SRHIP_EPI_BF16>, gs, dim3(256), 0, s, g); break;
      case SRHIP_EPI_GELU_BF16: SR_LAUNCH(gemm_small_kernel<SRHIP_EPI_GELU_BF16>, gs, dim3(256), 0, s, g); break;
      case SRHIP_EPI_RESID_F32:
        if (ln) SR_LAUNCH(gemm_small_kernel<EPI_RESID_LN>, gs, dim3(256), 0, s, g); else SR_LAUNCH(gemm_small_kernel<SRHIP_EPI_RESID_F32>, gs, dim3(256), 0, s, g);
        break;
      case SRHIP_EPI_DGELU_BF16: SR_LAUNCH(gemm_small_kernel<SRHIP_EPI_DGELU_BF16>, gs, dim3(256), 0, s, g); break;
      default: return SR_EINVAL;
    }
    SR_CHECK_LAUNCH();
    return SR_OK;
  }
  switch (epilogue) {
    case SRHIP_EPI_BF16: SR_LAUNCH(gemm_nt_kernel<SRHIP_EPI_BF16>, grid3, dim3(256), 0, s, g); break;
    case SRHIP_EPI_GELU_BF16: SR_LAUNCH(gemm_nt_kernel<SRHIP_EPI_GELU_BF16>, grid3, dim3(256), 0, s, g); break;
    case SRHIP_EPI_RESID_F32:
      if (ln) SR_LAUNCH(gemm_nt_kernel<EPI_RESID_LN>, grid3, dim3(256), 0, s, g); else SR_LAUNCH(gemm_nt_kernel<SRHIP_EPI_RESID_F32>, grid3, dim3(256), 0, s, g);
      break;
    case SRHIP_EPI_DGELU_BF16: SR_LAUNCH(gemm_nt_kernel<SRHIP_EPI_DGELU_BF16>, grid3, dim3(256), 0, s, g); break;
    case SRHIP_EPI_F32: SR_LAUNCH(gemm_nt_kernel<SRHIP_EPI_F32>, grid3, dim3(256), 0, s, g); break;
    default: return SR_EINVAL;
  }
  SR_CHECK_LAUNCH();
  return SR_OK;
}

#ifdef SRHIP_TUNING
extern "C" int srhip_gemm_debug(long long* out_host, int n) {
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(srhip_gemm_dbg), (size_t)n * sizeof(long long)) == hipSuccess ? SR_OK : SR_EINVAL;
}
#endif
extern "C" int srhip_gemm_nt(int epilogue, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                             int M, int N, int K, const float* bias, const float* row_scale, int rows_per_sample,
                             const void* aux_in, void* aux_out, int ldaux, float alpha, float beta, void* stream) {
  return gemm_nt_impl(epilogue, A, lda, B, ldb, C, ldc, M, N, K, bias, row_scale, rows_per_sample, aux_in, aux_out, ldaux, alpha, beta, 0u, 0u,
                      1.0f, stream);
}

extern "C" int srhip_gemm_nt_dropout(int epilogue, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                                     const float* bias, const void* aux_in, void* aux_out, int ldaux, unsigned drop_key, unsigned drop_thresh,
                                     float drop_scale, void* stream) {
  if (epilogue != SRHIP_EPI_GELU_BF16 && epilogue != SRHIP_EPI_DGELU_BF16 && epilogue != SRHIP_EPI_RESID_F32) return SR_EINVAL;
  if ((ldc != N || (N & 1)) && drop_thresh) return SR_EINVAL;        // (one dropout hash per aligned element pair of the [M, N] output)
  return gemm_nt_impl(epilogue, A, lda, B, ldb, C, ldc, M, N, K, bias, nullptr, 0, aux_in, aux_out, ldaux, 1.0f, 0.0f, drop_key, drop_thresh,
                      drop_scale, stream);
}

extern "C" int srhip_gemm_nt_resid_dropout(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K,
                                           const float* bias, const float* resid, int ldresid, unsigned drop_key, unsigned drop_thresh,
                                           float drop_scale, void* stream) {
  if ((ldc != N || (N & 1)) && drop_thresh) return SR_EINVAL;        // the dropout index is the row-major index of the [M, N] output; one hash per aligned pair
  return gemm_nt_impl(SRHIP_EPI_RESID_F32, A, lda, B, ldb, C, ldc, M, N, K, bias, nullptr, 0, resid, nullptr, ldresid, 1.0f, 0.0f, drop_key,
                      drop_thresh, drop_scale, stream);
}

// ... with the residual taken as LayerNorm(C) of the PRE-LayerNorm sums that C holds (in place): C = LN(C; mean, rstd, gamma, beta) + dropout(A B^T +
// bias).  The post-LN encoders' inference rows then never materialise the fp32 LayerNorm output (srhip_postln_fwd with x == NULL writes the
// bf16 operand and the statistics only: 6 instead of 10 bytes per element of a launch that is 11 % of the BERT leg's kernel time).
extern "C" int srhip_gemm_nt_resid_ln_dropout(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K,
                                              const float* bias, const float* ln_mean, const float* ln_rstd, const float* ln_gamma,
                                              const float* ln_beta, unsigned drop_key, unsigned drop_thresh, float drop_scale, void* stream) {
  if ((ldc != N || (N & 1)) && drop_thresh) return SR_EINVAL;
  if (!ln_mean || !ln_rstd || !ln_gamma || !ln_beta || (((uintptr_t)ln_gamma | (uintptr_t)ln_beta) & 15)) return SR_EINVAL;
  const float* ln[4] = {ln_mean, ln_rstd, ln_gamma, ln_beta};
  return gemm_nt_impl(SRHIP_EPI_RESID_F32, A, lda, B, ldb, C, ldc, M, N, K, bias, nullptr, 0, nullptr, nullptr, 0, 1.0f, 0.0f, drop_key,
                      drop_thresh, drop_scale, stream, ln);
}

extern "C" int srhip_gemm_nt_grouped_n64_f32(const srhip_group_desc* desc_dev, int n_problems, int total_tiles, float alpha, float beta,
                                             void* stream) {
  if (!desc_dev || n_problems <= 0 || n_problems > 4096 || total_tiles <= 0) return SR_EINVAL;
  SR_LAUNCH(gemm_grouped_n64_f32_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, desc_dev, n_problems, alpha, beta);
  SR_CHECK_LAUNCH();
  return SR_OK;
}

extern "C" int srhip_gemm_nt_grouped_f32(const srhip_group_desc* desc_dev, int n_problems, int total_tiles, float alpha, float beta,
                                         void* stream) {
  if (!desc_dev || n_problems <= 0 || n_problems > 4096 || total_tiles <= 0) return SR_EINVAL;
  SR_LAUNCH(gemm_grouped_f32_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, desc_dev, n_problems, alpha, beta);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
