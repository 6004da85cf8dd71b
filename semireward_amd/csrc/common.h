// Shared device helpers for libsrhip (gfx950 / CDNA4 only: wave = 64, MFMA 16x16x32 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <tuple>
#include <utility>

#define SR_OK 0
#define SR_EINVAL (-1)
#define SR_ELAUNCH (-2)

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

#define SR_CHECK_LAUNCH()                                  \
  do {                                                     \
    hipError_t e__ = hipGetLastError();                    \
    if (e__ != hipSuccess) return SR_ELAUNCH - (int)e__;   \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// fp32 -> bf16, round-to-nearest-even, on the gfx950 hardware converter (v_cvt_pk_bf16_f32: one instruction per PAIR).
// (A hand-rolled integer RNE with a NaN test compiled to one divergent exec-mask region per value -- 209 of them in
// the attention kernel -- and dominated its runtime.)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }

// raw v_exp_f32 (2^x): arguments on the hot paths are <= 0 (x - rowmax), so the OCML range/denormal fix-up is dead weight
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// all-reduce over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15) on the vector ALU: quad swaps, half-row mirror, row mirror -- four
// full-rate adds instead of four ds_bpermute round trips (__shfl_xor)
__device__ __forceinline__ float row16_sum(float v) {
#define SR_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
  SR_DPP_ADD(0xB1);      // quad_perm [1, 0, 3, 2]
  SR_DPP_ADD(0x4E);      // quad_perm [2, 3, 0, 1]
  SR_DPP_ADD(0x141);     // row_half_mirror
  SR_DPP_ADD(0x140);     // row_mirror
#undef SR_DPP_ADD
  return v;
}

// all-reduce over the four 16-lane rows of a wave with the gfx950 row-swap instructions (instead of two ds_bpermute round trips):
// permlane16_swap(a, b) = {a.r0 b.r0 a.r2 b.r2, a.r1 b.r1 a.r3 b.r3} (tools/pl_probe.hip), permlane32_swap(a, b) = {a.lo b.lo, a.hi b.hi}
__device__ __forceinline__ float rows_sum4(float x) {
  const unsigned u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned v = __float_as_uint(__uint_as_float(a[0]) + __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// erf via Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, branch-free, one v_exp + one v_rcp).  The OCML erff costs ~40
// instructions with branches: in the fc1 epilogue it took 80 us of a 220 us GEMM.  The GELU outputs are rounded to bf16
// (2^-9 relative) right after, so 1.5e-7 absolute is invisible; semantics stay nn.GELU() "exact erf" (vit.py:63,72).
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float r = 1.0f - poly * __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
  return copysignf(r, x);
}
// GELU(x) = x/2 + |x/2| * erf(|x|/sqrt2) with the same 7.1.26 polynomial, constants folded so that the exponential is a bare
// v_exp_f32 of -(w*w) (w = |x| sqrt(log2 e / 2)): 11 full-rate VALU ops + v_rcp + v_exp.  (The fused MLP kernel evaluates
// 196608 of these per 128-row tile; at ~28 ops each the VALU time exceeded the MFMA time of the two products.)
__device__ __forceinline__ float gelu_erf(float x) {
  const float w = fabsf(x) * 0.8493218f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.27273747f, w, 1.0f));
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float r = fmaf(-poly, __builtin_amdgcn_exp2f(-(w * w)), 1.0f);   // erf(|x| / sqrt 2)
  const float h = 0.5f * x;
  return fmaf(fabsf(h), r, h);
}
// GELU of TWO values without transcendentals, for the fused projection + MLP kernel (rows without a backward), whose main loop is bound by the
// vector ALU: gelu(x) = x * Phi(x), Phi(x) = 1/2 + xc * R(xc^2), xc = x clamped to +-X*, R = a degree-8 polynomial in xc^2 (weighted minimax
// fit of erf(x / sqrt 2) / (2 x) on [0, 4.25]); X* = 4.252893 is where the polynomial's Phi reaches 1 (so beyond it Phi is exactly 0 / 1 up to
// fp32 rounding of the Horner chain, 5e-6).  15 instructions per PAIR (2 v_med3 + 12 packed fp32 + the bf16 pack) against 11 + v_rcp + v_exp
// (quarter rate: 19 issue slots) per VALUE for gelu_erf.  |gelu - exact| <= max(7e-5, 5e-6 |x|): 6.7e-5 inside the clamp, |x| times the
// fp32 residue of Phi(+-X*) beyond it (tests/test_gpu_kernels.py pins the bound) -- below half a bf16 quantum of the result wherever
// |gelu| >= 0.03, and the result is rounded to bf16 right after.  The gradient rows keep gelu_erf (their pre-activation and GELU output feed the backward).
__device__ __forceinline__ f32x2_t gelu_poly2(f32x2_t v) {
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
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.5f * x * x * 1.4426950408889634f);
  return cdf + x * pdf;
}

// Counter-based dropout generator (train-mode nn.Dropout of the BERT / Wav2Vec2 encoders, semilearn/nets/bert/bert.py:15,36 and the HF
// modules it calls): element with row-major linear index i of a dropout site is KEPT iff fmix32(i * 0x9E3779B1 + key) >= thresh, with
// thresh = floor(p * 2^32) and key = the 32-bit site key the host derives from (call seed, site id).  Stateless, so the backward
// regenerates the mask instead of storing it; oracle/bert_ref.py:keep_mask is the same function in numpy.
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
// Dropout decisions: ONE hash decides TWO neighbouring elements -- element i of a site (row-major linear index) belongs to pair i >> 1; the even
// element is kept iff the low 16 bits of fmix32(pair * 0x9E3779B1 + key) reach the threshold's upper 16 bits (p resolved to 2^-16), the odd one
// iff the high 16 bits do.  The hash costs two quarter-rate 32-bit multiplies: per element it was 74 % of the vector-ALU work of the attention
// forward at L = 512 and 7 of 26 us of a 256 x 256 GEMM tile whose epilogue carries a dropout (Wav2Vec2's activation dropout on the
// 3072-wide hidden rows).  oracle/bert_ref.keep_mask restates it; every fixture with a dropout was regenerated through the reference.
__device__ __forceinline__ bool drop_pair_keep(uint32_t h, int odd, uint32_t thresh) { return (odd ? (h >> 16) : (h & 0xffffu)) >= (thresh >> 16); }
__device__ __forceinline__ bool drop_keep(uint32_t idx, uint32_t key, uint32_t thresh) {
  return drop_pair_keep(fmix32((idx >> 1) * 0x9E3779B1u + key), (int)(idx & 1u), thresh);
}
// four consecutive elements i0 .. i0 + 3 (i0 even: every row length in use is even and lanes own aligned quads): two hashes
__device__ __forceinline__ void drop_keep4(uint32_t i0, uint32_t key, uint32_t thresh, bool (&k)[4]) {
  const uint32_t h0 = fmix32((i0 >> 1) * 0x9E3779B1u + key), h1 = fmix32(((i0 >> 1) + 1u) * 0x9E3779B1u + key);
  k[0] = drop_pair_keep(h0, 0, thresh); k[1] = drop_pair_keep(h0, 1, thresh);
  k[2] = drop_pair_keep(h1, 0, thresh); k[3] = drop_pair_keep(h1, 1, thresh);
}
// two consecutive elements i0, i0 + 1 (i0 even): one hash
__device__ __forceinline__ void drop_keep2(uint32_t i0, uint32_t key, uint32_t thresh, bool& k0, bool& k1) {
  const uint32_t h = fmix32((i0 >> 1) * 0x9E3779B1u + key);
  k0 = drop_pair_keep(h, 0, thresh); k1 = drop_pair_keep(h, 1, thresh);
}
// Attention probabilities [B, H, N, N] (the one 4-D site): pairs are taken WITHIN a query row -- pair index = row * ceil(N / 2) + (key >> 1) with
// row = (b H + h) N + q -- so that a lane's four consecutive keys are two whole pairs whatever the parity of N.
__device__ __forceinline__ uint32_t drop_pair_hash(uint32_t row, uint32_t nh, uint32_t pair, uint32_t key) {
  return fmix32((row * nh + pair) * 0x9E3779B1u + key);
}
// one element (the backward's key-on-lane layouts, where neighbouring registers are neighbouring QUERIES)
__device__ __forceinline__ bool drop_keep_attn(uint32_t row, uint32_t N, uint32_t key_idx, uint32_t key, uint32_t thresh) {
  return drop_pair_keep(drop_pair_hash(row, (N + 1u) >> 1, key_idx >> 1, key), (int)(key_idx & 1u), thresh);
}

// XCD-aware bijective block remap (blocks b, b+8, b+16.. share an XCD/L2): returns the
// logical work-group id so that each XCD walks a contiguous chunk of the tile list.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Tuning switches (tile splits, knock-outs, schedule variants for tools/microbench.py and the A/B scripts) exist in the tuning build only
// (SRHIP_TUNING_BUILD=1 -> -DSRHIP_TUNING).  The shipped library reads two environment variables, both test hooks: SRHIP_GEMM = tile | small | big256 | big128 | big2wg (force that
// kernel), bigold (the lockstep 256 x 256 kernel wherever the plan says 256 x 256), bigoldf (forced), big256r8 (eight-slot ring) (pins the GEMM
// tile kernel so that tests/test_gpu_kernels.py reaches every kernel with every epilogue) and SRHIP_FLEXMATCH_GENERAL (the global-memory
// FlexMatch path that tables too large for LDS take).
#ifdef SRHIP_TUNING
#include <stdlib.h>
#define SR_TUNE_ENV(name) getenv(name)
#else
#define SR_TUNE_ENV(name) ((const char*)nullptr)
#endif

// Every launch of the library: the plain launch, or -- while bench.py's roofline pass has profiling on (prof.hip) -- the same launch with a
// (start, stop) event pair bound to the dispatch, whose elapsed time is the kernel's own execution time.
extern bool g_sr_prof_on;
bool sr_prof_take(hipEvent_t* es, hipEvent_t* ee);
template <typename... P, size_t... I, typename... A>
static inline void sr_launch_timed_(void (*kern)(P...), std::index_sequence<I...>, dim3 g, dim3 b, unsigned sm, hipStream_t s, hipEvent_t es,
                                    hipEvent_t ee, A&&... a) {
  std::tuple<P...> params{static_cast<P>(a)...};          // the arguments converted to the kernel's parameter types
  void* ptrs[] = {(void*)&std::get<I>(params)...};
  (void)hipExtLaunchKernel((const void*)kern, g, b, ptrs, sm, s, es, ee, 0);
}
template <typename... P, typename... A>
static inline void sr_launch_timed(void (*kern)(P...), dim3 g, dim3 b, unsigned sm, hipStream_t s, hipEvent_t es, hipEvent_t ee, A&&... a) {
  sr_launch_timed_(kern, std::index_sequence_for<P...>{}, g, b, sm, s, es, ee, static_cast<A&&>(a)...);
}
#define SR_LAUNCH(kern, grid, block, sm, stream, ...)                                                                \
  do {                                                                                                              \
    hipEvent_t es__, ee__;                                                                                          \
    if (g_sr_prof_on && sr_prof_take(&es__, &ee__)) sr_launch_timed(kern, grid, block, sm, stream, es__, ee__, __VA_ARGS__); \
    else hipLaunchKernelGGL(kern, grid, block, sm, stream, __VA_ARGS__);                                            \
  } while (0)
