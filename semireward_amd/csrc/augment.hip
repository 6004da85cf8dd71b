// Device-side image augmentation of the USB CV input pipeline: RandomCrop(reflect padding) + RandomHorizontalFlip +
// RandAugment(n ops) + Cutout + ToTensor + Normalize for a whole batch in ONE launch, uint8 HWC in, normalised fp32 NCHW out.
//
// Replaces (host CPU, 12 PIL worker processes in the reference):
//   semilearn/datasets/cv_datasets/cifar.py:34-49          transform_weak / transform_strong
//   semilearn/datasets/augmentation/randaugment.py:16-196  the 14 ops, Cutout, RandAugment.__call__
// Byte/integer work, bit-exact with Pillow's algorithms (lookup tables from per-channel histograms, float32 blends with truncation,
// 3x3 smoothing with copied borders, 16.16 fixed-point nearest-neighbour affine walk) -- oracle/augment_ref.py restates them and is pinned to
// the reference's own functions.  One 256-thread workgroup per image; the image ping-pongs between two global scratch planes (L2-resident:
// 3 KB at 32x32, 150 KB at 224x224), histograms / tables / reductions live in LDS.  Every random draw is an input (parameter blocks filled
// by semireward_amd/data/augment.py); floating-point contraction is switched off for the file so that no FMA changes a byte.
#include "common.h"
#include "srhip.h"

// hipcc contracts a * b + c into an FMA by default; Pillow's C code and CPython round after the multiply, and one differently rounded product
// moves a truncated byte by 1 (tools/f64_probe.hip: with contraction off the float64 table arithmetic is bit-identical to CPython's).
#pragma clang fp contract(off)

namespace {

constexpr int IPN = 64, DPN = 32, MAXS = 256;
enum { OP_AUTOCONTRAST = 0, OP_BRIGHTNESS, OP_COLOR, OP_CONTRAST, OP_EQUALIZE, OP_IDENTITY, OP_POSTERIZE, OP_ROTATE, OP_SHARPNESS, OP_SHEARX,
       OP_SHEARY, OP_SOLARIZE, OP_TRANSLATEX, OP_TRANSLATEY };

__device__ __forceinline__ unsigned char blend8(int a, int b, float al) {
  // Pillow Blend.c: (UINT8)((int)a + alpha * ((int)b - (int)a)), float32 multiply then add (no contraction)
  const float t = (float)a + al * (float)(b - a);
  return (unsigned char)(int)t;
}
__device__ __forceinline__ int gray8(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

__global__ __launch_bounds__(256) void augment_kernel(const unsigned char* __restrict__ src, int H0, int W0, int S, int pad,
                                                     const int* __restrict__ ip, const double* __restrict__ dp,
                                                     unsigned char* __restrict__ scratch, float* __restrict__ out,
                                                     unsigned char* __restrict__ out_u8, float m0, float m1, float m2, float s0, float s1, float s2) {
  __shared__ int hist[3][256];
  __shared__ int lut[3][256];
  __shared__ int tabx[MAXS], taby[MAXS];
  __shared__ long long red[256];
  __shared__ int sh_i[8];
  const int b = blockIdx.x, tid = threadIdx.x, npx = S * S;
  const int* ipb = ip + (size_t)b * IPN;
  const double* dpb = dp + (size_t)b * DPN;
  unsigned char* A = scratch + (size_t)b * 2 * npx * 3;
  unsigned char* B = A + (size_t)npx * 3;
  // ---- RandomCrop (reflect padding) + RandomHorizontalFlip
  {
    const int ci = ipb[0], cj = ipb[1], flip = ipb[2];
    const unsigned char* im = src + (size_t)ipb[8] * H0 * W0 * 3;
    for (int p = tid; p < npx; p += 256) {
      const int y = p / S, x = p - y * S, xs = flip ? S - 1 - x : x;
      int py = ci + y - pad, px = cj + xs - pad;
      py = py < 0 ? -py : (py >= H0 ? 2 * (H0 - 1) - py : py);
      px = px < 0 ? -px : (px >= W0 ? 2 * (W0 - 1) - px : px);
      const unsigned char* s = im + ((size_t)py * W0 + px) * 3;
      A[p * 3] = s[0]; A[p * 3 + 1] = s[1]; A[p * 3 + 2] = s[2];
    }
  }
  __syncthreads();
  const int n_ops = ipb[3];
  for (int k = 0; k < n_ops; ++k) {
    const int* q = ipb + 16 + 12 * k;
    const double* d = dpb + 8 * k;
    const int op = q[0];
    if (op == OP_IDENTITY) continue;
    if (op == OP_AUTOCONTRAST || op == OP_EQUALIZE || op == OP_POSTERIZE || op == OP_SOLARIZE) {
      // ---- lookup-table ops (ImageOps)
      if (op == OP_AUTOCONTRAST || op == OP_EQUALIZE) {
        for (int i = tid; i < 768; i += 256) (&hist[0][0])[i] = 0;
        __syncthreads();
        for (int p = tid; p < npx * 3; p += 256) atomicAdd(&hist[p % 3][A[p]], 1);
        __syncthreads();
      }
      if (op == OP_AUTOCONTRAST) {
        if (tid < 3) {
          int lo = 0, hi = 255;
          while (lo < 255 && hist[tid][lo] == 0) ++lo;
          while (hi > 0 && hist[tid][hi] == 0) --hi;
          sh_i[tid * 2] = lo; sh_i[tid * 2 + 1] = hi;
        }
        __syncthreads();
        for (int c = 0; c < 3; ++c) {
          const int lo = sh_i[c * 2], hi = sh_i[c * 2 + 1];
          int v = tid;
          if (hi > lo) {
            const double scale = ((255.0) / ((double)(hi - lo))), offset = ((-(double)lo) * (scale));
            const int t = (int)((double)tid * scale + offset);
            v = t < 0 ? 0 : (t > 255 ? 255 : t);
          }
          lut[c][tid] = v;
        }
      } else if (op == OP_EQUALIZE) {
        if (tid < 3) {
          int nz = 0, last = 0;
          long long total = 0;
          for (int i = 0; i < 256; ++i) if (hist[tid][i]) { ++nz; last = i; total += hist[tid][i]; }
          const long long step = nz > 1 ? (total - hist[tid][last]) / 255 : 0;
          if (!step) {
            for (int i = 0; i < 256; ++i) lut[tid][i] = i;
          } else {
            long long n = step / 2;
            for (int i = 0; i < 256; ++i) { const long long v = n / step; lut[tid][i] = v > 255 ? 255 : (int)v; n += hist[tid][i]; }
          }
        }
      } else if (op == OP_POSTERIZE) {
        for (int c = 0; c < 3; ++c) lut[c][tid] = tid & q[8];
      } else {
        const double th = d[0];
        for (int c = 0; c < 3; ++c) lut[c][tid] = ((double)tid < th) ? tid : 255 - tid;
      }
      __syncthreads();
      for (int p = tid; p < npx * 3; p += 256) A[p] = (unsigned char)lut[p % 3][A[p]];
      __syncthreads();
    } else if (op == OP_BRIGHTNESS || op == OP_COLOR || op == OP_CONTRAST || op == OP_SHARPNESS) {
      // ---- ImageEnhance: blend(degenerate, image, alpha)
      const float al = (float)d[0];
      int mean = 0;
      if (op == OP_CONTRAST) {
        long long s = 0;
        for (int p = tid; p < npx; p += 256) s += gray8(A[p * 3], A[p * 3 + 1], A[p * 3 + 2]);
        red[tid] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
        mean = (int)(((((double)red[0]) / ((double)npx))) + (0.5));
        __syncthreads();
      }
      if (op == OP_SHARPNESS) {
        // ImageFilter.SMOOTH -> B (float32, rows y+1, y, y-1; starts from 0.5; borders copied)
        const float k1 = ((1.0f) / (13.0f)), k5 = ((5.0f) / (13.0f));
        for (int p = tid; p < npx * 3; p += 256) {
          const int c = p % 3, px = p / 3, y = px / S, x = px - y * S;
          unsigned char r = A[p];
          if (y > 0 && y < S - 1 && x > 0 && x < S - 1) {
            float ss = 0.5f;
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {
              const int yy = y + 1 - rr;
              const unsigned char* row = A + ((size_t)yy * S) * 3 + c;
              const float kc = rr == 1 ? k5 : k1;
              float t = (float)row[(x - 1) * 3] * k1 + (float)row[x * 3] * kc;
              t = t + (float)row[(x + 1) * 3] * k1;
              ss = ((ss) + (t));
            }
            r = ss <= 0.0f ? 0 : (ss >= 255.0f ? 255 : (unsigned char)(int)ss);
          }
          B[p] = r;
        }
        __syncthreads();
      }
      for (int p = tid; p < npx; p += 256) {
        const int r = A[p * 3], g = A[p * 3 + 1], bl = A[p * 3 + 2];
        int d0, d1, d2;
        if (op == OP_BRIGHTNESS) d0 = d1 = d2 = 0;
        else if (op == OP_COLOR) d0 = d1 = d2 = gray8(r, g, bl);
        else if (op == OP_CONTRAST) d0 = d1 = d2 = mean;
        else { d0 = B[p * 3]; d1 = B[p * 3 + 1]; d2 = B[p * 3 + 2]; }
        A[p * 3] = blend8(d0, r, al); A[p * 3 + 1] = blend8(d1, g, al); A[p * 3 + 2] = blend8(d2, bl, al);
      }
      __syncthreads();
    } else {
      // ---- Image.rotate / Image.transform(AFFINE), NEAREST, fill 0: gather A -> B
      if (q[1]) {                               // no rotation / shear: ImagingScaleAffine, float64 accumulation of the source coordinate
        if (tid == 0) {
          double v = d[1];
          for (int x = 0; x < S; ++x) { tabx[x] = v < 0 ? -1 : (int)v; v = ((v) + (d[3])); }
          v = d[2];
          for (int y = 0; y < S; ++y) { taby[y] = v < 0 ? -1 : (int)v; v = ((v) + (d[4])); }
        }
        __syncthreads();
        for (int p = tid; p < npx; p += 256) {
          const int y = p / S, x = p - y * S, xin = tabx[x], yin = taby[y];
          const bool ok = xin >= 0 && xin < S && yin >= 0 && yin < S;
          const unsigned char* s = A + ((size_t)yin * S + xin) * 3;
          B[p * 3] = ok ? s[0] : 0; B[p * 3 + 1] = ok ? s[1] : 0; B[p * 3 + 2] = ok ? s[2] : 0;
        }
      } else {                                  // Geometry.c affine_fixed: 16.16 fixed point
        const int a0 = q[2], a1 = q[3], a2 = q[4], a3 = q[5], a4 = q[6], a5 = q[7];
        for (int p = tid; p < npx; p += 256) {
          const int y = p / S, x = p - y * S;
          const int xin = (a2 + y * a1 + x * a0) >> 16, yin = (a5 + y * a4 + x * a3) >> 16;
          const bool ok = xin >= 0 && xin < S && yin >= 0 && yin < S;
          const unsigned char* s = A + ((size_t)(ok ? yin : 0) * S + (ok ? xin : 0)) * 3;
          B[p * 3] = ok ? s[0] : 0; B[p * 3 + 1] = ok ? s[1] : 0; B[p * 3 + 2] = ok ? s[2] : 0;
        }
      }
      __syncthreads();
      unsigned char* t = A; A = B; B = t;
    }
  }
  // ---- Cutout (ImageDraw.rectangle, both ends inclusive), ToTensor + Normalize
  const int xa = ipb[4], ya = ipb[5], xb = ipb[6], yb = ipb[7];
  const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
  const unsigned char cc[3] = {125, 123, 114};
  for (int p = tid; p < npx * 3; p += 256) {
    const int c = p % 3, px = p / 3, y = px / S, x = px - y * S;
    unsigned char v = A[p];
    if (xa >= 0 && x >= xa && x <= xb && y >= ya && y <= yb) v = cc[c];
    if (out_u8) out_u8[(size_t)b * npx * 3 + p] = v;
    out[((size_t)b * 3 + c) * npx + px] = ((float)v / 255.0f - mean[c]) / stdv[c];
  }
}

}  // namespace

extern "C" int srhip_augment(const unsigned char* src, int n_src, int H0, int W0, int B, int S, int pad, const int* ip, const double* dp,
                             unsigned char* scratch, float* out, unsigned char* out_u8, const float* mean3, const float* std3, void* stream) {
  if (!src || !ip || !dp || !scratch || !out || !mean3 || !std3 || B <= 0 || S <= 1 || S > MAXS || pad < 0 || pad >= H0 || pad >= W0 || n_src <= 0)
    return SR_EINVAL;
  if (H0 + 2 * pad < S || W0 + 2 * pad < S) return SR_EINVAL;
  SR_LAUNCH(augment_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, src, H0, W0, S, pad, ip, dp, scratch, out, out_u8, mean3[0],
                     mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  SR_CHECK_LAUNCH();
  return SR_OK;
}
