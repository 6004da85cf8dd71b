// probe: does hipcc keep  (int)(ix * scale + offset)  (float64, separate roundings) bit-identical to CPython?
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)
__global__ void k(int lo, int hi, int* out, double* dbg) {
  const int t = threadIdx.x;
  const double scale = 255.0 / (double)(hi - lo), offset = -(double)lo * scale;
  const double v = (double)t * scale + offset;
  out[t] = (int)v;
  dbg[t] = v;
  if (t == 0) { dbg[256] = scale; dbg[257] = offset; }
}
int main(int argc, char** argv) {
  int lo = atoi(argv[1]), hi = atoi(argv[2]);
  int* o; double* d;
  hipMalloc(&o, 256 * 4); hipMalloc(&d, 258 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, lo, hi, o, d);
  int ho[256]; double hd[258];
  hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost); hipMemcpy(hd, d, sizeof hd, hipMemcpyDeviceToHost);
  printf("%a %a\n", hd[256], hd[257]);
  for (int i = 0; i < 256; ++i) printf("%d %a\n", ho[i], hd[i]);
  return 0;
}
