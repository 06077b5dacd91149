// Probe: lane layout of v_mfma_f32_32x32x2_f32 (A[i=l&31][k=l>>5], B[k=l>>5][j=l&31] expected).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void k(float* o) {
  int l = threadIdx.x;
  float a = (float)(1 + l), b = (float)(100 * (1 + l));
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) o[l * 16 + r] = c[r];
}
int main() {
  float* d; hipMalloc(&d, 64 * 16 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[1024]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
    int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float exp = (1.f + row) * 100.f * (1 + col) + (1.f + row + 32) * 100.f * (1 + col + 32);
    if (h[l * 16 + r] != exp) { if (bad < 8) printf("l=%d r=%d got %g exp %g\n", l, r, h[l*16+r], exp); ++bad; }
  }
  printf("bad=%d\n", bad);
  return 0;
}
