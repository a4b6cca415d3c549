// Does v_mfma_f32_32x32x16_f16 keep f16 subnormal INPUTS on gfx950?  (MI200 flushed them; the exact-f32 K1's f16 route
// relies on subnormal planes carrying an absolute error of 2^-25, not being flushed.)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_f16_denorm tools/mfma_f16_denorm.hip && /tmp/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
__global__ void k(float* out) {
  const int lane = threadIdx.x;
  half8 a, b;
  // A row r (= lane & 31), k-slice 8 (lane >> 5) .. +7: every element 2^-20 (subnormal in f16: min normal 2^-14)
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)9.5367431640625e-07f; b[e] = (_Float16)1024.0f; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);      // 16 products of 2^-20 * 2^10 = 16 * 2^-10 = 2^-6
  half8 a2, b2;
  for (int e = 0; e < 8; ++e) { a2[e] = (_Float16)6.103515625e-05f; b2[e] = (_Float16)5.9604644775390625e-08f; }   // 2^-14 (normal) x 2^-24 (smallest subnormal)
  f32x16 d = {0};
  d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b2, d, 0, 0, 0);    // 16 * 2^-38 = 2^-34: a normal f32
  if (lane == 0) { out[0] = c[0]; out[1] = d[0]; out[2] = (float)a[0]; out[3] = (float)b2[0]; }
}
int main() {
  float* o; hipMalloc(&o, 16); hipMemset(o, 0, 16);
  k<<<1, 64>>>(o);
  float h[4]; hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
  printf("sum of 16 products 2^-20 x 2^10: %.9g (expected 0.015625 if subnormal inputs are kept, 0 if flushed)\n", h[0]);
  printf("sum of 16 products 2^-14 x 2^-24: %.9g (expected %.9g)\n", h[1], 16.0 * 6.103515625e-05 * 5.9604644775390625e-08);
  printf("operands as stored: %.9g %.9g\n", h[2], h[3]);
  return 0;
}
