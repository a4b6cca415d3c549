// microbench2.hip — what does the K1 softmax epilogue really cost per 32x32 tile, and how well does
// it overlap with the 4 MFMAs of the next tile?  (opaque-register tricks stop hoisting)
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;
constexpr int ITERS = 2048;
#define OPAQUE(x) asm volatile("" : "+v"(x))

template <int MODE>
__global__ void k(float* out) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3c00 + (threadIdx.x & 7)); b[i] = (short)(0x3c00 + i); }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = -0.01f * (i + (threadIdx.x & 3));
  float l = 0.f, l2 = 0.f, m = 1.0f;
  for (int it = 0; it < ITERS; ++it) {
    if (MODE & 1) {  // 4 MFMAs
      OPAQUE(a); OPAQUE(b);
      f32x16 z = {0};
      z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, z, 0, 0, 0);
      z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, z, 0, 0, 0);
      z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, z, 0, 0, 0);
      z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, z, 0, 0, 0);
      if (MODE & 64) { c = z; }  // consume depends on THIS tile's MFMAs
      else { float s = z[0]; OPAQUE(s); m += s * 0.f; }
    }
    for (int i = 0; i < 16; ++i) OPAQUE(c[i]);
    if (MODE & 2) {  // max tree
      float t = fmaxf(fmaxf(fmaxf(c[0], c[1]), c[2]), fmaxf(fmaxf(c[3], c[4]), c[5]));
      t = fmaxf(t, fmaxf(fmaxf(fmaxf(c[6], c[7]), c[8]), fmaxf(fmaxf(c[9], c[10]), c[11])));
      t = fmaxf(t, fmaxf(fmaxf(fmaxf(c[12], c[13]), c[14]), c[15]));
      m = fmaxf(m, t * 1e-3f);
    }
    if (MODE & 4) {  // fma + exp + add, one accumulator
#pragma unroll
      for (int i = 0; i < 16; ++i) l += __builtin_amdgcn_exp2f(__builtin_fmaf(c[i], 1.44f, -m));
    }
    if (MODE & 8) {  // exp + add only (prescaled logits, C-init carries -M2)
#pragma unroll
      for (int i = 0; i < 16; ++i) l += __builtin_amdgcn_exp2f(c[i]);
    }
    if (MODE & 16) {  // fma + exp, two accumulators
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        l += __builtin_amdgcn_exp2f(__builtin_fmaf(c[i], 1.44f, -m));
        l2 += __builtin_amdgcn_exp2f(__builtin_fmaf(c[i + 1], 1.44f, -m));
      }
    }
    if (MODE & 32) {  // exp only (no add): keep results alive
#pragma unroll
      for (int i = 0; i < 16; ++i) { float e = __builtin_amdgcn_exp2f(c[i]); OPAQUE(e); }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = l + l2 + m;
}

template <int MODE>
void run(const char* name, float* out) {
  for (int wps : {1, 2, 3, 4, 8}) {
    const int blocks = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, blocks, 256, 0, 0, out); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, blocks, 256, 0, 0, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s wps %d : %7.1f cyc/tile/SIMD @2.4GHz\n", name, wps, ms / 5 * 1e-3 * 2.4e9 / (ITERS * (double)wps));
  }
}

int main() {
  float* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  run<1>("4 mfma only", out);
  run<2>("max tree (8 max3)", out);
  run<4>("16x(fma,exp,add)", out);
  run<8>("16x(exp,add)", out);
  run<16>("16x(fma,exp,add) 2 accumulators", out);
  run<32>("16x exp", out);
  run<2 | 4>("max + 16x(fma,exp,add)", out);
  run<1 | 2 | 4>("4 mfma (indep) + max + 16x(fma,exp,add)", out);
  run<1 | 2 | 4 | 64>("4 mfma -> consumed by max + 16x(fma,exp,add)", out);
  run<1 | 2 | 8 | 64>("4 mfma -> consumed by max + 16x(exp,add)", out);
  return 0;
}
