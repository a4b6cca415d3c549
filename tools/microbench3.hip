// microbench3.hip — exact instruction ORDER of the K1 epilogue (hand-pinned with asm volatile):
// how do dependent fma->exp->add chains issue, and does grouping independent ops help?
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;
constexpr int ITERS = 2048;

#define FMA(d, s) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(s), "v"(c1), "v"(nm))
#define EXP(d) asm volatile("v_exp_f32 %0, %0" : "+v"(d))
#define ADD(acc, s) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(s))
#define MFMA(c, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define MFMA0(c, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b))

template <int ORDER, bool WITH_MFMA>
__global__ void k(float* out) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3c00 + (threadIdx.x & 7)); b[i] = (short)(0x3c00 + i); }
  f32x16 cur, nxt;
  for (int i = 0; i < 16; ++i) cur[i] = -0.01f * (i + (threadIdx.x & 3));
  nxt = cur;
  float c1 = 1.44f, nm = -1.0f, l0 = 0.f, l1 = 0.f;
  float t[16];
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (WITH_MFMA) { if (s == 0) MFMA0(nxt, a, b); else MFMA(nxt, a, b); }
      if (ORDER == 1) {        // f e f e f e f e a a a a, one accumulator
#pragma unroll
        for (int e = 0; e < 4; ++e) { FMA(t[4 * s + e], cur[4 * s + e]); EXP(t[4 * s + e]); }
#pragma unroll
        for (int e = 0; e < 4; ++e) ADD(l0, t[4 * s + e]);
      } else if (ORDER == 2) { // f f f f e e e e, adds of the previous group, two accumulators
#pragma unroll
        for (int e = 0; e < 4; ++e) FMA(t[4 * s + e], cur[4 * s + e]);
#pragma unroll
        for (int e = 0; e < 4; ++e) EXP(t[4 * s + e]);
        if (s > 0) {
          ADD(l0, t[4 * s - 4]); ADD(l1, t[4 * s - 3]); ADD(l0, t[4 * s - 2]); ADD(l1, t[4 * s - 1]);
        }
      } else if (ORDER == 3) { // prescaled: exp directly on the accumulator, adds of the previous group
#pragma unroll
        for (int e = 0; e < 4; ++e) { t[4 * s + e] = cur[4 * s + e]; EXP(t[4 * s + e]); }
        if (s > 0) {
          ADD(l0, t[4 * s - 4]); ADD(l1, t[4 * s - 3]); ADD(l0, t[4 * s - 2]); ADD(l1, t[4 * s - 1]);
        }
      } else if (ORDER == 4) { // all 16 fma, then 16 exp, then 16 add (4 accumulators) per tile, mfma spread
#pragma unroll
        for (int e = 0; e < 4; ++e) FMA(t[4 * s + e], cur[4 * s + e]);
      }
    }
    if (ORDER == 2 || ORDER == 3) { ADD(l0, t[12]); ADD(l1, t[13]); ADD(l0, t[14]); ADD(l1, t[15]); }
    if (ORDER == 4) {
#pragma unroll
      for (int e = 0; e < 16; ++e) EXP(t[e]);
#pragma unroll
      for (int e = 0; e < 16; e += 2) { ADD(l0, t[e]); ADD(l1, t[e + 1]); }
    }
    if (WITH_MFMA) {   // swap roles without copies: next iteration consumes nxt
      asm volatile("" : "+v"(nxt));
      f32x16 tmp = cur; cur = nxt; nxt = tmp;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = l0 + l1 + cur[0];
}

template <int ORDER, bool M>
void run(const char* name, float* out) {
  for (int wps : {1, 2, 3, 4, 8}) {
    const int blocks = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<ORDER, M>), blocks, 256, 0, 0, out); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<ORDER, M>), blocks, 256, 0, 0, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s wps %d : %7.1f ns/tile/SIMD\n", name, wps, ms / 5 * 1e6 / (ITERS * (double)wps));
  }
}

int main() {
  float* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  run<1, false>("O1 fefefefe aaaa (1 acc), no mfma", out);
  run<2, false>("O2 ffff eeee a'a'a'a' (2 acc), no mfma", out);
  run<3, false>("O3 eeee a'a'a'a' (prescaled), no mfma", out);
  run<4, false>("O4 16f 16e 16a (2 acc), no mfma", out);
  run<1, true>("O1 + interleaved mfma", out);
  run<2, true>("O2 + interleaved mfma", out);
  run<3, true>("O3 + interleaved mfma", out);
  run<4, true>("O4 + mfma with the fmas", out);
  return 0;
}
