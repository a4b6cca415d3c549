// microbench4.hip — does an MFMA with its accumulator in AGPRs overlap better with a VALU-bound
// stream than the VGPR-form MFMA?  Same pinned epilogue order as microbench3 O2.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;
constexpr int ITERS = 2048;
#define FMA(d, s) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(s), "v"(c1), "v"(nm))
#define EXP(d) asm volatile("v_exp_f32 %0, %0" : "+v"(d))
#define ADD(acc, s) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(s))

template <int FORM, bool WITH_MFMA>   // FORM 0: VGPR accumulator, 1: AGPR accumulator
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
      if (WITH_MFMA) {
        if (FORM == 0) {
          if (s == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(nxt) : "v"(a), "v"(b));
          else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(nxt) : "v"(a), "v"(b));
        } else {
          if (s == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(nxt) : "v"(a), "v"(b));
          else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(nxt) : "v"(a), "v"(b));
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) FMA(t[4 * s + e], cur[4 * s + e]);
#pragma unroll
      for (int e = 0; e < 4; ++e) EXP(t[4 * s + e]);
      if (s > 0) { ADD(l0, t[4 * s - 4]); ADD(l1, t[4 * s - 3]); ADD(l0, t[4 * s - 2]); ADD(l1, t[4 * s - 1]); }
    }
    ADD(l0, t[12]); ADD(l1, t[13]); ADD(l0, t[14]); ADD(l1, t[15]);
    if (WITH_MFMA) {
      // consume: next iteration reads the fresh accumulator (AGPR form pays 16 v_accvgpr_read here)
      if (FORM == 1) { asm volatile("s_nop 7"); for (int i = 0; i < 16; ++i) { float r; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(nxt[i])); cur[i] = r; } }
      else { asm volatile("" : "+v"(nxt)); f32x16 tmp = cur; cur = nxt; nxt = tmp; }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = l0 + l1 + cur[0];
}

template <int FORM, bool M>
void run(const char* name, float* out) {
  for (int wps : {1, 2, 3, 4}) {
    const int blocks = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<FORM, M>), blocks, 256, 0, 0, out); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<FORM, M>), blocks, 256, 0, 0, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s wps %d : %7.1f ns/tile/SIMD\n", name, wps, ms / 5 * 1e6 / (ITERS * (double)wps));
  }
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  run<0, false>("epilogue only", out);
  run<0, true>("epilogue + 4 MFMA, VGPR accumulator", out);
  run<1, true>("epilogue + 4 MFMA, AGPR accumulator (+16 reads)", out);
  return 0;
}
