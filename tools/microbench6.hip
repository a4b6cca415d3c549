// microbench6.hip — what does the MFMA work of one K1 item (32 keys x 32 queries x D=64 per wave) cost beside
// the VALU-bound epilogue when it is issued as 4 x v_mfma_f32_32x32x16_bf16 (16-register accumulator
// chain) versus 8 x v_mfma_f32_16x16x32_bf16 (four 4-register tiles x 2 k-steps)?  Epilogue = the direct
// kernel's per-item VALU stream: 16 exp2, 16 add, 7 max3 + max, med3, cmp+cndmask, max.
// hipcc --offload-arch=gfx950 -O3 tools/microbench6.hip -o /tmp/mb6 && /tmp/mb6
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;
constexpr int ITERS = 2048;
#define EXP(d, s) asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(s))
#define ADD(acc, s) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(s))
#define MAX3(d, a, b, c) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c))

// MODE 0: no MFMA; 1: 4 x 32x32x16 chain; 2: 8 x 16x16x32 (4 tiles x 2 k-steps); 3: 8 x 32x32x16 (D = 128);
// 4: 16 x 16x16x32 (D = 128)
template <int MODE>
__global__ void k(float* out) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3c00 + (threadIdx.x & 7)); b[i] = (short)(0x3c00 + i); }
  f32x16 cur, nxt;
  f32x4 n4[4];
  for (int i = 0; i < 16; ++i) cur[i] = -0.01f * (i + (threadIdx.x & 3));
  nxt = cur;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) n4[j][i] = cur[4 * j + i];
  float l = 0.f, m = -1e30f, m2 = -1e30f;
  int tb = 0;
  float t[16], x0, x1, x2, x3, x4, y0, y1, tm;
  for (int it = 0; it < ITERS; ++it) {
    constexpr int NM = MODE == 0 ? 0 : MODE == 1 ? 4 : MODE == 2 ? 8 : MODE == 3 ? 8 : 16;
    constexpr int G = 4;   // epilogue is cut into 4 groups interleaved with the MFMAs
#pragma unroll
    for (int s = 0; s < G; ++s) {
#pragma unroll
      for (int q = 0; q < NM / G; ++q) {
        const int j = s * (NM / G) + q;
        if (MODE == 1 || MODE == 3) {
          if (j == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(nxt) : "v"(a), "v"(b));
          else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(nxt) : "v"(a), "v"(b));
        } else if (MODE == 2 || MODE == 4) {
          constexpr int KS = MODE == 2 ? 2 : 4;
          const int tile = j / KS, ks = j % KS;
          if (ks == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(n4[tile]) : "v"(a), "v"(b));
          else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(n4[tile]) : "v"(a), "v"(b));
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { EXP(t[4 * s + e], cur[4 * s + e]); ADD(l, t[4 * s + e]); }
      if (s == 0) { MAX3(x0, cur[0], cur[1], cur[2]); MAX3(x1, cur[3], cur[4], cur[5]); }
      if (s == 1) { MAX3(x2, cur[6], cur[7], cur[8]); MAX3(x3, cur[9], cur[10], cur[11]); }
      if (s == 2) { MAX3(x4, cur[12], cur[13], cur[14]); MAX3(y0, x0, x1, x2); MAX3(y1, x3, x4, cur[15]); }
      if (s == 3) {
        asm volatile("v_max_f32 %0, %1, %2" : "=v"(tm) : "v"(y0), "v"(y1));
        asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(m2) : "v"(m), "v"(tm));
        asm volatile("v_cmp_gt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(tb) : "v"(tm), "v"(m), "v"(it) : "vcc");
        asm volatile("v_max_f32 %0, %0, %1" : "+v"(m) : "v"(tm));
      }
    }
    if (MODE == 1 || MODE == 3) { asm volatile("" : "+v"(nxt)); f32x16 tmp = cur; cur = nxt; nxt = tmp; }
    if (MODE == 2 || MODE == 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(n4[j])); for (int i = 0; i < 4; ++i) { const float v = cur[4 * j + i]; cur[4 * j + i] = n4[j][i]; n4[j][i] = v; } }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = l + m + m2 + tb + cur[0];
}

template <int MODE>
void run(const char* name, float* out) {
  for (int wps : {1, 2, 3, 4}) {
    const int blocks = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), blocks, 256, 0, 0, out); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE>), blocks, 256, 0, 0, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s wps %d : %7.1f ns/item/SIMD\n", name, wps, ms / 5 * 1e6 / (ITERS * (double)wps));
  }
}
int main() {
  float* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  run<0>("epilogue only (44 VALU)", out);
  run<1>("epilogue + 4 x mfma_32x32x16 (D=64)", out);
  run<2>("epilogue + 8 x mfma_16x16x32 (D=64, 4 tiles)", out);
  run<3>("epilogue + 8 x mfma_32x32x16 (D=128)", out);
  run<4>("epilogue + 16 x mfma_16x16x32 (D=128, 4 tiles)", out);
  return 0;
}
