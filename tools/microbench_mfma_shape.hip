// microbench_mfma_shape.hip — v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16 at the same output tile per wave, on
// RANDOM operands (round 2's version of this file ran on constant operands, which ranks the shapes by cycles and misses the
// clock the chip holds: MI355X_MICROARCH.md "DVFS give-back" item 7; that file was later overwritten — this is its successor
// and the source behind profiles/r02_microbench_mfma_shape.txt's question, re-asked properly in
// profiles/r05_k1_mfma_shape_ab.txt).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_mfma_shape.hip -o /tmp/microbench_mfma_shape && /tmp/microbench_mfma_shape
// Per wave and "item": one 32 x 32 x 64 tile = 4 x 32x32x16 or 8 x 16x16x32 (two k-steps x four sub-tiles), operands in
// registers (8 fragments of random bf16 in [-2, 2)), then EPI x {v_exp_f32 + v_add_f32} on the 16 results (EPI = 0: bare matrix
// loop; EPI = 16: K1's softmax epilogue, exp2 + add per logit).  Reports wall time, shader cycles per item (s_memtime), the
// clock the chip held (s_memtime / s_memrealtime x 100 MHz) and items per second, for 1 and 3 waves per SIMD.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ inline short rnd_bf16(uint32_t& s) {
  s = s * 1664525u + 1013904223u;
  // sign, exponent 126..127 (|x| in [0.5, 2)), 7 random mantissa bits
  return (short)(((s >> 16) & 0x8000u) | ((126u + ((s >> 9) & 1u)) << 7) | ((s >> 20) & 0x7Fu));
}

template <int SHAPE, int EPI>
__global__ __launch_bounds__(256) void loop_kernel(long long* __restrict__ stamps, float* __restrict__ sink, int iters, uint32_t seed) {
  uint32_t s = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) { a[i][e] = rnd_bf16(s); b[i][e] = rnd_bf16(s); }
  float l = 0.f;
  const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  f32x16 acc;
  for (int it = 0; it < iters; ++it) {
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if constexpr (SHAPE == 32) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[k], acc, 0, 0, 0);
    } else {
      f32x4 t[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(u & 1) * 2 + ks], b[(u >> 1) * 2 + ks], t[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) { acc[4 * u] = t[u][0]; acc[4 * u + 1] = t[u][1]; acc[4 * u + 2] = t[u][2]; acc[4 * u + 3] = t[u][3]; }
    }
    if constexpr (EPI > 0) {
#pragma unroll
      for (int i = 0; i < EPI; ++i) l += __builtin_amdgcn_exp2f(acc[i & 15] * 0.125f);
    } else {
      asm volatile("" :: "v"(acc));
    }
    // the next item's operands differ from this one's (one fragment re-drawn: the data stay random, the cost is 8 SALU-free VALU ops per 4 items)
    if ((it & 3) == 3) { a[it & 3][it & 7] = rnd_bf16(s); }
  }
  const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  sink[blockIdx.x * 256 + threadIdx.x] = l + acc[0];
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int SHAPE, int EPI>
int run(const char* name, int wgs_per_cu, long long* d_st, float* d_sink) {
  const int grid = 256 * wgs_per_cu, iters = 40000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int warm = 0; warm < 2; ++warm) loop_kernel<SHAPE, EPI><<<grid, 256>>>(d_st, d_sink, iters, 7u);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int rep = 0; rep < 10; ++rep) loop_kernel<SHAPE, EPI><<<grid, 256>>>(d_st, d_sink, iters, 11u + rep);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> st(2 * grid);
  CK(hipMemcpy(st.data(), d_st, sizeof(long long) * 2 * grid, hipMemcpyDeviceToHost));
  double cyc = 0, ref = 0;
  for (int i = 0; i < grid; ++i) { cyc += (double)st[2 * i]; ref += (double)st[2 * i + 1]; }
  const double items = 10.0 * grid * 4.0 * iters;              // 4 waves per workgroup
  printf("%-44s %d wave(s)/SIMD: %8.3f ms  %6.1f cycles/item/wave  clock %4.0f MHz  %.3e items/s  %.0f TFLOP/s\n", name, wgs_per_cu,
         ms / 10, cyc / grid / iters, 100.0 * cyc / ref, items / (ms * 1e-3), items * 2.0 * 32 * 32 * 64 / (ms * 1e-3) * 1e-12);
  return 0;
}

int main() {
  long long* d_st; float* d_sink;
  CK(hipMalloc(&d_st, sizeof(long long) * 2 * 256 * 4));
  CK(hipMalloc(&d_sink, sizeof(float) * 256 * 4 * 256));
  for (int w : {1, 3}) {
    if (run<32, 0>("32x32x16 x4, bare", w, d_st, d_sink)) return 1;
    if (run<16, 0>("16x16x32 x8, bare", w, d_st, d_sink)) return 1;
    if (run<32, 16>("32x32x16 x4 + 16 (exp2 + add)  [K1's item]", w, d_st, d_sink)) return 1;
    if (run<16, 16>("16x16x32 x8 + 16 (exp2 + add)  [K1's item]", w, d_st, d_sink)) return 1;
  }
  return 0;
}
