// microbench.hip — issue-rate probes that size the K1 (exp vs MFMA) and K3 (VALU) kernels.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f32x2 = __attribute__((ext_vector_type(2))) float;

constexpr int ITERS = 4096;

__global__ void k_fma(float* out, float a, float b) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], a, b);
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_pkfma(float* out, float a, float b) {
  f32x2 x[8]; f32x2 av = {a, a}, bv = {b, b};
  for (int i = 0; i < 8; ++i) x[i] = f32x2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(av), "v"(bv));
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_exp(float* out, float a) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i * 0.1f;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]) * 0.0f + x[i];  // exp + fma
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_exponly(float* out, float a) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = -(threadIdx.x * 0.001f + i * 0.1f);
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mfma(float* out) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x); b[i] = (short)(0x3f00 + i); }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int it = 0; it < ITERS / 4; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
  }
  float s = 0; for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// MFMA + softmax-like VALU on the previous tile: 4 MFMA (one 32x32 tile at D=64) then
// NEXP exps + NFMA fmas on 16 values.
template <int NEXP, int NOTHER>
__global__ void k_mix(float* out) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3c00 + threadIdx.x); b[i] = (short)(0x3c00 + i); }
  f32x16 c = {0};
  float l = 0.f, m = 1.0f;
  for (int it = 0; it < ITERS / 4; ++it) {
    f32x16 z = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, z, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NEXP; ++i) l += __builtin_amdgcn_exp2f(__builtin_fmaf(c[i], 1.44f, -m));
#pragma unroll
    for (int i = 0; i < NOTHER; ++i) m = __builtin_fmaxf(m, c[i & 15] * 0.5f);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = l + m;
}

template <typename F>
double timeit(F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5.0 * 1e-3;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  float* out; hipMalloc(&out, 256 * 8 * 4 * 256 * 4 * sizeof(float));
  for (int wps : {1, 2, 4, 8}) {  // waves per SIMD
    const int blocks = 256 * wps, threads = 256;
    const double nw = (double)blocks * 4;  // waves
    double t;
    t = timeit([&] { hipLaunchKernelGGL(k_fma, blocks, threads, 0, 0, out, 1.0001f, 0.5f); });
    printf("wps %d  v_fma_f32     : %.2f cyc/inst/SIMD @2.4GHz  (%.1f TFLOP/s)\n", wps, t * 2.4e9 / (ITERS * 8.0 * wps), nw * 64 * ITERS * 8 * 2 / t * 1e-12);
    t = timeit([&] { hipLaunchKernelGGL(k_pkfma, blocks, threads, 0, 0, out, 1.0001f, 0.5f); });
    printf("wps %d  v_pk_fma_f32  : %.2f cyc/inst/SIMD  (%.1f TFLOP/s)\n", wps, t * 2.4e9 / (ITERS * 8.0 * wps), nw * 64 * ITERS * 8 * 4 / t * 1e-12);
    t = timeit([&] { hipLaunchKernelGGL(k_exponly, blocks, threads, 0, 0, out, 1.0f); });
    printf("wps %d  v_exp_f32     : %.2f cyc/inst/SIMD  (%.2f Texp/s)\n", wps, t * 2.4e9 / (ITERS * 8.0 * wps), nw * 64 * ITERS * 8 / t * 1e-12);
    t = timeit([&] { hipLaunchKernelGGL(k_exp, blocks, threads, 0, 0, out, 1.0f); });
    printf("wps %d  exp+fma pair  : %.2f cyc/pair/SIMD\n", wps, t * 2.4e9 / (ITERS * 8.0 * wps));
    t = timeit([&] { hipLaunchKernelGGL(k_mfma, blocks, threads, 0, 0, out); });
    printf("wps %d  mfma32x32x16  : %.2f cyc/inst/SIMD  (%.1f TFLOP/s)\n", wps, t * 2.4e9 / (ITERS * 1.0 * wps), nw * ITERS * 32768.0 / t * 1e-12);
    t = timeit([&] { hipLaunchKernelGGL((k_mix<16, 8>), blocks, threads, 0, 0, out); });
    printf("wps %d  4mfma+16exp+16fma+8max : %.1f cyc/tile/SIMD (mfma-only floor 128)\n", wps, t * 2.4e9 / (ITERS / 4.0 * wps));
    t = timeit([&] { hipLaunchKernelGGL((k_mix<0, 8>), blocks, threads, 0, 0, out); });
    printf("wps %d  4mfma+8max             : %.1f cyc/tile/SIMD\n", wps, t * 2.4e9 / (ITERS / 4.0 * wps));
    t = timeit([&] { hipLaunchKernelGGL((k_mix<8, 8>), blocks, threads, 0, 0, out); });
    printf("wps %d  4mfma+8exp+8fma+8max   : %.1f cyc/tile/SIMD\n", wps, t * 2.4e9 / (ITERS / 4.0 * wps));
  }
  return 0;
}
