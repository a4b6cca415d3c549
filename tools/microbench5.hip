// microbench5.hip — issue cost of the ops in the K1 epilogue, one kind at a time (inline asm so the
// compiler cannot fuse or drop them): v_max_f32, v_max3_f32, v_add_f32, v_cmp+v_cndmask, v_exp_f32.
// hipcc --offload-arch=gfx950 -O3 tools/microbench5.hip -o /tmp/mb5 && /tmp/mb5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ITERS = 2048;

#define KERNEL(NAME, BODY)                                                   \
  __global__ void NAME(float* out) {                                          \
    float x[8], y[8];                                                         \
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = i * 0.37f - threadIdx.x * 0.002f; } \
    for (int it = 0; it < ITERS; ++it) {                                      \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) { BODY }                  \
    }                                                                         \
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + y[i];                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                           \
  }

KERNEL(k_max2, asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));)
KERNEL(k_max3, asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 7]));)
KERNEL(k_add, asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));)
KERNEL(k_fma, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 7]));)
KERNEL(k_fmac, asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 7]));)
KERNEL(k_exp, asm volatile("v_exp_f32 %0, %1" : "=v"(x[i]) : "v"(y[i]));)
KERNEL(k_cmpsel, asm volatile("v_cmp_gt_f32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(y[i]) : "vcc");)
KERNEL(k_expadd, asm volatile("v_exp_f32 %1, %1\n v_add_f32 %0, %0, %1" : "+v"(x[i]), "+v"(y[i]));)
KERNEL(k_min3, asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 7]));)
KERNEL(k_maxu, asm volatile("v_max_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));)
KERNEL(k_max3u, asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 7]));)
KERNEL(k_max3i, asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 7]));)
KERNEL(k_cmpu, asm volatile("v_cmp_gt_u32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(y[i]) : "vcc");)
KERNEL(k_med3, asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 7]));)
KERNEL(k_pkadd, asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(reinterpret_cast<double*>(x) + (i & 3))) : "v"(*(reinterpret_cast<double*>(y) + (i & 3))));)

template <typename K>
void run(const char* name, K kern, int ops_per_body) {
  float* out;
  hipMalloc(&out, 256 * 4 * 256 * 16 * sizeof(float));
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = p.multiProcessorCount * wps;   // 256 threads = 1 wave per SIMD per block
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, 256>>>(out);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) kern<<<blocks, 256>>>(out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double inst_per_simd = (double)ITERS * 8 * ops_per_body * wps;
    printf("%-12s wps %d: %.2f cyc/inst/SIMD @2.4GHz\n", name, wps, ms * 1e-3 * 2.4e9 / inst_per_simd);
  }
  hipFree(out);
}

int main() {
  run("v_max_f32", k_max2, 1);
  run("v_max3_f32", k_max3, 1);
  run("v_min3_f32", k_min3, 1);
  run("v_add_f32", k_add, 1);
  run("v_fma_f32", k_fma, 1);
  run("v_fmac_f32", k_fmac, 1);
  run("v_exp_f32", k_exp, 1);
  run("cmp+cndmask", k_cmpsel, 2);
  run("exp+add", k_expadd, 2);
  run("v_pk_add_f32", k_pkadd, 1);
  run("v_max_u32", k_maxu, 1);
  run("v_max3_u32", k_max3u, 1);
  run("v_max3_i32", k_max3i, 1);
  run("cmp_u32+cnd", k_cmpu, 2);
  run("v_med3_f32", k_med3, 1);
  return 0;
}
