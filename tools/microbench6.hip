// microbench6.hip — issue cost of the f64 ops in the sampler's weight (estimate_pose), one kind at a time (inline asm so the
// compiler cannot fuse or drop them).   hipcc --offload-arch=gfx950 -O3 tools/microbench6.hip -o /tmp/mb6 && /tmp/mb6
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int ITERS = 2048;

#define KERNEL(NAME, BODY)                                                   \
  __global__ void NAME(double* out) {                                         \
    double x[8], y[8];                                                        \
    float f[8]; int n[8];                                                     \
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001 + i; y[i] = i * 0.37 - threadIdx.x * 0.002; f[i] = (float)y[i]; n[i] = i - 3; } \
    for (int it = 0; it < ITERS; ++it) {                                      \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) { BODY }                  \
    }                                                                         \
    double s = 0; for (int i = 0; i < 8; ++i) s += x[i] + y[i] + f[i] + n[i]; \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                           \
  }

KERNEL(k_fma, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 7]));)
KERNEL(k_fmas, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "s"(0.123456789));)
KERNEL(k_add, asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));)
KERNEL(k_mul, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));)
KERNEL(k_max, asm volatile("v_max_f64 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));)
KERNEL(k_rndne, asm volatile("v_rndne_f64 %0, %1" : "=v"(x[i]) : "v"(y[i]));)
KERNEL(k_cvt_f64_f32, asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x[i]) : "v"(f[i]));)
KERNEL(k_cvt_i32_f64, asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n[i]) : "v"(y[i]));)
KERNEL(k_ldexp, asm volatile("v_ldexp_f64 %0, %1, %2" : "=v"(x[i]) : "v"(y[i]), "v"(n[i]));)
KERNEL(k_mov64, asm volatile("v_mov_b64 %0, %1" : "=v"(x[i]) : "v"(y[i]));)
KERNEL(k_lshladd, asm volatile("v_lshl_add_u32 %0, %1, 20, %2" : "=v"(n[i]) : "v"(n[(i + 1) & 7]), "v"(n[(i + 2) & 7]));)
KERNEL(k_cmpsel, asm volatile("v_cmp_gt_f64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(n[i]) : "v"(x[i]), "v"(y[i]), "v"(n[(i + 1) & 7]) : "vcc");)

template <typename K>
void run(const char* name, K kern, int ops_per_body) {
  double* out;
  hipMalloc(&out, 256 * 4 * 256 * 16 * sizeof(double));
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
    printf("%-14s wps %d: %.2f cyc/inst/SIMD @2.4GHz\n", name, wps, ms * 1e-3 * 2.4e9 / inst_per_simd);
  }
  hipFree(out);
}

int main() {
  run("v_fma_f64", k_fma, 1);
  run("v_fma_f64 sgpr", k_fmas, 1);
  run("v_add_f64", k_add, 1);
  run("v_mul_f64", k_mul, 1);
  run("v_max_f64", k_max, 1);
  run("v_rndne_f64", k_rndne, 1);
  run("v_cvt_f64_f32", k_cvt_f64_f32, 1);
  run("v_cvt_i32_f64", k_cvt_i32_f64, 1);
  run("v_ldexp_f64", k_ldexp, 1);
  run("v_mov_b64", k_mov64, 1);
  run("v_lshl_add_u32", k_lshladd, 1);
  run("cmp_f64+cnd", k_cmpsel, 2);
  return 0;
}
