// mfma_scale_probe.hip — what v_mfma_scale_f32_32x32x64_f8f6f4 does with the registers it is given (gfx950).
// The A/B lane maps of the block-scaled instruction are not in the guides ("check the map with exact integer data"), so this
// program runs ONE instruction per test on register images supplied by tools/mfma_scale_probe.py, which builds them under a
// layout hypothesis and compares D with numpy.  Second part: cycles per instruction (s_memtime) for fp8 / fp6 / fp4 operands.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_scale_probe.hip -o /tmp/mfma_scale_probe
//   /tmp/mfma_scale_probe in.bin out.bin
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

using i32x8 = __attribute__((ext_vector_type(8))) int;
using f32x16 = __attribute__((ext_vector_type(16))) float;

struct Test {
  int fmt;                 // 0 fp8 e4m3, 2 fp6 e2m3, 4 fp4 e2m1 (both operands)
  uint32_t a[64][8], b[64][8], sa[64], sb[64];
};

template <int FMT>
__device__ f32x16 one(const i32x8& a, const i32x8& b, int sa, int sb) {
  f32x16 c;
#pragma unroll
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, FMT, FMT, 0, sa, 0, sb);
}

__global__ void probe_kernel(const Test* __restrict__ tests, int n, float* __restrict__ out) {
  const int lane = threadIdx.x;
  for (int t = 0; t < n; ++t) {
    i32x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (int)tests[t].a[lane][i]; b[i] = (int)tests[t].b[lane][i]; }
    const int sa = (int)tests[t].sa[lane], sb = (int)tests[t].sb[lane];
    f32x16 d;
    if (tests[t].fmt == 0) d = one<0>(a, b, sa, sb);
    else if (tests[t].fmt == 2) d = one<2>(a, b, sa, sb);
    else d = one<4>(a, b, sa, sb);
#pragma unroll
    for (int i = 0; i < 16; ++i) out[((size_t)t * 64 + lane) * 16 + i] = d[i];
  }
}

// cycles per instruction: NACC independent accumulators, ITER rounds, one wave per SIMD (grid = CUs, block = 256)
template <int FMT, int NACC>
__global__ __launch_bounds__(256) void rate_kernel(long long* __restrict__ cyc, float* __restrict__ sink, int iters, uint32_t seed) {
  i32x8 a, b;
  uint32_t s = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    s = s * 1664525u + 1013904223u; a[i] = (int)(s & 0x77777777u);      // random codes of moderate size in every format
    s = s * 1664525u + 1013904223u; b[i] = (int)(s & 0x77777777u);
  }
  f32x16 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], FMT, FMT, 0, 127, 0, 127);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc[j][i];
  sink[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

using bf16x8 = __attribute__((ext_vector_type(8))) short;
template <int NACC>
__global__ __launch_bounds__(256) void rate_bf16_kernel(long long* __restrict__ cyc, float* __restrict__ sink, int iters, uint32_t seed) {
  bf16x8 a, b;
  uint32_t s = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    s = s * 1664525u + 1013904223u; a[i] = (short)(0x3F00 | (s >> 25) | ((s >> 8) & 0x8000));
    s = s * 1664525u + 1013904223u; b[i] = (short)(0x3F00 | (s >> 25) | ((s >> 8) & 0x8000));
  }
  f32x16 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc[j][i];
  sink[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc >= 3) {
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int n = 0;
    if (fread(&n, 4, 1, f) != 1) return 1;
    std::vector<Test> tests(n);
    if (fread(tests.data(), sizeof(Test), n, f) != (size_t)n) { fprintf(stderr, "short input\n"); return 1; }
    fclose(f);
    Test* d_t; float* d_o;
    CK(hipMalloc(&d_t, sizeof(Test) * n));
    CK(hipMalloc(&d_o, sizeof(float) * 64 * 16 * n));
    CK(hipMemcpy(d_t, tests.data(), sizeof(Test) * n, hipMemcpyHostToDevice));
    probe_kernel<<<1, 64>>>(d_t, n, d_o);
    CK(hipDeviceSynchronize());
    std::vector<float> out((size_t)n * 64 * 16);
    CK(hipMemcpy(out.data(), d_o, out.size() * 4, hipMemcpyDeviceToHost));
    f = fopen(argv[2], "wb");
    fwrite(out.data(), 4, out.size(), f);
    fclose(f);
    printf("probe: %d tests written\n", n);
  }
  // rates
  long long* d_c; float* d_s;
  const int grid = 256, iters = 20000;
  CK(hipMalloc(&d_c, 8 * grid));
  CK(hipMalloc(&d_s, 4 * grid * 256));
  std::vector<long long> c(grid);
  auto report = [&](const char* name, int nacc) {
    hipDeviceSynchronize();
    hipMemcpy(c.data(), d_c, 8 * grid, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : c) s += (double)v;
    printf("%-28s %d accumulators: %.2f cycles per instruction per SIMD\n", name, nacc, s / grid / ((double)iters * nacc));
  };
  for (int rep = 0; rep < 2; ++rep) {
    rate_kernel<0, 4><<<grid, 256>>>(d_c, d_s, iters, 1u); report("scale 32x32x64 fp8", 4);
    rate_kernel<2, 4><<<grid, 256>>>(d_c, d_s, iters, 2u); report("scale 32x32x64 fp6", 4);
    rate_kernel<4, 4><<<grid, 256>>>(d_c, d_s, iters, 3u); report("scale 32x32x64 fp4", 4);
    rate_kernel<4, 1><<<grid, 256>>>(d_c, d_s, iters, 3u); report("scale 32x32x64 fp4", 1);
    rate_bf16_kernel<4><<<grid, 256>>>(d_c, d_s, iters, 4u); report("32x32x16 bf16", 4);
  }
  return 0;
}
