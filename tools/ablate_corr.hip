// ablate_corr.hip — timing-only ablation harness for K1 (cdna guide §7 "Ablate"): builds
// corr_argmax.hip with one ISR_ABL_* switch and times isr_corr_argmax on random bf16 data.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form \
//         -DISR_ABL_NOEXP tools/ablate_corr.hip -o /tmp/abl && /tmp/abl
#include "../imagesequenceregistrationfor6dposeestimationlabeling_amd/csrc/capi_common.hip"
#include "../imagesequenceregistrationfor6dposeestimationlabeling_amd/csrc/corr_argmax.hip"
#include <vector>
int main(int argc, char** argv) {
  const int P = argc > 1 ? atoi(argv[1]) : 1048576, N = 20000, D = 64;
  std::vector<uint16_t> q((size_t)P * D), k((size_t)N * D);
  srand(1);
  for (auto& x : q) x = 0x3c00 + (rand() & 0x1ff) + ((rand() & 1) << 15);
  for (auto& x : k) x = 0x3c00 + (rand() & 0x1ff) + ((rand() & 1) << 15);
  uint16_t *dq, *dk; int32_t* idx; float* lp; void* ws;
  hipMalloc(&dq, q.size() * 2); hipMalloc(&dk, k.size() * 2); hipMalloc(&idx, P * 4); hipMalloc(&lp, P * 4);
  const size_t wsb = isr_corr_argmax_workspace_bytes(P, N, D, 0);
  hipMalloc(&ws, wsb);
  hipMemcpy(dq, q.data(), q.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dk, k.data(), k.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  isr_corr_argmax(dq, dk, P, N, D, D, D, 0, idx, lp, nullptr, ws, wsb, nullptr);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) isr_corr_argmax(dq, dk, P, N, D, D, D, 0, idx, lp, nullptr, ws, wsb, nullptr);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double tiles = (double)P / 32 * (N / 32.0) / 1024;
  printf("%.3f ms  %.1f ns/tile/SIMD\n", ms / 5, ms / 5 * 1e6 / tiles);
  return 0;
}
