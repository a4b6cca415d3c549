// corr_logsoftmax.hip — K1, materialising variant for small P: out[p][n] = log_softmax_n <Q[p], K[n]>.
//
// Replaces  corr_matrix_log = torch.log_softmax(queries @ obj_keys.T, dim=1)      poseEstSurf.py:70
//           and getCors(leaves > 1), whose caller then runs topk on the matrix    inference.py:143-145
// The estimate_pose path keeps the (n x m) matrix resident (it max-pools and samples it), so this
// variant writes it: one workgroup per query row, three sweeps over the keys (max, sum, write) with
// the logit recomputed each time as a k-ordered f32 fmaf chain (bf16 inputs are widened exactly).
// HBM-bound on the P*N*4-byte output; the keys (N*D elements) stay L2-resident.
#include "isr_common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxD = 256;

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }

template <typename T>
__global__ __launch_bounds__(kThreads) void logsoftmax_rows_kernel(const T* __restrict__ Q,
                                                                   const T* __restrict__ K, int N, int D,
                                                                   int ldq, int ldk, float* __restrict__ out,
                                                                   int64_t ldo) {
  __shared__ float q[kMaxD];
  __shared__ float red[kThreads / 64];
  __shared__ float bcast;
  const int p = blockIdx.x;
  for (int d = threadIdx.x; d < D; d += kThreads) q[d] = to_f32(Q[(size_t)p * ldq + d]);
  __syncthreads();
  auto logit = [&](int n) {
    const T* k = K + (size_t)n * ldk;
    float acc = 0.f;
    for (int d = 0; d < D; ++d) acc = __builtin_fmaf(q[d], to_f32(k[d]), acc);
    return acc;
  };
  auto block_reduce = [&](float v, bool is_max) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float o = __shfl_down(v, off, 64);
      v = is_max ? fmaxf(v, o) : v + o;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0)
      bcast = is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : ((red[0] + red[1]) + red[2]) + red[3];
    __syncthreads();
    return bcast;
  };
  float m = -__builtin_inff();
  for (int n = threadIdx.x; n < N; n += kThreads) m = fmaxf(m, logit(n));
  m = block_reduce(m, true);
  float s = 0.f;
  for (int n = threadIdx.x; n < N; n += kThreads) s += __expf(logit(n) - m);
  s = block_reduce(s, false);
  const float lse = m + __logf(s);
  float* o = out + (size_t)p * ldo;
  for (int n = threadIdx.x; n < N; n += kThreads) o[n] = logit(n) - lse;
}

}  // namespace

extern "C" int isr_corr_logsoftmax(const void* Q, const void* K, int P, int N, int D, int ldq, int ldk,
                                   int dtype, float* out, int64_t ldo, isr_stream_t stream_) {
  ISR_REQUIRE(Q && K && out, "isr_corr_logsoftmax: null pointer");
  ISR_REQUIRE(P > 0 && N > 0 && D > 0 && D <= kMaxD, "isr_corr_logsoftmax: P=%d N=%d D=%d (D <= %d)", P, N, D, kMaxD);
  ISR_REQUIRE(ldq >= D && ldk >= D && ldo >= N, "isr_corr_logsoftmax: leading dimensions too small");
  hipStream_t stream = isr::as_stream(stream_);
  if (dtype == ISR_DTYPE_BF16)
    logsoftmax_rows_kernel<uint16_t><<<P, kThreads, 0, stream>>>(static_cast<const uint16_t*>(Q),
                                                                 static_cast<const uint16_t*>(K), N, D, ldq, ldk, out, ldo);
  else if (dtype == ISR_DTYPE_F32)
    logsoftmax_rows_kernel<float><<<P, kThreads, 0, stream>>>(static_cast<const float*>(Q),
                                                              static_cast<const float*>(K), N, D, ldq, ldk, out, ldo);
  else {
    isr::set_error("isr_corr_logsoftmax: dtype %d", dtype);
    return ISR_ERR_ARG;
  }
  ISR_CHECK_LAUNCH("logsoftmax_rows_kernel");
  return ISR_OK;
}
