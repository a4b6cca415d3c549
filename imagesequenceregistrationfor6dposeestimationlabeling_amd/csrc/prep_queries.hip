// prep_queries.hip — the hand-off between the descriptor network and K1.
//
// Replaces, on the device and without a host round trip (inference.py:248-279):
//     imfeats   = imfeatsfull[..., 0:12][:, ::3, ::3]          channels-last feature map, every 3rd pixel
//     inputMask = cropMask[:, :, 0][::3, ::3]
//     maskIds   = torch.where(inputMask)                        row-major order
//     maskedfeats = imfeats[0][maskIds]                         (P, 12)
//     ep2d[:, 0] = maskIds[1]; ep2d[:, 1] = maskIds[0]          (col, row) in the subsampled grid
// and hands K1 its operands in the layout it wants: rows padded to the MFMA depth (zero columns),
// rounded ONCE to bf16 — after the log2(e) prescale when the log2 domain is asked for — or kept f32.
// The count of masked pixels stays on the device (n_dev): K1 runs over the capacity rows (rows past
// the count are zero queries), isr_select_top_dev / isr_gather_corr / isr_pnp_ransac take the
// count from the device.
// Ordered stream compaction: per-block counts -> one-block scan -> scatter (as select_top.hip).
#include "isr_common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr float kLog2e = 1.4426950408889634f;

__device__ __forceinline__ uint16_t bf16_rne(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

__device__ __forceinline__ int block_exclusive_scan(int v, int* total) {
  __shared__ int wsum[kThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(inc, off, 64);
    if (lane >= off) inc += o;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kThreads / 64; ++w) {
    if (w < wave) base += wsum[w];
    tot += wsum[w];
  }
  *total = tot;
  return base + inc - v;
}

// pixel s of the subsampled grid (row-major): is it inside the mask?
__device__ __forceinline__ bool masked(const uint8_t* mask, int mask_pix_stride, int W, int Ws, int step, int s) {
  const int r = s / Ws, c = s % Ws;
  return mask[((size_t)(r * step) * W + (size_t)c * step) * mask_pix_stride] != 0;
}

// Every kernel carries the image on blockIdx.z (strides in elements): a group of crops costs the same three
// launches as one; the single-image entry point is the B = 1 case of the same kernels.
__global__ __launch_bounds__(kThreads) void prep_count_kernel(const uint8_t* __restrict__ mask, int mask_pix_stride,
                                                              size_t mask_img_stride, int W, int Ws, int step, int S,
                                                              int32_t* __restrict__ block_counts) {
  mask += blockIdx.z * mask_img_stride;
  block_counts += (size_t)blockIdx.z * gridDim.x;
  const int s = blockIdx.x * kThreads + threadIdx.x;
  const int f = (s < S) && masked(mask, mask_pix_stride, W, Ws, step, s);
  int total;
  (void)block_exclusive_scan(f, &total);
  if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}

__global__ void prep_scan_kernel(int32_t* __restrict__ block_counts, int nblocks, int32_t* __restrict__ n_dev) {
  __shared__ int32_t tsum[1024];
  block_counts += (size_t)blockIdx.z * nblocks;
  n_dev += blockIdx.z;
  const int t = threadIdx.x;
  const int per = (nblocks + 1023) / 1024;
  int32_t s = 0;
  for (int j = 0; j < per; ++j) {
    const int b = t * per + j;
    if (b < nblocks) s += block_counts[b];
  }
  tsum[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int32_t v = (t >= off) ? tsum[t - off] : 0;
    __syncthreads();
    tsum[t] += v;
    __syncthreads();
  }
  int32_t run = (t > 0) ? tsum[t - 1] : 0;
  for (int j = 0; j < per; ++j) {
    const int b = t * per + j;
    if (b < nblocks) {
      const int32_t c = block_counts[b];
      block_counts[b] = run;
      run += c;
    }
  }
  if (t == 1023) *n_dev = tsum[1023];
}

template <int DTYPE>
__global__ __launch_bounds__(kThreads) void prep_scatter_kernel(
    const float* __restrict__ feat, size_t feat_img_stride, int W, int C, int c0, int D, const uint8_t* __restrict__ mask,
    int mask_pix_stride, size_t mask_img_stride, int Ws, int step, int S, const int32_t* __restrict__ block_off, int ldq,
    void* __restrict__ Q, float* __restrict__ pix_xy) {
  const size_t img = blockIdx.z;
  feat += img * feat_img_stride;
  mask += img * mask_img_stride;
  block_off += img * gridDim.x;
  Q = static_cast<char*>(Q) + img * (size_t)S * ldq * (DTYPE == ISR_DTYPE_F32 ? 4 : 2);
  pix_xy += img * (size_t)S * 2;
  const int s = blockIdx.x * kThreads + threadIdx.x;
  const int f = (s < S) && masked(mask, mask_pix_stride, W, Ws, step, s);
  int total;
  const int o = block_off[blockIdx.x] + block_exclusive_scan(f, &total);
  if (!f) return;
  const int r = s / Ws, c = s % Ws;
  const float* src = feat + ((size_t)(r * step) * W + (size_t)c * step) * C + c0;
  if (DTYPE == ISR_DTYPE_F32) {
    float* dst = static_cast<float*>(Q) + (size_t)o * ldq;
    for (int d = 0; d < D; ++d) dst[d] = src[d];
  } else {
    uint16_t* dst = static_cast<uint16_t*>(Q) + (size_t)o * ldq;
    for (int d = 0; d < D; ++d) dst[d] = bf16_rne(DTYPE == ISR_DTYPE_BF16_LOG2 ? src[d] * kLog2e : src[d]);
  }
  pix_xy[2 * (size_t)o] = (float)c;        // ep2d[:, 0] = maskIds[1]
  pix_xy[2 * (size_t)o + 1] = (float)r;    // ep2d[:, 1] = maskIds[0]
}

}  // namespace

static size_t prep_ws_bytes(int H, int W, int step, int B) {
  const size_t S = (size_t)((H + step - 1) / step) * ((W + step - 1) / step);
  return isr::align_up(((S + kThreads - 1) / kThreads) * 4 * (size_t)B, 256) + 256;
}

extern "C" size_t isr_prep_queries_workspace_bytes(int H, int W, int step) {
  if (H <= 0 || W <= 0 || step <= 0) return 0;
  return prep_ws_bytes(H, W, step, 1);
}

extern "C" size_t isr_prep_queries_batch_workspace_bytes(int H, int W, int step, int B) {
  if (H <= 0 || W <= 0 || step <= 0 || B <= 0) return 0;
  return prep_ws_bytes(H, W, step, B);
}

extern "C" int isr_prep_queries_batch(const float* feat, int B, int H, int W, int C, int c0, int D, const uint8_t* mask,
                                      int mask_pix_stride, int step, int dtype, int ldq, void* Q, float* pix_xy,
                                      int32_t* n_dev, void* ws, size_t ws_bytes, isr_stream_t stream_) {
  ISR_REQUIRE(feat && mask && Q && pix_xy && n_dev, "isr_prep_queries: null pointer");
  ISR_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0 && C > 0 && D > 0 && c0 >= 0 && c0 + D <= C && step > 0 && mask_pix_stride > 0,
              "isr_prep_queries: B=%d H=%d W=%d C=%d c0=%d D=%d step=%d mask stride=%d", B, H, W, C, c0, D, step,
              mask_pix_stride);
  ISR_REQUIRE(ldq >= D, "isr_prep_queries: ldq=%d < D=%d", ldq, D);
  ISR_REQUIRE(dtype == ISR_DTYPE_BF16 || dtype == ISR_DTYPE_F32 || dtype == ISR_DTYPE_BF16_LOG2,
              "isr_prep_queries: dtype %d", dtype);
  if (!ws || ws_bytes < prep_ws_bytes(H, W, step, B)) {
    isr::set_error("isr_prep_queries: workspace %zu < %zu", ws_bytes, prep_ws_bytes(H, W, step, B));
    return ISR_ERR_WORKSPACE;
  }
  hipStream_t stream = isr::as_stream(stream_);
  const int Hs = (H + step - 1) / step, Ws = (W + step - 1) / step;
  const int S = Hs * Ws;
  const int nblocks = (S + kThreads - 1) / kThreads;
  isr::Workspace w(ws, ws_bytes);
  int32_t* bc = w.take<int32_t>((size_t)nblocks * B);
  const size_t esz = dtype == ISR_DTYPE_F32 ? 4 : 2;
  const size_t fstride = (size_t)H * W * C, mstride = (size_t)H * W * mask_pix_stride;
  // rows past an image's count and the padding columns are zero: K1 runs over all B * S capacity rows
  ISR_CHECK_HIP(hipMemsetAsync(Q, 0, (size_t)B * S * ldq * esz, stream));
  prep_count_kernel<<<dim3(nblocks, 1, B), kThreads, 0, stream>>>(mask, mask_pix_stride, mstride, W, Ws, step, S, bc);
  prep_scan_kernel<<<dim3(1, 1, B), 1024, 0, stream>>>(bc, nblocks, n_dev);
#define ISR_PREP(DT)                                                                                                    \
  prep_scatter_kernel<DT><<<dim3(nblocks, 1, B), kThreads, 0, stream>>>(feat, fstride, W, C, c0, D, mask, mask_pix_stride, \
                                                                        mstride, Ws, step, S, bc, ldq, Q, pix_xy)
  if (dtype == ISR_DTYPE_F32) ISR_PREP(ISR_DTYPE_F32);
  else if (dtype == ISR_DTYPE_BF16) ISR_PREP(ISR_DTYPE_BF16);
  else ISR_PREP(ISR_DTYPE_BF16_LOG2);
#undef ISR_PREP
  ISR_CHECK_LAUNCH("prep_queries kernels");
  return ISR_OK;
}

extern "C" int isr_prep_queries(const float* feat, int H, int W, int C, int c0, int D, const uint8_t* mask,
                                int mask_pix_stride, int step, int dtype, int ldq, void* Q, float* pix_xy,
                                int32_t* n_dev, void* ws, size_t ws_bytes, isr_stream_t stream_) {
  return isr_prep_queries_batch(feat, 1, H, W, C, c0, D, mask, mask_pix_stride, step, dtype, ldq, Q, pix_xy, n_dev, ws,
                                ws_bytes, stream_);
}
