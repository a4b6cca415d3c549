// crop_normalize.hip — the image front end of the per-image loop, SURVEY 8(f)-2:
//   x, y, w, h = cv2.boundingRect(mask[:, :, 0])                        inference.py:202
//   cropRGB  = cv2.warpAffine(rgb,  M, (224, 224))                       inference.py:224
//   cropMask = cv2.warpAffine(mask, M, (224, 224))                       inference.py:225
//   if useMask: cropRGB[cropMask[:, :, 0] == 0] = 0                      inference.py:226-227
//   inputIM = normalize(cropRGB).astype("float32")  -> (3, 224, 224)     inference.py:135-141, 232
// as two kernels with the image on blockIdx.z, so a group of images costs two launches and the crop
// never visits the host.  HBM-bound byte work: one pass over the mask for the box, then every output
// pixel reads its four source neighbours (the crop's footprint of the frame, once) and writes 3 f32 + 1 u8.
//
// warpAffine here is the textbook definition (OpenCV is not in the image: parity unpinned): the inverse
// map dst -> src in f64, bilinear weights in f64, neighbours outside the frame count as 0
// (BORDER_CONSTANT, value 0, cv2's default), the sum rounded half-to-even to u8.  OpenCV itself
// [from memory] quantises the sample position to 1/32 px and the weights to 2^-15 before it sums, so its
// bytes can differ from these by a grey level at some pixels.  oracle/preprocess_oracle.py states the
// same arithmetic in NumPy, operation for operation.
#include "isr_common.hpp"

namespace {

constexpr int kMaxBatch = 16;

struct WarpBatch {
  double minv[kMaxBatch][6];   // dst (x, y, 1) -> src (x, y), row-major 2x3
};

// cv2.boundingRect of channel 0: {x, y, w, h} of the non-zero pixels, {0, 0, 0, 0} when there are none.
__global__ __launch_bounds__(256) void mask_bbox_kernel(const uint8_t* __restrict__ mask, int H, int W, int C,
                                                        int32_t* __restrict__ bbox) {
  __shared__ int32_t red[4][4];
  const int b = blockIdx.z;
  const uint8_t* m = mask + (size_t)b * H * W * C;
  int x0 = W, y0 = H, x1 = -1, y1 = -1;
  for (int i = threadIdx.x; i < H * W; i += 256) {
    if (m[(size_t)i * C]) {
      const int y = i / W, x = i - y * W;
      x0 = min(x0, x); x1 = max(x1, x); y0 = min(y0, y); y1 = max(y1, y);
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    x0 = min(x0, __shfl_xor(x0, o, 64)); y0 = min(y0, __shfl_xor(y0, o, 64));
    x1 = max(x1, __shfl_xor(x1, o, 64)); y1 = max(y1, __shfl_xor(y1, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    const int w = threadIdx.x >> 6;
    red[w][0] = x0; red[w][1] = y0; red[w][2] = x1; red[w][3] = y1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      red[0][0] = min(red[0][0], red[w][0]); red[0][1] = min(red[0][1], red[w][1]);
      red[0][2] = max(red[0][2], red[w][2]); red[0][3] = max(red[0][3], red[w][3]);
    }
    const bool any = red[0][2] >= 0;
    bbox[4 * b] = any ? red[0][0] : 0;
    bbox[4 * b + 1] = any ? red[0][1] : 0;
    bbox[4 * b + 2] = any ? red[0][2] - red[0][0] + 1 : 0;
    bbox[4 * b + 3] = any ? red[0][3] - red[0][1] + 1 : 0;
  }
}

// bilinear sample of channel c at (sx, sy); outside neighbours are 0; rounded half-to-even to u8
__device__ __forceinline__ uint8_t sample_u8(const uint8_t* __restrict__ img, int H, int W, int C, int c, double sx,
                                             double sy) {
  const double fx0 = floor(sx), fy0 = floor(sy);
  const double fx = sx - fx0, fy = sy - fy0;
  const long x0 = (long)fx0, y0 = (long)fy0;
  auto px = [&](long x, long y) -> double {
    return (x >= 0 && x < W && y >= 0 && y < H) ? (double)img[((size_t)y * W + x) * C + c] : 0.0;
  };
  const double w00 = (1.0 - fx) * (1.0 - fy), w01 = fx * (1.0 - fy), w10 = (1.0 - fx) * fy, w11 = fx * fy;
  const double v = ((px(x0, y0) * w00 + px(x0 + 1, y0) * w01) + px(x0, y0 + 1) * w10) + px(x0 + 1, y0 + 1) * w11;
  return (uint8_t)rint(v);                       // 0 <= v <= 255
}

struct NormStats { double mu[3], sd[3]; };

__global__ __launch_bounds__(256) void crop_normalize_kernel(
    const uint8_t* __restrict__ rgb, const uint8_t* __restrict__ mask, int H, int W, int Cm, WarpBatch wb, int r,
    int use_mask, NormStats ns, float* __restrict__ out, uint8_t* __restrict__ crop_mask) {
  const int b = blockIdx.z;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= r * r) return;
  const int y = p / r, x = p - y * r;
  const double* mi = wb.minv[b];
  const double sx = (mi[0] * (double)x + mi[1] * (double)y) + mi[2];
  const double sy = (mi[3] * (double)x + mi[4] * (double)y) + mi[5];
  const uint8_t* im = rgb + (size_t)b * H * W * 3;
  const uint8_t* mk = mask + (size_t)b * H * W * Cm;
  const uint8_t m = sample_u8(mk, H, W, Cm, 0, sx, sy);
  crop_mask[(size_t)b * r * r + p] = m;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    uint8_t v = sample_u8(im, H, W, 3, c, sx, sy);
    if (use_mask && m == 0) v = 0;
    // normalize(): img / 255, (img - mu) / std in f64, then .astype("float32")
    out[((size_t)b * 3 + c) * r * r + p] = (float)((((double)v / 255.0) - ns.mu[c]) / ns.sd[c]);
  }
}

}  // namespace

extern "C" int isr_mask_bbox(const uint8_t* mask, int B, int H, int W, int C, int32_t* bbox_dev, isr_stream_t stream) {
  ISR_REQUIRE(mask && bbox_dev, "isr_mask_bbox: null pointer");
  ISR_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0 && C > 0, "isr_mask_bbox: B=%d H=%d W=%d C=%d", B, H, W, C);
  mask_bbox_kernel<<<dim3(1, 1, B), 256, 0, isr::as_stream(stream)>>>(mask, H, W, C, bbox_dev);
  ISR_CHECK_LAUNCH("mask_bbox_kernel");
  return ISR_OK;
}

extern "C" int isr_crop_normalize(const uint8_t* rgb, const uint8_t* mask, int B, int H, int W, int mask_channels,
                                  const double* M_host, int out_size, int use_mask, const double* mean3,
                                  const double* std3, float* out, uint8_t* crop_mask, isr_stream_t stream) {
  ISR_REQUIRE(rgb && mask && M_host && mean3 && std3 && out && crop_mask, "isr_crop_normalize: null pointer");
  ISR_REQUIRE(B > 0 && H > 0 && W > 0 && mask_channels > 0 && out_size > 0,
              "isr_crop_normalize: B=%d H=%d W=%d mask_channels=%d out_size=%d", B, H, W, mask_channels, out_size);
  NormStats ns;
  for (int c = 0; c < 3; ++c) {
    ns.mu[c] = mean3[c];
    ns.sd[c] = std3[c];
    ISR_REQUIRE(std3[c] != 0.0, "isr_crop_normalize: std[%d] = 0", c);
  }
  const int r = out_size;
  for (int b0 = 0; b0 < B; b0 += kMaxBatch) {
    const int nb = (B - b0 < kMaxBatch) ? B - b0 : kMaxBatch;
    WarpBatch wb;
    for (int b = 0; b < kMaxBatch; ++b) {
      const double* M = M_host + 6 * (size_t)(b0 + (b < nb ? b : 0));
      // M maps source -> crop (the reference passes it to cv2.warpAffine without WARP_INVERSE_MAP): invert it
      const double det = M[0] * M[4] - M[1] * M[3];
      ISR_REQUIRE(det != 0.0, "isr_crop_normalize: singular affine for image %d", b0 + b);
      const double i00 = M[4] / det, i01 = -M[1] / det, i10 = -M[3] / det, i11 = M[0] / det;
      wb.minv[b][0] = i00; wb.minv[b][1] = i01; wb.minv[b][2] = -(i00 * M[2] + i01 * M[5]);
      wb.minv[b][3] = i10; wb.minv[b][4] = i11; wb.minv[b][5] = -(i10 * M[2] + i11 * M[5]);
    }
    crop_normalize_kernel<<<dim3((r * r + 255) / 256, 1, nb), 256, 0, isr::as_stream(stream)>>>(
        rgb + (size_t)b0 * H * W * 3, mask + (size_t)b0 * H * W * mask_channels, H, W, mask_channels, wb, r, use_mask, ns,
        out + (size_t)b0 * 3 * r * r, crop_mask + (size_t)b0 * r * r);
  }
  ISR_CHECK_LAUNCH("crop_normalize_kernel");
  return ISR_OK;
}
