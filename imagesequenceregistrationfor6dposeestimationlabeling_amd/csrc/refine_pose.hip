// refine_pose.hip — a16: the objective of refine_pose() (pose_refine.py:58-91) and its gradient.
//
//   p_img  = (K_crop [R|t]) X,  p = p_img[:2] / p_img[2]
//   score  = -( mean_i <keys_i, bilinear(query_img, p_i)> - mean_i bilinear(denom_img, p_i) ) / 2
// with F.grid_sample(align_corners=False, padding_mode='border', mode='bilinear') on
// p_norm = (p + 0.5) * 2 / res - 1, i.e. the sample position in pixel units is p itself (pixel
// centres at integers) clamped to [0, res-1].  The reference differentiates this with autograd and
// a cv2.Rodrigues round trip; only the translation is live there (pose_refine.py:73-76 builds R from
// a constant), so the analytic gradient the reference-shaped entry returns is d score / d t.
// isr_refine_objective_full adds d score / d R (9, row-major: sum_i g_i X_i^T with g_i the gradient with
// respect to the camera-frame point) for the evidently intended variant that also optimises the rotation
// (the caller chains it with the Rodrigues Jacobian).
// One thread per visible surface point, f64 accumulation, fixed-shape tree reduction.
#include "isr_common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kBlocks = 64;
constexpr int kAcc = 14;

struct P34 { double p[12]; double k[9]; };

// ---- interpolation != 'bilinear' (pose_refine.py:60-68 forwards `mode=interpolation` to F.grid_sample): separable taps
// per axis, as torch's grid sampler defines them for align_corners=False, padding_mode='border':
//   nearest  the source coordinate is clipped to [0, res-1], then rounded half-to-even (nearbyint); no gradient;
//   bicubic  the source coordinate is NOT clipped; cubic-convolution coefficients with A = -0.75 on the fractional part;
//            each of the four taps is clipped to [0, res-1] (get_value_bounded); gradient = derivative of the coefficients.
struct Taps {
  int n;
  int ix[4];
  double w[4], dw[4];
};

template <int MODE>
__device__ __forceinline__ Taps make_taps(double u, int res) {
  Taps T;
  const double hi = (double)(res - 1);
  if (MODE == 1) {
    const double uc = fmin(fmax(u, 0.0), hi);
    T.n = 1;
    T.ix[0] = (int)rint(uc);
    T.w[0] = 1.0;
    T.dw[0] = 0.0;
  } else {
    const double fl = floor(fmin(fmax(u, -4.0), hi + 4.0));       // far outside every tap clips to the border anyway
    const double t = fmin(fmax(u, -4.0), hi + 4.0) - fl;
    const int x0 = (int)fl;
    const double A = -0.75;
    const double x1 = t + 1.0, s = 1.0 - t, x3 = 2.0 - t;
    T.n = 4;
    T.w[0] = ((A * x1 - 5.0 * A) * x1 + 8.0 * A) * x1 - 4.0 * A;
    T.w[1] = ((A + 2.0) * t - (A + 3.0)) * t * t + 1.0;
    T.w[2] = ((A + 2.0) * s - (A + 3.0)) * s * s + 1.0;
    T.w[3] = ((A * x3 - 5.0 * A) * x3 + 8.0 * A) * x3 - 4.0 * A;
    const bool inside = u > -4.0 && u < hi + 4.0;                  // beyond that the value is constant in u
    T.dw[0] = inside ? (3.0 * A * x1 - 10.0 * A) * x1 + 8.0 * A : 0.0;
    T.dw[1] = inside ? (3.0 * (A + 2.0) * t - 2.0 * (A + 3.0)) * t : 0.0;
    T.dw[2] = inside ? -((3.0 * (A + 2.0) * s - 2.0 * (A + 3.0)) * s) : 0.0;
    T.dw[3] = inside ? -((3.0 * A * x3 - 10.0 * A) * x3 + 8.0 * A) : 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int x = x0 - 1 + i;
      T.ix[i] = x < 0 ? 0 : (x > res - 1 ? res - 1 : x);
    }
  }
  return T;
}

template <int MODE>
__global__ __launch_bounds__(kThreads) void refine_obj_taps_kernel(const float* __restrict__ X, const float* __restrict__ keys,
                                                                   int N, int e, const float* __restrict__ qimg,
                                                                   const float* __restrict__ denom, int res, P34 P,
                                                                   double* __restrict__ partial) {
  __shared__ double red[kThreads / 64][kAcc];
  double acc[kAcc];
#pragma unroll
  for (int q = 0; q < kAcc; ++q) acc[q] = 0.0;
  for (int i = blockIdx.x * kThreads + threadIdx.x; i < N; i += kBlocks * kThreads) {
    const double x = X[3 * (size_t)i], y = X[3 * (size_t)i + 1], z = X[3 * (size_t)i + 2];
    const double px = P.p[0] * x + P.p[1] * y + P.p[2] * z + P.p[3];
    const double py = P.p[4] * x + P.p[5] * y + P.p[6] * z + P.p[7];
    const double pz = P.p[8] * x + P.p[9] * y + P.p[10] * z + P.p[11];
    const double ipz = 1.0 / pz;
    const double u = px * ipz, v = py * ipz;
    const Taps tx = make_taps<MODE>(u, res), ty = make_taps<MODE>(v, res);
    double nom = 0.0, dnx = 0.0, dny = 0.0, den = 0.0, ddx = 0.0, ddy = 0.0;
    for (int b = 0; b < ty.n; ++b)
      for (int a = 0; a < tx.n; ++a) {
        const size_t o = (size_t)ty.ix[b] * res + tx.ix[a];
        double dot = 0.0;
        for (int c = 0; c < e; ++c) dot += (double)keys[(size_t)i * e + c] * (double)qimg[o * e + c];
        const double d = denom[o];
        const double w = ty.w[b] * tx.w[a], wu = ty.w[b] * tx.dw[a], wv = ty.dw[b] * tx.w[a];
        nom += w * dot; dnx += wu * dot; dny += wv * dot;
        den += w * d;   ddx += wu * d;   ddy += wv * d;
      }
    const double fx = dnx - ddx, fy = dny - ddy;
    acc[0] += nom;
    acc[1] += den;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double g = (fx * (P.k[j] - u * P.k[6 + j]) + fy * (P.k[3 + j] - v * P.k[6 + j])) * ipz;
      acc[2 + j] += g;
      acc[5 + 3 * j] += g * x;
      acc[5 + 3 * j + 1] += g * y;
      acc[5 + 3 * j + 2] += g * z;
    }
  }
#pragma unroll
  for (int q = 0; q < kAcc; ++q) {
    double s = acc[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = s;
  }
  __syncthreads();
  if (threadIdx.x < kAcc)
    partial[(size_t)blockIdx.x * kAcc + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

__global__ __launch_bounds__(kThreads) void refine_obj_kernel(const float* __restrict__ X, const float* __restrict__ keys,
                                                              int N, int e, const float* __restrict__ qimg,
                                                              const float* __restrict__ denom, int res, P34 P,
                                                              double* __restrict__ partial) {
  __shared__ double red[kThreads / 64][kAcc];
  double acc[kAcc];                 // sum nominator, sum denominator, d/dt (3) and d/dR (9) of (nom - den)
#pragma unroll
  for (int q = 0; q < kAcc; ++q) acc[q] = 0.0;
  for (int i = blockIdx.x * kThreads + threadIdx.x; i < N; i += kBlocks * kThreads) {
    const double x = X[3 * (size_t)i], y = X[3 * (size_t)i + 1], z = X[3 * (size_t)i + 2];
    const double px = P.p[0] * x + P.p[1] * y + P.p[2] * z + P.p[3];
    const double py = P.p[4] * x + P.p[5] * y + P.p[6] * z + P.p[7];
    const double pz = P.p[8] * x + P.p[9] * y + P.p[10] * z + P.p[11];
    const double ipz = 1.0 / pz;
    const double u = px * ipz, v = py * ipz;
    // border padding: clamp, zero gradient outside
    const double hi = (double)(res - 1);
    const double uc = fmin(fmax(u, 0.0), hi), vc = fmin(fmax(v, 0.0), hi);
    const double gu = (u > 0.0 && u < hi) ? 1.0 : 0.0, gv = (v > 0.0 && v < hi) ? 1.0 : 0.0;
    int x0 = (int)floor(uc), y0 = (int)floor(vc);
    x0 = x0 > res - 2 ? res - 2 : x0;
    y0 = y0 > res - 2 ? res - 2 : y0;
    if (res < 2) { x0 = 0; y0 = 0; }
    const int x1 = res < 2 ? 0 : x0 + 1, y1 = res < 2 ? 0 : y0 + 1;
    const double wx = uc - x0, wy = vc - y0;
    const size_t o00 = (size_t)y0 * res + x0, o10 = (size_t)y0 * res + x1, o01 = (size_t)y1 * res + x0,
                 o11 = (size_t)y1 * res + x1;
    double nom = 0.0, dnx = 0.0, dny = 0.0;
    for (int c = 0; c < e; ++c) {
      const double k = keys[(size_t)i * e + c];
      const double v00 = qimg[o00 * e + c], v10 = qimg[o10 * e + c], v01 = qimg[o01 * e + c], v11 = qimg[o11 * e + c];
      nom += k * ((1 - wy) * ((1 - wx) * v00 + wx * v10) + wy * ((1 - wx) * v01 + wx * v11));
      dnx += k * ((1 - wy) * (v10 - v00) + wy * (v11 - v01));
      dny += k * ((1 - wx) * (v01 - v00) + wx * (v11 - v10));
    }
    const double d00 = denom[o00], d10 = denom[o10], d01 = denom[o01], d11 = denom[o11];
    const double den = (1 - wy) * ((1 - wx) * d00 + wx * d10) + wy * ((1 - wx) * d01 + wx * d11);
    const double ddx = (1 - wy) * (d10 - d00) + wy * (d11 - d01);
    const double ddy = (1 - wx) * (d01 - d00) + wx * (d11 - d10);
    const double fx = (dnx - ddx) * gu, fy = (dny - ddy) * gv;  // d(nom - den)/d(u, v)
    acc[0] += nom;
    acc[1] += den;
    // d(u,v)/dt = (K_row0 - u K_row2, K_row1 - v K_row2) / pz
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double g = (fx * (P.k[j] - u * P.k[6 + j]) + fy * (P.k[3 + j] - v * P.k[6 + j])) * ipz;   // d/d(camera point)_j
      acc[2 + j] += g;
      acc[5 + 3 * j] += g * x;
      acc[5 + 3 * j + 1] += g * y;
      acc[5 + 3 * j + 2] += g * z;
    }
  }
#pragma unroll
  for (int q = 0; q < kAcc; ++q) {
    double s = acc[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = s;
  }
  __syncthreads();
  if (threadIdx.x < kAcc)
    partial[(size_t)blockIdx.x * kAcc + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// nout = 4: {score, d/dt}; 13: {score, d/dt (3), d/dR (9)}
__global__ void refine_obj_reduce_kernel(const double* __restrict__ partial, int N, double* __restrict__ out, int nout) {
  __shared__ double v[kAcc];
  if (threadIdx.x < kAcc) {
    double s = 0.0;
    for (int b = 0; b < kBlocks; ++b) s += partial[(size_t)b * kAcc + threadIdx.x];
    v[threadIdx.x] = s;
  }
  __syncthreads();
  const double n = (double)N;
  if (threadIdx.x == 0) out[0] = -(v[0] / n - v[1] / n) / 2.0;
  if (threadIdx.x >= 1 && threadIdx.x < nout) out[threadIdx.x] = -(v[1 + threadIdx.x] / n) / 2.0;
}

}  // namespace

static int refine_impl(const float* X, const float* keys, int N, int e, const float* query_img, const float* denom_img,
                       int res, int mode, const double* Kcrop, const double* Rt, double* out, int nout, void* ws,
                       size_t ws_bytes, isr_stream_t stream_) {
  ISR_REQUIRE(X && keys && query_img && denom_img && Kcrop && Rt && out, "isr_refine_objective: null pointer");
  ISR_REQUIRE(N > 0 && e > 0 && res > 0, "isr_refine_objective: N=%d e=%d res=%d", N, e, res);
  ISR_REQUIRE(mode >= ISR_INTERP_BILINEAR && mode <= ISR_INTERP_BICUBIC, "isr_refine_objective: interpolation mode %d", mode);
  const size_t need = sizeof(double) * kBlocks * kAcc + 256;
  if (!ws || ws_bytes < need) {
    isr::set_error("isr_refine_objective: workspace %zu < %zu", ws_bytes, need);
    return ISR_ERR_WORKSPACE;
  }
  P34 P;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c)
      P.p[4 * r + c] = Kcrop[3 * r] * Rt[c] + Kcrop[3 * r + 1] * Rt[4 + c] + Kcrop[3 * r + 2] * Rt[8 + c];
  for (int i = 0; i < 9; ++i) P.k[i] = Kcrop[i];
  hipStream_t stream = isr::as_stream(stream_);
  isr::Workspace w(ws, ws_bytes);
  double* partial = w.take<double>((size_t)kBlocks * kAcc);
  if (mode == ISR_INTERP_NEAREST)
    refine_obj_taps_kernel<1><<<kBlocks, kThreads, 0, stream>>>(X, keys, N, e, query_img, denom_img, res, P, partial);
  else if (mode == ISR_INTERP_BICUBIC)
    refine_obj_taps_kernel<2><<<kBlocks, kThreads, 0, stream>>>(X, keys, N, e, query_img, denom_img, res, P, partial);
  else
    refine_obj_kernel<<<kBlocks, kThreads, 0, stream>>>(X, keys, N, e, query_img, denom_img, res, P, partial);
  refine_obj_reduce_kernel<<<1, 64, 0, stream>>>(partial, N, out, nout);
  ISR_CHECK_LAUNCH("refine objective kernels");
  return ISR_OK;
}

extern "C" int isr_refine_objective(const float* X, const float* keys, int N, int e, const float* query_img,
                                    const float* denom_img, int res, int interpolation, const double* Kcrop,
                                    const double* Rt, double* out4, void* ws, size_t ws_bytes, isr_stream_t stream) {
  return refine_impl(X, keys, N, e, query_img, denom_img, res, interpolation, Kcrop, Rt, out4, 4, ws, ws_bytes, stream);
}

extern "C" int isr_refine_objective_full(const float* X, const float* keys, int N, int e, const float* query_img,
                                         const float* denom_img, int res, int interpolation, const double* Kcrop,
                                         const double* Rt, double* out13, void* ws, size_t ws_bytes, isr_stream_t stream) {
  return refine_impl(X, keys, N, e, query_img, denom_img, res, interpolation, Kcrop, Rt, out13, 13, ws, ws_bytes, stream);
}
