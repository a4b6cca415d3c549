// ransac.hip — K2: PnP + RANSAC on gfx950.
//
// Replaces  pnp(h3d, h2d, cam, itr, reperr, flag=SOLVEPNP_P3P):
//             cv2.solvePnPRansac(h3d, h2d, cam, None, iterationsCount=itr,
//                                reprojectionError=reperr, flags=cv2.SOLVEPNP_P3P); cv2.Rodrigues
//           inference.py:123-134 (= finalposes.py:20-30 = choosePose.py:23-33), call site :293.
// OpenCV is not in the image and the reference pins no version or test vector, so this file OWNS
// the algorithm (DESIGN.md "K2"); the CPU statement of it is oracle/pnp_oracle.py.
//
//   1. hypotheses   thread h: Philox4x32-10(key = seed, counter = h) -> 4 correspondence indices
//                   (mulhi(x, M)); P3P on the first three in f64 (degenerate-conic method: one
//                   root of the pencil cubic by Newton, the singular conic split into its two
//                   lines, each line intersected with a second conic), <= 4 poses; the pose with
//                   the smallest reprojection error on the 4th point is kept.
//   2. scoring      all H hypotheses x all M correspondences, f32, division-free inlier test
//                       z > 0  and  (x - u z)^2 + (y - v z)^2 <= (reperr z)^2
//                   lane = correspondence (coalesced loads, 4 per lane in registers), loop over
//                   hypotheses with the 3x4 projection in SGPRs; ballot + s_bcnt1 counts a wave's
//                   inliers, LDS integer atomics per block, one global integer atomic per
//                   (block, hypothesis): deterministic.
//                   Adaptive termination (cv2.solvePnPRansac's `confidence`, default 0.99): hypotheses
//                   are scored in stages [0,32), [32,96), [96,224), ... (boundaries 32 (2^k - 1)); a
//                   stage runs only if, at the boundary b before it, the best count c so far does
//                   NOT yet give the confidence:  (1 - (c/M)^4)^b > 1 - confidence  (the standard
//                   RANSAC rule for 4-point samples, evaluated with multiplications only so that
//                   oracle and device decide identically).  confidence >= 1: every hypothesis.
//   3. best         arg-max count, lowest h on ties; its inlier bitmask.
//   4. refit        Gauss-Newton on the reprojection error over the inliers, f64, fixed-shape
//                   tree reduction of J^T J / J^T r, 6x6 Cholesky on the device.  Then ONE round of
//                   local optimisation: the inlier mask is recomputed under the refitted pose (same
//                   f32 test) and the refit repeated on it — a hypothesis from 4 noisy points captures
//                   only part of the consensus set, and with the adaptive termination the loop may
//                   stop after 32 of them.  The reported inliers are those of the RETURNED pose (the mask is
//                   evaluated once more under the final refit; cv2 reports the RANSAC model's consensus set
//                   instead, as far as is known — a documented deviation, like the confidence rule).  A refit
//                   over fewer than 4 correspondences is skipped (the pose it started from is kept).
//   5. compaction   inlier mask -> ascending int32 indices.
// Every step reads M and the status from device memory: the whole chain is enqueued without a
// host round trip.
// Every kernel carries an IMAGE dimension (blockIdx.z) with the cameras and seeds of up to kMaxBatch
// images in the kernel arguments: the per-image chain of inference.py:293 over a group of images is
// one chain of launches for the whole group (isr_pnp_ransac_batch).  The single-image entry points
// are the B = 1 case of the same kernels: bit-identical by construction.
#include "isr_common.hpp"
#include "p3p_device.hpp"

namespace {

using namespace isr_p3p;

constexpr int kMaxBatch = 16;    // images per set_imgs_kernel launch (kernel-argument space: 16 x (144 + 8) B)
constexpr int kChainMax = 128;   // images per launch chain (the scratch of one chain: ~100 KB per image at H = 500)

struct ImgBatch {
  Cam cam[kMaxBatch];
  uint32_t seed_lo[kMaxBatch], seed_hi[kMaxBatch];
};

// The cameras and seeds of a chain's images live in DEVICE memory (the caller's scratch): the kernels index them by
// blockIdx.z, so one chain serves up to kChainMax images — round 2 passed them by value, 16 images per chain of ~40
// launches, and a 64-crop group at the reference's shape spent more host time launching than the GPU spent working.
struct ImgDev {
  Cam cam;
  uint32_t seed_lo, seed_hi;
};

__global__ void set_imgs_kernel(ImgBatch ib, int nb, ImgDev* __restrict__ dst) {
  const int b = threadIdx.x;
  if (b >= nb) return;
  dst[b].cam = ib.cam[b];
  dst[b].seed_lo = ib.seed_lo[b];
  dst[b].seed_hi = ib.seed_hi[b];
}

__device__ __forceinline__ void p3p_body(const float* __restrict__ p3d, const float* __restrict__ p2d,
                                         const int32_t* __restrict__ M_dev, int M_cap, const ImgDev& img, int H,
                                         double* __restrict__ Rt_out, uint8_t* __restrict__ ok_out,
                                         int32_t* __restrict__ sample_out) {
  const int b = blockIdx.z;
  const Cam& cam = img.cam;
  const uint32_t seed_lo = img.seed_lo, seed_hi = img.seed_hi;
  p3d += (size_t)b * M_cap * 3; p2d += (size_t)b * M_cap * 2;
  Rt_out += (size_t)b * H * 12; ok_out += (size_t)b * H;
  if (sample_out) sample_out += (size_t)b * H * 4;
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= H) return;
  const int M = M_dev[b];
  double* Rt = Rt_out + 12 * (size_t)h;
  uint8_t ok = 0;
  int s[4] = {0, 0, 0, 0};
  if (M >= 4) {
    uint32_t rnd[4];
    philox4x32_10((uint32_t)h, 0u, 0u, 0u, seed_lo, seed_hi, rnd);
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = (int)(((uint64_t)rnd[j] * (uint64_t)M) >> 32);
    // a sample that repeats a correspondence is rejected (degenerate, and its 4th-point test ties)
    const bool distinct = s[0] != s[1] && s[0] != s[2] && s[0] != s[3] && s[1] != s[2] && s[1] != s[3] && s[2] != s[3];
    P3PIn in;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      in.x[j] = {(double)p3d[3 * (size_t)s[j]], (double)p3d[3 * (size_t)s[j] + 1], (double)p3d[3 * (size_t)s[j] + 2]};
      const double u = p2d[2 * (size_t)s[j]], v = p2d[2 * (size_t)s[j] + 1];
      V3 y = {cam.ki[0] * u + cam.ki[1] * v + cam.ki[2], cam.ki[3] * u + cam.ki[4] * v + cam.ki[5],
              cam.ki[6] * u + cam.ki[7] * v + cam.ki[8]};
      in.y[j] = scale(y, 1.0 / sqrt(dot(y, y)));
    }
    double lam[4][3];
    const int n = distinct ? p3p_depths(in, lam) : 0;
    const V3 X4 = {(double)p3d[3 * (size_t)s[3]], (double)p3d[3 * (size_t)s[3] + 1], (double)p3d[3 * (size_t)s[3] + 2]};
    const double u4 = p2d[2 * (size_t)s[3]], v4 = p2d[2 * (size_t)s[3] + 1];
    double best = 1e300;
    for (int i = 0; i < n; ++i) {
      double cand[12];
      if (!pose_from_depths(in, lam[i], cand)) continue;
      const double xc = cand[0] * X4.x + cand[1] * X4.y + cand[2] * X4.z + cand[3];
      const double yc = cand[4] * X4.x + cand[5] * X4.y + cand[6] * X4.z + cand[7];
      const double zc = cand[8] * X4.x + cand[9] * X4.y + cand[10] * X4.z + cand[11];
      if (!(zc > 0.0)) continue;
      const double px = cam.k[0] * xc + cam.k[1] * yc + cam.k[2] * zc;
      const double py = cam.k[3] * xc + cam.k[4] * yc + cam.k[5] * zc;
      const double pz = cam.k[6] * xc + cam.k[7] * yc + cam.k[8] * zc;
      const double eu = px / pz - u4, ev = py / pz - v4;
      const double e = eu * eu + ev * ev;
      if (e < best) {
        best = e;
        ok = 1;
#pragma unroll
        for (int k = 0; k < 12; ++k) Rt[k] = cand[k];
      }
    }
  }
  if (!ok) {
#pragma unroll
    for (int k = 0; k < 12; ++k) Rt[k] = (k % 5 == 0) ? 1.0 : 0.0;
  }
  ok_out[h] = ok;
  if (sample_out) {
#pragma unroll
    for (int j = 0; j < 4; ++j) sample_out[4 * (size_t)h + j] = s[j];
  }
}

__global__ void p3p_kernel(const float* __restrict__ p3d, const float* __restrict__ p2d,
                           const int32_t* __restrict__ M_dev, int M_cap, const ImgDev* __restrict__ ib, int H,
                           double* __restrict__ Rt_out, uint8_t* __restrict__ ok_out,
                           int32_t* __restrict__ sample_out) {
  p3p_body(p3d, p2d, M_dev, M_cap, ib[blockIdx.z], H, Rt_out, ok_out, sample_out);
}

// one image, camera and seed by value: isr_p3p_hypotheses has no scratch to put them in
__global__ void p3p_single_kernel(const float* __restrict__ p3d, const float* __restrict__ p2d,
                                  const int32_t* __restrict__ M_dev, int M_cap, ImgDev img, int H,
                                  double* __restrict__ Rt_out, uint8_t* __restrict__ ok_out,
                                  int32_t* __restrict__ sample_out) {
  p3p_body(p3d, p2d, M_dev, M_cap, img, H, Rt_out, ok_out, sample_out);
}

// Every root of the P3P solver for S independent 3-point problems (test / diagnostic entry point: the
// production kernels keep one root per sample, this one shows the whole set so it can be compared with
// the oracle's root set): X (S,3,3) f64 object points, uv (S,3,2) f64 pixels -> poses (S,4,12), n (S).
__global__ void p3p_all_roots_kernel(const double* __restrict__ X, const double* __restrict__ uv, Cam cam, int S,
                                     double* __restrict__ poses, int32_t* __restrict__ n_out) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  P3PIn in;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    in.x[j] = {X[9 * (size_t)s + 3 * j], X[9 * (size_t)s + 3 * j + 1], X[9 * (size_t)s + 3 * j + 2]};
    in.y[j] = bearing(cam, uv[6 * (size_t)s + 2 * j], uv[6 * (size_t)s + 2 * j + 1]);
  }
  double lam[4][3];
  const int n = p3p_depths(in, lam);
  int k = 0;
  for (int i = 0; i < n; ++i) {
    double cand[12];
    if (!pose_from_depths(in, lam[i], cand)) continue;
#pragma unroll
    for (int e = 0; e < 12; ++e) poses[48 * (size_t)s + 12 * k + e] = cand[e];
    ++k;
  }
  n_out[s] = k;
}

// ------------------------------------------------------------------------------- scoring
// Pm (H,12) f32 = float(K [R|t]) with the f64 fma order of oracle/isr_oracle.c:proj_matrix_f32.
// Also zeroes the image's inlier counters (no memset launch).
__global__ void proj_matrix_kernel(const double* __restrict__ Rt, const ImgDev* __restrict__ ib, int H, float* __restrict__ Pm,
                                   int32_t* __restrict__ n_inl) {
  const int b = blockIdx.z;
  const Cam& cam = ib[b].cam;
  Rt += (size_t)b * H * 12; Pm += (size_t)b * H * 12; n_inl += (size_t)b * H;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < H) n_inl[e] = 0;
  if (e >= H * 12) return;
  const int h = e / 12, r = (e % 12) / 4, c = e % 4;
  const double* T = Rt + 12 * (size_t)h;
  Pm[e] = (float)fma(cam.k[3 * r + 2], T[8 + c], fma(cam.k[3 * r + 1], T[4 + c], cam.k[3 * r] * T[c]));
}

__device__ __forceinline__ bool inlier_f32(const float* __restrict__ Pm, float X, float Y, float Z,
                                           float u, float v, float reperr) {
  const float x = __builtin_fmaf(Pm[2], Z, __builtin_fmaf(Pm[1], Y, __builtin_fmaf(Pm[0], X, Pm[3])));
  const float y = __builtin_fmaf(Pm[6], Z, __builtin_fmaf(Pm[5], Y, __builtin_fmaf(Pm[4], X, Pm[7])));
  const float z = __builtin_fmaf(Pm[10], Z, __builtin_fmaf(Pm[9], Y, __builtin_fmaf(Pm[8], X, Pm[11])));
  const float ex = __builtin_fmaf(-u, z, x), ey = __builtin_fmaf(-v, z, y);
  const float e2 = __builtin_fmaf(ey, ey, ex * ex);
  const float lim = reperr * z;
  return (z > 0.0f) & (e2 <= lim * lim);   // bitwise: straight-line code
}

constexpr int kScoreThreads = 256;
#ifndef ISR_SCORE_CPL
#define ISR_SCORE_CPL 4
#endif
#ifndef ISR_SCORE_HC
#define ISR_SCORE_HC 32
#endif
constexpr int kCPL = ISR_SCORE_CPL;   // correspondences per lane
constexpr int kHC = ISR_SCORE_HC;     // hypotheses per block: grid.y = ceil(H / kHC) keeps >= 8 waves per SIMD
constexpr int kMaxH = 8192;
constexpr int kStage0 = 32;           // hypotheses of the first scoring stage (a multiple of kHC)
static_assert(kStage0 % kHC == 0, "stage boundaries are chunk boundaries");

// RANSAC's stopping rule after b hypotheses with best inlier count c of M, 4-point samples:
//   (1 - (c/M)^4)^b <= 1 - confidence.   Multiplications only (binary powering in a fixed order):
// IEEE arithmetic, the same decision in oracle/pnp_oracle.py:stop_rule bit for bit.
__host__ __device__ inline bool ransac_stop(int c, int M, int b, double one_minus_conf) {
  if (c < 4 || M <= 0 || !(one_minus_conf > 0.0)) return false;
  const double w = (double)c / (double)M;
  const double w2 = w * w;
  double base = 1.0 - w2 * w2, q = 1.0;
  for (int e = b; e > 0; e >>= 1) {
    if (e & 1) q = q * base;
    base = base * base;
  }
  return q <= one_minus_conf;
}

// Hypotheses [h_lo, h_hi) of every image; blockIdx.y counts kHC-chunks from h_lo.  h_lo > 0: the stage
// first applies the stopping rule at its lower boundary (best count over hypotheses < h_lo).
__global__ __launch_bounds__(kScoreThreads) void score_kernel(
    const float* __restrict__ p3d, const float* __restrict__ p2d, const int32_t* __restrict__ M_dev, int M_cap,
    const float* __restrict__ Pm, const uint8_t* __restrict__ ok, int H, int h_lo, int h_hi, double one_minus_conf,
    float reperr, int32_t* __restrict__ n_inl) {
  __shared__ int32_t cnt[kHC];
  __shared__ __attribute__((aligned(16))) float Ps[kHC][12];   // the block's projection matrices: one
  __shared__ uint8_t oks[kHC];                                  // coalesced load, then LDS broadcasts
  const int b = blockIdx.z;
  p3d += (size_t)b * M_cap * 3; p2d += (size_t)b * M_cap * 2;
  Pm += (size_t)b * H * 12; ok += (size_t)b * H; n_inl += (size_t)b * H;
  const int M = M_dev[b];
  const int base = blockIdx.x * (kScoreThreads * kCPL);
  if (base >= M) return;  // block-uniform
  if (h_lo > 0) {         // stage gate (block-uniform): counts of the earlier stages are final
    __shared__ int32_t cmax[kScoreThreads / 64];
    int c = 0;
    for (int i = threadIdx.x; i < h_lo; i += kScoreThreads) c = max(c, ok[i] ? n_inl[i] : 0);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) c = max(c, __shfl_xor(c, o, 64));
    if ((threadIdx.x & 63) == 0) cmax[threadIdx.x >> 6] = c;
    __syncthreads();
    c = max(max(cmax[0], cmax[1]), max(cmax[2], cmax[3]));
    if (ransac_stop(c, M, h_lo, one_minus_conf)) return;
  }
  const int h0 = h_lo + blockIdx.y * kHC;
  const int h1 = min(h_hi, h0 + kHC);
  if (threadIdx.x < kHC) {
    cnt[threadIdx.x] = 0;
    oks[threadIdx.x] = (h0 + (int)threadIdx.x < h1) ? ok[h0 + threadIdx.x] : 0;
  }
  for (int i = threadIdx.x; i < (h1 - h0) * 12; i += kScoreThreads) (&Ps[0][0])[i] = Pm[12 * (size_t)h0 + i];
  __syncthreads();
  float X[kCPL], Y[kCPL], Z[kCPL], U[kCPL], V[kCPL];
  bool valid[kCPL];
#pragma unroll
  for (int c = 0; c < kCPL; ++c) {
    const int m = base + c * kScoreThreads + threadIdx.x;
    valid[c] = m < M;
    const int mm = valid[c] ? m : 0;
    X[c] = p3d[3 * (size_t)mm]; Y[c] = p3d[3 * (size_t)mm + 1]; Z[c] = p3d[3 * (size_t)mm + 2];
    U[c] = p2d[2 * (size_t)mm]; V[c] = p2d[2 * (size_t)mm + 1];
  }
  const int lane = threadIdx.x & 63;
  // Everything the loop branches on or counts with lives in scalar registers: the hypotheses' ok flags as one 32-bit
  // mask, the lanes' validity as one ballot per correspondence slot, and each test as the AND of the two compares'
  // own condition masks (a ballot of a bool expression costs a v_cndmask + v_cmp_ne per slot on top of the compares).
  static_assert(kHC <= 32, "ok mask of a block's hypotheses is one 32-bit word");
  const uint32_t okmask = (uint32_t)__ballot(lane < kHC && oks[lane < kHC ? lane : 0] != 0);
  unsigned long long vmask[kCPL];
#pragma unroll
  for (int c = 0; c < kCPL; ++c) vmask[c] = __ballot(valid[c]);
  int cntv = 0;  // lane j: this wave's count for hypothesis h0 + j (one v_writelane per hypothesis, one LDS atomic per wave)
  for (int h = h0; h < h1; ++h) {
    if (!((okmask >> (h - h0)) & 1u)) continue;  // uniform
    // the matrix once per hypothesis (three LDS broadcasts), then straight-line tests
    const float4 r0 = *reinterpret_cast<const float4*>(&Ps[h - h0][0]);
    const float4 r1 = *reinterpret_cast<const float4*>(&Ps[h - h0][4]);
    const float4 r2 = *reinterpret_cast<const float4*>(&Ps[h - h0][8]);
    int c_wave = 0;
#pragma unroll
    for (int c = 0; c < kCPL; ++c) {
      // same operation order as inlier_f32
      const float x = __builtin_fmaf(r0.z, Z[c], __builtin_fmaf(r0.y, Y[c], __builtin_fmaf(r0.x, X[c], r0.w)));
      const float y = __builtin_fmaf(r1.z, Z[c], __builtin_fmaf(r1.y, Y[c], __builtin_fmaf(r1.x, X[c], r1.w)));
      const float z = __builtin_fmaf(r2.z, Z[c], __builtin_fmaf(r2.y, Y[c], __builtin_fmaf(r2.x, X[c], r2.w)));
      const float ex = __builtin_fmaf(-U[c], z, x), ey = __builtin_fmaf(-V[c], z, y);
      const float e2 = __builtin_fmaf(ey, ey, ex * ex);
      const float lim = reperr * z;
      c_wave += __popcll(__ballot(z > 0.0f) & __ballot(e2 <= lim * lim) & vmask[c]);
    }
    // gfx9 allows one SGPR per VALU instruction on the constant bus: the lane select travels in M0, bound as an INPUT operand
    // (the compiler loads it and knows the register is in use — round 4 wrote M0 inside the asm and listed it as a clobber,
    // which hipcc answers with "may not be preserved")
    asm volatile("v_writelane_b32 %0, %1, m0" : "+v"(cntv) : "s"(c_wave), "{m0}"(h - h0));
  }
  if (lane < kHC && cntv) atomicAdd(&cnt[lane], cntv);
  __syncthreads();
  if (threadIdx.x < h1 - h0 && cnt[threadIdx.x]) atomicAdd(&n_inl[h0 + threadIdx.x], cnt[threadIdx.x]);
}

// One block per image: best = arg-max n_inl over ok hypotheses (lowest h on ties); status = count >= 4.
// Also clears the image's Gauss-Newton convergence flag (no memset launch).
// n_eval_dev (nullable): how many hypotheses the staged loop scored for this image — the stage rule replayed
// on the final counts (a stage that did not run left its counts at 0, and the replay stops before it).
__global__ void best_kernel(const int32_t* __restrict__ n_inl, const uint8_t* __restrict__ ok, int H,
                            int32_t* __restrict__ best_dev, int32_t* __restrict__ status_dev,
                            const double* __restrict__ Rt, double* __restrict__ pose_dev,
                            int32_t* __restrict__ gn_state, const int32_t* __restrict__ M_dev,
                            double one_minus_conf, int32_t* __restrict__ n_eval_dev) {
  __shared__ int32_t sc[256], sh[256];
  __shared__ int32_t smax[16];        // best count inside stage s = [32 (2^s - 1), 32 (2^(s+1) - 1))
  const int b = blockIdx.z;
  n_inl += (size_t)b * H; ok += (size_t)b * H; Rt += (size_t)b * H * 12;
  best_dev += b;
  if (status_dev) status_dev += b;
  if (pose_dev) pose_dev += (size_t)b * 12;
  if (gn_state && threadIdx.x == 0) { gn_state[4 * b] = 0; gn_state[4 * b + 1] = 0; }
  if (threadIdx.x < 16) smax[threadIdx.x] = 0;
  if (n_eval_dev) __syncthreads();
  int bc = -1, bh = -1;
  for (int h = threadIdx.x; h < H; h += 256) {
    if (ok[h] && n_inl[h] > bc) { bc = n_inl[h]; bh = h; }  // ascending h per thread: lowest kept
    if (n_eval_dev && ok[h] && n_inl[h] > 0) atomicMax(&smax[31 - __clz(h / kStage0 + 1)], n_inl[h]);
  }
  if (n_eval_dev) {
    __syncthreads();
    if (threadIdx.x == 0) {
      int n_eval = H, c = 0;
      if (one_minus_conf > 0.0) {
        for (int st = 0; st < 16; ++st) {
          c = max(c, smax[st]);
          const int bound = kStage0 * ((2 << st) - 1);     // upper boundary of stage st
          if (bound >= H) break;
          if (ransac_stop(c, M_dev[b], bound, one_minus_conf)) { n_eval = bound; break; }
        }
      }
      n_eval_dev[b] = n_eval;
    }
  }
  sc[threadIdx.x] = bc; sh[threadIdx.x] = bh;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      const int oc = sc[threadIdx.x + off], oh = sh[threadIdx.x + off];
      if (oc > sc[threadIdx.x] || (oc == sc[threadIdx.x] && oh >= 0 && (sh[threadIdx.x] < 0 || oh < sh[threadIdx.x]))) {
        sc[threadIdx.x] = oc; sh[threadIdx.x] = oh;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *best_dev = sh[0];
    if (status_dev) *status_dev = (sh[0] >= 0 && sc[0] >= 4) ? 1 : 0;
  }
  if (pose_dev && threadIdx.x < 12) {
    const int b = sh[0];
    pose_dev[threadIdx.x] = (b >= 0) ? Rt[12 * (size_t)b + threadIdx.x] : ((threadIdx.x % 5 == 0) ? 1.0 : 0.0);
  }
}

// Local-optimisation round: the f32 projection matrix of each image's REFITTED pose (same f64 fma order
// as proj_matrix_kernel) and a fresh Gauss-Newton convergence flag.
__global__ void refined_proj_kernel(const double* __restrict__ pose, const ImgDev* __restrict__ ib, float* __restrict__ Pm,
                                    int32_t* __restrict__ gn_state) {
  const int b = blockIdx.x, e = threadIdx.x;
  const Cam& cam = ib[b].cam;
  const double* T = pose + 12 * (size_t)b;
  if (e < 12) {
    const int r = e / 4, c = e % 4;
    Pm[12 * (size_t)b + e] = (float)fma(cam.k[3 * r + 2], T[8 + c], fma(cam.k[3 * r + 1], T[4 + c], cam.k[3 * r] * T[c]));
  }
  if (e == 0) { gn_state[4 * b] = 0; gn_state[4 * b + 1] = 0; }
}

__global__ void best_mask_kernel(const float* __restrict__ p3d, const float* __restrict__ p2d,
                                 const int32_t* __restrict__ M_dev, int M_cap, int H,
                                 const float* __restrict__ Pm, const int32_t* __restrict__ best_dev,
                                 float reperr, uint32_t* __restrict__ mask, int mask_words) {
  const int img = blockIdx.z;
  p3d += (size_t)img * M_cap * 3; p2d += (size_t)img * M_cap * 2; Pm += (size_t)img * H * 12;
  mask += (size_t)img * mask_words;
  const int m = blockIdx.x * blockDim.x + threadIdx.x;  // blockDim multiple of 64
  const int M = M_dev[img];
  const int b = best_dev ? best_dev[img] : 0;             // no table of hypotheses: Pm holds one matrix per image
  bool in = false;
  if (m < M && b >= 0)
    in = inlier_f32(Pm + 12 * (size_t)b, p3d[3 * (size_t)m], p3d[3 * (size_t)m + 1], p3d[3 * (size_t)m + 2],
                    p2d[2 * (size_t)m], p2d[2 * (size_t)m + 1], reperr);
  const unsigned long long bal = __ballot(in);
  const int lane = threadIdx.x & 63;
  const int w = m >> 5;
  if ((lane == 0 || lane == 32) && w < (M_cap + 31) / 32)
    mask[w] = (lane == 0) ? (uint32_t)bal : (uint32_t)(bal >> 32);
}

// ---------------------------------------------------------------------------------- refit
#ifndef ISR_GN_SPLIT_SOLVE
#define ISR_GN_SPLIT_SOLVE 1      // 0: the solve in gn_accumulate_kernel's last workgroup (round 3 .. first half of round 4)
#endif
constexpr int kRefThreads = 256;
#ifndef ISR_GN_STOP_ROT2
#define ISR_GN_STOP_ROT2 1e-18      // (1e-9 rad)^2
#endif
#ifndef ISR_GN_STOP_TRANS2
#define ISR_GN_STOP_TRANS2 1e-14    // (1e-7 |t|)^2
#endif
constexpr double kGnStopRot2 = ISR_GN_STOP_ROT2, kGnStopTrans2 = ISR_GN_STOP_TRANS2;
#ifndef ISR_REF_BLOCKS
#define ISR_REF_BLOCKS 64
#endif
constexpr int kRefBlocks = ISR_REF_BLOCKS;      // workgroups per image of a Gauss-Newton accumulation
constexpr int kNAcc = 29;  // 21 (upper J^T J) + 6 (J^T r) + 1 (sum r^2) + 1 (number of correspondences used)

// One block: 28 lanes sum the block partials in order (fixed order: reproducible), lane 0 solves
// the 6x6 normal equations by Cholesky and retracts:  R <- Q(w) R,  t <- Q(w) t + dt  with Q(w) the
// rotation of the unit quaternion (1, w/2)/|.| (sqrt only: no sin/cos, so the step is plain IEEE
// arithmetic).  state[0] = 1 once the step is below 1e-12: later launches return at once.
// (gn_solve_kernel, a launch of its own behind gn_accumulate_kernel: see there.)
__device__ void gn_solve(const double* __restrict__ partial, int nblocks, double* __restrict__ Rt, int32_t* __restrict__ state) {
  __shared__ double s[kNAcc];
  if (threadIdx.x < kNAcc) {
    // blocks [nblocks, kRefBlocks) were not launched: their correspondences m = b * 256 + t lie beyond M_cap, their sums
    // would be exact zeros — leaving them out changes no bit of the total
    double a = 0.0;
    for (int b = 0; b < nblocks; ++b) a += partial[(size_t)b * kNAcc + threadIdx.x];
    s[threadIdx.x] = a;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  // fewer than 4 correspondences (a local-optimisation mask that collapsed): the system is rank deficient or
  // nearly so — keep the pose this refit started from (oracle/pnp_oracle.py:refine does the same)
  if (s[28] < 4.0) { state[0] = 1; return; }
  double A[6][6], g[6];
  int k = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) { A[i][j] = s[k]; A[j][i] = s[k]; ++k; }
  for (int i = 0; i < 6; ++i) g[i] = -s[21 + i];
  // Cholesky A = L L^T (tiny relative damping keeps a rank-deficient system finite)
  double tr = 0.0;
  for (int i = 0; i < 6; ++i) tr += A[i][i];
  for (int i = 0; i < 6; ++i) A[i][i] += 1e-14 * tr;
  double L[6][6];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) L[i][j] = 0.0;
  for (int j = 0; j < 6; ++j) {
    double d = A[j][j];
    for (int p = 0; p < j; ++p) d -= L[j][p] * L[j][p];
    if (!(d > 0.0)) { state[0] = 1; return; }  // not positive definite: keep the pose
    L[j][j] = sqrt(d);
    for (int i = j + 1; i < 6; ++i) {
      double v = A[i][j];
      for (int p = 0; p < j; ++p) v -= L[i][p] * L[j][p];
      L[i][j] = v / L[j][j];
    }
  }
  double y[6], x[6];
  for (int i = 0; i < 6; ++i) {
    double v = g[i];
    for (int p = 0; p < i; ++p) v -= L[i][p] * y[p];
    y[i] = v / L[i][i];
  }
  for (int i = 5; i >= 0; --i) {
    double v = y[i];
    for (int p = i + 1; p < 6; ++p) v -= L[p][i] * x[p];
    x[i] = v / L[i][i];
  }
  // quaternion (1, w/2) normalised
  const double hx = 0.5 * x[0], hy = 0.5 * x[1], hz = 0.5 * x[2];
  const double nrm = 1.0 / sqrt(1.0 + hx * hx + hy * hy + hz * hz);
  const double qw = nrm, qx = hx * nrm, qy = hy * nrm, qz = hz * nrm;
  const double Q[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw),
                       2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw),
                       2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)};
  double T[12], O[12];
  for (int i = 0; i < 12; ++i) T[i] = Rt[i];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 4; ++c) O[4 * r + c] = Q[3 * r] * T[c] + Q[3 * r + 1] * T[4 + c] + Q[3 * r + 2] * T[8 + c];
    O[4 * r + 3] += x[3 + r];
  }
  for (int i = 0; i < 12; ++i) Rt[i] = O[i];
  // converged: the step just APPLIED was below kGnStopRot rad and kGnStopTrans of |t| (squared: dw, dt).  Gauss-Newton on
  // this problem converges quadratically near the solution, so the step after one of 1e-9 would be ~1e-18: the pose is final to
  // double precision already, and the remaining launches of the fixed iteration count leave at once.  (Until round 5 the bounds
  // were 1e-12 rad / 1e-10: one more working iteration per refit for nothing.)  oracle/pnp_oracle.py:refine applies the same
  // rule with the same numbers.
  const double dw = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
  const double dt = x[3] * x[3] + x[4] * x[4] + x[5] * x[5];
  const double tt = O[3] * O[3] + O[7] * O[7] + O[11] * O[11];
  if (dw < kGnStopRot2 && dt < kGnStopTrans2 * (tt + 1.0)) state[0] = 1;
}


// One correspondence's terms of the Gauss-Newton normal equations (both refit routes: the same operations, the same bits).
// Explicit fused multiply-adds: the library is built with -ffp-contract=off, and as plain `a * b + c * d` expressions these
// 60 product-sums were 2 multiplications + 2 additions each — the refit is bound by f64 issue (8 192 waves x ~15
// correspondences per launch, 24 launches per bench step) and ran ~210 f64 instructions per correspondence; now ~115.
__device__ __forceinline__ void gn_point(const double (&T)[12], const Cam& cam, double X, double Y, double Z, double ox, double oy,
                                         double (&acc)[kNAcc]) {
  const double xc = __builtin_fma(T[0], X, __builtin_fma(T[1], Y, __builtin_fma(T[2], Z, T[3])));
  const double yc = __builtin_fma(T[4], X, __builtin_fma(T[5], Y, __builtin_fma(T[6], Z, T[7])));
  const double zc = __builtin_fma(T[8], X, __builtin_fma(T[9], Y, __builtin_fma(T[10], Z, T[11])));
  const double px = __builtin_fma(cam.k[0], xc, __builtin_fma(cam.k[1], yc, cam.k[2] * zc));
  const double py = __builtin_fma(cam.k[3], xc, __builtin_fma(cam.k[4], yc, cam.k[5] * zc));
  const double pz = __builtin_fma(cam.k[6], xc, __builtin_fma(cam.k[7], yc, cam.k[8] * zc));
  const double ipz = 1.0 / pz;
  const double u = px * ipz, v = py * ipz;
  const double ru = u - ox, rv = v - oy;
  // d(u,v)/dXc = (K_row - (u,v) K_row2) / pz
  const double a0 = __builtin_fma(-u, cam.k[6], cam.k[0]) * ipz, a1 = __builtin_fma(-u, cam.k[7], cam.k[1]) * ipz,
               a2 = __builtin_fma(-u, cam.k[8], cam.k[2]) * ipz;
  const double b0 = __builtin_fma(-v, cam.k[6], cam.k[3]) * ipz, b1 = __builtin_fma(-v, cam.k[7], cam.k[4]) * ipz,
               b2 = __builtin_fma(-v, cam.k[8], cam.k[5]) * ipz;
  // Xc' = Xc + w x Xc + dt  ->  dXc/dw = -[Xc]x, dXc/dt = I
  const double Ju[6] = {__builtin_fma(a2, yc, -(a1 * zc)), __builtin_fma(a0, zc, -(a2 * xc)), __builtin_fma(a1, xc, -(a0 * yc)), a0, a1, a2};
  const double Jv[6] = {__builtin_fma(b2, yc, -(b1 * zc)), __builtin_fma(b0, zc, -(b2 * xc)), __builtin_fma(b1, xc, -(b0 * yc)), b0, b1, b2};
  int k = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = i; j < 6; ++j) { acc[k] = __builtin_fma(Ju[i], Ju[j], __builtin_fma(Jv[i], Jv[j], acc[k])); ++k; }
#pragma unroll
  for (int i = 0; i < 6; ++i) acc[21 + i] = __builtin_fma(Ju[i], ru, __builtin_fma(Jv[i], rv, acc[21 + i]));
  acc[27] = __builtin_fma(ru, ru, __builtin_fma(rv, rv, acc[27]));
  acc[28] += 1.0;
}

__global__ __launch_bounds__(kRefThreads) void gn_accumulate_kernel(
    const float* __restrict__ p3d, const float* __restrict__ p2d, const int32_t* __restrict__ M_dev, int M_cap,
    const uint32_t* __restrict__ mask, int mask_words, const ImgDev* __restrict__ ib, const double* __restrict__ Rt,
    const int32_t* __restrict__ status_dev, int32_t* __restrict__ state,
    double* __restrict__ partial) {
  __shared__ double red[kRefThreads / 64][kNAcc];
  __shared__ int last;
  const int img = blockIdx.z;
  const Cam& cam = ib[img].cam;
  p3d += (size_t)img * M_cap * 3; p2d += (size_t)img * M_cap * 2;
  if (mask) mask += (size_t)img * mask_words;
  Rt += (size_t)img * 12; state += 4 * img; partial += (size_t)img * kRefBlocks * kNAcc;
  if (state[0]) return;  // converged (block-uniform)
  if (status_dev && status_dev[img] == 0) return;   // no pose: nothing to refit (block-uniform)
  double acc[kNAcc];
#pragma unroll
  for (int i = 0; i < kNAcc; ++i) acc[i] = 0.0;
  const int M = M_dev[img];
  const bool live = (status_dev == nullptr) || (status_dev[img] != 0);
  if (live) {
    double T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = Rt[i];
    for (int m = blockIdx.x * kRefThreads + threadIdx.x; m < M; m += kRefBlocks * kRefThreads) {
      if (mask && !((mask[m >> 5] >> (m & 31)) & 1u)) continue;
      gn_point(T, cam, p3d[3 * (size_t)m], p3d[3 * (size_t)m + 1], p3d[3 * (size_t)m + 2], p2d[2 * (size_t)m], p2d[2 * (size_t)m + 1], acc);
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < kNAcc; ++i) {
    double s = acc[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) red[wave][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < kNAcc)
    partial[(size_t)blockIdx.x * kNAcc + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
#if ISR_GN_SPLIT_SOLVE
  (void)last;
}

// The solve of one Gauss-Newton step, one workgroup per image, in a launch of its own behind gn_accumulate_kernel.  Until
// the second half of round 4 the LAST workgroup of gn_accumulate_kernel solved (a ticket behind __threadfence()): on gfx950
// an agent-scope fence is `buffer_wbl2 sc1` + `buffer_inv sc1` — a write-back and an invalidation of the XCD's L2 — in
// every wave of every one of the 2 048 workgroups of every launch, beside a K1 whose keys live in that L2.  The kernel
// boundary orders the partial sums for free.
__global__ __launch_bounds__(64) void gn_solve_kernel(const double* __restrict__ partial, int nblocks, double* __restrict__ Rt,
                                                       const int32_t* __restrict__ status_dev, int32_t* __restrict__ state) {
  const int img = blockIdx.x;
  state += 4 * img;
  if (state[0]) return;                                  // converged
  if (status_dev && status_dev[img] == 0) return;        // no pose: nothing was accumulated
  gn_solve(partial + (size_t)img * kRefBlocks * kNAcc, nblocks, Rt + (size_t)img * 12, state);
}
#else
  // the last workgroup to deliver its sums (agent-scope fences order them before the ticket) solves
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = (atomicAdd(&state[1], 1) == (int)gridDim.x - 1);
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (threadIdx.x == 0) state[1] = 0;
  gn_solve(partial, (int)gridDim.x, const_cast<double*>(Rt), state);
}
#endif

// ------------------------------------------------------------------- refit, small correspondence sets
// The whole refit of an image — Gauss-Newton over the RANSAC inliers, the local-optimisation round (inliers of the
// refitted pose, refit again) and the final mask under the returned pose — by ONE workgroup in ONE launch, for
// capacities up to kSmallM (the reference's 75 x 75 crops: <= 5 625 correspondences).  The multi-launch route
// (12 gn_accumulate + 2 refined_proj + 2 best_mask launches per group) spends, at that size, 86 us per launch on
// fences, tickets and a serial solve; here the pose and the convergence flag live in LDS, thread t owns the
// correspondences m = t + 256 i, the 29 sums go through the same shuffle tree + wave order as a block of the
// multi-launch route (capacity-independent: a crop registered alone or in a group gives the same bits), and
// gn_solve runs on the LDS copy.  Same per-correspondence arithmetic as gn_accumulate_kernel / best_mask_kernel.
constexpr int kSmallM = 8192;

__global__ __launch_bounds__(kRefThreads) void gn_small_kernel(
    const float* __restrict__ p3d, const float* __restrict__ p2d, const int32_t* __restrict__ M_dev, int M_cap,
    uint32_t* __restrict__ mask, int mask_words, const ImgDev* __restrict__ ib, int iters, float reperr,
    double* __restrict__ pose, const int32_t* __restrict__ status_dev) {
  __shared__ double red[kRefThreads / 64][kNAcc];
  __shared__ double part[kNAcc];
  __shared__ double Ts[12];
  __shared__ float Pm[12];
  __shared__ int32_t st[2];
  const int img = blockIdx.x, tid = threadIdx.x;
  if (status_dev && status_dev[img] == 0) return;          // no pose: nothing to refit (block-uniform)
  const Cam& cam = ib[img].cam;
  p3d += (size_t)img * M_cap * 3; p2d += (size_t)img * M_cap * 2;
  mask += (size_t)img * mask_words; pose += (size_t)img * 12;
  const int M = M_dev[img];
  if (tid < 12) Ts[tid] = pose[tid];
  __syncthreads();
  // the inlier mask of the pose in Ts (best_mask_kernel with one matrix per image)
  auto remask = [&]() {
    if (tid < 12) {
      const int r = tid / 4, c = tid % 4;
      Pm[tid] = (float)fma(cam.k[3 * r + 2], Ts[8 + c], fma(cam.k[3 * r + 1], Ts[4 + c], cam.k[3 * r] * Ts[c]));
    }
    __syncthreads();
    for (int m0 = 0; m0 < M_cap; m0 += kRefThreads) {
      const int m = m0 + tid;
      bool in = false;
      if (m < M)
        in = inlier_f32(Pm, p3d[3 * (size_t)m], p3d[3 * (size_t)m + 1], p3d[3 * (size_t)m + 2], p2d[2 * (size_t)m],
                        p2d[2 * (size_t)m + 1], reperr);
      const unsigned long long bal = __ballot(in);
      const int lane = tid & 63, w = m >> 5;
      if ((lane == 0 || lane == 32) && w < (M_cap + 31) / 32) mask[w] = (lane == 0) ? (uint32_t)bal : (uint32_t)(bal >> 32);
    }
    __syncthreads();
  };
  for (int round = 0; round < 2; ++round) {
    if (round == 1) remask();                               // local optimisation: the inliers of the refitted pose
    if (tid == 0) { st[0] = 0; st[1] = 0; }
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
      if (st[0]) break;                                     // converged (block-uniform: LDS flag read after a barrier)
      double acc[kNAcc];
#pragma unroll
      for (int i = 0; i < kNAcc; ++i) acc[i] = 0.0;
      double T[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) T[i] = Ts[i];
      for (int m = tid; m < M; m += kRefThreads) {
        if (!((mask[m >> 5] >> (m & 31)) & 1u)) continue;
        gn_point(T, cam, p3d[3 * (size_t)m], p3d[3 * (size_t)m + 1], p3d[3 * (size_t)m + 2], p2d[2 * (size_t)m], p2d[2 * (size_t)m + 1], acc);
      }
      const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
      for (int i = 0; i < kNAcc; ++i) {
        double v = acc[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[wave][i] = v;
      }
      __syncthreads();
      if (tid < kNAcc) part[tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
      __syncthreads();
      gn_solve(part, 1, Ts, st);                            // thread 0 solves on the LDS pose; ends with thread 0 only
      __syncthreads();
    }
    __syncthreads();
  }
  remask();                                                 // the reported inliers are those of the returned pose
  if (tid < 12) pose[tid] = Ts[tid];
}

// ------------------------------------------------------------------------------ compaction
// Bitmask -> ascending indices: per-block popcounts, a one-block scan of the block totals, then
// every word scatters its set bits at its exclusive prefix.  kCompBlock words per block.
constexpr int kCompBlock = 256;

__device__ __forceinline__ int block_scan_excl(int v, int* total) {
  __shared__ int wsum[kCompBlock / 64];
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
  for (int w = 0; w < kCompBlock / 64; ++w) {
    if (w < wave) base += wsum[w];
    tot += wsum[w];
  }
  *total = tot;
  return base + inc - v;
}

__global__ __launch_bounds__(kCompBlock) void mask_count_kernel(const uint32_t* __restrict__ mask, int mask_words,
                                                                const int32_t* __restrict__ M_dev,
                                                                const int32_t* __restrict__ status_dev,
                                                                int32_t* __restrict__ block_counts) {
  const int img = blockIdx.z;
  mask += (size_t)img * mask_words; block_counts += (size_t)img * gridDim.x;
  const int W = (status_dev && status_dev[img] == 0) ? 0 : (M_dev[img] + 31) / 32;
  const int w = blockIdx.x * kCompBlock + threadIdx.x;
  int total;
  (void)block_scan_excl(w < W ? __popc(mask[w]) : 0, &total);
  if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}

__global__ void mask_scan_kernel(int32_t* __restrict__ block_counts, int nblocks, int32_t* __restrict__ n_out) {
  __shared__ int32_t tsum[1024];
  block_counts += (size_t)blockIdx.z * nblocks; n_out += blockIdx.z;
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
  if (t == 1023) *n_out = tsum[1023];
}

__global__ __launch_bounds__(kCompBlock) void mask_scatter_kernel(const uint32_t* __restrict__ mask, int mask_words,
                                                                  const int32_t* __restrict__ M_dev,
                                                                  const int32_t* __restrict__ status_dev,
                                                                  const int32_t* __restrict__ block_off,
                                                                  int32_t* __restrict__ out, int M_cap) {
  const int img = blockIdx.z;
  mask += (size_t)img * mask_words; block_off += (size_t)img * gridDim.x; out += (size_t)img * M_cap;
  const int W = (status_dev && status_dev[img] == 0) ? 0 : (M_dev[img] + 31) / 32;
  const int w = blockIdx.x * kCompBlock + threadIdx.x;
  uint32_t bits = w < W ? mask[w] : 0u;
  int total;
  int o = block_off[blockIdx.x] + block_scan_excl(__popc(bits), &total);
  while (bits) {
    out[o++] = w * 32 + (__ffs(bits) - 1);
    bits &= bits - 1;
  }
}

inline int mask_words_of(int M_cap) { return (M_cap + 31) / 32 + 2; }
inline int comp_blocks_of(int M_cap) { return ((M_cap + 31) / 32 + kCompBlock - 1) / kCompBlock; }

struct RansacWs {
  double* Rt;        // B x H x 12
  float* Pm;         // B x H x 12
  uint8_t* ok;       // B x H
  int32_t* n_inl;    // B x H
  int32_t* best;     // B
  uint32_t* mask;    // B x mask_words
  double* partial;   // B x kRefBlocks x kNAcc
  int32_t* state;    // B x 4: GN convergence flag
  int32_t* cblocks;  // B x comp_blocks: compaction block counts
  ImgDev* imgs;      // B: cameras and seeds of the chain's images
};

bool make_batch(const double* Kcams, const uint64_t* seeds, int B, ImgBatch* ib) {
  for (int b = 0; b < B; ++b) {
    if (!make_cam(Kcams + 9 * (size_t)b, &ib->cam[b])) return false;
    const uint64_t s = seeds ? seeds[b] : 0;
    ib->seed_lo[b] = (uint32_t)s;
    ib->seed_hi[b] = (uint32_t)(s >> 32);
  }
  for (int b = B; b < kMaxBatch; ++b) { ib->cam[b] = ib->cam[0]; ib->seed_lo[b] = ib->seed_hi[b] = 0; }
  return true;
}

size_t carve(isr::Workspace& w, int M_cap, int H, int B, RansacWs* o) {
  o->Rt = w.take<double>((size_t)B * H * 12);
  o->Pm = w.take<float>((size_t)B * H * 12);
  o->ok = w.take<uint8_t>((size_t)B * H);
  o->n_inl = w.take<int32_t>((size_t)B * H);
  o->best = w.take<int32_t>(B + 4);
  o->mask = w.take<uint32_t>((size_t)B * mask_words_of(M_cap));
  o->partial = w.take<double>((size_t)B * kRefBlocks * kNAcc);
  o->state = w.take<int32_t>((size_t)B * 4);
  o->cblocks = w.take<int32_t>((size_t)B * (comp_blocks_of(M_cap) + 1));
  o->imgs = w.take<ImgDev>(B);
  return w.off;
}

// cameras / seeds of `B` images (host) -> the chain's device array, kMaxBatch per launch
int upload_imgs(const double* Kcams, const uint64_t* seeds, int B, ImgDev* dst, hipStream_t stream, const char* who) {
  for (int b0 = 0; b0 < B; b0 += kMaxBatch) {
    const int nb = (B - b0 < kMaxBatch) ? B - b0 : kMaxBatch;
    ImgBatch ib;
    if (!make_batch(Kcams + 9 * (size_t)b0, seeds ? seeds + b0 : nullptr, nb, &ib)) {
      isr::set_error("%s: singular camera matrix", who);
      return ISR_ERR_ARG;
    }
    set_imgs_kernel<<<1, kMaxBatch, 0, stream>>>(ib, nb, dst + b0);
  }
  return ISR_OK;
}


}  // namespace

extern "C" size_t isr_pnp_ransac_workspace_bytes(int M_cap, int H) {
  if (M_cap <= 0 || H <= 0) return 0;
  isr::Workspace w(nullptr, 0);
  RansacWs o;
  return carve(w, M_cap, H, 1, &o) + 512;
}

extern "C" size_t isr_pnp_ransac_batch_workspace_bytes(int M_cap, int H, int B) {
  if (M_cap <= 0 || H <= 0 || B <= 0) return 0;
  isr::Workspace w(nullptr, 0);
  RansacWs o;
  return carve(w, M_cap, H, B < kChainMax ? B : kChainMax, &o) + 512;
}

extern "C" int isr_p3p_hypotheses(const float* p3d, const float* p2d, const int32_t* M_dev, int M_cap,
                                  const double* Kcam, int H, uint64_t seed, double* Rt, uint8_t* ok,
                                  int32_t* sample, isr_stream_t stream) {
  ISR_REQUIRE(p3d && p2d && M_dev && Kcam && Rt && ok, "isr_p3p_hypotheses: null pointer");
  ISR_REQUIRE(M_cap > 0 && H > 0, "isr_p3p_hypotheses: M_cap=%d H=%d", M_cap, H);
  ImgDev img;
  ISR_REQUIRE(make_cam(Kcam, &img.cam), "isr_p3p_hypotheses: singular camera matrix");
  img.seed_lo = (uint32_t)seed;
  img.seed_hi = (uint32_t)(seed >> 32);
  p3p_single_kernel<<<dim3((H + 63) / 64, 1, 1), 64, 0, isr::as_stream(stream)>>>(p3d, p2d, M_dev, M_cap, img, H, Rt, ok, sample);
  ISR_CHECK_LAUNCH("p3p_kernel");
  return ISR_OK;
}

extern "C" int isr_p3p_all_roots(const double* X, const double* uv, const double* Kcam, int S, double* poses,
                                 int32_t* n_roots, isr_stream_t stream) {
  ISR_REQUIRE(X && uv && Kcam && poses && n_roots && S > 0, "isr_p3p_all_roots: bad argument");
  Cam cam;
  ISR_REQUIRE(make_cam(Kcam, &cam), "isr_p3p_all_roots: singular camera matrix");
  p3p_all_roots_kernel<<<(S + 63) / 64, 64, 0, isr::as_stream(stream)>>>(X, uv, cam, S, poses, n_roots);
  ISR_CHECK_LAUNCH("p3p_all_roots_kernel");
  return ISR_OK;
}

static int score_impl(const float* p3d, const float* p2d, const int32_t* M_dev, int M_cap, int B, const ImgDev* ib,
                      const double* Rt, const uint8_t* ok, int H, double confidence, float reperr, float* Pm,
                      int32_t* n_inl, int32_t* best_dev, uint32_t* best_mask, int32_t* status_dev, double* pose_dev,
                      int32_t* gn_state, int32_t* n_eval_dev, hipStream_t stream) {
  proj_matrix_kernel<<<dim3((H * 12 + 255) / 256, 1, B), 256, 0, stream>>>(Rt, ib, H, Pm, n_inl);
  const int nblk = (M_cap + kScoreThreads * kCPL - 1) / (kScoreThreads * kCPL);
  const double omc = (confidence >= 1.0) ? 0.0 : 1.0 - confidence;
  // stages [0,32), [32,96), [96,224), ...: boundaries 32 (2^k - 1); one stage when every hypothesis is wanted
  for (int lo = 0, len = (omc > 0.0) ? kStage0 : H; lo < H; lo += len, len *= 2) {
    const int hi = (lo + len < H) ? lo + len : H;
    score_kernel<<<dim3(nblk, (hi - lo + kHC - 1) / kHC, B), kScoreThreads, 0, stream>>>(
        p3d, p2d, M_dev, M_cap, Pm, ok, H, lo, hi, omc, reperr, n_inl);
    if (hi == H) break;
  }
  best_kernel<<<dim3(1, 1, B), 256, 0, stream>>>(n_inl, ok, H, best_dev, status_dev, Rt, pose_dev, gn_state, M_dev, omc,
                                                 n_eval_dev);
  if (best_mask)
    best_mask_kernel<<<dim3((M_cap + 255) / 256, 1, B), 256, 0, stream>>>(p3d, p2d, M_dev, M_cap, H, Pm, best_dev, reperr,
                                                                          best_mask, mask_words_of(M_cap));
  ISR_CHECK_LAUNCH("ransac score kernels");
  return ISR_OK;
}

extern "C" int isr_ransac_score(const float* p3d, const float* p2d, const int32_t* M_dev, int M_cap,
                                const double* Kcam, const double* Rt, const uint8_t* ok, int H,
                                float reperr, int32_t* n_inl, int32_t* best_dev, uint32_t* best_mask,
                                void* ws, size_t ws_bytes, isr_stream_t stream_) {
  ISR_REQUIRE(p3d && p2d && M_dev && Kcam && Rt && ok && n_inl && best_dev, "isr_ransac_score: null pointer");
  ISR_REQUIRE(M_cap > 0 && H > 0 && H <= kMaxH, "isr_ransac_score: M_cap=%d H=%d (H <= %d)", M_cap, H, kMaxH);
  if (!ws || ws_bytes < isr_pnp_ransac_workspace_bytes(M_cap, H)) {
    isr::set_error("isr_ransac_score: workspace %zu < %zu", ws_bytes, isr_pnp_ransac_workspace_bytes(M_cap, H));
    return ISR_ERR_WORKSPACE;
  }
  isr::Workspace w(ws, ws_bytes);
  float* Pm = w.take<float>((size_t)H * 12);
  ImgDev* imgs = w.take<ImgDev>(1);
  const int rc = upload_imgs(Kcam, nullptr, 1, imgs, isr::as_stream(stream_), "isr_ransac_score");
  if (rc != ISR_OK) return rc;
  return score_impl(p3d, p2d, M_dev, M_cap, 1, imgs, Rt, ok, H, 1.0, reperr, Pm, n_inl, best_dev, best_mask,
                    nullptr, nullptr, nullptr, nullptr, isr::as_stream(stream_));
}

static int refine_impl(const float* p3d, const float* p2d, const int32_t* M_dev, int M_cap, int B, const uint32_t* mask,
                       const ImgDev* ib, int iters, double* Rt_io, const int32_t* status_dev, double* partial,
                       int32_t* state, hipStream_t stream) {
  // block b owns the correspondences m = b * 256 + t (+ multiples of kRefBlocks * 256): only blocks that can own one
  // are launched (75 x 75 crops: 22 of 64), which leaves every sum — a fixed tree per block, blocks in order — unchanged
  const int nblocks = (M_cap + kRefThreads - 1) / kRefThreads < kRefBlocks ? (M_cap + kRefThreads - 1) / kRefThreads : kRefBlocks;
  for (int it = 0; it < iters; ++it) {
    gn_accumulate_kernel<<<dim3(nblocks, 1, B), kRefThreads, 0, stream>>>(p3d, p2d, M_dev, M_cap, mask, mask_words_of(M_cap),
                                                                             ib, Rt_io, status_dev, state, partial);
#if ISR_GN_SPLIT_SOLVE
    gn_solve_kernel<<<B, 64, 0, stream>>>(partial, nblocks, Rt_io, status_dev, state);
#endif
  }
  ISR_CHECK_LAUNCH("pnp refine kernels");
  return ISR_OK;
}

extern "C" int isr_pnp_refine(const float* p3d, const float* p2d, const int32_t* M_dev, int M_cap,
                              const uint32_t* mask, const double* Kcam, int iters, double* Rt_io, void* ws,
                              size_t ws_bytes, isr_stream_t stream) {
  ISR_REQUIRE(p3d && p2d && M_dev && Kcam && Rt_io, "isr_pnp_refine: null pointer");
  ISR_REQUIRE(M_cap > 0 && iters >= 0, "isr_pnp_refine: M_cap=%d iters=%d", M_cap, iters);
  const size_t need = sizeof(double) * kRefBlocks * kNAcc + sizeof(ImgDev) + 1024;
  if (!ws || ws_bytes < need) {
    isr::set_error("isr_pnp_refine: workspace %zu < %zu", ws_bytes, need);
    return ISR_ERR_WORKSPACE;
  }
  isr::Workspace w(ws, ws_bytes);
  double* partial = w.take<double>((size_t)kRefBlocks * kNAcc);
  int32_t* state = w.take<int32_t>(4);
  ImgDev* imgs = w.take<ImgDev>(1);
  const int rc = upload_imgs(Kcam, nullptr, 1, imgs, isr::as_stream(stream), "isr_pnp_refine");
  if (rc != ISR_OK) return rc;
  ISR_CHECK_HIP(hipMemsetAsync(state, 0, 4 * sizeof(int32_t), isr::as_stream(stream)));
  return refine_impl(p3d, p2d, M_dev, M_cap, 1, mask, imgs, iters, Rt_io, nullptr, partial, state, isr::as_stream(stream));
}

// the chain for B <= kMaxBatch images: hypotheses, scoring, best + mask, refit, compaction
static int ransac_chain(const float* p3d, const float* p2d, const int32_t* M_dev, int M_cap, int B, const ImgDev* ib,
                        int H, double confidence, float reperr, int refine_iters, double* pose_dev, int32_t* inl_idx,
                        int32_t* n_inl_dev, int32_t* status_dev, int32_t* n_eval_dev, const RansacWs& b, hipStream_t stream) {
  p3p_kernel<<<dim3((H + 63) / 64, 1, B), 64, 0, stream>>>(p3d, p2d, M_dev, M_cap, ib, H, b.Rt, b.ok, nullptr);
  ISR_CHECK_LAUNCH("p3p_kernel");
  int rc = score_impl(p3d, p2d, M_dev, M_cap, B, ib, b.Rt, b.ok, H, confidence, reperr, b.Pm, b.n_inl, b.best, b.mask,
                      status_dev, pose_dev, b.state, n_eval_dev, stream);
  if (rc != ISR_OK) return rc;
  if (refine_iters > 0 && M_cap <= kSmallM) {
    // small correspondence sets: both refit rounds and the final mask in one launch, one workgroup per image
    gn_small_kernel<<<B, kRefThreads, 0, stream>>>(p3d, p2d, M_dev, M_cap, b.mask, mask_words_of(M_cap), ib, refine_iters, reperr,
                                                   pose_dev, status_dev);
    ISR_CHECK_LAUNCH("gn_small_kernel");
  } else {
  rc = refine_impl(p3d, p2d, M_dev, M_cap, B, b.mask, ib, refine_iters, pose_dev, status_dev, b.partial, b.state, stream);
  if (rc != ISR_OK) return rc;
  if (refine_iters > 0) {   // local optimisation: inliers of the refitted pose, refit on them (b.Pm's first B rows are free now)
    refined_proj_kernel<<<B, 64, 0, stream>>>(pose_dev, ib, b.Pm, b.state);
    best_mask_kernel<<<dim3((M_cap + 255) / 256, 1, B), 256, 0, stream>>>(p3d, p2d, M_dev, M_cap, 1, b.Pm, nullptr, reperr,
                                                                          b.mask, mask_words_of(M_cap));
    rc = refine_impl(p3d, p2d, M_dev, M_cap, B, b.mask, ib, refine_iters, pose_dev, status_dev, b.partial, b.state, stream);
    if (rc != ISR_OK) return rc;
    // the reported inliers are those of the RETURNED pose: the mask once more, under the final refit
    refined_proj_kernel<<<B, 64, 0, stream>>>(pose_dev, ib, b.Pm, b.state);
    best_mask_kernel<<<dim3((M_cap + 255) / 256, 1, B), 256, 0, stream>>>(p3d, p2d, M_dev, M_cap, 1, b.Pm, nullptr, reperr,
                                                                          b.mask, mask_words_of(M_cap));
  }
  }
  const int cb = comp_blocks_of(M_cap), mw = mask_words_of(M_cap);
  mask_count_kernel<<<dim3(cb, 1, B), kCompBlock, 0, stream>>>(b.mask, mw, M_dev, status_dev, b.cblocks);
  mask_scan_kernel<<<dim3(1, 1, B), 1024, 0, stream>>>(b.cblocks, cb, n_inl_dev);
  mask_scatter_kernel<<<dim3(cb, 1, B), kCompBlock, 0, stream>>>(b.mask, mw, M_dev, status_dev, b.cblocks, inl_idx, M_cap);
  ISR_CHECK_LAUNCH("mask compaction kernels");
  return ISR_OK;
}

extern "C" int isr_pnp_ransac(const float* p3d, const float* p2d, const int32_t* M_dev, int M_cap,
                              const double* Kcam, int H, uint64_t seed, float reperr, double confidence,
                              int refine_iters, double* pose_dev, int32_t* inl_idx, int32_t* n_inl_dev,
                              int32_t* status_dev, int32_t* n_eval_dev, void* ws, size_t ws_bytes, isr_stream_t stream_) {
  ISR_REQUIRE(p3d && p2d && M_dev && Kcam && pose_dev && inl_idx && n_inl_dev && status_dev,
              "isr_pnp_ransac: null pointer");
  ISR_REQUIRE(M_cap > 0 && H > 0 && H <= kMaxH, "isr_pnp_ransac: M_cap=%d H=%d (H <= %d)", M_cap, H, kMaxH);
  if (!ws || ws_bytes < isr_pnp_ransac_workspace_bytes(M_cap, H)) {
    isr::set_error("isr_pnp_ransac: workspace %zu < %zu", ws_bytes, isr_pnp_ransac_workspace_bytes(M_cap, H));
    return ISR_ERR_WORKSPACE;
  }
  isr::Workspace w(ws, ws_bytes);
  RansacWs b;
  carve(w, M_cap, H, 1, &b);
  ISR_REQUIRE(confidence > 0.0, "isr_pnp_ransac: confidence=%g must be > 0 (>= 1: score every hypothesis)", confidence);
  const int urc = upload_imgs(Kcam, &seed, 1, b.imgs, isr::as_stream(stream_), "isr_pnp_ransac");
  if (urc != ISR_OK) return urc;
  return ransac_chain(p3d, p2d, M_dev, M_cap, 1, b.imgs, H, confidence, reperr, refine_iters, pose_dev, inl_idx, n_inl_dev,
                      status_dev, n_eval_dev, b, isr::as_stream(stream_));
}

extern "C" int isr_pnp_ransac_batch(const float* p3d, const float* p2d, const int32_t* M_dev, int M_cap, int B,
                                    const double* Kcams, int H, const uint64_t* seeds, float reperr, double confidence,
                                    int refine_iters, double* pose_dev, int32_t* inl_idx, int32_t* n_inl_dev,
                                    int32_t* status_dev, int32_t* n_eval_dev, void* ws, size_t ws_bytes,
                                    isr_stream_t stream_) {
  ISR_REQUIRE(p3d && p2d && M_dev && Kcams && seeds && pose_dev && inl_idx && n_inl_dev && status_dev,
              "isr_pnp_ransac_batch: null pointer");
  ISR_REQUIRE(M_cap > 0 && H > 0 && H <= kMaxH && B > 0, "isr_pnp_ransac_batch: M_cap=%d H=%d (H <= %d) B=%d", M_cap, H, kMaxH, B);
  ISR_REQUIRE(confidence > 0.0, "isr_pnp_ransac_batch: confidence=%g must be > 0 (>= 1: score every hypothesis)", confidence);
  if (!ws || ws_bytes < isr_pnp_ransac_batch_workspace_bytes(M_cap, H, B)) {
    isr::set_error("isr_pnp_ransac_batch: workspace %zu < %zu", ws_bytes, isr_pnp_ransac_batch_workspace_bytes(M_cap, H, B));
    return ISR_ERR_WORKSPACE;
  }
  hipStream_t stream = isr::as_stream(stream_);
  for (int b0 = 0; b0 < B; b0 += kChainMax) {        // one chain of launches per kChainMax images
    const int nb = (B - b0 < kChainMax) ? B - b0 : kChainMax;
    isr::Workspace w(ws, ws_bytes);               // chunks run one after the other on the stream: same scratch
    RansacWs wsb;
    carve(w, M_cap, H, nb, &wsb);
    const int urc = upload_imgs(Kcams + 9 * (size_t)b0, seeds + b0, nb, wsb.imgs, stream, "isr_pnp_ransac_batch");
    if (urc != ISR_OK) return urc;
    const int rc = ransac_chain(p3d + (size_t)b0 * M_cap * 3, p2d + (size_t)b0 * M_cap * 2, M_dev + b0, M_cap, nb, wsb.imgs, H,
                                confidence, reperr, refine_iters, pose_dev + (size_t)b0 * 12, inl_idx + (size_t)b0 * M_cap,
                                n_inl_dev + b0, status_dev + b0, n_eval_dev ? n_eval_dev + b0 : nullptr, wsb, stream);
    if (rc != ISR_OK) return rc;
  }
  return ISR_OK;
}
