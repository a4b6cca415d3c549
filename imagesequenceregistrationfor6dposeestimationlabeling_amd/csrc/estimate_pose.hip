// estimate_pose.hip — device stages of estimate_pose(), the SurfEmb-style sample-and-score pose
// estimator (poseEstSurf.py:11-261; a6/a7 of SURVEY.md §8).
//
//   isr_ep_prepare      poseEstSurf.py:47-69   logsigmoid / max-pool / avg-pool of the mask logits,
//                                              avg-pooled queries
//   isr_ep_pool_corr    poseEstSurf.py:97-107  3x3 spatial max-pool of the (n x m) log-correspondence
//                                              matrix (the reference loops over key chunks "to avoid oom")
//   isr_ep_sample       poseEstSurf.py:111-119 inversion sampling of (pixel, key) pairs from corr^alpha
//                                              WITHOUT the 4.4e8-element cumsum: f64 row / chunk sums,
//                                              then a 3-level search per sample
//   isr_ep_p3p          poseEstSurf.py:133-144 the 10 000-iteration Python loop over cv2.solveP3P:
//                                              one thread per sample, all roots, ordered by the 4th
//                                              point, Philox pick
//   isr_zbuf_score      poseEstSurf.py:182-237 batch_score: project every vertex, z-buffer by
//                                              atomicMin on packed (ordered z, vertex) u64 (replaces
//                                              torch_scatter.scatter_min), mask / coordinate scores
// The (n x m) matrix itself comes from isr_corr_logsoftmax.  RNG is explicit (Philox4x32-10): the
// reference uses torch.rand on the device and an unseeded np.random.randint.
#include "isr_common.hpp"
#include "p3p_device.hpp"

namespace {

using namespace isr_p3p;

__device__ __forceinline__ float logsigmoidf(float x) {  // torch: min(x, 0) - log1p(exp(-|x|))
  return fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
}

// ---------------------------------------------------------------- prepare: masks and queries
__global__ void ep_mask_kernel(const float* __restrict__ lgts, int r, int s, int res, float* __restrict__ lp0,
                               float* __restrict__ nlp0, float* __restrict__ mask_prob) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= res * res) return;
  const int oy = o / res, ox = o % res;
  float mx = -__builtin_inff(), nmx = -__builtin_inff(), sum = 0.f;
  for (int dy = 0; dy < s; ++dy)
    for (int dx = 0; dx < s; ++dx) {
      const float x = lgts[(size_t)(oy * s + dy) * r + ox * s + dx];
      mx = fmaxf(mx, logsigmoidf(x));
      nmx = fmaxf(nmx, logsigmoidf(-x));
      sum += x;
    }
  lp0[o] = mx;
  nlp0[o] = nmx;
  const float avg = sum / (float)(s * s);
  mask_prob[o] = 1.f / (1.f + expf(-avg));
}

__global__ void ep_pool3_kernel(const float* __restrict__ a0, const float* __restrict__ b0, int res, int do_pool,
                                float* __restrict__ a, float* __restrict__ b) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= res * res) return;
  if (!do_pool) { a[o] = a0[o]; b[o] = b0[o]; return; }
  const int oy = o / res, ox = o % res;
  float ma = -__builtin_inff(), mb = -__builtin_inff();
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx) {
      const int y = oy + dy, x = ox + dx;
      if (y >= 0 && y < res && x >= 0 && x < res) {
        ma = fmaxf(ma, a0[y * res + x]);
        mb = fmaxf(mb, b0[y * res + x]);
      }
    }
  a[o] = ma;
  b[o] = mb;
}

__global__ void ep_queries_kernel(const float* __restrict__ qimg, int r, int e, int s, int res, float* __restrict__ q) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)res * res * e) return;
  const int c = (int)(t % e);
  const int o = (int)(t / e);
  const int oy = o / res, ox = o % res;
  float sum = 0.f;
  for (int dy = 0; dy < s; ++dy)
    for (int dx = 0; dx < s; ++dx) sum += qimg[((size_t)(oy * s + dy) * r + ox * s + dx) * e + c];
  q[(size_t)o * e + c] = sum / (float)(s * s);
}

// ------------------------------------------------ avg_queries = False: per-pixel correlation, pooled
// poseEstSurf.py:72-96.  The reference evaluates log_softmax(pixel descriptor . keys) at FULL resolution
// (in patches, to fit memory), keeps the value at each scale x scale block's centre pixel (offset scale // 2) as
// the sampling matrix and the block MAXIMUM as the scoring matrix.  Here: one workgroup per output cell holds
// the block's scale^2 descriptors in LDS and sweeps the keys three times (per-pixel max, per-pixel sum, write),
// logits recomputed each time as k-ordered f32 fmaf chains — the full-resolution (r^2 x m) matrix (15.8 GB at
// r = 222, m = 80 000) never exists.
constexpr int kMaxBlockPix = 16;   // scale <= 4
constexpr int kMaxE = 64;

__global__ __launch_bounds__(256) void ep_patch_corr_kernel(const float* __restrict__ query_img, const float* __restrict__ keys,
                                                            int r, int e, int scale, int res, int m,
                                                            float* __restrict__ corr_centre, float* __restrict__ corr_blockmax) {
  __shared__ float q[kMaxBlockPix][kMaxE];
  __shared__ float red[4][kMaxBlockPix];
  __shared__ float lse_s[kMaxBlockPix];
  const int o = blockIdx.x, oy = o / res, ox = o % res;
  const int npix = scale * scale;
  for (int i = threadIdx.x; i < npix * e; i += 256) {
    const int p = i / e, d = i % e;
    const int y = oy * scale + p / scale, x = ox * scale + p % scale;
    q[p][d] = query_img[((size_t)y * r + x) * e + d];
  }
  __syncthreads();
  auto logits = [&](int k, float* out) {
    const float* kr = keys + (size_t)k * e;
    for (int p = 0; p < npix; ++p) out[p] = 0.f;
    for (int d = 0; d < e; ++d) {
      const float kv = kr[d];
      for (int p = 0; p < npix; ++p) out[p] = __builtin_fmaf(q[p][d], kv, out[p]);
    }
  };
  auto block_reduce = [&](float* v, bool is_max) {       // v[npix] -> lse_s[npix] (as broadcast storage)
    for (int p = 0; p < npix; ++p) {
      float a = v[p];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float b = __shfl_down(a, off, 64);
        a = is_max ? fmaxf(a, b) : a + b;
      }
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][p] = a;
    }
    __syncthreads();
    if (threadIdx.x < npix) {
      const int p = threadIdx.x;
      lse_s[p] = is_max ? fmaxf(fmaxf(red[0][p], red[1][p]), fmaxf(red[2][p], red[3][p]))
                        : ((red[0][p] + red[1][p]) + red[2][p]) + red[3][p];
    }
    __syncthreads();
  };
  float l[kMaxBlockPix], mx[kMaxBlockPix], sm[kMaxBlockPix];
  for (int p = 0; p < npix; ++p) mx[p] = -__builtin_inff();
  for (int k = threadIdx.x; k < m; k += 256) {
    logits(k, l);
    for (int p = 0; p < npix; ++p) mx[p] = fmaxf(mx[p], l[p]);
  }
  block_reduce(mx, true);
  for (int p = 0; p < npix; ++p) { mx[p] = lse_s[p]; sm[p] = 0.f; }
  __syncthreads();
  for (int k = threadIdx.x; k < m; k += 256) {
    logits(k, l);
    for (int p = 0; p < npix; ++p) sm[p] += __expf(l[p] - mx[p]);
  }
  block_reduce(sm, false);
  for (int p = 0; p < npix; ++p) mx[p] = mx[p] + __logf(lse_s[p]);     // per-pixel log-sum-exp
  const int centre = (scale / 2) * scale + scale / 2;
  for (int k = threadIdx.x; k < m; k += 256) {
    logits(k, l);
    float best = -__builtin_inff();
    for (int p = 0; p < npix; ++p) best = fmaxf(best, l[p] - mx[p]);
    corr_centre[(size_t)o * m + k] = l[centre] - mx[centre];
    corr_blockmax[(size_t)o * m + k] = best;
  }
}

// ------------------------------------------------------- 3x3 spatial max-pool of corr_log (n x m)
__global__ void ep_pool_corr_kernel(const float* __restrict__ in, int res, int m, float* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int o = blockIdx.y;
  if (k >= m) return;
  const int oy = o / res, ox = o % res;
  float mx = -__builtin_inff();
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx) {
      const int y = oy + dy, x = ox + dx;
      if (y >= 0 && y < res && x >= 0 && x < res) mx = fmaxf(mx, in[(size_t)(y * res + x) * m + k]);
    }
  out[(size_t)o * m + k] = mx;
}

// ------------------------------------------------------------------------ inversion sampling
// weight of (pixel o, key k): (exp(corr_log) * mask_prob)^alpha, evaluated in f64 from the f32 inputs as
// exp(alpha * corr_log) * mask_prob^alpha so that a CPU restatement agrees to an ulp.  Two providers of corr_log[o][k]:
//   matrix  the (n x m) array isr_corr_logsoftmax / isr_ep_corr_matrices wrote (isr_ep_sample);
//   direct  <Q[g(o)], keys[k]> (k-ordered fmaf chain from 0) - lse[g(o)], formed in registers: the same bits as the matrix
//           element, and the matrix never exists (isr_ep_sample_direct).
// Both add a row's weights in ONE order, an 8-ary tree over the key index: 8 consecutive keys sequentially (a group), 8 group
// sums sequentially (a block of 64), 8 block sums (a chunk of 512: what is stored), 8 chunk sums (a super-chunk of 4 096),
// the super-chunks of a row in order, rows in a fixed-shape scan — so the two providers return identical indices, and the
// sampler descends that tree with a wave: 20 + 8 + 8 + 8 + 8 steps at m = 80 000 instead of a walk over 157 chunks and
// 512 keys.
constexpr int kChunk = 512;
constexpr int kGroup = 8;            // kChunk / kGroup = 64 groups: one per lane of the sampler's wave
constexpr int kBlock = 64;           // 8 groups
constexpr int kSuper = 8;            // chunks per super-chunk
constexpr int kRowsPerBlock = 256;

// exp(x) in f64 for the sampler's arguments (x = alpha * log-probability <= ~0), table-driven: x = (64 e + j) ln2/64 + r with
// |r| <= ln2/128, exp(x) = 2^e * 2^(j/64) * exp(r) — the 64 correctly rounded 2^(j/64) sit in LDS, exp(r) is a degree-5
// polynomial (truncation 3.5e-17), one v_ldexp_f64 (gradual underflow below e^-708, 0 below e^-745; the argument is
// clamped at -750 so that -inf gives 0, not NaN).  Branch-free, 16 f64 / integer instructions of which 7 are v_fma_f64 —
// the kernel that adds the 4.4e8 weights is bound by exactly those (v_fma_f64 issues at 6.1-6.5 cycles, the rest at 4.4-5,
// profiles/r03_microbench_f64_op_rates.txt); the degree-13 polynomial without a table (15 fma) was 0.54 ms against 0.45.
// The Horner steps are v_fma_f64 with the coefficient in a scalar register pair — left alone the compiler keeps the
// coefficients in VGPRs and copies each one into the destination of a v_fmac_f64.  Within 2 ulp of the correctly rounded value.
__device__ const double kExp2Tab[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0};

__device__ __forceinline__ void load_exp_table(double* tab_lds) {       // call with all threads, then __syncthreads()
  if (threadIdx.x < 64) tab_lds[threadIdx.x] = kExp2Tab[threadIdx.x];
}

__device__ __forceinline__ double fma_sc(double a, double b, double c_scalar) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_scalar));
  return d;
}

__device__ __forceinline__ double ep_exp(double x, const double* tab) {
  x = __builtin_fmax(x, -750.0);
  const double nf = __builtin_rint(x * 92.33248261689366);            // 64 / ln 2
  double r = __builtin_fma(nf, -0.010830424696905538, x);              // ln2/64, upper 32 bits: nf * hi is exact
  r = __builtin_fma(nf, 6.563929801064195e-13, r);                     // minus its remainder
  const int ni = (int)nf;
  const double T = tab[ni & 63];
  double p = 8.33333333333333333333e-03;                               // 1/5!
  p = fma_sc(p, r, 4.16666666666666666667e-02);                        // 1/4!
  p = fma_sc(p, r, 1.66666666666666666667e-01);                        // 1/3!
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_ldexp(__dmul_rn(T, p), ni >> 6);
}

// explicit roundings: no fused multiply-add across the weight and the running sum (the oracle multiplies, then adds)
__device__ __forceinline__ double ep_weight(float cl, double mpa, double alpha, const double* tab) {
  return __dmul_rn(ep_exp(__dmul_rn(alpha, (double)cl), tab), mpa);
}

__global__ void ep_mpa_kernel(const float* __restrict__ mask_prob, int n, double alpha, double* __restrict__ mpa) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o < n) mpa[o] = pow((double)mask_prob[o], alpha);
}

// test / validation aid: the weights themselves
__global__ void ep_weights_kernel(const float* __restrict__ corr_log, const double* __restrict__ mpa, int n, int m,
                                  double alpha, double* __restrict__ w) {
  __shared__ double tab[64];
  load_exp_table(tab);
  __syncthreads();
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)n * m) w[i] = ep_weight(corr_log[i], mpa[i / m], alpha, tab);
}

// matrix provider: workgroup = 256 rows x one chunk.  The chunk is walked in 32-key slices: the slice of every row is
// read coalesced (half a wave per row), parked in LDS, then thread t adds row t's 32 weights in key order.
__global__ __launch_bounds__(kRowsPerBlock) void ep_chunk_sums_kernel(const float* __restrict__ corr_log,
                                                                      const double* __restrict__ mpa_all, int n, int m, int nchunk,
                                                                      double alpha, double* __restrict__ chunk_sums) {
  __shared__ float tile[kRowsPerBlock][33];
  __shared__ double tab[64];
  load_exp_table(tab);
  const int tid = threadIdx.x, c = blockIdx.x, o0 = blockIdx.y * kRowsPerBlock;
  const int o = o0 + tid;
  const double mpa = o < n ? mpa_all[o] : 0.0;
  const int k0 = c * kChunk, kend = min(m, k0 + kChunk);
  double s = 0.0, s64 = 0.0;
  for (int kb = k0; kb < kend; kb += 32) {
    __syncthreads();
    for (int i = tid; i < kRowsPerBlock * 32; i += kRowsPerBlock) {
      const int rr = i >> 5, kk = i & 31;
      tile[rr][kk] = (o0 + rr < n && kb + kk < kend) ? corr_log[(size_t)(o0 + rr) * m + kb + kk] : 0.f;
    }
    __syncthreads();
    const int cnt = min(32, kend - kb);
    for (int j0 = 0; j0 < cnt; j0 += kGroup) {
      double s8 = 0.0;
      for (int j = j0; j < min(cnt, j0 + kGroup); ++j) s8 = __dadd_rn(s8, ep_weight(tile[tid][j], mpa, alpha, tab));
      s64 = __dadd_rn(s64, s8);
    }
    if (((kb - k0) & 32) || kb + 32 >= kend) {      // a block of 64 keys (two slices) is complete
      s = __dadd_rn(s, s64);
      s64 = 0.0;
    }
  }
  if (o < n) chunk_sums[(size_t)o * nchunk + c] = s;
}

// the grid of descriptors a direct provider reads: output pixel o = (oy, ox) of the res x res grid takes its query from grid
// pixel g(o) = oy * sy + ox * sx + off (avg_queries: the pooled queries themselves, sy = res, sx = 1, off = 0; per-pixel
// queries: the centre pixel of the scale x scale block of the r x r crop, sy = scale * r, sx = scale, off = (scale/2)(r+1))
struct QGrid {
  const float* q;      // (grid pixels, e)
  const float* lse;    // (grid pixels): log-sum-exp of the pixel's logits over all keys (K1)
  int e, res, sy, sx, off;
  __device__ __forceinline__ int pixel(int o) const { return (o / res) * sy + (o % res) * sx + off; }
};

// row `i` of an (rows, e) f32 array into DP registers, zero past e; 16-byte loads when e is a multiple of 4
template <int DP>
__device__ __forceinline__ void load_row(const float* __restrict__ a, size_t i, int e, float (&v)[DP]) {
  const float* r = a + i * e;
  if ((e & 3) == 0) {
#pragma unroll
    for (int d4 = 0; d4 < DP / 4; ++d4) {
      const float4 x = 4 * d4 < e ? reinterpret_cast<const float4*>(r)[d4] : make_float4(0.f, 0.f, 0.f, 0.f);
      v[4 * d4] = x.x; v[4 * d4 + 1] = x.y; v[4 * d4 + 2] = x.z; v[4 * d4 + 3] = x.w;
    }
  } else {
#pragma unroll
    for (int d = 0; d < DP; ++d) v[d] = d < e ? r[d] : 0.f;
  }
}

template <int DP>
__device__ __forceinline__ float chain(const float (&a)[DP], const float (&b)[DP]) {
  float acc = 0.f;
#pragma unroll
  for (int d = 0; d < DP; ++d) acc = __builtin_fmaf(a[d], b[d], acc);
  return acc;
}

template <int DP>
__device__ __forceinline__ void load_query(const QGrid& g, int o, float (&q)[DP], float* lse) {
  const int px = g.pixel(o);
  load_row<DP>(g.q, (size_t)px, g.e, q);
  *lse = g.lse[px];
}

// direct provider: workgroup = 256 rows (thread = row, its query in registers) x one chunk of keys, staged through LDS in
// slices of 8192 / DP keys and read back as broadcasts; per (row, key): DP fma, one subtraction, the f64 weight, one add.
template <int DP>
__global__ __launch_bounds__(kRowsPerBlock) void ep_chunk_sums_direct_kernel(QGrid g, const double* __restrict__ mpa_all,
                                                                             const float* __restrict__ keys, int n, int m,
                                                                             int nchunk, double alpha,
                                                                             double* __restrict__ chunk_sums) {
  constexpr int kSlice = DP <= 16 ? kChunk : 8192 / DP;     // keys per LDS stage (a multiple of kGroup)
  __shared__ __attribute__((aligned(16))) float ks[kSlice * DP];
  __shared__ double tab[64];
  load_exp_table(tab);
  const int tid = threadIdx.x, c = blockIdx.x;
  const int o = blockIdx.y * kRowsPerBlock + tid;
  float q[DP], lse = 0.f;
  if (o < n) load_query<DP>(g, o, q, &lse);
  else {
#pragma unroll
    for (int d = 0; d < DP; ++d) q[d] = 0.f;
  }
  const double mpa = o < n ? mpa_all[o] : 0.0;
  const int k0 = c * kChunk, kend = min(m, k0 + kChunk);
  double s = 0.0;
  for (int kb = k0; kb < kend; kb += kSlice) {
    __syncthreads();
    for (int i = tid; i < kSlice * DP; i += kRowsPerBlock) {
      const int kk = i / DP, d = i % DP;
      ks[i] = (kb + kk < kend && d < g.e) ? keys[(size_t)(kb + kk) * g.e + d] : 0.f;
    }
    __syncthreads();
    const int cnt = min(kSlice, kend - kb);
    auto weight = [&](int j) {
      const float4* k4 = reinterpret_cast<const float4*>(ks + j * DP);
      float acc = 0.f;
#pragma unroll
      for (int d4 = 0; d4 < DP / 4; ++d4) {
        const float4 kk = k4[d4];
        acc = __builtin_fmaf(q[4 * d4], kk.x, acc);
        acc = __builtin_fmaf(q[4 * d4 + 1], kk.y, acc);
        acc = __builtin_fmaf(q[4 * d4 + 2], kk.z, acc);
        acc = __builtin_fmaf(q[4 * d4 + 3], kk.w, acc);
      }
      return ep_weight(acc - lse, mpa, alpha, tab);
    };
    for (int jb = 0; jb < cnt; jb += kBlock) {          // kSlice is a multiple of kBlock: blocks do not straddle slices
      const int je = min(cnt, jb + kBlock);
      double s64 = 0.0;
      int j0 = jb;
      for (; j0 + kGroup <= je; j0 += kGroup) {         // full groups: eight independent weights, then their chain of adds
        double w[kGroup];
#pragma unroll
        for (int i = 0; i < kGroup; ++i) w[i] = weight(j0 + i);
        double s8 = 0.0;
#pragma unroll
        for (int i = 0; i < kGroup; ++i) s8 = __dadd_rn(s8, w[i]);
        s64 = __dadd_rn(s64, s8);
      }
      if (j0 < je) {
        double s8 = 0.0;
        for (int j = j0; j < je; ++j) s8 = __dadd_rn(s8, weight(j));
        s64 = __dadd_rn(s64, s8);
      }
      s = __dadd_rn(s, s64);
    }
  }
  if (o < n) chunk_sums[(size_t)o * nchunk + c] = s;
}

// The same sums with the logits on the matrix cores: v_mfma_f32_32x32x2_f32 is bit for bit the k-ordered fmaf chain (K1's
// exact-f32 path rests on that), so a wave forms a 32-key x 32-row tile of logits with DP / 2 MFMAs instead of 16 x DP fma
// per lane, and the VALU is left with the f64 weights alone (26 of the 41 instructions per element of the kernel above).
// Tile layout (A = keys, B = queries): lane (r, h) = (lane & 31, lane >> 5) holds, for row r, the keys 8a + 4h + b of the
// tile in register 4a + b — half h of each 8-key group.  The group's chain of adds starts in the h = 0 lane (keys 0..3),
// crosses to the h = 1 lane with one v_permlane32_swap per 32-bit half, and ends there (keys 4..7); block and chunk sums
// live in the h = 1 lanes.  Workgroup = 4 waves = 128 rows x one chunk.
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ double from_lower_half(double v) {       // lanes 32..63 receive lanes 0..31's value
  unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]);
}

template <int DP>
__global__ __launch_bounds__(256) void ep_chunk_sums_mfma_kernel(QGrid g, const double* __restrict__ mpa_all,
                                                                 const float* __restrict__ keys, int n, int m, int nchunk,
                                                                 double alpha, double* __restrict__ chunk_sums) {
  constexpr int KS = DP / 2;
  constexpr int LD = DP + 1;                                    // odd dword stride: conflict-free reads down a column
  constexpr int kSlice = DP <= 16 ? kChunk : 8192 / DP;         // keys per LDS stage (a multiple of kBlock)
  __shared__ float ks[kSlice * LD];
  __shared__ double tab[64];
  load_exp_table(tab);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  const int o = blockIdx.y * 128 + wave * 32 + r;
  const int oc = min(o, n - 1);
  const int px = g.pixel(oc);
  float bq[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) bq[s] = 2 * s + h < g.e ? g.q[(size_t)px * g.e + 2 * s + h] : 0.f;
  const float lse = g.lse[px];
  const double mpa = mpa_all[oc];
  const int c = blockIdx.x, k0 = c * kChunk, kend = min(m, k0 + kChunk);
  double s512 = 0.0;
  for (int kb = k0; kb < kend; kb += kSlice) {
    __syncthreads();
    for (int i = tid; i < kSlice * DP; i += 256) {
      const int kk = i / DP, d = i % DP;
      ks[kk * LD + d] = (kb + kk < kend && d < g.e) ? keys[(size_t)(kb + kk) * g.e + d] : 0.f;
    }
    __syncthreads();
    const int cnt = min(kSlice, kend - kb);
    for (int jb = 0; jb < cnt; jb += kBlock) {
      double s64 = 0.0;
#pragma unroll
      for (int tile = 0; tile < kBlock / 32; ++tile) {
        const int t0 = jb + 32 * tile;
        if (t0 >= cnt) break;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const float* arow = ks + (t0 + r) * LD + h;
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[2 * s], bq[s], acc, 0, 0, 0);
        double part[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {                    // this lane's half of group a: keys t0 + 8a + 4h + (0..3)
          const int kq = t0 + 8 * a + 4 * h;
          double w[4];
#pragma unroll
          for (int b = 0; b < 4; ++b) w[b] = kq + b < cnt ? ep_weight(acc[4 * a + b] - lse, mpa, alpha, tab) : 0.0;
          const double from_h0 = from_lower_half(__dadd_rn(__dadd_rn(__dadd_rn(w[0], w[1]), w[2]), w[3]));
          part[a] = __dadd_rn(__dadd_rn(__dadd_rn(__dadd_rn(from_h0, w[0]), w[1]), w[2]), w[3]);    // h = 1: the group sum
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) s64 = __dadd_rn(s64, part[a]);     // (a missing group adds +0.0: exact)
      }
      s512 = __dadd_rn(s512, s64);
    }
  }
  if (h == 1 && o < n) chunk_sums[(size_t)o * nchunk + c] = s512;
}

// a wave per row: the row's chunk sums are read coalesced, then added in chunk order (every lane runs the same chain on
// broadcast values); ep_row_scan_kernel does the inclusive scan over rows.
__device__ __forceinline__ double bcast_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// lane L's super-chunk of a row: its (up to) 8 chunk sums and their sequential sum
__device__ __forceinline__ double super_chunk(const double* __restrict__ row_chunks, int nchunk, int sup, double (&c8)[kSuper]) {
  double sc = 0.0;
#pragma unroll
  for (int j = 0; j < kSuper; ++j) {
    const int c = sup * kSuper + j;
    c8[j] = c < nchunk ? row_chunks[c] : 0.0;
    if (c < nchunk) sc = __dadd_rn(sc, c8[j]);
  }
  return sc;
}

__global__ __launch_bounds__(256) void ep_row_sums_kernel(const double* __restrict__ chunk_sums, int n, int nchunk,
                                                          double* __restrict__ row_sums) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= n) return;
  const int nsuper = (nchunk + kSuper - 1) / kSuper;
  double s = 0.0;
  for (int base = 0; base < nsuper; base += 64) {
    double c8[kSuper];
    const double sc = super_chunk(chunk_sums + (size_t)o * nchunk, nchunk, base + lane, c8);
    const int cnt = min(64, nsuper - base);
    for (int i = 0; i < cnt; ++i) s = __dadd_rn(s, bcast_f64(sc, i));
  }
  if (lane == 0) row_sums[o] = s;
}

// inclusive scan of the row sums by ONE workgroup of 1024 threads: thread t adds its contiguous slice of rows in order,
// the slice totals are scanned across the block (Hillis-Steele in LDS), each thread then rewrites its slice.  (Round 2
// scanned the 5 476 rows with one thread: 375 us per estimate_pose call.)  A fixed shape: run-to-run reproducible.
__global__ __launch_bounds__(1024) void ep_row_scan_kernel(const double* __restrict__ row_sums, int n, double* __restrict__ row_cum) {
  __shared__ double tot[1024];
  const int t = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int lo = t * per, hi = min(n, lo + per);
  double s = 0.0;
  for (int o = lo; o < hi; ++o) s += row_sums[o];
  tot[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const double v = (t >= off) ? tot[t - off] : 0.0;
    __syncthreads();
    tot[t] += v;
    __syncthreads();
  }
  double run = (t > 0) ? tot[t - 1] : 0.0;
  for (int o = lo; o < hi; ++o) { run += row_sums[o]; row_cum[o] = run; }
}

// sample (s, j): u = (x + 0.5) / 2^32 of Philox(counter = (s,1,0,0)); index = first flat position whose inclusive
// cumulative weight reaches u * total (np.searchsorted, side='left').  One WAVE per draw: every lane bisects the row scan
// (uniform), the row's chunk sums are read 64 at a time and walked in order, then lane l rebuilds group l of the chunk
// (its kGroup weights: the only expensive part, done 64-wide), the group sums are walked, and the winning group's weights.
// DP = 0: the matrix provider (corr_log), else the direct one.  (Round 3, one THREAD per draw: 0.46 ms of dependent
// 500-cycle steps for 40 000 draws.)
template <int DP>
__global__ __launch_bounds__(256) void ep_sample_kernel(const float* __restrict__ corr_log, QGrid g,
                                                        const float* __restrict__ keys, const double* __restrict__ mpa_all, int n,
                                                        int m, int nchunk, double alpha, const double* __restrict__ chunk_sums,
                                                        const double* __restrict__ row_cum, int n_samples, uint32_t seed_lo,
                                                        uint32_t seed_hi, int64_t* __restrict__ corr_idx) {
  __shared__ double wbuf[4][kChunk];
  __shared__ double tab[64];
  load_exp_table(tab);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t = min(blockIdx.x * 4 + wave, n_samples * 4 - 1);       // 4 draws per sample: the grid is exact
  const int smp = t >> 2, j = t & 3;
  uint32_t rnd[4];
  philox4x32_10((uint32_t)smp, 1u, 0u, 0u, seed_lo, seed_hi, rnd);
  const double u = ((double)rnd[j] + 0.5) * (1.0 / 4294967296.0);
  const double target = u * row_cum[n - 1];
  // row: first o with row_cum[o] >= target — a 64-ary search (three dependent loads at n = 5 476 instead of thirteen)
  int lo = 0, hi = n;
  while (hi - lo > 64) {
    const int step = (hi - lo + 63) / 64;
    const int idx = min(hi - 1, lo + (lane + 1) * step - 1);
    const unsigned long long ge = __ballot(row_cum[idx] >= target);
    const int f = ge ? __ffsll((long long)ge) - 1 : 63;
    const int nlo = lo + f * step;
    hi = min(hi, nlo + step);
    lo = nlo;
  }
  {
    const unsigned long long ge = __ballot(lo + lane < hi && row_cum[lo + lane] >= target);
    lo = ge ? lo + __ffsll((long long)ge) - 1 : hi - 1;
  }
  const int o = lo;
  double rem = target - (o ? row_cum[o - 1] : 0.0);
  // every level below: the first child whose inclusive cumulative reaches what is left of the target, the last one otherwise
  // super-chunk (lane = super-chunk, 64 at a time), then the chunk inside it
  const int nsuper = (nchunk + kSuper - 1) / kSuper;
  int c;
  {
    const double* row_chunks = chunk_sums + (size_t)o * nchunk;
    double acc = 0.0, c8[kSuper];
    int sup = nsuper - 1, base = 0;
    bool found = false;
    for (; base < nsuper; base += 64) {
      const double sc = super_chunk(row_chunks, nchunk, base + lane, c8);
      const int cnt = min(64, nsuper - base);
      for (int i = 0; i < cnt; ++i) {
        if (base + i == nsuper - 1) { found = true; break; }
        const double nx = __dadd_rn(acc, bcast_f64(sc, i));
        if (nx >= rem) { sup = base + i; found = true; break; }
        acc = nx;
      }
      if (found) break;
    }
    rem -= acc;
    const int nch = min(kSuper, nchunk - sup * kSuper);
    int cj = nch - 1;
    acc = 0.0;
#pragma unroll
    for (int jj = 0; jj < kSuper; ++jj) {
      const double v = bcast_f64(c8[jj], sup - base);
      if (jj < nch - 1 && cj == nch - 1) {
        const double nx = __dadd_rn(acc, v);
        if (nx >= rem) cj = jj; else acc = nx;
      }
    }
    rem -= acc;
    c = sup * kSuper + cj;
  }
  // the chunk's weights, 64 consecutive keys at a time (coalesced rows), parked in LDS
  const double mpa = mpa_all[o];
  const int k0 = c * kChunk, kend = min(m, k0 + kChunk);
  if constexpr (DP == 0) {
#pragma unroll
    for (int i = 0; i < kChunk / 64; ++i) {
      const int k = k0 + i * 64 + lane;
      wbuf[wave][i * 64 + lane] = k < kend ? ep_weight(corr_log[(size_t)o * m + k], mpa, alpha, tab) : 0.0;
    }
  } else {
    constexpr int DQ = DP ? DP : 4;
    float q[DQ], lse;
    load_query<DQ>(g, o, q, &lse);
#pragma unroll
    for (int i = 0; i < kChunk / 64; ++i) {
      const int k = k0 + i * 64 + lane;
      float kv[DQ];
      load_row<DQ>(keys, (size_t)min(k, kend - 1), g.e, kv);
      wbuf[wave][i * 64 + lane] = k < kend ? ep_weight(chain<DQ>(q, kv) - lse, mpa, alpha, tab) : 0.0;
    }
  }
  __syncthreads();
  // lane = group: its kGroup weights and their sum; lanes 0..7 = blocks of 8 groups
  const int nkeys = kend - k0;
  const int kg = lane * kGroup;
  double w[kGroup];
#pragma unroll
  for (int i = 0; i < kGroup; ++i) w[i] = wbuf[wave][kg + i];
  double s8 = 0.0;
#pragma unroll
  for (int i = 0; i < kGroup; ++i)
    if (kg + i < nkeys) s8 = __dadd_rn(s8, w[i]);
  __syncthreads();
  wbuf[wave][lane] = s8;                     // the group sums (the weights are in registers now)
  __syncthreads();
  double s64 = 0.0;
  if (lane < kChunk / kBlock) {
#pragma unroll
    for (int i = 0; i < kBlock / kGroup; ++i)
      if ((lane * (kBlock / kGroup) + i) * kGroup < nkeys) s64 = __dadd_rn(s64, wbuf[wave][lane * (kBlock / kGroup) + i]);
  }
  // descend: block, group, key
  auto pick = [&](int nchild, auto value) {    // value(i): child i's sum, wave-uniform
    int sel = nchild - 1;
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const double v = value(i);
      if (i < nchild - 1 && sel == nchild - 1) {
        const double nx = __dadd_rn(acc, v);
        if (nx >= rem) sel = i; else acc = nx;
      }
    }
    rem -= acc;
    return sel;
  };
  const int nblocks = (nkeys + kBlock - 1) / kBlock;
  const int bi = pick(nblocks, [&](int i) { return bcast_f64(s64, i); });
  const int ngroups = min(kBlock / kGroup, (nkeys - bi * kBlock + kGroup - 1) / kGroup);
  const int gi = bi * (kBlock / kGroup) + pick(ngroups, [&](int i) { return bcast_f64(s8, bi * (kBlock / kGroup) + i); });
  const int cnt = min(kGroup, nkeys - gi * kGroup);
  const int kk = pick(cnt, [&](int i) { return bcast_f64(w[i], gi); });
  if (lane == 0 && blockIdx.x * 4 + wave < n_samples * 4) corr_idx[t] = (int64_t)o * m + k0 + gi * kGroup + kk;
}

// ------------------------------------------------------------------------------------- P3P
__global__ void ep_p3p_kernel(const int64_t* __restrict__ corr_idx, int res, int m, const float* __restrict__ obj_pts,
                              Cam cam, int S, uint32_t seed_lo, uint32_t seed_hi, double* __restrict__ poses,
                              uint8_t* __restrict__ ok_out) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  int64_t ci[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) ci[j] = corr_idx[4 * (size_t)s + j];
  const bool distinct = ci[0] != ci[1] && ci[0] != ci[2] && ci[0] != ci[3] && ci[1] != ci[2] && ci[1] != ci[3] && ci[2] != ci[3];
  V3 X[4];
  double uu[4], vv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int pix = (int)(ci[j] / m), k = (int)(ci[j] % m);
    uu[j] = (double)(pix % res);
    vv[j] = (double)(pix / res);
    X[j] = {(double)obj_pts[3 * (size_t)k], (double)obj_pts[3 * (size_t)k + 1], (double)obj_pts[3 * (size_t)k + 2]};
  }
  double* out = poses + 12 * (size_t)s;
  uint8_t ok = 0;
  if (distinct) {
    P3PIn in;
#pragma unroll
    for (int j = 0; j < 3; ++j) { in.x[j] = X[j]; in.y[j] = bearing(cam, uu[j], vv[j]); }
    double lam[4][3];
    const int n = p3p_depths(in, lam);
    double cand[4][12], err[4];
    int nv = 0;
    for (int i = 0; i < n; ++i) {
      if (!pose_from_depths(in, lam[i], cand[nv])) continue;
      double e2;
      err[nv] = reproj_err2(cam, cand[nv], X[3], uu[3], vv[3], &e2) ? e2 : 1e300;
      ++nv;
    }
    if (nv > 0) {
      // order by the 4th point's reprojection error (stable insertion sort of indices)
      int ord[4] = {0, 1, 2, 3};
      for (int a = 1; a < nv; ++a) {
        const int x = ord[a];
        int b = a - 1;
        while (b >= 0 && err[ord[b]] > err[x]) { ord[b + 1] = ord[b]; --b; }
        ord[b + 1] = x;
      }
      uint32_t rnd[4];
      philox4x32_10((uint32_t)s, 2u, 0u, 0u, seed_lo, seed_hi, rnd);
      const int pick = (int)(((uint64_t)rnd[0] * (uint64_t)nv) >> 32);
      const int sel = ord[pick];
      ok = 1;
      for (int q = 0; q < 12; ++q) {
        double v = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) v = (i == sel) ? cand[i][q] : v;  // no runtime-indexed register array
        out[q] = v;
      }
    }
  }
  if (!ok)
    for (int q = 0; q < 12; ++q) out[q] = 0.0;
  ok_out[s] = ok;
}

// --------------------------------------------------------------------------- z-buffer scoring
__device__ __forceinline__ uint32_t ordered_u32(float f) {
  const uint32_t b = __float_as_uint(f);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float unordered_f32(uint32_t u) {
  return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

struct K9f { float k[9]; };
constexpr size_t kFusedLdsMax = 144 * 1024;      // of gfx950's 160 KB per workgroup: res <= 135 with the fused z-buffer

__global__ void zbuf_project_kernel(const float* __restrict__ pts, int m, const float* __restrict__ Rt, int B, K9f K,
                                    int res, unsigned long long* __restrict__ zbuf) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (v >= m) return;
  const float* T = Rt + 12 * (size_t)b;  // uniform per block row: scalar loads
  const float x = pts[3 * (size_t)v], y = pts[3 * (size_t)v + 1], z = pts[3 * (size_t)v + 2];
  // obj_pts @ R^T + t, then @ K^T, in f32 as the reference (torch f32 matmuls)
  const float cx = x * T[0] + y * T[1] + z * T[2] + T[3];
  const float cy = x * T[4] + y * T[5] + z * T[6] + T[7];
  const float cz = x * T[8] + y * T[9] + z * T[10] + T[11];
  const float ix = cx * K.k[0] + cy * K.k[1] + cz * K.k[2];
  const float iy = cx * K.k[3] + cy * K.k[4] + cz * K.k[5];
  const float iz = cx * K.k[6] + cy * K.k[7] + cz * K.k[8];
  const float ux = rintf(ix / iz), uy = rintf(iy / iz);  // torch.round_: half to even
  if (!(ux >= 0.f) || !(ux < (float)res) || !(uy >= 0.f) || !(uy < (float)res)) return;  // ignore bin (NaN too)
  const int pix = (int)uy * res + (int)ux;
  const unsigned long long key = ((unsigned long long)ordered_u32(cz) << 32) | (unsigned)v;
  // (a read-and-compare in front of the atomic was measured in round 3: 1.63 ms against 0.5 ms — the returning load costs
  // more than the fire-and-forget atomic it saves)
  atomicMin(&zbuf[(size_t)b * res * res + pix], key);
}

// one block per pose: mask score = mean over ALL pixels of (hit ? mask_log_prob : neg_mask_log_prob) / ln 2,
// coord score = mean over hit pixels of corr_log[pixel, nearest vertex] / ln m  (-inf without hits)
__global__ __launch_bounds__(256) void zbuf_score_kernel(const unsigned long long* __restrict__ zbuf, int n, int m,
                                                         const float* __restrict__ mlp, const float* __restrict__ nmlp,
                                                         const float* __restrict__ corr_log, float* __restrict__ pose_score,
                                                         float* __restrict__ mask_score, float* __restrict__ coord_score) {
  __shared__ double red[3][4];
  const int b = blockIdx.x;
  double sm = 0.0, sc = 0.0, cnt = 0.0;
  for (int pix = threadIdx.x; pix < n; pix += 256) {
    const unsigned long long key = zbuf[(size_t)b * n + pix];
    bool hit = false;
    if (key != ~0ull) {
      const float z = unordered_f32((uint32_t)(key >> 32));
      if (z > 0.f) {
        hit = true;
        sc += (double)corr_log[(size_t)pix * m + (uint32_t)key];
        cnt += 1.0;
      }
    }
    sm += (double)(hit ? mlp[pix] : nmlp[pix]);
  }
  double v[3] = {sm, sc, cnt};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    double s = v[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[i][threadIdx.x >> 6] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double tm = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
    const double tc = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
    const double tn = ((red[2][0] + red[2][1]) + red[2][2]) + red[2][3];
    const float ms = (float)(tm / (double)n / 0.6931471805599453);
    const float cs = tn > 0.0 ? (float)(tc / tn / log((double)m)) : -__builtin_inff();
    mask_score[b] = ms;
    coord_score[b] = cs;
    pose_score[b] = ms + cs;
  }
}

// the same scores with corr_log[pixel, vertex] formed on the spot (no matrix): the pooled log-correspondence of output pixel
// (py, px) and key v is the maximum, over the grid pixels of its window, of <q_g, key_v> (k-ordered fmaf chain) - lse_g —
// window = the output pixels (py-1..py+1, px-1..px+1) inside the res x res grid when `pool` (F.max_pool2d pads with -inf),
// else the pixel itself, each output pixel standing for win x win grid pixels (1: pooled queries, scale: per-pixel queries).
// The elements are the ones isr_ep_corr_matrices / isr_ep_patch_corr + isr_ep_pool_corr write, bit for bit.
template <int DP>
__global__ __launch_bounds__(256) void zbuf_score_direct_kernel(const unsigned long long* __restrict__ zbuf, int res, int m,
                                                                const float* __restrict__ mlp, const float* __restrict__ nmlp,
                                                                const float* __restrict__ qgrid, const float* __restrict__ lse_grid,
                                                                int g_pitch, int e, int win, int pool,
                                                                const float* __restrict__ keys, float* __restrict__ pose_score,
                                                                float* __restrict__ mask_score, float* __restrict__ coord_score) {
  __shared__ double red[3][4];
  const int b = blockIdx.x, n = res * res;
  double sm = 0.0, sc = 0.0, cnt = 0.0;
  for (int pix = threadIdx.x; pix < n; pix += 256) {
    const unsigned long long key = zbuf[(size_t)b * n + pix];
    bool hit = false;
    if (key != ~0ull) {
      const float z = unordered_f32((uint32_t)(key >> 32));
      if (z > 0.f) {
        hit = true;
        float k[DP];
        load_row<DP>(keys, (size_t)(uint32_t)key, e, k);
        const int py = pix / res, px = pix % res;
        const int y0 = (pool ? max(0, py - 1) : py) * win, y1 = (pool ? min(res, py + 2) : py + 1) * win;
        const int x0 = (pool ? max(0, px - 1) : px) * win, x1 = (pool ? min(res, px + 2) : px + 1) * win;
        float best = -__builtin_inff();
        for (int gy = y0; gy < y1; ++gy)
          for (int gx = x0; gx < x1; ++gx) {
            const size_t gp = (size_t)gy * g_pitch + gx;
            float qv[DP];
            load_row<DP>(qgrid, gp, e, qv);
            best = fmaxf(best, chain<DP>(qv, k) - lse_grid[gp]);
          }
        sc += (double)best;
        cnt += 1.0;
      }
    }
    sm += (double)(hit ? mlp[pix] : nmlp[pix]);
  }
  double v[3] = {sm, sc, cnt};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    double s = v[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[i][threadIdx.x >> 6] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double tm = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
    const double tc = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
    const double tn = ((red[2][0] + red[2][1]) + red[2][2]) + red[2][3];
    const float ms = (float)(tm / (double)n / 0.6931471805599453);
    const float cs = tn > 0.0 ? (float)(tc / tn / log((double)m)) : -__builtin_inff();
    mask_score[b] = ms;
    coord_score[b] = cs;
    pose_score[b] = ms + cs;
  }
}

// The direct scorer, fused with its z-buffer: one workgroup of 512 threads per pose keeps the res x res buffer of packed
// (ordered z, vertex) words in LDS (43.8 KB at res = 74), projects all m vertices into it with LDS atomics, evaluates the
// pooled log-correspondence of every hit pixel 512-wide into the vertex half of the pixel's word, and its first 256 threads
// then add the pixels in exactly zbuf_score_kernel's order (thread t: pixels t, t + 256, ... in f64; 64-lane tree; four wave partials).
// No global z-buffer, no memset, no global atomics: round 3's zbuf_project_kernel was 0.5 ms per 224 poses (4.4 ms per
// 1 000) of L2 atomics on ~50 vertices per covered pixel.
// maximum over the 64 lanes without the LDS crossbar: four DPP steps inside each row of 16, then the four row leaders
__device__ __forceinline__ float wave_max_f32(float v) {
  auto step = [&](auto ctrl) {
    const int o = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), decltype(ctrl)::value, 0xF, 0xF, false);
    v = fmaxf(v, __int_as_float(o));
  };
  step(std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
  step(std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
  step(std::integral_constant<int, 0x141>{});     // row_half_mirror
  step(std::integral_constant<int, 0x140>{});     // row_mirror
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

template <int DP, int THREADS>
__global__ __launch_bounds__(THREADS) void zbuf_fused_direct_kernel(const float* __restrict__ pts, int m, const float* __restrict__ Rt,
                                                                    K9f K, int res, const float* __restrict__ mlp,
                                                                    const float* __restrict__ nmlp, const float* __restrict__ qgrid,
                                                                    const float* __restrict__ lse_grid, int g_pitch, int e, int win,
                                                                    int pool, const float* __restrict__ keys,
                                                                    float* __restrict__ pose_score, float* __restrict__ mask_score,
                                                                    float* __restrict__ coord_score) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long zb[];     // n words: (ordered z, vertex), later (ordered z, score);
  __shared__ double red[3][4];                                               // then n u16: the pixels that were hit
  __shared__ int nhit_s;
  const int b = blockIdx.x, n = res * res, tid = threadIdx.x;
  uint16_t* hits = reinterpret_cast<uint16_t*>(zb + n);
  for (int i = tid; i < n; i += THREADS) zb[i] = ~0ull;
  if (tid == 0) nhit_s = 0;
  __syncthreads();
  const float* T = Rt + 12 * (size_t)b;
#pragma unroll 4
  for (int v = tid; v < m; v += THREADS) {
    const float x = pts[3 * (size_t)v], y = pts[3 * (size_t)v + 1], z = pts[3 * (size_t)v + 2];
    const float cx = x * T[0] + y * T[1] + z * T[2] + T[3];
    const float cy = x * T[4] + y * T[5] + z * T[6] + T[7];
    const float cz = x * T[8] + y * T[9] + z * T[10] + T[11];
    const float ix = cx * K.k[0] + cy * K.k[1] + cz * K.k[2];
    const float iy = cx * K.k[3] + cy * K.k[4] + cz * K.k[5];
    const float iz = cx * K.k[6] + cy * K.k[7] + cz * K.k[8];
    const float ux = rintf(ix / iz), uy = rintf(iy / iz);
    if ((ux >= 0.f) && (ux < (float)res) && (uy >= 0.f) && (uy < (float)res))
      atomicMin(&zb[(int)uy * res + (int)ux], ((unsigned long long)ordered_u32(cz) << 32) | (unsigned)v);
  }
  __syncthreads();
  auto is_hit = [](unsigned long long key) { return key != ~0ull && unordered_f32((uint32_t)(key >> 32)) > 0.f; };
  auto window = [&](int pix, int* y0, int* y1, int* x0, int* x1) {
    const int py = pix / res, px = pix % res;
    *y0 = (pool ? max(0, py - 1) : py) * win; *y1 = (pool ? min(res, py + 2) : py + 1) * win;
    *x0 = (pool ? max(0, px - 1) : px) * win; *x1 = (pool ? min(res, px + 2) : px + 1) * win;
  };
  const int wmax = (pool ? 3 : 1) * win;
  if (wmax * wmax <= 16) {
    // small windows (pooled queries: 3 x 3): one thread per output pixel
    for (int pix = tid; pix < n; pix += THREADS) {
      const unsigned long long key = zb[pix];
      if (is_hit(key)) {
        float k[DP];
        load_row<DP>(keys, (size_t)(uint32_t)key, e, k);
        int y0, y1, x0, x1;
        window(pix, &y0, &y1, &x0, &x1);
        float best = -__builtin_inff();
        for (int gy = y0; gy < y1; ++gy)
          for (int gx = x0; gx < x1; ++gx) {
            const size_t gp = (size_t)gy * g_pitch + gx;
            float qv[DP];
            load_row<DP>(qgrid, gp, e, qv);
            best = fmaxf(best, chain<DP>(qv, k) - lse_grid[gp]);
          }
        zb[pix] = (key & 0xFFFFFFFF00000000ull) | __float_as_uint(best);      // the vertex has served: its slot takes the score
      }
    }
  } else {
    // large windows (per-pixel queries: 9 x 9 crop pixels): one WAVE per hit pixel, lane = window element — consecutive
    // lanes read consecutive crop pixels (dense 48-byte rows), where one thread per output pixel made every load
    // instruction touch 64 rows 3 pixels apart.  The hit pixels are listed first so that a wave takes two at a time (their
    // loads overlap); the maximum over the window goes through DPP, not the LDS crossbar.
    for (int pix = tid; pix < n; pix += THREADS)
      if (is_hit(zb[pix])) hits[atomicAdd(&nhit_s, 1)] = (uint16_t)pix;
    __syncthreads();
    const int nhit = nhit_s, wave = tid >> 6, lane = tid & 63;
    constexpr int NW = THREADS / 64;
    // lane -> its two elements (dy, dx) of the UNCLIPPED wmax x wmax window (a third one per 128 more); elements outside the
    // grid are masked, so no division by the clipped window's width in the loop
    const int da_y = lane / wmax, da_x = lane % wmax, db_y = (lane + 64) / wmax, db_x = (lane + 64) % wmax;
    const int gmax = res * win;
    auto eval = [&](int pix, unsigned long long key, bool live) {
      float k[DP];
      load_row<DP>(keys, (size_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)key), e, k);
      const int py = pix / res, px = pix % res;
      const int yb = (pool ? py - 1 : py) * win, xb = (pool ? px - 1 : px) * win;
      float best = -__builtin_inff();
      auto element = [&](int dy, int dx, float (&qv)[DP], float* l) {
        const int gy = yb + dy, gx = xb + dx;
        const bool ok = live && dy < wmax && gy >= 0 && gy < gmax && gx >= 0 && gx < gmax;
        const size_t gp = ok ? (size_t)gy * g_pitch + gx : 0;
        load_row<DP>(qgrid, gp, e, qv);
        *l = lse_grid[gp];
        return ok;
      };
      float qa[DP], qb[DP], la, lb;
      const bool va = element(da_y, da_x, qa, &la), vb = element(db_y, db_x, qb, &lb);
      const float fa = chain<DP>(qa, k) - la, fb = chain<DP>(qb, k) - lb;
      best = fmaxf(va ? fa : best, vb ? fb : best);
      for (int i = lane + 128; i < wmax * wmax; i += 64) {         // windows of more than 128 elements (scale >= 4)
        float qc[DP], lc;
        if (element(i / wmax, i % wmax, qc, &lc)) best = fmaxf(best, chain<DP>(qc, k) - lc);
      }
      return wave_max_f32(best);
    };
    for (int h0 = wave; h0 < nhit; h0 += 2 * NW) {
      const int h1 = h0 + NW;
      const bool two = h1 < nhit;
      const int p0 = hits[h0], p1 = hits[two ? h1 : h0];
      const unsigned long long k0 = zb[p0], k1 = zb[p1];
      const float b0 = eval(p0, k0, true), b1 = eval(p1, k1, two);
      if (lane == 0) {
        zb[p0] = (k0 & 0xFFFFFFFF00000000ull) | __float_as_uint(b0);
        if (two) zb[p1] = (k1 & 0xFFFFFFFF00000000ull) | __float_as_uint(b1);
      }
    }
  }
  __syncthreads();
  double sm = 0.0, sc = 0.0, cnt = 0.0;
  if (tid < 256) {
    for (int pix = tid; pix < n; pix += 256) {
      const unsigned long long key = zb[pix];
      const bool hit = is_hit(key);
      if (hit) { sc += (double)__uint_as_float((uint32_t)key); cnt += 1.0; }
      sm += (double)(hit ? mlp[pix] : nmlp[pix]);
    }
    double v3[3] = {sm, sc, cnt};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double s = v3[i];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
      if ((tid & 63) == 0) red[i][tid >> 6] = s;
    }
  }
  __syncthreads();
  if (tid == 0) {
    const double tm = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
    const double tc = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
    const double tn = ((red[2][0] + red[2][1]) + red[2][2]) + red[2][3];
    const float ms = (float)(tm / (double)n / 0.6931471805599453);
    const float cs = tn > 0.0 ? (float)(tc / tn / log((double)m)) : -__builtin_inff();
    mask_score[b] = ms;
    coord_score[b] = cs;
    pose_score[b] = ms + cs;
  }
}

}  // namespace

extern "C" int isr_ep_prepare(const float* mask_lgts, const float* query_img, int r, int e, int scale, int max_pool,
                              float* mask_log_prob, float* neg_mask_log_prob, float* mask_prob, float* queries,
                              void* ws, size_t ws_bytes, isr_stream_t stream_) {
  ISR_REQUIRE(mask_lgts && query_img && mask_log_prob && neg_mask_log_prob && mask_prob && queries,
              "isr_ep_prepare: null pointer");
  ISR_REQUIRE(r > 0 && e > 0 && scale > 0 && r / scale > 0, "isr_ep_prepare: r=%d e=%d scale=%d", r, e, scale);
  const int res = r / scale, n = res * res;
  if (!ws || ws_bytes < 2 * sizeof(float) * (size_t)n + 512) {
    isr::set_error("isr_ep_prepare: workspace %zu < %zu", ws_bytes, 2 * sizeof(float) * (size_t)n + 512);
    return ISR_ERR_WORKSPACE;
  }
  hipStream_t stream = isr::as_stream(stream_);
  isr::Workspace w(ws, ws_bytes);
  float* lp0 = w.take<float>(n);
  float* nlp0 = w.take<float>(n);
  ep_mask_kernel<<<(n + 255) / 256, 256, 0, stream>>>(mask_lgts, r, scale, res, lp0, nlp0, mask_prob);
  ep_pool3_kernel<<<(n + 255) / 256, 256, 0, stream>>>(lp0, nlp0, res, max_pool, mask_log_prob, neg_mask_log_prob);
  const long tq = (long)n * e;
  ep_queries_kernel<<<(unsigned)((tq + 255) / 256), 256, 0, stream>>>(query_img, r, e, scale, res, queries);
  ISR_CHECK_LAUNCH("ep_prepare kernels");
  return ISR_OK;
}

extern "C" int isr_ep_pool_corr(const float* corr_log, int res, int m, float* pooled, isr_stream_t stream) {
  ISR_REQUIRE(corr_log && pooled && corr_log != pooled, "isr_ep_pool_corr: null or aliased pointer");
  ISR_REQUIRE(res > 0 && m > 0 && res * res <= 65535, "isr_ep_pool_corr: res=%d m=%d", res, m);
  ep_pool_corr_kernel<<<dim3((m + 255) / 256, res * res), 256, 0, isr::as_stream(stream)>>>(corr_log, res, m, pooled);
  ISR_CHECK_LAUNCH("ep_pool_corr_kernel");
  return ISR_OK;
}

// ------------------------------------------------------------------------ hypothesis pruning
// poseEstSurf.py:147-169, one thread per sample: the largest pairwise pixel distance of the first three
// correspondences (f32, as the reference's float32 p2d), the depth window from the object diameter, and the
// sign of normal . camera-ray at the three object points (f64: f32 points, f64 normals and poses, as there).
__global__ void ep_prune_kernel(const int64_t* __restrict__ corr_idx, const double* __restrict__ poses,
                                const uint8_t* __restrict__ ok, const float* __restrict__ obj_pts,
                                const double* __restrict__ normals, int S, int res, int m, double dist_min_px,
                                double z_min, double z_max, int do_prune, float* __restrict__ dist_2d,
                                uint8_t* __restrict__ size_mask, uint8_t* __restrict__ normals_mask,
                                uint8_t* __restrict__ keep) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  float px[3], py[3];
  int k3[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int64_t c = corr_idx[4 * (size_t)s + j];
    const int pix = (int)(c / m);
    k3[j] = (int)(c % m);
    px[j] = (float)(pix % res);
    py[j] = (float)(pix / res);
  }
  float d = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const float dx = px[a] - px[b], dy = py[a] - py[b];
      d = fmaxf(d, sqrtf(dx * dx + dy * dy));
    }
  const double* T = poses + 12 * (size_t)s;
  const double z = T[11];
  const bool sz = (z_min < z) && (z < z_max);
  bool nm = true;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double X = obj_pts[3 * (size_t)k3[j]], Y = obj_pts[3 * (size_t)k3[j] + 1], Z = obj_pts[3 * (size_t)k3[j] + 2];
    const double nx = normals[3 * (size_t)k3[j]], ny = normals[3 * (size_t)k3[j] + 1], nz = normals[3 * (size_t)k3[j] + 2];
    double dot = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double nc = (nx * T[4 * r] + ny * T[4 * r + 1]) + nz * T[4 * r + 2];                  // n3d @ R^T
      const double pc = ((X * T[4 * r] + Y * T[4 * r + 1]) + Z * T[4 * r + 2]) + T[4 * r + 3];     // p3d @ R^T + t
      dot += nc * pc;
    }
    nm = nm && (dot < 0.0);
  }
  dist_2d[s] = d;
  size_mask[s] = sz;
  normals_mask[s] = nm;
  keep[s] = ok[s] && (!do_prune || (((double)d >= dist_min_px) && sz && nm));
}

// One block: ordered compaction of the kept samples (S <= a few 10 000) and the f32 [R|t] rows of the first
// max_eval of them — the poses handed to batch_score (poseEstSurf.py:167, 174-177).
__global__ __launch_bounds__(1024) void ep_compact_kernel(const uint8_t* __restrict__ keep, const double* __restrict__ poses, int S,
                                                          int max_eval, int32_t* __restrict__ keep_idx, int32_t* __restrict__ n_keep,
                                                          float* __restrict__ Rt32) {
  __shared__ int32_t tsum[1024];
  const int t = threadIdx.x;
  const int per = (S + 1023) / 1024;
  int c = 0;
  for (int j = 0; j < per; ++j) {
    const int s = t * per + j;
    if (s < S) c += keep[s];
  }
  tsum[t] = c;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int v = (t >= off) ? tsum[t - off] : 0;
    __syncthreads();
    tsum[t] += v;
    __syncthreads();
  }
  int o = (t > 0) ? tsum[t - 1] : 0;
  for (int j = 0; j < per; ++j) {
    const int s = t * per + j;
    if (s < S && keep[s]) {
      keep_idx[o] = s;
      if (o < max_eval)
        for (int e = 0; e < 12; ++e) Rt32[12 * (size_t)o + e] = (float)poses[12 * (size_t)s + e];
      ++o;
    }
  }
  if (t == 1023) *n_keep = tsum[1023];
}

extern "C" int isr_ep_prune(const int64_t* corr_idx, const double* poses, const uint8_t* ok, const float* obj_pts,
                            const double* obj_normals, int S, int res, int m, double K00, double obj_diameter,
                            double dist_2d_min, int do_prune, int max_eval, float* dist_2d, uint8_t* size_mask,
                            uint8_t* normals_mask, uint8_t* keep, int32_t* keep_idx, int32_t* n_keep, float* Rt32,
                            isr_stream_t stream_) {
  ISR_REQUIRE(corr_idx && poses && ok && obj_pts && obj_normals && dist_2d && size_mask && normals_mask && keep && keep_idx &&
                  n_keep && Rt32, "isr_ep_prune: null pointer");
  ISR_REQUIRE(S > 0 && res > 0 && m > 0 && max_eval > 0, "isr_ep_prune: S=%d res=%d m=%d max_eval=%d", S, res, m, max_eval);
  hipStream_t stream = isr::as_stream(stream_);
  const double z_min = K00 * obj_diameter / ((double)res * 20.0), z_max = K00 * obj_diameter / ((double)res * 0.5);
  ep_prune_kernel<<<(S + 255) / 256, 256, 0, stream>>>(corr_idx, poses, ok, obj_pts, obj_normals, S, res, m,
                                                       dist_2d_min * (double)res, z_min, z_max, do_prune, dist_2d, size_mask,
                                                       normals_mask, keep);
  ep_compact_kernel<<<1, 1024, 0, stream>>>(keep, poses, S, max_eval, keep_idx, n_keep, Rt32);
  ISR_CHECK_LAUNCH("ep_prune kernels");
  return ISR_OK;
}

extern "C" int isr_ep_patch_corr_cells(const float* query_img, const float* obj_keys, int r, int e, int scale, int m,
                                       float* corr_centre, float* corr_blockmax, isr_stream_t stream) {
  ISR_REQUIRE(query_img && obj_keys && corr_centre && corr_blockmax, "isr_ep_patch_corr: null pointer");
  ISR_REQUIRE(r > 0 && m > 0 && e > 0 && e <= kMaxE && scale >= 1 && scale * scale <= kMaxBlockPix && r / scale > 0,
              "isr_ep_patch_corr: r=%d e=%d (<= %d) scale=%d (<= 4) m=%d", r, e, kMaxE, scale, m);
  const int res = r / scale;
  ep_patch_corr_kernel<<<res * res, 256, 0, isr::as_stream(stream)>>>(query_img, obj_keys, r, e, scale, res, m, corr_centre,
                                                                      corr_blockmax);
  ISR_CHECK_LAUNCH("ep_patch_corr_kernel");
  return ISR_OK;
}

extern "C" size_t isr_ep_sample_workspace_bytes(int n, int m) {
  if (n <= 0 || m <= 0) return 0;
  const size_t nchunk = ((size_t)m + kChunk - 1) / kChunk;
  return isr::align_up(sizeof(double) * n * nchunk, 256) + 3 * isr::align_up(sizeof(double) * n, 256) + 512;
}

namespace {

struct SampleWs { double *chunk_sums, *row_sums, *row_cum, *mpa; int nchunk; };

int sample_ws(const char* who, const float* mask_prob, int n, int m, double alpha, void* ws, size_t ws_bytes, hipStream_t stream,
              SampleWs* out) {
  if (!ws || ws_bytes < isr_ep_sample_workspace_bytes(n, m)) {
    isr::set_error("%s: workspace %zu < %zu", who, ws_bytes, isr_ep_sample_workspace_bytes(n, m));
    return ISR_ERR_WORKSPACE;
  }
  isr::Workspace w(ws, ws_bytes);
  out->nchunk = (m + kChunk - 1) / kChunk;
  out->chunk_sums = w.take<double>((size_t)n * out->nchunk);
  out->row_sums = w.take<double>(n);
  out->row_cum = w.take<double>(n);
  out->mpa = w.take<double>(n);
  ep_mpa_kernel<<<(n + 255) / 256, 256, 0, stream>>>(mask_prob, n, alpha, out->mpa);
  return ISR_OK;
}

void sample_rows(const SampleWs& sw, int n, hipStream_t stream) {
  ep_row_sums_kernel<<<(n + 3) / 4, 256, 0, stream>>>(sw.chunk_sums, n, sw.nchunk, sw.row_sums);
  ep_row_scan_kernel<<<1, 1024, 0, stream>>>(sw.row_sums, n, sw.row_cum);
}

}  // namespace

extern "C" int isr_ep_sample(const float* corr_log, const float* mask_prob, int n, int m, double alpha, int n_samples,
                             uint64_t seed, int64_t* corr_idx, void* ws, size_t ws_bytes, isr_stream_t stream_) {
  ISR_REQUIRE(corr_log && mask_prob && corr_idx, "isr_ep_sample: null pointer");
  ISR_REQUIRE(n > 0 && m > 0 && n_samples > 0, "isr_ep_sample: n=%d m=%d n_samples=%d", n, m, n_samples);
  hipStream_t stream = isr::as_stream(stream_);
  SampleWs sw;
  const int rc = sample_ws("isr_ep_sample", mask_prob, n, m, alpha, ws, ws_bytes, stream, &sw);
  if (rc != ISR_OK) return rc;
  ep_chunk_sums_kernel<<<dim3(sw.nchunk, (n + kRowsPerBlock - 1) / kRowsPerBlock), kRowsPerBlock, 0, stream>>>(
      corr_log, sw.mpa, n, m, sw.nchunk, alpha, sw.chunk_sums);
  sample_rows(sw, n, stream);
  ep_sample_kernel<0><<<n_samples, 256, 0, stream>>>(corr_log, QGrid{}, nullptr, sw.mpa, n, m, sw.nchunk, alpha, sw.chunk_sums,
                                                     sw.row_cum, n_samples, (uint32_t)seed, (uint32_t)(seed >> 32), corr_idx);
  ISR_CHECK_LAUNCH("ep_sample kernels");
  return ISR_OK;
}

extern "C" int isr_ep_sample_direct(const float* qgrid, const float* lse_grid, int g_pitch, int e, int win, int res,
                                    const float* mask_prob, const float* keys, int m, double alpha, int n_samples,
                                    uint64_t seed, int64_t* corr_idx, void* ws, size_t ws_bytes, isr_stream_t stream_) {
  ISR_REQUIRE(qgrid && lse_grid && mask_prob && keys && corr_idx, "isr_ep_sample_direct: null pointer");
  ISR_REQUIRE(res > 0 && m > 0 && n_samples > 0 && e > 0 && e <= 128 && win >= 1 && g_pitch >= res * win,
              "isr_ep_sample_direct: res=%d m=%d n_samples=%d e=%d (<= 128) win=%d g_pitch=%d (>= res * win)", res, m, n_samples,
              e, win, g_pitch);
  hipStream_t stream = isr::as_stream(stream_);
  const int n = res * res;
  SampleWs sw;
  const int rc = sample_ws("isr_ep_sample_direct", mask_prob, n, m, alpha, ws, ws_bytes, stream, &sw);
  if (rc != ISR_OK) return rc;
  const QGrid g{qgrid, lse_grid, e, res, win * g_pitch, win, (win / 2) * (g_pitch + 1)};
  const dim3 grid(sw.nchunk, (n + kRowsPerBlock - 1) / kRowsPerBlock);
  const uint32_t slo = (uint32_t)seed, shi = (uint32_t)(seed >> 32);
  const bool valu = isr::tuning(ISR_TUNE_EP_WSUM_VALU) != 0;
  const dim3 grid_mfma(sw.nchunk, (n + 127) / 128);
#define ISR_EP_DIRECT(DPv)                                                                                                    \
  do {                                                                                                                        \
    if (valu)                                                                                                                 \
      ep_chunk_sums_direct_kernel<DPv><<<grid, kRowsPerBlock, 0, stream>>>(g, sw.mpa, keys, n, m, sw.nchunk, alpha, sw.chunk_sums); \
    else                                                                                                                      \
      ep_chunk_sums_mfma_kernel<DPv><<<grid_mfma, 256, 0, stream>>>(g, sw.mpa, keys, n, m, sw.nchunk, alpha, sw.chunk_sums);   \
    sample_rows(sw, n, stream);                                                                                               \
    ep_sample_kernel<DPv><<<n_samples, 256, 0, stream>>>(nullptr, g, keys, sw.mpa, n, m, sw.nchunk, alpha, sw.chunk_sums,       \
                                                         sw.row_cum, n_samples, slo, shi, corr_idx);                          \
  } while (0)
  if (e == 12) ISR_EP_DIRECT(12);          // the reference's descriptors (12 channels): no padded lanes in the chain
  else if (e <= 16) ISR_EP_DIRECT(16);
  else if (e <= 32) ISR_EP_DIRECT(32);
  else if (e <= 64) ISR_EP_DIRECT(64);
  else ISR_EP_DIRECT(128);
#undef ISR_EP_DIRECT
  ISR_CHECK_LAUNCH("ep_sample_direct kernels");
  return ISR_OK;
}

extern "C" int isr_ep_sample_weights(const float* corr_log, const float* mask_prob, int n, int m, double alpha, double* weights,
                                     void* ws, size_t ws_bytes, isr_stream_t stream_) {
  ISR_REQUIRE(corr_log && mask_prob && weights, "isr_ep_sample_weights: null pointer");
  ISR_REQUIRE(n > 0 && m > 0, "isr_ep_sample_weights: n=%d m=%d", n, m);
  hipStream_t stream = isr::as_stream(stream_);
  SampleWs sw;
  const int rc = sample_ws("isr_ep_sample_weights", mask_prob, n, m, alpha, ws, ws_bytes, stream, &sw);
  if (rc != ISR_OK) return rc;
  const size_t tot = (size_t)n * m;
  ep_weights_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, stream>>>(corr_log, sw.mpa, n, m, alpha, weights);
  ISR_CHECK_LAUNCH("ep_weights_kernel");
  return ISR_OK;
}

extern "C" int isr_ep_p3p(const int64_t* corr_idx, int res, int m, const float* obj_pts, const double* Kcam, int S,
                          uint64_t seed, double* poses, uint8_t* ok, isr_stream_t stream) {
  ISR_REQUIRE(corr_idx && obj_pts && Kcam && poses && ok, "isr_ep_p3p: null pointer");
  ISR_REQUIRE(res > 0 && m > 0 && S > 0, "isr_ep_p3p: res=%d m=%d S=%d", res, m, S);
  Cam cam;
  ISR_REQUIRE(make_cam(Kcam, &cam), "isr_ep_p3p: singular camera matrix");
  ep_p3p_kernel<<<(S + 63) / 64, 64, 0, isr::as_stream(stream)>>>(corr_idx, res, m, obj_pts, cam, S, (uint32_t)seed,
                                                                 (uint32_t)(seed >> 32), poses, ok);
  ISR_CHECK_LAUNCH("ep_p3p_kernel");
  return ISR_OK;
}

extern "C" size_t isr_zbuf_score_workspace_bytes(int B, int res) {
  if (B <= 0 || res <= 0) return 0;
  return sizeof(unsigned long long) * (size_t)B * res * res + 512;
}

extern "C" int isr_zbuf_score(const float* obj_pts, int m, const float* Rt, int B, const double* Kcam, int res,
                              const float* mask_log_prob, const float* neg_mask_log_prob, const float* corr_log,
                              float* pose_score, float* mask_score, float* coord_score, void* ws, size_t ws_bytes,
                              isr_stream_t stream_) {
  ISR_REQUIRE(obj_pts && Rt && Kcam && mask_log_prob && neg_mask_log_prob && corr_log && pose_score && mask_score &&
                  coord_score, "isr_zbuf_score: null pointer");
  ISR_REQUIRE(m > 0 && B > 0 && B <= 65535 && res > 0, "isr_zbuf_score: m=%d B=%d res=%d", m, B, res);
  if (!ws || ws_bytes < isr_zbuf_score_workspace_bytes(B, res)) {
    isr::set_error("isr_zbuf_score: workspace %zu < %zu", ws_bytes, isr_zbuf_score_workspace_bytes(B, res));
    return ISR_ERR_WORKSPACE;
  }
  hipStream_t stream = isr::as_stream(stream_);
  const int n = res * res;
  isr::Workspace w(ws, ws_bytes);
  unsigned long long* zbuf = w.take<unsigned long long>((size_t)B * n);
  ISR_CHECK_HIP(hipMemsetAsync(zbuf, 0xFF, sizeof(unsigned long long) * (size_t)B * n, stream));
  K9f K;
  for (int i = 0; i < 9; ++i) K.k[i] = (float)Kcam[i];
  zbuf_project_kernel<<<dim3((m + 255) / 256, B), 256, 0, stream>>>(obj_pts, m, Rt, B, K, res, zbuf);
  zbuf_score_kernel<<<B, 256, 0, stream>>>(zbuf, n, m, mask_log_prob, neg_mask_log_prob, corr_log, pose_score, mask_score,
                                           coord_score);
  ISR_CHECK_LAUNCH("zbuf kernels");
  return ISR_OK;
}

extern "C" int isr_zbuf_score_direct(const float* obj_pts, int m, const float* Rt, int B, const double* Kcam, int res,
                                     const float* mask_log_prob, const float* neg_mask_log_prob, const float* qgrid,
                                     const float* lse_grid, int g_pitch, int e, int win, int pool, const float* keys,
                                     float* pose_score, float* mask_score, float* coord_score, void* ws, size_t ws_bytes,
                                     isr_stream_t stream_) {
  ISR_REQUIRE(obj_pts && Rt && Kcam && mask_log_prob && neg_mask_log_prob && qgrid && lse_grid && keys && pose_score &&
                  mask_score && coord_score, "isr_zbuf_score_direct: null pointer");
  ISR_REQUIRE(m > 0 && B > 0 && B <= 65535 && res > 0 && e > 0 && e <= 128 && win >= 1 && g_pitch >= res * win,
              "isr_zbuf_score_direct: m=%d B=%d res=%d e=%d (<= 128) win=%d g_pitch=%d (>= res * win)", m, B, res, e, win, g_pitch);
  hipStream_t stream = isr::as_stream(stream_);
  const int n = res * res;
  K9f K;
  for (int i = 0; i < 9; ++i) K.k[i] = (float)Kcam[i];
  const size_t lds = (size_t)n * (sizeof(unsigned long long) + sizeof(uint16_t));
  if (lds <= kFusedLdsMax && n <= 65535) {       // the z-buffer lives in LDS: one launch, no scratch
    // 1 024-thread workgroups are resident one per CU (measured): best while the poses fit the chip in one round
    const bool wide = B <= 256;
#define ISR_ZB_FUSED_T(DPv, TH)                                                                                               \
  do {                                                                                                                        \
    if (lds > 64 * 1024)     /* per device, so not cached in a static */                                                      \
      ISR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&zbuf_fused_direct_kernel<DPv, TH>),                    \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFusedLdsMax));                     \
    zbuf_fused_direct_kernel<DPv, TH><<<B, TH, lds, stream>>>(obj_pts, m, Rt, K, res, mask_log_prob, neg_mask_log_prob, qgrid, \
                                                              lse_grid, g_pitch, e, win, pool ? 1 : 0, keys, pose_score,      \
                                                              mask_score, coord_score);                                       \
  } while (0)
#define ISR_ZB_FUSED(DPv)                                                                                                     \
  do {                                                                                                                        \
    if (wide) ISR_ZB_FUSED_T(DPv, 1024);                                                                                      \
    else ISR_ZB_FUSED_T(DPv, 512);                                                                                            \
  } while (0)
    if (e == 12) ISR_ZB_FUSED(12);
    else if (e <= 16) ISR_ZB_FUSED(16);
    else if (e <= 32) ISR_ZB_FUSED(32);
    else if (e <= 64) ISR_ZB_FUSED(64);
    else ISR_ZB_FUSED(128);
#undef ISR_ZB_FUSED
#undef ISR_ZB_FUSED_T
    ISR_CHECK_LAUNCH("zbuf_fused_direct_kernel");
    return ISR_OK;
  }
  if (!ws || ws_bytes < isr_zbuf_score_workspace_bytes(B, res)) {
    isr::set_error("isr_zbuf_score_direct: workspace %zu < %zu", ws_bytes, isr_zbuf_score_workspace_bytes(B, res));
    return ISR_ERR_WORKSPACE;
  }
  isr::Workspace w(ws, ws_bytes);
  unsigned long long* zbuf = w.take<unsigned long long>((size_t)B * n);
  ISR_CHECK_HIP(hipMemsetAsync(zbuf, 0xFF, sizeof(unsigned long long) * (size_t)B * n, stream));
  zbuf_project_kernel<<<dim3((m + 255) / 256, B), 256, 0, stream>>>(obj_pts, m, Rt, B, K, res, zbuf);
#define ISR_ZB_DIRECT(DPv)                                                                                                    \
  zbuf_score_direct_kernel<DPv><<<B, 256, 0, stream>>>(zbuf, res, m, mask_log_prob, neg_mask_log_prob, qgrid, lse_grid, g_pitch, e, \
                                                       win, pool ? 1 : 0, keys, pose_score, mask_score, coord_score)
  if (e <= 16) ISR_ZB_DIRECT(16);
  else if (e <= 32) ISR_ZB_DIRECT(32);
  else if (e <= 64) ISR_ZB_DIRECT(64);
  else ISR_ZB_DIRECT(128);
#undef ISR_ZB_DIRECT
  ISR_CHECK_LAUNCH("zbuf direct kernels");
  return ISR_OK;
}

// ------------------------------------------------------------------------------------------ the whole call
// estimate_pose (poseEstSurf.py:11-261) as ONE entry point: pooling, the rows' log-sum-exps (K1), matrix-free sampling, P3P,
// pruning and ordered selection, matrix-free scoring — the launches pose_est_surf.estimate_pose issues stage by stage, from
// one host call, with every intermediate carved from `ws`.  The stream is synchronised ONCE, where the reference's Python sizes
// its outputs: the number of surviving poses.
namespace {

struct EpLayout {
  float *mlp, *nmlp, *mprob, *queries, *lse;
  int32_t* k1_idx;
  int64_t* corr_idx;
  double* poses;
  uint8_t* keep;
  int32_t *keep_idx, *n_keep;
  void *ws_prep, *ws_k1, *ws_sample;
  size_t b_prep, b_k1, b_sample;
};

size_t ep_layout(isr::Workspace& w, int r, int e, int m, int scale, int max_poses, int avg_queries, EpLayout* L) {
  const int res = r / scale, n = res * res;
  const int rows = avg_queries ? n : r * r;                 // descriptor rows whose log-sum-exps K1 computes
  L->mlp = w.take<float>(n);
  L->nmlp = w.take<float>(n);
  L->mprob = w.take<float>(n);
  L->queries = w.take<float>((size_t)n * e);
  L->lse = w.take<float>(rows);
  L->k1_idx = w.take<int32_t>(rows);
  L->corr_idx = w.take<int64_t>((size_t)max_poses * 4);
  L->poses = w.take<double>((size_t)max_poses * 12);
  L->keep = w.take<uint8_t>(max_poses);
  L->keep_idx = w.take<int32_t>(max_poses);
  L->n_keep = w.take<int32_t>(4);
  L->b_prep = 2 * sizeof(float) * (size_t)n + 1024;
  L->ws_prep = w.take<char>(L->b_prep);
  L->b_k1 = isr_corr_argmax_workspace_bytes(rows, m, e, ISR_DTYPE_F32);
  L->ws_k1 = w.take<char>(L->b_k1);
  L->b_sample = isr_ep_sample_workspace_bytes(n, m);
  L->ws_sample = w.take<char>(L->b_sample);
  return w.off;
}

}  // namespace

extern "C" size_t isr_estimate_pose_workspace_bytes(int r, int e, int m, int scale, int max_poses, int max_pose_evaluations,
                                                    int avg_queries) {
  if (r <= 0 || e <= 0 || m <= 0 || scale <= 0 || r / scale <= 0 || max_poses <= 0 || max_pose_evaluations <= 0) return 0;
  isr::Workspace w(nullptr, 0);
  EpLayout L;
  size_t bytes = ep_layout(w, r, e, m, scale, max_poses, avg_queries, &L);
  const size_t zb = isr_zbuf_score_workspace_bytes(max_pose_evaluations, r / scale);      // only used when the z-buffer does not fit the LDS
  return bytes + zb + 1024;
}

extern "C" int isr_estimate_pose(const float* mask_lgts, const float* query_img, int r, int e, const float* obj_pts,
                                 const double* obj_normals, const float* obj_keys, int m, double obj_diameter,
                                 const double* Kcam, int max_poses, int max_pose_evaluations, int down_sample_scale,
                                 double alpha, double dist_2d_min, int max_pool, int avg_queries, int do_prune, uint64_t seed,
                                 float* Rt32, float* pose_scores, float* mask_scores, float* coord_scores, float* dist_2d,
                                 uint8_t* size_mask, uint8_t* normals_mask, uint8_t* solved, int32_t* n_poses_host,
                                 int32_t* n_keep_host, void* ws, size_t ws_bytes, isr_stream_t stream_) {
  ISR_REQUIRE(mask_lgts && query_img && obj_pts && obj_normals && obj_keys && Kcam && Rt32 && pose_scores && mask_scores &&
                  coord_scores && dist_2d && size_mask && normals_mask && solved && n_poses_host,
              "isr_estimate_pose: null pointer");
  ISR_REQUIRE(r > 0 && e > 0 && e <= 128 && m > 0 && down_sample_scale > 0 && r / down_sample_scale > 0 && max_poses > 0 &&
                  max_pose_evaluations > 0 && max_pose_evaluations <= 65535,
              "isr_estimate_pose: r=%d e=%d (<= 128) m=%d scale=%d max_poses=%d max_pose_evaluations=%d (<= 65535)", r, e, m,
              down_sample_scale, max_poses, max_pose_evaluations);
  const size_t need = isr_estimate_pose_workspace_bytes(r, e, m, down_sample_scale, max_poses, max_pose_evaluations, avg_queries);
  if (!ws || ws_bytes < need) {
    isr::set_error("isr_estimate_pose: workspace %zu < %zu", ws_bytes, need);
    return ISR_ERR_WORKSPACE;
  }
  hipStream_t stream = isr::as_stream(stream_);
  const int scale = down_sample_scale, res = r / scale, n = res * res;
  isr::Workspace w(ws, ws_bytes);
  EpLayout L;
  ep_layout(w, r, e, m, scale, max_poses, avg_queries, &L);
  void* ws_zb = w.take<char>(isr_zbuf_score_workspace_bytes(max_pose_evaluations, res));
  const size_t b_zb = isr_zbuf_score_workspace_bytes(max_pose_evaluations, res);
  // the camera of the pooled grid (poseEstSurf.py:42-45)
  double Ks[9];
  for (int i = 0; i < 9; ++i) Ks[i] = Kcam[i];
  Ks[2] += 0.5; Ks[5] += 0.5;
  for (int i = 0; i < 6; ++i) Ks[i] /= (double)scale;
  Ks[2] -= 0.5; Ks[5] -= 0.5;
  int rc = isr_ep_prepare(mask_lgts, query_img, r, e, scale, max_pool, L.mlp, L.nmlp, L.mprob, L.queries, L.ws_prep, L.b_prep, stream_);
  if (rc != ISR_OK) return rc;
  // descriptor grid: pooled queries (win 1) or the crop's own pixels (win = scale)
  const float* qgrid = avg_queries ? L.queries : query_img;
  const int rows = avg_queries ? n : r * r, pitch = avg_queries ? res : r, win = avg_queries ? 1 : scale;
  // the rows' log-sum-exps: an lse-only call of K1 (idx == nullptr)
  rc = isr_corr_argmax(qgrid, obj_keys, rows, m, e, e, e, ISR_DTYPE_F32, nullptr, nullptr, L.lse, L.ws_k1, L.b_k1, stream_);
  if (rc != ISR_OK) return rc;
  rc = isr_ep_sample_direct(qgrid, L.lse, pitch, e, win, res, L.mprob, obj_keys, m, alpha, max_poses, seed, L.corr_idx, L.ws_sample,
                            L.b_sample, stream_);
  if (rc != ISR_OK) return rc;
  rc = isr_ep_p3p(L.corr_idx, res, m, obj_pts, Ks, max_poses, seed, L.poses, solved, stream_);
  if (rc != ISR_OK) return rc;
  rc = isr_ep_prune(L.corr_idx, L.poses, solved, obj_pts, obj_normals, max_poses, res, m, Ks[0], obj_diameter, dist_2d_min, do_prune,
                    max_pose_evaluations, dist_2d, size_mask, normals_mask, L.keep, L.keep_idx, L.n_keep, Rt32, stream_);
  if (rc != ISR_OK) return rc;
  int32_t n_keep = 0;
  ISR_CHECK_HIP(hipMemcpyAsync(&n_keep, L.n_keep, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  ISR_CHECK_HIP(hipStreamSynchronize(stream));                 // the one round trip: it sizes the outputs
  const int n_poses = n_keep < max_pose_evaluations ? n_keep : max_pose_evaluations;
  *n_poses_host = n_poses;
  if (n_keep_host) *n_keep_host = n_keep;
  if (n_poses > 0) {
    rc = isr_zbuf_score_direct(obj_pts, m, Rt32, n_poses, Ks, res, L.mlp, L.nmlp, qgrid, L.lse, pitch, e, win, max_pool, obj_keys,
                               pose_scores, mask_scores, coord_scores, ws_zb, b_zb, stream_);
    if (rc != ISR_OK) return rc;
  }
  return ISR_OK;
}

