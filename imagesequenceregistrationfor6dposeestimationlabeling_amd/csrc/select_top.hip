// select_top.hip — a2/a3: the "top 80 %" correspondence filter and correspondence assembly.
//
// Replaces (inference.py:282-290, = finalposes.py:221-229 = choosePose.py:289-297):
//     perc = int(0.8 * n); thr = torch.sort(in1[:, 0])[0][-perc + 1]      (n > 500)
//     thr = torch.sort(in1[:, 0])[0][-n + 1]                               (otherwise)
//     nidx = torch.where(in1[:, 0] > thr)[0]
// and the gathers of inference.py:274-280.
// One order statistic does not need a sort: a 3-pass (11/11/10-bit) radix select over the
// order-preserving integer image of the floats finds thr exactly; an ordered stream compaction
// (per-block counts -> scan -> scatter) writes the kept indices ascending, as torch.where does.
// HBM-bound integer work: 4 coalesced reads of logp (3 histogram passes + 1 flag pass) plus one
// more in the scatter; histograms use LDS atomics then one integer global atomic per bin per
// block, so results are deterministic.  The kept count stays on the device (M_dev).
// Every kernel carries an IMAGE dimension (blockIdx.z): the per-image loop of inference.py:163 over a
// group of images becomes ten launches for the whole group (isr_select_top_batch); the single-image
// entry points are the B = 1 case of the same kernels, so results are bit-identical by construction.
#include "isr_common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kPerThread = 8;
constexpr int kChunk = kThreads * kPerThread;  // elements per block

struct SelState {
  uint32_t prefix;     // bits fixed so far (aligned at their final position)
  uint32_t mask;       // which bits are fixed
  int32_t k;           // remaining rank inside the current bucket
  uint32_t thr_bits;   // float bits of thr once known
  int32_t n;           // number of elements (<= capacity)
};

__device__ __forceinline__ uint32_t ordered(float f) { return isr::ordered_bits(f); }
__device__ __forceinline__ float unordered(uint32_t u) {
  const uint32_t b = u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu);
  return __uint_as_float(b);
}

constexpr int kHistInts = 3 * 2048;   // the three histograms of one image

template <int SHIFT, int BITS>
__global__ __launch_bounds__(kThreads) void hist_kernel(const float* __restrict__ x, int64_t ld,
                                                        const SelState* __restrict__ st,
                                                        int32_t* __restrict__ hist) {
  constexpr int NB = 1 << BITS;
  __shared__ int32_t h[NB];
  x += blockIdx.z * ld; st += blockIdx.z; hist += (size_t)blockIdx.z * kHistInts;
  const int P = st->n;   // the element count: the host's P, or the device-side count (isr_select_top_dev)
  if (blockIdx.x * kChunk >= P) return;   // block-uniform: nothing of this image falls in this block
  for (int i = threadIdx.x; i < NB; i += kThreads) h[i] = 0;
  __syncthreads();
  const uint32_t prefix = st->prefix, mask = st->mask;
  const int base = blockIdx.x * kChunk;
#pragma unroll
  for (int e = 0; e < kPerThread; ++e) {
    const int i = base + e * kThreads + threadIdx.x;
    if (i < P) {
      const uint32_t u = ordered(x[i]);
      if ((u & mask) == prefix) atomicAdd(&h[(u >> SHIFT) & (NB - 1)], 1);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NB; i += kThreads)
    if (h[i]) atomicAdd(&hist[i], h[i]);
}

// One block: find the bin holding rank k, narrow the prefix.  Serial over <= 2048 bins by one
// wave's lanes in chunks — negligible next to the passes over logp.
template <int SHIFT, int BITS, bool LAST>
__global__ void pick_kernel(const int32_t* __restrict__ hist, SelState* __restrict__ st,
                            float* __restrict__ thr_out) {
  constexpr int NB = 1 << BITS;
  hist += (size_t)blockIdx.z * kHistInts; st += blockIdx.z;
  if (thr_out) thr_out += blockIdx.z;
  __shared__ int32_t cum[NB];
  // inclusive scan, 1024 threads x (NB/1024) bins
  const int t = threadIdx.x;
  constexpr int PER = (NB + 1023) / 1024;
  int32_t loc[PER];
  int32_t s = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int b = t * PER + j;
    s += (b < NB) ? hist[b] : 0;
    loc[j] = s;
  }
  __shared__ int32_t tsum[1024];
  tsum[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const int32_t v = (t >= off) ? tsum[t - off] : 0;
    __syncthreads();
    tsum[t] += v;
    __syncthreads();
  }
  const int32_t before = (t > 0) ? tsum[t - 1] : 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int b = t * PER + j;
    if (b < NB) cum[b] = before + loc[j];
  }
  __syncthreads();
  const int32_t k = st->k;
  if (LAST && st->n == 0 && t == 0) {               // empty input: keep nothing
    st->thr_bits = __float_as_uint(__builtin_inff());
    if (thr_out) *thr_out = __builtin_inff();
  }
  // the unique bin b with cum[b-1] <= k < cum[b]
  for (int b = t; b < NB; b += 1024) {
    const int32_t lo = b ? cum[b - 1] : 0;
    if (lo <= k && k < cum[b]) {
      const uint32_t prefix = st->prefix | ((uint32_t)b << SHIFT);
      st->prefix = prefix;
      st->mask = st->mask | ((uint32_t)(NB - 1) << SHIFT);
      st->k = k - lo;
      if (LAST) {
        const float thr = unordered(prefix);
        st->thr_bits = __float_as_uint(thr);
        if (thr_out) *thr_out = thr;
      }
    }
  }
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

// Thread t of a block owns the kPerThread CONSECUTIVE elements base + t*kPerThread .. (ordered).
__global__ __launch_bounds__(kThreads) void count_kernel(const float* __restrict__ x, int64_t ld,
                                                         const SelState* __restrict__ st,
                                                         int32_t* __restrict__ block_counts) {
  x += blockIdx.z * ld; st += blockIdx.z; block_counts += (size_t)blockIdx.z * gridDim.x;
  const float thr = __uint_as_float(st->thr_bits);
  const int P = st->n;
  const int i0 = blockIdx.x * kChunk + threadIdx.x * kPerThread;
  int c = 0;
#pragma unroll
  for (int e = 0; e < kPerThread; ++e)
    if (i0 + e < P) c += x[i0 + e] > thr;
  int total;
  (void)block_exclusive_scan(c, &total);
  if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}

__global__ void scan_blocks_kernel(int32_t* __restrict__ block_counts, int nblocks,
                                   int32_t* __restrict__ M_dev) {
  // one block of 1024 threads per image; each thread scans a contiguous run
  __shared__ int32_t tsum[1024];
  block_counts += (size_t)blockIdx.z * nblocks; M_dev += blockIdx.z;
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
      block_counts[b] = run;  // exclusive offset
      run += c;
    }
  }
  if (t == 1023) *M_dev = tsum[1023];
}

__global__ __launch_bounds__(kThreads) void scatter_kernel(const float* __restrict__ x, int64_t ld,
                                                           const SelState* __restrict__ st,
                                                           const int32_t* __restrict__ block_off,
                                                           int32_t* __restrict__ keep, int64_t ldkeep) {
  x += blockIdx.z * ld; st += blockIdx.z; block_off += (size_t)blockIdx.z * gridDim.x; keep += blockIdx.z * ldkeep;
  const float thr = __uint_as_float(st->thr_bits);
  const int P = st->n;
  const int i0 = blockIdx.x * kChunk + threadIdx.x * kPerThread;
  bool f[kPerThread];
  int c = 0;
#pragma unroll
  for (int e = 0; e < kPerThread; ++e) {
    f[e] = (i0 + e < P) && (x[i0 + e] > thr);
    c += f[e];
  }
  int total;
  int o = block_off[blockIdx.x] + block_exclusive_scan(c, &total);
#pragma unroll
  for (int e = 0; e < kPerThread; ++e)
    if (f[e]) keep[o++] = i0 + e;
}

// rank into the ascending order, with Python's negative-index semantics (inference.py:282-287):
//   n > min_n: perc = int(frac n), sorted[-perc + 1];  otherwise sorted[-n + 1]
__host__ __device__ inline long select_rank(long n, double frac, int min_n) {
  if (n > min_n) {
    const long perc = (long)(frac * (double)n);
    if (perc == 1) return 0;                       // [-1 + 1] = [0]
    return (perc >= 1) ? n - perc + 1 : 1;         // [-0 + 1] is index 1
  }
  return (n >= 2) ? 1 : 0;
}

// n_dev == nullptr: n = P (checked on the host).  Otherwise n = min(P, *n_dev); n = 0, or a rank the
// reference would raise IndexError for, selects nothing: thr = +inf.
// One block per image: zeroes the image's three histograms (no memset launch) and sets its state.
// digits != nullptr: the image's first histogram was formed by K1's epilogue (isr_corr_argmax_digits) and is copied in.
__global__ void init_state_kernel(SelState* st, int32_t* __restrict__ hist, int P, const int32_t* __restrict__ n_dev,
                                  double frac, int min_n, const int32_t* __restrict__ digits) {
  st += blockIdx.z; hist += (size_t)blockIdx.z * kHistInts;
  if (digits) digits += (size_t)blockIdx.z * isr::kDigitBins;
  for (int i = threadIdx.x; i < kHistInts; i += blockDim.x) hist[i] = (digits && i < isr::kDigitBins) ? digits[i] : 0;
  if (threadIdx.x != 0) return;
  int n = P;
  if (n_dev) n = min(P, max(0, n_dev[blockIdx.z]));
  const long rank = select_rank(n, frac, min_n);
  st->prefix = 0; st->mask = 0; st->thr_bits = 0;
  if (n <= 0 || rank < 0 || rank >= n) {
    st->n = 0;                                     // every later pass sees an empty input
    st->k = 0;
  } else {
    st->n = n;
    st->k = (int32_t)rank;
  }
}

__global__ void gather_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ keep,
                              const int32_t* __restrict__ M_dev, const float* __restrict__ pts,
                              const float* __restrict__ pix_xy, float* __restrict__ p3d,
                              float* __restrict__ p2d, int64_t P, int64_t pix_stride) {
  const int b = blockIdx.z;
  idx += b * P; keep += b * P; pix_xy += b * pix_stride; p3d += b * P * 3; p2d += b * P * 2;
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M_dev[b]) return;
  const int p = keep[m];
  const int k = idx[p];
  p3d[3 * (size_t)m] = pts[3 * (size_t)k];
  p3d[3 * (size_t)m + 1] = pts[3 * (size_t)k + 1];
  p3d[3 * (size_t)m + 2] = pts[3 * (size_t)k + 2];
  p2d[2 * (size_t)m] = pix_xy[2 * (size_t)p];
  p2d[2 * (size_t)m + 1] = pix_xy[2 * (size_t)p + 1];
}

}  // namespace

static size_t select_ws_bytes(int P, int B) {
  const size_t nblocks = ((size_t)P + kChunk - 1) / kChunk;
  return isr::align_up(sizeof(SelState) * B, 256) + isr::align_up((size_t)kHistInts * 4 * B, 256) +
         isr::align_up(nblocks * 4 * B, 256) + 256;
}

extern "C" size_t isr_select_top_workspace_bytes(int P) {
  if (P <= 0) return 0;
  return select_ws_bytes(P, 1);
}

extern "C" size_t isr_select_top_batch_workspace_bytes(int P, int B) {
  if (P <= 0 || B <= 0) return 0;
  return select_ws_bytes(P, B);
}

static_assert(isr::kDigitBins == 2048 && isr::kDigitShift == 21, "the first pass of the radix select is hist_kernel<21, 11>");

static int select_top_impl(const float* logp, int P, int64_t ld, int B, const int32_t* n_dev, double frac, int min_n,
                           int32_t* keep, int32_t* M_dev, float* thr_dev, void* ws, size_t ws_bytes,
                           isr_stream_t stream_, const int32_t* digits = nullptr) {
  if (!ws || ws_bytes < select_ws_bytes(P, B)) {
    isr::set_error("isr_select_top: workspace %zu < %zu", ws_bytes, select_ws_bytes(P, B));
    return ISR_ERR_WORKSPACE;
  }
  hipStream_t stream = isr::as_stream(stream_);
  isr::Workspace w(ws, ws_bytes);
  SelState* st = w.take<SelState>(B);
  int32_t* h0 = w.take<int32_t>((size_t)kHistInts * B);
  int32_t* h1 = h0 + 2048;
  int32_t* h2 = h0 + 4096;
  const int nblocks = (P + kChunk - 1) / kChunk;
  int32_t* bc = w.take<int32_t>((size_t)nblocks * B);
  const dim3 gI(1, 1, B), gP(nblocks, 1, B);

  init_state_kernel<<<gI, 256, 0, stream>>>(st, h0, P, n_dev, frac, min_n, digits);
  if (!digits) hist_kernel<21, 11><<<gP, kThreads, 0, stream>>>(logp, ld, st, h0);
  pick_kernel<21, 11, false><<<gI, 1024, 0, stream>>>(h0, st, nullptr);
  hist_kernel<10, 11><<<gP, kThreads, 0, stream>>>(logp, ld, st, h1);
  pick_kernel<10, 11, false><<<gI, 1024, 0, stream>>>(h1, st, nullptr);
  hist_kernel<0, 10><<<gP, kThreads, 0, stream>>>(logp, ld, st, h2);
  pick_kernel<0, 10, true><<<gI, 1024, 0, stream>>>(h2, st, thr_dev);
  count_kernel<<<gP, kThreads, 0, stream>>>(logp, ld, st, bc);
  scan_blocks_kernel<<<gI, 1024, 0, stream>>>(bc, nblocks, M_dev);
  scatter_kernel<<<gP, kThreads, 0, stream>>>(logp, ld, st, bc, keep, (int64_t)P);
  ISR_CHECK_LAUNCH("select_top kernels");
  return ISR_OK;
}

extern "C" int isr_select_top(const float* logp, int P, double frac, int min_n, int32_t* keep,
                              int32_t* M_dev, float* thr_dev, void* ws, size_t ws_bytes,
                              isr_stream_t stream_) {
  ISR_REQUIRE(logp && keep && M_dev, "isr_select_top: null pointer");
  ISR_REQUIRE(P > 0, "isr_select_top: P=%d (the reference indexes an empty sort and raises)", P);
  const long rank = select_rank(P, frac, min_n);
  ISR_REQUIRE(rank >= 0 && rank < P, "isr_select_top: rank %ld out of range for P=%d (IndexError in the reference)", rank, P);
  return select_top_impl(logp, P, P, 1, nullptr, frac, min_n, keep, M_dev, thr_dev, ws, ws_bytes, stream_);
}

extern "C" int isr_select_top_dev(const float* logp, int P_cap, const int32_t* n_dev, double frac, int min_n,
                                  int32_t* keep, int32_t* M_dev, float* thr_dev, void* ws, size_t ws_bytes,
                                  isr_stream_t stream_) {
  ISR_REQUIRE(logp && keep && M_dev && n_dev, "isr_select_top_dev: null pointer");
  ISR_REQUIRE(P_cap > 0, "isr_select_top_dev: P_cap=%d", P_cap);
  return select_top_impl(logp, P_cap, P_cap, 1, n_dev, frac, min_n, keep, M_dev, thr_dev, ws, ws_bytes, stream_);
}

extern "C" int isr_select_top_batch(const float* logp, int P, int B, const int32_t* n_dev, double frac, int min_n,
                                    int32_t* keep, int32_t* M_dev, float* thr_dev, void* ws, size_t ws_bytes,
                                    isr_stream_t stream_) {
  ISR_REQUIRE(logp && keep && M_dev, "isr_select_top_batch: null pointer");
  ISR_REQUIRE(P > 0 && B > 0 && B <= 65535, "isr_select_top_batch: P=%d B=%d", P, B);
  if (!n_dev) {
    const long rank = select_rank(P, frac, min_n);
    ISR_REQUIRE(rank >= 0 && rank < P, "isr_select_top_batch: rank %ld out of range for P=%d", rank, P);
  }
  return select_top_impl(logp, P, P, B, n_dev, frac, min_n, keep, M_dev, thr_dev, ws, ws_bytes, stream_);
}

extern "C" int isr_select_top_batch_digits(const float* logp, int P, int B, const int32_t* n_dev, double frac, int min_n,
                                           const int32_t* digit_hist, int32_t* keep, int32_t* M_dev, float* thr_dev, void* ws,
                                           size_t ws_bytes, isr_stream_t stream_) {
  ISR_REQUIRE(logp && keep && M_dev && digit_hist, "isr_select_top_batch_digits: null pointer");
  ISR_REQUIRE(P > 0 && B > 0 && B <= 65535, "isr_select_top_batch_digits: P=%d B=%d", P, B);
  if (!n_dev) {
    const long rank = select_rank(P, frac, min_n);
    ISR_REQUIRE(rank >= 0 && rank < P, "isr_select_top_batch_digits: rank %ld out of range for P=%d", rank, P);
  }
  return select_top_impl(logp, P, P, B, n_dev, frac, min_n, keep, M_dev, thr_dev, ws, ws_bytes, stream_, digit_hist);
}

static int gather_impl(const int32_t* idx, const int32_t* keep, const int32_t* M_dev, int P, int B, const float* pts,
                       const float* pix_xy, int64_t pix_stride, float* p3d, float* p2d, isr_stream_t stream) {
  gather_kernel<<<dim3((P + 255) / 256, 1, B), 256, 0, isr::as_stream(stream)>>>(idx, keep, M_dev, pts, pix_xy,
                                                                                 p3d, p2d, (int64_t)P, pix_stride);
  ISR_CHECK_LAUNCH("gather_kernel");
  return ISR_OK;
}

extern "C" int isr_gather_corr(const int32_t* idx, const int32_t* keep, const int32_t* M_dev, int P,
                               const float* pts, int N, const float* pix_xy, float* p3d, float* p2d,
                               isr_stream_t stream) {
  ISR_REQUIRE(idx && keep && M_dev && pts && pix_xy && p3d && p2d, "isr_gather_corr: null pointer");
  ISR_REQUIRE(P > 0 && N > 0, "isr_gather_corr: P=%d N=%d", P, N);
  return gather_impl(idx, keep, M_dev, P, 1, pts, pix_xy, 0, p3d, p2d, stream);
}

extern "C" int isr_gather_corr_batch(const int32_t* idx, const int32_t* keep, const int32_t* M_dev, int P, int B,
                                     const float* pts, int N, const float* pix_xy, int shared_pix, float* p3d,
                                     float* p2d, isr_stream_t stream) {
  ISR_REQUIRE(idx && keep && M_dev && pts && pix_xy && p3d && p2d, "isr_gather_corr_batch: null pointer");
  ISR_REQUIRE(P > 0 && N > 0 && B > 0 && B <= 65535, "isr_gather_corr_batch: P=%d N=%d B=%d", P, N, B);
  return gather_impl(idx, keep, M_dev, P, B, pts, pix_xy, shared_pix ? 0 : (int64_t)P * 2, p3d, p2d, stream);
}
