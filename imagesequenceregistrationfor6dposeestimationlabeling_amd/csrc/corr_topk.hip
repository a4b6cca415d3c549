// corr_topk.hip — getCors(queries, feats, leaves > 1) without the (P x N) matrix.
//
// Replaces  cMat = torch.log_softmax(queries @ feats.T, -1); vals, idx = torch.topk(cMat, leaves, -1)
//           inference.py:145-149 (= finalposes.py:41-45 = choosePose.py:38-42), leaves > 1 (no call site of the reference
//           passes one; the signature allows it).  Round 3 materialised the matrix (isr_corr_logsoftmax) and ran torch.topk
//           on it: 1.8 GB at the reference's 5 625 x 80 000, 24.6 GB at 640 x 480 x 20 000.
// Here: the row log-sum-exps come from an lse-only isr_corr_argmax call (the caller passes them), and this file keeps, per
// query, the k <= 8 largest logits and their keys in registers while the keys stream by:
//   * logit = the k-ordered f32 fmaf chain from 0 — the value isr_corr_logsoftmax writes and the f32 oracle computes;
//   * a thread owns one query (descriptor in registers), a workgroup of 256 queries shares 32-key tiles of the key matrix
//     in LDS (every lane reads the same address: broadcast ds_read_b128), the key range is cut into `nr` ranges on
//     blockIdx.y so that small P still fills the chip;
//   * the list is sorted by (value descending, key ascending): a candidate enters only on a strict `>` against an entry, and
//     keys arrive in ascending order, so equal values keep the lower key first — torch documents no tie order, the build
//     defines this one (as for leaves = 1);
//   * corr_topk_merge_kernel merges the ranges' lists in ascending range order with the same rule and writes
//     vals = logit - lse, idx.
// f32 VALU work, D FMA per (query, key): not a matrix-core kernel — leaves > 1 is a rarely used corner of the call surface,
// bounded here by the f32 vector rate instead of by 4 P N bytes of HBM writes plus a sort.
#include "isr_common.hpp"

namespace {

constexpr int kTopThreads = 256;
constexpr int kTopK = 8;          // list length kept (leaves <= kTopK)
constexpr int kTopTile = 32;      // keys per LDS tile
constexpr int kTopMaxRanges = 64;

struct TopList {
  float v[kTopK];
  int i[kTopK];
};

__device__ __forceinline__ void top_init(TopList& t) {
#pragma unroll
  for (int j = 0; j < kTopK; ++j) { t.v[j] = -__builtin_inff(); t.i[j] = 0x7fffffff; }
}

// (v, n) enters behind every entry that is larger, or equal with a lower key
__device__ __forceinline__ bool top_before(float va, int ia, float vb, int ib) { return va > vb || (va == vb && ia < ib); }

__device__ __forceinline__ void top_insert(TopList& t, float v, int n) {
  if (!top_before(v, n, t.v[kTopK - 1], t.i[kTopK - 1])) return;
#pragma unroll
  for (int j = kTopK - 1; j >= 1; --j) {
    const bool up = top_before(v, n, t.v[j - 1], t.i[j - 1]);       // the candidate also beats entry j - 1: that one moves down
    const bool here = !up && top_before(v, n, t.v[j], t.i[j]);      // ... it does not: the candidate lands at j
    const float pv = t.v[j - 1];
    const int pi = t.i[j - 1];
    t.v[j] = up ? pv : (here ? v : t.v[j]);
    t.i[j] = up ? pi : (here ? n : t.i[j]);
  }
  if (top_before(v, n, t.v[0], t.i[0])) { t.v[0] = v; t.i[0] = n; }
}

template <int DP>
__global__ __launch_bounds__(kTopThreads) void corr_topk_kernel(const float* __restrict__ Q, const float* __restrict__ K, int P, int N,
                                                                 int D, int ldq, int ldk, int per, float* __restrict__ pv,
                                                                 int32_t* __restrict__ pi) {
  __shared__ __attribute__((aligned(16))) float tile[kTopTile][DP];
  const int q = blockIdx.x * kTopThreads + threadIdx.x;
  const int range = blockIdx.y;
  const int k0 = range * per, k1 = min(N, k0 + per);
  float qr[DP];
  {
    const int row = q < P ? q : P - 1;
#pragma unroll
    for (int d = 0; d < DP; ++d) qr[d] = d < D ? Q[(size_t)row * ldq + d] : 0.f;
  }
  TopList top;
  top_init(top);
  for (int kb = k0; kb < k1; kb += kTopTile) {          // block-uniform
    __syncthreads();
    for (int e = threadIdx.x; e < kTopTile * DP; e += kTopThreads) {
      const int kr = e / DP, d = e % DP;
      tile[kr][d] = (kb + kr < k1 && d < D) ? K[(size_t)(kb + kr) * ldk + d] : 0.f;
    }
    __syncthreads();
    const int nk = min(kTopTile, k1 - kb);
    for (int kr = 0; kr < nk; ++kr) {
      float acc = 0.f;
#pragma unroll
      for (int d = 0; d < DP; d += 4) {
        const float4 kv = *reinterpret_cast<const float4*>(&tile[kr][d]);
        acc = __builtin_fmaf(qr[d], kv.x, acc);
        acc = __builtin_fmaf(qr[d + 1], kv.y, acc);
        acc = __builtin_fmaf(qr[d + 2], kv.z, acc);
        acc = __builtin_fmaf(qr[d + 3], kv.w, acc);
      }
      top_insert(top, acc, kb + kr);
    }
  }
  if (q < P) {
    const size_t o = ((size_t)range * P + q) * kTopK;
#pragma unroll
    for (int j = 0; j < kTopK; ++j) { pv[o + j] = top.v[j]; pi[o + j] = top.i[j]; }
  }
}

// per query: the k best of the nr ranges' lists (ranges ascending in key index: on equal values the earlier range — the
// lower key — wins), vals = logit - lse
__global__ __launch_bounds__(256) void corr_topk_merge_kernel(const float* __restrict__ pv, const int32_t* __restrict__ pi, int P, int nr,
                                                               int k, const float* __restrict__ lse, int32_t* __restrict__ idx,
                                                               float* __restrict__ vals) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= P) return;
  TopList top;
  top_init(top);
  for (int r = 0; r < nr; ++r) {
    const size_t o = ((size_t)r * P + q) * kTopK;
#pragma unroll
    for (int j = 0; j < kTopK; ++j) {
      const int n = pi[o + j];
      if (n != 0x7fffffff) top_insert(top, pv[o + j], n);
    }
  }
  const float l = lse[q];
#pragma unroll
  for (int j = 0; j < kTopK; ++j)
    if (j < k) {
      idx[(size_t)q * k + j] = top.i[j] == 0x7fffffff ? -1 : top.i[j];     // -1: fewer than k keys exist
      vals[(size_t)q * k + j] = top.v[j] - l;
    }
}

int ranges_for(int P, int N) {
  const int qblocks = (P + kTopThreads - 1) / kTopThreads;
  int nr = (1024 + qblocks - 1) / qblocks;                         // ~4 workgroups per CU
  const int tiles = (N + kTopTile - 1) / kTopTile;
  if (nr > tiles) nr = tiles;
  if (nr > kTopMaxRanges) nr = kTopMaxRanges;
  return nr < 1 ? 1 : nr;
}

// the key ranges a call uses: `per` keys each (whole tiles), `nru` of them hold a key — the one place the workspace size and
// the launch take it from (round 4 reserved kTopMaxRanges lists per query whatever the shape: 1.26 GB at 640 x 480 queries,
// of which one range's 20 MB was used)
struct TopRanges { int per, nru; };
TopRanges top_ranges(int P, int N) {
  const int nr = ranges_for(P, N);
  const int tiles = (N + kTopTile - 1) / kTopTile;
  TopRanges t;
  t.per = (tiles + nr - 1) / nr * kTopTile;
  t.nru = (N + t.per - 1) / t.per;
  return t;
}

}  // namespace

extern "C" size_t isr_corr_topk_workspace_bytes(int P, int N) {
  if (P <= 0 || N <= 0) return 0;
  return (size_t)top_ranges(P, N).nru * P * kTopK * (sizeof(float) + sizeof(int32_t)) + 1024;
}

extern "C" int isr_corr_topk(const float* Q, const float* K, int P, int N, int D, int ldq, int ldk, int k, const float* lse,
                             int32_t* idx, float* vals, void* ws_, size_t ws_bytes, isr_stream_t stream_) {
  ISR_REQUIRE(Q && K && lse && idx && vals, "isr_corr_topk: null pointer");
  ISR_REQUIRE(P > 0 && N > 0 && D > 0 && D <= 128, "isr_corr_topk: P=%d N=%d D=%d (0 < D <= 128)", P, N, D);
  ISR_REQUIRE(ldq >= D && ldk >= D, "isr_corr_topk: ldq=%d ldk=%d < D=%d", ldq, ldk, D);
  ISR_REQUIRE(k >= 1 && k <= kTopK, "isr_corr_topk: leaves=%d (1 .. %d)", k, kTopK);
  if (!ws_ || ws_bytes < isr_corr_topk_workspace_bytes(P, N)) {
    isr::set_error("isr_corr_topk: workspace %zu < %zu", ws_bytes, isr_corr_topk_workspace_bytes(P, N));
    return ISR_ERR_WORKSPACE;
  }
  hipStream_t stream = isr::as_stream(stream_);
  isr::Workspace w(ws_, ws_bytes);
  const TopRanges tr = top_ranges(P, N);
  const int per = tr.per, nru = tr.nru;                              // keys per range (whole tiles), ranges that hold a key
  float* pv = w.take<float>((size_t)nru * P * kTopK);
  int32_t* pi = w.take<int32_t>((size_t)nru * P * kTopK);
  const dim3 grid((P + kTopThreads - 1) / kTopThreads, nru);
  if (D <= 16) corr_topk_kernel<16><<<grid, kTopThreads, 0, stream>>>(Q, K, P, N, D, ldq, ldk, per, pv, pi);
  else if (D <= 32) corr_topk_kernel<32><<<grid, kTopThreads, 0, stream>>>(Q, K, P, N, D, ldq, ldk, per, pv, pi);
  else if (D <= 64) corr_topk_kernel<64><<<grid, kTopThreads, 0, stream>>>(Q, K, P, N, D, ldq, ldk, per, pv, pi);
  else corr_topk_kernel<128><<<grid, kTopThreads, 0, stream>>>(Q, K, P, N, D, ldq, ldk, per, pv, pi);
  corr_topk_merge_kernel<<<(P + 255) / 256, 256, 0, stream>>>(pv, pi, P, nru, k, lse, idx, vals);
  ISR_CHECK_LAUNCH("corr topk kernels");
  return ISR_OK;
}
