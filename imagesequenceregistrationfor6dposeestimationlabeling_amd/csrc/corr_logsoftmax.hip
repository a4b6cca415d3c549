// corr_logsoftmax.hip — K1, materialising variant for small P: out[p][n] = log_softmax_n <Q[p], K[n]>.
//
// Replaces  corr_matrix_log = torch.log_softmax(queries @ obj_keys.T, dim=1)      poseEstSurf.py:70
//           and getCors(leaves > 1), whose caller then runs topk on the matrix    inference.py:143-145
// The estimate_pose path keeps the (n x m) matrix resident (it max-pools and samples it), so this variant
// writes it.  f32 descriptors (what poseEstSurf.py feeds it), D <= 128 — round 3:
//   1. the row log-sum-exps come from K1's exact-f32 path (isr_corr_argmax, `lse` output: the same k-ordered
//      fmaf-chain logits, sums merged in f64) — 0.3 ms at n = 5 476, m = 80 000, e = 12;
//   2. corr_rows_kernel writes out[p][k] = <q_p, key_k> - lse[p]: thread = key (its D values in registers for the
//      whole workgroup's life), the workgroup's query rows in LDS (broadcast reads), stores coalesced along k.
//      HBM-bound on the 4 n m output bytes.  With POOL the same pass also writes the 3 x 3 spatially max-pooled
//      matrix of poseEstSurf.py:97-107: a workgroup owns one image row y of the res x res grid, computes the logits of
//      rows y - 1, y, y + 1 for its keys column by column and keeps a 3 x 3 window of column maxima in registers —
//      27 extra fma per element instead of a second pass that re-reads the matrix nine times (round 2:
//      logsoftmax_rows_kernel 9.1 ms + ep_pool_corr_kernel 3.7 ms of the 16.5 ms estimate_pose call).
// bf16 inputs (getCors with leaves > 1 on bf16 descriptors; no call site in the reference) keep the one-workgroup-
// per-row kernel below: three sweeps over the keys (max, sum, write).
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

constexpr int kRowBlock = 64;    // query rows per workgroup of the plain (un-pooled) matrix kernel

// POOL = false: rows [blockIdx.y * kRowBlock, ...) x keys [blockIdx.x * 256, ...): out_raw = logit - lse.
// POOL = true : image row y = blockIdx.y of a res x res pixel grid (P = res^2): out_raw as above and
//               out_pool[y, x][k] = max over the 3 x 3 pixel neighbourhood (inside the grid) of (logit - lse).
template <int DP, bool POOL>
__global__ __launch_bounds__(kThreads) void corr_rows_kernel(const float* __restrict__ Q, const float* __restrict__ K, int P,
                                                             int N, int D, int ldq, int ldk, const float* __restrict__ lse,
                                                             int res, float* __restrict__ out_raw,
                                                             float* __restrict__ out_pool, int64_t ldo) {
  extern __shared__ __attribute__((aligned(16))) float smem[];     // query rows (stride DP) then their lse
  const int tid = threadIdx.x;
  const int n = blockIdx.x * kThreads + tid;
  // the rows this workgroup needs: POOL: image rows y-1 .. y+1 (clipped), else one block of kRowBlock rows
  const int y = blockIdx.y;
  const int r_lo = POOL ? max(0, y - 1) * res : y * kRowBlock;
  const int r_hi = POOL ? min(res, y + 2) * res : min(P, (y + 1) * kRowBlock);
  const int nrows = r_hi - r_lo;
  float* qs = smem;
  float* ls = smem + (size_t)nrows * DP;
  for (int i = tid; i < nrows * DP; i += kThreads) {
    const int r = i / DP, d = i % DP;
    qs[i] = d < D ? Q[(size_t)(r_lo + r) * ldq + d] : 0.f;
  }
  for (int i = tid; i < nrows; i += kThreads) ls[i] = lse[r_lo + i];
  float k[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) k[d] = (n < N && d < D) ? K[(size_t)n * ldk + d] : 0.f;
  __syncthreads();
  // logit of local row r for this thread's key: the k-ordered fmaf chain from 0 (K1's exact-f32 logit, bit for bit)
  auto val = [&](int r) {
    const float4* q4 = reinterpret_cast<const float4*>(qs + (size_t)r * DP);
    float acc = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < DP / 4; ++d4) {
      const float4 q = q4[d4];
      acc = __builtin_fmaf(q.x, k[4 * d4], acc);
      acc = __builtin_fmaf(q.y, k[4 * d4 + 1], acc);
      acc = __builtin_fmaf(q.z, k[4 * d4 + 2], acc);
      acc = __builtin_fmaf(q.w, k[4 * d4 + 3], acc);
    }
    return acc - ls[r];
  };
  if (!POOL) {
    if (n >= N) return;
    for (int r = 0; r < nrows; ++r) out_raw[(size_t)(r_lo + r) * ldo + n] = val(r);
    return;
  }
  // local row index of pixel (yy, x); rows outside the grid contribute -inf (F.max_pool2d pads with -inf)
  const float ninf = -__builtin_inff();
  const int base_m = (y - 1 >= 0) ? 0 : -1;                         // is image row y-1 present?  its local offset
  const int off_c = (y - 1 >= 0) ? res : 0;                         // local offset of image row y
  const bool has_p = y + 1 < res;
  auto column = [&](int x, float& centre) {                         // max over the (up to) three rows of column x
    float c0 = ninf, c2 = ninf;
    if (base_m == 0) c0 = val(x);
    centre = val(off_c + x);
    if (has_p) c2 = val(off_c + res + x);
    return fmaxf(fmaxf(c0, centre), c2);
  };
  float cen = 0.f, cen_next = 0.f;
  float left = ninf, mid = column(0, cen), right = ninf;
  for (int x = 0; x < res; ++x) {
    right = (x + 1 < res) ? column(x + 1, cen_next) : ninf;
    if (n < N) {
      const size_t o = (size_t)(y * res + x) * ldo + n;
      out_raw[o] = cen;
      out_pool[o] = fmaxf(fmaxf(left, mid), right);
    }
    left = mid; mid = right; cen = cen_next;
  }
}

// avg_queries = False (poseEstSurf.py:72-96): per-PIXEL log-softmax, pooled per scale x scale block.  A workgroup owns
// one row y of output cells: the `scale` pixel rows it needs sit in LDS with their log-sum-exps (from K1, one row per
// pixel of the r x r crop), thread = key; per cell the scale^2 logits are formed in registers, the centre pixel's value
// goes to corr_centre (the sampling matrix), the block maximum to corr_blockmax (the scoring matrix before the 3 x 3
// pool) — the 15.8 GB full-resolution matrix is never formed, and the keys are read once per cell row instead of
// three times per cell (round 2's one-workgroup-per-cell kernel: 45 ms of a 52 ms call).
template <int DP>
__global__ __launch_bounds__(kThreads) void corr_patch_kernel(const float* __restrict__ Qimg, const float* __restrict__ K, int r,
                                                              int res, int scale, int N, int D, const float* __restrict__ lse,
                                                              float* __restrict__ out_centre, float* __restrict__ out_bmax) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int n = blockIdx.x * kThreads + tid;
  const int y = blockIdx.y;
  const int wpix = res * scale;                     // pixel columns used (r // scale * scale)
  const int nrows = scale * wpix;
  float* qs = smem;
  float* ls = smem + (size_t)nrows * DP;
  for (int i = tid; i < nrows * DP; i += kThreads) {
    const int rr = i / DP, d = i % DP;
    const int py = y * scale + rr / wpix, px = rr % wpix;
    qs[i] = d < D ? Qimg[((size_t)py * r + px) * D + d] : 0.f;
  }
  for (int i = tid; i < nrows; i += kThreads) ls[i] = lse[(size_t)(y * scale + i / wpix) * r + i % wpix];
  float k[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) k[d] = (n < N && d < D) ? K[(size_t)n * D + d] : 0.f;
  __syncthreads();
  if (n >= N) return;
  const int cy = scale / 2, cx = scale / 2;
  for (int x = 0; x < res; ++x) {
    float best = -__builtin_inff(), cen = 0.f;
    for (int py = 0; py < scale; ++py)
      for (int px = 0; px < scale; ++px) {
        const int rr = py * wpix + x * scale + px;
        const float4* q4 = reinterpret_cast<const float4*>(qs + (size_t)rr * DP);
        float acc = 0.f;
#pragma unroll
        for (int d4 = 0; d4 < DP / 4; ++d4) {
          const float4 q = q4[d4];
          acc = __builtin_fmaf(q.x, k[4 * d4], acc);
          acc = __builtin_fmaf(q.y, k[4 * d4 + 1], acc);
          acc = __builtin_fmaf(q.z, k[4 * d4 + 2], acc);
          acc = __builtin_fmaf(q.w, k[4 * d4 + 3], acc);
        }
        const float v = acc - ls[rr];
        best = fmaxf(best, v);
        if (py == cy && px == cx) cen = v;
      }
    const size_t o = (size_t)(y * res + x) * N + n;
    out_centre[o] = cen;
    out_bmax[o] = best;
  }
}

template <bool POOL>
int launch_rows(const float* Q, const float* K, int P, int N, int D, int ldq, int ldk, const float* lse, int res, float* raw,
                float* pool, int64_t ldo, hipStream_t stream) {
  const int rows_per_block = POOL ? 3 * res : kRowBlock;
  const dim3 grid((N + kThreads - 1) / kThreads, POOL ? res : (P + kRowBlock - 1) / kRowBlock);
#define ISR_ROWS(DPv)                                                                                                   \
  do {                                                                                                                  \
    const size_t sh = ((size_t)rows_per_block * DPv + rows_per_block) * sizeof(float);                                  \
    if (sh > 64 * 1024) { isr::set_error("isr_corr_logsoftmax: %zu B of LDS (res=%d, D=%d)", sh, res, D); return ISR_ERR_UNSUPPORTED; } \
    corr_rows_kernel<DPv, POOL><<<grid, kThreads, sh, stream>>>(Q, K, P, N, D, ldq, ldk, lse, res, raw, pool, ldo);      \
  } while (0)
  if (D <= 16) ISR_ROWS(16);
  else if (D <= 32) ISR_ROWS(32);
  else if (D <= 64) ISR_ROWS(64);
  else ISR_ROWS(128);
#undef ISR_ROWS
  return ISR_OK;
}

// the row log-sum-exps through K1's exact-f32 path; carves idx / lse / K1 scratch from ws
int row_lse(const float* Q, const float* K, int P, int N, int D, int ldq, int ldk, void* ws, size_t ws_bytes, float** lse_out,
            isr_stream_t stream) {
  isr::Workspace w(ws, ws_bytes);
  int32_t* idx = w.take<int32_t>(P);
  float* lse = w.take<float>(P);
  void* k1 = w.take<char>(0);
  const size_t used = (size_t)(static_cast<char*>(k1) - static_cast<char*>(ws));
  *lse_out = lse;
  (void)idx;   // an lse-only call of K1 (idx == nullptr): no maxima tracked, nothing rechecked
  return isr_corr_argmax(Q, K, P, N, D, ldq, ldk, ISR_DTYPE_F32, nullptr, nullptr, lse, k1, ws_bytes - used, stream);
}

}  // namespace

extern "C" size_t isr_corr_logsoftmax_workspace_bytes(int P, int N, int D, int dtype) {
  if (P <= 0 || N <= 0 || D <= 0) return 0;
  if (dtype != ISR_DTYPE_F32 || D > 128) return 256;       // the three-sweep kernel needs no scratch
  return isr::align_up((size_t)P * 4, 256) * 2 + isr_corr_argmax_workspace_bytes(P, N, D, ISR_DTYPE_F32) + 512;
}

extern "C" int isr_ep_corr_matrices(const float* queries, const float* keys, int res, int m, int e, float* corr_raw,
                                    float* corr_pool, void* ws, size_t ws_bytes, isr_stream_t stream_) {
  ISR_REQUIRE(queries && keys && corr_raw, "isr_ep_corr_matrices: null pointer");
  ISR_REQUIRE(res > 0 && m > 0 && e > 0 && e <= 128, "isr_ep_corr_matrices: res=%d m=%d e=%d (e <= 128)", res, m, e);
  const int n = res * res;
  if (!ws || ws_bytes < isr_corr_logsoftmax_workspace_bytes(n, m, e, ISR_DTYPE_F32)) {
    isr::set_error("isr_ep_corr_matrices: workspace %zu < %zu", ws_bytes, isr_corr_logsoftmax_workspace_bytes(n, m, e, ISR_DTYPE_F32));
    return ISR_ERR_WORKSPACE;
  }
  float* lse = nullptr;
  int rc = row_lse(queries, keys, n, m, e, e, e, ws, ws_bytes, &lse, stream_);
  if (rc != ISR_OK) return rc;
  hipStream_t stream = isr::as_stream(stream_);
  const int DP = e <= 16 ? 16 : e <= 32 ? 32 : e <= 64 ? 64 : 128;
  const bool fused = corr_pool && (size_t)3 * res * (DP + 1) * sizeof(float) <= 64 * 1024;   // three image rows of queries in LDS
  rc = fused ? launch_rows<true>(queries, keys, n, m, e, e, e, lse, res, corr_raw, corr_pool, m, stream)
             : launch_rows<false>(queries, keys, n, m, e, e, e, lse, res, corr_raw, nullptr, m, stream);
  if (rc != ISR_OK) return rc;
  ISR_CHECK_LAUNCH("corr_rows_kernel");
  if (corr_pool && !fused) return isr_ep_pool_corr(corr_raw, res, m, corr_pool, stream_);     // wide descriptors / large crops
  return ISR_OK;
}

extern "C" int isr_ep_patch_corr_cells(const float* query_img, const float* obj_keys, int r, int e, int scale, int m,
                                       float* corr_centre, float* corr_blockmax, isr_stream_t stream);

extern "C" size_t isr_ep_patch_corr_workspace_bytes(int r, int m, int e) {
  if (r <= 0 || m <= 0 || e <= 0) return 0;
  return isr_corr_logsoftmax_workspace_bytes(r * r, m, e, ISR_DTYPE_F32);
}

extern "C" int isr_ep_patch_corr(const float* query_img, const float* obj_keys, int r, int e, int scale, int m,
                                 float* corr_centre, float* corr_blockmax, void* ws, size_t ws_bytes, isr_stream_t stream_) {
  ISR_REQUIRE(query_img && obj_keys && corr_centre && corr_blockmax, "isr_ep_patch_corr: null pointer");
  ISR_REQUIRE(r > 0 && m > 0 && e > 0 && scale >= 1 && r / scale > 0, "isr_ep_patch_corr: r=%d e=%d scale=%d m=%d", r, e, scale, m);
  const int res = r / scale;
  const int DP = e <= 16 ? 16 : e <= 32 ? 32 : e <= 64 ? 64 : 128;
  const size_t sh = (size_t)scale * res * scale * (DP + 1) * sizeof(float);
  if (e > 128 || sh > 64 * 1024 || !ws || ws_bytes < isr_ep_patch_corr_workspace_bytes(r, m, e))
    return isr_ep_patch_corr_cells(query_img, obj_keys, r, e, scale, m, corr_centre, corr_blockmax, stream_);   // one workgroup per cell
  float* lse = nullptr;
  const int rc = row_lse(query_img, obj_keys, r * r, m, e, e, e, ws, ws_bytes, &lse, stream_);     // one row per pixel of the crop
  if (rc != ISR_OK) return rc;
  hipStream_t stream = isr::as_stream(stream_);
  const dim3 grid((m + kThreads - 1) / kThreads, res);
  if (DP == 16) corr_patch_kernel<16><<<grid, kThreads, sh, stream>>>(query_img, obj_keys, r, res, scale, m, e, lse, corr_centre, corr_blockmax);
  else if (DP == 32) corr_patch_kernel<32><<<grid, kThreads, sh, stream>>>(query_img, obj_keys, r, res, scale, m, e, lse, corr_centre, corr_blockmax);
  else if (DP == 64) corr_patch_kernel<64><<<grid, kThreads, sh, stream>>>(query_img, obj_keys, r, res, scale, m, e, lse, corr_centre, corr_blockmax);
  else corr_patch_kernel<128><<<grid, kThreads, sh, stream>>>(query_img, obj_keys, r, res, scale, m, e, lse, corr_centre, corr_blockmax);
  ISR_CHECK_LAUNCH("corr_patch_kernel");
  return ISR_OK;
}

extern "C" int isr_corr_logsoftmax(const void* Q, const void* K, int P, int N, int D, int ldq, int ldk,
                                   int dtype, float* out, int64_t ldo, void* ws, size_t ws_bytes, isr_stream_t stream_) {
  ISR_REQUIRE(Q && K && out, "isr_corr_logsoftmax: null pointer");
  ISR_REQUIRE(P > 0 && N > 0 && D > 0 && D <= kMaxD, "isr_corr_logsoftmax: P=%d N=%d D=%d (D <= %d)", P, N, D, kMaxD);
  ISR_REQUIRE(ldq >= D && ldk >= D && ldo >= N, "isr_corr_logsoftmax: leading dimensions too small");
  hipStream_t stream = isr::as_stream(stream_);
  if (dtype == ISR_DTYPE_F32 && D <= 128) {
    if (!ws || ws_bytes < isr_corr_logsoftmax_workspace_bytes(P, N, D, dtype)) {
      isr::set_error("isr_corr_logsoftmax: workspace %zu < %zu", ws_bytes, isr_corr_logsoftmax_workspace_bytes(P, N, D, dtype));
      return ISR_ERR_WORKSPACE;
    }
    float* lse = nullptr;
    int rc = row_lse(static_cast<const float*>(Q), static_cast<const float*>(K), P, N, D, ldq, ldk, ws, ws_bytes, &lse, stream_);
    if (rc != ISR_OK) return rc;
    rc = launch_rows<false>(static_cast<const float*>(Q), static_cast<const float*>(K), P, N, D, ldq, ldk, lse, 0, out, nullptr,
                            ldo, stream);
    if (rc != ISR_OK) return rc;
    ISR_CHECK_LAUNCH("corr_rows_kernel");
    return ISR_OK;
  }
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
