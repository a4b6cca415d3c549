// nn_batched.hip — K3/K4: batched brute-force nearest neighbour with fused Chamfer / ADD-S /
// ICP reductions, hand-written for gfx950 (64-wide waves, LDS-broadcast target tiles).
//
// Replaces (reference file:line):
//   sklearn KDTree(leaf_size=2).query(k=1)        inference.py:118-120, choosePose.py:20-22 (ADD-S)
//   open3d compute_point_cloud_distance           verfication.py:97-101, icp.py:113-117 (Chamfer)
//   open3d evaluate_registration/registration_icp icp.py:97-103 (NN-with-radius + Kabsch sums)
//
// Work decomposition
//   grid = (query blocks, target splits, batch items); 256 threads; every lane owns RQ queries
//   in registers and the block streams its target range through LDS in 256-point SoA tiles.
//   All lanes read the same target (LDS broadcast), so one ds_read_b128 feeds 4 targets x RQ
//   queries x 64 lanes of VALU work.  Targets are taken 8 at a time: 6 VALU ops per pair for
//   the squared distance, a min3 tree per group, and the (value, index) update only runs in the
//   wave-uniform slow path when some lane improved — after warm-up that is rare, which keeps
//   the loop at ~6.6 VALU ops per pair instead of 9.
//   When a batch has too few queries to fill 256 CUs the target range is split over
//   blockIdx.y and the finalize kernel merges the partial winners (lowest index on ties).
//
// Numerics: transforms in f64 (fma chain, translation innermost) rounded to f32; search in f32
// with d2 = fmaf(dz,dz,fmaf(dy,dy,dx*dx)) and strict '<'; the winner's distance re-evaluated in
// f64.  Sums are fixed-shape trees, no float atomics.  oracle/isr_oracle.c:orc_nn_batched is the
// CPU statement of the same arithmetic.
#include "isr_common.hpp"

#include <cstdlib>

namespace {

#ifndef ISR_NN_PACKED_BATCH
#define ISR_NN_PACKED_BATCH 1   // 0: per-split partial arrays in the batched brute-force search (rounds 1-3)
#endif
constexpr int kThreads = 256;
constexpr int kTile = 256;  // targets per LDS tile
constexpr int kGroup = 8;   // targets per min3 group
constexpr int kNV = 18;     // sum_d, sum_d2, count, 15 covariance sums

__device__ __forceinline__ void xform64(const double* __restrict__ T, float x, float y, float z,
                                        double& ox, double& oy, double& oz) {
  if (T == nullptr) {
    ox = x; oy = y; oz = z;
    return;
  }
  const double dx = x, dy = y, dz = z;
  ox = fma(T[2], dz, fma(T[1], dy, fma(T[0], dx, T[3])));
  oy = fma(T[6], dz, fma(T[5], dy, fma(T[4], dx, T[7])));
  oz = fma(T[10], dz, fma(T[9], dy, fma(T[8], dx, T[11])));
}

constexpr int kUnresolved = -2;   // part_idx of a query the grid search handed to the brute-force pass

// Query r of a lane: a WAVE owns RQ x 64 consecutive queries (wave w of workgroup x: x RQ 256 + w RQ 64 .. + RQ 64), a lane's RQ
// queries sit 64 apart.  Consecutive rows of a spatially ordered cloud (registration.icp_point_to_point sorts along a Morton
// curve) are a compact patch, which is what the per-wave tile cull below lives on; loads stay coalesced (lane = fastest index).
template <int RQ>
__device__ __forceinline__ int query_of(int r) {
  return (int)blockIdx.x * RQ * kThreads + ((int)threadIdx.x >> 6) * (RQ * 64) + r * 64 + ((int)threadIdx.x & 63);
}

// Axis-aligned boxes of the target cloud's 256-point tiles (untransformed frame), one workgroup per tile: {lo xyz, hi xyz}.
__global__ __launch_bounds__(256) void tile_box_kernel(const float* __restrict__ tgt, int Nt, float* __restrict__ box) {
  __shared__ float red[6][4];
  const int j = blockIdx.x * 256 + threadIdx.x;
  float v[6] = {3.0e38f, 3.0e38f, 3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
  if (j < Nt) {
#pragma unroll
    for (int a = 0; a < 3; ++a) v[a] = v[3 + a] = tgt[3 * (size_t)j + a];
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      v[a] = fminf(v[a], __shfl_xor(v[a], o, 64));
      v[3 + a] = fmaxf(v[3 + a], __shfl_xor(v[3 + a], o, 64));
    }
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 6; ++a) red[a][threadIdx.x >> 6] = v[a];
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int a = threadIdx.x;
    const float r0 = red[a][0], r1 = red[a][1], r2 = red[a][2], r3 = red[a][3];
    box[6 * (size_t)blockIdx.x + a] = a < 3 ? fminf(fminf(r0, r1), fminf(r2, r3)) : fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
  }
}

// FILTER = false: the plain loop — per pair the difference form d2 = fmaf(dz,dz,fmaf(dy,dy,dx*dx)) (6 VALU ops), a min3
// tree per group of 8 targets, the (value, index) update in a wave-uniform slow path.
// FILTER = true: the search runs on the 3-FMA form  s = |t|^2 - 2 q.t  (= d2 - |q|^2 up to rounding: half the VALU work of the
// difference form) as a FILTER; the difference form d2 = fmaf(dz,dz,fmaf(dy,dy,dx*dx)) — the arithmetic that
// defines the result — is evaluated only for targets whose s could beat or tie the lane's best:
//   s <= best - |q|^2 + E,   E >= |s + |q|^2 - d2|.
// With u = 2^-24: the three fmas of s and the three of |t|^2 err by <= 6u (|q| + |t|)^2, |q|^2 by 3u |q|^2, d2 by
// 5u d2 <= 5u (|q| + |t|)^2 — together <= 14u (|q| + |t|)^2 <= 28u (|q|^2 + |t|^2); E = 64u (|q|^2 + max |t|^2 of
// the tile) + 8u best also covers the roundings of the threshold itself.  A group of 8 targets costs 24 fma + a
// min3 tree; when some lane's smallest s passes its threshold (wave-uniform), each of the 8 entries that passes on
// some lane (wave-uniform again: typically one or two) is evaluated exactly.  The winner is the lexicographic
// minimum of (d2, index), i.e. strict '<' in ascending index order: both loops return the same winners.  The
// filter pays when a lane's running minimum has settled — target ranges of several tiles (batches: 7.2 vs 6.0
// Tpairs/s on 32 x 20 000^2), or a warm start; on a one-tile range (one batch item split 79 ways) the minima are
// young, most groups pass the filter, and the plain loop is 25 % faster (tools/time_nn.py).
// warm != nullptr (ICP passes after the first): the lane starts from last pass's neighbour — its distance under
// the new T is an upper bound that is almost always the answer, so almost nothing passes the filter.
// unresolved != nullptr: second pass behind nn_grid_search_kernel (nsplit must be 1) — only blocks
// that hold a query marked kUnresolved run, and only those queries are stored.
// EXACT = true (the ICP loop, packed != nullptr): the lane also watches for NEAR TIES.  The search runs on f32-rounded
// coordinates; with u = 2^-24 and delta = u (|q| + max |t|) (the rounding displacement of the two points) the f32 value
// D32 of a pair differs from its exact squared distance by at most tol(D) = 6u D + 2 sqrt(D) delta + delta^2.  A target
// other than the lane's winner whose D32 lies within  best + 2.5 tol(best)  may be the exact (f64) nearest neighbour —
// Open3D's KD-tree (icp.py:96-103) searches doubles — so the lane raises amb[query]; icp_finalize_update_kernel then
// decides that query over all targets in f64.  Any such target is seen: the slow path is entered on  m <= thx  (thx =
// that widened bound, which only shrinks as best does) instead of  m < best, the filter's gate is built from thx, a
// winner that displaces a near-tied predecessor raises the flag too, and so does the packed atomicMin when the value
// it displaces (or fails to displace) — another target split's winner — is a near tie.  Cost in the fast path: none
// (the comparisons are against thx instead of best).  On 20 000-point clouds one or two queries per pass are flagged.
template <int RQ, bool FILTER, bool EXACT = false, bool CULL = false>
__global__ __launch_bounds__(kThreads) void nn_search_kernel(
    const float* __restrict__ qry, int Nq, const float* __restrict__ tgt, int Nt,
    const double* __restrict__ Tq, const double* __restrict__ Tt, int split_len, int nsplit,
    float* __restrict__ part_d2, int32_t* __restrict__ part_idx, const int32_t* __restrict__ skip,
    const int32_t* __restrict__ unresolved, unsigned long long* __restrict__ packed = nullptr,
    const int32_t* __restrict__ warm = nullptr, int32_t* __restrict__ amb = nullptr,
    const float* __restrict__ t2_bound = nullptr, const float* __restrict__ tile_box = nullptr, float cull_r2 = 0.f) {
  __shared__ __attribute__((aligned(16))) float lds[2][FILTER ? 4 : 3][kTile];   // FILTER: -2x, -2y, -2z, |t|^2; else x, y, z
  __shared__ float tile_t2[2][kThreads / 64];                       // largest |t|^2 of a tile, per staging wave
  if (skip && *skip) return;  // device-side ICP loop: converged, later iterations are no-ops
  if (unresolved && *unresolved == 0) return;

  const int tid = threadIdx.x;
  const int b = blockIdx.z;
  const int split = blockIdx.y;
  if (unresolved) {
    int mine = 0;
#pragma unroll
    for (int r = 0; r < RQ; ++r) {
      const int qi = query_of<RQ>(r);
      if (qi < Nq) mine |= part_idx[((size_t)b * nsplit + split) * Nq + qi] == kUnresolved;
    }
    if (!__syncthreads_or(mine)) return;
  }
  const double* tq = Tq ? Tq + 12 * (size_t)b : nullptr;
  const double* tt = Tt ? Tt + 12 * (size_t)b : nullptr;

  float qx[RQ], qy[RQ], qz[RQ], q2[RQ], best[RQ], thr[RQ];
  int bidx[RQ];
  // EXACT only: rounding displacement delta, widened bound thx, near-tie flag
  float dlt[RQ], thx[RQ];
  bool am[RQ];
  int widx[RQ];                           // the warm-start neighbour (-1: none)
  constexpr float kU0 = 5.9604645e-8f;    // 2^-24
  auto nearthr = [&](int r, float D) {    // D + 2.5 tol(D), inf for D = inf
    const float tol = __builtin_fmaf(6.f * kU0, D, __builtin_fmaf(2.f * __builtin_sqrtf(D), dlt[r], dlt[r] * dlt[r]));
    return __builtin_fmaf(2.5f, tol, D);
  };
#pragma unroll
  for (int r = 0; r < RQ; ++r) {
    int qi = query_of<RQ>(r);
    qi = qi < Nq ? qi : Nq - 1;  // clamp: out-of-range lanes compute a valid query, never store
    double x, y, z;
    xform64(tq, qry[3 * (size_t)qi], qry[3 * (size_t)qi + 1], qry[3 * (size_t)qi + 2], x, y, z);
    qx[r] = (float)x; qy[r] = (float)y; qz[r] = (float)z;
    q2[r] = __builtin_fmaf(qz[r], qz[r], __builtin_fmaf(qy[r], qy[r], qx[r] * qx[r]));
    best[r] = __builtin_inff();
    bidx[r] = -1;
    const int w = warm ? warm[(size_t)b * Nq + qi] : -1;
    widx[r] = (w >= 0 && w < Nt) ? w : -1;
    if (w >= 0 && w < Nt) {
      double tx, ty, tz;
      xform64(tt, tgt[3 * (size_t)w], tgt[3 * (size_t)w + 1], tgt[3 * (size_t)w + 2], tx, ty, tz);
      const float dx = qx[r] - (float)tx, dy = qy[r] - (float)ty, dz = qz[r] - (float)tz;
      best[r] = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
      bidx[r] = w;
    }
    if (EXACT) {
      dlt[r] = 1.001f * kU0 * (__builtin_sqrtf(q2[r]) + __builtin_sqrtf(*t2_bound));
      thx[r] = nearthr(r, best[r]);
      am[r] = false;
    }
  }

  const int t0 = split * split_len;
  const int t1 = min(Nt, t0 + split_len);
  const int ntiles = (t1 - t0 + kTile - 1) / kTile;
  constexpr float kU = 5.9604645e-8f;    // 2^-24

  // Per-wave tile cull (tile_box != nullptr: the ICP loop; targets untransformed).  A lane needs a target only if its f32
  // squared distance can reach the lane's bound — the warm neighbour's distance under the new transform (widened for near
  // ties when EXACT), capped by the radius beyond which nothing is a correspondence (cull_r2 > 0): a target farther than
  // that changes no output of the loop (a point whose neighbour lies beyond the radius contributes nothing, found or not).
  // The wave's bound is the largest of its lanes'; a tile whose box lies farther from the wave's query box than that, with
  // 1e-4 relative + 1e-4 absolute to spare (f32 roundings of the boxes, of the distance, of the bound: < 1e-5), holds
  // nothing any lane needs, and the wave skips its compute; a workgroup whose one tile every wave skips leaves at once.
  float wlo[3], whi[3], wb = __builtin_inff();
  if (CULL && tile_box) {
    float b = 0.f;
    wlo[0] = wlo[1] = wlo[2] = 3.0e38f; whi[0] = whi[1] = whi[2] = -3.0e38f;
#pragma unroll
    for (int r = 0; r < RQ; ++r) {
      float br = EXACT ? thx[r] : best[r];
      if (cull_r2 > 0.f) br = fminf(br, cull_r2);
      b = fmaxf(b, br);
      wlo[0] = fminf(wlo[0], qx[r]); wlo[1] = fminf(wlo[1], qy[r]); wlo[2] = fminf(wlo[2], qz[r]);
      whi[0] = fmaxf(whi[0], qx[r]); whi[1] = fmaxf(whi[1], qy[r]); whi[2] = fmaxf(whi[2], qz[r]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      b = fmaxf(b, __shfl_xor(b, o, 64));
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        wlo[a] = fminf(wlo[a], __shfl_xor(wlo[a], o, 64));
        whi[a] = fmaxf(whi[a], __shfl_xor(whi[a], o, 64));
      }
    }
    // wave-wide values: parked in scalar registers (seven SGPRs instead of seven VGPRs through the loop)
    wb = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(__builtin_fmaf(b, 1.0001f, 1.0e-4f))));
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      wlo[a] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(wlo[a])));
      whi[a] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(whi[a])));
    }
  }
  auto tile_culled = [&](int tile) {   // wave-uniform
    if (!CULL || !tile_box) return false;
    const float* bx = tile_box + 6 * (size_t)(t0 / kTile + tile);
    float d2 = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float g = fmaxf(0.f, fmaxf(wlo[a] - bx[3 + a], bx[a] - whi[a]));
      d2 = __builtin_fmaf(g, g, d2);
    }
    return d2 > wb;
  };
  if (CULL && tile_box && ntiles == 1) {
    if (!__syncthreads_or(tile_culled(0) ? 0 : 1)) return;   // block-uniform: nobody posts (no lane's neighbour can sit in this tile)
  }

  auto stage = [&](int tile, int buf) {
    const int j = t0 + tile * kTile + tid;
    if (!FILTER) {
      float x = 3.0e38f, y = 3.0e38f, z = 3.0e38f;  // padding: d2 = +inf, never wins
      if (j < t1) {
        double dx, dy, dz;
        xform64(tt, tgt[3 * (size_t)j], tgt[3 * (size_t)j + 1], tgt[3 * (size_t)j + 2], dx, dy, dz);
        x = (float)dx; y = (float)dy; z = (float)dz;
      }
      lds[buf][0][tid] = x;
      lds[buf][1][tid] = y;
      lds[buf][2][tid] = z;
      return;
    }
    float x = 0.f, y = 0.f, z = 0.f, t2 = 3.0e38f, t2m = 0.f;  // padding: score 3e38, never evaluated
    if (j < t1) {
      double dx, dy, dz;
      xform64(tt, tgt[3 * (size_t)j], tgt[3 * (size_t)j + 1], tgt[3 * (size_t)j + 2], dx, dy, dz);
      x = (float)dx; y = (float)dy; z = (float)dz;
      t2 = __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x));
      t2m = t2;
    }
    lds[buf][0][tid] = -2.f * x;
    lds[buf][1][tid] = -2.f * y;
    lds[buf][2][tid] = -2.f * z;
    lds[buf][FILTER ? 3 : 0][tid] = t2;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) t2m = fmaxf(t2m, __shfl_xor(t2m, o, 64));
    if ((tid & 63) == 0) tile_t2[buf][tid >> 6] = t2m;
  };
  auto threshold = [&](int r, float t2max) {
    return __builtin_fmaf(64.f * kU, q2[r] + t2max, __builtin_fmaf(EXACT ? thx[r] : best[r], 1.f + 8.f * kU, -q2[r]));
  };

  if (ntiles > 0) stage(0, 0);
  __syncthreads();
  for (int tile = 0; tile < ntiles; ++tile) {
    const int buf = tile & 1;
    if (tile + 1 < ntiles) stage(tile + 1, buf ^ 1);
    const int jbase = t0 + tile * kTile;
    // (scalar by construction — every input is a wave reduction — and told so, lest the branch be laid out with the barrier
    // on both of its sides under an execution mask)
    if (__builtin_amdgcn_readfirstlane(tile_culled(tile) ? 1 : 0) != 0) {   // the wave still takes part in the staging and its barrier
      __syncthreads();
      continue;
    }
    if (!FILTER) {
#pragma unroll 2
      for (int g = 0; g < kTile; g += kGroup) {
        float tx[kGroup], ty[kGroup], tz[kGroup];
#pragma unroll
        for (int v = 0; v < kGroup; v += 4) {
          const float4 a = *reinterpret_cast<const float4*>(&lds[buf][0][g + v]);
          const float4 c = *reinterpret_cast<const float4*>(&lds[buf][1][g + v]);
          const float4 e = *reinterpret_cast<const float4*>(&lds[buf][2][g + v]);
          tx[v] = a.x; tx[v + 1] = a.y; tx[v + 2] = a.z; tx[v + 3] = a.w;
          ty[v] = c.x; ty[v + 1] = c.y; ty[v + 2] = c.z; ty[v + 3] = c.w;
          tz[v] = e.x; tz[v + 1] = e.y; tz[v + 2] = e.z; tz[v + 3] = e.w;
        }
#pragma unroll
        for (int r = 0; r < RQ; ++r) {
          float d[kGroup];
#pragma unroll
          for (int v = 0; v < kGroup; ++v) {
            const float dx = qx[r] - tx[v], dy = qy[r] - ty[v], dz = qz[r] - tz[v];
            d[v] = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
          }
          // three v_min3 + one v_min (min-type ops issue at 0.6x the add rate: keep them few)
          const float m = fminf(fminf(fminf(d[3], d[4]), d[5]),
                                fminf(fminf(d[6], d[7]), fminf(fminf(d[0], d[1]), d[2])));
          if (EXACT ? __any(m <= thx[r]) : __any(m < best[r])) {  // wave-uniform
            const float best0 = best[r];
            const int bidx0 = bidx[r];
#pragma unroll
            for (int v = 0; v < kGroup; ++v) {
              const bool up = d[v] < best[r];
              best[r] = up ? d[v] : best[r];
              bidx[r] = up ? (jbase + g + v) : bidx[r];
            }
            if (EXACT) {
              // near ties of the group against its outcome: the bound only shrinks with best, so one bound — that of the
              // new best — decides for every entry and for the displaced best (one sqrt per group, not per entry)
              thx[r] = nearthr(r, best[r]);
              bool t = (bidx0 >= 0) & (bidx0 != bidx[r]) & (best0 <= thx[r]);
#pragma unroll
              for (int v = 0; v < kGroup; ++v) t |= (d[v] <= thx[r]) & (d[v] < 2.9e38f) & (jbase + g + v != bidx[r]);
              am[r] |= t;
            }
          }
        }
      }
      __syncthreads();
      continue;
    }
    const float t2max = fmaxf(fmaxf(tile_t2[buf][0], tile_t2[buf][1]), fmaxf(tile_t2[buf][2], tile_t2[buf][3]));
#pragma unroll
    for (int r = 0; r < RQ; ++r) thr[r] = threshold(r, t2max);
#pragma unroll 2
    for (int g = 0; g < kTile; g += kGroup) {
      float tx[kGroup], ty[kGroup], tz[kGroup], tw[kGroup];
#pragma unroll
      for (int v = 0; v < kGroup; v += 4) {
        const float4 a = *reinterpret_cast<const float4*>(&lds[buf][0][g + v]);
        const float4 c = *reinterpret_cast<const float4*>(&lds[buf][1][g + v]);
        const float4 e = *reinterpret_cast<const float4*>(&lds[buf][2][g + v]);
        const float4 w = *reinterpret_cast<const float4*>(&lds[buf][FILTER ? 3 : 0][g + v]);
        tx[v] = a.x; tx[v + 1] = a.y; tx[v + 2] = a.z; tx[v + 3] = a.w;
        ty[v] = c.x; ty[v + 1] = c.y; ty[v + 2] = c.z; ty[v + 3] = c.w;
        tz[v] = e.x; tz[v + 1] = e.y; tz[v + 2] = e.z; tz[v + 3] = e.w;
        tw[v] = w.x; tw[v + 1] = w.y; tw[v + 2] = w.z; tw[v + 3] = w.w;
      }
#pragma unroll
      for (int r = 0; r < RQ; ++r) {
        float sc[kGroup];
#pragma unroll
        for (int v = 0; v < kGroup; ++v)
          sc[v] = __builtin_fmaf(qx[r], tx[v], __builtin_fmaf(qy[r], ty[v], __builtin_fmaf(qz[r], tz[v], tw[v])));
        // three v_min3 + one v_min (min-type ops issue at 0.6x the add rate: keep them few)
        const float m = fminf(fminf(fminf(sc[3], sc[4]), sc[5]),
                              fminf(fminf(sc[6], sc[7]), fminf(fminf(sc[0], sc[1]), sc[2])));
        if (__any(m <= thr[r])) {  // wave-uniform
          const float th = thr[r];   // the group's candidates against ONE threshold: a lowered best only prunes more
          const float best0 = best[r];
          const int bidx0 = bidx[r];
          float dd[kGroup];          // EXACT: the exact values of the entries that passed on this lane (inf otherwise)
#pragma unroll
          for (int v = 0; v < kGroup; ++v) {
            if (EXACT) dd[v] = __builtin_inff();
            if (__any(sc[v] <= th)) {           // wave-uniform per entry: typically one or two of the eight
              // -0.5 * (-2 t) = t exactly: the same differences as dx = qx - tx
              const float dx = qx[r] + 0.5f * tx[v], dy = qy[r] + 0.5f * ty[v], dz = qz[r] + 0.5f * tz[v];
              const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
              const int j = jbase + g + v;
              const bool pass = (sc[v] <= th) & (sc[v] < 2.9e38f);            // 3e38: padding
              const bool up = pass & ((d2 < best[r]) | ((d2 == best[r]) & (j < bidx[r])));
              if (EXACT) dd[v] = pass ? d2 : dd[v];
              best[r] = up ? d2 : best[r];
              bidx[r] = up ? j : bidx[r];
            }
          }
          if (EXACT) {               // as in the plain loop: one bound, that of the group's outcome
            thx[r] = nearthr(r, best[r]);
            bool t = (bidx0 >= 0) & (bidx0 != bidx[r]) & (best0 <= thx[r]);
#pragma unroll
            for (int v = 0; v < kGroup; ++v) t |= (dd[v] <= thx[r]) & (jbase + g + v != bidx[r]);
            am[r] |= t;
          }
          thr[r] = threshold(r, t2max);
        }
      }
    }
    __syncthreads();
  }

  if (packed) {
    // one 64-bit atomic min per (query, target split) instead of per-split partial arrays: d2 >= 0, so its bit
    // pattern orders like the value, and equal distances keep the lower index — the very winner the ascending merge
    // of the partials picks, whatever order the atomics land in.  A warm-started lane that found nothing better than
    // its starting neighbour w posts only from the split that owns w: every other split would post the same word
    // (30 of 31 ICP passes: ~1 atomic per point instead of one per split).  The atomics of a lane's RQ queries are
    // issued together and their returned words examined afterwards (EXACT).
    unsigned long long old[RQ];
    bool posted[RQ];
#pragma unroll
    for (int r = 0; r < RQ; ++r) {
      const int qi = query_of<RQ>(r);
      const bool own = widx[r] < 0 || bidx[r] != widx[r] || (widx[r] >= t0 && widx[r] < t1);
      posted[r] = qi < Nq && bidx[r] >= 0 && own;
      old[r] = ~0ull;
      if (posted[r]) {
        const unsigned long long mine = ((unsigned long long)__float_as_uint(best[r]) << 32) | (unsigned int)bidx[r];
        if (EXACT) old[r] = atomicMin(&packed[(size_t)b * Nq + qi], mine);
        else atomicMin(&packed[(size_t)b * Nq + qi], mine);
      }
    }
    if (EXACT) {
#pragma unroll
      for (int r = 0; r < RQ; ++r) {
        const int qi = query_of<RQ>(r);
        // the value this one displaced, or failed to displace, is another target split's winner: a near tie between
        // the two is a near tie of the query (every pair of posting splits meets here through the slot's history)
        const float od = __uint_as_float((unsigned int)(old[r] >> 32));
        const int oi = (int)(unsigned int)old[r];
        if (posted[r] && old[r] != ~0ull && oi != bidx[r]) am[r] |= fmaxf(od, best[r]) <= nearthr(r, fminf(od, best[r]));
        if (qi < Nq && am[r]) amb[(size_t)b * Nq + qi] = 1;
      }
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < RQ; ++r) {
    const int qi = query_of<RQ>(r);
    if (qi < Nq) {
      const size_t o = ((size_t)b * nsplit + split) * Nq + qi;
      if (unresolved && part_idx[o] != kUnresolved) continue;
      part_d2[o] = best[r];
      part_idx[o] = bidx[r];
    }
  }
}

#include "nn_grid.hpp"   // the two exact grid searches, their build kernels and workspace helpers

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Merge the per-split winners, re-evaluate the distance in f64, apply the radius, and reduce
// the block's sums into part_sums[b][blockIdx.x][kNV] (fixed tree: lanes, then waves in order).
template <bool WANT_COV>
__global__ __launch_bounds__(kThreads) void nn_finalize_kernel(
    const float* __restrict__ qry, int Nq, const float* __restrict__ tgt,
    const double* __restrict__ Tq, const double* __restrict__ Tt, int nsplit, double radius,
    const float* __restrict__ part_d2, const int32_t* __restrict__ part_idx, int b0,
    int32_t* __restrict__ nn_idx, double* __restrict__ nn_d, double* __restrict__ part_sums,
    const int32_t* __restrict__ skip, const unsigned long long* __restrict__ packed = nullptr) {
  __shared__ double red[kThreads / 64][kNV];
  if (skip && *skip) return;
  const int tid = threadIdx.x;
  const int bl = blockIdx.y;  // batch item inside this chunk
  const int b = b0 + bl;
  const int qi = blockIdx.x * kThreads + tid;
  const double* tq = Tq ? Tq + 12 * (size_t)b : nullptr;
  const double* tt = Tt ? Tt + 12 * (size_t)b : nullptr;

  double v[kNV];
#pragma unroll
  for (int k = 0; k < kNV; ++k) v[k] = 0.0;

  if (qi < Nq) {
    float best = __builtin_inff();
    int bi = -1;
    if (packed) {          // the target splits met in one 64-bit atomic min per query (isr_nn_batched, brute force)
      const unsigned long long pk = packed[(size_t)bl * Nq + qi];
      if (pk != ~0ull) { best = __uint_as_float((unsigned int)(pk >> 32)); bi = (int)(unsigned int)pk; }
    } else
    for (int s = 0; s < nsplit; ++s) {
      const size_t o = ((size_t)bl * nsplit + s) * Nq + qi;
      const float d2 = part_d2[o];
      if (d2 < best) { best = d2; bi = part_idx[o]; }
    }
    (void)best;
    double d = __builtin_inf();
    bool counted = false;
    if (bi >= 0) {
      double q0, q1, q2, t0, t1, t2;
      xform64(tq, qry[3 * (size_t)qi], qry[3 * (size_t)qi + 1], qry[3 * (size_t)qi + 2], q0, q1, q2);
      xform64(tt, tgt[3 * (size_t)bi], tgt[3 * (size_t)bi + 1], tgt[3 * (size_t)bi + 2], t0, t1, t2);
      const double ex = q0 - t0, ey = q1 - t1, ez = q2 - t2;
      const double s2 = fma(ez, ez, fma(ey, ey, ex * ex));
      d = sqrt(s2);
      counted = (radius < 0.0) || (s2 <= radius * radius);
      if (counted) {
        v[0] = d; v[1] = s2; v[2] = 1.0;
        if (WANT_COV) {
          const double q[3] = {q0, q1, q2}, t[3] = {t0, t1, t2};
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            v[3 + r] = q[r];
            v[6 + r] = t[r];
#pragma unroll
            for (int c = 0; c < 3; ++c) v[9 + 3 * r + c] = q[r] * t[c];
          }
        }
      }
    }
    if (nn_idx) nn_idx[(size_t)b * Nq + qi] = counted ? bi : -1;
    if (nn_d) nn_d[(size_t)b * Nq + qi] = d;
  }

  constexpr int nv = WANT_COV ? kNV : 3;
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int k = 0; k < nv; ++k) {
    const double s = wave_sum(v[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (tid < kNV) {
    double s = 0.0;
    if (tid < nv) s = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
    part_sums[((size_t)b * gridDim.x + blockIdx.x) * kNV + tid] = s;
  }
}

__global__ void nn_reduce_kernel(const double* __restrict__ part_sums, int nblk, int B,
                                 double* __restrict__ sum_d, double* __restrict__ sum_d2,
                                 int32_t* __restrict__ n_in, double* __restrict__ cov) {
  const int b = blockIdx.x * blockDim.y + threadIdx.y;
  const int k = threadIdx.x;
  if (b >= B || k >= kNV) return;
  double s = 0.0;
  for (int i = 0; i < nblk; ++i) s += part_sums[((size_t)b * nblk + i) * kNV + k];
  if (k == 0 && sum_d) sum_d[b] = s;
  if (k == 1 && sum_d2) sum_d2[b] = s;
  if (k == 2 && n_in) n_in[b] = (int32_t)s;
  if (k >= 3 && cov) cov[(size_t)b * 16 + (k - 3)] = s;
  if (k == 2 && cov) cov[(size_t)b * 16 + 15] = s;  // the count again, so one copy of cov carries everything
}


// a8 ADD (inference.py:116-117): mean_v || Ta v - Tb v ||, one block per pose pair, f64.
__global__ __launch_bounds__(kThreads) void add_metric_kernel(const float* __restrict__ verts, int V,
                                                              const double* __restrict__ Ta,
                                                              const double* __restrict__ Tb,
                                                              double* __restrict__ out) {
  __shared__ double red[kThreads / 64];
  const int b = blockIdx.x;
  const double* ta = Ta ? Ta + 12 * (size_t)b : nullptr;
  const double* tb = Tb ? Tb + 12 * (size_t)b : nullptr;
  double s = 0.0;
  for (int i = threadIdx.x; i < V; i += kThreads) {
    const float x = verts[3 * (size_t)i], y = verts[3 * (size_t)i + 1], z = verts[3 * (size_t)i + 2];
    double a0, a1, a2, b0, b1, b2;
    xform64(ta, x, y, z, a0, a1, a2);
    xform64(tb, x, y, z, b0, b1, b2);
    const double ex = a0 - b0, ey = a1 - b1, ez = a2 - b2;
    s += sqrt(fma(ez, ez, fma(ey, ey, ex * ex)));
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[b] = (((red[0] + red[1]) + red[2]) + red[3]) / (double)V;
}


// ------------------------------------------------------------------------------- K4 ICP loop
// registration_icp(source, target, threshold, init, PointToPoint) of icp.py:101-103, enqueued as
// max_iter + 1 evaluation passes with NO host round trip: after each pass one thread reduces the
// block sums, applies Open3D's stopping rule (|d fitness| < rel_fitness and |d rmse| < rel_rmse, or
// the iteration budget) and, if it continues, composes T <- dT T where dT is the rigid fit of the
// matched pairs (Horn's closed form: largest eigenvector of the 4x4 quaternion matrix by cyclic
// Jacobi — always a proper rotation, equal to the SVD/Kabsch solution).  A device flag turns the
// remaining launches into no-ops once the rule fires.
struct IcpState {
  double prev_fit, prev_rmse;
  int32_t iter, done;
  int32_t ticket, pad;   // workgroups that have delivered their sums in the current pass
};

// Cyclic Jacobi on a symmetric 4x4; every loop has constant bounds and is unrolled so A and V
// stay in registers (indexed dynamically they live in scratch memory: 60 us per call instead of 10).
__device__ void jacobi4_largest(double A[4][4], double q[4]) {
  double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  for (int sweep = 0; sweep < 16; ++sweep) {
    double off = 0.0, dia = 0.0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      dia += A[p][p] * A[p][p];
#pragma unroll
      for (int r = p + 1; r < 4; ++r) off += A[p][r] * A[p][r];
    }
    // off-diagonal mass below 1e-34 of the diagonal's: the eigenvector is converged to ~1e-17 — Jacobi
    // converges quadratically, so the sweeps this saves (it used to run until off underflowed) are pure
    // serial latency in a one-thread tail
    if (off <= 1e-34 * dia) break;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int r = p + 1; r < 4; ++r) {
        if (A[p][r] != 0.0) {
          const double theta = (A[r][r] - A[p][p]) / (2.0 * A[p][r]);
          const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#pragma unroll
          for (int k = 0; k < 4; ++k) {  // A <- A J
            const double akp = A[k][p], akr = A[k][r];
            A[k][p] = c * akp - sn * akr;
            A[k][r] = sn * akp + c * akr;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {  // A <- J^T A
            const double apk = A[p][k], ark = A[r][k];
            A[p][k] = c * apk - sn * ark;
            A[r][k] = sn * apk + c * ark;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const double vkp = V[k][p], vkr = V[k][r];
            V[k][p] = c * vkp - sn * vkr;
            V[k][r] = sn * vkp + c * vkr;
          }
        }
      }
  }
  double lam = A[0][0];
#pragma unroll
  for (int k = 0; k < 4; ++k) q[k] = V[k][0];
#pragma unroll
  for (int j = 1; j < 4; ++j) {
    if (A[j][j] > lam) {
      lam = A[j][j];
#pragma unroll
      for (int k = 0; k < 4; ++k) q[k] = V[k][j];
    }
  }
}

constexpr int kIcpLanes = 16;

// The end of an ICP evaluation pass, by ONE workgroup of any size: reduce the block sums in a fixed
// shape (lane l of sum k adds blocks l, l + kIcpLanes, ...; the kIcpLanes partials are added in lane
// order: run-to-run reproducible and the same for every caller), apply the stopping rule, compose T.
__device__ void icp_reduce_and_update(const double* __restrict__ part_sums, int nblk, int Ns, int max_iter,
                                      double rel_fitness, double rel_rmse, double* __restrict__ T,
                                      IcpState* __restrict__ st, double* __restrict__ result) {
  __shared__ double part[kNV][kIcpLanes];
  __shared__ double v[kNV];
  for (int idx = threadIdx.x; idx < kNV * kIcpLanes; idx += blockDim.x) {
    const int k = idx / kIcpLanes, l = idx % kIcpLanes;
    double s = 0.0;
    for (int i = l; i < nblk; i += kIcpLanes) s += part_sums[(size_t)i * kNV + k];
    part[k][l] = s;
  }
  __syncthreads();
  if (threadIdx.x < kNV) {
    double s = 0.0;
    for (int l = 0; l < kIcpLanes; ++l) s += part[threadIdx.x][l];
    v[threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  const double n = v[2];
  const double fit = n / (double)Ns;
  const double rmse = n > 0 ? sqrt(v[1] / n) : 0.0;
  const int it = st->iter;
  result[0] = fit; result[1] = rmse; result[2] = (double)it; result[3] = n;
  const bool conv = it > 0 && fabs(st->prev_fit - fit) < rel_fitness && fabs(st->prev_rmse - rmse) < rel_rmse;
  if (conv || it >= max_iter || n < 3) { st->done = 1; return; }
  st->prev_fit = fit; st->prev_rmse = rmse; st->iter = it + 1;
  // rigid fit of the matched pairs: S = sum (q - mq)(t - mt)^T
  double mq[3], mt[3], S[3][3];
  for (int a = 0; a < 3; ++a) { mq[a] = v[3 + a] / n; mt[a] = v[6 + a] / n; }
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) S[a][b] = v[9 + 3 * a + b] - n * mq[a] * mt[b];
  double N4[4][4] = {
      {S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0]},
      {S[1][2] - S[2][1], S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2]},
      {S[2][0] - S[0][2], S[0][1] + S[1][0], -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1]},
      {S[0][1] - S[1][0], S[2][0] + S[0][2], S[1][2] + S[2][1], -S[0][0] - S[1][1] + S[2][2]}};
  double q[4];
  jacobi4_largest(N4, q);
  const double nq = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double w = q[0] * nq, x = q[1] * nq, y = q[2] * nq, z = q[3] * nq;
  const double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)},
                          {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
                          {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
  double U[3][4];
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) U[a][b] = R[a][b];
    U[a][3] = mt[a] - (R[a][0] * mq[0] + R[a][1] * mq[1] + R[a][2] * mq[2]);
  }
  double Tn[12];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 4; ++b)
      Tn[4 * a + b] = U[a][0] * T[b] + U[a][1] * T[4 + b] + U[a][2] * T[8 + b] + (b == 3 ? U[a][3] : 0.0);
  for (int k = 0; k < 12; ++k) T[k] = Tn[k];
}

// Brute-force loop: the search left one packed (f32 d2 bits, index) minimum per source point.  Every
// workgroup finishes its 256 points exactly as nn_finalize_kernel<true> does (f64 distance of the
// winner, radius test, the 18 sums by the same fixed tree), re-arms the packed slots, and takes a
// ticket; the LAST workgroup to arrive (agent-scope fences order the block sums before the ticket)
// runs the update.  Two launches per ICP iteration, no partial arrays.
__global__ __launch_bounds__(kThreads) void icp_finalize_update_kernel(
    const float* __restrict__ src, int Ns, const float* __restrict__ tgt, int Nt, double radius,
    unsigned long long* __restrict__ packed, int32_t* __restrict__ prev_idx, int32_t* __restrict__ amb,
    double* __restrict__ part_sums, int max_iter,
    double rel_fitness, double rel_rmse, double* __restrict__ T, IcpState* __restrict__ st, double* __restrict__ result) {
  __shared__ double red[kThreads / 64][kNV];
  __shared__ int last;
  __shared__ int nflag, flist[kThreads];
  __shared__ double wd[4][kThreads / 64];
  __shared__ int wi[4][kThreads / 64];
  if (st->done) return;
  const int tid = threadIdx.x;
  const int qi = blockIdx.x * kThreads + tid;
  // ---- near ties the search flagged (nn_search_kernel<.., EXACT>): the exact nearest neighbour over ALL targets in
  // f64 — squared distance by the fma chain the sums below use, lowest index on exact ties — decided by the whole
  // workgroup, one flagged point at a time (one or two points per pass on 20 000-point clouds, in one or two workgroups)
  if (tid == 0) nflag = 0;
  __syncthreads();
  if (qi < Ns && amb[qi]) {
    amb[qi] = 0;
    flist[atomicAdd(&nflag, 1)] = qi;
  }
  __syncthreads();
  const int nf = nflag;                             // block-uniform
  // up to kRes flagged points per sweep over the targets: a target is loaded once and tried against each of them
  // (the sweep is bound by load latency — round 3's first version resolved one point per sweep, ~35 us each)
  constexpr int kRes = 4;
  for (int f0 = 0; f0 < nf; f0 += kRes) {
    double qx[kRes], qy[kRes], qz[kRes], bd[kRes];
    int bi[kRes];
#pragma unroll
    for (int k = 0; k < kRes; ++k) {
      const int pq = flist[min(f0 + k, nf - 1)];
      xform64(T, src[3 * (size_t)pq], src[3 * (size_t)pq + 1], src[3 * (size_t)pq + 2], qx[k], qy[k], qz[k]);
      bd[k] = __builtin_inf();
      bi[k] = 0x7fffffff;
    }
#pragma unroll 4
    for (int j = tid; j < Nt; j += kThreads) {
      const double tx = tgt[3 * (size_t)j], ty = tgt[3 * (size_t)j + 1], tz = tgt[3 * (size_t)j + 2];
#pragma unroll
      for (int k = 0; k < kRes; ++k) {
        const double ex = qx[k] - tx, ey = qy[k] - ty, ez = qz[k] - tz;
        const double s2 = fma(ez, ez, fma(ey, ey, ex * ex));
        if (s2 < bd[k]) { bd[k] = s2; bi[k] = j; }  // ascending j per thread: ties keep the lower index
      }
    }
#pragma unroll
    for (int k = 0; k < kRes; ++k) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double od = __shfl_xor(bd[k], off, 64);
        const int oi = __shfl_xor(bi[k], off, 64);
        if (od < bd[k] || (od == bd[k] && oi < bi[k])) { bd[k] = od; bi[k] = oi; }
      }
      if ((tid & 63) == 0) { wd[k][tid >> 6] = bd[k]; wi[k][tid >> 6] = bi[k]; }
    }
    __syncthreads();
    if (tid < kRes && f0 + tid < nf) {
      double d = wd[tid][0];
      int i = wi[tid][0];
      for (int w = 1; w < kThreads / 64; ++w)
        if (wd[tid][w] < d || (wd[tid][w] == d && wi[tid][w] < i)) { d = wd[tid][w]; i = wi[tid][w]; }
      packed[flist[f0 + tid]] = ((unsigned long long)__float_as_uint((float)d) << 32) | (unsigned int)i;
    }
    __syncthreads();
  }
  double v[kNV];
#pragma unroll
  for (int k = 0; k < kNV; ++k) v[k] = 0.0;
  if (qi < Ns) {
    const unsigned long long pk = packed[qi];
    packed[qi] = ~0ull;
    prev_idx[qi] = pk != ~0ull ? (int)(unsigned int)pk : -1;     // the next pass starts from this neighbour
    if (pk != ~0ull) {
      const int bi = (int)(unsigned int)pk;
      double q0, q1, q2;
      xform64(T, src[3 * (size_t)qi], src[3 * (size_t)qi + 1], src[3 * (size_t)qi + 2], q0, q1, q2);
      const double t0 = tgt[3 * (size_t)bi], t1 = tgt[3 * (size_t)bi + 1], t2 = tgt[3 * (size_t)bi + 2];
      const double ex = q0 - t0, ey = q1 - t1, ez = q2 - t2;
      const double s2 = fma(ez, ez, fma(ey, ey, ex * ex));
      if (s2 <= radius * radius) {
        v[0] = sqrt(s2); v[1] = s2; v[2] = 1.0;
        const double q[3] = {q0, q1, q2}, t[3] = {t0, t1, t2};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          v[3 + r] = q[r];
          v[6 + r] = t[r];
#pragma unroll
          for (int c = 0; c < 3; ++c) v[9 + 3 * r + c] = q[r] * t[c];
        }
      }
    }
  }
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int k = 0; k < kNV; ++k) {
    const double s = wave_sum(v[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (tid < kNV) part_sums[(size_t)blockIdx.x * kNV + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
  __threadfence();                                  // the block's sums are visible device-wide ...
  __syncthreads();
  if (tid == 0) last = (atomicAdd(&st->ticket, 1) == (int)gridDim.x - 1);   // ... before its ticket is
  __syncthreads();
  if (!last) return;                                // block-uniform
  __threadfence();
  if (tid == 0) st->ticket = 0;
  icp_reduce_and_update(part_sums, gridDim.x, Ns, max_iter, rel_fitness, rel_rmse, T, st, result);
}

__global__ void icp_init_kernel(IcpState* st, double* T, unsigned long long* packed, int32_t* amb, int Ns) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < Ns) { packed[i] = ~0ull; amb[i] = 0; }
  if (i != 0) return;
  st->prev_fit = 0; st->prev_rmse = 0; st->iter = 0; st->done = 0; st->ticket = 0;
  T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}

// max |t|^2 over the target cloud (f32, rounded up a little): the rounding-displacement bound of the EXACT search
__global__ __launch_bounds__(kThreads) void cloud_r2max_kernel(const float* __restrict__ tgt, int Nt, float* __restrict__ out) {
  __shared__ float red[kThreads / 64];
  float m = 0.f;
  for (int j = threadIdx.x; j < Nt; j += kThreads) {
    const float x = tgt[3 * (size_t)j], y = tgt[3 * (size_t)j + 1], z = tgt[3 * (size_t)j + 2];
    m = fmaxf(m, __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x)));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) *out = 1.0001f * fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

struct NNPlan {
  int rq;          // queries per lane
  int qblocks;     // search grid.x
  int nsplit;      // search grid.y
  int split_len;   // targets per split (multiple of kTile)
  int fblocks;     // finalize grid.x
  int bchunk;      // batch items per search launch
  bool grid;       // uniform-grid search (+ brute-force pass for unresolved queries) instead of brute force
  bool tile;       // block-cooperative grid search (queries binned too)
};

constexpr size_t kPartBudget = size_t(192) << 20;  // bytes of per-query partials per chunk

NNPlan make_plan(int Nq, int Nt, int B, bool brute_only = false) {
  NNPlan p;
  p.rq = (Nq >= 4 * kThreads) ? 4 : 1;
  // tuning knobs (experiments only): ISR_TUNE_NN_PLAN_RQ / _BLOCKS
  const int rq_knob = isr::tuning(ISR_TUNE_NN_PLAN_RQ);
  const long want_env = isr::tuning(ISR_TUNE_NN_PLAN_BLOCKS);
  if (rq_knob == 1 || rq_knob == 4) p.rq = rq_knob;
  p.qblocks = (Nq + p.rq * kThreads - 1) / (p.rq * kThreads);
  // 256 CUs x 8 workgroups are resident at once; aim for two such rounds.  More key-range splits
  // would even out the last round, but every split restarts its running minimum, and while a
  // minimum is young the wave-uniform update path runs for most groups (tools/nn_plan_sweep.py:
  // 63 pairs of 20 000^2 points: 3.67 ms at 8192 wanted blocks, 3.18 ms at 4096).
#ifndef ISR_NN_WANT_BLOCKS
#define ISR_NN_WANT_BLOCKS 4096
#endif
  const long want = want_env > 0 ? want_env : ISR_NN_WANT_BLOCKS;
  long ns = (want + (long)p.qblocks * B - 1) / ((long)p.qblocks * B);
  const int max_split = (Nt + kTile - 1) / kTile;
  if (ns < 1) ns = 1;
  if (ns > max_split) ns = max_split;
  int tiles_per_split = (max_split + (int)ns - 1) / (int)ns;
  p.split_len = tiles_per_split * kTile;
  p.nsplit = (Nt + p.split_len - 1) / p.split_len;
  p.fblocks = (Nq + kThreads - 1) / kThreads;
  // Three exact searches (tests compare them bit for bit; the ISR_TUNE_NN_PATH knob forces one):
  //  0 brute force — the default for one or a few batch items (ICP steps, single ADD-S / Chamfer calls):
  //    at 20 000 points the launch is a few dozen microseconds and nothing is cheaper to set up;
  //  2 block-cooperative grid — the default for batches (the Chamfer pick, the n x n vote): measured
  //    (tools/nn_tile_sweep.py, profiles/r01_nn_grid_vs_brute.txt) 5x / 12x faster than brute force at
  //    20 000 / 50 000 points when the two clouds are rotated copies a few degrees apart, 4.6x / 8.7x at
  //    15 degrees, and on a par with it (1.0x / 1.3x) for unrelated orientations, where its growing box
  //    ends in the full scan;
  //  1 per-lane grid — opt-in only: thread-private, latency-bound cell walks; 4-25x SLOWER than brute
  //    force once neighbours are more than a cell or two away.
  p.grid = false;
  p.tile = B >= 4 && Nq >= 1024 && Nt >= 4096 && (long)Nq * B >= (1L << 17);
  if (const int path = isr::tuning(ISR_TUNE_NN_PATH); path >= 0) { p.grid = path == 1; p.tile = path == 2; }
  if (brute_only) p.grid = p.tile = false;
  if (p.grid || p.tile) {
    p.nsplit = 1;
    p.split_len = max_split * kTile;
  }
  const size_t per_b = (size_t)p.nsplit * Nq * 8;
  long bc = (long)(kPartBudget / (per_b ? per_b : 1));
  if (bc < 1) bc = 1;
  if (bc > B) bc = B;
  if (bc > 65535) bc = 65535;  // grid.z / grid.y limit
  p.bchunk = (int)bc;
  return p;
}

// the brute-force search of a plan: plain loop or filter loop (nn_search_kernel) — the filter needs settled minima,
// i.e. a warm start or a target range of several tiles per workgroup
constexpr int kFilterMinTiles = 4;
void launch_search(const NNPlan& p, const dim3& grid, hipStream_t stream, const float* qry, int Nq, const float* tgt, int Nt,
                   const double* tq, const double* tt, float* part_d2, int32_t* part_idx, const int32_t* skip,
                   const int32_t* unresolved, unsigned long long* packed, const int32_t* warm, int32_t* amb = nullptr,
                   const float* t2_bound = nullptr, const float* tile_box = nullptr, float cull_r2 = 0.f) {
  bool filter = warm != nullptr || p.split_len >= kFilterMinTiles * kTile;
  // tuning hook (experiments only) for cold searches; a warm start always takes the filter loop (its tie rule —
  // equal distance, lower index — is what makes a warm-started lane return the cold winner)
  if (const int f = isr::tuning(ISR_TUNE_NN_FILTER); f >= 0 && !warm) filter = f == 1;
#define ISR_SEARCH(RQv, Fv)                                                                                              \
  nn_search_kernel<RQv, Fv><<<grid, kThreads, 0, stream>>>(qry, Nq, tgt, Nt, tq, tt, p.split_len, p.nsplit, part_d2, part_idx, \
                                                           skip, unresolved, packed, warm)
  // (the tile cull is compiled into the warm-started filter kernel only: in the cold plain loop its state took the kernel from
  // 167 to 191 registers — and beside K1, whose three waves per SIMD hold 504 of the 512, a wave that needs more than one K1
  // wave's 168 waits for two of them to leave: the ICP's first pass went from 0.8 to 6.7 ms inside the step)
#define ISR_SEARCH_X(RQv, Fv)                                                                                            \
  nn_search_kernel<RQv, Fv, true, Fv><<<grid, kThreads, 0, stream>>>(qry, Nq, tgt, Nt, tq, tt, p.split_len, p.nsplit, part_d2, \
                                                                     part_idx, skip, unresolved, packed, warm, amb, t2_bound, tile_box, cull_r2)
  if (amb) {       // the ICP loop: near ties are flagged for the exact decision (packed slots, no target transform)
    if (p.rq == 4) { if (filter) ISR_SEARCH_X(4, true); else ISR_SEARCH_X(4, false); }
    else { if (filter) ISR_SEARCH_X(1, true); else ISR_SEARCH_X(1, false); }
    return;
  }
  if (p.rq == 4) { if (filter) ISR_SEARCH(4, true); else ISR_SEARCH(4, false); }
  else { if (filter) ISR_SEARCH(1, true); else ISR_SEARCH(1, false); }
#undef ISR_SEARCH_X
#undef ISR_SEARCH
}

}  // namespace

extern "C" size_t isr_nn_batched_workspace_bytes(int Nq, int Nt, int B) {
  if (Nq <= 0 || Nt <= 0 || B <= 0) return 0;
  const NNPlan p = make_plan(Nq, Nt, B);
  size_t n = 0;
  n += isr::align_up((size_t)p.bchunk * p.nsplit * Nq * sizeof(float), 256);
  n += isr::align_up((size_t)p.bchunk * p.nsplit * Nq * sizeof(int32_t), 256);
  n += isr::align_up((size_t)B * p.fblocks * kNV * sizeof(double), 256);
  if (p.grid) n += grid_ws_bytes(Nt);
  if (p.tile) n += tile_ws_bytes(Nq, Nt);
  return n + 256;
}

extern "C" int isr_nn_batched(const float* qry, int Nq, const float* tgt, int Nt, const double* Tq,
                              const double* Tt, int B, double radius, double* sum_d,
                              double* sum_d2, int32_t* n_in, int32_t* nn_idx, double* nn_d,
                              double* cov, void* ws, size_t ws_bytes, isr_stream_t stream_) {
  ISR_REQUIRE(qry && tgt, "isr_nn_batched: null cloud pointer");
  ISR_REQUIRE(Nq > 0 && Nt > 0 && B > 0, "isr_nn_batched: Nq=%d Nt=%d B=%d must be positive", Nq,
              Nt, B);
  ISR_REQUIRE(sum_d || sum_d2 || n_in || nn_idx || nn_d || cov, "isr_nn_batched: no output requested");
  if (!ws || ws_bytes < isr_nn_batched_workspace_bytes(Nq, Nt, B)) {
    isr::set_error("isr_nn_batched: workspace %zu < %zu", ws_bytes,
                   isr_nn_batched_workspace_bytes(Nq, Nt, B));
    return ISR_ERR_WORKSPACE;
  }
  hipStream_t stream = isr::as_stream(stream_);
  const NNPlan p = make_plan(Nq, Nt, B);
  isr::Workspace w(ws, ws_bytes);
  float* part_d2 = w.take<float>((size_t)p.bchunk * p.nsplit * Nq);
  int32_t* part_idx = w.take<int32_t>((size_t)p.bchunk * p.nsplit * Nq);
  double* part_sums = w.take<double>((size_t)B * p.fblocks * kNV);
  GridWs gw{};
  TileWs tw{};
  if (p.grid) {
    gw = take_grid(w, Nt);
    build_grid(gw, tgt, Nt, stream);
    ISR_CHECK_LAUNCH("grid build");
  }
  if (p.tile) {
    tw = take_tile(w, Nq, Nt);
    build_tile(tw, qry, Nq, tgt, Nt, stream);
    ISR_CHECK_LAUNCH("tile grid build");
  }
  // beyond the radius nothing is counted; the true distance is only owed when nn_d is requested
  const float stop_radius = (radius >= 0.0 && !nn_d) ? (float)(radius * (1.0 + 1e-6)) : -1.f;

  for (int b0 = 0; b0 < B; b0 += p.bchunk) {
    const int nb = (B - b0 < p.bchunk) ? (B - b0) : p.bchunk;
    const double* tq = Tq ? Tq + 12 * (size_t)b0 : nullptr;
    const double* tt = Tt ? Tt + 12 * (size_t)b0 : nullptr;
    const dim3 grid(p.qblocks, p.nsplit, nb);
    const int32_t* unres = nullptr;
    if (p.grid) {
      nn_grid_search_kernel<<<dim3(p.fblocks, nb), kThreads, 0, stream>>>(qry, Nq, gw.desc, gw.start, gw.sorted, tq, tt,
                                                                          stop_radius, part_d2, part_idx,
                                                                          gw.unresolved, nullptr);
      ISR_CHECK_LAUNCH("nn_grid_search_kernel");
      unres = gw.unresolved;   // sticky across chunks: later chunks only re-check their own marks
    }
    // Brute force over several target splits: the splits' winners meet in ONE 64-bit atomic min per query (d2 >= 0, so its bit
    // pattern orders like the value; equal distances keep the lower index — the winner the ascending merge of per-split
    // partials picked) instead of nsplit x 8 bytes written and read back per query (round 3: 76 MB of HBM traffic for 5.6 MB
    // of clouds on the 32 x 20 000^2 batch).  The slots live in the partial arrays' space.
    unsigned long long* packed = nullptr;
    if (!p.tile && !p.grid && p.nsplit > 1 && ISR_NN_PACKED_BATCH) {
      packed = reinterpret_cast<unsigned long long*>(part_d2);
      ISR_CHECK_HIP(hipMemsetAsync(packed, 0xFF, (size_t)nb * Nq * sizeof(unsigned long long), stream));
    }
    if (p.tile)
      launch_tile_search(tw, Nq, Nt, tq, tt, nb, stop_radius, part_d2, part_idx, nullptr, stream);
    else
      launch_search(p, grid, stream, qry, Nq, tgt, Nt, tq, tt, packed ? nullptr : part_d2, packed ? nullptr : part_idx, nullptr, unres,
                    packed, nullptr);
    ISR_CHECK_LAUNCH("nn_search_kernel");
    const dim3 fgrid(p.fblocks, nb);
    if (cov)
      nn_finalize_kernel<true><<<fgrid, kThreads, 0, stream>>>(qry, Nq, tgt, Tq, Tt, p.nsplit,
                                                               radius, part_d2, part_idx, b0,
                                                               nn_idx, nn_d, part_sums, nullptr, packed);
    else
      nn_finalize_kernel<false><<<fgrid, kThreads, 0, stream>>>(qry, Nq, tgt, Tq, Tt, p.nsplit,
                                                                radius, part_d2, part_idx, b0,
                                                                nn_idx, nn_d, part_sums, nullptr, packed);
    ISR_CHECK_LAUNCH("nn_finalize_kernel");
  }
  if (sum_d || sum_d2 || n_in || cov) {
    const dim3 rblock(32, 8);
    nn_reduce_kernel<<<(B + 7) / 8, rblock, 0, stream>>>(part_sums, p.fblocks, B, sum_d, sum_d2,
                                                         n_in, cov);
    ISR_CHECK_LAUNCH("nn_reduce_kernel");
  }
  return ISR_OK;
}

extern "C" int isr_add_metric(const float* verts, int V, const double* Ta, const double* Tb, int B,
                              double* mean_out, isr_stream_t stream) {
  ISR_REQUIRE(verts && mean_out, "isr_add_metric: null pointer");
  ISR_REQUIRE(V > 0 && B > 0, "isr_add_metric: V=%d B=%d", V, B);
  add_metric_kernel<<<B, kThreads, 0, isr::as_stream(stream)>>>(verts, V, Ta, Tb, mean_out);
  ISR_CHECK_LAUNCH("add_metric_kernel");
  return ISR_OK;
}

extern "C" size_t isr_icp_workspace_bytes(int Ns, int Nt) {
  if (Ns <= 0 || Nt <= 0) return 0;
  const NNPlan p = make_plan(Ns, Nt, 1, true);
  return isr::align_up((size_t)p.fblocks * kNV * sizeof(double), 256) + isr::align_up((size_t)Ns * 8, 256) +
         2 * isr::align_up((size_t)Ns * 4, 256) + isr::align_up((size_t)((Nt + kTile - 1) / kTile) * 6 * sizeof(float), 256) + 2048;
}

extern "C" int isr_icp_point_to_point(const float* src, int Ns, const float* tgt, int Nt, double threshold,
                                      int max_iter, double rel_fitness, double rel_rmse, double* T_io,
                                      double* result, void* ws, size_t ws_bytes, isr_stream_t stream_) {
  ISR_REQUIRE(src && tgt && T_io && result, "isr_icp_point_to_point: null pointer");
  ISR_REQUIRE(Ns > 0 && Nt > 0 && max_iter >= 0 && threshold > 0, "isr_icp_point_to_point: Ns=%d Nt=%d max_iter=%d", Ns, Nt, max_iter);
  if (!ws || ws_bytes < isr_icp_workspace_bytes(Ns, Nt)) {
    isr::set_error("isr_icp_point_to_point: workspace %zu < %zu", ws_bytes, isr_icp_workspace_bytes(Ns, Nt));
    return ISR_ERR_WORKSPACE;
  }
  hipStream_t stream = isr::as_stream(stream_);
  // Always the brute-force search (the grid searches lose here: most source points have their neighbour 5-20 mm
  // away and a 20 mm ball covers a large part of the object — 18 ms / 103 ms against 2.5 ms per loop, round 2), with
  // near ties decided in f64: every pass returns the EXACT nearest neighbours, as a KD-tree on doubles does.
  const NNPlan p = make_plan(Ns, Nt, 1, true);
  isr::Workspace w(ws, ws_bytes);
  double* part_sums = w.take<double>((size_t)p.fblocks * kNV);
  IcpState* st = w.take<IcpState>(1);
  unsigned long long* packed = w.take<unsigned long long>(Ns);
  int32_t* prev_idx = w.take<int32_t>(Ns);
  int32_t* amb = w.take<int32_t>(Ns);
  float* t2_bound = w.take<float>(1);
  float* tile_box = w.take<float>((size_t)((Nt + kTile - 1) / kTile) * 6);
  const bool warm = isr::tuning(ISR_TUNE_ICP_WARM) != 0;          // tuning knob: 0 keeps every pass on the cold kernel
  icp_init_kernel<<<(Ns + kThreads - 1) / kThreads, kThreads, 0, stream>>>(st, T_io, packed, amb, Ns);
  cloud_r2max_kernel<<<1, kThreads, 0, stream>>>(tgt, Nt, t2_bound);
  // per-wave tile cull of the searches (nn_search_kernel): boxes of the target's 256-point tiles, once per call; the radius
  // caps every lane's bound (f32, rounded up: the finalize's f64 test s2 <= threshold^2 decides what counts).  Pays when the
  // clouds' rows are spatially ordered (registration.icp_point_to_point sorts them); otherwise every box is the whole cloud
  // and nothing is skipped.
#ifndef ISR_ICP_CULL
#define ISR_ICP_CULL 1
#endif
  const float cull_r2 = (float)(threshold * threshold * 1.0001 + 1e-6);
  if (ISR_ICP_CULL) tile_box_kernel<<<(Nt + kTile - 1) / kTile, 256, 0, stream>>>(tgt, Nt, tile_box);
  const dim3 grid(p.qblocks, p.nsplit, 1);
  for (int it = 0; it <= max_iter; ++it) {
    // two launches per pass: search with one packed atomic min per (point, target split), then the
    // finalize whose last workgroup runs the update
    // (from the second pass on: the warm-started filter search, bounded by the previous pass's neighbour)
    const int32_t* w_idx = (it > 0 && warm) ? prev_idx : nullptr;
    launch_search(p, grid, stream, src, Ns, tgt, Nt, T_io, nullptr, nullptr, nullptr, &st->done, nullptr, packed, w_idx, amb,
                  t2_bound, ISR_ICP_CULL ? tile_box : nullptr, cull_r2);
    icp_finalize_update_kernel<<<p.fblocks, kThreads, 0, stream>>>(src, Ns, tgt, Nt, threshold, packed, prev_idx, amb, part_sums,
                                                                   max_iter, rel_fitness, rel_rmse, T_io, st, result);
  }
  ISR_CHECK_LAUNCH("icp kernels");
  return ISR_OK;
}
