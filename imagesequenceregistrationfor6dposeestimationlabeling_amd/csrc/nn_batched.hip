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

// unresolved != nullptr: second pass behind nn_grid_search_kernel (nsplit must be 1) — only blocks
// that hold a query marked kUnresolved run, and only those queries are stored.
template <int RQ>
__global__ __launch_bounds__(kThreads) void nn_search_kernel(
    const float* __restrict__ qry, int Nq, const float* __restrict__ tgt, int Nt,
    const double* __restrict__ Tq, const double* __restrict__ Tt, int split_len, int nsplit,
    float* __restrict__ part_d2, int32_t* __restrict__ part_idx, const int32_t* __restrict__ skip,
    const int32_t* __restrict__ unresolved) {
  __shared__ __attribute__((aligned(16))) float lds[2][3][kTile];
  if (skip && *skip) return;  // device-side ICP loop: converged, later iterations are no-ops
  if (unresolved && *unresolved == 0) return;

  const int tid = threadIdx.x;
  const int b = blockIdx.z;
  const int split = blockIdx.y;
  if (unresolved) {
    int mine = 0;
#pragma unroll
    for (int r = 0; r < RQ; ++r) {
      const int qi = (blockIdx.x * RQ + r) * kThreads + tid;
      if (qi < Nq) mine |= part_idx[((size_t)b * nsplit + split) * Nq + qi] == kUnresolved;
    }
    if (!__syncthreads_or(mine)) return;
  }
  const double* tq = Tq ? Tq + 12 * (size_t)b : nullptr;
  const double* tt = Tt ? Tt + 12 * (size_t)b : nullptr;

  float qx[RQ], qy[RQ], qz[RQ], best[RQ];
  int bidx[RQ];
#pragma unroll
  for (int r = 0; r < RQ; ++r) {
    int qi = (blockIdx.x * RQ + r) * kThreads + tid;
    qi = qi < Nq ? qi : Nq - 1;  // clamp: out-of-range lanes compute a valid query, never store
    double x, y, z;
    xform64(tq, qry[3 * (size_t)qi], qry[3 * (size_t)qi + 1], qry[3 * (size_t)qi + 2], x, y, z);
    qx[r] = (float)x; qy[r] = (float)y; qz[r] = (float)z;
    best[r] = __builtin_inff();
    bidx[r] = -1;
  }

  const int t0 = split * split_len;
  const int t1 = min(Nt, t0 + split_len);
  const int ntiles = (t1 - t0 + kTile - 1) / kTile;

  auto stage = [&](int tile, int buf) {
    const int j = t0 + tile * kTile + tid;
    float x = 3.0e38f, y = 3.0e38f, z = 3.0e38f;  // padding: d2 = +inf, never wins
    if (j < t1) {
      double dx, dy, dz;
      xform64(tt, tgt[3 * (size_t)j], tgt[3 * (size_t)j + 1], tgt[3 * (size_t)j + 2], dx, dy, dz);
      x = (float)dx; y = (float)dy; z = (float)dz;
    }
    lds[buf][0][tid] = x;
    lds[buf][1][tid] = y;
    lds[buf][2][tid] = z;
  };

  if (ntiles > 0) stage(0, 0);
  __syncthreads();
  for (int tile = 0; tile < ntiles; ++tile) {
    const int buf = tile & 1;
    if (tile + 1 < ntiles) stage(tile + 1, buf ^ 1);
    const int jbase = t0 + tile * kTile;
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
        if (__any(m < best[r])) {  // wave-uniform: rare after the first few tiles
#pragma unroll
          for (int v = 0; v < kGroup; ++v) {
            const bool up = d[v] < best[r];
            best[r] = up ? d[v] : best[r];
            bidx[r] = up ? (jbase + g + v) : bidx[r];
          }
        }
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int r = 0; r < RQ; ++r) {
    const int qi = (blockIdx.x * RQ + r) * kThreads + tid;
    if (qi < Nq) {
      const size_t o = ((size_t)b * nsplit + split) * Nq + qi;
      if (unresolved && part_idx[o] != kUnresolved) continue;
      part_d2[o] = best[r];
      part_idx[o] = bidx[r];
    }
  }
}

// ------------------------------------------------------------------------------ uniform-grid search
// Exact nearest neighbour without visiting every target.  The target cloud is binned ONCE, in its
// own (untransformed) frame, into a uniform grid (counting sort by cell; x fastest, so a run of cells
// along x is one contiguous range of the sorted points).  A query is carried into that frame with
// the inverse of the batch item's target transform only to decide WHICH cells to visit; every
// candidate's distance is evaluated exactly as the brute-force kernel does (f64 transform of the
// target point, rounded to f32, d2 = fmaf chain), and the winner is the lexicographic minimum of
// (d2, index) — the result is bit-identical to scanning all Nt targets.  Cells are visited in
// growing cubes around the query's cell; every point outside the cube of half-width k cells is
// farther than k h in the grid frame, so the search stops once best <= (0.985 k h - slack): the
// factor covers target transforms that are rigid to 0.5 % (checked per item, otherwise the item
// goes to the brute-force pass) and f32 rounding of coordinates.
constexpr int kMaxCells = 1 << 20;
constexpr int kMaxRing = 8;        // beyond: the query is handed to the brute-force pass

struct GridDesc {
  double gmin[3];
  double h, inv_h;
  int nx, ny, nz, ncell;
};

__global__ __launch_bounds__(1024) void grid_bbox_kernel(const float* __restrict__ tgt, int Nt, double cell_scale,
                                                         GridDesc* __restrict__ g, int32_t* __restrict__ unresolved) {
  __shared__ float smin[3][16], smax[3][16];
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int j = threadIdx.x; j < Nt; j += blockDim.x) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = tgt[3 * (size_t)j + a];
      lo[a] = fminf(lo[a], v);
      hi[a] = fmaxf(hi[a], v);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64));
    }
    if ((threadIdx.x & 63) == 0) { smin[a][threadIdx.x >> 6] = lo[a]; smax[a][threadIdx.x >> 6] = hi[a]; }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  double ext[3], emax = 0.0;
  for (int a = 0; a < 3; ++a) {
    float l = smin[a][0], u = smax[a][0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { l = fminf(l, smin[a][w]); u = fmaxf(u, smax[a][w]); }
    g->gmin[a] = l;
    ext[a] = (double)u - (double)l;
    emax = fmax(emax, ext[a]);
  }
  // a surface sampled by Nt points has spacing ~ extent / sqrt(Nt): cell_scale^2 points per occupied cell
  double h = cell_scale * emax / sqrt((double)Nt);
  if (!(h > 0.0)) h = 1.0;
  int nx, ny, nz;
  for (;;) {
    nx = (int)(ext[0] / h) + 1; ny = (int)(ext[1] / h) + 1; nz = (int)(ext[2] / h) + 1;
    if ((double)nx * ny * nz <= (double)kMaxCells) break;
    h *= 1.26;
  }
  g->h = h; g->inv_h = 1.0 / h;
  g->nx = nx; g->ny = ny; g->nz = nz; g->ncell = nx * ny * nz;
  if (unresolved) *unresolved = 0;
}

__device__ __forceinline__ int grid_axis_cell(double v, double gmin, double inv_h, int n) {
  const double c = floor((v - gmin) * inv_h);
  return c < 0.0 ? 0 : (c > (double)(n - 1) ? n - 1 : (int)c);
}

__global__ __launch_bounds__(kThreads) void grid_count_kernel(const float* __restrict__ tgt, int Nt,
                                                              const GridDesc* __restrict__ g,
                                                              int32_t* __restrict__ cid, int32_t* __restrict__ count) {
  const int j = blockIdx.x * kThreads + threadIdx.x;
  if (j >= Nt) return;
  const int cx = grid_axis_cell(tgt[3 * (size_t)j], g->gmin[0], g->inv_h, g->nx);
  const int cy = grid_axis_cell(tgt[3 * (size_t)j + 1], g->gmin[1], g->inv_h, g->ny);
  const int cz = grid_axis_cell(tgt[3 * (size_t)j + 2], g->gmin[2], g->inv_h, g->nz);
  const int c = (cz * g->ny + cy) * g->nx + cx;
  cid[j] = c;
  atomicAdd(&count[c], 1);
}

// exclusive scan of count[0..ncell) into start[0..ncell]; one block, contiguous chunk per thread
__global__ __launch_bounds__(1024) void grid_scan_kernel(const GridDesc* __restrict__ g, const int32_t* __restrict__ count,
                                                         int32_t* __restrict__ start) {
  __shared__ int32_t part[1024];
  const int n = g->ncell;
  const int per = (n + 1023) / 1024;
  const int lo = threadIdx.x * per, hi = min(n, lo + per);
  int32_t s = 0;
  for (int i = lo; i < hi; ++i) s += count[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {   // Hillis-Steele inclusive scan of the chunk sums
    const int32_t v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int32_t run = threadIdx.x ? part[threadIdx.x - 1] : 0;
  for (int i = lo; i < hi; ++i) { start[i] = run; run += count[i]; }
  if (threadIdx.x == 1023) start[n] = part[1023];
}

__global__ __launch_bounds__(kThreads) void grid_scatter_kernel(const float* __restrict__ tgt, int Nt,
                                                                const int32_t* __restrict__ cid,
                                                                const int32_t* __restrict__ start,
                                                                int32_t* __restrict__ count, float4* __restrict__ sorted) {
  const int j = blockIdx.x * kThreads + threadIdx.x;
  if (j >= Nt) return;
  const int c = cid[j];
  const int pos = start[c] + atomicSub(&count[c], 1) - 1;   // order inside a cell is irrelevant: (d2, index) decides
  sorted[pos] = make_float4(tgt[3 * (size_t)j], tgt[3 * (size_t)j + 1], tgt[3 * (size_t)j + 2], __int_as_float(j));
}

// One thread per (batch item, query).  stop_radius >= 0: nothing beyond that distance matters to the
// caller (ICP correspondences), so the search also stops once the visited cube covers it.
__global__ __launch_bounds__(kThreads) void nn_grid_search_kernel(
    const float* __restrict__ qry, int Nq, const GridDesc* __restrict__ g, const int32_t* __restrict__ start,
    const float4* __restrict__ sorted, const double* __restrict__ Tq, const double* __restrict__ Tt,
    float stop_radius, float* __restrict__ part_d2, int32_t* __restrict__ part_idx,
    int32_t* __restrict__ unresolved, const int32_t* __restrict__ skip) {
  if (skip && *skip) return;
  const int b = blockIdx.y;
  const int qi = blockIdx.x * kThreads + threadIdx.x;
  if (qi >= Nq) return;
  const double* tq = Tq ? Tq + 12 * (size_t)b : nullptr;
  const double* tt = Tt ? Tt + 12 * (size_t)b : nullptr;
  const size_t o = (size_t)b * Nq + qi;

  double x, y, z;
  xform64(tq, qry[3 * (size_t)qi], qry[3 * (size_t)qi + 1], qry[3 * (size_t)qi + 2], x, y, z);
  const float qx = (float)x, qy = (float)y, qz = (float)z;

  // the query in the grid's frame: inverse of the target transform (adjugate; rigidity checked)
  double mx = qx, my = qy, mz = qz;
  bool rigid = true;
  if (tt) {
    const double r00 = tt[0], r01 = tt[1], r02 = tt[2], r10 = tt[4], r11 = tt[5], r12 = tt[6], r20 = tt[8],
                 r21 = tt[9], r22 = tt[10];
    const double g00 = r00 * r00 + r10 * r10 + r20 * r20, g11 = r01 * r01 + r11 * r11 + r21 * r21,
                 g22 = r02 * r02 + r12 * r12 + r22 * r22, g01 = r00 * r01 + r10 * r11 + r20 * r21,
                 g02 = r00 * r02 + r10 * r12 + r20 * r22, g12 = r01 * r02 + r11 * r12 + r21 * r22;
    const double dev = fmax(fmax(fmax(fabs(g00 - 1.0), fabs(g11 - 1.0)), fabs(g22 - 1.0)),
                            fmax(fmax(fabs(g01), fabs(g02)), fabs(g12)));
    rigid = dev < 3.0e-3;   // eigenvalues of R^T R within 1 +- 9e-3: singular values above 0.995
    const double c00 = r11 * r22 - r12 * r21, c01 = r02 * r21 - r01 * r22, c02 = r01 * r12 - r02 * r11;
    const double c10 = r12 * r20 - r10 * r22, c11 = r00 * r22 - r02 * r20, c12 = r02 * r10 - r00 * r12;
    const double c20 = r10 * r21 - r11 * r20, c21 = r01 * r20 - r00 * r21, c22 = r00 * r11 - r01 * r10;
    const double id = 1.0 / (r00 * c00 + r01 * c10 + r02 * c20);
    const double ex = qx - tt[3], ey = qy - tt[7], ez = qz - tt[11];
    mx = (c00 * ex + c01 * ey + c02 * ez) * id;
    my = (c10 * ex + c11 * ey + c12 * ez) * id;
    mz = (c20 * ex + c21 * ey + c22 * ez) * id;
  }
  if (!rigid) {
    part_idx[o] = kUnresolved;
    atomicAdd(unresolved, 1);
    return;
  }
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  const int cx = grid_axis_cell(mx, g->gmin[0], g->inv_h, nx);
  const int cy = grid_axis_cell(my, g->gmin[1], g->inv_h, ny);
  const int cz = grid_axis_cell(mz, g->gmin[2], g->inv_h, nz);
  const float h = (float)g->h;
  const float slack = 4.0e-6f * (fabsf(qx) + fabsf(qy) + fabsf(qz));

  float best = __builtin_inff();
  int bidx = -1;
  auto scan = [&](int row, int x0, int x1) {   // cells [x0, x1] of one x-row: a contiguous run
    x0 = max(x0, 0);
    x1 = min(x1, nx - 1);
    if (x0 > x1) return;
    const int s0 = start[row + x0], s1 = start[row + x1 + 1];
    for (int k = s0; k < s1; ++k) {
      const float4 p = sorted[k];
      double tx, ty, tz;
      xform64(tt, p.x, p.y, p.z, tx, ty, tz);
      const float dx = qx - (float)tx, dy = qy - (float)ty, dz = qz - (float)tz;
      const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
      const int j = __float_as_int(p.w);
      if (d2 < best || (d2 == best && j < bidx)) { best = d2; bidx = j; }
    }
  };
  bool done = false;
  for (int k = 1; k <= kMaxRing && !done; ++k) {
    for (int dz = -k; dz <= k; ++dz) {
      const int zz = cz + dz;
      if (zz < 0 || zz >= nz) continue;
      for (int dy = -k; dy <= k; ++dy) {
        const int yy = cy + dy;
        if (yy < 0 || yy >= ny) continue;
        const int row = (zz * ny + yy) * nx;
        if (k == 1 || dz == -k || dz == k || dy == -k || dy == k) {
          scan(row, cx - k, cx + k);        // a face row of the shell (k = 1: the whole 3x3x3 cube)
        } else {
          scan(row, cx - k, cx - k);        // inner rows: only the two end cells are new
          scan(row, cx + k, cx + k);
        }
      }
    }
    const float lim = 0.985f * (float)k * h - slack;
    if (lim > 0.f && best <= lim * lim) done = true;                       // nothing outside can beat it
    else if (stop_radius >= 0.f && lim > stop_radius) done = true;         // nothing outside can count
    else if (k >= nx && k >= ny && k >= nz) done = true;                   // the cube already holds every cell
  }
  if (done) {
    part_d2[o] = best;
    part_idx[o] = bidx;
  } else {
    part_idx[o] = kUnresolved;
    atomicAdd(unresolved, 1);
  }
}

// ---------------------------------------------------------------------- block-cooperative grid search
// The same exactness argument as nn_grid_search_kernel, with the brute-force kernel's arithmetic
// intensity: the QUERIES are binned too (coarse cells of ~256 points), a workgroup takes one query
// cell (<= 256 neighbouring queries, one per lane), bounds it by its axis-aligned box in the target
// grid's frame, and streams the targets of the cells around that box through LDS — ring by ring,
// every lane against every candidate, winner = lexicographic minimum of (f32 d2, index).  A target
// outside the box grown by k cells is farther than k h from every query of the workgroup, so the
// search stops once the WORST best distance of the workgroup is below 0.985 k h - slack (or the grown
// box covers the radius the caller cares about, or the whole grid).  Beyond kTileRings rings, or with
// a target transform that is not rigid, the workgroup scans every target: never worse than brute force.
constexpr int kTileRings = 10;
constexpr int kTileRuns = 1024;     // x-runs of cells per ring that fit the LDS list (else: full scan)

struct QBlock {
  int32_t start, count;             // range of the cell-sorted queries
};

__global__ __launch_bounds__(kThreads) void qblocks_count_kernel(const GridDesc* __restrict__ g,
                                                                 const int32_t* __restrict__ start,
                                                                 int32_t* __restrict__ nblk) {
  const int c = blockIdx.x * kThreads + threadIdx.x;
  if (c >= g->ncell) return;
  nblk[c] = (start[c + 1] - start[c] + kThreads - 1) / kThreads;
}

__global__ __launch_bounds__(kThreads) void qblocks_fill_kernel(const GridDesc* __restrict__ g,
                                                                const int32_t* __restrict__ start,
                                                                const int32_t* __restrict__ bstart,
                                                                QBlock* __restrict__ table) {
  const int c = blockIdx.x * kThreads + threadIdx.x;
  if (c >= g->ncell) return;
  const int s0 = start[c], cnt = start[c + 1] - s0;
  int o = bstart[c];
  for (int j = 0; j < cnt; j += kThreads) table[o++] = QBlock{s0 + j, min(kThreads, cnt - j)};
}

__device__ __forceinline__ float block_reduce_max(float v, float* red) {   // red: kThreads / 64 floats
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(kThreads, 8) void nn_tile_search_kernel(
    const float4* __restrict__ qsorted, const QBlock* __restrict__ qtable, const GridDesc* __restrict__ gq,
    const int32_t* __restrict__ qbstart, int Nq, const GridDesc* __restrict__ g, const int32_t* __restrict__ start,
    const float4* __restrict__ sorted, int Nt, const double* __restrict__ Tq, const double* __restrict__ Tt, int nb,
    float stop_radius, float* __restrict__ part_d2, int32_t* __restrict__ part_idx, const int32_t* __restrict__ skip) {
  __shared__ __attribute__((aligned(16))) float tile[3][kTile];
  __shared__ int32_t tidx[kTile];
  __shared__ int32_t run_start[kTileRuns], run_pref[kTileRuns + 1];
  __shared__ float red[kThreads / 64];
  __shared__ int32_t scan_w[kThreads / 64];
  if (skip && *skip) return;
  const int tid = threadIdx.x;
  const int nqb = qbstart[gq->ncell];                 // query blocks (device-side count)
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  const double gh = g->h, ginv = g->inv_h;
  const float h = (float)gh;

  for (long item = blockIdx.x; item < (long)nqb * nb; item += gridDim.x) {
    const int b = (int)(item / nqb);
    const QBlock qb = qtable[item % nqb];
    const double* tq = Tq ? Tq + 12 * (size_t)b : nullptr;
    const double* tt = Tt ? Tt + 12 * (size_t)b : nullptr;
    const bool valid = tid < qb.count;
    const float4 qp = qsorted[qb.start + (valid ? tid : 0)];
    const int qorig = __float_as_int(qp.w);
    double x, y, z;
    xform64(tq, qp.x, qp.y, qp.z, x, y, z);
    const float qx = (float)x, qy = (float)y, qz = (float)z;

    // the query in the target grid's frame (inverse of the target transform; rigidity checked)
    double mx = qx, my = qy, mz = qz;
    bool rigid = true;
    if (tt) {
      const double r00 = tt[0], r01 = tt[1], r02 = tt[2], r10 = tt[4], r11 = tt[5], r12 = tt[6], r20 = tt[8],
                   r21 = tt[9], r22 = tt[10];
      const double g00 = r00 * r00 + r10 * r10 + r20 * r20, g11 = r01 * r01 + r11 * r11 + r21 * r21,
                   g22 = r02 * r02 + r12 * r12 + r22 * r22, g01 = r00 * r01 + r10 * r11 + r20 * r21,
                   g02 = r00 * r02 + r10 * r12 + r20 * r22, g12 = r01 * r02 + r11 * r12 + r21 * r22;
      const double dev = fmax(fmax(fmax(fabs(g00 - 1.0), fabs(g11 - 1.0)), fabs(g22 - 1.0)),
                              fmax(fmax(fabs(g01), fabs(g02)), fabs(g12)));
      rigid = dev < 3.0e-3;
      const double c00 = r11 * r22 - r12 * r21, c01 = r02 * r21 - r01 * r22, c02 = r01 * r12 - r02 * r11;
      const double c10 = r12 * r20 - r10 * r22, c11 = r00 * r22 - r02 * r20, c12 = r02 * r10 - r00 * r12;
      const double c20 = r10 * r21 - r11 * r20, c21 = r01 * r20 - r00 * r21, c22 = r00 * r11 - r01 * r10;
      const double id = 1.0 / (r00 * c00 + r01 * c10 + r02 * c20);
      const double ex = qx - tt[3], ey = qy - tt[7], ez = qz - tt[11];
      mx = (c00 * ex + c01 * ey + c02 * ez) * id;
      my = (c10 * ex + c11 * ey + c12 * ez) * id;
      mz = (c20 * ex + c21 * ey + c22 * ez) * id;
    }
    // the workgroup's box in cell coordinates (floor), and the coordinate magnitude for the slack
    const float big = 3.0e38f;
    const float fx = (float)floor((mx - g->gmin[0]) * ginv), fy = (float)floor((my - g->gmin[1]) * ginv),
                fz = (float)floor((mz - g->gmin[2]) * ginv);
    const float bx1 = block_reduce_max(valid ? fx : -big, red), bx0 = -block_reduce_max(valid ? -fx : -big, red);
    const float by1 = block_reduce_max(valid ? fy : -big, red), by0 = -block_reduce_max(valid ? -fy : -big, red);
    const float bz1 = block_reduce_max(valid ? fz : -big, red), bz0 = -block_reduce_max(valid ? -fz : -big, red);
    const float mag = block_reduce_max(valid ? fabsf(qx) + fabsf(qy) + fabsf(qz) : 0.f, red);
    const float slack = 4.0e-6f * mag;
    auto clampi = [](float v, int n) { return v < 0.f ? 0 : (v > (float)(n - 1) ? n - 1 : (int)v); };

    float best = __builtin_inff();
    int bidx = -1;
    // stream the candidates listed in run_start / run_pref (n_runs runs, C points) against the lanes
    auto scan_runs = [&](int n_runs, int C) {
      for (int t0 = 0; t0 < C; t0 += kTile) {
        const int p = t0 + tid;
        float cx = big, cy = big, cz = big;                  // padding: d2 = +inf
        int ci = 0x7fffffff;
        if (p < C) {
          int lo = 0, hi = n_runs;                           // last run with run_pref[r] <= p
          while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (run_pref[mid] <= p) lo = mid; else hi = mid;
          }
          const float4 c = sorted[run_start[lo] + (p - run_pref[lo])];
          double tx, ty, tz;
          xform64(tt, c.x, c.y, c.z, tx, ty, tz);
          cx = (float)tx; cy = (float)ty; cz = (float)tz;
          ci = __float_as_int(c.w);
        }
        __syncthreads();                                     // previous tile fully consumed
        tile[0][tid] = cx; tile[1][tid] = cy; tile[2][tid] = cz; tidx[tid] = ci;
        __syncthreads();
        const int nvalid = min(kTile, C - t0);
        for (int g0 = 0; g0 < nvalid; g0 += kGroup) {
          float tx[kGroup], ty[kGroup], tz[kGroup];
#pragma unroll
          for (int v = 0; v < kGroup; v += 4) {
            const float4 a = *reinterpret_cast<const float4*>(&tile[0][g0 + v]);
            const float4 c = *reinterpret_cast<const float4*>(&tile[1][g0 + v]);
            const float4 e = *reinterpret_cast<const float4*>(&tile[2][g0 + v]);
            tx[v] = a.x; tx[v + 1] = a.y; tx[v + 2] = a.z; tx[v + 3] = a.w;
            ty[v] = c.x; ty[v + 1] = c.y; ty[v + 2] = c.z; ty[v + 3] = c.w;
            tz[v] = e.x; tz[v + 1] = e.y; tz[v + 2] = e.z; tz[v + 3] = e.w;
          }
          float d[kGroup];
#pragma unroll
          for (int v = 0; v < kGroup; ++v) {
            const float dx = qx - tx[v], dy = qy - ty[v], dz = qz - tz[v];
            d[v] = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
          }
          const float m = fminf(fminf(fminf(d[3], d[4]), d[5]),
                                fminf(fminf(d[6], d[7]), fminf(fminf(d[0], d[1]), d[2])));
          if (__any(m <= best)) {                            // <=: an equal distance with a lower index wins
#pragma unroll
            for (int v = 0; v < kGroup; ++v) {
              const int j = tidx[g0 + v];
              const bool up = (d[v] < best) | ((d[v] == best) & (j < bidx));   // bitwise: no branches
              best = up ? d[v] : best;
              bidx = up ? j : bidx;
            }
          }
        }
      }
      __syncthreads();
    };

    // Box growth: scan the workgroup's own box grown by one cell (the box alone can never be final:
    // its margin is zero); then jump straight to the box whose margin k h covers the worst
    // best-distance found so far (one more shell and every lane is final); while some lane has seen
    // no target at all the margin doubles.  A shell that would list more than half of the cloud, or
    // does not fit the run list, is replaced by the full scan.
    bool done = false;
    if (rigid) {
      int px0 = 0, px1 = -1, py0 = 0, py1 = -1, pz0 = 0, pz1 = -1;     // previous (inner) cell box: empty
      int k = 1;
      const int kcap = stop_radius >= 0.f ? (int)ceilf((stop_radius + slack) / (0.985f * h)) + 1 : 0x3fffffff;
      for (int iter = 0; iter < 8 && !done; ++iter) {
        const int x0 = clampi(bx0 - k, nx), x1 = clampi(bx1 + k, nx), y0 = clampi(by0 - k, ny),
                  y1 = clampi(by1 + k, ny), z0 = clampi(bz0 - k, nz), z1 = clampi(bz1 + k, nz);
        const bool grew = iter == 0 || x0 != px0 || x1 != px1 || y0 != py0 || y1 != py1 || z0 != pz0 || z1 != pz1;
        const int rows = (y1 - y0 + 1) * (z1 - z0 + 1);
        if (2 * rows > kTileRuns) break;                         // too wide for the run list: full scan below
        if (grew) {
          // run list: row (z, y) contributes [x0, x1], or its two new end pieces when the row was inside
          for (int i = tid; i < rows; i += kThreads) {
            const int zz = z0 + i / (y1 - y0 + 1), yy = y0 + i % (y1 - y0 + 1);
            const int row = (zz * ny + yy) * nx;
            const bool inner = px1 >= px0 && zz >= pz0 && zz <= pz1 && yy >= py0 && yy <= py1;
            int a0 = x0, a1 = x1, c0 = 0, c1 = -1;
            if (inner) { a1 = px0 - 1; c0 = px1 + 1; c1 = x1; }
            const int s0 = a1 >= a0 ? start[row + a0] : 0, e0 = a1 >= a0 ? start[row + a1 + 1] : 0;
            const int s1 = c1 >= c0 ? start[row + c0] : 0, e1 = c1 >= c0 ? start[row + c1 + 1] : 0;
            run_start[2 * i] = s0; run_pref[2 * i] = e0 - s0;
            run_start[2 * i + 1] = s1; run_pref[2 * i + 1] = e1 - s1;
          }
          __syncthreads();
          // exclusive scan of the run lengths (<= kTileRuns entries, 4 per thread)
          const int n_runs = 2 * rows;
          int loc[kTileRuns / kThreads], sum = 0;
#pragma unroll
          for (int j = 0; j < kTileRuns / kThreads; ++j) {
            const int e = tid * (kTileRuns / kThreads) + j;
            loc[j] = e < n_runs ? run_pref[e] : 0;
            sum += loc[j];
          }
          int inc = sum;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(inc, o, 64);
            if ((tid & 63) >= o) inc += u;
          }
          if ((tid & 63) == 63) scan_w[tid >> 6] = inc;
          __syncthreads();
          int base = 0, total = 0;
#pragma unroll
          for (int w = 0; w < kThreads / 64; ++w) {
            if (w < (tid >> 6)) base += scan_w[w];
            total += scan_w[w];
          }
          if (2 * total > Nt) break;                             // block-uniform: cheaper to scan everything once
          int run = base + inc - sum;
#pragma unroll
          for (int j = 0; j < kTileRuns / kThreads; ++j) {
            const int e = tid * (kTileRuns / kThreads) + j;
            if (e < n_runs) run_pref[e] = run;
            run += loc[j];
          }
          if (tid == 0) run_pref[n_runs] = total;
          __syncthreads();
          scan_runs(n_runs, total);
        }
        px0 = x0; px1 = x1; py0 = y0; py1 = y1; pz0 = z0; pz1 = z1;
        const float worst = block_reduce_max(valid ? best : 0.f, red);
        const float lim = 0.985f * (float)k * h - slack;
        if (lim > 0.f && worst <= lim * lim) done = true;                        // every lane's winner is final
        else if (k >= kcap) done = true;                                         // nothing outside can count
        else if (x0 == 0 && y0 == 0 && z0 == 0 && x1 == nx - 1 && y1 == ny - 1 && z1 == nz - 1) done = true;
        else {
          int kn = k ? 2 * k : 1;
          if (worst < 3.0e38f) kn = max(k + 1, (int)ceilf((sqrtf(worst) * 1.0001f + slack) / (0.985f * h)));
          k = min(kn, kcap);
        }
      }
    }
    if (!done) {                                                 // every target: exact whatever the geometry
      if (tid == 0) { run_start[0] = 0; run_pref[0] = 0; run_pref[1] = Nt; }
      __syncthreads();
      scan_runs(1, Nt);
    }
    if (valid) {
      const size_t o = (size_t)b * Nq + qorig;
      part_d2[o] = best;
      part_idx[o] = bidx;
    }
    __syncthreads();
  }
}

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
    const int32_t* __restrict__ skip) {
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
    for (int s = 0; s < nsplit; ++s) {
      const size_t o = ((size_t)bl * nsplit + s) * Nq + qi;
      const float d2 = part_d2[o];
      if (d2 < best) { best = d2; bi = part_idx[o]; }
    }
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
};

// Cyclic Jacobi on a symmetric 4x4; every loop has constant bounds and is unrolled so A and V
// stay in registers (indexed dynamically they live in scratch memory: 60 us per call instead of 10).
__device__ void jacobi4_largest(double A[4][4], double q[4]) {
  double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  for (int sweep = 0; sweep < 16; ++sweep) {
    double off = 0.0;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int r = p + 1; r < 4; ++r) off += A[p][r] * A[p][r];
    if (off < 1e-300) break;
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

__global__ __launch_bounds__(kNV * kIcpLanes) void icp_update_kernel(const double* __restrict__ part_sums, int nblk, int Ns, int max_iter,
                                  double rel_fitness, double rel_rmse, double* __restrict__ T,
                                  IcpState* __restrict__ st, double* __restrict__ result) {
  // launched with kIcpLanes lanes per sum: lane l of sum k adds blocks l, l + kIcpLanes, ... and the
  // kIcpLanes partial sums are added in lane order (fixed shape: run-to-run reproducible)
  __shared__ double part[kNV][kIcpLanes];
  __shared__ double v[kNV];
  if (st->done) return;
  {
    const int k = threadIdx.x / kIcpLanes, l = threadIdx.x % kIcpLanes;
    if (k < kNV) {
      double s = 0.0;
      for (int i = l; i < nblk; i += kIcpLanes) s += part_sums[(size_t)i * kNV + k];
      part[k][l] = s;
    }
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

__global__ void icp_init_kernel(IcpState* st, double* T) {
  st->prev_fit = 0; st->prev_rmse = 0; st->iter = 0; st->done = 0;
  T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
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

constexpr double kCellScaleTgt = 6.0;    // cooperative search: ~36 targets per occupied cell
constexpr double kCellScaleQry = 16.0;   // ~256 queries per occupied cell = one workgroup
constexpr int kTileGrid = 2048;          // persistent workgroups of nn_tile_search_kernel

// device-side pieces of a grid built in the caller's workspace
struct GridWs {
  GridDesc* desc;
  int32_t* unresolved;
  int32_t* count;    // kMaxCells + 1
  int32_t* start;    // kMaxCells + 1
  int32_t* cid;      // Nt
  float4* sorted;    // Nt
};

size_t grid_ws_bytes(int Nt) {
  return isr::align_up(sizeof(GridDesc) + 64, 256) + 2 * isr::align_up((size_t)(kMaxCells + 1) * 4, 256) +
         isr::align_up((size_t)Nt * 4, 256) + isr::align_up((size_t)Nt * 16, 256);
}

// cooperative search: target grid + query grid + the query-block table
struct TileWs {
  GridWs t, q;
  int32_t* bstart;   // kMaxCells + 1: first block of every query cell; [ncell] = number of blocks
  QBlock* table;     // <= Nq entries
};

size_t tile_ws_bytes(int Nq, int Nt) {
  return grid_ws_bytes(Nt) + grid_ws_bytes(Nq) + isr::align_up((size_t)(kMaxCells + 1) * 4, 256) +
         isr::align_up((size_t)(Nq + 1) * sizeof(QBlock), 256);
}

GridWs take_grid(isr::Workspace& w, int Nt) {
  GridWs g;
  char* head = w.take<char>(isr::align_up(sizeof(GridDesc) + 64, 256));
  g.desc = reinterpret_cast<GridDesc*>(head);
  g.unresolved = reinterpret_cast<int32_t*>(head + isr::align_up(sizeof(GridDesc), 16));
  g.count = w.take<int32_t>(kMaxCells + 1);
  g.start = w.take<int32_t>(kMaxCells + 1);
  g.cid = w.take<int32_t>(Nt);
  g.sorted = w.take<float4>(Nt);
  return g;
}

// bin an (untransformed) cloud: 1 memset + 4 small kernels, all on `stream`.  cell_scale: cell edge in
// units of the surface sampling distance extent / sqrt(N)
void build_grid(const GridWs& g, const float* tgt, int Nt, hipStream_t stream, double cell_scale = 2.5) {
  (void)hipMemsetAsync(g.count, 0, (size_t)(kMaxCells + 1) * 4, stream);
  grid_bbox_kernel<<<1, 1024, 0, stream>>>(tgt, Nt, cell_scale, g.desc, g.unresolved);
  const int nb = (Nt + kThreads - 1) / kThreads;
  grid_count_kernel<<<nb, kThreads, 0, stream>>>(tgt, Nt, g.desc, g.cid, g.count);
  grid_scan_kernel<<<1, 1024, 0, stream>>>(g.desc, g.count, g.start);
  grid_scatter_kernel<<<nb, kThreads, 0, stream>>>(tgt, Nt, g.cid, g.start, g.count, g.sorted);
}

TileWs take_tile(isr::Workspace& w, int Nq, int Nt) {
  TileWs tw;
  tw.t = take_grid(w, Nt);
  tw.q = take_grid(w, Nq);
  tw.bstart = w.take<int32_t>(kMaxCells + 1);
  tw.table = w.take<QBlock>(Nq + 1);
  return tw;
}

// both grids and the query-block table; everything stays on the device
void build_tile(const TileWs& tw, const float* qry, int Nq, const float* tgt, int Nt, hipStream_t stream) {
  double st = kCellScaleTgt, sq = kCellScaleQry;
  if (const char* e = getenv("ISR_NN_TILE")) (void)sscanf(e, "%lf,%lf", &st, &sq);   // tuning hook (experiments only)
  build_grid(tw.t, tgt, Nt, stream, st);
  build_grid(tw.q, qry, Nq, stream, sq);
  const int cb = (kMaxCells + kThreads - 1) / kThreads;      // ncell is a device value: cover the maximum
  (void)hipMemsetAsync(tw.q.count, 0, (size_t)(kMaxCells + 1) * 4, stream);
  qblocks_count_kernel<<<cb, kThreads, 0, stream>>>(tw.q.desc, tw.q.start, tw.q.count);
  grid_scan_kernel<<<1, 1024, 0, stream>>>(tw.q.desc, tw.q.count, tw.bstart);
  qblocks_fill_kernel<<<cb, kThreads, 0, stream>>>(tw.q.desc, tw.q.start, tw.bstart, tw.table);
}

void launch_tile_search(const TileWs& tw, int Nq, int Nt, const double* tq, const double* tt, int nb, float stop_radius,
                        float* part_d2, int32_t* part_idx, const int32_t* skip, hipStream_t stream) {
  nn_tile_search_kernel<<<kTileGrid, kThreads, 0, stream>>>(tw.q.sorted, tw.table, tw.q.desc, tw.bstart, Nq, tw.t.desc,
                                                            tw.t.start, tw.t.sorted, Nt, tq, tt, nb, stop_radius,
                                                            part_d2, part_idx, skip);
}

constexpr size_t kPartBudget = size_t(192) << 20;  // bytes of per-query partials per chunk

NNPlan make_plan(int Nq, int Nt, int B) {
  NNPlan p;
  p.rq = (Nq >= 4 * kThreads) ? 4 : 1;
  // tuning hook (experiments only): ISR_NN_PLAN="rq,want_blocks"
  static const char* env = getenv("ISR_NN_PLAN");
  long want_env = 0;
  if (env) { int rq = 0; if (sscanf(env, "%d,%ld", &rq, &want_env) >= 1 && (rq == 1 || rq == 4)) p.rq = rq; }
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
  // Three exact searches (tests compare them bit for bit; ISR_NN_GRID=0/1/2 forces one):
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
  if (const char* ge = getenv("ISR_NN_GRID")) {
    if (ge[0]) { p.grid = ge[0] == '1'; p.tile = ge[0] == '2'; }   // empty = unset
  }
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
    if (p.tile)
      launch_tile_search(tw, Nq, Nt, tq, tt, nb, stop_radius, part_d2, part_idx, nullptr, stream);
    else if (p.rq == 4)
      nn_search_kernel<4><<<grid, kThreads, 0, stream>>>(qry, Nq, tgt, Nt, tq, tt, p.split_len,
                                                         p.nsplit, part_d2, part_idx, nullptr, unres);
    else
      nn_search_kernel<1><<<grid, kThreads, 0, stream>>>(qry, Nq, tgt, Nt, tq, tt, p.split_len,
                                                         p.nsplit, part_d2, part_idx, nullptr, unres);
    ISR_CHECK_LAUNCH("nn_search_kernel");
    const dim3 fgrid(p.fblocks, nb);
    if (cov)
      nn_finalize_kernel<true><<<fgrid, kThreads, 0, stream>>>(qry, Nq, tgt, Tq, Tt, p.nsplit,
                                                               radius, part_d2, part_idx, b0,
                                                               nn_idx, nn_d, part_sums, nullptr);
    else
      nn_finalize_kernel<false><<<fgrid, kThreads, 0, stream>>>(qry, Nq, tgt, Tq, Tt, p.nsplit,
                                                                radius, part_d2, part_idx, b0,
                                                                nn_idx, nn_d, part_sums, nullptr);
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
  return isr_nn_batched_workspace_bytes(Ns, Nt, 1) + 1024;   // includes the grid when the plan uses one
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
  const NNPlan p = make_plan(Ns, Nt, 1);
  isr::Workspace w(ws, ws_bytes);
  float* part_d2 = w.take<float>((size_t)p.nsplit * Ns);
  int32_t* part_idx = w.take<int32_t>((size_t)p.nsplit * Ns);
  double* part_sums = w.take<double>((size_t)p.fblocks * kNV);
  IcpState* st = w.take<IcpState>(1);
  icp_init_kernel<<<1, 1, 0, stream>>>(st, T_io);
  GridWs gw{};
  TileWs tw{};
  if (p.grid) {           // the target never moves: one grid serves every iteration
    gw = take_grid(w, Nt);
    build_grid(gw, tgt, Nt, stream);
  }
  if (p.tile) {           // neither does the source in its own frame
    tw = take_tile(w, Ns, Nt);
    build_tile(tw, src, Ns, tgt, Nt, stream);
  }
  const float stop_radius = (float)(threshold * (1.0 + 1e-6));
  const dim3 grid(p.qblocks, p.nsplit, 1), fgrid(p.fblocks, 1);
  for (int it = 0; it <= max_iter; ++it) {
    if (p.tile)
      launch_tile_search(tw, Ns, Nt, T_io, nullptr, 1, stop_radius, part_d2, part_idx, &st->done, stream);
    else if (p.grid)      // with a stop radius every query resolves: no brute-force pass
      nn_grid_search_kernel<<<dim3(p.fblocks, 1), kThreads, 0, stream>>>(src, Ns, gw.desc, gw.start, gw.sorted, T_io,
                                                                         nullptr, stop_radius, part_d2, part_idx,
                                                                         gw.unresolved, &st->done);
    else if (p.rq == 4)
      nn_search_kernel<4><<<grid, kThreads, 0, stream>>>(src, Ns, tgt, Nt, T_io, nullptr, p.split_len, p.nsplit,
                                                         part_d2, part_idx, &st->done, nullptr);
    else
      nn_search_kernel<1><<<grid, kThreads, 0, stream>>>(src, Ns, tgt, Nt, T_io, nullptr, p.split_len, p.nsplit,
                                                         part_d2, part_idx, &st->done, nullptr);
    nn_finalize_kernel<true><<<fgrid, kThreads, 0, stream>>>(src, Ns, tgt, T_io, nullptr, p.nsplit, threshold, part_d2,
                                                             part_idx, 0, nullptr, nullptr, part_sums, &st->done);
    icp_update_kernel<<<1, kNV * kIcpLanes, 0, stream>>>(part_sums, p.fblocks, Ns, max_iter, rel_fitness, rel_rmse, T_io, st, result);
  }
  ISR_CHECK_LAUNCH("icp kernels");
  return ISR_OK;
}
