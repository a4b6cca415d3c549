// nn_grid.hpp — the two exact grid searches of K3 (per-lane and block-cooperative), the counting-sort
// grid build and the workspace helpers.  Included by nn_batched.hip inside its anonymous namespace,
// after kThreads / kTile / kGroup / kUnresolved and xform64 are defined: the searches evaluate every
// candidate with the brute-force kernel's arithmetic (nn_search_kernel), which is what makes their
// results bit-identical to it.
#pragma once

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
// intensity: the QUERIES are binned too (coarse cells of ~170 points), a workgroup takes up to 64
// neighbouring queries of one cell (one per lane), bounds it by its axis-aligned box in the target
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
                                                                 const int32_t* __restrict__ start, int tb,
                                                                 int32_t* __restrict__ nblk) {
  const int c = blockIdx.x * kThreads + threadIdx.x;
  if (c >= g->ncell) return;
  nblk[c] = (start[c + 1] - start[c] + tb - 1) / tb;     // tb = queries per search workgroup
}

__global__ __launch_bounds__(kThreads) void qblocks_fill_kernel(const GridDesc* __restrict__ g,
                                                                const int32_t* __restrict__ start,
                                                                const int32_t* __restrict__ bstart, int tb,
                                                                QBlock* __restrict__ table) {
  const int c = blockIdx.x * kThreads + threadIdx.x;
  if (c >= g->ncell) return;
  const int s0 = start[c], cnt = start[c + 1] - s0;
  int o = bstart[c];
  for (int j = 0; j < cnt; j += tb) table[o++] = QBlock{s0 + j, min(tb, cnt - j)};
}

template <int TB>
__device__ __forceinline__ float block_reduce_max(float v, float* red) {   // red: TB / 64 floats
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  if (TB == 64) return v;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float m = red[0];
#pragma unroll
  for (int w = 1; w < TB / 64; ++w) m = fmaxf(m, red[w]);
  return m;
}

// (no occupancy promise in the launch bounds: the compiler reports 3 waves per SIMD at TB = 256 and 5 at TB = 64 — the tile arrays
// and the per-lane candidate state — and a second argument it cannot meet only earns a warning)
template <int TB>
__global__ __launch_bounds__(TB) void nn_tile_search_kernel(
    const float4* __restrict__ qsorted, const QBlock* __restrict__ qtable, const GridDesc* __restrict__ gq,
    const int32_t* __restrict__ qbstart, int Nq, const GridDesc* __restrict__ g, const int32_t* __restrict__ start,
    const float4* __restrict__ sorted, int Nt, const double* __restrict__ Tq, const double* __restrict__ Tt, int nb,
    float stop_radius, float* __restrict__ part_d2, int32_t* __restrict__ part_idx, const int32_t* __restrict__ skip) {
  __shared__ __attribute__((aligned(16))) float tile[3][kTile];
  __shared__ int32_t tidx[kTile];
  __shared__ int32_t run_start[kTileRuns], run_pref[kTileRuns + 1];
  __shared__ float red[TB / 64];
  __shared__ int32_t scan_w[TB / 64];
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
    const float bx1 = block_reduce_max<TB>(valid ? fx : -big, red), bx0 = -block_reduce_max<TB>(valid ? -fx : -big, red);
    const float by1 = block_reduce_max<TB>(valid ? fy : -big, red), by0 = -block_reduce_max<TB>(valid ? -fy : -big, red);
    const float bz1 = block_reduce_max<TB>(valid ? fz : -big, red), bz0 = -block_reduce_max<TB>(valid ? -fz : -big, red);
    const float mag = block_reduce_max<TB>(valid ? fabsf(qx) + fabsf(qy) + fabsf(qz) : 0.f, red);
    const float slack = 4.0e-6f * mag;
    auto clampi = [](float v, int n) { return v < 0.f ? 0 : (v > (float)(n - 1) ? n - 1 : (int)v); };

    float best = __builtin_inff();
    int bidx = -1;
    // stream the candidates listed in run_start / run_pref (n_runs runs, C points) against the lanes
    auto scan_runs = [&](int n_runs, int C) {
      constexpr int CPT = kTile / TB;                        // candidates staged per thread and tile
      // the NEXT tile's candidates are fetched (run lookup + one global load each) before the current tile is
      // scanned and transformed after it: their latency hides behind the scan instead of in front of it
      float4 raw[CPT];
      auto issue = [&](int t0) {
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
          const int p = t0 + tid + u * TB;
          raw[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (p < C) {
            int lo = 0, hi = n_runs;                         // last run with run_pref[r] <= p
            while (hi - lo > 1) {
              const int mid = (lo + hi) >> 1;
              if (run_pref[mid] <= p) lo = mid; else hi = mid;
            }
            raw[u] = sorted[run_start[lo] + (p - run_pref[lo])];
          }
        }
      };
      issue(0);
      for (int t0 = 0; t0 < C; t0 += kTile) {
        float cx[CPT], cy[CPT], cz[CPT];
        int ci[CPT];
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
          const int p = t0 + tid + u * TB;
          cx[u] = big; cy[u] = big; cz[u] = big;             // padding: d2 = +inf
          ci[u] = 0x7fffffff;
          if (p < C) {
            double tx, ty, tz;
            xform64(tt, raw[u].x, raw[u].y, raw[u].z, tx, ty, tz);
            cx[u] = (float)tx; cy[u] = (float)ty; cz[u] = (float)tz;
            ci[u] = __float_as_int(raw[u].w);
          }
        }
        __syncthreads();                                     // previous tile fully consumed
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
          const int q = tid + u * TB;
          tile[0][q] = cx[u]; tile[1][q] = cy[u]; tile[2][q] = cz[u]; tidx[q] = ci[u];
        }
        __syncthreads();
        if (t0 + kTile < C) issue(t0 + kTile);
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
          for (int i = tid; i < rows; i += TB) {
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
          int loc[kTileRuns / TB], sum = 0;
#pragma unroll
          for (int j = 0; j < kTileRuns / TB; ++j) {
            const int e = tid * (kTileRuns / TB) + j;
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
          for (int w = 0; w < TB / 64; ++w) {
            if (w < (tid >> 6)) base += scan_w[w];
            total += scan_w[w];
          }
          if (2 * total > Nt) break;                             // block-uniform: cheaper to scan everything once
          int run = base + inc - sum;
#pragma unroll
          for (int j = 0; j < kTileRuns / TB; ++j) {
            const int e = tid * (kTileRuns / TB) + j;
            if (e < n_runs) run_pref[e] = run;
            run += loc[j];
          }
          if (tid == 0) run_pref[n_runs] = total;
          __syncthreads();
          scan_runs(n_runs, total);
        }
        px0 = x0; px1 = x1; py0 = y0; py1 = y1; pz0 = z0; pz1 = z1;
        const float worst = block_reduce_max<TB>(valid ? best : 0.f, red);
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

constexpr double kCellScaleTgt = 6.0;    // cooperative search: ~36 targets per occupied cell
constexpr double kCellScaleQry = 13.0;   // ~170 queries per occupied cell, searched by workgroups of kTileTB
constexpr int kTileTB = 64;              // one wave per workgroup: tighter boxes, milder worst lane, no block
                                         // barriers (tools/nn_tile_sweep.py: 25-30 % faster than 256 x 16.0)
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
// queries per search workgroup and the two cell scales; ISR_NN_TILE="target scale,query scale,threads"
// is a tuning hook for experiments (threads in {64, 128, 256})
struct TileCfg {
  double st, sq;
  int tb;
};

TileCfg tile_cfg() {
  TileCfg c{kCellScaleTgt, kCellScaleQry, kTileTB};
  if (const int v = isr::tuning(ISR_TUNE_NN_TILE_ST); v > 0) c.st = v * 1e-3;
  if (const int v = isr::tuning(ISR_TUNE_NN_TILE_SQ); v > 0) c.sq = v * 1e-3;
  if (const int v = isr::tuning(ISR_TUNE_NN_TILE_TB); v > 0) c.tb = v;
  if (c.tb != 64 && c.tb != 128 && c.tb != 256) c.tb = kTileTB;
  return c;
}

void build_tile(const TileWs& tw, const float* qry, int Nq, const float* tgt, int Nt, hipStream_t stream) {
  const TileCfg cfg = tile_cfg();
  const double st = cfg.st, sq = cfg.sq;
  const int tb = cfg.tb;
  build_grid(tw.t, tgt, Nt, stream, st);
  build_grid(tw.q, qry, Nq, stream, sq);
  const int cb = (kMaxCells + kThreads - 1) / kThreads;      // ncell is a device value: cover the maximum
  (void)hipMemsetAsync(tw.q.count, 0, (size_t)(kMaxCells + 1) * 4, stream);
  qblocks_count_kernel<<<cb, kThreads, 0, stream>>>(tw.q.desc, tw.q.start, tb, tw.q.count);
  grid_scan_kernel<<<1, 1024, 0, stream>>>(tw.q.desc, tw.q.count, tw.bstart);
  qblocks_fill_kernel<<<cb, kThreads, 0, stream>>>(tw.q.desc, tw.q.start, tw.bstart, tb, tw.table);
}

void launch_tile_search(const TileWs& tw, int Nq, int Nt, const double* tq, const double* tt, int nb, float stop_radius,
                        float* part_d2, int32_t* part_idx, const int32_t* skip, hipStream_t stream) {
  const int tb = tile_cfg().tb;
#define ISR_TILE_LAUNCH(TBv)                                                                                      \
  nn_tile_search_kernel<TBv><<<kTileGrid * (kThreads / TBv), TBv, 0, stream>>>(                                   \
      tw.q.sorted, tw.table, tw.q.desc, tw.bstart, Nq, tw.t.desc, tw.t.start, tw.t.sorted, Nt, tq, tt, nb,        \
      stop_radius, part_d2, part_idx, skip)
  if (tb == 64) ISR_TILE_LAUNCH(64);
  else if (tb == 128) ISR_TILE_LAUNCH(128);
  else ISR_TILE_LAUNCH(256);
#undef ISR_TILE_LAUNCH
}

