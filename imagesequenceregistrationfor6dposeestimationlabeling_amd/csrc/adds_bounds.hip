// adds_bounds.hip — rigorous lower / upper bounds of ADD-S for a batch of pose pairs, from a distance field of the
// (static) surface cloud: the n x n vote of choosePose.py:121-145 needs only the DECISION  ADDS(...) < 0.1 * diameter.
//
//   ADDS(verts, gtR, gtT, R, T) = mean_v  min_s | (gtR v + gtT) - (R s + T) |        (inference.py:118-120, choosePose.py:20-22)
//                               = mean_v  dist(E v, S),   E = [R|T]^-1 [gtR|gtT]      (a rigid motion keeps distances)
//
// dist(., S) is 1-Lipschitz, so with F[c] = the exact distance from the centre of cell c of a uniform grid (edge h) to S,
// every x has  F[c] - |x - centre(c)| <= dist(x, S) <= F[c] + |x - centre(c)|  for ANY cell c: the kernel takes the cell that
// holds x (|x - centre| <= h sqrt(3) / 2, 0.48 h on average), or the nearest cell of the grid when a far-off pose carries x
// outside it (the bracket is then wider, and the distance to the surface's bounding box is a second lower bound).  One gather
// per vertex bounds an item's ADD-S from both sides; an item whose bounds straddle the threshold is evaluated exactly by
// isr_nn_batched, everything else is decided here — the same booleans as evaluating every item (tests compare the error
// matrices), at ~40 VALU operations and one 4-byte gather per vertex instead of a nearest-neighbour search.
//
// Tt need not be exactly orthonormal (the reference loads f32-born poses from pred_R.npy): with A its 3 x 3 part and
// eta >= ||A^T A - I||_2,  | |A^T w - s| - |w - A s| | <= eta (|s| + |w - A s|)  for every surface point s, so each vertex's
// bracket is widened by eta (max|s| + its upper bound) — 1e-5 mm at eta = 1e-7 and object scale, nothing for f64 rotations;
// a Tt that is far from rigid only makes the brackets useless (everything is searched), never wrong.
//
// The field is built once per surface cloud by the caller (cell centres through isr_nn_batched: exact distances).
#include "isr_common.hpp"

namespace {

constexpr int kBThreads = 256;

__global__ __launch_bounds__(kBThreads) void adds_bounds_kernel(const float* __restrict__ verts, int V, const double* __restrict__ Tq,
                                                                const double* __restrict__ Tt, const float* __restrict__ field,
                                                                double g0, double g1, double g2, double inv_h, double h, int nx, int ny,
                                                                int nz, float b0, float b1, float b2, float b3, float b4, float b5,
                                                                double* __restrict__ lb_sum, double* __restrict__ ub_sum) {
  __shared__ float E[12];
  __shared__ double eta_s;
  __shared__ double red[2][kBThreads / 64];
  const int b = blockIdx.x;
  if (threadIdx.x == 0) {
    // E = Tt^-1 Tq in f64 (Tt rigid: R^T, -R^T t), handed to the lanes as f32: the vertices live at object scale
    const double* q = Tq + 12 * (size_t)b;
    double R[9], t[3];
    eta_s = 0.0;
    if (Tt) {
      const double* p = Tt + 12 * (size_t)b;
      double g = 0.0;                                   // max |(A^T A - I)_ij|; ||.||_2 <= ||.||_F <= 3 max
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
          g = fmax(g, fabs(p[i] * p[j] + p[4 + i] * p[4 + j] + p[8 + i] * p[8 + j] - (i == j ? 1.0 : 0.0)));
      eta_s = 3.0 * g;
      for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) R[3 * i + j] = p[i] * q[j] + p[4 + i] * q[4 + j] + p[8 + i] * q[8 + j];
        t[i] = p[i] * (q[3] - p[3]) + p[4 + i] * (q[7] - p[7]) + p[8 + i] * (q[11] - p[11]);
      }
    } else {
      for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) R[3 * i + j] = q[4 * i + j]; t[i] = q[4 * i + 3]; }
    }
    for (int i = 0; i < 3; ++i) { E[4 * i] = (float)R[3 * i]; E[4 * i + 1] = (float)R[3 * i + 1]; E[4 * i + 2] = (float)R[3 * i + 2]; E[4 * i + 3] = (float)t[i]; }
  }
  __syncthreads();
  double lb = 0.0, ub = 0.0;
  for (int v = threadIdx.x; v < V; v += kBThreads) {
    const float X = verts[3 * (size_t)v], Y = verts[3 * (size_t)v + 1], Z = verts[3 * (size_t)v + 2];
    const float x = __builtin_fmaf(E[2], Z, __builtin_fmaf(E[1], Y, __builtin_fmaf(E[0], X, E[3])));
    const float y = __builtin_fmaf(E[6], Z, __builtin_fmaf(E[5], Y, __builtin_fmaf(E[4], X, E[7])));
    const float z = __builtin_fmaf(E[10], Z, __builtin_fmaf(E[9], Y, __builtin_fmaf(E[8], X, E[11])));
    // the cell that holds x, or the nearest cell of the grid when x lies outside it
    const double fx = ((double)x - g0) * inv_h, fy = ((double)y - g1) * inv_h, fz = ((double)z - g2) * inv_h;
    const double cx = fmin(fmax(floor(fx), 0.0), (double)(nx - 1)), cy = fmin(fmax(floor(fy), 0.0), (double)(ny - 1)),
                 cz = fmin(fmax(floor(fz), 0.0), (double)(nz - 1));
    const float d = field[((size_t)(int)cz * ny + (int)cy) * nx + (int)cx];
    // r >= |x - centre|: the f32 centre the field was evaluated at sits within 2e-5 of the f64 one at object scale
    const float ux = (float)((fx - cx - 0.5) * h), uy = (float)((fy - cy - 0.5) * h), uz = (float)((fz - cz - 0.5) * h);
    const float r = sqrtf(ux * ux + uy * uy + uz * uz) * 1.000001f + 2e-5f;
    const float ex = fmaxf(fmaxf(b0 - x, x - b3), 0.f), ey = fmaxf(fmaxf(b1 - y, y - b4), 0.f), ez = fmaxf(fmaxf(b2 - z, z - b5), 0.f);
    const float box = sqrtf(ex * ex + ey * ey + ez * ez) * 0.999999f;      // every surface point lies inside its bounding box
    lb += (double)fmaxf(fmaxf(0.f, d - r), box);
    ub += (double)(d + r);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { lb += __shfl_xor(lb, o, 64); ub += __shfl_xor(ub, o, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = lb; red[1][threadIdx.x >> 6] = ub; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double l = 0.0, u = 0.0;
    for (int w = 0; w < kBThreads / 64; ++w) { l += red[0][w]; u += red[1][w]; }
    // a Tt that is not exactly orthonormal: widen by eta (V max|s| + sum of the upper bounds); max|s| from the surface's box
    const double sx = fmax(fabs((double)b0), fabs((double)b3)), sy = fmax(fabs((double)b1), fabs((double)b4)),
                 sz = fmax(fabs((double)b2), fabs((double)b5));
    const double widen = eta_s * ((double)V * sqrt(sx * sx + sy * sy + sz * sz) + u) * 1.01;
    lb_sum[b] = l - widen;
    ub_sum[b] = u + widen;
  }
}

}  // namespace

extern "C" int isr_adds_bounds(const float* verts, int V, const double* Tq, const double* Tt, int B, const float* field,
                               const double* grid_min, double h, int nx, int ny, int nz, const float* surface_bbox,
                               double* lb_sum, double* ub_sum, isr_stream_t stream_) {
  ISR_REQUIRE(verts && Tq && field && grid_min && surface_bbox && lb_sum && ub_sum, "isr_adds_bounds: null pointer");
  ISR_REQUIRE(V > 0 && B > 0 && h > 0.0 && nx > 0 && ny > 0 && nz > 0 && (long long)nx * ny * nz < (1ll << 31),
              "isr_adds_bounds: V=%d B=%d h=%g grid %d x %d x %d", V, B, h, nx, ny, nz);
  adds_bounds_kernel<<<B, kBThreads, 0, isr::as_stream(stream_)>>>(
      verts, V, Tq, Tt, field, grid_min[0], grid_min[1], grid_min[2], 1.0 / h, h, nx, ny, nz,
      surface_bbox[0], surface_bbox[1], surface_bbox[2], surface_bbox[3], surface_bbox[4], surface_bbox[5], lb_sum, ub_sum);
  ISR_CHECK_LAUNCH("adds bounds kernel");
  return ISR_OK;
}
