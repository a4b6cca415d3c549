// p3p_device.hpp — device-side Philox4x32-10 and the f64 P3P solver shared by ransac.hip (K2,
// cv2.solvePnPRansac replacement) and estimate_pose.hip (cv2.solveP3P replacement,
// poseEstSurf.py:138).  Only + - x / sqrt fma: plain IEEE arithmetic, no libm transcendentals.
#pragma once
#include "isr_common.hpp"

namespace isr_p3p {

// ---------------------------------------------------------------------------------- Philox
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// ------------------------------------------------------------------------------ small f64 math
struct V3 { double x, y, z; };
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return fma(a.z, b.z, fma(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ V3 scale(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }

// adjugate of a symmetric 3x3 given as s = {a00,a01,a02,a11,a12,a22}; result same packing
__device__ __forceinline__ void sym_adj(const double* s, double* o) {
  o[0] = s[3] * s[5] - s[4] * s[4];
  o[1] = s[2] * s[4] - s[1] * s[5];
  o[2] = s[1] * s[4] - s[2] * s[3];
  o[3] = s[0] * s[5] - s[2] * s[2];
  o[4] = s[1] * s[2] - s[0] * s[4];
  o[5] = s[0] * s[3] - s[1] * s[1];
}
__device__ __forceinline__ double sym_det(const double* s, const double* adj) {
  return s[0] * adj[0] + s[1] * adj[1] + s[2] * adj[2];
}
// trace(adj(A) B) for symmetric packed A-adjugate and B
__device__ __forceinline__ double sym_tr(const double* a, const double* b) {
  return a[0] * b[0] + a[3] * b[3] + a[5] * b[5] + 2.0 * (a[1] * b[1] + a[2] * b[2] + a[4] * b[4]);
}
__device__ __forceinline__ double sym_quad(const double* s, const double* u, const double* v) {
  // u^T S v
  return u[0] * (s[0] * v[0] + s[1] * v[1] + s[2] * v[2]) +
         u[1] * (s[1] * v[0] + s[3] * v[1] + s[4] * v[2]) +
         u[2] * (s[2] * v[0] + s[4] * v[1] + s[5] * v[2]);
}

// One real root of x^3 + b x^2 + c x + d by Newton from a start on the convex side of an outer
// root (monotone convergence); fixed op sequence.
__device__ double cubic_root(double b, double c, double d) {
  double r;
  const double disc = b * b - 3.0 * c;
  if (disc >= 0.0) {
    const double v = sqrt(disc);
    const double t1 = (-b - v) / 3.0;  // local max
    double k = ((t1 + b) * t1 + c) * t1 + d;
    if (k > 0.0) {
      r = t1 - sqrt(-k / (3.0 * t1 + b));  // left of the local max: leftmost root
    } else {
      const double t2 = (-b + v) / 3.0;  // local min
      k = ((t2 + b) * t2 + c) * t2 + d;
      r = t2 + sqrt(-k / (3.0 * t2 + b));
    }
  } else {
    r = -b / 3.0;
    if (fabs((3.0 * r + 2.0 * b) * r + c) < 1e-4) r += 1.0;
  }
  for (int it = 0; it < 50; ++it) {
    const double f = ((r + b) * r + c) * r + d;
    const double fp = (3.0 * r + 2.0 * b) * r + c;
    if (fp == 0.0) break;
    const double step = f / fp;
    r -= step;
    if (fabs(step) <= 1e-15 * fabs(r)) break;
  }
  return r;
}

struct P3PIn {
  V3 x[3];  // object points
  V3 y[3];  // unit bearing vectors
};

// Degenerate-conic P3P.  Writes up to 4 depth triples into lam[][3]; returns the count.
__device__ int p3p_depths(const P3PIn& in, double lam[4][3]) {
  const V3 d12 = sub(in.x[0], in.x[1]), d13 = sub(in.x[0], in.x[2]), d23 = sub(in.x[1], in.x[2]);
  const double a12 = dot(d12, d12), a13 = dot(d13, d13), a23 = dot(d23, d23);
  const double b12 = dot(in.y[0], in.y[1]), b13 = dot(in.y[0], in.y[2]), b23 = dot(in.y[1], in.y[2]);
  if (!(a12 > 0.0) || !(a13 > 0.0) || !(a23 > 0.0)) return 0;
  // Lambda^T D1 Lambda = 0, Lambda^T D2 Lambda = 0 (packed symmetric 00,01,02,11,12,22)
  const double D1[6] = {a23, -a23 * b12, 0.0, a23 - a12, a12 * b23, -a12};
  const double D2[6] = {a23, 0.0, -a23 * b13, -a13, a13 * b23, a23 - a13};
  double A1[6], A2[6];
  sym_adj(D1, A1);
  sym_adj(D2, A2);
  const double c3 = sym_det(D2, A2), c0 = sym_det(D1, A1);
  const double c2 = sym_tr(A2, D1), c1 = sym_tr(A1, D2);
  if (!(fabs(c3) > 0.0)) return 0;
  const double g = cubic_root(c2 / c3, c1 / c3, c0 / c3);
  double D0[6], B[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) D0[i] = fma(g, D2[i], D1[i]);
  sym_adj(D0, B);
  // B = -p p^T for a real line pair: take the largest diagonal magnitude
  int i = 0;
  double bd = B[0];
  if (fabs(B[3]) > fabs(bd)) { bd = B[3]; i = 1; }
  if (fabs(B[5]) > fabs(bd)) { bd = B[5]; i = 2; }
  if (!(bd < 0.0)) return 0;
  const double inv = 1.0 / sqrt(-bd);
  double p[3];
  if (i == 0) { p[0] = -B[0] * inv; p[1] = -B[1] * inv; p[2] = -B[2] * inv; }
  else if (i == 1) { p[0] = -B[1] * inv; p[1] = -B[3] * inv; p[2] = -B[4] * inv; }
  else { p[0] = -B[2] * inv; p[1] = -B[4] * inv; p[2] = -B[5] * inv; }
  // N = D0 + [p]x  = 2 m l^T (rank 1): rows are multiples of l, columns multiples of m
  const double N[3][3] = {{D0[0], D0[1] - p[2], D0[2] + p[1]},
                          {D0[1] + p[2], D0[3], D0[4] - p[0]},
                          {D0[2] - p[1], D0[4] + p[0], D0[5]}};
  int bj = 0, bk = 0;
  double bm = 0.0;
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (fabs(N[j][k]) > bm) { bm = fabs(N[j][k]); bj = j; bk = k; }
  if (!(bm > 0.0)) return 0;
  double line[2][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    line[0][k] = (bj == 0) ? N[0][k] : (bj == 1) ? N[1][k] : N[2][k];
    line[1][k] = (bk == 0) ? N[k][0] : (bk == 1) ? N[k][1] : N[k][2];
  }
  // second conic for the intersection: the better scaled of D1 / D2 is immaterial; use D1, and
  // recover the scale from the largest a_ij.
  int n = 0;
  for (int li = 0; li < 2; ++li) {
    const double* l = line[li];
    int k = 0;
    if (fabs(l[1]) > fabs(l[k])) k = 1;
    if (fabs(l[2]) > fabs(l[k])) k = 2;
    const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
    double u[3] = {0, 0, 0}, v[3] = {0, 0, 0};
    const double lk = (k == 0) ? l[0] : (k == 1) ? l[1] : l[2];
    const double l1 = (k1 == 0) ? l[0] : (k1 == 1) ? l[1] : l[2];
    const double l2 = (k2 == 0) ? l[0] : (k2 == 1) ? l[1] : l[2];
    // basis of the plane l^T Lambda = 0
    double uk = -l1 / lk, vk = -l2 / lk;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      u[c] = (c == k1) ? 1.0 : (c == k) ? uk : 0.0;
      v[c] = (c == k2) ? 1.0 : (c == k) ? vk : 0.0;
    }
    const double qa = sym_quad(D1, u, u), qb = sym_quad(D1, u, v), qc = sym_quad(D1, v, v);
    // qa mu^2 + 2 qb mu nu + qc nu^2 = 0
    const double disc = qb * qb - qa * qc;
    if (!(disc >= 0.0)) continue;
    const double sq = sqrt(disc);
    for (int sgn = 0; sgn < 2; ++sgn) {
      double mu, nu;
      const double num = sgn ? (-qb - sq) : (-qb + sq);
      if (fabs(qa) >= fabs(qc)) { if (qa == 0.0) continue; mu = num / qa; nu = 1.0; }
      else { mu = 1.0; nu = num / qc; }
      double dvec[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) dvec[c] = fma(mu, u[c], nu * v[c]);
      // orientation: all depths positive
      if (dvec[0] < 0.0 && dvec[1] < 0.0 && dvec[2] < 0.0) {
        dvec[0] = -dvec[0]; dvec[1] = -dvec[1]; dvec[2] = -dvec[2];
      }
      if (!(dvec[0] > 0.0 && dvec[1] > 0.0 && dvec[2] > 0.0)) continue;
      // scale from the largest side
      double q, a;
      if (a12 >= a13 && a12 >= a23) { q = dvec[0] * dvec[0] + dvec[1] * dvec[1] - 2.0 * b12 * dvec[0] * dvec[1]; a = a12; }
      else if (a13 >= a23) { q = dvec[0] * dvec[0] + dvec[2] * dvec[2] - 2.0 * b13 * dvec[0] * dvec[2]; a = a13; }
      else { q = dvec[1] * dvec[1] + dvec[2] * dvec[2] - 2.0 * b23 * dvec[1] * dvec[2]; a = a23; }
      if (!(q > 0.0)) continue;
      const double s = sqrt(a / q);
      double L[3] = {dvec[0] * s, dvec[1] * s, dvec[2] * s};
      // polish: Newton on the three distance equations
      for (int it = 0; it < 3; ++it) {
        const double r0 = L[0] * L[0] + L[1] * L[1] - 2.0 * b12 * L[0] * L[1] - a12;
        const double r1 = L[0] * L[0] + L[2] * L[2] - 2.0 * b13 * L[0] * L[2] - a13;
        const double r2 = L[1] * L[1] + L[2] * L[2] - 2.0 * b23 * L[1] * L[2] - a23;
        const double J00 = 2.0 * (L[0] - b12 * L[1]), J01 = 2.0 * (L[1] - b12 * L[0]);
        const double J10 = 2.0 * (L[0] - b13 * L[2]), J12 = 2.0 * (L[2] - b13 * L[0]);
        const double J21 = 2.0 * (L[1] - b23 * L[2]), J22 = 2.0 * (L[2] - b23 * L[1]);
        // J = [[J00,J01,0],[J10,0,J12],[0,J21,J22]]
        const double det = -J00 * J12 * J21 - J01 * J10 * J22;
        if (!(fabs(det) > 0.0)) break;
        const double id = 1.0 / det;
        // delta = J^-1 r via the adjugate
        const double e0 = (-J12 * J21 * r0 - J01 * J22 * r1 + J01 * J12 * r2) * id;
        const double e1 = (-J10 * J22 * r0 + J00 * J22 * r1 - J00 * J12 * r2) * id;
        const double e2 = (J10 * J21 * r0 - J00 * J21 * r1 - J01 * J10 * r2) * id;
        L[0] -= e0; L[1] -= e1; L[2] -= e2;
      }
      if (!(L[0] > 0.0 && L[1] > 0.0 && L[2] > 0.0)) continue;
      if (n < 4) { lam[n][0] = L[0]; lam[n][1] = L[1]; lam[n][2] = L[2]; ++n; }
    }
  }
  return n;
}

// Pose from depths: R (x_i - x_j) = Y_i - Y_j with Y_i = lam_i y_i.
__device__ bool pose_from_depths(const P3PIn& in, const double* L, double* Rt) {
  const V3 Y0 = scale(in.y[0], L[0]), Y1 = scale(in.y[1], L[1]), Y2 = scale(in.y[2], L[2]);
  const V3 xa = sub(in.x[0], in.x[1]), xb = sub(in.x[0], in.x[2]), xc = cross(xa, xb);
  const V3 ya = sub(Y0, Y1), yb = sub(Y0, Y2), yc = cross(ya, yb);
  // X = [xa xb xc] (columns); X^-1 rows = cross products / det
  const double det = dot(xa, cross(xb, xc));
  if (!(fabs(det) > 0.0)) return false;
  const double id = 1.0 / det;
  const V3 r0 = scale(cross(xb, xc), id), r1 = scale(cross(xc, xa), id), r2 = scale(cross(xa, xb), id);
  // R = Ya r0^T + Yb r1^T + Yc r2^T
  const double R[9] = {
      ya.x * r0.x + yb.x * r1.x + yc.x * r2.x, ya.x * r0.y + yb.x * r1.y + yc.x * r2.y, ya.x * r0.z + yb.x * r1.z + yc.x * r2.z,
      ya.y * r0.x + yb.y * r1.x + yc.y * r2.x, ya.y * r0.y + yb.y * r1.y + yc.y * r2.y, ya.y * r0.z + yb.y * r1.z + yc.y * r2.z,
      ya.z * r0.x + yb.z * r1.x + yc.z * r2.x, ya.z * r0.y + yb.z * r1.y + yc.z * r2.y, ya.z * r0.z + yb.z * r1.z + yc.z * r2.z};
  const V3 x0 = in.x[0];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    Rt[4 * r] = R[3 * r]; Rt[4 * r + 1] = R[3 * r + 1]; Rt[4 * r + 2] = R[3 * r + 2];
  }
  Rt[3] = Y0.x - (R[0] * x0.x + R[1] * x0.y + R[2] * x0.z);
  Rt[7] = Y0.y - (R[3] * x0.x + R[4] * x0.y + R[5] * x0.z);
  Rt[11] = Y0.z - (R[6] * x0.x + R[7] * x0.y + R[8] * x0.z);
  bool fin = true;
#pragma unroll
  for (int i = 0; i < 12; ++i) fin = fin && (fabs(Rt[i]) < 1e300);
  return fin;
}


struct Cam { double k[9]; double ki[9]; };

inline bool make_cam(const double* K, Cam* cam) {
  for (int i = 0; i < 9; ++i) cam->k[i] = K[i];
  const double a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], k = K[8];
  const double A = e * k - f * h, B = -(d * k - f * g), C = d * h - e * g;
  const double det = a * A + b * B + c * C;
  if (!(det != 0.0)) return false;
  const double id = 1.0 / det;
  cam->ki[0] = A * id; cam->ki[1] = -(b * k - c * h) * id; cam->ki[2] = (b * f - c * e) * id;
  cam->ki[3] = B * id; cam->ki[4] = (a * k - c * g) * id; cam->ki[5] = -(a * f - c * d) * id;
  cam->ki[6] = C * id; cam->ki[7] = -(a * h - b * g) * id; cam->ki[8] = (a * e - b * d) * id;
  return true;
}

// unit bearing vector of pixel (u, v)
__device__ __forceinline__ V3 bearing(const Cam& cam, double u, double v) {
  V3 y = {cam.ki[0] * u + cam.ki[1] * v + cam.ki[2], cam.ki[3] * u + cam.ki[4] * v + cam.ki[5],
          cam.ki[6] * u + cam.ki[7] * v + cam.ki[8]};
  return scale(y, 1.0 / sqrt(dot(y, y)));
}

// squared reprojection error of X under [R|t] (12 doubles); returns false when z <= 0
__device__ __forceinline__ bool reproj_err2(const Cam& cam, const double* Rt, V3 X, double u, double v, double* e2) {
  const double xc = Rt[0] * X.x + Rt[1] * X.y + Rt[2] * X.z + Rt[3];
  const double yc = Rt[4] * X.x + Rt[5] * X.y + Rt[6] * X.z + Rt[7];
  const double zc = Rt[8] * X.x + Rt[9] * X.y + Rt[10] * X.z + Rt[11];
  if (!(zc > 0.0)) return false;
  const double px = cam.k[0] * xc + cam.k[1] * yc + cam.k[2] * zc;
  const double py = cam.k[3] * xc + cam.k[4] * yc + cam.k[5] * zc;
  const double pz = cam.k[6] * xc + cam.k[7] * yc + cam.k[8] * zc;
  const double eu = px / pz - u, ev = py / pz - v;
  *e2 = eu * eu + ev * ev;
  return true;
}

}  // namespace isr_p3p
