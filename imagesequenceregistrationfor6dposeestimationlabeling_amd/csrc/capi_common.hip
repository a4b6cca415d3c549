// capi_common.hip — error text, version, device probe, and the relative-pose table kernel.
#include "isr_common.hpp"

#include <atomic>
#include <cstdlib>
#include <cstring>

namespace isr {

char* last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
}

namespace {

struct Tuning {
  std::atomic<int> v[ISR_TUNE_COUNT];
  Tuning() {
    const int defaults[ISR_TUNE_COUNT] = {-1, -1, 1, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < ISR_TUNE_COUNT; ++i) v[i].store(defaults[i], std::memory_order_relaxed);
    // the environment is consulted here and nowhere else: once, before any entry point can run
    if (const char* e = getenv("ISR_NN_GRID")) { if (e[0] >= '0' && e[0] <= '2') v[ISR_TUNE_NN_PATH] = e[0] - '0'; }
    if (const char* e = getenv("ISR_NN_FILTER")) { if (e[0] == '0' || e[0] == '1') v[ISR_TUNE_NN_FILTER] = e[0] - '0'; }
    if (const char* e = getenv("ISR_ICP_WARM")) { if (e[0] == '0') v[ISR_TUNE_ICP_WARM] = 0; }
    if (const char* e = getenv("ISR_NN_PLAN")) {
      int rq = 0; long want = 0;
      if (sscanf(e, "%d,%ld", &rq, &want) >= 1) {
        if (rq == 1 || rq == 4) v[ISR_TUNE_NN_PLAN_RQ] = rq;
        if (want > 0) v[ISR_TUNE_NN_PLAN_BLOCKS] = (int)want;
      }
    }
    if (const char* e = getenv("ISR_NN_TILE")) {
      double st = 0, sq = 0; int tb = 0;
      (void)sscanf(e, "%lf,%lf,%d", &st, &sq, &tb);
      if (st > 0) v[ISR_TUNE_NN_TILE_ST] = (int)(st * 1000.0 + 0.5);
      if (sq > 0) v[ISR_TUNE_NN_TILE_SQ] = (int)(sq * 1000.0 + 0.5);
      if (tb > 0) v[ISR_TUNE_NN_TILE_TB] = tb;
    }
  }
};

Tuning& tuning_state() {
  static Tuning t;      // constructed once (thread-safe), at the first knob access
  return t;
}

// touch the state when the library is loaded, so the environment is read before any thread can change it
const int tuning_loaded = (tuning_state(), 0);

}  // namespace

int tuning(int knob) { return tuning_state().v[knob].load(std::memory_order_relaxed); }

}  // namespace isr

extern "C" int isr_tuning_set(int knob, int value) {
  ISR_REQUIRE(knob >= 0 && knob < ISR_TUNE_COUNT, "isr_tuning_set: unknown knob %d", knob);
  isr::tuning_state().v[knob].store(value, std::memory_order_relaxed);
  return ISR_OK;
}

extern "C" int isr_tuning_get(int knob) {
  if (knob < 0 || knob >= ISR_TUNE_COUNT) return 0;
  return isr::tuning(knob);
}

extern "C" int isr_abi_version(void) { return ISR_ABI_VERSION; }

extern "C" const char* isr_last_error(void) { return isr::last_error_buf(); }

extern "C" int isr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

namespace {

// a10 compute_rel_poses (choosePose.py:43-51): [R_i^T R_j | t_j - t_i]
// a12 calculate_relative_pose (verfication.py:9-19): [R_j|t_j] inv([R_i|t_i]); the inverse of a
// homogeneous 4x4 with general 3x3 block A is [A^-1 | -A^-1 t]; A^-1 by the adjugate so a
// slightly non-orthonormal prediction is inverted, not transposed, as np.linalg.inv would.
__global__ void rel_pose_table_kernel(const double* __restrict__ R, const double* __restrict__ t,
                                      int n, int i0, int rows, int mode, double* __restrict__ out) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)rows * n) return;
  const int i = i0 + (int)(e / n), j = (int)(e % n);
  const double* Ri = R + 9 * (size_t)i;
  const double* Rj = R + 9 * (size_t)j;
  const double* ti = t + 3 * (size_t)i;
  const double* tj = t + 3 * (size_t)j;
  double* o = out + 12 * (size_t)e;
  if (mode == 0) {
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c)
        o[4 * r + c] = fma(Ri[6 + r], Rj[6 + c], fma(Ri[3 + r], Rj[3 + c], Ri[r] * Rj[c]));
      o[4 * r + 3] = tj[r] - ti[r];
    }
  } else {
    // adjugate inverse of Ri
    double inv[9];
    const double a = Ri[0], b = Ri[1], c = Ri[2], d = Ri[3], e2 = Ri[4], f = Ri[5], g = Ri[6],
                 h = Ri[7], k = Ri[8];
    const double A = e2 * k - f * h, Bc = -(d * k - f * g), C = d * h - e2 * g;
    const double det = a * A + b * Bc + c * C;
    const double id = 1.0 / det;
    inv[0] = A * id; inv[1] = -(b * k - c * h) * id; inv[2] = (b * f - c * e2) * id;
    inv[3] = Bc * id; inv[4] = (a * k - c * g) * id; inv[5] = -(a * f - c * d) * id;
    inv[6] = C * id; inv[7] = -(a * h - b * g) * id; inv[8] = (a * e2 - b * d) * id;
    for (int r = 0; r < 3; ++r) {
      double m[3];
      for (int cc = 0; cc < 3; ++cc)
        m[cc] = fma(Rj[3 * r + 2], inv[6 + cc], fma(Rj[3 * r + 1], inv[3 + cc], Rj[3 * r] * inv[cc]));
      o[4 * r] = m[0]; o[4 * r + 1] = m[1]; o[4 * r + 2] = m[2];
      o[4 * r + 3] = tj[r] - fma(m[2], ti[2], fma(m[1], ti[1], m[0] * ti[0]));
    }
  }
}

}  // namespace

extern "C" int isr_rel_pose_table(const double* R, const double* t, int n, int i0, int i1, int mode,
                                  double* out, isr_stream_t stream) {
  ISR_REQUIRE(R && t && out, "isr_rel_pose_table: null pointer");
  ISR_REQUIRE(n > 0 && i0 >= 0 && i1 > i0 && i1 <= n, "isr_rel_pose_table: rows [%d,%d) of n=%d", i0,
              i1, n);
  ISR_REQUIRE(mode == 0 || mode == 1, "isr_rel_pose_table: mode %d", mode);
  const long total = (long)(i1 - i0) * n;
  const int threads = 256;
  rel_pose_table_kernel<<<(unsigned)((total + threads - 1) / threads), threads, 0,
                          isr::as_stream(stream)>>>(R, t, n, i0, i1 - i0, mode, out);
  ISR_CHECK_LAUNCH("rel_pose_table_kernel");
  return ISR_OK;
}
