// isr_common.hpp — shared host-side plumbing for libisr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/isr_hip.h"

namespace isr {

// Thread-local text behind isr_last_error().
char* last_error_buf();
void set_error(const char* fmt, ...);

// Value of a tuning knob (ISR_TUNE_*): a relaxed atomic load, no environment access.
int tuning(int knob);

inline hipStream_t as_stream(isr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Launch check: hipGetLastError after a kernel launch, no synchronisation.
#define ISR_CHECK_LAUNCH(what)                                                  \
  do {                                                                          \
    hipError_t e__ = hipGetLastError();                                         \
    if (e__ != hipSuccess) {                                                    \
      ::isr::set_error("%s: %s", what, hipGetErrorString(e__));                 \
      return ISR_ERR_HIP;                                                       \
    }                                                                           \
  } while (0)

#define ISR_CHECK_HIP(expr)                                                     \
  do {                                                                          \
    hipError_t e__ = (expr);                                                    \
    if (e__ != hipSuccess) {                                                    \
      ::isr::set_error("%s: %s", #expr, hipGetErrorString(e__));                \
      return ISR_ERR_HIP;                                                       \
    }                                                                           \
  } while (0)

#define ISR_REQUIRE(cond, ...)                                                  \
  do {                                                                          \
    if (!(cond)) {                                                              \
      ::isr::set_error(__VA_ARGS__);                                            \
      return ISR_ERR_ARG;                                                       \
    }                                                                           \
  } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Bump allocator over the caller's workspace; every carve is 256-byte aligned.
struct Workspace {
  char* base;
  size_t size;
  size_t off = 0;
  Workspace(void* p, size_t n) : base(static_cast<char*>(p)), size(n) {}
  template <typename T>
  T* take(size_t count) {
    off = align_up(off, 256);
    T* p = reinterpret_cast<T*>(base + off);
    off += count * sizeof(T);
    return p;
  }
  bool ok() const { return off <= size; }
};

constexpr int kWave = 64;  // gfx950 wavefront

// The order-preserving unsigned image of a float (radix select of the top-80 % cut, csrc/select_top.hip; its leading
// 11-bit digit is what K1's epilogue counts for isr_corr_argmax_digits): u(a) < u(b)  <=>  a < b for non-NaN a, b.
__host__ __device__ __forceinline__ uint32_t ordered_bits(float f) {
  union { float f; uint32_t u; } c;
  c.f = f;
  return c.u ^ ((c.u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
constexpr int kDigitBins = 2048;    // bins of the leading digit (bits 21 .. 31 of ordered_bits)
constexpr int kDigitShift = 21;

}  // namespace isr
