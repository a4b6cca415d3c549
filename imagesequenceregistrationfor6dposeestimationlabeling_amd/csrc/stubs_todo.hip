// Temporary: entry points declared in isr_hip.h that are not implemented yet return
// ISR_ERR_UNSUPPORTED.  Each moves to its own .hip file as it lands; this file must end up empty.
#include "isr_common.hpp"
#define ISR_TODO(name)                          \
  do {                                          \
    isr::set_error(name ": not implemented");   \
    return ISR_ERR_UNSUPPORTED;                 \
  } while (0)

extern "C" int isr_corr_logsoftmax(const void*, const void*, int, int, int, int, int, int, float*, int64_t, isr_stream_t) { ISR_TODO("isr_corr_logsoftmax"); }
extern "C" size_t isr_select_top_workspace_bytes(int) { return 0; }
extern "C" int isr_select_top(const float*, int, double, int, int32_t*, int32_t*, float*, void*, size_t, isr_stream_t) { ISR_TODO("isr_select_top"); }
extern "C" int isr_gather_corr(const int32_t*, const int32_t*, const int32_t*, int, const float*, int, const float*, float*, float*, isr_stream_t) { ISR_TODO("isr_gather_corr"); }
extern "C" size_t isr_pnp_ransac_workspace_bytes(int, int) { return 0; }
extern "C" int isr_p3p_hypotheses(const float*, const float*, const int32_t*, int, const double*, int, uint64_t, double*, uint8_t*, int32_t*, isr_stream_t) { ISR_TODO("isr_p3p_hypotheses"); }
extern "C" int isr_ransac_score(const float*, const float*, const int32_t*, int, const double*, const double*, const uint8_t*, int, float, int32_t*, int32_t*, uint32_t*, isr_stream_t) { ISR_TODO("isr_ransac_score"); }
extern "C" int isr_pnp_refine(const float*, const float*, const int32_t*, int, const uint32_t*, const double*, int, double*, void*, size_t, isr_stream_t) { ISR_TODO("isr_pnp_refine"); }
extern "C" int isr_pnp_ransac(const float*, const float*, const int32_t*, int, const double*, int, uint64_t, float, int, double*, int32_t*, int32_t*, int32_t*, void*, size_t, isr_stream_t) { ISR_TODO("isr_pnp_ransac"); }
