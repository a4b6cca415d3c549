// Temporary: entry points declared in isr_hip.h that are not implemented yet return
// ISR_ERR_UNSUPPORTED.  Each moves to its own .hip file as it lands; this file must end up empty.
#include "isr_common.hpp"
#define ISR_TODO(name)                          \
  do {                                          \
    isr::set_error(name ": not implemented");   \
    return ISR_ERR_UNSUPPORTED;                 \
  } while (0)

extern "C" int isr_corr_logsoftmax(const void*, const void*, int, int, int, int, int, int, float*, int64_t, isr_stream_t) { ISR_TODO("isr_corr_logsoftmax"); }
