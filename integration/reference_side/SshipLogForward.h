// Reference-side binding, shared by the three adapters (SuperPoint.h, LightGlue.h, EigenPlaces.h): everything libsuperslam_hip.so reports
// through its log callback (sship_set_log_callback, include/sship.h: level 0 trace .. 4 error) goes to the reference's own logger, the
// SLOG_* macros of include/Logging.h:21-26 - so a failing hipMalloc, a rejected weights file or a pool exhaustion inside the library shows up
// in the SuperSLAM log with the library's message, next to the adapter's own "…(HIP): initialize failed" line.
// Installed once per process by the first adapter object constructed; the callback may be invoked from any thread that calls the library
// (spdlog loggers are thread-safe).
#ifndef SSHIP_LOG_FORWARD_H_
#define SSHIP_LOG_FORWARD_H_

#include "Logging.h"
#include "sship.h"

namespace superslam_hip_adapter {
inline void forward_library_log(int level, const char* msg) {
  const char* m = msg ? msg : "";
  switch (level) {
    case 0: SLOG_TRACE("libsuperslam_hip: {}", m); break;
    case 1: SLOG_DEBUG("libsuperslam_hip: {}", m); break;
    case 2: SLOG_INFO("libsuperslam_hip: {}", m); break;
    case 3: SLOG_WARN("libsuperslam_hip: {}", m); break;
    default: SLOG_ERROR("libsuperslam_hip: {}", m); break;
  }
}
inline void install_log_forwarding() {
  static const bool installed = (sship_set_log_callback(&forward_library_log), true);
  (void)installed;
}
}  // namespace superslam_hip_adapter

#endif  // SSHIP_LOG_FORWARD_H_
