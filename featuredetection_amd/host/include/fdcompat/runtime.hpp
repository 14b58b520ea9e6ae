// fdcompat/runtime.hpp -- glue between the reference-shaped C++ classes and the C ABI (fd_hip.h):
// one process-wide context and the status-code -> exception translation.  The reference reports
// errors as std::invalid_argument / std::runtime_error / std::logic_error caught in main()
// (ffpDetectApp.cpp:503-515); the same types are thrown here.
#pragma once
#include <stdexcept>
#include <string>
#include "fd_hip.h"

namespace fdhost {

// Process-wide context (device FD_DEVICE or 0, private stream).  Throws std::runtime_error when no
// gfx950 device is usable: there is no CPU fallback.
fd_ctx* context();

inline void check(int rc) {
    if (rc == FD_OK) return;
    std::string msg = fd_last_error(context());
    switch (rc) {
        case FD_ERR_INVALID_ARGUMENT: throw std::invalid_argument(msg);
        case FD_ERR_LOGIC: throw std::logic_error(msg);
        default: throw std::runtime_error(msg);
    }
}

}  // namespace fdhost
