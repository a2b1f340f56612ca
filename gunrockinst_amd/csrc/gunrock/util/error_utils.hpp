// error_utils.hpp -- HIP error reporting for the MI355X frontier engine.
//
// Mirrors the reporting contract of the reference's util::GRError
// (gunrock/util/error_utils.cu:21-78): failures are printed to stderr as
//   [file, line] message (HIP error N: string)
// and the error code is handed back so callers can `if (retval = GRError(...)) break;`.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdio>

namespace gunrock {
namespace util {

inline hipError_t GRError(hipError_t error, const char *message, const char *filename, int line,
                          bool print = true)
{
    if (error != hipSuccess && print) {
        std::fprintf(stderr, "[%s, %d] %s (HIP error %d: %s)\n", filename, line, message,
                     static_cast<int>(error), hipGetErrorString(error));
        std::fflush(stderr);
    }
    return error;
}

// Checks (and clears) the sticky launch error.
inline hipError_t GRError(const char *message, const char *filename, int line, bool print = true)
{
    return GRError(hipGetLastError(), message, filename, line, print);
}

inline hipError_t GRError(hipError_t error, bool print = true)
{
    if (error != hipSuccess && print) {
        std::fprintf(stderr, "(HIP error %d: %s)\n", static_cast<int>(error), hipGetErrorString(error));
        std::fflush(stderr);
    }
    return error;
}

}  // namespace util
}  // namespace gunrock

#define GR_CHECK(call, msg)                                                                  \
    do {                                                                                     \
        if ((retval = gunrock::util::GRError((call), (msg), __FILE__, __LINE__))) return retval; \
    } while (0)
