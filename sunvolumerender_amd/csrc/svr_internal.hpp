// svr_internal.hpp -- what svr_api.hip offers the other translation units of libsvr_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

namespace svr {
int report_error(int code, const char* msg);   // records (or, in fatal mode, prints and exits); returns code
int ensure_ready();                            // device selected, context created; 0 on success
hipStream_t current_stream();                  // the caller's stream (svr_set_stream)

inline int failf(int code, const char* fmt, ...)
{
    char buf[768];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return report_error(code, buf);
}
}
