// Shared host-side helpers for libidelucs_hip.so (error reporting, device properties).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/idelucs_hip.h"

namespace idl {

// thread-local "last error" text behind idl_last_error()
void set_error(const char *fmt, ...);
const char *get_error();

struct DeviceInfo {
    int cus;          // compute units
    int lds_per_cu;   // bytes
    int max_dyn_lds;  // bytes one workgroup may use
};
// cached per device; returns IDL_ERR_HIP when there is no usable device
int device_info(DeviceInfo *out);

}  // namespace idl

#define IDL_HIP_TRY(expr)                                                                      \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess) {                                                               \
            idl::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__,   \
                           __LINE__);                                                          \
            return IDL_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

#define IDL_REQUIRE(cond, msg)                     \
    do {                                           \
        if (!(cond)) {                             \
            idl::set_error("bad argument: %s", msg); \
            return IDL_ERR_ARG;                    \
        }                                          \
    } while (0)
