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

// ---- recorded launches ("plans"): several voters of one ensemble run the same launch sequence on different buffers; instead of
// one launch per voter, each voter's launch is RECORDED (idl_plan_begin + the ordinary launcher + idl_plan_end: every argument
// check and grid computation of the launcher is reused), the records are copied to the device once, and idl_plan_launch runs all
// of them as ONE launch with the voter index in the grid (blockIdx.y, or .z for the InfoNCE passes).
constexpr int PLAN_BYTES = 1024;       // one record: PlanHead + the kernel's parameter struct at PLAN_PARAMS
constexpr int PLAN_PARAMS = 64;
enum { PLAN_MID_FWD = 1, PLAN_NCE = 2, PLAN_MID_BWD = 3, PLAN_RMSPROP = 4, PLAN_WGRAD_RMSPROP = 5, PLAN_L1_FWD = 6,
       PLAN_L1_PLANES = 7, PLAN_REDUCE = 8, PLAN_WGRAD_XPLANES = 9 };      // (7-9: the launches of the step's two-plane form)
struct PlanHead {
    int32_t kind, variant;
    uint32_t grid[3], block, lds;      // of one voter's launch
    uint32_t grid2[3];                 // second launch of the record (InfoNCE pass 2)
};
static_assert(sizeof(PlanHead) <= PLAN_PARAMS, "plan header does not fit");
// the record the next supporting launcher on this thread fills instead of launching (and clears); NULL = launch normally
void *take_plan();
// the batched launches of the other translation units
int nce_plan_launch(const PlanHead &h, const void *dev_plans, int n_voters, hipStream_t stream);

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
