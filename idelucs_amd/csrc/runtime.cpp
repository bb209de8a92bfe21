// runtime.cpp -- error text, device discovery (host side of libidelucs_hip.so).
#include <stdarg.h>
#include <string.h>

#include <mutex>

#include "common.h"

namespace idl {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char *get_error() { return g_err; }

static std::mutex g_mu;
static constexpr int MAX_DEV = 64;
static DeviceInfo g_info[MAX_DEV];
static bool g_have[MAX_DEV];

int device_info(DeviceInfo *out)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= MAX_DEV) {
        set_error("no usable HIP device (hipGetDevice: %s); libidelucs_hip has no CPU fallback",
                  e == hipSuccess ? "device index out of range" : hipGetErrorString(e));
        return IDL_ERR_HIP;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_have[dev]) {
        hipDeviceProp_t p;
        e = hipGetDeviceProperties(&p, dev);
        if (e != hipSuccess) {
            set_error("hipGetDeviceProperties failed: %s", hipGetErrorString(e));
            return IDL_ERR_HIP;
        }
        if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
            set_error("device %d is %s; libidelucs_hip is built for gfx950 (MI355X) only", dev, p.gcnArchName);
            return IDL_ERR_HIP;
        }
        DeviceInfo di;
        di.cus = p.multiProcessorCount;
        di.lds_per_cu = 160 * 1024;  // CDNA4: 160 KiB per CU (maxSharedMemoryPerMultiProcessor reports it too)
        if (p.maxSharedMemoryPerMultiProcessor > 0) di.lds_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
        di.max_dyn_lds = 160 * 1024;
        g_info[dev] = di;
        g_have[dev] = true;
    }
    *out = g_info[dev];
    return IDL_OK;
}

namespace {
thread_local void *g_plan = nullptr;
thread_local bool g_plan_filled = false;
}
void *take_plan()
{
    void *p = g_plan;
    g_plan = nullptr;
    if (p) g_plan_filled = true;
    return p;
}

}  // namespace idl

extern "C" {

const char *idl_last_error(void) { return idl::get_error(); }

int64_t idl_plan_bytes(void) { return idl::PLAN_BYTES; }

int idl_plan_begin(void *host_plan)
{
    IDL_REQUIRE(host_plan, "plan_begin: NULL record");
    memset(host_plan, 0, idl::PLAN_BYTES);
    idl::g_plan = host_plan;
    idl::g_plan_filled = false;
    return IDL_OK;
}

int idl_plan_end(void)
{
    const bool pending = idl::g_plan != nullptr, filled = idl::g_plan_filled;
    idl::g_plan = nullptr;
    idl::g_plan_filled = false;
    IDL_REQUIRE(!pending && filled, "plan_end: the call after idl_plan_begin was not a launcher that can be recorded (or failed)");
    return IDL_OK;
}

int idl_abi_version(void) { return 1; }

int idl_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, d) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

}  // extern "C"
