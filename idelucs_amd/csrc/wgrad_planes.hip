// wgrad_planes.hip -- idl_wgrad_rmsprop_xplanes: the kernel of wgrad_planes_device.h as a launch of its own (tests, tools/bench_planes.py; in the
// step it carries the optimizer tail: train_step.hip, idl_wgrad_xplanes_rms).
#include <stdlib.h>
#include "dev_env.h"

#include "common.h"
#include "wgrad_planes_device.h"

namespace {

using namespace wgp_dev;

// (the compiler's allocator stops at v199, the K-steps' operand sets live above it: wgrad_planes_device.h)
__global__ __launch_bounds__(NT, 1) __attribute__((amdgpu_num_vgpr(200))) void wgrad_dplanes_kernel(XpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char wgd_smem[];
    dplanes_body(a, wgd_smem, [](int) {});
}

}  // namespace

extern "C" int idl_wgrad_xplanes_supported(int m, int n_out, int n_in)
{
    return (m % dpl::KC2 == 0 && m / dpl::KC2 >= dpl::NS && n_out >= TM && n_out % TM == 0 && n_in >= TN && n_in % TN == 0 &&
            (int64_t)m * (n_in + 1024) < (1ll << 29) && (int64_t)n_out * n_in < (1ll << 29) && (int64_t)m * n_out < (1ll << 30)) ? 1 : 0;
}

extern "C" int idl_wgrad_rmsprop_xplanes(const void *dy_hi, const void *dy_lo, int *dy_scale, const void *x_hi, const void *x_lo, int ld_x, int m, int n_out, int n_in,
                                         float *grad, float *W, float *square_avg, const float *hyper, void *w_hi, void *w_lo, int *overflow_flag, void *stream)
{
    IDL_REQUIRE(dy_hi && dy_lo && dy_scale && x_hi && x_lo && idl_wgrad_xplanes_supported(m, n_out, n_in), "wgrad_xplanes: m % 64 == 0, m >= 192, n_out % 64 == 0, n_in % 128 == 0");
    IDL_REQUIRE(ld_x >= n_in && ld_x <= n_in + 1024 && (ld_x & 7) == 0, "wgrad_xplanes: n_in <= ld_x <= n_in + 1024, 8 | ld_x");
    IDL_REQUIRE((W != nullptr) == (square_avg != nullptr) && (W != nullptr || grad != nullptr), "wgrad_xplanes: give W and square_avg (fused update) and/or grad");
    IDL_REQUIRE(W == nullptr || hyper != nullptr, "wgrad_xplanes: hyper is needed for the fused update");
    IDL_REQUIRE((w_hi != nullptr) == (w_lo != nullptr) && (w_hi == nullptr || (W != nullptr && overflow_flag != nullptr)), "wgrad_xplanes: W's planes need both planes, W and the flag");
    IDL_REQUIRE((((uintptr_t)dy_hi | (uintptr_t)dy_lo | (uintptr_t)x_hi | (uintptr_t)x_lo | (uintptr_t)grad | (uintptr_t)W | (uintptr_t)square_avg) & 15u) == 0 &&
                (((uintptr_t)w_hi | (uintptr_t)w_lo) & 7u) == 0, "wgrad_xplanes: buffers 16-byte aligned, W's planes 8-byte");
    XpArgs a{};
    a.dyh = (const uint16_t *)dy_hi; a.dyl = (const uint16_t *)dy_lo; a.dy_scale = dy_scale;
    a.xh = (const uint16_t *)x_hi; a.xl = (const uint16_t *)x_lo; a.grad = grad; a.W = W; a.V = square_avg;
    a.wh = (uint16_t *)w_hi; a.wl = (uint16_t *)w_lo; a.over = overflow_flag; a.hyper = hyper;
    a.m = m; a.n_out = n_out; a.n_in = n_in; a.ldx = ld_x;
    a.tiles_m = n_out / TM; a.tiles = a.tiles_m * (n_in / TN);
    static const int wgp_dbg = idl::dev_env("wgp_dbg") ? atoi(idl::dev_env("wgp_dbg")) : 0;      // (timing ablations; wgrad_planes_device.h)
    a.dbg = wgp_dbg;
    static bool attr_set[64] = {};
    int dev = 0;
    IDL_HIP_TRY(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        IDL_HIP_TRY(hipFuncSetAttribute((const void *)wgrad_dplanes_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + 16));
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(wgrad_dplanes_kernel, dim3((unsigned)a.tiles), dim3(NT), LDS_BYTES + 16, (hipStream_t)stream, a);      // (+ 16: the tail's words)
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}
