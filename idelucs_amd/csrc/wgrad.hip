// wgrad.hip -- the weight-gradient kernel of wgrad_device.h on its own (idl_wgrad_rmsprop): dW = dy^T x on the fp32 matrix cores,
// optionally with the RMSprop update in the epilogue.  The training step uses the same tiles as the head of its optimizer launch
// (idl_wgrad_rmsprop_step, train_step.hip); this entry point serves tests, tools/bench_wgrad.py and callers that want the
// gradient alone.
#include <stdlib.h>
#include "dev_env.h"
#include <string.h>

#include "common.h"
#include "wgrad_device.h"

namespace {

template <int VARIANT>
__global__ __launch_bounds__(wg_dev::THREADS) void wgrad_q16_kernel(wg_dev::WgArgs a)
{
    extern __shared__ wg_dev::f32x4_t wg_img[];
    wg_dev::q16_tile<VARIANT>(a, (int)blockIdx.x, wg_img);
}

// diagnostic: the same tiles with clock stamps around the MFMA stream (idl_debug_wgrad_clock)
template <int VARIANT>
__global__ __launch_bounds__(wg_dev::THREADS) void wgrad_q16_clock_kernel(wg_dev::WgArgs a, uint64_t *clk)
{
    extern __shared__ wg_dev::f32x4_t wg_img[];
    wg_dev::q16_tile<VARIANT, true>(a, (int)blockIdx.x, wg_img, clk);
}

}  // namespace

// the tiles' LDS image is 64 KB of dynamic shared memory: raised once per device for every instantiation
static int raise_lds_limit()
{
    static bool attr_set[64] = {};
    int dev = 0;
    IDL_HIP_TRY(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        IDL_HIP_TRY(hipFuncSetAttribute((const void *)wgrad_q16_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, wg_dev::IMG_BYTES));
        IDL_HIP_TRY(hipFuncSetAttribute((const void *)wgrad_q16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, wg_dev::IMG_BYTES));
        IDL_HIP_TRY(hipFuncSetAttribute((const void *)wgrad_q16_clock_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, wg_dev::IMG_BYTES));
        IDL_HIP_TRY(hipFuncSetAttribute((const void *)wgrad_q16_clock_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, wg_dev::IMG_BYTES));
        attr_set[dev] = true;
    }
    return IDL_OK;
}

extern "C" {

int idl_wgrad_supported(int m, int n_out, int n_in)
{
    return wg_dev::supported(m, n_out, n_in) ? 1 : 0;
}

static int wgrad_launch(const float *dy, const float *x, int m, int n_out, int n_in, float *grad, float *W, float *square_avg,
                        const float *hyper, void *stream, uint16_t *w_hi, uint16_t *w_lo, int *over)
{
    IDL_REQUIRE(dy && x && idl_wgrad_supported(m, n_out, n_in), "wgrad: m % 32 == 0, n_out % 64 == 0, n_in % 128 == 0");
    IDL_REQUIRE((W != nullptr) == (square_avg != nullptr) && (W != nullptr || grad != nullptr), "wgrad: give W and square_avg (fused update) and/or grad");
    IDL_REQUIRE(W == nullptr || hyper != nullptr, "wgrad: hyper is needed for the fused update");
    IDL_REQUIRE((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)grad | (uintptr_t)W | (uintptr_t)square_avg) & 15u) == 0, "wgrad: buffers must be 16-byte aligned");
    IDL_REQUIRE((int64_t)m * n_in < (1ll << 29) && (int64_t)n_out * n_in < (1ll << 29), "wgrad: operands beyond 2^31 bytes");
    wg_dev::WgArgs a{};
    a.p = wg_dev::WgProblem{dy, x, grad, W, square_avg, n_out, n_in, w_hi, w_lo, over};
    a.hyper = hyper; a.m = m;
    a.tiles_m = n_out / wg_dev::TM;
    a.tiles = a.tiles_m * (n_in / wg_dev::TN);
    static const bool noload = [] { const char *e = idl::dev_env("wgrad_kernel"); return e != nullptr && strcmp(e, "noload") == 0; }();
    if (const int rc = raise_lds_limit(); rc != IDL_OK) return rc;
    const dim3 grid((unsigned)a.tiles), block(wg_dev::THREADS);
    if (noload) hipLaunchKernelGGL((wgrad_q16_kernel<1>), grid, block, wg_dev::IMG_BYTES, (hipStream_t)stream, a);   // diagnostic
    else hipLaunchKernelGGL((wgrad_q16_kernel<0>), grid, block, wg_dev::IMG_BYTES, (hipStream_t)stream, a);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

int idl_wgrad_rmsprop(const float *dy, const float *x, int m, int n_out, int n_in, float *grad, float *W, float *square_avg,
                      const float *hyper, void *stream)
{
    return wgrad_launch(dy, x, m, n_out, n_in, grad, W, square_avg, hyper, stream, nullptr, nullptr, nullptr);
}

// ... the updated W also written as two fp16 planes (planes.h: what idl_l1_planes reads); *overflow_flag is set to 1 if an entry left their range
int idl_wgrad_rmsprop_planes(const float *dy, const float *x, int m, int n_out, int n_in, float *grad, float *W, float *square_avg,
                             const float *hyper, void *w_hi, void *w_lo, int *overflow_flag, void *stream)
{
    IDL_REQUIRE(W && w_hi && w_lo && overflow_flag && ((((uintptr_t)w_hi) | ((uintptr_t)w_lo)) & 7u) == 0, "wgrad_rmsprop_planes: W, both planes (8-byte aligned) and the flag");
    return wgrad_launch(dy, x, m, n_out, n_in, grad, W, square_avg, hyper, stream, (uint16_t *)w_hi, (uint16_t *)w_lo, overflow_flag);
}

int idl_debug_wgrad_clock(const float *dy, const float *x, int m, int n_out, int n_in, float *grad, int variant, uint64_t *stamps, void *stream)
{
    IDL_REQUIRE(dy && x && grad && stamps && idl_wgrad_supported(m, n_out, n_in), "debug_wgrad_clock: operands as for idl_wgrad_rmsprop, grad and stamps given");
    IDL_REQUIRE(variant == 0 || variant == 2, "debug_wgrad_clock: variant 0 (the product loop) or 2 (operands loaded once)");
    IDL_REQUIRE((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)grad) & 15u) == 0 && (((uintptr_t)stamps) & 7u) == 0, "debug_wgrad_clock: alignment");
    IDL_REQUIRE((int64_t)m * n_in < (1ll << 29) && (int64_t)n_out * n_in < (1ll << 29), "debug_wgrad_clock: operands beyond 2^31 bytes");
    wg_dev::WgArgs a{};
    a.p = wg_dev::WgProblem{dy, x, grad, nullptr, nullptr, n_out, n_in};
    a.m = m;
    a.tiles_m = n_out / wg_dev::TM;
    a.tiles = a.tiles_m * (n_in / wg_dev::TN);
    if (const int rc = raise_lds_limit(); rc != IDL_OK) return rc;
    const dim3 grid((unsigned)a.tiles), block(wg_dev::THREADS);
    if (variant == 2) hipLaunchKernelGGL((wgrad_q16_clock_kernel<2>), grid, block, wg_dev::IMG_BYTES, (hipStream_t)stream, a, stamps);
    else hipLaunchKernelGGL((wgrad_q16_clock_kernel<0>), grid, block, wg_dev::IMG_BYTES, (hipStream_t)stream, a, stamps);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

}  // extern "C"
