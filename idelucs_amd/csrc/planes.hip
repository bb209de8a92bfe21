// planes.hip -- the two-fp16-plane operand form (planes.h) outside the step's fused launches: idl_split_planes (an fp32 tensor -> its
// planes: W1 when an epoch begins, the first batch of an epoch, tests) and idl_l1_planes (the layer-1 forward tiles of
// l1_planes_device.h: the first launch of the two-plane step).
#include <stdlib.h>
#include "dev_env.h"
#include <string.h>

#include "common.h"
#include "l1_planes_device.h"

namespace {

__global__ __launch_bounds__(256) void split_planes_kernel(const float4 *__restrict__ src, int64_t n4, float sc, uint2 *__restrict__ hi,
                                                           uint2 *__restrict__ lo, int *__restrict__ flag)
{
    bool over = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = src[i];
        uint2 h, l;
        over |= idl_planes::split4(v.x * sc, v.y * sc, v.z * sc, v.w * sc, h, l);
        hi[i] = h; lo[i] = l;
    }
    if (over && flag != nullptr) *flag = 1;
}

__global__ __launch_bounds__(l1p_dev::THREADS, 1) void l1_planes_kernel(l1p_dev::L1pArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char l1p_smem[];
    l1p_dev::l1p_body(a, (int)blockIdx.x, l1p_smem);
}

}  // namespace

extern "C" {

int idl_planes_exponent(int which) { return which == 0 ? idl_planes::X_EXP : (which == 1 ? idl_planes::W_EXP : idl_planes::DR1_K_FIRST); }

int idl_dr1_scale_words(void) { return idl_planes::DR1_WORDS; }

int idl_split_planes(const float *src, int64_t n, int exponent, void *hi, void *lo, int *overflow_flag, void *stream)
{
    IDL_REQUIRE(n >= 0 && (n & 3) == 0 && exponent >= -60 && exponent <= 60, "split_planes: 4 | n, |exponent| <= 60");
    if (n == 0) return IDL_OK;
    IDL_REQUIRE(src && hi && lo, "split_planes: NULL buffer");
    IDL_REQUIRE((((uintptr_t)src) & 15u) == 0 && ((((uintptr_t)hi) | ((uintptr_t)lo)) & 7u) == 0, "split_planes: src 16-byte, planes 8-byte aligned");
    const int64_t n4 = n / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)src, n4, ldexpf(1.f, exponent),
                       (uint2 *)hi, (uint2 *)lo, overflow_flag);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

int idl_l1_planes_parts(void) { return l1p_dev::KSPLIT; }

int idl_l1_planes_supported(int m, int n_out, int n_in) { return l1p_dev::supported(m, n_out, n_in) ? 1 : 0; }

int idl_l1_planes(const void *w_hi, const void *w_lo, int ld_w, const void *x_hi, const void *x_lo, int ld_x, int m, int n_out, int n_in, float *part, void *stream)
{
    IDL_REQUIRE(ld_w >= n_in && ld_x >= n_in && ld_w <= n_in + 1024 && ld_x <= n_in + 1024 && (ld_w & 7) == 0 && (ld_x & 7) == 0, "l1_planes: n_in <= ld <= n_in + 1024, 8 | ld");
    IDL_REQUIRE(w_hi && w_lo && x_hi && x_lo && part && l1p_dev::supported(m, n_out, n_in),
                "l1_planes: m % 128 == 0, n_out % 128 == 0, n_in % 512 == 0, n_in >= 1024");
    IDL_REQUIRE(((((uintptr_t)w_hi) | ((uintptr_t)w_lo) | ((uintptr_t)x_hi) | ((uintptr_t)x_lo) | ((uintptr_t)part)) & 15u) == 0, "l1_planes: buffers must be 16-byte aligned");
    static const int l1p_dbg = idl::dev_env("l1p_dbg") ? atoi(idl::dev_env("l1p_dbg")) : 0;      // (timing ablations; l1_planes_device.h)
    l1p_dev::L1pArgs a{(const uint16_t *)w_hi, (const uint16_t *)w_lo, (const uint16_t *)x_hi, (const uint16_t *)x_lo, part, m, n_out, n_in,
                       (n_out / l1p_dev::TM) * (m / l1p_dev::TN) * l1p_dev::KSPLIT, ld_w, ld_x, l1p_dbg};
    if (void *plan = idl::take_plan()) {          // recorded, not launched (idl_plan_begin): several voters in one launch, train_step.hip
        static_assert(sizeof(l1p_dev::L1pArgs) + idl::PLAN_PARAMS <= idl::PLAN_BYTES, "L1pArgs does not fit a plan record");
        idl::PlanHead h{};
        h.kind = idl::PLAN_L1_PLANES; h.grid[0] = (unsigned)a.n_tiles; h.grid[1] = 1; h.grid[2] = 1; h.block = l1p_dev::THREADS; h.lds = l1p_dev::LDS_BYTES;
        memcpy(plan, &h, sizeof(h));
        memcpy((unsigned char *)plan + idl::PLAN_PARAMS, &a, sizeof(a));
        return IDL_OK;
    }
    static bool attr_set[64] = {};
    int dev = 0;
    IDL_HIP_TRY(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        IDL_HIP_TRY(hipFuncSetAttribute((const void *)l1_planes_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, l1p_dev::LDS_BYTES));
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(l1_planes_kernel, dim3((unsigned)a.n_tiles), dim3(l1p_dev::THREADS), l1p_dev::LDS_BYTES, (hipStream_t)stream, a);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

}  // extern "C"
