// l1_fwd.hip -- the layer-1 forward of NetLinear, a1^T = W1 x^T, on this package's own fp32 MFMA tiles: the first launch of the training step's FP32 FORM
// (IDELUCS_PLANES=0, shapes the two-plane kernels do not take, the automatic fall-back of a run whose data left the planes' range).
//
// Reference: Linear(F,512) of idelucs/PytorchUtils.py:38, called for both views of a batch at idelucs/models.py:124-125.  A workgroup owns a
// 64 (hidden units) x 32 (batch rows) tile; the bias, ReLU and Dropout are the mid-forward launch's.  (Round 4's form with those and the K-split of
// Linear(512, 64) in this kernel's epilogue was measured slower in the step and left the library in round 6: DESIGN, History.)  train_step.hip's
// l1_rms_kernel runs the same tiles with the previous step's optimizer tail as rider workgroups.
//
// The product: both operands are K-contiguous (W1 [512, F], x [m, F]), so a lane's 16-byte read gives four consecutive k of one row
// and the four MFMAs that consume it contract over the SAME permuted k set on both sides -- no transpose anywhere.  Operands
// go global -> LDS by LDS-DMA in whole 256-byte row segments (global_load_lds_dwordx4: 4 rows x 256 B per wave-instruction;
// fragment-shaped loads straight to registers would be 16 rows x 64 B per instruction and, with no sharing between the four
// waves, 48 B/clk/CU through the vector-memory path), STAGES = 3 K-chunks of 64 resident (73 728 bytes of LDS), the DMAs two chunks
// ahead of their use, issued by four dedicated loader waves, ONE raw s_barrier per chunk
// placed so that no read waits on it: the fragments of K-step t + 1 are read into a second register set while the eight MFMAs
// of step t run, and the barrier that publishes chunk c + 1 sits in front of the LAST step of chunk c.  The LDS image is
// lane-linear (the DMA's rule), so the bank swizzle is applied to the SOURCE address: 16-byte slot s of row r is fetched from
// slot s ^ (r & 15), and read back with the same XOR -- every ds_read_b128 lane group then covers 16 distinct slots.
// fp32 MFMA runs at the vector rate (64 FLOP/clk/SIMD): 2048 x 32 = 65 536 matrix-pipe cycles per wave, 27.4 us at the
// 2.39 GHz the chip holds under this stream (profiles/r04_mfma_clock.txt); everything else has to hide in that shadow.
#include <stdlib.h>
#include "dev_env.h"
#include <string.h>

#include "l1_device.h"

namespace {

using namespace l1_dev;

template <int DBG = 0>
__global__ __launch_bounds__(THREADS) void l1_fwd_kernel(L1Args a)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char l1_smem[];
    l1_fwd_body<DBG>(a, (int)blockIdx.x, l1_smem);
}

static int raise_lds_limit()
{
    static bool attr_set[64] = {};
    int dev = 0;
    IDL_HIP_TRY(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        IDL_HIP_TRY(hipFuncSetAttribute((const void *)l1_fwd_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        IDL_HIP_TRY(hipFuncSetAttribute((const void *)l1_fwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        IDL_HIP_TRY(hipFuncSetAttribute((const void *)l1_fwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        IDL_HIP_TRY(hipFuncSetAttribute((const void *)l1_fwd_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set[dev] = true;
    }
    return IDL_OK;
}

}  // namespace

extern "C" {

int idl_l1_fwd_supported(int m, int n_hidden, int n_in)
{
    return (n_hidden == H1 && m >= TR && (m % TR) == 0 && n_in >= STAGES * KC && (n_in % KC) == 0 && (int64_t)m * n_in < (1ll << 29) &&
            (int64_t)n_hidden * n_in < (1ll << 29)) ? 1 : 0;
}

int idl_l1_fwd(const float *W1, const float *x, int m, int n_in, float *a1_t, void *stream)
{
    IDL_REQUIRE(W1 && x && a1_t && idl_l1_fwd_supported(m, H1, n_in), "l1_fwd: Linear(n_in, 512), m % 32 == 0, n_in % 64 == 0, n_in >= 192");
    IDL_REQUIRE((((uintptr_t)W1 | (uintptr_t)x | (uintptr_t)a1_t) & 15u) == 0, "l1_fwd: buffers must be 16-byte aligned");
    IDL_REQUIRE(idl::take_plan() == nullptr, "l1_fwd: not a recordable launch");
    if (const int rc = raise_lds_limit(); rc != IDL_OK) return rc;
    const int n_tiles = (H1 / TH) * (m / TR);
    const L1Args a{W1, x, a1_t, m, n_in, n_tiles, 0};
    const dim3 grid((unsigned)n_tiles), block(THREADS);
    static const int dbg = [] { const char *e = idl::dev_env("l1_debug"); return e != nullptr ? atoi(e) : 0; }();
    if (dbg == 1) hipLaunchKernelGGL((l1_fwd_kernel<1>), grid, block, LDS_BYTES, (hipStream_t)stream, a);
    else if (dbg == 2) hipLaunchKernelGGL((l1_fwd_kernel<2>), grid, block, LDS_BYTES, (hipStream_t)stream, a);
    else if (dbg == 3) hipLaunchKernelGGL((l1_fwd_kernel<3>), grid, block, LDS_BYTES, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(l1_fwd_kernel<0>, grid, block, LDS_BYTES, (hipStream_t)stream, a);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

}  // extern "C"
