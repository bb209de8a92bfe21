// l1_fwd.hip -- the layer-1 forward of NetLinear on this package's own fp32 MFMA tiles, with the work that follows it in its epilogue.
//
// Reference: idelucs/PytorchUtils.py:38-45 (Linear(F,512) . ReLU . Dropout(.5) . Linear(512,64)), called for both views of a batch
// at idelucs/models.py:124-125.  Round 3 ran the product W1 x^T on hipBLASLt (32.4 us of a 111.8 us step) and the next launch
// (mid_fwd_kernel) then spent 6.9 of its 11.0 us waiting for the 2 MB that product had just written -- the one fusion a library
// kernel blocks.  Here a workgroup owns a 64 (hidden units) x 32 (batch rows) tile of a1^T = W1 x^T and, from the
// accumulator registers,
//     adds the bias, applies ReLU + Dropout (the same Philox stream as relu_dropout_fwd_kernel / mid_fwd_kernel: counter = float4
//     index of the row-major [m, 512] activation array), stores r1 for the backward,
//     and forms ITS 64 hidden units' share of lat = r1 W2^T (the C/D registers of the first product ARE the A operand of the
//     second: lane (l, q) holds r1[row l][4 q + reg], and a 16x16x4 MFMA taking register `reg` of every lane contracts over
//     {4 q + reg}: B = W2[c][4 q + reg] is one float4 per lane) into lat_part[h-tile][m][64]  (8 partial sums, added up by the head
//     kernel in a fixed order: deterministic, no atomics).
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

template <bool EPILOGUE, int DBG = 0>
__global__ __launch_bounds__(THREADS) void l1_fwd_kernel(L1Args a)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char l1_smem[];
    l1_fwd_body<EPILOGUE, DBG>(a, (int)blockIdx.x, l1_smem);
}

// several voters in one launch: voter blockIdx.y takes its arguments from its plan record (common.h)
static_assert(sizeof(L1Args) + idl::PLAN_PARAMS <= idl::PLAN_BYTES, "L1Args does not fit a plan record");
__global__ __launch_bounds__(THREADS) void l1_fwd_batched_kernel(const unsigned char *__restrict__ plans)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char l1_smem[];
    const L1Args &p = *(const L1Args *)(plans + (size_t)blockIdx.y * idl::PLAN_BYTES + idl::PLAN_PARAMS);
    l1_fwd_body<true, 0>(p, (int)blockIdx.x, l1_smem);
}

static int raise_lds_limit()
{
    static bool attr_set[64] = {};
    int dev = 0;
    IDL_HIP_TRY(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        IDL_HIP_TRY(hipFuncSetAttribute((const void *)l1_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        IDL_HIP_TRY(hipFuncSetAttribute((const void *)l1_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        IDL_HIP_TRY(hipFuncSetAttribute((const void *)l1_fwd_batched_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        IDL_HIP_TRY(hipFuncSetAttribute((const void *)l1_fwd_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        IDL_HIP_TRY(hipFuncSetAttribute((const void *)l1_fwd_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        IDL_HIP_TRY(hipFuncSetAttribute((const void *)l1_fwd_kernel<false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set[dev] = true;
    }
    return IDL_OK;
}

}  // namespace

int idl::l1_plan_launch(const idl::PlanHead &h, const void *dev_plans, int n_voters, hipStream_t stream)
{
    if (const int rc = raise_lds_limit(); rc != IDL_OK) return rc;
    hipLaunchKernelGGL(l1_fwd_batched_kernel, dim3(h.grid[0], (unsigned)n_voters), dim3(h.block), h.lds, stream, (const unsigned char *)dev_plans);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

extern "C" {

int idl_l1_fwd_supported(int m, int n_hidden, int n_in)
{
    return (n_hidden == H1 && m >= TR && (m % TR) == 0 && n_in >= STAGES * KC && (n_in % KC) == 0 && (int64_t)m * n_in < (1ll << 29) &&
            (int64_t)n_hidden * n_in < (1ll << 29)) ? 1 : 0;
}

int idl_l1_fwd_parts(void) { return H1 / TH; }

static int l1_launch(const float *W1, const float *x, const float *b1, const float *W2, int m, int n_in, int train, uint64_t seed,
                     const int64_t *ctl, float *r1, int r1_transposed, float *lat_part, const idl_dev::GatherArgs &g, int64_t t0, int64_t t1,
                     void *stream)
{
    IDL_REQUIRE(W1 && x && r1 && idl_l1_fwd_supported(m, H1, n_in), "l1_fwd: Linear(n_in, 512), m % 32 == 0, n_in % 64 == 0, n_in >= 192");
    IDL_REQUIRE((((uintptr_t)W1 | (uintptr_t)x | (uintptr_t)r1 | (uintptr_t)b1 | (uintptr_t)W2 | (uintptr_t)lat_part) & 15u) == 0, "l1_fwd: buffers must be 16-byte aligned");
    const bool epi = lat_part != nullptr;
    IDL_REQUIRE(!epi || (b1 && W2 && ctl), "l1_fwd: the fused epilogue needs b1, W2 and ctl");
    IDL_REQUIRE(epi || r1_transposed, "l1_fwd: the bare product writes the transposed image");
    IDL_REQUIRE(epi || t1 == t0, "l1_fwd: only the form with the epilogue carries riders");
    if (const int rc = raise_lds_limit(); rc != IDL_OK) return rc;
    const int n_tiles = (H1 / TH) * (m / TR);
    const L1Args a{W1, x, b1, W2, r1, lat_part, ctl, seed, m, n_in, train, r1_transposed, n_tiles, (int)t0, (int)t1, g, 0};
    const dim3 grid((unsigned)(n_tiles + (t1 - t0 + RIDER_TILES - 1) / RIDER_TILES)), block(THREADS);
    if (void *plan = idl::take_plan()) {          // recorded, not launched (idl_plan_begin): the batched form has the epilogue
        IDL_REQUIRE(epi, "l1_fwd: only the form with the epilogue can be recorded");
        idl::PlanHead h{};
        h.kind = idl::PLAN_L1_FWD; h.grid[0] = grid.x; h.grid[1] = 1; h.grid[2] = 1; h.block = THREADS; h.lds = LDS_BYTES;
        memcpy(plan, &h, sizeof(h));
        memcpy((unsigned char *)plan + idl::PLAN_PARAMS, &a, sizeof(a));
        return IDL_OK;
    }
    static const int dbg = [] { const char *e = idl::dev_env("l1_debug"); return e != nullptr ? atoi(e) : 0; }();
    if (!epi && dbg == 1) hipLaunchKernelGGL((l1_fwd_kernel<false, 1>), grid, block, LDS_BYTES, (hipStream_t)stream, a);
    else if (!epi && dbg == 2) hipLaunchKernelGGL((l1_fwd_kernel<false, 2>), grid, block, LDS_BYTES, (hipStream_t)stream, a);
    else if (!epi && dbg == 3) hipLaunchKernelGGL((l1_fwd_kernel<false, 3>), grid, block, LDS_BYTES, (hipStream_t)stream, a);
    else if (epi) hipLaunchKernelGGL(l1_fwd_kernel<true>, grid, block, LDS_BYTES, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(l1_fwd_kernel<false>, grid, block, LDS_BYTES, (hipStream_t)stream, a);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

int idl_l1_fwd(const float *W1, const float *x, const float *b1, const float *W2, int m, int n_in, int train, uint64_t seed,
               const int64_t *ctl, float *r1, int r1_transposed, float *lat_part, void *stream)
{
    return l1_launch(W1, x, b1, W2, m, n_in, train, seed, ctl, r1, r1_transposed, lat_part, idl_dev::GatherArgs{}, 0, 0, stream);
}

int idl_l1_fwd_gather(const float *W1, const float *x, const float *b1, const float *W2, int m, int n_in, int train, uint64_t seed,
                      const int64_t *ctl, float *r1, int r1_transposed, float *lat_part,
                      const float *feats, int64_t n, int64_t fdim, int64_t view_stride, const int64_t *pair_idx, const int64_t *base,
                      int64_t base_add, int64_t n_pairs, int64_t batch, const double *mean, const double *scale,
                      const double *inv_scale, float *y, int part, int part_end, int parts, void *stream)
{
    IDL_REQUIRE(parts >= 1 && part >= 0 && part <= part_end && part_end <= parts, "l1_fwd_gather: need 0 <= part <= part_end <= parts");
    idl_dev::GatherArgs g{};
    int64_t t0 = 0, t1 = 0;
    if (feats != nullptr) {                  // (feats == NULL: no batch assembly in this launch)
        IDL_REQUIRE(pair_idx && mean && scale && y && n >= 1 && fdim >= 4 && (fdim & 3) == 0 && batch >= 1 && n_pairs >= 0,
                    "l1_fwd_gather: bad gather arguments (4 | f)");
        g = idl_dev::GatherArgs{feats, n, fdim, view_stride, pair_idx, base, batch, n_pairs, mean, scale, inv_scale, y, base_add};
        const int64_t ng = idl_dev::gather_tiles<RIDER_ROWS>(fdim, batch);
        t0 = ng * part / parts; t1 = ng * part_end / parts;
    }
    return l1_launch(W1, x, b1, W2, m, n_in, train, seed, ctl, r1, r1_transposed, lat_part, g, t0, t1, stream);
}

}  // extern "C"
