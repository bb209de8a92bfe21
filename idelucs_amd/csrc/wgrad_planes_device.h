// wgrad_planes_device.h (the device side; wgrad_planes.hip: the launch on its own; train_step.hip: with the optimizer tail) -- the weight gradient of Linear(F,512)
// with RMSprop in its epilogue on the fp16 MATRIX CORES, BOTH operands read as the two fp16 planes their producers wrote (planes.h): the batch's from the
// workgroups that assembled it, dr1's from mid_bwd (train_step.hip).  idl_wgrad_rmsprop_xplanes / idl_wgrad_xplanes_rms: the last launch of the step's
// two-plane form (IDELUCS_PLANES=1, the default).
//
// Reference: Linear(F,512).weight.grad = dy^T x (torch autograd, idelucs/models.py:131) and optimizer.step() (models.py:132, RMSprop(lr, weight_decay=0.01),
// models.py:88), as wgrad_device.h states them.  The product is dy0 x0 + dy0 x1 + dy1 x0 with fp32 accumulators (22 significand bits a factor: fp32-grade
// against a float64 product, tests/test_gpu_planes.py).
//
// A 64 x 128 tile over the whole contraction: four LOADER waves bring both operands into LDS by LDS-DMA (a loader's twelve 1 KiB instructions a chunk of
// 64 rows, everything but the chunk's base formed once; three stages of 48 KB, two chunks in flight), four COMPUTING waves read their operands with
// ds_read_b64_tr_b16 -- both lie with the contraction index as the slow one -- and an epilogue that turns the tile around through LDS, applies RMSprop from
// the accumulators (W / square_avg requested before the first product) and writes the updated W1 also as planes for the next step's layer-1 product.
// History (DESIGN): round 5's form took dy in fp32 and split it in the loader waves behind every chunk's barrier -- a request -> wait -> split -> deposit chain
// of 7.4 of the loop's 17 us, with a register ring whose hazard (a compiler copy in front of its wait) only cold caches showed; round 6 moved the split to
// dy's producer and deleted that body, its ring and the ring's guard.
// WHERE IT STANDS (MI355X, m = 1024, 512 x 4096): 24.5-25 us in the step (round 5: 28.8-29.2), the gradient alone 17 us in tools/bench_planes.py's harness; LDS is
// the bound now: the computing waves read 24 KB a K-step (128 B/clk) beside 64 B/clk of DMA writes, of the 256 B/clk the array gives reads.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "planes.h"
#include "wgrad_device.h"

namespace wgp_dev {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 64, TN = 128, NT = 512;                   // 4 computing + 4 loader waves
constexpr int ROWB = 256;                                    // bytes of a k-row of a plane of x in LDS
constexpr int LDS_BYTES = 147456;                            // three stages (dpl::STAGE); + 16 bytes for the tail's words behind them

struct XpArgs {
    const uint16_t *dyh, *dyl;             // dy's two planes [m][n_out] as its producer wrote them (mid_bwd)
    int *dy_scale;                         // the words of their scale (planes.h DR1_WORDS: [0] this step's exponent, read; [1] the next step's, written here)
    const uint16_t *xh, *xl;               // the batch's planes [m][ldx]
    float *grad, *W, *V;
    uint16_t *wh, *wl;                     // W's planes (written) or NULL
    int *over;
    const float *hyper;
    int m, n_out, n_in, ldx, tiles_m, tiles;
    int dbg;                               // diagnostics (IDELUCS_DEV=wgp_dbg; wrong results): 8 no epilogue
};

__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }      // cdna_hip_programming.md T10 (b)

__device__ __forceinline__ void dma16(uint32_t voff, const void *sbase, uint32_t lds_byte)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
}

// dplanes_body.  dy's producer (mid_bwd, train_step.hip) writes dr1 as two fp16 planes with a scale it takes from the previous step's largest |dr1| and
// leaves the exponent in a word this kernel reads in its epilogue; the first loader wave of workgroup 0 derives the NEXT step's exponent from the
// producer's maxima when its last chunk is in.
// LDS images.  x as before: 256-byte rows, 16-byte slots XORed with swz(row) (T10 (b)).  dy: its 64 columns are 128 bytes a row; eight rows form a
// 1 KiB group of two 8-row x 32-column subtiles (T10 (a)): off(row, c8) = 1024 (row / 8) + 512 (c8 / 4) + 64 (row % 8) + 16 ((c8 % 4) ^ ((row / 4) % 4))
// -- a DMA instruction fills one group, and the 32 lanes of a transposed read's half cover 256 consecutive bytes.
// A K-step -- six MFMAs with the next step's twelve transposed reads between them and the wait for those at the end -- is ONE asm statement on
// registers the compiler does not allocate (below): no register a read targets is ever a compiler-visible value in flight.
namespace dpl {
constexpr int KC2 = 64, NS = 3;
constexpr int DPLANE = KC2 * 128, XPLANE = KC2 * 256;          // 8 192, 16 384
constexpr int DSTAGE2 = 2 * DPLANE, STAGE = DSTAGE2 + 2 * XPLANE;   // 16 384, 49 152
constexpr int PERW = 12;                                     // DMA instructions a loader wave issues per chunk: 4 of dy + 8 of x
static_assert(NS * STAGE == LDS_BYTES, "three stages");
}  // namespace dpl

template <class Tail>
__device__ __forceinline__ void dplanes_body(const XpArgs &a, unsigned char *smem, Tail tail)
{
    using namespace dpl;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    int tile = bid;
    if ((a.tiles & 7) == 0) tile = (bid & 7) * (a.tiles >> 3) + (bid >> 3);
    const int h0 = (tile % a.tiles_m) * TM, f0 = (tile / a.tiles_m) * TN;
    const int nc = a.m / KC2;                                // >= 3 (the launcher)
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    if (tid == 0) { int *sw = (int *)(smem + LDS_BYTES); sw[0] = 0; sw[1] = 0; sw[2] = 0; sw[3] = 0; }      // (sw[3]: the tail's meeting point)
    if (wv >= 4) {
        // ================= a loader: plane lw / 2 of both operands, half lw % 2 of a chunk's rows
        const int lw = wv - 4, pl = lw >> 1, half = lw & 1;
        const uint16_t *const bd = pl ? a.dyl : a.dyh, *const bx = pl ? a.xl : a.xh;
        uint32_t vd[4], vx[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {                        // dy: group 4 half + i; LDS byte 16 lane of the group = (subtile lane / 32, row (lane / 4) % 8, slot lane % 4)
            const int r = (lane >> 2) & 7, row = 8 * (4 * half + i) + r, c8 = 4 * (lane >> 5) + ((lane & 3) ^ ((row >> 2) & 3));
            vd[i] = (uint32_t)((row * a.n_out + h0 + 8 * c8) * 2);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {                        // x: rows 4 (8 half + i) .. + 3
            const int row = 4 * (8 * half + i) + (lane >> 4), src = (lane & 15) ^ swz(row);
            vx[i] = (uint32_t)((row * a.ldx + f0 + src * 8) * 2);
        }
        const uint32_t ld = (uint32_t)(pl * DPLANE + 4 * half * 1024), lx = (uint32_t)(DSTAGE2 + pl * XPLANE + 8 * half * 1024);
        const int64_t cd = (int64_t)KC2 * a.n_out * 2, cx = (int64_t)KC2 * a.ldx * 2;
        auto issue = [&](int c) {
            const uint32_t st = lds0 + (uint32_t)((c % NS) * STAGE);
            const char *pd = (const char *)bd + c * cd, *px = (const char *)bx + c * cx;
#pragma unroll
            for (int i = 0; i < 4; ++i) dma16(vd[i], pd, st + ld + (uint32_t)(i * 1024));
#pragma unroll
            for (int i = 0; i < 8; ++i) dma16(vx[i], px, st + lx + (uint32_t)(i * 1024));
        };
        issue(0); issue(1); issue(2);                        // nc >= 3
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * PERW) : "memory");
        __builtin_amdgcn_s_barrier();                        // B_0: chunk 0 is in LDS
        for (int i = 0; i < nc; ++i) {                       // B_{i + 1}: chunk i + 1 readable, chunk i's stage free
            if (i + 2 < nc) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PERW) : "memory");      // (chunk i + 2 may still be in flight)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (i + 3 < nc) issue(i + 3);
        }
        if (bid == 0 && wv == 4) {
            // the exponent of the NEXT step's planes of dy, from the largest |dy| its producer's workgroups saw in this step (their 64 words): the
            // producer of the next step starts with one scalar load instead of a reduction
            const unsigned int w = (unsigned int)a.dy_scale[4 + lane];
            unsigned int mx = w;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const unsigned int y = (unsigned int)__shfl_xor((int)mx, o, 64); mx = mx > y ? mx : y; }
            int k = idl_planes::DR1_K_FIRST;
            const float mxf = __uint_as_float(mx);
            if (mxf > 0.f && mxf < 3.0e38f) {
                int e;
                (void)frexpf(mxf, &e);
                k = idl_planes::DR1_K_TARGET - e;
                k = k < -100 ? -100 : (k > 100 ? 100 : k);
            }
            if (lane == 0) a.dy_scale[1] = k;
        }
        tail(tid - 256);
        return;
    }
    // ================= a computing wave: 32 (h) x 64 (f) of the tile
    const int wm = (wv >> 1) * 32, wn = (wv & 1) * 64;
    f32x16 hi[2], lo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) { hi[j][e] = 0.f; lo[j][e] = 0.f; }
    const int q = (lane & 15) >> 2, p = lane & 3, ch = (lane >> 4) & 1, kg = lane >> 5;
    uint32_t bd[2], bx[2][2];                                // the lane's byte offsets inside a stage: [h] for dy, [j][h] for x
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = 8 * kg + 4 * h + q;
        const int c8 = (wm >> 3) + 2 * ch + (p >> 1);
        bd[h] = (uint32_t)(1024 * (row >> 3) + 512 * (c8 >> 2) + 64 * (row & 7) + 16 * ((c8 & 3) ^ ((row >> 2) & 3)) + 8 * (p & 1));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int slot = ((wn + 32 * j + 16 * ch + 4 * p) >> 3) ^ swz(row);
            bx[j][h] = (uint32_t)(DSTAGE2 + row * ROWB + slot * 16 + 8 * (p & 1));
        }
    }
    // THE K-STEP IS ONE ASM STATEMENT ON FIXED REGISTERS: the six MFMAs of step t on operand set v[200:223] (or v[224:247]) with the twelve transposed
    // reads of step t + 1 into the other set between the first four (three a gap: the last two MFMAs cover their latency), and the wait for them at
    // its end.  A transposed read delivers HALF an MFMA operand, so the operands cannot be compiler values without the compiler choosing where the
    // halves meet; they live in registers the compiler never allocates instead -- the kernels that inline this body carry amdgpu_num_vgpr(200): its
    // allocator stops at v199, v[200:247] belong to these statements (each names the set it writes as clobbered; the set it reads it wrote itself).
    // Set layout: + 0 dy0 (h0 h1), + 4 dy1, + 8 x0 (j = 0), + 12 x1 (j = 0), + 16 x0 (j = 1), + 20 x1 (j = 1).  Every accumulator sees its products
    // in the order the builtin form of round 5 used (hi: dy0 x0; lo: dy0 x1, then dy1 x0).
    static_assert(DPLANE == 8192 && XPLANE == 16384 && ROWB == 256, "the immediates below");
#define WGD_STEP_A(KD, KX, BD0, BD1, BX00, BX01, BX10, BX11) \
    asm volatile("v_mfma_f32_32x32x16_f16 %0, v[200:203], v[208:211], %0\n\t" \
                 "ds_read_b64_tr_b16 v[224:225], %4 offset:" #KD "\n\t" \
                 "ds_read_b64_tr_b16 v[226:227], %5 offset:" #KD "\n\t" \
                 "ds_read_b64_tr_b16 v[232:233], %6 offset:" #KX "\n\t" \
                 "v_mfma_f32_32x32x16_f16 %1, v[200:203], v[212:215], %1\n\t" \
                 "ds_read_b64_tr_b16 v[234:235], %7 offset:" #KX "\n\t" \
                 "ds_read_b64_tr_b16 v[236:237], %6 offset:16384+" #KX "\n\t" \
                 "ds_read_b64_tr_b16 v[238:239], %7 offset:16384+" #KX "\n\t" \
                 "v_mfma_f32_32x32x16_f16 %2, v[200:203], v[216:219], %2\n\t" \
                 "ds_read_b64_tr_b16 v[240:241], %8 offset:" #KX "\n\t" \
                 "ds_read_b64_tr_b16 v[242:243], %9 offset:" #KX "\n\t" \
                 "ds_read_b64_tr_b16 v[244:245], %8 offset:16384+" #KX "\n\t" \
                 "v_mfma_f32_32x32x16_f16 %3, v[200:203], v[220:223], %3\n\t" \
                 "ds_read_b64_tr_b16 v[246:247], %9 offset:16384+" #KX "\n\t" \
                 "ds_read_b64_tr_b16 v[228:229], %4 offset:8192+" #KD "\n\t" \
                 "ds_read_b64_tr_b16 v[230:231], %5 offset:8192+" #KD "\n\t" \
                 "v_mfma_f32_32x32x16_f16 %1, v[204:207], v[208:211], %1\n\t" \
                 "v_mfma_f32_32x32x16_f16 %3, v[204:207], v[216:219], %3\n\t" \
                 "s_waitcnt lgkmcnt(0)\n\t" \
                 : "+v"(hi[0]), "+v"(lo[0]), "+v"(hi[1]), "+v"(lo[1]) \
                 : "v"(BD0), "v"(BD1), "v"(BX00), "v"(BX01), "v"(BX10), "v"(BX11) \
                 : "memory", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247")
#define WGD_STEP_B(KD, KX, BD0, BD1, BX00, BX01, BX10, BX11) \
    asm volatile("v_mfma_f32_32x32x16_f16 %0, v[224:227], v[232:235], %0\n\t" \
                 "ds_read_b64_tr_b16 v[200:201], %4 offset:" #KD "\n\t" \
                 "ds_read_b64_tr_b16 v[202:203], %5 offset:" #KD "\n\t" \
                 "ds_read_b64_tr_b16 v[208:209], %6 offset:" #KX "\n\t" \
                 "v_mfma_f32_32x32x16_f16 %1, v[224:227], v[236:239], %1\n\t" \
                 "ds_read_b64_tr_b16 v[210:211], %7 offset:" #KX "\n\t" \
                 "ds_read_b64_tr_b16 v[212:213], %6 offset:16384+" #KX "\n\t" \
                 "ds_read_b64_tr_b16 v[214:215], %7 offset:16384+" #KX "\n\t" \
                 "v_mfma_f32_32x32x16_f16 %2, v[224:227], v[240:243], %2\n\t" \
                 "ds_read_b64_tr_b16 v[216:217], %8 offset:" #KX "\n\t" \
                 "ds_read_b64_tr_b16 v[218:219], %9 offset:" #KX "\n\t" \
                 "ds_read_b64_tr_b16 v[220:221], %8 offset:16384+" #KX "\n\t" \
                 "v_mfma_f32_32x32x16_f16 %3, v[224:227], v[244:247], %3\n\t" \
                 "ds_read_b64_tr_b16 v[222:223], %9 offset:16384+" #KX "\n\t" \
                 "ds_read_b64_tr_b16 v[204:205], %4 offset:8192+" #KD "\n\t" \
                 "ds_read_b64_tr_b16 v[206:207], %5 offset:8192+" #KD "\n\t" \
                 "v_mfma_f32_32x32x16_f16 %1, v[228:231], v[232:235], %1\n\t" \
                 "v_mfma_f32_32x32x16_f16 %3, v[228:231], v[240:243], %3\n\t" \
                 "s_waitcnt lgkmcnt(0)\n\t" \
                 : "+v"(hi[0]), "+v"(lo[0]), "+v"(hi[1]), "+v"(lo[1]) \
                 : "v"(BD0), "v"(BD1), "v"(BX00), "v"(BX01), "v"(BX10), "v"(BX11) \
                 : "memory", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223")
#define WGD_FIRST(BD0, BD1, BX00, BX01, BX10, BX11) \
    asm volatile("ds_read_b64_tr_b16 v[200:201], %0 offset:0\n\t" \
                 "ds_read_b64_tr_b16 v[202:203], %1 offset:0\n\t" \
                 "ds_read_b64_tr_b16 v[204:205], %0 offset:8192\n\t" \
                 "ds_read_b64_tr_b16 v[206:207], %1 offset:8192\n\t" \
                 "ds_read_b64_tr_b16 v[208:209], %2 offset:0\n\t" \
                 "ds_read_b64_tr_b16 v[210:211], %3 offset:0\n\t" \
                 "ds_read_b64_tr_b16 v[212:213], %2 offset:16384\n\t" \
                 "ds_read_b64_tr_b16 v[214:215], %3 offset:16384\n\t" \
                 "ds_read_b64_tr_b16 v[216:217], %4 offset:0\n\t" \
                 "ds_read_b64_tr_b16 v[218:219], %5 offset:0\n\t" \
                 "ds_read_b64_tr_b16 v[220:221], %4 offset:16384\n\t" \
                 "ds_read_b64_tr_b16 v[222:223], %5 offset:16384\n\t" \
                 "s_waitcnt lgkmcnt(0)\n\t" \
                 : : "v"(BD0), "v"(BD1), "v"(BX00), "v"(BX01), "v"(BX10), "v"(BX11) \
                 : "memory", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223")

    // the wave's pieces of W and square_avg (the epilogue's map) are requested before the first product (nothing else of a computing wave goes to memory: nothing waits on them)
    f32x4 w_pre[8], v_pre[8];
    if (a.W != nullptr) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = lane + 64 * u, rl = idx >> 4, c4 = idx & 15;
            const int64_t at = (int64_t)(h0 + wm + rl) * a.n_in + f0 + wn + 4 * c4;
            w_pre[u] = *(const f32x4 *)(a.W + at); v_pre[u] = *(const f32x4 *)(a.V + at);
        }
    }
    __builtin_amdgcn_s_barrier();                            // B_0
    WGD_FIRST(lds0 + bd[0], lds0 + bd[1], lds0 + bx[0][0], lds0 + bx[0][1], lds0 + bx[1][0], lds0 + bx[1][1]);
    for (int i = 0; i < nc; ++i) {
        // K-step ks of a stage: rows 16 ks .. + 15 -- + 2048 ks bytes in dy's image, + 4096 ks in x's
        const uint32_t st = lds0 + (uint32_t)((i % NS) * STAGE);
        const uint32_t d0 = st + bd[0], d1 = st + bd[1], x00 = st + bx[0][0], x01 = st + bx[0][1], x10 = st + bx[1][0], x11 = st + bx[1][1];
        WGD_STEP_A(2048, 4096, d0, d1, x00, x01, x10, x11);
        WGD_STEP_B(4096, 8192, d0, d1, x00, x01, x10, x11);
        WGD_STEP_A(6144, 12288, d0, d1, x00, x01, x10, x11);
        __builtin_amdgcn_s_barrier();                        // B_{i + 1}: this wave's reads of chunk i are in registers; chunk i + 1 is in LDS
        const uint32_t sn = lds0 + (uint32_t)(((i + 1 < nc ? i + 1 : i) % NS) * STAGE);      // (behind the last chunk: stale bytes nobody uses)
        const uint32_t e0 = sn + bd[0], e1 = sn + bd[1], y00 = sn + bx[0][0], y01 = sn + bx[0][1], y10 = sn + bx[1][0], y11 = sn + bx[1][1];
        WGD_STEP_B(0, 0, e0, e1, y00, y01, y10, y11);
    }
#undef WGD_STEP_A
#undef WGD_STEP_B
#undef WGD_FIRST
    // (the compiler does not know the statements above hold MFMAs: the wait states between an MFMA's write and a VALU read of its accumulator are
    //  paid here, once, by hand)
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(hi[0]), "+v"(lo[0]), "+v"(hi[1]), "+v"(lo[1]));
    if (a.dbg & 8) return;
    // ---- epilogue (the wave turns its 32 x 64 block around through LDS: every wave is past the last barrier, the stages are
    // free), with the exponent of dy's scale read from its producer's word
    const int kexp = a.dy_scale[0];
    const float inv = __builtin_ldexpf(1.f, -(kexp + idl_planes::X_EXP));
    wg_dev::Hyper hy{0.f, 0.f, 0.f, 0.f, 0.f};
    if (a.W != nullptr) hy = wg_dev::Hyper{a.hyper[0], a.hyper[1], a.hyper[2], a.hyper[3], a.hyper[4]};
    constexpr int EP = 68;
    float *img = (float *)smem + wv * (32 * EP);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int rl = (e >> 2) * 8 + (lane >> 5) * 4 + (e & 3), cl = 32 * j + (lane & 31);
            img[rl * EP + cl] = (hi[j][e] + lo[j][e]) * inv;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    bool over = false;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int idx = lane + 64 * u, rl = idx >> 4, c4 = idx & 15;
        const f32x4 g4 = *(const f32x4 *)(img + rl * EP + 4 * c4);
        const int64_t at = (int64_t)(h0 + wm + rl) * a.n_in + f0 + wn + 4 * c4;
        if (a.grad != nullptr) *(f32x4 *)(a.grad + at) = g4;
        if (a.W != nullptr) {
            f32x4 w4 = w_pre[u], v4 = v_pre[u];
#pragma unroll
            for (int e = 0; e < 4; ++e) { float w = w4[e], v = v4[e]; wg_dev::rms_update(g4[e], w, v, hy); w4[e] = w; v4[e] = v; }
            *(f32x4 *)(a.W + at) = w4;
            *(f32x4 *)(a.V + at) = v4;
            if (a.wh != nullptr) {
                constexpr float ps = (float)(1 << idl_planes::W_EXP);
                uint2 h, l2;
                over |= idl_planes::split4(w4[0] * ps, w4[1] * ps, w4[2] * ps, w4[3] * ps, h, l2);
                *(uint2 *)(a.wh + at) = h; *(uint2 *)(a.wl + at) = l2;
            }
        }
    }
    if (over) *a.over = 1;
}

}  // namespace wgp_dev
