// wgrad_planes_device.h (the device side; wgrad_planes.hip: the launch on its own; train_step.hip: with the optimizer tail) -- the weight gradient of Linear(F,512) with RMSprop in its epilogue on the fp16 MATRIX CORES, the batch read as the
// two fp16 planes its assemblers wrote (planes.h): idl_wgrad_rmsprop_xplanes, the last launch of the step's two-plane form (IDELUCS_PLANES=1).
//
// Reference: Linear(F,512).weight.grad = dy^T x (torch autograd, idelucs/models.py:131) and optimizer.step() (models.py:132,
// RMSprop(lr, weight_decay=0.01), models.py:88), as wgrad_device.h states them.  The product is dy0 x0 + dy0 x1 + dy1 x0 with fp32
// accumulators (22 significand bits a factor: closer to a float64 product than the fp32 tiles', tests/test_gpu_planes.py).
//
// A 64 x 128 tile over the whole contraction, four computing waves reading their operands out of LDS with ds_read_b64_tr_b16 -- both
// operands lie with the contraction index as the slow one -- and an epilogue that turns the tile around through LDS (the shape of this
// round's first kernel of the family, which split BOTH operands in its loader waves: removed, History), with the LARGE operand taken off
// the loaders' hands: x's planes arrive by LDS-DMA (16 instructions of 1 KiB a chunk of
// 32 rows, whole 256-byte row segments, the 16-byte slots swizzled on the source side), three chunks ahead, into a ring of six stages;
// only dy (64 columns, 1 / 3 of the bytes, fp32 from mid_bwd) is still split by the loader waves, 8 values a lane a chunk, with the
// tensor's scale 2^k taken from a recent launch's largest |dy| (2^k max ~ 2^12: 16 x headroom, values clamped at +-65 000), kept as tagged
// words (launch number << 32 | float bits: a newer launch's tag outranks what a word held, nothing is ever reset; the first launch takes
// K_FIRST).  A coarse k is enough: an entry 2^-15 of the largest still has its absolute error below 2^-28 of the largest.  x's planes carry
// 2^3 (a standardised feature is at most sqrt(N - 1): 8 sqrt(N) < 65 504 up to N = 6.7e7 sequences).
// The epilogue also writes the updated W1 as planes for the next step's layer-1 product; W and square_avg are requested before the first
// product (in the step: 100.6 -> 98.6 us).
//
// WHERE IT STANDS (MI355X, m = 1024, 512 x 4096, a HIP graph of 20 launches, tools/bench_planes.py): 27.3 us with the update and W1's planes
// (the fp32 tiles: 41-42), the gradient alone 20.8; in the step 27.9 us against 36.5-37.  (30.6 / 23.8 and 30.1 in the step before the scale's
// words were cut from one per loader wave, read by every wave in a plain loop -- 16 dependent round trips, 2.7 us at the head of every launch --
// to one per workgroup, read by one idle computing wave with all its requests in flight while the loaders' first requests are out.)  The loop is a chain of per-chunk latencies, not
// a throughput limit (ablations with requests, MFMAs, LDS reads and the deposit switched off one by one, us of the gradient-only
// launch): everything 24.9; no epilogue 23.2; no MFMAs 21.4; no DMA (four dword requests in their place) 21.6; neither 21.1; no
// requests at all 14.2; no LDS reads either 9.0 (of which ~4.6 is this harness's launch) -- per chunk (0.65 us): the barrier round
// ~0.14 us, the LDS reads behind their waits ~0.16, a chunk's six requests ~0.2, MFMAs 0.1 of 0.16 hidden.  Tried and no better:
// "touches" (one dword of every cache line of the chunk nine further on, dropped into spare LDS by the computing waves, so that the
// loaders' requests would hit L2): 26.9 against 25.7, the step 100.2 against 98.6 -- the requests are not waiting for first touches.
// (THE trap on the way, silent with warm caches: a request whose target is a register must keep that register out of the compiler's
// hands until it lands.  Passing the registers THROUGH the wait that covers them ("+v") orders the asm statements, not the copies the
// compiler makes of the values: it copied the dy ring in front of its waits.  The loaders' ring now lives in fixed registers the compiler
// never sees in flight -- below; the computing waves' LDS reads keep the pass-through form with straight-line waits, their ISA holds no
// copy between a read and its wait, and tests/test_build_resources.py + the cold-cache tests of tests/test_gpu_planes.py watch over both.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "planes.h"
#include "wgrad_device.h"

namespace wgp_dev {


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4v __attribute__((__vector_size__(4 * sizeof(short))));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 64, TN = 128, KC = 32, NT = 512;          // 4 computing + 4 loader waves
constexpr int ROWB = 256;                                    // bytes of a k-row of a plane in LDS (dy uses 128 of them, scattered by the swizzle)
constexpr int PLANE = KC * ROWB;                             // 8 192
constexpr int DSTAGE = 2 * PLANE, XSTAGE = 2 * PLANE;        // dy0 dy1 | x0 x1
constexpr int PF = 3;                                        // chunks requested ahead
constexpr int ND = 3, NX = PF + 3;                           // stages of dy (deposited from a register ring) and of x (written by the DMA when it is REQUESTED)
constexpr int DY_BYTES = ND * DSTAGE;                        // 49 152
constexpr int LDS_BYTES = DY_BYTES + NX * XSTAGE;            // 147 456
constexpr int K_FIRST = 10, K_TARGET = 12;
constexpr int STATE_SLOTS = 4096;                            // (idl_wgrad_split_state_words() = 2 x 4096 words: room for the 3 arrays below)
constexpr int STATE_ARRAY = 2048;                            // three arrays of a word per workgroup (launch number % 3) inside it: <= 2048 tiles

struct XpArgs {
    const float *dy;                       // [m][n_out]
    const uint16_t *xh, *xl;               // the batch's planes [m][ldx]
    float *grad, *W, *V;
    uint16_t *wh, *wl;                     // W's planes (written) or NULL
    int *over;
    const float *hyper;
    const long long *ctl;
    unsigned long long *state;             // [3][STATE_ARRAY] tagged maxima of |dy| (launch number << 32 | float bits), a word per workgroup, by launch number % 3
    int m, n_out, n_in, ldx, tiles_m, tiles;
    int dbg;                               // diagnostics (IDELUCS_WGP_DBG; wrong results): 8 no epilogue
};

__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }      // cdna_hip_programming.md T10 (b)

__device__ __forceinline__ void dma16(uint32_t voff, const void *sbase, uint32_t lds_byte)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
}

// four fp32 values -> their two fp16 planes (scaled, clamped), 8 bytes each
__device__ __forceinline__ void split4c(const f32x4 v, const float sc, uint2 &p0, uint2 &p1)
{
    const float s0 = __builtin_amdgcn_fmed3f(v[0] * sc, -65000.f, 65000.f), s1 = __builtin_amdgcn_fmed3f(v[1] * sc, -65000.f, 65000.f);
    const float s2 = __builtin_amdgcn_fmed3f(v[2] * sc, -65000.f, 65000.f), s3 = __builtin_amdgcn_fmed3f(v[3] * sc, -65000.f, 65000.f);
    uint32_t h01, h23, l01, l23;
    float r0, r1, r2, r3;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h01) : "v"(s0), "v"(s1));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h23) : "v"(s2), "v"(s3));
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h01), "v"(s0));
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h01), "v"(s1));
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(h23), "v"(s2));
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(h23), "v"(s3));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l01) : "v"(r0), "v"(r1));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l23) : "v"(r2), "v"(r3));
    p0 = uint2{h01, h23};
    p1 = uint2{l01, l23};
}

// tail(tix): what the four loader waves (tix = 0 .. 255) go on with when their last chunk is in -- nothing (the kernel on its own), or the
// step's optimizer tail (train_step.hip: wgrad_xplanes_rms_kernel), which then runs under the computing waves' epilogue.  It must not use
// s_barrier (the computing waves do not come): rmsprop_body's SPIN form meets on an LDS counter, sw[3] below.
template <class Tail>
__device__ __forceinline__ void xplanes_body(const XpArgs &a, unsigned char *smem, Tail tail)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    int tile = bid;                                          // (wgrad_device.h: an XCD's tiles share dy and a 512-column panel of x)
    if ((a.tiles & 7) == 0) tile = (bid & 7) * (a.tiles >> 3) + (bid >> 3);
    const int h0 = (tile % a.tiles_m) * TM, f0 = (tile / a.tiles_m) * TN;
    const int nc = a.m / KC;                                 // >= 6 (the launcher)
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    // the launch's number and the scale of dy
    const unsigned long long t = a.ctl != nullptr ? (unsigned long long)a.ctl[0] + 1ull : 1ull;
    // Launch t writes its words into array t % 3 and reads the arrays of t - 1 and t - 2, which nobody writes while it runs: every workgroup of a
    // launch derives the SAME scale whenever it starts (on a GPU shared with another process not all tiles are resident at once, and a tile
    // that starts late must not see this launch's own words).  The words of t - 1 if there are any, else those of t - 2 (a step of another
    // form in between -- the epoch's partial last batch runs the fp32 tiles and moves the counter on), else the default scale.
    // ONE wave reads them and hands the exponent to the others through LDS: the loaders that split dy and the computing waves that scale the
    // tile back must agree.  It is a COMPUTING wave (idle until the first chunk is in): the loaders have their first requests out before they
    // meet it at the barrier.
    if (a.dbg & 192) {                                       // EXPERIMENT: the x DMA (64) / + dy-sized DMA (128) alone, free-running, three chunks in flight
        if (wv < 4) return;
        const int lw = wv - 4;
        for (int c = 0; c < nc; ++c) {
            const uint32_t xs = lds0 + (uint32_t)((c % NX) * XSTAGE) + DY_BYTES;
            for (int i = 0; i < ((a.dbg & 128) ? 6 : 4); ++i) {
                const int j = lw * 4 + (i & 3), pq = j >> 3, blk = j & 7;
                const int row = blk * 4 + (lane >> 4), slot = lane & 15, src = slot ^ swz(row);
                if (i < 4) dma16((uint32_t)((row * a.ldx + f0 + src * 8) * 2), (const char *)(pq == 0 ? a.xh : a.xl) + (int64_t)c * KC * a.ldx * 2, xs + (uint32_t)(pq * PLANE + blk * 1024));
                else dma16((uint32_t)(((lw * 8 + (i & 1) * 4 + (lane >> 4)) * a.n_out + h0 + (lane & 15) * 4) * 4), (const char *)a.dy + (int64_t)c * KC * a.n_out * 4, lds0 + (uint32_t)((c % ND) * DSTAGE + (lw * 2 + (i & 1)) * 1024));
            }
            if (a.dbg & 128) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    float mxp = 0.f;
    if (wv == 0) {
        // a word per WORKGROUP of a launch (its four loader waves fold their maxima in LDS first), all requests of a lane in flight before the
        // first compare: with a word per loader wave and a plain loop every wave of every launch spent 16 dependent round trips here (2.7 us)
        const int n_words = (int)gridDim.x;
        float mx[2] = {0.f, 0.f};
        int has[2] = {0, 0};
#pragma unroll
        for (int back = 0; back < 2; ++back) {               // back = 0: launch t - 1, 1: launch t - 2
            const uint32_t tag = (uint32_t)t - 1u - (uint32_t)back;
            const unsigned long long *pv = a.state + (tag % 3u) * STATE_ARRAY;
            for (int i0 = 0; i0 < n_words; i0 += 4 * 64) {
                unsigned long long w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int i = i0 + 64 * j + lane; w[j] = pv[i < n_words ? i : n_words - 1]; }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((uint32_t)(w[j] >> 32) == tag && tag != 0u) { mx[back] = fmaxf(mx[back], __uint_as_float((uint32_t)w[j])); has[back] = 1; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mx[0] = fmaxf(mx[0], __shfl_xor(mx[0], o, 64)); mx[1] = fmaxf(mx[1], __shfl_xor(mx[1], o, 64));
            has[0] |= __shfl_xor(has[0], o, 64); has[1] |= __shfl_xor(has[1], o, 64);
        }
        mxp = has[0] ? mx[0] : (has[1] ? mx[1] : 0.f);
        int k = K_FIRST;
        if (mxp > 0.f) {
            int e;
            (void)frexpf(mxp, &e);
            k = K_TARGET - e;
            k = k < -100 ? -100 : (k > 100 ? 100 : k);
        }
        if (lane == 0) { int *sw = (int *)(smem + LDS_BYTES); sw[0] = k; sw[1] = 0; sw[2] = 0; sw[3] = 0; }      // exponent | this launch's maximum | loaders done | the tail's meeting point
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    int kexp = 0;
    if (wv >= 4) {
        // ================= a loader: a chunk is 2 requests of 16 bytes (dy, into the register ring) + 4 DMA instructions (x's planes, into the
        // x stage of the chunk).  All six count on vmcnt and retire in order: a chunk is in when at most 6 x (chunks requested behind it) are out.
        const int lt = tid - 256, lw = wv - 4;
        float mx = 0.f, sc_dy = 1.f;                         // (the scale: set behind the barrier below)
        uint32_t va[2], la;
        {
            const int row = lt >> 3, sl = lt & 7;            // a lane's item of dy: row, 8 consecutive columns = a 16-byte slot of a plane's row
            va[0] = (uint32_t)((row * a.n_out + h0 + 8 * sl) * 4);
            if (a.dbg & 32) va[0] = (uint32_t)((((h0 >> 6) * a.m + row) * 64 + 8 * sl) * 4);      // EXPERIMENT: dy block-major
            va[1] = va[0] + 16;
            la = (uint32_t)(row * ROWB + ((sl ^ swz(row)) << 4));
        }
        uint32_t vx[4], lx[4];                               // x: DMA instruction j = 4 lw + i: plane j / 8, k-rows 4 (j % 8) .. + 3
        const uint16_t *px[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = lw * 4 + i, pq = j >> 3, blk = j & 7;
            const int row = blk * 4 + (lane >> 4), slot = lane & 15, src = slot ^ swz(row);
            vx[i] = (uint32_t)((row * a.ldx + f0 + src * 8) * 2);
            if (a.dbg & 16) vx[i] = (uint32_t)(((((f0 >> 6) + (src >> 3)) * a.m + row) * 64 + (src & 7) * 8) * 2);      // EXPERIMENT: chunk-major x planes
            lx[i] = (uint32_t)(DY_BYTES + pq * PLANE + blk * 1024);
            px[i] = pq == 0 ? a.xh : a.xl;
        }
        int64_t ca = (int64_t)KC * a.n_out * 4, cx = (int64_t)KC * a.ldx * 2;      // a chunk's bytes of rows
        if (a.dbg & 16) cx = (int64_t)KC * 64 * 2;
        if (a.dbg & 32) ca = (int64_t)KC * 64 * 4;
        // THE RING'S REGISTERS ARE FIXED: v[232:255], eight a slot, outside what the compiler allocates (the kernel needs ~200 of its 256; the
        // requests name them as clobbers).  A request whose target is a compiler-visible value is unsafe however its wait is written: the
        // compiler takes the value for available at once and may COPY the register before the wait -- it did, in front of the three-way
        // branch of the waits: `v_mov v[24:31], v[0:7]; s_waitcnt vmcnt(0)` -- and the copy holds whatever the register held: the right
        // data when the request had landed long before (warm caches: every test), garbage when it had not (a 512 MB fill in front of the
        // launch, another process on the GPU: a different result every run).  Here the values enter the compiler's view only through the
        // v_movs BEHIND the wait, inside one asm statement.
        auto request = [&](int c, int slot) {
            const char *pa = (const char *)a.dy + c * ca;
            if (slot == 0)
                asm volatile("global_load_dwordx4 v[232:235], %0, %2\n\tglobal_load_dwordx4 v[236:239], %1, %2" : : "v"(va[0]), "v"(va[1]), "s"(pa)
                             : "memory", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239");
            else if (slot == 1)
                asm volatile("global_load_dwordx4 v[240:243], %0, %2\n\tglobal_load_dwordx4 v[244:247], %1, %2" : : "v"(va[0]), "v"(va[1]), "s"(pa)
                             : "memory", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247");
            else if (slot == 2)
                asm volatile("global_load_dwordx4 v[248:251], %0, %2\n\tglobal_load_dwordx4 v[252:255], %1, %2" : : "v"(va[0]), "v"(va[1]), "s"(pa)
                             : "memory", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
            const uint32_t xs = lds0 + (uint32_t)((c % NX) * XSTAGE);
#pragma unroll
            for (int i = 0; i < 4; ++i) dma16(vx[i], (const char *)px[i] + c * cx, xs + lx[i]);
        };
#define WGP_FETCH(N, B0, B1, B2, B3, B4, B5, B6, B7)                                                                                  \
    asm volatile("s_waitcnt vmcnt(" #N ")\n\tv_mov_b32 %0, " #B0 "\n\tv_mov_b32 %1, " #B1 "\n\tv_mov_b32 %2, " #B2 "\n\tv_mov_b32 %3, " #B3     \
                 "\n\tv_mov_b32 %4, " #B4 "\n\tv_mov_b32 %5, " #B5 "\n\tv_mov_b32 %6, " #B6 "\n\tv_mov_b32 %7, " #B7                           \
                 : "=v"(r0[0]), "=v"(r0[1]), "=v"(r0[2]), "=v"(r0[3]), "=v"(r1[0]), "=v"(r1[1]), "=v"(r1[2]), "=v"(r1[3]) : : "memory")
#define WGP_FETCH_SLOT(N)                                                                                                             \
    do {                                                                                                                              \
        if (slot == 0) WGP_FETCH(N, v232, v233, v234, v235, v236, v237, v238, v239);                                                  \
        else if (slot == 1) WGP_FETCH(N, v240, v241, v242, v243, v244, v245, v246, v247);                                             \
        else WGP_FETCH(N, v248, v249, v250, v251, v252, v253, v254, v255);                                                            \
    } while (0)
        // (uniform) wait until the requests of at most `chunks` later chunks are in flight, then take the slot's eight registers
        auto fetch = [&](int chunks, int slot, f32x4 &r0, f32x4 &r1) {
            if (chunks >= 2) WGP_FETCH_SLOT(12);
            else if (chunks == 1) WGP_FETCH_SLOT(6);
            else WGP_FETCH_SLOT(0);
        };
        auto deposit = [&](int c, int chunks_behind, int slot) {
            f32x4 r0, r1;
            fetch(chunks_behind, slot, r0, r1);
            if (a.dbg & 4) return;
            const uint32_t st = lds0 + (uint32_t)((c % ND) * DSTAGE);
            uint2 a0, a1, b0, b1;
            split4c(r0, sc_dy, a0, a1);
            split4c(r1, sc_dy, b0, b1);
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fmaxf(fabsf(r0[e]), fabsf(r1[e])));
            *(__attribute__((address_space(3))) u32x4 *)(uintptr_t)(st + la) = u32x4{a0.x, a0.y, b0.x, b0.y};
            *(__attribute__((address_space(3))) u32x4 *)(uintptr_t)(st + PLANE + la) = u32x4{a1.x, a1.y, b1.x, b1.y};
        };
#pragma unroll
        for (int sl = 0; sl < PF; ++sl) request(sl, sl);
        __builtin_amdgcn_s_barrier();                        // the exponent is in LDS
        kexp = *(const int *)(smem + LDS_BYTES);
        sc_dy = __builtin_ldexpf(1.f, kexp);
        deposit(0, 2, 0); request(PF, 0);                    // x stage 3: never used yet
        deposit(1, 2, 1); request(PF + 1, 1);                // x stage 4
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // chunks 0 and 1 are in LDS
        for (int base = 0; base < nc; base += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {                   // while the others read chunk i: chunk d = i + 2 into the dy stage chunk i - 1 left,
                const int i = base + u;                      // and the DMA of chunk d + PF = i + 5 into the x stage chunk i - 1 left
                if (i >= nc) break;
                const int d = i + 2, slot = (u + 2) % PF;
                if (d < nc) {
                    // requested so far: up to min(d + PF - 1, nc - 1)
                    deposit(d, nc - 1 - d < PF - 1 ? nc - 1 - d : PF - 1, slot);
                    if (d + PF < nc) request(d + PF, slot);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
#undef WGP_FETCH
#undef WGP_FETCH_SLOT
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)); mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 8, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 4, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64)); mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        if (lane == 0) {                                     // (|dy| >= 0: the bit patterns order as the values do)
            unsigned int *sw = (unsigned int *)(smem + LDS_BYTES);
            atomicMax(&sw[1], __float_as_uint(mx));
            if (atomicAdd(&sw[2], 1u) == 3u)                 // the last of the four loader waves: the workgroup's word
                a.state[((uint32_t)t % 3u) * STATE_ARRAY + bid] = ((t & 0xFFFFFFFFull) << 32) | (unsigned long long)atomicMax(&sw[1], 0u);
        }
        tail(tid - 256);
        return;
    }
    // ================= a computing wave: 32 (h) x 64 (f) of the tile
    __builtin_amdgcn_s_barrier();                            // the exponent is in LDS
    kexp = *(const int *)(smem + LDS_BYTES);
    const int wm = (wv >> 1) * 32, wn = (wv & 1) * 64;
    f32x16 hi[2], lo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) { hi[j][e] = 0.f; lo[j][e] = 0.f; }
    const int q = (lane & 15) >> 2, p = lane & 3, ch = (lane >> 4) & 1, kg = lane >> 5;
    auto lane_base = [&](int col0, int h) {
        const int row = 8 * kg + 4 * h + q;
        const int slot = ((col0 + 16 * ch + 4 * p) >> 3) ^ swz(row);
        return (uint32_t)(row * ROWB + slot * 16 + 8 * (p & 1));
    };
    uint32_t ba[2], bb[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        ba[h] = lds0 + lane_base(wm, h);
        bb[0][h] = lds0 + DY_BYTES + lane_base(wn, h);
        bb[1][h] = lds0 + DY_BYTES + lane_base(wn + 32, h);
    }
    // The operands of K-step t + 1 are read into a second register set while the six MFMAs of step t run, the reads as inline asm with the
    // waits placed by hand and the set passed THROUGH the wait (l1_planes_device.h tells why); a chunk is two steps with the chunk's barrier
    // between them:   read F1 = (i, 1) | MFMA F0 | wait F1 | barrier | read F0 = (i + 1, 0) | MFMA F1
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    struct Set { u32x2 r[12]; };                             // [operand: dy0 dy1 x0(j=0) x1(j=0) x0(j=1) x1(j=1)][h]
#define WGP_TR(DST, ADDR, IMM) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #IMM : "=v"(DST) : "v"(ADDR))
#define WGP_READ_SET(F, AD0, AD1, AX00, AX01, AX10, AX11, K)                                                                         \
    do {                                                                                                                              \
        if (K == 0) {                                                                                                                 \
            WGP_TR(F.r[0], AD0, 0); WGP_TR(F.r[1], AD1, 0); WGP_TR(F.r[2], AD0, 8192); WGP_TR(F.r[3], AD1, 8192);                      \
            WGP_TR(F.r[4], AX00, 0); WGP_TR(F.r[5], AX01, 0); WGP_TR(F.r[6], AX00, 8192); WGP_TR(F.r[7], AX01, 8192);                  \
            WGP_TR(F.r[8], AX10, 0); WGP_TR(F.r[9], AX11, 0); WGP_TR(F.r[10], AX10, 8192); WGP_TR(F.r[11], AX11, 8192);                \
        } else {                                                                                                                      \
            WGP_TR(F.r[0], AD0, 4096); WGP_TR(F.r[1], AD1, 4096); WGP_TR(F.r[2], AD0, 12288); WGP_TR(F.r[3], AD1, 12288);              \
            WGP_TR(F.r[4], AX00, 4096); WGP_TR(F.r[5], AX01, 4096); WGP_TR(F.r[6], AX00, 12288); WGP_TR(F.r[7], AX01, 12288);          \
            WGP_TR(F.r[8], AX10, 4096); WGP_TR(F.r[9], AX11, 4096); WGP_TR(F.r[10], AX10, 12288); WGP_TR(F.r[11], AX11, 12288);        \
        }                                                                                                                             \
    } while (0)
#define WGP_LWAIT(N, F)                                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(F.r[0]), "+v"(F.r[1]), "+v"(F.r[2]), "+v"(F.r[3]), "+v"(F.r[4]), "+v"(F.r[5]),    \
                 "+v"(F.r[6]), "+v"(F.r[7]), "+v"(F.r[8]), "+v"(F.r[9]), "+v"(F.r[10]), "+v"(F.r[11]))
    static_assert(PLANE == 8192 && 16 * ROWB == 4096 && KC == 32, "the immediates and the schedule");
    auto op = [](const u32x2 lo2, const u32x2 hi2) { return __builtin_bit_cast(f16x8, u32x4{lo2.x, lo2.y, hi2.x, hi2.y}); };
    auto mma = [&](const Set &f) {
        const f16x8 a0 = op(f.r[0], f.r[1]), a1 = op(f.r[2], f.r[3]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f16x8 b0 = op(f.r[4 + 4 * j], f.r[5 + 4 * j]), b1 = op(f.r[6 + 4 * j], f.r[7 + 4 * j]);
            hi[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, hi[j], 0, 0, 0);
            lo[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, lo[j], 0, 0, 0);
            lo[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, lo[j], 0, 0, 0);
        }
    };
    // the wave's pieces of W and square_avg (the epilogue's map: 8 x 16 bytes of each a lane) are requested HERE, before the first product: 16 MB a
    // launch that used to be fetched behind the last MFMA (the computing waves have no other memory requests: nothing waits on these)
    f32x4 w_pre[8], v_pre[8];
    if (a.W != nullptr) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = lane + 64 * u, rl = idx >> 4, c4 = idx & 15;
            const int64_t at = (int64_t)(h0 + wm + rl) * a.n_in + f0 + wn + 4 * c4;
            w_pre[u] = *(const f32x4 *)(a.W + at); v_pre[u] = *(const f32x4 *)(a.V + at);
        }
    }
    Set fs0, fs1;
    __builtin_amdgcn_s_barrier();
    {
        const uint32_t d0 = ba[0], d1 = ba[1], x00 = bb[0][0], x01 = bb[0][1], x10 = bb[1][0], x11 = bb[1][1];
        WGP_READ_SET(fs0, d0, d1, x00, x01, x10, x11, 0);
    }
    for (int i = 0; i < nc; ++i) {
        const uint32_t sd = (uint32_t)((i % ND) * DSTAGE), sx = (uint32_t)((i % NX) * XSTAGE);
        {
            const uint32_t d0 = ba[0] + sd, d1 = ba[1] + sd, x00 = bb[0][0] + sx, x01 = bb[0][1] + sx, x10 = bb[1][0] + sx, x11 = bb[1][1] + sx;
            WGP_READ_SET(fs1, d0, d1, x00, x01, x10, x11, 1);
        }
        WGP_LWAIT(12, fs0);
        __builtin_amdgcn_sched_barrier(0);
        mma(fs0);
        __builtin_amdgcn_sched_barrier(0);
        WGP_LWAIT(0, fs1);
        __builtin_amdgcn_s_barrier();
        if (i + 1 < nc) {
            const uint32_t sd1 = (uint32_t)(((i + 1) % ND) * DSTAGE), sx1 = (uint32_t)(((i + 1) % NX) * XSTAGE);
            const uint32_t d0 = ba[0] + sd1, d1 = ba[1] + sd1, x00 = bb[0][0] + sx1, x01 = bb[0][1] + sx1, x10 = bb[1][0] + sx1, x11 = bb[1][1] + sx1;
            WGP_READ_SET(fs0, d0, d1, x00, x01, x10, x11, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma(fs1);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef WGP_TR
#undef WGP_READ_SET
#undef WGP_LWAIT
    if (a.dbg & 8) return;
    // ---- epilogue: the wave turns its 32 x 64 block around through LDS and works on 16-byte pieces of rows
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
