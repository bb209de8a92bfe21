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
    const float *dy;                       // [m][n_out] fp32 (split by the loader waves: xplanes_body), or NULL with
    const uint16_t *dyh, *dyl;             // ... dy's two planes [m][n_out] as its producer wrote them (mid_bwd; dplanes_body) and
    int *dy_scale;                         // ... the words of their scale (planes.h DR1_WORDS: [0] this step's exponent, read; [1] the next step's, written here)
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
            va[0] = (uint32_t)((row * a.n_out + h0 + 8 * sl) * 4); va[1] = va[0] + 16;
            la = (uint32_t)(row * ROWB + ((sl ^ swz(row)) << 4));
        }
        uint32_t vx[4], lx[4];                               // x: DMA instruction j = 4 lw + i: plane j / 8, k-rows 4 (j % 8) .. + 3
        const uint16_t *px[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = lw * 4 + i, pq = j >> 3, blk = j & 7;
            const int row = blk * 4 + (lane >> 4), slot = lane & 15, src = slot ^ swz(row);
            vx[i] = (uint32_t)((row * a.ldx + f0 + src * 8) * 2);
            lx[i] = (uint32_t)(DY_BYTES + pq * PLANE + blk * 1024);
            px[i] = pq == 0 ? a.xh : a.xl;
        }
        const int64_t ca = (int64_t)KC * a.n_out * 4, cx = (int64_t)KC * a.ldx * 2;      // a chunk's bytes of rows
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



// ============================================================================================================================
// dplanes_body (round 6): the same tile with dy ALSO arriving as planes by LDS-DMA -- its producer (mid_bwd, train_step.hip) writes dr1 as two
// fp16 planes with a scale it takes from the previous steps' largest |dr1| (the tagged words xplanes_body's loaders kept now belong to mid_bwd's
// workgroups) and leaves the exponent in a word this kernel reads in its epilogue.  What that removes from the loop, measured in round 5's
// ablation of xplanes_body (its header): the loaders' request -> wait -> split -> deposit chain behind every chunk's barrier (7.4 of the loop's
// ~17 us), the fixed-register ring with its hazard, the scan of the scale words at the head of every launch.  What it changes besides: chunks of
// 64 rows (16 barriers a launch instead of 32; a stage = 16 KB of dy + 32 KB of x, three stages, two chunks in flight: 96 KB a CU), a loader
// wave's twelve 1 KiB DMA instructions a chunk with everything but the chunk's base formed once.
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
static_assert(NS * STAGE == LDS_BYTES, "the stages fill what xplanes_body's rings do");
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
    // in xplanes_body's order (hi: dy0 x0; lo: dy0 x1, then dy1 x0).
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

    // the wave's pieces of W and square_avg (the epilogue's map) are requested before the first product, as in xplanes_body
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
    // ---- epilogue: as xplanes_body's (the wave turns its 32 x 64 block around through LDS: every wave is past the last barrier, the stages are
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
