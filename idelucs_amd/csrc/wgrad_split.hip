// wgrad_split.hip -- EXPERIMENTAL, not in the step and not reachable from the trainer (idl_wgrad_rmsprop_split; tests/test_gpu_encoder.py,
// tools/bench_wgrad_split.py).  KNOWN HAZARD, kept as the record of the A/B only: the loader waves' request ring is the form that was
// WRONG in wgrad_planes_device.h until round 5's last hours -- asm loads into compiler-visible registers, the wait a branch of three asm
// statements the registers pass through; the compiler may copy such a register in front of the wait, and the copy is right only when
// the load has landed (warm caches: the tests here).  The cure is in wgrad_planes_device.h (fixed registers, fetched behind the wait in one
// asm); this kernel was superseded by it and not converted.  Do not build on it.
//
// The
// weight gradient of Linear(F,512) with RMSprop in its epilogue, as wgrad_device.h's tiles do it, but the product dW = dy^T x on the
// fp16 MATRIX CORES from operands split into two fp16 planes INSIDE the kernel,
//     v 2^k = v0 + v1   (v0 = fp16(v 2^k), v1 = fp16(v 2^k - v0): 22 significand bits),
// with the three products dy0 x0, dy0 x1, dy1 x0 accumulated in fp32 (what is dropped, dy1 x1, is 2^-22 of a product).  Measured
// against a float64 product such a sum is as close as the fp32 tiles' and closer than the fp32 library GEMM's (5.2e-7 of the largest
// entry here, 7.9e-7 the fp32 tiles, 1.3e-6 the library) -- the fp32 pipe rounds after every one of its 1024 additions too.
//
// Nothing around the kernel would change: it reads dy [m][n_out] and x [m][n_in] in fp32 AS THEY LIE (the contraction index m is the
// slow one) and writes W / square_avg in fp32.  Four LOADER waves fetch a chunk of 32 rows (64 columns of dy, 128 of x) with 16-byte
// buffer loads four chunks ahead (a register ring), scale, split and store the planes into LDS; four COMPUTING waves (2 x 2, each
// 32 x 64 of the 64 x 128 tile, the whole contraction) take a lane's operand -- 8 consecutive k of ONE column -- out of two
// ds_read_b64_tr_b16 (the hardware transpose of a 4 x 16 block) and issue v_mfma_f32_32x32x16_f16.  Three chunks are resident; one
// barrier a chunk; the epilogue turns the wave's block around through LDS and works on 16-byte pieces of rows.
//
// WHERE IT STANDS (MI355X, m = 1024, 512 x 4096, a HIP graph of 20 launches; tools/bench_wgrad_split.py): gradient only 32.2 us, with
// the update 38.9 -- the fp32 tiles: 35.2 / 41.0.  A little faster, far from the 22.0 us the same product takes from PRE-SPLIT planes
// brought in by LDS-DMA (csrc/probe_split.hip): the split inside the kernel keeps the loaders on the critical path.  Ablation
// (IDELUCS_WGS_DBG: 4 no LDS reads / MFMAs, 8 no epilogue, 16 a quarter of the chunks, 128 stamps): without reads and MFMAs 26.8, without the
// epilogue 30.9.  What was taken out on the way: the compiler's waits for the request ring (it cannot count the requests in flight
// across the loop's back edge and waited for all but the last five: 37.0 -> 32.2 with inline-asm requests, waits placed by hand
// and the ring's registers passed THROUGH the wait -- without that the first arithmetic on a request's output is moved up to the
// request: NaNs); 1 024 atomic maxima on ONE word (12 us; now a word per loader wave, reduced by the next launch); eight loader waves
// (50.6 us); per-read address arithmetic of the transposed reads (formed once per lane: the k16 step leaves the swizzle alone);
// 32 four-byte stores a lane in the epilogue (10 / 24 us; now 16-byte pieces of rows through LDS).  Tried with the hand-placed waits and
// no better: eight loader waves (33.9 / 39.6), chunks of 64 rows with two stages and eight loaders (32.4 / 38.4), raised priority for
// the loaders (32.7 / 38.5), the split in hand-picked instructions (80 instead of 130 a chunk: the same time).  LDS bank conflicts:
// none (SQ_LDS_BANK_CONFLICT 0).  Stamps of a loader wave (IDELUCS_WGS_DBG=128, shader cycles a chunk): wait for its requests 239, split +
// LDS stores issued 839-940, next requests + stores done 180, barrier 90 -- the phase that issues 80 vector instructions and twelve
// 8-byte LDS stores takes 840 cycles whatever the vector instructions are (760 with six 16-byte stores: a lane's item is now a whole
// 16-byte slot of a plane's row); the kernel's time did not follow -- workgroup 0's loader is not the slowest wave of the launch.
//
// Scales: x by 2^3 (a standardised feature is at most sqrt(N - 1) in size: 8 sqrt(N) < 65 504 up to N = 6.7e7); dy by 2^k with k from
// the PREVIOUS launch's largest |dy| (2^k max ~ 2^12: 16 x headroom, values clamped at +-65 000), kept as tagged words (launch
// number << 32 | float bits, by the launch's parity: a newer launch's tag outranks what a word held, so nothing is ever reset; the
// first launch takes k = 10).  A coarse k is enough: an entry 2^-15 of the largest still has its absolute error below 2^-28 of
// the largest.
#include <stdlib.h>

#include "common.h"
#include "wgrad_device.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4v __attribute__((__vector_size__(4 * sizeof(short))));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 64, TN = 128, KC = 32, STAGES = 3, NT = 512;       // 4 computing + 4 loader waves (8 loaders: 50.6 us against 39.0)
constexpr int NL = NT - 256;                                 // loader threads
constexpr int ROWB = 256;                                    // bytes of a k-row of a plane in LDS (dy uses 128 of them, scattered by the swizzle)
constexpr int PLANE = KC * ROWB;                             // 8 192
constexpr int STAGE = 4 * PLANE;                             // dy0 dy1 x0 x1
constexpr int LDS_BYTES = STAGES * STAGE;                    // 98 304
constexpr int X_EXP = 3;
constexpr int K_FIRST = 10, K_TARGET = 12;
constexpr int STATE_SLOTS = 4096;                            // >= 4 loader waves x the largest grid (1024 workgroups)

struct SplitWgArgs {
    const float *dy, *x;
    float *grad, *W, *V;
    const float *hyper;
    const long long *ctl;                  // ctl[0]: the step counter (NULL: launch number 1)
    unsigned long long *state;             // [2][STATE_SLOTS] tagged maxima of |dy| (launch number << 32 | float bits), by launch parity, a word per loader wave
    int m, n_out, n_in, tiles_m, tiles;
    int dbg;                               // diagnostics (IDELUCS_WGS_DBG; wrong results): 1 no requests, 2 no split (raw stores), 4 no LDS reads / MFMAs
};

__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }      // cdna_hip_programming.md T10 (b)

// four fp32 values -> their two fp16 planes (scaled by sc, clamped into fp16's range), 8 bytes each.  Hand-picked instructions: the
// compiler's form unpacked the high plane again to subtract it (~21 vector instructions per 16 bytes; the loaders' split was 940
// shader cycles of a chunk's 1 455, stamps: IDELUCS_WGS_DBG=128); here v_cvt_pk_f16_f32 (two values an instruction, round to nearest
// even) and v_fma_mix_f32, which reads the fp16 half as it lies: s - x0 in one instruction, exact.
template <bool CLAMP>
__device__ __forceinline__ void split4(const f32x4 v, const float sc, uint2 &p0, uint2 &p1)
{
    float s0 = v[0] * sc, s1 = v[1] * sc, s2 = v[2] * sc, s3 = v[3] * sc;
    if (CLAMP) {
        s0 = __builtin_amdgcn_fmed3f(s0, -65000.f, 65000.f); s1 = __builtin_amdgcn_fmed3f(s1, -65000.f, 65000.f);
        s2 = __builtin_amdgcn_fmed3f(s2, -65000.f, 65000.f); s3 = __builtin_amdgcn_fmed3f(s3, -65000.f, 65000.f);
    }
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

__global__ __launch_bounds__(NT, 1) void wgrad_split_kernel(SplitWgArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    int tile = bid;                                          // (wgrad_device.h: an XCD's tiles share dy and a 512-column panel of x)
    if ((a.tiles & 7) == 0) tile = (bid & 7) * (a.tiles >> 3) + (bid >> 3);
    const int h0 = (tile % a.tiles_m) * TM, f0 = (tile / a.tiles_m) * TN;
    const int nc = (a.dbg & 16) ? 8 : a.m / KC;              // >= 3 (the launcher)
    // the launch's number and the scale of dy
    const unsigned long long t = a.ctl != nullptr ? (unsigned long long)a.ctl[0] + 1ull : 1ull;
    // (every loader wave of the previous launch left its own word -- 1 024 atomic maxima on ONE word cost a launch 12 us -- and every wave
    //  of this launch reduces them for itself: the words whose tag is the previous launch's)
    float mxp = 0.f;
    {
        const unsigned long long *pv = a.state + ((t - 1ull) & 1ull) * STATE_SLOTS;
        const int n_words = 4 * (int)gridDim.x;
        for (int i = lane; i < n_words; i += 64) {
            const unsigned long long w = pv[i];
            if ((w >> 32) == ((t - 1ull) & 0xFFFFFFFFull)) mxp = fmaxf(mxp, __uint_as_float((uint32_t)w));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mxp = fmaxf(mxp, __shfl_xor(mxp, o, 64));
    }
    int kexp = K_FIRST;
    if (mxp > 0.f) {
        const float mx = mxp;
        int e;
        (void)frexpf(mx, &e);                                // mx = f 2^e, 0.5 <= f < 1: mx 2^(K_TARGET - e) in [2^(K_TARGET-1), 2^K_TARGET)
        kexp = K_TARGET - e;
        kexp = kexp < -100 ? -100 : (kexp > 100 ? 100 : kexp);
    }
    if (wv >= 4) {
        // ================= a loader: 6 requests of 16 bytes a chunk (2 of dy, 4 of x), split, 12 stores of 8 bytes
        const int lt = tid - 256;
        const float sc_dy = __builtin_ldexpf(1.f, kexp), sc_x = __builtin_ldexpf(1.f, X_EXP);
        // The requests run PF chunks ahead in a register ring (a chunk is 0.16 us of matrix-pipe time, a request ~1 us away); a slot is a
        // fixed register set: the loop is unrolled over the ring.
        constexpr int PF = 4;
        f32x4 ra[PF][2], rb[PF][4];
        float mx = 0.f;
        // The requests are inline asm with the waits placed by hand: left to the compiler the ring does not survive the loop's back edge
        // (it cannot count the requests in flight from the passes before and waits for all but the last few: `s_waitcnt vmcnt(5)` in front
        // of a deposit whose own requests are 18 back -- every pass then paid a round trip; wgrad_device.h tells the same story).  The
        // loaders issue no other vector-memory instruction inside the loop, and the counter retires in order: a deposit of chunk d waits
        // for `6 x (chunks requested behind d)`.
        // a lane's ITEM is a 16-byte slot of a plane's row = 8 consecutive values = two adjacent 16-byte requests: one of dy, two of x a chunk;
        // the two planes of an item are ONE 16-byte LDS store each (8-byte stores: twelve a chunk instead of six)
        uint32_t va[2], vb[4], la[1], lb[2];                 // global byte offsets inside a chunk; LDS byte offsets inside a stage's plane
        {
            const int row = lt >> 3, sl = lt & 7;
            va[0] = (uint32_t)((row * a.n_out + h0 + 8 * sl) * 4); va[1] = va[0] + 16;
            la[0] = (uint32_t)(row * ROWB + ((sl ^ swz(row)) << 4));
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = lt + NL * u, row = j >> 4, sl = j & 15;
            vb[2 * u] = (uint32_t)((row * a.n_in + f0 + 8 * sl) * 4); vb[2 * u + 1] = vb[2 * u] + 16;
            lb[u] = (uint32_t)(row * ROWB + ((sl ^ swz(row)) << 4));
        }
        const int64_t ca = (int64_t)KC * a.n_out * 4, cb = (int64_t)KC * a.n_in * 4;             // a chunk's bytes of rows
        auto request = [&](int c, int slot) {
            const char *pa = (const char *)a.dy + c * ca, *pb = (const char *)a.x + c * cb;
#pragma unroll
            for (int u = 0; u < 2; ++u) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ra[slot][u]) : "v"(va[u]), "s"(pa) : "memory");
#pragma unroll
            for (int u = 0; u < 4; ++u) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(rb[slot][u]) : "v"(vb[u]), "s"(pb) : "memory");
        };
        // (the slot's registers pass THROUGH the wait: the compiler takes an asm's output for ready at once and would move the first
        //  arithmetic on it up to the request)
#define WGS_WAIT(N, SL) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(ra[SL][0]), "+v"(ra[SL][1]), "+v"(rb[SL][0]), "+v"(rb[SL][1]), "+v"(rb[SL][2]), "+v"(rb[SL][3]) : : "memory")
        auto wait_behind = [&](int chunks, int slot) {       // (uniform) the requests of `chunks` later chunks may still be in flight
            if (chunks >= 3) WGS_WAIT(18, slot);
            else if (chunks == 2) WGS_WAIT(12, slot);
            else if (chunks == 1) WGS_WAIT(6, slot);
            else WGS_WAIT(0, slot);
        };
        auto deposit = [&](int c, int slot) {
            const uint32_t st = (uint32_t)(uintptr_t)smem + (uint32_t)((c % STAGES) * STAGE);
            {
                uint2 a0, a1, b0, b1;
                split4<true>(ra[slot][0], sc_dy, a0, a1);
                split4<true>(ra[slot][1], sc_dy, b0, b1);
#pragma unroll
                for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fmaxf(fabsf(ra[slot][0][e]), fabsf(ra[slot][1][e])));
                *(__attribute__((address_space(3))) u32x4 *)(uintptr_t)(st + la[0]) = u32x4{a0.x, a0.y, b0.x, b0.y};
                *(__attribute__((address_space(3))) u32x4 *)(uintptr_t)(st + PLANE + la[0]) = u32x4{a1.x, a1.y, b1.x, b1.y};
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                uint2 a0, a1, b0, b1;
                split4<false>(rb[slot][2 * u], sc_x, a0, a1);        // (|x| 2^3 < 65 504 for N < 6.7e7 sequences)
                split4<false>(rb[slot][2 * u + 1], sc_x, b0, b1);
                *(__attribute__((address_space(3))) u32x4 *)(uintptr_t)(st + 2 * PLANE + lb[u]) = u32x4{a0.x, a0.y, b0.x, b0.y};
                *(__attribute__((address_space(3))) u32x4 *)(uintptr_t)(st + 3 * PLANE + lb[u]) = u32x4{a1.x, a1.y, b1.x, b1.y};
            }
        };
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the scale's words above: nothing else of this wave is in flight from here on)
#pragma unroll
        for (int sl = 0; sl < PF; ++sl) request(sl, sl);     // nc >= 2 PF (the launcher)
        wait_behind(3, 0); deposit(0, 0); request(PF, 0);
        wait_behind(3, 1); deposit(1, 1); request(PF + 1, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // chunks 0 and 1 are in LDS
        for (int base = 0; base < nc; base += PF) {          // nc % PF == 0 (the launcher)
#pragma unroll
            for (int u = 0; u < PF; ++u) {                   // while the others read chunk base + u: chunk d into the stage chunk d - 3 left
                const int d = base + u + 2, slot = (u + 2) % PF;
                const bool stamp = (a.dbg & 128) && bid == 0 && tid == 256;
                uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
                if (stamp) c0 = __builtin_amdgcn_s_memtime();
                if (d < nc) {
                    // chunks requested behind d: d + 1 .. min(d + PF - 1, nc - 1)
                    wait_behind(nc - 1 - d < PF - 1 ? nc - 1 - d : PF - 1, slot);
                    if (stamp) c1 = __builtin_amdgcn_s_memtime();
                    deposit(d, slot);
                    if (stamp) c2 = __builtin_amdgcn_s_memtime();
                    if (d + PF < nc) request(d + PF, slot);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (stamp) c3 = __builtin_amdgcn_s_memtime();
                __builtin_amdgcn_s_barrier();
                if (stamp && d < nc && d + PF < nc) {
                    c4 = __builtin_amdgcn_s_memtime();
                    unsigned long long *o = a.state + 2 * STATE_SLOTS - 8;       // (the last words of the state: diagnostics only)
                    o[0] += c1 - c0; o[1] += c2 - c1; o[2] += c3 - c2; o[3] += c4 - c3; o[4] += 1;
                }
            }
        }
#undef WGS_WAIT
        // this launch's largest |dy| (of the tile's column block; every tile row of workgroups sees all rows) for the next launch
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)); mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 8, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 4, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64)); mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        if (lane == 0) a.state[(t & 1ull) * STATE_SLOTS + 4 * bid + (wv - 4)] = ((t & 0xFFFFFFFFull) << 32) | (unsigned long long)__float_as_uint(mx);
        return;
    }
    // ================= a computing wave: 32 (h) x 64 (f) of the tile
    const int wm = (wv >> 1) * 32, wn = (wv & 1) * 64;
    f32x16 hi[2], lo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) { hi[j][e] = 0.f; lo[j][e] = 0.f; }
    // the transposed read of lane l (probe_split.hip): 16-lane group l / 16 -> column half (l / 16) & 1, k group l / 32; inside the group
    // lane 4 q + p supplies k-row q, columns 4 p .. 4 p + 3, and receives column l % 16, rows 0 .. 3.  A lane's LDS address for block
    // column col0 and k-row 8 kg + 4 h + q is formed ONCE: the k16 step adds 16 rows, which leaves the swizzle (row & 3, (row >> 2) & 3)
    // alone, so step, plane and stage are constant offsets (the instruction's offset field / one scalar add) -- formed per read, the
    // addresses were ~200 vector instructions a chunk beside 12 MFMAs.
    const int q = (lane & 15) >> 2, p = lane & 3, ch = (lane >> 4) & 1, kg = lane >> 5;
    auto lane_base = [&](int col0, int h) {
        const int row = 8 * kg + 4 * h + q;
        const int slot = ((col0 + 16 * ch + 4 * p) >> 3) ^ swz(row);
        return (uint32_t)(row * ROWB + slot * 16 + 8 * (p & 1));
    };
    uint32_t ba[2], bb[2][2];                                // [h] for the wave's dy block; [block][h] for its two x blocks
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        ba[h] = (uint32_t)(uintptr_t)smem + lane_base(wm, h);
        bb[0][h] = (uint32_t)(uintptr_t)smem + 2 * PLANE + lane_base(wn, h);
        bb[1][h] = (uint32_t)(uintptr_t)smem + 2 * PLANE + lane_base(wn + 32, h);
    }
    auto tr8 = [&](const uint32_t (&base)[2], uint32_t off) {             // 8 consecutive k of the lane's column: two transposed reads
        f16x8 out;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const s16x4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v *)(uintptr_t)(base[h] + off));
            const f16x4 f = __builtin_bit_cast(f16x4, v);
            out[4 * h] = f[0]; out[4 * h + 1] = f[1]; out[4 * h + 2] = f[2]; out[4 * h + 3] = f[3];
        }
        return out;
    };
    __builtin_amdgcn_s_barrier();
    for (int i = 0; i < nc; ++i) {
        const uint32_t so = (uint32_t)((i % STAGES) * STAGE);
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            if (a.dbg & 4) continue;
            const uint32_t o = so + (uint32_t)(ks * 16 * ROWB);
            const f16x8 a0 = tr8(ba, o), a1 = tr8(ba, o + PLANE);
            f16x8 b0[2], b1[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) { b0[j] = tr8(bb[j], o); b1[j] = tr8(bb[j], o + PLANE); }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                hi[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0[j], hi[j], 0, 0, 0);
                lo[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1[j], lo[j], 0, 0, 0);
                lo[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0[j], lo[j], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (a.dbg & 8) return;
    // ---- epilogue.  C/D layout of 32x32: lane l, register e -> row (e / 4) * 8 + (l / 32) * 4 + e % 4, column l % 32: written out of the
    // registers that is 32 four-byte stores a lane (10 us of store issue for the gradient alone, 24 with W and square_avg).  The wave
    // turns its 32 x 64 block around through LDS (the stages are free: every wave is past the last barrier) and works on 16-byte
    // pieces of rows: 8 a lane, whole 256-byte row segments per quarter wave.
    const float inv = __builtin_ldexpf(1.f, -(kexp + X_EXP));
    wg_dev::Hyper hy{0.f, 0.f, 0.f, 0.f, 0.f};
    if (a.W != nullptr) hy = wg_dev::Hyper{a.hyper[0], a.hyper[1], a.hyper[2], a.hyper[3], a.hyper[4]};
    constexpr int EP = 68;                                   // floats of a row in the wave's LDS image (64 + 4: rows 4 apart do not meet in a bank)
    float *img = (float *)smem + wv * (32 * EP);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int rl = (e >> 2) * 8 + (lane >> 5) * 4 + (e & 3), cl = 32 * j + (lane & 31);
            img[rl * EP + cl] = (hi[j][e] + lo[j][e]) * inv;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the wave's own image: no barrier)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int idx = lane + 64 * u, rl = idx >> 4, c4 = idx & 15;
        const f32x4 g4 = *(const f32x4 *)(img + rl * EP + 4 * c4);
        const int64_t at = (int64_t)(h0 + wm + rl) * a.n_in + f0 + wn + 4 * c4;
        if (a.grad != nullptr) *(f32x4 *)(a.grad + at) = g4;
        if (a.W != nullptr) {
            f32x4 w4 = *(const f32x4 *)(a.W + at), v4 = *(const f32x4 *)(a.V + at);
#pragma unroll
            for (int e = 0; e < 4; ++e) { float w = w4[e], v = v4[e]; wg_dev::rms_update(g4[e], w, v, hy); w4[e] = w; v4[e] = v; }
            *(f32x4 *)(a.W + at) = w4;
            *(f32x4 *)(a.V + at) = v4;
        }
    }
}

}  // namespace

extern "C" int idl_wgrad_split_state_words(void) { return 2 * STATE_SLOTS; }

extern "C" int idl_wgrad_rmsprop_split(const float *dy, const float *x, int m, int n_out, int n_in, float *grad, float *W, float *square_avg,
                                       const float *hyper, const long long *ctl, unsigned long long *state, void *stream)
{
    IDL_REQUIRE(dy && x && state && m % (4 * KC) == 0 && m / KC >= 8 && n_out % TM == 0 && n_in % TN == 0, "wgrad_split: m % 128 == 0, m >= 256, n_out % 64 == 0, n_in % 128 == 0");
    IDL_REQUIRE((W != nullptr) == (square_avg != nullptr) && (W != nullptr || grad != nullptr), "wgrad_split: give W and square_avg (fused update) and/or grad");
    IDL_REQUIRE(W == nullptr || hyper != nullptr, "wgrad_split: hyper is needed for the fused update");
    IDL_REQUIRE((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)grad | (uintptr_t)W | (uintptr_t)square_avg) & 15u) == 0 && (((uintptr_t)state) & 7u) == 0, "wgrad_split: buffers 16-byte aligned, state 8-byte");
    IDL_REQUIRE((int64_t)m * n_in < (1ll << 29) && (int64_t)n_out * n_in < (1ll << 29), "wgrad_split: operands beyond 2^31 bytes");
    IDL_REQUIRE((n_out / TM) * (n_in / TN) * 4 <= STATE_SLOTS, "wgrad_split: more than 1024 tiles");
    SplitWgArgs a{};
    a.dy = dy; a.x = x; a.grad = grad; a.W = W; a.V = square_avg; a.hyper = hyper; a.ctl = ctl; a.state = state;
    a.m = m; a.n_out = n_out; a.n_in = n_in;
    a.tiles_m = n_out / TM; a.tiles = a.tiles_m * (n_in / TN);
    a.dbg = getenv("IDELUCS_WGS_DBG") ? atoi(getenv("IDELUCS_WGS_DBG")) : 0;
    static bool attr_set = false;
    if (!attr_set) {
        IDL_HIP_TRY(hipFuncSetAttribute((const void *)wgrad_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL(wgrad_split_kernel, dim3((unsigned)a.tiles), dim3(NT), LDS_BYTES, (hipStream_t)stream, a);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}
