// l1_device.h -- the device side of the layer-1 forward tiles (l1_fwd.hip has the story): shared with train_step.hip, whose
// l1_rms_kernel runs these tiles with the PREVIOUS step's optimizer tail as rider workgroups beside them (round 5).
#pragma once
#include "common.h"
#include "philox_device.h"
#include "scaler_device.h"

namespace l1_dev {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int TH = 64, TR = 32;            // workgroup tile: hidden units x batch rows
constexpr int KC = 64;                     // floats of K per chunk = one 256-byte row segment = 16 slots of 16 bytes
#ifndef L1_STAGES
#define L1_STAGES 3
#endif
constexpr int STAGES = L1_STAGES;                  // chunks resident: the DMAs run STAGES - 1 chunks (3 072 matrix-pipe cycles) ahead of their use
constexpr int A_BYTES = TH * KC * 4, B_BYTES = TR * KC * 4, STAGE_BYTES = A_BYTES + B_BYTES;
constexpr int LDS_BYTES = STAGES * STAGE_BYTES;      // 73 728: two workgroups per CU (a tile workgroup and a rider, below)
constexpr int H1 = 512, H2 = 64;
constexpr int NP = 4;                      // loader waves (24 DMA instructions per chunk: 16 x 4 rows of A, 8 x 4 rows of B)
constexpr int THREADS = 256 + 64 * NP;
static_assert(16 % NP == 0 && 8 % NP == 0, "a loader takes whole instructions of A and of B");

// (Round 4's form with bias / ReLU / Dropout / the K-split of Linear(512, 64) in these tiles' epilogue and batch-assembly riders behind them -- measured 114.7 us a
//  step against 111.8, kept opt-in for two rounds -- left the library in round 6: the tiles are the fp32 form's plain product a1^T = W1 x^T.)
struct L1Args {
    const float *W1, *x;        // [512, K], [m, K]
    float *r1;                  // a1^T = W1 x^T: [512, m]
    int m, K;
    int n_tiles;
    int prio;                   // > 0: the computing waves raise their priority to it (riders share the CU: train_step.hip's l1_rms_kernel)
};

// LDS-DMA: 64 lanes x 16 bytes -> LDS bytes [lds_byte, lds_byte + 1024), lane-linear; source = sbase + voff (per lane)
__device__ __forceinline__ void dma16(uint32_t voff, const void *sbase, uint32_t lds_byte)
{
    uint32_t keep;                       // (M0 is the compiler's: saved and restored, as vectorise.hip's dma16 does)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

struct Frag { f32x4_t a0, a1, b; };

// ds_read_b128 from LDS byte address `addr` + a compile-time offset (kept in the instruction's offset field: no VALU)
__device__ __forceinline__ f32x4_t lds_read16(uint32_t addr, uint32_t imm)
{
    return *(const __attribute__((address_space(3))) f32x4_t *)(uintptr_t)(addr + imm);
}

// one K-step: the four k-elements of the fragments' float4 -> eight MFMAs (256 matrix-pipe cycles)
__device__ __forceinline__ void mma_step(const Frag &f, f32x4_t &c0, f32x4_t &c1)
{
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a0[e], f.b[e], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a1[e], f.b[e], c1, 0, 0, 0);
    }
}

// DBG (a diagnostic, IDELUCS_DEV=l1_debug, wrong results): 1 = no DMA inside the main loop (barriers stay); 2 = no barrier and no DMA inside
// the main loop; 3 = as 2 and no LDS reads either (the MFMA stream alone)
template <int DBG = 0>
__device__ __forceinline__ void l1_fwd_body(const L1Args &a, const int bid_in, unsigned char *smem)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (bid_in >= a.n_tiles) return;
    const int l = lane & 15, q = lane >> 4, wh = (wv >> 1) & 1, wr = wv & 1;
    // ---- tile of this workgroup.  Workgroups go to the 8 XCDs round-robin: the 32 of an XCD take 4 h-tiles x (n_rt / 4) row-tiles, so
    // an XCD's L2 streams 256 rows of W1 and m / 4 rows of x (8 MB at cfg2, all 32 tiles in K-lockstep) instead of all 24 MB
    const int n_rt = a.m / TR;
    int ht, rt;
    if ((n_rt & 3) == 0) {
        const int xcd = bid_in & 7, j = bid_in >> 3;
        ht = 4 * (xcd & 1) + (j & 3);
        rt = (n_rt >> 2) * (xcd >> 1) + (j >> 2);
    } else { ht = bid_in & 7; rt = bid_in >> 3; }
    const int h0 = ht * TH, r0 = rt * TR;
    const int K = a.K;
    const int nc = K / KC;                                   // >= STAGES (idl_l1_fwd_supported)
    if (wv >= 4) {
        // ================= a LOADER wave: every global access of the main loop.  An LDS-DMA costs the issuing wave tens of cycles
        // and the matrix pipe buffers ONE instruction: issued by the computing waves (this kernel's first form: 43.8 us) every DMA
        // left the pipe idle.  Loader p of NP owns DMA instructions p, p + NP, ... of a chunk's 24 (16 of A: rows 4 j .. 4 j + 3;
        // 8 of B); a lane fetches slot (lane & 15) ^ (row & 15) of its row: the bank swizzle, applied on the source side.
        const int p = wv - 4;
        constexpr int SHARE = 24 / NP;
        uint32_t off[SHARE], dst[SHARE];
#pragma unroll
        for (int i = 0; i < SHARE; ++i) {
            const int j = i * NP + p;                        // 0..15: A, 16..23: B
            const bool isA = i < 16 / NP;                    // (NP divides 16 and 8: a loader's first 16 / NP instructions are A's)
            const int row = 4 * (isA ? j : j - 16) + (lane >> 4), slot = (lane & 15) ^ (row & 15);
            off[i] = (uint32_t)((((isA ? h0 : r0) + row) * K + 4 * slot) * 4);
            dst[i] = (uint32_t)((isA ? 0 : A_BYTES) + (isA ? j : j - 16) * 1024);
        }
        const unsigned char *pa = (const unsigned char *)a.W1, *pb = (const unsigned char *)a.x;
        const uint32_t lds0 = (uint32_t)(uintptr_t)smem;     // LDS byte offset of the dynamic segment
        auto issue = [&](uint32_t stage_byte) {
#pragma unroll
            for (int i = 0; i < SHARE; ++i) dma16(off[i], i < 16 / NP ? pa : pb, lds0 + stage_byte + dst[i]);
            pa += KC * 4; pb += KC * 4;
        };
#pragma unroll
        for (int n = 0; n < STAGES; ++n) issue((uint32_t)(n * STAGE_BYTES));       // chunks 0 .. STAGES - 1 (nc >= STAGES)
        vm_wait<(STAGES - 1) * SHARE>();
        __builtin_amdgcn_s_barrier();                        // B_0: chunk 0 is in LDS
        uint32_t s_free = 0;                                 // stage of chunk c
        if (DBG >= 2) return;
#pragma unroll 1
        for (int c = 0; c < nc; ++c) {                       // B_{c + 1}: chunk c + 1 readable, chunk c's stage free (B_nc: nothing, see the compute loop)
            if (DBG == 1) { __builtin_amdgcn_s_barrier(); continue; }
            if (c + STAGES <= nc) vm_wait<(STAGES - 2) * SHARE>(); else vm_wait<0>();   // (chunks c + 2 .. c + STAGES - 1 may still fly)
            __builtin_amdgcn_s_barrier();
            if (c + STAGES < nc) issue(s_free);              // chunk c + STAGES
            s_free = s_free + STAGE_BYTES == (uint32_t)LDS_BYTES ? 0u : s_free + STAGE_BYTES;
        }
        return;                                              // (an ended wave no longer counts towards the barriers of the epilogue)
    }
    // ================= a COMPUTE wave: LDS reads and MFMAs only
    if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (a.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (a.prio >= 3) __builtin_amdgcn_s_setprio(3);
    // ---- fragment reads: row 32 wh + 16 hb + l of A, row 16 wr + l of B; slot (4 t + q) ^ l.  fp32 MFMA holds the SIMD's vector issue
    // for its whole 32 cycles, so every VALU instruction between two MFMAs is a bubble in the matrix pipe (this loop's second
    // form computed its LDS addresses per step: 5 v_add + 3 ds_read cost 72 cycles per 256-cycle step, 38.1 us against the MFMA
    // stream's 30.4).  All addresses are formed HERE, once: 16 registers (4 k-steps x {A, B} x {stages 0-1, stages 2-3}); inside
    // the loop a read is base register + immediate offset, and the loop is unrolled over the four stages.
    static_assert(STAGE_BYTES + 16 * 256 < 65536, "read_step's immediate offsets");
    constexpr int NB = (STAGES + 1) / 2;                     // one base register set per two stages
    uint32_t adA[NB][4], adB[NB][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint32_t sl = (uint32_t)(((4 * t + q) ^ l) * 16);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            adA[b][t] = (uint32_t)(uintptr_t)smem + (uint32_t)(2 * b * STAGE_BYTES + (32 * wh + l) * 256) + sl;
            adB[b][t] = (uint32_t)(uintptr_t)smem + (uint32_t)(2 * b * STAGE_BYTES + A_BYTES + (16 * wr + l) * 256) + sl;
        }
    }
    auto rd = [&](int st, int t) {                           // (st, t are constants after unrolling)
        Frag f;
        const uint32_t o = (uint32_t)((st & 1) * STAGE_BYTES);
        f.a0 = lds_read16(adA[st >> 1][t], o);
        f.a1 = lds_read16(adA[st >> 1][t], o + 16 * 256);
        f.b = lds_read16(adB[st >> 1][t], o);
        return f;
    };
    __builtin_amdgcn_s_barrier();                            // B_0
    f32x4_t c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
    Frag f0 = rd(0, 0), f1 = f0;
    int c = 0;
#pragma unroll 1
    while (c < nc) {                                         // unrolled over the stages; leaves in the middle when nc is no multiple
#pragma unroll
        for (int st = 0; st < STAGES; ++st) {
            // (sched_barrier: left alone, the compiler sinks every read to just in front of its first use -- one register set, the
            //  whole LDS latency exposed once per step; the reads of step t + 1 must issue BEFORE the MFMAs of step t)
            if (DBG < 3) f1 = rd(st, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma_step(f0, c0, c1);
            __builtin_amdgcn_sched_barrier(0);
            if (DBG < 3) f0 = rd(st, 2);
            __builtin_amdgcn_sched_barrier(0);
            mma_step(f1, c0, c1);
            __builtin_amdgcn_sched_barrier(0);
            if (DBG < 3) f1 = rd(st, 3);
            __builtin_amdgcn_sched_barrier(0);
            mma_step(f0, c0, c1);
            __builtin_amdgcn_sched_barrier(0);
            // the last step runs behind the barrier that publishes the next chunk and frees this chunk's stage (every read of
            // this chunk has been issued; the lgkmcnt wait is for the wave's OWN last reads: f1 must be in registers before the
            // stage is overwritten)
            // (unconditional, also behind the last chunk -- the loaders match it with one empty barrier, the fragments read there
            //  are never used: a branch here makes the compiler wait for ALL outstanding LDS reads at the join, every chunk)
            ++c;
            if (DBG < 2) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (DBG < 3) f0 = rd((st + 1) % STAGES, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma_step(f1, c0, c1);
            __builtin_amdgcn_sched_barrier(0);
            if (c == nc) break;
        }
    }
    // a1^T = W1 x^T, no bias (the mid-forward launch adds it)
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        const f32x4_t v = hb ? c1 : c0;
        const int hb0 = h0 + 32 * wh + 16 * hb + 4 * q, r = r0 + 16 * wr + l;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) a.r1[(int64_t)(hb0 + reg) * a.m + r] = v[reg];
    }
}

}  // namespace l1_dev
