// l1_planes_device.h -- the layer-1 forward product a1^T = W1 x^T on the fp16 MATRIX CORES from operands kept as two fp16 planes
// (planes.h): the device side, shared by planes.hip (idl_l1_planes: the step's first launch) and train_step.hip (the same tiles with the
// previous step's optimizer tail riding behind them, idl_l1_planes_rms: a measured variant, IDELUCS_PLANES_REDUCE=mid).
//
// Reference: Linear(F,512) of idelucs/PytorchUtils.py:38-45, called for both views of a batch at idelucs/models.py:124-125.  The fp32
// tiles of l1_device.h sit on the fp32 matrix pipe's floor (27.4 us at 512 x 1024 x 4096; 33.6 measured); the fp16 pipe is sixteen
// times faster, and with BOTH operands split in two planes the three products w0 x0 + w0 x1 + w1 x0 carry 22 significand bits per
// factor -- measured against a float64 product the sum is closer than the fp32 library GEMM (the probe: csrc/probe_split.hip,
// profiles/r05_probe_split_mfma.txt; tests/test_gpu_encoder.py).  What is left is data movement: 4 bytes an element, as fp32.
//
// 128 x 128 tiles of a1^T with an 8-way split of K, ONE K SLICE PER XCD (workgroups go to the XCDs round-robin: slice = blockIdx % 8, so an
// XCD's L2 streams one eighth of both operands; whole-K tiles of 64 x 32 would pull 600 MB a launch out of L2).  Four LOADER waves bring
// chunks of 64 k (four planes x 128 rows x 128 bytes: whole cache lines of the source) into LDS by LDS-DMA, two chunks resident, the
// 16-byte slots of a row swizzled on the SOURCE side (the DMA writes lane-linear); four COMPUTING waves (2 x 2, 64 x 64 each = 2 x 2 blocks
// of 32 x 32) read their fragments with ds_read_b128 one K-step ahead of the MFMAs (v_mfma_f32_32x32x16_f16); one barrier a chunk.  The
// partial sums go out as part[8][512][m] fp32 (scaled back by 2^-(W_EXP + X_EXP)); idl_reduce_parts_rms adds the eight in a fixed order.
// The operands' roles are symmetric: called with the batch's planes as "W1" and W1's as the "batch" the same tiles give part[8][m][512]
// (the step of n_clusters > 48, whose activations are not transposed).
// WHERE IT STANDS: 18.5-19.8 us in the step; what it moves out of L2 (134 MB at the ~10 TB/s this access pattern gets: 13.4 us) + 3.8 us of
// partial-sum stores + launch; computing alone 6.4 us, MFMA pipe busy 0.20.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "planes.h"

namespace l1p_dev {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 128, TN = 128, KC = 64, KSPLIT = 8, STAGES = 2;
constexpr int THREADS = 512;                                 // 4 computing + 4 loader waves
constexpr int ROWB = KC * 2;                                 // bytes of a row of a plane in a chunk: 128 = one cache line of the source
constexpr int PLANE = TM * ROWB;                             // bytes of one plane of a chunk (16 384)
constexpr int STAGE = 4 * PLANE;                             // w0 w1 x0 x1 (65 536)
// A chunk is 64 k: a row's 128 bytes are ONE cache line of the source (chunks of 32 k asked the L2 for every line twice, half a line
// each time: 22.5 -> 20.6 us in tools/bench_planes.py's harness), two chunks resident (more did not help at 32 k: three, four, five stages 22.0 / 22.4 / 22.5 us)
constexpr int LDS_BYTES = STAGES * STAGE;                    // 131 072
constexpr int PER = 16;                                      // DMA instructions (1 KiB = 8 rows each) a loader issues per chunk: 4 planes x 16 blocks / 4 loaders

struct L1pArgs {
    const uint16_t *wh, *wl;       // W1's planes [n_out][K]
    const uint16_t *xh, *xl;       // the batch's planes [m][K]
    float *part;                   // [KSPLIT][n_out][m]
    int m, n_out, K, n_tiles;      // n_tiles = (n_out / 128) (m / 128) KSPLIT
    int ldw, ldx;                  // elements between two rows of W1's / the batch's planes (>= K, multiples of 8)
    int dbg;                       // diagnostics (IDELUCS_L1P_DBG; wrong results): 1 no DMA, 4 no stores of the partial sums
};

__host__ __device__ inline bool supported(int m, int n_out, int K)
{
    return m >= TN && (m % TN) == 0 && n_out >= TM && (n_out % TM) == 0 && (K % (KSPLIT * KC)) == 0 && K / (KSPLIT * KC) >= STAGES &&
           (int64_t)m * (K + 1024) < (1ll << 30) && (int64_t)n_out * (K + 1024) < (1ll << 30);
}

__device__ __forceinline__ void dma16(uint32_t voff, const void *sbase, uint32_t lds_byte)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
}

__device__ __forceinline__ void l1p_body(const L1pArgs &g, const int bid, unsigned char *smem)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = g.m / TN;
    const int s = bid % KSPLIT, t = bid / KSPLIT;
    const int m0 = (t / tiles_n) * TM, n0 = (t % tiles_n) * TN;      // hidden units, batch rows
    const int kr = g.K / KSPLIT, kb = s * kr, nc = kr / KC;           // nc >= 2 (supported())
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    if (wv >= 4) {                                           // ---- a loader
        const int lw = wv - 4;
        auto issue = [&](int k0, uint32_t lds_stage) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int j = lw * PER + i;                  // plane j / 16 (w0 w1 x0 x1), rows 8 (j % 16) .. + 7
                const int pq = j >> 4, blk = j & 15;
                const int row = blk * 8 + (lane >> 3), slot = lane & 7, src = slot ^ (row & 7);
                const uint16_t *base = pq == 0 ? g.wh : (pq == 1 ? g.wl : (pq == 2 ? g.xh : g.xl));
                const int64_t r = (pq < 2 ? m0 : n0) + row;
                const uint32_t voff = (uint32_t)((r * (pq < 2 ? g.ldw : g.ldx) + k0 + src * 8) * 2);
                dma16(voff, base, lds_stage + (uint32_t)(pq * PLANE + blk * 1024));
            }
        };
        const bool dma = !(g.dbg & 1);
        if (dma) { issue(kb, lds0); issue(kb + KC, lds0 + STAGE); }          // nc >= 2 (supported())
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PER) : "memory");
        __builtin_amdgcn_s_barrier();                        // B_0: chunk 0 is in LDS
        for (int c = 0; c < nc; ++c) {                       // B_{c + 1}: chunk c + 1 readable, chunk c's stage free
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (chunk c + 1 is the only one in flight)
            __builtin_amdgcn_s_barrier();
            if (c + 2 < nc && dma) issue(kb + (c + 2) * KC, lds0 + (uint32_t)((c & 1) * STAGE));
        }
        return;
    }
    const int wm = (wv >> 1) * 64, wn = (wv & 1) * 64;
    f32x16 hi[2][2], lo[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { hi[i][j][e] = 0.f; lo[i][j][e] = 0.f; }
    __builtin_amdgcn_s_barrier();                            // B_0
    const int r = lane & 31, kg = lane >> 5;
    // The fragments of K-step t + 1 are read into a second register set while the twelve MFMAs of step t run (in program order -- eight
    // reads, a wait, twelve MFMAs per step -- the matrix pipe idled through every read: 12.5 us of computing for 5.1 us of MFMAs,
    // IDELUCS_L1P_DBG); a chunk is four steps, and the barrier that publishes chunk c + 1 (and frees chunk c's stage) sits in front of its last:
    //     read F1 = (c, 1) | MFMA F0;  read F0 = (c, 2) | MFMA F1;  read F1 = (c, 3) | MFMA F0;  wait F1, barrier, read F0 = (c + 1, 0) | MFMA F1
    // The reads are inline asm with the waits placed by hand and the set's registers passed THROUGH the wait: the compiler cannot count
    // LDS reads in flight across the loop's back edge and put `s_waitcnt lgkmcnt(0)` in front of every step's first MFMA, i.e. it waited
    // for the set it had just requested.
    static_assert(KC == 64 && PLANE == 16384 && STAGES == 2, "the immediates and the schedule below");
    struct Frags { u32x4 a[2][2], b[2][2]; };            // [plane][32-row block]
    // the lane's byte offset inside a stage for step ks: row (wm | wn) + r, 16-byte slot (2 ks + kg) ^ (row & 7); plane and 32-row block are immediates
    uint32_t oa[4], ob[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        oa[ks] = (uint32_t)((wm + r) * ROWB + (((2 * ks + kg) ^ (r & 7)) << 4));
        ob[ks] = (uint32_t)((wn + r) * ROWB + (((2 * ks + kg) ^ (r & 7)) << 4));
    }
#define L1P_READ(DST, ADDR, IMM) asm volatile("ds_read_b128 %0, %1 offset:" #IMM : "=v"(DST) : "v"(ADDR))
#define L1P_READ_SET(F, OA, OB)                                                                                                   \
    do {                                                                                                                          \
        L1P_READ(F.a[0][0], OA, 0); L1P_READ(F.a[0][1], OA, 4096); L1P_READ(F.a[1][0], OA, 16384); L1P_READ(F.a[1][1], OA, 20480); \
        L1P_READ(F.b[0][0], OB, 32768); L1P_READ(F.b[0][1], OB, 36864); L1P_READ(F.b[1][0], OB, 49152); L1P_READ(F.b[1][1], OB, 53248); \
    } while (0)
#define L1P_WAIT(N, F)                                                                                                            \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(F.a[0][0]), "+v"(F.a[0][1]), "+v"(F.a[1][0]), "+v"(F.a[1][1]), "+v"(F.b[0][0]), \
                 "+v"(F.b[0][1]), "+v"(F.b[1][0]), "+v"(F.b[1][1]))
    auto mma = [&](const Frags &f) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f16x8 a0 = __builtin_bit_cast(f16x8, f.a[0][i]), a1 = __builtin_bit_cast(f16x8, f.a[1][i]);
                const f16x8 b0 = __builtin_bit_cast(f16x8, f.b[0][j]), b1 = __builtin_bit_cast(f16x8, f.b[1][j]);
                hi[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, hi[i][j], 0, 0, 0);
                lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, lo[i][j], 0, 0, 0);
                lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, lo[i][j], 0, 0, 0);
            }
    };
    Frags f0, f1;
    {
        const uint32_t a = lds0 + oa[0], b = lds0 + ob[0];
        L1P_READ_SET(f0, a, b);
    }
    for (int c = 0; c < nc; ++c) {
        const uint32_t st = lds0 + (uint32_t)((c & 1) * STAGE);
        { const uint32_t a = st + oa[1], b = st + ob[1]; L1P_READ_SET(f1, a, b); }
        L1P_WAIT(8, f0);                                     // F0 is in; the eight reads of F1 may be in flight
        __builtin_amdgcn_sched_barrier(0);
        mma(f0);
        __builtin_amdgcn_sched_barrier(0);
        { const uint32_t a = st + oa[2], b = st + ob[2]; L1P_READ_SET(f0, a, b); }
        L1P_WAIT(8, f1);
        __builtin_amdgcn_sched_barrier(0);
        mma(f1);
        __builtin_amdgcn_sched_barrier(0);
        { const uint32_t a = st + oa[3], b = st + ob[3]; L1P_READ_SET(f1, a, b); }
        L1P_WAIT(8, f0);
        __builtin_amdgcn_sched_barrier(0);
        mma(f0);
        __builtin_amdgcn_sched_barrier(0);
        L1P_WAIT(0, f1);                                     // (the wave's reads of chunk c are in registers before its stage is refilled)
        __builtin_amdgcn_s_barrier();                        // B_{c + 1}
        if (c + 1 < nc) {
            const uint32_t sn = lds0 + (uint32_t)(((c + 1) & 1) * STAGE);
            const uint32_t a = sn + oa[0], b = sn + ob[0];
            L1P_READ_SET(f0, a, b);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma(f1);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef L1P_READ
#undef L1P_READ_SET
#undef L1P_WAIT
    // C/D layout of 32x32: lane l, register e -> row (e / 4) * 8 + (l / 32) * 4 + e % 4, column l % 32
    if (g.dbg & 4) return;
    const float inv = __builtin_ldexpf(1.f, -(idl_planes::W_EXP + idl_planes::X_EXP));
    float *out = g.part + (int64_t)s * g.n_out * g.m;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm + 32 * i + (e >> 2) * 8 + (lane >> 5) * 4 + (e & 3), col = n0 + wn + 32 * j + (lane & 31);
                out[(int64_t)row * g.m + col] = (hi[i][j][e] + lo[i][j][e]) * inv;
            }
}

}  // namespace l1p_dev
