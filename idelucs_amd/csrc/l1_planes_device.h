// l1_planes_device.h -- the layer-1 forward product a1^T = W1 x^T on the fp16 MATRIX CORES from operands kept as two fp16 planes
// (planes.h): the device side, shared by planes.hip (idl_l1_planes: the step's first launch) and train_step.hip (l1_planes_batched_kernel: a rank's
// voters in lockstep).
//
// Reference: Linear(F,512) of idelucs/PytorchUtils.py:38-45, called for both views of a batch at idelucs/models.py:124-125.  The fp32
// tiles of l1_device.h sit on the fp32 matrix pipe's floor (27.4 us at 512 x 1024 x 4096; 33.6 measured); the fp16 pipe is sixteen
// times faster, and with BOTH operands split in two planes the three products w0 x0 + w0 x1 + w1 x0 carry 22 significand bits per
// factor -- measured against a float64 product the sum is closer than the fp32 library GEMM on the step's dense sums (the probe: tools/probe_split.hip,
// profiles/r05_probe_split_mfma.txt; tests/test_gpu_planes.py, profiles/r06_planes_adversarial.txt).  What is left is data movement: 4 bytes an element, as fp32.
//
// 128 x 128 tiles of a1^T with an 8-way split of K, ONE K SLICE PER XCD (workgroups go to the XCDs round-robin: slice = blockIdx % 8, so an
// XCD's L2 streams one eighth of both operands; whole-K tiles of 64 x 32 would pull 600 MB a launch out of L2).  Four LOADER waves bring
// chunks of 64 k (four planes x 128 rows x 128 bytes: whole cache lines of the source) into LDS by LDS-DMA, two chunks resident, the
// 16-byte slots of a row swizzled on the SOURCE side (the DMA writes lane-linear); four COMPUTING waves (2 x 2, 64 x 64 each = 2 x 2 blocks
// of 32 x 32) read their fragments with ds_read_b128 one K-step ahead of the MFMAs (v_mfma_f32_32x32x16_f16); one barrier a chunk.  The
// partial sums go out as part[8][512][m] fp32 (scaled back by 2^-(W_EXP + X_EXP)); idl_reduce_parts_rms adds the eight in a fixed order.
// The operands' roles are symmetric: called with the batch's planes as "W1" and W1's as the "batch" the same tiles give part[8][m][512]
// (the step of n_clusters > 48, whose activations are not transposed).
// WHERE IT STANDS (round 6): 15.3 us in the step (round 5: 18.5-19.8).  The loaders alone, free-running, stream the launch's 134 MB out of L2 in 4 us (33 TB/s; 13 us
// without the K-slice-per-XCD mapping); computing alone 6.4 us; together 11 + 3.8 us of partial-sum stores + launch: per K-step the LDS serves 32 KB of fragment reads and
// 16 KB of DMA writes (384 clocks at 128 B a clock) beside 384 clocks of MFMAs per SIMD -- two pipes at equal load that a one-step look-ahead keeps ~half busy (MFMA 0.23).
// Four chunks of 32 k resident instead of two of 64 k: 17.9 us (not a latency pipeline); DESIGN 4.4 and History.
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
    int dbg;                       // diagnostics (IDELUCS_DEV=l1p_dbg; wrong results): 1 no DMA, 4 no stores of the partial sums
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
    if (g.dbg & 128) return;                                 // EXPERIMENT: the launch alone
    int s = bid % KSPLIT, t = bid / KSPLIT;
    if (g.dbg & 64) { s = 0; t = 0; }                        // EXPERIMENT: every workgroup reads the same operands
    if (g.dbg & 256) { s = bid / 32; t = bid % 32; }         // EXPERIMENT: an XCD sees every K slice
    const int m0 = (t / tiles_n) * TM, n0 = (t % tiles_n) * TN;      // hidden units, batch rows
    const int kr = g.K / KSPLIT, kb = s * kr, nc = kr / KC;           // nc >= 2 (supported())
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    if (wv >= 4) {                                           // ---- a loader
        const int lw = wv - 4;
        // loader lw brings plane lw (w0 w1 x0 x1) of a chunk: 16 instructions of 8 rows each; everything but k0 and the stage is formed once
        static_assert(PER == 16, "a loader wave owns one plane of a chunk");
        const uint16_t *const base = lw == 0 ? g.wh : (lw == 1 ? g.wl : (lw == 2 ? g.xh : g.xl));
        const int ld = lw < 2 ? g.ldw : g.ldx, rows0 = lw < 2 ? m0 : n0;
        const int rr = lane >> 3;
        uint32_t vo[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {                      // LDS slot lane & 7 of row i * 8 + rr holds source slot (lane & 7) ^ ((row >> 1) & 7)
            const int src = (lane & 7) ^ ((4 * (i & 1) + (rr >> 1)) & 7);
            vo[i] = (uint32_t)(((rows0 + i * 8 + rr) * ld + src * 8) * 2);
        }
        const uint32_t lds_plane = (uint32_t)(lw * PLANE);
        auto issue = [&](int k0, uint32_t lds_stage) {
            const char *b = (const char *)base + (int64_t)k0 * 2;
#pragma unroll
            for (int i = 0; i < PER; ++i) dma16(vo[i], b, lds_stage + lds_plane + (uint32_t)(i * 1024));
        };
        const bool dma = !(g.dbg & 1);
        if (g.dbg & 48) {                                    // EXPERIMENT: the loaders alone, free-running (16: two chunks in flight, 32: one)
            for (int c = 0; c < nc; ++c) {
                issue(kb + c * KC, lds0 + (uint32_t)((c & 1) * STAGE));
                if (g.dbg & 32) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PER) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
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
    if (g.dbg & 48) return;
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
    // A K-step is ONE asm statement: the twelve MFMAs of step t with the eight ds_read_b128 of step t + 1 BETWEEN them (a read behind each of
    // the first eight), and the wait for those reads at its END.  (Round 5 issued the eight reads in front of the twelve MFMAs: a wave cannot
    // issue its first MFMA before the LDS has taken all eight, and with four waves at the same point of the chunk the last one waited ~200
    // cycles a step with the matrix pipe idle -- 9.2 us of computing for 5.1 us of MFMAs.)  The registers a read targets are early-clobber
    // outputs of the statement that also waits for them: they are complete when the compiler first sees them, so no copy of a register in
    // flight can exist (the hazard class of DESIGN's "the ring that was copied before its wait").
    // A chunk is four steps; the barrier that publishes chunk c + 1 (and frees chunk c's stage) sits in front of its last:
    //     MFMA (c,0) | read (c,1);  MFMA (c,1) | read (c,2);  MFMA (c,2) | read (c,3);  barrier;  MFMA (c,3) | read (c + 1,0)
    // LDS image: a row of a plane is 128 bytes (64 k); its 16-byte slots are XORed with (row >> 1) & 7 -- the 16 lanes ds_read_b128 serves in
    // one cycle ({0-3, 12-15, 20-27}, ...) then cover all 64 banks (row & 7, round 5's key, put two of them on every bank).
    static_assert(KC == 64 && PLANE == 16384 && STAGES == 2, "the immediates and the schedule below");
    struct Frags { u32x4 a00, a01, a10, a11, b00, b01, b10, b11; };      // a<plane><32-row block>
    uint32_t oa[4], ob[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        oa[ks] = (uint32_t)((wm + r) * ROWB + (((2 * ks + kg) ^ ((r >> 1) & 7)) << 4));
        ob[ks] = (uint32_t)((wn + r) * ROWB + (((2 * ks + kg) ^ ((r >> 1) & 7)) << 4));
    }
#define L1P_STEP(C, N, PA, PB)                                                                                                       \
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %16, %20, %0\n\tds_read_b128 %8, %24 offset:0\n\t"                                      \
                 "v_mfma_f32_32x32x16_f16 %4, %16, %22, %4\n\tds_read_b128 %9, %24 offset:4096\n\t"                                   \
                 "v_mfma_f32_32x32x16_f16 %1, %16, %21, %1\n\tds_read_b128 %10, %24 offset:16384\n\t"                                 \
                 "v_mfma_f32_32x32x16_f16 %5, %16, %23, %5\n\tds_read_b128 %11, %24 offset:20480\n\t"                                 \
                 "v_mfma_f32_32x32x16_f16 %2, %17, %20, %2\n\tds_read_b128 %12, %25 offset:32768\n\t"                                 \
                 "v_mfma_f32_32x32x16_f16 %6, %17, %22, %6\n\tds_read_b128 %13, %25 offset:36864\n\t"                                 \
                 "v_mfma_f32_32x32x16_f16 %3, %17, %21, %3\n\tds_read_b128 %14, %25 offset:49152\n\t"                                 \
                 "v_mfma_f32_32x32x16_f16 %7, %17, %23, %7\n\tds_read_b128 %15, %25 offset:53248\n\t"                                 \
                 "v_mfma_f32_32x32x16_f16 %4, %18, %20, %4\n\t"                                                                       \
                 "v_mfma_f32_32x32x16_f16 %5, %18, %21, %5\n\t"                                                                       \
                 "v_mfma_f32_32x32x16_f16 %6, %19, %20, %6\n\t"                                                                       \
                 "v_mfma_f32_32x32x16_f16 %7, %19, %21, %7\n\t"                                                                       \
                 "s_waitcnt lgkmcnt(0)"                                                                                               \
                 : "+v"(hi[0][0]), "+v"(hi[0][1]), "+v"(hi[1][0]), "+v"(hi[1][1]), "+v"(lo[0][0]), "+v"(lo[0][1]), "+v"(lo[1][0]),    \
                   "+v"(lo[1][1]), "=&v"(N.a00), "=&v"(N.a01), "=&v"(N.a10), "=&v"(N.a11), "=&v"(N.b00), "=&v"(N.b01), "=&v"(N.b10), \
                   "=&v"(N.b11)                                                                                                       \
                 : "v"(C.a00), "v"(C.a01), "v"(C.a10), "v"(C.a11), "v"(C.b00), "v"(C.b01), "v"(C.b10), "v"(C.b11), "v"(PA), "v"(PB)  \
                 : "memory")
    // (operands: %0-%3 hi[i][j], %4-%7 lo[i][j]; %16 %17 W's high plane rows 0-31 / 32-63, %18 %19 its low plane; %20 %21 the batch's high plane,
    //  %22 %23 its low plane.  Every accumulator sees its products in round 5's order -- hi: w0 x0; lo: w0 x1, then w1 x0 -- so the sums are bit for
    //  bit what the builtin form gave.)
    Frags f0, f1;
    {
        const uint32_t pa = lds0 + oa[0], pb = lds0 + ob[0];
        asm volatile("ds_read_b128 %0, %8 offset:0\n\tds_read_b128 %1, %8 offset:4096\n\tds_read_b128 %2, %8 offset:16384\n\tds_read_b128 %3, %8 offset:20480\n\t"
                     "ds_read_b128 %4, %9 offset:32768\n\tds_read_b128 %5, %9 offset:36864\n\tds_read_b128 %6, %9 offset:49152\n\tds_read_b128 %7, %9 offset:53248\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(f0.a00), "=&v"(f0.a01), "=&v"(f0.a10), "=&v"(f0.a11), "=&v"(f0.b00), "=&v"(f0.b01), "=&v"(f0.b10), "=&v"(f0.b11)
                     : "v"(pa), "v"(pb) : "memory");
    }
    for (int c = 0; c < nc; ++c) {
        const uint32_t st = lds0 + (uint32_t)((c & 1) * STAGE);
        { const uint32_t pa = st + oa[1], pb = st + ob[1]; L1P_STEP(f0, f1, pa, pb); }
        { const uint32_t pa = st + oa[2], pb = st + ob[2]; L1P_STEP(f1, f0, pa, pb); }
        { const uint32_t pa = st + oa[3], pb = st + ob[3]; L1P_STEP(f0, f1, pa, pb); }
        __builtin_amdgcn_s_barrier();                        // B_{c + 1}: this wave's reads of chunk c are in registers; chunk c + 1 is in LDS
        {                                                    // (behind the last chunk: a read of stale bytes nobody uses)
            const uint32_t sn = lds0 + (uint32_t)(((c + 1 < nc ? c + 1 : c) & 1) * STAGE);
            const uint32_t pa = sn + oa[0], pb = sn + ob[0];
            L1P_STEP(f1, f0, pa, pb);
        }
    }
#undef L1P_STEP
    // (the compiler does not know that the statements above hold MFMAs: the wait states between an MFMA's write and a VALU read of its
    //  accumulator -- at most 18 for this shape -- are paid here, once, by hand)
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(hi[0][0]), "+v"(hi[0][1]), "+v"(hi[1][0]), "+v"(hi[1][1]), "+v"(lo[0][0]), "+v"(lo[0][1]), "+v"(lo[1][0]), "+v"(lo[1][1]));
    if (g.dbg & 4) return;
    // ---- epilogue: the wave turns its 64 x 64 block around through LDS (every wave is past the last barrier: nobody reads a stage any more; the
    // wave's image is its own) and stores 16 bytes a lane -- 16 store instructions a lane where the accumulators' own layout needs 64.
    // C/D layout of 32x32: lane l, register e -> row (e / 4) * 8 + (l / 32) * 4 + e % 4, column l % 32
    const float inv = __builtin_ldexpf(1.f, -(idl_planes::W_EXP + idl_planes::X_EXP));
    if (g.dbg & 2) {                                         // EXPERIMENT: round 5's epilogue (a dword a lane and store)
        float *o2 = g.part + (int64_t)s * g.n_out * g.m;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = m0 + wm + 32 * i + (e >> 2) * 8 + (lane >> 5) * 4 + (e & 3), col = n0 + wn + 32 * j + (lane & 31);
                    o2[(int64_t)row * g.m + col] = (hi[i][j][e] + lo[i][j][e]) * inv;
                }
        return;
    }
    constexpr int EP = 68;
    static_assert(4 * 64 * EP * 4 <= LDS_BYTES, "the four waves' epilogue images");
    float *img = (float *)smem + wv * (64 * EP);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rl = 32 * i + (e >> 2) * 8 + (lane >> 5) * 4 + (e & 3), cl = 32 * j + (lane & 31);
                img[rl * EP + cl] = (hi[i][j][e] + lo[i][j][e]) * inv;
            }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    float *out = g.part + (int64_t)s * g.n_out * g.m + (int64_t)(m0 + wm) * g.m + n0 + wn;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int idx = lane + 64 * u, rl = idx >> 4, c4 = idx & 15;
        const float4 v = *(const float4 *)(img + rl * EP + 4 * c4);
        *(float4 *)(out + (int64_t)rl * g.m + 4 * c4) = v;
    }
}

}  // namespace l1p_dev
