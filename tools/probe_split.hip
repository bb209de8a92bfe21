// probe_split.hip -- a PROBE, not part of the product path (tools/probe_split_mfma.py; DESIGN.md History, "the two products on the
// bf16 matrix cores").  Question: what would the step's two big fp32 products cost on the bf16 MFMA pipe (16 x the fp32 MFMA rate)
// with every operand split three ways, x = x0 + x1 + x2 (bf16 each, 24 significand bits in all), and the six products
// x0 y0, x0 y1, x1 y0, x1 y1, x0 y2, x2 y0 accumulated in fp32 (what is dropped is of the order of fp32's own rounding)?
//
// The kernel: C[M][N] = A[M][K] B[N][K]^T (both operands K-contiguous: the layer-1 forward's shape, 1024 x 512 x 4096) from
// PRE-SPLIT operands (three bf16 planes each -- in the step the producers of x, W1 and dr1 would write them), 128 x 128 tiles
// with an 8-way split of K (256 workgroups; per-CU L2 traffic of 64 x 32 whole-K tiles would be 600 MB a launch: L2-bound), a K slice per XCD,
// partial sums written as fp32 [S][M][N].  Four waves, each 64 x 64 = 2 x 2 blocks of v_mfma_f32_32x32x16_bf16; operands staged
// through LDS by LDS-DMA (global_load_lds_dwordx4) two chunks of 32 k deep, the 16-byte slots of a 64-byte row swizzled on the
// source side so that the fragment reads (ds_read_b128) are conflict-free.
#include "../idelucs_amd/csrc/common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 128, TN = 128, KC = 32;                   // tile; k per chunk (64 bytes of bf16 a row)
constexpr int PLANE_A = TM * KC * 2, PLANE_B = TN * KC * 2;  // bytes of one plane of a chunk
constexpr int STAGE = 3 * PLANE_A + 3 * PLANE_B;             // 49 152
constexpr int STAGES = 2;
constexpr int NT = 256;

__device__ __forceinline__ void dma16(uint32_t voff, const void *sbase, uint32_t lds_byte)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte) : "memory");
}

struct SplitArgs {
    const uint16_t *a[3], *b[3];       // planes [M][K], [N][K]
    float *cpart;                      // [S][M][N]
    int M, N, K, S, products;          // products: 6 (the full set), 3 (x0y0, x0y1, x1y0), 1 (x0y0: plain bf16)
};

// chunk c of the workgroup's K range into stage st: 48 DMA instructions of 1 KiB (16 rows x 64 bytes), 12 per wave
__device__ __forceinline__ void issue_chunk(const SplitArgs &g, int m0, int n0, int k0, uint32_t lds_stage, int wv, int lane)
{
    // instruction j (0..47): plane j / 8 of A (0..2) or B (3..5), rows 16 (j % 8) .. + 15; lane -> row lane / 4, slot lane % 4 of the
    // LDS row, which holds source chunk slot ^ ((row >> 2) & 3)
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const int j = wv * 12 + i;
        const int pl = j >> 3, blk = j & 7;
        const bool isA = pl < 3;
        const int row = blk * 16 + (lane >> 2), slot = lane & 3, src = slot ^ ((row >> 2) & 3);
        const uint16_t *base = isA ? g.a[pl] : g.b[pl - 3];
        const int64_t r = (isA ? m0 : n0) + row;
        const uint32_t voff = (uint32_t)((r * g.K + k0 + src * 8) * 2);
        dma16(voff, base, lds_stage + (uint32_t)(pl * PLANE_A + blk * 1024));      // (PLANE_A == PLANE_B)
    }
}

__device__ __forceinline__ bf16x8 frag(const unsigned char *smem, int plane_off, int row, int chunk)
{
    const int slot = chunk ^ ((row >> 2) & 3);
    const u32x4 v = *(const __attribute__((address_space(3))) u32x4 *)(uintptr_t)((uint32_t)(uintptr_t)smem + plane_off + row * 64 + slot * 16);
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(NT, 1) void split_gemm_kernel(SplitArgs g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = g.N / TN;
    // (workgroups go to the 8 XCDs round-robin: with the K slice = blockIdx % S an XCD's L2 streams ONE slice of both operands -- 4.7 MB at
    //  S = 8 -- instead of all 38 MB: with slice = blockIdx / tiles every XCD read everything, 300 MB a launch, 26 us whatever the products)
    const int s = blockIdx.x % g.S, t = blockIdx.x / g.S;
    const int m0 = (t / tiles_n) * TM, n0 = (t % tiles_n) * TN;
    const int kr = g.K / g.S, kb = s * kr, nc = kr / KC;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const int wm = (wv >> 1) * 64, wn = (wv & 1) * 64;       // the wave's 64 x 64 corner in the tile
    f32x16 hi[2][2], lo[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { hi[i][j][e] = 0.f; lo[i][j][e] = 0.f; }
    issue_chunk(g, m0, n0, kb, lds0, wv, lane);
    for (int c = 0; c < nc; ++c) {
        if (c + 1 < nc) issue_chunk(g, m0, n0, kb + (c + 1) * KC, lds0 + (uint32_t)(((c + 1) & 1) * STAGE), wv, lane);
        if (c + 1 < nc) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // chunk c is in LDS (every wave's share)
        const unsigned char *st = smem + (c & 1) * STAGE;
        const int r = lane & 31, kg = lane >> 5;             // fragment row / its 8-deep k group
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            const int chunk = ks * 2 + kg;
            bf16x8 fa[3][2], fb[3][2];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[pl][i] = frag(st, pl * PLANE_A, wm + 32 * i + r, chunk);
                    fb[pl][i] = frag(st, 3 * PLANE_A + pl * PLANE_B, wn + 32 * i + r, chunk);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    hi[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][j], hi[i][j], 0, 0, 0);
                    if (g.products >= 3) {
                        lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[1][j], lo[i][j], 0, 0, 0);
                        lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[0][j], lo[i][j], 0, 0, 0);
                    }
                    if (g.products >= 6) {
                        lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[1][j], lo[i][j], 0, 0, 0);
                        lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[2][j], lo[i][j], 0, 0, 0);
                        lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2][i], fb[0][j], lo[i][j], 0, 0, 0);
                    }
                }
        }
        __builtin_amdgcn_s_barrier();                        // every wave has read chunk c: its stage may be refilled (by the issue of c + 2)
    }
    // C/D layout of 32x32: lane l, register e -> row (e / 4) * 8 + (l / 32) * 4 + e % 4, column l % 32
    float *out = g.cpart + (int64_t)s * g.M * g.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm + 32 * i + (e >> 2) * 8 + (lane >> 5) * 4 + (e & 3), col = n0 + wn + 32 * j + (lane & 31);
                out[(int64_t)row * g.N + col] = hi[i][j][e] + lo[i][j][e];
            }
}

// The same product with the staging of csrc/l1_device.h: four LOADER waves issue every LDS-DMA (a compute wave that issues one leaves
// the matrix pipe idle meanwhile), three chunks resident, one barrier per chunk.
constexpr int STAGES2 = 3;
// PLANES = 3, bf16: the six (three, one) products above.  PLANES = 2, F16: x = x0 + x1 in fp16 (22 significand bits; the caller scales
// a tensor into fp16's range), the three products x0 y0, x0 y1, x1 y0 -- 4 bytes an element, as fp32
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <int PLANES, bool F16>
__global__ __launch_bounds__(2 * NT, 1) void split_gemm_kernel2(SplitArgs g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = g.N / TN;
    // (workgroups go to the 8 XCDs round-robin: with the K slice = blockIdx % S an XCD's L2 streams ONE slice of both operands -- 4.7 MB at
    //  S = 8 -- instead of all 38 MB: with slice = blockIdx / tiles every XCD read everything, 300 MB a launch, 26 us whatever the products)
    const int s = blockIdx.x % g.S, t = blockIdx.x / g.S;
    const int m0 = (t / tiles_n) * TM, n0 = (t % tiles_n) * TN;
    const int kr = g.K / g.S, kb = s * kr, nc = kr / KC;       // nc >= STAGES2 (checked by the launcher)
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    constexpr int PER = PLANES * 4;                          // DMA instructions a loader issues per chunk (2 x PLANES planes x 8 blocks / 4 loaders)
    if (wv >= 4) {                                           // ---- a loader
        const int lw = wv - 4;
        auto issue = [&](int k0, uint32_t lds_stage) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int j = lw * PER + i;                  // 0 .. 8 * 2 * PLANES - 1: plane-of-A / plane-of-B major, 16-row block minor
                const int pq = j >> 3, blk = j & 7;
                const bool isA = pq < PLANES;
                const int pl = isA ? pq : pq - PLANES;
                const int row = blk * 16 + (lane >> 2), slot = lane & 3, src = slot ^ ((row >> 2) & 3);
                const uint16_t *base = isA ? g.a[pl] : g.b[pl];
                const int64_t r = (isA ? m0 : n0) + row;
                const uint32_t voff = (uint32_t)((r * g.K + k0 + src * 8) * 2);
                dma16(voff, base, lds_stage + (uint32_t)((isA ? pl : 3 + pl) * PLANE_A + blk * 1024));
            }
        };
#pragma unroll
        for (int c = 0; c < STAGES2; ++c) issue(kb + c * KC, lds0 + (uint32_t)(c * STAGE));
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * PER) : "memory");
        __builtin_amdgcn_s_barrier();                        // B_0: chunk 0 is in LDS
        for (int c = 0; c < nc; ++c) {                       // B_{c + 1}: chunk c + 1 readable, chunk c's stage free
            if (c + STAGES2 <= nc) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PER) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (c + STAGES2 < nc) issue(kb + (c + STAGES2) * KC, lds0 + (uint32_t)((c % STAGES2) * STAGE));
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
    for (int c = 0; c < nc; ++c) {
        const unsigned char *st = smem + (c % STAGES2) * STAGE;
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            const int chunk = ks * 2 + kg;
            bf16x8 fa[PLANES][2], fb[PLANES][2];
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[pl][i] = frag(st, pl * PLANE_A, wm + 32 * i + r, chunk);
                    fb[pl][i] = frag(st, 3 * PLANE_A + pl * PLANE_B, wn + 32 * i + r, chunk);
                }
            auto mma = [&](const bf16x8 &x, const bf16x8 &y, f32x16 acc) {
                if (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), acc, 0, 0, 0);
                return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0);
            };
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    hi[i][j] = mma(fa[0][i], fb[0][j], hi[i][j]);
                    if (g.products >= 3) {
                        lo[i][j] = mma(fa[0][i], fb[1][j], lo[i][j]);
                        lo[i][j] = mma(fa[1][i], fb[0][j], lo[i][j]);
                    }
                    if (PLANES == 3 && g.products >= 6) {
                        lo[i][j] = mma(fa[1][i], fb[1][j], lo[i][j]);
                        lo[i][j] = mma(fa[0][i], fb[PLANES - 1][j], lo[i][j]);
                        lo[i][j] = mma(fa[PLANES - 1][i], fb[0][j], lo[i][j]);
                    }
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (the wave's own reads of chunk c are in registers before its stage is refilled)
        __builtin_amdgcn_s_barrier();                        // B_{c + 1}
    }
    float *out = g.cpart + (int64_t)s * g.M * g.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm + 32 * i + (e >> 2) * 8 + (lane >> 5) * 4 + (e & 3), col = n0 + wn + 32 * j + (lane & 31);
                out[(int64_t)row * g.N + col] = hi[i][j][e] + lo[i][j][e];
            }
}

// The weight gradient's form: C[M][N] = sum_k At[k][M] Bt[k][N] with the operands as they LIE in memory for dW1 = dr1^T x (dr1 [rows][512],
// x [rows][4096]: the contraction index is the slow one) -- two fp16 planes each, three products.  A chunk is 32 k-rows of 128 columns
// (256 bytes a row) per plane, brought in by LDS-DMA as it lies; the MFMA operand of a lane -- 8 consecutive k of ONE column -- comes out
// of two ds_read_b64_tr_b16 (the hardware transpose: a block of 4 k-rows x 16 columns per 16 lanes), the 16-byte slots of a row
// swizzled by cdna_hip_programming.md's T10 rule (b) so that rows 256 bytes apart do not meet in a bank.
typedef short s16x4v __attribute__((__vector_size__(4 * sizeof(short))));
constexpr int T_ROWB = TM * 2;                               // bytes of a k-row of a plane in LDS (128 columns)
constexpr int T_PLANE = KC * T_ROWB;                         // 8 192
constexpr int T_STAGE = 4 * T_PLANE;                         // A0 A1 B0 B1
__device__ __forceinline__ int t_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__global__ __launch_bounds__(2 * NT, 1) void split_gemm_kernel_t(SplitArgs g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = g.N / TN;
    const int s = blockIdx.x % g.S, t = blockIdx.x / g.S;
    const int m0 = (t / tiles_n) * TM, n0 = (t % tiles_n) * TN;
    const int kr = g.K / g.S, kb = s * kr, nc = kr / KC;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    constexpr int PER = 8;                                   // 32 DMA instructions a chunk (4 planes x 8 blocks of 4 rows), 8 per loader
    if (wv >= 4) {
        const int lw = wv - 4;
        auto issue = [&](int k0, uint32_t lds_stage) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int j = lw * PER + i;                  // plane j / 8 (A0 A1 B0 B1), rows 4 (j % 8) .. + 3
                const int pq = j >> 3, blk = j & 7;
                const bool isA = pq < 2;
                const int row = blk * 4 + (lane >> 4), slot = lane & 15, src = slot ^ t_swz(row);
                const uint16_t *base = isA ? g.a[pq] : g.b[pq - 2];
                const int ld = isA ? g.M : g.N;
                const uint32_t voff = (uint32_t)((((int64_t)(k0 + row)) * ld + (isA ? m0 : n0) + src * 8) * 2);
                dma16(voff, base, lds_stage + (uint32_t)(pq * T_PLANE + blk * 1024));
            }
        };
#pragma unroll
        for (int c = 0; c < STAGES2; ++c) issue(kb + c * KC, lds0 + (uint32_t)(c * T_STAGE));
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * PER) : "memory");
        __builtin_amdgcn_s_barrier();
        for (int c = 0; c < nc; ++c) {
            if (c + STAGES2 <= nc) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PER) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (c + STAGES2 < nc) issue(kb + (c + STAGES2) * KC, lds0 + (uint32_t)((c % STAGES2) * T_STAGE));
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
    __builtin_amdgcn_s_barrier();
    // the transposed read of lane l: 16-lane group l / 16 -> column half (l / 16) & 1, k group l / 32; inside the group lane 4 q + p supplies
    // k-row q, columns 4 p .. 4 p + 3 (8 bytes: half p & 1 of the 16-byte slot p >> 1) and receives column l % 16, rows 0 .. 3
    const int q = (lane & 15) >> 2, p = lane & 3, ch = (lane >> 4) & 1, kg = lane >> 5;
    auto tr8 = [&](const unsigned char *plane, int col0, int krow0) {      // 8 consecutive k of column col0 + 16 ch + l % 16
        f16x8 out;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = krow0 + 4 * h + q;
            const int slot = ((col0 + 16 * ch + 4 * p) >> 3) ^ t_swz(row);
            const uint32_t addr = (uint32_t)(uintptr_t)plane + row * T_ROWB + slot * 16 + 8 * (p & 1);
            const s16x4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v *)(uintptr_t)addr);
            const f16x4 f = __builtin_bit_cast(f16x4, v);
            out[4 * h] = f[0]; out[4 * h + 1] = f[1]; out[4 * h + 2] = f[2]; out[4 * h + 3] = f[3];
        }
        return out;
    };
    for (int c = 0; c < nc; ++c) {
        const unsigned char *st = smem + (c % STAGES2) * T_STAGE;
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            f16x8 fa[2][2], fb[2][2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[pl][i] = tr8(st + pl * T_PLANE, wm + 32 * i, ks * 16 + 8 * kg);
                    fb[pl][i] = tr8(st + (2 + pl) * T_PLANE, wn + 32 * i, ks * 16 + 8 * kg);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    hi[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[0][j], hi[i][j], 0, 0, 0);
                    if (g.products >= 3) {
                        lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[1][j], lo[i][j], 0, 0, 0);
                        lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][i], fb[0][j], lo[i][j], 0, 0, 0);
                    }
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    float *out = g.cpart + (int64_t)s * g.M * g.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm + 32 * i + (e >> 2) * 8 + (lane >> 5) * 4 + (e & 3), col = n0 + wn + 32 * j + (lane & 31);
                out[(int64_t)row * g.N + col] = hi[i][j][e] + lo[i][j][e];
            }
}

}  // namespace

extern "C" int idl_debug_split_gemm(const void *a0, const void *a1, const void *a2, const void *b0, const void *b1, const void *b2,
                                    float *cpart, int M, int N, int K, int S, int products, void *stream)
{
    IDL_REQUIRE(a0 && a1 && a2 && b0 && b1 && b2 && cpart, "debug_split_gemm: NULL buffer");
    IDL_REQUIRE(M % TM == 0 && N % TN == 0 && S >= 1 && K % (S * KC) == 0 && (K / S) / KC >= 2, "debug_split_gemm: 128 | M, N; 32 S | K");
    const bool loaders = (products & 16) != 0;               // products + 16: the form with loader waves; + 32: two fp16 planes (3 or 1 products)
    const bool f16 = (products & 32) != 0;
    const bool transposed = (products & 64) != 0;           // + 64: operands [K][M], [K][N] (fp16 planes, loader waves)
    products &= 15;
    IDL_REQUIRE(!transposed || (f16 && loaders), "debug_split_gemm: the transposed form takes two fp16 planes");
    IDL_REQUIRE(!f16 || (loaders && products <= 3), "debug_split_gemm: the fp16 form has loader waves and at most three products");
    IDL_REQUIRE(products == 1 || products == 3 || products == 6, "debug_split_gemm: products = 1, 3 or 6 (+ 16: loader waves)");
    IDL_REQUIRE(!loaders || (K / S) / KC >= STAGES2, "debug_split_gemm: at least three chunks per workgroup");
    IDL_REQUIRE((int64_t)M * K < (1ll << 30) && (int64_t)N * K < (1ll << 30), "debug_split_gemm: 32-bit offsets");
    SplitArgs g{};
    g.a[0] = (const uint16_t *)a0; g.a[1] = (const uint16_t *)a1; g.a[2] = (const uint16_t *)a2;
    g.b[0] = (const uint16_t *)b0; g.b[1] = (const uint16_t *)b1; g.b[2] = (const uint16_t *)b2;
    g.cpart = cpart; g.M = M; g.N = N; g.K = K; g.S = S; g.products = products;
    static bool attr_set = false;
    if (!attr_set) {
        IDL_HIP_TRY(hipFuncSetAttribute((const void *)split_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, STAGES * STAGE));
        IDL_HIP_TRY(hipFuncSetAttribute((const void *)split_gemm_kernel2<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, STAGES2 * STAGE));
        IDL_HIP_TRY(hipFuncSetAttribute((const void *)split_gemm_kernel2<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, STAGES2 * STAGE));
        IDL_HIP_TRY(hipFuncSetAttribute((const void *)split_gemm_kernel_t, hipFuncAttributeMaxDynamicSharedMemorySize, STAGES2 * T_STAGE));
        attr_set = true;
    }
    if (transposed) hipLaunchKernelGGL(split_gemm_kernel_t, dim3((unsigned)((M / TM) * (N / TN) * S)), dim3(2 * NT), STAGES2 * T_STAGE, (hipStream_t)stream, g);
    else if (f16) hipLaunchKernelGGL((split_gemm_kernel2<2, true>), dim3((unsigned)((M / TM) * (N / TN) * S)), dim3(2 * NT), STAGES2 * STAGE, (hipStream_t)stream, g);
    else if (loaders) hipLaunchKernelGGL((split_gemm_kernel2<3, false>), dim3((unsigned)((M / TM) * (N / TN) * S)), dim3(2 * NT), STAGES2 * STAGE, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(split_gemm_kernel, dim3((unsigned)((M / TM) * (N / TN) * S)), dim3(NT), STAGES * STAGE, (hipStream_t)stream, g);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}
