// nce_fused.hip -- info_nce_loss (reference idelucs/LossFunctions.py:65-98) forward + gradient w.r.t. the
// normalised latents, fused on the fp32 matrix cores of gfx950; the [2B, 2B] similarity matrix is never
// written to memory.
//
//   f [m, 64]  rows L2-normalised (m = 2B; rows [0,B) view 1, [B,2B) view 2), S = f f^T
//   pass 1     rowsum_r = sum_{j != r} exp(S_rj / T)        (|S/T| <= 1/T: no max subtraction needed)
//              pos_r    = S_{r,(r+B) mod m} / T
//   pass 2     lse_r = log rowsum_r, loss_r = lse_r - pos_r,
//              G_r  = sum_{j != r} (exp(S_rj/T - lse_r) + exp(S_rj/T - lse_j)) f_j      == ((E + E^T) f)_r
//   so that d(mean loss)/df = (G - 2 f_pos) / (m T)  (consumed by head_bwd_kernel).
//
// Tiling: a workgroup owns 16 rows r and a quarter of the columns j; each of its 4 wavefronts walks 16-column
// tiles.  A tile is computed TRANSPOSED, S'[j][r] = sum_k f[j0+j][k] f[r0+r][k], with 16
// v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains).  In the C/D layout a lane then holds S'[j = 4q+reg][r = l]
// (q = lane>>4, l = lane&15), which is exactly the A operand of the second product G[r][c] += E[r][j] f[j][c]
// when its four k-steps are taken as j = 4q + reg -- no cross-lane movement, no LDS round trip.  The k order
// of both products is permuted the same way on the A and the B side (lane q owns k = 16q..16q+15 of the first
// product), which MFMA permits because it only sums over k.
#include <string.h>

#include "common.h"
#include "iic_device.h"
#include "nce_device.h"

namespace {

// Diagnostic phase stamps of the two InfoNCE passes (`make STAMPS=1`, tools/stamps_mid.py: modes 3 and 4 of idl_debug_phase_stamps); compiled out otherwise
#ifdef IDL_PHASE_STAMPS
__device__ uint64_t *nce_phase_stamps = nullptr;
__device__ int nce_phase_mode = 0;
#define NCE_PHASE_BUF(mode) ((nce_phase_mode == (mode) && nce_phase_stamps != nullptr && blockIdx.y == 0 && blockIdx.x < 64) ? nce_phase_stamps + 8 * blockIdx.x : nullptr)
#define NCE_PHASE_STAMP(buf, slot) do { if ((buf) != nullptr && threadIdx.x == 0) (buf)[(slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define NCE_PHASE_BUF(mode) nullptr
#define NCE_PHASE_STAMP(buf, slot) do { (void)(buf); } while (0)
#endif

constexpr int NCE_SPLIT = 8;     // column quarters -> (m/16) * 4 workgroups
constexpr int NCE_MAX_M = 2048;  // rows (2 * batch) whose lse fit the LDS table


using nce_dev::f32x4;
using nce_dev::sim_tile;
using nce_dev::load_rows;

// The IIC core is a single-workgroup computation that is independent of the InfoNCE branch: when asked (P0 != NULL) it
// rides along as ONE extra workgroup (blockIdx.x == m/16, blockIdx.y == 0) of pass 1 instead of a launch of its own.
struct IicJob { float *P0; int C; float lamb, eps, w_iic; float *scratch; float *out; const float *z; int cols; };      // cols: spare columns of pass 1 (0 = 1)

// The IIC joint P0 = z1^T z2 ([C, B] x [B, C], z1 = rows [0, B) of z, z2 = rows [B, 2B); reference LossFunctions.py:57-58 as
// one contraction) rides along too: a [C, C] product is far too small to be worth a GEMM launch of its own.  The spare
// workgroups of pass 1 (column blockIdx.x == m/16) each take 16 x 16 output tiles (v_mfma_f32_16x16x4_f32), the four waves
// splitting the batch, partial tiles added through LDS; the core then runs as the spare workgroup of PASS 2, when P0 is complete.
// A[i = c1][k = b] = z[b][i0 + l], B[k = b][j = c2] = z[B + b][j0 + l]; classes >= C enter as zeros and are not stored.
__device__ __forceinline__ void iic_joint_tiles(const float *z, int m, int C, float *P0, int first, int stride)
{
    __shared__ float red[4][256];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l = lane & 15, q = lane >> 4;
    const int B = m / 2, ct = (C + 15) / 16;
    const int per = ((B + 15) / 16) * 4;                     // rows per wave, a multiple of 4
    const int kb = wv * per, ke = (kb + per < B) ? kb + per : B;
    for (int tile = first; tile < ct * ct; tile += stride) {
        const int i0 = (tile / ct) * 16, j0 = (tile % ct) * 16;
        const bool ia = i0 + l < C, jb = j0 + l < C;
        const float *pa = z + (ia ? i0 + l : 0), *pb = z + (int64_t)B * C + (jb ? j0 + l : 0);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = kb; k0 < ke; k0 += 128) {              // 64 independent loads in flight per lane, then 32 MFMAs
            float av[32], bv[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const int k = k0 + 4 * u + q;
                const bool ok = k < ke;
                const int kc = ok ? k : ke - 1;
                const float ta = pa[kc * C], tb = pb[kc * C];
                av[u] = (ok && ia) ? ta : 0.f;
                bv[u] = (ok && jb) ? tb : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 32; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wv][(4 * q + r) * 16 + l] = acc[r];          // C/D: row = 4 q + reg, col = l
        __syncthreads();
        const int c1 = i0 + (tid >> 4), c2 = j0 + (tid & 15);
        if (c1 < C && c2 < C) P0[c1 * C + c2] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    }
}

// pass 1: partial row sums [NCE_SPLIT][m] and the positive logits pos[m]
__device__ __forceinline__ void nce_pass1_body(const float *f, int m, float inv_t, float *rowsum_part, float *pos, const IicJob &iic)
{
    if ((int)blockIdx.x >= m / 16) {
        // (ahead of the similarity waves it shares SIMDs with: beside a wave that issues fp32 matrix instructions back to back a
        //  wave of equal priority gets an issue slot only now and then -- stamps in the optimizer launch showed 6 us of such work
        //  taking 40)
        __builtin_amdgcn_s_setprio(3);
        const int cols = iic.cols > 0 ? iic.cols : 1;        // (n_clusters > 48, round 6: as many spare columns as the joint has tiles / NCE_SPLIT -- a tile each)
        if (iic.z != nullptr) iic_joint_tiles(iic.z, m, iic.C, iic.P0, ((int)blockIdx.x - m / 16) * NCE_SPLIT + (int)blockIdx.y, cols * NCE_SPLIT);     // core: pass 2
        else if (blockIdx.y == 0 && iic.P0 != nullptr) iic_core_small(iic.P0, iic.C, iic.lamb, iic.eps, iic.w_iic, iic.out);
        return;
    }
    __shared__ float sh[4][16];
    uint64_t *const stp = NCE_PHASE_BUF(3);
    NCE_PHASE_STAMP(stp, 0);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l = lane & 15, q = lane >> 4;
    const int r0 = blockIdx.x * 16, ntiles = m / 16;
    const int t0 = (int)((int64_t)blockIdx.y * ntiles / NCE_SPLIT), t1 = (int)((int64_t)(blockIdx.y + 1) * ntiles / NCE_SPLIT);
    float rb[16];
    load_rows(f, r0, l, q, rb);
    const int r = r0 + l, pr = (r + m / 2) % m;
    float sum = 0.f;
    auto fold = [&](const f32x4 s, int t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int j = t * 16 + 4 * q + g;
            const float x = s[g] * inv_t;
            if (j != r) sum += __expf(x);
            if (j == pr) pos[r] = x;
        }
    };
    int t = t0 + wv;
    for (; t + 4 < t1; t += 8) {              // two tiles per turn: loads of both in flight, MFMA chains interleaved
        f32x4 sa, sb;
        nce_dev::sim_tile2(f, t * 16, (t + 4) * 16, rb, l, q, sa, sb);
        fold(sa, t);
        fold(sb, t + 4);
    }
    if (t < t1) fold(sim_tile(f, t * 16, rb, l, q), t);
    // this lane holds the partial of row r over its j's (q, reg); add the four q groups, then the four waves
    NCE_PHASE_STAMP(stp, 1);                     // (wave 0: its tiles' loads arrived, products and exponentials done)
    sum = idl_dev::add_xor32(idl_dev::add_xor16(sum));
    if (q == 0) sh[wv][l] = sum;
    __syncthreads();
    NCE_PHASE_STAMP(stp, 2);
    if (threadIdx.x < 16) rowsum_part[(int64_t)blockIdx.y * m + r0 + threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
    NCE_PHASE_STAMP(stp, 3);
}

// pass 2: lse / loss rows, and the partial products G_part[NCE_SPLIT][m][64]
__device__ __forceinline__ void nce_pass2_body(const float *f, int m, float inv_t, const float *rowsum_part, const float *pos,
                                               float *lse, float *loss_rows, float *G_part, const IicJob &iic)
{
    if ((int)blockIdx.x == m / 16) {         // spare column (only launched when the joint was formed in pass 1): the IIC core
        __builtin_amdgcn_s_setprio(3);
        if (blockIdx.y == 0) iic_core_small(iic.P0, iic.C, iic.lamb, iic.eps, iic.w_iic, iic.out);
        return;
    }
    __shared__ float red[4][16][64];     // per-wave G tiles [row][c]
    __shared__ float lse_col[NCE_MAX_M / NCE_SPLIT + 16];   // lse of the columns this workgroup visits (for E^T) ...
    __shared__ float lse_own[16];                            // ... and of its own 16 rows
    uint64_t *const stp = NCE_PHASE_BUF(4);
    NCE_PHASE_STAMP(stp, 0);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l = lane & 15, q = lane >> 4;
    const int r0 = blockIdx.x * 16, ntiles = m / 16;
    const int t0 = (int)((int64_t)blockIdx.y * ntiles / NCE_SPLIT), t1 = (int)((int64_t)(blockIdx.y + 1) * ntiles / NCE_SPLIT);
    const int c0 = 16 * t0, ncol = 16 * (t1 - t0);
    for (int i = threadIdx.x; i < ncol + 16; i += 256) {     // only the rows this workgroup looks at, not all m of them
        const int row = i < ncol ? c0 + i : r0 + (i - ncol);
        float sm = 0.f;
#pragma unroll
        for (int p = 0; p < NCE_SPLIT; ++p) sm += rowsum_part[(int64_t)p * m + row];
        const float v = __logf(sm);
        if (i < ncol) lse_col[i] = v; else lse_own[i - ncol] = v;
    }
    float rb[16];
    load_rows(f, r0, l, q, rb);
    const int r = r0 + l;
    __syncthreads();
    NCE_PHASE_STAMP(stp, 1);                     // (the partial row sums read, their logarithms in LDS)
    const float lse_r = lse_own[l];
    if (blockIdx.y == 0 && wv == 0 && q == 0) { lse[r] = lse_r; loss_rows[r] = lse_r - pos[r]; }
    f32x4 g[4];                           // G[r = 4q+reg][c = 16*ct + l] for ct = 0..3
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) g[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    // G[r][c] += sum_j E[r][j] f[j][c]: A = E[r = l][k -> j = 4q + step], B = f[j0 + 4q + step][16 ct + l]
    auto weights = [&](const f32x4 s, int j0, float (&e)[4]) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int j = j0 + 4 * q + gq;
            const float x = s[gq] * inv_t;
            e[gq] = (j == r) ? 0.f : __expf(x - lse_r) + __expf(x - lse_col[j - c0]);
        }
    };
    int t = t0 + wv;
    for (; t + 4 < t1; t += 8) {              // two tiles per turn: every load of both tiles is issued before the first product
        const int ja = t * 16, jb = (t + 4) * 16;
        float fa[4][4], fb[4][4];             // B operands of the second product, [step][ct]
#pragma unroll
        for (int step = 0; step < 4; ++step)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                fa[step][ct] = f[(int64_t)(ja + 4 * q + step) * 64 + l + 16 * ct];
                fb[step][ct] = f[(int64_t)(jb + 4 * q + step) * 64 + l + 16 * ct];
            }
        f32x4 sa, sb;
        nce_dev::sim_tile2(f, ja, jb, rb, l, q, sa, sb);
        float ea[4], eb[4];
        weights(sa, ja, ea);
        weights(sb, jb, eb);
#pragma unroll
        for (int step = 0; step < 4; ++step)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                g[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ea[step], fa[step][ct], g[ct], 0, 0, 0);
                g[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(eb[step], fb[step][ct], g[ct], 0, 0, 0);
            }
    }
    if (t < t1) {
        const int j0 = t * 16;
        const f32x4 s = sim_tile(f, j0, rb, l, q);
        float e[4];
        weights(s, j0, e);
#pragma unroll
        for (int step = 0; step < 4; ++step) {
            const float *frow = f + (int64_t)(j0 + 4 * q + step) * 64 + l;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) g[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(e[step], frow[16 * ct], g[ct], 0, 0, 0);
        }
    }
    NCE_PHASE_STAMP(stp, 2);                     // (wave 0: both products of its tiles)
    // C/D layout of g[ct]: row = 4q + reg, col = l  ->  add the four waves through LDS, write [16][64]
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) red[wv][4 * q + reg][16 * ct + l] = g[ct][reg];
    __syncthreads();
    NCE_PHASE_STAMP(stp, 3);
    float *dst = G_part + ((int64_t)blockIdx.y * m + r0) * 64;
    for (int i = threadIdx.x; i < 16 * 64; i += 256) {
        const int rr = i >> 6, c = i & 63;
        dst[i] = (red[0][rr][c] + red[1][rr][c]) + (red[2][rr][c] + red[3][rr][c]);
    }
    NCE_PHASE_STAMP(stp, 4);
}

__global__ __launch_bounds__(256) void nce_pass1_kernel(const float *f, int m, float inv_t, float *rowsum_part, float *pos, IicJob iic)
{
    nce_pass1_body(f, m, inv_t, rowsum_part, pos, iic);
}

__global__ __launch_bounds__(256) void nce_pass2_kernel(const float *f, int m, float inv_t, const float *rowsum_part, const float *pos,
                                                        float *lse, float *loss_rows, float *G_part, IicJob iic)
{
    nce_pass2_body(f, m, inv_t, rowsum_part, pos, lse, loss_rows, G_part, iic);
}

// several voters in one launch: voter blockIdx.z takes its arguments from its plan record (common.h)
struct NceParams { const float *f; int m; float inv_t; float *rowsum_part, *pos, *lse, *loss_rows, *G_part; IicJob iic; };
static_assert(sizeof(NceParams) + idl::PLAN_PARAMS <= idl::PLAN_BYTES, "NceParams does not fit a plan record");

__global__ __launch_bounds__(256) void nce_pass1_batched_kernel(const unsigned char *__restrict__ plans)
{
    const NceParams &p = *(const NceParams *)(plans + (size_t)blockIdx.z * idl::PLAN_BYTES + idl::PLAN_PARAMS);
    nce_pass1_body(p.f, p.m, p.inv_t, p.rowsum_part, p.pos, p.iic);
}

__global__ __launch_bounds__(256) void nce_pass2_batched_kernel(const unsigned char *__restrict__ plans)
{
    const NceParams &p = *(const NceParams *)(plans + (size_t)blockIdx.z * idl::PLAN_BYTES + idl::PLAN_PARAMS);
    nce_pass2_body(p.f, p.m, p.inv_t, p.rowsum_part, p.pos, p.lse, p.loss_rows, p.G_part, p.iic);
}


// The IIC joint P0 = z[0:m/2]^T z[m/2:m] (reference LossFunctions.py:57-58) for any n_clusters: a workgroup of four waves per 16 x 16
// tile, the batch dimension dealt to the waves and taken in steps of four through v_mfma_f32_16x16x4_f32.  For n_clusters <= 48 the spare workgroups of InfoNCE pass 1 do this;
// above, it was a [C, m/2] x [m/2, C] GEMM whose library heuristic (untuned: short jobs) runs it on ONE workgroup, 118 us at C = 200.
__global__ __launch_bounds__(256) void iic_joint_kernel(const float *__restrict__ z, int m, int C, float *__restrict__ P0)
{
    __shared__ float part[3][64][4];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, l = lane & 15, q = lane >> 4;
    const int c1 = blockIdx.x * 16, c2 = blockIdx.y * 16, half = m / 2;
    const bool ok1 = c1 + l < C, ok2 = c2 + l < C;
    // the batch rows are dealt to the four waves in blocks of 128 (32 MFMA steps: every load of a block is in flight before its first use)
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int kb = 128 * wv; kb < half; kb += 512) {
        const float *za = z + (size_t)(kb + q) * C + (ok1 ? c1 + l : 0);             // A[i = l][k = q] = z[b = kb + 4 u + q][c1 + l]
        const float *zb = z + (size_t)(half + kb + q) * C + (ok2 ? c2 + l : 0);      // B[k = q][j = l] = z[m/2 + b][c2 + l]
        float av[32], bv[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const bool in = kb + 4 * u + q < half;
            av[u] = (ok1 && in) ? za[(size_t)(4 * u) * C] : 0.f;
            bv[u] = (ok2 && in) ? zb[(size_t)(4 * u) * C] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 32; u += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u + 1], bv[u + 1], acc1, 0, 0, 0);
        }
    }
    f32x4 acc = acc0 + acc1;
    if (wv > 0) { part[wv - 1][lane][0] = acc[0]; part[wv - 1][lane][1] = acc[1]; part[wv - 1][lane][2] = acc[2]; part[wv - 1][lane][3] = acc[3]; }
    __syncthreads();
    if (wv == 0 && ok2) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int r = c1 + 4 * q + reg;
            if (r < C) P0[(size_t)r * C + c2 + l] = ((acc[reg] + part[0][lane][reg]) + part[1][lane][reg]) + part[2][lane][reg];
        }
    }
}

// out[M x N] = A^T B for A [K x M] and B [K x N] row-major (the shape of a weight gradient dy^T x with few outputs): iic_joint_kernel's tiles for any two
// operands.  Round 6: dW3 = dlogits^T r2 ([C x m] x [m x 64]) at n_clusters > 48, where the library's kernel for the transposed-A product took 9 us.
__global__ __launch_bounds__(256) void at_b_kernel(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb, int K, int M, int N,
                                                   float *__restrict__ out, int ldo)
{
    __shared__ float part[3][64][4];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, l = lane & 15, q = lane >> 4;
    const int c1 = blockIdx.x * 16, c2 = blockIdx.y * 16;
    const bool ok1 = c1 + l < M, ok2 = c2 + l < N;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int kb = 128 * wv; kb < K; kb += 512) {             // the contraction index is dealt to the four waves in blocks of 128 (32 MFMA steps, every load in flight first)
        const float *pa = A + (ok1 ? c1 + l : 0), *pb = B + (ok2 ? c2 + l : 0);
        float av[32], bv[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) {                       // (clamped, not predicated)
            const int k = kb + 4 * u + q, kk = k < K ? k : K - 1;
            av[u] = pa[(size_t)kk * lda]; bv[u] = pb[(size_t)kk * ldb];
        }
#pragma unroll
        for (int u = 0; u < 32; u += 2) {
            const bool in0 = kb + 4 * u + q < K, in1 = kb + 4 * (u + 1) + q < K;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32((ok1 && in0) ? av[u] : 0.f, (ok2 && in0) ? bv[u] : 0.f, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32((ok1 && in1) ? av[u + 1] : 0.f, (ok2 && in1) ? bv[u + 1] : 0.f, acc1, 0, 0, 0);
        }
    }
    f32x4 acc = acc0 + acc1;
    if (wv > 0) { part[wv - 1][lane][0] = acc[0]; part[wv - 1][lane][1] = acc[1]; part[wv - 1][lane][2] = acc[2]; part[wv - 1][lane][3] = acc[3]; }
    __syncthreads();
    if (wv == 0 && ok2) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int r = c1 + 4 * q + reg;
            if (r < M) out[(size_t)r * ldo + c2 + l] = ((acc[reg] + part[0][lane][reg]) + part[1][lane][reg]) + part[2][lane][reg];
        }
    }
}

}  // namespace

extern "C" {

// out[M x N] = A^T B, A [K x M] (lda), B [K x N] (ldb), all row-major fp32; a weight gradient with few outputs (dW3 = dlogits^T r2: models.py:130's backward
// through Linear(64, C)) as 16 x 16 MFMA tiles, the sum over K in a fixed order
int idl_at_b(const float *A, int lda, const float *B, int ldb, int K, int M, int N, float *out, int ldo, void *stream)
{
    IDL_REQUIRE(A && B && out && K >= 1 && M >= 1 && N >= 1 && lda >= M && ldb >= N && ldo >= N && M <= 65535 * 16 && N <= 65535 * 16, "at_b: NULL buffer or bad shape");
    hipLaunchKernelGGL(at_b_kernel, dim3((unsigned)((M + 15) / 16), (unsigned)((N + 15) / 16)), dim3(256), 0, (hipStream_t)stream, A, lda, B, ldb, K, M, N, out, ldo);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

int64_t idl_nce_fused_workspace(int m)
{
    if (m < 32 || (m % 32) != 0 || m > NCE_MAX_M) return -1;     // unsupported shape: use idl_nce_rows + GEMMs
    return ((int64_t)NCE_SPLIT * m + m) * 4;          // partial row sums + positive logits
}

int idl_nce_fused_parts(void) { return NCE_SPLIT; }

#ifdef IDL_PHASE_STAMPS
int nce_set_phase_stamps(uint64_t *st, int on)      // (train_step.hip: idl_debug_phase_stamps)
{
    IDL_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(nce_phase_stamps), &st, sizeof(st)));
    IDL_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(nce_phase_mode), &on, sizeof(on)));
    return IDL_OK;
}
#endif

static int nce_launch(const float *f, int m, float temperature, float *lse, float *loss_rows, float *G_part, void *workspace,
                      const IicJob &iic, void *stream)
{
    IDL_REQUIRE(f && lse && loss_rows && G_part && workspace, "NULL buffer");
    IDL_REQUIRE(m >= 32 && (m % 32) == 0 && m <= NCE_MAX_M && temperature > 0.f, "nce_fused: m must be a multiple of 32 in 32..2048, T > 0");
    IDL_REQUIRE((((uintptr_t)f) & 15u) == 0, "f must be 16-byte aligned");
    float *rowsum_part = (float *)workspace, *pos = rowsum_part + (size_t)NCE_SPLIT * m;
    const float inv_t = 1.f / temperature;
    const dim3 grid((unsigned)(m / 16), NCE_SPLIT);
    const dim3 grid1((unsigned)(m / 16 + (iic.P0 != nullptr ? (iic.cols > 0 ? iic.cols : 1) : 0)), NCE_SPLIT);
    const bool core2 = iic.z != nullptr && iic.cols == 0;    // the core as the spare workgroup of pass 2 (n_clusters <= 48)
    if (void *plan = idl::take_plan()) {          // recorded, not launched (idl_plan_begin): both passes in one record
        IDL_REQUIRE(iic.cols == 0, "nce_fused: the joint of n_clusters > 48 cannot be recorded");
        const dim3 g2 = iic.z != nullptr ? grid1 : grid;
        idl::PlanHead h{};
        h.kind = idl::PLAN_NCE; h.grid[0] = grid1.x; h.grid[1] = grid1.y; h.grid[2] = 1; h.block = 256;
        h.grid2[0] = g2.x; h.grid2[1] = g2.y; h.grid2[2] = 1;
        memcpy(plan, &h, sizeof(h));
        const NceParams p{f, m, inv_t, rowsum_part, pos, lse, loss_rows, G_part, iic};
        memcpy((unsigned char *)plan + idl::PLAN_PARAMS, &p, sizeof(p));
        return IDL_OK;
    }
    hipLaunchKernelGGL(nce_pass1_kernel, grid1, dim3(256), 0, (hipStream_t)stream, f, m, inv_t, rowsum_part, pos, iic);
    hipLaunchKernelGGL(nce_pass2_kernel, core2 ? grid1 : grid, dim3(256), 0, (hipStream_t)stream, f, m, inv_t,
                       (const float *)rowsum_part, (const float *)pos, lse, loss_rows, G_part, iic);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

int idl_nce_fused(const float *f, int m, float temperature, float *lse, float *loss_rows, float *G_part, void *workspace,
                  void *stream)
{
    return nce_launch(f, m, temperature, lse, loss_rows, G_part, workspace, IicJob{}, stream);
}

int idl_nce_fused_iic(const float *f, int m, float temperature, float *lse, float *loss_rows, float *G_part, void *workspace,
                      float *P0, int C, float lamb, float eps, float w_iic, float *iic_scratch, float *out, void *stream)
{
    IDL_REQUIRE(P0 && iic_scratch && out && C >= 1 && C <= 48, "nce_fused_iic: n_clusters must be in 1..48 (larger: idl_iic_core)");
    return nce_launch(f, m, temperature, lse, loss_rows, G_part, workspace, IicJob{P0, C, lamb, eps, w_iic, iic_scratch, out, nullptr, 0}, stream);
}

// the InfoNCE passes with the IIC JOINT of n_clusters > 48 formed by spare workgroups of pass 1, a 16 x 16 tile each (the core: idl_iic_core / idl_iic_core_dz)
int idl_nce_fused_joint(const float *f, int m, float temperature, float *lse, float *loss_rows, float *G_part, void *workspace, const float *z, float *P0, int C,
                        void *stream)
{
    IDL_REQUIRE(z && P0 && C >= 1 && C <= 256, "nce_fused_joint: NULL buffer or n_clusters outside 1..256");
    const int ct = (C + 15) / 16;
    return nce_launch(f, m, temperature, lse, loss_rows, G_part, workspace, IicJob{P0, C, 0.f, 0.f, 0.f, nullptr, nullptr, z, (ct * ct + NCE_SPLIT - 1) / NCE_SPLIT}, stream);
}

int idl_iic_joint(const float *z, int m, int C, float *P0, void *stream)
{
    IDL_REQUIRE(z && P0 && C >= 1 && C <= 4096 && m >= 2 && (m % 2) == 0, "iic_joint: NULL buffer, n_clusters outside 1..4096 or an odd batch");
    const unsigned t = (unsigned)((C + 15) / 16);
    hipLaunchKernelGGL(iic_joint_kernel, dim3(t, t), dim3(256), 0, (hipStream_t)stream, z, m, C, P0);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

int idl_nce_fused_iic_z(const float *f, int m, float temperature, float *lse, float *loss_rows, float *G_part, void *workspace,
                        const float *z, float *P0, int C, float lamb, float eps, float w_iic, float *iic_scratch, float *out, void *stream)
{
    IDL_REQUIRE(z && P0 && iic_scratch && out && C >= 1 && C <= 48, "nce_fused_iic_z: n_clusters must be in 1..48 (larger: a GEMM + idl_iic_core)");
    return nce_launch(f, m, temperature, lse, loss_rows, G_part, workspace, IicJob{P0, C, lamb, eps, w_iic, iic_scratch, out, z, 0}, stream);
}

}  // extern "C"

int idl::nce_plan_launch(const idl::PlanHead &h, const void *dev_plans, int n_voters, hipStream_t stream)
{
    const unsigned char *dp = (const unsigned char *)dev_plans;
    hipLaunchKernelGGL(nce_pass1_batched_kernel, dim3(h.grid[0], h.grid[1], (unsigned)n_voters), dim3(h.block), 0, stream, dp);
    hipLaunchKernelGGL(nce_pass2_batched_kernel, dim3(h.grid2[0], h.grid2[1], (unsigned)n_voters), dim3(h.block), 0, stream, dp);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}
