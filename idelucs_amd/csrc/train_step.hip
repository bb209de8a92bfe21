// train_step.hip -- the non-GEMM pieces of one contrastive-IIC optimizer step, fused for gfx950.
//
// The reference step (idelucs/models.py:117-133) is ~160 small PyTorch kernels around four dense
// layers; on MI355X that is launch-bound (0.1 ms of GEMM inside a 1.8 ms step).  Here the step is an
// explicit forward/backward: the dense products stay on hipBLASLt (MFMA) and every other piece is
// one of the kernels below, so a step is ~20 launches that replay as a HIP graph.
//
//   relu_dropout_fwd      nn.ReLU + nn.Dropout(0.5) after Linear(F,512)        PytorchUtils.py:40-41
//   head_fwd              latent -> L2-normalise (LossFunctions.py:79); ReLU + Dropout + Linear(64,C)
//                         + Softmax (PytorchUtils.py:45-50)
//   nce_lse / nce_esym    info_nce_loss forward and d/dS without masks         LossFunctions.py:65-98
//   iic_core              IID_loss value and d/dP of the C x C joint           LossFunctions.py:20-62
//   head_bwd              softmax/Linear/Dropout/ReLU backward + normalise backward, per row
//   col_sum, relu_dropout_bwd_colsum   bias gradients, ReLU/Dropout backward
//   rmsprop_step          torch.optim.RMSprop(lr, alpha=.99, eps=1e-8, weight_decay=.01) models.py:88
//
// Dropout uses Philox4x32-10 keyed by a seed with counter (element block, layer, step): the mask is
// never stored -- a kept, active unit is exactly one whose output is > 0, which is all the backward
// needs.  `ctl` is a small device-resident control block so that graph replays advance without
// host involvement:  ctl[0] = optimizer-step counter, ctl[1] = offset of the next batch in the
// shuffled pair list.
#include <stdlib.h>
#include "dev_env.h"
#include <string.h>

#include "common.h"
#include "iic_device.h"
#include "nce_device.h"
#include "scaler_device.h"
#include "wave_ops.h"
#include "wgrad_device.h"
#include "l1_device.h"
#include "l1_planes_device.h"
#include "wgrad_planes_device.h"
#include "philox_device.h"

namespace {

constexpr int H2 = 64;        // latent width of NetLinear == one wavefront
constexpr int MAX_CPL = 4;    // classes per lane: n_clusters <= 256
constexpr int COL_PARTS = 64; // row chunks of the partial column sums (bias gradients); 16 rows each at m = 1024

using idl_dev::U4;
using idl_dev::philox;        // philox_device.h: shared with l1_fwd.hip, which takes over mid_fwd_kernel's layer-1 dropout

// Diagnostic phase stamps of the two middle kernels (a build with -DIDL_PHASE_STAMPS: `make STAMPS=1`; tools/stamps_mid.py): when
// armed (idl_debug_phase_stamps), workgroups 0..63 of the stamped kernel leave up to eight s_memrealtime marks (100 MHz) each in
// the stamp buffer.  Compiled out otherwise: the two scalar loads that arm them sit on the kernels' critical path (+1.2 us).
#ifdef IDL_PHASE_STAMPS
__device__ uint64_t *idl_phase_stamps = nullptr;
__device__ int idl_phase_mode = 0;           // 1: the mid-forward kernel stamps, 2: the mid-backward kernel
#define IDL_PHASE_BUF(mode, bid) ((idl_phase_mode == (mode) && idl_phase_stamps != nullptr && (bid) < 64) ? idl_phase_stamps + 8 * (bid) : nullptr)
#define IDL_PHASE_STAMP(buf, slot) do { if ((buf) != nullptr && threadIdx.x == 0) (buf)[(slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define IDL_PHASE_BUF(mode, bid) nullptr
#define IDL_PHASE_STAMP(buf, slot) do { (void)(buf); } while (0)
#endif

__device__ __forceinline__ float wave_sum(float v) { return idl_dev::wave_sum_f(v); }
__device__ __forceinline__ float wave_max(float v) { return idl_dev::wave_max_f(v); }

// ---------------------------------------------------------------- ReLU + Dropout(0.5), in place
__global__ __launch_bounds__(256) void relu_dropout_fwd_kernel(float4 *a, int64_t n4, int train, uint64_t seed, const int64_t *ctl,
                                                               uint32_t layer)
{
    const uint32_t step = (uint32_t)ctl[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = a[i];
        float s0 = 1.f, s1 = 1.f, s2 = 1.f, s3 = 1.f;
        if (train) {
            const U4 r = philox((uint32_t)i, layer, step, (uint32_t)(i >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
            s0 = (r.x >> 31) ? 2.f : 0.f; s1 = (r.y >> 31) ? 2.f : 0.f;
            s2 = (r.z >> 31) ? 2.f : 0.f; s3 = (r.w >> 31) ? 2.f : 0.f;
        }
        v.x = v.x > 0.f ? v.x * s0 : 0.f; v.y = v.y > 0.f ? v.y * s1 : 0.f;
        v.z = v.z > 0.f ? v.z * s2 : 0.f; v.w = v.w > 0.f ? v.w * s3 : 0.f;
        a[i] = v;
    }
}

// ---------------------------------------------------------------- head forward: one wave per row
// x = this lane's latent value lat[row][lane]  ->  f = lat/max(||lat||,1e-12), inv; r2 = dropout(relu(lat));
// z = softmax(r2 W3^T + b3).  `sh` = 64 floats of LDS private to the calling wave.
__device__ __forceinline__ void head_row(float x, int row, int lane, float *sh, const float *W3, const float *b3, int C, int train,
                                         uint64_t seed, uint32_t step, float *f, float *inv, float *r2, float *z)
{
    const float nrm = fmaxf(sqrtf(wave_sum(x * x)), 1e-12f);       // F.normalize(dim=1), eps 1e-12
    f[(int64_t)row * H2 + lane] = x / nrm;
    if (lane == 0) inv[row] = 1.f / nrm;
    float s = 1.f;
    if (train) {
        const U4 r = philox((uint32_t)row, 2u, step, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
        const uint32_t word = (lane < 32) ? r.x : r.y;
        s = ((word >> (lane & 31)) & 1u) ? 2.f : 0.f;
    }
    const float a = x > 0.f ? x * s : 0.f;
    r2[(int64_t)row * H2 + lane] = a;
    sh[lane] = a;
    __builtin_amdgcn_wave_barrier();
    float lg[MAX_CPL];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < MAX_CPL; ++t) {
        const int c = t * 64 + lane;
        lg[t] = -INFINITY;
        if (c < C) {
            float acc = b3[c];
            const float *wr = W3 + (int64_t)c * H2;
#pragma unroll 8
            for (int h = 0; h < H2; ++h) acc = fmaf(sh[h], wr[h], acc);
            lg[t] = acc;
            mx = fmaxf(mx, acc);
        }
    }
    mx = wave_max(mx);
    float den = 0.f;
#pragma unroll
    for (int t = 0; t < MAX_CPL; ++t) {
        const int c = t * 64 + lane;
        if (c < C) { lg[t] = __expf(lg[t] - mx); den += lg[t]; }
    }
    den = wave_sum(den);
#pragma unroll
    for (int t = 0; t < MAX_CPL; ++t) {
        const int c = t * 64 + lane;
        if (c < C) z[(int64_t)row * C + c] = lg[t] / den;
    }
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256) void head_fwd_kernel(const float *lat, const float *W3, const float *b3, int m, int C, int train,
                                                       uint64_t seed, const int64_t *ctl, float *f, float *inv, float *r2, float *z)
{
    __shared__ float sh[4][H2];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + w;
    if (row >= m) return;
    head_row(lat[(int64_t)row * H2 + lane], row, lane, sh[w], W3, b3, C, train, seed, (uint32_t)ctl[0], f, inv, r2, z);
}

// ---------------------------------------------------------------- fused middle of the forward pass (fp32 MFMA)
// a1[m,512] = Linear(F,512) output (bias included).  One 1024-thread workgroup per 16-row tile:
//   r1 = Dropout(ReLU(a1))  (written back in place: the backward needs it)
//   lat = r1 W2^T + b2      (v_mfma_f32_16x16x4_f32; the 16 waves split K = 512 into 32-wide slices -- each lane owns 8
//                            consecutive k of one row, so A is read and masked exactly once -- partial tiles added through LDS)
//   head_row() one row per wave (normalise, ReLU/Dropout, Linear(64,C), softmax)
// replacing relu_dropout_fwd + one hipBLASLt GEMM + head_fwd.  Same dropout streams as those kernels.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int H1 = 512;
constexpr int MID_WAVES = 16;
constexpr int MID_GATHER_ROWS = 4;   // rows per 256-thread gather tile when the batch assembly rides in mid_fwd / mid_bwd

// (Round 5's variant that added idl_l1_planes' eight K-slice partial sums up HERE -- 16 MB read by 64 workgroups: +8 us -- left in round 6: idl_reduce_parts_rms does it on every CU.)
template <bool TIN>       // TIN: a1 is stored transposed, [512, m] (the orientation hipBLASLt runs the layer-1 product fastest in)
__device__ __forceinline__ void mid_fwd_body(float *__restrict__ a1, const float *__restrict__ b1, const float *__restrict__ W2,
                                                                  const float *__restrict__ b2, const float *__restrict__ W3,
                                                                  const float *__restrict__ b3, int m, int C, int train, uint64_t seed,
                                                                  const int64_t *__restrict__ ctl, float *__restrict__ f,
                                                                  float *__restrict__ inv, float *__restrict__ r2, float *__restrict__ z,
                                                                  int n_rows_wg, int tile0, int tile1, const idl_dev::GatherArgs &gth, const int bid)
{
    if (bid >= n_rows_wg) {          // spare workgroups: tiles [tile0, tile1) of the next batch (see mid_bwd_kernel)
        const int blk = tile0 + (bid - n_rows_wg) * 4 + (int)(threadIdx.x >> 8);
        if (blk < tile1) idl_dev::gather_tile<MID_GATHER_ROWS>(gth, (int64_t)blk, (int)(threadIdx.x & 255));
        return;
    }
    // part[wave][row][col'], col' = (col + 16 (row >> 2)) & 63: the four row-quads of one MFMA store hit disjoint banks.
    // Once the K-slices are added up the same memory holds the r2 tile (A operand of the logits) and the logits tile.
    __shared__ float part[MID_WAVES][16][H2];
    float (*R2t)[H2 + 4] = (float (*)[H2 + 4])&part[0][0][0];               // [16][68]
    float (*LG)[64 * MAX_CPL + 1] = (float (*)[64 * MAX_CPL + 1])&part[2][0][0];   // [16][257]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l = lane & 15, q = lane >> 4;
    const int r0 = bid * 16;
    uint64_t *stf = IDL_PHASE_BUF(1, bid);
    IDL_PHASE_STAMP(stf, 0);
    const int k0 = 32 * wv + 8 * q;              // this lane's 8 consecutive k of row r0 + l
    // ---- every global read of the kernel is issued here, before the first dependent instruction
    float4 *src = (float4 *)(a1 + (int64_t)(r0 + l) * H1 + k0);
    float *srcT = a1 + (int64_t)k0 * m + r0 + l;             // element (row r0 + l, column k0 + i) of the transposed image: srcT[i * m]
    float4 av[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    float4 bb[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    float4 bw[4][2];
    auto load_bw = [&]() {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {             // lat[:, 16 ct .. 16 ct + 15]: B[k][c] = W2[16 ct + c][k]
            const float4 *wsrc = (const float4 *)(W2 + (int64_t)(16 * ct + l) * H1 + k0);
            bw[ct][0] = wsrc[0]; bw[ct][1] = wsrc[1];
        }
    };
    float t8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (TIN) {
#pragma unroll
        for (int i = 0; i < 8; ++i) t8[i] = srcT[(int64_t)i * m];
        av[0] = make_float4(t8[0], t8[1], t8[2], t8[3]); av[1] = make_float4(t8[4], t8[5], t8[6], t8[7]);
    } else {
        av[0] = src[0]; av[1] = src[1];
    }
    if (b1 != nullptr) { bb[0] = *(const float4 *)(b1 + k0); bb[1] = *(const float4 *)(b1 + k0 + 4); }
    load_bw();
    const int nct = (C + 15) / 16;               // column tiles of the logits; wave wv < nct owns tile wv
    float4 w3f[4];                               // B[k = 16 q + s][c = l] = W3[16 wv + l][16 q + s]
    float b3v = 0.f;
    if (wv < nct) {
        const int c = min(16 * wv + l, C - 1);
        const float4 *wsrc = (const float4 *)(W3 + (int64_t)c * H2 + 16 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) w3f[i] = wsrc[i];
        b3v = b3[c];
    }
    const float b2v = b2[lane];
    const uint32_t step = (uint32_t)ctl[0];
    {
    // ---- ReLU + Dropout of layer 1, in place
    float a[8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float4 v = av[i];
        v.x += bb[i].x; v.y += bb[i].y; v.z += bb[i].z; v.w += bb[i].w;      // Linear(F,512) bias when the product came without it
        float s0 = 1.f, s1 = 1.f, s2 = 1.f, s3 = 1.f;
        if (train) {                             // identical stream to relu_dropout_fwd_kernel: counter = float4 index of the flat array
            const int64_t idx4 = ((int64_t)(r0 + l) * H1 + k0) / 4 + i;
            const U4 r = philox((uint32_t)idx4, 1u, step, (uint32_t)(idx4 >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
            s0 = (r.x >> 31) ? 2.f : 0.f; s1 = (r.y >> 31) ? 2.f : 0.f;
            s2 = (r.z >> 31) ? 2.f : 0.f; s3 = (r.w >> 31) ? 2.f : 0.f;
        }
        v.x = v.x > 0.f ? v.x * s0 : 0.f; v.y = v.y > 0.f ? v.y * s1 : 0.f;
        v.z = v.z > 0.f ? v.z * s2 : 0.f; v.w = v.w > 0.f ? v.w * s3 : 0.f;
        if (TIN) { srcT[(int64_t)(4 * i) * m] = v.x; srcT[(int64_t)(4 * i + 1) * m] = v.y; srcT[(int64_t)(4 * i + 2) * m] = v.z; srcT[(int64_t)(4 * i + 3) * m] = v.w; }
        else src[i] = v;
        a[4 * i] = v.x; a[4 * i + 1] = v.y; a[4 * i + 2] = v.z; a[4 * i + 3] = v.w;
    }
    IDL_PHASE_STAMP(stf, 1);                     // (wave 0: its loads have arrived, ReLU / Dropout done)
    // ---- lat = r1 W2^T: this wave's K-slice
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4 b = bw[ct][i];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * i], b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * i + 1], b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * i + 2], b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * i + 3], b.w, acc, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) part[wv][4 * q + reg][(16 * ct + l + 16 * q) & 63] = acc[reg];   // C/D: row = 4q+reg, col = l
    }
    IDL_PHASE_STAMP(stf, 2);
    __syncthreads();
    IDL_PHASE_STAMP(stf, 3);
    }
    // ---- wave wv owns row wv of the tile: add the 16 K-slices and the bias; normalise; ReLU + Dropout of the latent
    const int row = r0 + wv, cs = (lane + 16 * (wv >> 2)) & 63;
    float x = b2v;
#pragma unroll
    for (int w = 0; w < MID_WAVES; ++w) x += part[w][wv][cs];
    const float nrm = fmaxf(sqrtf(wave_sum(x * x)), 1e-12f);       // F.normalize(dim=1), eps 1e-12
    float sc = 1.f;
    if (train) {                                 // as head_row
        const U4 r = philox((uint32_t)row, 2u, step, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
        const uint32_t word = (lane < 32) ? r.x : r.y;
        sc = ((word >> (lane & 31)) & 1u) ? 2.f : 0.f;
    }
    const float ar = x > 0.f ? x * sc : 0.f;
    if (row < m) {
        f[(int64_t)row * H2 + lane] = x / nrm;
        if (lane == 0) inv[row] = 1.f / nrm;
        r2[(int64_t)row * H2 + lane] = ar;
    }
    __syncthreads();                             // part is dead
    R2t[wv][lane] = ar;
    __syncthreads();
    IDL_PHASE_STAMP(stf, 4);
    // ---- logits tile wv = r2[16 x 64] W3[16 wv .. 16 wv + 15]^T + b3
    if (wv < nct) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 t = *(const float4 *)&R2t[l][16 * q + 4 * i];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t.x, w3f[i].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t.y, w3f[i].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t.z, w3f[i].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(t.w, w3f[i].w, acc, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) LG[4 * q + reg][16 * wv + l] = acc[reg] + b3v;
    }
    __syncthreads();
    IDL_PHASE_STAMP(stf, 5);
    // ---- softmax of row wv
    float lg[MAX_CPL];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < MAX_CPL; ++t) {
        const int c = t * 64 + lane;
        lg[t] = c < C ? LG[wv][c] : -INFINITY;
        mx = fmaxf(mx, lg[t]);
    }
    mx = wave_max(mx);
    float den = 0.f;
#pragma unroll
    for (int t = 0; t < MAX_CPL; ++t) {
        const int c = t * 64 + lane;
        if (c < C) { lg[t] = __expf(lg[t] - mx); den += lg[t]; }
    }
    den = wave_sum(den);
    if (row < m) {
#pragma unroll
        for (int t = 0; t < MAX_CPL; ++t) {
            const int c = t * 64 + lane;
            if (c < C) z[(int64_t)row * C + c] = lg[t] / den;
        }
    }
    IDL_PHASE_STAMP(stf, 6);
}
struct MidFwdParams {
    float *a1; const float *b1, *W2, *b2, *W3, *b3; int m, C, train; uint64_t seed; const int64_t *ctl; float *f, *inv, *r2, *z;
    int n_rows_wg, tile0, tile1; idl_dev::GatherArgs gth;
};
static_assert(sizeof(MidFwdParams) + idl::PLAN_PARAMS <= idl::PLAN_BYTES, "MidFwdParams does not fit a plan record");

template <bool TIN>
__global__ __launch_bounds__(64 * MID_WAVES) void mid_fwd_kernel(float *__restrict__ a1, const float *__restrict__ b1, const float *__restrict__ W2,
                                                                  const float *__restrict__ b2, const float *__restrict__ W3,
                                                                  const float *__restrict__ b3, int m, int C, int train, uint64_t seed,
                                                                  const int64_t *__restrict__ ctl, float *__restrict__ f,
                                                                  float *__restrict__ inv, float *__restrict__ r2, float *__restrict__ z,
                                                                  int n_rows_wg, int tile0, int tile1, idl_dev::GatherArgs gth)
{
    mid_fwd_body<TIN>(a1, b1, W2, b2, W3, b3, m, C, train, seed, ctl, f, inv, r2, z, n_rows_wg, tile0, tile1, gth, (int)blockIdx.x);
}

// the same for several voters in one launch: the grid is (voters, workgroups of one voter) -- the VOTER index runs fastest, so the
// computing workgroups of every voter are dispatched before anybody's batch-assembly workgroups (a workgroup of this kernel fills a
// CU: voter by voter, the second half of the voters would wait behind the first half's streaming workgroups); each voter takes its
// arguments from its plan record (common.h)
template <bool TIN>
__global__ __launch_bounds__(64 * MID_WAVES) void mid_fwd_batched_kernel(const unsigned char *__restrict__ plans)
{
    const MidFwdParams &p = *(const MidFwdParams *)(plans + (size_t)blockIdx.x * idl::PLAN_BYTES + idl::PLAN_PARAMS);
    mid_fwd_body<TIN>(p.a1, p.b1, p.W2, p.b2, p.W3, p.b3, p.m, p.C, p.train, p.seed, p.ctl, p.f, p.inv, p.r2, p.z, p.n_rows_wg, p.tile0, p.tile1, p.gth,
                      (int)blockIdx.y);
}

// ---------------------------------------------------------------- InfoNCE on S = f f^T (un-scaled)
// per row r: lse_r = logsumexp_{j != r} S_rj / T;  loss_r = lse_r - S_r,pos(r) / T,  pos(r) = (r + m/2) mod m
__global__ __launch_bounds__(256) void nce_lse_kernel(const float *S, int m, float inv_t, float *lse, float *loss_rows)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + w;
    if (row >= m) return;
    const float *s = S + (int64_t)row * m;
    float l;
    if (m <= 1024) {                    // one pass over the row: 16 values per lane stay in registers
        float v[16];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = lane + 64 * i;
            v[i] = (j < m && j != row) ? s[j] * inv_t : -INFINITY;
            mx = fmaxf(mx, v[i]);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += __expf(v[i] - mx);      // exp(-inf) = 0 for the masked entries
        l = mx + __logf(wave_sum(sum));
    } else {
        float mx = -INFINITY;
        for (int j = lane; j < m; j += 64) if (j != row) mx = fmaxf(mx, s[j] * inv_t);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int j = lane; j < m; j += 64) if (j != row) sum += __expf(s[j] * inv_t - mx);
        l = mx + __logf(wave_sum(sum));
    }
    if (lane == 0) {
        lse[row] = l;
        loss_rows[row] = l - s[(row + m / 2) % m] * inv_t;
    }
}

// S <- E + E^T with E_rj = exp(S_rj/T - lse_r) (j != r), 0 on the diagonal; S is symmetric
__global__ __launch_bounds__(256) void nce_esym_kernel(float *S, int m, float inv_t, const float *lse)
{
    const int64_t total = (int64_t)m * m;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / m), j = (int)(i - (int64_t)r * m);
        const float v = S[i] * inv_t;
        S[i] = (r == j) ? 0.f : __expf(v - lse[r]) + __expf(v - lse[j]);
    }
}

// ---------------------------------------------------------------- IIC on the C x C joint (one workgroup): iic_device.h
__global__ __launch_bounds__(1024) void iic_core_rows_kernel(const float *P0, int C, float lamb, float eps, float *grad, double *part)
{
    extern __shared__ float iic_L[];
    iic_core_rows_multi(P0, C, lamb, eps, grad, part, iic_L);
}

__global__ __launch_bounds__(256) void iic_core_shift_kernel(float *P0, int C, float w_iic, float *out, const float *grad, const double *part, int n_part)
{
    iic_core_shift(P0, C, w_iic, out, grad, part, n_part);
}

// Round 6, the step of n_clusters > 48: what the middle backward needs of the IIC gradient is z dP0 for all rows, and dP0 = w_iic (g - gp) / s is
// the rows launch's unshifted gradient g moved by the global sum gp -- for a softmax row (its entries add up to 1)
// so the shift launch and the library GEMM behind it (5.1 + 5.0 us at 200 classes) are ONE launch of 16 x 16 MFMA tiles that takes g - gp as its B operand
// (rounded exactly as the shift launch rounds it; C <= 256: 64 K-steps).  LossFunctions.py:20-62.
__global__ __launch_bounds__(256) void iic_dz_kernel(const float *__restrict__ z, int m, int C, const float *__restrict__ grad, const double *__restrict__ part,
                                                     int n_part, float w_iic, float *__restrict__ dzs, float *__restrict__ out)
{
    // a workgroup: 16 rows of z (staged in LDS by coalesced loads, once) x 4 column tiles, a wave each; every B operand of a tile (a column of g: 64-byte
    // segments) requested before the first product.  g enters as g - gp, rounded as iic_core_shift rounds it.
    __shared__ float zs[16][260];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, l = lane & 15, q = lane >> 4;
    const int ct = (C + 15) / 16, groups = (ct + 3) / 4;
    const int r0 = ((int)blockIdx.x / groups) * 16, c0 = (((int)blockIdx.x % groups) * 4 + wv) * 16;
    {
        float zv[16];                                        // thread t: column t of the block's 16 rows (clamped loads, all in flight)
        const int cc = tid < C ? tid : C - 1;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) zv[rr] = z[(int64_t)(r0 + rr < m ? r0 + rr : m - 1) * C + cc];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) if (tid < C) zs[rr][tid] = r0 + rr < m ? zv[rr] : 0.f;
    }
    __shared__ double ps[64];                                // the rows launch's partial sums (2 n_part + 1 <= 51 doubles): one trip, then added up from LDS
    if (tid < 2 * n_part + 1) ps[tid] = part[tid];
    const bool okc = c0 + l < C;
    const float *gb = grad + (okc ? c0 + l : 0);             // B[k = q][j = l] = g[k0 + q][c0 + l] - gp
    float bv[64];
#pragma unroll
    for (int u = 0; u < 64; ++u) {                           // (clamped, not predicated: a load under a branch is waited for before the next is issued)
        const int k = 4 * u + q;
        bv[u] = gb[(int64_t)(k < C ? k : C - 1) * C];
    }
    __syncthreads();
    double l1 = 0.0, g1 = 0.0;                               // the gradient's global sum and the joint's total, in iic_core_shift's order
    for (int i = 0; i < n_part; ++i) { l1 += ps[2 * i]; g1 += ps[2 * i + 1]; }
    const float gp = (float)g1, s = (float)ps[2 * n_part];
    if (blockIdx.x == 0 && tid == 0) out[3] = (float)l1;
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 64; u += 2) {                        // A[i = l][k = q] = z[r0 + l][4 u + q] (zero beyond C: the products vanish)
        const int k0 = 4 * u + q, k1 = k0 + 4;
        const float a0 = k0 < C ? zs[l][k0] : 0.f, a1 = k1 < C ? zs[l][k1] : 0.f;
        const float b0 = (k0 < C && okc) ? bv[u] - gp : 0.f, b1 = (k1 < C && okc) ? bv[u + 1] - gp : 0.f;
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
    }
    const float sc = w_iic / s;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {                      // C/D: row = 4 q + reg, column = l
        const int r = r0 + 4 * q + reg;
        if (r < m && okc) dzs[(int64_t)r * C + c0 + l] = sc * (acc0[reg] + acc1[reg]);
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void iic_core_kernel(float *P0, int C, float lamb, float eps, float w_iic, float *scratch,
                                                      float *out)
{
    if (NT == 256) iic_core_small(P0, C, lamb, eps, w_iic, out);     // C <= 48 (launcher)
    else iic_core_rows(P0, C, lamb, eps, w_iic, out);                // 48 < C <= 256
}

// ---------------------------------------------------------------- head backward: one wave per row
// inputs: z, r2, f, inv, G = (E + E^T) f, dP0 (scaled by w_iic), W3;  outputs: dlogits [m,C], dlat [m,64]
__global__ __launch_bounds__(256) void head_bwd_kernel(const float *z, const float *r2, const float *f, const float *inv,
                                                       const float *G, int g_parts, const float *dP0, const float *W3, int m, int C,
                                                       int train, float nce_coef, float *dlogits, float *dlat, const float *dzs)
{
    // dzs != NULL: dzs = z dP0 ([m, C], one GEMM by the caller -- at n_clusters = 200 the per-row product below is 40 000 FMAs
    // behind global reads of a 160 KB matrix); row r then takes its partner's row of it.
    __shared__ float shz[4][256];
    __shared__ float shd[4][256];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + w;
    if (row >= m) return;
    const int B = m / 2;
    const bool first = row < B;
    const int prow = first ? row + B : row - B;
    if (dzs == nullptr) for (int c = lane; c < C; c += 64) shz[w][c] = z[(int64_t)prow * C + c];
    __builtin_amdgcn_wave_barrier();
    // dz[c] = sum_c' zp[c'] dP0[c',c] for both views: dP0 is symmetric (see iic_core_kernel), so view 1's
    // dP0[c,c'] is read as dP0[c',c] -- coalesced across the lanes
    float zc[MAX_CPL], dz[MAX_CPL];
    float dot = 0.f;
#pragma unroll
    for (int t = 0; t < MAX_CPL; ++t) {
        const int c = t * 64 + lane;
        zc[t] = 0.f; dz[t] = 0.f;
        if (c < C) {
            zc[t] = z[(int64_t)row * C + c];
            float acc = 0.f;
            if (dzs != nullptr) acc = dzs[(int64_t)prow * C + c];
            else {
#pragma unroll 4
                for (int k = 0; k < C; ++k) acc = fmaf(shz[w][k], dP0[k * C + c], acc);
            }
            dz[t] = acc;
            dot += acc * zc[t];
        }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int t = 0; t < MAX_CPL; ++t) {
        const int c = t * 64 + lane;
        if (c < C) {
            const float dl = zc[t] * (dz[t] - dot);       // softmax backward
            dlogits[(int64_t)row * C + c] = dl;
            shd[w][c] = dl;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // classifier branch: dr2[h] = sum_c dlogits[c] W3[c,h]; through Dropout+ReLU (kept & active <=> r2 > 0)
    float dr = 0.f;
    for (int c = 0; c < C; ++c) dr = fmaf(shd[w][c], W3[(int64_t)c * H2 + lane], dr);
    const float a = r2[(int64_t)row * H2 + lane];
    float dl_cls = a > 0.f ? dr * (train ? 2.f : 1.f) : 0.f;
    // contrastive branch: df = nce_coef * (G_r - 2 f_pos); through f = lat / ||lat||
    const float fr = f[(int64_t)row * H2 + lane];
    float gsum = G[(int64_t)row * H2 + lane];
    for (int p = 1; p < g_parts; ++p) gsum += G[((int64_t)p * m + row) * H2 + lane];      // partial slabs of idl_nce_fused
    const float df = nce_coef * (gsum - 2.f * f[(int64_t)prow * H2 + lane]);
    const float proj = wave_sum(fr * df);
    dlat[(int64_t)row * H2 + lane] = dl_cls + (df - fr * proj) * inv[row];
}

// ---------------------------------------------------------------- column sums (bias gradients)
// partial[p, c] = sum over the rows of chunk p of x[r, c]  (p = blockIdx.y, COL_PARTS chunks); the chunks
// are added up, in order, by rmsprop_kernel -- deterministic and without an extra launch.

struct ColJob { float *x; const float *act; float *partial; int n; float scale; };   // act != NULL: ReLU/Dropout backward in place
struct ColJobs {
    ColJob j[3]; int first_block[4]; int m; int64_t *ctl; int64_t batch_advance;   // ctl != NULL: ctl[1] += batch_advance
    // optional 4th job (blockIdx.x == first_block[3]): partial weight gradient of the last layer, dW3_part[p][c][h] =
    // sum over row chunk p of dlogits[r][c] * r2[r][h]  (C x 64 outputs, reduced over the chunks by rmsprop_kernel)
    const float *dlogits; const float *r2; float *dW3_part; int C;
};

__global__ __launch_bounds__(256) void col_partial_kernel(ColJobs jobs)
{
    __shared__ float sh[4][64];
    if (jobs.ctl != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) jobs.ctl[1] += jobs.batch_advance;
    if ((int)blockIdx.x == jobs.first_block[3]) {              // dW3 partial of row chunk blockIdx.y
        __shared__ float dl[32][48], rr[32][H2];
        const int m = jobs.m, C = jobs.C;
        const int rows = (m + COL_PARTS - 1) / COL_PARTS;
        const int r0 = blockIdx.y * rows;
        const int r1 = (r0 + rows < m) ? r0 + rows : m;
        float acc[12];                                          // ceil(48 * 64 / 256) outputs per thread
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] = 0.f;
        for (int t0 = r0; t0 < r1; t0 += 32) {
            const int nr = (r1 - t0 < 32) ? r1 - t0 : 32;
            __syncthreads();
            for (int i = threadIdx.x; i < nr * C; i += 256) { const int r = i / C, c = i - r * C; dl[r][c] = jobs.dlogits[(int64_t)(t0 + r) * C + c]; }
            for (int i = threadIdx.x; i < nr * H2; i += 256) rr[i >> 6][i & 63] = jobs.r2[(int64_t)t0 * H2 + i];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int o = threadIdx.x + 256 * i;            // o = c * 64 + h
                if (o < C * H2) {
                    const int c = o >> 6, h = o & 63;
                    float a = acc[i];
                    for (int r = 0; r < nr; ++r) a = fmaf(dl[r][c], rr[r][h], a);
                    acc[i] = a;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int o = threadIdx.x + 256 * i;
            if (o < C * H2) jobs.dW3_part[(int64_t)blockIdx.y * C * H2 + o] = acc[i];
        }
        return;
    }
    int k = 0;
    while (k < 2 && (int)blockIdx.x >= jobs.first_block[k + 1]) ++k;
    const ColJob job = jobs.j[k];
    const int m = jobs.m, n = job.n;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = ((int)blockIdx.x - jobs.first_block[k]) * 64 + lane;
    const int rows = (m + COL_PARTS - 1) / COL_PARTS;
    const int r0 = blockIdx.y * rows;
    int r1 = r0 + rows;
    if (r1 > m) r1 = m;
    float acc = 0.f;
    if (c < n) {
        for (int r = r0 + w; r < r1; r += 4) {
            float v = job.x[(int64_t)r * n + c];
            if (job.act != nullptr) {   // x = d(output of Dropout) -> d(pre-activation), in place
                v = job.act[(int64_t)r * n + c] > 0.f ? v * job.scale : 0.f;
                job.x[(int64_t)r * n + c] = v;
            }
            acc += v;
        }
    }
    sh[w][lane] = acc;
    __syncthreads();
    if (w == 0 && c < n) job.partial[(int64_t)blockIdx.y * n + c] = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
}

// ---------------------------------------------------------------- fused middle of the backward pass (fp32 MFMA)
// One 1024-thread workgroup per row chunk p of the partial sums (16 rows at m = 1024), replacing head_bwd + the hipBLASLt GEMM
// dr1 = dlat W2 + bias_grads:
//   1. head backward, one row per wave (as head_bwd_kernel): dlogits, dlat -> global and LDS
//   2. dr1 = (dlat W2) masked by the layer-1 ReLU/Dropout (act1 > 0, x2 in training): v_mfma_f32_16x16x4_f32, K = 64, each wave
//      two of the 32 column tiles (its W2 fragments are loaded once, before phase 1); column sums -> partial1[p]
//   3. partial2[p] = column sums of dlat, partial3[p] = column sums of dlogits, dW3_part[p] = dlogits^T r2 over the chunk's rows
struct MidBwdArgs {
    const float *z, *r2, *f, *inv, *G, *dP0, *W3, *W2, *act1;
    float *dlogits, *dlat, *dr1, *partial1, *partial2, *partial3, *dW3_part;
    int64_t *ctl; int64_t batch_advance;
    int g_parts, m, C, train; float nce_coef;
    int act1_t;               // act1 is stored transposed, [512, m]
    // dr1 as two fp16 planes for the dW1 tiles' LDS-DMA (wgrad_planes_device.h: dplanes_body), in place of the fp32 tensor: the planes, the words of
    // their scale (planes.h: DR1_WORDS), the overflow flag
    uint16_t *dr1h, *dr1l;
    int *dscale, *dover;
    const float *dzs;         // BIG (n_clusters > 48): z dP0 for all rows ([m, C], one GEMM by the caller) -- row r takes its partner's row; NULL: the product per row, in the kernel
};

// (tile1 > tile0: the workgroups behind the COL_PARTS computing ones assemble tiles [tile0, tile1) of the NEXT batch into a
// second x buffer, four 256-thread gather tiles of MID_GATHER_ROWS rows each -- this launch leaves three quarters of the CUs idle and is latency-bound, the gather is
// pure streaming, and nothing in this step reads or writes what it touches.)
// (Round 3's variant with InfoNCE pass 2 and the IIC core inside this launch -- 32.8 us against 9.5 + 13.5: the similarity tiles on this launch's 64 CUs -- was
// removed in round 6: DESIGN, History.)
template <bool BIG = false, bool DP = false>      // BIG: n_clusters > 48 (the per-row products then walk dP0 / W3 in memory); a separate instance, so that its
                                          // registers do not count against the n_clusters <= 48 one the training step runs.  DP: dr1 goes out as two fp16 planes
__device__ __forceinline__ void mid_bwd_body(const MidBwdArgs &a, int tile0, int tile1, const idl_dev::GatherArgs &gth, const int bid)
{
    extern __shared__ float mid_dyn[];        // BIG: W3 [C][64]
    if (bid >= COL_PARTS) {
        const int blk = tile0 + (bid - COL_PARTS) * 4 + (int)(threadIdx.x >> 8);
        if (blk < tile1) idl_dev::gather_tile<MID_GATHER_ROWS>(gth, (int64_t)blk, (int)(threadIdx.x & 255));
        return;
    }
    __shared__ float DL[16][H2 + 4];      // dlat rows of the tile; +4: the 16 rows of an A-fragment read hit disjoint banks
    __shared__ float R2s[16][H2];
    __shared__ float DLG[16][64 * MAX_CPL];
    __shared__ float shz[MID_WAVES][64 * MAX_CPL];
    __shared__ float sP[48 * 48], sW3[48 * H2];      // dP0 and W3 for n_clusters <= 48: read once per workgroup, not once per row
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l = lane & 15, q = lane >> 4, tid = threadIdx.x;
    const int m = a.m, C = a.C, B = m / 2;
    constexpr bool small = !BIG;
    const int rows = (m + COL_PARTS - 1) / COL_PARTS;
    const int r0 = bid * rows;
    const int r1 = (r0 + rows < m) ? r0 + rows : m;
    uint64_t *stb = IDL_PHASE_BUF(2, bid);
    IDL_PHASE_STAMP(stb, 0);
    if (a.ctl != nullptr && bid == 0 && tid == 0) a.ctl[1] += a.batch_advance;
    __shared__ unsigned int s_dmx, s_dcnt;
    // the exponent of this launch's planes: what the previous step's dW1 launch derived from that step's largest |dr1| (one scalar load, first used
    // behind the head backward)
    const int dk = DP ? a.dscale[1] : 0;
    if (DP && tid == 0) { s_dmx = 0u; s_dcnt = 0u; }
    unsigned int dmx = 0u;                       // the largest |dr1| this thread wrote, as its bit pattern (NaN and Inf order above every finite value)
    if (small) {
        for (int i = tid; i < C * C; i += 64 * MID_WAVES) sP[i] = a.dP0[i];
        for (int i = tid; i < C * H2; i += 64 * MID_WAVES) sW3[i] = a.W3[i];
    } else {
        for (int i = tid; i < C * H2; i += 64 * MID_WAVES) mid_dyn[i] = a.W3[i];      // BIG: W3 (51 KB at 200 classes) in dynamic LDS -- read once per workgroup, not once per row
    }
    // B fragments of this wave's two column tiles.  The wave owns the 32 columns [32 wv, 32 wv + 32) of dr1 and its two tiles take
    // them INTERLEAVED -- tile j, lane l = column 32 wv + 2 l + j -- so that a lane's B values of both tiles are one 8-byte read
    // (B[k = 16 q + s][c] = W2[k][c]; 16 lanes x 8 B = one full 128-byte line per q) and the dr1 stores below are 8-byte stores
    float bw[2][16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float2 w2 = *(const float2 *)(a.W2 + (int64_t)(16 * q + s) * H1 + 32 * wv + 2 * l);
        bw[0][s] = w2.x; bw[1][s] = w2.y;
    }
    const float scale = a.train ? 2.f : 1.f;
    float cs1[2] = {0.f, 0.f}, s23 = 0.f, acc3[3] = {0.f, 0.f, 0.f};
    for (int t0 = r0; t0 < r1; t0 += 16) {
        const int nr = (r1 - t0 < 16) ? r1 - t0 : 16;
        // the layer-1 activations this lane masks with in phase 2: requested now, needed after the head backward
        float a1v[2][4];
        if (a.act1_t && nr == 16 && (m & 3) == 0) {          // transposed image: this lane's four rows of a column are one 16-byte read
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float4 t4 = *(const float4 *)(a.act1 + (int64_t)(32 * wv + 2 * l + j) * m + t0 + 4 * q);
                a1v[j][0] = t4.x; a1v[j][1] = t4.y; a1v[j][2] = t4.z; a1v[j][3] = t4.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int rl = 4 * q + reg;
                    a1v[j][reg] = rl >= nr ? 0.f : a.act1_t ? a.act1[(int64_t)(32 * wv + 2 * l + j) * m + t0 + rl]
                                                            : a.act1[(int64_t)(t0 + rl) * H1 + 32 * wv + 2 * l + j];
                }
        }
        IDL_PHASE_STAMP(stb, 1);
        __syncthreads();
        IDL_PHASE_STAMP(stb, 2);                 // (every wave's prologue requests have arrived: sP, sW3 staged)
        // ---- 1. head backward of row t0 + wv
        if constexpr (small) { if (wv < nr) {
            // every global read of the row first, then arithmetic on registers and LDS only
            const int row = t0 + wv;
            const int prow = row < B ? row + B : row - B;
            // (clamped, not predicated: the compiler puts a load under a condition into a branch of its own and drains every load in flight in
            //  front of it -- two extra trips to memory at the head of this launch's longest chain)
            const bool cl = lane < C;
            const int lc = cl ? lane : 0;
            const float zp_ = a.z[(int64_t)prow * C + lc], zc_ = a.z[(int64_t)row * C + lc];
            const float act = a.r2[(int64_t)row * H2 + lane];
            const float fr = a.f[(int64_t)row * H2 + lane], fp = a.f[(int64_t)prow * H2 + lane];
            float gp[8];                       // the first 8 partial products (idl_nce_fused_parts() = 8) are requested up front
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) gp[pp] = a.G[((int64_t)(pp < a.g_parts ? pp : 0) * m + row) * H2 + lane];
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) gp[pp] = pp < a.g_parts ? gp[pp] : 0.f;
            const float zp = cl ? zp_ : 0.f, zc = cl ? zc_ : 0.f;
            const float invr = a.inv[row];
            shz[wv][lane] = zp;
            __builtin_amdgcn_wave_barrier();
            float dz = 0.f;
            if (cl) {
#pragma unroll 2                              // (4: four row pointers at a stride known only at run time; the fourth lived in scratch)
                for (int k = 0; k < C; ++k) dz = fmaf(shz[wv][k], sP[k * C + lane], dz);
            }
            const float dot = wave_sum(dz * zc);
            const float dl = zc * (dz - dot);                     // softmax backward (0 for lanes >= C)
            if (cl) a.dlogits[(int64_t)row * C + lane] = dl;
            DLG[wv][lane] = dl;
            __builtin_amdgcn_wave_barrier();
            float dr = 0.f;
#pragma unroll 4
            for (int c = 0; c < C; ++c) dr = fmaf(DLG[wv][c], sW3[c * H2 + lane], dr);
            const float dl_cls = act > 0.f ? dr * scale : 0.f;
            float gsum = gp[0];
#pragma unroll
            for (int pp = 1; pp < 8; ++pp) if (pp < a.g_parts) gsum += gp[pp];
            for (int pp = 8; pp < a.g_parts; ++pp) gsum += a.G[((int64_t)pp * m + row) * H2 + lane];
            const float df = a.nce_coef * (gsum - 2.f * fp);
            const float proj = wave_sum(fr * df);
            const float d = dl_cls + (df - fr * proj) * invr;
            a.dlat[(int64_t)row * H2 + lane] = d;
            DL[wv][lane] = d;
            R2s[wv][lane] = act;
        } } else if (wv < nr) {
            const int row = t0 + wv;
            const int prow = row < B ? row + B : row - B;
            for (int c = lane; c < C; c += 64) shz[wv][c] = a.z[(int64_t)prow * C + c];
            __builtin_amdgcn_wave_barrier();
            float zc[MAX_CPL], dz[MAX_CPL];
            float dot = 0.f;
#pragma unroll
            for (int t = 0; t < MAX_CPL; ++t) {
                const int c = t * 64 + lane;
                zc[t] = 0.f; dz[t] = 0.f;
                if (c < C) {
                    zc[t] = a.z[(int64_t)row * C + c];
                    float acc = 0.f;
                    if (a.dzs != nullptr) acc = a.dzs[(int64_t)prow * C + c];
                    else {
#pragma unroll 4
                        for (int k = 0; k < C; ++k) acc = fmaf(shz[wv][k], a.dP0[k * C + c], acc);
                    }
                    dz[t] = acc;
                    dot += acc * zc[t];
                }
            }
            dot = wave_sum(dot);
#pragma unroll
            for (int t = 0; t < MAX_CPL; ++t) {
                const int c = t * 64 + lane;
                if (c < C) {
                    const float dl = zc[t] * (dz[t] - dot);       // softmax backward
                    a.dlogits[(int64_t)row * C + c] = dl;
                    DLG[wv][c] = dl;
                }
            }
            __builtin_amdgcn_wave_barrier();
            float dr = 0.f;
#pragma unroll 4
            for (int c = 0; c < C; ++c) dr = fmaf(DLG[wv][c], mid_dyn[c * H2 + lane], dr);
            const float act = a.r2[(int64_t)row * H2 + lane];
            const float dl_cls = act > 0.f ? dr * scale : 0.f;
            const float fr = a.f[(int64_t)row * H2 + lane];
            float gsum = a.G[(int64_t)row * H2 + lane];
            for (int pp = 1; pp < a.g_parts; ++pp) gsum += a.G[((int64_t)pp * m + row) * H2 + lane];
            const float df = a.nce_coef * (gsum - 2.f * a.f[(int64_t)prow * H2 + lane]);
            const float proj = wave_sum(fr * df);
            const float d = dl_cls + (df - fr * proj) * a.inv[row];
            a.dlat[(int64_t)row * H2 + lane] = d;
            DL[wv][lane] = d;
            R2s[wv][lane] = act;
        }
        if (wv >= nr) {
            DL[wv][lane] = 0.f;
            R2s[wv][lane] = 0.f;
            for (int c = lane; c < C; c += 64) DLG[wv][c] = 0.f;
        }
        IDL_PHASE_STAMP(stb, 3);
        __syncthreads();
        IDL_PHASE_STAMP(stb, 4);
        // ---- 2. dr1 tile = DL[16 x 64] W2[64 x 512]
        float av[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 t = *(const float4 *)&DL[l][16 * q + 4 * i];
            av[4 * i] = t.x; av[4 * i + 1] = t.y; av[4 * i + 2] = t.z; av[4 * i + 3] = t.w;
        }
        {
            f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bw[0][s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bw[1][s], acc1, 0, 0, 0);
            }
            const int col = 32 * wv + 2 * l;                       // this lane's two adjacent columns (tile 0, tile 1)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int rl = 4 * q + reg;                        // C/D: row = 4q + reg, col = l
                if (rl < nr) {
                    const float v0 = a1v[0][reg] > 0.f ? acc0[reg] * scale : 0.f;
                    const float v1 = a1v[1][reg] > 0.f ? acc1[reg] * scale : 0.f;
                    if constexpr (DP) {                        // two fp16 planes of the scaled, clamped pair (planes.h), 4 bytes of each
                        const float dsc = __builtin_ldexpf(1.f, dk);
                        const float s0 = __builtin_amdgcn_fmed3f(v0 * dsc, -idl_planes::LIMIT, idl_planes::LIMIT);
                        const float s1 = __builtin_amdgcn_fmed3f(v1 * dsc, -idl_planes::LIMIT, idl_planes::LIMIT);
                        uint32_t h01, l01;
                        float q0, q1;
                        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h01) : "v"(s0), "v"(s1));
                        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(q0) : "v"(h01), "v"(s0));
                        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(q1) : "v"(h01), "v"(s1));
                        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l01) : "v"(q0), "v"(q1));
                        *(uint32_t *)(a.dr1h + (int64_t)(t0 + rl) * H1 + col) = h01;
                        *(uint32_t *)(a.dr1l + (int64_t)(t0 + rl) * H1 + col) = l01;
                        const unsigned int b0 = __float_as_uint(v0) & 0x7FFFFFFFu, b1 = __float_as_uint(v1) & 0x7FFFFFFFu;
                        dmx = dmx > b0 ? dmx : b0; dmx = dmx > b1 ? dmx : b1;
                    } else
                    *(float2 *)(a.dr1 + (int64_t)(t0 + rl) * H1 + col) = make_float2(v0, v1);
                    cs1[0] += v0;
                    cs1[1] += v1;
                }
            }
        }
        IDL_PHASE_STAMP(stb, 5);
        // ---- 3. the small partial sums, from LDS (rows >= nr are zero)
        if (tid < H2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s23 += DL[r][tid];
        } else if (tid < H2 + C) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s23 += DLG[r][tid - H2];
        }
        if (a.dW3_part != nullptr) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int o = tid + 1024 * i;                      // o = c * 64 + h, C <= 48
                if (o < C * H2) {
                    const int c = o >> 6, h = o & 63;
                    float x = acc3[i];
#pragma unroll
                    for (int r = 0; r < 16; ++r) x = fmaf(DLG[r][c], R2s[r][h], x);
                    acc3[i] = x;
                }
            }
        }
    }
    // (the epilogue's indices are formed HERE, from an opaque copy of the thread number: formed in front of the row loop they were
    //  kept alive across it in scratch -- 16 bytes per lane at the 128-register cap of a 1 024-thread workgroup)
    int te = tid;
    asm volatile("" : "+v"(te));
    const int le = te & 15, qe = (te >> 4) & 3, we = te >> 6;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float v = cs1[j];
        v = idl_dev::add_xor32(idl_dev::add_xor16(v));
        if (qe == 0) a.partial1[(int64_t)bid * H1 + 32 * we + 2 * le + j] = v;
    }
    if constexpr (DP) {                          // the workgroup's largest |dr1| -> its word of the scale's words; beyond the planes' range (or NaN): the flag
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned int y = (unsigned int)__shfl_xor((int)dmx, o, 64); dmx = dmx > y ? dmx : y; }
        if ((te & 63) == 0) {
            atomicMax(&s_dmx, dmx);
            if (atomicAdd(&s_dcnt, 1u) == (unsigned int)(MID_WAVES - 1)) {
                const unsigned int wmx = atomicMax(&s_dmx, 0u);
                a.dscale[4 + bid] = (int)wmx;
                if (bid == 0) a.dscale[0] = dk;
                if (!(__uint_as_float(wmx) * __builtin_ldexpf(1.f, dk) <= idl_planes::LIMIT)) *a.dover = 1;
            }
        }
    }
    if (te < H2) a.partial2[(int64_t)bid * H2 + te] = s23;
    else if (te < H2 + C) a.partial3[(int64_t)bid * C + te - H2] = s23;
    if (a.dW3_part != nullptr) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int o = te + 1024 * i;
            if (o < C * H2) a.dW3_part[(int64_t)bid * C * H2 + o] = acc3[i];
        }
    }
    IDL_PHASE_STAMP(stb, 6);
}
struct MidBwdParams { MidBwdArgs a; int tile0, tile1; idl_dev::GatherArgs gth; };
static_assert(sizeof(MidBwdParams) + idl::PLAN_PARAMS <= idl::PLAN_BYTES, "MidBwdParams does not fit a plan record");

template <bool BIG = false, bool DP = false>
__global__ __launch_bounds__(64 * MID_WAVES) void mid_bwd_kernel(MidBwdArgs a, int tile0, int tile1, idl_dev::GatherArgs gth)
{
    mid_bwd_body<BIG, DP>(a, tile0, tile1, gth, (int)blockIdx.x);
}

// several voters in one launch, grid (voters, workgroups of one voter) as in mid_fwd_batched_kernel (n_clusters <= 48, separate
// InfoNCE passes)
__global__ __launch_bounds__(64 * MID_WAVES) void mid_bwd_batched_kernel(const unsigned char *__restrict__ plans)
{
    const MidBwdParams &p = *(const MidBwdParams *)(plans + (size_t)blockIdx.x * idl::PLAN_BYTES + idl::PLAN_PARAMS);
    if (p.a.dr1h != nullptr) mid_bwd_body<false, true>(p.a, p.tile0, p.tile1, p.gth, (int)blockIdx.y);
    else mid_bwd_body<false, false>(p.a, p.tile0, p.tile1, p.gth, (int)blockIdx.y);
}

// ---------------------------------------------------------------- RMSprop over all parameter tensors
struct RmsArgs {
    const float *loss_rows;   // optional step-loss assembly (models.py:128): see idl_rmsprop_step
    float *out;
    int loss_m;
    float w_nce, w_iic;
    float *p[8];
    const float *g[8];
    float *v[8];
    int64_t n[8];
    int parts[8];     // g[t] holds parts[t] stacked partial gradients [parts, n] (1 = a plain gradient)
    int count;
    // optional: the gradient of tensor wg_t is not read but computed here as wg_dy^T wg_x ([wg_m, n_out]^T [wg_m, n_in]) by
    // wg_tiles extra workgroups, one 16 x 16 tile each, which also apply the update to their tile (a.n[wg_t] is 0 then)
    const float *wg_dy, *wg_x; float *wg_grad;
    int wg_t, wg_m, wg_n_out, wg_n_in, wg_tiles, wg_xt;      // wg_xt: wg_x is stored transposed, [n_in, wg_m]
    int first[9];     // optimizer blocks [first[t], first[t + 1]) belong to tensor t: as many as the tensor needs, not one grid row each
};

// One 16 x 16 tile of dy^T x on the fp32 matrix cores (v_mfma_f32_16x16x4_f32): the four waves split the contraction index m,
// the partial tiles are added through LDS in a fixed order, then RMSprop on the tile.  red: 1024 floats of LDS.
// AHEAD (the launch with the dW1 tiles at its head, where these workgroups run behind the tiles and every microsecond of their chain
// is the launch's tail): the parameter / square_avg elements and BOTH 128-row batches of a wave's 256 rows are requested before
// the first product -- one trip to memory instead of three (128 more registers, which that launch has and rmsprop_kernel, whose
// streaming blocks want 16 waves per CU, has not).
// SPIN: the four waves meet on an LDS counter instead of s_barrier (the loader waves of wgrad_xplanes_rms_kernel, whose workgroup's other
// four waves are elsewhere: an s_barrier would wait for them); *spin is zero when the workgroup starts and is used once.
template <bool SPIN>
__device__ __forceinline__ void tail_sync(const int tix, unsigned int *spin)
{
    if constexpr (!SPIN) { __syncthreads(); }
    else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (this wave's partial tile is in LDS)
        if ((tix & 63) == 0) atomicAdd(spin, 1u);
        while (*(volatile unsigned int *)spin < 4u) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    }
}

template <bool AHEAD, bool SPIN = false>
__device__ __forceinline__ void wgrad_tile_rms(const RmsArgs &a, int tile, float lr, float alpha, float eps, float wd, float oma, float *red, const int tix,
                                               unsigned int *spin)
{
    const int tid = tix, lane = tid & 63, wv = tid >> 6, l = lane & 15, q = lane >> 4;
    const int tn = a.wg_n_in / 16;
    const int i0 = (tile / tn) * 16, j0 = (tile % tn) * 16;
    const int lda = a.wg_n_out, ldb = a.wg_n_in, m = a.wg_m;
    const int per = ((m + 15) / 16) * 4;                     // rows per wave, a multiple of 4
    const int kb = wv * per, ke = (kb + per < m) ? kb + per : m;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    const float *pa = a.wg_dy + i0 + l, *pb = a.wg_x + j0 + l;    // A[i = l][k = q] = dy[k][i0 + l], B[k = q][j = l] = x[k][j0 + l]
    const int oo = (i0 + (tid >> 4)) * ldb + j0 + (tid & 15);     // this thread's element of the tile in the epilogue
    float pi = 0.f, vi = 0.f;
    if (AHEAD) { pi = a.p[a.wg_t][oo]; vi = a.v[a.wg_t][oo]; }
    if (AHEAD && a.wg_xt && per == 256 && kb + per <= m) {
        const float *pbt = a.wg_x + (int64_t)(j0 + l) * m;
        float av[64], bv[64];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int u4 = 0; u4 < 8; ++u4) {
                const float4 t4 = *(const float4 *)(pbt + kb + 128 * h + 32 * q + 4 * u4);
                bv[32 * h + 4 * u4] = t4.x; bv[32 * h + 4 * u4 + 1] = t4.y; bv[32 * h + 4 * u4 + 2] = t4.z; bv[32 * h + 4 * u4 + 3] = t4.w;
            }
#pragma unroll
            for (int u = 0; u < 32; ++u) av[32 * h + u] = pa[(kb + 128 * h + 32 * q + u) * lda];
        }
#pragma unroll
        for (int u = 0; u < 64; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
    } else if (a.wg_xt && (per & 127) == 0 && kb + per <= m) {
        // x stored transposed ([n_in, m]): lane (l, q) walks 32 consecutive k of ITS column per batch (16-byte reads), and MFMA step
        // u takes k = k0 + 32 q + u on both operands -- any assignment of k to (step, q) is a valid order of the same sum
        const float *pbt = a.wg_x + (int64_t)(j0 + l) * m;
        for (int k0 = kb; k0 < ke; k0 += 128) {
            float av[32], bv[32];
#pragma unroll
            for (int u4 = 0; u4 < 8; ++u4) {
                const float4 t4 = *(const float4 *)(pbt + k0 + 32 * q + 4 * u4);
                bv[4 * u4] = t4.x; bv[4 * u4 + 1] = t4.y; bv[4 * u4 + 2] = t4.z; bv[4 * u4 + 3] = t4.w;
            }
#pragma unroll
            for (int u = 0; u < 32; ++u) av[u] = pa[(k0 + 32 * q + u) * lda];
#pragma unroll
            for (int u = 0; u < 32; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
        }
    } else if (a.wg_xt) {
        for (int k0 = kb; k0 < ke; k0 += 4) {
            const int k = k0 + q;
            const bool ok = k < ke;
            const float ta = ok ? pa[k * lda] : 0.f, tb = ok ? a.wg_x[(int64_t)(j0 + l) * m + k] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ta, tb, acc, 0, 0, 0);
        }
    } else
    for (int k0 = kb; k0 < ke; k0 += 128) {                  // 64 independent loads in flight per lane, then 32 MFMAs
        float av[32], bv[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int k = k0 + 4 * u + q;
            const bool ok = k < ke;
            const int kc = ok ? k : ke - 1;
            const float ta = pa[kc * lda], tb = pb[kc * ldb];
            av[u] = ok ? ta : 0.f;
            bv[u] = ok ? tb : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wv * 256 + (4 * q + r) * 16 + l] = acc[r];          // C/D: row = 4 q + reg, col = l
    tail_sync<SPIN>(tix, spin);
    const float g = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
    const int o = oo;
    if (a.wg_grad != nullptr) a.wg_grad[o] = g;
    float *p = a.p[a.wg_t], *v = a.v[a.wg_t];
    if (!AHEAD) { pi = p[o]; vi = v[o]; }
    wg_dev::rms_update(g, pi, vi, wg_dev::Hyper{lr, alpha, eps, wd, oma});
    v[o] = vi;
    p[o] = pi;
}

// Grid: when g.y != NULL the first n_gather = gather_blocks() blocks assemble the NEXT batch (nothing in this launch writes what they
// read: the batch offset ctl[1] was already advanced by the bias-gradient launch of this step, and x is no longer read by this
// step's GEMMs); they go first because a gathered row is a dependent chain of HBM latencies that the streaming optimizer blocks
// behind them hide.  The following gx * count blocks are optimizer blocks (tensor = block / gx).
constexpr int RMS_UNROLL = 4;         // 16-byte elements per thread of the streaming update

constexpr int RMS_THREADS = 256;   // threads of an optimizer block (the launch may carry a fifth, idle wave: wgrad_rmsprop_kernel)

template <bool AHEAD = false, bool SPIN = false>
__device__ __forceinline__ void rmsprop_body(const RmsArgs &a, const float *hyper, int64_t *ctl, int64_t batch_advance, int gx,
                                             int n_gather, const idl_dev::GatherArgs &g, const int blk, const int tix, unsigned int *spin = nullptr)
{
    // grid order: weight-gradient tiles (dependent chains of strided loads: first, so that the streaming blocks behind them hide
    // their latency), then the gather blocks, then the optimizer blocks
    const float lr = hyper[0], alpha = hyper[1], eps = hyper[2], wd = hyper[3], oma = hyper[4];
    if (blk < a.wg_tiles) {
        __shared__ float red[1024];
        wgrad_tile_rms<AHEAD, SPIN>(a, blk, lr, alpha, eps, wd, oma, red, tix, spin);
        return;
    }
    const int b0 = blk - a.wg_tiles;
    if (b0 < n_gather) {
        idl_dev::gather_block(g, (int64_t)b0, tix);
        return;
    }
    const int bid = b0 - n_gather;
    int t = 0;
    while (t + 1 < a.count && bid >= a.first[t + 1]) ++t;
    const int bx = bid - a.first[t];
    gx = a.first[t + 1] - a.first[t];
    if (t < a.count) {
        float *p = a.p[t]; const float *g = a.g[t]; float *v = a.v[t];
        const int64_t n = a.n[t];
        if (a.parts[t] == 1 && (n & 3) == 0 && ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)v)) & 15u) == 0) {
            float4 *p4 = (float4 *)p; const float4 *g4 = (const float4 *)g; float4 *v4 = (float4 *)v;
            // (one update formula for every tensor and every launch form -- wg_dev::rms_update, also the epilogue of the dW1 tiles: a
            //  voter's weights must not depend on which launch sequence trained it beyond the order of the sums)
            const wg_dev::Hyper hy{lr, alpha, eps, wd, oma};
            auto upd = [&](float4 &pi, float4 &vi, const float4 gr) {
                wg_dev::rms_update(gr.x, pi.x, vi.x, hy); wg_dev::rms_update(gr.y, pi.y, vi.y, hy);
                wg_dev::rms_update(gr.z, pi.z, vi.z, hy); wg_dev::rms_update(gr.w, pi.w, vi.w, hy);
            };
            // four 16-byte elements per thread, all twelve loads in flight before the first use.  (The weight-gradient tiles put this
            // kernel at 104 registers = 16 waves per CU: with two elements per thread the 1024 blocks of W1 did not all fit at once.)
            const int64_t n4 = n / 4, stride = (int64_t)gx * RMS_THREADS;
            for (int64_t i = (int64_t)bx * RMS_THREADS + tix; i < n4; i += RMS_UNROLL * stride) {
                float4 pv[RMS_UNROLL], vv[RMS_UNROLL], gv[RMS_UNROLL];
#pragma unroll
                for (int u = 0; u < RMS_UNROLL; ++u) {
                    const int64_t iu = i + u * stride < n4 ? i + u * stride : i;        // clamped, not predicated: the loads stay unconditional
                    pv[u] = p4[iu]; vv[u] = v4[iu]; gv[u] = g4[iu];
                }
#pragma unroll
                for (int u = 0; u < RMS_UNROLL; ++u) {
                    if (i + u * stride < n4) { upd(pv[u], vv[u], gv[u]); v4[i + u * stride] = vv[u]; p4[i + u * stride] = pv[u]; }
                }
            }
        } else
        for (int64_t i = (int64_t)bx * RMS_THREADS + tix; i < n; i += (int64_t)gx * RMS_THREADS) {
            const float pi = p[i];
            float gr;
            if (a.parts[t] == COL_PARTS) {          // 32 independent loads in flight, summed in a fixed order
                float part[COL_PARTS];
#pragma unroll
                for (int q = 0; q < COL_PARTS; ++q) part[q] = g[(int64_t)q * n + i];
                gr = part[0];
#pragma unroll
                for (int q = 1; q < COL_PARTS; ++q) gr += part[q];
            } else {
                gr = g[i];
                for (int q = 1; q < a.parts[t]; ++q) gr += g[(int64_t)q * n + i];
            }
            float pn = pi, vi = v[i];
            wg_dev::rms_update(gr, pn, vi, wg_dev::Hyper{lr, alpha, eps, wd, oma});
            v[i] = vi;
            p[i] = pn;
        }
    }
    if (bid == 0 && tix == 0) { ctl[0] += 1; ctl[1] += batch_advance; }
    if (a.out != nullptr && bid == a.first[a.count] - 1 && tix < 64) {
        // step loss = w_nce * mean(loss_rows) + w_iic * IIC (left in out[3] by iic_core_kernel); out[1] = running sum
        const float iic = a.out[3], run = a.out[1];           // requested together with the rows: one trip
        float part[16];
        float acc = 0.f;
        if (a.loss_m <= 1024) {                               // every load in flight before the first add (same order of additions)
#pragma unroll
            for (int j = 0; j < 16; ++j) { const int i = tix + 64 * j; part[j] = i < a.loss_m ? a.loss_rows[i] : 0.f; }
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += part[j];
        } else {
            for (int i = tix; i < a.loss_m; i += 64) acc += a.loss_rows[i];
        }
        acc = wave_sum(acc) / (float)a.loss_m;
        if (tix == 0) { const float l = a.w_nce * acc + a.w_iic * iic; a.out[0] = l; a.out[1] = run + l; a.out[2] = acc; }
    }
}

struct RmsParams { RmsArgs a; const float *hyper; int64_t *ctl; int64_t batch_advance; int n_gather; idl_dev::GatherArgs g; };
static_assert(sizeof(RmsParams) + idl::PLAN_PARAMS <= idl::PLAN_BYTES, "RmsParams does not fit a plan record");

__global__ __launch_bounds__(256) void rmsprop_kernel(RmsArgs a, const float *hyper, int64_t *ctl, int64_t batch_advance, int gx,
                                                      int n_gather, idl_dev::GatherArgs g)
{
    rmsprop_body(a, hyper, ctl, batch_advance, gx, n_gather, g, (int)blockIdx.x, (int)threadIdx.x);
}

// The optimizer launch with the LARGE weight gradient at its head (wgrad_device.h): the first w.tiles workgroups each accumulate a
// 64 x 128 tile of dW1 = dr1^T x on the matrix cores and apply RMSprop to it from their registers; the workgroups behind them are
// the optimizer launch as before (the small tensors, the dW2 tiles, step loss, step counter) and run beside them -- a tile
// workgroup is one wave per SIMD, so both fit a CU.  One launch, one boundary and the 8 MB gradient's round trip fewer than
// hipBLASLt's GEMM followed by rmsprop_kernel.
template <bool STAMPS = false, bool SKIP_TILES = false>      // STAMPS (IDELUCS_DEV=stamps=1, a diagnostic): every workgroup leaves {start, end, hardware id, XCC id | shader cycles << 8} in `stamps`
__global__ __launch_bounds__(wg_dev::THREADS) void wgrad_rmsprop_kernel(wg_dev::WgArgs w, RmsArgs a, const float *hyper, int64_t *ctl,
                                                            int64_t batch_advance, int n_gather, idl_dev::GatherArgs g, uint64_t *stamps)
{
    uint64_t t0 = 0, c0 = 0;
    if (STAMPS) { t0 = __builtin_amdgcn_s_memrealtime(); c0 = __builtin_amdgcn_s_memtime(); }
    extern __shared__ wg_dev::f32x4_t wg_img[];
    // The launch has the tiles' 320 threads; behind the tiles the fifth wave has nothing to do and LEAVES HERE, before any barrier,
    // in every instantiation: the barriers of rmsprop_body (written for 256 threads) and of the stamp epilogue below then pair up
    // among the same four waves at the same program points (an ended wave no longer counts towards s_barrier) -- ADVICE r3:
    // before, the STAMPS build had the fifth wave meet the epilogue's barrier while the others stood at rmsprop_body's.
    if ((int)blockIdx.x >= w.tiles && threadIdx.x >= 256) return;
    if ((int)blockIdx.x < w.tiles) { if (!SKIP_TILES) wg_dev::q16_tile<0>(w, (int)blockIdx.x, wg_img); }
    else {
        // a tile wave issues matrix instructions back to back and, being the older wave of its SIMD, wins every arbitration: at equal
        // priority these workgroups crawl beside it (measured with the stamps: 40 us for 5 us of work, and the tiles 3..16 us longer
        // wherever they met a dW2 tile); ahead of it they are gone after a few microseconds
        __builtin_amdgcn_s_setprio(3);
        rmsprop_body<true>(a, hyper, ctl, batch_advance, 0, n_gather, g, (int)blockIdx.x - w.tiles, (int)threadIdx.x);
    }
    if (STAMPS) {
        __syncthreads();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0 && blockIdx.x < 1024) {
            uint64_t *d = stamps + 4 * (size_t)blockIdx.x;
            d[0] = t0; d[1] = __builtin_amdgcn_s_memrealtime();
            d[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);          // HW_REG_HW_ID
            // XCC id in the low byte; above it the workgroup's life in SHADER cycles (s_memtime): with d[1] - d[0] (100 MHz ticks)
            // the clock the chip held while this workgroup ran (tools/mfma_clock.py)
            d[3] = (uint64_t)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xffu) | ((__builtin_amdgcn_s_memtime() - c0) << 8);
        }
    }
}

// The layer-1 forward tiles of step t + 1 (l1_device.h) with the optimizer TAIL of step t as rider workgroups (round 5).
// In the optimizer launch the workgroups behind the dW1 tiles (the dW2 tiles, the small tensors, step loss, step counter) cannot
// become resident beside a tile workgroup, so they run when tiles leave: 5.8 us of serial tail on otherwise idle CUs
// (profiles/r04_k_stamps_optimizer_launch.txt).  Their results (W2, W3, the biases, the step counter the dropout stream reads) are
// first read by the mid-forward launch of the NEXT step, behind that step's layer-1 product -- a launch whose tiles need 94
// registers and 72 KB of LDS and leave every CU room for a second workgroup.  There the tail has 30 us of slack: it takes
// ~1 000 matrix-pipe cycles per SIMD from the tiles it meets, at their side instead of behind them.  Riders are the LAST blocks of
// the grid (every CU has its tile first) and run the plain rmsprop_body: no look-ahead registers (the whole launch lives within
// 128), no priority.
__global__ __launch_bounds__(l1_dev::THREADS, 4) void l1_rms_kernel(l1_dev::L1Args l, RmsArgs a, const float *hyper, int64_t *ctl, int64_t batch_advance)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char l1_rms_smem[];
    if ((int)blockIdx.x < l.n_tiles) { l1_dev::l1_fwd_body<0>(l, (int)blockIdx.x, l1_rms_smem); return; }
    if (threadIdx.x >= RMS_THREADS) return;                  // (before any barrier: rmsprop_body is written for 256 threads)
    rmsprop_body<false>(a, hyper, ctl, batch_advance, 0, 0, idl_dev::GatherArgs{}, (int)blockIdx.x - l.n_tiles, (int)threadIdx.x);
}

// The launch between the two-plane layer-1 tiles (idl_l1_planes) and mid_fwd: its first r.blocks workgroups add the KPARTS partial sums
// part[p][i] in ascending p into part[0][i] (16-byte elements, all requests of a thread in flight before the first add) -- 16 MB read by
// every CU at once instead of by mid_fwd's 64 workgroups -- and the workgroups behind them are the previous step's optimizer tail
// (rmsprop_body), which nothing in this launch depends on: the two overlap.
struct ReduceArgs { float4 *part; int64_t slab4; int n_parts, blocks; const int64_t *ctl_src; int64_t *ctl_snap; };      // slab4: 16-byte elements of a slab
// (ctl_snap, only without a tail in the launch: a copy of the step counter for the dW1 launch that ends this step -- its workgroups read
//  the counter when they start and its own loader waves move it when they end: on a GPU shared with another process not every tile starts
//  before the first one ends)
constexpr int RED_U = 2;
__global__ __launch_bounds__(256) void reduce_rms_kernel(ReduceArgs r, RmsArgs a, const float *hyper, int64_t *ctl, int64_t batch_advance, int with_tail)
{
    // grid order: the tail's workgroups FIRST (theirs is the longer chain: 7.1 us on its own against the sums' 4.7), the sums behind them
    const int n_tail = (int)gridDim.x - r.blocks;
    if ((int)blockIdx.x >= n_tail) {
        if (r.ctl_snap != nullptr && (int)blockIdx.x == n_tail && threadIdx.x == 0) *r.ctl_snap = r.ctl_src[0];
        const int64_t i0 = (int64_t)((int)blockIdx.x - n_tail) * (256 * RED_U) + threadIdx.x;
        float4 v[RED_U][l1p_dev::KSPLIT];
#pragma unroll
        for (int u = 0; u < RED_U; ++u) {
            const int64_t i = i0 + 256 * u < r.slab4 ? i0 + 256 * u : r.slab4 - 1;     // clamped, not predicated
#pragma unroll
            for (int p = 0; p < l1p_dev::KSPLIT; ++p) v[u][p] = r.part[(int64_t)p * r.slab4 + i];
        }
#pragma unroll
        for (int u = 0; u < RED_U; ++u) {
            float4 acc = v[u][0];
#pragma unroll
            for (int p = 1; p < l1p_dev::KSPLIT; ++p) { acc.x += v[u][p].x; acc.y += v[u][p].y; acc.z += v[u][p].z; acc.w += v[u][p].w; }
            if (i0 + 256 * u < r.slab4) r.part[i0 + 256 * u] = acc;
        }
        return;
    }
    // (the look-ahead form of the dW2 tiles -- every operand of a wave requested before its first product: one trip to memory instead of
    //  three -- as behind the dW1 tiles of wgrad_rmsprop_kernel: here too the tail's chain is the launch's length, 9.3 us without)
    if (with_tail) rmsprop_body<true>(a, hyper, ctl, batch_advance, 0, 0, idl_dev::GatherArgs{}, (int)blockIdx.x, (int)threadIdx.x);
}

// The dW1 tiles of the two-plane form (wgrad_planes_device.h) with THIS step's optimizer tail carried by their loader waves: a workgroup's four
// loaders have their last chunk in two chunks before its computing waves start the epilogue (16 MB of W / square_avg / plane stores, ~5 us
// of the launch), and from then on have nothing to do -- block b of the tail (b < n_tail <= the tiles) is run by the loaders of workgroup
// b, in rmsprop_body's SPIN form (no s_barrier: the computing waves would not come).  The step counter and batch offset move at the
// END of this launch (every workgroup read the counter when it started: all tiles are resident at once, one per CU; the launcher
// refuses more tiles than CUs), so the next step's first launch adds up its partial sums alone (4.7 us instead of 9.4 with the tail).
// (wgrad_planes_device.h: dplanes_body, whose K-steps own v[200:247])
__global__ __launch_bounds__(wgp_dev::NT, 1) __attribute__((amdgpu_num_vgpr(200))) void wgrad_dplanes_rms_kernel(wgp_dev::XpArgs x, RmsArgs a, const float *hyper,
                                                                                                                 int64_t *ctl, int64_t batch_advance, int n_tail)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char wgd_rms_smem[];
    wgp_dev::dplanes_body(x, wgd_rms_smem, [&](int tix) {
        if ((int)blockIdx.x < n_tail)
            rmsprop_body<true, true>(a, hyper, ctl, batch_advance, 0, 0, idl_dev::GatherArgs{}, (int)blockIdx.x, tix,
                                     (unsigned int *)(wgd_rms_smem + wgp_dev::LDS_BYTES) + 3);
    });
}

// several voters in one launch: voter blockIdx.y takes its arguments from its plan record
__global__ __launch_bounds__(256) void rmsprop_batched_kernel(const unsigned char *__restrict__ plans)
{
    const RmsParams &p = *(const RmsParams *)(plans + (size_t)blockIdx.y * idl::PLAN_BYTES + idl::PLAN_PARAMS);
    rmsprop_body(p.a, p.hyper, p.ctl, p.batch_advance, 0, p.n_gather, p.g, (int)blockIdx.x, (int)threadIdx.x);
}

// ... and the optimizer launch with the dW1 tiles at its head (wgrad_rmsprop_kernel) for several voters
struct WgRmsParams { wg_dev::WgArgs w; RmsParams r; };
static_assert(sizeof(WgRmsParams) + idl::PLAN_PARAMS <= idl::PLAN_BYTES, "WgRmsParams does not fit a plan record");

__global__ __launch_bounds__(wg_dev::THREADS) void wgrad_rmsprop_batched_kernel(const unsigned char *__restrict__ plans)
{
    extern __shared__ wg_dev::f32x4_t wg_img[];
    const WgRmsParams &p = *(const WgRmsParams *)(plans + (size_t)blockIdx.y * idl::PLAN_BYTES + idl::PLAN_PARAMS);
    if ((int)blockIdx.x >= p.w.tiles && threadIdx.x >= 256) return;        // (as in wgrad_rmsprop_kernel: before any barrier)
    if ((int)blockIdx.x < p.w.tiles) wg_dev::q16_tile<0>(p.w, (int)blockIdx.x, wg_img);
    else {
        __builtin_amdgcn_s_setprio(3);
        rmsprop_body<true>(p.r.a, p.r.hyper, p.r.ctl, p.r.batch_advance, 0, p.r.n_gather, p.r.g, (int)blockIdx.x - p.w.tiles, (int)threadIdx.x);
    }
}

// a pointer read from a plan record is the same in every lane, but the compiler need not know: scalar operands of inline asm want it said
template <class T>
__device__ __forceinline__ T *uniform_ptr(T *p)
{
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (T *)(uintptr_t)(((uint64_t)hi << 32) | lo);
}

// ... and the launches of the step's two-plane form for several voters (blockIdx.y = voter, arguments from its plan record)
__global__ __launch_bounds__(l1p_dev::THREADS, 1) void l1_planes_batched_kernel(const unsigned char *__restrict__ plans)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char l1p_b_smem[];
    l1p_dev::L1pArgs p = *(const l1p_dev::L1pArgs *)(plans + (size_t)blockIdx.y * idl::PLAN_BYTES + idl::PLAN_PARAMS);
    p.wh = uniform_ptr(p.wh); p.wl = uniform_ptr(p.wl); p.xh = uniform_ptr(p.xh); p.xl = uniform_ptr(p.xl);      // (bases of the LDS-DMA: scalar operands)
    l1p_dev::l1p_body(p, (int)blockIdx.x, l1p_b_smem);
}

__global__ __launch_bounds__(256) void reduce_batched_kernel(const unsigned char *__restrict__ plans)
{
    const ReduceArgs &r = *(const ReduceArgs *)(plans + (size_t)blockIdx.y * idl::PLAN_BYTES + idl::PLAN_PARAMS);
    if ((int)blockIdx.x >= r.blocks) return;
    if (r.ctl_snap != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *r.ctl_snap = r.ctl_src[0];
    const int64_t i0 = (int64_t)blockIdx.x * (256 * RED_U) + threadIdx.x;
    float4 v[RED_U][l1p_dev::KSPLIT];
#pragma unroll
    for (int u = 0; u < RED_U; ++u) {
        const int64_t i = i0 + 256 * u < r.slab4 ? i0 + 256 * u : r.slab4 - 1;
#pragma unroll
        for (int p = 0; p < l1p_dev::KSPLIT; ++p) v[u][p] = r.part[(int64_t)p * r.slab4 + i];
    }
#pragma unroll
    for (int u = 0; u < RED_U; ++u) {
        float4 acc = v[u][0];
#pragma unroll
        for (int p = 1; p < l1p_dev::KSPLIT; ++p) { acc.x += v[u][p].x; acc.y += v[u][p].y; acc.z += v[u][p].z; acc.w += v[u][p].w; }
        if (i0 + 256 * u < r.slab4) r.part[i0 + 256 * u] = acc;
    }
}

struct XpRmsParams { wgp_dev::XpArgs x; RmsArgs a; const float *hyper; int64_t *ctl; int64_t batch_advance; int n_tail; };
static_assert(sizeof(XpRmsParams) + idl::PLAN_PARAMS <= idl::PLAN_BYTES, "XpRmsParams does not fit a plan record");
static_assert(sizeof(ReduceArgs) + idl::PLAN_PARAMS <= idl::PLAN_BYTES, "ReduceArgs does not fit a plan record");

__global__ __launch_bounds__(wgp_dev::NT, 1) __attribute__((amdgpu_num_vgpr(200))) void wgrad_xplanes_rms_batched_kernel(const unsigned char *__restrict__ plans)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char wgp_b_smem[];
    const XpRmsParams &p = *(const XpRmsParams *)(plans + (size_t)blockIdx.y * idl::PLAN_BYTES + idl::PLAN_PARAMS);
    wgp_dev::XpArgs x = p.x;
    x.xh = uniform_ptr(x.xh); x.xl = uniform_ptr(x.xl); x.dyh = uniform_ptr(x.dyh); x.dyl = uniform_ptr(x.dyl);
    auto tail = [&](int tix) {
        if ((int)blockIdx.x < p.n_tail)
            rmsprop_body<true, true>(p.a, p.hyper, p.ctl, p.batch_advance, 0, 0, idl_dev::GatherArgs{}, (int)blockIdx.x, tix,
                                     (unsigned int *)(wgp_b_smem + wgp_dev::LDS_BYTES) + 3);
    };
    wgp_dev::dplanes_body(x, wgp_b_smem, tail);
}

}  // namespace

// the BIG instances (n_clusters > 48) keep W3 in dynamic LDS beside 62 KB of static LDS: the limit is raised once per device
template <bool DP>
static int launch_mid_bwd_big(const MidBwdArgs &a, unsigned grid, int t0, int t1, const idl_dev::GatherArgs &g, void *stream)
{
    const int lds = a.C * H2 * (int)sizeof(float);
    static bool attr_set[64] = {};
    int dev = 0;
    IDL_HIP_TRY(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        IDL_HIP_TRY(hipFuncSetAttribute((const void *)mid_bwd_kernel<true, DP>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * MAX_CPL * H2 * (int)sizeof(float)));
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((mid_bwd_kernel<true, DP>), dim3(grid), dim3(64 * MID_WAVES), lds, (hipStream_t)stream, a, t0, t1, g);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

extern "C" {

int idl_relu_dropout_fwd(float *a, int64_t n, int train, uint64_t seed, const int64_t *ctl, int layer, void *stream)
{
    IDL_REQUIRE(a && ctl && n >= 0 && (n & 3) == 0 && (((uintptr_t)a) & 15u) == 0, "relu_dropout_fwd: n % 4 == 0, 16-byte aligned");
    if (n == 0) return IDL_OK;
    int64_t g = (n / 4 + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(relu_dropout_fwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (float4 *)a, n / 4, train, seed, ctl,
                       (uint32_t)layer);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

int idl_head_fwd(const float *lat, const float *W3, const float *b3, int m, int C, int train, uint64_t seed, const int64_t *ctl,
                 float *f, float *inv, float *r2, float *z, void *stream)
{
    IDL_REQUIRE(lat && W3 && b3 && ctl && f && inv && r2 && z, "NULL buffer");
    IDL_REQUIRE(m >= 1 && C >= 1 && C <= 64 * MAX_CPL, "head_fwd: n_clusters must be in 1..256");
    hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, (hipStream_t)stream, lat, W3, b3, m, C, train, seed,
                       ctl, f, inv, r2, z);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

int idl_mid_fwd(float *a1, const float *W2, const float *b2, const float *W3, const float *b3, int m, int C, int train,
                uint64_t seed, const int64_t *ctl, float *f, float *inv, float *r2, float *z, void *stream)
{
    IDL_REQUIRE(a1 && W2 && b2 && W3 && b3 && ctl && f && inv && r2 && z, "NULL buffer");
    IDL_REQUIRE(m >= 16 && (m % 16) == 0 && C >= 1 && C <= 64 * MAX_CPL, "mid_fwd: m must be a multiple of 16, n_clusters in 1..256");
    IDL_REQUIRE((((uintptr_t)a1 | (uintptr_t)W2 | (uintptr_t)W3) & 15u) == 0, "a1 / W2 / W3 must be 16-byte aligned");
    hipLaunchKernelGGL(mid_fwd_kernel<false>, dim3((unsigned)(m / 16)), dim3(64 * MID_WAVES), 0, (hipStream_t)stream, a1, (const float *)nullptr, W2, b2,
                       W3, b3, m, C, train, seed, ctl, f, inv, r2, z, m / 16, 0, 0, idl_dev::GatherArgs{});
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

static int mid_fwd_gather_impl(float *a1, const float *b1, int a1_transposed, const float *W2, const float *b2, const float *W3, const float *b3, int m,
                       int C, int train, uint64_t seed, const int64_t *ctl, float *f, float *inv, float *r2, float *z,
                       const float *feats, int64_t n, int64_t fdim, int64_t view_stride, const int64_t *pair_idx, const int64_t *base,
                       int64_t base_add, int64_t n_pairs, int64_t batch, const double *mean, const double *scale,
                       const double *inv_scale, float *y, int part, int part_end, int parts, void *stream, uint16_t *yh, uint16_t *yl, int *over = nullptr)
{
    IDL_REQUIRE(a1 && W2 && b2 && W3 && b3 && ctl && f && inv && r2 && z, "NULL buffer");
    IDL_REQUIRE(m >= 16 && (m % 16) == 0 && C >= 1 && C <= 64 * MAX_CPL, "mid_fwd: m must be a multiple of 16, n_clusters in 1..256");
    IDL_REQUIRE((((uintptr_t)a1 | (uintptr_t)W2 | (uintptr_t)W3) & 15u) == 0, "a1 / W2 / W3 must be 16-byte aligned");
    IDL_REQUIRE(parts >= 1 && part >= 0 && part <= part_end && part_end <= parts, "mid_fwd_gather: need 0 <= part <= part_end <= parts");
    IDL_REQUIRE(b1 == nullptr || (((uintptr_t)b1) & 15u) == 0, "mid_fwd_gather: b1 must be 16-byte aligned");
    IDL_REQUIRE(a1_transposed == 0 || a1_transposed == 1, "mid_fwd_gather: a1_transposed is 0 or 1");
    IDL_REQUIRE((yh != nullptr) == (yl != nullptr) && (yh == nullptr || (feats != nullptr && ((((uintptr_t)yh) | ((uintptr_t)yl)) & 7u) == 0)),
                "mid_fwd_gather: both planes of the next batch (8-byte aligned) or neither; planes need the batch assembly");
    idl_dev::GatherArgs g{};
    int64_t t0 = 0, t1 = 0;
    if (feats != nullptr) {                  // (feats == NULL: no batch assembly in this launch)
        IDL_REQUIRE(pair_idx && mean && scale && (y || yh) && n >= 1 && fdim >= 4 && (fdim & 3) == 0 && batch >= 1 && n_pairs >= 0,
                    "mid_fwd_gather: bad gather arguments (4 | f)");
        g = idl_dev::GatherArgs{feats, n, fdim, view_stride, pair_idx, base, batch, n_pairs, mean, scale, inv_scale, y, base_add, yh, yl, over};
        const int64_t ng = idl_dev::gather_tiles<MID_GATHER_ROWS>(fdim, batch);
        t0 = ng * part / parts; t1 = ng * part_end / parts;
    }
    const dim3 grid((unsigned)(m / 16 + (t1 - t0 + 3) / 4));
    if (void *plan = idl::take_plan()) {          // recorded, not launched (idl_plan_begin)
        idl::PlanHead h{};
        h.kind = idl::PLAN_MID_FWD; h.variant = a1_transposed; /* 0 row-major, 1 transposed */ h.grid[0] = grid.x; h.grid[1] = 1; h.grid[2] = 1; h.block = 64 * MID_WAVES;
        memcpy(plan, &h, sizeof(h));
        const MidFwdParams p{a1, b1, W2, b2, W3, b3, m, C, train, seed, ctl, f, inv, r2, z, m / 16, (int)t0, (int)t1, g};
        memcpy((unsigned char *)plan + idl::PLAN_PARAMS, &p, sizeof(p));
        return IDL_OK;
    }
    if (a1_transposed) hipLaunchKernelGGL(mid_fwd_kernel<true>, grid, dim3(64 * MID_WAVES), 0, (hipStream_t)stream, a1, b1, W2, b2, W3, b3,
                                          m, C, train, seed, ctl, f, inv, r2, z, m / 16, (int)t0, (int)t1, g);
    else hipLaunchKernelGGL(mid_fwd_kernel<false>, grid, dim3(64 * MID_WAVES), 0, (hipStream_t)stream, a1, b1, W2, b2, W3, b3,
                            m, C, train, seed, ctl, f, inv, r2, z, m / 16, (int)t0, (int)t1, g);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

int idl_mid_fwd_gather(float *a1, const float *b1, int a1_transposed, const float *W2, const float *b2, const float *W3, const float *b3, int m,
                       int C, int train, uint64_t seed, const int64_t *ctl, float *f, float *inv, float *r2, float *z,
                       const float *feats, int64_t n, int64_t fdim, int64_t view_stride, const int64_t *pair_idx, const int64_t *base,
                       int64_t base_add, int64_t n_pairs, int64_t batch, const double *mean, const double *scale,
                       const double *inv_scale, float *y, int part, int part_end, int parts, void *stream)
{
    return mid_fwd_gather_impl(a1, b1, a1_transposed, W2, b2, W3, b3, m, C, train, seed, ctl, f, inv, r2, z, feats, n, fdim, view_stride, pair_idx, base,
                               base_add, n_pairs, batch, mean, scale, inv_scale, y, part, part_end, parts, stream, nullptr, nullptr);
}

// ... the same launch with the next batch ALSO written as two fp16 planes (planes.h) by the assembling workgroups
int idl_mid_fwd_gather_planes(float *a1, const float *b1, int a1_transposed, const float *W2, const float *b2, const float *W3, const float *b3, int m,
                              int C, int train, uint64_t seed, const int64_t *ctl, float *f, float *inv, float *r2, float *z,
                              const float *feats, int64_t n, int64_t fdim, int64_t view_stride, const int64_t *pair_idx, const int64_t *base,
                              int64_t base_add, int64_t n_pairs, int64_t batch, const double *mean, const double *scale,
                              const double *inv_scale, float *y, void *y_hi, void *y_lo, int *overflow_flag, int part, int part_end, int parts,
                              void *stream)
{
    return mid_fwd_gather_impl(a1, b1, a1_transposed, W2, b2, W3, b3, m, C, train, seed, ctl, f, inv, r2, z, feats, n, fdim, view_stride, pair_idx, base,
                               base_add, n_pairs, batch, mean, scale, inv_scale, y, part, part_end, parts, stream, (uint16_t *)y_hi, (uint16_t *)y_lo,
                               overflow_flag);
}

int idl_nce_rows(float *S, int m, float temperature, float *lse, float *loss_rows, void *stream)
{
    IDL_REQUIRE(S && lse && loss_rows && m >= 2 && (m % 2) == 0 && temperature > 0.f, "nce_rows: even m >= 2, T > 0");
    const float inv_t = 1.f / temperature;
    hipLaunchKernelGGL(nce_lse_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S, m, inv_t, lse, loss_rows);
    int64_t g = ((int64_t)m * m + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(nce_esym_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, S, m, inv_t, lse);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

int idl_iic_core(float *P0, int C, float lamb, float eps, float w_iic, float *scratch, float *out, void *stream)
{
    IDL_REQUIRE(P0 && scratch && out, "NULL buffer");
    IDL_REQUIRE(C >= 1 && C <= 64 * MAX_CPL, "iic_core: n_clusters must be in 1..256");
    if (C <= 48) hipLaunchKernelGGL(iic_core_kernel<256>, dim3(1), dim3(256), 0, (hipStream_t)stream, P0, C, lamb, eps, w_iic, scratch, out);
    else if (C <= 200) {                     // the joint fits LDS (rows padded to an odd stride: C * (C | 1) * 4 + 1.9 KB <= 160 KB)
        const int lds = C * (C | 1) * 4;
        static bool attr_set[64] = {};
        int dev = 0;
        IDL_HIP_TRY(hipGetDevice(&dev));
        if (dev >= 0 && dev < 64 && !attr_set[dev]) {
            IDL_HIP_TRY(hipFuncSetAttribute((const void *)iic_core_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 200 * 201 * 4));
            attr_set[dev] = true;
        }
        // scratch: [0, C*C) the unshifted gradient; then (16-byte aligned) 2 doubles per workgroup of the first launch + the total
        const int G = (C + IIC_RPW - 1) / IIC_RPW;
        IDL_REQUIRE((((uintptr_t)scratch) & 15u) == 0, "iic_core: scratch must be 16-byte aligned");
        double *part = (double *)(scratch + ((C * C + 3) & ~3));           // 4 G + 2 <= 2 C - 3 floats for C >= 49
        hipLaunchKernelGGL(iic_core_rows_kernel, dim3(G), dim3(1024), lds, (hipStream_t)stream, P0, C, lamb, eps, scratch, part);
        hipLaunchKernelGGL(iic_core_shift_kernel, dim3((C * C + 1023) / 1024), dim3(256), 0, (hipStream_t)stream, P0, C, w_iic, out, scratch, part, G);
    }
    else hipLaunchKernelGGL(iic_core_kernel<512>, dim3(1), dim3(512), 0, (hipStream_t)stream, P0, C, lamb, eps, w_iic, scratch, out);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

// idl_iic_core for 48 < n_clusters <= 200 WITHOUT its second launch, and z dP0 instead: dzs[r][c] = sum_k z[r][k] dP0[k][c] for all m rows (what
// idl_mid_bwd_gather_planes takes as z_dP0); P0 keeps the joint, out[3] the IIC loss.  scratch as idl_iic_core's.
int idl_iic_core_dz(const float *P0, int C, float lamb, float eps, float w_iic, float *scratch, float *out, const float *z, int m, float *dzs, void *stream)
{
    IDL_REQUIRE(P0 && scratch && out && z && dzs && C > 48 && C <= 200 && m >= 1, "iic_core_dz: NULL buffer, or n_clusters outside 49..200");
    IDL_REQUIRE((((uintptr_t)scratch) & 15u) == 0, "iic_core_dz: scratch must be 16-byte aligned");
    const int lds = C * (C | 1) * 4;
    static bool attr_set[64] = {};
    int dev = 0;
    IDL_HIP_TRY(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        IDL_HIP_TRY(hipFuncSetAttribute((const void *)iic_core_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 200 * 201 * 4));
        attr_set[dev] = true;
    }
    const int G = (C + IIC_RPW - 1) / IIC_RPW;
    double *part = (double *)(scratch + ((C * C + 3) & ~3));
    hipLaunchKernelGGL(iic_core_rows_kernel, dim3(G), dim3(1024), lds, (hipStream_t)stream, P0, C, lamb, eps, scratch, part);
    const int groups = ((C + 15) / 16 + 3) / 4;              // (a workgroup: 16 rows x 4 column tiles)
    hipLaunchKernelGGL(iic_dz_kernel, dim3((unsigned)(((m + 15) / 16) * groups)), dim3(256), 0, (hipStream_t)stream, z, m, C, (const float *)scratch, (const double *)part, G, w_iic,
                       dzs, out);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

int idl_head_bwd(const float *z, const float *r2, const float *f, const float *inv, const float *G, int g_parts, const float *dP0,
                 const float *W3, int m, int C, int train, float nce_coef, float *dlogits, float *dlat, void *stream)
{
    IDL_REQUIRE(g_parts >= 1 && g_parts <= 16, "head_bwd: g_parts outside 1..16");
    IDL_REQUIRE(z && r2 && f && inv && G && dP0 && W3 && dlogits && dlat, "NULL buffer");
    IDL_REQUIRE(m >= 2 && (m % 2) == 0 && C >= 1 && C <= 64 * MAX_CPL, "head_bwd: even m, n_clusters in 1..256");
    hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, (hipStream_t)stream, z, r2, f, inv, G, g_parts, dP0, W3, m, C,
                       train, nce_coef, dlogits, dlat, (const float *)nullptr);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

int idl_head_bwd_dz(const float *z, const float *r2, const float *f, const float *inv, const float *G, int g_parts, const float *dzs,
                    const float *W3, int m, int C, int train, float nce_coef, float *dlogits, float *dlat, void *stream)
{
    IDL_REQUIRE(g_parts >= 1 && g_parts <= 16, "head_bwd_dz: g_parts outside 1..16");
    IDL_REQUIRE(z && r2 && f && inv && G && dzs && W3 && dlogits && dlat, "NULL buffer");
    IDL_REQUIRE(m >= 2 && (m % 2) == 0 && C >= 1 && C <= 64 * MAX_CPL, "head_bwd_dz: even m, n_clusters in 1..256");
    hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, (hipStream_t)stream, z, r2, f, inv, G, g_parts,
                       (const float *)nullptr, W3, m, C, train, nce_coef, dlogits, dlat, dzs);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

int idl_mid_bwd(const float *z, const float *r2, const float *f, const float *inv, const float *G, int g_parts, const float *dP0,
                const float *W3, const float *W2, const float *act1, int m, int C, int train, float nce_coef, float *dlogits, float *dlat,
                float *dr1, float *partial1, float *partial2, float *partial3, float *dW3_partial, int64_t *ctl, int64_t batch_advance,
                void *stream)
{
    IDL_REQUIRE(g_parts >= 1 && g_parts <= 16, "mid_bwd: g_parts outside 1..16");
    IDL_REQUIRE(z && r2 && f && inv && G && dP0 && W3 && W2 && act1 && dlogits && dlat && dr1 && partial1 && partial2 && partial3, "NULL buffer");
    IDL_REQUIRE(m >= 2 && (m % 2) == 0 && C >= 1 && C <= 64 * MAX_CPL, "mid_bwd: even m, n_clusters in 1..256");
    IDL_REQUIRE(dW3_partial == nullptr || C <= 48, "mid_bwd: dW3 partials need n_clusters <= 48");
    MidBwdArgs a{};
    a.z = z; a.r2 = r2; a.f = f; a.inv = inv; a.G = G; a.dP0 = dP0; a.W3 = W3; a.W2 = W2; a.act1 = act1;
    a.dlogits = dlogits; a.dlat = dlat; a.dr1 = dr1; a.partial1 = partial1; a.partial2 = partial2; a.partial3 = partial3;
    a.dW3_part = dW3_partial; a.ctl = ctl; a.batch_advance = batch_advance;
    a.g_parts = g_parts; a.m = m; a.C = C; a.train = train; a.nce_coef = nce_coef;
    if (C <= 48) hipLaunchKernelGGL((mid_bwd_kernel<false, false>), dim3(COL_PARTS), dim3(64 * MID_WAVES), 0, (hipStream_t)stream, a, 0, 0, idl_dev::GatherArgs{});
    else return launch_mid_bwd_big<false>(a, COL_PARTS, 0, 0, idl_dev::GatherArgs{}, stream);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}


static int mid_bwd_gather_impl(const float *z, const float *r2, const float *f, const float *inv, const float *G, int g_parts, const float *dP0,
                       const float *W3, const float *W2, const float *act1, int m, int C, int train, float nce_coef, float *dlogits,
                       float *dlat, float *dr1, float *partial1, float *partial2, float *partial3, float *dW3_partial,
                       const float *feats, int64_t n, int64_t fdim, int64_t view_stride, const int64_t *pair_idx, const int64_t *base,
                       int64_t base_add, int64_t n_pairs, int64_t batch, const double *mean, const double *scale,
                       const double *inv_scale, float *y, int part, int part_end, int parts, int act1_transposed, void *stream,
                       uint16_t *yh, uint16_t *yl, int *over = nullptr, uint16_t *dr1h = nullptr, uint16_t *dr1l = nullptr, int *dscale = nullptr,
                       const float *dzs = nullptr)
{
    IDL_REQUIRE((dr1h != nullptr) == (dr1l != nullptr) && (dr1h == nullptr || (dscale && over && ((((uintptr_t)dr1h) | ((uintptr_t)dr1l)) & 15u) == 0)),
                "mid_bwd_gather: dr1's planes need both planes (16-byte aligned), the words of their scale and the flag");
    IDL_REQUIRE((yh != nullptr) == (yl != nullptr) && (yh == nullptr || (feats != nullptr && ((((uintptr_t)yh) | ((uintptr_t)yl)) & 7u) == 0)),
                "mid_bwd_gather: both planes of the next batch (8-byte aligned) or neither; planes need the batch assembly");
    IDL_REQUIRE(parts >= 1 && part >= 0 && part <= part_end && part_end <= parts, "mid_bwd_gather: need 0 <= part <= part_end <= parts");
    IDL_REQUIRE(g_parts >= 1 && g_parts <= 16, "mid_bwd: g_parts outside 1..16");
    IDL_REQUIRE(z && r2 && f && inv && G && dP0 && W3 && W2 && act1 && dlogits && dlat && (dr1 || dr1h) && partial1 && partial2 && partial3, "NULL buffer");
    IDL_REQUIRE(m >= 2 && (m % 2) == 0 && C >= 1 && C <= 64 * MAX_CPL, "mid_bwd: even m, n_clusters in 1..256");
    IDL_REQUIRE(dW3_partial == nullptr || C <= 48, "mid_bwd: dW3 partials need n_clusters <= 48");
    MidBwdArgs a{};
    a.dr1h = dr1h; a.dr1l = dr1l; a.dscale = dscale; a.dover = over; a.dzs = dzs;
    IDL_REQUIRE(dzs == nullptr || C > 48, "mid_bwd_gather: z dP0 as an input is the n_clusters > 48 form");
    a.z = z; a.r2 = r2; a.f = f; a.inv = inv; a.G = G; a.dP0 = dP0; a.W3 = W3; a.W2 = W2; a.act1 = act1;
    a.dlogits = dlogits; a.dlat = dlat; a.dr1 = dr1; a.partial1 = partial1; a.partial2 = partial2; a.partial3 = partial3;
    a.dW3_part = dW3_partial; a.ctl = nullptr; a.batch_advance = 0;
    a.g_parts = g_parts; a.m = m; a.C = C; a.train = train; a.nce_coef = nce_coef; a.act1_t = act1_transposed ? 1 : 0;
    idl_dev::GatherArgs g{};
    int64_t t0 = 0, t1 = 0;
    if (feats != nullptr) {                  // (feats == NULL: no batch assembly in this launch)
        IDL_REQUIRE(pair_idx && mean && scale && (y || yh) && n >= 1 && fdim >= 4 && (fdim & 3) == 0 && batch >= 1 && n_pairs >= 0,
                    "mid_bwd_gather: bad gather arguments (4 | f)");
        g = idl_dev::GatherArgs{feats, n, fdim, view_stride, pair_idx, base, batch, n_pairs, mean, scale, inv_scale, y, base_add, yh, yl, over};
        const int64_t ng = idl_dev::gather_tiles<MID_GATHER_ROWS>(fdim, batch);
        t0 = ng * part / parts; t1 = ng * part_end / parts;
    }
    if (void *plan = idl::take_plan()) {          // recorded, not launched (idl_plan_begin)
        IDL_REQUIRE(C <= 48, "mid_bwd_gather: only the n_clusters <= 48 form can be recorded");
        idl::PlanHead h{};
        h.kind = idl::PLAN_MID_BWD; h.grid[0] = (unsigned)(COL_PARTS + (t1 - t0 + 3) / 4); h.grid[1] = 1; h.grid[2] = 1; h.block = 64 * MID_WAVES;
        memcpy(plan, &h, sizeof(h));
        const MidBwdParams p{a, (int)t0, (int)t1, g};
        memcpy((unsigned char *)plan + idl::PLAN_PARAMS, &p, sizeof(p));
        return IDL_OK;
    }
    if (C > 48) return dr1h != nullptr ? launch_mid_bwd_big<true>(a, (unsigned)(COL_PARTS + (t1 - t0 + 3) / 4), (int)t0, (int)t1, g, stream)
                                       : launch_mid_bwd_big<false>(a, (unsigned)(COL_PARTS + (t1 - t0 + 3) / 4), (int)t0, (int)t1, g, stream);
    if (dr1h != nullptr) hipLaunchKernelGGL((mid_bwd_kernel<false, true>), dim3((unsigned)(COL_PARTS + (t1 - t0 + 3) / 4)), dim3(64 * MID_WAVES), 0, (hipStream_t)stream,
                                            a, (int)t0, (int)t1, g);
    else if (C <= 48) hipLaunchKernelGGL((mid_bwd_kernel<false, false>), dim3((unsigned)(COL_PARTS + (t1 - t0 + 3) / 4)), dim3(64 * MID_WAVES), 0, (hipStream_t)stream, a, (int)t0,
                                    (int)t1, g);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

int idl_mid_bwd_gather(const float *z, const float *r2, const float *f, const float *inv, const float *G, int g_parts, const float *dP0,
                       const float *W3, const float *W2, const float *act1, int m, int C, int train, float nce_coef, float *dlogits,
                       float *dlat, float *dr1, float *partial1, float *partial2, float *partial3, float *dW3_partial,
                       const float *feats, int64_t n, int64_t fdim, int64_t view_stride, const int64_t *pair_idx, const int64_t *base,
                       int64_t base_add, int64_t n_pairs, int64_t batch, const double *mean, const double *scale,
                       const double *inv_scale, float *y, int part, int part_end, int parts, int act1_transposed, void *stream)
{
    return mid_bwd_gather_impl(z, r2, f, inv, G, g_parts, dP0, W3, W2, act1, m, C, train, nce_coef, dlogits, dlat, dr1, partial1, partial2, partial3,
                               dW3_partial, feats, n, fdim, view_stride, pair_idx, base, base_add, n_pairs, batch, mean, scale, inv_scale, y, part,
                               part_end, parts, act1_transposed, stream, nullptr, nullptr);
}

// ... the same launch with the next batch ALSO written as two fp16 planes (planes.h) by the assembling workgroups
int idl_mid_bwd_gather_planes(const float *z, const float *r2, const float *f, const float *inv, const float *G, int g_parts, const float *dP0,
                              const float *W3, const float *W2, const float *act1, int m, int C, int train, float nce_coef, float *dlogits,
                              float *dlat, float *dr1, float *partial1, float *partial2, float *partial3, float *dW3_partial,
                              const float *feats, int64_t n, int64_t fdim, int64_t view_stride, const int64_t *pair_idx, const int64_t *base,
                              int64_t base_add, int64_t n_pairs, int64_t batch, const double *mean, const double *scale,
                              const double *inv_scale, float *y, void *y_hi, void *y_lo, int *overflow_flag, int part, int part_end, int parts,
                              int act1_transposed, void *dr1_hi, void *dr1_lo, int *dr1_scale, const float *z_dP0, void *stream)
{
    return mid_bwd_gather_impl(z, r2, f, inv, G, g_parts, dP0, W3, W2, act1, m, C, train, nce_coef, dlogits, dlat, dr1, partial1, partial2, partial3,
                               dW3_partial, feats, n, fdim, view_stride, pair_idx, base, base_add, n_pairs, batch, mean, scale, inv_scale, y, part,
                               part_end, parts, act1_transposed, stream, (uint16_t *)y_hi, (uint16_t *)y_lo, overflow_flag, (uint16_t *)dr1_hi,
                               (uint16_t *)dr1_lo, dr1_scale, z_dP0);
}

int idl_col_sum_parts(void) { return COL_PARTS; }

static int launch_col_jobs(ColJobs &jobs, int njobs, void *stream)
{
    int nb = 0;
    for (int k = 0; k < 3; ++k) {
        jobs.first_block[k] = nb;
        if (k < njobs) nb += (jobs.j[k].n + 63) / 64;
    }
    jobs.first_block[3] = nb;
    if (jobs.dW3_part != nullptr) nb += 1;
    hipLaunchKernelGGL(col_partial_kernel, dim3((unsigned)nb, COL_PARTS), dim3(256), 0, (hipStream_t)stream, jobs);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

int idl_col_sum(const float *x, int m, int n, float *partial, void *stream)
{
    IDL_REQUIRE(x && partial && m >= 1 && n >= 1, "col_sum: NULL buffer or empty");
    ColJobs jobs{};
    jobs.m = m;
    jobs.j[0] = ColJob{(float *)x, nullptr, partial, n, 1.f};
    return launch_col_jobs(jobs, 1, stream);
}

int idl_relu_dropout_bwd_colsum(float *dx, const float *act, int m, int n, int train, float *partial, void *stream)
{
    IDL_REQUIRE(dx && act && partial && m >= 1 && n >= 1, "relu_dropout_bwd_colsum: NULL buffer or empty");
    ColJobs jobs{};
    jobs.m = m;
    jobs.j[0] = ColJob{dx, act, partial, n, train ? 2.f : 1.f};
    return launch_col_jobs(jobs, 1, stream);
}

int idl_bias_grads(float *dx1, const float *act1, int n1, float *partial1, const float *x2, int n2, float *partial2,
                   const float *x3, int n3, float *partial3, int m, int train, int64_t *ctl, int64_t batch_advance,
                   const float *r2, float *dW3_partial, void *stream)
{
    IDL_REQUIRE(dW3_partial == nullptr || (r2 != nullptr && n2 == H2 && n3 <= 48), "bias_grads: dW3 partials need r2, latent width 64, n_clusters <= 48");
    IDL_REQUIRE(dx1 && act1 && partial1 && x2 && partial2 && x3 && partial3 && m >= 1 && n1 >= 1 && n2 >= 1 && n3 >= 1,
                "bias_grads: NULL buffer or empty");
    ColJobs jobs{};
    jobs.m = m;
    jobs.j[0] = ColJob{dx1, act1, partial1, n1, train ? 2.f : 1.f};
    jobs.j[1] = ColJob{(float *)x2, nullptr, partial2, n2, 1.f};
    jobs.j[2] = ColJob{(float *)x3, nullptr, partial3, n3, 1.f};
    jobs.ctl = ctl;
    jobs.batch_advance = batch_advance;
    jobs.dlogits = x3; jobs.r2 = r2; jobs.dW3_part = dW3_partial; jobs.C = n3;
    return launch_col_jobs(jobs, 3, stream);
}

// IDELUCS_DEV=stamps=1: a device buffer of 1024 x 4 uint64 the stamped kernels write to (idl_debug_stamps copies it out)
static uint64_t *stamp_buffer()
{
    static uint64_t *buf = [] {
        const char *e = idl::dev_env("stamps");
        uint64_t *p = nullptr;
        if (e != nullptr && e[0] == '1' && hipMalloc((void **)&p, 1024 * 4 * sizeof(uint64_t)) == hipSuccess) (void)hipMemset(p, 0, 1024 * 4 * sizeof(uint64_t));
        else p = nullptr;
        return p;
    }();
    return buf;
}

static int g_phase_mode_host = 0;     // idl_debug_phase_stamps: while armed, the optimizer launch leaves the stamp buffer alone

static int rmsprop_launch(int count, float *const *params, const float *const *grads, const int32_t *grad_parts,
                          float *const *square_avg, const int64_t *sizes, const float *hyper, int64_t *ctl, int64_t batch_advance,
                          const float *loss_rows, int loss_m, float w_nce, float w_iic, float *out, const idl_dev::GatherArgs &g,
                          void *stream, int wg_index = -1, const float *wg_dy = nullptr, const float *wg_x = nullptr, int wg_m = 0,
                          int wg_n_out = 0, int wg_n_in = 0, float *wg_grad = nullptr, int wg_x_transposed = 0,
                          const wg_dev::WgArgs *big = nullptr, int big_index = -1, const l1_dev::L1Args *l1 = nullptr, int skip_index = -1,
                          const ReduceArgs *red = nullptr, const wgp_dev::XpArgs *xp = nullptr)
{
    IDL_REQUIRE(count >= 1 && count <= 8 && params && grads && square_avg && sizes && hyper && ctl, "rmsprop_step: 1..8 tensors");
    RmsArgs a{};
    a.count = count;
    a.loss_rows = loss_rows; a.out = (loss_rows != nullptr && loss_m > 0) ? out : nullptr; a.loss_m = loss_m; a.w_nce = w_nce; a.w_iic = w_iic;
    int64_t mx = 0;
    for (int i = 0; i < count; ++i) {
        a.p[i] = params[i]; a.g[i] = grads[i]; a.v[i] = square_avg[i]; a.n[i] = sizes[i];
        a.parts[i] = grad_parts ? grad_parts[i] : 1;
        IDL_REQUIRE(a.parts[i] >= 1, "rmsprop_step: grad_parts must be >= 1");
        if (sizes[i] > mx) mx = sizes[i];
    }
    if (wg_index >= 0) {
        IDL_REQUIRE(wg_index < count && wg_dy && wg_x && wg_m >= 1 && wg_n_out >= 16 && (wg_n_out % 16) == 0 && wg_n_in >= 16 &&
                    (wg_n_in % 16) == 0 && sizes[wg_index] == (int64_t)wg_n_out * wg_n_in && (int64_t)wg_m * wg_n_in < (1ll << 31),
                    "rmsprop_step: in-launch weight gradient needs n_out, n_in multiples of 16 and sizes[wg_index] == n_out * n_in");
        a.wg_t = wg_index; a.wg_dy = wg_dy; a.wg_x = wg_x; a.wg_grad = wg_grad; a.wg_m = wg_m; a.wg_n_out = wg_n_out; a.wg_n_in = wg_n_in;
        a.wg_tiles = (wg_n_out / 16) * (wg_n_in / 16);
        a.wg_xt = wg_x_transposed ? 1 : 0;
        IDL_REQUIRE(!wg_x_transposed || ((wg_m & 3) == 0 && (((uintptr_t)wg_x) & 15u) == 0), "rmsprop_step: transposed wg_x needs 4 | m and 16-byte alignment");
        a.n[wg_index] = 0;                  // its optimizer blocks have nothing to do: the tile workgroups update it
        mx = 0;
        for (int i = 0; i < count; ++i) if (a.n[i] > mx) mx = a.n[i];
    }
    (void)mx;
    if (skip_index >= 0 && skip_index < count) a.n[skip_index] = 0;     // (updated elsewhere: the dW1 tiles' own launch)
    if (big != nullptr) {                   // tensor big_index is updated by the tile workgroups at the head of the launch
        IDL_REQUIRE(big_index >= 0 && big_index < count && big_index != wg_index &&
                    sizes[big_index] == (int64_t)big->p.n_out * big->p.n_in && big->p.W == params[big_index] && big->p.V == square_avg[big_index],
                    "wgrad_rmsprop_step: w1_index must name the tensor the tiles update");
        a.n[big_index] = 0;
    }
    int nb_total = 0;
    for (int i = 0; i < count; ++i) {
        const bool vec = a.parts[i] == 1 && (a.n[i] & 3) == 0 && ((((uintptr_t)a.p[i]) | ((uintptr_t)a.g[i]) | ((uintptr_t)a.v[i])) & 15u) == 0;
        int64_t nb = vec ? (a.n[i] / 4 + 256 * RMS_UNROLL - 1) / (256 * RMS_UNROLL) : (a.n[i] + 255) / 256;     // float4 path: RMS_UNROLL 16-byte elements per thread
        if (nb > 1024) nb = 1024;
        a.first[i] = nb_total;
        nb_total += (int)nb;
    }
    if (nb_total == 0) nb_total = 1;        // (step counter / loss assembly still need a block)
    for (int i = count; i <= 8; ++i) a.first[i] = nb_total;
    const int64_t extra = g.y != nullptr ? idl_dev::gather_blocks(g.f, g.batch) : 0;
    if (xp != nullptr) {                    // ... carried by the loader waves of the two-plane dW1 tiles (wgrad_xplanes_rms_kernel)
        IDL_REQUIRE(big == nullptr && l1 == nullptr && red == nullptr && extra == 0, "wgrad_xplanes_rms: no other tiles, no batch assembly");
        idl::DeviceInfo di;
        if (const int rc = idl::device_info(&di); rc != IDL_OK) return rc;
        IDL_REQUIRE(nb_total + a.wg_tiles <= xp->tiles && xp->tiles <= di.cus, "wgrad_xplanes_rms: the tail's blocks need a tile each, and every tile its own CU");
        if (void *plan = idl::take_plan()) {      // recorded, not launched (idl_plan_begin)
            idl::PlanHead h{};
            h.kind = idl::PLAN_WGRAD_XPLANES; h.grid[0] = (unsigned)xp->tiles; h.grid[1] = 1; h.grid[2] = 1; h.block = wgp_dev::NT; h.lds = wgp_dev::LDS_BYTES + 16;
            memcpy(plan, &h, sizeof(h));
            const XpRmsParams xr{*xp, a, hyper, ctl, batch_advance, nb_total + a.wg_tiles};
            memcpy((unsigned char *)plan + idl::PLAN_PARAMS, &xr, sizeof(xr));
            return IDL_OK;
        }
        static bool attr_set[64] = {};
        int dev = 0;
        IDL_HIP_TRY(hipGetDevice(&dev));
        if (dev >= 0 && dev < 64 && !attr_set[dev]) {
            IDL_HIP_TRY(hipFuncSetAttribute((const void *)wgrad_dplanes_rms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, wgp_dev::LDS_BYTES + 16));
            attr_set[dev] = true;
        }
        hipLaunchKernelGGL(wgrad_dplanes_rms_kernel, dim3((unsigned)xp->tiles), dim3(wgp_dev::NT), wgp_dev::LDS_BYTES + 16, (hipStream_t)stream, *xp, a, hyper,
                           ctl, batch_advance, nb_total + a.wg_tiles);
        IDL_HIP_TRY(hipGetLastError());
        return IDL_OK;
    }
    if (red != nullptr) {                   // ... beside the workgroups that add up the two-plane layer-1 tiles' partial sums (reduce_rms_kernel)
        IDL_REQUIRE(big == nullptr && l1 == nullptr && extra == 0 && idl::take_plan() == nullptr, "reduce_parts_rms: no tiles, no batch assembly, not recordable");
        hipLaunchKernelGGL(reduce_rms_kernel, dim3((unsigned)(red->blocks + nb_total + a.wg_tiles)), dim3(256), 0, (hipStream_t)stream, *red, a, hyper, ctl,
                           batch_advance, 1);
        IDL_HIP_TRY(hipGetLastError());
        return IDL_OK;
    }
    if (l1 != nullptr) {                    // the optimizer blocks ride behind the layer-1 forward tiles of the next step (l1_rms_kernel)
        IDL_REQUIRE(big == nullptr && extra == 0 && idl::take_plan() == nullptr, "l1_fwd_rms: no dW1 tiles, no batch assembly, not recordable");
        static bool attr_set[64] = {};
        int dev = 0;
        IDL_HIP_TRY(hipGetDevice(&dev));
        if (dev >= 0 && dev < 64 && !attr_set[dev]) {
            IDL_HIP_TRY(hipFuncSetAttribute((const void *)l1_rms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, l1_dev::LDS_BYTES));
            attr_set[dev] = true;
        }
        hipLaunchKernelGGL(l1_rms_kernel, dim3((unsigned)(l1->n_tiles + nb_total + a.wg_tiles)), dim3(l1_dev::THREADS), l1_dev::LDS_BYTES,
                           (hipStream_t)stream, *l1, a, hyper, ctl, batch_advance);
        IDL_HIP_TRY(hipGetLastError());
        return IDL_OK;
    }
    if (void *plan = idl::take_plan()) {          // recorded, not launched (idl_plan_begin)
        idl::PlanHead h{};
        const RmsParams p{a, hyper, ctl, batch_advance, (int)extra, g};
        if (big != nullptr) {                       // the dW1 tiles ride at the head of every voter's share of the launch
            h.kind = idl::PLAN_WGRAD_RMSPROP; h.grid[0] = (unsigned)(big->tiles + nb_total + extra + a.wg_tiles); h.grid[1] = 1; h.grid[2] = 1;
            h.block = wg_dev::THREADS;
            memcpy(plan, &h, sizeof(h));
            const WgRmsParams wp{*big, p};
            memcpy((unsigned char *)plan + idl::PLAN_PARAMS, &wp, sizeof(wp));
            return IDL_OK;
        }
        h.kind = idl::PLAN_RMSPROP; h.grid[0] = (unsigned)(nb_total + extra + a.wg_tiles); h.grid[1] = 1; h.grid[2] = 1; h.block = 256;
        memcpy(plan, &h, sizeof(h));
        memcpy((unsigned char *)plan + idl::PLAN_PARAMS, &p, sizeof(p));
        return IDL_OK;
    }
    if (big != nullptr) {
        const dim3 grid((unsigned)(big->tiles + nb_total + extra + a.wg_tiles));
        static const bool skip = [] { const char *e = idl::dev_env("stamps_skip_tiles"); return e != nullptr && e[0] == '1'; }();
        static bool attr_set[64] = {};
        int dev = 0;
        IDL_HIP_TRY(hipGetDevice(&dev));
        if (dev >= 0 && dev < 64 && !attr_set[dev]) {
            IDL_HIP_TRY(hipFuncSetAttribute((const void *)wgrad_rmsprop_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, wg_dev::IMG_BYTES));
            IDL_HIP_TRY(hipFuncSetAttribute((const void *)wgrad_rmsprop_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, wg_dev::IMG_BYTES));
            IDL_HIP_TRY(hipFuncSetAttribute((const void *)wgrad_rmsprop_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, wg_dev::IMG_BYTES));
            attr_set[dev] = true;
        }
        const dim3 block(wg_dev::THREADS);
        const hipStream_t st_ = (hipStream_t)stream;
        if (uint64_t *st = g_phase_mode_host ? nullptr : stamp_buffer(); st != nullptr && skip)
            hipLaunchKernelGGL((wgrad_rmsprop_kernel<true, true>), grid, block, wg_dev::IMG_BYTES, st_, *big, a, hyper, ctl, batch_advance, (int)extra, g, st);
        else if (st != nullptr)
            hipLaunchKernelGGL(wgrad_rmsprop_kernel<true>, grid, block, wg_dev::IMG_BYTES, st_, *big, a, hyper, ctl, batch_advance, (int)extra, g, st);
        else
            hipLaunchKernelGGL(wgrad_rmsprop_kernel<false>, grid, block, wg_dev::IMG_BYTES, st_, *big, a, hyper, ctl, batch_advance, (int)extra, g,
                               (uint64_t *)nullptr);
    }
    else
        hipLaunchKernelGGL(rmsprop_kernel, dim3((unsigned)(nb_total + extra + a.wg_tiles)), dim3(256), 0, (hipStream_t)stream, a, hyper, ctl,
                           batch_advance, 0, (int)extra, g);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

int idl_rmsprop_step(int count, float *const *params, const float *const *grads, const int32_t *grad_parts,
                     float *const *square_avg, const int64_t *sizes, const float *hyper, int64_t *ctl, int64_t batch_advance,
                     const float *loss_rows, int loss_m, float w_nce, float w_iic, float *out, void *stream)
{
    idl_dev::GatherArgs g{};
    return rmsprop_launch(count, params, grads, grad_parts, square_avg, sizes, hyper, ctl, batch_advance, loss_rows, loss_m, w_nce, w_iic,
                          out, g, stream);
}

int idl_rmsprop_step_gather(int count, float *const *params, const float *const *grads, const int32_t *grad_parts,
                            float *const *square_avg, const int64_t *sizes, const float *hyper, int64_t *ctl,
                            const float *loss_rows, int loss_m, float w_nce, float w_iic, float *out,
                            const float *feats, int64_t n, int64_t f, int64_t view_stride, const int64_t *pair_idx, int64_t n_pairs,
                            int64_t batch, const double *mean, const double *scale, const double *inv_scale, float *y, void *stream)
{
    IDL_REQUIRE(feats && pair_idx && mean && scale && y && n >= 1 && f >= 1 && batch >= 1 && n_pairs >= 0, "rmsprop_step_gather: bad gather arguments");
    idl_dev::GatherArgs g{feats, n, f, view_stride, pair_idx, ctl + 1, batch, n_pairs, mean, scale, inv_scale, y};
    return rmsprop_launch(count, params, grads, grad_parts, square_avg, sizes, hyper, ctl, 0, loss_rows, loss_m, w_nce, w_iic, out, g, stream);
}

int idl_rmsprop_step_gather_wgrad(int count, float *const *params, const float *const *grads, const int32_t *grad_parts,
                                  float *const *square_avg, const int64_t *sizes, const float *hyper, int64_t *ctl,
                                  const float *loss_rows, int loss_m, float w_nce, float w_iic, float *out,
                                  const float *feats, int64_t n, int64_t f, int64_t view_stride, const int64_t *pair_idx, int64_t n_pairs,
                                  int64_t batch, const double *mean, const double *scale, const double *inv_scale, float *y,
                                  int wg_index, const float *wg_dy, const float *wg_x, int wg_x_transposed, int wg_m, int wg_n_out, int wg_n_in,
                                  float *wg_grad, int64_t batch_advance, void *stream)
{
    idl_dev::GatherArgs g{};
    if (feats != nullptr) {
        IDL_REQUIRE(pair_idx && mean && scale && y && n >= 1 && f >= 1 && batch >= 1 && n_pairs >= 0, "rmsprop_step_gather_wgrad: bad gather arguments");
        g = idl_dev::GatherArgs{feats, n, f, view_stride, pair_idx, ctl + 1, batch, n_pairs, mean, scale, inv_scale, y};
    }
    return rmsprop_launch(count, params, grads, grad_parts, square_avg, sizes, hyper, ctl, batch_advance, loss_rows, loss_m, w_nce, w_iic, out, g,
                          stream, wg_index, wg_dy, wg_x, wg_m, wg_n_out, wg_n_in, wg_grad, wg_x_transposed);
}

int idl_wgrad_rmsprop_step(int count, float *const *params, const float *const *grads, const int32_t *grad_parts,
                           float *const *square_avg, const int64_t *sizes, const float *hyper, int64_t *ctl,
                           const float *loss_rows, int loss_m, float w_nce, float w_iic, float *out,
                           int w1_index, const float *w1_dy, const float *w1_x, int w1_m, int w1_n_out, int w1_n_in, float *w1_grad,
                           int wg_index, const float *wg_dy, const float *wg_x, int wg_x_transposed, int wg_m, int wg_n_out, int wg_n_in,
                           float *wg_grad, int64_t batch_advance, void *stream)
{
    IDL_REQUIRE(count >= 1 && count <= 8 && params && square_avg && w1_index >= 0 && w1_index < count, "wgrad_rmsprop_step: w1_index outside the tensors");
    IDL_REQUIRE(w1_dy && w1_x && wg_dev::supported(w1_m, w1_n_out, w1_n_in), "wgrad_rmsprop_step: m % 32 == 0, n_out % 64 == 0, n_in % 128 == 0");
    IDL_REQUIRE((((uintptr_t)w1_dy | (uintptr_t)w1_x | (uintptr_t)w1_grad | (uintptr_t)params[w1_index] | (uintptr_t)square_avg[w1_index]) & 15u) == 0,
                "wgrad_rmsprop_step: buffers must be 16-byte aligned");
    IDL_REQUIRE((int64_t)w1_m * w1_n_in < (1ll << 29) && (int64_t)w1_n_out * w1_n_in < (1ll << 29), "wgrad_rmsprop_step: operands beyond 2^31 bytes");
    wg_dev::WgArgs w{};
    w.p = wg_dev::WgProblem{w1_dy, w1_x, w1_grad, params[w1_index], square_avg[w1_index], w1_n_out, w1_n_in};
    w.hyper = hyper; w.m = w1_m;
    w.tiles_m = w1_n_out / wg_dev::TM;
    w.tiles = w.tiles_m * (w1_n_in / wg_dev::TN);
    idl_dev::GatherArgs g{};
    return rmsprop_launch(count, params, grads, grad_parts, square_avg, sizes, hyper, ctl, batch_advance, loss_rows, loss_m, w_nce, w_iic, out, g,
                          stream, wg_index, wg_dy, wg_x, wg_m, wg_n_out, wg_n_in, wg_grad, wg_x_transposed, &w, w1_index);
}

int idl_l1_fwd_rms(const float *W1, const float *x, int m, int n_in, float *r1_transposed,
                   int count, float *const *params, const float *const *grads, const int32_t *grad_parts,
                   float *const *square_avg, const int64_t *sizes, const float *hyper, int64_t *ctl,
                   const float *loss_rows, int loss_m, float w_nce, float w_iic, float *out, int w1_index,
                   int wg_index, const float *wg_dy, const float *wg_x, int wg_x_transposed, int wg_m, int wg_n_out, int wg_n_in,
                   float *wg_grad, int64_t batch_advance, void *stream)
{
    IDL_REQUIRE(W1 && x && r1_transposed && idl_l1_fwd_supported(m, l1_dev::H1, n_in), "l1_fwd_rms: Linear(n_in, 512), m % 32 == 0, n_in % 64 == 0, n_in >= 192");
    IDL_REQUIRE((((uintptr_t)W1 | (uintptr_t)x | (uintptr_t)r1_transposed) & 15u) == 0, "l1_fwd_rms: buffers must be 16-byte aligned");
    const int n_tiles = (l1_dev::H1 / l1_dev::TH) * (m / l1_dev::TR);
    static const int prio = [] { const char *e = idl::dev_env("l1_prio"); return e ? atoi(e) : 0; }();     // (A/B knob: measured)
    const l1_dev::L1Args l{W1, x, r1_transposed, m, n_in, n_tiles, prio};
    idl_dev::GatherArgs g{};
    return rmsprop_launch(count, params, grads, grad_parts, square_avg, sizes, hyper, ctl, batch_advance, loss_rows, loss_m, w_nce, w_iic, out, g,
                          stream, wg_index, wg_dy, wg_x, wg_m, wg_n_out, wg_n_in, wg_grad, wg_x_transposed, nullptr, -1, &l, w1_index);
}


// part[p][i], p < idl_l1_planes_parts(), i < slab_elems (4 | slab_elems): part[0][i] = ((part[0][i] + part[1][i]) + ...) in ascending p.
// With count >= 1: the previous step's optimizer tail in the same launch (arguments as idl_l1_fwd_rms); count == 0: the sums alone.
int idl_reduce_parts_rms(float *part, int64_t slab_elems, const int64_t *step_counter, int64_t *step_snapshot,
                         int count, float *const *params, const float *const *grads, const int32_t *grad_parts,
                         float *const *square_avg, const int64_t *sizes, const float *hyper, int64_t *ctl,
                         const float *loss_rows, int loss_m, float w_nce, float w_iic, float *out, int w1_index,
                         int wg_index, const float *wg_dy, const float *wg_x, int wg_x_transposed, int wg_m, int wg_n_out, int wg_n_in,
                         float *wg_grad, int64_t batch_advance, void *stream)
{
    IDL_REQUIRE(part && slab_elems >= 4 && (slab_elems & 3) == 0 && (((uintptr_t)part) & 15u) == 0 && slab_elems < (1ll << 31), "reduce_parts_rms: 4 | slab_elems, 16-byte aligned");
    IDL_REQUIRE((step_counter != nullptr) == (step_snapshot != nullptr) && (step_snapshot == nullptr || count == 0),
                "reduce_parts_rms: the step counter's snapshot needs both pointers and a launch without the tail (which moves the counter)");
    ReduceArgs r{(float4 *)part, slab_elems / 4, l1p_dev::KSPLIT, (int)((slab_elems / 4 + 256 * RED_U - 1) / (256 * RED_U)), step_counter, step_snapshot};
    if (count == 0) {
        if (void *plan = idl::take_plan()) {      // recorded, not launched (idl_plan_begin)
            idl::PlanHead h{};
            h.kind = idl::PLAN_REDUCE; h.grid[0] = (unsigned)r.blocks; h.grid[1] = 1; h.grid[2] = 1; h.block = 256;
            memcpy(plan, &h, sizeof(h));
            memcpy((unsigned char *)plan + idl::PLAN_PARAMS, &r, sizeof(r));
            return IDL_OK;
        }
        hipLaunchKernelGGL(reduce_rms_kernel, dim3((unsigned)r.blocks), dim3(256), 0, (hipStream_t)stream, r, RmsArgs{}, (const float *)nullptr, (int64_t *)nullptr,
                           (int64_t)0, 0);
        IDL_HIP_TRY(hipGetLastError());
        return IDL_OK;
    }
    idl_dev::GatherArgs g{};
    return rmsprop_launch(count, params, grads, grad_parts, square_avg, sizes, hyper, ctl, batch_advance, loss_rows, loss_m, w_nce, w_iic, out, g,
                          stream, wg_index, wg_dy, wg_x, wg_m, wg_n_out, wg_n_in, wg_grad, wg_x_transposed, nullptr, -1, nullptr, w1_index, &r);
}

// idl_wgrad_rmsprop_xplanes (W updated, its planes written) with THIS step's optimizer tail carried by the tiles' loader waves (tail arguments as
// idl_l1_fwd_rms; w1_index: the tensor the tiles update, left out of the tail).  Needs a CU per tile and no more tail blocks than tiles.
int idl_wgrad_xplanes_rms(const void *dy_hi, const void *dy_lo, int *dy_scale, const void *x_hi, const void *x_lo, int ld_x, int m, int n_out, int n_in, float *grad, float *W,
                          float *square_avg, void *w_hi, void *w_lo, int *overflow_flag,
                          int count, float *const *params, const float *const *grads, const int32_t *grad_parts,
                          float *const *square_avg_all, const int64_t *sizes, const float *hyper, int64_t *ctl,
                          const float *loss_rows, int loss_m, float w_nce, float w_iic, float *out, int w1_index,
                          int wg_index, const float *wg_dy, const float *wg_x, int wg_x_transposed, int wg_m, int wg_n_out, int wg_n_in,
                          float *wg_grad, int64_t batch_advance, void *stream)
{
    IDL_REQUIRE(dy_hi && dy_lo && dy_scale && x_hi && x_lo && W && square_avg && hyper && ctl && w_hi && w_lo && overflow_flag, "wgrad_xplanes_rms: NULL buffer");
    IDL_REQUIRE(idl_wgrad_xplanes_supported(m, n_out, n_in), "wgrad_xplanes_rms: m % 64 == 0, m >= 192, n_out % 64 == 0, n_in % 128 == 0");
    IDL_REQUIRE(ld_x >= n_in && ld_x <= n_in + 1024 && (ld_x & 7) == 0, "wgrad_xplanes_rms: n_in <= ld_x <= n_in + 1024, 8 | ld_x");
    IDL_REQUIRE((((uintptr_t)dy_hi | (uintptr_t)dy_lo | (uintptr_t)x_hi | (uintptr_t)x_lo | (uintptr_t)grad | (uintptr_t)W | (uintptr_t)square_avg) & 15u) == 0 &&
                (((uintptr_t)w_hi | (uintptr_t)w_lo) & 7u) == 0, "wgrad_xplanes_rms: buffers 16-byte aligned, W's planes 8-byte");
    wgp_dev::XpArgs x{};
    x.dyh = (const uint16_t *)dy_hi; x.dyl = (const uint16_t *)dy_lo; x.dy_scale = dy_scale;
    x.xh = (const uint16_t *)x_hi; x.xl = (const uint16_t *)x_lo; x.grad = grad; x.W = W; x.V = square_avg;
    x.wh = (uint16_t *)w_hi; x.wl = (uint16_t *)w_lo; x.over = overflow_flag; x.hyper = hyper;
    x.m = m; x.n_out = n_out; x.n_in = n_in; x.ldx = ld_x;
    x.tiles_m = n_out / wgp_dev::TM; x.tiles = x.tiles_m * (n_in / wgp_dev::TN);
    static const int wgp_dbg = idl::dev_env("wgp_dbg") ? atoi(idl::dev_env("wgp_dbg")) : 0;      // (timing ablations; wgrad_planes_device.h)
    x.dbg = wgp_dbg;
    idl_dev::GatherArgs g{};
    return rmsprop_launch(count, params, grads, grad_parts, square_avg_all, sizes, hyper, ctl, batch_advance, loss_rows, loss_m, w_nce, w_iic, out, g,
                          stream, wg_index, wg_dy, wg_x, wg_m, wg_n_out, wg_n_in, wg_grad, wg_x_transposed, nullptr, -1, nullptr, w1_index, nullptr, &x);
}

#ifdef IDL_PHASE_STAMPS
int nce_set_phase_stamps(uint64_t *st, int on);
#endif

int idl_debug_phase_stamps(int on)
{
#ifdef IDL_PHASE_STAMPS
    uint64_t *st = on ? stamp_buffer() : nullptr;
    IDL_REQUIRE(!on || st != nullptr, "debug stamps are off (IDELUCS_DEV=stamps=1 before the first launch)");
    IDL_HIP_TRY(hipDeviceSynchronize());
    if (st != nullptr) IDL_HIP_TRY(hipMemset(st, 0, 1024 * 4 * sizeof(uint64_t)));
    IDL_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(idl_phase_stamps), &st, sizeof(st)));
    IDL_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(idl_phase_mode), &on, sizeof(on)));
    { const int rc = nce_set_phase_stamps(st, on); if (rc != IDL_OK) return rc; }      // (modes 3, 4: the InfoNCE passes, nce_fused.hip)
    g_phase_mode_host = on;
    return IDL_OK;
#else
    (void)on; (void)g_phase_mode_host;
    IDL_REQUIRE(false, "this build has no phase stamps (make -C idelucs_amd/csrc STAMPS=1)");
#endif
}

int idl_debug_stamps(uint64_t *host_out)
{
    IDL_REQUIRE(host_out != nullptr, "NULL buffer");
    uint64_t *st = stamp_buffer();
    IDL_REQUIRE(st != nullptr, "debug stamps are off (IDELUCS_DEV=stamps=1 before the first launch)");
    IDL_HIP_TRY(hipDeviceSynchronize());
    IDL_HIP_TRY(hipMemcpy(host_out, st, 1024 * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return IDL_OK;
}

int idl_plan_launch(const void *host_plans, const void *dev_plans, int n_voters, void *stream)
{
    IDL_REQUIRE(host_plans && dev_plans && n_voters >= 1 && n_voters <= 65535, "plan_launch: NULL records or n_voters outside 1..65535");
    idl::PlanHead h;
    memcpy(&h, host_plans, sizeof(h));
    for (int v = 1; v < n_voters; ++v)        // one launch serves every voter: their launches must have the same shape
        IDL_REQUIRE(memcmp((const unsigned char *)host_plans + (size_t)v * idl::PLAN_BYTES, &h, sizeof(h)) == 0,
                    "plan_launch: the voters' records differ in kind or launch shape");
    const unsigned char *dp = (const unsigned char *)dev_plans;
    const hipStream_t st = (hipStream_t)stream;
    switch (h.kind) {
    case idl::PLAN_MID_FWD:
        if (h.variant) hipLaunchKernelGGL(mid_fwd_batched_kernel<true>, dim3((unsigned)n_voters, h.grid[0]), dim3(h.block), 0, st, dp);
        else hipLaunchKernelGGL(mid_fwd_batched_kernel<false>, dim3((unsigned)n_voters, h.grid[0]), dim3(h.block), 0, st, dp);
        break;
    case idl::PLAN_MID_BWD:
        hipLaunchKernelGGL(mid_bwd_batched_kernel, dim3((unsigned)n_voters, h.grid[0]), dim3(h.block), 0, st, dp);
        break;
    case idl::PLAN_RMSPROP:
        hipLaunchKernelGGL(rmsprop_batched_kernel, dim3(h.grid[0], (unsigned)n_voters), dim3(h.block), 0, st, dp);
        break;
    case idl::PLAN_WGRAD_RMSPROP: {
        static bool attr_set[64] = {};
        int dev = 0;
        IDL_HIP_TRY(hipGetDevice(&dev));
        if (dev >= 0 && dev < 64 && !attr_set[dev]) {
            IDL_HIP_TRY(hipFuncSetAttribute((const void *)wgrad_rmsprop_batched_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, wg_dev::IMG_BYTES));
            attr_set[dev] = true;
        }
        hipLaunchKernelGGL(wgrad_rmsprop_batched_kernel, dim3(h.grid[0], (unsigned)n_voters), dim3(h.block), wg_dev::IMG_BYTES, st, dp);
        break;
    }
    case idl::PLAN_L1_PLANES:
    case idl::PLAN_WGRAD_XPLANES: {
        static bool attr_set[64] = {};
        int dev = 0;
        IDL_HIP_TRY(hipGetDevice(&dev));
        if (dev >= 0 && dev < 64 && !attr_set[dev]) {
            IDL_HIP_TRY(hipFuncSetAttribute((const void *)l1_planes_batched_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, l1p_dev::LDS_BYTES));
            IDL_HIP_TRY(hipFuncSetAttribute((const void *)wgrad_xplanes_rms_batched_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, wgp_dev::LDS_BYTES + 16));
            attr_set[dev] = true;
        }
        if (h.kind == idl::PLAN_L1_PLANES)
            hipLaunchKernelGGL(l1_planes_batched_kernel, dim3(h.grid[0], (unsigned)n_voters), dim3(h.block), l1p_dev::LDS_BYTES, st, dp);
        else
            hipLaunchKernelGGL(wgrad_xplanes_rms_batched_kernel, dim3(h.grid[0], (unsigned)n_voters), dim3(h.block), wgp_dev::LDS_BYTES + 16, st, dp);
        break;
    }
    case idl::PLAN_REDUCE:
        hipLaunchKernelGGL(reduce_batched_kernel, dim3(h.grid[0], (unsigned)n_voters), dim3(h.block), 0, st, dp);
        break;
    case idl::PLAN_NCE:
        return idl::nce_plan_launch(h, dev_plans, n_voters, st);
    default:
        IDL_REQUIRE(false, "plan_launch: not a recorded launch");
    }
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

}  // extern "C"
