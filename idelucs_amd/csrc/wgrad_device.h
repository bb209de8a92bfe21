// wgrad_device.h -- weight gradient of Linear(F,512) on the fp32 matrix cores with the RMSprop update in the epilogue: the device
// side, shared by wgrad.hip (the kernel on its own) and train_step.hip (the same tiles as the head of the optimizer launch).
//
// Reference: loss.backward() produces Linear(F,512).weight.grad = dy^T x (torch autograd, idelucs/models.py:131) and
// optimizer.step() (models.py:132, torch.optim.RMSprop(lr, weight_decay=0.01), models.py:88) then reads that 8 MB gradient back
// together with the parameter and its square_avg.  Here the gradient tile never leaves the registers it was accumulated in:
//   g = dy^T x + wd W;  v = alpha v + (1 - alpha) g^2;  W -= lr g / (sqrt(v) + eps).
//
// Both operands are read the way they sit in memory: dy [m, n_out] and x [m, n_in] are row-major with the contraction index m
// as the ROW, which is the MFMA operand map (lane: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15]) -- no
// transposes, no LDS image, no barrier.  A workgroup owns a 64 x 128 tile as 2 x 2 waves of 32 rows x 64 columns, each wave the whole
// contraction (no cross-wave reduction: round 2's split-K kernel spent 8 us in that tail).
//
// A k-QUAD per ring slot: one 8-byte load (A: rows h0 + 2 i + rb of dy) and one 16-byte load (B: columns f0 + 4 j + cb of x) feed
// EIGHT v_mfma_f32_16x16x4_f32 (2 x 4 accumulators of 16 x 16) = 256 matrix-pipe cycles; whole 128 / 256-byte row segments per
// quarter wave.  The loads run D = 8 quads ahead in a register ring.  In the C/D layout (row 4 q + reg, column l) a lane's four
// column blocks are ADJACENT elements of a row, so the epilogue is 16-byte accesses of W and square_avg, requested before the
// first product.
//
// The main loop is ONE asm statement (wgrad_loop.inc, written by tools/gen_wgrad_loop.py).  Left to the compiler the ring does not
// survive: its slots are renamed across the back edge (v_mov copies of registers whose loads are in flight force s_waitcnt
// vmcnt(0) at the end of every pass) and every load carries 64-bit VALU address arithmetic.  In the asm a slot is a fixed
// register, a load is `global_load v, voffset, s[base]` with the base advanced by two scalar adds, and slot u is consumed behind
// `s_waitcnt vmcnt(2 (D - 1))`: exactly the loads issued after its own stay in flight.
//
// Measured on MI355X at m = 1024, n_out = 512, n_in = 4096 (tools/bench_wgrad.py, back to back in a HIP graph; profiles/r03_*):
// the MFMA stream alone (loads removed) 31.3 us -- 65 536 matrix-pipe cycles per wave, i.e. the chip holds ~2.2 GHz under this
// load, and the tuned hipBLASLt kernel (32.3 us in the step) sits on that floor too; gradient only 34.2 us; with the fused update
// 38.6 us, against 32.3 + 11.8 us for hipBLASLt followed by the optimizer launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "planes.h"
#include "wgrad_loop.inc"

namespace wg_dev {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int TM = 64, TN = 128;      // workgroup tile
constexpr int RING = 8;               // k-quads in flight per wave

struct WgProblem {
    const float *dy, *x;      // [m, n_out], [m, n_in]
    float *grad;              // [n_out, n_in] or NULL
    float *W, *V;             // parameter and square_avg, or NULL (then grad must be given)
    int n_out, n_in;
    uint16_t *Wh, *Wl;        // optional: the updated W ALSO as two fp16 planes (planes.h, scale 2^W_EXP) for the next layer-1 product
    int *over;                // ... and the flag raised when an entry left the planes' range
};

struct WgArgs {
    WgProblem p;
    const float *hyper;       // [lr, alpha, eps, weight_decay, 1 - alpha]
    int m, tiles_m, tiles;
};

struct Hyper { float lr, alpha, eps, wd, oma; };

// (v_sqrt_f32 and v_rcp_f32, 1 ulp each, instead of the correctly rounded sqrtf and IEEE division: two instructions instead of
//  ~25 per element -- 32 elements per lane stand between the last MFMA and the last store, 2 us of VALU per launch with the exact
//  forms; the step moves a weight by lr * g / (sqrt(v) + eps), so the parameter differs by < 1e-7 of the step)
__device__ __forceinline__ void rms_update(float g, float &p, float &v, const Hyper &h)
{
    const float gi = g + h.wd * p;                   // grad.add(param, alpha=weight_decay)
    v = v * h.alpha + h.oma * gi * gi;               // square_avg.mul_(alpha).addcmul_(g, g, value=1-alpha)
    p = p - h.lr * (gi * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + h.eps));        // param.addcdiv_(grad, sqrt(v)+eps, value=-lr)
}

__host__ __device__ inline bool supported(int m, int n_out, int n_in)
{
    return m >= 4 * RING && (m % (4 * RING)) == 0 && n_out >= TM && (n_out % TM) == 0 && n_in >= TN && (n_in % TN) == 0;
}

// One 64 x 128 tile by one 256-thread workgroup (bid = tile index).  VARIANT 1: no loads in the loop (a diagnostic: what the
// MFMA stream alone takes; wrong results).
// The workgroup is FIVE waves (THREADS = 320): waves 0..3 compute; wave 4 fetches the tile's W and square_avg elements (64 KB) into
// the LDS image `img` while they do, laid out so that a compute wave's epilogue reads are lane-contiguous 16-byte reads.  (A
// wave's vector-memory operations retire in order: requested by the compute waves themselves -- as in this kernel's first form --
// those 16 MB from HBM stood between every wave and its first operands, ~2 us per launch.)
constexpr int THREADS = 320;
constexpr int IMG_BYTES = 2 * 4 * 8 * 64 * 16;       // [W | square_avg][compute wave][rb * 4 + reg][lane] float4

// VARIANT 2 (a diagnostic): the ring loaded once with real operands and never refilled -- the MFMA stream alone on the data's bit
// patterns.  CLK (a diagnostic, idl_debug_wgrad_clock): every compute wave stamps s_memtime (shader cycles) and s_memrealtime
// (100 MHz) right before the first and right after the last MFMA into clk[(bid * 4 + wave) * 4 ..]: the clock the chip holds
// under this stream is (t1 - t0) / (r1 - r0) x 100 MHz.  No stamp executes in the product instantiation.
template <int VARIANT = 0, bool CLK = false>
__device__ __forceinline__ void q16_tile(const WgArgs &a, const int bid, f32x4_t *img, uint64_t *clk = nullptr)
{
    constexpr int D = RING;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = lane & 15, q = lane >> 4;
    // consecutive tiles of an XCD (blocks b, b + 8, ...) share dy (2 MB at cfg2) and a 512-column panel of x (2 MB): L2
    int tile = bid;
    if ((a.tiles & 7) == 0) tile = (bid & 7) * (a.tiles >> 3) + (bid >> 3);
    const int lda = a.p.n_out, ldb = a.p.n_in;
    if (wv == 4) {                           // the fetch wave
        if (a.p.W != nullptr) {
            const int th0 = (tile % a.tiles_m) * TM, tf0 = (tile / a.tiles_m) * TN;
#pragma unroll 1
            for (int b = 0; b < 4; ++b) {            // batches of 16 requests (64 registers: the kernel's budget is 160)
                const float *src = b < 2 ? a.p.W : a.p.V;
                f32x4_t t[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {      // (b & 1) * 16 + i = compute wave * 8 + rb * 4 + reg
                    const int w = 2 * (b & 1) + (i >> 3), rb = (i >> 2) & 1, reg = i & 3;
                    t[i] = *(const f32x4_t *)(src + (th0 + 32 * (w >> 1) + 2 * (4 * q + reg) + rb) * ldb + tf0 + 64 * (w & 1) + 4 * l);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) img[(b * 16 + i) * 64 + lane] = t[i];
            }
        }
        __syncthreads();
        return;
    }
    const int h0 = (tile % a.tiles_m) * TM + 32 * (wv >> 1), f0 = (tile / a.tiles_m) * TN + 64 * (wv & 1);
    // accumulator [rb][cb] register reg = element (h0 + 2 (4 q + reg) + rb, f0 + 4 l + cb)
    const uint32_t oa = (uint32_t)((q * lda + h0 + 2 * l) * 4), ob = (uint32_t)((q * ldb + f0 + 4 * l) * 4);
    const uint32_t sa = (uint32_t)(4 * lda * 4), sb = (uint32_t)(4 * ldb * 4);
    const uint32_t passes = (uint32_t)((a.m >> 2) / D - 1);   // ring passes after the first (m / 4 k-quads, a multiple of D)
    f32x4_t c00 = {0.f, 0.f, 0.f, 0.f}, c01 = c00, c02 = c00, c03 = c00, c10 = c00, c11 = c00, c12 = c00, c13 = c00;
#define WGRAD_ASM(NAME)                                                                                                              \
    asm volatile(NAME##_LOOP                                                                                                         \
                 : [c00] "+v"(c00), [c01] "+v"(c01), [c02] "+v"(c02), [c03] "+v"(c03), [c10] "+v"(c10), [c11] "+v"(c11),             \
                   [c12] "+v"(c12), [c13] "+v"(c13)                                                                                  \
                 : [oa] "v"(oa), [ob] "v"(ob), [pa] "s"(a.p.dy), [pb] "s"(a.p.x), [sa] "s"(sa), [sb] "s"(sb), [n] "s"(passes)        \
                 : NAME##_CLOBBERS)
    uint64_t t0 = 0, r0 = 0, t1 = 0, r1 = 0;
    if constexpr (CLK) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) : : "memory");
    if constexpr (VARIANT == 1) WGRAD_ASM(WGRAD_Q16_RING8_NOLOAD);
    else if constexpr (VARIANT == 2) WGRAD_ASM(WGRAD_Q16_RING8_PREFILL);
    else WGRAD_ASM(WGRAD_Q16_RING8);
#undef WGRAD_ASM
    if constexpr (CLK) {
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) : : "memory");
        if (lane == 0 && clk != nullptr) { uint64_t *d = clk + ((size_t)bid * 4 + wv) * 4; d[0] = t0; d[1] = t1; d[2] = r0; d[3] = r1; }
    }
    __syncthreads();                         // the fetch wave's image is complete
    Hyper hy{};
    if (a.hyper != nullptr) hy = Hyper{a.hyper[0], a.hyper[1], a.hyper[2], a.hyper[3], a.hyper[4]};
    const f32x4_t *acc[2][4] = {{&c00, &c01, &c02, &c03}, {&c10, &c11, &c12, &c13}};
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int o = (h0 + 2 * (4 * q + reg) + rb) * ldb + f0 + 4 * l;
            const f32x4_t g = {(*acc[rb][0])[reg], (*acc[rb][1])[reg], (*acc[rb][2])[reg], (*acc[rb][3])[reg]};
            if (a.p.grad != nullptr) *(f32x4_t *)(a.p.grad + o) = g;
            if (a.p.W != nullptr) {
                f32x4_t p = img[(wv * 8 + rb * 4 + reg) * 64 + lane], v = img[(32 + wv * 8 + rb * 4 + reg) * 64 + lane];
#pragma unroll
                for (int e = 0; e < 4; ++e) { float pe = p[e], ve = v[e]; rms_update(g[e], pe, ve, hy); p[e] = pe; v[e] = ve; }
                *(f32x4_t *)(a.p.V + o) = v; *(f32x4_t *)(a.p.W + o) = p;
                if (a.p.Wh != nullptr) {
                    constexpr float ps = (float)(1 << idl_planes::W_EXP);
                    uint2 h, l2;
                    if (idl_planes::split4(p[0] * ps, p[1] * ps, p[2] * ps, p[3] * ps, h, l2)) *a.p.over = 1;
                    *(uint2 *)(a.p.Wh + o) = h; *(uint2 *)(a.p.Wl + o) = l2;
                }
            }
        }
}

}  // namespace wg_dev
