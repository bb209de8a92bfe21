// wgrad.hip -- weight gradient of a Linear layer on the fp32 matrix cores with the RMSprop update in the epilogue (opt-in).
//
// Reference: loss.backward() produces Linear(F,512).weight.grad = dy^T x (torch autograd, idelucs/models.py:131) and
// optimizer.step() (models.py:132, torch.optim.RMSprop(lr, weight_decay=0.01), models.py:88) then reads that 8 MB gradient
// back together with the parameter and its square_avg.  Here the gradient tile never leaves the chip:
//   g = dy^T x + wd W;  v = alpha v + (1 - alpha) g^2;  W -= lr g / (sqrt(v) + eps)
// is applied to the 128 x 64 tile a workgroup has just accumulated (v_mfma_f32_32x32x2_f32: exact f32 products, f32
// accumulate -- the arithmetic of the hipBLASLt kernel it replaces).
//
// Both operands are read the way they sit in memory: dy [m, n_out] and x [m, n_in] are row-major with the contraction index m
// as the ROW, which is exactly the 32x32x2 operand map (lane l: A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31]) -- no
// transposes, no LDS image, no barrier in the main loop.
//
// Measured on MI355X at m = 1024, n_out = 512, n_in = 4096 (tools/bench_wgrad.py, back to back in a HIP graph): gradient only
// 40.2 us, with the fused update 43.3 us; the tuned hipBLASLt kernel does the bare product in 32-37 us, and the optimizer launch
// that would remain costs about what the fusion saves, so fused.py keeps hipBLASLt + idl_rmsprop_step by default
// (IDELUCS_WGRAD_FUSED=1 switches this kernel in).  What costs: with the main loop's loads removed the launch still takes 35.8 us
// against 27.3 us of MFMA issue -- accumulator set-up, the cross-wave reduction and the 16 MB parameter write are a serial tail.
#include "common.h"

namespace {

typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 64;

struct WgProblem {
    const float *dy, *x;      // [m, n_out], [m, n_in]
    float *grad;              // [n_out, n_in] or NULL
    float *W, *V;             // parameter and square_avg, or NULL (then grad must be given)
    int n_out, n_in;
};

struct WgArgs {
    WgProblem p;
    const float *hyper;       // [lr, alpha, eps, weight_decay, 1 - alpha]
    int m, tiles_m, tiles;
};

struct Hyper { float lr, alpha, eps, wd, oma; };

__device__ __forceinline__ void rms_update(float g, float &p, float &v, const Hyper &h)
{
    const float gi = g + h.wd * p;                   // grad.add(param, alpha=weight_decay)
    v = v * h.alpha + h.oma * gi * gi;               // square_avg.mul_(alpha).addcmul_(g, g, value=1-alpha)
    p = p - h.lr * (gi / (sqrtf(v) + h.eps));        // param.addcdiv_(grad, sqrt(v)+eps, value=-lr)
}

// ---------------------------------------------------------------- split-K over the four waves of a workgroup
// Every wave accumulates the WHOLE 128 x 64 tile of its workgroup over its own quarter of the contraction index: one 16-byte
// load gives a lane the A operands of four row blocks (rows h0 + 4 i + r), one 8-byte load the B operands of two column
// blocks (columns f0 + 2 j + c) -- 24 bytes per lane feed eight MFMAs (512 matrix-pipe cycles), the operand traffic of an
// LDS-tiled kernel without its barriers.  The loads are whole 128-byte lines (rows k, k + 1 of dy and x ARE the operand
// map).  D k-steps of loads are in flight per wave.  The four partial tiles meet once, in LDS, where the epilogue adds them in
// a fixed order and applies RMSprop with 16-byte accesses to W and square_avg.
constexpr int SK_D = 8;
constexpr int SK_PITCH = BN + 4;                       // floats; 128 rows x 68 x 4 B = 34 816 B per wave
constexpr int SK_LDS_BYTES = 4 * BM * SK_PITCH * 4;    // 139 264 B

template <int D>
__global__ __launch_bounds__(256) void wgrad_splitk_kernel(WgArgs a)
{
    extern __shared__ float lds[];
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = a.m;
    int tile = blockIdx.x;
    if ((a.tiles & 7) == 0) tile = (blockIdx.x & 7) * (a.tiles >> 3) + (blockIdx.x >> 3);
    const int h0 = (tile % a.tiles_m) * BM, f0 = (tile / a.tiles_m) * BN;
    const int lda = a.p.n_out, ldb = a.p.n_in;
    // ---- the epilogue's W / square_avg tile is requested first: it arrives while the products run
    const int c4 = tid & 15, rr = tid >> 4;                  // epilogue map: 16 float4 per tile row, 16 rows per pass
    f32x4_t pw[8], pv[8];
    if (a.p.W != nullptr) {
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int o = (h0 + rr + 16 * ps) * ldb + f0 + 4 * c4;
            pw[ps] = *(const f32x4_t *)(a.p.W + o); pv[ps] = *(const f32x4_t *)(a.p.V + o);
        }
    }
    // a k-step = rows 2 s, 2 s + 1 of dy and x; m is a multiple of 8 D (checked by the launcher), so every wave owns per = m / 8
    // steps, a whole number of rings.  Waves 0, 1 walk their share upwards and waves 2, 3 downwards: the D steps a ring reads
    // past the end of a share then always fall into a neighbour's share -- valid memory, no clamps in the loop.
    const int per = m >> 3;
    const int first = wv < 2 ? wv * per : (wv + 1) * per - 1;
    const int64_t sa = (wv < 2 ? 2 : -2) * (int64_t)lda, sb = (wv < 2 ? 2 : -2) * (int64_t)ldb;      // wave-uniform strides
    const float *ra = a.p.dy + (int64_t)(2 * first) * lda + h0 + (lane >> 5) * lda + 4 * (lane & 31);
    const float *rb = a.p.x + (int64_t)(2 * first) * ldb + f0 + (lane >> 5) * ldb + 2 * (lane & 31);
    f32x16_t acc[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.f;
#define SK_STEP(ta, tb)                                                                          \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                              \
        acc[r][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[r], tb[0], acc[r][0], 0, 0, 0);      \
        acc[r][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[r], tb[1], acc[r][1], 0, 0, 0);      \
    }
    f32x4_t av[D]; f32x2_t bv[D];
#pragma unroll
    for (int u = 0; u < D; ++u) {
        av[u] = *(const f32x4_t *)ra; bv[u] = *(const f32x2_t *)rb;
        ra += sa; rb += sb;
        asm volatile("" ::: "memory");                       // same issue order as the loop: its vmcnt waits then allow 2 (D - 1) loads in flight
    }
    for (int base = 0; base < per; base += D) {              // slot u holds step base + u, refilled with step base + D + u
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const f32x4_t ta = av[u]; const f32x2_t tb = bv[u];
            SK_STEP(ta, tb)
            av[u] = *(const f32x4_t *)ra; bv[u] = *(const f32x2_t *)rb;
            ra += sa; rb += sb;
        }
#pragma unroll
        for (int u = 0; u < D; ++u) {                        // keep each refill behind its own step's MFMAs (a true ring)
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        }
    }
#undef SK_STEP
    // ---- the four partial tiles -> LDS.  C/D: col j = lane & 31, row i = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5);
    //      element (h0 + 4 i + r, f0 + 2 j + c)
    {
        float *mine = lds + wv * BM * SK_PITCH;
        const int j = lane & 31, ih = 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = (e & 3) + 8 * (e >> 2) + ih;
                f32x2_t v2 = {acc[r][0][e], acc[r][1][e]};
                *(f32x2_t *)(mine + (4 * i + r) * SK_PITCH + 2 * j) = v2;
            }
    }
    __syncthreads();
    Hyper hy{};
    if (a.hyper != nullptr) hy = Hyper{a.hyper[0], a.hyper[1], a.hyper[2], a.hyper[3], a.hyper[4]};
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
        const int row = rr + 16 * ps;
        const int o = (h0 + row) * ldb + f0 + 4 * c4;
        const float *q = lds + row * SK_PITCH + 4 * c4;
        const f32x4_t p0 = *(const f32x4_t *)q, p1 = *(const f32x4_t *)(q + BM * SK_PITCH),
                      p2 = *(const f32x4_t *)(q + 2 * BM * SK_PITCH), p3 = *(const f32x4_t *)(q + 3 * BM * SK_PITCH);
        const f32x4_t g = (p0 + p1) + (p2 + p3);
        if (a.p.grad != nullptr) *(f32x4_t *)(a.p.grad + o) = g;
        if (a.p.W != nullptr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { float p = pw[ps][e], v = pv[ps][e]; rms_update(g[e], p, v, hy); pw[ps][e] = p; pv[ps][e] = v; }
            *(f32x4_t *)(a.p.V + o) = pv[ps]; *(f32x4_t *)(a.p.W + o) = pw[ps];
        }
    }
}

}  // namespace

extern "C" {

int idl_wgrad_supported(int m, int n_out, int n_in)
{
    return (m >= 8 * SK_D && (m % (8 * SK_D)) == 0 && n_out >= BM && (n_out % BM) == 0 && n_in >= BN && (n_in % BN) == 0) ? 1 : 0;
}

int idl_wgrad_rmsprop(const float *dy, const float *x, int m, int n_out, int n_in, float *grad, float *W, float *square_avg,
                      const float *hyper, void *stream)
{
    IDL_REQUIRE(dy && x && idl_wgrad_supported(m, n_out, n_in), "wgrad: m % 64 == 0, n_out % 128 == 0, n_in % 64 == 0");
    IDL_REQUIRE((W != nullptr) == (square_avg != nullptr) && (W != nullptr || grad != nullptr), "wgrad: give W and square_avg (fused update) and/or grad");
    IDL_REQUIRE(W == nullptr || hyper != nullptr, "wgrad: hyper is needed for the fused update");
    IDL_REQUIRE((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)grad | (uintptr_t)W | (uintptr_t)square_avg) & 15u) == 0, "wgrad: buffers must be 16-byte aligned");
    IDL_REQUIRE((int64_t)m * n_in < (1ll << 31) && (int64_t)n_out * n_in < (1ll << 31), "wgrad: operands beyond 2^31 elements");
    WgArgs a{};
    a.p = WgProblem{dy, x, grad, W, square_avg, n_out, n_in};
    a.hyper = hyper; a.m = m;
    a.tiles_m = n_out / BM;
    a.tiles = a.tiles_m * (n_in / BN);
    static bool attr_set[64] = {};
    int dev = 0;
    IDL_HIP_TRY(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        IDL_HIP_TRY(hipFuncSetAttribute((const void *)wgrad_splitk_kernel<SK_D>, hipFuncAttributeMaxDynamicSharedMemorySize, SK_LDS_BYTES));
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((wgrad_splitk_kernel<SK_D>), dim3((unsigned)a.tiles), dim3(256), SK_LDS_BYTES, (hipStream_t)stream, a);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

}  // extern "C"
