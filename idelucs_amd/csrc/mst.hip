// mst.hip -- Prim's minimum spanning tree of the mutual-reachability graph on gfx950: the O(N^2 D) part of HDBSCAN, which the
// n_clusters = 0 mode of the reference runs on the last voter's latent (idelucs/__main__.py:153-156: hdbscan.HDBSCAN(
// min_cluster_size = N // 100 + 1)).  `hdbscan` is absent here; sklearn.cluster.HDBSCAN is the stand-in, and this
// restates ITS tree construction -- sklearn/cluster/_hdbscan/_linkage.pyx:111-224 (mst_from_data_matrix), v1.7.2:
//     current = 0; repeat N-1 times: mark current in the tree; for every j outside the tree (ascending):
//         mrd = max(core[current], core[j], ||x_current - x_j||);  if mrd < min_reach[j]: min_reach[j] = mrd, source[j] = current
//         candidate = the j with the smallest min_reach[j] (strict <, so the FIRST such j), with its source
//     edge (source, j, min_reach[j]); current = j
// -- so that the edges, and with them sklearn's own condensed-tree code that runs on them afterwards, are the ones sklearn gets.
// Distances are formed exactly as sklearn's EuclideanDistance64 does (float64: t = a - b; d += t * t, in feature order, then one
// sqrt; no fused multiply-add), from the points stored feature-major ([D][N]: lanes read consecutive points) in float32 when
// the data is float32-exact (the latent is) or float64 otherwise.
//
// One launch per tree node: the scan over the N points is one grid-wide pass (every workgroup leaves its best candidate), and the
// NEXT launch starts by reducing those candidates -- every workgroup for itself -- to learn which node was added.  A launch
// boundary is the cheapest grid barrier on this GPU.  A step is a chain of dependent memory round trips (~1.2 us
// each), so the chain is kept short: a point's state is 16 bytes (min_reach, with "in the tree" folded in as -1, and its core
// distance), every thread has the state of its first four points in flight BEFORE it learns which node was added, the source
// of an edge is looked up only for the one edge that is recorded, and the point's coordinates are touched only where its entry
// can still change (mrd >= max(core[cur], core[j]) whatever the distance is).
#include <string.h>

#include "common.h"
#include "wave_ops.h"

namespace {

struct Cand { double w; int64_t j; };

struct PrimArgs {
    const void *xt;            // [D][N] float32 or float64
    const double *core;        // [N]
    int64_t n; int d;
    double *min_reach;         // [N], +inf at start; -1 once the point is in the tree
    int64_t *source;           // [N]
    Cand *cand[2];             // per-workgroup candidates of the even / odd steps, [gridDim.x] each
    int64_t *mst_cur, *mst_next; double *mst_w;     // [N-1] edges in the order they were added
    // the 8-bit filter (FILTER kernels): y_j = offset + scale * code_j is a point near x_j, so
    //     ||x_cur - x_j|| >= scale * sqrt(sum_k (code_cur,k - code_j,k)^2) - resid[cur] - resid[j]
    // and a pair whose lower bound cannot undercut min_reach[j] needs no exact distance: 72 bytes per point instead of 4 d
    const uint32_t *codes;     // [D/4][N]: four consecutive features of a point per word
    const uint32_t *qq;        // [N] sum_k code^2
    const float *resid;        // [N] >= ||x_j - y_j||
    double scale;
};

constexpr int PRIM_NT = 256;
constexpr int PRIM_MAX_D = 256;
constexpr int PRIM_AHEAD = 4;          // points per thread whose state is loaded before the added node is known

__device__ __forceinline__ bool better(double w, int64_t j, double bw, int64_t bj) { return w < bw || (w == bw && j < bj); }

// the best (w, j) of the workgroup, in every thread
__device__ __forceinline__ void block_best(double &bw, int64_t &bj, double *sw, int64_t *sj)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double ow = __shfl_xor(bw, o, 64);
        const int64_t oj = __shfl_xor(bj, o, 64);
        if (better(ow, oj, bw, bj)) { bw = ow; bj = oj; }
    }
    const int tid = threadIdx.x;
    if ((tid & 63) == 0) { sw[tid >> 6] = bw; sj[tid >> 6] = bj; }
    __syncthreads();
    bw = sw[0]; bj = sj[0];
#pragma unroll
    for (int w = 1; w < PRIM_NT / 64; ++w) if (better(sw[w], sj[w], bw, bj)) { bw = sw[w]; bj = sj[w]; }
    __syncthreads();
}

template <typename T, bool FILTER>
__global__ __launch_bounds__(PRIM_NT) void prim_step_kernel(PrimArgs a, int64_t step, int scan, int n_part)
{
    __shared__ double sw[PRIM_NT / 64];
    __shared__ int64_t sj[PRIM_NT / 64];
    __shared__ double xc[PRIM_MAX_D];
    __shared__ uint32_t qc[PRIM_MAX_D / 4];
    const int tid = threadIdx.x;
    const int64_t n = a.n;
    const int64_t stride = (int64_t)gridDim.x * PRIM_NT;
    const int64_t j0 = (int64_t)blockIdx.x * PRIM_NT + tid;
    // ---- the state of this thread's first points: in flight while the added node is being found
    double mr_a[PRIM_AHEAD], cj_a[PRIM_AHEAD];
    if (scan) {
#pragma unroll
        for (int i = 0; i < PRIM_AHEAD; ++i) {
            const int64_t j = j0 + i * stride;
            mr_a[i] = j < n ? a.min_reach[j] : -1.0;
            cj_a[i] = j < n ? a.core[j] : 0.0;
        }
    }
    // ---- which node did the previous step add?  (reduce its per-workgroup candidates; every workgroup for itself)
    int64_t cur = 0;
    double cur_w = 0.0;
    if (step > 0) {
        const Cand *pc = a.cand[(step - 1) & 1];
        double bw = __builtin_inf(); int64_t bj = INT64_MAX;
        for (int g = tid; g < n_part; g += PRIM_NT) {
            const Cand c = pc[g];
            if (better(c.w, c.j, bw, bj)) { bw = c.w; bj = c.j; }
        }
        block_best(bw, bj, sw, sj);
        cur = bj; cur_w = bw;
    }
    int64_t cur_src = 0;
    const bool recorder = blockIdx.x == 0 && tid == 0 && step > 0;
    if (recorder) cur_src = a.source[cur];                  // (needed only when the edge is written, at the end)
    if (scan) {
        if (blockIdx.x == 0 && tid == 0) a.min_reach[cur] = -1.0;            // in the tree (this launch skips it by number)
        const T *xt = (const T *)a.xt;
        const int d = a.d;
        const int d4 = d >> 2;
        for (int k = tid; k < d; k += PRIM_NT) xc[k] = (double)xt[(int64_t)k * n + cur];
        if (FILTER) for (int k = tid; k < d4; k += PRIM_NT) qc[k] = a.codes[(int64_t)k * n + cur];
        const double cc = a.core[cur];
        double slack = 0.0; uint32_t qq_c = 0;
        if (FILTER) { slack = (double)a.resid[cur]; qq_c = a.qq[cur]; }
        __syncthreads();
        double bw = __builtin_inf(); int64_t bj = INT64_MAX;
        // true when the pair (cur, j) cannot lower min_reach[j] = mr: its distance is at least `lb`
        auto filtered_out = [&](int64_t j, double mr, double floor_cj) -> bool {
            uint32_t dot = 0;
#pragma unroll 16
            for (int k = 0; k < d4; ++k) dot = __builtin_amdgcn_udot4(qc[k], a.codes[(int64_t)k * n + j], dot, false);
            const int64_t i2 = (int64_t)qq_c + (int64_t)a.qq[j] - 2 * (int64_t)dot;                 // exact: sum_k (code_cur - code_j)^2
            const double lb = a.scale * __dsqrt_rn((double)(i2 > 0 ? i2 : 0)) * (1.0 - 1e-12) - slack - (double)a.resid[j];
            return fmax(floor_cj, lb) >= mr;
        };
        auto visit = [&](int64_t j, double mr, double cj) {
            if (mr < 0.0 || j == cur) return;
            const double floor_cj = fmax(cc, cj);
            if (floor_cj < mr && !(FILTER && filtered_out(j, mr, floor_cj))) {
                double acc = 0.0;
#pragma unroll 8
                for (int k = 0; k < d; ++k) {
                    const double t = xc[k] - (double)xt[(int64_t)k * n + j];
                    acc = idl_dev::square_then_add(acc, t);                  // no contraction: sklearn's loop is mul then add
                }
                const double mrd = fmax(floor_cj, __dsqrt_rn(acc));
                if (mrd < mr) { mr = mrd; a.min_reach[j] = mr; a.source[j] = cur; }
            }
            if (better(mr, j, bw, bj)) { bw = mr; bj = j; }
        };
#pragma unroll
        for (int i = 0; i < PRIM_AHEAD; ++i) visit(j0 + i * stride, mr_a[i], cj_a[i]);
        for (int64_t j = j0 + PRIM_AHEAD * stride; j < n; j += stride) visit(j, a.min_reach[j], a.core[j]);
        block_best(bw, bj, sw, sj);
        if (tid == 0) a.cand[step & 1][blockIdx.x] = Cand{bw, bj};
    }
    if (recorder) { a.mst_cur[step - 1] = cur_src; a.mst_next[step - 1] = cur; a.mst_w[step - 1] = cur_w; }
}

__global__ void prim_init_kernel(double *min_reach, int64_t *source, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        min_reach[i] = __builtin_inf(); source[i] = 1;
    }
}

inline int prim_grid(int64_t n)
{
    int64_t g = (n + PRIM_NT * PRIM_AHEAD - 1) / (PRIM_NT * PRIM_AHEAD);     // every thread's points fit the look-ahead up to 10^6 points
    if (g > 1024) g = 1024;
    return (int)(g < 1 ? 1 : g);
}

inline int64_t align256(int64_t b) { return (b + 255) & ~(int64_t)255; }

}  // namespace

static int prim_run(const void *xt, int is_f64, const double *core, int64_t n, int d, int64_t *mst_cur, int64_t *mst_next, double *mst_w,
                    void *workspace, const uint32_t *codes, const uint32_t *qq, const float *resid, double scale, void *stream)
{
    IDL_REQUIRE(xt && core && mst_cur && mst_next && mst_w && workspace, "mst_prim: NULL buffer");
    IDL_REQUIRE(n >= 2 && d >= 1 && d <= PRIM_MAX_D, "mst_prim: need n >= 2 points of 1..256 features");
    IDL_REQUIRE((((uintptr_t)workspace) & 255u) == 0, "mst_prim: workspace must be 256-byte aligned");
    const bool filter = codes != nullptr;
    if (filter) {
        IDL_REQUIRE(qq && resid && scale > 0.0, "mst_prim_q8: NULL filter buffer or non-positive scale");
        IDL_REQUIRE(d % 4 == 0, "mst_prim_q8: the number of features must be a multiple of 4");
    }
    unsigned char *w = (unsigned char *)workspace;
    PrimArgs a{};
    a.xt = xt; a.core = core; a.n = n; a.d = d;
    a.min_reach = (double *)w; w += align256(n * 8);
    a.source = (int64_t *)w; w += align256(n * 8);
    w += align256(n);                                        // (unused since the tree mark moved into min_reach)
    const int g = prim_grid(n);
    a.cand[0] = (Cand *)w; w += align256((int64_t)g * (int64_t)sizeof(Cand));
    a.cand[1] = (Cand *)w;
    a.mst_cur = mst_cur; a.mst_next = mst_next; a.mst_w = mst_w;
    a.codes = codes; a.qq = qq; a.resid = resid; a.scale = scale;
    const hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(prim_init_kernel, dim3(256), dim3(256), 0, st, a.min_reach, a.source, n);
    void (*kern)(PrimArgs, int64_t, int, int) =
        is_f64 ? (filter ? prim_step_kernel<double, true> : prim_step_kernel<double, false>)
               : (filter ? prim_step_kernel<float, true> : prim_step_kernel<float, false>);
    for (int64_t step = 0; step < n; ++step) {              // step n - 1 + 1: the launch that only records the last edge
        const int scan = step < n - 1 ? 1 : 0;
        hipLaunchKernelGGL(kern, dim3(scan ? g : 1), dim3(PRIM_NT), 0, st, a, step, scan, g);
    }
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

extern "C" {

int64_t idl_mst_prim_workspace(int64_t n)
{
    if (n < 1) return 256;
    return align256(n * 8) + align256(n * 8) + align256(n) + 2 * align256((int64_t)prim_grid(n) * (int64_t)sizeof(Cand)) + 256;
}

int idl_mst_prim(const void *xt, int is_f64, const double *core, int64_t n, int d, int64_t *mst_cur, int64_t *mst_next, double *mst_w,
                 void *workspace, void *stream)
{
    return prim_run(xt, is_f64, core, n, d, mst_cur, mst_next, mst_w, workspace, nullptr, nullptr, nullptr, 0.0, stream);
}

int idl_mst_prim_q8(const void *xt, int is_f64, const double *core, int64_t n, int d, const uint32_t *codes, const uint32_t *qq,
                    const float *resid, double scale, int64_t *mst_cur, int64_t *mst_next, double *mst_w, void *workspace, void *stream)
{
    IDL_REQUIRE(codes, "mst_prim_q8: NULL codes");
    return prim_run(xt, is_f64, core, n, d, mst_cur, mst_next, mst_w, workspace, codes, qq, resid, scale, stream);
}

}  // extern "C"
