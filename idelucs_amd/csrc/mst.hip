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
// boundary is the cheapest grid barrier on this GPU.  The pass reads 17 B per point outside the tree, and the
// point's coordinates only where its entry can still change.
#include <string.h>

#include "common.h"

namespace {

struct Cand { double w; int64_t j, src; };

struct PrimArgs {
    const void *xt;            // [D][N] float32 or float64
    const double *core;        // [N]
    int64_t n; int d;
    double *min_reach;         // [N], +inf at start
    int64_t *source;           // [N]
    uint8_t *in_tree;          // [N], zero at start
    Cand *cand[2];             // per-workgroup candidates of the even / odd steps, [gridDim.x] each
    int64_t *mst_cur, *mst_next; double *mst_w;     // [N-1] edges in the order they were added
};

constexpr int PRIM_NT = 256;
constexpr int PRIM_MAX_D = 256;

__device__ __forceinline__ bool better(double w, int64_t j, double bw, int64_t bj) { return w < bw || (w == bw && j < bj); }

template <typename T>
__global__ __launch_bounds__(PRIM_NT) void prim_step_kernel(PrimArgs a, int64_t step, int scan, int n_part)
{
    __shared__ double sw[PRIM_NT];
    __shared__ int64_t sj[PRIM_NT], ss[PRIM_NT];
    __shared__ double xc[PRIM_MAX_D];
    const int tid = threadIdx.x;
    // ---- which node did the previous step add?  (reduce its per-workgroup candidates; every workgroup for itself)
    int64_t cur = 0;
    if (step > 0) {
        const Cand *pc = a.cand[(step - 1) & 1];
        double bw = __builtin_inf(); int64_t bj = INT64_MAX, bs = 0;
        for (int g = tid; g < n_part; g += PRIM_NT) {
            const Cand c = pc[g];
            if (better(c.w, c.j, bw, bj)) { bw = c.w; bj = c.j; bs = c.src; }
        }
        sw[tid] = bw; sj[tid] = bj; ss[tid] = bs;
        __syncthreads();
        for (int s = PRIM_NT / 2; s > 0; s >>= 1) {
            if (tid < s && better(sw[tid + s], sj[tid + s], sw[tid], sj[tid])) { sw[tid] = sw[tid + s]; sj[tid] = sj[tid + s]; ss[tid] = ss[tid + s]; }
            __syncthreads();
        }
        cur = sj[0];
        if (blockIdx.x == 0 && tid == 0) { a.mst_cur[step - 1] = ss[0]; a.mst_next[step - 1] = cur; a.mst_w[step - 1] = sw[0]; }
        __syncthreads();
    }
    if (!scan) return;                      // (the launch after the last step only records its edge)
    if (blockIdx.x == 0 && tid == 0) a.in_tree[cur] = 1;
    const T *xt = (const T *)a.xt;
    const int64_t n = a.n;
    const int d = a.d;
    for (int k = tid; k < d; k += PRIM_NT) xc[k] = (double)xt[(int64_t)k * n + cur];
    __syncthreads();
    const double cc = a.core[cur];
    double bw = __builtin_inf(); int64_t bj = INT64_MAX, bs = 0;
    for (int64_t j = (int64_t)blockIdx.x * PRIM_NT + tid; j < n; j += (int64_t)gridDim.x * PRIM_NT) {
        if (a.in_tree[j] || j == cur) continue;
        const double cj = a.core[j];
        double mr = a.min_reach[j];
        int64_t src = a.source[j];
        // mrd >= max(core[cur], core[j]) whatever the distance is: when that bound already reaches min_reach[j] nothing can change,
        // and the point itself (256 of the ~280 bytes this pass would read for it) is not touched.  Most points sit at their floor
        // min_reach[j] == core[j] after a few visits, so the pass reads 17 bytes per point instead of 280.
        if (fmax(cc, cj) < mr) {
            double acc = 0.0;
#pragma unroll 8
            for (int k = 0; k < d; ++k) {
                const double t = xc[k] - (double)xt[(int64_t)k * n + j];
                acc = __dadd_rn(acc, __dmul_rn(t, t));          // no contraction: sklearn's loop is mul then add
            }
            const double mrd = fmax(fmax(cc, cj), __dsqrt_rn(acc));
            if (mrd < mr) { mr = mrd; src = cur; a.min_reach[j] = mr; a.source[j] = src; }
        }
        if (better(mr, j, bw, bj)) { bw = mr; bj = j; bs = src; }
    }
    sw[tid] = bw; sj[tid] = bj; ss[tid] = bs;
    __syncthreads();
    for (int s = PRIM_NT / 2; s > 0; s >>= 1) {
        if (tid < s && better(sw[tid + s], sj[tid + s], sw[tid], sj[tid])) { sw[tid] = sw[tid + s]; sj[tid] = sj[tid + s]; ss[tid] = ss[tid + s]; }
        __syncthreads();
    }
    if (tid == 0) a.cand[step & 1][blockIdx.x] = Cand{sw[0], sj[0], ss[0]};
}

__global__ void prim_init_kernel(double *min_reach, int64_t *source, uint8_t *in_tree, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        min_reach[i] = __builtin_inf(); source[i] = 1; in_tree[i] = 0;
    }
}

inline int prim_grid(int64_t n)
{
    int64_t g = (n + PRIM_NT - 1) / PRIM_NT;
    if (g > 1024) g = 1024;
    return (int)(g < 1 ? 1 : g);
}

inline int64_t align256(int64_t b) { return (b + 255) & ~(int64_t)255; }

}  // namespace

extern "C" {

int64_t idl_mst_prim_workspace(int64_t n)
{
    if (n < 1) return 256;
    return align256(n * 8) + align256(n * 8) + align256(n) + 2 * align256((int64_t)prim_grid(n) * (int64_t)sizeof(Cand)) + 256;
}

int idl_mst_prim(const void *xt, int is_f64, const double *core, int64_t n, int d, int64_t *mst_cur, int64_t *mst_next, double *mst_w,
                 void *workspace, void *stream)
{
    IDL_REQUIRE(xt && core && mst_cur && mst_next && mst_w && workspace, "mst_prim: NULL buffer");
    IDL_REQUIRE(n >= 2 && d >= 1 && d <= PRIM_MAX_D, "mst_prim: need n >= 2 points of 1..256 features");
    IDL_REQUIRE((((uintptr_t)workspace) & 255u) == 0, "mst_prim: workspace must be 256-byte aligned");
    unsigned char *w = (unsigned char *)workspace;
    PrimArgs a{};
    a.xt = xt; a.core = core; a.n = n; a.d = d;
    a.min_reach = (double *)w; w += align256(n * 8);
    a.source = (int64_t *)w; w += align256(n * 8);
    a.in_tree = (uint8_t *)w; w += align256(n);
    const int g = prim_grid(n);
    a.cand[0] = (Cand *)w; w += align256((int64_t)g * (int64_t)sizeof(Cand));
    a.cand[1] = (Cand *)w;
    a.mst_cur = mst_cur; a.mst_next = mst_next; a.mst_w = mst_w;
    const hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(prim_init_kernel, dim3(256), dim3(256), 0, st, a.min_reach, a.source, a.in_tree, n);
    for (int64_t step = 0; step < n; ++step) {              // step n - 1 + 1: the launch that only records the last edge
        const int scan = step < n - 1 ? 1 : 0;
        if (is_f64) hipLaunchKernelGGL(prim_step_kernel<double>, dim3(scan ? g : 1), dim3(PRIM_NT), 0, st, a, step, scan, g);
        else hipLaunchKernelGGL(prim_step_kernel<float>, dim3(scan ? g : 1), dim3(PRIM_NT), 0, st, a, step, scan, g);
    }
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

}  // extern "C"
