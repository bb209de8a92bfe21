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
// boundary is the cheapest grid barrier on this GPU (a last-workgroup-to-arrive reduction was measured here: the
// device-scope release it needs writes the L2 back, 40 us a step; with written-through atomics instead, 22 us against 13).
// A step is a chain of dependent memory round trips (~1.2 us each) and little else, so the chain is kept short and narrow:
//   - a point's state is 16 bytes (min_reach, with "in the tree" folded in as -1, and its core distance); the coordinates are
//     touched only where the entry can still change (mrd >= max(core[cur], core[j]) whatever the distance is);
//   - everything that does not depend on the added node -- the state of a thread's points, their codes -- is in flight before
//     the candidates are reduced; the source of an edge is looked up only for the one edge that is recorded;
//   - the points that do need an exact distance (one in six on a latent of tight clusters) are queued in LDS and shared out
//     over the workgroup's lanes: one pass of 64 loads per lane instead of up to four with most lanes idle;
//   - the exact distance issues its 64 loads before it uses the first.
//
// idl_mst_prim_local puts an 8-bit lower bound in front of the exact distances.  The caller orders the points so that neighbours
// in memory are neighbours in space, cuts that order into groups and codes every point inside its group's box:
// y_j = lo_g + scale_g * code_j, resid_j >= ||x_j - y_j||.  Then ||x_cur - x_j|| >= ||x_cur - y_j|| - resid_j, and the first
// term is 64 bytes of codes against (x_cur - lo_g) / scale_g, which a workgroup computes once per 256 points.  A latent of tight,
// far-apart clusters (BASELINE cfg5: clusters 0.3 wide, 140 apart) needs the codes to be LOCAL: whether a far point's entry can
// still drop is decided in the third digit of its distance.  A pair whose bound cannot undercut min_reach[j] skips the exact
// distance, which would have changed nothing -- the tree is the same, edge for edge.  Ties are broken on the points' ORIGINAL
// numbers (`orig`), as sklearn's scan over j would.
#include <string.h>

#include "common.h"
#include "wave_ops.h"

namespace {

struct Cand { double w; int64_t j, p; };       // weight, original number (ties), position in memory

struct PrimArgs {
    const void *xt;            // [D][N] float32 or float64, memory order
    const double *core;        // [N], memory order
    int64_t n; int d;
    double *min_reach;         // [N], +inf at start; -1 once the point is in the tree
    int64_t *source;           // [N] original number of the tree node that set min_reach
    Cand *cand[2];             // per-workgroup candidates of the even / odd steps, [gridDim.x] each
    int64_t *mst_cur, *mst_next; double *mst_w;     // [N-1] edges in the order they were added (original numbers)
    const int32_t *orig;       // [N] original number of the point at each position; NULL = the identity
    int64_t start;             // position of original point 0
    // the local 8-bit filter (FILTER kernels)
    const uint32_t *codes;     // [D/4][N]: four consecutive features of a point per word
    const float *resid;        // [N] >= ||x_j - y_j||
    const int32_t *gid;        // [N] group of the point at each position (non-decreasing)
    const float *glo;          // [G][D] the group's box corner
    const double *gscale;      // [G] its code step
};

constexpr int PRIM_NT = 256;           // threads of a workgroup (a multiple of PRIM_SUB: 1024 was measured, 16.5 us a step against 13.2)
constexpr int PRIM_SUB = 256;
constexpr int PRIM_MAX_D = 256;
constexpr int PRIM_FILTER_D = 64;       // features the filter's registers hold
constexpr int PRIM_AHEAD = 4;          // points per thread whose state is loaded before the added node is known
constexpr int PRIM_RUNS = (PRIM_NT / PRIM_SUB) * PRIM_AHEAD;      // runs of a workgroup

__device__ __forceinline__ bool better(double w, int64_t j, double bw, int64_t bj) { return w < bw || (w == bw && j < bj); }

// the best (w, j; p) of the workgroup, in every thread
__device__ __forceinline__ void block_best(double &bw, int64_t &bj, int64_t &bp, double *sw, int64_t *sj, int64_t *sp)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double ow = __shfl_xor(bw, o, 64);
        const int64_t oj = __shfl_xor(bj, o, 64), op = __shfl_xor(bp, o, 64);
        if (better(ow, oj, bw, bj)) { bw = ow; bj = oj; bp = op; }
    }
    const int tid = threadIdx.x;
    if ((tid & 63) == 0) { sw[tid >> 6] = bw; sj[tid >> 6] = bj; sp[tid >> 6] = bp; }
    __syncthreads();
    bw = sw[0]; bj = sj[0]; bp = sp[0];
#pragma unroll
    for (int w = 1; w < PRIM_NT / 64; ++w) if (better(sw[w], sj[w], bw, bj)) { bw = sw[w]; bj = sj[w]; bp = sp[w]; }
    __syncthreads();
}

// Diagnostic build (make STAMPS=1; tools/stamps_prim.py): workgroups 0, 1/3 and 2/3 of the grid add the time (s_memrealtime, 100 MHz)
// they spend between six marks of every step to prim_phase_sum; idl_debug_prim_phases reads and clears the sums.
#ifdef IDL_PHASE_STAMPS
__device__ unsigned long long prim_phase_sum[8];
#define PRIM_MARK(slot) do { if (stamping) { const uint64_t now_ = __builtin_amdgcn_s_memrealtime(); atomicAdd(&prim_phase_sum[slot], (unsigned long long)(now_ - last_)); last_ = now_; } } while (0)
#else
#define PRIM_MARK(slot) do { } while (0)
#endif

template <typename T, bool FILTER, bool D64>
__global__ __launch_bounds__(PRIM_NT, 1024 / PRIM_NT) void prim_step_kernel(PrimArgs a, int64_t step, int scan, int n_part)
{
    __shared__ double sw[PRIM_NT / 64];
    __shared__ int64_t sj[PRIM_NT / 64], sp[PRIM_NT / 64];
    __shared__ double xc[PRIM_MAX_D];
    __shared__ float up[PRIM_RUNS][PRIM_FILTER_D];    // (x_cur - lo_g) / scale_g for the group of each of the workgroup's runs
    __shared__ int q_n;                               // the exact distances still to do: owner, its min_reach (in: old, out: new), its floor
    __shared__ unsigned short q_item[PRIM_NT * PRIM_AHEAD];
    __shared__ double q_mr[PRIM_NT * PRIM_AHEAD], q_floor[PRIM_NT * PRIM_AHEAD];
    const int tid = threadIdx.x, sub = tid / PRIM_SUB;
#ifdef IDL_PHASE_STAMPS
    const bool stamping = tid == 0 && scan && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 3 || blockIdx.x == 2 * gridDim.x / 3);
    uint64_t last_ = __builtin_amdgcn_s_memrealtime();
#endif
    const int64_t n = a.n;
    const int d = a.d;
    const int64_t vblock = (int64_t)blockIdx.x * (PRIM_NT / PRIM_SUB) + sub;             // the sub-block's number among all
    const int64_t stride = (int64_t)gridDim.x * PRIM_NT;                               // between a thread's points
    const int64_t p0 = vblock * PRIM_SUB + (tid % PRIM_SUB);
    const T *xt = (const T *)a.xt;
    // ---- everything that does not depend on which node was added is put in flight first: the state of this thread's first points,
    // the group of each of its runs (the run's first point's), the points' own groups and -- once the state says they are outside
    // the tree -- their codes.  The candidates of the previous step are read meanwhile.
    double mr_a[PRIM_AHEAD], cj_a[PRIM_AHEAD];
    int64_t o_a[PRIM_AHEAD];
    bool in_run[PRIM_AHEAD];
    uint32_t cw[PRIM_AHEAD][PRIM_FILTER_D / 4];
    float rs[PRIM_AHEAD], run_scale[PRIM_AHEAD];
    int my_run_g = 0;                                        // group of the run this thread prepares a feature of (FILTER)
    if (scan) {
        int run_g[PRIM_AHEAD], g_own[PRIM_AHEAD];
#pragma unroll
        for (int i = 0; i < PRIM_AHEAD; ++i) {
            const int64_t p = p0 + i * stride;
            mr_a[i] = p < n ? a.min_reach[p] : -1.0;
            cj_a[i] = p < n ? a.core[p] : 0.0;
            o_a[i] = p < n ? (a.orig ? (int64_t)a.orig[p] : p) : 0;
            if (FILTER) {
                const int64_t first = vblock * PRIM_SUB + i * stride;
                run_g[i] = first < n ? a.gid[first] : 0;
                g_own[i] = p < n ? a.gid[p] : -1;
            }
        }
        if (FILTER) {
            static_assert(PRIM_RUNS * 64 == PRIM_NT && PRIM_FILTER_D == 64 && PRIM_AHEAD == 4, "one (run, feature) per thread");
            const int rr = tid >> 6;                         // run sub * 4 + i, feature tid & 63
            const int64_t first = ((int64_t)blockIdx.x * (PRIM_NT / PRIM_SUB) + (rr >> 2)) * PRIM_SUB + (rr & 3) * stride;
            my_run_g = first < n ? a.gid[first] : 0;
            const int d4 = d >> 2;
            const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.codes, 0, 0xffffffff, 0x00020000);
#pragma unroll
            for (int i = 0; i < PRIM_AHEAD; ++i) {
                const int64_t p = p0 + i * stride;
                in_run[i] = mr_a[i] >= 0.0 && g_own[i] == run_g[i];
                rs[i] = 0.f;
                run_scale[i] = (float)a.gscale[run_g[i]];
                if (in_run[i]) {
                    rs[i] = a.resid[p];
#pragma unroll
                    for (int k = 0; k < PRIM_FILTER_D / 4; ++k)
                        cw[i][k] = D64 ? __builtin_amdgcn_raw_buffer_load_b32(c_rsrc, (uint32_t)p * 4u, k * (int)n * 4, 0)
                                       : (k < d4 ? a.codes[(int64_t)k * n + p] : 0u);
                }
            }
        }
    }
    // ---- which node did the previous step add?  (reduce its per-workgroup candidates; every workgroup for itself)
    int64_t cur = a.start, cur_o = 0;
    double cur_w = 0.0;
    if (step > 0) {
        const Cand *pc = a.cand[(step - 1) & 1];
        double bw = __builtin_inf(); int64_t bj = INT64_MAX, bp = 0;
        for (int g = tid; g < n_part; g += PRIM_NT) {
            const Cand c = pc[g];
            if (better(c.w, c.j, bw, bj)) { bw = c.w; bj = c.j; bp = c.p; }
        }
        block_best(bw, bj, bp, sw, sj, sp);
        cur = bp; cur_o = bj; cur_w = bw;
    }
    PRIM_MARK(0);                                            // candidates read and reduced (the state and code loads are in flight)
    int64_t cur_src = 0;
    const bool recorder = blockIdx.x == 0 && tid == 0 && step > 0;
    if (recorder) cur_src = a.source[cur];                  // (needed only when the edge is written, at the end)
    if (scan) {
        if (blockIdx.x == 0 && tid == 0) a.min_reach[cur] = -1.0;            // in the tree (this launch skips it by position)
        if (tid == 0) q_n = 0;
        for (int k = tid; k < d; k += PRIM_NT) xc[k] = (double)xt[(int64_t)k * n + cur];
        const double cc = a.core[cur];
        if (FILTER) {
            const int k = tid & 63;
            const double sc = a.gscale[my_run_g];
            const float lo = k < d ? a.glo[(int64_t)my_run_g * d + k] : 0.f;
            __syncthreads();
            up[tid >> 6][k] = k < d ? (float)((xc[k] - (double)lo) / sc) : 0.f;          // (zero beyond d: those code words are zero too)
        }
        __syncthreads();
        PRIM_MARK(1);                                        // the added node's coordinates, the runs' boxes
        const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.xt, 0, 0xffffffff, 0x00020000);
        const int col_bytes = (int)n * 4;                    // (D64 kernels: n * 64 * 4 < 2^31, checked on the host)
        double bw = __builtin_inf(); int64_t bj = INT64_MAX, bp = 0;
        // sqrt(sum_k (x_cur,k - x_p,k)^2), product and sum each rounded, in feature order
        auto exact = [&](int64_t p, double &mr, double floor_cj) {
            double acc = 0.0;
            if (D64) {                                       // 64 float32 features (the latent): every load first, no tests in between;
                uint32_t v[64];                              // buffer loads: one descriptor + a scalar column offset, not 64 address pairs
#pragma unroll
                for (int k = 0; k < 64; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, (uint32_t)p * 4u, k * col_bytes, 0);
#pragma unroll
                for (int k = 0; k < 64; ++k) {
                    if ((k & 7) == 0) __builtin_amdgcn_sched_barrier(0);     // (or all 64 are widened to double up front: 128 registers)
                    const double t = xc[k] - (double)__uint_as_float(v[k]);
                    acc = idl_dev::square_then_add(acc, t);                  // no contraction: sklearn's loop is mul then add
                }
            } else {
#pragma unroll 8
                for (int k = 0; k < d; ++k) {
                    const double t = xc[k] - (double)xt[(int64_t)k * n + p];
                    acc = idl_dev::square_then_add(acc, t);
                }
            }
            const double mrd = fmax(floor_cj, __dsqrt_rn(acc));
            if (mrd < mr) { mr = mrd; a.min_reach[p] = mr; a.source[p] = cur_o; }
        };
        bool act[PRIM_AHEAD], need[PRIM_AHEAD];
        double floor_a[PRIM_AHEAD];
#pragma unroll
        for (int i = 0; i < PRIM_AHEAD; ++i) {
            const int64_t p = p0 + i * stride;
            act[i] = mr_a[i] >= 0.0 && p != cur;                             // (positions past the end carry -1)
            floor_a[i] = fmax(cc, cj_a[i]);
            need[i] = act[i] && floor_a[i] < mr_a[i];                        // else: mrd >= the floor >= min_reach, nothing can change
        }
        if (FILTER) {
#pragma unroll
            for (int i = 0; i < PRIM_AHEAD; ++i) {
                if (!(need[i] && in_run[i])) continue;
                float acc = 0.f;
                const float *u = up[sub * PRIM_AHEAD + i];
#pragma unroll
                for (int k = 0; k < PRIM_FILTER_D / 4; ++k) {
                    const uint32_t w = cw[i][k];
                    const float t0 = u[4 * k] - (float)(w & 255u), t1 = u[4 * k + 1] - (float)((w >> 8) & 255u);
                    const float t2 = u[4 * k + 2] - (float)((w >> 16) & 255u), t3 = u[4 * k + 3] - (float)(w >> 24);
                    acc = fmaf(t0, t0, acc); acc = fmaf(t1, t1, acc); acc = fmaf(t2, t2, acc); acc = fmaf(t3, t3, acc);
                }
                // ||x_cur - y_p|| less the fp32 rounding of the loop above (a few 1e-6 of it, and 1e-4 of a code step where u and the code
                // cancel), less the point's own residual: when even that cannot undercut min_reach[p], the exact distance changes nothing
                const double sc = (double)run_scale[i];
                const double lb = sc * (double)sqrtf(acc) * (1.0 - 4e-5) - 0.01 * sc - (double)rs[i];
                if (fmax(floor_a[i], lb) >= mr_a[i]) need[i] = false;
            }
        }
        PRIM_MARK(2);                                        // state and codes arrived, bounds evaluated
        // ---- the exact distances that are left, shared out over the workgroup: owner (thread, i) queues, any lane computes
        int slot[PRIM_AHEAD];
#pragma unroll
        for (int i = 0; i < PRIM_AHEAD; ++i) {
            slot[i] = -1;
            if (need[i]) {
                slot[i] = atomicAdd(&q_n, 1);
                q_item[slot[i]] = (unsigned short)(tid * PRIM_AHEAD + i);
                q_mr[slot[i]] = mr_a[i]; q_floor[slot[i]] = floor_a[i];
            }
        }
        __syncthreads();
        for (int s = tid; s < q_n; s += PRIM_NT) {
            const int item = q_item[s], t_own = item / PRIM_AHEAD, i_own = item % PRIM_AHEAD;
            const int64_t p = ((int64_t)blockIdx.x * (PRIM_NT / PRIM_SUB) + t_own / PRIM_SUB) * PRIM_SUB + (t_own % PRIM_SUB) + i_own * stride;
            double mr = q_mr[s];
            exact(p, mr, q_floor[s]);
            q_mr[s] = mr;
        }
        __syncthreads();
        PRIM_MARK(3);                                        // exact distances
#pragma unroll
        for (int i = 0; i < PRIM_AHEAD; ++i) {
            const int64_t p = p0 + i * stride;
            if (slot[i] >= 0) mr_a[i] = q_mr[slot[i]];
            if (act[i] && better(mr_a[i], o_a[i], bw, bj)) { bw = mr_a[i]; bj = o_a[i]; bp = p; }
        }
        for (int64_t p = p0 + PRIM_AHEAD * stride; p < n; p += stride) {        // (beyond the look-ahead, n > 2^20: no filter)
            double mr = a.min_reach[p];
            if (mr < 0.0 || p == cur) continue;
            const double floor_cj = fmax(cc, a.core[p]);
            if (floor_cj < mr) exact(p, mr, floor_cj);
            const int64_t o = a.orig ? (int64_t)a.orig[p] : p;
            if (better(mr, o, bw, bj)) { bw = mr; bj = o; bp = p; }
        }
        block_best(bw, bj, bp, sw, sj, sp);
        if (tid == 0) a.cand[step & 1][blockIdx.x] = Cand{bw, bj, bp};
        PRIM_MARK(4);                                        // candidate reduced and left
#ifdef IDL_PHASE_STAMPS
        if (stamping) { atomicAdd(&prim_phase_sum[5], (unsigned long long)q_n); atomicAdd(&prim_phase_sum[6], 1ull); }
#endif
    }
    if (recorder) { a.mst_cur[step - 1] = cur_src; a.mst_next[step - 1] = cur_o; a.mst_w[step - 1] = cur_w; }
}

__global__ void prim_init_kernel(double *min_reach, int64_t *source, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        min_reach[i] = __builtin_inf(); source[i] = 1;
    }
}

inline int prim_grid(int64_t n)
{
    int64_t g = (n + PRIM_NT - 1) / PRIM_NT;
    if (g > 1024) g = 1024;                                  // four workgroups per CU; from 2^18 points a thread has several, up to 2^20 they all fit its look-ahead
    return (int)(g < 1 ? 1 : g);
}

inline int64_t align256(int64_t b) { return (b + 255) & ~(int64_t)255; }

int prim_run(PrimArgs a, int is_f64, bool filter, void *workspace, void *stream)
{
    IDL_REQUIRE(a.xt && a.core && a.mst_cur && a.mst_next && a.mst_w && workspace, "mst_prim: NULL buffer");
    IDL_REQUIRE(a.n >= 2 && a.n < (1ll << 28) && a.d >= 1 && a.d <= PRIM_MAX_D, "mst_prim: need 2 <= n < 2^28 points of 1..256 features");
    IDL_REQUIRE((((uintptr_t)workspace) & 255u) == 0, "mst_prim: workspace must be 256-byte aligned");
    IDL_REQUIRE(a.start >= 0 && a.start < a.n, "mst_prim: start position out of range");
    const int64_t n = a.n;
    unsigned char *w = (unsigned char *)workspace;
    a.min_reach = (double *)w; w += align256(n * 8);
    a.source = (int64_t *)w; w += align256(n * 8);
    const int g = prim_grid(n);
    a.cand[0] = (Cand *)w; w += align256((int64_t)g * (int64_t)sizeof(Cand));
    a.cand[1] = (Cand *)w;
    const hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(prim_init_kernel, dim3(256), dim3(256), 0, st, a.min_reach, a.source, n);
    void (*kern)(PrimArgs, int64_t, int, int);
    if (a.d == 64 && !is_f64 && n * 64 * 4 < (1ll << 31))     // the latent: 64 float32 features
        kern = filter ? prim_step_kernel<float, true, true> : prim_step_kernel<float, false, true>;
    else
        kern = is_f64 ? (filter ? prim_step_kernel<double, true, false> : prim_step_kernel<double, false, false>)
                      : (filter ? prim_step_kernel<float, true, false> : prim_step_kernel<float, false, false>);
    for (int64_t step = 0; step < n; ++step) {              // step n - 1 + 1: the launch that only records the last edge
        const int scan = step < n - 1 ? 1 : 0;
        hipLaunchKernelGGL(kern, dim3(scan ? g : 1), dim3(PRIM_NT), 0, st, a, step, scan, g);
    }
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

}  // namespace

extern "C" {

int64_t idl_mst_prim_workspace(int64_t n)
{
    if (n < 1) return 256;
    return 2 * align256(n * 8) + 2 * align256((int64_t)prim_grid(n) * (int64_t)sizeof(Cand)) + 256;
}

int idl_debug_prim_phases(unsigned long long *out8)
{
#ifdef IDL_PHASE_STAMPS
    IDL_REQUIRE(out8 != nullptr, "debug_prim_phases: NULL buffer");
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    IDL_HIP_TRY(hipDeviceSynchronize());
    IDL_HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(prim_phase_sum), sizeof(zero)));
    IDL_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(prim_phase_sum), zero, sizeof(zero)));
    return IDL_OK;
#else
    (void)out8;
    idl::set_error("bad argument: %s", "debug_prim_phases: this build has no phase stamps (make STAMPS=1)");
    return IDL_ERR_ARG;
#endif
}

int idl_mst_prim(const void *xt, int is_f64, const double *core, int64_t n, int d, int64_t *mst_cur, int64_t *mst_next, double *mst_w,
                 void *workspace, void *stream)
{
    PrimArgs a{};
    a.xt = xt; a.core = core; a.n = n; a.d = d; a.mst_cur = mst_cur; a.mst_next = mst_next; a.mst_w = mst_w;
    return prim_run(a, is_f64, false, workspace, stream);
}

int idl_mst_prim_local(const void *xt, int is_f64, const double *core, int64_t n, int d, const int32_t *orig, int64_t start,
                       const uint32_t *codes, const float *resid, const int32_t *gid, const float *glo,
                       const double *gscale, int64_t *mst_cur, int64_t *mst_next, double *mst_w, void *workspace, void *stream)
{
    IDL_REQUIRE(orig && codes && resid && gid && glo && gscale, "mst_prim_local: NULL order / filter buffer");
    IDL_REQUIRE(d % 4 == 0 && d <= PRIM_FILTER_D, "mst_prim_local: the number of features must be a multiple of 4, at most 64");
    IDL_REQUIRE(n < (1ll << 31), "mst_prim_local: at most 2^31 - 1 points");
    PrimArgs a{};
    a.xt = xt; a.core = core; a.n = n; a.d = d; a.mst_cur = mst_cur; a.mst_next = mst_next; a.mst_w = mst_w;
    a.orig = orig; a.start = start; a.codes = codes; a.resid = resid; a.gid = gid; a.glo = glo; a.gscale = gscale;
    return prim_run(a, is_f64, true, workspace, stream);
}

}  // extern "C"
