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
#include <stdio.h>
#include "dev_env.h"
#include <stdlib.h>
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

// A workgroup barrier that orders LDS traffic only.  __syncthreads() is `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier`: every barrier of a
// step kernel then also waits for the global loads the wave has just requested for LATER (a whole round trip) and for its global
// stores to be acknowledged (the bounds, the records, the edges: 1-2 us each; tools/stamps_lazy.py found 1 us per barrier).  The
// step kernels exchange nothing through global memory inside a workgroup, so their barriers wait for LDS alone.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// the same on the VALU alone (wave_ops.h: DPP inside the rows, permlane swaps across them -- a __shfl_xor butterfly over three 64-bit
// values is 36 dependent trips through the LDS crossbar): the smallest weight of the wave first, then the smallest (number, position)
// pair among the lanes that hold it, packed into one word (numbers < 2^31, positions < 2^32: idl_mst_prim_lazy's sizes).  The
// lexicographic minimum `better` defines; "nothing" = (inf, INT64_MAX, 0) as in block_best.  sk: PRIM_NT / 64 words.
__device__ __forceinline__ void block_best_packed(double &bw, int64_t &bj, int64_t &bp, double *sw, unsigned long long *sk)
{
    const double wmin = idl_dev::wave_min_d(bw);
    uint64_t key = (bw == wmin && bj != INT64_MAX) ? (((uint64_t)bj << 32) | (uint64_t)(uint32_t)bp) : ~0ull;
    key = idl_dev::wave_min_u64(key);
    const int tid = threadIdx.x;
    if ((tid & 63) == 0) { sw[tid >> 6] = wmin; sk[tid >> 6] = key; }
    lds_barrier();
    double w = sw[0]; uint64_t k = sk[0];
#pragma unroll
    for (int v = 1; v < PRIM_NT / 64; ++v) if (sw[v] < w || (sw[v] == w && sk[v] < k)) { w = sw[v]; k = sk[v]; }
    lds_barrier();
    bw = w;
    bj = k == ~0ull ? INT64_MAX : (int64_t)(k >> 32);
    bp = k == ~0ull ? 0 : (int64_t)(uint32_t)k;
}

// Diagnostic build (make STAMPS=1; tools/stamps_prim.py): workgroups 0, 1/3 and 2/3 of the grid add the time (s_memrealtime, 100 MHz)
// they spend between six marks of every step to prim_phase_sum; idl_debug_prim_phases reads and clears the sums.
#ifdef IDL_PHASE_STAMPS
__device__ unsigned long long prim_phase_sum[8];
#define PRIM_MARK(slot) do { if (stamping) { const uint64_t now_ = __builtin_amdgcn_s_memrealtime(); atomicAdd(&prim_phase_sum[slot], (unsigned long long)(now_ - last_)); last_ = now_; } } while (0)
#else
#define PRIM_MARK(slot) do { } while (0)
#endif
// the same for lazy_step_kernel: [kind][slot], kind 0 = workgroup 0, 1 = another workgroup with an awake run, 2 = one that only keeps balls;
// slots 0..6 the phases, 7 the count, 8 (kind 0) entry - the previous launch's last exit, 9 (kind 0) the previous launch's length
#ifdef IDL_PHASE_STAMPS
__device__ unsigned long long lazy_phase_sum[3][12];
__device__ unsigned long long lazy_clock[2];               // [0] the latest exit of any workgroup, [1] workgroup 0's entry
#define LZ_MARK(slot) do { if (stamping) { const uint64_t now_ = __builtin_amdgcn_s_memrealtime(); atomicAdd(&lazy_phase_sum[kind_][slot], (unsigned long long)(now_ - last_)); last_ = now_; } } while (0)
#define LZ_EXIT() do { if (threadIdx.x == 0 && (blockIdx.x & 7) == 0) atomicMax(&lazy_clock[0], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)
#else
#define LZ_MARK(slot) do { } while (0)
#define LZ_EXIT() do { } while (0)
#endif

template <typename T, bool FILTER, bool D64>
__global__ __launch_bounds__(PRIM_NT, 3) void prim_step_kernel(PrimArgs a, int64_t step, int scan, int n_part)
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


// ================================================================================================================ lazy Prim
// The scan above touches every point outside the tree at every step, and on a latent of far-apart clusters almost all of them for
// nothing: while the tree grows inside one cluster, the points of the others neither can be selected (their min_reach is the
// distance to that cluster, hundreds of times the weights being added) nor need their min_reach be current.  Here a GROUP of points
// may SLEEP: the steps skip it, and it keeps a lower bound of what its min_reach values could have become,
//     lb(g) = min( min_reach of its points when it fell asleep,  min over the nodes added since of  ||x_t - c_g|| - R_g )
// (c_g, R_g: a ball around its points; mrd(t, p) >= ||x_t - x_p|| >= ||x_t - c_g|| - R_g).  A step's winner is committed only while
// its weight is strictly below every sleeping group's bound -- then no sleeping point could have been the argmin, ties included.
// Otherwise the launch STALLS (nothing is changed; the launches queued behind it fall through), and the host has the sleeping
// groups CATCH UP: every sleeping point meets the nodes added since its group fell asleep, in their order, with the same strict-<
// update -- the state is then what the plain scan would have left.  That is the same N^2 / 2 pairs of work, but as a blocked
// loop: a thread keeps its point's 64 coordinates in registers, the nodes stream through LDS 32 at a time, and every distance is
// formed exactly (no bound, no gather) at the f64 VALU rate instead of the memory system's.  A census then decides who sleeps next
// (groups whose smallest min_reach is well above the weights being added), a re-scan of the last node rebuilds the candidates, and
// the steps go on.  Same tree, edge for edge (tests compare with the plain scan).  (Round 3 measured a list of the workgroups with a
// waking run against launching them all: 12.8 us a step against 12.2.  Round 5's steps take the list -- lazy_list_kernel: with the
// bounds kept by a quarter of the workgroups the others have nothing to start for.)
struct LazyState {
    long long n_tree;          // nodes in the tree (the start included)
    long long cur_p, cur_o;    // the node added last: position, original number
    double cur_w, ema;         // its edge's weight; running mean of the recent weights
    int stalled, cand_par;     // 1: the winner could not be committed; which candidate buffer is valid
    int fresh, pad;            // (unused since round 6)
};

struct LazyArgs {
    LazyState *st;             // [2], by launch parity
    int64_t *tree_p;           // [n] positions in the order they were added
    unsigned char *run_asleep; // [ceil(n / 256)] 1: every point of the run sleeps (or is in the tree)
    unsigned char *pas;        // [n] 1: the point's group sleeps
    int n_groups;
    const int64_t *gfirst;     // [G + 1] first position of each group
    int *asleep;               // [G] 0 awake, 1 asleep, 2 nothing left outside the tree
    int64_t *upto;             // [G] the tree nodes [0, upto) have met the group
    double *minmr;             // [G] smallest min_reach of its points outside the tree, as of upto (+inf: none left)
    double *lbp;               // [2][G] by launch parity: lower bound of mrd(t, p) over the nodes added since upto and the group's points
    const double *gc;          // [G][64] centre of the group's ball
    const double *gr;          // [G] its radius (>= ||x_p - c_g||)
    const float *xrow;         // [n][64] the points row-major (a node's coordinates in one piece)
    double *gmin; int64_t *galive;       // census scratch [G]
    int *wg_list;              // [1 + full_grid] wg_list[0] = L; then the step's workgroups: number | awake runs << 16 (lazy_list_kernel)
    int full_grid;             // prim_grid(n): positions are laid out by it whatever the launched grid is
};

constexpr int LAZY_NCH = 32;           // nodes a catch-up pass stages in LDS at a time

__global__ __launch_bounds__(PRIM_NT, 3) void lazy_step_kernel(PrimArgs a, LazyArgs z, int64_t launch, int rescan)
{
    __shared__ double sw[PRIM_NT / 64];
    __shared__ int64_t sj[PRIM_NT / 64];
    __shared__ double xc[PRIM_FILTER_D];
    __shared__ float up[PRIM_RUNS][PRIM_FILTER_D];
    __shared__ int q_n;
    __shared__ unsigned short q_item[PRIM_NT * PRIM_AHEAD];
    __shared__ double q_mr[PRIM_NT * PRIM_AHEAD], q_floor[PRIM_NT * PRIM_AHEAD];
    static_assert(PRIM_NT == PRIM_SUB, "one sub-block per workgroup here");
    const int tid = threadIdx.x;
#ifdef IDL_PHASE_STAMPS
    uint64_t last_ = __builtin_amdgcn_s_memrealtime();
    const uint64_t entry_ = last_;
    bool stamping = tid == 0 && !rescan && (blockIdx.x == 0 || blockIdx.x % 61 == 7);   // (a sample: ~500 workgroups adding to the same words distort what they time)
    int kind_ = 2;
#endif
    const int64_t n = a.n;
    const int G = z.n_groups;
    const int par = (int)(launch & 1);
    // (this workgroup's number and its run flags: one word of the list, requested in front of the state of the walk -- two loads, one wait)
    const int entry = z.wg_list[1 + blockIdx.x];
    const LazyState S = z.st[par];
    const int wg = __builtin_amdgcn_readfirstlane(entry) & 0xFFFF, wg_mask = __builtin_amdgcn_readfirstlane(entry) >> 16;
    LazyState *nx = &z.st[par ^ 1];
    const bool lead = wg == 0 && tid == 0;                   // (workgroup 0 keeps a bound: always listed, always first)
    // sleeping group g's bound is kept by wave g % 4 of workgroup g / 4 (round 5; it was workgroup g: every workgroup of the grid then
    // ran the whole decision, and the grid no longer fits the chip at once when the kernel takes more than 128 registers)
    const bool ball_duty = wg * (PRIM_NT / 64) < G;
    const int ball_g = wg * (PRIM_NT / 64) + (tid >> 6);
    const bool has_ball = ball_g < G;
    if (S.stalled || S.n_tree >= n) {                        // fall through: the state and the bounds are handed on unchanged
        if (lead) *nx = S;
        if (has_ball && (tid & 63) == 0) z.lbp[(par ^ 1) * G + ball_g] = z.lbp[par * G + ball_g];
        return;
    }
    const int64_t stride = (int64_t)z.full_grid * PRIM_NT;
    const int64_t p0 = (int64_t)wg * PRIM_NT + tid;
    const float *xt = (const float *)a.xt;
    // ---- A step is a chain of memory round trips; whatever does not need the previous trip's answer is requested together.
    // Trip 1: the four run flags.  Trip 2, all in flight before anything of it is waited for: min_reach, core distance and sleep flag
    // of the awake runs' points -- 17 BYTES A POINT, nothing else -- the previous step's candidates, every group's bound (not only
    // the sleeping groups': whether a group sleeps is in the same trip), the ball this workgroup keeps.  Trip 3: the new node.
    // Trip 4: group, residual and the 64 bytes of codes of the points whose entry CAN still fall (floor < min_reach: a few per
    // workgroup and step), and the original numbers of the points that can become the workgroup's candidate (ties go by them).
    // Trip 5: the exact distances that survive the bound.
    // (Round 5.  Half the runs of a 10^6-point job are awake in a typical step -- 480 workgroups, stamps: tools/stamps_lazy.py -- and
    //  the round-3 kernel read 97 bytes of state and codes for every one of their points in every step: 48 MB a launch, 12 us at
    //  4 TB/s.  It was bound by that, not by its round trips: with all of them merged and the same bytes it took 13.1 us.)
    bool run_on[PRIM_AHEAD];
    bool any_on = false;
#pragma unroll
    for (int i = 0; i < PRIM_AHEAD; ++i) {
        run_on[i] = ((wg_mask >> i) & 1) != 0;
        any_on |= run_on[i];
    }
    Cand *cand_out = a.cand[S.cand_par ^ 1];
    if (!any_on && !ball_duty) {                             // nothing to scan, nothing to keep: leave an empty candidate
        if (tid == 0) cand_out[blockIdx.x] = Cand{__builtin_inf(), INT64_MAX, 0};
        LZ_EXIT();
        return;
    }
#ifdef IDL_PHASE_STAMPS
    kind_ = blockIdx.x == 0 ? 0 : any_on ? 1 : 2;
    if (stamping && blockIdx.x == 0) {
        const unsigned long long prev_exit = atomicMax(&lazy_clock[0], 0ull), prev_entry = lazy_clock[1];
        if (prev_entry != 0 && prev_exit > prev_entry && entry_ > prev_exit) {
            atomicAdd(&lazy_phase_sum[0][8], (unsigned long long)(entry_ - prev_exit));
            atomicAdd(&lazy_phase_sum[0][9], prev_exit - prev_entry);
            atomicAdd(&lazy_phase_sum[0][10], 1ull);
        }
        lazy_clock[1] = entry_;
    }
#endif
    LZ_MARK(0);                                              // the run flags
    // ---- trip 2, requests
    double mr_a[PRIM_AHEAD], cj_a[PRIM_AHEAD];
    double run_scale[PRIM_AHEAD];                            // (float)gscale of the run's group; rounded where it is used: the request is not waited for here
    int run_g[PRIM_AHEAD], pas_a[PRIM_AHEAD];
    static_assert(PRIM_RUNS * 64 == PRIM_NT && PRIM_FILTER_D == 64 && PRIM_AHEAD == 4, "one (run, feature) per thread");
#pragma unroll
    for (int i = 0; i < PRIM_AHEAD; ++i) {
        const int64_t first = (int64_t)wg * PRIM_NT + i * stride;
        const int64_t p = p0 + i * stride;
        mr_a[i] = -1.0; cj_a[i] = 0.0; run_g[i] = 0; pas_a[i] = 1;
        if (run_on[i]) {                                     // (uniform)
            const int64_t pc = p < n ? p : first;            // a lane beyond the end reads the run's first point; dropped below
            mr_a[i] = a.min_reach[pc];
            cj_a[i] = a.core[pc];
            pas_a[i] = z.pas[pc];
            run_g[i] = a.gid[first];
        }
    }
    constexpr int PRE = 4;                                   // candidates / group bounds a thread requests up front: 4 x 256 covers 2^20 points
    Cand cv[PRE];
    int g_as[PRE];
    double g_mm[PRE], g_lb[PRE];
    if (!rescan) {
        const Cand *pc = a.cand[S.cand_par];
#pragma unroll
        for (int it = 0; it < PRE; ++it) {
            const int g = tid + it * PRIM_NT;
            cv[it] = Cand{__builtin_inf(), INT64_MAX, 0};
            if (g < (int)gridDim.x) cv[it] = pc[g];
            g_as[it] = 0; g_mm[it] = 0.0; g_lb[it] = 0.0;
            if (g < G) { g_as[it] = z.asleep[g]; g_mm[it] = z.minmr[g]; g_lb[it] = z.lbp[par * G + g]; }
        }
    }
    double ball_c = 0.0, ball_lb = 0.0, ball_r = 0.0;        // this workgroup's ball (its first wave)
    int ball_as = 0;
    if (has_ball) {
        ball_lb = z.lbp[par * G + ball_g];
        ball_as = z.asleep[ball_g];
        ball_c = z.gc[(int64_t)ball_g * PRIM_FILTER_D + (tid & 63)];
        ball_r = z.gr[ball_g];
    }
    // (nothing above has been waited for.  The values are pinned HERE: left alone, the compiler moves the first use of a run's state
    //  up into the block that requested it, and the wait with it -- one round trip per run again)
    asm volatile("" ::: "memory");
#define LZ_PIN(x) asm volatile("" : "+v"(x))
#pragma unroll
    for (int i = 0; i < PRIM_AHEAD; ++i) { LZ_PIN(mr_a[i]); LZ_PIN(cj_a[i]); LZ_PIN(run_g[i]); LZ_PIN(pas_a[i]); }
    if (!rescan) {
#pragma unroll
        for (int it = 0; it < PRE; ++it) { LZ_PIN(cv[it].w); LZ_PIN(cv[it].j); LZ_PIN(cv[it].p); LZ_PIN(g_as[it]); LZ_PIN(g_mm[it]); LZ_PIN(g_lb[it]); }
    }
    LZ_PIN(ball_lb); LZ_PIN(ball_as); LZ_PIN(ball_c); LZ_PIN(ball_r);
#undef LZ_PIN
    LZ_MARK(1);                                              // trip 2 answered
    // ---- trip 2, answers: the state first (requested first)
    int my_run_g;
    {
        const int rr = tid >> 6;
#pragma unroll
        for (int i = 0; i < PRIM_AHEAD; ++i) {
            const int64_t p = p0 + i * stride;
            const bool on = run_on[i] && p < n;
            if (!on || pas_a[i] != 0) mr_a[i] = -1.0;        // (a lane beyond the end; a run that straddles a sleeping and a waking group)
            if (!on) cj_a[i] = 0.0;
            run_scale[i] = a.gscale[run_g[i]];
        }
        my_run_g = rr == 0 ? run_g[0] : rr == 1 ? run_g[1] : rr == 2 ? run_g[2] : run_g[3];
    }
    const double my_sc = a.gscale[my_run_g];                 // (in flight during the decision)
    const float my_lo = a.glo[(int64_t)my_run_g * PRIM_FILTER_D + (tid & 63)];
    // ---- the winner of the previous scan; it is committed only if no sleeping group could hold a better point
    int64_t cur = S.cur_p, cur_o = S.cur_o;
    double cur_w = S.cur_w;
    if (!rescan) {
        const Cand *pc = a.cand[S.cand_par];
        double bw = __builtin_inf(); int64_t bj = INT64_MAX, bp = 0;
#pragma unroll
        for (int it = 0; it < PRE; ++it) if (better(cv[it].w, cv[it].j, bw, bj)) { bw = cv[it].w; bj = cv[it].j; bp = cv[it].p; }
        for (int g = tid + PRE * PRIM_NT; g < (int)gridDim.x; g += PRIM_NT) {
            const Cand c = pc[g];
            if (better(c.w, c.j, bw, bj)) { bw = c.w; bj = c.j; bp = c.p; }
        }
        block_best_packed(bw, bj, bp, sw, (unsigned long long *)sj);
        double lb = __builtin_inf();
#pragma unroll
        for (int it = 0; it < PRE; ++it) if (g_as[it] == 1) lb = fmin(lb, fmin(g_mm[it], g_lb[it]));              // (2: nothing left in it)
        for (int g = tid + PRE * PRIM_NT; g < G; g += PRIM_NT) if (z.asleep[g] == 1) lb = fmin(lb, fmin(z.minmr[g], z.lbp[par * G + g]));
        lb = idl_dev::wave_min_d(lb);
        if ((tid & 63) == 0) sw[tid >> 6] = lb;
        lds_barrier();
        lb = fmin(fmin(sw[0], sw[1]), fmin(sw[2], sw[3]));
        lds_barrier();
        if (!(bw < lb)) {                                    // STALL (every workgroup decides the same from the same data)
            if (lead) { *nx = S; nx->stalled = 1; }
            if (has_ball && (tid & 63) == 0) z.lbp[(par ^ 1) * G + ball_g] = ball_lb;
            return;
        }
        cur = bp; cur_o = bj; cur_w = bw;
        LZ_MARK(2);                                          // decided
        if (lead) {
            a.mst_cur[S.n_tree - 1] = a.source[cur]; a.mst_next[S.n_tree - 1] = cur_o; a.mst_w[S.n_tree - 1] = cur_w;
            a.min_reach[cur] = -1.0;                         // in the tree (this launch skips it by position)
            z.tree_p[S.n_tree] = cur;
        }
    }
    if (lead) {
        LazyState t = S;
        if (!rescan) { t.n_tree = S.n_tree + 1; t.cur_p = cur; t.cur_o = cur_o; t.cur_w = cur_w; t.ema = S.ema + (cur_w - S.ema) * (1.0 / 64.0); }
        t.cand_par = S.cand_par ^ 1;
        *nx = t;
    }
    if (tid < PRIM_FILTER_D) xc[tid] = (double)xt[(int64_t)tid * n + cur];
    if (tid == 0) q_n = 0;
    const double cc = a.core[cur];
    {
        const int k = tid & 63;
        lds_barrier();
        up[tid >> 6][k] = (float)((xc[k] - (double)my_lo) / my_sc);
    }
    // ---- sleeping group g's bound meets the new node (workgroup g, its first wave)
    if (has_ball) {                                          // (a whole wave or none of it)
        double lbv = ball_lb;
        if (!rescan && ball_as == 1) {
            const double t = xc[tid & 63] - ball_c;
            const double d2 = idl_dev::wave_sum_d(t * t);      // (any order of the sum: the bound keeps 1e-12 of slack)
            const double b = __dsqrt_rn(d2) * (1.0 - 1e-12) - ball_r;
            lbv = fmin(lbv, b > 0.0 ? b : 0.0);
        }
        if ((tid & 63) == 0) z.lbp[(par ^ 1) * G + ball_g] = lbv;
    }
    lds_barrier();
    LZ_MARK(3);                                              // the new node, the runs' boxes, the ball
    if (!any_on) {
        if (tid == 0) cand_out[blockIdx.x] = Cand{__builtin_inf(), INT64_MAX, 0};
#ifdef IDL_PHASE_STAMPS
        if (stamping) atomicAdd(&lazy_phase_sum[kind_][7], 1ull);
#endif
        LZ_EXIT();
        return;
    }
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.xt, 0, 0xffffffff, 0x00020000);
    const int col_bytes = (int)n * 4;
    double bw = __builtin_inf(); int64_t bj = INT64_MAX, bp = 0;
    auto exact = [&](int64_t p, double &mr, double floor_cj) {
        double acc = 0.0;
        uint32_t v[64];
#pragma unroll
        for (int k = 0; k < 64; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, (uint32_t)p * 4u, k * col_bytes, 0);
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            if ((k & 7) == 0) __builtin_amdgcn_sched_barrier(0);
            const double t = xc[k] - (double)__uint_as_float(v[k]);
            acc = idl_dev::square_then_add(acc, t);
        }
        const double mrd = fmax(floor_cj, __dsqrt_rn(acc));
        if (mrd < mr) { mr = mrd; a.min_reach[p] = mr; a.source[p] = cur_o; }
    };
    bool act[PRIM_AHEAD], need[PRIM_AHEAD];
    double floor_a[PRIM_AHEAD];
#pragma unroll
    for (int i = 0; i < PRIM_AHEAD; ++i) {
        const int64_t p = p0 + i * stride;
        act[i] = mr_a[i] >= 0.0 && p != cur;
        floor_a[i] = fmax(cc, cj_a[i]);
        need[i] = act[i] && floor_a[i] < mr_a[i];
    }
    // the original numbers the candidate will be chosen by (ties), requested with trip 4: the workgroup's smallest weight will be
    // held either by a point that holds the wave's smallest weight NOW or by a point whose entry falls in this step
    int oj[PRIM_AHEAD];
    {
        double wpre = __builtin_inf();
#pragma unroll
        for (int i = 0; i < PRIM_AHEAD; ++i) if (act[i]) wpre = fmin(wpre, mr_a[i]);
        wpre = idl_dev::wave_min_d(wpre);
#pragma unroll
        for (int i = 0; i < PRIM_AHEAD; ++i) {
            const int64_t p = p0 + i * stride;
            oj[i] = 0x7FFFFFFF;
            if (act[i] && (need[i] || mr_a[i] == wpre)) oj[i] = a.orig[p];
        }
    }
    // ---- trip 4: what the bound needs, for the points whose entry can still fall
    {
        uint32_t cw[PRIM_AHEAD][PRIM_FILTER_D / 4];
        float rs[PRIM_AHEAD];
        int g_own[PRIM_AHEAD];
        const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.codes, 0, 0xffffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < PRIM_AHEAD; ++i) {
            const int64_t p = p0 + i * stride;
            g_own[i] = -1; rs[i] = 0.f;
#pragma unroll
            for (int k = 0; k < PRIM_FILTER_D / 4; ++k) cw[i][k] = 0u;
            if (need[i]) {
                g_own[i] = a.gid[p];
                rs[i] = a.resid[p];
#pragma unroll
                for (int k = 0; k < PRIM_FILTER_D / 4; ++k) cw[i][k] = __builtin_amdgcn_raw_buffer_load_b32(c_rsrc, (uint32_t)p * 4u, k * (int)n * 4, 0);
            }
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < PRIM_AHEAD; ++i) { asm volatile("" : "+v"(g_own[i])); asm volatile("" : "+v"(rs[i])); }
#pragma unroll
        for (int i = 0; i < PRIM_AHEAD; ++i) {
            if (!(need[i] && g_own[i] == run_g[i])) continue;    // (a point of another group than its run's: no codes in this box, exact distance)
            float acc = 0.f;
            const float *u = up[i];
#pragma unroll
            for (int k = 0; k < PRIM_FILTER_D / 4; ++k) {
                const uint32_t w = cw[i][k];
                const float t0 = u[4 * k] - (float)(w & 255u), t1 = u[4 * k + 1] - (float)((w >> 8) & 255u);
                const float t2 = u[4 * k + 2] - (float)((w >> 16) & 255u), t3 = u[4 * k + 3] - (float)(w >> 24);
                acc = fmaf(t0, t0, acc); acc = fmaf(t1, t1, acc); acc = fmaf(t2, t2, acc); acc = fmaf(t3, t3, acc);
            }
            const double sc = (double)(float)run_scale[i];
            const double lb = sc * (double)sqrtf(acc) * (1.0 - 4e-5) - 0.01 * sc - (double)rs[i];      // (prim_step_kernel has the reasoning)
            if (fmax(floor_a[i], lb) >= mr_a[i]) need[i] = false;
        }
    }
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
    lds_barrier();
    LZ_MARK(4);                                              // bounds evaluated, queue filled
    for (int s = tid; s < q_n; s += PRIM_NT) {
        const int item = q_item[s], t_own = item / PRIM_AHEAD, i_own = item % PRIM_AHEAD;
        const int64_t p = (int64_t)wg * PRIM_NT + t_own + i_own * stride;
        double mr = q_mr[s];
        exact(p, mr, q_floor[s]);
        q_mr[s] = mr;
    }
    lds_barrier();
    LZ_MARK(5);                                              // exact distances
    // ---- the workgroup's candidate: smallest weight first, then the smallest original number among the points that hold it (the
    // lexicographic minimum `better` defines; their numbers came with trip 4)
    double wmin = __builtin_inf();
#pragma unroll
    for (int i = 0; i < PRIM_AHEAD; ++i) {
        if (slot[i] >= 0) mr_a[i] = q_mr[slot[i]];
        if (act[i]) wmin = fmin(wmin, mr_a[i]);
    }
    wmin = idl_dev::wave_min_d(wmin);
    {
#pragma unroll
        for (int i = 0; i < PRIM_AHEAD; ++i) {
            const int64_t p = p0 + i * stride;
            if (act[i] && mr_a[i] == wmin && (int64_t)oj[i] < bj) { bj = (int64_t)oj[i]; bp = p; }
        }
        bw = bj != INT64_MAX ? wmin : __builtin_inf();       // (a lane that holds none of the wave's smallest entries offers nothing)
    }
    block_best_packed(bw, bj, bp, sw, (unsigned long long *)sj);
    if (tid == 0) cand_out[blockIdx.x] = Cand{bw, bj, bp};
    LZ_MARK(6);                                              // candidate reduced and left
#ifdef IDL_PHASE_STAMPS
    if (stamping) { atomicAdd(&lazy_phase_sum[kind_][7], 1ull); atomicAdd(&lazy_phase_sum[kind_][11], (unsigned long long)q_n); }
#endif
    LZ_EXIT();
}

// every sleeping point meets the nodes tree_p[upto[g] .. n_tree) in their order: its coordinates in registers, the nodes through LDS
__global__ __launch_bounds__(256) void lazy_catchup_kernel(PrimArgs a, LazyArgs z, int64_t n_tree)
{
    __shared__ float xn[LAZY_NCH][PRIM_FILTER_D];
    __shared__ double cn[LAZY_NCH];
    __shared__ int64_t on[LAZY_NCH], tn[LAZY_NCH];
    __shared__ int64_t s_first;
    __shared__ double ballb[2][LAZY_NCH];                    // lower bound of ||x_node - x_p|| over the points of the run's first / last point's group
    __shared__ int s_g[2];
    const int tid = threadIdx.x;
    const int64_t p = (int64_t)blockIdx.x * 256 + tid;
    const bool inside = p < a.n;
    double mr = inside ? a.min_reach[p] : -1.0;
    const int g = inside ? a.gid[p] : 0;
    const bool mine = inside && mr >= 0.0 && z.asleep[g] != 0;
    const int64_t my_first = mine ? z.upto[g] : n_tree;
    if (tid == 0) {
        s_first = n_tree;
        const int64_t pl = (int64_t)blockIdx.x * 256 + 255 < a.n ? (int64_t)blockIdx.x * 256 + 255 : a.n - 1;
        s_g[0] = a.gid[(int64_t)blockIdx.x * 256]; s_g[1] = a.gid[pl];
    }
    __syncthreads();
    if (my_first < n_tree) atomicMin((unsigned long long *)&s_first, (unsigned long long)my_first);
    __syncthreads();
    const int64_t j0 = s_first;
    if (j0 >= n_tree) return;
    float xp[PRIM_FILTER_D];
    {
        const float4 *src = (const float4 *)(z.xrow + (inside ? p : 0) * PRIM_FILTER_D);
#pragma unroll
        for (int i = 0; i < PRIM_FILTER_D / 4; ++i) { const float4 t = src[i]; xp[4 * i] = t.x; xp[4 * i + 1] = t.y; xp[4 * i + 2] = t.z; xp[4 * i + 3] = t.w; }
    }
    const double cj = inside ? a.core[p] : 0.0;
    int64_t src_new = -1;
    for (int64_t jb = j0; jb < n_tree; jb += LAZY_NCH) {
        const int cnt = (int)(n_tree - jb < LAZY_NCH ? n_tree - jb : LAZY_NCH);
        __syncthreads();
        if (tid < cnt) { const int64_t t = z.tree_p[jb + tid]; tn[tid] = t; cn[tid] = a.core[t]; on[tid] = (int64_t)a.orig[t]; }
        __syncthreads();
        for (int idx = tid; idx < cnt * PRIM_FILTER_D; idx += 256) xn[idx >> 6][idx & 63] = z.xrow[tn[idx >> 6] * PRIM_FILTER_D + (idx & 63)];
        __syncthreads();
        // (round 5) a node that is far from a whole group's ball is far from each of its points: ||x_node - x_p|| >= ||x_node - c_g|| - r_g
        // (r_g >= ||x_p - c_g||), formed once per (node, group of the run's first / last point) instead of 64 multiply-adds per
        // (node, point) -- the float32 distances of far pairs were most of this pass (34-40 ms a call, 64-190 calls a job)
        for (int pr = tid >> 6; pr < 2 * cnt; pr += 4) {
            const int i = pr >> 1, w = pr & 1, gg = s_g[w];
            const double t = (double)xn[i][tid & 63] - z.gc[(int64_t)gg * PRIM_FILTER_D + (tid & 63)];
            const double d2 = idl_dev::wave_sum_d(t * t);
            const double bd = __dsqrt_rn(d2) * (1.0 - 1e-12) - z.gr[gg];
            if ((tid & 63) == 0) ballb[w][i] = bd > 0.0 ? bd : 0.0;
        }
        __syncthreads();
        if (!mine) continue;
        const int which = g == s_g[0] ? 0 : g == s_g[1] ? 1 : -1;
        for (int i = 0; i < cnt; ++i) {
            if (jb + i < my_first) continue;
            const double floor_cj = fmax(cn[i], cj);
            if (!(floor_cj < mr)) continue;
            if (which >= 0 && ballb[which][i] >= mr) continue;
            // the distance in float32 first (differences of float32 values, 64 fused multiply-adds: within 1e-5 of the true one): when
            // even that less its error cannot undercut min_reach, the exact one changes nothing
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 s0 = {0.f, 0.f}, s1 = {0.f, 0.f};           // (packed, two chains: only a bound, any order of summation does)
#pragma unroll
            for (int k = 0; k < PRIM_FILTER_D; k += 4) {
                const f32x2 t0 = f32x2{xn[i][k], xn[i][k + 1]} - f32x2{xp[k], xp[k + 1]};
                const f32x2 t1 = f32x2{xn[i][k + 2], xn[i][k + 3]} - f32x2{xp[k + 2], xp[k + 3]};
                s0 = __builtin_elementwise_fma(t0, t0, s0); s1 = __builtin_elementwise_fma(t1, t1, s1);
            }
            const float acc32 = (s0.x + s0.y) + (s1.x + s1.y);
            if (fmax(floor_cj, (double)sqrtf(acc32) * (1.0 - 2e-5)) >= mr) continue;
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < PRIM_FILTER_D; ++k) {
                const double t = (double)xn[i][k] - (double)xp[k];
                acc = idl_dev::square_then_add(acc, t);      // the scan's arithmetic: product and sum each rounded, in feature order
            }
            const double mrd = fmax(floor_cj, __dsqrt_rn(acc));
            if (mrd < mr) { mr = mrd; src_new = on[i]; }
        }
    }
    if (mine && src_new >= 0) { a.min_reach[p] = mr; a.source[p] = src_new; }
}

// census, workgroup per group: the smallest min_reach of its points outside the tree and their number
__global__ __launch_bounds__(256) void lazy_census_kernel(PrimArgs a, LazyArgs z)
{
    __shared__ double smin[4];
    __shared__ long long scnt[4];
    const int g = blockIdx.x, tid = threadIdx.x;
    double m = __builtin_inf();
    long long c = 0;
    for (int64_t p = z.gfirst[g] + tid; p < z.gfirst[g + 1]; p += 256) {
        const double mr = a.min_reach[p];
        if (mr >= 0.0) { m = fmin(m, mr); ++c; }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { m = fmin(m, __shfl_xor(m, o, 64)); c += __shfl_xor(c, o, 64); }
    if ((tid & 63) == 0) { smin[tid >> 6] = m; scnt[tid >> 6] = c; }
    __syncthreads();
    if (tid == 0) { z.gmin[g] = fmin(fmin(smin[0], smin[1]), fmin(smin[2], smin[3])); z.galive[g] = scnt[0] + scnt[1] + scnt[2] + scnt[3]; }
}

// who sleeps from now on (one workgroup): groups whose smallest min_reach is well above both the weights being added and the best
// candidate there is (so that the next step cannot stall); a group with nothing left outside the tree sleeps for good.  With few
// points left (the tail: noise, joined one by one at rising weights) nobody sleeps.
__global__ __launch_bounds__(256) void lazy_policy_kernel(PrimArgs a, LazyArgs z, int par, int64_t n_tree, int allow_sleep)
{
    __shared__ double sbest[4];
    __shared__ long long salive[4];
    const int tid = threadIdx.x, G = z.n_groups;
    double best = __builtin_inf();
    long long alive = 0;
    for (int g = tid; g < G; g += 256) { best = fmin(best, z.gmin[g]); alive += z.galive[g]; }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { best = fmin(best, __shfl_xor(best, o, 64)); alive += __shfl_xor(alive, o, 64); }
    if ((tid & 63) == 0) { sbest[tid >> 6] = best; salive[tid >> 6] = alive; }
    __syncthreads();
    best = fmin(fmin(sbest[0], sbest[1]), fmin(sbest[2], sbest[3]));
    alive = salive[0] + salive[1] + salive[2] + salive[3];
    const double ema = z.st[par].ema;
    const bool tail = !allow_sleep || alive * 32 < a.n;
    const double theta = fmax(2.0 * ema, 1.5 * best);
    __shared__ double xcur[PRIM_FILTER_D];
    if (tid < PRIM_FILTER_D) xcur[tid] = (double)z.xrow[z.st[par].cur_p * PRIM_FILTER_D + tid];
    __syncthreads();
    for (int g = tid; g < G; g += 256) {
        const bool dead = z.galive[g] == 0;
        // ... and whose ball is as far from where the tree is growing: a loose group (noise) near the tree would be woken by the very
        // next node that comes within its radius of its centre
        double d2 = 0.0;
        for (int k = 0; k < PRIM_FILTER_D; ++k) { const double t = xcur[k] - z.gc[(int64_t)g * PRIM_FILTER_D + k]; d2 += t * t; }
        const double ball = __dsqrt_rn(d2) - z.gr[g];
        const bool sleep = dead || (!tail && z.gmin[g] > theta && ball > theta);
        z.asleep[g] = dead ? 2 : sleep ? 1 : 0;
        z.minmr[g] = z.gmin[g];
        z.lbp[g] = __builtin_inf(); z.lbp[G + g] = __builtin_inf();
        z.upto[g] = n_tree;
    }
}

// the per-point and per-run flags of the sleeping groups
__global__ __launch_bounds__(256) void lazy_flags_kernel(PrimArgs a, LazyArgs z)
{
    __shared__ int s_any;
    const int tid = threadIdx.x;
    const int64_t p = (int64_t)blockIdx.x * 256 + tid;
    if (tid == 0) s_any = 0;
    __syncthreads();
    bool awake = false;
    if (p < a.n) {
        const int sl = z.asleep[a.gid[p]];
        z.pas[p] = (unsigned char)(sl != 0);
        awake = sl == 0 && a.min_reach[p] >= 0.0;
    }
    if (awake) s_any = 1;
    __syncthreads();
    if (tid == 0) z.run_asleep[blockIdx.x] = s_any ? 0 : 1;
}

// The step's grid (round 5): the workgroups that keep a sleeping group's bound (the first ceil(G / 4)) and those with an awake run,
// in ascending order, each with its four run flags -- built whenever the flags change (a census), read back by the host as the next
// launches' grid.  Half the workgroups of a 10^6-point job had nothing to do but to start and leave, and the last of 977 entered
// 3-4 us behind the first (tools/stamps_lazy.py).
__global__ __launch_bounds__(1024) void lazy_list_kernel(PrimArgs a, LazyArgs z)
{
    __shared__ int wcount[16];
    const int t = threadIdx.x, fg = z.full_grid;
    const int keepers = (z.n_groups + PRIM_NT / 64 - 1) / (PRIM_NT / 64);
    int mask = 0;
    if (t < fg) {
#pragma unroll
        for (int i = 0; i < PRIM_AHEAD; ++i) {
            const int64_t first = ((int64_t)t + (int64_t)i * fg) * PRIM_NT;
            if (first < a.n && z.run_asleep[first / PRIM_NT] == 0) mask |= 1 << i;
        }
    }
    const bool keep = t < fg && (t < keepers || mask != 0);
    const unsigned long long bal = __ballot(keep);
    const int before = __popcll(bal & ((1ull << (t & 63)) - 1ull));
    if ((t & 63) == 0) wcount[t >> 6] = __popcll(bal);
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (t >> 6); ++w) base += wcount[w];
    if (keep) z.wg_list[1 + base + before] = t | (mask << 16);
    if (t == 0) { int total = 0; for (int w = 0; w < 16; ++w) total += wcount[w]; z.wg_list[0] = total; }
}

__global__ void lazy_init_kernel(PrimArgs a, LazyArgs z)
{
    const int64_t n = a.n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        a.min_reach[i] = i == a.start ? -1.0 : __builtin_inf(); a.source[i] = 1; z.pas[i] = 0;
        if (i % 256 == 0) z.run_asleep[i / 256] = 0;
    }
    if (blockIdx.x == 0) {
        for (int g = threadIdx.x; g < z.n_groups; g += blockDim.x) {
            z.asleep[g] = 0; z.upto[g] = 1; z.minmr[g] = __builtin_inf(); z.lbp[g] = __builtin_inf(); z.lbp[z.n_groups + g] = __builtin_inf();
        }
        if (threadIdx.x == 0) {
            LazyState t{};
            t.n_tree = 1; t.cur_p = a.start; t.cur_o = (long long)a.orig[a.start]; t.cur_w = 0.0; t.ema = 0.0; t.stalled = 0; t.cand_par = 0;
            z.st[0] = t; z.st[1] = t;
            z.tree_p[0] = a.start;
        }
    }
}


// (Round 5's two multi-node variants of the lazy step -- lazy_reduce_kernel + lazy_multi_kernel: the decision in a launch of its own; lazy_fold_kernel: the
//  decision inside the step -- were exact, measured no faster on BASELINE cfg5's latent (History) and opt-in; round 6 removed them.  One of them
//  carried a correctness condition nobody had explained -- a build of lazy_fold_kernel that spilled 12 bytes a lane produced wrong records -- and a kernel with an
//  unexplained condition does not ship.  `git log` has the 1 090 lines.)

struct LazyLayout { int64_t min_reach, source, cand0, cand1, st, tree_p, run_asleep, pas, asleep, upto, minmr, lbp, gmin, galive, wg_list, total; };

inline LazyLayout lazy_layout(int64_t n, int n_groups)
{
    LazyLayout l{};
    int64_t o = 0;
    const int g = prim_grid(n);
    auto take = [&](int64_t bytes) { const int64_t at = o; o += align256(bytes); return at; };
    l.min_reach = take(n * 8); l.source = take(n * 8);
    l.cand0 = take((int64_t)g * (int64_t)sizeof(Cand)); l.cand1 = take((int64_t)g * (int64_t)sizeof(Cand));
    l.st = take(2 * (int64_t)sizeof(LazyState)); l.tree_p = take(n * 8);
    l.run_asleep = take((n + 255) / 256); l.pas = take(n);
    l.asleep = take((int64_t)n_groups * 4); l.upto = take((int64_t)n_groups * 8); l.minmr = take((int64_t)n_groups * 8);
    l.lbp = take((int64_t)n_groups * 16); l.gmin = take((int64_t)n_groups * 8); l.galive = take((int64_t)n_groups * 8);
    l.wg_list = take((int64_t)(g + 1) * 4);
    l.total = o + 256;
    return l;
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

int idl_debug_lazy_phases(unsigned long long *out36)
{
#ifdef IDL_PHASE_STAMPS
    IDL_REQUIRE(out36 != nullptr, "debug_lazy_phases: NULL buffer");
    unsigned long long zero[36] = {};
    IDL_HIP_TRY(hipDeviceSynchronize());
    IDL_HIP_TRY(hipMemcpyFromSymbol(out36, HIP_SYMBOL(lazy_phase_sum), sizeof(zero)));
    IDL_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(lazy_phase_sum), zero, sizeof(zero)));
    return IDL_OK;
#else
    (void)out36;
    idl::set_error("bad argument: %s", "debug_lazy_phases: this build has no phase stamps (make STAMPS=1)");
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

int64_t idl_mst_prim_lazy_workspace(int64_t n, int n_groups)
{
    if (n < 1 || n_groups < 1) return 256;
    return lazy_layout(n, n_groups).total;
}

int idl_mst_prim_lazy(const void *xt, const float *xrow, const double *core, int64_t n, int d, const int32_t *orig, int64_t start,
                      const uint32_t *codes, const float *resid, const int32_t *gid, const float *glo, const double *gscale, int n_groups,
                      const int64_t *gfirst, const double *gcentre, const double *gradius, int64_t *mst_cur, int64_t *mst_next,
                      double *mst_w, void *workspace, void *stream, int64_t *stats3)
{
    IDL_REQUIRE(xt && xrow && core && orig && codes && resid && gid && glo && gscale && gfirst && gcentre && gradius, "mst_prim_lazy: NULL buffer");
    IDL_REQUIRE(mst_cur && mst_next && mst_w && workspace, "mst_prim_lazy: NULL buffer");
    IDL_REQUIRE(d == PRIM_FILTER_D, "mst_prim_lazy: points must have 64 float32 coordinates");
    // lazy_step_kernel looks at a thread's PRIM_AHEAD points and has no strided tail (prim_step_kernel's loop behind the look-ahead):
    // every position must fall inside grid x PRIM_NT x PRIM_AHEAD, or points past it would never become candidates
    IDL_REQUIRE(n >= 65536 && n <= (int64_t)prim_grid(n) * PRIM_NT * PRIM_AHEAD, "mst_prim_lazy: 65536 <= n <= 2^20 points (use idl_mst_prim_local beyond)");
    IDL_REQUIRE(n_groups >= 1 && n_groups <= prim_grid(n), "mst_prim_lazy: more groups than workgroups of a step");
    IDL_REQUIRE((((uintptr_t)workspace) & 255u) == 0 && start >= 0 && start < n, "mst_prim_lazy: workspace alignment / start position");
    const LazyLayout l = lazy_layout(n, n_groups);
    unsigned char *w = (unsigned char *)workspace;
    PrimArgs a{};
    a.xt = xt; a.core = core; a.n = n; a.d = d; a.mst_cur = mst_cur; a.mst_next = mst_next; a.mst_w = mst_w;
    a.orig = orig; a.start = start; a.codes = codes; a.resid = resid; a.gid = gid; a.glo = glo; a.gscale = gscale;
    a.min_reach = (double *)(w + l.min_reach); a.source = (int64_t *)(w + l.source);
    a.cand[0] = (Cand *)(w + l.cand0); a.cand[1] = (Cand *)(w + l.cand1);
    LazyArgs z{};
    z.st = (LazyState *)(w + l.st); z.tree_p = (int64_t *)(w + l.tree_p); z.run_asleep = w + l.run_asleep; z.pas = w + l.pas;
    z.n_groups = n_groups; z.gfirst = gfirst; z.asleep = (int *)(w + l.asleep); z.upto = (int64_t *)(w + l.upto);
    z.minmr = (double *)(w + l.minmr); z.lbp = (double *)(w + l.lbp); z.gc = gcentre; z.gr = gradius; z.xrow = xrow;
    z.gmin = (double *)(w + l.gmin); z.galive = (int64_t *)(w + l.galive);
    z.wg_list = (int *)(w + l.wg_list); z.full_grid = prim_grid(n);
    const hipStream_t st = (hipStream_t)stream;
    const int grid = prim_grid(n);
    int step_grid = grid;                                    // lazy_step_kernel's: the listed workgroups (lazy_list_kernel)
    const unsigned runs = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(lazy_init_kernel, dim3(256), dim3(256), 0, st, a, z);
    auto step = [&](int64_t ln, int rescan) { hipLaunchKernelGGL(lazy_step_kernel, dim3(step_grid), dim3(PRIM_NT), 0, st, a, z, ln, rescan); };
    auto relist = [&]() -> int {                             // the flags changed: the step's grid from the device's list
        hipLaunchKernelGGL(lazy_list_kernel, dim3(1), dim3(1024), 0, st, a, z);
        int L = 0;
        IDL_HIP_TRY(hipMemcpyAsync(&L, z.wg_list, sizeof(int), hipMemcpyDeviceToHost, st));
        IDL_HIP_TRY(hipStreamSynchronize(st));
        IDL_REQUIRE(L >= 1 && L <= grid, "mst_prim_lazy: workgroup list out of range");
        step_grid = L;
        return IDL_OK;
    };
    int64_t launch = 0, stalls = 0, censuses = 0;
    if (int rc = relist()) return rc;
    step(launch, 1); ++launch;       // the first scan: cur = the start
    // the steps are queued in chunks; after each the host looks at the state: done, stalled, or time for a census
    static const int chunk = idl::dev_env("mst_chunk") ? atoi(idl::dev_env("mst_chunk")) : 512;
    static const int64_t census_every = idl::dev_env("mst_census") ? atoll(idl::dev_env("mst_census")) : 16384;
    static const bool sleep_on = !(idl::dev_env("mst_sleep") && atoi(idl::dev_env("mst_sleep")) == 0);
    int64_t last_census = 0, last_stall_at = -1000000, quick = 0, no_sleep_until = 0;
    LazyState S{};
    for (;;) {
        for (int i = 0; i < chunk; ++i) { step(launch, 0); ++launch; }
        IDL_HIP_TRY(hipMemcpyAsync(&S, z.st + (launch & 1), sizeof(S), hipMemcpyDeviceToHost, st));
        IDL_HIP_TRY(hipStreamSynchronize(st));
        if (S.n_tree >= n) break;
        if (launch > 3 * n + (1 << 22)) { idl::set_error("mst_prim_lazy: no progress (%lld nodes after %lld launches)", (long long)S.n_tree, (long long)launch); return IDL_ERR_HIP; }
        const bool first = last_census == 0 && S.n_tree >= 1024;
        if (!(S.stalled || first || S.n_tree - (last_census ? last_census : 1) >= census_every)) continue;
        if (S.stalled) {
            ++stalls;
            if (S.n_tree - last_stall_at < 256) { if (++quick > 16) { no_sleep_until = S.n_tree + 32768; quick = 0; } } else quick = 0;
            last_stall_at = S.n_tree;
        }
        ++censuses;
        static const bool dbg = idl::dev_env("mst_debug") != nullptr;
        if (dbg && (censuses < 60 || censuses % 100 == 0))
            fprintf(stderr, "[idl] lazy prim: census %lld at launch %lld: n_tree %lld stalled %d cur_w %.6g ema %.6g quick %lld no_sleep_until %lld\n",
                    (long long)censuses, (long long)launch, (long long)S.n_tree, S.stalled, S.cur_w, S.ema, (long long)quick, (long long)no_sleep_until);
        hipLaunchKernelGGL(lazy_catchup_kernel, dim3(runs), dim3(256), 0, st, a, z, (int64_t)S.n_tree);
        hipLaunchKernelGGL(lazy_census_kernel, dim3((unsigned)n_groups), dim3(256), 0, st, a, z);
        hipLaunchKernelGGL(lazy_policy_kernel, dim3(1), dim3(256), 0, st, a, z, (int)(launch & 1), (int64_t)S.n_tree,
                           (sleep_on && S.n_tree >= no_sleep_until) ? 1 : 0);
        hipLaunchKernelGGL(lazy_flags_kernel, dim3(runs), dim3(256), 0, st, a, z);
        if (int rc = relist()) return rc;
        if (S.stalled) { S.stalled = 0; IDL_HIP_TRY(hipMemcpyAsync(z.st + (launch & 1), &S, sizeof(S), hipMemcpyHostToDevice, st)); IDL_HIP_TRY(hipStreamSynchronize(st)); }
        step(launch, 1); ++launch;   // candidates of the awake set
        last_census = S.n_tree;
    }
    IDL_HIP_TRY(hipGetLastError());
    if (stats3) { stats3[0] = launch; stats3[1] = stalls; stats3[2] = censuses; }
    return IDL_OK;
}

}  // extern "C"
