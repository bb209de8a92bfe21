// knn.hip -- core distances of the fine-grained mode's HDBSCAN (reference idelucs/__main__.py:83,153-156:
// hdbscan.HDBSCAN(min_cluster_size = N // 100 + 1) on the [N, 64] latent; the stand-in sklearn.cluster.HDBSCAN takes min_samples =
// min_cluster_size): for every point the distance to its k-th nearest neighbour, itself included, k = N // 100 + 1 -- at BASELINE
// cfg5 (N = 10^6) the 10 001-th smallest of 10^6 distances, for 10^6 points.
//
// Round 2 materialised the float64 Gram-form distance matrix in 8 GB row blocks and ran torch.topk on each: 8.6 TB written and
// radix-selected, 29 of the device HDBSCAN's 73 s.  Here the distances are never written:
//
//   idl_knn_window   ONE pass over all pairs on the fp32 matrix cores.  The caller gives every row a window [lo, hi) that brackets
//                    its k-th distance (order statistics of a column sample); the pass counts the columns below lo and keeps the
//                    few thousand inside the window (squared distance + index) in a per-row slot.  A workgroup owns 256 rows (four
//                    waves x four 16-row A tiles held in registers) and streams every 16-column B tile once: 64 MFMAs per 4 KB
//                    tile per wave -- matrix-pipe bound; the 256 MB of points are read N / 256 times from L2 / Infinity Cache.
//                    The Gram form |a|^2 + |b|^2 - 2 a.b loses what the norms exceed the distance by, and a latent of tight,
//                    far-apart clusters has norms 10^3 x its neighbour distances: every wave therefore works in coordinates
//                    centred on ITS first row (b - m costs 16 subtractions per lane and tile).  The caller orders the points so
//                    that neighbours in memory are neighbours in space; then |a - m| is a cluster diameter, the columns that can
//                    fall under hi have |b - m| <= |a - m| + sqrt(hi), and the rounding bound (`delta`, written per row) is a
//                    few 1e-5 of the distances that matter -- whatever the far columns' norms are.
//   idl_knn_select   per row: radix select (4 x 8 bits, LDS histograms) of the (k - below)-th smallest kept value v; then EXACTLY:
//                    the Gram form in fp32 errs by at most eps = delta / 2, so every column kept below v - delta is truly below
//                    the k-th and every true candidate lies within v +- delta; the members of that band (a handful) get their
//                    float64 distance from the difference vector, in sklearn's order of operations, and the k-th is read off
//                    them.  A row whose window missed (sample unlucky, band touching the window's edge, slot or band full) is
//                    flagged and recomputed by the caller's exact path.
#include "common.h"
#include "wave_ops.h"

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int KD = 64;            // point dimension (the latent width of NetLinear / myNet)
constexpr int ROWS_WG = 256;      // rows of a workgroup: 4 waves x 64
constexpr int BAND_CAP = 1024;    // columns of the exact band a row can hold

struct WindowArgs {
    const float *x;               // [n, 64] the points, ordered so that neighbours in memory are neighbours in space
    const float *lo, *hi;         // per row of this launch (index row - row0)
    int64_t n, row0, rows;
    int32_t *cnt_lo, *cand_cnt;   // per row of this launch
    float *delta;                 // per row of this launch: twice the rounding bound of the values this pass computed for it
    float *cand_d2; int32_t *cand_idx;   // [rows, cap]
    int cap;
};

// |computed - true squared distance| <= GRAM_ERR * (|a - m|^2 + |b - m|^2): the centring's rounding (4 u), the two norms (64 u each
// way at worst), the 64-term product sum on the matrix cores (64 u), the final sums and the window comparison done against
// lo - |a|^2 (8 u); u = 2^-23 leaves a factor two for the matrix cores' internal rounding.
constexpr float GRAM_ERR = 144.0f * 1.1920929e-7f;

// (Round 4, measured and NOT adopted: staging a row's kept columns in an LDS ring and storing them eight at a time, 2 x 32 contiguous
//  bytes instead of two scattered 4-byte stores per column -- the 0.9 TB of sector traffic at cfg5 was taken for the gap between
//  0.38 and 0.60 of the MFMA peak.  It is not: 2.80 s per pass at 10^6 points against 2.19 s, same rows, bit-identical results.  fp32
//  MFMA issues at the vector rate, so the two ds_write + the flush branch per kept column cost matrix time, while a global store is
//  one VMEM issue that the matrix pipe does not wait for; and 64 KB of LDS halves the workgroups per CU.)
__global__ __launch_bounds__(256) void knn_window_kernel(WindowArgs a)
{
    __shared__ int slot_n[ROWS_WG];
    __shared__ float centre[4][KD];       // per wave: its first row
    __shared__ float sq_row[ROWS_WG];     // |a - m|^2 per row
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l = lane & 15, q = lane >> 4;
    const int64_t last = a.row0 + a.rows - 1;
    const int64_t rbase = a.row0 + (int64_t)blockIdx.x * ROWS_WG + 64 * wv;     // first row of this wave
    slot_n[tid] = 0;
    centre[wv][lane] = a.x[(rbase <= last ? rbase : last) * KD + lane];
    __syncthreads();
    // A tiles: row rbase + 16 rt + l, k = 16 q + s  (A[i = l][k = q] per MFMA step s: the k order is permuted the same way on the B side)
    float av[4][16];
    const float *mv = &centre[wv][16 * q];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const int64_t r = rbase + 16 * rt + l;
        const float4 *src = (const float4 *)(a.x + (r <= last ? r : last) * KD + 16 * q);
        float part = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 t = src[i];
            av[rt][4 * i] = t.x - mv[4 * i]; av[rt][4 * i + 1] = t.y - mv[4 * i + 1]; av[rt][4 * i + 2] = t.z - mv[4 * i + 2]; av[rt][4 * i + 3] = t.w - mv[4 * i + 3];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) part = fmaf(av[rt][s], av[rt][s], part);
        part += __shfl_xor(part, 16, 64); part += __shfl_xor(part, 32, 64);
        if (q == 0) sq_row[64 * wv + 16 * rt + l] = part;
    }
    __syncthreads();
    // per C/D element (row 4 q + reg of tile rt, column l): the window with the row's norm taken off, and the count below it
    float lo_m[4][4], hi_m[4][4];
    int below[4][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int rl = 64 * wv + 16 * rt + 4 * q + reg;
            const int64_t rr = a.row0 + (int64_t)blockIdx.x * ROWS_WG + rl;
            const bool okr = rr <= last;
            const float sqa = sq_row[rl];
            const float lo = okr ? a.lo[rr - a.row0] : -1.f, hi = okr ? a.hi[rr - a.row0] : -1.f;     // a row past the end: nothing below, nothing inside
            lo_m[rt][reg] = lo - sqa; hi_m[rt][reg] = hi - sqa;
            below[rt][reg] = 0;
            if (okr && l == 0) {
                const float reach = sqrtf(sqa) + sqrtf(fmaxf(hi, 0.f));                              // |b - m| of any column that can come out under hi
                a.delta[rr - a.row0] = 2.f * GRAM_ERR * (sqa + 1.001f * reach * reach);
            }
        }
    // B tile t: columns 16 t + l, k = 16 q + s
    const int64_t ntile = (a.n + 15) / 16;
    auto load_b = [&](int64_t t, float4 (&raw)[4]) {
        const int64_t j = 16 * t + l;
        const float4 *src = (const float4 *)(a.x + (j < a.n ? j : a.n - 1) * KD + 16 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) raw[i] = src[i];
    };
    float4 r0[4], r1[4];
    load_b(0, r0);
    auto tile = [&](int64_t t, const float4 (&raw)[4]) {
        float bv[16];
        float sqc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bv[4 * i] = raw[i].x - mv[4 * i]; bv[4 * i + 1] = raw[i].y - mv[4 * i + 1]; bv[4 * i + 2] = raw[i].z - mv[4 * i + 2]; bv[4 * i + 3] = raw[i].w - mv[4 * i + 3];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) sqc = fmaf(bv[s], bv[s], sqc);
        sqc += __shfl_xor(sqc, 16, 64); sqc += __shfl_xor(sqc, 32, 64);
        const int j = (int)(16 * t) + l;
        if (j >= a.n) sqc = 3.0e38f;                             // a column past the end is infinitely far
        f32x4_t acc[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][s], bv[s], acc[rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const float t2 = sqc - 2.f * acc[rt][reg];                             // squared distance less the row's norm
                below[rt][reg] += t2 < lo_m[rt][reg] ? 1 : 0;
                if (t2 >= lo_m[rt][reg] && t2 < hi_m[rt][reg]) {
                    const int rl = 64 * wv + 16 * rt + 4 * q + reg;                    // row inside the workgroup
                    const int pos = atomicAdd(&slot_n[rl], 1);
                    if (pos < a.cap) {
                        const int64_t o = ((int64_t)blockIdx.x * ROWS_WG + rl) * a.cap + pos;
                        a.cand_d2[o] = t2 + sq_row[rl]; a.cand_idx[o] = j;
                    }
                }
            }
    };
    for (int64_t t = 0; t < ntile; t += 2) {                   // two tiles per turn: the next tile's loads are in flight during the MFMAs
        if (t + 1 < ntile) load_b(t + 1, r1);
        tile(t, r0);
        if (t + 1 < ntile) {
            if (t + 2 < ntile) load_b(t + 2, r0);
            tile(t + 1, r1);
        }
    }
    // counts: this lane's rows 4 q + reg of tile rt, over its columns l -> add over the 16 lanes of the row
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            int v = below[rt][reg];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            const int64_t rr = rbase + 16 * rt + 4 * q + reg;
            if (l == 0 && rr <= last) a.cnt_lo[rr - a.row0] = v;
        }
    __syncthreads();
    {
        const int64_t rr = a.row0 + (int64_t)blockIdx.x * ROWS_WG + tid;
        if (rr <= last) a.cand_cnt[rr - a.row0] = slot_n[tid];
    }
}

struct SelectArgs {
    const float *x;                      // [n, 64] the points themselves (float64 values that float32 holds exactly)
    const float *lo, *hi, *delta;        // per row of this launch
    int64_t n, row0, rows, k;
    const int32_t *cnt_lo, *cand_cnt;
    const float *cand_d2; const int32_t *cand_idx;
    int cap;
    double *core;                        // [n]
    int32_t *status;                     // per row of this launch: 0 done, else why not (1 window missed, 2 slot full, 3 band at the edge, 4 band full)
};

__device__ __forceinline__ uint32_t fkey(float f)        // order-preserving map float -> uint32
{
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ __launch_bounds__(256) void knn_select_kernel(SelectArgs a)
{
    __shared__ int hist[256];
    __shared__ int sh_bin, sh_rank, band_n, below_band, wave_tot[4];
    __shared__ int band_idx[BAND_CAP];
    __shared__ double band_val[BAND_CAP];
    const int tid = threadIdx.x;
    const int64_t rloc = blockIdx.x, row = a.row0 + rloc;
    const int cnt = a.cand_cnt[rloc];
    const int64_t want = a.k - (int64_t)a.cnt_lo[rloc];        // 1-based rank among the kept columns
    if (cnt > a.cap) { if (tid == 0) a.status[rloc] = 2; return; }
    if (want < 1 || want > cnt) { if (tid == 0) a.status[rloc] = 1; return; }
    const float *d2 = a.cand_d2 + rloc * a.cap;
    const int32_t *ix = a.cand_idx + rloc * a.cap;
    // ---- radix select: the `want`-th smallest key, 8 bits per pass from the top.  Keys are taken relative to the window's lower
    // end, so that only the bits in which the kept values differ are walked (a window spans ~2^20 floats: three passes, and the
    // first histogram is spread over its bins instead of piling every lane's atomic on one)
    const float lo_f = a.lo[rloc];
    const uint32_t kbase = fkey(lo_f);
    const uint32_t span = fkey(a.hi[rloc]) - kbase;
    const int passes = span ? (32 - __builtin_clz(span) + 7) / 8 : 1;
    uint32_t prefix = 0, mask = 0;
    int rank = (int)want - 1;                                   // 0-based rank among the keys that match the prefix
    for (int pass = 0; pass < passes; ++pass) {
        const int shift = 8 * (passes - 1 - pass);
        hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < cnt; i += 256) {
            // (a kept value is stored as t2 + |a - m|^2 AFTER the window test on t2: rounding can leave it an ulp below lo or at hi.
            //  Clamped into [lo, hi] before keying -- unclamped, key - kbase would wrap and the value be binned as the largest, or alias
            //  under the walked bits, shifting the selected rank: ADVICE r3.  A selection that lands on a clamped end fails the
            //  edge test below and the row goes to the exact path.)
            uint32_t key = fkey(fmaxf(d2[i], lo_f)) - kbase;
            key = key > span ? span : key;
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
        }
        __syncthreads();
        {   // the bin that holds rank `rank`: block-wide inclusive prefix of the histogram (wave scan + the wave totals)
            const int h = hist[tid];
            int inc = h;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if ((tid & 63) >= o) inc += t; }
            if ((tid & 63) == 63) wave_tot[tid >> 6] = inc;
            __syncthreads();
            int base = 0;
            for (int w = 0; w < (tid >> 6); ++w) base += wave_tot[w];
            inc += base;
            if (rank >= inc - h && rank < inc) { sh_bin = tid; sh_rank = rank - (inc - h); }
        }
        __syncthreads();
        prefix |= (uint32_t)sh_bin << shift; mask |= 255u << shift; rank = sh_rank;
        __syncthreads();
    }
    // prefix + kbase is the key of v; recover v
    const uint32_t vkey = prefix + kbase;
    const uint32_t vb = (vkey & 0x80000000u) ? (vkey & 0x7FFFFFFFu) : ~vkey;
    const float v = __uint_as_float(vb);
    const float dl = a.delta[rloc];
    if (!(v - dl > a.lo[rloc]) || !(v + dl < a.hi[rloc])) { if (tid == 0) a.status[rloc] = 3; return; }
    // ---- the band [v - delta, v + delta]: members and the count of kept columns below it
    if (tid == 0) { band_n = 0; below_band = 0; }
    __syncthreads();
    int mine_below = 0;
    for (int i = tid; i < cnt; i += 256) {
        const float x = d2[i];
        if (x < v - dl) ++mine_below;
        else if (x <= v + dl) { const int p = atomicAdd(&band_n, 1); if (p < BAND_CAP) band_idx[p] = ix[i]; }
    }
    if (mine_below) atomicAdd(&below_band, mine_below);
    __syncthreads();
    const int nb = band_n;
    if (nb > BAND_CAP) { if (tid == 0) a.status[rloc] = 4; return; }
    // ---- exact squared distances of the band members, float64 from the difference vector (sklearn: t = a - b; d += t * t)
    for (int p = tid; p < nb; p += 256) {
        const float *pa = a.x + row * KD, *pb = a.x + (int64_t)band_idx[p] * KD;
        double d = 0.0;
        for (int c = 0; c < KD; ++c) { const double t = (double)pa[c] - (double)pb[c]; d = idl_dev::square_then_add(d, t); }
        band_val[p] = d;
    }
    __syncthreads();
    const int target = (int)want - 1 - below_band;              // 0-based rank inside the band
    if (target < 0 || target >= nb) { if (tid == 0) a.status[rloc] = 3; return; }
    for (int p = tid; p < nb; p += 256) {
        const double mv = band_val[p];
        int r = 0;
        for (int o = 0; o < nb; ++o) { const double ov = band_val[o]; r += (ov < mv || (ov == mv && o < p)) ? 1 : 0; }
        if (r == target) { a.core[row] = __dsqrt_rn(mv); a.status[rloc] = 0; }
    }
}

// ---------------------------------------------------------------- silhouette: per-cluster distance sums (posthoc.compute_results)
// sklearn.metrics.silhouette_score needs, for every point, the sum of its distances to the points of every cluster: N^2 distances,
// each used once.  Round 2 formed them in [4096 x N] float32 blocks with a GEMM and five elementwise passes (40 TB of traffic at
// 10^6 points, 10 s); this is idl_knn_window's pass with another epilogue.  The caller orders the points by cluster and pads every
// cluster to whole 16-column tiles (weight 0 on the padding): a tile then belongs to ONE cluster, the distances add up in 16
// registers per lane, and those are written out -- once, by their only owner -- when the tile's cluster changes.
struct SilArgs {
    const float *x;               // [n, 64] the points, cluster by cluster, padded
    const float *w;               // [n] 1 for a point, 0 for padding
    const int32_t *tile_cluster;  // [n / 16] cluster of each 16-column tile
    int64_t n; int n_clusters;
    float *sums;                  // [n, n_clusters] sum over the cluster's points of the distance to them
};

__global__ __launch_bounds__(256) void silhouette_sums_kernel(SilArgs a)
{
    __shared__ float centre[4][KD];
    __shared__ float sq_row[ROWS_WG];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l = lane & 15, q = lane >> 4;
    const int64_t last = a.n - 1;
    const int64_t rbase = (int64_t)blockIdx.x * ROWS_WG + 64 * wv;
    centre[wv][lane] = a.x[(rbase <= last ? rbase : last) * KD + lane];
    __syncthreads();
    float av[4][16];
    const float *mv = &centre[wv][16 * q];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const int64_t r = rbase + 16 * rt + l;
        const float4 *src = (const float4 *)(a.x + (r <= last ? r : last) * KD + 16 * q);
        float part = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 t = src[i];
            av[rt][4 * i] = t.x - mv[4 * i]; av[rt][4 * i + 1] = t.y - mv[4 * i + 1]; av[rt][4 * i + 2] = t.z - mv[4 * i + 2]; av[rt][4 * i + 3] = t.w - mv[4 * i + 3];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) part = fmaf(av[rt][s], av[rt][s], part);
        part += __shfl_xor(part, 16, 64); part += __shfl_xor(part, 32, 64);
        if (q == 0) sq_row[64 * wv + 16 * rt + l] = part;
    }
    __syncthreads();
    float sqa[4][4], tot[4][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) { sqa[rt][reg] = sq_row[64 * wv + 16 * rt + 4 * q + reg]; tot[rt][reg] = 0.f; }
    const int64_t ntile = a.n / 16;
    auto load_b = [&](int64_t t, float4 (&raw)[4], float &wj) {
        const int64_t j = 16 * t + l;
        const float4 *src = (const float4 *)(a.x + j * KD + 16 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) raw[i] = src[i];
        wj = a.w[j];
    };
    auto flush = [&](int c) {                                  // the sums of cluster c are complete: add up the 16 column lanes, write, clear
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                float v = tot[rt][reg];
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                const int64_t rr = rbase + 16 * rt + 4 * q + reg;
                if (l == 0 && rr <= last) a.sums[rr * a.n_clusters + c] = v;
                tot[rt][reg] = 0.f;
            }
    };
    int cur_c = a.tile_cluster[0];
    auto tile = [&](int64_t t, const float4 (&raw)[4], float wj) {
        const int c = a.tile_cluster[t];
        if (c != cur_c) { flush(cur_c); cur_c = c; }
        float bv[16];
        float sqc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bv[4 * i] = raw[i].x - mv[4 * i]; bv[4 * i + 1] = raw[i].y - mv[4 * i + 1]; bv[4 * i + 2] = raw[i].z - mv[4 * i + 2]; bv[4 * i + 3] = raw[i].w - mv[4 * i + 3];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) sqc = fmaf(bv[s], bv[s], sqc);
        sqc += __shfl_xor(sqc, 16, 64); sqc += __shfl_xor(sqc, 32, 64);
        f32x4_t acc[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][s], bv[s], acc[rt], 0, 0, 0);
        const bool own = 16 * t + 15 >= rbase && 16 * t < rbase + 64;      // the tile holds some of this wave's own points: d(i, i) = 0 exactly
        const int64_t j = 16 * t + l;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                float d = __builtin_amdgcn_sqrtf(fmaxf((sqa[rt][reg] + sqc) - 2.f * acc[rt][reg], 0.f)) * wj;
                if (own && j == rbase + 16 * rt + 4 * q + reg) d = 0.f;
                tot[rt][reg] += d;
            }
    };
    float4 r0[4], r1[4];
    float w0, w1;
    load_b(0, r0, w0);
    for (int64_t t = 0; t < ntile; t += 2) {
        if (t + 1 < ntile) load_b(t + 1, r1, w1);
        tile(t, r0, w0);
        if (t + 1 < ntile) {
            if (t + 2 < ntile) load_b(t + 2, r0, w0);
            tile(t + 1, r1, w1);
        }
    }
    flush(cur_c);
}

}  // namespace

extern "C" {

int idl_knn_window(const float *x, int64_t n, int d, const float *lo, const float *hi, int64_t row0, int64_t rows, int32_t *cnt_lo,
                   int32_t *cand_cnt, float *delta, float *cand_d2, int32_t *cand_idx, int cap, void *stream)
{
    IDL_REQUIRE(x && lo && hi && cnt_lo && cand_cnt && delta && cand_d2 && cand_idx, "knn_window: NULL buffer");
    IDL_REQUIRE(d == KD, "knn_window: points must have 64 coordinates");
    IDL_REQUIRE(n >= 1 && n < (1ll << 31) && row0 >= 0 && rows >= 1 && row0 + rows <= n && cap >= 1, "knn_window: bad sizes");
    IDL_REQUIRE((((uintptr_t)x) & 15u) == 0, "knn_window: x must be 16-byte aligned");
    IDL_REQUIRE((rows + ROWS_WG - 1) / ROWS_WG * (int64_t)ROWS_WG * cap < (1ll << 40), "knn_window: slot buffer too large");
    WindowArgs a{x, lo, hi, n, row0, rows, cnt_lo, cand_cnt, delta, cand_d2, cand_idx, cap};
    hipLaunchKernelGGL(knn_window_kernel, dim3((unsigned)((rows + ROWS_WG - 1) / ROWS_WG)), dim3(256), 0, (hipStream_t)stream, a);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

int idl_knn_select(const float *x, int64_t n, int d, const float *lo, const float *hi, const float *delta, int64_t row0,
                   int64_t rows, int64_t k, const int32_t *cnt_lo, const int32_t *cand_cnt, const float *cand_d2, const int32_t *cand_idx,
                   int cap, double *core, int32_t *status, void *stream)
{
    IDL_REQUIRE(x && lo && hi && delta && cnt_lo && cand_cnt && cand_d2 && cand_idx && core && status, "knn_select: NULL buffer");
    IDL_REQUIRE(d == KD, "knn_select: points must have 64 coordinates");
    IDL_REQUIRE(n >= 1 && row0 >= 0 && rows >= 1 && row0 + rows <= n && cap >= 1 && k >= 1 && k <= n, "knn_select: bad sizes");
    SelectArgs a{x, lo, hi, delta, n, row0, rows, k, cnt_lo, cand_cnt, cand_d2, cand_idx, cap, core, status};
    hipLaunchKernelGGL(knn_select_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, a);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

int idl_silhouette_sums(const float *x, const float *w, const int32_t *tile_cluster, int64_t n, int d, int n_clusters, float *sums, void *stream)
{
    IDL_REQUIRE(x && w && tile_cluster && sums, "silhouette_sums: NULL buffer");
    IDL_REQUIRE(d == KD, "silhouette_sums: points must have 64 coordinates");
    IDL_REQUIRE(n >= 16 && n % 16 == 0 && n < (1ll << 31) && n_clusters >= 1, "silhouette_sums: n must be a positive multiple of 16 (clusters padded to whole tiles)");
    IDL_REQUIRE((((uintptr_t)x) & 15u) == 0, "silhouette_sums: x must be 16-byte aligned");
    SilArgs a{x, w, tile_cluster, n, n_clusters, sums};
    hipLaunchKernelGGL(silhouette_sums_kernel, dim3((unsigned)((n + ROWS_WG - 1) / ROWS_WG)), dim3(256), 0, (hipStream_t)stream, a);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

}  // extern "C"
