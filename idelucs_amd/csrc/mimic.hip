// mimic.hip -- "fast mode" mimic-mutation generator for gfx950: substitution edits drawn on the
// device from a counter-based RNG, in the edit format the vectoriser consumes.
//
// What it replaces: the site selection of idelucs/utils.py:54-135 -- transition(p): every base
// independently w.p. p (utils.py:69-70); transversion(p): likewise, target chosen uniformly between
// the two transversions (utils.py:111-118); transition_transversion: both, sequentially
// (utils.py:131-135); Random_N(n): n positions uniform with replacement (utils.py:93).
// The reference draws these from numpy's global MT19937 and Python's `random`; that stream cannot
// be reproduced on a GPU, so bit-parity with the reference lives in the host-RNG "compat" path
// (idelucs_amd/utils.py) and THIS generator is statistically equivalent: identical per-base site
// probabilities and target distributions, checked bit-for-bit against oracle/idelucs_oracle.c's
// restatement of the spec below and statistically against the reference's rates.
//
// Spec (one wavefront per (sequence, view); lane l owns bases [l*seg, min(L,(l+1)*seg)), seg = ceil(L/64)):
//   site views   q = 1-(1-p_ts)(1-p_tv).  Sites are placed by exact geometric gap sampling: draw
//                r = Philox4x32-10(key = seed, ctr = (draw#, lane, seq, view)); gap-1 = the largest
//                j <= J with r.x < T[j], T[j] = floor((1-q)^j * 2^32) (T[0] = +inf, J = 1024;
//                j == J means "no site in the next J bases", resampled -- the geometric law is
//                memoryless, so both the per-lane restart and the tail cut are exact).  Site type
//                from r.y: < A transition only, < B transversion only, else both;
//                transversion flavour = r.z & 1.  op = 2 | 1,3 | 3,1  (XOR on A0 C1 G2 T3).
//   Random_N     lane i < n: pos = mulhi(Philox(ctr = (i, 0, seq, view | 1<<16)).x, L); positions are
//                wave-sorted ascending (duplicates kept); op = 0 (N).  Nothing is emitted for L = 0.
// Two passes (count, exclusive scan, fill) so that the caller sizes `edits` exactly.
#include "common.h"
#include <math.h>

namespace {

constexpr int J = 1024;            // gap table length
constexpr int MAX_VIEWS = 64;

struct MimicParams {
    double one_minus_q[MAX_VIEWS];
    uint32_t thr_ts_only[MAX_VIEWS];   // A
    uint32_t thr_tv_only[MAX_VIEWS];   // B (cumulative)
    int32_t n_rand[MAX_VIEWS];
    uint8_t has_sites[MAX_VIEWS];
    uint8_t kind[MAX_VIEWS];           // 0 mixed (use A/B), 1 transition only, 2 transversion only
    float inv_log2_keep[MAX_VIEWS];    // 1 / log2(1 - q): first guess of the gap, j ~ log2(r / 2^32) / log2(1 - q)
};

struct U4 { uint32_t x, y, z, w; };

__host__ __device__ inline U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

// T[v][0..J]: T[0] unused (+inf), T[j] = floor((1-q)^j * 2^32) by repeated float64 multiplication
__global__ void mimic_table_kernel(MimicParams p, int n_views, uint32_t *tables)
{
    const int v = blockIdx.x;
    if (threadIdx.x != 0 || v >= n_views) return;
    uint32_t *T = tables + (size_t)v * (J + 1);
    T[0] = 0xFFFFFFFFu;
    if (!p.has_sites[v]) return;
    double x = 1.0;
    for (int j = 1; j <= J; ++j) {
        x = x * p.one_minus_q[v];
        T[j] = (uint32_t)(x * 4294967296.0);
    }
}

__device__ __forceinline__ uint32_t wave_sort_u32(uint32_t key, int lane)
{
    // bitonic sort across the 64 lanes, ascending
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const uint32_t other = __shfl_xor(key, j, 64);
            const bool up = ((lane & k) == 0);
            const bool lower = ((lane & j) == 0);
            const uint32_t mn = key < other ? key : other, mx = key < other ? other : key;
            key = (lower == up) ? mn : mx;
        }
    }
    return key;
}

template <bool FILL>
__global__ __launch_bounds__(64) void mimic_kernel(MimicParams p, int n_views, const int64_t *lengths, int64_t n, uint32_t k0,
                                                   uint32_t k1, const uint32_t *tables, int64_t *counts_or_off, uint32_t *edits,
                                                   int64_t capacity, uint16_t *lane_counts)
{
    __shared__ uint32_t T[J + 1];
    const int lane = threadIdx.x;
    const int64_t items = n * n_views;
    int cur_view = -1;
    for (int64_t it = blockIdx.x; it < items; it += gridDim.x) {
        // view-major items keep the table resident: item = v * n + s
        const int v = (int)(it / n);
        const int64_t s = it - (int64_t)v * n;
        const int64_t L = lengths[s];
        uint32_t my_count = 0;
        int64_t base = 0;
        if (FILL) base = counts_or_off[it];

        if (p.n_rand[v] > 0) {
            uint32_t key = 0xFFFFFFFFu;
            const int nr = L > 0 ? p.n_rand[v] : 0;
            if (lane < nr) {
                const U4 r = philox4x32_10((uint32_t)lane, 0u, (uint32_t)s, (uint32_t)v | (1u << 16), k0, k1);
                key = (uint32_t)(((uint64_t)r.x * (uint64_t)(uint32_t)L) >> 32);
            }
            key = wave_sort_u32(key, lane);
            if (FILL) { if (lane < nr && base + lane < capacity) edits[base + lane] = key; }       // op 0 = N
            else if (lane == 0) counts_or_off[it] = nr;
            continue;
        }
        if (!p.has_sites[v]) {
            if (!FILL && lane == 0) counts_or_off[it] = 0;
            continue;
        }
        if (cur_view != v) {
            __syncthreads();
            for (int j = lane; j <= J; j += 64) T[j] = tables[(size_t)v * (J + 1) + j];
            __syncthreads();
            cur_view = v;
        }
        const int64_t seg = (L + 63) / 64;
        const int64_t lo = (int64_t)lane * seg;
        int64_t hi = lo + seg;
        if (hi > L) hi = L;
        const uint32_t A = p.thr_ts_only[v], B = p.thr_tv_only[v];
        const int kind = p.kind[v];
        const float ilk = p.inv_log2_keep[v];

        // in FILL mode each lane needs its output offset: the per-lane counts were saved by the count pass
        uint32_t lane_off = 0;
        if (FILL) {
            const uint32_t c = lane_counts[it * 64 + lane];
            uint32_t incl = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
            lane_off = incl - c;
        }
        {
            int64_t pos = lo - 1;
            for (uint32_t d = 0; pos < hi; ++d) {
                const U4 r = philox4x32_10(d, (uint32_t)lane, (uint32_t)s, (uint32_t)v, k0, k1);
                // the largest a in [0, J] with r.x < T[a] (T[0] = +inf, T decreasing): a float guess, then exact steps on the
                // integer table -- the same a the binary search finds, in ~2 LDS reads instead of 10 dependent ones
                int a = (int)fminf(__log2f(((float)r.x + 0.5f) * 2.3283064365386963e-10f) * ilk, (float)J);
                if (a < 0) a = 0;
                while (a < J && r.x < T[a + 1]) ++a;
                while (a > 0 && !(r.x < T[a])) --a;
                if (a == J) { pos += J; continue; }
                pos += a + 1;
                if (pos >= hi) break;
                if (FILL) {
                    const uint32_t flav = 1u | ((r.z & 1u) << 1);        // transversion: ^1 or ^3
                    uint32_t op = (r.y < A) ? 2u : (r.y < B) ? flav : (2u ^ flav);
                    if (kind == 1) op = 2u;
                    if (kind == 2) op = flav;
                    const int64_t o = base + lane_off + my_count;
                    if (o < capacity) edits[o] = (uint32_t)pos | (op << 30);
                }
                ++my_count;
            }
        }
        if (!FILL) {
            lane_counts[it * 64 + lane] = (uint16_t)my_count;      // a lane owns <= ceil(L/64) <= 2^24 bases; sites/lane << 65536 for p < 1
            uint32_t tot = my_count;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
            if (lane == 0) counts_or_off[it] = tot;
        }
    }
}


// ---------------------------------------------------------------- one pass, fixed slots (round 3)
// The two-pass protocol above draws every site TWICE (count, scan, fill: 0.41 + 0.02 + 0.51 ms at cfg2, bound by Philox's
// quarter-rate integer multiplies) only to pack the edits of all (view, sequence) items back to back.  Here item (v, s) owns a
// SLOT of cap[v] edits at base[v] + s * cap[v] -- cap[v] a bound the host derives from the longest sequence and the view's
// site probability -- and the vectoriser is told (begin, end) per item instead of a CSR array (idl_vectorise_ranges): the
// sites are drawn once, into LDS, and copied out behind a wave prefix sum.  Same spec, same sites, same order inside an
// item as mimic_kernel (tests compare the two bit for bit).  An item that does not fit its slot raises *overflow and is
// truncated (the caller falls back to the exact two-pass protocol); a lane with more sites than its LDS buffer holds sends its
// item through a second draw, straight to memory, inside this kernel.

struct SlotLayout { int64_t base[MAX_VIEWS]; int32_t cap[MAX_VIEWS]; };

template <int GB, int RB4>        // GB: items (consecutive sequences of one view) a wave draws at once; RB4: sites a lane buffers per item
__global__ __launch_bounds__(64) void mimic_slots_kernel(MimicParams p, int n_views, const int64_t *lengths, int64_t n, uint32_t k0, uint32_t k1,
                                                         const uint32_t *tables, SlotLayout sl, int64_t *ranges, uint32_t *edits, int32_t *overflow)
{
    __shared__ uint32_t T[J + 1];
    __shared__ uint32_t buf[GB * RB4][64];                 // (RB sites of one item, or RB4 of each of GB items)
    __shared__ uint32_t cnt4[GB][64];
    const int lane = threadIdx.x;
    int cur_view = -1;
    // work units: GB consecutive sequences of one view (the last unit of a view may be shorter); view-major, so the table stays resident
    const int64_t per_view = (n + GB - 1) / GB;
    const int64_t units = per_view * n_views;
    for (int64_t un = blockIdx.x; un < units; un += gridDim.x) {
        const int v = (int)(un / per_view);
        const int64_t s0 = (un - (int64_t)v * per_view) * GB;
        const int nb = (int)((n - s0) < GB ? (n - s0) : GB);
        const uint32_t cap = (uint32_t)sl.cap[v];
        if (p.n_rand[v] > 0) {
            for (int b = 0; b < nb; ++b) {
                const int64_t s = s0 + b, it = (int64_t)v * n + s;
                const int64_t L = lengths[s];
                const int64_t slot = sl.base[v] + s * (int64_t)sl.cap[v];
                uint32_t key = 0xFFFFFFFFu;
                const int nr = L > 0 ? p.n_rand[v] : 0;
                if (lane < nr) {
                    const U4 r = philox4x32_10((uint32_t)lane, 0u, (uint32_t)s, (uint32_t)v | (1u << 16), k0, k1);
                    key = (uint32_t)(((uint64_t)r.x * (uint64_t)(uint32_t)L) >> 32);
                }
                key = wave_sort_u32(key, lane);
                if (lane < nr && (uint32_t)lane < cap) edits[slot + lane] = key;       // op 0 = N
                if (lane == 0) { ranges[2 * it] = slot; ranges[2 * it + 1] = slot + ((uint32_t)nr < cap ? (uint32_t)nr : cap); if ((uint32_t)nr > cap) *overflow = 1; }
            }
            continue;
        }
        if (!p.has_sites[v]) {
            if (lane < nb) { const int64_t s = s0 + lane, it = (int64_t)v * n + s; const int64_t slot = sl.base[v] + s * (int64_t)sl.cap[v]; ranges[2 * it] = slot; ranges[2 * it + 1] = slot; }
            continue;
        }
        if (cur_view != v) {
            __syncthreads();
            for (int j = lane; j <= J; j += 64) T[j] = tables[(size_t)v * (J + 1) + j];
            __syncthreads();
            cur_view = v;
        }
        const uint32_t A = p.thr_ts_only[v], B = p.thr_tv_only[v];
        const int kind = p.kind[v];
        const float ilk = p.inv_log2_keep[v];
        int64_t Lb[GB];
#pragma unroll
        for (int b = 0; b < GB; ++b) Lb[b] = b < nb ? lengths[s0 + b] : 0;
        // one gap of lane `lane` of sequence s: position reached, or -1 for "J bases without a site"; e = the edit when a site was drawn
        auto gap = [&](uint32_t d, int64_t s, int64_t &pos, uint32_t &e) -> bool {
            const U4 r = philox4x32_10(d, (uint32_t)lane, (uint32_t)s, (uint32_t)v, k0, k1);
            int a = (int)fminf(__log2f(((float)r.x + 0.5f) * 2.3283064365386963e-10f) * ilk, (float)J);
            if (a < 0) a = 0;
            while (a < J && r.x < T[a + 1]) ++a;
            while (a > 0 && !(r.x < T[a])) --a;
            if (a == J) { pos += J; return false; }
            pos += a + 1;
            const uint32_t flav = 1u | ((r.z & 1u) << 1);                // transversion: ^1 or ^3
            uint32_t op = (r.y < A) ? 2u : (r.y < B) ? flav : (2u ^ flav);
            if (kind == 1) op = 2u;
            if (kind == 2) op = flav;
            e = (uint32_t)pos | (op << 30);
            return true;
        };
        // ---- the lanes walk their segments of the unit's items one after the other, each at its own pace: a lane that is through with
        // item b starts on item b + 1 while its neighbours still draw -- the wave runs for max-over-lanes of the SUM of draws (a lane
        // draws 3.4 times per item at cfg2, the unluckiest of 64 nine times: 38 % of the lanes busy; over four items 64 %)
        {
            int b = 0;
            uint32_t d = 0, c = 0;
            int64_t L = Lb[0], seg = (L + 63) / 64, lo = (int64_t)lane * seg, hi = lo + seg < L ? lo + seg : L, pos = lo - 1;
            while (b < nb) {
                bool done = !(pos < hi);
                if (!done) {
                    uint32_t e = 0;
                    const bool site = gap(d++, s0 + b, pos, e);
                    if (site && pos < hi) { if (c < (uint32_t)RB4) buf[b * RB4 + c][lane] = e; ++c; }
                    done = site && !(pos < hi);
                }
                if (done) {
                    cnt4[b][lane] = c;
                    ++b; d = 0; c = 0;
                    if (b < nb) {
                        L = Lb[0];
#pragma unroll
                        for (int q = 1; q < GB; ++q) if (b == q) L = Lb[q];
                        seg = (L + 63) / 64; lo = (int64_t)lane * seg; hi = lo + seg < L ? lo + seg : L; pos = lo - 1;
                    }
                }
            }
        }
        // ---- per item: place the lanes' sites behind a wave prefix sum
        for (int b = 0; b < nb; ++b) {
            const int64_t s = s0 + b, it = (int64_t)v * n + s;
            const int64_t slot = sl.base[v] + s * (int64_t)sl.cap[v];
            const uint32_t my = cnt4[b][lane];
            uint32_t incl = my;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
            const uint32_t off = incl - my, total = (uint32_t)__shfl((int)incl, 63, 64);
            if (__ballot(my > (uint32_t)RB4) == 0ull) {
                for (uint32_t i = 0; i < my; ++i) if (off + i < cap) edits[slot + off + i] = buf[b * RB4 + i][lane];
            } else {                                          // a crowded lane somewhere: draw the item again, straight to memory (offsets are known now)
                const int64_t L = Lb[b], seg = (L + 63) / 64, lo = (int64_t)lane * seg, hi = lo + seg < L ? lo + seg : L;
                int64_t pos = lo - 1;
                uint32_t i = 0;
                for (uint32_t d = 0; pos < hi; ++d) {
                    uint32_t e = 0;
                    if (!gap(d, s, pos, e)) continue;
                    if (pos >= hi) break;
                    if (off + i < cap) edits[slot + off + i] = e;
                    ++i;
                }
            }
            if (lane == 0) {
                ranges[2 * it] = slot; ranges[2 * it + 1] = slot + (total < cap ? total : cap);
                if (total > cap) *overflow = 1;
            }
        }
    }
}

// in-place exclusive scan of m int64 counts in three launches: per-block scan (1024 x SCAN_PER elements per block) ->
// scan of the block totals (one workgroup) -> add the block offsets; off[m] = total
constexpr int SCAN_PER = 4;

__global__ __launch_bounds__(1024) void scan_block_kernel(int64_t *off, int64_t m, int64_t *block_tot)
{
    __shared__ int64_t part[1024];
    const int t = threadIdx.x;
    const int64_t a = ((int64_t)blockIdx.x * 1024 + t) * SCAN_PER;
    int64_t v[SCAN_PER], sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_PER; ++i) { v[i] = (a + i < m) ? off[a + i] : 0; sum += v[i]; }
    part[t] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int64_t x = (t >= o) ? part[t - o] : 0;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    int64_t run = part[t] - sum;
#pragma unroll
    for (int i = 0; i < SCAN_PER; ++i) { if (a + i < m) off[a + i] = run; run += v[i]; }
    if (t == 1023) block_tot[blockIdx.x] = part[1023];
}

__global__ __launch_bounds__(1024) void scan_tops_kernel(int64_t *block_tot, int64_t nb, int64_t *off_last, int64_t *total)
{
    // nb <= 1024 * 1024 block totals, scanned by one workgroup in chunks of 1024
    __shared__ int64_t part[1024];
    __shared__ int64_t carry;
    const int t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += 1024) {
        const int64_t x0 = (base + t < nb) ? block_tot[base + t] : 0;
        part[t] = x0;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int64_t x = (t >= o) ? part[t - o] : 0;
            __syncthreads();
            part[t] += x;
            __syncthreads();
        }
        if (base + t < nb) block_tot[base + t] = carry + part[t] - x0;
        __syncthreads();
        if (t == 1023) carry += part[1023];
        __syncthreads();
    }
    if (t == 0) { *off_last = carry; *total = carry; }
}

__global__ __launch_bounds__(1024) void scan_add_kernel(int64_t *off, int64_t m, const int64_t *block_tot)
{
    const int64_t a = ((int64_t)blockIdx.x * 1024 + threadIdx.x) * SCAN_PER;
    const int64_t add = block_tot[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_PER; ++i) if (a + i < m) off[a + i] += add;
}

}  // namespace

extern "C" {

// workspace: gap tables | total | block totals of the scan | per-lane counts (uint16 x 64 per (view, sequence))
static int64_t ws_tables(int n_views) { return (((int64_t)n_views * (J + 1) * 4 + 15) / 16) * 16; }
static int64_t ws_scan_blocks(int64_t items) { return (items + 1024 * SCAN_PER - 1) / (1024 * SCAN_PER); }

int idl_mimic_max_random_n(void) { return 64; }      // Random_N draws the device generator sorts inside one wavefront

int64_t idl_mimic_workspace(int64_t n, int n_views)
{
    if (n < 0 || n_views < 1 || n_views > MAX_VIEWS) return -1;
    const int64_t items = n * n_views;
    return ws_tables(n_views) + 16 + (ws_scan_blocks(items) + 1) * 8 + items * 64 * 2 + 64;
}

static int mimic_params(int n_views, const double *p_transition, const double *p_transversion, const int32_t *n_random_n, MimicParams &p)
{
    for (int v = 0; v < n_views; ++v) {
        const double a = p_transition[v], b = p_transversion[v];
        IDL_REQUIRE(a >= 0.0 && a < 1.0 && b >= 0.0 && b < 1.0, "probabilities must be in [0, 1)");
        IDL_REQUIRE(n_random_n[v] >= 0 && n_random_n[v] <= 64, "n_random_n outside 0..64");
        IDL_REQUIRE(!(n_random_n[v] > 0 && (a > 0.0 || b > 0.0)), "a view is either a site view or a Random_N view");
        const double keep = (1.0 - a) * (1.0 - b);
        const double q = 1.0 - keep;
        p.one_minus_q[v] = keep;
        p.n_rand[v] = n_random_n[v];
        p.has_sites[v] = q > 0.0;
        if (q > 0.0) {
            const double fa = (a * (1.0 - b)) / q * 4294967296.0;
            const double fb = ((1.0 - a) * b) / q * 4294967296.0;
            const double A = fa >= 4294967295.0 ? 4294967295.0 : fa;
            double B = A + fb;
            if (B > 4294967295.0) B = 4294967295.0;
            p.thr_ts_only[v] = (uint32_t)A;
            p.thr_tv_only[v] = (uint32_t)B;
            p.kind[v] = (b == 0.0) ? 1 : (a == 0.0) ? 2 : 0;
            p.inv_log2_keep[v] = (float)(1.0 / log2(keep));
        }
    }
    return IDL_OK;
}

int idl_mimic_check_lengths(int64_t max_len, int n_views, const double *p_transition, const double *p_transversion)
{
    IDL_REQUIRE(n_views >= 1 && n_views <= MAX_VIEWS && p_transition && p_transversion, "n_views outside 1..64 or NULL buffer");
    IDL_REQUIRE(max_len >= 0 && max_len < (1ll << 30), "mutated sequences longer than 2^30 bases are not supported (an edit holds its position in 30 bits)");
    const double per_lane = (double)((max_len + 63) / 64);
    for (int v = 0; v < n_views; ++v) {
        const double mean = per_lane * (1.0 - (1.0 - p_transition[v]) * (1.0 - p_transversion[v]));
        IDL_REQUIRE(!(per_lane > 65535.0 && mean + 12.0 * sqrt(mean) + 64.0 >= 65535.0),
                    "a sequence this long at these mutation rates exceeds the generator's 16-bit per-lane site counter");
    }
    return IDL_OK;
}

int idl_mimic_edits(const int64_t *lengths, int64_t n, int n_views, const double *p_transition,
                    const double *p_transversion, const int32_t *n_random_n, uint64_t seed,
                    int64_t *edit_off, uint32_t *edits, int64_t edits_capacity, int64_t *total_edits,
                    void *workspace, void *stream)
{
    IDL_REQUIRE(n >= 0 && n_views >= 1 && n_views <= MAX_VIEWS, "n < 0 or n_views outside 1..64");
    IDL_REQUIRE(p_transition && p_transversion && n_random_n && edit_off && workspace, "NULL buffer");
    IDL_REQUIRE(n < (1ll << 32), "more than 2^32 sequences");
    MimicParams p{};
    {
        const int prc = mimic_params(n_views, p_transition, p_transversion, n_random_n, p);
        if (prc != IDL_OK) return prc;
    }
    idl::DeviceInfo di;
    int rc = idl::device_info(&di);
    if (rc != IDL_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    IDL_REQUIRE((((uintptr_t)workspace) & 15u) == 0, "workspace must be 16-byte aligned");
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const int64_t items = n * n_views;
    const int64_t nb = ws_scan_blocks(items);
    uint32_t *tables = (uint32_t *)workspace;
    int64_t *d_total = (int64_t *)((uint8_t *)workspace + ws_tables(n_views));
    int64_t *block_tot = d_total + 2;
    uint16_t *lane_counts = (uint16_t *)(block_tot + nb + 1);
    if (items == 0) {
        IDL_HIP_TRY(hipMemsetAsync(edit_off, 0, sizeof(int64_t), st));
        if (total_edits) *total_edits = 0;
        return IDL_OK;
    }
    int64_t grid = (int64_t)di.cus * 32;
    if (grid > items) grid = items;
    hipLaunchKernelGGL(mimic_table_kernel, dim3((unsigned)n_views), dim3(64), 0, st, p, n_views, tables);
    if (edits == nullptr) {
        hipLaunchKernelGGL(mimic_kernel<false>, dim3((unsigned)grid), dim3(64), 0, st, p, n_views, lengths, n, k0, k1, tables,
                           edit_off, (uint32_t *)nullptr, (int64_t)0, lane_counts);
        hipLaunchKernelGGL(scan_block_kernel, dim3((unsigned)nb), dim3(1024), 0, st, edit_off, items, block_tot);
        hipLaunchKernelGGL(scan_tops_kernel, dim3(1), dim3(1024), 0, st, block_tot, nb, edit_off + items, d_total);
        hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nb), dim3(1024), 0, st, edit_off, items, block_tot);
        IDL_HIP_TRY(hipGetLastError());
        if (total_edits) {
            IDL_HIP_TRY(hipMemcpyAsync(total_edits, d_total, sizeof(int64_t), hipMemcpyDeviceToHost, st));
            IDL_HIP_TRY(hipStreamSynchronize(st));
        }
        return IDL_OK;
    }
    // fill pass: edit_off must hold the offsets produced by the count pass with the same arguments
    IDL_REQUIRE(edits_capacity >= 0, "negative capacity");
    hipLaunchKernelGGL(mimic_kernel<true>, dim3((unsigned)grid), dim3(64), 0, st, p, n_views, lengths, n, k0, k1, tables,
                       edit_off, edits, edits_capacity, lane_counts);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}


// slot capacity of view v for sequences up to max_len bases: the expected number of sites + 10 sigma + 32 (never more than max_len)
static int64_t slot_cap(double a, double b, int n_rand, int64_t max_len)
{
    if (n_rand > 0) return n_rand;
    const double q = 1.0 - (1.0 - a) * (1.0 - b);
    if (q <= 0.0) return 0;
    const double mean = q * (double)max_len;
    int64_t cap = (int64_t)ceil(mean + 10.0 * sqrt(mean) + 32.0);
    if (cap > max_len) cap = max_len;
    return (cap + 3) & ~(int64_t)3;
}

int64_t idl_mimic_slots_workspace(int n_views) { return (n_views < 1 || n_views > MAX_VIEWS) ? -1 : ws_tables(n_views); }

int64_t idl_mimic_slots_capacity(int64_t n, int n_views, const double *p_transition, const double *p_transversion, const int32_t *n_random_n,
                                 int64_t max_len)
{
    if (n < 0 || n_views < 1 || n_views > MAX_VIEWS || !p_transition || !p_transversion || !n_random_n || max_len < 0) return -1;
    int64_t total = 0;
    for (int v = 0; v < n_views; ++v) total += n * slot_cap(p_transition[v], p_transversion[v], n_random_n[v], max_len);
    return total;
}

int idl_mimic_edits_slots(const int64_t *lengths, int64_t n, int n_views, const double *p_transition, const double *p_transversion,
                          const int32_t *n_random_n, uint64_t seed, int64_t max_len, int64_t *edit_ranges, uint32_t *edits,
                          int64_t edits_capacity, int32_t *overflow, void *workspace, void *stream)
{
    IDL_REQUIRE(n >= 0 && n_views >= 1 && n_views <= MAX_VIEWS, "n < 0 or n_views outside 1..64");
    IDL_REQUIRE(p_transition && p_transversion && n_random_n && edit_ranges && overflow && workspace, "NULL buffer");
    IDL_REQUIRE(n < (1ll << 32), "more than 2^32 sequences");
    int rc = idl_mimic_check_lengths(max_len, n_views, p_transition, p_transversion);
    if (rc != IDL_OK) return rc;
    MimicParams p{};
    rc = mimic_params(n_views, p_transition, p_transversion, n_random_n, p);
    if (rc != IDL_OK) return rc;
    SlotLayout sl{};
    int64_t total = 0;
    for (int v = 0; v < n_views; ++v) {
        const int64_t cap = slot_cap(p_transition[v], p_transversion[v], n_random_n[v], max_len);
        IDL_REQUIRE(cap < (1ll << 31), "slot capacity beyond 2^31");
        sl.base[v] = total; sl.cap[v] = (int32_t)cap;
        total += n * cap;
    }
    IDL_REQUIRE(edits_capacity >= total && (edits != nullptr || total == 0), "edits is smaller than idl_mimic_slots_capacity()");
    IDL_REQUIRE((((uintptr_t)workspace) & 15u) == 0, "workspace must be 16-byte aligned");
    idl::DeviceInfo di;
    rc = idl::device_info(&di);
    if (rc != IDL_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int64_t items = n * n_views;
    if (items == 0) return IDL_OK;
    uint32_t *tables = (uint32_t *)workspace;
    int64_t grid;
    // two items per wave, 8 buffered sites per lane and item (8.6 KB of LDS: 18 waves per CU), 128 workgroups per CU -- swept at cfg2:
    // <1, 24> x 32 per CU (one item at a time, the round-3 start) 0.82 ms, <4, 12> 0.74, <4, 8> 0.62, <3, 8> 0.58, <2, 12> 0.61, <2, 8> 0.55
    constexpr int GEN_ITEMS = 2, GEN_BUF = 8;
    grid = (int64_t)di.cus * 128;
    const int64_t units = (n + GEN_ITEMS - 1) / GEN_ITEMS * n_views;
    if (grid > units) grid = units;
    hipLaunchKernelGGL(mimic_table_kernel, dim3((unsigned)n_views), dim3(64), 0, st, p, n_views, tables);
    auto kern = mimic_slots_kernel<GEN_ITEMS, GEN_BUF>;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64), 0, st, p, n_views, lengths, n, (uint32_t)seed, (uint32_t)(seed >> 32),
                       (const uint32_t *)tables, sl, edit_ranges, edits, overflow);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

}  // extern "C"
