// vectorise.hip -- the k-mer / CGR / canonical vectoriser for gfx950 (MI355X).
//
// Two kernels share this file:
//   vectorise2_kernel (default): four wavefronts per sequence around one LDS histogram; the un-mutated sequence is
//       counted once and every mimic view is produced as window deltas (see the "v2" block comment below);
//   vectorise_kernel  (v1): one wavefront per sequence, 4096-base chunks, a full recount per view with the edits
//       applied through an LDS staging image.  Kept for IDL_INIT_FROM_OUT (the accumulate-on-top scalar API), where
//       every view starts from a caller-provided row.
// Both finish a view with ONE pass over the histogram (pseudocount / normalise / CGR permutation / reverse-complement
// collapse) and one coalesced row store -- the only global write of the sequence.
//
// Reference semantics restated here (not translated: the reference is a scalar byte loop):
//   idelucs/kmers.pyx:2-50   kmer_counts  -> window = k consecutive valid bases, first base in
//                                            the most significant 2-bit pair of the index
//   idelucs/kmers.pyx:53-123 cgr          -> same windows, pixel index = fixed bit permutation
//   idelucs/utils.py:191-221 kmer_rev_comp-> canonical collapse with int truncation
//   idelucs/utils.py:242-250 ones-init, counts / sum(counts)
//   idelucs/utils.py:54-135  transforms   -> substitution edits (XOR on 2-bit codes / set-N)
#include <stdlib.h>
#include "dev_env.h"

#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "common.h"

namespace {

struct VecArgs {
    const uint4 *codes;
    const uint2 *mask;
    const int64_t *slot_off;
    const int64_t *lengths;
    int64_t n;
    int mode, init, out_kind, n_views;
    const uint32_t *edits;
    const int64_t *edit_off;
    int eo_shift;   // 0: edit_off is the CSR array [n_views * n + 1]; 1: it is [n_views * n][2] = (begin, end) per (view, sequence) (idl_vectorise_ranges)
    void *out;
    int64_t view_stride;
    int sc_slots;   // v2: slots (64 bases) staged in LDS at a time
    int64_t max_len; // upper bound on the sequence lengths, 0 = unknown
    int *redo_count; // v3 device scratch: [0] number of sequences v3 left to the second pass (v2 with redo != 0 takes exactly those), [1] head of v3's sequence queue, [2..] the list of those sequences
    int redo, v3_sc; //   (a sequence is v3's iff its edits fit ecap, its pairs lcap and its slots v3_sc)
    int chunk;       // v3: consecutive sequences per workgroup and round (chunks are dealt round-robin over the workgroups)
    int ecap, lcap;  // v3: LDS capacity for the staged edits of all views / for the recorded (edit, window) pairs
    unsigned long long *dbg;   // diagnostic (IDELUCS_DEV=vec_dbg): per-workgroup cycle sums of the phases
    int ablate;      // diagnostic builds only (IDELUCS_DEV=vec_ablate): 1 no row stores, 2 no H0 count, 4 no deltas, 8 raw epilogue
};

// first / one-past-last edit of item it = view * n + sequence, in either layout of edit_off
__device__ __forceinline__ int64_t eo_begin(const VecArgs &a, int64_t it) { return a.edit_off[it << a.eo_shift]; }
__device__ __forceinline__ int64_t eo_end(const VecArgs &a, int64_t it) { return a.edit_off[(it << a.eo_shift) + 1]; }


constexpr int STAGE_DWORDS = 256 + 128;  // 64 lanes x (4 code words + 2 mask words)
constexpr int V3_REDO_CAP = 1023;        // sequences the v3 kernel can list for the second pass on v2 (beyond that the second pass scans)

__device__ __forceinline__ uint32_t spread_bits(uint32_t x)
{
    x = (x | (x << 4)) & 0x0F0Fu;
    x = (x | (x << 2)) & 0x3333u;
    x = (x | (x << 1)) & 0x5555u;
    return x;
}

// k-mer index (A0 C1 G2 T3, oldest base in the top pair) whose CGR pixel is `o` = (i << K) + j.
// kmers.pyx:110-123: bit p of i/j belongs to the p-th (oldest first) base of the window;
// (ibit, jbit): A(1,0) C(0,0) G(0,1) T(1,1)  =>  code = jbit<<1 | !(ibit ^ jbit).
template <int K>
__device__ __forceinline__ uint32_t cgr_pixel_to_kmer(uint32_t o)
{
    constexpr uint32_t M = (1u << K) - 1u;
    uint32_t i = o >> K, j = o & M;
    uint32_t ir = __brev(i) >> (32 - K), jr = __brev(j) >> (32 - K);  // base p -> bit K-1-p
    uint32_t lo = ~(ir ^ jr) & M;
    return (spread_bits(jr) << 1) | spread_bits(lo);
}

// reverse complement of a k-mer index (utils.py:191-206): complement every base (3 - code),
// reverse the order of the 2-bit groups.
template <int K>
__device__ __forceinline__ uint32_t revcomp(uint32_t b)
{
    constexpr uint32_t KM = (1u << (2 * K)) - 1u;
    uint32_t r = __brev(~b & KM) >> (32 - 2 * K);                  // reverses bits inside the pairs too
    return ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);   // ... so swap them back
}

__device__ __forceinline__ int64_t wave_sum_i64(int64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Count the 16 windows that END in `cur` (16 bases, first base in the top pair).
//   prev : the 16 bases before `cur`
//   M    : (invalid bits of prev's 16 bases) << 16 | invalid bits of cur's 16 bases, base j at bit 15-j
template <int K>
__device__ __forceinline__ uint32_t count_word(uint32_t *hist, uint32_t prev, uint32_t cur, uint32_t M)
{
    constexpr uint32_t KM = (1u << (2 * K)) - 1u;
    // bit 15-j of `inv` is set when any of the K bases of the window ending at base j is invalid
    uint32_t inv = M;
#pragma unroll
    for (int t = 1; t < K; ++t) inv |= (M >> t);
    inv &= 0xFFFFu;
    const uint64_t w = ((uint64_t)prev << 32) | cur;
    if (__ballot(inv != 0u) == 0ull) {  // wave-uniform fast path: every window of every lane is valid
#pragma unroll
        for (int j = 0; j < 16; ++j) atomicAdd(&hist[(uint32_t)(w >> (30 - 2 * j)) & KM], 1u);
        return 16u;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (!((inv >> (15 - j)) & 1u)) atomicAdd(&hist[(uint32_t)(w >> (30 - 2 * j)) & KM], 1u);
    }
    return 16u - (uint32_t)__popc(inv);
}

template <int K>
__global__ __launch_bounds__(64) void vectorise_kernel(VecArgs a)
{
    constexpr int F = 1 << (2 * K);
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *hist = lds;
    uint32_t *st_codes = lds + F;
    uint32_t *st_mask = st_codes + 256;
    const int lane = threadIdx.x;
    const int64_t row_len = (a.mode == IDL_MODE_CANONICAL)
                                ? ((K % 2 == 0) ? (F + (1 << K)) / 2 : F / 2)
                                : F;

    for (int64_t s = blockIdx.x; s < a.n; s += gridDim.x) {
        const int64_t slot0 = a.slot_off[s];
        const int64_t nslots = (a.lengths[s] + 63) >> 6;       // a record occupies ceil(len / 64) slots from slot_off[s] (records need not be adjacent)

        for (int v = 0; v < a.n_views; ++v) {
            const int64_t out_base = (int64_t)v * a.view_stride + s * row_len;

            // ---------------- histogram init
            if (a.init == IDL_INIT_FROM_OUT) {
                const uint32_t *src = (const uint32_t *)a.out + out_base;
                // the histogram is kept in k-mer order; a CGR row is stored in pixel order
                for (int i = lane; i < F; i += 64)
                    hist[(a.mode == IDL_MODE_CGR) ? cgr_pixel_to_kmer<K>((uint32_t)i) : (uint32_t)i] = src[i];
            } else {
                const uint32_t iv = (a.init == IDL_INIT_ONE) ? 1u : 0u;
                for (int i = lane; i < F; i += 64) hist[i] = iv;
            }
            __syncthreads();

            int64_t ecur = 0, eend = 0;
            if (a.edits != nullptr) {
                ecur = eo_begin(a, (int64_t)v * a.n + s);
                eend = eo_end(a, (int64_t)v * a.n + s);
            }

            uint32_t carry_code = 0u, carry_mask = 0xFFFFFFFFu;  // nothing before the sequence: invalid
            uint32_t cnt = 0u;

            for (int64_t c0 = 0; c0 < nslots; c0 += 64) {
                const int64_t slot = c0 + lane;
                uint4 cw = make_uint4(0u, 0u, 0u, 0u);
                uint2 mw = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
                if (slot < nslots) {
                    cw = a.codes[slot0 + slot];
                    mw = a.mask[slot0 + slot];
                }

                // ---------------- apply this view's edits that fall inside the chunk
                if (ecur < eend) {
                    const int64_t chunk_lo = c0 * 64, chunk_hi = chunk_lo + 4096;
                    *(uint4 *)(st_codes + lane * 4) = cw;
                    *(uint2 *)(st_mask + lane * 2) = mw;
                    __syncthreads();
                    while (ecur < eend) {
                        const int64_t e = ecur + lane;
                        bool act = false;
                        if (e < eend) {
                            const uint32_t ed = a.edits[e];
                            const int64_t pos = (int64_t)(ed & 0x3FFFFFFFu);
                            act = pos < chunk_hi;
                            if (act && pos >= chunk_lo) {
                                const uint32_t rel = (uint32_t)(pos - chunk_lo), op = ed >> 30;
                                if (op == 0u) atomicOr(&st_mask[rel >> 5], 0x80000000u >> (rel & 31u));
                                else atomicXor(&st_codes[rel >> 4], op << (30u - 2u * (rel & 15u)));
                            }
                        }
                        const int na = __popcll(__ballot(act));
                        ecur += na;
                        if (na < 64) break;
                    }
                    __syncthreads();
                    cw = *(uint4 *)(st_codes + lane * 4);
                    mw = *(uint2 *)(st_mask + lane * 2);
                    __syncthreads();
                }

                // ---------------- history of the previous 16 bases (previous lane / previous chunk)
                uint32_t pc = __shfl_up(cw.w, 1, 64);
                uint32_t pm = __shfl_up(mw.y, 1, 64);
                if (lane == 0) { pc = carry_code; pm = carry_mask; }
                carry_code = __shfl(cw.w, 63, 64);
                carry_mask = __shfl(mw.y, 63, 64);

                cnt += count_word<K>(hist, pc, cw.x, (pm << 16) | (mw.x >> 16));
                cnt += count_word<K>(hist, cw.x, cw.y, mw.x);
                cnt += count_word<K>(hist, cw.y, cw.z, (mw.x << 16) | (mw.y >> 16));
                cnt += count_word<K>(hist, cw.z, cw.w, mw.y);
            }
            __syncthreads();

            // ---------------- epilogue: one pass over the histogram, one coalesced global write
            const int64_t windows = wave_sum_i64((int64_t)cnt);

            if (a.mode == IDL_MODE_CANONICAL) {
                // utils.py:208-221: canonical b (b <= rc): int32((h[b] + h[rc]) * 0.5); ascending order
                int64_t part = 0;
                if (a.out_kind != IDL_OUT_COUNTS_I32) {
                    for (int b0 = 0; b0 < F; b0 += 64) {
                        const uint32_t b = b0 + lane;
                        if (b < (uint32_t)F) {
                            const uint32_t rc = revcomp<K>(b);
                            if (b <= rc) part += (int32_t)(hist[b] + hist[rc]) / 2;
                        }
                    }
                }
                const int64_t S = wave_sum_i64(part);
                int rank0 = 0;
                for (int b0 = 0; b0 < F; b0 += 64) {
                    const uint32_t b = b0 + lane;
                    bool canon = false;
                    int32_t val = 0;
                    if (b < (uint32_t)F) {
                        const uint32_t rc = revcomp<K>(b);
                        canon = b <= rc;
                        if (canon) val = (int32_t)(hist[b] + hist[rc]) / 2;
                    }
                    const uint64_t bal = __ballot(canon);
                    if (canon) {
                        const int64_t o = out_base + rank0 + __popcll(bal & ((1ull << lane) - 1ull));
                        if (a.out_kind == IDL_OUT_COUNTS_I32) ((int32_t *)a.out)[o] = val;
                        else if (a.out_kind == IDL_OUT_FREQ_F64) ((double *)a.out)[o] = (double)val / (double)S;
                        else ((float *)a.out)[o] = (float)((double)val / (double)S);
                    }
                    rank0 += __popcll(bal);
                }
            } else {
                // row sum: every bin started at `init` (or at the caller's value) and got `windows` increments
                int64_t S;
                if (a.init == IDL_INIT_FROM_OUT) S = 0;  // counts only
                else S = windows + ((a.init == IDL_INIT_ONE) ? (int64_t)F : 0);
                const bool small = S < (1ll << 24);  // then int -> float32 is exact and f32 division == f64 division rounded
                const float Sf = (float)S;
                const double Sd = (double)S;
                if (a.mode == IDL_MODE_KMER && F >= 256 && a.out_kind != IDL_OUT_FREQ_F64) {
                    for (int i4 = lane; i4 < F / 4; i4 += 64) {
                        const uint4 h = *(const uint4 *)(hist + i4 * 4);
                        if (a.out_kind == IDL_OUT_COUNTS_I32) {
                            *(uint4 *)((uint32_t *)a.out + out_base + i4 * 4) = h;
                        } else {
                            float4 f;
                            if (small) {
                                f.x = (float)h.x / Sf; f.y = (float)h.y / Sf;
                                f.z = (float)h.z / Sf; f.w = (float)h.w / Sf;
                            } else {
                                f.x = (float)((double)h.x / Sd); f.y = (float)((double)h.y / Sd);
                                f.z = (float)((double)h.z / Sd); f.w = (float)((double)h.w / Sd);
                            }
                            *(float4 *)((float *)a.out + out_base + i4 * 4) = f;
                        }
                    }
                } else {
                    for (int i = lane; i < F; i += 64) {
                        const uint32_t src = (a.mode == IDL_MODE_CGR) ? cgr_pixel_to_kmer<K>((uint32_t)i) : (uint32_t)i;
                        const uint32_t h = hist[src];
                        if (a.out_kind == IDL_OUT_COUNTS_I32) ((uint32_t *)a.out)[out_base + i] = h;
                        else if (a.out_kind == IDL_OUT_FREQ_F64) ((double *)a.out)[out_base + i] = (double)h / Sd;
                        else ((float *)a.out)[out_base + i] = small ? (float)h / Sf : (float)((double)h / Sd);
                    }
                }
            }
            __syncthreads();  // the histogram is re-initialised for the next view
        }
    }
}


// =====================================================================================================
// v2: delta views.  The mimic views of a sequence differ from the un-mutated sequence at ~1.5 % of
// the positions, so the sequence is counted ONCE (H0) and a view is produced by moving the <= k windows
// around each edit from their old bin to their new bin.  Because every substitution is an XOR on a 2-bit
// code (or "becomes N"), the mutated window is old_kmer ^ (the XORs of the edits inside the window) -- the
// staged copy of the sequence is never modified -- and an N edit simply invalidates the window.  After the
// view's row has been written the moves are undone (replayed from an LDS list when it fits).
// The sequence is staged in LDS one "super-chunk" (sc_slots slots + a halo slot of history) at a time;
// lanes own 16-base words (a 10 kbp sequence is 9.8 rounds of 64 lanes, not 2.4 rounds of 4096 bases);
// invalid windows go to garbage bins instead of being branched around; the epilogue divides by
// multiplying with the row's correctly rounded reciprocal plus one FMA correction (Markstein; verified
// bit-identical to float32(float64 division) for all 1 <= c <= S < 2^24 by tools/markstein_check.c).
// =====================================================================================================
constexpr int V2_LIST_CAP = 768;   // (edit, window) pairs recorded per view for a replay-undo
constexpr int V2_EDIT_CAP = 192;   // edits of a view preloaded into LDS
constexpr int V2_WAVES = 4;        // wavefronts per sequence

// B16: two 16-bit bins per LDS word (valid while every count stays < 65536, i.e. sequences shorter than ~65 kbp):
// half the histogram footprint -> more sequences resident per CU.
template <int K, bool B16>
struct V2 {
    static constexpr int NT = 64 * V2_WAVES;   // threads (V2_WAVES wavefronts) that share one sequence and one LDS histogram
    static constexpr int F = 1 << (2 * K);
    static constexpr uint32_t KM = (1u << (2 * K)) - 1u;
    static constexpr uint32_t VM = (1u << K) - 1u;
    static constexpr int HD = B16 ? (F / 2 + 4) & ~3 : F + 4;      // histogram words incl. 4 garbage bins (kept a multiple of 4)

    static __device__ __forceinline__ void bump(uint32_t *hist, uint32_t bin, uint32_t sign)   // sign = 1 or 0xFFFFFFFF (-1)
    {
        if (B16) atomicAdd(&hist[bin >> 1], sign << (16u * (bin & 1u)));
        else atomicAdd(&hist[bin], sign);
    }
    static __device__ __forceinline__ uint32_t get(const uint32_t *hist, uint32_t bin)
    {
        return B16 ? (hist[bin >> 1] >> (16u * (bin & 1u))) & 0xFFFFu : hist[bin];
    }

    // (previous word : word D) of the staged codes and the matching 32 invalid bits
    static __device__ __forceinline__ void fetch(const uint32_t *cod, const uint32_t *msk, int D, uint64_t &w, uint32_t &M)
    {
        w = ((uint64_t)cod[D - 1] << 32) | cod[D];
        const uint32_t m0 = msk[(D >> 1) - 1], m1 = msk[D >> 1];
        M = (D & 1) ? m1 : (uint32_t)((((uint64_t)m0 << 32) | m1) >> 16);
    }

    // global -> LDS: slots [sc_first - 1, sc_first + nloc); LDS slot 0 is the halo (history of 64 bases)
    static __device__ __forceinline__ void stage(const VecArgs &a, int64_t slot0, int64_t sc_first, int nloc, uint32_t *cod,
                                                 uint32_t *msk, int lane)
    {
        __syncthreads();
        for (int i = lane; i < nloc + 1; i += NT) {
            const int64_t g = sc_first - 1 + i;
            uint4 c = make_uint4(0u, 0u, 0u, 0u);
            uint2 m = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
            if (g >= 0) { c = a.codes[slot0 + g]; m = a.mask[slot0 + g]; }
            *(uint4 *)(cod + i * 4) = c;
            *(uint2 *)(msk + i * 2) = m;
        }
        __syncthreads();
    }

    // first edit index in [0, ne) whose position is >= pos (edits sorted); wave-uniform
    template <typename EP>
    static __device__ __forceinline__ int lower_bound(EP E, int ne, int64_t pos)
    {
        int lo = 0, hi = ne;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((int64_t)(E[mid] & 0x3FFFFFFFu) < pos) lo = mid + 1; else hi = mid;
        }
        return lo;
    }

    // One lane per (edit, window offset) pair.  Pair (e, t) is the window ending at w = p_e + t; it is owned by
    // edit e iff w <= min(p_e + K - 1, p_{e+1} - 1) (the latest edit at or before w), and handled here iff it ends
    // inside the staged range [lo, hi) and inside the sequence.  old = the window as staged; new = old ^ (XORs of
    // all edits inside the window), invalid if one of them is an N edit.  hist[old] += s_old, hist[new] += s_new.
    template <bool REC, typename EP>
    static __device__ __forceinline__ int pair_pass(EP E, int ne, int i0, int64_t lo, int64_t hi, int64_t L, const uint32_t *cod,
                                                    const uint32_t *msk, uint32_t *hist, uint16_t *list_old, uint16_t *list_new,
                                                    uint32_t s_old, uint32_t s_new, int lane)
    {
        // all positions are taken relative to `lo` and fit 32 bits (edit positions are < 2^30)
        const int hi_r = (int)(hi - lo);
        const int end_r = (int)((L < hi ? L : hi) - lo) - 1;       // last window end handled here
        const int lo32 = (int)(lo > 0x3FFFFFFF ? 0x3FFFFFFF : lo); // an edit at p has p - lo = p - lo32 whenever it can matter
        int dwin = 0;
        const int npairs = (ne - i0) * K;
        for (int q0 = 0; q0 < npairs; q0 += NT) {
            const int q = q0 + lane;
            bool inside = false;
            uint32_t ko = 0xFFFFu, kn = 0xFFFFu;
            if (q < npairs) {
                const int ei = i0 + q / K, t = q % K;
                const uint32_t ed = E[ei];
                const int p = (int)(ed & 0x3FFFFFFFu) - lo32;           // relative edit position (may be slightly negative)
                inside = p < hi_r;
                int last = p + K - 1;
                if (ei + 1 < ne) { const int nx = (int)(E[ei + 1] & 0x3FFFFFFFu) - lo32 - 1; if (nx < last) last = nx; }
                if (last > end_r) last = end_r;
                const int w = p + t;
                if (w <= last && w >= 0) {
                    const uint32_t rel = (uint32_t)w + 64u;
                    const int D = (int)(rel >> 4), j = (int)(rel & 15u);
                    uint64_t ww; uint32_t M;
                    fetch(cod, msk, D, ww, M);
                    if (((M >> (15 - j)) & VM) == 0u) {
                        ko = (uint32_t)(ww >> (30 - 2 * j)) & KM;
                        uint32_t xm = (ed >> 30) << (2 * t);
                        bool dead = (ed >> 30) == 0u;
                        for (int i = ei - 1; i >= 0; --i) {          // earlier edits that also lie inside this window (rare)
                            const uint32_t e2 = E[i];
                            const int d = w - ((int)(e2 & 0x3FFFFFFFu) - lo32);
                            if (d >= K) break;
                            dead |= (e2 >> 30) == 0u;
                            xm ^= (e2 >> 30) << (2 * d);
                        }
                        kn = dead ? 0xFFFFu : (ko ^ xm);
                    }
                }
            }
            if (REC) { if (q < npairs && q < V2_LIST_CAP) { list_old[q] = (uint16_t)ko; list_new[q] = (uint16_t)kn; } }
            if (ko != kn) {
                if (ko != 0xFFFFu) { bump(hist, ko, s_old); --dwin; }
                if (kn != 0xFFFFu) { bump(hist, kn, s_new); ++dwin; }
            }
            if (__ballot(inside) == 0ull) break;      // sorted: every later edit lies beyond this super-chunk (per wave)
        }
        __syncthreads();
        return dwin;
    }

    static __device__ __forceinline__ void replay(int npairs, uint32_t *hist, const uint16_t *list_old, const uint16_t *list_new, int lane)
    {
        for (int q = lane; q < npairs; q += NT) {
            const uint32_t o = list_old[q], nw = list_new[q];
            if (o != nw) {
                if (nw != 0xFFFFu) bump(hist, nw, 0xFFFFFFFFu);
                if (o != 0xFFFFu) bump(hist, o, 1u);
            }
        }
        __syncthreads();
    }

    // count every window ending in the staged super-chunk (nloc slots = 4*nloc words)
    static __device__ __forceinline__ uint32_t count_all(const uint32_t *cod, const uint32_t *msk, int nloc, uint32_t *hist, int lane)
    {
        uint32_t cnt = 0;
        const int nd = nloc * 4;
        for (int d0 = 0; d0 < nd; d0 += NT) {
            const int d = d0 + lane;
            if (d < nd) {
                uint64_t w; uint32_t M;
                fetch(cod, msk, 4 + d, w, M);
                uint32_t inv = M;
#pragma unroll
                for (int t = 1; t < K; ++t) inv |= (M >> t);
                inv &= 0xFFFFu;
                if (__ballot(inv != 0u) == 0ull) {               // uniform over the active lanes
#pragma unroll
                    for (int j = 0; j < 16; ++j) bump(hist, (uint32_t)(w >> (30 - 2 * j)) & KM, 1u);
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {               // invalid windows land in a garbage bin past the histogram
                        const uint32_t km = (uint32_t)(w >> (30 - 2 * j)) & KM;
                        bump(hist, ((inv >> (15 - j)) & 1u) ? (uint32_t)F + (lane & 3) : km, 1u);
                    }
                }
                cnt += 16u - (uint32_t)__popc(inv);
            }
        }
        __syncthreads();
        return cnt;
    }

    // all delta work of one view in one direction (forward: s_old = -1, s_new = +1)
    template <bool REC, typename EP>
    static __device__ __forceinline__ int view_pass(const VecArgs &a, EP E, int ne, int64_t slot0, int64_t nslots, int64_t nsc, int SC,
                                                    int64_t L, uint32_t *cod, uint32_t *msk, uint32_t *hist, uint16_t *list_old,
                                                    uint16_t *list_new, uint32_t s_old, uint32_t s_new, int lane)
    {
        if (nsc == 1) return pair_pass<REC>(E, ne, 0, 0, nslots * 64, L, cod, msk, hist, list_old, list_new, s_old, s_new, lane);
        int dwin = 0;
        for (int64_t sc = 0; sc < nsc; ++sc) {
            const int nloc = (int)((nslots - sc * SC) < SC ? (nslots - sc * SC) : SC);
            const int64_t lo = sc * SC * 64, hi = lo + (int64_t)nloc * 64;
            stage(a, slot0, sc * SC, nloc, cod, msk, lane);
            const int i0 = lower_bound(E, ne, lo - (K - 1));
            dwin += pair_pass<false>(E, ne, i0, lo, hi, L, cod, msk, hist, list_old, list_new, s_old, s_new, lane);
        }
        return dwin;
    }
};

// second launch-bound argument = waves per SIMD wanted: 8 workgroups/CU with 16-bit bins (16 KB LDS each), 6 with 32-bit bins
template <int K, bool B16>
__global__ __launch_bounds__(64 * V2_WAVES, B16 ? 8 : 6) void vectorise2_kernel(VecArgs a)
{
    using T = V2<K, B16>;
    constexpr int F = T::F;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t *hist = lds;                       // F bins + 4 garbage bins (32- or 16-bit)
    uint32_t *cod = lds + T::HD;                // (SC + 1) slots x 4 words, slot 0 = halo
    const int SC = a.sc_slots;
    uint32_t *msk = cod + (SC + 1) * 4;         // (SC + 1) slots x 2 words
    uint32_t *ed_lds = msk + (SC + 1) * 2;      // V2_EDIT_CAP edits of the current view
    uint16_t *list_old = (uint16_t *)(ed_lds + V2_EDIT_CAP);
    uint16_t *list_new = list_old + V2_LIST_CAP;
    int64_t *red = (int64_t *)(list_old + 2 * V2_LIST_CAP);   // V2_WAVES partial sums
    int64_t *vrange = red + V2_WAVES;                         // [2 * n_views] edit ranges (begin, end) of the views
    constexpr int NT = T::NT;
    const int lane = threadIdx.x;                             // thread index within the sequence's workgroup
    auto block_sum = [&](int64_t v) -> int64_t {
        v = wave_sum_i64(v);
        __syncthreads();
        if ((lane & 63) == 0) red[lane >> 6] = v;
        __syncthreads();
        int64_t t = 0;
#pragma unroll
        for (int w = 0; w < V2_WAVES; ++w) t += red[w];
        return t;
    };
    const int64_t row_len = (a.mode == IDL_MODE_CANONICAL) ? ((K % 2 == 0) ? (F + (1 << K)) / 2 : F / 2) : F;
    const uint32_t iv = (a.init == IDL_INIT_ONE) ? 1u : 0u;
    const uint32_t ivw = B16 ? iv * 0x00010001u : iv;

    // second pass after v3: nothing left to do -> exit; a few sequences -> exactly those (v3 listed them); many -> scan all and skip v3's
    const int redo_n = a.redo != 0 ? *(const volatile int *)a.redo_count : 0;
    if (a.redo != 0 && redo_n == 0) return;
    const bool redo_listed = a.redo != 0 && redo_n <= V3_REDO_CAP;
    const int64_t loop_n = redo_listed ? (int64_t)redo_n : a.n;
    for (int64_t si = blockIdx.x; si < loop_n; si += gridDim.x) {
        const int64_t s = redo_listed ? ((const int64_t *)(a.redo_count + 2))[si] : si;
        const int64_t slot0 = a.slot_off[s];
        const int64_t L = a.lengths[s];
        const int64_t nslots = (L + 63) >> 6;
        const int64_t nsc = (nslots + SC - 1) / SC;
        if (a.redo != 0 && !redo_listed) {      // second pass without a list: only the sequences v3 left alone (the predicate of vectorise3_kernel's stage_next, restated)
            int64_t te = 0, dmax = 0;
            if (a.edits != nullptr)
                for (int v = 0; v < a.n_views; ++v) {
                    int64_t d = eo_end(a, (int64_t)v * a.n + s) - eo_begin(a, (int64_t)v * a.n + s);
                    d = d < 0 ? 0 : (d > 0x3FFFFFF ? 0x3FFFFFF : d);
                    te += d;
                    dmax = d > dmax ? d : dmax;
                }
            // (redo == 2: the first pass was vectorise4_kernel, whose registers also bound a view's edits and the staged slots)
            const bool first_did = a.redo == 2 ? (nslots <= 64 * 3 && te <= (int64_t)a.ecap)
                                               : (te <= (int64_t)a.ecap && te * K <= (int64_t)a.lcap);
            if (first_did && nslots >= 0 && nslots <= (int64_t)a.v3_sc && L >= 0 && L <= nslots * 64) continue;
        }

        for (int i = lane; i < T::HD / 4; i += NT) *(uint4 *)(hist + i * 4) = make_uint4(ivw, ivw, ivw, ivw);

        // ---------------- H0: the un-mutated sequence, counted once (sc 0 last, so a short sequence stays staged)
        uint32_t cnt0 = 0;
        for (int64_t sc = nsc - 1; sc >= 0; --sc) {
            const int nloc = (int)((nslots - sc * SC) < SC ? (nslots - sc * SC) : SC);
            T::stage(a, slot0, sc * SC, nloc, cod, msk, lane);
            if (!(a.ablate & 2)) cnt0 += T::count_all(cod, msk, nloc, hist, lane);
        }
        const int64_t windows0 = block_sum((int64_t)cnt0);

        // the views' edit ranges, loaded once per sequence (one thread per view) while H0 is being counted
        __syncthreads();
        if (lane < a.n_views) {
            int64_t b0 = 0, e0 = 0;
            if (a.edits != nullptr) { b0 = eo_begin(a, (int64_t)lane * a.n + s); e0 = eo_end(a, (int64_t)lane * a.n + s); }
            vrange[2 * lane] = b0;
            vrange[2 * lane + 1] = e0;
        }
        __syncthreads();
        // the view with the most edits goes last: nothing has to be undone (or recorded) after the last view
        int vlast = a.n_views - 1;
        {
            int64_t best = -1;
            for (int v = 0; v < a.n_views; ++v) {
                const int64_t c = vrange[2 * v + 1] - vrange[2 * v];
                if (c > best) { best = c; vlast = v; }
            }
        }
        auto view_at = [&](int vi) -> int { return (vi == a.n_views - 1) ? vlast : (vi < vlast ? vi : vi + 1); };
        // register prefetch of a view's edits (<= one per thread when the list fits the LDS cache)
        uint32_t pre = 0u;
        auto prefetch = [&](int vi) {
            if (vi < a.n_views) {
                const int v2 = view_at(vi);
                const int64_t b2 = vrange[2 * v2], n2 = vrange[2 * v2 + 1] - b2;
                if (n2 <= V2_EDIT_CAP && lane < n2) pre = a.edits[b2 + lane];
            }
        };
        static_assert(V2_EDIT_CAP <= 64 * V2_WAVES, "one prefetched edit per thread");
        prefetch(0);
        for (int vi = 0; vi < a.n_views; ++vi) {
            const int v = view_at(vi);
            const int64_t out_base = (int64_t)v * a.view_stride + s * row_len;
            const int64_t eb = vrange[2 * v], ee = vrange[2 * v + 1];
            const int ne = (a.ablate & 4) ? 0 : (int)(ee - eb);
            const bool mutated = ne > 0 && nsc > 0;
            const bool in_lds = ne <= V2_EDIT_CAP;      // edits preloaded into LDS
            const bool rec = in_lds && nsc == 1 && ne * K <= V2_LIST_CAP && vi + 1 < a.n_views;
            const uint32_t *eg = a.edits + eb;

            // ---------------- forward: hist = H0 - (old windows) + (new windows)
            int dwin = 0;
            if (mutated) {
                if (in_lds) {
                    __syncthreads();
                    if (lane < ne) ed_lds[lane] = pre;
                    __syncthreads();
                    if (rec) dwin = T::template view_pass<true>(a, (const uint32_t *)ed_lds, ne, slot0, nslots, nsc, SC, L, cod, msk, hist, list_old, list_new, 0xFFFFFFFFu, 1u, lane);
                    else dwin = T::template view_pass<false>(a, (const uint32_t *)ed_lds, ne, slot0, nslots, nsc, SC, L, cod, msk, hist, list_old, list_new, 0xFFFFFFFFu, 1u, lane);
                } else {
                    dwin = T::template view_pass<false>(a, eg, ne, slot0, nslots, nsc, SC, L, cod, msk, hist, list_old, list_new, 0xFFFFFFFFu, 1u, lane);
                }
            }
            prefetch(vi + 1);
            const int64_t windows = windows0 + block_sum((int64_t)dwin);
            __syncthreads();

            // ---------------- epilogue: one pass over the histogram, one coalesced global write
            if (a.mode == IDL_MODE_CANONICAL) {
              if (lane < 64) {          // ranks come from a running ballot prefix: one wavefront walks the bins in order
                int64_t part = 0;
                if (a.out_kind != IDL_OUT_COUNTS_I32) {
                    for (int b0 = 0; b0 < F; b0 += 64) {
                        const uint32_t b = b0 + lane;
                        if (b < (uint32_t)F) {
                            const uint32_t rc = revcomp<K>(b);
                            if (b <= rc) part += (int32_t)(T::get(hist, b) + T::get(hist, rc)) / 2;
                        }
                    }
                }
                const int64_t S = wave_sum_i64(part);
                int rank0 = 0;
                for (int b0 = 0; b0 < F; b0 += 64) {
                    const uint32_t b = b0 + lane;
                    bool canon = false;
                    int32_t val = 0;
                    if (b < (uint32_t)F) {
                        const uint32_t rc = revcomp<K>(b);
                        canon = b <= rc;
                        if (canon) val = (int32_t)(T::get(hist, b) + T::get(hist, rc)) / 2;
                    }
                    const uint64_t bal = __ballot(canon);
                    if (canon) {
                        const int64_t o = out_base + rank0 + __popcll(bal & ((1ull << lane) - 1ull));
                        if (a.out_kind == IDL_OUT_COUNTS_I32) ((int32_t *)a.out)[o] = val;
                        else if (a.out_kind == IDL_OUT_FREQ_F64) ((double *)a.out)[o] = (double)val / (double)S;
                        else ((float *)a.out)[o] = (float)((double)val / (double)S);
                    }
                    rank0 += __popcll(bal);
                }
              }
            } else {
                const int64_t S = windows + (iv ? (int64_t)F : 0);
                const bool small = S < (1ll << 24);   // ints exact in float32; Markstein division == float32(float64 division)
                const float Sf = (float)S, rS = 1.0f / Sf;
                const double Sd = (double)S;
                if (a.mode == IDL_MODE_KMER && F >= 256 && a.out_kind != IDL_OUT_FREQ_F64) {
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const f32x2 rS2 = {rS, rS}, nS2 = {-Sf, -Sf};
                    auto emit = [&](int i4, uint4 h) {         // four consecutive bins starting at 4*i4
                        if (a.ablate & 1) { if (h.x == 0xFFFFFFF0u) *(uint4 *)((uint32_t *)a.out + out_base + i4 * 4) = h; return; }
                        if (a.out_kind == IDL_OUT_COUNTS_I32 || (a.ablate & 8)) {
                            *(uint4 *)((uint32_t *)a.out + out_base + i4 * 4) = h;
                        } else {
                            float4 f;
                            if (small) {                           // q = c * r; q' = fma(fma(-q, S, c), r, q), two bins per packed op
                                const f32x2 c01 = {(float)h.x, (float)h.y}, c23 = {(float)h.z, (float)h.w};
                                const f32x2 q01 = c01 * rS2, q23 = c23 * rS2;
                                const f32x2 o01 = __builtin_elementwise_fma(__builtin_elementwise_fma(q01, nS2, c01), rS2, q01);
                                const f32x2 o23 = __builtin_elementwise_fma(__builtin_elementwise_fma(q23, nS2, c23), rS2, q23);
                                f.x = o01.x; f.y = o01.y; f.z = o23.x; f.w = o23.y;
                            } else {
                                f.x = (float)((double)h.x / Sd); f.y = (float)((double)h.y / Sd);
                                f.z = (float)((double)h.z / Sd); f.w = (float)((double)h.w / Sd);
                            }
                            *(float4 *)((float *)a.out + out_base + i4 * 4) = f;
                        }
                    };
                    if (B16) {
                        for (int i8 = lane; i8 < F / 8; i8 += NT) {
                            const uint4 h = *(const uint4 *)(hist + i8 * 4);
                            emit(2 * i8, make_uint4(h.x & 0xFFFFu, h.x >> 16, h.y & 0xFFFFu, h.y >> 16));
                            emit(2 * i8 + 1, make_uint4(h.z & 0xFFFFu, h.z >> 16, h.w & 0xFFFFu, h.w >> 16));
                        }
                    } else {
                        for (int i4 = lane; i4 < F / 4; i4 += NT) emit(i4, *(const uint4 *)(hist + i4 * 4));
                    }
                } else {
                    for (int i = lane; i < F; i += NT) {
                        const uint32_t src = (a.mode == IDL_MODE_CGR) ? cgr_pixel_to_kmer<K>((uint32_t)i) : (uint32_t)i;
                        const uint32_t h = T::get(hist, src);
                        if (a.out_kind == IDL_OUT_COUNTS_I32) ((uint32_t *)a.out)[out_base + i] = h;
                        else if (a.out_kind == IDL_OUT_FREQ_F64) ((double *)a.out)[out_base + i] = (double)h / Sd;
                        else ((float *)a.out)[out_base + i] = small ? (float)h / Sf : (float)((double)h / Sd);
                    }
                }
            }
            __syncthreads();

            // ---------------- undo: hist back to H0 for the next view (not needed after the last one)
            if (mutated && vi + 1 < a.n_views) {
                if (rec) T::replay(ne * K, hist, list_old, list_new, lane);
                else if (in_lds) (void)T::template view_pass<false>(a, (const uint32_t *)ed_lds, ne, slot0, nslots, nsc, SC, L, cod, msk, hist, list_old, list_new, 1u, 0xFFFFFFFFu, lane);
                else (void)T::template view_pass<false>(a, eg, ne, slot0, nslots, nsc, SC, L, cod, msk, hist, list_old, list_new, 1u, 0xFFFFFFFFu, lane);
            }
        }
        __syncthreads();
    }
}

// =====================================================================================================
// v3: the software-pipelined form of v2 for sequences that fit one staged super-chunk (the bench shape).
// Measured on v2 (profiles/r02_a_vectorise_ablation.txt): with the row stores removed the launch still takes 1.62 of its
// 1.96 ms, and with counting, deltas and stores ALL removed 0.77 ms remain -- a workgroup spends its time in dependent trips to
// memory (slot range -> packed bases -> edit ranges -> edits), in 40 barriers per sequence, and, because vmcnt retires in
// order, every wait for such a load also drains the row stores issued before it.  v3 removes those:
//   * everything a sequence needs (slot range, length, edit ranges; then packed bases, invalid mask, every view's edits)
//     arrives by LDS-DMA (global_load_lds) issued ONE sequence ahead (meta: two ahead) by wave 0 -- no register, no dependent
//     round trip on the critical path, no ordinary global load in the loop at all; the DMA is retired by a COUNTED vmcnt placed
//     in front of the last view's row stores, so no store is ever waited for;
//   * ONE pass evaluates the (edit, window) pairs of ALL views from the pristine staged copy, while H0 is being counted
//     (LDS atomics commute), and records them as (old bin, new bin) in LDS; a view change is then "undo list a + apply list b"
//     in one phase; the histogram a thread owns is read into registers, so dividing / storing view a overlaps that phase;
//   * 9 barriers per sequence instead of 40; window counts travel through LDS counters instead of block reductions.
// A sequence whose edits / pairs do not fit the LDS tables takes a slower in-kernel path (per-view pair passes, like v2).
// =====================================================================================================
constexpr int V3_MAXV = 8;                      // views
constexpr int V3_META = 8 + 4 * V3_MAXV;        // dwords of a meta ring entry: slot_off[s], slot_off[s+1], lengths[s], pad; per view edit_off[v*n+s], [..+1]
constexpr int V3_VT = 4;                        // dwords of a view-table row: edits, first edit's index in LDS, first pair's index in the list, spare
constexpr int V3_VTAB = (V3_MAXV + 1) * V3_VT;  // row 0 = header: flags (1 fast, 2 edits staged), total pairs, staged slots, clamped length

__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(uintptr_t)p; }   // LDS byte offset = low half of the flat address

// LDS-DMA, hidden from the compiler's vmcnt bookkeeping on purpose (cdna_hip_programming.md 5.7): destination = M0 + lane * size
__device__ __forceinline__ void dma16(const void *gsrc, uint32_t lds_byte)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte) : "memory");
}
__device__ __forceinline__ void dma4(const void *gsrc, uint32_t lds_byte)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// wait until at most `younger` of this wave's vector-memory operations are outstanding (a smaller count is always safe)
__device__ __forceinline__ void vm_wait_at_most(int younger)
{
    if (younger >= 48) vm_wait<48>();
    else if (younger >= 32) vm_wait<32>();
    else if (younger >= 16) vm_wait<16>();
    else if (younger >= 12) vm_wait<12>();
    else if (younger >= 8) vm_wait<8>();
    else if (younger >= 4) vm_wait<4>();
    else if (younger >= 3) vm_wait<3>();
    else if (younger >= 2) vm_wait<2>();
    else if (younger >= 1) vm_wait<1>();
    else vm_wait<0>();
}

// RL (k = 4 only): log2 of the number of COPIES of the histogram.  A 4^4-bin histogram is 1 KB, and an LDS atomic from 64 lanes to
// random bins pays the banks, not the bins: 32 lanes of a group over 32 banks put ~3.5 addresses on the busiest one, and the
// read-modify-write takes the bank for each (measured, k = 5..6: 4.5-6 atomics per clock and CU where the instruction's own issue
// would allow 16).  With bin b of copy c at word b * 2^RL + c and lane l adding to copy l mod 2^RL, the 32 lanes of a group hit
// 32 different banks (RL = 5; two lanes per bank at RL = 4) whatever the bins are -- SURVEY section 7's "per-wave private
// sub-histograms" taken to the lane.  The copies are added up when a memory wave reads a row out (integer sums: order-free, and a
// window may leave through another copy than it entered by, since only the sum is ever read).  Only the COUNT goes to the copies
// (10 000 of a sequence's ~15 000 atomics): behind it the compute waves add the copies up into one plain 4^k-bin histogram -- a
// thread per bin, the copies zeroed on the way -- and the views (undo / apply lists, row reads) live there as they do for k >= 5.
// (The first form kept the views in the copies and let the two memory waves add them up per row: 16 ds_read_b128 per lane and
// view against the other workgroups' atomics, 5.5 k cycles a view -- 1.18 ms per launch with 32 copies against 0.74 with 8.)
template <int K, int RL = 0>
struct V3 {
    using T = V2<K, false>;
    static constexpr int F = T::F, HC = RL ? (F + 4) << RL : 0;           // words of the copies (F bins + 4 garbage bins each)
    static constexpr int HD = HC + T::HD;                                  // ... followed by the histogram proper: F bins + 4 garbage bins
    static constexpr uint32_t KM = T::KM, VM = T::VM;
    static_assert(RL == 0 || (F + 4) * 4 < 65536, "bin byte offsets travel as 16 bits");
    // byte offset of bin (given as its byte offset in ONE copy) in the copy of the lane whose offset is lo = (lane mod 2^RL) * 4
    static __device__ __forceinline__ uint32_t at(uint32_t bin4, uint32_t lo) { return RL ? (bin4 << RL) + lo : bin4; }

    // All K windows of ONE edit: the windows ending at p .. p + K - 1 that this edit owns (it is the latest edit at or before
    // their end: w <= min(p + K - 1, next edit - 1, last base)).  One 32-base fetch serves the K old bins; the XORs / N flags of
    // the earlier edits that still reach into those windows (rare) are gathered once as X (2 bits per base back from p) and
    // Dm (1 bit per base), so window t changes by (X << 2t) & KM and is dead when (Dm << t) & VM.  Per window: the BYTE offsets of
    // the old and the new bin in the histogram, old | new << 16, a missing / invalid window pointing at a garbage bin behind the
    // histogram (so a list entry is always two unconditional LDS adds); returns (valid new windows) - (valid old windows).
    template <typename EP>
    static __device__ __forceinline__ int edit_eval(EP E, int ne, int ei, int end_r, const uint32_t *cod, const uint32_t *msk, uint32_t garbage,
                                                    uint32_t (&pr)[K])
    {
        const uint32_t ed = E[ei];
        const int p = (int)(ed & 0x3FFFFFFFu);
        int last = p + K - 1;
        if (ei + 1 < ne) { const int nx = (int)(E[ei + 1] & 0x3FFFFFFFu) - 1; if (nx < last) last = nx; }
        if (last > end_r) last = end_r;
        uint32_t X = ed >> 30, Dm = (ed >> 30) == 0u ? 1u : 0u;
        for (int i = ei - 1; i >= 0; --i) {
            const uint32_t e2 = E[i];
            const int d = p - (int)(e2 & 0x3FFFFFFFu);
            if (d >= K) break;
            X ^= (e2 >> 30) << (2 * d);
            Dm |= ((e2 >> 30) == 0u ? 1u : 0u) << d;
        }
        // 32 staged bases ending at the end of the word that holds base min(p + K - 1, last base) (staged index = position + 64: slot 0 is the halo)
        const int pt = (p + K - 1 < end_r) ? p + K - 1 : end_r;    // never fetch past the last base (positions are < L, so pt >= p)
        const uint32_t top = (uint32_t)pt + 64u;
        const int D = (int)(top >> 4);
        uint64_t ww; uint32_t M;
        T::fetch(cod, msk, D, ww, M);
        const int e0 = 15 - (int)(top & 15u) + (pt - p);           // bases between p and the end of the fetched word
        int dw = 0;
#pragma unroll
        for (int t = 0; t < K; ++t) {
            uint32_t ao = garbage, an = garbage;
            const int sh = (e0 - t) & 31;                            // window t ends sh bases before the end of the word (negative only past `last`)
            if (p + t <= last && ((M >> sh) & VM) == 0u) {
                const uint32_t ko = (uint32_t)(ww >> (2 * sh)) & KM;
                ao = ko << 2;
                --dw;
                if (((Dm << t) & VM) == 0u) { an = (ko ^ ((X << (2 * t)) & KM)) << 2; ++dw; }
            }
            pr[t] = ao | (an << 16);
        }
        return dw;
    }

    // one list entry: the window leaves the bin at byte offset (e & 0xFFFF) and enters the one at (e >> 16) (sign = 1), or back
    static __device__ __forceinline__ void move(uint32_t *hist, uint32_t e, uint32_t sign)
    {
        atomicAdd((uint32_t *)((char *)hist + (e & 0xFFFFu)), 0u - sign);
        atomicAdd((uint32_t *)((char *)hist + (e >> 16)), sign);
    }
    // ... into the copies (while the sequence is being counted): this lane's copy
    static __device__ __forceinline__ void move_copy(uint32_t *copies, uint32_t e, uint32_t sign, uint32_t lo)
    {
        atomicAdd((uint32_t *)((char *)copies + at(e & 0xFFFFu, lo)), 0u - sign);
        atomicAdd((uint32_t *)((char *)copies + at(e >> 16, lo)), sign);
    }
    // bin `b` of the histogram proper = pseudocount + the sum of its copies, which are zeroed (RL > 0; a thread per bin).  The 2^RL
    // copies of a bin are NPB 16-byte pieces; LPR lanes share a 256-byte bank row, so lane l starts at piece l / LPR: every
    // ds_read_b128 lane group covers 16 different slots
    static __device__ __forceinline__ void fold_copies(uint32_t *copies, uint32_t *hist, int b, uint32_t iv)
    {
        if constexpr (RL > 0) {
            constexpr int NPB = (1 << RL) / 4, LPR = 64 >> RL;       // (RL = 2: one piece per bin, 16 lanes to a bank row: no rotation needed)
            uint32_t *base = copies + ((uint32_t)b << RL);
            const int rot = (b & 63) / (LPR > 0 ? LPR : 1);
            uint32_t sum = iv;
#pragma unroll
            for (int j = 0; j < NPB; ++j) {
                const int pc = (j + rot) & (NPB - 1);
                const uint4 h = *(const uint4 *)(base + pc * 4);
                sum += (h.x + h.y) + (h.z + h.w);
                *(uint4 *)(base + pc * 4) = make_uint4(0u, 0u, 0u, 0u);
            }
            hist[b] = sum;
        }
    }

    // every window ending in the staged sequence (v2's count_all without its barrier); returns this thread's valid windows
    template <int NT>
    static __device__ __forceinline__ uint32_t count_all(const uint32_t *cod, const uint32_t *msk, int nloc, uint32_t *hist, int tid)
    {
        uint32_t cnt = 0;
        const uint32_t lo = RL ? ((uint32_t)tid & ((1u << RL) - 1u)) << 2 : 0u;
        const int nd = nloc * 4;
        for (int d0 = 0; d0 < nd; d0 += NT) {
            const int d = d0 + tid;
            if (d < nd) {
                uint64_t w; uint32_t M;
                T::fetch(cod, msk, 4 + d, w, M);
                uint32_t inv = M;
#pragma unroll
                for (int t = 1; t < K; ++t) inv |= (M >> t);
                inv &= 0xFFFFu;
                // byte offset of window j's bin: ((prev:cur) >> (30 - 2j)) & KM, times 4 -- one funnel shift + one AND per window
                const uint32_t hi = (uint32_t)(w >> 32), lw = (uint32_t)w;
                constexpr uint32_t KM4 = KM << 2;
                if (__ballot(inv != 0u) == 0ull) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const uint32_t ad = (j < 15 ? __builtin_amdgcn_alignbit(hi, lw, 28 - 2 * (j < 15 ? j : 0)) : (lw << 2)) & KM4;
                        atomicAdd((uint32_t *)((char *)hist + at(ad, lo)), 1u);
                    }
                } else {
                    const uint32_t garbage = ((uint32_t)F + (tid & 3)) << 2;
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const uint32_t ad = (j < 15 ? __builtin_amdgcn_alignbit(hi, lw, 28 - 2 * (j < 15 ? j : 0)) : (lw << 2)) & KM4;
                        atomicAdd((uint32_t *)((char *)hist + at(((inv >> (15 - j)) & 1u) ? garbage : ad, lo)), 1u);
                    }
                }
                cnt += 16u - (uint32_t)__popc(inv);
            }
        }
        return cnt;
    }
};

// Workgroup = 4 COMPUTE waves (LDS only: counting, edit evaluation, list phases; they never issue a vector-memory instruction)
// + 1 MEMORY wave (every LDS-DMA load of the next sequence and every row store: it reads the finished histogram into
// registers between two barriers, then converts and stores while the compute waves are already moving the histogram to the
// next view).  A full store queue -- the chip's write stream is the bound of this kernel -- therefore stalls only the wave that
// has nothing else to do, and the DMA waits of that wave never queue behind another wave's work.
constexpr int V3_NC = 384;                      // compute threads (6 waves)
constexpr int V3_MW = 2;                        // memory waves, each holding 1 / V3_MW of a row
constexpr int V3_NT = V3_NC + 64 * V3_MW;

template <int K, bool DIAG, int RL>
__global__ __launch_bounds__(V3_NT, 8) void vectorise3_kernel(VecArgs a)
{
    using W = V3<K, RL>;
    // RP: store instructions per row and memory wave: a lane holds RP 16-byte pieces of the row.  4^k in 512..4096: each memory wave
    // its part of every row.  4^k = 256 (k = 4): a row is ONE wave's 16 bytes a lane, and the memory waves take the views in turn.
    constexpr int F = W::F, HD = W::HD, HC = W::HC, NC = V3_NC, RP = (F == 256) ? 1 : (F / 4) / 64 / V3_MW;
    constexpr bool TURNS = F == 256;
    static_assert(TURNS || ((F / 4) % (64 * V3_MW) == 0 && RP >= 1 && RP <= 16), "a row is held by the memory waves: 4^k in 256..4096");
    static_assert(RL == 0 || (RL >= 2 && RL <= 5), "2^RL copies of the histogram: 4 .. 32");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int SC = a.sc_slots, P = a.n_views;
    const int SET = (SC + 1) * 6 + a.ecap;                  // words of one staging set
    uint32_t *copies = lds;                                 // RL > 0: 2^RL copies of (F bins + 4 garbage bins), the COUNT's target
    uint32_t *hist = lds + HC;                              // F bins + 4 garbage bins
    uint32_t *sets = lds + HD;                              // two staging sets (sequence it lives in set it & 1), each:
    //   cod  (SC + 1) slots x 4 words, slot 0 = halo (zeros) | msk  (SC + 1) slots x 2 words, slot 0 = halo (all invalid) | edl  a.ecap edits of all views
    uint32_t *list = sets + 2 * SET;                        // a.lcap pairs: old bin | new bin << 16
    uint32_t *meta = list + a.lcap;                         // ring of 3 entries
    uint32_t *vtab = meta + 3 * V3_META;                    // two view tables (this sequence / the next)
    int32_t *ctr = (int32_t *)(vtab + 2 * V3_VTAB);         // [0] valid windows of the un-mutated sequence, [1 + v] window delta of view v
    volatile int32_t *qb = ctr + 16;                        // first sequence of the last four batches this workgroup pulled
    const int tid = threadIdx.x, lane = tid & 63;
    const int mw = __builtin_amdgcn_readfirstlane(tid >> 6) - NC / 64;         // >= 0: memory wave number
    const bool mem = mw >= 0;
    const uint32_t iv = (a.init == IDL_INIT_ONE) ? 1u : 0u;
    const bool has_edits = a.edits != nullptr;
    const uint32_t lo4 = RL ? ((uint32_t)tid & ((1u << RL) - 1u)) << 2 : 0u;      // byte offset of this lane's copy inside a bin's group of copies

    // this workgroup's sequences: batches of a.chunk consecutive sequences pulled from a global counter (one returning atomic per
    // batch, issued by a compute wave -- they have no other vector-memory traffic -- three sequences before the batch is needed).
    // Dynamic: the rows being written chip-wide stay in a narrow window, every workgroup's row stores walk through C rows of every
    // view in order, and no workgroup ends more than a sequence after another (+ 10-15 % store rate over a fixed stride on
    // boxes with slow HBM: tools/store_pattern.hip).  Iteration i of this workgroup is sequence qb[(i / C) & 3] + i % C.
    const int QC = a.chunk > 0 ? a.chunk : 1;
    int *qhead = a.redo_count + 1;
    auto seq_index = [&](int64_t i) -> int64_t { return (int64_t)qb[(i / QC) & 3] + i % QC; };
    // ---------------- memory wave: slot range, length and edit ranges of sequence s -> meta ring entry r (one DMA instruction)
    auto dma_meta = [&](int64_t s, int r) {
        int l = lane;
        asm volatile("" : "+v"(l));            // keeps the per-lane source address from being hoisted out of the sequence loop (and spilled there)
        const uint32_t *src = nullptr;
        if (l < 4) src = (const uint32_t *)(a.slot_off + s) + l;
        else if (l < 6) src = (const uint32_t *)(a.lengths + s) + (l - 4);
        else if (l >= 8 && l < 8 + 4 * P && has_edits) src = (const uint32_t *)(a.edit_off + (((int64_t)((l - 8) >> 2) * a.n + s) << a.eo_shift)) + ((l - 8) & 3);
        if (src != nullptr) dma4(src, __builtin_amdgcn_readfirstlane(lds_addr(meta + r * V3_META)));
    };
    // ---------------- memory wave: view table of the sequence described by ring entry r + the DMA of its packed bases, mask and
    // edits into staging set `st`.  A sequence that does not fit the tables is only flagged (the launcher's second pass takes it).
    auto stage_next = [&](int r, uint32_t *vt, uint32_t *st) {
        int ln = lane;
        asm volatile("" : "+v"(ln));           // per-lane source addresses are formed here, per call: hoisted out of the sequence loop they are spilled
        uint32_t *cod = st, *msk = st + (SC + 1) * 4, *edl = st + (SC + 1) * 6;
        const uint32_t *M = meta + r * V3_META;
        const int64_t slot0 = (int64_t)(((uint64_t)M[1] << 32) | M[0]);
        const int64_t L64 = (int64_t)(((uint64_t)M[5] << 32) | M[4]);
        const int64_t nsl64 = (L64 + 63) >> 6;          // ceil(len / 64) slots from slot_off[s]: records need not be adjacent in the packed buffers
        int64_t eb = 0;
        int ne = 0;
        if (ln < P) {
            eb = (int64_t)(((uint64_t)M[9 + 4 * ln] << 32) | M[8 + 4 * ln]);
            const int64_t d = (int64_t)(((uint64_t)M[11 + 4 * ln] << 32) | M[10 + 4 * ln]) - eb;
            ne = (int)(d < 0 ? 0 : (d > 0x3FFFFFF ? 0x3FFFFFF : d));
        }
        int incl = ne;
#pragma unroll
        for (int o = 1; o < V3_MAXV; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (ln >= o) incl += t; }
        const int eoff = incl - ne;
        const int total_e = __shfl(incl, P - 1, 64);
        const bool fast = total_e <= a.ecap && (int64_t)total_e * K <= (int64_t)a.lcap && nsl64 >= 0 && nsl64 <= (int64_t)SC &&
                          L64 >= 0 && L64 <= nsl64 * 64;
        const int nslots = fast ? (int)nsl64 : 0;
        if (mw == 0) {
            if (ln < P) { vt[(1 + ln) * V3_VT] = (uint32_t)ne; vt[(1 + ln) * V3_VT + 1] = (uint32_t)eoff; vt[(1 + ln) * V3_VT + 2] = (uint32_t)(eoff * K); }
            // the view with the most edits goes last (its list is applied once and never undone): found here, once per sequence
            int best = ne, vl = ln < P ? ln : 0;
            if (ln >= P) best = -1;
#pragma unroll
            for (int o = 1; o < V3_MAXV; o <<= 1) {
                const int b2 = __shfl_xor(best, o, 64), v2 = __shfl_xor(vl, o, 64);
                if (b2 > best || (b2 == best && v2 < vl)) { best = b2; vl = v2; }
            }
            if (ln == 0) { vt[0] = (fast ? 1u : 0u) | ((uint32_t)vl << 8); vt[1] = (uint32_t)(total_e * K); vt[2] = (uint32_t)nslots; vt[3] = (uint32_t)(fast ? L64 : 0); }
        }
        if (!fast) return;
        if (mw == 0) {
            for (int i0 = 0; i0 < nslots; i0 += 64)
                if (i0 + ln < nslots) dma16(a.codes + slot0 + i0 + ln, __builtin_amdgcn_readfirstlane(lds_addr(cod + 4 + i0 * 4)));
            if (V3_MW > 1) return;
        }
        for (int i0 = 0; i0 < 2 * nslots; i0 += 64)
            if (i0 + ln < 2 * nslots) dma4((const uint32_t *)(a.mask + slot0) + i0 + ln, __builtin_amdgcn_readfirstlane(lds_addr(msk + 2 + i0)));
        for (int v = 0; v < P; ++v) {
            const int nev = __shfl(ne, v, 64), eov = __shfl(eoff, v, 64);
            const int64_t ebv = (int64_t)(((uint64_t)(uint32_t)__shfl((int)(eb >> 32), v, 64) << 32) | (uint32_t)__shfl((int)eb, v, 64));
            for (int i0 = 0; i0 < nev; i0 += 64)
                if (i0 + ln < nev) dma4(a.edits + ebv + i0 + ln, __builtin_amdgcn_readfirstlane(lds_addr(edl + eov + i0)));
        }
    };
    // ---------------- compute waves
    auto clear_hist = [&]() {
        // (RL > 0: the histogram proper is written whole by fold_copies, which also zeroes the copies it reads)
        if constexpr (RL == 0) { for (int i = tid; i < (F + 4) / 4; i += NC) *(uint4 *)(hist + i * 4) = make_uint4(iv, iv, iv, iv); }
        if (tid <= V3_MAXV) ctr[tid] = 0;
    };

    // ---------------- memory wave: a view's row, in registers between reading the histogram and storing it
    struct Row { uint4 h[RP]; int64_t S, s; int v; };
    auto row_load = [&](Row &rw, int v, int64_t s) {
#pragma unroll
        for (int j = 0; j < RP; ++j) rw.h[j] = *(const uint4 *)(hist + (lane + ((TURNS ? 0 : mw * RP) + j) * 64) * 4);
        rw.S = (int64_t)ctr[0] + (int64_t)ctr[1 + v] + (iv ? (int64_t)F : 0);
        rw.v = v; rw.s = s;
    };
    auto row_store = [&](const Row &rw) {
        if (a.ablate & 1) { if (rw.h[0].x != 0xFFFFFFF0u) return; }
        // one address, formed here (not where the row was loaded: sixteen 64-bit addresses held across the barrier cost 32 registers);
        // the pieces are 1 KB apart
        int l4 = lane * 4;
        asm volatile("" : "+v"(l4));
        float *dst = (float *)a.out + ((int64_t)rw.v * a.view_stride + rw.s * (int64_t)F) + (TURNS ? 0 : mw * RP * 256) + l4;
        if (a.out_kind == IDL_OUT_COUNTS_I32) {
#pragma unroll
            for (int j = 0; j < RP; ++j) *(uint4 *)((uint32_t *)dst + j * 256) = rw.h[j];
            return;
        }
        // S <= max_len + 4^k < 2^24 in this kernel (the launcher bounds max_len): ints are exact in float32 and the Markstein form
        // q = c * r; q' = fma(fma(-q, S, c), r, q) equals float32(float64 division) (tools/markstein_check.c); two bins per packed op
        const float Sf = (float)rw.S, rS = 1.0f / Sf;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 rS2 = {rS, rS}, nS2 = {-Sf, -Sf};
#pragma unroll
        for (int j = 0; j < RP; ++j) {
            const uint4 h = rw.h[j];
            const f32x2 c01 = {(float)h.x, (float)h.y}, c23 = {(float)h.z, (float)h.w};
            const f32x2 q01 = c01 * rS2, q23 = c23 * rS2;
            const f32x2 o01 = __builtin_elementwise_fma(__builtin_elementwise_fma(q01, nS2, c01), rS2, q01);
            const f32x2 o23 = __builtin_elementwise_fma(__builtin_elementwise_fma(q23, nS2, c23), rS2, q23);
            *(float4 *)(dst + j * 256) = make_float4(o01.x, o01.y, o23.x, o23.y);
        }
    };

    // ---------------- prologue: halos, tables, the first two meta entries, the first sequence
    if (tid < 8) sets[(tid >> 2) * SET + (tid & 3)] = 0u;
    if (tid < 4) sets[(tid >> 1) * SET + (SC + 1) * 4 + (tid & 1)] = 0xFFFFFFFFu;
    for (int i = tid; i < 3 * V3_META + 2 * V3_VTAB; i += V3_NT) meta[i] = 0u;
    if (!mem) clear_hist();
    if constexpr (RL > 0) { for (int i = tid; i < HC / 4; i += V3_NT) *(uint4 *)(copies + i * 4) = make_uint4(0u, 0u, 0u, 0u); }
    const int QPRO = QC == 1 ? 3 : 2;             // batches pulled up front: iterations 0..2 must be known before the loop starts
    if (tid == 0) { for (int j = 0; j < 4; ++j) qb[j] = j < QPRO ? atomicAdd(qhead, QC) : 0x7FFFFFFF; }
    __syncthreads();
    if (mw == 0 && seq_index(0) < a.n) {
        dma_meta(seq_index(0), 0);
        if (seq_index(1) < a.n) dma_meta(seq_index(1), 1);
        vm_wait<0>();
    }
    __syncthreads();
    if (mem && seq_index(0) < a.n) { stage_next(0, vtab, sets); vm_wait<0>(); }
    __syncthreads();

    // diagnostic stamps (DIAG build, a.dbg != NULL): cycles per phase, summed over this workgroup's sequences, compute wave 0 and the memory wave
    unsigned long long tp[DIAG ? 10 : 1] = {0}, tl = 0;
    const bool stamp = DIAG && a.dbg != nullptr && (tid < 64 || mw == 0);
    auto mark = [&](int k) { if constexpr (DIAG) { if (stamp) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tp[k] += t - tl; tl = t; } } };
    if constexpr (DIAG) { if (stamp) tl = __builtin_amdgcn_s_memtime(); }

    // what both roles read off the view table of the current sequence (the same values: the barrier schedule must match)
    struct Seq { const uint32_t *vt; int TP, nslots, L, vlast; bool fast; };
    auto seq_of = [&](int64_t it) -> Seq {
        Seq q;
        q.vt = vtab + (it & 1) * V3_VTAB;
        q.fast = (q.vt[0] & 1u) != 0u;
        q.TP = (a.ablate & 4) ? 0 : (int)q.vt[1]; q.nslots = (int)q.vt[2]; q.L = (int)q.vt[3];
        q.vlast = (int)((q.vt[0] >> 8) & 0xFFu);      // the view with the most edits goes last (stage_next)
        return q;
    };
    auto view_at = [&](const Seq &q, int vi) -> int { return (vi == P - 1) ? q.vlast : (vi < q.vlast ? vi : vi + 1); };

    // Barrier schedule of one sequence, identical in both roles: [P1 | stage the next sequence] B { [row load] B [lists | stores] B } x P;
    // a sequence left to the second pass: [stage the next sequence] B.
    if (mem) {
        // =============================================================== the memory wave
        int r = 0;
        for (int64_t it = 0;; ++it) {
            const int64_t s = seq_index(it);
            if (s >= a.n) break;
            const Seq q = seq_of(it);
            const int r1 = (r + 1 == 3) ? 0 : r + 1, r2 = (r1 + 1 == 3) ? 0 : r1 + 1;
            // while the compute waves count this sequence: the next sequence's data and the one after's meta leave for LDS (the other
            // staging set was last read one sequence ago)
            if (seq_index(it + 1) < a.n) stage_next(r1, vtab + ((it + 1) & 1) * V3_VTAB, sets + ((it + 1) & 1) * SET);
            if (mw == 0) { const int64_t s2 = seq_index(it + 2); if (s2 < a.n) dma_meta(s2, r2); }
            mark(5);
            if (q.fast) {
                __syncthreads();                                // P1 of the compute waves is over
                if constexpr (RL > 0) __syncthreads();          // ... and they have folded the copies into the histogram
                mark(2);
                for (int vi = 0; vi < P; ++vi) {
                    const bool mine = !TURNS || (vi & (V3_MW - 1)) == mw;      // k = 4: the memory waves take the views in turn
                    Row row;
                    if (mine) row_load(row, view_at(q, vi), s);
                    mark(3);
                    __syncthreads();
                    mark(4);
                    // before the last row goes out, retire the DMA issued above: the younger operations are the row stores this wave
                    // has issued since, so no store is waited for
                    if (vi + 1 == P) vm_wait_at_most((a.ablate & 1) ? 0 : (TURNS ? (P - 1 + (V3_MW - 1 - mw)) / V3_MW : (P - 1) * RP));
                    if (mine) row_store(row);
                    mark(6);
                    __syncthreads();
                    mark(7);
                }
            } else {
                if (lane == 0 && mw == 0) {                                   // the launcher's second pass (v2) takes this sequence
                    const int slot = atomicAdd(a.redo_count, 1);
                    if (slot < V3_REDO_CAP) ((int64_t *)(a.redo_count + 2))[slot] = s;
                }
                vm_wait<0>();
                __syncthreads();
            }
            r = r1;
        }
    } else {
        // =============================================================== the compute waves
        for (int64_t it = 0;; ++it) {
            if (seq_index(it) >= a.n) break;
            const Seq q = seq_of(it);
            // the batch that iteration it + 3 starts is pulled now and published by this sequence's first barrier
            const bool grab = tid == 0 && (it + 3) % QC == 0 && (it + 3) / QC >= QPRO;
            int grabbed = 0;
            if (grab) grabbed = atomicAdd(qhead, QC);
            if (!q.fast) { if (grab) qb[((it + 3) / QC) & 3] = grabbed; __syncthreads(); continue; }
            const uint32_t *vt = q.vt;
            const uint32_t *cod = sets + (it & 1) * SET, *msk = cod + (SC + 1) * 4, *edl = cod + (SC + 1) * 6;
            // ---------------- P1: count the un-mutated sequence, evaluate every edit of every view
            {
                uint32_t c0 = (a.ablate & 2) ? 0u : W::template count_all<NC>(cod, msk, q.nslots, RL ? copies : hist, tid);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) c0 += __shfl_xor(c0, o, 64);
                if (lane == 0) atomicAdd(&ctr[0], (int32_t)c0);
            }
            mark(0);
            {
                const int vfirst = view_at(q, 0);
                const int TE = q.TP / K;                     // edits of all views; lane per edit, K list entries each
                for (int e0 = 0; e0 < TE; e0 += NC) {
                    const int e = e0 + tid;
                    if (e < TE) {
                        int v = 0;
                        for (int u = 1; u < P; ++u) if (e >= (int)vt[(1 + u) * V3_VT + 1]) v = u;      // edit bases ascend with the view index
                        const int ne = (int)vt[(1 + v) * V3_VT], eo = (int)vt[(1 + v) * V3_VT + 1];
                        uint32_t pr[K];
                        const int d = W::edit_eval((const uint32_t *)(edl + eo), ne, e - eo, q.L - 1, cod, msk, ((uint32_t)F + (tid & 3)) << 2, pr);
#pragma unroll
                        for (int t = 0; t < K; ++t) list[e * K + t] = pr[t];
                        if (d != 0) atomicAdd(&ctr[1 + v], d);
                        if (v == vfirst) {
#pragma unroll
                            for (int t = 0; t < K; ++t) { if constexpr (RL > 0) W::move_copy(copies, pr[t], 1u, lo4); else W::move(hist, pr[t], 1u); }
                        }
                    }
                }
            }
            if (grab) qb[((it + 3) / QC) & 3] = grabbed;
            mark(1);
            __syncthreads();
            if constexpr (RL > 0) {          // the copies -> the histogram the views live in (+ pseudocount), a thread per bin
                for (int b = tid; b < F; b += NC) W::fold_copies(copies, hist, b, iv);
                __syncthreads();
            }
            mark(2);
            for (int vi = 0; vi < P; ++vi) {
                mark(3);
                __syncthreads();            // the memory wave has the row of view vi in registers
                mark(4);
                if (vi + 1 < P) {           // hist: view v -> view v2 in one phase (LDS atomics commute)
                    const int v = view_at(q, vi), v2 = view_at(q, vi + 1);
                    const int na = (a.ablate & 4) ? 0 : (int)vt[(1 + v) * V3_VT] * K, pa = (int)vt[(1 + v) * V3_VT + 2];
                    const int nb = (a.ablate & 4) ? 0 : (int)vt[(1 + v2) * V3_VT] * K, pb = (int)vt[(1 + v2) * V3_VT + 2];
                    for (int x = tid; x < na; x += NC) W::move(hist, list[pa + x], 0xFFFFFFFFu);       // undo view v
                    for (int x = tid; x < nb; x += NC) W::move(hist, list[pb + x], 1u);                // apply view v2
                } else {
                    clear_hist();
                }
                mark(6);
                __syncthreads();
                mark(7);
            }
        }
    }
    if constexpr (DIAG) {
        if (stamp && lane == 0) {
            unsigned long long *d = a.dbg + ((int64_t)blockIdx.x * 2 + (mem ? 1 : 0)) * 10;
            for (int k = 0; k < 10; ++k) d[k] = tp[k];
        }
    }
}

// =====================================================================================================
// v4 (round 6): ONE WAVEFRONT PER SEQUENCE for the small histograms, k = 4 and k = 5 (north_star's literal design).
// v3's workgroup-per-sequence pipeline pays ~9 barriers and a DMA hand-over per sequence: at k = 6 they hide under 64 KB of row
// stores, at k = 4 (a 1 KB row) they ARE the kernel -- 18.1 k cycles a sequence of which ~4 k barriers and 7 k the memory waves' DMA
// issue, against 3.7 k if the LDS atomics were the bound.  Here a wave owns a sequence from its packed bases to its rows:
//   * its slice of LDS holds the staged sequence (halo slot + ceil(L / 64) slots), the edits of all views, the histogram, and at k = 4 the
//     2^RL per-lane copies the count goes to -- no list of (old bin, new bin) pairs: a view is undone by evaluating its edits again (V3<K, RL>'s helpers: the same count, edit
//     evaluation, list entries and fold as v3, so the rows are v3's bit for bit);
//   * no s_barrier anywhere: the LDS unit executes a wave's operations in order, a wave-level fence keeps the compiler's order;
//   * the NEXT sequence's packed bases, mask and edits are requested (into registers) behind the count of the current one and written
//     to LDS when the current one's last row is out: the trip to memory hides under the view phases;
//   * a view: evaluate its edits (a lane per edit, K list entries each), apply, read the row out of LDS, convert (Markstein, as v3), store,
//     undo from the list -- the last view is not undone (the next sequence's fold / clear rewrites the histogram).
// Sequences are dealt round-robin over the launch's waves.  A sequence that does not fit the slice (v3's predicate: edits, pairs,
// slots) is listed for the second pass on v2, as v3 does.
// =====================================================================================================
constexpr int V4_MAX_WAVES = 12;                // waves of a workgroup: as many as the slices of LDS allow (the launcher decides)
constexpr int V4_SR = 3;                        // rounds of 64 slots a sequence may take (12 288 bases); longer inputs stay on v3 / v2
constexpr int V4_FR = 10;                       // rounds of 64 edits (ALL views of a sequence, one after the other) prefetched into registers; more: second pass

// OM ("other modes"): the CGR and canonical rows (reference utils.py:279-317, 208-221 -- round 6) as epilogues over the finished histogram, in an
// instance of their own so that their registers (a lane's 4^k / 64 collapsed bins) do not count against the plain rows'.
template <int K, int RL, bool OM = false>
__global__ __launch_bounds__(64 * V4_MAX_WAVES) void vectorise4_kernel(VecArgs a)
{
    using W = V3<K, RL>;
    constexpr int F = W::F, HC = W::HC, RP = F / 256;
    constexpr int ROW_CANON = (K % 2 == 0) ? (F + (1 << K)) / 2 : F / 2;
    const int64_t row_len = (OM && a.mode == IDL_MODE_CANONICAL) ? ROW_CANON : F;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int P = a.n_views, SC = a.v3_sc;
    const int slice = (HC + (F + 4) + (SC + 1) * 6 + a.ecap + 3) & ~3;
    uint32_t *copies = lds + (size_t)wv * slice, *hist = copies + HC, *cod = hist + F + 4, *msk = cod + (SC + 1) * 4, *E = msk + (SC + 1) * 2;
    const uint32_t iv = (a.init == IDL_INIT_ONE) ? 1u : 0u;
    const uint32_t garbage = ((uint32_t)F + (lane & 3)) << 2;
    for (int i = lane; i < HC; i += 64) copies[i] = 0u;
    if (lane < 4) cod[lane] = 0u;                            // the halo slot: no bases before the sequence
    if (lane < 2) msk[lane] = 0xFFFFFFFFu;
    // the sum over the wave without a trip through the LDS crossbar (six ds_bpermute in a row cost ~700 cycles, five times a sequence): rows of 16 by
    // DPP (quad swaps, half-row mirror, row mirror), the four rows by readlane.  Every lane is active where this is called.
    auto wave_sum = [](int v) {
        v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);
        return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
    };
    auto fence = []() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    // what a sequence brings: metadata (uniform), then its packed bases / mask / edits in registers until the slice is free
    struct Meta { int64_t s, slot0, eb[V3_MAXV]; int L, nslots, ne[V3_MAXV], te, fast; };
    struct Regs { uint4 c[V4_SR]; uint2 m[V4_SR]; uint32_t e[V4_FR]; };
    // (sequences are dealt round-robin over all waves of the launch: a shared counter -- one returning atomic per sequence on ONE address -- serialised the
    //  launch at ~12 ns a sequence: 1.2 ms for 100 000)
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    int64_t next_s = (int64_t)blockIdx.x * (blockDim.x >> 6) + wv;
    auto pull = [&](Meta &q) {
        q.s = next_s < a.n ? next_s : a.n;
        next_s += n_waves;
        q.fast = 0; q.te = 0; q.L = 0; q.nslots = 0; q.slot0 = 0;
        if (q.s >= a.n) return;
        q.slot0 = a.slot_off[q.s];
        const int64_t L = a.lengths[q.s];
        const int64_t nslots = (L + 63) >> 6;
        int64_t te = 0;
#pragma unroll
        for (int v = 0; v < V3_MAXV; ++v) {
            q.eb[v] = 0; q.ne[v] = 0;
            if (v < P && a.edits != nullptr) {
                const int64_t b = eo_begin(a, (int64_t)v * a.n + q.s), e = eo_end(a, (int64_t)v * a.n + q.s);
                const int64_t d = e - b < 0 ? 0 : (e - b > 0x3FFFFFF ? 0x3FFFFFF : e - b);
                q.eb[v] = b; q.ne[v] = (int)d; te += d;
            }
        }
        // (restated by vectorise2_kernel's second pass, redo == 2; the launcher keeps ecap <= 64 * V4_FR)
        const bool ok = nslots <= (int64_t)SC && nslots <= 64 * V4_SR && L >= 0 && te <= (int64_t)a.ecap;
        q.L = (int)L; q.nslots = (int)nslots; q.te = (int)te; q.fast = ok ? 1 : 0;
    };
    // (every load unconditional within its uniform branch, its index clamped: a load under a per-lane condition is put into a branch of its own and waited
    //  for before the next is issued -- 35 dependent trips to memory a sequence in this lambda's first form)
    auto request = [&](const Meta &q, Regs &r) {
        if (!q.fast) return;
        const int ns1 = q.nslots > 0 ? q.nslots - 1 : 0;
#pragma unroll
        for (int i = 0; i < V4_SR; ++i) {
            const int slot = i * 64 + lane;
            const int64_t g = q.slot0 + (slot < ns1 ? slot : ns1);
            r.c[i] = a.codes[g]; r.m[i] = a.mask[g];
        }
        if (q.te > 0) {
            // the edits of the views follow one another in the slice: flat index f -> (view, index in the view) by a chain of selects
#pragma unroll
            for (int j = 0; j < V4_FR; ++j) {
                int f = j * 64 + lane;
                f = f < q.te ? f : q.te - 1;
                int64_t base = q.eb[0];
                int start = 0, cum = q.ne[0];
#pragma unroll
                for (int v = 1; v < V3_MAXV; ++v) {
                    const bool in = f >= cum;
                    base = in ? q.eb[v] : base; start = in ? cum : start;
                    cum += q.ne[v];
                }
                r.e[j] = a.edits[base + (f - start)];
            }
        }
    };
    auto deposit = [&](const Meta &q, const Regs &r) {       // registers -> the slice (the previous sequence is done with it)
        if (!q.fast) return;
#pragma unroll
        for (int i = 0; i < V4_SR; ++i) {
            const int slot = i * 64 + lane;
            if (slot < q.nslots) { *(uint4 *)(cod + (slot + 1) * 4) = r.c[i]; *(uint2 *)(msk + (slot + 1) * 2) = r.m[i]; }
        }
#pragma unroll
        for (int j = 0; j < V4_FR; ++j) { const int f = j * 64 + lane; if (f < q.te) E[f] = r.e[j]; }
    };
    Meta cur, nxt;
    Regs rg;
    pull(cur);
    request(cur, rg);
    while (cur.s < a.n) {
        pull(nxt);                                           // (its loads are uniform and wanted only behind the count below)
        if (!cur.fast) {                                     // the launcher's second pass (v2) takes this sequence
            if (lane == 0) { const int slot = atomicAdd(a.redo_count, 1); if (slot < V3_REDO_CAP) ((int64_t *)(a.redo_count + 2))[slot] = cur.s; }
            request(nxt, rg);
            cur = nxt;
            continue;
        }
        deposit(cur, rg);
        if constexpr (RL == 0) { for (int i = lane; i < (F + 4) / 4; i += 64) *(uint4 *)(hist + i * 4) = make_uint4(iv, iv, iv, iv); }
        fence();
        // ---------------- the un-mutated sequence, counted once
        const int windows = wave_sum((int)W::template count_all<64>(cod, msk, cur.nslots, RL ? copies : hist, lane));
        request(nxt, rg);                                    // the next sequence's trip to memory hides under the views of this one
        if constexpr (RL > 0) { for (int b = lane; b < F; b += 64) W::fold_copies(copies, hist, b, iv); }
        fence();
        // the un-mutated histogram stays in registers (four bins a lane and 256): a view is undone by writing it back -- RP conflict-free stores
        // instead of the view's atomics a second time.  (k = 6, the CGR / canonical instance only: 64 registers would hold it; there a view is undone by
        // evaluating its edits again with the other sign, and the rows leave piece by piece.)
        constexpr bool KEEP = RP <= 4;
        uint4 h0[KEEP ? RP : 1];
        if constexpr (KEEP) {
#pragma unroll
            for (int j = 0; j < RP; ++j) h0[j] = *(const uint4 *)(hist + (lane + 64 * j) * 4);
        }
        // ---------------- the views: apply, row out, restore
        int eo = 0;
#pragma unroll 1
        for (int vi = 0; vi < P; ++vi) {
            int ne = 0;                                       // (a select chain: a run-time index would put the array into scratch)
#pragma unroll
            for (int v = 0; v < V3_MAXV; ++v) ne = v == vi ? cur.ne[v] : ne;
            const uint32_t *Ev = E + eo;
            int dwt = 0;
            for (int e0 = 0; e0 < ne; e0 += 64) {
                const int ei = e0 + lane;
                if (ei < ne) {
                    uint32_t pr[K];
                    dwt += W::edit_eval(Ev, ne, ei, cur.L - 1, cod, msk, garbage, pr);
#pragma unroll
                    for (int t = 0; t < K; ++t) W::move(hist, pr[t], 1u);
                }
            }
            uint4 h[KEEP ? RP : 1];
            int64_t S = (int64_t)windows + (iv ? (int64_t)F : 0);
            if (ne > 0) { fence(); S += (int64_t)wave_sum(dwt); }
            const int64_t row = (int64_t)vi * a.view_stride + cur.s * row_len;
            bool stored = false;
            if constexpr (OM) {
                if (a.mode == IDL_MODE_CANONICAL) {
                    // utils.py:208-221: for every k-mer b <= rc(b), ascending: int32((c[b] + c[rc]) * 0.5) -- truncating, palindromes keep their count --, the row
                    // normalised by ITS OWN sum (utils.py:246-250).  A lane takes bins lane, lane + 64, ...; ranks from a running ballot, as in v2.
                    // (two walks over the bins -- the sum, then the rows -- instead of a lane's 4^k / 64 collapsed bins in registers: 16 at k = 5, where they spilled)
                    int part = 0;
#pragma unroll 4
                    for (int t = 0; t < F / 64; ++t) {
                        const uint32_t b = 64u * t + lane, rc = revcomp<K>(b);
                        if (b <= rc) part += (int32_t)(hist[b] + hist[rc]) / 2;
                    }
                    const int Sc = wave_sum(part);
                    const float Sf = (float)Sc, rS = 1.0f / Sf;
                    int rank0 = 0;
#pragma unroll 4
                    for (int t = 0; t < F / 64; ++t) {
                        const uint32_t b = 64u * t + lane, rc = revcomp<K>(b);
                        const bool cn = b <= rc;
                        const uint64_t bal = __ballot(cn);
                        if (cn) {
                            const uint32_t val = (uint32_t)((int32_t)(hist[b] + hist[rc]) / 2);
                            const int64_t o = row + rank0 + __popcll(bal & ((1ull << lane) - 1ull));
                            if (a.out_kind == IDL_OUT_COUNTS_I32) ((uint32_t *)a.out)[o] = val;
                            else { const float c = (float)val, q = c * rS; ((float *)a.out)[o] = fmaf(fmaf(-q, Sf, c), rS, q); }     // (Sc < 2^24: == float32(float64 division))
                        }
                        rank0 += __popcll(bal);
                    }
                    stored = true;
                } else if (a.mode == IDL_MODE_CGR) {
                    // the histogram is kept in k-mer order; a CGR row is stored in pixel order (kmers.pyx:53-123 through cgr_pixel_to_kmer)
                    if constexpr (KEEP) {
#pragma unroll
                        for (int j = 0; j < RP; ++j) {
                            const uint32_t i0 = (uint32_t)(lane + 64 * j) * 4u;
                            h[j] = make_uint4(hist[cgr_pixel_to_kmer<K>(i0)], hist[cgr_pixel_to_kmer<K>(i0 + 1)], hist[cgr_pixel_to_kmer<K>(i0 + 2)], hist[cgr_pixel_to_kmer<K>(i0 + 3)]);
                        }
                    } else {
                        const float Sf = (float)S, rS = 1.0f / Sf;
                        float *dst = (float *)a.out + row + lane * 4;
#pragma unroll 4
                        for (int j = 0; j < RP; ++j) {
                            const uint32_t i0 = (uint32_t)(lane + 64 * j) * 4u;
                            const uint4 hh = make_uint4(hist[cgr_pixel_to_kmer<K>(i0)], hist[cgr_pixel_to_kmer<K>(i0 + 1)], hist[cgr_pixel_to_kmer<K>(i0 + 2)], hist[cgr_pixel_to_kmer<K>(i0 + 3)]);
                            if (a.out_kind == IDL_OUT_COUNTS_I32) *(uint4 *)((uint32_t *)dst + j * 256) = hh;
                            else {
                                const float c0 = (float)hh.x, c1 = (float)hh.y, c2 = (float)hh.z, c3 = (float)hh.w;
                                const float q0 = c0 * rS, q1 = c1 * rS, q2 = c2 * rS, q3 = c3 * rS;
                                *(float4 *)(dst + j * 256) = make_float4(fmaf(fmaf(-q0, Sf, c0), rS, q0), fmaf(fmaf(-q1, Sf, c1), rS, q1),
                                                                         fmaf(fmaf(-q2, Sf, c2), rS, q2), fmaf(fmaf(-q3, Sf, c3), rS, q3));
                            }
                        }
                        stored = true;
                    }
                }
            }
            if constexpr (KEEP) {
                if (!OM || a.mode == IDL_MODE_KMER) {
                    if (ne > 0) {
#pragma unroll
                        for (int j = 0; j < RP; ++j) h[j] = *(const uint4 *)(hist + (lane + 64 * j) * 4);
                    } else {
#pragma unroll
                        for (int j = 0; j < RP; ++j) h[j] = h0[j];
                    }
                }
                if (ne > 0 && vi + 1 < P) {                  // (the LDS unit takes a wave's operations in order: the stores follow the loads)
#pragma unroll
                    for (int j = 0; j < RP; ++j) *(uint4 *)(hist + (lane + 64 * j) * 4) = h0[j];
                    fence();
                }
            } else if (ne > 0 && vi + 1 < P) {               // k = 6: back to the un-mutated histogram by the same pairs with the other sign
                fence();
                for (int e0 = 0; e0 < ne; e0 += 64) {
                    const int ei = e0 + lane;
                    if (ei < ne) {
                        uint32_t pr[K];
                        (void)W::edit_eval(Ev, ne, ei, cur.L - 1, cod, msk, garbage, pr);
#pragma unroll
                        for (int t = 0; t < K; ++t) W::move(hist, pr[t], 0xFFFFFFFFu);
                    }
                }
                fence();
            }
            if (KEEP && !stored) {
                float *dst = (float *)a.out + row + lane * 4;
                if (a.out_kind == IDL_OUT_COUNTS_I32) {
#pragma unroll
                    for (int j = 0; j < RP; ++j) *(uint4 *)((uint32_t *)dst + j * 256) = h[j];
                } else {
                    // S <= max_len + 4^k < 2^24 (the launcher bounds max_len): the Markstein form equals float32(float64 division), as in v3
                    const float Sf = (float)S, rS = 1.0f / Sf;
#pragma unroll
                    for (int j = 0; j < RP; ++j) {
                        const float c0 = (float)h[j].x, c1 = (float)h[j].y, c2 = (float)h[j].z, c3 = (float)h[j].w;
                        const float q0 = c0 * rS, q1 = c1 * rS, q2 = c2 * rS, q3 = c3 * rS;
                        *(float4 *)(dst + j * 256) = make_float4(fmaf(fmaf(-q0, Sf, c0), rS, q0), fmaf(fmaf(-q1, Sf, c1), rS, q1),
                                                                 fmaf(fmaf(-q2, Sf, c2), rS, q2), fmaf(fmaf(-q3, Sf, c3), rS, q3));
                    }
                }
            }
            eo += ne;
        }
        cur = nxt;
    }
}

// collapse a host-provided histogram (idl_kmer_rev_comp): one wave, counts modified in place like utils.py:216-217
template <int K>
__global__ __launch_bounds__(64) void collapse_kernel(int32_t *counts, int32_t *out)
{
    constexpr int F = 1 << (2 * K);
    const int lane = threadIdx.x;
    int rank0 = 0;
    for (int b0 = 0; b0 < F; b0 += 64) {
        const uint32_t b = b0 + lane;
        bool canon = false;
        int32_t val = 0;
        if (b < (uint32_t)F) {
            const uint32_t rc = revcomp<K>(b);
            canon = b <= rc;
            if (canon) val = (int32_t)((uint32_t)counts[b] + (uint32_t)counts[rc]) / 2;
        }
        const uint64_t bal = __ballot(canon);
        if (canon) {
            out[rank0 + __popcll(bal & ((1ull << lane) - 1ull))] = val;
            counts[b] = val;  // only canonical entries are written; reads of rc > b never race with them
        }
        rank0 += __popcll(bal);
    }
}

// The invalid-mask of records WITHOUT an invalid base is the tail padding their length implies: base b of a 64-base slot sits at bit
// 31 - (b mod 32) of word b / 32 (host_ingest.cpp: Packer), padding = every base from the record's length on.  One wave per record.
__global__ __launch_bounds__(256) void mask_from_lengths_kernel(uint2 *mask, const int64_t *slot_off, const int64_t *lengths, const uint8_t *sent, int64_t n)
{
    const int lane = threadIdx.x & 63;
    for (int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); s < n; s += (int64_t)gridDim.x * 4) {
        if (sent != nullptr && sent[s] != 0) continue;
        const int64_t len = lengths[s], nslots = (len + 63) >> 6, s0 = slot_off[s];
        for (int64_t i = lane; i < nslots; i += 64) {
            const int64_t v = len - i * 64;                      // valid bases of this slot: >= 64 but in the last one
            uint2 m = make_uint2(0u, 0u);
            if (v < 64) {
                m.x = v >= 32 ? 0u : (v <= 0 ? 0xFFFFFFFFu : 0xFFFFFFFFu >> (int)v);
                m.y = v <= 32 ? 0xFFFFFFFFu : 0xFFFFFFFFu >> (int)(v - 32);
            }
            mask[s0 + i] = m;
        }
    }
}

// the device words through which v3 hands over to the second pass (redo count, queue head, redo list): one block per (device,
// stream), allocated on first use and kept -- launches on different streams of a device may be in flight together (a build on one
// stream, predict_features on another), launches on one stream are ordered
int *redo_counter(hipStream_t st)
{
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, int *> blocks;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    int *&p = blocks[std::make_pair(dev, st)];
    if (p == nullptr && hipMalloc((void **)&p, 8 + 8 * (size_t)V3_REDO_CAP) != hipSuccess) p = nullptr;
    return p;
}

template <int K, int RL>
int launch_vectorise3_k(VecArgs &a, const idl::DeviceInfo &di, hipStream_t st, size_t lds, int per_cu)
{
    const bool dbg = idl::dev_env("vec_dbg") != nullptr;
    const void *fn = dbg ? (const void *)vectorise3_kernel<K, true, RL> : (const void *)vectorise3_kernel<K, false, RL>;
    if (lds > 64 * 1024) IDL_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t grid = (int64_t)di.cus * per_cu;
    if (grid > a.n) grid = a.n;
    a.redo_count = redo_counter(st);
    if (a.redo_count == nullptr) { idl::set_error("cannot allocate the v3 hand-over word"); return IDL_ERR_HIP; }
    IDL_HIP_TRY(hipMemsetAsync(a.redo_count, 0, 2 * sizeof(int), st));      // [0] sequences left to the second pass, [1] head of the sequence queue
    if (dbg) { IDL_HIP_TRY(hipMalloc((void **)&a.dbg, (size_t)grid * 20 * 8)); IDL_HIP_TRY(hipMemset(a.dbg, 0, (size_t)grid * 20 * 8)); }
    if (dbg) hipLaunchKernelGGL((vectorise3_kernel<K, true, RL>), dim3((unsigned)grid), dim3(V3_NT), lds, st, a);
    else hipLaunchKernelGGL((vectorise3_kernel<K, false, RL>), dim3((unsigned)grid), dim3(V3_NT), lds, st, a);
    IDL_HIP_TRY(hipGetLastError());
    if (dbg) {          // diagnostic only: synchronises, prints the mean cycles per sequence of each phase for compute wave 0 and the memory wave
        std::vector<unsigned long long> h((size_t)grid * 20);
        IDL_HIP_TRY(hipMemcpy(h.data(), a.dbg, h.size() * 8, hipMemcpyDeviceToHost));
        (void)hipFree(a.dbg);
        a.dbg = nullptr;
        static const char *nm[10] = {"count", "edits", "barrier(P1)", "row load", "barrier(row)", "dma issue", "lists|stores", "barrier(lists)", "-", "-"};
        for (int w = 0; w < 2; ++w) {
            fprintf(stderr, "[idl] v3 cycles per sequence, %s wave:", w ? "memory" : "compute");
            double tot = 0;
            for (int k = 0; k < 8; ++k) { double sum = 0; for (int64_t b = 0; b < grid; ++b) sum += (double)h[(size_t)(b * 2 + w) * 10 + k]; fprintf(stderr, " %s %.0f", nm[k], sum / (double)a.n); tot += sum / (double)a.n; }
            fprintf(stderr, " | total %.0f\n", tot);
        }
    }
    // second pass: the v2 kernel on the sequences v3 left alone (their edits / pairs exceed the LDS tables); exits at once when there are none
    {
        VecArgs b = a;
        b.redo = 1;
        b.v3_sc = a.sc_slots;
        b.sc_slots = 160;
        const size_t lds2 = (size_t)((1 << (2 * K)) + 4 + (b.sc_slots + 1) * 6 + V2_EDIT_CAP + V2_LIST_CAP + 2 * V2_WAVES + 4 * b.n_views) * 4;
        int64_t g2 = (int64_t)di.cus * 4;
        if (g2 > a.n) g2 = a.n;
        hipLaunchKernelGGL((vectorise2_kernel<K, false>), dim3((unsigned)g2), dim3(64 * V2_WAVES), lds2, st, b);
        IDL_HIP_TRY(hipGetLastError());
    }
    return IDL_OK;
}

// v3 takes the hot shape only: plain k-mer rows (float32 or int32), k = 4..6, <= 8 views, every sequence within one staged
// super-chunk (the host's length bound says so), fresh histograms.  Everything else stays on v2 / v1.  k = 4 counts into 2^RL
// copies of its 1 KB histogram (V3<K, RL>): RL = 5 is conflict-free and fits three workgroups per CU at 10 kbp, RL = 4 four.
template <int K, int RL>
int launch_vectorise3_rl(const VecArgs &a_in, const idl::DeviceInfo &di, hipStream_t st, bool *done)
{
    {
        constexpr int HW = V3<K, RL>::HD;                          // histogram words (k = 4: the copies + the histogram proper)
        VecArgs a = a_in;
        int want = 3;
        if (const char *e = idl::dev_env("vec")) want = atoi(e);
        if (want != 3 || a.mode != IDL_MODE_KMER || a.init == IDL_INIT_FROM_OUT || a.out_kind == IDL_OUT_FREQ_F64 ||
            a.n_views > V3_MAXV || a.max_len <= 0 || a.max_len > 64 * 2048 || a.n > 0x7F000000ll)
            return IDL_OK;
        a.sc_slots = (int)((a.max_len + 63) / 64);
        // LDS tables for the edits of all views and their K windows each: at least 3.5 % of the bases + slack, and whatever else
        // fits at four workgroups per CU (an edit costs 2 + K words: two staging sets and the list); IDELUCS_DEV=v3_ec / _LC override
        int ec = (int)(a.max_len * 35 / 1000) + 64, lc;
        {
            const int fixed = HW + 2 * (a.sc_slots + 1) * 6 + 3 * V3_META + 2 * V3_VTAB + 16 + 4;
            int wgs = 4;                                  // workgroups per CU the tables are sized for: four, or as many as the fixed part + the minimum leaves
            while (wgs > 2 && ((di.lds_per_cu / wgs - 1024) / 4 - fixed) / (2 + K) < ec) --wgs;
            const int fit = ((di.lds_per_cu / wgs - 1024) / 4 - fixed) / (2 + K);
            if (fit > ec) ec = fit > 4096 ? 4096 : fit;
        }
        if (const char *e = idl::dev_env("v3_ec")) { const int t = atoi(e); if (t >= 0 && t <= 16384) ec = t; }
        ec &= ~7;
        lc = ec * K;
        if (const char *e = idl::dev_env("v3_lc")) { const int t = atoi(e); if (t >= 0 && t <= 65536) lc = t; }
        if (a.edits == nullptr) { ec = 0; lc = 0; }
        a.ecap = ec; a.lcap = lc;
        a.chunk = 2;
        if (const char *e = idl::dev_env("v3_chunk")) { const int t = atoi(e); if (t >= 1 && t <= 4096) a.chunk = t; }
        if (const char *e = idl::dev_env("vec_ablate")) a.ablate = atoi(e);
        const size_t lds = (size_t)(HW + 2 * ((a.sc_slots + 1) * 6 + ec) + lc + 3 * V3_META + 2 * V3_VTAB + 16 + 4) * 4;
        if ((int)lds > di.max_dyn_lds || lds > 80 * 1024) return IDL_OK;                 // (fewer than two workgroups per CU: v2's chunked staging is the better fit)
        // workgroups per CU by LDS, with 1 KB of slack each (measured: five 32 032-byte workgroups do NOT become resident
        // together although 5 x 32 032 < 160 KB and the occupancy query says 5 -- the fifth ran after the others)
        int per_cu = (int)((size_t)di.lds_per_cu / (lds + 1024));
        if (per_cu > 4) per_cu = 4;                   // 8 waves per workgroup, <= 64 registers: 32 waves per CU
        if (const char *e = idl::dev_env("wg_per_cu")) { const int t = atoi(e); if (t >= 1 && t <= 32) per_cu = t; }
        if (per_cu < 1) per_cu = 1;
        if (getenv("IDELUCS_DEBUG"))
            fprintf(stderr, "[idl] vectorise k=%d v3 lds=%zu B (histogram copies %d, staged slots %d, edits %d, pairs %d) -> %d workgroups/CU\n", K, lds, 1 << RL, a.sc_slots, ec, lc, per_cu);
        const int rc = launch_vectorise3_k<K, RL>(a, di, st, lds, per_cu);
        if (rc != IDL_OK) return rc;
        *done = true;
    }
    return IDL_OK;
}

// v4 takes what v3 takes at k = 4 / 5 when every sequence fits its wave's slice of LDS (<= 12 288 bases); IDELUCS_DEV=vec=3 keeps v3 (the cross-check tests)
template <int K, int RL>
int launch_vectorise4(const VecArgs &a_in, const idl::DeviceInfo &di, hipStream_t st, bool *done)
{
    VecArgs a = a_in;
    int want = 4;
    if (const char *e = idl::dev_env("vec")) want = atoi(e);
    const bool om = a.mode == IDL_MODE_CGR || a.mode == IDL_MODE_CANONICAL;
    if (K == 6 && !om) return IDL_OK;                        // (plain rows at k = 6 are v3's: bound by the write stream there)
    // (k = 6, CGR / canonical: 2.29 / 4.99 ms at 100 000 x 10 kbp x 4 views against v2's 3.05 / 7.53.  The collapse is latency-bound -- 4096 bins x 4 views a sequence walked by
    //  one wave in 64 turns, twice --, not bank- or instruction-bound: summing the pairs in place in a skewed, conflict-free order (lane l of
    //  turn t takes b = l | ((l + t) mod 64) << 6) and restoring them by a third walk was built and measured: 6.0 ms at k = 6, no change at k = 4 / 5; rc(b) from per-lane and per-turn
    //  halves (no bit reversal in the loop) measured too: 5.1 ms -- a walk costs ~1.3 ms whatever its body: 64 dependent turns of LDS latency at two waves a SIMD)
    if (want != 4 || (a.mode != IDL_MODE_KMER && !om) || a.init == IDL_INIT_FROM_OUT || a.out_kind == IDL_OUT_FREQ_F64 || a.n_views > V3_MAXV || a.max_len <= 0 ||
        a.max_len > 64 * 64 * V4_SR || a.n > 0x7F000000ll)
        return IDL_OK;
    using W = V3<K, RL>;
    a.v3_sc = (int)((a.max_len + 63) / 64);
    a.sc_slots = a.v3_sc;
    int ec = a.edits == nullptr ? 0 : (((int)(a.max_len * 35 / 1000) + 64) * a.n_views / 4 + 7) & ~7;      // edits of all views: 3.5 % of the bases + slack for four views, scaled
    if (ec > 64 * V4_FR) ec = 64 * V4_FR;
    if (const char *e = idl::dev_env("v3_ec")) { const int t = atoi(e); if (t >= 0 && t <= 64 * V4_FR) ec = t & ~7; }
    a.ecap = ec; a.lcap = 0;
    const int slice = (W::HC + (W::F + 4) + (a.v3_sc + 1) * 6 + ec + 3) & ~3;
    // as many waves a workgroup as its CU's LDS holds slices (one workgroup per CU: nothing is shared between the waves but the CU)
    int waves = (di.lds_per_cu - 2048) / (slice * 4);
    if (waves > V4_MAX_WAVES) waves = V4_MAX_WAVES;
    if (const char *e = idl::dev_env("v4_waves")) { const int t = atoi(e); if (t >= 1 && t <= waves) waves = t; }
    if (waves < 4) return IDL_OK;                            // (long sequences: v3 / v2)
    const size_t lds = (size_t)slice * 4 * waves;
    if ((int)lds > di.max_dyn_lds) return IDL_OK;
    const void *fn = (om || K == 6) ? (const void *)vectorise4_kernel<K, RL, true> : (const void *)vectorise4_kernel<K, RL, K == 6>;
    if (lds > 64 * 1024) IDL_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int per_cu = 1, lc = 0;
    int64_t grid = (int64_t)di.cus * per_cu;
    if (grid * waves > a.n) grid = (a.n + waves - 1) / waves;
    a.redo_count = redo_counter(st);
    if (a.redo_count == nullptr) { idl::set_error("cannot allocate the v3 hand-over word"); return IDL_ERR_HIP; }
    IDL_HIP_TRY(hipMemsetAsync(a.redo_count, 0, 2 * sizeof(int), st));
    if (getenv("IDELUCS_DEBUG"))
        fprintf(stderr, "[idl] vectorise k=%d v4 (a wave per sequence) lds=%zu B a workgroup of %d waves (copies %d, staged slots %d, edits %d, pairs %d) -> %d workgroups/CU\n",
                K, lds, waves, 1 << RL, a.v3_sc, ec, lc, per_cu);
    if (om || K == 6) hipLaunchKernelGGL((vectorise4_kernel<K, RL, true>), dim3((unsigned)grid), dim3(64 * waves), lds, st, a);
    else hipLaunchKernelGGL((vectorise4_kernel<K, RL, K == 6>), dim3((unsigned)grid), dim3(64 * waves), lds, st, a);
    IDL_HIP_TRY(hipGetLastError());
    {       // second pass: v2 on the sequences v4 left alone; exits at once when there are none
        VecArgs b = a;
        b.redo = 2;
        b.sc_slots = 160;
        const size_t lds2 = (size_t)((1 << (2 * K)) + 4 + (b.sc_slots + 1) * 6 + V2_EDIT_CAP + V2_LIST_CAP + 2 * V2_WAVES + 4 * b.n_views) * 4;
        int64_t g2 = (int64_t)di.cus * 4;
        if (g2 > a.n) g2 = a.n;
        hipLaunchKernelGGL((vectorise2_kernel<K, false>), dim3((unsigned)g2), dim3(64 * V2_WAVES), lds2, st, b);
        IDL_HIP_TRY(hipGetLastError());
    }
    *done = true;
    return IDL_OK;
}

template <int K>
int launch_vectorise3(const VecArgs &a_in, const idl::DeviceInfo &di, hipStream_t st, bool *done)
{
    *done = false;
    if constexpr (K == 4 || K == 5 || K == 6) {
        int rc;
        if constexpr (K == 4) {
            // copies of the histogram a wave counts into: 4 -- measured at 100 000 x 10 kbp, 4 views, one box (gpurun_out r06): no edits / philox edits
            // 0.351 / 0.649 ms with 4, 0.349 / 0.738 with 8, 0.453 / 1.025 with 16 (fewer waves fit); v3 on the same box: 0.621 / 0.725
            int rl = 2;
            if (const char *e = idl::dev_env("v4_copies")) { const int t = atoi(e); rl = t == 16 ? 4 : (t == 8 ? 3 : (t == 1 ? 0 : 2)); }
            rc = rl == 4 ? launch_vectorise4<K, 4>(a_in, di, st, done) : (rl == 2 ? launch_vectorise4<K, 2>(a_in, di, st, done) :
                 (rl == 0 ? launch_vectorise4<K, 0>(a_in, di, st, done) : launch_vectorise4<K, 3>(a_in, di, st, done)));
        } else rc = launch_vectorise4<K, 0>(a_in, di, st, done);
        if (rc != IDL_OK || *done) return rc;
    }
    if constexpr (K == 4) {
        // copies of the histogram the count goes to: 16 by default -- measured at 100 000 x 10 kbp, 4 views (gpurun_out/r05_e): 0.708 ms
        // with 16 (four workgroups per CU), 0.714 with 8, 0.844 with 32 (conflict-free, but 41 KB of LDS: three per CU); v2: 1.393
        int rl = 4;
        if (const char *e = idl::dev_env("v3_copies")) { const int t = atoi(e); rl = t == 32 ? 5 : (t == 8 ? 3 : 4); }
        if (rl == 5) return launch_vectorise3_rl<K, 5>(a_in, di, st, done);
        if (rl == 4) return launch_vectorise3_rl<K, 4>(a_in, di, st, done);
        return launch_vectorise3_rl<K, 3>(a_in, di, st, done);
    } else if constexpr (K >= 5 && K <= 6) {
        // (k = 5 with the count in 4 / 8 copies was measured too, gpurun_out/r05_s: 0.904 / 1.139 ms against 0.804 without -- folding
        //  1024 bins x copies costs more than the conflicts it saves; bit-identical rows)
        return launch_vectorise3_rl<K, 0>(a_in, di, st, done);
    }
    return IDL_OK;
}

template <int K>
int launch_vectorise(const VecArgs &a_in, const idl::DeviceInfo &di, hipStream_t st)
{
    constexpr int F = 1 << (2 * K);
    {
        bool done = false;
        const int rc = launch_vectorise3<K>(a_in, di, st, &done);
        if (rc != IDL_OK || done) return rc;
    }
    VecArgs a = a_in;
    bool v1 = (a.init == IDL_INIT_FROM_OUT);            // accumulate-on-top needs a per-view start: single-pass kernel
    if (const char *e = idl::dev_env("vec")) { if (atoi(e) == 1) v1 = true; }   // force the v1 kernel (cross-check tests)
    int sc = 160;                                       // 10240 bases staged at a time (cfg2's 10 kbp in one super-chunk)
    if (const char *e = idl::dev_env("sc_slots")) { const int t = atoi(e); if (t >= 1 && t <= 4096) sc = t; }
    a.sc_slots = sc;
    if (const char *e = idl::dev_env("vec_ablate")) a.ablate = atoi(e);
    // 16-bit bins are possible while no count can reach 65536 (count <= max_len + 1).  Measured on MI355X (cfg2): 8
    // workgroups/CU with 16-bit bins (64 VGPRs, 80 B/lane of spill) run 2.16 ms, 6 workgroups/CU with 32-bit bins
    // (79 VGPRs) 1.98 ms -- so 32-bit is the default and 16-bit an opt-in experiment (IDELUCS_DEV=bins=16).
    bool b16 = false;
    if (const char *e = idl::dev_env("bins")) b16 = atoi(e) == 16 && !v1 && a.max_len > 0 && a.max_len <= 65000;
    const int hd = b16 ? (F / 2 + 4) & ~3 : F + 4;
    const size_t lds = v1 ? (size_t)(F + STAGE_DWORDS) * 4
                          : (size_t)(hd + (sc + 1) * 6 + V2_EDIT_CAP + V2_LIST_CAP + 2 * V2_WAVES + 4 * a.n_views) * 4;
    if ((int)lds > di.max_dyn_lds) {
        idl::set_error("k=%d needs %zu bytes of LDS per wavefront; device allows %d", K, lds, di.max_dyn_lds);
        return IDL_ERR_ARG;
    }
    const void *fn = v1 ? (const void *)vectorise_kernel<K>
                        : (b16 ? (const void *)vectorise2_kernel<K, true> : (const void *)vectorise2_kernel<K, false>);
    if (lds > 64 * 1024) IDL_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    IDL_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, v1 ? 64 : 64 * V2_WAVES, lds));
    if (const char *e = idl::dev_env("wg_per_cu")) { const int t = atoi(e); if (t >= 1 && t <= 32) per_cu = t; }
    if (per_cu > 16) per_cu = 16;
    if (per_cu < 1) per_cu = 1;
    if (getenv("IDELUCS_DEBUG"))
        fprintf(stderr, "[idl] vectorise k=%d %s lds=%zu B -> %d workgroups/CU\n", K, v1 ? "v1" : (b16 ? "v2/16-bit bins" : "v2/32-bit bins"), lds, per_cu);
    int64_t grid = (int64_t)di.cus * per_cu;
    if (grid > a.n) grid = a.n;
    if (v1) hipLaunchKernelGGL(vectorise_kernel<K>, dim3((unsigned)grid), dim3(64), lds, st, a);
    else if (b16) hipLaunchKernelGGL((vectorise2_kernel<K, true>), dim3((unsigned)grid), dim3(64 * V2_WAVES), lds, st, a);
    else hipLaunchKernelGGL((vectorise2_kernel<K, false>), dim3((unsigned)grid), dim3(64 * V2_WAVES), lds, st, a);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

int dispatch_vectorise(int k, const VecArgs &a, hipStream_t st)
{
    idl::DeviceInfo di;
    int rc = idl::device_info(&di);
    if (rc != IDL_OK) return rc;
    switch (k) {
    case 1: return launch_vectorise<1>(a, di, st);
    case 2: return launch_vectorise<2>(a, di, st);
    case 3: return launch_vectorise<3>(a, di, st);
    case 4: return launch_vectorise<4>(a, di, st);
    case 5: return launch_vectorise<5>(a, di, st);
    case 6: return launch_vectorise<6>(a, di, st);
    case 7: return launch_vectorise<7>(a, di, st);
    }
    idl::set_error("k=%d outside 1..%d", k, IDL_MAX_K);
    return IDL_ERR_ARG;
}

struct DevBuf {  // RAII for the scalar host-pointer entry points
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};

int scalar_count(const uint8_t *seq, int64_t len, int k, int32_t *counts, int mode)
{
    IDL_REQUIRE(k >= 1 && k <= IDL_MAX_K, "k outside 1..IDL_MAX_K");
    IDL_REQUIRE(len >= 0 && (len == 0 || seq != nullptr) && counts != nullptr, "NULL buffer");
    IDL_REQUIRE(len < (1ll << 31), "sequence longer than 2^31 (kmers.pyx:16 int contiglength)");
    idl::DeviceInfo di;
    int rc = idl::device_info(&di);
    if (rc != IDL_OK) return rc;
    const int64_t F = 1ll << (2 * k);
    const int64_t slots = (len + 63) / 64;
    const int64_t byte_off[2] = {0, len};
    int64_t slot_off[2];
    uint8_t *h_codes = (uint8_t *)malloc((size_t)(slots > 0 ? slots : 1) * 24);
    if (!h_codes) { idl::set_error("out of host memory"); return IDL_ERR_NOMEM; }
    uint8_t *h_mask = h_codes + (size_t)(slots > 0 ? slots : 1) * 16;
    rc = idl_pack(seq, byte_off, 1, h_codes, h_mask, slot_off);
    if (rc != IDL_OK) { free(h_codes); return rc; }

    DevBuf d_codes, d_mask, d_meta, d_out;
    struct FreeHost { uint8_t *p; ~FreeHost() { free(p); } } fh{h_codes};
    IDL_HIP_TRY(hipMalloc(&d_codes.p, (size_t)(slots > 0 ? slots : 1) * 16));
    IDL_HIP_TRY(hipMalloc(&d_mask.p, (size_t)(slots > 0 ? slots : 1) * 8));
    IDL_HIP_TRY(hipMalloc(&d_meta.p, 3 * sizeof(int64_t)));
    IDL_HIP_TRY(hipMalloc(&d_out.p, (size_t)F * 4));
    if (slots > 0) {
        IDL_HIP_TRY(hipMemcpy(d_codes.p, h_codes, (size_t)slots * 16, hipMemcpyHostToDevice));
        IDL_HIP_TRY(hipMemcpy(d_mask.p, h_mask, (size_t)slots * 8, hipMemcpyHostToDevice));
    }
    const int64_t meta[3] = {slot_off[0], slot_off[1], len};
    IDL_HIP_TRY(hipMemcpy(d_meta.p, meta, sizeof(meta), hipMemcpyHostToDevice));
    IDL_HIP_TRY(hipMemcpy(d_out.p, counts, (size_t)F * 4, hipMemcpyHostToDevice));

    VecArgs a{};
    a.codes = (const uint4 *)d_codes.p;
    a.mask = (const uint2 *)d_mask.p;
    a.slot_off = (const int64_t *)d_meta.p;
    a.lengths = (const int64_t *)d_meta.p + 2;
    a.n = 1;
    a.mode = mode;
    a.init = IDL_INIT_FROM_OUT;
    a.out_kind = IDL_OUT_COUNTS_I32;
    a.n_views = 1;
    a.out = d_out.p;
    a.view_stride = F;
    rc = dispatch_vectorise(k, a, nullptr);
    if (rc != IDL_OK) return rc;
    IDL_HIP_TRY(hipMemcpy(counts, d_out.p, (size_t)F * 4, hipMemcpyDeviceToHost));
    return IDL_OK;
}

}  // namespace

extern "C" {

int64_t idl_row_len(int mode, int k)
{
    if (k < 1 || k > 15) return -1;
    const int64_t F = 1ll << (2 * k);
    if (mode == IDL_MODE_CANONICAL) return (k % 2 == 0) ? (F + (1ll << k)) / 2 : F / 2;  // models.py:61-62
    return F;
}

int idl_kmer_counts(const uint8_t *seq, int64_t len, int k, int32_t *counts)
{
    return scalar_count(seq, len, k, counts, IDL_MODE_KMER);
}

int idl_cgr(const uint8_t *seq, int64_t len, int k, int32_t *counts)
{
    return scalar_count(seq, len, k, counts, IDL_MODE_CGR);
}

int idl_kmer_rev_comp(int32_t *counts, int k, int32_t *out)
{
    IDL_REQUIRE(k >= 1 && k <= IDL_MAX_K, "k outside 1..IDL_MAX_K");
    IDL_REQUIRE(counts != nullptr && out != nullptr, "NULL buffer");
    idl::DeviceInfo di;
    int rc = idl::device_info(&di);
    if (rc != IDL_OK) return rc;
    const int64_t F = 1ll << (2 * k), C = idl_row_len(IDL_MODE_CANONICAL, k);
    DevBuf d_counts, d_out;
    IDL_HIP_TRY(hipMalloc(&d_counts.p, (size_t)F * 4));
    IDL_HIP_TRY(hipMalloc(&d_out.p, (size_t)C * 4));
    IDL_HIP_TRY(hipMemcpy(d_counts.p, counts, (size_t)F * 4, hipMemcpyHostToDevice));
    int32_t *dc = (int32_t *)d_counts.p, *dout = (int32_t *)d_out.p;
    switch (k) {
    case 1: hipLaunchKernelGGL(collapse_kernel<1>, dim3(1), dim3(64), 0, nullptr, dc, dout); break;
    case 2: hipLaunchKernelGGL(collapse_kernel<2>, dim3(1), dim3(64), 0, nullptr, dc, dout); break;
    case 3: hipLaunchKernelGGL(collapse_kernel<3>, dim3(1), dim3(64), 0, nullptr, dc, dout); break;
    case 4: hipLaunchKernelGGL(collapse_kernel<4>, dim3(1), dim3(64), 0, nullptr, dc, dout); break;
    case 5: hipLaunchKernelGGL(collapse_kernel<5>, dim3(1), dim3(64), 0, nullptr, dc, dout); break;
    case 6: hipLaunchKernelGGL(collapse_kernel<6>, dim3(1), dim3(64), 0, nullptr, dc, dout); break;
    case 7: hipLaunchKernelGGL(collapse_kernel<7>, dim3(1), dim3(64), 0, nullptr, dc, dout); break;
    }
    IDL_HIP_TRY(hipGetLastError());
    IDL_HIP_TRY(hipMemcpy(counts, d_counts.p, (size_t)F * 4, hipMemcpyDeviceToHost));
    IDL_HIP_TRY(hipMemcpy(out, d_out.p, (size_t)C * 4, hipMemcpyDeviceToHost));
    return IDL_OK;
}

int idl_mask_from_lengths(void *mask, const int64_t *slot_off, const int64_t *lengths, const uint8_t *sent, int64_t n, void *stream)
{
    IDL_REQUIRE(n >= 0 && (n == 0 || (mask && slot_off && lengths)), "mask_from_lengths: NULL buffer");
    if (n == 0) return IDL_OK;
    idl::DeviceInfo di;
    const int rc = idl::device_info(&di);
    if (rc != IDL_OK) return rc;
    int64_t grid = (n + 3) / 4;
    if (grid > (int64_t)di.cus * 32) grid = (int64_t)di.cus * 32;
    hipLaunchKernelGGL(mask_from_lengths_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (uint2 *)mask, slot_off, lengths, sent, n);
    IDL_HIP_TRY(hipGetLastError());
    return IDL_OK;
}

static int vectorise_impl(const void *codes, const void *mask, const int64_t *slot_off, const int64_t *lengths,
                  int64_t n, int k, int mode, int init, int out_kind,
                  int n_views, const uint32_t *edits, const int64_t *edit_off, int eo_shift,
                  void *out, int64_t view_stride, int64_t max_len, void *stream)
{
    IDL_REQUIRE(k >= 1 && k <= IDL_MAX_K, "k outside 1..IDL_MAX_K");
    IDL_REQUIRE(n >= 0 && n_views >= 1 && n_views <= 64, "n < 0 or n_views outside 1..64");
    IDL_REQUIRE(mode == IDL_MODE_KMER || mode == IDL_MODE_CGR || mode == IDL_MODE_CANONICAL, "unknown mode");
    IDL_REQUIRE(init == IDL_INIT_ZERO || init == IDL_INIT_ONE || init == IDL_INIT_FROM_OUT, "unknown init");
    IDL_REQUIRE(out_kind == IDL_OUT_COUNTS_I32 || out_kind == IDL_OUT_FREQ_F32 || out_kind == IDL_OUT_FREQ_F64,
                "unknown out_kind");
    IDL_REQUIRE(!(init == IDL_INIT_FROM_OUT && (out_kind != IDL_OUT_COUNTS_I32 || mode == IDL_MODE_CANONICAL)),
                "IDL_INIT_FROM_OUT needs IDL_OUT_COUNTS_I32 and a non-canonical mode");
    IDL_REQUIRE((edits == nullptr) == (edit_off == nullptr), "edits and edit_off must both be given or both NULL");
    IDL_REQUIRE(view_stride >= n * idl_row_len(mode, k) || n_views == 1, "view_stride smaller than one view");
    if (n == 0) return IDL_OK;
    IDL_REQUIRE(codes && mask && slot_off && lengths && out, "NULL buffer");
    IDL_REQUIRE(((uintptr_t)codes & 15u) == 0 && ((uintptr_t)mask & 7u) == 0 && ((uintptr_t)out & 15u) == 0,
                "codes/out must be 16-byte aligned, mask 8-byte aligned");
    VecArgs a{};
    a.codes = (const uint4 *)codes;
    a.mask = (const uint2 *)mask;
    a.slot_off = slot_off;
    a.lengths = lengths;
    a.n = n;
    a.mode = mode;
    a.init = init;
    a.out_kind = out_kind;
    a.n_views = n_views;
    a.edits = edits;
    a.edit_off = edit_off;
    a.eo_shift = eo_shift;
    a.out = out;
    a.view_stride = view_stride;
    a.max_len = max_len;
    return dispatch_vectorise(k, a, (hipStream_t)stream);
}

int idl_vectorise(const void *codes, const void *mask, const int64_t *slot_off, const int64_t *lengths,
                  int64_t n, int k, int mode, int init, int out_kind,
                  int n_views, const uint32_t *edits, const int64_t *edit_off,
                  void *out, int64_t view_stride, int64_t max_len, void *stream)
{
    return vectorise_impl(codes, mask, slot_off, lengths, n, k, mode, init, out_kind, n_views, edits, edit_off, 0, out, view_stride, max_len, stream);
}

int idl_vectorise_ranges(const void *codes, const void *mask, const int64_t *slot_off, const int64_t *lengths,
                         int64_t n, int k, int mode, int init, int out_kind,
                         int n_views, const uint32_t *edits, const int64_t *edit_ranges,
                         void *out, int64_t view_stride, int64_t max_len, void *stream)
{
    return vectorise_impl(codes, mask, slot_off, lengths, n, k, mode, init, out_kind, n_views, edits, edit_ranges, 1, out, view_stride, max_len, stream);
}

}  // extern "C"
