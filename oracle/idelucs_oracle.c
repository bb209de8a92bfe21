/*
 * oracle/idelucs_oracle.c  --  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded restatement of the reference algorithm for the
 * hot path (Kari-Genomics-Lab/iDeLUCS @ 2024_08_07).  It is the checker the
 * HIP path is compared against; it is NOT shipped, NOT a fallback, and nothing
 * under idelucs_amd/ may import, link or call it.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit against
 * golden vectors produced by importing the real reference in the build
 * container (tests/golden/make_golden.py -> tests/golden/ npz + json files;
 * tests/test_oracle_*.py).  The reference's own tests hold no vectors
 * (reference tests/test_import.py:1-6 only imports the package).
 *
 * Each function cites the reference lines it restates (paths relative to the
 * reference repo root).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- base code: A0 C1 G2 T3, everything else 4 ("skip").
 * Restates the 256-entry `encoding` table, idelucs/kmers.pyx:19-34: only the
 * UPPER-CASE bytes 65,67,71,84 map to 0..3. */
static inline unsigned orc_code(uint8_t b)
{
    switch (b) {
    case 'A': return 0u;
    case 'C': return 1u;
    case 'G': return 2u;
    case 'T': return 3u;
    default:  return 4u;
    }
}

/* idelucs/kmers.pyx:2-50  kmer_counts(seq, k, counts): sliding window, first
 * base of a k-mer in the most significant 2-bit pair, window restarts after any
 * byte that is not A/C/G/T, ACCUMULATES into counts[4^k] (int32, wraps). */
int orc_kmer_counts(const uint8_t *seq, int64_t len, int k, int32_t *counts)
{
    if (k < 1 || k > 15 || len < 0) return -1;
    const uint32_t mask = (uint32_t)((1u << (2 * k)) - 1u);   /* kmers.pyx:17 */
    uint32_t kmer = 0;
    int countdown = k - 1;                                     /* kmers.pyx:15 */
    for (int64_t i = 0; i < len; ++i) {
        unsigned c = orc_code(seq[i]);
        if (c == 4u) countdown = k;                            /* kmers.pyx:42-43 */
        kmer = ((kmer << 2) | c) & mask;                       /* kmers.pyx:45 */
        if (countdown == 0)
            counts[kmer] = (int32_t)((uint32_t)counts[kmer] + 1u);   /* kmers.pyx:47-48 */
        else
            countdown -= 1;                                    /* kmers.pyx:49-50 */
    }
    return 0;
}

/* idelucs/kmers.pyx:53-123  cgr(seq, k, CGR): chaos-game pixel of each window.
 * i-bit: A1 C0 G0 T1 (kmers.pyx:70-85), j-bit: A0 C0 G1 T1 (kmers.pyx:87-102);
 * the NEWEST base enters at bit n_bp (kmers.pyx:110-112) and the oldest leaves
 * by >>1 after each emitted window (kmers.pyx:120-123); index (i<<k)+j. */
int orc_cgr(const uint8_t *seq, int64_t len, int k, int32_t *out)
{
    if (k < 1 || k > 15 || len < 0) return -1;
    uint32_t ci = 0, cj = 0;
    int n_bp = 0;
    for (int64_t i = 0; i < len; ++i) {
        unsigned ib, jb;
        switch (seq[i]) {
        case 'A': ib = 1; jb = 0; break;
        case 'C': ib = 0; jb = 0; break;
        case 'G': ib = 0; jb = 1; break;
        case 'T': ib = 1; jb = 1; break;
        default:  ib = 2; jb = 2; break;
        }
        if (ib < 2) {
            ci |= ib << n_bp;
            cj |= jb << n_bp;
            n_bp += 1;
        } else {
            n_bp = 0; ci = 0; cj = 0;                          /* kmers.pyx:113-116 */
        }
        if (n_bp == k) {
            uint32_t idx = (ci << k) + cj;
            out[idx] = (int32_t)((uint32_t)out[idx] + 1u);
            ci >>= 1; cj >>= 1; n_bp -= 1;
        }
    }
    return 0;
}

/* idelucs/utils.py:191-206  reverse_complement(x, k): swap the two bits of each
 * pair, complement within 2k bits, then reverse the 2k-bit string.  (Net effect:
 * the true reverse complement under A0 C1 G2 T3.)  Restated literally. */
uint32_t orc_reverse_complement(uint32_t x, int k)
{
    const int numbits = 2 * k;
    const uint32_t m = 0xAAAAAAAAu;
    x = ((x >> 1) & (m >> 1)) | ((x << 1) & m);
    x = (uint32_t)(((uint64_t)1 << numbits) - 1u - x);
    uint32_t rev = 0;
    for (int b = 0; b < numbits; ++b) { rev = (rev << 1) | (x & 1u); x >>= 1; }
    return rev;
}

/* idelucs/utils.py:208-221  kmer_rev_comp(counts, k): for every kmer <= rc(kmer)
 * in ascending order: c[kmer] = int32((c[kmer] + c[rc]) * 0.5) -- the product is
 * a float64 stored back into an int32 array (truncation toward zero); a
 * palindrome is doubled then halved.  Returns the canonical entries, ascending.
 * `counts` is modified in place exactly like the reference does. */
int orc_kmer_rev_comp(int32_t *counts, int k, int32_t *out)
{
    const uint32_t n = 1u << (2 * k);
    int m = 0;
    for (uint32_t kmer = 0; kmer < n; ++kmer) {
        uint32_t rc = orc_reverse_complement(kmer, k);
        if (kmer <= rc) {
            /* numpy int32 += int32 wraps; then int32 * 0.5 -> float64 -> int32 (C cast) */
            int32_t s = (int32_t)((uint32_t)counts[kmer] + (uint32_t)counts[rc]);
            counts[kmer] = s;
            counts[kmer] = (int32_t)((double)counts[kmer] * 0.5);
            out[m++] = counts[kmer];
        }
    }
    return m;
}

/* idelucs/utils.py:42-50  check_sequence translate step: lower->upper, u/U->T,
 * IUPAC ambiguity codes and '-' -> N, delete " \t\n\r"; any other byte is an
 * error.  Returns the output length, or -(pos+1) of the first offending INPUT
 * byte whose translation is not in ACGTN. */
int64_t orc_check_sequence(const uint8_t *in, int64_t len, uint8_t *out)
{
    static const char from[] = "acgtuUswkmyrbdhvnSWKMYRBDHV-";
    static const char to[]   = "ACGTTTNNNNNNNNNNNNNNNNNNNNNN";
    uint8_t tab[256];
    for (int i = 0; i < 256; ++i) tab[i] = (uint8_t)i;
    for (int i = 0; from[i]; ++i) tab[(uint8_t)from[i]] = (uint8_t)to[i];
    int64_t n = 0;
    for (int64_t i = 0; i < len; ++i) {
        uint8_t b = in[i];
        if (b == ' ' || b == '\t' || b == '\n' || b == '\r') continue;
        uint8_t t = tab[b];
        if (!(t == 'A' || t == 'C' || t == 'G' || t == 'T' || t == 'N')) return -(i + 1);
        out[n++] = t;
    }
    return n;
}

/* idelucs/utils.py:242-250: counts start at ONE, then counts / sum(counts) in
 * float64.  `counts` here already include the pseudocount. */
void orc_normalise_f64(const int32_t *counts, int n, double *out)
{
    int64_t s = 0;
    for (int i = 0; i < n; ++i) s += counts[i];
    for (int i = 0; i < n; ++i) out[i] = (double)counts[i] / (double)s;
}

/* ------------------------------------------------------------------------
 * Below: restatements of THIS repo's device-side conventions (not of reference
 * code) so that device buffers can be checked independently of the product's
 * host code.
 * ------------------------------------------------------------------------ */

/* Substitution-edit semantics used by the device vectoriser: edit = pos | op<<30,
 * op 0 = set invalid (N), op 1..3 = XOR the 2-bit code (A0 C1 G2 T3) with op.
 * An XOR on an invalid byte leaves it invalid.  Applied to a cleaned ACGTN byte
 * string this must reproduce what the reference transforms (utils.py:54-135) do
 * for the same sites/choices. */
void orc_apply_edits(uint8_t *seq, int64_t len, const uint32_t *edits, int64_t n_edits)
{
    static const char dec[4] = {'A', 'C', 'G', 'T'};
    for (int64_t e = 0; e < n_edits; ++e) {
        uint32_t pos = edits[e] & 0x3FFFFFFFu, op = edits[e] >> 30;
        if ((int64_t)pos >= len) continue;
        if (op == 0u) { seq[pos] = 'N'; continue; }
        unsigned c = orc_code(seq[pos]);
        if (c == 4u) continue;
        seq[pos] = (uint8_t)dec[c ^ op];
    }
}

/* Packed device layout (include/idelucs_hip.h): sequence s occupies slots of 64 bases; per slot
 * four little-endian uint32 code words (base j of word w at bits 31-2j..30-2j: first base in the
 * most significant pair) and two little-endian uint32 mask words (base j of word w at bit 31-j,
 * 1 = not A/C/G/T).  Positions past the sequence end are marked invalid. */
void orc_pack(const uint8_t *seq, int64_t len, uint8_t *codes, uint8_t *mask)
{
    int64_t slots = (len + 63) / 64;
    uint32_t *cw = (uint32_t *)codes, *mw = (uint32_t *)mask;
    memset(codes, 0, (size_t)slots * 16);
    memset(mask, 0, (size_t)slots * 8);
    for (int64_t i = 0; i < slots * 64; ++i) {
        unsigned c = (i < len) ? orc_code(seq[i]) : 4u;
        if (c == 4u) mw[i >> 5] |= 0x80000000u >> (i & 31);
        else cw[i >> 4] |= (uint32_t)c << (30 - 2 * (i & 15));
    }
}

/* ------------------------------------------------------------------------
 * Restatement of the "fast mode" mimic generator spec (idelucs_amd/csrc/mimic.hip header):
 * Philox4x32-10, exact geometric gap sampling from an integer threshold table, per-lane segments
 * of ceil(L/64) bases, Random_N by mulhi + sort.  Statistically equivalent to the reference's
 * transforms (idelucs/utils.py:54-135); bit-exact with the device by construction of the spec.
 * ------------------------------------------------------------------------ */
#define ORC_J 1024

static void orc_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* site view: returns the number of edits written (sorted by position) */
int64_t orc_mimic_sites(int64_t L, uint32_t seq_index, uint32_t view, double p_ts, double p_tv, uint64_t seed,
                        uint32_t *out, int64_t cap)
{
    const double keep = (1.0 - p_ts) * (1.0 - p_tv);
    const double q = 1.0 - keep;
    if (!(q > 0.0)) return 0;
    static uint32_t T[ORC_J + 1];
    T[0] = 0xFFFFFFFFu;
    double x = 1.0;
    for (int j = 1; j <= ORC_J; ++j) { x = x * keep; T[j] = (uint32_t)(x * 4294967296.0); }
    const double fa = (p_ts * (1.0 - p_tv)) / q * 4294967296.0;
    const double fb = ((1.0 - p_ts) * p_tv) / q * 4294967296.0;
    const double Ad = fa >= 4294967295.0 ? 4294967295.0 : fa;
    double Bd = Ad + fb;
    if (Bd > 4294967295.0) Bd = 4294967295.0;
    const uint32_t A = (uint32_t)Ad, B = (uint32_t)Bd;
    const int kind = (p_tv == 0.0) ? 1 : (p_ts == 0.0) ? 2 : 0;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const int64_t seg = (L + 63) / 64;
    int64_t n = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int64_t lo = (int64_t)lane * seg;
        int64_t hi = lo + seg;
        if (hi > L) hi = L;
        int64_t pos = lo - 1;
        for (uint32_t d = 0; pos < hi; ++d) {
            uint32_t r[4];
            orc_philox(d, (uint32_t)lane, seq_index, view, k0, k1, r);
            int a = 0, b = ORC_J;
            while (a < b) { int m = (a + b + 1) >> 1; if (r[0] < T[m]) a = m; else b = m - 1; }
            if (a == ORC_J) { pos += ORC_J; continue; }
            pos += a + 1;
            if (pos >= hi) break;
            const uint32_t flav = 1u | ((r[2] & 1u) << 1);
            uint32_t op = (r[1] < A) ? 2u : (r[1] < B) ? flav : (2u ^ flav);
            if (kind == 1) op = 2u;
            if (kind == 2) op = flav;
            if (n < cap) out[n] = (uint32_t)pos | (op << 30);
            ++n;
        }
    }
    return n;
}

static int orc_cmp_u32(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

/* Random_N view: n_rand <= 64 positions, ascending, op 0 */
int64_t orc_mimic_random_n(int64_t L, uint32_t seq_index, uint32_t view, int n_rand, uint64_t seed, uint32_t *out)
{
    if (L <= 0) return 0;
    for (int i = 0; i < n_rand; ++i) {
        uint32_t r[4];
        orc_philox((uint32_t)i, 0u, seq_index, view | (1u << 16), (uint32_t)seed, (uint32_t)(seed >> 32), r);
        out[i] = (uint32_t)(((uint64_t)r[0] * (uint64_t)(uint32_t)L) >> 32);
    }
    qsort(out, (size_t)n_rand, sizeof(uint32_t), orc_cmp_u32);
    return n_rand;
}
