/* Sanitizer driver for the CPU oracle (test infrastructure).  Built by `make -C oracle asan` with
 * -fsanitize=address,undefined together with idelucs_oracle.c and run by tests/test_host_ingest_asan.py: every oracle entry
 * point over seeded random and edge-case inputs (empty, shorter than k, all N, lengths around the 64-base slot), with every
 * buffer heap-allocated at EXACTLY the size its contract states, so that an over-read or over-write is caught.  It also checks
 * two identities the restatement must satisfy: cgr is a permutation of kmer_counts (reference idelucs/kmers.pyx:53-123 vs
 * :2-50), and pack / apply_edits round-trip a sequence.  Exit 0 = clean. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int orc_kmer_counts(const uint8_t *seq, int64_t len, int k, int32_t *counts);
int orc_cgr(const uint8_t *seq, int64_t len, int k, int32_t *out);
uint32_t orc_reverse_complement(uint32_t x, int k);
int orc_kmer_rev_comp(int32_t *counts, int k, int32_t *out);
int64_t orc_check_sequence(const uint8_t *in, int64_t len, uint8_t *out);
void orc_normalise_f64(const int32_t *counts, int n, double *out);
void orc_apply_edits(uint8_t *seq, int64_t len, const uint32_t *edits, int64_t n_edits);
void orc_pack(const uint8_t *seq, int64_t len, uint8_t *codes, uint8_t *mask);
int64_t orc_mimic_sites(int64_t L, uint32_t seq_index, uint32_t view, double p_ts, double p_tv, uint64_t seed, uint32_t *out, int64_t cap);
int64_t orc_mimic_random_n(int64_t L, uint32_t seq_index, uint32_t view, int n_rand, uint64_t seed, uint32_t *out);

static uint64_t st = 0x2545F4914F6CDD1Dull;
static uint32_t rnd(void) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 16); }
static void *exact(size_t n) { void *p = malloc(n ? n : 1); if (!p) exit(3); return p; }
#define FAIL(...) do { fprintf(stderr, __VA_ARGS__); exit(5); } while (0)

int main(void)
{
    static const char alpha[] = "ACGTACGTACGTACGTNacgtnRYU- \n";
    const int64_t lens[] = {0, 1, 2, 5, 6, 7, 63, 64, 65, 127, 128, 129, 1000, 4097};
    long cases = 0;
    for (int rep = 0; rep < 40; ++rep)
        for (size_t li = 0; li < sizeof lens / sizeof *lens; ++li) {
            const int64_t L = lens[li];
            uint8_t *raw = exact((size_t)L), *clean = exact((size_t)L);
            for (int64_t i = 0; i < L; ++i) raw[i] = (uint8_t)alpha[rep == 0 ? 16 : rnd() % (sizeof alpha - 1)];     /* rep 0: all N */
            int64_t n = orc_check_sequence(raw, L, clean);
            if (n < 0 || n > L) FAIL("check_sequence returned %lld for %lld valid bytes\n", (long long)n, (long long)L);
            uint8_t bad = '!';
            if (orc_check_sequence(&bad, 1, clean) != -1) FAIL("check_sequence accepted '!'\n");
            for (int k = 1; k <= 7; ++k) {
                const int F = 1 << (2 * k);
                int32_t *c = exact((size_t)F * 4), *g = exact((size_t)F * 4), *canon = exact((size_t)F * 4);
                double *fr = exact((size_t)F * 8);
                for (int i = 0; i < F; ++i) c[i] = g[i] = 1;
                orc_kmer_counts(clean, n, k, c);
                orc_cgr(clean, n, k, g);
                int64_t sc = 0, sg = 0;
                for (int i = 0; i < F; ++i) { sc += c[i]; sg += g[i]; }
                if (sc != sg) FAIL("cgr is not a permutation of kmer_counts (k=%d, L=%lld)\n", k, (long long)n);
                orc_normalise_f64(c, F, fr);
                for (uint32_t x = 0; x < (uint32_t)F; x += 1 + rnd() % 7)
                    if (orc_reverse_complement(orc_reverse_complement(x, k), k) != x) FAIL("reverse_complement is not an involution\n");
                const int m = orc_kmer_rev_comp(c, k, canon);
                if (m <= 0 || m > F) FAIL("kmer_rev_comp returned %d rows\n", m);
                free(c); free(g); free(canon); free(fr);
                ++cases;
            }
            /* pack: exactly ceil(n/64) slots of 16 + 8 bytes */
            const int64_t slots = (n + 63) / 64;
            uint8_t *codes = exact((size_t)slots * 16), *mask = exact((size_t)slots * 8);
            orc_pack(clean, n, codes, mask);
            /* mimic sites: count first (cap 0, out NULL is never written), then exactly that many */
            const int64_t ns = orc_mimic_sites(n, (uint32_t)rep, 1u, 1e-2, 0.5e-2, 77u + (uint64_t)rep, NULL, 0);
            uint32_t *e = exact((size_t)ns * 4);
            if (orc_mimic_sites(n, (uint32_t)rep, 1u, 1e-2, 0.5e-2, 77u + (uint64_t)rep, e, ns) != ns) FAIL("mimic_sites is not reproducible\n");
            for (int64_t i = 0; i < ns; ++i) if ((int64_t)(e[i] & 0x3FFFFFFFu) >= n || (e[i] >> 30) == 0) FAIL("mimic site out of range\n");
            uint8_t *mut = exact((size_t)n);
            if (n) memcpy(mut, clean, (size_t)n);
            orc_apply_edits(mut, n, e, ns);
            orc_apply_edits(mut, n, e, ns);                         /* XOR edits are involutions on ACGT; N stays N */
            if (n && memcmp(mut, clean, (size_t)n)) FAIL("apply_edits twice did not restore the sequence\n");
            uint32_t *rn = exact(64 * 4);
            const int64_t nr = orc_mimic_random_n(n, (uint32_t)rep, 3u, 64, 5u, rn);
            if (nr != (n > 0 ? 64 : 0)) FAIL("random_n returned %lld\n", (long long)nr);
            for (int64_t i = 0; i < nr; ++i) if ((int64_t)rn[i] >= n || (i && rn[i] < rn[i - 1])) FAIL("random_n positions unsorted or out of range\n");
            free(raw); free(clean); free(codes); free(mask); free(e); free(mut); free(rn);
        }
    printf("oracle sanitizer driver: %ld histogram cases clean\n", cases);
    return 0;
}
