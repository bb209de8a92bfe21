// Is q' = fma(fma(-q, S, c), r, q) with r = RN(1/S), q = RN(c*r) always equal to RN(c/S) for integers 1 <= c <= S < 2^24 - 1 ?
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
static inline float mk(float c, float S, float r) { float q = c * r; float rem = fmaf(-q, S, c); return fmaf(rem, r, q); }
int main(int argc, char **argv) {
    long bad = 0, total = 0;
    // (1) dense block: every S in [4097, 4097+200000), every c in [1, min(S, 70000)]
    #pragma omp parallel for reduction(+:bad,total) schedule(dynamic, 256)
    for (long S = 4097; S < 4097 + 200000; ++S) {
        float Sf = (float)S, r = 1.0f / Sf;
        long cmax = S < 70000 ? S : 70000;
        for (long c = 1; c <= cmax; ++c) {
            float ref = (float)((double)c / (double)S);
            if (mk((float)c, Sf, r) != ref) ++bad;
            ++total;
        }
    }
    printf("dense: %ld pairs, %ld mismatches\n", total, bad);
    // (2) random over the whole domain
    long bad2 = 0, tot2 = 0;
    #pragma omp parallel for reduction(+:bad2,tot2)
    for (int t = 0; t < 64; ++t) {
        uint64_t x = 0x9E3779B97F4A7C15ull * (t + 1);
        for (long i = 0; i < 40000000; ++i) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            uint32_t S = (uint32_t)(x % 16777214u) + 1u;            // 1 .. 2^24-2
            uint32_t c = (uint32_t)((x >> 32) % S) + 1u;
            float Sf = (float)S, r = 1.0f / Sf;
            if (mk((float)c, Sf, r) != (float)((double)c / (double)S)) ++bad2;
            ++tot2;
        }
    }
    printf("random: %ld pairs, %ld mismatches\n", tot2, bad2);
    // (3) the excluded operand
    float Sf = 16777215.0f, r = 1.0f / Sf; long b3 = 0;
    for (long c = 1; c <= 16777215; ++c) if (mk((float)c, Sf, r) != (float)((double)c / 16777215.0)) ++b3;
    printf("S = 2^24-1: %ld mismatches\n", b3);
    return 0;
}
