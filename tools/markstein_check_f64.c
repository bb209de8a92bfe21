// float32( (double)t / s ) vs float32( markstein64(t, s) ) for float t, double s:  q = t*r, q' = fma(fma(-q, s, t), r, q), r = RN(1/s)
#include <math.h>
#include <stdio.h>
#include <stdint.h>
int main() {
    long bad = 0, bad64 = 0, tot = 0;
    #pragma omp parallel for reduction(+:bad,bad64,tot)
    for (int th = 0; th < 64; ++th) {
        uint64_t x = 0x9E3779B97F4A7C15ull * (th + 7);
        for (long i = 0; i < 60000000; ++i) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            // scale ~ 1e-6 .. 1e-3 (std of k-mer frequencies), t ~ +-1e-3 float
            double s = ldexp(1.0 + (double)(x >> 12) / 4503599627370496.0, -20 + (int)(x % 11));
            float t = (float)((double)(int64_t)(x >> 20) / 4.0e15 * 1e-3) * ((x & 1) ? 1.f : -1.f);
            double r = 1.0 / s, q = (double)t * r, q2 = fma(fma(-q, s, (double)t), r, q);
            double ref = (double)t / s;
            if (q2 != ref) ++bad64;
            if ((float)q2 != (float)ref) ++bad;
            ++tot;
        }
    }
    printf("%ld samples: %ld float64 quotients differ, %ld float32 results differ\n", tot, bad64, bad);
    return 0;
}
