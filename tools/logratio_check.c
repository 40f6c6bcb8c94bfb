/* Accuracy of the short-range logarithm used for tick returns (fmk_log_ratio in fmk_common.h) against glibc's log and a
 * long double reference, on quotients of prices that differ by a few ticks.    gcc -O2 -ffp-contract=off tools/logratio_check.c -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

static double log_ratio(double p, double pm)
{
    const double x = p / pm, f = x - 1.0;
    if (!(fabs(f) <= 0.015625)) return log(x);
    double q = -1.0 / 12.0;
    q = fma(f, q, 1.0 / 11.0); q = fma(f, q, -1.0 / 10.0); q = fma(f, q, 1.0 / 9.0); q = fma(f, q, -1.0 / 8.0);
    q = fma(f, q, 1.0 / 7.0); q = fma(f, q, -1.0 / 6.0); q = fma(f, q, 1.0 / 5.0); q = fma(f, q, -1.0 / 4.0);
    q = fma(f, q, 1.0 / 3.0); q = fma(f, q, -0.5);
    return fma(f * f, q, f);
}

int main(void)
{
    uint64_t s = 88172645463325252ULL;
    long n = 20000000, diff_glibc = 0, not_cr_mine = 0, not_cr_glibc = 0;
    double worst = 0.0;
    for (long i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double base = 0.5 + (double)(s % 4000000) * 0.01;           /* 0.5 .. 40000 on a 0.01 grid */
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const int k = (int)(s % 41) - 20;                                 /* up to 20 ticks of 0.01 (and k = 0) */
        const double p = base + k * 0.01, pm = base;
        if (!(p > 0.0)) continue;
        const double mine = log_ratio(p, pm), g = log(p / pm);
        const long double ref = logl((long double)(p / pm));
        const double cr = (double)ref;                                    /* correctly rounded up to double rounding */
        if (mine != g) ++diff_glibc;
        if (mine != cr) ++not_cr_mine;
        if (g != cr) ++not_cr_glibc;
        if (ref != 0.0L) {
            const double e = (double)fabsl(((long double)mine - ref) / (ref * 0x1p-53L));   /* in units of 2^-53 relative */
            if (e > worst) worst = e;
        }
    }
    printf("%ld quotients: short-range log != glibc log in %ld; not the rounded long-double value: short-range %ld, glibc %ld; worst relative error %.3f x 2^-53\n",
           n, diff_glibc, not_cr_mine, not_cr_glibc, worst);
    return 0;
}
