/* The device's logarithm (csrc/fmk_log.h: glibc's double-precision log restated with the FMA contractions of libm's `fma` build)
 * against the HOST's log(), which is what the oracle and Numba-compiled reference code call.  The header is plain C as well: this
 * program runs the SAME source on the host and counts the arguments on which it differs from log():
 *   (1) price quotients a few ticks apart (the table-free branch around 1),
 *   (2) a sweep of that branch's interval,
 *   (3) quotients of prices up to a factor 2^40 apart, random doubles over the WHOLE positive range (every exponent, subnormals
 *       included) and the special values (the table branch).
 *     gcc -O2 -ffp-contract=off -mfma tools/logratio_check.c -lm && ./a.out            -> all zeros on this image
 * exit status: 0 iff there is no difference. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../finmlkit_amd/csrc/fmk_log.h"
#include "../finmlkit_amd/csrc/fmk_exp.h"

static uint64_t s = 88172645463325252ULL;
static uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static int same(double a, double b)
{
    if (a != a && b != b) return 1;                                   /* NaN for NaN (the sign is not part of the contract) */
    return memcmp(&a, &b, 8) == 0;
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 20000000;
    long diff = 0, tried = 0, far = 0, diff_far = 0;
    for (long i = 0; i < n; ++i) {
        const double base = 0.5 + (double)(rnd() % 4000000) * 0.01;      /* 0.5 .. 40000 on a 0.01 grid */
        const int k = (int)(rnd() % 41) - 20;                             /* up to 20 ticks of 0.01 (and k = 0) */
        const double p = base + k * 0.01, x = p / base;
        if (!(p > 0.0)) continue;
        if (!(x >= 0.9375 && x < 0x1.109p+0)) { ++far; if (!same(fmk_log_host(x), log(x))) ++diff_far; continue; }
        ++tried;
        if (!same(fmk_log_host(x), log(x))) ++diff;
    }
    long diff2 = 0, n2 = 0;
    for (double x = 0.9375; x < 0x1.109p+0; x += 0x1.3p-29) { ++n2; if (!same(fmk_log_host(x), log(x))) ++diff2; }
    /* the table branch */
    long diff3 = 0, n3 = 0;
    for (long i = 0; i < n; ++i) {                                        /* quotients of far-apart prices */
        const double a = 0.01 * (double)(1 + rnd() % 100000000), b = ldexp(0.01 * (double)(1 + rnd() % 100000000), (int)(rnd() % 81) - 40);
        const double x = a / b;
        ++n3; if (!same(fmk_log_host(x), log(x))) ++diff3;
    }
    for (long i = 0; i < n; ++i) {                                        /* random bit patterns: every exponent, both signs, NaNs */
        union { uint64_t u; double d; } v;
        v.u = rnd();
        ++n3; if (!same(fmk_log_host(v.d), log(v.d))) ++diff3;
        v.u &= 0x000fffffffffffffULL;                                     /* ... and a subnormal */
        ++n3; if (!same(fmk_log_host(v.d), log(v.d))) ++diff3;
    }
    const double special[] = {0.0, -0.0, 1.0, INFINITY, -INFINITY, NAN, 0x1p-1074, 0x1p-1022, 0x1.fffffffffffffp1023, 0.9375, 0x1.109p+0,
                              0x1.108ffffffffffp+0, 0x1.dffffffffffffp-1, -1.0, 2.0, 0.5, 0x1.6p-1, 0x1.5ffffffffffffp-1, 0x1.6p0};
    for (unsigned i = 0; i < sizeof special / sizeof special[0]; ++i) { ++n3; if (!same(fmk_log_host(special[i]), log(special[i]))) ++diff3; }
    for (int e = -1074; e <= 1023; ++e)                                   /* around every power of two */
        for (int d = -2; d <= 2; ++d) {
            union { uint64_t u; double d; } v;
            v.d = ldexp(1.0, e);
            v.u += (uint64_t)(int64_t)d;
            ++n3; if (!same(fmk_log_host(v.d), log(v.d))) ++diff3;
        }
    printf("price quotients: %ld of %ld differ from the host's log (+ %ld of the %ld outside the table-free interval); "
           "sweep of the interval: %ld of %ld differ; table branch, whole double range: %ld of %ld differ\n",
           diff, tried, diff_far, far, diff2, n2, diff3, n3);
    /* exp (csrc/fmk_exp.h against the host's exp()):
     *   (4) ewmst's arguments -dt / half_life: gaps of 1 ns .. 100 s against half lives of 0.1 s .. 1 h (the k == 0 form and its edge),
     *   (5) a sweep across the edge of the k == 0 form and of every table entry, random doubles over the WHOLE range (both signs, every
     *       exponent), the overflow / underflow / subnormal-result ranges and the special values. */
    long diff4 = 0, n4 = 0, small4 = 0, diff5 = 0, n5 = 0;
    for (long i = 0; i < n; ++i) {
        const double gap_ns = (double)(1 + rnd() % (i & 1 ? 100000000000ULL : 5000000ULL));
        const double hl = i & 2 ? 60.0 : 0.1 * (double)(1 + rnd() % 36000);
        const double x = -((gap_ns / 1e9) / hl);
        ++n4; if (!same(fmk_exp_host(x), exp(x))) ++diff4;
        if (fma(x, FMK_EXP_INVLN2N, FMK_EXP_SHIFT) == FMK_EXP_SHIFT) ++small4;
    }
    for (double x = -0.75; x < 0.75; x += 0x1.3p-27) { ++n5; if (!same(fmk_exp_host(x), exp(x))) ++diff5; }
    for (long i = 0; i < n; ++i) {
        union { uint64_t u; double d; } v;
        v.u = rnd();
        ++n5; if (!same(fmk_exp_host(v.d), exp(v.d))) ++diff5;
        const double x = -760.0 + 1480.0 * (double)(rnd() >> 11) * 0x1p-53;          /* [-760, 720): results over the whole range */
        ++n5; if (!same(fmk_exp_host(x), exp(x))) ++diff5;
        const double u = -745.2 + 37.0 * (double)(rnd() >> 11) * 0x1p-53;            /* subnormal results */
        ++n5; if (!same(fmk_exp_host(u), exp(u))) ++diff5;
    }
    const double especial[] = {0.0, -0.0, 1.0, -1.0, INFINITY, -INFINITY, NAN, 0x1p-1074, -0x1p-1074, 0x1p-54, -0x1p-54, 0x1.fffffffffffffp-55,
                               -0x1.fffffffffffffp-55, 512.0, -512.0, 1024.0, -1024.0, 709.782712893384, 709.7827128933841, -708.3964185322641,
                               -745.1332191019411, -745.1332191019412, 0x1.62e42fefa39efp-9, -0x1.62e42fefa39efp-9, 0x1.fffffffffffffp1023};
    for (unsigned i = 0; i < sizeof especial / sizeof especial[0]; ++i) { ++n5; if (!same(fmk_exp_host(especial[i]), exp(especial[i]))) ++diff5; }
    for (int e = -1074; e <= 1023; ++e)
        for (int d = -2; d <= 2; ++d)
            for (int sg = 0; sg < 2; ++sg) {
                union { uint64_t u; double d; } v;
                v.d = ldexp(1.0, e);
                v.u += (uint64_t)(int64_t)d;
                if (sg) v.d = -v.d;
                ++n5; if (!same(fmk_exp_host(v.d), exp(v.d))) ++diff5;
            }
    printf("exp of -dt / half_life: %ld of %ld differ from the host's exp (%ld through the table-free form); exp, whole double range: %ld of %ld differ\n",
           diff4, n4, small4, diff5, n5);
    return diff || diff2 || diff3 || diff_far || diff4 || diff5 ? 1 : 0;
}
