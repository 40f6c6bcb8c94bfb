/* The logarithm of tick returns (fmk_log_ratio in csrc/fmk_common.h) against the HOST's log(), which is what the oracle and
 * Numba-compiled reference code call.  For x = p / pm in [1 - 2^-4, 1 + 0x1.09p-4) glibc's log (2.28+, the ARM
 * optimized-routines algorithm) takes a table-free branch; near1() restates that branch with the FMA contractions of libm's `fma`
 * build (what x86-64 hosts with FMA3 select) -- the device code is the same sequence of IEEE operations.  This program counts the
 * arguments on which the restatement and the host's log() differ: price quotients a few ticks apart, and a sweep of the interval.
 * -DNO_FMA evaluates the same source without contractions (libm's generic build): how far two glibc variants are apart.
 *     gcc -O2 -ffp-contract=off -mfma tools/logratio_check.c -lm && ./a.out            -> "0 ... 0" on this image
 * exit status: 0 iff there is no difference. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

static const double B[11] = {
 -0x1p-1, 0x1.5555555555577p-2, -0x1.ffffffffffdcbp-3, 0x1.999999995dd0cp-3, -0x1.55555556745a7p-3, 0x1.24924a344de3p-3,
 -0x1.fffffa4423d65p-4, 0x1.c7184282ad6cap-4, -0x1.999eb43b068ffp-4, 0x1.78182f7afd085p-4, -0x1.5521375d145cdp-4 };

static double near1(double x)
{
    const double r = x - 1.0, r2 = r * r, r3 = r * r2;
    double w = r * 0x1p27;
    const double rhi = r + w - w, rlo = r - rhi;
    w = rhi * rhi * B[0];
    const double hi = r + w;
    double lo = r - hi + w;
#ifndef NO_FMA
    double q = fma(r3, B[10], fma(r2, B[9], fma(r, B[8], B[7])));
    q = fma(r3, q, fma(r2, B[6], fma(r, B[5], B[4])));
    q = fma(r3, q, fma(r2, B[3], fma(r, B[2], B[1])));
    lo = fma(B[0] * rlo, rhi + r, lo);
    return fma(r3, q, lo) + hi;
#else
    double y = r3 * (B[1] + r * B[2] + r2 * B[3] + r3 * (B[4] + r * B[5] + r2 * B[6] + r3 * (B[7] + r * B[8] + r2 * B[9] + r3 * B[10])));
    lo += B[0] * rlo * (rhi + r);
    y += lo;
    y += hi;
    return y;
#endif
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 20000000;
    uint64_t s = 88172645463325252ULL;
    long diff = 0, tried = 0, far = 0;
    for (long i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double base = 0.5 + (double)(s % 4000000) * 0.01;           /* 0.5 .. 40000 on a 0.01 grid */
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const int k = (int)(s % 41) - 20;                                 /* up to 20 ticks of 0.01 (and k = 0) */
        const double p = base + k * 0.01, x = p / base;
        if (!(p > 0.0)) continue;
        if (!(x >= 0.9375 && x < 0x1.109p+0)) { ++far; continue; }        /* the device library's log serves these */
        ++tried;
        if (near1(x) != log(x)) ++diff;
    }
    long diff2 = 0, n2 = 0;
    for (double x = 0.9375; x < 0x1.109p+0; x += 0x1.3p-29) { ++n2; if (near1(x) != log(x)) ++diff2; }
    printf("price quotients: %ld of %ld differ from the host's log (%ld more lie outside the table-free interval); "
           "sweep of the interval: %ld of %ld differ\n", diff, tried, far, diff2, n2);
    return diff || diff2 ? 1 : 0;
}
