// fmk_exp.h -- exp(x) as THE HOST computes it, on the device: glibc's double-precision exp (2.28 and later:
// sysdeps/ieee754/dbl-64/e_exp.c, the ARM optimized-routines algorithm) restated operation by operation with the FMA contractions of
// libm's `fma` build, the variant x86-64 hosts with FMA3 select (ifunc) -- read off the instructions of __exp_fma in this image's
// libm.so.6 (glibc 2.35), the way csrc/fmk_log.h restates log.  ewmst's per-tick weight is alpha = 1 - exp(-dt / half_life)
// (feature/core/volatility.py:178-179): for gaps of micro- and milliseconds against a half life of seconds exp(x) is 1 - k 2^-53 with a
// small k, so alpha carries the ROUNDING of exp in its leading digits and the reference's outputs carry it with them; the device
// library's exp is as accurate but rounds in its own way (round 5: a worst case of 3e-9 relative on ewmst, met by a mixed contract).
//   exp(x) = 2^(k / N) e^r,  N = 128,  k = RN(x N / ln2) by the shift trick,  r = x - k ln2 / N in two parts, |r| <= ln2 / 256;
//   2^(k / N) = (scale, tail) from the table (fmk_exptab.h: glibc's own, extracted from the host's libm by tools/extract_glibc_tables.py);
//   e^r - 1 ~ r + r^2 (C2 + r C3) + r^4 (C4 + r C5);   result = scale + scale * (tail + that).
// fmk_exp_small is the same sequence for k == 0 (|x| < ln2 / 256 = 2.7e-3: every tick gap below 0.27 % of the half life), where the
// table entry is (1, 0) and drops out: 8 instructions.  |x| < 2^-54 -> 1 + x (either form); |x| >= 512 goes through glibc's special case (results
// near the overflow / underflow threshold are assembled in two steps); |x| >= 1024, NaN and the infinities like glibc.
// tools/logratio_check.c runs THE SAME SOURCE on the host against exp() over the whole double range; tests/test_host_logic.py runs it.
#pragma once
#include <math.h>
#include <stdint.h>

#ifdef __HIPCC__
#define FMK_EXP_CONST static __device__ const
#define FMK_EXP_FN __device__ __forceinline__
#define FMK_EXP_FAR __device__ __noinline__ static
#else
#define FMK_EXP_CONST static const
#define FMK_EXP_FN static inline
#define FMK_EXP_FAR static
#endif
#include "fmk_exptab.h"
// |x| below this: fma(x, FMK_EXP_INVLN2N, FMK_EXP_SHIFT) == FMK_EXP_SHIFT, i.e. k == 0 (a little under ln2 / 256)
#define FMK_EXP_SMALL_BELOW 0x1.62e42fefa39p-9

// k == 0: the caller has checked fma(x, FMK_EXP_INVLN2N, FMK_EXP_SHIFT) == FMK_EXP_SHIFT (|x| < 2^-54, glibc's 1 + x, comes out the same: the
// polynomial's terms vanish against x and 1 + x rounds to 1)
FMK_EXP_FN double fmk_exp_small(double x)
{
    const double r2 = x * x;
    const double t23 = fma(x, FMK_EXP_C3, FMK_EXP_C2), t45 = fma(x, FMK_EXP_C5, FMK_EXP_C4);
    const double tmp = fma(r2 * r2, t45, fma(t23, r2, x));          // tail + r == r: the tail of 2^0 is +0
    return 1.0 + tmp;                                                // fma(1, tmp, 1)
}

// every argument (not inlined on the device: a table look-up per lane, rarely run)
FMK_EXP_FAR double fmk_exp_general(double x)
{
    union { double d; uint64_t u; } v;
    v.d = x;
    unsigned abstop = (unsigned)(v.u >> 52) & 0x7ffu;
    if (abstop - 0x3c9u > 0x3eu) {
        if (abstop < 0x3c9u) return 1.0 + x;                         // |x| < 2^-54
        if (abstop >= 0x409u) {                                      // |x| >= 1024, inf, NaN
            if (v.u == 0xfff0000000000000ull) return 0.0;
            if (abstop >= 0x7ffu) return 1.0 + x;
            return (v.u >> 63) ? 0.0 : INFINITY;                     // glibc: __math_uflow / __math_oflow
        }
        abstop = 0;                                                  // |x| in [512, 1024): the result may leave the normal range
    }
    double kd = fma(x, FMK_EXP_INVLN2N, FMK_EXP_SHIFT);
    v.d = kd;
    const uint64_t ki = v.u;
    kd -= FMK_EXP_SHIFT;
    const double r = fma(kd, FMK_EXP_NEGLN2LON, fma(kd, FMK_EXP_NEGLN2HIN, x));
    const unsigned idx = 2u * (unsigned)(ki & 127u);
    v.u = FMK_EXP_T[idx];
    const double tail = v.d;
    uint64_t sbits = FMK_EXP_T[idx + 1] + (ki << 45);
    const double r2 = r * r;
    const double t23 = fma(r, FMK_EXP_C3, FMK_EXP_C2), t45 = fma(r, FMK_EXP_C5, FMK_EXP_C4);
    const double tmp = fma(r2 * r2, t45, fma(t23, r2, tail + r));
    if (abstop == 0) {
        if ((ki & 0x80000000ull) == 0) {                             // k > 0: the scale may overflow, the result need not
            v.u = sbits - (1009ull << 52);
            return 0x1p1009 * fma(v.d, tmp, v.d);
        }
        v.u = sbits + (1022ull << 52);                               // k < 0: the result may be subnormal -- round it once
        const double scale = v.d, st = scale * tmp;
        double y = scale + st;
        if (y < 1.0) {
            double lo = (scale - y) + st;
            const double hi = 1.0 + y;
            lo = ((1.0 - hi) + y) + lo;
            y = (hi + lo) - 1.0;
            if (y == 0.0) y = 0.0;                                   // not -0
        }
        return 0x1p-1022 * y;
    }
    v.u = sbits;
    return fma(v.d, tmp, v.d);
}

// the host's exp(x)
FMK_EXP_FN double fmk_exp_host(double x)
{
    if (fma(x, FMK_EXP_INVLN2N, FMK_EXP_SHIFT) == FMK_EXP_SHIFT) return fmk_exp_small(x);
    return fmk_exp_general(x);
}
