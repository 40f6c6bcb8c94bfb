// fmk_log.h -- log(x) as THE HOST computes it, on the device: glibc's double-precision log (2.28 and later:
// sysdeps/ieee754/dbl-64/e_log.c, the ARM optimized-routines algorithm) restated operation by operation with the FMA contractions
// of libm's `fma` build, the variant x86-64 hosts with FMA3 select (ifunc) -- read off the instructions of __log_fma in this
// image's libm.so.6 (glibc 2.35).  The CPU oracle and Numba-compiled reference code call the host's log(), and a CUSUM close can
// turn on its last bit (logic.py:196-200), so the device's own log() -- correctly rounded in a different way -- is not good enough.
//   * x in [1 - 2^-4, 1 + 0x1.09p-4): r = x - 1 and a degree-11 polynomial whose leading terms r - r^2 / 2 are formed in
//     double-double (every price quotient of ticks less than 6 % apart);
//   * every other positive finite x: x = 2^k z with z in [0x1.6p-1, 0x1.6p0), the subinterval i of 128 that holds z gives c_i
//     near its centre with 1 / c_i and log c_i tabulated (fmk_logtab.h: glibc's own table, extracted from the host's libm by
//     tools/extract_glibc_tables.py), r = fma(z, 1 / c_i, -1), log x = k ln2 + log c_i + log1p(r) with a degree-5 polynomial
//     (round 5 handed these arguments to the device library's log);
//   * subnormals are scaled by 2^52 first; +0 / -0 -> -inf, negative -> NaN, +inf -> +inf, NaN -> NaN like glibc (the sign of a NaN
//     result is not part of the contract).
// tools/logratio_check.c holds the same sequence in C and compares it with the host's log() over the whole double range;
// tests/test_host_logic.py runs it.  The contract is the FMA build: libm's generic build differs on ~8 arguments in 10^6.
#pragma once
#include <math.h>
#include <stdint.h>

#ifdef __HIPCC__
#define FMK_LOG_CONST static __device__ const
#define FMK_LOG_FN __device__ __forceinline__
#define FMK_LOG_FAR __device__ __noinline__ static
#else
#define FMK_LOG_CONST static const
#define FMK_LOG_FN static inline
#define FMK_LOG_FAR static
#endif
#include "fmk_logtab.h"

// the table-free branch: x in [1 - 2^-4, 1 + 0x1.09p-4)
FMK_LOG_FN double fmk_log_near1(double x)
{
    const double r = x - 1.0, r2 = r * r, r3 = r * r2;
    double q = fma(r3, -0x1.5521375d145cdp-4, fma(r2, 0x1.78182f7afd085p-4, fma(r, -0x1.999eb43b068ffp-4, 0x1.c7184282ad6cap-4)));
    q = fma(r3, q, fma(r2, -0x1.fffffa4423d65p-4, fma(r, 0x1.24924a344de3p-3, -0x1.55555556745a7p-3)));
    q = fma(r3, q, fma(r2, 0x1.999999995dd0cp-3, fma(r, -0x1.ffffffffffdcbp-3, 0x1.5555555555577p-2)));
    double w = r * 0x1p27;
    const double rhi = r + w - w, rlo = r - rhi;                     // r = rhi + rlo, rhi on 26 bits: rhi * rhi is exact
    w = rhi * rhi * -0.5;
    const double hi = r + w;
    double lo = r - hi + w;
    lo = fma(-0.5 * rlo, rhi + r, lo);
    return fma(r3, q, lo) + hi;
}

// every other argument (not inlined on the device: ~60 instructions, rarely run)
FMK_LOG_FAR double fmk_log_table(double x)
{
    union { double d; uint64_t u; } v;
    v.d = x;
    uint64_t ix = v.u;
    const unsigned top = (unsigned)(ix >> 48);
    if (top - 0x0010u >= 0x7ff0u - 0x0010u) {
        if (ix * 2 == 0) return -INFINITY;                           // +-0 (glibc: divide-by-zero)
        if (ix == 0x7ff0000000000000ull) return x;                   // +inf
        if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) return NAN;   // negative, NaN (glibc: invalid)
        v.d = x * 0x1p52;                                            // subnormal: normalise
        ix = v.u - (52ull << 52);
    }
    // x = 2^k z, z in [OFF, 2 OFF), OFF = 0x3fe6000000000000; subinterval i of 128
    const uint64_t tmp = ix - 0x3fe6000000000000ull;
    const int i = (int)((tmp >> (52 - 7)) & 127);
    const int k = (int)((int64_t)tmp >> 52);
    v.u = ix - (tmp & (0xfffull << 52));
    const double z = v.d, invc = FMK_LOG_T[2 * i], logc = FMK_LOG_T[2 * i + 1], kd = (double)k;
    const double r = fma(z, invc, -1.0);
    const double w = fma(kd, FMK_LOG_LN2HI, logc);
    const double hi = w + r;
    const double lo = fma(kd, FMK_LOG_LN2LO, w - hi + r);
    const double r2 = r * r;
    const double t1 = fma(r, FMK_LOG_A[2], FMK_LOG_A[1]);
    const double t2 = fma(r, FMK_LOG_A[4], FMK_LOG_A[3]);
    const double lo2 = fma(r2, FMK_LOG_A[0], lo);
    return fma(r * r2, fma(t2, r2, t1), lo2) + hi;
}

// the host's log(x)
FMK_LOG_FN double fmk_log_host(double x)
{
    if (x >= 0.9375 && x < 0x1.109p+0) return x == 1.0 ? 0.0 : fmk_log_near1(x);
    return fmk_log_table(x);
}

#ifdef __HIPCC__
// log(p / pm) for tick returns (comp_lagged_returns feature/core/utils.py:45-51, _cusum_bar_indexer bar/logic.py:196-200): the quotient as
// the reference rounds it, then the host's logarithm of it
__device__ __forceinline__ double fmk_log_ratio(double p, double pm) { return fmk_log_host(p / pm); }
#endif
