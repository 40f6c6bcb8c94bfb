// fmk_f32tie.h -- float32 outputs that are a float64 sum rounded once (base.py: `volume`, the order-flow volumes /
// dollars / spreads).  The kernels add in (lane, tree) order, the reference in tick order: the two float64 sums differ
// by at most ~len * 2^-53 * sum|terms|, which changes the float32 result only when the sum sits that close to a
// float32 rounding boundary.  Kernels test their per-bar sums with fmk_near_f32_tie and put the few bars that are
// too close on a redo list (list[0]: count, list[32...]: bar numbers), which a second kernel walks in tick order.
// Bound convention: callers pass eps * |sum| with eps = 2 * len * 2^-52; |sum| equals sum|terms| for the one-signed
// terms of the domain (amounts >= 0, prices >= 0).
#pragma once
#include "fmk_common.h"

// Is the float64 value s so close to a float32 rounding boundary that a perturbation of `bound` could change
// (float)s?  (distance of s to the midpoint between the two neighbouring float32 values)
__device__ __forceinline__ bool fmk_near_f32_tie(double s, double bound)
{
    const float f = (float)s;
    const double af = fabs((double)f);
    if (!(af > 1e-30) || isinf(f)) return false;                 // 0, denormal range, inf, NaN: nothing to flip
    const int ex = ilogb(af);
    const double up = ldexp(1.0, ex - 23);                       // float32 spacing above |f|
    const double down = af == ldexp(1.0, ex) ? 0.5 * up : up;    // ... and below (a power of two sits on a binade edge)
    const double half = 0.5 * (fabs(s) >= af ? up : down);
    // (written so that an UNDEFINED bound -- a NaN among the terms it was computed from, e.g. a NaN amount elsewhere in the bar while this
    //  extremum was reached before it -- says "near": the bar then takes the tick-order redo.  `<= bound` said "not near" and the bar kept
    //  the parallel order's last bit: tools/fuzz_longbars.py 150 361, case 70, cum_dollars_max 2.7e-12 below a boundary)
    return !(half - fabs(fabs(s) - af) > bound);
}

__device__ __forceinline__ double fmk_f32tie_eps(int64_t len) { return 4.6e-16 * (double)(len + 1); }   // > 2 len 2^-52
