// fmk_footprint.h -- device helpers shared by the footprint kernels (fmk_footprint.hip, fmk_barflow.hip):
// level rounding, NumPy's pairwise float32 summation, the exactness certificate of the integer-unit path and
// comp_footprint_features on the per-wave LDS histogram (finmlkit/bar/base.py:615-850).
#ifndef FMK_FOOTPRINT_H
#define FMK_FOOTPRINT_H
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "fmk_common.h"
#include "fmk_pairwise.h"
#include "fmk_dpp.h"

struct FpOut {
    int32_t *price_levels;
    float *buy_volumes, *sell_volumes;
    int32_t *buy_ticks, *sell_ticks;
    uint8_t *buy_imbalances, *sell_imbalances;
    uint16_t *buy_imbalances_sum, *sell_imbalances_sum;
    int32_t *cot_price_levels;
    int16_t *imb_max_run_signed;
    double *vp_skew, *vp_gini;
};
static_assert(sizeof(FpOut) == sizeof(fmk_footprint_out), "ABI struct mismatch");

#define FP_MAX_LEVELS 2048                 // widest bar whose histogram lives in LDS with the 24 B / level layout
#define FP_MAX_LEVELS_LDS 4096             // ... with the 16 B / level layout of the classes from 512 levels (one wave per CU there)
#define FP_MAX_LEVELS_GLOBAL (1 << 24)     // wider bars: histogram in global scratch (one wave per bar)
#define FP_Q_UNKNOWN 0x7FFFFFFF

// int(round(x)) with Python's round-half-even == rint() in the default rounding mode
__device__ __forceinline__ int64_t fp_level(double price, double tick) { return (int64_t)rint(price / tick); }
// Same value with one multiply instead of a float64 division on the per-tick path: price*(1/tick) and
// price/tick differ by <= 3.3e-16 relative, so rint() can only disagree when the quotient is that close
// to a half-integer -- in that (rare, divergent) case the exact division decides.
__device__ __forceinline__ int64_t fp_level(double price, double tick, double inv_tick)
{
    const double q = price * inv_tick;
    const double r = rint(q);
    if (0.5 - fabs(q - r) <= fabs(q) * 1e-15) return (int64_t)rint(price / tick);
    return (int64_t)r;
}

// 32-bit flavour for the per-tick path (price levels are int32 in the output arrays, base.py:675): a level beyond
// +-2^31 saturates and is then reported as out of range like any other tick outside [low, high]
__device__ __forceinline__ int fp_level32(double price, double tick, double inv_tick)
{
    const double q = price * inv_tick;
    double r = rint(q);
    if (0.5 - fabs(q - r) <= fabs(q) * 1e-15) r = rint(price / tick);
    return (int)r;
}

// NumPy pairwise float32 sum over an LDS array (fmk_pairwise.h holds the wave-level algorithm)
__device__ __forceinline__ float fp_pairwise_f32(const float *a, int n, int lane, int *stk)
{
    return fmk_pairwise_f32([a](int i) { return a[i]; }, n, lane, stk);
}

// ---------------------------------------------------------------------------------------
// histogram accumulation
// ---------------------------------------------------------------------------------------
// lowest set bit of |a| as a power-of-two exponent (INT_MAX for 0, INT_MIN for inf/NaN)
__device__ __forceinline__ int fp_lowbit_exp(float a)
{
    const uint32_t u = __float_as_uint(a) & 0x7FFFFFFFu;
    if (u == 0) return 0x7FFFFFFF;
    const int ex = (int)(u >> 23);
    const uint32_t mant = u & 0x7FFFFFu;
    if (ex == 255) return (int)0x80000000;
    if (ex == 0) return -149 + __builtin_ctz(mant);
    return ex - 150 + __builtin_ctz(mant | 0x800000u);
}
__device__ __forceinline__ int fp_lowbit_exp(double a)
{
    const uint64_t u = (uint64_t)__double_as_longlong(a) & 0x7FFFFFFFFFFFFFFFull;
    if (u == 0) return 0x7FFFFFFF;
    const int ex = (int)(u >> 52);
    const uint64_t mant = u & 0xFFFFFFFFFFFFFull;
    if (ex == 2047) return (int)0x80000000;
    if (ex == 0) return -1074 + __builtin_ctzll(mant);
    return ex - 1075 + __builtin_ctzll(mant | (1ull << 52));
}

struct FpStats {
    int lbmin;       // min lowest-set-bit exponent over the accumulated amounts (FP_Q_UNKNOWN if all zero)
    double atot;     // sum of |amount|
    bool units_ok;   // exact path only: every amount was a non-negative multiple of 2^q below 2^31 units
    bool bad;        // a tick fell outside the level range (base.py:719)
};

// exact (integer-unit) sweep: st.atot counts UNITS of 2^q; every amount was a whole non-negative number of units and
// every partial sum stays below 2^24 units, so each float32 add of the reference is exact whatever its order
__device__ __forceinline__ bool fp_certified_units(const FpStats &st) { return st.units_ok && st.atot < 16777216.0; }

// every float32 add of a bar with these statistics is exact at quantum 2^q
__device__ __forceinline__ bool fp_certified(const FpStats &st, int q)
{
    if (st.lbmin == FP_Q_UNKNOWN) return true;                 // only zeros
    return st.units_ok && q <= st.lbmin && q >= -149 && q <= 100 && st.atot < ldexp(1.0, 24 + q);
}

// The same per (level, side) KEY (round 3): the reference's float32 sums are per key, so what must stay below 2^24 units is every
// key's total, not the bar's (a bar of more than 8192 ticks of ~2^11 units each fails the bar-level test whatever its levels look
// like).  Non-negative whole units, the bar's total below 2^32 (no 32-bit counter has wrapped), every key below 2^24: every partial
// sum of every key is exactly representable.  Wave-level: all lanes call, the result is uniform.
__device__ __forceinline__ bool fp_certified_units_per_key(const FpStats &st, const unsigned *units, int nkeys, int lane)
{
    if (!(st.units_ok && st.atot < 4294967296.0)) return false;
    unsigned m = 0;
    for (int k = lane; k < nkeys; k += 64) m = units[k] > m ? units[k] : m;
    return (unsigned)fmk_dpp_reduce((int)(m >> 1), 0, FmkOpMax()) < (16777216u >> 1);     // (m >> 1: compared as signed ints)
}
// ... and after a tick-ordered sweep, whose vol[] holds the (rounded) float32 key sums: rounding is monotone and 2^(24+q) is a
// float32, so a rounded sum below it means a true sum below it.  All terms are non-negative multiples of 2^q (lbmin, units_ok).
__device__ __forceinline__ bool fp_certified_per_key(const FpStats &st, int q, const float *vol, int nkeys, int lane)
{
    if (st.lbmin == FP_Q_UNKNOWN) return true;
    if (!(st.units_ok && q <= st.lbmin && q >= -149 && q <= 100 && st.atot < ldexp(1.0, 32 + q))) return false;
    float m = 0.f;
    bool neg = false;
    for (int k = lane; k < nkeys; k += 64) { const float v = vol[k]; neg |= !(v >= 0.f); m = v > m ? v : m; }
    const double mm = fmk_dpp_reduce((double)m, 0.0, FmkOpMax());
    return __ballot(neg) == 0 && mm < ldexp(1.0, 24 + q);
}

// Level rows + comp_footprint_features (base.py:755-850) of ONE bar from the wave's LDS histogram:
//   vol[2L] float32 (buy = 2l, sell = 2l+1), cnt[2L], aux[2*lmax] scratch, stk[64] ints.  `base` = CSR row offset.
// fast_sum (histograms in LDS with lmax >= 512): the two np.sum over the levels by the parallel tree routine (fmk_np_sum ->
// fmk_pairwise_par), its tables in the idle second half of aux.  The node-by-node walk of fmk_pairwise costs ~50 000 cycles per
// sum over 1 500 levels -- bars that cover many levels in few ticks (a fine price_tick_size on a fast market) spent most of their
// time there (tools/widebench.py: 19.6 ms per 2e8 ticks at 1 500 levels per bar against 0.9 ms at 40).
__device__ __forceinline__ void fp_emit_bar(const FpOut &o, int64_t b, int64_t base, int L, int64_t low, int lmax,
                                            double imb_mult, int lane, float *vol, int *cnt, float *aux, int *stk,
                                            bool fast_sum = false)
{
    // aux == nullptr (the wave kernel's classes of 512 levels and more): no copy of the level totals -- buy + sell is formed again
    // where it is needed -- and the tree routine's tables in stk (FMK_PW_PAR_STK ints there): 16 instead of 24 B of LDS per level
    int *pstk = aux ? (int *)(aux + lmax) : stk;
    // ---- pass A: write the level rows, total[l] = buy + sell (float32), argmax, vwap numerator
    float *tot = aux;
    float best = -INFINITY;
    int best_i = 0x7FFFFFFF;
    double num = 0.0;
    for (int l = lane; l < L; l += 64) {
        const float bv = vol[2 * l], sv = vol[2 * l + 1];
        const int bc = cnt[2 * l], sc = cnt[2 * l + 1];
        o.price_levels[base + l] = (int32_t)(low + l);
        o.buy_volumes[base + l] = bv;
        o.sell_volumes[base + l] = sv;
        o.buy_ticks[base + l] = bc;
        o.sell_ticks[base + l] = sc;
        const float t = bv + sv;                                     // base.py:822
        if (aux) tot[l] = t;
        // first argmax within my lanes; np.argmax treats a NaN as the maximum: the FIRST NaN wins and is never replaced
        if (t > best || (t != t && best == best)) { best = t; best_i = l; }
        num += (double)(low + l) * (double)t;
    }
    // first argmax across lanes (ties -> lowest index)
    const bool narrow = L <= 64;               // (wave-uniform) one level per lane: the common case -- ~40 levels per one-minute bar
    if (narrow) {
        // ONE max-reduction of a packed key on the DPP path (round 4): the value as an order-preserving 32-bit pattern (NaN above
        // everything: np.argmax keeps the first NaN), then the lowest index -- instead of six butterflies of two ds_bpermute each
        long long key = -1;
        if (best_i != 0x7FFFFFFF) {
            const float tz = best + 0.0f;                                 // (-0.0 and +0.0 are equal to np.argmax)
            unsigned u = __float_as_uint(tz);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            if (best != best) u = 0xFFFFFFFFu;
            key = (long long)(((unsigned long long)u << 31) | (unsigned long long)(0x7FFFFFFF - best_i));
        }
        key = (long long)fmk_dpp_reduce((int64_t)key, (int64_t)-1, FmkOpMax());
        best_i = key < 0 ? 0 : 0x7FFFFFFF - (int)(key & 0x7FFFFFFF);      // empty guard: np.argmax -> 0
    } else {
#pragma unroll
    for (int x = 32; x > 0; x >>= 1) {
        float ob = __shfl_xor(best, x, 64);
        int oi = __shfl_xor(best_i, x, 64);
        const bool on = ob != ob, bn = best != best;
        const bool take = on ? (!bn || oi < best_i) : (!bn && (ob > best || (ob == best && oi < best_i)));
        if (take) { best = ob; best_i = oi; }
    }
    if (best_i == 0x7FFFFFFF) best_i = 0;      // empty guard: np.argmax -> 0
    }
    num = fmk_dpp_reduce(num, 0.0, FmkOpAdd());
    __builtin_amdgcn_wave_barrier();
    const float total = fast_sum ? (aux ? fmk_np_sum([tot](int i) { return tot[i]; }, L, lane, pstk)
                                        : fmk_np_sum([vol](int i) { return vol[2 * i] + vol[2 * i + 1]; }, L, lane, pstk))
                                 : fp_pairwise_f32(tot, L, lane, stk);          // total_volumes.sum()
    const bool stats = total > 0.f && L > 0;                         // base.py:836
    const double vwap = stats ? num / (double)total : 0.0;

    // ---- pass B: imbalance flags, run signs, skew, q^2
    int *sign = cnt;                           // cnt area is free now: sign[0..L), q2 at cnt + lmax
    float *q2 = (float *)(cnt + lmax);
    unsigned bsum = 0, ssum = 0;
    double skew = 0.0;
    unsigned long long mask_p = 0, mask_n = 0;                       // narrow bars: the levels whose run sign is +1 / -1
    for (int l0 = 0; l0 < L; l0 += 64) {
        const int l = l0 + lane;
        bool bi = false, si = false;
        if (l < L) {
            const float bv = vol[2 * l], sv = vol[2 * l + 1];
            if (l < L - 1) si = (double)sv > (double)vol[2 * (l + 1)] * imb_mult;         // base.py:797
            if (l >= 1) bi = (double)bv > (double)vol[2 * (l - 1) + 1] * imb_mult;        // base.py:798
            o.buy_imbalances[base + l] = bi;
            o.sell_imbalances[base + l] = si;
            sign[l] = bi ? 1 : (si ? -1 : 0);
            const float t = aux ? tot[l] : bv + sv;
            if (stats) {
                skew += ((double)(low + l) - vwap) * (double)t;
                const float q = t / total;
                q2[l] = q * q;
            }
        }
        const unsigned long long bb = __ballot(bi), sb = __ballot(si);
        bsum += __popcll(bb);
        ssum += __popcll(sb);
        mask_p = bb;                                                 // (meaningful when L <= 64: one iteration)
        mask_n = sb & ~bb;                                           // buy wins where both are set (base.py:807)
    }
    skew = fmk_dpp_reduce(skew, 0.0, FmkOpAdd());
    __builtin_amdgcn_wave_barrier();
    double gini = 0.0;
    if (stats) gini = (double)(1.0f - (fast_sum ? fmk_np_sum([q2](int i) { return q2[i]; }, L, lane, pstk)
                                               : fp_pairwise_f32(q2, L, lane, stk)));   // base.py:847-848 (float32)
    // ---- longest signed run (base.py:801-819).  The reference scans the levels once; here every lane scans a
    //      contiguous segment (prefix run, first-longest run inside, run state at its end) and the 64 summaries
    //      are folded in segment order -- same result, incl. "the FIRST run of maximal length wins".
    int max_run = 0, max_sign = 0;
    if (narrow) {
        // Narrow bars (round 4): the runs of equal sign are runs of set bits in two 64-bit masks.  m &= m >> 1 leaves, after k steps,
        // the STARTS of the runs of length > k: the last non-empty mask names the longest runs, its lowest bit the first of them.
        // The reference keeps the run that REACHES the maximal length first (run > max_run, strictly): start + length - 1, i.e. --
        // at equal length -- the lower start.  A dozen scalar instructions instead of a 16-step loop on every lane and a fold.
        int kp = 0, kn = 0, sp = 0, sn = 0;
        for (unsigned long long m = mask_p; m; m &= m >> 1) { ++kp; sp = (int)__builtin_ctzll(m); }
        for (unsigned long long m = mask_n; m; m &= m >> 1) { ++kn; sn = (int)__builtin_ctzll(m); }
        if (kp > kn || (kp == kn && kp > 0 && sp < sn)) { max_run = kp; max_sign = 1; }
        else if (kn > 0) { max_run = kn; max_sign = -1; }
    } else {
    int p_len = 0, p_sign = 0, s_len = 0, s_sign = 0, b_len = 0, b_sign = 0, n_seg = 0;
    {
        int seg = (L + 63) / 64;
        if (seg < 16) seg = 16;                                        // narrow bars: few segments, short fold
        const int l0 = lane * seg;
        const int l1 = l0 + seg < L ? l0 + seg : L;
        n_seg = l1 > l0 ? l1 - l0 : 0;
        int run = 0, rs = 0;
        bool in_prefix = true;
        for (int l = l0; l < l1; ++l) {
            const int sg = sign[l];
            if (sg != 0 && sg == rs) run += 1;
            else if (sg != 0) { run = 1; rs = sg; }
            else { run = 0; rs = 0; }
            if (in_prefix) {
                if (l == l0) { if (sg == 0) in_prefix = false; else { p_sign = sg; p_len = 1; } }
                else if (sg == p_sign) p_len += 1;
                else in_prefix = false;
            }
            if (run > b_len) { b_len = run; b_sign = rs; }
        }
        s_len = run; s_sign = rs;
    }
    {
        int run = 0, run_sign = 0;
        for (int k = 0; k < 64; ++k) {
            const int n_k = __builtin_amdgcn_readlane(n_seg, k);
            if (n_k == 0) break;                                   // segments are filled from lane 0 upwards
            const int pl = __builtin_amdgcn_readlane(p_len, k), ps = __builtin_amdgcn_readlane(p_sign, k);
            if (pl == n_k) {                                       // the whole segment is one signed run
                if (ps == run_sign) run += pl; else { run = pl; run_sign = ps; }
                if (run > max_run) { max_run = run; max_sign = run_sign; }
            } else {
                if (pl > 0) {
                    const int cand = ps == run_sign ? run + pl : pl;
                    if (cand > max_run) { max_run = cand; max_sign = ps; }
                }
                const int bl = __builtin_amdgcn_readlane(b_len, k), bs = __builtin_amdgcn_readlane(b_sign, k);
                if (bl > max_run) { max_run = bl; max_sign = bs; }
                run = __builtin_amdgcn_readlane(s_len, k);
                run_sign = __builtin_amdgcn_readlane(s_sign, k);
            }
        }
    }
    }
    if (lane == 0) {
        o.buy_imbalances_sum[b] = (uint16_t)bsum;
        o.sell_imbalances_sum[b] = (uint16_t)ssum;
        o.cot_price_levels[b] = (int32_t)(low + best_i);
        o.imb_max_run_signed[b] = (int16_t)(max_run * max_sign);
        o.vp_skew[b] = stats ? skew / (double)total : 0.0;
        o.vp_gini[b] = gini;
    }
    __builtin_amdgcn_wave_barrier();
}

#endif
