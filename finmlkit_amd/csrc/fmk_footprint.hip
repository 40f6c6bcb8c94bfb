// fmk_footprint.hip -- comp_bar_footprints + comp_footprint_features
// (finmlkit/bar/base.py:615-850) on gfx950, CSR output.
//
// The reference appends one small array per bar to Python lists (serial, base.py:615 "not
// parallelizable").  Here the ragged result is CSR: level_offsets[B+1] (exclusive scan of the
// per-bar level counts, phase 1) + flat per-level arrays, filled by one wave per bar (phase 2):
//
//   * the bar's dense level histogram lives in the wave's slice of LDS:
//       vol[2L] f32 (buy/sell interleaved), cnt[2L] i32, aux[2L]              = 24 B / level
//   * ticks stream in 64-tick chunks (price 512 B + amount 256 B + side 64 B per load, coalesced,
//     software-pipelined; 13 B/tick).  level = int(round(price/tick)) - int(round(low/tick)) with
//     round-half-even like the reference (base.py:688-703); the per-tick quotient is one multiply by
//     1/tick, with the exact division as a guarded fallback when the product is within 1e-15 (rel.)
//     of a half-integer, so the rounded level is always the reference's.
//   * the reference rounds the level volume to float32 on EVERY add, in tick order (base.py:713-717),
//     so the sum is order-sensitive in general.  Two ways to get its bits:
//       - exact path: if all amounts of the bar are non-negative multiples of 2^q and the bar's total
//         is < 2^(24+q), every partial sum is exactly representable and the order cannot matter.  The
//         amounts are then accumulated as integer units with LDS integer atomics (ds_add_u32; the
//         float atomic ds_add_f32 measured ~25x slower) and converted back at the end.  The certificate
//         is evaluated from the data of the bar itself; q is remembered per wave.
//       - ordered path (any input): per chunk the lanes are grouped by (level, side) key with a
//         ballot loop; inside a group the running float32 value is passed from the lane of rank t-1 to
//         rank t by a lane gather (tick order), the first lane starts from LDS and the last one stores.
//         Distinct keys proceed in parallel, no atomics.
//   * comp_footprint_features runs on the LDS histogram: diagonal imbalance flags (float32 product
//     like NumPy: float32 array * Python float), longest signed run, first argmax (COT), and the
//     float32 sums total / gini with NumPy's pairwise summation order reproduced exactly
//     (8-accumulator leaves of <=128 elements evaluated by 8 lanes, recursive halving above);
//     vp_skew is mathematically 0 (rounding noise in the reference) and is evaluated in float64.
//
// Bars are binned by level count so that the common narrow bars run at high occupancy:
// L<=128 (3 KB LDS/wave), <=512 (12 KB), <=2048 (48 KB, one wave per workgroup).
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "fmk_common.h"
#include "fmk_dpp.h"
#include "fmk_scan.h"

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

#define FP_MAX_LEVELS 2048
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

// ---------------------------------------------------------------------------------------
// phase 1: level counts per bar -> exclusive scan
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fp_level_counts(const double *__restrict__ lows,
                                                         const double *__restrict__ highs, int64_t nb, double tick,
                                                         int64_t *__restrict__ counts, unsigned long long *max_levels)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t L = 0;
    if (i < nb) {
        L = fp_level(highs[i], tick) - fp_level(lows[i], tick) + 1;   // base.py:688-690
        if (L < 0) L = 0;
        counts[i] = L;
    }
    int64_t m = fmk_wave_max(L);
    if (fmk_lane() == 0 && m > 0) atomicMax(max_levels, (unsigned long long)m);
}

extern "C" int fmk_comp_bar_footprints_size_dev(fmk_ctx *ctx, const double *d_bar_lows, const double *d_bar_highs,
                                                int64_t n_bars, double price_tick_size, int64_t *d_level_offsets,
                                                int64_t *total_levels, int64_t *max_levels)
{
    if (n_bars < 1) return fmk_set_error(ctx, FMK_E_ARG, "Bar close indices must contain at least two elements.");
    if (!(price_tick_size > 0)) return fmk_set_error(ctx, FMK_E_ARG, "price_tick_size must be > 0");
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    unsigned long long *d_max = (unsigned long long *)ctx->d_mail;
    FMK_HIP(ctx, hipMemsetAsync(d_max, 0, 8, ctx->stream));
    k_fp_level_counts<<<(unsigned)fmk_ceil_div(n_bars, 256), 256, 0, ctx->stream>>>(
        d_bar_lows, d_bar_highs, n_bars, price_tick_size, d_level_offsets, d_max);
    FMK_LAUNCH_CHECK(ctx);
    FMK_TRY(fmk_exclusive_scan_i64(ctx, d_level_offsets, d_level_offsets, n_bars, true));
    FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[0], d_level_offsets + n_bars, 8, hipMemcpyDeviceToHost, ctx->stream));
    FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[1], d_max, 8, hipMemcpyDeviceToHost, ctx->stream));
    FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *total_levels = ctx->h_mail[0];
    *max_levels = ctx->h_mail[1];
    return FMK_OK;
}

// ---------------------------------------------------------------------------------------
// NumPy pairwise float32 sum over an LDS array, evaluated by the whole wave (uniform result)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float fp_pw_leaf(const float *a, int n, int lane)
{
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    const int nm = n - (n & 7);
    float r = 0.f;
    if (lane < 8) {
        r = a[lane];
        for (int i = 8 + lane; i < nm; i += 8) r += a[i];
    }
    float t = r + __shfl_down(r, 1, 64);     // lanes 0,2,4,6: r0+r1, r2+r3, r4+r5, r6+r7
    float u = t + __shfl_down(t, 2, 64);     // lanes 0,4
    float res = __shfl(u, 0, 64) + __shfl(u, 4, 64);
    for (int i = nm; i < n; ++i) res += a[i];
    return res;
}

// stk: per-wave LDS scratch of 4*16 ints (explicit recursion stack: off, len, phase, left)
__device__ __forceinline__ float fp_pairwise_f32(const float *a, int n, int lane, int *stk)
{
    if (n <= 128) return fp_pw_leaf(a, n, lane);
    int *s_off = stk, *s_len = stk + 16, *s_ph = stk + 32;
    float *s_left = (float *)(stk + 48);
    int sp = 1;
    if (lane == 0) { s_off[0] = 0; s_len[0] = n; s_ph[0] = 0; }
    __builtin_amdgcn_wave_barrier();
    float ret = 0.f;
    bool have = false;
    while (sp > 0) {
        const int top = sp - 1;
        const int off = fmk_uniform(s_off[top]), len = fmk_uniform(s_len[top]), ph = fmk_uniform(s_ph[top]);
        int n2 = len / 2;
        n2 -= n2 % 8;
        if (!have) {
            if (len <= 128) { ret = fp_pw_leaf(a + off, len, lane); have = true; --sp; }
            else {
                if (lane == 0) { s_off[sp] = off; s_len[sp] = n2; s_ph[sp] = 0; }
                ++sp;
            }
        } else {
            if (ph == 0) {
                if (lane == 0) { s_left[top] = ret; s_ph[top] = 1; s_off[sp] = off + n2; s_len[sp] = len - n2; s_ph[sp] = 0; }
                ++sp;
                have = false;
            } else {
                ret = __builtin_bit_cast(float, fmk_uniform(__builtin_bit_cast(int, s_left[top]))) + ret;
                --sp;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    return ret;
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

// One pass over the bar's ticks [s+1, e].
//   EXACT  : vol[] is used as uint32 units of 2^q, updated with integer LDS atomics (order-free).
//   !EXACT : vol[] holds float32 running sums updated in tick order (group / rank / lane-gather chains).
template <bool AF64, bool EXACT>
__device__ __forceinline__ FpStats fp_accumulate(const double *__restrict__ price, const void *__restrict__ amount,
                                                 const int8_t *__restrict__ side, int64_t s, int64_t e, int64_t low,
                                                 int L, double tick, double inv_tick, int lane, float *vol, int *cnt,
                                                 int q)
{
    typedef typename std::conditional<AF64, double, float>::type AmtT;
    unsigned *units = (unsigned *)vol;
    int lbmin = FP_Q_UNKNOWN;
    double atot = 0.0;
    bool bad = false, units_ok = true;
    // software pipeline: the loads of chunk c+1 are in flight while chunk c is processed
    double p_n = 0.0;
    AmtT a_n = 0;
    int sd_n = 0;
    if (s + 1 + lane <= e) {
        p_n = price[s + 1 + lane];
        a_n = ((const AmtT *)amount)[s + 1 + lane];
        sd_n = side[s + 1 + lane];
    }
    for (int64_t j0 = s + 1; j0 <= e; j0 += 64) {
        const int64_t j = j0 + lane;
        const double p = p_n;
        const AmtT a = a_n;
        const int sd = sd_n;
        if (j + 64 <= e) {
            p_n = price[j + 64];
            a_n = ((const AmtT *)amount)[j + 64];
            sd_n = side[j + 64];
        }
        bool pending = false;
        int key = -1;
        if (j <= e) {
            const int64_t lvl = fp_level(p, tick, inv_tick) - low;    // base.py:700-707
            if (lvl < 0 || lvl >= L) bad = true;                      // base.py:719
            else if (sd == 1 || sd == -1) {
                pending = true;
                key = (int)lvl * 2 + (sd == 1 ? 0 : 1);
                const int lb = fp_lowbit_exp(a);
                lbmin = lb < lbmin ? lb : lbmin;
                atot += fabs((double)a);
            }
        }
        if constexpr (EXACT) {
            if (pending) {
                const double u = ldexp((double)a, -q);                // exact scaling
                const bool ok = u >= 0.0 && u < 2147483648.0 && u == rint(u);
                units_ok &= ok;
                if (ok) atomicAdd(&units[key], (unsigned)u);
                atomicAdd(&cnt[key], 1);
            }
            continue;
        }
        // ---- group the pending lanes by key: every lane learns the lane mask of its key
        uint64_t grp = 0;
        for (uint64_t rem = __ballot(pending); rem != 0;) {
            const int leader = __ffsll((unsigned long long)rem) - 1;
            const int k = __builtin_amdgcn_readlane(key, leader);
            const uint64_t m = __ballot(key == k);                    // non-pending lanes carry key -1
            if (key == k) grp = m;
            rem &= ~m;
        }
        // ---- float32 accumulation in tick (= lane) order inside every key group: the first lane of a
        //      group starts from the LDS value, the lane of rank t takes the running value of rank t-1
        //      by a lane gather, the last lane stores.
        const uint64_t below = grp & (((uint64_t)1 << lane) - 1);
        const int rank = __popcll(below);
        const int gsize = __popcll(grp);
        const int prev_lane = rank > 0 ? 63 - __clzll((unsigned long long)below) : lane;
        float acc = 0.f;
        if (pending && rank == 0) {
            acc = vol[key];
            if constexpr (AF64) acc = (float)((double)acc + a);       // f32 element += f64 amount
            else acc = acc + a;
            cnt[key] += gsize;                                        // one writer per key: no atomic
        }
        const int rounds = fmk_dpp_reduce(gsize, 0, FmkOpMax());
        for (int t = 1; t < rounds; ++t) {
            const float v = __shfl(acc, prev_lane, 64);
            if (pending && rank == t) {
                if constexpr (AF64) acc = (float)((double)v + a);
                else acc = v + a;
            }
        }
        if (pending && rank == gsize - 1) vol[key] = acc;
        __builtin_amdgcn_wave_barrier();
    }
    FpStats st;
    st.lbmin = fmk_dpp_reduce(lbmin, FP_Q_UNKNOWN, FmkOpMin());
    st.atot = fmk_dpp_reduce(atot, 0.0, FmkOpAdd());
    st.units_ok = __ballot(!units_ok) == 0;
    st.bad = __ballot(bad) != 0;
    __builtin_amdgcn_wave_barrier();
    return st;
}

// every float32 add of a bar with these statistics is exact at quantum 2^q
__device__ __forceinline__ bool fp_certified(const FpStats &st, int q)
{
    if (st.lbmin == FP_Q_UNKNOWN) return true;                 // only zeros
    return st.units_ok && q <= st.lbmin && q >= -149 && q <= 100 && st.atot < ldexp(1.0, 24 + q);
}

// ---------------------------------------------------------------------------------------
// phase 2: one wave per bar
// ---------------------------------------------------------------------------------------
template <bool AF64>
__global__ __launch_bounds__(256) void k_bar_footprints(const double *__restrict__ price,
                                                        const void *__restrict__ amount,
                                                        const int8_t *__restrict__ side,
                                                        const int64_t *__restrict__ ci, int64_t nb, double tick,
                                                        const double *__restrict__ lows, float m32,
                                                        const int64_t *__restrict__ off, int lmin, int lmax,
                                                        FpOut o, unsigned long long *n_bad, int force_ordered)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = fmk_lane();
    const int wib = fmk_uniform((int)(threadIdx.x >> 6));
    const int wpb = blockDim.x >> 6;
    const size_t per_wave = (size_t)lmax * 24 + 256;
    unsigned char *mine = smem + (size_t)wib * per_wave;
    float *vol = (float *)mine;                                   // [2*lmax]  buy = 2l, sell = 2l+1
    int *cnt = (int *)(mine + (size_t)lmax * 8);                  // [2*lmax]
    float *aux = (float *)(mine + (size_t)lmax * 16);             // [2*lmax]  tot[], later q2[]
    int *stk = (int *)(mine + (size_t)lmax * 24);                 // 64 ints
    const int64_t wave0 = (int64_t)blockIdx.x * wpb + wib;
    const int64_t nwaves = (int64_t)gridDim.x * wpb;
    const double inv_tick = 1.0 / tick;
    int wq = FP_Q_UNKNOWN;        // quantum exponent the previous bar of this wave certified with
    for (int64_t b = wave0; b < nb; b += nwaves) {
        const int64_t base = fmk_uniform(off[b]);
        const int L = (int)fmk_uniform(off[b + 1] - base);
        if (L <= lmin || L > lmax) continue;          // handled by another launch (or L == 0)
        const int64_t s = fmk_uniform(ci[b]);
        const int64_t e = fmk_uniform(ci[b + 1]);
        const int64_t low = fp_level(lows[b], tick);
        for (int k = lane; k < 2 * L; k += 64) { vol[k] = 0.f; cnt[k] = 0; }
        __builtin_amdgcn_wave_barrier();
        // Exact path first (with the quantum that worked for the previous bar); when its certificate
        // fails the bar's own statistics give the right quantum for one retry, else the ordered path runs.
        FpStats st;
        bool done = false;
        if (!force_ordered && wq != FP_Q_UNKNOWN) {
            st = fp_accumulate<AF64, true>(price, amount, side, s, e, low, L, tick, inv_tick, lane, vol, cnt, wq);
            done = fp_certified(st, wq);
            if (!done) {
                for (int k = lane; k < 2 * L; k += 64) { vol[k] = 0.f; cnt[k] = 0; }
                __builtin_amdgcn_wave_barrier();
                const int q2 = st.lbmin;
                FpStats probe = st;
                probe.units_ok = true;
                if (q2 != FP_Q_UNKNOWN && q2 != (int)0x80000000 && fp_certified(probe, q2)) {
                    st = fp_accumulate<AF64, true>(price, amount, side, s, e, low, L, tick, inv_tick, lane, vol, cnt, q2);
                    done = fp_certified(st, q2);
                    if (done) wq = q2;
                    else {
                        for (int k = lane; k < 2 * L; k += 64) { vol[k] = 0.f; cnt[k] = 0; }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            if (done) {       // units -> float32 (exact)
                unsigned *units = (unsigned *)vol;
                const int qq = wq;
                for (int k = lane; k < 2 * L; k += 64) vol[k] = ldexpf((float)units[k], qq);
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (!done) {
            st = fp_accumulate<AF64, false>(price, amount, side, s, e, low, L, tick, inv_tick, lane, vol, cnt, 0);
            // remember a usable quantum for the next bar (if this bar would have certified)
            FpStats probe = st;
            probe.units_ok = true;
            const bool usable = st.lbmin != FP_Q_UNKNOWN && st.lbmin != (int)0x80000000 && fp_certified(probe, st.lbmin);
            wq = usable ? st.lbmin : FP_Q_UNKNOWN;
        }
        if (st.bad && lane == 0 && n_bad) atomicAdd(n_bad, 1ULL);
        __builtin_amdgcn_wave_barrier();

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
            tot[l] = t;
            if (t > best) { best = t; best_i = l; }                      // first argmax within my lanes
            num += (double)(low + l) * (double)t;
        }
        // first argmax across lanes (ties -> lowest index)
#pragma unroll
        for (int x = 32; x > 0; x >>= 1) {
            float ob = __shfl_xor(best, x, 64);
            int oi = __shfl_xor(best_i, x, 64);
            if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
        }
        if (best_i == 0x7FFFFFFF) best_i = 0;      // all-NaN / empty guard: np.argmax -> 0
        num = fmk_dpp_reduce(num, 0.0, FmkOpAdd());
        __builtin_amdgcn_wave_barrier();
        const float total = fp_pairwise_f32(tot, L, lane, stk);          // total_volumes.sum()
        const bool stats = total > 0.f && L > 0;                         // base.py:836
        const double vwap = stats ? num / (double)total : 0.0;

        // ---- pass B: imbalance flags, run signs, skew, q^2
        int *sign = cnt;                           // cnt area is free now: sign[0..L), q2 at cnt + lmax
        float *q2 = (float *)(cnt + lmax);
        unsigned bsum = 0, ssum = 0;
        double skew = 0.0;
        for (int l0 = 0; l0 < L; l0 += 64) {
            const int l = l0 + lane;
            bool bi = false, si = false;
            if (l < L) {
                const float bv = vol[2 * l], sv = vol[2 * l + 1];
                if (l < L - 1) si = sv > vol[2 * (l + 1)] * m32;          // base.py:797
                if (l >= 1) bi = bv > vol[2 * (l - 1) + 1] * m32;         // base.py:798
                o.buy_imbalances[base + l] = bi;
                o.sell_imbalances[base + l] = si;
                sign[l] = bi ? 1 : (si ? -1 : 0);
                const float t = tot[l];
                if (stats) {
                    skew += ((double)(low + l) - vwap) * (double)t;
                    const float q = t / total;
                    q2[l] = q * q;
                }
            }
            bsum += __popcll(__ballot(bi));
            ssum += __popcll(__ballot(si));
        }
        skew = fmk_dpp_reduce(skew, 0.0, FmkOpAdd());
        __builtin_amdgcn_wave_barrier();
        double gini = 0.0;
        if (stats) gini = (double)(1.0f - fp_pairwise_f32(q2, L, lane, stk));   // base.py:847-848 (float32)
        // ---- longest signed run (base.py:801-819), sequential over the levels
        if (lane == 0) {
            int max_run = 0, max_sign = 0, run = 0, run_sign = 0;
            for (int l = 0; l < L; ++l) {
                const int sg = sign[l];
                if (sg != 0 && sg == run_sign) run += 1;
                else if (sg != 0) { run = 1; run_sign = sg; }
                else { run = 0; run_sign = 0; }
                if (run > max_run) { max_run = run; max_sign = run_sign; }
            }
            o.buy_imbalances_sum[b] = (uint16_t)bsum;
            o.sell_imbalances_sum[b] = (uint16_t)ssum;
            o.cot_price_levels[b] = (int32_t)(low + best_i);
            o.imb_max_run_signed[b] = (int16_t)(max_run * max_sign);
            o.vp_skew[b] = stats ? skew / (double)total : 0.0;
            o.vp_gini[b] = gini;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <bool AF64>
static int fp_launch(fmk_ctx *ctx, const double *p, const void *a, const int8_t *sd, const int64_t *ci, int64_t nb,
                     double tick, const double *lows, float m32, const int64_t *off, int lmin, int lmax, int wpb,
                     const FpOut &o, unsigned long long *n_bad)
{
    static int force_ordered = -1;    // developer knob: FMK_FP_ORDERED=1 disables the exact (integer) path
    if (force_ordered < 0) { const char *v = getenv("FMK_FP_ORDERED"); force_ordered = v ? atoi(v) : 0; }
    const size_t smem = (size_t)wpb * ((size_t)lmax * 24 + 256);
    int64_t blocks = fmk_ceil_div(nb, wpb);
    const int64_t cap = (int64_t)ctx->n_cu * 64;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    k_bar_footprints<AF64><<<(unsigned)blocks, wpb * 64, smem, ctx->stream>>>(p, a, sd, ci, nb, tick, lows, m32, off,
                                                                            lmin, lmax, o, n_bad, force_ordered);
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}

extern "C" int fmk_comp_bar_footprints_fill_dev(fmk_ctx *ctx, const double *d_price, const void *d_amount,
                                                int amount_is_f64, int64_t n, const int64_t *d_close_idx,
                                                int64_t n_idx, const int8_t *d_side, double price_tick_size,
                                                const double *d_bar_lows, double imbalance_factor,
                                                const int64_t *d_level_offsets, int64_t max_levels,
                                                const fmk_footprint_out *d_out, int64_t *d_n_bad_level)
{
    if (n_idx < 2) return fmk_set_error(ctx, FMK_E_ARG, "Bar close indices must contain at least two elements.");
    if (n <= 0 || !d_side || !d_out) return fmk_set_error(ctx, FMK_E_ARG, "comp_bar_footprints: bad arguments");
    if (max_levels > FP_MAX_LEVELS)
        return fmk_set_error(ctx, FMK_E_CAPACITY,
                             "comp_bar_footprints: a bar spans %lld price levels; this build supports <= %d per bar",
                             (long long)max_levels, FP_MAX_LEVELS);
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t nb = n_idx - 1;
    FpOut o;
    memcpy(&o, d_out, sizeof(o));
    const float m32 = (float)imbalance_factor;   // float32 array * Python float -> float32 (NEP 50)
    unsigned long long *bad = (unsigned long long *)d_n_bad_level;
    static const int LMAX[3] = {128, 512, 2048};
    static const int WPB[3] = {4, 4, 1};
    int lmin = 0;
    for (int k = 0; k < 3; ++k) {
        if (k > 0 && max_levels <= LMAX[k - 1]) break;
        int rc = amount_is_f64
                     ? fp_launch<true>(ctx, d_price, d_amount, d_side, d_close_idx, nb, price_tick_size, d_bar_lows,
                                       m32, d_level_offsets, lmin, LMAX[k], WPB[k], o, bad)
                     : fp_launch<false>(ctx, d_price, d_amount, d_side, d_close_idx, nb, price_tick_size, d_bar_lows,
                                        m32, d_level_offsets, lmin, LMAX[k], WPB[k], o, bad);
        if (rc) return rc;
        lmin = LMAX[k];
    }
    return FMK_OK;
}
