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
//       - ordered path (any input).  float32 amounts (round 6, fp_accumulate_sorted): a segment of 256 ticks is sorted
//         by (key, tick) -- one lane-ordered LDS atomic per tick for its rank within its key, a scan for the keys'
//         slices, one LDS store per tick -- and one lane per key adds its slice in tick order.  float64 amounts /
//         histograms in global scratch: per chunk the lanes are grouped by (level, side) key; inside a group the
//         running float32 value passes from the lane of rank t-1 to rank t (tick order).
//   * comp_footprint_features runs on the LDS histogram: diagonal imbalance flags (float32 product
//     like NumPy: float32 array * Python float), longest signed run, first argmax (COT), and the
//     float32 sums total / gini with NumPy's pairwise summation order reproduced exactly
//     (8-accumulator leaves of <=128 elements evaluated by 8 lanes, recursive halving above);
//     vp_skew is mathematically 0 (rounding noise in the reference) and is evaluated in float64.
//
// Bars are binned by level count so that the common narrow bars run at high occupancy:
// L<=128 (3 KB LDS/wave), <=512 (12 KB), <=2048 (48 KB, one wave per workgroup); bars wider than that (a fine
// tick on a volatile hour: 10^4-10^5 levels) run the same code with the histogram in a per-wave slice of global
// scratch instead of LDS (generic pointers; slow but unlimited up to 2^24 levels).
#include "fmk_footprint.h"
#include "fmk_median.h"
#include "fmk_scan.h"

// cfg 4 at 26 B/tick (round 3): the wave that sweeps a bar for its footprint has every amount of the bar in hand, so the median
// trade size (base.py:401-404) is taken HERE instead of by a pass of its own over the amount column (k_bar_median_small, 4 B/tick,
// 1.35 ms per 1e9 ticks).  Holding the bar's 1 200 keys for a search after the sweep was built first and measured SLOWER (5.4 KB of
// LDS per wave or 21 more VGPRs: the sweep is latency-bound and lost its occupancy, 7.8 -> 8.6 ms, profiles/r03_cfg4.txt).  What the
// sweep does instead costs ~10 instructions per 64 ticks and 1 KB of LDS:
//   * the wave carries a BRACKET [blo, bhi] of order-preserving keys from its previous bar -- the keys at the ranks R below and R
//     above that bar's middle (R ~ 2.4 sqrt(ticks): the middle ranks of the next bar of a slowly changing size distribution land
//     inside with probability > 0.999);
//   * per chunk it counts the keys below the bracket and appends the keys inside it to a candidate list in LDS (<= 256);
//   * after the sweep, if the two middle ranks fall inside the candidates, they are selected there EXACTLY (fmk_median.h's
//     bisection + cross-lane sort on four registers per lane) and the bracket moves to the candidates' ranks -R / +R;
//   * otherwise (first bar of a wave, a jump in the distribution, more than 256 candidates) the bar's amounts are re-read and
//     searched by the generic selection (L2 hits: the sweep has just streamed them), which also re-seeds the bracket.  A heavy tie
//     at the median (a size that most trades share) gives the degenerate bracket blo == bhi, which needs no list at all.
// The result is np.median's bits in every case -- the bracket only decides how much work it takes.  Bars of more than
// FP_MED_MAX_TICKS ticks are flagged for the long-bar median kernels (fmk_median_launch), as after k_bar_median_small.
#define FP_MED_MAX_TICKS 2048
#define FP_MED_CAP 256
struct FpMed {
    uint32_t *cand;            // [FP_MED_CAP] in the wave's LDS slice
    uint32_t blo, bhi;         // bracket (keys), wave-uniform
    int have;                  // bracket valid
    int below, ncand;          // this sweep: keys < blo, keys in [blo, bhi]
    uint32_t kmin, kmax;       // per lane: smallest / largest key seen (NaN detection)
};

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
    // the atomic is attempted only when it would raise the current maximum: issued unconditionally it was one same-address
    // atomic per wave -- 7.8e5 of them at ~10 ns for the 5e7 one-second bars of a 1e9-tick tape: 8.9 ms for a 1.2 GB pass
    int64_t m = fmk_wave_max(L);
    if (fmk_lane() == 0 && m > 0 && (unsigned long long)m > __atomic_load_n(max_levels, __ATOMIC_RELAXED))
        atomicMax(max_levels, (unsigned long long)m);
}

extern "C" int fmk_comp_bar_footprints_size_dev(fmk_ctx *ctx, const double *d_bar_lows, const double *d_bar_highs,
                                                int64_t n_bars, double price_tick_size, int64_t *d_level_offsets,
                                                int64_t *total_levels, int64_t *max_levels)
{
    if (n_bars == 0) {   // zero bars (base.py:615-752 has no length check): one offset, no levels
        FMK_HIP(ctx, hipSetDevice(ctx->device));
        FMK_HIP(ctx, hipMemsetAsync(d_level_offsets, 0, 8, ctx->stream));
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (total_levels) *total_levels = 0;
        if (max_levels) *max_levels = 0;
        return FMK_OK;
    }
    if (n_bars < 0) return fmk_set_error(ctx, FMK_E_ARG, "negative dimensions are not allowed");
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
        // branch-light: levels are int32 in the output, so 32-bit level arithmetic; one unsigned compare for the range
        const bool in_bar = j <= e;
        const int lvl = fp_level32(p, tick, inv_tick) - (int)low;     // base.py:700-707
        const bool inside = (unsigned)lvl < (unsigned)L;
        bad |= in_bar && !inside;                                     // base.py:719
        const bool pending = in_bar && inside && (sd == 1 || sd == -1);
        const int key = pending ? lvl * 2 + (sd == 1 ? 0 : 1) : -1;
        if constexpr (!EXACT) {                                       // statistics that pick the quantum of later bars
            if (pending) {
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
                if (ok) { atomicAdd(&units[key], (unsigned)u); atot += u; }    // atot: units (fp_certified_units)
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

// Same sweep, straight-line (LDS histograms only; round 2).  The loop above costs 161 VALU + 130 SALU instructions per 64
// ticks (profiles/r01_flow_sq_counters.txt), most of it branches around `pending`, 64-bit index compares and, on the
// tick-ordered path, the ballot loop that groups the lanes by key (one iteration per distinct key: ~20 per chunk).
//   * no per-tick branch: one predicate per lane, the rare half-integer guard of the level rounding is a wave-uniform branch;
//   * EXACT: amount -> units with v_ldexp / v_cvt and ONE compare for "whole, non-negative, below 2^31";
//   * tick order WITHOUT grouping: a same-address LDS atomic with return hands out its old values in ascending lane order
//     (checked on the device before first use, fp_lds_atomics_in_lane_order; tools/ldsorder_probe.hip), so
//         rank = atomicAdd(&cnt[key], 1) - (cnt[key] before the chunk)
//     is the tick order of the lane inside its (level, side) group, and round t lets the lanes of rank t do a plain
//     read - add - write on vol[key]: distinct keys in a round, float32 adds of one key strictly in tick order.
template <bool AF64, bool EXACT>
__device__ __forceinline__ FpStats fp_accumulate_lean(const double *__restrict__ price, const void *__restrict__ amount,
                                                      const int8_t *__restrict__ side, int64_t s, int64_t e, int64_t low,
                                                      int L, double tick, double inv_tick, int lane, float *vol, int *cnt,
                                                      int q, FpMed *med = nullptr)
{
    typedef typename std::conditional<AF64, double, float>::type AmtT;
    unsigned *units = (unsigned *)vol;
    int lbmin = FP_Q_UNKNOWN;
    double atot = 0.0;
    bool bad = false, units_ok = true;
    const int ilow = (int)low;
    const bool med_on = !AF64 && med != nullptr && med->have;
    uint32_t m_lo = 0, m_hi = 0, m_kmin = 0xFFFFFFFFu, m_kmax = 0;
    int m_below = 0, m_n = 0;
    if (med_on) { m_lo = med->blo; m_hi = med->bhi; }
    const double *pp = price + (s + 1);
    const AmtT *ap = (const AmtT *)amount + (s + 1);
    const int8_t *sp = side + (s + 1);
    const int total = (int)(e - s);                                   // ticks of the bar (<= 2^31: one wave per bar)
    double p_n = 0.0;
    AmtT a_n = 0;
    int sd_n = 0;
    if (lane < total) { p_n = pp[lane]; a_n = ap[lane]; sd_n = sp[lane]; }
    for (int j0 = 0; j0 < total; j0 += 64) {
        const int j = j0 + lane;
        const double p = p_n;
        const AmtT a = a_n;
        const int sd = sd_n;
        if (j + 64 < total) { p_n = pp[j + 64]; a_n = ap[j + 64]; sd_n = sp[j + 64]; }
        const bool in_bar = j < total;
        if constexpr (!AF64) {
            if (med_on) {       // the median's bracket: every tick of the bar, signed or not, inside the level range or not
                const uint32_t k = MedKey<false>::tokey(__float_as_uint(a));
                m_kmin = (in_bar && k < m_kmin) ? k : m_kmin;
                m_kmax = (in_bar && k > m_kmax) ? k : m_kmax;
                m_below += __popcll(__ballot(in_bar && k < m_lo));
                const bool inb = in_bar && k >= m_lo && k <= m_hi;
                const uint64_t bm = __ballot(inb);
                const int pos = m_n + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0));
                if (inb && pos < FP_MED_CAP) med->cand[pos] = k;
                m_n += __popcll(bm);
            }
        }
        // level = int(round(price / tick)) - low (base.py:700-707): one multiply; the exact division decides the (rare) products
        // within 1e-15 of a half-integer -- as a wave-uniform branch
        const double qq = p * inv_tick;
        double r = rint(qq);
        if (__ballot(in_bar && 0.5 - fabs(qq - r) <= fabs(qq) * 1e-15) != 0) {
            if (0.5 - fabs(qq - r) <= fabs(qq) * 1e-15) r = rint(p / tick);
        }
        const int lvl = (int)r - ilow;
        const bool inside = (unsigned)lvl < (unsigned)L;
        bad |= in_bar && !inside;                                     // base.py:719
        const bool pending = in_bar && inside && (sd == 1 || sd == -1);
        const int key = pending ? lvl * 2 + (sd < 0 ? 1 : 0) : 0;
        if constexpr (EXACT) {
            unsigned ui;
            bool ok;
            double ud;
            if constexpr (AF64) {
                ud = ldexp((double)a, -q);
                ok = ud >= 0.0 && ud < 2147483648.0 && ud == rint(ud);
                ui = ok ? (unsigned)ud : 0u;
            } else {
                const float u = ldexpf(a, -q);                        // exact scaling
                ui = (unsigned)u;                                     // saturating convert, NaN -> 0
                ok = (float)ui == u && u < 2147483648.f;              // whole, non-negative, below 2^31
                ud = (double)u;
            }
            units_ok &= ok || !pending;
            if (pending && ok) {
                atomicAdd(&units[key], ui);
                atomicAdd(&cnt[key], 1);
                atot += ud;                                           // units (fp_certified_units)
            }
        } else {
            if (pending) {                                            // statistics that pick the quantum of later bars
                const int lb = fp_lowbit_exp(a);
                lbmin = lb < lbmin ? lb : lbmin;
                atot += fabs((double)a);
            }
            const int before = pending ? cnt[key] : 0;
            __builtin_amdgcn_wave_barrier();
            const int rank = pending ? atomicAdd(&cnt[key], 1) - before : -1;     // old values come in lane (= tick) order
            const int rounds = fmk_dpp_reduce(rank, -1, FmkOpMax()) + 1;
            for (int t = 0; t < rounds; ++t) {
                if (rank == t) {
                    const float v = vol[key];
                    if constexpr (AF64) vol[key] = (float)((double)v + a);       // f32 element += f64 amount
                    else vol[key] = v + a;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    if (med_on) { med->below = m_below; med->ncand = m_n; med->kmin = m_kmin; med->kmax = m_kmax; }
    FpStats st;
    st.lbmin = fmk_dpp_reduce(lbmin, FP_Q_UNKNOWN, FmkOpMin());
    st.atot = fmk_dpp_reduce(atot, 0.0, FmkOpAdd());
    st.units_ok = __ballot(!units_ok) == 0;
    st.bad = __ballot(bad) != 0;
    __builtin_amdgcn_wave_barrier();
    return st;
}

// np.median of the bar's n_t (1 .. FP_MED_MAX_TICKS) float32 amounts after the sweep (see the head of this file); moves the bracket.
// The bracket is kept as [v1 - w, v2 + w] in key units around the bar's middle keys; its half-width w follows the candidate count
// (the aim: 2R + 2 candidates, R ~ 2.4 sqrt(ticks)), so the accepting path needs ONE selection on four registers per lane.
__device__ __forceinline__ double fp_median_finish(FpMed &med, uint32_t &width, int &n_fallback, const float *__restrict__ amount,
                                                   int64_t start, int64_t n_t, int lane, uint32_t *buf)
{
    typedef MedKey<false> MK;
    const int64_t k1 = (n_t - 1) >> 1, k2 = n_t >> 1;
    int R = (int)(2.4f * sqrtf((float)n_t));
    R = R < 32 ? 32 : (R > 100 ? 100 : R);
    uint32_t v1 = 0, v2 = 0;
    bool ok = false, isnan = false;
    if (med.have) {
        const uint32_t kmin = med_wave_umin<uint32_t>(med.kmin), kmax = med_wave_umax<uint32_t>(med.kmax);
        isnan = kmin < MK::KEY_NEG_INF || kmax > MK::KEY_POS_INF;
        const int64_t below = med.below, nc = med.ncand;
        if (isnan) ok = true;                                         // np.median: NaN; the bracket stays
        else if (below <= k1 && k2 < below + nc) {
            if (med.blo == med.bhi) { v1 = v2 = med.blo; ok = true; }  // every candidate is the same key: no list needed
            else if (nc <= FP_MED_CAP) {
                MedBar<false, FP_MED_CAP / 64, false> cb;
#pragma unroll
                for (int r = 0; r < FP_MED_CAP / 64; ++r) {
                    const int j = r * 64 + lane;
                    cb.key[r] = j < (int)nc ? med.cand[j] : MK::MAXK;
                }
                cb.amount = nullptr; cb.start = 0; cb.cnt = nc; cb.lane = lane;
                (void)med_rank_pair<false, FP_MED_CAP / 64, false>(cb, buf, k1 - below, k2 - below, v1, v2);
                v1 = (uint32_t)fmk_uniform((int)v1);
                v2 = (uint32_t)fmk_uniform((int)v2);
                // too many candidates: narrower next time; too few (the middle came close to an end): wider
                const int aim = 2 * R + 2;
                if (nc > aim + aim / 4) width -= width >> 2;
                else if (nc < aim - aim / 4) width += (width >> 2) + 1;
                med.blo = v1 > width ? v1 - width : 0;
                med.bhi = v2 < 0xFFFFFFFFu - width ? v2 + width : 0xFFFFFFFFu;
                ok = true;
            }
        }
    }
    if (!ok) {
        // generic selection on the amounts themselves (re-read on every pass: L2 hits), then the bracket for the next bar
        ++n_fallback;
        MedBar<false, 0, false> gb;
        gb.amount = amount; gb.start = start; gb.cnt = n_t; gb.lane = lane;
        if (!med_rank_pair<false, 0, false>(gb, buf, k1, k2, v1, v2)) { isnan = true; med.have = 0; }
        else {
            const int64_t ra = k1 - R > 0 ? k1 - R : 0, rb = k2 + R < n_t - 1 ? k2 + R : n_t - 1;
            uint32_t nlo, nhi, t;
            (void)med_rank_pair<false, 0, false>(gb, buf, ra, ra, nlo, t);
            (void)med_rank_pair<false, 0, false>(gb, buf, rb, rb, nhi, t);
            const int64_t inside = gb.count_le(nhi) - (nlo > 0 ? gb.count_le(nlo - 1) : 0);
            if (inside > FP_MED_CAP) { nlo = v1; nhi = v2; }             // ties: the (possibly degenerate) bracket of the middle keys
            med.blo = (uint32_t)fmk_uniform((int)nlo);
            med.bhi = (uint32_t)fmk_uniform((int)nhi);
            v1 = (uint32_t)fmk_uniform((int)v1);
            v2 = (uint32_t)fmk_uniform((int)v2);
            const uint32_t wl = v1 - med.blo, wh = med.bhi - v2;
            width = wl > wh ? wl : wh;
            med.have = 1;
        }
    }
    if (isnan) return NAN;
    return (n_t & 1) ? MK::value(v1) : (MK::value(v1) + MK::value(v2)) / 2.0;   // np.median: mean of the two middle elements
}

// Are same-address LDS atomics with return applied in ascending lane order?  (They are on gfx950; the tick-ordered sweep above
// relies on it, so the device is ASKED once per process -- 7 access patterns x 64 repetitions x 64 waves -- and the ballot-loop
// sweep stays the fallback.)
__global__ __launch_bounds__(64) void k_fp_lds_order_probe(int *out_bad)
{
    __shared__ int slot[64];
    const int lane = threadIdx.x & 63;
    int bad = 0;
    for (int pattern = 0; pattern < 7; ++pattern)
        for (int rep = 0; rep < 64; ++rep) {
            slot[lane] = 0;
            __builtin_amdgcn_wave_barrier();
            int key;
            switch (pattern) {
            case 0: key = 0; break;
            case 1: key = lane & 3; break;
            case 2: key = lane >> 4; break;
            case 3: key = (lane * 7 + rep + (int)blockIdx.x) % 5; break;
            case 4: key = (lane ^ rep) & 7; break;
            default: key = (int)(((unsigned)lane * 2654435761u) >> 27) % (1 + (rep + (int)blockIdx.x) % 9); break;
            }
            const bool act = pattern < 6 ? true : ((lane * 13 + rep) % 3 != 0);
            int old = -1;
            if (act) old = atomicAdd(&slot[key], 1);
            __builtin_amdgcn_wave_barrier();
            const uint64_t am = __ballot(act);
            int expect = 0;
            for (int l = 0; l < 64; ++l) {
                const int kl = __shfl(key, l, 64);
                if (l < lane && ((am >> l) & 1) && kl == key) ++expect;
            }
            bad += act && old != expect;
        }
    if (__ballot(bad != 0) != 0 && lane == 0) atomicAdd(out_bad, 1);
}

static int fp_lds_atomics_in_lane_order(fmk_ctx *ctx)
{
    static int known = -1;
    if (known >= 0) return known;
    int *d = (int *)(ctx->d_mail + 30);
    if (hipMemsetAsync(d, 0, 4, ctx->stream) != hipSuccess) return 0;
    k_fp_lds_order_probe<<<64, 64, 0, ctx->stream>>>(d);
    int bad = 1;
    if (hipMemcpyAsync(&bad, d, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return 0;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return 0;
    known = bad == 0 ? 1 : 0;
    return known;
}

// ---------------------------------------------------------------------------------------
// The tick-ordered sweep for float32 amounts, SORTED (round 6).  fp_accumulate_lean's ordered path serves a 64-tick chunk in as many
// dependent LDS read-add-write rounds as its busiest (level, side) key has ticks -- ~20 on a tape whose trades cluster on a few levels,
// ~4 000 cycles per chunk: the footprint sweep of full-mantissa sizes (what real sizes are) was bound by that latency, not by its
// ~110 instructions per chunk (profiles/r06_cfg4_fused.txt).  Here a SEGMENT of 256 ticks is sorted by (key, tick) first -- one LDS
// atomic per tick gives its rank within its key in tick order (lane-ordered atomics, chunks in order), an exclusive scan over the
// segment's key range gives every key its slice, one LDS store per tick scatters the amounts -- and then every key's float32 sum is a
// plain sequential loop of ONE lane over its own contiguous slice, all keys in parallel, each in the reference's order
// (base.py:713-717; what the workgroup-per-bar kernel does for a whole bar).  The level sums that come out are the same float32
// operations on the same operands in the same order per key.  A segment whose keys span more than 256 values (a price that moves more
// than 128 levels within 256 trades) takes the rounds.  scnt[FP_SEG] (zero on entry, left zero) and sorted[FP_SEG]: 2 KB of LDS per wave.
// ---------------------------------------------------------------------------------------
#define FP_SEG 256
#define FP_SORT_BYTES (FP_SEG * 8)
// STATS: also measure the quantum a later bar could certify with (lowest set bit, magnitude); a wave whose bars keep failing that test
// sweeps without (k_bar_footprints: re-probed every 32nd bar).
template <bool STATS>
__device__ __forceinline__ FpStats fp_accumulate_sorted(const double *__restrict__ price, const float *__restrict__ amount,
                                                        const int8_t *__restrict__ side, int64_t s, int64_t e, int64_t low, int L,
                                                        double tick, double inv_tick, int lane, float *vol, int *cnt, int *scnt,
                                                        float *sorted, FpMed *med)
{
    int lbmin = FP_Q_UNKNOWN;
    double atot = 0.0;
    bool bad = false;
    const int ilow = (int)low;
    const bool med_on = med != nullptr && med->have;
    uint32_t m_lo = 0, m_hi = 0, m_kmin = 0xFFFFFFFFu, m_kmax = 0;
    int m_below = 0, m_n = 0;
    if (med_on) { m_lo = med->blo; m_hi = med->bhi; }
    const double *pp = price + (s + 1);
    const float *ap = amount + (s + 1);
    const int8_t *sp = side + (s + 1);
    const int total = (int)(e - s);                                   // ticks of the bar (<= 2^31: one wave per bar)
    for (int j0 = 0; j0 < total; j0 += FP_SEG) {
        double p[4];
        float a[4];
        int sd[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {                                 // the segment's loads, all in flight together
            const int j = j0 + 64 * c + lane;
            const bool in = j < total;
            p[c] = in ? pp[j] : 0.0; a[c] = in ? ap[j] : 0.f; sd[c] = in ? (int)sp[j] : 0;
        }
        int key[4];
        bool pend[4];
        int kmin = 0x7FFFFFFF, kmax = -1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool in_bar = j0 + 64 * c + lane < total;
            if (med_on) {       // the median's bracket: every tick of the bar, signed or not, inside the level range or not
                const uint32_t k = MedKey<false>::tokey(__float_as_uint(a[c]));
                m_kmin = (in_bar && k < m_kmin) ? k : m_kmin;
                m_kmax = (in_bar && k > m_kmax) ? k : m_kmax;
                m_below += __popcll(__ballot(in_bar && k < m_lo));
                const bool inb = in_bar && k >= m_lo && k <= m_hi;
                const uint64_t bm = __ballot(inb);
                const int pos = m_n + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0));
                if (inb && pos < FP_MED_CAP) med->cand[pos] = k;
                m_n += __popcll(bm);
            }
            // level = int(round(price / tick)) - low (base.py:700-707), as in fp_accumulate_lean
            const double qq = p[c] * inv_tick;
            double r = rint(qq);
            if (__ballot(in_bar && 0.5 - fabs(qq - r) <= fabs(qq) * 1e-15) != 0) {
                if (0.5 - fabs(qq - r) <= fabs(qq) * 1e-15) r = rint(p[c] / tick);
            }
            const int lvl = (int)r - ilow;
            const bool inside = (unsigned)lvl < (unsigned)L;
            bad |= in_bar && !inside;                                 // base.py:719
            pend[c] = in_bar && inside && (sd[c] == 1 || sd[c] == -1);
            key[c] = pend[c] ? lvl * 2 + (sd[c] < 0 ? 1 : 0) : 0;
            if (pend[c]) {
                if constexpr (STATS) {                                // statistics that pick the quantum of later bars
                    const int lb = fp_lowbit_exp(a[c]);
                    lbmin = lb < lbmin ? lb : lbmin;
                    atot += fabs((double)a[c]);
                }
                kmin = key[c] < kmin ? key[c] : kmin;
                kmax = key[c] > kmax ? key[c] : kmax;
            }
        }
        if (2 * L <= FP_SEG) { kmin = 0; kmax = 2 * L - 1; }          // a bar of <= 128 levels: its keys ARE the slots
        else {
            kmin = fmk_dpp_reduce(kmin, 0x7FFFFFFF, FmkOpMin());
            kmax = fmk_dpp_reduce(kmax, -1, FmkOpMax());
        }
        if (kmax < 0) continue;                                       // no signed tick inside the level range
        if (kmax - kmin >= FP_SEG) {                                  // keys too far apart for one slice table: the rounds, chunk by chunk
#pragma unroll 1
            for (int c = 0; c < 4; ++c) {
                const int before = pend[c] ? cnt[key[c]] : 0;
                __builtin_amdgcn_wave_barrier();
                const int rank = pend[c] ? atomicAdd(&cnt[key[c]], 1) - before : -1;
                const int rounds = fmk_dpp_reduce(rank, -1, FmkOpMax()) + 1;
                for (int t = 0; t < rounds; ++t) {
                    if (rank == t) vol[key[c]] = vol[key[c]] + a[c];
                    __builtin_amdgcn_wave_barrier();
                }
            }
            continue;
        }
        // rank of every tick within its key, in tick order: lane-ordered atomics, the chunks one after the other
        int rk[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            rk[c] = pend[c] ? atomicAdd(&scnt[key[c] - kmin], 1) : 0;
            __builtin_amdgcn_wave_barrier();
        }
        // every key's slice of sorted[]: exclusive scan of the counts; the table word becomes base << 16 | count
        {
            int4 c4 = *(const int4 *)&scnt[4 * lane];
            const int t = c4.x + c4.y + c4.z + c4.w;
            const int ex = fmk_dpp_iscan(t, 0, FmkOpAdd()) - t;
            const int b0 = ex, b1 = b0 + c4.x, b2 = b1 + c4.y, b3 = b2 + c4.z;
            c4.x |= b0 << 16; c4.y |= b1 << 16; c4.z |= b2 << 16; c4.w |= b3 << 16;
            __builtin_amdgcn_wave_barrier();
            *(int4 *)&scnt[4 * lane] = c4;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (pend[c]) sorted[(scnt[key[c] - kmin] >> 16) + rk[c]] = a[c];
        __builtin_amdgcn_wave_barrier();
        // the sums: slot = lane + 64 r (neighbouring keys on different lanes), each a sequential float32 loop in tick order
        const int span = kmax - kmin;
#pragma unroll 1
        for (int r = 0; r < 4 && 64 * r <= span; ++r) {
            const int slot = lane + 64 * r;
            const int w = scnt[slot];
            const int n_k = w & 0xFFFF;
            if (n_k) {
                const int base = w >> 16, k = kmin + slot;
                float sum = vol[k];
                int i = 0;
                for (; i + 4 <= n_k; i += 4) {
                    const float x0 = sorted[base + i], x1 = sorted[base + i + 1], x2 = sorted[base + i + 2], x3 = sorted[base + i + 3];
                    sum = sum + x0; sum = sum + x1; sum = sum + x2; sum = sum + x3;
                }
                for (; i < n_k; ++i) sum = sum + sorted[base + i];
                vol[k] = sum;
                cnt[k] += n_k;
            }
        }
        __builtin_amdgcn_wave_barrier();
        *(int4 *)&scnt[4 * lane] = make_int4(0, 0, 0, 0);             // (every word carries its base by now: the whole table)
        __builtin_amdgcn_wave_barrier();
    }
    if (med_on) { med->below = m_below; med->ncand = m_n; med->kmin = m_kmin; med->kmax = m_kmax; }
    FpStats st;
    st.lbmin = FP_Q_UNKNOWN;
    st.atot = 0.0;
    if constexpr (STATS) {
        st.lbmin = fmk_dpp_reduce(lbmin, FP_Q_UNKNOWN, FmkOpMin());
        st.atot = fmk_dpp_reduce(atot, 0.0, FmkOpAdd());
    }
    st.units_ok = true;
    st.bad = __ballot(bad) != 0;
    __builtin_amdgcn_wave_barrier();
    return st;
}

// ---------------------------------------------------------------------------------------
// phase 2: one wave per bar
// ---------------------------------------------------------------------------------------
// FAST: the classes of 512 levels and more (fp_emit_bar's fast_sum; a template flag so that the 128 / 256-level instantiations keep their code)
template <bool AF64, bool GLOBAL, bool MED = false, bool FAST = false>
__global__ __launch_bounds__(256) void k_bar_footprints(const double *__restrict__ price,
                                                        const void *__restrict__ amount,
                                                        const int8_t *__restrict__ side,
                                                        const int64_t *__restrict__ ci, int64_t nb, double tick,
                                                        const double *__restrict__ lows, double imb_mult,
                                                        const int64_t *__restrict__ off, int lmin, int lmax,
                                                        FpOut o, unsigned long long *n_bad, int force_ordered,
                                                        unsigned char *gscratch, int lean,
                                                        const unsigned long long *only = nullptr,
                                                        double *__restrict__ o_median = nullptr,
                                                        int *__restrict__ saw_long = nullptr,
                                                        int64_t skip_above = INT64_MAX, int skip_lmax = 0 /* longer bars of <= skip_lmax levels: k_bar_footprints_wide */)
{
    static_assert(!(MED && AF64), "the in-sweep median serves float32 amounts");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = fmk_lane();
    const int wib = fmk_uniform((int)(threadIdx.x >> 6));
    const int wpb = blockDim.x >> 6;
    // FAST: 16 B per level (no aux area) + the tree routine's tables
    constexpr size_t sort_bytes = (!AF64 && !GLOBAL) ? (size_t)FP_SORT_BYTES : 0;     // fp_accumulate_sorted's slice table + sorted amounts
    const size_t per_wave = (FAST ? (size_t)lmax * 16 + FMK_PW_PAR_STK * 4
                                  : (size_t)lmax * 24 + 256 + ((MED && !GLOBAL) ? (size_t)FP_MED_CAP * 4 : 0)) + sort_bytes;
    // the wave's histogram: LDS for the three narrow classes (LDS-typed pointers: ds_add / ds_read), a slice of global
    // scratch for bars wider than 2048 levels (same code; a wave's own stores are visible to its later loads)
    unsigned char *mine;
    if constexpr (GLOBAL) mine = gscratch + ((size_t)blockIdx.x * wpb + wib) * per_wave;
    else mine = smem + (size_t)wib * per_wave;
    float *vol = (float *)mine;                                   // [2*lmax]  buy = 2l, sell = 2l+1
    int *cnt = (int *)(mine + (size_t)lmax * 8);                  // [2*lmax]
    float *aux = FAST ? nullptr : (float *)(mine + (size_t)lmax * 16);             // [2*lmax]  tot[], later q2[]
    int *stk = (int *)(mine + (size_t)lmax * (FAST ? 16 : 24));   // 64 ints (FAST: FMK_PW_PAR_STK)
    // the median's bracket and candidate list (LDS classes with the straight-line sweep only; otherwise every bar takes the
    // generic selection on its re-read amounts)
    FpMed med;
    med.cand = (MED && !GLOBAL) ? (uint32_t *)(mine + (size_t)lmax * 24 + 256) : nullptr;
    med.blo = med.bhi = 0; med.have = 0; med.below = med.ncand = 0; med.kmin = 0xFFFFFFFFu; med.kmax = 0;
    FpMed *medp = (MED && !GLOBAL && lean) ? &med : nullptr;
    int *scnt = nullptr;
    float *sorted = nullptr;
    if constexpr (!AF64 && !GLOBAL) {
        scnt = (int *)(mine + per_wave - sort_bytes);
        sorted = (float *)(scnt + FP_SEG);
        for (int k = lane; k < FP_SEG; k += 64) scnt[k] = 0;
        __builtin_amdgcn_wave_barrier();
    }
    uint32_t med_width = 0;
    int med_fallbacks = 0;
    const int64_t wave0 = (int64_t)blockIdx.x * wpb + wib;
    const int64_t nwaves = (int64_t)gridDim.x * wpb;
    const double inv_tick = 1.0 / tick;
    int wq = FP_Q_UNKNOWN;        // quantum exponent the previous bar of this wave certified with
    int no_quantum = 0;           // bars in a row whose tick-ordered sweep found no usable quantum
    // `only` (list mode: [0] = count, [32...] = bar numbers): the bars k_bar_footprints_lanes left to this schedule
    const int64_t todo = only ? (int64_t)only[0] : nb;
    for (int64_t it = wave0; it < todo; it += nwaves) {
        const int64_t b = only ? fmk_uniform((int64_t)only[32 + it]) : it;
        const int64_t base = fmk_uniform(off[b]);
        const int L = (int)fmk_uniform(off[b + 1] - base);
        if (L <= lmin || L > lmax) continue;          // handled by another launch (or L == 0)
        const int64_t s = fmk_uniform(ci[b]);
        const int64_t e = fmk_uniform(ci[b + 1]);
        if (e - s > skip_above && L <= skip_lmax) {                // a workgroup has taken this bar (k_bar_footprints_wide)
            if constexpr (MED) {                                       // ... and its median is the long-bar kernels'
                if (lane == 0 && __hip_atomic_load(saw_long, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                    __hip_atomic_store(saw_long, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            continue;
        }
        const int64_t low = fp_level(lows[b], tick);
        for (int k = lane; k < 2 * L; k += 64) { vol[k] = 0.f; cnt[k] = 0; }
        __builtin_amdgcn_wave_barrier();
        // Exact path first (with the quantum that worked for the previous bar); when its certificate fails the
        // tick-ordered path runs.
        FpStats st;
        bool done = false;
        if (!force_ordered && wq != FP_Q_UNKNOWN) {
            bool did = false;
            if constexpr (!GLOBAL) {
                if (lean) { st = fp_accumulate_lean<AF64, true>(price, amount, side, s, e, low, L, tick, inv_tick, lane, vol, cnt, wq, medp); did = true; }
            }
            if (!did) st = fp_accumulate<AF64, true>(price, amount, side, s, e, low, L, tick, inv_tick, lane, vol, cnt, wq);
            done = fp_certified_units(st) || fp_certified_units_per_key(st, (const unsigned *)vol, 2 * L, lane);
            if (done) {       // units -> float32 (exact)
                unsigned *units = (unsigned *)vol;
                const int qq = wq;
                for (int k = lane; k < 2 * L; k += 64) vol[k] = ldexpf((float)units[k], qq);
                __builtin_amdgcn_wave_barrier();
            } else {          // finer or larger amounts than the quantum in use: tick-ordered sweep, which also
                              // measures the quantum for the following bars
                for (int k = lane; k < 2 * L; k += 64) { vol[k] = 0.f; cnt[k] = 0; }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (!done) {
            bool did = false;
            if constexpr (!GLOBAL) {
                if constexpr (!AF64) {
                    if (lean) {
                        // full mantissas never certify: after four bars in a row without a usable quantum the wave stops measuring one
                        // (the statistics are ~14 of the sweep's ~100 instructions per chunk), and looks again every 32nd bar
                        const bool stats = no_quantum < 4 || (no_quantum & 31) == 0;
                        if (stats) st = fp_accumulate_sorted<true>(price, (const float *)amount, side, s, e, low, L, tick, inv_tick, lane, vol, cnt, scnt, sorted, medp);
                        else st = fp_accumulate_sorted<false>(price, (const float *)amount, side, s, e, low, L, tick, inv_tick, lane, vol, cnt, scnt, sorted, medp);
                        did = true;
                    }
                } else if (lean) { st = fp_accumulate_lean<AF64, false>(price, amount, side, s, e, low, L, tick, inv_tick, lane, vol, cnt, 0, medp); did = true; }
            }
            if (!did) st = fp_accumulate<AF64, false>(price, amount, side, s, e, low, L, tick, inv_tick, lane, vol, cnt, 0);
            // remember a usable quantum for the next bar (if this bar would have certified)
            FpStats probe = st;
            probe.units_ok = true;
            const bool usable = st.lbmin != FP_Q_UNKNOWN && st.lbmin != (int)0x80000000 &&
                                (fp_certified(probe, st.lbmin) || fp_certified_per_key(probe, st.lbmin, vol, 2 * L, lane));
            wq = usable ? st.lbmin : FP_Q_UNKNOWN;
            no_quantum = usable ? 0 : no_quantum + 1;
        }
        if (st.bad && lane == 0 && n_bad) atomicAdd(n_bad, 1ULL);
        __builtin_amdgcn_wave_barrier();

        fp_emit_bar(o, b, base, L, low, lmax, imb_mult, lane, vol, cnt, aux, stk, FAST);
        if constexpr (MED) {
            // np.median of the bar's trade sizes (base.py:401-404); an empty bar has median 0 (base.py:352-361)
            const int64_t n_t = e - s;
            __builtin_amdgcn_wave_barrier();
            double m = 0.0;
            if (n_t > FP_MED_MAX_TICKS) {
                if (lane == 0 && __hip_atomic_load(saw_long, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                    __hip_atomic_store(saw_long, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                med.have = medp ? med.have : 0;
                continue;
            } else if (n_t > 0) {
                if (!medp) med.have = 0;                                  // nothing was collected: generic selection
                m = fp_median_finish(med, med_width, med_fallbacks, (const float *)amount, s + 1, n_t, lane, (uint32_t *)stk);
            }
            if (lane == 0) o_median[b] = m;
            __builtin_amdgcn_wave_barrier();
        }
    }
    if constexpr (MED) {
        if (lane == 0 && med_fallbacks) atomicAdd(saw_long + 1, med_fallbacks);     // diagnostics (fmk_diag_fp_median_fallbacks)
    }
}

// ---------------------------------------------------------------------------------------
// Bars of more than FPW_MIN ticks and at most FPW_MAX_LEVELS levels (hourly, daily bars; the histogram takes 24 B of LDS per level:
// up to 147 KB of the CU's 160): a WORKGROUP per bar on ONE LDS histogram
// (round 3).  With a wave per bar the 580 daily bars of a 1e9-tick tape are 580 waves streaming 1.7e6 ticks each: 73 ms.  The
// integer-unit sweep is order-free, so sixteen waves take a sixteenth of the bar each and add into the shared histogram with LDS
// atomics; the certificate is the per-key one (fp_certified_units_per_key: a daily bar's TOTAL is far beyond 2^24 units, its
// levels are not).  The quantum 2^q travels from bar to bar as in the wave kernel; without one (first bar of the workgroup, or
// the last one failed) a statistics pass over the bar finds it.  A bar that does not certify -- amounts with full mantissas,
// negative or non-finite amounts -- is swept in tick order by wave 0 alone, as before.  Level rows and features: wave 0
// (fp_emit_bar), while the second workgroup of the CU sweeps.
// ---------------------------------------------------------------------------------------
#define FPW_MIN 16384
#define FPW_WAVES 16
#define FPW_MAX_LEVELS 6144

// lowest-bit / magnitude statistics of the pending ticks (signed side, inside the level range) of (s, e]: what the tick-ordered
// sweep reports, without a histogram
template <bool AF64>
__device__ __forceinline__ FpStats fp_stats_lean(const double *__restrict__ price, const void *__restrict__ amount,
                                                 const int8_t *__restrict__ side, int64_t s, int64_t e, int64_t low, int L,
                                                 double tick, double inv_tick, int lane)
{
    typedef typename std::conditional<AF64, double, float>::type AmtT;
    int lbmin = FP_Q_UNKNOWN;
    double atot = 0.0;
    bool bad = false;
    const int ilow = (int)low;
    const double *pp = price + (s + 1);
    const AmtT *ap = (const AmtT *)amount + (s + 1);
    const int8_t *sp = side + (s + 1);
    const int total = (int)(e - s);
    for (int j0 = 0; j0 < total; j0 += 256) {
        double p[4];
        AmtT a[4];
        int sd[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u * 64 + lane;
            const int jc = j < total ? j : total - 1;
            p[u] = pp[jc]; a[u] = ap[jc]; sd[u] = sp[jc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in_bar = j0 + u * 64 + lane < total;
            const int lvl = fp_level32(p[u], tick, inv_tick) - ilow;
            const bool inside = (unsigned)lvl < (unsigned)L;
            bad |= in_bar && !inside;
            if (in_bar && inside && (sd[u] == 1 || sd[u] == -1)) {
                const int lb = fp_lowbit_exp(a[u]);
                lbmin = lb < lbmin ? lb : lbmin;
                atot += fabs((double)a[u]);
            }
        }
    }
    FpStats st;
    st.lbmin = fmk_dpp_reduce(lbmin, FP_Q_UNKNOWN, FmkOpMin());
    st.atot = fmk_dpp_reduce(atot, 0.0, FmkOpAdd());
    st.units_ok = true;
    st.bad = __ballot(bad) != 0;
    return st;
}

// (level, side) key of one tick as the sweeps compute it (base.py:700-707: level = int(round(price / tick)) - low, one multiply, the
// exact division deciding the products within 1e-15 of a half-integer as a wave-uniform branch); -1: not pending (outside the bar,
// unsigned tick, outside the level range -- the latter sets `bad`, base.py:719).  All lanes of the wave call.
__device__ __forceinline__ int fp_key_lean(double p, int sd, bool in_bar, double tick, double inv_tick, int ilow, int L, bool &bad)
{
    const double qq = p * inv_tick;
    double r = rint(qq);
    if (__ballot(in_bar && 0.5 - fabs(qq - r) <= fabs(qq) * 1e-15) != 0) {
        if (0.5 - fabs(qq - r) <= fabs(qq) * 1e-15) r = rint(p / tick);
    }
    const int lvl = (int)r - ilow;
    const bool inside = (unsigned)lvl < (unsigned)L;
    bad |= in_bar && !inside;
    return (in_bar && inside && (sd == 1 || sd == -1)) ? lvl * 2 + (sd < 0 ? 1 : 0) : -1;
}

// ticks per (level, side) key of (s, e] into cnt[] (LDS atomics: any order), and the statistics of the pending float32 amounts that
// pick the quantum (what fp_stats_lean reports); st.bad: a tick fell outside the level range.  Per-lane partial statistics are
// ACCUMULATED into lb / at (the caller reduces them once).
__device__ __forceinline__ bool fp_count_lean(const double *__restrict__ price, const float *__restrict__ amount,
                                              const int8_t *__restrict__ side, int64_t s, int64_t e, int64_t low, int L, double tick,
                                              double inv_tick, int lane, int *cnt, int &lb, double &at)
{
    bool bad = false;
    const int ilow = (int)low;
    const double *pp = price + (s + 1);
    const float *ap = amount + (s + 1);
    const int8_t *sp = side + (s + 1);
    const int total = (int)(e - s);
    for (int j0 = 0; j0 < total; j0 += 256) {
        double p[4];
        float a[4];
        int sd[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u * 64 + lane;
            const int jc = j < total ? j : total - 1;
            p[u] = pp[jc]; a[u] = ap[jc]; sd[u] = sp[jc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = fp_key_lean(p[u], sd[u], j0 + u * 64 + lane < total, tick, inv_tick, ilow, L, bad);
            if (key >= 0) {
                atomicAdd(&cnt[key], 1);
                const int l2 = fp_lowbit_exp(a[u]);
                lb = l2 < lb ? l2 : lb;
                at += fabs((double)a[u]);
            }
        }
    }
    return __ballot(bad) != 0;
}

// The float32 amounts of (s, e] moved to sorted[] in (key, tick) order: cursor[key] starts at the key's first slot (exclusive scan
// of the counts).  ONE wave, chunk after chunk: an LDS atomic with return hands out its old values in ascending lane order
// (fp_lds_atomics_in_lane_order) and a wave's LDS instructions complete in the order they were issued, so a key's slots are filled
// in tick order.  Four chunks are in flight: their atomics are issued back to back.
__device__ __forceinline__ void fp_scatter_lean(const double *__restrict__ price, const float *__restrict__ amount,
                                                const int8_t *__restrict__ side, int64_t s, int64_t e, int64_t low, int L, double tick,
                                                double inv_tick, int lane, int *cursor, float *__restrict__ sorted)
{
    bool bad = false;
    const int ilow = (int)low;
    const double *pp = price + (s + 1);
    const float *ap = amount + (s + 1);
    const int8_t *sp = side + (s + 1);
    const int total = (int)(e - s);
    double pn[4];
    float an[4];
    int sn[4];
    auto load = [&](int j0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u * 64 + lane;
            const int jc = j < total ? j : total - 1;
            pn[u] = pp[jc]; an[u] = ap[jc]; sn[u] = sp[jc];
        }
    };
    if (total > 0) load(0);
    for (int j0 = 0; j0 < total; j0 += 256) {
        double p[4];
        float a[4];
        int sd[4], key[4], pos[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { p[u] = pn[u]; a[u] = an[u]; sd[u] = sn[u]; }
        if (j0 + 256 < total) load(j0 + 256);
#pragma unroll
        for (int u = 0; u < 4; ++u) key[u] = fp_key_lean(p[u], sd[u], j0 + u * 64 + lane < total, tick, inv_tick, ilow, L, bad);
#pragma unroll
        for (int u = 0; u < 4; ++u) pos[u] = key[u] >= 0 ? atomicAdd(&cursor[key[u]], 1) : 0;     // chunk u before chunk u + 1
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (key[u] >= 0) sorted[pos[u]] = a[u];
    }
}

template <bool AF64>
__global__ __launch_bounds__(64 * FPW_WAVES) void k_bar_footprints_wide(const double *__restrict__ price, const void *__restrict__ amount,
                                                                      const int8_t *__restrict__ side, const int64_t *__restrict__ ci,
                                                                      const int64_t *__restrict__ list, double tick,
                                                                      const double *__restrict__ lows, double imb_mult,
                                                                      const int64_t *__restrict__ off, FpOut o,
                                                                      unsigned long long *n_bad, int force_ordered, int lean,
                                                                      int lmax, float *__restrict__ sorted_all,
                                                                      unsigned long long *__restrict__ defer, int nseg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *vol = (float *)smem;                                        // [2 * lmax]  buy = 2l, sell = 2l + 1
    int *cnt = (int *)(smem + (size_t)lmax * 8);
    float *aux = (float *)(smem + (size_t)lmax * 16);
    int *stk = (int *)(smem + (size_t)lmax * 24);                      // 64 ints
    __shared__ double s_atot[FPW_WAVES];
    __shared__ int s_lb[FPW_WAVES], s_flag[FPW_WAVES];                 // s_flag: bit 0 units not ok, bit 1 bad level
    __shared__ unsigned s_umax[FPW_WAVES];
    const int lane = fmk_lane();
    const int w = fmk_uniform((int)(threadIdx.x >> 6));
    const double inv_tick = 1.0 / tick;
    unsigned *units = (unsigned *)vol;
    int wq = FP_Q_UNKNOWN;                                             // (block-uniform) the quantum the previous bar certified with
    int no_quantum = 0;                                                // bars whose first ticks did not lead to a certified sweep
    const int64_t n_list = list[0];
    for (int64_t it = blockIdx.x; it < n_list; it += gridDim.x) {
        const int64_t b = list[1 + it];
        const int64_t base = off[b];
        const int L = (int)(off[b + 1] - base);
        if (L <= 0 || L > lmax) continue;                              // no rows / the global-scratch class of the wave kernel
        const int64_t s = ci[b], e = ci[b + 1];
        const int64_t low = fp_level(lows[b], tick);
        int64_t seg = (e - s + FPW_WAVES - 1) / FPW_WAVES;
        seg = (seg + 63) & ~(int64_t)63;
        const int64_t s_w = s + (int64_t)w * seg < e ? s + (int64_t)w * seg : e;
        const int64_t e_w = s_w + seg < e ? s_w + seg : e;
        auto zero = [&]() {
            __syncthreads();
            for (int k = (int)threadIdx.x; k < 2 * L; k += 64 * FPW_WAVES) { vol[k] = 0.f; cnt[k] = 0; }
            __syncthreads();
        };
        // all waves: combine the per-wave statistics (block-uniform result)
        auto combine = [&](const FpStats &mine) -> FpStats {
            if (lane == 0) { s_atot[w] = mine.atot; s_lb[w] = mine.lbmin; s_flag[w] = (mine.units_ok ? 0 : 1) | (mine.bad ? 2 : 0); }
            __syncthreads();
            FpStats t;
            t.atot = 0.0; t.lbmin = FP_Q_UNKNOWN; t.units_ok = true; t.bad = false;
            for (int k = 0; k < FPW_WAVES; ++k) {
                t.atot += s_atot[k];
                t.lbmin = s_lb[k] < t.lbmin ? s_lb[k] : t.lbmin;
                t.units_ok = t.units_ok && !(s_flag[k] & 1);
                t.bad = t.bad || (s_flag[k] & 2);
            }
            __syncthreads();
            return t;
        };
        // the integer-unit sweep with quantum 2^q by all waves: true when every float32 add of the reference is certified exact
        bool bad_level = false;
        auto attempt = [&](int q) -> bool {
            zero();
            const FpStats mine = fp_accumulate_lean<AF64, true>(price, amount, side, s_w, e_w, low, L, tick, inv_tick, lane, vol, cnt, q);
            const FpStats t = combine(mine);
            bad_level = t.bad;
            if (!t.units_ok || !(t.atot < 4294967296.0)) return false;  // (no 32-bit counter can have wrapped)
            if (t.atot < 16777216.0) return true;
            unsigned m = 0;
            for (int k = (int)threadIdx.x; k < 2 * L; k += 64 * FPW_WAVES) m = units[k] > m ? units[k] : m;
            m = (unsigned)fmk_dpp_reduce((int)(m >> 1), 0, FmkOpMax());   // (halved: compared as signed ints)
            if (lane == 0) s_umax[w] = m;
            __syncthreads();
            unsigned mm = 0;
            for (int k = 0; k < FPW_WAVES; ++k) mm = s_umax[k] > mm ? s_umax[k] : mm;
            __syncthreads();
            return mm < (16777216u >> 1);
        };
        bool done = false;
        int q_used = wq;
        bool parallel_ok = false;                                          // the tick-ordered path of this kernel can take the bar
        if constexpr (!AF64) parallel_ok = lean && sorted_all != nullptr;
        // could a sweep with quantum 2^q certify?  (whole units below 2^32 in all: otherwise not even the per-key test can pass)
        auto hopeful = [&](const FpStats &t, int q) {
            return t.lbmin != (int)0x80000000 && q >= -149 && q <= 100 && q != wq && t.atot < ldexp(1.0, 32 + q);
        };
        if (!force_ordered && wq != FP_Q_UNKNOWN) done = attempt(wq);
        if (!done && !force_ordered && wq == FP_Q_UNKNOWN && no_quantum < 4) {
            // No quantum from the previous bar (a workgroup's first bar: EVERY bar of a tape of daily bars -- 580 bars on 512 workgroups): a
            // guess from the bar's first 4 096 ticks instead of a statistics sweep over all of it.  The attempt itself is what certifies,
            // so sizes that would certify take ONE order-free sweep (round 6: daily bars with dyadic sizes 13.7 -> 12.0 ms per 1e9 ticks).
            const int64_t sm = e - s < 4096 ? e - s : 4096;
            const int64_t a_lo = s + (int64_t)w * 256 < s + sm ? s + (int64_t)w * 256 : s + sm;
            const int64_t a_hi = a_lo + 256 < s + sm ? a_lo + 256 : s + sm;
            FpStats t = combine(fp_stats_lean<AF64>(price, amount, side, a_lo, a_hi, low, L, tick, inv_tick, lane));
            t.atot *= (double)(e - s) / (double)sm;
            const int q2 = t.lbmin == FP_Q_UNKNOWN ? 0 : t.lbmin;
            if (hopeful(t, q2)) { done = attempt(q2); q_used = q2; }
            if (!done) ++no_quantum;
        }
        // the counts per (segment, key) of the tick-ordered path and the statistics that pick a quantum come from ONE sweep
        const int K = 2 * L;
        // the counter arrays of the segments: nseg - 2 in the LDS behind the histogram, then the aux and the vol areas of the
        // histogram itself (both idle until the key sums are written; an integer attempt in between overwrites vol: recount)
        const int sstride = 2 * lmax;
        auto seg_arr = [&](int g) -> int * {
            return g < nseg - 2 ? (int *)(smem + (size_t)lmax * 24 + 256) + (size_t)g * sstride : (g == nseg - 2 ? (int *)aux : (int *)vol);
        };
        int64_t gseg = (e - s + nseg - 1) / nseg;
        gseg = (gseg + 63) & ~(int64_t)63;
        if (!done && parallel_ok) {
            if constexpr (!AF64) {
                auto count_sweep = [&]() -> FpStats {
                    __syncthreads();
                    for (int g = 0; g < nseg; ++g) {
                        int *sc = seg_arr(g);
                        for (int k = (int)threadIdx.x; k < K; k += 64 * FPW_WAVES) sc[k] = 0;
                    }
                    __syncthreads();
                    // wave w takes a sixteenth of the ticks; where its range straddles two segments it counts the parts separately
                    FpStats mine;
                    mine.lbmin = FP_Q_UNKNOWN; mine.atot = 0.0; mine.units_ok = true; mine.bad = false;
                    for (int g = 0; g < nseg; ++g) {
                        const int64_t g_lo = s + (int64_t)g * gseg, g_hi = g_lo + gseg < e ? g_lo + gseg : e;     // (g_lo, g_hi]
                        const int64_t lo = s_w > g_lo ? s_w : g_lo, hi = e_w < g_hi ? e_w : g_hi;
                        if (hi > lo)
                            mine.bad |= fp_count_lean(price, (const float *)amount, side, lo, hi, low, L, tick, inv_tick, lane, seg_arr(g),
                                                      mine.lbmin, mine.atot);
                    }
                    mine.lbmin = fmk_dpp_reduce(mine.lbmin, FP_Q_UNKNOWN, FmkOpMin());
                    mine.atot = fmk_dpp_reduce(mine.atot, 0.0, FmkOpAdd());
                    return combine(mine);
                };
                const FpStats t = count_sweep();
                bad_level = t.bad;
                const int q2 = t.lbmin == FP_Q_UNKNOWN ? 0 : t.lbmin;      // only zeros: any quantum
                if (!force_ordered && hopeful(t, q2)) {
                    done = attempt(q2);
                    q_used = q2;
                    if (!done) bad_level = count_sweep().bad;              // (the attempt used vol: the last segment's counters)
                }
            }
        } else if (!done && !force_ordered) {
            const FpStats t = combine(fp_stats_lean<AF64>(price, amount, side, s_w, e_w, low, L, tick, inv_tick, lane));
            const int q2 = t.lbmin == FP_Q_UNKNOWN ? 0 : t.lbmin;
            if (hopeful(t, q2)) { done = attempt(q2); q_used = q2; }
        }
        if (done) {
            wq = q_used;
            no_quantum = 0;
            for (int k = (int)threadIdx.x; k < 2 * L; k += 64 * FPW_WAVES) vol[k] = ldexpf((float)units[k], q_used);
        } else {
            // Tick order: the float32 level sums round on every add (base.py:713-717) -- what real sizes (full float32 mantissas) always
            // take.  One wave walking the bar chunk by chunk (fp_accumulate_lean) needs ~2.7 us per 64 ticks: 73 ms per 1e9 ticks of
            // daily bars, and inside this kernel it left fifteen waves idle.  Instead the bar's amounts are SORTED by (key, tick) --
            // counts per key by all waves, an exclusive scan, a stable scatter by one wave whose per-chunk cost is one LDS atomic and
            // one store (fp_scatter_lean) -- and every key's float32 sum is then a plain sequential loop of ONE lane over its own
            // contiguous slice: all keys in parallel, each in the reference's order.
            wq = FP_Q_UNKNOWN;
            if (!parallel_ok) {
                // float64 amounts (f32 element += f64 value) / no lane-ordered LDS atomics: the wave-per-bar kernel takes the bar
                if (threadIdx.x == 0) defer[32 + atomicAdd(defer, 1ULL)] = (unsigned long long)b;
                continue;
            }
            if constexpr (!AF64) {
                // The scatter is the serial part, so the bar is cut into `nseg` SEGMENTS of ticks (as many as the LDS left by the
                // histogram holds counter arrays for: 16 up to ~600 levels, 1 beyond ~5 000) that scatter concurrently, one wave each:
                // segment g's slots of a key start where the segments before it end -- counts per (segment, key) first.
                __syncthreads();
                // totals per key -> cnt[]; their exclusive scan; cursors per (segment, key).  Thread t owns the keys t * per .. + per - 1
                const int per = (K + 64 * FPW_WAVES - 1) / (64 * FPW_WAVES);
                const int k0 = (int)threadIdx.x * per;
                int tot = 0;
                for (int k = k0; k < k0 + per && k < K; ++k) {
                    int c = 0;
                    for (int g = 0; g < nseg; ++g) c += seg_arr(g)[k];
                    cnt[k] = c;
                    tot += c;
                }
                int inc = tot;
#pragma unroll
                for (int o2 = 1; o2 < 64; o2 <<= 1) { const int v = __shfl_up(inc, o2, 64); if (lane >= o2) inc += v; }
                if (lane == 63) s_umax[w] = (unsigned)inc;
                __syncthreads();
                int base0 = inc - tot;
                for (int k = 0; k < w; ++k) base0 += (int)s_umax[k];
                for (int k = k0; k < k0 + per && k < K; ++k) {
                    for (int g = 0; g < nseg; ++g) { int *sc = seg_arr(g); const int c = sc[k]; sc[k] = base0; base0 += c; }
                }
                __syncthreads();
                float *sorted = sorted_all + (s + 1);
                if (w < nseg) {
                    const int64_t g_lo = s + (int64_t)w * gseg, g_hi = g_lo + gseg < e ? g_lo + gseg : e;
                    if (g_hi > g_lo)
                        fp_scatter_lean(price, (const float *)amount, side, g_lo, g_hi, low, L, tick, inv_tick, lane, seg_arr(w), sorted);
                }
                __syncthreads();
                // the last segment's cursor is now the END of the key's slice
                const int *cursor = seg_arr(nseg - 1);                   // (the vol area: thread k reads its end, then writes its sum there)
                for (int k = (int)threadIdx.x; k < K; k += 64 * FPW_WAVES) {
                    const int end = cursor[k], c = cnt[k];
                    const float *src = sorted + (end - c);
                    float v = 0.f;
                    int i = 0;
                    for (; i + 8 <= c; i += 8) {
                        const float x0 = src[i], x1 = src[i + 1], x2 = src[i + 2], x3 = src[i + 3], x4 = src[i + 4], x5 = src[i + 5],
                                    x6 = src[i + 6], x7 = src[i + 7];
                        v += x0; v += x1; v += x2; v += x3; v += x4; v += x5; v += x6; v += x7;
                    }
                    for (; i < c; ++i) v += src[i];
                    vol[k] = v;
                }
            }
        }
        __syncthreads();
        if (w == 0) {
            if (bad_level && lane == 0 && n_bad) atomicAdd(n_bad, 1ULL);
            fp_emit_bar(o, b, base, L, low, lmax, imb_mult, lane, vol, cnt, aux, stk, lmax >= 512);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// ONE LANE PER BAR (round 2, float32 amounts): footprints of streams of very short bars (1-second bars: ~20 ticks, 2-4
// price levels).  The wave-per-bar kernel above spends a whole wave -- tile loads, histogram reset, the level passes of
// fp_emit_bar with their wave scans -- on 20 ticks: 53.8 ms per 1e9 ticks of 1-second bars against 3.5 ms for 1-minute bars.
// Here a wave takes 64 consecutive bars; lane l runs the reference's loops for bar l (base.py:700-719 over its ticks,
// comp_footprint_features over its <= FL_MAXL levels) on a private histogram in its registers, so the float32 level sums are
// added in tick order by construction and the level statistics are plain sequential code: NumPy's pairwise float32 sum degenerates to its n < 8 loop / its one 8-accumulator fold.  Ticks arrive like in
// k_bar_dir_lanes (fmk_barflow.hip): per step the wave stages each lane's 16-tick aligned block, 16 / 8 / 4 coalesced load
// instructions into registers one step ahead, LDS rows padded 16 -> 17.
// Bars with more than FL_MAXL levels or more than `max_len` ticks go on a list for the wave-per-bar kernel (list mode).
// ---------------------------------------------------------------------------------------
#define FL_MAXL 8
#define FL_T 16
#define FL_ROW 17
#define FL_WAVES 2
__device__ __forceinline__ float fl_pairwise(const float (&a)[FL_MAXL], int n)      // np.sum of n <= 8 float32
{
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

__global__ __launch_bounds__(64 * FL_WAVES) void k_bar_footprints_lanes(const double *__restrict__ price,
                                                                       const float *__restrict__ amount,
                                                                       const int8_t *__restrict__ side,
                                                                       const int64_t *__restrict__ ci, int64_t nb, double tick,
                                                                       const double *__restrict__ lows, double imb_mult,
                                                                       const int64_t *__restrict__ off, FpOut o,
                                                                       unsigned long long *n_bad,
                                                                       unsigned long long *__restrict__ grp_mask,
                                                                       int64_t *__restrict__ grp_cnt, int64_t max_len)
{
    __shared__ double s_p[FL_WAVES][64 * FL_ROW];
    __shared__ float s_a[FL_WAVES][64 * FL_ROW];
    __shared__ uint32_t s_s[FL_WAVES][64 * 5];
    __shared__ int64_t s_blk[FL_WAVES][64];
    const int lane = fmk_lane();
    const int wib = fmk_uniform((int)(threadIdx.x >> 6));
    double *sP = s_p[wib];
    float *sA = s_a[wib];
    uint32_t *sS = s_s[wib];
    const signed char *sS8 = (const signed char *)sS;
    int64_t *sB = s_blk[wib];
    const int64_t nwaves = (int64_t)gridDim.x * FL_WAVES;
    const double inv_tick = 1.0 / tick;
    const int64_t last_tick = ci[nb];                                   // no bar reaches beyond it: the bound of every load
    for (int64_t w = (int64_t)blockIdx.x * FL_WAVES + wib; w * 64 < nb; w += nwaves) {
        const int64_t b = w * 64 + lane;
        const bool has = b < nb;
        const int64_t s = has ? ci[b] : 0, e = has ? ci[b + 1] : 0;
        const int64_t base = has ? off[b] : 0;
        const int64_t L64 = has ? off[b + 1] - base : 0;
        const int64_t len = e - s;
        const bool mine = has && L64 >= 1 && L64 <= FL_MAXL && len <= max_len;
        const bool other = has && L64 >= 1 && !mine;                    // L == 0: no rows, nothing written (as the wave kernel)
        // which of the 64 bars stay for the wave-per-bar kernel: one mask and one count per group, compacted into a list by a
        // scan afterwards (k_fl_compact) -- an atomicAdd on one list counter per wave was 7.8e5 same-address atomics for 5e7 bars
        const unsigned long long ob = __builtin_amdgcn_ballot_w64(other);
        if (lane == 0) { grp_mask[w] = ob; grp_cnt[w] = __builtin_popcountll(ob); }
        const int L = mine ? (int)L64 : 0;
        const int ilow = mine ? (int)fp_level(lows[b], tick) : 0;
        // the lane's histogram lives in REGISTERS (slot = 2 * level + side; 16 float32 sums, 16 counts): a tick updates it
        // with 16 compare / select / add triples.  An LDS histogram ([slot][lane], read-modify-write per tick) was built
        // first: two dependent LDS round trips per tick at 6 waves per CU -- 23 ms per 1e9 ticks of 1-second bars.
        float hv[2 * FL_MAXL];
        int hc[2 * FL_MAXL];
#pragma unroll
        for (int k = 0; k < 2 * FL_MAXL; ++k) { hv[k] = 0.f; hc[k] = 0; }
        const bool active = mine && len > 0;
        const int64_t start = s + 1;
        const int64_t first_blk = start >> 4;
        const int nsteps = active ? (int)((e >> 4) - first_blk + 1) : 0;
        const int steps = fmk_dpp_reduce(nsteps, 0, FmkOpMax());
        bool bad = false;
        double pr[16];
        float2 ar[8];
        uint32_t sr[4];
        auto issue = [&](int step) {                                   // the blocks of `step` -> registers
            sB[lane] = step < nsteps ? first_blk + step : (int64_t)-1;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int64_t rb = sB[4 * k + (lane >> 4)];
                int64_t idx = rb * FL_T + (lane & 15);
                idx = idx <= last_tick ? idx : last_tick;
                pr[k] = rb >= 0 ? price[idx] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int64_t rb = sB[8 * k + (lane >> 3)];
                const int64_t idx = rb * FL_T + (lane & 7) * 2;
                float2 v = make_float2(0.f, 0.f);
                if (rb >= 0) {
                    if (idx + 1 <= last_tick) v = *(const float2 *)(amount + idx);
                    else if (idx <= last_tick) v.x = amount[idx];
                }
                ar[k] = v;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t rb = sB[16 * k + (lane >> 2)];
                const int64_t idx = rb * FL_T + (lane & 3) * 4;
                uint32_t v = 0;
                if (rb >= 0) {
                    if (idx + 3 <= last_tick) v = *(const uint32_t *)(side + idx);
                    else
                        for (int q = 0; q < 4; ++q)
                            if (idx + q <= last_tick) v |= (uint32_t)(uint8_t)side[idx + q] << (8 * q);
                }
                sr[k] = v;
            }
            __builtin_amdgcn_wave_barrier();
        };
        if (steps > 0) issue(0);
        for (int step = 0; step < steps; ++step) {
#pragma unroll
            for (int k = 0; k < 16; ++k) sP[(4 * k + (lane >> 4)) * FL_ROW + (lane & 15)] = pr[k];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int at = (8 * k + (lane >> 3)) * FL_ROW + (lane & 7) * 2;
                sA[at] = ar[k].x; sA[at + 1] = ar[k].y;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) sS[(16 * k + (lane >> 2)) * 5 + (lane & 3)] = sr[k];
            __builtin_amdgcn_wave_barrier();
            if (step + 1 < steps) issue(step + 1);
            if (step < nsteps) {                                       // base.py:700-719 over the lane's ticks of this block
                const int64_t blk0 = (first_blk + step) * FL_T;
                const int lo = start > blk0 ? (int)(start - blk0) : 0;
                const int hi = e - blk0 < 15 ? (int)(e - blk0) : 15;
                const double *rowP = sP + lane * FL_ROW;
                const float *rowA = sA + lane * FL_ROW;
                const signed char *rowS = sS8 + lane * 20;
#pragma unroll 4
                for (int j = 0; j < FL_T; ++j) {
                    const bool valid = j >= lo && j <= hi;
                    const double p = rowP[j];
                    const int sd = (int)rowS[j];
                    const float a = rowA[j];
                    const int lvl = fp_level32(p, tick, inv_tick) - ilow;
                    const bool inside = (unsigned)lvl < (unsigned)L;
                    bad |= valid && !inside;                               // base.py:719
                    const int slot = (valid && inside && (sd == 1 || sd == -1)) ? 2 * lvl + (sd < 0 ? 1 : 0) : -1;
#pragma unroll
                    for (int k = 0; k < 2 * FL_MAXL; ++k) {
                        hv[k] += slot == k ? a : 0.f;                      // float32 +=, in tick order (x + 0.0f == x)
                        hc[k] += slot == k ? 1 : 0;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        const unsigned long long bb = __builtin_amdgcn_ballot_w64(bad);
        if (bb && lane == 0 && n_bad) atomicAdd(n_bad, (unsigned long long)__builtin_popcountll(bb));
        if (!mine) continue;
        // ---- level rows + comp_footprint_features (base.py:755-850), sequentially over the lane's L <= 8 levels
        float bv[FL_MAXL], sv[FL_MAXL], tot[FL_MAXL];
        float best = -INFINITY;
        int best_i = 0;
        double num = 0.0;
#pragma unroll
        for (int l = 0; l < FL_MAXL; ++l) {
            bv[l] = 0.f; sv[l] = 0.f; tot[l] = 0.f;
            if (l < L) {
                bv[l] = hv[2 * l]; sv[l] = hv[2 * l + 1];
                o.price_levels[base + l] = (int32_t)(ilow + l);
                o.buy_volumes[base + l] = bv[l];
                o.sell_volumes[base + l] = sv[l];
                o.buy_ticks[base + l] = hc[2 * l];
                o.sell_ticks[base + l] = hc[2 * l + 1];
                tot[l] = bv[l] + sv[l];                                    // base.py:822
                if (tot[l] > best || (tot[l] != tot[l] && best == best)) { best = tot[l]; best_i = l; }   // np.argmax: the first maximum, a NaN being one
                num += (double)(ilow + l) * (double)tot[l];
            }
        }
        const float total = fl_pairwise(tot, L);
        const bool stats = total > 0.f;                                    // base.py:836 (L >= 1 here)
        const double vwap = stats ? num / (double)total : 0.0;
        unsigned bsum = 0, ssum = 0;
        double skew = 0.0;
        float q2[FL_MAXL];
        int max_run = 0, max_sign = 0, run = 0, run_sign = 0;
#pragma unroll
        for (int l = 0; l < FL_MAXL; ++l) {
            q2[l] = 0.f;
            if (l < L) {
                const bool si = l < L - 1 && (double)sv[l] > (double)bv[l + 1 < FL_MAXL ? l + 1 : l] * imb_mult;   // base.py:797
                const bool bi = l >= 1 && (double)bv[l] > (double)sv[l >= 1 ? l - 1 : 0] * imb_mult;              // base.py:798
                o.buy_imbalances[base + l] = bi;
                o.sell_imbalances[base + l] = si;
                bsum += bi ? 1u : 0u; ssum += si ? 1u : 0u;
                const int sg = bi ? 1 : (si ? -1 : 0);                     // base.py:801-819
                if (sg != 0 && sg == run_sign) run += 1;
                else if (sg != 0) { run = 1; run_sign = sg; }
                else { run = 0; run_sign = 0; }
                if (run > max_run) { max_run = run; max_sign = run_sign; }
                if (stats) {
                    skew += ((double)(ilow + l) - vwap) * (double)tot[l];
                    const float q = tot[l] / total;
                    q2[l] = q * q;
                }
            }
        }
        double gini = 0.0;
        if (stats) gini = (double)(1.0f - fl_pairwise(q2, L));             // base.py:847-848 (float32)
        o.buy_imbalances_sum[b] = (uint16_t)bsum;
        o.sell_imbalances_sum[b] = (uint16_t)ssum;
        o.cot_price_levels[b] = (int32_t)(ilow + best_i);
        o.imb_max_run_signed[b] = (int16_t)(max_run * max_sign);
        o.vp_skew[b] = stats ? skew / (double)total : 0.0;
        o.vp_gini[b] = gini;
    }
}

// list of the bars k_bar_footprints_lanes left over: rest[0] = count (from the scan's total), rest[32 + pos[g] + k] = the k-th
// set bit of group g
__global__ __launch_bounds__(256) void k_fl_compact(const unsigned long long *__restrict__ grp_mask,
                                                    const int64_t *__restrict__ pos, int64_t groups,
                                                    unsigned long long *__restrict__ rest)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g == 0) rest[0] = (unsigned long long)pos[groups];
    if (g >= groups) return;
    unsigned long long m = grp_mask[g];
    int64_t at = 32 + pos[g];
    while (m) {
        const int bit = __builtin_ctzll(m);
        rest[at++] = (unsigned long long)(g * 64 + bit);
        m &= m - 1;
    }
}

template <bool AF64>
static int fp_launch(fmk_ctx *ctx, const double *p, const void *a, const int8_t *sd, const int64_t *ci, int64_t nb,
                     double tick, const double *lows, double imb_mult, const int64_t *off, int lmin, int lmax, int wpb,
                     const FpOut &o, unsigned long long *n_bad, const unsigned long long *only = nullptr,
                     double *d_median = nullptr, int *saw_long = nullptr, int64_t skip_above = INT64_MAX, int skip_lmax = 0,
                     hipStream_t st = nullptr /* the context's stream when null */)
{
    if (!st) st = ctx->stream;
    const int force_ordered = 0;    // disables the exact (integer) path
    const bool med = d_median != nullptr && !AF64;
    const bool fast = !med && lmax >= 512 && lmax <= FP_MAX_LEVELS_LDS;          // the 16 B / level layout (k_bar_footprints<.., FAST>)
    // waves per workgroup.  A workgroup keeps its wave slots until its SLOWEST wave has finished its bars; on bars of unequal length
    // (lognormal one-minute bars) that idles the slots of the others, so streams of many bars take one-wave workgroups: the
    // dispatcher then balances per wave
    const int wpb_in = wpb;
    const size_t sort_bytes = AF64 ? 0 : (size_t)FP_SORT_BYTES;
    size_t smem = fast ? (size_t)wpb * ((size_t)lmax * 16 + FMK_PW_PAR_STK * 4 + sort_bytes)
                       : (size_t)wpb * ((size_t)lmax * 24 + 256 + (med ? (size_t)FP_MED_CAP * 4 : 0) + sort_bytes);
    int64_t blocks = fmk_ceil_div(nb, wpb);
    int64_t cap = (int64_t)ctx->n_cu * 64 * (wpb_in / wpb);                        // (the same number of waves in the grid)
    unsigned char *gscratch = nullptr;
    if (lmax > (med ? FP_MAX_LEVELS : FP_MAX_LEVELS_LDS)) {
        // wide bars: one wave per workgroup, histogram in global scratch (<= 8 GB in total, >= 16 waves)
        const size_t per_wave = (size_t)lmax * 24 + 256;
        cap = (int64_t)(((size_t)8 << 30) / per_wave);
        if (cap < 16) cap = 16;
        if (cap > (int64_t)ctx->n_cu * 32) cap = (int64_t)ctx->n_cu * 32;     // (every wave slot of the chip: 65 -> 47 ms per 2e8 ticks of 6 000-level bars against n_cu * 8)
        if (blocks > cap) blocks = cap;
        void *scr;
        st = ctx->stream;                                            // (the context scratch is ordered on the context's stream)
        FMK_TRY(fmk_scratch(ctx, per_wave * (size_t)blocks + 256, &scr));
        gscratch = (unsigned char *)scr;
        smem = 0;
    }
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if constexpr (!AF64) {
        if (med) {
            if (gscratch)
                k_bar_footprints<false, true, true><<<(unsigned)blocks, wpb * 64, 0, st>>>(
                    p, a, sd, ci, nb, tick, lows, imb_mult, off, lmin, lmax, o, n_bad, force_ordered, gscratch, 0, only, d_median,
                    saw_long, skip_above, skip_lmax);
            else {
                // the bracket lives in a wave from bar to bar and every wave's FIRST bar takes the generic selection: as many
                // workgroups as are resident at once (grid-stride over the bars), not 64 per CU
                static int occ[5] = {0, 0, 0, 0, 0};
                const int slot = lmax <= 128 ? 0 : (lmax <= 256 ? 1 : (lmax <= 512 ? 2 : (lmax <= 1024 ? 3 : 4)));
                if (!occ[slot]) {
                    int nblk = 0;
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, k_bar_footprints<false, false, true>, wpb * 64, smem) !=
                            hipSuccess || nblk < 1)
                        nblk = 2;
                    occ[slot] = nblk;
                }
                int64_t resident = (int64_t)ctx->n_cu * occ[slot];
                if (const char *v = getenv("FMK_FLOW_MEDIAN_DEFER")) {  // "1:blocks=N" (tests: few waves, many bars per wave)
                    const char *b = strstr(v, "blocks=");
                    if (b && atoi(b + 7) > 0) resident = atoi(b + 7);
                }
                if (blocks > resident) blocks = resident;
                k_bar_footprints<false, false, true><<<(unsigned)blocks, wpb * 64, smem, st>>>(
                    p, a, sd, ci, nb, tick, lows, imb_mult, off, lmin, lmax, o, n_bad, force_ordered, nullptr,
                    fp_lds_atomics_in_lane_order(ctx), only, d_median, saw_long, skip_above, skip_lmax);
            }
            FMK_LAUNCH_CHECK(ctx);
            return FMK_OK;
        }
    }
    if (gscratch)
        k_bar_footprints<AF64, true><<<(unsigned)blocks, wpb * 64, 0, st>>>(p, a, sd, ci, nb, tick, lows, imb_mult, off,
                                                                                   lmin, lmax, o, n_bad, force_ordered,
                                                                                   gscratch, 0, only, nullptr, nullptr, skip_above, skip_lmax);
    else if (fast) {
        if (smem > 48 * 1024)
            (void)hipFuncSetAttribute((const void *)k_bar_footprints<AF64, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        k_bar_footprints<AF64, false, false, true><<<(unsigned)blocks, wpb * 64, smem, st>>>(p, a, sd, ci, nb, tick, lows, imb_mult,
                                                                                       off, lmin, lmax, o, n_bad,
                                                                                       force_ordered, nullptr,
                                                                                       fp_lds_atomics_in_lane_order(ctx), only,
                                                                                       nullptr, nullptr, skip_above, skip_lmax);
    } else
        k_bar_footprints<AF64, false><<<(unsigned)blocks, wpb * 64, smem, st>>>(p, a, sd, ci, nb, tick, lows, imb_mult,
                                                                                       off, lmin, lmax, o, n_bad,
                                                                                       force_ordered, nullptr,
                                                                                       fp_lds_atomics_in_lane_order(ctx), only,
                                                                                       nullptr, nullptr, skip_above, skip_lmax);
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
    if (n_idx == 1) return FMK_OK;   // zero bars: nothing to fill
    if (n_idx < 1) return fmk_set_error(ctx, FMK_E_ARG, "negative dimensions are not allowed");
    if (n <= 0 || !d_side || !d_out) return fmk_set_error(ctx, FMK_E_ARG, "comp_bar_footprints: bad arguments");
    if (max_levels > FP_MAX_LEVELS_GLOBAL)
        return fmk_set_error(ctx, FMK_E_CAPACITY,
                             "comp_bar_footprints: a bar spans %lld price levels; this build supports <= %d per bar",
                             (long long)max_levels, FP_MAX_LEVELS_GLOBAL);
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    {   // a sizing call that took the one-pass kernel left the rows staged (fmk_fused.h)
        int handled = 0;
        FMK_TRY(fmk_fused_fill(ctx, d_price, d_amount, amount_is_f64, n, d_close_idx, n_idx, d_side, price_tick_size, d_bar_lows,
                               imbalance_factor, d_level_offsets, max_levels, d_out, d_n_bad_level, &handled));
        if (handled) return FMK_OK;
    }
    return fmk_footprints_fill_classes(ctx, d_price, d_amount, amount_is_f64, d_close_idx, n_idx - 1, d_side,
                                       price_tick_size, d_bar_lows, imbalance_factor, d_level_offsets, 0,
                                       max_levels, d_out, d_n_bad_level, n, nullptr);
}

int fmk_footprints_fill_classes(fmk_ctx *ctx, const double *d_price, const void *d_amount, int amount_is_f64,
                                const int64_t *d_close_idx, int64_t nb, const int8_t *d_side, double price_tick_size,
                                const double *d_bar_lows, double imb_mult, const int64_t *d_level_offsets, int lmin_start,
                                int64_t max_levels, const fmk_footprint_out *d_out, int64_t *d_n_bad_level, int64_t n_ticks,
                                double *d_median, const unsigned long long *only_list)
{
    // imb_mult stays float64: array(float32) * float64 is float64 under Numba typing (the production path); NumPy 2 / NEP 50
    // would round the product to float32 -- they differ only for inexact products (decimal lots), see oracle/fmk_oracle.c
    FpOut o;
    memcpy(&o, d_out, sizeof(o));
    unsigned long long *bad = (unsigned long long *)d_n_bad_level;
    // (the 256-level class: on bars of unequal length the longer ones -- more levels -- ran in the 512 class at 50 KB of LDS per
    // workgroup, three workgroups per CU: profiles/r03_real_bar_lengths.txt)
    // (the 1 024-level class, two waves per workgroup: bars of 513 .. 1 024 levels ran with ONE wave per workgroup at 49 KB of LDS,
    // three waves per CU: tools/widebench.py)
    // ... and classes of 4 096 / 6 144 levels on LDS (16 B per level: 67 / 100 KB per wave) instead of a histogram in global scratch
    // (half-step classes from 512 levels: a bar of 1 100 levels in the 1 536 class leaves room for six waves per CU, in the 2 048 class for four)
    constexpr int NCLS = 10;
    const bool lds_wide = !d_median;                                    // (the in-sweep median keeps the 24 B layout: up to 2 048 levels)
    const int LMAX[NCLS] = {128, 256, 512, lds_wide ? 768 : 1024, 1024, lds_wide ? 1536 : FP_MAX_LEVELS, FP_MAX_LEVELS,
                            lds_wide ? 3072 : FP_MAX_LEVELS, lds_wide ? FP_MAX_LEVELS_LDS : FP_MAX_LEVELS,
                            (int)max_levels};                           // last class: global-scratch histogram
    static const int WPB[NCLS] = {4, 4, 4, 2, 2, 1, 1, 1, 1, 1};
    if (only_list) {
        // the bars a caller lists (cfg 4's one-pass kernel hands over the bars outside its class): the wave-per-bar classes in list mode
        int lmin_l = 0, rc_l = FMK_OK;
        for (int k = 0; k < NCLS && rc_l == FMK_OK; ++k) {
            if (k > 0 && max_levels <= LMAX[k - 1]) break;
            const int lm = (k >= 3 && max_levels < LMAX[k]) ? (int)max_levels : LMAX[k];
            if (LMAX[k] > lmin_l)
                rc_l = amount_is_f64
                           ? fp_launch<true>(ctx, d_price, d_amount, d_side, d_close_idx, nb, price_tick_size, d_bar_lows, imb_mult,
                                             d_level_offsets, lmin_l, lm, WPB[k], o, bad, only_list)
                           : fp_launch<false>(ctx, d_price, d_amount, d_side, d_close_idx, nb, price_tick_size, d_bar_lows, imb_mult,
                                              d_level_offsets, lmin_l, lm, WPB[k], o, bad, only_list);
            lmin_l = LMAX[k];
        }
        return rc_l;
    }
    // Very short bars (1-second bars and the like): one lane per bar first, the wave-per-bar classes below then only see the
    // bars it listed (more than FL_MAXL levels, or long).  Developer knob FMK_FP_LANES: 0 never, 2 whenever the layout allows.
    // Measured at 1e9 ticks (profiles/r02_fp_lanes.txt).
    unsigned long long *rest = nullptr;
    {
        const char *lv = getenv("FMK_FP_LANES");
        const int mode = lv ? atoi(lv) : 1;
        const bool ok = !amount_is_f64 && lmin_start == 0 && ((uintptr_t)d_amount & 7) == 0 && ((uintptr_t)d_side & 3) == 0;
        const bool fit = n_ticks > 0 && nb >= (int64_t)ctx->n_cu * 64 * 4 && n_ticks / nb <= 64;
        if (d_median && ok && mode != 0 && (fit || mode == 2)) {
            // the lane-per-bar schedule does not carry the median: the amounts-only pass serves it, the classes below then run
            // without it
            FMK_TRY(fmk_median_small_launch(ctx, (const float *)d_amount, d_close_idx, nb, d_median, n_ticks));
            d_median = nullptr;
        }
        if (ok && mode != 0 && (fit || mode == 2)) {
            const int64_t groups = fmk_ceil_div(nb, 64);
            unsigned long long *grp_mask = nullptr;
            int64_t *grp_cnt = nullptr;
            int arc = fmk_alloc(ctx, (size_t)(nb + 32) * 8, (void **)&rest);
            if (arc == FMK_OK) arc = fmk_alloc(ctx, (size_t)groups * 8, (void **)&grp_mask);
            if (arc == FMK_OK) arc = fmk_alloc(ctx, (size_t)(groups + 1) * 8, (void **)&grp_cnt);
            if (arc != FMK_OK) {
                if (rest) (void)fmk_free(ctx, rest);
                if (grp_mask) (void)fmk_free(ctx, grp_mask);
                return arc;
            }
            int64_t blocks = fmk_ceil_div(groups, FL_WAVES);
            const int64_t cap = (int64_t)ctx->n_cu * 32;
            if (blocks > cap) blocks = cap;
            k_bar_footprints_lanes<<<(unsigned)blocks, 64 * FL_WAVES, 0, ctx->stream>>>(
                d_price, (const float *)d_amount, d_side, d_close_idx, nb, price_tick_size, d_bar_lows, imb_mult, d_level_offsets, o,
                bad, grp_mask, grp_cnt, 4096);
            FMK_LAUNCH_CHECK(ctx);
            arc = fmk_exclusive_scan_i64(ctx, grp_cnt, grp_cnt, groups, true);
            if (arc == FMK_OK) k_fl_compact<<<(unsigned)fmk_ceil_div(groups, 256), 256, 0, ctx->stream>>>(grp_mask, grp_cnt, groups, rest);
            (void)fmk_free(ctx, grp_mask);
            (void)fmk_free(ctx, grp_cnt);
            if (arc != FMK_OK) { (void)fmk_free(ctx, rest); return arc; }
            FMK_LAUNCH_CHECK(ctx);
        }
    }
    int lmin = 0, rc = FMK_OK;
    int *saw_long = (int *)(ctx->d_mail + 16);
    if (d_median && amount_is_f64) {      // float64 amounts: the in-sweep median is a float32 schedule
        if (rest) (void)fmk_free(ctx, rest);
        return fmk_set_error(ctx, FMK_E_ARG, "footprints with the median trade size: float32 amounts only");
    }
    if (d_median) {
        const hipError_t e = hipMemsetAsync(saw_long, 0, 2 * sizeof(int), ctx->stream);   // [0] long bars seen, [1] generic selections
        if (e != hipSuccess) { if (rest) (void)fmk_free(ctx, rest); FMK_HIP(ctx, e); }
    }
    // bars of more than FPW_MIN ticks (and at most FP_MAX_LEVELS levels): a workgroup per bar, from a list; the wave-per-bar classes
    // below skip them
    int64_t skip_above = INT64_MAX;
    int skip_lmax = 0;
    float *wide_sorted = nullptr;
    unsigned long long *wide_defer = nullptr;
    {
        if (lmin_start == 0 && n_ticks > FPW_MIN) {
            int64_t *wl = nullptr;
            rc = fmk_long_bar_list(ctx, d_close_idx, nb, n_ticks, FPW_MIN, nullptr, &wl);
            // scratch of the tick-ordered path: the bars' amounts sorted by (key, tick), in the slots of the bar's own tick range;
            // the list of the bars handed back to the wave-per-bar kernel (at most n_ticks / FPW_MIN of them)
            if (rc == FMK_OK && !amount_is_f64) (void)fmk_alloc(ctx, (size_t)(n_ticks + 64) * 4, (void **)&wide_sorted);   // (none: those bars are deferred)
            if (rc == FMK_OK) rc = fmk_alloc(ctx, (size_t)(n_ticks / FPW_MIN + 2 + 32) * 8, (void **)&wide_defer);
            if (rc == FMK_OK) {
                const hipError_t me = hipMemsetAsync(wide_defer, 0, 8, ctx->stream);
                if (me != hipSuccess) rc = fmk_set_error(ctx, FMK_E_HIP, "hipMemsetAsync: %s", hipGetErrorString(me));
            }
            if (rc == FMK_OK) {
                // the widest bar of the call bounds the histogram (one workgroup per CU beyond ~3 000 levels, two below)
                const int wl_max = (int)(max_levels < 64 ? 64 : (max_levels > FPW_MAX_LEVELS ? FPW_MAX_LEVELS : max_levels));
                // ... plus the counter arrays of the tick-ordered path's segments (8 B per level and segment) in what is left of 158 KB
                const size_t hist = (size_t)wl_max * 24 + 256, avail = (size_t)158 * 1024;
                int nlds = (int)((avail - hist) / ((size_t)wl_max * 8));
                nlds = nlds > 14 ? 14 : nlds;
                const int nseg = nlds + 2;                         // ... plus the histogram's own aux and vol areas
                const size_t smem = hist + (size_t)nlds * wl_max * 8;
                if (smem > 48 * 1024) {
                    (void)hipFuncSetAttribute((const void *)k_bar_footprints_wide<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
                    (void)hipFuncSetAttribute((const void *)k_bar_footprints_wide<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
                }
                const unsigned grid = (unsigned)(ctx->n_cu * 2);
                const int lean = fp_lds_atomics_in_lane_order(ctx);
                if (amount_is_f64)
                    k_bar_footprints_wide<true><<<grid, 64 * FPW_WAVES, smem, ctx->stream>>>(
                        d_price, d_amount, d_side, d_close_idx, wl, price_tick_size, d_bar_lows, imb_mult, d_level_offsets, o, bad,
                        0, lean, wl_max, wide_sorted, wide_defer, nseg);
                else
                    k_bar_footprints_wide<false><<<grid, 64 * FPW_WAVES, smem, ctx->stream>>>(
                        d_price, d_amount, d_side, d_close_idx, wl, price_tick_size, d_bar_lows, imb_mult, d_level_offsets, o, bad,
                        0, lean, wl_max, wide_sorted, wide_defer, nseg);
                const hipError_t le = hipGetLastError();
                (void)fmk_free(ctx, wl);
                wl = nullptr;
                if (le != hipSuccess) {
                    if (rest) (void)fmk_free(ctx, rest);
                    if (wide_sorted) (void)fmk_free(ctx, wide_sorted);
                    if (wide_defer) (void)fmk_free(ctx, wide_defer);
                    FMK_HIP(ctx, le);
                }
                skip_above = FPW_MIN;
                skip_lmax = wl_max;
            }
            if (wl) (void)fmk_free(ctx, wl);                       // (an allocation above failed: rc says so)
        }
    }
    // The classes take disjoint bars, and on a tape whose bars differ in length the first class (<= 128 levels) does nearly all the
    // work while each wider one walks a few long bars with little parallelism (lognormal one-minute bars: 5.4 ms + six launches of
    // 0.15 .. 0.36 ms one after the other).  So the wider LDS classes run BESIDE the first one, on the context's auxiliary stream
    // (round 4).
    hipStream_t side = nullptr;
    {
        if (max_levels > LMAX[0] && lmin_start == 0 && fmk_ctx_aux(ctx) == FMK_OK) {
            side = ctx->aux;
            hipError_t e = hipEventRecord(ctx->aev[3], ctx->stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(side, ctx->aev[3], 0);
            if (e != hipSuccess) side = nullptr;
        }
    }
    for (int k = 0; k < NCLS && rc == FMK_OK; ++k) {
        if (k > 0 && max_levels <= LMAX[k - 1]) break;
        // (the widest class of a call needs no more LDS than the call's widest bar)
        const int lm = (k >= 3 && max_levels < LMAX[k]) ? (int)max_levels : LMAX[k];
        if (LMAX[k] > lmin_start && LMAX[k] > lmin) {
            hipStream_t st = (k > 0 && side) ? side : nullptr;
            rc = amount_is_f64
                     ? fp_launch<true>(ctx, d_price, d_amount, d_side, d_close_idx, nb, price_tick_size, d_bar_lows,
                                       imb_mult, d_level_offsets, lmin, lm, WPB[k], o, bad, rest, nullptr, nullptr, skip_above, skip_lmax, st)
                     : fp_launch<false>(ctx, d_price, d_amount, d_side, d_close_idx, nb, price_tick_size, d_bar_lows,
                                        imb_mult, d_level_offsets, lmin, lm, WPB[k], o, bad, rest, d_median, saw_long, skip_above, skip_lmax, st);
        }
        lmin = LMAX[k];
    }
    if (side) {                                                     // join: everything behind this point sees every class' results
        hipError_t e = hipEventRecord(ctx->aev[3], side);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->aev[3], 0);
        if (e != hipSuccess && rc == FMK_OK) rc = fmk_set_error(ctx, FMK_E_HIP, "footprints: joining the side stream: %s", hipGetErrorString(e));
    }
    // the long bars the workgroup kernel handed back (float64 amounts in tick order): the same classes once more, in list mode
    if (wide_defer && rc == FMK_OK) {
        lmin = 0;
        for (int k = 0; k < NCLS && rc == FMK_OK; ++k) {
            if (k > 0 && max_levels <= LMAX[k - 1]) break;
            const int lm = (k >= 3 && max_levels < LMAX[k]) ? (int)max_levels : LMAX[k];
            if (LMAX[k] > lmin)
                rc = amount_is_f64
                         ? fp_launch<true>(ctx, d_price, d_amount, d_side, d_close_idx, nb, price_tick_size, d_bar_lows, imb_mult,
                                           d_level_offsets, lmin, lm, WPB[k], o, bad, wide_defer)
                         : fp_launch<false>(ctx, d_price, d_amount, d_side, d_close_idx, nb, price_tick_size, d_bar_lows, imb_mult,
                                            d_level_offsets, lmin, lm, WPB[k], o, bad, wide_defer);
            lmin = LMAX[k];
        }
    }
    if (wide_sorted) (void)fmk_free(ctx, wide_sorted);
    if (wide_defer) (void)fmk_free(ctx, wide_defer);
    if (rest) (void)fmk_free(ctx, rest);                               // stream-ordered: the launches above are queued before it
    // bars of more than FP_MED_MAX_TICKS ticks: the long-bar median kernels, when a sweep flagged one
    if (rc == FMK_OK && d_median) rc = fmk_median_launch(ctx, d_amount, 0, d_close_idx, nb, FP_MED_MAX_TICKS, saw_long, d_median, n_ticks);
    return rc;
}

extern "C" int fmk_comp_bar_footprints_fill_median_dev(fmk_ctx *ctx, const double *d_price, const void *d_amount,
                                                       int amount_is_f64, int64_t n, const int64_t *d_close_idx,
                                                       int64_t n_idx, const int8_t *d_side, double price_tick_size,
                                                       const double *d_bar_lows, double imbalance_factor,
                                                       const int64_t *d_level_offsets, int64_t max_levels,
                                                       const fmk_footprint_out *d_out, int64_t *d_n_bad_level,
                                                       double *d_median)
{
    if (n_idx == 1) return FMK_OK;   // zero bars: nothing to fill
    if (n_idx < 1) return fmk_set_error(ctx, FMK_E_ARG, "negative dimensions are not allowed");
    if (n <= 0 || !d_side || !d_out) return fmk_set_error(ctx, FMK_E_ARG, "comp_bar_footprints: bad arguments");
    if (max_levels > FP_MAX_LEVELS_GLOBAL)
        return fmk_set_error(ctx, FMK_E_CAPACITY,
                             "comp_bar_footprints: a bar spans %lld price levels; this build supports <= %d per bar",
                             (long long)max_levels, FP_MAX_LEVELS_GLOBAL);
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    {   // a sizing call that took the one-pass kernel left the rows staged (fmk_fused.h); its medians are done
        int handled = 0;
        FMK_TRY(fmk_fused_fill(ctx, d_price, d_amount, amount_is_f64, n, d_close_idx, n_idx, d_side, price_tick_size, d_bar_lows,
                               imbalance_factor, d_level_offsets, max_levels, d_out, d_n_bad_level, &handled));
        if (handled) return FMK_OK;
    }
    return fmk_footprints_fill_classes(ctx, d_price, d_amount, amount_is_f64, d_close_idx, n_idx - 1, d_side,
                                       price_tick_size, d_bar_lows, imbalance_factor, d_level_offsets, 0,
                                       max_levels, d_out, d_n_bad_level, n, d_median);
}

// diagnostics: how many bars of the last fmk_comp_bar_footprints_fill_median_dev call took the generic selection (bracket miss)
extern "C" int fmk_diag_fp_median_fallbacks(fmk_ctx *ctx, int64_t *count)
{
    int v[2] = {0, 0};
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    FMK_HIP(ctx, hipMemcpyAsync(v, ctx->d_mail + 16, sizeof v, hipMemcpyDeviceToHost, ctx->stream));
    FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *count = v[1];
    return FMK_OK;
}
