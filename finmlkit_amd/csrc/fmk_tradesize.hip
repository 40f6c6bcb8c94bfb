// fmk_tradesize.hip -- comp_bar_trade_size_features (finmlkit/bar/base.py:549-612) on gfx950.
// First "next" row of SURVEY.md 8(f): same bar segments, reuses the exact order-statistic search of
// fmk_median.h for np.percentile(amounts_bar, 95).
//
// One wave per bar, three passes over the bar's amounts (the 2nd/3rd hit L2; bars of <= 2048 ticks keep
// their keys in registers for the percentile):
//   pass 1  sum, block volume (amounts > theta*theta_mult)            -> mean_size_rel, pct_block
//   select  ranks floor(0.95(n-1)) and +1, NumPy 'linear' interpolation -> size_95_rel
//   pass 3  sum (a/total)^2                                            -> size_gini
// Arithmetic is float64 throughout and the results are rounded once to float32.  For float64 amounts
// this is the reference's arithmetic up to summation order; for float32 amounts the reference (both its
// NumPy and its Numba mode) accumulates the sums in float32, which this kernel deliberately does not
// reproduce -- the float64 sums are the more accurate value and differ by ~1e-7 relative (documented
// tolerance in tests/test_gpu_next.py).
#include <math.h>

#include "fmk_median.h"
#include "fmk_pairwise.h"

template <bool AF64, int NREG>
__device__ __forceinline__ double ts_percentile95(const void *amount, int64_t start, int64_t cnt, int lane,
                                                  typename MedKey<AF64>::K *buf, double thr = 0.0, double *block_out = nullptr)
{
    typedef MedKey<AF64> MK;
    MedBar<AF64, NREG, false> bar;
    bar.amount = amount; bar.start = start; bar.cnt = cnt; bar.lane = lane;
    bar.load_all();
    if constexpr (NREG > 0 && !AF64) {
        // float32 bars held in registers: the block volume (amounts above the threshold; a float64 sum of float32 values is
        // exact in any order) comes from the same registers -- no separate pass over the bar (2.8 of 9.4 ms per 1e9 ticks)
        if (block_out) {
            double bl = 0.0;
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                const double a = bar.key[r] != MK::MAXK ? MK::value(bar.key[r]) : 0.0;
                bl += a > thr ? a : 0.0;
            }
            *block_out = fmk_wave_sum(bl);
        }
    }
    // NumPy 'linear' method: virtual index (n-1)*q IN THE ARRAY'S DTYPE, neighbours floor and floor+1 (clipped), _lerp
    const double vidx = AF64 ? (double)(cnt - 1) * 0.95 : (double)((float)(cnt - 1) * (95.0f / 100.0f));
    const double fl = floor(vidx);
    const int64_t k1 = (int64_t)fl < cnt - 1 ? (int64_t)fl : cnt - 1;
    const int64_t k2 = k1 + 1 < cnt ? k1 + 1 : cnt - 1;
    typename MK::K v1, v2;
    if (!med_rank_pair<AF64, NREG, false>(bar, buf, k1, k2, v1, v2)) return NAN;
    const double a = MK::value(v1), b = MK::value(v2);
    if constexpr (!AF64) {
        // float32 amounts: NumPy 2.2 does EVERY step of np.percentile in the array's dtype -- q = 95 / float32(100), the
        // virtual index float32(n - 1) * q, the weight, the difference and both interpolation forms (oracle: orc_percentile_f32,
        // validated bit for bit against np.percentile).  The reference's slices are float32 after TradesData's merge.
        const float q32 = 95.0f / 100.0f;
        const float vi = (float)(cnt - 1) * q32;
        if (vi >= (float)(cnt - 1)) return b;            // k2 is clipped to the last element
        const float t32 = vi - floorf(vi);
        const float a32 = (float)a, b32 = (float)b, d32 = b32 - a32;
        float r32 = a32 + d32 * t32;
        if (t32 >= 0.5f) r32 = b32 - d32 * (1.0f - t32);
        return (double)r32;
    }
    const double t = vidx - fl;
    const double d = b - a;
    double r = a + d * t;
    if (t >= 0.5) r = b - d * (1.0 - t);
    if (d == 0.0) r = a;
    return r;
}

// np.percentile(., 95) of float32 bars beyond the register classes (more than 2048 ticks), by a workgroup per bar (med_block_select:
// radix select of the two ranks) in a pass of its own: the value is exactly a float32 (NumPy interpolates in the array's dtype),
// so it travels in o_p95[b] and k_bar_trade_size turns it into size_95_rel.  One wave bisecting the value range with a re-read of
// the bar per step took 35 / 57 / 383 ms per 1e9 ticks at 10-minute / hourly / daily bars.  Bars with irregular close indices
// (below -1 or beyond the array: Python slice semantics) stay with the old path.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_ts_p95_long(const float *__restrict__ amount, const int64_t *__restrict__ ci,
                                                         const int64_t *__restrict__ list, int64_t n, float *__restrict__ o_p95)
{
    typedef MedKey<false> MK;
    const int64_t n_list = list[0];
    for (int64_t q = blockIdx.x; q < n_list; q += gridDim.x) {
        const int64_t b = list[1 + q], s = ci[b], e = ci[b + 1];
        if (!(s >= -1 && e <= n - 1)) continue;
        const int64_t cnt = e - s, start = s + 1;
        const float q32 = 95.0f / 100.0f;
        const float vi = (float)(cnt - 1) * q32;             // the virtual index in the array's dtype (ts_percentile95)
        const double fl = floor((double)vi);
        const int64_t k1 = (int64_t)fl < cnt - 1 ? (int64_t)fl : cnt - 1;
        const int64_t k2 = k1 + 1 < cnt ? k1 + 1 : cnt - 1;
        MK::K v1, v2;
        bool any_nan;
        med_block_select<false, THREADS>(amount, start, cnt, k1, k2, v1, v2, any_nan);
        if (threadIdx.x == 0) {
            const float a32 = (float)MK::value(v1), b32 = (float)MK::value(v2);
            float r32;
            if (any_nan) r32 = NAN;
            else if (vi >= (float)(cnt - 1)) r32 = b32;
            else {
                const float t32 = vi - floorf(vi), d32 = b32 - a32;
                r32 = a32 + d32 * t32;
                if (t32 >= 0.5f) r32 = b32 - d32 * (1.0f - t32);
            }
            o_p95[b] = r32;
        }
    }
}

template <bool AF64>
__global__ __launch_bounds__(256) void k_bar_trade_size(const void *__restrict__ amount,
                                                        const double *__restrict__ theta,
                                                        const int64_t *__restrict__ ci, int64_t nb, int64_t n,
                                                        double theta_mult, float *__restrict__ o_mean, float *__restrict__ o_p95,
                                                        float *__restrict__ o_pct, float *__restrict__ o_gini, int p95_done)
{
    typedef typename MedKey<AF64>::K K;
    __shared__ K sbuf[4][64];
    __shared__ __attribute__((aligned(8))) int s_stk[4][FMK_PW_PAR_STK];      // scratch of the pairwise sums
    const int lane = fmk_lane();
    const int wib = fmk_uniform((int)(threadIdx.x >> 6));
    const int wpb = blockDim.x >> 6;
    const int64_t wave0 = (int64_t)blockIdx.x * wpb + wib;
    const int64_t nwaves = (int64_t)gridDim.x * wpb;
    K *buf = sbuf[wib];
    for (int64_t b = wave0; b < nb; b += nwaves) {
        const int64_t s = fmk_uniform(ci[b]);
        const int64_t e_raw = fmk_uniform(ci[b + 1]);
        // The reference takes the bar as a SLICE, amounts[start:end + 1] (base.py:590): an end index past the array is
        // clamped, not an error (its own test_block_volume passes end == len(amounts)).  The empty-bar guard
        // (start > end, base.py:584) looks at the raw indices; a slice left empty by the clamp gives mean([]) = NaN and
        // a zero total -> the same all-NaN row.
        // Python slice bounds: a negative start / stop wraps by n once and is then clamped to [0, n] -- a close index below -1
        // must not become a read in front of the column (it selects the reference's wrapped, usually empty, slice)
        int64_t start = s + 1, stop = e_raw + 1;
        start = start < 0 ? (start + n > 0 ? start + n : 0) : (start < n ? start : n);
        stop = stop < 0 ? (stop + n > 0 ? stop + n : 0) : (stop < n ? stop : n);
        const int64_t cnt = (e_raw - s > 0 && stop > start) ? stop - start : 0;
        float mean_rel = NAN, p95_rel = NAN, pct = NAN, gini = NAN;      // base.py:576-579
        const double th = theta[b];
        if (cnt > 0 && th != 0.0) {                                      // base.py:586-587
            const double thr = th * theta_mult;
            double sum = 0.0, block = 0.0;
            if constexpr (AF64) {
                // block_volume += amount runs in tick order in the reference (base.py:600-603) and float64 addition is not
                // associative: the (few) amounts above the threshold are added one by one, in order, by the whole wave
                for (int64_t j0 = 0; j0 < cnt; j0 += 64) {
                    const int64_t j = j0 + lane;
                    const double a = j < cnt ? fmk_amt<AF64>(amount, start + j) : 0.0;
                    sum += a;
                    uint64_t m = __ballot(j < cnt && a > thr);
                    while (m) {
                        const int l = __ffsll((unsigned long long)m) - 1;
                        m &= m - 1;
                        const int64_t bits = __double_as_longlong(a);
                        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, l);
                        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)((uint64_t)bits >> 32), l);
                        block += __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
                    }
                }
                sum = fmk_wave_sum(sum);
            } else if (!(cnt <= 64 * 32 && cnt <= FMK_PW_MAX_N)) {      // (else: `sum` is the pairwise total, `block` comes with the percentile)
                for (int64_t j = lane; j < cnt; j += 64) {
                    const double a = fmk_amt<AF64>(amount, start + j);
                    sum += a;
                    if (a > thr) block += a;                // float32 amounts: the float64 sum is exact in any order
                }
                sum = fmk_wave_sum(sum);
                block = fmk_wave_sum(block);
            }
            double mean = sum / (double)cnt;
            // float32 amounts (what TradesData's merge produces): the reference's np.mean / .sum() of the float32 slice are
            // NumPy PAIRWISE float32 sums and the mean divides in float32 (base.py:591-596 in NumPy semantics; oracle:
            // orc_pairwise_f32, 0 ulp against the reference's kit frames).  `sum` then is that rounded total.
            // float64 amounts: the same trees in float64 (np.mean / .sum() of a float64 slice are pairwise too).
            float tf = 0.f;
            const bool np_rule = cnt <= FMK_PW_BIG_MAX_N;   // longer bars than the explicit stack holds: tree-ordered sums
            const bool f32_rule = !AF64 && np_rule;
            if (np_rule) {
                if constexpr (!AF64) {
                    const float *af = (const float *)amount + start;
                    tf = fmk_pairwise_big([af](int i) { return af[i]; }, (int)cnt, lane, s_stk[wib]);
                    mean = (double)(tf / (float)cnt);
                    sum = (double)tf;
                } else {
                    const double *ad = (const double *)amount + start;
                    sum = fmk_pairwise_big([ad](int i) { return ad[i]; }, (int)cnt, lane, s_stk[wib]);
                    mean = sum / (double)cnt;
                }
            }
            mean_rel = (float)log1p(mean / thr);
            const int nreg = (int)((cnt + 63) >> 6);
            double p95;
            double *bo = (!AF64 && np_rule) ? &block : nullptr;
            if (cnt > 64 * 32) {
                if (!AF64 && p95_done && s >= -1 && e_raw <= n - 1) p95 = (double)o_p95[b];      // k_ts_p95_long
                else p95 = ts_percentile95<AF64, 0>(amount, start, cnt, lane, buf);
            }
            else if (nreg <= 4) p95 = ts_percentile95<AF64, 4>(amount, start, cnt, lane, buf, thr, bo);
            else if (nreg <= 12) p95 = ts_percentile95<AF64, 12>(amount, start, cnt, lane, buf, thr, bo);
            else if (nreg <= 20) p95 = ts_percentile95<AF64, 20>(amount, start, cnt, lane, buf, thr, bo);
            else if constexpr (!AF64) p95 = ts_percentile95<AF64, 32>(amount, start, cnt, lane, buf, thr, bo);
            else if (nreg <= 24) p95 = ts_percentile95<AF64, 24>(amount, start, cnt, lane, buf);
            else p95 = ts_percentile95<AF64, 0>(amount, start, cnt, lane, buf);
            p95_rel = (float)log1p(p95 / thr);
            if (sum != 0.0) {                                            // base.py:597-598
                pct = (float)(block / sum);
                if (cnt == 1) gini = 0.f;
                else if (f32_rule) {
                    // 1 - sum((a / total)^2) with float32 quotients, squares and pairwise sum (base.py:609)
                    const float *af = (const float *)amount + start;
                    const float t32 = tf;
                    gini = 1.0f - fmk_pairwise_big([af, t32](int i) { const float q = af[i] / t32; return q * q; }, (int)cnt, lane,
                                               s_stk[wib]);
                } else if (np_rule) {
                    const double *ad = (const double *)amount + start;
                    const double td = sum;
                    gini = (float)(1.0 - fmk_pairwise_big([ad, td](int i) { const double q = ad[i] / td; return q * q; }, (int)cnt, lane,
                                                       s_stk[wib]));
                } else {
                    double sq = 0.0;
                    for (int64_t j = lane; j < cnt; j += 64) {
                        const double q = fmk_amt<AF64>(amount, start + j) / sum;
                        sq += q * q;
                    }
                    gini = (float)(1.0 - fmk_wave_sum(sq));
                }
            }
        }
        if (lane == 0) { o_mean[b] = mean_rel; o_p95[b] = p95_rel; o_pct[b] = pct; o_gini[b] = gini; }
    }
}

extern "C" int fmk_comp_bar_trade_size_dev(fmk_ctx *ctx, const void *d_amount, int amount_is_f64, int64_t n,
                                           const double *d_theta, const int64_t *d_close_idx, int64_t n_idx,
                                           double theta_mult, float *d_mean_size_rel, float *d_size_95_rel,
                                           float *d_pct_block, float *d_size_gini)
{
    if (n_idx == 1) return FMK_OK;   // zero bars (base.py:549-612 checks theta's length only)
    if (n_idx < 1) return fmk_set_error(ctx, FMK_E_ARG, "negative dimensions are not allowed");
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t nb = n_idx - 1;
    int64_t blocks = fmk_ceil_div(nb, 4);
    const int64_t cap = (int64_t)ctx->n_cu * 64;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if (amount_is_f64) {
        k_bar_trade_size<true><<<(unsigned)blocks, 256, 0, ctx->stream>>>(d_amount, d_theta, d_close_idx, nb, n,
                                                                          theta_mult, d_mean_size_rel, d_size_95_rel,
                                                                          d_pct_block, d_size_gini, 0);
        FMK_LAUNCH_CHECK(ctx);
        return FMK_OK;
    }
    // float32 amounts: the percentile of the bars beyond the register classes first (256 threads per bar up to 8192 ticks, 1024 beyond)
    int64_t *list_mid = nullptr, *list_long = nullptr;
    FMK_TRY(fmk_long_bar_list(ctx, d_close_idx, nb, n, 64 * 32, nullptr, &list_mid, 8192));
    int rc = fmk_long_bar_list(ctx, d_close_idx, nb, n, 8192, nullptr, &list_long);
    if (rc == FMK_OK) {
        k_ts_p95_long<256><<<(unsigned)(ctx->n_cu * 8), 256, 0, ctx->stream>>>((const float *)d_amount, d_close_idx, list_mid, n,
                                                                            d_size_95_rel);
        k_ts_p95_long<1024><<<(unsigned)(ctx->n_cu * 2), 1024, 0, ctx->stream>>>((const float *)d_amount, d_close_idx, list_long, n,
                                                                             d_size_95_rel);
        k_bar_trade_size<false><<<(unsigned)blocks, 256, 0, ctx->stream>>>(d_amount, d_theta, d_close_idx, nb, n,
                                                                           theta_mult, d_mean_size_rel, d_size_95_rel,
                                                                           d_pct_block, d_size_gini, 1);
    }
    const hipError_t le = hipGetLastError();
    (void)fmk_free(ctx, list_mid);
    if (list_long) (void)fmk_free(ctx, list_long);
    FMK_TRY(rc);
    FMK_HIP(ctx, le);
    return FMK_OK;
}
