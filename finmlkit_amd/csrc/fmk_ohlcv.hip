// fmk_ohlcv.hip -- comp_bar_ohlcv (finmlkit/bar/base.py:306-407) on gfx950.
//
// Layout: price f64[N], amount f32|f64[N], close_idx i64[B+1] resident in HBM.
// One 64-lane wave owns one bar (bars are contiguous tick ranges, so the wave streams
// 512 B (price) + 256 B (amount) fully coalesced per load instruction, four independent
// loads in flight per lane), keeps hi/lo/sum(vol)/sum(price*vol) in registers and folds them
// with a 6-step xor butterfly when the bar ends.  No LDS, no atomics, no MFMA: the kernel is
// bounded by HBM read bandwidth, 12 B/tick (f32 amounts) + 60 B/bar written.
//
// The median trade size (base.py:403) is a second kernel, see fmk_median.hip.
//
// Floating point: price*volume is rounded before it is added (-ffp-contract=off), exactly
// like the reference; the per-bar float64 sums are accumulated lane-strided and then
// tree-reduced, i.e. in a different order than the reference's sequential loop
// (|rel. diff| ~ 1e-16, the north-star tolerance is 1e-9).
#include <math.h>

#include "fmk_common.h"

template <bool AF64>
__global__ __launch_bounds__(256) void k_bar_ohlcv(const double *__restrict__ price,
                                                   const void *__restrict__ amount,
                                                   const int64_t *__restrict__ ci, int64_t nb, int64_t n,
                                                   double *__restrict__ o_open, double *__restrict__ o_high,
                                                   double *__restrict__ o_low, double *__restrict__ o_close,
                                                   float *__restrict__ o_vol, double *__restrict__ o_vwap,
                                                   int64_t *__restrict__ o_trades)
{
    const int lane = fmk_lane();
    const int wpb = blockDim.x >> 6;
    const int64_t wave0 = (int64_t)blockIdx.x * wpb + fmk_uniform((int)(threadIdx.x >> 6));
    const int64_t nwaves = (int64_t)gridDim.x * wpb;
    for (int64_t b = wave0; b < nb; b += nwaves) {
        const int64_t s = fmk_uniform(ci[b]);
        const int64_t e = fmk_uniform(ci[b + 1]);
        if (e <= s) {   // empty bar (base.py:352-361): previous close, Python-style negative wrap
            if (lane == 0) {
                double p = price[fmk_wrap(e, n)];
                o_open[b] = p; o_high[b] = p; o_low[b] = p; o_close[b] = p;
                o_vol[b] = 0.f; o_vwap[b] = 0.0; o_trades[b] = 0;
            }
            continue;
        }
        const int64_t start = s + 1;
        double hi = -INFINITY, lo = INFINITY, tv = 0.0, td = 0.0;
        int64_t j = start + lane;
        for (; j + 192 <= e; j += 256) {
            double p0 = price[j], p1 = price[j + 64], p2 = price[j + 128], p3 = price[j + 192];
            double a0 = fmk_amt<AF64>(amount, j), a1 = fmk_amt<AF64>(amount, j + 64);
            double a2 = fmk_amt<AF64>(amount, j + 128), a3 = fmk_amt<AF64>(amount, j + 192);
            hi = fmax(fmax(hi, p0), fmax(p1, fmax(p2, p3)));
            lo = fmin(fmin(lo, p0), fmin(p1, fmin(p2, p3)));
            tv += a0; td += p0 * a0;
            tv += a1; td += p1 * a1;
            tv += a2; td += p2 * a2;
            tv += a3; td += p3 * a3;
        }
        for (; j <= e; j += 64) {
            double p0 = price[j];
            double a0 = fmk_amt<AF64>(amount, j);
            hi = fmax(hi, p0);
            lo = fmin(lo, p0);
            tv += a0; td += p0 * a0;
        }
        hi = fmk_wave_max(hi);
        lo = fmk_wave_min(lo);
        tv = fmk_wave_sum(tv);
        td = fmk_wave_sum(td);
        if (lane == 0) {
            o_open[b] = price[start];
            o_close[b] = price[e];
            o_high[b] = hi;
            o_low[b] = lo;
            o_vol[b] = (float)tv;
            o_vwap[b] = tv > 0.0 ? td / tv : 0.0;   // base.py:398
            o_trades[b] = e - s;
        }
    }
}

static unsigned ohlcv_grid(fmk_ctx *ctx, int64_t nb)
{
    int64_t blocks = fmk_ceil_div(nb, 4);
    int64_t cap = (int64_t)ctx->n_cu * 64;   // grid-stride beyond this
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

extern "C" int fmk_comp_bar_ohlcv_dev(fmk_ctx *ctx, const double *d_price, const void *d_amount,
                                      int amount_is_f64, int64_t n, const int64_t *d_close_idx, int64_t n_idx,
                                      double *d_open, double *d_high, double *d_low, double *d_close,
                                      float *d_volume, double *d_vwap, int64_t *d_trades, double *d_median)
{
    if (n_idx < 2)   // base.py:334-335
        return fmk_set_error(ctx, FMK_E_ARG, "Bar close indices must contain at least two elements.");
    if (n <= 0) return fmk_set_error(ctx, FMK_E_ARG, "comp_bar_ohlcv: empty price array");
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t nb = n_idx - 1;
    const unsigned grid = ohlcv_grid(ctx, nb);
    if (amount_is_f64)
        k_bar_ohlcv<true><<<grid, 256, 0, ctx->stream>>>(d_price, d_amount, d_close_idx, nb, n, d_open, d_high,
                                                        d_low, d_close, d_volume, d_vwap, d_trades);
    else
        k_bar_ohlcv<false><<<grid, 256, 0, ctx->stream>>>(d_price, d_amount, d_close_idx, nb, n, d_open, d_high,
                                                         d_low, d_close, d_volume, d_vwap, d_trades);
    FMK_LAUNCH_CHECK(ctx);
    if (d_median) return fmk_comp_bar_median_dev(ctx, d_amount, amount_is_f64, n, d_close_idx, n_idx, d_median);
    return FMK_OK;
}

