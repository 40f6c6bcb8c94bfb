// fmk_host.hip -- host-pointer flavours of the C ABI (the NumPy drop-in boundary).
// Each function uploads its inputs, runs the same device entry point the resident pipeline
// uses, downloads the outputs and synchronises.  No arithmetic happens here.
#include <vector>

#include "fmk_common.h"

namespace {

// RAII bag of device allocations tied to one call
struct DevBag {
    fmk_ctx *ctx;
    std::vector<void *> ptrs;
    explicit DevBag(fmk_ctx *c) : ctx(c) {}
    ~DevBag()
    {
        (void)hipStreamSynchronize(ctx->stream);
        for (void *p : ptrs) (void)fmk_free(ctx, p);      // back to the context's caching allocator
    }
    int alloc(size_t bytes, void **out)
    {
        int rc = fmk_alloc(ctx, bytes, out);
        if (rc == FMK_OK) ptrs.push_back(*out);
        return rc;
    }
    template <typename T>
    int up(const T *host, int64_t count, T **out)
    {
        void *p;
        FMK_TRY(alloc(sizeof(T) * (size_t)(count > 0 ? count : 1), &p));
        if (count > 0) FMK_TRY(fmk_h2d(ctx, p, host, sizeof(T) * (size_t)count));
        *out = (T *)p;
        return FMK_OK;
    }
    template <typename T>
    int out(int64_t count, T **o)
    {
        void *p;
        FMK_TRY(alloc(sizeof(T) * (size_t)(count > 0 ? count : 1), &p));
        *o = (T *)p;
        return FMK_OK;
    }
    int up_amount(const void *host, int is_f64, int64_t count, void **o)
    {
        size_t es = is_f64 ? 8 : 4;
        FMK_TRY(alloc(es * (size_t)(count > 0 ? count : 1), o));
        if (count > 0) FMK_TRY(fmk_h2d(ctx, *o, host, es * (size_t)count));
        return FMK_OK;
    }
};

template <typename T>
int down(fmk_ctx *ctx, T *host, const T *dev, int64_t count)
{
    if (!host || count <= 0) return FMK_OK;
    return fmk_d2h(ctx, host, dev, sizeof(T) * (size_t)count);
}

}  // namespace

extern "C" {

int fmk_time_bar_indexer(fmk_ctx *ctx, const int64_t *ts, int64_t n, double interval_seconds, int64_t *clock,
                         int64_t *close_idx, int64_t capacity, int64_t *n_edges)
{
    if (n <= 0) return fmk_set_error(ctx, FMK_E_ARG, "time_bar_indexer: empty timestamps");
    int64_t ne, e0, d;
    int rc = fmk_time_bar_clock(ts[0], ts[n - 1], interval_seconds, &ne, &e0, &d);
    if (rc) return fmk_set_error(ctx, rc, "%s", fmk_last_error(nullptr));
    *n_edges = ne;
    if (!clock || !close_idx) return FMK_OK;
    if (capacity < ne) return fmk_set_error(ctx, FMK_E_CAPACITY, "time_bar_indexer: capacity");
    DevBag bag(ctx);
    int64_t *d_ts, *d_clock, *d_idx;
    FMK_TRY(bag.up(ts, n, &d_ts));
    FMK_TRY(bag.out(ne, &d_clock));
    FMK_TRY(bag.out(ne, &d_idx));
    FMK_TRY(fmk_time_bar_indexer_dev(ctx, d_ts, n, e0, d, ne, d_clock, d_idx));
    FMK_TRY(down(ctx, clock, d_clock, ne));
    FMK_TRY(down(ctx, close_idx, d_idx, ne));
    return FMK_OK;
}

int fmk_comp_bar_ohlcv(fmk_ctx *ctx, const double *price, const void *amount, int amount_is_f64, int64_t n,
                       const int64_t *close_idx, int64_t n_idx, double *open_, double *high, double *low,
                       double *close_, float *volume, double *vwap, int64_t *trades, double *median)
{
    if (n_idx < 2)
        return fmk_set_error(ctx, FMK_E_ARG, "Bar close indices must contain at least two elements.");
    const int64_t nb = n_idx - 1;
    DevBag bag(ctx);
    double *d_p;
    void *d_a;
    int64_t *d_ci;
    FMK_TRY(bag.up(price, n, &d_p));
    FMK_TRY(bag.up_amount(amount, amount_is_f64, n, &d_a));
    FMK_TRY(bag.up(close_idx, n_idx, &d_ci));
    double *d_o, *d_h, *d_l, *d_c, *d_vw, *d_med = nullptr;
    float *d_v;
    int64_t *d_t;
    FMK_TRY(bag.out(nb, &d_o));
    FMK_TRY(bag.out(nb, &d_h));
    FMK_TRY(bag.out(nb, &d_l));
    FMK_TRY(bag.out(nb, &d_c));
    FMK_TRY(bag.out(nb, &d_v));
    FMK_TRY(bag.out(nb, &d_vw));
    FMK_TRY(bag.out(nb, &d_t));
    if (median) FMK_TRY(bag.out(nb, &d_med));
    FMK_TRY(fmk_comp_bar_ohlcv_dev(ctx, d_p, d_a, amount_is_f64, n, d_ci, n_idx, d_o, d_h, d_l, d_c, d_v, d_vw,
                                   d_t, d_med));
    FMK_TRY(down(ctx, open_, d_o, nb));
    FMK_TRY(down(ctx, high, d_h, nb));
    FMK_TRY(down(ctx, low, d_l, nb));
    FMK_TRY(down(ctx, close_, d_c, nb));
    FMK_TRY(down(ctx, volume, d_v, nb));
    FMK_TRY(down(ctx, vwap, d_vw, nb));
    FMK_TRY(down(ctx, trades, d_t, nb));
    FMK_TRY(down(ctx, median, d_med, nb));
    return FMK_OK;
}

}  // extern "C"

extern "C" {

int fmk_comp_bar_directional(fmk_ctx *ctx, const double *price, const void *amount, int amount_is_f64, int64_t n,
                             const int64_t *close_idx, int64_t n_idx, const int8_t *side,
                             const fmk_directional_out *out)
{
    // base.py:409-546 has no length check on bar_close_indices (only comp_bar_ohlcv has, base.py:334-335): one element
    // means zero bars and empty outputs (the reference's tests use that for comp_bar_footprints)
    if (n_idx == 1) return FMK_OK;
    if (n_idx < 1) return fmk_set_error(ctx, FMK_E_ARG, "negative dimensions are not allowed");
    const int64_t nb = n_idx - 1;
    DevBag bag(ctx);
    double *d_p;
    void *d_a;
    int64_t *d_ci, *d_nz;
    int8_t *d_s;
    FMK_TRY(bag.up(price, n, &d_p));
    FMK_TRY(bag.up_amount(amount, amount_is_f64, n, &d_a));
    FMK_TRY(bag.up(close_idx, n_idx, &d_ci));
    FMK_TRY(bag.up(side, n, &d_s));
    FMK_TRY(bag.out(1, &d_nz));
    FMK_TRY(fmk_memset(ctx, d_nz, 0, 8));
    fmk_directional_out d;
    int64_t **di[] = {&d.ticks_buy, &d.ticks_sell, &d.cum_ticks_min, &d.cum_ticks_max};
    float **df[] = {&d.volume_buy, &d.volume_sell, &d.dollars_buy, &d.dollars_sell, &d.mean_spread, &d.max_spread,
                    &d.cum_volumes_min, &d.cum_volumes_max, &d.cum_dollars_min, &d.cum_dollars_max};
    for (auto p : di) FMK_TRY(bag.out(nb, p));
    for (auto p : df) FMK_TRY(bag.out(nb, p));
    FMK_TRY(fmk_comp_bar_directional_dev(ctx, d_p, d_a, amount_is_f64, n, d_ci, n_idx, d_s, &d, d_nz));
    FMK_TRY(down(ctx, out->ticks_buy, d.ticks_buy, nb));
    FMK_TRY(down(ctx, out->ticks_sell, d.ticks_sell, nb));
    FMK_TRY(down(ctx, out->volume_buy, d.volume_buy, nb));
    FMK_TRY(down(ctx, out->volume_sell, d.volume_sell, nb));
    FMK_TRY(down(ctx, out->dollars_buy, d.dollars_buy, nb));
    FMK_TRY(down(ctx, out->dollars_sell, d.dollars_sell, nb));
    FMK_TRY(down(ctx, out->mean_spread, d.mean_spread, nb));
    FMK_TRY(down(ctx, out->max_spread, d.max_spread, nb));
    FMK_TRY(down(ctx, out->cum_ticks_min, d.cum_ticks_min, nb));
    FMK_TRY(down(ctx, out->cum_ticks_max, d.cum_ticks_max, nb));
    FMK_TRY(down(ctx, out->cum_volumes_min, d.cum_volumes_min, nb));
    FMK_TRY(down(ctx, out->cum_volumes_max, d.cum_volumes_max, nb));
    FMK_TRY(down(ctx, out->cum_dollars_min, d.cum_dollars_min, nb));
    FMK_TRY(down(ctx, out->cum_dollars_max, d.cum_dollars_max, nb));
    int64_t nz = 0;
    FMK_TRY(fmk_d2h(ctx, &nz, d_nz, 8));
    if (nz > 0)   // base.py:536: current_cum_spread / (buy + sell) with buy + sell == 0
        return fmk_set_error(ctx, FMK_E_ZERODIV, "division by zero: %lld bar(s) without a signed tick", (long long)nz);
    return FMK_OK;
}

int fmk_comp_bar_footprints(fmk_ctx *ctx, const double *price, const void *amount, int amount_is_f64, int64_t n,
                            const int64_t *close_idx, int64_t n_idx, const int8_t *side, double price_tick_size,
                            const double *bar_lows, const double *bar_highs, double imbalance_factor,
                            int64_t *level_offsets, const fmk_footprint_out *out)
{
    // base.py:615-752 has no length check: one element = zero bars = empty outputs
    // (reference test tests/bars/test_comp_bar_footprints.py::test_comp_bar_footprints_empty_bar)
    if (n_idx == 1) { if (level_offsets) level_offsets[0] = 0; return FMK_OK; }
    if (n_idx < 1) return fmk_set_error(ctx, FMK_E_ARG, "negative dimensions are not allowed");
    const int64_t nb = n_idx - 1;
    DevBag bag(ctx);
    double *d_lo, *d_hi;
    int64_t *d_off;
    FMK_TRY(bag.up(bar_lows, nb, &d_lo));
    FMK_TRY(bag.up(bar_highs, nb, &d_hi));
    FMK_TRY(bag.out(nb + 1, &d_off));
    int64_t total = 0, maxl = 0;
    FMK_TRY(fmk_comp_bar_footprints_size_dev(ctx, d_lo, d_hi, nb, price_tick_size, d_off, &total, &maxl));
    FMK_TRY(down(ctx, level_offsets, d_off, nb + 1));
    if (!out) return FMK_OK;
    double *d_p;
    void *d_a;
    int64_t *d_ci, *d_bad;
    int8_t *d_s;
    FMK_TRY(bag.up(price, n, &d_p));
    FMK_TRY(bag.up_amount(amount, amount_is_f64, n, &d_a));
    FMK_TRY(bag.up(close_idx, n_idx, &d_ci));
    FMK_TRY(bag.up(side, n, &d_s));
    FMK_TRY(bag.out(1, &d_bad));
    FMK_TRY(fmk_memset(ctx, d_bad, 0, 8));
    fmk_footprint_out d;
    FMK_TRY(bag.out(total, &d.price_levels));
    FMK_TRY(bag.out(total, &d.buy_volumes));
    FMK_TRY(bag.out(total, &d.sell_volumes));
    FMK_TRY(bag.out(total, &d.buy_ticks));
    FMK_TRY(bag.out(total, &d.sell_ticks));
    FMK_TRY(bag.out(total, &d.buy_imbalances));
    FMK_TRY(bag.out(total, &d.sell_imbalances));
    FMK_TRY(bag.out(nb, &d.buy_imbalances_sum));
    FMK_TRY(bag.out(nb, &d.sell_imbalances_sum));
    FMK_TRY(bag.out(nb, &d.cot_price_levels));
    FMK_TRY(bag.out(nb, &d.imb_max_run_signed));
    FMK_TRY(bag.out(nb, &d.vp_skew));
    FMK_TRY(bag.out(nb, &d.vp_gini));
    FMK_TRY(fmk_memset(ctx, d.buy_imbalances_sum, 0, 2 * (size_t)nb));
    FMK_TRY(fmk_memset(ctx, d.sell_imbalances_sum, 0, 2 * (size_t)nb));
    FMK_TRY(fmk_memset(ctx, d.cot_price_levels, 0, 4 * (size_t)nb));
    FMK_TRY(fmk_memset(ctx, d.imb_max_run_signed, 0, 2 * (size_t)nb));
    FMK_TRY(fmk_memset(ctx, d.vp_skew, 0, 8 * (size_t)nb));
    FMK_TRY(fmk_memset(ctx, d.vp_gini, 0, 8 * (size_t)nb));
    FMK_TRY(fmk_comp_bar_footprints_fill_dev(ctx, d_p, d_a, amount_is_f64, n, d_ci, n_idx, d_s, price_tick_size, d_lo,
                                             imbalance_factor, d_off, maxl, &d, d_bad));
    FMK_TRY(down(ctx, out->price_levels, d.price_levels, total));
    FMK_TRY(down(ctx, out->buy_volumes, d.buy_volumes, total));
    FMK_TRY(down(ctx, out->sell_volumes, d.sell_volumes, total));
    FMK_TRY(down(ctx, out->buy_ticks, d.buy_ticks, total));
    FMK_TRY(down(ctx, out->sell_ticks, d.sell_ticks, total));
    FMK_TRY(down(ctx, out->buy_imbalances, d.buy_imbalances, total));
    FMK_TRY(down(ctx, out->sell_imbalances, d.sell_imbalances, total));
    FMK_TRY(down(ctx, out->buy_imbalances_sum, d.buy_imbalances_sum, nb));
    FMK_TRY(down(ctx, out->sell_imbalances_sum, d.sell_imbalances_sum, nb));
    FMK_TRY(down(ctx, out->cot_price_levels, d.cot_price_levels, nb));
    FMK_TRY(down(ctx, out->imb_max_run_signed, d.imb_max_run_signed, nb));
    FMK_TRY(down(ctx, out->vp_skew, d.vp_skew, nb));
    FMK_TRY(down(ctx, out->vp_gini, d.vp_gini, nb));
    int64_t bad = 0;
    FMK_TRY(fmk_d2h(ctx, &bad, d_bad, 8));
    if (bad > 0)   // base.py:719
        return fmk_set_error(ctx, FMK_E_LEVEL, "Something went wrong! Invalid price level index!");
    return FMK_OK;
}

int fmk_comp_bar_trade_size(fmk_ctx *ctx, const void *amount, int amount_is_f64, int64_t n, const double *theta,
                            const int64_t *close_idx, int64_t n_idx, double theta_mult, float *mean_size_rel,
                            float *size_95_rel, float *pct_block, float *size_gini)
{
    if (n_idx == 1) return FMK_OK;   // base.py:549-612 checks theta's length only: zero bars, empty outputs
    if (n_idx < 1) return fmk_set_error(ctx, FMK_E_ARG, "negative dimensions are not allowed");
    const int64_t nb = n_idx - 1;
    DevBag bag(ctx);
    void *d_a;
    double *d_th;
    int64_t *d_ci;
    float *d_o[4];
    FMK_TRY(bag.up_amount(amount, amount_is_f64, n, &d_a));
    FMK_TRY(bag.up(theta, nb, &d_th));
    FMK_TRY(bag.up(close_idx, n_idx, &d_ci));
    for (int k = 0; k < 4; ++k) FMK_TRY(bag.out(nb, &d_o[k]));
    FMK_TRY(fmk_comp_bar_trade_size_dev(ctx, d_a, amount_is_f64, n, d_th, d_ci, n_idx, theta_mult, d_o[0], d_o[1],
                                        d_o[2], d_o[3]));
    FMK_TRY(down(ctx, mean_size_rel, d_o[0], nb));
    FMK_TRY(down(ctx, size_95_rel, d_o[1], nb));
    FMK_TRY(down(ctx, pct_block, d_o[2], nb));
    return down(ctx, size_gini, d_o[3], nb);
}

int fmk_comp_lagged_returns(fmk_ctx *ctx, const int64_t *ts, const double *close_, int64_t n,
                            double return_window_sec, int is_log, double *out)
{
    if (!(return_window_sec > 0))
        return fmk_set_error(ctx, FMK_E_ARG, "The return window must be greater than zero.");
    if (n <= 0) return FMK_OK;
    DevBag bag(ctx);
    int64_t *d_ts;
    double *d_c, *d_o;
    FMK_TRY(bag.up(ts, n, &d_ts));
    FMK_TRY(bag.up(close_, n, &d_c));
    FMK_TRY(bag.out(n, &d_o));
    FMK_TRY(fmk_comp_lagged_returns_dev(ctx, d_ts, d_c, n, return_window_sec, is_log, d_o));
    return down(ctx, out, d_o, n);
}

int fmk_ewmst(fmk_ctx *ctx, const int64_t *ts, const double *y, int64_t n, double half_life, double sigma_floor,
              int mean0, double *out)
{
    if (n <= 0) return FMK_OK;
    DevBag bag(ctx);
    int64_t *d_ts;
    double *d_y, *d_o;
    FMK_TRY(bag.up(ts, n, &d_ts));
    FMK_TRY(bag.up(y, n, &d_y));
    FMK_TRY(bag.out(n, &d_o));
    FMK_TRY(fmk_ewmst_dev(ctx, d_ts, d_y, n, half_life, sigma_floor, mean0, d_o));
    return down(ctx, out, d_o, n);
}

int fmk_ewms(fmk_ctx *ctx, const double *y, int64_t n, int64_t span, double *out)
{
    if (n <= 0) return FMK_OK;
    DevBag bag(ctx);
    double *d_y, *d_o;
    FMK_TRY(bag.up(y, n, &d_y));
    FMK_TRY(bag.out(n, &d_o));
    FMK_TRY(fmk_ewms_dev(ctx, d_y, n, span, d_o));
    return down(ctx, out, d_o, n);
}

int fmk_realized_vol(fmk_ctx *ctx, const double *r, int64_t n, int64_t window, int is_sample, double *out)
{
    if (window < 1) return fmk_set_error(ctx, FMK_E_ARG, "window must be at least 1");
    if (n <= 0) return FMK_OK;
    DevBag bag(ctx);
    double *d_r, *d_o;
    FMK_TRY(bag.up(r, n, &d_r));
    FMK_TRY(bag.out(n, &d_o));
    FMK_TRY(fmk_realized_vol_dev(ctx, d_r, n, window, is_sample, d_o));
    return down(ctx, out, d_o, n);
}

int fmk_merge_split_trades(fmk_ctx *ctx, const int64_t *ts, const double *price, const float *amount,
                           const uint8_t *is_buyer_maker, int64_t n, int64_t *out_ts, double *out_price,
                           float *out_amount, int8_t *out_side, int64_t capacity, int64_t *n_merged)
{
    if (n <= 0) return fmk_set_error(ctx, FMK_E_ARG, "merge_split_trades: empty input");
    DevBag bag(ctx);
    int64_t *d_ts, *d_ots = nullptr;
    double *d_p, *d_op = nullptr;
    float *d_a, *d_oa = nullptr;
    uint8_t *d_m = nullptr;
    int8_t *d_os = nullptr;
    FMK_TRY(bag.up(ts, n, &d_ts));
    FMK_TRY(bag.up(price, n, &d_p));
    FMK_TRY(bag.up(amount, n, &d_a));
    if (is_buyer_maker) FMK_TRY(bag.up(is_buyer_maker, n, &d_m));
    if (out_ts) {
        FMK_TRY(bag.out(n, &d_ots));
        FMK_TRY(bag.out(n, &d_op));
        FMK_TRY(bag.out(n, &d_oa));
        if (out_side && is_buyer_maker) FMK_TRY(bag.out(n, &d_os));
    }
    int64_t m = 0;
    FMK_TRY(fmk_merge_split_trades_dev(ctx, d_ts, d_p, d_a, d_m, n, d_ots, d_op, d_oa, d_os, out_ts ? n : 0, &m));
    if (n_merged) *n_merged = m;
    if (!out_ts) return FMK_OK;
    if (capacity < m)
        return fmk_set_error(ctx, FMK_E_CAPACITY, "merge_split_trades: %lld merged trades, capacity %lld", (long long)m,
                             (long long)capacity);
    FMK_TRY(down(ctx, out_ts, d_ots, m));
    FMK_TRY(down(ctx, out_price, d_op, m));
    FMK_TRY(down(ctx, out_amount, d_oa, m));
    if (d_os) FMK_TRY(down(ctx, out_side, d_os, m));
    return FMK_OK;
}

int fmk_comp_trade_side_vector(fmk_ctx *ctx, const double *price, int64_t n, int8_t *out)
{
    if (n <= 0) return fmk_set_error(ctx, FMK_E_ARG, "comp_trade_side_vector: empty input");
    DevBag bag(ctx);
    double *d_p;
    int8_t *d_o;
    FMK_TRY(bag.up(price, n, &d_p));
    FMK_TRY(bag.out(n, &d_o));
    FMK_TRY(fmk_comp_trade_side_vector_dev(ctx, d_p, n, d_o));
    return down(ctx, out, d_o, n);
}

int fmk_cusum_bar_indexer(fmk_ctx *ctx, const int64_t *ts, const double *price, double *sigma, int64_t n,
                          double sigma_floor, double sigma_mult, int64_t *out, int64_t capacity, int64_t *n_out)
{
    if (n <= 0) return fmk_set_error(ctx, FMK_E_ARG, "Prices, timestamps, and sigma arrays must have the same length.");
    DevBag bag(ctx);
    int64_t *d_ts, *d_o = nullptr;
    double *d_p, *d_s;
    FMK_TRY(bag.up(ts, n, &d_ts));
    FMK_TRY(bag.up(price, n, &d_p));
    FMK_TRY(bag.up((const double *)sigma, n, &d_s));
    if (out) FMK_TRY(bag.out(n, &d_o));
    int64_t m = 0;
    FMK_TRY(fmk_cusum_bar_indexer_dev(ctx, d_ts, d_p, d_s, n, sigma_floor, sigma_mult, d_o, out ? n : 0, &m, nullptr));
    if (n_out) *n_out = m;
    FMK_TRY(down(ctx, sigma, (const double *)d_s, n));          // the reference forward-fills sigma in place
    if (!out) return FMK_OK;
    if (capacity < m)
        return fmk_set_error(ctx, FMK_E_CAPACITY, "cusum: %lld close indices, capacity %lld", (long long)m,
                             (long long)capacity);
    return down(ctx, out, (const int64_t *)d_o, m);
}

int fmk_volume_profile_rolling(fmk_ctx *ctx, const int64_t *bar_ts, const double *highs, const double *lows,
                               const int64_t *level_offsets, const int32_t *price_levels, const float *buy_volumes,
                               const float *sell_volumes, int64_t n_bars, int64_t first_bar, int64_t window_ns,
                               int64_t n_bins, double price_tick, double va_pct, int32_t *poc, int32_t *hva, int32_t *lva,
                               float *pct)
{
    if (n_bars <= 0) return fmk_set_error(ctx, FMK_E_ARG, "Input arrays should have the same length and be non-empty.");
    const int64_t nl = level_offsets[n_bars];
    DevBag bag(ctx);
    int64_t *d_ts, *d_off;
    double *d_hi, *d_lo;
    int32_t *d_pl, *d_poc, *d_hva, *d_lva;
    float *d_b, *d_s, *d_pct;
    FMK_TRY(bag.up(bar_ts, n_bars, &d_ts));
    FMK_TRY(bag.up(highs, n_bars, &d_hi));
    FMK_TRY(bag.up(lows, n_bars, &d_lo));
    FMK_TRY(bag.up(level_offsets, n_bars + 1, &d_off));
    FMK_TRY(bag.up(price_levels, nl, &d_pl));
    FMK_TRY(bag.up(buy_volumes, nl, &d_b));
    FMK_TRY(bag.up(sell_volumes, nl, &d_s));
    FMK_TRY(bag.out(n_bars, &d_poc));
    FMK_TRY(bag.out(n_bars, &d_hva));
    FMK_TRY(bag.out(n_bars, &d_lva));
    FMK_TRY(bag.out(n_bars, &d_pct));
    FMK_TRY(fmk_volume_profile_rolling_dev(ctx, d_ts, d_hi, d_lo, d_off, d_pl, d_b, d_s, n_bars, first_bar, window_ns,
                                           n_bins, price_tick, va_pct, d_poc, d_hva, d_lva, d_pct));
    FMK_TRY(down(ctx, poc, (const int32_t *)d_poc, n_bars));
    FMK_TRY(down(ctx, hva, (const int32_t *)d_hva, n_bars));
    FMK_TRY(down(ctx, lva, (const int32_t *)d_lva, n_bars));
    return down(ctx, pct, (const float *)d_pct, n_bars);
}

}  // extern "C"
