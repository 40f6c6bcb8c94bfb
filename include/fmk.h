/*
 * fmk.h -- C ABI of libfmk_hip.so, the MI355X (gfx950) tick->bar engine.
 *
 * This is the drop-in boundary for the hot path of quantscious/finmlkit
 * (tick arrays -> bar close indices -> per-bar OHLCV / order-flow / footprint,
 * plus the tick-level volatility loops).  The reference has no FFI of its own:
 * its boundary is "a module-level Python function over contiguous 1-D NumPy
 * arrays" (finmlkit/bar/logic.py, finmlkit/bar/base.py:306-850,
 * finmlkit/feature/core/{utils,volatility}.py).  Each entry point below
 * replaces exactly one of those functions; the ctypes binding a maintainer
 * would add is shown in INTEGRATION.md and shipped in finmlkit_amd/_ffi.py.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ / torch types.
 *   - every function returns an int status (FMK_OK == 0, negative = error);
 *     fmk_last_error() gives the text.  The Python shim maps the codes to
 *     the exception types the reference raises (see FMK_E_* below).
 *   - *_dev functions take DEVICE pointers and enqueue on the context's HIP
 *     stream without synchronising (fmk_ctx_sync to wait).  The functions
 *     without the suffix take HOST pointers (the NumPy drop-in): they upload,
 *     run the same kernels, download and synchronise.
 *   - `amount` columns may be float32 (the reference's preprocessed dtype,
 *     data_model.py:332-342) or float64: `amount_is_f64` selects.
 *   - close-index arrays follow the reference: n_idx = n_bars + 1 entries,
 *     entry 0 is the "open" edge (may be -1), bar i covers ticks
 *     close_idx[i]+1 .. close_idx[i+1] inclusive (base.py:364,377).
 *   - the caller owns every buffer; the library keeps no host pointer after
 *     a call returns.
 *   - there is NO CPU fallback: without a usable HIP device fmk_ctx_create
 *     fails with FMK_E_NODEVICE.
 */
#ifndef FMK_H
#define FMK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMK_OK 0
#define FMK_E_ARG (-1)      /* ValueError (base.py:332-335, utils.py:33-34, ...) */
#define FMK_E_CAPACITY (-2) /* caller buffer too small (two-phase protocol misuse) */
#define FMK_E_LEVEL (-3)    /* ValueError "Invalid price level index" (base.py:719) */
#define FMK_E_ZERODIV (-4)  /* ZeroDivisionError (base.py:536: bar without signed tick) */
#define FMK_E_NOMEM (-5)    /* MemoryError */
#define FMK_E_HIP (-6)      /* RuntimeError: HIP runtime failure, see fmk_last_error */
#define FMK_E_NODEVICE (-7) /* RuntimeError: no gfx950 device / HIP runtime unusable */
#define FMK_E_COMM (-8)     /* RuntimeError: multi-GPU rendezvous / RCCL failure or a peer that never arrived */

#define FMK_ABI_VERSION 1

typedef struct fmk_ctx fmk_ctx;

/* ---- context, memory, timing --------------------------------------------------------- */
int fmk_abi_version(void);
int fmk_device_count(int *count);
int fmk_ctx_create(int device, fmk_ctx **out);
int fmk_ctx_destroy(fmk_ctx *ctx);
int fmk_ctx_sync(fmk_ctx *ctx);
/* Release everything the context keeps between calls: scratch, the caching allocator's free blocks, the work
 * buffers of the threshold indexers (up to ~20 B/tick).  For long-lived processes; never required. */
int fmk_ctx_trim(fmk_ctx *ctx);
/* hipStream_t of the context (for interop with other HIP users, e.g. RCCL). */
void *fmk_ctx_stream(fmk_ctx *ctx);
/* Text of the last error on this context (ctx == NULL: last context-less error). */
const char *fmk_last_error(const fmk_ctx *ctx);

int fmk_alloc(fmk_ctx *ctx, size_t bytes, void **dptr);
int fmk_free(fmk_ctx *ctx, void *dptr);
int fmk_memset(fmk_ctx *ctx, void *dptr, int value, size_t bytes);
int fmk_h2d(fmk_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes); /* sync */
int fmk_d2h(fmk_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes); /* sync */
/* Several host columns to the device by one call (a TradesData frame -> HBM), complete on return.  Default: each column through
 * the runtime's own pageable copy (56 GB/s = 98 % of the pinned rate on the MI355X boxes, profiles/r03_apibench.txt);
 * FMK_UPLOAD_THREADS=n in the environment: n worker threads stage 4 MiB chunks through their own pinned double buffers and
 * streams instead (for hosts whose runtime copies pageable memory through one bounce buffer). */
int fmk_h2d_columns(fmk_ctx *ctx, int n_cols, void *const *dst_dev, const void *const *src_host, const size_t *bytes);
int fmk_d2d(fmk_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes);  /* async */
int fmk_mem_info(fmk_ctx *ctx, size_t *free_bytes, size_t *total_bytes);

/* hipEvent timer on the context stream: ms between start and stop (stop syncs). */
int fmk_timer_start(fmk_ctx *ctx);
int fmk_timer_stop(fmk_ctx *ctx, double *elapsed_ms);
/* Non-blocking event pairs (per-kernel timing inside a timed region): record on the context
 * stream, read the elapsed time after fmk_ctx_sync. */
int fmk_event_create(fmk_ctx *ctx, void **event);
int fmk_event_destroy(fmk_ctx *ctx, void *event);
int fmk_event_record(fmk_ctx *ctx, void *event);
int fmk_event_elapsed(fmk_ctx *ctx, void *start, void *stop, double *elapsed_ms);
/* Per-launch timing of the dominant kernel (k_bar_ohlcv_small / k_bar_ohlcv of comp_bar_ohlcv):
 * while enabled, every fmk_comp_bar_ohlcv_dev call brackets that ONE kernel launch with a HIP event
 * pair on the context stream (ring of 256).  fmk_profile_read synchronises and returns the durations. */
/* Volume / dollar bar indexers: the parallel algorithms work on exactly computed sums and count the decisions that
 * land within the rounding drift of the reference's float64 running sum (n_uncertified).  on = 0 (default):
 * volume bars -- fragile decisions are settled by replaying their bar with the reference's sequential sum (every
 * bar starts from 0, so it can be replayed alone) and n_uncertified comes back 0: at no measurable cost for continuous
 * amounts, at (fragile ticks) x (bar length) additions for decimal lots with round thresholds; when that would cost more
 * than it, and for any uncertified dollar-bar input (the carried remainder never resets: no bar can be replayed alone),
 * the input is redone by the reference's own loop, operation for operation (one wave, 15-22 ns per tick).
 * on = 1: the parallel result is returned as is, each uncertified decision may differ from the reference by one tick
 * (resident pipelines of 1e9 ticks). */
int fmk_ctx_set_fast_threshold(fmk_ctx *ctx, int on);
/* Enqueue-only mode.  on = 0 (default): fmk_comp_bar_ohlcv_dev waits for its first kernel and reads back whether any bar was left
 * for the long-bar schedules; when none was -- streams of 1-minute bars and the like -- the ~36 launches that serve such bars are not
 * issued at all (they would exit at once, ~0.25 ms per call).  on = 1: the call never waits; everything is enqueued unconditionally
 * and it returns before its kernels have run -- for callers that overlap the call with other streams' work and must not block
 * (the sharded step of finmlkit_amd/dist.py between its halo exchange and its boundary bar).  One wait remains in this mode:
 * fmk_time_bars_ohlcv_dev takes the long-bar census from its INDEX stages (~0.1 ms into the call, while its first OHLCV launch runs:
 * the device never idles for it) instead of enqueuing the ~36 launches blindly -- 0.25 ms per sharded step (FMK_TB_PIPE_EO_CENSUS=0
 * restores the blind enqueue).  A call whose tick array is no longer than the first kernel's reach enqueues nothing more either way. */
int fmk_ctx_set_enqueue_only(fmk_ctx *ctx, int on);
int fmk_profile_enable(fmk_ctx *ctx, int on);
int fmk_profile_read(fmk_ctx *ctx, double *ms, int capacity, int *count);
int fmk_profile_count(fmk_ctx *ctx, int64_t *total);   /* launches timed since enable (> 256: the ring holds the last 256, oldest at slot total mod 256) */

/* ---- synthetic tick stream (same definition as oracle/orc_synth) ----- */
/* Writes ticks [first, first+n) of stream `seed` into device columns (any may be NULL). */
int fmk_synth_trades_dev(fmk_ctx *ctx, uint64_t seed, int64_t first, int64_t n, uint64_t gap_mod,
                         int64_t *d_ts, double *d_price, float *d_amount, int8_t *d_side);

/* ---- bar indexers: finmlkit/bar/logic.py ---------------------------------------------- */
/* _time_bar_indexer (logic.py:12-51).  fmk_time_bar_clock is pure host arithmetic: the
 * float64 clock of logic.py:30-39 as NumPy evaluates it -> edge k = first_edge + k*delta. */
int fmk_time_bar_clock(int64_t ts_first, int64_t ts_last, double interval_seconds,
                       int64_t *n_edges, int64_t *first_edge, int64_t *delta);
/* Fills d_clock[n_edges] and d_close_idx[n_edges] (searchsorted(ts, clock, 'right') - 1). */
int fmk_time_bar_indexer_dev(fmk_ctx *ctx, const int64_t *d_ts, int64_t n, int64_t first_edge,
                             int64_t delta, int64_t n_edges, int64_t *d_clock,
                             int64_t *d_close_idx);
/* Host flavour, two-phase: call with clock == NULL to get *n_edges, then with buffers. */
int fmk_time_bar_indexer(fmk_ctx *ctx, const int64_t *ts, int64_t n, double interval_seconds,
                         int64_t *clock, int64_t *close_idx, int64_t capacity, int64_t *n_edges);

/* _tick_bar_indexer (logic.py:54-84): closed form.  *n_idx receives the number of close
 * indices (first is 0); d_close_idx may be NULL to query the count only. */
int fmk_tick_bar_indexer_dev(fmk_ctx *ctx, int64_t n, int64_t threshold, int64_t *d_close_idx,
                             int64_t capacity, int64_t *n_idx);
/* _volume_bar_indexer (logic.py:87-115) and _dollar_bar_indexer (logic.py:118-149).
 * Two-phase: d_close_idx == NULL -> count only.  *n_uncertified (may be NULL) receives the
 * number of close decisions whose margin to the threshold was below the floating-point
 * reordering bound (see DESIGN.md "threshold bars"); 0 means certified bit-identical. */
int fmk_volume_bar_indexer_dev(fmk_ctx *ctx, const void *d_amount, int amount_is_f64, int64_t n,
                               double threshold, int64_t *d_close_idx, int64_t capacity,
                               int64_t *n_idx, int64_t *n_uncertified);
int fmk_dollar_bar_indexer_dev(fmk_ctx *ctx, const double *d_price, const void *d_amount,
                               int amount_is_f64, int64_t n, double threshold,
                               int64_t *d_close_idx, int64_t capacity, int64_t *n_idx,
                               int64_t *n_uncertified);
/* d_out[i] = d_in[idx[i]] for int64 (close_ts = timestamps[close_indices], kit.py:66). */
int fmk_gather_i64_dev(fmk_ctx *ctx, const int64_t *d_in, int64_t n_in, const int64_t *d_idx,
                       int64_t n_idx, int64_t *d_out);

/* ---- per-bar reducers: finmlkit/bar/base.py ------------------------------------------- */
/* comp_bar_ohlcv (base.py:306-407).  d_median may be NULL (skip the order statistic). */
int fmk_comp_bar_ohlcv_dev(fmk_ctx *ctx, const double *d_price, const void *d_amount,
                           int amount_is_f64, int64_t n, const int64_t *d_close_idx,
                           int64_t n_idx, double *d_open, double *d_high, double *d_low,
                           double *d_close, float *d_volume, double *d_vwap, int64_t *d_trades,
                           double *d_median);
/* TimeBarKit.build_ohlcv on resident columns in ONE call (kit.py:42-66 -> base.py:126-158): _time_bar_indexer
 * (logic.py:12-51) + comp_bar_ohlcv (base.py:306-407).  Results = fmk_time_bar_indexer_dev(first_edge, delta, n_edges)
 * followed by fmk_comp_bar_ohlcv_dev on its close indices, bit for bit (d_clock may be NULL, d_close_idx[n_edges] is
 * always written; ts_first / ts_last = d_ts[0] / d_ts[n-1], the two values the clock was made from).  For streams of
 * 1-minute-sized bars (mean bar length above 600 ticks, float32 amounts) the call is pipelined: the clock edges of the first eighth of
 * the bars, then OHLCV + median of those bars while the remaining edges are searched on a second stream (csrc/fmk_ohlcv.hip). */
int fmk_time_bars_ohlcv_dev(fmk_ctx *ctx, const int64_t *d_ts, const double *d_price, const void *d_amount,
                            int amount_is_f64, int64_t n, int64_t ts_first, int64_t ts_last, int64_t first_edge,
                            int64_t delta, int64_t n_edges, int64_t *d_clock, int64_t *d_close_idx,
                            double *d_open, double *d_high, double *d_low, double *d_close, float *d_volume,
                            double *d_vwap, int64_t *d_trades, double *d_median);
/* The order statistic alone (np.median of the bar's trade sizes, base.py:373-403). */
int fmk_comp_bar_median_dev(fmk_ctx *ctx, const void *d_amount, int amount_is_f64, int64_t n,
                            const int64_t *d_close_idx, int64_t n_idx, double *d_median);
int fmk_comp_bar_ohlcv(fmk_ctx *ctx, const double *price, const void *amount, int amount_is_f64,
                       int64_t n, const int64_t *close_idx, int64_t n_idx, double *open_,
                       double *high, double *low, double *close_, float *volume, double *vwap,
                       int64_t *trades, double *median);

/* comp_bar_directional_features (base.py:409-546).  Outputs in the reference's tuple order.
 * d_n_zero_div (device int64, may be NULL) counts bars without a signed tick, for which the
 * reference raises ZeroDivisionError; their mean_spread is written as NaN.  The host flavour
 * returns FMK_E_ZERODIV in that case (outputs are still filled). */
typedef struct fmk_directional_out {
    int64_t *ticks_buy, *ticks_sell;
    float *volume_buy, *volume_sell, *dollars_buy, *dollars_sell;
    float *mean_spread, *max_spread;
    int64_t *cum_ticks_min, *cum_ticks_max;
    float *cum_volumes_min, *cum_volumes_max, *cum_dollars_min, *cum_dollars_max;
} fmk_directional_out;
int fmk_comp_bar_directional_dev(fmk_ctx *ctx, const double *d_price, const void *d_amount,
                                 int amount_is_f64, int64_t n, const int64_t *d_close_idx,
                                 int64_t n_idx, const int8_t *d_side,
                                 const fmk_directional_out *d_out, int64_t *d_n_zero_div);
int fmk_comp_bar_directional(fmk_ctx *ctx, const double *price, const void *amount,
                             int amount_is_f64, int64_t n, const int64_t *close_idx,
                             int64_t n_idx, const int8_t *side, const fmk_directional_out *out);

/* comp_bar_footprints + comp_footprint_features (base.py:615-850), CSR output.
 * Phase 1: level_offsets[n_bars+1] from the bars' lows/highs (exclusive scan of
 *          int(round(high/tick)) - int(round(low/tick)) + 1); *total_levels and *max_levels
 *          are returned to the host (synchronises).
 * Phase 2: fills the flat per-level arrays (length total_levels) and the per-bar arrays. */
typedef struct fmk_footprint_out {
    int32_t *price_levels;   /* flat [total_levels] */
    float *buy_volumes, *sell_volumes;
    int32_t *buy_ticks, *sell_ticks;
    uint8_t *buy_imbalances, *sell_imbalances;   /* numpy bool */
    uint16_t *buy_imbalances_sum, *sell_imbalances_sum;   /* per bar [n_bars] */
    int32_t *cot_price_levels;
    int16_t *imb_max_run_signed;
    double *vp_skew, *vp_gini;
} fmk_footprint_out;
int fmk_comp_bar_footprints_size_dev(fmk_ctx *ctx, const double *d_bar_lows,
                                     const double *d_bar_highs, int64_t n_bars,
                                     double price_tick_size, int64_t *d_level_offsets,
                                     int64_t *total_levels, int64_t *max_levels);
int fmk_comp_bar_footprints_fill_dev(fmk_ctx *ctx, const double *d_price, const void *d_amount,
                                     int amount_is_f64, int64_t n, const int64_t *d_close_idx,
                                     int64_t n_idx, const int8_t *d_side, double price_tick_size,
                                     const double *d_bar_lows, double imbalance_factor,
                                     const int64_t *d_level_offsets, int64_t max_levels,
                                     const fmk_footprint_out *d_out, int64_t *d_n_bad_level);
/* Host flavour, two-phase: out == NULL -> fills level_offsets only. */
int fmk_comp_bar_footprints(fmk_ctx *ctx, const double *price, const void *amount,
                            int amount_is_f64, int64_t n, const int64_t *close_idx, int64_t n_idx,
                            const int8_t *side, double price_tick_size, const double *bar_lows,
                            const double *bar_highs, double imbalance_factor,
                            int64_t *level_offsets, const fmk_footprint_out *out);

/* comp_bar_trade_size_features (base.py:549-612).  theta[n_bars] float64;
 * outputs float32[n_bars]: mean_size_rel, size_95_rel, pct_block, size_gini (NaN where the reference
 * leaves NaN: empty bar, theta == 0, zero total volume). */
int fmk_comp_bar_trade_size_dev(fmk_ctx *ctx, const void *d_amount, int amount_is_f64, int64_t n,
                                const double *d_theta, const int64_t *d_close_idx, int64_t n_idx,
                                double theta_mult, float *d_mean_size_rel, float *d_size_95_rel,
                                float *d_pct_block, float *d_size_gini);
int fmk_comp_bar_trade_size(fmk_ctx *ctx, const void *amount, int amount_is_f64, int64_t n,
                            const double *theta, const int64_t *close_idx, int64_t n_idx,
                            double theta_mult, float *mean_size_rel, float *size_95_rel,
                            float *pct_block, float *size_gini);

/* ---- cfg 4: time bars + order-flow + footprints ----------------------------------------
 * Replaces BarBuilderBase.build_ohlcv + build_directional_features + build_footprints (base.py:126-300) called back to back (38 B/tick
 * as three reducers).  Two phases like comp_bar_footprints (the CSR row count must reach the caller's allocator).  (Round 1's
 * fmk_bars_fused_size_dev / _fill_dev -- comp_bar_ohlcv, then the two other reducers back to back -- were superseded by the entry
 * points below and removed in round 6.) */
/* cfg 4 in TWO passes over the ticks (round 2; 26 B/tick): build_ohlcv + build_directional_features semantics from one read of
 * price / amount / side (float32 amounts: one kernel; float64 amounts: the two kernels back to back), then the CSR level
 * counts.  The second pass is fmk_comp_bar_footprints_fill_dev with the offsets and lows produced here. */
int fmk_bars_flow_size_dev(fmk_ctx *ctx, const double *d_price, const void *d_amount, int amount_is_f64, int64_t n,
                           const int64_t *d_close_idx, int64_t n_idx, const int8_t *d_side, double price_tick_size,
                           double *d_open, double *d_high, double *d_low, double *d_close, float *d_volume, double *d_vwap,
                           int64_t *d_trades, double *d_median /* may be NULL */, const fmk_directional_out *d_dir,
                           int64_t *d_n_zero_div, int64_t *d_level_offsets, int64_t *total_levels, int64_t *max_levels);
/* cfg 4 at 26 B/tick (round 3): the same first pass, but the median trade size of comp_bar_ohlcv (base.py:401-404) may be left
 * to the footprint sweep, whose waves hold every amount of their bar anyway -- no pass of its own over the amount column.
 * *median_deferred = 1: nothing was written to d_median, the caller passes it to fmk_comp_bar_footprints_fill_median_dev;
 * 0: d_median is complete (float64 amounts, very short / very long bars) and the plain fill call follows. */
int fmk_bars_flow_size_defer_dev(fmk_ctx *ctx, const double *d_price, const void *d_amount, int amount_is_f64, int64_t n,
                                 const int64_t *d_close_idx, int64_t n_idx, const int8_t *d_side, double price_tick_size,
                                 double *d_open, double *d_high, double *d_low, double *d_close, float *d_volume,
                                 double *d_vwap, int64_t *d_trades, double *d_median, const fmk_directional_out *d_dir,
                                 int64_t *d_n_zero_div, int64_t *d_level_offsets, int64_t *total_levels, int64_t *max_levels,
                                 int *median_deferred);
/* comp_bar_footprints' fill phase (as fmk_comp_bar_footprints_fill_dev) that ALSO writes the per-bar median trade size
 * (float32 amounts only; d_median NULL = the plain fill). */
int fmk_comp_bar_footprints_fill_median_dev(fmk_ctx *ctx, const double *d_price, const void *d_amount, int amount_is_f64,
                                            int64_t n, const int64_t *d_close_idx, int64_t n_idx, const int8_t *d_side,
                                            double price_tick_size, const double *d_bar_lows, double imbalance_factor,
                                            const int64_t *d_level_offsets, int64_t max_levels,
                                            const fmk_footprint_out *d_out, int64_t *d_n_bad_level, double *d_median);

/* ---- tick-level feature loops: finmlkit/feature/core ---------------------------------- */
/* comp_lagged_returns (core/utils.py:12-64). */
int fmk_comp_lagged_returns_dev(fmk_ctx *ctx, const int64_t *d_ts, const double *d_close,
                                int64_t n, double return_window_sec, int is_log, double *d_out);
int fmk_comp_lagged_returns(fmk_ctx *ctx, const int64_t *ts, const double *close_, int64_t n,
                            double return_window_sec, int is_log, double *out);
/* ewmst / ewmst_mean0 (core/volatility.py:139-219, 72-136). */
int fmk_ewmst_dev(fmk_ctx *ctx, const int64_t *d_ts, const double *d_y, int64_t n,
                  double half_life, double sigma_floor, int mean0, double *d_out);
int fmk_ewmst(fmk_ctx *ctx, const int64_t *ts, const double *y, int64_t n, double half_life,
              double sigma_floor, int mean0, double *out);
/* Shards of ONE series across GPUs: tick 0 of the arrays is the last tick of the previous shard
 * (rank 0: its own first tick).  _map: affine map (a, a2, bV, bV2, bSy, bSyy) of ticks 1..n-1 -> d_map_out[6];
 * _apply: outputs from the incoming state d_state_in[4] = (V, V2, Sy, Syy) (device pointers; NULL = zeros). */
int fmk_ewmst_shard_map_dev(fmk_ctx *ctx, const int64_t *d_ts, const double *d_y, int64_t n, double half_life,
                            int mean0, double *d_map_out);
int fmk_ewmst_shard_apply_dev(fmk_ctx *ctx, const int64_t *d_ts, const double *d_y, int64_t n, double half_life,
                              double sigma_floor, int mean0, const double *d_state_in, double *d_out);
/* ewms (core/volatility.py:9-69): fixed alpha = 2/(span+1); span <= 1 -> all NaN. */
int fmk_ewms_dev(fmk_ctx *ctx, const double *d_y, int64_t n, int64_t span, double *d_out);
int fmk_ewms(fmk_ctx *ctx, const double *y, int64_t n, int64_t span, double *out);
/* realized_vol (core/volatility.py:256-286): rolling sqrt(nansum(r^2)/(valid - is_sample)) over `window`
 * elements; NaN where fewer than 2 valid returns or before the first full window.  FMK_E_ARG: window < 1 (the host
 * layer answers window == 0 itself: all NaN, like the reference; negative windows stay rejected). */
int fmk_realized_vol_dev(fmk_ctx *ctx, const double *d_r, int64_t n, int64_t window, int is_sample,
                         double *d_out);
int fmk_realized_vol(fmk_ctx *ctx, const double *r, int64_t n, int64_t window, int is_sample, double *out);

/* ---- rolling volume profile: finmlkit/feature/core/volume.py:133-456 ("next" rank 2) --------- */
/* volume_profile_rolling on CSR footprints (level_offsets[n_bars+1] + flat level arrays, the layout
 * fmk_comp_bar_footprints produces).  first_bar = searchsorted(bar_ts, bar_ts[0] + window_ns) (volume.py:432):
 * earlier bars keep 0.  n_bins < 0: no bucketing (n_bins=None).  Outputs: POC / HVA / LVA in tick units (int32) and
 * the share of volume above the POC (float32).  FMK_E_LEVEL: a level outside its window, or a single-level window
 * with bucketing (the reference raises); FMK_E_ZERODIV: n_bins == 0; FMK_E_CAPACITY: a window wider than 8192
 * levels. */
int fmk_volume_profile_rolling_dev(fmk_ctx *ctx, const int64_t *d_bar_ts, const double *d_highs, const double *d_lows,
                                   const int64_t *d_level_offsets, const int32_t *d_price_levels,
                                   const float *d_buy_volumes, const float *d_sell_volumes, int64_t n_bars,
                                   int64_t first_bar, int64_t window_ns, int64_t n_bins, double price_tick,
                                   double va_pct, int32_t *d_poc, int32_t *d_hva, int32_t *d_lva, float *d_pct);
int fmk_volume_profile_rolling(fmk_ctx *ctx, const int64_t *bar_ts, const double *highs, const double *lows,
                               const int64_t *level_offsets, const int32_t *price_levels, const float *buy_volumes,
                               const float *sell_volumes, int64_t n_bars, int64_t first_bar, int64_t window_ns,
                               int64_t n_bins, double price_tick, double va_pct, int32_t *poc, int32_t *hva,
                               int32_t *lva, float *pct);

/* calc_volume_percentage_above_poc (volume.py:367-391) on ONE profile: share of the volume on levels above `poc_price`
 * (NumPy-pairwise float32 total, the levels above added in order in float64, float64 quotient = the Numba-typed function). */
int fmk_calc_volume_percentage_above_poc_dev(fmk_ctx *ctx, const int32_t *d_price_levels, const float *d_volumes, int64_t n,
                                             int32_t poc_price, double *d_out);
int fmk_calc_volume_percentage_above_poc(fmk_ctx *ctx, const int32_t *price_levels, const float *volumes, int64_t n,
                                         int32_t poc_price, double *out);

/* The three stages of volume_profile_rolling as the reference exposes them (host arrays; the rolling entry above is the hot path):
 * aggregate_footprint (volume.py:134-203): the window [start_ts, end_ts] of bars (searchsorted left / right, the one-bar fallback),
 *   its dense level range from min(lows) / max(highs) (int(round(x / price_tick))), buy / sell volumes of the window's footprints
 *   (CSR: level_offsets[n_bars + 1]) added bar after bar in float32.  aligned_buy == NULL: sizes only (*min_level, *n_levels).
 * bucket_price_levels (volume.py:207-275): odd-width buckets over [min, max] of the levels, float32 sums in element order, the
 *   leftover bucket when the last level falls past the last edge.  binned_* == NULL: *n_out only.
 * comp_poc_hva_lva (volume.py:278-365): first argmax, then the value area grown two levels at a time towards the heavier side
 *   until va_pct of the total is covered.  THE CONTRACT: the total is np.sum of the float32 array as NumPy computes it (pairwise
 *   float32 -- the reference's pinned pure-Python mode; a jitted np.sum is a sequential float32 loop and can differ in the last bit),
 *   the walk's scalars are float64 (the typed reading of the function's `0.0` literals; pure Python under NumPy 2 adds its np.float32
 *   scalars in float32).  The readings coincide whenever the partial sums are exact, and on all 400 lognormal float32 profiles of
 *   tests/golden/volume_profile_stages.npz (`poc_lognormal`, made by the reference itself); a knife-edge profile whose covered volume
 *   equals the threshold to the last float32 bit can move HVA / LVA by one step between them. */
int fmk_aggregate_footprint(fmk_ctx *ctx, const int64_t *bar_ts, const double *highs, const double *lows,
                            const int64_t *level_offsets, const int32_t *price_levels, const float *buy_volumes,
                            const float *sell_volumes, int64_t n_bars, int64_t start_ts, int64_t end_ts, double price_tick,
                            int32_t *min_level, int64_t *n_levels, float *aligned_buy, float *aligned_sell, int64_t capacity);
int fmk_bucket_price_levels(fmk_ctx *ctx, const int32_t *all_price_levels, const float *total_volumes, int64_t n, int64_t n_bins,
                            int32_t *binned_price_levels, float *binned_volumes, int64_t capacity, int64_t *n_out);
int fmk_comp_poc_hva_lva(fmk_ctx *ctx, const int32_t *price_levels, const float *volumes, int64_t n, double va_pct,
                         int32_t *poc_price, int32_t *hva_price, int32_t *lva_price);

/* ---- CUSUM bars: finmlkit/bar/logic.py:152-221 ("next" rank 3) ------------------------------ */
/* _cusum_bar_indexer: sigma is forward-filled IN PLACE from its first non-NaN entry (like the reference); the
 * result starts with that entry's index, then one index per close.  d_out == NULL: count only (*n_out).
 * *n_rounds (may be NULL): rounds the parallel-in-time fixed point needed (chunks opened, when the chain walk for rarely
 * reached thresholds served the call).  FMK_E_CAPACITY: capacity < *n_out. */
int fmk_cusum_bar_indexer_dev(fmk_ctx *ctx, const int64_t *d_ts, const double *d_price, double *d_sigma, int64_t n,
                              double sigma_floor, double sigma_mult, int64_t *d_out, int64_t capacity,
                              int64_t *n_out, int64_t *n_rounds);
int fmk_cusum_bar_indexer(fmk_ctx *ctx, const int64_t *ts, const double *price, double *sigma, int64_t n,
                          double sigma_floor, double sigma_mult, int64_t *out, int64_t capacity, int64_t *n_out);

/* ---- TradesData(preprocess=True) loops: finmlkit/bar/utils.py ("next" rank 4) ------------ */
/* merge_split_trades (bar/utils.py:263-329): trades with the head's timestamp, maker flag and price (|dp| < 1e-8)
 * are merged, amounts summed in float32 in trade order; side = -1 if is_buyer_maker else 1 (is_buyer_maker may be
 * NULL: no side output).  *n_merged receives the number of merged trades; pass out_ts == NULL to only count.
 * FMK_E_CAPACITY when capacity < *n_merged. */
int fmk_merge_split_trades_dev(fmk_ctx *ctx, const int64_t *d_ts, const double *d_price, const float *d_amount,
                               const uint8_t *d_is_buyer_maker, int64_t n, int64_t *d_out_ts, double *d_out_price,
                               float *d_out_amount, int8_t *d_out_side, int64_t capacity, int64_t *n_merged);
int fmk_merge_split_trades(fmk_ctx *ctx, const int64_t *ts, const double *price, const float *amount,
                           const uint8_t *is_buyer_maker, int64_t n, int64_t *out_ts, double *out_price,
                           float *out_amount, int8_t *out_side, int64_t capacity, int64_t *n_merged);
/* comp_trade_side_vector (bar/utils.py:26-46): tick rule, side[0] = 0. */
int fmk_comp_trade_side_vector_dev(fmk_ctx *ctx, const double *d_price, int64_t n, int8_t *d_out);
int fmk_comp_trade_side_vector(fmk_ctx *ctx, const double *price, int64_t n, int8_t *out);

/* ---- TimeBarReader._resample: finmlkit/bar/io.py:890-950 ("next" rank 4, second half) --------------------------
 * Bars -> coarser bars.  Group g covers the rows [seg[g], seg[g+1]) of the input frame (contiguous; the host computes the
 * groups with pandas' own index.floor(timeframe)).  Per group: open = first / close = last non-NaN, high = max, low = min,
 * volume = pandas' Kahan-compensated sum in the column's dtype (float32 stays float32), trades = integer sum,
 * vwap = float32(sum(vwap * volume) / sum(volume)) with NumPy's dtype promotion of the product, median_trade_size =
 * float32 of the trades-weighted median of the rows' medians (searchsorted(cumsum(w), total / 2, 'left')).
 * o_valid[g] = 0 where every open of the group is NaN (the reference drops those rows, io.py:948). */
int fmk_resample_bars_dev(fmk_ctx *ctx, const int64_t *d_seg, int64_t n_groups, const double *d_open, const double *d_high,
                          const double *d_low, const double *d_close, const void *d_volume, int volume_is_f64,
                          const int64_t *d_trades, const void *d_vwap, int vwap_is_f64, const double *d_median,
                          double *d_o_open, double *d_o_high, double *d_o_low, double *d_o_close, void *d_o_volume,
                          int64_t *d_o_trades, float *d_o_vwap, float *d_o_median, uint8_t *d_o_valid);
int fmk_resample_bars(fmk_ctx *ctx, const int64_t *seg, int64_t n_groups, int64_t n_rows, const double *open,
                      const double *high, const double *low, const double *close, const void *volume, int volume_is_f64,
                      const int64_t *trades, const void *vwap, int vwap_is_f64, const double *median, double *o_open,
                      double *o_high, double *o_low, double *o_close, void *o_volume, int64_t *o_trades, float *o_vwap,
                      float *o_median, uint8_t *o_valid);

/* ---- multi-GPU: one neighbour halo exchange per step (BASELINE cfg 5) ------------------------
 * The reference has no distributed code (last row); these entry points are new.  Rank r of `world` holds
 * a contiguous tick range of one stream; the trailing partial bar of rank r travels as raw ticks to rank r+1.
 * No PyTorch: librccl is dlopen'ed by fmk_comm_create(FMK_COMM_RCCL) and the ranks of the node meet in a shared-memory
 * file (`rendezvous_path`, the same string on every rank; rank 0 creates it and unlinks it once all have attached).
 * Every wait has a deadline (`timeout_s`, <= 0: 120 s): a missing peer is FMK_E_COMM, not a hang. */
typedef struct fmk_comm fmk_comm;
#define FMK_COMM_RCCL 0      /* ncclSend / ncclRecv over xGMI on the communicator's own stream */
#define FMK_COMM_HOST 1      /* host-staged through the rendezvous segment; ctx may be NULL (host pointers) */
#define FMK_COMM_SELF_LOOP 1 /* flags: with world == 1 the rank is its own left and right neighbour (diagnostic) */
int fmk_comm_create(fmk_ctx *ctx, int transport, const char *rendezvous_path, int rank, int world, int flags,
                    size_t ring_bytes /* host transport: bytes of each rank's receive ring, 0 = 1 MiB */,
                    double timeout_s, fmk_comm **out);
int fmk_comm_destroy(fmk_comm *comm);
const char *fmk_comm_last_error(const fmk_comm *comm);
/* Set-up phase, HOST buffers, blocking: `bytes` (<= 4096) from every rank, in rank order; barrier = empty gather. */
int fmk_comm_allgather(fmk_comm *comm, const void *send, size_t bytes, void *recv);
int fmk_comm_barrier(fmk_comm *comm);
/* One exchange: column slice i of `send_*` goes to rank+1 (ignored on the last rank), `recv_*` arrives from rank-1
 * (ignored on rank 0); sizes in bytes, zero-length columns allowed, both sides must agree on them.
 * RCCL: enqueued as ONE ncclGroup on the communicator's stream, which first waits (event) for everything already
 * enqueued on the context's stream; returns at once.  fmk_comm_wait_dev makes the context's stream wait (event) for
 * that exchange -- kernels enqueued between the two calls overlap it.  HOST: synchronous, complete on return. */
int fmk_comm_halo_exchange_dev(fmk_comm *comm, int n_cols, const void *const *send_ptrs, const size_t *send_bytes,
                               void *const *recv_ptrs, const size_t *recv_bytes);
int fmk_comm_wait_dev(fmk_comm *comm);
int fmk_comm_sync(fmk_comm *comm); /* host wait for the communicator's stream, bounded by timeout_s: FMK_E_COMM when the
                                      exchange has not completed by then (call it BEFORE fmk_ctx_sync when a peer may be gone) */
/* Per-exchange timing: the next <= 64 exchanges are timed on their own -- RCCL: a HIP event pair on the communicator's
 * stream around the ncclGroup (device time of the send/recv alone); HOST: wall clock of the staged copy. */
int fmk_comm_profile_enable(fmk_comm *comm, int on);
int fmk_comm_profile_read(fmk_comm *comm, double *ms, int capacity, int *count);
/* First contact with a node, as text: the devices this process sees, the peer-access matrix (hipDeviceCanAccessPeer), the librccl that
 * would be loaded and its version, the environment that decides how RCCL shares memory between processes.  Needs no context and no
 * GPU (`python -m finmlkit_amd.dist --selftest` prints it, bench.py logs it per rank when N > 1). */
int fmk_comm_describe(char *buf, size_t cap);
/* Up to 8 small device column slices copied by one launch on the context's stream (boundary-bar assembly). */
int fmk_copy_cols_dev(fmk_ctx *ctx, int n_cols, const void *const *src, void *const *dst, const size_t *bytes);

/* Diagnostics (bandwidth / latency probes, counters read by tests and tools) are NOT part of this ABI: include/fmk_diag.h. */

#ifdef __cplusplus
}
#endif
#endif /* FMK_H */
