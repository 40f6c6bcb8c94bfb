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
        for (void *p : ptrs) (void)hipFree(p);
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
