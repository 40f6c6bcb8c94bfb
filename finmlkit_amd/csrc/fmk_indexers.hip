// fmk_indexers.hip -- bar close-index builders (finmlkit/bar/logic.py).
//
//   time bars   : float64 clock on the host (exact NumPy emulation, O(1)), then one binary
//                 search per clock edge on the device-resident int64 timestamp column.
//   tick bars   : closed form of the counter loop.
//   volume/dollar bars: see fmk_threshold.hip.
#include <math.h>

#include "fmk_common.h"

// ---------------------------------------------------------------------------------------
// _time_bar_indexer  logic.py:12-51
// ---------------------------------------------------------------------------------------

// npy_floor_divide for doubles: what `np.int64 // float` evaluates (logic.py:33).
static double npy_floor_divide(double a, double b)
{
    if (b == 0.0) return a / b;
    double mod = fmod(a, b);
    double div = (a - mod) / b;
    if (mod != 0.0) {
        if ((b < 0) != (mod < 0)) { mod += b; div -= 1.0; }
    }
    double fd;
    if (div != 0.0) {
        fd = floor(div);
        if (div - fd > 0.5) fd += 1.0;
    } else {
        fd = copysign(0.0, a / b);
    }
    return fd;
}

extern "C" int fmk_time_bar_clock(int64_t ts_first, int64_t ts_last, double interval_seconds, int64_t *n_edges,
                                  int64_t *first_edge, int64_t *delta)
{
    // bar_interval_ns = interval_seconds * 1e9 is a float64 (logic.py:30)
    double I = interval_seconds * 1e9;
    if (!(I > 0.0)) return fmk_set_error(nullptr, FMK_E_ARG, "interval_seconds must be > 0");
    double start = npy_floor_divide((double)ts_first, I) * I;     // logic.py:33
    double last = ceil((double)ts_last / I) * I;                  // logic.py:36
    double stop = last + I + 1.0;                                 // logic.py:39
    // np.arange(start, stop, I, dtype=int64): len = ceil((stop-start)/I); the int64 fill rule is
    // v[0]=int64(start), v[1]=int64(start+I), v[i]=v[0]+i*(v[1]-v[0]).
    double len = ceil((stop - start) / I);
    if (!(len > 0)) { *n_edges = 0; *first_edge = 0; *delta = 0; return FMK_OK; }
    *n_edges = (int64_t)len;
    *first_edge = (int64_t)start;
    *delta = (int64_t)(start + I) - (int64_t)start;
    return FMK_OK;
}

// Two-level search.  A coarse sample ts[j*4096] (N/4096 entries, L2-resident) is gathered first; every
// clock edge then bisects the sample (cache hits) and finishes inside one 4096-tick window (32 KB):
// ~8 HBM-latency probes per edge instead of ~20.
#define TB_COARSE_SHIFT 12

__global__ __launch_bounds__(256) void k_time_bar_coarse(const int64_t *__restrict__ ts, int64_t n,
                                                         int64_t *__restrict__ coarse, int64_t m)
{
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) coarse[j] = ts[j << TB_COARSE_SHIFT];
}

// one thread per clock edge: close_idx[k] = searchsorted(ts, edge_k, side='right') - 1
__global__ __launch_bounds__(256) void k_time_bar_index(const int64_t *__restrict__ ts, int64_t n, int64_t e0,
                                                        int64_t d, int64_t ne, const int64_t *__restrict__ coarse,
                                                        int64_t m, int64_t *__restrict__ clock,
                                                        int64_t *__restrict__ idx)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ne) return;
    int64_t edge = e0 + k * d;
    // number of sample points <= edge
    int64_t lo = 0, hi = m;
    while (lo < hi) {
        int64_t mid = lo + ((hi - lo) >> 1);
        if (coarse[mid] <= edge) lo = mid + 1; else hi = mid;
    }
    // ts[(lo-1)*4096] <= edge < ts[lo*4096]  ->  the answer lies in that window
    int64_t wlo = lo == 0 ? 0 : ((lo - 1) << TB_COARSE_SHIFT) + 1;
    int64_t whi = lo == 0 ? 0 : (lo << TB_COARSE_SHIFT);
    if (whi > n) whi = n;
    lo = wlo; hi = whi;
    while (lo < hi) {
        int64_t mid = lo + ((hi - lo) >> 1);
        if (ts[mid] <= edge) lo = mid + 1; else hi = mid;
    }
    if (clock) clock[k] = edge;
    idx[k] = lo - 1;
}

extern "C" int fmk_time_bar_indexer_dev(fmk_ctx *ctx, const int64_t *d_ts, int64_t n, int64_t first_edge,
                                        int64_t delta, int64_t n_edges, int64_t *d_clock, int64_t *d_close_idx)
{
    if (n <= 0 || n_edges < 0) return fmk_set_error(ctx, FMK_E_ARG, "time_bar_indexer: empty input");
    if (n_edges == 0) return FMK_OK;
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t m = fmk_ceil_div(n, (int64_t)1 << TB_COARSE_SHIFT);
    void *scr;
    FMK_TRY(fmk_scratch(ctx, (size_t)m * 8, &scr));
    int64_t *coarse = (int64_t *)scr;
    k_time_bar_coarse<<<(unsigned)fmk_ceil_div(m, 256), 256, 0, ctx->stream>>>(d_ts, n, coarse, m);
    FMK_LAUNCH_CHECK(ctx);
    k_time_bar_index<<<(unsigned)fmk_ceil_div(n_edges, 256), 256, 0, ctx->stream>>>(d_ts, n, first_edge, delta,
                                                                                   n_edges, coarse, m, d_clock,
                                                                                   d_close_idx);
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}

// ---------------------------------------------------------------------------------------
// _tick_bar_indexer  logic.py:54-84 -- closed form.
// cum starts at 1 for tick 0 and is reset to 0 after a close, so with t = max(threshold,1)
// the closes are the indices k*t - 1 (k = 1,2,...) that are >= 1, preceded by the entry 0.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tick_bar_index(int64_t t, int64_t skip, int64_t m, int64_t *out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    out[i] = i == 0 ? 0 : (i + skip) * t - 1;
}

extern "C" int fmk_tick_bar_indexer_dev(fmk_ctx *ctx, int64_t n, int64_t threshold, int64_t *d_close_idx,
                                        int64_t capacity, int64_t *n_idx)
{
    if (n <= 0) return fmk_set_error(ctx, FMK_E_ARG, "tick_bar_indexer: empty input");
    int64_t t = threshold < 1 ? 1 : threshold;
    int64_t skip = t == 1 ? 1 : 0;            // k*1-1 = 0 is not a close (loop starts at i = 1)
    int64_t kmax = n / t;                     // largest k with k*t - 1 <= n - 1
    int64_t m = 1 + (kmax - skip > 0 ? kmax - skip : 0);
    *n_idx = m;
    if (!d_close_idx) return FMK_OK;
    if (capacity < m) return fmk_set_error(ctx, FMK_E_CAPACITY, "tick_bar_indexer: capacity %lld < %lld",
                                           (long long)capacity, (long long)m);
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    k_tick_bar_index<<<(unsigned)fmk_ceil_div(m, 256), 256, 0, ctx->stream>>>(t, skip, m, d_close_idx);
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}

// ---------------------------------------------------------------------------------------
// close_ts = timestamps[close_indices]  (kit.py:66, 100, 136)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather_i64(const int64_t *__restrict__ in, int64_t n_in,
                                                    const int64_t *__restrict__ idx, int64_t m,
                                                    int64_t *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    out[i] = in[fmk_wrap(idx[i], n_in)];
}

extern "C" int fmk_gather_i64_dev(fmk_ctx *ctx, const int64_t *d_in, int64_t n_in, const int64_t *d_idx,
                                  int64_t n_idx, int64_t *d_out)
{
    if (n_idx <= 0) return FMK_OK;
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    k_gather_i64<<<(unsigned)fmk_ceil_div(n_idx, 256), 256, 0, ctx->stream>>>(d_in, n_in, d_idx, n_idx, d_out);
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}
