// fmk_indexers.hip -- bar close-index builders (finmlkit/bar/logic.py).
//
//   time bars   : float64 clock on the host (exact NumPy emulation, O(1)), then one binary
//                 search per clock edge on the device-resident int64 timestamp column.
//   tick bars   : closed form of the counter loop.
//   volume/dollar bars: see fmk_threshold.hip.
#include <math.h>
#include <stdlib.h>

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
// clock edge then searches the sample (cache hits) and finishes inside one 4096-tick window (32 KB).
// Both searches guess by interpolation before they bisect (tb_last_le): 162 -> ~100 us for the 833 K
// edges of 1e9 ticks, 7 % of the 1-minute headline step (rocprofv3 timeline, round 3).
#define TB_COARSE_SHIFT 12

__global__ __launch_bounds__(256) void k_time_bar_coarse(const int64_t *__restrict__ ts, int64_t n,
                                                         int64_t *__restrict__ coarse, int64_t m, int *__restrict__ clear = nullptr)
{
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (clear && j == 0) *clear = 0;                                   // (a flag word the next launches raise: saves a memset node)
    if (j < m) coarse[j] = ts[j << TB_COARSE_SHIFT];
}

// The last index i in [lo, hi) with at(i) <= edge, given at(lo) = vlo <= edge and -- when hi_real -- at(hi) = vhi > edge (otherwise hi
// is one past the end).  Timestamps are close to evenly spaced between two known points, so the position is first GUESSED by linear
// interpolation and the two ends of a bracket of `radius` around the guess are probed at once (independent loads: one memory
// round trip); the bisection then runs inside the bracket -- or, when the guess was off, in what is left on its side.  Every probe
// keeps the invariant, so the result is the bisection's whatever the spacing.
template <class F>
__device__ __forceinline__ int64_t tb_last_le(F at, int64_t lo, int64_t hi, int64_t vlo, int64_t vhi, bool hi_real, int64_t edge,
                                              int64_t radius)
{
    if (hi_real && hi - lo > 2 * radius + 2 && vhi > vlo) {
        const double f = (double)(edge - vlo) / (double)(vhi - vlo);
        const int64_t g = lo + (int64_t)(f * (double)(hi - lo));
        int64_t a = g - radius, b = g + radius;
        a = a <= lo ? lo + 1 : (a >= hi ? hi - 1 : a);
        b = b >= hi ? hi - 1 : (b <= lo ? lo + 1 : b);
        const int64_t va = at(a), vb = at(b);
        if (va > edge) hi = a;
        else if (vb <= edge) lo = b;
        else { lo = a; hi = b; }
    }
    while (hi - lo > 1) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if (at(mid) <= edge) lo = mid; else hi = mid;
    }
    return lo;
}

// The window search of the two-level indexer with fewer LINES: the indexer is bound by random 64-byte line fetches (~6 per edge with
// tb_last_le's bracket of +-48 around the interpolated position and the bisection inside it).  Here: ONE probe at the interpolated
// position g; the residual (edge - ts[g]) divided by the window's mean gap moves the guess to within a few ticks (the window's own
// drift is gone, what is left is the noise of ~|residual| gaps); both ends of a bracket of +-6 around that are probed (one or two
// neighbouring lines) and the bisection finishes inside it.  Every probe keeps ts[lo] <= edge < ts[hi], so a guess that is off only
// costs probes.  ~3 lines per edge instead of ~6.
template <class F>
__device__ __forceinline__ int64_t tb_last_le_secant(F at, int64_t lo, int64_t hi, int64_t vlo, int64_t vhi, bool hi_real, int64_t edge)
{
    if (hi_real && hi - lo > 32 && vhi > vlo) {
        const double gap = (double)(vhi - vlo) / (double)(hi - lo);
        int64_t g = lo + (int64_t)((double)(edge - vlo) / gap);
        g = g <= lo ? lo + 1 : (g >= hi ? hi - 1 : g);
        const int64_t vg = at(g);
        int64_t g2 = g + (int64_t)floor((double)(edge - vg) / gap);
        if (vg <= edge) lo = g; else hi = g;
        int64_t a = g2 - 6, b = g2 + 6;
        a = a <= lo ? lo + 1 : (a >= hi ? hi - 1 : a);
        b = b >= hi ? hi - 1 : (b <= lo ? lo + 1 : b);
        if (hi - lo > 2) {
            const int64_t va = at(a), vb = at(b);
            if (va > edge) hi = a;
            else if (vb <= edge) lo = b;
            else { lo = a; hi = b; }
        }
    }
    while (hi - lo > 1) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if (at(mid) <= edge) lo = mid; else hi = mid;
    }
    return lo;
}

// searchsorted(ts, edge, side='right') - 1 through the sample table (the body of k_time_bar_index)
__device__ __forceinline__ int64_t tb_index_of(const int64_t *__restrict__ ts, int64_t n, const int64_t *__restrict__ coarse, int64_t m,
                                               int64_t edge, int secant)
{
    const int64_t c_first = coarse[0], c_last = coarse[m - 1];
    if (edge < c_first) return -1;
    const int64_t j = edge >= c_last ? m - 1
                                     : tb_last_le([coarse](int64_t i) { return coarse[i]; }, 0, m - 1, c_first, c_last, true, edge, 8);
    const int64_t lo = j << TB_COARSE_SHIFT;
    const bool hi_real = j + 1 < m;
    const int64_t hi = hi_real ? (j + 1) << TB_COARSE_SHIFT : n;
    const int64_t vlo = coarse[j], vhi = hi_real ? coarse[j + 1] : 0;
    if (secant) return tb_last_le_secant([ts](int64_t i) { return ts[i]; }, lo, hi, vlo, vhi, hi_real, edge);
    return tb_last_le([ts](int64_t i) { return ts[i]; }, lo, hi, vlo, vhi, hi_real, edge, 48);
}

// one thread per clock edge: close_idx[k] = searchsorted(ts, edge_k, side='right') - 1
__global__ __launch_bounds__(256) void k_time_bar_index(const int64_t *__restrict__ ts, int64_t n, int64_t e0,
                                                        int64_t d, int64_t ne, const int64_t *__restrict__ coarse,
                                                        int64_t m, int64_t *__restrict__ clock,
                                                        int64_t *__restrict__ idx, int secant)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ne) return;
    const int64_t edge = e0 + k * d;
    if (clock) clock[k] = edge;
    idx[k] = tb_index_of(ts, n, coarse, m, edge, secant);
}

// Edges [k0, k1) of the clock -- one STAGE of the pipelined time-bar step (fmk_time_bars_ohlcv_dev, fmk_ohlcv.hip): the first stage
// is short and runs in front of the first OHLCV launch, the second runs on the context's auxiliary stream beside it.  The stage also
// answers the question the OHLCV call otherwise has to wait for its main kernel to ask: is any bar longer than `long_min` ticks?
// (bar k = ticks idx[k]+1 .. idx[k+1]: neighbours meet in LDS, the last thread of a workgroup searches edge k+1 itself) -- so the
// host learns it ~2 ms before the main kernel ends and the call returns without waiting for the device.
__global__ __launch_bounds__(256) void k_time_bar_index_stage(const int64_t *__restrict__ ts, int64_t n, int64_t e0, int64_t d,
                                                              int64_t ne, const int64_t *__restrict__ coarse, int64_t m,
                                                              int64_t k0, int64_t k1, int64_t *__restrict__ clock,
                                                              int64_t *__restrict__ idx, int *__restrict__ saw_long, int64_t long_min,
                                                              int secant)
{
    __shared__ int64_t sidx[256];
    // grid-stride over groups of 256 edges: the stage that runs BESIDE an OHLCV launch is given a small grid, so that it does not
    // take the wave slots of the launch it is meant to hide behind
    for (int64_t g = k0 + (int64_t)blockIdx.x * 256; g < k1; g += (int64_t)gridDim.x * 256) {
        const int64_t k = g + threadIdx.x;
        const bool live = k < k1;
        int64_t me = 0;
        if (live) {
            const int64_t edge = e0 + k * d;
            me = tb_index_of(ts, n, coarse, m, edge, secant);
            if (clock) clock[k] = edge;
            idx[k] = me;
        }
        __syncthreads();                                                 // (the previous group's readers are done)
        sidx[threadIdx.x] = me;
        __syncthreads();
        if (live && k + 1 < ne) {
            const int64_t nx = (threadIdx.x + 1 < 256 && k + 1 < k1) ? sidx[threadIdx.x + 1]
                                                                      : tb_index_of(ts, n, coarse, m, e0 + (k + 1) * d, secant);
            if (nx - me > long_min && __hip_atomic_load(saw_long, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                __hip_atomic_store(saw_long, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// the sample table of a column in the context's scratch, on the context's stream (-> *coarse, *m)
int fmk_time_bar_coarse_launch(fmk_ctx *ctx, const int64_t *d_ts, int64_t n, const int64_t **coarse, int64_t *m_out, int *clear)
{
    const int64_t m = fmk_ceil_div(n, (int64_t)1 << TB_COARSE_SHIFT);
    void *scr;
    FMK_TRY(fmk_scratch(ctx, (size_t)m * 8, &scr));
    k_time_bar_coarse<<<(unsigned)fmk_ceil_div(m, 256), 256, 0, ctx->stream>>>(d_ts, n, (int64_t *)scr, m, clear);
    FMK_LAUNCH_CHECK(ctx);
    *coarse = (const int64_t *)scr;
    *m_out = m;
    return FMK_OK;
}

int fmk_time_bar_index_stage(fmk_ctx *ctx, hipStream_t st, const int64_t *d_ts, int64_t n, int64_t e0, int64_t d, int64_t ne,
                             const int64_t *coarse, int64_t m, int64_t k0, int64_t k1, int64_t *d_clock, int64_t *d_idx,
                             int *saw_long, int64_t long_min, int64_t max_blocks)
{
    if (k1 <= k0) return FMK_OK;
    int64_t blocks = fmk_ceil_div(k1 - k0, 256);
    if (max_blocks > 0 && blocks > max_blocks) blocks = max_blocks;
    const int secant = 1;                  // the bracket-of-48 window search
    k_time_bar_index_stage<<<(unsigned)blocks, 256, 0, st>>>(d_ts, n, e0, d, ne, coarse, m, k0, k1, d_clock,
                                                                                  d_idx, saw_long, long_min, secant);
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}

extern "C" int fmk_time_bar_indexer_dev(fmk_ctx *ctx, const int64_t *d_ts, int64_t n, int64_t first_edge,
                                        int64_t delta, int64_t n_edges, int64_t *d_clock, int64_t *d_close_idx)
{
    if (n <= 0 || n_edges < 0) return fmk_set_error(ctx, FMK_E_ARG, "time_bar_indexer: empty input");
    if (n_edges == 0) return FMK_OK;
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    // (an interpolation search on the column without the sample table was measured SLOWER -- 0.235 vs 0.124 ms for the 833 K edges of 1e9
    // ticks, 11.6 vs 2.6 ms for 5e7 one-second edges, profiles/r04_indexer.txt -- the indexer is bound by random line fetches, not by the
    // length of its probe chain; the sample table keeps the first level in L2.  Removed in round 6.)
    const int64_t m = fmk_ceil_div(n, (int64_t)1 << TB_COARSE_SHIFT);
    void *scr;
    FMK_TRY(fmk_scratch(ctx, (size_t)m * 8, &scr));
    int64_t *coarse = (int64_t *)scr;
    k_time_bar_coarse<<<(unsigned)fmk_ceil_div(m, 256), 256, 0, ctx->stream>>>(d_ts, n, coarse, m);
    FMK_LAUNCH_CHECK(ctx);
    const int secant = 1;                  // the bracket-of-48 window search
    k_time_bar_index<<<(unsigned)fmk_ceil_div(n_edges, 256), 256, 0, ctx->stream>>>(d_ts, n, first_edge, delta,
                                                                                   n_edges, coarse, m, d_clock,
                                                                                   d_close_idx, secant);
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
