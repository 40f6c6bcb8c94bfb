// fmk_ticklevel.hip -- tick-level feature loops of finmlkit/feature/core on gfx950.
//
// comp_lagged_returns (core/utils.py:12-64): the reference does one full binary search per tick.
//   Here every tick gallops BACKWARDS from its own index (1,2,4,... ticks) until the float64
//   comparison `float64(ts[k]) <= float64(ts[i]) - window_ns` flips, then bisects that bracket:
//   2*log2(ticks per window) probes that all land in cache lines its neighbours just touched.
//   Traffic: ts 8 + close 8 read, 8 written per tick (+ the lagged close, L2 hit).
//
// ewmst / ewmst_mean0 (core/volatility.py:139-219, 72-136): four coupled first-order linear
//   recurrences x' = a_t * x + b_t with a_t = exp(-dt/half_life).  Affine maps compose
//   associatively, so the sequential loop becomes a device-wide scan over (a, a^2, bV, bV2, bSy, bSyy):
//   per-tile aggregate -> one block scans the tile aggregates -> per-tile rescan with the carry-in.
//   Inside a thread 8 consecutive ticks are composed in the reference's own order.
//   Traffic: ts 8 + y 8 read twice (aggregate + rescan), 8 written per tick.
#include <math.h>

#include "fmk_common.h"

// ---------------------------------------------------------------------------------------
// comp_lagged_returns
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lagged_returns(const int64_t *__restrict__ ts,
                                                        const double *__restrict__ close, int64_t n, double w_ns,
                                                        int is_log, double *__restrict__ out)
{
    const double first_full = (double)ts[0] + w_ns;     // utils.py:42 (searchsorted side='left')
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double ti = (double)ts[i];
        double r = NAN;
        if (!(ti < first_full)) {                        // i >= start_idx
            const double target = ti - w_ns;             // float64, like the reference (utils.py:45)
            // largest k with float64(ts[k]) <= target ; the reference needs 0 <= k < i
            if (!(ti <= target)) {
                int64_t hi = i;                          // f(hi) false
                int64_t step = 1, lo = i - 1;
                while (lo >= 0 && !((double)ts[lo] <= target)) {
                    hi = lo;
                    step <<= 1;
                    lo = i - step;
                }
                if (lo < 0) lo = -1;                     // f(-1) "true" sentinel
                while (hi - lo > 1) {
                    int64_t mid = lo + ((hi - lo) >> 1);
                    if ((double)ts[mid] <= target) lo = mid; else hi = mid;
                }
                if (lo >= 0) {
                    const double c0 = close[lo];
                    if (c0 != 0.0) r = is_log ? log(close[i] / c0) : close[i] / c0 - 1.0;
                    else r = INFINITY;                   // utils.py:57-60
                }
            }
        }
        out[i] = r;
    }
}

extern "C" int fmk_comp_lagged_returns_dev(fmk_ctx *ctx, const int64_t *d_ts, const double *d_close, int64_t n,
                                           double return_window_sec, int is_log, double *d_out)
{
    if (!(return_window_sec > 0))   // utils.py:33-34
        return fmk_set_error(ctx, FMK_E_ARG, "The return window must be greater than zero.");
    if (n <= 0) return FMK_OK;
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    int64_t blocks = fmk_ceil_div(n, 256);
    const int64_t cap = (int64_t)ctx->n_cu * 32;
    if (blocks > cap) blocks = cap;
    k_lagged_returns<<<(unsigned)blocks, 256, 0, ctx->stream>>>(d_ts, d_close, n, return_window_sec * 1e9, is_log,
                                                              d_out);
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}

// ---------------------------------------------------------------------------------------
// ewmst / ewmst_mean0
// ---------------------------------------------------------------------------------------
struct EwMap {   // x -> a*x + b for the four states (V2 decays with a2 = a*a)
    double a, a2, bV, bV2, bSy, bSyy;
};

__device__ __forceinline__ EwMap ew_identity() { return EwMap{1.0, 1.0, 0.0, 0.0, 0.0, 0.0}; }

// first `f`, then `g`
__device__ __forceinline__ EwMap ew_compose(const EwMap &f, const EwMap &g)
{
    EwMap r;
    r.a = f.a * g.a;
    r.a2 = f.a2 * g.a2;
    r.bV = g.a * f.bV + g.bV;
    r.bV2 = g.a2 * f.bV2 + g.bV2;
    r.bSy = g.a * f.bSy + g.bSy;
    r.bSyy = g.a * f.bSyy + g.bSyy;
    return r;
}

// the reference's per-tick update as a map (volatility.py:176-201 / 110-124)
template <bool MEAN0>
__device__ __forceinline__ EwMap ew_tick(int64_t t_prev, int64_t t_cur, double y, double half_life)
{
    const double dt = (double)(t_cur - t_prev) / 1e9;
    const double alpha = 1.0 - exp(-dt / half_life);
    const double om = 1.0 - alpha;
    const bool nan = isnan(y);
    EwMap m;
    m.a = om;
    m.a2 = om * om;
    if constexpr (MEAN0) {
        m.bV = nan ? 0.0 : alpha;            // V  (weights) : decays only on NaN
        m.bV2 = 0.0;
        m.bSy = 0.0;
        m.bSyy = nan ? 0.0 : alpha * (y * y);   // U
    } else {
        m.bV = alpha;
        m.bV2 = alpha * alpha;
        m.bSy = nan ? 0.0 : alpha * y;
        m.bSyy = nan ? 0.0 : alpha * y * y;
    }
    return m;
}

// sequentially apply one tick to a state, in the reference's exact operation order
template <bool MEAN0>
__device__ __forceinline__ void ew_step(double &V, double &V2, double &Sy, double &Syy, int64_t t_prev, int64_t t_cur,
                                        double y, double half_life)
{
    const double dt = (double)(t_cur - t_prev) / 1e9;
    const double alpha = 1.0 - exp(-dt / half_life);
    const double om = 1.0 - alpha;
    const bool nan = isnan(y);
    if constexpr (MEAN0) {
        if (nan) { Syy = om * Syy; V = om * V; }
        else { Syy = alpha * (y * y) + om * Syy; V = alpha + om * V; }
    } else {
        V = alpha + om * V;
        V2 = alpha * alpha + (om * om) * V2;
        if (nan) { Sy = om * Sy; Syy = om * Syy; }
        else { Sy = alpha * y + om * Sy; Syy = alpha * y * y + om * Syy; }
    }
}

template <bool MEAN0>
__device__ __forceinline__ double ew_sigma(double V, double V2, double Sy, double Syy, double sigma_floor)
{
    if constexpr (MEAN0) {
        double var = V > 0.0 ? Syy / V : NAN;     // volatility.py:127-133
        if (var < 0.0) var = 0.0;
        double s = sqrt(var);
        if (s < sigma_floor) s = sigma_floor;
        return s;
    } else {
        if (!(V > 0.0)) return NAN;               // volatility.py:204-217
        const double mean = Sy / V, e2 = Syy / V;
        const double var_raw = e2 - mean * mean;
        const double denom = V - (V2 / V);
        const double var = (denom > 0.0 && var_raw > 0.0) ? var_raw * (V / denom) : 0.0;
        double s = sqrt(var);
        if (s < sigma_floor) s = sigma_floor;
        return s;
    }
}

#define EW_THREADS 256
#define EW_ITEMS 8
#define EW_TILE (EW_THREADS * EW_ITEMS)

__device__ __forceinline__ EwMap ew_shfl_up(const EwMap &m, int d)
{
    EwMap r;
    r.a = __shfl_up(m.a, d, 64); r.a2 = __shfl_up(m.a2, d, 64);
    r.bV = __shfl_up(m.bV, d, 64); r.bV2 = __shfl_up(m.bV2, d, 64);
    r.bSy = __shfl_up(m.bSy, d, 64); r.bSyy = __shfl_up(m.bSyy, d, 64);
    return r;
}

// inclusive scan of maps across the 256 threads of a block (thread order = tick order);
// returns the EXCLUSIVE prefix for this thread, *block_total = composition of all threads
__device__ __forceinline__ EwMap ew_block_exclusive(const EwMap &mine, EwMap *lds /*[4]*/, EwMap *block_total)
{
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
    EwMap inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        EwMap o = ew_shfl_up(inc, d);
        if (lane >= d) inc = ew_compose(o, inc);
    }
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    EwMap pre = ew_identity();
    for (int k = 0; k < w; ++k) pre = ew_compose(pre, lds[k]);
    EwMap tot = lds[0];
    for (int k = 1; k < 4; ++k) tot = ew_compose(tot, lds[k]);
    *block_total = tot;
    // exclusive prefix of this thread = (waves before) o (lanes before in my wave)
    EwMap prev = ew_shfl_up(inc, 1);
    if (lane == 0) prev = ew_identity();
    __syncthreads();
    return ew_compose(pre, prev);
}

template <bool MEAN0>
__global__ __launch_bounds__(EW_THREADS) void k_ew_tile_maps(const int64_t *__restrict__ ts,
                                                             const double *__restrict__ y, int64_t n,
                                                             double half_life, EwMap *__restrict__ tile_map)
{
    __shared__ EwMap lds[4];
    const int64_t i0 = (int64_t)blockIdx.x * EW_TILE + (int64_t)threadIdx.x * EW_ITEMS;
    EwMap m = ew_identity();
    int64_t tprev = (i0 >= 1 && i0 - 1 < n) ? ts[i0 - 1] : 0;
#pragma unroll
    for (int k = 0; k < EW_ITEMS; ++k) {
        const int64_t i = i0 + k;
        if (i >= 1 && i < n) {
            const int64_t tc = ts[i];
            m = ew_compose(m, ew_tick<MEAN0>(tprev, tc, y[i], half_life));
            tprev = tc;
        } else if (i == 0 && n > 0) {
            tprev = ts[0];
        }
    }
    EwMap tot;
    (void)ew_block_exclusive(m, lds, &tot);
    if (threadIdx.x == 0) tile_map[blockIdx.x] = tot;
}

// one block: exclusive scan (composition) of the tile maps in place
__global__ __launch_bounds__(EW_THREADS) void k_ew_scan_tiles(EwMap *__restrict__ tile_map, int64_t tiles)
{
    __shared__ EwMap lds[4];
    __shared__ EwMap run_s;
    if (threadIdx.x == 0) run_s = ew_identity();
    __syncthreads();
    for (int64_t b = 0; b < tiles; b += EW_THREADS) {
        const int64_t i = b + threadIdx.x;
        EwMap m = i < tiles ? tile_map[i] : ew_identity();
        EwMap tot;
        EwMap ex = ew_block_exclusive(m, lds, &tot);
        EwMap run = run_s;
        if (i < tiles) tile_map[i] = ew_compose(run, ex);
        __syncthreads();
        if (threadIdx.x == 0) run_s = ew_compose(run, tot);
        __syncthreads();
    }
}

template <bool MEAN0>
__global__ __launch_bounds__(EW_THREADS) void k_ew_apply(const int64_t *__restrict__ ts, const double *__restrict__ y,
                                                         int64_t n, double half_life, double sigma_floor,
                                                         const EwMap *__restrict__ tile_pre,
                                                         double *__restrict__ out)
{
    __shared__ EwMap lds[4];
    const int64_t i0 = (int64_t)blockIdx.x * EW_TILE + (int64_t)threadIdx.x * EW_ITEMS;
    // pass 1: my 8-tick map (same as k_ew_tile_maps)
    EwMap m = ew_identity();
    int64_t tprev0 = (i0 >= 1 && i0 - 1 < n) ? ts[i0 - 1] : 0;
    int64_t tl[EW_ITEMS];
    double yl[EW_ITEMS];
    {
        int64_t tprev = tprev0;
#pragma unroll
        for (int k = 0; k < EW_ITEMS; ++k) {
            const int64_t i = i0 + k;
            tl[k] = 0; yl[k] = 0.0;
            if (i < n) { tl[k] = ts[i]; yl[k] = y[i]; }
            if (i >= 1 && i < n) {
                m = ew_compose(m, ew_tick<MEAN0>(tprev, tl[k], yl[k], half_life));
                tprev = tl[k];
            } else if (i == 0 && n > 0) {
                tprev = tl[k];
            }
        }
    }
    EwMap tot;
    EwMap ex = ew_block_exclusive(m, lds, &tot);
    ex = ew_compose(tile_pre[blockIdx.x], ex);
    // state entering my first tick (initial state is all-zero, so state = b parts of the prefix map)
    double V = ex.bV, V2 = ex.bV2, Sy = ex.bSy, Syy = ex.bSyy;
    int64_t tprev = tprev0;
#pragma unroll
    for (int k = 0; k < EW_ITEMS; ++k) {
        const int64_t i = i0 + k;
        if (i >= n) break;
        if (i == 0) { out[0] = NAN; tprev = tl[k]; continue; }    // volatility.py:174
        ew_step<MEAN0>(V, V2, Sy, Syy, tprev, tl[k], yl[k], half_life);
        tprev = tl[k];
        out[i] = ew_sigma<MEAN0>(V, V2, Sy, Syy, sigma_floor);
    }
}

extern "C" int fmk_ewmst_dev(fmk_ctx *ctx, const int64_t *d_ts, const double *d_y, int64_t n, double half_life,
                             double sigma_floor, int mean0, double *d_out)
{
    if (n <= 0) return FMK_OK;
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t tiles = fmk_ceil_div(n, EW_TILE);
    void *scr;
    FMK_TRY(fmk_scratch(ctx, (size_t)tiles * sizeof(EwMap), &scr));
    EwMap *tm = (EwMap *)scr;
    if (mean0) {
        k_ew_tile_maps<true><<<(unsigned)tiles, EW_THREADS, 0, ctx->stream>>>(d_ts, d_y, n, half_life, tm);
        FMK_LAUNCH_CHECK(ctx);
        k_ew_scan_tiles<<<1, EW_THREADS, 0, ctx->stream>>>(tm, tiles);
        FMK_LAUNCH_CHECK(ctx);
        k_ew_apply<true><<<(unsigned)tiles, EW_THREADS, 0, ctx->stream>>>(d_ts, d_y, n, half_life, sigma_floor, tm,
                                                                          d_out);
    } else {
        k_ew_tile_maps<false><<<(unsigned)tiles, EW_THREADS, 0, ctx->stream>>>(d_ts, d_y, n, half_life, tm);
        FMK_LAUNCH_CHECK(ctx);
        k_ew_scan_tiles<<<1, EW_THREADS, 0, ctx->stream>>>(tm, tiles);
        FMK_LAUNCH_CHECK(ctx);
        k_ew_apply<false><<<(unsigned)tiles, EW_THREADS, 0, ctx->stream>>>(d_ts, d_y, n, half_life, sigma_floor, tm,
                                                                           d_out);
    }
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}
