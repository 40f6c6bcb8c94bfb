// fmk_ticklevel.hip -- tick-level feature loops of finmlkit/feature/core on gfx950.
//
// comp_lagged_returns (core/utils.py:12-64): the reference does one full binary search per tick.
//   The lag index is monotone in the tick index, so a tile of 1024 ticks only ever looks inside
//   [lag(first tick of the tile), last tick]: that slice of `ts` is staged once in LDS with coalesced loads
//   and every tick bisects it there with the reference's float64 comparison
//   `float64(ts[k]) <= float64(ts[i]) - window_ns`.  Windows longer than the 24 KB stage (2048 ticks) fall back to a
//   per-tick backward gallop (1,2,4,... ticks) + bisection on global memory.
//   Traffic: ts 8 (+ window overlap) + close 8 read, 8 written per tick (+ the lagged close, L2 hit).
//
// ewmst / ewmst_mean0 (core/volatility.py:139-219, 72-136): four coupled first-order linear
//   recurrences x' = a_t * x + b_t with a_t = exp(-dt/half_life).  Affine maps compose
//   associatively, so the sequential loop becomes a device-wide scan over (a, a^2, bV, bV2, bSy, bSyy):
//   per-tile aggregate -> hierarchical scan of the tile aggregates -> per-tile rescan with the carry-in.
//   Tiles are loaded coalesced and transposed through LDS so that every thread owns 8 CONSECUTIVE ticks,
//   which it composes in the reference's own order.
//   Traffic: ts 8 + y 8 read twice (aggregate + rescan), 8 written per tick.
#include <math.h>

#include "fmk_common.h"

// ---------------------------------------------------------------------------------------
// comp_lagged_returns
// ---------------------------------------------------------------------------------------
// largest k in [-1, i) with float64(ts[k]) <= target, by backward gallop + bisection on global memory
__device__ __forceinline__ int64_t lr_search_global(const int64_t *__restrict__ ts, int64_t i, double target)
{
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
    return lo;
}

#define LR_TILE 1024          // ticks per workgroup (4 per thread)
#define LR_CAP 3072           // float64 timestamps staged in LDS (24 KB): tile + look-back window

// One workgroup per tile of LR_TILE consecutive ticks.  The lag index is monotone in the tick index, so the
// whole tile searches inside [lag(first tick), last tick]: that range is staged ONCE in LDS (coalesced) and
// every tick bisects it there; only windows longer than the LDS stage fall back to the per-tick global search.
// lag index of the first tick of every tile (one thread per tile: ~14 dependent probes each, all tiles in
// parallel), so that the tile kernel knows where its LDS stage starts without a serial search
__global__ __launch_bounds__(256) void k_lr_tile_start(const int64_t *__restrict__ ts, int64_t n, double w_ns,
                                                       int64_t tiles, int64_t *__restrict__ tile_start)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tiles) return;
    const int64_t i = t * LR_TILE;
    const int64_t lo = lr_search_global(ts, i, (double)ts[i] - w_ns);
    tile_start[t] = lo < 0 ? 0 : lo;
}

__global__ __launch_bounds__(256) void k_lagged_returns(const int64_t *__restrict__ ts,
                                                        const double *__restrict__ close, int64_t n, double w_ns,
                                                        int is_log, const int64_t *__restrict__ tile_start,
                                                        double *__restrict__ out)
{
    __shared__ double s_ts[LR_CAP];                      // float64(ts): the comparison type of the reference
    const double first_full = (double)ts[0] + w_ns;     // utils.py:42 (searchsorted side='left')
    const int64_t i_first = (int64_t)blockIdx.x * LR_TILE;
    const int64_t i_last = (i_first + LR_TILE < n ? i_first + LR_TILE : n) - 1;
    const int64_t r0 = tile_start[blockIdx.x];           // lag(first tick of the tile), clamped to >= 0
    const int64_t len = i_last - r0 + 1;
    const bool staged = len <= LR_CAP;
    if (staged)
        for (int64_t k = threadIdx.x; k < len; k += 256) s_ts[k] = (double)ts[r0 + k];
    __syncthreads();
    // 4 ticks per thread, processed in three unrolled phases so that the four LDS bisections, then the eight
    // global loads, are in flight together (the loop is latency-, not bandwidth-bound)
    constexpr int PER = LR_TILE / 256;
    int64_t lag[PER];
    bool live[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int64_t i = i_first + threadIdx.x + 256 * k;
        lag[k] = -1;
        live[k] = false;
        if (i > i_last) continue;
        const double ti = staged ? s_ts[i - r0] : (double)ts[i];
        if (ti < first_full) continue;                   // i < start_idx -> NaN
        const double target = ti - w_ns;                 // float64, like the reference (utils.py:45)
        if (ti <= target) continue;                      // lag would be >= i -> NaN (utils.py:47)
        live[k] = true;
        // largest j with float64(ts[j]) <= target ; the reference needs 0 <= j < i
        if (staged) {
            int lo = -1, hi = (int)(i - r0);             // staged offsets; f(hi) false
            while (hi - lo > 1) {
                const int mid = lo + ((hi - lo) >> 1);
                if (s_ts[mid] <= target) lo = mid; else hi = mid;
            }
            // lo == -1: nothing in the stage is <= target.  The stage starts at lag(first tick of the tile)
            // (or at tick 0), and lag is monotone, so this only happens when there is no lag at all.
            lag[k] = lo < 0 ? (r0 > 0 ? lr_search_global(ts, i, target) : -1) : r0 + lo;
        } else {
            lag[k] = lr_search_global(ts, i, target);
        }
    }
    double c1[PER], c0[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int64_t i = i_first + threadIdx.x + 256 * k;
        c1[k] = 0.0; c0[k] = 0.0;
        if (live[k] && lag[k] >= 0) { c1[k] = close[i]; c0[k] = close[lag[k]]; }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int64_t i = i_first + threadIdx.x + 256 * k;
        if (i > i_last) continue;
        double r = NAN;
        if (live[k] && lag[k] >= 0) {
            if (c0[k] != 0.0) r = is_log ? log(c1[k] / c0[k]) : c1[k] / c0[k] - 1.0;
            else r = INFINITY;                           // utils.py:57-60
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
    const int64_t blocks = fmk_ceil_div(n, LR_TILE);
    void *scr;
    FMK_TRY(fmk_scratch(ctx, (size_t)blocks * 8, &scr));
    int64_t *tile_start = (int64_t *)scr;
    k_lr_tile_start<<<(unsigned)fmk_ceil_div(blocks, 256), 256, 0, ctx->stream>>>(d_ts, n, return_window_sec * 1e9,
                                                                                 blocks, tile_start);
    FMK_LAUNCH_CHECK(ctx);
    k_lagged_returns<<<(unsigned)blocks, 256, 0, ctx->stream>>>(d_ts, d_close, n, return_window_sec * 1e9, is_log,
                                                              tile_start, d_out);
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

// Coalesced tile load through LDS: thread t loads elements t + 256*r (full 2 KB wavefronts), then reads
// back its 8 CONSECUTIVE ticks (rows padded 8 -> 9 doubles: conflict-free ds_read_b64).  tl[k] / yl[k] are
// tick i0+k (0 beyond n); *tprev0 is ts[i0-1] (ts[0] for the very first tick).
__device__ __forceinline__ void ew_load_tile(const int64_t *__restrict__ ts, const double *__restrict__ y, int64_t n,
                                             int64_t *s_ts, double *s_y, int64_t (&tl)[EW_ITEMS],
                                             double (&yl)[EW_ITEMS], int64_t *tprev0)
{
    const int64_t base = (int64_t)blockIdx.x * EW_TILE;
#pragma unroll
    for (int r = 0; r < EW_ITEMS; ++r) {
        const int e = r * EW_THREADS + threadIdx.x;          // element of the tile
        const int64_t i = base + e;
        const int slot = (e >> 3) * 9 + (e & 7);
        s_ts[slot] = i < n ? ts[i] : 0;
        s_y[slot] = i < n ? y[i] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EW_ITEMS; ++k) {
        tl[k] = s_ts[threadIdx.x * 9 + k];
        yl[k] = s_y[threadIdx.x * 9 + k];
    }
    const int64_t i0 = base + (int64_t)threadIdx.x * EW_ITEMS;
    int64_t tp = 0;
    if (threadIdx.x > 0) tp = s_ts[(threadIdx.x - 1) * 9 + 7];
    else if (i0 >= 1 && i0 - 1 < n) tp = ts[i0 - 1];
    *tprev0 = tp;
}

#define EW_LDS_ELEMS (EW_THREADS * 9)

template <bool MEAN0>
__device__ __forceinline__ EwMap ew_thread_map(const int64_t (&tl)[EW_ITEMS], const double (&yl)[EW_ITEMS],
                                               int64_t tprev0, int64_t n, double half_life)
{
    const int64_t i0 = (int64_t)blockIdx.x * EW_TILE + (int64_t)threadIdx.x * EW_ITEMS;
    EwMap m = ew_identity();
    int64_t tprev = tprev0;
#pragma unroll
    for (int k = 0; k < EW_ITEMS; ++k) {
        const int64_t i = i0 + k;
        if (i >= 1 && i < n) {
            m = ew_compose(m, ew_tick<MEAN0>(tprev, tl[k], yl[k], half_life));
            tprev = tl[k];
        } else if (i == 0 && n > 0) {
            tprev = tl[k];
        }
    }
    return m;
}

template <bool MEAN0>
__global__ __launch_bounds__(EW_THREADS) void k_ew_tile_maps(const int64_t *__restrict__ ts,
                                                             const double *__restrict__ y, int64_t n,
                                                             double half_life, EwMap *__restrict__ tile_map)
{
    __shared__ EwMap lds[4];
    __shared__ int64_t s_ts[EW_LDS_ELEMS];
    __shared__ double s_y[EW_LDS_ELEMS];
    int64_t tl[EW_ITEMS], tprev0;
    double yl[EW_ITEMS];
    ew_load_tile(ts, y, n, s_ts, s_y, tl, yl, &tprev0);
    EwMap m = ew_thread_map<MEAN0>(tl, yl, tprev0, n, half_life);
    EwMap tot;
    (void)ew_block_exclusive(m, lds, &tot);
    if (threadIdx.x == 0) tile_map[blockIdx.x] = tot;
}

// Hierarchical exclusive scan (composition) of an array of maps, in place:
//   k_ew_group_maps : composition of each group of 256 consecutive maps
//   (recursion on the group maps)
//   k_ew_group_apply: exclusive scan inside each group, prefixed by the group's exclusive prefix
__global__ __launch_bounds__(EW_THREADS) void k_ew_group_maps(const EwMap *__restrict__ maps, int64_t m,
                                                              EwMap *__restrict__ group_map)
{
    __shared__ EwMap lds[4];
    const int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x;
    EwMap v = i < m ? maps[i] : ew_identity();
    EwMap tot;
    (void)ew_block_exclusive(v, lds, &tot);
    if (threadIdx.x == 0) group_map[blockIdx.x] = tot;
}

__global__ __launch_bounds__(EW_THREADS) void k_ew_group_apply(EwMap *__restrict__ maps, int64_t m,
                                                               const EwMap *__restrict__ group_pre /* may be null */)
{
    __shared__ EwMap lds[4];
    const int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x;
    EwMap v = i < m ? maps[i] : ew_identity();
    EwMap tot;
    EwMap ex = ew_block_exclusive(v, lds, &tot);
    if (group_pre) ex = ew_compose(group_pre[blockIdx.x], ex);
    if (i < m) maps[i] = ex;
}

static int ew_scan_maps(fmk_ctx *ctx, EwMap *maps, int64_t m, EwMap *work)
{
    const int64_t groups = fmk_ceil_div(m, EW_THREADS);
    if (groups <= 1) {
        k_ew_group_apply<<<1, EW_THREADS, 0, ctx->stream>>>(maps, m, nullptr);
        FMK_LAUNCH_CHECK(ctx);
        return FMK_OK;
    }
    k_ew_group_maps<<<(unsigned)groups, EW_THREADS, 0, ctx->stream>>>(maps, m, work);
    FMK_LAUNCH_CHECK(ctx);
    FMK_TRY(ew_scan_maps(ctx, work, groups, work + groups));
    k_ew_group_apply<<<(unsigned)groups, EW_THREADS, 0, ctx->stream>>>(maps, m, work);
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}

template <bool MEAN0>
__global__ __launch_bounds__(EW_THREADS) void k_ew_apply(const int64_t *__restrict__ ts, const double *__restrict__ y,
                                                         int64_t n, double half_life, double sigma_floor,
                                                         const EwMap *__restrict__ tile_pre,
                                                         double *__restrict__ out)
{
    __shared__ EwMap lds[4];
    __shared__ int64_t s_ts[EW_LDS_ELEMS];
    __shared__ double s_y[EW_LDS_ELEMS];
    int64_t tl[EW_ITEMS], tprev0;
    double yl[EW_ITEMS];
    ew_load_tile(ts, y, n, s_ts, s_y, tl, yl, &tprev0);
    EwMap m = ew_thread_map<MEAN0>(tl, yl, tprev0, n, half_life);
    EwMap tot;
    EwMap ex = ew_block_exclusive(m, lds, &tot);
    ex = ew_compose(tile_pre[blockIdx.x], ex);
    // state entering my first tick (initial state is all-zero, so state = b parts of the prefix map)
    double V = ex.bV, V2 = ex.bV2, Sy = ex.bSy, Syy = ex.bSyy;
    int64_t tprev = tprev0;
    const int64_t i0 = (int64_t)blockIdx.x * EW_TILE + (int64_t)threadIdx.x * EW_ITEMS;
    double res[EW_ITEMS];
#pragma unroll
    for (int k = 0; k < EW_ITEMS; ++k) {
        const int64_t i = i0 + k;
        res[k] = NAN;                                              // volatility.py:174 (out[0])
        if (i >= n) continue;
        if (i == 0) { tprev = tl[k]; continue; }
        ew_step<MEAN0>(V, V2, Sy, Syy, tprev, tl[k], yl[k], half_life);
        tprev = tl[k];
        res[k] = ew_sigma<MEAN0>(V, V2, Sy, Syy, sigma_floor);
    }
    // coalesced store through the (now free) LDS tile
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EW_ITEMS; ++k) s_y[threadIdx.x * 9 + k] = res[k];
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * EW_TILE;
#pragma unroll
    for (int r = 0; r < EW_ITEMS; ++r) {
        const int e = r * EW_THREADS + threadIdx.x;
        const int64_t i = base + e;
        if (i < n) out[i] = s_y[(e >> 3) * 9 + (e & 7)];
    }
}

extern "C" int fmk_ewmst_dev(fmk_ctx *ctx, const int64_t *d_ts, const double *d_y, int64_t n, double half_life,
                             double sigma_floor, int mean0, double *d_out)
{
    if (n <= 0) return FMK_OK;
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t tiles = fmk_ceil_div(n, EW_TILE);
    // tile maps + the (geometrically shrinking) group maps of the hierarchical scan
    int64_t work_maps = 0;
    for (int64_t g = fmk_ceil_div(tiles, EW_THREADS); ; g = fmk_ceil_div(g, EW_THREADS)) {
        work_maps += g;
        if (g <= 1) break;
    }
    void *scr;
    FMK_TRY(fmk_scratch(ctx, (size_t)(tiles + work_maps + 2) * sizeof(EwMap), &scr));
    EwMap *tm = (EwMap *)scr;
    EwMap *work = tm + tiles;
    if (mean0) k_ew_tile_maps<true><<<(unsigned)tiles, EW_THREADS, 0, ctx->stream>>>(d_ts, d_y, n, half_life, tm);
    else k_ew_tile_maps<false><<<(unsigned)tiles, EW_THREADS, 0, ctx->stream>>>(d_ts, d_y, n, half_life, tm);
    FMK_LAUNCH_CHECK(ctx);
    FMK_TRY(ew_scan_maps(ctx, tm, tiles, work));
    if (mean0)
        k_ew_apply<true><<<(unsigned)tiles, EW_THREADS, 0, ctx->stream>>>(d_ts, d_y, n, half_life, sigma_floor, tm, d_out);
    else
        k_ew_apply<false><<<(unsigned)tiles, EW_THREADS, 0, ctx->stream>>>(d_ts, d_y, n, half_life, sigma_floor, tm, d_out);
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}
