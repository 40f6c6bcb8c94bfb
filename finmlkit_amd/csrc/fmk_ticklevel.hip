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
#include "fmk_log.h"
#include "fmk_exp.h"
#include "fmk_dpp.h"

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
    // 4 CONSECUTIVE ticks per thread: the lag index is monotone in the tick, so only the first tick bisects the whole stage
    // (~12 dependent LDS probes); each of the other three gallops forward from its predecessor's answer (2-4 probes).  (First
    // version: ticks t, t + 256, ... per thread and four independent bisections -- 48 probes per thread, 7.9 ms per 1e9 ticks.)
    constexpr int PER = LR_TILE / 256;
    const int64_t i0 = i_first + (int64_t)threadIdx.x * PER;
    int64_t lag[PER];
    bool live[PER];
    int prev_lo = -1;                                    // staged offset of the previous tick's lag (-1: none yet)
    bool have_prev = false;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int64_t i = i0 + k;
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
            if (have_prev && prev_lo >= 0) {
                // ts is non-decreasing, so is the target: the answer is >= the previous tick's.  Gallop, then bisect.
                lo = prev_lo;
                int step = 1;
                while (lo + step < hi && s_ts[lo + step] <= target) { lo += step; step <<= 1; }
                hi = lo + step < hi ? lo + step : hi;
            }
            while (hi - lo > 1) {
                const int mid = lo + ((hi - lo) >> 1);
                if (s_ts[mid] <= target) lo = mid; else hi = mid;
            }
            prev_lo = lo;
            have_prev = true;
            // lo == -1: nothing in the stage is <= target.  The stage starts at lag(first tick of the tile)
            // (or at tick 0), and lag is monotone, so this only happens when there is no lag at all.
            lag[k] = lo < 0 ? (r0 > 0 ? lr_search_global(ts, i, target) : -1) : r0 + lo;
        } else {
            lag[k] = lr_search_global(ts, i, target);
        }
    }
    double c1[PER], c0[PER];
    const bool vec = i0 + PER - 1 <= i_last && ((uintptr_t)close & 15) == 0 && ((uintptr_t)out & 15) == 0;
    if (vec) {
        const double2 *q = (const double2 *)(close + i0);
#pragma unroll
        for (int k = 0; k < PER / 2; ++k) { const double2 v = q[k]; c1[2 * k] = v.x; c1[2 * k + 1] = v.y; }
    } else {
#pragma unroll
        for (int k = 0; k < PER; ++k) c1[k] = i0 + k <= i_last ? close[i0 + k] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) c0[k] = (live[k] && lag[k] >= 0) ? close[lag[k]] : 0.0;
    double res[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        double r = NAN;
        if (live[k] && lag[k] >= 0) {
            if (c0[k] != 0.0) r = is_log ? fmk_log_ratio(c1[k], c0[k]) : c1[k] / c0[k] - 1.0;
            else r = INFINITY;                           // utils.py:57-60
        }
        res[k] = r;
    }
    if (vec) {
        double2 *q = (double2 *)(out + i0);
#pragma unroll
        for (int k = 0; k < PER / 2; ++k) q[k] = make_double2(res[2 * k], res[2 * k + 1]);
    } else {
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if (i0 + k <= i_last) out[i0 + k] = res[k];
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

// first `f`, then `g`.  Fused multiply-adds (round 6; the library is built with contraction off, for the reference's own expressions):
// the reference has no such operation -- its loop applies one tick at a time, and ew_step restates THAT -- so the composition only
// has to be accurate, and every one of these is issued per tick and pass (6 instructions instead of 10).
__device__ __forceinline__ EwMap ew_compose(const EwMap &f, const EwMap &g)
{
    EwMap r;
    r.a = f.a * g.a;
    r.a2 = f.a2 * g.a2;
    r.bV = fma(g.a, f.bV, g.bV);
    r.bV2 = fma(g.a2, f.bV2, g.bV2);
    r.bSy = fma(g.a, f.bSy, g.bSy);
    r.bSyy = fma(g.a, f.bSyy, g.bSyy);
    return r;
}

// The per-tick update as a map (volatility.py:176-201 / 110-124 / 44-52).  MODE 0: ewmst (volatility.py:139-219)   1: ewmst_mean0 (:72-136)
// 2: ewms (:9-69; fixed alpha, no timestamps -- the four states are Sw, Sw2, Sy, Sy2).  `al` is the tick's alpha = 1 - exp(-dt / half_life)
// for MODE 0 / 1, the fixed 1 - alpha for MODE 2.  A tick that is not there (beyond n, or tick 0 of the time-stamped modes, which the
// reference skips) is handed in as al = 0, y = 0: the map is then EXACTLY the identity and the sequential step
// leaves the state as it is, so the per-tick code carries no range checks (ew_mask_ticks).
template <int MODE>
__device__ __forceinline__ EwMap ew_tick_map(double al, double y)
{
    const bool nan = isnan(y);
    EwMap m;
    if constexpr (MODE == 2) {
        m.a = al;
        m.a2 = al * al;
        m.bV = nan ? 0.0 : 1.0;
        m.bV2 = nan ? 0.0 : 1.0;
        m.bSy = nan ? 0.0 : y;
        m.bSyy = nan ? 0.0 : y * y;
        return m;
    }
    const double om = 1.0 - al;
    m.a = om;
    m.a2 = om * om;
    if constexpr (MODE == 1) {
        m.bV = nan ? 0.0 : al;               // V  (weights) : decays only on NaN
        m.bV2 = 0.0;
        m.bSy = 0.0;
        m.bSyy = nan ? 0.0 : al * (y * y);   // U
    } else {
        m.bV = al;
        m.bV2 = al * al;
        m.bSy = nan ? 0.0 : al * y;
        m.bSyy = nan ? 0.0 : al * y * y;
    }
    return m;
}

// x / v for positive normal v with r = RN(1 / v): the product corrected by its exact remainder (Markstein's last step -- what
// the hardware's own v_div_fmas sequence does, minus the range scaling these operands do not need): the correctly rounded
// quotient, i.e. the IEEE division's result, in 3 instructions instead of ~11.
__device__ __forceinline__ double ew_div(double x, double v, double r)
{
    const double q = x * r;
    return fma(fma(-q, v, x), r, q);
}

// The time constant of MODE 0 / 1 with its correctly rounded reciprocal from the host (r == 0: an odd half life -- zero,
// negative, non-finite, an all-ones significand -- takes the plain divisions); MODE 2: `hl` is the fixed 1 - alpha.
struct EwHl { double hl, r; };

// alpha of the thread's eight ticks, volatility.py:178-179: dt = (t - t_prev) / 1e9; alpha = 1 - exp(-dt / half_life) -- the reference's two
// divisions as correctly rounded quotients (same bits), then THE HOST's exp (csrc/fmk_exp.h: glibc's, restated; round 5 called the device
// library's).  Parity note: for gaps of nanoseconds against a half life of seconds exp(x) is 1 - k * 2^-53 with a single-digit k and the
// reference's alpha carries a relative error of 1e-2 ... 1e-7; a MORE accurate alpha (a polynomial for -expm1 was tried) moves the outputs
// by up to 4e-7 relative and fails the parity fuzz -- the reference's rounding is part of its result.
// The two data-dependent choices are taken ONCE per eight ticks, wave-uniform (round 6; they stood inside the per-tick code, and every
// tick was its own basic block with ~60 register copies between them): the odd half life, and whether every argument of the wave lies in
// exp's table-free range (|x| < ln2 / 256: gaps below 0.27 % of the half life -- 8 instructions per exp instead of ~45).
#define EW_THREADS 256
#define EW_ITEMS 8
#define EW_TILE (EW_THREADS * EW_ITEMS)
__device__ __forceinline__ void ew_alphas(const int64_t (&tl)[EW_ITEMS], int64_t tprev0, EwHl h, double (&al)[EW_ITEMS])
{
    double x[EW_ITEMS];
    int64_t tp = tprev0;
    if (h.r != 0.0) {
#pragma unroll
        for (int k = 0; k < EW_ITEMS; ++k) {
            x[k] = -ew_div(ew_div((double)(tl[k] - tp), 1e9, 1e-9), h.hl, h.r);       // 1e-9 == RN(1 / 1e9)
            tp = tl[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < EW_ITEMS; ++k) {
            x[k] = -(ew_div((double)(tl[k] - tp), 1e9, 1e-9) / h.hl);
            tp = tl[k];
        }
    }
    bool general = false;
#pragma unroll
    for (int k = 0; k < EW_ITEMS; ++k) general |= !(fabs(x[k]) < FMK_EXP_SMALL_BELOW);
    if (__builtin_amdgcn_ballot_w64(general) != 0) {
#pragma unroll
        for (int k = 0; k < EW_ITEMS; ++k) al[k] = 1.0 - fmk_exp_host(x[k]);
    } else {
#pragma unroll
        for (int k = 0; k < EW_ITEMS; ++k) al[k] = 1.0 - fmk_exp_small(x[k]);
    }
}
// 1 / x for a positive normal x, correctly rounded in practice: v_rcp_f64 and two Newton steps
__device__ __forceinline__ double ew_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}

// sequentially apply one tick to a state, in the reference's operation order; `alpha` is the tick's alpha (MODE 2: the fixed
// 1 - alpha) as the map phase computed it -- exp is evaluated once per tick and pass
template <int MODE>
__device__ __forceinline__ void ew_step(double &V, double &V2, double &Sy, double &Syy, double alpha, double y)
{
    const bool nan = isnan(y);
    if constexpr (MODE == 2) {
        const double om = alpha, w = nan ? 0.0 : 1.0;
        V = om * V + w;
        V2 = (om * om) * V2 + w;
        if (nan) { Sy = om * Sy; Syy = om * Syy; }
        else { Sy = om * Sy + y; Syy = om * Syy + y * y; }
        return;
    }
    const double om = 1.0 - alpha;
    if constexpr (MODE == 1) {
        if (nan) { Syy = om * Syy; V = om * V; }
        else { Syy = alpha * (y * y) + om * Syy; V = alpha + om * V; }
    } else {
        V = alpha + om * V;
        V2 = alpha * alpha + (om * om) * V2;
        if (nan) { Sy = om * Sy; Syy = om * Syy; }
        else { Sy = alpha * y + om * Sy; Syy = alpha * y * y + om * Syy; }
    }
}

// sqrt(x) for the closing expressions: the device library's correctly rounded sequence (v_rsq_f64, one coupled step for root and half
// reciprocal root, two residual corrections) without its range scaling and special-value selects -- 10 instructions instead of 21 --
// when every lane of the wave has x in [2^-767, 2^1000] (the library scales below 2^-767); the library's sqrt otherwise (zeros, NaN,
// infinities, subnormal variances).  Same operations on the same operands in range: the same bits.
__device__ __forceinline__ double ew_sqrt(double x)
{
    if (__builtin_amdgcn_ballot_w64(!(x >= 0x1p-767 && x <= 0x1p+1000)) != 0) return sqrt(x);
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    g = fma(fma(-g, g, x), h, g);
    g = fma(fma(-g, g, x), h, g);
    return g;
}

// The closing quotients use ew_div with ONE refined reciprocal per divisor (three quotients by V share it: 5 + 3 x 3 instructions
// instead of three ~11-instruction IEEE divisions).  Correct rounding matters here: e2 - mean * mean and V - V2 / V cancel to
// EXACTLY 0 for a window with one valid sample in the reference, and "== 0" decides between 0.0 / NaN and a 1e-13 residue.
// The reference's closing expressions (volatility.py:54-67 / 127-133 / 204-217).
template <int MODE>
__device__ __forceinline__ double ew_sigma(double V, double V2, double Sy, double Syy, double sigma_floor)
{
    if constexpr (MODE == 2) {                    // volatility.py:54-67
        if (!(V > 0.0)) return NAN;
        const double rV = ew_rcp(V);
        const double mean = ew_div(Sy, V, rV);
        const double den = V - ew_div(V2, V, rV);
        if (!(den > 0.0)) return NAN;
        double var = ew_div((ew_div(Syy, V, rV) - mean * mean) * V, den, ew_rcp(den));
        if (!(var > 0.0)) var = isnan(var) ? var : 0.0;
        return ew_sqrt(var);
    } else if constexpr (MODE == 1) {
        double var = V > 0.0 ? ew_div(Syy, V, ew_rcp(V)) : NAN;     // volatility.py:127-133
        if (var < 0.0) var = 0.0;
        double s = ew_sqrt(var);
        if (s < sigma_floor) s = sigma_floor;
        return s;
    } else {
        if (!(V > 0.0)) return NAN;               // volatility.py:204-217
        const double rV = ew_rcp(V);
        const double mean = ew_div(Sy, V, rV), e2 = ew_div(Syy, V, rV);
        const double var_raw = e2 - mean * mean;
        const double denom = V - ew_div(V2, V, rV);
        const double var = (denom > 0.0 && var_raw > 0.0) ? var_raw * ew_div(V, denom, ew_rcp(denom)) : 0.0;
        double s = ew_sqrt(var);
        if (s < sigma_floor) s = sigma_floor;
        return s;
    }
}


__device__ __forceinline__ EwMap ew_shfl_up(const EwMap &m, int d)
{
    EwMap r;
    r.a = __shfl_up(m.a, d, 64); r.a2 = __shfl_up(m.a2, d, 64);
    r.bV = __shfl_up(m.bV, d, 64); r.bV2 = __shfl_up(m.bV2, d, 64);
    r.bSy = __shfl_up(m.bSy, d, 64); r.bSyy = __shfl_up(m.bSyy, d, 64);
    return r;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ EwMap ew_dpp(const EwMap &m)
{
    return EwMap{fmk_dpp<CTRL, ROW_MASK>(1.0, m.a), fmk_dpp<CTRL, ROW_MASK>(1.0, m.a2), fmk_dpp<CTRL, ROW_MASK>(0.0, m.bV),
                 fmk_dpp<CTRL, ROW_MASK>(0.0, m.bV2), fmk_dpp<CTRL, ROW_MASK>(0.0, m.bSy), fmk_dpp<CTRL, ROW_MASK>(0.0, m.bSyy)};
}

// inclusive scan of maps across the 256 threads of a block (thread order = tick order);
// returns the EXCLUSIVE prefix for this thread, *block_total = composition of all threads
template <int NW = 4>
__device__ __forceinline__ EwMap ew_block_exclusive(const EwMap &mine, EwMap *lds /*[NW]*/, EwMap *block_total)
{
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
    EwMap inc = mine;
    // ordered scan over the lanes on the DPP path (the order of fmk_dpp_iscan; lanes without a source compose with the identity,
    // which is exact).  The shuffle version cost 84 ds_bpermute per 512 ticks.
    inc = ew_compose(ew_dpp<FMK_DPP_ROW_SHR(1), 0xF>(inc), inc);
    inc = ew_compose(ew_dpp<FMK_DPP_ROW_SHR(2), 0xF>(inc), inc);
    inc = ew_compose(ew_dpp<FMK_DPP_ROW_SHR(4), 0xF>(inc), inc);
    inc = ew_compose(ew_dpp<FMK_DPP_ROW_SHR(8), 0xF>(inc), inc);
    inc = ew_compose(ew_dpp<FMK_DPP_ROW_BCAST15, 0xA>(inc), inc);
    inc = ew_compose(ew_dpp<FMK_DPP_ROW_BCAST31, 0xC>(inc), inc);
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    EwMap pre = ew_identity();
    for (int k = 0; k < w; ++k) pre = ew_compose(pre, lds[k]);
    EwMap tot = lds[0];
    for (int k = 1; k < NW; ++k) tot = ew_compose(tot, lds[k]);
    *block_total = tot;
    // exclusive prefix of this thread = (waves before) o (lanes before in my wave)
    EwMap prev = ew_shfl_up(inc, 1);
    if (lane == 0) prev = ew_identity();
    __syncthreads();
    return ew_compose(pre, prev);
}

// The thread's 8 consecutive ticks straight from memory, 16 bytes per load (4 + 4 instructions; every 128-byte line is shared
// by two lanes): no LDS.  A transposing tile in LDS (round 1) cost 36.8 KB per workgroup -- four workgroups per CU -- and
// these kernels are bound by the work they have in flight, not by how their loads coalesce: k_ew_tile_maps 5.5 -> 2.7 ms per 1e9
// ticks (6.0 TB/s), k_ew_apply with direct 16-byte stores as well: see profiles/r02_ewmst_direct_loads.txt.
// `whole`: every tick of the workgroup's tile exists and none of them is tick 0 (uniform over the workgroup: tiles 1 .. the last full one);
// the other tiles read tick by tick with range checks (0 for what is not there).  *tprev0 is ts[i0 - 1] (0 in front of tick 0).
typedef long long ew_l2 __attribute__((ext_vector_type(2), aligned(8)));       // 16-byte accesses on an 8-byte alignment promise:
typedef double ew_d2 __attribute__((ext_vector_type(2), aligned(8)));          // a shard's arrays start one tick before a 64-byte boundary
__device__ __forceinline__ bool ew_whole_tile(int64_t tile, int64_t n, int tile_ticks = EW_TILE) { return tile > 0 && (tile + 1) * tile_ticks <= n; }

__device__ __forceinline__ void ew_load8(const double *__restrict__ src, int64_t i0, int64_t n, bool whole, double (&v)[EW_ITEMS])
{
    if (whole) {
        const ew_d2 *q = (const ew_d2 *)(src + i0);
#pragma unroll
        for (int k = 0; k < EW_ITEMS / 2; ++k) { const ew_d2 t = q[k]; v[2 * k] = t.x; v[2 * k + 1] = t.y; }
    } else {
#pragma unroll
        for (int k = 0; k < EW_ITEMS; ++k) v[k] = i0 + k < n ? src[i0 + k] : 0.0;
    }
}
__device__ __forceinline__ void ew_load_ticks(const int64_t *__restrict__ ts, const double *__restrict__ y, int64_t i0, int64_t n, bool whole,
                                              int64_t (&tl)[EW_ITEMS], double (&yl)[EW_ITEMS], int64_t *tprev0)
{
    if (!ts) {                                                      // ewms: no timestamps
#pragma unroll
        for (int k = 0; k < EW_ITEMS; ++k) tl[k] = 0;
        *tprev0 = 0;
    } else if (whole) {
        const ew_l2 *q = (const ew_l2 *)(ts + i0);
#pragma unroll
        for (int k = 0; k < EW_ITEMS / 2; ++k) { const ew_l2 v = q[k]; tl[2 * k] = v.x; tl[2 * k + 1] = v.y; }
        *tprev0 = ts[i0 - 1];
    } else {
#pragma unroll
        for (int k = 0; k < EW_ITEMS; ++k) tl[k] = i0 + k < n ? ts[i0 + k] : 0;
        *tprev0 = (i0 >= 1 && i0 - 1 < n) ? ts[i0 - 1] : 0;
    }
    ew_load8(y, i0, n, whole, yl);
}
// the thread's 8 consecutive results as four 16-byte stores
__device__ __forceinline__ void ew_store8(double *__restrict__ dst, int64_t i0, int64_t n, bool whole, const double (&v)[EW_ITEMS])
{
    if (whole) {
        ew_d2 *q = (ew_d2 *)(dst + i0);
#pragma unroll
        for (int k = 0; k < EW_ITEMS / 2; ++k) { ew_d2 t; t.x = v[2 * k]; t.y = v[2 * k + 1]; q[k] = t; }
    } else {
#pragma unroll
        for (int k = 0; k < EW_ITEMS; ++k)
            if (i0 + k < n) dst[i0 + k] = v[k];
    }
}

// a tile that is not `whole`: the ticks that are not there become the identity (see ew_tick_map).  ewms needs none of it: it has no
// skipped first tick and no shard entry point, so what the ticks beyond n do to the last tile's aggregate is never read.
template <int MODE>
__device__ __forceinline__ void ew_mask_ticks(int64_t i0, int64_t n, double (&al)[EW_ITEMS], double (&yl)[EW_ITEMS])
{
    if constexpr (MODE == 2) return;
#pragma unroll
    for (int k = 0; k < EW_ITEMS; ++k) {
        const int64_t i = i0 + k;
        if (!(i >= 1 && i < n)) { al[k] = 0.0; yl[k] = 0.0; }
    }
}

// The thread's ticks of tile `tile`: yl[] the values, al[] the alphas (MODE 2: the fixed 1 - alpha) -- kept for the apply phase, exp is
// evaluated once per tick and pass -- and the composition of the eight tick maps.
template <int MODE, int THREADS = EW_THREADS>
__device__ __forceinline__ EwMap ew_thread_ticks(const int64_t *__restrict__ ts, const double *__restrict__ y, int64_t tile, int64_t n,
                                                 EwHl half_life, double (&yl)[EW_ITEMS], double (&al)[EW_ITEMS])
{
    const int64_t i0 = tile * (THREADS * EW_ITEMS) + (int64_t)threadIdx.x * EW_ITEMS;
    const bool whole = ew_whole_tile(tile, n, THREADS * EW_ITEMS);
    {
        int64_t tl[EW_ITEMS], tprev0;
        ew_load_ticks(MODE == 2 ? nullptr : ts, y, i0, n, whole, tl, yl, &tprev0);
        if constexpr (MODE == 2) {
#pragma unroll
            for (int k = 0; k < EW_ITEMS; ++k) al[k] = half_life.hl;
        } else
            ew_alphas(tl, tprev0, half_life, al);
    }
    if (!whole) ew_mask_ticks<MODE>(i0, n, al, yl);
    EwMap m = ew_tick_map<MODE>(al[0], yl[0]);
#pragma unroll
    for (int k = 1; k < EW_ITEMS; ++k) m = ew_compose(m, ew_tick_map<MODE>(al[k], yl[k]));
    return m;
}

// ... and the sequential pass over them from the state (V, V2, Sy, Syy) entering the thread's first tick: the reference's update in
// its own operation order, its closing expression per tick, the results stored.  out[0] = NaN (volatility.py:174).
template <int MODE, int THREADS = EW_THREADS>
__device__ __forceinline__ void ew_thread_apply(double V, double V2, double Sy, double Syy, const double (&yl)[EW_ITEMS],
                                                const double (&al)[EW_ITEMS], double sigma_floor, int64_t tile, int64_t n,
                                                double *__restrict__ out)
{
    const int64_t i0 = tile * (THREADS * EW_ITEMS) + (int64_t)threadIdx.x * EW_ITEMS;
    const bool whole = ew_whole_tile(tile, n, THREADS * EW_ITEMS);
    double res[EW_ITEMS];
#pragma unroll
    for (int k = 0; k < EW_ITEMS; ++k) {
        // the products of alpha and y that the tick map formed in front of the scan are formed again here (5 instructions per tick):
        // kept, they are 80 registers live across the scan -- the compiler spilled them
        double a = al[k], yy = yl[k];
        if constexpr (MODE == 2) asm volatile("" : "+v"(yy));        // ewms: al[] is one constant
        else asm volatile("" : "+v"(a), "+v"(yy));
        ew_step<MODE>(V, V2, Sy, Syy, a, yy);
        res[k] = ew_sigma<MODE>(V, V2, Sy, Syy, sigma_floor);
    }
    if (MODE != 2 && i0 == 0) res[0] = NAN;
    ew_store8(out, i0, n, whole, res);
}

template <int MODE>
__global__ __launch_bounds__(EW_THREADS) void k_ew_tile_maps(const int64_t *__restrict__ ts,
                                                             const double *__restrict__ y, int64_t n,
                                                             EwHl half_life, EwMap *__restrict__ tile_map)
{
    __shared__ EwMap lds[4];
    double yl[EW_ITEMS], al[EW_ITEMS];
    const EwMap m = ew_thread_ticks<MODE>(ts, y, blockIdx.x, n, half_life, yl, al);
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

// the state entering the thread's first tick: the prefix map applied to the initial state (all-zero unless this is a shard of a longer
// series: then the state the earlier shards leave behind, fmk_ewmst_shard_*)
__device__ __forceinline__ void ew_enter(const EwMap &ex, const double *__restrict__ state_in, double &V, double &V2, double &Sy, double &Syy)
{
    V = ex.bV; V2 = ex.bV2; Sy = ex.bSy; Syy = ex.bSyy;
    if (state_in) {
        V = fma(ex.a, state_in[0], ex.bV); V2 = fma(ex.a2, state_in[1], ex.bV2);
        Sy = fma(ex.a, state_in[2], ex.bSy); Syy = fma(ex.a, state_in[3], ex.bSyy);
    }
}

template <int MODE>
// five waves per SIMD (96 VGPRs, no spill): 7.47 ms per 1e9 ticks; six (80 VGPRs, 8 spill instructions): 7.6; four: 7.46 (round 6)
__global__ __launch_bounds__(EW_THREADS, 5) void k_ew_apply(const int64_t *__restrict__ ts, const double *__restrict__ y,
                                                         int64_t n, EwHl half_life, double sigma_floor,
                                                         const EwMap *__restrict__ tile_pre,
                                                         const double *__restrict__ state_in,
                                                         double *__restrict__ out)
{
    __shared__ EwMap lds[4];
    double yl[EW_ITEMS], al[EW_ITEMS];
    const EwMap m = ew_thread_ticks<MODE>(ts, y, blockIdx.x, n, half_life, yl, al);
    EwMap tot;
    EwMap ex = ew_block_exclusive(m, lds, &tot);
    ex = ew_compose(tile_pre[blockIdx.x], ex);
    double V, V2, Sy, Syy;
    ew_enter(ex, state_in, V, V2, Sy, Syy);
    ew_thread_apply<MODE>(V, V2, Sy, Syy, yl, al, sigma_floor, blockIdx.x, n, out);
}

// ---------------------------------------------------------------------------------------------------------------------
// ONE pass (ewmst / ewmst_mean0 / ewms): tile aggregates + decoupled look-back + apply in the same kernel, so ts and y are
// read once (16 B/tick instead of 2 x 16) and exp() is evaluated once per tick (the per-tick decay factors stay in
// registers between the map phase and the apply phase).  Inter-workgroup hand-off as the CDNA4 guide prescribes (Guideline
// 16, form R2: the data IS the flag): a tile's aggregate map and, later, its inclusive prefix map are published as twelve
// 8-byte {tag, 32-bit word} granules each with relaxed agent-scope stores; a consumer re-reads the twelve granules until
// every tag matches -- no fences, no flags, nothing that depends on XCD placement.  Spins are bounded: a tile that never shows up
// raises a sticky error word in pinned host memory (reported by the next fmk_ctx_sync / fmk_d2h) instead of hanging the device.
#define EW_SPIN_LIMIT (1u << 22)
#define EW_W1 64                         // tiles one R record covers

// Round 5: the look-back in TWO memory round trips whatever the number of tiles in flight (the scheme of fmk_dollar_onepass.h, for maps
// instead of sums).  Round 2's version walked back one tile per dependent cross-XCD round trip: 25.9 ms per 1e9 ticks against 16.0 for
// the two passes.  Per tile three records of six doubles, each written ONCE, each with its own tag word in a compact tag array (a poll
// reads 64 consecutive tags, not 64 records):
//   A[t]  the tile's own map                      R[t]  the composition of A over the 64 tiles in front of t
//   P[t]  the composition of A over ALL tiles in front of t (its exclusive prefix)
// Writer: the record's granules, then the tag, all relaxed.  Reader: the tag (polled), then the granules until each is valid.
// Lane l of the look-back waits for A[t - 1 - l] and, at the same time, for P or R of tile u_l = t - 64 (l + 1); R(u_l) covers the tiles
// [u_l - 64, u_l), so prefix(t) = P(u_f) then R(u_{f-1}) ... R(u_0) then R(t), f the nearest lane whose tile already has its prefix.
// R and P lie residue-major (slot = (t mod 64) * groups + t / 64): the 64 tags one look-back polls are consecutive.
struct EwDesc {
    unsigned long long *tagA, *tagR, *tagP;
    unsigned long long *A, *R, *P;       // [tiles][12] granules (R, P: by slot)
    unsigned long long *ticket;          // the next tile to hand out (zeroed with the tags)
    int64_t groups;
};
__device__ __forceinline__ int64_t ew_slot(int64_t t, int64_t groups) { return (t & (EW_W1 - 1)) * groups + t / EW_W1; }

// A record is twelve {valid, 32-bit word} granules (the data IS the flag: no fence -- a release / acquire fence at agent scope writes
// back / invalidates the whole L2, which cost this kernel 59 ms per 1e9 ticks when every tile did two of each); the compact tag only
// says "worth reading now".
__device__ __forceinline__ void ew_publish(unsigned long long *rec, unsigned long long *tag, const EwMap &m, int lane)
{
    if (lane < 12) {
        const int f = lane >> 1;
        const double d = f == 0 ? m.a : f == 1 ? m.a2 : f == 2 ? m.bV : f == 3 ? m.bV2 : f == 4 ? m.bSy : m.bSyy;
        const unsigned long long bits = (unsigned long long)__double_as_longlong(d);
        const unsigned word = (lane & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
        __hip_atomic_store(rec + lane, (1ULL << 32) | word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) __hip_atomic_store(tag, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// this lane's record (the tag said it is there): read until every granule is valid
__device__ __forceinline__ EwMap ew_fetch(const unsigned long long *rec)
{
    unsigned long long g[12];
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            g[k] = __hip_atomic_load(rec + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = ok && (g[k] >> 32) == 1ULL;
        }
        if (ok) break;
        __builtin_amdgcn_s_sleep(1);
    }
    auto dbl = [&](int f) { return __longlong_as_double((long long)((g[2 * f + 1] << 32) | (g[2 * f] & 0xFFFFFFFFULL))); };
    return EwMap{dbl(0), dbl(1), dbl(2), dbl(3), dbl(4), dbl(5)};
}
// lane 0 holds the NEWEST map, lane 63 the oldest: the composition oldest first, wave-uniform
__device__ __forceinline__ EwMap ew_wave_compose_newest_first(EwMap x)
{
    x = ew_compose(x, ew_dpp<FMK_DPP_ROW_SHR(1), 0xF>(x));
    x = ew_compose(x, ew_dpp<FMK_DPP_ROW_SHR(2), 0xF>(x));
    x = ew_compose(x, ew_dpp<FMK_DPP_ROW_SHR(4), 0xF>(x));
    x = ew_compose(x, ew_dpp<FMK_DPP_ROW_SHR(8), 0xF>(x));
    x = ew_compose(x, ew_dpp<FMK_DPP_ROW_BCAST15, 0xA>(x));
    x = ew_compose(x, ew_dpp<FMK_DPP_ROW_BCAST31, 0xC>(x));
    return EwMap{fmk_last_lane(x.a), fmk_last_lane(x.a2), fmk_last_lane(x.bV), fmk_last_lane(x.bV2), fmk_last_lane(x.bSy), fmk_last_lane(x.bSyy)};
}

// wave 0 of tile `tile` > 0, after its own map went out: the exclusive prefix map of the tile; false: a wait gave up
__device__ __forceinline__ bool ew_lookback(const EwDesc &D, int64_t tile, int lane, EwMap *out)
{
    const int64_t ia = tile - 1 - lane, ib = tile - EW_W1 * ((int64_t)lane + 1);
    bool have_a = ia < 0, have_p = ib <= 0 || tile <= EW_W1, have_r = false;     // nothing there: the identity
    bool published = false;
    EwMap r = ew_identity();
    for (unsigned polls = 1;; ++polls) {
        if (!have_a) have_a = __hip_atomic_load(D.tagA + ia, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        if (!have_p) {
            const int64_t sl = ew_slot(ib, D.groups);
            have_p = __hip_atomic_load(D.tagP + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            if (!have_p && !have_r) have_r = __hip_atomic_load(D.tagR + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        }
        if (!published && __ballot(!have_a) == 0) {
            r = ew_wave_compose_newest_first(ia >= 0 ? ew_fetch(D.A + 12 * ia) : ew_identity());    // R(tile)
            published = true;
            if (tile > EW_W1) ew_publish(D.R + 12 * ew_slot(tile, D.groups), D.tagR + ew_slot(tile, D.groups), r, lane);
        }
        if (published) {
            if (tile <= EW_W1) { *out = r; return true; }            // the window reached tile 0: R is the prefix
            const uint64_t based = __ballot(have_p), missing = __ballot(!have_p && !have_r);
            if (based != 0) {
                const int first = __builtin_ctzll(based);             // the nearest tile that has its prefix
                if ((missing & ((1ULL << first) - 1)) == 0) {
                    EwMap x = ew_identity();
                    if (lane <= first && ib > 0) x = ew_fetch((lane == first ? D.P : D.R) + 12 * ew_slot(ib, D.groups));
                    *out = ew_compose(ew_wave_compose_newest_first(x), r);
                    return true;
                }
            }
        }
        if (polls > EW_SPIN_LIMIT) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

// ONE TILE PER WORKGROUP in dispatch order, the tile's ticks by direct 16-byte loads (round 5; round 2's form was a persistent grid with
// the tile staged in LDS: every workgroup reached its look-back at the same moment and the whole device waited out the two round trips,
// generation after generation -- 18.7 ms per 1e9 ticks, removed in round 6).  Here the workgroups of a CU are at different points of
// their tiles and the SIMDs stay busy with the others' arithmetic while one waits.  Forward progress: a workgroup takes its tile from an atomic ticket, so every tile it can
// wait for has started; a wait that gives up all the same raises the sticky error word instead of hanging.
// Four waves per SIMD (112 .. 128 registers, no spill): 9.5 ms per 1e9 ticks; five: 13.2 ms, six: 15.6 ms (they spill).  THREADS: the
// workgroup = the tile (x 8 ticks); 256 is the measured optimum (64 / 128 / 512 / 1024: 24.1 / 18.7 / 12.3 / 21.9 ms).
template <int MODE, int THREADS, int OCC>
__global__ __launch_bounds__(THREADS, OCC) void k_ew_onepass_d(const int64_t *__restrict__ ts, const double *__restrict__ y, int64_t n,
                                                                EwHl half_life, double sigma_floor, const double *__restrict__ state_in,
                                                                double *__restrict__ out, EwDesc D, int64_t *err_word)
{
    __shared__ EwMap lds[THREADS / 64];
    __shared__ EwMap s_excl;
    __shared__ unsigned long long s_ticket;
    const int lane = fmk_lane();
    if (threadIdx.x == 0) s_ticket = atomicAdd(D.ticket, 1ULL);       // a ticket, not blockIdx.x: every earlier tile has started (k_dl1)
    __syncthreads();
    const int64_t tile = (int64_t)s_ticket;
    double yl[EW_ITEMS], al[EW_ITEMS];
    const EwMap m = ew_thread_ticks<MODE, THREADS>(ts, y, tile, n, half_life, yl, al);
    EwMap tot;
    EwMap ex = ew_block_exclusive<THREADS / 64>(m, lds, &tot);
    if (threadIdx.x < 64) {
        EwMap excl = ew_identity();
        ew_publish(D.A + 12 * tile, D.tagA + tile, tot, lane);
        if (tile > 0 && !ew_lookback(D, tile, lane, &excl)) {
            if (lane == 0) *err_word = 1;
            excl = ew_identity();
        }
        ew_publish(D.P + 12 * ew_slot(tile, D.groups), D.tagP + ew_slot(tile, D.groups), excl, lane);
        if (lane == 0) s_excl = excl;
    }
    __syncthreads();
    ex = ew_compose(s_excl, ex);
    double V, V2, Sy, Syy;
    ew_enter(ex, state_in, V, V2, Sy, Syy);
    ew_thread_apply<MODE, THREADS>(V, V2, Sy, Syy, yl, al, sigma_floor, tile, n, out);
}

// composition of all tile maps in order (ONE block): the map of the whole series, x -> a*x + b per state
__global__ __launch_bounds__(EW_THREADS) void k_ew_total(const EwMap *__restrict__ maps, int64_t m, double *out6)
{
    __shared__ EwMap lds[4];
    EwMap run = ew_identity();
    for (int64_t b = 0; b < m; b += EW_THREADS) {
        const int64_t i = b + threadIdx.x;
        EwMap v = i < m ? maps[i] : ew_identity();
        EwMap tot;
        (void)ew_block_exclusive(v, lds, &tot);
        run = ew_compose(run, tot);
    }
    if (threadIdx.x == 0) {
        out6[0] = run.a; out6[1] = run.a2; out6[2] = run.bV; out6[3] = run.bV2; out6[4] = run.bSy; out6[5] = run.bSyy;
    }
}

// d_map_out != nullptr: only the map of the series is produced (6 doubles on the device); else the outputs,
// starting from d_state_in (4 doubles on the device, nullptr = zeros)
template <int MODE>
static int ew_run(fmk_ctx *ctx, const int64_t *d_ts, const double *d_y, int64_t n, double half_life,
                  double sigma_floor, double *d_out, const double *d_state_in = nullptr, double *d_map_out = nullptr)
{
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    // the kernels take the decay RATE for the time-stamped modes (ew_alpha), the fixed 1 - alpha for ewms
    // MODE 0 / 1: the half life with its correctly rounded reciprocal (ew_alpha); MODE 2: the fixed 1 - alpha
    EwHl hl{half_life, 0.0};
    if (MODE != 2 && half_life > 0.0 && half_life < 1e300 && half_life > 1e-300) {
        unsigned long long bits;
        memcpy(&bits, &half_life, 8);
        if ((bits & 0xFFFFFFFFFFFFFULL) != 0xFFFFFFFFFFFFFULL) hl.r = 1.0 / half_life;
    }
    const int64_t tiles = fmk_ceil_div(n, EW_TILE);
    // tile maps + the (geometrically shrinking) group maps of the hierarchical scan
    int64_t work_maps = 0;
    for (int64_t g = fmk_ceil_div(tiles, EW_THREADS); ; g = fmk_ceil_div(g, EW_THREADS)) {
        work_maps += g;
        if (g <= 1) break;
    }
    void *scr;
    // The one-pass kernel (16 + 8 B/tick instead of 2 x 16 + 8) is correct -- the same suite runs through it -- and SLOWER at every size
    // measured, every round: 25.9 vs 16.0 ms (round 2, one predecessor per round trip), 9.3 vs 9.7 (round 5, two-level look-back), 9.5 vs
    // 8.0 ms (round 6: the two passes lost more instructions than it did).  A workgroup's four waves sit out two dependent round trips
    // between its map phase and its apply phase, and at 112 .. 128 registers only four workgroups share a CU to cover for each other;
    // k_ew_tile_maps meanwhile runs at the HBM rate (16 GB in 2.5 ms).  profiles/r06_ewmst.txt.  The two passes stay the default;
    // FMK_EW_ONE_PASS=1 (read per call) selects the one-pass kernel.
    const char *opv = getenv("FMK_EW_ONE_PASS");
    if (!d_map_out && opv && atoi(opv)) {
        // one pass: two-level look-back over tagged records (the tags are zeroed before every launch)
        // (workgroups of 64 / 128 / 512 / 1024 threads -- tiles of 512 .. 8 192 ticks -- were measured: 24.1 / 18.7 / 12.3 / 21.9 ms against 9.5)
        const int64_t otiles = tiles;
        const int64_t groups = fmk_ceil_div(otiles, EW_W1), slots = groups * EW_W1;
        const size_t tag_bytes = (size_t)(otiles + 2 * slots) * 8, rec_bytes = (size_t)(otiles + 2 * slots) * 96;
        FMK_TRY(fmk_scratch(ctx, tag_bytes + rec_bytes + 64, &scr));
        FMK_HIP(ctx, hipMemsetAsync(scr, 0, tag_bytes + rec_bytes + 8, ctx->stream)); // tags, granules, the ticket counter
        EwDesc D;
        D.tagA = (unsigned long long *)scr; D.tagR = D.tagA + otiles; D.tagP = D.tagR + slots;
        D.A = D.tagP + slots; D.R = D.A + 12 * otiles; D.P = D.R + 12 * slots;
        D.groups = groups;
        D.ticket = (unsigned long long *)((char *)scr + tag_bytes + rec_bytes);
        k_ew_onepass_d<MODE, EW_THREADS, 4><<<(unsigned)otiles, EW_THREADS, 0, ctx->stream>>>(d_ts, d_y, n, hl, sigma_floor, d_state_in, d_out, D, ctx->h_mail + 40);
        FMK_LAUNCH_CHECK(ctx);
        return FMK_OK;
    }
    FMK_TRY(fmk_scratch(ctx, (size_t)(tiles + work_maps + 2) * sizeof(EwMap), &scr));
    EwMap *tm = (EwMap *)scr;
    EwMap *work = tm + tiles;
    // (round 4 tried leaving the alphas of the map pass in d_out for the apply pass: what exp saved the 8 GB of stores cost -- profiles/r04_ewmst.txt)
    k_ew_tile_maps<MODE><<<(unsigned)tiles, EW_THREADS, 0, ctx->stream>>>(d_ts, d_y, n, hl, tm);
    FMK_LAUNCH_CHECK(ctx);
    if (d_map_out) {
        k_ew_total<<<1, EW_THREADS, 0, ctx->stream>>>(tm, tiles, d_map_out);
        FMK_LAUNCH_CHECK(ctx);
        return FMK_OK;
    }
    FMK_TRY(ew_scan_maps(ctx, tm, tiles, work));
    k_ew_apply<MODE><<<(unsigned)tiles, EW_THREADS, 0, ctx->stream>>>(d_ts, d_y, n, hl, sigma_floor, tm, d_state_in, d_out);
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}

// diagnostics (include/fmk_diag.h): the device's exp over an array -- tests compare it with the host's exp() bit for bit
__global__ void k_diag_exp(const double *__restrict__ x, int64_t n, double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fmk_exp_host(x[i]);
}
extern "C" int fmk_diag_exp_dev(fmk_ctx *ctx, const double *d_x, int64_t n, double *d_out)
{
    if (n <= 0) return FMK_OK;
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    k_diag_exp<<<(unsigned)fmk_ceil_div(n, 256), 256, 0, ctx->stream>>>(d_x, n, d_out);
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}

extern "C" int fmk_ewmst_dev(fmk_ctx *ctx, const int64_t *d_ts, const double *d_y, int64_t n, double half_life,
                             double sigma_floor, int mean0, double *d_out)
{
    if (n <= 0) return FMK_OK;
    return mean0 ? ew_run<1>(ctx, d_ts, d_y, n, half_life, sigma_floor, d_out)
                 : ew_run<0>(ctx, d_ts, d_y, n, half_life, sigma_floor, d_out);
}

/* Shards of one series (multi-GPU): tick 0 of the arrays is the LAST tick of the previous shard (it only provides the
 * previous timestamp; rank 0 passes its own arrays, whose tick 0 the reference skips anyway).
 *   fmk_ewmst_shard_map_dev  : the affine map (a, a2, bV, bV2, bSy, bSyy) of ticks 1..n-1 -> d_map_out[6] (device)
 *   fmk_ewmst_shard_apply_dev: outputs for ticks 1..n-1 starting from d_state_in[4] = (V, V2, Sy, Syy) (device);
 *                              d_out[0] = NaN.
 * The caller composes the maps of the earlier shards (x -> a*x + b, in shard order) into its incoming state. */
extern "C" int fmk_ewmst_shard_map_dev(fmk_ctx *ctx, const int64_t *d_ts, const double *d_y, int64_t n, double half_life,
                                       int mean0, double *d_map_out)
{
    if (n <= 0 || !d_map_out) return fmk_set_error(ctx, FMK_E_ARG, "ewmst_shard_map: bad arguments");
    return mean0 ? ew_run<1>(ctx, d_ts, d_y, n, half_life, 0.0, nullptr, nullptr, d_map_out)
                 : ew_run<0>(ctx, d_ts, d_y, n, half_life, 0.0, nullptr, nullptr, d_map_out);
}

extern "C" int fmk_ewmst_shard_apply_dev(fmk_ctx *ctx, const int64_t *d_ts, const double *d_y, int64_t n,
                                         double half_life, double sigma_floor, int mean0, const double *d_state_in,
                                         double *d_out)
{
    if (n <= 0) return FMK_OK;
    return mean0 ? ew_run<1>(ctx, d_ts, d_y, n, half_life, sigma_floor, d_out, d_state_in)
                 : ew_run<0>(ctx, d_ts, d_y, n, half_life, sigma_floor, d_out, d_state_in);
}

__global__ void k_fill_nan(double *out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = NAN;
}

/* ewms (volatility.py:9-69): fixed-alpha exponentially weighted std, alpha = 2/(span+1); span <= 1 -> all NaN */
extern "C" int fmk_ewms_dev(fmk_ctx *ctx, const double *d_y, int64_t n, int64_t span, double *d_out)
{
    if (n <= 0) return FMK_OK;
    if (span <= 1) {
        FMK_HIP(ctx, hipSetDevice(ctx->device));
        k_fill_nan<<<(unsigned)fmk_ceil_div(n, 256), 256, 0, ctx->stream>>>(d_out, n);
        FMK_LAUNCH_CHECK(ctx);
        return FMK_OK;
    }
    const double alpha = 2.0 / ((double)span + 1.0);
    return ew_run<2>(ctx, nullptr, d_y, n, 1.0 - alpha, 0.0, d_out);
}

// ---------------------------------------------------------------------------------------
// realized_vol (volatility.py:256-286): out[i] = sqrt(nansum(r[i-W+1..i]^2) / (valid - is_sample)), NaN unless
// valid > 1 and i >= W-1.
//
// Rolling sums WITHOUT subtraction (a prefix-sum difference would cancel: one outlier return in the prefix destroys
// the digits of a quiet window): the index axis is cut into segments of length W aligned at multiples of W; a
// window [s, i] of length W covers a suffix of s's segment and a prefix of i's segment, so
//     sum = suffix_in_segment[s] + prefix_in_segment[i]        (just prefix[i] when s starts a segment)
// -- only additions of non-negative terms, relative error a few ulp like the reference's pairwise sum.
// W <= RV_MAX_W: one kernel, the region [tile - (W-1), tile end) staged in LDS, both segmented scans in LDS/registers
// (8 B/tick read + 8 written).  Larger W: segment scans through global scratch (one block per segment).
// ---------------------------------------------------------------------------------------
#define RV_THREADS 256
#define RV_C 25                                  // region elements per thread (odd: conflict-free ds_read_b64)
#define RV_R (RV_THREADS * RV_C)                 // 6400 region elements: 51200 B of squares + 12800 B of counts
#define RV_MAX_W 2048
#define RV_SMALL_W 640

// exclusive carry of a segmented sum across the block's threads, in thread order (REV = false) or reversed.
// (s, f) = my chunk's aggregate: f = chunk contains a segment boundary, s = sum of the elements after (before,
// when reversed) the last boundary -- or of the whole chunk if there is none.
template <bool REV>
__device__ __forceinline__ double rv_block_carry(double s, int f, double *w_s, int *w_f)
{
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double os = REV ? __shfl_down(s, d, 64) : __shfl_up(s, d, 64);
        const int of = REV ? __shfl_down(f, d, 64) : __shfl_up(f, d, 64);
        const bool has = REV ? (lane + d < 64) : (lane >= d);
        if (has) {
            if (!f) s = os + s;
            f |= of;
        }
    }
    if (lane == (REV ? 0 : 63)) { w_s[w] = s; w_f[w] = f; }
    __syncthreads();
    // carry entering my wave
    double cs = 0.0;
    if (!REV) { for (int k = 0; k < w; ++k) cs = w_f[k] ? w_s[k] : cs + w_s[k]; }
    else { for (int k = 3; k > w; --k) cs = w_f[k] ? w_s[k] : cs + w_s[k]; }
    // neighbour lane's inclusive value
    double ps = REV ? __shfl_down(s, 1, 64) : __shfl_up(s, 1, 64);
    int pf = REV ? __shfl_down(f, 1, 64) : __shfl_up(f, 1, 64);
    if (lane == (REV ? 63 : 0)) { ps = 0.0; pf = 0; }
    __syncthreads();
    return pf ? ps : cs + ps;
}

// C = region elements per thread: 25 (64 KB of LDS, 249 VGPRs: two workgroups per CU) serves windows up to 2048; windows up to
// RV_SMALL_W take C = 9 (23 KB, 110 VGPRs: four workgroups per CU; 1e9 ticks, W = 20: 8.9 -> 5.9 ms, W = 500: 10.0 -> ~7.5 ms with
// 28 % of halo; C = 5 and C = 15 were measured too and are no better than their neighbours)
template <int C>
__global__ __launch_bounds__(RV_THREADS) void k_realized_vol(const double *__restrict__ r, int64_t n, int64_t W,
                                                             int is_sample, int64_t T, double *__restrict__ out)
{
    __shared__ double s_sq[(RV_THREADS * C)];
    __shared__ unsigned short s_cnt[(RV_THREADS * C)];
    __shared__ double w_s[4];
    __shared__ int w_f[4];
    __shared__ int w_c[4];
    const int tid = threadIdx.x, lane = fmk_lane(), wv = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * T;
    const int64_t g0 = base - (W - 1);                 // global index of region element 0 (may be negative)
    const int R = (int)(T + W - 1);                    // <= (RV_THREADS * C)
    for (int e = tid; e < (RV_THREADS * C); e += RV_THREADS) {
        const int64_t g = g0 + e;
        double v = NAN;
        if (e < R && g >= 0 && g < n) v = r[g];
        const bool ok = !isnan(v);
        s_sq[e] = ok ? v * v : 0.0;
        s_cnt[e] = ok ? 1 : 0;
    }
    __syncthreads();
    const int e0 = tid * C;
    double sq[C], pre[C];
    int cn[C];
#pragma unroll
    for (int k = 0; k < C; ++k) { sq[k] = s_sq[e0 + k]; cn[k] = s_cnt[e0 + k]; }
    int64_t m0 = (g0 + e0) % W;                        // position of my first element inside its segment
    if (m0 < 0) m0 += W;
    // ---- forward: prefix inside the segment
    double run = 0.0;
    int f = 0, csum = 0;
    {
        int64_t m = m0;
#pragma unroll
        for (int k = 0; k < C; ++k) {
            if (m == 0) { run = 0.0; f = 1; }
            run += sq[k];
            pre[k] = run;
            csum += cn[k];
            cn[k] = csum;                              // inclusive count inside my chunk
            if (++m == W) m = 0;
        }
    }
    const double cf = rv_block_carry<false>(run, f, w_s, w_f);
    {
        int64_t m = m0;
        bool open = true;                              // no segment start seen yet -> the carry applies
#pragma unroll
        for (int k = 0; k < C; ++k) {
            if (m == 0) open = false;
            if (open) pre[k] = cf + pre[k];
            if (++m == W) m = 0;
        }
    }
    // ---- plain inclusive count scan across threads
    int cinc = csum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(cinc, d, 64);
        if (lane >= d) cinc += o;
    }
    if (lane == 63) w_c[wv] = cinc;
    __syncthreads();
    int cbase = cinc - csum;
    for (int k = 0; k < wv; ++k) cbase += w_c[k];
    // ---- backward: suffix inside the segment (segment END at m == W-1)
    double suf[C];
    run = 0.0;
    f = 0;
    {
        int64_t m = m0 + C - 1;
        m %= W;
#pragma unroll
        for (int k = C - 1; k >= 0; --k) {
            if (m == W - 1) { run = 0.0; f = 1; }
            run += sq[k];
            suf[k] = run;
            if (--m < 0) m = W - 1;
        }
    }
    const double cb = rv_block_carry<true>(run, f, w_s, w_f);
    {
        int64_t m = (m0 + C - 1) % W;
        bool open = true;
#pragma unroll
        for (int k = C - 1; k >= 0; --k) {
            if (m == W - 1) open = false;
            if (open) suf[k] = cb + suf[k];
            if (--m < 0) m = W - 1;
        }
    }
    // publish suffix sums + inclusive counts (every thread has its chunk in registers by now)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < C; ++k) { s_sq[e0 + k] = suf[k]; s_cnt[e0 + k] = (unsigned short)(cbase + cn[k]); }
    __syncthreads();
    double res[C];
    {
        int64_t m = m0;
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const int e = e0 + k;
            const int es = e - (int)(W - 1);            // region element of the window start
            double o = NAN;
            if (es >= 0 && g0 + es >= 0) {              // i >= W-1
                const double sum = (m == W - 1) ? pre[k] : s_sq[es] + pre[k];
                const int valid = (cbase + cn[k]) - (es > 0 ? (int)s_cnt[es - 1] : 0);
                if (valid > 1) o = sqrt(sum / (double)(is_sample ? valid - 1 : valid));
            }
            res[k] = o;
            if (++m == W) m = 0;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < C; ++k) s_sq[e0 + k] = res[k];
    __syncthreads();
    for (int e = (int)(W - 1) + tid; e < R; e += RV_THREADS) {
        const int64_t i = g0 + e;
        if (i < n) out[i] = s_sq[e];
    }
}

// ---- W > RV_MAX_W: per-segment scans through global scratch ------------------------------------------
#define RVG_ITEMS 4
#define RVG_CHUNK (RV_THREADS * RVG_ITEMS)

// block-wide inclusive scan of (double, int) over RVG_CHUNK elements held RVG_ITEMS per thread (blocked layout);
// returns totals through *ts / *tc
__device__ __forceinline__ void rvg_scan(double (&v)[RVG_ITEMS], int (&c)[RVG_ITEMS], double *w_s, int *w_c,
                                         double *ts, int *tc)
{
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 1; k < RVG_ITEMS; ++k) { v[k] += v[k - 1]; c[k] += c[k - 1]; }
    double s = v[RVG_ITEMS - 1];
    int q = c[RVG_ITEMS - 1];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double os = __shfl_up(s, d, 64);
        const int oq = __shfl_up(q, d, 64);
        if (lane >= d) { s += os; q += oq; }
    }
    if (lane == 63) { w_s[w] = s; w_c[w] = q; }
    __syncthreads();
    double bs = __shfl_up(s, 1, 64);                  // exclusive prefix by shift, never by subtraction
    if (lane == 0) bs = 0.0;
    int bq = q - c[RVG_ITEMS - 1];
    double tot = 0.0;
    int totc = 0;
    for (int k = 0; k < 4; ++k) {
        if (k < w) { bs += w_s[k]; bq += w_c[k]; }
        tot += w_s[k];
        totc += w_c[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RVG_ITEMS; ++k) { v[k] += bs; c[k] += bq; }
    *ts = tot;
    *tc = totc;
}

// one block per segment [b*W, min(n, (b+1)*W)): prefix (REV = false) or suffix (REV = true) sums of r^2 and of the
// valid flags inside the segment
template <bool REV>
__global__ __launch_bounds__(RV_THREADS) void k_rv_segment_scan(const double *__restrict__ r, int64_t n, int64_t W,
                                                                double *__restrict__ sums, int *__restrict__ cnts)
{
    __shared__ double w_s[4];
    __shared__ int w_c[4];
    const int64_t lo = (int64_t)blockIdx.x * W, hi = (lo + W < n) ? lo + W : n;
    const int64_t len = hi - lo;
    double carry = 0.0;
    int ccarry = 0;
    for (int64_t c0 = 0; c0 < len; c0 += RVG_CHUNK) {
        double v[RVG_ITEMS];
        int c[RVG_ITEMS];
        int64_t idx[RVG_ITEMS];
#pragma unroll
        for (int k = 0; k < RVG_ITEMS; ++k) {
            const int64_t p = c0 + (int64_t)threadIdx.x * RVG_ITEMS + k;       // position along the scan direction
            idx[k] = p < len ? (REV ? hi - 1 - p : lo + p) : -1;
            const double x = idx[k] >= 0 ? r[idx[k]] : NAN;
            const bool ok = !isnan(x);
            v[k] = ok ? x * x : 0.0;
            c[k] = ok ? 1 : 0;
        }
        double ts;
        int tc;
        rvg_scan(v, c, w_s, w_c, &ts, &tc);
#pragma unroll
        for (int k = 0; k < RVG_ITEMS; ++k)
            if (idx[k] >= 0) { sums[idx[k]] = carry + v[k]; cnts[idx[k]] = ccarry + c[k]; }
        carry += ts;
        ccarry += tc;
    }
}

__global__ __launch_bounds__(256) void k_rv_combine(const double *__restrict__ pre, const int *__restrict__ cpre,
                                                    const double *__restrict__ suf, const int *__restrict__ csuf,
                                                    int64_t n, int64_t W, int is_sample, double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double o = NAN;
    const int64_t s = i - W + 1;
    if (s >= 0) {
        const bool whole = (s % W) == 0;
        const double sum = whole ? pre[i] : suf[s] + pre[i];
        const int valid = whole ? cpre[i] : csuf[s] + cpre[i];
        if (valid > 1) o = sqrt(sum / (double)(is_sample ? valid - 1 : valid));
    }
    out[i] = o;
}

extern "C" int fmk_realized_vol_dev(fmk_ctx *ctx, const double *d_r, int64_t n, int64_t window, int is_sample,
                                    double *d_out)
{
    if (window < 1) return fmk_set_error(ctx, FMK_E_ARG, "window must be at least 1");
    if (n <= 0) return FMK_OK;
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    if (window > n) {                                   // no full window anywhere
        k_fill_nan<<<(unsigned)fmk_ceil_div(n, 256), 256, 0, ctx->stream>>>(d_out, n);
        FMK_LAUNCH_CHECK(ctx);
        return FMK_OK;
    }
    if (window <= RV_SMALL_W) {
        const int64_t T = RV_THREADS * 9 - (window - 1);
        k_realized_vol<9><<<(unsigned)fmk_ceil_div(n, T), RV_THREADS, 0, ctx->stream>>>(d_r, n, window, is_sample, T, d_out);
        FMK_LAUNCH_CHECK(ctx);
        return FMK_OK;
    }
    if (window <= RV_MAX_W) {
        const int64_t T = RV_R - (window - 1);
        k_realized_vol<RV_C><<<(unsigned)fmk_ceil_div(n, T), RV_THREADS, 0, ctx->stream>>>(d_r, n, window, is_sample, T, d_out);
        FMK_LAUNCH_CHECK(ctx);
        return FMK_OK;
    }
    void *scr;
    FMK_TRY(fmk_scratch(ctx, (size_t)n * 24 + 64, &scr));
    double *pre = (double *)scr, *suf = pre + n;
    int *cpre = (int *)(suf + n), *csuf = cpre + n;
    const int64_t segs = fmk_ceil_div(n, window);
    k_rv_segment_scan<false><<<(unsigned)segs, RV_THREADS, 0, ctx->stream>>>(d_r, n, window, pre, cpre);
    FMK_LAUNCH_CHECK(ctx);
    k_rv_segment_scan<true><<<(unsigned)segs, RV_THREADS, 0, ctx->stream>>>(d_r, n, window, suf, csuf);
    FMK_LAUNCH_CHECK(ctx);
    k_rv_combine<<<(unsigned)fmk_ceil_div(n, 256), 256, 0, ctx->stream>>>(pre, cpre, suf, csuf, n, window, is_sample, d_out);
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}
