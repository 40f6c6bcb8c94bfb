// fmk_barflow.hip -- order-flow features (comp_bar_directional_features, base.py:409-546) on LDS tiles, and the
// cfg-4 entry points (OHLCV -> order-flow + footprints), on gfx950.
//
// Tiles.  A bar is streamed in tiles of up to 512 ticks: coalesced loads (price 512 B, amount 256 B, side 64 B per
// instruction) -> LDS, stored TRANSPOSED so that lane k owns r = 1, 2, 4 or 8 CONSECUTIVE ticks (rows padded
// r -> r+1: conflict-free ds_read_b64).
//
// Directional walk (bf_dir_tile).  Every lane walks its r ticks sequentially: running signed tick / volume /
// dollar sums, their local min / max, buy / sell sums and the spread terms are plain register updates (previous tick
// = previous loop iteration; the first tick of a lane reads the previous lane's last tick from LDS).  ONE wave scan
// of the lane totals per tile (DPP, fmk_dpp.h) turns the local extrema into bar-level ones:
//     min over the lane's ticks of (carry + exclusive lane prefix + local running sum).
// The first version scanned every 64-tick chunk (3 scans per chunk); this schedule scans once per 512 ticks.
//
// k_bar_dir: one wave per bar (fmk_comp_bar_directional_dev).  The other waves of the SIMD cover the load latency
// of a tile; a register-prefetch pipeline across tiles and bars was measured and is not faster (3.13 vs 3.05 ms at
// 1e9 ticks): the kernel is VALU-bound (~106 VALU instructions per 64 ticks), not latency-bound.
// The cfg-4 entry points (fmk_bars_flow_size_dev / _defer_dev) live here as well.
// float64 sums are combined in (lane-sequential, then tree) order; bars whose sums land within that reordering's
// noise of a float32 rounding tie are redone in tick order (k_bar_dir_redo), so the float32 outputs are the
// reference's bit for bit.
#include "fmk_footprint.h"
#include "fmk_f32tie.h"
#include "fmk_median.h"
#include "fmk_scan.h"

// developer knob FMK_DIR_FORCE_REDO=1 (tests): every bar of the wave-per-bar / workgroup-per-bar kernels goes on the redo list; =2: ... and the
// list is served by the wave-per-bar walkers (k_bar_dir_redo, which also serve the one-pass path's short bars) instead of the chunk-record kernel
__device__ int bf_force_redo = 0;
// diagnostics of the tick-order redo (fmk_diag_dir_redo): (bar, column) pairs redone, 512-term tiles, tiles added term by term
__device__ unsigned long long bf_redo_stats[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // [3 + r]: pairs of row r
// developer knob -DBF_REDO_TIMING: 100 MHz wall-clock ticks spent in the phases of k_bar_dir_redo_par (wave 0), read through
// fmk_diag_dir_redo's slots 3 .. 7 instead of the per-column pair counts
#ifdef BF_REDO_TIMING
#define BF_T(slot) do { if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&bf_redo_stats[slot], now_ - t_last_); t_last_ = now_; } } while (0)
#else
#define BF_T(slot) do { } while (0)
#endif
static int bf_sync_force_redo(fmk_ctx *ctx)
{
    static int current = 0;
    const char *v = getenv("FMK_DIR_FORCE_REDO");
    const int want = v ? (atoi(v) != 0) : 0;
    if (want != current) {
        FMK_HIP(ctx, hipMemcpyToSymbolAsync(HIP_SYMBOL(bf_force_redo), &want, sizeof(int), 0, hipMemcpyHostToDevice, ctx->stream));
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        current = want;
    }
    return FMK_OK;
}

struct FlowDirOut {
    int64_t *ticks_buy, *ticks_sell;
    float *volume_buy, *volume_sell, *dollars_buy, *dollars_sell;
    float *mean_spread, *max_spread;
    int64_t *cum_ticks_min, *cum_ticks_max;
    float *cum_volumes_min, *cum_volumes_max, *cum_dollars_min, *cum_dollars_max;
};
static_assert(sizeof(FlowDirOut) == sizeof(fmk_directional_out), "ABI struct mismatch");

#define BF_SLOTS 576                    // 64 rows x (8 + 1 pad)
#define BF_INIT_MIN 1000000000          // base.py:459-464
#define BF_INIT_MAX (-1000000000)

struct FlowDir {                        // per-lane accumulators of one bar
    double vb, vs, db, ds, cs, mxs;
    double da, va;                   // sums of |price x amount| and (float64 amounts) |amount| over the signed ticks: see bf_dir_write
    double vmin, vmax, dmin, dmax;
    int tmin, tmax, nbuy, nsell;
    // ticks from the bar's start to the END of the tile in which each float extremum was last improved (an upper bound of its
    // position: the rounding-noise bound of a running sum's extremum grows with the number of additions behind it)
    int64_t kvmin, kvmax, kdmin, kdmax;
    int64_t tiles_end;               // wave-uniform: ticks from the bar's start to the end of the current tile
    // wave-uniform
    int carry_t;
    double carry_v, carry_d;
    double prev_price;
    int prev_side;
};

__device__ __forceinline__ void bf_dir_init(FlowDir &d)
{
    d.vb = d.vs = d.db = d.ds = d.cs = d.mxs = 0.0;
    d.da = d.va = 0.0;
    d.vmin = d.dmin = 1e9; d.vmax = d.dmax = -1e9;       // base.py:461-464
    d.tmin = BF_INIT_MIN; d.tmax = BF_INIT_MAX;
    d.nbuy = d.nsell = 0;
    d.kvmin = d.kvmax = d.kdmin = d.kdmax = 0;
    d.tiles_end = 0;
    d.carry_t = 0; d.carry_v = d.carry_d = 0.0;
    d.prev_price = 0.0; d.prev_side = 0;
}

// slot of tick t of a tile whose lanes own r = 1 << lr consecutive ticks
__device__ __forceinline__ int bf_slot(int t, int lr) { return (t >> lr) * ((1 << lr) + 1) + (t & ((1 << lr) - 1)); }

// tile shape for `rem` remaining ticks: lanes own 1 << lr ticks, the tile holds tn of them
__device__ __forceinline__ void bf_shape(int64_t rem, int &lr, int &tn)
{
    lr = rem > 256 ? 3 : (rem > 128 ? 2 : (rem > 64 ? 1 : 0));
    tn = (int)(rem < ((int64_t)64 << lr) ? rem : ((int64_t)64 << lr));
}

// v_min_f64 / v_max_f64 without the canonicalising v_max_f64 x, x, x the compiler puts in front of fmin / fmax on loop
// carried values (5 extra VALU instructions per tick in a VALU-bound loop).  Operands here are sums / earlier
// min-max results, never signalling NaNs; a quiet NaN operand is ignored exactly like fmin / fmax do.
__device__ __forceinline__ double bf_min(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double bf_max(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// directional walk over one LDS tile (tn ticks, lanes own r = 1 << lr consecutive ones)
template <typename AmtT>
__device__ __forceinline__ void bf_dir_tile(int lane, int lr, int tn, const double *sP, const AmtT *sA, const int8_t *sS,
                                            FlowDir &d)
{
    const int r = 1 << lr;
    const int row = lane * (r + 1);
    double pp = d.prev_price;
    int ps = d.prev_side;
    if (lane > 0) { pp = sP[row - 2]; ps = sS[row - 2]; }          // previous lane's last tick (row - 1 is its pad)
    int rt = 0, ltmin = 0x7FFFFFFF, ltmax = (int)0x80000000;
    double rv = 0.0, rd = 0.0, lvmin = INFINITY, lvmax = -INFINITY, ldmin = INFINITY, ldmax = -INFINITY;
    for (int i = 0; i < r; ++i) {
        const bool valid = lane * r + i < tn;
        const double p = sP[row + i];
        const double av = (double)sA[row + i];
        const int sd = sS[row + i];
        if (valid && sd != ps) {                             // base.py:495-500
            const double sp = fabs(p - pp);
            d.mxs = bf_max(d.mxs, sp);
            d.cs += sp;
        }
        pp = p; ps = sd;
        const bool buy = valid && sd == 1, sell = valid && sd == -1;
        const double pv = p * av;
        if (buy) { d.vb += av; d.db += pv; d.nbuy += 1; }
        if (sell) { d.vs += av; d.ds += pv; d.nsell += 1; }
        if (buy || sell) {                                   // base.py:518-527: signed ticks only
            d.da += fabs(pv);
            if constexpr (sizeof(AmtT) == 8) d.va += fabs(av);
            rt += buy ? 1 : -1;
            rv += buy ? av : -av;
            rd += buy ? pv : -pv;
            ltmin = rt < ltmin ? rt : ltmin; ltmax = rt > ltmax ? rt : ltmax;
            lvmin = bf_min(lvmin, rv); lvmax = bf_max(lvmax, rv);
            ldmin = bf_min(ldmin, rd); ldmax = bf_max(ldmax, rd);
        }
    }
    // lane totals -> exclusive prefixes (one scan set per tile); local extrema -> bar extrema
    const int it = fmk_dpp_iscan(rt, 0, FmkOpAdd());
    const double iv = fmk_dpp_iscan(rv, 0.0, FmkOpAdd());
    const double id = fmk_dpp_iscan(rd, 0.0, FmkOpAdd());
    const int et = fmk_dpp_shift_up1(it, 0);
    const double ev = fmk_dpp_shift_up1(iv, 0.0), ed = fmk_dpp_shift_up1(id, 0.0);
    if (ltmin != 0x7FFFFFFF) {
        const int bt = d.carry_t + et;
        const double bv = d.carry_v + ev, bd = d.carry_d + ed;
        d.tmin = bt + ltmin < d.tmin ? bt + ltmin : d.tmin;
        d.tmax = bt + ltmax > d.tmax ? bt + ltmax : d.tmax;
        const int64_t te = d.tiles_end + tn;
        if (bv + lvmin < d.vmin) d.kvmin = te;
        if (bv + lvmax > d.vmax) d.kvmax = te;
        if (bd + ldmin < d.dmin) d.kdmin = te;
        if (bd + ldmax > d.dmax) d.kdmax = te;
        d.vmin = fmin(d.vmin, bv + lvmin); d.vmax = fmax(d.vmax, bv + lvmax);
        d.dmin = fmin(d.dmin, bd + ldmin); d.dmax = fmax(d.dmax, bd + ldmax);
    }
    d.tiles_end += tn;
    d.carry_t += fmk_last_lane(it);
    d.carry_v += fmk_last_lane(iv);
    d.carry_d += fmk_last_lane(id);
    const int last = bf_slot(tn - 1, lr);
    d.prev_price = __longlong_as_double(fmk_uniform((int64_t)__double_as_longlong(sP[last])));
    d.prev_side = fmk_uniform((int)sS[last]);
}

// The float32 sums of ONE bar in the reference's tick order (base.py:466-546), for the few bars whose float64 sums
// land within rounding noise of a float32 tie, where the tree-ordered sums of the wave could round the other way.
// A sequential float64 sum cannot be re-associated, but its TERMS can be prepared in parallel and the seven sums are
// independent of each other: per 64-tick chunk lane t computes the seven terms of tick t (buy / sell volume and
// dollars, spread, signed volume and dollars; 0.0 where the reference skips the update: x + 0.0 == x) into LDS rows,
// then lane r walks row r in tick order -- one ds_read, one add and the running min / max per tick.
// Integer outputs and max_spread do not depend on the order; k_bar_dir has already written them.
#define BF_SEQ_ROW 65                   // 64 ticks + 1 pad: the seven row walkers hit different banks
template <typename AmtT>
__device__ __forceinline__ void bf_dir_sequential(const FlowDirOut &o, int64_t b, int lane, double *rows,
                                                  const double *__restrict__ price, const AmtT *__restrict__ amount,
                                                  const int8_t *__restrict__ side, int64_t start, int64_t e, int64_t n)
{
    double acc = 0.0, mn = 1e9, mx = -1e9;              // base.py:461-464
    int64_t nflow = 0;
    int prev = e > start ? (int)side[fmk_wrap(start - 1, n)] : 0;       // base.py:485-488
    double pprev = e >= start ? price[fmk_wrap(start - 1, n)] : 0.0;
    double p = 0.0, v = 0.0;
    int sd = 0;
    bool started = false;                               // a signed tick has been met (wave-uniform)
    if (start + lane <= e) { p = price[start + lane]; v = (double)amount[start + lane]; sd = side[start + lane]; }
    for (int64_t j0 = start; j0 <= e; j0 += 64) {
        const bool valid = j0 + lane <= e;
        const double pp = fmk_dpp_shift_up1(p, pprev);
        const int sp_side = fmk_dpp_shift_up1(sd, prev);
        const double pv = p * v, sp = fabs(p - pp);
        const bool buy = valid && sd == 1, sell = valid && sd == -1;
        rows[0 * BF_SEQ_ROW + lane] = buy ? v : 0.0;
        rows[1 * BF_SEQ_ROW + lane] = sell ? v : 0.0;
        rows[2 * BF_SEQ_ROW + lane] = buy ? pv : 0.0;
        rows[3 * BF_SEQ_ROW + lane] = sell ? pv : 0.0;
        rows[4 * BF_SEQ_ROW + lane] = valid && sd != sp_side ? sp : 0.0;
        rows[5 * BF_SEQ_ROW + lane] = buy ? v : sell ? -v : 0.0;
        rows[6 * BF_SEQ_ROW + lane] = buy ? pv : sell ? -pv : 0.0;
        const uint64_t flow = __ballot(buy || sell);    // ticks that move the running sums (base.py:506-530)
        nflow += __popcll(flow);
        pprev = fmk_last_lane(p);
        prev = fmk_last_lane(sd);
        // next chunk's loads fly while the rows are walked
        const int64_t jn = j0 + 64 + lane;
        p = 0.0; v = 0.0; sd = 0;
        if (jn <= e) { p = price[jn]; v = (double)amount[jn]; sd = side[jn]; }
        __builtin_amdgcn_wave_barrier();
        if (lane < 7) {
            const double *row = rows + lane * BF_SEQ_ROW;
            if (started) {
                // A signed tick lies behind: the running sums of rows 5 / 6 (the only rows whose extrema are used) do not
                // move on the other ticks -- their terms are 0.0 -- so every tick's value may enter the extrema: three
                // instructions per tick instead of five plus the mask arithmetic (this loop is what a redone hourly or daily
                // bar costs: profiles/r02_long_bars.txt)
#pragma unroll 16
                for (int k = 0; k < 64; ++k) {
                    acc += row[k];
                    mn = fmin(mn, acc);
                    mx = fmax(mx, acc);
                }
            } else {
#pragma unroll 16
                for (int k = 0; k < 64; ++k) {
                    acc += row[k];
                    const double cand = (flow >> k) & 1 ? acc : NAN;     // fmin / fmax ignore NaN: no update on side 0
                    mn = fmin(mn, cand);
                    mx = fmax(mx, cand);
                }
            }
        }
        started = started || flow != 0;
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) o.volume_buy[b] = (float)acc;
    if (lane == 1) o.volume_sell[b] = (float)acc;
    if (lane == 2) o.dollars_buy[b] = (float)acc;
    if (lane == 3) o.dollars_sell[b] = (float)acc;
    if (lane == 4) o.mean_spread[b] = nflow == 0 ? NAN : (float)(acc / (double)nflow);
    if (lane == 5) { o.cum_volumes_min[b] = (float)mn; o.cum_volumes_max[b] = (float)mx; }
    if (lane == 6) { o.cum_dollars_min[b] = (float)mn; o.cum_dollars_max[b] = (float)mx; }
}

// the bar's (or a bar segment's) accumulators folded over the wave: wave-uniform
struct FlowTotals {
    double vb, vs, db, ds, cs, mxs;
    double da, va;
    double vmin, vmax, dmin, dmax;
    int tb, tsell, tmin, tmax;
    int64_t kvmin, kvmax, kdmin, kdmax;      // upper bounds of the extrema's positions (ticks from the bar's start)
};

// position bound of a reduced extremum: the largest bound among the lanes that hold the reduced value
__device__ __forceinline__ int64_t bf_pos_of(double lane_val, double reduced, int64_t lane_pos)
{
    return fmk_dpp_reduce(lane_val == reduced ? lane_pos : (int64_t)0, (int64_t)0, FmkOpMax());
}

__device__ __forceinline__ FlowTotals bf_dir_fold(const FlowDir &d)
{
    FlowTotals t;
    t.vb = fmk_dpp_reduce(d.vb, 0.0, FmkOpAdd()); t.vs = fmk_dpp_reduce(d.vs, 0.0, FmkOpAdd());
    t.db = fmk_dpp_reduce(d.db, 0.0, FmkOpAdd()); t.ds = fmk_dpp_reduce(d.ds, 0.0, FmkOpAdd());
    t.cs = fmk_dpp_reduce(d.cs, 0.0, FmkOpAdd()); t.mxs = fmk_dpp_reduce(d.mxs, 0.0, FmkOpMax());
    t.da = fmk_dpp_reduce(d.da, 0.0, FmkOpAdd()); t.va = fmk_dpp_reduce(d.va, 0.0, FmkOpAdd());
    t.tb = fmk_dpp_reduce(d.nbuy, 0, FmkOpAdd()); t.tsell = fmk_dpp_reduce(d.nsell, 0, FmkOpAdd());
    t.tmin = fmk_dpp_reduce(d.tmin, BF_INIT_MIN, FmkOpMin());
    t.tmax = fmk_dpp_reduce(d.tmax, BF_INIT_MAX, FmkOpMax());
    t.vmin = fmk_dpp_reduce(d.vmin, 1e9, FmkOpMin()); t.vmax = fmk_dpp_reduce(d.vmax, -1e9, FmkOpMax());
    t.dmin = fmk_dpp_reduce(d.dmin, 1e9, FmkOpMin()); t.dmax = fmk_dpp_reduce(d.dmax, -1e9, FmkOpMax());
    t.kvmin = bf_pos_of(d.vmin, t.vmin, d.kvmin); t.kvmax = bf_pos_of(d.vmax, t.vmax, d.kvmax);
    t.kdmin = bf_pos_of(d.dmin, t.dmin, d.kdmin); t.kdmax = bf_pos_of(d.dmax, t.dmax, d.kdmax);
    return t;
}

// write the 14 per-bar outputs (lane 0) and put the bar on the redo list when a float32 output could round the other way
template <typename AmtT>
__device__ __forceinline__ void bf_dir_write(const FlowDirOut &o, int64_t b, int lane, const FlowTotals &t,
                                             unsigned long long *n_zero_div, int64_t start, int64_t e,
                                             unsigned long long *redo)
{
    const double vb = t.vb, vs = t.vs, db = t.db, ds = t.ds, cs = t.cs, mxs = t.mxs;
    const int tb = t.tb, tsell = t.tsell, tmin = t.tmin, tmax = t.tmax;
    const double vmin = t.vmin, vmax = t.vmax, dmin = t.dmin, dmax = t.dmax;
    if (lane == 0 && tb + tsell == 0 && n_zero_div) atomicAdd(n_zero_div, 1ULL);   // reference: ZeroDivisionError (base.py:536)
    // The sums above are float64 in (lane, tree[, wave]) order, the reference's in tick order.  float32 outputs can only differ when
    // a sum sits within the two orders' rounding noise of a float32 rounding boundary; those bars go on the redo list (redo[0]:
    // count, redo[32...]: bar number | column mask << 48) and k_bar_dir_redo_par adds the flagged columns in tick order.
    //   * one-signed sums (buy / sell volume and dollars, spread): recursive summation of len terms errs by at most (len - 1) u S
    //     (u = 2^-53, S the sum), the lane-sequential / tree / tile-carry order here by at most (len / 64 + len / 512 + 22) u S:
    //     together below 1.05 (len + 64) u S.
    //   * extrema of the running SIGNED sums: every partial sum -- in either order -- has magnitude at most M = max(|min|, |max|)
    //     (+ the deviation itself), and running error analysis bounds the recursive sum's error at tick k by u (|s_2| + ... + |s_k|)
    //     <= len u M; a value of the parallel order is reached through at most d = len / 512 + 16 additions of partial sums over
    //     contiguous ranges, each below 2 M: d u 2 M (d taken as len / 512 + 32).  Round 2 bounded these by len u (buy + sell): for a daily bar that is 7 units
    //     against a float32 spacing of 1 at M ~ 1e7 -- EVERY long bar was redone, ~20 ms of one wave each.
    //   * float32 amounts: the volume sums and the running signed volume are float64 sums of 24-bit terms -- exact in any order
    //     (amounts of a bar spanning less than 2^29 in magnitude, the assumption of every kernel here): they cannot tie.
    // All operands are wave-uniform, so the decision is too.
    const double len = (double)(e - start + 65);
    const double eps = 1.17e-16 * len;
    const double mean = tb + tsell == 0 ? 0.0 : cs / (double)(tb + tsell);
    const double mv = fmax(fabs(vmin), fabs(vmax)), md = fmax(fabs(dmin), fabs(dmax));
    // ... an extremum reached after k ticks has only k additions behind it in the reference's order: len -> k (+ a tile, the
    // granularity the position is known at).  Most extrema of small magnitude -- fine float32 spacing -- are early ones.
    // Absolute error bound of an extremum reached after k ticks, M = the largest magnitude a running sum takes, A = the sum of the
    // terms' magnitudes (buy + sell).  The reference's recursive sum: one rounding per tick, each at most u M: k u M.  The parallel
    // order: every NODE of its dependency tree rounds once, by at most u |node|; the nodes of one level are sums over disjoint tick
    // ranges, so their magnitudes add up to at most A -- (levels) u A with <= 24 levels (8 ticks in a lane, 6 scan steps, the
    // composition of up to 16 waves, slack) -- and the carry from tile to tile is a chain of k / 512 nodes below 2 M each.
    // (History: round 3 charged the parallel order 2 M per LEVEL, which counts one node per level -- not a bound; the first repair in
    // round 4 charged 2 M per NODE, 3 (k + 64) u M in all -- a bound, but three times the reference's own term: daily bars were
    // redone three times as often, order flow on them 6.6 -> 17.9 ms per 1e9 ticks.  This one is a bound and costs what round 3 did.)
    auto eps_at = [](int64_t k, double M, double A) {
        const double kk = (double)(k + 64);
        return 1.13e-16 * ((kk + 4.0 * (kk / 512.0 + 2.0)) * M + 24.0 * A);
    };
    unsigned mask = 0;
    // (magnitudes: a tape of NEGATIVE prices -- tools/fuzz_fused.py drew one, seed 7707 case 9 -- has negative dollar sums; with the signed
    //  sum as the bound the test said "not near" for every bar and five of 1 649 kept the parallel order's last bit.)
    // Both bounds take |sum| for the sum of the terms' magnitudes, which holds while the terms of a bar have ONE sign.  t.da is that sum
    // of magnitudes itself (one more addition per signed tick): a bar in which it exceeds |db| + |ds| has price x amount terms of both
    // signs -- its sums cancel, the bounds would come out too small -- and goes to the tick-order redo with every dollar column,
    // whatever its values.  (Below the 1e-9 the test allows for da's own rounding, A is understated by less than that factor: the 1.8 %
    // between 1.13e-16 and 2^-53 in eps_at, and the 5 % in eps, cover it.  A NaN sum compares false and takes the tests below, as before.)
    const double adb = fabs(db), ads = fabs(ds);
    if (t.da > (adb + ads) * (1.0 + 1e-9)) mask |= (1u << 2) | (1u << 3) | (1u << 6);
    else {
        if (fmk_near_f32_tie(db, eps * adb)) mask |= 1u << 2;
        if (fmk_near_f32_tie(ds, eps * ads)) mask |= 1u << 3;
        if (tb + tsell > 0 && (fmk_near_f32_tie(dmin, eps_at(t.kdmin, md, adb + ads)) || fmk_near_f32_tie(dmax, eps_at(t.kdmax, md, adb + ads)))) mask |= 1u << 6;
    }
    if (fmk_near_f32_tie(mean, (eps + 1.2e-16) * fabs(mean))) mask |= 1u << 4;
    if constexpr (sizeof(AmtT) == 8) {
        if (t.va > (fabs(vb) + fabs(vs)) * (1.0 + 1e-9)) mask |= (1u << 0) | (1u << 1) | (1u << 5);       // amounts of both signs
        else {
            if (fmk_near_f32_tie(vb, eps * fabs(vb))) mask |= 1u << 0;
            if (fmk_near_f32_tie(vs, eps * fabs(vs))) mask |= 1u << 1;
            if (tb + tsell > 0 && (fmk_near_f32_tie(vmin, eps_at(t.kvmin, mv, fabs(vb) + fabs(vs))) || fmk_near_f32_tie(vmax, eps_at(t.kvmax, mv, fabs(vb) + fabs(vs))))) mask |= 1u << 5;
        }
    }
    if (bf_force_redo != 0) mask = 0x7F;                               // (tests: every bar through the tick-order redo)
    if (mask && lane == 0) redo[32 + atomicAdd(redo, 1ULL)] = (unsigned long long)b | ((unsigned long long)mask << 48);
    if (lane == 0) {
        o.ticks_buy[b] = tb; o.ticks_sell[b] = tsell;
        o.volume_buy[b] = (float)vb; o.volume_sell[b] = (float)vs;
        o.dollars_buy[b] = (float)db; o.dollars_sell[b] = (float)ds;
        o.max_spread[b] = (float)mxs;
        o.mean_spread[b] = tb + tsell == 0 ? NAN : (float)mean;
        o.cum_ticks_min[b] = tmin; o.cum_ticks_max[b] = tmax;
        o.cum_volumes_min[b] = (float)vmin; o.cum_volumes_max[b] = (float)vmax;
        o.cum_dollars_min[b] = (float)dmin; o.cum_dollars_max[b] = (float)dmax;
    }
}

// fold the lanes and write the 14 per-bar outputs
template <typename AmtT>
__device__ __forceinline__ void bf_dir_emit(const FlowDirOut &o, int64_t b, int lane, const FlowDir &d,
                                            unsigned long long *n_zero_div, int64_t start, int64_t e,
                                            unsigned long long *redo)
{
    const FlowTotals t = bf_dir_fold(d);
    bf_dir_write<AmtT>(o, b, lane, t, n_zero_div, start, e, redo);
}

// ---------------------------------------------------------------------------------------
// directional only: one wave per bar
// ---------------------------------------------------------------------------------------
template <bool AF64, int WPB = 4>
__global__ __launch_bounds__(64 * WPB) void k_bar_dir(const double *__restrict__ price, const void *__restrict__ amount,
                                                 const int8_t *__restrict__ side, const int64_t *__restrict__ ci,
                                                 int64_t nb, int64_t n, FlowDirOut o, unsigned long long *n_zero_div,
                                                 unsigned long long *redo, const unsigned long long *only = nullptr,
                                                 int64_t skip_above = INT64_MAX /* longer bars: k_bar_dir_wide */)
{
    typedef typename std::conditional<AF64, double, float>::type AmtT;
    __shared__ double s_p[WPB][BF_SLOTS];
    __shared__ AmtT s_a[WPB][BF_SLOTS];
    __shared__ int8_t s_s[WPB][640];
    const int lane = fmk_lane();
    const int wib = fmk_uniform((int)(threadIdx.x >> 6));
    double *sP = s_p[wib];
    AmtT *sA = s_a[wib];
    int8_t *sS = s_s[wib];
    const AmtT *am = (const AmtT *)amount;
    const int64_t wave0 = (int64_t)blockIdx.x * WPB + wib;
    const int64_t nwaves = (int64_t)gridDim.x * WPB;
    // `only` (list mode: [0] = count, [32...] = bar numbers): the bars k_bar_dir_lanes left to this schedule
    const int64_t todo = only ? (int64_t)only[0] : nb;
    for (int64_t it = wave0; it < todo; it += nwaves) {
        const int64_t b = only ? fmk_uniform((int64_t)only[32 + it]) : it;
        const int64_t s = fmk_uniform(ci[b]);
        const int64_t e = fmk_uniform(ci[b + 1]);
        const int64_t start = s + 1;
        if (e - s > skip_above) continue;
        FlowDir d;
        bf_dir_init(d);
        if (e > s) {
            d.prev_price = price[fmk_wrap(start - 1, n)];
            d.prev_side = e - s > 1 ? (int)side[fmk_wrap(start - 1, n)] : 0;    // base.py:485-488
        }
        int64_t j0 = start, rem = e - s;
        while (rem > 0) {
            int lr, tn;
            bf_shape(rem, lr, tn);
            // uniform base pointer + 32-bit lane offset: one address VGPR for all chunks.  No register prefetch of
            // the next tile: the other waves of the SIMD cover the load latency (keeps the kernel at 4 waves/SIMD).
            const double *pb = price + j0;
            const AmtT *ab = am + j0;
            const int8_t *sb = side + j0;
            double pr[8];
            AmtT ar[8];
            int sr[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                pr[c] = 0.0; ar[c] = 0; sr[c] = 0;
                const int t = c * 64 + lane;
                if (c < (1 << lr) && t < tn) { pr[c] = pb[t]; ar[c] = ab[t]; sr[c] = sb[t]; }
            }
            // slot(c * 64 + lane) is linear in c: one multiply-add per chunk instead of shift / mask / multiply
            const int slot0 = bf_slot(lane, lr);
            const int cstride = (64 >> lr) * ((1 << lr) + 1);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (c < (1 << lr)) {
                    const int sl = slot0 + c * cstride;
                    sP[sl] = pr[c]; sA[sl] = ar[c]; sS[sl] = (int8_t)sr[c];
                }
            }
            __builtin_amdgcn_wave_barrier();
            bf_dir_tile<AmtT>(lane, lr, tn, sP, sA, sS, d);
            __builtin_amdgcn_wave_barrier();
            j0 += tn;
            rem -= tn;
        }
        bf_dir_emit<AmtT>(o, b, lane, d, n_zero_div, start, e, redo);
    }
}


// ---------------------------------------------------------------------------------------
// Bars of more than BFW_MIN ticks (hourly, daily bars): a WORKGROUP per bar (round 3).  With one wave per bar 580 daily bars are 580
// waves on 1024 SIMDs, each alone with its load latency: 10 ms per 1e9 ticks before any redo.  The order-flow quantities compose
// over contiguous segments -- sums add, and the extrema of the running signed sums are  min over segments of (sum of the earlier
// segments + the segment's own extremum)  -- so eight waves take an eighth of the bar each (whole 512-tick tiles; the tick before
// a segment comes from memory) and wave 0 combines the eight results in order.  Same tie test and redo list as k_bar_dir.
// ---------------------------------------------------------------------------------------
#define BFW_MIN 16384
#define BFW_WAVES 8
struct FlowSeg {                     // one wave's segment, wave-uniform values
    FlowTotals t;
    int net_t;
    double net_v, net_d;
};

template <bool AF64>
__global__ __launch_bounds__(64 * BFW_WAVES, 4) void k_bar_dir_wide(const double *__restrict__ price, const void *__restrict__ amount,
                                                               const int8_t *__restrict__ side, const int64_t *__restrict__ ci,
                                                               const int64_t *__restrict__ list, int64_t n, FlowDirOut o,
                                                               unsigned long long *n_zero_div, unsigned long long *redo)
{
    typedef typename std::conditional<AF64, double, float>::type AmtT;
    __shared__ double s_p[BFW_WAVES][BF_SLOTS];
    __shared__ AmtT s_a[BFW_WAVES][BF_SLOTS];
    __shared__ int8_t s_s[BFW_WAVES][640];
    __shared__ FlowSeg s_seg[BFW_WAVES];
    const int lane = fmk_lane();
    const int w = fmk_uniform((int)(threadIdx.x >> 6));
    double *sP = s_p[w];
    AmtT *sA = s_a[w];
    int8_t *sS = s_s[w];
    const AmtT *am = (const AmtT *)amount;
    const int64_t n_list = list[0];
    for (int64_t q = blockIdx.x; q < n_list; q += gridDim.x) {
        const int64_t b = list[1 + q], s = ci[b], e = ci[b + 1], start = s + 1, cnt = e - s;
        int64_t seg = (cnt + BFW_WAVES - 1) / BFW_WAVES;
        seg = (seg + 511) & ~(int64_t)511;                           // whole tiles
        const int64_t j_lo = start + (int64_t)w * seg;
        const int64_t j_hi = j_lo + seg - 1 < e ? j_lo + seg - 1 : e;   // inclusive
        FlowDir d;
        bf_dir_init(d);
        d.tiles_end = (int64_t)w * seg;                              // positions count from the bar's start
        if (j_lo <= e) {
            d.prev_price = price[fmk_wrap(j_lo - 1, n)];
            // base.py:485-488: the bar's first tick compares with the tick before the bar (wrapped), except in a one-tick bar
            d.prev_side = (w > 0 || cnt > 1) ? (int)side[fmk_wrap(j_lo - 1, n)] : 0;
        }
        int64_t j0 = j_lo, rem = j_hi - j_lo + 1;
        while (rem > 0) {
            int lr, tn;
            bf_shape(rem, lr, tn);
            const double *pb = price + j0;
            const AmtT *ab = am + j0;
            const int8_t *sb = side + j0;
            double pr[8];
            AmtT ar[8];
            int sr[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                pr[c] = 0.0; ar[c] = 0; sr[c] = 0;
                const int t = c * 64 + lane;
                if (c < (1 << lr) && t < tn) { pr[c] = pb[t]; ar[c] = ab[t]; sr[c] = sb[t]; }
            }
            const int slot0 = bf_slot(lane, lr);
            const int cstride = (64 >> lr) * ((1 << lr) + 1);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (c < (1 << lr)) {
                    const int sl = slot0 + c * cstride;
                    sP[sl] = pr[c]; sA[sl] = ar[c]; sS[sl] = (int8_t)sr[c];
                }
            }
            __builtin_amdgcn_wave_barrier();
            bf_dir_tile<AmtT>(lane, lr, tn, sP, sA, sS, d);
            __builtin_amdgcn_wave_barrier();
            j0 += tn;
            rem -= tn;
        }
        const FlowTotals t = bf_dir_fold(d);
        __syncthreads();                                             // (the previous bar's segments have been read)
        if (lane == 0) { s_seg[w].t = t; s_seg[w].net_t = d.carry_t; s_seg[w].net_v = d.carry_v; s_seg[w].net_d = d.carry_d; }
        __syncthreads();
        if (w == 0) {
            FlowTotals a = s_seg[0].t;
            int ct = s_seg[0].net_t;
            double cv = s_seg[0].net_v, cd = s_seg[0].net_d;
            for (int k = 1; k < BFW_WAVES; ++k) {
                const FlowTotals &x = s_seg[k].t;
                a.vb += x.vb; a.vs += x.vs; a.db += x.db; a.ds += x.ds; a.cs += x.cs;
                a.da += x.da; a.va += x.va;
                a.mxs = fmax(a.mxs, x.mxs);
                a.tb += x.tb; a.tsell += x.tsell;
                if (x.tmin != BF_INIT_MIN) {                         // the segment met a signed tick: its extrema count
                    a.tmin = ct + x.tmin < a.tmin ? ct + x.tmin : a.tmin;
                    a.tmax = ct + x.tmax > a.tmax ? ct + x.tmax : a.tmax;
                    if (cv + x.vmin < a.vmin) a.kvmin = x.kvmin;
                    if (cv + x.vmax > a.vmax) a.kvmax = x.kvmax;
                    if (cd + x.dmin < a.dmin) a.kdmin = x.kdmin;
                    if (cd + x.dmax > a.dmax) a.kdmax = x.kdmax;
                    a.vmin = fmin(a.vmin, cv + x.vmin); a.vmax = fmax(a.vmax, cv + x.vmax);
                    a.dmin = fmin(a.dmin, cd + x.dmin); a.dmax = fmax(a.dmax, cd + x.dmax);
                }
                ct += s_seg[k].net_t; cv += s_seg[k].net_v; cd += s_seg[k].net_d;
            }
            bf_dir_write<AmtT>(o, b, lane, a, n_zero_div, start, e, redo);
        }
    }
}

// the bars k_bar_dir put on the redo list, one wave each, in tick order
template <bool AF64>
__global__ __launch_bounds__(256) void k_bar_dir_redo(const double *__restrict__ price, const void *__restrict__ amount,
                                                      const int8_t *__restrict__ side, const int64_t *__restrict__ ci,
                                                      int64_t n, FlowDirOut o, const unsigned long long *__restrict__ redo)
{
    typedef typename std::conditional<AF64, double, float>::type AmtT;
    __shared__ double s_rows[4][7 * BF_SEQ_ROW];
    const int lane = fmk_lane();
    const int64_t count = (int64_t)redo[0];
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t i = (int64_t)blockIdx.x * 4 + fmk_uniform((int)(threadIdx.x >> 6)); i < count; i += nwaves) {
        const int64_t b = fmk_uniform((int64_t)(redo[32 + i] & 0xFFFFFFFFFFFFULL));
        const int64_t s = fmk_uniform(ci[b]), e = fmk_uniform(ci[b + 1]);
        bf_dir_sequential<AmtT>(o, b, lane, s_rows[threadIdx.x >> 6], price, (const AmtT *)amount, side, s + 1, e, n);
    }
}

// ---------------------------------------------------------------------------------------
// The reference's SEQUENTIAL float64 sums of a redone bar, in parallel (round 3).  A recursive sum cannot be re-associated -- but
// while the running sum s stays inside one binade [2^e, 2^(e+1)) every addition is  s <- s + rnd_g(x)  with rnd_g = rounding to the
// grid g = 2^(e-52) of that binade, and rnd_g(x) does not depend on s unless x lies exactly half way between two grid points (then
// round-to-even looks at the parity of s / g).  So for a chunk of 64 terms:
//     y_k = (C + x_k) - C,  C = 1.5 * 2^e        (the machine's own rounding to the grid; exact: Fast2Sum with |x_k| < C)
//     r_k = x_k - y_k                           (exact remainder; |r_k| == g / 2  <=>  a tie)
//     S_k = s + (y_1 + ... + y_k)               (multiples of g below 2^53 g: EXACT in any order -> a DPP scan)
// and if no term ties, every |x_k| < 2^(e-5) (so that all subset sums of the y stay below 2^53 g) and every S_k lies strictly inside
// the binade, then S_1 .. S_64 ARE the reference's partial sums, bit for bit (induction over k).  Signed sums work on |s| with the
// terms' signs flipped along.  The BINADE of the running sum is known in advance from an approximate (re-associated) prefix sum
// whenever the sum is not within rounding noise of a power of two.  So, per super-block of 512 chunks (32 768 ticks):
//   A. all waves, chunk by chunk: the approximate start value S~ of the chunk (a scan of the chunks' plain totals) gives its
//      binade e; the 64 terms are rounded to that binade's grid and the chunk is GOOD if no term ties, every |x| < 2^(e-5) and
//      S~ + (y_1 + .. + y_k) stays inside the binade by a margin that covers S~'s own error.  A good chunk is reduced to three
//      EXACT numbers: its total T = sum y and the extrema minP / maxP of its inclusive prefixes.  Sixteen consecutive good chunks of
//      one binade and sign fold into a GROUP record the same way (total, extrema of  total of the chunks before + the chunk's
//      extrema);
//   B. one wave walks the group records in order: a group whose actual start s has the predicted binade and sign and whose partial
//      sums stay strictly inside it is  s <- s + T  -- exact, and the reference's 1024 roundings in one step -- with extrema
//      s + minP / s + maxP; any other group is walked chunk by chunk with the same test, and a chunk that fails it term by term.
//      (Exactness of a group's numbers: they are sums of multiples of g; as long as the true running sum stays inside the binade
//      every partial result is below 2^53 g and the additions are exact; the first partial result that leaves the binade -- exact
//      or rounded, rounding is monotone -- fails the range test, and the test sees the minimum and maximum over ALL of them.)
// About 1 % of the chunks of a daily bar take the slow road.  A workgroup takes one (bar, column) pair at a time -- the seven
// columns (buy / sell volume, buy / sell dollars, spread, signed volume, signed dollars) are independent sums.
// ---------------------------------------------------------------------------------------
#define RS_CH 512
#define RS_WAVES 8                    // 512 threads: 256 VGPRs per lane (two raw tiles, the terms and their prefixes live at once)
#define RS_POOL 186                   // chunks whose terms wait in LDS for the term-by-term walk
#define RS_UNROLL 4
#define RS_GRP 16                     // chunks per group record
#define RS_SUB 4                      // chunks per sub-group record (what a group that does not fit the actual sum falls back to)
#define RS_NONE ((int)0x80000000)
// TIES.  A term that lies exactly half way between two grid points rounds to even, i.e. by the PARITY of the running sum's last
// bit -- and ties are not rare: 4 % of the prices on a 0.01 grid are multiples of 0.25, their product with a 24-bit size is exact,
// and its lowest set bit is half a grid step for ~1 % of the terms of a daily bar's dollar sums (every second chunk holds one:
// profiles/r03_long_bars.txt).  With S = |s| / g and a tie x = (m + 1/2) g:  S <- S + m + ((S + m) & 1), which is EVEN afterwards
// whatever S was.  So only a chunk's FIRST tie depends on the incoming parity, and a record carries two variants -- index 0 / 1 =
// the parity of S at the chunk's start -- which differ by one grid step from that tie on.  `pout`: the parity after the chunk
// (absolute when the chunk holds a tie, otherwise the parity of its total, to be xor-ed onto the incoming one).
struct RsRec {
    double T[2], minP[2], maxP[2];
    int e;                            // binade of |s| the record was computed for; RS_NONE: not usable
    short neg;                        // sign of s it was computed for
    unsigned char has_tie, pout;
};

// `count` consecutive records folded into one (chunks into a sub-group, sub-groups into a group): totals add, the extrema are
// (total of the records before) + the record's own, each for the two incoming parities; a record without a usable part, or parts
// of different binades / signs, gives RS_NONE.  The parity map of the fold is again "absolute" or "xor": the two incoming parities
// end up equal exactly when some part on the way made them so.
__device__ __forceinline__ RsRec rs_compose(const RsRec *r0, int count)
{
    RsRec g;
    g.e = r0[0].e; g.neg = r0[0].neg; g.has_tie = 0; g.pout = 0;
    int qv[2] = {0, 1};
#pragma unroll
    for (int v = 0; v < 2; ++v) { g.T[v] = 0.0; g.minP[v] = INFINITY; g.maxP[v] = -INFINITY; }
    for (int k = 0; k < count; ++k) {
        const RsRec r = r0[k];
        if (r.e == RS_NONE || r.e != g.e || r.neg != g.neg) { g.e = RS_NONE; break; }
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int u = qv[v];
            const double rmn = u ? r.minP[1] : r.minP[0], rmx = u ? r.maxP[1] : r.maxP[0], rt = u ? r.T[1] : r.T[0];
            g.minP[v] = fmin(g.minP[v], g.T[v] + rmn);
            g.maxP[v] = fmax(g.maxP[v], g.T[v] + rmx);
            g.T[v] += rt;
            qv[v] = r.has_tie ? (int)r.pout : (u ^ (int)r.pout);
        }
    }
    g.has_tie = qv[0] == qv[1];
    g.pout = (unsigned char)qv[0];
    return g;
}

// the row's terms of one chunk (64 ticks from j0, lane = tick; 0.0 where the reference skips the update) with unconditional loads
// (clamped into the bar): the compiler batches them.  `flow`: the tick is a signed one.
template <typename AmtT>
__device__ __forceinline__ double rs_chunk_term(int row, int64_t j0, int lane, int64_t start, int64_t e_bar, int64_t n,
                                                const double *__restrict__ price, const AmtT *__restrict__ am,
                                                const int8_t *__restrict__ side, bool &flow)
{
    const int64_t j = j0 + lane;
    const bool valid = j <= e_bar;
    const int64_t jc = valid ? j : e_bar;
    const int64_t jp = fmk_wrap(jc - 1, n);
    const double p = price[jc], pprev = price[jp];
    const double v = (double)am[jc];
    const int sd = side[jc];
    int sprev = side[jp];
    if (jc == start && !(e_bar > start)) sprev = 0;                      // base.py:485-488: a one-tick bar
    const bool buy = valid && sd == 1, sell = valid && sd == -1;
    flow = buy || sell;
    const double pv = p * v;
    switch (row) {                                                       // (wave-uniform)
    case 0: return buy ? v : 0.0;
    case 1: return sell ? v : 0.0;
    case 2: return buy ? pv : 0.0;
    case 3: return sell ? pv : 0.0;
    case 4: return valid && sd != sprev ? fabs(p - pprev) : 0.0;       // base.py:485-500: |price change| where the side changes
    case 5: return buy ? v : sell ? -v : 0.0;
    default: return buy ? pv : sell ? -pv : 0.0;
    }
}

// a tile of eight chunks as loaded (lane = tick of the chunk) plus the tick before the tile
template <typename AmtT>
struct RsRaw {
    double p[8];
    AmtT v[8];
    int sd[8];
    double pp;
    int ps;
};

template <bool AF64>
__global__ __launch_bounds__(64 * RS_WAVES) void k_bar_dir_redo_par(const double *__restrict__ price, const void *__restrict__ amount,
                                                                  const int8_t *__restrict__ side, const int64_t *__restrict__ ci,
                                                                  int64_t n, FlowDirOut o, const unsigned long long *__restrict__ redo)
{
    typedef typename std::conditional<AF64, double, float>::type AmtT;
    const AmtT *am = (const AmtT *)amount;
    __shared__ RsRec s_rec[RS_CH];
    __shared__ RsRec s_grp[RS_CH / RS_GRP];
    __shared__ RsRec s_sub[RS_CH / RS_SUB];
    __shared__ double s_tot[RS_CH], s_abs[RS_CH], s_cabs[RS_CH];
    // per-wave tiles while the records are made; afterwards the POOL: the terms of the chunks that will be added term by term
    __shared__ double s_stage[(RS_POOL + 1) * 64];                       // (+ one slot for a chunk fetched during the walk)
    __shared__ unsigned long long s_flow[RS_CH];
    __shared__ double s_carry[4];                                        // s, mn, mx, started (wave 0 -> all)
    __shared__ long long s_nflow;
    __shared__ int s_npool;
    constexpr int POOL_N = RS_POOL;
    static_assert(RS_POOL * 64 >= RS_WAVES * BF_SLOTS, "the tiles are staged in the pool's memory");
    const int lane = fmk_lane();
    const int w = fmk_uniform((int)(threadIdx.x >> 6));
    const int64_t count = (int64_t)redo[0];
    // item = column * count + list entry: the flagged columns of ONE bar land on different workgroups
    for (int64_t item = blockIdx.x; item < count * 7; item += gridDim.x) {
        const int row = (int)(item / count);
        const unsigned long long entry = redo[32 + item % count];
        if (!((entry >> (48 + row)) & 1)) continue;                      // this column is not near a float32 tie (block-uniform)
        const int64_t b = (int64_t)(entry & 0xFFFFFFFFFFFFULL);
        const int64_t start = ci[b] + 1, e_bar = ci[b + 1];
        const int64_t n_ch = (e_bar - start + 64) >> 6;
        const bool extrema = row >= 5;
        __syncthreads();
        if (threadIdx.x == 0) { s_carry[0] = 0.0; s_carry[1] = 1e9; s_carry[2] = -1e9; s_carry[3] = 0.0; s_nflow = 0; }   // base.py:461-464
        __syncthreads();
#ifdef BF_REDO_TIMING
        unsigned long long t_last_ = wall_clock64();
#endif
        for (int64_t c0 = 0; c0 < n_ch; c0 += RS_CH) {
            const int nc = (int)(n_ch - c0 < RS_CH ? n_ch - c0 : RS_CH);
            // A wave takes TILES of eight chunks: the terms are computed in tick order across the lanes (coalesced loads, the next
            // tile's in flight while this one is worked on) and turned through LDS so that lane l owns the eight consecutive terms
            // 8 l .. 8 l + 7 -- a chunk is then eight lanes (half a DPP row): three butterfly steps per reduction instead of six
            // scan steps per chunk, eight chunks at a time.
            const int nt = (nc + 7) >> 3;
            double *stage = s_stage + w * BF_SLOTS;
            const int my_chunk = lane >> 3;                                  // chunk of the tile this lane's terms belong to
            auto load_raw = [&](int tile, RsRaw<AmtT> &r) {
                const int64_t tb = start + (c0 + (int64_t)tile * 8) * 64;
                if (tb + 511 <= e_bar) {                                     // (wave-uniform) one address per column, immediate offsets
                    const double *pb = price + tb + lane;
                    const AmtT *ab = am + tb + lane;
                    const int8_t *sb = side + tb + lane;
#pragma unroll
                    for (int c = 0; c < 8; ++c) { r.p[c] = pb[c * 64]; r.v[c] = ab[c * 64]; r.sd[c] = sb[c * 64]; }
                } else {
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        int64_t j = tb + c * 64 + lane;
                        j = j <= e_bar ? j : e_bar;
                        r.p[c] = price[j]; r.v[c] = am[j]; r.sd[c] = side[j];
                    }
                }
                const int64_t jp = fmk_wrap((tb <= e_bar ? tb : e_bar) - 1, n);
                r.pp = price[jp]; r.ps = side[jp];
            };
            auto terms = [&](int tile, const RsRaw<AmtT> &r, double (&x)[8]) -> unsigned {
                const int64_t tb = start + (c0 + (int64_t)tile * 8) * 64;
                unsigned fb = 0;
                double carry_p = r.pp;
                int carry_s = (tb == start && !(e_bar > start)) ? 0 : r.ps;     // base.py:485-488: a one-tick bar
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const bool valid = tb + c * 64 + lane <= e_bar;
                    const double p = r.p[c], v = (double)r.v[c];
                    const int sd = r.sd[c];
                    const bool buy = valid && sd == 1, sell = valid && sd == -1;
                    const double pv = p * v;
                    double t;
                    switch (row) {                                           // (block-uniform)
                    case 0: t = buy ? v : 0.0; break;
                    case 1: t = sell ? v : 0.0; break;
                    case 2: t = buy ? pv : 0.0; break;
                    case 3: t = sell ? pv : 0.0; break;
                    case 4: {                                                // base.py:485-500: |price change| where the side changes
                        const double pprev = fmk_dpp_shift_up1(p, carry_p);
                        const int sprev = fmk_dpp_shift_up1(sd, carry_s);
                        carry_p = fmk_last_lane(p); carry_s = fmk_last_lane(sd);
                        t = valid && sd != sprev ? fabs(p - pprev) : 0.0;
                        break;
                    }
                    case 5: t = buy ? v : sell ? -v : 0.0; break;
                    default: t = buy ? pv : sell ? -pv : 0.0; break;
                    }
                    x[c] = t;
                    fb |= (buy || sell) ? 1u << c : 0u;
                }
                return fb;
            };
            auto turn = [&](const double (&x)[8], double (&q)[8]) {          // tick order across lanes -> eight consecutive per lane
#pragma unroll
                for (int c = 0; c < 8; ++c) stage[(c * 8 + (lane >> 3)) * 9 + (lane & 7)] = x[c];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 8; ++i) q[i] = stage[lane * 9 + i];
                __builtin_amdgcn_wave_barrier();
            };
            // ---- A1: plain chunk totals (any order), sum of magnitudes, which ticks are signed
            {
                RsRaw<AmtT> rn;
                if (w < nt) load_raw(w, rn);
                for (int tile = w; tile < nt; tile += RS_WAVES) {
                    double x[8], q[8];
                    const unsigned fb = terms(tile, rn, x);
                    if (tile + RS_WAVES < nt) load_raw(tile + RS_WAVES, rn);
                    turn(x, q);
                    double lt = 0.0, la = 0.0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) { lt += q[i]; la += fabs(q[i]); }
                    const double t = fmk_half_sum(lt), ta = fmk_half_sum(la);
                    const int c = tile * 8 + my_chunk;
                    if ((lane & 7) == 0 && c < nc) { s_tot[c] = t; s_abs[c] = ta; s_cabs[c] = ta; }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const uint64_t fm = __ballot((fb >> k) & 1);
                        if (lane == 0 && tile * 8 + k < nc) s_flow[tile * 8 + k] = fm;
                    }
                }
            }
            __syncthreads();
            BF_T(3);
            // ---- A2: exclusive prefix of the totals (approximate), total magnitude of the block: wave 0, RS_CH / 64 entries per lane
            if (w == 0) {
                double loc[RS_CH / 64], acc = 0.0, aabs = 0.0;
                long long nf = 0;
#pragma unroll
                for (int k = 0; k < RS_CH / 64; ++k) {
                    const int c = lane * (RS_CH / 64) + k;
                    loc[k] = acc;
                    acc += c < nc ? s_tot[c] : 0.0;
                    aabs += c < nc ? s_abs[c] : 0.0;
                    nf += c < nc ? __popcll(s_flow[c]) : 0;
                }
                const double incl = fmk_dpp_iscan(acc, 0.0, FmkOpAdd());
                const double off = incl - acc;
                const double tabs = fmk_dpp_reduce(aabs, 0.0, FmkOpAdd());
                const int64_t nft = fmk_dpp_reduce((int64_t)nf, (int64_t)0, FmkOpAdd());
                if (lane == 0) { s_nflow += nft; s_npool = 0; }
#pragma unroll
                for (int k = 0; k < RS_CH / 64; ++k) {
                    const int c = lane * (RS_CH / 64) + k;
                    if (c < nc) { s_tot[c] = off + loc[k]; s_abs[c] = tabs; }
                }
            }
            __syncthreads();
            BF_T(4);
            // ---- A3: the chunk records (same tiles, now from L2; lane l holds terms 8 l .. 8 l + 7 of the tile, its chunk is its half row)
            const double s0 = s_carry[0];
            auto pow2 = [](int k) { return __longlong_as_double((long long)(k + 1023) << 52); };
            {
                RsRaw<AmtT> rn;
                if (w < nt) load_raw(w, rn);
                for (int tile = w; tile < nt; tile += RS_WAVES) {
                    double x[8], q[8];
                    terms(tile, rn, x);
                    if (tile + RS_WAVES < nt) load_raw(tile + RS_WAVES, rn);
                    turn(x, q);
                    const int c = tile * 8 + my_chunk;
                    const int cc = c < nc ? c : nc - 1;
                    const double Sa = s0 + s_tot[cc];                        // approximate start value of the chunk
                    const double margin = (fabs(s0) + s_abs[cc]) * 9.1e-13;  // 2^-40 (|s0| + sum |x|): >> the prefix's rounding error
                    const double as = fabs(Sa);
                    const bool neg = Sa < 0.0;
                    const bool usable = as > margin && as >= 2.3e-308 && as < INFINITY;
                    int e = (int)((__double_as_longlong(as) >> 52) & 0x7FF) - 1023;
                    const bool e_ok = e > -900 && e < 1000;
                    e = e_ok ? e : 0;
                    const double lo2 = pow2(e), hi2 = pow2(e + 1), lim = pow2(e - 1), half_g = pow2(e - 53);
                    const double C = 1.5 * lo2;
                    bool bad = false;
                    double acc = 0.0, pq[8];
                    unsigned tie_byte = 0, par_byte = 0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const double xs = neg ? -q[i] : q[i];
                        const double y = (C + xs) - C;                       // |xs| < 2^(e-1): C + xs stays in C's binade -> y = rnd_g(xs)
                        const double rr = xs - y;
                        const bool tie = fabs(rr) == half_g;
                        const double yf = tie ? xs - half_g : y;             // a tie counts as its LOWER grid point m g (exact)
                        bad |= !(fabs(xs) < lim);
                        tie_byte |= tie ? 1u << i : 0u;
                        // parity of yf / g: C / g is even and C + yf is exact, so it is the last mantissa bit of C + yf
                        par_byte |= (unsigned)(__double_as_longlong(C + yf) & 1) << i;
                        acc += yf;
                        pq[i] = acc;
                    }
                    // sum |y| <= sum |x| + 64 g < 2^(e+1) = 2^53 g (tested below): the lane's partial sums and the scan are exact.
                    // (A bound on the SUM, not 64 x the largest term: signed rows hover within a few hundred terms of zero.)
                    const double incl = fmk_half_iscan_add(acc, lane);
                    const double base = incl - acc;
                    const double Tf = fmk_half_sum(acc);                     // (multiples of g: exact in any order)
                    double rmin0, rmax0, rmin1, rmax1, T0, T1;
                    int pout;
                    uint64_t TM = 0;
                    if (__ballot(tie_byte != 0) == 0) {                      // (no tie in the wave's eight chunks: the common case)
                        double pmin = INFINITY, pmax = -INFINITY;
#pragma unroll
                        for (int i = 0; i < 8; ++i) { const double P = base + pq[i]; pmin = bf_min(pmin, P); pmax = bf_max(pmax, P); }
                        rmin0 = rmin1 = fmk_half_min(pmin); rmax0 = rmax1 = fmk_half_max(pmax);
                        T0 = T1 = Tf;
                        int lp = (int)(__builtin_popcount(par_byte) & 1);    // parity of the chunk's total: xor over the half row
                        lp ^= __builtin_amdgcn_update_dpp(lp, lp, DPP_XOR1, 0xF, 0xF, false);
                        lp ^= __builtin_amdgcn_update_dpp(lp, lp, DPP_XOR2, 0xF, 0xF, false);
                        lp ^= __builtin_amdgcn_update_dpp(lp, lp, DPP_HALF_MIRROR, 0xF, 0xF, false);
                        pout = lp & 1;
                    } else {
                        // the chunk's ties, resolved in order for incoming parity 0 (cm0: the ties that round UP); parity 1 flips the first
                        const int sh = 8 * (lane & 7);
                        TM = fmk_half_or((uint64_t)tie_byte << sh);
                        const uint64_t AM = fmk_half_or((uint64_t)par_byte << sh);
                        uint64_t cm0 = 0, first_bit = 0;
                        {
                            uint64_t rest = TM;
                            int prev = -1, par = 0;
                            while (rest) {                                   // (uniform inside the half row; a handful of ties at most)
                                const int t = __builtin_ctzll(rest);
                                rest &= rest - 1;
                                const uint64_t between = AM & (((1ULL << t) - 1) & ~(prev >= 0 ? ((2ULL << prev) - 1) : 0ULL));
                                par ^= (int)(__builtin_popcountll(between) & 1);
                                const int c1 = par ^ (int)((AM >> t) & 1);   // (S + m) & 1
                                if (c1) cm0 |= 1ULL << t;
                                if (prev < 0) first_bit = 1ULL << t;
                                par = 0;                                     // even after a tie
                                prev = t;
                            }
                            const uint64_t after = AM & ~(prev >= 0 ? ((2ULL << prev) - 1) : 0ULL);
                            // with a tie: the parity after the chunk; without: the parity of the chunk's total (prev < 0: all of AM)
                            pout = (int)(__builtin_popcountll(after) & 1);
                        }
                        const uint64_t cm1 = cm0 ^ first_bit;
                        const double g1 = 2.0 * half_g;
                        double pmin0 = INFINITY, pmax0 = -INFINITY, pmin1 = INFINITY, pmax1 = -INFINITY;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const uint64_t upto = (2ULL << (sh + i)) - 1;    // the ticks of the chunk up to this one
                            const double P = base + pq[i];
                            const double P0 = P + g1 * (double)__builtin_popcountll(cm0 & upto), P1 = P + g1 * (double)__builtin_popcountll(cm1 & upto);
                            pmin0 = bf_min(pmin0, P0); pmax0 = bf_max(pmax0, P0);
                            pmin1 = bf_min(pmin1, P1); pmax1 = bf_max(pmax1, P1);
                        }
                        rmin0 = fmk_half_min(pmin0); rmax0 = fmk_half_max(pmax0);
                        rmin1 = fmk_half_min(pmin1); rmax1 = fmk_half_max(pmax1);
                        T0 = Tf + g1 * (double)__builtin_popcountll(cm0); T1 = Tf + g1 * (double)__builtin_popcountll(cm1);
                    }
                    const double rmin = bf_min(rmin0, rmin1), rmax = bf_max(rmax0, rmax1);
                    const uint64_t bb = __ballot(bad);
                    const bool seg_bad = ((bb >> (lane & ~7)) & 0xFF) != 0;
                    // the inclusive prefixes stay inside the binade by a margin that covers the approximate base
                    const bool good = usable && e_ok && !seg_bad && s_cabs[cc] < hi2 * 0.999999999 &&
                                      as + rmin > lo2 + margin && as + rmax < hi2 - margin && as > lo2 + margin && as < hi2 - margin;
#ifdef BF_REDO_WHY
                    constexpr bool getenv_why_by_row = BF_REDO_WHY == 2;      // -DBF_REDO_WHY=2: chunks with a tie / a large term, per column
                    if ((lane & 7) == 0 && c < nc && !good) {                // developer knob: why a chunk has no record (slots 3 ..)
                        const int why = !usable || !e_ok ? 0 : (seg_bad ? 1 : (!(s_cabs[cc] < hi2 * 0.999999999) ? 2 :
                                        (!(as + rmin > lo2 + margin) ? 3 : (!(as + rmax < hi2 - margin) ? 4 : 5))));
                        if (getenv_why_by_row) { if (why == 1) atomicAdd(&bf_redo_stats[3 + row], 1ULL); }
                        else atomicAdd(&bf_redo_stats[3 + why], 1ULL);
                    }
#endif
                    if ((lane & 7) == 0 && c < nc) {
                        RsRec r;
                        r.T[0] = good ? T0 : -1.0; r.T[1] = good ? T1 : -1.0;
                        r.minP[0] = good ? rmin0 : 0.0; r.minP[1] = good ? rmin1 : 0.0;
                        r.maxP[0] = good ? rmax0 : 0.0; r.maxP[1] = good ? rmax1 : 0.0;
                        r.e = good ? e : RS_NONE; r.neg = (short)neg;
                        r.has_tie = TM != 0; r.pout = (unsigned char)pout;
                        s_rec[c] = r;
                    }
                }
            }
            __syncthreads();
            BF_T(5);
            // ---- A4: group records (a thread per group of RS_GRP chunks); a pool slot for every chunk without a record (T = slot, or -1)
            const int ng = (nc + RS_GRP - 1) / RS_GRP, nsb = (nc + RS_SUB - 1) / RS_SUB;
            if ((int)threadIdx.x < nsb) {
                const int c0s = (int)threadIdx.x * RS_SUB;
                s_sub[threadIdx.x] = rs_compose(s_rec + c0s, nc - c0s < RS_SUB ? nc - c0s : RS_SUB);
            }
            __syncthreads();
            if ((int)threadIdx.x < ng) {
                const int b0 = (int)threadIdx.x * (RS_GRP / RS_SUB);
                s_grp[threadIdx.x] = rs_compose(s_sub + b0, nsb - b0 < RS_GRP / RS_SUB ? nsb - b0 : RS_GRP / RS_SUB);
            }
            __syncthreads();
            for (int c = (int)threadIdx.x; c < nc; c += 64 * RS_WAVES) {
                if (s_rec[c].e == RS_NONE) {
                    const int slot = atomicAdd(&s_npool, 1);
                    s_rec[c].T[0] = slot < POOL_N ? (double)slot : -1.0;
                }
            }
            __syncthreads();
            // ---- A5: the pool (all waves): the terms of those chunks, in tick order
            for (int c = w; c < nc; c += RS_WAVES) {
                if (s_rec[c].e != RS_NONE || s_rec[c].T[0] < 0.0) continue;    // (wave-uniform)
                bool fl;
                s_stage[(int)s_rec[c].T[0] * 64 + lane] = rs_chunk_term<AmtT>(row, start + (c0 + c) * 64, lane, start, e_bar, n, price, am, side, fl);
            }
            __syncthreads();
            BF_T(6);
            // ---- B: the walk, in order (wave 0; every lane carries the same values)
            if (w == 0) {
                double s = s_carry[0], mn = s_carry[1], mx = s_carry[2];
                bool started = s_carry[3] != 0.0;
                // one record (a chunk's or a group's): true when s, mn, mx have been advanced over it
                auto step = [&](const RsRec &r) -> bool {
                    const double as = fabs(s);
                    if (!((started || !extrema) && r.e != RS_NONE && as >= 2.3e-308 && (s < 0.0) == (r.neg != 0))) return false;
                    const int e = (int)((__double_as_longlong(as) >> 52) & 0x7FF) - 1023;
                    const int v = (int)(__double_as_longlong(as) & 1);       // parity of |s| / g when |s| is in the record's binade
                    const double lo2 = pow2(r.e), hi2 = pow2(r.e + 1);
                    const double lo_v = as + (v ? r.minP[1] : r.minP[0]), hi_v = as + (v ? r.maxP[1] : r.maxP[0]);     // exact inside the binade
                    if (!(e == r.e && lo_v > lo2 && hi_v < hi2)) return false;
                    mn = fmin(mn, s < 0.0 ? -hi_v : lo_v);
                    mx = fmax(mx, s < 0.0 ? -lo_v : hi_v);
                    const double rt = v ? r.T[1] : r.T[0];
                    s = s < 0.0 ? -(as + rt) : as + rt;
                    return true;
                };
                for (int g = 0; g < ng; ++g) {
                    if (step(s_grp[g])) continue;
#ifdef BF_REDO_TIMING
                    if (lane == 0) atomicAdd(&bf_redo_stats[8], 1ULL);       // groups walked part by part
#endif
                    const int c_end = (g + 1) * RS_GRP < nc ? (g + 1) * RS_GRP : nc;
                    for (int c = g * RS_GRP; c < c_end; ++c) {
                        // at a sub-group's first chunk: the four chunks in one step when they fit
                        if ((c & (RS_SUB - 1)) == 0 && step(s_sub[c / RS_SUB])) { c += RS_SUB - 1; continue; }
                        const RsRec r = s_rec[c];
                        if (step(r)) continue;
                        // term by term: EVERY lane adds all 64 terms (broadcast LDS reads): one to three instructions per term, no
                        // cross-lane traffic inside the dependent chain.  The terms wait in the pool; a chunk whose record did not
                        // fit the actual s (or that found the pool full) fetches them now.
                        int slot = RS_POOL;
                        if (r.e == RS_NONE && r.T[0] >= 0.0) slot = (int)r.T[0];
                        else {
                            bool fl;
                            s_stage[RS_POOL * 64 + lane] = rs_chunk_term<AmtT>(row, start + (c0 + c) * 64, lane, start, e_bar, n, price, am, side, fl);
#ifdef BF_REDO_TIMING
                            if (lane == 0) atomicAdd(&bf_redo_stats[9], 1ULL);   // chunks fetched during the walk (record did not fit s / pool full)
#endif
                        }
                        const uint64_t fm = s_flow[c];
                        __builtin_amdgcn_wave_barrier();
                        // the terms into registers in batches (ds_read_b128 of ONE LDS array: through a pointer that could be either of
                        // two arrays every read was a flat load the adds waited for), then the chain
                        const double *xs = s_stage + slot * 64;
#pragma unroll 1
                        for (int h = 0; h < 64; h += 16) {                   // sixteen terms at a time: 32 VGPRs (all 64: spills)
                            double xv[16];
#pragma unroll
                            for (int k = 0; k < 16; ++k) xv[k] = xs[h + k];
                            if (!extrema) {
#pragma unroll
                                for (int k = 0; k < 16; ++k) s += xv[k];
                            } else if (started) {
#pragma unroll
                                for (int k = 0; k < 16; ++k) { s += xv[k]; mn = bf_min(mn, s); mx = bf_max(mx, s); }
                            } else {
#pragma unroll
                                for (int k = 0; k < 16; ++k) {
                                    s += xv[k];
                                    started = started || ((fm >> (h + k)) & 1);
                                    if (started) { mn = fmin(mn, s); mx = fmax(mx, s); }
                                }
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                        if (lane == 0) atomicAdd(&bf_redo_stats[2], 1ULL);
                    }
                }
                if (lane == 0) { s_carry[0] = s; s_carry[1] = mn; s_carry[2] = mx; s_carry[3] = started ? 1.0 : 0.0; }
            }
            __syncthreads();
            BF_T(7);
        }
        if (threadIdx.x == 0) {
            atomicAdd(&bf_redo_stats[0], 1ULL);
            atomicAdd(&bf_redo_stats[1], (unsigned long long)n_ch);
#if !defined(BF_REDO_TIMING) && !defined(BF_REDO_WHY)
            atomicAdd(&bf_redo_stats[3 + row], 1ULL);
#endif
            const double sum = s_carry[0];
            switch (row) {
            case 0: o.volume_buy[b] = (float)sum; break;
            case 1: o.volume_sell[b] = (float)sum; break;
            case 2: o.dollars_buy[b] = (float)sum; break;
            case 3: o.dollars_sell[b] = (float)sum; break;
            case 4: o.mean_spread[b] = s_nflow == 0 ? NAN : (float)(sum / (double)s_nflow); break;
            case 5: o.cum_volumes_min[b] = (float)s_carry[1]; o.cum_volumes_max[b] = (float)s_carry[2]; break;
            default: o.cum_dollars_min[b] = (float)s_carry[1]; o.cum_dollars_max[b] = (float)s_carry[2]; break;
            }
        }
    }
}

// redo launch: the chunk-record kernel (FMK_DIR_FORCE_REDO=2: one wave per bar, seven lanes adding term by term)
template <bool AF64>
static void bf_redo_launch(fmk_ctx *ctx, unsigned rblocks, const double *d_price, const void *d_amount, const int8_t *d_side,
                           const int64_t *d_close_idx, int64_t n, const FlowDirOut &o, const unsigned long long *redo)
{
    const char *rv = getenv("FMK_DIR_FORCE_REDO");
    if (!rv || atoi(rv) != 2)
        k_bar_dir_redo_par<AF64><<<(unsigned)(ctx->n_cu * 2), 64 * RS_WAVES, 0, ctx->stream>>>(d_price, d_amount, d_side, d_close_idx, n, o, redo);
    else
        k_bar_dir_redo<AF64><<<rblocks, 256, 0, ctx->stream>>>(d_price, d_amount, d_side, d_close_idx, n, o, redo);
}


// ---------------------------------------------------------------------------------------
// Bars ordered by length (round 4).  The lane-per-bar schedule below is done with a wave when its LONGEST bar is:
// on bars of unequal length (real one-minute bars: lognormal, sigma ~1) most lanes idle.  A counting sort by quarter-octave length
// class (the four values of the two bits under the leading one: bars of a class differ by < 19 %), longest class first, gives every
// wave 64 bars of about the same length; inside a class the bars keep the order of their 1 024-bar blocks (locality).
//   k_bs_hist: per block of 1 024 bars, the count of every class -> hist[class][block]     (exclusive scan: fmk_scan.h)
//   k_bs_scatter: perm[offset[class][block] + rank inside (block, class)] = bar
// ---------------------------------------------------------------------------------------
#define BS_CLASSES 128
#define BS_BLOCK 1024
__device__ __forceinline__ int bs_class(int64_t len)            // 0: the longest ... 127: empty / negative
{
    if (len <= 0) return BS_CLASSES - 1;
    const int e = 63 - __builtin_clzll((unsigned long long)len);
    const int q = e >= 2 ? (int)((len >> (e - 2)) & 3) : (int)((len << (2 - e)) & 3);
    int c = 4 * e + q;                                           // grows with the length; < 126 for len < 2^31
    if (c > BS_CLASSES - 2) c = BS_CLASSES - 2;
    return BS_CLASSES - 2 - c;
}
__global__ __launch_bounds__(256) void k_bs_hist(const int64_t *__restrict__ ci, int64_t nb, int64_t nblk, int64_t *__restrict__ hist)
{
    __shared__ int cnt[BS_CLASSES];
    if (threadIdx.x < BS_CLASSES) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int r = 0; r < BS_BLOCK / 256; ++r) {
        const int64_t b = (int64_t)blockIdx.x * BS_BLOCK + r * 256 + threadIdx.x;
        if (b < nb) atomicAdd(&cnt[bs_class(ci[b + 1] - ci[b])], 1);
    }
    __syncthreads();
    if (threadIdx.x < BS_CLASSES) hist[(int64_t)threadIdx.x * nblk + blockIdx.x] = cnt[threadIdx.x];
}
__global__ __launch_bounds__(256) void k_bs_scatter(const int64_t *__restrict__ ci, int64_t nb, int64_t nblk,
                                                    const int64_t *__restrict__ off, int64_t *__restrict__ perm)
{
    __shared__ int cnt[BS_CLASSES];
    __shared__ int64_t base[BS_CLASSES];
    if (threadIdx.x < BS_CLASSES) { cnt[threadIdx.x] = 0; base[threadIdx.x] = off[(int64_t)threadIdx.x * nblk + blockIdx.x]; }
    __syncthreads();
    for (int r = 0; r < BS_BLOCK / 256; ++r) {
        const int64_t b = (int64_t)blockIdx.x * BS_BLOCK + r * 256 + threadIdx.x;
        if (b < nb) {
            const int c = bs_class(ci[b + 1] - ci[b]);
            perm[base[c] + atomicAdd(&cnt[c], 1)] = b;
        }
    }
}
// per class, the number of bars: tot[c] = sum over the blocks of hist[c][.]
__global__ __launch_bounds__(256) void k_bs_totals(const int64_t *__restrict__ hist, int64_t nblk, int64_t *__restrict__ tot)
{
    __shared__ int64_t ws[4];
    const int c = blockIdx.x;
    int64_t acc = 0;
    for (int64_t b = threadIdx.x; b < nblk; b += 256) acc += hist[(int64_t)c * nblk + b];
    acc = fmk_wave_sum(acc);
    if (fmk_lane() == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) tot[c] = ws[0] + ws[1] + ws[2] + ws[3];
}
// The census: hist (*hist_out: a block of the context's pool, BS_CLASSES x nblk counts + the class totals behind them) and, on the
// host, whether the bars are of UNEQUAL length: fewer than 80 % of them inside the best window of five adjacent classes (a factor
// 2.4 in length).  One small read-back; it replaces the one the lane schedule used to make after the fact.
static int bf_bar_census(fmk_ctx *ctx, const int64_t *d_ci, int64_t nb, int64_t **hist_out, bool *uneven)
{
    *hist_out = nullptr;
    *uneven = false;
    const int64_t nblk = fmk_ceil_div(nb, BS_BLOCK);
    void *p_hist = nullptr;
    FMK_TRY(fmk_alloc(ctx, (size_t)(BS_CLASSES * nblk + BS_CLASSES + 1) * 8, &p_hist));
    int64_t *hist = (int64_t *)p_hist, *tot = hist + BS_CLASSES * nblk + 1;
    k_bs_hist<<<(unsigned)nblk, 256, 0, ctx->stream>>>(d_ci, nb, nblk, hist);
    k_bs_totals<<<BS_CLASSES, 256, 0, ctx->stream>>>(hist, nblk, tot);
    int64_t h[BS_CLASSES];
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h, tot, sizeof(h), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)fmk_free(ctx, p_hist); return fmk_set_error(ctx, FMK_E_HIP, "bar census: %s", hipGetErrorString(e)); }
    int64_t best = 0, run = 0;
    for (int c = 0; c < BS_CLASSES - 1; ++c) {                       // (class 127: empty bars -- they cost nothing anywhere)
        run += h[c];
        if (c >= 5) run -= h[c - 5];
        best = run > best ? run : best;
    }
    *uneven = (double)best < 0.8 * (double)(nb - h[BS_CLASSES - 1]);
    *hist_out = hist;
    return FMK_OK;
}
// the bars in order of their class (longest first) from the census' counts -> *perm_out: a block of the context's pool (the caller
// frees it after queueing its kernels); `hist` is consumed (scanned in place)
static int bf_sort_bars(fmk_ctx *ctx, const int64_t *d_ci, int64_t nb, int64_t *hist, int64_t **perm_out)
{
    *perm_out = nullptr;
    const int64_t nblk = fmk_ceil_div(nb, BS_BLOCK);
    void *p_perm = nullptr;
    FMK_TRY(fmk_alloc(ctx, (size_t)nb * 8, &p_perm));
    int rc = fmk_exclusive_scan_i64(ctx, hist, hist, BS_CLASSES * nblk, false);
    if (rc == FMK_OK) {
        k_bs_scatter<<<(unsigned)nblk, 256, 0, ctx->stream>>>(d_ci, nb, nblk, hist, (int64_t *)p_perm);
        if (hipGetLastError() != hipSuccess) rc = fmk_set_error(ctx, FMK_E_HIP, "bf_sort_bars: launch failed");
    }
    if (rc != FMK_OK) { (void)fmk_free(ctx, p_perm); return rc; }
    *perm_out = (int64_t *)p_perm;
    return FMK_OK;
}

// ---------------------------------------------------------------------------------------
// directional only, ONE LANE PER BAR (round 2, float32 amounts): the schedule for streams of many moderate bars.
// A wave takes 64 consecutive bars; lane l walks bar l tick by tick with the reference's own loop (base.py:476-546: same
// operations, same order, float64 accumulators), so
//   * there is nothing to scan or to combine across lanes (the wave-per-bar schedule above spends as many VALU instructions
//     on its per-tile scans, tile shapes and carries as on the ticks: ~60 + ~45 per 64 ticks),
//   * every float64 sum is the reference's sequential sum: no float32 tie test, no redo pass.
// Memory.  The lanes' bars lie 1 200 ticks apart, so a lane's own loads would touch 64 lines per instruction.  Instead the
// wave stages, per step, the 16-tick aligned block each lane is at: row r of the tile = 128 B of price, 64 B of amount, 16 B
// of side of lane r's block, fetched by 16 / 8 / 4 fully coalesced instructions (16 / 8 / 4 lanes per row) into registers
// ONE STEP AHEAD, written to LDS (rows padded 16 -> 17 elements: conflict-free per-lane reads) when the previous step's walk
// is done.  Every 128-byte line is fetched once.
// A step in which every live lane is inside its bar and has met a signed tick runs without per-tick predicates (~39 VALU
// per tick-row); first / last steps of a bar run the predicated form.  Bars longer than `max_len` are left to k_bar_dir
// (list mode): one long bar would hold its wave for ~75 us per 1e3 ticks.
// ---------------------------------------------------------------------------------------
#define DL_T 16
#define DL_ROW 17
#define DL_WAVES 2                      // 29.6 KB of LDS per workgroup: five per CU
// OHLC: comp_bar_ohlcv's open / high / low / close / volume / vwap / trades (base.py:352-400) ride along -- the lane already holds
// the tick's price and amount; six more instructions per tick, the reference's own sequential sums.  (Not the median trade size:
// fmk_median_small_launch.)  Bars left on the list get theirs from k_bar_ohlcv (`any_long` is its go flag).
struct DlOhlcOut { double *open, *high, *low, *close; float *vol; double *vwap; int64_t *trades; int *any_long;
                   int64_t ohlc_max = INT64_MAX; /* bars longer than this get open .. trades from another launch (cfg 4, sorted bars) */ };
template <bool OHLC>
__global__ __launch_bounds__(64 * DL_WAVES) void k_bar_dir_lanes(const double *__restrict__ price, const float *__restrict__ amount,
                                                       const int8_t *__restrict__ side, const int64_t *__restrict__ ci,
                                                       int64_t nb, int64_t n, FlowDirOut o, unsigned long long *n_zero_div,
                                                       unsigned long long *long_list, int64_t max_len, DlOhlcOut oo,
                                                       const int64_t *__restrict__ perm = nullptr)
{
    __shared__ double s_p[DL_WAVES][64 * DL_ROW];
    __shared__ float s_a[DL_WAVES][64 * DL_ROW];
    __shared__ uint32_t s_s[DL_WAVES][64 * 5];                                // 16 side bytes + 4 of padding per row
    __shared__ int64_t s_blk[DL_WAVES][64];
    const int lane = fmk_lane();
    const int wib = fmk_uniform((int)(threadIdx.x >> 6));
    double *sP = s_p[wib];
    float *sA = s_a[wib];
    uint32_t *sS = s_s[wib];
    const signed char *sS8 = (const signed char *)sS;
    int64_t *sB = s_blk[wib];
    const int64_t nwaves = (int64_t)gridDim.x * DL_WAVES;
    for (int64_t w = (int64_t)blockIdx.x * DL_WAVES + wib; w * 64 < nb; w += nwaves) {
        // perm (may be null): the bars ordered by length, longest first (bf_sort_bars) -- a wave's 64 bars then are about equally long
        const bool has = w * 64 + lane < nb;
        const int64_t b = has ? (perm ? perm[w * 64 + lane] : w * 64 + lane) : nb;
        const int64_t s = has ? ci[b] : 0, e = has ? ci[b + 1] : 0;
        const int64_t len = e - s;
        // A lane walks ITS bar; the wave is done when its longest bar is.  That pays while the 64 bars are about equally long -- the
        // schedule spends ~39 VALU instructions per tick ROW against the ~55 per 64 ticks of the wave-per-bar kernel, so it needs
        // ~70 % of its lanes busy.  Real one-minute bars are not like that (lognormal lengths, sigma ~1: a quiet minute of 100 ticks
        // next to a busy one of 8 000): a wave whose bars fill less than 70 % of (64 x its longest bar) keeps only its short bars
        // (<= 192 ticks: a wave per bar is the worse deal for those) and lists the others for k_bar_dir.  Measured on 1e9 ticks in
        // lognormal bars of sigma 1 (tools/realbars.py): 8.3 ms with every bar on a lane, 3.9 ms with a wave per bar.
        int64_t thr_w = max_len;
        {
            const bool cand = has && len > 0 && len <= max_len;
            const int64_t wsum = fmk_dpp_reduce(cand ? len : (int64_t)0, (int64_t)0, FmkOpAdd());
            const int64_t wmax = fmk_dpp_reduce(cand ? len : (int64_t)0, (int64_t)0, FmkOpMax());
            const int nact = __builtin_popcountll(__builtin_amdgcn_ballot_w64(cand));
            // (bars in order of length -- perm -- fill their waves by construction, and the caller then relies on every bar up to
            //  max_len having been walked here)
            if (!perm && wmax > 192 && (double)wsum < 0.7 * (double)wmax * (double)nact) thr_w = 192;
        }
        const bool is_long = has && len > thr_w;
        const unsigned long long lb = __builtin_amdgcn_ballot_w64(is_long);
        if (lb) {
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(long_list, (unsigned long long)__builtin_popcountll(lb));
            base = (unsigned long long)fmk_uniform((int64_t)base);
            if (is_long) long_list[32 + base + __builtin_popcountll(lb & ((1ULL << lane) - 1))] = (unsigned long long)b;
            if constexpr (OHLC) {
                if (lane == 0 && __hip_atomic_load(oo.any_long, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                    __hip_atomic_store(oo.any_long, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        const bool active = has && len > 0 && !is_long;
        const int64_t start = s + 1;
        const int64_t first_blk = start >> 4;
        const int nsteps = active ? (int)((e >> 4) - first_blk + 1) : 0;
        const int steps = fmk_dpp_reduce(nsteps, 0, FmkOpMax());
        // the reference's accumulators (base.py:476-488)
        double vb = 0.0, vs = 0.0, db = 0.0, ds = 0.0, cs = 0.0, mxs = 0.0, cv = 0.0, cd = 0.0;
        double vmin = 1e9, vmax = -1e9, dmin = 1e9, dmax = -1e9;
        int ct = 0, tmin = BF_INIT_MIN, tmax = BF_INIT_MAX, nbuy = 0, nsell = 0;
        bool seen = false;                                             // a signed tick has updated the extrema
        double b_first = 0.0, b_hi = 0.0, b_lo = 0.0, b_tv = 0.0, b_td = 0.0;       // comp_bar_ohlcv's loop state (OHLC)
        bool opened = false;
        double pp = 0.0;
        int ps = 0;
        if (active) {
            pp = price[fmk_wrap(s, n)];
            ps = len > 1 ? (int)side[fmk_wrap(s, n)] : 0;              // base.py:485-488
        }
        double pr[16];
        float2 ar[8];
        uint32_t sr[4];
        // every block of the wave's bars ends before the arrays do (all but the stream's last wave): no bound checks on the loads,
        // idle rows re-read the wave's first block (28 loads per step: the checks were ~110 of a step's ~1000 VALU instructions)
        const int64_t e_max = fmk_dpp_reduce(active ? e : (int64_t)0, (int64_t)0, FmkOpMax());
        const bool safe = e_max + DL_T < n;
        const int64_t blk_any = fmk_uniform(fmk_dpp_reduce(active ? first_blk : (int64_t)INT64_MAX, (int64_t)INT64_MAX, FmkOpMin()));
        auto issue = [&](int step) {                                   // the blocks of `step` -> registers
            if (safe) {
                sB[lane] = step < nsteps ? first_blk + step : blk_any;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < 16; ++k) pr[k] = price[sB[4 * k + (lane >> 4)] * DL_T + (lane & 15)];
#pragma unroll
                for (int k = 0; k < 8; ++k) ar[k] = *(const float2 *)(amount + sB[8 * k + (lane >> 3)] * DL_T + (lane & 7) * 2);
#pragma unroll
                for (int k = 0; k < 4; ++k) sr[k] = *(const uint32_t *)(side + sB[16 * k + (lane >> 2)] * DL_T + (lane & 3) * 4);
                __builtin_amdgcn_wave_barrier();
                return;
            }
            sB[lane] = step < nsteps ? first_blk + step : (int64_t)-1;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int64_t rb = sB[4 * k + (lane >> 4)];
                int64_t idx = rb * DL_T + (lane & 15);
                idx = idx < n ? idx : n - 1;
                pr[k] = rb >= 0 ? price[idx] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int64_t rb = sB[8 * k + (lane >> 3)];
                const int64_t idx = rb * DL_T + (lane & 7) * 2;
                float2 v = make_float2(0.f, 0.f);
                if (rb >= 0) {
                    if (idx + 1 < n) v = *(const float2 *)(amount + idx);
                    else if (idx < n) v.x = amount[idx];
                }
                ar[k] = v;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t rb = sB[16 * k + (lane >> 2)];
                const int64_t idx = rb * DL_T + (lane & 3) * 4;
                uint32_t v = 0;
                if (rb >= 0) {
                    if (idx + 3 < n) v = *(const uint32_t *)(side + idx);
                    else
                        for (int q = 0; q < 4; ++q)
                            if (idx + q < n) v |= (uint32_t)(uint8_t)side[idx + q] << (8 * q);
                }
                sr[k] = v;
            }
            __builtin_amdgcn_wave_barrier();
        };
        if (steps > 0) issue(0);
        for (int step = 0; step < steps; ++step) {
#pragma unroll
            for (int k = 0; k < 16; ++k) sP[(4 * k + (lane >> 4)) * DL_ROW + (lane & 15)] = pr[k];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int at = (8 * k + (lane >> 3)) * DL_ROW + (lane & 7) * 2;
                sA[at] = ar[k].x; sA[at + 1] = ar[k].y;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) sS[(16 * k + (lane >> 2)) * 5 + (lane & 3)] = sr[k];
            __builtin_amdgcn_wave_barrier();
            if (step + 1 < steps) issue(step + 1);
            // ---- the lane's 16 ticks
            const bool live = step < nsteps;
            const int64_t blk0 = (first_blk + step) * DL_T;
            int lo = 16, hi = -1;
            if (live) {
                lo = start > blk0 ? (int)(start - blk0) : 0;
                hi = e - blk0 < 15 ? (int)(e - blk0) : 15;
            }
            const bool fast = __builtin_amdgcn_ballot_w64(live && !(lo == 0 && hi == 15 && seen && (!OHLC || opened))) == 0;
            const double *rowP = sP + lane * DL_ROW;
            const float *rowA = sA + lane * DL_ROW;
            const signed char *rowS = sS8 + lane * 20;
            if (fast) {
                if (live) {
#pragma unroll
                for (int j = 0; j < DL_T; ++j) {
                    const double p = rowP[j];
                    const double a = (double)rowA[j];
                    const int sd = (int)rowS[j];
                    const double sp = sd != ps ? fabs(p - pp) : 0.0;   // base.py:495-500 (max / += of 0.0 change nothing)
                    mxs = bf_max(mxs, sp);
                    cs += sp;
                    pp = p; ps = sd;
                    const double pv = p * a;
                    if constexpr (OHLC) {                              // base.py:377-391 (every live lane has opened its bar)
                        b_hi = p > b_hi ? p : b_hi;
                        b_lo = p < b_lo ? p : b_lo;
                        b_tv += a;
                        b_td += pv;
                    }
                    const bool buy = sd == 1, sell = sd == -1;
                    // x += cond ? y : 0.0 as fma(1.0 or 0.0, y, x): the product is exact, so the one rounding is the addition's
                    // (two instructions instead of two selects and an add); likewise the signed terms with sf = -1, 0, +1
                    const double ib = buy ? 1.0 : 0.0, is = sell ? 1.0 : 0.0;
                    vb = fma(ib, a, vb); db = fma(ib, pv, db); nbuy += buy ? 1 : 0;
                    vs = fma(is, a, vs); ds = fma(is, pv, ds); nsell += sell ? 1 : 0;
                    const double sf = (double)sd;
                    ct += sd; cv = fma(sf, a, cv); cd = fma(sf, pv, cd);
                    tmin = ct < tmin ? ct : tmin; tmax = ct > tmax ? ct : tmax;    // an unsigned tick repeats a candidate
                    vmin = bf_min(vmin, cv); vmax = bf_max(vmax, cv);
                    dmin = bf_min(dmin, cd); dmax = bf_max(dmax, cd);
                }
                }
            } else {
#pragma unroll 4
                for (int j = 0; j < DL_T; ++j) {
                    if (j < lo || j > hi) continue;
                    const double p = rowP[j];
                    const double a = (double)rowA[j];
                    const int sd = (int)rowS[j];
                    if (sd != ps) {
                        const double sp = fabs(p - pp);
                        mxs = bf_max(mxs, sp);
                        cs += sp;
                    }
                    pp = p; ps = sd;
                    const double pv = p * a;
                    if constexpr (OHLC) {
                        if (!opened) { b_first = p; b_hi = p; b_lo = p; opened = true; }     // base.py:371-372
                        if (p > b_hi) b_hi = p;
                        if (p < b_lo) b_lo = p;
                        b_tv += a;
                        b_td += pv;
                    }
                    if (sd == 1) { nbuy += 1; vb += a; db += pv; ct += 1; cv += a; cd += pv; }
                    else if (sd == -1) { nsell += 1; vs += a; ds += pv; ct -= 1; cv -= a; cd -= pv; }
                    else continue;
                    seen = true;
                    tmin = ct < tmin ? ct : tmin; tmax = ct > tmax ? ct : tmax;
                    vmin = bf_min(vmin, cv); vmax = bf_max(vmax, cv);
                    dmin = bf_min(dmin, cd); dmax = bf_max(dmax, cd);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // ---- one bar per lane: coalesced stores
        const bool zero = has && !is_long && nbuy + nsell == 0;
        const unsigned long long zb = __builtin_amdgcn_ballot_w64(zero);
        if (zb && lane == 0 && n_zero_div) atomicAdd(n_zero_div, (unsigned long long)__builtin_popcountll(zb));
        if (has && !is_long) {
            o.ticks_buy[b] = nbuy; o.ticks_sell[b] = nsell;
            o.volume_buy[b] = (float)vb; o.volume_sell[b] = (float)vs;
            o.dollars_buy[b] = (float)db; o.dollars_sell[b] = (float)ds;
            o.max_spread[b] = (float)mxs;
            o.mean_spread[b] = nbuy + nsell == 0 ? NAN : (float)(cs / (double)(nbuy + nsell));
            o.cum_ticks_min[b] = tmin; o.cum_ticks_max[b] = tmax;
            o.cum_volumes_min[b] = (float)vmin; o.cum_volumes_max[b] = (float)vmax;
            o.cum_dollars_min[b] = (float)dmin; o.cum_dollars_max[b] = (float)dmax;
            if constexpr (OHLC) {
                if (len > oo.ohlc_max) {
                    // (written by comp_bar_ohlcv's size classes on the auxiliary stream, beside this kernel)
                } else if (len > 0) {
                    oo.open[b] = b_first; oo.close[b] = pp;            // pp: the price of the bar's last tick
                    oo.high[b] = b_hi; oo.low[b] = b_lo;
                    oo.vol[b] = (float)b_tv;
                    oo.vwap[b] = b_tv > 0.0 ? b_td / b_tv : 0.0;       // base.py:398
                    oo.trades[b] = len;
                } else {                                               // empty bar: previous close (base.py:352-361)
                    const double pz = price[fmk_wrap(e, n)];
                    oo.open[b] = pz; oo.high[b] = pz; oo.low[b] = pz; oo.close[b] = pz;
                    oo.vol[b] = 0.f; oo.vwap[b] = 0.0; oo.trades[b] = 0;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// cfg 4, first half (round 2): comp_bar_ohlcv (+ median) INSIDE the directional kernel -- float32 amounts.
// The tile loader already holds 8 strided ticks per lane in registers before it transposes them into LDS: max / min /
// sum(vol) / sum(price*vol) over those registers cost ~5 VALU instructions per 64 ticks (every instruction covers 64 ticks),
// and their bit patterns are the median's keys: up to three 512-tick tiles (1536 ticks) stay in 24 registers per lane and
// feed the exact order-statistic search of fmk_median.h when the bar ends.  One read of price / amount / side (13 B/tick)
// then serves build_ohlcv AND build_directional_features: cfg 4 reads 26 B/tick instead of 38.  MEASURED at 1e9 ticks
// (tools/cfg4bench.py, profiles/r02_cfg4.txt): this kernel 5.7 ms at its natural 153 VGPRs (3 waves per SIMD), 5.3 ms held to
// 128 (launch bounds, 68 B of scratch) -- against 2.31 + 2.95 ms for k_bar_ohlcv_small + k_bar_dir back to back: the SAME
// time, a third fewer bytes.  The directional walk is VALU-bound, so a shared read buys bytes, not milliseconds.
// Bars longer than 1536 ticks get their median from k_bar_median (flag `saw_long`).
// ---------------------------------------------------------------------------------------
struct FlowOhlcvOut {
    double *open, *high, *low, *close;
    float *vol;
    double *vwap;
    int64_t *trades;
    double *median;
};
#define BF_MED_TILES 3

template <bool MEDIAN>
__global__ __launch_bounds__(256, 4) void k_bar_ohlcv_dir(const double *__restrict__ price, const float *__restrict__ amount,
                                                       const int8_t *__restrict__ side, const int64_t *__restrict__ ci,
                                                       int64_t nb, int64_t n, FlowDirOut o, unsigned long long *n_zero_div,
                                                       unsigned long long *redo, FlowOhlcvOut oo, int *saw_long)
{
    typedef MedKey<false> MK;
    __shared__ double s_p[4][BF_SLOTS];
    __shared__ float s_a[4][BF_SLOTS];
    __shared__ int8_t s_s[4][640];
    __shared__ uint32_t s_buf[4][64];
    const int lane = fmk_lane();
    const int wib = fmk_uniform((int)(threadIdx.x >> 6));
    double *sP = s_p[wib];
    float *sA = s_a[wib];
    int8_t *sS = s_s[wib];
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wib;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t b = wave0; b < nb; b += nwaves) {
        const int64_t s = fmk_uniform(ci[b]);
        const int64_t e = fmk_uniform(ci[b + 1]);
        const int64_t start = s + 1;
        FlowDir d;
        bf_dir_init(d);
        if (e > s) {
            d.prev_price = price[fmk_wrap(start - 1, n)];
            d.prev_side = e - s > 1 ? (int)side[fmk_wrap(start - 1, n)] : 0;    // base.py:485-488
        }
        double hi = -INFINITY, lo = INFINITY, tv = 0.0, td = 0.0;
        MedBar<false, BF_MED_TILES * 8, false> bar;
        if constexpr (MEDIAN) {
#pragma unroll
            for (int k = 0; k < BF_MED_TILES * 8; ++k) bar.key[k] = MK::MAXK;
        }
        int64_t j0 = start, rem = e - s;
        // one tile: coalesced loads -> OHLCV partials (+ keys) from the registers -> transposed LDS tile -> directional walk
        auto tile = [&](auto ti_c) {
            constexpr int TI = decltype(ti_c)::value;                 // < BF_MED_TILES: the tile's keys are kept
            int lr, tn;
            bf_shape(rem, lr, tn);
            const double *pb = price + j0;
            const float *ab = amount + j0;
            const int8_t *sb = side + j0;
            double pr[8];
            float ar[8];
            int sr[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                pr[c] = 0.0; ar[c] = 0.f; sr[c] = 0;
                const int t = c * 64 + lane;
                if (c < (1 << lr) && t < tn) { pr[c] = pb[t]; ar[c] = ab[t]; sr[c] = sb[t]; }
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const bool valid = c < (1 << lr) && c * 64 + lane < tn;
                const double a = (double)ar[c];
                hi = valid ? fmax(hi, pr[c]) : hi;                    // NaN prices lose (base.py:380-383)
                lo = valid ? fmin(lo, pr[c]) : lo;
                tv += valid ? a : 0.0;
                td += valid ? pr[c] * a : 0.0;
                if constexpr (MEDIAN && TI < BF_MED_TILES) {
                    if (valid) bar.key[TI * 8 + c] = MK::tokey(__float_as_uint(ar[c]));
                }
            }
            const int slot0 = bf_slot(lane, lr);
            const int cstride = (64 >> lr) * ((1 << lr) + 1);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (c < (1 << lr)) {
                    const int sl = slot0 + c * cstride;
                    sP[sl] = pr[c]; sA[sl] = ar[c]; sS[sl] = (int8_t)sr[c];
                }
            }
            __builtin_amdgcn_wave_barrier();
            bf_dir_tile<float>(lane, lr, tn, sP, sA, sS, d);
            __builtin_amdgcn_wave_barrier();
            j0 += tn;
            rem -= tn;
        };
        if (rem > 0) tile(std::integral_constant<int, 0>{});
        if (rem > 0) tile(std::integral_constant<int, 1>{});
        if (rem > 0) tile(std::integral_constant<int, 2>{});
        while (rem > 0) tile(std::integral_constant<int, BF_MED_TILES>{});
        bf_dir_emit<float>(o, b, lane, d, n_zero_div, start, e, redo);
        // ---- comp_bar_ohlcv (base.py:306-407)
        const int64_t cnt = e - s;
        if (cnt <= 0) {                                               // empty bar: previous close (base.py:352-361)
            if (lane == 0) {
                const double pz = price[fmk_wrap(e, n)];
                oo.open[b] = pz; oo.high[b] = pz; oo.low[b] = pz; oo.close[b] = pz;
                oo.vol[b] = 0.f; oo.vwap[b] = 0.0; oo.trades[b] = 0;
                if (oo.median) oo.median[b] = 0.0;
            }
            continue;
        }
        hi = fmk_dpp_reduce(hi, (double)-INFINITY, FmkOpMax());
        lo = fmk_dpp_reduce(lo, (double)INFINITY, FmkOpMin());
        tv = fmk_dpp_reduce(tv, 0.0, FmkOpAdd());
        td = fmk_dpp_reduce(td, 0.0, FmkOpAdd());
        if (lane == 0) {
            const double first = price[start];
            oo.open[b] = first;
            oo.close[b] = price[e];
            oo.high[b] = first != first ? first : hi;                 // a NaN first price never loses (base.py:371-382)
            oo.low[b] = first != first ? first : lo;
            oo.vol[b] = (float)tv;                                    // float32 amounts: exact in any order
            oo.vwap[b] = tv > 0.0 ? td / tv : 0.0;                    // base.py:398
            oo.trades[b] = cnt;
        }
        if constexpr (MEDIAN) {
            if (cnt <= (int64_t)BF_MED_TILES * 512) {
                bar.amount = amount; bar.start = start; bar.cnt = cnt; bar.lane = lane;
                const double m = med_search<false, BF_MED_TILES * 8, false>(bar, s_buf[wib]);
                if (lane == 0) oo.median[b] = m;
            } else if (lane == 0 && __hip_atomic_load(saw_long, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                __hip_atomic_store(saw_long, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // k_bar_median takes it
            }
        }
    }
}

#include "fmk_fused.h"

// ---------------------------------------------------------------------------------------
// cfg 4 in one pass (fmk_fused.h): host side
// ---------------------------------------------------------------------------------------
struct FuState {                      // what the sizing call leaves for the fill call (ctx->fused)
    void *block;                      // one allocation: staged rows [nb * FU_LV] x 4 arrays, L[nb], fp_list[nb + 32], the kernel's FuArgs
    FuStage stg;
    unsigned long long *fp_list;
    int64_t nb, n_fp;
    const int64_t *ci;                // the close indices the rows belong to (the fill call must name the same)
};

void fmk_fused_release(fmk_ctx *ctx)
{
    FuState *st = (FuState *)ctx->fused;
    if (!st) return;
    if (st->block) (void)fmk_free(ctx, st->block);
    free(st);
    ctx->fused = nullptr;
}

// Does the one-pass kernel serve this tape?  float32 amounts; bars of 96 .. ~1 400 ticks on average (shorter: the several-bars-per-wave
// schedules are ahead; longer: most ticks would lie in bars beyond FU_MAXT); a sample of the amounts certifies -- whole multiples of
// their common power of two below 2^23 of them (full-mantissa sizes never do: every bar would come back on the lists after a wasted
// sweep; such tapes get the kernel without its histogram, see bars_flow_fused_ok).
__global__ __launch_bounds__(256) void k_fu_census(const float *__restrict__ amount, int64_t n, int *__restrict__ out /* [0] min low bit, [1] max exponent, [2] bad */)
{
    const int64_t stride = n / 4096 > 0 ? n / 4096 : 1;
    int lb = FP_Q_UNKNOWN, mx = -1000, bad = 0;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < 4096; k += (int64_t)gridDim.x * blockDim.x) {
        // 16 consecutive amounts at each of 4 096 places
        const int64_t j0 = k * stride;
        for (int64_t j = j0; j < j0 + 16 && j < n; ++j) {
            const float a = amount[j];
            const int l = fp_lowbit_exp(a);
            if (l == (int)0x80000000 || a < 0.f) bad = 1;
            else if (l != FP_Q_UNKNOWN) { lb = l < lb ? l : lb; int ex; (void)frexpf(a, &ex); mx = ex > mx ? ex : mx; }
        }
    }
    lb = fmk_dpp_reduce(lb, (int)FP_Q_UNKNOWN, FmkOpMin());
    mx = fmk_dpp_reduce(mx, -1000, FmkOpMax());
    if (fmk_lane() == 0) { atomicMin(out, lb); atomicMax(out + 1, mx); if (bad) atomicOr(out + 2, 1); }
    if (__ballot(bad != 0) != 0 && fmk_lane() == 0) atomicOr(out + 2, 1);
}

// -> *mode: 0 the schedules below, 1 the one-pass kernel with the footprint histogram (sizes that certify), 2 the one-pass kernel for
// OHLCV + median + order flow only (float32 sizes that do not: full mantissas; the footprints by their own tick-ordered sweep).
// FMK_FUSED: 0 never, 2 / 3 force mode 1 / 2 whenever the dtype allows (tests), unset / 1: by the rules above.
static int bars_flow_fused_ok(fmk_ctx *ctx, const void *d_amount, int amount_is_f64, int64_t n, const int64_t *d_ci, int64_t nb,
                              int *mode_out)
{
    *mode_out = 0;
    const char *fv = getenv("FMK_FUSED");
    const int mode = fv ? atoi(fv) : 1;
    if (mode == 0 || amount_is_f64 || nb < 1) return FMK_OK;
    if (mode == 2 || mode == 3) { *mode_out = mode - 1; return FMK_OK; }
    const int64_t mean = n / nb;
    // Where the one pass wins (tools/cfg4bench.py N interval, 1e9 ticks, profiles/r06_cfg4_fused.txt): its per-bar tail (~1 200 instructions
    // whatever the bar's length) loses to the lane schedules below ~240 ticks per bar -- 26.8 against 17.7 ms at 100, 17.8 / 13.9 at 160,
    // 12.1 / 12.5 at 240, 10.2 / 11.2 at 300, 6.3 / 7.9 at 600, 4.9 / 7.3 at 1 200, 4.8 / 8.7 at 1 400 -- and beyond one tile per bar (1 800:
    // 12.6 / 8.9).  Without the histogram (sizes that do not certify) it only pays where the two-pass form starts to hand bars to its
    // long-bar classes: 7.05 / 7.33 at 1 260, 7.32 / 7.88 at 1 340, 7.45 / 8.56 at 1 400, but 9.3 / 8.2 at 600 and 14.6 / 11.9 at 300.
    if (mean < 256 || mean > 1400 || nb < (int64_t)ctx->n_cu * 8) return FMK_OK;
    int *d = (int *)(ctx->d_mail + 44);
    const int init[3] = {FP_Q_UNKNOWN, -1000, 0};
    FMK_HIP(ctx, hipMemcpyAsync(d, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
    k_fu_census<<<16, 256, 0, ctx->stream>>>((const float *)d_amount, n, d);
    FMK_LAUNCH_CHECK(ctx);
    int got[3];
    FMK_HIP(ctx, hipMemcpyAsync(got, d, sizeof got, hipMemcpyDeviceToHost, ctx->stream));
    // ... and the bars are of about equal length (the census of bf_bar_census: 80 % of them within a factor 2.4).  On a tape of
    // lognormal lengths most TICKS lie in bars of several tiles, whose medians the long-bar kernels take in a pass of their own --
    // measured at sigma = 1, full-mantissa sizes: 14.2 ms against 11.9 for the sorted-lane schedule below.  One wait for both answers.
    const int64_t nblk = fmk_ceil_div(nb, BS_BLOCK);
    void *p_hist = nullptr;
    FMK_TRY(fmk_alloc(ctx, (size_t)(BS_CLASSES * nblk + BS_CLASSES + 1) * 8, &p_hist));
    int64_t *hist = (int64_t *)p_hist, *tot = hist + BS_CLASSES * nblk + 1;
    k_bs_hist<<<(unsigned)nblk, 256, 0, ctx->stream>>>(d_ci, nb, nblk, hist);
    k_bs_totals<<<BS_CLASSES, 256, 0, ctx->stream>>>(hist, nblk, tot);
    int64_t h[BS_CLASSES];
    hipError_t ce = hipGetLastError();
    if (ce == hipSuccess) ce = hipMemcpyAsync(h, tot, sizeof(h), hipMemcpyDeviceToHost, ctx->stream);
    if (ce == hipSuccess) ce = hipStreamSynchronize(ctx->stream);
    (void)fmk_free(ctx, p_hist);
    if (ce != hipSuccess) return fmk_set_error(ctx, FMK_E_HIP, "bar census: %s", hipGetErrorString(ce));
    {
        int64_t best = 0, run = 0;
        for (int c = 0; c < BS_CLASSES - 1; ++c) {
            run += h[c];
            if (c >= 5) run -= h[c - 5];
            best = run > best ? run : best;
        }
        if ((double)best < 0.8 * (double)(nb - h[BS_CLASSES - 1])) return FMK_OK;      // uneven: the schedules below
    }
    // largest sampled amount < 2^mx: below 2^23 units of 2^lb when mx - lb <= 23 (one binade of slack for what the sample missed)
    const bool certifies = got[2] == 0 && (got[0] == FP_Q_UNKNOWN || got[1] - got[0] <= 22);
    *mode_out = certifies ? 1 : ((got[2] == 0 && mean >= 1250) ? 2 : 0);   // (negative / non-finite sizes in the sample: the schedules below)
    return FMK_OK;
}

static int bars_flow_fused(fmk_ctx *ctx, const double *d_price, const float *d_amount, int64_t n, const int64_t *d_close_idx,
                           int64_t n_idx, const int8_t *d_side, double price_tick_size, double *d_open, double *d_high, double *d_low,
                           double *d_close, float *d_volume, double *d_vwap, int64_t *d_trades, double *d_median,
                           const fmk_directional_out *d_dir, int64_t *d_n_zero_div, int64_t *d_level_offsets, int64_t *total_levels,
                           int64_t *max_levels, bool units)
{
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    FMK_TRY(bf_sync_force_redo(ctx));
    fmk_fused_release(ctx);
    const int64_t nb = n_idx - 1;
    FlowDirOut o;
    memcpy(&o, d_dir, sizeof(o));
    FuState *st = (FuState *)calloc(1, sizeof(FuState));
    if (!st) return fmk_set_error(ctx, FMK_E_NOMEM, "calloc");
    const size_t rows = units ? (size_t)nb * FU_LV : 0;                 // (no staging without the histogram)
    const size_t lbytes = ((size_t)nb * 4 + 255) & ~(size_t)255, fbytes = ((size_t)(nb + 32) * 8 + 255) & ~(size_t)255;
    int64_t blocks = fmk_ceil_div(nb, 4);
    {
        const int64_t cap = (int64_t)ctx->n_cu * 16;
        if (blocks > cap) blocks = cap;
        if (blocks < 1) blocks = 1;
    }
    const size_t sbytes = (size_t)blocks * 256 * 8;                      // the kernel's scrap area: 8 bytes per lane
    const size_t bytes = rows * 16 + lbytes + fbytes + ((sizeof(FuArgs) + 255) & ~(size_t)255) + sbytes;
    int rc = fmk_alloc(ctx, bytes, &st->block);
    if (rc != FMK_OK) { free(st); return rc; }
    unsigned char *blk = (unsigned char *)st->block;
    st->stg.bv = (float *)blk; st->stg.sv = (float *)(blk + rows * 4);
    st->stg.bc = (int *)(blk + rows * 8); st->stg.sc = (int *)(blk + rows * 12);
    st->stg.L = (int *)(blk + rows * 16);
    st->fp_list = (unsigned long long *)(blk + rows * 16 + lbytes);
    FuArgs *d_args = (FuArgs *)(blk + rows * 16 + lbytes + fbytes);
    unsigned long long *d_scrap = (unsigned long long *)(blk + rows * 16 + lbytes + fbytes + ((sizeof(FuArgs) + 255) & ~(size_t)255));
    st->nb = nb; st->ci = d_close_idx; st->n_fp = -1;
    ctx->fused = st;
    // two redo lists: the one-pass kernel's own (bars of <= FU_MAXT ticks: one wave per bar walks them, k_bar_dir_redo) and k_bar_dir's
    // (any length: the chunk-record kernel).  On the bench tape 0.19 % of the bars are TRUE near-ties -- its prices and sizes lie on
    // grids that put running dollar sums exactly on float32 midpoints -- and the chunk-record kernel took 0.41 ms for those 1 591 short bars
    unsigned long long *redo;
    rc = fmk_scratch(ctx, (size_t)(nb + 32) * 24, (void **)&redo);
    if (rc != FMK_OK) { fmk_fused_release(ctx); return rc; }
    unsigned long long *dir_list = redo + nb + 32, *redo_fu = redo + 2 * (nb + 32);
    int *saw_long = (int *)(ctx->d_mail + 18);
    FuLists li{redo_fu, redo, dir_list, st->fp_list, saw_long, saw_long + 1};
    auto fail = [&](int code) { fmk_fused_release(ctx); return code; };
#define FU_HIP(expr) do { const hipError_t e__ = (expr); if (e__ != hipSuccess) return fail(fmk_set_error(ctx, FMK_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e__))); } while (0)
    FU_HIP(hipMemsetAsync(redo, 0, 8, ctx->stream));
    FU_HIP(hipMemsetAsync(redo_fu, 0, 8, ctx->stream));
    FU_HIP(hipMemsetAsync(dir_list, 0, 8, ctx->stream));
    FU_HIP(hipMemsetAsync(st->fp_list, 0, 8, ctx->stream));
    FU_HIP(hipMemsetAsync(saw_long, 0, 2 * sizeof(int), ctx->stream));
    FuArgs h_args;
    h_args.amount = d_amount;
    h_args.oo = FuOhlcv{d_open, d_high, d_low, d_close, d_volume, d_vwap, d_trades, d_median};
    h_args.o = o; h_args.stg = st->stg; h_args.li = li; h_args.scrap = d_scrap;
    FU_HIP(hipMemcpyAsync(d_args, &h_args, sizeof h_args, hipMemcpyHostToDevice, ctx->stream));   // (pageable source: copied before the call returns)
    // The medians of the bars of several tiles (more than FU_MAXT ticks) need nothing but the amounts: the long-bar kernels run on the
    // context's auxiliary stream BESIDE the one-pass kernels (on a tape of lognormal bar lengths they are 2.5 ms of a 14 ms call when
    // they come behind).  They cannot see the one-pass kernel's flag there, so they are launched whatever the tape holds (on a tape
    // without such bars: an empty list).  While both streams carry launches of this call no freed block changes sides (fmk_pool_defer).
    bool side_med = false;
    if (d_median && nb >= 4096 && fmk_ctx_aux(ctx) == FMK_OK) {
        FU_HIP(hipEventRecord(ctx->aev[0], ctx->stream));
        FU_HIP(hipStreamWaitEvent(ctx->aux, ctx->aev[0], 0));
        (void)fmk_pool_defer(ctx, 1);
        hipStream_t keep = ctx->stream;
        ctx->stream = ctx->aux;
        rc = fmk_median_launch(ctx, d_amount, 0, d_close_idx, nb, FU_MAXT, nullptr, d_median, n);
        ctx->stream = keep;
        if (rc == FMK_OK && hipEventRecord(ctx->aev[1], ctx->aux) != hipSuccess) rc = fmk_set_error(ctx, FMK_E_HIP, "hipEventRecord");
        if (rc != FMK_OK) { (void)hipStreamSynchronize(ctx->aux); (void)fmk_pool_defer(ctx, 0); return fail(rc); }
        side_med = true;
    }
    struct SideGuard {                                                   // (every return below passes here)
        fmk_ctx *c; bool on;
        ~SideGuard() { if (on) { (void)hipStreamSynchronize(c->aux); (void)fmk_pool_defer(c, 0); } }
    } side_guard{ctx, side_med};
    if (units) {
        if (d_median)
            k_fu_bars<true, true><<<(unsigned)blocks, 256, 0, ctx->stream>>>(d_price, d_amount, d_side, d_close_idx, nb, n, price_tick_size, d_args);
        else
            k_fu_bars<false, true><<<(unsigned)blocks, 256, 0, ctx->stream>>>(d_price, d_amount, d_side, d_close_idx, nb, n, price_tick_size, d_args);
        FU_HIP(hipGetLastError());
        k_fu_long<true><<<(unsigned)(ctx->n_cu * 4), 256, 0, ctx->stream>>>(d_price, d_amount, d_side, d_close_idx, nb, n, price_tick_size, d_args);
    } else {
        if (d_median)
            k_fu_bars<true, false><<<(unsigned)blocks, 256, 0, ctx->stream>>>(d_price, d_amount, d_side, d_close_idx, nb, n, price_tick_size, d_args);
        else
            k_fu_bars<false, false><<<(unsigned)blocks, 256, 0, ctx->stream>>>(d_price, d_amount, d_side, d_close_idx, nb, n, price_tick_size, d_args);
        FU_HIP(hipGetLastError());
        k_fu_long<false><<<(unsigned)(ctx->n_cu * 4), 256, 0, ctx->stream>>>(d_price, d_amount, d_side, d_close_idx, nb, n, price_tick_size, d_args);
    }
    FU_HIP(hipGetLastError());
    // bars of more than FU_LONGEST ticks: open .. trades by comp_bar_ohlcv's leftover pass; of more than FU_MAXT ticks: the median by
    // the long-bar kernels (both look at their flag on the device)
    rc = fmk_ohlcv_leftover_launch(ctx, d_price, d_amount, 0, d_close_idx, nb, n, FU_LONGEST, saw_long + 1, d_open, d_high, d_low, d_close,
                                   d_volume, d_vwap, d_trades);
    if (rc == FMK_OK && d_median && !side_med) rc = fmk_median_launch(ctx, d_amount, 0, d_close_idx, nb, FU_MAXT, saw_long, d_median, n);
    if (rc != FMK_OK) return fail(rc);
    if (side_med) FU_HIP(hipStreamWaitEvent(ctx->stream, ctx->aev[1], 0));
    // the order flow of the listed bars (outside the class: empty, long, uncertified sizes, sides other than +-1, prices <= 0), then the
    // tick-order redo of both kernels' float32 ties
    int64_t dblocks = fmk_ceil_div(nb, 4);
    if (dblocks > 2048) dblocks = 2048;
    k_bar_dir<false><<<(unsigned)dblocks, 256, 0, ctx->stream>>>(d_price, d_amount, d_side, d_close_idx, nb, n, o,
                                                               (unsigned long long *)d_n_zero_div, redo, dir_list);
    FU_HIP(hipGetLastError());
    bf_redo_launch<false>(ctx, (unsigned)dblocks, d_price, d_amount, d_side, d_close_idx, n, o, redo);
    FU_HIP(hipGetLastError());
    k_bar_dir_redo<false><<<(unsigned)dblocks, 256, 0, ctx->stream>>>(d_price, d_amount, d_side, d_close_idx, n, o, redo_fu);
    FU_HIP(hipGetLastError());
    FU_HIP(hipMemcpyAsync(&ctx->h_mail[14], st->fp_list, 8, hipMemcpyDeviceToHost, ctx->stream));
    FU_HIP(hipMemcpyAsync(&ctx->h_mail[15], dir_list, 8, hipMemcpyDeviceToHost, ctx->stream));
    FU_HIP(hipMemcpyAsync(&ctx->h_mail[16], redo_fu, 8, hipMemcpyDeviceToHost, ctx->stream));
#undef FU_HIP
    rc = fmk_comp_bar_footprints_size_dev(ctx, d_low, d_high, nb, price_tick_size, d_level_offsets, total_levels, max_levels);
    if (rc != FMK_OK) return fail(rc);
    st->n_fp = ctx->h_mail[14];                                      // (the sizing call has waited for the stream)
    if (!units) fmk_fused_release(ctx);                              // nothing staged: the fill call is the ordinary one
    return FMK_OK;
}

// The fill call of a sizing call that took the one-pass kernel: the staged rows -> CSR rows + per-bar features, the listed bars by
// the class kernels.  *handled = 0: no state for these close indices (the caller runs the ordinary fill).
int fmk_fused_fill(fmk_ctx *ctx, const double *d_price, const void *d_amount, int amount_is_f64, int64_t n, const int64_t *d_close_idx,
                   int64_t n_idx, const int8_t *d_side, double price_tick_size, const double *d_bar_lows, double imbalance_factor,
                   const int64_t *d_level_offsets, int64_t max_levels, const fmk_footprint_out *d_out, int64_t *d_n_bad_level,
                   int *handled)
{
    *handled = 0;
    FuState *st = (FuState *)ctx->fused;
    if (!st) return FMK_OK;
    if (st->ci != d_close_idx || st->nb != n_idx - 1 || amount_is_f64 || st->n_fp < 0) { fmk_fused_release(ctx); return FMK_OK; }
    *handled = 1;
    const int64_t nb = st->nb;
    FpOut o;
    memcpy(&o, d_out, sizeof(o));
    int64_t blocks = fmk_ceil_div(nb, 4);
    const int64_t cap = (int64_t)ctx->n_cu * 32;
    if (blocks > cap) blocks = cap;
    k_fu_emit<<<(unsigned)blocks, 256, 0, ctx->stream>>>(st->stg, d_bar_lows, price_tick_size, imbalance_factor, d_level_offsets, nb, o);
    int rc = FMK_OK;
    if (hipGetLastError() != hipSuccess) rc = fmk_set_error(ctx, FMK_E_HIP, "k_fu_emit: launch failed");
    if (rc == FMK_OK && st->n_fp > 0)
        rc = fmk_footprints_fill_classes(ctx, d_price, d_amount, 0, d_close_idx, nb, d_side, price_tick_size, d_bar_lows, imbalance_factor,
                                         d_level_offsets, 0, max_levels, d_out, d_n_bad_level, n, nullptr, st->fp_list);
    fmk_fused_release(ctx);                                          // (stream-ordered: the launches above are queued before the free)
    return rc;
}

#ifdef FU_TIMING
extern "C" int fmk_diag_fused_phases(fmk_ctx *ctx, int64_t *out4)
{
    unsigned long long v[4], z[4] = {0, 0, 0, 0};
    FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FMK_HIP(ctx, hipMemcpyFromSymbol(v, HIP_SYMBOL(fu_phase_cycles), sizeof v));
    FMK_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(fu_phase_cycles), z, sizeof z));
    for (int i = 0; i < 4; ++i) out4[i] = (int64_t)v[i];
    return FMK_OK;
}
#endif
// diagnostics of the last one-pass sizing call: bars handed to the class kernels (footprints), to k_bar_dir (order flow), and the
// (bar, columns) entries of the tick-order redo
extern "C" int fmk_diag_fused_last(fmk_ctx *ctx, int64_t *n_fp_list, int64_t *n_dir_list, int64_t *n_redo)
{
    *n_fp_list = ctx->h_mail[14];
    *n_dir_list = ctx->h_mail[15];
    *n_redo = ctx->h_mail[16];
    return FMK_OK;
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
extern "C" int fmk_comp_bar_directional_dev(fmk_ctx *ctx, const double *d_price, const void *d_amount,
                                            int amount_is_f64, int64_t n, const int64_t *d_close_idx,
                                            int64_t n_idx, const int8_t *d_side, const fmk_directional_out *d_out,
                                            int64_t *d_n_zero_div)
{
    if (n_idx == 1) return FMK_OK;   // zero bars (base.py:409-546 has no length check, unlike comp_bar_ohlcv)
    if (n_idx < 1) return fmk_set_error(ctx, FMK_E_ARG, "negative dimensions are not allowed");
    if (n <= 0 || !d_side || !d_out) return fmk_set_error(ctx, FMK_E_ARG, "comp_bar_directional: bad arguments");
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    FMK_TRY(bf_sync_force_redo(ctx));
    const int64_t nb = n_idx - 1;
    FlowDirOut o;
    memcpy(&o, d_out, sizeof(o));
    int64_t blocks = fmk_ceil_div(nb, 4);
    const int64_t cap = (int64_t)ctx->n_cu * 64;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    unsigned long long *redo;
    FMK_TRY(fmk_scratch(ctx, (size_t)(nb + 32) * 16, (void **)&redo));
    FMK_HIP(ctx, hipMemsetAsync(redo, 0, 8, ctx->stream));
    const unsigned rblocks = (unsigned)(blocks < 4096 ? blocks : 4096);
    // bars of more than BFW_MIN ticks: a workgroup per bar, from a list
    const bool wide_on = true;
    const int64_t skip_above = wide_on ? BFW_MIN : INT64_MAX;
    auto wide = [&]() -> int {
        if (!wide_on) return FMK_OK;
        int64_t *wl = nullptr;
        FMK_TRY(fmk_long_bar_list(ctx, d_close_idx, nb, n, BFW_MIN, nullptr, &wl));
        if (amount_is_f64)
            k_bar_dir_wide<true><<<(unsigned)(ctx->n_cu * 2), 64 * BFW_WAVES, 0, ctx->stream>>>(
                d_price, d_amount, d_side, d_close_idx, wl, n, o, (unsigned long long *)d_n_zero_div, redo);
        else
            k_bar_dir_wide<false><<<(unsigned)(ctx->n_cu * 2), 64 * BFW_WAVES, 0, ctx->stream>>>(
                d_price, d_amount, d_side, d_close_idx, wl, n, o, (unsigned long long *)d_n_zero_div, redo);
        const hipError_t le = hipGetLastError();
        FMK_TRY(fmk_free(ctx, wl));
        FMK_HIP(ctx, le);
        return FMK_OK;
    };
    // One lane per bar when there are enough moderate bars to fill the chip with 64-bar waves (developer knob
    // FMK_DIR_LANES: 0 never, 2 whenever the layout allows it); float32 amounts, 8- / 4-byte aligned amount / side columns.
    const char *lv = getenv("FMK_DIR_LANES");
    const int lanes_mode = lv ? atoi(lv) : 1;
    const bool lanes_ok = !amount_is_f64 && ((uintptr_t)d_amount & 7) == 0 && ((uintptr_t)d_side & 3) == 0;
    // measured at 1e9 ticks (profiles/r02_dir_lanes.txt): 20-tick bars 66.5 -> 5.5 ms, 200-tick bars 7.6 -> 3.0 ms, 1 200-tick bars
    // 3.1-3.2 -> 2.98 ms (the lane schedule's loads without bound checks and its conditional sums as fma(1.0 or 0.0, y, x) put it
    // ahead there too; before: 3.53), 12 000-tick bars: wave per bar
    const bool lanes_fit = nb >= (int64_t)ctx->n_cu * 64 * 4 && n / nb <= 2048;
    if (lanes_ok && lanes_mode != 0 && (lanes_fit || lanes_mode == 2)) {
        unsigned long long *long_list = redo + nb + 32;
        FMK_HIP(ctx, hipMemsetAsync(long_list, 0, 8, ctx->stream));
        int64_t lblocks = fmk_ceil_div(fmk_ceil_div(nb, 64), DL_WAVES);
        const int64_t lcap = (int64_t)ctx->n_cu * 40;
        if (lblocks > lcap) lblocks = lcap;
        k_bar_dir_lanes<false><<<(unsigned)lblocks, 64 * DL_WAVES, 0, ctx->stream>>>(d_price, (const float *)d_amount, d_side,
                                                                                    d_close_idx, nb, n, o,
                                                                                    (unsigned long long *)d_n_zero_div, long_list,
                                                                                    8192, DlOhlcOut{});
        FMK_LAUNCH_CHECK(ctx);
        k_bar_dir<false><<<(unsigned)(blocks < 2048 ? blocks : 2048), 256, 0, ctx->stream>>>(
            d_price, d_amount, d_side, d_close_idx, nb, n, o, (unsigned long long *)d_n_zero_div, redo, long_list, skip_above);
        FMK_LAUNCH_CHECK(ctx);
        FMK_TRY(wide());
        bf_redo_launch<false>(ctx, rblocks, d_price, d_amount, d_side, d_close_idx, n, o, redo);
        FMK_LAUNCH_CHECK(ctx);
        return FMK_OK;
    }
    if (amount_is_f64) {
        k_bar_dir<true><<<(unsigned)blocks, 256, 0, ctx->stream>>>(d_price, d_amount, d_side, d_close_idx, nb, n, o,
                                                                 (unsigned long long *)d_n_zero_div, redo, nullptr, skip_above);
        FMK_LAUNCH_CHECK(ctx);
        FMK_TRY(wide());
        bf_redo_launch<true>(ctx, rblocks, d_price, d_amount, d_side, d_close_idx, n, o, redo);
    } else {
        const int dwpb = 4;                // one-wave workgroups (see fp_launch, fmk_footprint.hip)
        if (dwpb == 1)
            k_bar_dir<false, 1><<<(unsigned)(blocks * 4), 64, 0, ctx->stream>>>(d_price, d_amount, d_side, d_close_idx, nb, n, o,
                                                                              (unsigned long long *)d_n_zero_div, redo, nullptr, skip_above);
        else
        k_bar_dir<false><<<(unsigned)blocks, 256, 0, ctx->stream>>>(d_price, d_amount, d_side, d_close_idx, nb, n, o,
                                                                  (unsigned long long *)d_n_zero_div, redo, nullptr, skip_above);
        FMK_LAUNCH_CHECK(ctx);
        FMK_TRY(wide());
        bf_redo_launch<false>(ctx, rblocks, d_price, d_amount, d_side, d_close_idx, n, o, redo);
    }
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}

// cfg 4 in two passes over the ticks (26 B/tick): this call = OHLCV (+ median) + order-flow features from ONE read, then the
// CSR level counts; fmk_comp_bar_footprints_fill_dev is the second pass.  float64 amounts take the separate kernels.
static int bars_flow_size(fmk_ctx *ctx, const double *d_price, const void *d_amount, int amount_is_f64, int64_t n,
                          const int64_t *d_close_idx, int64_t n_idx, const int8_t *d_side, double price_tick_size,
                          double *d_open, double *d_high, double *d_low, double *d_close, float *d_volume,
                          double *d_vwap, int64_t *d_trades, double *d_median, const fmk_directional_out *d_dir,
                          int64_t *d_n_zero_div, int64_t *d_level_offsets, int64_t *total_levels,
                          int64_t *max_levels, int *median_deferred);

extern "C" int fmk_bars_flow_size_dev(fmk_ctx *ctx, const double *d_price, const void *d_amount, int amount_is_f64, int64_t n,
                                      const int64_t *d_close_idx, int64_t n_idx, const int8_t *d_side, double price_tick_size,
                                      double *d_open, double *d_high, double *d_low, double *d_close, float *d_volume,
                                      double *d_vwap, int64_t *d_trades, double *d_median, const fmk_directional_out *d_dir,
                                      int64_t *d_n_zero_div, int64_t *d_level_offsets, int64_t *total_levels,
                                      int64_t *max_levels)
{
    return bars_flow_size(ctx, d_price, d_amount, amount_is_f64, n, d_close_idx, n_idx, d_side, price_tick_size, d_open, d_high,
                          d_low, d_close, d_volume, d_vwap, d_trades, d_median, d_dir, d_n_zero_div, d_level_offsets,
                          total_levels, max_levels, nullptr);
}

// The same pass, but the median trade size may be LEFT to the footprint sweep (cfg 4 at 26 B/tick): on streams of 600..2048-tick
// bars with float32 amounts nothing is computed for d_median here and *median_deferred comes back 1 -- the caller then hands
// d_median to fmk_comp_bar_footprints_fill_median_dev, whose waves hold every amount of their bar anyway.  Otherwise the
// median is complete on return (*median_deferred = 0) and the plain fill call follows.
extern "C" int fmk_bars_flow_size_defer_dev(fmk_ctx *ctx, const double *d_price, const void *d_amount, int amount_is_f64,
                                            int64_t n, const int64_t *d_close_idx, int64_t n_idx, const int8_t *d_side,
                                            double price_tick_size, double *d_open, double *d_high, double *d_low,
                                            double *d_close, float *d_volume, double *d_vwap, int64_t *d_trades,
                                            double *d_median, const fmk_directional_out *d_dir, int64_t *d_n_zero_div,
                                            int64_t *d_level_offsets, int64_t *total_levels, int64_t *max_levels,
                                            int *median_deferred)
{
    if (!median_deferred) return fmk_set_error(ctx, FMK_E_ARG, "bars_flow: median_deferred must not be NULL");
    *median_deferred = 0;
    return bars_flow_size(ctx, d_price, d_amount, amount_is_f64, n, d_close_idx, n_idx, d_side, price_tick_size, d_open, d_high,
                          d_low, d_close, d_volume, d_vwap, d_trades, d_median, d_dir, d_n_zero_div, d_level_offsets,
                          total_levels, max_levels, median_deferred);
}

static int bars_flow_size(fmk_ctx *ctx, const double *d_price, const void *d_amount, int amount_is_f64, int64_t n,
                          const int64_t *d_close_idx, int64_t n_idx, const int8_t *d_side, double price_tick_size,
                          double *d_open, double *d_high, double *d_low, double *d_close, float *d_volume,
                          double *d_vwap, int64_t *d_trades, double *d_median, const fmk_directional_out *d_dir,
                          int64_t *d_n_zero_div, int64_t *d_level_offsets, int64_t *total_levels,
                          int64_t *max_levels, int *median_deferred)
{
    if (n_idx < 2) return fmk_set_error(ctx, FMK_E_ARG, "Bar close indices must contain at least two elements.");
    if (n <= 0 || !d_side || !d_dir) return fmk_set_error(ctx, FMK_E_ARG, "bars_flow: bad arguments");
    if (!(price_tick_size > 0)) return fmk_set_error(ctx, FMK_E_ARG, "price_tick_size must be > 0");
    // short bars: the fused kernel is a wave-per-bar schedule; comp_bar_ohlcv and the order-flow features each have a
    // several-bars-per-wave schedule of their own (k_bar_ohlcv_lanes, k_bar_dir_lanes)
    const bool short_bars = n / (n_idx - 1) < 600;
    const char *flv = getenv("FMK_FLOW_LANES");
    const int flow_lanes = flv ? atoi(flv) : 1;
    // long bars (hourly, daily): comp_bar_ohlcv and the order-flow features each have workgroup-per-bar schedules of their own
    // (fmk_ohlcv.hip: k_bar_ohlcv_mid / _wide, here: k_bar_dir_wide); the fused wave-per-bar kernel below is for the middle
    const bool long_bars = n / (n_idx - 1) > 8192;
    // the common class in ONE pass over the ticks (fmk_fused.h; round 6): 13 B/tick read once for all three families
    {
        int fused_mode = 0;
        FMK_TRY(bars_flow_fused_ok(ctx, d_amount, amount_is_f64, n, d_close_idx, n_idx - 1, &fused_mode));
        if (fused_mode)
            return bars_flow_fused(ctx, d_price, (const float *)d_amount, n, d_close_idx, n_idx, d_side, price_tick_size, d_open, d_high,
                                   d_low, d_close, d_volume, d_vwap, d_trades, d_median, d_dir, d_n_zero_div, d_level_offsets,
                                   total_levels, max_levels, fused_mode == 1);
        fmk_fused_release(ctx);                                          // (a stale state of an earlier call)
    }
    if (amount_is_f64 || short_bars || long_bars) {
        // Long bars (hourly, daily), float32 sizes: comp_bar_ohlcv and the order-flow features share nothing but the input columns -- the
        // first runs on the context's auxiliary stream beside the second (cfg 4 at hourly / daily bars 12.1 / 16.0 -> 11.0 / 13.9 ms per
        // 1e9 ticks; streams of short bars gain nothing and keep the plain order).  While both streams carry launches of this call no
        // freed block goes back to the allocator's free list (fmk_pool_defer).
        if (long_bars && !amount_is_f64) {
            FMK_TRY(fmk_ctx_aux(ctx));
            FMK_HIP(ctx, hipEventRecord(ctx->aev[0], ctx->stream));
            FMK_HIP(ctx, hipStreamWaitEvent(ctx->aux, ctx->aev[0], 0));
            (void)fmk_pool_defer(ctx, 1);
            hipStream_t keep = ctx->stream;
            ctx->stream = ctx->aux;
            int rc = fmk_comp_bar_ohlcv_dev(ctx, d_price, d_amount, amount_is_f64, n, d_close_idx, n_idx, d_open, d_high, d_low,
                                            d_close, d_volume, d_vwap, d_trades, d_median);
            ctx->stream = keep;
            if (rc == FMK_OK && hipEventRecord(ctx->aev[1], ctx->aux) != hipSuccess) rc = fmk_set_error(ctx, FMK_E_HIP, "hipEventRecord");
            if (rc == FMK_OK)
                rc = fmk_comp_bar_directional_dev(ctx, d_price, d_amount, amount_is_f64, n, d_close_idx, n_idx, d_side, d_dir, d_n_zero_div);
            if (rc == FMK_OK && hipStreamWaitEvent(ctx->stream, ctx->aev[1], 0) != hipSuccess) rc = fmk_set_error(ctx, FMK_E_HIP, "hipStreamWaitEvent");
            if (rc != FMK_OK) (void)hipStreamSynchronize(ctx->aux);             // (nothing of this call may outlive its error return)
            const int rf = fmk_pool_defer(ctx, 0);
            FMK_TRY(rc);
            FMK_TRY(rf);
        } else {
        FMK_TRY(fmk_comp_bar_ohlcv_dev(ctx, d_price, d_amount, amount_is_f64, n, d_close_idx, n_idx, d_open, d_high, d_low,
                                       d_close, d_volume, d_vwap, d_trades, d_median));
        FMK_TRY(fmk_comp_bar_directional_dev(ctx, d_price, d_amount, amount_is_f64, n, d_close_idx, n_idx, d_side, d_dir,
                                             d_n_zero_div));
        }
    } else if (flow_lanes != 0 && ((uintptr_t)d_amount & 7) == 0 && ((uintptr_t)d_side & 3) == 0 &&
               (flow_lanes == 2 || (n_idx - 1 >= (int64_t)ctx->n_cu * 64 * 4 && n / (n_idx - 1) <= 2048))) {
        // Streams of many 600..2048-tick bars (1-minute bars): ONE lane per bar walks price / amount / side for the order-flow
        // features AND open / high / low / close / volume / vwap (k_bar_dir_lanes<true>: 13 B/tick), the median trade size comes
        // from the amounts alone (fmk_median_small_launch: 4 B/tick).  Measured against the fused wave-per-bar kernel below
        // (k_bar_ohlcv_dir, 13 B/tick, VALU-bound): profiles/r02_cfg4.txt.  Developer knob FMK_FLOW_LANES: 0 never, 2 whenever the layout allows.
        FMK_HIP(ctx, hipSetDevice(ctx->device));
        const int64_t nb = n_idx - 1;
        FlowDirOut o;
        memcpy(&o, d_dir, sizeof(o));
        // Bars of UNEQUAL length (real one-minute bars: lognormal, sigma ~1; round 4): a census of the lengths by quarter-octave class
        // first -- when fewer than 80 % of the bars lie within a factor 2.4 of each other, the lanes take the bars in order of their
        // length (bf_sort_bars), so that a wave's 64 lanes finish together instead of waiting for its longest bar.  FMK_FLOW_SORT:
        // 0 never, 1 always, unset: by the census.  (Before the scratch below is taken: the sort's scan uses the context scratch.)
        const char *sv = getenv("FMK_FLOW_SORT");
        int sort_mode = sv ? atoi(sv) : -1;
        int64_t *perm = nullptr;
        if (sort_mode != 0 && d_median) {
            int64_t *hist = nullptr;
            bool uneven = false;
            FMK_TRY(bf_bar_census(ctx, d_close_idx, nb, &hist, &uneven));
            if (sort_mode < 0) sort_mode = uneven ? 1 : 0;
            int rc = sort_mode ? bf_sort_bars(ctx, d_close_idx, nb, hist, &perm) : FMK_OK;
            (void)fmk_free(ctx, hist);
            FMK_TRY(rc);
        } else sort_mode = 0;
        unsigned long long *redo;
        FMK_TRY(fmk_scratch(ctx, (size_t)(nb + 32) * 16, (void **)&redo));
        unsigned long long *long_list = redo + nb + 32;
        int *any_long = (int *)(ctx->d_mail + 20);
        FMK_HIP(ctx, hipMemsetAsync(redo, 0, 8, ctx->stream));
        FMK_HIP(ctx, hipMemsetAsync(long_list, 0, 8, ctx->stream));
        FMK_HIP(ctx, hipMemsetAsync(any_long, 0, sizeof(int), ctx->stream));
        int64_t lblocks = fmk_ceil_div(fmk_ceil_div(nb, 64), DL_WAVES);
        const int64_t lcap = (int64_t)ctx->n_cu * 40;
        if (lblocks > lcap) lblocks = lcap;
        // Sorted bars with the median (round 4, late): the medians and the open .. trades of the bars beyond 1 344 ticks come from comp_bar_ohlcv's
        // size classes -- kernels that share nothing with the order-flow kernels but the input columns.  They run on the context's auxiliary
        // stream BESIDE the lane kernel, the long-bar order flow and the redo (the lane kernel then leaves open .. trades of those bars alone:
        // two writers of one value in two association orders would race); the footprint sizing waits for both.
        const bool side_ohlcv = sort_mode && d_median;
        DlOhlcOut oo{d_open, d_high, d_low, d_close, d_volume, d_vwap, d_trades, any_long};
        // (bars of about equal length -- no sort: the lane kernel writes open .. trades of every bar it serves, and only the MEDIANS, an
        //  amounts-only pass, run beside it)
        const char *dv0 = getenv("FMK_FLOW_MEDIAN_DEFER");
        const bool side_median = !sort_mode && d_median && !(median_deferred && dv0 && atoi(dv0));
        if (side_ohlcv || side_median) {
            if (side_ohlcv) oo.ohlc_max = 1344;                                  // 64 * FMK_SMALL_NCH: the reach of k_bar_median_small's class
            FMK_TRY(fmk_ctx_aux(ctx));
            FMK_HIP(ctx, hipEventRecord(ctx->aev[0], ctx->stream));              // (the close indices may come from a launch still in flight)
            FMK_HIP(ctx, hipStreamWaitEvent(ctx->aux, ctx->aev[0], 0));
            (void)fmk_pool_defer(ctx, 1);                                        // until the streams are joined: no freed block changes sides
        }
        struct DeferGuard {                                                      // (every return below passes here)
            fmk_ctx *c; bool on;
            ~DeferGuard() { if (on) { (void)hipStreamSynchronize(c->aux); (void)fmk_pool_defer(c, 0); } }
        } defer_guard{ctx, side_ohlcv || side_median};
        if (side_median) {
            hipStream_t keep = ctx->stream;
            ctx->stream = ctx->aux;
            const int rc = fmk_median_small_launch(ctx, (const float *)d_amount, d_close_idx, nb, d_median, n);
            ctx->stream = keep;
            FMK_TRY(rc);
            FMK_HIP(ctx, hipEventRecord(ctx->aev[1], ctx->aux));
        }
        k_bar_dir_lanes<true><<<(unsigned)lblocks, 64 * DL_WAVES, 0, ctx->stream>>>(d_price, (const float *)d_amount, d_side,
                                                                                   d_close_idx, nb, n, o,
                                                                                   (unsigned long long *)d_n_zero_div, long_list,
                                                                                   8192, oo, perm);
        {
            const hipError_t le = hipGetLastError();
            if (perm && !side_ohlcv) { (void)fmk_free(ctx, perm); perm = nullptr; }   // (side_ohlcv: freed behind the auxiliary stream's allocations)
            if (le != hipSuccess && perm) { (void)fmk_free(ctx, perm); perm = nullptr; }
            FMK_HIP(ctx, le);
        }
        int64_t blocks = fmk_ceil_div(nb, 4);
        if (blocks > 2048) blocks = 2048;
        const int dwpb = 4;
        if (dwpb == 1)
            k_bar_dir<false, 1><<<(unsigned)(blocks * 4), 64, 0, ctx->stream>>>(d_price, d_amount, d_side, d_close_idx, nb, n, o,
                                                                              (unsigned long long *)d_n_zero_div, redo, long_list);
        else
        k_bar_dir<false><<<(unsigned)blocks, 256, 0, ctx->stream>>>(d_price, d_amount, d_side, d_close_idx, nb, n, o,
                                                                   (unsigned long long *)d_n_zero_div, redo, long_list);
        bf_redo_launch<false>(ctx, (unsigned)blocks, d_price, d_amount, d_side, d_close_idx, n, o, redo);
        {
            const hipError_t le = hipGetLastError();
            if (le != hipSuccess && perm) { (void)fmk_free(ctx, perm); perm = nullptr; }
            FMK_HIP(ctx, le);
        }
        if (side_ohlcv) {
            hipStream_t keep = ctx->stream;
            ctx->stream = ctx->aux;                                              // every launch, wait and allocation of the call below: the auxiliary stream
            int rc = fmk_median_small_ohlcv_long_launch(ctx, d_price, (const float *)d_amount, d_close_idx, nb, n, d_open, d_high, d_low,
                                                        d_close, d_volume, d_vwap, d_trades, d_median);
            ctx->stream = keep;
            if (perm) { (void)fmk_free(ctx, perm); perm = nullptr; }
            FMK_TRY(rc);
            FMK_HIP(ctx, hipEventRecord(ctx->aev[1], ctx->aux));
            FMK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->aev[1], 0));
            return fmk_comp_bar_footprints_size_dev(ctx, d_low, d_high, n_idx - 1, price_tick_size, d_level_offsets, total_levels,
                                                    max_levels);
        }
        // How many bars did the lane kernel hand on?  On a stream of about equally long bars none, and their OHLC is done.  On real
        // one-minute bars (lognormal lengths) most waves keep only their short bars (k_bar_dir_lanes: the 70 % rule): the bars that
        // carry most of the ticks are then better served by comp_bar_ohlcv's own size classes (one pass, median included) than by
        // the generic leftover kernel plus the stand-alone median kernels -- measured at sigma 1: 20.0 ms for this call against
        // 14.3 ms for the three functions apart (tools/realcfg4.py).  One 8-byte read-back; the call waits for its sizing pass anyway.
        FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[13], long_list, 8, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (sort_mode && d_median) {
            // bars in order of length: the lanes have served every bar up to 8 192 ticks; the longer ones (listed) got their order flow from
            // k_bar_dir above; medians (and open .. trades of the bars beyond 1 344 ticks) by comp_bar_ohlcv's size classes
            FMK_TRY(fmk_median_small_ohlcv_long_launch(ctx, d_price, (const float *)d_amount, d_close_idx, nb, n, d_open, d_high, d_low,
                                                       d_close, d_volume, d_vwap, d_trades, d_median));
            return fmk_comp_bar_footprints_size_dev(ctx, d_low, d_high, n_idx - 1, price_tick_size, d_level_offsets, total_levels,
                                                    max_levels);
        }
        if (side_median) FMK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->aev[1], 0));    // the medians of the auxiliary stream
        if (ctx->h_mail[13] > nb / 50) {
            FMK_TRY(fmk_comp_bar_ohlcv_dev(ctx, d_price, d_amount, 0, n, d_close_idx, n_idx, d_open, d_high, d_low, d_close, d_volume,
                                           d_vwap, d_trades, d_median));
            return fmk_comp_bar_footprints_size_dev(ctx, d_low, d_high, n_idx - 1, price_tick_size, d_level_offsets, total_levels,
                                                    max_levels);
        }
        FMK_TRY(fmk_ohlcv_leftover_launch(ctx, d_price, d_amount, 0, d_close_idx, nb, n, 8192, any_long, d_open, d_high, d_low,
                                          d_close, d_volume, d_vwap, d_trades));
        // FMK_FLOW_MEDIAN_DEFER=1 leaves the median to the footprint sweep (26 B/tick in all).  OFF by default: measured slower --
        // the sweep is latency-bound, the bracket bookkeeping and the per-bar selection add ~46 VALU instructions per 64 ticks
        // and cost it two of its seven waves per SIMD: 9.35 / 10.4 ms against 7.96 / 9.0 ms with the amounts-only median pass
        // (profiles/r03_cfg4.txt).  The path stays (exact, tested: tests/test_gpu_fused.py) for streams where it may pay.
        const char *dv = getenv("FMK_FLOW_MEDIAN_DEFER");
        const int defer_ok = dv ? atoi(dv) : 0;
        if (d_median && median_deferred && defer_ok) *median_deferred = 1;
        else if (d_median && !side_median) FMK_TRY(fmk_median_small_launch(ctx, (const float *)d_amount, d_close_idx, nb, d_median, n));
    } else {
        FMK_HIP(ctx, hipSetDevice(ctx->device));
        const int64_t nb = n_idx - 1;
        FlowDirOut o;
        memcpy(&o, d_dir, sizeof(o));
        FlowOhlcvOut oo{d_open, d_high, d_low, d_close, d_volume, d_vwap, d_trades, d_median};
        int64_t blocks = fmk_ceil_div(nb, 4);
        const int64_t cap = (int64_t)ctx->n_cu * 64;
        if (blocks > cap) blocks = cap;
        if (blocks < 1) blocks = 1;
        unsigned long long *redo;
        FMK_TRY(fmk_scratch(ctx, (size_t)(nb + 32) * 8, (void **)&redo));
        FMK_HIP(ctx, hipMemsetAsync(redo, 0, 8, ctx->stream));
        int *saw_long = (int *)(ctx->d_mail + 16);
        FMK_HIP(ctx, hipMemsetAsync(saw_long, 0, sizeof(int), ctx->stream));
        if (d_median)
            k_bar_ohlcv_dir<true><<<(unsigned)blocks, 256, 0, ctx->stream>>>(d_price, (const float *)d_amount, d_side, d_close_idx,
                                                                           nb, n, o, (unsigned long long *)d_n_zero_div, redo, oo,
                                                                           saw_long);
        else
            k_bar_ohlcv_dir<false><<<(unsigned)blocks, 256, 0, ctx->stream>>>(d_price, (const float *)d_amount, d_side, d_close_idx,
                                                                            nb, n, o, (unsigned long long *)d_n_zero_div, redo, oo,
                                                                            saw_long);
        FMK_LAUNCH_CHECK(ctx);
        const unsigned rblocks = (unsigned)(blocks < 4096 ? blocks : 4096);
        bf_redo_launch<false>(ctx, rblocks, d_price, d_amount, d_side, d_close_idx, n, o, redo);
        FMK_LAUNCH_CHECK(ctx);
        if (d_median) FMK_TRY(fmk_median_launch(ctx, d_amount, 0, d_close_idx, nb, (int64_t)BF_MED_TILES * 512, saw_long, d_median, n));
    }
    return fmk_comp_bar_footprints_size_dev(ctx, d_low, d_high, n_idx - 1, price_tick_size, d_level_offsets, total_levels,
                                            max_levels);
}

// (In-kernel fusion of the order-flow and footprint halves was built and measured twice in round 1 -- one wave doing both halves on a
// shared LDS tile: 11.2 ms; two waves per bar, one per half, on double-buffered tiles: 10.3 ms at 1e9 ticks -- against 3.0 + 3.7 ms
// for the two kernels: both halves were VALU-bound, not HBM-bound.  Round 6's one-pass kernel, fmk_fused.h, is the form that pays.)
// diagnostics since the last call: {(bar, column) pairs redone in tick order, 512-term tiles walked, tiles added term by term,
// pairs of row 0 .. 6 (buy / sell volume, buy / sell dollars, spread, signed volume, signed dollars)}
extern "C" int fmk_diag_dir_redo(fmk_ctx *ctx, int64_t *out10)
{
    unsigned long long v[10], z[10];
    memset(v, 0, sizeof v);
    memset(z, 0, sizeof z);
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FMK_HIP(ctx, hipMemcpyFromSymbol(v, HIP_SYMBOL(bf_redo_stats), sizeof v));
    FMK_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(bf_redo_stats), z, sizeof z));
    for (int i = 0; i < 10; ++i) out10[i] = (int64_t)v[i];
    return FMK_OK;
}
