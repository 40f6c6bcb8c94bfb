// fmk_directional.hip -- comp_bar_directional_features (finmlkit/bar/base.py:409-546) on gfx950.
//
// One wave owns one bar and streams it in 64-tick chunks (price 512 B + amount 256 B + side 64 B
// per load instruction, coalesced): 13 B/tick (f32 amounts) + 88 B/bar written.
//   * buy/sell tick counts: ballot popcounts (SGPR), exact.
//   * buy/sell volume and dollar sums, cumulative spread: lane-strided float64 partial sums,
//     xor-butterfly at the end of the bar, rounded once to float32 like the reference.
//   * "spread" needs the previous tick's price and side: lane-1 via a shuffle, lane 0 from the
//     previous chunk's lane 63 (first chunk: tick start-1 with the reference's Python negative-
//     index wrap, base.py:485-500).
//   * min/max of the running signed tick/volume/dollar imbalance: 6-step inclusive wave scan of
//     the signed contributions per chunk on the DPP path (fmk_dpp.h, no LDS round trips) + wave-uniform carry; every lane tracks min/max of its
//     own prefix values, folded at the end of the bar.
// The float64 prefix values are tree-summed (not the reference's sequential order): the float32
// outputs are identical except for a ~1e-8 chance of a 1-ulp flip (see tests/_golden.py).
#include <math.h>

#include "fmk_common.h"
#include "fmk_dpp.h"

struct DirOut {
    int64_t *ticks_buy, *ticks_sell;
    float *volume_buy, *volume_sell, *dollars_buy, *dollars_sell;
    float *mean_spread, *max_spread;
    int64_t *cum_ticks_min, *cum_ticks_max;
    float *cum_volumes_min, *cum_volumes_max, *cum_dollars_min, *cum_dollars_max;
};
static_assert(sizeof(DirOut) == sizeof(fmk_directional_out), "ABI struct mismatch");

template <bool AF64>
__global__ __launch_bounds__(256) void k_bar_directional(const double *__restrict__ price,
                                                         const void *__restrict__ amount,
                                                         const int8_t *__restrict__ side,
                                                         const int64_t *__restrict__ ci, int64_t nb, int64_t n,
                                                         DirOut o, unsigned long long *n_zero_div)
{
    const int lane = fmk_lane();
    const int wpb = blockDim.x >> 6;
    const int64_t wave0 = (int64_t)blockIdx.x * wpb + fmk_uniform((int)(threadIdx.x >> 6));
    const int64_t nwaves = (int64_t)gridDim.x * wpb;
    for (int64_t b = wave0; b < nb; b += nwaves) {
        const int64_t s = fmk_uniform(ci[b]);
        const int64_t e = fmk_uniform(ci[b + 1]);
        const int64_t start = s + 1;
        const int64_t cnt = e - s;
        // per-lane accumulators
        double vb = 0, vs = 0, db = 0, ds = 0, cs = 0, mxs = 0;
        int64_t tmin = 1000000000LL, tmax = -1000000000LL;      // base.py:459-460
        double vmin = 1e9, vmax = -1e9, dmin = 1e9, dmax = -1e9;
        // wave-uniform state
        int64_t tb = 0, tsell = 0, carry_t = 0;
        double carry_v = 0, carry_d = 0;
        int prev_side = 0;
        double prev_price = 0;
        if (cnt > 0) {
            prev_price = price[fmk_wrap(start - 1, n)];
            prev_side = cnt > 1 ? (int)side[fmk_wrap(start - 1, n)] : 0;    // base.py:485-488
        }
        // software pipeline: the loads of chunk c+1 are in flight while chunk c is processed
        double p_n = 0, a_n = 0;
        int sd_n = 0;
        if (start + lane <= e) { p_n = price[start + lane]; a_n = fmk_amt<AF64>(amount, start + lane); sd_n = side[start + lane]; }
        for (int64_t j0 = start; j0 <= e; j0 += 64) {
            const int64_t j = j0 + lane;
            const bool valid = j <= e;
            const double p = p_n, a = a_n;
            const int sd = sd_n;
            p_n = 0; a_n = 0; sd_n = 0;
            if (j + 64 <= e) { p_n = price[j + 64]; a_n = fmk_amt<AF64>(amount, j + 64); sd_n = side[j + 64]; }
            const double pp = fmk_dpp_shift_up1(p, prev_price);      // lane 0: last tick of the previous chunk
            const int ps = fmk_dpp_shift_up1(sd, prev_side);
            if (valid && sd != ps) {                           // base.py:495-500
                double sp = fabs(p - pp);
                mxs = fmax(mxs, sp);
                cs += sp;
            }
            prev_price = fmk_last_lane(p);
            prev_side = fmk_last_lane(sd);
            const bool buy = valid && sd == 1, sell = valid && sd == -1;
            const double pv = p * a;
            if (buy) { vb += a; db += pv; }
            if (sell) { vs += a; ds += pv; }
            tb += __popcll(__ballot(buy));
            tsell += __popcll(__ballot(sell));
            const int t = (int)buy - (int)sell;
            const double sv = buy ? a : (sell ? -a : 0.0);
            const double sdol = buy ? pv : (sell ? -pv : 0.0);
            const int it = fmk_dpp_iscan(t, 0, FmkOpAdd());
            const double iv = fmk_dpp_iscan(sv, 0.0, FmkOpAdd());
            const double id = fmk_dpp_iscan(sdol, 0.0, FmkOpAdd());
            if (t != 0) {                                       // base.py:518-527: signed ticks only
                const int64_t ct = carry_t + it;
                const double cv = carry_v + iv, cd = carry_d + id;
                tmin = ct < tmin ? ct : tmin; tmax = ct > tmax ? ct : tmax;
                vmin = fmin(vmin, cv); vmax = fmax(vmax, cv);
                dmin = fmin(dmin, cd); dmax = fmax(dmax, cd);
            }
            carry_t += fmk_last_lane(it);
            carry_v += fmk_last_lane(iv);
            carry_d += fmk_last_lane(id);
        }
        vb = fmk_dpp_reduce(vb, 0.0, FmkOpAdd()); vs = fmk_dpp_reduce(vs, 0.0, FmkOpAdd());
        db = fmk_dpp_reduce(db, 0.0, FmkOpAdd()); ds = fmk_dpp_reduce(ds, 0.0, FmkOpAdd());
        cs = fmk_dpp_reduce(cs, 0.0, FmkOpAdd()); mxs = fmk_dpp_reduce(mxs, 0.0, FmkOpMax());
        tmin = fmk_dpp_reduce(tmin, (int64_t)1000000000LL, FmkOpMin());
        tmax = fmk_dpp_reduce(tmax, (int64_t)-1000000000LL, FmkOpMax());
        vmin = fmk_dpp_reduce(vmin, 1e9, FmkOpMin()); vmax = fmk_dpp_reduce(vmax, -1e9, FmkOpMax());
        dmin = fmk_dpp_reduce(dmin, 1e9, FmkOpMin()); dmax = fmk_dpp_reduce(dmax, -1e9, FmkOpMax());
        if (lane == 0) {
            o.ticks_buy[b] = tb; o.ticks_sell[b] = tsell;
            o.volume_buy[b] = (float)vb; o.volume_sell[b] = (float)vs;
            o.dollars_buy[b] = (float)db; o.dollars_sell[b] = (float)ds;
            o.max_spread[b] = (float)mxs;
            if (tb + tsell == 0) {       // reference: ZeroDivisionError (base.py:536)
                o.mean_spread[b] = NAN;
                if (n_zero_div) atomicAdd(n_zero_div, 1ULL);
            } else {
                o.mean_spread[b] = (float)(cs / (double)(tb + tsell));
            }
            o.cum_ticks_min[b] = tmin; o.cum_ticks_max[b] = tmax;
            o.cum_volumes_min[b] = (float)vmin; o.cum_volumes_max[b] = (float)vmax;
            o.cum_dollars_min[b] = (float)dmin; o.cum_dollars_max[b] = (float)dmax;
        }
    }
}

extern "C" int fmk_comp_bar_directional_dev(fmk_ctx *ctx, const double *d_price, const void *d_amount,
                                            int amount_is_f64, int64_t n, const int64_t *d_close_idx,
                                            int64_t n_idx, const int8_t *d_side, const fmk_directional_out *d_out,
                                            int64_t *d_n_zero_div)
{
    if (n_idx < 2)
        return fmk_set_error(ctx, FMK_E_ARG, "Bar close indices must contain at least two elements.");
    if (n <= 0 || !d_side || !d_out) return fmk_set_error(ctx, FMK_E_ARG, "comp_bar_directional: bad arguments");
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t nb = n_idx - 1;
    int64_t blocks = fmk_ceil_div(nb, 4);
    const int64_t cap = (int64_t)ctx->n_cu * 64;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    DirOut o;
    memcpy(&o, d_out, sizeof(o));
    if (amount_is_f64)
        k_bar_directional<true><<<(unsigned)blocks, 256, 0, ctx->stream>>>(
            d_price, d_amount, d_side, d_close_idx, nb, n, o, (unsigned long long *)d_n_zero_div);
    else
        k_bar_directional<false><<<(unsigned)blocks, 256, 0, ctx->stream>>>(
            d_price, d_amount, d_side, d_close_idx, nb, n, o, (unsigned long long *)d_n_zero_div);
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}
