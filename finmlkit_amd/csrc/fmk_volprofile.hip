// fmk_volprofile.hip -- volume_profile_rolling (finmlkit/feature/core/volume.py:403-456) on
// gfx950: rolling-window aggregation of the CSR footprints, optional bucketing, POC / value area, share of volume
// above the POC.  Direct consumer of the footprint kernel's output (no ragged Python lists).
//
// One wave per output bar i:
//   window [s, e) of bars by two binary searches on the bar timestamps (aggregate_footprint :158-166);
//   level range from min(lows) / max(highs) of the window, rounded half-even like Python's round();
//   the window's dense level histogram lives in the wave's LDS slice; the bars of the window are added ONE AFTER THE
//   OTHER (lanes across the levels of a bar, which are distinct), i.e. every level sees its float32 adds in the
//   reference's order (:196-200);
//   bucket_price_levels (:206-275): one lane per bin adds its levels in level order (float32);
//   comp_poc_hva_lva (:278-369): NumPy-pairwise float32 total, first argmax, the value-area walk by lane 0 with
//   float64 scalars (Numba-typed semantics); calc_volume_percentage_above_poc (:372-400).
#include "fmk_footprint.h"

#define VP_MAX_LEVELS 8192                  // widest window whose histogram lives in LDS
#define VP_MAX_LEVELS_GLOBAL (1 << 24)      // wider windows: histogram in global scratch

__device__ __forceinline__ int64_t vp_lower(const int64_t *a, int64_t n, int64_t key)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = lo + (hi - lo) / 2; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}
__device__ __forceinline__ int64_t vp_upper(const int64_t *a, int64_t n, int64_t key)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = lo + (hi - lo) / 2; if (a[mid] <= key) lo = mid + 1; else hi = mid; }
    return lo;
}

// The windows of all bars, ONE LANE per bar: start / end by bisection of the bar timestamps, level range from the window's lows
// and highs -> win[4 * (i - first)] = {s, e, minl, maxl}, and the widest window (sizes the histograms).  The first version did
// this with one WAVE per bar (20 + 20 dependent loads of the two bisections with 63 lanes idle) and then once more in the main
// kernel: 12.4 ms per 8.3e5 bars with 30-bar windows.
__global__ __launch_bounds__(256) void k_vp_windows(const int64_t *__restrict__ ts, const double *__restrict__ highs,
                                                    const double *__restrict__ lows, int64_t nb, int64_t first,
                                                    int64_t window_ns, double tick, int64_t *__restrict__ win,
                                                    unsigned long long *max_levels)
{
    const int64_t i = first + (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t L = 0;
    if (i < nb) {
        const int64_t end_ts = ts[i];
        int64_t s = vp_lower(ts, nb, end_ts - window_ns);
        const int64_t e = vp_upper(ts, nb, end_ts);
        if (s == e) s = s - 1 > 0 ? s - 1 : 0;                   // volume.py:164-166
        double mn = INFINITY, mx = -INFINITY;
        for (int64_t t = s; t < e; ++t) { mn = fmin(mn, lows[t]); mx = fmax(mx, highs[t]); }
        const int64_t minl = (int64_t)rint(mn / tick), maxl = (int64_t)rint(mx / tick);   // int(round(x / price_tick)), half-even
        int64_t *w = win + 4 * (i - first);
        w[0] = s; w[1] = e; w[2] = minl; w[3] = maxl;
        L = maxl - minl + 1;
    }
    // attempted only when it would raise the value: one same-address atomic per BAR serialises at ~10 ns each (8 ms for 8e5 bars)
    L = fmk_dpp_reduce(L, (int64_t)0, FmkOpMax());
    if (fmk_lane() == 0 && L > 0 && (unsigned long long)L > __atomic_load_n(max_levels, __ATOMIC_RELAXED))
        atomicMax(max_levels, (unsigned long long)L);
}

// status bits written to *status
#define VP_BAD_LEVEL 1      // a footprint level outside its window's range
#define VP_ONE_LEVEL 2      // bucketing a window that spans a single level (the reference raises there)

template <bool GLOBAL>
__global__ __launch_bounds__(256) void k_volume_profile(const int64_t *__restrict__ ts, const double *__restrict__ highs,
                                                        const double *__restrict__ lows,
                                                        const int64_t *__restrict__ off,
                                                        const int32_t *__restrict__ levels,
                                                        const float *__restrict__ buy, const float *__restrict__ sell,
                                                        int64_t nb, int64_t first, int64_t window_ns, int64_t n_bins,
                                                        double tick, double va_pct, int cap, int32_t *__restrict__ poc,
                                                        int32_t *__restrict__ hva, int32_t *__restrict__ lva,
                                                        float *__restrict__ pct, unsigned int *status,
                                                        unsigned char *gscratch, const int64_t *__restrict__ win)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = fmk_lane();
    const int wib = fmk_uniform((int)(threadIdx.x >> 6));
    const int wpb = blockDim.x >> 6;
    const size_t per_wave = (size_t)(cap + 8) * 8 + 256;
    unsigned char *mine;                                          // LDS-typed unless the window is very wide
    if constexpr (GLOBAL) mine = gscratch + ((size_t)blockIdx.x * wpb + wib) * per_wave;
    else mine = smem + (size_t)wib * per_wave;
    float *ab = (float *)mine;                                    // [cap + 8] buy sums, later totals
    float *as = ab + (cap + 8);                                   // [cap + 8] sell sums, later binned volumes
    int *stk = (int *)(as + (cap + 8));                           // 64 ints (pairwise-sum stack)
    const int64_t nwaves = (int64_t)gridDim.x * wpb;
    for (int64_t i = first + (int64_t)blockIdx.x * wpb + wib; i < nb; i += nwaves) {
        const int64_t *wi = win + 4 * (i - first);
        const int64_t s = fmk_uniform(wi[0]), e = fmk_uniform(wi[1]), minl = fmk_uniform(wi[2]), maxl = fmk_uniform(wi[3]);
        const int64_t Ll = maxl - minl + 1;
        if (Ll < 1 || Ll > cap) { if (lane == 0) atomicOr(status, VP_BAD_LEVEL); continue; }
        const int L = (int)Ll;
        for (int k = lane; k < L; k += 64) { ab[k] = 0.f; as[k] = 0.f; }
        __builtin_amdgcn_wave_barrier();
        // ---- aggregate_footprint: bars in order, lanes across the (distinct) levels of a bar
        bool bad = false;
        // (the CSR offsets of up to 63 bars by one coalesced load, a bar's first 64 levels fetched while the previous bar is added)
        for (int64_t t0 = s; t0 < e; t0 += 63) {
            const int nbar = (int)(e - t0 < 63 ? e - t0 : 63);
            const int64_t my_off = lane <= nbar ? off[t0 + lane] : 0;
            int64_t r0 = fmk_readlane(my_off, 0), r1 = fmk_readlane(my_off, 1);
            int lv = 0;
            float bv = 0.f, sv = 0.f;
            bool has = r0 + lane < r1;
            if (has) { lv = levels[r0 + lane]; bv = buy[r0 + lane]; sv = sell[r0 + lane]; }
            for (int q = 0; q < nbar; ++q) {
                const int64_t c0 = r0, c1 = r1;
                const int clv = lv;
                const float cbv = bv, csv = sv;
                const bool chas = has;
                if (q + 1 < nbar) {                               // the next bar's first rows, in flight during the adds below
                    r0 = r1; r1 = fmk_readlane(my_off, q + 2);
                    has = r0 + lane < r1;
                    if (has) { lv = levels[r0 + lane]; bv = buy[r0 + lane]; sv = sell[r0 + lane]; }
                }
                if (chas) {
                    const int64_t idx = (int64_t)clv - minl;
                    if (idx < 0 || idx >= L) bad = true;
                    else { ab[idx] += cbv; as[idx] += csv; }      // float32 += float32, one add per level per bar
                }
                for (int64_t r = c0 + 64 + lane; r < c1; r += 64) {
                    const int64_t idx = (int64_t)levels[r] - minl;
                    if (idx < 0 || idx >= L) { bad = true; continue; }
                    ab[idx] += buy[r];
                    as[idx] += sell[r];
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (__ballot(bad) != 0 && lane == 0) atomicOr(status, VP_BAD_LEVEL);
        for (int k = lane; k < L; k += 64) ab[k] = ab[k] + as[k];  // total_volumes = buy + sell (float32)
        __builtin_amdgcn_wave_barrier();
        // ---- bucket_price_levels
        float *vol = ab;
        int np_ = L;
        int64_t bw = 1, nbk = 0;
        if (n_bins >= 0) {
            const int64_t range = maxl - minl;
            bw = range / n_bins;
            if (bw < 1) bw = 1;
            if (bw % 2 == 0) bw += 1;
            const int64_t n_edges = (range + bw + bw - 1) / bw;   // len(arange(min, max + bw, bw))
            nbk = n_edges - 1;
            if (n_edges < 2) { if (lane == 0) atomicOr(status, VP_ONE_LEVEL); continue; }
            np_ = (int)(nbk + ((range / bw >= nbk) ? 1 : 0));     // + leftover bin iff the last level falls past the bins
            for (int b = lane; b < np_; b += 64) {
                const int64_t k0 = (int64_t)b * bw;
                const int64_t k1 = (b == nbk) ? L : (k0 + bw < L ? k0 + bw : L);
                float acc = 0.f;
                for (int64_t k = k0; k < k1; ++k) acc += ab[k];   // float32 adds in level order (volume.py:262-270)
                as[b] = acc;
            }
            __builtin_amdgcn_wave_barrier();
            vol = as;
        }
        // ---- comp_poc_hva_lva
        const float total = fp_pairwise_f32(vol, np_, lane, stk);
        float best = -INFINITY;
        int best_i = 0x7FFFFFFF;
        for (int k = lane; k < np_; k += 64) {
            const float v = vol[k];
            if (v > best || (v != v && best == best)) { best = v; best_i = k; }       // (np.argmax: the first NaN is the maximum)
        }
#pragma unroll
        for (int x = 32; x > 0; x >>= 1) {
            const float ob = __shfl_xor(best, x, 64);
            const int oi = __shfl_xor(best_i, x, 64);
            const bool on = ob != ob, bn = best != best;
            const bool take = on ? (!bn || oi < best_i) : (!bn && (ob > best || (ob == best && oi < best_i)));
            if (take) { best = ob; best_i = oi; }
        }
        if (best_i == 0x7FFFFFFF) best_i = 0;                     // empty guard: np.argmax -> 0
        if (lane == 0) {
            const int n = np_;
            auto pl = [&](int k) -> int32_t {
                if (n_bins < 0) return (int32_t)(minl + k);
                if (k < nbk) return (int32_t)(minl + (int64_t)k * bw + (bw - 1) / 2);     // (e_k + e_k+1 - 1) // 2
                return (int32_t)maxl;
            };
            const int pi = best_i;
            const int32_t poc_price = pl(pi);
            const double va_thrs = (double)total * (va_pct / 100.0);
            double cum = vol[pi];
            int32_t hv = poc_price, lv = poc_price;
            int up = pi + 1, down = pi - 1;
            double cu = 0.0, cd = 0.0;
            if (up < n) { cu = vol[up]; if (up + 1 < n) cu += vol[up + 1]; }
            if (down >= 0) { cd = vol[down]; if (down - 1 >= 0) cd += vol[down - 1]; }
            while (cum < va_thrs) {
                if (cu > cd) {
                    cum += cu;
                    hv = pl(up + 1 < n - 1 ? up + 1 : n - 1);
                    up += 2;
                    cu = -1.0;
                    if (up < n) { cu = vol[up]; if (up + 1 < n) cu += vol[up + 1]; }
                } else if (cu < cd) {
                    cum += cd;
                    lv = pl(down - 1 > 0 ? down - 1 : 0);
                    down -= 2;
                    cd = -1.0;
                    if (down >= 0) { cd = vol[down]; if (down - 1 >= 0) cd += vol[down - 1]; }
                } else if (cu == cd && cd != -1.0) {
                    cum += cu + cd;
                    hv = pl(up + 1 < n - 1 ? up + 1 : n - 1);
                    lv = pl(down - 1 > 0 ? down - 1 : 0);
                    up += 2; down -= 2;
                    cu = -1.0;
                    if (up < n) { cu = vol[up]; if (up + 1 < n) cu += vol[up + 1]; }
                    cd = -1.0;
                    if (down >= 0) { cd = vol[down]; if (down - 1 >= 0) cd += vol[down - 1]; }
                } else break;                                      // the reference's "stuck in loop" exit
            }
            float p = 0.f;
            if (!(total <= 0.f)) {                                 // calc_volume_percentage_above_poc (volume.py:378: `<= 0`, so a NaN total goes on)
                double above = 0.0;
                for (int k = 0; k < n; ++k) if (pl(k) > poc_price) above += (double)vol[k];
                if (!(above <= 0.0)) p = (float)(above / (double)total);
            }
            poc[i] = poc_price; hva[i] = hv; lva[i] = lv; pct[i] = p;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

extern "C" int fmk_volume_profile_rolling_dev(fmk_ctx *ctx, const int64_t *d_bar_ts, const double *d_highs,
                                              const double *d_lows, const int64_t *d_level_offsets,
                                              const int32_t *d_price_levels, const float *d_buy_volumes,
                                              const float *d_sell_volumes, int64_t n_bars, int64_t first_bar,
                                              int64_t window_ns, int64_t n_bins, double price_tick, double va_pct,
                                              int32_t *d_poc, int32_t *d_hva, int32_t *d_lva, float *d_pct)
{
    if (n_bars <= 0) return fmk_set_error(ctx, FMK_E_ARG, "Input arrays should have the same length and be non-empty.");
    if (!(price_tick > 0)) return fmk_set_error(ctx, FMK_E_ARG, "price_tick must be > 0");
    if (n_bins == 0) return fmk_set_error(ctx, FMK_E_ZERODIV, "integer division or modulo by zero");   // range // n_bins
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    FMK_HIP(ctx, hipMemsetAsync(d_poc, 0, (size_t)n_bars * 4, ctx->stream));
    FMK_HIP(ctx, hipMemsetAsync(d_hva, 0, (size_t)n_bars * 4, ctx->stream));
    FMK_HIP(ctx, hipMemsetAsync(d_lva, 0, (size_t)n_bars * 4, ctx->stream));
    FMK_HIP(ctx, hipMemsetAsync(d_pct, 0, (size_t)n_bars * 4, ctx->stream));
    if (first_bar >= n_bars) return FMK_OK;
    unsigned long long *d_max = (unsigned long long *)ctx->d_mail;
    unsigned int *d_status = (unsigned int *)(d_max + 1);
    FMK_HIP(ctx, hipMemsetAsync(d_max, 0, 16, ctx->stream));
    const int64_t work = n_bars - first_bar;
    void *win_v = nullptr;
    FMK_TRY(fmk_alloc(ctx, (size_t)work * 32, &win_v));
    int64_t *d_win = (int64_t *)win_v;
    struct WinGuard { fmk_ctx *c; void *p; ~WinGuard() { (void)fmk_free(c, p); } } win_guard{ctx, win_v};
    k_vp_windows<<<(unsigned)fmk_ceil_div(work, 256), 256, 0, ctx->stream>>>(d_bar_ts, d_highs, d_lows, n_bars, first_bar,
                                                                            window_ns, price_tick, d_win, d_max);
    FMK_LAUNCH_CHECK(ctx);
    FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[0], d_max, 8, hipMemcpyDeviceToHost, ctx->stream));
    FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t max_levels = ctx->h_mail[0];
    if (max_levels > VP_MAX_LEVELS_GLOBAL)
        return fmk_set_error(ctx, FMK_E_CAPACITY, "volume_profile_rolling: a window spans %lld price levels; this build "
                             "supports <= %d", (long long)max_levels, VP_MAX_LEVELS_GLOBAL);
    int cap = 1024, wpb = 4;
    if (max_levels > 1024) { cap = max_levels > 4096 ? 8192 : 4096; wpb = 1; }
    if (max_levels > VP_MAX_LEVELS) cap = (int)max_levels;
    size_t smem = (size_t)wpb * ((size_t)(cap + 8) * 8 + 256);
    int64_t blocks = fmk_ceil_div(work, wpb);
    int64_t capb = (int64_t)ctx->n_cu * 32;
    unsigned char *gscratch = nullptr;
    if (max_levels > VP_MAX_LEVELS) {                             // histogram in global scratch (<= 8 GB, >= 16 waves)
        const size_t per_wave = smem;
        capb = (int64_t)(((size_t)8 << 30) / per_wave);
        if (capb < 16) capb = 16;
        if (capb > (int64_t)ctx->n_cu * 8) capb = (int64_t)ctx->n_cu * 8;
        if (blocks > capb) blocks = capb;
        void *scr;
        FMK_TRY(fmk_scratch(ctx, per_wave * (size_t)blocks + 256, &scr));
        gscratch = (unsigned char *)scr;
        smem = 0;
    }
    if (smem > 64 * 1024)
        FMK_HIP(ctx, hipFuncSetAttribute((const void *)k_volume_profile<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)smem));
    if (blocks > capb) blocks = capb;
    if (gscratch)
        k_volume_profile<true><<<(unsigned)blocks, wpb * 64, 0, ctx->stream>>>(
            d_bar_ts, d_highs, d_lows, d_level_offsets, d_price_levels, d_buy_volumes, d_sell_volumes, n_bars, first_bar,
            window_ns, n_bins, price_tick, va_pct, cap, d_poc, d_hva, d_lva, d_pct, d_status, gscratch, d_win);
    else
        k_volume_profile<false><<<(unsigned)blocks, wpb * 64, smem, ctx->stream>>>(
            d_bar_ts, d_highs, d_lows, d_level_offsets, d_price_levels, d_buy_volumes, d_sell_volumes, n_bars, first_bar,
            window_ns, n_bins, price_tick, va_pct, cap, d_poc, d_hva, d_lva, d_pct, d_status, nullptr, d_win);
    FMK_LAUNCH_CHECK(ctx);
    FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[1], d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
    FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const unsigned st = (unsigned)(ctx->h_mail[1] & 0xFFFFFFFF);
    if (st & VP_BAD_LEVEL) return fmk_set_error(ctx, FMK_E_LEVEL, "volume_profile_rolling: footprint level outside its window");
    if (st & VP_ONE_LEVEL)
        return fmk_set_error(ctx, FMK_E_LEVEL, "volume_profile_rolling: a window spans a single price level; it cannot be "
                             "bucketed (the reference raises a broadcast ValueError here)");
    return FMK_OK;
}


// ---- calc_volume_percentage_above_poc (volume.py:367-391) as a stand-alone call ----------------------------------------
// One wave: NumPy-pairwise float32 total (np.sum of a float32 array), then the reference's sequential `+=` over the levels
// above the POC in float64 (Numba's type unification of `volume_above_poc = 0.0; += float32`), float64 quotient.
__global__ __launch_bounds__(64) void k_pct_above_poc(const int32_t *__restrict__ levels, const float *__restrict__ vol,
                                                      int n, int32_t poc_price, double *__restrict__ out)
{
    __shared__ __attribute__((aligned(8))) int stk[FMK_PW_STK_F32];
    const int lane = fmk_lane();
    const float total = fmk_pairwise_f32([&](int i) { return vol[i]; }, n, lane, stk);
    if (lane == 0) {
        double r = 0.0;
        if (!(total <= 0.f)) {                                     // volume.py:378-379: `<= 0` -- a NaN total is not caught, NaN comes out
            double above = 0.0;
            for (int k = 0; k < n; ++k)
                if (levels[k] > poc_price) above += (double)vol[k];
            if (!(above <= 0.0)) r = above / (double)total;                // :387-388
        }
        *out = r;
    }
}

extern "C" int fmk_calc_volume_percentage_above_poc_dev(fmk_ctx *ctx, const int32_t *d_price_levels, const float *d_volumes,
                                                        int64_t n, int32_t poc_price, double *d_out)
{
    if (n < 0 || n > FMK_PW_MAX_N) return fmk_set_error(ctx, FMK_E_ARG, "calc_volume_percentage_above_poc: %lld levels", (long long)n);
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    k_pct_above_poc<<<1, 64, 0, ctx->stream>>>(d_price_levels, d_volumes, (int)n, poc_price, d_out);
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}

extern "C" int fmk_calc_volume_percentage_above_poc(fmk_ctx *ctx, const int32_t *price_levels, const float *volumes,
                                                    int64_t n, int32_t poc_price, double *out)
{
    *out = 0.0;
    if (n <= 0) return FMK_OK;                                  // np.sum of nothing is 0 -> 0.0 (volume.py:377-379)
    void *d_l = nullptr, *d_v = nullptr, *d_o = nullptr;
    int rc = fmk_alloc(ctx, (size_t)n * 4, &d_l);
    if (rc == FMK_OK) rc = fmk_alloc(ctx, (size_t)n * 4, &d_v);
    if (rc == FMK_OK) rc = fmk_alloc(ctx, 8, &d_o);
    if (rc == FMK_OK) rc = fmk_h2d(ctx, d_l, price_levels, (size_t)n * 4);
    if (rc == FMK_OK) rc = fmk_h2d(ctx, d_v, volumes, (size_t)n * 4);
    if (rc == FMK_OK) rc = fmk_calc_volume_percentage_above_poc_dev(ctx, (const int32_t *)d_l, (const float *)d_v, n, poc_price, (double *)d_o);
    if (rc == FMK_OK) rc = fmk_d2h(ctx, out, d_o, 8);
    if (d_l) fmk_free(ctx, d_l);
    if (d_v) fmk_free(ctx, d_v);
    if (d_o) fmk_free(ctx, d_o);
    return rc;
}


// ---- the three stages of volume_profile_rolling as stand-alone calls (round 5; SURVEY 8(f) row 2 lists them by name) ----------------
// aggregate_footprint (volume.py:134-203), bucket_price_levels (:207-275), comp_poc_hva_lva (:278-365): the same device code
// paths as inside k_volume_profile -- bars added one after the other, lanes across a bar's (distinct) levels; one lane per bin adding
// in element order; NumPy-pairwise float32 total, first argmax, the value-area walk by one lane -- each on ONE window / profile, one
// wave.  Host flavour only: these are the reference's helper functions on NumPy arrays; the rolling kernel is the hot path.

// aggregate: ab / as[0 .. L) += the buy / sell volumes of bars [s, e), level l -> slot l - minl (np.searchsorted on the complete
// range :196); a level outside the range -> *bad (the reference's fancy-index `+=` would raise IndexError or fold it onto the end)
__global__ __launch_bounds__(64) void k_vp_aggregate_one(const int64_t *__restrict__ off, const int32_t *__restrict__ levels,
                                                         const float *__restrict__ buy, const float *__restrict__ sell, int64_t s,
                                                         int64_t e, int64_t minl, int L, float *__restrict__ ab, float *__restrict__ as,
                                                         int *__restrict__ bad_out)
{
    const int lane = fmk_lane();
    bool bad = false;
    for (int k = lane; k < L; k += 64) { ab[k] = 0.f; as[k] = 0.f; }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    for (int64_t t = s; t < e; ++t) {                              // bars in order: every level sees its float32 adds in the reference's order
        const int64_t r0 = off[t], r1 = off[t + 1];
        for (int64_t r = r0 + lane; r < r1; r += 64) {
            const int64_t idx = (int64_t)levels[r] - minl;
            if (idx < 0 || idx >= L) { bad = true; continue; }
            ab[idx] += buy[r];
            as[idx] += sell[r];
        }
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
    }
    if (__ballot(bad) != 0 && lane == 0) *bad_out = 1;
}

extern "C" int fmk_aggregate_footprint(fmk_ctx *ctx, const int64_t *bar_ts, const double *highs, const double *lows,
                                       const int64_t *level_offsets, const int32_t *price_levels, const float *buy_volumes,
                                       const float *sell_volumes, int64_t n_bars, int64_t start_ts, int64_t end_ts, double price_tick,
                                       int32_t *min_level, int64_t *n_levels, float *aligned_buy, float *aligned_sell, int64_t capacity)
{
    if (n_bars <= 0) return fmk_set_error(ctx, FMK_E_ARG, "Input arrays should have the same length and be non-empty.");
    if (!(price_tick > 0)) return fmk_set_error(ctx, FMK_E_ARG, "price_tick must be > 0");
    // the window and its level range on the host, exactly as volume.py:158-186 (searchsorted left / right, the one-bar fallback,
    // min / max of the window's lows / highs, int(round(x / price_tick)) = half-even)
    int64_t s = 0, e = 0;
    { int64_t lo = 0, hi = n_bars; while (lo < hi) { const int64_t mid = lo + (hi - lo) / 2; if (bar_ts[mid] < start_ts) lo = mid + 1; else hi = mid; } s = lo; }
    { int64_t lo = 0, hi = n_bars; while (lo < hi) { const int64_t mid = lo + (hi - lo) / 2; if (bar_ts[mid] <= end_ts) lo = mid + 1; else hi = mid; } e = lo; }
    if (s == e) s = s - 1 > 0 ? s - 1 : 0;
    if (s >= e) return fmk_set_error(ctx, FMK_E_ARG, "zero-size array to reduction operation minimum which has no identity");   // np.min of an empty window
    double mn = INFINITY, mx = -INFINITY;
    for (int64_t t = s; t < e; ++t) { mn = fmin(mn, lows[t]); mx = fmax(mx, highs[t]); }
    const int64_t minl = (int64_t)nearbyint(mn / price_tick), maxl = (int64_t)nearbyint(mx / price_tick);
    const int64_t L = maxl >= minl ? maxl - minl + 1 : 0;
    *min_level = (int32_t)minl;
    *n_levels = L;
    if (!aligned_buy || !aligned_sell) return FMK_OK;              // size query
    if (capacity < L) return fmk_set_error(ctx, FMK_E_CAPACITY, "aggregate_footprint: capacity %lld < %lld levels", (long long)capacity, (long long)L);
    if (L == 0) return FMK_OK;
    if (L > VP_MAX_LEVELS_GLOBAL) return fmk_set_error(ctx, FMK_E_CAPACITY, "aggregate_footprint: %lld price levels", (long long)L);
    const int64_t r0 = level_offsets[s], r1 = level_offsets[e], rows = r1 - r0, nw = e - s;
    void *d_off = nullptr, *d_lv = nullptr, *d_b = nullptr, *d_s = nullptr, *d_ab = nullptr, *d_as = nullptr, *d_bad = nullptr;
    int rc = fmk_alloc(ctx, (size_t)(nw + 1) * 8, &d_off);
    if (rc == FMK_OK) rc = fmk_alloc(ctx, (size_t)(rows > 0 ? rows : 1) * 4, &d_lv);
    if (rc == FMK_OK) rc = fmk_alloc(ctx, (size_t)(rows > 0 ? rows : 1) * 4, &d_b);
    if (rc == FMK_OK) rc = fmk_alloc(ctx, (size_t)(rows > 0 ? rows : 1) * 4, &d_s);
    if (rc == FMK_OK) rc = fmk_alloc(ctx, (size_t)L * 4, &d_ab);
    if (rc == FMK_OK) rc = fmk_alloc(ctx, (size_t)L * 4, &d_as);
    if (rc == FMK_OK) rc = fmk_alloc(ctx, 8, &d_bad);
    if (rc == FMK_OK) rc = fmk_h2d(ctx, d_off, level_offsets + s, (size_t)(nw + 1) * 8);
    if (rc == FMK_OK && rows > 0) rc = fmk_h2d(ctx, d_lv, price_levels + r0, (size_t)rows * 4);
    if (rc == FMK_OK && rows > 0) rc = fmk_h2d(ctx, d_b, buy_volumes + r0, (size_t)rows * 4);
    if (rc == FMK_OK && rows > 0) rc = fmk_h2d(ctx, d_s, sell_volumes + r0, (size_t)rows * 4);
    int bad = 0;
    if (rc == FMK_OK) {
        hipError_t he = hipSetDevice(ctx->device);
        if (he == hipSuccess) he = hipMemsetAsync(d_bad, 0, 8, ctx->stream);
        if (he == hipSuccess) {
            // (the offsets were uploaded from entry s on: rebase the rows by -r0 through the pointers)
            k_vp_aggregate_one<<<1, 64, 0, ctx->stream>>>((const int64_t *)d_off, (const int32_t *)d_lv - r0, (const float *)d_b - r0,
                                                          (const float *)d_s - r0, 0, nw, minl, (int)L, (float *)d_ab, (float *)d_as, (int *)d_bad);
            he = hipGetLastError();
        }
        if (he != hipSuccess) rc = fmk_set_error(ctx, FMK_E_HIP, "aggregate_footprint: %s", hipGetErrorString(he));
    }
    if (rc == FMK_OK) rc = fmk_d2h(ctx, aligned_buy, d_ab, (size_t)L * 4);
    if (rc == FMK_OK) rc = fmk_d2h(ctx, aligned_sell, d_as, (size_t)L * 4);
    if (rc == FMK_OK) rc = fmk_d2h(ctx, &bad, d_bad, 4);
    void *all[] = {d_off, d_lv, d_b, d_s, d_ab, d_as, d_bad};
    for (void *p : all) if (p) fmk_free(ctx, p);
    if (rc == FMK_OK && bad) return fmk_set_error(ctx, FMK_E_LEVEL, "aggregate_footprint: footprint level outside its window");
    return rc;
}

// bucket: one wave.  min / max of the levels, odd bin width, edges = arange(min, max + width, width); bin of a level by its offset
// (np.digitize on those edges - 1), every bin's float32 sum in ELEMENT order; the leftover bin exists iff the LAST element falls
// past the bins (:249).  out_n = bins (+ 1).
__global__ __launch_bounds__(64) void k_vp_bucket_one(const int32_t *__restrict__ levels, const float *__restrict__ vol, int n,
                                                      int64_t n_bins, int32_t *__restrict__ out_levels, float *__restrict__ out_vol,
                                                      int cap, int *__restrict__ out_n, int *__restrict__ st)
{
    const int lane = fmk_lane();
    int mn = INT32_MAX, mx = INT32_MIN;
    for (int i = lane; i < n; i += 64) { const int v = levels[i]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
    mn = fmk_dpp_reduce(mn, INT32_MAX, FmkOpMin());
    mx = fmk_dpp_reduce(mx, INT32_MIN, FmkOpMax());
    const int64_t range = (int64_t)mx - mn;
    int64_t bw = range / n_bins;                                   // (n_bins == 0 is refused on the host: ZeroDivisionError)
    if (bw < 1) bw = 1;
    if (bw % 2 == 0) bw += 1;
    const int64_t n_edges = (range + bw + bw - 1) / bw;            // len(arange(min, max + bw, bw))
    const int64_t nbk = n_edges - 1;
    if (n_edges < 2) { if (lane == 0) { *st = VP_ONE_LEVEL; *out_n = 0; } return; }
    const int64_t last_bin = ((int64_t)levels[n - 1] - mn) / bw;
    const bool leftover = last_bin >= nbk;
    const int np_ = (int)(nbk + (leftover ? 1 : 0));
    if (lane == 0) *out_n = np_;
    if (np_ > cap) return;                                         // size query
    bool oob = false;
    for (int b = lane; b < np_; b += 64) {
        float acc = 0.f;
        for (int i = 0; i < n; ++i) {
            int64_t k = ((int64_t)levels[i] - mn) / bw;
            if (k > nbk) k = nbk;
            if (k == nbk && !leftover) { oob = true; continue; }   // (the reference indexes past its array here)
            if (k == b) acc += vol[i];
        }
        out_vol[b] = acc;
        out_levels[b] = b < nbk ? (int32_t)(mn + (int64_t)b * bw + (bw - 1) / 2) : (int32_t)mx;   // (e_k + e_k+1 - 1) // 2; the leftover bin: max
    }
    if (__ballot(oob) != 0 && lane == 0) *st = VP_BAD_LEVEL;
}

extern "C" int fmk_bucket_price_levels(fmk_ctx *ctx, const int32_t *all_price_levels, const float *total_volumes, int64_t n,
                                       int64_t n_bins, int32_t *binned_price_levels, float *binned_volumes, int64_t capacity,
                                       int64_t *n_out)
{
    *n_out = 0;
    if (n <= 0) return fmk_set_error(ctx, FMK_E_ARG, "zero-size array to reduction operation minimum which has no identity");
    if (n > FMK_PW_MAX_N) return fmk_set_error(ctx, FMK_E_ARG, "bucket_price_levels: %lld levels", (long long)n);
    if (n_bins == 0) return fmk_set_error(ctx, FMK_E_ZERODIV, "integer division or modulo by zero");
    void *d_l = nullptr, *d_v = nullptr, *d_ol = nullptr, *d_ov = nullptr, *d_m = nullptr;
    const int64_t cap = binned_price_levels && binned_volumes ? capacity : 0;
    int rc = fmk_alloc(ctx, (size_t)n * 4, &d_l);
    if (rc == FMK_OK) rc = fmk_alloc(ctx, (size_t)n * 4, &d_v);
    if (rc == FMK_OK) rc = fmk_alloc(ctx, (size_t)(cap > 0 ? cap : 1) * 4, &d_ol);
    if (rc == FMK_OK) rc = fmk_alloc(ctx, (size_t)(cap > 0 ? cap : 1) * 4, &d_ov);
    if (rc == FMK_OK) rc = fmk_alloc(ctx, 16, &d_m);
    if (rc == FMK_OK) rc = fmk_h2d(ctx, d_l, all_price_levels, (size_t)n * 4);
    if (rc == FMK_OK) rc = fmk_h2d(ctx, d_v, total_volumes, (size_t)n * 4);
    int m[2] = {0, 0};
    if (rc == FMK_OK) {
        hipError_t he = hipSetDevice(ctx->device);
        if (he == hipSuccess) he = hipMemsetAsync(d_m, 0, 16, ctx->stream);
        if (he == hipSuccess) {
            k_vp_bucket_one<<<1, 64, 0, ctx->stream>>>((const int32_t *)d_l, (const float *)d_v, (int)n, n_bins, (int32_t *)d_ol,
                                                       (float *)d_ov, (int)cap, (int *)d_m, (int *)d_m + 1);
            he = hipGetLastError();
        }
        if (he != hipSuccess) rc = fmk_set_error(ctx, FMK_E_HIP, "bucket_price_levels: %s", hipGetErrorString(he));
    }
    if (rc == FMK_OK) rc = fmk_d2h(ctx, m, d_m, 8);
    if (rc == FMK_OK && m[1] == 0 && m[0] > 0 && m[0] <= cap) {
        rc = fmk_d2h(ctx, binned_price_levels, d_ol, (size_t)m[0] * 4);
        if (rc == FMK_OK) rc = fmk_d2h(ctx, binned_volumes, d_ov, (size_t)m[0] * 4);
    }
    void *all[] = {d_l, d_v, d_ol, d_ov, d_m};
    for (void *p : all) if (p) fmk_free(ctx, p);
    if (rc != FMK_OK) return rc;
    if (m[1] & VP_ONE_LEVEL) return fmk_set_error(ctx, FMK_E_LEVEL, "bucket_price_levels: a single price level cannot be bucketed (the reference raises a broadcast ValueError here)");
    if (m[1] & VP_BAD_LEVEL) return fmk_set_error(ctx, FMK_E_LEVEL, "bucket_price_levels: a price level past the last bin in front of the last element");
    *n_out = m[0];
    if (cap > 0 && m[0] > cap) return fmk_set_error(ctx, FMK_E_CAPACITY, "bucket_price_levels: capacity %lld < %d bins", (long long)capacity, m[0]);
    return FMK_OK;
}

// POC / value area of one profile: the walk of k_volume_profile on explicit level values
__global__ __launch_bounds__(64) void k_vp_poc_one(const int32_t *__restrict__ levels, const float *__restrict__ vol, int n,
                                                   double va_pct, int32_t *__restrict__ out3)
{
    __shared__ __attribute__((aligned(8))) int stk[FMK_PW_STK_F32];
    const int lane = fmk_lane();
    const float total = fmk_pairwise_f32([&](int i) { return vol[i]; }, n, lane, stk);   // np.sum of a float32 array
    float best = -INFINITY;
    int best_i = 0x7FFFFFFF;
    for (int k = lane; k < n; k += 64) {
        const float v = vol[k];
        if (v > best || (v != v && best == best)) { best = v; best_i = k; }              // np.argmax: first maximum, the first NaN wins
    }
#pragma unroll
    for (int x = 32; x > 0; x >>= 1) {
        const float ob = __shfl_xor(best, x, 64);
        const int oi = __shfl_xor(best_i, x, 64);
        const bool on = ob != ob, bn = best != best;
        const bool take = on ? (!bn || oi < best_i) : (!bn && (ob > best || (ob == best && oi < best_i)));
        if (take) { best = ob; best_i = oi; }
    }
    if (best_i == 0x7FFFFFFF) best_i = 0;
    if (lane != 0) return;
    const int pi = best_i;
    const int32_t poc_price = levels[pi];
    const double va_thrs = (double)total * (va_pct / 100.0);
    double cum = vol[pi];
    int32_t hv = poc_price, lv = poc_price;
    int up = pi + 1, down = pi - 1;
    double cu = 0.0, cd = 0.0;
    if (up < n) { cu = vol[up]; if (up + 1 < n) cu += vol[up + 1]; }
    if (down >= 0) { cd = vol[down]; if (down - 1 >= 0) cd += vol[down - 1]; }
    while (cum < va_thrs) {
        if (cu > cd) {
            cum += cu;
            hv = levels[up + 1 < n - 1 ? up + 1 : n - 1];
            up += 2;
            cu = -1.0;
            if (up < n) { cu = vol[up]; if (up + 1 < n) cu += vol[up + 1]; }
        } else if (cu < cd) {
            cum += cd;
            lv = levels[down - 1 > 0 ? down - 1 : 0];
            down -= 2;
            cd = -1.0;
            if (down >= 0) { cd = vol[down]; if (down - 1 >= 0) cd += vol[down - 1]; }
        } else if (cu == cd && cd != -1.0) {
            cum += cu + cd;
            hv = levels[up + 1 < n - 1 ? up + 1 : n - 1];
            lv = levels[down - 1 > 0 ? down - 1 : 0];
            up += 2; down -= 2;
            cu = -1.0;
            if (up < n) { cu = vol[up]; if (up + 1 < n) cu += vol[up + 1]; }
            cd = -1.0;
            if (down >= 0) { cd = vol[down]; if (down - 1 >= 0) cd += vol[down - 1]; }
        } else break;                                              // the reference's "stuck in loop" exit
    }
    out3[0] = poc_price; out3[1] = hv; out3[2] = lv;
}

extern "C" int fmk_comp_poc_hva_lva(fmk_ctx *ctx, const int32_t *price_levels, const float *volumes, int64_t n, double va_pct,
                                    int32_t *poc_price, int32_t *hva_price, int32_t *lva_price)
{
    if (n <= 0) return fmk_set_error(ctx, FMK_E_ARG, "attempt to get argmax of an empty sequence");
    if (n > FMK_PW_MAX_N) return fmk_set_error(ctx, FMK_E_ARG, "comp_poc_hva_lva: %lld levels", (long long)n);
    void *d_l = nullptr, *d_v = nullptr, *d_o = nullptr;
    int rc = fmk_alloc(ctx, (size_t)n * 4, &d_l);
    if (rc == FMK_OK) rc = fmk_alloc(ctx, (size_t)n * 4, &d_v);
    if (rc == FMK_OK) rc = fmk_alloc(ctx, 16, &d_o);
    if (rc == FMK_OK) rc = fmk_h2d(ctx, d_l, price_levels, (size_t)n * 4);
    if (rc == FMK_OK) rc = fmk_h2d(ctx, d_v, volumes, (size_t)n * 4);
    if (rc == FMK_OK) {
        hipError_t he = hipSetDevice(ctx->device);
        if (he == hipSuccess) {
            k_vp_poc_one<<<1, 64, 0, ctx->stream>>>((const int32_t *)d_l, (const float *)d_v, (int)n, va_pct, (int32_t *)d_o);
            he = hipGetLastError();
        }
        if (he != hipSuccess) rc = fmk_set_error(ctx, FMK_E_HIP, "comp_poc_hva_lva: %s", hipGetErrorString(he));
    }
    int32_t o[3] = {0, 0, 0};
    if (rc == FMK_OK) rc = fmk_d2h(ctx, o, d_o, 12);
    if (d_l) fmk_free(ctx, d_l);
    if (d_v) fmk_free(ctx, d_v);
    if (d_o) fmk_free(ctx, d_o);
    *poc_price = o[0]; *hva_price = o[1]; *lva_price = o[2];
    return rc;
}
