// fmk_median.hip -- median trade size per bar (np.median over the bar's amounts,
// finmlkit/bar/base.py:373-403) on gfx950.  Exact order statistics, no sort of the bar.
//
// One wave owns one bar.  The bar's amounts are loaded ONCE (coalesced, all loads in flight
// together) into registers as order-preserving integer keys (float bits -> unsigned).  The two
// middle ranks are then bracketed by bisection on the KEY VALUE range [min,max]: every step is
// a branch-free `count(key <= pivot)` over the register file (one v_cmp + ballot popcount per
// register, the count lands wave-uniform in SGPRs).  As soon as <= 64 candidates remain in the
// bracket they are compacted through a 64-slot LDS line, bitonic-sorted across the 64 lanes and
// the two middle elements are read off by lane index.  Uniform-ish data needs ~5-9 counting
// steps instead of the 32/64 of a bitwise search; degenerate data is bounded by the key width.
// Bars longer than 64*32 (f32) / 64*24 (f64) ticks run the same search re-reading their
// amounts (L2/MALL-resident for anything but giant bars).
//
// This file holds the stand-alone kernel (any bar length).  For bars of <= FMK_SMALL_BAR_TICKS
// ticks comp_bar_ohlcv uses the fused kernel of fmk_ohlcv.hip, which already has the amounts in
// registers; `min_cnt` lets this kernel process only the longer bars in that case.
//
// Traffic: amount column once (4 or 8 B/tick) + 8 B/bar.
#include "fmk_median.h"

// Bars beyond the register classes (more than 2048 float32 / 1536 float64 ticks: 2-minute bars and up on the bench tape).  The first
// version ran the same value-range bisection with ONE wave re-reading its bar at every step: 20 ms per 1e9 ticks at 2 400-tick
// bars, 0.31 s at hourly bars, 13.7 s at daily bars (580 waves on the whole chip).  Now a WORKGROUP takes such a bar and
// selects the two middle ranks by radix: BITS / 8 passes over the bar, 256-bin histograms of the current digit in LDS for the
// keys that share the prefix found so far (one histogram while both ranks still share it), a block scan picks the digit.
// The bars are found like in k_bar_median's leftover pass: 64 close indices per coalesced load.
#define ML_MIN(F64) ((F64) ? 64 * 24 : 64 * 32)
#define ML_THREADS 1024                  // threads per bar beyond ML_MID_MAX ticks (bins 0..255 of the scans are the first 256)
#define ML_MID_MAX 8192
template <bool AF64, int THREADS>
__global__ __launch_bounds__(THREADS) void k_bar_median_long(const void *__restrict__ amount, const int64_t *__restrict__ ci,
                                                         const int64_t *__restrict__ list, const int *__restrict__ go,
                                                         double *__restrict__ o_median)
{
    if (go && *go == 0) return;                          // the fused small-bar kernel saw no long bar
    typedef MedKey<AF64> MK;
    const int64_t n_list = list[0];
    for (int64_t q = blockIdx.x; q < n_list; q += gridDim.x) {
        const int64_t b = list[1 + q], s = ci[b], e = ci[b + 1];
        const int64_t cnt = e - s, start = s + 1;
        typename MK::K k1, k2;
        bool any_nan;
        med_block_select<AF64, THREADS>(amount, start, cnt, (cnt - 1) >> 1, cnt >> 1, k1, k2, any_nan);
        if (threadIdx.x == 0) {
            const double v1 = MK::value(k1), v2 = MK::value(k2);
            // np.median: mean of the two middle elements == (a + b) / 2.0 ; odd count: the middle one; NaN if any NaN
            o_median[b] = any_nan ? (double)NAN : (cnt & 1) ? v1 : (v1 + v2) / 2.0;
        }
    }
}

// bars of more than min_cnt ticks -> list (one thread per bar, one atomic per wave)
__global__ __launch_bounds__(256) void k_long_bar_list(const int64_t *__restrict__ ci, int64_t nb, int64_t min_cnt, int64_t max_cnt,
                                                       const int *__restrict__ go, int64_t *__restrict__ list, int64_t cap)
{
    if (go && *go == 0) return;
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t cnt = b < nb ? ci[b + 1] - ci[b] : 0;
    const bool is_long = cnt > min_cnt && cnt <= max_cnt;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(is_long);
    if (m == 0) return;
    const int lane = fmk_lane();
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd((unsigned long long *)list, (unsigned long long)__builtin_popcountll(m));
    base = (unsigned long long)fmk_uniform((int64_t)base);
    const int64_t pos = (int64_t)base + __builtin_popcountll(m & ((1ULL << lane) - 1));
    if (is_long && pos < cap) list[1 + pos] = b;
}

// K lists at once -- list k holds the bars of edge[k] < ticks <= edge[k + 1] -- from ONE pass over the close indices into ONE
// allocation (free lists[0]): the six size classes of comp_bar_ohlcv's middle range used to cost six memsets and six launches.
struct LongBarEdges { int64_t edge[FMK_MAX_BAR_LISTS + 1]; int64_t *list[FMK_MAX_BAR_LISTS]; int64_t cap[FMK_MAX_BAR_LISTS]; int k; };
// 1 024 bars per workgroup; a list's slots are claimed once per WORKGROUP (the waves' counts meet in LDS): on a stream of unequal bars
// nearly every wave has bars for several lists, and one atomic per wave and list on eight counters cost 0.27 ms per call at 8e5 bars.
#define LBL_THREADS 1024
__global__ __launch_bounds__(LBL_THREADS) void k_long_bar_lists(const int64_t *__restrict__ ci, int64_t nb, LongBarEdges L, const int *__restrict__ go)
{
    if (go && *go == 0) return;
    __shared__ int s_cnt[FMK_MAX_BAR_LISTS][LBL_THREADS / 64];
    __shared__ unsigned long long s_base[FMK_MAX_BAR_LISTS];
    const int64_t b = (int64_t)blockIdx.x * LBL_THREADS + threadIdx.x;
    const int64_t cnt = b < nb ? ci[b + 1] - ci[b] : 0;
    const int lane = fmk_lane(), w = (int)(threadIdx.x >> 6);
    const bool any = cnt > L.edge[0] && cnt <= L.edge[L.k];
    if (__syncthreads_or(any) == 0) return;
    int mine = -1, pos_in_wave = 0;
    for (int k = 0; k < L.k; ++k) {
        const bool in = cnt > L.edge[k] && cnt <= L.edge[k + 1];
        const unsigned long long m = __builtin_amdgcn_ballot_w64(in);
        if (lane == 0) s_cnt[k][w] = __builtin_popcountll(m);
        if (in) { mine = k; pos_in_wave = __builtin_popcountll(m & ((1ULL << lane) - 1)); }
    }
    __syncthreads();
    if ((int)threadIdx.x < L.k) {
        int tot = 0;
        for (int q = 0; q < LBL_THREADS / 64; ++q) { const int c = s_cnt[threadIdx.x][q]; s_cnt[threadIdx.x][q] = tot; tot += c; }
        s_base[threadIdx.x] = tot ? atomicAdd((unsigned long long *)L.list[threadIdx.x], (unsigned long long)tot) : 0ULL;
    }
    __syncthreads();
    if (mine >= 0) {
        const int64_t pos = (int64_t)s_base[mine] + s_cnt[mine][w] + pos_in_wave;
        if (pos < L.cap[mine]) L.list[mine][1 + pos] = b;
    }
}

int fmk_long_bar_lists(fmk_ctx *ctx, const int64_t *d_close_idx, int64_t nb, int64_t n, int k, const int64_t *edge, const int *d_go,
                       int64_t **lists)
{
    if (k < 1 || k > FMK_MAX_BAR_LISTS) return fmk_set_error(ctx, FMK_E_ARG, "fmk_long_bar_lists: %d lists", k);
    LongBarEdges L;
    L.k = k;
    size_t total = 0;
    for (int q = 0; q <= k; ++q) L.edge[q] = edge[q];
    for (int q = 0; q < k; ++q) {
        int64_t cap = n / (edge[q] > 0 ? edge[q] : 1) + 2;          // bars of more than edge[q] ticks: fewer than n / edge[q]
        if (cap > nb) cap = nb;
        L.cap[q] = cap;
        total += (size_t)(cap + 1);
    }
    void *p = nullptr;
    FMK_TRY(fmk_alloc(ctx, total * 8, &p));
    int64_t *at = (int64_t *)p;
    for (int q = 0; q < k; ++q) { L.list[q] = at; lists[q] = at; at += L.cap[q] + 1; }
    // the counters sit at the heads of the lists: one memset over the block (a few MB at most) instead of one per list
    FMK_HIP(ctx, hipMemsetAsync(p, 0, total * 8, ctx->stream));
    k_long_bar_lists<<<(unsigned)fmk_ceil_div(nb, LBL_THREADS), LBL_THREADS, 0, ctx->stream>>>(d_close_idx, nb, L, d_go);
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}

int fmk_long_bar_list(fmk_ctx *ctx, const int64_t *d_close_idx, int64_t nb, int64_t n, int64_t min_cnt, const int *d_go,
                      int64_t **list, int64_t max_cnt)
{
    int64_t cap = n / (min_cnt > 0 ? min_cnt : 1) + 2;               // bars of more than min_cnt ticks: fewer than n / min_cnt
    if (cap > nb) cap = nb;
    void *p = nullptr;
    FMK_TRY(fmk_alloc(ctx, (size_t)(cap + 1) * 8, &p));
    *list = (int64_t *)p;
    FMK_HIP(ctx, hipMemsetAsync(p, 0, 8, ctx->stream));
    k_long_bar_list<<<(unsigned)fmk_ceil_div(nb, 256), 256, 0, ctx->stream>>>(d_close_idx, nb, min_cnt, max_cnt, d_go, *list, cap);
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}

template <bool AF64>
__global__ __launch_bounds__(256) void k_bar_median(const void *__restrict__ amount,
                                                    const int64_t *__restrict__ ci, int64_t nb, int64_t min_cnt,
                                                    const int *__restrict__ go, double *__restrict__ o_median,
                                                    int64_t skip_lo = 0, int64_t skip_hi = 0)
{
    if (go && *go == 0) return;                          // the fused small-bar kernel saw no long bar
    typedef typename MedKey<AF64>::K K;
    __shared__ K sbuf[4][64];
    const int lane = fmk_lane();
    const int wib = fmk_uniform((int)(threadIdx.x >> 6));
    const int wpb = blockDim.x >> 6;
    const int64_t wave0 = (int64_t)blockIdx.x * wpb + wib;
    const int64_t nwaves = (int64_t)gridDim.x * wpb;
    K *buf = sbuf[wib];
    auto do_bar = [&](int64_t b, int64_t s, int64_t e) {
        const int64_t cnt = e - s;
        double m = 0.0;
        if (cnt > 0) {
            const int64_t start = s + 1;
            const int nreg = (int)((cnt + 63) >> 6);
            if (cnt > ML_MIN(AF64)) return;                   // k_bar_median_long
            if (cnt > skip_lo && cnt <= skip_hi) return;      // k_bar_ohlcv_mid wrote it
            else if (nreg <= 1) m = med_select<AF64, 1>(amount, start, cnt, lane, buf);
            else if (nreg <= 4) m = med_select<AF64, 4>(amount, start, cnt, lane, buf);
            else if (nreg <= 8) m = med_select<AF64, 8>(amount, start, cnt, lane, buf);
            else if (nreg <= 12) m = med_select<AF64, 12>(amount, start, cnt, lane, buf);
            else if (nreg <= 16) m = med_select<AF64, 16>(amount, start, cnt, lane, buf);
            else if (nreg <= 20) m = med_select<AF64, 20>(amount, start, cnt, lane, buf);
            else if (nreg <= 24) m = med_select<AF64, 24>(amount, start, cnt, lane, buf);
            else if constexpr (!AF64) m = med_select<AF64, 32>(amount, start, cnt, lane, buf);
        }
        if (lane == 0) o_median[b] = m;
    };
    if (min_cnt > 0) {
        // the leftover pass of a small-bar kernel: 64 bars per step, one coalesced load of their close indices, then only the
        // bars it left (longer than min_cnt) get the wave -- walking the bars one by one cost a dependent load per bar, 2.3 ms
        // per 2.5e7 bars of which a few per cent were long
        const int64_t ngroups = (nb + 63) >> 6;
        for (int64_t g = wave0; g < ngroups; g += nwaves) {
            const int64_t bl = g * 64 + lane;
            int64_t s_l = 0, e_l = 0;
            if (bl < nb) { s_l = ci[bl]; e_l = ci[bl + 1]; }
            unsigned long long todo = __builtin_amdgcn_ballot_w64(bl < nb && e_l - s_l > min_cnt);
            while (todo) {
                const int bit = fmk_uniform((int)__builtin_ctzll(todo));
                todo &= todo - 1;
                do_bar(g * 64 + bit, fmk_readlane(s_l, bit), fmk_readlane(e_l, bit));
            }
        }
        return;
    }
    for (int64_t b = wave0; b < nb; b += nwaves) do_bar(b, fmk_uniform(ci[b]), fmk_uniform(ci[b + 1]));
}

int fmk_median_launch(fmk_ctx *ctx, const void *d_amount, int amount_is_f64, const int64_t *d_close_idx, int64_t nb,
                      int64_t min_cnt, const int *d_go, double *d_median, int64_t n_ticks, int64_t skip_lo, int64_t skip_hi,
                      int64_t skip_above)
{
    int64_t blocks = fmk_ceil_div(nb, 4);
    const int64_t cap = (int64_t)ctx->n_cu * 64;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if (amount_is_f64)
        k_bar_median<true><<<(unsigned)blocks, 256, 0, ctx->stream>>>(d_amount, d_close_idx, nb, min_cnt, d_go, d_median, skip_lo, skip_hi);
    else
        k_bar_median<false><<<(unsigned)blocks, 256, 0, ctx->stream>>>(d_amount, d_close_idx, nb, min_cnt, d_go, d_median, skip_lo, skip_hi);
    FMK_LAUNCH_CHECK(ctx);
    // bars beyond the register classes: a workgroup per bar, the bars from lists -- 256 threads for bars up to ML_MID_MAX ticks (eight
    // workgroups per CU: at 2 400-tick bars 1024 threads per bar left most of them waiting at the barriers, 12.9 ms per 1e9 ticks),
    // 1024 threads beyond (two per CU)
    int64_t lo_cnt = ML_MIN(amount_is_f64);
    if (skip_hi > lo_cnt && skip_lo <= lo_cnt) lo_cnt = skip_hi;      // the skipped range covers the start of the mid list
    int64_t *list_mid = nullptr, *list_long = nullptr;
    FMK_TRY(fmk_long_bar_list(ctx, d_close_idx, nb, n_ticks, lo_cnt, d_go, &list_mid, ML_MID_MAX));
    int rc = fmk_long_bar_list(ctx, d_close_idx, nb, n_ticks, ML_MID_MAX > lo_cnt ? ML_MID_MAX : lo_cnt, d_go, &list_long, skip_above);
    if (rc == FMK_OK) {
        if (amount_is_f64) {
            k_bar_median_long<true, 256><<<(unsigned)(ctx->n_cu * 8), 256, 0, ctx->stream>>>(d_amount, d_close_idx, list_mid, d_go, d_median);
            k_bar_median_long<true, ML_THREADS><<<(unsigned)(ctx->n_cu * 2), ML_THREADS, 0, ctx->stream>>>(d_amount, d_close_idx, list_long, d_go, d_median);
        } else {
            k_bar_median_long<false, 256><<<(unsigned)(ctx->n_cu * 8), 256, 0, ctx->stream>>>(d_amount, d_close_idx, list_mid, d_go, d_median);
            k_bar_median_long<false, ML_THREADS><<<(unsigned)(ctx->n_cu * 2), ML_THREADS, 0, ctx->stream>>>(d_amount, d_close_idx, list_long, d_go, d_median);
        }
    }
    const hipError_t le = hipGetLastError();
    (void)fmk_free(ctx, list_mid);
    if (list_long) (void)fmk_free(ctx, list_long);
    FMK_TRY(rc);
    FMK_HIP(ctx, le);
    return FMK_OK;
}

int fmk_median_long_list_launch(fmk_ctx *ctx, const void *d_amount, const int64_t *d_close_idx, const int64_t *d_list, double *d_median)
{
    k_bar_median_long<false, ML_THREADS><<<(unsigned)(ctx->n_cu * 2), ML_THREADS, 0, ctx->stream>>>(d_amount, d_close_idx, d_list, nullptr,
                                                                                              d_median);
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}

extern "C" int fmk_comp_bar_median_dev(fmk_ctx *ctx, const void *d_amount, int amount_is_f64, int64_t n,
                                       const int64_t *d_close_idx, int64_t n_idx, double *d_median)
{
    if (n_idx < 2)
        return fmk_set_error(ctx, FMK_E_ARG, "Bar close indices must contain at least two elements.");
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    return fmk_median_launch(ctx, d_amount, amount_is_f64, d_close_idx, n_idx - 1, 0, nullptr, d_median, n);
}
