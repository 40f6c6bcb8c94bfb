// fmk_ohlcv.hip -- comp_bar_ohlcv (finmlkit/bar/base.py:306-407) on gfx950.
//
// Layout: price f64[N], amount f32|f64[N], close_idx i64[B+1] resident in HBM.
// One 64-lane wave owns one bar (bars are contiguous tick ranges, so the wave streams
// 512 B (price) + 256 B (amount) fully coalesced per load instruction), keeps
// hi/lo/sum(vol)/sum(price*vol) in registers and folds them with a 6-step xor butterfly when the
// bar ends.  No atomics, no MFMA: bounded by HBM read bandwidth, 12 B/tick (f32 amounts) +
// 8 B/bar read + 60..68 B/bar written.
//
// Two kernels:
//   k_bar_ohlcv_small  bars of <= 64*NCH ticks (up to 21 chunks = 1344 ticks: every 1-minute bar of the
//                      benchmark stream).  ALL loads of the bar are issued up front (one HBM round
//                      trip per bar instead of one per unrolled batch), and -- when the median trade
//                      size is requested (base.py:403) -- the amounts that are already in registers
//                      feed the exact order-statistic search of fmk_median.h directly: the amount
//                      column is read ONCE for OHLCV + median.
//   k_bar_ohlcv        any bar length (unroll-4 streaming loop); with min_cnt > 0 it only takes the
//                      bars the small kernel skipped.  Long bars get their median from k_bar_median.
//
// Floating point: price*volume is rounded before it is added (-ffp-contract=off), exactly
// like the reference; the per-bar float64 sums are accumulated lane-strided and then
// tree-reduced, i.e. in a different order than the reference's sequential loop
// (|rel. diff| ~ 1e-16, the north-star tolerance is 1e-9).  Both kernels use the same order.
#include <math.h>
#include <stdlib.h>

#include <utility>

#include "fmk_common.h"
#include "fmk_dpp.h"
#include "fmk_f32tie.h"
#include "fmk_median.h"

#define FMK_SMALL_NCH 21
#define FMK_PACKED_MAX_MEAN 64          // mean ticks per bar up to which the lane-per-bar schedule is used (float32 amounts):
                                        // at 60 ticks 9.8 vs 12.6 ms for the wave-per-bar kernel, at 80 ticks 18.6 vs 12.0 (1e9 ticks)

struct OhlcvOut {
    double *open, *high, *low, *close;
    float *vol;
    double *vwap;
    int64_t *trades;
    double *median;
    // float64 amounts only (null otherwise; last member so the other kernarg offsets do not move): redo list of the bars
    // whose volume sum sits within summation-order noise of a float32 rounding boundary (fmk_f32tie.h)
    unsigned long long *vol_redo;
};

__device__ __forceinline__ void ohlcv_empty(const OhlcvOut &o, int64_t b, const double *price, int64_t e, int64_t n)
{
    // empty bar (base.py:352-361): previous close, Python-style negative wrap of prices[end]
    double p = price[fmk_wrap(e, n)];
    o.open[b] = p; o.high[b] = p; o.low[b] = p; o.close[b] = p;
    o.vol[b] = 0.f; o.vwap[b] = 0.0; o.trades[b] = 0;
    if (o.median) o.median[b] = 0.0;
}

template <bool AF64>
__device__ __forceinline__ void ohlcv_finish(const OhlcvOut &o, int64_t b, const double *price, int64_t start,
                                             int64_t e, double hi, double lo, double tv, double td, int lane,
                                             bool reduced = false)
{
    // DPP reductions (fmk_dpp.h): the four 6-step xor butterflies that stood here were 48 ds_bpermute round trips per bar
    if (!reduced) {
        hi = fmk_dpp_reduce(hi, (double)-INFINITY, FmkOpMax());
        lo = fmk_dpp_reduce(lo, (double)INFINITY, FmkOpMin());
        tv = fmk_dpp_reduce(tv, 0.0, FmkOpAdd());
        td = fmk_dpp_reduce(td, 0.0, FmkOpAdd());
    }
    if (lane == 0) {
        // base.py:371-382 seeds high / low with the bar's first price and updates them with `>` / `<`: NaNs later in the
        // bar lose every comparison (the fmax / fmin above), but a NaN FIRST price never loses one -> high = low = NaN
        const double first = price[start];
        o.open[b] = first;
        o.close[b] = price[e];
        o.high[b] = first != first ? first : hi;
        o.low[b] = first != first ? first : lo;
        o.vol[b] = (float)tv;
        o.vwap[b] = tv > 0.0 ? td / tv : 0.0;   // base.py:398
        o.trades[b] = e - start + 1;
        // float32 amounts: 24-bit terms, so the float64 sum of a bar is exact in any order (as long as the amounts of a
        // bar span less than 2^29 in magnitude) and (float)tv is the reference's value.  float64 amounts: the order
        // matters in the last bits; bars that close to a float32 tie are redone in tick order by k_bar_vol_redo.
        if constexpr (AF64) {
            if (o.vol_redo && fmk_near_f32_tie(tv, fmk_f32tie_eps(e - start + 1) * fabs(tv)))
                o.vol_redo[32 + atomicAdd(o.vol_redo, 1ULL)] = (unsigned long long)b;
        }
    }
}

// float64 amounts: `volume` of the listed bars as the reference adds it (base.py:377-398: one float64 accumulator in
// tick order, cast once).  One wave per bar: 64 amounts per coalesced load into an LDS row, lane 0 adds them in order.
__global__ __launch_bounds__(256) void k_bar_vol_redo(const double *__restrict__ amount, const int64_t *__restrict__ ci,
                                                      float *__restrict__ vol,
                                                      const unsigned long long *__restrict__ redo)
{
    __shared__ double s_row[4][64];
    const int lane = fmk_lane();
    const int wib = fmk_uniform((int)(threadIdx.x >> 6));
    const int64_t count = (int64_t)redo[0];
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t i = (int64_t)blockIdx.x * 4 + wib; i < count; i += nwaves) {
        const int64_t b = fmk_uniform((int64_t)redo[32 + i]);
        const int64_t start = fmk_uniform(ci[b]) + 1, e = fmk_uniform(ci[b + 1]);
        double tv = 0.0;
        double v = start + lane <= e ? amount[start + lane] : 0.0;
        for (int64_t j0 = start; j0 <= e; j0 += 64) {
            s_row[wib][lane] = v;                               // lanes past the bar hold 0.0: x + 0.0 == x
            const int64_t jn = j0 + 64 + lane;
            v = jn <= e ? amount[jn] : 0.0;                     // next chunk in flight while lane 0 adds
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
#pragma unroll 16
                for (int k = 0; k < 64; ++k) tv += s_row[wib][k];
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) vol[b] = (float)tv;
    }
}

// Bars of more than OH_WIDE_MIN ticks (hourly and daily bars): a WORKGROUP of 1024 threads per bar instead of one wave -- with a
// wave per bar 580 daily bars kept 580 waves busy on the whole chip: 241 ms per 1e9 ticks (10.7 ms for hourly bars).  Same
// quantities; the float64 sums are block-reduced (wave DPP trees, then the 16 wave totals in order), which the contract
// allows: volume is exact in any order for float32 amounts (ties of float64 amounts go to k_bar_vol_redo as before), vwap is
// a reassociated sum (<= 1e-9).  The bars are found like in the leftover pass above: 64 close indices per coalesced load.
#define OH_WIDE_MIN 16384
#define OH_WIDE_THREADS 1024
struct OhlBracket;
struct OhlCount;
template <bool AF64, bool MED = false>
__global__ __launch_bounds__(OH_WIDE_THREADS) void k_bar_ohlcv_wide(const double *__restrict__ price, const void *__restrict__ amount,
                                                                   const int64_t *__restrict__ ci, const int64_t *__restrict__ list,
                                                                   const int *__restrict__ go, OhlcvOut o,
                                                                   const uint32_t *__restrict__ brk = nullptr /* OhlBracket[] */,
                                                                   uint32_t *__restrict__ cand = nullptr,
                                                                   int64_t *__restrict__ res = nullptr /* OhlCount[] */,
                                                                   int64_t base = 0 /* first tick the scratch covers (multiple of 16) */)
{
    if (go && *go == 0) return;
    __shared__ double s_red[4][OH_WIDE_THREADS / 64];
    __shared__ int s_ncand;
    __shared__ int64_t s_below[OH_WIDE_THREADS / 64];
    __shared__ int s_nan;
    const int tid = threadIdx.x, lane = fmk_lane(), w = tid >> 6;
    const int64_t n_list = list[0];
    for (int64_t q = blockIdx.x; q < n_list; q += gridDim.x) {
        const int64_t b = list[1 + q], s = ci[b], e = ci[b + 1], start = s + 1;
        double hi = -INFINITY, lo = INFINITY, tv = 0.0, td = 0.0;
        // MED: the sizes inside the bracket go to the bar's candidate slots, the keys below it are counted (float32 amounts)
        uint32_t blo = 0, bhi = 0;
        int64_t below = 0;
        bool nan = false;
        uint32_t *mycand = nullptr;
        int cap = 0;
        if constexpr (MED) {
            blo = brk[4 * q]; bhi = brk[4 * q + 1];
            mycand = cand + ((start - base) >> 2);
            cap = (int)((e - s) >> 2);
            if (tid == 0) { s_ncand = 0; s_nan = 0; }
            __syncthreads();
        }
        auto med_tick = [&](int64_t jj, bool in_bar) {
            if constexpr (MED) {
                const uint32_t raw = in_bar ? ((const uint32_t *)amount)[jj] : 0u;
                const uint32_t k = MedKey<false>::tokey(raw);
                nan |= in_bar && (k < MedKey<false>::KEY_NEG_INF || k > MedKey<false>::KEY_POS_INF);
                below += (in_bar && k < blo) ? 1 : 0;
                const bool inb = in_bar && k >= blo && k <= bhi;
                const uint64_t m = __ballot(inb);
                if (m) {
                    int base = 0;
                    if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&s_ncand, (int)__popcll(m));
                    base = __builtin_amdgcn_readlane(base, (int)__builtin_ctzll(m));
                    const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                    if (inb && pos < cap) mycand[pos] = raw;
                }
            }
        };
        int64_t j = start + tid;
        for (; j + 3 * OH_WIDE_THREADS <= e; j += 4 * OH_WIDE_THREADS) {       // four loads of each column in flight
            const double p0 = price[j], p1 = price[j + OH_WIDE_THREADS], p2 = price[j + 2 * OH_WIDE_THREADS],
                         p3 = price[j + 3 * OH_WIDE_THREADS];
            const double a0 = fmk_amt<AF64>(amount, j), a1 = fmk_amt<AF64>(amount, j + OH_WIDE_THREADS),
                         a2 = fmk_amt<AF64>(amount, j + 2 * OH_WIDE_THREADS), a3 = fmk_amt<AF64>(amount, j + 3 * OH_WIDE_THREADS);
            hi = fmax(fmax(hi, p0), fmax(p1, fmax(p2, p3)));
            lo = fmin(fmin(lo, p0), fmin(p1, fmin(p2, p3)));
            tv += a0; td += p0 * a0;
            tv += a1; td += p1 * a1;
            tv += a2; td += p2 * a2;
            tv += a3; td += p3 * a3;
            med_tick(j, true); med_tick(j + OH_WIDE_THREADS, true); med_tick(j + 2 * OH_WIDE_THREADS, true);
            med_tick(j + 3 * OH_WIDE_THREADS, true);
        }
        for (; j - tid <= e; j += OH_WIDE_THREADS) {                           // (whole waves: the ballots of med_tick)
            const bool in_bar = j <= e;
            const int64_t jc = in_bar ? j : e;
            const double p0 = price[jc], a0 = in_bar ? fmk_amt<AF64>(amount, jc) : 0.0;
            hi = fmax(hi, p0);
            lo = fmin(lo, p0);
            tv += a0; td += in_bar ? p0 * a0 : 0.0;
            med_tick(jc, in_bar);
        }
        hi = fmk_dpp_reduce(hi, (double)-INFINITY, FmkOpMax());
        lo = fmk_dpp_reduce(lo, (double)INFINITY, FmkOpMin());
        tv = fmk_dpp_reduce(tv, 0.0, FmkOpAdd());
        td = fmk_dpp_reduce(td, 0.0, FmkOpAdd());
        if constexpr (MED) {
            below = fmk_dpp_reduce(below, (int64_t)0, FmkOpAdd());
            if (lane == 0) s_below[w] = below;
            if (__ballot(nan) != 0 && lane == 0) s_nan = 1;
        }
        if (lane == 0) { s_red[0][w] = hi; s_red[1][w] = lo; s_red[2][w] = tv; s_red[3][w] = td; }
        __syncthreads();
        if constexpr (MED) {
            if (tid == 0) {
                int64_t bl = 0;
                for (int k = 0; k < OH_WIDE_THREADS / 64; ++k) bl += s_below[k];
                res[3 * q] = bl; res[3 * q + 1] = s_ncand; res[3 * q + 2] = s_nan;
            }
        }
        if (w == 0) {
            hi = s_red[0][0]; lo = s_red[1][0]; tv = s_red[2][0]; td = s_red[3][0];
            for (int k = 1; k < OH_WIDE_THREADS / 64; ++k) {
                hi = fmax(hi, s_red[0][k]); lo = fmin(lo, s_red[1][k]); tv += s_red[2][k]; td += s_red[3][k];
            }
            ohlcv_finish<AF64>(o, b, price, start, e, hi, lo, tv, td, lane, true);
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Bars of more than 16 384 ticks (float32 amounts; round 3): the median from the SAME pass as open / high / low / close / volume /
// vwap.  k_bar_median_long selects the two middle ranks by radix -- three passes over the bar with 2048-bin LDS-atomic histograms:
// hourly bars 5.4 ms, daily bars 8.6 ms per 1e9 ticks against 2.0 / 2.5 ms without the median.  Here:
//   1. k_bar_med_sample: a systematic sample of the bar (every 16th / 64th / 256th size: it covers the whole bar evenly, whatever
//      the time of day does to the sizes) -> the keys at the sample ranks s/2 -+ g, g = 1.9 sqrt(s) + 4: a bracket [blo, bhi] that
//      holds the bar's middle ranks with probability > 0.9998 and 3.8 / sqrt(s) of its ticks (12 % at 16 385 ticks, 2 % for a day);
//   2. k_bar_ohlcv_wide<MED>: the one pass -- besides the sums it counts the keys below the bracket and appends the sizes inside it
//      to the bar's candidate buffer (wave-aggregated, one LDS atomic per wave and chunk);
//   3. k_bar_med_finish: if the middle ranks fall among the candidates they are selected there exactly (the same radix select, on a
//      few per cent of the bar); otherwise the bar goes on a list for k_bar_median_long.  np.median's bits either way.
// Scratch: sample slots [(start - base) / 16, ...) and candidate slots [(start - base) / 4, ...) of two arrays indexed by the bar's
// own tick range (bars are disjoint, so no offsets have to be computed); base = the first listed bar's start rounded down to 16,
// so the arrays cover only the span of the listed bars (k_list_span) -- the whole tick axis when the call must not wait.
// ---------------------------------------------------------------------------------------------------------------------
struct OhlBracket { uint32_t blo, bhi; int ok; int pad; };
struct OhlCount { int64_t below; int64_t ncand; int64_t nan; };

__device__ __forceinline__ int ohl_stride(int64_t cnt) { return cnt <= 65536 ? 16 : (cnt <= ((int64_t)1 << 21) ? 64 : 256); }

__global__ __launch_bounds__(256) void k_bar_med_sample(const float *__restrict__ amount, const int64_t *__restrict__ ci,
                                                        const int64_t *__restrict__ list, const int *__restrict__ go,
                                                        float *__restrict__ samp, OhlBracket *__restrict__ brk, float g_scale,
                                                        int64_t base)
{
    if (go && *go == 0) return;
    const int64_t n_list = list[0];
    for (int64_t q = blockIdx.x; q < n_list; q += gridDim.x) {
        const int64_t b = list[1 + q], s0 = ci[b], e = ci[b + 1], start = s0 + 1, cnt = e - s0;
        const int stride = ohl_stride(cnt);
        const int64_t ns = cnt / stride;                              // >= 1024
        float *mine = samp + ((start - base) >> 4);
        for (int64_t j = threadIdx.x; j < ns; j += 256) mine[j] = amount[start + j * stride];
        __syncthreads();
        const int g = (int)(1.9f * g_scale * sqrtf((float)ns)) + 4;
        const int64_t r_lo = ns / 2 - g > 0 ? ns / 2 - g : 0, r_hi = ns / 2 + g < ns - 1 ? ns / 2 + g : ns - 1;
        uint32_t klo, khi;
        bool any_nan;
        med_block_select<false, 256>(mine, 0, ns, r_lo, r_hi, klo, khi, any_nan);
        if (threadIdx.x == 0) brk[q] = OhlBracket{klo, khi, any_nan ? 0 : 1, 0};
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_bar_med_finish(const int64_t *__restrict__ ci, const int64_t *__restrict__ list,
                                                        const int *__restrict__ go, const uint32_t *__restrict__ cand,
                                                        const OhlCount *__restrict__ res, int64_t *__restrict__ fallback,
                                                        double *__restrict__ o_median, int64_t base)
{
    if (go && *go == 0) return;
    typedef MedKey<false> MK;
    const int64_t n_list = list[0];
    for (int64_t q = blockIdx.x; q < n_list; q += gridDim.x) {
        const int64_t b = list[1 + q], s0 = ci[b], e = ci[b + 1], start = s0 + 1, cnt = e - s0;
        const OhlCount r = res[q];
        const int64_t k1 = (cnt - 1) >> 1, k2 = cnt >> 1, cap = cnt >> 2;
        if (r.nan) { if (threadIdx.x == 0) o_median[b] = NAN; continue; }             // np.median of a bar with a NaN size
        if (!(r.below <= k1 && k2 < r.below + r.ncand && r.ncand <= cap)) {             // bracket missed: the full radix select
            if (threadIdx.x == 0) fallback[1 + atomicAdd((unsigned long long *)fallback, 1ULL)] = b;
            continue;
        }
        uint32_t v1, v2;
        bool any_nan;
        med_block_select<false, 256>(cand + ((start - base) >> 2), 0, r.ncand, k1 - r.below, k2 - r.below, v1, v2, any_nan);
        if (threadIdx.x == 0) o_median[b] = (cnt & 1) ? MK::value(v1) : (MK::value(v1) + MK::value(v2)) / 2.0;
        __syncthreads();
    }
}

// the wide bars of a call: list them, a workgroup per listed bar (two workgroups of 1024 threads per CU).  *median_done = 1 when the
// median of these bars was taken by the same pass (float32 amounts: sample bracket, see above) -- fmk_median_launch then skips them.
int fmk_median_long_list_launch(fmk_ctx *ctx, const void *d_amount, const int64_t *d_close_idx, const int64_t *d_list, double *d_median);

// listed bars: count, first tick and one past the last tick of the span they cover (one block)
__global__ __launch_bounds__(256) void k_list_span(const int64_t *__restrict__ ci, const int64_t *__restrict__ list, int64_t *__restrict__ out)
{
    __shared__ long long s_lo, s_hi;
    if (threadIdx.x == 0) { s_lo = INT64_MAX; s_hi = 0; }
    __syncthreads();
    const int64_t n_list = list[0];
    long long lo = INT64_MAX, hi = 0;
    for (int64_t q = threadIdx.x; q < n_list; q += 256) {
        const int64_t b = list[1 + q];
        const long long s = ci[b] + 1, e = ci[b + 1] + 1;
        lo = s < lo ? s : lo;
        hi = e > hi ? e : hi;
    }
    atomicMin(&s_lo, lo);
    atomicMax(&s_hi, hi);
    __syncthreads();
    if (threadIdx.x == 0) { out[0] = n_list; out[1] = n_list ? s_lo : 0; out[2] = n_list ? s_hi : 0; }
}

template <bool AF64>
static int oh_wide_launch(fmk_ctx *ctx, const double *p, const void *a, const int64_t *ci, int64_t nb, int64_t n, const int *go,
                          const OhlcvOut &o, int *median_done = nullptr, int64_t wide_min = OH_WIDE_MIN)
{
    int64_t *list = nullptr;
    FMK_TRY(fmk_long_bar_list(ctx, ci, nb, n, wide_min, go, &list));
    if constexpr (!AF64) {
        if (o.median && median_done) {
            const int64_t cap = n / wide_min + 2;                  // (the list's capacity: fmk_long_bar_list)
            // The scratch is sized from the listed bars: their count and tick span come back in one 24-byte copy (this path is
            // only reached when the stream has bars beyond 8 192 ticks, or in enqueue-only mode).  An empty list ends the call
            // here; in enqueue-only mode nothing may be waited for, so the scratch covers the whole tick axis (1.25 B/tick).
            int64_t base = 0, span = n;
            if (!ctx->enqueue_only) {
                int64_t *d_span = ctx->d_mail + 48, *h_span = ctx->h_mail + 20;
                k_list_span<<<1, 256, 0, ctx->stream>>>(ci, list, d_span);
                FMK_LAUNCH_CHECK(ctx);
                FMK_HIP(ctx, hipMemcpyAsync(h_span, d_span, 24, hipMemcpyDeviceToHost, ctx->stream));
                FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
                if (h_span[0] == 0) {                               // no bar for the workgroup kernels
                    FMK_TRY(fmk_free(ctx, list));
                    *median_done = 1;
                    return FMK_OK;
                }
                base = h_span[1] & ~(int64_t)15;
                span = h_span[2] - base;
            }
            float *samp = nullptr;
            uint32_t *cand = nullptr;
            OhlBracket *brk = nullptr;
            int64_t *res = nullptr, *fallback = nullptr;
            int rc = fmk_alloc(ctx, (size_t)((span >> 4) + 64) * 4, (void **)&samp);
            if (rc == FMK_OK) rc = fmk_alloc(ctx, (size_t)((span >> 2) + 64) * 4, (void **)&cand);
            if (rc == FMK_OK) rc = fmk_alloc(ctx, (size_t)cap * sizeof(OhlBracket), (void **)&brk);
            if (rc == FMK_OK) rc = fmk_alloc(ctx, (size_t)cap * 3 * 8, (void **)&res);
            if (rc == FMK_OK) rc = fmk_alloc(ctx, (size_t)(cap + 1) * 8, (void **)&fallback);
            const bool have_scratch = rc == FMK_OK;
            hipError_t le = hipSuccess;
            if (have_scratch) le = hipMemsetAsync(fallback, 0, 8, ctx->stream);
            if (have_scratch && le == hipSuccess) {
                const char *gv = getenv("FMK_WIDE_MED_GSCALE");      // developer knob (tests): 0 = brackets that usually miss
                k_bar_med_sample<<<(unsigned)(ctx->n_cu * 4), 256, 0, ctx->stream>>>((const float *)a, ci, list, go, samp, brk,
                                                                                   gv ? (float)atof(gv) : 1.0f, base);
                k_bar_ohlcv_wide<false, true><<<(unsigned)(ctx->n_cu * 2), OH_WIDE_THREADS, 0, ctx->stream>>>(
                    p, a, ci, list, go, o, (const uint32_t *)brk, cand, res, base);
                k_bar_med_finish<<<(unsigned)(ctx->n_cu * 4), 256, 0, ctx->stream>>>(ci, list, go, cand, (const OhlCount *)res, fallback,
                                                                                   o.median, base);
                le = hipGetLastError();
                rc = FMK_OK;
                if (le == hipSuccess) rc = fmk_median_long_list_launch(ctx, a, ci, fallback, o.median);
            }
            if (samp) (void)fmk_free(ctx, samp);
            if (cand) (void)fmk_free(ctx, cand);
            if (brk) (void)fmk_free(ctx, brk);
            if (res) (void)fmk_free(ctx, res);
            if (fallback) (void)fmk_free(ctx, fallback);
            if (have_scratch) {
                (void)fmk_free(ctx, list);
                FMK_TRY(rc);
                FMK_HIP(ctx, le);
                *median_done = 1;
                return FMK_OK;
            }
            // no memory for the scratch: the sums by the plain workgroup kernel below, the medians by the radix-select kernels
            // (fmk_median_launch, *median_done stays 0) -- slower, not an error
            if (rc != FMK_E_NOMEM) { (void)fmk_free(ctx, list); FMK_TRY(rc); }
        }
    }
    k_bar_ohlcv_wide<AF64><<<(unsigned)(ctx->n_cu * 2), OH_WIDE_THREADS, 0, ctx->stream>>>(p, a, ci, list, go, o);
    const hipError_t le = hipGetLastError();
    FMK_TRY(fmk_free(ctx, list));
    FMK_HIP(ctx, le);
    return FMK_OK;
}

template <bool AF64>
__global__ __launch_bounds__(256) void k_bar_ohlcv(const double *__restrict__ price,
                                                   const void *__restrict__ amount,
                                                   const int64_t *__restrict__ ci, int64_t nb, int64_t n,
                                                   int64_t min_cnt, const int *__restrict__ go, OhlcvOut o,
                                                   int64_t skip_lo = 0, int64_t skip_hi = 0)
{
    if (go && *go == 0) return;                              // the small-bar kernel saw no long bar
    const int lane = fmk_lane();
    const int wpb = blockDim.x >> 6;
    const int64_t wave0 = (int64_t)blockIdx.x * wpb + fmk_uniform((int)(threadIdx.x >> 6));
    const int64_t nwaves = (int64_t)gridDim.x * wpb;
    auto do_bar = [&](int64_t b, int64_t s, int64_t e) {
        if (e <= s) {
            if (lane == 0) ohlcv_empty(o, b, price, e, n);
            return;
        }
        if (e - s > OH_WIDE_MIN) return;                      // k_bar_ohlcv_wide: a workgroup per bar
        if (e - s > skip_lo && e - s <= skip_hi) return;      // k_bar_ohlcv_mid: a workgroup per bar, median included
        const int64_t start = s + 1;
        double hi = -INFINITY, lo = INFINITY, tv = 0.0, td = 0.0;
        int64_t j = start + lane;
        for (; j + 192 <= e; j += 256) {
            double p0 = price[j], p1 = price[j + 64], p2 = price[j + 128], p3 = price[j + 192];
            double a0 = fmk_amt<AF64>(amount, j), a1 = fmk_amt<AF64>(amount, j + 64);
            double a2 = fmk_amt<AF64>(amount, j + 128), a3 = fmk_amt<AF64>(amount, j + 192);
            hi = fmax(fmax(hi, p0), fmax(p1, fmax(p2, p3)));
            lo = fmin(fmin(lo, p0), fmin(p1, fmin(p2, p3)));
            tv += a0; td += p0 * a0;
            tv += a1; td += p1 * a1;
            tv += a2; td += p2 * a2;
            tv += a3; td += p3 * a3;
        }
        for (; j <= e; j += 64) {
            double p0 = price[j];
            double a0 = fmk_amt<AF64>(amount, j);
            hi = fmax(hi, p0);
            lo = fmin(lo, p0);
            tv += a0; td += p0 * a0;
        }
        ohlcv_finish<AF64>(o, b, price, start, e, hi, lo, tv, td, lane);
    };
    if (min_cnt > 0) {
        // the leftover pass of a small-bar kernel: 64 bars per step, one coalesced load of their close indices, then only the
        // bars it left (longer than min_cnt) get the wave (walking the bars one by one cost a dependent load per bar)
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

// One bar of <= 64*NCH ticks.  EXACT: the bar has exactly NCH chunks, so chunks 0..NCH-2 are full and
// need neither index clamping nor predication.
template <bool AF64, int NCH, bool EXACT, bool MEDIAN>
__device__ __forceinline__ void small_bar(const double *__restrict__ price, const void *__restrict__ amount,
                                          int64_t b, int64_t start, int64_t e, int64_t cnt, int lane,
                                          typename MedKey<AF64>::K *buf, const OhlcvOut &o)
{
    typedef MedKey<AF64> MK;
    typedef typename MK::K K;     // raw bit pattern type of one amount
    // ---- issue every load of the bar: branch-free; offsets of possibly-partial chunks are clamped to the
    //      bar's last tick (re-reads one element of a line that is fetched anyway: no extra HBM traffic)
    const double *pb = price + start;                    // wave-uniform bases + 32-bit lane offsets
    const K *ab = (const K *)amount + start;
    const unsigned last = (unsigned)(cnt - 1);
    double p[NCH];
    K araw[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        unsigned idx = (unsigned)(c * 64 + lane);
        if (!EXACT || c == NCH - 1) idx = idx < last ? idx : last;
        p[c] = pb[idx];
        araw[c] = ab[idx];
    }
    // ---- OHLCV accumulation, same per-lane order as k_bar_ohlcv
    double hi = -INFINITY, lo = INFINITY, tv = 0.0, td = 0.0;
    MedBar<AF64, NCH, EXACT> bar;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        double a;
        if constexpr (AF64) a = __longlong_as_double((long long)araw[c]);
        else a = (double)__uint_as_float(araw[c]);
        hi = fmax(hi, p[c]);                             // clamped duplicates cannot change max/min
        lo = fmin(lo, p[c]);
        const double pa = p[c] * a;
        if (!EXACT || c == NCH - 1) {
            const bool valid = (unsigned)(c * 64 + lane) <= last;
            tv += valid ? a : 0.0;
            td += valid ? pa : 0.0;
            if constexpr (MEDIAN) bar.key[c] = valid ? MK::tokey(araw[c]) : MK::MAXK;
        } else {
            tv += a;
            td += pa;
            if constexpr (MEDIAN) bar.key[c] = MK::tokey(araw[c]);
        }
    }
    // A bar of <= 64 ticks gets the SAME sums whichever schedule serves it (a result that depended on the stream's mean bar
    // length would make a sharded run differ from the un-sharded one in the last bit): both this kernel and k_bar_ohlcv_lanes add
    // the 64 tick slots (0.0 beyond the bar) as the balanced binary tree of fmk_dpp_reduce -- ((x0+x1)+(x2+x3))+... in tick
    // order.  (A first version made both add in the reference's sequential order; here that was a 64-step readlane loop per bar,
    // 27 ms per 1e9 ticks of 60-tick bars.)
    ohlcv_finish<AF64>(o, b, price, start, e, hi, lo, tv, td, lane);
    if constexpr (MEDIAN) {
        bar.amount = amount; bar.start = start; bar.cnt = cnt; bar.lane = lane;
        const double m = med_search<AF64, NCH, EXACT>(bar, buf);
        if (lane == 0) o.median[b] = m;
    }
}

// MAXNCH: the longest bar the instantiation serves, in 64-tick chunks.  FMK_SMALL_NCH (21): the 1-minute-bar kernel, held to
// four waves per SIMD by the 21-chunk class' registers.  4: streams of 65..256-tick bars -- one wave per bar pays a full memory
// round trip per bar, so what matters there is how many bars a CU has in flight: without the long classes the kernel fits eight
// waves per SIMD (1e9 ticks, ohlcv + median: 80-tick bars 12.0 -> 9.0 ms, 100-tick 9.9 -> 7.3, 200-tick 5.3 -> 4.1; equal at 240).
template <bool AF64, bool MEDIAN, int MAXNCH = FMK_SMALL_NCH>
__global__ __launch_bounds__(256, (MAXNCH <= 4 ? 8 : MAXNCH <= 10 ? 6 : 4)) void k_bar_ohlcv_small(const double *__restrict__ price,
                                                            const void *__restrict__ amount,
                                                            const int64_t *__restrict__ ci, int64_t nb, int64_t n,
                                                            int *__restrict__ saw_long, OhlcvOut o)
{
    typedef typename MedKey<AF64>::K K;
    __shared__ K sbuf[4][64];
    const int lane = fmk_lane();
    const int wib = fmk_uniform((int)(threadIdx.x >> 6));
    const int wpb = blockDim.x >> 6;
    const int64_t wave0 = (int64_t)blockIdx.x * wpb + wib;
    const int64_t nwaves = (int64_t)gridDim.x * wpb;
    K *buf = sbuf[wib];
    for (int64_t b = wave0; b < nb; b += nwaves) {
        const int64_t s = fmk_uniform(ci[b]);
        const int64_t e = fmk_uniform(ci[b + 1]);
        const int64_t cnt = e - s;
        if (cnt > 64 * MAXNCH) {                             // long bar: left to the generic kernels
            // The flag only ever becomes 1: look first (shared reads do not serialise), store if still clear.  An
            // atomicOr per long bar serialises on the one address (measured: 100 000 long bars -> 1.13 ms in this
            // otherwise idle kernel, 11 ns each).  No state is kept across bars: a wave-uniform "already raised" bit
            // made the register allocator park scalars in VGPR lanes throughout the small_bar bodies.
            if (lane == 0 && __hip_atomic_load(saw_long, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                __hip_atomic_store(saw_long, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        if (cnt <= 0) {
            if (lane == 0) ohlcv_empty(o, b, price, e, n);
            continue;
        }
        const int64_t start = s + 1;
        const int nch = (int)((cnt + 63) >> 6);
        // exact-size code for the chunk counts a ~1200-tick (1-minute) bar takes, size classes below
        if constexpr (MAXNCH <= 4) {
            switch (nch) {
            case 1: small_bar<AF64, 1, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            case 2: small_bar<AF64, 2, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            case 3: small_bar<AF64, 3, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            default: small_bar<AF64, 4, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            }
        } else if constexpr (MAXNCH <= 10) {
            // exact-size code for every chunk count (full chunks need neither clamping nor predication)
            switch (nch) {
            case 1: small_bar<AF64, 1, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            case 2: small_bar<AF64, 2, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            case 3: small_bar<AF64, 3, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            case 4: small_bar<AF64, 4, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            case 5: small_bar<AF64, 5, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            case 6: small_bar<AF64, 6, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            case 7: small_bar<AF64, 7, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            case 8: small_bar<AF64, 8, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            case 9: small_bar<AF64, 9, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            default: small_bar<AF64, 10, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            }
        } else {
            switch (nch) {
            case 11: small_bar<AF64, 11, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            case 12: small_bar<AF64, 12, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            case 13: small_bar<AF64, 13, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            case 14: small_bar<AF64, 14, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            case 15: small_bar<AF64, 15, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            case 16: small_bar<AF64, 16, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            case 17: small_bar<AF64, 17, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            case 18: small_bar<AF64, 18, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            case 19: small_bar<AF64, 19, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            case 20: small_bar<AF64, 20, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            case 21: small_bar<AF64, 21, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o); break;
            default:
                if (nch <= 1) small_bar<AF64, 1, true, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o);
                else if (nch <= 4) small_bar<AF64, 4, false, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o);
                else if (nch <= 10) small_bar<AF64, 10, false, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o);
                else small_bar<AF64, 16, false, MEDIAN>(price, amount, b, start, e, cnt, lane, buf, o);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The time-bar step in one call (fmk_time_bars_ohlcv_dev): what the indexer stages need to know.  (Round 4 also had the clock-edge
// search INSIDE the OHLCV kernel -- k_time_bars_ohlcv, behind an environment switch -- one launch instead of two and measured
// slower than the pipelined indexer below: profiles/r04_indexer.txt; removed in round 6.)
// ---------------------------------------------------------------------------------------------------------------------
struct TbFuse {
    const int64_t *ts;
    int64_t e0, d;                  // clock: edge k = e0 + k * d (fmk_time_bar_clock)
    int64_t t_first, t_last;        // ts[0], ts[n - 1] (the caller has them: they define the clock)
    int64_t *clock, *idx;           // [nb + 1] outputs
};


// ---------------------------------------------------------------------------------------------------------------------
// The median trade size ALONE for bars of <= 64 * 21 ticks (float32 amounts): k_bar_ohlcv_small's amount loads, keys and
// order-statistic search without the price column (4 B/tick).  For cfg 4's first half the order-flow kernel k_bar_dir_lanes walks
// price / amount / side anyway and takes open / high / low / close / volume / vwap along for six instructions per tick; what it cannot
// do in one lane is the median of 1 200 amounts.  Longer bars: k_bar_median through the flag, as after k_bar_ohlcv_small.
// Round 6: the order-statistic search is the carried-bracket one of the one-pass cfg 4 kernels (fmk_median.h: fu_median) -- the wave
// keeps a bracket of keys around its previous bar's middle, one sweep of 2 NCH compares usually proves both middle ranks inside, the
// candidates go to one key per lane and the bisection finishes on that register -- instead of a bisection of the whole key range and a
// 21-stage cross-lane sort per bar: np.median's bits either way.
template <int NCH>
__device__ __forceinline__ void median_bar(const void *__restrict__ amount, int64_t b, int64_t start, int64_t cnt, int lane,
                                           uint32_t *buf, double *__restrict__ o_median, FuMed &med)
{
    typedef MedKey<false> MK;
    const uint32_t *ab = (const uint32_t *)amount + start;
    const unsigned last = (unsigned)(cnt - 1);
    uint32_t key[NCH];
    bool nan = false;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        unsigned idx = (unsigned)(c * 64 + lane);
        if (c == NCH - 1) idx = idx < last ? idx : last;
        const uint32_t raw = ab[idx];
        nan |= (raw & 0x7FFFFFFFu) > 0x7F800000u;                     // (a clamped duplicate of the last tick cannot add a NaN that is not there)
        key[c] = (c < NCH - 1 || (unsigned)(c * 64 + lane) <= last) ? MK::tokey(raw) : MK::MAXK;
    }
    const bool any_nan = __ballot(nan) != 0;
    double m;
    if (any_nan) { m = NAN; med.have = 0; }                           // np.median of a bar with a NaN amount
    else m = fu_median<NCH>(key, (int)cnt, lane, false, med, buf);
    if (lane == 0) o_median[b] = m;
}

__global__ __launch_bounds__(256) void k_bar_median_small(const float *__restrict__ amount, const int64_t *__restrict__ ci,
                                                          int64_t nb, int *__restrict__ saw_long, double *__restrict__ o_median)
{
    __shared__ uint32_t sbuf[4][64];
    const int lane = fmk_lane();
    const int wib = fmk_uniform((int)(threadIdx.x >> 6));
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wib;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    uint32_t *buf = sbuf[wib];
    FuMed med;
    med.lo = med.hi = 0; med.width = 64; med.have = 0;
    for (int64_t b = wave0; b < nb; b += nwaves) {
        const int64_t s = fmk_uniform(ci[b]);
        const int64_t e = fmk_uniform(ci[b + 1]);
        const int64_t cnt = e - s;
        if (cnt > 64 * FMK_SMALL_NCH) {
            if (lane == 0 && __hip_atomic_load(saw_long, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                __hip_atomic_store(saw_long, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        if (cnt <= 0) { if (lane == 0) o_median[b] = 0.0; continue; }     // base.py:352-361
        const int64_t start = s + 1;
        switch ((int)((cnt + 63) >> 6)) {
#define FMK_MB(N) case N: median_bar<N>(amount, b, start, cnt, lane, buf, o_median, med); break;
            FMK_MB(1) FMK_MB(2) FMK_MB(3) FMK_MB(4) FMK_MB(5) FMK_MB(6) FMK_MB(7) FMK_MB(8) FMK_MB(9) FMK_MB(10) FMK_MB(11)
            FMK_MB(12) FMK_MB(13) FMK_MB(14) FMK_MB(15) FMK_MB(16) FMK_MB(17) FMK_MB(18) FMK_MB(19) FMK_MB(20)
#undef FMK_MB
        default: median_bar<21>(amount, b, start, cnt, lane, buf, o_median, med); break;
        }
    }
}

// median trade size of every bar, float32 amounts: the small-bar kernel + k_bar_median for the bars beyond its classes
int fmk_median_small_launch(fmk_ctx *ctx, const float *d_amount, const int64_t *d_close_idx, int64_t nb, double *d_median,
                            int64_t n_ticks)
{
    int *saw_long = (int *)(ctx->d_mail + 16);
    FMK_HIP(ctx, hipMemsetAsync(saw_long, 0, sizeof(int), ctx->stream));
    int64_t blocks = fmk_ceil_div(nb, 4);
    const int64_t cap = (int64_t)ctx->n_cu * 64;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    k_bar_median_small<<<(unsigned)blocks, 256, 0, ctx->stream>>>(d_amount, d_close_idx, nb, saw_long, d_median);
    FMK_LAUNCH_CHECK(ctx);
    return fmk_median_launch(ctx, d_amount, 0, d_close_idx, nb, 64 * FMK_SMALL_NCH, saw_long, d_median, n_ticks);
}

// ---------------------------------------------------------------------------------------------------------------------
// Short bars (float32 amounts), one LANE per bar.  The reference's other caller builds 1-SECOND bars (AddTimeBarH5,
// bar/io.py:484-485: ~20 ticks per bar on the SURVEY 8(d) stream).  One wave per bar then uses 20 of 64 lanes and pays four
// cross-lane butterflies plus a 64-lane sort per bar: 46 ms per 1e9 ticks, 0.34 TB/s.  A first packed schedule (whole bars
// side by side in the 64 lanes, one tick per lane, a segmented DPP scan of four values and one sort of (segment, key)
// pairs) got 12.4 ms but still spent ~4.7 wave instructions per tick (profiles/r02_short_bars_before.txt).  This one spends
// ~1: a wave takes the next <= 64 whole bars whose ticks fit its LDS tile; the tile is filled with coalesced loads (the bars
// are one contiguous tick range); then lane l walks bar l's ticks in the tile -- max / min / sum(vol) / sum(price*vol) are
// plain sequential register updates IN THE REFERENCE'S TICK ORDER (base.py:377-391: vwap comes out bit-identical, not just
// within 1e-9) -- and sorts the bar's keys with a fixed compare-exchange network on its own registers (all lanes in
// lockstep, two instructions per exchange, no cross-lane traffic at all).  Outputs are written coalesced, one bar per lane.
// Bars longer than 64 ticks are left to the generic kernels (flag `saw_long`).

// bitonic sorting network on N registers of ONE lane; every index is a template constant, so the keys stay in VGPRs
template <int I, int J, int K, int N>
__device__ __forceinline__ void lb_ce(uint32_t (&r)[N])
{
    constexpr int l = I ^ J;
    if constexpr (l > I) {
        const uint32_t a = r[I], b = r[l];
        const uint32_t mn = a < b ? a : b, mx = a < b ? b : a;
        if constexpr ((I & K) == 0) { r[I] = mn; r[l] = mx; }
        else { r[I] = mx; r[l] = mn; }
    }
}
template <int J, int K, int N, int... I>
__device__ __forceinline__ void lb_stage(uint32_t (&r)[N], std::integer_sequence<int, I...>)
{
    (lb_ce<I, J, K, N>(r), ...);
}
template <int J, int K, int N>
__device__ __forceinline__ void lb_js(uint32_t (&r)[N])
{
    lb_stage<J, K, N>(r, std::make_integer_sequence<int, N>{});
    if constexpr (J > 1) lb_js<J / 2, K, N>(r);
}
template <int K, int N>
__device__ __forceinline__ void lb_ks(uint32_t (&r)[N])
{
    lb_js<K / 2, K, N>(r);
    if constexpr (K < N) lb_ks<K * 2, N>(r);
}
template <int N>
__device__ __forceinline__ void lb_sort(uint32_t (&r)[N]) { lb_ks<2, N>(r); }

template <int N, int... I>
__device__ __forceinline__ uint32_t lb_pick_seq(const uint32_t (&r)[N], int idx, std::integer_sequence<int, I...>)
{
    uint32_t v = r[0];
    ((v = idx == I ? r[I] : v), ...);
    return v;
}
template <int N>
__device__ __forceinline__ uint32_t lb_pick(const uint32_t (&r)[N], int idx)
{
    return lb_pick_seq<N>(r, idx, std::make_integer_sequence<int, N>{});
}

// Block K (ticks 8K .. 8K+7) of a lane's bar: high / low, the eight tick slots' contribution to the two sums and the median
// keys.  The sums are the balanced binary tree over the N tick slots in tick order (0.0 beyond the bar) -- the combining order
// of fmk_dpp_reduce, which serves the same bar in the wave-per-bar kernels (small_bar<NCH = 1>): a tree over the block's 8
// slots here, then a binary counter over the blocks (lv / ld: partial sums of 8, 16, 32 slots waiting for their right halves).
// K is a template constant, so the counter is resolved at compile time.
template <bool MEDIAN, int N, int K, int NR>
__device__ __forceinline__ void lb_block(const double *tp, const uint32_t *ta, int off, int L, int Lmax, int zero_at, double &hi, double &lo,
                                         double (&lv)[3], double (&ld)[3], double &tv, double &td, uint32_t (&r)[NR])
{
    typedef MedKey<false> MK;
    constexpr int j0 = 8 * K;
    double cv = 0.0, cd = 0.0;
    // a block no bar of the wave reaches is skipped (0.0 contributions: x + 0.0 == x, the tree is unchanged).  The wave-uniform
    // branch is also what keeps the blocks apart: as straight-line code the eight blocks were scheduled into one another and
    // the kernel needed > 256 VGPRs.
    if (j0 < Lmax) {
        double p[8];
        uint32_t raw[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int at = j0 + q < L ? off + j0 + q : zero_at;  // idle iterations read the tile's zero slot: price 0, amount 0
            p[q] = tp[at];
            raw[q] = ta[at];
        }
        double xv[8], xd[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool in = j0 + q < L;
            const double a = (double)__uint_as_float(raw[q]);
            if (in) {
                hi = p[q] > hi ? p[q] : hi;                      // `>` / `<` like the reference: a NaN never wins
                lo = p[q] < lo ? p[q] : lo;
            }
            xv[q] = a;                                           // 0.0 and 0.0 * 0.0 for the idle slots: no selects
            xd[q] = p[q] * a;
            if constexpr (MEDIAN) r[j0 + q] = in ? MK::tokey(raw[q]) : MK::MAXK;
        }
        cv = ((xv[0] + xv[1]) + (xv[2] + xv[3])) + ((xv[4] + xv[5]) + (xv[6] + xv[7]));
        cd = ((xd[0] + xd[1]) + (xd[2] + xd[3])) + ((xd[4] + xd[5]) + (xd[6] + xd[7]));
    } else if constexpr (MEDIAN) {
#pragma unroll
        for (int q = 0; q < 8; ++q) r[j0 + q] = MK::MAXK;
    }
    constexpr int NB = N / 8;
    if constexpr ((K & 1) == 0 && NB > 1) { lv[0] = cv; ld[0] = cd; }
    else {
        if constexpr (NB > 1) { cv = lv[0] + cv; cd = ld[0] + cd; }
        if constexpr ((K & 2) == 0 && NB > 2) { lv[1] = cv; ld[1] = cd; }
        else {
            if constexpr (NB > 2) { cv = lv[1] + cv; cd = ld[1] + cd; }
            if constexpr ((K & 4) == 0 && NB > 4) { lv[2] = cv; ld[2] = cd; }
            else {
                if constexpr (NB > 4) { cv = lv[2] + cv; cd = ld[2] + cd; }
                tv = cv; td = cd;                                // K == NB - 1: the root
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);                           // keep the blocks apart: 8 ticks of loads live at a time
    asm volatile("" ::: "memory");                               // (the tree sums made the blocks independent: without a compiler
                                                                 //  barrier all 64 ticks were loaded up front, 256 VGPRs)
    if constexpr (K + 1 < NB) lb_block<MEDIAN, N, K + 1, NR>(tp, ta, off, L, Lmax, zero_at, hi, lo, lv, ld, tv, td, r);
}

// One bar of L <= N ticks per lane (L == 0: idle lane): the reference's loop body (base.py:377-391) over the lane's slice of
// the tile, eight ticks at a time (more loads in flight only cost registers: the LDS is next door), then the median from a
// fixed N-key sorting network on the lane's own registers.
template <bool MEDIAN, int N>
__device__ __forceinline__ void lb_bar(const double *tp, const uint32_t *ta, int off, int L, int zero_at, double &hi, double &lo,
                                       double &tv, double &td, double &med)
{
    typedef MedKey<false> MK;
    uint32_t r[MEDIAN ? N : 1];
    double lv[3] = {0.0, 0.0, 0.0}, ld[3] = {0.0, 0.0, 0.0};
    const int Lmax = fmk_dpp_reduce(L, 0, FmkOpMax());               // wave-uniform: the longest bar of the wave
    lb_block<MEDIAN, N, 0, (MEDIAN ? N : 1)>(tp, ta, off, L, Lmax, zero_at, hi, lo, lv, ld, tv, td, r);
    if constexpr (MEDIAN) {
        lb_sort<N>(r);
        const uint32_t v1 = lb_pick<N>(r, (L - 1) >> 1), v2 = lb_pick<N>(r, L >> 1);
        const uint32_t kmx = lb_pick<N>(r, L > 0 ? L - 1 : 0), kmn = r[0];
        if (kmn < MK::KEY_NEG_INF || kmx > MK::KEY_POS_INF) med = NAN;   // a NaN amount: np.median is NaN
        else med = (L & 1) ? MK::value(v1) : (MK::value(v1) + MK::value(v2)) / 2.0;
    }
}

// LB_TILE ticks per wave tile (12 B each).  1024 (two waves: 24 KB per workgroup) beat 2048 at every bar length from 10 to 40
// ticks (4.5 vs 4.8 .. 7.6 ms per 1e9 ticks): the waves a CU can hold matter more than filling all 64 lanes
template <bool MEDIAN, int LB_TILE>
__global__ __launch_bounds__(128) void k_bar_ohlcv_lanes(const double *__restrict__ price, const float *__restrict__ amount,
                                                         const int64_t *__restrict__ ci, int64_t nb, int64_t n,
                                                         int *__restrict__ saw_long, OhlcvOut o)
{
    __shared__ double s_p[2][LB_TILE + 2];                            // [LB_TILE]: the zero slot of idle iterations
    __shared__ uint32_t s_a[2][LB_TILE + 2];
    __shared__ int64_t s_ci[2][66];
    const int lane = fmk_lane();
    const int w = fmk_uniform((int)(threadIdx.x >> 6));
    double *tp = s_p[w];
    uint32_t *ta = s_a[w];
    if (lane == 0) { tp[LB_TILE] = 0.0; ta[LB_TILE] = 0u; }
    const int64_t ngroups = (nb + 63) >> 6;
    const int64_t nwaves = (int64_t)gridDim.x * 2;
    for (int64_t g = (int64_t)blockIdx.x * 2 + w; g < ngroups; g += nwaves) {
        const int64_t B0 = g * 64;
        const int nbg = (int)(nb - B0 < 64 ? nb - B0 : 64);
        __builtin_amdgcn_wave_barrier();
        if (lane <= nbg) s_ci[w][lane] = ci[B0 + lane];
        if (lane == 0 && nbg == 64) s_ci[w][64] = ci[B0 + 64];
        __builtin_amdgcn_wave_barrier();
        int bl = 0;
        while (bl < nbg) {
            const int64_t s0 = s_ci[w][bl];
            const int idx = bl + 1 + lane;
            const bool valid = idx <= nbg;
            const int64_t e_l = valid ? s_ci[w][idx] : INT64_MAX;
            const int64_t s_l = valid ? s_ci[w][idx - 1] : 0;
            const int m = __popcll(__ballot(valid && e_l - s0 <= LB_TILE));     // closes ascend: a prefix of the lanes
            if (m == 0) {                                                        // one bar longer than the tile
                if (lane == 0 && __hip_atomic_load(saw_long, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                    __hip_atomic_store(saw_long, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bl += 1;
                continue;
            }
            const int ntick = (int)(s_ci[w][bl + m] - s0);                       // ticks of the m bars: (s0, s0 + ntick]
            // ---- fill the tile: one contiguous range, coalesced
            __builtin_amdgcn_wave_barrier();
            const double *gp = price + s0 + 1;
            const uint32_t *ga = (const uint32_t *)amount + s0 + 1;
#pragma unroll 4
            for (int j = lane; j < ntick; j += 64) {
                tp[j] = gp[j];
                ta[j] = ga[j];
            }
            __builtin_amdgcn_wave_barrier();
            // ---- one lane per bar
            const bool owner = lane < m;
            const int L = owner ? (int)(e_l - s_l) : 0;
            const bool mine = owner && L > 0 && L <= 64;
            const int off = mine ? (int)(s_l - s0) : 0;
            const uint64_t longer = __ballot(owner && L > 64);                   // wave-long bars: generic kernels
            if (longer && lane == 0 && __hip_atomic_load(saw_long, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                __hip_atomic_store(saw_long, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int Lw = mine ? L : 0;
            const bool any_gt32 = __ballot(Lw > 32) != 0;
            const double first = ntick > 0 ? tp[off] : 0.0;
            double hi = first, lo = first, tv = 0.0, td = 0.0;                   // base.py:371-372: seeded with the first price
            double med = 0.0;
            if (any_gt32) lb_bar<MEDIAN, 64>(tp, ta, off, Lw, LB_TILE, hi, lo, tv, td, med);
            else lb_bar<MEDIAN, 32>(tp, ta, off, Lw, LB_TILE, hi, lo, tv, td, med);
            if (owner) {
                const int64_t b = B0 + bl + lane;
                if (mine) {
                    o.open[b] = first;
                    o.close[b] = tp[off + L - 1];
                    o.high[b] = hi;
                    o.low[b] = lo;
                    o.vol[b] = (float)tv;
                    o.vwap[b] = tv > 0.0 ? td / tv : 0.0;
                    o.trades[b] = L;
                    if constexpr (MEDIAN) o.median[b] = med;
                } else if (L == 0) {
                    ohlcv_empty(o, b, price, e_l, n);
                }
            }
            bl += m;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// SIXTEEN LANES PER BAR (round 3, float32 amounts): streams of 33 .. 256-tick bars (2-second to 10-second bars), four bars per wave.
// One wave per such bar leaves 44 of 64 lanes idle on a 20-tick chunk and pays a 64-lane reduction tree and a 64-lane selection per
// bar; one LANE per bar (k_bar_ohlcv_lanes) needs a 64-key sorting network in registers: comp_bar_ohlcv with the median took 6.5 /
// 9.4 / 8.1 / 6.6 ms per 1e9 ticks at 40 / 60 / 80 / 100-tick bars against 2.4 ms at 1 200 (profiles/r03_median_humps.txt).
// Here a row of 16 lanes takes a bar: element i sits in lane i % 16, register i / 16 (sixteen coalesced loads per row and column,
// all in flight together).  Sums keep the association of every other schedule -- the 64 "virtual lanes" v = i % 64 each add their
// elements in chunk order, then the balanced tree over v (fmk_dpp_reduce): lane ri carries the virtual lanes ri, ri + 16, ri + 32,
// ri + 48 as four accumulators, fmk_row_sum is the tree inside each virtual row, (T0 + T1) + (T2 + T3) its top -- so a bar gets the
// same bits whichever schedule serves it (a sharded run equals the un-sharded one).  The median: bisection on the key VALUE with
// row-wide counts until ONE key is left in the bracket (~log2(L) + 2 steps), as in k_bar_trade_size_rows.
// Bars of more than 256 ticks are flagged for the generic kernels.
// ---------------------------------------------------------------------------------------------------------------------
#define OHR_WAVES 4
#define OHR_CAND 16                     // keys left in the median bracket when the row sorts them (k_bar_ohlcv_rows)
// The body of k_bar_ohlcv_rows for a wave whose longest bar takes NR registers per lane (NR * 16 >= its ticks): every loop over the
// registers has a compile-time trip count.  With one body for all lengths (loops to 16 guarded by the wave's register count) the kernel
// issued 1 341 VALU instructions per four 80-tick bars, 490 without the median, and is bound by exactly that (rocprofv3 SQ counters:
// 4.19e9 VALU instructions x 4 cycles / 1 024 SIMDs = the 6.5 ms it took; profiles/r04_short_bars.txt).
// LPB lanes per bar: 16 (a DPP row, four bars per wave, <= 256 ticks) or 8 (a half row, eight bars per wave, <= 128 ticks: streams of 33 .. 46
// ticks per bar, where a row leaves half its lanes idle and the lane-per-bar schedule pays a 64-key network per bar).  The sums keep the
// association of every other schedule in both layouts: element i belongs to virtual lane i % 64 = LPB * (r % NA) + ri with NA = 64 / LPB
// accumulators per lane, added in chunk order; the tree over the virtual lanes is the butterfly inside the LPB lanes, then the balanced
// tree over the NA partial sums.
template <int LPB> __device__ __forceinline__ double ohr_sum(double v)
{
    v += fmk_dpp<DPP_XOR1, 0xF>(v, v);
    v += fmk_dpp<DPP_XOR2, 0xF>(v, v);
    if constexpr (LPB >= 8) v += fmk_dpp<DPP_HALF_MIRROR, 0xF>(v, v);
    if constexpr (LPB == 16) v += fmk_dpp<DPP_MIRROR, 0xF>(v, v);
    return v;
}
template <int LPB> __device__ __forceinline__ double ohr_max(double v)
{
    v = fmax(v, fmk_dpp<DPP_XOR1, 0xF>(v, v));
    v = fmax(v, fmk_dpp<DPP_XOR2, 0xF>(v, v));
    if constexpr (LPB >= 8) v = fmax(v, fmk_dpp<DPP_HALF_MIRROR, 0xF>(v, v));
    if constexpr (LPB == 16) v = fmax(v, fmk_dpp<DPP_MIRROR, 0xF>(v, v));
    return v;
}
template <int LPB> __device__ __forceinline__ double ohr_min(double v)
{
    v = fmin(v, fmk_dpp<DPP_XOR1, 0xF>(v, v));
    v = fmin(v, fmk_dpp<DPP_XOR2, 0xF>(v, v));
    if constexpr (LPB >= 8) v = fmin(v, fmk_dpp<DPP_HALF_MIRROR, 0xF>(v, v));
    if constexpr (LPB == 16) v = fmin(v, fmk_dpp<DPP_MIRROR, 0xF>(v, v));
    return v;
}
template <int LPB> __device__ __forceinline__ int ohr_isum(int v)
{
    v += __builtin_amdgcn_update_dpp(v, v, DPP_XOR1, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(v, v, DPP_XOR2, 0xF, 0xF, false);
    if constexpr (LPB >= 8) v += __builtin_amdgcn_update_dpp(v, v, DPP_HALF_MIRROR, 0xF, 0xF, false);
    if constexpr (LPB == 16) v += __builtin_amdgcn_update_dpp(v, v, DPP_MIRROR, 0xF, 0xF, false);
    return v;
}
template <int LPB> __device__ __forceinline__ uint32_t ohr_umin(uint32_t v)
{
    uint32_t q;
    q = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_XOR1, 0xF, 0xF, false); v = q < v ? q : v;
    q = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_XOR2, 0xF, 0xF, false); v = q < v ? q : v;
    if constexpr (LPB >= 8) { q = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_HALF_MIRROR, 0xF, 0xF, false); v = q < v ? q : v; }
    if constexpr (LPB == 16) { q = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_MIRROR, 0xF, 0xF, false); v = q < v ? q : v; }
    return v;
}
template <int LPB> __device__ __forceinline__ uint32_t ohr_umax(uint32_t v)
{
    uint32_t q;
    q = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_XOR1, 0xF, 0xF, false); v = q > v ? q : v;
    q = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_XOR2, 0xF, 0xF, false); v = q > v ? q : v;
    if constexpr (LPB >= 8) { q = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_HALF_MIRROR, 0xF, 0xF, false); v = q > v ? q : v; }
    if constexpr (LPB == 16) { q = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_MIRROR, 0xF, 0xF, false); v = q > v ? q : v; }
    return v;
}
template <bool MEDIAN, int NR, int LPB = 16>
__device__ __forceinline__ void ohr_bars(const double *__restrict__ price, const float *__restrict__ amount, int64_t s_b, int L, bool mine,
                                         int64_t b, int row, int ri, int lane, int w, uint32_t (*s_cand)[64], const OhlcvOut &o, int nreg)
{
    typedef MedKey<false> MK;
    constexpr int NA = 64 / LPB;                  // accumulators per lane = virtual lanes per lane
#define OHR_LIVE(r_) (NR < 16 || (r_) < nreg)      /* NR == 16: the generic body, loops guarded by the wave's register count */
        const double *pb = price + (mine ? s_b + 1 : 0);
        const uint32_t *ab = (const uint32_t *)amount + (mine ? s_b + 1 : 0);
        // ---- every load of the four bars
        double p[NR];
        uint32_t araw[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            p[r] = 0.0; araw[r] = 0u;
            if (OHR_LIVE(r)) {
                const int i = r * LPB + ri;
                if (i < L) { p[r] = pb[i]; araw[r] = ab[i]; }
            }
        }
        const double first = mine ? pb[0] : 0.0, lastp = mine ? pb[L - 1] : 0.0;      // (lines fetched anyway)
        // (without the median the sizes are only used as doubles, and the compiler sinks the conversion into each register's guarded load
        //  block: load, wait, convert, sixteen times -- 4.4 against 3.0 ms at 80-tick bars.  Each raw word is pinned where it is used, in
        //  the accumulation loop behind the last load, so every load is in flight before the first wait.  With the median the pins
        //  stand together right here -- all loads complete, then all arithmetic: 4.56 against 5.32 ms at 80-tick bars, measured)
        if constexpr (MEDIAN) {
#pragma unroll
            for (int r = 0; r < NR; ++r) asm volatile("" : "+v"(araw[r]));
        }
        // ---- accumulation: virtual lane v = LPB * (r % NA) + ri, chunk r / NA
        double hi = -INFINITY, lo = INFINITY, tv[NA], td[NA];
#pragma unroll
        for (int k = 0; k < NA; ++k) { tv[k] = 0.0; td[k] = 0.0; }
        uint32_t key[MEDIAN ? NR : 1];
        uint32_t kmn = MK::MAXK, kmx = 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (OHR_LIVE(r)) {
                const bool valid = r * LPB + ri < L;
                if constexpr (!MEDIAN) asm volatile("" : "+v"(araw[r]));
                const double a = (double)__uint_as_float(araw[r]);
                hi = valid ? fmax(hi, p[r]) : hi;
                lo = valid ? fmin(lo, p[r]) : lo;
                tv[r % NA] += valid ? a : 0.0;
                td[r % NA] += valid ? p[r] * a : 0.0;
                if constexpr (MEDIAN) {
                    key[r] = valid ? MK::tokey(araw[r]) : MK::MAXK;
                    kmn = key[r] < kmn ? key[r] : kmn;
                    kmx = (valid && key[r] > kmx) ? key[r] : kmx;
                }
            } else if constexpr (MEDIAN) key[r] = MK::MAXK;
        }
        hi = ohr_max<LPB>(hi);
        lo = ohr_min<LPB>(lo);
        double T[NA], D[NA];
#pragma unroll
        for (int k = 0; k < NA; ++k) { T[k] = ohr_sum<LPB>(tv[k]); D[k] = ohr_sum<LPB>(td[k]); }
        double tvs, tds;
        {                                                                // the balanced tree over the NA partial sums
#pragma unroll
            for (int h = 1; h < NA; h <<= 1)
#pragma unroll
                for (int k = 0; k < NA; k += 2 * h) { T[k] += T[k + h]; D[k] += D[k + h]; }
            tvs = T[0]; tds = D[0];
        }
        double med = 0.0;
        if constexpr (MEDIAN) {
            kmn = ohr_umin<LPB>(kmn);
            kmx = ohr_umax<LPB>(kmx);
            const int k1 = (L - 1) >> 1, k2 = L >> 1;
            // smallest v with count(key <= v) > k1: invariant c_lo = count(<= lo) <= k1 < count(<= hi) = c_hi
            uint32_t blo = kmn - 1, bhi = kmx;
            int c_lo = 0, c_hi = L;
            for (int step = 0; step < 40; ++step) {
                const bool open = L > 0 && bhi - blo > 1 && c_hi - c_lo > LPB;
                if (__ballot(open) == 0) break;
                if (step == 10 || step == 15 || step == 20) {
                    // still open after ~log2(L) + 3 halvings: the target key is TIED (decimal lot sizes) and the count never falls to
                    // one -- the bracket then goes on halving an empty key range down to one value, 27 steps.  Snap it to the
                    // smallest and largest key actually inside (the counts at both ends stay what they are); all equal: done
                    uint32_t mn_in = MK::MAXK, mx_in = 0;
#pragma unroll
                    for (int r = 0; r < NR; ++r)
                        if (OHR_LIVE(r)) {
                            const bool in = key[r] > blo && key[r] <= bhi;
                            mn_in = (in && key[r] < mn_in) ? key[r] : mn_in;
                            mx_in = (in && key[r] > mx_in) ? key[r] : mx_in;
                        }
                    mn_in = ohr_umin<LPB>(mn_in);
                    mx_in = ohr_umax<LPB>(mx_in);
                    if (open) { bhi = mx_in; blo = (mn_in == mx_in ? mx_in : mn_in) - 1; }
                    continue;
                }
                const uint32_t pivot = blo + ((bhi - blo) >> 1);
                int c = 0;
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    if (OHR_LIVE(r)) c += key[r] <= pivot ? 1 : 0;
                c = ohr_isum<LPB>(c);
                if (open) { if (c > k1) { bhi = pivot; c_hi = c; } else { blo = pivot; c_lo = c; } }
            }
            const bool few = L > 0 && bhi - blo > 1;                         // <= LPB keys left in (blo, bhi]
            const uint32_t bhi_in = bhi;                                     // (count(key <= bhi_in) == c_hi)
            uint32_t v2_sorted = 0;
            if (__ballot(few) != 0) {
                // The bisection on the key VALUE used to run until ONE key was left: ~22 steps of ~25 instructions for the 80 keys of a
                // bar whose sizes span twelve binades.  Now it stops at <= 16 candidates (~6 steps); the row packs them into its sixteen
                // lanes through LDS and sorts them with a ten-step network on the DPP data path; rank k1 - c_lo of the sorted row is v1.
                int mcnt = 0;
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    if (OHR_LIVE(r)) mcnt += (key[r] > blo && key[r] <= bhi) ? 1 : 0;
                int pos = mcnt;
                if constexpr (LPB == 16) {
                    pos += fmk_dpp_i32<FMK_DPP_ROW_SHR(1), 0xF>(0, pos);
                    pos += fmk_dpp_i32<FMK_DPP_ROW_SHR(2), 0xF>(0, pos);
                    pos += fmk_dpp_i32<FMK_DPP_ROW_SHR(4), 0xF>(0, pos);
                    pos += fmk_dpp_i32<FMK_DPP_ROW_SHR(8), 0xF>(0, pos);
                } else {                                                      // (a source lane in the other half row contributes nothing)
                    int t_;
                    t_ = fmk_dpp_i32<FMK_DPP_ROW_SHR(1), 0xF>(0, pos); pos += ri >= 1 ? t_ : 0;
                    t_ = fmk_dpp_i32<FMK_DPP_ROW_SHR(2), 0xF>(0, pos); pos += ri >= 2 ? t_ : 0;
                    if constexpr (LPB == 8) { t_ = fmk_dpp_i32<FMK_DPP_ROW_SHR(4), 0xF>(0, pos); pos += ri >= 4 ? t_ : 0; }
                }
                pos -= mcnt;                                                  // keys of the row's lower lanes that are in the bracket
                s_cand[w][lane] = MK::MAXK;
                __builtin_amdgcn_wave_barrier();
                if (few) {
#pragma unroll
                    for (int r = 0; r < NR; ++r)
                        if (OHR_LIVE(r) && key[r] > blo && key[r] <= bhi) s_cand[w][row * LPB + (pos++ & (LPB - 1))] = key[r];
                }
                __builtin_amdgcn_wave_barrier();
                uint32_t v = s_cand[w][lane];
                __builtin_amdgcn_wave_barrier();
                // sorting network over the 16 lanes of a row, every exchange ascending: merge of size 2, 4, 8, 16 = one "flip" (i <-> size-1-i)
                // and then half-cleaners at distance size/4 .. 1
#define OHR_CE(PARTNER, LOWER) { const uint32_t q_ = (uint32_t)(PARTNER); const uint32_t mn_ = q_ < v ? q_ : v, mx_ = q_ < v ? v : q_; v = (LOWER) ? mn_ : mx_; }
                OHR_CE(__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_XOR1, 0xF, 0xF, false), (ri & 1) == 0)
                OHR_CE(__builtin_amdgcn_update_dpp((int)v, (int)v, 0x1B /* quad_perm [3,2,1,0] */, 0xF, 0xF, false), (ri & 2) == 0)
                OHR_CE(__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_XOR1, 0xF, 0xF, false), (ri & 1) == 0)
                if constexpr (LPB >= 8) {
                    OHR_CE(__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_HALF_MIRROR, 0xF, 0xF, false), (ri & 4) == 0)
                    OHR_CE(__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_XOR2, 0xF, 0xF, false), (ri & 2) == 0)
                    OHR_CE(__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_XOR1, 0xF, 0xF, false), (ri & 1) == 0)
                }
                if constexpr (LPB == 16) {
                    OHR_CE(__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_MIRROR, 0xF, 0xF, false), (ri & 8) == 0)
                    OHR_CE(__shfl_xor((int)v, 4, 64), (ri & 4) == 0)
                    OHR_CE(__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_XOR2, 0xF, 0xF, false), (ri & 2) == 0)
                    OHR_CE(__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_XOR1, 0xF, 0xF, false), (ri & 1) == 0)
                }
#undef OHR_CE
                const int want = few ? k1 - c_lo : 0;                         // 0 <= k1 - c_lo < c_hi - c_lo <= LPB
                const uint32_t got = (uint32_t)__shfl((int)v, row * LPB + (want & (LPB - 1)), 64);
                v2_sorted = (uint32_t)__shfl((int)v, row * LPB + ((want + 1) & (LPB - 1)), 64);      // rank k1 + 1, used when it is inside the bracket
                if (few) bhi = got;
            }
            const uint32_t v1 = bhi;
            // rank k2 (= k1 or k1 + 1): inside the bracket it is the sorted row's next entry (or the same tied key); only when k1 was
            // the bracket's LAST rank is it the smallest key above the bracket -- a scan over the registers, taken by the whole wave
            // when one of its four bars needs it
            uint32_t v2 = v1;
            const bool beyond = L > 0 && k2 != k1 && k2 >= c_hi;
            if (L > 0 && k2 != k1 && !beyond && few) v2 = v2_sorted;
            if (__ballot(beyond) != 0) {
                uint32_t nxt = MK::MAXK;
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    if (OHR_LIVE(r)) nxt = (key[r] > bhi_in && key[r] < nxt) ? key[r] : nxt;
                nxt = ohr_umin<LPB>(nxt);
                if (beyond) v2 = nxt;
            }
            if (kmn < MK::KEY_NEG_INF || kmx > MK::KEY_POS_INF) med = NAN;   // a NaN amount: np.median is NaN
            else med = (L & 1) ? MK::value(v1) : (MK::value(v1) + MK::value(v2)) / 2.0;
        }
        if (mine && ri == 0) {
            o.open[b] = first;
            o.close[b] = lastp;
            o.high[b] = first != first ? first : hi;                         // base.py:371-382: a NaN FIRST price never loses
            o.low[b] = first != first ? first : lo;
            o.vol[b] = (float)tvs;
            o.vwap[b] = tvs > 0.0 ? tds / tvs : 0.0;                         // base.py:398
            o.trades[b] = L;
            if constexpr (MEDIAN) o.median[b] = med;
        }
#undef OHR_LIVE
}

template <bool MEDIAN, int LPB = 16>
__global__ __launch_bounds__(64 * OHR_WAVES) void k_bar_ohlcv_rows(const double *__restrict__ price, const float *__restrict__ amount,
                                                                 const int64_t *__restrict__ ci, int64_t nb, int64_t n,
                                                                 int *__restrict__ saw_long, OhlcvOut o)
{
    __shared__ uint32_t s_cand[OHR_WAVES][64];
    const int lane = fmk_lane();
    const int w = fmk_uniform((int)(threadIdx.x >> 6));
    const int row = lane / LPB, ri = lane % LPB;
    constexpr int BPW = 64 / LPB, LMAX = 16 * LPB;      // bars per wave, longest bar served
    const int64_t niter = (nb + BPW - 1) / BPW;
    const int64_t nwaves = (int64_t)gridDim.x * OHR_WAVES;
    for (int64_t it = (int64_t)blockIdx.x * OHR_WAVES + w; it < niter; it += nwaves) {
        const int64_t b = BPW * it + row;
        const bool have = b < nb;
        const int64_t s_b = have ? ci[b] : 0, e_b = have ? ci[b + 1] : 0;
        const int64_t len_b = e_b - s_b;
        const bool is_long = have && len_b > LMAX;
        if (__ballot(is_long) != 0 && lane == 0 && __hip_atomic_load(saw_long, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
            __hip_atomic_store(saw_long, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool mine = have && len_b >= 1 && len_b <= LMAX;
        if (have && len_b <= 0 && ri == 0) ohlcv_empty(o, b, price, e_b, n);          // base.py:352-361
        if (__ballot(mine) == 0) continue;
        const int L = mine ? (int)len_b : 0;
        const int nreg = (fmk_dpp_reduce(L, 0, FmkOpMax()) + LPB - 1) / LPB;               // wave-uniform
        // ---- the body specialised for the register count of the wave's longest bar
        // (without the median the one generic body is the faster kernel -- 3.0 against 4.2 ms at 80-tick bars with nine bodies: the reducer alone
        //  is not bound by VALU issue, and the code of nine bodies does not stay in the instruction cache)
        if constexpr (MEDIAN) {
#define OHR_CASE(NR_) if (nreg <= NR_) { ohr_bars<MEDIAN, NR_, LPB>(price, amount, s_b, L, mine, b, row, ri, lane, w, s_cand, o, nreg); continue; }
            OHR_CASE(4) OHR_CASE(5) OHR_CASE(6) OHR_CASE(7) OHR_CASE(8) OHR_CASE(10) OHR_CASE(12)
#undef OHR_CASE
        }
        ohr_bars<MEDIAN, 16, LPB>(price, amount, s_b, L, mine, b, row, ri, lane, w, s_cand, o, nreg);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Bars of 1 345 .. 8 192 ticks (float32 amounts; round 3): ONE pass by a workgroup of 256 threads -- comp_bar_ohlcv AND the median.
// Such bars (75-second, 2-minute, 5-minute bars on the bench tape) used to cost two to three passes: the generic streaming kernel
// for open / high / low / close / volume / vwap, then k_bar_median's wave-per-bar register classes (<= 2 048 ticks) or
// k_bar_median_long's three radix passes with LDS-atomic histograms -- 5.5 .. 7.7 ms per 1e9 ticks against 2.4 ms at 1 200-tick
// bars (profiles/r03_median_humps.txt).  Here every thread keeps the keys of its <= 32 ticks in registers while the sums go by, and
// the two middle ranks come from a bisection on the key VALUE range with block-wide counts (a ballot popcount per register and
// wave, the four wave counts through LDS: one barrier per step) down to <= 64 candidates, which one wave sorts.
// Sums: thread t adds its elements t, t + 256, ... in order, wave DPP trees, then the four wave totals in order -- volume is exact
// in any order for float32 amounts, vwap a reassociated sum (<= 1e-9), like k_bar_ohlcv_wide.  The bars come from a list made from
// their own lengths, so a bar takes this path whatever stream it is part of (a sharded run equals the un-sharded one).
// ---------------------------------------------------------------------------------------------------------------------
#define OHM_MIN (64 * FMK_SMALL_NCH)
#define OHM_MAX 16384                   // beyond: the one-pass wide kernel with the sample-bracket median (1024-thread workgroups
                                        // holding 32 / 64 keys per thread were measured for 16 385 .. 65 536 ticks: 6.0 .. 9.1 ms)
template <int NW>
struct OhmShared {
    double red[4][NW];
    uint32_t kmn[NW], kmx[NW];
    int cnt[2][NW];
    uint32_t below[NW], above[NW];
    int ncand[NW];
    uint32_t buf[64];
};

template <bool MEDIAN, int NREG, int THREADS>
__device__ __forceinline__ void ohm_bar(const double *__restrict__ price, const float *__restrict__ amount, int64_t b, int64_t s,
                                        int64_t e, const OhlcvOut &o, OhmShared<THREADS / 64> &sh)
{
    constexpr int NW = THREADS / 64;
    typedef MedKey<false> MK;
    const int tid = threadIdx.x, lane = fmk_lane(), w = tid >> 6;
    const int64_t start = s + 1;
    const int cnt = (int)(e - s);
    const double *pb = price + start;
    const uint32_t *ab = (const uint32_t *)amount + start;
    const unsigned last = (unsigned)(cnt - 1);
    double hi = -INFINITY, lo = INFINITY, tv = 0.0, td = 0.0;
    uint32_t key[MEDIAN ? NREG : 1];
    uint32_t kmn = MK::MAXK, kmx = 0;
#pragma unroll
    for (int r0 = 0; r0 < NREG; r0 += 4) {                           // four loads of each column in flight
        double p[4];
        uint32_t araw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned idx = (unsigned)((r0 + q) * THREADS + tid);
            idx = idx < last ? idx : last;
            p[q] = pb[idx];
            araw[q] = ab[idx];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool valid = (unsigned)((r0 + q) * THREADS + tid) <= last;
            const double a = (double)__uint_as_float(araw[q]);
            hi = fmax(hi, p[q]);                                     // clamped duplicates cannot change max / min
            lo = fmin(lo, p[q]);
            tv += valid ? a : 0.0;
            td += valid ? p[q] * a : 0.0;
            if constexpr (MEDIAN) {
                const uint32_t k = valid ? MK::tokey(araw[q]) : MK::MAXK;
                key[r0 + q] = k;
                kmn = k < kmn ? k : kmn;
                kmx = (valid && k > kmx) ? k : kmx;
            }
        }
    }
    hi = fmk_dpp_reduce(hi, (double)-INFINITY, FmkOpMax());
    lo = fmk_dpp_reduce(lo, (double)INFINITY, FmkOpMin());
    tv = fmk_dpp_reduce(tv, 0.0, FmkOpAdd());
    td = fmk_dpp_reduce(td, 0.0, FmkOpAdd());
    if constexpr (MEDIAN) { kmn = med_wave_umin<uint32_t>(kmn); kmx = med_wave_umax<uint32_t>(kmx); }
    __syncthreads();                                                 // (the previous bar's shared values have been read)
    if (lane == 0) {
        sh.red[0][w] = hi; sh.red[1][w] = lo; sh.red[2][w] = tv; sh.red[3][w] = td;
        if constexpr (MEDIAN) { sh.kmn[w] = kmn; sh.kmx[w] = kmx; }
    }
    __syncthreads();
    if (w == 0) {
        hi = sh.red[0][0]; lo = sh.red[1][0]; tv = sh.red[2][0]; td = sh.red[3][0];
#pragma unroll
        for (int k = 1; k < NW; ++k) {                                   // the wave totals in order
            hi = fmax(hi, sh.red[0][k]); lo = fmin(lo, sh.red[1][k]); tv += sh.red[2][k]; td += sh.red[3][k];
        }
        ohlcv_finish<false>(o, b, price, start, e, hi, lo, tv, td, lane, true);
    }
    if constexpr (MEDIAN) {
        // ---- the two middle ranks: every thread walks the same (block-uniform) bisection
        uint32_t mn = sh.kmn[0], mx = sh.kmx[0];
#pragma unroll
        for (int k = 1; k < NW; ++k) { mn = sh.kmn[k] < mn ? sh.kmn[k] : mn; mx = sh.kmx[k] > mx ? sh.kmx[k] : mx; }
        const int k1 = (cnt - 1) >> 1, k2 = cnt >> 1;
        uint32_t v1 = 0, v2 = 0;
        const bool isnan = mn < MK::KEY_NEG_INF || mx > MK::KEY_POS_INF;
        if (!isnan) {
            // invariant: count(key <= blo) = clo <= k1  and  count(key <= bhi) = chi > k2
            uint32_t blo = mn - 1, bhi = mx;
            int clo = 0, chi = cnt, par = 0, nstep = 0;
            bool done = false;
            while (!done) {
                if (chi - clo <= 64) {
                    // <= 64 candidates in (blo, bhi]: wave by wave into the line, one wave sorts
                    int mine = 0;
#pragma unroll
                    for (int r = 0; r < NREG; ++r) mine += med_popc(key[r] > blo && key[r] <= bhi);
                    if (lane == 0) sh.ncand[w] = mine;
                    if (tid < 64) sh.buf[tid] = MK::MAXK;
                    __syncthreads();
                    int base = 0;
                    for (int k = 0; k < w; ++k) base += sh.ncand[k];
#pragma unroll
                    for (int r = 0; r < NREG; ++r) {
                        const bool in = key[r] > blo && key[r] <= bhi;
                        const uint64_t m = __ballot(in);
                        const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                        if (in) sh.buf[pos] = key[r];
                        base += __popcll(m);
                    }
                    __syncthreads();
                    if (w == 0) {
                        const uint32_t v = med_bitonic64<uint32_t>(sh.buf[lane], lane);
                        v1 = __shfl(v, k1 - clo, 64);
                        v2 = __shfl(v, k2 - clo, 64);
                    }
                    done = true;
                } else if (bhi - blo == 1) { v1 = v2 = bhi; done = true; }       // all candidates are the same key
                else {
                    if (nstep == 10 || nstep == 16 || nstep == 22) {
                        // more than 64 keys left after this many halvings: the middle ranks sit in a TIE (decimal lot sizes) and the
                        // bracket would go on halving an empty key range down to one value (27 steps).  Snap it to the smallest
                        // and largest key inside (the counts at its ends stay what they are); all equal: one more round ends it
                        uint32_t a = MK::MAXK, bb = 0;
#pragma unroll
                        for (int r = 0; r < NREG; ++r) {
                            const uint32_t k = key[r];
                            const bool in = k > blo && k <= bhi;
                            a = (in && k < a) ? k : a;
                            bb = (in && k > bb) ? k : bb;
                        }
                        a = med_wave_umin<uint32_t>(a);
                        bb = med_wave_umax<uint32_t>(bb);
                        if (lane == 0) { sh.below[w] = a; sh.above[w] = bb; }
                        __syncthreads();
                        uint32_t mn_in = sh.below[0], mx_in = sh.above[0];
#pragma unroll
                        for (int k = 1; k < NW; ++k) {
                            mn_in = sh.below[k] < mn_in ? sh.below[k] : mn_in;
                            mx_in = sh.above[k] > mx_in ? sh.above[k] : mx_in;
                        }
                        __syncthreads();
                        bhi = mx_in;
                        blo = (mn_in == mx_in ? mx_in : mn_in) - 1;
                        ++nstep;
                        continue;
                    }
                    ++nstep;
                    const uint32_t pivot = blo + ((bhi - blo) >> 1);
                    int c = 0;
#pragma unroll
                    for (int r = 0; r < NREG; ++r) c += med_popc(key[r] <= pivot);
                    if (lane == 0) sh.cnt[par][w] = c;
                    __syncthreads();
                    c = 0;
#pragma unroll
                    for (int k = 0; k < NW; ++k) c += sh.cnt[par][k];
                    par ^= 1;
                    if (c > k2) { bhi = pivot; chi = c; }
                    else if (c <= k1) { blo = pivot; clo = c; }
                    else {
                        // k1 < c <= k2: the pivot separates the two ranks -- largest key <= pivot, smallest key > pivot
                        uint32_t a = 0, bb = MK::MAXK;
#pragma unroll
                        for (int r = 0; r < NREG; ++r) {
                            const uint32_t k = key[r];
                            a = (k <= pivot && k > a) ? k : a;
                            bb = (k > pivot && k < bb) ? k : bb;
                        }
                        a = med_wave_umax<uint32_t>(a);
                        bb = med_wave_umin<uint32_t>(bb);
                        if (lane == 0) { sh.below[w] = a; sh.above[w] = bb; }
                        __syncthreads();
                        v1 = sh.below[0]; v2 = sh.above[0];
#pragma unroll
                        for (int k = 1; k < NW; ++k) {
                            v1 = sh.below[k] > v1 ? sh.below[k] : v1;
                            v2 = sh.above[k] < v2 ? sh.above[k] : v2;
                        }
                        done = true;
                    }
                }
            }
        }
        if (tid == 0) o.median[b] = isnan ? (double)NAN : (cnt & 1) ? MK::value(v1) : (MK::value(v1) + MK::value(v2)) / 2.0;
    }
}

template <bool MEDIAN, int NREG, int THREADS>
__global__ __launch_bounds__(THREADS) void k_bar_ohlcv_mid(const double *__restrict__ price, const float *__restrict__ amount,
                                                          const int64_t *__restrict__ ci, const int64_t *__restrict__ list,
                                                          const int *__restrict__ go, OhlcvOut o)
{
    if (go && *go == 0) return;
    __shared__ OhmShared<THREADS / 64> sh;
    const int64_t n_list = list[0];
    for (int64_t q = blockIdx.x; q < n_list; q += gridDim.x) {
        const int64_t b = list[1 + q], s = ci[b], e = ci[b + 1];
        ohm_bar<MEDIAN, NREG, THREADS>(price, amount, b, s, e, o, sh);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Bars of 1 345 .. 4 096 ticks (float32 amounts; round 3): still ONE WAVE per bar and one pass, the keys of the whole bar in
// registers (32 or 64 per lane), the loads in phases of 16 chunks so that the price registers are reused.  k_bar_ohlcv_small stops
// at 21 chunks (its up-front loads hold the 1-minute kernel at four waves per SIMD); beyond it the bars took the generic streaming
// kernel plus a median kernel of their own -- 75-second bars 6.1 ms per 1e9 ticks against 2.4 ms for 60-second bars.  The
// workgroup-per-bar kernel above pays ~15 000 cycles of barriers and reductions per bar, which only bars beyond ~4 000 ticks
// amortise (profiles/r03_median_humps.txt).  Same per-lane order and tree as every wave-per-bar schedule: identical bits.
// ---------------------------------------------------------------------------------------------------------------------
template <bool MEDIAN, int NKEY, int PH>
__global__ __launch_bounds__(256) void k_bar_ohlcv_phased(const double *__restrict__ price, const float *__restrict__ amount,
                                                          const int64_t *__restrict__ ci, const int64_t *__restrict__ list,
                                                          const int *__restrict__ go, OhlcvOut o)
{
    if (go && *go == 0) return;
    typedef MedKey<false> MK;
    __shared__ uint32_t sbuf[4][64];
    const int lane = fmk_lane();
    const int wib = fmk_uniform((int)(threadIdx.x >> 6));
    uint32_t *buf = sbuf[wib];
    const int64_t n_list = list[0];
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t q = (int64_t)blockIdx.x * 4 + wib; q < n_list; q += nwaves) {
        const int64_t b = fmk_uniform(list[1 + q]);
        const int64_t s = fmk_uniform(ci[b]), e = fmk_uniform(ci[b + 1]);
        const int64_t start = s + 1, cnt = e - s;
        const int nch = (int)((cnt + 63) >> 6);
        const double *pb = price + start;
        const uint32_t *ab = (const uint32_t *)amount + start;
        const unsigned last = (unsigned)(cnt - 1);
        double hi = -INFINITY, lo = INFINITY, tv = 0.0, td = 0.0;
        MedBar<false, (MEDIAN ? NKEY : 1), false> bar;
#pragma unroll
        for (int ph = 0; ph < NKEY / PH; ++ph) {
            if (ph * PH < nch) {                                       // wave-uniform
                double p[PH];
                uint32_t araw[PH];
#pragma unroll
                for (int c = 0; c < PH; ++c) {
                    unsigned idx = (unsigned)((ph * PH + c) * 64 + lane);
                    idx = idx < last ? idx : last;
                    p[c] = pb[idx];
                    araw[c] = ab[idx];
                }
#pragma unroll
                for (int c = 0; c < PH; ++c) {
                    const bool valid = (unsigned)((ph * PH + c) * 64 + lane) <= last;
                    const double a = (double)__uint_as_float(araw[c]);
                    hi = fmax(hi, p[c]);                               // clamped duplicates cannot change max / min
                    lo = fmin(lo, p[c]);
                    tv += valid ? a : 0.0;
                    td += valid ? p[c] * a : 0.0;
                    if constexpr (MEDIAN) bar.key[ph * PH + c] = valid ? MK::tokey(araw[c]) : MK::MAXK;
                }
            } else if constexpr (MEDIAN) {
#pragma unroll
                for (int c = 0; c < PH; ++c) bar.key[ph * PH + c] = MK::MAXK;
            }
        }
        ohlcv_finish<false>(o, b, price, start, e, hi, lo, tv, td, lane);
        if constexpr (MEDIAN) {
            bar.amount = amount; bar.start = start; bar.cnt = cnt; bar.lane = lane;
            const double m = med_search<false, NKEY, false, true>(bar, buf);
            if (lane == 0) o.median[b] = m;
        }
    }
}

static unsigned ohlcv_grid(fmk_ctx *ctx, int64_t nb)
{
    int64_t blocks = fmk_ceil_div(nb, 4);
    const int per_cu = 64;                  // workgroups per CU in the grid
    int64_t cap = (int64_t)ctx->n_cu * per_cu;   // grid-stride beyond this
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

// What the first kernel of a comp_bar_ohlcv call left behind (bars longer than `long_min` ticks; `saw_long`: its flag)
template <bool AF64>
static int ohlcv_leftovers(fmk_ctx *ctx, const double *p, const void *a, const int64_t *ci, int64_t nb, int64_t n,
                           const OhlcvOut &o, int *saw_long, int64_t long_min, unsigned grid)
{
    // long bars (if any): the generic kernels exit at once when the flag is clear.  float32 bars of 1 345 .. 8 192 ticks: one pass by
    // a workgroup each, median included
    int64_t skip_lo = 0, skip_hi = 0;
    if constexpr (!AF64) {
        {
            skip_lo = OHM_MIN;
            skip_hi = OHM_MAX;
            // 1 345 .. 2 048, .. 3 072, .. 4 096, .. 6 144 ticks: a wave per bar with 32 / 48 / 64 / 96 key registers; 6 145 .. 8 192: a workgroup per bar
            constexpr int NL = 6;
            int64_t *list[NL] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
            static const int64_t edge[NL + 1] = {OHM_MIN, 2048, 3072, 4096, 6144, 8192, OHM_MAX};
            int rc = fmk_long_bar_lists(ctx, ci, nb, n, NL, edge, saw_long, list);      // (one pass, one allocation: list[0] owns it)
            // measured per 1e9 ticks, ohlcv + median (profiles/r03_median_humps.txt): 4 400 / 5 200 / 6 000-tick bars 3.6 / 3.4 / 3.1 ms
            // with 96 key registers per lane against 5.1 / 4.5 / 4.0 ms by the workgroup kernel; 7 000 / 8 000-tick bars 5.5 / 5.2 ms
            // with 128 key registers (spills) against 3.6 / 3.3 ms by the workgroup kernel
            if (rc == FMK_OK) {
                const float *af = (const float *)a;
                const unsigned g = (unsigned)(ctx->n_cu * 8);
                if (o.median) {
                    k_bar_ohlcv_phased<true, 32, 16><<<g, 256, 0, ctx->stream>>>(p, af, ci, list[0], saw_long, o);
                    k_bar_ohlcv_phased<true, 48, 8><<<g, 256, 0, ctx->stream>>>(p, af, ci, list[1], saw_long, o);
                    k_bar_ohlcv_phased<true, 64, 8><<<g, 256, 0, ctx->stream>>>(p, af, ci, list[2], saw_long, o);
                    k_bar_ohlcv_phased<true, 96, 8><<<g, 256, 0, ctx->stream>>>(p, af, ci, list[3], saw_long, o);
                    k_bar_ohlcv_mid<true, 32, 256><<<g, 256, 0, ctx->stream>>>(p, af, ci, list[4], saw_long, o);
                    k_bar_ohlcv_mid<true, 32, 512><<<g / 2, 512, 0, ctx->stream>>>(p, af, ci, list[5], saw_long, o);
                } else {
                    k_bar_ohlcv_phased<false, 32, 16><<<g, 256, 0, ctx->stream>>>(p, af, ci, list[0], saw_long, o);
                    k_bar_ohlcv_phased<false, 48, 8><<<g, 256, 0, ctx->stream>>>(p, af, ci, list[1], saw_long, o);
                    k_bar_ohlcv_phased<false, 64, 8><<<g, 256, 0, ctx->stream>>>(p, af, ci, list[2], saw_long, o);
                    k_bar_ohlcv_phased<false, 96, 8><<<g, 256, 0, ctx->stream>>>(p, af, ci, list[3], saw_long, o);
                    k_bar_ohlcv_mid<false, 32, 256><<<g, 256, 0, ctx->stream>>>(p, af, ci, list[4], saw_long, o);
                    k_bar_ohlcv_mid<false, 32, 512><<<g / 2, 512, 0, ctx->stream>>>(p, af, ci, list[5], saw_long, o);
                }
            }
            const hipError_t le = hipGetLastError();
            if (list[0]) (void)fmk_free(ctx, list[0]);
            FMK_TRY(rc);
            FMK_HIP(ctx, le);
        }
    }
    k_bar_ohlcv<AF64><<<grid, 256, 0, ctx->stream>>>(p, a, ci, nb, n, long_min, saw_long, o, skip_lo, skip_hi);
    int wide_median_done = 0;
    const int64_t wide_min = skip_hi > OH_WIDE_MIN ? skip_hi : OH_WIDE_MIN;       // the workgroup classes reach further than the generic kernel
    FMK_TRY(oh_wide_launch<AF64>(ctx, p, a, ci, nb, n, saw_long, o, &wide_median_done, wide_min));
    FMK_LAUNCH_CHECK(ctx);
    if (AF64) {
        k_bar_vol_redo<<<grid < 4096 ? grid : 4096, 256, 0, ctx->stream>>>((const double *)a, ci, o.vol, o.vol_redo);
        FMK_LAUNCH_CHECK(ctx);
    }
    if (o.median)
        return fmk_median_launch(ctx, a, AF64, ci, nb, long_min, saw_long, o.median, n, skip_lo, skip_hi,
                                 wide_median_done ? wide_min : INT64_MAX);
    return FMK_OK;
}

template <bool AF64>
static int ohlcv_launch(fmk_ctx *ctx, const double *p, const void *a, const int64_t *ci, int64_t nb, int64_t n,
                        const OhlcvOut &o_in, int variant, const TbFuse *tb = nullptr)
{
    OhlcvOut o = o_in;
    const unsigned grid = ohlcv_grid(ctx, nb);
    // tb: the close indices are not there yet (fmk_time_bars_ohlcv_dev) -- the 1-minute schedule pipelines the indexer with the bar
    // kernel in two stages, every other schedule runs the separate indexer first
    int64_t pipe_ka = 0;                     // > 0: bars [0, pipe_ka) are the first stage of the pipelined time-bar step
    if (tb) {
        const int mid2 = 600;
        // PIPELINED time-bar step (the default for the 1-minute schedule, float32 amounts): the indexer in two stages -- see below
        const int split = 8;               // 1 / share of the bars in the first stage (0: off)
        const char *msv = getenv("FMK_TB_PIPE_MIN_STAGE");            // developer knob (tests): bars in the first stage from which the
        const int64_t min_stage = msv ? atoll(msv) : 4096;            // step is pipelined (read on every call)
        if (!AF64 && split >= 2 && variant != 0 && n / nb > mid2 && nb / split >= min_stage && min_stage >= 256) {
            pipe_ka = (nb / split) & ~(int64_t)255;
        } else
            FMK_TRY(fmk_time_bar_indexer_dev(ctx, tb->ts, n, tb->e0, tb->d, nb + 1, tb->clock, tb->idx));
        ci = tb->idx;
    }
    if (AF64) {   // redo list of near-tie volume sums (fmk_f32tie.h); nothing else here uses the context scratch
        FMK_TRY(fmk_scratch(ctx, (size_t)(nb + 32) * 8, (void **)&o.vol_redo));
        FMK_HIP(ctx, hipMemsetAsync(o.vol_redo, 0, 8, ctx->stream));
    }
    // time the dominant launch only (the one-bar boundary launch of a sharded step is not it)
    const int slot = (ctx->profile_on && nb >= 64 && pipe_ka == 0) ? (ctx->profile_n++ & (FMK_PROFILE_SLOTS - 1)) : -1;
    if (slot >= 0) FMK_HIP(ctx, hipEventRecord(ctx->kev[slot][0], ctx->stream));
    if (variant == 0) {   // generic streaming kernel only (+ stand-alone median)
        k_bar_ohlcv<AF64><<<grid, 256, 0, ctx->stream>>>(p, a, ci, nb, n, 0, nullptr, o);
        FMK_LAUNCH_CHECK(ctx);
        if (slot >= 0) FMK_HIP(ctx, hipEventRecord(ctx->kev[slot][1], ctx->stream));
        FMK_TRY(oh_wide_launch<AF64>(ctx, p, a, ci, nb, n, nullptr, o));
        if (AF64) {
            k_bar_vol_redo<<<grid < 4096 ? grid : 4096, 256, 0, ctx->stream>>>((const double *)a, ci, o.vol, o.vol_redo);
            FMK_LAUNCH_CHECK(ctx);
        }
        if (o.median) return fmk_median_launch(ctx, a, AF64, ci, nb, 0, nullptr, o.median, n);
        return FMK_OK;
    }
    if constexpr (!AF64) {
        if (pipe_ka > 0) {
            // ---- the pipelined time-bar step (round 4).  The indexer (0.13 ms: sample table + one search per clock edge, bound by
            // random line fetches) used to stand in front of a 2.2 ms kernel that cannot start without it, and the call ended with a
            // read-back of the long-bar flag (memset + copy + host wait + the host's way back into the next call: another 0.05 ms
            // of idle device per step; rocprofv3 timeline in profiles/r04_step_timeline.txt).  Now: sample table and the edges of
            // the first 1/8 of the bars on the context's stream (~25 us), OHLCV + median of those bars, and BESIDE that launch the
            // remaining edges on the auxiliary stream; the second OHLCV launch waits for them by event.  Both index stages also take
            // the census (is any bar longer than 1 344 ticks?), so the flag reaches the host while the first OHLCV launch is still
            // running: the call decides about the leftover passes and returns without ever waiting for a kernel it launched.
            FMK_TRY(fmk_ctx_aux(ctx));
            const int idx_bpc = 2;         // workgroups per CU of the second index stage (0: no cap)
            int *saw_long = (int *)(ctx->d_mail + 16);
            const int64_t *coarse = nullptr;
            int64_t m = 0;
            const int64_t ne = nb + 1, long_min = 64 * FMK_SMALL_NCH;
            FMK_TRY(fmk_time_bar_coarse_launch(ctx, tb->ts, n, &coarse, &m, saw_long));          // (also clears the flag)
            FMK_TRY(fmk_time_bar_index_stage(ctx, ctx->stream, tb->ts, n, tb->e0, tb->d, ne, coarse, m, 0, pipe_ka + 1, tb->clock,
                                             tb->idx, saw_long, long_min, 0));
            FMK_HIP(ctx, hipEventRecord(ctx->aev[0], ctx->stream));
            FMK_HIP(ctx, hipStreamWaitEvent(ctx->aux, ctx->aev[0], 0));
            FMK_TRY(fmk_time_bar_index_stage(ctx, ctx->aux, tb->ts, n, tb->e0, tb->d, ne, coarse, m, pipe_ka + 1, ne, tb->clock,
                                             tb->idx, saw_long, long_min, (int64_t)ctx->n_cu * idx_bpc));
            FMK_HIP(ctx, hipEventRecord(ctx->aev[1], ctx->aux));
            // (also in enqueue-only mode: this wait ends when the index stages do, ~0.1 ms into a 2 ms launch -- the device never idles
            //  for it, and it saves the ~36 empty launches of the leftover passes, 0.25 ms per step of the sharded path)
            int *h_saw = (int *)(ctx->h_mail + 50);
            FMK_HIP(ctx, hipMemcpyAsync(h_saw, saw_long, sizeof(int), hipMemcpyDeviceToHost, ctx->aux));
            FMK_HIP(ctx, hipEventRecord(ctx->aev[2], ctx->aux));
            for (int stage = 0; stage < 2; ++stage) {
                const int64_t b0 = stage ? pipe_ka : 0, cnt = stage ? nb - pipe_ka : pipe_ka;
                OhlcvOut q = o;
                q.open += b0; q.high += b0; q.low += b0; q.close += b0; q.vol += b0; q.vwap += b0; q.trades += b0;
                if (q.median) q.median += b0;
                if (stage) FMK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->aev[1], 0));
                const int sl = (ctx->profile_on) ? (ctx->profile_n++ & (FMK_PROFILE_SLOTS - 1)) : -1;      // each launch of the dominant kernel on its own
                if (sl >= 0) FMK_HIP(ctx, hipEventRecord(ctx->kev[sl][0], ctx->stream));
                const unsigned g = ohlcv_grid(ctx, cnt);
                if (!o.median) k_bar_ohlcv_small<AF64, false><<<g, 256, 0, ctx->stream>>>(p, a, ci + b0, cnt, n, saw_long, q);
                else k_bar_ohlcv_small<AF64, true><<<g, 256, 0, ctx->stream>>>(p, a, ci + b0, cnt, n, saw_long, q);
                FMK_LAUNCH_CHECK(ctx);
                if (sl >= 0) FMK_HIP(ctx, hipEventRecord(ctx->kev[sl][1], ctx->stream));
            }
            {
                FMK_HIP(ctx, hipEventSynchronize(ctx->aev[2]));          // the index stages' census: long before the kernels end
                if (*h_saw == 0) return FMK_OK;
            }
            return ohlcv_leftovers<AF64>(ctx, p, a, ci, nb, n, o, saw_long, 64 * FMK_SMALL_NCH, grid);
        }
    }
    // small bars: all loads up front (+ fused median); long bars: generic kernels on the rest
    // launch bounds measured on MI355X: forcing >4 waves/SIMD on the fused-median kernel makes the compiler
    // serialise the up-front loads (2.7 ms -> 3.5..4.3 ms at N = 1e9); 4 waves/SIMD (102 VGPRs) is the optimum.
    int *saw_long = (int *)(ctx->d_mail + 16);               // set by the small kernel iff a long bar exists
    FMK_HIP(ctx, hipMemsetAsync(saw_long, 0, sizeof(int), ctx->stream));
    // mean bar length (the tick array's length over the bars is an upper bound) picks the schedule: several whole bars per
    // LANE below FMK_PACKED_MAX_MEAN ticks per bar (k_bar_ohlcv_lanes), one bar per wave above
    const int packed_max = FMK_PACKED_MAX_MEAN;              // (0 disables the packed schedule)
    int64_t long_min = 64 * FMK_SMALL_NCH;
    const int mid_max = 210;                 // mean ticks per bar up to which the 65..256-tick instantiation serves the stream
    const int mid2_max = 600;                // ... and the <= 640-tick instantiation
    const int rows_on = 1;                   // the sixteen-lanes-per-bar schedule
    const int rows_min = 57;                 // mean ticks per bar from which rows replace the lane schedule (without the median)
    // With the median: eight lanes per bar (bars of <= 128 ticks) serve streams of 33 .. 63 ticks per bar,
    // sixteen lanes per bar from 64 (profiles/r04_short_bars.txt: 5.8 / 5.3 / 4.9 / 4.3 ms at 34 / 40 / 46 / 60 ticks against the lane
    // schedule's 7.0 / 6.6 / 7.0 / 8.9; the half rows are ahead of the rows up to ~100 ticks on equal bars, but a stream's longer bars
    // -- twice its mean -- must still fit the schedule: 63 x 2 <= 128).  Without the median the lane schedule stays ahead up to 56.
    // (Four lanes per bar, sixteen bars per wave, was measured too: 6.9 / 6.2 / 5.4 ms at 20 / 26 / 34 ticks -- behind; not dispatched.)
    const int half_min = 33;
    const int64_t rows_from = o.median ? 64 : rows_min;
    if (!AF64 && rows_on && o.median && half_min > 0 && nb >= 64 && n / nb >= half_min && n / nb < rows_from) {
        if constexpr (!AF64) {
            int64_t blocks = fmk_ceil_div(fmk_ceil_div(nb, 8), OHR_WAVES);
            const int64_t cap = (int64_t)ctx->n_cu * 32;
            if (blocks > cap) blocks = cap;
            k_bar_ohlcv_rows<true, 8><<<(unsigned)blocks, 64 * OHR_WAVES, 0, ctx->stream>>>(p, (const float *)a, ci, nb, n, saw_long, o);
        }
        long_min = 128;
    } else if (!AF64 && nb >= 64 && n / nb <= packed_max && !(rows_on && n / nb >= rows_from)) {
        int64_t blocks = fmk_ceil_div(fmk_ceil_div(nb, 64), 2);
        const int64_t cap = (int64_t)ctx->n_cu * 96;
        if (blocks > cap) blocks = cap;
        // bars of 45 .. 64 ticks fill a 1 024-tick tile with 16 .. 22 bars only: a 2 048-tick tile for those -- 50 / 60-tick bars 7.4 / 9.5 ->
        // 6.7 / 8.9 ms per 1e9 ticks, 34 / 40-tick bars are better off with the small tile (7.0 / 6.6 against 7.9 / 7.4) (profiles/r04_short_bars.txt)
        if (n / nb > 44) {
            if (!o.median) k_bar_ohlcv_lanes<false, 2048><<<(unsigned)blocks, 128, 0, ctx->stream>>>(p, (const float *)a, ci, nb, n, saw_long, o);
            else k_bar_ohlcv_lanes<true, 2048><<<(unsigned)blocks, 128, 0, ctx->stream>>>(p, (const float *)a, ci, nb, n, saw_long, o);
        } else
        if (!o.median) k_bar_ohlcv_lanes<false, 1024><<<(unsigned)blocks, 128, 0, ctx->stream>>>(p, (const float *)a, ci, nb, n, saw_long, o);
        else k_bar_ohlcv_lanes<true, 1024><<<(unsigned)blocks, 128, 0, ctx->stream>>>(p, (const float *)a, ci, nb, n, saw_long, o);
        long_min = 64;
    } else if (!AF64 && rows_on && nb >= 64 && n / nb <= mid_max) {
        // streams of 65..256-tick bars (float32 amounts): sixteen lanes per bar, four bars per wave; longer bars are left
        if constexpr (!AF64) {
            int64_t blocks = fmk_ceil_div(fmk_ceil_div(nb, 4), OHR_WAVES);
            const int64_t cap = (int64_t)ctx->n_cu * 32;
            if (blocks > cap) blocks = cap;
            if (!o.median) k_bar_ohlcv_rows<false><<<(unsigned)blocks, 64 * OHR_WAVES, 0, ctx->stream>>>(p, (const float *)a, ci, nb, n, saw_long, o);
            else k_bar_ohlcv_rows<true><<<(unsigned)blocks, 64 * OHR_WAVES, 0, ctx->stream>>>(p, (const float *)a, ci, nb, n, saw_long, o);
        }
        long_min = 256;
    } else if (n / nb <= mid_max) {
        // streams of 65..256-tick bars: the instantiation without the long classes (eight waves per SIMD); longer bars are left
        int64_t blocks = fmk_ceil_div(nb, 4);
        const int64_t cap = (int64_t)ctx->n_cu * 128;
        if (blocks > cap) blocks = cap;
        if (!o.median) k_bar_ohlcv_small<AF64, false, 4><<<(unsigned)blocks, 256, 0, ctx->stream>>>(p, a, ci, nb, n, saw_long, o);
        else k_bar_ohlcv_small<AF64, true, 4><<<(unsigned)blocks, 256, 0, ctx->stream>>>(p, a, ci, nb, n, saw_long, o);
        long_min = 256;
    } else if (mid_max > 0 && n / nb <= mid2_max) {
        // ... and of 257..640-tick bars: size classes up to 10 chunks (six waves per SIMD)
        int64_t blocks = fmk_ceil_div(nb, 4);
        const int64_t cap = (int64_t)ctx->n_cu * 96;
        if (blocks > cap) blocks = cap;
        if (!o.median) k_bar_ohlcv_small<AF64, false, 10><<<(unsigned)blocks, 256, 0, ctx->stream>>>(p, a, ci, nb, n, saw_long, o);
        else k_bar_ohlcv_small<AF64, true, 10><<<(unsigned)blocks, 256, 0, ctx->stream>>>(p, a, ci, nb, n, saw_long, o);
        long_min = 640;
    } else if (!o.median) k_bar_ohlcv_small<AF64, false><<<grid, 256, 0, ctx->stream>>>(p, a, ci, nb, n, saw_long, o);
    else k_bar_ohlcv_small<AF64, true><<<grid, 256, 0, ctx->stream>>>(p, a, ci, nb, n, saw_long, o);
    FMK_LAUNCH_CHECK(ctx);
    if (slot >= 0) FMK_HIP(ctx, hipEventRecord(ctx->kev[slot][1], ctx->stream));
    // Everything below serves the bars the first kernel left behind -- six list kernels, six size classes, the generic kernel, the
    // workgroup kernels and their median passes: ~36 launches that exit at once when there is no such bar, ~0.25 ms per call at
    // ~7 us each (10 % of the 1-minute headline pass, rocprofv3 of bench.py at the end of round 3).  So the flag is read back first:
    // one 4-byte copy and a wait for the kernel that is the call's work anyway.  fmk_ctx_set_enqueue_only(ctx, 1) (the sharded
    // step: two waits per step cost it 0.9 ms) : enqueue everything without looking.
    if constexpr (!AF64) {
        if (n <= long_min) return FMK_OK;                          // no bar is longer than the tick array (the sharded step's boundary bar)
        const int census_sync = 1;
        if (census_sync && !ctx->enqueue_only) {
            int *h_saw = (int *)(ctx->h_mail + 12);
            FMK_HIP(ctx, hipMemcpyAsync(h_saw, saw_long, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (*h_saw == 0) return FMK_OK;
        }
    }
    return ohlcv_leftovers<AF64>(ctx, p, a, ci, nb, n, o, saw_long, long_min, grid);
}

// comp_bar_ohlcv (without the median) for the bars longer than min_cnt that another kernel left behind (`go`: its flag)
int fmk_ohlcv_leftover_launch(fmk_ctx *ctx, const double *p, const void *a, int amount_is_f64, const int64_t *ci, int64_t nb,
                              int64_t n, int64_t min_cnt, const int *go, double *d_open, double *d_high, double *d_low,
                              double *d_close, float *d_volume, double *d_vwap, int64_t *d_trades)
{
    OhlcvOut o{d_open, d_high, d_low, d_close, d_volume, d_vwap, d_trades, nullptr, nullptr};
    const unsigned grid = ohlcv_grid(ctx, nb);
    if (amount_is_f64) {
        k_bar_ohlcv<true><<<grid, 256, 0, ctx->stream>>>(p, a, ci, nb, n, min_cnt, go, o);
        FMK_TRY(oh_wide_launch<true>(ctx, p, a, ci, nb, n, go, o));
    } else {
        k_bar_ohlcv<false><<<grid, 256, 0, ctx->stream>>>(p, a, ci, nb, n, min_cnt, go, o);
        FMK_TRY(oh_wide_launch<false>(ctx, p, a, ci, nb, n, go, o));
    }
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}

// cfg 4's first half on bars of unequal length (fmk_barflow.hip, float32 amounts): the lane-per-bar order-flow kernel has written
// open .. trades of the bars it walked; the MEDIAN of a bar of <= 1 344 ticks comes from the amounts alone (k_bar_median_small), and
// the bars beyond that take comp_bar_ohlcv's own size classes -- one pass each over price and amount, the median fused (their open ..
// trades are written again, in this file's summation order: the values comp_bar_ohlcv itself gives).  The stand-alone median kernels
// that served those bars before cost 3.0 ms per 1e9 ticks of lognormal bars, the classes 2.0 (profiles/r04_cfg4.txt).
int fmk_median_small_ohlcv_long_launch(fmk_ctx *ctx, const double *p, const float *a, const int64_t *ci, int64_t nb, int64_t n,
                                       double *d_open, double *d_high, double *d_low, double *d_close, float *d_volume,
                                       double *d_vwap, int64_t *d_trades, double *d_median)
{
    int *saw_long = (int *)(ctx->d_mail + 16);
    FMK_HIP(ctx, hipMemsetAsync(saw_long, 0, sizeof(int), ctx->stream));
    int64_t blocks = fmk_ceil_div(nb, 4);
    const int64_t cap = (int64_t)ctx->n_cu * 64;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    k_bar_median_small<<<(unsigned)blocks, 256, 0, ctx->stream>>>(a, ci, nb, saw_long, d_median);
    FMK_LAUNCH_CHECK(ctx);
    OhlcvOut o{d_open, d_high, d_low, d_close, d_volume, d_vwap, d_trades, d_median, nullptr};
    return ohlcv_leftovers<false>(ctx, p, a, ci, nb, n, o, saw_long, 64 * FMK_SMALL_NCH, ohlcv_grid(ctx, nb));
}

extern "C" int fmk_comp_bar_ohlcv_dev(fmk_ctx *ctx, const double *d_price, const void *d_amount,
                                      int amount_is_f64, int64_t n, const int64_t *d_close_idx, int64_t n_idx,
                                      double *d_open, double *d_high, double *d_low, double *d_close,
                                      float *d_volume, double *d_vwap, int64_t *d_trades, double *d_median)
{
    if (n_idx < 2)   // base.py:334-335
        return fmk_set_error(ctx, FMK_E_ARG, "Bar close indices must contain at least two elements.");
    if (n <= 0) return fmk_set_error(ctx, FMK_E_ARG, "comp_bar_ohlcv: empty price array");
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t nb = n_idx - 1;
    OhlcvOut o{d_open, d_high, d_low, d_close, d_volume, d_vwap, d_trades, d_median, nullptr};
    const int variant = 1;     // (0: the generic kernels only)
    return amount_is_f64 ? ohlcv_launch<true>(ctx, d_price, d_amount, d_close_idx, nb, n, o, variant)
                         : ohlcv_launch<false>(ctx, d_price, d_amount, d_close_idx, nb, n, o, variant);
}

// _time_bar_indexer + comp_bar_ohlcv in one call (TimeBarKit.build_ohlcv on resident columns; bench.py's step): the results of
// fmk_time_bar_indexer_dev(first_edge, delta, n_edges) followed by fmk_comp_bar_ohlcv_dev on its close indices, bit for bit; for
// streams of 1-minute-sized bars (mean bar length above 600 ticks) the indexer runs in two stages, pipelined with the bar kernel.
extern "C" int fmk_time_bars_ohlcv_dev(fmk_ctx *ctx, const int64_t *d_ts, const double *d_price, const void *d_amount,
                                       int amount_is_f64, int64_t n, int64_t ts_first, int64_t ts_last, int64_t first_edge,
                                       int64_t delta, int64_t n_edges, int64_t *d_clock, int64_t *d_close_idx, double *d_open,
                                       double *d_high, double *d_low, double *d_close, float *d_volume, double *d_vwap,
                                       int64_t *d_trades, double *d_median)
{
    if (n <= 0) return fmk_set_error(ctx, FMK_E_ARG, "time_bars_ohlcv: empty input");
    if (n_edges < 2)   // base.py:334-335
        return fmk_set_error(ctx, FMK_E_ARG, "Bar close indices must contain at least two elements.");
    if (!d_close_idx) return fmk_set_error(ctx, FMK_E_ARG, "time_bars_ohlcv: d_close_idx is required");
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t nb = n_edges - 1;
    OhlcvOut o{d_open, d_high, d_low, d_close, d_volume, d_vwap, d_trades, d_median, nullptr};
    const TbFuse tb{d_ts, first_edge, delta, ts_first, ts_last, d_clock, d_close_idx};
    const int variant = 1;
    return amount_is_f64 ? ohlcv_launch<true>(ctx, d_price, d_amount, d_close_idx, nb, n, o, variant, &tb)
                         : ohlcv_launch<false>(ctx, d_price, d_amount, d_close_idx, nb, n, o, variant, &tb);
}
