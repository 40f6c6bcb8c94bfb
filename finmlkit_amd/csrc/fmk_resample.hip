// fmk_resample.hip -- TimeBarReader._resample (finmlkit/bar/io.py:890-950): bars -> coarser bars.
// The reference groups the rows of a bar frame by `index.floor(timeframe)` and aggregates with pandas:
//   open first / high max / low min / close last (NaN skipped, io.py:917-922), volume and trades summed (:923-924),
//   vwap = sum(vwap * volume) / sum(volume) cast to float32 (:928-930), median_trade_size = the trades-weighted median of
//   the rows' medians, sizes[searchsorted(cumsum(weights), total * 0.5, 'left')] after an argsort (:933-946), float32.
// pandas' groupby sum is a KAHAN-compensated sequential sum in the column's own dtype (float32 volume -> float32
// accumulator and compensation; checked against pandas 2.3.3 in oracle/gen_resample.py), so the sums here are that
// recurrence, row by row, not a tree.  The host layer (finmlkit_amd/bar/io.py) computes the group keys with pandas' own
// `floor` and hands over contiguous segments; one wave aggregates one group:
//   rows in chunks of 64, coalesced column loads; first / last non-NaN by ballot, max / min and the integer trade count by
//   wave reductions; the two Kahan chains run in lane 0 over an LDS copy of the chunk;
//   the weighted median needs no sort: the answer is the smallest size s with W(size <= s) >= total / 2 (ties in size share a
//   value, so the order an unstable argsort gives them cannot matter), found by bisection on the order-preserving 64-bit key
//   with one masked integer wave sum per step.
#include "fmk_common.h"
#include "fmk_dpp.h"

namespace {

__device__ __forceinline__ uint64_t rs_key(double x)
{
    if (x != x) return ~0ULL;                                    // NaN sorts last (np.argsort)
    if (x == 0.0) x = 0.0;                                       // -0.0 == 0.0
    const uint64_t b = (uint64_t)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}
__device__ __forceinline__ double rs_unkey(uint64_t k)
{
    if (k == ~0ULL) return __longlong_as_double(0x7FF8000000000000LL);
    const uint64_t b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFULL) : ~k;
    return __longlong_as_double((int64_t)b);
}

template <typename T>
__device__ __forceinline__ void rs_kahan(T &sum, T &comp, T val)   // pandas/_libs/groupby.pyx group_sum
{
    if (val != val) return;                                      // NaN rows are skipped
    const T y = val - comp;
    const T t = sum + y;
    comp = (t - sum) - y;
    if (comp != comp) comp = 0;                                  // inf - inf
    sum = t;
}

template <bool F64>
__device__ __forceinline__ double rs_load(const void *p, int64_t i)
{
    if constexpr (F64) return ((const double *)p)[i];
    else return (double)((const float *)p)[i];
}

template <bool VF64, bool WF64>
__global__ __launch_bounds__(256) void k_resample(const int64_t *__restrict__ seg, int64_t n_groups,
                                                  const double *__restrict__ open, const double *__restrict__ high,
                                                  const double *__restrict__ low, const double *__restrict__ close,
                                                  const void *__restrict__ volume, const int64_t *__restrict__ trades,
                                                  const void *__restrict__ vwap, const double *__restrict__ median,
                                                  double *__restrict__ o_open, double *__restrict__ o_high,
                                                  double *__restrict__ o_low, double *__restrict__ o_close,
                                                  void *__restrict__ o_volume, int64_t *__restrict__ o_trades,
                                                  float *__restrict__ o_vwap, float *__restrict__ o_median,
                                                  uint8_t *__restrict__ o_valid)
{
    typedef typename std::conditional<VF64, double, float>::type V;
    typedef typename std::conditional<(VF64 || WF64), double, float>::type P;     // dtype of vwap * volume
    __shared__ V s_vol[4][64];
    __shared__ P s_pv[4][64];
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const double NaN = __longlong_as_double(0x7FF8000000000000LL);
    for (int64_t g = (int64_t)blockIdx.x * 4 + w; g < n_groups; g += nwaves) {
        const int64_t s = fmk_uniform(seg[g]), e = fmk_uniform(seg[g + 1]);
        double f_open = NaN, l_close = NaN, hi = NaN, lo = NaN;
        bool have_open = false;
        int64_t tr = 0;
        V vs = 0, vc = 0;
        P ps = 0, pc = 0;
        uint64_t kmin = ~0ULL;
        uint64_t my_key = ~0ULL;                                 // (key, weight) of this lane's row when the group fits a wave
        int64_t my_w = 0;
        for (int64_t c0 = s; c0 < e; c0 += 64) {
            const int64_t i = c0 + lane;
            const bool in = i < e;
            const double op = in ? open[i] : NaN, cl = in ? close[i] : NaN;
            const double h = in ? high[i] : NaN, l = in ? low[i] : NaN;
            const double vol = in ? rs_load<VF64>(volume, i) : NaN;
            const double vw = in ? rs_load<WF64>(vwap, i) : NaN;
            const int64_t t = in ? trades[i] : 0;
            const uint64_t key = in ? rs_key(median[i]) : ~0ULL;
            // first / last non-NaN (pandas first / last skip NaN)
            const uint64_t bo = __ballot(op == op), bc = __ballot(cl == cl);
            if (!have_open && bo) { f_open = __shfl(op, __ffsll((unsigned long long)bo) - 1, 64); have_open = true; }
            if (bc) l_close = __shfl(cl, 63 - __clzll((unsigned long long)bc), 64);
            // max / min skipping NaN: fmax / fmin return the other operand for a NaN
            const double ch = fmk_dpp_reduce(h, NaN, [](double a, double b) { return fmax(a, b); });
            const double cw = fmk_dpp_reduce(l, NaN, [](double a, double b) { return fmin(a, b); });
            hi = fmax(hi, ch);
            lo = fmin(lo, cw);
            tr += fmk_dpp_reduce(t, (int64_t)0, [](int64_t a, int64_t b) { return a + b; });
            if (in && key < kmin) kmin = key;                    // per-lane minimum, folded after the loop
            if (c0 == s) { my_key = key; my_w = t; }
            // the two compensated sums in row order: lane 0 walks the chunk from LDS
            s_vol[w][lane] = (V)vol;
            s_pv[w][lane] = (P)((P)vw * (P)vol);                 // float64 * float32 -> float64; float32 * float32 -> float32
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                const int cnt = (int)(e - c0 < 64 ? e - c0 : 64);
                for (int j = 0; j < cnt; ++j) {
                    rs_kahan<V>(vs, vc, s_vol[w][j]);
                    rs_kahan<P>(ps, pc, s_pv[w][j]);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // ---- trades-weighted median of the rows' medians (io.py:933-946)
        {   // smallest key over the group
            uint64_t a = kmin;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const uint64_t b = __shfl_xor(a, o, 64); a = b < a ? b : a; }
            kmin = a;
        }
        const double cutoff = (double)tr * 0.5;                  // cum_w[-1] * 0.5 (integers in float64: exact)
        uint64_t ans = kmin;
        if (e > s && e - s <= 64 && cutoff > 0.0) {
            // the group fits the wave (one row per lane): W(size <= my size) for every lane at once -- a wave-uniform loop over
            // the rows, no cross-lane reduction inside -- and the answer is the smallest key whose W reaches the cutoff.  (The
            // bisection below costs 64 rounds of a masked wave sum: 5e7 one-second rows -> 1-minute bars 5.2 -> 3.0 ms.)
            const int cnt = (int)(e - s);
            int64_t wle = 0;
            for (int j = 0; j < cnt; ++j) {
                const uint64_t kj = (uint64_t)fmk_readlane((int64_t)my_key, j);
                const int64_t wj = fmk_readlane(my_w, j);
                wle += kj <= my_key ? wj : 0;
            }
            uint64_t cand = (lane < cnt && (double)wle >= cutoff) ? my_key : ~0ULL;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const uint64_t b = __shfl_xor(cand, o, 64); cand = b < cand ? b : cand; }
            ans = cand;
        } else if (e > s && cutoff > 0.0) {
            const bool small = false;
            uint64_t lo_k = 0, hi_k = ~0ULL;
            while (lo_k < hi_k) {
                const uint64_t mid = lo_k + ((hi_k - lo_k) >> 1);
                int64_t wle = 0;
                if (small) {
                    wle = my_key <= mid ? my_w : 0;
                } else {
                    for (int64_t i = s + lane; i < e; i += 64)
                        if (rs_key(median[i]) <= mid) wle += trades[i];
                }
                wle = fmk_dpp_reduce(wle, (int64_t)0, [](int64_t a, int64_t b) { return a + b; });
                if ((double)wle >= cutoff) hi_k = mid;
                else lo_k = mid + 1;
            }
            ans = lo_k;
        }
        if (lane == 0) {
            o_open[g] = f_open; o_high[g] = hi; o_low[g] = lo; o_close[g] = l_close;
            ((V *)o_volume)[g] = vs;
            o_trades[g] = tr;
            if constexpr (VF64 || WF64) o_vwap[g] = (float)((double)ps / (double)vs);   // float64 / float32 -> float64
            else o_vwap[g] = (float)ps / (float)vs;                                     // float32 / float32
            o_median[g] = e > s ? (float)rs_unkey(ans) : __int_as_float(0x7FC00000);
            o_valid[g] = have_open ? 1 : 0;                      // dropna(subset=["open"]) (io.py:948) is the host's
        }
    }
}

}  // namespace

extern "C" int fmk_resample_bars_dev(fmk_ctx *ctx, const int64_t *d_seg, int64_t n_groups, const double *d_open,
                                     const double *d_high, const double *d_low, const double *d_close, const void *d_volume,
                                     int volume_is_f64, const int64_t *d_trades, const void *d_vwap, int vwap_is_f64,
                                     const double *d_median, double *d_o_open, double *d_o_high, double *d_o_low,
                                     double *d_o_close, void *d_o_volume, int64_t *d_o_trades, float *d_o_vwap,
                                     float *d_o_median, uint8_t *d_o_valid)
{
    if (n_groups <= 0) return FMK_OK;
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    int64_t blocks = fmk_ceil_div(n_groups, 4);
    const int64_t cap = (int64_t)ctx->n_cu * 32;
    if (blocks > cap) blocks = cap;
#define RS_LAUNCH(A, B)                                                                                                   \
    k_resample<A, B><<<(unsigned)blocks, 256, 0, ctx->stream>>>(d_seg, n_groups, d_open, d_high, d_low, d_close, d_volume,  \
                                                                d_trades, d_vwap, d_median, d_o_open, d_o_high, d_o_low,  \
                                                                d_o_close, d_o_volume, d_o_trades, d_o_vwap, d_o_median,  \
                                                                d_o_valid)
    if (volume_is_f64) { if (vwap_is_f64) RS_LAUNCH(true, true); else RS_LAUNCH(true, false); }
    else { if (vwap_is_f64) RS_LAUNCH(false, true); else RS_LAUNCH(false, false); }
#undef RS_LAUNCH
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}

extern "C" int fmk_resample_bars(fmk_ctx *ctx, const int64_t *seg, int64_t n_groups, int64_t n_rows, const double *open,
                                 const double *high, const double *low, const double *close, const void *volume,
                                 int volume_is_f64, const int64_t *trades, const void *vwap, int vwap_is_f64,
                                 const double *median, double *o_open, double *o_high, double *o_low, double *o_close,
                                 void *o_volume, int64_t *o_trades, float *o_vwap, float *o_median, uint8_t *o_valid)
{
    if (n_groups <= 0) return FMK_OK;
    if (n_rows < 0 || seg[0] < 0 || seg[n_groups] > n_rows)
        return fmk_set_error(ctx, FMK_E_ARG, "resample: segment offsets outside the %lld rows", (long long)n_rows);
    const size_t vs = volume_is_f64 ? 8 : 4, ws = vwap_is_f64 ? 8 : 4;
    void *d[20] = {nullptr};
    int rc = FMK_OK;
    auto up = [&](int k, const void *h, size_t bytes) {
        if (rc != FMK_OK) return;
        rc = fmk_alloc(ctx, bytes ? bytes : 8, &d[k]);
        if (rc == FMK_OK && bytes) rc = fmk_h2d(ctx, d[k], h, bytes);
    };
    auto mk = [&](int k, size_t bytes) { if (rc == FMK_OK) rc = fmk_alloc(ctx, bytes ? bytes : 8, &d[k]); };
    const size_t R = (size_t)n_rows, G = (size_t)n_groups;
    up(0, seg, (G + 1) * 8); up(1, open, R * 8); up(2, high, R * 8); up(3, low, R * 8); up(4, close, R * 8);
    up(5, volume, R * vs); up(6, trades, R * 8); up(7, vwap, R * ws); up(8, median, R * 8);
    mk(9, G * 8); mk(10, G * 8); mk(11, G * 8); mk(12, G * 8); mk(13, G * vs); mk(14, G * 8); mk(15, G * 4); mk(16, G * 4);
    mk(17, G);
    if (rc == FMK_OK)
        rc = fmk_resample_bars_dev(ctx, (const int64_t *)d[0], n_groups, (const double *)d[1], (const double *)d[2],
                                   (const double *)d[3], (const double *)d[4], d[5], volume_is_f64, (const int64_t *)d[6], d[7],
                                   vwap_is_f64, (const double *)d[8], (double *)d[9], (double *)d[10], (double *)d[11],
                                   (double *)d[12], d[13], (int64_t *)d[14], (float *)d[15], (float *)d[16], (uint8_t *)d[17]);
    void *host[9] = {o_open, o_high, o_low, o_close, o_volume, o_trades, o_vwap, o_median, o_valid};
    const size_t hb[9] = {G * 8, G * 8, G * 8, G * 8, G * vs, G * 8, G * 4, G * 4, G};
    for (int k = 0; k < 9 && rc == FMK_OK; ++k) rc = fmk_d2h(ctx, host[k], d[9 + k], hb[k]);
    (void)hipStreamSynchronize(ctx->stream);
    for (int k = 0; k < 18; ++k) if (d[k]) fmk_free(ctx, d[k]);
    return rc;
}
