/*
 * fmk_oracle.c -- CPU restatement of the finmlkit tick->bar hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP
 * engine in finmlkit_amd/csrc.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product package never does.
 *
 * Every function restates one reference function in plain scalar C, in the
 * reference's own evaluation order (sequential accumulation, same rounding
 * points), following the semantics the reference has when its @njit
 * functions are *typed* (Numba): accumulators initialised with `0.0` are
 * float64.  The reference's pure-Python CI mode (NUMBA_DISABLE_JIT=1) agrees
 * with that whenever the amount column is float64, or float32 with exactly
 * representable partial sums (NumPy-2 promotion makes `0.0 + np.float32`
 * a float32 there) -- the golden fixtures in tests/golden are generated on
 * such inputs; see oracle/gen_golden.py and DESIGN.md "Oracle pinning".
 *
 * Parity status: PINNED, three ways.
 * (1) tests/test_oracle_golden.py checks every function here against fixtures
 *     produced by importing the reference itself on synthetic inputs
 *     (oracle/gen_golden.py -> the .npz files under tests/golden).
 * (2) tests/test_refcalls_oracle.py replays every call the reference's OWN
 *     tests make to these functions: oracle/record_reference_tests.py runs the
 *     14 reference test files that touch the path (116 tests) against the
 *     reference with the path's functions wrapped and stores arguments +
 *     results of the 160 calls (tests/golden/reference_test_calls.npz); 130
 *     replay here, 30 have no C counterpart (listed with reasons in the test).
 * (3) tests/test_refcalls_oracle.py also replays oracle/edge_sweep.py's 151
 *     degenerate inputs put to the reference (tests/golden/edge_calls.npz):
 *     137 comparable, 14 marked with the reason they are not.
 * Until (2) existed this header claimed a pin against the reference's
 * known-answer tests that had not been built; building it found two
 * deviations, both fixed here and in the HIP path: a one-element
 * bar_close_indices means zero bars for every reducer except comp_bar_ohlcv
 * (the reference has the length check only there), and
 * comp_bar_trade_size_features takes bars as slices, so an end index past the
 * array is clamped (this file used to read past the array instead).
 * (4) reference-made vectors at the sizes the fixtures of (1)-(3) do not reach (round 3): the trade-size reducer over bar lengths
 *     of 0 .. 90 000 ticks (oracle/gen_tradesize_lengths.py) and the four bar reducers on bars of 70 001 .. 194 999 ticks
 *     (oracle/gen_longbars.py).  Building (4) found that np.sum adds an array in chunks of 8 192 elements (orc_pairwise_f32 below):
 *     this file had used one pairwise tree, right up to 8 192 elements and an ulp off in ~30 % of the longer bars.
 * (5) live, in the build container: tools/fuzz_reference.py runs tools/fuzz_parity.py's random cases with the reference's modules
 *     in the package's place (tests/test_reference_live.py: 600 fixed-seed cases; campaigns of 11 500 more were clean).
 *
 * Citations are relative to /root/reference/.
 */
#ifdef _OPENMP
#include <omp.h>
#endif
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_E_ARG (-1)       /* ValueError in the reference */
#define ORC_E_CAPACITY (-2)  /* caller buffer too small */
#define ORC_E_LEVEL (-3)     /* "Invalid price level index" base.py:719 */
#define ORC_E_ZERODIV (-4)   /* ZeroDivisionError base.py:536 */
#define ORC_E_NOMEM (-5)

/* ------------------------------------------------------------------ */
/* Synthetic tick stream; shared definition with the  */
/* device generator in finmlkit_amd/csrc/fmk_synth.hip.                */
/* ------------------------------------------------------------------ */
static inline uint64_t orc_mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

#define ORC_T0 1700000000000000000LL
#define ORC_K0 1000000LL

/* Ticks [first, first+n) of the stream `seed`; gap_mod = 100000000 for the
 * dense stream, 500000000000 for the sparse (empty-bar) variant. */
int orc_synth(uint64_t seed, int64_t first, int64_t n, uint64_t gap_mod,
              int64_t *ts, double *price, float *amount, int8_t *side)
{
    if (first < 0 || n < 0 || gap_mod == 0) return ORC_E_ARG;
    int64_t t = ORC_T0, k = ORC_K0;
    for (int64_t i = 0; i < first + n; ++i) {
        uint64_t h = orc_mix64(seed + (uint64_t)i);
        t += 1 + (int64_t)(h % gap_mod);
        unsigned b = (unsigned)(h >> 62);
        k += (b == 3) - (b == 0);
        if (i >= first) {
            int64_t o = i - first;
            if (ts) ts[o] = t;
            if (price) price[o] = (double)k * 0.01;
            if (amount) amount[o] = (float)(1 + ((h >> 8) & 4095)) * 0.0009765625f;
            if (side) side[o] = ((h >> 40) & 1) ? 1 : -1;
        }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------ */
/* NumPy arithmetic the reference leans on                             */
/* ------------------------------------------------------------------ */

/* npy_floor_divide for doubles (what `np.int64 // float` evaluates). */
static double orc_npy_floor_divide(double a, double b)
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

/* np.sum over a contiguous float32 / float64 array.  Two layers (numpy 2.2):
 *  - the ufunc reduction hands the inner loop at most NPY_BUFSIZE = 8192 elements at a time, buffered or not, and adds the
 *    chunks' results one after the other: ((c0 + c1) + c2) + ...   (found in round 3 when reference-made vectors for bars of more
 *    than 8192 ticks were added: np.mean of a float32 slice of 12 000 elements differs from the whole-array tree in ~30 % of the
 *    bars by one ulp; oracle/gen_tradesize_lengths.py, tests/golden/trade_size_lengths_reference.npz);
 *  - inside a chunk: pairwise summation with the 8-accumulator leaf (numpy/_core/src/umath/loops_utils.h.src, pairwise_sum). */
#define ORC_NP_BUFSIZE 8192
static float orc_pairwise_tree_f32(const float *a, int64_t n)
{
    if (n < 8) {
        float r = 0.f;
        for (int64_t i = 0; i < n; ++i) r += a[i];
        return r;
    } else if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return orc_pairwise_tree_f32(a, n2) + orc_pairwise_tree_f32(a + n2, n - n2);
    }
}
static float orc_pairwise_f32(const float *a, int64_t n)
{
    if (n <= ORC_NP_BUFSIZE) return orc_pairwise_tree_f32(a, n);
    float r = orc_pairwise_tree_f32(a, ORC_NP_BUFSIZE);
    for (int64_t i = ORC_NP_BUFSIZE; i < n; i += ORC_NP_BUFSIZE)
        r += orc_pairwise_tree_f32(a + i, n - i < ORC_NP_BUFSIZE ? n - i : ORC_NP_BUFSIZE);
    return r;
}

static double orc_pairwise_tree_f64(const double *a, int64_t n)
{
    if (n < 8) {
        double r = 0.;
        for (int64_t i = 0; i < n; ++i) r += a[i];
        return r;
    } else if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return orc_pairwise_tree_f64(a, n2) + orc_pairwise_tree_f64(a + n2, n - n2);
    }
}
static double orc_pairwise_f64(const double *a, int64_t n)
{
    if (n <= ORC_NP_BUFSIZE) return orc_pairwise_tree_f64(a, n);
    double r = orc_pairwise_tree_f64(a, ORC_NP_BUFSIZE);
    for (int64_t i = ORC_NP_BUFSIZE; i < n; i += ORC_NP_BUFSIZE)
        r += orc_pairwise_tree_f64(a + i, n - i < ORC_NP_BUFSIZE ? n - i : ORC_NP_BUFSIZE);
    return r;
}

static int orc_cmp_f64(const void *x, const void *y)
{
    double a = *(const double *)x, b = *(const double *)y;
    return (a > b) - (a < b);
}

/* np.median of a float64 vector (scratch is clobbered). */
static double orc_median(double *scratch, int64_t n)
{
    if (n <= 0) return 0.0;
    for (int64_t i = 0; i < n; ++i)
        if (isnan(scratch[i])) return NAN;
    qsort(scratch, (size_t)n, sizeof(double), orc_cmp_f64);
    if (n & 1) return scratch[n / 2];
    return (scratch[n / 2 - 1] + scratch[n / 2]) / 2.0;
}

/* np.percentile(x, q) with the default 'linear' method, NumPy 2.2 (numpy/lib/_function_base_impl.py: percentile divides q
 * by a.dtype.type(100); _QuantileMethods['linear'] virtual index (n - 1) * q; _get_indexes; _get_gamma; _lerp) -- every
 * step in the ARRAY's dtype.  For a float32 array that is float32 arithmetic throughout (as_f32: the values in scratch
 * are float32 values held in doubles); validated bit for bit against np.percentile for n = 1..3000 in both dtypes while
 * this was written.  The reference's trade-size slices are float32 after TradesData's merge (utils.py merge_split_trades
 * returns float32 amounts).  What Numba's own np.percentile does for float32 input could not be run here.
 * scratch is clobbered. */
static double orc_percentile_f32(double *scratch, int64_t n, float q)
{
    float q32 = q / 100.0f;
    float vi = (float)(n - 1) * q32;
    if (vi >= (float)(n - 1)) return scratch[n - 1];
    float fl = floorf(vi);
    int64_t lo = (int64_t)fl;
    float t = vi - fl;
    float a = (float)scratch[lo], b = (float)scratch[lo + 1];
    float d = b - a;
    float m = d * t;
    float r = a + m;
    if (t >= 0.5f) { float u = 1.0f - t; float m2 = d * u; r = b - m2; }
    return (double)r;
}

static double orc_percentile(double *scratch, int64_t n, double q, int as_f32)
{
    for (int64_t i = 0; i < n; ++i)
        if (isnan(scratch[i])) return NAN;
    qsort(scratch, (size_t)n, sizeof(double), orc_cmp_f64);
    if (as_f32) return orc_percentile_f32(scratch, n, (float)q);
    double vidx = (q / 100.0) * (double)(n - 1);
    double fl = floor(vidx);
    int64_t lo = (int64_t)fl;
    int64_t hi = lo + 1 < n ? lo + 1 : n - 1;
    double t = vidx - fl;
    double a = scratch[lo], b = scratch[hi];
    double d = b - a;
    double r = a + d * t;
    if (t >= 0.5) r = b - d * (1.0 - t);
    if (d == 0.0) r = a;
    return r;
}

/* searchsorted(ts, key, side='right') on int64 */
static int64_t orc_upper_bound_i64(const int64_t *a, int64_t n, int64_t key)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] <= key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* ------------------------------------------------------------------ */
/* Row 1: _time_bar_indexer   finmlkit/bar/logic.py:12-51              */
/* ------------------------------------------------------------------ */

/* The float64 clock construction of logic.py:30-39, as NumPy evaluates
 * it: start = ts[0] // I * I; last = ceil(ts[-1]/I)*I;
 * arange(start, last + I + 1, I, dtype=int64) whose fill rule is
 * first=int64(start), delta=int64(start+I)-first, v[i]=first+i*delta. */
int orc_time_bar_clock(int64_t ts_first, int64_t ts_last, double interval_seconds,
                       int64_t *n_edges, int64_t *first_edge, int64_t *delta)
{
    double I = interval_seconds * 1e9;
    if (!(I > 0.0)) return ORC_E_ARG;
    double start = orc_npy_floor_divide((double)ts_first, I) * I;
    double last = ceil((double)ts_last / I) * I;
    double stop = last + I + 1.0;
    double len = ceil((stop - start) / I);
    if (len <= 0) { *n_edges = 0; *first_edge = 0; *delta = 0; return ORC_OK; }
    *n_edges = (int64_t)len;
    *first_edge = (int64_t)start;
    *delta = (int64_t)(start + I) - (int64_t)start;
    return ORC_OK;
}

int orc_time_bar_indexer(const int64_t *ts, int64_t n, double interval_seconds,
                         int64_t *clock, int64_t *close_idx, int64_t capacity,
                         int64_t *n_edges_out)
{
    if (n <= 0) return ORC_E_ARG;
    int64_t ne, e0, d;
    int rc = orc_time_bar_clock(ts[0], ts[n - 1], interval_seconds, &ne, &e0, &d);
    if (rc) return rc;
    *n_edges_out = ne;
    if (!clock || !close_idx) return ORC_OK;
    if (capacity < ne) return ORC_E_CAPACITY;
    for (int64_t i = 0; i < ne; ++i) {
        int64_t e = (i == 0) ? e0 : e0 + i * d;
        clock[i] = e;
        close_idx[i] = orc_upper_bound_i64(ts, n, e) - 1;   /* logic.py:42 */
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------ */
/* Rows 2-4: tick / volume / dollar indexers  logic.py:54-149          */
/* Each returns the number of close indices written (first is 0).      */
/* ------------------------------------------------------------------ */
int64_t orc_tick_bar_indexer(int64_t n, int64_t threshold, int64_t *out, int64_t cap)
{
    int64_t m = 0;
    if (n <= 0) return ORC_E_ARG;
    if (m < cap) out[m] = 0;
    ++m;                                             /* logic.py:74 */
    int64_t cum = 1;                                 /* logic.py:76 */
    for (int64_t i = 1; i < n; ++i) {
        cum += 1;
        if (cum >= threshold) {                      /* logic.py:80 */
            if (m < cap) out[m] = i;
            ++m;
            cum = 0;                                 /* logic.py:82 */
        }
    }
    return m;
}

int64_t orc_volume_bar_indexer(const void *volumes, int is_f64, int64_t n,
                               double threshold, int64_t *out, int64_t cap)
{
    if (n <= 0) return ORC_E_ARG;
    const float *vf = (const float *)volumes;
    const double *vd = (const double *)volumes;
    int64_t m = 0;
    if (m < cap) out[m] = 0;
    ++m;
    double cum = is_f64 ? vd[0] : (double)vf[0];     /* logic.py:108 */
    for (int64_t i = 1; i < n; ++i) {
        cum += is_f64 ? vd[i] : (double)vf[i];
        if (cum >= threshold) {                      /* logic.py:111 */
            if (m < cap) out[m] = i;
            ++m;
            cum = 0.0;                               /* logic.py:113: reset */
        }
    }
    return m;
}

int64_t orc_dollar_bar_indexer(const double *prices, const void *volumes, int is_f64,
                               int64_t n, double threshold, int64_t *out, int64_t cap)
{
    if (n <= 0) return ORC_E_ARG;
    const float *vf = (const float *)volumes;
    const double *vd = (const double *)volumes;
    int64_t m = 0;
    if (m < cap) out[m] = 0;
    ++m;
    double cum = prices[0] * (is_f64 ? vd[0] : (double)vf[0]);   /* logic.py:142 */
    for (int64_t i = 1; i < n; ++i) {
        double d = prices[i] * (is_f64 ? vd[i] : (double)vf[i]);
        cum = cum + d;
        if (cum >= threshold) {                      /* logic.py:145 */
            if (m < cap) out[m] = i;
            ++m;
            cum = cum - threshold;                   /* logic.py:147: carry */
        }
    }
    return m;
}

/* CUSUM indexer logic.py:152-221 ("next" row; sigma is forward-filled IN
 * PLACE exactly like the reference does). */
int64_t orc_cusum_bar_indexer(const int64_t *ts, const double *prices, double *sigma,
                              int64_t n, double sigma_floor, double sigma_mult,
                              int64_t *out, int64_t cap)
{
    if (n <= 0) return ORC_E_ARG;
    int64_t first = 0;
    for (int64_t i = 0; i < n; ++i)
        if (!isnan(sigma[i])) { first = i; break; }
    for (int64_t i = first; i < n; ++i)
        if (isnan(sigma[i])) sigma[i] = sigma[i - 1 < 0 ? n - 1 : i - 1];
    int64_t m = 0;
    if (m < cap) out[m] = first;
    ++m;
    double s_pos = 0.0, s_neg = 0.0;
    int64_t i = first + 1;
    while (i < n) {
        double ret = log(prices[i] / prices[i - 1]);
        s_pos = fmax(0.0, s_pos + ret);
        s_neg = fmin(0.0, s_neg + ret);
        if (i + 1 < n && ts[i] == ts[i + 1]) { ++i; continue; }   /* logic.py:206-209 */
        double lam = sigma_mult * sigma[i];
        if (!(lam > sigma_floor)) lam = sigma_floor;
        if (isnan(sigma_mult * sigma[i])) lam = sigma_mult * sigma[i];
        if (s_pos >= lam) { if (m < cap) out[m] = i; ++m; s_pos = 0.0; }
        else if (s_neg <= -lam) { if (m < cap) out[m] = i; ++m; s_neg = 0.0; }
        ++i;
    }
    return m;
}

/* ------------------------------------------------------------------ */
/* Row 5: comp_bar_ohlcv   finmlkit/bar/base.py:306-407                */
/* ------------------------------------------------------------------ */
static inline int64_t orc_wrap(int64_t i, int64_t n) { return i < 0 ? i + n : i; }

/* ORC_THREADS > 1 in the environment: the per-bar loops of rows 5-7 run as OpenMP parallel-for over BARS.  Bars are
 * independent in the reference (comp_bar_ohlcv / directional are numba.prange loops, base.py:349, 468; comp_bar_footprints
 * is a serial loop over independent bars, base.py:682) and the arithmetic inside a bar is untouched, so results do not
 * depend on the thread count (tests/test_oracle_golden.py runs the goldens with 1 and 4 threads). */
static int orc_threads(void)
{
    const char *e = getenv("ORC_THREADS");
    return (e && atoi(e) > 1) ? atoi(e) : 1;
}

int orc_comp_bar_ohlcv(const double *prices, const void *volumes, int is_f64, int64_t n,
                       const int64_t *close_idx, int64_t n_idx,
                       double *o, double *h, double *l, double *c, float *vol,
                       double *vwap, int64_t *trades, double *median)
{
    if (n_idx < 2) return ORC_E_ARG;                 /* base.py:334-335 */
    const float *vf = (const float *)volumes;
    const double *vd = (const double *)volumes;
    int64_t nb = n_idx - 1;
    int64_t maxcnt = 1;
    for (int64_t i = 0; i < nb; ++i) {
        int64_t cnt = close_idx[i + 1] - close_idx[i];
        if (cnt > maxcnt) maxcnt = cnt;
    }
    /* Bars are independent (the reference runs them under numba.prange, base.py:349): with ORC_THREADS > 1 in the
     * environment the bar loop is an OpenMP parallel-for -- the arithmetic inside a bar is untouched, so results do
     * not depend on the thread count.  Used by bench.py's cpu_baseline on all host cores. */
    int nthreads = orc_threads();
    double *scratch_all = median ? (double *)malloc(sizeof(double) * (size_t)maxcnt * (size_t)nthreads) : NULL;
    if (median && !scratch_all) return ORC_E_NOMEM;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
    for (int64_t i = 0; i < nb; ++i) {
#ifdef _OPENMP
        double *scratch = scratch_all ? scratch_all + (size_t)omp_get_thread_num() * (size_t)maxcnt : NULL;
#else
        double *scratch = scratch_all;
#endif
        int64_t start = close_idx[i], end = close_idx[i + 1];
        if (start == end) {                          /* base.py:352-361 */
            double p = prices[orc_wrap(end, n)];
            o[i] = h[i] = l[i] = c[i] = p;
            vol[i] = 0.f; vwap[i] = 0.0; trades[i] = 0;
            if (median) median[i] = 0.0;
            continue;
        }
        start += 1;
        double hi = prices[start], lo = prices[start];
        double tv = 0.0, td = 0.0;
        int64_t cnt = end - start + 1;
        for (int64_t j = start; j <= end; ++j) {     /* base.py:377-391 */
            double p = prices[j];
            double v = is_f64 ? vd[j] : (double)vf[j];
            if (scratch) scratch[j - start] = v;
            if (p > hi) hi = p;
            if (p < lo) lo = p;
            tv += v;
            td += p * v;
        }
        o[i] = prices[start]; c[i] = prices[end]; h[i] = hi; l[i] = lo;
        vol[i] = (float)tv;
        vwap[i] = tv > 0 ? td / tv : 0.0;            /* base.py:398 */
        trades[i] = cnt;
        if (median) median[i] = cnt > 0 ? orc_median(scratch, cnt) : 0.0;
    }
    free(scratch_all);
    return ORC_OK;
}

/* ------------------------------------------------------------------ */
/* Row 6: comp_bar_directional_features   base.py:409-546              */
/* Returns ORC_E_ZERODIV (after filling everything else, mean_spread   */
/* = NaN for the offending bars) when a bar has no signed tick.        */
/* ------------------------------------------------------------------ */
int orc_comp_bar_directional(const double *prices, const void *volumes, int is_f64,
                             int64_t n, const int64_t *close_idx, int64_t n_idx,
                             const int8_t *sides,
                             int64_t *ticks_buy, int64_t *ticks_sell,
                             float *volume_buy, float *volume_sell,
                             float *dollars_buy, float *dollars_sell,
                             float *mean_spread, float *max_spread,
                             int64_t *cum_ticks_min, int64_t *cum_ticks_max,
                             float *cum_volumes_min, float *cum_volumes_max,
                             float *cum_dollars_min, float *cum_dollars_max)
{
    if (n_idx == 1) return ORC_OK;                   /* zero bars: base.py:409-546 has no length check */
    if (n_idx < 1) return ORC_E_ARG;
    const float *vf = (const float *)volumes;
    const double *vd = (const double *)volumes;
    int64_t nb = n_idx - 1;
    int rc = ORC_OK;
    const int nthreads = orc_threads();
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads) reduction(min : rc)
#endif
    for (int64_t i = 0; i < nb; ++i) {
        int64_t start = close_idx[i] + 1, end = close_idx[i + 1];
        int64_t tb = 0, tsell = 0, ct = 0;
        double vb = 0, vs = 0, db = 0, ds = 0, cv = 0, cd = 0;
        double mxs = 0.0, cs = 0.0;
        int64_t ctmin = 1000000000LL, ctmax = -1000000000LL;       /* base.py:459-460 */
        double cvmin = 1e9, cvmax = -1e9, cdmin = 1e9, cdmax = -1e9;
        int prev = (end > start) ? sides[orc_wrap(start - 1, n)] : 0;   /* base.py:485-488 */
        for (int64_t j = start; j <= end; ++j) {
            int cur = sides[j];
            if (cur != prev) {                                      /* base.py:495-500 */
                double sp = fabs(prices[j] - prices[orc_wrap(j - 1, n)]);
                if (sp > mxs) mxs = sp;
                cs += sp;
            }
            prev = cur;
            double v = is_f64 ? vd[j] : (double)vf[j];
            double pv = prices[j] * v;
            if (cur == 1) {
                tb += 1; vb += v; db += pv; ct += 1; cv += v; cd += pv;
            } else if (cur == -1) {
                tsell += 1; vs += v; ds += pv; ct -= 1; cv -= v; cd -= pv;
            } else {
                continue;                                           /* base.py:518-519 */
            }
            if (ct > ctmax) ctmax = ct;
            if (ct < ctmin) ctmin = ct;
            if (cv > cvmax) cvmax = cv;
            if (cv < cvmin) cvmin = cv;
            if (cd > cdmax) cdmax = cd;
            if (cd < cdmin) cdmin = cd;
        }
        ticks_buy[i] = tb; ticks_sell[i] = tsell;
        volume_buy[i] = (float)vb; volume_sell[i] = (float)vs;
        dollars_buy[i] = (float)db; dollars_sell[i] = (float)ds;
        max_spread[i] = (float)mxs;
        if (tb + tsell == 0) { mean_spread[i] = NAN; if (ORC_E_ZERODIV < rc) rc = ORC_E_ZERODIV; }   /* base.py:536 (codes < 0) */
        else mean_spread[i] = (float)(cs / (double)(tb + tsell));
        cum_ticks_min[i] = ctmin; cum_ticks_max[i] = ctmax;
        cum_volumes_min[i] = (float)cvmin; cum_volumes_max[i] = (float)cvmax;
        cum_dollars_min[i] = (float)cdmin; cum_dollars_max[i] = (float)cdmax;
    }
    return rc;
}

/* ------------------------------------------------------------------ */
/* "next" row 1: comp_bar_trade_size_features  base.py:549-612         */
/* ------------------------------------------------------------------ */
int orc_comp_bar_trade_size(const void *amounts, int is_f64, int64_t n, const double *theta,
                            const int64_t *close_idx, int64_t n_idx, double theta_mult,
                            float *mean_size_rel, float *size_95_rel, float *pct_block,
                            float *size_gini)
{
    if (n_idx == 1) return ORC_OK;                   /* zero bars: base.py:549-612 checks theta's length only */
    if (n_idx < 1) return ORC_E_ARG;
    const float *vf = (const float *)amounts;
    const double *vd = (const double *)amounts;
    int64_t nb = n_idx - 1;
    int64_t maxcnt = 1;
    for (int64_t i = 0; i < nb; ++i) {
        int64_t cnt = close_idx[i + 1] - close_idx[i];
        if (cnt > maxcnt) maxcnt = cnt;
    }
    /* bars are independent (numba.prange in the reference, base.py:581): with ORC_THREADS > 1 the bar loop is an OpenMP parallel-for
     * with per-thread scratch, the arithmetic inside a bar untouched (the full-size parity test runs it on all host cores) */
    int nthreads = orc_threads();
    double *sd_all = (double *)malloc(sizeof(double) * (size_t)maxcnt * (size_t)nthreads);
    float *sf_all = (float *)malloc(sizeof(float) * (size_t)maxcnt * (size_t)nthreads);
    double *sq_all = (double *)malloc(sizeof(double) * (size_t)maxcnt * (size_t)nthreads);
    if (!sd_all || !sf_all || !sq_all) { free(sd_all); free(sf_all); free(sq_all); return ORC_E_NOMEM; }
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads)
#endif
    for (int64_t i = 0; i < nb; ++i) {
#ifdef _OPENMP
        const size_t tnum = (size_t)omp_get_thread_num();
#else
        const size_t tnum = 0;
#endif
        double *sd = sd_all + tnum * (size_t)maxcnt;
        float *sf = sf_all + tnum * (size_t)maxcnt;
        double *sq = sq_all + tnum * (size_t)maxcnt;
        mean_size_rel[i] = size_95_rel[i] = pct_block[i] = size_gini[i] = NAN;
        int64_t start = close_idx[i] + 1, end = close_idx[i + 1];
        if (start > end) continue;                   /* base.py:584, on the raw indices */
        if (theta[i] == 0.0) continue;
        double thr = theta[i] * theta_mult;
        /* base.py:590 takes a SLICE, amounts[start:end + 1]: an end past the array is clamped (the reference's own
         * test_block_volume passes end == len(amounts)); a slice left empty gives mean([]) = NaN and total 0 -> NaN row */
        if (end > n - 1) end = n - 1;
        int64_t cnt = end - start + 1;
        if (cnt <= 0) continue;
        /* np.mean / .sum over the slice: pairwise in the slice dtype
         * (float32 slices use a float32 pairwise sum, mean divides in
         * float32 -- NumPy semantics of the reference's CI mode). */
        double mean, total;
        if (is_f64) {
            total = orc_pairwise_f64(vd + start, cnt);
            mean = total / (double)cnt;
        } else {
            float tf = orc_pairwise_f32(vf + start, cnt);
            total = (double)tf;
            mean = (double)(float)(tf / (float)cnt);
        }
        for (int64_t j = 0; j < cnt; ++j) sd[j] = is_f64 ? vd[start + j] : (double)vf[start + j];
        double p95 = orc_percentile(sd, cnt, 95.0, !is_f64);
        if (is_f64) {
            mean_size_rel[i] = (float)log1p(mean / thr);
            size_95_rel[i] = (float)log1p(p95 / thr);
        } else {
            mean_size_rel[i] = (float)log1p(mean / thr);
            size_95_rel[i] = (float)log1p(p95 / thr);
        }
        if (total == 0) continue;
        double block = 0.0;
        for (int64_t j = start; j <= end; ++j) {
            double a = is_f64 ? vd[j] : (double)vf[j];
            if (a > thr) block += a;
        }
        pct_block[i] = (float)(block / total);
        if (cnt == 1) { size_gini[i] = 0.f; continue; }
        if (is_f64) {
            for (int64_t j = 0; j < cnt; ++j) { double q = vd[start + j] / total; sq[j] = q * q; }
            size_gini[i] = (float)(1.0 - orc_pairwise_f64(sq, cnt));
        } else {
            float tf = (float)total;
            for (int64_t j = 0; j < cnt; ++j) { float q = vf[start + j] / tf; sf[j] = q * q; }
            size_gini[i] = 1.0f - orc_pairwise_f32(sf, cnt);
        }
    }
    free(sd_all); free(sf_all); free(sq_all);
    return ORC_OK;
}

/* ------------------------------------------------------------------ */
/* Row 8: comp_footprint_features   base.py:755-850                    */
/* ------------------------------------------------------------------ */
static void orc_footprint_features(const int32_t *levels, const float *buy, const float *sell,
                                   int64_t L, double mult,
                                   uint8_t *buy_imb, uint8_t *sell_imb,
                                   int *max_run_signed, int32_t *cot, double *skew, double *gini,
                                   float *tmpf, double *tmpd)
{
    /* array(float32) * float64 scalar: Numba's typed semantics (the production path) promote to float64, the product is
     * exact to ~1e-16; NumPy >= 2 (NEP 50, the pure-Python mode the fixtures are recorded in) would round it to float32.
     * The two differ only when the float32-rounded product crosses the other volume (decimal lots: 0.3f > 0.1f * 3.0 is
     * False in float32 and True in float64); every recorded reference call has exact products, where they agree. */
    for (int64_t k = 0; k < L; ++k) buy_imb[k] = sell_imb[k] = 0;
    if (L > 1) {                                                    /* base.py:795-798 */
        for (int64_t k = 0; k < L - 1; ++k) sell_imb[k] = (double)sell[k] > (double)buy[k + 1] * mult;
        for (int64_t k = 1; k < L; ++k) buy_imb[k] = (double)buy[k] > (double)sell[k - 1] * mult;
    }
    int max_run = 0, max_sign = 0, run = 0, run_sign = 0;           /* base.py:801-819 */
    for (int64_t k = 0; k < L; ++k) {
        int sign = buy_imb[k] ? 1 : (sell_imb[k] ? -1 : 0);
        if (sign != 0 && sign == run_sign) run += 1;
        else if (sign != 0) { run = 1; run_sign = sign; }
        else { run = 0; run_sign = 0; }
        if (run > max_run) { max_run = run; max_sign = run_sign; }
    }
    *max_run_signed = max_run * max_sign;
    for (int64_t k = 0; k < L; ++k) tmpf[k] = buy[k] + sell[k];     /* base.py:822 */
    float total = orc_pairwise_f32(tmpf, L);
    int64_t arg = 0;                                                /* np.argmax: first maximum; the first NaN counts as one */
    for (int64_t k = 1; k < L; ++k) if (tmpf[k] > tmpf[arg] || (tmpf[k] != tmpf[k] && tmpf[arg] == tmpf[arg])) arg = k;
    *cot = levels[arg];
    *skew = 0.0; *gini = 0.0;
    if (total > 0 && L > 0) {                                       /* base.py:836-848 */
        for (int64_t k = 0; k < L; ++k) tmpd[k] = (double)levels[k] * (double)tmpf[k];
        double vwap = orc_pairwise_f64(tmpd, L) / (double)total;
        double dot = 0.0;
        for (int64_t k = 0; k < L; ++k) dot += ((double)levels[k] - vwap) * (double)tmpf[k];
        *skew = dot / (double)total;
        for (int64_t k = 0; k < L; ++k) { float q = tmpf[k] / total; tmpf[k] = q * q; }
        *gini = (double)(1.0f - orc_pairwise_f32(tmpf, L));
    }
}

/* Standalone entry for comp_footprint_features (tests of row 8). */
int orc_comp_footprint_features(const int32_t *levels, const float *buy, const float *sell,
                                int64_t L, double mult, uint8_t *buy_imb, uint8_t *sell_imb,
                                int32_t *max_run_signed, int32_t *cot, double *skew, double *gini)
{
    if (L <= 0) return ORC_E_ARG;
    float *tf = (float *)malloc(sizeof(float) * (size_t)L);
    double *td = (double *)malloc(sizeof(double) * (size_t)L);
    if (!tf || !td) { free(tf); free(td); return ORC_E_NOMEM; }
    int run;
    orc_footprint_features(levels, buy, sell, L, mult, buy_imb, sell_imb, &run, cot, skew, gini, tf, td);
    *max_run_signed = run;
    free(tf); free(td);
    return ORC_OK;
}

/* ------------------------------------------------------------------ */
/* Row 7: comp_bar_footprints   base.py:615-752, CSR output            */
/* Phase 1 (flat arrays NULL): fills level_offsets[n_bars+1].          */
/* Phase 2: fills the flat arrays (length level_offsets[n_bars]) and   */
/* the per-bar arrays.                                                 */
/* ------------------------------------------------------------------ */
int orc_comp_bar_footprints(const double *prices, const void *amounts, int is_f64, int64_t n,
                            const int64_t *close_idx, int64_t n_idx, const int8_t *sides,
                            double price_tick_size, const double *bar_lows,
                            const double *bar_highs, double imbalance_factor,
                            int64_t *level_offsets,
                            int32_t *price_levels, float *buy_vol, float *sell_vol,
                            int32_t *buy_ticks, int32_t *sell_ticks,
                            uint8_t *buy_imb, uint8_t *sell_imb,
                            uint16_t *buy_imb_sum, uint16_t *sell_imb_sum,
                            int32_t *cot, int16_t *max_run, double *vp_skew, double *vp_gini)
{
    (void)n;
    /* base.py:615-752 has no length check on bar_close_indices: one element = zero bars = empty outputs
     * (pinned by the reference's tests/bars/test_comp_bar_footprints.py::test_comp_bar_footprints_empty_bar) */
    if (n_idx == 1) { if (level_offsets) level_offsets[0] = 0; return ORC_OK; }
    if (n_idx < 1) return ORC_E_ARG;
    const float *vf = (const float *)amounts;
    const double *vd = (const double *)amounts;
    int64_t nb = n_idx - 1;
    int64_t off = 0, maxL = 1;
    for (int64_t i = 0; i < nb; ++i) {
        level_offsets[i] = off;
        int64_t low = (int64_t)nearbyint(bar_lows[i] / price_tick_size);    /* base.py:688-689 */
        int64_t high = (int64_t)nearbyint(bar_highs[i] / price_tick_size);
        int64_t L = high - low + 1;
        if (L < 0) L = 0;
        if (L > maxL) maxL = L;
        off += L;
    }
    level_offsets[nb] = off;
    if (!price_levels) return ORC_OK;
    const int nthreads = orc_threads();
    float *tf_all = (float *)malloc(sizeof(float) * (size_t)maxL * (size_t)nthreads);
    double *td_all = (double *)malloc(sizeof(double) * (size_t)maxL * (size_t)nthreads);
    if (!tf_all || !td_all) { free(tf_all); free(td_all); return ORC_E_NOMEM; }
    int bad_level = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads) reduction(| : bad_level)
#endif
    for (int64_t i = 0; i < nb; ++i) {
#ifdef _OPENMP
        float *tf = tf_all + (size_t)omp_get_thread_num() * (size_t)maxL;
        double *td = td_all + (size_t)omp_get_thread_num() * (size_t)maxL;
#else
        float *tf = tf_all;
        double *td = td_all;
#endif
        int bar_bad = 0;
        int64_t start = close_idx[i] + 1, end = close_idx[i + 1];
        int64_t low = (int64_t)nearbyint(bar_lows[i] / price_tick_size);
        int64_t base = level_offsets[i], L = level_offsets[i + 1] - base;
        for (int64_t k = 0; k < L; ++k) {
            price_levels[base + k] = (int32_t)(low + k);
            buy_vol[base + k] = sell_vol[base + k] = 0.f;
            buy_ticks[base + k] = sell_ticks[base + k] = 0;
        }
        for (int64_t j = start; j <= end; ++j) {                    /* base.py:700-719 */
            int64_t lvl = (int64_t)nearbyint(prices[j] / price_tick_size) - low;
            if (lvl < 0 || lvl >= L) { bar_bad = 1; break; }             /* base.py:719 raises: reported after the loop */
            int sd = sides[j];
            if (sd == 1) {
                /* float32 element += amount: rounded to float32 on every add */
                buy_vol[base + lvl] = is_f64 ? (float)((double)buy_vol[base + lvl] + vd[j])
                                             : buy_vol[base + lvl] + vf[j];
                buy_ticks[base + lvl] += 1;
            } else if (sd == -1) {
                sell_vol[base + lvl] = is_f64 ? (float)((double)sell_vol[base + lvl] + vd[j])
                                              : sell_vol[base + lvl] + vf[j];
                sell_ticks[base + lvl] += 1;
            }
        }
        if (bar_bad) { bad_level |= 1; continue; }
        if (L <= 0) {   /* cannot happen with consistent highs/lows */
            buy_imb_sum[i] = sell_imb_sum[i] = 0; cot[i] = 0; max_run[i] = 0;
            vp_skew[i] = vp_gini[i] = 0.0;
            continue;
        }
        int run; int32_t c; double sk, gi;
        orc_footprint_features(price_levels + base, buy_vol + base, sell_vol + base, L,
                               imbalance_factor, buy_imb + base, sell_imb + base,
                               &run, &c, &sk, &gi, tf, td);
        unsigned bs = 0, ss = 0;
        for (int64_t k = 0; k < L; ++k) { bs += buy_imb[base + k]; ss += sell_imb[base + k]; }
        buy_imb_sum[i] = (uint16_t)bs; sell_imb_sum[i] = (uint16_t)ss;   /* base.py:738-739 */
        cot[i] = c; max_run[i] = (int16_t)run; vp_skew[i] = sk; vp_gini[i] = gi;
    }
    free(tf_all); free(td_all);
    return bad_level ? ORC_E_LEVEL : ORC_OK;
}

/* ------------------------------------------------------------------ */
/* Row 9: comp_lagged_returns   finmlkit/feature/core/utils.py:12-64   */
/* searchsorted(int64 array, float64 key) compares in float64.         */
/* ------------------------------------------------------------------ */
static int64_t orc_ss_f64key(const int64_t *a, int64_t n, double key, int right)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = lo + ((hi - lo) >> 1);
        double v = (double)a[mid];
        int go = right ? (v <= key) : (v < key);
        if (go) lo = mid + 1; else hi = mid;
    }
    return lo;
}

int orc_comp_lagged_returns(const int64_t *ts, const double *close, int64_t n,
                            double window_sec, int is_log, double *out)
{
    if (!(window_sec > 0)) return ORC_E_ARG;          /* utils.py:33-34 */
    for (int64_t i = 0; i < n; ++i) out[i] = NAN;
    if (n == 0) return ORC_OK;
    double w = window_sec * 1e9;
    int64_t start_idx = orc_ss_f64key(ts, n, (double)ts[0] + w, 0);    /* utils.py:42 */
    for (int64_t i = start_idx; i < n; ++i) {
        double target = (double)ts[i] - w;
        int64_t lag = orc_ss_f64key(ts, n, target, 1) - 1;             /* utils.py:46 */
        if (lag >= 0 && lag < i) {
            if (close[lag] != 0.0)
                out[i] = is_log ? log(close[i] / close[lag]) : close[i] / close[lag] - 1.0;
            else
                out[i] = INFINITY;
        }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------ */
/* Row 10: ewmst / ewmst_mean0 / ewms / realized_vol                   */
/* finmlkit/feature/core/volatility.py:9-286                           */
/* ------------------------------------------------------------------ */
int orc_ewmst(const int64_t *ts, const double *y, int64_t n, double half_life,
              double sigma_floor, double *out)
{
    if (n == 0) return ORC_OK;
    double V = 0, V2 = 0, Sy = 0, Syy = 0;
    int64_t last = ts[0];
    out[0] = NAN;
    for (int64_t i = 1; i < n; ++i) {
        double dt = (double)(ts[i] - last) / 1e9;
        last = ts[i];
        double alpha = 1.0 - exp(-dt / half_life);
        double om = 1.0 - alpha;
        double yi = y[i];
        V = alpha + om * V;
        V2 = alpha * alpha + (om * om) * V2;
        if (isnan(yi)) { Sy = om * Sy; Syy = om * Syy; }
        else { Sy = alpha * yi + om * Sy; Syy = alpha * yi * yi + om * Syy; }
        if (V > 0.0) {
            double mean = Sy / V, e2 = Syy / V;
            double var_raw = e2 - mean * mean;
            double denom = V - (V2 / V);
            double var = (denom > 0.0 && var_raw > 0.0) ? var_raw * (V / denom) : 0.0;
            double s = sqrt(var);
            if (s < sigma_floor) s = sigma_floor;
            out[i] = s;
        } else out[i] = NAN;
    }
    return ORC_OK;
}

int orc_ewmst_mean0(const int64_t *ts, const double *y, int64_t n, double half_life,
                    double sigma_floor, double *out)
{
    if (n == 0) return ORC_OK;
    double U = 0, V = 0;
    int64_t last = ts[0];
    out[0] = NAN;
    for (int64_t i = 1; i < n; ++i) {
        double dt = (double)(ts[i] - last) / 1e9;
        last = ts[i];
        double alpha = 1.0 - exp(-dt / half_life);
        double yt = y[i];
        if (isnan(yt)) { U = (1.0 - alpha) * U; V = (1.0 - alpha) * V; }
        else { U = alpha * (yt * yt) + (1.0 - alpha) * U; V = alpha + (1.0 - alpha) * V; }
        double var = V > 0.0 ? U / V : NAN;
        if (var < 0.0) var = 0.0;
        double s = sqrt(var);
        if (s < sigma_floor) s = sigma_floor;
        out[i] = s;
    }
    return ORC_OK;
}

int orc_ewms(const double *y, int64_t n, int64_t span, double *out)
{
    if (span <= 1) { for (int64_t i = 0; i < n; ++i) out[i] = NAN; return ORC_OK; }
    double alpha = 2.0 / ((double)span + 1.0), om = 1.0 - alpha;
    double om2 = om * om;   /* one_minus_alpha ** 2 */
    double Sw = 0, Sw2 = 0, Sy = 0, Sy2 = 0;
    for (int64_t t = 0; t < n; ++t) {
        double yt = y[t];
        int nan = isnan(yt);
        Sw = om * Sw + (nan ? 0.0 : 1.0);
        Sw2 = om2 * Sw2 + (nan ? 0.0 : 1.0);
        if (!nan) { Sy = om * Sy + yt; Sy2 = om * Sy2 + yt * yt; }
        else { Sy = om * Sy; Sy2 = om * Sy2; }
        if (Sw > 0.0) {
            double mean = Sy / Sw;
            double den = Sw - (Sw2 / Sw);
            if (den > 0.0) {
                double var = (Sy2 / Sw - mean * mean) * Sw / den;
                if (!(var > 0.0)) var = isnan(var) ? var : 0.0;
                out[t] = sqrt(var);
            } else out[t] = NAN;
        } else out[t] = NAN;
    }
    return ORC_OK;
}

int orc_realized_vol(const double *r, int64_t n, int64_t window, int is_sample, double *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = NAN;
    if (window < 1) return ORC_E_ARG;
    double *sq = (double *)malloc(sizeof(double) * (size_t)window);
    if (!sq) return ORC_E_NOMEM;
    for (int64_t i = window - 1; i < n; ++i) {
        const double *w = r + (i - window + 1);
        int64_t valid = 0;
        for (int64_t k = 0; k < window; ++k) {
            int nan = isnan(w[k]);
            valid += !nan;
            sq[k] = nan ? 0.0 : w[k] * w[k];       /* np.nansum(r_window ** 2) */
        }
        if (valid > 1) {
            double div = is_sample ? (double)(valid - 1) : (double)valid;
            out[i] = sqrt(orc_pairwise_f64(sq, window) / div);
        }
    }
    free(sq);
    return ORC_OK;
}

/* comp_price_tick_size  finmlkit/bar/utils.py:49-81 (host-side helper of
 * build_footprints; restated so the Python host logic can be checked). */
static int64_t orc_gcd(int64_t a, int64_t b) { while (b) { int64_t t = a % b; a = b; b = t; } return a; }

int orc_comp_price_tick_size(const double *prices, int64_t n, double *out)
{
    if (n <= 0) return ORC_E_ARG;
    int64_t m = n < 10000 ? n : 10000;
    double *s = (double *)malloc(sizeof(double) * (size_t)m);
    if (!s) return ORC_E_NOMEM;
    for (int64_t i = 0; i < m; ++i) s[i] = nearbyint(prices[i] * 1e12) / 1e12;   /* np.round(.,12) */
    qsort(s, (size_t)m, sizeof(double), orc_cmp_f64);
    int64_t u = 0;
    for (int64_t i = 0; i < m; ++i) if (i == 0 || s[i] != s[u - 1]) s[u++] = s[i];
    if (u <= 1) { free(s); *out = 0.0; return ORC_OK; }
    double mind = INFINITY;
    for (int64_t i = 1; i < u; ++i) { double d = s[i] - s[i - 1]; if (d > 0 && d < mind) mind = d; }
    double scale = pow(10.0, -floor(log10(mind)));
    int64_t tick = 0, prev = (int64_t)nearbyint(s[0] * scale);
    for (int64_t i = 1; i < u; ++i) {
        int64_t cur = (int64_t)nearbyint(s[i] * scale);
        int64_t d = cur - prev; prev = cur;
        if (d > 0) {
            tick = tick == 0 ? d : orc_gcd(tick, d);
            if (tick == 1) break;
        }
    }
    free(s);
    *out = (double)tick / scale;
    return ORC_OK;
}

/* ------------------------------------------------------------------ */
/* "Next" rank 4: the loops of TradesData(preprocess=True)              */
/* merge_split_trades       finmlkit/bar/utils.py:263-329              */
/* comp_trade_side_vector   finmlkit/bar/utils.py:26-46 (+ :10-23)     */
/* ------------------------------------------------------------------ */
int64_t orc_merge_split_trades(const int64_t *ts, const double *prices, const float *amounts,
                               const uint8_t *is_buyer_maker, int64_t n,
                               int64_t *o_ts, double *o_px, float *o_am, int8_t *o_side)
{
    if (n <= 0) return ORC_E_ARG;
    int64_t m = 0;
    o_ts[0] = ts[0]; o_px[0] = prices[0]; o_am[0] = amounts[0];
    int head_maker = is_buyer_maker ? (is_buyer_maker[0] != 0) : 0;
    if (is_buyer_maker) o_side[0] = head_maker ? -1 : 1;
    for (int64_t i = 1; i < n; ++i) {
        int same = ts[i] == o_ts[m] && fabs(prices[i] - o_px[m]) < 1e-8;          /* utils.py:300-301 */
        if (is_buyer_maker) same = same && ((is_buyer_maker[i] != 0) == head_maker);   /* utils.py:303-304 */
        if (same) {
            o_am[m] = o_am[m] + amounts[i];                                        /* float32 += float32 */
        } else {
            ++m;
            o_ts[m] = ts[i]; o_px[m] = prices[i]; o_am[m] = amounts[i];
            if (is_buyer_maker) { head_maker = is_buyer_maker[i] != 0; o_side[m] = head_maker ? -1 : 1; }
        }
    }
    return m + 1;
}

int orc_comp_trade_side_vector(const double *prices, int64_t n, int8_t *out)
{
    if (n <= 0) return ORC_E_ARG;
    int prev = 0;
    out[0] = 0;
    for (int64_t i = 1; i < n; ++i) {
        double dp = prices[i] - prices[i - 1];
        if (fabs(dp) > 1e-12) prev = dp > 0 ? 1 : (dp < 0 ? -1 : 0);             /* np.sign(dp) */
        out[i] = (int8_t)prev;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------ */
/* "Next" rank 2: rolling volume profile                               */
/* finmlkit/feature/core/volume.py: aggregate_footprint :133-203,      */
/* bucket_price_levels :206-275, comp_poc_hva_lva :278-369,            */
/* calc_volume_percentage_above_poc :372-400, volume_profile_rolling   */
/* :403-456.  Footprints in CSR form (off[B+1] + flat level arrays).   */
/* Typed (Numba) semantics: float32 arrays, float64 scalars.           */
/* ORC_E_LEVEL: a window spans ONE price level with bucketing on -- the */
/* reference (pure-Python mode) raises a broadcast ValueError there.    */
/* ------------------------------------------------------------------ */
static int64_t orc_lower_i64(const int64_t *a, int64_t n, int64_t key)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) { int64_t mid = lo + (hi - lo) / 2; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}
static int64_t orc_upper_i64(const int64_t *a, int64_t n, int64_t key)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) { int64_t mid = lo + (hi - lo) / 2; if (a[mid] <= key) lo = mid + 1; else hi = mid; }
    return lo;
}

/* calc_volume_percentage_above_poc, finmlkit/feature/core/volume.py:367-391, stand-alone (typed semantics: float32
 * np.sum, float64 accumulator and quotient) */
double orc_calc_volume_percentage_above_poc(const int32_t *pl, const float *vol, int64_t n, int32_t poc_price);

/* comp_poc_hva_lva + calc_volume_percentage_above_poc on one profile */
static void orc_poc_hva_lva(const int32_t *pl, const float *vol, int64_t n, double va_pct,
                            int32_t *poc, int32_t *hva, int32_t *lva, float *pct)
{
    float total = orc_pairwise_f32(vol, n);                       /* np.sum(float32 array) */
    int64_t pi = 0;
    for (int64_t k = 1; k < n; ++k) if (vol[k] > vol[pi] || (vol[k] != vol[k] && vol[pi] == vol[pi])) pi = k; /* np.argmax (first NaN = max) */
    int32_t poc_price = pl[pi];
    double va_thrs = (double)total * (va_pct / 100.0);
    double cum = vol[pi];
    int32_t hv = poc_price, lv = poc_price;
    int64_t up = pi + 1, down = pi - 1;
    double cu = 0.0, cd = 0.0;
    if (up < n) { cu = vol[up]; if (up + 1 < n) cu += vol[up + 1]; }
    if (down >= 0) { cd = vol[down]; if (down - 1 >= 0) cd += vol[down - 1]; }
    while (cum < va_thrs) {
        if (cu > cd) {
            cum += cu;
            hv = pl[up + 1 < n - 1 ? up + 1 : n - 1];
            up += 2;
            cu = -1.0;
            if (up < n) { cu = vol[up]; if (up + 1 < n) cu += vol[up + 1]; }
        } else if (cu < cd) {
            cum += cd;
            lv = pl[down - 1 > 0 ? down - 1 : 0];
            down -= 2;
            cd = -1.0;
            if (down >= 0) { cd = vol[down]; if (down - 1 >= 0) cd += vol[down - 1]; }
        } else if (cu == cd && cd != -1.0) {
            cum += cu + cd;
            hv = pl[up + 1 < n - 1 ? up + 1 : n - 1];
            lv = pl[down - 1 > 0 ? down - 1 : 0];
            up += 2; down -= 2;
            cu = -1.0;
            if (up < n) { cu = vol[up]; if (up + 1 < n) cu += vol[up + 1]; }
            cd = -1.0;
            if (down >= 0) { cd = vol[down]; if (down - 1 >= 0) cd += vol[down - 1]; }
        } else break;                                              /* "BUG! Stuck in loop" branch */
    }
    *poc = poc_price; *hva = hv; *lva = lv;
    float p = 0.0f;
    if (!(total <= 0.0f)) {                                         /* volume.py:378: `<= 0` lets a NaN total through */
        double above = 0.0;
        for (int64_t k = 0; k < n; ++k) if (pl[k] > poc_price) above += vol[k];
        if (!(above <= 0.0)) p = (float)(above / (double)total);
    }
    *pct = p;
}

int orc_volume_profile_rolling(const int64_t *ts, const double *highs, const double *lows,
                               const int64_t *off, const int32_t *levels, const float *buy, const float *sell,
                               int64_t nb, int64_t window_ns, int64_t n_bins, double tick, double va_pct,
                               int32_t *poc, int32_t *hva, int32_t *lva, float *pct)
{
    if (nb <= 0) return ORC_E_ARG;
    for (int64_t i = 0; i < nb; ++i) { poc[i] = hva[i] = lva[i] = 0; pct[i] = 0.0f; }
    int64_t first = orc_lower_i64(ts, nb, ts[0] + window_ns);
    for (int64_t i = first; i < nb; ++i) {
        int64_t end_ts = ts[i], start_ts = end_ts - window_ns;
        int64_t s = orc_lower_i64(ts, nb, start_ts), e = orc_upper_i64(ts, nb, end_ts);
        if (s == e) s = s - 1 > 0 ? s - 1 : 0;
        double mn = lows[s], mx = highs[s];
        for (int64_t t = s + 1; t < e; ++t) { if (lows[t] < mn) mn = lows[t]; if (highs[t] > mx) mx = highs[t]; }
        int64_t minl = (int64_t)nearbyint(mn / tick), maxl = (int64_t)nearbyint(mx / tick);
        int64_t L = maxl - minl + 1;
        if (L < 1) return ORC_E_ARG;
        float *ab = (float *)calloc((size_t)L, sizeof(float)), *as = (float *)calloc((size_t)L, sizeof(float));
        float *tot = (float *)malloc(sizeof(float) * (size_t)(L + 1));
        int32_t *pl = (int32_t *)malloc(sizeof(int32_t) * (size_t)(L + 1));
        if (!ab || !as || !tot || !pl) { free(ab); free(as); free(tot); free(pl); return ORC_E_NOMEM; }
        for (int64_t t = s; t < e; ++t)
            for (int64_t r = off[t]; r < off[t + 1]; ++r) {
                int64_t idx = (int64_t)levels[r] - minl;            /* searchsorted on a dense arange */
                if (idx < 0 || idx >= L) { free(ab); free(as); free(tot); free(pl); return ORC_E_LEVEL; }
                ab[idx] += buy[r];
                as[idx] += sell[r];
            }
        int64_t np_ = L;
        for (int64_t k = 0; k < L; ++k) { tot[k] = ab[k] + as[k]; pl[k] = (int32_t)(minl + k); }
        if (n_bins >= 0) {                                         /* bucket_price_levels */
            if (n_bins == 0) { free(ab); free(as); free(tot); free(pl); return ORC_E_ZERODIV; }
            int64_t range = maxl - minl;
            int64_t bw = range / n_bins;                           /* non-negative: floor == trunc */
            if (bw < 1) bw = 1;
            if (bw % 2 == 0) bw += 1;
            int64_t n_edges = (maxl + bw - minl + bw - 1) / bw;    /* len(arange(min, max + bw, bw)) */
            int64_t nbk = n_edges - 1;
            if (n_edges < 2) { free(ab); free(as); free(tot); free(pl); return ORC_E_LEVEL; }
            /* bin of level v: (#edges <= v) - 1 = min((v - min) / bw, nbk) */
            float *bv = (float *)calloc((size_t)(nbk + 1), sizeof(float));
            int32_t *bp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nbk + 1));
            if (!bv || !bp) { free(bv); free(bp); free(ab); free(as); free(tot); free(pl); return ORC_E_NOMEM; }
            for (int64_t k = 0; k < nbk; ++k) {
                int64_t e0 = minl + k * bw, e1 = e0 + bw;
                int64_t ssum = e0 + e1 - 1;
                bp[k] = (int32_t)(ssum >= 0 ? ssum / 2 : -((-ssum + 1) / 2));      /* floor division */
            }
            bp[nbk] = (int32_t)maxl;
            for (int64_t k = 0; k < L; ++k) {
                int64_t b = k / bw;
                if (b > nbk) b = nbk;
                bv[b] += tot[k];
            }
            /* volume.py:244-252: the extra "leftover" bin exists iff the LAST level falls past the last full bin */
            np_ = nbk + ((range / bw >= nbk) ? 1 : 0);
            for (int64_t k = 0; k < np_; ++k) { tot[k] = bv[k]; pl[k] = bp[k]; }
            free(bv); free(bp);
        }
        orc_poc_hva_lva(pl, tot, np_, va_pct, &poc[i], &hva[i], &lva[i], &pct[i]);
        free(ab); free(as); free(tot); free(pl);
    }
    return ORC_OK;
}


double orc_calc_volume_percentage_above_poc(const int32_t *pl, const float *vol, int64_t n, int32_t poc_price)
{
    float total = orc_pairwise_f32(vol, n);                       /* volume.py:377 np.sum(volumes) */
    if (total <= 0.f) return 0.0;                                  /* :378-379 (a NaN total is not caught) */
    double above = 0.0;
    for (int64_t k = 0; k < n; ++k) if (pl[k] > poc_price) above += vol[k];   /* :381-384 */
    if (above <= 0.0) return 0.0;                                  /* :387-388 */
    return above / (double)total;                                 /* :390 */
}


/* ------------------------------------------------------------------ */
/* TimeBarReader._resample, finmlkit/bar/io.py:890-950                  */
/* Rows [seg[g], seg[g+1]) form group g (the caller groups by            */
/* index.floor(timeframe), :913).  pandas' groupby sum is a Kahan-       */
/* compensated sequential sum in the column dtype (pandas/_libs/         */
/* groupby.pyx group_sum; verified in oracle/gen_resample.py).           */
/* ------------------------------------------------------------------ */
#define ORC_KAHAN(T, sum, comp, val)                     \
    do {                                                 \
        T v__ = (val);                                   \
        if (v__ == v__) {                                \
            T y__ = v__ - (comp);                        \
            T t__ = (sum) + y__;                         \
            (comp) = (t__ - (sum)) - y__;                \
            if ((comp) != (comp)) (comp) = 0;            \
            (sum) = t__;                                 \
        }                                                \
    } while (0)

typedef struct { double size; double w; int64_t row; } orc_rs_pair;
static int orc_rs_cmp(const void *a, const void *b)
{
    const orc_rs_pair *x = (const orc_rs_pair *)a, *y = (const orc_rs_pair *)b;
    int xn = x->size != x->size, yn = y->size != y->size;
    if (xn || yn) return xn - yn;                        /* NaN last (np.argsort) */
    if (x->size < y->size) return -1;
    if (x->size > y->size) return 1;
    return (x->row > y->row) - (x->row < y->row);        /* any order among ties gives the same VALUE (io.py:937-946) */
}

int orc_resample_bars(const int64_t *seg, int64_t n_groups, const double *open, const double *high, const double *low,
                      const double *close, const void *volume, int volume_is_f64, const int64_t *trades, const void *vwap,
                      int vwap_is_f64, const double *median, double *o_open, double *o_high, double *o_low, double *o_close,
                      void *o_volume, int64_t *o_trades, float *o_vwap, float *o_median, uint8_t *o_valid)
{
    for (int64_t g = 0; g < n_groups; ++g) {
        int64_t s = seg[g], e = seg[g + 1];
        double f_open = NAN, l_close = NAN, hi = NAN, lo = NAN;
        int have = 0;
        int64_t tr = 0;
        float vs32 = 0, vc32 = 0, ps32 = 0, pc32 = 0;
        double vs64 = 0, vc64 = 0, ps64 = 0, pc64 = 0;
        for (int64_t i = s; i < e; ++i) {
            if (!have && open[i] == open[i]) { f_open = open[i]; have = 1; }          /* "first" skips NaN (:918) */
            if (close[i] == close[i]) l_close = close[i];                             /* "last" (:921) */
            if (high[i] == high[i] && !(hi >= high[i])) hi = high[i];                 /* "max" / "min" skip NaN */
            if (low[i] == low[i] && !(lo <= low[i])) lo = low[i];
            tr += trades[i];                                                          /* :923 */
            if (volume_is_f64) {
                double v = ((const double *)volume)[i];
                double w = vwap_is_f64 ? ((const double *)vwap)[i] : (double)((const float *)vwap)[i];
                ORC_KAHAN(double, vs64, vc64, v);
                ORC_KAHAN(double, ps64, pc64, w * v);                                 /* :929 */
            } else if (vwap_is_f64) {
                float v = ((const float *)volume)[i];
                ORC_KAHAN(float, vs32, vc32, v);
                ORC_KAHAN(double, ps64, pc64, ((const double *)vwap)[i] * (double)v); /* float64 * float32 -> float64 */
            } else {
                float v = ((const float *)volume)[i];
                float p = ((const float *)vwap)[i] * v;                               /* float32 * float32 -> float32 */
                ORC_KAHAN(float, vs32, vc32, v);
                ORC_KAHAN(float, ps32, pc32, p);
            }
        }
        o_open[g] = f_open; o_high[g] = hi; o_low[g] = lo; o_close[g] = l_close; o_trades[g] = tr;
        if (volume_is_f64) { ((double *)o_volume)[g] = vs64; o_vwap[g] = (float)(ps64 / vs64); }
        else if (vwap_is_f64) { ((float *)o_volume)[g] = vs32; o_vwap[g] = (float)(ps64 / (double)vs32); }
        else { ((float *)o_volume)[g] = vs32; o_vwap[g] = ps32 / vs32; }
        o_valid[g] = (uint8_t)have;
        /* w_median (:933-946): argsort by size, cumsum of the float64 weights, first index with cum >= total * 0.5 */
        float med = NAN;
        if (e > s) {
            orc_rs_pair *pr = (orc_rs_pair *)malloc((size_t)(e - s) * sizeof(orc_rs_pair));
            if (!pr) return ORC_E_ARG;
            for (int64_t i = s; i < e; ++i) { pr[i - s].size = median[i]; pr[i - s].w = (double)trades[i]; pr[i - s].row = i; }
            qsort(pr, (size_t)(e - s), sizeof(orc_rs_pair), orc_rs_cmp);
            double tot = 0.0;
            for (int64_t i = 0; i < e - s; ++i) tot += pr[i].w;
            double cutoff = tot * 0.5, cum = 0.0;
            int64_t idx = e - s;                          /* searchsorted(..., 'left') past the end would raise; cannot: */
            for (int64_t i = 0; i < e - s; ++i) {         /* cum[-1] >= cutoff whenever the weights are >= 0            */
                cum += pr[i].w;
                if (cum >= cutoff) { idx = i; break; }
            }
            if (idx < e - s) med = (float)pr[idx].size;
            free(pr);
        }
        o_median[g] = med;
    }
    return ORC_OK;
}
