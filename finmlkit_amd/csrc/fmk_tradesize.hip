// fmk_tradesize.hip -- comp_bar_trade_size_features (finmlkit/bar/base.py:549-612) on gfx950.
// First "next" row of SURVEY.md 8(f): same bar segments, reuses the exact order-statistic search of
// fmk_median.h for np.percentile(amounts_bar, 95).
//
// One wave per bar, three passes over the bar's amounts (the 2nd/3rd hit L2; bars of <= 2048 ticks keep
// their keys in registers for the percentile):
//   pass 1  sum, block volume (amounts > theta*theta_mult)            -> mean_size_rel, pct_block
//   select  ranks floor(0.95(n-1)) and +1, NumPy 'linear' interpolation -> size_95_rel
//   pass 3  sum (a/total)^2                                            -> size_gini
// Arithmetic is float64 throughout and the results are rounded once to float32.  For float64 amounts
// this is the reference's arithmetic up to summation order; for float32 amounts the reference (both its
// NumPy and its Numba mode) accumulates the sums in float32, which this kernel deliberately does not
// reproduce -- the float64 sums are the more accurate value and differ by ~1e-7 relative (documented
// tolerance in tests/test_gpu_next.py).
#include <math.h>

#include <utility>

#include "fmk_median.h"
#include "fmk_pairwise.h"
#include "fmk_dpp.h"
#include "fmk_scan.h"

// (float)log1p(v) as a call: its constants do not occupy registers across a kernel's bar loop (defined with the one-read kernels)
static __device__ __attribute__((noinline)) float tsm_log1p(double v);

template <bool AF64, int NREG>
__device__ __forceinline__ double ts_percentile95(const void *amount, int64_t start, int64_t cnt, int lane,
                                                  typename MedKey<AF64>::K *buf, double thr = 0.0, double *block_out = nullptr)
{
    typedef MedKey<AF64> MK;
    MedBar<AF64, NREG, false> bar;
    bar.amount = amount; bar.start = start; bar.cnt = cnt; bar.lane = lane;
    bar.load_all();
    if constexpr (NREG > 0 && !AF64) {
        // float32 bars held in registers: the block volume (amounts above the threshold; a float64 sum of float32 values is
        // exact in any order) comes from the same registers -- no separate pass over the bar (2.8 of 9.4 ms per 1e9 ticks)
        if (block_out) {
            double bl = 0.0;
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                const double a = bar.key[r] != MK::MAXK ? MK::value(bar.key[r]) : 0.0;
                bl += a > thr ? a : 0.0;
            }
            *block_out = fmk_wave_sum(bl);
        }
    }
    // NumPy 'linear' method: virtual index (n-1)*q IN THE ARRAY'S DTYPE, neighbours floor and floor+1 (clipped), _lerp
    const double vidx = AF64 ? (double)(cnt - 1) * 0.95 : (double)((float)(cnt - 1) * (95.0f / 100.0f));
    const double fl = floor(vidx);
    const int64_t k1 = (int64_t)fl < cnt - 1 ? (int64_t)fl : cnt - 1;
    const int64_t k2 = k1 + 1 < cnt ? k1 + 1 : cnt - 1;
    typename MK::K v1, v2;
    if (!med_rank_pair<AF64, NREG, false>(bar, buf, k1, k2, v1, v2)) return NAN;
    const double a = MK::value(v1), b = MK::value(v2);
    if constexpr (!AF64) {
        // float32 amounts: NumPy 2.2 does EVERY step of np.percentile in the array's dtype -- q = 95 / float32(100), the
        // virtual index float32(n - 1) * q, the weight, the difference and both interpolation forms (oracle: orc_percentile_f32,
        // validated bit for bit against np.percentile).  The reference's slices are float32 after TradesData's merge.
        const float q32 = 95.0f / 100.0f;
        const float vi = (float)(cnt - 1) * q32;
        if (vi >= (float)(cnt - 1)) return b;            // k2 is clipped to the last element
        const float t32 = vi - floorf(vi);
        const float a32 = (float)a, b32 = (float)b, d32 = b32 - a32;
        float r32 = a32 + d32 * t32;
        if (t32 >= 0.5f) r32 = b32 - d32 * (1.0f - t32);
        return (double)r32;
    }
    const double t = vidx - fl;
    const double d = b - a;
    double r = a + d * t;
    if (t >= 0.5) r = b - d * (1.0 - t);
    if (d == 0.0) r = a;
    return r;
}

// np.percentile(., 95) of float32 bars beyond the register classes (more than 2048 ticks), by a workgroup per bar (med_block_select:
// radix select of the two ranks) in a pass of its own: the value is exactly a float32 (NumPy interpolates in the array's dtype),
// so it travels in o_p95[b] and k_bar_trade_size turns it into size_95_rel.  One wave bisecting the value range with a re-read of
// the bar per step took 35 / 57 / 383 ms per 1e9 ticks at 10-minute / hourly / daily bars.  Bars with irregular close indices
// (below -1 or beyond the array: Python slice semantics) stay with the old path.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_ts_p95_long(const float *__restrict__ amount, const int64_t *__restrict__ ci,
                                                         const int64_t *__restrict__ list, int64_t n, float *__restrict__ o_p95,
                                                         int64_t skip_lo = INT64_MAX, int64_t skip_hi = 0 /* bars of skip_lo < ticks <= skip_hi: k_bar_trade_size_wide selects itself */)
{
    typedef MedKey<false> MK;
    const int64_t n_list = list[0];
    for (int64_t q = blockIdx.x; q < n_list; q += gridDim.x) {
        const int64_t b = list[1 + q], s = ci[b], e = ci[b + 1];
        if (!(s >= -1 && e <= n - 1) || (e - s > skip_lo && e - s <= skip_hi)) continue;
        const int64_t cnt = e - s, start = s + 1;
        const float q32 = 95.0f / 100.0f;
        const float vi = (float)(cnt - 1) * q32;             // the virtual index in the array's dtype (ts_percentile95)
        const double fl = floor((double)vi);
        const int64_t k1 = (int64_t)fl < cnt - 1 ? (int64_t)fl : cnt - 1;
        const int64_t k2 = k1 + 1 < cnt ? k1 + 1 : cnt - 1;
        MK::K v1, v2;
        bool any_nan;
        med_block_select<false, THREADS>(amount, start, cnt, k1, k2, v1, v2, any_nan);
        if (threadIdx.x == 0) {
            const float a32 = (float)MK::value(v1), b32 = (float)MK::value(v2);
            float r32;
            if (any_nan) r32 = NAN;
            else if (vi >= (float)(cnt - 1)) r32 = b32;
            else {
                const float t32 = vi - floorf(vi), d32 = b32 - a32;
                r32 = a32 + d32 * t32;
                if (t32 >= 0.5f) r32 = b32 - d32 * (1.0f - t32);
            }
            o_p95[b] = r32;
        }
    }
}

// SMALL: the instantiation for bars of at most 20 x 64 ticks only (float32): without the 32-key class the kernel needs fewer registers
// and a fourth wave fits the SIMD; the full instantiation then serves the longer bars (`cnt_lo`: it skips the others).
template <bool AF64, bool SMALL = false>
__global__ __launch_bounds__(256, SMALL ? 4 : 1) void k_bar_trade_size(const void *__restrict__ amount,
                                                        const double *__restrict__ theta,
                                                        const int64_t *__restrict__ ci, int64_t nb, int64_t n,
                                                        double theta_mult, float *__restrict__ o_mean, float *__restrict__ o_p95,
                                                        float *__restrict__ o_pct, float *__restrict__ o_gini, int p95_done,
                                                        const unsigned long long *__restrict__ only = nullptr,
                                                        int64_t skip_above = INT64_MAX /* longer regular bars: the workgroup kernels' */,
                                                        int64_t cnt_lo = INT64_MIN /* bars of at most this many ticks: another launch's */,
                                                        int64_t skip_hi = FMK_PW_BIG_MAX_N /* ... up to this many ticks */)
{
    static_assert(!(SMALL && AF64), "the small-bar instantiation serves float32 amounts");
    typedef typename MedKey<AF64>::K K;
    __shared__ K sbuf[4][64];
    __shared__ __attribute__((aligned(8))) int s_stk[4][FMK_PW_PAR_STK];      // scratch of the pairwise sums
    const int lane = fmk_lane();
    const int wib = fmk_uniform((int)(threadIdx.x >> 6));
    const int wpb = blockDim.x >> 6;
    const int64_t wave0 = (int64_t)blockIdx.x * wpb + wib;
    const int64_t nwaves = (int64_t)gridDim.x * wpb;
    K *buf = sbuf[wib];
    // `only` (list mode: [0] = count, [32...] = bar numbers): the bars the lane-per-bar / row-per-bar schedules left to this one
    const int64_t todo = only ? (int64_t)only[0] : nb;
    for (int64_t it = wave0; it < todo; it += nwaves) {
        const int64_t b = only ? fmk_uniform((int64_t)only[32 + it]) : it;
        const int64_t s = fmk_uniform(ci[b]);
        const int64_t e_raw = fmk_uniform(ci[b + 1]);
        // The reference takes the bar as a SLICE, amounts[start:end + 1] (base.py:590): an end index past the array is
        // clamped, not an error (its own test_block_volume passes end == len(amounts)).  The empty-bar guard
        // (start > end, base.py:584) looks at the raw indices; a slice left empty by the clamp gives mean([]) = NaN and
        // a zero total -> the same all-NaN row.
        // Python slice bounds: a negative start / stop wraps by n once and is then clamped to [0, n] -- a close index below -1
        // must not become a read in front of the column (it selects the reference's wrapped, usually empty, slice)
        if (!AF64 && e_raw - s > skip_above && e_raw - s <= skip_hi && s >= -1 && e_raw <= n - 1) continue;
        if (SMALL ? e_raw - s > 64 * 20 : e_raw - s <= cnt_lo) continue;
        int64_t start = s + 1, stop = e_raw + 1;
        start = start < 0 ? (start + n > 0 ? start + n : 0) : (start < n ? start : n);
        stop = stop < 0 ? (stop + n > 0 ? stop + n : 0) : (stop < n ? stop : n);
        const int64_t cnt = (e_raw - s > 0 && stop > start) ? stop - start : 0;
        float mean_rel = NAN, p95_rel = NAN, pct = NAN, gini = NAN;      // base.py:576-579
        const double th = theta[b];
        if (cnt > 0 && th != 0.0) {                                      // base.py:586-587
            const double thr = th * theta_mult;
            double sum = 0.0, block = 0.0;
            if constexpr (AF64) {
                // block_volume += amount runs in tick order in the reference (base.py:600-603) and float64 addition is not
                // associative: the (few) amounts above the threshold are added one by one, in order, by the whole wave
                for (int64_t j0 = 0; j0 < cnt; j0 += 64) {
                    const int64_t j = j0 + lane;
                    const double a = j < cnt ? fmk_amt<AF64>(amount, start + j) : 0.0;
                    sum += a;
                    uint64_t m = __ballot(j < cnt && a > thr);
                    while (m) {
                        const int l = __ffsll((unsigned long long)m) - 1;
                        m &= m - 1;
                        const int64_t bits = __double_as_longlong(a);
                        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, l);
                        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)((uint64_t)bits >> 32), l);
                        block += __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
                    }
                }
                sum = fmk_wave_sum(sum);
            } else if (!(cnt <= 64 * 32 && cnt <= FMK_PW_MAX_N)) {      // (else: `sum` is the pairwise total, `block` comes with the percentile)
                for (int64_t j = lane; j < cnt; j += 64) {
                    const double a = fmk_amt<AF64>(amount, start + j);
                    sum += a;
                    if (a > thr) block += a;                // float32 amounts: the float64 sum is exact in any order
                }
                sum = fmk_wave_sum(sum);
                block = fmk_wave_sum(block);
            }
            double mean = sum / (double)cnt;
            // float32 amounts (what TradesData's merge produces): the reference's np.mean / .sum() of the float32 slice are
            // NumPy PAIRWISE float32 sums and the mean divides in float32 (base.py:591-596 in NumPy semantics; oracle:
            // orc_pairwise_f32, 0 ulp against the reference's kit frames).  `sum` then is that rounded total.
            // float64 amounts: the same trees in float64 (np.mean / .sum() of a float64 slice are pairwise too).
            float tf = 0.f;
            const bool np_rule = cnt <= FMK_PW_BIG_MAX_N;   // longer bars than the explicit stack holds: tree-ordered sums
            const bool f32_rule = !AF64 && np_rule;
            if (np_rule) {
                if constexpr (!AF64) {
                    const float *af = (const float *)amount + start;
                    tf = fmk_np_sum([af](int i) { return af[i]; }, (int)cnt, lane, s_stk[wib]);
                    mean = (double)(tf / (float)cnt);
                    sum = (double)tf;
                } else {
                    const double *ad = (const double *)amount + start;
                    sum = fmk_np_sum([ad](int i) { return ad[i]; }, (int)cnt, lane, s_stk[wib]);
                    mean = sum / (double)cnt;
                }
            }
            mean_rel = (float)log1p(mean / thr);
            const int nreg = (int)((cnt + 63) >> 6);
            double p95;
            double *bo = (!AF64 && np_rule) ? &block : nullptr;
            if (!SMALL && cnt > 64 * 32) {
                if (!AF64 && p95_done && s >= -1 && e_raw <= n - 1) p95 = (double)o_p95[b];      // k_ts_p95_long
                else p95 = ts_percentile95<AF64, 0>(amount, start, cnt, lane, buf);
            }
            else if (nreg <= 4) p95 = ts_percentile95<AF64, 4>(amount, start, cnt, lane, buf, thr, bo);
            else if (nreg <= 12) p95 = ts_percentile95<AF64, 12>(amount, start, cnt, lane, buf, thr, bo);
            else if (nreg <= 20) p95 = ts_percentile95<AF64, 20>(amount, start, cnt, lane, buf, thr, bo);
            else if constexpr (SMALL) p95 = NAN;                         // (not reached: longer bars are skipped above)
            else if constexpr (!AF64) p95 = ts_percentile95<AF64, 32>(amount, start, cnt, lane, buf, thr, bo);
            else if (nreg <= 24) p95 = ts_percentile95<AF64, 24>(amount, start, cnt, lane, buf);
            else p95 = ts_percentile95<AF64, 0>(amount, start, cnt, lane, buf);
            p95_rel = (float)log1p(p95 / thr);
            if (sum != 0.0) {                                            // base.py:597-598
                pct = (float)(block / sum);
                if (cnt == 1) gini = 0.f;
                else if (f32_rule) {
                    // 1 - sum((a / total)^2) with float32 quotients, squares and pairwise sum (base.py:609)
                    const float *af = (const float *)amount + start;
                    const float t32 = tf;
                    gini = 1.0f - fmk_np_sum([af, t32](int i) { const float q = af[i] / t32; return q * q; }, (int)cnt, lane,
                                               s_stk[wib]);
                } else if (np_rule) {
                    const double *ad = (const double *)amount + start;
                    const double td = sum;
                    gini = (float)(1.0 - fmk_np_sum([ad, td](int i) { const double q = ad[i] / td; return q * q; }, (int)cnt, lane,
                                                       s_stk[wib]));
                } else {
                    double sq = 0.0;
                    for (int64_t j = lane; j < cnt; j += 64) {
                        const double q = fmk_amt<AF64>(amount, start + j) / sum;
                        sq += q * q;
                    }
                    gini = (float)(1.0 - fmk_wave_sum(sq));
                }
            }
        }
        if (lane == 0) { o_mean[b] = mean_rel; o_p95[b] = p95_rel; o_pct[b] = pct; o_gini[b] = gini; }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Bars of 1 .. 1 920 ticks (1-minute bars of a liquid tape), float32 amounts, regular close indices: one wave per bar that reads
// the bar ONCE (round 3).  k_bar_trade_size walks the bar three times (np.sum's tree, the order-statistic search, the tree of the
// squared shares), builds the tree's shape twice through LDS and divides by the total with the IEEE macro: ~4 000 wave instructions
// per 1 200-tick bar, 5.1 ms per 1e9 ticks.  Here
//   * np.sum's tree over n <= 1 928 elements has at most four levels of splits, so a leaf is the end of a 4-bit PATH from the root
//     (tsm_leaf: the split rule n2 = n / 2 rounded down to a multiple of 8 applied along the path, no LDS); groups of eight lanes
//     take a leaf each -- lane i of a group is NumPy's accumulator r_i -- in two rounds of eight leaves, and the bar's sizes STAY
//     in those registers (x[round][k]: element 8k + i of the leaf; t[round]: the leaf's tail element i, one per lane);
//   * both trees (the float32 total, then sum((a / total)^2)) fold inside the row by DPP shifts -- ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)),
//     the tail added element by element from the neighbours' registers -- and the sixteen leaf values are combined up the four levels
//     by sixteen lanes of one row, left + right where the node had split: every addition has the recursion's operands and order;
//   * a / total is formed as float(double(a) * (1 / double(total))): the product is within 2^-52 of the quotient, a quotient of two
//     float32 is at least 2^-49 (relative) away from every rounding boundary of float32, so the result is the correctly rounded
//     float32 quotient whenever it is a normal number; a bar with a subnormal quotient (v_cmp_class) repeats with the division;
//   * np.percentile's two ranks are searched on the same registers in the FLOAT domain (count of x <= pivot by ballots, the pivot
//     bisecting the order-preserving integer image of the value range); once at most 64 candidates are left they are compacted into
//     one register through LDS and the bisection goes on there until it separates the two ranks (no cross-lane sort);
//   * a NaN size makes the total NaN (the slots that hold no element are NaN too and are never added), so NaN bars are found there.
// ---------------------------------------------------------------------------------------------------------------------
#define TSM_MIN 0                      // (bars of up to 128 ticks are one leaf: eight lanes busy -- still one read instead of three passes)
#define TSM_MAX 1920                   // (np.sum's tree needs a fifth level from 1 929 elements)
#define FMK_DPP_ROW_SHL(n) (0x100 + (n))

template <int CTRL>
__device__ __forceinline__ float tsm_dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// order-preserving signed image of a float's bits (-0 < +0; every non-NaN value in [tsm_key(-inf), tsm_key(+inf)]) and back
// (an opaque copy: keeps the compiler from carrying the dozens of lane masks derived from a leaf's length across the whole bar)
__device__ __forceinline__ int tsm_opaque(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ float tsm_opaque(float v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ int tsm_key(float x) { const int b = __float_as_int(x); return b >= 0 ? b : b ^ 0x7FFFFFFF; }
__device__ __forceinline__ float tsm_val(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7FFFFFFF); }

// The node of np.sum's tree over n elements at the end of the D-bit path `path` (most significant bit first: 0 = left half):
// [off, off + len).  alive: the path ends in a leaf of its own (a leaf reached before the path is used up belongs to the path whose
// remaining bits are zero; len = 0 otherwise).  sp bit l: the node at level l of the path had split.
template <int D = 4>
__device__ __forceinline__ void tsm_leaf(int n, int path, int &off, int &len, unsigned &sp)
{
    off = 0; len = n; sp = 0;
    bool alive = true;
#pragma unroll
    for (int l = 0; l < D; ++l) {
        const bool bit = (path >> (D - 1 - l)) & 1;
        const bool split = len > 128;
        const int n2 = (len >> 1) & ~7;
        sp |= split ? 1u << l : 0u;
        alive = alive && (split || !bit);
        off += split && bit ? n2 : 0;
        len = split ? (bit ? len - n2 : n2) : len;
    }
    len = alive ? len : 0;
}

// One of the bar's two trees: f maps a size to the summand.  Round R holds the leaves of the paths 2 * group + R (one round: a tree
// without a split at level 3).  The slots without an element hold +0.0 and are added like elements: x + 0 is x unless x is -0, and a
// partial sum is -0 only if every element so far was -0 -- so only a total of zero can differ (in its sign), and the caller repeats
// such a bar with PRED (the slots skipped, as the recursion does).
template <int KMAX, int ROUNDS, bool PRED, class F>
__device__ __forceinline__ float tsm_tree(const float (&x)[ROUNDS][KMAX], const float (&t)[ROUNDS], const int (&len)[ROUNDS],
                                          unsigned sp_slot, F f, int lane, float *slots)
{
    const int i8 = lane & 7, grp = lane >> 3;
#pragma unroll
    for (int R = 0; R < ROUNDS; ++R) {
        const int lenR = PRED ? tsm_opaque(len[R]) : 0;
        const int nm = lenR & ~7;
        float r = f(x[R][0]);
#pragma unroll
        for (int k = 1; k < KMAX; ++k) {
            const float y = f(x[R][k]);
            if constexpr (PRED) r = 8 * k < nm ? r + y : r;
            else r += y;
        }
        const float a = r + tsm_dpp<FMK_DPP_ROW_SHL(1)>(r);            // lanes 0, 2, 4, 6 of the group: r0+r1, r2+r3, r4+r5, r6+r7
        const float u = a + tsm_dpp<FMK_DPP_ROW_SHL(2)>(a);            // lanes 0, 4
        float res = u + tsm_dpp<FMK_DPP_ROW_SHL(4)>(u);                // lane 0
        const float ty = f(t[R]);
        if constexpr (PRED) {
            res = nm + 0 < lenR ? res + ty : res;
            res = nm + 1 < lenR ? res + tsm_dpp<FMK_DPP_ROW_SHL(1)>(ty) : res;
            res = nm + 2 < lenR ? res + tsm_dpp<FMK_DPP_ROW_SHL(2)>(ty) : res;
            res = nm + 3 < lenR ? res + tsm_dpp<FMK_DPP_ROW_SHL(3)>(ty) : res;
            res = nm + 4 < lenR ? res + tsm_dpp<FMK_DPP_ROW_SHL(4)>(ty) : res;
            res = nm + 5 < lenR ? res + tsm_dpp<FMK_DPP_ROW_SHL(5)>(ty) : res;
            res = nm + 6 < lenR ? res + tsm_dpp<FMK_DPP_ROW_SHL(6)>(ty) : res;
        } else {
            res += ty;
            res += tsm_dpp<FMK_DPP_ROW_SHL(1)>(ty);
            res += tsm_dpp<FMK_DPP_ROW_SHL(2)>(ty);
            res += tsm_dpp<FMK_DPP_ROW_SHL(3)>(ty);
            res += tsm_dpp<FMK_DPP_ROW_SHL(4)>(ty);
            res += tsm_dpp<FMK_DPP_ROW_SHL(5)>(ty);
            res += tsm_dpp<FMK_DPP_ROW_SHL(6)>(ty);
        }
        if (i8 == 0) slots[(ROUNDS == 4 ? 4 : 2) * grp + R] = res;
    }
    __builtin_amdgcn_wave_barrier();
    float v = slots[lane & (ROUNDS == 4 ? 31 : 15)];
    __builtin_amdgcn_wave_barrier();
    if constexpr (ROUNDS == 4) {
        // five levels of splits (bars of up to 3 848 ticks): 32 slots in the lanes of two rows; the root's halves meet across them
        float w5;
        w5 = v + tsm_dpp<FMK_DPP_ROW_SHL(1)>(v); v = ((sp_slot >> 4) & 1) && (lane & 1) == 0 ? w5 : v;
        w5 = v + tsm_dpp<FMK_DPP_ROW_SHL(2)>(v); v = ((sp_slot >> 3) & 1) && (lane & 3) == 0 ? w5 : v;
        w5 = v + tsm_dpp<FMK_DPP_ROW_SHL(4)>(v); v = ((sp_slot >> 2) & 1) && (lane & 7) == 0 ? w5 : v;
        w5 = v + tsm_dpp<FMK_DPP_ROW_SHL(8)>(v); v = ((sp_slot >> 1) & 1) && (lane & 15) == 0 ? w5 : v;
        const float left = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
        const float right = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
        const unsigned sp0 = (unsigned)__builtin_amdgcn_readlane((int)sp_slot, 0);
        return (sp0 & 1) ? left + right : left;
    }
    // up the levels: slot g is the leftmost slot of its node at level l iff its low 4 - l bits are zero; the right child starts 2^(3-l) slots on
    float w;
    if constexpr (ROUNDS == 2) { w = v + tsm_dpp<FMK_DPP_ROW_SHL(1)>(v); v = ((sp_slot >> 3) & 1) && (lane & 1) == 0 ? w : v; }
    w = v + tsm_dpp<FMK_DPP_ROW_SHL(2)>(v); v = ((sp_slot >> 2) & 1) && (lane & 3) == 0 ? w : v;
    w = v + tsm_dpp<FMK_DPP_ROW_SHL(4)>(v); v = ((sp_slot >> 1) & 1) && (lane & 7) == 0 ? w : v;
    w = v + tsm_dpp<FMK_DPP_ROW_SHL(8)>(v); v = (sp_slot & 1) && (lane & 15) == 0 ? w : v;
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
}

// ---- W waves per bar (W = 2, 4, 8, 16: bars of up to 1 912 * W ticks): the top log2(W) levels of np.sum's tree cut the bar into W
// sub-trees of at most 1 928 elements, wave w takes the sub-tree at the end of the path w and does with it what the one-wave kernel
// does with a bar; what the bar needs as a whole goes through LDS, one workgroup barrier per exchange (two buffers in turn: a wave
// can be one exchange ahead of the slowest, never two):
//   the sub-trees' values, combined left + right up the top levels (both trees); the block volume; smallest and largest size;
//   every count of the percentile search (each wave counts its own registers; all waves follow the same search); the candidates,
//   compacted per wave and collected by wave 0, which finishes the search alone while the others go on with the shares.
struct TsmX {                          // the exchange buffers of a workgroup (W > 1)
    float f[2][16];
    int i[2][16];
    double d[2][16];
    float cand[16][64];
};
template <int W>
struct TsmWg {
    TsmX *x;
    int wib, lane, par;
    int nchunk, nwl;                   // nchunk > 1: a bar of several np.sum chunks, four waves per chunk; nwl: the last chunk's sub-trees
    // every wave contributes a wave-uniform value and receives all W of them
    template <class T, class G>
    __device__ __forceinline__ void all(T (*buf)[16], T v, G got)
    {
        if (lane == 0) buf[par][wib] = v;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < W; ++q) got(q, buf[par][q]);
        par ^= 1;
    }
    // left + right up the top levels of a chunk's tree.  A bar of several chunks (eight waves: two, sixteen: up to four): waves
    // 4c .. 4c + 3 hold the sub-trees of chunk c -- four of exactly 2 048 elements for a full chunk, nwl of at most 1 928 for the
    // last --, and np.sum adds the chunks one after the other
    __device__ __forceinline__ float tree(float v)
    {
        if constexpr (W == 1) return v;
        else {
            float p[W];
            all(x->f, v, [&](int q, float u) { p[q] = u; });
            if (W >= 8 && nchunk > 1) {
#pragma unroll
                for (int st = 1; st < 4; st <<= 1)
#pragma unroll
                    for (int q = 0; q < W; q += 2 * st)
                        if ((q >> 2) < nchunk - 1 || st < nwl) p[q] = p[q] + p[q + st];
                float tot = p[0];
#pragma unroll
                for (int c = 1; c < W / 4; ++c)
                    if (c < nchunk) tot = tot + p[4 * c];
                return tot;
            } else {
#pragma unroll
                for (int st = 1; st < W; st <<= 1)
#pragma unroll
                    for (int q = 0; q < W; q += 2 * st) p[q] = p[q] + p[q + st];
                return p[0];
            }
        }
    }
    __device__ __forceinline__ int sum(int v)
    {
        if constexpr (W == 1) return v;
        else { int r = 0; all(x->i, v, [&](int, int u) { r += u; }); return r; }
    }
    __device__ __forceinline__ int imin(int v)
    {
        if constexpr (W == 1) return v;
        else { int r = 0x7FFFFFFF; all(x->i, v, [&](int, int u) { r = u < r ? u : r; }); return r; }
    }
    __device__ __forceinline__ int imax(int v)
    {
        if constexpr (W == 1) return v;
        else { int r = (int)0x80000000; all(x->i, v, [&](int, int u) { r = u > r ? u : r; }); return r; }
    }
    __device__ __forceinline__ double dsum(double v)
    {
        if constexpr (W == 1) return v;
        else { double r = 0.0; all(x->d, v, [&](int, double u) { r += u; }); return r; }
    }
};

// One bar (W == 1) or one wave's sub-tree [a, a + wcnt) of a bar of cnt sizes (W > 1) -> the arguments of the two log1p columns
// (evaluated for 32 bars at a time by the caller), pct_block, size_gini; with W > 1 the results are wave 0's.
template <int KMAX, int ROUNDS, int W>
__device__ __forceinline__ void tsm_bar(const float *__restrict__ a, int wcnt, int cnt, const int (&off2)[ROUNDS < 2 ? 2 : ROUNDS],
                                        const int (&len2)[ROUNDS < 2 ? 2 : ROUNDS],
                                        unsigned sp_slot, double th, double theta_mult, int lane, float *slots, float *cbuf,
                                        TsmWg<W> &wg, double &mean_arg, double &p95_arg, float &pct, float &gini)
{
    const int i8 = lane & 7;
    float x[ROUNDS][KMAX], t[ROUNDS];
    int len[ROUNDS];
    // ---- the sizes: element 8k + i of the lane's leaf, and the leaf's tail element i; +0.0 where there is none
#pragma unroll
    for (int R = 0; R < ROUNDS; ++R) {
        len[R] = len2[R];
        const int nm = len[R] & ~7;
        const float *p = a + off2[R] + i8;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) x[R][k] = 8 * k < nm ? p[8 * k] : 0.f;
        t[R] = (i8 < 7 && nm + i8 < len[R]) ? p[nm] : 0.f;
    }
    constexpr int npad_all = 64 * ROUNDS * (KMAX + 1);
    const int npad = npad_all - wcnt;
    const double thr = th * theta_mult;
    // ---- np.sum / np.mean of the float32 slice
    float tf = wg.tree(tsm_tree<KMAX, ROUNDS, false>(x, t, len, sp_slot, [](float v) { return v; }, lane, slots));
    if (tf == 0.f) tf = wg.tree(tsm_tree<KMAX, ROUNDS, true>(x, t, len, sp_slot, [](float v) { return v; }, lane, slots));   // (the sign of a zero total)
    const double mean = (double)(tf / (float)cnt), sum = (double)tf;
    mean_arg = mean / thr;
    // ---- smallest and largest size (the empty slots count as sizes of 0 here: a wider bracket is still a bracket)
    float mnl = x[0][0], mxl = x[0][0];
#pragma unroll
    for (int R = 0; R < ROUNDS; ++R) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) { mnl = __builtin_fminf(mnl, x[R][k]); mxl = __builtin_fmaxf(mxl, x[R][k]); }
        mnl = __builtin_fminf(mnl, t[R]); mxl = __builtin_fmaxf(mxl, t[R]);
    }
    const int kmn = wg.imin(fmk_dpp_reduce(tsm_key(mnl), 0x7FFFFFFF, FmkOpMin()));
    const int kmx = wg.imax(fmk_dpp_reduce(tsm_key(mxl), (int)0x80000000, FmkOpMax()));
    // ---- block volume: (double)a > thr  <=>  a > the largest float32 not above thr
    float thr_f = (float)thr;
    if ((double)thr_f > thr) thr_f = tsm_val(tsm_key(thr_f) - 1);
    double block = 0.0;
    if (tsm_val(kmx) > thr_f || tf != tf) {                           // (the largest size of a bar with a NaN may be one)
        double bl = 0.0;
#pragma unroll
        for (int R = 0; R < ROUNDS; ++R) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k) bl += (double)(x[R][k] > thr_f ? x[R][k] : 0.f);
            bl += (double)(t[R] > thr_f ? t[R] : 0.f);
        }
        block = wg.dsum(fmk_dpp_reduce(bl, 0.0, FmkOpAdd()));
    }
    // ---- np.percentile(., 95), NumPy 2.2's float32 arithmetic (ts_percentile95)
    float p95 = NAN;
    bool has_nan = false;
    if (tf != tf) {                                                    // a NaN size, or +inf and -inf: look
        bool nn = false;
#pragma unroll
        for (int R = 0; R < ROUNDS; ++R) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k) nn |= x[R][k] != x[R][k];
            nn |= t[R] != t[R];
        }
        has_nan = wg.sum(__builtin_amdgcn_ballot_w64(nn) != 0 ? 1 : 0) != 0;
    }
    if (!has_nan) {
        const float q32 = 95.0f / 100.0f;
        const float vi = (float)(cnt - 1) * q32;
        const int fl = (int)floorf(vi);
        const int k1 = fl < cnt - 1 ? fl : cnt - 1;
        const int k2 = k1 + 1 < cnt ? k1 + 1 : cnt - 1;
        // invariant: count(x <= val(lo)) = clo <= k1 and count(x <= val(hi)) = chi > k2, counts of the bar's sizes (the empty slots
        // are taken off: they are <= the pivot iff 0 is); float compares: -0 and +0 count alike, so the image of -0 is never used
        // as the lower end below the smallest size
        int lo = kmn - 1, hi = kmx, clo = 0, chi = cnt;
        if (lo == -1) lo = -2;
        float v1 = 0.f, v2 = 0.f;
        bool done = false;
        // is the element of slot (R, k) / the tail slot of round R a size of the bar?
        int olen[ROUNDS];
        auto real_x = [&](int R, int k) { return 8 * k < (olen[R] & ~7); };
        auto real_t = [&](int R) { return i8 < 7 && (olen[R] & ~7) + i8 < olen[R]; };
        while (chi - clo > 64) {
            if ((unsigned)hi - (unsigned)lo == 1u) { v1 = v2 = tsm_val(hi); done = true; break; }
            const int pivot = lo + (int)(((unsigned)hi - (unsigned)lo) >> 1);
            const float pf = tsm_val(pivot);
            int c = 0.f <= pf ? -npad : 0;
#pragma unroll
            for (int R = 0; R < ROUNDS; ++R) {
#pragma unroll
                for (int k = 0; k < KMAX; ++k) c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(x[R][k] <= pf));
                c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(t[R] <= pf));
            }
            c = wg.sum(c);
            if (c > k2) { hi = pivot; chi = c; }
            else if (c <= k1) { lo = pivot; clo = c; }
            else {
                float bel = -INFINITY, abv = INFINITY;
#pragma unroll
                for (int R = 0; R < ROUNDS; ++R) olen[R] = tsm_opaque(len[R]);
#pragma unroll
                for (int R = 0; R < ROUNDS; ++R) {
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) {
                        bel = real_x(R, k) && x[R][k] <= pf ? __builtin_fmaxf(bel, x[R][k]) : bel;
                        abv = real_x(R, k) && x[R][k] > pf ? __builtin_fminf(abv, x[R][k]) : abv;
                    }
                    bel = real_t(R) && t[R] <= pf ? __builtin_fmaxf(bel, t[R]) : bel;
                    abv = real_t(R) && t[R] > pf ? __builtin_fminf(abv, t[R]) : abv;
                }
                v1 = tsm_val(wg.imax(fmk_dpp_reduce(tsm_key(bel), (int)0x80000000, FmkOpMax())));
                v2 = tsm_val(wg.imin(fmk_dpp_reduce(tsm_key(abv), 0x7FFFFFFF, FmkOpMin())));
                done = true;
                break;
            }
        }
        if (!done) {
            // at most 64 candidates in (val(lo), val(hi)]: one per lane (of wave 0), and on with the bisection
            const float vlo = tsm_val(lo), vhi = tsm_val(hi);
            __builtin_amdgcn_wave_barrier();
            cbuf[lane] = NAN;
            __builtin_amdgcn_wave_barrier();
            int base = 0;
            auto put = [&](float v, bool real) {
                const bool in = real && v > vlo && v <= vhi;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(in);
                const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, (unsigned)base));
                if (in) cbuf[pos] = v;
                base += __builtin_popcountll(m);
            };
            if (vlo < 0.f && 0.f <= vhi) {                             // the empty slots are inside: leave them out by their place
#pragma unroll
                for (int R = 0; R < ROUNDS; ++R) olen[R] = tsm_opaque(len[R]);
#pragma unroll
                for (int R = 0; R < ROUNDS; ++R) {
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) put(x[R][k], real_x(R, k));
                    put(t[R], real_t(R));
                }
            } else {
#pragma unroll
                for (int R = 0; R < ROUNDS; ++R) {
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) put(x[R][k], true);
                    put(t[R], true);
                }
            }
            __builtin_amdgcn_wave_barrier();
            float cv = cbuf[lane];
            if constexpr (W > 1) {
                // wave 0 collects: lane l takes the l-th candidate in wave order
                int src_w = 0, src_at = lane, before = 0;
                bool found = false;
                wg.all(wg.x->i, base, [&](int q, int m) {
                    if (!found && lane < before + m) { src_w = q; src_at = lane - before; found = true; }
                    before += m;
                });
                cv = found ? wg.x->cand[src_w][src_at] : NAN;
            }
            if (W == 1 || wg.wib == 0) {
                const int cbase = clo;
                for (;;) {
                    if ((unsigned)hi - (unsigned)lo == 1u) { v1 = v2 = tsm_val(hi); break; }
                    const int pivot = lo + (int)(((unsigned)hi - (unsigned)lo) >> 1);
                    const float pf = tsm_val(pivot);
                    const int c = cbase + __builtin_popcountll(__builtin_amdgcn_ballot_w64(cv <= pf));
                    if (c > k2) { hi = pivot; chi = c; }
                    else if (c <= k1) { lo = pivot; clo = c; }
                    else {
                        const float bel = cv <= pf ? cv : -INFINITY, abv = cv > pf ? cv : INFINITY;
                        v1 = tsm_val(fmk_dpp_reduce(tsm_key(bel), (int)0x80000000, FmkOpMax()));
                        v2 = tsm_val(fmk_dpp_reduce(tsm_key(abv), 0x7FFFFFFF, FmkOpMin()));
                        break;
                    }
                }
            }
        }
        if (vi >= (float)(cnt - 1)) p95 = v2;
        else {
            const float t32 = vi - floorf(vi), d32 = v2 - v1;
            p95 = v1 + d32 * t32;
            if (t32 >= 0.5f) p95 = v2 - d32 * (1.0f - t32);
        }
    }
    p95_arg = (double)p95 / thr;
    // ---- pct_block and size_gini = 1 - sum((a / total)^2), float32 quotients, squares and pairwise sum (base.py:597-609)
    pct = NAN; gini = NAN;
    if (sum != 0.0) {
        pct = (float)(block / sum);
        const double rinv = 1.0 / (double)tf;
        bool sub = false;
        float g = tsm_tree<KMAX, ROUNDS, false>(x, t, len, sp_slot, [rinv, &sub](float v) {
            const float q = (float)((double)tsm_opaque(v) * rinv);      // (opaque: or the float64 images of all sizes are kept from the block volume on)
            sub |= __builtin_isfpclass(q, 0x0090);                     // a subnormal quotient: not covered by the argument above
            return q * q; }, lane, slots);
        if (wg.sum(__builtin_amdgcn_ballot_w64(sub) != 0 ? 1 : 0) != 0)
            g = tsm_tree<KMAX, ROUNDS, true>(x, t, len, sp_slot, [tf](float v) { const float q = v / tf; return q * q; }, lane, slots);
        gini = cnt == 1 ? 0.f : 1.0f - wg.tree(g);                           // (base.py:606-607: a bar of one tick)
    }
}

// (a call, so that the constants of log1p do not occupy registers across the bar loop)
static __device__ __attribute__((noinline)) float tsm_log1p(double v) { return (float)log1p(v); }

// A bar of TSM_MAX < n <= TSM_MAX5 ticks whose tree has five levels of splits and leaves of at most 95 elements everywhere (decided
// from n alone: the nodes of a level lie between the left-most -- shortest -- and the right-most one) fits ONE wave: four rounds of
// eight leaves, eleven accumulator terms per lane, 48 size registers.  Two waves per bar cost twice the time per tick of one (the
// exchanges), so the one-wave kernel takes these bars -- 2-minute bars of the bench tape: 2 400 ticks, 32 leaves of 75 -- and the
// two-wave kernel skips them (the same test in both).
#define TSM_MAX5 3848                  // (a sixth level from 3 849 elements)
__device__ __forceinline__ bool tsm_one_wave5(int n)
{
    if (n <= TSM_MAX || n > TSM_MAX5) return false;
    int nlo = n, nhi = n, depth = 0;
    bool same_depth = true;
    while (nhi > 128) {
        same_depth = same_depth && nlo > 128;
        nlo = (nlo >> 1) & ~7;
        nhi -= (nhi >> 1) & ~7;
        ++depth;
    }
    return same_depth && depth == 5 && nhi <= 95;
}

// `rest` (a list in k_bar_trade_size's format: [0] = count, [32 ...] = bar numbers): the bars neither this kernel nor the
// workgroup kernels (regular bars of TSM_MAX < ticks <= wg_hi) take
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, 4) void k_bar_trade_size_mid(const float *__restrict__ amount, const double *__restrict__ theta,
                                                                      const int64_t *__restrict__ ci, int64_t nb, int64_t n,
                                                                      double theta_mult, float *__restrict__ o_mean,
                                                                      float *__restrict__ o_p95, float *__restrict__ o_pct,
                                                                      float *__restrict__ o_gini, unsigned long long *__restrict__ rest,
                                                                      int64_t wg_hi)
{
    __shared__ float s_slots[WAVES][16];
    __shared__ unsigned s_sp[WAVES][16];
    __shared__ float s_cbuf[WAVES][64];
    const int lane = fmk_lane();
    const int wib = fmk_uniform((int)(threadIdx.x >> 6));
    const int64_t wave0 = (int64_t)blockIdx.x * WAVES + wib;
    const int64_t nwaves = (int64_t)gridDim.x * WAVES;
    TsmWg<1> wg{nullptr, wib, lane, 0, 1, 0};
    // mean_size_rel and size_95_rel are log1p(. / threshold) in float64: the arguments of up to 32 bars wait in the lanes (2j: the
    // mean of the j-th waiting bar, 2j + 1: its percentile) and are evaluated together -- one log1p per 32 bars instead of two per bar
    double parg = 0.0;
    int64_t pbar = 0;
    int npend = 0;
    auto flush = [&]() {
        const float lg = tsm_log1p(parg);
        if (lane < 2 * npend) ((lane & 1) ? o_p95 : o_mean)[pbar] = lg;
        npend = 0;
    };
    for (int64_t b = wave0; b < nb; b += nwaves) {
        const int64_t s = fmk_uniform(ci[b]), e = fmk_uniform(ci[b + 1]);
        const bool regular = s >= -1 && e <= n - 1;
        if (!(e - s > TSM_MIN && e - s <= TSM_MAX && regular)) {                                  // another launch's
            if (lane == 0 && !(regular && e - s > TSM_MAX && e - s <= wg_hi)) rest[32 + atomicAdd(rest, 1ULL)] = (unsigned long long)b;
            continue;
        }
        const int cnt = (int)(e - s);
        const double th = theta[b];
        double mean_arg = NAN, p95_arg = NAN;
        float pct = NAN, gini = NAN;                                         // base.py:576-579
        if (th != 0.0) {                                                     // base.py:586-587
            // the leaves of this lane's group in the two rounds (paths 2g and 2g + 1), and what the tree's shape says about the
            // instantiation: no split at level 3 (no odd path ends in a leaf) -> one round; leaves of at most 95 elements (every bar
            // of 1 025 .. 1 296 ticks, most bars of up to 760) -> eleven accumulator terms per lane
            int off2[2], len2[2]; unsigned sp2[2];
            const int grp = lane >> 3;
            tsm_leaf(cnt, 2 * grp, off2[0], len2[0], sp2[0]);
            tsm_leaf(cnt, 2 * grp + 1, off2[1], len2[1], sp2[1]);
            const bool one_round = __builtin_amdgcn_ballot_w64(len2[1] > 0) == 0;
            const bool small_leaves = __builtin_amdgcn_ballot_w64(len2[0] > 95 || len2[1] > 95) == 0;
            float *sl = s_slots[wib], *cb = s_cbuf[wib];
            // the split flags of path g for lane g < 16 (the lanes that combine the leaf values)
            if ((lane & 7) == 0) { s_sp[wib][2 * grp] = sp2[0]; s_sp[wib][2 * grp + 1] = sp2[1]; }
            __builtin_amdgcn_wave_barrier();
            const unsigned sp_slot = s_sp[wib][lane & 15];
            __builtin_amdgcn_wave_barrier();
            const float *a = amount + s + 1;
            if (one_round) {
                if (small_leaves) tsm_bar<11, 1, 1>(a, cnt, cnt, off2, len2, sp_slot, th, theta_mult, lane, sl, cb, wg, mean_arg, p95_arg, pct, gini);
                else tsm_bar<16, 1, 1>(a, cnt, cnt, off2, len2, sp_slot, th, theta_mult, lane, sl, cb, wg, mean_arg, p95_arg, pct, gini);
            } else {
                if (small_leaves) tsm_bar<11, 2, 1>(a, cnt, cnt, off2, len2, sp_slot, th, theta_mult, lane, sl, cb, wg, mean_arg, p95_arg, pct, gini);
                else tsm_bar<16, 2, 1>(a, cnt, cnt, off2, len2, sp_slot, th, theta_mult, lane, sl, cb, wg, mean_arg, p95_arg, pct, gini);
            }
        }
        if (lane == 0) { o_pct[b] = pct; o_gini[b] = gini; }
        if ((lane >> 1) == npend) { parg = (lane & 1) ? p95_arg : mean_arg; pbar = b; }
        if (++npend == 32) flush();
    }
    if (npend > 0) flush();
}

// One wave per bar, five levels: the bars of `list` (1 920 < ticks <= 3 824) that pass tsm_one_wave5.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, 3) void k_bar_trade_size_mid5(const float *__restrict__ amount, const double *__restrict__ theta,
                                                                       const int64_t *__restrict__ ci, const int64_t *__restrict__ list,
                                                                       int64_t n, double theta_mult, float *__restrict__ o_mean,
                                                                       float *__restrict__ o_p95, float *__restrict__ o_pct,
                                                                       float *__restrict__ o_gini)
{
    __shared__ float s_slots[WAVES][32];
    __shared__ unsigned s_sp[WAVES][32];
    __shared__ float s_cbuf[WAVES][64];
    const int lane = fmk_lane();
    const int wib = fmk_uniform((int)(threadIdx.x >> 6));
    const int64_t wave0 = (int64_t)blockIdx.x * WAVES + wib;
    const int64_t nwaves = (int64_t)gridDim.x * WAVES;
    TsmWg<1> wg{nullptr, wib, lane, 0, 1, 0};
    double parg = 0.0;                                                       // (the waiting log1p arguments, as in k_bar_trade_size_mid)
    int64_t pbar = 0;
    int npend = 0;
    auto flush = [&]() {
        const float lg = tsm_log1p(parg);
        if (lane < 2 * npend) ((lane & 1) ? o_p95 : o_mean)[pbar] = lg;
        npend = 0;
    };
    const int64_t n_list = list[0];
    for (int64_t q = wave0; q < n_list; q += nwaves) {
        const int64_t b = fmk_uniform(list[1 + q]);
        const int64_t s = fmk_uniform(ci[b]), e = fmk_uniform(ci[b + 1]);
        if (!(s >= -1 && e <= n - 1) || !tsm_one_wave5((int)(e - s))) continue;          // (the two-wave kernel's, or the three-pass one's)
        const int cnt = (int)(e - s);
        const double th = theta[b];
        double mean_arg = NAN, p95_arg = NAN;
        float pct = NAN, gini = NAN;
        if (th != 0.0) {
            // the leaves at the ends of the paths 4g .. 4g + 3
            int off4[4], len4[4]; unsigned sp4[4];
            const int grp = lane >> 3;
#pragma unroll
            for (int R = 0; R < 4; ++R) tsm_leaf<5>(cnt, 4 * grp + R, off4[R], len4[R], sp4[R]);
            if ((lane & 7) == 0) {
#pragma unroll
                for (int R = 0; R < 4; ++R) s_sp[wib][4 * grp + R] = sp4[R];
            }
            __builtin_amdgcn_wave_barrier();
            const unsigned sp_slot = s_sp[wib][lane & 31];
            __builtin_amdgcn_wave_barrier();
            tsm_bar<11, 4, 1>(amount + s + 1, cnt, cnt, off4, len4, sp_slot, th, theta_mult, lane, s_slots[wib], s_cbuf[wib], wg, mean_arg,
                              p95_arg, pct, gini);
        }
        if (lane == 0) { o_pct[b] = pct; o_gini[b] = gini; }
        if ((lane >> 1) == npend) { parg = (lane & 1) ? p95_arg : mean_arg; pbar = b; }
        if (++npend == 32) flush();
    }
    if (npend > 0) flush();
}

#define TSM_WG_PER_WAVE 1912
// Sixteen waves hold a bar of up to four of np.sum's chunks, four waves per chunk -- if its LAST chunk (m elements) fits four sub-trees
// of at most 1 928 elements (m <= 4 * 1 912) or is a full one; the other bars of 15 841 .. 32 768 ticks stay with k_bar_trade_size_wide
// (the same test in both kernels).
__device__ __forceinline__ bool tsm_quad16_fits(int64_t n)
{
    if (n <= FMK_NP_BUFSIZE + 4 * TSM_WG_PER_WAVE || n > 4 * FMK_NP_BUFSIZE) return false;
    const int64_t m = n - ((n + FMK_NP_BUFSIZE - 1) / FMK_NP_BUFSIZE - 1) * FMK_NP_BUFSIZE;
    return m <= 4 * TSM_WG_PER_WAVE || m == FMK_NP_BUFSIZE;
}

// W waves per bar: the regular bars of `list` (fmk_long_bar_lists: 1 920 / 3 824 / 7 648 / 15 840 < ticks <= 3 824 / 7 648 / 15 840 / 32 768 for 2 / 4 / 8 / 16 waves;
// beyond 8 192 ticks a bar is several of np.sum's chunks, four waves per chunk).
template <int W>
__global__ __launch_bounds__(64 * W, 4) void k_bar_trade_size_wg(const float *__restrict__ amount, const double *__restrict__ theta,
                                                                  const int64_t *__restrict__ ci, const int64_t *__restrict__ list,
                                                                  int64_t n, double theta_mult, float *__restrict__ o_mean,
                                                                  float *__restrict__ o_p95, float *__restrict__ o_pct,
                                                                  float *__restrict__ o_gini)
{
    constexpr int LW = W == 2 ? 1 : W == 4 ? 2 : W == 8 ? 3 : 4;
    static_assert((1 << LW) == W, "2, 4, 8 or 16 waves per bar");
    __shared__ TsmX sx;
    __shared__ float s_slots[W][16];
    __shared__ unsigned s_sp[W][16];
    const int lane = fmk_lane();
    const int wib = fmk_uniform((int)(threadIdx.x >> 6));
    TsmWg<W> wg{&sx, wib, lane, 0, 1, 0};
    double parg = 0.0;                                                       // (wave 0: the waiting log1p arguments, as above)
    int64_t pbar = 0;
    int npend = 0;
    auto flush = [&]() {
        const float lg = tsm_log1p(parg);
        if (lane < 2 * npend) ((lane & 1) ? o_p95 : o_mean)[pbar] = lg;
        npend = 0;
    };
    const int64_t n_list = list[0];
    for (int64_t q = blockIdx.x; q < n_list; q += gridDim.x) {
        const int64_t b = fmk_uniform(list[1 + q]);
        const int64_t s = fmk_uniform(ci[b]), e = fmk_uniform(ci[b + 1]);
        if (!(s >= -1 && e <= n - 1)) continue;                              // (irregular close indices: the three-pass kernel's)
        const int cnt = (int)(e - s);
        if (W == 2 && tsm_one_wave5(cnt)) continue;                          // (the one-wave kernel's: five levels, small leaves)
        // the wave's sub-tree.  np.sum adds a bar of more than 8 192 ticks in chunks of 8 192 (fmk_np_sum): up to 8 192 ticks the bar
        // is ONE tree and the W waves follow the split rule along the bits of the wave number; eight or sixteen waves also take a bar
        // of two .. four chunks, four waves per chunk -- a full chunk is four sub-trees of exactly 2 048 elements (a 2 048-element
        // tree splits evenly: four levels), the last one (m ticks) 1, 2 or 4 sub-trees of at most 1 928 (the other waves hold nothing)
        int woff = 0, wlen = cnt, rlen = cnt;
        bool small_leaves = true;
        wg.nchunk = 1; wg.nwl = 0;
        if (W >= 8 && cnt > FMK_NP_BUFSIZE) {
            const int nchunk = (cnt + FMK_NP_BUFSIZE - 1) / FMK_NP_BUFSIZE;
            const int m = cnt - (nchunk - 1) * FMK_NP_BUFSIZE;               // the last chunk: 1 .. 8 192 elements
            const int j1 = m <= TSM_WG_PER_WAVE ? 0 : m <= 2 * TSM_WG_PER_WAVE ? 1 : (m <= 4 * TSM_WG_PER_WAVE || m == FMK_NP_BUFSIZE) ? 2 : 3;
            if (nchunk > W / 4 || j1 == 3) continue;                         // (k_bar_trade_size_wide's: tsm_quad16_fits)
            wg.nchunk = nchunk; wg.nwl = 1 << j1;
            const int chunk = wib >> 2, pth = wib & 3;
            if (chunk < nchunk - 1) { woff = chunk * FMK_NP_BUFSIZE + 2048 * pth; wlen = 2048; }     // (8 192 splits evenly: 4 x 2 048)
            else if (chunk == nchunk - 1 && pth < (1 << j1)) {
                woff = chunk * FMK_NP_BUFSIZE; wlen = m;
                for (int l = 0; l < j1; ++l) {
                    const int n2 = (wlen >> 1) & ~7;
                    if ((pth >> (j1 - 1 - l)) & 1) { woff += n2; wlen -= n2; }
                    else wlen = n2;
                }
            } else { woff = 0; wlen = 0; }
            small_leaves = false;                                            // (a full chunk's leaves hold 128 elements)
        } else {
            if (cnt > FMK_NP_BUFSIZE) continue;                              // (never: the list's edges)
#pragma unroll
            for (int l = 0; l < LW; ++l) {
                const int n2 = (wlen >> 1) & ~7;
                if ((wib >> (LW - 1 - l)) & 1) { woff += n2; wlen -= n2; }
                else wlen = n2;
                rlen -= (rlen >> 1) & ~7;
            }
            if (rlen > TSM_MAX + 8) continue;                                // (never: the list's upper edge)
            // eleven accumulator terms per lane are enough if every leaf of the bar's tree has at most 95 elements: the nodes of a
            // level lie between the left-most (shortest) and the right-most (longest) one; decided from the bar's length alone, the
            // same in every wave
            int nlo = cnt, nhi = cnt;
            while (nhi > 128) {
                small_leaves = small_leaves && nlo > 128;                    // (a leaf next to a node that still splits: up to 128 elements)
                nlo = (nlo >> 1) & ~7;
                nhi -= (nhi >> 1) & ~7;
            }
            small_leaves = small_leaves && nhi <= 95;
        }
        const double th = theta[b];
        double mean_arg = NAN, p95_arg = NAN;
        float pct = NAN, gini = NAN;
        if (th != 0.0) {
            int off2[2], len2[2]; unsigned sp2[2];
            const int grp = lane >> 3;
            tsm_leaf(wlen, 2 * grp, off2[0], len2[0], sp2[0]);
            tsm_leaf(wlen, 2 * grp + 1, off2[1], len2[1], sp2[1]);
            if ((lane & 7) == 0) { s_sp[wib][2 * grp] = sp2[0]; s_sp[wib][2 * grp + 1] = sp2[1]; }
            __builtin_amdgcn_wave_barrier();
            const unsigned sp_slot = s_sp[wib][lane & 15];
            __builtin_amdgcn_wave_barrier();
            const float *a = amount + s + 1 + woff;
            if (small_leaves) tsm_bar<11, 2, W>(a, wlen, cnt, off2, len2, sp_slot, th, theta_mult, lane, s_slots[wib], sx.cand[wib], wg, mean_arg, p95_arg, pct, gini);
            else tsm_bar<16, 2, W>(a, wlen, cnt, off2, len2, sp_slot, th, theta_mult, lane, s_slots[wib], sx.cand[wib], wg, mean_arg, p95_arg, pct, gini);
        }
        if (wib == 0) {
            if (lane == 0) { o_pct[b] = pct; o_gini[b] = gini; }
            if ((lane >> 1) == npend) { parg = (lane & 1) ? p95_arg : mean_arg; pbar = b; }
            if (++npend == 32) flush();
        }
    }
    if (wib == 0 && npend > 0) flush();
}

// ---------------------------------------------------------------------------------------------------------------------
// Bars of more than TSW_MIN = 32 768 ticks (hourly, daily bars), float32 amounts, regular close indices: a WORKGROUP per bar (round 3).
// One wave per bar walks a daily bar's NumPy trees alone -- 33.6 ms per 1e9 ticks of daily bars, 10 ms of hourly ones.  np.sum
// adds the bar in chunks of 8 192 elements, the pairwise tree inside a chunk (fmk_np_sum); the tree of a chunk is a function of
// its length only (halves of n / 2 rounded down to a multiple of 8), so a chunk is eight sub-trees at the ends of the 3-bit paths
// through its top, the waves evaluate the sub-trees of all chunks in turn with the wave-level routine (fmk_pairwise_big: the
// recursion does not know where it started), a thread per chunk folds its eight values left + right and one thread adds the
// chunks in order.  Twice: the float32 total and sum((a / total)^2); the block volume is a float64 sum of float32 values, exact
// in any order: a plain sweep.  Same bits as the wave kernel: every addition has NumPy's operands and order.  (Until the chunks
// were found -- DESIGN.md section 5 -- the top of ONE tree over the whole bar was cut level by level into 33 .. 128 sub-trees.)
// np.percentile(., 95) of bars beyond 65 536 ticks rides on that sweep (k_ts_p95_long's radix select is three more passes over
// the bar: 3.4 / 6.0 ms per 1e9 ticks at hourly / daily bars): like the long-bar median of fmk_ohlcv.hip, a systematic sample of the
// bar (every 64th / 256th size) gives a bracket of keys around the sample's 95 % rank (-+ 3.8 standard deviations of that rank), the sweep counts
// the keys below the bracket and appends the sizes inside it to the bar's candidate slots, and the two ranks are selected among
// the candidates exactly; a bracket that misses (or overflows: heavy ties) falls back to the radix select of the whole bar.
// ---------------------------------------------------------------------------------------------------------------------
#define TSW_MIN 16384                  // sixteen waves per bar beyond, four from TSW_MID_MIN (a workgroup's fixed cost per bar -- the
#define TSW_MID_MIN 4096               // barriers of the cut, the scans and the selections -- grows with its waves)
#define TSW_SAMPLE_MIN 32768           // shorter bars: the radix select on the (L2-resident) bar itself -- its fixed cost decides
template <int TSW_WAVES>
__global__ __launch_bounds__(64 * TSW_WAVES) void k_bar_trade_size_wide(const float *__restrict__ amount, const double *__restrict__ theta,
                                                                      const int64_t *__restrict__ ci, const int64_t *__restrict__ list,
                                                                      int64_t n, double theta_mult, float *__restrict__ o_mean,
                                                                      float *__restrict__ o_p95, float *__restrict__ o_pct,
                                                                      float *__restrict__ o_gini, float *__restrict__ samp,
                                                                      uint32_t *__restrict__ cand, int64_t min_cnt, int64_t max_cnt,
                                                                      int skip_quad16 = 0 /* the bars k_bar_trade_size_wg<16> takes */)
{
    typedef MedKey<false> MK;
    __shared__ float s_tot;
    __shared__ __attribute__((aligned(8))) int s_stk[TSW_WAVES][FMK_PW_PAR_STK];
    __shared__ double s_blk[TSW_WAVES];
    __shared__ float s_p95;
    __shared__ int s_ncand, s_nan;
    __shared__ int64_t s_below[TSW_WAVES];
    const int tid = (int)threadIdx.x;
    const int lane = fmk_lane();
    const int w = fmk_uniform((int)(threadIdx.x >> 6));
    const int64_t n_list = list[0];
    for (int64_t q = blockIdx.x; q < n_list; q += gridDim.x) {
        const int64_t b = list[1 + q], s = ci[b], e = ci[b + 1];
        const int64_t cnt64 = e - s;
        if (!(s >= -1 && e <= n - 1) || cnt64 <= min_cnt || cnt64 > max_cnt) continue;     // another launch's (or the wave kernel's: same test there)
        if (skip_quad16 && tsm_quad16_fits(cnt64)) continue;
        const int cnt = (int)cnt64;
        const float *af = amount + (s + 1);
        const double th = theta[b];
        if (th == 0.0) {                                                  // base.py:586-587
            if (tid == 0) { o_mean[b] = NAN; o_p95[b] = NAN; o_pct[b] = NAN; o_gini[b] = NAN; }
            continue;
        }
        const double thr = th * theta_mult;
        // ---- np.sum's structure (fmk_np_sum): chunks of 8 192 elements added one after the other, the pairwise tree inside a chunk.
        //      Work items: a chunk of at least 2 048 elements is the eight sub-trees at the ends of the 3-bit paths through the top
        //      of its tree (all of those nodes split: they hold more than 128 elements), a shorter last chunk is one item.  The waves
        //      take the items in turn (fmk_pairwise_big: the recursion does not know where it started), the values meet in the bar's
        //      own scratch (behind its sample slots), a thread per chunk folds ((v0+v1)+(v2+v3))+((v4+v5)+(v6+v7)) and one thread adds
        //      the chunks in order.
        const int nchunk = (cnt + FMK_NP_BUFSIZE - 1) / FMK_NP_BUFSIZE, nitem = 8 * nchunk;
        float *gv = samp + ((s + 1) >> 4) + (cnt >> 5);
        auto item = [&](int k, int &off, int &len) {
            const int c = k >> 3, sub = k & 7;
            off = c * FMK_NP_BUFSIZE;
            len = cnt - off < FMK_NP_BUFSIZE ? cnt - off : FMK_NP_BUFSIZE;
            if (len < 2048) { len = sub == 0 ? len : 0; return; }
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const int n2 = (len >> 1) & ~7;
                if ((sub >> (2 - l)) & 1) { off += n2; len -= n2; }
                else len = n2;
            }
        };
        __syncthreads();
        if (tid == 0) { s_ncand = 0; s_nan = 0; }
        __syncthreads();
        // the items' values (gv) -> the bar's: chunk by chunk
        auto combine = [&]() -> float {
            __syncthreads();
            for (int c = tid; c < nchunk; c += 64 * TSW_WAVES) {
                const int clen = cnt - c * FMK_NP_BUFSIZE < FMK_NP_BUFSIZE ? cnt - c * FMK_NP_BUFSIZE : FMK_NP_BUFSIZE;
                const float *g = gv + 8 * c;
                if (clen >= 2048) gv[8 * c] = ((g[0] + g[1]) + (g[2] + g[3])) + ((g[4] + g[5]) + (g[6] + g[7]));
            }
            __syncthreads();
            if (tid == 0) {
                float acc = gv[0];
                for (int c = 1; c < nchunk; ++c) acc = acc + gv[8 * c];
                s_tot = acc;
            }
            __syncthreads();
            return s_tot;
        };
        // ---- np.percentile(., 95)
        const float q32 = 95.0f / 100.0f;
        const float vi = (float)(cnt - 1) * q32;                          // the virtual index in the array's dtype (ts_percentile95)
        const double vfl = floor((double)vi);
        const int64_t k1 = (int64_t)vfl < cnt - 1 ? (int64_t)vfl : cnt - 1;
        const int64_t k2 = k1 + 1 < cnt ? k1 + 1 : cnt - 1;
        const bool sampled = cnt > TSW_SAMPLE_MIN;
        uint32_t *mycand = cand + ((s + 1) >> 2);
        const int cap = cnt >> 2;
        MK::K klo = 0, khi = 0;
        if (sampled) {
            // the bracket from a systematic sample
            const int stride = cnt <= (1 << 21) ? 64 : 256;
            const int64_t nsamp = cnt / stride;                           // >= 1024
            float *mine = samp + ((s + 1) >> 4);
            for (int64_t j = tid; j < nsamp; j += 64 * TSW_WAVES) mine[j] = af[j * stride];
            __syncthreads();
            // the rank of a sample quantile scatters with sqrt(ns p (1 - p)) = 0.218 sqrt(ns) at p = 0.95
            const int g = (int)(0.83f * sqrtf((float)nsamp)) + 4;
            const int64_t c = (int64_t)((double)k1 / (double)(cnt - 1) * (double)(nsamp - 1));
            const int64_t r_lo = c - g > 0 ? c - g : 0, r_hi = c + g + 1 < nsamp - 1 ? c + g + 1 : nsamp - 1;
            bool sn;
            med_block_select<false, 64 * TSW_WAVES>(mine, 0, nsamp, r_lo, r_hi, klo, khi, sn);
            __syncthreads();
        }
        // ---- np.sum of the float32 slice (pairwise)
        for (int k = w; k < nitem; k += TSW_WAVES) {
            int ioff, ilen;
            item(k, ioff, ilen);
            if (ilen == 0) continue;
            const float *a0 = af + ioff;
            const float r = fmk_pairwise_big([a0](int i) { return a0[i]; }, ilen, lane, s_stk[w]);
            if (lane == 0) gv[k] = r;
        }
        // ---- the block volume (base.py:599-603: a float64 sum of float32 values) by a plain sweep; with it the percentile's counts
        double blk = 0.0;
        int64_t below = 0;
        bool nan = false;
        for (int j0 = 0; j0 < cnt; j0 += 64 * TSW_WAVES) {                // (whole waves: the ballots)
            const int j = j0 + tid;
            const bool in_bar = j < cnt;
            const float af_j = in_bar ? af[j] : 0.f;
            const double a = (double)af_j;
            blk += a > thr ? a : 0.0;
            if (sampled) {
                const uint32_t raw = __float_as_uint(af_j);
                const uint32_t k = MK::tokey(raw);
                nan |= in_bar && (k < MK::KEY_NEG_INF || k > MK::KEY_POS_INF);
                below += (in_bar && k < klo) ? 1 : 0;
                const bool inb = in_bar && k >= klo && k <= khi;
                const uint64_t m = __ballot(inb);
                if (m) {
                    int base = 0;
                    if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&s_ncand, (int)__popcll(m));
                    base = __builtin_amdgcn_readlane(base, (int)__builtin_ctzll(m));
                    const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                    if (inb && pos < cap) mycand[pos] = raw;
                }
            }
        }
        blk = fmk_wave_sum(blk);
        below = fmk_dpp_reduce(below, (int64_t)0, FmkOpAdd());
        if (lane == 0) { s_blk[w] = blk; s_below[w] = below; }
        if (__ballot(nan) != 0 && lane == 0) s_nan = 1;
        const float tf = combine();                                        // (its barriers also publish s_blk / s_below / the candidates)
        {
            int64_t bl = 0;
            for (int k = 0; k < TSW_WAVES; ++k) bl += s_below[k];
            const int64_t nc = s_ncand;
            MK::K v1 = 0, v2 = 0;
            bool any_nan = s_nan != 0;
            if (!any_nan) {                                                // (block-uniform)
                if (sampled && bl <= k1 && k2 < bl + nc && nc <= cap)
                    med_block_select<false, 64 * TSW_WAVES>(mycand, 0, nc, k1 - bl, k2 - bl, v1, v2, any_nan);
                else
                    med_block_select<false, 64 * TSW_WAVES>(af, 0, cnt, k1, k2, v1, v2, any_nan);     // short bar / bracket missed or overflowed
            }
            if (tid == 0) {
                const float a32 = (float)MK::value(v1), b32 = (float)MK::value(v2);
                float r32;
                if (any_nan) r32 = NAN;
                else if (vi >= (float)(cnt - 1)) r32 = b32;
                else {
                    const float t32 = vi - floorf(vi), d32 = b32 - a32;
                    r32 = a32 + d32 * t32;
                    if (t32 >= 0.5f) r32 = b32 - d32 * (1.0f - t32);
                }
                s_p95 = r32;
            }
        }
        // ---- sum((a / total)^2): float32 quotients, squares and pairwise sum (base.py:609)
        __syncthreads();
        if (tf != 0.f) {
            for (int k = w; k < nitem; k += TSW_WAVES) {
                int ioff, ilen;
                item(k, ioff, ilen);
                if (ilen == 0) continue;
                const float *a0 = af + ioff;
                const float r = fmk_pairwise_big([a0, tf](int i) { const float x = a0[i] / tf; return x * x; }, ilen, lane, s_stk[w]);
                if (lane == 0) gv[k] = r;
            }
        }
        const float sq = tf != 0.f ? combine() : 0.f;                      // (block-uniform)
        if (tid == 0) {
            double block = 0.0;
            for (int k = 0; k < TSW_WAVES; ++k) block += s_blk[k];
            const double sum = (double)tf, mean = (double)(tf / (float)cnt);       // np.mean divides in float32
            float pct = NAN, gini = NAN;
            if (sum != 0.0) {                                             // base.py:597-598
                pct = (float)(block / sum);
                gini = 1.0f - sq;
            }
            o_mean[b] = (float)log1p(mean / thr);
            o_p95[b] = (float)log1p((double)s_p95 / thr);
            o_pct[b] = pct;
            o_gini[b] = gini;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// ONE LANE PER BAR (round 3, float32 amounts): streams of very short bars -- the reference's other caller builds 1-second bars
// (bar/io.py:484-485: ~20 ticks).  The wave-per-bar kernel above spends a wave, three passes, a cross-lane selection and the
// pairwise-tree machinery on 20 numbers: 91 ms per 1e9 ticks of 1-second bars (profiles/r02_next_rows.txt).  Here a wave takes
// the next <= 64 whole bars whose amounts fit its LDS tile (coalesced fill, 4 B/tick -- the loader of k_bar_ohlcv_lanes) and lane l
// runs the reference's statements for bar l on its own registers:
//   * np.sum / np.mean of the float32 slice: NumPy's pairwise sum of n <= 64 elements IS its leaf -- the plain loop for n < 8,
//     eight accumulators over the elements 8k + i folded ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus the n % 8 tail otherwise --
//     which one lane evaluates in NumPy's order; the same for sum((a / total)^2) (base.py:609);
//   * block volume: float64 sum of the float32 sizes above the threshold (exact in any order; base.py:599-603 with the typed
//     float64 accumulator);
//   * np.percentile(., 95): the lane's keys through the fixed sorting network of fmk_ohlcv.hip's lane kernel (lb_sort), the two
//     order statistics picked by index, NumPy's float32 lerp as in ts_percentile95.
// Bars of more than 64 ticks, with irregular close indices (below -1 / beyond the column: Python slice semantics) or not fitting
// the tile go on a list for the wave-per-bar kernel (list mode).
// ---------------------------------------------------------------------------------------------------------------------
#define TSL_TILE 2048
#define TSL_WAVES 2

// bitonic sorting network on N registers of ONE lane (every index a template constant: the keys stay in VGPRs)
template <int I, int J, int K, int N>
__device__ __forceinline__ void tsl_ce(uint32_t (&r)[N])
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
__device__ __forceinline__ void tsl_stage(uint32_t (&r)[N], std::integer_sequence<int, I...>) { (tsl_ce<I, J, K, N>(r), ...); }
template <int J, int K, int N>
__device__ __forceinline__ void tsl_js(uint32_t (&r)[N])
{
    tsl_stage<J, K, N>(r, std::make_integer_sequence<int, N>{});
    if constexpr (J > 1) tsl_js<J / 2, K, N>(r);
}
template <int K, int N>
__device__ __forceinline__ void tsl_ks(uint32_t (&r)[N])
{
    tsl_js<K / 2, K, N>(r);
    if constexpr (K < N) tsl_ks<K * 2, N>(r);
}
template <int N>
__device__ __forceinline__ void tsl_sort(uint32_t (&r)[N]) { tsl_ks<2, N>(r); }
template <int N, int... I>
__device__ __forceinline__ uint32_t tsl_pick_seq(const uint32_t (&r)[N], int idx, std::integer_sequence<int, I...>)
{
    uint32_t v = r[0];
    ((v = idx == I ? r[I] : v), ...);
    return v;
}
template <int N>
__device__ __forceinline__ uint32_t tsl_pick(const uint32_t (&r)[N], int idx) { return tsl_pick_seq<N>(r, idx, std::make_integer_sequence<int, N>{}); }

// NumPy's pairwise leaf of f(a[0..L)) for one lane, L <= N: blocks of eight elements, all lanes in lockstep up to the wave's
// longest bar (Lmax, wave-uniform); `at(i)` gives element i of the lane's bar (LDS), idle slots are not touched
template <int N, class F>
__device__ __forceinline__ float tsl_leaf(F val, int L, int Lmax)
{
    const int nm = L - (L & 7);
    float r[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float small = 0.f;                                                 // n < 8: res = 0; res += a[i]
#pragma unroll
    for (int K = 0; K < N / 8; ++K) {
        if (8 * K < Lmax) {                                            // wave-uniform
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = 8 * K + q;
                const bool in_tree = i < nm;
                const float v = i < L ? val(i) : 0.f;
                if (K == 0) { r[q] = in_tree ? v : r[q]; if (L < 8 && i < L) small += v; }
                else r[q] = in_tree ? r[q] + v : r[q];
            }
        }
    }
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    if (L < 8) return small;
#pragma unroll
    for (int q = 0; q < 7; ++q) {                                      // the n % 8 tail, in order
        const int i = nm + q;
        if (i < L) res += val(i);
    }
    return res;
}

template <int N>
__device__ __forceinline__ void tsl_bar(const uint32_t *ta, int off, int L, double thr, float &mean_rel, float &p95_rel, float &pct,
                                        float &gini)
{
    typedef MedKey<false> MK;
    const int Lmax = fmk_dpp_reduce(L, 0, FmkOpMax());
    const float *fa = (const float *)ta + off;
    // ---- total (np.sum of the float32 slice), block volume, keys
    const float tf = tsl_leaf<N>([fa](int i) { return fa[i]; }, L, Lmax);
    uint32_t r[N];
    double block = 0.0;
#pragma unroll
    for (int K = 0; K < N / 8; ++K) {
        if (8 * K < Lmax) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = 8 * K + q;
                const uint32_t raw = i < L ? ta[off + i] : 0u;
                r[i] = i < L ? MK::tokey(raw) : MK::MAXK;
                const double a = (double)__uint_as_float(raw);
                block += (i < L && a > thr) ? a : 0.0;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) r[8 * K + q] = MK::MAXK;
        }
    }
    const double mean = (double)(tf / (float)L);                       // np.mean of a float32 slice divides in float32
    const double sum = (double)tf;
    mean_rel = (float)log1p(mean / thr);
    // ---- np.percentile(., 95), every step in float32 (ts_percentile95)
    tsl_sort<N>(r);
    const float vi = (float)(L - 1) * (95.0f / 100.0f);
    const int fl = (int)floorf(vi);
    const int k1 = fl < L - 1 ? fl : L - 1;
    const int k2 = k1 + 1 < L ? k1 + 1 : L - 1;
    const uint32_t v1 = tsl_pick<N>(r, k1), v2 = tsl_pick<N>(r, k2);
    const uint32_t kmx = tsl_pick<N>(r, L > 0 ? L - 1 : 0), kmn = r[0];
    double p95;
    if (kmn < MK::KEY_NEG_INF || kmx > MK::KEY_POS_INF) p95 = NAN;     // a NaN size: np.percentile is NaN
    else {
        const float a32 = (float)MK::value(v1), b32 = (float)MK::value(v2);
        if (vi >= (float)(L - 1)) p95 = (double)b32;
        else {
            const float t32 = vi - floorf(vi), d32 = b32 - a32;
            float r32 = a32 + d32 * t32;
            if (t32 >= 0.5f) r32 = b32 - d32 * (1.0f - t32);
            p95 = (double)r32;
        }
    }
    p95_rel = (float)log1p(p95 / thr);
    // ---- pct_block, size_gini (base.py:597-609)
    pct = NAN; gini = NAN;
    const bool have_total = sum != 0.0;
    const float g = tsl_leaf<N>([fa, tf](int i) { const float q = fa[i] / tf; return q * q; }, L, Lmax);
    if (have_total) {
        pct = (float)(block / sum);
        gini = L == 1 ? 0.f : 1.0f - g;
    }
}

__global__ __launch_bounds__(64 * TSL_WAVES) void k_bar_trade_size_lanes(const float *__restrict__ amount,
                                                                       const double *__restrict__ theta,
                                                                       const int64_t *__restrict__ ci, int64_t nb, int64_t n,
                                                                       double theta_mult, float *__restrict__ o_mean,
                                                                       float *__restrict__ o_p95, float *__restrict__ o_pct,
                                                                       float *__restrict__ o_gini,
                                                                       unsigned long long *__restrict__ grp_mask,
                                                                       int64_t *__restrict__ grp_cnt)
{
    __shared__ uint32_t s_a[TSL_WAVES][TSL_TILE + 2];
    __shared__ int64_t s_ci[TSL_WAVES][66];
    const int lane = fmk_lane();
    const int w = fmk_uniform((int)(threadIdx.x >> 6));
    uint32_t *ta = s_a[w];
    const int64_t ngroups = (nb + 63) >> 6;
    const int64_t nwaves = (int64_t)gridDim.x * TSL_WAVES;
    for (int64_t g = (int64_t)blockIdx.x * TSL_WAVES + w; g < ngroups; g += nwaves) {
        const int64_t B0 = g * 64;
        const int nbg = (int)(nb - B0 < 64 ? nb - B0 : 64);
        __builtin_amdgcn_wave_barrier();
        if (lane <= nbg) s_ci[w][lane] = ci[B0 + lane];
        if (lane == 0 && nbg == 64) s_ci[w][64] = ci[B0 + 64];
        __builtin_amdgcn_wave_barrier();
        unsigned long long rest = 0;                                   // bars of this group left to the wave-per-bar kernel
        // a group whose close indices are not an ascending run inside the column goes to the wave kernel whole
        {
            const bool valid = lane < nbg;
            const int64_t s_l = valid ? s_ci[w][lane] : 0, e_l = valid ? s_ci[w][lane + 1] : 0;
            const bool bad = valid && !(s_l >= -1 && e_l >= s_l && e_l <= n - 1);
            if (__ballot(bad) != 0) {
                rest = nbg == 64 ? ~0ULL : ((1ULL << nbg) - 1);
                if (lane == 0) { grp_mask[g] = rest; grp_cnt[g] = __popcll(rest); }
                continue;
            }
        }
        int bl = 0;
        while (bl < nbg) {
            const int64_t s0 = s_ci[w][bl];
            const int idx = bl + 1 + lane;
            const bool valid = idx <= nbg;
            const int64_t e_l = valid ? s_ci[w][idx] : INT64_MAX;
            const int64_t s_l = valid ? s_ci[w][idx - 1] : 0;
            const int m = __popcll(__ballot(valid && e_l - s0 <= TSL_TILE));    // closes ascend: a prefix of the lanes
            if (m == 0) { rest |= 1ULL << bl; bl += 1; continue; }              // one bar longer than the tile
            const int ntick = (int)(s_ci[w][bl + m] - s0);
            __builtin_amdgcn_wave_barrier();
            const uint32_t *ga = (const uint32_t *)amount + s0 + 1;
#pragma unroll 4
            for (int j = lane; j < ntick; j += 64) ta[j] = ga[j];
            __builtin_amdgcn_wave_barrier();
            const bool owner = lane < m;
            const int L = owner ? (int)(e_l - s_l) : 0;
            const int64_t b = B0 + bl + lane;
            const double th = owner ? theta[b] : 0.0;
            const bool mine = owner && L > 0 && L <= 64;
            const uint64_t longer = __ballot(owner && L > 64);
            rest |= (unsigned long long)longer << bl;
            const int off = mine ? (int)(s_l - s0) : 0;
            const int Lw = mine ? L : 0;
            const double thr = th * theta_mult;
            float mean_rel = NAN, p95_rel = NAN, pct = NAN, gini = NAN;          // base.py:576-579
            if (__ballot(Lw > 32) != 0) tsl_bar<64>(ta, off, Lw, thr, mean_rel, p95_rel, pct, gini);
            else if (__ballot(Lw > 16) != 0) tsl_bar<32>(ta, off, Lw, thr, mean_rel, p95_rel, pct, gini);
            else tsl_bar<16>(ta, off, Lw, thr, mean_rel, p95_rel, pct, gini);
            if (owner && L <= 64) {
                const bool live = L > 0 && th != 0.0;                            // base.py:584-587: empty bar / theta == 0 -> NaN row
                o_mean[b] = live ? mean_rel : NAN;
                o_p95[b] = live ? p95_rel : NAN;
                o_pct[b] = live ? pct : NAN;
                o_gini[b] = live ? gini : NAN;
            }
            bl += m;
        }
        if (lane == 0) { grp_mask[g] = rest; grp_cnt[g] = __popcll(rest); }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// SIXTEEN LANES PER BAR (round 3, float32 amounts): bars of up to 256 ticks (10-second bars: ~200 ticks), four bars per wave.
// The wave-per-bar kernel needs ~34 000 cycles for such a bar (three passes, the tree walk of fmk_pairwise_big, a 64-lane
// selection): 17.5 ms per 1e9 ticks of 10-second bars.  A row of 16 lanes is exactly what NumPy's tree of a <= 256-element sum needs:
//   * n <= 128 is ONE leaf -- eight accumulators r_i over the elements 8k + i: lanes 0..7 of the row are the accumulators, each adds
//     its <= 16 elements in order, the fold ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) is three DPP butterflies (quad_perm, quad_perm,
//     row_half_mirror: a + b in either operand order is the same float), the n % 8 tail follows;
//   * 128 < n <= 256 is TWO leaves (n2 = n/2 rounded down to a multiple of 8, and the rest): lanes 8..15 take the right one,
//     left + right by one more butterfly (row_mirror);
//   * the 95th percentile: the row holds the bar's keys, 16 per lane; a bisection on the key VALUE with row-wide counts
//     (16 compares + four DPP adds per step) finds the smallest key v with count(key <= v) > k1, one more pass gives its successor;
//   * block volume: float64 sum of float32 sizes, exact in any order.
// The four bars of a wave are consecutive, so their amounts are one contiguous range: a coalesced fill of the wave's LDS tile.
// Bars of more than 256 ticks, groups that do not fit the tile, irregular close indices: the leftover list (wave per bar).
// ---------------------------------------------------------------------------------------------------------------------
#define TSR_WAVES 4
// NumPy's pairwise sum of f(a[0..L)), L <= 256, by one row (result in every lane of the row).  steps: wave-uniform bound of the
// accumulator chains (ceil(longest leaf / 8)).
template <class F>
__device__ __forceinline__ float tsr_pairwise(F val, int L, int ri, int steps)
{
    // the row's (<= 2) leaves
    int n2 = L / 2;
    n2 -= n2 % 8;
    const bool two = L > 128;
    const int half = ri >> 3, j = ri & 7;
    const int hoff = (two && half) ? n2 : 0;
    const int hlen = two ? (half ? L - n2 : n2) : (half ? 0 : L);
    const int nm = hlen - (hlen & 7);
    float r = j < nm ? val(hoff + j) : 0.f;
    for (int k = 1; k < steps; ++k) {
        const int i = 8 * k + j;
        if (i < nm) r += val(hoff + i);
    }
    float res = r + fmk_row_xor_f32(r, 1);
    res = res + fmk_row_xor_f32(res, 2);
    res = res + fmk_row_xor_f32(res, 4);
    if (hlen < 8) {                                                    // n < 8: res = 0; res += a[i]
        res = 0.f;
        for (int i = 0; i < 7; ++i)
            if (i < hlen) res += val(hoff + i);
    } else {
        for (int q = 0; q < 7; ++q)                                    // the n % 8 tail, in order
            if (nm + q < hlen) res += val(hoff + nm + q);
    }
    const float other = fmk_row_xor_f32(res, 8);                             // the other half row's leaf
    const float left = half ? other : res, right = half ? res : other;
    return two ? left + right : left;
}

__global__ __launch_bounds__(64 * TSR_WAVES) void k_bar_trade_size_rows(const float *__restrict__ amount,
                                                                      const double *__restrict__ theta,
                                                                      const int64_t *__restrict__ ci, int64_t nb, int64_t n,
                                                                      double theta_mult, float *__restrict__ o_mean,
                                                                      float *__restrict__ o_p95, float *__restrict__ o_pct,
                                                                      float *__restrict__ o_gini,
                                                                      const unsigned long long *__restrict__ only,
                                                                      unsigned long long *__restrict__ rest_out)
{
    typedef MedKey<false> MK;
    __shared__ uint32_t s_a[TSR_WAVES][4][256];                        // a row's bar
    __shared__ unsigned long long s_left[TSR_WAVES][64];               // bars this wave leaves to the wave-per-bar kernel
    const int lane = fmk_lane();
    const int w = fmk_uniform((int)(threadIdx.x >> 6));
    const int row = lane >> 4, ri = lane & 15;
    uint32_t *ta = s_a[w][row];
    const float *fb = (const float *)ta;
    unsigned long long *left = s_left[w];
    int n_left = 0;                                                    // wave-uniform
    // `only` (list mode: [0] = count, [32...] = bar numbers): the bars the lane-per-bar schedule left over
    const int64_t todo = only ? (int64_t)only[0] : nb;
    const int64_t niter = (todo + 3) >> 2;
    const int64_t nwaves = (int64_t)gridDim.x * TSR_WAVES;
    auto flush = [&]() {
        if (n_left == 0) return;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(rest_out, (unsigned long long)n_left);
        base = (unsigned long long)fmk_uniform((int64_t)base);
        __builtin_amdgcn_wave_barrier();
        if (lane < n_left) rest_out[32 + base + lane] = left[lane];
        __builtin_amdgcn_wave_barrier();
        n_left = 0;
    };
    double parg = 0.0;
    int64_t pbar = -1;
    int npend = 0;
    auto flush_log = [&]() {
        if (npend == 0) return;
        const float lg = tsm_log1p(parg);
        if ((ri >> 1) < npend && pbar >= 0) ((ri & 1) ? o_p95 : o_mean)[pbar] = lg;
        npend = 0;
    };
    for (int64_t it = (int64_t)blockIdx.x * TSR_WAVES + w; it < niter; it += nwaves) {
        const int64_t q = 4 * it + row;                                // four bars per wave, one per row of 16 lanes
        const bool have = q < todo;
        const int64_t b = have ? (only ? (int64_t)only[32 + q] : q) : 0;
        const int64_t s_b = have ? ci[b] : 0, e_b = have ? ci[b + 1] : 0;
        const bool regular = s_b >= -1 && e_b >= s_b && e_b <= n - 1;
        const int64_t len_b = e_b - s_b;
        // (an empty bar too: its NaN row.  Not the bars of 249 .. 255 ticks: their right half -- n - (n / 2 rounded down to a multiple of
        // 8) = 129 .. 135 elements -- splits once more in NumPy's tree, three leaves; found by the fuzzer late in round 3, the row
        // schedule had summed it as ONE leaf: an ulp in mean_size_rel / pct_block of such bars)
        const bool mine = have && regular && len_b <= 256 && len_b - ((len_b >> 1) & ~(int64_t)7) <= 128;
        // ---- the others: remembered, handed on in blocks of up to 64
        {
            const uint64_t lo = __ballot(have && !mine && ri == 0);    // lanes 0, 16, 32, 48 speak for their rows
            if (lo) {
                const int pos = n_left + __popcll(lo & ((1ULL << lane) - 1));
                if (have && !mine && ri == 0) left[pos] = (unsigned long long)b;
                n_left += __popcll(lo);
                __builtin_amdgcn_wave_barrier();
                if (n_left > 60) flush();
            }
        }
        if (__ballot(mine) == 0) continue;
        const int L = mine ? (int)len_b : 0;
        // ---- the row's bar into its quarter of the tile (16 lanes, 64 B segments)
        // (all sixteen loads of a lane in flight at once -- element r * 16 + ri is also key r of the lane --, then the tile; a loop of
        // load -> store rounds waited for HBM seven times per 100-tick bar)
        __builtin_amdgcn_wave_barrier();
        uint32_t raw[16];
        {
            const uint32_t *ga = (const uint32_t *)amount + s_b + 1;
#pragma unroll
            for (int r = 0; r < 16; ++r) raw[r] = r * 16 + ri < L ? ga[r * 16 + ri] : 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) ta[r * 16 + ri] = raw[r];
        }
        __builtin_amdgcn_wave_barrier();
        const double th = have ? theta[b] : 0.0;
        const double thr = th * theta_mult;
        // chains: the longest leaf of the wave, in steps of eight
        const int leaf = L > 128 ? L - (L / 2 - (L / 2) % 8) : L;
        const int steps = (fmk_dpp_reduce(leaf, 0, FmkOpMax()) + 7) >> 3;
        const float tf = tsr_pairwise([fb](int i) { return fb[i]; }, L, ri, steps);
        // ---- keys (16 per lane: element r * 16 + ri in key[r]), block volume
        uint32_t key[16];
        uint32_t kmn = MK::MAXK, kmx = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = r * 16 + ri;
            key[r] = i < L ? MK::tokey(raw[r]) : MK::MAXK;
            kmn = key[r] < kmn ? key[r] : kmn;
            kmx = (i < L && key[r] > kmx) ? key[r] : kmx;
        }
        kmn = fmk_row_umin(kmn);
        kmx = fmk_row_umax(kmx);
        // block volume: (double)a > thr  <=>  a > the largest float32 not above thr; skipped when no size of the wave's bars is
        double block = 0.0;
        float thr_f = (float)thr;
        if ((double)thr_f > thr) thr_f = tsm_val(tsm_key(thr_f) - 1);
        const bool nan_bar = kmn < MK::KEY_NEG_INF || kmx > MK::KEY_POS_INF;
        if (__ballot(L > 0 && (nan_bar || (float)MK::value(kmx) > thr_f)) != 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float a = __uint_as_float(raw[r]);
                block += (double)((r * 16 + ri < L && a > thr_f) ? a : 0.f);
            }
            block = fmk_row_sum(block);
        }
        // ---- the two order statistics of np.percentile(., 95)
        const float vi = (float)(L - 1) * (95.0f / 100.0f);
        const int fl = (int)floorf(vi);
        const int k1 = fl < L - 1 ? fl : L - 1;
        const int k2 = k1 + 1 < L ? k1 + 1 : L - 1;
        const int nreg = (fmk_dpp_reduce(L, 0, FmkOpMax()) + 15) >> 4;  // registers that hold keys anywhere in the wave
        // smallest v in [kmn, kmx] with count(key <= v) > k1: invariant c_lo = count(<= lo) <= k1 < count(<= hi) = c_hi.  Once
        // exactly ONE key is left in (lo, hi] it is the answer (a masked maximum) -- ~log2(L) + 2 steps instead of the ~26 a
        // bisection of the key range down to one value takes; only a tied target runs the bisection to its end
        uint32_t lo = kmn - 1, hi = kmx;                               // (lo = kmn - 1 counts 0 keys; a wrap at kmn == 0 is a NaN bar)
        int c_lo = 0, c_hi = L;
        for (int step = 0; step < 40; ++step) {
            const bool open = L > 0 && hi - lo > 1 && c_hi - c_lo > 1;
            if (__ballot(open) == 0) break;
            if (step == 10 || step == 15 || step == 20) {
                // a TIED target (decimal lots) never leaves one key in the bracket: snap the bracket to the smallest and largest key
                // inside it (k_bar_ohlcv_rows explains); all equal: done
                uint32_t mn_in = MK::MAXK, mx_in = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (r < nreg) {
                        const bool in = key[r] > lo && key[r] <= hi;
                        mn_in = (in && key[r] < mn_in) ? key[r] : mn_in;
                        mx_in = (in && key[r] > mx_in) ? key[r] : mx_in;
                    }
                mn_in = fmk_row_umin(mn_in);
                mx_in = fmk_row_umax(mx_in);
                if (open) { hi = mx_in; lo = (mn_in == mx_in ? mx_in : mn_in) - 1; }
                continue;
            }
            const uint32_t pivot = lo + ((hi - lo) >> 1);
            int c = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (r < nreg) c += key[r] <= pivot ? 1 : 0;
            c = fmk_row_sum(c);
            if (open) { if (c > k1) { hi = pivot; c_hi = c; } else { lo = pivot; c_lo = c; } }
        }
        if (L > 0 && hi - lo > 1) {                                    // one key in (lo, hi]: find it
            uint32_t only_key = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) only_key = (key[r] > lo && key[r] <= hi && key[r] > only_key) ? key[r] : only_key;
            hi = fmk_row_umax(only_key);
        }
        const uint32_t v1 = hi;
        int c1 = 0;
        uint32_t nxt = MK::MAXK;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            c1 += key[r] <= v1 ? 1 : 0;
            nxt = (key[r] > v1 && key[r] < nxt) ? key[r] : nxt;
        }
        c1 = fmk_row_sum(c1);
        nxt = fmk_row_umin(nxt);
        const uint32_t v2 = (c1 > k2 || k2 == k1) ? v1 : nxt;
        double p95;
        if (kmn < MK::KEY_NEG_INF || kmx > MK::KEY_POS_INF) p95 = NAN;   // a NaN size
        else {
            const float a32 = (float)MK::value(v1), b32 = (float)MK::value(v2);
            if (vi >= (float)(L - 1)) p95 = (double)b32;
            else {
                const float t32 = vi - floorf(vi), d32 = b32 - a32;
                float r32 = a32 + d32 * t32;
                if (t32 >= 0.5f) r32 = b32 - d32 * (1.0f - t32);
                p95 = (double)r32;
            }
        }
        // ---- results (base.py:591-609)
        const double mean = (double)(tf / (float)L), sum = (double)tf;
        // the shares a / total as float64 products (k_bar_trade_size_mid explains why that is the correctly rounded float32 quotient
        // unless it is subnormal: then the wave repeats with the division)
        const double rinv = 1.0 / (double)tf;
        bool sub = false;
        float gsum = tsr_pairwise([fb, rinv, &sub](int i) {
            const float qq = (float)((double)fb[i] * rinv);
            sub |= __builtin_isfpclass(qq, 0x0090);
            return qq * qq; }, L, ri, steps);
        if (__ballot(sub && L > 0 && tf != 0.f) != 0)
            gsum = tsr_pairwise([fb, tf](int i) { const float qq = fb[i] / tf; return qq * qq; }, L, ri, steps);
        const bool live = L > 0 && th != 0.0;
        if (mine && ri == 0) {
            float pct = NAN, gini = NAN;
            if (live && sum != 0.0) {
                pct = (float)(block / sum);
                gini = L == 1 ? 0.f : 1.0f - gsum;
            }
            o_pct[b] = pct; o_gini[b] = gini;
        }
        // mean_size_rel and size_95_rel = log1p(. / threshold) in float64: the arguments of a row's last eight bars wait in its lanes
        // (2j: the mean of the j-th, 2j + 1: its percentile) and are evaluated together
        if ((ri >> 1) == npend) { parg = live ? (ri & 1 ? p95 : mean) / thr : NAN; pbar = mine ? b : -1; }
        if (++npend == 8) flush_log();
    }
    flush_log();
    flush();
}

// list of the bars the lane kernel left over: rest[0] = count (from the scan's total), rest[32 + pos[g] + k] = the k-th set bit of group g
static __global__ __launch_bounds__(256) void k_tsl_compact(const unsigned long long *__restrict__ grp_mask,
                                                            const int64_t *__restrict__ pos, int64_t groups,
                                                            unsigned long long *__restrict__ rest)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g == 0) rest[0] = (unsigned long long)pos[groups];
    if (g >= groups) return;
    unsigned long long m = grp_mask[g];
    int64_t at = 32 + pos[g];
    while (m) {
        const int bit = __builtin_ctzll(m);
        rest[at++] = (unsigned long long)(g * 64 + bit);
        m &= m - 1;
    }
}

extern "C" int fmk_comp_bar_trade_size_dev(fmk_ctx *ctx, const void *d_amount, int amount_is_f64, int64_t n,
                                           const double *d_theta, const int64_t *d_close_idx, int64_t n_idx,
                                           double theta_mult, float *d_mean_size_rel, float *d_size_95_rel,
                                           float *d_pct_block, float *d_size_gini)
{
    if (n_idx == 1) return FMK_OK;   // zero bars (base.py:549-612 checks theta's length only)
    if (n_idx < 1) return fmk_set_error(ctx, FMK_E_ARG, "negative dimensions are not allowed");
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t nb = n_idx - 1;
    int64_t blocks = fmk_ceil_div(nb, 4);
    const int64_t cap = (int64_t)ctx->n_cu * 64;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if (amount_is_f64) {
        k_bar_trade_size<true><<<(unsigned)blocks, 256, 0, ctx->stream>>>(d_amount, d_theta, d_close_idx, nb, n,
                                                                          theta_mult, d_mean_size_rel, d_size_95_rel,
                                                                          d_pct_block, d_size_gini, 0);
        FMK_LAUNCH_CHECK(ctx);
        return FMK_OK;
    }
    // Streams of short bars: one lane per bar (bars up to 64 ticks) and / or sixteen lanes per bar (up to 256 ticks) first, each
    // handing the bars it cannot take to the next schedule through a list; the wave-per-bar kernel sees what is left.
    // Developer knob FMK_TS_LANES: 0 never, 2 lanes -> rows -> waves whenever the layout allows, 3 rows -> waves.
    {
        const char *lv = getenv("FMK_TS_LANES");
        const int mode = lv ? atoi(lv) : 1;
        const bool many = nb >= (int64_t)ctx->n_cu * 64;
        const bool use_lanes = mode == 2 || (mode == 1 && many && n / nb <= 56);
        const bool use_rows = use_lanes || mode == 3 || (mode == 1 && many && n / nb <= 224);
        if (mode != 0 && ((uintptr_t)d_amount & 3) == 0 && use_rows) {
            const int64_t groups = fmk_ceil_div(nb, 64);
            unsigned long long *rest = nullptr, *rest2 = nullptr, *grp_mask = nullptr;
            int64_t *grp_cnt = nullptr;
            int arc = fmk_alloc(ctx, (size_t)(nb + 32) * 8, (void **)&rest2);
            if (arc == FMK_OK && use_lanes) arc = fmk_alloc(ctx, (size_t)(nb + 32) * 8, (void **)&rest);
            if (arc == FMK_OK && use_lanes) arc = fmk_alloc(ctx, (size_t)groups * 8, (void **)&grp_mask);
            if (arc == FMK_OK && use_lanes) arc = fmk_alloc(ctx, (size_t)(groups + 1) * 8, (void **)&grp_cnt);
            hipError_t le = hipSuccess;
            if (arc == FMK_OK) le = hipMemsetAsync(rest2, 0, 8, ctx->stream);
            if (arc == FMK_OK && le == hipSuccess && use_lanes) {
                int64_t lb = fmk_ceil_div(groups, TSL_WAVES);
                const int64_t lcap = (int64_t)ctx->n_cu * 32;
                if (lb > lcap) lb = lcap;
                k_bar_trade_size_lanes<<<(unsigned)lb, 64 * TSL_WAVES, 0, ctx->stream>>>(
                    (const float *)d_amount, d_theta, d_close_idx, nb, n, theta_mult, d_mean_size_rel, d_size_95_rel, d_pct_block,
                    d_size_gini, grp_mask, grp_cnt);
                arc = fmk_exclusive_scan_i64(ctx, grp_cnt, grp_cnt, groups, true);
                if (arc == FMK_OK)
                    k_tsl_compact<<<(unsigned)fmk_ceil_div(groups, 256), 256, 0, ctx->stream>>>(grp_mask, grp_cnt, groups, rest);
            }
            if (arc == FMK_OK && le == hipSuccess) {
                int64_t rb = use_lanes ? (int64_t)ctx->n_cu * 8 : fmk_ceil_div(fmk_ceil_div(nb, 4), TSR_WAVES);
                if (rb > (int64_t)ctx->n_cu * 16) rb = (int64_t)ctx->n_cu * 16;
                if (rb < 1) rb = 1;
                k_bar_trade_size_rows<<<(unsigned)rb, 64 * TSR_WAVES, 0, ctx->stream>>>(
                    (const float *)d_amount, d_theta, d_close_idx, nb, n, theta_mult, d_mean_size_rel, d_size_95_rel, d_pct_block,
                    d_size_gini, use_lanes ? rest : nullptr, rest2);
                // what is left: wave per bar (their percentile by the in-kernel search; a long bar re-reads itself per step)
                k_bar_trade_size<false><<<(unsigned)(ctx->n_cu * 16), 256, 0, ctx->stream>>>(
                    d_amount, d_theta, d_close_idx, nb, n, theta_mult, d_mean_size_rel, d_size_95_rel, d_pct_block, d_size_gini, 0,
                    rest2);
            }
            if (le == hipSuccess) le = hipGetLastError();
            if (rest) (void)fmk_free(ctx, rest);
            if (rest2) (void)fmk_free(ctx, rest2);
            if (grp_mask) (void)fmk_free(ctx, grp_mask);
            if (grp_cnt) (void)fmk_free(ctx, grp_cnt);
            FMK_TRY(arc);
            FMK_HIP(ctx, le);
            return FMK_OK;
        }
    }
    // float32 amounts: the percentile of the bars beyond the register classes first (256 threads per bar up to 8192 ticks, 1024 beyond)
    int64_t *list_mid = nullptr, *list_long = nullptr;
    FMK_TRY(fmk_long_bar_list(ctx, d_close_idx, nb, n, 64 * 32, nullptr, &list_mid, 8192));
    int rc = fmk_long_bar_list(ctx, d_close_idx, nb, n, 8192, nullptr, &list_long);
    int64_t *list_w = nullptr;                                     // the bars a workgroup may take
    if (rc == FMK_OK) rc = fmk_long_bar_list(ctx, d_close_idx, nb, n, 2048, nullptr, &list_w);
    if (rc == FMK_OK) {
        // bars beyond TSW_MID_MIN ticks: a workgroup per bar, percentile included; scratch: sample slots [start / 16 ...) and candidate slots [start / 4 ...) of the bars
        bool wide_on = n > TSW_MID_MIN;
        // regular bars of 129 .. 1 920 ticks: the one-read wave kernel; up to 1 912 x 16 ticks: 2 .. 16 waves of it per bar
        // (amounts that are not 4-byte aligned: the three-pass kernels, and a workgroup per bar from TSW_MID_MIN ticks)
        const bool mid_on = ((uintptr_t)d_amount & 3) == 0;
        // eight waves hold two of np.sum's chunks (8 192 + 4 x 1 912 ticks), sixteen up to four -- those whose last chunk fits four
        // sub-trees (tsm_quad16_fits); the others of 15 841 .. 32 768 ticks are k_bar_trade_size_wide's, which skips the ones that fit
        const int64_t wg8_top = (int64_t)FMK_NP_BUFSIZE + 4 * TSM_WG_PER_WAVE;
        int64_t wide_min = mid_on ? wg8_top : (int64_t)TSW_MID_MIN;   // shortest bar (ticks) k_bar_trade_size_wide takes
        if (mid_on && wide_min > wg8_top) wide_min = wg8_top;
        const bool quad16 = mid_on && wide_on && wide_min == wg8_top;
        const int64_t wg_upper = mid_on ? (quad16 ? 4 * (int64_t)FMK_NP_BUFSIZE : (wide_on ? wide_min : wg8_top)) : 0;
        // regular bars of cov_lo < ticks <= cov_hi are taken by the workgroup kernels (their percentile included)
        const int64_t cov_lo = mid_on ? (int64_t)TSM_MAX : (wide_on ? wide_min : INT64_MAX);
        const int64_t cov_hi = wide_on ? (int64_t)FMK_PW_BIG_MAX_N : wg_upper;
        k_ts_p95_long<256><<<(unsigned)(ctx->n_cu * 8), 256, 0, ctx->stream>>>((const float *)d_amount, d_close_idx, list_mid, n,
                                                                            d_size_95_rel, cov_lo, cov_hi);
        float *samp = nullptr;
        uint32_t *cand = nullptr;
        if (wide_on) {
            rc = fmk_alloc(ctx, (size_t)((n >> 4) + 64) * 4, (void **)&samp);
            if (rc == FMK_OK) rc = fmk_alloc(ctx, (size_t)((n >> 2) + 64) * 4, (void **)&cand);
        }
        int64_t *wg_lists[4] = {nullptr, nullptr, nullptr, nullptr};
        if (rc == FMK_OK && mid_on) {
            int64_t edge[5] = {TSM_MAX, 2 * TSM_WG_PER_WAVE, 4 * TSM_WG_PER_WAVE, FMK_NP_BUFSIZE + 4 * TSM_WG_PER_WAVE, 4 * FMK_NP_BUFSIZE};
            for (int q = 0; q < 5; ++q) if (edge[q] > wg_upper) edge[q] = wg_upper;
            rc = fmk_long_bar_lists(ctx, d_close_idx, nb, n, 4, edge, nullptr, wg_lists);
        }
        if (rc == FMK_OK) {
        k_ts_p95_long<1024><<<(unsigned)(ctx->n_cu * 2), 1024, 0, ctx->stream>>>((const float *)d_amount, d_close_idx, list_long, n,
                                                                             d_size_95_rel, cov_lo, cov_hi);
        if (wide_on) {
            const int64_t split = wide_min > TSW_MIN ? wide_min : TSW_MIN;
            if (wide_min < split)
                k_bar_trade_size_wide<4><<<(unsigned)(ctx->n_cu * 8), 256, 0, ctx->stream>>>(
                    (const float *)d_amount, d_theta, d_close_idx, list_w, n, theta_mult, d_mean_size_rel, d_size_95_rel, d_pct_block,
                    d_size_gini, samp, cand, wide_min, split, quad16 ? 1 : 0);
            k_bar_trade_size_wide<16><<<(unsigned)(ctx->n_cu * 2), 1024, 0, ctx->stream>>>(
                (const float *)d_amount, d_theta, d_close_idx, list_w, n, theta_mult, d_mean_size_rel, d_size_95_rel, d_pct_block,
                d_size_gini, samp, cand, split, (int64_t)FMK_PW_BIG_MAX_N, quad16 ? 1 : 0);
        }
        unsigned long long *rest = nullptr;                           // the bars the one-read kernels leave to the three-pass ones
        if (mid_on) {
            const float *af = (const float *)d_amount;
            k_bar_trade_size_mid5<4><<<(unsigned)(ctx->n_cu * 16), 256, 0, ctx->stream>>>(af, d_theta, d_close_idx, wg_lists[0], n, theta_mult,
                                                                                      d_mean_size_rel, d_size_95_rel, d_pct_block, d_size_gini);
            k_bar_trade_size_wg<2><<<(unsigned)(ctx->n_cu * 16), 128, 0, ctx->stream>>>(af, d_theta, d_close_idx, wg_lists[0], n, theta_mult,
                                                                                    d_mean_size_rel, d_size_95_rel, d_pct_block, d_size_gini);
            k_bar_trade_size_wg<4><<<(unsigned)(ctx->n_cu * 8), 256, 0, ctx->stream>>>(af, d_theta, d_close_idx, wg_lists[1], n, theta_mult,
                                                                                   d_mean_size_rel, d_size_95_rel, d_pct_block, d_size_gini);
            k_bar_trade_size_wg<8><<<(unsigned)(ctx->n_cu * 4), 512, 0, ctx->stream>>>(af, d_theta, d_close_idx, wg_lists[2], n, theta_mult,
                                                                                   d_mean_size_rel, d_size_95_rel, d_pct_block, d_size_gini);
            k_bar_trade_size_wg<16><<<(unsigned)(ctx->n_cu * 2), 1024, 0, ctx->stream>>>(af, d_theta, d_close_idx, wg_lists[3], n, theta_mult,
                                                                                     d_mean_size_rel, d_size_95_rel, d_pct_block, d_size_gini);
            rc = fmk_alloc(ctx, (size_t)(nb + 32) * 8, (void **)&rest);
            if (rc == FMK_OK && hipMemsetAsync(rest, 0, 8, ctx->stream) != hipSuccess) rc = FMK_E_HIP;
            if (rc == FMK_OK)
                k_bar_trade_size_mid<4><<<(unsigned)blocks, 256, 0, ctx->stream>>>(af, d_theta, d_close_idx, nb, n, theta_mult,
                                                                                   d_mean_size_rel, d_size_95_rel, d_pct_block,
                                                                                   d_size_gini, rest, cov_hi);
        }
        // the other bars of at most 1 280 ticks by the instantiation without the 32-key class, the rest by the full one
        if (rc == FMK_OK) {
        k_bar_trade_size<false, true><<<(unsigned)blocks, 256, 0, ctx->stream>>>(d_amount, d_theta, d_close_idx, nb, n,
                                                                                 theta_mult, d_mean_size_rel, d_size_95_rel,
                                                                                 d_pct_block, d_size_gini, 1, rest, INT64_MAX);
        k_bar_trade_size<false><<<(unsigned)blocks, 256, 0, ctx->stream>>>(d_amount, d_theta, d_close_idx, nb, n,
                                                                           theta_mult, d_mean_size_rel, d_size_95_rel,
                                                                           d_pct_block, d_size_gini, 1, rest,
                                                                           cov_lo, (int64_t)64 * 20, cov_hi);
        }
        if (rest) (void)fmk_free(ctx, rest);
        }
        if (wg_lists[0]) (void)fmk_free(ctx, wg_lists[0]);
        if (samp) (void)fmk_free(ctx, samp);
        if (cand) (void)fmk_free(ctx, cand);
    }
    const hipError_t le = hipGetLastError();
    (void)fmk_free(ctx, list_mid);
    if (list_long) (void)fmk_free(ctx, list_long);
    if (list_w) (void)fmk_free(ctx, list_w);
    FMK_TRY(rc);
    FMK_HIP(ctx, le);
    return FMK_OK;
}
