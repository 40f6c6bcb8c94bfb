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
// Traffic: amount column once (4 or 8 B/tick) + 8 B/bar.
#include <math.h>

#include "fmk_common.h"

template <bool AF64> struct MedKey;
template <> struct MedKey<false> {
    typedef uint32_t K;
    static constexpr int BITS = 32;
    static constexpr K MAXK = 0xFFFFFFFFu;
    __device__ static __forceinline__ K load(const void *p, int64_t j)
    {
        uint32_t u = ((const uint32_t *)p)[j];
        return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    }
    __device__ static __forceinline__ double value(K k)
    {
        uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
        return (double)__uint_as_float(u);
    }
    __device__ static __forceinline__ bool is_nan(K k)
    {
        uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
        return (u & 0x7FFFFFFFu) > 0x7F800000u;
    }
};
template <> struct MedKey<true> {
    typedef uint64_t K;
    static constexpr int BITS = 64;
    static constexpr K MAXK = 0xFFFFFFFFFFFFFFFFull;
    __device__ static __forceinline__ K load(const void *p, int64_t j)
    {
        uint64_t u = ((const uint64_t *)p)[j];
        return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
    }
    __device__ static __forceinline__ double value(K k)
    {
        uint64_t u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
        return __longlong_as_double((long long)u);
    }
    __device__ static __forceinline__ bool is_nan(K k)
    {
        uint64_t u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
        return (u & 0x7FFFFFFFFFFFFFFFull) > 0x7FF0000000000000ull;
    }
};

__device__ __forceinline__ int med_popc(bool p) { return __popcll(__ballot(p)); }

template <typename K>
__device__ __forceinline__ K med_wave_umin(K v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { K w = __shfl_xor(v, o, 64); v = w < v ? w : v; }
    return v;
}
template <typename K>
__device__ __forceinline__ K med_wave_umax(K v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { K w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
    return v;
}

// ascending bitonic sort of one key per lane
template <typename K>
__device__ __forceinline__ K med_bitonic64(K v, int lane)
{
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            K w = __shfl_xor(v, j, 64);
            bool up = (lane & k) == 0;
            bool lower = (lane & j) == 0;
            K mn = w < v ? w : v, mx = w < v ? v : w;
            v = (lower == up) ? mn : mx;
        }
    }
    return v;
}

// Abstract access to the bar's keys: NREG > 0 -> register file, NREG == 0 -> re-read from memory.
template <bool AF64, int NREG>
struct MedBar {
    typedef MedKey<AF64> MK;
    typedef typename MK::K K;
    K key[NREG > 0 ? NREG : 1];
    const void *amount;
    int64_t start, cnt;
    int lane;

    __device__ __forceinline__ bool load_all()   // returns "any NaN in my lane"
    {
        bool nan = false;
        if constexpr (NREG > 0) {
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                int64_t j = (int64_t)r * 64 + lane;
                K k = MK::MAXK;
                if (j < cnt) {
                    k = MK::load(amount, start + j);
                    nan |= MK::is_nan(k);
                }
                key[r] = k;
            }
        } else {
            for (int64_t j = lane; j < cnt; j += 64) nan |= MK::is_nan(MK::load(amount, start + j));
        }
        return nan;
    }
    __device__ __forceinline__ void minmax(K &mn, K &mx)
    {
        K a = MK::MAXK, b = 0;
        if constexpr (NREG > 0) {
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                bool valid = (int64_t)r * 64 + lane < cnt;
                a = key[r] < a ? key[r] : a;
                b = (valid && key[r] > b) ? key[r] : b;
            }
        } else {
            for (int64_t j = lane; j < cnt; j += 64) {
                K k = MK::load(amount, start + j);
                a = k < a ? k : a;
                b = k > b ? k : b;
            }
        }
        mn = med_wave_umin<K>(a);
        mx = med_wave_umax<K>(b);
    }
    // number of keys <= pivot (pivot < MAXK, so register sentinels never count)
    __device__ __forceinline__ int64_t count_le(K pivot)
    {
        if constexpr (NREG > 0) {
            int c = 0;
#pragma unroll
            for (int r = 0; r < NREG; ++r) c += med_popc(key[r] <= pivot);
            return c;
        } else {
            int64_t c = 0;
            for (int64_t j = lane; j < cnt; j += 64) c += MK::load(amount, start + j) <= pivot;
            return fmk_wave_sum(c);
        }
    }
    // largest key <= pivot and smallest key > pivot (the latter among real keys only)
    __device__ __forceinline__ void split(K pivot, K &below, K &above)
    {
        K a = 0, b = MK::MAXK;
        if constexpr (NREG > 0) {
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                bool valid = (int64_t)r * 64 + lane < cnt;
                K k = key[r];
                a = (k <= pivot && k > a) ? k : a;
                b = (valid && k > pivot && k < b) ? k : b;
            }
        } else {
            for (int64_t j = lane; j < cnt; j += 64) {
                K k = MK::load(amount, start + j);
                a = (k <= pivot && k > a) ? k : a;
                b = (k > pivot && k < b) ? k : b;
            }
        }
        below = med_wave_umax<K>(a);
        above = med_wave_umin<K>(b);
    }
    // write the keys in (lo, hi] (at most 64 of them) to buf[0..m), one slot each
    __device__ __forceinline__ void compact(K lo, K hi, K *buf)
    {
        int base = 0;
        if constexpr (NREG > 0) {
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                K k = key[r];
                bool in = k > lo && k <= hi;
                uint64_t m = __ballot(in);
                int pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                if (in) buf[pos] = k;
                base += __popcll(m);
            }
        } else {
            for (int64_t j0 = 0; j0 < cnt; j0 += 64) {
                int64_t j = j0 + lane;
                K k = j < cnt ? MK::load(amount, start + j) : MK::MAXK;
                bool in = j < cnt && k > lo && k <= hi;
                uint64_t m = __ballot(in);
                int pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                if (in) buf[pos] = k;
                base += __popcll(m);
            }
        }
    }
};

template <bool AF64, int NREG>
__device__ __forceinline__ double med_select(const void *amount, int64_t start, int64_t cnt, int lane,
                                          typename MedKey<AF64>::K *buf)
{
    typedef MedKey<AF64> MK;
    typedef typename MK::K K;
    MedBar<AF64, NREG> bar;
    bar.amount = amount; bar.start = start; bar.cnt = cnt; bar.lane = lane;
    bool nan = bar.load_all();
    if (__ballot(nan) != 0) return NAN;    // np.median propagates NaN
    const int64_t k1 = (cnt - 1) >> 1, k2 = cnt >> 1;   // the two middle ranks (equal when cnt is odd)
    K mn, mx;
    bar.minmax(mn, mx);
    // invariant: count(key <= lo) = clo <= k1  and  count(key <= hi) = chi > k2
    K lo = mn - 1, hi = mx;     // mn >= 1 for every non-NaN float key
    int64_t clo = 0, chi = cnt;
    K v1, v2;
    for (;;) {
        if (chi - clo <= 64) {
            // <= 64 candidates in (lo, hi]: compact -> sort across lanes -> read the ranks
            buf[lane] = MK::MAXK;
            __builtin_amdgcn_wave_barrier();
            bar.compact(lo, hi, buf);
            __builtin_amdgcn_wave_barrier();
            K v = med_bitonic64<K>(buf[lane], lane);
            __builtin_amdgcn_wave_barrier();
            v1 = __shfl(v, (int)(k1 - clo), 64);
            v2 = __shfl(v, (int)(k2 - clo), 64);
            break;
        }
        if (hi - lo == 1) { v1 = v2 = hi; break; }      // all candidates are the same key
        K pivot = lo + ((hi - lo) >> 1);
        int64_t c = bar.count_le(pivot);
        if (c > k2) { hi = pivot; chi = c; }
        else if (c <= k1) { lo = pivot; clo = c; }
        else {   // k1 < c <= k2: the pivot separates the two middle ranks
            bar.split(pivot, v1, v2);
            break;
        }
    }
    // np.median: mean of the two middle elements == (a + b) / 2.0 ; odd count: the middle one
    return (cnt & 1) ? MK::value(v1) : (MK::value(v1) + MK::value(v2)) / 2.0;
}

template <bool AF64>
__global__ __launch_bounds__(256) void k_bar_median(const void *__restrict__ amount,
                                                    const int64_t *__restrict__ ci, int64_t nb,
                                                    double *__restrict__ o_median)
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
        double m = 0.0;
        if (cnt > 0) {
            const int64_t start = s + 1;
            const int nreg = (int)((cnt + 63) >> 6);
            if (cnt > 64 * 32) m = med_select<AF64, 0>(amount, start, cnt, lane, buf);
            else if (nreg <= 1) m = med_select<AF64, 1>(amount, start, cnt, lane, buf);
            else if (nreg <= 4) m = med_select<AF64, 4>(amount, start, cnt, lane, buf);
            else if (nreg <= 8) m = med_select<AF64, 8>(amount, start, cnt, lane, buf);
            else if (nreg <= 12) m = med_select<AF64, 12>(amount, start, cnt, lane, buf);
            else if (nreg <= 16) m = med_select<AF64, 16>(amount, start, cnt, lane, buf);
            else if (nreg <= 20) m = med_select<AF64, 20>(amount, start, cnt, lane, buf);
            else if (nreg <= 24) m = med_select<AF64, 24>(amount, start, cnt, lane, buf);
            else if constexpr (!AF64) m = med_select<AF64, 32>(amount, start, cnt, lane, buf);
            else m = med_select<AF64, 0>(amount, start, cnt, lane, buf);
        }
        if (lane == 0) o_median[b] = m;
    }
}

extern "C" int fmk_comp_bar_median_dev(fmk_ctx *ctx, const void *d_amount, int amount_is_f64, int64_t n,
                                       const int64_t *d_close_idx, int64_t n_idx, double *d_median)
{
    (void)n;
    if (n_idx < 2)
        return fmk_set_error(ctx, FMK_E_ARG, "Bar close indices must contain at least two elements.");
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t nb = n_idx - 1;
    int64_t blocks = fmk_ceil_div(nb, 4);
    const int64_t cap = (int64_t)ctx->n_cu * 64;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if (amount_is_f64)
        k_bar_median<true><<<(unsigned)blocks, 256, 0, ctx->stream>>>(d_amount, d_close_idx, nb, d_median);
    else
        k_bar_median<false><<<(unsigned)blocks, 256, 0, ctx->stream>>>(d_amount, d_close_idx, nb, d_median);
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}
