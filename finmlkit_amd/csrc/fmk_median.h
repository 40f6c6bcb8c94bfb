// fmk_median.h -- exact per-bar order statistics (shared by fmk_median.hip and the fused small-bar
// kernel of fmk_ohlcv.hip).  See fmk_median.hip for the algorithm description.
#pragma once
#include <math.h>

#include "fmk_common.h"
#include "fmk_dpp.h"

// Order-preserving map float bits -> unsigned ("key").  Every non-NaN float maps into
// [key(-inf), key(+inf)]; NaNs map outside that interval, so a bar contains a NaN iff the
// min/max of its keys leaves it -- no per-element NaN test is needed.
template <bool AF64> struct MedKey;
template <> struct MedKey<false> {
    typedef uint32_t K;
    static constexpr int BITS = 32;
    static constexpr K MAXK = 0xFFFFFFFFu;
    static constexpr K KEY_NEG_INF = 0x007FFFFFu;   // ~0xFF800000
    static constexpr K KEY_POS_INF = 0xFF800000u;   // 0x7F800000 | sign
    __device__ static __forceinline__ K tokey(K u) { return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
    __device__ static __forceinline__ K load(const void *p, int64_t j) { return tokey(((const uint32_t *)p)[j]); }
    __device__ static __forceinline__ double value(K k)
    {
        uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
        return (double)__uint_as_float(u);
    }
};
template <> struct MedKey<true> {
    typedef uint64_t K;
    static constexpr int BITS = 64;
    static constexpr K MAXK = 0xFFFFFFFFFFFFFFFFull;
    static constexpr K KEY_NEG_INF = 0x000FFFFFFFFFFFFFull;
    static constexpr K KEY_POS_INF = 0xFFF0000000000000ull;
    __device__ static __forceinline__ K tokey(K u) { return (u >> 63) ? ~u : (u | 0x8000000000000000ull); }
    __device__ static __forceinline__ K load(const void *p, int64_t j) { return tokey(((const uint64_t *)p)[j]); }
    __device__ static __forceinline__ double value(K k)
    {
        uint64_t u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
        return __longlong_as_double((long long)u);
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

// The bar's keys.  NREG > 0: register file, key[r] of lane l is tick r*64+l, unused slots hold the
// sentinel MAXK.  EXACT: the bar has exactly NREG chunks, so only register NREG-1 can hold sentinels.
// NREG == 0: keys are re-read from memory on every pass (bars too long for the register file).
template <bool AF64, int NREG, bool EXACT = false>
struct MedBar {
    typedef MedKey<AF64> MK;
    typedef typename MK::K K;
    K key[NREG > 0 ? NREG : 1];
    const void *amount;
    int64_t start, cnt;
    int lane;

    __device__ __forceinline__ void load_all()
    {
        if constexpr (NREG > 0) {
            // raw words first, keys afterwards: with the key conversion inside each guarded load the compiler waits for every load
            // before it issues the next (LW x NREG in the ISA, tools/isa_loadwaits.py) -- the words are pinned behind the last load
            K raw[NREG];
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                const int64_t j = (int64_t)r * 64 + lane;
                raw[r] = 0;
                if (j < cnt) raw[r] = ((const K *)amount)[start + j];
            }
#pragma unroll
            for (int r = 0; r < NREG; ++r) asm volatile("" : "+v"(raw[r]));
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                const int64_t j = (int64_t)r * 64 + lane;
                key[r] = j < cnt ? MK::tokey(raw[r]) : MK::MAXK;
            }
        }
    }
    __device__ __forceinline__ bool maybe_invalid(int r) const { return !EXACT || r == NREG - 1; }
    __device__ __forceinline__ void minmax(K &mn, K &mx)
    {
        K a = MK::MAXK, b = 0;
        if constexpr (NREG > 0) {
#pragma unroll
            for (int r = 0; r < NREG; ++r) {
                const K k = key[r];
                a = k < a ? k : a;
                if (maybe_invalid(r)) b = (k != MK::MAXK && k > b) ? k : b;
                else b = k > b ? k : b;
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
                const K k = key[r];
                a = (k <= pivot && k > a) ? k : a;
                b = (k > pivot && k < b) ? k : b;      // a sentinel can only "win" when no real key is > pivot
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
    // smallest and largest key in (lo, hi] (register classes)
    __device__ __forceinline__ void inside_minmax(K lo, K hi, K &mn, K &mx)
    {
        K a = MK::MAXK, b = 0;
#pragma unroll
        for (int r = 0; r < (NREG > 0 ? NREG : 1); ++r) {
            const K k = key[r];
            const bool in = k > lo && k <= hi;
            a = (in && k < a) ? k : a;
            b = (in && k > b) ? k : b;
        }
        mn = med_wave_umin<K>(a);
        mx = med_wave_umax<K>(b);
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

// Exact order statistics of ranks k1 <= k2 <= k1+1 (0-based) of the bar's keys; bar.key[] (register
// path) must already hold the keys.  Returns false when the bar contains a NaN (keys outside [-inf, +inf]).
// SNAP (register classes): more than 64 keys left after 10 / 16 / 22 halvings means the ranks sit in a TIE (decimal lot sizes) and the
// bracket would go on halving an empty key range down to one value; it is then set to the smallest and largest key inside it
// (the counts at its ends stay what they are).  Off where the instantiation's register budget is tuned (the fused small-bar kernel).
template <bool AF64, int NREG, bool EXACT, bool SNAP = false>
__device__ __forceinline__ bool med_rank_pair(MedBar<AF64, NREG, EXACT> &bar, typename MedKey<AF64>::K *buf,
                                              int64_t k1, int64_t k2, typename MedKey<AF64>::K &v1,
                                              typename MedKey<AF64>::K &v2)
{
    typedef MedKey<AF64> MK;
    typedef typename MK::K K;
    const int lane = bar.lane;
    const int64_t cnt = bar.cnt;
    K mn, mx;
    bar.minmax(mn, mx);
    if (mn < MK::KEY_NEG_INF || mx > MK::KEY_POS_INF) return false;
    // invariant: count(key <= lo) = clo <= k1  and  count(key <= hi) = chi > k2
    K lo = mn - 1, hi = mx;
    int64_t clo = 0, chi = cnt;
    for (int nstep = 0;; ++nstep) {
        if constexpr (SNAP && NREG > 0) {
            if (chi - clo > 64 && hi - lo > 1 && (nstep == 10 || nstep == 16 || nstep == 22)) {
                K mn_in, mx_in;
                bar.inside_minmax(lo, hi, mn_in, mx_in);
                hi = mx_in;
                lo = (mn_in == mx_in ? mx_in : mn_in) - 1;
                continue;
            }
        }
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
        else {   // k1 < c <= k2: the pivot separates the two ranks
            bar.split(pivot, v1, v2);
            break;
        }
    }
    return true;
}

// np.median of the bar's amounts (NaN if any amount is NaN, like NumPy).
template <bool AF64, int NREG, bool EXACT, bool SNAP = false>
__device__ __forceinline__ double med_search(MedBar<AF64, NREG, EXACT> &bar, typename MedKey<AF64>::K *buf)
{
    typedef MedKey<AF64> MK;
    typename MK::K v1, v2;
    const int64_t cnt = bar.cnt;
    if (!med_rank_pair<AF64, NREG, EXACT, SNAP>(bar, buf, (cnt - 1) >> 1, cnt >> 1, v1, v2)) return NAN;
    // np.median: mean of the two middle elements == (a + b) / 2.0 ; odd count: the middle one
    return (cnt & 1) ? MK::value(v1) : (MK::value(v1) + MK::value(v2)) / 2.0;
}

template <bool AF64, int NREG>
__device__ __forceinline__ double med_select(const void *amount, int64_t start, int64_t cnt, int lane,
                                             typename MedKey<AF64>::K *buf)
{
    MedBar<AF64, NREG, false> bar;
    bar.amount = amount; bar.start = start; bar.cnt = cnt; bar.lane = lane;
    bar.load_all();
    return med_search<AF64, NREG, false>(bar, buf);
}

// Exact order statistics of ranks rk0 <= rk1 (0-based) of a bar by a whole WORKGROUP (THREADS threads, all of them call): radix
// select on 11-bit digits, most significant first -- 3 passes over a float32 bar (11 + 11 + 10 bits), 6 over a float64 one --
// with 2048-bin LDS histograms of the current digit for the keys that share the prefix found so far (one histogram while both
// ranks still share it); a block scan picks the digit.  (8-bit digits: 4 / 8 passes, 6.5 ms instead of 5.4 per 1e9 float32 ticks.)
// k1 / k2 / any_nan are valid in every thread afterwards (any_nan: a key outside [-inf, +inf]).
#define MED_SEL_BITS 11
#define MED_SEL_BINS (1 << MED_SEL_BITS)
template <bool AF64, int THREADS>
__device__ __forceinline__ void med_block_select(const void *__restrict__ amount, int64_t start, int64_t cnt, int64_t rank0,
                                                 int64_t rank1, typename MedKey<AF64>::K &k1, typename MedKey<AF64>::K &k2,
                                                 bool &any_nan)
{
    typedef MedKey<AF64> MK;
    typedef typename MK::K K;
    constexpr int D = (MK::BITS + MED_SEL_BITS - 1) / MED_SEL_BITS;
    constexpr int PER = MED_SEL_BINS / THREADS;              // bins per thread in the scans
    constexpr int NW = THREADS / 64;
    __shared__ unsigned hist[2][MED_SEL_BINS];
    __shared__ K s_prefix[2];
    __shared__ int64_t s_rank[2];
    __shared__ unsigned s_wsum[2][NW];
    __shared__ int s_nan;
    const int tid = threadIdx.x, lane = fmk_lane(), w = tid >> 6;
    __syncthreads();                                     // (the previous bar's results have been read)
    if (tid == 0) { s_prefix[0] = 0; s_prefix[1] = 0; s_rank[0] = rank0; s_rank[1] = rank1; s_nan = 0; }
    bool nan = false;
#pragma unroll 1
    for (int p = 0; p < D; ++p) {
#pragma unroll
        for (int q = 0; q < PER; ++q) { hist[0][tid + q * THREADS] = 0; hist[1][tid + q * THREADS] = 0; }
        __syncthreads();
        const K pre0 = s_prefix[0], pre1 = s_prefix[1];
        const int64_t rk0 = s_rank[0], rk1 = s_rank[1];
        const bool same = pre0 == pre1;
        const int rest = MK::BITS - MED_SEL_BITS * (p + 1);  // bits below this digit (negative in the last pass of an uneven split)
        const int shift = rest > 0 ? rest : 0;
        const int width = rest >= 0 ? MED_SEL_BITS : MED_SEL_BITS + rest;
        const unsigned mask = (1u << width) - 1u;
        for (int64_t j = tid; j < cnt; j += THREADS) {
            const K k = MK::load(amount, start + j);
            if (p == 0) nan |= k < MK::KEY_NEG_INF || k > MK::KEY_POS_INF;
            const K hi = p == 0 ? (K)0 : (K)(k >> (shift + width));
            const unsigned d = (unsigned)(k >> shift) & mask;
            if (hi == pre0) atomicAdd(&hist[0][d], 1u);
            if (!same && hi == pre1) atomicAdd(&hist[1][d], 1u);
        }
        __syncthreads();
        // thread `tid` owns the bins tid * PER .. + PER - 1: inclusive prefix over all bins, then the bin that holds the rank
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned *hh = hist[same ? 0 : t];
            unsigned h[PER], tot = 0;
#pragma unroll
            for (int q = 0; q < PER; ++q) { h[q] = hh[tid * PER + q]; tot += h[q]; }
            unsigned inc = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
            if (lane == 63) s_wsum[t][w] = inc;
            __syncthreads();
            unsigned base = 0;
            for (int k = 0; k < w; ++k) base += s_wsum[t][k];
            int64_t cum = (int64_t)base + inc - tot;         // keys in the bins before mine
            const int64_t rk = t == 0 ? rk0 : rk1;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                if (cum <= rk && rk < cum + h[q]) {          // exactly one bin of one thread
                    s_prefix[t] = (K)(((t == 0 ? pre0 : pre1) << width) | (K)(tid * PER + q));
                    s_rank[t] = rk - cum;
                }
                cum += h[q];
            }
        }
        __syncthreads();
    }
    if (nan) s_nan = 1;
    __syncthreads();
    k1 = s_prefix[0]; k2 = s_prefix[1]; any_nan = s_nan != 0;
}

// ---- median trade size of a bar whose amounts sit in R registers per lane (any lane layout: only counts matter).
// fmk_median.h's search bisects the KEY range from [min, max] -- ten-odd rounds of R compares each before 64 candidates are left, then a
// compaction and a 21-stage cross-lane sort.  Here (the post-walk phase is most of this kernel's instructions):
//   * the wave carries a BRACKET (lo, hi] of keys around its previous bar's middle keys, as wide as it takes to catch ~100 keys
//     (the width adapts); consecutive bars of a tape have about the same size distribution, so ONE sweep of 2 R compares usually
//     proves that both middle ranks lie inside;
//   * register rounds only while more than 64 candidates are left (one or two), then the candidates go to one key per lane and the
//     bisection continues on that single register (a compare and a ballot per round) until the ranks are pinned -- no sort;
//   * a bracket miss (first bar of a wave, a jump in the size distribution) or an amount the walk flagged (NaN: np.median returns
//     NaN) takes the full range [min, max] as before.
// The result is np.median's bits in every case; the bracket only decides how much work it takes.
struct FuMed {
    uint32_t lo, hi;         // the bracket (lo, hi] in key units, wave-uniform
    uint32_t width;          // half-width beyond the middle keys it was built with
    int have;
};

__device__ __forceinline__ uint32_t fu_wave_umin(uint32_t v)
{
    return (uint32_t)fmk_dpp_reduce((int)(v ^ 0x80000000u), (int)0x7FFFFFFF, FmkOpMin()) ^ 0x80000000u;
}
__device__ __forceinline__ uint32_t fu_wave_umax(uint32_t v)
{
    return (uint32_t)fmk_dpp_reduce((int)(v ^ 0x80000000u), (int)0x80000000, FmkOpMax()) ^ 0x80000000u;
}

template <int R>
__device__ __forceinline__ double fu_median(const uint32_t (&key)[R], int cnt, int lane, bool maybe_nan, FuMed &med, uint32_t *buf)
{
    typedef MedKey<false> MK;
    const int k1 = (cnt - 1) >> 1, k2 = cnt >> 1;
    uint32_t lo = 0, hi = 0;
    int clo = 0, chi = cnt;
    bool inside = false;
    if (med.have && !maybe_nan) {
        int c1 = 0, c2 = 0;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            c1 += med_popc(key[i] <= med.lo);
            c2 += med_popc(key[i] <= med.hi);              // (hi < MAXK: the sentinels of the unused slots never count)
        }
        if (c1 <= k1 && k2 < c2) {
            inside = true; lo = med.lo; hi = med.hi; clo = c1; chi = c2;
            // about a hundred keys inside: wide enough for the next bar's middle ranks (sqrt(cnt) / 2 ~ 17 ranks of sampling noise on
            // either side), narrow enough for one register round
            const int nc = c2 - c1;
            if (nc > 144) med.width -= med.width >> 2;
            else if (nc < 80) med.width += (med.width >> 2) + 1;
        }
    }
    if (!inside) {
        uint32_t a = MK::MAXK, bmax = 0;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const uint32_t k = key[i];
            a = k < a ? k : a;
            bmax = (k != MK::MAXK && k > bmax) ? k : bmax;
        }
        const uint32_t mn = fu_wave_umin(a), mx = fu_wave_umax(bmax);
        if (mn < MK::KEY_NEG_INF || mx > MK::KEY_POS_INF) { med.have = 0; return NAN; }     // a NaN amount: np.median -> NaN
        lo = mn - 1; hi = mx; clo = 0; chi = cnt;
    }
    // invariant: count(key <= lo) = clo <= k1  and  count(key <= hi) = chi > k2
    uint32_t v1 = 0, v2 = 0;
    bool found = false;
    while (chi - clo > 64) {
        if (hi - lo == 1) { v1 = v2 = hi; found = true; break; }         // a tie of more than 64 equal keys at the middle
        const uint32_t pivot = lo + ((hi - lo) >> 1);
        int c = 0;
#pragma unroll
        for (int i = 0; i < R; ++i) c += med_popc(key[i] <= pivot);
        if (c > k2) { hi = pivot; chi = c; }
        else if (c <= k1) { lo = pivot; clo = c; }
        else {                                                           // k1 < c <= k2: the pivot separates the two middle ranks
            uint32_t a = 0, bb = MK::MAXK;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const uint32_t k = key[i];
                a = (k <= pivot && k > a) ? k : a;
                bb = (k > pivot && k < bb) ? k : bb;
            }
            v1 = fu_wave_umax(a); v2 = fu_wave_umin(bb);
            found = true;
            break;
        }
    }
    uint32_t cmin = 0, cmax = 0;
    if (!found) {
        // <= 64 candidates in (lo, hi]: one per lane
        buf[lane] = MK::MAXK;
        __builtin_amdgcn_wave_barrier();
        int base = 0;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const uint32_t k = key[i];
            const bool in = k > lo && k <= hi;
            const uint64_t m = __ballot(in);
            const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            if (in) buf[pos] = k;
            base += __popcll(m);
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t c0 = buf[lane];                                   // MAXK beyond the candidates
        __builtin_amdgcn_wave_barrier();
        if (!inside) {                                                   // (a fresh bracket takes its width from the candidates' span)
            cmin = fu_wave_umin(c0);
            cmax = fu_wave_umax(lane < chi - clo ? c0 : 0u);
        }
        // the bisection goes on over this one register (it holds every key of (lo, hi]: count(key <= pivot) = clo + the candidates <= pivot)
        for (;;) {
            if (hi - lo == 1) { v1 = v2 = hi; break; }
            const uint32_t pivot = lo + ((hi - lo) >> 1);
            const int c = clo + med_popc(c0 <= pivot);
            if (c > k2) hi = pivot;
            else if (c <= k1) lo = pivot;
            else {
                v1 = fu_wave_umax(c0 <= pivot ? c0 : 0u);
                v2 = fu_wave_umin(c0 > pivot ? c0 : MK::MAXK);
                break;
            }
        }
    }
    // the bracket for the wave's next bar
    if (!inside) {
        if (found) med.width = 64;
        else {
            const uint32_t wl = v1 - cmin, wh = cmax - v2;
            med.width = wl > wh ? wl : wh;
        }
        med.have = 1;
    }
    med.lo = v1 > med.width + 1 ? v1 - med.width - 1 : 0;
    med.hi = v2 < 0xFFFFFFFEu - med.width ? v2 + med.width : 0xFFFFFFFEu;
    return (cnt & 1) ? MK::value(v1) : (MK::value(v1) + MK::value(v2)) / 2.0;   // np.median: mean of the two middle elements
}

