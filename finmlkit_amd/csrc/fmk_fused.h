// fmk_fused.h -- cfg 4 from ONE read of the tape (round 6): build_ohlcv + build_directional_features + build_footprints
// (finmlkit/bar/base.py:132-169, 171-212, 247-300 over comp_bar_ohlcv :306-407, comp_bar_directional_features :409-546,
// comp_bar_footprints :615-752, comp_footprint_features :755-850) for the common class of bars -- float32 amounts that are whole
// multiples of a power of two, sides +-1, positive prices on the tick grid, at most FU_MAXT ticks and FU_LV (128) price levels per bar.
// Included by fmk_barflow.hip (it uses that file's FlowDirOut, redo list and tie test).
//
// Layout: LANES OWN CONSECUTIVE TICKS.  A bar of cnt ticks is one tile: lane l owns ticks [l cnt / 64, (l + 1) cnt / 64) -- R - 1 or
// R of them, R = ceil(cnt / 64) -- and reads its R prices / amounts / sides with 16-byte vector loads straight from global memory
// (tools/ownedread.py: lanes that own 8 consecutive ticks stream price + amount + side at the same 6.3 TB/s as the one-tick-per-lane
// layout of the other reducers; a lane's loads share cache lines with its neighbours', no load leaves the bar).  Everything the
// reference computes with a running state is then a plain register walk over the lane's own ticks plus ONE wave scan per bar:
//   * order flow: the running signed tick count and signed volume are INTEGERS (the volume in units of the bar's quantum 2^q:
//     exact, so the float64 sums of the reference are reproduced whatever the order), the running signed dollar sum a float64;
//     their extrema are  min over lanes of (exclusive lane prefix + the lane's local extremum);
//   * buy / sell counts, volumes and dollars follow from the totals and the signed totals (every tick of the class is signed);
//   * footprint: one 64-bit LDS atomic per tick on a 128-level x 2-side histogram indexed by (level mod 128) -- the bar's lowest level
//     need not be known while the ticks are swept -- count in the high word, units in the low word;
//   * OHLCV from the same registers, the median trade size from the amounts left in them (fmk_median.h).
// 13 B/tick are read once.  The level rows go to a staging area of FU_LV rows per bar (the CSR offsets need a scan over all bars);
// k_fu_emit turns them into the CSR rows and comp_footprint_features' per-bar values once the offsets exist.
// Bars outside the class go on lists: `dir_list` (order flow by k_bar_dir), `fp_list` (footprints by k_bar_footprints' classes),
// bars of more than FU_MAXT ticks also raise `saw_long` (OHLCV + median by the leftover passes).  float32 outputs that are a
// rounded float64 sum of inexact terms (the dollar columns, mean_spread) take the tie test of fmk_f32tie.h with THIS order's bound
// and go on k_bar_dir's redo list.
#pragma once

#define FU_MAXR 24
#define FU_MAXT (64 * FU_MAXR)
#define FU_Q_UNKNOWN 0x7FFFFFFF
#define FU_LV 128                      // price levels per bar the histogram and the staging rows hold (level mod FU_LV is the slot)

struct FuOhlcv {
    double *open, *high, *low, *close;
    float *vol;
    double *vwap;
    int64_t *trades;
    double *median;
};
struct FuStage {             // level rows of the bars the fused kernel finished: row l of bar b at [b * FU_LV + l]
    float *bv, *sv;
    int *bc, *sc;
    int *L;                  // [nb]: levels staged for the bar, -1: not staged (the bar is on fp_list)
};
struct FuLists {             // [0] = count, entries from [32]
    unsigned long long *redo;        // bar | column mask << 48 (k_bar_dir's redo list)
    unsigned long long *dir_list;    // bars whose order flow k_bar_dir computes
    unsigned long long *fp_list;     // bars whose footprint the class kernels compute
    int *saw_long;                   // a bar of more than FU_MAXT ticks was met
};

struct FuArgs {              // the kernel's output pointers, read from device memory where they are used (31 pointers as kernel
    FuOhlcv oo;              // arguments would live in SGPRs from the first instruction on and be spilled to VGPR lanes)
    FlowDirOut o;
    FuStage stg;
    FuLists li;
};

typedef double fu_d2 __attribute__((ext_vector_type(2), aligned(8)));
typedef float fu_f4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float fu_f2 __attribute__((ext_vector_type(2), aligned(4)));
typedef unsigned fu_u2 __attribute__((ext_vector_type(2), aligned(1)));
typedef unsigned fu_u1 __attribute__((aligned(1)));
typedef unsigned short fu_h1 __attribute__((aligned(1)));

// R consecutive elements from p (no element beyond p[R - 1] is touched)
template <int R>
__device__ __forceinline__ void fu_load_price(const double *__restrict__ p, double (&v)[R])
{
#pragma unroll
    for (int i = 0; i + 1 < R; i += 2) {
        const fu_d2 t = *(const fu_d2 *)(p + i);
        v[i] = t.x; v[i + 1] = t.y;
    }
    if constexpr (R & 1) v[R - 1] = p[R - 1];
}
template <int R>
__device__ __forceinline__ void fu_load_amount(const float *__restrict__ p, float (&v)[R])
{
#pragma unroll
    for (int i = 0; i + 3 < R; i += 4) {
        const fu_f4 t = *(const fu_f4 *)(p + i);
        v[i] = t.x; v[i + 1] = t.y; v[i + 2] = t.z; v[i + 3] = t.w;
    }
    constexpr int D = R & ~3;
    if constexpr ((R & 3) >= 2) {
        const fu_f2 t = *(const fu_f2 *)(p + D);
        v[D] = t.x; v[D + 1] = t.y;
    }
    if constexpr (R & 1) v[R - 1] = p[R - 1];
}
// R consecutive side bytes packed four to a word (word k = bytes 4k .. 4k + 3, missing ones 0)
template <int R>
__device__ __forceinline__ void fu_load_side(const int8_t *__restrict__ p, unsigned (&w)[(R + 3) / 4])
{
#pragma unroll
    for (int i = 0; i + 7 < R; i += 8) {
        const fu_u2 t = *(const fu_u2 *)(p + i);
        w[i / 4] = t.x; w[i / 4 + 1] = t.y;
    }
    constexpr int D8 = R & ~7;
    if constexpr ((R & 7) >= 4) w[D8 / 4] = *(const fu_u1 *)(p + D8);
    constexpr int D4 = R & ~3;
    if constexpr ((R & 3) != 0) {
        unsigned x = 0;
        if constexpr ((R & 3) >= 2) x = *(const fu_h1 *)(p + D4);
        if constexpr (R & 1) x |= (unsigned)(uint8_t)p[R - 1] << (8 * ((R - 1) & 3));
        w[D4 / 4] = x;
    }
}

// the R ticks of a lane, issued in groups of eight ticks (price 4 x 16 B, amount 2 x 16 B, side 8 B), the remainder last
template <int R>
__device__ __forceinline__ void fu_load_all(const double *__restrict__ pb, const float *__restrict__ ab, const int8_t *__restrict__ sb,
                                            double (&p)[R], float (&a)[R], unsigned (&sw)[(R + 3) / 4])
{
    constexpr int G = R / 8;
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            const fu_d2 t = *(const fu_d2 *)(pb + 8 * g + i);
            p[8 * g + i] = t.x; p[8 * g + i + 1] = t.y;
        }
#pragma unroll
        for (int i = 0; i < 8; i += 4) {
            const fu_f4 t = *(const fu_f4 *)(ab + 8 * g + i);
            a[8 * g + i] = t.x; a[8 * g + i + 1] = t.y; a[8 * g + i + 2] = t.z; a[8 * g + i + 3] = t.w;
        }
        const fu_u2 t = *(const fu_u2 *)(sb + 8 * g);
        sw[2 * g] = t.x; sw[2 * g + 1] = t.y;
        __builtin_amdgcn_sched_barrier(0);                 // (keeps the groups in this order)
    }
    constexpr int D = 8 * G, T = R - D;                    // the last T < 8 ticks
    if constexpr (T > 0) {
        double pt[T];
        float at[T];
        unsigned st[(T + 3) / 4];
        fu_load_price<T>(pb + D, pt);
        fu_load_amount<T>(ab + D, at);
        fu_load_side<T>(sb + D, st);
#pragma unroll
        for (int i = 0; i < T; ++i) { p[D + i] = pt[i]; a[D + i] = at[i]; }
#pragma unroll
        for (int i = 0; i < (T + 3) / 4; ++i) sw[D / 4 + i] = st[i];
    }
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

__device__ __forceinline__ int fu_lowbit_min(int a, int b) { return b < a ? b : a; }

// One bar of 64 (R - 1) < cnt <= 64 R ticks.  `hist`: the wave's 2 FU_LV x 8 B histogram (zero on entry, zero on return).
// wq: the quantum exponent the wave's previous bar certified with (FU_Q_UNKNOWN: none yet); updated.
template <int R>
__device__ __forceinline__ void fu_bar(const double *__restrict__ price, const float *__restrict__ amount,
                                       const int8_t *__restrict__ side, int64_t b, int64_t start, int64_t e, int cnt, int64_t n,
                                       double tick, double inv_tick, int lane, unsigned long long *hist, uint32_t *mbuf, int &wq,
                                       FuMed &med, const FuArgs *__restrict__ args, bool want_median)
{
    // (the argument block is written by the host before the launch and never by a kernel: constant address space -> scalar loads at the
    //  point of use instead of 64-lane vector loads, which is what a plain pointer gets behind the walk's LDS atomics and stores)
    typedef const FuArgs __attribute__((address_space(4))) *FuArgsC;
    const FuArgsC cargs = (FuArgsC)(uintptr_t)args;
    const auto &oo = cargs->oo;
    const auto &o = cargs->o;
    const auto &stg = cargs->stg;
    const auto &li = cargs->li;
    // ---- ownership and loads
    const unsigned off = ((unsigned)lane * (unsigned)cnt) >> 6;
    const unsigned off1 = ((unsigned)(lane + 1) * (unsigned)cnt) >> 6;
    const int len = (int)(off1 - off);                     // R - 1 or R (cnt < 64: 0 or 1)
    const double *pb = price + start + off;
    const float *ab = amount + start + off;
    const int8_t *sb = side + start + off;
    // the tick in front of the lane's first one (lane 0: the tick in front of the bar, Python's wrap-around for index -1, base.py:485-500):
    // requested FIRST -- loads return in order, and the walk's first tick needs these two
    const int64_t jprev = fmk_wrap(start + (int64_t)off - 1, n);
    double pp = price[jprev];
    int ps = side[jprev];
    __builtin_amdgcn_sched_barrier(0);
    double p[R];
    float a[R];
    unsigned sw[(R + 3) / 4];
    // issue order: eight ticks of all three columns at a time, so that the walk's first ticks wait for nine loads, not for all of them
    fu_load_all<R>(pb, ab, sb, p, a, sw);
    if (cnt == 1) ps = 0;                                  // base.py:485-488: a one-tick bar compares with 0 (whichever lane owns the tick)
    // ---- the quantum: the previous bar's, or (first bar of the wave, or after a failure) the lowest set bit of this bar's amounts
    if (wq == FU_Q_UNKNOWN) {
        // (selects, no branches per amount: zero -> "unknown", inf / NaN -> INT_MIN, which the walk flags anyway)
        int lb = FP_Q_UNKNOWN;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const uint32_t u = __float_as_uint(a[i]) & 0x7FFFFFFFu;
            const int ex = (int)(u >> 23);
            const uint32_t mant = (u & 0x7FFFFFu) | (ex != 0 ? 0x800000u : 0u);
            int l = (ex != 0 ? ex - 150 : -149) + (int)__builtin_ctz(mant | 0x80000000u);
            l = u == 0 ? (int)FP_Q_UNKNOWN : (ex == 255 ? (int)0x80000000 : l);
            l = (i < R - 1 || len == R) ? l : (int)FP_Q_UNKNOWN;
            lb = l < lb ? l : lb;
        }
        lb = fmk_dpp_reduce(lb, (int)FP_Q_UNKNOWN, FmkOpMin());
        wq = (lb == FP_Q_UNKNOWN || lb == (int)0x80000000) ? 0 : lb;     // only zeros: any quantum serves; inf / NaN: the walk flags them
        if (wq < -140) wq = -140;
        if (wq > 100) wq = 100;
    }
    const int q = wq;
    // ---- the walk over the lane's own ticks
    double hi = -INFINITY, lo = INFINITY, td = 0.0;
    double cs = 0.0, mxs = 0.0;
    double rd = 0.0, dmin = INFINITY, dmax = -INFINITY;
    int st = 0, tmin = 0x7FFFFFFF, tmax = (int)0x80000000;
    int cu = 0, umin = 0x7FFFFFFF, umax = (int)0x80000000;
    unsigned utot = 0;
    unsigned bad = 0;                                      // != 0: a side other than +-1, an amount that is not a whole number of units below 2^23
    bool offgrid = false;                                  // a price that is not within 0.49 ticks of a level (or NaN)
#pragma unroll
    for (int i = 0; i < R; ++i) {
        if (i < R - 1 || len == R) {
            const int s = (int)(int8_t)(sw[i / 4] >> (8 * (i & 3)));
            const double pi = p[i];
            const float ai = a[i];
            bad |= (unsigned)(s + 1) & ~2u;
            // spread (base.py:495-500)
            const double sp = fabs(pi - pp);
            const double spe = s != ps ? sp : 0.0;
            mxs = bf_max(mxs, spe);
            cs += spe;
            pp = pi; ps = s;
            hi = bf_max(hi, pi);
            lo = bf_min(lo, pi);
            // units of 2^q
            const float u = ldexpf(ai, -q);
            const unsigned ui = (unsigned)u;               // saturating, NaN -> 0
            bad |= (unsigned)!((float)ui == u) | (ui >> 23);
            utot += ui;
            st += s;
            tmin = st < tmin ? st : tmin; tmax = st > tmax ? st : tmax;
            cu += __mul24(s, (int)ui);
            umin = cu < umin ? cu : umin; umax = cu > umax ? cu : umax;
            // dollars: the product rounded as the reference rounds it, its sign from the side (pv >= 0 in the class)
            const double pv = pi * (double)ai;
            td += pv;
            const uint32_t pvh = ((uint32_t)__double2hiint(pv) & 0x7FFFFFFFu) | ((uint32_t)s & 0x80000000u);
            const double spv = __hiloint2double((int)pvh, __double2loint(pv));
            rd += spv;
            dmin = bf_min(dmin, rd); dmax = bf_max(dmax, rd);
            // footprint level (base.py:700-707)
            const double qq = pi * inv_tick;
            const double r = rint(qq);
            offgrid |= !(fabs(qq - r) < 0.49);
            const int lvl = (int)r;
            const unsigned key = ((unsigned)(lvl << 1) | ((unsigned)s >> 31)) & (2u * FU_LV - 1u);
#ifndef FU_EXP_NOATOMIC
            atomicAdd(&hist[key], ((unsigned long long)1 << 32) | (unsigned long long)ui);
#else
            bad |= key >> 20;
#endif
            // (folded here: left alone, the ORs and the integer min / max chains are re-associated into trees behind the walk -- a live
            //  register per tick and chain: 150 VGPRs at R = 20.  Every second tick: min3 / max3 take two ticks at a time)
            asm volatile("" : "+v"(bad));
            if ((i & 1) || i == R - 1) asm volatile("" : "+v"(tmin), "+v"(tmax), "+v"(umin), "+v"(umax));
        }
        // one tick after the other: the scheduler otherwise hoists every tick's independent arithmetic in front of the running sums'
        // dependency chains (255 VGPRs at R = 16); the other waves of the SIMD fill the chains' latency
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- fold
    const double hi_w = fmk_dpp_reduce(hi, (double)-INFINITY, FmkOpMax());
    const double lo_w = fmk_dpp_reduce(lo, (double)INFINITY, FmkOpMin());
    // (the lanes' dollar totals as an inclusive scan: its last lane is the bar's total, and lane l's value bounds the terms that stand
    //  in front of an extremum found in lane l -- the tie test below)
    const double itd = fmk_dpp_iscan(td, 0.0, FmkOpAdd());
    const double td_w = fmk_last_lane(itd);
    const double cs_w = fmk_dpp_reduce(cs, 0.0, FmkOpAdd());
    const double mxs_w = fmk_dpp_reduce(mxs, 0.0, FmkOpMax());
    // (the unit total as a float64: exact, and free of the 32-bit wrap a long bar of large units could produce)
    const double ut_w = fmk_dpp_reduce((double)utot, 0.0, FmkOpAdd());
    const bool any_bad = __ballot(bad != 0) != 0;
    const bool any_off = __ballot(offgrid) != 0;
    // exclusive prefixes of the lanes' net ticks / units / dollars
    const int it = fmk_dpp_iscan(st, 0, FmkOpAdd());
    const int iu = fmk_dpp_iscan(cu, 0, FmkOpAdd());
    const double idl = fmk_dpp_iscan(rd, 0.0, FmkOpAdd());
    const int et = it - st, eu = iu - cu;
    const double ed = fmk_dpp_shift_up1(idl, 0.0);
    const bool has = len > 0;
    const int tmin_w = fmk_dpp_reduce(has ? et + tmin : 0x7FFFFFFF, 0x7FFFFFFF, FmkOpMin());
    const int tmax_w = fmk_dpp_reduce(has ? et + tmax : (int)0x80000000, (int)0x80000000, FmkOpMax());
    const int umin_w = fmk_dpp_reduce(has ? eu + umin : 0x7FFFFFFF, 0x7FFFFFFF, FmkOpMin());
    const int umax_w = fmk_dpp_reduce(has ? eu + umax : (int)0x80000000, (int)0x80000000, FmkOpMax());
    const double dmin_l = ed + dmin, dmax_l = ed + dmax;
    const double dmin_w = fmk_dpp_reduce(dmin_l, (double)INFINITY, FmkOpMin());
    const double dmax_w = fmk_dpp_reduce(dmax_l, (double)-INFINITY, FmkOpMax());
    const int st_w = fmk_last_lane(it), cu_w = fmk_last_lane(iu);
    const double rd_w = fmk_last_lane(idl);

    // ---- comp_bar_ohlcv (base.py:306-407): volume from the exact unit total when the units are certified, else the float64 sum
    const bool units_ok = !any_bad && ut_w < 2147483648.0;
    double tv_w;
    if (units_ok) tv_w = ldexp(ut_w, q);
    else {
        double tv = 0.0;
#pragma unroll
        for (int i = 0; i < R; ++i)
            if (i < R - 1 || len == R) tv += (double)a[i];
        tv_w = fmk_dpp_reduce(tv, 0.0, FmkOpAdd());
    }
    const double first = price[start];
    if (lane == 0) {
        oo.open[b] = first;
        oo.close[b] = price[e];
        oo.high[b] = first != first ? first : hi_w;          // a NaN first price never loses (base.py:371-382)
        oo.low[b] = first != first ? first : lo_w;
        oo.vol[b] = (float)tv_w;
        oo.vwap[b] = tv_w > 0.0 ? td_w / tv_w : 0.0;          // base.py:398
        oo.trades[b] = cnt;
    }
    // ---- comp_bar_directional_features (base.py:409-546)
    const bool dir_ok = units_ok && lo_w > 0.0 && first == first && td_w < INFINITY;      // (a NaN / inf price anywhere: not this class)
    if (!dir_ok) {
        if (lane == 0) li.dir_list[32 + atomicAdd(li.dir_list, 1ULL)] = (unsigned long long)b;
    } else {
        // every tick is a buy or a sell: counts and volumes from the totals and the signed totals (integers: exact)
        const int tb = (cnt + st_w) >> 1, tsell = (cnt - st_w) >> 1;
        const double vb = ldexp(0.5 * (ut_w + (double)cu_w), q), vs = ldexp(0.5 * (ut_w - (double)cu_w), q);
        // ... the dollar sums likewise, but these are float64 sums of rounded products: db = (td + rd) / 2 differs from the reference's
        // tick-order sum of the buy terms by the rounding noise of both orders -- the tie test below knows
        const double db = 0.5 * (td_w + rd_w), ds = 0.5 * (td_w - rd_w);
        const double mean = cs_w / (double)cnt;
        // Bounds (u = 2^-53, A = the sum of the terms' magnitudes = td, every term >= 0 in the class):
        //   * db / ds.  Reference: the recursive sum of len terms of one sign, (len - 1) u S.  Here: td and rd each through <= R additions
        //     in a lane and a scan tree of 6 levels whose nodes are disjoint ranges, then one addition and a halving: (R + 8) u A.
        //   * extrema of the running signed sum, for an extremum found in lane l after k = (ticks up to the end of lane l) ticks:
        //     reference k u M (M = the largest magnitude the running sum takes); here the lane's own running sum (<= R additions of values
        //     below A_l, the terms up to lane l) + the exclusive prefix (lane totals through the scan: (R + 6) u A_l) + one addition.
        //     The bound belongs to the POSITION: an early extremum of small magnitude -- fine float32 spacing -- has few terms in front of
        //     it (with the bar's totals instead, 0.2 % of the bench's bars went to the redo; every lane evaluates the test on its own
        //     candidate, which costs what the wave-uniform test cost).
        const double len_d = (double)(cnt + 1);
        const double uA = 1.17e-16 * (double)(2 * R + 16) * td_w;
        const double md = fmax(fabs(dmin_w), fabs(dmax_w));
        const double b_l = 1.17e-16 * ((double)(off1 + 1) * md + (double)(2 * R + 16) * itd);
        unsigned mask = 0;
        if (fmk_near_f32_tie(db, 1.17e-16 * len_d * db + uA)) mask |= 1u << 2;
        if (fmk_near_f32_tie(ds, 1.17e-16 * len_d * ds + uA)) mask |= 1u << 3;
        {
            const bool tmn = has && dmin_l == dmin_w && fmk_near_f32_tie(dmin_l, b_l);
            const bool tmx = has && dmax_l == dmax_w && fmk_near_f32_tie(dmax_l, b_l);
            if (__ballot(tmn || tmx) != 0) mask |= 1u << 6;
        }
        if (fmk_near_f32_tie(mean, (1.17e-16 * (len_d + (double)(R + 8)) + 1.2e-16) * fabs(mean))) mask |= 1u << 4;
        if (bf_force_redo != 0) mask = 0x7F;
        if (lane == 0) {
            if (mask) li.redo[32 + atomicAdd(li.redo, 1ULL)] = (unsigned long long)b | ((unsigned long long)mask << 48);
            o.ticks_buy[b] = tb; o.ticks_sell[b] = tsell;
            o.volume_buy[b] = (float)vb; o.volume_sell[b] = (float)vs;
            o.dollars_buy[b] = (float)db; o.dollars_sell[b] = (float)ds;
            o.max_spread[b] = (float)mxs_w;
            o.mean_spread[b] = (float)mean;
            o.cum_ticks_min[b] = tmin_w; o.cum_ticks_max[b] = tmax_w;
            o.cum_volumes_min[b] = (float)ldexp((double)umin_w, q); o.cum_volumes_max[b] = (float)ldexp((double)umax_w, q);
            o.cum_dollars_min[b] = (float)dmin_w; o.cum_dollars_max[b] = (float)dmax_w;
        }
    }
    // ---- footprint rows -> staging (levels low .. high, at most FU_LV; the histogram slot of a level is level mod FU_LV)
    {
        const double qlo = lo_w * inv_tick, qhi = hi_w * inv_tick;
        const int lowl = (int)rint(qlo), highl = (int)rint(qhi);
        const int L = highl - lowl + 1;
        // lane l reads levels l and l + 64, and clears slot pairs l and l + 64 (every slot)
        const unsigned slot0 = (unsigned)(lowl + lane) & (FU_LV - 1u), slot1 = (unsigned)(lowl + lane + 64) & (FU_LV - 1u);
        const unsigned long long hb0 = hist[2 * slot0], hs0 = hist[2 * slot0 + 1];
        const unsigned long long hb1 = hist[2 * slot1], hs1 = hist[2 * slot1 + 1];
        {
            unsigned long long z = 0ULL;
            asm volatile("" : "+v"(z));                         // (a zero made HERE: as a loop invariant it lives in four registers through every walk)
            hist[2 * lane] = z; hist[2 * lane + 1] = z;
            hist[2 * lane + 128] = z; hist[2 * lane + 129] = z;
        }
        const unsigned ub0 = (unsigned)hb0, us0 = (unsigned)hs0, ub1 = (unsigned)hb1, us1 = (unsigned)hs1;
        // every (level, side) total below 2^24 units: each float32 add of the reference is exact whatever its order (fp_certified_units_per_key)
        const bool big = __ballot((lane < L && ((ub0 | us0) >> 24) != 0) || (lane + 64 < L && ((ub1 | us1) >> 24) != 0)) != 0;
        const bool fp_ok = dir_ok && !any_off && L >= 1 && L <= FU_LV && !big && fabs(qlo) < 1e9 && fabs(qhi) < 1e9;
        if (fp_ok) {
            const int64_t at = b * FU_LV + lane;
            if (lane < L) {
                stg.bv[at] = ldexpf((float)ub0, q);
                stg.sv[at] = ldexpf((float)us0, q);
                stg.bc[at] = (int)(hb0 >> 32);
                stg.sc[at] = (int)(hs0 >> 32);
            }
            if (lane + 64 < L) {
                stg.bv[at + 64] = ldexpf((float)ub1, q);
                stg.sv[at + 64] = ldexpf((float)us1, q);
                stg.bc[at + 64] = (int)(hb1 >> 32);
                stg.sc[at + 64] = (int)(hs1 >> 32);
            }
            if (lane == 0) stg.L[b] = L;
        } else if (lane == 0) {
            stg.L[b] = -1;
            li.fp_list[32 + atomicAdd(li.fp_list, 1ULL)] = (unsigned long long)b;
        }
        __builtin_amdgcn_wave_barrier();
    }
    // a bar that did not certify: the next bar of the wave measures its own quantum
    if (!units_ok) wq = FU_Q_UNKNOWN;
    // ---- median trade size (base.py:401-404) from the amounts in the registers
#ifdef FU_EXP_NOMED
    want_median = false;
#endif
    if (want_median) {
        typedef MedKey<false> MK;
        uint32_t key[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            // (opaque: otherwise the |amount| this needs is computed once with the quantum pre-pass's, in FRONT of the walk, and a
            //  second copy of every amount lives through it)
            uint32_t u = __float_as_uint(a[i]);
            asm volatile("" : "+v"(u));
            const uint32_t k = MK::tokey(u);
            key[i] = (i < R - 1 || len == R) ? k : MK::MAXK;
        }
        const double m = fu_median<R>(key, cnt, lane, any_bad, med, mbuf);
        if (lane == 0) oo.median[b] = m;
    }
}

template <bool MEDIAN>
#ifndef FU_WAVES
#define FU_WAVES 4
#endif
__global__ __launch_bounds__(256, FU_WAVES) void k_fu_bars(const double *__restrict__ price, const float *__restrict__ amount,
                                                    const int8_t *__restrict__ side, const int64_t *__restrict__ ci, int64_t nb,
                                                    int64_t n, double tick, const FuArgs *__restrict__ args)
{
    __shared__ unsigned long long s_hist[4][2 * FU_LV];
    __shared__ uint32_t s_buf[4][64];
    const int lane = fmk_lane();
    const int wib = fmk_uniform((int)(threadIdx.x >> 6));
    unsigned long long *hist = s_hist[wib];
    hist[2 * lane] = 0ULL; hist[2 * lane + 1] = 0ULL; hist[2 * lane + 128] = 0ULL; hist[2 * lane + 129] = 0ULL;
    __builtin_amdgcn_wave_barrier();
    const double inv_tick = 1.0 / tick;
    int wq = FU_Q_UNKNOWN;
    FuMed med{0u, 0u, 0u, 0};
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wib;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t b = wave0; b < nb; b += nwaves) {
        const int64_t s = fmk_uniform(ci[b]);
        const int64_t e = fmk_uniform(ci[b + 1]);
        const int64_t cnt64 = e - s;
        if (cnt64 > FU_MAXT || cnt64 <= 0) {
            if (lane == 0) {
                const FuOhlcv &oo = args->oo;
                const FuLists &li = args->li;
                if (cnt64 <= 0) {                                  // empty bar: previous close (base.py:352-361)
                    const double pz = price[fmk_wrap(e, n)];
                    oo.open[b] = pz; oo.high[b] = pz; oo.low[b] = pz; oo.close[b] = pz;
                    oo.vol[b] = 0.f; oo.vwap[b] = 0.0; oo.trades[b] = 0;
                    if (MEDIAN) oo.median[b] = 0.0;
                } else if (__hip_atomic_load(li.saw_long, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                    __hip_atomic_store(li.saw_long, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                li.dir_list[32 + atomicAdd(li.dir_list, 1ULL)] = (unsigned long long)b;
                args->stg.L[b] = -1;
                li.fp_list[32 + atomicAdd(li.fp_list, 1ULL)] = (unsigned long long)b;
            }
            continue;
        }
        const int cnt = (int)cnt64;
        const int64_t start = s + 1;
#define FU_CASE(RR) case RR: fu_bar<RR>(price, amount, side, b, start, e, cnt, n, tick, inv_tick, lane, hist, s_buf[wib], wq, med, args, MEDIAN); break;
        switch ((cnt + 63) >> 6) {
            FU_CASE(1) FU_CASE(2) FU_CASE(3) FU_CASE(4) FU_CASE(5) FU_CASE(6) FU_CASE(7) FU_CASE(8)
            FU_CASE(9) FU_CASE(10) FU_CASE(11) FU_CASE(12) FU_CASE(13) FU_CASE(14) FU_CASE(15) FU_CASE(16)
            FU_CASE(17) FU_CASE(18) FU_CASE(19) FU_CASE(20) FU_CASE(21) FU_CASE(22) FU_CASE(23)
            default: fu_bar<24>(price, amount, side, b, start, e, cnt, n, tick, inv_tick, lane, hist, s_buf[wib], wq, med, args, MEDIAN); break;
        }
#undef FU_CASE
    }
}

// The staged level rows -> CSR rows + comp_footprint_features (base.py:755-850): one wave per bar, the rows through the wave's slice
// of LDS in the layout fp_emit_bar reads (vol[2 l] / vol[2 l + 1], cnt[...], a 128-level class).
__global__ __launch_bounds__(256) void k_fu_emit(FuStage stg, const double *__restrict__ lows, double tick, double imb_mult,
                                                 const int64_t *__restrict__ off, int64_t nb, FpOut o)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_mem[4][FU_LV * 24 + 256];
    const int lane = fmk_lane();
    const int wib = fmk_uniform((int)(threadIdx.x >> 6));
    unsigned char *mine = s_mem[wib];
    float *vol = (float *)mine;
    int *cnt = (int *)(mine + FU_LV * 8);
    float *aux = (float *)(mine + FU_LV * 16);
    int *stk = (int *)(mine + FU_LV * 24);
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wib;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t b = wave0; b < nb; b += nwaves) {
        const int L = fmk_uniform(stg.L[b]);
        if (L < 0) continue;
        const int64_t base = fmk_uniform(off[b]);
        const int64_t low = fp_level(lows[b], tick);
        for (int l = lane; l < L; l += 64) {
            const int64_t at = b * FU_LV + l;
            vol[2 * l] = stg.bv[at]; vol[2 * l + 1] = stg.sv[at];
            cnt[2 * l] = stg.bc[at]; cnt[2 * l + 1] = stg.sc[at];
        }
        __builtin_amdgcn_wave_barrier();
        fp_emit_bar(o, b, base, L, low, FU_LV, imb_mult, lane, vol, cnt, aux, stk);
    }
}
