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
#define FU_LONGEST 262144             // a longer bar is not walked by one wave: open .. trades, order flow and footprint by the other kernels
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
    unsigned long long *redo;        // bar | column mask << 48: float32 ties of bars of one tile (k_bar_dir_redo: a wave per bar)
    unsigned long long *redo_long;   // ... of longer bars (k_bar_dir's own redo list: the chunk-record kernel)
    unsigned long long *dir_list;    // bars whose order flow k_bar_dir computes
    unsigned long long *fp_list;     // bars whose footprint the class kernels compute
    int *saw_long;                   // a bar of more than FU_MAXT ticks was met (its median: fmk_median_launch)
    int *saw_huge;                   // a bar of more than FU_LONGEST ticks was met (its open .. trades: the leftover pass)
};

struct FuArgs {              // the kernel's output pointers, read from device memory where they are used (31 pointers as kernel
    const void *amount;      // arguments would live in SGPRs from the first instruction on and be spilled to VGPR lanes)
    FuOhlcv oo;
    FlowDirOut o;
    FuStage stg;
    FuLists li;
    unsigned long long *scrap;       // 8 bytes per lane of the launch (FuPend::scrap)
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

// (the carried-bracket median of a bar whose keys sit in registers -- FuMed, fu_median -- lives in fmk_median.h: k_bar_median_small uses it too)

// ---- float64 wave reductions for this kernel's fold (most of its instructions are behind the walk).  fmk_dpp_reduce is a scan: every
// step sets up an identity for the lanes without a source (two moves), moves both halves and combines -- five to seven instructions
// (fmax / fmin add a canonicalising v_max x, x): ~40 per reduction.  Here: ROTATIONS inside each row of 16 lanes (every lane has a
// source: no identity), then row_bcast:15 into rows 1, 3 and row_bcast:31 into row 3, whose lanes then hold the result; the rows a
// broadcast leaves alone keep whatever the destination held (a value that cannot hurt: nothing reads them afterwards).  Three
// instructions per step, 20 per reduction.  The sum's order is fixed (the rotation tree seen from lane 63): deterministic.
#define FU_DPP_ROW_ROR(n) (0x120 + (n))
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double fu_dppmov(double old, double v)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
#define FU_REDUCE(NAME, OP)                                                                       \
    __device__ __forceinline__ double NAME(double v)                                              \
    {                                                                                             \
        double t = fu_dppmov<FU_DPP_ROW_ROR(1), 0xF>(v, v); v = OP(v, t);                         \
        t = fu_dppmov<FU_DPP_ROW_ROR(2), 0xF>(t, v); v = OP(v, t);                                \
        t = fu_dppmov<FU_DPP_ROW_ROR(4), 0xF>(t, v); v = OP(v, t);                                \
        t = fu_dppmov<FU_DPP_ROW_ROR(8), 0xF>(t, v); v = OP(v, t);                                \
        t = fu_dppmov<FMK_DPP_ROW_BCAST15, 0xA>(t, v); v = OP(v, t);                              \
        t = fu_dppmov<FMK_DPP_ROW_BCAST31, 0xC>(t, v); v = OP(v, t);                              \
        return fmk_last_lane(v);                                                                  \
    }
__device__ __forceinline__ double fu_op_add(double a, double b) { return a + b; }
FU_REDUCE(fu_rmax, bf_max)
FU_REDUCE(fu_rmin, bf_min)
FU_REDUCE(fu_rsum, fu_op_add)
#undef FU_REDUCE

__device__ __forceinline__ int fu_lowbit_min(int a, int b) { return b < a ? b : a; }

// ---- the walk over one TILE of a bar: tcnt <= 64 R ticks from global index j0, lane l owning ticks [l tcnt / 64, (l + 1) tcnt / 64)
// of it -- R - 1 or R of them.  UNITS: the certified class (integer volume in units of 2^q, footprint histogram); otherwise the
// volume sums are float64 (exact in any order for float32 amounts, the assumption of every reducer here) and no level is computed.
// ---- deferred outputs.  The 22 per-bar values (FuOhlcv's 8 columns, FlowDirOut's 14) are one-lane stores, and on gfx950 stores count
// on the same counter as loads, in order: the next bar's "wait for my first loads" also waited for these stores to be acknowledged
// (0.6 ms of the kernel's 4.0 on the bench tape).  So a bar's values wait in the wave's LDS slots and are stored right BEHIND the
// next bar's load requests -- by 22 lanes at once, each with the column pointer it fetched when the kernel began -- and have a
// whole bar's time to complete.  slot k (8 B) = column k; [22] = the bar, [23] = which columns are valid.
#ifndef FU_DEFER_UNITS
#define FU_DEFER_UNITS 0
#endif
#define FU_NCOL 23                   // ... and (UNITS) the staged level count of the bar, an int32 column: slot [27]
#define FU_COL_IS64 0x303EFu         // open high low close . vwap trades median | ticks_buy ticks_sell . . . . . . cum_ticks_min cum_ticks_max . . . .
struct FuPend {
    unsigned long long *slot;        // the wave's 28 slots in LDS (null: store at once)
    unsigned long long col;          // lane k < 22: the base pointer of column k
    unsigned long long scrap;        // the lane's own 8 bytes of a scrap area: where a lane that has nothing to store writes
    unsigned long long *hist;        // UNITS: the histogram the pending bar was swept into (the wave has two and alternates): its level
                                     // rows go to the staging area in the same flush -- [24] = levels (0: nothing staged), [25] = lowest level, [26] = q
};
// NO BRANCH in here: the compiler's wait-count pass is conservative at every basic-block boundary -- a flush with `if (lane < L)`
// around its stores got an s_waitcnt vmcnt(0) in front of every block, i.e. it waited for the bar's loads, issued four stores, and
// then WAITED FOR THE STORES (measured: 4.0 -> 5.3 ms).  A lane that has nothing to store writes its own 8 bytes of a scrap area.
template <bool UNITS>
__device__ __forceinline__ void fu_flush(const FuPend &pd, int lane, const FuArgs *__restrict__ args)
{
    const unsigned valid = (unsigned)fmk_uniform((int)(unsigned)pd.slot[23]);
    const unsigned long long b = pd.slot[22];
    const bool vk = lane < FU_NCOL && ((valid >> lane) & 1u);
    const bool k64 = (FU_COL_IS64 >> lane) & 1u;
    const unsigned long long v = pd.slot[lane < 22 ? lane : 27];
    *(unsigned long long *)((vk && k64) ? pd.col + 8 * b : pd.scrap) = v;
    *(unsigned *)((vk && !k64) ? pd.col + 4 * b : pd.scrap) = (unsigned)v;
    if constexpr (UNITS) {
        typedef const FuArgs __attribute__((address_space(4))) *FuArgsC;
        const auto &stg = ((FuArgsC)(uintptr_t)args)->stg;
        const int L = valid ? (int)(unsigned)pd.slot[24] : 0;
        const int lowl = (int)(unsigned)pd.slot[25], q = (int)(unsigned)pd.slot[26];
        const unsigned slot0 = (unsigned)(lowl + lane) & (FU_LV - 1u), slot1 = (unsigned)(lowl + lane + 64) & (FU_LV - 1u);
        const unsigned long long hb0 = pd.hist[2 * slot0], hs0 = pd.hist[2 * slot0 + 1];
        const unsigned long long hb1 = pd.hist[2 * slot1], hs1 = pd.hist[2 * slot1 + 1];
        const int64_t at = (int64_t)b * FU_LV + lane;
        const bool p0 = lane < L, p1 = lane + 64 < L;
        float *const sc_f = (float *)pd.scrap;
        int *const sc_i = (int *)pd.scrap;
        *(p0 ? stg.bv + at : sc_f) = ldexpf((float)(unsigned)hb0, q);
        *(p0 ? stg.sv + at : sc_f) = ldexpf((float)(unsigned)hs0, q);
        *(p0 ? stg.bc + at : sc_i) = (int)(hb0 >> 32);
        *(p0 ? stg.sc + at : sc_i) = (int)(hs0 >> 32);
        *(p1 ? stg.bv + at + 64 : sc_f) = ldexpf((float)(unsigned)hb1, q);
        *(p1 ? stg.sv + at + 64 : sc_f) = ldexpf((float)(unsigned)hs1, q);
        *(p1 ? stg.bc + at + 64 : sc_i) = (int)(hb1 >> 32);
        *(p1 ? stg.sc + at + 64 : sc_i) = (int)(hs1 >> 32);
        // (the pending bar's histogram is clean again: lane l clears slot pairs l and l + 64, every slot)
        __builtin_amdgcn_wave_barrier();
        unsigned long long z = 0ULL;
        asm volatile("" : "+v"(z));
        pd.hist[2 * lane] = z; pd.hist[2 * lane + 1] = z;
        pd.hist[2 * lane + 128] = z; pd.hist[2 * lane + 129] = z;
    }
    __builtin_amdgcn_wave_barrier();
    pd.slot[23] = 0ULL;
}

// developer build -DFU_TIMING: shader cycles every wave spends in the phases of a one-tile bar -- [0] from the first load to the end of the
// walk, [2] from there to the end of the bar; [3] / [1]: sums of the clock at the start of the median / at the end of the bar (their
// difference is the time in the median) -- summed over the waves (fmk_diag_fused_phases)
#ifdef FU_TIMING
__device__ unsigned long long fu_phase_cycles[4] = {0, 0, 0, 0};
#define FU_T(slot) do { const unsigned long long now_ = __builtin_readcyclecounter(); if (fmk_lane() == 0) atomicAdd(&fu_phase_cycles[slot], now_ - t_last_); t_last_ = now_; } while (0)
#define FU_TEND() do { if (fmk_lane() == 0) atomicAdd(&fu_phase_cycles[1], t_last_); } while (0)
#else
#define FU_T(slot) do { } while (0)
#define FU_TEND() do { } while (0)
#endif

struct FuTile {                // what the walk leaves in each lane: tile-local running sums (from 0 at the lane's first tick) and their extrema
    int st, tmin, tmax;
    int cu, umin, umax;        // UNITS
    double rv, vmin, vmax;     // !UNITS
    double rd, dmin, dmax;
    double td;                 // the lane's sum of price x amount
    unsigned utot;             // UNITS: the lane's units (< 2^28)
    double tv;                 // !UNITS: the lane's sum of amounts
    int len;
    unsigned off1;             // ticks of the tile up to the end of the lane
    double first, last;        // wave-uniform: the prices of the tile's first and last tick (not an OVERLAP tile)
};
struct FuBar {                 // accumulated over the tiles of a bar
    double hi, lo, cs, mxs;    // per lane
    double ut, tv;             // per lane: units (UNITS) / amounts (!UNITS) as float64 sums
    unsigned bad;              // per lane: != 0: a side other than +-1, an amount that is not a whole number of units below 2^23 (UNITS) / negative (!UNITS)
    bool offgrid;              // a price not within 0.49 ticks of a level, or NaN (UNITS)
    int tminA, tmaxA, uminA, umaxA;      // per lane: candidates for the bar's extrema (carry + lane prefix + local extremum)
    double vminA, vmaxA, dminA, dmaxA;
    double aminA, amaxA;       // the sum of |terms| in front of (and inside) the lane that holds dminA / dmaxA, and the tick count: the tie bound
    int kmin, kmax;
    int ct, cu;                // wave-uniform carries: the running sums at the end of the tiles done
    double cv, cd, cA;
    int done;                  // ticks of the tiles done
    int tiles;
};

// OVERLAP (R = FU_MAXR, tcnt = FU_MAXT only): the last tile of a bar of several tiles is a WHOLE tile that ends with the bar, so no
// load leaves the bar; its first `ov` ticks belong to the tile in front of it and are skipped (one body for every remainder length
// instead of a second set of exact-size ones).
template <int R, bool UNITS, bool OVERLAP = false>
__device__ __forceinline__ void fu_walk(const double *__restrict__ price, const float *__restrict__ amount, const int8_t *__restrict__ side,
                                        int64_t j0, int tcnt, bool one_tick_bar, int64_t n, int q, double inv_tick, int lane,
                                        unsigned long long *hist, FuBar &B, FuTile &T, float (&keep)[R], int ov = 0, const FuPend *pend = nullptr,
                                        const FuArgs *__restrict__ pargs = nullptr)
{
    const unsigned off = ((unsigned)lane * (unsigned)tcnt) >> 6;
    unsigned off1 = ((unsigned)(lane + 1) * (unsigned)tcnt) >> 6;
    int len = (int)(off1 - off);                           // R - 1 or R (tcnt < 64: 0 or 1)
    const int vfrom = OVERLAP ? ov - (int)off : 0;         // OVERLAP: the lane's ticks from position vfrom on are this tile's
    const double *pb = price + j0 + off;
    const float *ab = amount + j0 + off;
    const int8_t *sb = side + j0 + off;
    // the tick in front of the lane's first one (the bar's first lane: the tick in front of the bar, Python's wrap-around for index -1,
    // base.py:485-500): requested FIRST -- loads return in order, and the walk's first tick needs these two
    const int64_t jprev = fmk_wrap(j0 + (int64_t)off + (OVERLAP && vfrom > 0 ? vfrom : 0) - 1, n);
    double pp = price[jprev];
    int ps = side[jprev];
    __builtin_amdgcn_sched_barrier(0);
    double p[R];
    float a[R];
    unsigned sw[(R + 3) / 4];
    // issue order: eight ticks of all three columns at a time, so that the walk's first ticks wait for nine loads, not for all of them
    fu_load_all<R>(pb, ab, sb, p, a, sw);
    if (one_tick_bar) ps = 0;                              // base.py:485-488: a one-tick bar compares with 0 (whichever lane owns the tick)
    double hi = B.hi, lo = B.lo, cs = B.cs, mxs = B.mxs;
    double td = 0.0, tv = 0.0;
    double rd = 0.0, dmin = INFINITY, dmax = -INFINITY;
    double rv = 0.0, vmin = INFINITY, vmax = -INFINITY;
    int st = 0, tmin = 0x7FFFFFFF, tmax = (int)0x80000000;
    int cu = 0, umin = 0x7FFFFFFF, umax = (int)0x80000000;
    unsigned utot = 0;
    unsigned bad = B.bad;
    bool offgrid = B.offgrid;
    if constexpr (!OVERLAP) {
        // the tile's first tick: the first tick of the first lane that owns one (every lane has loaded its p[0]: all of them take part)
        const int fl = fmk_uniform((int)__builtin_ctzll(__ballot(len > 0)));
        T.first = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(p[0]), fl), __builtin_amdgcn_readlane(__double2loint(p[0]), fl));
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
        // (OVERLAP: no branch per tick -- with one, the running sums' updates are sunk behind the walk and every tick's terms stay live;
        //  a tick of the overlap takes part as a zero-sized trade at the previous tick's price and side)
        if (OVERLAP || i < R - 1 || len == R) {
            const bool valid = !OVERLAP || i >= vfrom;
            int s = (int)(int8_t)(sw[i / 4] >> (8 * (i & 3)));
            double pi = p[i];
            float ai = a[i];
            // (the tick's inputs through an opaque statement: everything the tick computes comes BEHIND the previous tick's -- see the end
            //  of the loop body)
            asm volatile("" : "+v"(pi), "+v"(ai));
            if constexpr (OVERLAP) { s = valid ? s : ps; pi = valid ? pi : pp; ai = valid ? ai : 0.f; }
            const int sv = OVERLAP ? (valid ? s : 0) : s;         // the tick's step of the running sums
            bad |= (unsigned)(s + 1) & ~2u;
            // spread (base.py:495-500)
            const double sp = fabs(pi - pp);
            const double spe = s != ps ? sp : 0.0;
            mxs = bf_max(mxs, spe);
            cs += spe;
            pp = pi; ps = s;
            hi = bf_max(hi, pi);
            lo = bf_min(lo, pi);
            st += sv;
            tmin = st < tmin ? st : tmin; tmax = st > tmax ? st : tmax;
            unsigned ui = 0;
            const double av = (double)ai;
            if constexpr (UNITS) {
                // units of 2^q
                const float u = ldexpf(ai, -q);
                ui = (unsigned)u;                            // saturating, NaN -> 0
                bad |= (unsigned)!((float)ui == u) | (ui >> 23);
                utot += ui;
                cu += __mul24(sv, (int)ui);
                umin = cu < umin ? cu : umin; umax = cu > umax ? cu : umax;
            } else {
                bad |= __float_as_uint(ai) >> 31;           // a negative amount: the sign trick below needs price x amount >= 0
                tv += av;
                const uint32_t avh = ((uint32_t)__double2hiint(av) & 0x7FFFFFFFu) | ((uint32_t)s & 0x80000000u);
                rv += __hiloint2double((int)avh, __double2loint(av));
                vmin = bf_min(vmin, rv); vmax = bf_max(vmax, rv);
            }
            // dollars: the product rounded as the reference rounds it, its sign from the side (pv >= 0 in the class)
            const double pv = pi * av;
            td += pv;
            const uint32_t pvh = ((uint32_t)__double2hiint(pv) & 0x7FFFFFFFu) | ((uint32_t)s & 0x80000000u);
            const double spv = __hiloint2double((int)pvh, __double2loint(pv));
            rd += spv;
            dmin = bf_min(dmin, rd); dmax = bf_max(dmax, rd);
            if constexpr (UNITS) {
                // footprint level (base.py:700-707)
                const double qq = pi * inv_tick;
                const double r = rint(qq);
                offgrid |= !(fabs(qq - r) < 0.49);
                const int lvl = (int)r;
                const unsigned key = ((unsigned)(lvl << 1) | ((unsigned)s >> 31)) & (2u * FU_LV - 1u);
                atomicAdd(&hist[key], ((unsigned long long)(valid ? 1u : 0u) << 32) | (unsigned long long)ui);
            }
            // (Every running value is pinned here.  Left alone, instruction selection orders the unrolled ticks' arithmetic freely -- only
            //  memory operations and volatile statements keep their order: the ORs and the integer min / max chains were re-associated
            //  into trees behind the walk, whole float64 sum chains were sunk behind it with their terms spilled, products of later
            //  ticks were hoisted: 150 .. 255 VGPRs at R = 20.  The integer extrema every second tick: min3 / max3 take two ticks.)
            asm volatile("" : "+v"(bad), "+v"(st), "+v"(cs), "+v"(td), "+v"(rd), "+v"(mxs), "+v"(hi), "+v"(lo), "+v"(dmin), "+v"(dmax));
            if constexpr (UNITS) asm volatile("" : "+v"(cu));
            else asm volatile("" : "+v"(tv), "+v"(rv), "+v"(vmin), "+v"(vmax));
            if ((i & 1) || i == R - 1) {
                asm volatile("" : "+v"(tmin), "+v"(tmax));
                if constexpr (UNITS) asm volatile("" : "+v"(umin), "+v"(umax), "+v"(utot));
            }
        }
        // one tick after the other: the scheduler otherwise hoists every tick's independent arithmetic in front of the running sums'
        // dependency chains (255 VGPRs at R = 16); the other waves of the SIMD fill the chains' latency
        __builtin_amdgcn_sched_barrier(0);
    }
    // The wave's previous bar leaves HERE, behind this bar's last load wait.  (gfx950 counts loads and stores on one counter and they
    // complete out of order with respect to each other, so while both kinds are pending the compiler can only wait for ALL of
    // them: stores issued in front of a load wait are waited for -- their full round trip -- with the loads.  From here they have this
    // bar's fold, outputs and median to complete before the next bar asks for its ticks; the open and close prices come from the
    // registers, so nothing behind this point loads from global memory.)
    if (pend) fu_flush<UNITS>(*pend, lane, pargs);
    if constexpr (!OVERLAP) {
        // (the tile's last tick is lane 63's last: lane 63 always owns R ticks)
        T.last = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(p[R - 1]), 63), __builtin_amdgcn_readlane(__double2loint(p[R - 1]), 63));
    }
    if constexpr (OVERLAP) {
        len = R - vfrom; len = len < 0 ? 0 : (len > R ? R : len);
        off1 = (int)off + R > ov ? off + R - (unsigned)ov : 0u;
    }
    B.hi = hi; B.lo = lo; B.cs = cs; B.mxs = mxs; B.bad = bad; B.offgrid = offgrid;
    T.st = st; T.tmin = tmin; T.tmax = tmax; T.cu = cu; T.umin = umin; T.umax = umax;
    T.rv = rv; T.vmin = vmin; T.vmax = vmax; T.rd = rd; T.dmin = dmin; T.dmax = dmax;
    T.td = td; T.utot = utot; T.tv = tv; T.len = len; T.off1 = off1;
    // the amounts stay for the median (a bar of one tile); slots beyond the lane's ticks are marked in fu_finish by `len`
    if constexpr (!OVERLAP) {
#pragma unroll
        for (int i = 0; i < R; ++i) keep[i] = a[i];
    }
}

// a tile's lane results -> the bar's candidates and carries (ONE set of wave scans per tile)
template <bool UNITS>
__device__ __forceinline__ void fu_fold(FuBar &B, const FuTile &T, int tcnt)
{
    const bool has = T.len > 0;
    const int it = fmk_dpp_iscan(T.st, 0, FmkOpAdd());
    const int et = B.ct + (it - T.st);
    const int tmn = has ? et + T.tmin : 0x7FFFFFFF, tmx = has ? et + T.tmax : (int)0x80000000;
    B.tminA = tmn < B.tminA ? tmn : B.tminA; B.tmaxA = tmx > B.tmaxA ? tmx : B.tmaxA;
    B.ct += fmk_last_lane(it);
    if constexpr (UNITS) {
        const int iu = fmk_dpp_iscan(T.cu, 0, FmkOpAdd());
        const int eu = B.cu + (iu - T.cu);
        const int umn = has ? eu + T.umin : 0x7FFFFFFF, umx = has ? eu + T.umax : (int)0x80000000;
        B.uminA = umn < B.uminA ? umn : B.uminA; B.umaxA = umx > B.umaxA ? umx : B.umaxA;
        B.cu += fmk_last_lane(iu);
        B.ut += (double)T.utot;
    } else {
        const double iv = fmk_dpp_iscan(T.rv, 0.0, FmkOpAdd());
        const double ev = B.cv + fmk_dpp_shift_up1(iv, 0.0);
        B.vminA = bf_min(B.vminA, ev + T.vmin); B.vmaxA = bf_max(B.vmaxA, ev + T.vmax);      // (a lane without ticks: +-inf)
        B.cv += fmk_last_lane(iv);
        B.tv += T.tv;
    }
    const double idl = fmk_dpp_iscan(T.rd, 0.0, FmkOpAdd());
    const double itd = fmk_dpp_iscan(T.td, 0.0, FmkOpAdd());
    const double ed = B.cd + fmk_dpp_shift_up1(idl, 0.0);
    const double dmn = ed + T.dmin, dmx = ed + T.dmax;
    const double ain = B.cA + itd;
    const int kin = B.done + (int)T.off1;
    if (dmn < B.dminA) { B.dminA = dmn; B.aminA = ain; B.kmin = kin; }
    if (dmx > B.dmaxA) { B.dmaxA = dmx; B.amaxA = ain; B.kmax = kin; }
    B.cd += fmk_last_lane(idl);
    B.cA += fmk_last_lane(itd);
    B.done += tcnt;
    B.tiles += 1;
}

__device__ __forceinline__ void fu_bar_init(FuBar &B)
{
    B.hi = -INFINITY; B.lo = INFINITY; B.cs = 0.0; B.mxs = 0.0; B.ut = 0.0; B.tv = 0.0;
    B.bad = 0; B.offgrid = false;
    B.tminA = B.uminA = 0x7FFFFFFF; B.tmaxA = B.umaxA = (int)0x80000000;
    B.vminA = B.dminA = INFINITY; B.vmaxA = B.dmaxA = -INFINITY;
    B.aminA = B.amaxA = 0.0; B.kmin = B.kmax = 0;
    B.ct = B.cu = 0; B.cv = B.cd = B.cA = 0.0; B.done = 0; B.tiles = 0;
}

// ---- the bar's outputs from the folded tiles: comp_bar_ohlcv, comp_bar_directional_features, the footprint rows -> staging, the median
// KR: the registers per lane that hold the amounts of a bar of ONE tile (0: a bar of several tiles -- no median here)
template <bool UNITS, bool MEDIAN, int KR>
__device__ __forceinline__ void fu_finish(const double *__restrict__ price, int64_t b, int64_t start, int64_t e, int64_t cnt, int lane,
                                          int q, double inv_tick, FuBar &B, int last_len, float (&keep)[KR > 0 ? KR : 1], unsigned long long *hist,
                                          double tile_first, double tile_last,
                                          uint32_t *mbuf, int &wq, FuMed &med, const FuArgs *__restrict__ args, const FuPend *pend = nullptr)
{
    // (the argument block is written by the host before the launch and never by a kernel: constant address space -> scalar loads at the
    //  point of use instead of 64-lane vector loads, which is what a plain pointer gets behind the walk's LDS atomics and stores)
    typedef const FuArgs __attribute__((address_space(4))) *FuArgsC;
    const FuArgsC cargs = (FuArgsC)(uintptr_t)args;
    const auto &oo = cargs->oo;
    const auto &o = cargs->o;
    const auto &stg = cargs->stg;
    const auto &li = cargs->li;
    const double hi_w = fu_rmax(B.hi);
    const double lo_w = fu_rmin(B.lo);
    const double cs_w = fu_rsum(B.cs);
    const double mxs_w = fu_rmax(B.mxs);
    const double td_w = B.cA, rd_w = B.cd;
    const int st_w = B.ct;
    const bool any_bad = __ballot(B.bad != 0) != 0;
    const bool any_off = __ballot(B.offgrid) != 0;
    const int tmin_w = fmk_dpp_reduce(B.tminA, 0x7FFFFFFF, FmkOpMin());
    const int tmax_w = fmk_dpp_reduce(B.tmaxA, (int)0x80000000, FmkOpMax());
    const double dmin_w = fu_rmin(B.dminA);
    const double dmax_w = fu_rmax(B.dmaxA);
    // ---- volume: the exact unit total (UNITS, when the units certify) or the float64 sum of the amounts
    double tv_w, vb, vs, cvmin, cvmax;
    bool vol_ok;
    if constexpr (UNITS) {
        const double ut_w = fu_rsum(B.ut);
        vol_ok = !any_bad && ut_w < 2147483648.0;
        if (vol_ok) tv_w = ldexp(ut_w, q);
        else {
            // (uncertified sizes: the float64 sum of the amounts as comp_bar_ohlcv adds them -- one tile's are still in the registers, a
            //  longer bar's are read again; its order-flow and footprint go to the lists anyway)
            double tv = 0.0;
            if constexpr (KR > 0) {
#pragma unroll
                for (int i = 0; i < KR; ++i) tv += (i < KR - 1 || last_len == KR) ? (double)keep[i] : 0.0;
            } else {
                for (int64_t j = start + lane; j <= e; j += 64) tv += (double)((const float *)cargs->amount)[j];
            }
            tv_w = fu_rsum(tv);
        }
        const int umin_w = fmk_dpp_reduce(B.uminA, 0x7FFFFFFF, FmkOpMin());
        const int umax_w = fmk_dpp_reduce(B.umaxA, (int)0x80000000, FmkOpMax());
        vb = ldexp(0.5 * (ut_w + (double)B.cu), q); vs = ldexp(0.5 * (ut_w - (double)B.cu), q);
        cvmin = ldexp((double)umin_w, q); cvmax = ldexp((double)umax_w, q);
    } else {
        tv_w = fu_rsum(B.tv);
        vol_ok = !any_bad;
        vb = 0.5 * (tv_w + B.cv); vs = 0.5 * (tv_w - B.cv);
        cvmin = fu_rmin(B.vminA);
        cvmax = fu_rmax(B.vmaxA);
    }
    // ---- comp_bar_ohlcv (base.py:306-407)
    const double first = KR > 0 ? tile_first : price[start];
    unsigned pvalid = 0x7Fu;
    if (lane == 0) {
        const double last = KR > 0 ? tile_last : price[e];
        const double hi_o = first != first ? first : hi_w, lo_o = first != first ? first : lo_w;      // a NaN first price never loses (base.py:371-382)
        const double vwap = tv_w > 0.0 ? td_w / tv_w : 0.0;  // base.py:398
        if (pend) {
            unsigned long long *ps_ = pend->slot;
            ps_[0] = (unsigned long long)__double_as_longlong(first); ps_[1] = (unsigned long long)__double_as_longlong(hi_o);
            ps_[2] = (unsigned long long)__double_as_longlong(lo_o); ps_[3] = (unsigned long long)__double_as_longlong(last);
            ps_[4] = (unsigned long long)__float_as_uint((float)tv_w); ps_[5] = (unsigned long long)__double_as_longlong(vwap);
            ps_[6] = (unsigned long long)cnt; ps_[22] = (unsigned long long)b;
        } else {
            oo.open[b] = first; oo.close[b] = last; oo.high[b] = hi_o; oo.low[b] = lo_o;
            oo.vol[b] = (float)tv_w; oo.vwap[b] = vwap; oo.trades[b] = cnt;
        }
    }
    // ---- comp_bar_directional_features (base.py:409-546)
    const bool dir_ok = vol_ok && lo_w > 0.0 && first == first && td_w < INFINITY;      // (a NaN / inf price anywhere: not this class)
    if (!dir_ok) {
        if (lane == 0) li.dir_list[32 + atomicAdd(li.dir_list, 1ULL)] = (unsigned long long)b;
    } else {
        // every tick is a buy or a sell: counts and volumes from the totals and the signed totals (exact)
        const int64_t tb = (cnt + st_w) >> 1, tsell = (cnt - st_w) >> 1;
        // ... the dollar sums likewise, but these are float64 sums of rounded products: db = (td + rd) / 2 differs from the reference's
        // tick-order sum of the buy terms by the rounding noise of both orders -- the tie test below knows
        const double db = 0.5 * (td_w + rd_w), ds = 0.5 * (td_w - rd_w);
        const double mean = cs_w / (double)cnt;
        // Bounds (u = 2^-53, A = the sum of the terms' magnitudes = td, every term >= 0 in the class, t = the bar's tiles):
        //   * db / ds.  Reference: the recursive sum of len terms of one sign, (len - 1) u S.  Here: td and rd each through <= R additions
        //     in a lane, a scan tree of 6 levels whose nodes are disjoint ranges and the carry from tile to tile, then one addition and a
        //     halving: (R + 8 + t) u A.
        //   * extrema of the running signed sum, for an extremum found in a lane that ends k ticks into the bar, A_k the terms up to there:
        //     reference k u M (M = the largest magnitude the running sum takes); here the lane's own running sum (<= R additions of values
        //     below A_k) + the exclusive prefix (lane totals through the scan, the tile carries: (R + 6 + t) u A_k) + two additions.
        //     The bound belongs to the POSITION: an early extremum of small magnitude -- fine float32 spacing -- has few terms in front.
        //     Every lane evaluates the test on its own candidate (which costs what a wave-uniform test costs).
        const double len_d = (double)(cnt + 1);
        const double terms = (double)(2 * FU_MAXR + 16 + 2 * B.tiles);
        const double uA = 1.17e-16 * terms * td_w;
        const double md = fmax(fabs(dmin_w), fabs(dmax_w));
        unsigned mask = 0;
        if (fmk_near_f32_tie(db, 1.17e-16 * len_d * db + uA)) mask |= 1u << 2;
        if (fmk_near_f32_tie(ds, 1.17e-16 * len_d * ds + uA)) mask |= 1u << 3;
        {
            const bool tmn = B.dminA == dmin_w && fmk_near_f32_tie(B.dminA, 1.17e-16 * ((double)(B.kmin + 1) * md + terms * B.aminA));
            const bool tmx = B.dmaxA == dmax_w && fmk_near_f32_tie(B.dmaxA, 1.17e-16 * ((double)(B.kmax + 1) * md + terms * B.amaxA));
            if (__ballot(tmn || tmx) != 0) mask |= 1u << 6;
        }
        if (fmk_near_f32_tie(mean, (1.17e-16 * (len_d + terms) + 1.2e-16) * fabs(mean))) mask |= 1u << 4;
        if (bf_force_redo != 0) mask = 0x7F;
        if (lane == 0) {
            if (mask) {
                // (bars of one tile: the wave-per-bar redo; longer ones: k_bar_dir's chunk-record kernel)
                unsigned long long *rl = cnt <= FU_MAXT ? li.redo : li.redo_long;
                rl[32 + atomicAdd(rl, 1ULL)] = (unsigned long long)b | ((unsigned long long)mask << 48);
            }
            if (pend) {
                unsigned long long *ps_ = pend->slot;
                ps_[8] = (unsigned long long)tb; ps_[9] = (unsigned long long)tsell;
                ps_[10] = (unsigned long long)__float_as_uint((float)vb); ps_[11] = (unsigned long long)__float_as_uint((float)vs);
                ps_[12] = (unsigned long long)__float_as_uint((float)db); ps_[13] = (unsigned long long)__float_as_uint((float)ds);
                ps_[14] = (unsigned long long)__float_as_uint((float)mean); ps_[15] = (unsigned long long)__float_as_uint((float)mxs_w);
                ps_[16] = (unsigned long long)(long long)tmin_w; ps_[17] = (unsigned long long)(long long)tmax_w;
                ps_[18] = (unsigned long long)__float_as_uint((float)cvmin); ps_[19] = (unsigned long long)__float_as_uint((float)cvmax);
                ps_[20] = (unsigned long long)__float_as_uint((float)dmin_w); ps_[21] = (unsigned long long)__float_as_uint((float)dmax_w);
            } else {
                o.ticks_buy[b] = tb; o.ticks_sell[b] = tsell;
                o.volume_buy[b] = (float)vb; o.volume_sell[b] = (float)vs;
                o.dollars_buy[b] = (float)db; o.dollars_sell[b] = (float)ds;
                o.max_spread[b] = (float)mxs_w;
                o.mean_spread[b] = (float)mean;
                o.cum_ticks_min[b] = tmin_w; o.cum_ticks_max[b] = tmax_w;
                o.cum_volumes_min[b] = (float)cvmin; o.cum_volumes_max[b] = (float)cvmax;
                o.cum_dollars_min[b] = (float)dmin_w; o.cum_dollars_max[b] = (float)dmax_w;
            }
        }
        pvalid |= 0x3FFF00u;
    }
    // ---- footprint rows -> staging (levels low .. high, at most FU_LV; the histogram slot of a level is level mod FU_LV)
    if constexpr (UNITS) {
        const double qlo = lo_w * inv_tick, qhi = hi_w * inv_tick;
        const int lowl = (int)rint(qlo), highl = (int)rint(qhi);
        const int L = highl - lowl + 1;
        // lane l reads levels l and l + 64, and clears slot pairs l and l + 64 (every slot)
        const unsigned slot0 = (unsigned)(lowl + lane) & (FU_LV - 1u), slot1 = (unsigned)(lowl + lane + 64) & (FU_LV - 1u);
        const unsigned long long hb0 = hist[2 * slot0], hs0 = hist[2 * slot0 + 1];
        const unsigned long long hb1 = hist[2 * slot1], hs1 = hist[2 * slot1 + 1];
        if (!pend) {
            unsigned long long z = 0ULL;
            asm volatile("" : "+v"(z));                         // (a zero made HERE: as a loop invariant it lives in four registers through every walk)
            hist[2 * lane] = z; hist[2 * lane + 1] = z;
            hist[2 * lane + 128] = z; hist[2 * lane + 129] = z;
        }
        const unsigned ub0 = (unsigned)hb0, us0 = (unsigned)hs0, ub1 = (unsigned)hb1, us1 = (unsigned)hs1;
        // every (level, side) total below 2^24 units: each float32 add of the reference is exact whatever its order (fp_certified_units_per_key)
        const bool big = __ballot((lane < L && ((ub0 | us0) >> 24) != 0) || (lane + 64 < L && ((ub1 | us1) >> 24) != 0)) != 0;
        const bool fp_ok = dir_ok && !any_off && L >= 1 && L <= FU_LV && !big && fabs(qlo) < 1e9 && fabs(qhi) < 1e9;
        if (fp_ok && pend) {
            // (the rows leave with the bar's other values, behind the next bar's load requests: fu_flush reads this histogram again)
            if (lane == 0) {
                pend->slot[24] = (unsigned long long)(unsigned)L; pend->slot[25] = (unsigned long long)(unsigned)lowl;
                pend->slot[26] = (unsigned long long)(unsigned)q; pend->slot[27] = (unsigned long long)(unsigned)L;
            }
            pvalid |= 1u << 22;
        } else if (fp_ok) {
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
        } else {
            if (lane == 0) {
                if (pend) { pend->slot[24] = 0ULL; pend->slot[27] = 0xFFFFFFFFULL; }
                else stg.L[b] = -1;
                li.fp_list[32 + atomicAdd(li.fp_list, 1ULL)] = (unsigned long long)b;
            }
            if (pend) pvalid |= 1u << 22;
        }
        __builtin_amdgcn_wave_barrier();
        // a bar that did not certify: the next bar of the wave measures its own quantum
        if (!vol_ok) wq = FU_Q_UNKNOWN;
    }
    // ---- median trade size (base.py:401-404) from the amounts in the registers (a bar of one tile; longer ones: fmk_median_launch)
#ifdef FU_TIMING
    { const unsigned long long now_ = __builtin_readcyclecounter(); if (lane == 0) atomicAdd(&fu_phase_cycles[3], now_); }
#endif
    if constexpr (MEDIAN && KR > 0) {
        typedef MedKey<false> MK;
        uint32_t key[KR];
#pragma unroll
        for (int i = 0; i < KR; ++i) {
            // (opaque: otherwise the |amount| this needs is computed once with the quantum pre-pass's, in FRONT of the walk, and a
            //  second copy of every amount lives through it)
            uint32_t u = __float_as_uint(keep[i]);
            asm volatile("" : "+v"(u));
            key[i] = (i < KR - 1 || last_len == KR) ? MK::tokey(u) : MK::MAXK;
        }
        const double m = fu_median<KR>(key, (int)cnt, lane, UNITS ? any_bad : tv_w != tv_w, med, mbuf);
        if (lane == 0) {
            if (pend) pend->slot[7] = (unsigned long long)__double_as_longlong(m);
            else oo.median[b] = m;
        }
        pvalid |= 0x80u;
    }
    if (pend) {
        if (lane == 0) pend->slot[23] = (unsigned long long)pvalid;
        __builtin_amdgcn_wave_barrier();
    }
}

// the quantum of a tile's amounts: the lowest set bit over them (selects, no branches per amount: zero -> "unknown", inf / NaN -> INT_MIN,
// which the walk flags anyway)
template <int R>
__device__ __forceinline__ int fu_quantum(const float *__restrict__ amount, int64_t j0, int tcnt, int lane)
{
    int lb = FP_Q_UNKNOWN;
    for (int j = lane; j < tcnt; j += 64) {
        const uint32_t u = __float_as_uint(amount[j0 + j]) & 0x7FFFFFFFu;
        const int ex = (int)(u >> 23);
        const uint32_t mant = (u & 0x7FFFFFu) | (ex != 0 ? 0x800000u : 0u);
        int l = (ex != 0 ? ex - 150 : -149) + (int)__builtin_ctz(mant | 0x80000000u);
        l = u == 0 ? (int)FP_Q_UNKNOWN : (ex == 255 ? (int)0x80000000 : l);
        lb = l < lb ? l : lb;
    }
    lb = fmk_dpp_reduce(lb, (int)FP_Q_UNKNOWN, FmkOpMin());
    int wq = (lb == FP_Q_UNKNOWN || lb == (int)0x80000000) ? 0 : lb;     // only zeros: any quantum serves; inf / NaN: the walk flags them
    if (wq < -140) wq = -140;
    if (wq > 100) wq = 100;
    return wq;
}

#ifndef FU_WAVES
#define FU_WAVES 4
#endif
// the bars of one tile (1 .. FU_MAXT ticks): exact-size code per tick count per lane.  Longer bars raise `saw_long` and are k_fu_long's
// (two kernels: with both paths in one, the register allocator spilled loop-carried values of the bar loop and every bar began with
// a chain of scratch reloads in front of its first load).
template <bool MEDIAN, bool UNITS>
__global__ __launch_bounds__(256, FU_WAVES) void k_fu_bars(const double *__restrict__ price, const float *__restrict__ amount,
                                                    const int8_t *__restrict__ side, const int64_t *__restrict__ ci, int64_t nb,
                                                    int64_t n, double tick, const FuArgs *__restrict__ args)
{
    __shared__ unsigned long long s_hist[4][UNITS ? 4 * FU_LV : 2];      // two histograms per wave: the bar being swept, the bar whose rows wait
    __shared__ uint32_t s_buf[4][64];
    const int lane = fmk_lane();
    const int wib = fmk_uniform((int)(threadIdx.x >> 6));
    unsigned long long *hist = s_hist[wib];
    if constexpr (UNITS) {
#pragma unroll
        for (int k = 0; k < 8; ++k) hist[k * 64 + lane] = 0ULL;
    }
    __builtin_amdgcn_wave_barrier();
    const double inv_tick = 1.0 / tick;
    int wq = FU_Q_UNKNOWN;
    FuMed med{0u, 0u, 0u, 0};
    typedef const FuArgs __attribute__((address_space(4))) *FuArgsC;
    const FuArgsC cargs = (FuArgsC)(uintptr_t)args;
    __shared__ unsigned long long s_pend[4][28];
    FuPend pd;
    pd.slot = s_pend[wib];
    pd.hist = UNITS ? hist + 2 * FU_LV : hist;
    pd.scrap = (unsigned long long)(args->scrap + ((size_t)blockIdx.x * 256 + threadIdx.x));
    // Which variant defers?  MEASURED (1e9 ticks, 833 323 one-minute bars): without the histogram the deferred columns take the kernel
    // from 7.1 to 3.9 ms; WITH it -- columns, or columns and level rows (two histograms per wave) -- the kernel got slower with every
    // store that was moved, 4.0 -> 4.2 .. 5.5 ms, wherever the flush stood (behind the load requests, behind the walk).  So that
    // variant stores at once (FU_DEFER_UNITS = 1 keeps the other form reachable for measurements).
    const FuPend *const pdp = (!UNITS || FU_DEFER_UNITS) ? &pd : nullptr;
    // (FuOhlcv's 8 pointers, then FlowDirOut's 14; lane 22: FuStage's L, the fifth pointer behind them)
    pd.col = lane < 22 ? ((const unsigned long long *)&args->oo)[lane] : (lane == 22 && UNITS ? ((const unsigned long long *)&args->oo)[26] : 0ULL);
    if (lane == 0) pd.slot[23] = 0ULL;
    __builtin_amdgcn_wave_barrier();
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wib;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t b = wave0; b < nb; b += nwaves) {
        const int64_t s = fmk_uniform(ci[b]);
        const int64_t e = fmk_uniform(ci[b + 1]);
        const int64_t cnt = e - s;
        if (cnt <= 0 || cnt > FU_MAXT) {
            if (lane == 0) {
                const auto &oo = cargs->oo;
                const auto &li = cargs->li;
                if (cnt <= 0) {
                    // an empty bar: previous close (base.py:352-361); its order flow (a ZeroDivisionError of the reference) and its footprint (one
                    // level of zeros) by the list kernels
                    const double pz = price[fmk_wrap(e, n)];
                    oo.open[b] = pz; oo.high[b] = pz; oo.low[b] = pz; oo.close[b] = pz;
                    oo.vol[b] = 0.f; oo.vwap[b] = 0.0; oo.trades[b] = 0;
                    if (MEDIAN) oo.median[b] = 0.0;
                    li.dir_list[32 + atomicAdd(li.dir_list, 1ULL)] = (unsigned long long)b;
                    if (UNITS) {
                        cargs->stg.L[b] = -1;
                        li.fp_list[32 + atomicAdd(li.fp_list, 1ULL)] = (unsigned long long)b;
                    }
                } else if (__hip_atomic_load(li.saw_long, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                    __hip_atomic_store(li.saw_long, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // k_fu_long, fmk_median_launch
            }
            continue;
        }
        const int64_t start = s + 1;
        const int tcnt = (int)cnt;
        // the quantum: the previous bar's, or (first bar of the wave, or after a failure) the lowest set bit of this bar's amounts
        if (UNITS && wq == FU_Q_UNKNOWN) wq = fu_quantum<0>(amount, start, tcnt, lane);
        const int q = wq;
        FuBar B;
        FuTile T;
        fu_bar_init(B);
#ifdef FU_TIMING
        unsigned long long t_last_ = __builtin_readcyclecounter();
#endif
#define FU_CASE(RR) { float keep[RR]; \
                      fu_walk<RR, UNITS>(price, amount, side, start, tcnt, cnt == 1, n, q, inv_tick, lane, hist, B, T, keep, 0, pdp, args); \
                      FU_T(0); \
                      fu_fold<UNITS>(B, T, tcnt); \
                      fu_finish<UNITS, MEDIAN, RR>(price, b, start, e, cnt, lane, q, inv_tick, B, T.len, keep, hist, T.first, T.last, s_buf[wib], wq, med, args, pdp); \
                      FU_T(2); FU_TEND(); }
        switch ((tcnt + 63) >> 6) {
        case 1: FU_CASE(1) break; case 2: FU_CASE(2) break; case 3: FU_CASE(3) break; case 4: FU_CASE(4) break;
        case 5: FU_CASE(5) break; case 6: FU_CASE(6) break; case 7: FU_CASE(7) break; case 8: FU_CASE(8) break;
        case 9: FU_CASE(9) break; case 10: FU_CASE(10) break; case 11: FU_CASE(11) break; case 12: FU_CASE(12) break;
        case 13: FU_CASE(13) break; case 14: FU_CASE(14) break; case 15: FU_CASE(15) break; case 16: FU_CASE(16) break;
        case 17: FU_CASE(17) break; case 18: FU_CASE(18) break; case 19: FU_CASE(19) break; case 20: FU_CASE(20) break;
        case 21: FU_CASE(21) break; case 22: FU_CASE(22) break; case 23: FU_CASE(23) break;
        default: FU_CASE(24) break;
        }
#undef FU_CASE
        if constexpr (UNITS && FU_DEFER_UNITS) { unsigned long long *t = hist; hist = pd.hist; pd.hist = t; }   // the next bar sweeps into the other histogram
    }
    if (pdp) fu_flush<UNITS>(pd, lane, args);               // the wave's last bar
}

// the bars of several tiles (more than FU_MAXT ticks; launched behind k_fu_bars, returns at once when that kernel met none): whole
// tiles of FU_MAXT ticks, the last one a whole tile that ends with the bar (its overlap with the tile in front is skipped); the
// carries run from tile to tile.  64 bars per step: one coalesced load of their close indices, then only the long ones get the wave.
// Bars beyond FU_LONGEST ticks are not walked by one wave: open .. trades by comp_bar_ohlcv's leftover pass (`saw_huge`), order flow
// and footprint by the list kernels.  Medians of all of these: fmk_median_launch.
template <bool UNITS>
__global__ __launch_bounds__(256, FU_WAVES) void k_fu_long(const double *__restrict__ price, const float *__restrict__ amount,
                                                    const int8_t *__restrict__ side, const int64_t *__restrict__ ci, int64_t nb,
                                                    int64_t n, double tick, const FuArgs *__restrict__ args)
{
    typedef const FuArgs __attribute__((address_space(4))) *FuArgsC;
    const FuArgsC cargs = (FuArgsC)(uintptr_t)args;
    if (*cargs->li.saw_long == 0) return;
    __shared__ unsigned long long s_hist[4][UNITS ? 2 * FU_LV : 2];
    __shared__ uint32_t s_buf[4][64];
    const int lane = fmk_lane();
    const int wib = fmk_uniform((int)(threadIdx.x >> 6));
    unsigned long long *hist = s_hist[wib];
    if constexpr (UNITS) { hist[2 * lane] = 0ULL; hist[2 * lane + 1] = 0ULL; hist[2 * lane + 128] = 0ULL; hist[2 * lane + 129] = 0ULL; }
    __builtin_amdgcn_wave_barrier();
    const double inv_tick = 1.0 / tick;
    int wq = FU_Q_UNKNOWN;
    FuMed med{0u, 0u, 0u, 0};
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wib;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t ngroups = (nb + 63) >> 6;
    for (int64_t g = wave0; g < ngroups; g += nwaves) {
        const int64_t bl = g * 64 + lane;
        int64_t s_l = 0, e_l = 0;
        if (bl < nb) { s_l = ci[bl]; e_l = ci[bl + 1]; }
        unsigned long long todo = __builtin_amdgcn_ballot_w64(bl < nb && e_l - s_l > FU_MAXT);
        while (todo) {
            const int bit = fmk_uniform((int)__builtin_ctzll(todo));
            todo &= todo - 1;
            const int64_t b = g * 64 + bit;
            const int64_t s = fmk_readlane(s_l, bit), e = fmk_readlane(e_l, bit);
            const int64_t cnt = e - s;
            if (cnt > FU_LONGEST) {
                if (lane == 0) {
                    const auto &li = cargs->li;
                    if (__hip_atomic_load(li.saw_huge, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
                        __hip_atomic_store(li.saw_huge, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    li.dir_list[32 + atomicAdd(li.dir_list, 1ULL)] = (unsigned long long)b;
                    if (UNITS) {
                        cargs->stg.L[b] = -1;
                        li.fp_list[32 + atomicAdd(li.fp_list, 1ULL)] = (unsigned long long)b;
                    }
                }
                continue;
            }
            const int64_t start = s + 1;
            if (UNITS && wq == FU_Q_UNKNOWN) wq = fu_quantum<0>(amount, start, FU_MAXT, lane);
            const int q = wq;
            FuBar B;
            FuTile T;
            fu_bar_init(B);
            float keep[FU_MAXR];
            int64_t j0 = start, rem = cnt;
            while (rem > FU_MAXT) {
                fu_walk<FU_MAXR, UNITS>(price, amount, side, j0, FU_MAXT, false, n, q, inv_tick, lane, hist, B, T, keep);
                fu_fold<UNITS>(B, T, FU_MAXT);
                j0 += FU_MAXT; rem -= FU_MAXT;
            }
            fu_walk<FU_MAXR, UNITS, true>(price, amount, side, e - FU_MAXT + 1, FU_MAXT, false, n, q, inv_tick, lane, hist, B, T, keep,
                                          FU_MAXT - (int)rem);
            fu_fold<UNITS>(B, T, (int)rem);
            float none[1];
            fu_finish<UNITS, false, 0>(price, b, start, e, cnt, lane, q, inv_tick, B, 0, none, hist, 0.0, 0.0, s_buf[wib], wq, med, args);
        }
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
