// fmk_dollar_exact.hip -- _dollar_bar_indexer (finmlkit/bar/logic.py:118-149): the EXACT tier behind the closed form of
// fmk_dollar.hip.  The closed form decides in exact arithmetic and can only COUNT the decisions that fall inside the rounding
// drift of the reference's float64 running sum (logic.py:143-147: `cum += p*v; if cum >= thr: ...; cum = cum - thr` -- the
// carry never resets the drift: ~232 of 1.16e6 closes at 1e9 ticks).  This tier reconstructs the reference's float64 state
// itself, in parallel, and replays only the bars that need it.
//
// Why that is possible.  Let u = ulp(thr).  With every increment d_i = fl(p_i*v_i) below thr:
//  (1) after a close the carry `cum - thr` is exact (Sterbenz) and a multiple of u: the state at every bar start is k*u for
//      an integer k < 2^53.
//  (2) inside a bar, while cum is in a binade below thr's, fl(k*u + y) = k*u + fl(y) (k*u is an even multiple of the finer
//      rounding unit): the rounding errors do not depend on k at all.  In thr's own binade a tie rounds to even, which depends
//      on k mod 2; the closing add may reach the binade above (unit 2u), where the rounding depends on k mod 2 and its tie on
//      k mod 4.  Nothing else depends on k -- PROVIDED the add lands in the same binade and on the same side of thr for every
//      state the reference can be in.
//  (3) the reference's state differs from the exact-arithmetic one by at most (i+1)*2^-52*thr at tick i (sum of the add
//      roundings, each <= 2^-53 * 2*thr).  A bar none of whose adds comes within that margin of a power of two, and none of
//      whose decisions within it of thr, therefore maps its start state to its end state by a function with
//      f(k + 4) = f(k) + 4, which four float64 trajectories (k~-1 .. k~+2 around the exact-arithmetic carry k~) tabulate.
//  Such functions compose (integers only), so ONE scan over the bars gives the reference's true state at the start of every
//  bar.  The other bars ("flagged": ~0.3 % at 1e9 ticks) are then simply replayed from their true start state with the
//  reference's own operations until the walk meets a close of the closed form again; where a replay disagrees with the bar's
//  tabulated function (a real flip: the close moves by a tick) the bar becomes a constant in the scan and the scan is
//  repeated -- a fixed point after (number of dependent flips + 1) rounds.  No decision is left uncertified.
//
//   k_dlx_bars      one lane per bar, loads by the wave: four trajectories from (k~ - 1 + j)*u, flags, f = (base, kappa[4])
//   k_dlx_scan      compose the bars' functions (block = 1024 bars); phase 1 applies them: k_in per bar
//   k_dlx_blocks    one wave walks the block aggregates
//   k_dlx_resolve   one thread per flagged bar: the reference loop from k_in*u until it is back on a closed-form close
//   k_dlx_commit    rewrite the closes of the bars whose replay differed
// Cost at 1e9 ticks / 1.16e6 bars: one more read of price + amount by k_dlx_bars, the rest is per-bar data.  Streams with an increment >= thr (a backlog of closes) are not covered: the caller takes the serial walk.
#include <math.h>
#include <stdlib.h>

#include "fmk_common.h"
#include <type_traits>

#define DLX_MAXC 6                       // closes one replay may record before it has to be back on the closed form
#define DLX_CONST INT64_MIN              // DlxFn.base of a constant function
#define DLX_BLOCK_BARS 1024              // 256 threads x 4 bars

struct DlxFn {                           // f(k) = kap[(k - base) & 3] + ((k - base) & ~3), or the constant kap[0]
    int64_t base;
    int64_t kap[4];
    // >= 0: the composition STARTS behind stretch bar `seg` (round 4, streams with increments >= thr): its input is that bar's
    // output state sval[seg], whatever came before; -1: an ordinary function of the incoming state
    int64_t seg;
};

struct DlxRes {                          // state of one flagged bar between rounds
    int32_t mode;                        // 0: the replay agrees with the bar's own function; 1: replaced by a constant
    int32_t span;                        // bars covered by the replay (1 unless a close moved across a bar boundary)
    int32_t nclose;                      // closes the replay recorded
    int32_t to_end;                      // the replay ran to the end of the stream
    int64_t kout;
    int64_t close[DLX_MAXC];
};

__device__ __forceinline__ int64_t dlx_eval(const DlxFn &f, int64_t k)      // (the part behind f.seg, if any)
{
    if (f.base == DLX_CONST) return f.kap[0];
    const int64_t t = k - f.base, r = t & 3;
    return f.kap[r] + (t - r);
}
// ... with the reset: the state behind bar f.seg comes from the walk over the stretch bars (k_dlx_walk)
__device__ __forceinline__ int64_t dlx_apply(const DlxFn &f, int64_t k, const int64_t *__restrict__ sval)
{
    return dlx_eval(f, f.seg >= 0 ? sval[f.seg] : k);
}

// first f, then g
__device__ __forceinline__ DlxFn dlx_compose(const DlxFn &f, const DlxFn &g)
{
    if (g.seg >= 0 || g.base == DLX_CONST) return g;              // g does not look at its input
    DlxFn h;
    h.seg = f.seg;
    if (f.base == DLX_CONST) {
        h.base = DLX_CONST;
        h.kap[0] = dlx_eval(g, f.kap[0]);
        h.kap[1] = h.kap[2] = h.kap[3] = 0;
        return h;
    }
    h.base = f.base;
#pragma unroll
    for (int j = 0; j < 4; ++j) h.kap[j] = dlx_eval(g, f.kap[j]);
    return h;
}

// ---- stretch bars (round 4).  An increment w >= thr closes its bar on the spot and leaves a backlog: the next ~w / thr ticks each
// close a one-tick bar while cum works its way down.  Those adds happen in binades ABOVE thr's, where the rounding depends on
// more low bits of the incoming state than the four trajectories of k_dlx_bars cover (f(k + 4) = f(k) + 4 no longer holds), so
// such bars have no composable function: they are STRETCH bars, and the state behind each is computed from the true state in front
// of it -- two float64 operations (the reference's closing add and its carry) by one lane walking them in order (k_dlx_walk), the
// ordinary bars between two of them composed in parallel as before.
//   type 1 (backlog): the bar starts at or above thr (certainly): one tick, cum = k u + d, cum - thr.
//   type 2 (whale close): an ordinary start, then a closing add that lands beyond the binade above thr's.  The adds BEFORE it are
//          ordinary (period 4 in k): fn.kap[] holds the four trajectories' float64 values in front of the closing add (as bit
//          patterns), C_r; the true value is C_r + (k - base - r) u exactly (point (2) of the header); then + d, - thr.
#define DLX_T_BACKLOG 1
#define DLX_T_WHALE 2
__device__ __forceinline__ int64_t dlx_stretch_op(int type, const DlxFn &f, double dlast, int64_t k, double thr, double u, double inv_u)
{
    double c;
    if (type == DLX_T_BACKLOG) c = (double)k * u;
    else {
        const int64_t t = k - f.base, r = t & 3;
        c = __longlong_as_double(f.kap[r]) + (double)(t - r) * u;
    }
    c += dlast;                                                      // logic.py:143
    c = c - thr;                                                     // logic.py:147 (exact: Sterbenz or a coarser grid than u)
    return llrint(c * inv_u);
}

__device__ __forceinline__ void vol_giveup(unsigned long long *w)    // (one store, not one atomic per bar)
{
    if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ULL)
        __hip_atomic_store(w, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool AF64>
__device__ __forceinline__ double dlx_d(const double *price, const void *amount, int64_t i)
{
    return price[i] * fmk_amt<AF64>(amount, i);       // rounded once, like prices[i] * volumes[i] (logic.py:143)
}

__device__ __forceinline__ int64_t dlx_binade(double x) { return __double_as_longlong(x) >> 52; }   // sign + exponent

#define DLX_T 16                         // ticks a lane takes per round: 32 -> 16 -> 8 gave 5.7 / 3.5 / 4.6 ms at 1e9 ticks (the LDS rows
                                         // of a wave shrink with it: 2 -> 4 -> 8 waves per SIMD, but twice the rounds each time)
#define DLX_R_MAX 8                      // bars per lane (strided by 64: evens out the bar lengths within a wave) -- as many
                                         // as leave the chip >= 8192 workgroups: a wave is one long dependent loop

// One LANE per bar -- the adds of a bar are a dependent chain -- but the loads are the wave's: a lane that streamed its own
// bar would touch its own cache line with every load instruction (64 lines for 512 useful bytes; the first version of this
// kernel took 31.6 ms at 1e9 ticks, 0.4 TB/s; this one 3.5).  Per round every lane publishes the next DLX_T ticks it needs; the wave fetches
// the 64 segments with coalesced loads (a quarter wave per 128-byte segment), forms the rounded products and parks them in LDS,
// one padded row per lane; then every lane walks its own row.
template <bool AF64>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_dlx_bars(const double *__restrict__ price, const void *__restrict__ amount, int64_t n,
                                                  double thr, double u, double inv_u, double m_rel, double m_abs,
                                                  const int64_t *__restrict__ ci, int64_t nb,
                                                  const int64_t *__restrict__ carry_k, DlxFn *__restrict__ fn,
                                                  int32_t *__restrict__ fidx, int32_t *__restrict__ owner,
                                                  int64_t *__restrict__ flist, unsigned long long *__restrict__ n_flag,
                                                  int bars_per_lane, int whales, double m_extra /* backlog term of the drift */,
                                                  int64_t tu /* thr in units of u */, double top /* 2 x the power of two above thr */,
                                                  unsigned char *__restrict__ btype, double *__restrict__ dlast)
{
    __shared__ double rows[2][64][DLX_T + 1];               // 2 waves x 16.9 KB
    __shared__ int64_t s_pos[2][64], s_end[2][64], s_prev[2][64];
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
    float keep[64 / (64 / DLX_T)];                            // (float32 amounts) the second half of the amount line of each segment this lane loads for
#pragma unroll
    for (int q = 0; q < 64 / (64 / DLX_T); ++q) keep[q] = 0.f;
    s_prev[w][lane] = -1;
    const int64_t wave = (int64_t)blockIdx.x * 2 + w;
    const int64_t bar0 = wave * (64 * (int64_t)bars_per_lane) + lane;
    int r = 0;
    int64_t b = bar0, pos = 0, end = -1, kb = 0;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, m = 0;
    double p0 = 0, p1 = 0, p2 = 0, p3 = 0, dcl = 0;           // a whale close: the trajectories in front of the closing add, its increment
    bool tail = false, flag = false, active = false, wclose = false;
    auto open_bar = [&]() {                          // next bar of this lane, or none
        for (;;) {
            active = r < bars_per_lane && b <= nb;
            if (!active) return;
            tail = b == nb;
            pos = ci[b] + 1;
            end = tail ? n - 1 : ci[b + 1];
            // how far the reference's state can be from a simulated trajectory at the end of this bar: its drift against
            // exact arithmetic (3), the rounding of the exact carry to k~, the spread of the trajectories, their parity wiggle
            m = ((double)(end + 2) + m_extra) * m_rel + m_abs;
            flag = false;
            wclose = false;
            kb = 0;
            bool backlog = false;
            if (b == 0) {
                c0 = c1 = c2 = c3 = dlx_d<AF64>(price, amount, 0);   // cum = prices[0] * volumes[0] (logic.py:140): exact
            } else {
                const int64_t kc = carry_k[b];
                if (kc < 0) { vol_giveup(n_flag + 2); kb = 0; }      // a backlog of more than 500 thresholds: no exact tier
                else kb = kc - 1;
                if (kb < 0) kb = 0;
                // the bar STARTS at or above the threshold for every state the reference can be in: a backlog bar (type 1)
                backlog = whales && !tail && (double)kc * u - m >= thr && pos <= end;
                c0 = (double)kb * u; c1 = (double)(kb + 1) * u; c2 = (double)(kb + 2) * u; c3 = (double)(kb + 3) * u;
            }
            if (btype) btype[b] = 0;
            if (pos <= end && !backlog) return;
            // an empty tail (nothing can close behind the last close), or a backlog bar: no trajectories
            DlxFn f;
            f.base = kb; f.kap[0] = kb; f.kap[1] = kb + 1; f.kap[2] = kb + 2; f.kap[3] = kb + 3; f.seg = -1;
            if (backlog) {
                if (pos != end) vol_giveup(n_flag + 2);              // (its first add closes it: the closed form says so too)
                btype[b] = DLX_T_BACKLOG;
                dlast[b] = dlx_d<AF64>(price, amount, pos);
            }
            fn[b] = f; owner[b] = (int32_t)b; fidx[b] = -1;
            ++r; b += 64;
        }
    };
    open_bar();
    while (__ballot(active) != 0) {
        // the window a lane asks for starts on a DLX_T-tick boundary (whole 128-byte lines of prices when the column is
        // line-aligned): only the first round of a bar is partial.  Windows at the lane's exact position made the kernel fetch
        // 19.75 GB for its 12 (2 x FETCH_SIZE); aligned: 12.4 GB (profiles/r02_dollar_two_pass.txt)
        s_pos[w][lane] = active ? (pos & ~(int64_t)(DLX_T - 1)) : -1;
        s_end[w][lane] = active ? end : -2;
        __builtin_amdgcn_wave_barrier();
        constexpr int SPI = 64 / DLX_T;                              // segments per load instruction
        const int half = lane / DLX_T, j32 = lane & (DLX_T - 1);
        // All loads of the round in flight together -- which the loop above this comment used to SAY and not do: with the product and its
        // LDS store inside each guarded block the compiler waited for every pair of loads before it issued the next (LLWW x 16 in the
        // ISA: sixteen memory round trips per round, and with 4 waves per SIMD that latency WAS the kernel: 3.8 ms per 1e9 ticks).
        // Now: the raw words of all sixteen segments into registers, pinned behind the last load, then the products.
        typedef typename std::conditional<AF64, double, float>::type DlxAmt;
        double lp[64 / SPI];
        DlxAmt la[64 / SPI];
        // float32 amounts: a window of 16 ticks is HALF a 128-byte line of amounts, and the other half was asked for a round later --
        // by then L2 had turned over and the line came from memory again (16 GB fetched for the kernel's 12).
        // Round 5: a window that starts a line also fetches the line's second half (the same lane serves the same segment and tick
        // offset in the next round) and keeps it in a register; s_prev says whether this round continues the last one's window.
#pragma unroll
        for (int q = 0; q < 64 / SPI; ++q) {
            const int seg = q * SPI + half;
            const int64_t p0 = s_pos[w][seg], e0 = s_end[w][seg];
            const int64_t tick = p0 + j32;
            lp[q] = 0.0; la[q] = (DlxAmt)0;
            if constexpr (AF64) {
                if (p0 >= 0 && tick <= e0) { lp[q] = price[tick]; la[q] = ((const DlxAmt *)amount)[tick]; }
            } else {
                const bool second = (p0 & DLX_T) != 0;
                const bool reuse = second && p0 >= 0 && s_prev[w][seg] == p0 - DLX_T;
                if (p0 >= 0 && tick <= e0) { lp[q] = price[tick]; la[q] = reuse ? keep[q] : ((const DlxAmt *)amount)[tick]; }
                if (p0 >= 0 && !second && tick + DLX_T < n) keep[q] = ((const DlxAmt *)amount)[tick + DLX_T];
            }
        }
#pragma unroll
        for (int q = 0; q < 64 / SPI; ++q) {
            asm volatile("" : "+v"(lp[q])); asm volatile("" : "+v"(la[q]));
            if constexpr (!AF64) asm volatile("" : "+v"(keep[q]));
        }
#pragma unroll
        for (int q = 0; q < 64 / SPI; ++q)
            rows[w][q * SPI + half][j32] = lp[q] * (double)la[q];   // rounded once, like prices[i] * volumes[i] (logic.py:143): dlx_d
        __builtin_amdgcn_wave_barrier();
        s_prev[w][lane] = active ? (pos & ~(int64_t)(DLX_T - 1)) : -1;   // (read by the loaders of the NEXT round, behind its barrier)
        int cnt = 0;
        const int o = (int)(pos & (DLX_T - 1));                     // my first tick inside the window
        if (active) cnt = (int)(end - pos + 1 < DLX_T - o ? end - pos + 1 : DLX_T - o);
        const bool closes_here = active && !tail && pos + cnt - 1 == end;     // the closing add is the last tick of this round
#pragma unroll 4
        for (int j = 0; j < DLX_T; ++j) {
            if (j < cnt) {
                const double d = rows[w][lane][(o + j) & (DLX_T - 1)];
                const bool closing = closes_here && j == cnt - 1;
                if (closing) { p0 = c0; p1 = c1; p2 = c2; p3 = c3; dcl = d; }
                c0 += d; c1 += d; c2 += d; c3 += d;
                const double lo = c0 - m, hi = c3 + m;
                if (closing && whales && hi >= top) {
                    wclose = true;                               // lands (or may land) beyond the binade above thr's: a whale close (type 2)
                    flag |= !(lo >= thr);
                } else {
                    bool bad = dlx_binade(lo) != dlx_binade(hi); // the add must land in one binade for every possible state
                    if (closing) bad |= !(lo >= thr);            // the closing add reaches the threshold for every state
                    else bad |= hi >= thr;                       // every other add stays below it
                    flag |= bad;
                }
            }
        }
        pos += cnt;
        const bool finished = active && pos > end;
        bool report = false;
        if (finished) {
            DlxFn f;
            f.base = kb;
            f.seg = -1;
            if (wclose) {
                f.kap[0] = __double_as_longlong(p0); f.kap[1] = __double_as_longlong(p1);
                f.kap[2] = __double_as_longlong(p2); f.kap[3] = __double_as_longlong(p3);
                btype[b] = DLX_T_WHALE;
                dlast[b] = dcl;
            } else if (!tail) {
                f.kap[0] = llrint((c0 - thr) * inv_u);           // cum - thr is exact and a multiple of u (1)
                f.kap[1] = llrint((c1 - thr) * inv_u);
                f.kap[2] = llrint((c2 - thr) * inv_u);
                f.kap[3] = llrint((c3 - thr) * inv_u);
                if (b == 0) f.base = DLX_CONST;                  // the first bar starts from a known state: a constant
            } else {
                f.kap[0] = kb; f.kap[1] = kb + 1; f.kap[2] = kb + 2; f.kap[3] = kb + 3;   // identity
            }
            fn[b] = f;
            owner[b] = (int32_t)b;
            report = flag;
            if (!flag) fidx[b] = -1;
        }
        const uint64_t rep = __ballot(report);
        if (rep) {                                               // one atomic per wave and round, not one per flagged bar
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(n_flag, (unsigned long long)__popcll(rep));
            base = __shfl(base, 0, 64);
            if (report) {
                const int64_t fi = (int64_t)base + __popcll(rep & ((1ULL << lane) - 1));
                flist[fi] = b;
                fidx[b] = (int32_t)fi;
            }
        }
        if (finished) { ++r; b += 64; open_bar(); }
    }
}

// the function bar x contributes to the scan in this round
__device__ __forceinline__ DlxFn dlx_effective(int64_t x, const DlxFn *fn, const int32_t *fidx, const int32_t *owner,
                                               const DlxRes *res, const unsigned char *btype)
{
    const int32_t o = owner[x];
    const int32_t f = fidx[o];
    if (f >= 0 && res && res[f].mode == 1) {         // the bar, or the bar whose replay absorbed it, is a constant
        DlxFn c;
        c.base = DLX_CONST;
        c.kap[0] = res[f].kout;
        c.kap[1] = c.kap[2] = c.kap[3] = 0;
        c.seg = -1;
        return c;
    }
    if (btype[x]) {                                  // a stretch bar: what follows it starts from ITS output (identity behind a reset)
        DlxFn c;
        c.base = 0; c.kap[0] = 0; c.kap[1] = 1; c.kap[2] = 2; c.kap[3] = 3;
        c.seg = x;
        return c;
    }
    return fn[x];
}
__device__ __forceinline__ bool dlx_is_stretch(int64_t x, const int32_t *fidx, const int32_t *owner, const DlxRes *res,
                                               const unsigned char *btype)
{
    if (!btype[x]) return false;
    const int32_t f = fidx[owner[x]];
    return !(f >= 0 && res && res[f].mode == 1);     // (a constant, or absorbed by one, is no longer a stretch bar)
}

// phase 0: blk[block] = composition of the block's 1024 bars, and for every stretch bar the composition of the bars of its block IN
// FRONT of it (spre: what k_dlx_walk applies to the block's incoming state, or to the previous stretch bar's output, to get the
// state in front of the bar).  phase 1: kin[x] = state at the start of bar x.
__global__ __launch_bounds__(256) void k_dlx_scan(const DlxFn *__restrict__ fn, const int32_t *__restrict__ fidx,
                                                  const int32_t *__restrict__ owner, const DlxRes *__restrict__ res,
                                                  int64_t nbar, int phase, DlxFn *__restrict__ blk,
                                                  const int64_t *__restrict__ blk_in, int64_t *__restrict__ kin,
                                                  const unsigned char *__restrict__ btype, DlxFn *__restrict__ spre,
                                                  const int64_t *__restrict__ sval, int *__restrict__ scount = nullptr)
{
    __shared__ DlxFn buf[2][256];
    const int t = threadIdx.x;
    const int64_t x0 = (int64_t)blockIdx.x * DLX_BLOCK_BARS + (int64_t)t * 4;
    DlxFn mine[4];
    DlxFn acc;
    acc.base = 0; acc.kap[0] = 0; acc.kap[1] = 1; acc.kap[2] = 2; acc.kap[3] = 3; acc.seg = -1;      // identity
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (x0 + j < nbar) {
            mine[j] = dlx_effective(x0 + j, fn, fidx, owner, res, btype);
            acc = dlx_compose(acc, mine[j]);
        }
    }
    buf[0][t] = acc;
    __syncthreads();
    int cur = 0;
    for (int o = 1; o < 256; o <<= 1) {             // inclusive scan of the threads' functions, earlier bars applied first
        DlxFn v = buf[cur][t];
        if (t >= o) v = dlx_compose(buf[cur][t - o], v);
        buf[cur ^ 1][t] = v;
        __syncthreads();
        cur ^= 1;
    }
    // the composition of the block's bars in front of this thread's first bar
    DlxFn ex;
    ex.base = 0; ex.kap[0] = 0; ex.kap[1] = 1; ex.kap[2] = 2; ex.kap[3] = 3; ex.seg = -1;
    if (t > 0) ex = buf[cur][t - 1];
    if (phase == 0) {
        if (t == 255) blk[blockIdx.x] = buf[cur][255];
        int mystretch = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (x0 + j < nbar) {
                if (mine[j].seg == x0 + j) { spre[x0 + j] = ex; ++mystretch; }   // (mine[j].seg == its own index: an effective stretch bar)
                ex = dlx_compose(ex, mine[j]);
            }
        }
        if (scount) {
            __shared__ int s_cnt;
            if (t == 0) s_cnt = 0;
            __syncthreads();
            if (mystretch) atomicAdd(&s_cnt, mystretch);
            __syncthreads();
            if (t == 0) scount[blockIdx.x] = s_cnt;
        }
        return;
    }
    const int64_t kb = blk_in[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (x0 + j < nbar) {
            kin[x0 + j] = dlx_apply(ex, kb, sval);
            ex = dlx_compose(ex, mine[j]);
        }
    }
}

// one wave: the lanes fetch 64 block aggregates, lane 0 applies them in order (streams without stretch bars)
__global__ __launch_bounds__(64) void k_dlx_blocks(const DlxFn *__restrict__ blk, int64_t nblk, int64_t *__restrict__ blk_in)
{
    __shared__ DlxFn s[64];
    const int lane = threadIdx.x;
    int64_t k = 0;                                   // bar 0 is a constant: the start value never matters
    for (int64_t g = 0; g < nblk; g += 64) {
        if (g + lane < nblk) s[lane] = blk[g + lane];
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            const int lim = (int)(nblk - g < 64 ? nblk - g : 64);
            for (int j = 0; j < lim; ++j) {
                blk_in[g + j] = k;
                k = dlx_eval(s[j], k);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Streams WITH stretch bars.  The states behind them are a strictly serial chain (nothing composes across a stretch bar), so the chain
// is made as short as it can be: an ordered list of EVENTS -- every stretch bar, and the end of every scan block -- each with the
// function that leads from the state behind the previous event to its own result, prepared for all events in parallel:
//   c(S) = Cc[r] + ((S / u - base) - r) u,  r = (S / u - base) mod 4          (a constant Cc[0] when the function is one)
//   block end       : the state is c(S)                       (Cc[r] = kap[r] u of the block's composition since its last stretch bar)
//   stretch, general: the state is (c(S) + d) - thr           (Cc[r] = the float64 value in front of the bar's closing add for the
//                     incoming state base + r: the bars since the last event composed with the bar's own four trajectories)
//   stretch, light  : the state is (S + d) - thr              (a backlog bar right behind another stretch bar: nothing in between)
// and ONE wave walks them: 64 events per coalesced load (the next loads in flight), the state in real units (the reference's own
// float64 cum) in a register.  ~12 instructions per light event, ~32 per general one.  First version (the walk fetching type,
// pre-function, trajectories and increment per bar, a chain of dependent loads per 64 bars, integer states): 30 ms per 1e9 ticks
// with 1e5 block trades; this one: see profiles/r04_dollar_block_trades.txt.
struct DlxEvent {
    double base;                         // of the function since the last event, in units of u
    double cc[4];
    double d;                            // the closing increment (stretch events)
    int32_t flags;                       // DLX_EV_*
    int32_t idx;                         // bar (stretch) or scan block (block end)
};
#define DLX_EV_STRETCH 1
#define DLX_EV_LIGHT 2
#define DLX_EV_CONST 4

// exclusive scan of (stretch bars + 1) over the scan blocks: where each block's events start; total -> off[nblk]
__global__ __launch_bounds__(1024) void k_dlx_evoff(const int *__restrict__ scount, int64_t nblk, int64_t *__restrict__ off)
{
    __shared__ int64_t ws[16];
    __shared__ int64_t run;
    if (threadIdx.x == 0) run = 0;
    __syncthreads();
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
    for (int64_t b0 = 0; b0 < nblk; b0 += 1024) {
        const int64_t b = b0 + threadIdx.x;
        const int64_t v = b < nblk ? (int64_t)scount[b] + 1 : 0;
        const int64_t inc = fmk_wave_iscan(v);
        if (lane == 63) ws[w] = inc;
        __syncthreads();
        int64_t pre = run;
        for (int q = 0; q < w; ++q) pre += ws[q];
        if (b < nblk) off[b] = pre + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) run = pre + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) off[nblk] = run;
}

__device__ __forceinline__ void dlx_event_fn(DlxEvent &e, const DlxFn &g, double u)      // block end / backlog bar: c(S) = g(S / u) u
{
    e.base = (double)g.base;
    if (g.base == DLX_CONST) {
        e.flags |= DLX_EV_CONST;
        e.base = 0.0;
        e.cc[0] = (double)g.kap[0] * u; e.cc[1] = e.cc[2] = e.cc[3] = 0.0;
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) e.cc[r] = (double)g.kap[r] * u;
    }
}

// the events of one scan block, in order (same bar ownership as k_dlx_scan: thread t has bars 4t .. 4t + 3 of the block)
__global__ __launch_bounds__(256) void k_dlx_events(const DlxFn *__restrict__ fn, const int32_t *__restrict__ fidx,
                                                    const int32_t *__restrict__ owner, const DlxRes *__restrict__ res,
                                                    const unsigned char *__restrict__ btype, const DlxFn *__restrict__ spre,
                                                    const double *__restrict__ dlast, const DlxFn *__restrict__ blk,
                                                    const int64_t *__restrict__ off, int64_t nbar, double u,
                                                    DlxEvent *__restrict__ ev)
{
    __shared__ int wsum[4];
    const int t = threadIdx.x, lane = fmk_lane(), w = t >> 6;
    const int64_t x0 = (int64_t)blockIdx.x * DLX_BLOCK_BARS + (int64_t)t * 4;
    bool st[4];
    int mine = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        st[j] = x0 + j < nbar && dlx_is_stretch(x0 + j, fidx, owner, res, btype);
        mine += st[j] ? 1 : 0;
    }
    const int inc = fmk_wave_iscan(mine);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int rank = inc - mine;
    for (int q = 0; q < w; ++q) rank += wsum[q];
    const int64_t o = off[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (!st[j]) continue;
        const int64_t x = x0 + j;
        const DlxFn pre = spre[x];
        const int type = btype[x];
        DlxEvent e;
        e.flags = DLX_EV_STRETCH;
        e.idx = (int32_t)x;
        e.d = dlast[x];
        const bool ident = pre.base == 0 && pre.kap[0] == 0 && pre.kap[1] == 1 && pre.kap[2] == 2 && pre.kap[3] == 3;
        if (type == DLX_T_BACKLOG) {
            if (ident) { e.flags |= DLX_EV_LIGHT; e.base = 0.0; e.cc[0] = e.cc[1] = e.cc[2] = e.cc[3] = 0.0; }
            else dlx_event_fn(e, pre, u);
        } else {
            // the bars since the last event, then the whale bar's own trajectories in front of its closing add
            const DlxFn f = fn[x];
            if (pre.base == DLX_CONST) {
                const int64_t t2 = pre.kap[0] - f.base, r2 = t2 & 3;
                e.flags |= DLX_EV_CONST;
                e.base = 0.0;
                e.cc[0] = __longlong_as_double(f.kap[r2]) + (double)(t2 - r2) * u;
                e.cc[1] = e.cc[2] = e.cc[3] = 0.0;
            } else {
                e.base = (double)pre.base;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t t2 = pre.kap[r] - f.base, r2 = t2 & 3;
                    e.cc[r] = __longlong_as_double(f.kap[r2]) + (double)(t2 - r2) * u;
                }
            }
        }
        ev[o + rank] = e;
        ++rank;
    }
    if (t == 255) {                                  // (its running rank is the block's count)
        DlxEvent e;
        e.flags = 0;
        e.idx = (int32_t)blockIdx.x;
        e.d = 0.0;
        dlx_event_fn(e, blk[blockIdx.x], u);
        ev[o + rank] = e;
    }
}

__global__ __launch_bounds__(64) void k_dlx_walk(const DlxEvent *__restrict__ ev, const int64_t *__restrict__ off, int64_t nblk,
                                                 double thr, double u, double inv_u, double *__restrict__ out)
{
    const int lane = threadIdx.x;
    const int64_t nev = off[nblk];
    double S = 0.0;                                  // the reference's cum behind the last event (bar 0 is a constant: the start never matters)
    DlxEvent cur, nxt;
    cur.flags = 0; cur.base = 0; cur.d = 0; cur.cc[0] = cur.cc[1] = cur.cc[2] = cur.cc[3] = 0; cur.idx = 0;
    if (lane < nev) cur = ev[lane];
    for (int64_t e0 = 0; e0 < nev; e0 += 64) {
        nxt = cur;
        if (e0 + 64 + lane < nev) nxt = ev[e0 + 64 + lane];          // in flight while this group is walked
        const int lim = (int)(nev - e0 < 64 ? nev - e0 : 64);
        double mine = 0.0;
        for (int j = 0; j < lim; ++j) {
            const int fl = __builtin_amdgcn_readlane(cur.flags, j);
            const double d = __longlong_as_double(fmk_readlane((int64_t)__double_as_longlong(cur.d), j));
            double c;
            if (fl & DLX_EV_LIGHT) c = S;
            else {
                const double base = __longlong_as_double(fmk_readlane((int64_t)__double_as_longlong(cur.base), j));
                const double tt = S * inv_u - base;                   // exact: both integers (in units of u), the difference small
                const double r = tt - 4.0 * floor(tt * 0.25);         // tt mod 4, in [0, 4)
                const int ri = (fl & DLX_EV_CONST) ? 0 : (int)r;
                const double sel = ri == 0 ? cur.cc[0] : (ri == 1 ? cur.cc[1] : (ri == 2 ? cur.cc[2] : cur.cc[3]));   // (every lane, on its own record)
                const double cc = __longlong_as_double(fmk_readlane((int64_t)__double_as_longlong(sel), j));
                c = (fl & DLX_EV_CONST) ? cc : cc + (tt - r) * u;
            }
            if (fl & DLX_EV_STRETCH) { c += d; c = c - thr; }         // logic.py:143, 147
            S = c;
            if (lane == j) mine = S;
        }
        if (lane < lim) out[e0 + lane] = mine;
        cur = nxt;
    }
}

// the walk's results back where the scans read them: sval[bar] (units of u) for the stretch bars, blk_in[block] for the blocks
__global__ __launch_bounds__(256) void k_dlx_evout(const DlxEvent *__restrict__ ev, const double *__restrict__ out,
                                                   const int64_t *__restrict__ off, int64_t nblk, double inv_u,
                                                   int64_t *__restrict__ sval, int64_t *__restrict__ blk_in)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e == 0) blk_in[0] = 0;
    if (e >= off[nblk]) return;
    const int32_t fl = ev[e].flags, idx = ev[e].idx;
    const int64_t k = llrint(out[e] * inv_u);
    if (fl & DLX_EV_STRETCH) sval[idx] = k;
    else if (idx + 1 < nblk) blk_in[idx + 1] = k;
}

template <bool AF64>
__global__ __launch_bounds__(64) void k_dlx_resolve(const double *__restrict__ price, const void *__restrict__ amount,
                                                    int64_t n, double thr, double u, double inv_u,
                                                    const int64_t *__restrict__ ci, int64_t nb,
                                                    const DlxFn *__restrict__ fn, const int64_t *__restrict__ kin,
                                                    const int64_t *__restrict__ flist, int64_t n_flag,
                                                    int32_t *owner, DlxRes *res, unsigned long long *changed,
                                                    unsigned long long *giveup, const unsigned char *__restrict__ btype,
                                                    const int64_t *__restrict__ sval)
{
    // ONE WAVE per flagged bar (round 5; it was one thread: ~860 dependent global loads per replay, 0.7 ms per call at 1e9 ticks for
    // ~3 500 replays).  The loads are the wave's -- 64 ticks per step, the next step's already in flight -- and every lane runs the
    // reference's loop on the same values (readlane); lane 0 alone writes.
    const int64_t f = blockIdx.x;
    const int lane = threadIdx.x;
    if (f >= n_flag) return;
    const int64_t b = flist[f];
    DlxRes old = res[f];
    if (owner[b] != (int32_t)b) {
        // absorbed by the replay of an earlier bar: no state of its own.  Give back what it had claimed.
        if (old.mode == 1 && lane == 0) {
            for (int64_t x = b + 1; x < b + old.span && x <= nb; ++x)
                if (owner[x] == (int32_t)b) owner[x] = (int32_t)x;
            old.mode = 0; old.span = 1;
            res[f] = old;
            atomicAdd(changed, 1ULL);
        }
        return;
    }
    // the reference's loop (logic.py:141-147) from its true state at the start of bar b
    double c = b == 0 ? dlx_d<AF64>(price, amount, 0) : (double)kin[b] * u;
    DlxRes r;
    r.mode = 1; r.span = 1; r.nclose = 0; r.to_end = 0; r.kout = 0;
    for (int j = 0; j < DLX_MAXC; ++j) r.close[j] = -1;
    int q = 0;
    bool synced = false, lost = false;
    int64_t i = ci[b] + 1;
    double dnext = i + lane < n ? dlx_d<AF64>(price, amount, i + lane) : 0.0;
    for (bool done = false; i < n && !done; i += 64) {
        const double dl = dnext;
        if (i + 64 + lane < n) dnext = dlx_d<AF64>(price, amount, i + 64 + lane);
        const int lim = (int)(n - i < 64 ? n - i : 64);
        for (int j = 0; j < lim; ++j) {
            c += __longlong_as_double(fmk_readlane((int64_t)__double_as_longlong(dl), j));
            if (c >= thr) {
                if (q >= DLX_MAXC) { lost = true; done = true; break; }
                const int64_t at = i + j;
#pragma unroll
                for (int z = 0; z < DLX_MAXC; ++z) if (z == q) r.close[z] = at;
                ++q;
                c = c - thr;
                if (b + q <= nb && at == ci[b + q]) { synced = true; done = true; break; }     // back on a close of the closed form
            }
        }
    }
    if (lane != 0) return;
    if (lost) { atomicAdd(giveup, 1ULL); return; }
    r.nclose = q;
    r.kout = llrint(c * inv_u);
    if (synced) r.span = q;
    else { r.to_end = 1; r.span = (int32_t)(nb - b + 1); }                  // through the tail
    // what the bar's own function predicted: one bar, its closed-form close, f(k_in)
    bool same;
    if (old.mode == 0) {
        if (b == nb) same = r.to_end && q == 0;
        else same = synced && q == 1 && r.kout == (btype[b] ? sval[b] : dlx_eval(fn[b], kin[b]));   // (a stretch bar: the walk's output)
        if (same) return;
    } else {
        same = old.span == r.span && old.nclose == r.nclose && old.to_end == r.to_end && old.kout == r.kout;
        for (int j = 0; j < DLX_MAXC; ++j) same = same && old.close[j] == r.close[j];
    }
    const int64_t old_span = old.mode == 1 ? old.span : 1;
    bool touched = !same;
    for (int64_t x = b + 1; x < b + r.span && x <= nb; ++x)                // (re-)assert what this replay covers
        if (owner[x] != (int32_t)b) { owner[x] = (int32_t)b; touched = true; }
    for (int64_t x = b + r.span; x < b + old_span && x <= nb; ++x)
        if (owner[x] == (int32_t)b) { owner[x] = (int32_t)x; touched = true; }
    if (!same) res[f] = r;
    if (touched) atomicAdd(changed, 1ULL);
}

__global__ __launch_bounds__(64) void k_dlx_commit(const int64_t *__restrict__ flist, int64_t n_flag,
                                                   const int32_t *__restrict__ owner, const DlxRes *__restrict__ res,
                                                   int64_t *__restrict__ out, int64_t cap, int64_t *count_out)
{
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_flag) return;
    const int64_t b = flist[f];
    const DlxRes r = res[f];
    if (r.mode != 1 || owner[b] != (int32_t)b) return;
    for (int j = 0; j < r.nclose; ++j)
        if (b + 1 + j < cap) out[b + 1 + j] = r.close[j];
    if (r.to_end) *count_out = 1 + b + r.nclose;      // the leading 0, the closes before bar b, the closes of this replay
}

#define DLX_TRY(expr)                       \
    do {                                    \
        rc = (expr);                        \
        if (rc != FMK_OK) goto done;        \
    } while (0)
#define DLX_HIP(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e__ = (expr);                                                                               \
        if (e__ != hipSuccess) {                                                                               \
            rc = fmk_set_error(ctx, FMK_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            goto done;                                                                                         \
        }                                                                                                      \
    } while (0)

int fmk_dollar_exact(fmk_ctx *ctx, const double *d_price, const void *d_amount, int is_f64, int64_t n, double thr,
                     int64_t *d_close_idx, const int64_t *d_carry_k, int64_t *count, int *status, double extra_ticks, int whales)
{
    *status = 1;
    const int64_t nb = *count - 1;                   // closes of the closed form
    int ex;
    (void)frexp(thr, &ex);
    if (!(thr > 0.0) || !isfinite(thr) || ex < -900 || ex > 900 || nb < 0 || nb + 1 >= (int64_t)INT32_MAX) return FMK_OK;
    const double u = ldexp(1.0, ex - 53), inv_u = ldexp(1.0, 53 - ex);
    // test knob (read per call): FMK_DL_MARGIN_SCALE >= 1 widens the margin -> more bars are replayed (never fewer: exactness
    // does not depend on it)
    const char *mv = getenv("FMK_DL_MARGIN_SCALE");
    double mscale = mv ? atof(mv) : 1.0;
    if (!(mscale >= 1.0)) mscale = 1.0;
    // (m_rel: the reference's drift per add, + 1/128 for the carries of the one-pass closed form, whose fixed point truncates every
    //  product by less than ulp(thr) / 256: fmk_dollar_onepass.h)
    const double m_rel = ldexp(thr, -52) * mscale * (1.0 + 1.0 / 128.0), m_abs = 64.0 * u * mscale;   // the carry k~ is good to a few units
    const int64_t nbar = nb + 1;                     // + the tail
    const int64_t nblk = fmk_ceil_div(nbar, DLX_BLOCK_BARS);
    void *p_fn = nullptr, *p_fidx = nullptr, *p_owner = nullptr, *p_flist = nullptr, *p_kin = nullptr, *p_blk = nullptr,
         *p_blkin = nullptr, *p_res = nullptr, *p_cnt = nullptr, *p_btype = nullptr, *p_dlast = nullptr, *p_spre = nullptr,
         *p_sval = nullptr, *p_scount = nullptr, *p_off = nullptr, *p_ev = nullptr, *p_out = nullptr;
    const int64_t tu = llrint(thr * inv_u);          // thr in units of u: an integer in [2^52, 2^53)
    const double top = ldexp(1.0, ex + 1);           // thr in [2^(ex-1), 2^ex): closing sums from here on are whale closes
    int rc = FMK_OK;
    unsigned long long *cnt;
    int64_t n_flag = 0;
    int rounds = 0;
    int bpl = (int)(nbar / ((int64_t)128 * 8192));
    bpl = bpl < 1 ? 1 : (bpl > DLX_R_MAX ? DLX_R_MAX : bpl);
    const unsigned gb = (unsigned)fmk_ceil_div(nbar, (int64_t)128 * bpl);
    DLX_TRY(fmk_alloc(ctx, (size_t)nbar * sizeof(DlxFn), &p_fn));
    DLX_TRY(fmk_alloc(ctx, (size_t)nbar * 4, &p_fidx));
    DLX_TRY(fmk_alloc(ctx, (size_t)nbar * 4, &p_owner));
    DLX_TRY(fmk_alloc(ctx, (size_t)nbar * 8, &p_flist));
    DLX_TRY(fmk_alloc(ctx, (size_t)nbar * 8, &p_kin));
    DLX_TRY(fmk_alloc(ctx, (size_t)nblk * sizeof(DlxFn), &p_blk));
    DLX_TRY(fmk_alloc(ctx, (size_t)nblk * 8, &p_blkin));
    DLX_TRY(fmk_alloc(ctx, 64, &p_cnt));
    DLX_TRY(fmk_alloc(ctx, (size_t)nbar, &p_btype));
    DLX_TRY(fmk_alloc(ctx, (size_t)nbar * 8, &p_dlast));
    if (whales) {
        DLX_TRY(fmk_alloc(ctx, (size_t)nbar * sizeof(DlxFn), &p_spre));
        DLX_TRY(fmk_alloc(ctx, (size_t)nbar * 8, &p_sval));
        DLX_TRY(fmk_alloc(ctx, (size_t)nblk * 4, &p_scount));
        DLX_TRY(fmk_alloc(ctx, (size_t)(nblk + 1) * 8, &p_off));
        DLX_TRY(fmk_alloc(ctx, (size_t)(nbar + nblk) * sizeof(DlxEvent), &p_ev));
        DLX_TRY(fmk_alloc(ctx, (size_t)(nbar + nblk) * 8, &p_out));
    }
    cnt = (unsigned long long *)p_cnt;               // [0] flagged bars, [1] changed, [2] gave up, [3] new count
    DLX_HIP(hipMemsetAsync(cnt, 0, 64, ctx->stream));
    if (is_f64)
        k_dlx_bars<true><<<gb, 128, 0, ctx->stream>>>(d_price, d_amount, n, thr, u, inv_u, m_rel, m_abs, d_close_idx, nb, d_carry_k,
                                                      (DlxFn *)p_fn, (int32_t *)p_fidx, (int32_t *)p_owner, (int64_t *)p_flist, cnt, bpl,
                                                      whales, extra_ticks, tu, top, (unsigned char *)p_btype, (double *)p_dlast);
    else
        k_dlx_bars<false><<<gb, 128, 0, ctx->stream>>>(d_price, d_amount, n, thr, u, inv_u, m_rel, m_abs, d_close_idx, nb, d_carry_k,
                                                       (DlxFn *)p_fn, (int32_t *)p_fidx, (int32_t *)p_owner, (int64_t *)p_flist, cnt, bpl,
                                                       whales, extra_ticks, tu, top, (unsigned char *)p_btype, (double *)p_dlast);
    DLX_HIP(hipGetLastError());
    DLX_HIP(hipMemcpyAsync(ctx->h_mail, cnt, 24, hipMemcpyDeviceToHost, ctx->stream));
    DLX_HIP(hipStreamSynchronize(ctx->stream));
    n_flag = ctx->h_mail[0];
    if (ctx->h_mail[2] != 0) goto done;              // a backlog beyond 500 thresholds (or a bar the closed form and the states disagree about)
    if (n_flag == 0) { *status = 0; goto done; }     // no bar needs a replay: every decision of the closed form is certain
    DLX_TRY(fmk_alloc(ctx, (size_t)n_flag * sizeof(DlxRes), &p_res));
    DLX_HIP(hipMemsetAsync(p_res, 0, (size_t)n_flag * sizeof(DlxRes), ctx->stream));
    {
        const unsigned gf = (unsigned)fmk_ceil_div(n_flag, 64);
        for (rounds = 0; rounds < 64; ++rounds) {
            DLX_HIP(hipMemsetAsync(cnt + 1, 0, 8, ctx->stream));
            k_dlx_scan<<<(unsigned)nblk, 256, 0, ctx->stream>>>((const DlxFn *)p_fn, (const int32_t *)p_fidx, (const int32_t *)p_owner,
                                                                (const DlxRes *)p_res, nbar, 0, (DlxFn *)p_blk, nullptr, nullptr,
                                                                (const unsigned char *)p_btype, (DlxFn *)p_spre, nullptr, (int *)p_scount);
            if (whales) {
                k_dlx_evoff<<<1, 1024, 0, ctx->stream>>>((const int *)p_scount, nblk, (int64_t *)p_off);
                k_dlx_events<<<(unsigned)nblk, 256, 0, ctx->stream>>>((const DlxFn *)p_fn, (const int32_t *)p_fidx, (const int32_t *)p_owner,
                                                                      (const DlxRes *)p_res, (const unsigned char *)p_btype,
                                                                      (const DlxFn *)p_spre, (const double *)p_dlast, (const DlxFn *)p_blk,
                                                                      (const int64_t *)p_off, nbar, u, (DlxEvent *)p_ev);
                k_dlx_walk<<<1, 64, 0, ctx->stream>>>((const DlxEvent *)p_ev, (const int64_t *)p_off, nblk, thr, u, inv_u, (double *)p_out);
                k_dlx_evout<<<(unsigned)fmk_ceil_div(nbar + nblk, 256), 256, 0, ctx->stream>>>((const DlxEvent *)p_ev, (const double *)p_out,
                                                                                                (const int64_t *)p_off, nblk, inv_u,
                                                                                                (int64_t *)p_sval, (int64_t *)p_blkin);
            }
            else
                k_dlx_blocks<<<1, 64, 0, ctx->stream>>>((const DlxFn *)p_blk, nblk, (int64_t *)p_blkin);
            k_dlx_scan<<<(unsigned)nblk, 256, 0, ctx->stream>>>((const DlxFn *)p_fn, (const int32_t *)p_fidx, (const int32_t *)p_owner,
                                                                (const DlxRes *)p_res, nbar, 1, nullptr, (const int64_t *)p_blkin,
                                                                (int64_t *)p_kin, (const unsigned char *)p_btype, nullptr,
                                                                (const int64_t *)p_sval);
            if (is_f64)
                k_dlx_resolve<true><<<(unsigned)n_flag, 64, 0, ctx->stream>>>(d_price, d_amount, n, thr, u, inv_u, d_close_idx, nb, (const DlxFn *)p_fn,
                                                                (const int64_t *)p_kin, (const int64_t *)p_flist, n_flag,
                                                                (int32_t *)p_owner, (DlxRes *)p_res, cnt + 1, cnt + 2,
                                                                (const unsigned char *)p_btype, (const int64_t *)p_sval);
            else
                k_dlx_resolve<false><<<(unsigned)n_flag, 64, 0, ctx->stream>>>(d_price, d_amount, n, thr, u, inv_u, d_close_idx, nb, (const DlxFn *)p_fn,
                                                                 (const int64_t *)p_kin, (const int64_t *)p_flist, n_flag,
                                                                 (int32_t *)p_owner, (DlxRes *)p_res, cnt + 1, cnt + 2,
                                                                 (const unsigned char *)p_btype, (const int64_t *)p_sval);
            DLX_HIP(hipGetLastError());
            DLX_HIP(hipMemcpyAsync(ctx->h_mail, cnt + 1, 16, hipMemcpyDeviceToHost, ctx->stream));
            DLX_HIP(hipStreamSynchronize(ctx->stream));
            if (ctx->h_mail[1] != 0) goto done;      // a replay did not get back onto the closed form: serial walk
            if (ctx->h_mail[0] == 0) break;          // a quiet round: every state is the reference's
        }
        if (rounds >= 64) goto done;
        DLX_HIP(hipMemcpyAsync(cnt + 3, count, 8, hipMemcpyHostToDevice, ctx->stream));
        k_dlx_commit<<<gf, 64, 0, ctx->stream>>>((const int64_t *)p_flist, n_flag, (const int32_t *)p_owner, (const DlxRes *)p_res,
                                                 d_close_idx, *count + 16, (int64_t *)(cnt + 3));
        DLX_HIP(hipGetLastError());
        DLX_HIP(hipMemcpyAsync(ctx->h_mail, cnt + 3, 8, hipMemcpyDeviceToHost, ctx->stream));
        DLX_HIP(hipStreamSynchronize(ctx->stream));
        *count = ctx->h_mail[0];
        *status = 0;
    }
done:
    (void)n_flag;
    if (p_out) fmk_free(ctx, p_out);
    if (p_ev) fmk_free(ctx, p_ev);
    if (p_off) fmk_free(ctx, p_off);
    if (p_scount) fmk_free(ctx, p_scount);
    if (p_sval) fmk_free(ctx, p_sval);
    if (p_spre) fmk_free(ctx, p_spre);
    if (p_dlast) fmk_free(ctx, p_dlast);
    if (p_btype) fmk_free(ctx, p_btype);
    if (p_res) fmk_free(ctx, p_res);
    if (p_cnt) fmk_free(ctx, p_cnt);
    if (p_blkin) fmk_free(ctx, p_blkin);
    if (p_blk) fmk_free(ctx, p_blk);
    if (p_kin) fmk_free(ctx, p_kin);
    if (p_flist) fmk_free(ctx, p_flist);
    if (p_owner) fmk_free(ctx, p_owner);
    if (p_fidx) fmk_free(ctx, p_fidx);
    if (p_fn) fmk_free(ctx, p_fn);
    return rc;
}
