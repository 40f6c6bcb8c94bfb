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

#define DLX_MAXC 6                       // closes one replay may record before it has to be back on the closed form
#define DLX_CONST INT64_MIN              // DlxFn.base of a constant function
#define DLX_BLOCK_BARS 1024              // 256 threads x 4 bars

struct DlxFn {                           // f(k) = kap[(k - base) & 3] + ((k - base) & ~3), or the constant kap[0]
    int64_t base;
    int64_t kap[4];
};

struct DlxRes {                          // state of one flagged bar between rounds
    int32_t mode;                        // 0: the replay agrees with the bar's own function; 1: replaced by a constant
    int32_t span;                        // bars covered by the replay (1 unless a close moved across a bar boundary)
    int32_t nclose;                      // closes the replay recorded
    int32_t to_end;                      // the replay ran to the end of the stream
    int64_t kout;
    int64_t close[DLX_MAXC];
};

__device__ __forceinline__ int64_t dlx_eval(const DlxFn &f, int64_t k)
{
    if (f.base == DLX_CONST) return f.kap[0];
    const int64_t t = k - f.base, r = t & 3;
    return f.kap[r] + (t - r);
}

// first f, then g
__device__ __forceinline__ DlxFn dlx_compose(const DlxFn &f, const DlxFn &g)
{
    if (g.base == DLX_CONST) return g;
    DlxFn h;
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
__global__ __launch_bounds__(128) void k_dlx_bars(const double *__restrict__ price, const void *__restrict__ amount, int64_t n,
                                                  double thr, double u, double inv_u, double m_rel, double m_abs,
                                                  const int64_t *__restrict__ ci, int64_t nb,
                                                  const int64_t *__restrict__ carry_k, DlxFn *__restrict__ fn,
                                                  int32_t *__restrict__ fidx, int32_t *__restrict__ owner,
                                                  int64_t *__restrict__ flist, unsigned long long *__restrict__ n_flag,
                                                  int bars_per_lane)
{
    __shared__ double rows[2][64][DLX_T + 1];               // 2 waves x 16.9 KB
    __shared__ int64_t s_pos[2][64], s_end[2][64];
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * 2 + w;
    const int64_t bar0 = wave * (64 * (int64_t)bars_per_lane) + lane;
    int r = 0;
    int64_t b = bar0, pos = 0, end = -1, kb = 0;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, m = 0;
    bool tail = false, flag = false, active = false;
    auto open_bar = [&]() {                          // next bar of this lane, or none
        for (;;) {
            active = r < bars_per_lane && b <= nb;
            if (!active) return;
            tail = b == nb;
            pos = ci[b] + 1;
            end = tail ? n - 1 : ci[b + 1];
            // how far the reference's state can be from a simulated trajectory at the end of this bar: its drift against
            // exact arithmetic (3), the rounding of the exact carry to k~, the spread of the trajectories, their parity wiggle
            m = (double)(end + 2) * m_rel + m_abs;
            flag = false;
            kb = 0;
            if (b == 0) {
                c0 = c1 = c2 = c3 = dlx_d<AF64>(price, amount, 0);   // cum = prices[0] * volumes[0] (logic.py:140): exact
            } else {
                kb = carry_k[b] - 1;
                if (kb < 0) kb = 0;
                c0 = (double)kb * u; c1 = (double)(kb + 1) * u; c2 = (double)(kb + 2) * u; c3 = (double)(kb + 3) * u;
            }
            if (pos <= end) return;
            // an empty tail: nothing can close behind the last close
            DlxFn f;
            f.base = kb; f.kap[0] = kb; f.kap[1] = kb + 1; f.kap[2] = kb + 2; f.kap[3] = kb + 3;
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
#pragma unroll
        for (int sgm = 0; sgm < 64; sgm += SPI) {           // all loads of the round in flight together
            const int seg = sgm + half;
            const int64_t p0 = s_pos[w][seg], e0 = s_end[w][seg];
            const int64_t tick = p0 + j32;
            double v = 0.0;
            if (p0 >= 0 && tick <= e0) v = dlx_d<AF64>(price, amount, tick);
            rows[w][seg][j32] = v;
        }
        __builtin_amdgcn_wave_barrier();
        int cnt = 0;
        const int o = (int)(pos & (DLX_T - 1));                     // my first tick inside the window
        if (active) cnt = (int)(end - pos + 1 < DLX_T - o ? end - pos + 1 : DLX_T - o);
        const bool closes_here = active && !tail && pos + cnt - 1 == end;     // the closing add is the last tick of this round
#pragma unroll 4
        for (int j = 0; j < DLX_T; ++j) {
            if (j < cnt) {
                const double d = rows[w][lane][(o + j) & (DLX_T - 1)];
                c0 += d; c1 += d; c2 += d; c3 += d;
                const double lo = c0 - m, hi = c3 + m;
                bool bad = dlx_binade(lo) != dlx_binade(hi);     // the add must land in one binade for every possible state
                if (closes_here && j == cnt - 1) bad |= !(lo >= thr);   // the closing add reaches the threshold for every state
                else bad |= hi >= thr;                           // every other add stays below it
                flag |= bad;
            }
        }
        pos += cnt;
        const bool finished = active && pos > end;
        bool report = false;
        if (finished) {
            DlxFn f;
            f.base = kb;
            if (!tail) {
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
                                               const DlxRes *res)
{
    const int32_t o = owner[x];
    const int32_t f = fidx[o];
    if (f >= 0 && res && res[f].mode == 1) {         // the bar, or the bar whose replay absorbed it, is a constant
        DlxFn c;
        c.base = DLX_CONST;
        c.kap[0] = res[f].kout;
        c.kap[1] = c.kap[2] = c.kap[3] = 0;
        return c;
    }
    return fn[x];
}

// phase 0: blk[block] = composition of the block's 1024 bars.  phase 1: kin[x] = state at the start of bar x.
__global__ __launch_bounds__(256) void k_dlx_scan(const DlxFn *__restrict__ fn, const int32_t *__restrict__ fidx,
                                                  const int32_t *__restrict__ owner, const DlxRes *__restrict__ res,
                                                  int64_t nbar, int phase, DlxFn *__restrict__ blk,
                                                  const int64_t *__restrict__ blk_in, int64_t *__restrict__ kin)
{
    __shared__ DlxFn buf[2][256];
    const int t = threadIdx.x;
    const int64_t x0 = (int64_t)blockIdx.x * DLX_BLOCK_BARS + (int64_t)t * 4;
    DlxFn mine[4];
    DlxFn acc;
    acc.base = 0; acc.kap[0] = 0; acc.kap[1] = 1; acc.kap[2] = 2; acc.kap[3] = 3;      // identity
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (x0 + j < nbar) {
            mine[j] = dlx_effective(x0 + j, fn, fidx, owner, res);
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
    if (phase == 0) {
        if (t == 255) blk[blockIdx.x] = buf[cur][255];
        return;
    }
    int64_t k = blk_in[blockIdx.x];
    if (t > 0) k = dlx_eval(buf[cur][t - 1], k);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (x0 + j < nbar) {
            kin[x0 + j] = k;
            k = dlx_eval(mine[j], k);
        }
    }
}

// one wave: the lanes fetch 64 block aggregates, lane 0 applies them in order
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

template <bool AF64>
__global__ __launch_bounds__(64) void k_dlx_resolve(const double *__restrict__ price, const void *__restrict__ amount,
                                                    int64_t n, double thr, double u, double inv_u,
                                                    const int64_t *__restrict__ ci, int64_t nb,
                                                    const DlxFn *__restrict__ fn, const int64_t *__restrict__ kin,
                                                    const int64_t *__restrict__ flist, int64_t n_flag,
                                                    int32_t *owner, DlxRes *res, unsigned long long *changed,
                                                    unsigned long long *giveup)
{
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_flag) return;
    const int64_t b = flist[f];
    DlxRes old = res[f];
    if (owner[b] != (int32_t)b) {
        // absorbed by the replay of an earlier bar: no state of its own.  Give back what it had claimed.
        if (old.mode == 1) {
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
    for (; i < n; ++i) {
        c += dlx_d<AF64>(price, amount, i);
        if (c >= thr) {
            if (q >= DLX_MAXC) { lost = true; break; }
            r.close[q++] = i;
            c = c - thr;
            if (b + q <= nb && i == ci[b + q]) { synced = true; break; }     // back on a close of the closed form
        }
    }
    if (lost) { atomicAdd(giveup, 1ULL); return; }
    r.nclose = q;
    r.kout = llrint(c * inv_u);
    if (synced) r.span = q;
    else { r.to_end = 1; r.span = (int32_t)(nb - b + 1); }                  // through the tail
    // what the bar's own function predicted: one bar, its closed-form close, f(k_in)
    bool same;
    if (old.mode == 0) {
        if (b == nb) same = r.to_end && q == 0;
        else same = synced && q == 1 && r.kout == dlx_eval(fn[b], kin[b]);
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
                     int64_t *d_close_idx, const int64_t *d_carry_k, int64_t *count, int *status)
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
    const double m_rel = ldexp(thr, -52) * mscale, m_abs = 64.0 * u * mscale;   // the carry k~ is good to a few units
    const int64_t nbar = nb + 1;                     // + the tail
    const int64_t nblk = fmk_ceil_div(nbar, DLX_BLOCK_BARS);
    void *p_fn = nullptr, *p_fidx = nullptr, *p_owner = nullptr, *p_flist = nullptr, *p_kin = nullptr, *p_blk = nullptr,
         *p_blkin = nullptr, *p_res = nullptr, *p_cnt = nullptr;
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
    cnt = (unsigned long long *)p_cnt;               // [0] flagged bars, [1] changed, [2] gave up, [3] new count
    DLX_HIP(hipMemsetAsync(cnt, 0, 64, ctx->stream));
    if (is_f64)
        k_dlx_bars<true><<<gb, 128, 0, ctx->stream>>>(d_price, d_amount, n, thr, u, inv_u, m_rel, m_abs, d_close_idx, nb, d_carry_k,
                                                      (DlxFn *)p_fn, (int32_t *)p_fidx, (int32_t *)p_owner, (int64_t *)p_flist, cnt, bpl);
    else
        k_dlx_bars<false><<<gb, 128, 0, ctx->stream>>>(d_price, d_amount, n, thr, u, inv_u, m_rel, m_abs, d_close_idx, nb, d_carry_k,
                                                       (DlxFn *)p_fn, (int32_t *)p_fidx, (int32_t *)p_owner, (int64_t *)p_flist, cnt, bpl);
    DLX_HIP(hipGetLastError());
    DLX_HIP(hipMemcpyAsync(ctx->h_mail, cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
    DLX_HIP(hipStreamSynchronize(ctx->stream));
    n_flag = ctx->h_mail[0];
    if (n_flag == 0) { *status = 0; goto done; }     // no bar needs a replay: every decision of the closed form is certain
    DLX_TRY(fmk_alloc(ctx, (size_t)n_flag * sizeof(DlxRes), &p_res));
    DLX_HIP(hipMemsetAsync(p_res, 0, (size_t)n_flag * sizeof(DlxRes), ctx->stream));
    {
        const unsigned gf = (unsigned)fmk_ceil_div(n_flag, 64);
        for (rounds = 0; rounds < 64; ++rounds) {
            DLX_HIP(hipMemsetAsync(cnt + 1, 0, 8, ctx->stream));
            k_dlx_scan<<<(unsigned)nblk, 256, 0, ctx->stream>>>((const DlxFn *)p_fn, (const int32_t *)p_fidx, (const int32_t *)p_owner,
                                                                (const DlxRes *)p_res, nbar, 0, (DlxFn *)p_blk, nullptr, nullptr);
            k_dlx_blocks<<<1, 64, 0, ctx->stream>>>((const DlxFn *)p_blk, nblk, (int64_t *)p_blkin);
            k_dlx_scan<<<(unsigned)nblk, 256, 0, ctx->stream>>>((const DlxFn *)p_fn, (const int32_t *)p_fidx, (const int32_t *)p_owner,
                                                                (const DlxRes *)p_res, nbar, 1, nullptr, (const int64_t *)p_blkin,
                                                                (int64_t *)p_kin);
            if (is_f64)
                k_dlx_resolve<true><<<gf, 64, 0, ctx->stream>>>(d_price, d_amount, n, thr, u, inv_u, d_close_idx, nb, (const DlxFn *)p_fn,
                                                                (const int64_t *)p_kin, (const int64_t *)p_flist, n_flag,
                                                                (int32_t *)p_owner, (DlxRes *)p_res, cnt + 1, cnt + 2);
            else
                k_dlx_resolve<false><<<gf, 64, 0, ctx->stream>>>(d_price, d_amount, n, thr, u, inv_u, d_close_idx, nb, (const DlxFn *)p_fn,
                                                                 (const int64_t *)p_kin, (const int64_t *)p_flist, n_flag,
                                                                 (int32_t *)p_owner, (DlxRes *)p_res, cnt + 1, cnt + 2);
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
    {
        const char *vv = getenv("FMK_DL_VERBOSE");
        if (vv && atoi(vv))
            fprintf(stderr, "[fmk_dollar_exact] n=%lld bars=%lld flagged=%lld rounds=%d status=%d rc=%d\n", (long long)n,
                    (long long)nb, (long long)n_flag, rounds + 1, *status, rc);
    }
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
