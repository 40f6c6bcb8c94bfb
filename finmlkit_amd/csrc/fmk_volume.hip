// fmk_volume.hip -- _volume_bar_indexer (finmlkit/bar/logic.py:87-115), parallel jump tables.
//
// Reference recurrence (sequential):   cum += v_i ; if cum >= thr: close at i, cum = 0   (RESET).
// Because of the reset, the state after a close is only the POSITION of that close: the bar that starts
// after a close at tick j ends at nxt(j) = min{ j' > j : sum(v[j+1..j']) >= thr }, and the closes are
// the orbit c_1, nxt(c_1), nxt(nxt(c_1)), ...  -- a pointer chain of B links.  Chains started from
// different ticks do not merge, so the chain cannot be guessed; it CAN be composed hierarchically:
//
//   level 0  (k_vol_level0, one workgroup per block of S ticks, S >= longest bar):
//            prefix sums of the block + S ticks of look-ahead in LDS; nxt(j) for EVERY tick j of the
//            block by bisection in LDS; pointer doubling in LDS turns nxt into
//            E0[j] = first chain node >= end of the block, C0[j] = chain nodes inside the block.
//   level k  (k_vol_level_up): a block of S*2^k ticks is two blocks of level k-1.  The chain enters any
//            block within its first S ticks (S >= longest bar), so a table over those S entry ticks is
//            enough:  E^k[i] = E^{k-1}_right[ E^{k-1}_left[i] ],  C^k = C_left + C_right.   O(N/2^k) work.
//   top-down (k_vol_descend): the root block's entry is the first close c_1; every block hands its entry
//            to the left child and E_left[entry] (+ the close count so far) to the right child.
//   emit     (k_vol_emit): every level-0 block walks its <= S/bar-length chain nodes and writes them to
//            their final output slots.
// Total work O(N) + 2*log2(N/S) tiny launches; no sequential pass over the stream.
//
// Arithmetic: bar sums are differences of block-local float64 prefix sums (not the reference's
// sequential sum from the bar start).  Every decision gets a class (0 certain, 1 within (1e-11 + 2^-52 * length) * thr of
// the threshold, 2 exact tie -- fragile unless the sums are exact in float64).  Exact mode (the default): a fragile decision
// is settled by replaying its bar with the reference's sequential sum -- inline here, in a pass of its own in the global
// tier, on the chain only (k_vol_emit's list -> k_vol_verify) for what remains; n_uncertified comes back 0.  Fast mode:
// the classes are only counted, on the chain.  Tiers by mean bar length: these LDS tables (S = 2048), the global tables
// further down (to 64 K ticks), the chain walk (beyond).  Domain: thr > 0, v >= 0, N < 2^31 -- anything else falls back
// to the serial walk of fmk_threshold.hip.
#include <math.h>
#include <stdlib.h>

#include "fmk_common.h"

#define VOL_END 0xFFFFFFFFu
// threads per workgroup of k_vol_level0.  The workgroup's LDS (53.3 KB: three per CU) does not depend on it, so this sets
// the waves per CU: 256 -> 12 waves, 14.6 ms per 1e9 ticks at 865-tick bars; 512 -> 24 waves, 11.8 ms; 1024 -> 32 waves
// (two workgroups: the wave limit) but 19.6 ms -- barriers across 16 waves, and a thread's monotone walk covers 2 ticks
#define VOL_THREADS 512
#define VOL_RADIX 16                    // blocks composed per table level (LDS tier)

int fmk_threshold_serial(fmk_ctx *ctx, int dollar, const double *d_price, const void *d_amount, int is_f64, int64_t n,
                         double thr, int64_t *d_close_idx, int64_t capacity, int64_t *n_idx, int64_t *n_unc);

// status word bits
#define VOL_ST_OVERFLOW 1   // a bar is longer than S ticks
#define VOL_ST_BAD 2        // negative / NaN volume
#define VOL_ST_INEXACT 4    // an amount that is not a multiple of 2^-20 below 2^20: sums are not exact in float64

// A decision whose margin is EXACTLY zero (sum == threshold in this file's evaluation order) is certain only when every sum
// involved is exact in float64 -- then the reference's sequential sum is the same number.  That holds for streams of
// multiples of 2^-20 below 2^20 with a threshold below 2^31 (bars and 8192-tick prefixes stay below 2^33: 53 bits), e.g.
// the synthetic stream and integer lots.  For anything else (decimal lots: 100 x 0.1 is 9.99999999999998 summed in tick
// order, 10.0 in other orders) an exact tie is as fragile as a near tie.  The kernels that read the amounts set
// VOL_ST_INEXACT; ties are recorded as their own class (fragile byte 2) and count as fragile when the bit is set.
__device__ __forceinline__ bool vol_amount_inexact(double v) { return !(v * 1048576.0 == rint(v * 1048576.0) && v < 1048576.0); }
// The reference's loop for one bar (logic.py:107-113): cum += v[t] for t = from .. last in tick order, close at the first
// cum >= thr (not before tick `min_close`: tick 0 cannot close).  Returns the close tick or -1.  Eight amounts are loaded
// before they are added: with the exit test between a load and the next one, a lane had ONE load in flight and ran at
// ~230 ns per tick.
template <bool AF64>
__device__ __forceinline__ int64_t vol_replay(const void *__restrict__ amount, int64_t from, int64_t last, double cum, double thr,
                                              int64_t min_close)
{
    // Every lane replays its own bar, so a wave's load touches 64 different lines and the pass runs at the rate the CU
    // generates addresses (measured: 3.5e10 additions in 69 ms on 256 CUs = one lane-address per clock per CU).  Hence
    // 16-byte loads -- 2 doubles or 4 floats per address -- and sixteen amounts in flight before the first is added.
    constexpr int ELEM = AF64 ? 8 : 4;
    int64_t t = from;
    auto step = [&](double v, int64_t at) -> bool { cum += v; return cum >= thr && at >= min_close; };
    for (; t <= last && (((uintptr_t)amount + (uintptr_t)t * ELEM) & 15) != 0; ++t)
        if (step(fmk_amt<AF64>(amount, t), t)) return t;
    for (; t + 15 <= last; t += 16) {
        double d[16];
        if constexpr (AF64) {
            const double2 *p = (const double2 *)((const double *)amount + t);
#pragma unroll
            for (int k = 0; k < 8; ++k) { const double2 x = p[k]; d[2 * k] = x.x; d[2 * k + 1] = x.y; }
        } else {
            const float4 *p = (const float4 *)((const float *)amount + t);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 x = p[k];
                d[4 * k] = (double)x.x; d[4 * k + 1] = (double)x.y; d[4 * k + 2] = (double)x.z; d[4 * k + 3] = (double)x.w;
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (step(d[k], t + k)) return t + k;
    }
    for (; t <= last; ++t)
        if (step(fmk_amt<AF64>(amount, t), t)) return t;
    return -1;
}

// The decisions of the global-table and chain-walk tiers are sums of a double-double block part (exact) and block-LOCAL prefix sums
// (512 ticks, plain doubles).  A local prefix carries the rounding of its own scan, at most a few hundred ulps of ITS magnitude --
// which has nothing to do with the threshold's: behind a block trade of 7e11 the prefixes of the same block are good to ~1e-4,
// and a bar of small trades that starts there had its decision "certified" with the 1e-11 * thr margin, one tick late
// (tools/fuzz_volume.py seed 5202 case 241, round 5).  Every margin that involves a local prefix therefore adds 2^-42 of it.
__device__ __forceinline__ double vol_prefix_slack(double local_prefix) { return 2.2737367544323206e-13 * fabs(local_prefix); }

__device__ __forceinline__ void vol_flag(int *status, int bit)     // one atomic per kernel, not per wave
{
    if (!(__atomic_load_n(status, __ATOMIC_RELAXED) & bit)) atomicOr(status, bit);
}
// ... and the THRESHOLD has to be such a multiple too: the decisions compare a prefix with (earlier prefix + thr), and with
// thr = 2840.5000000000005 (one ulp above a multiple of 1/8; found by tools/fuzz_parity.py seed 778 case 121) that sum rounds
// to the grid -- a bar whose amounts add up to 2840.5 looked like an exact tie and closed, where the reference's
// `cum >= thr` is false.
__device__ __forceinline__ bool vol_thr_inexact(double thr) { return !(thr * 1048576.0 == rint(thr * 1048576.0) && thr < 2147483648.0); }
__device__ __forceinline__ bool vol_ties_fragile(const int *status, double thr)
{
    return (__atomic_load_n(status, __ATOMIC_RELAXED) & VOL_ST_INEXACT) || vol_thr_inexact(thr);
}

// `fragile` (one byte per tick, may be null): the decision nxt(j) was within the certification margin.  Only decisions ON
// THE CHAIN matter: their ordinals (decision q produces out[q]; q == count is the final "no further close") are appended
// to `list` ([0] = how many, then the ordinals) for k_vol_verify.
// capacity of the list (8 MB): each entry replays ONE bar, so the replay work is bounded by the stream length whatever the
// count; decimal lots with a round threshold tie on a sizeable share of their bars (tests: 1e6 ticks, thr 25 -> thousands)
#define VOL_LIST_CAP (1 << 20)
__device__ __forceinline__ void vol_list_append(int64_t *list, int64_t q)
{
    const unsigned long long pos = atomicAdd((unsigned long long *)list, 1ULL);
    if (pos < VOL_LIST_CAP) list[1 + pos] = q;
}

template <bool AF64, int S>
__global__ __launch_bounds__(VOL_THREADS) void k_vol_level0(const void *__restrict__ amount, int64_t n, double thr,
                                                            uint32_t *__restrict__ nxt, uint32_t *__restrict__ E0,
                                                            uint32_t *__restrict__ C0, uint32_t *__restrict__ root,
                                                            int *__restrict__ status, int *__restrict__ root_tie,
                                                            unsigned char *__restrict__ fragile, int64_t *__restrict__ list,
                                                            int replay)
{
    constexpr int PER = 2 * S / VOL_THREADS;          // prefix elements per thread
    constexpr int EPT = S / VOL_THREADS;              // table entries per thread
    // Lp[i] = sum of the first i ticks from the block start, stored PADDED (one slot per 8): thread t later probes
    // around element 8t + (bar length), i.e. with a lane stride of 8 doubles -- unpadded that is a 16-way bank conflict
#define LP(i) Lp[(i) + ((i) >> 3)]
    // dynamic LDS (S = 2048: 53.3 KB = exactly three workgroups per CU; S = 4096, 106 KB, was the tier for streams whose
    // longest bar exceeds 2048 ticks): [Lp | Eb | Cb | wtot]
    constexpr int LPN = 2 * S + 1 + (2 * S + 1) / 8 + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char vol_smem[];
    double *Lp = (double *)vol_smem;
    uint32_t *Eb = (uint32_t *)(vol_smem + (size_t)LPN * 8);
    uint32_t *Cb = Eb + S;
    double *wtot = (double *)(Cb + S);
    const int64_t bs = (int64_t)blockIdx.x * S;       // first tick of the block
    const int tid = threadIdx.x, lane = fmk_lane(), w = tid >> 6;
    // ---- block-local prefix sums over [bs, bs + 2S)
    double loc[PER];
    double run = 0.0;
    bool bad = false, inexact = false;
    {
        // the thread's PER consecutive amounts: ONE 16-byte load per four float32 (two float64) when the whole group lies inside
        // the stream (block starts are multiples of S, so the group is 16-byte aligned whenever the column is) -- eight scalar
        // loads with a 32-byte lane stride made every instruction touch 16 lines
        const int64_t jb = bs + (int64_t)tid * PER;
        double vv[PER];
        const bool al16 = ((uintptr_t)amount & 15) == 0;
        if (jb + PER <= n && al16) {
            if constexpr (AF64) {
                const double2 *q = (const double2 *)((const double *)amount + jb);
#pragma unroll
                for (int k = 0; k < PER / 2; ++k) { const double2 t2 = q[k]; vv[2 * k] = t2.x; vv[2 * k + 1] = t2.y; }
            } else {
                const float4 *q = (const float4 *)((const float *)amount + jb);
#pragma unroll
                for (int k = 0; k < PER / 4; ++k) {
                    const float4 t4 = q[k];
                    vv[4 * k] = (double)t4.x; vv[4 * k + 1] = (double)t4.y; vv[4 * k + 2] = (double)t4.z; vv[4 * k + 3] = (double)t4.w;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < PER; ++k) vv[k] = jb + k < n ? fmk_amt<AF64>(amount, jb + k) : 0.0;
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const double v = vv[k];
            if (jb + k < n) { bad |= !(v >= 0.0); inexact |= vol_amount_inexact(v); }
            run += v;
            loc[k] = run;
        }
    }
    double inc = fmk_wave_iscan(run);
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    double pre = __shfl_up(inc, 1, 64);               // exclusive prefix of the thread totals
    if (lane == 0) pre = 0.0;
    {
        double wp = 0.0;
        for (int q = 0; q < w; ++q) wp += wtot[q];
        pre = wp + pre;
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) LP(tid * PER + k + 1) = pre + loc[k];
    if (tid == 0) LP(0) = 0.0;
    if (__ballot(bad) != 0 && lane == 0) atomicOr(status, VOL_ST_BAD);
    if (__ballot(inexact) != 0 && lane == 0) vol_flag(status, VOL_ST_INEXACT);
    // are exact ties of THIS block's decisions certain?  They involve only the block's own 2S ticks
    const bool ties = __syncthreads_or(inexact ? 1 : 0) != 0 || vol_thr_inexact(thr);
    // Exact mode (replay != 0): a fragile decision is settled on the spot by the reference's own computation for that bar --
    // cum = 0, += v in tick order from the tick after j (logic.py:107-113) -- so the tables are built from links that are
    // either certain by their margin or computed exactly.  Work: (fragile ticks) x (bar length) additions spread over all
    // lanes; decimal lots with a round threshold tie on ~1/4 of their ticks, continuous amounts on ~1e-9 of them.
    auto replay_from = [&](int64_t j, double cum, bool *too_long) -> uint32_t {
        const int64_t lim = j + S < n - 1 ? j + S : n - 1;
        const int64_t m = vol_replay<AF64>(amount, j + 1, lim, cum, thr, 1);
        if (m >= 0) return (uint32_t)m;
        if (lim < n - 1) *too_long = true;                          // no close within S ticks although data remains
        return VOL_END;
    };
    // ---- nxt(j) for every tick of the block: smallest m > i+1 with Lp[m] - Lp[i+1] >= thr.
    //      Thread t owns the EPT CONSECUTIVE ticks i = t*EPT + q: nxt is non-decreasing in i, so after one bisection
    //      for its first tick the thread only walks forward (amortised ~1 probe per tick instead of log2(2S) = 12).
    const int64_t remain = n - bs;                                  // ticks available from the block start
    const int mmax = (int)(remain < 2 * S ? remain : 2 * S);        // Lp[0..mmax] are valid
    // The decisions are differences of block-local prefix sums over up to 2S ticks: their own rounding (<= 2S * 2^-53 of the
    // largest prefix, twice) must stay inside the margin even when that prefix dwarfs the threshold (one giant trade followed
    // by small ones: an ulp of the prefix can exceed 1e-11 * thr and a decision that differs from the reference's sequential
    // sum would be classed certain) -- hence the second term, 2^-40 = 2 * 4096 * 2^-53 of the block's last prefix
    const double tol = 1e-11 * thr + 9.094947017729282e-13 * fabs(LP(mmax));
    int carry_lo = 0;                                               // m of the previous tick of this thread
    bool ovf = false;
    for (int q = 0; q < EPT; ++q) {
        const int i = tid * EPT + q;                                // tick bs + i
        uint32_t nx = VOL_END, cc = 0;
        unsigned char frag = 0;                                      // 1: within the margin, 2: exact tie
        if (i < remain) {
            cc = 1;
            const double target = LP(i + 1) + thr;
            int lo = i + 2, hi = i + 1 + S;                          // close tick = bs + m - 1 in (j, j + S]
            if (hi > mmax) hi = mmax;
            if (lo <= hi && LP(hi) >= target) {
                if (q > 0 && carry_lo >= lo) {
                    // monotone: the answer is >= the previous tick's; walk forward from there
                    lo = carry_lo;
                    while (lo < hi && LP(lo) < target) ++lo;
                } else {
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (LP(mid) >= target) hi = mid; else lo = mid + 1;
                    }
                }
                carry_lo = lo;
                nx = (uint32_t)(bs + lo - 1);
                const double over = LP(lo) - target, under = target - LP(lo - 1);
                frag = ((over > 0.0 && over <= tol) || (lo - 1 > i + 1 && under <= tol)) ? 1 : (over == 0.0 ? 2 : 0);
            } else if (i + 1 + S <= mmax) {
                ovf = true;                                          // no close within S ticks although data remains
                carry_lo = 0;
            } else {
                if (hi >= lo) frag = target - LP(hi) <= tol ? 1 : 0;
                carry_lo = 0;
            }
        }
        Eb[i] = nx;
        Cb[i] = cc | ((uint32_t)frag << 30);                         // the class rides in the count until the copy below:
    }                                                                // S = 2048 is 53.3 KB, exactly three workgroups per CU
    if (__ballot(ovf) != 0 && lane == 0) atomicOr(status, VOL_ST_OVERFLOW);   // one atomic per wave, not per tick
    __syncthreads();
    // first bar (block 0): tick 0 is counted but cannot close -> first j >= 1 with P_j >= thr
    if (blockIdx.x == 0 && tid == 0) {
        uint32_t r = VOL_END;
        int lo = 2, hi = mmax < S ? mmax : S;           // keeps the root inside the first S ticks
        if (lo <= hi && LP(hi) >= thr) {
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (LP(mid) >= thr) hi = mid; else lo = mid + 1;
            }
            r = (uint32_t)(lo - 1);
            const double over = LP(lo) - thr, under = thr - LP(lo - 1);       // decision 1 (the first bar)
            const bool near = (over > 0.0 && over <= tol) || (lo - 1 >= 2 && under <= tol);
            if (replay) {
                if (near || (over == 0.0 && ties)) {                // cum = volumes[0], then += from tick 1 (logic.py:107)
                    bool too_long = false;
                    r = replay_from(0, fmk_amt<AF64>(amount, 0), &too_long);
                    if (too_long || (r != VOL_END && r >= (uint32_t)S)) atomicOr(status, VOL_ST_OVERFLOW);
                }
            } else if (near) vol_list_append(list, 1);
            else if (over == 0.0) *root_tie = 1;        // k_vol_emit lists it when the stream is not exactly summable
        } else if (S <= mmax) {
            atomicOr(status, VOL_ST_OVERFLOW);
        } else if (hi >= 1 && thr - LP(hi) <= tol) {
            // the whole (short) stream comes within the margin of one bar
            if (replay) { bool too_long = false; r = replay_from(0, fmk_amt<AF64>(amount, 0), &too_long); }
            else vol_list_append(list, 1);
        }
        *root = r;
    }
    __syncthreads();
    if (replay) {
        // ---- exact mode: the block's live fragile ticks, compacted (the prefix sums are dead now: their LDS holds the list)
        //      and dealt out evenly -- replaying where they are found left 4 of 5 lanes idle, because ~1/5 of a decimal
        //      stream's ticks are fragile but SOME lane of every wave is at every step (61 -> 18 ms per 2e8 ticks of tenth lots in 865-tick bars)
        int *lcount = (int *)Lp;
        unsigned short *llist = (unsigned short *)(lcount + 1);
        if (tid == 0) *lcount = 0;
        __syncthreads();
        for (int q = 0; q < EPT; ++q) {
            const int i = q * VOL_THREADS + tid;
            const uint32_t cls = Cb[i] >> 30;
            if (cls == 1 || (cls == 2 && ties)) llist[atomicAdd(lcount, 1)] = (unsigned short)i;
        }
        __syncthreads();
        const int nlist = *lcount;
        bool too_long = false;
        for (int it = tid; it < nlist; it += VOL_THREADS) {
            const int i = llist[it];
            Eb[i] = replay_from(bs + i, 0.0, &too_long);
            Cb[i] &= 0x3FFFFFFFu;
        }
        if (__ballot(too_long) != 0 && lane == 0) atomicOr(status, VOL_ST_OVERFLOW);
        __syncthreads();
    }
    for (int q = 0; q < EPT; ++q) {                                  // coalesced copy of the chain links
        const int i = q * VOL_THREADS + tid;
        nxt[bs + i] = Eb[i];
        const uint32_t cf = Cb[i];
        fragile[bs + i] = (unsigned char)(cf >> 30);
        Cb[i] = cf & 0x3FFFFFFFu;
    }
    __syncthreads();
    // ---- pointer doubling inside the block: E -> first node >= block end, C -> nodes inside the block
    const uint32_t bend = (uint32_t)(bs + S);
    for (;;) {
        uint32_t e2[EPT], c2[EPT];
        int changed = 0;
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int i = q * VOL_THREADS + tid;
            const uint32_t e = Eb[i];
            e2[q] = e;
            c2[q] = Cb[i];
            if (e != VOL_END && e < bend) {
                const int i2 = (int)(e - (uint32_t)bs);
                e2[q] = Eb[i2];
                c2[q] += Cb[i2];
                changed = 1;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int i = q * VOL_THREADS + tid;
            Eb[i] = e2[q];
            Cb[i] = c2[q];
        }
        if (!__syncthreads_or(changed)) break;
    }
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int i = q * VOL_THREADS + tid;
        E0[bs + i] = Eb[i];
        C0[bs + i] = Cb[i];
    }
}

// tables of level k from level k-1: one thread per (block, entry).  The table span S = 1 << ls is a run-time value: the
// LDS tiers use 2048 / 4096, the global tier (below) whatever power of two covers the longest bar.
__global__ __launch_bounds__(256) void k_vol_level_up(int ls, const uint32_t *__restrict__ Ep, const uint32_t *__restrict__ Cp,
                                                      int64_t nblk_prev, int64_t span_prev /* ticks per prev block */,
                                                      uint32_t *__restrict__ Ek, uint32_t *__restrict__ Ck,
                                                      int64_t nblk, int *__restrict__ status)
{
    const int64_t S = (int64_t)1 << ls;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblk * S) return;
    const int64_t b = t >> ls;
    const int64_t i = t & (S - 1);
    const int64_t left = 2 * b, right = 2 * b + 1;
    uint32_t x = Ep[left * S + i];
    uint32_t c = Cp[left * S + i];
    if (x != VOL_END && right < nblk_prev) {
        const int64_t i2 = (int64_t)x - right * span_prev;
        if (i2 < 0 || i2 >= S) { atomicOr(status, VOL_ST_OVERFLOW); x = VOL_END; }
        else {
            c += Cp[right * S + i2];
            x = Ep[right * S + i2];
        }
    }
    Ek[t] = x;
    Ck[t] = c;
}

// entries / output offsets of level k-1 from level k
__global__ __launch_bounds__(256) void k_vol_descend(int ls, const uint32_t *__restrict__ ent_k, const int64_t *__restrict__ off_k,
                                                     int64_t nblk_k, int64_t span_prev,
                                                     const uint32_t *__restrict__ Ep, const uint32_t *__restrict__ Cp,
                                                     int64_t nblk_prev, uint32_t *__restrict__ ent_p,
                                                     int64_t *__restrict__ off_p)
{
    const int64_t S = (int64_t)1 << ls;
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk_k) return;
    const uint32_t e = ent_k[b];
    const int64_t o = off_k[b];
    const int64_t left = 2 * b, right = 2 * b + 1;
    ent_p[left] = e;
    off_p[left] = o;
    if (right < nblk_prev) {
        uint32_t x = VOL_END;
        int64_t oo = o;
        if (e != VOL_END) {
            const int64_t i = (int64_t)e - left * span_prev;       // entry lies in the left child's first S ticks
            if (i >= 0 && i < S) {
                x = Ep[left * S + i];
                oo = o + Cp[left * S + i];
            } else {
                x = e;                                              // (cannot happen when S >= longest bar)
            }
        }
        ent_p[right] = x;
        off_p[right] = oo;
    }
}

// Radix-4 flavours (round 2): level q+1 composes FOUR blocks of level q.  Composing two at a time writes and re-reads a
// table of N / 2^k entries at every one of the ~20 levels (8 B read + 8 B gathered + 8 B written per entry: 24 GB per 1e9
// ticks, 4.2 of the indexer's 12.1 ms); four at a time writes N / 4^q entries with three dependent gathers each:
// 13 GB, half the launches, and a third fewer tables to keep.  Measured at 1e9 ticks (cfg-3 threshold): radix 2 / 4 / 8 / 16 / 32 ->
// 11.6 / 9.5 / 8.9 / 8.7 / 8.6 ms for the whole indexer; VOL_RADIX = 16 (the kernels take the radix as an argument).
__global__ __launch_bounds__(256) void k_vol_level_up4(int ls, const uint32_t *__restrict__ Ep, const uint32_t *__restrict__ Cp,
                                                       int64_t nblk_prev, int64_t span_prev, uint32_t *__restrict__ Ek,
                                                       uint32_t *__restrict__ Ck, int64_t nblk, int *__restrict__ status,
                                                       int radix)
{
    const int64_t S = (int64_t)1 << ls;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nblk * S) return;
    const int64_t b = t >> ls;
    const int64_t i = t & (S - 1);
    const int64_t c0 = (int64_t)radix * b;
    uint32_t x = Ep[c0 * S + i];
    uint32_t c = Cp[c0 * S + i];
    for (int j = 1; j < radix; ++j) {
        const int64_t child = c0 + j;
        if (x == VOL_END || child >= nblk_prev) break;
        const int64_t i2 = (int64_t)x - child * span_prev;           // the chain enters the next child in its first S ticks
        if (i2 < 0 || i2 >= S) { atomicOr(status, VOL_ST_OVERFLOW); x = VOL_END; break; }
        c += Cp[child * S + i2];
        x = Ep[child * S + i2];
    }
    Ek[t] = x;
    Ck[t] = c;
}

__global__ __launch_bounds__(256) void k_vol_descend4(int ls, const uint32_t *__restrict__ ent_k, const int64_t *__restrict__ off_k,
                                                      int64_t nblk_k, int64_t span_prev, const uint32_t *__restrict__ Ep,
                                                      const uint32_t *__restrict__ Cp, int64_t nblk_prev,
                                                      uint32_t *__restrict__ ent_p, int64_t *__restrict__ off_p, int radix)
{
    const int64_t S = (int64_t)1 << ls;
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk_k) return;
    uint32_t e = ent_k[b];
    int64_t o = off_k[b];
    const int64_t c0 = (int64_t)radix * b;
    ent_p[c0] = e;
    off_p[c0] = o;
    for (int j = 1; j < radix; ++j) {
        const int64_t child = c0 + j;
        if (child >= nblk_prev) break;
        if (e != VOL_END) {
            const int64_t i = (int64_t)e - (child - 1) * span_prev;  // the entry lies in the previous child's first S ticks
            if (i >= 0 && i < S) {
                o += Cp[(child - 1) * S + i];
                e = Ep[(child - 1) * S + i];
            }                                                        // (else: cannot happen when S >= longest bar)
        }
        ent_p[child] = e;
        off_p[child] = o;
    }
}

__global__ __launch_bounds__(256) void k_vol_emit(int ls, const uint32_t *__restrict__ ent0, const int64_t *__restrict__ off0,
                                                  int64_t nblk0, const uint32_t *__restrict__ nxt,
                                                  int64_t *__restrict__ out, int64_t cap,
                                                  const unsigned char *__restrict__ fragile, int64_t *__restrict__ list,
                                                  const int *__restrict__ status, const int *__restrict__ root_tie, double thr)
{
    const uint64_t S = (uint64_t)1 << ls;
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ties = vol_ties_fragile(status, thr);
    if (b == 0 && cap > 0) out[0] = 0;                              // logic.py:104
    if (b == 0 && root_tie && *root_tie && ties) vol_list_append(list, 1);
    if (b >= nblk0) return;
    uint32_t j = ent0[b];
    int64_t o = off0[b];
    const uint64_t bend = (uint64_t)(b + 1) * S;
    while (j != VOL_END && (uint64_t)j < bend) {
        if (o < cap) out[o] = (int64_t)j;
        ++o;
        const unsigned char f = fragile[j];
        if (f == 1 || (f == 2 && ties)) vol_list_append(list, o);
        j = nxt[j];
    }
}

struct VolCache {
    fmk_ctx *ctx;
    const void *amount;
    int64_t n;
    double thr;
    int is_f64;
    int64_t count, unc;
    int64_t *dbuf;
    int64_t cap;
    void *work;
    size_t work_bytes;
    void *work2, *work3;          // global tier: chain links + fragile bytes, tables
    size_t work2_bytes, work3_bytes;
    int64_t *d_list;              // [1 + VOL_LIST_CAP] fragile decisions on the chain
};
static VolCache &vol_cache(fmk_ctx *ctx)     // one per context (slot 0), created on first use
{
    if (!ctx->idx_cache[0]) ctx->idx_cache[0] = new VolCache();
    return *(VolCache *)ctx->idx_cache[0];
}

void fmk_volume_trim(fmk_ctx *ctx)
{
    VolCache *c = (VolCache *)ctx->idx_cache[0];
    if (!c) return;
    if (c->dbuf) (void)hipFree(c->dbuf);
    if (c->work) (void)hipFree(c->work);
    if (c->work2) (void)hipFree(c->work2);
    if (c->work3) (void)hipFree(c->work3);
    if (c->d_list) (void)hipFree(c->d_list);
    delete c;
    ctx->idx_cache[0] = nullptr;
}

static int vol_certify(fmk_ctx *ctx, const void *a, int is_f64, int64_t n, double thr, VolCache &c);

#include "fmk_volume_exact.h"     // the exact-sum tier (round 4): k_vx_level0, k_vx_emit, vx_run

// returns FMK_OK, 1 (the next tier), 2 (negative / NaN amounts), 3 (a replayed decision disagrees: serial walk) or an error
template <bool AF64, int S>
static int vol_run(fmk_ctx *ctx, const void *a, int64_t n, double thr, VolCache &c)
{
    constexpr int LS = 11;
    static_assert(S == (1 << LS), "table span");
    const int64_t nblk0 = fmk_ceil_div(n, S);
    // levels: nblk[k] = ceil(nblk0 / R^k) until 1 (radix-R composition, k_vol_level_up4)
    const int RAD = VOL_RADIX;
    int64_t nblk[64], spanq[64];
    int K = 0;
    nblk[0] = nblk0;
    spanq[0] = S;
    while (nblk[K] > 1) { nblk[K + 1] = (nblk[K] + RAD - 1) / RAD; spanq[K + 1] = spanq[K] * RAD; ++K; }
    // workspace layout (uint32 tables + per-level entry/offset arrays)
    size_t tbl = 0;
    for (int k = 0; k <= K; ++k) tbl += (size_t)nblk[k] * S;
    size_t ents = 0;
    for (int k = 0; k <= K; ++k) ents += (size_t)nblk[k];
    const size_t bytes = ((size_t)nblk0 * S + 2 * tbl) * 4 + ents * (4 + 8) + 256 + (size_t)nblk0 * S;
    if (!c.d_list) FMK_HIP(ctx, hipMalloc((void **)&c.d_list, ((size_t)1 + VOL_LIST_CAP) * 8));
    FMK_HIP(ctx, hipMemsetAsync(c.d_list, 0, 8, ctx->stream));
    if (c.work_bytes < bytes) {
        if (c.work) FMK_HIP(ctx, hipFree(c.work));
        c.work = nullptr; c.work_bytes = 0;
        FMK_HIP(ctx, hipMalloc(&c.work, bytes));
        c.work_bytes = bytes;
    }
    uint32_t *nxt = (uint32_t *)c.work;
    uint32_t *Eall = nxt + (size_t)nblk0 * S;
    uint32_t *Call = Eall + tbl;
    int64_t *offall = (int64_t *)(Call + tbl);
    uint32_t *entall = (uint32_t *)(offall + ents);
    unsigned char *fragile = (unsigned char *)(entall + ents);
    uint32_t *E[64], *C[64], *ent[64];
    int64_t *off[64];
    {
        size_t to = 0, eo = 0;
        for (int k = 0; k <= K; ++k) {
            E[k] = Eall + to; C[k] = Call + to; to += (size_t)nblk[k] * S;
            ent[k] = entall + eo; off[k] = offall + eo; eo += (size_t)nblk[k];
        }
    }
    int *d_status = (int *)(ctx->d_mail + 32);
    uint32_t *d_root = (uint32_t *)(ctx->d_mail + 33);
    int *d_root_tie = (int *)(ctx->d_mail + 35);
    FMK_HIP(ctx, hipMemsetAsync(ctx->d_mail + 32, 0, 32, ctx->stream));
    {
        constexpr size_t lds = (size_t)(2 * S + 1 + (2 * S + 1) / 8 + 1) * 8 + (size_t)S * 8 + 64;
        if (lds > 64 * 1024)
            FMK_HIP(ctx, hipFuncSetAttribute((const void *)k_vol_level0<AF64, S>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)lds));
        k_vol_level0<AF64, S><<<(unsigned)nblk0, VOL_THREADS, lds, ctx->stream>>>(a, n, thr, nxt, E[0], C[0], d_root,
                                                                                  d_status, d_root_tie, fragile, c.d_list,
                                                                                  ctx->fast_threshold ? 0 : 1);
    }
    FMK_LAUNCH_CHECK(ctx);
    for (int k = 1; k <= K; ++k) {
        const int64_t tot = nblk[k] * S;
        k_vol_level_up4<<<(unsigned)fmk_ceil_div(tot, 256), 256, 0, ctx->stream>>>(
            LS, E[k - 1], C[k - 1], nblk[k - 1], spanq[k - 1], E[k], C[k], nblk[k], d_status, RAD);
        FMK_LAUNCH_CHECK(ctx);
    }
    // root entry + total count
    FMK_HIP(ctx, hipMemcpyAsync(ctx->h_mail, ctx->d_mail + 32, 24, hipMemcpyDeviceToHost, ctx->stream));
    FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int status = (int)(ctx->h_mail[0] & 0xFFFFFFFF);
    const uint32_t root = (uint32_t)(ctx->h_mail[1] & 0xFFFFFFFFu);
    if (status & VOL_ST_BAD) return 2;              // negative / NaN volumes: only the serial walk reproduces those
    if (status & VOL_ST_OVERFLOW) return 1;
    int64_t closes = 0;
    if (root != VOL_END) {
        if ((int64_t)root >= S) return 1;
        uint32_t cnt = 0;
        FMK_HIP(ctx, hipMemcpyAsync(&cnt, C[K] + root, 4, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        closes = cnt;
    }
    c.count = closes + 1;
    if (c.dbuf && c.cap < c.count) { FMK_HIP(ctx, hipFree(c.dbuf)); c.dbuf = nullptr; }
    if (!c.dbuf) { FMK_HIP(ctx, hipMalloc((void **)&c.dbuf, (size_t)c.count * 8)); c.cap = c.count; }
    // top-down
    const int64_t one = 1;
    FMK_HIP(ctx, hipMemcpyAsync(ent[K], &root, 4, hipMemcpyHostToDevice, ctx->stream));
    FMK_HIP(ctx, hipMemcpyAsync(off[K], &one, 8, hipMemcpyHostToDevice, ctx->stream));
    for (int k = K; k >= 1; --k) {
        k_vol_descend4<<<(unsigned)fmk_ceil_div(nblk[k], 256), 256, 0, ctx->stream>>>(
            LS, ent[k], off[k], nblk[k], spanq[k - 1], E[k - 1], C[k - 1], nblk[k - 1], ent[k - 1], off[k - 1], RAD);
        FMK_LAUNCH_CHECK(ctx);
    }
    k_vol_emit<<<(unsigned)fmk_ceil_div(nblk0, 256), 256, 0, ctx->stream>>>(LS, ent[0], off[0], nblk0, nxt, c.dbuf, c.cap,
                                                                            fragile, c.d_list, d_status, d_root_tie, thr);
    FMK_LAUNCH_CHECK(ctx);
    return vol_certify(ctx, a, AF64 ? 1 : 0, n, thr, c);
}

// ---------------------------------------------------------------------------------------
// Long bars (> 4096 ticks): the jump tables cannot span them, but long bars are FEW (<= N/4096), so the chain is simply
// walked -- with the whole wave searching for each next close:
//   k_vc_prefix   : inclusive prefix sums Lp[j] inside 512-tick blocks (float64) + block totals, one wave per block
//   k_vc_wgtotals, k_vc_scan, k_vc_expand : exclusive double-double scan of the block totals -> Bb[0..nblk]
//                   (serial scan over N/2048 workgroup totals, expanded to the N/512 blocks in parallel)
//   k_vc_chase    : ONE wave; per close: 64 block totals (requested one close ahead) locate the block of the crossing,
//                   8 coalesced loads of that block's 512 prefixes locate the tick.  sum(c+1..m) = (Bb[blk(m)] -
//                   Bb[blk(c)]) + (Lp[m] - Lp[c]).  ~1.6 us per close; the cost model behind the design is measured
//                   (tools/hoplat.py: a dependent load instruction of a lone wave costs 600-770 cycles, and several in a
//                   row do not overlap) -- see the kernel's comment.  Build with -DVC_TIMING for cycles per phase.
// Decisions within (1e-11 + 2^-52 * bar length) * thr of the threshold are counted as uncertified, as in the tables.
// ---------------------------------------------------------------------------------------
#define VC_BLOCK 512                    // ticks per prefix block: 64 lanes x 8 consecutive ticks resolve a crossing in ONE round
#define VC_WG_TICKS 2048                // ticks per workgroup of k_vc_prefix (4 waves x 64 lanes x 8): one block per wave
struct VcDD { double hi, lo; };
__device__ __forceinline__ VcDD vc_two_sum(double a, double b) { double s = a + b, bb = s - a; return VcDD{s, (a - (s - bb)) + (b - bb)}; }
__device__ __forceinline__ VcDD vc_add(VcDD x, VcDD y)
{
    VcDD s = vc_two_sum(x.hi, y.hi);
    s.lo += x.lo + y.lo;
    const double h = s.hi + s.lo;
    return VcDD{h, s.lo - (h - s.hi)};
}
__device__ __forceinline__ double vc_diff(VcDD a, VcDD b)      // a - b rounded to double
{
    VcDD d = vc_two_sum(a.hi, -b.hi);
    return d.hi + (d.lo + (a.lo - b.lo));
}

// one wave per 512-tick block; Lp = inclusive prefix inside the block.  The block is taken as 8 ROWS of 64 ticks -- row k is
// ticks 64k .. 64k + 63, lane l holds tick 64k + l -- so every load and store is coalesced; each row gets a wave scan and the
// rows chain through a running carry.  (Lanes owning 8 consecutive ticks made every load / store instruction touch 64
// lines: 5.4 ms per 1e9 ticks for 12 GB.)
template <bool AF64>
__global__ __launch_bounds__(256) void k_vc_prefix(const void *__restrict__ amount, int64_t n, double *__restrict__ Lp,
                                                   double *__restrict__ totals, int *__restrict__ status)
{
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
    const int64_t blk = (int64_t)blockIdx.x * (VC_WG_TICKS / VC_BLOCK) + w;
    const int64_t bs = blk * VC_BLOCK;
    if (bs >= n) return;
    double v[8];
    bool bad = false, inexact = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int64_t j = bs + 64 * k + lane;
        v[k] = j < n ? fmk_amt<AF64>(amount, j) : 0.0;
        bad |= !(v[k] >= 0.0);
        inexact |= vol_amount_inexact(v[k]);
    }
    if (__ballot(bad) != 0 && lane == 0) atomicOr(status, VOL_ST_BAD);   // prefix sums must not decrease: serial walk instead
    if (__ballot(inexact) != 0 && lane == 0) vol_flag(status, VOL_ST_INEXACT);
    double carry = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double inc = carry + fmk_wave_iscan(v[k]);
        const int64_t j = bs + 64 * k + lane;
        if (j < n) Lp[j] = inc;
        carry = __shfl(inc, 63, 64);
    }
    if (lane == 0) totals[blk] = carry;
}

// Bb[k] = sum of totals[0..k) in double-double, k = 0..m (one block, 8 records per thread and round)
__global__ __launch_bounds__(256) void k_vc_scan(const double *__restrict__ totals, int64_t m, VcDD *__restrict__ Bb)
{
    __shared__ VcDD lds[4];
    __shared__ VcDD run_s;
    if (threadIdx.x == 0) run_s = VcDD{0.0, 0.0};
    __syncthreads();
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
    for (int64_t b = 0; b < m; b += 2048) {
        const int64_t i0 = b + (int64_t)threadIdx.x * 8;
        VcDD loc[8], s = VcDD{0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            loc[k] = s;
            if (i0 + k < m) s = vc_add(s, VcDD{totals[i0 + k], 0.0});
        }
        VcDD inc = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const VcDD o = VcDD{__shfl_up(inc.hi, d, 64), __shfl_up(inc.lo, d, 64)};
            if (lane >= d) inc = vc_add(o, inc);
        }
        if (lane == 63) lds[w] = inc;
        __syncthreads();
        VcDD pre = run_s;
        for (int q = 0; q < w; ++q) pre = vc_add(pre, lds[q]);
        VcDD prev = VcDD{__shfl_up(inc.hi, 1, 64), __shfl_up(inc.lo, 1, 64)};
        if (lane == 0) prev = VcDD{0.0, 0.0};
        const VcDD base = vc_add(pre, prev);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (i0 + k < m) Bb[i0 + k] = vc_add(base, loc[k]);
        __syncthreads();
        if (threadIdx.x == 255) run_s = vc_add(pre, inc);
        __syncthreads();
    }
    if (threadIdx.x == 0) Bb[m] = run_s;
}

// Bb[4w + i] = BW[w] + sum of the first i block totals of workgroup w (double-double): the serial scan runs over the
// N/2048 workgroup totals only, this expands it to the N/512 prefix blocks in parallel
__global__ __launch_bounds__(256) void k_vc_expand(const double *__restrict__ totals, const VcDD *__restrict__ BW, int64_t nwg,
                                                   int64_t nblk, VcDD *__restrict__ Bb)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w > nwg) return;
    if (w == nwg) { Bb[nblk] = BW[nwg]; return; }
    VcDD run = BW[w];
    constexpr int R = VC_WG_TICKS / VC_BLOCK;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int64_t b = w * R + i;
        if (b < nblk) { Bb[b] = run; run = vc_add(run, VcDD{totals[b], 0.0}); }
    }
}

// workgroup totals from the block totals (4 per workgroup)
__global__ __launch_bounds__(256) void k_vc_wgtotals(const double *__restrict__ totals, int64_t nblk, int64_t nwg,
                                                     double *__restrict__ wg)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwg) return;
    constexpr int R = VC_WG_TICKS / VC_BLOCK;
    double t = 0.0;                       // <= 2048 amounts: the same float64 sum the 2048-tick version formed per workgroup
#pragma unroll
    for (int i = 0; i < R; ++i)
        if (w * R + i < nblk) t += totals[w * R + i];
    wg[w] = t;
}

// value of lane `src` (wave-uniform index) for all lanes: v_readlane, no LDS round trip (__shfl compiles to ds_bpermute)
__device__ __forceinline__ double vc_lane(double v, int src)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)((unsigned long long)b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__global__ __launch_bounds__(64) void k_vc_chase(const double *__restrict__ Lp, const VcDD *__restrict__ Bb, int64_t n,
                                                 int64_t nblk, double thr, int64_t *__restrict__ closes, int64_t cap,
                                                 int64_t *__restrict__ result /* [0] count */, int64_t *__restrict__ list,
                                                 const int *__restrict__ status)
{
    // One wave, bound by its own dependent instruction chain, not by memory (measured with a cycle counter per phase at
    // 5000-tick bars: ~3900 cycles per close whether the crossing was located by 64 + 32 scattered probes, by 8 loads of 8
    // consecutive ticks per lane, or by 8 coalesced loads, cold or warmed by a helper wave; the ISA showed why: every
    // __shfl with a run-time lane is two ds_bpermute + wait, ~100 cycles each for a lone wave).  Hence:
    //   * all broadcasts use v_readlane: their source lanes come from ballots and are wave-uniform;
    //   * the in-block round is 8 COALESCED loads (instruction k: lane l reads tick 64k + l of the block); prefix sums do
    //     not decrease (v >= 0), so the crossing is the first (k, lane) in that order;
    //   * the block totals for the NEXT close are requested as soon as this close's block is known (their addresses only
    //     need bstar), so that round overlaps the in-block round;
    //   * values the wave already holds are carried, never re-loaded or recomputed from sums (Lp[c], Bb[blk(c)],
    //     Bb[bstar], Lp[close - 1]): every comparison sees the operands a plain re-load would see.
    const int lane = fmk_lane();
    int64_t c = -1, cnt = 0;
    const bool ties = vol_ties_fragile(status, thr);
    if (cap > 0 && lane == 0) closes[0] = 0;
    cnt = 1;                                                       // the opening entry (logic.py:104)
    int64_t bc = 0;
    VcDD Bc = Bb[0];
    double Lc = 0.0;
    bool long_bars = false;
    // window of block totals starting at block `wb`: lane l holds Bb[wb + l + 1]
    int64_t wb = 0;
    VcDD win = (wb + lane < nblk) ? Bb[wb + lane + 1] : VcDD{0.0, 0.0};
#ifdef VC_TIMING
    long long tA = 0, tB = 0, tC = 0, t0 = 0, t1 = 0, iters = 0;
#define VC_T(x) x = __builtin_readcyclecounter()
#else
#define VC_T(x)
#endif
    for (;;) {
        VC_T(t0);
        // ---- block of the crossing: first b >= bc with sum(c+1 .. end of b) >= thr.  `win` covers [wb, wb + 64), wb == bc
        int64_t bstar = -1;
        VcDD Bs = Bc, Bfirst = Bc;                                 // Bfirst = Bb[b0]
        int64_t bfrom = bc;
        if (long_bars) {
            // bars of more than 64 blocks: one gather 64 blocks apart finds the group of 64 that holds the crossing
            // (one load instruction per 4096 blocks instead of one per 64)
            for (;;) {
                int64_t pb = bfrom + 64 * (int64_t)(lane + 1);     // group l ends before block pb
                if (pb > nblk) pb = nblk;
                const VcDD x = Bb[pb];
                const uint64_t mg = __ballot(vc_diff(x, Bc) - Lc >= thr);
                if (mg) {
                    const int fg = __ffsll((unsigned long long)mg) - 1;
                    if (fg > 0) {
                        Bfirst = VcDD{vc_lane(x.hi, fg - 1), vc_lane(x.lo, fg - 1)};       // Bb[bfrom + 64 * fg]
                        bfrom += 64 * (int64_t)fg;
                    }
                    break;
                }
                if (bfrom + 64 * 64 >= nblk) { bfrom = nblk; break; }                       // nothing left crosses
                Bfirst = VcDD{vc_lane(x.hi, 63), vc_lane(x.lo, 63)};
                bfrom += 64 * 64;
            }
        }
        for (int64_t b0 = bfrom; b0 < nblk && bstar < 0; b0 += 64) {
            const int64_t b = b0 + lane;
            VcDD nb = win;
            if (b0 != wb) nb = (b < nblk) ? Bb[b + 1] : VcDD{0.0, 0.0};
            const bool hit = b < nblk && vc_diff(nb, Bc) - Lc >= thr;
            const uint64_t m = __ballot(hit);
            if (m) {
                const int f0 = __ffsll((unsigned long long)m) - 1;
                bstar = b0 + f0;
                const int src = f0 > 0 ? f0 - 1 : 0;
                const VcDD prev = VcDD{vc_lane(nb.hi, src), vc_lane(nb.lo, src)};
                Bs = f0 > 0 ? prev : Bfirst;                       // Bb[bstar]
            } else {
                Bfirst = VcDD{vc_lane(nb.hi, 63), vc_lane(nb.lo, 63)};            // Bb[b0 + 64]
            }
        }
        if (bstar < 0) {
            // the remaining ticks do not fill a bar -- decision `cnt`, fragile when they come within the margin
            const double rest = vc_diff(Bb[nblk], Bc) - Lc;
            if (lane == 0 && thr - rest <= (1e-11 + 2.3e-16 * (double)(n - 1 - c)) * thr + vol_prefix_slack(Lc)) vol_list_append(list, cnt);
            break;
        }
        // request the next close's window now: it starts at bstar whatever tick of the block closes
        wb = bstar;
        win = (wb + lane < nblk) ? Bb[wb + lane + 1] : VcDD{0.0, 0.0};
#ifdef VC_TIMING
        VC_T(t1); tA += t1 - t0; t0 = t1; ++iters;
#endif
        const double off = vc_diff(Bs, Bc) - Lc;                   // sum(c+1 .. m) = off + Lp[m] for m in block bstar
        const int64_t bstart = bstar * VC_BLOCK;
        const int64_t lo = bstar == bc ? (c + 1 > 1 ? c + 1 : 1) : bstart;                  // tick 0 cannot close
        const int64_t hi = bstart + VC_BLOCK - 1 < n - 1 ? bstart + VC_BLOCK - 1 : n - 1;
        // ---- ONE round inside the block, 8 coalesced loads
        double lv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t j = bstart + k * 64 + lane;
            lv[k] = j <= hi ? Lp[j] : 0.0;
        }
        // first (k, lane) whose prefix crosses: 8 ballots, all scalar work
        int kk = -1, g = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t j = bstart + k * 64 + lane;
            const uint64_t mk = __ballot(j >= lo && j <= hi && off + lv[k] >= thr);
            if (kk < 0 && mk) { kk = k; g = __ffsll((unsigned long long)mk) - 1; }
        }
        int64_t mclose;
        if (kk < 0) {
            // the block was chosen because its END crosses (double-double test); plain doubles may round it just below
            mclose = hi;
            kk = (int)((hi - bstart) >> 6);
            g = (int)((hi - bstart) & 63);
        } else {
            mclose = bstart + (int64_t)kk * 64 + g;
        }
        // Lp[mclose] and Lp[mclose - 1]: lane g (or g - 1) of load kk, or lane 63 of load kk - 1 -- kk is wave-uniform
        double cur = lv[0], prv = 0.0;
#pragma unroll
        for (int k = 1; k < 8; ++k)
            if (k == kk) { cur = lv[k]; prv = lv[k - 1]; }
        const double l_at = vc_lane(cur, g);
        const double l_before = g > 0 ? vc_lane(cur, g - 1) : vc_lane(prv, 63);
        const double s_at = off + l_at;
#ifdef VC_TIMING
        VC_T(t1); tB += t1 - t0; t0 = t1;
#endif
        // ---- certification
        const double tol = (1e-11 + 2.3e-16 * (double)(mclose - c)) * thr + vol_prefix_slack(Lc) + vol_prefix_slack(l_at);
        const double over = s_at - thr;
        double under = INFINITY;
        if (mclose - 1 >= lo) under = thr - (off + l_before);       // mclose - 1 >= lo >= block start: inside this block
        else if (mclose - 1 > c && mclose - 1 >= 1) under = thr - off;   // first tick of its block: the sum up to the block before
        // over < 0: the close forced onto the block's last tick above (the double-double block test says "crosses", the plain
        // sum stays below -- tools/fuzz_volume.py seed 97015 case 2997: a threshold equal to the correctly rounded total)
        if (lane == 0 && (over < 0.0 || (over > 0.0 && over <= tol) || under <= tol || (ties && over == 0.0)))
            vol_list_append(list, cnt);
        if (cnt < cap && lane == 0) closes[cnt] = mclose;
        ++cnt;
        long_bars = bstar - bc >= 64;      // beyond one 64-block window; the next bar is probably as long as this one
        c = mclose;
        bc = bstar;
        Bc = Bs;
        Lc = l_at;
#ifdef VC_TIMING
        VC_T(t1); tC += t1 - t0;
#endif
    }
#ifdef VC_TIMING
    if (lane == 0) printf("vc_chase timing: closes %lld  cycles/close: block search %.0f  in-block %.0f  certify+store %.0f\n",
                          iters, (double)tA / iters, (double)tB / iters, (double)tC / iters);
#endif
    if (lane == 0) result[0] = cnt;
}

// total_only: stop after the prefix / scan and return the stream's total volume through *total (used to decide whether
// the 4096-tick tables are worth trying before the chain walk)
template <bool AF64>
static int vol_chase(fmk_ctx *ctx, const void *a, int64_t n, double thr, VolCache &c, bool total_only, double *total,
                     bool have_prefix = false, int64_t sample = 0)
{
    // sample > 0: prefix + total over the first `sample` ticks only (tier estimate); have_prefix: the full prefix of an
    // earlier total_only call in this same entry-point call is still in c.work
    if (sample > 0 && sample < n) n = sample;
    const int64_t nblk = fmk_ceil_div(n, VC_BLOCK);
    const size_t lp_bytes = ((size_t)n * 8 + 255) & ~(size_t)255;
    const size_t tot_bytes = ((size_t)nblk * 8 + 255) & ~(size_t)255;
    const size_t bb_bytes = ((size_t)(nblk + 1) * sizeof(VcDD) + 255) & ~(size_t)255;
    const int64_t nwg_all = fmk_ceil_div(n, VC_WG_TICKS);
    const size_t wg_bytes = ((size_t)nwg_all * 8 + 255) & ~(size_t)255;
    const size_t bw_bytes = ((size_t)(nwg_all + 1) * sizeof(VcDD) + 255) & ~(size_t)255;
    const size_t bytes = lp_bytes + tot_bytes + bb_bytes + wg_bytes + bw_bytes + 256;
    if (c.work_bytes < bytes) {
        if (c.work) FMK_HIP(ctx, hipFree(c.work));
        c.work = nullptr; c.work_bytes = 0;
        FMK_HIP(ctx, hipMalloc(&c.work, bytes));
        c.work_bytes = bytes;
    }
    double *Lp = (double *)c.work;
    double *totals = (double *)((char *)c.work + lp_bytes);
    VcDD *Bb = (VcDD *)((char *)c.work + lp_bytes + tot_bytes);
    double *wgt = (double *)((char *)c.work + lp_bytes + tot_bytes + bb_bytes);
    VcDD *BW = (VcDD *)((char *)c.work + lp_bytes + tot_bytes + bb_bytes + wg_bytes);
    int64_t *d_res = ctx->d_mail + 44;
    int *d_bad = (int *)(ctx->d_mail + 40);
    if (!have_prefix) {
        FMK_HIP(ctx, hipMemsetAsync(d_bad, 0, 8, ctx->stream));
        k_vc_prefix<AF64><<<(unsigned)fmk_ceil_div(n, VC_WG_TICKS), 256, 0, ctx->stream>>>(a, n, Lp, totals, d_bad);
        FMK_LAUNCH_CHECK(ctx);
        // serial double-double scan over the N/2048 workgroup totals, expanded to the N/512 blocks in parallel
        const int64_t nwg = fmk_ceil_div(n, VC_WG_TICKS);
        k_vc_wgtotals<<<(unsigned)fmk_ceil_div(nwg, 256), 256, 0, ctx->stream>>>(totals, nblk, nwg, wgt);
        k_vc_scan<<<1, 256, 0, ctx->stream>>>(wgt, nwg, BW);
        k_vc_expand<<<(unsigned)fmk_ceil_div(nwg + 1, 256), 256, 0, ctx->stream>>>(totals, BW, nwg, nblk, Bb);
        FMK_LAUNCH_CHECK(ctx);
    }
    if (total_only) {
        FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[6], &Bb[nblk].hi, 8, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[5], d_bad, 8, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(total, &ctx->h_mail[6], 8);
        if (ctx->h_mail[5] & VOL_ST_BAD) return 2;      // negative / NaN amounts: only the serial walk reproduces those
        return FMK_OK;
    }
    int64_t cap = c.dbuf ? c.cap : 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (!c.dbuf) {
            cap = n / 4096 + 1024;
            if (attempt == 1) cap = c.count;
            FMK_HIP(ctx, hipMalloc((void **)&c.dbuf, (size_t)cap * 8));
            c.cap = cap;
        }
        if (!c.d_list) FMK_HIP(ctx, hipMalloc((void **)&c.d_list, ((size_t)1 + VOL_LIST_CAP) * 8));
        FMK_HIP(ctx, hipMemsetAsync(c.d_list, 0, 8, ctx->stream));
        k_vc_chase<<<1, 64, 0, ctx->stream>>>(Lp, Bb, n, nblk, thr, c.dbuf, c.cap, d_res, c.d_list, d_bad);
        FMK_LAUNCH_CHECK(ctx);
        FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[6], d_res, 8, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[5], d_bad, 8, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->h_mail[5] & VOL_ST_BAD) return 2;
        c.count = ctx->h_mail[6];
        if (c.count <= c.cap) return vol_certify(ctx, a, AF64 ? 1 : 0, n, thr, c);
        FMK_HIP(ctx, hipFree(c.dbuf));                              // more closes than expected: exact size, once more
        c.dbuf = nullptr;
    }
    return fmk_set_error(ctx, FMK_E_HIP, "volume chase: capacity");
}

// ---------------------------------------------------------------------------------------
// Bars of ~3 000 .. ~60 000 ticks: GLOBAL jump tables.  The LDS tables stop at 4096 ticks and the chain walk pays ~1.4 us
// per close (450 ms at 3 500-tick bars); in between, the same composition runs on tables in HBM:
//   k_vg_nxt     nxt(j) for EVERY tick from the chase's prefix sums (Lp inside 512-tick blocks + double-double block
//                bases Bb).  nxt is monotone in j, so the 512 source ticks of a wave land in a narrow destination window:
//                the wave finds each tick's destination BLOCK from 64 block bases in LDS (one coalesced load per 32 768
//                ticks of look-ahead), then stages each needed destination block (8 coalesced loads) in LDS and every lane
//                searches it for its ticks (gallop + bisection, continuing from its previous tick's answer).
//                ~N x (8 + 8 + 4) bytes of traffic, no dependent global chain.
//   k_vg_level0  table span S = power of two >= the longest bar FROM ANY TICK (measured by k_vg_nxt).  Every tick follows
//                the read-only links until they leave its S-block: S / bar length hops, 1-3 when S is chosen that tight.
//   levels up, descent, emit: the kernels of the LDS tiers with S at run time.
// Decisions within the certification margin are recorded per tick (one byte); k_vol_emit lists those ON THE CHAIN and
// k_vol_verify replays exactly those bars with the reference's sequential float64 sum.
// ---------------------------------------------------------------------------------------
#define VG_DEND 0x7FFFFFFF              // destination block of a tick whose bar never closes
// destination prefixes in LDS, padded one slot per 8 (a lane's probes sit ~8 doubles from its neighbour's: unpadded that is
// an 8-way bank conflict); entries past the block's valid ticks hold +inf ("crossed"), 8 more slots for the local probe
#define VG_SLP(i) s_lp[(i) + ((i) >> 3)]
#define VG_SLP_DOUBLES (VC_BLOCK + 8 + (VC_BLOCK + 8) / 8 + 1)

// first index in [0, 512] whose prefix crosses (512: none).  Prefixes do not decrease, so "crossed" is false..false
// true..true: three rounds of 8 INDEPENDENT reads (strides 64, 8, 1) instead of 9 dependent bisection steps -- no loop,
// no divergence, and nothing to go wrong on garbage input.
__device__ __forceinline__ int vg_search(const double *s_lp, double b, double l0, double thr)
{
    int seg = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) seg += (b + (VG_SLP(64 * i + 63) - l0) >= thr) ? 0 : 1;
    if (seg == 8) return VC_BLOCK;
    const int b1 = 64 * seg;
    int c2 = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) c2 += (b + (VG_SLP(b1 + 8 * i + 7) - l0) >= thr) ? 0 : 1;
    const int b2 = b1 + 8 * (c2 < 7 ? c2 : 7);
    int c3 = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) c3 += (b + (VG_SLP(b2 + i) - l0) >= thr) ? 0 : 1;
    return b2 + c3;
}

__global__ __launch_bounds__(256, 4) void k_vg_nxt(const double *__restrict__ Lp, const VcDD *__restrict__ Bb, int64_t n,
                                                int64_t nblk, double thr, uint32_t *__restrict__ nxt,
                                                unsigned char *__restrict__ fragile, unsigned *__restrict__ maxlen,
                                                const int *__restrict__ status, unsigned long long *__restrict__ n_fragile)
{
    __shared__ double s_end_all[4][64];
    __shared__ double s_lp_all[4][VG_SLP_DOUBLES];
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
    double *s_end = s_end_all[w], *s_lp = s_lp_all[w];
    const int64_t blk = (int64_t)blockIdx.x * 4 + w;
    const int64_t bs = blk * VC_BLOCK;
    if (bs >= n) return;                                          // waves are independent: wave barriers only
    if (lane < 8) VG_SLP(VC_BLOCK + lane) = INFINITY;
    const int64_t j0 = bs + (int64_t)lane * 8;                    // lane l owns ticks j0 .. j0 + 7
    double lpj[8], dB[8];
    int dblk[8];
    // (own prefixes: read blocked, 64 bytes per lane.  Loading them coalesced and handing them over through the padded tile,
    // which paid off in the streaming passes of fmk_dollar.hip, was measured here and is slower -- 21.6 -> 23.2 ms: the lines
    // were just written by k_vc_prefix, and the kernel is bound by its LDS searches, not by this load.)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        lpj[k] = j0 + k < n ? Lp[j0 + k] : 0.0;
        dblk[k] = j0 + k < n ? -1 : VG_DEND;
        dB[k] = 0.0;
    }
    const VcDD base = Bb[blk];
    unsigned char frag[8];                                        // 1: within the margin, 2: exact tie
#pragma unroll
    for (int k = 0; k < 8; ++k) frag[k] = 0;
    // ---- destination block of every tick: first d >= blk with sum(j+1 .. end of d) >= thr
    double prev_end = 0.0;                                        // sum(block start of blk .. start of block b0)
    for (int64_t r = 0;; ++r) {
        const int64_t b0 = blk + 64 * r;
        const int nv = (int)(nblk - b0 < 64 ? nblk - b0 : 64);
        s_end[lane] = lane < nv ? vc_diff(Bb[b0 + lane + 1], base) : -INFINITY;
        __builtin_amdgcn_wave_barrier();
        const double last = s_end[nv - 1];
        const bool more = b0 + 64 < nblk;
        bool pending = false;
        int lprev = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (dblk[k] != -1) continue;
            if (last - lpj[k] >= thr) {
                int lo = lprev, hi = nv - 1;
                if (s_end[lo] - lpj[k] >= thr) hi = lo;             // the usual case after the lane's first tick: same block
                else ++lo;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (s_end[mid] - lpj[k] >= thr) hi = mid; else lo = mid + 1;
                }
                dblk[k] = (int)(64 * r) + lo;
                dB[k] = lo > 0 ? s_end[lo - 1] : prev_end;
                lprev = lo;
            } else if (more) {
                pending = true;
            } else {
                // no further close: fragile if the rest of the stream comes within the margin of the threshold
                dblk[k] = VG_DEND;
                const double tol = (1e-11 + 2.3e-16 * (double)(n - 1 - (j0 + k))) * thr + vol_prefix_slack(lpj[k]);
                frag[k] = thr - (last - lpj[k]) <= tol ? 1 : 0;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (!more || __ballot(pending) == 0) break;
        prev_end = last;
    }
    // ---- inside the destination blocks, in ascending order, only those some tick needs
    uint32_t res[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) res[k] = VOL_END;
    int64_t wmax = 0;
    int cur = -1;
    for (;;) {
        int mine = VG_DEND;
#pragma unroll
        for (int k = 7; k >= 0; --k)
            if (dblk[k] > cur && dblk[k] < mine) mine = dblk[k];
        cur = (int)fmk_wave_min((int64_t)mine);
        if (cur == VG_DEND) break;
        const int64_t ds = (blk + cur) * VC_BLOCK;
        const int nvt = (int)(n - ds < VC_BLOCK ? n - ds : VC_BLOCK);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int t = 64 * q + lane;
            VG_SLP(t) = t < nvt ? Lp[ds + t] : INFINITY;
        }
        __builtin_amdgcn_wave_barrier();
        int iprev = 0;
        bool have = false;                                         // a tick of this lane already answered in this block
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (dblk[k] != cur) continue;
            const int64_t j = j0 + k;
            const double b = dB[k], l0 = lpj[k];
            int lo = cur == 0 ? lane * 8 + k + 1 : 0;              // m > j
            if (iprev > lo) lo = iprev;                            // nxt does not decrease
            int64_t m;
            // one round of 8 reads from the previous tick's answer settles most ticks (nxt moves ~1 per tick); the lane's
            // first tick of this block, and jumps past a large trade, take the three-round search
            int hi;
            {
                int c = 8;
                if (have) {
                    c = 0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) c += (b + (VG_SLP(lo + i) - l0) >= thr) ? 0 : 1;
                }
                hi = lo + c;
                if (c == 8) { hi = vg_search(s_lp, b, l0, thr); if (hi < lo) hi = lo; }
            }
            if (hi < nvt) {
                iprev = hi;
                m = ds + hi;
                const double tol = (1e-11 + 2.3e-16 * (double)(m - j)) * thr + vol_prefix_slack(l0) + vol_prefix_slack(VG_SLP(hi));
                const double over = b + (VG_SLP(hi) - l0) - thr;
                double under = INFINITY;
                if (m - 1 > j) under = thr - (hi > 0 ? b + (VG_SLP(hi - 1) - l0) : b - l0);
                frag[k] = ((over > 0.0 && over <= tol) || under <= tol) ? 1 : (over == 0.0 ? 2 : 0);
            } else {
                // the block was chosen because its END crosses (double-double); plain doubles rounded it just below
                m = ds + nvt < n ? ds + nvt : (int64_t)VOL_END;
                frag[k] = 1;
                iprev = nvt;
            }
            have = true;
            res[k] = (uint32_t)m;
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (res[k] != VOL_END && (int64_t)res[k] - (j0 + k) > wmax) wmax = (int64_t)res[k] - (j0 + k);
    if (j0 + 7 < n) {
        uint4 *o = (uint4 *)(nxt + j0);
        o[0] = make_uint4(res[0], res[1], res[2], res[3]);
        o[1] = make_uint4(res[4], res[5], res[6], res[7]);
        unsigned long long fb = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) fb |= (unsigned long long)frag[k] << (8 * k);
        *(unsigned long long *)(fragile + j0) = fb;
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (j0 + k < n) { nxt[j0 + k] = res[k]; fragile[j0 + k] = frag[k]; }
    }
    wmax = fmk_wave_max(wmax);
    if (lane == 0 && (unsigned)wmax > __atomic_load_n(maxlen, __ATOMIC_RELAXED)) atomicMax(maxlen, (unsigned)wmax);
    if ((blk & 63) == 0) {                                         // every 64th block: an estimate of the live fragile ticks
        const bool ties = vol_ties_fragile(status, thr);
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) cnt += (frag[k] == 1 || (frag[k] == 2 && ties)) ? 1 : 0;
        cnt = (int)fmk_wave_sum(cnt);
        if (lane == 0 && cnt) atomicAdd(n_fragile, (unsigned long long)cnt);
    }
}

// first close: tick 0 is counted but cannot close -> first m >= 1 with sum(0 .. m) >= thr (logic.py:104-108)
__global__ __launch_bounds__(64) void k_vg_root(const double *__restrict__ Lp, const VcDD *__restrict__ Bb, int64_t n,
                                                int64_t nblk, double thr, uint32_t *__restrict__ root,
                                                int64_t *__restrict__ list, const int *__restrict__ status)
{
    const int lane = fmk_lane();
    const bool ties = vol_ties_fragile(status, thr);
    int64_t bstar = -1;
    for (int64_t b0 = 0; b0 < nblk && bstar < 0; b0 += 64) {
        const int64_t b = b0 + lane;
        bool hit = false;
        if (b < nblk) { const VcDD x = Bb[b + 1]; hit = x.hi + x.lo >= thr; }
        const uint64_t mk = __ballot(hit);
        if (mk) bstar = b0 + __ffsll((unsigned long long)mk) - 1;
    }
    if (bstar < 0) {
        const VcDD t = Bb[nblk];
        if (lane == 0) {
            *root = VOL_END;
            if (thr - (t.hi + t.lo) <= (1e-11 + 2.3e-16 * (double)n) * thr) vol_list_append(list, 1);
        }
        return;
    }
    const VcDD bb = Bb[bstar];
    const double off = bb.hi + bb.lo;
    const int64_t bstart = bstar * VC_BLOCK;
    const int64_t lo = bstart > 1 ? bstart : 1;
    const int64_t hi = bstart + VC_BLOCK - 1 < n - 1 ? bstart + VC_BLOCK - 1 : n - 1;
    int64_t m = -1;
    for (int k = 0; k < 8 && m < 0; ++k) {
        const int64_t j = bstart + k * 64 + lane;
        const uint64_t mk = __ballot(j >= lo && j <= hi && off + Lp[j <= hi ? j : hi] >= thr);
        if (mk) m = bstart + k * 64 + __ffsll((unsigned long long)mk) - 1;
    }
    if (lane != 0) return;
    bool fr = false;
    if (m < 0) {
        if (hi >= lo) { m = hi; fr = true; }                       // rounded just below at the block's end
        else { *root = n > 1 ? 1u : VOL_END; if (n > 1) vol_list_append(list, 1); return; }   // one-tick block 0: let the replay decide
    } else {
        const double tol = (1e-11 + 2.3e-16 * (double)m) * thr + vol_prefix_slack(Lp[m]);
        const double over = off + Lp[m] - thr;
        double under = INFINITY;
        if (m - 1 >= lo) under = thr - (off + Lp[m - 1]);
        else if (m - 1 >= 1) under = thr - off;                    // m is the first tick of its block
        fr = (over > 0.0 && over <= tol) || under <= tol || (ties && over == 0.0);
    }
    *root = (uint32_t)m;
    if (fr) vol_list_append(list, 1);
}

__global__ __launch_bounds__(256) void k_vg_level0(const uint32_t *__restrict__ nxt, int64_t n, int ls, int64_t total,
                                                   uint32_t *__restrict__ E0, uint32_t *__restrict__ C0)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= total) return;
    if (j >= n) { E0[j] = VOL_END; C0[j] = 0; return; }
    const uint64_t bend = (uint64_t)((j >> ls) + 1) << ls;
    uint32_t e = nxt[j], c = 1;
    while (e != VOL_END && (uint64_t)e < bend) { e = nxt[e]; ++c; }
    E0[j] = e;
    C0[j] = c;
}

// Replay of the listed decisions with the reference's own arithmetic (logic.py:104-113): cum = 0 after a close, += v in
// tick order, close at the first cum >= thr.  Decision q starts after out[q - 1] (q == 1: at tick 0, which cannot close)
// and must end at out[q] (q == count: nowhere).
template <bool AF64>
__global__ __launch_bounds__(64) void k_vol_verify(const void *__restrict__ amount, int64_t n, double thr,
                                                   const int64_t *__restrict__ out, int64_t count,
                                                   const int64_t *__restrict__ list, int *__restrict__ mismatch)
{
    const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n_items = list[0] < VOL_LIST_CAP ? list[0] : VOL_LIST_CAP;
    if (item >= n_items) return;
    const int64_t q = list[1 + item];
    const int64_t start = q <= 1 ? 0 : out[q - 1] + 1;
    const int64_t expect = q < count ? out[q] : -1;
    const int64_t stop = expect >= 0 ? expect : n - 1;              // the replay may stop once it has passed the expected close
    const int64_t m = vol_replay<AF64>(amount, start, stop, 0.0, thr, 1);
    if (m != expect) atomicOr(mismatch, 1);
}

// Global tier, exact mode: every tick whose decision is fragile gets its link from the reference's own computation for
// the bar that starts after it (cum = 0, += v in tick order, logic.py:107-113), BEFORE the tables are built -- the tables
// then are exact by construction (k_vol_level0 does the same inline).  Thread 0 settles a listed first-bar decision too.
// Work: (fragile ticks) x (bar length) additions side by side, measured ~7e-13 s each (tools/certbench.py decimal: tenth
// lots tie on ~1/5 of their ticks); the host runs this pass only while that beats the serial walk's 19 ns per tick.
// Two cheaper-looking schemes were built and measured first, both on the chain only: (1) replay the fragile decisions of
// the emitted chain, patch the disagreeing links, rebuild -- each new stretch of chain has its own fragile decisions, half of
// which change again, and a shifted chain needs ~10 bars to re-join: 819, 763, 771, 769 ... listed decisions per round, a
// critical branching process; (2) let the thread that finds a disagreement follow the exact chain until it re-joins the
// emitted one -- converges in one round, but the emitted chain is wrong at ~10 % of its bars, so the exact chain is almost
// never on it and the thread walks the stream: 3.3 s for 2e7 ticks.
template <bool AF64>
__global__ __launch_bounds__(256) void k_vg_replay(const void *__restrict__ amount, int64_t n, double thr,
                                                   uint32_t *__restrict__ nxt, unsigned char *__restrict__ fragile,
                                                   const int *__restrict__ status, unsigned *__restrict__ maxlen,
                                                   uint32_t *__restrict__ root, int64_t *__restrict__ list, int only_root)
{
    // a block takes 2048 ticks: their live fragile ticks are compacted into LDS and dealt out evenly (see k_vol_level0:
    // replaying them where they sit leaves most lanes idle)
    __shared__ int lcount;
    __shared__ unsigned short llist[2048];
    const int tid = threadIdx.x;
    const int64_t bs = (int64_t)blockIdx.x * 2048;
    if (blockIdx.x == 0 && tid == 0 && list[0] > 0) {                // decision 1 was listed by k_vg_root
        const int64_t r = vol_replay<AF64>(amount, 1, n - 1, fmk_amt<AF64>(amount, 0), thr, 1);
        const uint32_t m = r < 0 ? VOL_END : (uint32_t)r;
        *root = m;
        list[0] = 0;
        if (m != VOL_END && m > __atomic_load_n(maxlen, __ATOMIC_RELAXED)) atomicMax(maxlen, m);
    }
    if (only_root) return;
    if (tid == 0) lcount = 0;
    __syncthreads();
    const bool ties = vol_ties_fragile(status, thr);
    const int64_t j0 = bs + (int64_t)tid * 8;
    if (j0 + 7 < n) {
        const unsigned long long fb = *(const unsigned long long *)(fragile + j0);
        if (fb != 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned f = (unsigned)(fb >> (8 * k)) & 0xFF;
                if (f == 1 || (f == 2 && ties)) llist[atomicAdd(&lcount, 1)] = (unsigned short)(tid * 8 + k);
            }
        }
    } else {
        for (int k = 0; k < 8 && j0 + k < n; ++k) {
            const unsigned f = fragile[j0 + k];
            if (f == 1 || (f == 2 && ties)) llist[atomicAdd(&lcount, 1)] = (unsigned short)(tid * 8 + k);
        }
    }
    __syncthreads();
    const int nlist = lcount;
    for (int it = tid; it < nlist; it += 256) {
        const int64_t j = bs + llist[it];
        const int64_t r = vol_replay<AF64>(amount, j + 1, n - 1, 0.0, thr, 1);
        nxt[j] = r < 0 ? VOL_END : (uint32_t)r;
        fragile[j] = 0;
        if (r >= 0 && (unsigned)(r - j) > __atomic_load_n(maxlen, __ATOMIC_RELAXED)) atomicMax(maxlen, (unsigned)(r - j));
    }
}

// Few fragile ticks (continuous amounts: ~2e-11 x length^2 per tick) and long bars: a single thread replaying a 30 000-tick
// bar runs at ~180 ns per tick (dependent, uncoalesced loads) and the whole pass waits for it.  So the live fragile ticks
// are compacted (k_vg_fragile_list) and, when they fit the list, each gets a WAVE: coalesced loads of 64 ticks, lane 0 adds
// them in tick order, 8 at a time on the assumption that none of them closes (the form of k_threshold_exact, ~20 ns per tick).
#define VG_REPLAY_LIST_CAP 65536
__global__ __launch_bounds__(256) void k_vg_fragile_list(const unsigned char *__restrict__ fragile, int64_t n, double thr,
                                                         const int *__restrict__ status, uint32_t *__restrict__ ticks,
                                                         unsigned long long *__restrict__ count)
{
    const bool ties = vol_ties_fragile(status, thr);
    const int64_t j0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (j0 >= n) return;
    if (j0 + 7 < n) {
        const unsigned long long fb = *(const unsigned long long *)(fragile + j0);
        if (fb == 0) return;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned f = (unsigned)(fb >> (8 * k)) & 0xFF;
            if (f == 1 || (f == 2 && ties)) {
                const unsigned long long pos = atomicAdd(count, 1ULL);
                if (pos < VG_REPLAY_LIST_CAP) ticks[pos] = (uint32_t)(j0 + k);
            }
        }
    } else {
        for (int64_t j = j0; j < n; ++j) {
            const unsigned f = fragile[j];
            if (f == 1 || (f == 2 && ties)) {
                const unsigned long long pos = atomicAdd(count, 1ULL);
                if (pos < VG_REPLAY_LIST_CAP) ticks[pos] = (uint32_t)j;
            }
        }
    }
}

template <bool AF64>
__global__ __launch_bounds__(64) void k_vg_replay_wave(const void *__restrict__ amount, int64_t n, double thr,
                                                       uint32_t *__restrict__ nxt, unsigned char *__restrict__ fragile,
                                                       const uint32_t *__restrict__ ticks, unsigned *__restrict__ maxlen)
{
    __shared__ __attribute__((aligned(16))) double s_v[64];
    const int lane = fmk_lane();
    const int64_t j = ticks[blockIdx.x];
    double cum = 0.0;
    int64_t m = -1;
    double cur = j + 1 + lane < n ? fmk_amt<AF64>(amount, j + 1 + lane) : 0.0;
    for (int64_t base = j + 1; base < n && m < 0; base += 64) {
        const double nx = base + 64 + lane < n ? fmk_amt<AF64>(amount, base + 64 + lane) : 0.0;
        s_v[lane] = cur;                                            // ticks past the end add 0.0: cum < thr stays
        __builtin_amdgcn_wave_barrier();
        int hit = -1;
        if (lane == 0) {
            for (int q8 = 0; q8 < 64 && hit < 0; q8 += 8) {
                double d[8], sp[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) d[k] = s_v[q8 + k];
                sp[0] = cum + d[0];
#pragma unroll
                for (int k = 1; k < 8; ++k) sp[k] = sp[k - 1] + d[k];
                bool any = false;
#pragma unroll
                for (int k = 0; k < 8; ++k) any |= sp[k] >= thr;
                if (!any) { cum = sp[7]; continue; }
#pragma unroll
                for (int k = 7; k >= 0; --k) if (sp[k] >= thr) hit = q8 + k;     // the first one (sums of v >= 0 do not decrease)
            }
        }
        hit = __builtin_amdgcn_readfirstlane(hit);
        if (hit >= 0) m = base + hit;
        __builtin_amdgcn_wave_barrier();
        cur = nx;
    }
    if (lane == 0) {
        nxt[j] = m < 0 ? VOL_END : (uint32_t)m;
        fragile[j] = 0;
        if (m >= 0 && (unsigned)(m - j) > __atomic_load_n(maxlen, __ATOMIC_RELAXED)) atomicMax(maxlen, (unsigned)(m - j));
    }
}

// certification of the listed decisions; returns FMK_OK with c.unc = 0 (all replayed and confirmed) or the raw count in
// fast mode, or 3 (a replay disagrees / more fragile decisions than the list holds -> serial walk)
static int vol_certify(fmk_ctx *ctx, const void *a, int is_f64, int64_t n, double thr, VolCache &c)
{
    int *d_mis = (int *)(ctx->d_mail + 41);
    FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[3], c.d_list, 8, hipMemcpyDeviceToHost, ctx->stream));
    FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t listed = ctx->h_mail[3];
    c.unc = listed;
    if (listed == 0 || ctx->fast_threshold) return FMK_OK;
    if (listed > VOL_LIST_CAP) return 3;
    FMK_HIP(ctx, hipMemsetAsync(d_mis, 0, 8, ctx->stream));
    if (is_f64) k_vol_verify<true><<<(unsigned)fmk_ceil_div(listed, 64), 64, 0, ctx->stream>>>(a, n, thr, c.dbuf, c.count, c.d_list, d_mis);
    else k_vol_verify<false><<<(unsigned)fmk_ceil_div(listed, 64), 64, 0, ctx->stream>>>(a, n, thr, c.dbuf, c.count, c.d_list, d_mis);
    FMK_LAUNCH_CHECK(ctx);
    FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[3], d_mis, 8, hipMemcpyDeviceToHost, ctx->stream));
    FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->h_mail[3] & 1) return 3;
    c.unc = 0;                                                      // every fragile decision replayed and confirmed
    return FMK_OK;
}

// workspace of the global tier (5 + 16 bytes per tick).  Returns 1 -- "use the next tier" -- when the device cannot
// provide it: the chain walk needs no tables, and a full HBM is no reason to fail the call.
static int vol_ensure(fmk_ctx *ctx, void **buf, size_t *have, size_t bytes)
{
    if (*have >= bytes) return FMK_OK;
    if (*buf) FMK_HIP(ctx, hipFree(*buf));
    *buf = nullptr; *have = 0;
    if (hipMalloc(buf, bytes) != hipSuccess) {
        (void)hipGetLastError();                                    // clear the sticky out-of-memory error
        *buf = nullptr;
        return 1;
    }
    *have = bytes;
    return FMK_OK;
}

// needs the full-stream prefix of a total_only vol_chase call in c.work.  Returns FMK_OK, 1 (not suitable: walk the chain)
static int vol_global_tables(fmk_ctx *ctx, const void *a, int is_f64, int64_t n, double thr, VolCache &c, double mean_len)
{
    const int64_t nblk = fmk_ceil_div(n, VC_BLOCK);
    const size_t lp_bytes = ((size_t)n * 8 + 255) & ~(size_t)255;
    const size_t tot_bytes = ((size_t)nblk * 8 + 255) & ~(size_t)255;
    const double *Lp = (const double *)c.work;
    const VcDD *Bb = (const VcDD *)((char *)c.work + lp_bytes + tot_bytes);
    const size_t nxt_bytes = ((size_t)n * 4 + 255) & ~(size_t)255;
    FMK_TRY(vol_ensure(ctx, &c.work2, &c.work2_bytes, nxt_bytes + (size_t)n + 256));
    if (!c.d_list) FMK_HIP(ctx, hipMalloc((void **)&c.d_list, ((size_t)1 + VOL_LIST_CAP) * 8));
    uint32_t *nxt = (uint32_t *)c.work2;
    unsigned char *fragile = (unsigned char *)c.work2 + nxt_bytes;
    uint32_t *d_root = (uint32_t *)(ctx->d_mail + 36);
    unsigned *d_maxlen = (unsigned *)(ctx->d_mail + 37);
    int *d_status = (int *)(ctx->d_mail + 38);
    FMK_HIP(ctx, hipMemsetAsync(ctx->d_mail + 36, 0, 24, ctx->stream));
    FMK_HIP(ctx, hipMemsetAsync(c.d_list, 0, 8, ctx->stream));
    const int *d_pstat = (const int *)(ctx->d_mail + 40);          // status of the prefix pass (k_vc_prefix)
    unsigned long long *d_nfrag = (unsigned long long *)(ctx->d_mail + 39);
    FMK_HIP(ctx, hipMemsetAsync(d_nfrag, 0, 8, ctx->stream));
    k_vg_nxt<<<(unsigned)fmk_ceil_div(nblk, 4), 256, 0, ctx->stream>>>(Lp, Bb, n, nblk, thr, nxt, fragile, d_maxlen, d_pstat,
                                                                       d_nfrag);
    k_vg_root<<<1, 64, 0, ctx->stream>>>(Lp, Bb, n, nblk, thr, d_root, c.d_list, d_pstat);
    FMK_LAUNCH_CHECK(ctx);
    FMK_HIP(ctx, hipMemcpyAsync(ctx->h_mail, ctx->d_mail + 36, 32, hipMemcpyDeviceToHost, ctx->stream));
    FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!ctx->fast_threshold) {
        // exact mode: settle the fragile ticks before the tables are built -- while that is cheaper than the serial walk
        const double est_fragile = (double)ctx->h_mail[3] * (nblk >= 64 ? 64.0 : (double)nblk);
        if (est_fragile * mean_len > 24000.0 * (double)n) return 3; // 7e-13 s per replayed addition against 19 ns per tick
        int64_t few = -1;                                           // live fragile ticks when they fit the wave-replay list
        if (est_fragile < (double)VG_REPLAY_LIST_CAP / 2) {
            uint32_t *ticks = (uint32_t *)(c.d_list + 1 + VOL_LIST_CAP / 2);     // upper half of the chain list: unused until the emit
            FMK_HIP(ctx, hipMemsetAsync(d_nfrag, 0, 8, ctx->stream));
            k_vg_fragile_list<<<(unsigned)fmk_ceil_div(fmk_ceil_div(n, 8), 256), 256, 0, ctx->stream>>>(fragile, n, thr, d_pstat,
                                                                                                      ticks, d_nfrag);
            FMK_LAUNCH_CHECK(ctx);
            FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[3], d_nfrag, 8, hipMemcpyDeviceToHost, ctx->stream));
            FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->h_mail[3] <= VG_REPLAY_LIST_CAP) few = ctx->h_mail[3];
            if (few > 0) {
                if (is_f64) k_vg_replay_wave<true><<<(unsigned)few, 64, 0, ctx->stream>>>(a, n, thr, nxt, fragile, ticks, d_maxlen);
                else k_vg_replay_wave<false><<<(unsigned)few, 64, 0, ctx->stream>>>(a, n, thr, nxt, fragile, ticks, d_maxlen);
            }
        }
        // thread per tick: whatever the wave pass did not take (everything when there are many), and a listed first decision
        const int only_root = few >= 0 ? 1 : 0;
        const unsigned rblocks = only_root ? 1u : (unsigned)fmk_ceil_div(n, 2048);
        if (is_f64) k_vg_replay<true><<<rblocks, 256, 0, ctx->stream>>>(a, n, thr, nxt, fragile, d_pstat, d_maxlen, d_root, c.d_list,
                                                                        only_root);
        else k_vg_replay<false><<<rblocks, 256, 0, ctx->stream>>>(a, n, thr, nxt, fragile, d_pstat, d_maxlen, d_root, c.d_list,
                                                                   only_root);
        FMK_LAUNCH_CHECK(ctx);
        FMK_HIP(ctx, hipMemcpyAsync(ctx->h_mail, ctx->d_mail + 36, 16, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    const uint32_t root = (uint32_t)(ctx->h_mail[0] & 0xFFFFFFFFu);
    const int64_t longest = (int64_t)(ctx->h_mail[1] & 0xFFFFFFFFu);
    int ls = 12;
    while (((int64_t)1 << ls) < longest) ++ls;
    const int64_t S = (int64_t)1 << ls;
    // every tick walks S / bar-length links in k_vg_level0 (~2 ps per link at this parallelism: 3 ms at S / length = 1.6 and
    // 1e9 ticks); the chain walk costs 1.4 us per CLOSE.  A quiet stretch that makes the longest bar 32 x the mean still
    // leaves the tables far ahead; beyond that (or a span of more than 4M ticks) the chain is walked
    if ((double)S > 32.0 * mean_len || ls > 22) return 1;
    const int64_t nblk0 = fmk_ceil_div(n, S);
    int64_t nb[64], spanq[64];
    int K = 0;
    nb[0] = nblk0;
    spanq[0] = S;
    while (nb[K] > 1) { nb[K + 1] = (nb[K] + VOL_RADIX - 1) / VOL_RADIX; spanq[K + 1] = spanq[K] * VOL_RADIX; ++K; }   // radix-16 levels
    size_t tbl = 0, ents = 0;
    for (int k = 0; k <= K; ++k) { tbl += (size_t)nb[k] * S; ents += (size_t)nb[k]; }
    FMK_TRY(vol_ensure(ctx, &c.work3, &c.work3_bytes, 2 * tbl * 4 + ents * (4 + 8) + 256));
    uint32_t *Eall = (uint32_t *)c.work3;
    uint32_t *Call = Eall + tbl;
    int64_t *offall = (int64_t *)(Call + tbl);
    uint32_t *entall = (uint32_t *)(offall + ents);
    uint32_t *E[64], *C[64], *ent[64];
    int64_t *off[64];
    {
        size_t to = 0, eo = 0;
        for (int k = 0; k <= K; ++k) {
            E[k] = Eall + to; C[k] = Call + to; to += (size_t)nb[k] * S;
            ent[k] = entall + eo; off[k] = offall + eo; eo += (size_t)nb[k];
        }
    }
    k_vg_level0<<<(unsigned)fmk_ceil_div(nblk0 * S, 256), 256, 0, ctx->stream>>>(nxt, n, ls, nblk0 * S, E[0], C[0]);
    FMK_LAUNCH_CHECK(ctx);
    for (int k = 1; k <= K; ++k) {
        k_vol_level_up4<<<(unsigned)fmk_ceil_div(nb[k] * S, 256), 256, 0, ctx->stream>>>(
            ls, E[k - 1], C[k - 1], nb[k - 1], spanq[k - 1], E[k], C[k], nb[k], d_status, VOL_RADIX);
        FMK_LAUNCH_CHECK(ctx);
    }
    int64_t closes = 0;
    if (root != VOL_END) {
        if ((int64_t)root >= S) return 1;
        FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[0], C[K] + root, 4, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[1], d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->h_mail[1] & VOL_ST_OVERFLOW) return 1;
        closes = (int64_t)(ctx->h_mail[0] & 0xFFFFFFFFu);
    }
    c.count = closes + 1;
    if (c.dbuf && c.cap < c.count) { FMK_HIP(ctx, hipFree(c.dbuf)); c.dbuf = nullptr; }
    if (!c.dbuf) { FMK_HIP(ctx, hipMalloc((void **)&c.dbuf, (size_t)c.count * 8)); c.cap = c.count; }
    const int64_t one = 1;
    FMK_HIP(ctx, hipMemcpyAsync(ent[K], &root, 4, hipMemcpyHostToDevice, ctx->stream));
    FMK_HIP(ctx, hipMemcpyAsync(off[K], &one, 8, hipMemcpyHostToDevice, ctx->stream));
    for (int k = K; k >= 1; --k) {
        k_vol_descend4<<<(unsigned)fmk_ceil_div(nb[k], 256), 256, 0, ctx->stream>>>(
            ls, ent[k], off[k], nb[k], spanq[k - 1], E[k - 1], C[k - 1], nb[k - 1], ent[k - 1], off[k - 1], VOL_RADIX);
        FMK_LAUNCH_CHECK(ctx);
    }
    k_vol_emit<<<(unsigned)fmk_ceil_div(nblk0, 256), 256, 0, ctx->stream>>>(ls, ent[0], off[0], nblk0, nxt, c.dbuf, c.cap,
                                                                            fragile, c.d_list, d_pstat, nullptr, thr);
    FMK_LAUNCH_CHECK(ctx);
    return vol_certify(ctx, a, is_f64, n, thr, c);
}

extern "C" int fmk_volume_bar_indexer_dev(fmk_ctx *ctx, const void *d_amount, int amount_is_f64, int64_t n,
                                          double threshold, int64_t *d_close_idx, int64_t capacity, int64_t *n_idx,
                                          int64_t *n_uncertified)
{
    if (n <= 0) return fmk_set_error(ctx, FMK_E_ARG, "threshold indexer: empty input");
    const bool serial = !(threshold > 0.0) || n >= ((int64_t)1 << 31) - 4096 || getenv("FMK_THRESHOLD_SERIAL");
    if (serial)
        return fmk_threshold_serial(ctx, 0, nullptr, d_amount, amount_is_f64, n, threshold, d_close_idx, capacity,
                                    n_idx, n_uncertified);
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    VolCache &c = vol_cache(ctx);
    const bool hit = c.ctx == ctx && !ctx->idx_stale[0] && c.amount == d_amount && c.n == n && c.thr == threshold &&
                     c.is_f64 == amount_is_f64 && c.dbuf && d_close_idx;
    if (!hit) {
        // Tier by mean bar length.  A sample of the head of the stream (prefix + total of the first 2^19 ticks: two tiny
        // launches and one read-back) decides whether the 2048-tick tables are worth building at all -- at 1e9 ticks a
        // doomed attempt costs 27 ms.  A misleading sample only costs time: every tier still reports overflow.
        double est_total = 0.0;
        const int64_t ns = n < ((int64_t)1 << 19) ? n : ((int64_t)1 << 19);
        int rcs = amount_is_f64 ? vol_chase<true>(ctx, d_amount, n, threshold, c, true, &est_total, false, ns)
                                : vol_chase<false>(ctx, d_amount, n, threshold, c, true, &est_total, false, ns);
        if (rcs && rcs != 2) return rcs;
        const double est_len = est_total > 0.0 ? (double)ns * threshold / est_total : 1e300;
        int rc = 1;
        // the exact-sum tier first (fmk_volume_exact.h): every window certifies that its float64 sums are exact, so there is no
        // tie zone to replay; entry tables for the first 1024 / 2048 ticks of 4096-tick blocks.  A window that does not certify or
        // a bar beyond the table span hands the call to the tiers below
        if (est_len < 1500.0) {
            // table span W >= the longest bar; the estimate is the MEAN length of the head of the stream, so each class leaves room
            // (x 1.6 / x 1.3) and a bar that is longer after all sends the call to the next class (flag, then one more attempt)
            if (est_len < 640.0)
                rc = amount_is_f64 ? vx_run<true, 3072, 1024, 512, false>(ctx, d_amount, n, threshold, c)
                                   : vx_run<false, 3072, 1024, 512, false>(ctx, d_amount, n, threshold, c);
            // (a (3840, 1280) class with 640 threads -- a third of the rows per tick -- was measured: 7.3 against 6.3 ms at 865-tick bars;
            //  256 threads with twice the ticks per thread: no better)
            if (rc == 1 && est_len < 1150.0)
                rc = amount_is_f64 ? vx_run<true, 2560, 1536, 512, false>(ctx, d_amount, n, threshold, c)
                                   : vx_run<false, 2560, 1536, 512, false>(ctx, d_amount, n, threshold, c);
            if (rc == 1)
                rc = amount_is_f64 ? vx_run<true, 4096, 2048, 512, true>(ctx, d_amount, n, threshold, c)
                                   : vx_run<false, 4096, 2048, 512, true>(ctx, d_amount, n, threshold, c);
            if (rc == 4) rc = 1;                                    // does not certify: the tiers below
        }
        if (rc == 1 && est_len < 1800.0)
            rc = amount_is_f64 ? vol_run<true, 2048>(ctx, d_amount, n, threshold, c)
                               : vol_run<false, 2048>(ctx, d_amount, n, threshold, c);
        if (rc == 1) {
            // a bar longer than 2048 ticks (or expected).  The exact mean bar length from the total volume decides the tier
            double total = 0.0;
            int rc2 = amount_is_f64 ? vol_chase<true>(ctx, d_amount, n, threshold, c, true, &total)
                                    : vol_chase<false>(ctx, d_amount, n, threshold, c, true, &total);
            if (rc2 && rc2 != 2) return rc2;
            if (rc2 == 2 || rcs == 2)
                return fmk_threshold_serial(ctx, 0, nullptr, d_amount, amount_is_f64, n, threshold, d_close_idx, capacity,
                                            n_idx, n_uncertified);
            const double mean_len = total > 0.0 ? (double)n * threshold / total : 1e300;
            bool prefix_ok = true;          // the table tiers build in c.work too: after one of them the prefix is gone
            if (est_len >= 1800.0 && mean_len < 1400.0) {    // the sample misled: short bars after all
                prefix_ok = false;
                rc = amount_is_f64 ? vol_run<true, 2048>(ctx, d_amount, n, threshold, c)
                                   : vol_run<false, 2048>(ctx, d_amount, n, threshold, c);
            }
            // bars beyond the 2048-tick LDS tables: global tables up to ~64K ticks (26 ms at 1e9 ticks whatever the length;
            // the 4096-tick LDS tables this used to try first cost 33 ms), the chain walk beyond
            if (rc == 1 && mean_len < 65536.0) {
                if (!prefix_ok) {            // a table attempt overwrote the prefix of the total pass
                    rc2 = amount_is_f64 ? vol_chase<true>(ctx, d_amount, n, threshold, c, true, &total)
                                        : vol_chase<false>(ctx, d_amount, n, threshold, c, true, &total);
                    if (rc2) return rc2;
                    prefix_ok = true;
                }
                rc = vol_global_tables(ctx, d_amount, amount_is_f64, n, threshold, c, mean_len);
            }
            if (rc == 1)     // few, long bars: walk the chain with wave-parallel searches (total pass's prefix if still there)
                rc = amount_is_f64 ? vol_chase<true>(ctx, d_amount, n, threshold, c, false, nullptr, prefix_ok)
                                   : vol_chase<false>(ctx, d_amount, n, threshold, c, false, nullptr, prefix_ok);
        }
        if (rc == 2 || rc == 3) rc = 1;
        if (rc == 1)     // negative volumes, or a fragile decision whose replay disagrees
            return fmk_threshold_serial(ctx, 0, nullptr, d_amount, amount_is_f64, n, threshold, d_close_idx, capacity,
                                        n_idx, n_uncertified);
        if (rc) return rc;
        c.ctx = ctx; c.amount = d_amount; c.n = n; c.thr = threshold; c.is_f64 = amount_is_f64;
        ctx->idx_key[0][0] = d_amount; ctx->idx_key[0][1] = nullptr; ctx->idx_stale[0] = 0;
    }
    if (c.unc > 0 && !ctx->fast_threshold) {      // see fmk_dollar_bar_indexer_dev
        c.ctx = nullptr;
        return fmk_threshold_serial(ctx, 0, nullptr, d_amount, amount_is_f64, n, threshold, d_close_idx, capacity, n_idx,
                                    n_uncertified);
    }
    *n_idx = c.count;
    if (n_uncertified) *n_uncertified = c.unc;
    if (!d_close_idx) return FMK_OK;
    if (capacity < c.count) return fmk_set_error(ctx, FMK_E_CAPACITY, "threshold indexer: capacity %lld < %lld",
                                                 (long long)capacity, (long long)c.count);
    FMK_HIP(ctx, hipMemcpyAsync(d_close_idx, c.dbuf, (size_t)c.count * 8, hipMemcpyDeviceToDevice, ctx->stream));
    c.ctx = nullptr;     // one-shot cache: the inputs may change behind the same pointers
    return FMK_OK;
}
