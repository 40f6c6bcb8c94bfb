// fmk_cusum.hip -- _cusum_bar_indexer (finmlkit/bar/logic.py:152-221) on gfx950.
//
// The reference is one sequential loop with a two-component state
//     s_pos = max(0, s_pos + r_i),  s_neg = min(0, s_neg + r_i),   r_i = log(p_i / p_{i-1})
// and a reset of ONE side when it crosses +-lambda_i at a tick that is not followed by a same-timestamp tick.
// Because only one side resets, the JOINT state after a close is not a function of the close position alone (unlike the
// volume bars).  (Each side on its own is: fmk_cusum_chain.hip walks the two chains when thresholds are rarely reached and
// hands every other call, and every decision it cannot certify, to this file.)  What the recursion does have is FORGETTING: the max / min
// clamps and the resets erase the memory of the incoming state after a few hundred ticks.  That makes it a good
// fit for a parallel-in-time fixed point:
//     round 0 : every chunk of C ticks is simulated from the state (0, 0)             -> out_0[k]
//     round r : chunk k is simulated from out_{r-1}[k-1]                               -> out_r[k]
//     stop when out_r == out_{r-1} (bitwise): every chunk then started from the true state of its predecessor,
//     i.e. the chunks reproduce the sequential loop exactly (chunk 0 is exact from round 0 on, so at most K rounds;
//     on real streams 2-4).
// One THREAD owns one chunk and runs the reference loop (same operations, same order), so the decisions are the
// reference's; a final pass with the converged states writes the close indices at their scanned offsets.  The loop
// inputs (log return, threshold / "cannot close" marker) are computed once and stored chunk-transposed so that
// every round streams 16 B/tick fully coalesced; a chunk whose incoming state did not change is skipped.
// Cost model: rounds ~ (ticks the state remembers) / CS_CHUNK.  Thresholds that are reached every few hundred ticks
// converge in 2-4 rounds; thresholds that are almost never reached (one close per 10^5 ticks: the clamps rarely bind)
// need hundreds of rounds -- still exact, but then the method has no advantage over the sequential loop.
// The sigma forward-fill (logic.py:176-189, done IN PLACE like the reference) is a "last non-NaN" scan.
#include <math.h>
#include <stdlib.h>

#include "fmk_common.h"
#include "fmk_log.h"
#include "fmk_scan.h"

#define CS_CHUNK 2048          // ticks per thread
#define CS_THREADS 64

// ---------------------------------------------------------------------------------------
// sigma forward fill + first non-NaN index
// ---------------------------------------------------------------------------------------
#define FF_THREADS 256
#define FF_ITEMS 8
#define FF_TILE (FF_THREADS * FF_ITEMS)

// last non-NaN value of the block's threads in thread order (NaN: none): exclusive value + block aggregate
__device__ __forceinline__ double ff_block_exclusive(double mine, double *lds /*[4]*/, double *block_total)
{
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
    double inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double o = __shfl_up(inc, d, 64);
        if (lane >= d && isnan(inc)) inc = o;
    }
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    double pre = NAN;
    for (int k = 0; k < w; ++k) pre = isnan(lds[k]) ? pre : lds[k];
    double tot = NAN;
    for (int k = 0; k < 4; ++k) tot = isnan(lds[k]) ? tot : lds[k];
    *block_total = tot;
    double prev = __shfl_up(inc, 1, 64);
    if (lane == 0) prev = NAN;
    __syncthreads();
    return isnan(prev) ? pre : prev;
}

// tile_last[b] = the tile's last non-NaN value (NaN: none); *first_valid = index of the stream's first non-NaN.
// Only the LAST valid value of the tile is needed here, i.e. the value at the largest valid index: an argmax, so the loads are
// coalesced (thread t: items t, t + 256, ...; its own last valid item is the one with the largest k).  The stream's first
// valid index is an atomicMin that is attempted only when it would lower the current value: issued unconditionally it was one
// same-address atomic per wave, 2 M of them at ~10 ns -- 22.5 ms for a kernel that reads 8 GB (now see profiles/).
__global__ __launch_bounds__(FF_THREADS) void k_ff_tile(const double *__restrict__ x, int64_t n,
                                                        double *__restrict__ tile_last,
                                                        unsigned long long *first_valid, int *__restrict__ tile_nan)
{
    __shared__ double lds_v[4];
    __shared__ int64_t lds_i[4];
    __shared__ int lds_c[4];
    const int64_t t0 = (int64_t)blockIdx.x * FF_TILE + (int64_t)threadIdx.x;
    double v[FF_ITEMS];
#pragma unroll
    for (int k = 0; k < FF_ITEMS; ++k) v[k] = t0 + (int64_t)k * FF_THREADS < n ? x[t0 + (int64_t)k * FF_THREADS] : NAN;
    double last = NAN;
    int64_t last_i = -1, first = INT64_MAX;
    int nan_cnt = 0;                                                 // NaNs of the tile (inside the array)
#pragma unroll
    for (int k = 0; k < FF_ITEMS; ++k) {
        const int64_t i = t0 + (int64_t)k * FF_THREADS;
        if (!isnan(v[k])) {
            last = v[k]; last_i = i;                                 // k ascends: the thread's last valid item
            if (first == INT64_MAX) first = i;
        } else if (i < n) ++nan_cnt;
    }
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
    first = fmk_wave_min(first);
    if (lane == 0 && first != INT64_MAX && (unsigned long long)first < __atomic_load_n(first_valid, __ATOMIC_RELAXED))
        atomicMin(first_valid, (unsigned long long)first);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {                               // value at the largest valid index of the wave
        const int64_t oi = __shfl_xor(last_i, d, 64);
        const double ov = __shfl_xor(last, d, 64);
        if (oi > last_i) { last_i = oi; last = ov; }
    }
    nan_cnt = fmk_wave_sum(nan_cnt);
    if (lane == 0) { lds_v[w] = last; lds_i[w] = last_i; lds_c[w] = nan_cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double bv = lds_v[0];
        int64_t bi = lds_i[0];
        for (int q = 1; q < 4; ++q) if (lds_i[q] > bi) { bi = lds_i[q]; bv = lds_v[q]; }
        tile_last[blockIdx.x] = bi >= 0 ? bv : NAN;
        tile_nan[blockIdx.x] = lds_c[0] + lds_c[1] + lds_c[2] + lds_c[3];
    }
}

// The first non-NaN index when it lies in the stream's head (the usual sigma, an EWM, is NaN at its first ticks only): the
// chain walk then reads sigma as it is, its summary pass reports a NaN it meets after that index, and only then -- or for the
// fixed point -- is the full pass above run.  Saves the 8 GB read of k_ff_tile (1.5 ms per 1e9 ticks).
__global__ __launch_bounds__(1024) void k_ff_head(const double *__restrict__ x, int64_t n_head, unsigned long long *first_valid)
{
    int64_t first = INT64_MAX;
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n_head && first == INT64_MAX; i += (int64_t)gridDim.x * 1024)
        if (!isnan(x[i])) first = i;
    first = fmk_wave_min(first);
    if (fmk_lane() == 0 && first != INT64_MAX) atomicMin(first_valid, (unsigned long long)first);
}

// total number of NaNs (one block): with the first valid index it tells whether there is anything to fill -- all NaNs lead
// iff their number equals that index -- and the scan and the 8 GB apply pass are skipped when there is not (the usual case:
// an EWM sigma is NaN at its first tick only)
__global__ __launch_bounds__(1024) void k_ff_count(const int *__restrict__ tile_nan, int64_t tiles, long long *total)
{
    __shared__ long long ws[16];
    long long acc = 0;
    for (int64_t i = threadIdx.x; i < tiles; i += 1024) acc += tile_nan[i];
    acc = fmk_wave_sum(acc);
    if (fmk_lane() == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (int k = 0; k < 16; ++k) t += ws[k];
        *total = t;
    }
}

__global__ __launch_bounds__(1024) void k_ff_scan_tiles(double *tile_last, int64_t tiles)
{
    __shared__ double ws[16];
    __shared__ double run;
    if (threadIdx.x == 0) run = NAN;
    __syncthreads();
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
    for (int64_t b = 0; b < tiles; b += 1024) {
        const int64_t i = b + threadIdx.x;
        const double v = i < tiles ? tile_last[i] : NAN;
        double inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double o = __shfl_up(inc, d, 64);
            if (lane >= d && isnan(inc)) inc = o;
        }
        if (lane == 63) ws[w] = inc;
        __syncthreads();
        double pre = run;
        for (int k = 0; k < w; ++k) pre = isnan(ws[k]) ? pre : ws[k];
        double prev = __shfl_up(inc, 1, 64);
        if (lane == 0) prev = NAN;
        if (i < tiles) tile_last[i] = isnan(prev) ? pre : prev;
        __syncthreads();
        if (threadIdx.x == 1023) run = isnan(inc) ? pre : inc;
        __syncthreads();
    }
}

__global__ __launch_bounds__(FF_THREADS) void k_ff_apply(double *__restrict__ x, int64_t n,
                                                         const double *__restrict__ tile_pre)
{
    __shared__ double lds[4];
    const int64_t i0 = (int64_t)blockIdx.x * FF_TILE + (int64_t)threadIdx.x * FF_ITEMS;
    double v[FF_ITEMS];
    double last = NAN;
#pragma unroll
    for (int k = 0; k < FF_ITEMS; ++k) {
        const int64_t i = i0 + k;
        v[k] = i < n ? x[i] : NAN;
        last = isnan(v[k]) ? last : v[k];
    }
    double tot;
    double cur = ff_block_exclusive(last, lds, &tot);
    if (isnan(cur)) cur = tile_pre[blockIdx.x];
#pragma unroll
    for (int k = 0; k < FF_ITEMS; ++k) {
        const int64_t i = i0 + k;
        if (isnan(v[k])) { if (i < n && !isnan(cur)) x[i] = cur; }     // logic.py:187-189 (only after the first valid)
        else cur = v[k];
    }
}

// ---------------------------------------------------------------------------------------
// chunk simulation
// ---------------------------------------------------------------------------------------
struct CsState { double sp, sn; };

// Per-tick inputs of the loop, computed ONCE and stored chunk-TRANSPOSED in blocks of 64 chunks -- element j of chunk k at
// [((k >> 6) * CS_CHUNK + j) * 64 + (k & 63)] -- so that the one-thread-per-chunk simulation below streams 512 contiguous
// bytes per wave, array and tick, one after the other:
//   ret[t] = log(p_i / p_{i-1})                                             (logic.py:200)
//   lam[t] = max(sigma_mult * sigma_i, sigma_floor), NaN inside a same-timestamp print block (logic.py:206-211):
//            a NaN threshold can never be reached, which is exactly "this tick cannot close a bar"
// for tick i = first + 1 + t, t = k * CS_CHUNK + j.  Tiles of 32 chunks x 64 ticks through LDS: every wave instruction of the
// read side takes 512 contiguous bytes of a column, the write side fills 256-byte halves of adjacent rows.  (The first layout,
// [j * chunks + k] with 64 x 32 tiles, read 256-byte pieces 16 KB apart and wrote rows 3.9 MB apart: 12.3 ms per 1e9 ticks.)
#define CS_PREP_TK 32           // chunks per tile
#define CS_PREP_TJ 64           // ticks per tile: 33.3 KB of LDS -> four workgroups per CU
__device__ __forceinline__ int64_t cs_tr_index(int64_t k, int j) { return (((k >> 6) * CS_CHUNK + j) << 6) + (k & 63); }
__global__ __launch_bounds__(256) void k_cusum_prep(const int64_t *__restrict__ ts, const double *__restrict__ price,
                                                    const double *__restrict__ sigma, int64_t n, int64_t first,
                                                    int64_t m, int64_t chunks, double sigma_floor, double sigma_mult,
                                                    double *__restrict__ t_ret, double *__restrict__ t_lam,
                                                    unsigned long long *nan_flag /* null: sigma has been forward filled */)
{
    __shared__ double s_r[CS_PREP_TK][CS_PREP_TJ + 1];
    __shared__ double s_l[CS_PREP_TK][CS_PREP_TJ + 1];
    const int64_t k0 = (int64_t)blockIdx.x * CS_PREP_TK;
    const int j0 = (int)blockIdx.y * CS_PREP_TJ;
    bool nan_sigma = false;
    {
        const int col = threadIdx.x & (CS_PREP_TJ - 1), row4 = threadIdx.x / CS_PREP_TJ;
        constexpr int RP = 256 / CS_PREP_TJ;                            // rows per pass
#pragma unroll
        for (int rr = 0; rr < CS_PREP_TK / RP; ++rr) {
            const int row = rr * RP + row4;                             // chunk inside the tile
            const int64_t t = (k0 + row) * CS_CHUNK + j0 + col;
            double r = 0.0, lam = NAN;
            if (k0 + row < chunks && t < m) {
                const int64_t i = first + 1 + t;
                r = fmk_log_ratio(price[i], price[i - 1]);
                const bool block = i + 1 < n && ts[i] == ts[i + 1];
                const double sg = sigma[i];
                nan_sigma |= sg != sg;
                if (!block) {
                    lam = sigma_mult * sg;
                    lam = sigma_floor > lam ? sigma_floor : lam;         // max(lam, floor): a NaN lam stays NaN
                }
            }
            s_r[row][col] = r;
            s_l[row][col] = lam;
        }
    }
    // sigma read as it is (not forward filled): a NaN after its first valid index makes the caller fill it and start over
    if (nan_flag && __builtin_amdgcn_ballot_w64(nan_sigma) != 0 && fmk_lane() == 0 && __atomic_load_n(nan_flag, __ATOMIC_RELAXED) == 0)
        atomicOr(nan_flag, 1ULL);
    __syncthreads();
    {
        const int col = threadIdx.x & (CS_PREP_TK - 1), jr8 = threadIdx.x / CS_PREP_TK;
        constexpr int JP = 256 / CS_PREP_TK;                            // ticks per pass
#pragma unroll
        for (int rr = 0; rr < CS_PREP_TJ / JP; ++rr) {
            const int jrow = rr * JP + jr8;                             // tick inside the tile
            const int64_t k = k0 + col;
            if (k < chunks) {
                const int64_t at = cs_tr_index(k, j0 + jrow);
                t_ret[at] = s_r[col][jrow];
                t_lam[at] = s_l[col][jrow];
            }
        }
    }
}

// One thread = one chunk of the reference loop (logic.py:199-219), same operations in the same order.
//   in      : states to start from (in[k-1] for chunk k; chunk 0 starts from (0, 0)); nullptr: all (0, 0)
//   last_in : the state chunk k was simulated from in the previous round -- unchanged input => unchanged output,
//             the chunk is skipped (on real streams most chunks converge after 2-3 rounds)
//   changed : number of chunks whose `out` differs from `prev_out`
//   closes  : nullptr, or the output array -- chunk k writes at closes[offsets[k] + ...]
__global__ __launch_bounds__(CS_THREADS) void k_cusum_chunks(const double *__restrict__ t_ret,
                                                             const double *__restrict__ t_lam, int64_t m,
                                                             int64_t first, int64_t chunks,
                                                             const CsState *__restrict__ in, CsState *__restrict__ out,
                                                             const CsState *__restrict__ prev_out,
                                                             CsState *__restrict__ last_in,
                                                             int64_t *__restrict__ counts, unsigned long long *changed,
                                                             const int64_t *__restrict__ offsets,
                                                             int64_t *__restrict__ closes)
{
    const int64_t k = (int64_t)blockIdx.x * CS_THREADS + threadIdx.x;
    if (k >= chunks) return;
    double sp = 0.0, sn = 0.0;
    if (in && k > 0) { sp = in[k - 1].sp; sn = in[k - 1].sn; }
    if (prev_out && last_in && !closes) {
        const bool same_in = __double_as_longlong(last_in[k].sp) == __double_as_longlong(sp) &&
                             __double_as_longlong(last_in[k].sn) == __double_as_longlong(sn);
        if (same_in) { out[k] = prev_out[k]; return; }                   // counts[k] already holds this result
    }
    if (last_in) { last_in[k].sp = sp; last_in[k].sn = sn; }
    const int64_t t0 = k * CS_CHUNK;
    const int len = (int)(m - t0 < CS_CHUNK ? m - t0 : CS_CHUNK);
    int64_t cnt = 0;
    int64_t w = closes ? offsets[k] : 0;
    const double *pr = t_ret + cs_tr_index(k, 0), *pl = t_lam + cs_tr_index(k, 0);
#pragma unroll 8
    for (int j = 0; j < len; ++j) {
        const double ret = pr[(int64_t)j * 64];
        const double lam = pl[(int64_t)j * 64];
        const double a = sp + ret, b = sn + ret;
        sp = a > 0.0 ? a : 0.0;                                   // max(0.0, s_pos + ret): NaN -> 0.0
        sn = b < 0.0 ? b : 0.0;                                   // min(0.0, s_neg + ret)
        if (sp >= lam) { if (closes) closes[w++] = first + 1 + t0 + j; ++cnt; sp = 0.0; }
        else if (sn <= -lam) { if (closes) closes[w++] = first + 1 + t0 + j; ++cnt; sn = 0.0; }
    }
    if (out) {
        out[k].sp = sp; out[k].sn = sn;
        if (changed && prev_out) {
            const bool same = __double_as_longlong(prev_out[k].sp) == __double_as_longlong(sp) &&
                              __double_as_longlong(prev_out[k].sn) == __double_as_longlong(sn);
            if (!same) atomicAdd(changed, 1ULL);
        }
    }
    if (counts) counts[k] = cnt;
}

__global__ void k_cusum_first(int64_t *closes, int64_t first) { closes[0] = first; }

// ---------------------------------------------------------------------------------------
// Sparse phase.  Thresholds that are rarely reached (the reference's default sigma_floor = 5e-4 on a quiet tape: one
// close per 2.4e5 ticks on the synthetic stream) leave long stretches in which neither clamp binds, so the truth
// advances ONE chunk per dense round (1059 rounds = 1.1 s at N = 1e9) while every interior chunk of such a stretch is
// recomputed from a still-wrong input in every round: O(N x rounds) work.  With S = exit states and last_in = the input
// a chunk was computed from, chunk k is CONSISTENT iff last_in[k] == S[k-1]; a HEAD is an inconsistent chunk whose
// predecessor is consistent -- one per dependency chain.  After two dense rounds, if heads are few:
//   k_cusum_mark : lists the heads;
//   k_cusum_walk : one WAVE per head recomputes ret / lam from the original, contiguous columns (same expressions, same
//                  device log => the same bits), walks the chunk with wave-uniform scalars and follows its run: on into
//                  chunk k + 1 while that is not another head and was not computed from the exit state just produced.
// Every chunk of a run is recomputed once per launch by exactly one wave: O(N) work, critical path = the longest chain.
// The fixed point is "no heads".  The first inconsistent chunk of the stream is always a head with a final predecessor, so
// it becomes right: the loop ends within `chunks` launches; a torn read of a neighbour that is being rewritten only
// produces an input the next mark pass rejects.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ bool cs_same(CsState a, CsState b)
{
    return __double_as_longlong(a.sp) == __double_as_longlong(b.sp) && __double_as_longlong(a.sn) == __double_as_longlong(b.sn);
}

__global__ __launch_bounds__(256) void k_cusum_mark(const CsState *__restrict__ S, const CsState *__restrict__ last_in,
                                                    int64_t chunks, unsigned char *__restrict__ active,
                                                    int *__restrict__ list, unsigned long long *count)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= chunks) return;
    bool act = false;                                      // chunk 0 always starts from (0, 0): consistent
    if (k > 0 && !cs_same(S[k - 1], last_in[k])) act = k == 1 || cs_same(S[k - 2], last_in[k - 1]);    // ... and k - 1 consistent
    active[k] = act ? 1 : 0;
    if (act) list[atomicAdd(count, 1ULL)] = (int)k;
}

__device__ __forceinline__ double cs_lane(double v, int src)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)((unsigned long long)b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__global__ __launch_bounds__(256) void k_cusum_walk(const int64_t *__restrict__ ts, const double *__restrict__ price,
                                                    const double *__restrict__ sigma, int64_t n, int64_t first, int64_t m,
                                                    int64_t chunks, double sigma_floor, double sigma_mult,
                                                    CsState *S, CsState *last_in, int64_t *counts,
                                                    const unsigned char *__restrict__ active, const int *__restrict__ list,
                                                    int64_t n_list, int max_steps)
{
    // 256 ticks at a time: all lanes compute ret / lam (the expressions of k_cusum_prep) into two LDS rows, then lane 0 alone
    // walks them in its own vector registers, branch-free (selects instead of if / else: the same values as the loop of
    // k_cusum_chunks).  A first version walked with wave-uniform scalars through v_readlane and scalar branches: 18
    // instructions but ~540 cycles per tick -- every tick went VGPR -> SGPR -> VALU -> VCC -> branch.
    constexpr int G = 4;                                            // 64-tick rows per group
    __shared__ __attribute__((aligned(16))) double s_r[4][64 * G], s_l[4][64 * G];
    const int lane = fmk_lane();
    const int wib = (int)(threadIdx.x >> 6);
    const int64_t w = (int64_t)blockIdx.x * 4 + wib;
    if (w >= n_list) return;
    int64_t k = fmk_uniform((int64_t)list[w]);
    for (int step = 0; step < max_steps; ++step) {
        CsState in = S[k - 1];                                          // k >= 1 on every path
        in.sp = cs_lane(in.sp, 0); in.sn = cs_lane(in.sn, 0);           // one consistent copy for the whole wave
        double sp = in.sp, sn = in.sn;
        const int64_t t0 = k * CS_CHUNK;
        const int len = (int)(m - t0 < CS_CHUNK ? m - t0 : CS_CHUNK);
        int64_t cnt = 0;
        // raw inputs of a group of 256 ticks (4 per lane), all twenty loads issued together and one group ahead.  Loaded one
        // after the other and only when needed (price -> log, then ts, then sigma) a 64-tick group cost three memory round
        // trips, ~10 us; with 64-tick groups prefetched one ahead the round trip (2-3 us) was still longer than the
        // walk of a group once that had been shortened to ~1 us: 256 ticks per group balance the two.
        double c_p[G], c_pm[G], c_sg[G];
        int64_t c_ts[G], c_tsn[G];
        auto fetch = [&](int j0, double (&p)[G], double (&pm)[G], double (&sg)[G], int64_t (&tsi)[G], int64_t (&tsn)[G]) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                int64_t i = first + 1 + t0 + j0 + 64 * g + lane;
                if (i > n - 1) i = n - 1;                               // lanes past the chunk: any valid address
                p[g] = price[i]; pm[g] = price[i - 1]; sg[g] = sigma[i]; tsi[g] = ts[i];
                tsn[g] = ts[i + 1 < n ? i + 1 : i];
            }
        };
        fetch(0, c_p, c_pm, c_sg, c_ts, c_tsn);
#ifdef CS_TIMING
        long long tA = 0, tB = 0, t0c = __builtin_readcyclecounter(), t1c;
#endif
        for (int j0 = 0; j0 < len; j0 += 64 * G) {
            double n_p[G], n_pm[G], n_sg[G];
            int64_t n_ts[G], n_tsn[G];
#pragma unroll
            for (int g = 0; g < G; ++g) { n_p[g] = 1.0; n_pm[g] = 1.0; n_sg[g] = 0.0; n_ts[g] = 0; n_tsn[g] = 1; }
            if (j0 + 64 * G < len) fetch(j0 + 64 * G, n_p, n_pm, n_sg, n_ts, n_tsn);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int jj = j0 + 64 * g + lane;
                double r = 0.0, lam = NAN;
                if (jj < len) {                                         // the expressions of k_cusum_prep
                    const int64_t i = first + 1 + t0 + jj;
                    r = fmk_log_ratio(c_p[g], c_pm[g]);
                    const bool block = i + 1 < n && c_ts[g] == c_tsn[g];
                    if (!block) {
                        lam = sigma_mult * c_sg[g];
                        lam = sigma_floor > lam ? sigma_floor : lam;
                    }
                }
                s_r[wib][64 * g + lane] = r;
                s_l[wib][64 * g + lane] = lam;
                c_p[g] = n_p[g]; c_pm[g] = n_pm[g]; c_sg[g] = n_sg[g]; c_ts[g] = n_ts[g]; c_tsn[g] = n_tsn[g];
            }
            __builtin_amdgcn_wave_barrier();
#ifdef CS_TIMING
            t1c = __builtin_readcyclecounter(); tA += t1c - t0c; t0c = t1c;
#endif
            const int lim = len - j0 < 64 * G ? len - j0 : 64 * G;
            if (lane == 0) {
                for (int q8 = 0; q8 < lim; q8 += 8) {
                    const int nq = lim - q8 < 8 ? lim - q8 : 8;
                    if (nq == 8) {
                        // speculate that none of these 8 ticks closes (in this regime closes are thousands of ticks apart):
                        // then the loop is add -> clamp per side and tick, the two sides are independent chains and the
                        // compares hang off them.  Same operations, same order as the careful form below.
                        double r8[8], l8[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) { r8[q] = s_r[wib][q8 + q]; l8[q] = s_l[wib][q8 + q]; }
                        double p = sp, g2 = sn;
                        bool any = false;
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const double a = p + r8[q], b = g2 + r8[q];
                            p = a > 0.0 ? a : 0.0;
                            g2 = b < 0.0 ? b : 0.0;
                            any |= (p >= l8[q]) | (g2 <= -l8[q]);
                        }
                        if (__builtin_amdgcn_ballot_w64(any) == 0) { sp = p; sn = g2; continue; }
                    }
                    for (int q = q8; q < q8 + nq; ++q) {                // the loop of k_cusum_chunks, as selects
                        const double ret = s_r[wib][q], lm = s_l[wib][q];
                        const double a = sp + ret, b = sn + ret;
                        sp = a > 0.0 ? a : 0.0;
                        sn = b < 0.0 ? b : 0.0;
                        const bool cp = sp >= lm;
                        const bool cn = !cp && sn <= -lm;
                        cnt += (cp || cn) ? 1 : 0;
                        sp = cp ? 0.0 : sp;
                        sn = cn ? 0.0 : sn;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
#ifdef CS_TIMING
            t1c = __builtin_readcyclecounter(); tB += t1c - t0c; t0c = t1c;
#endif
        }
#ifdef CS_TIMING
        if (lane == 0 && w == 0 && step == 0) printf("cusum walk, one chunk of %d ticks: compute r/lam %lld cycles, lane-0 walk %lld cycles\n", len, tA, tB);
#endif
        sp = cs_lane(sp, 0); sn = cs_lane(sn, 0);
        if (lane == 0) {
            last_in[k] = in;
            counts[k] = cnt;
            S[k].sp = sp; S[k].sn = sn;
        }
        // follow the run: on into chunk k + 1 unless it is another wave's head or was computed from exactly this exit state
        if (k + 1 >= chunks || active[k + 1]) break;
        const CsState nin = last_in[k + 1];
        const double nsp = cs_lane(nin.sp, 0), nsn = cs_lane(nin.sn, 0);
        if (__double_as_longlong(nsp) == __double_as_longlong(sp) && __double_as_longlong(nsn) == __double_as_longlong(sn)) break;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        ++k;
    }
}

#include "fmk_cusum_onepass.h"

static int64_t g_cs1_last[4];           // last call: 1 if the one-pass form answered, fix-up launches, chunks pending after the first, chunks
extern "C" int fmk_diag_cusum_onepass(int64_t *used, int64_t *fix_launches, int64_t *pending_first, int64_t *chunks)
{
    if (used) *used = g_cs1_last[0];
    if (fix_launches) *fix_launches = g_cs1_last[1];
    if (pending_first) *pending_first = g_cs1_last[2];
    if (chunks) *chunks = g_cs1_last[3];
    return FMK_OK;
}

// The one-pass form (fmk_cusum_onepass.h).  *done = 1: d_out[1..] and *total are the answer; *done = 0: nothing was written that the
// caller may use (the tape does not forget fast enough, or the fix-up did not settle in CS1_MAX_LAUNCHES launches): run the fixed point.
// *redo = true: sigma (not forward filled yet) has a NaN after its first valid index -- fill it and call again.
#define CS1_MAX_LAUNCHES 24
static int cs1_run(fmk_ctx *ctx, const int64_t *d_ts, const double *d_price, const double *d_sigma, int64_t n, int64_t first,
                   int64_t m, double sigma_floor, double sigma_mult, int64_t *d_out, int64_t capacity, bool check_nan,
                   int64_t *total, int64_t *rounds, int *done, bool *redo)
{
    *done = 0; *redo = false;
    const int64_t chunks = fmk_ceil_div(m, (int64_t)CS1_L);
    g_cs1_last[0] = 0; g_cs1_last[1] = 0; g_cs1_last[2] = 0; g_cs1_last[3] = chunks;
    const size_t scan_bytes = (((size_t)fmk_ceil_div(chunks + 1, FMK_SCAN_TILE) + 1) * 8 + 255) & ~(size_t)255;
    const size_t st_bytes = ((size_t)chunks * sizeof(CsState) + 255) & ~(size_t)255;
    const size_t cnt_bytes = ((size_t)(chunks + 1) * 8 + 255) & ~(size_t)255;
    const size_t c0_bytes = ((size_t)chunks * 4 + 255) & ~(size_t)255;
    const size_t fix_bytes = ((size_t)chunks * sizeof(Cs1Fix) + 255) & ~(size_t)255;
    const size_t row_bytes = ((size_t)chunks * CS1_L * 2 + 255) & ~(size_t)255;
    void *scr;
    FMK_TRY(fmk_scratch(ctx, scan_bytes + 5 * st_bytes + cnt_bytes + c0_bytes + fix_bytes + 2 * row_bytes, &scr));
    char *base = (char *)scr + scan_bytes;
    CsState *S = (CsState *)base, *S_read = (CsState *)(base + st_bytes), *last_in = (CsState *)(base + 2 * st_bytes);
    CsState *S0 = (CsState *)(base + 3 * st_bytes), *E = (CsState *)(base + 4 * st_bytes);
    int64_t *counts = (int64_t *)(base + 5 * st_bytes);
    int *C0 = (int *)(base + 5 * st_bytes + cnt_bytes);
    Cs1Fix *fix = (Cs1Fix *)(base + 5 * st_bytes + cnt_bytes + c0_bytes);
    unsigned short *staged = (unsigned short *)(base + 5 * st_bytes + cnt_bytes + c0_bytes + fix_bytes);
    unsigned short *patch = (unsigned short *)(base + 5 * st_bytes + cnt_bytes + c0_bytes + fix_bytes + row_bytes);
    unsigned long long *d_changed = (unsigned long long *)(ctx->d_mail + 1), *d_pending = (unsigned long long *)(ctx->d_mail + 3);
    unsigned long long *d_nan = (unsigned long long *)(ctx->d_mail + 5);
    if (check_nan) FMK_HIP(ctx, hipMemsetAsync(d_nan, 0, 8, ctx->stream));
    ctx->h_mail[5] = 0;
    FMK_HIP(ctx, hipMemsetAsync(fix, 0, fix_bytes, ctx->stream));
    {
        const unsigned g = (unsigned)fmk_ceil_div(chunks, (int64_t)CS1_TK);
        unsigned long long *nf = check_nan ? d_nan : nullptr;
        k_cs1_pass<<<g, 256, 0, ctx->stream>>>(d_ts, d_price, d_sigma, n, first, m, chunks, sigma_floor, sigma_mult, E, S0, C0, staged, nf);
        FMK_LAUNCH_CHECK(ctx);
    }
    FMK_HIP(ctx, hipMemcpyAsync(S, S0, (size_t)chunks * sizeof(CsState), hipMemcpyDeviceToDevice, ctx->stream));
    FMK_HIP(ctx, hipMemcpyAsync(last_in, E, (size_t)chunks * sizeof(CsState), hipMemcpyDeviceToDevice, ctx->stream));   // every record was made from E
    *rounds = 1;
    int limit = CS1_FIRST_LIMIT;
    int64_t launches = 0;
    if (check_nan && chunks <= 1) {
        FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[5], d_nan, 8, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->h_mail[5] != 0) { *redo = true; return FMK_OK; }
    }
    while (chunks > 1) {
        FMK_HIP(ctx, hipMemcpyAsync(S_read, S, (size_t)chunks * sizeof(CsState), hipMemcpyDeviceToDevice, ctx->stream));
        FMK_HIP(ctx, hipMemsetAsync(d_changed, 0, 8, ctx->stream));
        FMK_HIP(ctx, hipMemsetAsync(d_pending, 0, 8, ctx->stream));
        k_cs1_fix<<<(unsigned)fmk_ceil_div(chunks - 1, (int64_t)4), 256, 0, ctx->stream>>>(
            d_ts, d_price, d_sigma, n, first, m, chunks, sigma_floor, sigma_mult, E, S0, S_read, S, last_in, fix, patch, limit,
            d_changed, d_pending);
        FMK_LAUNCH_CHECK(ctx);
        ++launches; ++*rounds;
        FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[1], d_changed, 8, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[3], d_pending, 8, hipMemcpyDeviceToHost, ctx->stream));
        if (check_nan) FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[5], d_nan, 8, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (check_nan && ctx->h_mail[5] != 0) { *redo = true; return FMK_OK; }
        const int64_t changed = ctx->h_mail[1], pending = ctx->h_mail[3];
        g_cs1_last[1] = launches;
        if (launches == 1) {
            g_cs1_last[2] = pending;
            if (pending > chunks / 4 + 1) return FMK_OK;                  // this tape does not forget within CS1_FIRST_LIMIT ticks
        }
        limit = CS1_L;
        if (changed == 0 && pending == 0) break;
        if (launches >= CS1_MAX_LAUNCHES) return FMK_OK;
    }
    k_cs1_counts<<<(unsigned)fmk_ceil_div(chunks, (int64_t)256), 256, 0, ctx->stream>>>(C0, fix, chunks, counts);
    FMK_LAUNCH_CHECK(ctx);
    FMK_TRY(fmk_exclusive_scan_i64(ctx, counts, counts, chunks, true));
    FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[2], counts + chunks, 8, hipMemcpyDeviceToHost, ctx->stream));
    FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *total = ctx->h_mail[2];
    if (d_out) {
        if (capacity < *total + 1)
            return fmk_set_error(ctx, FMK_E_CAPACITY, "cusum: %lld close indices, capacity %lld", (long long)(*total + 1),
                                 (long long)capacity);
        k_cs1_emit<<<(unsigned)fmk_ceil_div(chunks, (int64_t)4), 256, 0, ctx->stream>>>(C0, fix, staged, patch, chunks, first, counts,
                                                                                      d_out + 1);
        FMK_LAUNCH_CHECK(ctx);
    }
    g_cs1_last[0] = 1;
    *done = 1;
    return FMK_OK;
}

int fmk_cusum_chain_tier(fmk_ctx *ctx, const int64_t *d_ts, const double *d_price, const double *d_sigma, int64_t n,
                         int64_t first, int64_t m, int64_t chunks, double sigma_floor, double sigma_mult, int64_t *d_out,
                         int64_t capacity, int64_t *total, int64_t *visits, int *done, int *nan_seen);

extern "C" int fmk_cusum_bar_indexer_dev(fmk_ctx *ctx, const int64_t *d_ts, const double *d_price, double *d_sigma,
                                         int64_t n, double sigma_floor, double sigma_mult, int64_t *d_out,
                                         int64_t capacity, int64_t *n_out, int64_t *n_rounds)
{
    if (n <= 0) return fmk_set_error(ctx, FMK_E_ARG, "Prices, timestamps, and sigma arrays must have the same length.");
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    g_cs1_last[0] = g_cs1_last[1] = g_cs1_last[2] = g_cs1_last[3] = 0;
    // ---- scratch layout: [scan tile sums | states a, b, last_in | counts | forward-fill tiles | ret^T | lam^T]
    const int64_t tiles = fmk_ceil_div(n, FF_TILE);
    const int64_t max_chunks = fmk_ceil_div(n, CS_CHUNK) + 1;
    const size_t scan_bytes = (((size_t)fmk_ceil_div(max_chunks, FMK_SCAN_TILE) + 1) * 8 + 255) & ~(size_t)255;
    const size_t st_bytes = ((size_t)max_chunks * sizeof(CsState) + 255) & ~(size_t)255;
    const size_t cnt_bytes = ((size_t)(max_chunks + 1) * 8 + 255) & ~(size_t)255;
    const size_t ff_bytes = ((size_t)tiles * 8 + 255) & ~(size_t)255;
    const size_t tr_bytes = ((size_t)((max_chunks + 63) & ~(int64_t)63) * CS_CHUNK * 8 + 255) & ~(size_t)255;   // whole blocks of 64 chunks
    void *scr;
    const size_t act_bytes = ((size_t)max_chunks + 255) & ~(size_t)255;
    const size_t list_bytes = ((size_t)max_chunks * 4 + 255) & ~(size_t)255;
    // the forward fill only needs its tile array; the fixed point's 16 B/tick of scratch are requested when it runs
    FMK_TRY(fmk_scratch(ctx, ff_bytes + (size_t)tiles * 4 + 256, &scr));
    double *tile_last = (double *)scr;
    int *tile_nan = (int *)((char *)scr + ff_bytes);
    unsigned long long *d_first = (unsigned long long *)ctx->d_mail;
    unsigned long long *d_changed = d_first + 1;
    // ---- forward fill of sigma (in place) + first non-NaN index
    const unsigned long long big = ~0ULL;
    int64_t first = 0;
    bool all_nan = false, filled = false;
    auto full_fill = [&]() -> int {
        FMK_TRY(fmk_scratch(ctx, ff_bytes + (size_t)tiles * 4 + 256, &scr));   // (the chain tier may have regrown the scratch)
        tile_last = (double *)scr;
        tile_nan = (int *)((char *)scr + ff_bytes);
        FMK_HIP(ctx, hipMemcpyAsync(d_first, &big, 8, hipMemcpyHostToDevice, ctx->stream));
        k_ff_tile<<<(unsigned)tiles, FF_THREADS, 0, ctx->stream>>>(d_sigma, n, tile_last, d_first, tile_nan);
        FMK_LAUNCH_CHECK(ctx);
        k_ff_count<<<1, 1024, 0, ctx->stream>>>(tile_nan, tiles, (long long *)(ctx->d_mail + 2));
        FMK_LAUNCH_CHECK(ctx);
        FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[0], d_first, 8, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[2], ctx->d_mail + 2, 8, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        first = ctx->h_mail[0];
        all_nan = (unsigned long long)first == big;
        if (!all_nan && ctx->h_mail[2] != first) {                // NaNs after the first valid entry: fill them (logic.py:187-189)
            k_ff_scan_tiles<<<1, 1024, 0, ctx->stream>>>(tile_last, tiles);
            FMK_LAUNCH_CHECK(ctx);
            k_ff_apply<<<(unsigned)tiles, FF_THREADS, 0, ctx->stream>>>(d_sigma, n, tile_last);
            FMK_LAUNCH_CHECK(ctx);
        }
        if (all_nan) first = 0;                                   // all NaN: logic.py:178 keeps index 0
        filled = true;
        return FMK_OK;
    };
    {   // the first valid index from the head alone; NaNs after it are the summary pass's to report (k_ff_head)
        const char *v = getenv("FMK_CUSUM_LAZY_FILL");             // developer knob: 0 = always run the full pass first
        const int64_t n_head = n < ((int64_t)1 << 20) ? n : ((int64_t)1 << 20);
        bool head_ok = false;
        if (!(v && atoi(v) == 0)) {
            FMK_HIP(ctx, hipMemcpyAsync(d_first, &big, 8, hipMemcpyHostToDevice, ctx->stream));
            k_ff_head<<<16, 1024, 0, ctx->stream>>>(d_sigma, n_head, d_first);
            FMK_LAUNCH_CHECK(ctx);
            FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[0], d_first, 8, hipMemcpyDeviceToHost, ctx->stream));
            FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if ((unsigned long long)ctx->h_mail[0] != big) { first = ctx->h_mail[0]; head_ok = true; }
        }
        if (!head_ok) FMK_TRY(full_fill());
    }

    // ---- parallel-in-time fixed point over the chunks of ticks first+1 .. n-1
    const int64_t m = n - (first + 1);
    const int64_t chunks = m > 0 ? fmk_ceil_div(m, CS_CHUNK) : 0;
    int64_t rounds = 0;
    int64_t total = 0;
    int chain_done = 0;
    if (chunks > 0) {                                             // thresholds rarely reached: fmk_cusum_chain.hip
        int nan_seen = 0;
        FMK_TRY(fmk_cusum_chain_tier(ctx, d_ts, d_price, d_sigma, n, first, m, chunks, sigma_floor, sigma_mult, d_out, capacity,
                                     &total, &rounds, &chain_done, filled ? nullptr : &nan_seen));
        if (!filled && !chain_done && nan_seen) {                 // the walk stopped for a NaN in sigma: fill it, once more
            FMK_TRY(full_fill());
            FMK_TRY(fmk_cusum_chain_tier(ctx, d_ts, d_price, d_sigma, n, first, m, chunks, sigma_floor, sigma_mult, d_out,
                                         capacity, &total, &rounds, &chain_done, nullptr));
        }
    } else if (!filled) FMK_TRY(full_fill());
    if (chain_done) {
        if (d_out && capacity < total + 1)
            return fmk_set_error(ctx, FMK_E_CAPACITY, "cusum: %lld close indices, capacity %lld", (long long)(total + 1),
                                 (long long)capacity);
    } else if (chunks > 0) for (;;) {
        // (sigma may still be unfilled: the prep pass then reports a NaN after the first valid index, and this block runs again)
        const bool check_nan = !filled;
        bool redo = false;
        {   // closes every few hundred ticks: one pass over the columns (fmk_cusum_onepass.h); FMK_CUSUM_ONEPASS=0: the fixed point only
            const char *v = getenv("FMK_CUSUM_ONEPASS");
            int one_done = 0;
            g_cs1_last[0] = 0;
            if (!(v && atoi(v) == 0)) {
                FMK_TRY(cs1_run(ctx, d_ts, d_price, d_sigma, n, first, m, sigma_floor, sigma_mult, d_out, capacity, check_nan, &total,
                                &rounds, &one_done, &redo));
                if (redo) { FMK_TRY(full_fill()); continue; }
                if (one_done) break;
            }
        }
        unsigned long long *d_nan = (unsigned long long *)(ctx->d_mail + 5);
        if (check_nan) FMK_HIP(ctx, hipMemsetAsync(d_nan, 0, 8, ctx->stream));
        ctx->h_mail[5] = 0;
        rounds = 0;
        FMK_TRY(fmk_scratch(ctx, scan_bytes + 3 * st_bytes + cnt_bytes + 2 * tr_bytes + act_bytes + list_bytes, &scr));
        char *base = (char *)scr + scan_bytes;
        CsState *st_a = (CsState *)base, *st_b = (CsState *)(base + st_bytes), *last_in = (CsState *)(base + 2 * st_bytes);
        int64_t *counts = (int64_t *)(base + 3 * st_bytes);
        double *t_ret = (double *)(base + 3 * st_bytes + cnt_bytes);
        double *t_lam = (double *)(base + 3 * st_bytes + cnt_bytes + tr_bytes);
        unsigned char *active = (unsigned char *)(base + 3 * st_bytes + cnt_bytes + 2 * tr_bytes);
        int *list = (int *)(base + 3 * st_bytes + cnt_bytes + 2 * tr_bytes + act_bytes);
        const dim3 pg((unsigned)fmk_ceil_div(chunks, CS_PREP_TK), CS_CHUNK / CS_PREP_TJ);
        k_cusum_prep<<<pg, 256, 0, ctx->stream>>>(d_ts, d_price, d_sigma, n, first, m, chunks, sigma_floor, sigma_mult, t_ret,
                                                 t_lam, check_nan ? d_nan : nullptr);
        FMK_LAUNCH_CHECK(ctx);
        const unsigned blocks = (unsigned)fmk_ceil_div(chunks, CS_THREADS);
        CsState *cur = st_a, *prev = st_b;
        k_cusum_chunks<<<blocks, CS_THREADS, 0, ctx->stream>>>(t_ret, t_lam, m, first, chunks, nullptr, cur, nullptr, last_in,
                                                              counts, nullptr, nullptr, nullptr);
        FMK_LAUNCH_CHECK(ctx);
        rounds = 1;
        for (;;) {
            CsState *t = cur; cur = prev; prev = t;               // prev = states of the last round
            FMK_HIP(ctx, hipMemsetAsync(d_changed, 0, 8, ctx->stream));
            k_cusum_chunks<<<blocks, CS_THREADS, 0, ctx->stream>>>(t_ret, t_lam, m, first, chunks, prev, cur, prev, last_in,
                                                                  counts, d_changed, nullptr, nullptr);
            FMK_LAUNCH_CHECK(ctx);
            ++rounds;
            FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[1], d_changed, 8, hipMemcpyDeviceToHost, ctx->stream));
            if (check_nan) FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[5], d_nan, 8, hipMemcpyDeviceToHost, ctx->stream));
            FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (check_nan && ctx->h_mail[5] != 0) { redo = true; break; }
            if (ctx->h_mail[1] == 0) break;                       // fixed point: every chunk started from the truth
            if (rounds > chunks + 2) return fmk_set_error(ctx, FMK_E_HIP, "cusum: fixed point did not converge");
            if (rounds >= 2) {
                // how many dependency chains are left?  (heads: inconsistent chunks with a consistent predecessor)
                FMK_HIP(ctx, hipMemsetAsync(d_changed, 0, 8, ctx->stream));
                k_cusum_mark<<<(unsigned)fmk_ceil_div(chunks, 256), 256, 0, ctx->stream>>>(cur, last_in, chunks, active, list,
                                                                                           d_changed);
                FMK_LAUNCH_CHECK(ctx);
                FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[1], d_changed, 8, hipMemcpyDeviceToHost, ctx->stream));
                FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
                int64_t n_list = ctx->h_mail[1];
                // chunks a wave follows per launch: many walks start from inputs that are not final yet and are redone,
                // so short legs waste little; measured at N = 1e9 in the slow regime: 4..16 within 3 %, 64: +30 %, 512: 3x
                const int walk_steps = 8;
                while (n_list > 0) {                              // sparse phase on the single state array `cur`
                    k_cusum_walk<<<(unsigned)fmk_ceil_div(n_list, 4), 256, 0, ctx->stream>>>(
                        d_ts, d_price, d_sigma, n, first, m, chunks, sigma_floor, sigma_mult, cur, last_in, counts, active,
                        list, n_list, walk_steps);
                    FMK_LAUNCH_CHECK(ctx);
                    ++rounds;
                    if (rounds > chunks + 2) return fmk_set_error(ctx, FMK_E_HIP, "cusum: fixed point did not converge");
                    FMK_HIP(ctx, hipMemsetAsync(d_changed, 0, 8, ctx->stream));
                    k_cusum_mark<<<(unsigned)fmk_ceil_div(chunks, 256), 256, 0, ctx->stream>>>(cur, last_in, chunks, active,
                                                                                               list, d_changed);
                    FMK_LAUNCH_CHECK(ctx);
                    FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[1], d_changed, 8, hipMemcpyDeviceToHost, ctx->stream));
                    FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
                    n_list = ctx->h_mail[1];
                }
                break;
            }
        }
        if (redo) { FMK_TRY(full_fill()); continue; }
        // counts -> offsets (+1 for the opening entry), total
        FMK_TRY(fmk_exclusive_scan_i64(ctx, counts, counts, chunks, true));
        FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[2], counts + chunks, 8, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        total = ctx->h_mail[2];
        if (d_out) {
            if (capacity < total + 1)
                return fmk_set_error(ctx, FMK_E_CAPACITY, "cusum: %lld close indices, capacity %lld", (long long)(total + 1),
                                     (long long)capacity);
            k_cusum_chunks<<<blocks, CS_THREADS, 0, ctx->stream>>>(t_ret, t_lam, m, first, chunks, cur, nullptr, nullptr,
                                                                  nullptr, nullptr, nullptr, counts, d_out + 1);
            FMK_LAUNCH_CHECK(ctx);
        }
        break;
    }
    if (d_out) {
        if (capacity < 1) return fmk_set_error(ctx, FMK_E_CAPACITY, "cusum: capacity 0");
        k_cusum_first<<<1, 1, 0, ctx->stream>>>(d_out, first);   // logic.py:193: the first valid trade opens bar 0
        FMK_LAUNCH_CHECK(ctx);
    }
    if (n_out) *n_out = total + 1;
    if (n_rounds) *n_rounds = rounds;
    return FMK_OK;
}
