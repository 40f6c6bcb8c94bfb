// fmk_cusum_chain.hip -- _cusum_bar_indexer (finmlkit/bar/logic.py:152-221) unless the thresholds are reached every few hundred ticks.
//
// The reference's default sigma_floor (5e-4) on a quiet tape closes a bar once per ~1e5 ticks.  The parallel-in-time fixed
// point of fmk_cusum.hip then needs as many rounds as the state remembers chunks (0.32 s at 1e9 ticks).  This tier uses
// what that regime offers instead: between two closes the loop
//     s_pos = max(0, s_pos + r_i)      s_neg = min(0, s_neg + r_i)                                   (logic.py:203-204)
// is a (max,+) / (min,+) recurrence, and a block of ticks acts on an incoming state s through four numbers per side:
//     exit state          max(A, s + B)         B = sum of r,  A = S_end - min_{1<=k<=end} S_k   (S = prefix sums in the block)
//     "does it close?"    max(P, s + Q) >= 0    P = max_i (S_i - min_{k<=i} S_k - lam_i),  Q = max_i (S_i - lam_i)
// over the ticks i that may close (lam_i = max(mult * sigma_i, floor); not inside a same-timestamp block, logic.py:207-211).
// Blocks compose associatively, so
//   k_cc_summary : one wave per 2048-tick chunk reduces its ticks to {B, A+, P+, Q+, A-, P-, Q-, U = max |S_i|}, per chunk and
//                  per 512-tick sub-block (8 consecutive ticks per lane by 16-byte loads, the lanes composed in order by a DPP
//                  scan): a stream pass, 24 B/tick;
//   k_cc_rate    : how often a side closes even from state 0 in the leading sub-blocks -- the estimate that picks this tier;
//   k_cc_sync / k_cc_pick / k_cc_ranges : chunk boundaries from which a side's state provably does not depend on the past
//                  (see k_cc_sync) cut each side's chain into up to 512 segments;
//   k_cc_walk    : one wave follows a segment: 64 chunk summaries per step (a scan of the exit maps gives every chunk its
//                  incoming state, a ballot the first chunk that may close); the same step over that chunk's four sub-block
//                  summaries names the first sub-block that may close, and only that is opened: its ticks are recomputed from
//                  the columns (by the workgroup's two waves), the walking wave takes 8 of them per lane, runs the reference's
//                  own operations over them from the state the lane scan hands it, the first event is emitted and the side
//                  that closed restarts from 0 at the next tick.
//                  ONLY the side that closed resets (logic.py:214-219), so each side is a chain of its own; k_cc_gather strings
//                  a side's segments together, k_cc_merge interleaves the two sides.  The sides couple in one place -- a
//                  positive close hides a negative one on the same tick (`if / elif`) -- so if the lists share a tick the
//                  joint walk (both sides in one wave, `if / elif` as written, one piece) answers instead.
// Arithmetic.  The block sums are not the reference's sequential float64 sum from the last reset, so every decision
// carries a margin: (ticks since that side's reset + 4096) * 2^-50 * (largest magnitude the side's state or a block
// prefix has reached since) -- 4x the worst-case distance between two float64 evaluation orders of the same recurrence plus
// a 1-ulp difference in log().  A chunk is skipped only when it stays below the threshold by more than the margin, a close is
// accepted only when it exceeds it by more than the margin; anything in between is settled by cc_replay: the side has been
// exactly 0.0 since its last reset, so the reference's own sequence of operations from there gives its state bit for bit.
// Non-finite returns (a price <= 0) are outside the algebra: status BAD, and the caller runs the fixed point of fmk_cusum.hip,
// which is the reference's loop operation for operation; so do an exhausted budget and a replay longer than 2^21 ticks.
// Cost: the summary pass (24 B/tick, 5 ms per 1e9 ticks) + ~8 ns per opened sub-block or event (profiles/r02_cusum_chain.txt).
#include <math.h>
#include <stdlib.h>

#include <vector>

#include "fmk_common.h"
#include "fmk_log.h"
#include "fmk_dpp.h"

#define CC_CHUNK 2048
#define CC_SUB 512                      // ticks a wave of k_cc_summary stages at a time
#define CC_ST_DONE 0
#define CC_ST_BUDGET 1
#define CC_ST_UNCERTAIN 2
#define CC_ST_BAD 3

struct CcSum { double B, U, Ap, Pp, Qp, An, Pn, Qn; };

struct CcState {                         // the walk's state between launches (device memory)
    int64_t chunk;                       // next chunk to look at
    double sp, sn;                       // states entering it
    int64_t reset_p, reset_n;            // tick (t index) of each side's last reset
    double mag_p, mag_n;                 // largest magnitude since
    int64_t n_out;                       // closes emitted so far
    int64_t visits;                      // chunks opened so far
    int32_t status, bad;                 // CC_ST_*; bad: k_cc_summary saw a non-finite return
    int64_t limit;                       // the walk of this state ends before this chunk (segments; otherwise no limit)
    int64_t list_off, list_cap;          // where its closes go in the lists buffer
    int32_t side, pad_;                  // 0: both sides jointly, 1: positive, 2: negative
};

__device__ __forceinline__ CcSum cc_identity()
{
    return CcSum{0.0, 0.0, -INFINITY, -INFINITY, -INFINITY, INFINITY, INFINITY, INFINITY};
}
// (left then right)
__device__ __forceinline__ CcSum cc_compose(const CcSum &l, const CcSum &r)
{
    CcSum o;
    o.B = l.B + r.B;
    o.U = fmax(l.U, fabs(l.B) + r.U);
    o.Ap = fmax(r.Ap, l.Ap + r.B);
    o.Pp = fmax(fmax(l.Pp, r.Pp), l.Ap + r.Qp);
    o.Qp = fmax(l.Qp, l.B + r.Qp);
    o.An = fmin(r.An, l.An + r.B);
    o.Pn = fmin(fmin(l.Pn, r.Pn), l.An + r.Qn);
    o.Qn = fmin(l.Qn, l.B + r.Qn);
    return o;
}
__device__ __forceinline__ double cc_shfl_down(double v, int d) { return __shfl_down(v, d, 64); }
__device__ __forceinline__ double cc_shfl_up(double v, int d) { return __shfl_up(v, d, 64); }
__device__ __forceinline__ double cc_bcast(double v, int src)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)((unsigned long long)b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__device__ __forceinline__ int64_t cc_bcast_i64(int64_t v, int src)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)((unsigned long long)v >> 32), src);
    return (int64_t)(((unsigned long long)hi << 32) | lo);
}

// r_i and lam_i of tick i = first + 1 + t: the expressions of k_cusum_prep (fmk_cusum.hip), NaN lam = "cannot close"
__device__ __forceinline__ void cc_tick(double p, double pm, double sg, int64_t tsi, int64_t tsn, bool has_next,
                                        double sigma_floor, double sigma_mult, double *r, double *lam)
{
    *r = fmk_log_ratio(p, pm);
    double l = NAN;
    if (!(has_next && tsi == tsn)) {
        l = sigma_mult * sg;
        l = sigma_floor > l ? sigma_floor : l;
    }
    *lam = l;
}

// ordered composition over the lanes on the DPP path (lane 63 ends up with lanes 0 .. 63 composed left to right; the order of
// fmk_dpp_iscan).  The first version was a shuffle-down tree: 96 ds_bpermute per 512 ticks and their LDS round trips.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ CcSum cc_sum_dpp(const CcSum &v)
{
    return CcSum{fmk_dpp<CTRL, ROW_MASK>(0.0, v.B), fmk_dpp<CTRL, ROW_MASK>(0.0, v.U),
                 fmk_dpp<CTRL, ROW_MASK>((double)-INFINITY, v.Ap), fmk_dpp<CTRL, ROW_MASK>((double)-INFINITY, v.Pp),
                 fmk_dpp<CTRL, ROW_MASK>((double)-INFINITY, v.Qp), fmk_dpp<CTRL, ROW_MASK>((double)INFINITY, v.An),
                 fmk_dpp<CTRL, ROW_MASK>((double)INFINITY, v.Pn), fmk_dpp<CTRL, ROW_MASK>((double)INFINITY, v.Qn)};
}
__device__ __forceinline__ CcSum cc_sum_iscan(CcSum v)
{
    v = cc_compose(cc_sum_dpp<FMK_DPP_ROW_SHR(1), 0xF>(v), v);
    v = cc_compose(cc_sum_dpp<FMK_DPP_ROW_SHR(2), 0xF>(v), v);
    v = cc_compose(cc_sum_dpp<FMK_DPP_ROW_SHR(4), 0xF>(v), v);
    v = cc_compose(cc_sum_dpp<FMK_DPP_ROW_SHR(8), 0xF>(v), v);
    v = cc_compose(cc_sum_dpp<FMK_DPP_ROW_BCAST15, 0xA>(v), v);
    v = cc_compose(cc_sum_dpp<FMK_DPP_ROW_BCAST31, 0xC>(v), v);
    return v;
}

// ---------------------------------------------------------------------------------------
// chunk summaries
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cc_summary(const int64_t *__restrict__ ts, const double *__restrict__ price,
                                                    const double *__restrict__ sigma, int64_t n, int64_t first, int64_t m,
                                                    int64_t chunk_lo, int64_t chunk_hi, int64_t chunks, double sigma_floor,
                                                    double sigma_mult, double *__restrict__ sums, double *__restrict__ subs,
                                                    CcState *state)
{
    const int lane = fmk_lane(), w = (int)(threadIdx.x >> 6);
    const int64_t k = chunk_lo + (int64_t)blockIdx.x * 4 + w;
    if (k >= chunk_hi) return;
    const int64_t t0 = k * CC_CHUNK;
    CcSum acc = cc_identity();
    bool bad = false, nan_sigma = false;
    typedef double cc_d2 __attribute__((ext_vector_type(2), aligned(8)));       // 16-byte loads on an 8-byte alignment promise
    typedef long long cc_l2 __attribute__((ext_vector_type(2), aligned(8)));
    for (int sub = 0; sub < CC_CHUNK / CC_SUB; ++sub) {
        // the lane's 8 consecutive ticks straight from the columns (price, sigma, timestamp: 4 x 16 bytes each, plus the price
        // before and the timestamp after them).  A first version loaded coalesced rows and handed them over through an LDS tile:
        // 36 KB per workgroup, 7.0 ms per 1e9 ticks (profiles/r02_ewmst_direct_loads.txt has the same lesson).
        const int64_t tq = t0 + sub * CC_SUB + 8 * lane;              // t of the lane's first tick
        const int64_t i0 = first + 1 + tq;
        double p[8], sg[8], pm0;
        int64_t a[8], an;
        if (i0 + 8 <= n - 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const cc_d2 vp = *(const cc_d2 *)(price + i0 + 2 * q);
                const cc_d2 vs = *(const cc_d2 *)(sigma + i0 + 2 * q);
                const cc_l2 vt = *(const cc_l2 *)(ts + i0 + 2 * q);
                p[2 * q] = vp.x; p[2 * q + 1] = vp.y; sg[2 * q] = vs.x; sg[2 * q + 1] = vs.y; a[2 * q] = vt.x; a[2 * q + 1] = vt.y;
            }
            pm0 = price[i0 - 1];
            an = ts[i0 + 8];
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                int64_t i = i0 + q;
                if (i > n - 1) i = n - 1;
                p[q] = price[i]; sg[q] = sigma[i]; a[q] = ts[i];
            }
            pm0 = price[(i0 - 1 > n - 1 ? n - 1 : i0 - 1)];
            an = ts[(i0 + 8 > n - 1 ? n - 1 : i0 + 8)];
        }
        CcSum me;
        {
            double S = 0.0, mn = INFINITY, mx = -INFINITY, U = 0.0, Pp = -INFINITY, Qp = -INFINITY, Pn = INFINITY, Qn = INFINITY;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                double r = 0.0, lam = NAN;
                if (tq + q < m) {
                    const int64_t i = i0 + q;
                    cc_tick(p[q], q == 0 ? pm0 : p[q > 0 ? q - 1 : 0], sg[q], a[q], q == 7 ? an : a[q < 7 ? q + 1 : 7], i + 1 < n,
                            sigma_floor, sigma_mult, &r, &lam);
                    bad |= !(fabs(r) < INFINITY);
                    nan_sigma |= sg[q] != sg[q];
                }
                S += r;
                mn = fmin(mn, S); mx = fmax(mx, S);
                U = fmax(U, fabs(S));
                const double vp = (S - mn) - lam, vq = S - lam, vn = (S - mx) + lam, vqn = S + lam;
                Pp = vp > Pp ? vp : Pp;                       // a NaN lam (cannot close) never enters
                Qp = vq > Qp ? vq : Qp;
                Pn = vn < Pn ? vn : Pn;
                Qn = vqn < Qn ? vqn : Qn;
            }
            me = CcSum{S, U, S - mn, Pp, Qp, S - mx, Pn, Qn};
        }
        me = cc_sum_iscan(me);                                // lane 63: the sub-block's 64 lane blocks composed in order
        if (lane == 63) {                                     // the 512-tick sub-block's own summary (the walk opens sub-blocks)
            const int64_t q = k * (CC_CHUNK / CC_SUB) + sub, nq = chunks * (CC_CHUNK / CC_SUB);
            subs[0 * nq + q] = me.B; subs[1 * nq + q] = me.U;
            subs[2 * nq + q] = me.Ap; subs[3 * nq + q] = me.Pp; subs[4 * nq + q] = me.Qp;
            subs[5 * nq + q] = me.An; subs[6 * nq + q] = me.Pn; subs[7 * nq + q] = me.Qn;
        }
        acc = cc_compose(acc, me);                            // meaningful on lane 63
    }
    if (lane == 63) {
        sums[0 * chunks + k] = acc.B; sums[1 * chunks + k] = acc.U;
        sums[2 * chunks + k] = acc.Ap; sums[3 * chunks + k] = acc.Pp; sums[4 * chunks + k] = acc.Qp;
        sums[5 * chunks + k] = acc.An; sums[6 * chunks + k] = acc.Pn; sums[7 * chunks + k] = acc.Qn;
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0 && !(__atomic_load_n(&state->bad, __ATOMIC_RELAXED) & 1))
        atomicOr(&state->bad, 1);
    if (__builtin_amdgcn_ballot_w64(nan_sigma) != 0 && lane == 0 && !(__atomic_load_n(&state->bad, __ATOMIC_RELAXED) & 2))
        atomicOr(&state->bad, 2);                                     // (matters only while sigma is not forward filled)
}

// states: [0] positive side, [1] negative side, [2 .. 2 + 2 (K - 1)) the later segments of the two sides (k_cc_ranges), then
// the joint walk
__global__ void k_cc_init(CcState *st, int n_states, int64_t list_cap, int64_t *flag)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i == 0) *flag = 0;
    if (i >= n_states) return;
    CcState *q = st + i;
    q->chunk = 0; q->sp = 0.0; q->sn = 0.0; q->reset_p = 0; q->reset_n = 0; q->mag_p = 0.0; q->mag_n = 0.0;
    q->n_out = 0; q->visits = 0; q->status = CC_ST_DONE; q->bad = 0;
    q->limit = i < 2 || i == n_states - 1 ? INT64_MAX : 0;           // segments are inactive until k_cc_ranges says otherwise
    q->list_off = i == 1 ? list_cap : 0; q->list_cap = list_cap;
    q->side = i == n_states - 1 ? 0 : i < 2 ? i + 1 : 0; q->pad_ = 0;
}

// ---------------------------------------------------------------------------------------
// the walk along the chain: wave 0 walks, the other waves of the workgroup only help to open a chunk
// ---------------------------------------------------------------------------------------
// The reference's operations for one side from a tick where its state is exactly 0.0 (t_from: the first tick after a reset,
// the start of the stream or of a segment) through t_to: s = max(0.0, s + ret) or min(0.0, s + ret), tick by tick.  The
// wave computes 64 returns at a time; the fold over them is wave-uniform (~30 cycles per tick).
#define CC_REPLAY_MAX ((int64_t)1 << 21)
__device__ __forceinline__ double cc_replay(const double *__restrict__ price, int64_t first, int64_t t_from, int64_t t_to, bool positive)
{
    const int lane = fmk_lane();
    double s = 0.0;
    for (int64_t base = t_from; base <= t_to; base += 64) {
        const int64_t t = base + lane, i = first + 1 + (t <= t_to ? t : t_to);
        const double r = fmk_log_ratio(price[i], price[i - 1]);
        const int cnt = (int)(t_to - base + 1 < 64 ? t_to - base + 1 : 64);
        for (int k = 0; k < cnt; ++k) {
            const double rk = cc_bcast(r, k);
            s = positive ? fmax(0.0, s + rk) : fmin(0.0, s + rk);
        }
    }
    return s;
}

#define CC_PAD(j) ((j) + ((j) >> 5))
#define CC_WALK_WAVES 2
struct CcMap { double B, Ap, An; };                                   // exit maps of the two sides: max(Ap, s + B), min(An, s + B)
__device__ __forceinline__ CcMap cc_map_compose(const CcMap &l, const CcMap &r)
{
    return CcMap{l.B + r.B, fmax(r.Ap, l.Ap + r.B), fmin(r.An, l.An + r.B)};
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ CcMap cc_map_dpp(const CcMap &v)
{
    return CcMap{fmk_dpp<CTRL, ROW_MASK>(0.0, v.B), fmk_dpp<CTRL, ROW_MASK>((double)-INFINITY, v.Ap),
                 fmk_dpp<CTRL, ROW_MASK>((double)INFINITY, v.An)};
}
// inclusive scan over the lanes on the DPP path (the order of fmk_dpp_iscan; a bpermute scan of three doubles cost ~2000
// cycles of dependent LDS-crossbar latency per step of the walk)
__device__ __forceinline__ CcMap cc_map_iscan(CcMap v)
{
    v = cc_map_compose(cc_map_dpp<FMK_DPP_ROW_SHR(1), 0xF>(v), v);
    v = cc_map_compose(cc_map_dpp<FMK_DPP_ROW_SHR(2), 0xF>(v), v);
    v = cc_map_compose(cc_map_dpp<FMK_DPP_ROW_SHR(4), 0xF>(v), v);
    v = cc_map_compose(cc_map_dpp<FMK_DPP_ROW_SHR(8), 0xF>(v), v);
    v = cc_map_compose(cc_map_dpp<FMK_DPP_ROW_BCAST15, 0xA>(v), v);
    v = cc_map_compose(cc_map_dpp<FMK_DPP_ROW_BCAST31, 0xC>(v), v);
    return v;
}
__device__ __forceinline__ CcMap cc_map_exclusive(const CcMap &inc)
{
    return CcMap{fmk_dpp_shift_up1(inc.B, 0.0), fmk_dpp_shift_up1(inc.Ap, (double)-INFINITY),
                 fmk_dpp_shift_up1(inc.An, (double)INFINITY)};
}

__global__ __launch_bounds__(64 * CC_WALK_WAVES, 2) void k_cc_walk(const int64_t *__restrict__ ts, const double *__restrict__ price,
                                                const double *__restrict__ sigma, int64_t n, int64_t first, int64_t m,
                                                int64_t chunk_limit, int64_t chunks, double sigma_floor, double sigma_mult,
                                                const double *__restrict__ sums, const double *__restrict__ subs,
                                                CcState *states, const CcState *st0, int bad_mask, int64_t visit_budget,
                                                double margin_scale, int64_t *__restrict__ all_lists,
                                                int64_t *__restrict__ closes, int64_t capacity)
{
    // joint != 0: one workgroup evaluates the loop as written (both sides, `if / elif`), closes -> closes[1 + ...].
    // joint == 0: workgroup 0 follows the positive side alone, workgroup 1 the negative side alone (each side's state after
    // its own close is 0 whatever the other side does, so the two chains are independent -- except that a positive close
    // hides a negative one on the same tick; the caller merges the two lists and re-runs jointly if they ever share a tick).
    // Every workgroup has a state of its own (`states + blockIdx.x`): a side from the start, a later SEGMENT of a side (from a
    // chunk boundary where k_cc_sync proved that the side's state no longer depends on the past), or the joint walk.
    CcState *state = states + blockIdx.x;
    const int side = state->side;
    const int joint = side == 0;
    const bool do_p = side != 2, do_n = side != 1;
    if (state->limit < chunk_limit) chunk_limit = state->limit;
    if (state->chunk >= chunk_limit) return;                            // (all waves: nothing to do, e.g. an unused segment)
    int64_t *lists = all_lists + state->list_off;
    const int64_t list_cap = state->list_cap;
    __shared__ double s_r[CC_SUB + CC_SUB / 8], s_l[CC_SUB + CC_SUB / 8];
    __shared__ int64_t s_cmd;                                           // sub-block (4 * chunk + sub) to open, -1: the walk is over
    const int lane = fmk_lane(), wv = (int)(threadIdx.x >> 6);
    // r / lam of the sub-block's rows wv, wv + WAVES, ... into LDS (all waves; the expressions of k_cusum_prep)
    auto open_rows = [&](int64_t q) {
        const int64_t tq = (q >> 2) * CC_CHUNK + (q & 3) * CC_SUB;      // t of the sub-block's first tick
        const int64_t left = m - tq;
        const int len = (int)(left < CC_SUB ? (left > 0 ? left : 0) : CC_SUB);
        constexpr int RW = CC_SUB / 64 / CC_WALK_WAVES;
        double p[RW], pm[RW], sg[RW];
        int64_t a[RW], b[RW];
#pragma unroll
        for (int g = 0; g < RW; ++g) {
            int64_t i = first + 1 + tq + 64 * (wv + CC_WALK_WAVES * g) + lane;
            if (i > n - 1) i = n - 1;
            p[g] = price[i]; pm[g] = price[i - 1]; sg[g] = sigma[i]; a[g] = ts[i]; b[g] = ts[i + 1 < n ? i + 1 : i];
        }
#pragma unroll
        for (int g = 0; g < RW; ++g) {
            const int j = 64 * (wv + CC_WALK_WAVES * g) + lane;
            double r = 0.0, lam = NAN;
            if (j < len) cc_tick(p[g], pm[g], sg[g], a[g], b[g], first + 1 + tq + j + 1 < n, sigma_floor, sigma_mult, &r, &lam);
            s_r[j + (j >> 3)] = r;
            s_l[j + (j >> 3)] = lam;
        }
    };
    if (wv != 0) {                                                      // helpers: two barriers per opened sub-block
        for (;;) {
            __syncthreads();
            const int64_t q = s_cmd;
            if (q < 0) return;
            open_rows(q);
            __syncthreads();
        }
    }
    int64_t c = state->chunk;
    double sp = state->sp, sn = state->sn, mag_p = state->mag_p, mag_n = state->mag_n;
    int64_t reset_p = state->reset_p, reset_n = state->reset_n, n_out = state->n_out, visits = state->visits;
    int status = (st0->bad & bad_mask) ? CC_ST_BAD : CC_ST_DONE;       // k_cc_summary reports there
    const double eps = margin_scale * 8.881784197001252e-16;              // 2^-50

    auto load = [&](int64_t c0) -> CcSum {
        const int64_t k = c0 + lane;
        if (k >= chunk_limit) return cc_identity();
        return CcSum{sums[0 * chunks + k], sums[1 * chunks + k], sums[2 * chunks + k], sums[3 * chunks + k],
                     sums[4 * chunks + k], sums[5 * chunks + k], sums[6 * chunks + k], sums[7 * chunks + k]};
    };
    // Summaries are loaded CC_NG groups at a time and nothing stays in flight across the loop's back edge (the compiler waits
    // for every outstanding load there: one group ahead cost the full ~3600 cycles of latency per step).  A bar of the slow
    // regime spans about two groups, so the batch issued before a chunk is opened usually reaches the next one.
    constexpr int CC_NG = 4;
    CcSum grp[CC_NG];
    auto load_batch = [&](int64_t c0) {
#pragma unroll
        for (int g = 0; g < CC_NG; ++g) grp[g] = load(c0 + 64 * g);
    };
    load_batch(c);
#ifdef CC_TIMING
    long long tG = 0, tL = 0, tW = 0, tc0 = __builtin_readcyclecounter(), tc1;
    long long nG = 0, nW = 0;
#define CC_TICK(acc) { tc1 = __builtin_readcyclecounter(); acc += tc1 - tc0; tc0 = tc1; }
#else
#define CC_TICK(acc)
#endif
    while (c < chunk_limit && status == CC_ST_DONE) {
        int64_t k = -1;                                                   // the first chunk that may close
        double U_k = 0.0, k_sp = 0.0, k_sn = 0.0;
#pragma unroll
        for (int g = 0; g < CC_NG; ++g) {
            if (k >= 0 || c >= chunk_limit) continue;
            const CcSum cur = grp[g];
            const CcMap inc = cc_map_iscan(CcMap{cur.B, cur.Ap, cur.An});
            const CcMap exc = cc_map_exclusive(inc);
            const double in_p = fmax(exc.Ap, sp + exc.B), in_n = fmin(exc.An, sn + exc.B);   // states entering this lane's chunk
            // magnitudes and margins of the group (the largest of any lane: conservative for the earlier ones)
            const double mg_p = fmax(mag_p, fmk_dpp_reduce(fabs(in_p) + 2.0 * cur.U, 0.0, FmkOpMax()));
            const double mg_n = fmax(mag_n, fmk_dpp_reduce(fabs(in_n) + 2.0 * cur.U, 0.0, FmkOpMax()));
            const int64_t t_end = (c + 64) * CC_CHUNK;
            const double mar_p = (double)(t_end - reset_p + 4096) * eps * mg_p;
            const double mar_n = (double)(t_end - reset_n + 4096) * eps * mg_n;
            const bool cand = (do_p && fmax(cur.Pp, in_p + cur.Qp) >= -mar_p) || (do_n && fmin(cur.Pn, in_n + cur.Qn) <= mar_n);
            const unsigned long long cb = __builtin_amdgcn_ballot_w64(cand);
            mag_p = mg_p; mag_n = mg_n;                                   // incl. the chunks skipped on the way to a candidate
            if (cb == 0) {
                const int64_t left = chunk_limit - c;
                const int last = left >= 64 ? 63 : (int)left - 1;
                const double oB = cc_bcast(inc.B, last), oAp = cc_bcast(inc.Ap, last), oAn = cc_bcast(inc.An, last);
                sp = fmax(oAp, sp + oB); sn = fmin(oAn, sn + oB);
                c = left >= 64 ? c + 64 : chunk_limit;                    // (the next launch resumes at `c`)
#ifdef CC_TIMING
                ++nG;
#endif
            } else {
                const int f = __builtin_ctzll(cb);
                k = c + f;
                k_sp = cc_bcast(in_p, f); k_sn = cc_bcast(in_n, f);
                U_k = cc_bcast(cur.U, f);
            }
        }
        if (k < 0) { load_batch(c); CC_TICK(tG) continue; }
        CC_TICK(tG)
        // ---- open it
        (void)U_k;
        sp = k_sp; sn = k_sn;
        c = k;
        load_batch(k + 1);                                                // the groups after this chunk, behind the work below
        const int64_t t0 = k * CC_CHUNK;
        // The chunk's four 512-tick sub-blocks have summaries of their own (k_cc_summary): the same step as above over them finds
        // the first sub-block that may close, and only that one is opened -- 8 ticks per lane instead of 32, 8 rows of log()
        // instead of 32.  After its events the walk goes on over the remaining sub-blocks' summaries.
        CcSum sb = cc_identity();
        {
            const int64_t nq = chunks * (CC_CHUNK / CC_SUB), q = k * (CC_CHUNK / CC_SUB) + lane;
            if (lane < CC_CHUNK / CC_SUB)
                sb = CcSum{subs[0 * nq + q], subs[1 * nq + q], subs[2 * nq + q], subs[3 * nq + q],
                           subs[4 * nq + q], subs[5 * nq + q], subs[6 * nq + q], subs[7 * nq + q]};
        }
        int s_from = 0;                                                   // first sub-block not yet passed
        while (s_from < CC_CHUNK / CC_SUB && status == CC_ST_DONE) {
            CcMap mm{sb.B, sb.Ap, sb.An};
            if (lane < s_from || lane >= CC_CHUNK / CC_SUB) mm = CcMap{0.0, -INFINITY, INFINITY};
            const CcMap sinc = cc_map_iscan(mm);
            const CcMap sexc = cc_map_exclusive(sinc);
            const double si_p = fmax(sexc.Ap, sp + sexc.B), si_n = fmin(sexc.An, sn + sexc.B);   // states entering the sub-block
            const bool mine = lane >= s_from && lane < CC_CHUNK / CC_SUB;
            const double sg_p = fmax(mag_p, fmk_dpp_reduce(mine ? fabs(si_p) + 2.0 * sb.U : 0.0, 0.0, FmkOpMax()));
            const double sg_n = fmax(mag_n, fmk_dpp_reduce(mine ? fabs(si_n) + 2.0 * sb.U : 0.0, 0.0, FmkOpMax()));
            const double sm_p = (double)(t0 + CC_CHUNK - reset_p + 4096) * eps * sg_p;
            const double sm_n = (double)(t0 + CC_CHUNK - reset_n + 4096) * eps * sg_n;
            const bool scand = mine && ((do_p && fmax(sb.Pp, si_p + sb.Qp) >= -sm_p) || (do_n && fmin(sb.Pn, si_n + sb.Qn) <= sm_n));
            const unsigned long long sc = __builtin_amdgcn_ballot_w64(scand);
            mag_p = sg_p; mag_n = sg_n;
            if (sc == 0) {                                                // the rest of the chunk passes without an event
                const double oB = cc_bcast(sinc.B, CC_CHUNK / CC_SUB - 1), oAp = cc_bcast(sinc.Ap, CC_CHUNK / CC_SUB - 1);
                const double oAn = cc_bcast(sinc.An, CC_CHUNK / CC_SUB - 1);
                sp = fmax(oAp, sp + oB); sn = fmin(oAn, sn + oB);
                break;
            }
            const int sf = __builtin_ctzll(sc);
            sp = cc_bcast(si_p, sf); sn = cc_bcast(si_n, sf);
            const double U_s = cc_bcast(sb.U, sf);
            if (visits >= visit_budget) { status = CC_ST_BUDGET; break; }
            ++visits;
            const int64_t ts0 = t0 + (int64_t)sf * CC_SUB;                // t of the sub-block's first tick
            if (lane == 0) s_cmd = k * (CC_CHUNK / CC_SUB) + sf;
            __syncthreads();
            open_rows(k * (CC_CHUNK / CC_SUB) + sf);
            __syncthreads();
            // the lane's 8 consecutive ticks, in registers for the passes below
            double rr[8], ll[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { rr[q] = s_r[9 * lane + q]; ll[q] = s_l[9 * lane + q]; }
            CC_TICK(tL)
            // Ticks that are decided (up to and including an event) are neutralised: whole lanes through `dead`, the event's own
            // lane by r = 0 / lam = NaN in its registers.  A block's minimum / maximum prefix includes the empty prefix here (mn,
            // mx start at 0): that only adds "reset before the block's first tick" to the exit map, max(A, B, s + B), which is
            // max(A, s + B) for every state the positive side can have (s >= 0; mirrored for the negative side) -- and it makes
            // r = 0 ticks and dead lanes exact identities, so the passes need no per-tick predicates.
            int dead = 0;
            for (;;) {
                // (1) each lane's exit map, scanned over the lanes
                double S = 0.0, mn = 0.0, mx = 0.0;
#pragma unroll
                for (int q = 0; q < 8; ++q) { S += rr[q]; mn = fmin(mn, S); mx = fmax(mx, S); }
                CcMap me{S, S - mn, S - mx};
                if (lane < dead) me = CcMap{0.0, 0.0, 0.0};
                const CcMap lexc = cc_map_exclusive(cc_map_iscan(me));
                const double lp = fmax(lexc.Ap, sp + lexc.B), ln = fmin(lexc.An, sn + lexc.B);   // states entering the lane's ticks
                // (2a) the reference's loop over the lane's ticks: which ticks come within the margins of a threshold (or beyond)?
                const double mp = (double)(ts0 + CC_SUB - reset_p + 4096) * eps * mag_p;
                const double mq = (double)(ts0 + CC_SUB - reset_n + 4096) * eps * mag_n;
                unsigned mask = 0;
                double ap = lp, an = ln;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    ap = fmax(ap + rr[q], 0.0);                           // max(0.0, s_pos + ret), min(0.0, s_neg + ret)
                    an = fmin(an + rr[q], 0.0);
                    mask |= ((do_p && ap - ll[q] >= -mp) || (do_n && an + ll[q] <= mq)) ? 1u << q : 0u;
                }
                if (lane < dead) mask = 0;
                const unsigned long long eb = __builtin_amdgcn_ballot_w64(mask != 0);
                if (eb == 0) { sp = cc_bcast(ap, 63); sn = cc_bcast(an, 63); break; }
                // (2b) the first such tick: the states there (wave-uniform trip count: no predicates)
                const int fl = __builtin_ctzll(eb);
                const int q0 = __builtin_ctz((unsigned)__builtin_amdgcn_readlane((int)mask, fl));
                double bp = lp, bn = ln, lamq = NAN;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q <= q0) { bp = fmax(bp + rr[q], 0.0); bn = fmin(bn + rr[q], 0.0); lamq = ll[q]; }
                const double dp = bp - lamq, dn = bn + lamq;
                // `if s_pos >= lam ... elif s_neg <= -lam`, each answer only when it is beyond the margin
                const int kind_l = (do_p && dp >= mp) ? 1 : ((!do_p || dp < -mp) && do_n && dn <= -mq) ? 2 : 3;
                int kind = __builtin_amdgcn_readlane(kind_l, fl);
                const int j = 8 * fl + q0;
                if (kind == 3) {
                    // Inside the margin: block sums cannot tell.  The side's state has been exactly 0.0 since its last reset,
                    // so the reference's own sequence of operations from there gives its state at this tick bit for bit
                    // (cc_replay), and `if s_pos >= lam ... elif s_neg <= -lam` is evaluated as written.  Tapes whose
                    // log-prices sit on a lattice meet round floors EXACTLY (the synthetic one: s = 2.0000000000054e-05
                    // against a floor of 2e-5, again and again); on others this is a once-in-1e11 event.
                    const double lam_e = cc_bcast(lamq, fl);
                    const int64_t te = ts0 + j;
                    kind = 0;
                    if (do_p) {
                        if (te - reset_p >= CC_REPLAY_MAX) { status = CC_ST_UNCERTAIN; break; }
                        if (cc_replay(price, first, reset_p, te, true) >= lam_e) kind = 1;
                        visits += (te - reset_p) >> 10;                   // the budget also bounds the replayed ticks
                    }
                    if (kind == 0 && do_n) {
                        if (te - reset_n >= CC_REPLAY_MAX) { status = CC_ST_UNCERTAIN; break; }
                        if (cc_replay(price, first, reset_n, te, false) <= -lam_e) kind = 2;
                        visits += (te - reset_n) >> 10;
                    }
                    ++visits;
                }
                if (kind != 0) {
                    if (joint) { if (lane == 0 && closes && 1 + n_out < capacity) closes[1 + n_out] = first + 1 + ts0 + j; }
                    else {
                        if (n_out >= list_cap) { status = CC_ST_BUDGET; break; }
                        if (lane == 0) lists[n_out] = first + 1 + ts0 + j;
                    }
                    ++n_out; ++visits;                                    // an event costs about as much as opening a sub-block
                }
                sp = kind == 1 ? 0.0 : cc_bcast(bp, fl);                  // states after that tick, the closing side reset
                sn = kind == 2 ? 0.0 : cc_bcast(bn, fl);
                if (kind == 1) { reset_p = ts0 + j + 1; mag_p = 2.0 * U_s; }            // (first tick after the reset)
                else if (kind == 2) { reset_n = ts0 + j + 1; mag_n = 2.0 * U_s; }
                mag_p = fmax(mag_p, fabs(sp)); mag_n = fmax(mag_n, fabs(sn));
                dead = fl;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q <= q0) { rr[q] = lane == fl ? 0.0 : rr[q]; ll[q] = lane == fl ? (double)NAN : ll[q]; }
            }
            s_from = sf + 1;
        }
        if (status != CC_ST_DONE) break;
        c = k + 1;
#ifdef CC_TIMING
        ++nW;
#endif
        CC_TICK(tW)
    }
    if (lane == 0) s_cmd = -1;
    __syncthreads();
#ifdef CC_TIMING
    if (lane == 0) printf("k_cc_walk: %lld group steps %lld cycles; %lld chunks opened: load + log %lld cycles, walk %lld cycles\n", nG, tG, nW, tL, tW);
#endif
    if (lane == 0) {
        state->chunk = c; state->sp = sp; state->sn = sn; state->mag_p = mag_p; state->mag_n = mag_n;
        state->reset_p = reset_p; state->reset_n = reset_n; state->n_out = n_out; state->visits = visits;
        state->status = status;
    }
}

// How often are the thresholds reached?  Sub-blocks (512 ticks) of the leading chunks in which a side closes even when it
// enters with state 0 (P >= 0): a lower bound of the closes that needs no walk.  *count += such (sub-block, side) pairs.
__global__ __launch_bounds__(256) void k_cc_rate(const double *__restrict__ subs, int64_t nq_all, int64_t nq, unsigned long long *count)
{
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int c = 0;
    if (q < nq) c = (subs[3 * nq_all + q] >= 0.0 ? 1 : 0) + (subs[6 * nq_all + q] <= 0.0 ? 1 : 0);
    c = fmk_wave_sum(c);
    if (fmk_lane() == 0 && c) atomicAdd(count, (unsigned long long)c);
}

// ---------------------------------------------------------------------------------------
// segments: chunk boundaries from which a side's chain does not depend on the past
// ---------------------------------------------------------------------------------------
// A side's state is forgotten whenever it is clamped to 0.0 (or closes).  Take a chunk boundary c.  Whatever happened before,
// the positive state leaving chunk c - 1 is below  s* = max(A, B - Q)  of that chunk: a trajectory that did not close inside
// it entered with s < min_i(lam_i - S_i) = -Q and leaves with max(A, s + B); one that did close restarted from 0 at its last
// close and leaves with at most S_end - min S = A.  If now, over the chunks c, c + 1, ..., (1) no trajectory starting below
// s* can close -- s* + max_i(S_i - lam_i) < 0 -- until (2) the one starting AT s* has been clamped -- s* + min_i S_i < 0 --
// then every trajectory, the true one included, was clamped to exactly 0.0 in that window without a close, and from its own
// clamp on it is BIT-identical to the trajectory that starts at the boundary with state 0.0 (float addition is monotone, so
// that one, being the smallest, is 0.0 wherever a larger one is).  A walk started at such a boundary with state 0 therefore
// emits the reference's closes, all of them, without knowing anything about the ticks before -- and the side's chain can
// be walked by many workgroups at once.  Both inequalities carry the walk's margin (block sums against the reference's
// sequential sums).  Mirrored for the negative side.  flags[side * chunks + c] = 1: boundary c is such a start.
#define CC_SYNC_MAX 96                                                  // chunks a window may span
__global__ __launch_bounds__(256) void k_cc_sync(const double *__restrict__ sums, int64_t chunks, int64_t lo, double margin_scale,
                                                 unsigned char *__restrict__ flags)
{
    const int64_t c = lo + 1 + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= chunks) return;
    auto load = [&](int64_t k) {
        return CcSum{sums[0 * chunks + k], sums[1 * chunks + k], sums[2 * chunks + k], sums[3 * chunks + k],
                     sums[4 * chunks + k], sums[5 * chunks + k], sums[6 * chunks + k], sums[7 * chunks + k]};
    };
    const double eps = margin_scale * 8.881784197001252e-16;              // 2^-50
    const CcSum prev = load(c - 1);
    double sp = fmax(prev.Ap, prev.B - prev.Qp), sn = fmin(prev.An, prev.B - prev.Qn);    // bounds of the states at the boundary
    const double m0 = (double)(CC_CHUNK + 4096) * eps;
    sp += m0 * (fabs(sp) + 2.0 * prev.U);
    sn -= m0 * (fabs(sn) + 2.0 * prev.U);
    int dp = sp < INFINITY ? 0 : 2, dn = sn > -INFINITY ? 0 : 2;          // 0 open, 1 proven, 2 failed
    if (!(sp >= 0.0)) dp = 2;                                             // (NaN summaries: never)
    if (!(sn <= 0.0)) dn = 2;
    CcSum acc = cc_identity();
    for (int64_t j = c; j < chunks && j < c + CC_SYNC_MAX && (dp == 0 || dn == 0); ++j) {
        acc = cc_compose(acc, load(j));
        const double ticks = (double)((j + 1 - c) * CC_CHUNK + 4096);
        if (dp == 0) {
            const double mar = ticks * eps * (sp + 2.0 * acc.U);
            if (!(sp + acc.Qp < -mar)) dp = 2;                            // something may close before everything is clamped
            else if (sp + (acc.B - acc.Ap) < -mar) dp = 1;                // min_i S_i = B - A
        }
        if (dn == 0) {
            const double mar = ticks * eps * (-sn + 2.0 * acc.U);
            if (!(sn + acc.Qn > mar)) dn = 2;
            else if (sn + (acc.B - acc.An) > mar) dn = 1;                 // max_i S_i = B - A
        }
    }
    flags[c] = dp == 1;
    flags[chunks + c] = dn == 1;
}

// segment k (1 .. K - 1) of a side starts at the first such boundary in its nominal range [lo + k w, lo + (k + 1) w), if any
__global__ __launch_bounds__(64) void k_cc_pick(const unsigned char *__restrict__ flags, int64_t chunks, int64_t lo, int K,
                                                int64_t *__restrict__ starts)
{
    const int side = (int)blockIdx.x / (K - 1), k = 1 + (int)blockIdx.x % (K - 1), lane = fmk_lane();
    const int64_t w = (chunks - lo) / K, a = lo + k * w, b = k == K - 1 ? chunks : a + w;
    int64_t found = -1;
    for (int64_t c0 = a; c0 < b && found < 0; c0 += 64) {
        const int64_t c = c0 + lane;
        const unsigned long long hit = __builtin_amdgcn_ballot_w64(c < b && flags[(int64_t)side * chunks + c] != 0);
        if (hit) found = c0 + __builtin_ctzll(hit);
    }
    if (lane == 0) starts[side * K + k] = found;
}

// the segments' states: each runs from its start to the next segment's start (one workgroup per side, thread k = segment k)
__global__ __launch_bounds__(1024) void k_cc_ranges(const int64_t *__restrict__ starts, int K, int64_t chunks, CcState *st,
                                                    int64_t list_cap, int64_t seg_base, int64_t seg_cap, int64_t *n_active)
{
    const int side = (int)blockIdx.x, k = (int)threadIdx.x;
    if (k >= K) return;
    int64_t next = chunks;                                              // the first started segment after this one
    for (int j = k + 1; j < K; ++j) {
        const int64_t a = starts[side * K + j];
        if (a >= 0) { next = a; break; }
    }
    if (k == 0) { st[side].limit = next; return; }                      // the side's first segment: from where the stream starts
    CcState *q = st + 2 + side * (K - 1) + (k - 1);
    const int64_t a = starts[side * K + k];
    q->side = side + 1;
    q->list_off = seg_base + ((int64_t)side * (K - 1) + (k - 1)) * seg_cap; q->list_cap = seg_cap;
    q->sp = 0.0; q->sn = 0.0; q->mag_p = 0.0; q->mag_n = 0.0; q->n_out = 0; q->visits = 0; q->status = CC_ST_DONE;
    if (a >= 0) { q->chunk = a; q->limit = next; q->reset_p = a * CC_CHUNK; q->reset_n = a * CC_CHUNK; }
    else { q->chunk = 0; q->limit = 0; q->reset_p = 0; q->reset_n = 0; }
    const unsigned long long on = __builtin_amdgcn_ballot_w64(a >= 0);
    if (fmk_lane() == (int)__builtin_ctzll(on | (1ULL << 63)) && on) atomicAdd((unsigned long long *)n_active, (unsigned long long)__builtin_popcountll(on));
}

// the segments' closes behind the first segment's, in order: the side's list
__global__ __launch_bounds__(256) void k_cc_gather(const CcState *__restrict__ st, int K, int64_t *__restrict__ lists, int64_t list_cap)
{
    const int side = (int)blockIdx.x / (K - 1), k = 1 + (int)blockIdx.x % (K - 1);
    const CcState *q = st + 2 + side * (K - 1) + (k - 1);
    const int64_t cnt = q->n_out;
    if (cnt == 0) return;
    __shared__ int64_t s_part[4];
    int64_t part = 0;                                                    // closes of the side's earlier segments
    for (int j = 1 + (int)threadIdx.x; j < k; j += 256) part += st[2 + side * (K - 1) + (j - 1)].n_out;
    part = fmk_wave_sum(part);
    if (fmk_lane() == 0) s_part[threadIdx.x >> 6] = part;
    __syncthreads();
    const int64_t off = st[side].n_out + s_part[0] + s_part[1] + s_part[2] + s_part[3];
    const int64_t *src = lists + q->list_off;
    int64_t *dst = lists + (int64_t)side * list_cap + off;
    for (int64_t i = threadIdx.x; i < cnt; i += 256)
        if (off + i < list_cap) dst[i] = src[i];
}

// the two sides' closes (each ascending) -> out[1 ..]; *coincide = 1 when a tick is in both
__global__ __launch_bounds__(256) void k_cc_merge(const int64_t *__restrict__ lp, int64_t np, const int64_t *__restrict__ ln,
                                                  int64_t nn, int64_t *__restrict__ out, int64_t capacity, int64_t *coincide)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= np + nn) return;
    const bool is_p = i < np;
    const int64_t self = is_p ? i : i - np, v = is_p ? lp[self] : ln[self];
    const int64_t *o = is_p ? ln : lp;
    int64_t lo = 0, hi = is_p ? nn : np;                              // entries of the other list before v (ties: positive first)
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const bool before = is_p ? o[mid] < v : o[mid] <= v;
        if (before) lo = mid + 1; else hi = mid;
    }
    if (is_p && lo < nn && ln[lo] == v) *coincide = 1;
    if (out && 1 + self + lo < capacity) out[1 + self + lo] = v;
}

static int64_t g_cc_last[4];            // tier used by the last call (0 fixed point, 1 this one), chunks opened, walk status, segments
extern "C" int fmk_diag_cusum_last(int64_t *tier, int64_t *opened, int64_t *status)
{
    if (tier) *tier = g_cc_last[0];
    if (opened) *opened = g_cc_last[1];
    if (status) *status = g_cc_last[2];
    return FMK_OK;
}

static double g_cc_rate;                // sub-blocks per chunk and side of the leading chunks in which a close is certain
extern "C" int fmk_diag_cusum_segments(int64_t *segments, double *rate)
{
    if (segments) *segments = g_cc_last[3];
    if (rate) *rate = g_cc_rate;
    return FMK_OK;
}

// The tier.  Returns FMK_OK with *done = 1 and the closes in d_out[1 .. *total] (d_out may be null: count only), or with
// *done = 0 when the caller has to run the fixed point (thresholds reached often, an uncertain decision, a bad return).
int fmk_cusum_chain_tier(fmk_ctx *ctx, const int64_t *d_ts, const double *d_price, const double *d_sigma, int64_t n,
                         int64_t first, int64_t m, int64_t chunks, double sigma_floor, double sigma_mult, int64_t *d_out,
                         int64_t capacity, int64_t *total, int64_t *visits, int *done, int *nan_seen)
{
    // nan_seen != null: sigma has not been forward filled (fmk_cusum.hip) -- its first valid index is `first`, and a NaN after
    // it (k_cc_summary reports one) ends the tier with *nan_seen = 1
    *done = 0;
    if (nan_seen) *nan_seen = 0;
    // developer knobs, read per call: FMK_CUSUM_CHAIN = 0 (never) / 1 (default) / 2 (no budget of opened chunks), optionally followed by
    // ":joint=1" (the one-workgroup joint walk only), ":segments=K" (segments of the walk after the sample, k_cc_sync; 1 = one walk per
    // side), ":sample=N" (walk N leading chunks in a launch of their own first) -- the shapes tests/test_gpu_cusum.py forces
    const char *v = getenv("FMK_CUSUM_CHAIN");
    const int mode = v ? atoi(v) : 1;
    auto chain_opt = [&](const char *key, long long dflt) -> long long {
        const char *e = getenv("FMK_CUSUM_CHAIN");
        const char *q = e ? strstr(e, key) : nullptr;
        return q ? atoll(q + strlen(key)) : dflt;
    };
    const bool force_joint = chain_opt(":joint=", 0) != 0;
    const long long segments_opt = chain_opt(":segments=", 512), sample_opt = chain_opt(":sample=", 0);
    v = getenv("FMK_CUSUM_CHAIN_MIN_CHUNKS");
    const int64_t min_chunks = v ? atoll(v) : 4096;
    v = getenv("FMK_CUSUM_MARGIN_SCALE");
    double margin_scale = v ? atof(v) : 1.0;
    if (!(margin_scale > 0.0)) margin_scale = 1.0;
    g_cc_last[0] = 0; g_cc_last[1] = 0; g_cc_last[2] = -1; g_cc_last[3] = 0;
    if (mode == 0 || chunks < min_chunks || chunks < 2) return FMK_OK;
    void *scr;
    const size_t sum_bytes = (size_t)chunks * 8 * sizeof(double);
    // per-side close lists: a side that fills its list ends the tier like an exhausted budget
    const int64_t list_cap = m < ((int64_t)1 << 22) ? m : ((int64_t)1 << 22);
    const size_t list_bytes = (size_t)list_cap * 8;
    const size_t sub_bytes = sum_bytes * (CC_CHUNK / CC_SUB);
    // segments of the walk after the sample (k_cc_sync); 1 = one walk per side
    int K = (int)segments_opt;
    if (K > 1024) K = 1024;
    if (K < 1 || force_joint) K = 1;
    const int n_states = 2 * K + 1, JOINT = 2 * K;                    // [pos, neg, later segments of pos, of neg, joint]
    const size_t flag_bytes = ((size_t)2 * chunks + 15) & ~(size_t)15;
    const size_t state_bytes = (sizeof(CcState) * n_states + 8 * (2 * K + 2) + 15) & ~(size_t)15;
    FMK_TRY(fmk_scratch(ctx, sum_bytes + sub_bytes + 4 * list_bytes + flag_bytes + state_bytes + 512, &scr));
    double *sums = (double *)scr;
    double *subs = (double *)((char *)scr + sum_bytes);
    int64_t *lists = (int64_t *)((char *)scr + sum_bytes + sub_bytes);           // [2][list_cap] the sides, [2][list_cap] segments
    unsigned char *flags = (unsigned char *)scr + sum_bytes + sub_bytes + 4 * list_bytes;
    CcState *st = (CcState *)(flags + flag_bytes);
    int64_t *d_coincide = (int64_t *)(st + n_states), *d_active = d_coincide + 1, *d_starts = d_coincide + 2;
    struct CcHost { CcState st[3]; int64_t coincide; };              // host copy: [0] positive, [1] negative, [2] joint
    k_cc_init<<<(unsigned)fmk_ceil_div(n_states, 64), 64, 0, ctx->stream>>>(st, n_states, list_cap, d_coincide);
    FMK_LAUNCH_CHECK(ctx);
    FMK_HIP(ctx, hipMemsetAsync(d_active, 0, 8, ctx->stream));
    // Which tier?  Measured at 1e9 ticks (profiles/r02_cusum_chain.txt): the segmented walk costs ~8.5 ms + 8 ns per opened
    // sub-block or event of the busier side, the fixed point 24 / 24 / 26 / 47 / 66 ms at 4.8 M / 2.4 M / 1.1 M / 398 K / 101 K
    // closes -- the walk wins up to ~3 visits per chunk and side.  The rate is estimated from the sub-block summaries of the
    // leading 2048 chunks (k_cc_rate: no walk, one small launch); the visit budgets below are the safety net for tapes whose
    // beginning is not representative.
    const double max_rate = 3.0, max_certain = 1.3;
    const int64_t sample_want = sample_opt > 0 ? sample_opt : 0;       // (tests) walk this many leading chunks in a launch of their own first
    const int64_t sample = chunks < sample_want ? chunks : sample_want;
    const int64_t lead = chunks < 2048 ? chunks : 2048;
    const int bad_mask = nan_seen ? 3 : 1;
    CcHost hh;
    memset(&hh, 0, sizeof(hh));
    auto fetch = [&]() -> int {
        FMK_HIP(ctx, hipMemcpyAsync(&hh.st[0], st, 2 * sizeof(CcState), hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipMemcpyAsync(&hh.st[2], st + JOINT, sizeof(CcState), hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return FMK_OK;
    };
    auto walk = [&](int joint, int64_t hi, int64_t budget) -> int {
        k_cc_walk<<<joint ? 1 : 2, 64 * CC_WALK_WAVES, 0, ctx->stream>>>(d_ts, d_price, d_sigma, n, first, m, hi, chunks, sigma_floor,
                                                                        sigma_mult, sums, subs, joint ? st + JOINT : st, st, bad_mask, budget,
                                                                        margin_scale, lists, d_out, d_out ? capacity : 0);
        FMK_LAUNCH_CHECK(ctx);
        return fetch();
    };
    // the rest of the stream after the sample, each side in up to K segments that start where k_cc_sync proved independence
    std::vector<CcState> hseg;
    int64_t active = 0;
    auto walk_segments = [&](int64_t lo, int64_t budget) -> int {
        const unsigned nb = (unsigned)(2 * (K - 1));
        k_cc_sync<<<(unsigned)fmk_ceil_div(chunks - lo - 1, 256), 256, 0, ctx->stream>>>(sums, chunks, lo, margin_scale, flags);
        FMK_LAUNCH_CHECK(ctx);
        k_cc_pick<<<nb, 64, 0, ctx->stream>>>(flags, chunks, lo, K, d_starts);
        FMK_LAUNCH_CHECK(ctx);
        k_cc_ranges<<<2, 1024, 0, ctx->stream>>>(d_starts, K, chunks, st, list_cap, 2 * list_cap, (2 * list_cap) / (2 * (K - 1)), d_active);
        FMK_LAUNCH_CHECK(ctx);
        k_cc_walk<<<(unsigned)(2 * K), 64 * CC_WALK_WAVES, 0, ctx->stream>>>(d_ts, d_price, d_sigma, n, first, m, chunks, chunks,
                                                                            sigma_floor, sigma_mult, sums, subs, st, st, bad_mask, budget,
                                                                            margin_scale, lists, d_out, d_out ? capacity : 0);
        FMK_LAUNCH_CHECK(ctx);
        hseg.resize((size_t)(2 * K));
        FMK_HIP(ctx, hipMemcpyAsync(hseg.data(), st, sizeof(CcState) * 2 * K, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipMemcpyAsync(&active, d_active, 8, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        // fold the segments into the two sides' records: counts add up, the worst status and the largest visit count stand
        for (int side = 0; side < 2; ++side) {
            CcState &h = hh.st[side];
            h = hseg[(size_t)side];
            int64_t tot = h.n_out, vis = h.visits;
            for (int k = 1; k < K; ++k) {
                const CcState &q = hseg[(size_t)(2 + side * (K - 1) + (k - 1))];
                tot += q.n_out; vis += q.visits;
                if (h.status == CC_ST_DONE && q.status != CC_ST_DONE) h.status = q.status;
            }
            if (tot > list_cap && h.status == CC_ST_DONE) h.status = CC_ST_BUDGET;
            if (h.status == CC_ST_DONE && tot > h.n_out) {
                // (one launch for both sides below)
            }
            h.n_out = tot; h.visits = vis;
        }
        if (hh.st[0].status == CC_ST_DONE && hh.st[1].status == CC_ST_DONE) {
            k_cc_gather<<<nb, 256, 0, ctx->stream>>>(st, K, lists, list_cap);
            FMK_LAUNCH_CHECK(ctx);
        }
        return FMK_OK;
    };
    auto summarize = [&](int64_t lo, int64_t hi) -> int {
        k_cc_summary<<<(unsigned)fmk_ceil_div(hi - lo, 4), 256, 0, ctx->stream>>>(d_ts, d_price, d_sigma, n, first, m, lo, hi,
                                                                                chunks, sigma_floor, sigma_mult, sums, subs, st);
        FMK_LAUNCH_CHECK(ctx);
        return FMK_OK;
    };
    const int j0 = force_joint ? 1 : 0;
    auto worst = [&]() { return j0 ? hh.st[2].status : (hh.st[0].status ? hh.st[0].status : hh.st[1].status); };
    auto spent = [&]() { return j0 ? hh.st[2].visits : (hh.st[0].visits > hh.st[1].visits ? hh.st[0].visits : hh.st[1].visits); };
    FMK_TRY(summarize(0, lead));
    g_cc_rate = -1.0;
    if (mode != 2) {
        unsigned long long cnt = 0;
        FMK_HIP(ctx, hipMemsetAsync(d_starts, 0, 8, ctx->stream));
        k_cc_rate<<<(unsigned)fmk_ceil_div(4 * lead, 256), 256, 0, ctx->stream>>>(subs, 4 * chunks, 4 * lead, (unsigned long long *)d_starts);
        FMK_LAUNCH_CHECK(ctx);
        FMK_HIP(ctx, hipMemcpyAsync(&cnt, d_starts, 8, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        g_cc_rate = (double)cnt / (2.0 * (double)lead);
        if (g_cc_rate > max_certain) { g_cc_last[2] = CC_ST_BUDGET; return FMK_OK; }     // thresholds reached often: the fixed point
    }
    if (lead < chunks) FMK_TRY(summarize(lead, chunks));
    const bool segmented = !j0 && K > 1 && chunks - sample >= 4 * (int64_t)K;
    // budgets of opened sub-blocks + events + replayed ticks / 1024: four times the rate up to which the walk is chosen
    int64_t budget = INT64_MAX, seg_budget = INT64_MAX;
    if (mode != 2) {
        budget = (int64_t)(4.0 * max_rate * (double)chunks) + 1000;
        seg_budget = (int64_t)(4.0 * max_rate * (double)(chunks / K + 1)) + 256;
    }
    hh.st[0].status = hh.st[1].status = hh.st[2].status = CC_ST_DONE;
    if (sample > 0) FMK_TRY(walk(j0, sample, budget));
    if (worst() == CC_ST_DONE && sample < chunks) {
        if (segmented) FMK_TRY(walk_segments(sample, seg_budget));
        else FMK_TRY(walk(j0, chunks, budget));
    }
    CcState h = hh.st[j0 ? 2 : 0];
    h.status = worst(); h.visits = spent();
    if (!j0 && h.status == CC_ST_DONE) {
        const int64_t np = hh.st[0].n_out, nn = hh.st[1].n_out;
        h.n_out = np + nn;
        if (np + nn > 0) {
            k_cc_merge<<<(unsigned)fmk_ceil_div(np + nn, 256), 256, 0, ctx->stream>>>(lists, np, lists + list_cap, nn, d_out,
                                                                                    d_out ? capacity : 0, d_coincide);
            FMK_LAUNCH_CHECK(ctx);
            FMK_HIP(ctx, hipMemcpyAsync(&hh.coincide, d_coincide, 8, hipMemcpyDeviceToHost, ctx->stream));
            FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (hh.coincide) {                                        // a positive and a negative close on one tick: `elif`
                FMK_TRY(walk(1, chunks, budget));
                h = hh.st[2];
            }
        }
    }
    if (nan_seen && (hh.st[0].bad & 2)) *nan_seen = 1;
    if (visits) *visits = h.visits;
    g_cc_last[1] = h.visits; g_cc_last[2] = h.status; g_cc_last[3] = active;
    if (h.status != CC_ST_DONE) return FMK_OK;
    *total = h.n_out;
    *done = 1;
    g_cc_last[0] = 1;
    return FMK_OK;
}
