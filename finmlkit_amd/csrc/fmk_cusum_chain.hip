// fmk_cusum_chain.hip -- _cusum_bar_indexer (finmlkit/bar/logic.py:152-221) when the thresholds are RARELY reached.
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
//   k_cc_summary : one wave per 2048-tick chunk reduces its ticks to {B, A+, P+, Q+, A-, P-, Q-, U = max |S_i|}
//                  (coalesced loads, 8 consecutive ticks per lane through LDS, an ordered tree over the lanes): a stream pass;
//   k_cc_walk    : one wave follows a chain: 64 chunk summaries per step (a scan of the exit maps gives every chunk its
//                  incoming state, a ballot the first chunk that may close), and only such a chunk is opened: its ticks are
//                  recomputed from the columns (by the four waves of the workgroup), the walking wave takes 32 of them per
//                  lane, runs the reference's own operations over them from the state the lane scan hands it, the first
//                  event is emitted and the side that closed restarts from 0 at the next tick.
//                  ONLY the side that closed resets (logic.py:214-219), so each side is a chain of its own: workgroup 0
//                  walks the positive side, workgroup 1 the negative side, k_cc_merge interleaves the two lists.  The sides
//                  couple in one place -- a positive close hides a negative one on the same tick (`if / elif`) -- so if the
//                  lists share a tick the joint walk (both sides in one wave, `if / elif` as written) answers instead.
// Arithmetic.  The block sums are not the reference's sequential float64 sum from the last reset, so every decision
// carries a margin: (ticks since that side's reset + 4096) * 2^-50 * (largest magnitude the side's state or a block
// prefix has reached since) -- 4x the worst-case distance between two float64 evaluation orders of the same recurrence plus
// a 1-ulp difference in log().  A chunk is skipped only when it stays below the threshold by more than the margin, a close is
// accepted only when it exceeds it by more than the margin; anything in between ends the tier (status UNCERTAIN) and the
// caller runs the fixed point of fmk_cusum.hip, which is the reference's loop operation for operation.  Non-finite returns
// (a price <= 0) are outside the algebra: status BAD, same fallback.  Expected uncertain decisions at 1e9 ticks: ~1e-2.
// Cost: the summary pass (24 B/tick, 7 ms per 1e9 ticks) + ~6 us per close (profiles/r02_cusum_chain.txt); the caller tries the first 2^22 ticks
// with a small budget of opened chunks first, so a tape whose thresholds are reached often never pays for this tier.
#include <math.h>
#include <stdlib.h>

#include "fmk_common.h"
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
    *r = log(p / pm);
    double l = NAN;
    if (!(has_next && tsi == tsn)) {
        l = sigma_mult * sg;
        l = sigma_floor > l ? sigma_floor : l;
    }
    *lam = l;
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
    bool bad = false;
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
        // ordered tree over the lanes: after step d, lanes that are multiples of 2d hold [lane, lane + 2d)
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            CcSum o;
            o.B = cc_shfl_down(me.B, d); o.U = cc_shfl_down(me.U, d);
            o.Ap = cc_shfl_down(me.Ap, d); o.Pp = cc_shfl_down(me.Pp, d); o.Qp = cc_shfl_down(me.Qp, d);
            o.An = cc_shfl_down(me.An, d); o.Pn = cc_shfl_down(me.Pn, d); o.Qn = cc_shfl_down(me.Qn, d);
            me = cc_compose(me, o);
        }
        if (lane == 0) {                                      // the 512-tick sub-block's own summary (the walk opens sub-blocks)
            const int64_t q = k * (CC_CHUNK / CC_SUB) + sub, nq = chunks * (CC_CHUNK / CC_SUB);
            subs[0 * nq + q] = me.B; subs[1 * nq + q] = me.U;
            subs[2 * nq + q] = me.Ap; subs[3 * nq + q] = me.Pp; subs[4 * nq + q] = me.Qp;
            subs[5 * nq + q] = me.An; subs[6 * nq + q] = me.Pn; subs[7 * nq + q] = me.Qn;
        }
        acc = cc_compose(acc, me);                            // meaningful on lane 0
    }
    if (lane == 0) {
        sums[0 * chunks + k] = acc.B; sums[1 * chunks + k] = acc.U;
        sums[2 * chunks + k] = acc.Ap; sums[3 * chunks + k] = acc.Pp; sums[4 * chunks + k] = acc.Qp;
        sums[5 * chunks + k] = acc.An; sums[6 * chunks + k] = acc.Pn; sums[7 * chunks + k] = acc.Qn;
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0 && !__atomic_load_n(&state->bad, __ATOMIC_RELAXED))
        atomicOr(&state->bad, 1);
}

__global__ void k_cc_init(CcState *st3, int64_t *flag)
{
    if (threadIdx.x == 0) *flag = 0;
    CcState *st = st3 + threadIdx.x;                                  // positive side, negative side, joint
    st->chunk = 0; st->sp = 0.0; st->sn = 0.0; st->reset_p = 0; st->reset_n = 0; st->mag_p = 0.0; st->mag_n = 0.0;
    st->n_out = 0; st->visits = 0; st->status = CC_ST_DONE; st->bad = 0;
}

// ---------------------------------------------------------------------------------------
// the walk along the chain: wave 0 walks, the other waves of the workgroup only help to open a chunk
// ---------------------------------------------------------------------------------------
#define CC_PAD(j) ((j) + ((j) >> 5))
#define CC_WALK_WAVES 8
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

__global__ __launch_bounds__(64 * CC_WALK_WAVES) void k_cc_walk(const int64_t *__restrict__ ts, const double *__restrict__ price,
                                                const double *__restrict__ sigma, int64_t n, int64_t first, int64_t m,
                                                int64_t chunk_limit, int64_t chunks, double sigma_floor, double sigma_mult,
                                                const double *__restrict__ sums, const double *__restrict__ subs,
                                                CcState *states, int64_t visit_budget,
                                                double margin_scale, int joint, int64_t *__restrict__ lists, int64_t list_cap,
                                                int64_t *__restrict__ closes, int64_t capacity)
{
    // joint != 0: one workgroup evaluates the loop as written (both sides, `if / elif`), closes -> closes[1 + ...].
    // joint == 0: workgroup 0 follows the positive side alone, workgroup 1 the negative side alone (each side's state after
    // its own close is 0 whatever the other side does, so the two chains are independent -- except that a positive close
    // hides a negative one on the same tick; the caller merges the two lists and re-runs jointly if they ever share a tick).
    const int side = joint ? 0 : (int)blockIdx.x + 1;
    const bool do_p = side != 2, do_n = side != 1;
    CcState *state = states + (joint ? 2 : (int)blockIdx.x);
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
    int status = states[0].bad ? CC_ST_BAD : CC_ST_DONE;              // k_cc_summary reports there
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
                const int kind = __builtin_amdgcn_readlane(kind_l, fl);
                if (kind == 3) { status = CC_ST_UNCERTAIN; break; }
                const int j = 8 * fl + q0;
                if (joint) { if (lane == 0 && closes && 1 + n_out < capacity) closes[1 + n_out] = first + 1 + ts0 + j; }
                else {
                    if (n_out >= list_cap) { status = CC_ST_BUDGET; break; }
                    if (lane == 0) lists[(int64_t)(side - 1) * list_cap + n_out] = first + 1 + ts0 + j;
                }
                ++n_out; ++visits;                                        // an event costs about as much as opening a sub-block
                sp = kind == 1 ? 0.0 : cc_bcast(bp, fl);                  // states after that tick, the closing side reset
                sn = kind == 2 ? 0.0 : cc_bcast(bn, fl);
                if (kind == 1) { reset_p = ts0 + j; mag_p = 2.0 * U_s; } else { reset_n = ts0 + j; mag_n = 2.0 * U_s; }
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

static int64_t g_cc_last[3];            // tier used by the last call (0 fixed point, 1 this one), chunks opened, walk status
extern "C" int fmk_diag_cusum_last(int64_t *tier, int64_t *opened, int64_t *status)
{
    if (tier) *tier = g_cc_last[0];
    if (opened) *opened = g_cc_last[1];
    if (status) *status = g_cc_last[2];
    return FMK_OK;
}

// The tier.  Returns FMK_OK with *done = 1 and the closes in d_out[1 .. *total] (d_out may be null: count only), or with
// *done = 0 when the caller has to run the fixed point (thresholds reached often, an uncertain decision, a bad return).
int fmk_cusum_chain_tier(fmk_ctx *ctx, const int64_t *d_ts, const double *d_price, const double *d_sigma, int64_t n,
                         int64_t first, int64_t m, int64_t chunks, double sigma_floor, double sigma_mult, int64_t *d_out,
                         int64_t capacity, int64_t *total, int64_t *visits, int *done)
{
    *done = 0;
    // developer knobs, read per call: FMK_CUSUM_CHAIN = 0 (never) / 1 (default) / 2 (no budget of opened chunks)
    const char *v = getenv("FMK_CUSUM_CHAIN");
    const int mode = v ? atoi(v) : 1;
    v = getenv("FMK_CUSUM_CHAIN_MIN_CHUNKS");
    const int64_t min_chunks = v ? atoll(v) : 4096;
    v = getenv("FMK_CUSUM_MARGIN_SCALE");
    double margin_scale = v ? atof(v) : 1.0;
    if (!(margin_scale > 0.0)) margin_scale = 1.0;
    g_cc_last[0] = 0; g_cc_last[1] = 0; g_cc_last[2] = -1;
    if (mode == 0 || chunks < min_chunks || chunks < 2) return FMK_OK;
    v = getenv("FMK_CUSUM_CHAIN_JOINT");
    const bool force_joint = v && atoi(v);                           // developer knob: the one-workgroup joint walk only
    void *scr;
    const size_t sum_bytes = (size_t)chunks * 8 * sizeof(double);
    // per-side close lists: a side that fills its list ends the tier like an exhausted budget
    const int64_t list_cap = m < ((int64_t)1 << 22) ? m : ((int64_t)1 << 22);
    const size_t list_bytes = (size_t)list_cap * 8;
    const size_t sub_bytes = sum_bytes * (CC_CHUNK / CC_SUB);
    FMK_TRY(fmk_scratch(ctx, sum_bytes + sub_bytes + 2 * list_bytes + 512, &scr));
    double *sums = (double *)scr;
    double *subs = (double *)((char *)scr + sum_bytes);
    int64_t *lists = (int64_t *)((char *)scr + sum_bytes + sub_bytes);
    struct CcHost { CcState st[3]; int64_t coincide; };
    CcHost *dev = (CcHost *)((char *)scr + sum_bytes + sub_bytes + 2 * list_bytes);
    CcState *st = dev->st;
    k_cc_init<<<1, 3, 0, ctx->stream>>>(st, &dev->coincide);
    FMK_LAUNCH_CHECK(ctx);
    // a sample first: the leading 2048 chunks with a budget of opened chunks that the slow regime never needs
    v = getenv("FMK_CUSUM_CHAIN_SAMPLE");                              // developer knob: chunks of the sample phase
    const int64_t sample_want = v && atoll(v) > 0 ? atoll(v) : 2048;
    const int64_t sample = chunks < sample_want ? chunks : sample_want;
    // Measured at 1e9 ticks (profiles/r02_cusum_chain.txt): the walk costs ~9 ms + 6 us per close, the fixed point 73 / 109 /
    // 164 / 320 ms at 101 K / 25 K / 11 K / 4 K closes -- the walk wins below ~0.04 closes per chunk.
    const double max_rate = 0.04;
    const int64_t sample_budget = mode == 2 ? INT64_MAX : 32 + (int64_t)(max_rate * (double)sample);
    CcHost hh;
    auto walk = [&](int joint, int64_t hi, int64_t budget) -> int {
        k_cc_walk<<<joint ? 1 : 2, 64 * CC_WALK_WAVES, 0, ctx->stream>>>(d_ts, d_price, d_sigma, n, first, m, hi, chunks, sigma_floor,
                                                                        sigma_mult, sums, subs, st, budget, margin_scale, joint, lists,
                                                                        list_cap, d_out, d_out ? capacity : 0);
        FMK_LAUNCH_CHECK(ctx);
        FMK_HIP(ctx, hipMemcpyAsync(&hh, dev, sizeof(CcHost), hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
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
    FMK_TRY(summarize(0, sample));
    FMK_TRY(walk(j0, sample, sample_budget));
    int64_t budget = sample_budget;
    if (worst() == CC_ST_DONE && sample < chunks) {
        // the rest with three times the sample's rate as a safety net (an exhausted budget hands over to the fixed point)
        budget = INT64_MAX;
        if (mode != 2) {
            const double rate = (double)(spent() + 1) / (double)sample;
            budget = (int64_t)(3.0 * rate * (double)chunks) + 1000;
            const int64_t cap = (int64_t)(2.0 * max_rate * (double)chunks) + 1000;
            if (budget > cap) budget = cap;
        }
        FMK_TRY(summarize(sample, chunks));
        FMK_TRY(walk(j0, chunks, budget));
    }
    CcState h = hh.st[j0 ? 2 : 0];
    h.status = worst(); h.visits = spent();
    if (!j0 && h.status == CC_ST_DONE) {
        const int64_t np = hh.st[0].n_out, nn = hh.st[1].n_out;
        h.n_out = np + nn;
        if (np + nn > 0) {
            k_cc_merge<<<(unsigned)fmk_ceil_div(np + nn, 256), 256, 0, ctx->stream>>>(lists, np, lists + list_cap, nn, d_out,
                                                                                    d_out ? capacity : 0, &dev->coincide);
            FMK_LAUNCH_CHECK(ctx);
            FMK_HIP(ctx, hipMemcpyAsync(&hh.coincide, &dev->coincide, 8, hipMemcpyDeviceToHost, ctx->stream));
            FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (hh.coincide) {                                        // a positive and a negative close on one tick: `elif`
                FMK_TRY(walk(1, chunks, budget));
                h = hh.st[2];
            }
        }
    }
    if (visits) *visits = h.visits;
    g_cc_last[1] = h.visits; g_cc_last[2] = h.status;
    if (h.status != CC_ST_DONE) return FMK_OK;
    *total = h.n_out;
    *done = 1;
    g_cc_last[0] = 1;
    return FMK_OK;
}
