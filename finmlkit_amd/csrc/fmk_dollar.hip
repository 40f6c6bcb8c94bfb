// fmk_dollar.hip -- _dollar_bar_indexer (finmlkit/bar/logic.py:118-149), parallel closed form.
//
// Reference recurrence (sequential):   cum += p_i*v_i ; if cum >= thr: close at i, cum -= thr.
// With D_i = sum_{k<=i} d_k (d_k = fl(p_k*v_k), the same rounded product the reference adds),
// M_i = floor(D_i / thr) and K_i = number of closes among ticks 1..i, the recurrence is
//     K_0 = 0,  K_i = min(K_{i-1} + 1, M_i)        (at most one close per tick, tick 0 never closes)
// whose solution is  K_i = i + min_{0<=j<=i} G_j  with G_0 = 0, G_j = M_j - j :  a prefix-MIN scan.
// Tick i closes iff G_i >= min_{j<i} G_j, and K_i is directly its slot in the output array -- no
// stream compaction pass.  Everything is a scan:
//   k_dl_tile_sums   tile sums of d in double-double (two-sum, ~2^-104 relative: "exact")
//   k_dl_scan_dd     exclusive scan of the tile sums (one block)
//   k_dl_tile_min    per tile: D_i -> M_i -> G_i, tile minimum
//   k_dl_scan_min    exclusive prefix-min of the tile minima (one block)
//   k_dl_emit        per tile: recompute G_i, running min, write out[K_i] = i
// Traffic: price 8 + amount 4 B/tick, three times (sum, min, emit) + 8 B per close; all three passes load coalesced (the
// two that scan in tick order hand the products to their owners through a padded LDS tile).
//
// Two-pass form (the common case, chosen by dl_run from what pass 1 learned): k_dl_tile_sums also takes the largest
// increment.  When it is below thr, M grows by at most 1 per tick, so G_j = M_j - j never increases, the running minimum
// is simply the previous tick's G and K_i = M_i: k_dl_tile_min / k_dl_scan_min are skipped, the close count comes from the
// double-double total on the host, and k_dl_emit ("simple") does ONE double-double division per thread and follows the
// remainder incrementally (a thread whose own increments reach thr falls back to a division per tick).  24 B/tick.
// The float64-drift replay that makes the result exact (default mode) lives in fmk_dollar_exact.hip.
//
// Exact arithmetic vs the reference's float64 running sum: the reference's `cum` carries its own
// rounding drift (<= (i+1)*2^-52*thr after i adds, the carry never resets it -- plus 2^-53 * W^2 / thr where increments of W in
// all reach the threshold by themselves: the backlog they leave is worked off at the magnitude of cum, see dl_run).  A decision is
// reported in n_uncertified when the exact cum lies within that bound (or 1e-11*thr) of the threshold;
// 0 means the close indices are provably the reference's.
#include <math.h>
#include <stdlib.h>

#include "fmk_common.h"
#include <type_traits>

struct DD { double hi, lo; };

__device__ __forceinline__ DD dd_make(double a) { return DD{a, 0.0}; }
__device__ __forceinline__ DD dd_fast2(double s, double e) { double h = s + e; return DD{h, e - (h - s)}; }
__device__ __forceinline__ DD dd_add(DD x, double y)
{
    double s = x.hi + y, bb = s - x.hi;
    double e = (x.hi - (s - bb)) + (y - bb);
    e += x.lo;
    return dd_fast2(s, e);
}
__device__ __forceinline__ DD dd_add(DD x, DD y)
{
    double s = x.hi + y.hi, bb = s - x.hi;
    double e = (x.hi - (s - bb)) + (y.hi - bb);
    double t = x.lo + y.lo, tb = t - x.lo;
    double f = (x.lo - (t - tb)) + (y.lo - tb);
    e += t;
    DD r = dd_fast2(s, e);
    r.lo += f;
    return dd_fast2(r.hi, r.lo);
}
__device__ __forceinline__ DD dd_shfl_up(DD v, int d) { return DD{__shfl_up(v.hi, d, 64), __shfl_up(v.lo, d, 64)}; }

// floor(D / thr) for D >= 0 in double-double; *frag receives the distance of D/thr to the nearest
// integer boundary in units of thr (for the certification count)
__device__ __forceinline__ int64_t dd_floor_div(DD D, double thr, double *frac_dist, double *rem = nullptr)
{
    double q = floor(D.hi / thr);
    double p = q * thr, e = fma(q, thr, -p);          // q*thr = p + e exactly
    double r = (D.hi - p) + (D.lo - e);
    while (r < 0.0) { q -= 1.0; r += thr; }
    while (r >= thr) { q += 1.0; r -= thr; }
    *frac_dist = fmin(r, thr - r) / thr;
    if (rem) *rem = r;                                // D - q*thr to ~1 ulp: the exact-arithmetic carry after a close
    return (int64_t)q;
}

#define DL_THREADS 256
#define DL_ITEMS 8
#define DL_TILE (DL_THREADS * DL_ITEMS)
#define DL_SEG_SHIFT 11                 // tiles per segment of the double-double scan: one round of k_dl_scan_dd
#define DL_SEGM_SHIFT 13                // ... of the prefix-min scan: one round of k_dl_scan_min
#define DL_EXTRA 16                     // spare output slots behind the closes of the closed form (fmk_dollar_exact.hip)

template <bool AF64>
__device__ __forceinline__ double dl_d(const double *price, const void *amount, int64_t i)
{
    return price[i] * fmk_amt<AF64>(amount, i);       // rounded once, like prices[i] * volumes[i]
}

// block-wide inclusive scan of DD thread totals; returns this thread's EXCLUSIVE prefix and the block total
__device__ __forceinline__ DD dl_block_exclusive(DD mine, DD *lds /*[4]*/, DD *total)
{
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
    DD inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        DD o = dd_shfl_up(inc, d);
        if (lane >= d) inc = dd_add(o, inc);
    }
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    DD pre = dd_make(0.0);
    for (int k = 0; k < w; ++k) pre = dd_add(pre, lds[k]);
    DD tot = lds[0];
    for (int k = 1; k < 4; ++k) tot = dd_add(tot, lds[k]);
    *total = tot;
    DD prev = dd_shfl_up(inc, 1);
    if (lane == 0) prev = dd_make(0.0);
    __syncthreads();
    return dd_add(pre, prev);
}

template <bool AF64>
__global__ __launch_bounds__(DL_THREADS) void k_dl_tile_sums(const double *__restrict__ price,
                                                             const void *__restrict__ amount, int64_t n,
                                                             DD *__restrict__ tile_sum, int *__restrict__ bad,
                                                             unsigned long long *__restrict__ dmax_bits, double thr,
                                                             double *__restrict__ whale_sum)
{
    __shared__ DD lds[4];
    // the tile's sum does not depend on the order of its terms (double-double: ~2^-104 relative), so the loads are COALESCED
    // -- thread t takes ticks t, t + 256, ... of the tile; owning 8 consecutive ticks, as the scans below must, makes every
    // load instruction touch 64 lines (2.7 TB/s instead of 5)
    const int64_t i0 = (int64_t)blockIdx.x * DL_TILE + (int64_t)threadIdx.x;
    DD s = dd_make(0.0);
    bool neg = false;
    double wave_whale = 0.0;
    double d[DL_ITEMS];
    {
        // the sixteen loads of the thread in flight together: with the product inside the guard the compiler waited for each pair
        // (tools/isa_loadwaits.py); the raw words are pinned behind the last load, then multiplied (rounded once: dl_d)
        typedef typename std::conditional<AF64, double, float>::type DlAmt;
        double lp[DL_ITEMS];
        DlAmt la[DL_ITEMS];
#pragma unroll
        for (int k = 0; k < DL_ITEMS; ++k) {
            const int64_t i = i0 + k * DL_THREADS;
            lp[k] = 0.0; la[k] = (DlAmt)0;
            if (i < n) { lp[k] = price[i]; la[k] = ((const DlAmt *)amount)[i]; }
        }
#pragma unroll
        for (int k = 0; k < DL_ITEMS; ++k) { asm volatile("" : "+v"(lp[k])); asm volatile("" : "+v"(la[k])); }
#pragma unroll
        for (int k = 0; k < DL_ITEMS; ++k) d[k] = lp[k] * (double)la[k];
    }
#pragma unroll
    for (int k = 0; k < DL_ITEMS; ++k) {
        neg |= !(d[k] >= 0.0);                       // negative or NaN increment: outside the closed form
        s = dd_add(s, d[k]);
    }
    if (__ballot(neg) != 0 && fmk_lane() == 0) atomicOr(bad, 1);
    {   // largest increment of the stream (the exact tier of fmk_dollar_exact.hip needs d_max < thr): non-negative doubles
        // order like their bit patterns; look first, a same-address atomic per wave is not free
        double m = d[0];
#pragma unroll
        for (int k = 1; k < DL_ITEMS; ++k) m = fmax(m, d[k]);
        m = fmk_wave_max(m);
        const unsigned long long mb = (unsigned long long)__double_as_longlong(m);
        if (fmk_lane() == 0 && !neg && mb > __hip_atomic_load(dmax_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(dmax_bits, mb);
        // ... and the sum of the increments that reach the threshold by themselves (whale trades): the reference's running sum then
        // stays ABOVE thr for a while -- one close per tick until the backlog is gone -- and its adds round at that magnitude, which
        // the drift bound of the decisions must know (dl_run)
        if (m >= thr) {                                             // (wave-uniform: m is the wave's maximum)
            double wsum = 0.0;
#pragma unroll
            for (int k = 0; k < DL_ITEMS; ++k) wsum += d[k] >= thr ? d[k] : 0.0;
            wsum = fmk_wave_sum(wsum);
            wave_whale = wsum;
        }
    }
    // the tile's share of that sum: a plain store per tile, added up by k_dl_whale_total.  (An atomicAdd per wave on ONE address cost
    // ~50 ns each, serialised: with 0.01 % block trades one wave in twenty has one -- k_dl_tile_sums 2.2 -> 7.4 ms per 1e9 ticks.)
    {
        __shared__ double s_wh[4];
        if (fmk_lane() == 0) s_wh[threadIdx.x >> 6] = neg ? 0.0 : wave_whale;
        __syncthreads();
        if (threadIdx.x == 0) whale_sum[blockIdx.x] = (s_wh[0] + s_wh[1]) + (s_wh[2] + s_wh[3]);
    }
    DD tot;
    (void)dl_block_exclusive(s, lds, &tot);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}

// W / thr (rounded up): the per-tile sums of the increments >= thr, added up by a few hundred workgroups (one block walking the
// 488 K tile records of 1e9 ticks took 0.21 ms -- on every dollar call, whales or not), one atomic each; *out is zeroed by the caller
__global__ __launch_bounds__(256) void k_dl_whale_total(const double *__restrict__ part, int64_t tiles, double thr, double *__restrict__ out)
{
    __shared__ double ws[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tiles; i += (int64_t)gridDim.x * 256) acc += part[i];
    acc = fmk_wave_sum(acc);
    if (fmk_lane() == 0) ws[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t = (ws[0] + ws[1]) + (ws[2] + ws[3]);
        if (t > 0.0) atomicAdd(out, t / thr * (1.0 + 1e-9));
    }
}

// exclusive scan in place (one block; every thread owns 8 consecutive records per round, so the block-wide scan and
// its barriers are paid once per 2048 records: 3.3 ms -> 0.4 ms for the 488 K tile sums of 1e9 ticks)
// Segmented: block s scans records [s * seg_len, (s + 1) * seg_len) on its own (exclusive WITHIN the segment) and leaves the
// segment's total in seg_tot[s]; a second, one-block call scans the ~240 segment totals, and the consumers add their
// segment's base.  As ONE block over all 488 K tile sums of 1e9 ticks this scan was 0.98 ms of the indexer's 11.
__global__ __launch_bounds__(DL_THREADS) void k_dl_scan_dd(DD *__restrict__ t_all, int64_t m_all, int64_t seg_len,
                                                           DD *__restrict__ seg_tot)
{
    __shared__ DD lds[4];
    __shared__ DD run_s;
    const int64_t lo = (int64_t)blockIdx.x * seg_len;
    DD *t = t_all + lo;
    const int64_t m = m_all - lo < seg_len ? m_all - lo : seg_len;
    if (threadIdx.x == 0) run_s = dd_make(0.0);
    __syncthreads();
    for (int64_t b = 0; b < m; b += (int64_t)DL_THREADS * 8) {
        const int64_t i0 = b + (int64_t)threadIdx.x * 8;
        DD loc[8];
        DD s = dd_make(0.0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            loc[k] = s;                                            // exclusive prefix inside my 8 records
            if (i0 + k < m) s = dd_add(s, t[i0 + k]);
        }
        DD tot;
        DD ex = dl_block_exclusive(s, lds, &tot);
        const DD base = dd_add(run_s, ex);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (i0 + k < m) t[i0 + k] = dd_add(base, loc[k]);
        __syncthreads();
        if (threadIdx.x == 0) run_s = dd_add(run_s, tot);
        __syncthreads();
    }
    if (seg_tot && threadIdx.x == 0) seg_tot[blockIdx.x] = run_s;
}

// G_i for the 8 ticks of this thread (G_0 = 0); returns the number of fragile ticks
template <bool AF64>
__device__ __forceinline__ int dl_thread_G(const double *price, const void *amount, int64_t n, double thr,
                                           const DD *tile_base, const DD *seg_base, DD *lds, int64_t (&G)[DL_ITEMS],
                                           double *rem = nullptr, double extra = 0.0 /* see dl_run: the backlog term of the drift */)
{
    // The thread owns 8 CONSECUTIVE ticks (the prefix runs in tick order) and loads them itself, 16 bytes per instruction (price
    // 4 x double2, amount 2 x float4 / 4 x double2).  History: eight 8-byte loads per array made every instruction touch 64
    // lines (2.8 TB/s); a coalesced load + padded LDS tile to hand the products to their owners fixed that at the price of 18 KB
    // of LDS and two barriers per workgroup; the 16-byte loads need neither (the lesson of profiles/r02_ewmst_direct_loads.txt).
    const int64_t t0 = (int64_t)blockIdx.x * DL_TILE;
    double d[DL_ITEMS];
    {
        const int64_t j0 = t0 + (int64_t)threadIdx.x * DL_ITEMS;
        const bool vec = j0 + DL_ITEMS <= n && ((uintptr_t)price & 15) == 0 && ((uintptr_t)amount & 15) == 0;
        if (vec) {
            double pp[DL_ITEMS], aa[DL_ITEMS];
            const double2 *qp = (const double2 *)(price + j0);
#pragma unroll
            for (int k = 0; k < DL_ITEMS / 2; ++k) { const double2 v = qp[k]; pp[2 * k] = v.x; pp[2 * k + 1] = v.y; }
            if constexpr (AF64) {
                const double2 *qa = (const double2 *)((const double *)amount + j0);
#pragma unroll
                for (int k = 0; k < DL_ITEMS / 2; ++k) { const double2 v = qa[k]; aa[2 * k] = v.x; aa[2 * k + 1] = v.y; }
            } else {
                const float4 *qa = (const float4 *)((const float *)amount + j0);
#pragma unroll
                for (int k = 0; k < DL_ITEMS / 4; ++k) {
                    const float4 v = qa[k];
                    aa[4 * k] = (double)v.x; aa[4 * k + 1] = (double)v.y; aa[4 * k + 2] = (double)v.z; aa[4 * k + 3] = (double)v.w;
                }
            }
#pragma unroll
            for (int k = 0; k < DL_ITEMS; ++k) d[k] = pp[k] * aa[k];       // rounded once, like prices[i] * volumes[i]
        } else {
#pragma unroll
            for (int k = 0; k < DL_ITEMS; ++k) d[k] = j0 + k < n ? dl_d<AF64>(price, amount, j0 + k) : 0.0;
        }
    }
    DD s = dd_make(0.0);
#pragma unroll
    for (int k = 0; k < DL_ITEMS; ++k) s = dd_add(s, d[k]);
    DD tot;
    DD ex = dl_block_exclusive(s, lds, &tot);
    // (M, r) = floor and remainder of D / thr BEFORE the thread's first tick, from the double-double prefix: one division per
    // thread.  Its 8 ticks then only add their increment to r and step M when r passes thr -- the exact-arithmetic form of the
    // reference's loop.  r is a plain double: 8 additions lose <= 8 * 2^-53 * 2 thr, five orders of magnitude inside the
    // 1e-11 * thr margin that flags a decision.  (A division, an fma and a double-double add per TICK stood here: the emit
    // pass ran at 2.3 TB/s, VALU-bound.)
    const DD D0 = dd_add(dd_add(seg_base[blockIdx.x >> DL_SEG_SHIFT], tile_base[blockIdx.x]), ex);
    double fd0, r;
    int64_t M = dd_floor_div(D0, thr, &fd0, &r);
    int frag = 0;
    const int64_t i0 = t0 + (int64_t)threadIdx.x * DL_ITEMS;
    bool big = false;                      // an increment of this thread reaches the threshold (whale trades): per-tick division
#pragma unroll
    for (int k = 0; k < DL_ITEMS; ++k) big |= d[k] >= thr;
    if (!big) {
#pragma unroll
        for (int k = 0; k < DL_ITEMS; ++k) {
            const int64_t i = i0 + k;
            G[k] = INT64_MAX;                 // neutral for min beyond the end
            if (i < n) {
                r += d[k];
                if (r >= thr) { r -= thr; M += 1; }      // d < thr: at most one step per tick
                if (rem) rem[k] = r;
                G[k] = i == 0 ? 0 : M - i;
                const double tol = fmax(1e-11, ((double)(i + 1) + extra) * 2.3e-16) * thr;
                frag += (i > 0 && fmin(r, thr - r) <= tol);
            }
        }
    } else {
        DD D = D0;
#pragma unroll
        for (int k = 0; k < DL_ITEMS; ++k) {
            const int64_t i = i0 + k;
            D = dd_add(D, d[k]);
            G[k] = INT64_MAX;
            if (i < n) {
                double fd, rr;
                const int64_t Mk = dd_floor_div(D, thr, &fd, &rr);
                if (rem) rem[k] = rr;
                G[k] = i == 0 ? 0 : Mk - i;
                const double tol = fmax(1e-11, ((double)(i + 1) + extra) * 2.3e-16);
                frag += (i > 0 && fd <= tol);
            }
        }
    }
    return frag;
}

__device__ __forceinline__ int64_t dl_block_min(int64_t v, int64_t *lds4)
{
    v = fmk_wave_min(v);
    if (fmk_lane() == 0) lds4[threadIdx.x >> 6] = v;
    __syncthreads();
    int64_t r = lds4[0];
    for (int k = 1; k < 4; ++k) r = lds4[k] < r ? lds4[k] : r;
    __syncthreads();
    return r;
}

template <bool AF64>
__global__ __launch_bounds__(DL_THREADS) void k_dl_tile_min(const double *__restrict__ price,
                                                            const void *__restrict__ amount, int64_t n, double thr,
                                                            const DD *__restrict__ tile_base, const DD *__restrict__ seg_base,
                                                            int64_t *__restrict__ tile_min)
{
    __shared__ DD lds[4];
    __shared__ int64_t lmin[4];
    int64_t G[DL_ITEMS];
    (void)dl_thread_G<AF64>(price, amount, n, thr, tile_base, seg_base, lds, G);
    int64_t m = G[0];
#pragma unroll
    for (int k = 1; k < DL_ITEMS; ++k) m = G[k] < m ? G[k] : m;
    m = dl_block_min(m, lmin);
    if (threadIdx.x == 0) tile_min[blockIdx.x] = m;
}

// exclusive prefix-min in place (one block, 8 consecutive records per thread and round); result[0] = overall minimum
__global__ __launch_bounds__(1024) void k_dl_scan_min(int64_t *__restrict__ t_all, int64_t m_all, int64_t seg_len,
                                                      int64_t *__restrict__ seg_min, int64_t *result)
{
    __shared__ int64_t ws[16];
    __shared__ int64_t run;
    const int64_t lo = (int64_t)blockIdx.x * seg_len;             // segmented like k_dl_scan_dd
    int64_t *t = t_all + lo;
    const int64_t m = m_all - lo < seg_len ? m_all - lo : seg_len;
    if (threadIdx.x == 0) run = INT64_MAX;
    __syncthreads();
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
    for (int64_t b = 0; b < m; b += 1024 * 8) {
        const int64_t i0 = b + (int64_t)threadIdx.x * 8;
        int64_t loc[8];
        int64_t mine = INT64_MAX;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            loc[k] = mine;                                         // exclusive prefix-min inside my 8 records
            const int64_t v = i0 + k < m ? t[i0 + k] : INT64_MAX;
            mine = v < mine ? v : mine;
        }
        int64_t inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int64_t o = __shfl_up(inc, d, 64);
            if (lane >= d) inc = o < inc ? o : inc;
        }
        if (lane == 63) ws[w] = inc;
        __syncthreads();
        int64_t pre = run;
        for (int k = 0; k < w; ++k) pre = ws[k] < pre ? ws[k] : pre;
        int64_t prev = __shfl_up(inc, 1, 64);
        if (lane == 0) prev = INT64_MAX;
        const int64_t ex = prev < pre ? prev : pre;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (i0 + k < m) t[i0 + k] = loc[k] < ex ? loc[k] : ex;
        __syncthreads();
        if (threadIdx.x == 1023) run = inc < pre ? inc : pre;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (seg_min) seg_min[blockIdx.x] = run;
        if (result) result[0] = run;
    }
}

template <bool AF64>
__global__ __launch_bounds__(DL_THREADS) void k_dl_emit(const double *__restrict__ price,
                                                        const void *__restrict__ amount, int64_t n, double thr,
                                                        const DD *__restrict__ tile_base, const DD *__restrict__ seg_base,
                                                        const int64_t *__restrict__ tile_premin,
                                                        const int64_t *__restrict__ seg_premin,
                                                        int64_t *__restrict__ out, int64_t cap,
                                                        unsigned long long *n_frag, int64_t *__restrict__ carry_k,
                                                        double inv_ulp, int simple, int64_t *__restrict__ last_M, double extra,
                                                        unsigned long long *__restrict__ area_out)
{
    __shared__ DD lds[4];
    __shared__ int64_t wmin[4];
    int64_t G[DL_ITEMS];
    double rem[DL_ITEMS];
    const int frag = dl_thread_G<AF64>(price, amount, n, thr, tile_base, seg_base, lds, G, rem, extra);
    // exclusive prefix-min over the block in tick order: thread-local then across threads
    int64_t tmin = G[0];
#pragma unroll
    for (int k = 1; k < DL_ITEMS; ++k) tmin = G[k] < tmin ? G[k] : tmin;
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
    int64_t inc = tmin;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int64_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc = o < inc ? o : inc;
    }
    if (lane == 63) wmin[w] = inc;
    __syncthreads();
    int64_t pre;                                     // min of all G before this tile (INT64_MAX for tile 0)
    if (simple) {
        // No increment reaches the threshold (d_max < thr): M grows by at most 1 per tick, so G_j = M_j - j never increases and
        // the prefix-min IS the previous tick's value -- G at the tick before this tile, from the tile's own base prefix.  The
        // whole minima pass (k_dl_tile_min + k_dl_scan_min: a third read of price and amount) is not needed: K_i = M_i.
        pre = INT64_MAX;
        if (blockIdx.x > 0) {
            double fd;
            const DD base = dd_add(seg_base[blockIdx.x >> DL_SEG_SHIFT], tile_base[blockIdx.x]);
            pre = dd_floor_div(base, thr, &fd) - ((int64_t)blockIdx.x * DL_TILE - 1);
        }
    } else {
        pre = tile_premin[blockIdx.x];
        const int64_t spre = seg_premin[blockIdx.x >> DL_SEGM_SHIFT];  // inside its segment, and of the segments before
        pre = spre < pre ? spre : pre;
    }
    for (int k = 0; k < w; ++k) pre = wmin[k] < pre ? wmin[k] : pre;
    int64_t prev = __shfl_up(inc, 1, 64);
    if (lane == 0) prev = INT64_MAX;
    int64_t mex = prev < pre ? prev : pre;           // min_{j < first tick of this thread} G_j
    const int64_t i0 = (int64_t)blockIdx.x * DL_TILE + (int64_t)threadIdx.x * DL_ITEMS;
    int64_t area = 0;                                // sum over this thread's ticks of the backlog M_i - K_i (in thresholds)
#pragma unroll
    for (int k = 0; k < DL_ITEMS; ++k) {
        const int64_t i = i0 + k;
        if (i < n) {
            if (!simple && i >= 1 && G[k] > mex) area += G[k] - mex;
            if (i >= 1 && G[k] >= mex) {             // close: K_i = i + mex is its slot
                const int64_t slot = i + mex;
                if (slot < cap) {
                    out[slot] = i;
                    // exact-arithmetic carry after this close in units of ulp(thr) (D_i - M_i*thr: with d_max < thr there is
                    // no backlog, M_i == K_i) -- the start state of the next bar's simulation in fmk_dollar_exact.hip
                    // With a backlog (an increment >= thr behind it: M_i > K_i) the state is (M_i - K_i) thresholds higher:
                    // M_i - K_i = G_i - min_{j<i} G_j.  Beyond 500 thresholds the unit count leaves int64's comfortable range: -1 (no
                    // exact tier for this stream).
                    const int64_t back = G[k] - mex;
                    carry_k[slot] = back == 0 ? llrint(rem[k] * inv_ulp)
                                              : (back <= 500 ? llrint(rem[k] * inv_ulp) + back * llrint(thr * inv_ulp) : -1);
                }
            }
            mex = G[k] < mex ? G[k] : mex;
            if (simple && i == n - 1) *last_M = G[k] + i;    // closes of the whole stream = M_{n-1}
        }
    }
    if (i0 == 0 && cap > 0) out[0] = 0;              // logic.py:138
    int f = (int)fmk_wave_sum(frag);
    if (lane == 0 && f) atomicAdd(n_frag, (unsigned long long)f);
    if (!simple) {
        // The backlog AREA of the stream, sum_i (M_i - K_i): the reference's add at tick i happens at a magnitude of at most
        // (M_i - K_i + 2) thr, so the rounding drift its backlogs add to the per-tick bound is 2^-52 thr x area / 2.  (The bound
        // the closed form certifies with, (W / thr)^2, assumes ALL increments >= thr pile into one backlog: at 1e5 block trades
        // of 1.2 thr each that is 1.3e10 tick equivalents where the area is 1e5; the exact tier flags its bars with the area.)
        area = fmk_wave_sum(area);
        if (lane == 0 && area) atomicAdd(area_out, (unsigned long long)area);
    }
}

struct DlCache {
    fmk_ctx *ctx;
    const void *amount;
    const double *price;
    int64_t n;
    double thr;
    int is_f64;
    int64_t count, unc;
    int64_t *dbuf;
    int64_t *carry;      // carry_k per close, same capacity
    int64_t cap;
    double dmax;
    double extra;        // the backlog term of the drift bound, in ticks ((W / thr)^2, W the sum of the increments >= thr)
    double area;         // ... and its tight form: the backlog area sum_i (M_i - K_i) (k_dl_emit), what the exact tier flags with
};
static DlCache &dl_cache(fmk_ctx *ctx)       // one per context (slot 1), created on first use
{
    if (!ctx->idx_cache[1]) ctx->idx_cache[1] = new DlCache();
    return *(DlCache *)ctx->idx_cache[1];
}

void fmk_dollar_trim(fmk_ctx *ctx)
{
    DlCache *c = (DlCache *)ctx->idx_cache[1];
    if (!c) return;
    if (c->dbuf) (void)hipFree(c->dbuf);
    if (c->carry) (void)hipFree(c->carry);
    delete c;
    ctx->idx_cache[1] = nullptr;
}

template <bool AF64>
static int dl_run(fmk_ctx *ctx, const double *p, const void *a, int64_t n, double thr, DlCache &c)
{
    const int64_t tiles = fmk_ceil_div(n, DL_TILE);
    void *scr;
    const int64_t gdd = fmk_ceil_div(tiles, (int64_t)1 << DL_SEG_SHIFT), gmn = fmk_ceil_div(tiles, (int64_t)1 << DL_SEGM_SHIFT);
    FMK_TRY(fmk_scratch(ctx, (size_t)tiles * (sizeof(DD) + 8 + 8) + (size_t)gdd * sizeof(DD) + (size_t)gmn * 8 + 64, &scr));
    DD *tsum = (DD *)scr;
    DD *segb = tsum + tiles;
    int64_t *tmin = (int64_t *)(segb + gdd);
    int64_t *segm = tmin + tiles;
    double *wpart = (double *)(segm + gmn + 1);                        // per-tile sums of the increments >= thr
    int64_t *d_res = ctx->d_mail + 24;
    int *d_bad = (int *)(ctx->d_mail + 26);
    unsigned long long *d_dmax = (unsigned long long *)(ctx->d_mail + 27);
    double *d_whale = (double *)(ctx->d_mail + 28);
    FMK_HIP(ctx, hipMemsetAsync(d_bad, 0, 24, ctx->stream));
    k_dl_tile_sums<AF64><<<(unsigned)tiles, DL_THREADS, 0, ctx->stream>>>(p, a, n, tsum, d_bad, d_dmax, thr, wpart);
    k_dl_whale_total<<<(unsigned)(tiles < 512 * 256 ? fmk_ceil_div(tiles, 256) : 512), 256, 0, ctx->stream>>>(wpart, tiles, thr, d_whale);
    FMK_LAUNCH_CHECK(ctx);
    k_dl_scan_dd<<<(unsigned)gdd, DL_THREADS, 0, ctx->stream>>>(tsum, tiles, (int64_t)1 << DL_SEG_SHIFT, segb);
    DD *d_total = (DD *)(ctx->d_mail + 44);                            // sum of all increments (double-double)
    k_dl_scan_dd<<<1, DL_THREADS, 0, ctx->stream>>>(segb, gdd, gdd, d_total);
    FMK_LAUNCH_CHECK(ctx);
    // what pass 1 learned: negative / NaN increments (-> serial walk), the largest increment, the total
    FMK_HIP(ctx, hipMemcpyAsync(ctx->h_mail + 1, d_bad, 24, hipMemcpyDeviceToHost, ctx->stream));
    FMK_HIP(ctx, hipMemcpyAsync(ctx->h_mail + 4, d_total, 16, hipMemcpyDeviceToHost, ctx->stream));
    FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if ((int)ctx->h_mail[1] != 0) return 1;
    memcpy(&c.dmax, &ctx->h_mail[2], 8);
    // The drift bound of a decision.  Without whales every add of the reference happens below 2 thr: <= (i + 1) * 2^-52 * thr after i
    // ticks.  An increment w >= thr leaves a backlog -- cum falls from ~w to below thr by one threshold per tick -- and the adds of
    // those ~w / thr ticks round at the magnitude of cum: their errors sum to <= 2^-53 * w^2 / (2 thr) ... for ALL whales together
    // (a second one may arrive before the first is worked off) <= 2^-53 * W^2 / thr, W their sum.  In units of 2.3e-16 * thr that is
    // `extra` = (W / thr)^2 more "ticks".  (Found by tools/fuzz_volume.py seed 81003, dollar cases 792 / 1143, late in round 3: with
    // the tick count alone two closes right after a whale were certified and wrong by one tick.)
    double whale_thr = 0.0;                                           // W / thr (k_dl_whale_total)
    memcpy(&whale_thr, &ctx->h_mail[3], 8);
    const double extra = whale_thr * whale_thr;
    c.extra = extra;
    const int simple = c.dmax < thr ? 1 : 0;
    if (!simple) {
        k_dl_tile_min<AF64><<<(unsigned)tiles, DL_THREADS, 0, ctx->stream>>>(p, a, n, thr, tsum, segb, tmin);
        FMK_LAUNCH_CHECK(ctx);
        k_dl_scan_min<<<(unsigned)gmn, 1024, 0, ctx->stream>>>(tmin, tiles, (int64_t)1 << DL_SEGM_SHIFT, segm, nullptr);
        k_dl_scan_min<<<1, 1024, 0, ctx->stream>>>(segm, gmn, gmn, nullptr, d_res);
        FMK_LAUNCH_CHECK(ctx);
        FMK_HIP(ctx, hipMemcpyAsync(ctx->h_mail, d_res, 8, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const int64_t gmin = ctx->h_mail[0] < 0 ? ctx->h_mail[0] : 0; // G_0 = 0 is part of every prefix
        c.count = (n - 1) + gmin + 1;                                   // K_{n-1} closes + the leading 0
    } else {
        // two passes: the number of closes is M_{n-1} = floor(D_{n-1} / thr), from the double-double total (same arithmetic as
        // dd_floor_div on the device; the emit pass reports the value it used and the two are compared)
        double hi, lo;
        memcpy(&hi, &ctx->h_mail[4], 8);
        memcpy(&lo, &ctx->h_mail[5], 8);
        double q = floor(hi / thr);
        const double pq = q * thr, eq = fma(q, thr, -pq);
        double r = (hi - pq) + (lo - eq);
        while (r < 0.0) { q -= 1.0; r += thr; }
        while (r >= thr) { q += 1.0; r -= thr; }
        c.count = (int64_t)q + 1;
    }
    if (c.dbuf && c.cap < c.count + DL_EXTRA) {
        FMK_HIP(ctx, hipFree(c.dbuf));
        FMK_HIP(ctx, hipFree(c.carry));
        c.dbuf = c.carry = nullptr;
    }
    if (!c.dbuf) {       // DL_EXTRA spare slots: the exact tier may find a few more closes at the end of the stream
        c.cap = c.count + DL_EXTRA;
        FMK_HIP(ctx, hipMalloc((void **)&c.dbuf, (size_t)c.cap * 8));
        FMK_HIP(ctx, hipMalloc((void **)&c.carry, (size_t)c.cap * 8));
    }
    unsigned long long *d_frag = (unsigned long long *)(ctx->d_mail + 25);
    unsigned long long *d_area = (unsigned long long *)(ctx->d_mail + 46);     // [d_frag + 1 would be d_bad: keep the area apart]
    FMK_HIP(ctx, hipMemsetAsync(d_frag, 0, 8, ctx->stream));
    FMK_HIP(ctx, hipMemsetAsync(d_area, 0, 8, ctx->stream));
    int ex;
    (void)frexp(thr, &ex);                                            // thr = m * 2^ex, m in [0.5, 1): ulp(thr) = 2^(ex - 53)
    FMK_HIP(ctx, hipMemsetAsync(d_res, 0xFF, 8, ctx->stream));
    k_dl_emit<AF64><<<(unsigned)tiles, DL_THREADS, 0, ctx->stream>>>(p, a, n, thr, tsum, segb, tmin, segm, c.dbuf,
                                                                     simple ? c.cap : c.count, d_frag, c.carry,
                                                                     ldexp(1.0, 53 - ex), simple, d_res, extra, d_area);
    FMK_LAUNCH_CHECK(ctx);
    FMK_HIP(ctx, hipMemcpyAsync(ctx->h_mail, d_frag, 8, hipMemcpyDeviceToHost, ctx->stream));
    FMK_HIP(ctx, hipMemcpyAsync(ctx->h_mail + 1, d_res, 8, hipMemcpyDeviceToHost, ctx->stream));
    FMK_HIP(ctx, hipMemcpyAsync(ctx->h_mail + 2, d_area, 8, hipMemcpyDeviceToHost, ctx->stream));
    FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    c.unc = ctx->h_mail[0];
    c.area = (double)ctx->h_mail[2];
    if (simple && ctx->h_mail[1] + 1 != c.count) {
        // the device's D_{n-1} (segment base + tile base + in-tile prefix) and the host's total are the same sum in two
        // association orders: they can only disagree about floor(D / thr) when D is within ~2^-100 of a multiple of thr
        const int64_t dev = ctx->h_mail[1] + 1;
        if (dev < 1 || dev > c.cap)
            return fmk_set_error(ctx, FMK_E_HIP, "dollar indexer: %lld closes on the device, %lld from the total",
                                 (long long)(dev - 1), (long long)(c.count - 1));
        c.count = dev;
    }
    return FMK_OK;
}

#include "fmk_dollar_onepass.h"

// The one-pass closed form (fmk_dollar_onepass.h).  -> FMK_OK (c.count / c.unc / c.dbuf / c.carry filled like dl_run does),
// 1 (not served: an increment outside [0, thr), a threshold outside the fixed-point range, a look-back that gave up -- the caller
// runs dl_run), or an error.
template <bool AF64>
static int dl1_run(fmk_ctx *ctx, const double *p, const void *a, int64_t n, double thr, DlCache &c)
{
    int ex;
    (void)frexp(thr, &ex);               // thr = m * 2^ex, m in [0.5, 1): ulp(thr) = 2^(ex - 53)
    if (!(thr > 0.0) || !isfinite(thr) || ex < -900 || ex > 900 || n < 2) return 1;
    Dl1Params P;
    P.scale = ldexp(1.0, DL1_F + 53 - ex);
    P.T = (uint64_t)(thr * P.scale);     // exact: thr / ulp(thr) is an integer in [2^52, 2^53)
    P.inv_t = 1.0 / (double)P.T;
    memcpy(&P.thr_bits, &thr, 8);
    P.tol_a = 1e-11 * (double)P.T;
    P.tol_b = 2.31e-16 * (double)P.T;    // the reference's drift per add (2.3e-16 thr, dl_run) + the truncation of the fixed point (2^-60 thr)
    // tile geometry: 512 threads x 16 ticks (8 192-tick tiles).  A tile waits ~5 us for its look-back with its ticks in registers and
    // loads nothing meanwhile, and a poll costs what a cache line of ticks costs, so few large tiles win: 256 x 8 / 256 x 16 / 512 x 8
    // / 512 x 16 / 1024 x 16 ran 3.9 / 3.1 / 3.3 / 2.7 / 3.2 ms per 1e9 ticks (profiles/r05_cfg3.txt).
    const int threads = 512, items = 16;
    const int64_t tiles = fmk_ceil_div(n, (int64_t)threads * items);
    const int64_t groups = (tiles + DL1_W1 - 1) / DL1_W1;
    const size_t desc_bytes = (size_t)tiles * 16 + (size_t)groups * DL1_W1 * 16;      // A[tiles][2], B[W1][groups][2] (fmk_dollar_onepass.h)
    void *scr;
    FMK_TRY(fmk_scratch(ctx, desc_bytes + (size_t)tiles * 32 + 64, &scr));            // (+ the time stamps of a -DDL1_TIMING build)
    unsigned long long *desc = (unsigned long long *)scr;
    int64_t *d_last = ctx->d_mail + 24;
    unsigned long long *d_frag = (unsigned long long *)(ctx->d_mail + 25);
    int *d_flags = (int *)(ctx->d_mail + 26);
    double *d_sum = (double *)(ctx->d_mail + 27);
    if (!c.dbuf || c.cap <= 0) {
        // no close buffers yet: their capacity from a strided sample of the products (x 1.3 + slack).  Only a guess -- the pass
        // itself reports the true count, and a guess that was too small costs one more pass with the exact capacity.
        const int64_t m = n < 65536 ? n : 65536, stride = n / m;
        FMK_HIP(ctx, hipMemsetAsync(d_sum, 0, 8, ctx->stream));
        k_dl1_sample<AF64><<<(unsigned)fmk_ceil_div(m, 256), 256, 0, ctx->stream>>>(p, a, n, stride, m, d_sum);
        FMK_LAUNCH_CHECK(ctx);
        FMK_HIP(ctx, hipMemcpyAsync(ctx->h_mail, d_sum, 8, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        double ssum;
        memcpy(&ssum, &ctx->h_mail[0], 8);
        double est = ssum / (double)m * (double)n / thr;
        if (!(est >= 0.0) || est > (double)n) est = (double)n;
        c.cap = (int64_t)(est * 1.3) + 65536 + DL_EXTRA;
        if (c.cap > n + 1 + DL_EXTRA) c.cap = n + 1 + DL_EXTRA;
        if (c.dbuf) { FMK_HIP(ctx, hipFree(c.dbuf)); c.dbuf = nullptr; }
        if (c.carry) { FMK_HIP(ctx, hipFree(c.carry)); c.carry = nullptr; }
        FMK_HIP(ctx, hipMalloc((void **)&c.dbuf, (size_t)c.cap * 8));
        FMK_HIP(ctx, hipMalloc((void **)&c.carry, (size_t)c.cap * 8));
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        FMK_HIP(ctx, hipMemsetAsync(desc, 0, desc_bytes, ctx->stream));
        unsigned long long *d_ticket = (unsigned long long *)((char *)desc + desc_bytes + (size_t)tiles * 32);
        FMK_HIP(ctx, hipMemsetAsync(d_ticket, 0, 8, ctx->stream));
        FMK_HIP(ctx, hipMemsetAsync(d_last, 0, 24, ctx->stream));
        unsigned long long *dB = desc + 2 * tiles;
        k_dl1<AF64, 16, 512><<<(unsigned)tiles, 512, 0, ctx->stream>>>(p, a, n, P, desc, dB, c.dbuf, c.carry, c.cap, d_last, d_frag, d_flags, d_ticket);
        FMK_LAUNCH_CHECK(ctx);
        FMK_HIP(ctx, hipMemcpyAsync(ctx->h_mail, d_last, 24, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
#ifdef DL1_TIMING
        {
            unsigned long long *tt = (unsigned long long *)malloc((size_t)tiles * 32);
            (void)hipMemcpy(tt, (char *)desc + desc_bytes, (size_t)tiles * 32, hipMemcpyDeviceToHost);
            double s_load = 0, s_lb = 0, s_cmp = 0, s_polls = 0; int64_t cnt = 0;
            unsigned long long t_min = ~0ULL, t_max = 0;
            for (int64_t t = 1; t < tiles - 1; ++t) {
                unsigned long long *q = tt + 4 * t;
                { const unsigned long long np = (0ULL - (q[0] >> 56)) & 0xFF; s_polls += (double)np; q[0] += np << 56; }
                s_load += (double)(q[1] - q[0]); s_lb += (double)(q[2] - q[1]); s_cmp += (double)(q[3] - q[2]);
                if (q[0] < t_min) t_min = q[0];
                if (q[3] > t_max) t_max = q[3];
                ++cnt;
            }
            fprintf(stderr, "[dl1] tiles %lld: start->sum %.2f us, look-back %.2f us, rest %.2f us, %.2f polls; kernel span %.3f ms\n",
                    (long long)tiles, s_load / cnt * 0.01, s_lb / cnt * 0.01, s_cmp / cnt * 0.01, s_polls / cnt, (double)(t_max - t_min) * 1e-5);
            free(tt);
        }
#endif
        if ((int)(ctx->h_mail[2] & 0xFFFFFFFF) != 0) return 1;
        const int64_t count = ctx->h_mail[0] + 1;                    // M_{n-1} closes + the leading 0
        if (count + DL_EXTRA <= c.cap) {
            c.count = count;
            c.unc = ctx->h_mail[1];
            c.dmax = nextafter(thr, 0.0);                             // all the callers ask is whether an increment reached thr
            c.extra = 0.0;
            c.area = 0.0;
            return FMK_OK;
        }
        FMK_HIP(ctx, hipFree(c.dbuf));
        FMK_HIP(ctx, hipFree(c.carry));
        c.dbuf = c.carry = nullptr;
        c.cap = count + DL_EXTRA;
        FMK_HIP(ctx, hipMalloc((void **)&c.dbuf, (size_t)c.cap * 8));
        FMK_HIP(ctx, hipMalloc((void **)&c.carry, (size_t)c.cap * 8));
    }
    return fmk_set_error(ctx, FMK_E_HIP, "dollar indexer: the close count changed between two passes over the same ticks");
}

int fmk_threshold_serial(fmk_ctx *ctx, int dollar, const double *d_price, const void *d_amount, int is_f64, int64_t n,
                         double thr, int64_t *d_close_idx, int64_t capacity, int64_t *n_idx, int64_t *n_unc);
// fmk_dollar_exact.hip: the reference's float64 state at every bar start, reconstructed in parallel; rewrites the closes the
// closed form got wrong in `close_idx` (capacity: count + DL_EXTRA).  *status: 0 = done (now exact), 1 = not applicable /
// gave up (caller takes the serial walk).
int fmk_dollar_exact(fmk_ctx *ctx, const double *d_price, const void *d_amount, int is_f64, int64_t n, double thr,
                     int64_t *d_close_idx, const int64_t *d_carry_k, int64_t *count, int *status, double extra_ticks, int whales);

// which path answered the last fmk_dollar_bar_indexer[_dev] call of this process (tests; include/fmk_diag.h): 0 the closed form alone
// (every decision certain), 1 closed form + exact tier, 2 closed form + exact tier on a stream with increments >= thr (stretch walk),
// 3 the serial walk (the fill half of a count-then-fill pair answers from the cache and leaves the value alone)
static int g_dl_last_path = -1;
static int g_dl_onepass = 0;           // the closed form of the last call came from the one-pass kernel (fmk_dollar_onepass.h)
extern "C" int fmk_diag_dollar_last(int64_t *path)
{
    *path = g_dl_last_path;
    return FMK_OK;
}

extern "C" int fmk_dollar_bar_indexer_dev(fmk_ctx *ctx, const double *d_price, const void *d_amount,
                                          int amount_is_f64, int64_t n, double threshold, int64_t *d_close_idx,
                                          int64_t capacity, int64_t *n_idx, int64_t *n_uncertified)
{
    if (n <= 0) return fmk_set_error(ctx, FMK_E_ARG, "threshold indexer: empty input");
    // the closed form needs thr > 0 and non-negative increments; anything else takes the serial walk
    if (!(threshold > 0.0) || getenv("FMK_THRESHOLD_SERIAL")) {
        g_dl_last_path = 3;
        return fmk_threshold_serial(ctx, 1, d_price, d_amount, amount_is_f64, n, threshold, d_close_idx, capacity,
                                    n_idx, n_uncertified);
    }
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    DlCache &c = dl_cache(ctx);
    const bool hit = c.ctx == ctx && !ctx->idx_stale[1] && c.amount == d_amount && c.price == d_price && c.n == n && c.thr == threshold &&
                     c.is_f64 == amount_is_f64 && c.dbuf && d_close_idx;
    if (!hit) {
        int rc = amount_is_f64 ? dl1_run<true>(ctx, d_price, d_amount, n, threshold, c)
                               : dl1_run<false>(ctx, d_price, d_amount, n, threshold, c);
        g_dl_onepass = rc == FMK_OK;
        if (rc == 1)
            rc = amount_is_f64 ? dl_run<true>(ctx, d_price, d_amount, n, threshold, c)
                               : dl_run<false>(ctx, d_price, d_amount, n, threshold, c);
        if (rc == 1) {
            g_dl_last_path = 3;
            return fmk_threshold_serial(ctx, 1, d_price, d_amount, amount_is_f64, n, threshold, d_close_idx, capacity,
                                        n_idx, n_uncertified);
        }
        if (rc) return rc;
        g_dl_last_path = 0;
        c.ctx = ctx; c.amount = d_amount; c.price = d_price; c.n = n; c.thr = threshold; c.is_f64 = amount_is_f64;
        ctx->idx_key[1][0] = d_amount; ctx->idx_key[1][1] = d_price; ctx->idx_stale[1] = 0;
    }
    // test knob (read per call): FMK_DL_FORCE_EXACT_TIER=1 runs the exact tier even when the closed form is already certain
    const char *fv = getenv("FMK_DL_FORCE_EXACT_TIER");
    const int force_sim = fv ? atoi(fv) : 0;
    if ((c.unc > 0 || force_sim) && !ctx->fast_threshold) {
        // a sum within the reference's rounding drift of the threshold: only its own sequence of float64 operations decides
        // like it does.  The exact tier reconstructs the reference's float64 state at every bar start (its rounding errors
        // are a function of the bar's own ticks and of the carried state modulo 4 ulp(thr)) and replays the few fragile bars
        // from it; streams it does not cover (an increment >= thr) take the serial walk (fmk_threshold.hip)
        int status = 1;
        if (!hit) {
            int rc = fmk_dollar_exact(ctx, d_price, d_amount, amount_is_f64, n, threshold, c.dbuf, c.carry, &c.count, &status,
                                      c.dmax >= threshold ? c.area : 0.0, c.dmax >= threshold ? 1 : 0);
            if (rc) { c.ctx = nullptr; return rc; }
            if (status == 0) { c.unc = 0; g_dl_last_path = c.dmax >= threshold ? 2 : 1; }
        } else if (hit) {
            status = c.unc == 0 ? 0 : 1;
        }
        if (status != 0) {
            c.ctx = nullptr;
            g_dl_last_path = 3;
            return fmk_threshold_serial(ctx, 1, d_price, d_amount, amount_is_f64, n, threshold, d_close_idx, capacity, n_idx,
                                        n_uncertified);
        }
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
