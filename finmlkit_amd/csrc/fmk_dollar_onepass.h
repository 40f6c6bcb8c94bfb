// fmk_dollar_onepass.h -- _dollar_bar_indexer (finmlkit/bar/logic.py:118-149): the closed form of fmk_dollar.hip in ONE pass over
// price and amount (round 5).  Included by fmk_dollar.hip behind its DlCache.
//
// The closed form needs, per tick, M_i = floor(D_i / thr) and r_i = D_i - M_i thr of the prefix sum D of the rounded products.  The
// reduce-then-scan of rounds 1-4 read both columns twice (k_dl_tile_sums, k_dl_emit: 24 B/tick) and carried D as a double-double
// (~140 VALU instructions per tick in the emit pass, which was bound by them).  Here:
//
//  * FIXED POINT instead of double-double.  With u = ulp(thr), every product d < thr becomes the integer x = trunc(d / s),
//    s = u / 256: thr / s = T is an exact integer below 2^61, the sums inside a tile are
//    96-bit integers (three dwords on the DPP data path: 6 instructions per scan step against ~24 for a double-double), and the
//    running remainder is one 64-bit add, one subtract, one compare and a select per tick.  Products of d >= thr / 1024 are
//    represented exactly (their own ulp is >= s); smaller ones lose less than s each: after i ticks the prefix is short by less
//    than (i + 1) u / 256, i.e. 1/256 of the reference's own rounding drift per add -- it is added to the drift bound that flags
//    a decision (2.31e-16 per tick instead of 2.3e-16), and the exact tier widens its margin by the same 1/256.
//  * ONE pass: the tile prefix comes from a decoupled look-back over the tiles in front (a descriptor of two self-validating
//    64-bit words per tile: 2 tag bits + 62 payload bits each, so aggregate and inclusive prefix need no fence and cannot be read
//    torn), not from a separate sum kernel.  12 B/tick.
//
// Scope: streams whose increments are all in [0, thr) -- no negative / NaN products, no block trades.  A tile that sees anything
// else raises a flag and the call takes the kernels of fmk_dollar.hip (prefix-min form, whale bounds, serial walk) exactly as
// before.  The look-back waits for tiles with smaller block indices; the hardware dispatches workgroups in index order, so the
// lowest unfinished tile is always resident -- should that ever not hold, the wait gives up after DL1_SPIN_LIMIT polls, raises
// the same flag, and the call falls back instead of hanging.
#pragma once
#include "fmk_dpp.h"
#ifndef DL1_CACHED_FIRST
#define DL1_CACHED_FIRST 0                 // 1: the first poll through the caches (measured: no gain, profiles/r05_cfg3.txt)
#endif

#define DL1_F 8                          // fraction bits below ulp(thr)
#define DL1_FLAG_RANGE 1                 // an increment outside [0, thr)
#define DL1_FLAG_TIMEOUT 2               // a look-back gave up
#define DL1_SPIN_LIMIT (1 << 22)         // polls of one descriptor (each >= 1 us of memory latency)
#define DL1_MASK62 0x3FFFFFFFFFFFFFFFULL

typedef unsigned __int128 dl1_u128;

struct Dl1U96 { uint64_t lo; uint32_t hi; };
__device__ __forceinline__ Dl1U96 dl1_add(Dl1U96 a, Dl1U96 b)
{
    Dl1U96 r;
    r.lo = a.lo + b.lo;
    r.hi = a.hi + b.hi + (r.lo < a.lo ? 1u : 0u);
    return r;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Dl1U96 dl1_dpp(Dl1U96 v)                     // lanes without a source receive 0
{
    Dl1U96 r;
    r.lo = (uint64_t)fmk_dpp<CTRL, ROW_MASK>((int64_t)0, (int64_t)v.lo);
    r.hi = (uint32_t)fmk_dpp_i32<CTRL, ROW_MASK>(0, (int)v.hi);
    return r;
}
__device__ __forceinline__ Dl1U96 dl1_wave_iscan(Dl1U96 v)
{
    v = dl1_add(dl1_dpp<FMK_DPP_ROW_SHR(1), 0xF>(v), v);
    v = dl1_add(dl1_dpp<FMK_DPP_ROW_SHR(2), 0xF>(v), v);
    v = dl1_add(dl1_dpp<FMK_DPP_ROW_SHR(4), 0xF>(v), v);
    v = dl1_add(dl1_dpp<FMK_DPP_ROW_SHR(8), 0xF>(v), v);
    v = dl1_add(dl1_dpp<FMK_DPP_ROW_BCAST15, 0xA>(v), v);
    v = dl1_add(dl1_dpp<FMK_DPP_ROW_BCAST31, 0xC>(v), v);
    return v;
}
__device__ __forceinline__ dl1_u128 dl1_wide(Dl1U96 v) { return ((dl1_u128)v.hi << 64) | v.lo; }

// sum over the wave of a 128-bit value (look-back only: one wave per tile)
__device__ __forceinline__ dl1_u128 dl1_wave_sum128(dl1_u128 v)
{
    uint64_t lo = (uint64_t)v, hi = (uint64_t)(v >> 64);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t l2 = (uint64_t)__shfl_xor((long long)lo, o, 64), h2 = (uint64_t)__shfl_xor((long long)hi, o, 64);
        const uint64_t s = lo + l2;
        hi = hi + h2 + (s < lo ? 1u : 0u);
        lo = s;
    }
    return ((dl1_u128)hi << 64) | lo;
}

__device__ __forceinline__ void dl1_publish(unsigned long long *d, dl1_u128 v, unsigned tag)
{
    const unsigned long long w0 = ((unsigned long long)tag << 62) | ((unsigned long long)v & DL1_MASK62);
    const unsigned long long w1 = ((unsigned long long)tag << 62) | ((unsigned long long)(v >> 62) & DL1_MASK62);
    __hip_atomic_store(d, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(d + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ dl1_u128 dl1_value(unsigned long long w0, unsigned long long w1)
{
    return ((dl1_u128)(w1 & DL1_MASK62) << 62) | (w0 & DL1_MASK62);
}

// wave 0 of tile `tile` (> 0): the sum of all products in front of the tile, in TWO memory round trips whatever the number of
// tiles in flight.  A plain decoupled look-back (64 aggregates per round trip until a tile with an inclusive prefix turns up) lets
// the inclusive frontier advance by 64 tiles per round trip (~0.5 us: the descriptors of other XCDs' tiles come from memory) --
// 2 048-tick tiles: 2.6e11 ticks/s, the kernel ran 3.9 ms per 1e9 ticks and 1.98 without the look-back (profiles/r05_cfg3.txt).
// Hence two descriptor arrays:
//   A[t]  tag 1: the tile's own sum
//   B[t]  tag 1: R(t) = sum of A over the 64 tiles in front of t (what t's first round reads anyway)
//         tag 2: base(t) = sum of A over ALL tiles in front of t
// Round 1: lane l waits for A[t - 1 - l]: R(t), published at once.  Round 2: lane l waits for B[t - 64 (l + 1)]; R(t - 64 j) covers
// tiles [t - 64 (j + 1), t - 64 j), so base(t) = R(t) + sum of R(u_j) in front of the nearest u_l that already has its base
// + that base: 4 096 tiles per round, further rounds (in practice never) likewise.
// B is laid out by residue: B[(t mod 64) * groups + t / 64], groups = ceil(tiles / 64) -- the 64 entries one look-back polls
// (t - 64, t - 128, ...) are then 1 KB of consecutive memory instead of 64 separate lines (with B[t] in tile order every poll was
// 128 memory requests per tile: the descriptor traffic rivalled the ticks' and a poll took ~3 us)
#ifndef DL1_W1
#define DL1_W1 64                        // tiles one R covers (a power of two <= 64)
#endif
__device__ __forceinline__ int64_t dl1_bslot(int64_t t, int64_t groups) { return (t & (DL1_W1 - 1)) * groups + t / DL1_W1; }
__device__ __forceinline__ dl1_u128 dl1_lookback(unsigned long long *A, unsigned long long *B, int64_t tile, int64_t groups, int *flags, bool *gave_up, int *npolls)
{
    const int lane = fmk_lane();
    *gave_up = false;
    // Both rounds' reads are in flight TOGETHER (a round trip to another XCD's descriptor is ~2.5 us under load, and a tile that
    // waits holds its registers without loading anything: one after the other the look-back took 9.4 us per tile and the kernel
    // 4.3 ms; profiles/r05_cfg3.txt): lane l polls A[tile - 1 - l] and B[tile - 64 (l + 1)] in one loop.
    const int64_t ia = lane < DL1_W1 ? tile - 1 - lane : -1, ib = tile - DL1_W1 - DL1_W1 * (int64_t)lane;
    unsigned long long a0 = 0, a1 = 0, b0 = 0, b1 = 0;
    bool have_a = ia < 0, have_b = ib <= 0 || tile <= DL1_W1;            // nothing there: a sum of 0 / a base of 0
    unsigned tb = 2;
    bool published = false, dead = false;
    dl1_u128 r = 0;
    for (int polls = 1;; ++polls) {
        if (polls == 1 && DL1_CACHED_FIRST) {
            // first look through the caches: descriptors only move forward (0 -> sum -> base), so a stale copy is never wrong, only
            // older; the tiles of an XCD poll overlapping windows, and what one of them has fetched the others find in L2
            if (!have_a) {
                a0 = __hip_atomic_load(A + 2 * ia, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                a1 = __hip_atomic_load(A + 2 * ia + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (!have_b) {
                b0 = __hip_atomic_load(B + 2 * dl1_bslot(ib, groups), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                b1 = __hip_atomic_load(B + 2 * dl1_bslot(ib, groups) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {
            if (!have_a) {
                a0 = __hip_atomic_load(A + 2 * ia, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                a1 = __hip_atomic_load(A + 2 * ia + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (!have_b) {
                b0 = __hip_atomic_load(B + 2 * dl1_bslot(ib, groups), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b1 = __hip_atomic_load(B + 2 * dl1_bslot(ib, groups) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (!have_a) { const unsigned t = (unsigned)(a0 >> 62); have_a = t != 0 && t == (unsigned)(a1 >> 62); }
        if (!have_b) {
            tb = (unsigned)(b0 >> 62);
            have_b = tb != 0 && tb == (unsigned)(b1 >> 62);
            dead = dead || (have_b && tb == 3);
        }
        if (!published && __ballot(!have_a) == 0) {                  // R(tile): at once, the tiles behind are waiting for it
            r = dl1_wave_sum128(dl1_value(a0, a1));
            published = true;
            if (tile > DL1_W1 && lane == 0) dl1_publish(B + 2 * dl1_bslot(tile, groups), r, 1);
        }
        // the nearest B with a base ends the walk: everything in front of it must be there, nothing behind it is needed
        const uint64_t based = __ballot(have_b && tb == 2), missing = __ballot(!have_b);
        const int first = based ? __builtin_ctzll(based) : 64;
        const bool front_ok = (missing & (first >= 63 ? ~0ULL : ((2ULL << first) - 1))) == 0;
        if (published && based != 0 && front_ok) {
            *npolls = polls;
            if (tile <= DL1_W1) return r;
            return r + dl1_wave_sum128(lane <= first ? dl1_value(b0, b1) : (dl1_u128)0);
        }
        bool stop = dead || polls >= DL1_SPIN_LIMIT;
        if ((polls & 63) == 0 && (__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & DL1_FLAG_TIMEOUT)) stop = true;
        // (no tile within 4 096 has its base although all have their R: cannot happen with fewer tiles in flight than that; if it
        //  ever does, the call falls back rather than walking on)
        if (published && based == 0 && missing == 0) stop = true;
        if (__ballot(stop) != 0) {
            if (lane == 0) atomicOr(flags, DL1_FLAG_TIMEOUT);
            *gave_up = true;
            return 0;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// floor(P / T) and the remainder, T < 2^61, P / T < 2^41
__device__ __forceinline__ void dl1_divmod(dl1_u128 P, uint64_t T, double inv_t, int64_t *q_out, uint64_t *r_out)
{
    const double pd = fma((double)(uint64_t)(P >> 64), 18446744073709551616.0, (double)(uint64_t)P);
    int64_t q = (int64_t)(pd * inv_t);                               // within 1 of the true quotient
    int64_t r = (int64_t)((uint64_t)P - (uint64_t)q * T);            // exact modulo 2^64, and the true value lies in (-2T, 3T)
    if (r < 0) { r += (int64_t)T; --q; }
    if (r < 0) { r += (int64_t)T; --q; }
    if (r >= (int64_t)T) { r -= (int64_t)T; ++q; }
    if (r >= (int64_t)T) { r -= (int64_t)T; ++q; }
    *q_out = q;
    *r_out = (uint64_t)r;
}

struct Dl1Params {
    double scale;                        // 2^(DL1_F + 53 - ex): product -> fixed point
    double inv_t;                        // 1 / T
    uint64_t T;                          // thr in fixed point
    uint64_t thr_bits;                   // bit pattern of thr: a product is in [0, thr) iff its pattern is below this, as unsigned
    double tol_a, tol_b;                 // the fragile margin of a decision at tick i: max(tol_a, (i + 1) tol_b), fixed point
};

template <bool AF64, int DL1_ITEMS, int DL1_THREADS>
__global__ __launch_bounds__(DL1_THREADS) void k_dl1(const double *__restrict__ price, const void *__restrict__ amount, int64_t n,
                                                     Dl1Params P, unsigned long long *__restrict__ dA, unsigned long long *__restrict__ dB,
                                                     int64_t *__restrict__ out, int64_t *__restrict__ carry_k, int64_t cap,
                                                     int64_t *__restrict__ last_m, unsigned long long *__restrict__ n_frag,
                                                     int *__restrict__ flags, unsigned long long *__restrict__ ticket)
{
    __shared__ Dl1U96 s_wave[DL1_THREADS / 64];
    __shared__ uint64_t s_base[2];
    constexpr int DL1_TILE = DL1_THREADS * DL1_ITEMS;
    const int tid = threadIdx.x, lane = fmk_lane(), w = tid >> 6;
    // The tile is a TICKET, not blockIdx.x (round 6): a tile waits for earlier tiles only, and a ticket holder's predecessors have all
    // started -- forward progress no longer rests on workgroups being dispatched in index order, which HIP does not promise.  One
    // atomic round trip in front of the loads: 6.00 -> 6.06 ms per 1e9 ticks.  (The bounded spin stays as the backstop.)
    __shared__ unsigned long long s_ticket;
    if (tid == 0) s_ticket = atomicAdd(ticket, 1ULL);
    __syncthreads();
    const int64_t tile = (int64_t)s_ticket;
    const int64_t groups = ((int64_t)gridDim.x + DL1_W1 - 1) / DL1_W1;
#ifdef DL1_TIMING
    const unsigned long long tq_start = wall_clock64();
#endif
    const int64_t t0 = tile * DL1_TILE, j0 = t0 + (int64_t)tid * DL1_ITEMS;
    // ---- the thread's eight consecutive ticks, 16 bytes per load (the pattern of dl_thread_G)
    double d[DL1_ITEMS];
    if (j0 + DL1_ITEMS <= n && ((uintptr_t)price & 15) == 0 && ((uintptr_t)amount & 15) == 0) {
        double pp[DL1_ITEMS], aa[DL1_ITEMS];
        const double2 *qp = (const double2 *)(price + j0);
#pragma unroll
        for (int k = 0; k < DL1_ITEMS / 2; ++k) { const double2 v = qp[k]; pp[2 * k] = v.x; pp[2 * k + 1] = v.y; }
        if constexpr (AF64) {
            const double2 *qa = (const double2 *)((const double *)amount + j0);
#pragma unroll
            for (int k = 0; k < DL1_ITEMS / 2; ++k) { const double2 v = qa[k]; aa[2 * k] = v.x; aa[2 * k + 1] = v.y; }
        } else {
            const float4 *qa = (const float4 *)((const float *)amount + j0);
#pragma unroll
            for (int k = 0; k < DL1_ITEMS / 4; ++k) {
                const float4 v = qa[k];
                aa[4 * k] = (double)v.x; aa[4 * k + 1] = (double)v.y; aa[4 * k + 2] = (double)v.z; aa[4 * k + 3] = (double)v.w;
            }
        }
#pragma unroll
        for (int k = 0; k < DL1_ITEMS; ++k) d[k] = pp[k] * aa[k];       // rounded once, like prices[i] * volumes[i] (logic.py:143)
    } else {
#pragma unroll
        for (int k = 0; k < DL1_ITEMS; ++k) d[k] = j0 + k < n ? price[j0 + k] * fmk_amt<AF64>(amount, j0 + k) : 0.0;
    }
    uint64_t x[DL1_ITEMS];
    Dl1U96 mine; mine.lo = 0; mine.hi = 0;                            // (sixteen products below 2^61 each: 65 bits)
    bool odd = false;
#pragma unroll
    for (int k = 0; k < DL1_ITEMS; ++k) {
        odd |= (uint64_t)__double_as_longlong(d[k]) >= P.thr_bits;   // negative, NaN, inf or >= thr (and -0.0: the older kernels sort it out)
        x[k] = (uint64_t)(d[k] * P.scale);
        const uint64_t t = mine.lo + x[k];
        mine.hi += t < mine.lo ? 1u : 0u;
        mine.lo = t;
    }
    if (__ballot(odd) != 0 && lane == 0) atomicOr(flags, DL1_FLAG_RANGE);
    // ---- prefix of the thread totals inside the tile: 96-bit integers on the DPP data path
    const Dl1U96 inc = dl1_wave_iscan(mine);
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    if (w == 0) {
        Dl1U96 tot = s_wave[0];
#pragma unroll
        for (int q = 1; q < DL1_THREADS / 64; ++q) tot = dl1_add(tot, s_wave[q]);
        dl1_u128 base = 0;
        if (lane == 0) dl1_publish(dA + 2 * tile, dl1_wide(tot), 1);
        if (tile > 0) {
            bool gave_up;
#ifdef DL1_TIMING
            const unsigned long long tq0 = wall_clock64();
#endif
            int npolls = 0;
            base = dl1_lookback(dA, dB, tile, groups, flags, &gave_up, &npolls);
#ifdef DL1_TIMING
            if (lane == 0) { unsigned long long *tt = dB + 2 * DL1_W1 * groups; tt[4 * tile] = tq_start - ((unsigned long long)npolls << 56); tt[4 * tile + 1] = tq0; tt[4 * tile + 2] = wall_clock64(); }
#endif
            if (lane == 0) dl1_publish(dB + 2 * dl1_bslot(tile, groups), base, gave_up ? 3 : 2);     // (tag 3: the chain stops here)
        } else if (lane == 0) dl1_publish(dB + 2 * dl1_bslot(tile, groups), 0, 2);
        if (lane == 0) { s_base[0] = (uint64_t)base; s_base[1] = (uint64_t)(base >> 64); }
    }
    __syncthreads();
    Dl1U96 ex = inc;                                                  // exclusive prefix of this thread inside its wave ...
    {
        const uint64_t lo = ex.lo - mine.lo;
        ex.hi = ex.hi - mine.hi - (lo > ex.lo ? 1u : 0u);
        ex.lo = lo;
    }
    for (int q = 0; q < w; ++q) ex = dl1_add(ex, s_wave[q]);         // ... inside its tile
    const dl1_u128 pre = (((dl1_u128)s_base[1] << 64) | s_base[0]) + dl1_wide(ex);
    int64_t m0;
    uint64_t r0;
    dl1_divmod(pre, P.T, P.inv_t, &m0, &r0);
    // ---- the reference's loop in exact arithmetic: r += d; if r >= thr: close, r -= thr
    //      a decision is fragile when the sum lies within the reference's rounding drift of thr: |r - T| <= tol
    const double told = fmax(P.tol_a, (double)(j0 + DL1_ITEMS) * P.tol_b);
    const uint64_t tol = (uint64_t)told, tol2 = 2 * tol;
    uint64_t r = r0;
    int closes = 0, frag = 0;
    const bool full = t0 + DL1_TILE <= n;
    if (full) {
#pragma unroll
        for (int k = 0; k < DL1_ITEMS; ++k) {
            r += x[k];
            const uint64_t t = r - P.T;
            frag += (t + tol <= tol2) ? 1 : 0;
            const bool c = r >= P.T;
            r = c ? t : r;
            closes += c ? 1 : 0;
        }
        if (j0 == 0) frag -= (x[0] - P.T + tol <= tol2) ? 1 : 0;      // tick 0 cannot close (logic.py:140-141): not a decision
    } else {
#pragma unroll
        for (int k = 0; k < DL1_ITEMS; ++k) {
            r += x[k];
            const uint64_t t = r - P.T;
            frag += (j0 + k < n && j0 + k > 0 && t + tol <= tol2) ? 1 : 0;
            const bool c = r >= P.T;
            r = c ? t : r;
            closes += c ? 1 : 0;
        }
    }
    if (closes) {                                                     // one thread in ~100: replay its eight ticks and write the closes
        uint64_t r2 = r0;
        int64_t m2 = m0;
#pragma unroll
        for (int k = 0; k < DL1_ITEMS; ++k) {                          // (unrolled: x[] must stay in registers)
            r2 += x[k];
            if (r2 >= P.T) {
                r2 -= P.T;
                ++m2;
                if (m2 < cap) {
                    out[m2] = j0 + k;
                    carry_k[m2] = (int64_t)((r2 + (1ULL << (DL1_F - 1))) >> DL1_F);     // the exact-arithmetic carry in units of ulp(thr)
                }
            }
        }
    }
    if (j0 <= n - 1 && n - 1 < j0 + DL1_ITEMS) *last_m = m0 + closes;   // closes of the whole stream = M_{n-1}
    if (j0 == 0 && cap > 0) { out[0] = 0; carry_k[0] = 0; }             // logic.py:138
#ifdef DL1_TIMING
    if (tid == 0) { unsigned long long *tt = dB + 2 * DL1_W1 * groups; tt[4 * tile + 3] = wall_clock64(); }
#endif
    if (__ballot(frag != 0) != 0) {
        const int f = fmk_dpp_reduce(frag, 0, FmkOpAdd());
        if (lane == 0 && f) atomicAdd(n_frag, (unsigned long long)f);
    }
}

// the sum of a strided sample of the products (for the capacity of the close buffers before the first call on a stream)
template <bool AF64>
__global__ __launch_bounds__(256) void k_dl1_sample(const double *__restrict__ price, const void *__restrict__ amount, int64_t n,
                                                    int64_t stride, int64_t m, double *__restrict__ sum)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double v = 0.0;
    if (g < m) {
        const int64_t i = g * stride;
        if (i < n) {
            v = price[i] * fmk_amt<AF64>(amount, i);
            if (!(v >= 0.0) || v > 1.7e308) v = 0.0;
        }
    }
    v = fmk_dpp_reduce(v, 0.0, FmkOpAdd());
    if (fmk_lane() == 0 && v > 0.0) atomicAdd(sum, v);
}
