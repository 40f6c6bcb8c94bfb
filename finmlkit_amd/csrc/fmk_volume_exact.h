// fmk_volume_exact.h -- _volume_bar_indexer (finmlkit/bar/logic.py:87-115), the EXACT-SUM tier of the jump tables (round 4).
// Included by fmk_volume.hip after its table kernels (k_vol_level_up4, k_vol_descend4) and VolCache.
//
// Why a second level 0.  k_vol_level0 builds, for EVERY tick of a 2048-tick block, the chain link nxt(j), a fragile byte and the pair
// (E, C) -- 13 B/tick written, the block's 2 x 2048 ticks read, i.e. 21 B/tick of HBM traffic for a 4 B/tick column -- and it has
// to carry a tie zone: its prefix sums are not the reference's sequential sums, so a decision within rounding noise of the
// threshold is replayed.  Two observations remove most of that:
//
//  (1) The float64 sum of float32 trade sizes is EXACT on every tape of sane dynamic range: if all amounts of a window are
//      multiples of q = 2^k (k = the lowest set mantissa bit seen) and every sum that is ever formed stays below 2^53 q, every
//      partial sum is a representable multiple of q -- in ANY order.  The reference's `cum` never exceeds thr + max amount, the
//      block's prefixes never exceed the block total.  Each workgroup certifies its own window (one min / max reduction beside the
//      prefix scan); under the certificate the decision `prefix[m] - prefix[j+1] >= thr` IS the reference's `cum >= thr`: no tie
//      zone, no fragile classes, no replay, no verification pass.  A window that does not certify (float64 sizes with full
//      mantissas, sizes spanning more than ~2^29) raises a flag and the call takes the older tier.
//  (2) The chain enters a block within its first W ticks (W >= the longest bar), so only W of a block's S entry ticks need a
//      table row.  With S = 4096 and W = 1024 or 2048 the tables are 2 or 4 B/tick instead of 8, the look-ahead that is read twice
//      is W / S of the block instead of all of it, and the links of all ticks leave the kernel as 16-bit offsets (2 B/tick) for the
//      emit pass, which then is one thread per block hopping through them.
//
// Traffic: 4 (1 + W/S) B/tick read, 2 + 8 W/S B/tick written: 9 (W = 1024) or 12 (W = 2048) B/tick against 21.
#pragma once
#include "fmk_dpp.h"

#define VX_NONE 0xFFFFu

template <bool AF64, int S, int W, int THREADS, bool PAD>
__global__ __launch_bounds__(THREADS) void k_vx_level0(const void *__restrict__ amount, int64_t n, double thr,
                                                       uint16_t *__restrict__ nxt16, uint32_t *__restrict__ EC0,
                                                       uint32_t *__restrict__ root,
                                                       int *__restrict__ status)
{
    constexpr int T = S + W;                          // ticks whose prefix sums the block needs
    constexpr int PER = T / THREADS;                  // consecutive ticks per thread in the prefix phase
    constexpr int EPT = S / THREADS;                  // consecutive ticks per thread in the link phase
    constexpr int NW = THREADS / 64;
    static_assert(T % THREADS == 0 && S % THREADS == 0 && PER % (AF64 ? 2 : 4) == 0, "tile shape");
    static_assert(W <= 0xFFFE, "16-bit link offsets");
    // The prefix table is padded: the prefix phase writes element 8 t + k from lane t -- a stride of 16 banks, four bank pairs for 32
    // lanes, an 8-way conflict on every one of its eight stores, and with them 54 % of the kernel's LDS cycles were conflict cycles
    // (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/r05_cfg3.txt; the kernel's time did not move when a fifth of its VALU
    // instructions went away: it waits for LDS).  PAD: one slot per 8 (the 4096-tick class, which has the room); otherwise one per
    // 32 -- 3 % more LDS, still four workgroups per CU -- which makes lane t's slot 8 t + k + t / 4: 32 distinct bank pairs.
#define VXP(i) Lp[PAD ? (i) + ((i) >> 3) : (i) + ((i) >> 5)]
    constexpr int LPN = PAD ? T + 1 + (T + 1) / 8 + 1 : T + 2 + (T + 2) / 32 + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char vx_smem[];
    double *Lp = (double *)vx_smem;
    uint16_t *nx = (uint16_t *)(vx_smem + (size_t)LPN * 8);
    double *wtot = (double *)(nx + S);                 // [NW] wave totals, [NW] wave maxima, [NW] wave minima
    // a window elsewhere already failed its certificate (or met a bad amount, or a bar beyond W ticks): the call is going to another
    // class or tier anyway
    // (asked by the whole workgroup at once: waves that saw the flag at different moments must not part at a barrier)
    if (__syncthreads_or(__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (VOL_ST_INEXACT | VOL_ST_BAD | VOL_ST_OVERFLOW))) return;
    const int64_t bs = (int64_t)blockIdx.x * S;
    const int tid = threadIdx.x, lane = fmk_lane(), w = tid >> 6;
    const int64_t remain = n - bs;
    const int mmax = (int)(remain < T ? remain : T);   // Lp[0..mmax] are valid
    // ---- the thread's PER consecutive amounts, 16 bytes per load
    double loc[PER];
    double run = 0.0, amax = 0.0, amin = 1.7e308;     // amin: the smallest POSITIVE amount
    int qexp = 4096;                                   // float64 sizes: the lowest set bit of any amount, as a power of two
    bool bad = false;
    {
        const int64_t jb = bs + (int64_t)tid * PER;
        double vv[PER];
        bool fast32 = false;
        if (jb + PER <= n && ((uintptr_t)amount & 15) == 0) {
            if constexpr (AF64) {
                const double2 *q = (const double2 *)((const double *)amount + jb);
#pragma unroll
                for (int k = 0; k < PER / 2; ++k) { const double2 t2 = q[k]; vv[2 * k] = t2.x; vv[2 * k + 1] = t2.y; }
            } else {
                // float32 sizes: the largest, the smallest positive and "any negative / NaN / inf" from the BIT PATTERNS -- non-negative
                // floats order like unsigned integers, everything else lies above 0x7F7FFFFF -- three 32-bit instructions per amount
                // where the float64 compares took eight (the kernel is bound by VALU issue).  (-0.0 counts as negative here: the
                // call then takes the serial walk, which gives the same closes.)
                const float4 *q = (const float4 *)((const float *)amount + jb);
                uint32_t bmax = 0, bmin1 = 0xFFFFFFFFu;
#pragma unroll
                for (int k = 0; k < PER / 4; ++k) {
                    const float4 t4 = q[k];
                    vv[4 * k] = (double)t4.x; vv[4 * k + 1] = (double)t4.y; vv[4 * k + 2] = (double)t4.z; vv[4 * k + 3] = (double)t4.w;
                    const uint32_t u0 = __float_as_uint(t4.x), u1 = __float_as_uint(t4.y), u2 = __float_as_uint(t4.z), u3 = __float_as_uint(t4.w);
                    bmax = max(max(bmax, u0), max(u1, max(u2, u3)));
                    bmin1 = min(min(bmin1, u0 - 1u), min(u1 - 1u, min(u2 - 1u, u3 - 1u)));   // (0 - 1 wraps to the top: zeros do not count)
                }
                fast32 = true;
                bad = bmax > 0x7F7FFFFFu;
                amax = bad ? 0.0 : (double)__uint_as_float(bmax);
                amin = bmin1 == 0xFFFFFFFFu || bad ? 1.7e308 : (double)__uint_as_float(bmin1 + 1u);
            }
        } else {
#pragma unroll
            for (int k = 0; k < PER; ++k) vv[k] = jb + k < n ? fmk_amt<AF64>(amount, jb + k) : 0.0;
        }
        // (ticks beyond n are 0.0: neutral for all four)
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const double v = vv[k];
            if (!fast32) {
                bad |= !(v >= 0.0) || v > 1.7e308;
                amax = fmax(amax, v);
                amin = fmin(amin, v > 0.0 ? v : 1.7e308);
            }
            if constexpr (AF64) {
                // the certificate's quantum for float64 sizes: q = 2^(lowest set bit of any amount) -- with q = ulp(smallest amount), as for
                // float32, no float64 window could ever certify (2^52 ulp(a) <= a), while float64 COPIES of float32 / dyadic sizes do
                const uint64_t bits = (uint64_t)__double_as_longlong(v);
                const uint32_t lo32 = (uint32_t)bits, hi32 = (uint32_t)(bits >> 32);
                const int tz = lo32 ? __builtin_ctz(lo32) : 32 + __builtin_ctz((hi32 & 0xFFFFFu) | 0x100000u);
                const int eq = (int)((hi32 >> 20) & 0x7FFu) - 1075 + tz;
                qexp = v > 0.0 && eq < qexp ? eq : qexp;
            }
            run += v;
            loc[k] = run;
        }
    }
    // DPP scans / reductions (fmk_dpp.h): no LDS round trips
    const double inc = fmk_dpp_iscan(run, 0.0, FmkOpAdd());
    amax = fmk_dpp_reduce(amax, 0.0, FmkOpMax());
    amin = fmk_dpp_reduce(amin, 1.7e308, FmkOpMin());
    if constexpr (AF64) qexp = fmk_dpp_reduce(qexp, 4096, FmkOpMin());
    if (lane == 63) wtot[w] = inc;
    if constexpr (AF64) amin = (double)qexp;           // (float64 sizes: from here on `amin` carries the quantum's exponent)
    if (lane == 0) { wtot[NW + w] = amax; wtot[2 * NW + w] = amin; }
    const bool any_bad = __syncthreads_or(bad ? 1 : 0) != 0;
    double pre = fmk_dpp_shift_up1(inc, 0.0);          // exclusive prefix of the thread totals inside the wave
    double total = 0.0;
    {
        double wp = 0.0;
        for (int q = 0; q < NW; ++q) {
            if (q == w) pre += wp;
            wp += wtot[q];
            amax = fmax(amax, wtot[NW + q]);
            amin = fmin(amin, wtot[2 * NW + q]);               // (float64 sizes: the waves' quantum exponents)
        }
        total = wp;
    }
    if (any_bad) { if (tid == 0) atomicOr(status, VOL_ST_BAD); return; }
    // ---- the certificate of this window.  Every amount is a multiple of q = ulp(smallest positive amount) -- of its own ulp, which is
    //      a power-of-two multiple of q -- (float32 sizes: ulp = 2^(e - 23), float64: 2^(e - 52)); block total and thr + max amount
    //      below 2^53 q.  (A first version took q from the lowest SET mantissa bit of every amount -- more tapes certify, e.g. decimal
    //      lots next to one tiny trade do not here -- at fifteen 64-bit instructions per amount in a kernel that is bound by VALU issue;
    //      float64 sizes take it from the lowest set bit all the same: with their own ulp nothing would certify.)
    if (AF64 ? amin < 4096.0 : amin < 1.7e308) {
        int qe;                                                                          // q = 2^qe
        if constexpr (AF64) qe = (int)amin;
        else qe = (int)((__double_as_longlong(amin) >> 52) & 0x7ff) - 1023 - 23;         // amin >= 2^e (float32 subnormals arrive normal)
        const double lim = ldexp(1.0, qe + 52);                                          // 2^52 q
        const bool exact = qe > -1040 && total < 2.0 * lim && thr < lim && amax < lim;
        if (!exact) { if (tid == 0) vol_flag(status, VOL_ST_INEXACT); return; }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) VXP(tid * PER + k + 1) = pre + loc[k];
    if (tid == 0) VXP(0) = 0.0;
    __syncthreads();
    // ---- nxt(j) for the block's S ticks: smallest m > i + 1 with Lp[m] - Lp[i + 1] >= thr (an exact difference, an exact compare).
    //      A thread owns EPT consecutive ticks; nxt is non-decreasing, so their answers lie in a narrow run of prefixes.  The kernel
    //      is bound by DEPENDENT LDS round trips (first version: a 12-step bisection, then a read-compare-advance walk per tick and one
    //      hop at a time per entry row -- ~50 round trips per thread, 7.1 ms per 1e9 ticks at 16 waves per CU, no faster than the
    //      level 0 it replaces although it moves half the bytes).  Hence: the first tick by an 8-ary search (8 independent reads per
    //      round: 4 rounds for 2048), then the run of prefixes behind its answer is fetched in batches of 8 independent reads and the
    //      EPT ticks are resolved against the batch in registers (a merge of two sorted runs).
    bool ovf = false;
    unsigned minoff = VX_NONE;                                           // the shortest link of this thread's ticks
    {
        const int i0 = tid * EPT;
        // -- first tick of the thread: 8-ary search in (i0 + 1, min(i0 + 1 + W, mmax)]
        int m0 = -1;                                                     // answer for tick i0 (index into Lp), -1: none
        if (i0 < remain) {
            const double base = VXP(i0 + 1);
            int lo = i0 + 2, hi = i0 + 1 + W;
            if (hi > mmax) hi = mmax;
            if (lo <= hi && VXP(hi) - base >= thr) {
                while (lo < hi) {                                        // (an 8-ary search with 8 independent reads per round was
                    const int mid = (lo + hi) >> 1;                      //  tried: 4 round trips instead of 11, but ~290 instead of ~90
                    if (VXP(mid) - base >= thr) hi = mid; else lo = mid + 1;   // instructions, and the kernel is bound by VALU issue)
                }
                m0 = lo;
            } else if (i0 + 1 + W <= mmax) ovf = true;                   // no close within W ticks although data remains
        }
        nx[i0] = (uint16_t)(m0 >= 0 ? (unsigned)(m0 - 1 - i0) : VX_NONE);
        if (m0 >= 0) minoff = (unsigned)(m0 - 1 - i0);
        // -- the other EPT - 1 ticks walk forward from the previous answer (nxt is non-decreasing)
        int cur = m0;
#pragma unroll 1
        for (int q = 1; q < EPT; ++q) {
            const int i = i0 + q;
            unsigned off = VX_NONE;
            if (i < remain) {
                int hi = i + 1 + W;
                if (hi > mmax) hi = mmax;
                const double base = VXP(i + 1);
                int lo = cur > i + 2 ? cur : i + 2;
                if (cur >= 0 && lo <= hi && VXP(hi) - base >= thr) {
                    while (lo < hi && VXP(lo) - base < thr) ++lo;
                    cur = lo;
                    off = (unsigned)(lo - 1 - i);
                } else {
                    if (i + 1 + W <= mmax) ovf = true;
                    cur = -1;
                }
            }
            nx[i] = (uint16_t)off;
            minoff = off < minoff ? off : minoff;
        }
    }
    if (__ballot(ovf) != 0 && lane == 0) vol_flag(status, VOL_ST_OVERFLOW);
    // first bar (block 0): tick 0 is counted but cannot close -> first j >= 1 with P_j >= thr (cum = volumes[0], logic.py:107)
    if (blockIdx.x == 0 && tid == 0) {
        uint32_t r = VOL_END;
        int lo = 2, hi = mmax < W ? mmax : W;           // keeps the root inside the first W ticks
        if (lo <= hi && VXP(hi) >= thr) {
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (VXP(mid) >= thr) hi = mid; else lo = mid + 1;
            }
            r = (uint32_t)(lo - 1);
        } else if (W <= mmax) {
            vol_flag(status, VOL_ST_OVERFLOW);
        }
        *root = r;
    }
    // (are all bars of this block at least THREADS ticks long?  then the entry rows come from a backward sweep, see below)
    const bool long_bars = __syncthreads_and(minoff >= (unsigned)THREADS) != 0;
    // ---- the links leave as 16-bit offsets (coalesced), the entry rows by following them through LDS
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int i = q * THREADS + tid;
        if (bs + i < n) nxt16[bs + i] = nx[i];
    }
    if (long_bars) {
        // Backward sweep (round 5).  (exit, count) of EVERY tick of the block, a chunk of THREADS ticks at a time from the block's end:
        // tick j's chain continues at t = j + off >= j + THREADS, i.e. in a chunk that is already done, so one LDS read gives its
        // (exit, count) and adding one node gives j's -- S ticks, one dependent round trip each, no divergence; the walk below costs
        // W rows x ~S / L hops per row (1.8x as many LDS round trips at W = 1536, L = 865, and lanes that finish at different hops).
        // The packed words overlay the prefix table, which is dead by now.
        uint32_t *ec = (uint32_t *)vx_smem;
#pragma unroll 1
        for (int ch = EPT - 1; ch >= 0; --ch) {
            const int j = ch * THREADS + tid;
            uint32_t wv = 0xFFFFu;                                     // beyond the data: no node, no exit
            if (j < remain) {
                const unsigned o = nx[j];
                if (o == VX_NONE) wv = (1u << 16) | 0xFFFFu;           // the chain ends in this block
                else {
                    const int t = j + (int)o;
                    wv = t >= S ? ((1u << 16) | (uint32_t)(t - S)) : ec[t] + (1u << 16);
                }
            }
            ec[j] = wv;
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < (W + THREADS - 1) / THREADS; ++r) {
            const int i = tid + r * THREADS;
            if (i < W) EC0[(int64_t)blockIdx.x * W + i] = ec[i];
        }
    } else {   // W / THREADS rows per thread, walked TOGETHER: one LDS round trip advances all of them by a hop
        constexpr int ROWS = (W + THREADS - 1) / THREADS;
        int e[ROWS];
        uint32_t C[ROWS];
        bool open[ROWS];
        uint32_t E[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int i = tid + r * THREADS;
            e[r] = i; E[r] = VOL_END;
            open[r] = i < W && i < remain;
            C[r] = open[r] ? 1u : 0u;
        }
        for (;;) {
            bool any = false;
            unsigned o[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) o[r] = open[r] ? (unsigned)nx[e[r]] : VX_NONE;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                if (!open[r]) continue;
                if (o[r] == VX_NONE) { open[r] = false; continue; }
                e[r] += (int)o[r];
                if (e[r] >= S) { E[r] = (uint32_t)(bs + e[r]); open[r] = false; continue; }
                ++C[r];
                any = true;
            }
            if (!any) break;
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int i = tid + r * THREADS;
            // one word per row: the count and where the chain leaves the block, as an offset into the next block's first W ticks
            // (two 32-bit arrays made this table 4.8 B/tick written here and read again by the first level-up: 1.2 of 6.5 ms)
            if (i < W) EC0[(int64_t)blockIdx.x * W + i] = (C[r] << 16) | (E[r] == VOL_END ? 0xFFFFu : E[r] - (uint32_t)(bs + S));
        }
    }
#undef VXP
}

// every level-0 block writes its chain nodes into their final slots: one thread per block hopping through the 16-bit links
__global__ __launch_bounds__(256) void k_vx_emit(int64_t S, const uint32_t *__restrict__ ent0, const int64_t *__restrict__ off0,
                                                 int64_t nblk0, const uint16_t *__restrict__ nxt16, int64_t *__restrict__ out,
                                                 int64_t cap)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0 && cap > 0) out[0] = 0;                              // logic.py:104
    if (b >= nblk0) return;
    uint32_t j = ent0[b];
    int64_t o = off0[b];
    const uint64_t bend = (uint64_t)(b + 1) * (uint64_t)S;
    while (j != VOL_END && (uint64_t)j < bend) {
        if (o < cap) out[o] = (int64_t)j;
        ++o;
        const unsigned d = nxt16[j];
        j = d == VX_NONE ? VOL_END : j + d;
    }
}

// k_vol_level_up4 / k_vol_descend4 for tables of W rows per block, W any number (theirs is a power of two): level q + 1 composes
// `radix` blocks of level q over the W entry ticks of the first of them
// a row of the packed level-0 table: count, and the chain's exit as an absolute tick (VOL_END: none)
__device__ __forceinline__ void vx_unpack0(uint32_t w, int64_t block, int64_t S, uint32_t *e, uint32_t *c)
{
    *c = w >> 16;
    *e = (w & 0xFFFFu) == 0xFFFFu ? VOL_END : (uint32_t)((block + 1) * S + (w & 0xFFFFu));
}

template <bool PACKED>
__global__ __launch_bounds__(256) void k_vx_level_up(int W, const uint32_t *__restrict__ Ep, const uint32_t *__restrict__ Cp,
                                                     int64_t nblk_prev, int64_t span_prev, uint32_t *__restrict__ Ek,
                                                     uint32_t *__restrict__ Ck, int64_t nblk, int *__restrict__ status, int radix)
{
    const int64_t b = blockIdx.x;
    const int i = (int)(blockIdx.y * blockDim.x + threadIdx.x);
    if (i >= W || b >= nblk) return;
    const int64_t c0 = (int64_t)radix * b;
    uint32_t x, c;
    if constexpr (PACKED) vx_unpack0(Ep[c0 * W + i], c0, span_prev, &x, &c);
    else { x = Ep[c0 * W + i]; c = Cp[c0 * W + i]; }
    for (int j = 1; j < radix; ++j) {
        const int64_t child = c0 + j;
        if (x == VOL_END || child >= nblk_prev) break;
        const int64_t i2 = (int64_t)x - child * span_prev;           // the chain enters the next child in its first W ticks
        if (i2 < 0 || i2 >= W) { vol_flag(status, VOL_ST_OVERFLOW); x = VOL_END; break; }
        if constexpr (PACKED) {
            uint32_t c2;
            vx_unpack0(Ep[child * W + i2], child, span_prev, &x, &c2);
            c += c2;
        } else {
            c += Cp[child * W + i2];
            x = Ep[child * W + i2];
        }
    }
    Ek[b * W + i] = x;
    Ck[b * W + i] = c;
}

template <bool PACKED>
__global__ __launch_bounds__(256) void k_vx_descend(int W, const uint32_t *__restrict__ ent_k, const int64_t *__restrict__ off_k,
                                                    int64_t nblk_k, int64_t span_prev, const uint32_t *__restrict__ Ep,
                                                    const uint32_t *__restrict__ Cp, int64_t nblk_prev,
                                                    uint32_t *__restrict__ ent_p, int64_t *__restrict__ off_p, int radix)
{
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
            const int64_t i = (int64_t)e - (child - 1) * span_prev;  // the entry lies in the previous child's first W ticks
            if (i >= 0 && i < W) {
                if constexpr (PACKED) {
                    uint32_t c2;
                    vx_unpack0(Ep[(child - 1) * W + i], child - 1, span_prev, &e, &c2);
                    o += c2;
                } else {
                    o += Cp[(child - 1) * W + i];
                    e = Ep[(child - 1) * W + i];
                }
            }
        }
        ent_p[child] = e;
        off_p[child] = o;
    }
}

// returns FMK_OK, 1 (a bar beyond W ticks: the next class), 4 (a window that does not certify: the older tier), 2 (negative / NaN
// amounts) or an error
template <bool AF64, int S, int W, int THREADS, bool PAD>
static int vx_run(fmk_ctx *ctx, const void *a, int64_t n, double thr, VolCache &c)
{
    const int64_t nblk0 = fmk_ceil_div(n, S);
    const int RAD = VOL_RADIX;
    int64_t nblk[64], spanq[64];
    int K = 0;
    nblk[0] = nblk0;
    spanq[0] = S;
    while (nblk[K] > 1) { nblk[K + 1] = (nblk[K] + RAD - 1) / RAD; spanq[K + 1] = spanq[K] * RAD; ++K; }
    size_t tbl = 0, ents = 0;
    for (int k = 0; k <= K; ++k) { tbl += (size_t)nblk[k] * W; ents += (size_t)nblk[k]; }
    const size_t nxt_bytes = (((size_t)nblk0 * S * 2) + 255) & ~(size_t)255;
    const size_t bytes = nxt_bytes + 2 * tbl * 4 + ents * (4 + 8) + 256;
    if (c.work_bytes < bytes) {
        if (c.work) FMK_HIP(ctx, hipFree(c.work));
        c.work = nullptr; c.work_bytes = 0;
        if (hipMalloc(&c.work, bytes) != hipSuccess) { (void)hipGetLastError(); c.work = nullptr; return 1; }
        c.work_bytes = bytes;
    }
    uint16_t *nxt16 = (uint16_t *)c.work;
    uint32_t *Eall = (uint32_t *)((char *)c.work + nxt_bytes);
    uint32_t *Call = Eall + tbl;
    int64_t *offall = (int64_t *)(Call + tbl);
    uint32_t *entall = (uint32_t *)(offall + ents);
    uint32_t *E[64], *C[64], *ent[64];
    int64_t *off[64];
    {
        size_t to = 0, eo = 0;
        for (int k = 0; k <= K; ++k) {
            E[k] = Eall + to; C[k] = Call + to; to += (size_t)nblk[k] * W;
            ent[k] = entall + eo; off[k] = offall + eo; eo += (size_t)nblk[k];
        }
    }
    int *d_status = (int *)(ctx->d_mail + 32);
    uint32_t *d_root = (uint32_t *)(ctx->d_mail + 33);
    FMK_HIP(ctx, hipMemsetAsync(ctx->d_mail + 32, 0, 32, ctx->stream));
    {
        constexpr int T = S + W;
        constexpr size_t lds = (size_t)(PAD ? T + 1 + (T + 1) / 8 + 1 : T + 2 + (T + 2) / 32 + 1) * 8 + (size_t)S * 2 + (size_t)(THREADS / 64) * 24 + 64;
        if (lds > 64 * 1024)
            FMK_HIP(ctx, hipFuncSetAttribute((const void *)k_vx_level0<AF64, S, W, THREADS, PAD>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_vx_level0<AF64, S, W, THREADS, PAD><<<(unsigned)nblk0, THREADS, lds, ctx->stream>>>(a, n, thr, nxt16, E[0], d_root, d_status);
    }
    FMK_LAUNCH_CHECK(ctx);
    // the status is known after level 0; the level-ups are cheap (N W / (S RAD) entries and less) and run regardless
    for (int k = 1; k <= K; ++k) {
        if (k == 1)
            k_vx_level_up<true><<<dim3((unsigned)nblk[k], (unsigned)fmk_ceil_div(W, 256)), 256, 0, ctx->stream>>>(
                W, E[0], nullptr, nblk[0], spanq[0], E[k], C[k], nblk[k], d_status, RAD);
        else
            k_vx_level_up<false><<<dim3((unsigned)nblk[k], (unsigned)fmk_ceil_div(W, 256)), 256, 0, ctx->stream>>>(
                W, E[k - 1], C[k - 1], nblk[k - 1], spanq[k - 1], E[k], C[k], nblk[k], d_status, RAD);
        FMK_LAUNCH_CHECK(ctx);
    }
    FMK_HIP(ctx, hipMemcpyAsync(ctx->h_mail, ctx->d_mail + 32, 24, hipMemcpyDeviceToHost, ctx->stream));
    FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int status = (int)(ctx->h_mail[0] & 0xFFFFFFFF);
    const uint32_t root = (uint32_t)(ctx->h_mail[1] & 0xFFFFFFFFu);
    if (status & VOL_ST_BAD) return 2;
    if (status & VOL_ST_INEXACT) return 4;                          // no class of this tier will serve the stream
    if (status & VOL_ST_OVERFLOW) return 1;
    int64_t closes = 0;
    if (root != VOL_END) {
        if ((int64_t)root >= W) return 1;
        uint32_t cnt = 0;
        FMK_HIP(ctx, hipMemcpyAsync(&cnt, (K == 0 ? E[0] : C[K]) + root, 4, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        closes = K == 0 ? cnt >> 16 : cnt;                           // (a single block: the packed level-0 row itself)
    }
    c.count = closes + 1;
    if (c.dbuf && c.cap < c.count) { FMK_HIP(ctx, hipFree(c.dbuf)); c.dbuf = nullptr; }
    if (!c.dbuf) { FMK_HIP(ctx, hipMalloc((void **)&c.dbuf, (size_t)c.count * 8)); c.cap = c.count; }
    const int64_t one = 1;
    FMK_HIP(ctx, hipMemcpyAsync(ent[K], &root, 4, hipMemcpyHostToDevice, ctx->stream));
    FMK_HIP(ctx, hipMemcpyAsync(off[K], &one, 8, hipMemcpyHostToDevice, ctx->stream));
    for (int k = K; k >= 1; --k) {
        if (k == 1)
            k_vx_descend<true><<<(unsigned)fmk_ceil_div(nblk[k], 256), 256, 0, ctx->stream>>>(
                W, ent[k], off[k], nblk[k], spanq[0], E[0], nullptr, nblk[0], ent[0], off[0], RAD);
        else
            k_vx_descend<false><<<(unsigned)fmk_ceil_div(nblk[k], 256), 256, 0, ctx->stream>>>(
                W, ent[k], off[k], nblk[k], spanq[k - 1], E[k - 1], C[k - 1], nblk[k - 1], ent[k - 1], off[k - 1], RAD);
        FMK_LAUNCH_CHECK(ctx);
    }
    k_vx_emit<<<(unsigned)fmk_ceil_div(nblk0, 256), 256, 0, ctx->stream>>>(S, ent[0], off[0], nblk0, nxt16, c.dbuf, c.cap);
    FMK_LAUNCH_CHECK(ctx);
    c.unc = 0;                                                      // exact sums: every decision is the reference's
    return FMK_OK;
}
