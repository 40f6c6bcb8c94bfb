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

#define VX_NONE 0xFFFFu

// lowest set bit of a positive finite double as a power of two (its exponent); INT_MAX for 0
__device__ __forceinline__ int vx_lsb_exp(double v)
{
    if (!(v > 0.0)) return 0x7fffffff;
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const int e = (int)((b >> 52) & 0x7ff);
    if (e == 0) return -100000;                                          // subnormal double: never certifies
    const unsigned long long mant = (b & 0xFFFFFFFFFFFFFull) | (1ull << 52);
    return e - 1075 + (int)__builtin_ctzll(mant);
}

template <bool AF64, int S, int W, int THREADS, bool PAD>
__global__ __launch_bounds__(THREADS) void k_vx_level0(const void *__restrict__ amount, int64_t n, double thr,
                                                       uint16_t *__restrict__ nxt16, uint32_t *__restrict__ E0,
                                                       uint32_t *__restrict__ C0, uint32_t *__restrict__ root,
                                                       int *__restrict__ status)
{
    constexpr int T = S + W;                          // ticks whose prefix sums the block needs
    constexpr int PER = T / THREADS;                  // consecutive ticks per thread in the prefix phase
    constexpr int EPT = S / THREADS;                  // consecutive ticks per thread in the link phase
    constexpr int NW = THREADS / 64;
    static_assert(T % THREADS == 0 && S % THREADS == 0 && PER % (AF64 ? 2 : 4) == 0, "tile shape");
    static_assert(W <= 0xFFFE, "16-bit link offsets");
#define VXP(i) Lp[PAD ? (i) + ((i) >> 3) : (i)]
    constexpr int LPN = PAD ? T + 1 + (T + 1) / 8 + 1 : T + 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char vx_smem[];
    double *Lp = (double *)vx_smem;
    uint16_t *nx = (uint16_t *)(vx_smem + (size_t)LPN * 8);
    double *wtot = (double *)(nx + S);                 // [NW] wave totals, then [NW] wave maxima
    int *wlsb = (int *)(wtot + 2 * NW);
    // a window elsewhere already failed its certificate (or met a bad amount): the call is going to the older tier anyway
    // (asked by the whole workgroup at once: waves that saw the flag at different moments must not part at a barrier)
    if (__syncthreads_or(__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (VOL_ST_INEXACT | VOL_ST_BAD))) return;
    const int64_t bs = (int64_t)blockIdx.x * S;
    const int tid = threadIdx.x, lane = fmk_lane(), w = tid >> 6;
    const int64_t remain = n - bs;
    const int mmax = (int)(remain < T ? remain : T);   // Lp[0..mmax] are valid
    // ---- the thread's PER consecutive amounts, 16 bytes per load
    double loc[PER];
    double run = 0.0, amax = 0.0;
    int lsb = 0x7fffffff;
    bool bad = false;
    {
        const int64_t jb = bs + (int64_t)tid * PER;
        double vv[PER];
        if (jb + PER <= n && ((uintptr_t)amount & 15) == 0) {
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
            if (jb + k < n) {
                bad |= !(v >= 0.0) || v > 1.7e308;
                const int l = vx_lsb_exp(v);
                lsb = l < lsb ? l : lsb;
                amax = fmax(amax, v);
            }
            run += v;
            loc[k] = run;
        }
    }
    const double inc = fmk_wave_iscan(run);
    amax = fmk_wave_max(amax);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int x = __shfl_xor(lsb, o, 64); lsb = x < lsb ? x : lsb; }
    if (lane == 63) wtot[w] = inc;
    if (lane == 0) { wtot[NW + w] = amax; wlsb[w] = lsb; }
    const bool any_bad = __syncthreads_or(bad ? 1 : 0) != 0;
    double pre = __shfl_up(inc, 1, 64);
    if (lane == 0) pre = 0.0;
    double total = 0.0;
    {
        double wp = 0.0;
        for (int q = 0; q < NW; ++q) {
            if (q == w) pre += wp;
            wp += wtot[q];
            amax = fmax(amax, wtot[NW + q]);
            lsb = wlsb[q] < lsb ? wlsb[q] : lsb;
        }
        total = wp;
    }
    if (any_bad) { if (tid == 0) atomicOr(status, VOL_ST_BAD); return; }
    // ---- the certificate of this window: every amount a multiple of q = 2^lsb, block total and thr + max amount below 2^53 q
    if (lsb != 0x7fffffff) {
        const double lim = ldexp(1.0, lsb + 52);                        // 2^52 q: thr and amax each below it -> their sum below 2^53 q
        const bool exact = lsb > -1000 && total < 2.0 * lim && thr < lim && amax < lim;
        if (!exact) { if (tid == 0) vol_flag(status, VOL_ST_INEXACT); return; }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) VXP(tid * PER + k + 1) = pre + loc[k];
    if (tid == 0) VXP(0) = 0.0;
    __syncthreads();
    // ---- nxt(j) for the block's S ticks: smallest m > i + 1 with Lp[m] - Lp[i + 1] >= thr (an exact difference, an exact compare).
    //      A thread owns EPT consecutive ticks: one bisection, then forward walks (nxt is non-decreasing)
    int carry_lo = 0;
    bool ovf = false;
#pragma unroll 1
    for (int q = 0; q < EPT; ++q) {
        const int i = tid * EPT + q;
        unsigned off = VX_NONE;
        if (i < remain) {
            const double base = VXP(i + 1);
            int lo = i + 2, hi = i + 1 + W;
            if (hi > mmax) hi = mmax;
            if (lo <= hi && VXP(hi) - base >= thr) {
                if (q > 0 && carry_lo >= lo) {
                    lo = carry_lo;
                    while (lo < hi && VXP(lo) - base < thr) ++lo;
                } else {
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (VXP(mid) - base >= thr) hi = mid; else lo = mid + 1;
                    }
                }
                carry_lo = lo;
                off = (unsigned)(lo - 1 - i);                            // close tick bs + lo - 1, offset 1 .. W
            } else {
                if (i + 1 + W <= mmax) ovf = true;                       // no close within W ticks although data remains
                carry_lo = 0;
            }
        }
        nx[i] = (uint16_t)off;
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
    __syncthreads();
    // ---- the links leave as 16-bit offsets (coalesced), the entry rows by following them through LDS
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int i = q * THREADS + tid;
        if (bs + i < n) nxt16[bs + i] = nx[i];
    }
    for (int i = tid; i < W; i += THREADS) {
        uint32_t E = VOL_END, C = 0;
        if (i < remain) {
            int e = i;
            C = 1;
            for (;;) {
                const unsigned o = nx[e];
                if (o == VX_NONE) break;
                e += (int)o;
                if (e >= S) { E = (uint32_t)(bs + e); break; }
                ++C;
            }
        }
        E0[(int64_t)blockIdx.x * W + i] = E;
        C0[(int64_t)blockIdx.x * W + i] = C;
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

// returns FMK_OK, 1 (the next tier: a window that does not certify, or a bar beyond W ticks), 2 (negative / NaN amounts) or an error
template <bool AF64, int S, int W, int THREADS, bool PAD>
static int vx_run(fmk_ctx *ctx, const void *a, int64_t n, double thr, VolCache &c)
{
    int LS = 0;
    while ((1 << LS) < W) ++LS;
    const int64_t nblk0 = fmk_ceil_div(n, S);
    static int RAD = 0;
    if (!RAD) { const char *v = getenv("FMK_VOL_RADIX"); RAD = v ? atoi(v) : VOL_RADIX; if (RAD < 2 || RAD > 64) RAD = VOL_RADIX; }
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
        constexpr size_t lds = (size_t)(PAD ? T + 1 + (T + 1) / 8 + 1 : T + 2) * 8 + (size_t)S * 2 + (size_t)(THREADS / 64) * 20 + 64;
        if (lds > 64 * 1024)
            FMK_HIP(ctx, hipFuncSetAttribute((const void *)k_vx_level0<AF64, S, W, THREADS, PAD>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_vx_level0<AF64, S, W, THREADS, PAD><<<(unsigned)nblk0, THREADS, lds, ctx->stream>>>(a, n, thr, nxt16, E[0], C[0], d_root,
                                                                                             d_status);
    }
    FMK_LAUNCH_CHECK(ctx);
    // the status is known after level 0; the level-ups are cheap (N W / (S RAD) entries and less) and run regardless
    for (int k = 1; k <= K; ++k) {
        const int64_t tot = nblk[k] * W;
        k_vol_level_up4<<<(unsigned)fmk_ceil_div(tot, 256), 256, 0, ctx->stream>>>(
            LS, E[k - 1], C[k - 1], nblk[k - 1], spanq[k - 1], E[k], C[k], nblk[k], d_status, RAD);
        FMK_LAUNCH_CHECK(ctx);
    }
    FMK_HIP(ctx, hipMemcpyAsync(ctx->h_mail, ctx->d_mail + 32, 24, hipMemcpyDeviceToHost, ctx->stream));
    FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int status = (int)(ctx->h_mail[0] & 0xFFFFFFFF);
    const uint32_t root = (uint32_t)(ctx->h_mail[1] & 0xFFFFFFFFu);
    if (status & VOL_ST_BAD) return 2;
    if (status & (VOL_ST_OVERFLOW | VOL_ST_INEXACT)) return 1;
    int64_t closes = 0;
    if (root != VOL_END) {
        if ((int64_t)root >= W) return 1;
        uint32_t cnt = 0;
        FMK_HIP(ctx, hipMemcpyAsync(&cnt, C[K] + root, 4, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        closes = cnt;
    }
    c.count = closes + 1;
    if (c.dbuf && c.cap < c.count) { FMK_HIP(ctx, hipFree(c.dbuf)); c.dbuf = nullptr; }
    if (!c.dbuf) { FMK_HIP(ctx, hipMalloc((void **)&c.dbuf, (size_t)c.count * 8)); c.cap = c.count; }
    const int64_t one = 1;
    FMK_HIP(ctx, hipMemcpyAsync(ent[K], &root, 4, hipMemcpyHostToDevice, ctx->stream));
    FMK_HIP(ctx, hipMemcpyAsync(off[K], &one, 8, hipMemcpyHostToDevice, ctx->stream));
    for (int k = K; k >= 1; --k) {
        k_vol_descend4<<<(unsigned)fmk_ceil_div(nblk[k], 256), 256, 0, ctx->stream>>>(
            LS, ent[k], off[k], nblk[k], spanq[k - 1], E[k - 1], C[k - 1], nblk[k - 1], ent[k - 1], off[k - 1], RAD);
        FMK_LAUNCH_CHECK(ctx);
    }
    k_vx_emit<<<(unsigned)fmk_ceil_div(nblk0, 256), 256, 0, ctx->stream>>>(S, ent[0], off[0], nblk0, nxt16, c.dbuf, c.cap);
    FMK_LAUNCH_CHECK(ctx);
    c.unc = 0;                                                      // exact sums: every decision is the reference's
    return FMK_OK;
}
