// fmk_threshold.hip -- _volume_bar_indexer / _dollar_bar_indexer (finmlkit/bar/logic.py:87-149).
//
// Both are sequential recurrences with a data-dependent reset (volume: cum = 0) or carry
// (dollar: cum -= threshold) -- the position of every close depends on all earlier ones.
// ROUND-1 IMPLEMENTATION (correct first, see DESIGN.md "threshold bars" for the parallel
// hierarchical-jump design that replaces it): ONE wave walks the stream in 1024-tick chunks.
//   * lane l owns 16 consecutive ticks: sequential float64 prefix inside the lane, 6-step wave
//     scan of the lane totals -> cum at every tick of the chunk in ~30 wave instructions.
//   * the first tick with cum >= threshold is found with one ballot; reset/carry is applied by
//     re-basing the chunk's prefix values (no re-scan), so a chunk costs O(1 + closes in chunk).
//   * the next chunk's loads are issued before the current one is processed (software prefetch:
//     the single wave has no neighbours to hide HBM latency behind).
// The in-chunk sums are tree+lane ordered instead of strictly sequential; a close decision whose
// margin |cum - threshold| is below 1e-11*threshold is counted in n_uncertified (0 = every
// decision is provably the reference's; amounts that sum exactly, like the synthetic dyadic
// stream, are always certified).
// A parallel reduction first bounds the number of closes (sum/threshold + 2) so the result can be
// written in one pass; the two-phase C-ABI call (count, then fill) reuses the cached result.
#include <math.h>

#include "fmk_common.h"

#define TH_ITEMS 16
#define TH_CHUNK (64 * TH_ITEMS)

template <bool AF64, bool DOLLAR>
__device__ __forceinline__ void th_load(const double *price, const void *amount, int64_t n, int64_t i0,
                                        double (&d)[TH_ITEMS])
{
#pragma unroll
    for (int k = 0; k < TH_ITEMS; ++k) {
        const int64_t i = i0 + k;
        double v = 0.0;
        if (i < n) {
            v = fmk_amt<AF64>(amount, i);
            if constexpr (DOLLAR) v = price[i] * v;      // rounded product, like the reference
        }
        d[k] = v;
    }
}

template <bool AF64, bool DOLLAR>
__global__ __launch_bounds__(64) void k_threshold_index(const double *__restrict__ price,
                                                        const void *__restrict__ amount, int64_t n, double thr,
                                                        int64_t *__restrict__ out, int64_t cap,
                                                        int64_t *__restrict__ result /*[2]: count, uncertified*/)
{
    const int lane = fmk_lane();
    int64_t m = 0, unc = 0;
    if (lane == 0 && cap > 0) out[0] = 0;       // logic.py:104 / 138
    m = 1;
    double cin = 0.0;                            // cum carried into the chunk
    const double tol = 1e-11 * fabs(thr);
    double cerr = 0.0;                           // bound on what the chunked sums have put into cin since it was last reset
    double cur[TH_ITEMS], nxt[TH_ITEMS];
    th_load<AF64, DOLLAR>(price, amount, n, (int64_t)lane * TH_ITEMS, cur);
    for (int64_t base = 0; base < n; base += TH_CHUNK) {
        if (base + TH_CHUNK < n)
            th_load<AF64, DOLLAR>(price, amount, n, base + TH_CHUNK + (int64_t)lane * TH_ITEMS, nxt);
        // inclusive prefix inside the lane, exclusive prefix across lanes
        double s[TH_ITEMS];
        s[0] = cur[0];
#pragma unroll
        for (int k = 1; k < TH_ITEMS; ++k) s[k] = s[k - 1] + cur[k];
        const double inc = fmk_wave_iscan(s[TH_ITEMS - 1]);
        // the exclusive prefix comes from the previous lane, not from inc - s[last]: with a NaN (or inf) increment in this
        // lane the subtraction would poison the lane's EARLIER ticks too, which the reference still closes on
        // (tools/fuzz_volume.py seed 2 case 185)
        double ex = __shfl_up(inc, 1, 64);
        if (lane == 0) ex = 0.0;
        const double chunk_total = __shfl(inc, 63, 64);
        // The chunk's sums are differences of chunk-wide prefixes: next to an increment of 1e12 every later tick of the chunk
        // carries an error of ulp(1e12), whatever the threshold (tools/fuzz_volume.py seed 97015 case 2356: decisions off by
        // a tick and not reported).  ~24 roundings at the magnitude of the chunk's absolute sum bound it; a NaN increment
        // is left out of that sum (the ticks before it still decide), an infinite one makes every decision of the chunk fragile.
        double la = 0.0;
#pragma unroll
        for (int k = 0; k < TH_ITEMS; ++k) { const double v = fabs(cur[k]); la += v == v ? v : 0.0; }
        const double cherr = 7.2e-15 * (fmk_wave_sum(la) + (cin == cin ? fabs(cin) : 0.0));
        const double btol = tol + cerr + cherr;
        bool closed_here = false;
        const int64_t i0 = base + (int64_t)lane * TH_ITEMS;
        double off = 0.0;                         // chunk-prefix value at the last close in this chunk
        int64_t last_close = base == 0 ? 0 : base - 1;   // ticks <= last_close cannot close (tick 0 never does)
        for (;;) {
            int kk = TH_ITEMS;
            double ck = 0.0;
            bool frag = false;
#pragma unroll
            for (int k = TH_ITEMS - 1; k >= 0; --k) {
                const int64_t i = i0 + k;
                const double c = cin + ((ex + s[k]) - off);
                const bool live = i > last_close && i < n;
                if (live && c >= thr) { kk = k; ck = c; }
                frag |= live && fabs(c - thr) <= btol;
            }
            const uint64_t hit = __ballot(kk < TH_ITEMS);
            if (hit == 0) {
                unc += __popcll(__ballot(frag));
                break;
            }
            const int l0 = __ffsll((unsigned long long)hit) - 1;
            const int k0 = __shfl(kk, l0, 64);
            const double c0 = __shfl(ck, l0, 64);
            const int64_t ic = base + (int64_t)l0 * TH_ITEMS + k0;
            // fragile decisions up to and including the close
            unc += __popcll(__ballot(frag && i0 <= ic)) > 0 ? 1 : 0;
            if (lane == 0 && m < cap) out[m] = ic;
            ++m;
            // re-base: prefix value of the chunk at the close
            double pk = 0.0;
#pragma unroll
            for (int k = 0; k < TH_ITEMS; ++k) pk = (k == k0) ? ex + s[k] : pk;
            off = __shfl(pk, l0, 64);
            cin = DOLLAR ? c0 - thr : 0.0;        // logic.py:147 carry / logic.py:113 reset
            last_close = ic;
            closed_here = true;
        }
        if (!(cherr < INFINITY)) ++unc;           // an infinite increment: inf - inf in the re-based sums
        cin = cin + (chunk_total - off);
        cerr = (DOLLAR || !closed_here) ? cerr + cherr : cherr;     // (the dollar carry is never reset)
#pragma unroll
        for (int k = 0; k < TH_ITEMS; ++k) cur[k] = nxt[k];
    }
    if (lane == 0) { result[0] = m; result[1] = unc; }
}

// The reference's loop, operation for operation (logic.py:107-113 / 141-147): cum = d_0; for i >= 1: cum += d_i; if
// cum >= thr: close at i and cum = 0 (volume) or cum -= thr (dollar).  The float64 running sum carries its own rounding
// drift -- for dollar bars it is never reset -- so when an exactly computed sum lands within that drift of the threshold
// (n_uncertified of the parallel algorithms), only the same sequence of float64 operations reproduces the reference's
// decision.  One wave: the 64 increments of a group are computed by the lanes (rounded product first, like the reference),
// lane 0 adds them in tick order from LDS; the next group's loads are in flight meanwhile.  Tens of ns per tick: the
// fallback of the NumPy-facing functions, not a path for 1e9 resident ticks (fmk_ctx_set_fast_threshold).
template <bool AF64, bool DOLLAR>
__global__ __launch_bounds__(64) void k_threshold_exact(const double *__restrict__ price, const void *__restrict__ amount,
                                                        int64_t n, double thr, int64_t *__restrict__ out, int64_t cap,
                                                        int64_t *__restrict__ result /*[2]: count, uncertified = 0*/)
{
    __shared__ __attribute__((aligned(16))) double s_d[64];
    const int lane = fmk_lane();
    int64_t m = 1;
    if (lane == 0 && cap > 0) out[0] = 0;        // logic.py:104 / 138
    double cum = 0.0;
    auto inc = [&](int64_t i) {
        double v = 0.0;
        if (i < n) {
            v = fmk_amt<AF64>(amount, i);
            if constexpr (DOLLAR) v = price[i] * v;
        }
        return v;
    };
    double cur = inc(lane);
    for (int64_t base = 0; base < n; base += 64) {
        const double nxt = inc(base + 64 + lane);
        s_d[lane] = cur;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            // The chain cum -> add -> compare -> select is the whole cost (one lane, nothing to overlap it with), so it is kept
            // to exactly those operations: 8 increments per LDS read, no branch per tick (a close is a select and a bit in
            // `hits`), the closes are written after the group.  65 ns per tick with a load, a branch and a store inside the
            // loop (tools/serialbench.py).
            const int lim = (int)(n - base < 64 ? n - base : 64);
            uint64_t hits = 0;
            if (base > 0 && lim == 64) {
                // full groups: nothing but the chain.  The close bits go through the ballot into scalar registers (lane 0 is
                // the only active lane), off the vector pipe and off the dependency chain; conditions on the tick index
                // (first tick, end of the stream) would put a VALU -> SALU -> VALU round trip through VCC into every step
#pragma unroll
                for (int q8 = 0; q8 < 64; q8 += 8) {
                    double d[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) d[k] = s_d[q8 + k];
                    // speculate that none of these 8 ticks closes (bars are mostly longer than that): then the reference's
                    // operations are just the 8 additions, in order -- one dependent instruction per tick, the compares hang
                    // off the chain.  Only a batch with a close pays for the add -> compare -> select chain per tick.
                    double sp[8];
                    sp[0] = cum + d[0];
#pragma unroll
                    for (int k = 1; k < 8; ++k) sp[k] = sp[k - 1] + d[k];
                    bool any = false;
#pragma unroll
                    for (int k = 0; k < 8; ++k) any |= sp[k] >= thr;
                    if (__builtin_amdgcn_ballot_w64(any) == 0) { cum = sp[7]; continue; }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        cum += d[k];
                        const bool hit = cum >= thr;
                        const double after = DOLLAR ? cum - thr : 0.0;     // logic.py:147 carry / logic.py:113 reset
                        hits |= (uint64_t)(__builtin_amdgcn_ballot_w64(hit) & 1) << (q8 + k);
                        cum = hit ? after : cum;
                    }
                }
            } else {
                for (int q = 0; q < lim; ++q) {
                    if (base + q == 0) { cum = s_d[0]; continue; }          // cum = volumes[0] (logic.py:107); tick 0 cannot close
                    cum += s_d[q];
                    if (cum >= thr) {
                        hits |= (uint64_t)1 << q;
                        cum = DOLLAR ? cum - thr : 0.0;
                    }
                }
            }
            while (hits) {
                const int q = __ffsll((unsigned long long)hits) - 1;
                hits &= hits - 1;
                if (m < cap) out[m] = base + q;
                ++m;
            }
        }
        __builtin_amdgcn_wave_barrier();
        cur = nxt;
    }
    if (lane == 0) { result[0] = m; result[1] = 0; }
}

// sum of the (dollar) volumes: bounds the number of closes
template <bool AF64, bool DOLLAR>
__global__ __launch_bounds__(256) void k_threshold_total(const double *__restrict__ price,
                                                         const void *__restrict__ amount, int64_t n, double *acc)
{
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double v = fabs(fmk_amt<AF64>(amount, i));
        if constexpr (DOLLAR) v = fabs(price[i]) * v;
        s += v;
    }
    s = fmk_wave_sum(s);
    if (fmk_lane() == 0) atomicAdd(acc, s);
}

struct ThCache {
    const void *amount;
    const double *price;
    int64_t n;
    double thr;
    int kind, is_f64;
    int64_t count, unc;
    int64_t *dbuf;
    int64_t cap;
    fmk_ctx *ctx;
    int exact;
};
static ThCache &th_cache(fmk_ctx *ctx)       // one per context (slot 2), created on first use
{
    if (!ctx->idx_cache[2]) { ThCache *c = new ThCache(); c->kind = -1; ctx->idx_cache[2] = c; }
    return *(ThCache *)ctx->idx_cache[2];
}

void fmk_threshold_trim(fmk_ctx *ctx)
{
    ThCache *c = (ThCache *)ctx->idx_cache[2];
    if (!c) return;
    if (c->dbuf) (void)hipFree(c->dbuf);
    delete c;
    ctx->idx_cache[2] = nullptr;
}

template <bool DOLLAR>
static int th_run(fmk_ctx *ctx, const double *d_price, const void *d_amount, int is_f64, int64_t n, double thr,
                  int64_t *d_close_idx, int64_t capacity, int64_t *n_idx, int64_t *n_unc, int exact)
{
    if (n <= 0) return fmk_set_error(ctx, FMK_E_ARG, "threshold indexer: empty input");
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    ThCache &c = th_cache(ctx);
    const bool hit = c.ctx == ctx && !ctx->idx_stale[2] && c.amount == d_amount && c.price == d_price && c.n == n && c.thr == thr &&
                     c.kind == (int)DOLLAR && c.is_f64 == is_f64 && c.dbuf && c.exact == exact;
    if (!(hit && d_close_idx)) {
        // bound the number of closes
        double *d_acc = (double *)ctx->d_mail;
        FMK_HIP(ctx, hipMemsetAsync(d_acc, 0, 8, ctx->stream));
        const unsigned blocks = (unsigned)(fmk_ceil_div(n, 256 * 16) < ctx->n_cu * 16 ? fmk_ceil_div(n, 256 * 16)
                                                                                      : ctx->n_cu * 16);
        if (is_f64) k_threshold_total<true, DOLLAR><<<blocks, 256, 0, ctx->stream>>>(d_price, d_amount, n, d_acc);
        else k_threshold_total<false, DOLLAR><<<blocks, 256, 0, ctx->stream>>>(d_price, d_amount, n, d_acc);
        FMK_LAUNCH_CHECK(ctx);
        double total = 0.0;
        FMK_HIP(ctx, hipMemcpyAsync(&total, d_acc, 8, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        int64_t bound = n;
        if (thr > 0 && total / thr + 2.0 < (double)n) bound = (int64_t)(total / thr) + 2;
        if (bound < 1) bound = 1;
        if (c.dbuf && c.cap < bound) { FMK_HIP(ctx, hipFree(c.dbuf)); c.dbuf = nullptr; }
        if (!c.dbuf) {
            FMK_HIP(ctx, hipMalloc((void **)&c.dbuf, (size_t)bound * 8));
            c.cap = bound;
        }
        int64_t *d_res = ctx->d_mail + 8;
        if (exact) {
            if (is_f64)
                k_threshold_exact<true, DOLLAR><<<1, 64, 0, ctx->stream>>>(d_price, d_amount, n, thr, c.dbuf, c.cap, d_res);
            else
                k_threshold_exact<false, DOLLAR><<<1, 64, 0, ctx->stream>>>(d_price, d_amount, n, thr, c.dbuf, c.cap, d_res);
        } else if (is_f64)
            k_threshold_index<true, DOLLAR><<<1, 64, 0, ctx->stream>>>(d_price, d_amount, n, thr, c.dbuf, c.cap, d_res);
        else
            k_threshold_index<false, DOLLAR><<<1, 64, 0, ctx->stream>>>(d_price, d_amount, n, thr, c.dbuf, c.cap, d_res);
        FMK_LAUNCH_CHECK(ctx);
        FMK_HIP(ctx, hipMemcpyAsync(ctx->h_mail, d_res, 16, hipMemcpyDeviceToHost, ctx->stream));
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        c.ctx = ctx; c.amount = d_amount; c.price = d_price; c.n = n; c.thr = thr; c.kind = (int)DOLLAR;
        ctx->idx_key[2][0] = d_amount; ctx->idx_key[2][1] = d_price; ctx->idx_stale[2] = 0;
        c.is_f64 = is_f64; c.count = ctx->h_mail[0]; c.unc = ctx->h_mail[1]; c.exact = exact;
        if (c.count > c.cap) return fmk_set_error(ctx, FMK_E_CAPACITY, "threshold indexer: internal bound exceeded");
    }
    *n_idx = c.count;
    if (n_unc) *n_unc = c.unc;
    if (!d_close_idx) return FMK_OK;
    if (capacity < c.count) return fmk_set_error(ctx, FMK_E_CAPACITY, "threshold indexer: capacity %lld < %lld",
                                                 (long long)capacity, (long long)c.count);
    FMK_HIP(ctx, hipMemcpyAsync(d_close_idx, c.dbuf, (size_t)c.count * 8, hipMemcpyDeviceToDevice, ctx->stream));
    c.ctx = nullptr;    // one-shot cache: the inputs may change behind the same pointers
    return FMK_OK;
}

// Serial walk, shared entry for both bar types (also the fallback of the parallel algorithms for
// inputs outside their domain: thr <= 0, negative increments, bars longer than the jump tables).
int fmk_threshold_serial(fmk_ctx *ctx, int dollar, const double *d_price, const void *d_amount, int is_f64, int64_t n,
                         double thr, int64_t *d_close_idx, int64_t capacity, int64_t *n_idx, int64_t *n_unc)
{
    const int exact = !ctx->fast_threshold;      // default: the reference's float64 loop, operation for operation
    return dollar ? th_run<true>(ctx, d_price, d_amount, is_f64, n, thr, d_close_idx, capacity, n_idx, n_unc, exact)
                  : th_run<false>(ctx, nullptr, d_amount, is_f64, n, thr, d_close_idx, capacity, n_idx, n_unc, exact);
}

// (the extern "C" entry points live in fmk_volume.hip / fmk_dollar.hip)
