// fmk_preprocess.hip -- the two sequential loops of TradesData(preprocess=True) on gfx950:
//   merge_split_trades      finmlkit/bar/utils.py:263-329   (stream compaction with float32 run sums)
//   comp_trade_side_vector  finmlkit/bar/utils.py:26-46     (tick rule = "last significant price move" scan)
//
// merge_split_trades.  A trade joins the current merged trade iff its timestamp and maker flag equal the HEAD's and
// |price - head price| < 1e-8.  Timestamp / flag equality with the head is equivalent to equality with the previous
// trade (all members of a run share them), so the stream splits into coarse runs of equal (timestamp, flag); only the
// price test is head-relative, and it is evaluated sequentially inside each coarse run by the thread that owns the
// run's first trade (runs are a handful of trades on real data).  Heads -> exclusive scan -> output slots; the thread
// of every head then adds its followers' float32 amounts in order (the reference rounds to float32 on every add).
//
// comp_trade_side_vector.  side[i] = sign(p[i] - p[i-1]) if |dp| > 1e-12 else side[i-1], side[0] = 0: the last
// significant move at or before i -- a scan under "right operand wins unless it is 0" (tile reduce, one-block scan
// of the tile aggregates, apply).
#include "fmk_common.h"
#include "fmk_scan.h"

// ---------------------------------------------------------------------------------------
// merge_split_trades
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ bool mg_coarse_head(const int64_t *ts, const uint8_t *ibm, int64_t i)
{
    if (i == 0) return true;
    if (ts[i] != ts[i - 1]) return true;
    return ibm && (ibm[i] != 0) != (ibm[i - 1] != 0);
}

// flags[i] = 1 iff trade i starts a merged trade (one byte per trade; every entry is written exactly once: a coarse head by its
// own thread, its followers by that thread too)
__global__ __launch_bounds__(256) void k_merge_flags(const int64_t *__restrict__ ts, const double *__restrict__ price,
                                                     const uint8_t *__restrict__ ibm, int64_t n,
                                                     uint8_t *__restrict__ flags)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !mg_coarse_head(ts, ibm, i)) return;
    double head = price[i];
    flags[i] = 1;
    for (int64_t j = i + 1; j < n && !mg_coarse_head(ts, ibm, j); ++j) {
        const bool same = fabs(price[j] - head) < 1e-8;          // utils.py:300-301 (vs the HEAD's price)
        flags[j] = same ? 0 : 1;
        if (!same) head = price[j];
    }
}

// Output slots.  The first version kept the flags as int64 and ran the generic device scan over them: 8 B/tick written, 24 B/tick
// through the scan, 16 B/tick read back by the emit pass -- more than the 42 B/tick of columns in and out (17.7 ms per 1e9
// ticks).  Now: one byte per flag, heads counted per 2048-tick tile (k_merge_count), the tile counts scanned by one block
// (fmk_scan.h), and the emit pass ranks the heads inside its tile itself.
#define MG_ITEMS 8
#define MG_TILE (256 * MG_ITEMS)
__device__ __forceinline__ uint64_t mg_load_flags(const uint8_t *__restrict__ flags, int64_t i0, int64_t n)
{
    if (i0 + MG_ITEMS <= n) return *(const uint64_t *)(flags + i0);          // (the flag array is 256-byte aligned, i0 a multiple of 8)
    uint64_t v = 0;
    for (int k = 0; k < MG_ITEMS; ++k)
        if (i0 + k < n) v |= (uint64_t)flags[i0 + k] << (8 * k);
    return v;
}
__global__ __launch_bounds__(256) void k_merge_count(const uint8_t *__restrict__ flags, int64_t n, int64_t *__restrict__ tile_cnt)
{
    __shared__ int sw[4];
    const int64_t i0 = (int64_t)blockIdx.x * MG_TILE + (int64_t)threadIdx.x * MG_ITEMS;
    int c = i0 < n ? __builtin_popcountll(mg_load_flags(flags, i0, n)) : 0;      // flags are 0 / 1 bytes
    c = fmk_wave_sum(c);
    if (fmk_lane() == 0) sw[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = sw[0] + sw[1] + sw[2] + sw[3];
}

// tile_off = exclusive scan of the tile counts.  Thread t takes the tile's trades t, t + 256, ...: coalesced column loads and
// (nearly) coalesced stores; a head's slot = tile offset + heads in the rows before + heads before it in its row (ballots).
// (With eight CONSECUTIVE trades per thread the column loads sat 64 bytes apart across the lanes: 45 ms per 1e9 ticks.)
__global__ __launch_bounds__(256) void k_merge_emit(const int64_t *__restrict__ ts, const double *__restrict__ price,
                                                    const float *__restrict__ amount, const uint8_t *__restrict__ ibm,
                                                    int64_t n, const uint8_t *__restrict__ flags,
                                                    const int64_t *__restrict__ tile_off,
                                                    int64_t *__restrict__ o_ts, double *__restrict__ o_price,
                                                    float *__restrict__ o_amount, int8_t *__restrict__ o_side)
{
    __shared__ int sw[MG_ITEMS][4];
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * MG_TILE;
    bool head[MG_ITEMS];
    int before[MG_ITEMS];                                         // heads before mine in my wave's 64 trades of the row
#pragma unroll
    for (int k = 0; k < MG_ITEMS; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        head[k] = i < n && flags[i] != 0;
        const unsigned long long b = __builtin_amdgcn_ballot_w64(head[k]);
        before[k] = __builtin_popcountll(b & ((1ULL << lane) - 1));
        if (lane == 0) sw[k][w] = __builtin_popcountll(b);
    }
    __syncthreads();
    int64_t row_off = tile_off[blockIdx.x];
#pragma unroll
    for (int k = 0; k < MG_ITEMS; ++k) {
        int in_row = 0;
        for (int q = 0; q < w; ++q) in_row += sw[k][q];
        if (head[k]) {
            const int64_t i = base + k * 256 + threadIdx.x, p = row_off + in_row + before[k];
            float acc = amount[i];
            for (int64_t j = i + 1; j < n && flags[j] == 0; ++j) acc += amount[j];     // utils.py:307 (float32 +=)
            o_ts[p] = ts[i];
            o_price[p] = price[i];
            o_amount[p] = acc;
            if (o_side) o_side[p] = ibm[i] ? -1 : 1;              // utils.py:296, 316
        }
        row_off += sw[k][0] + sw[k][1] + sw[k][2] + sw[k][3];
    }
}

extern "C" int fmk_merge_split_trades_dev(fmk_ctx *ctx, const int64_t *d_ts, const double *d_price,
                                          const float *d_amount, const uint8_t *d_is_buyer_maker, int64_t n,
                                          int64_t *d_out_ts, double *d_out_price, float *d_out_amount,
                                          int8_t *d_out_side, int64_t capacity, int64_t *n_merged)
{
    if (n <= 0) return fmk_set_error(ctx, FMK_E_ARG, "merge_split_trades: empty input");
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    // scratch: [tile offsets + total | flags[n]]
    const int64_t tiles = fmk_ceil_div(n, MG_TILE);
    const size_t off_bytes = (((size_t)tiles + 1) * 8 + 255) & ~(size_t)255;
    void *scr;
    FMK_TRY(fmk_scratch(ctx, off_bytes + (size_t)n + 256, &scr));
    int64_t *tile_off = (int64_t *)scr;
    uint8_t *flags = (uint8_t *)scr + off_bytes;
    k_merge_flags<<<(unsigned)fmk_ceil_div(n, 256), 256, 0, ctx->stream>>>(d_ts, d_price, d_is_buyer_maker, n, flags);
    FMK_LAUNCH_CHECK(ctx);
    k_merge_count<<<(unsigned)tiles, 256, 0, ctx->stream>>>(flags, n, tile_off);
    FMK_LAUNCH_CHECK(ctx);
    k_scan_tile_scan<<<1, 1024, 0, ctx->stream>>>(tile_off, tiles, tile_off + tiles);
    FMK_LAUNCH_CHECK(ctx);
    FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[0], tile_off + tiles, 8, hipMemcpyDeviceToHost, ctx->stream));
    FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t m = ctx->h_mail[0];
    if (n_merged) *n_merged = m;
    if (!d_out_ts) return FMK_OK;                                 // phase 1: count only
    if (capacity < m)
        return fmk_set_error(ctx, FMK_E_CAPACITY, "merge_split_trades: %lld merged trades, capacity %lld", (long long)m,
                             (long long)capacity);
    if (d_out_side && !d_is_buyer_maker)
        return fmk_set_error(ctx, FMK_E_ARG, "merge_split_trades: side output needs is_buyer_maker");
    k_merge_emit<<<(unsigned)tiles, 256, 0, ctx->stream>>>(d_ts, d_price, d_amount, d_is_buyer_maker, n, flags, tile_off, d_out_ts,
                                                          d_out_price, d_out_amount, d_out_side);
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}

// ---------------------------------------------------------------------------------------
// comp_trade_side_vector
// ---------------------------------------------------------------------------------------
#define SV_THREADS 256
#define SV_ITEMS 8
#define SV_TILE (SV_THREADS * SV_ITEMS)

__device__ __forceinline__ int sv_move(const double *price, int64_t i)
{
    if (i <= 0) return 0;
    const double dp = price[i] - price[i - 1];
    if (!(fabs(dp) > 1e-12)) return 0;                            // utils.py:20-23 (NaN: not significant)
    return dp > 0.0 ? 1 : -1;
}

// last non-zero move of the block's threads in thread order: exclusive value for this thread + block aggregate
__device__ __forceinline__ int sv_block_exclusive(int mine, int *lds /*[4]*/, int *block_total)
{
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
    int inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d && inc == 0) inc = o;
    }
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    int pre = 0;
    for (int k = 0; k < w; ++k) pre = lds[k] != 0 ? lds[k] : pre;
    int tot = 0;
    for (int k = 0; k < 4; ++k) tot = lds[k] != 0 ? lds[k] : tot;
    *block_total = tot;
    int prev = __shfl_up(inc, 1, 64);
    if (lane == 0) prev = 0;
    __syncthreads();
    return prev != 0 ? prev : pre;
}

__global__ __launch_bounds__(SV_THREADS) void k_side_tile(const double *__restrict__ price, int64_t n,
                                                          int8_t *__restrict__ tile_last)
{
    // the tile's LAST non-zero move = the move at the largest index that has one: an argmax, so the loads are coalesced
    // (thread t: ticks t, t + 256, ...); 8 consecutive ticks per thread made every load instruction touch 64 lines
    __shared__ int lds_m[4];
    __shared__ int lds_i[4];
    const int64_t t0 = (int64_t)blockIdx.x * SV_TILE + (int64_t)threadIdx.x;
    int last = 0, last_i = -1;
#pragma unroll
    for (int k = 0; k < SV_ITEMS; ++k) {
        const int64_t i = t0 + (int64_t)k * SV_THREADS;
        if (i < n) { const int m = sv_move(price, i); if (m != 0) { last = m; last_i = k * SV_THREADS + (int)threadIdx.x; } }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const int oi = __shfl_xor(last_i, d, 64), om = __shfl_xor(last, d, 64);
        if (oi > last_i) { last_i = oi; last = om; }
    }
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
    if (lane == 0) { lds_m[w] = last; lds_i[w] = last_i; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int bm = lds_m[0], bi = lds_i[0];
        for (int q = 1; q < 4; ++q) if (lds_i[q] > bi) { bi = lds_i[q]; bm = lds_m[q]; }
        tile_last[blockIdx.x] = (int8_t)(bi >= 0 ? bm : 0);
    }
}

// one block: tile_last[t] := last non-zero aggregate among tiles < t (exclusive), in place
__global__ __launch_bounds__(1024) void k_side_scan_tiles(int8_t *tile_last, int64_t tiles)
{
    __shared__ int ws[16];
    __shared__ int run;
    if (threadIdx.x == 0) run = 0;
    __syncthreads();
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
    for (int64_t b = 0; b < tiles; b += 1024) {
        const int64_t i = b + threadIdx.x;
        const int v = i < tiles ? (int)tile_last[i] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            if (lane >= d && inc == 0) inc = o;
        }
        if (lane == 63) ws[w] = inc;
        __syncthreads();
        int pre = run;
        for (int k = 0; k < w; ++k) pre = ws[k] != 0 ? ws[k] : pre;
        int prev = __shfl_up(inc, 1, 64);
        if (lane == 0) prev = 0;
        if (i < tiles) tile_last[i] = (int8_t)(prev != 0 ? prev : pre);
        __syncthreads();
        if (threadIdx.x == 1023) run = inc != 0 ? inc : pre;
        __syncthreads();
    }
}

__global__ __launch_bounds__(SV_THREADS) void k_side_apply(const double *__restrict__ price, int64_t n,
                                                           const int8_t *__restrict__ tile_pre, int8_t *__restrict__ out)
{
    __shared__ int lds[4];
    // moves computed from COALESCED loads (thread t: ticks t, t + 256, ...), one byte each into an LDS tile, then every thread
    // reads the 8 consecutive moves it owns as one 8-byte word: the fill below runs in tick order
    __shared__ __attribute__((aligned(8))) signed char s_mv[SV_TILE];
    const int64_t t0 = (int64_t)blockIdx.x * SV_TILE + (int64_t)threadIdx.x;
#pragma unroll
    for (int k = 0; k < SV_ITEMS; ++k) {
        const int64_t i = t0 + (int64_t)k * SV_THREADS;
        s_mv[k * SV_THREADS + (int)threadIdx.x] = (signed char)(i < n ? sv_move(price, i) : 0);
    }
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * SV_TILE + (int64_t)threadIdx.x * SV_ITEMS;
    const unsigned long long packed = *(const unsigned long long *)(s_mv + (int)threadIdx.x * SV_ITEMS);
    int mv[SV_ITEMS];
    int last = 0;
#pragma unroll
    for (int k = 0; k < SV_ITEMS; ++k) {
        mv[k] = (int)(signed char)(packed >> (8 * k));
        last = mv[k] != 0 ? mv[k] : last;
    }
    int tot;
    int cur = sv_block_exclusive(last, lds, &tot);
    if (cur == 0) cur = (int)tile_pre[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SV_ITEMS; ++k) {
        const int64_t i = i0 + k;
        cur = mv[k] != 0 ? mv[k] : cur;
        if (i < n) out[i] = (int8_t)cur;
    }
}

extern "C" int fmk_comp_trade_side_vector_dev(fmk_ctx *ctx, const double *d_price, int64_t n, int8_t *d_out)
{
    if (n <= 0) return fmk_set_error(ctx, FMK_E_ARG, "comp_trade_side_vector: empty input");
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t tiles = fmk_ceil_div(n, SV_TILE);
    void *scr;
    FMK_TRY(fmk_scratch(ctx, (size_t)tiles + 64, &scr));
    int8_t *tile_last = (int8_t *)scr;
    k_side_tile<<<(unsigned)tiles, SV_THREADS, 0, ctx->stream>>>(d_price, n, tile_last);
    FMK_LAUNCH_CHECK(ctx);
    k_side_scan_tiles<<<1, 1024, 0, ctx->stream>>>(tile_last, tiles);
    FMK_LAUNCH_CHECK(ctx);
    k_side_apply<<<(unsigned)tiles, SV_THREADS, 0, ctx->stream>>>(d_price, n, tile_last, d_out);
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}
