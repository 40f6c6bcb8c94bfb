// fmk_scan.h -- device-wide exclusive scan of int64 (reduce-then-scan, three launches).
//   k_scan_tile_sums : one 256-thread block per 4096-element tile -> tile total
//   k_scan_tile_scan : ONE 1024-thread block scans the tile totals in place (B/4096 elements)
//   k_scan_apply     : per tile: block scan of the elements + tile offset, in place
// Used for CSR offsets of the footprint output (B elements) -- tiny next to the tick streams.
#pragma once

#include "fmk_common.h"

#define FMK_SCAN_THREADS 256
#define FMK_SCAN_ROWS 16
#define FMK_SCAN_TILE (FMK_SCAN_THREADS * FMK_SCAN_ROWS)

static __global__ __launch_bounds__(FMK_SCAN_THREADS) void k_scan_tile_sums(const int64_t *__restrict__ in, int64_t n,
                                                                     int64_t *__restrict__ tile_sum)
{
    __shared__ int64_t sw[4];
    const int64_t base = (int64_t)blockIdx.x * FMK_SCAN_TILE;
    int64_t s = 0;
#pragma unroll 4
    for (int r = 0; r < FMK_SCAN_ROWS; ++r) {
        int64_t j = base + r * FMK_SCAN_THREADS + threadIdx.x;
        if (j < n) s += in[j];
    }
    s = fmk_wave_sum(s);
    if (fmk_lane() == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = sw[0] + sw[1] + sw[2] + sw[3];
}

// exclusive scan in place; total written to *total
static __global__ __launch_bounds__(1024) void k_scan_tile_scan(int64_t *t, int64_t m, int64_t *total)
{
    __shared__ int64_t ws[16];
    __shared__ int64_t run;
    if (threadIdx.x == 0) run = 0;
    __syncthreads();
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
    for (int64_t b = 0; b < m; b += 1024) {
        int64_t i = b + threadIdx.x;
        int64_t v = i < m ? t[i] : 0;
        int64_t iv = fmk_wave_iscan(v);
        if (lane == 63) ws[w] = iv;
        __syncthreads();
        int64_t o = run;
        for (int k = 0; k < w; ++k) o += ws[k];
        if (i < m) t[i] = o + iv - v;
        __syncthreads();
        if (threadIdx.x == 1023) run = o + iv;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = run;
}

// out[j] = tile_off[tile] + exclusive prefix within the tile; out[n] = total when j == n-1 is seen
static __global__ __launch_bounds__(FMK_SCAN_THREADS) void k_scan_apply(const int64_t *__restrict__ in, int64_t n,
                                                                 const int64_t *__restrict__ tile_off,
                                                                 int64_t *__restrict__ out)
{
    __shared__ int64_t ws[4];
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * FMK_SCAN_TILE;
    int64_t run = tile_off[blockIdx.x];
    for (int r = 0; r < FMK_SCAN_ROWS; ++r) {
        int64_t j = base + r * FMK_SCAN_THREADS + threadIdx.x;
        int64_t v = j < n ? in[j] : 0;
        int64_t iv = fmk_wave_iscan(v);
        if (lane == 63) ws[w] = iv;
        __syncthreads();
        int64_t o = run;
        for (int k = 0; k < w; ++k) o += ws[k];
        run += ws[0] + ws[1] + ws[2] + ws[3];
        if (j < n) out[j] = o + iv - v;
        __syncthreads();
    }
}

// Exclusive scan of d_in[0..n) into d_out[0..n) (may alias); d_out[n] receives the total when
// `write_total_at_n` is set.  Enqueued on the context stream.
static inline int fmk_exclusive_scan_i64(fmk_ctx *ctx, const int64_t *d_in, int64_t *d_out, int64_t n,
                                         bool write_total_at_n)
{
    if (n <= 0) {
        if (write_total_at_n) FMK_HIP(ctx, hipMemsetAsync(d_out, 0, 8, ctx->stream));
        return FMK_OK;
    }
    const int64_t tiles = fmk_ceil_div(n, FMK_SCAN_TILE);
    void *scr;
    FMK_TRY(fmk_scratch(ctx, (size_t)(tiles + 1) * 8, &scr));
    int64_t *tile = (int64_t *)scr;
    k_scan_tile_sums<<<(unsigned)tiles, FMK_SCAN_THREADS, 0, ctx->stream>>>(d_in, n, tile);
    FMK_LAUNCH_CHECK(ctx);
    int64_t *total = write_total_at_n ? d_out + n : tile + tiles;
    k_scan_tile_scan<<<1, 1024, 0, ctx->stream>>>(tile, tiles, total);
    FMK_LAUNCH_CHECK(ctx);
    k_scan_apply<<<(unsigned)tiles, FMK_SCAN_THREADS, 0, ctx->stream>>>(d_in, n, tile, d_out);
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}
