// fmk_synth.hip -- on-device generator of the synthetic tick stream of SURVEY.md 8(d).
//
// Definition (identical to oracle/fmk_oracle.c:orc_synth, which the parity tests use):
//   h_i   = mix64(seed + i)
//   ts_i  = T0 + sum_{k<=i} (1 + h_k mod gap_mod)           exact int64 prefix sum
//   k_i   = K0 + sum_{k<=i} step(h_k >> 62)   step = -1,0,0,+1   integer-grid random walk
//   price = double(k_i) * 0.01 ; amount = float(1 + (h_i>>8 & 4095)) * 2^-10 ; side = +-1 (bit 40)
// Three kernels: per-tile sums -> single-block scan of tile sums -> generate with offsets.
// A shard that starts at first > 0 first reduces the hashes of [0, first) (pure ALU work).
#include "fmk_common.h"

#define SY_T0 1700000000000000000LL
#define SY_K0 1000000LL
#define SY_THREADS 256
#define SY_ROWS 16
#define SY_TILE (SY_THREADS * SY_ROWS)

__device__ __forceinline__ void sy_decode(uint64_t h, uint64_t gap_mod, int64_t &gap, int64_t &step)
{
    gap = 1 + (int64_t)(h % gap_mod);
    unsigned b = (unsigned)(h >> 62);
    step = (int64_t)(b == 3) - (int64_t)(b == 0);
}

// sums of gaps/steps over the index range [lo, hi) accumulated into acc[0], acc[1]
__global__ __launch_bounds__(256) void k_synth_prefix_reduce(uint64_t seed, uint64_t gap_mod, int64_t lo,
                                                             int64_t hi, unsigned long long *acc)
{
    int64_t g = 0, s = 0;
    for (int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t a, b;
        sy_decode(fmk_mix64(seed + (uint64_t)i), gap_mod, a, b);
        g += a;
        s += b;
    }
    g = fmk_wave_sum(g);
    s = fmk_wave_sum(s);
    if (fmk_lane() == 0) {
        atomicAdd(&acc[0], (unsigned long long)g);
        atomicAdd(&acc[1], (unsigned long long)s);
    }
}

__global__ __launch_bounds__(SY_THREADS) void k_synth_tile_sums(uint64_t seed, uint64_t gap_mod, int64_t first,
                                                                int64_t n, int64_t *tile_g, int64_t *tile_s)
{
    __shared__ int64_t sg[4], ss[4];
    int64_t base = (int64_t)blockIdx.x * SY_TILE;
    int64_t g = 0, s = 0;
#pragma unroll 4
    for (int r = 0; r < SY_ROWS; ++r) {
        int64_t j = base + r * SY_THREADS + threadIdx.x;
        if (j < n) {
            int64_t a, b;
            sy_decode(fmk_mix64(seed + (uint64_t)(first + j)), gap_mod, a, b);
            g += a;
            s += b;
        }
    }
    g = fmk_wave_sum(g);
    s = fmk_wave_sum(s);
    if (fmk_lane() == 0) { sg[threadIdx.x >> 6] = g; ss[threadIdx.x >> 6] = s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        tile_g[blockIdx.x] = sg[0] + sg[1] + sg[2] + sg[3];
        tile_s[blockIdx.x] = ss[0] + ss[1] + ss[2] + ss[3];
    }
}

// In-place exclusive scan of two int64 arrays by one 1024-thread block; adds the carry-in
// read from carry[0], carry[1].
__global__ __launch_bounds__(1024) void k_synth_scan_tiles(int64_t *tg, int64_t *ts, int64_t m,
                                                           const unsigned long long *carry)
{
    __shared__ int64_t wg[16], ws[16];
    __shared__ int64_t run_g, run_s;
    if (threadIdx.x == 0) { run_g = (int64_t)carry[0]; run_s = (int64_t)carry[1]; }
    __syncthreads();
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
    for (int64_t b = 0; b < m; b += 1024) {
        int64_t i = b + threadIdx.x;
        int64_t g = i < m ? tg[i] : 0, s = i < m ? ts[i] : 0;
        int64_t ig = fmk_wave_iscan(g), is = fmk_wave_iscan(s);
        if (lane == 63) { wg[w] = ig; ws[w] = is; }
        __syncthreads();
        int64_t og = run_g, os = run_s;
        for (int k = 0; k < w; ++k) { og += wg[k]; os += ws[k]; }
        if (i < m) { tg[i] = og + ig - g; ts[i] = os + is - s; }
        __syncthreads();
        if (threadIdx.x == 1023) { run_g = og + ig; run_s = os + is; }
        __syncthreads();
    }
}

__global__ __launch_bounds__(SY_THREADS) void k_synth_generate(uint64_t seed, uint64_t gap_mod, int64_t first,
                                                               int64_t n, const int64_t *tile_g,
                                                               const int64_t *tile_s, int64_t *ts, double *price,
                                                               float *amount, int8_t *side)
{
    __shared__ int64_t wg[4], ws[4];
    const int lane = fmk_lane(), w = threadIdx.x >> 6;
    int64_t base = (int64_t)blockIdx.x * SY_TILE;
    int64_t run_g = tile_g[blockIdx.x], run_s = tile_s[blockIdx.x];
    for (int r = 0; r < SY_ROWS; ++r) {
        int64_t j = base + r * SY_THREADS + threadIdx.x;
        uint64_t h = fmk_mix64(seed + (uint64_t)(first + j));
        int64_t g = 0, s = 0;
        if (j < n) sy_decode(h, gap_mod, g, s);
        int64_t ig = fmk_wave_iscan(g), is = fmk_wave_iscan(s);
        if (lane == 63) { wg[w] = ig; ws[w] = is; }
        __syncthreads();
        int64_t og = run_g, os = run_s;
        for (int k = 0; k < w; ++k) { og += wg[k]; os += ws[k]; }
        run_g += wg[0] + wg[1] + wg[2] + wg[3];
        run_s += ws[0] + ws[1] + ws[2] + ws[3];
        if (j < n) {
            if (ts) ts[j] = SY_T0 + og + ig;
            if (price) price[j] = (double)(SY_K0 + os + is) * 0.01;
            if (amount) amount[j] = (float)(1 + ((h >> 8) & 4095)) * 0.0009765625f;
            if (side) side[j] = ((h >> 40) & 1) ? 1 : -1;
        }
        __syncthreads();
    }
}

extern "C" int fmk_synth_trades_dev(fmk_ctx *ctx, uint64_t seed, int64_t first, int64_t n, uint64_t gap_mod,
                                    int64_t *d_ts, double *d_price, float *d_amount, int8_t *d_side)
{
    if (first < 0 || n < 0 || gap_mod == 0) return fmk_set_error(ctx, FMK_E_ARG, "synth: bad range/gap_mod");
    if (n == 0) return FMK_OK;
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    int64_t tiles = fmk_ceil_div(n, SY_TILE);
    void *scr;
    FMK_TRY(fmk_scratch(ctx, (size_t)tiles * 16 + 64, &scr));
    unsigned long long *carry = (unsigned long long *)scr;
    int64_t *tg = (int64_t *)((char *)scr + 64);
    int64_t *tsum = tg + tiles;
    FMK_HIP(ctx, hipMemsetAsync(carry, 0, 16, ctx->stream));
    if (first > 0) {
        int64_t blocks = fmk_ceil_div(first, 256 * 64);
        if (blocks > 8192) blocks = 8192;
        k_synth_prefix_reduce<<<(unsigned)blocks, 256, 0, ctx->stream>>>(seed, gap_mod, 0, first, carry);
        FMK_LAUNCH_CHECK(ctx);
    }
    k_synth_tile_sums<<<(unsigned)tiles, SY_THREADS, 0, ctx->stream>>>(seed, gap_mod, first, n, tg, tsum);
    FMK_LAUNCH_CHECK(ctx);
    k_synth_scan_tiles<<<1, 1024, 0, ctx->stream>>>(tg, tsum, tiles, carry);
    FMK_LAUNCH_CHECK(ctx);
    k_synth_generate<<<(unsigned)tiles, SY_THREADS, 0, ctx->stream>>>(seed, gap_mod, first, n, tg, tsum, d_ts,
                                                                       d_price, d_amount, d_side);
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}
