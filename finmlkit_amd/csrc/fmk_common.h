// fmk_common.h -- shared host/device helpers of the gfx950 tick->bar engine.
// gfx950 only: wave = 64 lanes everywhere, no portability shims.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/fmk.h"

#define FMK_WAVE 64
#define FMK_PROFILE_SLOTS 256              // launches fmk_profile_* can time between enable and read

struct fmk_ctx {
    int device;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    char err[512];
    // grow-on-demand device scratch (scan partials, flags)
    void *scratch;
    size_t scratch_bytes;
    // small pinned host mailbox for flags / counters read back from the device
    int64_t *h_mail;   // 64 x int64
    int64_t *d_mail;   // 64 x int64
    int n_cu;
    // per-launch timing of the dominant kernel (fmk_profile_enable)
    int profile_on, profile_n;
    // threshold indexers: 0 (default) = inputs with uncertified decisions are redone by the exact sequential loop;
    // 1 = return the parallel result with its n_uncertified (fmk_ctx_set_fast_threshold)
    int fast_threshold;
    // 1 = calls never wait for the device where they have the choice (fmk_ctx_set_enqueue_only)
    int enqueue_only;
    hipEvent_t kev[FMK_PROFILE_SLOTS][2];
    // stream-ordered caching allocator behind fmk_alloc / fmk_free (fmk_api.hip): freed blocks are kept and handed
    // out again to later requests of (almost) the same size -- no hipMalloc / hipFree / synchronisation per call
    void *pool;
    // result / work caches of the threshold indexers (fmk_volume / fmk_dollar / fmk_threshold .hip), one per context
    void *idx_cache[3];
    // the caches are keyed on device POINTERS (amount, price): when fmk_free gives a block back that holds such a key, the
    // cache is marked stale, so a later allocation that recycles the address for other data cannot hit it
    const void *idx_key[3][2];
    int idx_stale[3];
    // pinned staging buffers / streams of fmk_h2d_columns (fmk_upload.hip), made on first use
    void *upload;
    // auxiliary stream + events of the pipelined time-bar step (fmk_ohlcv.hip), made on first use (fmk_ctx_aux)
    hipStream_t aux;
    hipEvent_t aev[4];
    // cfg 4 in one pass (fmk_fused.h): what the sizing call leaves for the fill call -- staged level rows, the list of bars the class
    // kernels still have to serve (fmk_barflow.hip: FuState); released by the fill call, the next sizing call, trim and destroy
    void *fused;
};
void fmk_fused_release(fmk_ctx *ctx);
int fmk_fused_fill(fmk_ctx *ctx, const double *d_price, const void *d_amount, int amount_is_f64, int64_t n, const int64_t *d_close_idx,
                   int64_t n_idx, const int8_t *d_side, double price_tick_size, const double *d_bar_lows, double imbalance_factor,
                   const int64_t *d_level_offsets, int64_t max_levels, const fmk_footprint_out *d_out, int64_t *d_n_bad_level,
                   int *handled);
int fmk_ctx_aux(fmk_ctx *ctx);
int fmk_pool_defer(fmk_ctx *ctx, int on);   // park fmk_free while a call launches on two streams (fmk_api.hip)
// fmk_indexers.hip: the time-bar indexer in stages (sample table, then edges [k0, k1) on a given stream, with the long-bar census)
int fmk_time_bar_coarse_launch(fmk_ctx *ctx, const int64_t *d_ts, int64_t n, const int64_t **coarse, int64_t *m_out, int *clear);
int fmk_time_bar_index_stage(fmk_ctx *ctx, hipStream_t st, const int64_t *d_ts, int64_t n, int64_t e0, int64_t d, int64_t ne,
                             const int64_t *coarse, int64_t m, int64_t k0, int64_t k1, int64_t *d_clock, int64_t *d_idx,
                             int *saw_long, int64_t long_min, int64_t max_blocks /* 0: one workgroup per 256 edges */);

int fmk_set_error(fmk_ctx *ctx, int code, const char *fmt, ...);
int fmk_scratch(fmk_ctx *ctx, size_t bytes, void **out);
// per-context result / work caches of the threshold indexers (released by fmk_ctx_trim and fmk_ctx_destroy)
void fmk_volume_trim(fmk_ctx *ctx);
void fmk_dollar_trim(fmk_ctx *ctx);
void fmk_threshold_trim(fmk_ctx *ctx);
void fmk_upload_trim(fmk_ctx *ctx);

// fmk_footprint.hip: footprint fill launches for the level classes wider than `lmin_start` (0: all bars)
int fmk_footprints_fill_classes(fmk_ctx *ctx, const double *d_price, const void *d_amount, int amount_is_f64,
                                const int64_t *d_close_idx, int64_t nb, const int8_t *d_side, double price_tick_size,
                                const double *d_bar_lows, double imb_mult, const int64_t *d_level_offsets, int lmin_start,
                                int64_t max_levels, const fmk_footprint_out *d_out, int64_t *d_n_bad_level,
                                int64_t n_ticks /* 0: unknown */,
                                double *d_median = nullptr /* float32 amounts: the median trade size from the same sweep */,
                                const unsigned long long *only_list = nullptr /* [0] = count, bars from [32]: only these bars, by the wave-per-bar classes */);
// median of the bars of more than min_cnt ticks (flag d_go), except those of skip_lo < ticks <= skip_hi (served by k_bar_ohlcv_mid)
int fmk_median_launch(fmk_ctx *ctx, const void *d_amount, int amount_is_f64, const int64_t *d_close_idx, int64_t nb,
                      int64_t min_cnt, const int *d_go, double *d_median, int64_t n_ticks, int64_t skip_lo = 0,
                      int64_t skip_hi = 0, int64_t skip_above = INT64_MAX /* bars of more ticks were served elsewhere */);
// the workgroup radix select (k_bar_median_long, 1024 threads) on a given list ([0] = count, then bar numbers), float32 amounts
int fmk_median_long_list_launch(fmk_ctx *ctx, const void *d_amount, const int64_t *d_close_idx, const int64_t *d_list, double *d_median);

// fmk_ohlcv.hip: pieces of comp_bar_ohlcv for cfg 4's first half (fmk_barflow.hip)
int fmk_ohlcv_leftover_launch(fmk_ctx *ctx, const double *p, const void *a, int amount_is_f64, const int64_t *ci, int64_t nb,
                              int64_t n, int64_t min_cnt, const int *go, double *d_open, double *d_high, double *d_low,
                              double *d_close, float *d_volume, double *d_vwap, int64_t *d_trades);
int fmk_median_small_launch(fmk_ctx *ctx, const float *d_amount, const int64_t *d_close_idx, int64_t nb, double *d_median,
                            int64_t n_ticks);
int fmk_median_small_ohlcv_long_launch(fmk_ctx *ctx, const double *p, const float *a, const int64_t *ci, int64_t nb, int64_t n,
                                       double *d_open, double *d_high, double *d_low, double *d_close, float *d_volume,
                                       double *d_vwap, int64_t *d_trades, double *d_median);
// fmk_median.hip: the bars of more than `min_cnt` ticks as a list ([0] = how many, then the bar numbers, any order) in a block of
// the context's pool (*list; the caller gives it back with fmk_free after queueing its kernels) -- the workgroup-per-bar
// kernels take their bars from it, so that a handful of very long bars spread over the whole chip
#define FMK_MAX_BAR_LISTS 8
// K lists in one pass and one allocation: list k = bars of edge[k] < ticks <= edge[k + 1]; free lists[0] (fmk_median.hip)
int fmk_long_bar_lists(fmk_ctx *ctx, const int64_t *d_close_idx, int64_t nb, int64_t n, int k, const int64_t *edge, const int *d_go,
                       int64_t **lists);
int fmk_long_bar_list(fmk_ctx *ctx, const int64_t *d_close_idx, int64_t nb, int64_t n, int64_t min_cnt, const int *d_go,
                      int64_t **list, int64_t max_cnt = INT64_MAX /* bars of min_cnt < ticks <= max_cnt */);

#define FMK_HIP(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            return fmk_set_error((ctx), e__ == hipErrorOutOfMemory ? FMK_E_NOMEM : FMK_E_HIP, \
                                 "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),     \
                                 __FILE__, __LINE__);                                        \
    } while (0)

#define FMK_TRY(expr)                   \
    do {                                \
        int rc__ = (expr);              \
        if (rc__ != FMK_OK) return rc__; \
    } while (0)

#define FMK_LAUNCH_CHECK(ctx) FMK_HIP(ctx, hipGetLastError())

static inline int64_t fmk_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

#ifdef __HIPCC__
// ---------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int fmk_lane() { return threadIdx.x & 63; }

// Make a wave-uniform 64-bit value provably uniform (SGPR pair) for the compiler.
__device__ __forceinline__ int64_t fmk_uniform(int64_t v)
{
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ int fmk_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// value of lane `src` (wave-uniform index) broadcast to the wave
__device__ __forceinline__ int64_t fmk_readlane(int64_t v, int src)
{
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)((uint64_t)v >> 32), src);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

template <typename T>
__device__ __forceinline__ T fmk_wave_sum(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double fmk_wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double fmk_wave_min(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int64_t fmk_wave_max(int64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int64_t w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
    return v;
}
__device__ __forceinline__ int64_t fmk_wave_min(int64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int64_t w = __shfl_xor(v, o, 64); v = w < v ? w : v; }
    return v;
}
// inclusive scan across the 64 lanes (Kogge-Stone, 6 steps)
template <typename T>
__device__ __forceinline__ T fmk_wave_iscan(T v)
{
    const int lane = fmk_lane();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        T w = __shfl_up(v, o, 64);
        if (lane >= o) v += w;
    }
    return v;
}

// amount column element j as float64 (float32 -> float64 is exact)
template <bool F64>
__device__ __forceinline__ double fmk_amt(const void *p, int64_t j)
{
    if constexpr (F64) return ((const double *)p)[j];
    else return (double)((const float *)p)[j];
}

// Python negative-index wrap of the reference (prices[-1] when close_idx[0] == -1)
__device__ __forceinline__ int64_t fmk_wrap(int64_t i, int64_t n) { return i < 0 ? i + n : i; }

// splitmix64-style counter hash shared with oracle/fmk_oracle.c (orc_mix64)
__host__ __device__ __forceinline__ uint64_t fmk_mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
#endif  // __HIPCC__
