// fmk_diag.hip -- calibration probe for the roofline figures of DESIGN.md: achievable read-only HBM bandwidth of a
// plain streaming kernel on this part (the tick reducers are read streams; the guide's ~6.3 TB/s ceiling is a COPY).
// Not used by any product path.  Built into its OWN library (lib/libfmk_diag.so, declared in include/fmk_diag.h): the drop-in ABI
// of libfmk_hip.so carries no probes.  The context type and fmk_set_error / fmk_h2d_columns come from libfmk_hip.so.
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "fmk_common.h"
#include "../../include/fmk_diag.h"

// variant 0: 16 B per lane per load (dwordx4), variant 1: 8 B per lane (dwordx2, what the reducers issue per price)
template <int VEC>
__global__ __launch_bounds__(256) void k_diag_read(const uint4 *__restrict__ p, int64_t n16, unsigned long long *sink)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    if (VEC == 0) {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
            const uint4 v = p[i];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    } else {
        const uint2 *q = (const uint2 *)p;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n16; i += stride) {
            const uint2 v = q[i];
            acc ^= v.x ^ v.y;
        }
    }
    if (acc == 0x9E3779B9u) atomicAdd(sink, 1ULL);        // practically never: keeps the loads alive
}

// variants 4 / 5: the same two widths as NON-TEMPORAL loads (nt: the lines are not kept in L2 for reuse)
template <int VEC>
__global__ __launch_bounds__(256) void k_diag_read_nt(const uint4 *__restrict__ p, int64_t n16, unsigned long long *sink)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    if (VEC == 0) {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const u4 *q = (const u4 *)p;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
            const u4 v = __builtin_nontemporal_load(q + i);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    } else {
        const unsigned long long *q = (const unsigned long long *)p;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n16; i += stride) {
            const unsigned long long v = __builtin_nontemporal_load(q + i);
            acc ^= (unsigned)v ^ (unsigned)(v >> 32);
        }
    }
    if (acc == 0x9E3779B9u) atomicAdd(sink, 1ULL);
}

// variant 2: 4 B per lane (dword, what the reducers issue per float32 amount); variant 3: 8 B per lane STORES (calibration of
// WRITE_SIZE on a known byte count; the buffer is overwritten with its own indices)
__global__ __launch_bounds__(256) void k_diag_read4(const unsigned *__restrict__ p, int64_t n4, unsigned long long *sink)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) acc ^= p[i];
    if (acc == 0x9E3779B9u) atomicAdd(sink, 1ULL);
}
__global__ __launch_bounds__(256) void k_diag_write8(unsigned long long *__restrict__ p, int64_t n8)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) p[i] = (unsigned long long)i;
}

extern "C" int fmk_diag_read_bandwidth(fmk_ctx *ctx, const void *d_buf, size_t bytes, int variant, int blocks_per_cu,
                                       double *elapsed_ms)
{
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t n16 = (int64_t)(bytes / 16);
    const unsigned blocks = (unsigned)(ctx->n_cu * (blocks_per_cu > 0 ? blocks_per_cu : 8));
    unsigned long long *sink = (unsigned long long *)(ctx->d_mail + 60);
    FMK_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    if (variant == 0) k_diag_read<0><<<blocks, 256, 0, ctx->stream>>>((const uint4 *)d_buf, n16, sink);
    else if (variant == 1) k_diag_read<1><<<blocks, 256, 0, ctx->stream>>>((const uint4 *)d_buf, n16, sink);
    else if (variant == 2) k_diag_read4<<<blocks, 256, 0, ctx->stream>>>((const unsigned *)d_buf, n16 * 4, sink);
    else if (variant == 4) k_diag_read_nt<0><<<blocks, 256, 0, ctx->stream>>>((const uint4 *)d_buf, n16, sink);
    else if (variant == 5) k_diag_read_nt<1><<<blocks, 256, 0, ctx->stream>>>((const uint4 *)d_buf, n16, sink);
    else k_diag_write8<<<blocks, 256, 0, ctx->stream>>>((unsigned long long *)d_buf, n16 * 2);
    FMK_LAUNCH_CHECK(ctx);
    FMK_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    FMK_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    FMK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *elapsed_ms = (double)ms;
    return FMK_OK;
}

// Two columns read in lock-step (tools/placement.py): does the allocation-dependent step time of the bar reducers come
// from the memory system seeing TWO concurrent streams (8 B and 4 B per element at the same index, two allocations), and
// does the one-wave-per-bar walk (a wave streams `seg` contiguous elements, then jumps by the number of waves) matter?
//   pattern 0: flat grid-stride over both columns      1: segment walk over both      2: segment walk over `a` only
template <int PATTERN>
__global__ __launch_bounds__(256) void k_diag_read2(const uint2 *__restrict__ a, const unsigned *__restrict__ b, int64_t n,
                                                    int seg, unsigned long long *sink)
{
    unsigned acc = 0;
    if (PATTERN == 0) {
        const int64_t stride = (int64_t)gridDim.x * blockDim.x;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            const uint2 v = a[i];
            acc ^= v.x ^ v.y ^ b[i];
        }
    } else if (PATTERN == 3) {
        // the reducers' own issue order: a wave asks for up to 16 rows of BOTH columns of its segment before it looks at any of them
        const int lane = threadIdx.x & 63;
        const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
        const int64_t nseg = n / seg;
        for (int64_t s = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); s < nseg; s += nwaves) {
            const int64_t base = s * seg;
            for (int j0 = 0; j0 < seg; j0 += 1024) {
                uint2 v[16];
                unsigned u[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int j = j0 + k * 64 + lane;
                    v[k] = j < seg ? a[base + j] : make_uint2(0u, 0u);
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int j = j0 + k * 64 + lane;
                    u[k] = j < seg ? b[base + j] : 0u;
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) acc ^= v[k].x ^ v[k].y ^ u[k];
            }
        }
    } else {
        const int lane = threadIdx.x & 63;
        const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
        const int64_t nseg = n / seg;
        for (int64_t s = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); s < nseg; s += nwaves) {
            const int64_t base = s * seg;
            for (int j = lane; j < seg; j += 64) {
                const uint2 v = a[base + j];
                acc ^= v.x ^ v.y;
                if (PATTERN == 1) acc ^= b[base + j];
            }
        }
    }
    if (acc == 0x9E3779B9u) atomicAdd(sink, 1ULL);
}

extern "C" int fmk_diag_read_two_streams(fmk_ctx *ctx, const void *d_a8, const void *d_b4, int64_t n, int pattern, int seg,
                                         int blocks_per_cu, double *elapsed_ms)
{
    if (n <= 0 || seg <= 0 || pattern < 0 || pattern > 3) return fmk_set_error(ctx, FMK_E_ARG, "diag: bad arguments");
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    const unsigned blocks = (unsigned)(ctx->n_cu * (blocks_per_cu > 0 ? blocks_per_cu : 8));
    unsigned long long *sink = (unsigned long long *)(ctx->d_mail + 60);
    FMK_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    if (pattern == 0) k_diag_read2<0><<<blocks, 256, 0, ctx->stream>>>((const uint2 *)d_a8, (const unsigned *)d_b4, n, seg, sink);
    else if (pattern == 1) k_diag_read2<1><<<blocks, 256, 0, ctx->stream>>>((const uint2 *)d_a8, (const unsigned *)d_b4, n, seg, sink);
    else if (pattern == 3) k_diag_read2<3><<<blocks, 256, 0, ctx->stream>>>((const uint2 *)d_a8, (const unsigned *)d_b4, n, seg, sink);
    else k_diag_read2<2><<<blocks, 256, 0, ctx->stream>>>((const uint2 *)d_a8, (const unsigned *)d_b4, n, seg, sink);
    FMK_LAUNCH_CHECK(ctx);
    FMK_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    FMK_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    FMK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *elapsed_ms = (double)ms;
    return FMK_OK;
}

// Dependent-access latency seen by ONE wave (tools/hoplat.py): hop h reads `loads` coalesced 512 B rows starting at
// element pos of an 8-byte array, then moves pos by `stride` elements plus a value-dependent 0 (so the next address
// depends on the data).  Reports shader cycles per hop.  What the volume chain walk pays per close.
__global__ __launch_bounds__(64) void k_diag_hops(const double *__restrict__ buf, int64_t n, int64_t stride, int loads, int hops,
                                                  long long *__restrict__ out)
{
    const int lane = threadIdx.x;
    int64_t pos = 0;
    double acc = 0.0;
    const long long t0 = __builtin_readcyclecounter();
    for (int h = 0; h < hops; ++h) {
        double v = 0.0;
        for (int k = 0; k < loads; ++k) v += buf[pos + k * 64 + lane];
        // lane 0's value decides the next position: always + stride for this data (zeros), but the hardware cannot know
        const double first = __longlong_as_double(((long long)__builtin_amdgcn_readfirstlane((int)((unsigned long long)__double_as_longlong(v) >> 32)) << 32));
        acc += v;
        pos += stride + (first > 1e300 ? 1 : 0);
        if (pos + loads * 64 >= n) pos = 0;
    }
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { out[0] = t1 - t0; out[1] = (long long)acc; }
}

extern "C" int fmk_diag_hop_latency(fmk_ctx *ctx, const void *d_buf, int64_t n, int64_t stride, int loads, int hops,
                                    double *cycles_per_hop, double *elapsed_ms)
{
    if (n <= 0 || stride <= 0 || loads < 1 || loads > 16 || hops < 1) return fmk_set_error(ctx, FMK_E_ARG, "diag: bad arguments");
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    long long *d_out = (long long *)(ctx->d_mail + 56);
    FMK_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    k_diag_hops<<<1, 64, 0, ctx->stream>>>((const double *)d_buf, n, stride, loads, hops, d_out);
    FMK_LAUNCH_CHECK(ctx);
    FMK_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    FMK_HIP(ctx, hipMemcpyAsync(&ctx->h_mail[8], d_out, 16, hipMemcpyDeviceToHost, ctx->stream));
    FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    FMK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *cycles_per_hop = (double)ctx->h_mail[8] / hops;
    *elapsed_ms = (double)ms;
    return FMK_OK;
}



// Three columns (price 8 B, amount 4 B, side 1 B per tick) read by one wave per `seg`-tick segment in tiles of 512 ticks, with
// lane l owning R CONSECUTIVE ticks of every group of 64 R (tools/ownedread.py; round 6): R = 1 is the reducers' chunk layout (one
// tick per lane and load: 24 loads per tile), R = 2 / 4 / 8 read 16-byte vectors per lane (12 / 8 / 7 loads per tile; for R >= 4 a
// price load instruction touches every second / every line of its 2 / 4 KB span and the R / 2 loads of a lane group share lines).
// The question: does the memory pipeline deliver the same bytes per second when lanes own consecutive ticks (a layout in which a
// running-sum walk needs one wave scan per tile instead of one per chunk)?  Not used by any product path.
template <int R>
__global__ __launch_bounds__(256) void k_diag_read_owned(const double *__restrict__ price, const float *__restrict__ amount,
                                                         const signed char *__restrict__ side, int64_t n, int seg,
                                                         unsigned long long *sink)
{
    typedef unsigned u4 __attribute__((ext_vector_type(4), aligned(4)));
    typedef unsigned u2 __attribute__((ext_vector_type(2), aligned(1)));
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t nseg = n / seg;
    unsigned acc = 0;
    for (int64_t s = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); s < nseg; s += nwaves) {
        const int64_t base = s * seg;
        const double *pb = price + base;
        const float *ab = amount + base;
        const signed char *sb = side + base;
        for (int t0 = 0; t0 < seg; t0 += 512) {
            const int last = seg - 1;
            if constexpr (R == 1) {
                uint2 pv[8];
                unsigned av[8];
                int sv[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    int j = t0 + c * 64 + lane;
                    j = j < last ? j : last;
                    pv[c] = ((const uint2 *)pb)[j];
                    av[c] = ((const unsigned *)ab)[j];
                    sv[c] = sb[j];
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) acc ^= pv[c].x ^ pv[c].y ^ av[c] ^ (unsigned)sv[c];
            } else if constexpr (R == 2) {
                u4 pv[4];
                uint2 av[4];
                unsigned short sv[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int j = t0 + g * 128 + 2 * lane;
                    j = j < last - 1 ? j : last - 1;
                    pv[g] = *(const u4 *)(pb + j);
                    av[g] = *(const uint2 *)(ab + j);
                    sv[g] = *(const unsigned short *)(sb + j);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) acc ^= pv[g].x ^ pv[g].y ^ pv[g].z ^ pv[g].w ^ av[g].x ^ av[g].y ^ sv[g];
            } else if constexpr (R == 4) {
                u4 pv[4];
                u4 av[2];
                unsigned sv[2];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    int j = t0 + g * 256 + 4 * lane;
                    j = j < last - 3 ? j : last - 3;
                    pv[2 * g] = *(const u4 *)(pb + j);
                    pv[2 * g + 1] = *(const u4 *)(pb + j + 2);
                    av[g] = *(const u4 *)(ab + j);
                    sv[g] = *(const unsigned *)(sb + j);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) acc ^= pv[g].x ^ pv[g].y ^ pv[g].z ^ pv[g].w;
#pragma unroll
                for (int g = 0; g < 2; ++g) acc ^= av[g].x ^ av[g].y ^ av[g].z ^ av[g].w ^ sv[g];
            } else {
                int j = t0 + 8 * lane;
                j = j < last - 7 ? j : last - 7;
                u4 pv[4];
                u4 av[2];
#pragma unroll
                for (int k = 0; k < 4; ++k) pv[k] = *(const u4 *)(pb + j + 2 * k);
#pragma unroll
                for (int k = 0; k < 2; ++k) av[k] = *(const u4 *)(ab + j + 4 * k);
                const u2 sv = *(const u2 *)(sb + j);
#pragma unroll
                for (int g = 0; g < 4; ++g) acc ^= pv[g].x ^ pv[g].y ^ pv[g].z ^ pv[g].w;
                acc ^= av[0].x ^ av[0].y ^ av[0].z ^ av[0].w ^ av[1].x ^ av[1].y ^ av[1].z ^ av[1].w ^ sv.x ^ sv.y;
            }
        }
    }
    if (acc == 0x9E3779B9u) atomicAdd(sink, 1ULL);
}

extern "C" int fmk_diag_read_owned(fmk_ctx *ctx, const double *d_price, const float *d_amount, const signed char *d_side, int64_t n,
                                   int seg, int r, int blocks_per_cu, double *elapsed_ms)
{
    if (n <= 0 || seg < 16 || (r != 1 && r != 2 && r != 4 && r != 8)) return fmk_set_error(ctx, FMK_E_ARG, "diag: bad arguments");
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    const unsigned blocks = (unsigned)(ctx->n_cu * (blocks_per_cu > 0 ? blocks_per_cu : 4));
    unsigned long long *sink = (unsigned long long *)(ctx->d_mail + 60);
    FMK_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    if (r == 1) k_diag_read_owned<1><<<blocks, 256, 0, ctx->stream>>>(d_price, d_amount, d_side, n, seg, sink);
    else if (r == 2) k_diag_read_owned<2><<<blocks, 256, 0, ctx->stream>>>(d_price, d_amount, d_side, n, seg, sink);
    else if (r == 4) k_diag_read_owned<4><<<blocks, 256, 0, ctx->stream>>>(d_price, d_amount, d_side, n, seg, sink);
    else k_diag_read_owned<8><<<blocks, 256, 0, ctx->stream>>>(d_price, d_amount, d_side, n, seg, sink);
    FMK_LAUNCH_CHECK(ctx);
    FMK_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    FMK_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    FMK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *elapsed_ms = (double)ms;
    return FMK_OK;
}


// Amounts with a full random 24-bit mantissa in [2^-7, 2): float32 sums of them are inexact in every order, like real trade
// sizes (the synthetic stream's dyadic amounts make every footprint bar take the certified integer path).  bench.py times
// cfg 4 on both.  Not used by any product path.
__global__ __launch_bounds__(256) void k_diag_fill_amounts(uint64_t seed, int64_t n, float *__restrict__ amount)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t h = fmk_mix64(seed ^ (0xA5A5A5A5ULL + (uint64_t)i));
        const uint32_t bits = (uint32_t)(h & 0x7FFFFFu) | ((120u + (uint32_t)((h >> 23) & 7u)) << 23);
        amount[i] = __uint_as_float(bits);
    }
}

extern "C" int fmk_diag_fill_amounts_dev(fmk_ctx *ctx, uint64_t seed, int64_t n, float *d_amount)
{
    if (n <= 0) return FMK_OK;
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    k_diag_fill_amounts<<<(unsigned)(ctx->n_cu * 16), 256, 0, ctx->stream>>>(seed, n, d_amount);
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}

// A one-thread kernel whose name tools/cfgprof_summarize.py looks for in a rocprofv3 trace: `which` = 1 opens the measured region of
// tools/cfgprof.py, 2 closes it (the dispatches between the two are the config's, everything else is set-up).
__global__ void k_diag_marker(int which, int *sink) { if (sink && which < 0) *sink = which; }
extern "C" int fmk_diag_marker_dev(fmk_ctx *ctx, int which)
{
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    k_diag_marker<<<1, 1, 0, ctx->stream>>>(which, nullptr);
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}

// Diagnostics: the box's host-to-device rate for one buffer of `bytes` -- pinned = 1: from hipHostMalloc memory (the link's
// ceiling), 0: plain hipMemcpy from malloc'ed memory (what a caller's NumPy array gets without fmk_h2d_columns), 2: fmk_h2d_columns
// from malloc'ed memory.  Best of three, GB/s.
extern "C" int fmk_diag_h2d_rate(fmk_ctx *ctx, size_t bytes, int mode, double *gbps)
{
    *gbps = 0.0;
    if (bytes == 0) return FMK_OK;
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    void *d = nullptr, *h = nullptr;
    FMK_HIP(ctx, hipMalloc(&d, bytes));
    hipError_t e = hipSuccess;
    if (mode == 1) e = hipHostMalloc(&h, bytes, hipHostMallocDefault);
    else h = malloc(bytes);
    if (e != hipSuccess || !h) { (void)hipFree(d); return fmk_set_error(ctx, FMK_E_NOMEM, "fmk_diag_h2d_rate: host buffer"); }
    memset(h, 1, bytes);
    double best = 0.0;
    int rc = FMK_OK;
    for (int r = 0; r < 4 && rc == FMK_OK; ++r) {
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        if (mode == 2) {
            void *dd[1] = {d};
            const void *ss[1] = {h};
            rc = fmk_h2d_columns(ctx, 1, dd, ss, &bytes);
        } else {
            e = hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);
            if (e != hipSuccess) rc = fmk_set_error(ctx, FMK_E_HIP, "hipMemcpy: %s", hipGetErrorString(e));
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const double s = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        if (r > 0 && s > 0 && (double)bytes / s / 1e9 > best) best = (double)bytes / s / 1e9;
    }
    if (mode == 1) (void)hipHostFree(h); else free(h);
    (void)hipFree(d);
    *gbps = best;
    return rc;
}
