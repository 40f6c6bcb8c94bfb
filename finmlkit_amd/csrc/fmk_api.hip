// fmk_api.hip -- context, memory, timing and error plumbing of libfmk_hip.so.
#include <stdarg.h>
#include <stdlib.h>

#include <map>
#include <vector>
#include <unordered_map>

#include "fmk_common.h"

static char g_err[512] = "";

int fmk_set_error(fmk_ctx *ctx, int code, const char *fmt, ...)
{
    char *dst = ctx ? ctx->err : g_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}

static void fmk_pool_destroy(fmk_ctx *ctx);
static void fmk_pool_flush(fmk_ctx *ctx);

extern "C" {

int fmk_abi_version(void) { return FMK_ABI_VERSION; }

const char *fmk_last_error(const fmk_ctx *ctx) { return ctx ? ctx->err : g_err; }

int fmk_device_count(int *count)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fmk_set_error(nullptr, FMK_E_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return FMK_OK;
}

int fmk_ctx_create(int device, fmk_ctx **out)
{
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fmk_set_error(nullptr, FMK_E_NODEVICE,
                             "no HIP device (hipGetDeviceCount: %s, count=%d); libfmk_hip has no CPU fallback",
                             hipGetErrorString(e), n);
    if (device < 0 || device >= n)
        return fmk_set_error(nullptr, FMK_E_ARG, "device %d out of range [0,%d)", device, n);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess)
        return fmk_set_error(nullptr, FMK_E_NODEVICE, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fmk_set_error(nullptr, FMK_E_NODEVICE, "device %d is %s; this library carries gfx950 code only",
                             device, prop.gcnArchName);
    fmk_ctx *c = (fmk_ctx *)calloc(1, sizeof(fmk_ctx));
    if (!c) return fmk_set_error(nullptr, FMK_E_NOMEM, "calloc");
    c->device = device;
    c->n_cu = prop.multiProcessorCount;
#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e2 = (x);                                                               \
        if (e2 != hipSuccess) {                                                            \
            fmk_set_error(nullptr, FMK_E_HIP, "%s: %s", #x, hipGetErrorString(e2));        \
            free(c);                                                                       \
            return FMK_E_HIP;                                                              \
        }                                                                                  \
    } while (0)
    CK(hipSetDevice(device));
    CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CK(hipEventCreate(&c->ev0));
    CK(hipEventCreate(&c->ev1));
    CK(hipHostMalloc((void **)&c->h_mail, 64 * sizeof(int64_t), hipHostMallocDefault));
    memset(c->h_mail, 0, 64 * sizeof(int64_t));
    CK(hipMalloc((void **)&c->d_mail, 64 * sizeof(int64_t)));
#undef CK
    *out = c;
    return FMK_OK;
}

int fmk_ctx_destroy(fmk_ctx *ctx)
{
    if (!ctx) return FMK_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    fmk_fused_release(ctx);
    fmk_pool_destroy(ctx);
    fmk_volume_trim(ctx);
    fmk_dollar_trim(ctx);
    fmk_threshold_trim(ctx);
    fmk_upload_trim(ctx);
    (void)hipFree(ctx->d_mail);
    (void)hipHostFree(ctx->h_mail);
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    if (ctx->kev[0][0])
        for (int i = 0; i < FMK_PROFILE_SLOTS; ++i) { (void)hipEventDestroy(ctx->kev[i][0]); (void)hipEventDestroy(ctx->kev[i][1]); }
    if (ctx->aux) {
        (void)hipStreamSynchronize(ctx->aux);
        for (int i = 0; i < 4; ++i) (void)hipEventDestroy(ctx->aev[i]);
        (void)hipStreamDestroy(ctx->aux);
    }
    (void)hipStreamDestroy(ctx->stream);
    free(ctx);
    return FMK_OK;
}

int fmk_ctx_trim(fmk_ctx *ctx)
{
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->scratch) { (void)hipFree(ctx->scratch); ctx->scratch = nullptr; ctx->scratch_bytes = 0; }
    fmk_fused_release(ctx);
    fmk_pool_flush(ctx);
    fmk_volume_trim(ctx);
    fmk_dollar_trim(ctx);
    fmk_threshold_trim(ctx);
    fmk_upload_trim(ctx);
    return FMK_OK;
}

// Kernels that wait for other workgroups inside a launch (the one-pass scans) bound their spins and, instead of hanging,
// raise h_mail[40] (pinned host memory the device writes directly); it is looked at whenever the host waits for the stream.
static int fmk_check_device_error(fmk_ctx *ctx)
{
    if (ctx->h_mail[40] != 0) {
        const long long code = (long long)ctx->h_mail[40];
        ctx->h_mail[40] = 0;
        return fmk_set_error(ctx, FMK_E_HIP, "a one-pass scan kernel gave up waiting for another workgroup (code %lld); "
                             "its output is not valid", code);
    }
    return FMK_OK;
}

int fmk_ctx_sync(fmk_ctx *ctx)
{
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return fmk_check_device_error(ctx);
}

void *fmk_ctx_stream(fmk_ctx *ctx) { return (void *)ctx->stream; }

// ---------------------------------------------------------------------------------------
// Caching allocator.  Everything the library enqueues runs on the context's ONE stream, so a block that is freed
// and handed out again is reused in stream order: no synchronisation is needed, and hipMalloc / hipFree (milliseconds
// for the multi-GB per-tick outputs) leave the steady state of a pipeline.  Blocks are matched by size (a cached
// block may be up to 1/8 larger than the request); when hipMalloc fails the cache is flushed and the call retried.
// FMK_POOL=0 in the environment disables caching.  Buffers that OTHER streams touch (the halo exchange of bench.py)
// must simply stay allocated while those streams use them -- as with any allocator.
// ---------------------------------------------------------------------------------------
struct FmkPool {
    std::multimap<size_t, void *> free_blocks;      // size -> block
    std::unordered_map<void *, size_t> live;        // block -> size (everything handed out)
    size_t cached_bytes = 0;
    bool enabled = true;
    // While a call runs launches on the context's stream AND on its auxiliary stream, a block freed by one side must not be handed to the
    // other (the free list is ordered by ONE stream's launch order): frees are parked here until the call has joined the streams.
    bool defer = false;
    std::vector<void *> deferred;
};

static FmkPool *fmk_pool(fmk_ctx *ctx)
{
    if (!ctx->pool) {
        FmkPool *p = new FmkPool();
        const char *v = getenv("FMK_POOL");
        if (v && atoi(v) == 0) p->enabled = false;
        ctx->pool = p;
    }
    return (FmkPool *)ctx->pool;
}

static void fmk_pool_flush(fmk_ctx *ctx)
{
    FmkPool *p = fmk_pool(ctx);
    if (p->free_blocks.empty()) return;
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : p->free_blocks) (void)hipFree(kv.second);
    p->free_blocks.clear();
    p->cached_bytes = 0;
}

static void fmk_pool_destroy(fmk_ctx *ctx)
{
    if (!ctx->pool) return;
    fmk_pool_flush(ctx);
    FmkPool *p = (FmkPool *)ctx->pool;
    for (auto &kv : p->live) (void)hipFree(kv.first);        // blocks that outlive the context go with it
    delete p;
    ctx->pool = nullptr;
}

int fmk_alloc(fmk_ctx *ctx, size_t bytes, void **dptr)
{
    *dptr = nullptr;
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    if (bytes == 0) bytes = 16;
    bytes = (bytes + 255) & ~(size_t)255;
    FmkPool *p = fmk_pool(ctx);
    if (p->enabled) {
        auto it = p->free_blocks.lower_bound(bytes);
        if (it != p->free_blocks.end() && it->first <= bytes + (bytes >> 3)) {
            *dptr = it->second;
            p->live[*dptr] = it->first;
            p->cached_bytes -= it->first;
            p->free_blocks.erase(it);
            return FMK_OK;
        }
    }
    hipError_t e = hipMalloc(dptr, bytes);
    if (e == hipErrorOutOfMemory && !p->free_blocks.empty()) {
        (void)hipGetLastError();
        fmk_pool_flush(ctx);
        e = hipMalloc(dptr, bytes);
    }
    if (e != hipSuccess) {
        *dptr = nullptr;
        return fmk_set_error(ctx, e == hipErrorOutOfMemory ? FMK_E_NOMEM : FMK_E_HIP, "hipMalloc(%zu) failed: %s", bytes,
                             hipGetErrorString(e));
    }
    p->live[*dptr] = bytes;
    return FMK_OK;
}

int fmk_free(fmk_ctx *ctx, void *dptr)
{
    if (!dptr) return FMK_OK;
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    FmkPool *p = fmk_pool(ctx);
    if (p->defer) {
        if (p->live.find(dptr) == p->live.end())
            return fmk_set_error(ctx, FMK_E_ARG, "fmk_free: %p is not a live block of this context (double free?)", dptr);
        for (void *q : p->deferred)
            if (q == dptr) return fmk_set_error(ctx, FMK_E_ARG, "fmk_free: %p freed twice", dptr);
        p->deferred.push_back(dptr);
        return FMK_OK;
    }
    auto it = p->live.find(dptr);
    if (it == p->live.end())                         // not handed out by fmk_alloc, or freed twice: a raw hipFree here
        return fmk_set_error(ctx, FMK_E_ARG,         // could release a block that sits in the free list (use after free)
                             "fmk_free: %p is not a live block of this context (double free?)", dptr);
    const size_t bytes = it->second;
    p->live.erase(it);
    for (int k = 0; k < 3; ++k)                      // result caches of the threshold indexers keyed on pointers into this block
        for (int q = 0; q < 2; ++q) {
            const char *key = (const char *)ctx->idx_key[k][q];
            if (key && key >= (const char *)dptr && key < (const char *)dptr + bytes) ctx->idx_stale[k] = 1;
        }
    if (!p->enabled) {
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        FMK_HIP(ctx, hipFree(dptr));
        return FMK_OK;
    }
    p->free_blocks.emplace(bytes, dptr);
    p->cached_bytes += bytes;
    return FMK_OK;
}

int fmk_memset(fmk_ctx *ctx, void *dptr, int value, size_t bytes)
{
    if (bytes == 0) return FMK_OK;
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    FMK_HIP(ctx, hipMemsetAsync(dptr, value, bytes, ctx->stream));
    return FMK_OK;
}

int fmk_h2d(fmk_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return FMK_OK;
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    FMK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FMK_OK;
}

int fmk_d2h(fmk_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return FMK_OK;
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    FMK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return fmk_check_device_error(ctx);
}

int fmk_d2d(fmk_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return FMK_OK;
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    FMK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return FMK_OK;
}

int fmk_mem_info(fmk_ctx *ctx, size_t *free_bytes, size_t *total_bytes)
{
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    FMK_HIP(ctx, hipMemGetInfo(free_bytes, total_bytes));
    if (ctx->pool) *free_bytes += ((FmkPool *)ctx->pool)->cached_bytes;     // cached blocks are available memory
    return FMK_OK;
}

int fmk_timer_start(fmk_ctx *ctx)
{
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    FMK_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return FMK_OK;
}

int fmk_timer_stop(fmk_ctx *ctx, double *elapsed_ms)
{
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    FMK_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    FMK_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    FMK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *elapsed_ms = (double)ms;
    return FMK_OK;
}

int fmk_event_create(fmk_ctx *ctx, void **event)
{
    hipEvent_t e;
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    FMK_HIP(ctx, hipEventCreate(&e));
    *event = (void *)e;
    return FMK_OK;
}

int fmk_event_destroy(fmk_ctx *ctx, void *event)
{
    if (event) FMK_HIP(ctx, hipEventDestroy((hipEvent_t)event));
    return FMK_OK;
}

int fmk_event_record(fmk_ctx *ctx, void *event)
{
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    FMK_HIP(ctx, hipEventRecord((hipEvent_t)event, ctx->stream));
    return FMK_OK;
}

int fmk_event_elapsed(fmk_ctx *ctx, void *start, void *stop, double *elapsed_ms)
{
    float ms = 0.f;
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    FMK_HIP(ctx, hipEventSynchronize((hipEvent_t)stop));
    FMK_HIP(ctx, hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop));
    *elapsed_ms = (double)ms;
    return FMK_OK;
}

int fmk_profile_enable(fmk_ctx *ctx, int on)
{
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    if (on && !ctx->kev[0][0])
        for (int i = 0; i < FMK_PROFILE_SLOTS; ++i) {
            FMK_HIP(ctx, hipEventCreate(&ctx->kev[i][0]));
            FMK_HIP(ctx, hipEventCreate(&ctx->kev[i][1]));
        }
    ctx->profile_on = on;
    ctx->profile_n = 0;
    return FMK_OK;
}

int fmk_ctx_set_fast_threshold(fmk_ctx *ctx, int on)
{
    ctx->fast_threshold = on ? 1 : 0;
    return FMK_OK;
}

int fmk_ctx_set_enqueue_only(fmk_ctx *ctx, int on)
{
    ctx->enqueue_only = on ? 1 : 0;
    return FMK_OK;
}

// launches timed since fmk_profile_enable(1): more than FMK_PROFILE_SLOTS means the ring has wrapped -- slot i then holds launch
// number (total - FMK_PROFILE_SLOTS + ((i - total) mod FMK_PROFILE_SLOTS)), i.e. the last FMK_PROFILE_SLOTS launches, oldest at
// slot total mod FMK_PROFILE_SLOTS
int fmk_profile_count(fmk_ctx *ctx, int64_t *total)
{
    *total = ctx->profile_n;
    return FMK_OK;
}

int fmk_profile_read(fmk_ctx *ctx, double *ms, int capacity, int *count)
{
    FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int n = ctx->profile_n < FMK_PROFILE_SLOTS ? ctx->profile_n : FMK_PROFILE_SLOTS;
    if (n > capacity) n = capacity;
    for (int i = 0; i < n; ++i) {
        float t = 0.f;
        FMK_HIP(ctx, hipEventElapsedTime(&t, ctx->kev[i][0], ctx->kev[i][1]));
        ms[i] = (double)t;
    }
    *count = n;
    return FMK_OK;
}

}  // extern "C"

int fmk_scratch(fmk_ctx *ctx, size_t bytes, void **out)
{
    if (bytes > ctx->scratch_bytes) {
        FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->scratch) FMK_HIP(ctx, hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
        size_t want = bytes + (bytes >> 2) + 4096;
        FMK_HIP(ctx, hipMalloc(&ctx->scratch, want));
        ctx->scratch_bytes = want;
    }
    *out = ctx->scratch;
    return FMK_OK;
}

// on = 1: park every fmk_free until on = 0 (then they are carried out in order).  For calls that launch on two streams at once.
int fmk_pool_defer(fmk_ctx *ctx, int on)
{
    FmkPool *p = fmk_pool(ctx);
    if (on) { p->defer = true; return FMK_OK; }
    p->defer = false;
    int rc = FMK_OK;
    std::vector<void *> todo;
    todo.swap(p->deferred);
    for (void *q : todo) { const int r = fmk_free(ctx, q); if (r != FMK_OK) rc = r; }
    return rc;
}


// the auxiliary stream of the pipelined time-bar step and its events (timing disabled: they only order the two streams)
int fmk_ctx_aux(fmk_ctx *ctx)
{
    if (ctx->aux) return FMK_OK;
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st;
    // DEFAULT priority.  The lowest priority was this stream's first setting (its kernels run beside a launch of the context's stream and
    // must not take that launch's wave slots) -- the capped grids of those kernels do that on their own, and a lowest-priority queue
    // holding a blocked barrier packet makes every small dispatch of the context's queue take ~45 us instead of ~5: the sharded step,
    // which enqueues ~30 of them per step with the next step already queued, went from 2.35 to 2.9..3.2 ms (profiles/r04_sharded_step.txt);
    // the single-GPU step measures the same either way.
    FMK_HIP(ctx, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int i = 0; i < 4; ++i) FMK_HIP(ctx, hipEventCreateWithFlags(&ctx->aev[i], hipEventDisableTiming));
    ctx->aux = st;
    return FMK_OK;
}

