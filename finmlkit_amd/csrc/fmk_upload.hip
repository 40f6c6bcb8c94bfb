// fmk_upload.hip -- host columns -> HBM for the host-pointer side of the boundary (BarBuilderBase._device, the host-pointer
// reducers): what the reference never needs (its arrays stay on the host, base.py:140-144) and what decides this build's wall time
// when a caller arrives with NumPy arrays -- 39 M ticks are 0.8 GB of columns and ~0.3 ms of kernels.
//
// Measured on the MI355X boxes (tools/apibench.py, profiles/r03_apibench.txt): the runtime's own copy from pageable memory reaches
// 56.2 GB/s = 98 % of the 57.5 GB/s a hipMemcpy from hipHostMalloc memory gets, so fmk_h2d_columns hands each column to
// hipMemcpyAsync as it is -- one call for the whole frame, one synchronisation.  The staged path below (worker threads copy 4 MiB
// chunks into their own pinned double buffers and enqueue hipMemcpyAsync on their own streams: 53-55 GB/s here with 4 workers,
// 30 GB/s with one) is kept for hosts whose runtime stages pageable copies through a single bounce buffer: FMK_UPLOAD_THREADS=n
// turns it on.  Nothing is pinned in place (hipHostRegister of a caller's array costs more than the copy).
#include <stdlib.h>

#include <atomic>
#include <thread>
#include <vector>

#include "fmk_common.h"

namespace {

constexpr size_t UP_CHUNK = (size_t)4 << 20;
constexpr int UP_MAX_THREADS = 16;

struct UpWorker {
    hipStream_t stream = nullptr;
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    int used[2] = {0, 0};
};

struct UpState {
    int n = 0;
    UpWorker w[UP_MAX_THREADS];
};

int up_threads()
{
    const char *e = getenv("FMK_UPLOAD_THREADS");                 // 0 / unset: the runtime's own pageable copy
    int v = e ? atoi(e) : 0;
    if (v < 0) v = 0;
    if (v > UP_MAX_THREADS) v = UP_MAX_THREADS;
    return v;
}

struct UpItem { int col; size_t off, len; };

}  // namespace

void fmk_upload_trim(fmk_ctx *ctx)
{
    UpState *st = (UpState *)ctx->upload;
    if (!st) return;
    for (int i = 0; i < st->n; ++i) {
        UpWorker &w = st->w[i];
        if (w.stream) (void)hipStreamSynchronize(w.stream);
        for (int k = 0; k < 2; ++k) {
            if (w.ev[k]) (void)hipEventDestroy(w.ev[k]);
            if (w.pin[k]) (void)hipHostFree(w.pin[k]);
        }
        if (w.stream) (void)hipStreamDestroy(w.stream);
    }
    delete st;
    ctx->upload = nullptr;
}

static int up_state(fmk_ctx *ctx, int want, UpState **out)
{
    UpState *st = (UpState *)ctx->upload;
    if (!st) { st = new UpState(); ctx->upload = st; }
    while (st->n < want) {
        UpWorker &w = st->w[st->n];
        FMK_HIP(ctx, hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            FMK_HIP(ctx, hipHostMalloc(&w.pin[k], UP_CHUNK, hipHostMallocDefault));
            FMK_HIP(ctx, hipEventCreateWithFlags(&w.ev[k], hipEventDisableTiming));
        }
        ++st->n;
    }
    *out = st;
    return FMK_OK;
}

extern "C" int fmk_h2d_columns(fmk_ctx *ctx, int n_cols, void *const *dst_dev, const void *const *src_host, const size_t *bytes)
{
    if (n_cols < 0 || n_cols > 64) return fmk_set_error(ctx, FMK_E_ARG, "fmk_h2d_columns: %d columns", n_cols);
    size_t total = 0;
    for (int c = 0; c < n_cols; ++c) {
        if (bytes[c] && (!dst_dev[c] || !src_host[c])) return fmk_set_error(ctx, FMK_E_ARG, "fmk_h2d_columns: NULL column %d", c);
        total += bytes[c];
    }
    if (total == 0) return FMK_OK;
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    // what is queued on the context's stream may still use the destination buffers (recycled blocks of the pool)
    FMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int T = up_threads();
    if (total < 2 * UP_CHUNK || T == 0) {                        // the runtime's copy (default; measured at 98 % of the pinned rate)
        // the BLOCKING call: on some boxes hipMemcpyAsync from pageable memory on a stream takes the runtime's single staging
        // buffer (29.9 GB/s measured where hipMemcpy of the same buffer reaches 56.6); the context's stream is idle here anyway
        for (int c = 0; c < n_cols; ++c)
            if (bytes[c]) FMK_HIP(ctx, hipMemcpy(dst_dev[c], src_host[c], bytes[c], hipMemcpyHostToDevice));
        return FMK_OK;
    }
    std::vector<UpItem> items;
    for (int c = 0; c < n_cols; ++c)
        for (size_t off = 0; off < bytes[c]; off += UP_CHUNK)
            items.push_back({c, off, bytes[c] - off < UP_CHUNK ? bytes[c] - off : UP_CHUNK});
    if ((size_t)T > items.size()) T = (int)items.size();
    UpState *st;
    FMK_TRY(up_state(ctx, T, &st));
    std::atomic<size_t> next{0};
    std::atomic<int> err{0};
    const int device = ctx->device;
    auto work = [&](int t) {
        UpWorker &w = st->w[t];
        if (hipSetDevice(device) != hipSuccess) { err.store((int)hipGetLastError()); return; }
        int k = 0;
        for (;;) {
            const size_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= items.size() || err.load(std::memory_order_relaxed)) break;
            const UpItem &it = items[i];
            hipError_t e = hipSuccess;
            if (w.used[k]) e = hipEventSynchronize(w.ev[k]);      // the DMA that last read this staging buffer
            if (e == hipSuccess) {
                memcpy(w.pin[k], (const char *)src_host[it.col] + it.off, it.len);
                e = hipMemcpyAsync((char *)dst_dev[it.col] + it.off, w.pin[k], it.len, hipMemcpyHostToDevice, w.stream);
            }
            if (e == hipSuccess) e = hipEventRecord(w.ev[k], w.stream);
            if (e != hipSuccess) { err.store((int)e); break; }
            w.used[k] = 1;
            k ^= 1;
        }
        const hipError_t e = hipStreamSynchronize(w.stream);
        if (e != hipSuccess) err.store((int)e);
        w.used[0] = w.used[1] = 0;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);                                                      // the calling thread is worker 0
    for (auto &x : th) x.join();
    if (err.load()) return fmk_set_error(ctx, FMK_E_HIP, "fmk_h2d_columns: %s", hipGetErrorString((hipError_t)err.load()));
    return FMK_OK;
}
