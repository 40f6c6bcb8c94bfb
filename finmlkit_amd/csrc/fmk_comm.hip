// fmk_comm.hip -- the multi-GPU side of the time-bar path (BASELINE cfg 5): rank r holds a
// contiguous tick range of one stream; the only data that crosses GPUs is the trailing partial bar of rank r, sent as
// raw ticks to rank r+1 -- ONE neighbour send/recv per step, no collective on the data path.  The reference has no
// distributed code (last row), so there is no reference interface to cite; the entry points are what a
// maintainer binds next to the reducers ("multi-GPU").
//
// Two transports behind the same calls:
//   FMK_COMM_RCCL  ncclSend / ncclRecv of librccl (dlopen'ed on first use: the single-GPU library has no RCCL
//                  dependency), grouped into one launch per step on the communicator's own HIP stream, ordered against
//                  the context's stream by events only -- no host synchronisation inside a step.
//   FMK_COMM_HOST  host-staged: device -> ring buffer in the rendezvous segment -> device.  For the tests (two
//                  processes on ONE GPU, or no GPU at all with ctx == NULL and host pointers) and as the fallback
//                  bench.py reports when RCCL cannot initialise.
// Ranks of one node meet in a shared-memory segment (a file the caller names; rank 0 creates it and unlinks it once
// every rank has attached).  It carries the ncclUniqueId, the small host all-gathers of the set-up phase (timestamps,
// halo lengths, timings) and the ring buffers of the host transport.  Every wait has a deadline -- set-up, the all-gathers,
// the host ring, and the host waits for the communicator's stream (fmk_comm_sync / _destroy poll hipStreamQuery): a missing
// peer is an error code (FMK_E_COMM), never a hang.  What cannot be bounded from here is a wait INSIDE the device queue:
// after fmk_comm_wait_dev the context's stream is ordered behind the exchange, so fmk_ctx_sync on it waits as long as the
// exchange does -- callers that must survive a dead peer call fmk_comm_sync (deadline) BEFORE fmk_ctx_sync.
#include <dlfcn.h>
#include <errno.h>
#include <stdarg.h>
#include <fcntl.h>
#include <sched.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <thread>

#include "fmk_common.h"

namespace {

constexpr uint64_t SEG_MAGIC = 0x464d4b434f4d4d33ULL;       // "FMKCOMM3"
constexpr size_t GATHER_MAX = 4096;                         // bytes per rank and all-gather
constexpr int MAX_WORLD = 64;

struct alignas(64) RankSlot {
    uint64_t gather_seq;                                    // generations this rank has published
    uint64_t attached;
    // ring buffer INTO this rank from its left neighbour (host transport)
    uint64_t produced, consumed;                            // byte counters
    char pad[32];
};

struct SegHeader {
    uint64_t magic;                                         // written last by rank 0
    uint64_t world, ring_bytes, total_bytes;
    uint64_t creator_pid, creator_start;                    // rank 0's pid and start time (/proc/<pid>/stat field 22): ONE process
    uint64_t creator_pidns;                                 // ... and the PID namespace that pid is meaningful in (0: unknown)
    RankSlot slot[MAX_WORLD];
};

// start time of a process in clock ticks since boot; 0 when it does not exist (or /proc is not there)
static uint64_t proc_start_ticks(long pid)
{
    char path[64], buf[1024];
    snprintf(path, sizeof path, "/proc/%ld/stat", pid);
    FILE *f = fopen(path, "r");
    if (!f) return 0;
    const size_t k = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[k] = 0;
    const char *p = strrchr(buf, ')');                      // the command name may contain anything, it ends at the LAST ')'
    if (!p) return 0;
    int field = 2;
    for (++p; *p; ++p)
        if (*p == ' ' && ++field == 22) return strtoull(p + 1, nullptr, 10);
    return 0;
}

// identity of this process' PID namespace (inode of /proc/self/ns/pid); 0 when it cannot be read
static uint64_t pidns_id()
{
    struct stat st;
    return stat("/proc/self/ns/pid", &st) == 0 ? (uint64_t)st.st_ino : 0;
}

static inline size_t gather_off(int world, int parity, int rank)
{
    return sizeof(SegHeader) + ((size_t)parity * world + rank) * GATHER_MAX;
}
static inline size_t ring_off(int world, size_t ring_bytes, int rank)
{
    return sizeof(SegHeader) + (size_t)2 * world * GATHER_MAX + (size_t)rank * ring_bytes;
}

static double now_s()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static inline void relax(int &spins)
{
    if (++spins < 200) return;
    if (spins < 2000) { sched_yield(); return; }
    timespec ts{0, 50000};
    nanosleep(&ts, nullptr);
}

// ---- librccl, resolved at run time -----------------------------------------------------------------------------------
typedef struct { char internal[128]; } nccl_unique_id;      // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void *nccl_comm;
struct Rccl {
    void *so = nullptr;
    int (*GetUniqueId)(nccl_unique_id *) = nullptr;
    int (*CommInitRank)(nccl_comm *, int, nccl_unique_id, int) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    int (*CommAbort)(nccl_comm) = nullptr;               // optional: tear-down without waiting for the peers
    int (*Send)(const void *, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int *) = nullptr;                  // optional
};
static Rccl g_rccl;

static const char *rccl_load()
{
    if (g_rccl.so) return nullptr;
    static char msg[384];
    const char *env = getenv("FMK_RCCL_LIB");
    const char *names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *so = nullptr;
    for (const char *nm : names) {
        if (!nm || !*nm) continue;
        so = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (so) break;
    }
    if (!so) {
        snprintf(msg, sizeof msg, "librccl not found (%s); set FMK_RCCL_LIB", dlerror());
        return msg;
    }
#define SYM(field, name)                                                         \
    do {                                                                         \
        *(void **)(&g_rccl.field) = dlsym(so, name);                             \
        if (!g_rccl.field) {                                                     \
            snprintf(msg, sizeof msg, "librccl lacks %s", name);                 \
            dlclose(so);                                                         \
            return msg;                                                          \
        }                                                                        \
    } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    *(void **)(&g_rccl.CommAbort) = dlsym(so, "ncclCommAbort");
    *(void **)(&g_rccl.GetVersion) = dlsym(so, "ncclGetVersion");
    g_rccl.so = so;
    return nullptr;
}

}  // namespace

struct fmk_comm {
    fmk_ctx *ctx;                   // may be NULL with the host transport (pointers are host pointers then)
    int transport, rank, world, flags;
    double timeout_s;
    char *seg;                      // the mapped rendezvous segment
    size_t seg_bytes, ring_bytes;
    uint64_t gather_gen;
    nccl_comm nccl;
    hipStream_t stream;             // RCCL launches go here
    hipEvent_t ev_ready, ev_done;   // ctx stream -> comm stream, comm stream -> ctx stream
    int exchange_pending;
    // per-exchange timing (fmk_comm_profile_enable): RCCL = event pair on the communicator's stream around the ncclGroup,
    // HOST = wall clock of the staged copy
    int profile_on, profile_n;
    hipEvent_t xev[64][2];
    double host_ms[64];
    char err[512];
};

namespace {

int comm_error(fmk_comm *c, int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof c->err, fmt, ap);
    va_end(ap);
    if (c->ctx) fmk_set_error(c->ctx, code, "%s", c->err);
    else fmk_set_error(nullptr, code, "%s", c->err);
    return code;
}

inline SegHeader *hdr(fmk_comm *c) { return (SegHeader *)c->seg; }

#define NCCL_TRY(c, expr)                                                                                        \
    do {                                                                                                         \
        int r__ = (expr);                                                                                        \
        if (r__ != 0) return comm_error((c), FMK_E_COMM, "%s: %s", #expr, g_rccl.GetErrorString(r__));           \
    } while (0)
#define COMM_HIP(c, expr)                                                                                        \
    do {                                                                                                         \
        hipError_t e__ = (expr);                                                                                 \
        if (e__ != hipSuccess) return comm_error((c), FMK_E_HIP, "%s: %s", #expr, hipGetErrorString(e__));       \
    } while (0)

// One attempt to attach.  *again is set (status FMK_OK) when the file found is a leftover of another run: the caller naps and
// tries again until the deadline -- a loop, not a recursion (at ~1 KB of locals per attempt a 120 s wait would run the
// stack out long before the deadline).
int seg_attach_once(fmk_comm *c, const char *path, size_t ring_bytes, double deadline, bool *again)
{
    const int world = c->world;
    const size_t total = ring_off(world, ring_bytes, world);
    *again = false;
    int fd = -1;
    size_t map_bytes = total;
    if (c->rank == 0) {
        // created and sized under a temporary name, then rename()d into place: a peer can never open a half-made file, and a
        // leftover of a run that died is REPLACED (its inode stays with whoever still has it open -- see the check below)
        char tmp[512];
        snprintf(tmp, sizeof tmp, "%s.%ld.tmp", path, (long)getpid());
        unlink(tmp);
        fd = open(tmp, O_RDWR | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
        if (fd < 0) return comm_error(c, FMK_E_COMM, "rendezvous %s: %s", tmp, strerror(errno));
        if (ftruncate(fd, (off_t)total) != 0) {
            close(fd);
            unlink(tmp);
            return comm_error(c, FMK_E_COMM, "rendezvous %s: ftruncate(%zu): %s", path, total, strerror(errno));
        }
        if (rename(tmp, path) != 0) {
            close(fd);
            unlink(tmp);
            return comm_error(c, FMK_E_COMM, "rendezvous %s: rename: %s", path, strerror(errno));
        }
    } else {
        int spins = 0;
        for (;;) {
            fd = open(path, O_RDWR | O_NOFOLLOW);
            if (fd >= 0) {
                struct stat st;
                if (fstat(fd, &st) == 0 && (size_t)st.st_size >= total) break;
                close(fd);
                fd = -1;
            }
            if (now_s() > deadline)
                return comm_error(c, FMK_E_COMM, "rank %d: rendezvous %s did not appear within %.0f s", c->rank, path,
                                  c->timeout_s);
            relax(spins);
        }
    }
    struct stat sfd;
    fstat(fd, &sfd);
    const ino_t ino = sfd.st_ino;
    const dev_t dev = sfd.st_dev;
    void *m = mmap(nullptr, map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return comm_error(c, FMK_E_COMM, "mmap(%s): %s", path, strerror(errno));
    c->seg = (char *)m;
    c->seg_bytes = map_bytes;
    c->ring_bytes = ring_bytes;
    SegHeader *h = hdr(c);
    if (c->rank == 0) {
        h->world = (uint64_t)world;
        h->ring_bytes = ring_bytes;
        h->total_bytes = total;
        h->creator_pid = (uint64_t)getpid();
        h->creator_start = proc_start_ticks((long)getpid());
        h->creator_pidns = pidns_id();
        __atomic_store_n(&h->magic, SEG_MAGIC, __ATOMIC_RELEASE);
    } else {
        int spins = 0;
        while (__atomic_load_n(&h->magic, __ATOMIC_ACQUIRE) != SEG_MAGIC) {
            if (now_s() > deadline)
                return comm_error(c, FMK_E_COMM, "rank %d: rendezvous %s was never initialised by rank 0", c->rank, path);
            relax(spins);
        }
        if (h->world != (uint64_t)world || h->ring_bytes != ring_bytes)
            return comm_error(c, FMK_E_COMM, "rank %d: rendezvous %s belongs to another job (world %llu, ring %llu)",
                              c->rank, path, (unsigned long long)h->world, (unsigned long long)h->ring_bytes);
        // a leftover of a dead run carries a valid header too (and old gather generations: a rank attached to it would pass
        // its first all-gather on the dead run's data).  Two checks: rank 0 unlinks the name only after EVERY rank has
        // attached, and this rank has not yet -- so the name must still lead to the inode mapped here (a rank 0 that came
        // later has replaced it); and the process that made the file must still be running (pid + start time name one
        // process instance) -- the creator of a leftover is gone.  Either fails: wait for this job's rank 0 and re-attach.
        // The /proc check only means something where the creator's pid does: in the creator's own PID namespace (ranks in
        // different namespaces sharing /dev/shm see no entry, or somebody else's, under that number).  There, no entry or another
        // start time = the creator is gone.  Elsewhere -- or when either side cannot name its namespace -- the inode check decides.
        struct stat sp;
        const uint64_t my_ns = pidns_id();
        const bool same_ns = my_ns != 0 && h->creator_pidns == my_ns && h->creator_start != 0;
        const bool creator_gone = same_ns && proc_start_ticks((long)h->creator_pid) != h->creator_start;
        if (stat(path, &sp) != 0 || sp.st_ino != ino || sp.st_dev != dev || creator_gone) {
            munmap(c->seg, c->seg_bytes);
            c->seg = nullptr;
            if (now_s() > deadline)
                return comm_error(c, FMK_E_COMM, "rank %d: rendezvous %s is a leftover of another run and was not replaced "
                                  "within %.0f s", c->rank, path, c->timeout_s);
            *again = true;
            return FMK_OK;
        }
    }
    __atomic_store_n(&h->slot[c->rank].attached, 1, __ATOMIC_RELEASE);
    return FMK_OK;
}

int seg_attach(fmk_comm *c, const char *path, size_t ring_bytes)
{
    const double deadline = now_s() + c->timeout_s;
    for (;;) {
        bool again = false;
        const int rc = seg_attach_once(c, path, ring_bytes, deadline, &again);
        if (rc != FMK_OK || !again) return rc;
        timespec nap{0, 2000000};
        nanosleep(&nap, nullptr);
    }
}

int gather(fmk_comm *c, const void *send, size_t bytes, void *recv)
{
    if (bytes > GATHER_MAX) return comm_error(c, FMK_E_ARG, "fmk_comm_allgather: %zu bytes per rank (max %zu)", bytes, GATHER_MAX);
    SegHeader *h = hdr(c);
    const uint64_t gen = ++c->gather_gen;
    const int par = (int)(gen & 1);
    // two generations of slots suffice: nobody can publish generation g+2 before everybody has read generation g,
    // because publishing g+2 requires having completed g+1, which requires everybody to have published g+1, which a
    // rank only does after it has read g
    if (bytes) memcpy(c->seg + gather_off(c->world, par, c->rank), send, bytes);
    __atomic_store_n(&h->slot[c->rank].gather_seq, gen, __ATOMIC_RELEASE);
    const double deadline = now_s() + c->timeout_s;
    for (int r = 0; r < c->world; ++r) {
        int spins = 0;
        while (__atomic_load_n(&h->slot[r].gather_seq, __ATOMIC_ACQUIRE) < gen) {
            if (now_s() > deadline)
                return comm_error(c, FMK_E_COMM, "rank %d: rank %d did not reach all-gather %llu within %.0f s", c->rank, r,
                                  (unsigned long long)gen, c->timeout_s);
            relax(spins);
        }
        if (bytes && recv) memcpy((char *)recv + (size_t)r * bytes, c->seg + gather_off(c->world, par, r), bytes);
    }
    return FMK_OK;
}

// copy between a (device or host) user buffer and the host ring
int copy_in(fmk_comm *c, void *ring_dst, const void *user_src, size_t bytes)
{
    if (c->ctx) COMM_HIP(c, hipMemcpy(ring_dst, user_src, bytes, hipMemcpyDeviceToHost));
    else memcpy(ring_dst, user_src, bytes);
    return FMK_OK;
}
int copy_out(fmk_comm *c, void *user_dst, const void *ring_src, size_t bytes)
{
    if (c->ctx) COMM_HIP(c, hipMemcpy(user_dst, ring_src, bytes, hipMemcpyHostToDevice));
    else memcpy(user_dst, ring_src, bytes);
    return FMK_OK;
}

// Host-staged exchange: a byte stream per neighbour pair through the receiver's ring; sending and receiving make
// progress alternately, so a message longer than the ring (or a rank that is its own neighbour) cannot deadlock.
int host_exchange(fmk_comm *c, int n_cols, const void *const *send_ptrs, const size_t *send_bytes, void *const *recv_ptrs,
                  const size_t *recv_bytes, int to, int from)
{
    SegHeader *h = hdr(c);
    if (c->ctx) {
        COMM_HIP(c, hipSetDevice(c->ctx->device));
        COMM_HIP(c, hipStreamSynchronize(c->ctx->stream));  // the buffers are produced / consumed on that stream
    }
    const size_t cap = c->ring_bytes;
    int sc = 0, rc = 0;
    size_t so = 0, ro = 0;                                   // column, offset inside it
    auto skip_empty = [&](int &col, const size_t *bytes) { while (col < n_cols && bytes[col] == 0) ++col; };
    if (to < 0) sc = n_cols;
    if (from < 0) rc = n_cols;
    skip_empty(sc, send_bytes);
    skip_empty(rc, recv_bytes);
    RankSlot *out = to >= 0 ? &h->slot[to] : nullptr;
    RankSlot *in = from >= 0 ? &h->slot[c->rank] : nullptr;
    char *out_ring = to >= 0 ? c->seg + ring_off(c->world, cap, to) : nullptr;
    char *in_ring = from >= 0 ? c->seg + ring_off(c->world, cap, c->rank) : nullptr;
    double deadline = now_s() + c->timeout_s;
    int spins = 0;
    while (sc < n_cols || rc < n_cols) {
        bool progressed = false;
        if (sc < n_cols) {
            const uint64_t prod = __atomic_load_n(&out->produced, __ATOMIC_RELAXED);
            const uint64_t cons = __atomic_load_n(&out->consumed, __ATOMIC_ACQUIRE);
            size_t room = cap - (size_t)(prod - cons);
            if (room) {
                size_t k = send_bytes[sc] - so;
                if (k > room) k = room;
                const size_t pos = (size_t)(prod % cap);
                if (k > cap - pos) k = cap - pos;
                FMK_TRY(copy_in(c, out_ring + pos, (const char *)send_ptrs[sc] + so, k));
                __atomic_store_n(&out->produced, prod + k, __ATOMIC_RELEASE);
                so += k;
                if (so == send_bytes[sc]) { ++sc; so = 0; skip_empty(sc, send_bytes); }
                progressed = true;
            }
        }
        if (rc < n_cols) {
            const uint64_t cons = __atomic_load_n(&in->consumed, __ATOMIC_RELAXED);
            const uint64_t prod = __atomic_load_n(&in->produced, __ATOMIC_ACQUIRE);
            size_t avail = (size_t)(prod - cons);
            if (avail) {
                size_t k = recv_bytes[rc] - ro;
                if (k > avail) k = avail;
                const size_t pos = (size_t)(cons % cap);
                if (k > cap - pos) k = cap - pos;
                FMK_TRY(copy_out(c, (char *)recv_ptrs[rc] + ro, in_ring + pos, k));
                __atomic_store_n(&in->consumed, cons + k, __ATOMIC_RELEASE);
                ro += k;
                if (ro == recv_bytes[rc]) { ++rc; ro = 0; skip_empty(rc, recv_bytes); }
                progressed = true;
            }
        }
        if (progressed) {
            spins = 0;
            deadline = now_s() + c->timeout_s;
        } else {
            if (now_s() > deadline)
                return comm_error(c, FMK_E_COMM, "rank %d: halo exchange stalled for %.0f s (neighbour gone?)", c->rank,
                                  c->timeout_s);
            relax(spins);
        }
    }
    return FMK_OK;
}

int rccl_exchange(fmk_comm *c, int n_cols, const void *const *send_ptrs, const size_t *send_bytes, void *const *recv_ptrs,
                  const size_t *recv_bytes, int to, int from)
{
    fmk_ctx *ctx = c->ctx;
    COMM_HIP(c, hipSetDevice(ctx->device));
    // the communicator's stream starts after everything enqueued so far on the context's stream (the previous step's
    // boundary kernel still reads the receive buffers) -- an event, not a host wait
    COMM_HIP(c, hipEventRecord(c->ev_ready, ctx->stream));
    COMM_HIP(c, hipStreamWaitEvent(c->stream, c->ev_ready, 0));
    const int slot = (c->profile_on && c->profile_n < 64) ? c->profile_n : -1;
    if (slot >= 0) COMM_HIP(c, hipEventRecord(c->xev[slot][0], c->stream));
    NCCL_TRY(c, g_rccl.GroupStart());
    // an error between GroupStart and GroupEnd must not leave the group open for every later call: remember it, close the
    // group, then report
    int r = 0;
    const char *what = "";
    if (to >= 0)
        for (int i = 0; i < n_cols && !r; ++i)
            if (send_bytes[i]) { r = g_rccl.Send(send_ptrs[i], send_bytes[i], /*ncclInt8*/ 0, to, c->nccl, c->stream); what = "ncclSend"; }
    if (from >= 0)
        for (int i = 0; i < n_cols && !r; ++i)
            if (recv_bytes[i]) { r = g_rccl.Recv(recv_ptrs[i], recv_bytes[i], /*ncclInt8*/ 0, from, c->nccl, c->stream); what = "ncclRecv"; }
    const int r2 = g_rccl.GroupEnd();
    if (r) return comm_error(c, FMK_E_COMM, "%s: %s", what, g_rccl.GetErrorString(r));
    if (r2) return comm_error(c, FMK_E_COMM, "ncclGroupEnd: %s", g_rccl.GetErrorString(r2));
    if (slot >= 0) {
        COMM_HIP(c, hipEventRecord(c->xev[slot][1], c->stream));
        c->profile_n = slot + 1;
    }
    COMM_HIP(c, hipEventRecord(c->ev_done, c->stream));
    return FMK_OK;
}

}  // namespace

extern "C" {

const char *fmk_comm_last_error(const fmk_comm *comm) { return comm ? comm->err : fmk_last_error(nullptr); }

int fmk_comm_create(fmk_ctx *ctx, int transport, const char *rendezvous_path, int rank, int world, int flags,
                    size_t ring_bytes, double timeout_s, fmk_comm **out)
{
    *out = nullptr;
    if (world < 1 || world > MAX_WORLD || rank < 0 || rank >= world)
        return fmk_set_error(ctx, FMK_E_ARG, "fmk_comm_create: rank %d of %d (at most %d ranks)", rank, world, MAX_WORLD);
    if (transport != FMK_COMM_RCCL && transport != FMK_COMM_HOST)
        return fmk_set_error(ctx, FMK_E_ARG, "fmk_comm_create: unknown transport %d", transport);
    if (transport == FMK_COMM_RCCL && !ctx)
        return fmk_set_error(ctx, FMK_E_ARG, "fmk_comm_create: the RCCL transport needs a context");
    if (!rendezvous_path || !*rendezvous_path)
        return fmk_set_error(ctx, FMK_E_ARG, "fmk_comm_create: no rendezvous path");
    fmk_comm *c = (fmk_comm *)calloc(1, sizeof(fmk_comm));
    if (!c) return fmk_set_error(ctx, FMK_E_NOMEM, "calloc");
    c->ctx = ctx;
    c->transport = transport;
    c->rank = rank;
    c->world = world;
    c->flags = flags;
    c->timeout_s = timeout_s > 0 ? timeout_s : 120.0;
    if (ring_bytes == 0) ring_bytes = (size_t)1 << 20;
    ring_bytes = (ring_bytes + 4095) & ~(size_t)4095;
    if (transport == FMK_COMM_RCCL) ring_bytes = 4096;       // unused there
    int rc = seg_attach(c, rendezvous_path, ring_bytes);
    auto fail = [&](int code) {
        if (ctx) fmk_set_error(ctx, code, "%s", c->err);
        else fmk_set_error(nullptr, code, "%s", c->err);
        fmk_comm_destroy(c);
        return code;
    };
    if (rc != FMK_OK) return fail(rc);
    nccl_unique_id id;
    memset(&id, 0, sizeof id);
    char why[384] = "";
    int ok = 1;
    if (transport == FMK_COMM_RCCL) {
        const char *e = rccl_load();
        if (e) { ok = 0; snprintf(why, sizeof why, "%s", e); }
        if (ok && rank == 0) {
            int r = g_rccl.GetUniqueId(&id);
            if (r != 0) { ok = 0; snprintf(why, sizeof why, "ncclGetUniqueId: %s", g_rccl.GetErrorString(r)); }
        }
    }
    // everybody learns whether everybody can go on (a rank that cannot load librccl must not leave the others waiting
    // inside ncclCommInitRank), and rank 0's id
    struct { int ok; nccl_unique_id id; } mine, all[MAX_WORLD];
    mine.ok = ok;
    mine.id = id;
    rc = gather(c, &mine, sizeof mine, all);
    if (rc != FMK_OK) return fail(rc);
    if (rank == 0) unlink(rendezvous_path);                  // everybody is attached: the name can go
    for (int r = 0; r < world; ++r)
        if (!all[r].ok) {
            snprintf(c->err, sizeof c->err, "RCCL transport unavailable on rank %d%s%s", r, r == rank ? ": " : "",
                     r == rank ? why : "");
            return fail(FMK_E_COMM);
        }
    if (transport == FMK_COMM_RCCL) {
        hipError_t e = hipSetDevice(ctx->device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming);
        if (e != hipSuccess) {
            snprintf(c->err, sizeof c->err, "comm stream/events: %s", hipGetErrorString(e));
            return fail(FMK_E_HIP);
        }
        // ncclCommInitRank and the first exchange are where a broken fabric shows (IPC handles, P2P mappings): both run under
        // a deadline, and the ranks AGREE on the outcome before anyone returns -- a rank that came through must not be left
        // waiting in a later ncclRecv for a neighbour that has already fallen back to the host transport.
        struct Init { std::atomic<int> done{0}; int rc = 0; nccl_comm comm = nullptr; };
        Init *init = new Init;                                    // leaked on purpose if the call never returns
        const nccl_unique_id uid = all[0].id;
        const int dev = ctx->device;
        std::thread th([init, uid, world, rank, dev]() {
            (void)hipSetDevice(dev);
            init->rc = g_rccl.CommInitRank(&init->comm, world, uid, rank);
            init->done.store(1, std::memory_order_release);
        });
        int my_ok = 1;
        why[0] = 0;
        {
            const double deadline = now_s() + c->timeout_s;
            int spins = 0;
            while (!init->done.load(std::memory_order_acquire) && now_s() < deadline) relax(spins);
            if (!init->done.load(std::memory_order_acquire)) {
                th.detach();
                my_ok = 0;
                snprintf(why, sizeof why, "ncclCommInitRank(rank %d of %d) did not return within %.0f s", rank, world, c->timeout_s);
            } else {
                th.join();
                if (init->rc != 0) {
                    my_ok = 0;
                    snprintf(why, sizeof why, "ncclCommInitRank(rank %d of %d): %s", rank, world, g_rccl.GetErrorString(init->rc));
                } else c->nccl = init->comm;
                delete init;
            }
        }
        // one 8-byte exchange with the neighbours the step will talk to
        const bool loop = (flags & FMK_COMM_SELF_LOOP) && world == 1;
        const int to = loop ? 0 : (rank + 1 < world ? rank + 1 : -1), from = loop ? 0 : (rank > 0 ? rank - 1 : -1);
        void *probe = nullptr;
        if (my_ok && (to >= 0 || from >= 0)) {
            hipError_t e2 = hipMalloc(&probe, 16);
            int r = 0;
            if (e2 == hipSuccess) {
                r = g_rccl.GroupStart();
                if (!r && to >= 0) r = g_rccl.Send(probe, 8, 0, to, c->nccl, c->stream);
                if (!r && from >= 0) r = g_rccl.Recv((char *)probe + 8, 8, 0, from, c->nccl, c->stream);
                const int r2 = g_rccl.GroupEnd();
                if (!r) r = r2;
            }
            if (e2 != hipSuccess || r != 0) {
                my_ok = 0;
                snprintf(why, sizeof why, "first RCCL exchange (rank %d): %s", rank,
                         e2 != hipSuccess ? hipGetErrorString(e2) : g_rccl.GetErrorString(r));
            } else {
                const double deadline = now_s() + c->timeout_s;
                int spins = 0;
                hipError_t q;
                while ((q = hipStreamQuery(c->stream)) == hipErrorNotReady && now_s() < deadline) relax(spins);
                if (q != hipSuccess) {
                    my_ok = 0;
                    snprintf(why, sizeof why, "first RCCL exchange (rank %d) %s", rank,
                             q == hipErrorNotReady ? "did not complete in time" : hipGetErrorString(q));
                }
            }
        }
        int oks[MAX_WORLD];
        rc = gather(c, &my_ok, sizeof my_ok, oks);
        if (rc != FMK_OK) return fail(rc);
        for (int r = 0; r < world; ++r)
            if (!oks[r]) {
                snprintf(c->err, sizeof c->err, "RCCL transport failed on rank %d%s%s", r, r == rank ? ": " : "", r == rank ? why : "");
                // what may be stuck is left behind, not destroyed: ncclCommDestroy can wait for the peers, and a stream with an
                // exchange that never completes cannot be synchronised (fmk_comm_destroy does both)
                c->nccl = nullptr;
                if (!my_ok) c->stream = nullptr;
                if (my_ok && probe) (void)hipFree(probe);
                return fail(FMK_E_COMM);
            }
        if (probe) (void)hipFree(probe);
    }
    *out = c;
    return FMK_OK;
}

// Host wait for the communicator's stream with the communicator's deadline: a peer that died after set-up leaves an
// ncclRecv that never completes, and hipStreamSynchronize on it would hang the survivor for good.
static int comm_stream_wait(fmk_comm *c)
{
    if (!c->stream) return FMK_OK;
    const double deadline = now_s() + c->timeout_s;
    int spins = 0;
    hipError_t q;
    while ((q = hipStreamQuery(c->stream)) == hipErrorNotReady) {
        if (now_s() > deadline) return FMK_E_COMM;
        relax(spins);
    }
    return q == hipSuccess ? FMK_OK : FMK_E_HIP;
}

int fmk_comm_destroy(fmk_comm *c)
{
    if (!c) return FMK_OK;
    if (c->ctx) (void)hipSetDevice(c->ctx->device);
    int stuck = 0;
    if (c->stream && comm_stream_wait(c) == FMK_E_COMM) stuck = 1;
    if (c->nccl) {
        // a communicator with an exchange that will never complete is aborted (does not wait for the peers) or, when this
        // librccl has no ncclCommAbort, left behind -- ncclCommDestroy would block on the dead neighbour
        if (!stuck) (void)g_rccl.CommDestroy(c->nccl);
        else if (g_rccl.CommAbort) (void)g_rccl.CommAbort(c->nccl);
    }
    for (int i = 0; i < 64; ++i) {
        if (c->xev[i][0]) (void)hipEventDestroy(c->xev[i][0]);
        if (c->xev[i][1]) (void)hipEventDestroy(c->xev[i][1]);
    }
    if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->stream && !stuck) (void)hipStreamDestroy(c->stream);     // a stuck stream is abandoned, not waited for
    if (c->seg) munmap(c->seg, c->seg_bytes);
    free(c);
    return stuck ? FMK_E_COMM : FMK_OK;
}

int fmk_comm_allgather(fmk_comm *c, const void *send, size_t bytes, void *recv) { return gather(c, send, bytes, recv); }

int fmk_comm_barrier(fmk_comm *c) { return gather(c, nullptr, 0, nullptr); }

int fmk_comm_halo_exchange_dev(fmk_comm *c, int n_cols, const void *const *send_ptrs, const size_t *send_bytes,
                               void *const *recv_ptrs, const size_t *recv_bytes)
{
    if (n_cols < 0 || n_cols > 16) return comm_error(c, FMK_E_ARG, "fmk_comm_halo_exchange_dev: %d columns", n_cols);
    const bool loop = (c->flags & FMK_COMM_SELF_LOOP) && c->world == 1;
    const int to = loop ? 0 : (c->rank + 1 < c->world ? c->rank + 1 : -1);
    const int from = loop ? 0 : (c->rank > 0 ? c->rank - 1 : -1);
    if (to < 0 && from < 0) return FMK_OK;
    if (c->transport == FMK_COMM_HOST) {
        const double t0 = now_s();
        FMK_TRY(host_exchange(c, n_cols, send_ptrs, send_bytes, recv_ptrs, recv_bytes, to, from));
        if (c->profile_on && c->profile_n < 64) c->host_ms[c->profile_n++] = (now_s() - t0) * 1e3;
        return FMK_OK;
    }
    FMK_TRY(rccl_exchange(c, n_cols, send_ptrs, send_bytes, recv_ptrs, recv_bytes, to, from));
    c->exchange_pending = 1;
    return FMK_OK;
}

int fmk_comm_wait_dev(fmk_comm *c)
{
    if (c->transport != FMK_COMM_RCCL || !c->exchange_pending) return FMK_OK;
    COMM_HIP(c, hipSetDevice(c->ctx->device));
    COMM_HIP(c, hipStreamWaitEvent(c->ctx->stream, c->ev_done, 0));   // the context's stream continues after the exchange
    c->exchange_pending = 0;
    return FMK_OK;
}

int fmk_comm_sync(fmk_comm *c)
{
    if (c->stream) {
        COMM_HIP(c, hipSetDevice(c->ctx->device));
        const int rc = comm_stream_wait(c);
        if (rc == FMK_E_COMM)
            return comm_error(c, FMK_E_COMM, "rank %d: the halo exchange did not complete within %.0f s (neighbour gone?)",
                              c->rank, c->timeout_s);
        if (rc != FMK_OK) return comm_error(c, FMK_E_HIP, "communicator stream: %s", hipGetErrorString(hipGetLastError()));
    }
    return FMK_OK;
}

int fmk_comm_profile_enable(fmk_comm *c, int on)
{
    if (on && c->transport == FMK_COMM_RCCL && !c->xev[0][0]) {
        COMM_HIP(c, hipSetDevice(c->ctx->device));
        for (int i = 0; i < 64; ++i) {
            COMM_HIP(c, hipEventCreate(&c->xev[i][0]));
            COMM_HIP(c, hipEventCreate(&c->xev[i][1]));
        }
    }
    c->profile_on = on ? 1 : 0;
    c->profile_n = 0;
    return FMK_OK;
}

int fmk_comm_profile_read(fmk_comm *c, double *ms, int capacity, int *count)
{
    int n = c->profile_n < 64 ? c->profile_n : 64;
    if (n > capacity) n = capacity;
    if (c->transport == FMK_COMM_RCCL) {
        FMK_TRY(fmk_comm_sync(c));
        for (int i = 0; i < n; ++i) {
            float t = 0.f;
            COMM_HIP(c, hipEventElapsedTime(&t, c->xev[i][0], c->xev[i][1]));
            ms[i] = (double)t;
        }
    } else {
        for (int i = 0; i < n; ++i) ms[i] = c->host_ms[i];
    }
    *count = n;
    return FMK_OK;
}

}  // extern "C"

// ---- column helper of the sharded step: up to 8 small column slices copied by ONE launch ---------------------------
struct FmkCopyCols {
    const char *src[8];
    char *dst[8];
    unsigned long long bytes[8];
};

__global__ void k_copy_cols(FmkCopyCols c, int n_cols)
{
    const int col = blockIdx.y;
    if (col >= n_cols) return;
    const unsigned long long nb = c.bytes[col];
    const char *s = c.src[col];
    char *d = c.dst[col];
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < nb;
         i += (unsigned long long)gridDim.x * blockDim.x)
        d[i] = s[i];
}

extern "C" int fmk_copy_cols_dev(fmk_ctx *ctx, int n_cols, const void *const *src, void *const *dst, const size_t *bytes)
{
    if (n_cols < 0 || n_cols > 8) return fmk_set_error(ctx, FMK_E_ARG, "fmk_copy_cols_dev: %d columns (max 8)", n_cols);
    FmkCopyCols c;
    size_t mx = 0;
    for (int i = 0; i < n_cols; ++i) {
        c.src[i] = (const char *)src[i];
        c.dst[i] = (char *)dst[i];
        c.bytes[i] = bytes[i];
        if (bytes[i] > mx) mx = bytes[i];
    }
    if (mx == 0) return FMK_OK;
    FMK_HIP(ctx, hipSetDevice(ctx->device));
    size_t bx = (mx + 255) / 256;
    if (bx > 1024) bx = 1024;
    k_copy_cols<<<dim3((unsigned)bx, (unsigned)n_cols), 256, 0, ctx->stream>>>(c, n_cols);
    FMK_LAUNCH_CHECK(ctx);
    return FMK_OK;
}


// First contact with a node: what this process sees of it, as text -- the devices, which of them can map each
// other's memory (hipDeviceCanAccessPeer: what RCCL's point-to-point path over xGMI needs), the librccl this library would load and
// its version, and the environment that decides how RCCL shares memory between processes.  No context, no communicator, no device
// memory: it works (and says so) on a box without a GPU or without librccl.  `python -m finmlkit_amd.dist --selftest` prints it.
extern "C" int fmk_comm_describe(char *buf, size_t cap)
{
    if (!buf || cap < 64) return FMK_E_ARG;
    size_t at = 0;
    auto put = [&](const char *fmt, ...) {
        if (at >= cap - 1) return;
        va_list ap;
        va_start(ap, fmt);
        const int k = vsnprintf(buf + at, cap - at, fmt, ap);
        va_end(ap);
        if (k > 0) at += (size_t)k < cap - at ? (size_t)k : cap - at - 1;
    };
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { put("devices: hipGetDeviceCount: %s\n", hipGetErrorString(e)); n = 0; }
    else put("devices: %d\n", n);
    for (int i = 0; i < n && i < 16; ++i) {
        hipDeviceProp_t pr;
        char bus[32] = "?";
        if (hipGetDeviceProperties(&pr, i) != hipSuccess) { put("  [%d] hipGetDeviceProperties failed\n", i); continue; }
        (void)hipDeviceGetPCIBusId(bus, (int)sizeof bus, i);
        put("  [%d] %s %s, %d CUs, %.0f GiB, pci %s\n", i, pr.name, pr.gcnArchName, pr.multiProcessorCount,
            (double)pr.totalGlobalMem / (double)(1ull << 30), bus);
    }
    if (n > 1) {
        put("peer access (row can map column):\n");
        for (int i = 0; i < n && i < 16; ++i) {
            put("  [%d]", i);
            for (int j = 0; j < n && j < 16; ++j) {
                int ok = 0;
                if (i == j) { put(" ."); continue; }
                const hipError_t pe = hipDeviceCanAccessPeer(&ok, i, j);
                put(pe == hipSuccess ? (ok ? " 1" : " 0") : " ?");
            }
            put("\n");
        }
    }
    const char *why = rccl_load();
    if (why) put("librccl: NOT LOADED: %s\n", why);
    else {
        Dl_info di;
        const char *path = (dladdr((void *)g_rccl.Send, &di) && di.dli_fname) ? di.dli_fname : "?";
        int v = 0;
        if (g_rccl.GetVersion && g_rccl.GetVersion(&v) == 0) put("librccl: %s, ncclGetVersion %d (%d.%d.%d)\n", path, v, v / 10000, v / 100 % 100, v % 100);
        else put("librccl: %s (no ncclGetVersion)\n", path);
    }
    const char *keys[] = {"HSA_ENABLE_IPC_MODE_LEGACY", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "NCCL_DEBUG",
                          "NCCL_P2P_DISABLE", "NCCL_SHM_DISABLE", "NCCL_SOCKET_IFNAME", "RCCL_MSCCL_ENABLE", "FMK_RCCL_LIB", "FMK_DEVICE"};
    put("environment:");
    for (const char *k : keys) {
        const char *v = getenv(k);
        if (v) put(" %s=%s", k, v);
    }
    put("\n");
    if (!getenv("HSA_ENABLE_IPC_MODE_LEGACY") || strcmp(getenv("HSA_ENABLE_IPC_MODE_LEGACY"), "0") != 0)
        put("note: HSA_ENABLE_IPC_MODE_LEGACY is not 0 -- on hosts whose driver only supports dmabuf IPC, RCCL between processes fails with "
            "hipIpcGetMemHandle: invalid argument\n");
    return FMK_OK;
}
