// fmk_cusum_onepass.h -- the dense regime of _cusum_bar_indexer (finmlkit/bar/logic.py:152-221) in ONE pass over the columns.
// Included by fmk_cusum.hip (CsState, cs_lane, CS_* and the expressions of k_cusum_prep live there).
//
// The fixed point of fmk_cusum.hip stores the loop inputs chunk-transposed (16 B/tick written, then read by every round and once more
// by the pass that writes the closes): 88 GB for a 24 B/tick input at 1e9 ticks, 18.6 ms.  What makes that fixed point converge in two
// rounds -- the clamps and the resets erase the incoming state within a few hundred ticks -- also says that the walk of a chunk from
// the WRONG state (0, 0) and its walk from the TRUE state are the same walk from the tick on at which their states are equal: a state
// is two float64 that are never NaN and never -0.0 (every zero is the literal 0.0 of a clamp or a reset), so bitwise equality of the
// states is equality of everything that follows.  Hence:
//   pass A (k_cs1_pass)  : a workgroup owns 32 chunks of CS1_L ticks and goes through them 64 ticks at a time: all threads compute the
//                          returns / thresholds of the 32 x 64 tile from the raw columns (the expressions of k_cusum_prep, 512
//                          contiguous bytes per wave instruction and column) into LDS, then 32 lanes walk one chunk each through
//                          those 64 ticks (the loop of k_cusum_chunks).  The walk starts from (0, 0) CS1_W ticks IN FRONT of the
//                          chunk: on a tape that forgets, the state it reaches the chunk with (E) is already the true one for most
//                          chunks, and for those nothing is left to do.  From the chunk's first tick on it keeps the number of
//                          closes and the closes themselves as 16-bit offsets in the chunk's own staging row (CS1_L slots: a close
//                          on every tick fits, there is no overflow case), and the exit state.  Nothing per tick is written.
//   fix-up (k_cs1_fix)   : one wave per chunk whose true entry state (the exit state of the chunk before it) is not the one its
//                          record was made from.  Lane 0 walks from the true entry state, lane 1 from E, in lockstep over the same
//                          LDS rows (ret / lam recomputed from the raw columns
//                          with the same expressions => the same bits => lane 1 IS pass A's walk); every 8 ticks the two states are
//                          compared, and at the first boundary where they are equal the chunk is settled: its closes are lane 0's
//                          up to there (patch row) and pass A's from there on, its exit state is pass A's.  A chunk that does not
//                          merge before its end gets lane 0's exit state, which changes its successor's entry state: the launch is
//                          repeated until no exit state changes (entry states are read from a copy made before the launch).
//   emit (k_cs1_emit)    : counts = staged - (lane 1's closes before the merge) + (lane 0's) -> exclusive scan -> a wave per chunk
//                          writes patch row then the rest of the staging row as int64 tick indices.
// The first fix-up launch gives every chunk CS1_FIRST_LIMIT ticks to merge (the rest get their whole length in the next launch); if
// more than a quarter of the chunks have not merged by then the tape does not forget fast enough for this form and the caller runs
// the fixed point instead (no result of this file is used).
#pragma once

#define CS1_L 4096                 // ticks per chunk (16-bit offsets: <= 65536)
#define CS1_W 512                  // warm-up: pass A starts walking this many ticks BEFORE the chunk (a multiple of CS1_TJ)
#define CS1_TK 32                  // chunks per workgroup
#define CS1_TJ 64                  // ticks per tile
#define CS1_FIRST_LIMIT 512        // ticks a chunk may take to merge in the first fix-up launch
#define CS1_PREFETCH 0             // 1: the next tile's loads are issued before the walk (measured: 7.2 against 6.8 ms -- 173 VGPRs, a spill at three waves)

// ret / lam of tick i = first + 1 + t (the expressions of k_cusum_prep, operation for operation)
__device__ __forceinline__ void cs1_inputs(const int64_t *__restrict__ ts, const double *__restrict__ price,
                                           const double *__restrict__ sigma, int64_t n, int64_t i, double sigma_floor,
                                           double sigma_mult, double &r, double &lam, bool &nan_sigma)
{
    r = fmk_log_ratio(price[i], price[i - 1]);
    const bool block = i + 1 < n && ts[i] == ts[i + 1];
    const double sg = sigma[i];
    nan_sigma |= sg != sg;
    lam = NAN;
    if (!block) {
        lam = sigma_mult * sg;
        lam = sigma_floor > lam ? sigma_floor : lam;                     // max(lam, floor): a NaN lam stays NaN
    }
}

__global__ __launch_bounds__(256) void k_cs1_pass(const int64_t *__restrict__ ts, const double *__restrict__ price,
                                                  const double *__restrict__ sigma, int64_t n, int64_t first, int64_t m,
                                                  int64_t chunks, double sigma_floor, double sigma_mult,
                                                  CsState *__restrict__ E, CsState *__restrict__ S0, int *__restrict__ C0,
                                                  unsigned short *__restrict__ staged,
                                                  unsigned long long *nan_flag /* null: sigma has been forward filled */)
{
    __shared__ double s_r[CS1_TK][CS1_TJ + 1];
    __shared__ double s_l[CS1_TK][CS1_TJ + 1];
    const int64_t k0 = (int64_t)blockIdx.x * CS1_TK;
    const int col = threadIdx.x & (CS1_TJ - 1), row4 = threadIdx.x / CS1_TJ;
    constexpr int RP = 256 / CS1_TJ;                                     // rows per pass of the load phase
    constexpr int NR = CS1_TK / RP;                                      // rows per thread and tile
    const bool walker = threadIdx.x < CS1_TK;
    const int64_t kw = k0 + threadIdx.x;                                 // the walker's chunk
    double sp = 0.0, sn = 0.0;
    int cnt = 0;
    unsigned short *my = staged + (walker && kw < chunks ? kw : 0) * (int64_t)CS1_L;
    bool nan_sigma = false;
    // raw inputs of one tile: all loads first, the arithmetic when they are needed (the guard inside the row loop once made every
    // row load -> wait -> logarithm -> store on its own: ~17 us per tile)
    // price[i - 1] and ts[i + 1] are the neighbouring lanes' own loads except at the two ends of the 64-tick row: one packed halo word
    // per row (lane 0: the bits of price[i - 1], lane 63: ts[i + 1]) instead of two more arrays held across the walk
    double p[NR], sg[NR];
    int64_t tsi[NR];
    unsigned long long halo[NR];
    unsigned okm = 0;
    auto fetch = [&](int j0) {
        okm = 0;
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            const int row = rr * RP + row4;
            const int64_t t = (k0 + row) * CS1_L + j0 + col;
            const bool ok = j0 < CS1_L && k0 + row < chunks && t >= 0 && t < m;
            okm |= (ok ? 1u : 0u) << rr;
            const int64_t i = first + 1 + (ok ? t : 0);                  // (m > 0: first + 1 is a tick of the stream)
            p[rr] = price[i]; sg[rr] = sigma[i]; tsi[rr] = ts[i];
            const unsigned long long *ha = col == 0 ? (const unsigned long long *)(price + i - 1)
                                                    : (const unsigned long long *)(ts + (i + 1 < n ? i + 1 : i));
            halo[rr] = 0;
            if (col == 0 || col == CS1_TJ - 1) halo[rr] = *ha;
        }
    };
    if (CS1_PREFETCH) fetch(-CS1_W);
    // j0 < 0: the warm-up -- the same walk over the CS1_W ticks in front of the chunk (ticks before the stream's first change nothing),
    // closes not recorded; the state it arrives with at j0 == 0 is the chunk's entry state E, the walk from there is the chunk's record
    for (int j0 = -CS1_W; j0 < CS1_L; j0 += CS1_TJ) {
        if (!CS1_PREFETCH) fetch(j0);
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            const int row = rr * RP + row4;
            // (a valid tick's neighbours inside the row are valid ticks: t = 0 sits at lane 0, and a tick whose successor is past the
            //  stream has i + 1 == n, where the block test is false whatever the shifted value is)
            const double up = __shfl_up(p[rr], 1, 64);
            const int64_t dn = __shfl_down(tsi[rr], 1, 64);
            const double pm = col == 0 ? __longlong_as_double((long long)halo[rr]) : up;
            const int64_t tsn = col == CS1_TJ - 1 ? (int64_t)halo[rr] : dn;
            double r = 0.0, lam = NAN;                                   // outside the stream: a tick that changes nothing
            if ((okm >> rr) & 1u) {                                      // the expressions of k_cusum_prep, operation for operation
                const int64_t i = first + 1 + (k0 + row) * CS1_L + j0 + col;
                r = fmk_log_ratio(p[rr], pm);
                const bool block = i + 1 < n && tsi[rr] == tsn;
                nan_sigma |= sg[rr] != sg[rr];
                if (!block) {
                    lam = sigma_mult * sg[rr];
                    lam = sigma_floor > lam ? sigma_floor : lam;         // max(lam, floor): a NaN lam stays NaN
                }
            }
            s_r[row][col] = r;
            s_l[row][col] = lam;
        }
        if (CS1_PREFETCH) fetch(j0 + CS1_TJ);                            // the next tile's loads fly while this one is walked
        __syncthreads();
        if (walker && kw < chunks) {
            if (j0 == 0) { E[kw].sp = sp; E[kw].sn = sn; }
            // The loop of k_cusum_chunks (logic.py:199-219) as selects, 64 ticks in ONE basic block: the closes go into a bit mask and
            // are stored after the tile.  (With the store inside the loop every tick was its own block: LDS reads, then the chain,
            // then a branch.)
            unsigned long long mask = 0;
#pragma unroll
            for (int j8 = 0; j8 < CS1_TJ; j8 += 8) {
                double r8[8], l8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { r8[q] = s_r[threadIdx.x][j8 + q]; l8[q] = s_l[threadIdx.x][j8 + q]; }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const double a = sp + r8[q], b = sn + r8[q];
                    sp = a > 0.0 ? a : 0.0;                              // max(0.0, s_pos + ret): NaN -> 0.0
                    sn = b < 0.0 ? b : 0.0;                              // min(0.0, s_neg + ret)
                    const bool cp = sp >= l8[q];
                    const bool cn = !cp && sn <= -l8[q];
                    mask |= (unsigned long long)((cp | cn) ? 1 : 0) << (j8 + q);
                    sp = cp ? 0.0 : sp;
                    sn = cn ? 0.0 : sn;
                }
            }
            if (j0 >= 0)
                while (mask) {
                    my[cnt++] = (unsigned short)(j0 + __builtin_ctzll(mask));
                    mask &= mask - 1;
                }
        }
        __syncthreads();
    }
    if (nan_flag && __builtin_amdgcn_ballot_w64(nan_sigma) != 0 && fmk_lane() == 0 && __atomic_load_n(nan_flag, __ATOMIC_RELAXED) == 0)
        atomicOr(nan_flag, 1ULL);
    if (walker && kw < chunks) {
        S0[kw].sp = sp; S0[kw].sn = sn;
        C0[kw] = cnt;
    }
}

// per-chunk record of the fix-up: how many of pass A's closes lie before the merge, how many closes the true walk has there
struct Cs1Fix { int pfx_a, pfx_t; };

// One wave per chunk k >= 1.  S_read: the exit states as they were before this launch (entry state of k = S_read[k - 1]);
// S: the exit states this launch writes; last_in[k]: the entry state chunk k's record was made from (E[k] after pass A).
//   limit   : ticks the walk may take before it gives the chunk up for this launch (`pending` counts those)
//   changed : chunks whose exit state differs from S_read[k]
__global__ __launch_bounds__(256) void k_cs1_fix(const int64_t *__restrict__ ts, const double *__restrict__ price,
                                                 const double *__restrict__ sigma, int64_t n, int64_t first, int64_t m,
                                                 int64_t chunks, double sigma_floor, double sigma_mult,
                                                 const CsState *__restrict__ E, const CsState *__restrict__ S0,
                                                 const CsState *__restrict__ S_read,
                                                 CsState *__restrict__ S, CsState *__restrict__ last_in,
                                                 Cs1Fix *__restrict__ fix, unsigned short *__restrict__ patch, int limit,
                                                 unsigned long long *changed, unsigned long long *pending)
{
    __shared__ double s_r[4][64], s_l[4][64];
    const int lane = fmk_lane();
    const int wib = (int)(threadIdx.x >> 6);
    const int64_t k = (int64_t)blockIdx.x * 4 + wib + 1;
    if (k >= chunks) return;
    CsState in = S_read[k - 1];
    in.sp = cs_lane(in.sp, 0); in.sn = cs_lane(in.sn, 0);               // one copy for the whole wave
    {
        CsState li = last_in[k];
        li.sp = cs_lane(li.sp, 0); li.sn = cs_lane(li.sn, 0);
        if (cs_same(li, in)) return;                                     // the record was made from exactly this entry state
    }
    // lane 0: the true walk; lane 1: pass A's walk (from the chunk's warmed-up entry state); the other lanes carry lane 1's values
    // and are never looked at
    CsState e = E[k];
    e.sp = cs_lane(e.sp, 0); e.sn = cs_lane(e.sn, 0);
    double sp = lane == 0 ? in.sp : e.sp, sn = lane == 0 ? in.sn : e.sn;
    int cnt = 0;
    const int64_t t0 = k * CS1_L;
    const int len = (int)(m - t0 < CS1_L ? m - t0 : CS1_L);
    unsigned short *my = patch + k * (int64_t)CS1_L;
    bool merged = false, gave_up = false;
    // raw inputs one group of 64 ticks ahead
    double c_p, c_pm, c_sg; int64_t c_ts, c_tsn;
    auto fetch = [&](int j0, double &p, double &pm, double &sg, int64_t &tsi, int64_t &tsn) {
        int64_t i = first + 1 + t0 + j0 + lane;
        if (i > n - 1) i = n - 1;                                        // lanes past the chunk: any valid address
        p = price[i]; pm = price[i - 1]; sg = sigma[i]; tsi = ts[i];
        tsn = ts[i + 1 < n ? i + 1 : i];
    };
    fetch(0, c_p, c_pm, c_sg, c_ts, c_tsn);
    int j0 = 0;
    for (; j0 < len; j0 += 64) {
        if (j0 >= limit) { gave_up = true; break; }
        double n_p = 1.0, n_pm = 1.0, n_sg = 0.0; int64_t n_ts = 0, n_tsn = 1;
        if (j0 + 64 < len) fetch(j0 + 64, n_p, n_pm, n_sg, n_ts, n_tsn);
        {
            const int jj = j0 + lane;
            double r = 0.0, lam = NAN;
            if (jj < len) {                                              // the expressions of k_cusum_prep / cs1_inputs
                const int64_t i = first + 1 + t0 + jj;
                r = fmk_log_ratio(c_p, c_pm);
                const bool block = i + 1 < n && c_ts == c_tsn;
                if (!block) {
                    lam = sigma_mult * c_sg;
                    lam = sigma_floor > lam ? sigma_floor : lam;
                }
            }
            s_r[wib][lane] = r;
            s_l[wib][lane] = lam;
            c_p = n_p; c_pm = n_pm; c_sg = n_sg; c_ts = n_ts; c_tsn = n_tsn;
        }
        __builtin_amdgcn_wave_barrier();
        const int lim = len - j0 < 64 ? len - j0 : 64;
        int q8 = 0;
        for (; q8 < lim; q8 += 8) {
            // eight ticks in one basic block on every lane (rows past the chunk's end hold ret = 0, lam = NaN: ticks that change
            // nothing); closes as a bit mask, lane 0 stores its own after the batch
            double r8[8], l8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { r8[q] = s_r[wib][q8 + q]; l8[q] = s_l[wib][q8 + q]; }
            unsigned m8 = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {                                // the loop of k_cusum_chunks, as selects
                const double a = sp + r8[q], b = sn + r8[q];
                sp = a > 0.0 ? a : 0.0;
                sn = b < 0.0 ? b : 0.0;
                const bool cp = sp >= l8[q];
                const bool cn = !cp && sn <= -l8[q];
                m8 |= ((cp | cn) ? 1u : 0u) << q;
                sp = cp ? 0.0 : sp;
                sn = cn ? 0.0 : sn;
            }
            if (lane == 0) {
                int at = cnt;
                for (unsigned mm = m8; mm; mm &= mm - 1) my[at++] = (unsigned short)(j0 + q8 + __builtin_ctz(mm));
            }
            cnt += __builtin_popcount(m8);
            const double tp = cs_lane(sp, 0), tn = cs_lane(sn, 0), ap = cs_lane(sp, 1), an = cs_lane(sn, 1);
            if (__double_as_longlong(tp) == __double_as_longlong(ap) && __double_as_longlong(tn) == __double_as_longlong(an)) {
                merged = true;
                break;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (merged) break;
    }
    if (gave_up) {                                                       // not settled in this launch: the record stays as it was
        if (lane == 0) atomicAdd(pending, 1ULL);
        return;
    }
    const int cnt_t = __builtin_amdgcn_readlane(cnt, 0), cnt_a = __builtin_amdgcn_readlane(cnt, 1);
    CsState out;
    if (merged) out = S0[k];                                             // from the merge on the walk is pass A's
    else { out.sp = cs_lane(sp, 0); out.sn = cs_lane(sn, 0); }           // lane 0 reached the end of the chunk on its own
    out.sp = cs_lane(out.sp, 0); out.sn = cs_lane(out.sn, 0);
    if (lane == 0) {
        fix[k].pfx_a = cnt_a; fix[k].pfx_t = cnt_t;                      // (no merge: lane 1 walked the whole chunk, cnt_a == C0[k])
        last_in[k] = in;
        const CsState was = S_read[k];
        S[k] = out;
        if (!cs_same(was, out)) atomicAdd(changed, 1ULL);
    }
}

__global__ __launch_bounds__(256) void k_cs1_counts(const int *__restrict__ C0, const Cs1Fix *__restrict__ fix, int64_t chunks,
                                                    int64_t *__restrict__ counts)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= chunks) return;
    counts[k] = (int64_t)C0[k] - fix[k].pfx_a + fix[k].pfx_t;
}

// a wave per chunk: the true walk's closes before the merge (patch row), then pass A's from its pfx_a-th on (staging row)
__global__ __launch_bounds__(256) void k_cs1_emit(const int *__restrict__ C0, const Cs1Fix *__restrict__ fix,
                                                  const unsigned short *__restrict__ staged,
                                                  const unsigned short *__restrict__ patch, int64_t chunks, int64_t first,
                                                  const int64_t *__restrict__ offsets, int64_t *__restrict__ closes)
{
    const int lane = fmk_lane();
    const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= chunks) return;
    const int pa = fix[k].pfx_a, pt = fix[k].pfx_t, c0 = C0[k];
    const int64_t base = first + 1 + k * (int64_t)CS1_L;
    int64_t *dst = closes + offsets[k];
    const unsigned short *pp = patch + k * (int64_t)CS1_L, *ss = staged + k * (int64_t)CS1_L;
    for (int e = lane; e < pt; e += 64) dst[e] = base + pp[e];
    for (int e = lane; e < c0 - pa; e += 64) dst[pt + e] = base + ss[pa + e];
}
