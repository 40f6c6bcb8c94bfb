// fmk_pairwise.h -- NumPy's pairwise summation, float32 or float64 (the element type is what `load` returns) (numpy/_core/src/umath/loops_utils.h.src, pairwise_sum), evaluated
// by one wave with a wave-uniform result.  The reference's float32 reductions go through it (np.sum / np.mean of a float32
// slice, `.sum()` of float32 level totals), and float32 addition is not associative, so bit-identical outputs need the
// same tree:  n < 8: a plain loop;  n <= 128: eight accumulators over the elements 8k + i, folded as
// ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)), then the n % 8 tail;  n > 128: halves of n/2 rounded down to a multiple of 8.
// `load(i)` returns element i (an LDS array, a slice of a global column, a transformed value ...).
#pragma once
#include "fmk_common.h"

template <class F>
__device__ __forceinline__ auto fmk_pw_leaf(F load, int off, int n, int lane) -> decltype(load(0))
{
    typedef decltype(load(0)) T;
    if (n < 8) {
        T r = 0;
        for (int i = 0; i < n; ++i) r += load(off + i);
        return r;
    }
    const int nm = n - (n & 7);
    // The eight accumulators r_i = ((x_i + x_{8+i}) + x_{16+i}) + ... live in lanes 0..7.  All 64 lanes fetch (and transform)
    // the leaf's <= 128 elements with two coalesced loads; the chains then take their terms from the other lanes' registers in
    // NumPy's order.  (A first version let lanes 0..7 load their 16 elements one after the other: sixteen dependent global
    // loads per leaf, 24.6 ms per 1e9 ticks for comp_bar_trade_size_features on 1 200-tick bars.)
    const T v0 = lane < n ? load(off + lane) : (T)0;                   // (the n % 8 tail elements too: no dependent loads below)
    const T v1 = 64 + lane < n ? load(off + 64 + lane) : (T)0;
    const int i8 = lane & 7;
    T r = __shfl(v0, i8, 64);
    for (int k = 1; 8 * k < nm; ++k) r += k < 8 ? __shfl(v0, 8 * k + i8, 64) : __shfl(v1, 8 * (k - 8) + i8, 64);
    T t = r + __shfl_down(r, 1, 64);         // lanes 0,2,4,6: r0+r1, r2+r3, r4+r5, r6+r7
    T u = t + __shfl_down(t, 2, 64);         // lanes 0,4
    T res = __shfl(u, 0, 64) + __shfl(u, 4, 64);
    for (int i = nm; i < n; ++i) res += i < 64 ? __shfl(v0, i, 64) : __shfl(v1, i - 64, 64);
    return res;
}

#define FMK_PW_MAX_N (128 << 14)        // the explicit stack below holds 16 levels

#define FMK_PW_STK_F32 64               // ints of per-wave LDS scratch: off[16], len[16], phase[16], left[16]
#define FMK_PW_STK_F64 80               // ... with 8-byte left sums

// stk: per-wave LDS scratch (explicit recursion stack: off, len, phase, left), 8-byte aligned.  n <= FMK_PW_MAX_N.
template <class F>
__device__ __forceinline__ auto fmk_pairwise(F load, int n, int lane, int *stk) -> decltype(load(0))
{
    typedef decltype(load(0)) T;
    if (n <= 128) return fmk_pw_leaf(load, 0, n, lane);
    int *s_off = stk, *s_len = stk + 16, *s_ph = stk + 32;
    T *s_left = (T *)(stk + 48);
    int sp = 1;
    if (lane == 0) { s_off[0] = 0; s_len[0] = n; s_ph[0] = 0; }
    __builtin_amdgcn_wave_barrier();
    T ret = 0;
    bool have = false;
    while (sp > 0) {
        const int top = sp - 1;
        const int off = fmk_uniform(s_off[top]), len = fmk_uniform(s_len[top]), ph = fmk_uniform(s_ph[top]);
        int n2 = len / 2;
        n2 -= n2 % 8;
        if (!have) {
            if (len <= 128) { ret = fmk_pw_leaf(load, off, len, lane); have = true; --sp; }
            else {
                if (lane == 0) { s_off[sp] = off; s_len[sp] = n2; s_ph[sp] = 0; }
                ++sp;
            }
        } else {
            if (ph == 0) {
                if (lane == 0) { s_left[top] = ret; s_ph[top] = 1; s_off[sp] = off + n2; s_len[sp] = len - n2; s_ph[sp] = 0; }
                ++sp;
                have = false;
            } else {
                T left = s_left[top];
                if constexpr (sizeof(T) == 4) left = __builtin_bit_cast(T, fmk_uniform(__builtin_bit_cast(int, left)));
                else left = __builtin_bit_cast(T, fmk_uniform(__builtin_bit_cast(int64_t, left)));
                ret = left + ret;
                --sp;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    return ret;
}

template <class F>
__device__ __forceinline__ float fmk_pairwise_f32(F load, int n, int lane, int *stk)
{
    // np.sum(float32 array): chunks of NPY_BUFSIZE = 8192 elements one after the other, the tree inside a chunk (see fmk_np_sum below)
    if (n <= 8192) return fmk_pairwise(load, n, lane, stk);
    float r = fmk_pairwise(load, 8192, lane, stk);
    for (int i = 8192; i < n; i += 8192) {
        const int len = n - i < 8192 ? n - i : 8192;
        r = r + fmk_pairwise([&load, i](int k) { return load(i + k); }, len, lane, stk);
    }
    return r;
}


// The same sum without walking the tree node by node.  fmk_pairwise visits the nodes one after the other -- every visit a few
// dependent LDS round trips, every leaf waiting for its own loads: 16 leaves + 15 inner nodes for 1 200 elements cost ~50 000
// cycles, 6 ms per 1e9 ticks of 1 200-tick bars and sum.  The tree's shape depends on n alone, so here
//   * the list of leaves is built level by level: lane l owns entry l = (off, len); in each of six rounds the entries longer than
//     128 split into (off, n2) and (off + n2, len - n2), n2 = len / 2 rounded down to a multiple of 8, and a ballot gives every
//     entry its new place (order preserved); the six ballots are kept;
//   * groups of eight lanes take a leaf each, eight leaves at a time: lane i of a group is NumPy's accumulator r_i, adds its
//     elements 8k + i in order, the group folds ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)) and adds the tail;
//   * the rounds are undone in reverse: an entry that had split becomes left + right (two shuffles), others keep their value.
// Bit-identical by construction: every addition has the operands and the order of the recursion.  128 < n <= FMK_PW_PAR_MAX_N
// (leaves hold at least 64 elements, so at most 64 entries); other sizes go to fmk_pairwise.  stk: FMK_PW_PAR_STK ints of LDS.
#define FMK_PW_PAR_MAX_N 4096
#define FMK_PW_PAR_ROUNDS 6                                   // 4096 -> 128 takes five
#define FMK_PW_PAR_STK (80 + 128 + 128)
template <class F>
__device__ __forceinline__ auto fmk_pairwise_par(F load, int n, int lane, int *stk) -> decltype(load(0))
{
    typedef decltype(load(0)) T;
    if (n <= 128 || n > FMK_PW_PAR_MAX_N) return fmk_pairwise(load, n, lane, stk);
    int *l_off = stk + 80, *l_len = stk + 144;
    T *l_sum = (T *)(stk + 208);
    // ---- the leaves, level by level
    int off = 0, len = lane == 0 ? n : 0;                     // len == 0: no entry in this lane
    unsigned long long split_mask[FMK_PW_PAR_ROUNDS];
    const unsigned long long lt = (1ULL << lane) - 1;
#pragma unroll
    for (int d = 0; d < FMK_PW_PAR_ROUNDS; ++d) {
        const bool split = len > 128;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(split);
        split_mask[d] = m;
        if (m == 0) continue;                                 // (wave-uniform)
        const int pos = lane + __builtin_popcountll(m & lt);
        int n2 = len / 2;
        n2 -= n2 % 8;
        __builtin_amdgcn_wave_barrier();
        l_len[lane] = 0;
        __builtin_amdgcn_wave_barrier();
        if (len > 0) {
            l_off[pos] = off; l_len[pos] = split ? n2 : len;
            if (split) { l_off[pos + 1] = off + n2; l_len[pos + 1] = len - n2; }
        }
        __builtin_amdgcn_wave_barrier();
        off = l_off[lane]; len = l_len[lane];
    }
    const int n_leaf = __builtin_popcountll(__builtin_amdgcn_ballot_w64(len > 0));
    __builtin_amdgcn_wave_barrier();
    l_off[lane] = off; l_len[lane] = len;
    __builtin_amdgcn_wave_barrier();
    // ---- the leaf sums, eight leaves per round
    const int grp = lane >> 3, i8 = lane & 7;
    for (int l0 = 0; l0 < n_leaf; l0 += 8) {
        const int l = l0 + grp;
        const bool on = l < n_leaf;
        const int lo = on ? l_off[l] : 0, ll = on ? l_len[l] : 0;
        const int nm = ll - (ll & 7);                        // >= 64 for every leaf of a tree with n > 128
        T x[16], tl[7];
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = 8 * k < nm ? load(lo + 8 * k + i8) : (T)0;
#pragma unroll
        for (int q = 0; q < 7; ++q) tl[q] = nm + q < ll ? load(lo + nm + q) : (T)0;     // the tail, in flight with the rest
        T r = x[0];
#pragma unroll
        for (int k = 1; k < 16; ++k)
            if (8 * k < nm) r += x[k];
        T t = r + __shfl_down(r, 1, 64);                      // lanes 0,2,4,6 of the group: r0+r1, r2+r3, r4+r5, r6+r7
        T u = t + __shfl_down(t, 2, 64);                      // lanes 0,4
        T res = __shfl(u, grp * 8, 64) + __shfl(u, grp * 8 + 4, 64);
#pragma unroll
        for (int q = 0; q < 7; ++q)
            if (nm + q < ll) res += tl[q];
        if (on && i8 == 0) l_sum[l] = res;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- back up the levels: lane l holds the value of entry l
    T val = lane < n_leaf ? l_sum[lane] : (T)0;
#pragma unroll
    for (int d = FMK_PW_PAR_ROUNDS - 1; d >= 0; --d) {
        const unsigned long long m = split_mask[d];
        if (m == 0) continue;
        const int pos = lane + __builtin_popcountll(m & lt);
        const T a = __shfl(val, pos & 63, 64), b = __shfl(val, (pos + 1) & 63, 64);
        val = ((m >> lane) & 1) ? a + b : a;                  // left + right, as the recursion returns it
    }
    if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, val), 0));
    else {
        const int64_t b = __builtin_bit_cast(int64_t, val);
        return __builtin_bit_cast(T, fmk_readlane(b, 0));
    }
}

// ... and for n beyond FMK_PW_PAR_MAX_N: the TOP of the tree is walked node by node (fmk_pairwise's explicit stack), but a node of
// at most FMK_PW_PAR_MAX_N elements is a sub-tree of exactly the shape fmk_pairwise_par evaluates -- the recursion does not know
// where it started -- so it is handed over whole.  A 12 000-element bar is 4 hand-overs and 3 inner nodes instead of ~190 visits.
// stk: FMK_PW_PAR_STK ints (the walk uses the first 80, the sub-trees the rest).  n <= FMK_PW_BIG_MAX_N: the walk's 16 frames then
// reach from the root down to the sub-trees of 4096 elements.
#define FMK_PW_BIG_MAX_N (FMK_PW_PAR_MAX_N << 14)
template <class F>
__device__ __forceinline__ auto fmk_pairwise_big(F load, int n, int lane, int *stk) -> decltype(load(0))
{
    typedef decltype(load(0)) T;
    if (n <= FMK_PW_PAR_MAX_N) return fmk_pairwise_par(load, n, lane, stk);
    int *s_off = stk, *s_len = stk + 16, *s_ph = stk + 32;
    T *s_left = (T *)(stk + 48);
    int sp = 1;
    if (lane == 0) { s_off[0] = 0; s_len[0] = n; s_ph[0] = 0; }
    __builtin_amdgcn_wave_barrier();
    T ret = 0;
    bool have = false;
    while (sp > 0) {
        const int top = sp - 1;
        const int off = fmk_uniform(s_off[top]), len = fmk_uniform(s_len[top]), ph = fmk_uniform(s_ph[top]);
        int n2 = len / 2;
        n2 -= n2 % 8;
        if (!have) {
            if (len <= FMK_PW_PAR_MAX_N) {
                ret = fmk_pairwise_par([&load, off](int i) { return load(off + i); }, len, lane, stk);
                have = true; --sp;
            } else {
                if (lane == 0) { s_off[sp] = off; s_len[sp] = n2; s_ph[sp] = 0; }
                ++sp;
            }
        } else if (ph == 0) {
            if (lane == 0) { s_left[top] = ret; s_ph[top] = 1; s_off[sp] = off + n2; s_len[sp] = len - n2; s_ph[sp] = 0; }
            ++sp;
            have = false;
        } else {
            T left = s_left[top];
            if constexpr (sizeof(T) == 4) left = __builtin_bit_cast(T, fmk_uniform(__builtin_bit_cast(int, left)));
            else left = __builtin_bit_cast(T, fmk_uniform(__builtin_bit_cast(int64_t, left)));
            ret = left + ret;
            --sp;
        }
        __builtin_amdgcn_wave_barrier();
    }
    return ret;
}

// np.sum of a contiguous array as NumPy 2.2 evaluates it: the ufunc reduction hands pairwise_sum at most NPY_BUFSIZE = 8192
// elements at a time and adds the chunks' results one after the other, ((c0 + c1) + c2) + ... -- so the tree above is the sum of
// an array of up to 8 192 elements only (found in round 3 with reference-made vectors for bars of more than 8 192 ticks,
// tests/golden/trade_size_lengths_reference.npz; oracle: orc_pairwise_f32).  TREE: the wave-level tree routine for one chunk.
#define FMK_NP_BUFSIZE 8192
template <class F>
__device__ __forceinline__ auto fmk_np_sum(F load, int64_t n, int lane, int *stk) -> decltype(load(0))
{
    typedef decltype(load(0)) T;
    if (n <= FMK_NP_BUFSIZE) return fmk_pairwise_big(load, (int)n, lane, stk);
    T r = fmk_pairwise_big(load, FMK_NP_BUFSIZE, lane, stk);
    for (int64_t i = FMK_NP_BUFSIZE; i < n; i += FMK_NP_BUFSIZE) {
        const int len = n - i < FMK_NP_BUFSIZE ? (int)(n - i) : FMK_NP_BUFSIZE;
        const int base = (int)i;
        r = r + fmk_pairwise_big([&load, base](int k) { return load(base + k); }, len, lane, stk);
    }
    return r;
}
