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
    T r = 0;
    if (lane < 8) {
        r = load(off + lane);
        for (int i = 8 + lane; i < nm; i += 8) r += load(off + i);
    }
    T t = r + __shfl_down(r, 1, 64);         // lanes 0,2,4,6: r0+r1, r2+r3, r4+r5, r6+r7
    T u = t + __shfl_down(t, 2, 64);         // lanes 0,4
    T res = __shfl(u, 0, 64) + __shfl(u, 4, 64);
    for (int i = nm; i < n; ++i) res += load(off + i);
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
__device__ __forceinline__ float fmk_pairwise_f32(F load, int n, int lane, int *stk) { return fmk_pairwise(load, n, lane, stk); }

