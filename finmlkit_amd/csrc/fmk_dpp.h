// fmk_dpp.h -- wave64 cross-lane primitives on the DPP data path (gfx950 / GFX9 family).
//
// `__shfl_up/xor` lower to ds_bpermute_b32: every step is a round trip through the LDS crossbar
// (~100+ cycles of dependent latency).  A scan or reduction over 64 lanes is six dependent steps,
// so a kernel that scans every 64-tick chunk (comp_bar_directional_features) spends most of its time
// waiting on the crossbar.  The DPP modifiers (row_shr:n, row_bcast:15/31, wave_shr:1) move data
// between lanes inside the VALU pipeline: the canonical GCN sequence
//     row_shr:1, row_shr:2, row_shr:4, row_shr:8   (Kogge-Stone inside each 16-lane row)
//     row_bcast:15 (rows 1,3) , row_bcast:31 (rows 2,3)
// is an inclusive 64-lane scan in 6 VALU steps with no LDS traffic.  64-bit values move as two
// 32-bit halves.  The combining order is fixed, so results are deterministic.
#pragma once

#include "fmk_common.h"

#define FMK_DPP_ROW_SHR(n) (0x110 + (n))
#define FMK_DPP_WAVE_SHR1 0x138
#define FMK_DPP_ROW_BCAST15 0x142
#define FMK_DPP_ROW_BCAST31 0x143

// 32-bit move with DPP control; lanes without a source lane (or masked rows) receive `old`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int fmk_dpp_i32(int old, int v)
{
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int64_t fmk_dpp(int64_t old, int64_t v)
{
    int lo = fmk_dpp_i32<CTRL, ROW_MASK>((int)(uint32_t)old, (int)(uint32_t)v);
    int hi = fmk_dpp_i32<CTRL, ROW_MASK>((int)((uint64_t)old >> 32), (int)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int fmk_dpp(int old, int v) { return fmk_dpp_i32<CTRL, ROW_MASK>(old, v); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double fmk_dpp(double old, double v)
{
    return __longlong_as_double(fmk_dpp<CTRL, ROW_MASK>((int64_t)__double_as_longlong(old),
                                                         (int64_t)__double_as_longlong(v)));
}

struct FmkOpAdd { template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a + b; } };
struct FmkOpMin {
    __device__ __forceinline__ double operator()(double a, double b) const { return fmin(a, b); }
    __device__ __forceinline__ int64_t operator()(int64_t a, int64_t b) const { return a < b ? a : b; }
    __device__ __forceinline__ int operator()(int a, int b) const { return a < b ? a : b; }
};
struct FmkOpMax {
    __device__ __forceinline__ double operator()(double a, double b) const { return fmax(a, b); }
    __device__ __forceinline__ int64_t operator()(int64_t a, int64_t b) const { return a > b ? a : b; }
    __device__ __forceinline__ int operator()(int a, int b) const { return a > b ? a : b; }
};

// Inclusive scan over the 64 lanes; `ident` is the identity of `op` (what a missing source contributes).
template <typename T, typename Op>
__device__ __forceinline__ T fmk_dpp_iscan(T v, T ident, Op op)
{
    v = op(fmk_dpp<FMK_DPP_ROW_SHR(1), 0xF>(ident, v), v);
    v = op(fmk_dpp<FMK_DPP_ROW_SHR(2), 0xF>(ident, v), v);
    v = op(fmk_dpp<FMK_DPP_ROW_SHR(4), 0xF>(ident, v), v);
    v = op(fmk_dpp<FMK_DPP_ROW_SHR(8), 0xF>(ident, v), v);
    v = op(fmk_dpp<FMK_DPP_ROW_BCAST15, 0xA>(ident, v), v);   // rows 1 and 3 take lane 15 of the row before
    v = op(fmk_dpp<FMK_DPP_ROW_BCAST31, 0xC>(ident, v), v);   // rows 2 and 3 take lane 31
    return v;
}

// value of lane 63 broadcast to the wave (SGPRs)
__device__ __forceinline__ int fmk_last_lane(int v) { return __builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ int64_t fmk_last_lane(int64_t v)
{
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)((uint64_t)v >> 32), 63);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ double fmk_last_lane(double v)
{
    return __longlong_as_double(fmk_last_lane((int64_t)__double_as_longlong(v)));
}

// Reduction over the 64 lanes, result wave-uniform.
template <typename T, typename Op>
__device__ __forceinline__ T fmk_dpp_reduce(T v, T ident, Op op)
{
    return fmk_last_lane(fmk_dpp_iscan(v, ident, op));
}

// lane l receives the value of lane l-1; lane 0 receives `first`
template <typename T>
__device__ __forceinline__ T fmk_dpp_shift_up1(T v, T first)
{
    return fmk_dpp<FMK_DPP_WAVE_SHR1, 0xF>(first, v);
}

// ---- reductions inside a ROW of 16 lanes (four independent rows per wave), result in every lane of the row: xor butterflies on the
// DPP data path -- quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror (lane i <-> 7 - i: by then the quads are uniform),
// row_mirror (i <-> 15 - i).  For a sum this is the balanced binary tree over the row's 16 values in lane order (a + b and b + a
// are the same float), i.e. what fmk_dpp_reduce computes inside each row.
#define DPP_XOR1 0xB1            // quad_perm [1,0,3,2]
#define DPP_XOR2 0x4E            // quad_perm [2,3,0,1]
#define DPP_HALF_MIRROR 0x141    // lane i <-> 7 - i inside each half row
#define DPP_MIRROR 0x140         // lane i <-> 15 - i inside each row

__device__ __forceinline__ float fmk_row_xor_f32(float v, int which)
{
    const int x = __float_as_int(v);
    int y;
    switch (which) {
    case 1: y = __builtin_amdgcn_update_dpp(x, x, DPP_XOR1, 0xF, 0xF, false); break;
    case 2: y = __builtin_amdgcn_update_dpp(x, x, DPP_XOR2, 0xF, 0xF, false); break;
    case 4: y = __builtin_amdgcn_update_dpp(x, x, DPP_HALF_MIRROR, 0xF, 0xF, false); break;
    default: y = __builtin_amdgcn_update_dpp(x, x, DPP_MIRROR, 0xF, 0xF, false); break;
    }
    return __int_as_float(y);
}
__device__ __forceinline__ int fmk_row_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(v, v, DPP_XOR1, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(v, v, DPP_XOR2, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(v, v, DPP_HALF_MIRROR, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(v, v, DPP_MIRROR, 0xF, 0xF, false);
    return v;
}
__device__ __forceinline__ uint32_t fmk_row_umin(uint32_t v)
{
    uint32_t w;
    w = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_XOR1, 0xF, 0xF, false); v = w < v ? w : v;
    w = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_XOR2, 0xF, 0xF, false); v = w < v ? w : v;
    w = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_HALF_MIRROR, 0xF, 0xF, false); v = w < v ? w : v;
    w = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_MIRROR, 0xF, 0xF, false); v = w < v ? w : v;
    return v;
}
__device__ __forceinline__ uint32_t fmk_row_umax(uint32_t v)
{
    uint32_t w;
    w = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_XOR1, 0xF, 0xF, false); v = w > v ? w : v;
    w = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_XOR2, 0xF, 0xF, false); v = w > v ? w : v;
    w = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_HALF_MIRROR, 0xF, 0xF, false); v = w > v ? w : v;
    w = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_MIRROR, 0xF, 0xF, false); v = w > v ? w : v;
    return v;
}
__device__ __forceinline__ double fmk_row_sum(double v)
{
    v += fmk_dpp<DPP_XOR1, 0xF>(v, v);
    v += fmk_dpp<DPP_XOR2, 0xF>(v, v);
    v += fmk_dpp<DPP_HALF_MIRROR, 0xF>(v, v);
    v += fmk_dpp<DPP_MIRROR, 0xF>(v, v);
    return v;
}

__device__ __forceinline__ double fmk_row_max(double v)
{
    v = fmax(v, fmk_dpp<DPP_XOR1, 0xF>(v, v));
    v = fmax(v, fmk_dpp<DPP_XOR2, 0xF>(v, v));
    v = fmax(v, fmk_dpp<DPP_HALF_MIRROR, 0xF>(v, v));
    v = fmax(v, fmk_dpp<DPP_MIRROR, 0xF>(v, v));
    return v;
}
__device__ __forceinline__ double fmk_row_min(double v)
{
    v = fmin(v, fmk_dpp<DPP_XOR1, 0xF>(v, v));
    v = fmin(v, fmk_dpp<DPP_XOR2, 0xF>(v, v));
    v = fmin(v, fmk_dpp<DPP_HALF_MIRROR, 0xF>(v, v));
    v = fmin(v, fmk_dpp<DPP_MIRROR, 0xF>(v, v));
    return v;
}

// ---- the same inside a HALF ROW of 8 lanes (eight independent segments per wave): three butterfly steps; and the inclusive scan of
// a half row (row_shr:1 / 2 / 4; a lane whose source lies in the other half keeps its value)
__device__ __forceinline__ double fmk_half_sum(double v)
{
    v += fmk_dpp<DPP_XOR1, 0xF>(v, v);
    v += fmk_dpp<DPP_XOR2, 0xF>(v, v);
    v += fmk_dpp<DPP_HALF_MIRROR, 0xF>(v, v);
    return v;
}
__device__ __forceinline__ double fmk_half_max(double v)
{
    v = fmax(v, fmk_dpp<DPP_XOR1, 0xF>(v, v));
    v = fmax(v, fmk_dpp<DPP_XOR2, 0xF>(v, v));
    v = fmax(v, fmk_dpp<DPP_HALF_MIRROR, 0xF>(v, v));
    return v;
}
__device__ __forceinline__ double fmk_half_min(double v)
{
    v = fmin(v, fmk_dpp<DPP_XOR1, 0xF>(v, v));
    v = fmin(v, fmk_dpp<DPP_XOR2, 0xF>(v, v));
    v = fmin(v, fmk_dpp<DPP_HALF_MIRROR, 0xF>(v, v));
    return v;
}
__device__ __forceinline__ double fmk_half_iscan_add(double v, int lane)
{
    const int k = lane & 7;
    double t;
    t = fmk_dpp<FMK_DPP_ROW_SHR(1), 0xF>(0.0, v); v += k >= 1 ? t : 0.0;
    t = fmk_dpp<FMK_DPP_ROW_SHR(2), 0xF>(0.0, v); v += k >= 2 ? t : 0.0;
    t = fmk_dpp<FMK_DPP_ROW_SHR(4), 0xF>(0.0, v); v += k >= 4 ? t : 0.0;
    return v;
}

// bitwise OR over the eight lanes of a half row (result in every lane of it)
__device__ __forceinline__ uint64_t fmk_half_or(uint64_t v)
{
    v |= (uint64_t)fmk_dpp<DPP_XOR1, 0xF>((int64_t)v, (int64_t)v);
    v |= (uint64_t)fmk_dpp<DPP_XOR2, 0xF>((int64_t)v, (int64_t)v);
    v |= (uint64_t)fmk_dpp<DPP_HALF_MIRROR, 0xF>((int64_t)v, (int64_t)v);
    return v;
}
