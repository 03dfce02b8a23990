// fa_device.h -- wave64 device helpers for gfx950 (CDNA4).
#pragma once
#include "fa_internal.h"

#define FA_WAVE 64

__device__ __forceinline__ int fa_lane() { return (int)(threadIdx.x & 63); }

// Base code (A0 C1 G2 T3) i of a packed sequence.
__device__ __forceinline__ u32 fa_base_at(const u32 *w, int i) {
    return (w[i >> 4] >> ((i & 15) * 2)) & 3u;
}

// 64-bit window of packed bases starting at base i: bases i .. i+16+(15-(i&15))
// are valid, i.e. at least 17 and at most 32 bases.  Reads words i/16 and i/16+1.
__device__ __forceinline__ u64 fa_window64(const u32 *w, int i) {
    u64 lo = w[i >> 4], hi = w[(i >> 4) + 1];
    return ((hi << 32) | lo) >> ((i & 15) * 2);
}

// 8-mer (16 bits) starting at base i.
__device__ __forceinline__ u32 fa_kmer8(const u32 *w, int i) {
    return (u32)(fa_window64(w, i) & 0xFFFFu);
}

// Lane mask of a predicate.  Feed it ONE comparison: the compiler then emits just the
// v_cmp; a compound condition is first materialised as 0/1 and compared again (two
// extra VALU) -- combine masks with scalar and/or instead.
__device__ __forceinline__ u64 fa_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// n consecutive lanes starting at lane lo, as a mask (s_bfm_b64); 0 < n <= 63.
__device__ __forceinline__ u64 fa_lane_range(int lo, int n) { return ((1ull << n) - 1ull) << lo; }

// mask ? b : a per lane, the lane mask given as a scalar (v_cndmask with an SGPR pair)
__device__ __forceinline__ int fa_sel(u64 mask, int a, int b) {
    int r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
    return r;
}

// Wave-wide max of ints through DPP (no LDS traffic): prefix-max inside each row
// of 16, then row_bcast15 / row_bcast31 (gfx9 DPP controls, present on gfx950) carry
// it across rows; lane 63 holds the result.  The DPP modifier sits on the v_max
// itself, in place (lanes without a source keep their value): 6 VALU.  (Through
// __builtin_amdgcn_update_dpp the compiler emits copy + v_mov_dpp + v_max per step.)
// s_nop 1: a VGPR written by VALU needs 2 wait states before a DPP read.
__device__ __forceinline__ int fa_wave_max(int v) {
    asm volatile("s_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
                 : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}

// Same for values whose wave maximum is known to be >= 0: lanes without a DPP source
// read 0 (bound_ctrl), so the first step can write a fresh register and the input
// survives without a copy.
__device__ __forceinline__ int fa_wave_max_nonneg(int v) {
    int t;
    asm volatile("s_nop 1\n\t"
                 "v_max_i32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
                 : "=&v"(t)
                 : "v"(v));
    return __builtin_amdgcn_readlane(t, 63);
}

// 2 * x + (lane in mask): v_addc with the mask as carry-in -- (x << 1) | bit in one VALU op
__device__ __forceinline__ u32 fa_twice_plus(int x, u64 mask) {
    u32 r;
    u64 carry_out;
    asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(r), "=s"(carry_out) : "v"(x), "s"(mask));
    (void)carry_out;
    return r;
}

__device__ __forceinline__ int fa_wave_min(int v) { return -fa_wave_max(-v); }

// Force a wave-uniform value into SGPRs.  Arguments of out-of-line device
// functions arrive in VGPRs and the compiler must treat them as divergent;
// readfirstlane makes the uniformity explicit so branches on them stay scalar.
__device__ __forceinline__ int fa_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ u32 fa_uni(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u64 fa_uni(u64 v) {
    const u32 lo = fa_uni((u32)v), hi = fa_uni((u32)(v >> 32));
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ double fa_uni(double v) {
    return __longlong_as_double((long long)fa_uni((u64)__double_as_longlong(v)));
}
template <class T>
__device__ __forceinline__ T *fa_uni(T *p) { return (T *)fa_uni((u64)p); }
