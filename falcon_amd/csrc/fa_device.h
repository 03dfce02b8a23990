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

// Wave-wide max of ints through DPP (no LDS traffic):
// prefix-max inside each row of 16, then row_bcast15 / row_bcast31 (gfx9 DPP
// controls, present on gfx950) carry it across rows; lane 63 holds the result.
__device__ __forceinline__ int fa_wave_max(int v) {
    int t;
    t = __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false); v = max(v, t);  // row_shr:1
    t = __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false); v = max(v, t);  // row_shr:2
    t = __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false); v = max(v, t);  // row_shr:4
    t = __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false); v = max(v, t);  // row_shr:8
    t = __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false); v = max(v, t);  // row_bcast:15
    t = __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false); v = max(v, t);  // row_bcast:31
    return __builtin_amdgcn_readlane(v, 63);
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
