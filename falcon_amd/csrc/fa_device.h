// fa_device.h -- wave64 device helpers for gfx950 (CDNA4).
// (a classic include guard: tests/emu/simt pre-includes its host twin under the same name)
#ifndef FA_DEVICE_H
#define FA_DEVICE_H
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

// n consecutive lanes starting at lane lo, as a mask: one s_bfm_b64; 0 < n <= 63.
__device__ __forceinline__ u64 fa_lane_range(int lo, int n) {
    u64 m;
    asm("s_bfm_b64 %0, %1, %2" : "=s"(m) : "s"(n), "s"(lo));
    return m;
}
// mask with bit `clr` cleared and bit `set` set (s_bitset0_b64 / s_bitset1_b64)
__device__ __forceinline__ u64 fa_mask_clr_set(u64 m, int clr, int set) {
    asm("s_bitset0_b64 %0, %1\n\ts_bitset1_b64 %0, %2" : "+s"(m) : "s"(clr), "s"(set));
    return m;
}

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

// a + b + s with s wave-uniform: one v_add3_u32 (the compiler emits two adds)
__device__ __forceinline__ int fa_add3(int a, int b, int s) {
    int r;
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(s));
    return r;
}

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

// Copy through an opaque VALU move: the wait for the load that produced v is paid
// here, once, and the copy carries no pending-load state into the loops that read
// it (otherwise the compiler's conservative s_waitcnt vmcnt(0) at every such read
// also drains the stores in flight).
__device__ __forceinline__ u32 fa_settled(u32 v) {
    u32 r;
    asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

// prefix maximum of unsigned keys inside each row of 16 lanes, in place: 4 VALU
// (lanes without a source lane keep their value; s_nop 1: a VGPR written by VALU needs 2
// wait states before a DPP read)
__device__ __forceinline__ u32 fa_row_prefix_max_u32(u32 v) {
    asm volatile("s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf"
                 : "+v"(v));
    return v;
}
// the same over all 64 lanes: the two row broadcasts behind the in-row steps
__device__ __forceinline__ u32 fa_wave_prefix_max_u32(u32 v) {
    asm volatile("s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
                 : "+v"(v));
    return v;
}

// lane `l` (wave-uniform) of a and of b := the wave-uniform sa, sb; the other lanes keep theirs
// (M0 written once; v_writelane needs its lane select there when the data is an SGPR)
__device__ __forceinline__ void fa_writelane2(int &a, int &b, int sa, int sb, int l) {
    asm volatile("s_mov_b32 m0, %2\n\t"
                 "v_writelane_b32 %0, %3, m0\n\t"
                 "v_writelane_b32 %1, %4, m0"
                 : "+v"(a), "+v"(b)
                 : "s"(l), "s"(sa), "s"(sb)
                 : "m0");
}

// The lanes of a wavefront run in lockstep: where one lane reads what another lane wrote to
// LDS or HBM an instruction earlier, the hardware needs nothing.  This marks such places -- it
// keeps the compiler from moving memory operations across it and costs no instruction (the
// host-side SIMT emulator of tests/emu/simt turns it into a rendezvous of its 64 fibers).
__device__ __forceinline__ void fa_wave_sync() { __builtin_amdgcn_wave_barrier(); }
// The same between LDS operations on one array: nothing at all on the device -- the LDS executes
// a wavefront's operations in order, and the compiler keeps accesses that may alias in program
// order -- so that adjacent regions under one lane mask compile to one (the emulator: a rendezvous).
__device__ __forceinline__ void fa_lds_order() {}

// ---------------------------------------------------------------------------
// shared by the alignment kernels (k_align.hip, k_align_wide.hip)
// ---------------------------------------------------------------------------
// Snake (DW_banded.c:203-206): extend (x,y) while bases match, 16 bases per
// step: two packed words per sequence funnel-shifted (v_alignbit) into one
// 16-base window, xor, find-first-set.  ffbl(0) = -1 turns into a huge count
// that the min() against the remaining lengths clamps.
// v_ffbl_b32 as the hardware defines it: -1 for 0 (the compiler's ctz would be
// undefined there and __ffs costs a compare and a select to patch it).
__device__ __forceinline__ u32 ffbl_raw(u32 x) {
    u32 r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// (qa, ta: base offsets >= 0; unsigned word indices keep the address arithmetic in
// 32 bits: SGPR base + VGPR offset loads instead of 64-bit VALU adds)
template <class QP, class TP>
__device__ __forceinline__ u32 snake_step(QP qL, TP tL, u32 qa, u32 ta, u32 lim) {
    // bit offsets first: (a + b) << 1 is one VALU op, word index and shift derive from it
    // (v_alignbit only looks at the low 5 bits of the shift)
    const u32 qs = qa << 1, ts = ta << 1;
    const u32 qi = qs >> 5, ti = ts >> 5;
    const u32 qw = __builtin_amdgcn_alignbit(qL[qi + 1u], qL[qi], qs);
    const u32 tw = __builtin_amdgcn_alignbit(tL[ti + 1u], tL[ti], ts);
    return min(ffbl_raw(qw ^ tw) >> 1, lim);
}

// The same for code that already runs under the band's exec mask (every executing lane
// holds a cell): no predication by selects, and the loads only carry the band's lanes.
template <class QP, class TP>
__device__ __forceinline__ void snake16_band(QP qL, TP tL, int qb, int tb, int q_len, int t_len,
                                             int &x, int &y) {
    u32 mlast;
    {
        const u32 lim = min(16u, (u32)min(q_len - x, t_len - y));
        const u32 m = snake_step(qL, tL, (u32)(qb + x), (u32)(tb + y), lim);
        x += (int)m;
        y += (int)m;
        mlast = m;
    }
    while (fa_ballot(mlast == 16u)) {  // only lanes inside a run of >= 16 matches get here
        if (mlast == 16u) {
            const u32 lim = min(16u, (u32)min(q_len - x, t_len - y));
            const u32 m = snake_step(qL, tL, (u32)(qb + x), (u32)(tb + y), lim);
            x += (int)m;
            y += (int)m;
            mlast = m;
        }
    }
}

// act lanes hold a cell of the row: 0 <= x <= q_len, 0 <= y <= t_len (a cell that
// reached either end finishes the alignment in its own row, DW_banded.c:220).
template <class QP, class TP>
__device__ __forceinline__ void snake16(QP qL, TP tL, int qb, int tb, int q_len, int t_len,
                                        bool act, int &x, int &y) {
    u32 mlast;  // length of the lane's last step; 16 = a full window matched: go on
    {   // first step: every lane, predicated by selects (idle lanes probe offset 0)
        const int xs = act ? x : 0, ys = act ? y : 0;
        const u32 lim = min(16u, (u32)min(q_len - xs, t_len - ys));
        u32 m = snake_step(qL, tL, (u32)(qb + xs), (u32)(tb + ys), lim);
        m = act ? m : 0u;
        x += (int)m;
        y += (int)m;
        mlast = m;
    }
    u64 gm = fa_ballot(mlast == 16u);
    while (gm) {  // only lanes inside a run of >= 16 matches get here
        if (mlast == 16u) {
            const u32 lim = min(16u, (u32)min(q_len - x, t_len - y));
            const u32 m = snake_step(qL, tL, (u32)(qb + x), (u32)(tb + y), lim);
            x += (int)m;
            y += (int)m;
            mlast = m;
        }
        gm = fa_ballot(mlast == 16u);
    }
}
#endif  // FA_DEVICE_H
