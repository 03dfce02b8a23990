// fa_wave.h -- the wave64 primitive layer k_align2_core.h is written against (gfx950).
//
// The core of k_align2 uses per-lane values (`vi` / `vu`), lane masks (u64) and the
// cross-lane operations below, nothing else.  In the product they are plain registers and
// single instructions.  tests/emu/ holds a twin of this header that implements the same
// names on 64-element arrays, so that the kernel's control logic (two alignments per
// wavefront, band placement, parking, the iteration tape, the trace-back) runs against the
// CPU oracle in the `-m "not gpu"` suite; that twin is test infrastructure and is never
// compiled into the library.
#pragma once
#include "fa_device.h"

typedef int vi;    // one int per lane
typedef u32 vu;    // one u32 per lane
typedef bool vb;   // one predicate per lane (feed ONE comparison to w_ballot)

// the row loop of k_align2 as a hand-scheduled instruction stream (k_align2_rows.h)
#define W_ROWS_ASM 1
#define W_FN __device__ __forceinline__
#define W_NOINLINE __device__ __noinline__
// the lanes of mask `m` execute the block (the others keep their values)
#define W_WHERE(m) if (__builtin_amdgcn_inverse_ballot_w64(m))

W_FN vi w_lane() { return fa_lane(); }
W_FN u64 w_ballot(vb p) { return fa_ballot(p); }
// lane in m ? b : a
// (the predicate straight from the scalar mask: v_cndmask on an SGPR pair, no v_cmp)
W_FN vi w_sel(u64 m, vi a, vi b) { return __builtin_amdgcn_inverse_ballot_w64(m) ? b : a; }
W_FN vu w_selu(u64 m, vu a, vu b) { return __builtin_amdgcn_inverse_ballot_w64(m) ? b : a; }
// lane l takes the value of lane l - 1 (lane 0: 0) / of lane l + 1 (lane 63: 0)
W_FN vi w_from_below(vi v) { return __builtin_amdgcn_mov_dpp(v, 0x138, 0xf, 0xf, true); }  // wave_shr:1
W_FN vi w_from_above(vi v) { return __builtin_amdgcn_mov_dpp(v, 0x130, 0xf, 0xf, true); }  // wave_shl:1
// wave-uniform value of lane l (l wave-uniform)
W_FN int w_readlane(vi v, int l) { return __builtin_amdgcn_readlane(v, l); }
W_FN u32 w_readlaneu(vu v, int l) { return (u32)__builtin_amdgcn_readlane((int)v, l); }
// two registers, one lane select (M0 written once)
W_FN void w_writelane2(vu &a, vu &b, u32 sa, u32 sb, int l) {
    asm volatile("s_mov_b32 m0, %4\n\t"
                 "v_writelane_b32 %0, %2, m0\n\t"
                 "v_writelane_b32 %1, %3, m0"
                 : "+v"(a), "+v"(b)
                 : "s"(sa), "s"(sb), "s"(l)
                 : "m0");
}
// lane l of v := the wave-uniform s (l wave-uniform)
W_FN void w_setlane(vi &v, int s, int l) {
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(fa_uni(s)), "s"(fa_uni(l)) : "m0");
}
// lane dst (per lane, 0..63) receives this lane's v; lanes nobody sends to read 0, of two
// senders to one lane the higher lane wins
W_FN vi w_push_lanes(vi v, vi dst) { return __builtin_amdgcn_ds_permute(dst << 2, v); }
// lane l takes v of lane src (per-lane src, 0..63)
W_FN vi w_gather_lanes(vi v, vi src) { return __builtin_amdgcn_ds_bpermute(src << 2, v); }
W_FN int w_uni(int v) { return fa_uni(v); }
W_FN u32 w_uniu(u32 v) { return fa_uni(v); }

// inclusive prefix maximum over the lanes, unsigned: 4 in-row DPP steps, then the two row
// broadcasts (lanes without a source keep their value)
W_FN vu w_prefix_max(vu v) {
    // (the first step writes a fresh register -- lanes without a source read 0, the identity
    // of an unsigned maximum -- so the argument survives without a copy)
    vu t;
    asm volatile("s_nop 1\n\t"
                 "v_max_u32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
                 : "=&v"(t)
                 : "v"(v));
    return t;
}

// inclusive prefix sum over the lanes: the same six steps (lanes without a source add 0)
W_FN vu w_prefix_add(vu v) {
    vu t;
    asm volatile("s_nop 1\n\t"
                 "v_add_u32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
                 : "=&v"(t)
                 : "v"(v));
    return t;
}

// the minimum over all lanes (signed), wave-uniform: the same six steps, lanes without a source
// keep their value
W_FN int w_reduce_min(vi v) {
    vi t = v;
    asm volatile("s_nop 1\n\t"
                 "v_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_min_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_min_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_min_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_min_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_min_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
                 : "+v"(t));
    return __builtin_amdgcn_readlane(t, 63);
}

// The trace-back's chain over the rows in lanes base .. 63 (k_align2_core.h, a2_trace): row l of
// the path sits on lane j of its iteration, the row before it on lane j + step[l] + (bit j of row
// l's from_above mask).  Returns where the chain ends; lane l of `my` := the j of row l.  One
// unrolled stream of 64 steps of 40 bytes -- three records read by v_readlane, the lane kept by
// v_writelane, s_bitcmp1_b64 + s_addc_u32 -- entered at step `base` by a computed jump.
W_FN int w_chain(vu ra, vu rb, vi step, int j, int base, vi &my) {
    u32 sj = (u32)fa_uni(j), st;
    const u32 off = (u32)fa_uni(base) * 40u;
    asm volatile("s_getpc_b64 vcc\n"
                 ".Lwch_a_%=:\n\t"
                 "s_add_u32 vcc_lo, vcc_lo, %[off]\n\t"
                 "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
                 "s_add_u32 vcc_lo, vcc_lo, .Lwch_0_%=-.Lwch_a_%=\n\t"
                 "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
                 "s_setpc_b64 vcc\n"
                 ".Lwch_0_%=:\n\t"
                 ".irp l,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63\n\t"
                 "v_readlane_b32 vcc_lo, %[ra], \\l\n\t"
                 "v_readlane_b32 vcc_hi, %[rb], \\l\n\t"
                 "v_readlane_b32 %[st], %[step], \\l\n\t"
                 "v_writelane_b32 %[my], %[j], \\l\n\t"
                 "s_bitcmp1_b64 vcc, %[j]\n\t"
                 "s_addc_u32 %[j], %[j], %[st]\n\t"
                 ".endr"
                 : [my] "+v"(my), [j] "+s"(sj), [st] "=&s"(st)
                 : [ra] "v"(ra), [rb] "v"(rb), [step] "v"(step), [off] "s"(off)
                 : "vcc", "scc");
    return (int)sj;
}

// The same chain in a frame of its own (round 6): index i = lane + shift[l], the shifts chosen by the caller so
// that the path moves by (bit i of row l's SHIFTED mask) + c_l with c_l = -(l & 1) -- a constant of the unrolled
// stream, so no step is read: two v_readlane, v_writelane, s_bitcmp1_b64, s_addc_u32, 32 bytes a row.  Lane l of
// `my` := the i of row l.  From i = 31 the path stays inside 0 .. 63 whatever the 64 rows do (it gains a lane on
// even rows only, loses one on odd rows only).
#define W_CH2(L, C)                                   \
    "v_readlane_b32 vcc_lo, %[ra], " #L "\n\t"        \
    "v_readlane_b32 vcc_hi, %[rb], " #L "\n\t"        \
    "v_writelane_b32 %[my], %[i], " #L "\n\t"         \
    "s_bitcmp1_b64 vcc, %[i]\n\t"                     \
    "s_addc_u32 %[i], %[i], " #C "\n\t"
#define W_CH2_8(A, B, C, D, E, F, G, H) W_CH2(A, 0) W_CH2(B, -1) W_CH2(C, 0) W_CH2(D, -1) W_CH2(E, 0) W_CH2(F, -1) W_CH2(G, 0) W_CH2(H, -1)
W_FN int w_chain2(vu ra, vu rb, int i, int base, vi &my) {
    u32 si = (u32)fa_uni(i);
    const u32 off = (u32)fa_uni(base) * 32u;
    asm volatile("s_getpc_b64 vcc\n"
                 ".Lwc2_a_%=:\n\t"
                 "s_add_u32 vcc_lo, vcc_lo, %[off]\n\t"
                 "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
                 "s_add_u32 vcc_lo, vcc_lo, .Lwc2_0_%=-.Lwc2_a_%=\n\t"
                 "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
                 "s_setpc_b64 vcc\n"
                 ".Lwc2_0_%=:\n\t"
                 W_CH2_8(0, 1, 2, 3, 4, 5, 6, 7) W_CH2_8(8, 9, 10, 11, 12, 13, 14, 15)
                 W_CH2_8(16, 17, 18, 19, 20, 21, 22, 23) W_CH2_8(24, 25, 26, 27, 28, 29, 30, 31)
                 W_CH2_8(32, 33, 34, 35, 36, 37, 38, 39) W_CH2_8(40, 41, 42, 43, 44, 45, 46, 47)
                 W_CH2_8(48, 49, 50, 51, 52, 53, 54, 55) W_CH2_8(56, 57, 58, 59, 60, 61, 62, 63)
                 ".Lwc2_e_%=:\n\t"
                 ".if .Lwc2_e_%= - .Lwc2_0_%= != 2048\n\t.error \"w_chain2: a step is not 32 bytes\"\n\t.endif"
                 : [my] "+v"(my), [i] "+s"(si)
                 : [ra] "v"(ra), [rb] "v"(rb), [off] "s"(off)
                 : "vcc", "scc");
    return (int)si;
}
// (hi:lo) of every lane shifted by the lane's s: left for s >= 0, right for s < 0 (|s| < 64)
W_FN void w_shift64(vu lo, vu hi, vi s, vu &olo, vu &ohi) {
    const u64 m = ((u64)hi << 32) | lo;
    const u64 r = s >= 0 ? (m << (s & 63)) : (m >> ((-s) & 63));
    olo = (u32)r; ohi = (u32)(r >> 32);
}

// The tail of a band row as ONE instruction stream: the inclusive prefix maximum of `key`
// (as w_prefix_max) with the row's other work in the wait states of its DPP steps, where
// s_nop would sit otherwise -- a VGPR written by a VALU instruction may be read through DPP
// two instructions later at the earliest.  Beside the maximum:
//   fin  = (x >= qlen | y >= tlen) & act      big = (m > 254) & act      keyb = key + band
//   lane `slot` of rc_lo / rc_hi := the two halves of `fa`
// (11 instructions + 6 DPP steps + 1 s_nop, where the parts took 23.)
W_FN vu w_row_tail(vu key, vi x, vi qlen, vi y, vi tlen, vu m, u64 act, u32 band, u64 fa, int slot,
                   vu &rc_lo, vu &rc_hi, u64 &fin, u64 &big, vu &keyb) {
    vu t;
    u64 f, b;
    vu kb;
    asm volatile("v_cmp_ge_i32_e32 vcc, %7, %8\n\t"
                 "v_cmp_ge_i32_e64 %1, %9, %10\n\t"
                 "v_max_u32_dpp %0, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_or_b64 %1, %1, vcc\n\t"
                 "v_cmp_lt_u32_e32 vcc, 0xfe, %11\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_and_b64 %2, vcc, %12\n\t"
                 "s_and_b64 %1, %1, %12\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_u32_e32 %3, %13, %6\n\t"
                 "s_mov_b32 m0, %16\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_writelane_b32 %4, %14, m0\n\t"
                 "v_writelane_b32 %5, %15, m0\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
                 : "=&v"(t), "=&s"(f), "=&s"(b), "=&v"(kb), "+v"(rc_lo), "+v"(rc_hi)
                 : "v"(key), "v"(x), "v"(qlen), "v"(y), "v"(tlen), "v"(m), "s"(act), "s"(band),
                   "s"((u32)fa), "s"((u32)(fa >> 32)), "s"(slot)
                 : "vcc", "m0");
    fin = f; big = b; keyb = kb;
    return t;
}

W_FN vi w_min(vi a, vi b) { return min(a, b); }
W_FN vi w_max(vi a, vi b) { return max(a, b); }
W_FN vu w_minu(vu a, vu b) { return min(a, b); }
// find-first-set as the hardware defines it: 0xffffffff for 0
W_FN vu w_ffbl(vu x) { return ffbl_raw(x); }
// (hi:lo) >> (sh & 31), low word
W_FN vu w_alignbit(vu hi, vu lo, vu sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
// (x << 1) | (lane in m)
W_FN vu w_twice_plus(vi x, u64 m) { return fa_twice_plus(x, m); }

// scalar mask helpers
W_FN u64 w_lanes(int lo, int n) { return fa_lane_range(lo, n); }       // n lanes from lo; 0 < n <= 63
W_FN u64 w_bit_clr(u64 m, int b) {
    asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(b));
    return m;
}
W_FN u64 w_bit_set(u64 m, int b) {
    asm("s_bitset1_b64 %0, %1" : "+s"(m) : "s"(b));
    return m;
}
W_FN int w_lowest(u64 m) { return __builtin_ctzll(m); }                 // m != 0
W_FN int w_highest(u64 m) { return 63 - __builtin_clzll(m); }           // m != 0
W_FN int w_popc(u64 m) { return __builtin_popcountll(m); }
// bit j (per lane, 0..63) of the wave-uniform mask: one 64-bit shift and an and
W_FN vi w_bit_at(u64 mask, vi j) { return (vi)((mask >> (j & 63)) & 1ull); }
// sign-extended 16 bits of v from bit `off` (per lane)
W_FN vi w_bfe_i16(vu v, vu off) { return __builtin_amdgcn_sbfe((int)v, off, 16u); }
W_FN int w_span(u64 m) { return 64 - (__builtin_clzll(m) + __builtin_ctzll(m)); }  // highest - lowest + 1; m != 0
// per lane: the number of set bits of m below the lane
W_FN vi w_rank_in(u64 m) {
    return (vi)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}
// acc with byte J replaced by the low byte of m (v_perm_b32; the other bytes of m do not matter)
template <int J>
W_FN vu w_put_byte(vu acc, vu m) {
    return __builtin_amdgcn_perm(m, acc, 0x03020100u + ((4u - (u32)J) << (8 * J)));
}
// the same with the byte position in a wave-uniform selector (see a2_fast)
W_FN vu w_put_byte_sel(vu acc, vu m, u32 selector) { return __builtin_amdgcn_perm(m, acc, selector); }
// wave-uniform values travel through calls in the lanes of a register (an out-of-line
// device function takes and returns its arguments in VGPRs): lane I of p := s / is read
template <int I>
W_FN void w_pack_put(vu &p, u32 s) {
    // (the compiler keeps some uniform values on the vector unit: readfirstlane is free for
    // the ones that are in SGPRs already)
    asm("v_writelane_b32 %0, %1, %2" : "+v"(p) : "s"(fa_uni(s)), "n"(I));
}
template <int I>
W_FN u32 w_pack_get(vu p) { return (u32)__builtin_amdgcn_readlane((int)p, I); }
// a register whose content does not matter yet (no instruction)
W_FN vu w_undef() {
    vu v;
    asm volatile("" : "=v"(v));
    return v;
}

// ---- memory (global pointers are address space 1: SGPR base + VGPR offset forms) ----
typedef u32 w_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32 w_gu32;
typedef __attribute__((address_space(1))) uint8_t w_gu8;
typedef __attribute__((address_space(1))) w_u32x4 w_gu32x4;
typedef __attribute__((address_space(1))) u64 w_gu64;

// W_AT(T, base, i): element i of the array of T at `base`.  W_ADDR32 (a build switch, off): the byte offset is
// formed in 32 bits -- every array the kernels index this way is smaller than 4 GB (the packed words of a launch:
// < 2^28 of them, engine.hip cuts larger batches; a tape slot; a script) -- so that the access is `SGPR base +
// 32-bit VGPR offset`.  As it is, hipcc widens the index to 64 bits (it cannot know the product fits), keeps a
// register tuple with a zero high half for that through k_align2's whole event loop, and spills / reloads the
// tuple around it: 99 of the kernel's 200 scratch instructions, 16 bytes per lane each
// (profiles/r06_k_align2_bytes_by_site.txt).  Off until it has run on the GPU.
typedef __attribute__((address_space(1))) char w_gchar;
#if defined(W_ADDR32)
#define W_AT(T, base, i) (*(T *)((w_gchar *)(base) + (size_t)(u32)((u32)(i) * (u32)sizeof(T))))
#define W_AT_C(T, base, i) (*(const T *)((const w_gchar *)(base) + (size_t)(u32)((u32)(i) * (u32)sizeof(T))))
#else
#define W_AT(T, base, i) (((T *)(base))[i])
#define W_AT_C(T, base, i) (((const T *)(base))[i])
#endif
// words base[i], base[i + 1] (i per lane)
W_FN void w_load_pair(const u32 *base, vu i, vu &lo, vu &hi) {
    lo = W_AT_C(w_gu32, base, i);
    hi = W_AT_C(w_gu32, base, i + 1u);
}
W_FN vu w_load32(const u32 *base, vu i) { return W_AT_C(w_gu32, base, i); }
W_FN void w_store32(u32 *base, vu i, vu v) { W_AT(w_gu32, base, i) = v; }
W_FN vu w_load8(const uint8_t *base, vu i) { return (vu)((const w_gu8 *)base)[i]; }
W_FN void w_store64(u64 *base, vu i, vu lo, vu hi) { W_AT(w_gu64, base, i) = ((u64)hi << 32) | lo; }
W_FN void w_load64(const u64 *base, vu i, vu &lo, vu &hi) {
    const u64 v = W_AT_C(w_gu64, base, i);
    lo = (vu)v;
    hi = (vu)(v >> 32);
}
// 16-byte records
W_FN void w_store_x4(u32 *base16, vu i, vu a, vu b, vu c, vu d) {
    const w_u32x4 r = {a, b, c, d};
    W_AT(w_gu32x4, base16, i) = r;
}
W_FN void w_load_x4(const u32 *base16, vu i, vu &a, vu &b, vu &c, vu &d) {
    const w_u32x4 r = W_AT_C(w_gu32x4, base16, i);
    a = r.x; b = r.y; c = r.z; d = r.w;
}
// the same value from every lane (results, counters)
W_FN int w_atomic_add_lane0(int *p, int v) {
    // (no `if (lane == 0)`: hipcc 7.2 jump-threads such blocks across loop back edges)
    return w_uni(atomicAdd(p, fa_lane() == 0 ? v : 0));
}
W_FN void w_fence_block() { __threadfence_block(); }
// an alignment summary, the same record from every lane
W_FN void w_store_aln(FaAln *dst, const FaAln &r) {
    static_assert(sizeof(FaAln) == 40, "layout");
    typedef u32 u32x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(1))) u32x2 g_u32x2;
    const w_u32x4 a = {(u32)r.dist, (u32)r.q_e, (u32)r.t_e, (u32)r.size};
    const w_u32x4 b = {(u32)r.accept, (u32)r.n_ins, (u32)r.aligned, (u32)r.err};
    const u32x2 c = {(u32)(u64)r.cells, (u32)((u64)r.cells >> 32)};
    ((w_gu32x4 *)dst)[0] = a;
    ((w_gu32x4 *)dst)[1] = b;
    ((g_u32x2 *)dst)[4] = c;
}
W_FN void w_stat_add(unsigned long long *p, unsigned long long v) {
    if (p) atomicAdd(p, fa_lane() == 0 ? v : 0ull);
}

// LDS words of the wavefront (the kernel's dynamic __shared__ array)
W_FN u32 *w_lds() {
    extern __shared__ __attribute__((aligned(16))) u32 smem[];
    return smem;
}
W_FN void w_lds_store(u32 *l, vu i, vu v) { l[i] = v; }
W_FN vu w_lds_bcast(const u32 *l, int i) { return l[i]; }   // every lane reads word i
W_FN vu w_lds_load(const u32 *l, vu i) { return l[i]; }          // per-lane word
