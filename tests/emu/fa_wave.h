// tests/emu/fa_wave.h -- TEST INFRASTRUCTURE: a host-side twin of falcon_amd/csrc/fa_wave.h.
//
// Implements the wave64 primitives k_align2_core.h is written against on 64-element arrays,
// so that the kernel's source -- its control logic: two alignments per wavefront, band
// placement, parking, the iteration tape, the trace-back -- can be compiled with g++ and
// run against the CPU oracle in the `-m "not gpu"` suite (tests/test_emu_align2.py).  It is
// never compiled into libfalcon_amd.so, and nothing in the product includes it: the
// product's fa_wave.h maps the same names onto registers and single instructions.
//
// Semantics: `vi` / `vu` hold one value per lane; assignment and the compound operators only
// touch the lanes of the current exec mask (W_WHERE narrows it, like s_and_saveexec);
// ballots see only executing lanes.
#pragma once
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include "fa_types.h"

namespace emu {
extern thread_local uint64_t exec;
struct ExecScope {
    uint64_t saved;
    bool once;
    explicit ExecScope(uint64_t m) : saved(exec), once(true) { exec &= m; }
    ~ExecScope() { exec = saved; }
};
inline bool on(int l) { return (exec >> l) & 1ull; }
}  // namespace emu

template <class T>
struct wvec {
    T v[64];
    wvec() { for (int l = 0; l < 64; l++) v[l] = 0; }
    wvec(T s) { for (int l = 0; l < 64; l++) v[l] = s; }
    wvec(const wvec &o) { memcpy(v, o.v, sizeof(v)); }
    template <class U>
    explicit wvec(const wvec<U> &o) { for (int l = 0; l < 64; l++) v[l] = (T)o.v[l]; }
    // writes obey the exec mask
    wvec &operator=(const wvec &o) {
        for (int l = 0; l < 64; l++) if (emu::on(l)) v[l] = o.v[l];
        return *this;
    }
};
typedef wvec<int32_t> vi;
typedef wvec<uint32_t> vu;
struct vb { bool v[64]; };

#define EMU_BIN(op)                                                                      \
    template <class T> inline wvec<T> operator op(const wvec<T> &a, const wvec<T> &b) {  \
        wvec<T> r; for (int l = 0; l < 64; l++) r.v[l] = (T)(a.v[l] op b.v[l]); return r; } \
    template <class T> inline wvec<T> operator op(const wvec<T> &a, T b) { return a op wvec<T>(b); } \
    template <class T> inline wvec<T> operator op(T a, const wvec<T> &b) { return wvec<T>(a) op b; }
EMU_BIN(+) EMU_BIN(-) EMU_BIN(*) EMU_BIN(&) EMU_BIN(|) EMU_BIN(^)
#undef EMU_BIN
// mixed int / unsigned literals against the other vector type
inline vu operator+(const vu &a, int b) { return a + vu((uint32_t)b); }
inline vu operator-(const vu &a, int b) { return a - vu((uint32_t)b); }
inline vu operator&(const vu &a, int b) { return a & vu((uint32_t)b); }
inline vu operator|(const vu &a, int b) { return a | vu((uint32_t)b); }
inline vu operator*(const vu &a, int b) { return a * vu((uint32_t)b); }
inline vi operator&(const vi &a, unsigned b) { return a & vi((int32_t)b); }
template <class T> inline wvec<T> operator-(const wvec<T> &a) {
    wvec<T> r; for (int l = 0; l < 64; l++) r.v[l] = (T)(0 - a.v[l]); return r; }
// shifts: counts are per lane or uniform; a count >= 32 would be undefined on the host --
// the kernel never relies on it, so it is caught here
template <class T> inline wvec<T> operator<<(const wvec<T> &a, const vu &s) {
    wvec<T> r; for (int l = 0; l < 64; l++) r.v[l] = (T)((uint32_t)a.v[l] << (s.v[l] & 31u)); return r; }
template <class T> inline wvec<T> operator>>(const wvec<T> &a, const vu &s) {
    wvec<T> r; for (int l = 0; l < 64; l++) r.v[l] = (T)(a.v[l] >> (s.v[l] & 31u)); return r; }
template <class T> inline wvec<T> operator<<(const wvec<T> &a, int s) { return a << vu((uint32_t)s); }
template <class T> inline wvec<T> operator>>(const wvec<T> &a, int s) { return a >> vu((uint32_t)s); }
#define EMU_CMP(op)                                                                     \
    template <class T> inline vb operator op(const wvec<T> &a, const wvec<T> &b) {      \
        vb r; for (int l = 0; l < 64; l++) r.v[l] = a.v[l] op b.v[l]; return r; }       \
    template <class T> inline vb operator op(const wvec<T> &a, T b) { return a op wvec<T>(b); }
EMU_CMP(<) EMU_CMP(<=) EMU_CMP(>) EMU_CMP(>=) EMU_CMP(==) EMU_CMP(!=)
#undef EMU_CMP
inline vb operator==(const vu &a, int b) { return a == vu((uint32_t)b); }
inline vb operator>=(const vu &a, int b) { return a >= vu((uint32_t)b); }

#define W_FN static inline
#define W_NOINLINE static
#define W_WHERE(m) for (emu::ExecScope scope_((m)); scope_.once; scope_.once = false)

W_FN vi w_lane() { vi r; for (int l = 0; l < 64; l++) r.v[l] = l; return r; }
W_FN u64 w_ballot(const vb &p) {
    u64 m = 0;
    for (int l = 0; l < 64; l++) if (p.v[l] && emu::on(l)) m |= 1ull << l;
    return m;
}
W_FN vi w_sel(u64 m, const vi &a, const vi &b) { vi r; for (int l = 0; l < 64; l++) r.v[l] = ((m >> l) & 1) ? b.v[l] : a.v[l]; return r; }
W_FN vu w_selu(u64 m, const vu &a, const vu &b) { vu r; for (int l = 0; l < 64; l++) r.v[l] = ((m >> l) & 1) ? b.v[l] : a.v[l]; return r; }
W_FN vi w_from_below(const vi &v) { vi r; for (int l = 0; l < 64; l++) r.v[l] = l > 0 ? v.v[l - 1] : 0; return r; }
W_FN vi w_from_above(const vi &v) { vi r; for (int l = 0; l < 64; l++) r.v[l] = l < 63 ? v.v[l + 1] : 0; return r; }
W_FN int w_readlane(const vi &v, int l) { return v.v[l & 63]; }
W_FN u32 w_readlaneu(const vu &v, int l) { return v.v[l & 63]; }
W_FN void w_writelane2(vu &a, vu &b, u32 sa, u32 sb, int l) { a.v[l & 63] = sa; b.v[l & 63] = sb; }
W_FN void w_setlane(vi &v, int s, int l) { v.v[l & 63] = s; }
W_FN vi w_push_lanes(const vi &v, const vi &dst) { vi r(0); for (int l = 0; l < 64; l++) r.v[dst.v[l] & 63] = v.v[l]; return r; }
W_FN vi w_gather_lanes(const vi &v, const vi &src) { vi r; for (int l = 0; l < 64; l++) r.v[l] = v.v[src.v[l] & 63]; return r; }
W_FN int w_uni(int v) { return v; }
W_FN u32 w_uniu(u32 v) { return v; }
W_FN vu w_prefix_max(const vu &v) {
    vu r; u32 m = 0;
    for (int l = 0; l < 64; l++) { m = std::max(m, v.v[l]); r.v[l] = m; }
    return r;
}
W_FN vu w_prefix_add(const vu &v) {
    vu r; u32 m = 0;
    for (int l = 0; l < 64; l++) { m += v.v[l]; r.v[l] = m; }
    return r;
}
// the trace-back's chain over the rows in lanes base .. 63: row l of the path sits on lane j of
// its iteration, the next on j + step[l] + bit j of the row's from_above mask
W_FN int w_chain(const vu &ra, const vu &rb, const vi &step, int j, int base, vi &my) {
    for (int l = base & 63; l < 64; l++) {
        my.v[l] = j;
        const u64 mask = ((u64)rb.v[l] << 32) | ra.v[l];
        j += step.v[l] + (int)((mask >> (j & 63)) & 1ull);
    }
    return j;
}
// ... in a frame of its own: i = lane + shift, the path moves by the shifted mask's bit i and -(l & 1)
W_FN int w_chain2(const vu &ra, const vu &rb, int i, int base, vi &my) {
    for (int l = base & 63; l < 64; l++) {
        my.v[l] = i;
        const u64 mask = ((u64)rb.v[l] << 32) | ra.v[l];
        i += (int)((mask >> (i & 63)) & 1ull) - (l & 1);
    }
    return i;
}
W_FN void w_shift64(const vu &lo, const vu &hi, const vi &s, vu &olo, vu &ohi) {
    for (int l = 0; l < 64; l++) {
        const u64 m = ((u64)hi.v[l] << 32) | lo.v[l];
        const u64 r = s.v[l] >= 0 ? (m << (s.v[l] & 63)) : (m >> ((-s.v[l]) & 63));
        olo.v[l] = (u32)r; ohi.v[l] = (u32)(r >> 32);
    }
}
W_FN vu w_row_tail(const vu &key, const vi &x, const vi &qlen, const vi &y, const vi &tlen, const vu &m, u64 act,
                   u32 band, u64 fa, int slot, vu &rc_lo, vu &rc_hi, u64 &fin, u64 &big, vu &keyb) {
    fin = 0; big = 0;
    for (int l = 0; l < 64; l++) {
        if (!((act >> l) & 1)) continue;
        if (x.v[l] >= qlen.v[l] || y.v[l] >= tlen.v[l]) fin |= 1ull << l;
        if (m.v[l] > 254u) big |= 1ull << l;
    }
    for (int l = 0; l < 64; l++) keyb.v[l] = key.v[l] + band;
    rc_lo.v[slot & 63] = (u32)fa;
    rc_hi.v[slot & 63] = (u32)(fa >> 32);
    return w_prefix_max(key);
}
W_FN vi w_min(const vi &a, const vi &b) { vi r; for (int l = 0; l < 64; l++) r.v[l] = std::min(a.v[l], b.v[l]); return r; }
W_FN vi w_max(const vi &a, const vi &b) { vi r; for (int l = 0; l < 64; l++) r.v[l] = std::max(a.v[l], b.v[l]); return r; }
W_FN vu w_minu(const vu &a, const vu &b) { vu r; for (int l = 0; l < 64; l++) r.v[l] = std::min(a.v[l], b.v[l]); return r; }
W_FN vu w_minu(const vu &a, u32 b) { return w_minu(a, vu(b)); }
W_FN vu w_ffbl(const vu &x) { vu r; for (int l = 0; l < 64; l++) r.v[l] = x.v[l] ? (u32)__builtin_ctz(x.v[l]) : 0xffffffffu; return r; }
W_FN vu w_alignbit(const vu &hi, const vu &lo, const vu &sh) {
    vu r;
    for (int l = 0; l < 64; l++) r.v[l] = (u32)(((((u64)hi.v[l]) << 32) | lo.v[l]) >> (sh.v[l] & 31u));
    return r;
}
W_FN u64 w_lanes(int lo, int n) {  // (s_bfm_b64: both operands are taken modulo 64)
    const u64 ones = (n & 63) ? ((1ull << (n & 63)) - 1ull) : 0ull;
    return ones << (lo & 63);
}
W_FN u64 w_bit_clr(u64 m, int b) { return m & ~(1ull << (b & 63)); }
W_FN u64 w_bit_set(u64 m, int b) { return m | (1ull << (b & 63)); }
W_FN int w_lowest(u64 m) { return m ? __builtin_ctzll(m) : -1; }
W_FN int w_highest(u64 m) { return m ? 63 - __builtin_clzll(m) : -1; }
W_FN int w_popc(u64 m) { return __builtin_popcountll(m); }
W_FN vi w_bit_at(u64 mask, const vi &j) { vi r; for (int l = 0; l < 64; l++) r.v[l] = (int)((mask >> (j.v[l] & 63)) & 1ull); return r; }
W_FN vi w_bfe_i16(const vu &v, const vu &off) { vi r; for (int l = 0; l < 64; l++) r.v[l] = (int)(short)((v.v[l] >> (off.v[l] & 31u)) & 0xffffu); return r; }
W_FN int w_span(u64 m) { return m ? 64 - (__builtin_clzll(m) + __builtin_ctzll(m)) : 0; }
W_FN vi w_rank_in(u64 m) { vi r; for (int l = 0; l < 64; l++) r.v[l] = __builtin_popcountll(m & ((1ull << l) - 1ull)); return r; }
W_FN vu w_undef() { return vu(0xdeadbeefu); }
template <int J>
W_FN vu w_put_byte(const vu &acc, const vu &m) {
    vu r;
    for (int l = 0; l < 64; l++) r.v[l] = (acc.v[l] & ~(0xffu << (8 * J))) | ((m.v[l] & 0xffu) << (8 * J));
    return r;
}

W_FN vu w_put_byte_sel(const vu &acc, const vu &m, u32 selector) {
    vu r;  // v_perm_b32: result byte i = byte (selector byte i) of {m (4..7), acc (0..3)}
    for (int l = 0; l < 64; l++) {
        const uint64_t both = ((uint64_t)m.v[l] << 32) | acc.v[l];
        u32 o = 0;
        for (int i = 0; i < 4; i++) o |= (u32)((both >> (8 * ((selector >> (8 * i)) & 7u))) & 0xffu) << (8 * i);
        r.v[l] = o;
    }
    return r;
}
template <int I> W_FN void w_pack_put(vu &p, u32 s) { p.v[I] = s; }
template <int I> W_FN u32 w_pack_get(const vu &p) { return p.v[I]; }

// ---- memory: loads and stores of the executing lanes; every access is bounds-checked
// against the buffers the harness registered
namespace emu {
struct Region { const void *p; size_t bytes; bool writable; };
extern thread_local Region regions[16];
extern thread_local int n_regions;
void check(const void *p, size_t bytes, bool write, const char *what);
// optional accounting (scripts/a2_bytes.py): one object per wave-wide access gathers what its lanes touch
struct Acct { Acct(); ~Acct(); };
}
W_FN void w_load_pair(const u32 *base, const vu &i, vu &lo, vu &hi) {
    emu::Acct acct_;
    for (int l = 0; l < 64; l++) if (emu::on(l)) {
        emu::check(base + i.v[l], 8, false, "w_load_pair");
        lo.v[l] = base[i.v[l]];
        hi.v[l] = base[i.v[l] + 1u];
    }
}
W_FN vu w_load32(const u32 *base, const vu &i) {
    emu::Acct acct_;
    vu r;
    for (int l = 0; l < 64; l++) if (emu::on(l)) { emu::check(base + i.v[l], 4, false, "w_load32"); r.v[l] = base[i.v[l]]; }
    return r;
}
W_FN void w_store32(u32 *base, const vu &i, const vu &v) {
    emu::Acct acct_;
    for (int l = 0; l < 64; l++) if (emu::on(l)) { emu::check(base + i.v[l], 4, true, "w_store32"); base[i.v[l]] = v.v[l]; }
}
W_FN void w_store64(u64 *base, const vu &i, const vu &lo, const vu &hi) {
    emu::Acct acct_;
    for (int l = 0; l < 64; l++) if (emu::on(l)) {
        emu::check(base + i.v[l], 8, true, "w_store64");
        base[i.v[l]] = ((u64)hi.v[l] << 32) | lo.v[l];
    }
}
W_FN void w_load64(const u64 *base, const vu &i, vu &lo, vu &hi) {
    emu::Acct acct_;
    for (int l = 0; l < 64; l++) if (emu::on(l)) {
        emu::check(base + i.v[l], 8, false, "w_load64");
        lo.v[l] = (u32)base[i.v[l]];
        hi.v[l] = (u32)(base[i.v[l]] >> 32);
    }
}
W_FN void w_store_x4(u32 *base16, const vu &i, const vu &a, const vu &b, const vu &c, const vu &d) {
    emu::Acct acct_;
    for (int l = 0; l < 64; l++) if (emu::on(l)) {
        u32 *p = base16 + 4 * (size_t)i.v[l];
        emu::check(p, 16, true, "w_store_x4");
        p[0] = a.v[l]; p[1] = b.v[l]; p[2] = c.v[l]; p[3] = d.v[l];
    }
}
W_FN void w_load_x4(const u32 *base16, const vu &i, vu &a, vu &b, vu &c, vu &d) {
    emu::Acct acct_;
    for (int l = 0; l < 64; l++) if (emu::on(l)) {
        const u32 *p = base16 + 4 * (size_t)i.v[l];
        emu::check(p, 16, false, "w_load_x4");
        a.v[l] = p[0]; b.v[l] = p[1]; c.v[l] = p[2]; d.v[l] = p[3];
    }
}
W_FN int w_atomic_add_lane0(int *p, int v) { const int old = *p; *p += v; return old; }
W_FN void w_fence_block() {}
W_FN void w_store_aln(FaAln *dst, const FaAln &r) { emu::check(dst, sizeof(FaAln), true, "w_store_aln"); *dst = r; }
W_FN void w_stat_add(unsigned long long *p, unsigned long long v) { if (p) *p += v; }
namespace emu { extern thread_local u32 lds[1024]; }
W_FN u32 *w_lds() { return emu::lds; }
W_FN void w_lds_store(u32 *l, const vu &i, const vu &v) { for (int k = 0; k < 64; k++) if (emu::on(k)) l[i.v[k]] = v.v[k]; }
W_FN vu w_lds_bcast(const u32 *l, int i) { return vu(l[i]); }

// what the core takes from <algorithm> / the device library under plain names
using std::max;
using std::min;
W_FN vu w_lds_load(const u32 *l, const vu &i) { vu r; for (int k = 0; k < 64; k++) if (emu::on(k)) r.v[k] = l[i.v[k]]; return r; }
