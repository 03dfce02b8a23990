// tests/emu/simt/fa_device.h -- TEST INFRASTRUCTURE: host twin of falcon_amd/csrc/fa_device.h for
// the SIMT emulator (simt.h).  The product's header maps these names onto single instructions
// and inline assembly; here they are the same functions over the emulator's rendezvous
// primitives, so that k_msa.hip and k_score2.hip compile unchanged with g++.  Only the helpers
// the consensus-stage kernels use.
#ifndef FA_DEVICE_H
#define FA_DEVICE_H
#include "fa_internal.h"

#define FA_WAVE 64
#define FA_EMU 1

inline int fa_lane() { return simt::lane(); }
inline u32 fa_base_at(const u32 *w, int i) { return (w[i >> 4] >> ((i & 15) * 2)) & 3u; }
inline u64 fa_window64(const u32 *w, int i) {
    const u64 lo = w[i >> 4], hi = w[(i >> 4) + 1];
    return ((hi << 32) | lo) >> ((i & 15) * 2);
}
inline u32 fa_kmer8(const u32 *w, int i) { return (u32)(fa_window64(w, i) & 0xFFFFu); }
#define fa_ballot(p) simt::ballot((p), __LINE__)
#define fa_uni(v) simt_uni((v), __LINE__)
inline int simt_uni(int v, int site) { return simt::readfirstlane(v, site); }
inline u32 simt_uni(u32 v, int site) { return (u32)simt::readfirstlane((int)v, site); }
inline u64 simt_uni(u64 v, int site) {
    const u32 lo = (u32)simt::readfirstlane((int)(u32)v, site), hi = (u32)simt::readfirstlane((int)(u32)(v >> 32), site);
    return ((u64)hi << 32) | lo;
}
template <class T> inline T *simt_uni(T *p, int site) { return (T *)simt_uni((u64)(uintptr_t)p, site); }

inline u64 fa_lane_range(int lo, int n) { return (n >= 64 ? ~0ull : ((1ull << n) - 1ull)) << lo; }
inline int fa_sel(u64 mask, int a, int b) { return ((mask >> simt::lane()) & 1ull) ? b : a; }
inline u32 fa_settled(u32 v) { return v; }
#define fa_wave_sync() simt::sync(__LINE__)
#define fa_lds_order() simt::sync(__LINE__)

inline int simt_wave_max(int v, int site) {
    const simt::X x = simt::xchg((uint32_t)v, site);
    int r = -0x7fffffff - 1;
    for (int l = 0; l < 64; l++)
        if ((x.act >> l) & 1ull) r = std::max(r, (int)(uint32_t)x.v[l]);
    return r;
}
#define fa_wave_max(v) simt_wave_max((v), __LINE__)
#define fa_wave_min(v) (-simt_wave_max(-(v), __LINE__))

// inclusive prefix maximum of unsigned keys inside each row of 16 lanes / over all 64
inline u32 simt_prefix_max(u32 v, bool rows, int site) {
    const simt::X x = simt::xchg(v, site);
    const int me = simt::lane();
    u32 r = 0;
    for (int l = rows ? (me & ~15) : 0; l <= me; l++)
        if ((x.act >> l) & 1ull) r = std::max(r, (u32)x.v[l]);
    return r;
}
#define fa_row_prefix_max_u32(v) simt_prefix_max((v), true, __LINE__)
#define fa_wave_prefix_max_u32(v) simt_prefix_max((v), false, __LINE__)

inline void simt_writelane2(int &a, int &b, int sa, int sb, int l, int site) {
    simt::sync(site);  // (wave-uniform operands: nothing to exchange)
    if (simt::lane() == l) { a = sa; b = sb; }
}
#define fa_writelane2(a, b, sa, sb, l) simt_writelane2((a), (b), (sa), (sb), (l), __LINE__)
#endif  // FA_DEVICE_H
