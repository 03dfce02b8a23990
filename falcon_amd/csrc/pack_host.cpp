// pack_host.cpp -- ASCII -> 2 bits per base on the HOST, straight into the pinned buffer the
// batch's packed words are uploaded from (no HIP in this file).
//
// What it replaces: the per-call ASCII -> code loops of the reference
// (src/c/kmer_lookup.c:159-171, :236-249).  A batch used to travel as 1 byte per base
// (memcpy into pinned memory, PCIe, k_pack on the device); packed on the way into the pinned
// buffer it is a quarter of the host writes and of the PCIe bytes, and the device never sees
// the text.  Layout = k_pack's: 16 bases per u32, base i at bits 2 (i mod 16), A C G T =
// 0 1 2 3, zero words behind every sequence.
#include <cstdint>
#include <cstring>

#include "fa_host.h"

namespace {

// eight bases at once.  code = (bit 2, bit 1 ^ bit 2) of the character:
//   'A' 0x41 -> 0   'C' 0x43 -> 1   'G' 0x47 -> 2   'T' 0x54 -> 3
// A byte is one of the four iff, bits 1 and 2 cleared, it reads 0x41 -- or 0x50 when (bit 2,
// bit 1) = (1, 0), the pattern of 'T' ('E' 0x45 shares that pattern and fails the test).
inline uint32_t pack8(uint64_t x, uint64_t &bad) {
    const uint64_t ones = 0x0101010101010101ull;
    const uint64_t t_like = (x >> 2) & ~(x >> 1) & ones;
    bad |= (x & 0xF9F9F9F9F9F9F9F9ull) ^ (0x4141414141414141ull + t_like * 0x0F);
    uint64_t t = ((x >> 1) ^ (x >> 2)) & 0x0303030303030303ull;
    t = (t | (t >> 6)) & 0x000F000F000F000Full;
    t = (t | (t >> 12)) & 0x000000FF000000FFull;
    t = (t | (t >> 24)) & 0xFFFFull;
    return (uint32_t)t;
}

inline bool is_acgt(unsigned char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

}  // namespace

// Packs s[0 .. len) into out[0 .. n_out) (n_out >= ceil(len / 16); the words behind the
// sequence are zeroed).  Returns the position of the first byte that is not an upper-case
// A, C, G or T, or -1 (the packed words are then meaningless from that word on).
int fa_pack_host(const char *s, int len, uint32_t *out, long long n_out) {
    const int full = len / 16;
    uint64_t bad = 0;
    int first_bad_word = -1;
    for (int w = 0; w < full; w++) {
        uint64_t lo, hi;
        memcpy(&lo, s + 16 * w, 8);
        memcpy(&hi, s + 16 * w + 8, 8);
        uint64_t b = 0;
        out[w] = pack8(lo, b) | (pack8(hi, b) << 16);
        if (b && first_bad_word < 0) first_bad_word = w;
        bad |= b;
    }
    long long w = full;
    const int rest = len - 16 * full;
    bool bad_tail = false;
    if (rest > 0) {
        uint32_t v = 0;
        for (int j = 0; j < rest; j++) {
            const unsigned char c = (unsigned char)s[16 * full + j];
            v |= (uint32_t)(((c >> 1) ^ (c >> 2)) & 3u) << (2 * j);
            bad_tail |= !is_acgt(c);
        }
        out[w++] = v;
    }
    if (w < n_out) memset(out + w, 0, (size_t)(n_out - w) * sizeof(uint32_t));
    if (!bad && !bad_tail) return -1;
    const int from = first_bad_word >= 0 ? 16 * first_bad_word : 16 * full;
    for (int i = from; i < len; i++)
        if (!is_acgt((unsigned char)s[i])) return i;
    return -1;  // (not reached)
}
