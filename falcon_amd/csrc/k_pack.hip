// k_pack.hip -- ASCII -> 2-bit packing of a whole batch on the device (FALCON_AMD_DEVICE_PACK; the
// default packs on the host, pack_host.cpp).  Replaces the per-call ASCII -> code loops of the
// reference (src/c/kmer_lookup.c:159-171, :236-249).  HBM streaming work: byte in, quarter byte out.
#include "fa_device.h"

// --------------------------------------------------------------------------
// pack: one thread produces one u32 (16 bases) from one aligned 16-byte load.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack(const uint8_t *__restrict__ ascii,
                                              const u64 *__restrict__ ascii_off,
                                              const FaSeq *__restrict__ seq, int n_seq,
                                              u32 *__restrict__ words, u64 n_words,
                                              int *__restrict__ first_bad, int *__restrict__ bad_pile) {
    u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    // which sequence owns word w: last g with seq[g].woff <= w
    int lo = 0, hi = n_seq - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if ((u64)seq[mid].woff <= w) lo = mid; else hi = mid - 1;
    }
    const FaSeq s = seq[lo];
    u32 wi = (u32)(w - s.woff);
    int base0 = (int)wi * 16;
    u32 out = 0;
    bool bad = false;
    if (base0 < s.len) {
        const uint4 v = *reinterpret_cast<const uint4 *>(ascii + ascii_off[lo] + (u64)base0);
        const u32 q[4] = {v.x, v.y, v.z, v.w};
        int valid = min(16, s.len - base0);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            u32 c = (q[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
            // 'A'->0 'C'->1 'G'->2 'T'->3
            u32 code = ((c >> 1) ^ (c >> 2)) & 3u;
            if (j < valid) {
                out |= code << (2 * j);
                // anything but upper-case ACGT is outside the parity domain: the reference
                // aligns raw characters and codes other bytes 0xff in the seed index and 0
                // in reads (kmer_lookup.c:159-171, :236-249) -- refuse it, never fold it
                const bool acgt = (c & 0xE0u) == 0x40u && ((0x0010008Au >> (c & 31u)) & 1u);
                bad |= !acgt;
            }
        }
    }
    words[w] = out;
    if (bad) {
        atomicMin(first_bad, lo);
        if (bad_pile) bad_pile[s.pile] = 1;  // (piles fail alone: the engine takes this one out)
    }
}

void fa_launch_pack(const FaBatchDev &b, int *first_bad, int *bad_pile, hipStream_t s) {
    if (b.n_words == 0) return;
    unsigned grid = (unsigned)((b.n_words + 255) / 256);
    hipLaunchKernelGGL(k_pack, dim3(grid), dim3(256), 0, s, b.ascii, b.ascii_off, b.seq, b.n_seq,
                       b.words, b.n_words, first_bad, bad_pile);
}
