// tests/emu/simt/emu_index.cpp -- TEST INFRASTRUCTURE: runs the SOURCE of the seed-index kernel
// (falcon_amd/csrc/k_seed_index.hip) on the host through the SIMT emulator of simt.h.  The emulator
// holds one wavefront at a time, so the workgroup's 16 wavefronts go through every phase one after
// the other (si_pile<true>: one phase, no barrier); the "LDS" is three static arrays, T and P end at
// inaccessible pages.  Built by tests/emu/Makefile into libemu_index.so, driven by
// tests/test_emu_index.py against a numpy statement of kmer_lookup.c:140-192; the product never
// loads it.
#include <sys/mman.h>
#include <unistd.h>

#include "k_seed_index.hip"

namespace {
struct Guarded {
    char *map = nullptr;
    size_t map_bytes = 0;
    void *p = nullptr;
    void alloc(size_t bytes, int fill) {
        const size_t pg = (size_t)sysconf(_SC_PAGESIZE);
        const size_t body = (bytes + pg - 1) / pg * pg;
        map_bytes = body + 2 * pg;
        map = (char *)mmap(nullptr, map_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (map == (char *)MAP_FAILED) { perror("emu_index: mmap"); abort(); }
        mprotect(map, pg, PROT_NONE);
        mprotect(map + pg + body, pg, PROT_NONE);
        p = map + pg + body - ((bytes + 15) & ~(size_t)15);
        memset(map + pg, fill, body);
    }
    ~Guarded() { if (map) munmap(map, map_bytes); }
};
}  // namespace

// words: the seed's packed bases followed by two zero words (fa_internal.h); T: 65537 entries out;
// P: max(0, len - 8) entries out.  Returns the rendezvous count (a cost figure), -1 if the seed is
// the other kernel's.
extern "C" long long emu_seed_index(const u32 *words, int len, u32 *T_out, u32 *P_out) {
    if (len - FA_K > SI_MAX_POS) return -1;
    static __attribute__((aligned(16))) u32 cur[SI_WORDS];
    static u32 wtot[SI_NW];
    static u32 sw[SI_SEED_WORDS];
    memset(cur, 0xa5, sizeof cur);  // (the kernel zeroes what it uses)
    memset(sw, 0xa5, sizeof sw);
    const int n_pos = std::max(0, len - FA_K), n_words = (len + 15) / 16 + 2;
    Guarded gw, gt, gp;
    gw.alloc((size_t)n_words * 4, 0);
    gt.alloc((size_t)(FA_NKMER + 1) * 4, 0xee);
    gp.alloc((size_t)std::max(1, n_pos) * 4, 0xee);
    memcpy(gw.p, words, (size_t)n_words * 4);
    simt::g_wave.n_sync = 0;
    for (int phase = 0; phase < 5; phase++)
        for (int wv = 0; wv < SI_NW; wv++)
            simt::launch("k_seed_index", 1, [&] {
                si_pile<true>(phase, cur, wtot, sw, (const u32 *)gw.p, len, (u32 *)gt.p, (u32 *)gp.p,
                              wv * 64 + simt::lane());
            });
    memcpy(T_out, gt.p, (size_t)(FA_NKMER + 1) * 4);
    memcpy(P_out, gp.p, (size_t)n_pos * 4);
    return (long long)simt::g_wave.n_sync;
}
