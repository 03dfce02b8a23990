// tests/emu/simt/emu_index.cpp -- TEST INFRASTRUCTURE: runs the SOURCE of the seed-index kernel
// (falcon_amd/csrc/k_seed_index.hip) on the host through the SIMT emulator of simt.h.  The emulator
// runs the workgroup's 16 wavefronts one after the other from barrier to barrier; the kernels' own
// `__shared__` arrays are the LDS, T and P end at inaccessible pages.  Built by tests/emu/Makefile into libemu_index.so, driven by
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
// P: max(0, len - 8) entries out.  use_long: the kernel behind (k_seed_index_long, every_pile = 1)
// instead of the one a seed of this length gets.  Returns the rendezvous count (a cost figure).
extern "C" long long emu_seed_index(const u32 *words, int len, u32 *T_out, u32 *P_out, int use_long) {
    const int n_pos = std::max(0, len - FA_K), n_words = (len + 15) / 16 + 2;
    Guarded gw, gt, gp, gs, gpl;
    gw.alloc((size_t)n_words * 4, 0);
    gt.alloc((size_t)FA_IDX_STRIDE * 4, 0xee);
    gp.alloc((size_t)std::max(1, n_pos) * 4, 0xee);
    memcpy(gw.p, words, (size_t)n_words * 4);
    gs.alloc(sizeof(FaSeq), 0);
    *(FaSeq *)gs.p = FaSeq{0u, len, 0, 0};
    gpl.alloc(sizeof(FaPile), 0);
    FaPile *pm = (FaPile *)gpl.p;
    pm->first = 0; pm->n_seq = 1; pm->seed_len = len; pm->kidx_off = 0; pm->kpos_off = 0;
    simt::g_wave.n_sync = 0;
    const bool is_long = len - FA_K > SI_MAX_POS;
    // (both kernels are launched on every batch that holds a long seed; each leaves the other's piles alone)
    if (!use_long)
        simt::launch_waves("k_seed_index", 1, SI_NW, [&] {
            k_seed_index((const u32 *)gw.p, (const FaSeq *)gs.p, pm, (u32 *)gt.p, (u32 *)gp.p);
        });
    if (use_long || is_long)
        simt::launch_waves("k_seed_index_long", 1, SL_NT / 64, [&] {
            k_seed_index_long((const u32 *)gw.p, (const FaSeq *)gs.p, pm, (u32 *)gt.p, (u32 *)gp.p, use_long);
        });
    memcpy(T_out, gt.p, (size_t)(FA_NKMER + 1) * 4);
    memcpy(P_out, gp.p, (size_t)n_pos * 4);
    return (long long)simt::g_wave.n_sync;
}
