// tests/emu/simt/emu_chain.cpp -- TEST INFRASTRUCTURE: runs the SOURCE of the chaining kernel
// (falcon_amd/csrc/k_chain.hip) on the host through the SIMT emulator of simt.h: one pile (seed +
// reads, packed like a batch), the seed's index from the emulated k_seed_index (or handed in), every
// read's window out.  Built by tests/emu/Makefile into libemu_chain.so, driven by
// tests/test_emu_chain.py against the CPU oracle; the product never loads it.
#include <sys/mman.h>
#include <unistd.h>

#include <vector>

#include "k_chain.hip"

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
        if (map == (char *)MAP_FAILED) { perror("emu_chain: mmap"); abort(); }
        mprotect(map, pg, PROT_NONE);
        mprotect(map + pg + body, pg, PROT_NONE);
        p = map + pg + body - ((bytes + 15) & ~(size_t)15);
        memset(map + pg, fill, body);
    }
    ~Guarded() { if (map) munmap(map, map_bytes); }
};
}  // namespace

// One pile.  words: all sequences packed back to back (each 16-byte aligned, two zero words behind
// it); woff / len [n_seq]: word offset and length of every sequence, sequence 0 the seed; T (65537)
// and P (seed_len - 8): the seed's index; lds_bins: bins the block's LDS holds (the host's bound);
// out[n_seq]: the windows; probe_out (optional): the records of all reads back to back, probe_off
// [n_seq] where each read's start.  Returns the rendezvous count.
extern "C" long long emu_chain(const u32 *words, long long n_words, const u64 *woff, const int *len, int n_seq,
                               const u32 *T, const u32 *P, int lds_bins, FaRange *out, u64 *probe_out,
                               const u64 *probe_off, long long n_probe_words) {
    Guarded gw, gs, gp, gt, gpos, go, gr, gpr, gpo, glds;
    gw.alloc((size_t)n_words * 4, 0);
    memcpy(gw.p, words, (size_t)n_words * 4);
    gs.alloc((size_t)n_seq * sizeof(FaSeq), 0);
    FaSeq *seq = (FaSeq *)gs.p;
    for (int g = 0; g < n_seq; g++) seq[g] = FaSeq{(u32)woff[g], len[g], 0, g};
    gp.alloc(sizeof(FaPile), 0);
    FaPile *pile = (FaPile *)gp.p;
    pile->first = 0;
    pile->n_seq = n_seq;
    pile->seed_len = len[0];
    const int n_pos = std::max(0, len[0] - FA_K);
    gt.alloc((size_t)(FA_NKMER + 1) * 4, 0);
    memcpy(gt.p, T, (size_t)(FA_NKMER + 1) * 4);
    gpos.alloc((size_t)std::max(1, n_pos) * 4, 0);
    memcpy(gpos.p, P, (size_t)n_pos * 4);
    go.alloc((size_t)n_seq * 4, 0);
    for (int g = 0; g < n_seq; g++) ((int *)go.p)[g] = g;
    gr.alloc((size_t)n_seq * sizeof(FaRange), 0xee);
    gpr.alloc((size_t)(n_probe_words + 8) * 8, 0xee);
    gpo.alloc((size_t)n_seq * 8, 0);
    memcpy(gpo.p, probe_off, (size_t)n_seq * 8);
    ChainArgs A;
    A.words = (const u32 *)gw.p; A.seq = seq; A.pile = pile; A.kidx = (const u32 *)gt.p; A.kpos = (const u32 *)gpos.p;
    A.order = (const int *)go.p; A.n_seq = n_seq; A.out = (FaRange *)gr.p;
    A.probe = (u64 *)gpr.p; A.probe_off = (const u64 *)gpo.p;
    A.lds_bins = (lds_bins + 3) & ~3;
    glds.alloc((size_t)A.lds_bins * 2 * 4, 0xa5);
    simt::g_dyn_lds = glds.p;
    simt::g_wave.n_sync = 0;
    simt::launch("k_chain", (unsigned)n_seq, [&] { k_chain(A); });
    simt::g_dyn_lds = nullptr;
    memcpy(out, gr.p, (size_t)n_seq * sizeof(FaRange));
    if (probe_out) memcpy(probe_out, gpr.p, (size_t)n_probe_words * 8);
    return (long long)simt::g_wave.n_sync;
}
