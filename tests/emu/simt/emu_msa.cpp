// tests/emu/simt/emu_msa.cpp -- TEST INFRASTRUCTURE: runs the SOURCE of the consensus-stage kernels
// (falcon_amd/csrc/k_msa.hip: k_tags, k_tscan, k_links, k_backtrace; k_score2.hip: k_score2) on the
// host through the SIMT emulator of simt.h, from the alignment stage's outputs (edit scripts,
// alignment summaries, windows) to the consensus strings.  The host plan between the alignment
// and the MSA stage (pool offsets, the segment list) is restated here from engine.hip's
// msa_stage.  Every buffer sits right before an inaccessible page, so a kernel that runs off
// the end of one dies on the spot.  Built by tests/emu/Makefile into libemu_msa.so, driven by
// tests/test_emu_msa.py against the CPU oracle; the product never loads it.
#include <sys/mman.h>
#include <unistd.h>

#include <vector>

#include "k_msa.hip"
#include "k_links2.hip"
#include "k_score2.hip"

#include <signal.h>

namespace {
struct Guarded;
static Guarded *g_bufs[64];
static int g_nbufs = 0;
struct Guarded {
    char *map = nullptr;
    size_t map_bytes = 0, bytes = 0;
    void *p = nullptr;
    const char *name = "";
    void alloc(size_t bytes_, int fill) {
        const size_t bytes = bytes_;
        this->bytes = bytes;
        if (g_nbufs < 64) g_bufs[g_nbufs++] = this;
        const size_t pg = (size_t)sysconf(_SC_PAGESIZE);
        const size_t body = (bytes + pg - 1) / pg * pg;
        map_bytes = body + 2 * pg;
        map = (char *)mmap(nullptr, map_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (map == (char *)MAP_FAILED) { perror("emu_msa: mmap"); abort(); }
        mprotect(map, pg, PROT_NONE);
        mprotect(map + pg + body, pg, PROT_NONE);
        // (ends at the guard page; 16-byte aligned like the device allocations)
        p = map + pg + body - ((bytes + 15) & ~(size_t)15);
        memset(map + pg, fill, body);
    }
    ~Guarded() {
        if (map) munmap(map, map_bytes);
        for (int i = 0; i < g_nbufs; i++) if (g_bufs[i] == this) g_bufs[i] = g_bufs[--g_nbufs];
    }
};
// a kernel that leaves its buffers: say which kernel, block, lane and buffer
static void on_segv(int, siginfo_t *si, void *) {
    const char *a = (const char *)si->si_addr;
    fprintf(stderr, "emu_msa: %s block %u lane %d touched %p", simt::g_cw->kernel, simt::g_cw->block, simt::g_cw->cur, (void *)a);
    for (int i = 0; i < g_nbufs; i++) {
        const Guarded *g = g_bufs[i];
        if (a >= g->map && a < g->map + g->map_bytes)
            fprintf(stderr, ": %ld bytes past the start of buffer #%d (%zu bytes)", (long)(a - (const char *)g->p), i, g->bytes);
    }
    fprintf(stderr, "\n");
    _exit(99);
}
template <class T>
struct GBuf : Guarded {
    T *get() { return (T *)p; }
    GBuf(size_t n, int fill = 0xA5) { alloc(std::max<size_t>(n, 1) * sizeof(T), fill); }
};
}  // namespace

// what = bit 0: stop after k_links (graph only); score_out / nodes (optional) receive the
// per-pile score records and the node pool (node_cap records, piles back to back as planned:
// pile[p].node_off is filled in).
extern "C" int emu_msa(const u32 *words_in, u64 n_words, const FaSeq *seq_in, int n_seq, FaPile *pile_io, int n_pile,
                       const FaRange *range_in, const FaAln *aln_in, const u32 *script_in, u64 n_script,
                       const u64 *script_off_in, unsigned min_cov, int first_links_back, char *out_seq, int *out_eqv,
                       u64 out_slots, FaPileOut *pile_out, FaScoreOut *score_out, FaNode *nodes_out, u64 nodes_cap,
                       unsigned long long *n_sync_out, FaTInfo *tinfo_out, u32 *links_out, u64 links_cap, u16 *nlk_out,
                       int *todo_out) {
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = on_segv;
    sa.sa_flags = SA_SIGINFO;
    sigaction(SIGSEGV, &sa, nullptr);
    // ---- the host plan (engine.hip: fa_batch_create / fa_batch_submit / msa_stage)
    std::vector<FaTagAln> ta;
    std::vector<u32> acc_first(n_pile + 1);
    std::vector<u64> link_off(n_pile), link_cap(n_pile), t_off(n_pile);
    std::vector<int> seg_pile, seg_t0;
    std::vector<u32> seg_first(n_pile + 1);
    u64 node_off = 0, desc_tot = 4, ins_tot = 0, link_tot = 0, t_tot = 0, out = 0;
    for (int p = 0; p < n_pile; p++) {
        FaPile &pm = pile_io[p];
        u64 levels = (u64)pm.seed_len + 2, cols = 8;
        acc_first[p] = (u32)ta.size();
        for (int j = 1; j < pm.n_seq; j++) {
            const int g = pm.first + j;
            const FaAln &al = aln_in[g];
            if (!al.accept) continue;
            levels += (u64)al.n_ins;
            cols += (u64)al.size;
            FaTagAln x;
            x.desc_off = desc_tot + 1;
            x.ins_off = (u32)ins_tot;
            x.s2 = range_in[g].s2;
            x.g = g;
            x.pile = p;
            x.pad = 0;
            ta.push_back(x);
            desc_tot += (u64)al.t_e + 3;
            ins_tot += (u64)al.n_ins + 4;
        }
        pm.node_off = node_off;
        pm.node_cap = levels * 5;
        node_off += pm.node_cap;
        link_off[p] = link_tot;
        link_cap[p] = cols;
        link_tot += cols;
        t_off[p] = t_tot;
        seg_first[p] = (u32)seg_pile.size();
        t_tot += (u64)pm.seed_len;
        for (int t0 = 0; t0 < pm.seed_len; t0 += TSEG) { seg_pile.push_back(p); seg_t0.push_back(t0); }
        pm.out_off = out;
        out += 2 * (u64)pm.seed_len + 4;
    }
    acc_first[n_pile] = (u32)ta.size();
    seg_first[n_pile] = (u32)seg_pile.size();
    if (out > out_slots || (nodes_out && node_off > nodes_cap)) return -2;
    const size_t n_ta = ta.size(), n_seg = seg_pile.size();

    // ---- "device" buffers
    GBuf<u32> d_words(n_words), d_script(n_script + 8), d_desc(desc_tot + 8), d_links(link_tot + 8), d_acc_first(n_pile + 1);
    GBuf<FaSeq> d_seq(n_seq);
    GBuf<FaPile> d_pile(n_pile);
    GBuf<FaRange> d_range(n_seq);
    GBuf<FaAln> d_aln(n_seq);
    GBuf<u64> d_script_off(n_seq), d_t_off(n_pile + 1), d_link_off(n_pile), d_link_cap(n_pile);
    GBuf<FaTagAln> d_ta(n_ta + 1);
    GBuf<int> d_tcov(n_ta + 1), d_seg_cnt(2 * n_seg + 2, 0), d_seg_pile(n_seg + 1), d_seg_t0(n_seg + 1), d_wide(6 * (n_seg + 1) + 1, 0);
    GBuf<u32> d_seg_base(2 * n_seg + 2), d_seg_first(n_pile + 1);
    GBuf<unsigned long long> d_bound(n_pile + 1);
    GBuf<uint8_t> d_insb(ins_tot + 8);
    GBuf<FaTInfo> d_tinfo(t_tot + 8);
    GBuf<u16> d_lvl_nlink(node_off / 5 + 8);
    GBuf<FaScoreOut> d_score_out(n_pile);
    GBuf<FaNode> d_nodes(node_off + 8);
    GBuf<char> d_out_seq(out + 8);
    GBuf<int> d_out_eqv(out + 8);
    GBuf<FaPileOut> d_pile_out(n_pile);
    memcpy(d_words.get(), words_in, n_words * sizeof(u32));
    memcpy(d_script.get(), script_in, n_script * sizeof(u32));
    memcpy(d_seq.get(), seq_in, (size_t)n_seq * sizeof(FaSeq));
    memcpy(d_pile.get(), pile_io, (size_t)n_pile * sizeof(FaPile));
    memcpy(d_range.get(), range_in, (size_t)n_seq * sizeof(FaRange));
    memcpy(d_aln.get(), aln_in, (size_t)n_seq * sizeof(FaAln));
    memcpy(d_script_off.get(), script_off_in, (size_t)n_seq * sizeof(u64));
    memcpy(d_t_off.get(), t_off.data(), (size_t)n_pile * sizeof(u64));
    memcpy(d_link_off.get(), link_off.data(), (size_t)n_pile * sizeof(u64));
    memcpy(d_link_cap.get(), link_cap.data(), (size_t)n_pile * sizeof(u64));
    memcpy(d_acc_first.get(), acc_first.data(), (size_t)(n_pile + 1) * sizeof(u32));
    memcpy(d_seg_first.get(), seg_first.data(), (size_t)(n_pile + 1) * sizeof(u32));
    if (n_ta) memcpy(d_ta.get(), ta.data(), n_ta * sizeof(FaTagAln));
    if (n_seg) {
        memcpy(d_seg_pile.get(), seg_pile.data(), n_seg * sizeof(int));
        memcpy(d_seg_t0.get(), seg_t0.data(), n_seg * sizeof(int));
    }

    MsaArgs A;
    memset(&A, 0, sizeof(A));
    A.words = d_words.get(); A.seq = d_seq.get(); A.pile = d_pile.get(); A.range = d_range.get(); A.aln = d_aln.get();
    A.script = d_script.get(); A.script_off = d_script_off.get();
    A.ta = d_ta.get(); A.acc_first = d_acc_first.get(); A.n_acc_total = (int)n_ta; A.n_pile = n_pile;
    A.tcov = d_tcov.get(); A.desc = d_desc.get(); A.insb = d_insb.get(); A.seg_cnt = d_seg_cnt.get(); A.seg_base = d_seg_base.get();
    A.seg_first = d_seg_first.get(); A.bound = d_bound.get(); A.t_off = d_t_off.get();
    A.tinfo = d_tinfo.get(); A.links = d_links.get(); A.link_off = d_link_off.get(); A.link_cap = d_link_cap.get();
    A.lvl_nlink16 = d_lvl_nlink.get(); A.nodes = d_nodes.get();
    A.score_ovf = nullptr; A.score_out = d_score_out.get();
    A.out_seq = d_out_seq.get(); A.out_eqv = d_out_eqv.get(); A.pile_out = d_pile_out.get();
    A.seg_pile = d_seg_pile.get(); A.seg_t0 = d_seg_t0.get(); A.n_seg = (int)n_seg; A.min_cov = min_cov;
    A.wide_count = d_wide.get(); A.wide_list = d_wide.get() + 1;
    A.first_links_back = first_links_back;
    A.force_generic = 0; A.only_redo = 0;

    simt::g_wave.n_sync = 0;
    if (n_ta) simt::launch("k_tags", (unsigned)n_ta, [&] { k_tags(A); });
    simt::launch("k_sscan", (unsigned)n_pile, [&] { k_sscan(A); });
    if (n_seg) {
        A.links_old = getenv("EMU_MSA_LINKS1") ? 1 : 0;
        simt::launch("k_links2", (unsigned)n_seg, [&] { k_links2(A); });
        simt::launch("k_links2_big", (unsigned)std::min<size_t>(n_seg, 64), [&] { k_links2_big(A); });
        const unsigned wide_grid = (unsigned)std::min<size_t>(n_seg, 64);
        simt::launch("k_links<1>", wide_grid, [&] { k_links<1>(A); });
        simt::launch("k_links<2>", wide_grid, [&] { k_links<2>(A); });
        simt::launch("k_links<4>", wide_grid, [&] { k_links<4>(A); });
        simt::launch("k_links<8>", wide_grid, [&] { k_links<8>(A); });
        simt::launch("k_links<16>", wide_grid, [&] { k_links<16>(A); });
    }
    simt::launch_waves("k_score2", (unsigned)n_pile, 2, [&] { k_score2(A); });
    if (!getenv("EMU_MSA_NO_BACKTRACE")) simt::launch("k_backtrace", (unsigned)n_pile, [&] { k_backtrace(A); });

    memcpy(out_seq, d_out_seq.get(), out);
    memcpy(out_eqv, d_out_eqv.get(), out * sizeof(int));
    memcpy(pile_out, d_pile_out.get(), (size_t)n_pile * sizeof(FaPileOut));
    if (score_out) memcpy(score_out, d_score_out.get(), (size_t)n_pile * sizeof(FaScoreOut));
    if (nodes_out) memcpy(nodes_out, d_nodes.get(), node_off * sizeof(FaNode));
    if (n_sync_out) *n_sync_out = simt::g_wave.n_sync;
    if (todo_out) for (int l = 0; l < 6; l++) todo_out[l] = d_wide.get()[l * (n_seg + 1)];  // segments each k_links instance took
    // (debug views of the graph: position records, link words -- piles back to back at
    // link_cap = columns + 8 each, as planned -- and the links per level slot)
    if (tinfo_out) memcpy(tinfo_out, d_tinfo.get(), t_tot * sizeof(FaTInfo));
    if (links_out && link_tot <= links_cap) memcpy(links_out, d_links.get(), link_tot * sizeof(u32));
    if (nlk_out) memcpy(nlk_out, d_lvl_nlink.get(), (node_off / 5) * sizeof(u16));
    return 0;
}
