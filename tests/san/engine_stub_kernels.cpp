// tests/san/engine_stub_kernels.cpp -- TEST INFRASTRUCTURE: stand-ins for the kernel launchers the batch engine calls
// (falcon_amd/csrc/fa_internal.h), for the sanitizer build of the engine's host logic (tests/san/engine_san.cpp).  They
// compute nothing: every read is left without a window (range[g].ok = 0), so no alignment is accepted and every pile's
// consensus is empty -- the engine's whole control flow (staging, the front half, the planner's sizing of the MSA stage,
// the back half, fetch, results, freeing) runs, on the threads it runs on in the product.
#include <cstring>
#include "fa_internal.h"

u64 fa_align2_slot_words(u32 ring) { return (u64)ring * 20u + 2048u; }
u32 fa_align2_ring_for(int) { return 1024; }
size_t fa_align2_lds_bytes() { return 4096; }
int fa_align2_blocks_per_cu() { return 2; }
size_t fa_align_lds_bytes(int, int) { return 1024; }
int fa_align_blocks_per_cu(size_t) { return 2; }
void fa_launch_pack(const FaBatchDev &, int *, int *, hipStream_t) {}
void fa_launch_index(const FaBatchDev &, int, hipStream_t) {}
void fa_launch_chain(const FaBatchDev &b, int, hipStream_t) { memset(b.range, 0, (size_t)b.n_seq * sizeof(FaRange)); }
static void no_alignment(const FaBatchDev &b) { memset(b.aln, 0, (size_t)b.n_seq * sizeof(FaAln)); }
void fa_launch_align2(const FaBatchDev &b, const FaAlign2Arena &, double, int, const int *, int, u32, hipStream_t) { no_alignment(b); }
void fa_launch_align_wide(const FaBatchDev &b, const FaAlignArena &, double, int, hipStream_t) { no_alignment(b); }
void fa_launch_align(const FaBatchDev &b, const FaAlignArena &, int, int, double, hipStream_t) { no_alignment(b); }
void fa_launch_align_band(const FaBatchDev &b, const FaAlignArena &, int, int, double, int, hipStream_t) { no_alignment(b); }
void fa_launch_align_list(const FaBatchDev &, const FaAlignArena &, int, int, double, int, const int *, int, hipStream_t) {}
void fa_launch_trimwin(const FaBatchDev &, int, int *, u32 *, u64, int, int, hipStream_t) {}
void fa_launch_msa_front(const FaBatchDev &, const FaMsaDev &, unsigned, hipStream_t s, hipEvent_t a, hipEvent_t b) {
    if (a) hipEventRecord(a, s);
    if (b) hipEventRecord(b, s);
}
void fa_launch_msa_back(const FaBatchDev &b, const FaMsaDev &, unsigned, hipStream_t s, hipEvent_t a, hipEvent_t c) {
    memset(b.pile_out, 0, (size_t)b.n_pile * sizeof(FaPileOut));
    if (a) hipEventRecord(a, s);
    if (c) hipEventRecord(c, s);
}
void fa_touch_index() {}
void fa_touch_chain() {}
void fa_touch_align2() {}
void fa_touch_msa() {}
void fa_touch_links2() {}
void fa_touch_score1() {}
void fa_touch_score2() {}
