// k_align2.hip -- the banded O(ND) alignment stage, two alignments per wavefront over a
// shared iteration tape with one-byte DP cells (DW_banded.c:115-330).  The kernel's body is
// k_align2_core.h (read its header first); this file is the launch: one persistent
// wavefront per arena slot, work taken from the batch's queue (longest reads first) with
// one atomic per alignment.
//
// Integer / latency / issue bound like the rest of the path; no MFMA.
#include <algorithm>
#include <cstdlib>
#include "fa_wave.h"
#include "k_align2_core.h"

void fa_launch_align2_shadow(const A2Args &A, int grid, size_t lds, hipStream_t s);

__global__ __launch_bounds__(64, 8) void k_align2(A2Args A) {
    a2_wave(A, (int)blockIdx.x);
}

// u32 words of one slot: cells (ring x 64 bytes), records (ring x 16 bytes), escape list
u64 fa_align2_slot_words(u32 ring) {
    u64 w = (u64)ring * 16u + (u64)ring * 4u + (u64)A2_ESC_CAP * 2u;
    // slot stride = 4 KB x m + 1.75 KB: the slots' first pages (all waves start writing at
    // their slot's base) spread over the memory channels instead of piling on a few
    return ((w + 1023u) & ~(u64)1023u) + 448u;
}

size_t fa_align2_lds_bytes() { return A2_LDS_WORDS * sizeof(u32); }

int fa_align2_blocks_per_cu() {
    int nb = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_align2, 64, fa_align2_lds_bytes());
    if (e != hipSuccess || nb <= 0) nb = 8;
    return nb;
}

// The tape ring for a batch whose longest alignment may take `max_rows` rows (the bound of
// DW_banded.c:149, 0.3 (q + t); alignments that end up accepted walk ~0.2 (q + t)): the
// smallest power of two that holds 0.6 x that bound.  An alignment is admitted when its
// own bound fits the ring outright, or 0.6 x it does with room to wait parked; the few that
// then do outgrow the tape are handed back.
u32 fa_align2_ring_for(int max_rows) {
    u32 r = 1024;
    while (r < (u32)((u64)max_rows * 6 / 10) + 512u && r < (1u << 20)) r <<= 1;
    return r;
}

void fa_launch_align2(const FaBatchDev &b, const FaAlign2Arena &a, double max_diff, int band,
                      const int *order, int n_work, u32 word_base, hipStream_t s) {
    if (n_work == 0) return;
    A2Args A;
    A.words = b.words + word_base; A.word_base = word_base; A.seq = b.seq; A.pile = b.pile; A.range = b.range;
    A.order = order; A.n_work = n_work; A.counter = a.counter;
    A.cells = a.mem; A.recs = nullptr; A.esc = nullptr;
    A.slot_words = a.slot_words;
    A.ring = a.ring;
    A.script = b.script; A.script_off = b.script_off; A.aln = b.aln;
    A.band = band; A.max_diff = max_diff;
    A.stats = a.stats;
    {   // (FALCON_AMD_A2_DEBUG: timing experiments, see A2Args::debug -- results are not valid then)
        const char *e = getenv("FALCON_AMD_A2_DEBUG");
        A.debug = e ? atoi(e) : 0;
        e = getenv("FALCON_AMD_ESC_CAP");           // (tests: force the escape-list exit)
        A.esc_cap = e ? std::max(1, std::min(atoi(e), (int)A2_ESC_CAP)) : (int)A2_ESC_CAP;
        e = getenv("FALCON_AMD_WIDE_PATIENCE");     // (tests: force the wide-rows exit)
        A.wide_patience = e ? std::max(1, atoi(e)) : (int)A2_WIDE_PATIENCE;
    }
    (void)hipMemsetAsync(a.counter, 0, sizeof(int), s);
    // two alignments per wavefront: half as many wavefronts have work
    int grid = a.n_slot;
    if (grid > (n_work + 1) / 2) grid = (n_work + 1) / 2;
    if (getenv("FALCON_AMD_A2_SHADOW")) {  // (tests: k_align2_shadow.hip)
        fa_launch_align2_shadow(A, grid, fa_align2_lds_bytes(), s);
        return;
    }
    hipLaunchKernelGGL(k_align2, dim3(grid), dim3(64), fa_align2_lds_bytes(), s, A);
}

// (fa_warm: the code object of this file is loaded when one of its kernels is first looked at)
void fa_touch_align2() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(k_align2));
}
