// fa_internal.h -- structures shared by the HIP kernels and the host batch engine.
//
// Device data layout of one batch (everything stays resident in HBM between
// stages; see DESIGN.md "Data layout in HBM"):
//
//   words[]      2-bit packed bases, 16 per u32, base i at bits 2*(i%16);
//                every sequence starts on a 16-byte boundary and is followed by
//                >= 2 zero words so 64-bit window reads never leave the buffer.
//   seq[g]       per sequence: word offset, length, pile id, index inside pile
//                (index 0 = the seed/target, reference falcon.c:593).
//   pile[p]      per pile: first sequence, count, CSR k-mer table offsets,
//                node-arena and output offsets.
//   kidx[]       per pile 65537 u32: CSR bucket bounds of the seed's 8-mers
//   kpos[]       per pile: seed positions grouped by 8-mer, ascending
//   range[g]     k-mer chain window (s1,e1,s2,e2) + sanity-filter verdict
//   aln[g]       alignment summary (dist, ends, columns, accept verdict)
//   script[]     per accepted alignment: one u32 per edit row d:
//                (snake_length << 1) | from_above
//   nodes[]      per pile: MSA nodes, 5 per (t_pos,delta) level
//   out_seq/out_eqv  per pile: consensus written right-aligned in 2*T slots
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fa_types.h"

struct FaRowRec {   // one per edit row d of an alignment in flight (16 B)
    u32 off;        // first cell of the row in the slot's cell arena
    int min_k;
    u32 dlo, dhi;   // from_above bit of cells 0..63
};

struct FaRowExt {   // rows wider than 64 cells only: from_above bits of cells 64..191
    u64 d1, d2;
};

struct FaNode {     // 8 B
    int score_h;    // score in half units; -2 = the reference's -1 floor
    int link;       // (best_prev_node + 1) << 1 | upper ; prev -1 = alignment start
};

struct FaPileOut {
    int len;        // consensus length
    int start;      // first character inside the pile's output slots
    int n_aligned;
    int err;
    long long g_best_h;
};

struct FaTagAln {   // one per accepted alignment, grouped by pile in read order (host-built)
    u64 desc_off;   // offset of its tag words (one u32 per covered target position)
    u32 ins_off;    // offset of its inserted-base bytes
    int s2;         // first covered target position
    int g;          // sequence index
    int pile;
    int pad;
};

struct FaTInfo {    // per target position of a pile (k_tscan)
    u32 lvl_start;  // first level slot of the position (node id = slot * 5 + base)
    u32 link_start; // first link word of the position
    u16 cov;        // alignments covering it (falcon.c:357-360)
    u16 nlev;       // 1 + deepest insertion level
};

struct FaScoreOut { // per pile
    int g_node, g_ck, g_h;   // global best node, link index of its best link, its score
    int n_levels;
    int n_links;
    int err;
    int wide;       // 1: scores may outgrow the fast paths' keys (k_score1's generic path takes the pile)
    int redo;       // 1: k_score2 handed the pile on to k_score1
};

// ---- launcher prototypes (each .hip file owns its kernels) ----
struct FaBatchDev {
    // inputs
    const uint8_t *ascii;  // staged ASCII, each sequence 16-byte aligned
    const u64 *ascii_off;  // [n_seq]
    u32 *words;
    FaSeq *seq;
    FaPile *pile;
    int n_seq;
    int n_pile;
    u64 n_words;
    // stage buffers
    u32 *kidx;
    u32 *kpos;
    const int *order;      // sequence indices, longest first (k_align work queue)
    const int *chain_order;  // k_chain work list: pile-major, 8 interleaved streams (-1 = padding)
    int n_chain;
    u64 *probe;            // k_chain: one record per probe (k_chain.hip, CH_REC)
    const u64 *probe_off;  // [n_seq]
    FaRange *range;
    FaAln *aln;
    u32 *script;
    const u64 *script_off; // [n_seq]
    FaNode *nodes;
    char *out_seq;
    int *out_eqv;
    FaPileOut *pile_out;
};

struct FaAlignArena {
    u32 *cells;        // n_slot * cells_per_slot
    FaRowRec *rows;    // n_slot * rows_per_slot
    FaRowExt *rowx;    // n_slot * rows_per_slot
    u64 cells_per_slot;
    u64 rows_per_slot;
    int n_slot;
    int *counter;      // work-queue head
    u64 *prof;         // 8 debug counters (FA_ALIGN_PROF builds)
};

// k_align2.hip: two alignments per wavefront over an iteration tape (one-byte cells)
struct FaAlign2Arena {
    u32 *mem;          // n_slot * slot_words
    u64 slot_words;    // fa_align2_slot_words(ring)
    u32 ring;          // tape iterations per slot (a power of two)
    int n_slot;
    int *counter;      // work-queue head
    unsigned long long *stats;  // 8 counters (optional): pair / single iterations, placements, parkings, hand-backs, -, wide episodes, alignments
};
u64 fa_align2_slot_words(u32 ring);
u32 fa_align2_ring_for(int max_rows);
size_t fa_align2_lds_bytes();
int fa_align2_blocks_per_cu();
// alignments whose band tolerance is >= 64 and whose packed words fit 2^32 bases; what it
// cannot hold on its tape comes back with FaAln.err = 2 (repeat with fa_launch_align_list)
// (`order` / `n_work`: the stretch of the work list to align; `word_base`: the first packed word of its piles)
void fa_launch_align2(const FaBatchDev &b, const FaAlign2Arena &a, double max_diff, int band,
                      const int *order, int n_work, u32 word_base, hipStream_t s);

// first_bad: device int, preset to INT_MAX; receives the lowest sequence index holding a byte
// other than upper-case A, C, G, T.  bad_pile (optional, n_pile ints, zeroed): 1 for every
// pile with such a sequence
void fa_launch_pack(const FaBatchDev &b, int *first_bad, int *bad_pile, hipStream_t s);
void fa_launch_index(const FaBatchDev &b, int max_seed_len, hipStream_t s);
void fa_launch_chain(const FaBatchDev &b, int max_bins, hipStream_t s);
// general banded alignment for band tolerances beyond FA_ALIGN_MAXCH chunks (k_align_wide.hip)
#define FA_WIDE_BAND_MAX 2000
void fa_launch_align_wide(const FaBatchDev &b, const FaAlignArena &a, double max_diff, int band,
                          hipStream_t s);
// --trim windows (k_trimwin.hip): count_only = 1 stores every read's hit count in
// range[g].n_hit; the full pass needs scratch (n_slot x 4 x cap words) only when a read
// has more hits than fit LDS
void fa_launch_trimwin(const FaBatchDev &b, int n_slot, int *counter, u32 *scratch, u64 cap,
                       int mask_th, int count_only, hipStream_t s);
void fa_launch_align(const FaBatchDev &b, const FaAlignArena &a, int max_q_len, int max_t_len,
                     double max_diff, hipStream_t s);
void fa_launch_align_band(const FaBatchDev &b, const FaAlignArena &a, int max_q_len,
                          int max_t_len, double max_diff, int band, hipStream_t s);
void fa_launch_align_list(const FaBatchDev &b, const FaAlignArena &a, int max_q_len, int max_t_len,
                          double max_diff, int band, const int *order, int n_work, hipStream_t s);
struct FaMsaDev {
    const FaTagAln *ta;
    const u32 *acc_first;
    int n_acc_total;
    int *tcov;
    u32 *desc;
    uint8_t *insb;
    int *seg_cnt;          // 2 ints per segment of TSEG positions (zeroed before k_tags)
    u32 *seg_base;         // 2 per segment
    const u32 *seg_first;  // [n_pile]
    unsigned long long *bound;  // [n_pile]
    const u64 *t_off;
    FaTInfo *tinfo;
    u32 *links;
    const u64 *link_off;
    const u64 *link_cap;
    uint16_t *lvl_nlink16;
    int *score_ovf;
    FaScoreOut *score_out;
    const int *seg_pile;
    const int *seg_t0;
    int n_seg;
    int first_links_back;  // unitig mode: a read's first column links back to (s2 - 1, 0, '-')
    int force_generic;     // k_score1: every level through the generic path (tests)
    int score_mode;        // 0: k_score2, k_score1 for what it hands on; 1: k_score1 for every pile
    int links_mode;        // 0: k_links2, k_links for what it hands on; 1: k_links for every segment
    int *wide_count;
    int *wide_list;
};
// the consensus stage in two halves (k_msa.hip): graph building (k_tags, k_tscan, k_links) and
// the per-pile sequential walk (k_score, k_backtrace); events (optional) are recorded after
// k_tags + k_tscan, k_links, k_score, k_backtrace
void fa_launch_msa_front(const FaBatchDev &b, const FaMsaDev &m, unsigned min_cov, hipStream_t s,
                         hipEvent_t ev_tags, hipEvent_t ev_links);
void fa_launch_msa_back(const FaBatchDev &b, const FaMsaDev &m, unsigned min_cov, hipStream_t s,
                        hipEvent_t ev_score, hipEvent_t ev_backtrace);
// fa_warm: one kernel of every file of the consensus path looked at, so that its code object is loaded
void fa_touch_index();
void fa_touch_chain();
void fa_touch_align2();
void fa_touch_msa();
void fa_touch_links2();
void fa_touch_score1();
void fa_touch_score2();
size_t fa_align_lds_bytes(int max_q_len, int max_t_len);
int fa_align_blocks_per_cu(size_t lds_bytes);
