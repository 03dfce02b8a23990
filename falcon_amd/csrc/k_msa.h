// k_msa.h -- what the consensus-stage kernels share: the argument block, the tag word, the wave scans.
// (kernels: k_msa.hip -- tags, position scan, links, back-trace; k_score2.hip / k_score1.hip -- the score recurrence)
#pragma once
#include "fa_device.h"
#include <type_traits>

#define TSEG 128       // target positions per k_links wavefront
#define MAXACT 1024    // alignments overlapping one segment that k_links (the kernel behind k_links2) takes
#define INL 11         // inserted bases stored inline in a tag
#define BT_WIN 128     // levels per back-trace window (a multiple of 64)

// tag word of one covered target position (one u32, written exactly once):
//   bit 31      the alignment deletes the target base
//   bits 30..23 length of the insertion run that follows it (0..254)
//   bits 22..0  the run itself: up to INL bases inline (2 bits each, first base lowest),
//               longer runs the index of their first base in the alignment's byte list
#define TAG_DEL 0x80000000u
#define TAG_NINS_SHIFT 23
#define TAG_PAY_MASK 0x7fffffu
#define TG_WIN 1024    // target positions per k_tags LDS window
#define TG_BLK 16      // positions per k_links tag block
#define TCOV_LEAD 0x40000000  // tcov flag: the alignment opens with an insertion run (see k_tags)
// link word of one (node, previous node) pair of a level (update_col, falcon.c:232-263):
//   bits 15..0  alignments on the link      bits 18..16  the node's base (A0 C1 G2 T3 -4)
//   bits 29..19 the previous node's index delta * 5 + base at the previous position (delta 0)
//               or at this one (delta >= 1)   bit 30  no previous node (the alignment starts, :434)
#define LW_CNT_MASK 0xffffu
#define LW_NB_SHIFT 16
#define LW_PIDX_SHIFT 19
#define LW_START_BIT 30
// scores are bounded by the sum over levels of the coverage; a pile whose bound would not fit
// k_score2's 32-bit biased scores takes k_score1
#define SC_FAST_SCORE_MAX 2000000000ll

struct MsaArgs {
    const u32 *words;
    const FaSeq *seq;
    const FaPile *pile;
    const FaRange *range;
    const FaAln *aln;
    const u32 *script;
    const u64 *script_off;
    const FaTagAln *ta;        // accepted alignments, grouped by pile, read order
    const u32 *acc_first;      // [n_pile + 1]
    int n_acc_total;
    int n_pile;
    int *tcov;                 // [n_acc_total] covered target positions (k_tags)
    u32 *desc;                 // one tag word per covered target position
    uint8_t *insb;             // inserted bases of runs longer than INL (and all others)
    int *seg_cnt;              // per segment of TSEG positions: columns, inserted bases (k_tags)
    u32 *seg_base;             // per segment: its first link slot, its first level slot (k_sscan)
    const u32 *seg_first;      // [n_pile] a pile's first segment
    unsigned long long *bound; // [n_pile] sum over the positions of coverage x levels: no score exceeds it (k_links2)
    const u64 *t_off;          // [n_pile] position offset of the pile's per-t arrays
    FaTInfo *tinfo;            // per target position
    u32 *links;                // link words
    const u64 *link_off;       // [n_pile]
    const u64 *link_cap;       // [n_pile]
    u16 *lvl_nlink16;          // per level slot: number of links
    FaNode *nodes;
    int *score_ovf;            // per pile 2 x 256 x 5 ints: scores of levels >= SC_LCAP
    FaScoreOut *score_out;     // per pile
    char *out_seq;
    int *out_eqv;
    FaPileOut *pile_out;
    const int *seg_pile;       // k_links work list
    const int *seg_t0;
    int n_seg;
    int *wide_count;           // to-do lists (6 lists of 1 + n_seg ints: [count, segments...]): k_links2 ->
    int *wide_list;            // k_links2_big -> k_links<1> -> <2> -> <4> -> <8> -> <16>
    unsigned min_cov;
    int first_links_back;      // unitig mode (falcon.c:668-773): see k_links
    int force_generic;         // k_score1: every level through the generic path (tests)
    int only_redo;             // k_score1: only the piles k_score2 handed on (FaScoreOut.redo)
    int links_old;             // k_links2 hands every segment to k_links (A/B runs, tests of the fallback)
};

// the links of every segment (k_links2.hip); what it cannot hold goes to k_links through A.wide_count
void fa_launch_links2(const MsaArgs &A, hipStream_t s);

// the score recurrence, one wavefront per pile: k_score2 (k_score2.hip) takes every pile and
// hands what it does not hold -- FaScoreOut.redo -- to the general kernel (k_score1.hip)
void fa_launch_score2(const MsaArgs &A, hipStream_t s);
void fa_launch_score1(const MsaArgs &A, hipStream_t s);

// ---------------------------------------------------------------------------
// wave scans
// ---------------------------------------------------------------------------
// DPP inclusive scans over the 64 lanes: Kogge-Stone inside each row of 16
// (row_shr 1,2,4,8), then row_bcast15 / row_bcast31 carry the row totals across
// (gfx9 DPP controls).  ~12 VALU, no LDS crossbar traffic.
__device__ __forceinline__ int wave_incl_sum(int v, int lane) {
    (void)lane;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
    return v;
}
__device__ __forceinline__ int wave_incl_max(int v, int lane) {
    (void)lane;
    const int lowest = -0x7fffffff - 1;
    v = max(v, __builtin_amdgcn_update_dpp(lowest, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(lowest, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(lowest, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(lowest, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(lowest, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(lowest, v, 0x143, 0xc, 0xf, false));
    return v;
}
// tag word accessors
__device__ __forceinline__ int tag_nins(u32 w) { return (int)((w >> TAG_NINS_SHIFT) & 0xffu); }
// base `delta` (1-based) of the insertion run of a tag: runs of up to INL bases are
// inline in the low word, longer runs live entirely in the byte array
__device__ __forceinline__ int tag_ins_base(const MsaArgs &A, u32 ins_off, u32 w, int delta) {
    if (tag_nins(w) <= INL) return (int)((w >> (2 * (delta - 1))) & 3u);
    return (int)A.insb[ins_off + (w & TAG_PAY_MASK) + (u32)(delta - 1)];
}
