// k_msa.hip -- alignment tags -> MSA graph -> best path -> consensus.
//
// Restates get_align_tags() (src/c/falcon.c:106-162) and
// get_cns_from_align_tags() (src/c/falcon.c:308-558).  The reference builds the
// graph by inserting tags one at a time into a pointer tree and then scores it;
// both are serial per pile.  Here the graph construction -- the bulk of the
// work, and independent of the scores -- is data-parallel over alignments and
// over target positions; only the score recurrence itself stays sequential,
// and it is a few operations per link:
//
//   k_tags      one wavefront per accepted alignment.  Prefix scans over the
//               edit script turn it into position-major tags: one 4-byte word
//               per covered target position {deleted?, insertion-run length,
//               up to 11 inserted bases inline} (longer runs spill to a byte
//               array), each written exactly once, staged through an LDS window
//               and stored in whole lines.  Also counts, per SEGMENT of 128
//               positions, the columns and the inserted bases (the pools' sizes).
//   k_sscan     one wavefront per pile: prefix sums over the pile's segments ->
//               every segment's first link slot and first level slot.
//   k_links2    (k_links2.hip) one wavefront per segment, lanes = positions: the
//               links of every level, coverage, levels and level slots of every
//               position.  k_links<1..16> below is the kernel behind it:
//   k_links     one wavefront per (pile, segment of 128 target positions):
//               lanes = alignments overlapping the segment (compacted, in read
//               order).  For every (t, delta) level the lanes holding the same
//               (base, previous node) are one link of the reference
//               (update_col, falcon.c:232-263): grouped with ballots, group
//               size = link count, groups visited in lowest-lane order = the
//               reference's first-insertion order (Q5).  Emits one u32 per link.
//   k_score2    (k_score2.hip; k_score1.hip behind it) the score recurrence
//               (falcon.c:405-475) over the link words, -1 floor, strict '>', first
//               maximum; writes the 8-byte node records and the global best.
//   k_backtrace one wavefront per pile: the best path by pointer doubling inside a
//               128-level LDS window of node records (64 nodes per round), then the 64
//               nodes become characters and eqv values at once, written right-aligned
//               (falcon.c:494-528, no reversal pass).
//
// Integer only; per-column reduction over aligned bases; no MFMA.
#include "k_msa.h"

// ---------------------------------------------------------------------------
// k_tags
// ---------------------------------------------------------------------------
// Insertion depth (delta) of every row of a 64-row chunk.  Row d is an insertion
// column iff its edit bit is 0; it continues the run of the previous column iff
// row d-1 is an insertion with an empty snake.  carry_open = depth of the last
// row of the previous chunk (0 if that row is not an insertion).
__device__ __forceinline__ int chunk_delta(u32 ep, int d, int d0, bool is_ins, int lane,
                                           int carry_open, bool &cont, int &start_row) {
    cont = is_ins && d >= 2 && (ep & 1u) == 0u && (ep >> 1) == 0u;
    int start = (is_ins && !cont) ? d : -1;
    start = wave_incl_max(start, lane);
    start_row = start;
    if (!is_ins) return 0;
    return (start >= d0) ? (d - start + 1) : (carry_open + (d - d0) + 1);
}

// The script words of a chunk's rows: each lane its own (e), its predecessor's (ep) and
// its successor's (en) -- one load per chunk, a chunk ahead; neighbours come from DPP
// wave shifts, the chunk edges from the previous / next chunk's registers.
struct ScrChunk { u32 e, ep, en; };
__device__ __forceinline__ ScrChunk scr_chunk(u32 e_cur, u32 e_last_prev, u32 e_next_first, int lane) {
    ScrChunk c;
    c.e = e_cur;
    c.ep = (u32)__builtin_amdgcn_update_dpp((int)e_last_prev, (int)e_cur, 0x138, 0xf, 0xf, false);  // wave_shr:1
    c.en = (u32)__builtin_amdgcn_update_dpp((int)e_next_first, (int)e_cur, 0x130, 0xf, 0xf, false); // wave_shl:1
    (void)lane;
    return c;
}

__global__ __launch_bounds__(64) void k_tags(MsaArgs A) {
    const int lane = fa_lane();
    const int k = blockIdx.x;
    if (k >= A.n_acc_total) return;
    const FaTagAln ta = A.ta[k];
    const int g = ta.g;
    const FaAln al = A.aln[g];
    const FaRange rg = A.range[g];
    const u32 *scr = A.script + A.script_off[g];
    const u32 *rw = A.words + A.seq[g].woff;
    u32 *desc = A.desc + ta.desc_off;
    uint8_t *insb = A.insb + ta.ins_off;
    // what the alignment adds to the segments of TSEG positions it touches: columns (delta 0
    // tags) and inserted bases (tags of delta >= 1, counted at the position they hang off) --
    // k_sscan sizes every segment's link and level slots from the two sums
    int *segc = A.seg_cnt + 2 * (size_t)A.seg_first[ta.pile];
    __shared__ u32 segl[16];  // inserted bases of a chunk, by segment
    const int dist = al.dist, te = al.t_e;
    // Tag words leave in whole lines: the target positions a chunk of 64 script rows
    // consumes are built in this LDS window (zeros, then the chunk's tags) and stored
    // once, coalesced.  (Zero-filling the alignment's words up front and scattering the
    // tags over them afterwards wrote every line twice, the second time as 4-byte
    // fragments: 29 GB of HBM writes per launch instead of 10.)
    __shared__ u32 win[TG_WIN];
    // An alignment that opens with an insertion run (no target base consumed yet) hangs
    // that run off the position BEFORE its first one (get_align_tags: j = start - 1,
    // falcon.c:119-140), whose tag word lives in the slot before the alignment's own
    // (desc[-1], zero = no such run); with nothing before it (start 0) the reference stops
    // tagging at once (:138,:152): the alignment contributes nothing.  Cannot happen on the
    // falcon_sense path (windows open on a k-mer match); unitig windows open anywhere.
    const bool lead = dist >= 1 && (scr[0] >> 1) == 0u && (scr[1] & 1u) == 0u;
    if (lane == 0) desc[-1] = 0u;
    if (lead && rg.s2 == 0) {
        if (lane == 0) A.tcov[k] = 0;
        return;
    }

    // Tagging stops at the first column whose insertion depth reaches 255 (falcon.c:138-152):
    // dcut = that row (dist + 1 if none).  Rounds 1-3 walked the whole script once to find it
    // before the pass that writes the tags.  Such a run is 254 consecutive all-zero script words
    // (insertion, empty snake), which hold at least two whole chunks of 64 rows -- and the
    // first of them starts 191 rows or more before dcut, so nothing written up to there depends
    // on where the tagging ends.  The main pass therefore only looks for an all-zero chunk, and
    // the exact walk happens when it meets one (practically never).
    auto exact_dcut = [&]() {
        int carry_open = 0, first_bad = 0x7fffffff;
        u32 e_nx = (lane <= dist) ? scr[lane] : 1u, e_last = 1u;
        for (int d0 = 0; d0 <= dist; d0 += 64) {
            const int d = d0 + lane;
            const bool have = d <= dist;
            const u32 e_cur = e_nx;
            e_nx = (d + 64 <= dist) ? scr[d + 64] : 1u;
            const ScrChunk sc = scr_chunk(e_cur, e_last, 1u, lane);
            e_last = (u32)__builtin_amdgcn_readlane((int)e_cur, 63);
            const u32 e = sc.e;
            const bool is_ins = have && d >= 1 && (e & 1u) == 0u;
            bool cont;
            int sr;
            const int delta = chunk_delta(d >= 2 ? sc.ep : 1u, d, d0, is_ins, lane, carry_open, cont, sr);
            if (delta >= 255) first_bad = min(first_bad, d);
            carry_open = __builtin_amdgcn_readlane(delta, 63);
        }
        first_bad = fa_wave_min(first_bad);
        return first_bad <= dist ? first_bad : dist + 1;
    };
    int dcut = dist + 1;
    bool dcut_known = false;
    fa_wave_sync();

    // main pass over rows 0 .. dlast.  What a chunk needs from global memory is requested at the
    // END of the chunk before it, behind that chunk's stores (the counter a wait looks at
    // counts loads and stores alike, in order): the script words of the chunk after the next
    // one (the next chunk's are in registers already: a chunk's last row looks at its
    // successor), and 64 packed words of the read from the next chunk's first query base on,
    // out of which the inserted bases are taken with a lane shuffle.
    int dlast = dist;
    int carry_t = 0, carry_q = rg.s1, carry_i = 0, carry_open = 0, carry_run_i = 0;
    int carry_start_row = 0;
    u32 carry_inl = 0, carry_es = 0;
    const int rw_last = (A.seq[g].len + 15) / 16;  // (two zero words follow every sequence)
    // (every request unconditional, the index clamped: rows beyond dlast are masked where they are
    // used -- a load under a branch makes the compiler wait for it on the spot)
    // The registers take turns instead of being copied (a copy of a register a load is still under way
    // for is a wait for that load -- and for every store issued before it): a chunk's script words
    // are in `e_cur`, the next chunk's in `e_nx`; when the chunk is done, e_cur's register takes the
    // words of the chunk after the next, and the two swap names.  Likewise the read words.
    u32 e_a = scr[min(lane, dist)], e_b = scr[min(lane + 64, dist)], e_last = 1u;
    int rw_base = rg.s1 >> 4;
    u32 rw_a = rw[min(rw_base + lane, rw_last)], rw_b = 0;
    int d0 = 0;
    auto chunk = [&](u32 &e_cur, u32 &e_nx, u32 &rwn, u32 &rwn_nx) {
        const int d = d0 + lane;
        if (!dcut_known && d0 + 63 <= dlast && fa_ballot(e_cur != 0u) == 0) {  // (wave-uniform)
            dcut = exact_dcut();
            dlast = min(dist, dcut - 1);
            dcut_known = true;
        }
        const bool have = d <= dlast;
        // rows beyond dlast read as "deletion, no snake"
        ScrChunk sc;
        sc.e = have ? e_cur : 1u;
        sc.ep = (u32)__builtin_amdgcn_update_dpp((int)e_last, (int)sc.e, 0x138, 0xf, 0xf, false);  // wave_shr:1
        const u32 e_last_prev = e_last;
        e_last = (u32)__builtin_amdgcn_readlane((int)sc.e, 63);
        const u32 e = sc.e;
        const int m = have ? (int)(e >> 1) : 0;
        const bool is_edit = have && d >= 1;
        const bool is_del = is_edit && (e & 1u) != 0u;
        const bool is_ins = is_edit && (e & 1u) == 0u;
        const int tc = (is_del ? 1 : 0) + m, qc = (is_ins ? 1 : 0) + m;
        const int ts = wave_incl_sum(tc, lane), qs = wave_incl_sum(qc, lane);
        const int is = wave_incl_sum(is_ins ? 1 : 0, lane);
        const int tpos = carry_t + ts - tc;     // target index of this row's edit
        const int qpos = carry_q + qs - qc;     // query index of this row's inserted base
        const int iidx = carry_i + is - (is_ins ? 1 : 0);
        // the read words the NEXT chunk takes its inserted bases from: its first query base is known
        const int q_next = carry_q + __builtin_amdgcn_readlane(qs, 63);
        rwn_nx = rw[min((q_next >> 4) + lane, rw_last)];
        bool cont;
        int start_row = 0;
        const int delta = chunk_delta(d >= 2 ? sc.ep : 1u, d, d0, is_ins, lane, carry_open, cont, start_row);
        const bool from_prev_chunk = is_ins && (delta - 1 > d - d0);
        if (from_prev_chunk) start_row = carry_start_row;
        // index (in the alignment's insertion list) of the first base of my run
        const int run_i = is_ins ? (from_prev_chunk ? carry_run_i : iidx - (delta - 1)) : 0;
        // the column before my run (decides whether the run hangs off a deleted base):
        // a lane of this chunk, the last row of the previous one, or -- for a run that
        // began in an earlier chunk -- what that chunk found
        u32 es = (u32)__shfl((int)e, start_row - 1 - d0);
        if (start_row - 1 == d0 - 1) es = e_last_prev;
        if (from_prev_chunk) es = carry_es;
        if (start_row < 2) es = 0u;
        // inline bases of my run so far: OR of b << 2(delta-1) over the run's rows
        // (the inserted base: out of the 64 read words requested a chunk ago; a chunk whose
        // snakes carry it more than ~1000 query bases goes to memory itself)
        u32 b = 0, inl = 0;
        {
            const int wi = (qpos >> 4) - rw_base;
            if (fa_ballot(is_ins && wi >= 64) != 0) {
                if (is_ins) b = fa_settled(fa_base_at(rw, qpos));  // (waited for inside this rare branch)
            } else {
                const u32 word = (u32)__shfl((int)rwn, is_ins ? wi : 0);
                b = (word >> ((qpos & 15) * 2)) & 3u;
            }
        }
        if (is_ins) {
            insb[iidx] = (uint8_t)b;
            if (delta <= INL) inl = b << (2 * (delta - 1));
        }
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {  // 16 >= INL
            const u32 o = (u32)__shfl_up((int)inl, off);
            if (is_ins && delta - 1 >= off && lane >= off) inl |= o;
        }
        if (from_prev_chunk) inl |= carry_inl;
        // the run ends here unless the next row continues it (the last lane looks at the next
        // chunk's first row: the registers requested at the end of the chunk before this one --
        // their first use, this far down, is what the wait for them lands in front of)
        sc.en = (u32)__builtin_amdgcn_update_dpp(
            (int)(d0 + 64 <= dlast ? (u32)__builtin_amdgcn_readfirstlane((int)e_nx) : 1u), (int)sc.e, 0x130, 0xf, 0xf,
            false);  // wave_shl:1
        const u32 en = (d + 1 <= dlast) ? sc.en : 1u;
        const bool ends = is_ins && (m > 0 || (en & 1u) != 0u);
        // the chunk's window of target positions [w0, w0 + span)
        const int w0 = carry_t, span = __builtin_amdgcn_readlane(ts, 63);
        const bool in_lds = span <= TG_WIN;  // (a very long match run: straight to HBM)
        // (one wavefront: its LDS operations execute in order, so the zeros, the tags and the
        // reads of the store-out need no barrier between them -- a __syncthreads() here also
        // waited for the previous chunk's stores to be acknowledged, three times per chunk)
        fa_wave_sync();
        if (in_lds) { for (int i = lane; i < span; i += 64) win[i] = 0u; }
        else { for (int i = lane; i < span; i += 64) desc[w0 + i] = 0u; }
        if (lane < 16) segl[lane] = 0u;
        const int sg0 = max(0, rg.s2 + w0 - 1) / TSEG;  // the first segment a run of this chunk can hang off
        fa_wave_sync();
        // Every tag word has exactly one writer (no atomics): a deletion writes its
        // flag unless an insertion run hangs off the deleted base, in which case the
        // run's last row writes the whole word.
        if (is_del && !(m == 0 && d + 1 <= dlast && (en & 1u) == 0u)) {
            if (in_lds) win[tpos - w0] = TAG_DEL; else desc[tpos] = TAG_DEL;
        }
        if (ends) {
            const int us = tpos - 1;  // the target position the run hangs off (>= w0 - 1)
            const bool on_del = start_row >= 2 && (es & 1u) != 0u && (es >> 1) == 0u;
            const u32 word = (delta <= INL ? inl : ((u32)run_i & TAG_PAY_MASK)) |
                             ((u32)delta << TAG_NINS_SHIFT) | (on_del ? TAG_DEL : 0u);
            // (us == w0 - 1 lies in the window the previous chunk already stored: the
            // later store of the same wavefront to the same word wins)
            if (in_lds && us >= w0) win[us - w0] = word; else desc[us] = word;
            // (rounds 1-3 kept the deepest run and the sum of the runs PER POSITION, two global
            // atomics per run: 0.38 G per launch, two thirds of the kernel's HBM traffic.  The
            // levels of a position are found by k_links2 now, and the slots only need sums per
            // segment: the chunk's runs meet in LDS, a handful of atomics per chunk leave)
            const int sg = (rg.s2 + us) / TSEG;
            if (in_lds) atomicAdd(&segl[sg - sg0], (u32)delta);
            else atomicAdd(&segc[2 * sg + 1], delta);
        }
        fa_wave_sync();
        if (in_lds) {  // (four stores per trip: unrolled further, their addresses alone take 32 registers)
            u32 *dp = desc + w0 + lane;
#pragma unroll 4
            for (int i = lane; i < span; i += 64, dp += 64) *dp = win[i];
        }
        if (lane < 16 && segl[lane] != 0u) atomicAdd(&segc[2 * (sg0 + lane) + 1], (int)segl[lane]);
        carry_t += __builtin_amdgcn_readlane(ts, 63);
        carry_q += __builtin_amdgcn_readlane(qs, 63);
        // the request for the chunk after the next one, into the register this chunk is done with
        e_cur = scr[min(d + 128, dist)];
        rw_base = q_next >> 4;
        carry_i += __builtin_amdgcn_readlane(is, 63);
        carry_open = __builtin_amdgcn_readlane(delta, 63);
        carry_run_i = __builtin_amdgcn_readlane(run_i, 63);
        carry_start_row = __builtin_amdgcn_readlane(start_row, 63);
        carry_inl = (u32)__builtin_amdgcn_readlane((int)inl, 63);
        carry_es = (u32)__builtin_amdgcn_readlane((int)es, 63);
        d0 += 64;
    };
    while (d0 <= dlast) {
        chunk(e_a, e_b, rw_a, rw_b);
        if (d0 > dlast) break;
        chunk(e_b, e_a, rw_b, rw_a);
    }
    // rows >= dcut are dropped: the alignment covers only what the kept rows consumed
    const int t_cov = (dcut <= dist) ? carry_t : te;
    if (lane == 0) A.tcov[k] = t_cov | (lead ? TCOV_LEAD : 0);
    // its columns, per segment: positions [s2, s2 + t_cov)
    if (t_cov > 0) {
        const int sg_a = rg.s2 / TSEG, sg_b = (rg.s2 + t_cov - 1) / TSEG;
        for (int sg = sg_a + lane; sg <= sg_b; sg += 64) {
            const int lo = max(rg.s2, sg * TSEG), hi = min(rg.s2 + t_cov, (sg + 1) * TSEG);
            atomicAdd(&segc[2 * sg], hi - lo);
        }
    }
}

// ---------------------------------------------------------------------------
// k_sscan: per pile prefix sums over its segments of TSEG positions: where each segment's
// links and levels begin.  A segment has room for every tag k_tags counted into it -- its
// links are at most its tags; its levels at most its positions plus the inserted bases hanging
// off them -- so the segments are independent of one another (k_links2 numbers the levels and
// writes the links of a segment from these two starts; the gaps collect at the segments' ends).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_sscan(MsaArgs A) {
    const int lane = fa_lane();
    const int p = blockIdx.x;
    if (p >= A.n_pile) return;
    const FaPile pm = A.pile[p];
    const int T = pm.seed_len;
    const u32 sf = A.seg_first[p];
    const int n_sg = (T + TSEG - 1) / TSEG;
    u32 c_lvl = 0, c_link = 0;
    for (int s0 = 0; s0 < n_sg; s0 += 64) {
        const int sg = s0 + lane;
        int links = 0, levels = 0;
        if (sg < n_sg) {
            const int cols = A.seg_cnt[2 * (size_t)(sf + (u32)sg)], ins = A.seg_cnt[2 * (size_t)(sf + (u32)sg) + 1];
            links = cols + ins;
            levels = min(TSEG, T - sg * TSEG) + ins;
        }
        const int ks = wave_incl_sum(links, lane), ls = wave_incl_sum(levels, lane);
        if (sg < n_sg) {
            A.seg_base[2 * (size_t)(sf + (u32)sg)] = c_link + (u32)(ks - links);
            A.seg_base[2 * (size_t)(sf + (u32)sg) + 1] = c_lvl + (u32)(ls - levels);
        }
        c_link += (u32)__shfl(ks, 63);
        c_lvl += (u32)__shfl(ls, 63);
    }
    FaScoreOut so;
    so.g_node = -1; so.g_ck = 0; so.g_h = -2;
    so.n_levels = (int)c_lvl;   // (slots, not levels: k_links2 finds the levels)
    so.n_links = (int)c_link;
    so.err = ((u64)c_lvl * 5 > pm.node_cap || (u64)c_link > A.link_cap[p]) ? 1 : 0;
    so.wide = 0;                // (k_score2 looks at the bound k_links2 sums up)
    so.redo = 0;
    A.score_out[p] = so;  // every lane stores the same record
    A.bound[p] = 0ull;
}
// NCHT = 1: segments overlapped by <= 64 alignments (the normal case below 64x
// coverage); wider segments put themselves on a to-do list that the NCHT = 8
// instance works off in a second launch (no host round trip).
template <int NCHT>
__device__ __forceinline__ void links_segment(const MsaArgs &A, int sidx);

// The segments k_links2 (k_links2.hip: lanes = positions) could not hold, through to-do lists:
// instance NCHT works off the list the one before it filled -- k_links2's second instance fills
// the first of them -- and
// hands what is too wide for itself to the next (1 -> 2 -> 4 -> 8 -> 16 chunks of 64
// alignments).  Fixed grids looping over their list (how long it is only the device knows;
// usually it is empty).
template <int NCHT>
__global__ __launch_bounds__(64) void k_links(MsaArgs A) {
    constexpr int LVL = NCHT == 1 ? 1 : NCHT == 2 ? 2 : NCHT == 4 ? 3 : NCHT == 8 ? 4 : 5;
    const int *in = A.wide_count + LVL * (A.n_seg + 1);
    const int n_in = __builtin_amdgcn_readfirstlane(in[0]);
    for (int i = (int)blockIdx.x; i < n_in; i += (int)gridDim.x) {
        links_segment<NCHT>(A, __builtin_amdgcn_readfirstlane(in[1 + i]));
        __syncthreads();  // (the next segment reuses the LDS)
    }
}

template <int NCHT>
__device__ __forceinline__ void links_segment(const MsaArgs &A, int sidx) {
    // (only what this instance can take is kept: a segment with more is handed on; LDS per
    // wavefront decides how many of these latency-bound wavefronts a CU holds)
    __shared__ int act[NCHT * 64];
    const int lane = fa_lane();
    constexpr int LVL = NCHT == 1 ? 1 : NCHT == 2 ? 2 : NCHT == 4 ? 3 : NCHT == 8 ? 4 : 5;
    const int lstride = A.n_seg + 1;
    const int p = __builtin_amdgcn_readfirstlane(A.seg_pile[sidx]);
    const int t_lo = __builtin_amdgcn_readfirstlane(A.seg_t0[sidx]);
    const FaPile pm = A.pile[p];
    const int T = pm.seed_len;
    const int t_hi = min(T, t_lo + TSEG);
    const u32 *seedw = A.words + A.seq[pm.first].woff;
    const FaTInfo *ti = A.tinfo + A.t_off[p];
    u32 *links = A.links + A.link_off[p];
    u16 *nlk = A.lvl_nlink16 + pm.node_off / 5;
    if (A.score_out[p].err) return;

    // alignments of the pile overlapping [t_lo, t_hi), compacted in read order
    const u32 a0 = A.acc_first[p], a1 = A.acc_first[p + 1];
    int n_act = 0;
    for (u32 i0 = a0; i0 < a1; i0 += 64) {
        const u32 i = i0 + lane;
        bool ov = false;
        if (i < a1) {
            const int tcw = A.tcov[i], ld = (tcw & TCOV_LEAD) ? 1 : 0;
            const int s2 = A.ta[i].s2 - ld, tc = (tcw & ~TCOV_LEAD) + ld;  // a leading run sits at s2 - 1
            ov = tc > ld && s2 < t_hi && s2 + tc > t_lo;
        }
        const u64 m = __ballot(ov);
        const int rank = __popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
        if (ov && n_act + rank < NCHT * 64) act[n_act + rank] = (int)i;
        n_act += __popcll(m);
    }
    __syncthreads();
    if (n_act > MAXACT) {  // (a segment that outgrew k_links2's tables AND lies under more than 1024 alignments)
        if (lane == 0) A.score_out[p].err = 2;
        return;
    }
    const int nch = (n_act + 63) >> 6;
    if (nch > NCHT) {  // too wide for this instance: defer
        if (NCHT < 16 && lane == 0) {
            int *out = A.wide_count + (LVL + 1) * lstride;
            out[1 + atomicAdd(out, 1)] = sidx;
        }
        return;
    }

    // per-chunk lane state.  (pbv, pnv) = base and insertion depth of the last column
    // the lane's alignment contributed at the previous target position: the
    // predecessor of its delta-0 tag at this one (falcon.c:129-160)
    int s2v[NCHT], tcv[NCHT], pbv[NCHT], pnv[NCHT];
    const u32 *dptr[NCHT];
    u32 insoff[NCHT];
    bool leadv[NCHT];  // position s2v of the lane is a leading insertion run: no delta-0 column
#pragma unroll
    for (int c = 0; c < NCHT; c++) {
        s2v[c] = 0x7fffffff; tcv[c] = 0; pbv[c] = 0; pnv[c] = 0; insoff[c] = 0; leadv[c] = false;
        dptr[c] = A.desc;
        const int a = c * 64 + lane;
        if (c < nch && a < n_act) {
            const int i = act[a];
            const FaTagAln ta = A.ta[i];
            const int tcw = A.tcov[i], ld = (tcw & TCOV_LEAD) ? 1 : 0;
            leadv[c] = ld != 0;
            s2v[c] = ta.s2 - ld;
            tcv[c] = (tcw & ~TCOV_LEAD) + ld;
            insoff[c] = ta.ins_off;
            dptr[c] = A.desc + ta.desc_off - ld;
            const int u = t_lo - 1 - s2v[c];  // the position before the segment
            if (u >= 0 && u < tcv[c]) {
                const u32 w = dptr[c][u];
                const int pn = tag_nins(w);
                pnv[c] = pn;
                pbv[c] = pn > 0 ? tag_ins_base(A, insoff[c], w, pn)
                                : ((w & TAG_DEL) ? 4 : (int)fa_base_at(seedw, t_lo - 1));
            }
        }
    }

    // (not the unitig mode, no alignment opening with an insertion run: the branch-free key
    // building below)
    bool plain_seg = !A.first_links_back;
#pragma unroll
    for (int c = 0; c < NCHT; c++) plain_seg = plain_seg && fa_ballot(leadv[c]) == 0ull;

    // The segment's position records and seed bases are fetched once, lanes =
    // positions / words, and handed out with readlane; the tag words of position
    // t + 1 are requested while position t is being grouped.
    FaTInfo xr[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int tl = t_lo + 64 * h + lane;
        xr[h].lvl_start = 0; xr[h].link_start = 0; xr[h].cov = 0; xr[h].nlev = 0;
        if (tl < t_hi) xr[h] = ti[tl];
    }
    const int sw0 = t_lo >> 4;  // TSEG is a multiple of 16
    u32 seedv = (lane <= (t_hi - 1 - t_lo) >> 4) ? seedw[sw0 + lane] : 0u;
    // (loaded once, read with readlane at every position: without the opaque copy the
    // compiler waits vmcnt(0) before each of those reads, i.e. for every store and
    // prefetch in flight)
    seedv = fa_settled(seedv);
#pragma unroll
    for (int h = 0; h < 2; h++) {
        xr[h].lvl_start = fa_settled(xr[h].lvl_start);
        xr[h].link_start = fa_settled(xr[h].link_start);
        const u32 cn = fa_settled((u32)xr[h].cov | ((u32)xr[h].nlev << 16));
        xr[h].cov = (u16)(cn & 0xffffu);
        xr[h].nlev = (u16)(cn >> 16);
    }
    // Up to four chunks: the tag words are fetched 16 positions at a time, four 16-byte loads
    // per lane and chunk issued a block ahead, and handed to the positions through LDS
    // ([chunk][position][lane]).  One word per lane and position straight from HBM (what the
    // two widest instances still do) means a wait per position -- for the word AND, the
    // counter being shared, for the link stores of the position before (measured on the
    // one-chunk instance: 45 % of the wave cycles in s_waitcnt, 26 % L1 hit rate).
    constexpr bool STG = NCHT <= 4;
    u32 wnx[NCHT];
#pragma unroll
    for (int c = 0; c < NCHT; c++) {
        wnx[c] = 0u;
        const int u = t_lo - s2v[c];
        if (!STG && c < nch && u >= 0 && u < tcv[c]) wnx[c] = dptr[c][u];
    }
    __shared__ u32 tagw[STG ? NCHT * TG_BLK * 64 : 1];
    typedef u32 tag_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
    tag_u32x4 tnx[STG ? NCHT : 1][4];
    auto request_tags = [&](int tb) {  // positions tb .. tb + 15 of my alignments
#pragma unroll
        for (int c = 0; c < (STG ? NCHT : 1); c++) {
            const int u = tb - s2v[c];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                tnx[c][q] = (tag_u32x4){0u, 0u, 0u, 0u};
                // (a partly covered group is loaded whole: the words beside the alignment's
                // own belong to its neighbours or to the buffer's padding and are masked by
                // `covd` below)
                if (c < nch && tb < t_hi && u + 4 * q + 3 >= 0 && u + 4 * q < tcv[c])
                    tnx[c][q] = *reinterpret_cast<const tag_u32x4 *>(dptr[c] + (u + 4 * q));
            }
        }
    };
    if (STG) request_tags(t_lo);

    // The segment's links are written back to back from its first position's link slot: the
    // slots k_tscan hands out are sized by the tags (one per alignment and column) and the
    // links of a position are a fraction of that, so the gaps collect at the END of every
    // segment's range and the score kernels stage whole runs of levels with one copy.
    u32 out = (u32)__builtin_amdgcn_readlane((int)xr[0].link_start, 0);
    for (int t = t_lo; t < t_hi; t++) {
        const int j = t - t_lo;
        if (STG && (j & (TG_BLK - 1)) == 0) {  // a new block of 16 positions
            __syncthreads();
#pragma unroll
            for (int c = 0; c < (STG ? NCHT : 1); c++) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    tagw[(c * TG_BLK + 4 * q + 0) * 64 + lane] = tnx[c][q].x;
                    tagw[(c * TG_BLK + 4 * q + 1) * 64 + lane] = tnx[c][q].y;
                    tagw[(c * TG_BLK + 4 * q + 2) * 64 + lane] = tnx[c][q].z;
                    tagw[(c * TG_BLK + 4 * q + 3) * 64 + lane] = tnx[c][q].w;
                }
            }
            __syncthreads();
            request_tags(t + TG_BLK);
        }
        u32 x_lvl, x_link, x_cn;
        {
            const FaTInfo &xs = xr[0], &xt = xr[1];
            const u32 cn0 = (u32)xs.cov | ((u32)xs.nlev << 16), cn1 = (u32)xt.cov | ((u32)xt.nlev << 16);
            if (j < 64) {
                x_lvl = (u32)__builtin_amdgcn_readlane((int)xs.lvl_start, j);
                x_link = (u32)__builtin_amdgcn_readlane((int)xs.link_start, j);
                x_cn = (u32)__builtin_amdgcn_readlane((int)cn0, j);
            } else {
                x_lvl = (u32)__builtin_amdgcn_readlane((int)xt.lvl_start, j - 64);
                x_link = (u32)__builtin_amdgcn_readlane((int)xt.link_start, j - 64);
                x_cn = (u32)__builtin_amdgcn_readlane((int)cn1, j - 64);
            }
        }
        FaTInfo x;
        x.lvl_start = x_lvl; x.link_start = x_link; x.cov = (u16)(x_cn & 0xffffu); x.nlev = (u16)(x_cn >> 16);
        // tag words of the lanes covering t (requested one position ago); request t + 1
        bool covd[NCHT];
        u32 wtag[NCHT];
#pragma unroll
        for (int c = 0; c < NCHT; c++) {
            covd[c] = false; wtag[c] = wnx[c];
            if (c < nch) {
                const int u = t - s2v[c];
                covd[c] = u >= 0 && u < tcv[c];
                if (STG) wtag[c] = tagw[(c * TG_BLK + (j & (TG_BLK - 1))) * 64 + lane];
                else if (t + 1 < t_hi && u + 1 >= 0 && u + 1 < tcv[c]) wnx[c] = dptr[c][u + 1];
            }
        }
        if (x.cov == 0) continue;
        const int sb = (int)(((u32)__builtin_amdgcn_readlane((int)seedv, j >> 4) >> (2 * (j & 15))) & 3u);
        int nins[NCHT], base0[NCHT];
#pragma unroll
        for (int c = 0; c < NCHT; c++) {
            nins[c] = 0; base0[c] = 0;
            if (covd[c]) {
                nins[c] = tag_nins(wtag[c]);
                base0[c] = (wtag[c] & TAG_DEL) ? 4 : sb;
            }
        }
        bool plain = plain_seg;
#pragma unroll
        for (int c = 0; c < NCHT; c++) plain = plain && fa_ballot(nins[c] > INL) == 0ull;
        for (int dl = 0; dl < (int)x.nlev; dl++) {
            // key of every participating lane (-1 = none):
            //   node base | prev base << 3 | prev delta << 6 | start << 14
            // and its link word without the count (falcon.c:232-263 update_col):
            //   count | node base << LW_NB_SHIFT | prev score index (delta * 5 + base) << LW_PIDX_SHIFT | start << LW_START_BIT
            int key[NCHT];
            u32 wv[NCHT];
            if (plain) {
                // The falcon_sense path with runs of <= INL inserted bases (every position
                // but a handful): the same key and word as below, by selects -- every divergent
                // `if` costs an exec-mask round trip on the scalar pipe that bounds this kernel.
                // (covd is false in the chunks beyond nch.)
#pragma unroll
                for (int c = 0; c < NCHT; c++) {
                    if (dl == 0) {
                        const bool first = t == s2v[c];  // first column: no predecessor (falcon.c:434)
                        const int kprev = first ? ((5 << 3) | (1 << 14)) : ((pbv[c] << 3) | (pnv[c] << 6));
                        const u32 wprev = first ? (1u << LW_START_BIT) : ((u32)(pnv[c] * 5 + pbv[c]) << LW_PIDX_SHIFT);
                        key[c] = covd[c] ? (base0[c] | kprev) : -1;
                        wv[c] = covd[c] ? (((u32)base0[c] << LW_NB_SHIFT) | wprev) : 0u;
                        const bool upd = covd[c] && nins[c] == 0;
                        pbv[c] = upd ? base0[c] : pbv[c];
                        pnv[c] = upd ? 0 : pnv[c];
                    } else {
                        const bool part = covd[c] && nins[c] >= dl;
                        const int b = (int)((wtag[c] >> (2 * (dl - 1))) & 3u);
                        const int pb = dl == 1 ? base0[c] : (int)((wtag[c] >> (2 * (dl - 2))) & 3u);
                        key[c] = part ? (b | (pb << 3) | ((dl - 1) << 6)) : -1;
                        wv[c] = part ? (((u32)b << LW_NB_SHIFT) | ((u32)((dl - 1) * 5 + pb) << LW_PIDX_SHIFT)) : 0u;
                        const bool upd = part && nins[c] == dl;
                        pbv[c] = upd ? b : pbv[c];
                        pnv[c] = upd ? dl : pnv[c];
                    }
                }
            } else
#pragma unroll
            for (int c = 0; c < NCHT; c++) {
                key[c] = -1; wv[c] = 0;
                if (c < nch) {
                    const bool nocol = leadv[c] && t == s2v[c];  // insertion-only position
                    if (dl == 0) {
                        if (covd[c] && !nocol) {
                            if (t == s2v[c] && (t == 0 || !A.first_links_back)) {
                                // first column: no predecessor (p_t_pos == -1, falcon.c:434)
                                key[c] = base0[c] | (5 << 3) | (1 << 14);
                                wv[c] = ((u32)base0[c] << LW_NB_SHIFT) | (1u << LW_START_BIT);
                            } else if (t == s2v[c]) {
                                // unitig mode: get_align_tags adds the read's offset to its
                                // initial p_t_pos of -1 (:140), so the first column of a read
                                // placed at t > 0 links to (t - 1, delta 0) with the '.' base,
                                // which the scorer reads as '-' (:431)
                                key[c] = base0[c] | (4 << 3) | (0 << 6) | (1 << 15);
                                wv[c] = ((u32)base0[c] << LW_NB_SHIFT) | ((u32)4 << LW_PIDX_SHIFT);
                            } else {
                                key[c] = base0[c] | (pbv[c] << 3) | (pnv[c] << 6);
                                wv[c] = ((u32)base0[c] << LW_NB_SHIFT) | ((u32)(pnv[c] * 5 + pbv[c]) << LW_PIDX_SHIFT);
                            }
                        }
                    } else if (covd[c] && nins[c] >= dl) {
                        const int b = tag_ins_base(A, insoff[c], wtag[c], dl);
                        const int pb = dl == 1 ? base0[c] : tag_ins_base(A, insoff[c], wtag[c], dl - 1);
                        key[c] = b | (pb << 3) | ((dl - 1) << 6);
                        wv[c] = ((u32)b << LW_NB_SHIFT) | ((u32)((dl - 1) * 5 + pb) << LW_PIDX_SHIFT);
                        if (nocol && dl == 1) {
                            // the alignment's very first tag: no predecessor on the
                            // falcon_sense path, (t, delta 0, '.' read as '-') in unitig mode
                            if (A.first_links_back) {
                                key[c] = b | (4 << 3) | (1 << 15);
                                wv[c] = ((u32)b << LW_NB_SHIFT) | ((u32)4 << LW_PIDX_SHIFT);
                            } else {
                                key[c] = b | (5 << 3) | (1 << 14);
                                wv[c] = ((u32)b << LW_NB_SHIFT) | (1u << LW_START_BIT);
                            }
                        }
                        if (nins[c] == dl) { pbv[c] = b; pnv[c] = dl; }
                    }
                    if (dl == 0 && covd[c] && nins[c] == 0) { pbv[c] = base0[c]; pnv[c] = 0; }
                }
            }
            // Lanes with equal keys are one link (update_col, falcon.c:232-263).  The groups are
            // peeled off in the order of their lowest lanes -- the leaders: inside every node that
            // is the reference's first-insertion order (Q5); the nodes of a level interleave, which
            // k_score2 does not mind (rounds 1-3 emitted them node by node: five more ballots and
            // loop tests per level on the scalar pipe that bounds this kernel).  The leader takes
            // the group's rank and size by v_writelane and stores the word.  (Grouping through an
            // LDS table -- atomic minimum for the leader, atomic add for the size -- was built and
            // measured this round: fewer instructions, 17.6 ms instead of 13.8: thirty lanes of one
            // group are thirty conflicting updates of one LDS word.)
            int n_link = 0;
            if constexpr (NCHT == 1) {
                int myrank = -1, mycnt = 0;
                u64 rem = fa_ballot(key[0] >= 0);
                while (rem) {
                    const int ldr = __builtin_ctzll(rem);
                    const int kk = __builtin_amdgcn_readlane(key[0], ldr);
                    const u64 m = fa_ballot(key[0] == kk);
                    const int cnt = __popcll(m);
                    rem &= ~m;
                    fa_writelane2(myrank, mycnt, n_link, cnt, ldr);
                    n_link++;
                }
                if (myrank >= 0) links[out + (u32)myrank] = wv[0] | (u32)mycnt;
            } else {
                // several chunks: the same, the leader in the lowest chunk that still holds lanes
                int myrank[NCHT], mycnt[NCHT];
                u64 rem[NCHT];
                bool any = false;
#pragma unroll
                for (int c = 0; c < NCHT; c++) {
                    myrank[c] = -1; mycnt[c] = 0;
                    rem[c] = (c < nch) ? fa_ballot(key[c] >= 0) : 0ull;
                    any = any || rem[c] != 0ull;
                }
                while (any) {
                    int c0 = 0;
#pragma unroll
                    for (int c = NCHT - 1; c >= 0; c--)
                        if (rem[c]) c0 = c;
                    int kk = 0, ldr = 0;
#pragma unroll
                    for (int c = 0; c < NCHT; c++)
                        if (c == c0) {
                            ldr = __builtin_ctzll(rem[c]);
                            kk = __builtin_amdgcn_readlane(key[c], ldr);
                        }
                    int cnt = 0;
                    any = false;
#pragma unroll
                    for (int c = 0; c < NCHT; c++) {
                        if (c < nch) {
                            const u64 m = fa_ballot(key[c] == kk);
                            cnt += __popcll(m);
                            rem[c] &= ~m;
                            any = any || rem[c] != 0ull;
                        }
                    }
#pragma unroll
                    for (int c = 0; c < NCHT; c++)
                        if (c == c0) fa_writelane2(myrank[c], mycnt[c], n_link, cnt, ldr);
                    n_link++;
                }
#pragma unroll
                for (int c = 0; c < NCHT; c++)
                    if (myrank[c] >= 0) links[out + (u32)myrank[c]] = wv[c] | (u32)mycnt[c];
            }
            out += (u32)n_link;
            nlk[x.lvl_start + (u32)dl] = (u16)n_link;
        }
    }
}
// ---------------------------------------------------------------------------
// k_backtrace: one wavefront per pile (falcon.c:494-528).
//
// The walk from the best node back along the best-predecessor pointers is a pointer chase --
// one LDS read and a handful of scalar instructions per node when that is ALL a step does.
// So the walk only collects: 64 nodes per round, lane s keeping the s-th node's record, out
// of a 64-level window of node records in LDS (the window below it is requested from HBM
// while this one is walked).  What the reference does per step -- the character by base and
// case, the skipped '-', the eqv difference to the next node's score, the output index --
// then happens for the round's 64 nodes at once, lanes = nodes (round 3 did it per step on
// the scalar unit: 7.2 ms per 3072 piles, five times the chase).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64, 8) void k_backtrace(MsaArgs A) {
    // A window of BT_WIN levels (5 node records each) and, per node of it, the node 1, 2, 4, .. 32
    // steps down the path that starts there (a local index, BT_OUT once the path has left the
    // window or ended)
    constexpr int NW = 5 * BT_WIN;
    constexpr u16 BT_OUT = 0xffffu;
    __shared__ __attribute__((aligned(8))) u32 win[2 * NW];
    __shared__ u16 jump[6][NW];
    const int lane = fa_lane();
    const int p = blockIdx.x;
    if (p >= A.n_pile) return;
    const FaPile pm = A.pile[p];
    const FaScoreOut so = A.score_out[p];
    const int T = pm.seed_len;
    const uint2 *nodes = reinterpret_cast<const uint2 *>(A.nodes + pm.node_off);
    FaPileOut po;
    po.len = 0; po.start = 2 * T;
    po.n_aligned = (int)(A.acc_first[p + 1] - A.acc_first[p]);
    po.err = so.err;
    po.g_best_h = so.g_h;
    if (!so.err && so.g_node >= 0 && po.n_aligned > 0) {
        char *oseq = A.out_seq + pm.out_off;
        int *oeqv = A.out_eqv + pm.out_off;
        const unsigned lim = (unsigned)T * 2u;
        unsigned index = 0;                // characters written so far
        int win_lo = 0, n_win = 0;         // node ids [win_lo, win_lo + n_win) are in LDS
        int pf_lo = 0, pf_n = 0;           // ... and these are on their way, BT_PF records per lane
        constexpr int BT_PF = NW / 64;
        uint2 pf[BT_PF];
#pragma unroll
        for (int q = 0; q < BT_PF; q++) { pf[q].x = 0; pf[q].y = 0; }
        auto request_below = [&](int first_node) {  // the BT_WIN levels under the window
            const int lvl_hi = first_node / 5 - 1;
            const int lvl_lo = max(0, lvl_hi - (BT_WIN - 1));
            pf_lo = lvl_lo * 5;
            pf_n = lvl_hi < 0 ? 0 : (lvl_hi - lvl_lo + 1) * 5;
#pragma unroll
            for (int q = 0; q < BT_PF; q++) {
                const int i = lane + 64 * q;
                pf[q] = nodes[pf_lo + min(i, max(pf_n, 1) - 1)];  // (unconditional, the index clamped)
            }
        };
        // The walk along the best-predecessor pointers is a chain of dependent reads, one node per
        // LDS round trip however it is written (rounds 1-4: ~430 clocks per node, 4 ms for 3072
        // piles and 20 ms for 1024 deeper ones).  But the nodes of a window are all there, so the
        // chain is walked by pointer doubling instead: jump[j][v] = the node 2^j steps below v, six
        // data-parallel passes over the window, and lane s then finds the path's s-th node in six
        // reads (the bits of s) -- 64 nodes for a dozen round trips.
        auto window_at = [&](int node) {  // node (wave-uniform) becomes the top of the window
            fa_wave_sync();
            if ((unsigned)(node - pf_lo) < (unsigned)pf_n) {  // the usual case: the window below
                win_lo = pf_lo;
                n_win = pf_n;
#pragma unroll
                for (int q = 0; q < BT_PF; q++) {
                    const int i = lane + 64 * q;
                    if (i < pf_n) { win[2 * i] = pf[q].x; win[2 * i + 1] = pf[q].y; }
                }
            } else {  // the first window, and the jump of a zero back pointer (Q4)
                const int lvl_hi = node / 5;
                const int lvl_lo = max(0, lvl_hi - (BT_WIN - 1));
                win_lo = lvl_lo * 5;
                n_win = (lvl_hi - lvl_lo + 1) * 5;
                for (int i = lane; i < n_win; i += 64) {
                    const uint2 v = nodes[win_lo + i];
                    win[2 * i] = v.x;
                    win[2 * i + 1] = v.y;
                }
            }
            win_lo = fa_uni(win_lo);
            n_win = fa_uni(n_win);
            request_below(win_lo);
            pf_lo = fa_uni(pf_lo);
            pf_n = fa_uni(pf_n);
            fa_wave_sync();
            // (every pass reads all of a lane's BT_PF nodes, then all of their targets, then writes:
            // two LDS round trips per pass -- a loop over the nodes made it two per NODE)
            const int top = n_win - 1;
            {  // one step: the back pointer, if it stays inside
                u32 lk[BT_PF];
#pragma unroll
                for (int q = 0; q < BT_PF; q++) lk[q] = win[2 * min(lane + 64 * q, top) + 1];
#pragma unroll
                for (int q = 0; q < BT_PF; q++) {
                    const int v = lane + 64 * q, prev = (int)(lk[q] >> 1) - 1 - win_lo;
                    if (v < n_win) jump[0][v] = (prev >= 0 && prev < n_win) ? (u16)prev : BT_OUT;
                }
            }
#pragma unroll
            for (int j = 1; j < 6; j++) {
                fa_wave_sync();
                u16 h[BT_PF], g[BT_PF];
#pragma unroll
                for (int q = 0; q < BT_PF; q++) h[q] = jump[j - 1][min(lane + 64 * q, top)];
#pragma unroll
                for (int q = 0; q < BT_PF; q++) g[q] = jump[j - 1][min((int)h[q], top)];
#pragma unroll
                for (int q = 0; q < BT_PF; q++) {
                    const int v = lane + 64 * q;
                    if (v < n_win) jump[j][v] = h[q] == BT_OUT ? BT_OUT : g[q];
                }
            }
            fa_wave_sync();
        };
        int node = so.g_node;
        window_at(node);
        bool first = true;  // the round holds the path's first node, whose character comes from g_ck (Q2)
        for (;;) {
            // ---- lane s: the path's s-th node from `node` on, while the path stays in the window
            u32 cur = (u32)(node - win_lo);
#pragma unroll
            for (int j = 5; j >= 0; j--) {
                const u32 nx = jump[j][min(cur, (u32)(NW - 1))];
                if ((lane >> j) & 1) cur = cur == BT_OUT ? cur : nx;
            }
            const bool in = cur != BT_OUT;
            const int n = __popcll(fa_ballot(in));  // (a prefix of the lanes, lane 0 always)
            int my_node = 0, my_score = 0, my_link = 0;
            if (in) {
                my_node = win_lo + (int)cur;
                my_score = (int)win[2 * cur];
                my_link = (int)win[2 * cur + 1];
            }
            // what follows the round's last node: nothing (the path ends), or the next round's first
            const int prev = (__builtin_amdgcn_readlane(my_link, n - 1) >> 1) - 1;
            const bool last = prev < 0;
            if (!last && (unsigned)(prev - win_lo) >= (unsigned)n_win) window_at(prev);
            const int after_score = last ? 0 : (int)win[2 * (prev - win_lo)];
            // ---- what the reference does per node, for all of them (falcon.c:494-528): node i
            // yields a character unless it is the path's last one (:517-519, Q1), the character
            // is skipped when it is '-'; eqv = (int)score - (int)next node's score (Q6)
            const int n_out = last ? n - 1 : n;
            int next_score = __builtin_amdgcn_update_dpp(after_score, my_score, 0x130, 0xf, 0xf, false);  // wave_shl:1
            if (lane == n - 1) next_score = after_score;  // (the node the next round starts with)
            const int ck = (first && lane == 0) ? so.g_ck : my_node % 5;
            const u32 letters = (my_link & 1) ? 0x54474341u /* "ACGT" */ : 0x74676361u /* "acgt" */;
            // 0..3: the base, upper case where the coverage allowed; 4: '-'; a link index >= 5
            // (the first node only) leaves the reference's initial '$' (Q2)
            const int bb = ck < 4 ? (int)((letters >> (8 * (ck & 3))) & 0xffu) : (ck == 4 ? '-' : '$');
            const bool is_char = lane < n_out && bb != '-';
            const u64 m = fa_ballot(is_char);
            const unsigned idx = index + (unsigned)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
            if (is_char && idx < lim) {  // (index >= lim ends the walk, :519)
                const unsigned pos = lim - 1u - idx;
                oseq[pos] = (char)bb;
                oeqv[pos] = my_score / 2 - next_score / 2;
            }
            index = min(lim, index + (unsigned)__popcll(m));
            first = false;
            if (last || index >= lim) break;
            node = prev;
        }
        po.len = (int)index;
        po.start = (int)(lim - index);
    }
    A.pile_out[p] = po;  // every lane stores the same record
}
#ifndef FA_EMU  // (the host side below is not part of what the SIMT emulator of tests/emu/simt runs)
void fa_touch_msa() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(k_tags));
}
// ---------------------------------------------------------------------------
static MsaArgs msa_args(const FaBatchDev &b, const FaMsaDev &m, unsigned min_cov) {
    MsaArgs A;
    A.words = b.words; A.seq = b.seq; A.pile = b.pile; A.range = b.range; A.aln = b.aln;
    A.script = b.script; A.script_off = b.script_off;
    A.ta = m.ta; A.acc_first = m.acc_first; A.n_acc_total = m.n_acc_total; A.n_pile = b.n_pile;
    A.tcov = m.tcov; A.desc = m.desc; A.insb = m.insb; A.seg_cnt = m.seg_cnt; A.seg_base = m.seg_base; A.seg_first = m.seg_first; A.bound = m.bound; A.t_off = m.t_off;
    A.tinfo = m.tinfo; A.links = m.links; A.link_off = m.link_off; A.link_cap = m.link_cap;
    A.lvl_nlink16 = m.lvl_nlink16; A.nodes = b.nodes;
    A.score_ovf = m.score_ovf; A.score_out = m.score_out;
    A.out_seq = b.out_seq; A.out_eqv = b.out_eqv; A.pile_out = b.pile_out;
    A.seg_pile = m.seg_pile; A.seg_t0 = m.seg_t0; A.n_seg = m.n_seg; A.min_cov = min_cov;
    A.wide_count = m.wide_count; A.wide_list = m.wide_list;
    A.first_links_back = m.first_links_back;
    A.force_generic = m.force_generic;
    A.only_redo = 0;
    A.links_old = m.links_mode;
    return A;
}

// The graph-building half: tags, position scan, links -- throughput kernels.
// ev_tags / ev_links: recorded after k_tags + k_tscan and after k_links.
void fa_launch_msa_front(const FaBatchDev &b, const FaMsaDev &m, unsigned min_cov, hipStream_t s,
                         hipEvent_t ev_tags, hipEvent_t ev_links) {
    if (b.n_pile == 0) return;
    MsaArgs A = msa_args(b, m, min_cov);
    (void)hipMemsetAsync(m.seg_cnt, 0, (size_t)m.n_seg * 2 * sizeof(int), s);
    if (m.n_acc_total > 0) hipLaunchKernelGGL(k_tags, dim3(m.n_acc_total), dim3(64), 0, s, A);
    hipLaunchKernelGGL(k_sscan, dim3(b.n_pile), dim3(64), 0, s, A);
    if (ev_tags) (void)hipEventRecord(ev_tags, s);
    if (m.n_seg > 0) {
        // (the list heads)
        for (int l = 0; l < 6; l++) (void)hipMemsetAsync(m.wide_count + l * (m.n_seg + 1), 0, sizeof(int), s);
        fa_launch_links2(A, s);
        const int wide_grid = m.n_seg < 8192 ? m.n_seg : 8192;
        hipLaunchKernelGGL(k_links<1>, dim3(wide_grid), dim3(64), 0, s, A);
        hipLaunchKernelGGL(k_links<2>, dim3(wide_grid), dim3(64), 0, s, A);
        hipLaunchKernelGGL(k_links<4>, dim3(wide_grid), dim3(64), 0, s, A);
        hipLaunchKernelGGL(k_links<8>, dim3(wide_grid), dim3(64), 0, s, A);
        hipLaunchKernelGGL(k_links<16>, dim3(wide_grid), dim3(64), 0, s, A);
    }
    if (ev_links) (void)hipEventRecord(ev_links, s);
}

// The sequential half: one wavefront per pile walks the graph (scores, then the best path
// backwards).  Latency-bound at a few wavefronts per SIMD -- the engine runs it on a stream
// of its own, beside the next batch's throughput kernels.
void fa_launch_msa_back(const FaBatchDev &b, const FaMsaDev &m, unsigned min_cov, hipStream_t s,
                        hipEvent_t ev_score, hipEvent_t ev_backtrace) {
    if (b.n_pile == 0) return;
    MsaArgs A = msa_args(b, m, min_cov);
    if (m.score_mode == 0 && !m.force_generic) {
        fa_launch_score2(A, s);
        A.only_redo = 1;
    }
    fa_launch_score1(A, s);
    if (ev_score) (void)hipEventRecord(ev_score, s);
    hipLaunchKernelGGL(k_backtrace, dim3(b.n_pile), dim3(64), 0, s, A);
    if (ev_backtrace) (void)hipEventRecord(ev_backtrace, s);
}
#endif
