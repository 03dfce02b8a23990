// k_align2_core.h -- banded O(ND) alignment with trace-back, TWO alignments per wavefront.
//
// Restates align() of the reference (src/c/DW_banded.c:115-330) like k_align.hip does; what
// is different is how the work sits on the machine.  A band row of the falcon_sense
// workload holds ~27 diagonals, so one alignment per wavefront leaves 37 of 64 lanes (and
// the same share of every instruction issued) idle.  Here a wavefront carries two
// alignments ("tracks"): track 0 in the low lanes, track 1 in the high lanes, the boundary
// (`split`) movable.  One instruction stream advances both by one row per iteration; all
// lane sets are 64-bit scalar masks that simply hold both tracks' bits.
//
//   lanes      lane l of a track holds diagonal Kc + 2 l of the row it computed last.  Rows
//              alternate between two phases so that a band does not climb the lanes: an
//              even iteration computes diagonals Kc - 1 + 2 l (V[k-1] comes from lane l - 1,
//              V[k+1] is the lane's own value), an odd one Kc + 1 + 2 l (V[k-1] own,
//              V[k+1] from lane l + 1).  A band only moves when the alignment's diagonal
//              drifts; when a band reaches the edge of its share of the wave the row loop
//              lays both out again (`a2_replace`); when the two bands no longer fit together
//              it parks the narrower track (`a2_park`: its last row waits in a spare
//              register) and runs the wider one alone until they fit again (`a2_join`).
//              All three work on the row loop's own state; the wavefront's event loop
//              (`a2_wave`) fetches, finishes, and handles what they decline.
//   tape       the wavefront writes ONE record per iteration, shared by both tracks, into a
//              ring in its arena slot: 64 one-byte cells (the snake length of each lane's
//              cell, 255 = "look it up in the escape list"), the 64 from_above bits, and per
//              track the diagonal of lane 0 (or "took no part").  4 bytes per DP cell
//              became 1: the trace-back never needed x, only the snake length and the
//              direction of the cells on the path (the edit script IS that list).
//   trace      (DW_banded.c:264-319) walks a track's iterations backwards, 64 at a time:
//              the diagonal chain is resolved from the from_above bits, then all 64 rows
//              fetch their cell byte in parallel and emit `(snake << 1) | from_above`.
//
// Rows wider than 60 diagonals are computed alone through an LDS ring (`a2_wide`).  What is
// left outside -- a band that stays wide while a neighbour waits, an alignment too long for
// the tape ring, too many snakes of >= 255 bases -- is handed back (FaAln.err = 2)
// and repeated by the general one-alignment-per-wavefront kernel (k_align.hip) in a
// worst-case slot, the same way alignments that outgrew their slot always were.
//
// Written against fa_wave.h only (per-lane values, masks, cross-lane primitives), so that
// tests/emu can run this very source on 64-element arrays against the CPU oracle.
#pragma once
#ifndef W_FN
#error "include fa_wave.h (the product's, or the lane emulator's twin under tests/emu) before k_align2_core.h"
#endif

#define A2_NEG (-(1 << 29))      // "x" of the lanes outside a band's hull: loses every comparison
#define A2_INVALID 0x80000000u  // tape record: the track computed no row in this iteration
// what track 1's band-filter keys (x + y) carry on top of track 0's, so that one prefix maximum over
// the lanes serves both: above every real key (x + y < 2^18), and small enough that a lane
// outside the bands -- x = A2_NEG + 1 there, its key 2 A2_NEG + .. -- stays negative with it
#define A2_TOP 0x20000000u
#define A2_CONT 0x80000001u     // ... or: cells 64.. of the row the iteration before began (wide rows)
#define A2_WIDE_RING 256        // LDS ring of a wide row's V values, per row parity (<= 191 live diagonals)
// LDS words of a wavefront: four windows of 256 words on the two sequences of the two tracks
// (k_align2_rows.h), the word bases of the windows behind them -- and, while a track computes
// its wide rows, the two rings of a2_wide in words 256.. (the windows are filled again then)
// (4096 bytes in all: the word bases sit in the last four words of the last window, which ends four
// words early for them -- with 4 KB a wavefront, 32 of them leave a CU 32 KB of LDS for the kernels
// of the batch before, which run beside this one)
#define A2W_HDR 1020
#define A2W_INVALID 0xffffffffu
#define A2_LDS_WORDS 1024
// Placement policy, measured on the bench workload [MI355X] (scripts/r03_variants.sh +
// r03_sweep.sh, profiles/r03_policy_sweep.txt; taken while parking and joining still went
// through the event loop, ~800 instructions a trip -- now ~100 and a call): leaving the pair
// halves the rows per iteration, laying the two bands out again inside the row loop costs a
// few dozen instructions -- so two running tracks
// stay paired as long as their bands fit the 64 lanes AT ALL (0 free lanes: 54.2 ms; 2: 56.6;
// 4: 58.8; 6: 60.6), and a parked neighbour is looked at every 8 iterations (16: +0.5 ms,
// 32: +1.9) and joins with 6 lanes to spare (2..8: within 0.5 ms).
#ifndef A2_FREE_MIN
#define A2_FREE_MIN 0           // free lanes two running tracks need to stay paired
#endif
#ifndef A2_FREE_JOIN
#define A2_FREE_JOIN 6          // ... and to (re)join a parked or a new track
#endif
static_assert(A2_FREE_JOIN >= 4, "a new track's first cell needs its V[k+1] lane on its side of the boundary");
#define A2_MAX_N 60             // widest band a track may have (alone in the wave)
#ifndef A2_LOOK_EVERY
#define A2_LOOK_EVERY 8         // iterations a track runs alone before the wave checks whether its parked neighbour fits again
#endif
#define A2_ESC_CAP 1024         // escape entries per slot (snakes of >= 255 bases)
#define A2_WIDE_PATIENCE 512    // wide rows a track may take while its neighbour waits (reads that align badly open the
                                // band by a diagonal per row: from 61 to the 151 that end them takes ~200 rows)

struct A2Args {
    const u32 *words;
    const FaSeq *seq;
    const FaPile *pile;
    const FaRange *range;
    const int *order;
    int n_work;
    u32 word_base;     // `words` points at packed word `word_base` of the batch (base indices are 32 bits: a batch of
                       // 2^28 words or more is aligned in several launches, each on a stretch of whole piles)
    int *counter;
    u32 *cells;        // per slot: ring x 64 bytes, as words: ((it >> 2) & (ring/4 - 1)) * 64 + lane
    u32 *recs;         // per slot: ring x 16 bytes {mask lo, mask hi, K track 0, K track 1}
    u64 *esc;          // per slot: A2_ESC_CAP entries {(it << 6 | lane), snake length}
    u64 slot_words;    // u32 words between two slots' cells
    u32 ring;          // iterations per slot, a power of two >= 256
    u32 *script;
    const u64 *script_off;
    FaAln *aln;
    int band;
    double max_diff;
    unsigned long long *stats;  // A2_STAT_N counters (see A2_STAT_*)
    int debug;                  // timing experiments only: 1 = no trace-back (the scripts stay unwritten)
    // what the wavefront's event loop hands alignments back at: A2_ESC_CAP entries on the slot's
    // escape list, A2_WIDE_PATIENCE wide rows beside a waiting neighbour -- lower in tests that
    // force those exits (FALCON_AMD_ESC_CAP, FALCON_AMD_WIDE_PATIENCE)
    int esc_cap, wide_patience;
};
// pair / single iterations, placements, parkings, hand-backs, wide rows, wide episodes, recenterings
enum { A2_STAT_PAIR_IT = 0, A2_STAT_SINGLE_IT, A2_STAT_PLACE, A2_STAT_PARK, A2_STAT_BAIL,
       A2_STAT_ESC, A2_STAT_EXT, A2_STAT_TRACKS,
       // the hand-backs (A2_STAT_BAIL) by cause: the tape (an alignment whose rows do not fit the
       // ring, at the queue or once its share is used up), wide rows (a band beyond 191 diagonals, or
       // wide for longer than a waiting neighbour has patience), the slot's escape list full
       A2_STAT_BAIL_TAPE, A2_STAT_BAIL_WIDE, A2_STAT_BAIL_ESC, A2_STAT_N };

enum { A2_IDLE = 0, A2_RUN = 1, A2_PARKED = 2 };

struct A2Track {
    int state;
    int g;              // sequence index
    int q_len, t_len;
    int max_d;
    int d;              // rows computed
    int kc;             // diagonal of lane 0 of the last computed row (in v_x or v_park)
    int li, hin;        // lanes of the band hull of the last computed row (what passed the band filter)
    int best;           // best x + y so far (DW_banded.c best_m); -1 before row 0
    u32 it0;            // tape iteration of row 0
    u32 cells;          // DP cells evaluated
    u32 qb, tb;         // base index (into words[], 16 per word) of the two windows' first bases
    u32 script_lo, script_hi;  // word offset of its script
};

// ---------------------------------------------------------------------------------------
// the snake past its first 16 bases (rare: ~2 % of the rows): out of line, so that its
// loop and registers stay out of the row loop
// ---------------------------------------------------------------------------------------
struct A2Snake { vi x, y; vu m; };

template <class DUMMY>
W_NOINLINE A2Snake a2_snake_more(const u32 *words, vu vqb, vu vtb, vi vqlen, vi vtlen, vi x, vi y, vu m) {
    vu last = m;
    for (;;) {
        const u64 go = w_ballot(last == 16u);
        if (!go) break;
        W_WHERE(go) {
            const vu qa = vqb + (vu)x, ta = vtb + (vu)y;
            vu qlo, qhi, tlo, thi;
            w_load_pair(words, qa >> 4, qlo, qhi);
            w_load_pair(words, ta >> 4, tlo, thi);
            const vu qw = w_alignbit(qhi, qlo, qa << 1), tw = w_alignbit(thi, tlo, ta << 1);
            const vu lim = (vu)w_min(vqlen - x, vtlen - y);
            const vu s = w_minu(w_minu(w_ffbl(qw ^ tw) >> 1, lim), 16u);
            x = x + (vi)s;
            y = y + (vi)s;
            m = m + s;
            last = s;
        }
    }
    A2Snake r;
    r.x = x; r.y = y; r.m = m;
    return r;
}

// snakes of >= 255 bases do not fit the cell byte: (iteration, lane, length) of the lanes
// `big` go to the slot's escape list; returns the new entry count (A2_ESC_CAP + 1: the list
// is full).  Called outside any W_WHERE: a wave-uniform value assigned under a lane mask
// would count as divergent from there on.
template <class DUMMY>
W_NOINLINE int a2_escape(u64 *esc, int n_esc, u32 it, vu m, u32 big_lo, u32 big_hi) {
    const u64 big = ((u64)w_uniu(big_hi) << 32) | w_uniu(big_lo);
    n_esc = w_uni(n_esc);
    it = w_uniu(it);
    const int n_new = w_popc(big);
    if (n_esc + n_new > A2_ESC_CAP) return A2_ESC_CAP + 1;
    const vi lane = w_lane();
    const vu rank = (vu)w_rank_in(big);  // among the escaping lanes
    W_WHERE(big) { w_store64(esc, (vu)n_esc + rank, m, (vu)(it << 6) | (vu)lane); }
    return n_esc + n_new;
}

// ---------------------------------------------------------------------------------------
// one band row of up to two tracks.  P: phase (iteration parity), J: byte of the cell word
// ---------------------------------------------------------------------------------------
// (Per-lane and wave-uniform state live in SEPARATE structs, here and for the wavefront's
// state below: SROA merges neighbouring fields of one struct into vector values, and a
// vector with one per-lane element makes every scalar in it count as divergent -- the whole
// row loop then lands on the vector unit.)
struct A2HotV {   // per lane
    vi vx;            // x of the last computed row (lane l: diagonal Kc + 2 l)
    vi vnegk;         // -(diagonal the lane computes in an even iteration)
    vu vqb, vtb;      // window starts
    vi vqlen, vtlen;
    vu vtop;          // A2_TOP in track 1's lanes (PAIR), else 0
    vu vacc;          // the cell bytes of up to 4 iterations
    vu rc_mlo, rc_mhi;  // ring of the last <= 64 iterations' from_above masks (lane = it & 63)
    vu vm;            // the snake lengths of the last row
};
struct A2Hot {    // wave-uniform
    u64 act;          // lanes holding a cell in the next row
    u64 in;           // lanes of the last row inside the band filter (their hulls: what the next bands grow from)
    int lo0, hi0, lo1, hi1;  // the two bands of the next row (set where the rows begin and end, not by the rows)
    int best0, best1; // best_m per track (track 1's carries the key's top bit)
    u32 cells0, cells1;
    int split;        // PAIR: first lane of track 1
    u64 zone1;        // PAIR: lanes >= split
    u64 forbid_to0, forbid_to1;  // lanes the band hull may not reach before an even / odd row
    u64 fin;          // lanes whose cell reached an end of a sequence (this row)
    u64 ev;           // fin | hull on a forbidden lane
    u64 act_row;      // the lanes of the row that raised `ev`
    u64 big;          // lanes of the last row whose snake is >= 255 bases long
    u32 it;
    int n_esc;
    u32 kb0, kb1;     // what the tape records say about the two tracks (lane-0 diagonal | A2_INVALID)
    u32 n_replace;    // in-loop re-placements (statistics)
};

// Returns the lanes that held a cell in this row.
template <int P, int J, bool PAIR>
W_FN u64 a2_row(A2Hot &h, A2HotV &hv, const u32 *words, u64 *esc, int band, u32 byte_sel = 0u) {
    const u64 act = h.act;
    // (the cells of a row are counted where the row is computed)
    if (PAIR) { h.cells0 += (u32)w_popc(act & ~h.zone1); h.cells1 += (u32)w_popc(act & h.zone1); }
    else h.cells0 += (u32)w_popc(act);
    // V[k-1] + 1 and V[k+1] of the previous row (DW_banded.c:190-196)
    vi a1, b;
    if (P == 0) {
        a1 = w_from_below(hv.vx) + 1;
        b = hv.vx;
    } else {
        a1 = hv.vx + 1;
        b = w_from_above(hv.vx);
    }
    // from_above: k == min_k, or k != max_k and V[k-1] < V[k+1] -- the lanes outside the
    // hull of the previous row hold A2_NEG, so the two edges of a band decide themselves:
    // at min_k V[k-1] is A2_NEG (from above), at max_k V[k+1] is (from below)
    // (Lane 0 has no lane below it -- an even row reads 0 there, not A2_NEG: a cell on lane 0
    // is its band's min_k by position, and with V[k+1] = 0, the low edge of an alignment's
    // first rows, the comparison alone would get it wrong.)
    const u64 fa = (P == 0 ? (w_ballot(a1 <= b) | 1ull) : w_ballot(a1 <= b)) & act;
    vi x = w_sel(fa, a1, b);
    vi y = (P == 0) ? x + hv.vnegk : x + hv.vnegk - 1;
    vu m = w_undef();  // (idle lanes: whatever -- their cell bytes are never read)
#ifdef A2_HOOK_ROW
    A2_HOOK_ROW(P, PAIR, h, hv, a1, b, x, y, act, fa);
#endif
    W_WHERE(act) {
        // the snake (:203-206), 16 bases per step on 2-bit packed words
        const vu qa = hv.vqb + (vu)x, ta = hv.vtb + (vu)y;
        vu qlo, qhi, tlo, thi;
        w_load_pair(words, qa >> 4, qlo, qhi);
        w_load_pair(words, ta >> 4, tlo, thi);
        const vu qw = w_alignbit(qhi, qlo, qa << 1), tw = w_alignbit(thi, tlo, ta << 1);
        const vu lim = (vu)w_min(hv.vqlen - x, hv.vtlen - y);
        m = w_minu(w_minu(w_ffbl(qw ^ tw) >> 1, lim), 16u);
        x = x + (vi)m;
        y = y + (vi)m;
        if (w_ballot(m == 16u)) {
            const A2Snake r = a2_snake_more<void>(words, hv.vqb, hv.vtb, hv.vqlen, hv.vtlen, x, y, m);
            x = r.x; y = r.y; m = r.m;
        }
    }
    // (snakes of >= 255 bases do not fit the cell byte: such a row raises an event, and where
    // the row loop ends the lengths go to the escape list and the bytes become 255 -- a2_long_snakes)
    hv.vm = m;
    // the cell byte: the snake length
    if (J >= 0) hv.vacc = w_put_byte<(J >= 0 ? J : 0)>(hv.vacc, m);
    else hv.vacc = w_put_byte_sel(hv.vacc, m, byte_sel);  // (J < 0: the position comes as a v_perm selector)
    // band (:228-243): keep the hull of the cells with x + y >= best_m - band.  One prefix
    // maximum serves both tracks: track 1's keys carry the top bit.  In the wait states of
    // its DPP steps (w_row_tail): the cells that reached an end of a sequence (:220), the
    // snakes that do not fit a cell byte, the row's from_above bits into the tape record.
    const vu key = w_selu(act, 0u, (vu)(x + y) + hv.vtop);
    u64 fin, big;
    vu keyb;
    const vu pm = w_row_tail(key, x, hv.vqlen, y, hv.vtlen, m, act, (u32)band, fa, (int)(h.it & 63u),
                             hv.rc_mlo, hv.rc_mhi, fin, big, keyb);
    u64 in;
    if (PAIR) {
        h.best0 = max(h.best0, (int)w_readlaneu(pm, h.split - 1));
        h.best1 = (int)max((u32)h.best1, w_readlaneu(pm, 63));
        const vu vbest = w_selu(h.zone1, (vu)h.best0, (vu)h.best1);
        in = w_ballot(keyb >= vbest) & act;
    } else {
        h.best0 = max(h.best0, (int)w_readlaneu(pm, 63));
        in = w_ballot(keyb >= (vu)h.best0) & act;
    }
#ifdef A2_HOOK_IN
    A2_HOOK_IN(PAIR, in, h.zone1, m, act);
#endif
    // the next row's bands: the hulls, one diagonal wider on either side -- in lanes: one
    // lane down before an odd row, one lane up before an even one.  (Lane numbers never
    // become registers here: masks only.)
    u64 hull;
    if (PAIR) {
        const u64 in0 = in & ~h.zone1, in1 = in & h.zone1;
        const int span0 = w_span(in0), span1 = w_span(in1);
        hull = w_lanes(w_lowest(in0), span0) | w_lanes(w_lowest(in1), span1);
    } else {
        const int span0 = w_span(in);
        hull = w_lanes(w_lowest(in), span0);
    }
    // (the two hulls keep a lane's distance from the boundary between the tracks, so one
    // shift of the whole mask widens both; a hull on the wave's first or last lane loses the
    // bit that falls off -- the row raises an event then, and its own values are kept)
    const u64 nact = (P == 0) ? (hull | (hull >> 1)) : (hull | (hull << 1));
    hv.vx = w_sel(hull, A2_NEG, x);
    h.act = nact;
    h.in = in;
    h.fin = fin;
    h.big = big;
    h.ev = fin | big | (in & (P == 0 ? h.forbid_to1 : h.forbid_to0));
    h.it++;
    return act;
}

// ---------------------------------------------------------------------------------------
// the wavefront's state between two runs of the row loop
// ---------------------------------------------------------------------------------------
struct A2Lanes {   // per lane
    vi vx, vpark;
    vu vacc, rc_mlo, rc_mhi, rc_k0, rc_k1;
    vi vnegk, vqlen, vtlen;
    vu vqb, vtb, vtop;
};
struct A2Wave {    // wave-uniform
    A2Track T0, T1;
    u32 it;            // the next tape iteration
    int pair;          // the current placement runs both tracks
    int split;
    int n_esc;
    bool more;         // the work queue may hold more
    // the slot
    u32 *cells, *recs;
    u64 *esc;
    // statistics of this wavefront
    u32 st_pair, st_single, st_place, st_park, st_bail, st_tracks, st_wide, st_wide_rows, st_recenter;
    u32 st_bail_tape, st_bail_wide, st_bail_esc;
};

W_FN void a2_result(const A2Args &A, int g, int err, int aligned, int dist, int q_e, int t_e, int n_ins,
                    long long cells) {
    FaAln r;
    r.dist = dist; r.q_e = q_e; r.t_e = t_e;
    r.size = aligned ? (q_e + t_e + dist) / 2 : 0;  // DW_banded.c:248
    r.accept = (aligned && r.size > 500 && (double)r.dist / (double)r.size < A.max_diff) ? 1 : 0;  // falcon.c:629
    r.n_ins = n_ins; r.aligned = aligned; r.err = err; r.cells = cells;
    w_store_aln(A.aln + g, r);
}

// A track's LDS windows (k_align2_rows.h) no longer hold what its rows will read: a new
// alignment in the track, or the LDS was lent to the wide rows.  (The rows' statement, which
// the lane emulator runs, reads global memory: nothing to do there.)
W_FN void a2_win_invalidate(int ti) {
#if defined(W_ROWS_ASM) && !defined(A2_ROWS_C)
    u32 *l = w_lds();
    W_WHERE(3ull) { w_lds_store(l, (vu)(A2W_HDR + 2 * ti) + (vu)w_lane(), (vu)A2W_INVALID); }
#else
    (void)ti;
#endif
}

// The next alignment of the work queue into `t` (state A2_RUN, no row computed yet); false
// when the queue is empty.  Sequences that take no part (seeds, reads the window filter
// dropped) get their empty summary here.
W_FN bool a2_fetch(const A2Args &A, A2Track &t) {
    for (;;) {
        const int wi = w_atomic_add_lane0(A.counter, 1);
        if (wi >= A.n_work) return false;
        const int g = w_uni(A.order[wi]);
        const int q_idx = w_uni(A.seq[g].idx);
        const u32 q_woff = w_uniu(A.seq[g].woff) - A.word_base;
        const int pile_id = w_uni(A.seq[g].pile);
        const int s1 = w_uni(A.range[g].s1), e1 = w_uni(A.range[g].e1);
        const int s2 = w_uni(A.range[g].s2), e2 = w_uni(A.range[g].e2);
        const int rg_ok = w_uni(A.range[g].ok);
        if (q_idx == 0 || !rg_ok) {
            a2_result(A, g, 0, 0, 0, 0, 0, 0, 0);
            continue;
        }
        const int seed_g = w_uni(A.pile[pile_id].first);
        const u32 t_woff = w_uniu(A.seq[seed_g].woff) - A.word_base;
        const int q_len = e1 - s1, t_len = e2 - s2;  // falcon.c:626-627
        const int max_d = w_uni((int)(0.3 * (double)(q_len + t_len)));  // DW_banded.c:149
        if (max_d <= 0) {  // no row is ever computed (:183)
            a2_result(A, g, 0, 0, 0, 0, 0, 0, 0);
            continue;
        }
        // its rows must fit the tape: by their bound, or -- alignments that succeed walk two
        // thirds of it at most -- by half the bound with an eighth of the ring to spare (one
        // that then goes on and on is handed back when the tape is used up)
        if ((u32)max_d + 192u > A.ring && (u32)max_d / 2u + A.ring / 8u > A.ring) {
            a2_result(A, g, 2, 0, 0, 0, 0, 0, 0);
            w_stat_add(A.stats + A2_STAT_BAIL, 1ull);
            w_stat_add(A.stats + A2_STAT_BAIL_TAPE, 1ull);
            continue;
        }
        t.state = A2_RUN;
        t.g = g;
        t.q_len = q_len; t.t_len = t_len;
        t.max_d = max_d;
        t.d = 0;
        t.kc = 0; t.li = 0; t.hin = 0;
        t.best = -1;
        t.it0 = 0;
        t.cells = 0;
        t.qb = q_woff * 16u + (u32)s1;
        t.tb = t_woff * 16u + (u32)s2;
        const u64 so = A.script_off[g];
        t.script_lo = w_uniu((u32)so);
        t.script_hi = w_uniu((u32)(so >> 32));
        return true;
    }
}

// the band of a track's next row in its current lane coordinates, given the phase of the
// next iteration (a track without rows: one cell, wherever it will be put)
W_FN void a2_next_band(const A2Track &t, int p, int &lo, int &hi) {
    if (t.d == 0) { lo = 0; hi = 0; return; }
    if (p == 0) { lo = t.li; hi = t.hin + 1; }
    else        { lo = t.li - 1; hi = t.hin; }
}

// Place the tracks in the wave for the iterations to come: both side by side when they fit
// (with room to drift), else the wider one alone and the other parked.  Returns 0, or 1 + t
// when track t cannot run even alone (the caller hands it back).
W_FN int a2_place(A2Wave &w, A2Lanes &wl) {
    const int p = (int)(w.it & 1u);
    const bool have0 = w.T0.state != A2_IDLE, have1 = w.T1.state != A2_IDLE;
    int lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0;
    if (have0) a2_next_band(w.T0, p, lo0, hi0);
    if (have1) a2_next_band(w.T1, p, lo1, hi1);
    const int n0 = hi0 - lo0 + 1, n1 = hi1 - lo1 + 1;
#ifdef A2_HOOK_PLACE
    A2_HOOK_PLACE(have0, n0, have1, n1);
#endif
    bool run0 = have0, run1 = have1;
    if (have0 && have1) {
        const int free_lanes = 64 - n0 - n1;
        // (a track without rows joins like a parked one: the lane its first cell reads
        // V[k+1] = 0 from lies a lane above its band before an odd row, and must be on its
        // side of the boundary -- A2_FREE_JOIN >= 4 puts two lanes between band and boundary)
        const bool paired_now = w.pair && w.T0.state == A2_RUN && w.T1.state == A2_RUN &&
                                w.T0.d > 0 && w.T1.d > 0;
        if (free_lanes < (paired_now ? A2_FREE_MIN : A2_FREE_JOIN)) {
            if (n1 > n0) run0 = false; else run1 = false;   // the wider one goes on alone
        }
    }
    if (run0 && !run1 && n0 > A2_MAX_N) return 1;
    if (run1 && !run0 && n1 > A2_MAX_N) return 2;
    const bool pair = run0 && run1;
    int nl0 = 0, nl1 = 0, split;
    if (pair) {
        const int free_lanes = 64 - n0 - n1;
        const int g0 = free_lanes / 4, mid = free_lanes / 2;
        nl0 = g0;
        nl1 = g0 + n0 + mid;
        split = g0 + n0 + mid / 2;
    } else if (run0) {
        nl0 = (64 - n0) / 2;
        split = 64;
    } else {
        nl1 = (64 - n1) / 2;
        split = 0;
    }
    const vi lane = w_lane();
    const u64 zone1 = split >= 64 ? 0ull : (split <= 0 ? ~0ull : (~0ull << split));
    // the rows the tracks computed last, into their new lanes (a track without rows: zeros,
    // the reference's calloc'ed V, DW_banded.c:153)
    const int sh0 = lo0 - nl0, sh1 = lo1 - nl1;
    vi nx = 0;
    if (run0 && w.T0.d > 0) {
        const vi src = w.T0.state == A2_PARKED ? wl.vpark : wl.vx;
        nx = w_sel(zone1, w_gather_lanes(src, (lane + sh0) & 63), nx);
    }
    if (run1 && w.T1.d > 0) {
        const vi src = w.T1.state == A2_PARKED ? wl.vpark : wl.vx;
        nx = w_sel(zone1, nx, w_gather_lanes(src, (lane + sh1) & 63));
    }
    // a running track that stops here waits in vpark, in the lanes it has
    if ((have0 && !run0 && w.T0.state == A2_RUN) || (have1 && !run1 && w.T1.state == A2_RUN)) {
        wl.vpark = wl.vx;
        w.st_park++;
    }
    {   // outside the hulls of the rows just placed: A2_NEG (a track without rows: a single 0
        // where its first cell looks for V[k+1] -- its own lane before an even row, the lane
        // above before an odd one)
        u64 keep = 0ull;
        if (run0) keep |= w.T0.d > 0 ? w_lanes(w.T0.li - sh0, w.T0.hin - w.T0.li + 1) : (1ull << (nl0 + p));
        if (run1) keep |= w.T1.d > 0 ? w_lanes(w.T1.li - sh1, w.T1.hin - w.T1.li + 1) : (1ull << (nl1 + p));
        nx = w_sel(keep, A2_NEG, nx);
#ifdef A2_HOOK_KEEP
        A2_HOOK_KEEP(w, keep, run0, run1, sh0, sh1, nl0, nl1, p);
#endif
    }
    wl.vx = nx;
    if (run0) {
        if (w.T0.d == 0) {  // its first row: one cell, diagonal 0, on lane nl0
            w.T0.kc = -2 * nl0 - (p ? 1 : -1);
            w.T0.it0 = w.it;
        } else {
            w.T0.kc += 2 * sh0;
            w.T0.li -= sh0;
            w.T0.hin -= sh0;
        }
        w.T0.state = A2_RUN;
    } else if (have0) {
        w.T0.state = A2_PARKED;
    }
    if (run1) {
        if (w.T1.d == 0) {
            w.T1.kc = -2 * nl1 - (p ? 1 : -1);
            w.T1.it0 = w.it;
        } else {
            w.T1.kc += 2 * sh1;
            w.T1.li -= sh1;
            w.T1.hin -= sh1;
        }
        w.T1.state = A2_RUN;
    } else if (have1) {
        w.T1.state = A2_PARKED;
    }
    w.pair = pair ? 1 : 0;
    w.split = split;
    // per-lane constants of the lanes' track
    const int kb0 = w.T0.kc + p, kb1 = w.T1.kc + p;  // lane-0 diagonal of an odd iteration's row
    wl.vnegk = w_sel(zone1, -(kb0 - 1), -(kb1 - 1)) - 2 * lane;
    wl.vqlen = w_sel(zone1, w.T0.q_len, w.T1.q_len);
    wl.vtlen = w_sel(zone1, w.T0.t_len, w.T1.t_len);
    wl.vqb = w_selu(zone1, w.T0.qb, w.T1.qb);
    wl.vtb = w_selu(zone1, w.T0.tb, w.T1.tb);
    wl.vtop = pair ? w_selu(zone1, 0u, A2_TOP) : (vu)0u;
    // the tape says from here on which tracks take part and where their lane 0 is
    const u64 ahead = ~0ull << (w.it & 63u);
    wl.rc_k0 = w_selu(ahead, wl.rc_k0, run0 ? (u32)kb0 : A2_INVALID);
    wl.rc_k1 = w_selu(ahead, wl.rc_k1, run1 ? (u32)kb1 : A2_INVALID);
    w.st_place++;
    return 0;
}

// ---------------------------------------------------------------------------------------
// trace-back of a finished track (DW_banded.c:264-319): its rows, newest first, 64 tape
// iterations at a time.  TI: which of the two K fields of the tape records is the track's.
// Returns the number of query-only rows (FaAln.n_ins).
// ---------------------------------------------------------------------------------------
template <int TI>
W_FN int a2_trace(const A2Args &A, A2Wave &w, A2Lanes &wl, const A2Track &t, u32 it_f, int k_f, int fin_d) {
    if (A.debug & 1) return 0;
    const u32 ring_mask = A.ring - 1u, chunk_mask = (A.ring >> 2) - 1u;
    u32 *script = A.script + (((u64)t.script_hi << 32) | t.script_lo);
    const vi lane = w_lane();
    int kv = k_f;       // the path's diagonal at the newest row not yet resolved
    int r_top = fin_d;  // its row
    int n_ins = 0;
    for (u32 ih = it_f;; ih -= 64u) {
        vu itj = ih - (vu)lane;  // lane j looks at iteration ih - j
        const u64 have = w_ballot((vi)(itj - t.it0) >= 0);
        vu ra = 0u, rb = 0u, rc = 0u, rd = 0u;
        W_WHERE((A.debug & 8) ? 0ull : have) { w_load_x4(w.recs, itj & ring_mask, ra, rb, rc, rd); }
        const vu kb = TI ? rd : rc;
        const u64 valid = have & w_ballot((vi)kb > (vi)A2_CONT);  // (neither A2_INVALID nor A2_CONT)
        vi kc = (vi)kb - 1 + (vi)(itj & 1u);  // lane-0 diagonal of that iteration's row
        const int n_rows = w_popc(valid);
#ifdef A2_HOOK_TRACE
        A2_HOOK_TRACE(t, ih, have, valid, n_rows, r_top, kv, kb);
#endif
        if (n_rows > 0) {
            // The block's rows go to the TOP lanes, newest first: lanes base .. 63 (the chain below
            // is one unrolled instruction stream entered at lane `base`).  Iterations the track
            // took no part in (parked; continuation records of wide rows) drop out on the way.
            const int base = 64 - n_rows;
            if (base > 0) {
                const vi to = w_sel(valid, 0, base + w_rank_in(valid));  // (lane 0 holds no row when base > 0)
                kc = w_push_lanes(kc, to);
                ra = (vu)w_push_lanes((vi)ra, to);
                rb = (vu)w_push_lanes((vi)rb, to);
                itj = (vu)w_push_lanes((vi)itj, to);
            }
            // From a row on diagonal k = kc + 2 j the path goes to k + 1 (from_above: pre_k =
            // k + 1, DW_banded.c:190-196) or k - 1 of the row before -- lane l + 1, whose
            // lane-0 diagonal differs by an odd number: in lanes, j moves by one of two
            // per-row constants.  The chain itself is scalar work: one bit test and one add
            // per row, the records read lane by lane.
            // (dK + 1) / 2 if the path came from above, (dK - 1) / 2 if not: the second plus the bit.
            // (The oldest row, lane 63, has no row below it in this block: its step is 0, so what
            // the chain ends on is that row's lane plus its bit.)
            const vi dk = kc - w_from_above(kc);
            const vi step_dn = w_sel(1ull << 63, (dk - 1) >> 1, 0);
            vi my_lane = 0;
            const int j_top = (kv - w_readlane(kc, base)) >> 1;
            // no wide row in the block: no continuation record in it, and the newest row (whose
            // continuation records would lie in the block done before this one) is not one
            const bool plain = (w_ballot(kb == A2_CONT) & have) == 0ull && j_top < 64;
            if (plain && !(A.debug & 2)) {
                // six instructions a row (w_chain): three records read by v_readlane, the lane kept
                // by v_writelane, s_bitcmp1_b64 + s_addc_u32
                // (round 6: in a frame that moves with the rows -- index i = lane + s, s chosen so that the
                // path's move is the mask's bit plus the constant c = -(l & 1) of the unrolled stream: no
                // step to read, five instructions a row.  s[l + 1] = s[l] + c[l] - step[l], s[base] puts the
                // path on index 31, from where it cannot leave 0 .. 63 in 64 rows.)
                const u64 rows_m = ~0ull << base;
                const vi ds = w_sel(rows_m, 0, -(lane & 1) - step_dn);
                const vi s_sh = ((vi)w_prefix_add((vu)ds) - ds) + (31 - j_top);
                vu sa, sb;
                w_shift64(ra, rb, s_sh, sa, sb);
                const int i_end = w_chain2(sa, sb, 31, base, my_lane);
                my_lane = w_sel(rows_m, 0, my_lane - s_sh);
                // (the oldest row, lane 63: what the chain ends on is its lane plus its bit, c[63] = -1)
                const int j_end = i_end + 1 - w_readlane(s_sh, 63);
                kv = w_uni(w_readlane(kc, 63) + 2 * j_end - 1);
            } else {
                vi vj = j_top;
                for (int l = base; l < ((A.debug & 2) ? 0 : 64); l++) {
                    my_lane = w_sel(1ull << l, my_lane, vj);
                    vu lo = (vu)w_readlaneu(ra, l), hi = (vu)w_readlaneu(rb, l);
                    if (w_ballot(vj >= 64)) {
                        // a wide row: cells 64.. and their bits are in the iterations behind it
                        const vu at = ((vu)w_readlaneu(itj, l) + ((vu)vj >> 6)) & ring_mask;
                        vu xc, xd;
                        w_load_x4(w.recs, at, lo, hi, xc, xd);
                    }
                    const vu word = w_selu(w_ballot((vj & 32) != 0), lo, hi);
                    const vi bit = (vi)((word >> ((vu)vj & 31u)) & 1u);
                    if (l == 63) {
                        kv = w_uni(w_readlane(kc, l) + 2 * w_readlane(vj, 0) + 2 * w_readlane(bit, 0) - 1);
                    } else {
                        vj = vj + (w_readlane(step_dn, l) + bit);
                    }
                }
            }
            // every row's own bit, cell and script word, all rows at once
            const u64 rows = ~0ull << base;
            const vi r = r_top - (lane - base);
            vi my_dir = 0;
            W_WHERE((A.debug & 4) ? 0ull : rows) {
                vu word = w_selu(w_ballot((my_lane & 32) != 0), ra, rb);
                const u64 far = w_ballot(my_lane >= 64);
                if (far) {
                    W_WHERE(far) {
                        vu xa, xb, xc, xd;
                        w_load_x4(w.recs, (itj + ((vu)my_lane >> 6)) & ring_mask, xa, xb, xc, xd);
                        word = w_selu(w_ballot((my_lane & 32) != 0), xa, xb);
                    }
                }
                my_dir = (vi)((word >> ((vu)my_lane & 31u)) & 1u);
                const vu itc = itj + ((vu)my_lane >> 6);  // (a wide row's cell 64 c + l: c iterations on)
                const vu cw = w_load32(w.cells, ((itc >> 2) & chunk_mask) * 64u + ((vu)my_lane & 63u));
                vu m = (cw >> ((itc & 3u) * 8u)) & 255u;
                const u64 long_one = w_ballot(m == 255u);
                if (long_one) {
                    const vu want = (itc << 6) | ((vu)my_lane & 63u);
#ifdef A2_HOOK_ESC
                    A2_HOOK_ESC(t, w, long_one, want, itc, my_lane);
#endif
                    for (int e = 0; e < w.n_esc; e++) {
                        vu elo, ehi;
                        w_load64(w.esc, (vu)e, elo, ehi);
                        m = w_selu(w_ballot(ehi == want) & long_one, m, elo);
                    }
                }
                const vi dir = w_sel(w_ballot(r == 0), my_dir, 0);  // row 0 starts at (0, 0): no edit
                w_store32(script, (vu)r, (m << 1) | (vu)dir);
            }
            n_ins += w_popc(rows & w_ballot(my_dir == 0) & w_ballot(r > 0));
            r_top -= n_rows;
        }
        if (r_top < 0 || (int)(ih - t.it0) < 64) break;
    }
    return n_ins;
}

// the records of the current (incomplete) block of 64 iterations, and the cell word of the
// current (incomplete) group of 4, to the tape: a trace-back is about to read them
W_FN void a2_flush_partial(const A2Args &A, A2Wave &w, A2Lanes &wl) {
    const vi lane = w_lane();
    const int n = (int)(w.it & 63u);
    if (n > 0) {
        W_WHERE(w_lanes(0, n)) {
            w_store_x4(w.recs, (w.it - (u32)n + (vu)lane) & (A.ring - 1u), wl.rc_mlo, wl.rc_mhi, wl.rc_k0, wl.rc_k1);
        }
    }
    if (w.it & 3u) w_store32(w.cells, ((w.it >> 2) & ((A.ring >> 2) - 1u)) * 64u + (vu)lane, wl.vacc);
    w_fence_block();
}

// ---------------------------------------------------------------------------------------
// the row loop: runs until a row finishes a track, a band hull reaches a forbidden lane,
// or `budget` iterations are done
// ---------------------------------------------------------------------------------------

// A band hull reached the edge of its track's lanes and no track finished: lay the running
// tracks out again without leaving the row loop -- both side by side with the free lanes
// shared out afresh (the boundary between them moves), or the single running track back to
// the middle of the wave.  A few dozen instructions; what it cannot do (the two bands no
// longer fit together: one has to be parked) is a2_place's business.  Returns true when the
// rows can go on.
struct A2TrackConst {  // what the lanes of a track hold of it
    int q_len0, t_len0, q_len1, t_len1;
    u32 qb0, tb0, qb1, tb1;
};
struct A2Regs;
// (taken out of a2_fast's packed state where it is needed: kept in registers across the row stream, the eight
// values cost the function vector registers it does not have)
W_FN A2TrackConst a2_tc(vu sc);
// the lanes no band hull may reach (PAIR: `split` = first lane of track 1's share; the two
// lanes around the boundary stay free, so the two bands are never neighbours)
template <bool PAIR>
W_FN void a2_bounds(A2Hot &h) {
    if (PAIR) {
        h.zone1 = ~0ull << h.split;
        const u64 fence = 3ull << (h.split - 1);
        h.forbid_to1 = fence | 1ull;            // before an odd row a band grows one lane down
        h.forbid_to0 = fence | (1ull << 63);    // before an even row one lane up
    } else {
        h.zone1 = 0ull;
        h.forbid_to1 = h.forbid_to0 = 1ull | (1ull << 63);  // (the band then never spans all 64 lanes)
    }
}

template <bool PAIR>
W_FN bool a2_replace(A2Hot &h, A2HotV &hv, vu &rc_k0, vu &rc_k1, const A2TrackConst &tc) {
    if (h.fin | h.big) return false;
#ifdef A2_HOOK_REPLACE
    A2_HOOK_REPLACE(PAIR, h);
#endif
    const vi lane = w_lane();
    const u64 ahead = ~0ull << (h.it & 63u);  // the tape from this iteration on
    const int down = (int)(h.it & 1u);        // the next row is an odd one: its band starts a lane below the hull
    if (PAIR) {
        const u64 in0 = h.in & ~h.zone1, in1 = h.in & h.zone1;
        const int hull0 = w_span(in0), hull1 = w_span(in1), n0 = hull0 + 1, n1 = hull1 + 1;
        const int free_lanes = 64 - n0 - n1;
        if (free_lanes < A2_FREE_MIN) return false;
        const int g0 = free_lanes / 4, mid = free_lanes / 2;
        const int nl0 = g0, nl1 = g0 + n0 + mid;
        const int sh0 = w_lowest(in0) - down - nl0, sh1 = w_lowest(in1) - down - nl1;
        h.split = g0 + n0 + mid / 2;
        a2_bounds<true>(h);
        const vi vsh = w_sel(h.zone1, sh0, sh1);
        h.in = w_lanes(nl0 + down, hull0) | w_lanes(nl1 + down, hull1);
        hv.vx = w_sel(h.in, A2_NEG, w_gather_lanes(hv.vx, (lane + vsh) & 63));  // new lane l: what lane l + sh held
        h.kb0 += (u32)(2 * sh0); h.kb1 += (u32)(2 * sh1);  // (the diagonals go with the rows)
        hv.vnegk = w_sel(h.zone1, 1 - (int)h.kb0, 1 - (int)h.kb1) - 2 * lane;
        hv.vqlen = w_sel(h.zone1, tc.q_len0, tc.q_len1);
        hv.vtlen = w_sel(h.zone1, tc.t_len0, tc.t_len1);
        hv.vqb = w_selu(h.zone1, tc.qb0, tc.qb1);
        hv.vtb = w_selu(h.zone1, tc.tb0, tc.tb1);
        hv.vtop = w_selu(h.zone1, 0u, A2_TOP);
        rc_k0 = w_selu(ahead, rc_k0, h.kb0);
        rc_k1 = w_selu(ahead, rc_k1, h.kb1);
        h.act = w_lanes(nl0, n0) | w_lanes(nl1, n1);
    } else {
        const int hull0 = w_span(h.in), n0 = hull0 + 1;
        if (n0 > A2_MAX_N) return false;
        const int nl0 = (64 - n0) / 2;
        const int sh0 = w_lowest(h.in) - down - nl0;
        h.in = w_lanes(nl0 + down, hull0);
        hv.vx = w_sel(h.in, A2_NEG, w_gather_lanes(hv.vx, (lane + sh0) & 63));
        hv.vnegk = hv.vnegk - 2 * sh0;
        if (h.kb0 != A2_INVALID) { h.kb0 += (u32)(2 * sh0); rc_k0 = w_selu(ahead, rc_k0, h.kb0); }
        else                     { h.kb1 += (u32)(2 * sh0); rc_k1 = w_selu(ahead, rc_k1, h.kb1); }
        h.act = w_lanes(nl0, n0);
    }
    h.ev = 0ull;
    h.n_replace++;
    return true;
}

// ---------------------------------------------------------------------------------------
// The row loop proper, OUT OF LINE: inlined into the wavefront's event loop, hipcc folds it
// into that loop's state machine and every row pays for the copies at its joins (k_align.hip
// met the same).  Its state travels by value: the per-lane registers as they are, the
// wave-uniform ones in the lanes of one more register (a device function takes and returns
// everything in VGPRs).  Runs rows -- pairs of an even and an odd iteration -- until a row
// raises an event or iteration `it_end` is reached.
// ---------------------------------------------------------------------------------------
struct A2Regs {
    vi vx, vnegk, vqlen, vtlen;
    vu vqb, vtb, vtop, vacc, rc_mlo, rc_mhi, rc_k0, rc_k1;
    vi vpark;  // the last row of a parked track, in the lanes it had
    vu sc;  // lane i: the i-th wave-uniform value (A2_SC_*)
};
enum { A2_SC_ACT_LO = 0, A2_SC_ACT_HI, A2_SC_IN_LO, A2_SC_IN_HI, A2_SC_BEST0, A2_SC_BEST1,
       A2_SC_CELLS0, A2_SC_CELLS1, A2_SC_SPLIT, A2_SC_IT, A2_SC_NESC, A2_SC_KB0, A2_SC_KB1, A2_SC_IT_END,
       A2_SC_FIN_LO, A2_SC_FIN_HI, A2_SC_EV_LO, A2_SC_EV_HI, A2_SC_ROW_LO, A2_SC_ROW_HI,
       A2_SC_WORDS_LO, A2_SC_WORDS_HI, A2_SC_CELLS_LO, A2_SC_CELLS_HI, A2_SC_RECS_LO, A2_SC_RECS_HI,
       A2_SC_ESC_LO, A2_SC_ESC_HI, A2_SC_RING, A2_SC_BAND,
       A2_SC_QLEN0, A2_SC_TLEN0, A2_SC_QLEN1, A2_SC_TLEN1, A2_SC_QB0, A2_SC_TB0, A2_SC_QB1, A2_SC_TB1,
       A2_SC_IT_LAST, A2_SC_JOIN_AT, A2_SC_NREPLACE,
       // parking and joining inside the row loop's function (a2_park / a2_join):
       A2_SC_PARKED,      // 0: nobody; 1 / 2: track 0 / 1 waits in vpark with the state below
       A2_SC_P_KC, A2_SC_P_LI, A2_SC_P_HIN, A2_SC_P_BEST, A2_SC_P_CELLS,
       A2_SC_D0, A2_SC_D1,          // rows the tracks computed since the caller handed them over
       A2_SC_REM0, A2_SC_REM1,      // rows they had left then (max_d - d)
       A2_SC_TAPE_LAST,             // the iteration the tape allows the wave to reach
       A2_SC_ITS,                   // the iteration the current stretch began at
       A2_SC_AGAIN,                 // 1: the tracks were laid out anew -- call the function of MODE again
       A2_SC_MODE,                  // 1: both tracks run, 0: one
       A2_SC_IT_PAIR, A2_SC_IT_SINGLE, A2_SC_NPARK, A2_SC_NJOIN,
       // where the LDS windows of the two tracks stand (the row stream's own: k_align2_rows.h A2R_CONST_PAIR)
       A2_SC_CQ0, A2_SC_CT0, A2_SC_CQ1, A2_SC_CT1 };
static_assert(A2_SC_CT1 < 64, "the wave-uniform state must fit the lanes of one register");

W_FN A2TrackConst a2_tc(vu sc) {
    A2TrackConst tc;
    tc.q_len0 = (int)w_pack_get<A2_SC_QLEN0>(sc); tc.t_len0 = (int)w_pack_get<A2_SC_TLEN0>(sc);
    tc.q_len1 = (int)w_pack_get<A2_SC_QLEN1>(sc); tc.t_len1 = (int)w_pack_get<A2_SC_TLEN1>(sc);
    tc.qb0 = w_pack_get<A2_SC_QB0>(sc); tc.tb0 = w_pack_get<A2_SC_TB0>(sc);
    tc.qb1 = w_pack_get<A2_SC_QB1>(sc); tc.tb1 = w_pack_get<A2_SC_TB1>(sc);
    return tc;
}

// ---------------------------------------------------------------------------------------
// Leaving and re-forming the pair WITHOUT going back to the wavefront's event loop (a trip
// through it is ~800 instructions; two thirds of all trips were these two transitions).  Both
// work on the row loop's own state, the way a2_replace does, and leave the result packed for
// the other instance of a2_fast: the caller (a2_drive_call) only looks at AGAIN / MODE.
//   a2_park: the two bands no longer fit the wave.  The narrower track's last row, hull,
//            lane-0 diagonal, best_m and cell count go into vpark / the P_* fields, the wider
//            one is centred and goes on alone (what a2_place decides and does).
//   a2_join: the running band has become narrow enough for the waiting track to fit beside
//            it with A2_FREE_JOIN lanes to spare: both are laid out side by side.
// Neither touches anything before it knows it will go through with it; whatever they decline
// (a band too wide to run alone, no rows or tape left, a waiting track without rows) is the
// event loop's business as before.
// ---------------------------------------------------------------------------------------
W_FN int a2_rows_left(const A2Regs &r, bool t0, bool t1, u32 it, u32 &it_last) {
    int rem = 0x7fffffff;
    if (t0) rem = min(rem, (int)w_pack_get<A2_SC_REM0>(r.sc) - (int)w_pack_get<A2_SC_D0>(r.sc));
    if (t1) rem = min(rem, (int)w_pack_get<A2_SC_REM1>(r.sc) - (int)w_pack_get<A2_SC_D1>(r.sc));
    const int tape = (int)(w_pack_get<A2_SC_TAPE_LAST>(r.sc) - it);
    rem = min(rem, tape);
    it_last = it + (u32)max(rem, 0);
    return rem;
}

W_FN bool a2_park(A2Hot &h, A2HotV &hv, vu &rc_k0, vu &rc_k1, const A2TrackConst &tc, A2Regs &r) {
    const u64 in0 = h.in & ~h.zone1, in1 = h.in & h.zone1;
    if (in0 == 0ull || in1 == 0ull) return false;
    const int n0 = w_span(in0) + 1, n1 = w_span(in1) + 1;
    const bool park0 = n1 > n0;  // the wider one goes on alone (a2_place)
    if ((park0 ? n1 : n0) > A2_MAX_N) return false;
    u32 it_last;
    // (a track that has used up its rows is the event loop's to retire, not to be parked)
    if (a2_rows_left(r, true, true, h.it, it_last) <= 0) return false;
    (void)a2_rows_left(r, !park0, park0, h.it, it_last);
    const vi lane = w_lane();
    const int p_last = (int)(h.it & 1u) ^ 1;
    const u64 inp = park0 ? in0 : in1;
    const int p_li = w_lowest(inp), p_hin = w_highest(inp);
    w_pack_put<A2_SC_P_KC>(r.sc, (u32)((int)(park0 ? h.kb0 : h.kb1) - 1 + p_last));
    w_pack_put<A2_SC_P_LI>(r.sc, (u32)p_li);
    w_pack_put<A2_SC_P_HIN>(r.sc, (u32)p_hin);
    w_pack_put<A2_SC_P_BEST>(r.sc, park0 ? (u32)h.best0 : (u32)h.best1 - A2_TOP);
    w_pack_put<A2_SC_P_CELLS>(r.sc, park0 ? h.cells0 : h.cells1);
    w_pack_put<A2_SC_PARKED>(r.sc, park0 ? 1u : 2u);
    r.vpark = hv.vx;
    const u64 ahead = ~0ull << (h.it & 63u);
    // the runner plays "track 0" of the single loop, whichever it is
    if (park0) {
        h.best0 = (int)((u32)h.best1 - A2_TOP);
        h.cells0 = h.cells1;
        h.kb0 = A2_INVALID;
        rc_k0 = w_selu(ahead, rc_k0, A2_INVALID);
        hv.vqlen = tc.q_len1; hv.vtlen = tc.t_len1; hv.vqb = tc.qb1; hv.vtb = tc.tb1;
        hv.vnegk = (1 - (int)h.kb1) - 2 * lane;
    } else {
        h.kb1 = A2_INVALID;
        rc_k1 = w_selu(ahead, rc_k1, A2_INVALID);
        hv.vqlen = tc.q_len0; hv.vtlen = tc.t_len0; hv.vqb = tc.qb0; hv.vtb = tc.tb0;
        hv.vnegk = (1 - (int)h.kb0) - 2 * lane;
    }
    h.best1 = 0; h.cells1 = 0u;
    hv.vtop = 0u;
    h.in = park0 ? in1 : in0;
    h.split = park0 ? 0 : 64;
    h.zone1 = 0ull;
    h.ev = 0ull;
    (void)a2_replace<false>(h, hv, rc_k0, rc_k1, tc);  // centres the runner (its width was looked at above)
    const int join_at = max(1, 64 - A2_FREE_JOIN - (p_hin - p_li + 2));
    w_pack_put<A2_SC_IT_LAST>(r.sc, it_last);
    w_pack_put<A2_SC_IT_END>(r.sc, ((int)(it_last - h.it) > A2_LOOK_EVERY) ? h.it + A2_LOOK_EVERY : it_last);
    w_pack_put<A2_SC_JOIN_AT>(r.sc, (u32)join_at);
    w_pack_put<A2_SC_MODE>(r.sc, 0u);
    w_pack_put<A2_SC_AGAIN>(r.sc, 1u);
    w_pack_put<A2_SC_NPARK>(r.sc, w_pack_get<A2_SC_NPARK>(r.sc) + 1u);
#ifdef A2_HOOK_DRIVE
    A2_HOOK_DRIVE(0);
#endif
    return true;
}

W_FN bool a2_join(A2Hot &h, A2HotV &hv, vu &rc_k0, vu &rc_k1, const A2TrackConst &tc, A2Regs &r) {
    const u32 parked = w_pack_get<A2_SC_PARKED>(r.sc);
    if (parked == 0u || h.in == 0ull) return false;
    const bool run_is0 = parked == 2u;  // track 1 waits: the runner is track 0
    const int p_li = (int)w_pack_get<A2_SC_P_LI>(r.sc), p_hin = (int)w_pack_get<A2_SC_P_HIN>(r.sc);
    const int nr = w_span(h.in) + 1, np = p_hin - p_li + 2;
    const int free_lanes = 64 - nr - np;
    if (free_lanes < A2_FREE_JOIN) return false;
    u32 it_last;
    if (a2_rows_left(r, true, true, h.it, it_last) <= 0) return false;
    const vi lane = w_lane();
    const int down = (int)(h.it & 1u);  // the next row is an odd one: its band starts a lane below the hull
    const int n0 = run_is0 ? nr : np, n1 = run_is0 ? np : nr;
    const int g0 = free_lanes / 4, mid = free_lanes / 2;
    const int nl0 = g0, nl1 = g0 + n0 + mid;
    h.split = g0 + n0 + mid / 2;
    a2_bounds<true>(h);
    const int sh_r = w_lowest(h.in) - down - (run_is0 ? nl0 : nl1);
    const int sh_p = p_li - down - (run_is0 ? nl1 : nl0);
    const vi vr = w_gather_lanes(hv.vx, (lane + sh_r) & 63);
    const vi vp = w_gather_lanes(r.vpark, (lane + sh_p) & 63);
    h.in = w_lanes(nl0 + down, n0 - 1) | w_lanes(nl1 + down, n1 - 1);
    hv.vx = w_sel(h.in, A2_NEG, w_sel(h.zone1, run_is0 ? vr : vp, run_is0 ? vp : vr));
    const u32 kb_r = (run_is0 ? h.kb0 : h.kb1) + (u32)(2 * sh_r);
    const u32 kb_p = (u32)((int)w_pack_get<A2_SC_P_KC>(r.sc) + 2 * sh_p + down);
    h.kb0 = run_is0 ? kb_r : kb_p;
    h.kb1 = run_is0 ? kb_p : kb_r;
    const u32 best_r = (u32)h.best0, best_p = w_pack_get<A2_SC_P_BEST>(r.sc);  // (the runner played track 0)
    const u32 cells_r = h.cells0, cells_p = w_pack_get<A2_SC_P_CELLS>(r.sc);
    h.best0 = (int)(run_is0 ? best_r : best_p);
    h.best1 = (int)((run_is0 ? best_p : best_r) + A2_TOP);
    h.cells0 = run_is0 ? cells_r : cells_p;
    h.cells1 = run_is0 ? cells_p : cells_r;
    hv.vnegk = w_sel(h.zone1, 1 - (int)h.kb0, 1 - (int)h.kb1) - 2 * lane;
    hv.vqlen = w_sel(h.zone1, tc.q_len0, tc.q_len1);
    hv.vtlen = w_sel(h.zone1, tc.t_len0, tc.t_len1);
    hv.vqb = w_selu(h.zone1, tc.qb0, tc.qb1);
    hv.vtb = w_selu(h.zone1, tc.tb0, tc.tb1);
    hv.vtop = w_selu(h.zone1, 0u, A2_TOP);
    const u64 ahead = ~0ull << (h.it & 63u);
    rc_k0 = w_selu(ahead, rc_k0, h.kb0);
    rc_k1 = w_selu(ahead, rc_k1, h.kb1);
    h.act = w_lanes(nl0, n0) | w_lanes(nl1, n1);
    h.ev = 0ull;
    w_pack_put<A2_SC_PARKED>(r.sc, 0u);
    w_pack_put<A2_SC_IT_LAST>(r.sc, it_last);
    w_pack_put<A2_SC_IT_END>(r.sc, it_last);
    w_pack_put<A2_SC_JOIN_AT>(r.sc, 0u);
    w_pack_put<A2_SC_MODE>(r.sc, 1u);
    w_pack_put<A2_SC_AGAIN>(r.sc, 1u);
    w_pack_put<A2_SC_NJOIN>(r.sc, w_pack_get<A2_SC_NJOIN>(r.sc) + 1u);
#ifdef A2_HOOK_DRIVE
    A2_HOOK_DRIVE(1);
#endif
    return true;
}

// ---------------------------------------------------------------------------------------
// The rows between two events: at least one, until a row raises an event (h.ev != 0, h.act_row =
// the lanes that held a cell in it) or iteration `it_end` is reached.  Every fourth iteration
// the cell bytes go to the tape, every 64th the records.  This is the statement of what the
// row loop does; the product runs it as one hand-scheduled instruction stream
// (k_align2_rows.h: W_ROWS_ASM), the lane emulator runs this.
// ---------------------------------------------------------------------------------------
template <bool PAIR>
W_FN void a2_rows_c(A2Hot &h, A2HotV &hv, vu &rc_k0, vu &rc_k1, const u32 *words, u64 *esc, u32 *cells, u32 *recs,
                    u32 ring, int band, u32 it_end) {
    const vi lane = w_lane();
    // the cell byte of an iteration is byte it & 3 of the lane's word: v_perm selectors that
    // put it there, alternating between (0, 1) and (2, 3) with every pair of rows
    u32 sel_even = (h.it & 2u) ? 0x03040100u : 0x03020104u;
    u32 sel_odd = (h.it & 2u) ? 0x04020100u : 0x03020400u;
    for (;;) {
        if ((h.it & 1u) == 0u) {  // (a stretch may begin at an odd iteration: then with the odd row)
            const u64 lanes_ = a2_row<0, -1, PAIR>(h, hv, words, esc, band, sel_even);
            if (h.ev) { h.act_row = lanes_; return; }
            if (h.it == it_end) return;
        }
        const u64 lanes_ = a2_row<1, -1, PAIR>(h, hv, words, esc, band, sel_odd);
        if ((h.it & 3u) == 0u)
            w_store32(cells, (((h.it >> 2) - 1u) & ((ring >> 2) - 1u)) * 64u + (vu)lane, hv.vacc);
        if ((h.it & 63u) == 0u) {
            w_store_x4(recs, ((h.it - 64u) & (ring - 1u)) + (vu)lane, hv.rc_mlo, hv.rc_mhi, rc_k0, rc_k1);
            rc_k0 = h.kb0;
            rc_k1 = h.kb1;
        }
        sel_even ^= 0x03040100u ^ 0x03020104u;
        sel_odd ^= 0x04020100u ^ 0x03020400u;
        if (h.ev) { h.act_row = lanes_; return; }
        if (h.it == it_end) return;
    }
}

#if defined(W_ROWS_ASM) && !defined(A2_ROWS_C)
#include "k_align2_rows.h"
#else
// (without the hand-written stream the rows count their cells themselves: nothing to fold)
struct A2RowsV { vi tdn, tup; vu vcnt; };
template <bool PAIR>
W_FN void a2_fold_cells(A2Hot &, vu &) {}
W_FN void a2_rows_begin(A2RowsV &rv) { rv.tdn = A2_NEG; rv.tup = A2_NEG; rv.vcnt = 0u; }
W_FN void a2_rows_done(A2RowsV &) {}
#endif

#if defined(A2_SHADOW)
// one log entry: what differed (bit per field), where, and both values of the first scalar that did
W_FN void a2_shadow_log2(u32 what, u32 it_in, u32 it_c, u32 it_a, const u64 *v, u32 xc, u32 xa, u32 flags, u32 cc,
                         u32 ca);
W_FN bool a2_shadow_adopt();
template <bool PAIR>
W_FN void a2_rows_shadow(A2Hot &h, A2HotV &hv, vu &rc_k0, vu &rc_k1, A2RowsV &rv, const u32 *words, u64 *esc,
                         u32 *cells, u32 *recs, u32 ring, int band, u32 it_end, bool head, vu &sc) {
    const A2TrackConst tc = a2_tc(sc);
    // Both renderings run what a2_fast's loop runs -- rows, and the bands laid out again whenever a row's event
    // allows it -- until an event nobody in there resolves, or `it_end`: the stream lays the bands out again
    // inside the asm statement (round 6), the statement through a2_replace.
    A2Hot hc = h;
    A2HotV hvc = hv;
    vu k0c = rc_k0, k1c = rc_k1;
    const u32 it_in = h.it;
    for (;;) {
        a2_rows_c<PAIR>(hc, hvc, k0c, k1c, words, esc, cells, recs, ring, band, it_end);
        if (hc.ev && !a2_replace<PAIR>(hc, hvc, k0c, k1c, tc)) break;
        if (hc.it == it_end) break;
    }
    A2Hot ha = h;
    A2HotV hva = hv;
    vu k0a = rc_k0, k1a = rc_k1;
    A2RowsV rva = rv;
    rva.vcnt = 0u;
    for (bool first = head;; first = false) {
        // (a track's row 0 comes back alone, and the rows stop to have a window filled again: go on)
        a2_rows_asm<PAIR>(ha, hva, k0a, k1a, rva, words, cells, recs, ring, band, it_end, first, sc, it_end, 0);
        if (ha.ev) {
            if (PAIR) a2_fold_cells<true>(ha, rva.vcnt);
            if (!a2_replace<PAIR>(ha, hva, k0a, k1a, tc)) break;
        }
        if (ha.it == it_end) break;
    }
    a2_fold_cells<PAIR>(ha, rva.vcnt);
    u32 what = 0;
    if (ha.it != hc.it) what |= 1u;
    if (ha.act != hc.act) what |= 2u;
    if (ha.in != hc.in) what |= 4u;
    if (ha.fin != hc.fin) what |= 8u;
    if (ha.big != hc.big) what |= 16u;
    if ((ha.ev != 0ull) != (hc.ev != 0ull)) what |= 32u;
    if (ha.best0 != hc.best0) what |= 64u;
    if (PAIR && ha.best1 != hc.best1) what |= 128u;
    if (hc.ev && ha.act_row != hc.act_row) what |= 256u;
    const u64 dx = w_ballot(hva.vx != hvc.vx);
    if (dx) what |= 512u;
    if (w_ballot(k0a != k0c) | w_ballot(k1a != k1c)) what |= 1024u;
    if (ha.cells0 != hc.cells0 || (PAIR && ha.cells1 != hc.cells1)) what |= 2048u;
    if (hc.big && (w_ballot(hva.vm != hvc.vm) & hc.big)) what |= 4096u;
    if (ha.split != hc.split || ha.kb0 != hc.kb0 || ha.kb1 != hc.kb1 || ha.zone1 != hc.zone1) what |= 8192u;
    if (ha.n_replace != hc.n_replace) what |= 16384u;
    if (what) {
        const int l = dx ? w_lowest(dx) : 0;
        const u64 v[8] = {hc.act, ha.act, hc.in, ha.in, dx, h.act, hc.fin, ha.fin};
        a2_shadow_log2(what, it_in, hc.it, ha.it, v, (u32)w_readlane(hvc.vx, l), (u32)w_readlane(hva.vx, l),
                       (PAIR ? 1u : 0u) | (head ? 2u : 0u) | ((u32)l << 8) | ((u32)(h.split & 0xff) << 16),
                       hc.cells0, ha.cells0);
    }
    if (a2_shadow_adopt()) {  // (FALCON_AMD_A2_SHADOW=2: go on with the stream's state, as the product does)
        h = ha; hv = hva; rc_k0 = k0a; rc_k1 = k1a;
    } else {
        h = hc; hv = hvc; rc_k0 = k0c; rc_k1 = k1c;
    }
    rv = rva;
    rv.vcnt = 0u;
}
#endif

template <bool PAIR>
W_FN void a2_rows(A2Hot &h, A2HotV &hv, vu &rc_k0, vu &rc_k1, A2RowsV &rv, const u32 *words, u64 *esc, u32 *cells,
                  u32 *recs, u32 ring, int band, u32 it_end, bool head, vu &sc, u32 it_last, int join_at) {
#if defined(A2_SHADOW)
    // (k_align2_shadow.hip, tests only: both renderings of the rows from the same state, every
    // difference logged, the statement's result kept)
    a2_rows_shadow<PAIR>(h, hv, rc_k0, rc_k1, rv, words, esc, cells, recs, ring, band, it_end, head, sc);
    (void)it_last; (void)join_at;
#elif defined(W_ROWS_ASM) && !defined(A2_ROWS_C)
    a2_rows_asm<PAIR>(h, hv, rc_k0, rc_k1, rv, words, cells, recs, ring, band, it_end, head, sc, it_last, join_at);
#else
    (void)sc; (void)it_last; (void)join_at;
    a2_rows_c<PAIR>(h, hv, rc_k0, rc_k1, words, esc, cells, recs, ring, band, it_end);
#endif
}

template <bool PAIR>
W_NOINLINE A2Regs a2_fast(A2Regs r) {
    A2HotV hv;
    hv.vx = r.vx; hv.vnegk = r.vnegk; hv.vqlen = r.vqlen; hv.vtlen = r.vtlen;
    hv.vqb = r.vqb; hv.vtb = r.vtb; hv.vtop = r.vtop; hv.vacc = r.vacc;
    hv.rc_mlo = r.rc_mlo; hv.rc_mhi = r.rc_mhi;
    vu rc_k0 = r.rc_k0, rc_k1 = r.rc_k1;
    A2Hot h;
    h.act = ((u64)w_pack_get<A2_SC_ACT_HI>(r.sc) << 32) | w_pack_get<A2_SC_ACT_LO>(r.sc);
    h.lo0 = h.hi0 = h.lo1 = h.hi1 = 0;
    h.best0 = (int)w_pack_get<A2_SC_BEST0>(r.sc); h.best1 = (int)w_pack_get<A2_SC_BEST1>(r.sc);
    h.cells0 = w_pack_get<A2_SC_CELLS0>(r.sc); h.cells1 = w_pack_get<A2_SC_CELLS1>(r.sc);
    h.split = (int)w_pack_get<A2_SC_SPLIT>(r.sc);
    h.it = w_pack_get<A2_SC_IT>(r.sc);
    h.n_esc = (int)w_pack_get<A2_SC_NESC>(r.sc);
    h.kb0 = w_pack_get<A2_SC_KB0>(r.sc); h.kb1 = w_pack_get<A2_SC_KB1>(r.sc);
    u32 it_end = w_pack_get<A2_SC_IT_END>(r.sc);
    const u32 it_last = w_pack_get<A2_SC_IT_LAST>(r.sc);
    const int join_at = (int)w_pack_get<A2_SC_JOIN_AT>(r.sc);
    h.n_replace = w_pack_get<A2_SC_NREPLACE>(r.sc);
    const u32 *words = (const u32 *)(((u64)w_pack_get<A2_SC_WORDS_HI>(r.sc) << 32) | w_pack_get<A2_SC_WORDS_LO>(r.sc));
    u32 *cells = (u32 *)(((u64)w_pack_get<A2_SC_CELLS_HI>(r.sc) << 32) | w_pack_get<A2_SC_CELLS_LO>(r.sc));
    u32 *recs = (u32 *)(((u64)w_pack_get<A2_SC_RECS_HI>(r.sc) << 32) | w_pack_get<A2_SC_RECS_LO>(r.sc));
    u64 *esc = (u64 *)(((u64)w_pack_get<A2_SC_ESC_HI>(r.sc) << 32) | w_pack_get<A2_SC_ESC_LO>(r.sc));
    const u32 ring = w_pack_get<A2_SC_RING>(r.sc);
    const int band = (int)w_pack_get<A2_SC_BAND>(r.sc);
    a2_bounds<PAIR>(h);
    h.in = ((u64)w_pack_get<A2_SC_IN_HI>(r.sc) << 32) | w_pack_get<A2_SC_IN_LO>(r.sc);
    h.fin = 0ull; h.ev = 0ull; h.act_row = 0ull; h.big = 0ull;
    hv.vm = 0u;
    // (`it_end`: where to look up from the rows -- the end of the tracks' rows `it_last`, or,
    // with a neighbour parked, every A2_LOOK_EVERY iterations: when the running band has
    // become narrow enough (<= join_at lanes) the loop leaves, for the neighbour to join)
    A2RowsV rv;
    a2_rows_begin(rv);
    bool head = true;
#pragma clang loop unroll(disable)
    for (;;) {
        a2_rows<PAIR>(h, hv, rc_k0, rc_k1, rv, words, esc, cells, recs, ring, band, it_end, head, r.sc, it_last, join_at);
        head = false;
        // (the stream takes the look-ups a band is still too wide at in its stride: `it_end` follows)
        while (!PAIR && (int)(h.it - it_end) > 0)
            it_end = ((int)(it_last - it_end) > A2_LOOK_EVERY) ? it_end + A2_LOOK_EVERY : it_last;
        if (h.ev) {
            if (PAIR) a2_fold_cells<true>(h, rv.vcnt);  // (the boundary between the tracks is about to move)
            if (!a2_replace<PAIR>(h, hv, rc_k0, rc_k1, a2_tc(r.sc))) break;
        }
        if (h.it == it_end) {
            if (PAIR || it_end == it_last || w_span(h.in) + 1 <= join_at) break;
            it_end = ((int)(it_last - it_end) > A2_LOOK_EVERY) ? it_end + A2_LOOK_EVERY : it_last;
        }
    }
    a2_fold_cells<PAIR>(h, rv.vcnt);
    a2_rows_done(rv);
    const vi lane = w_lane();
    if (h.big) {
        // the last row had snakes of >= 255 bases: their lengths to the escape list, 255 into
        // their cell bytes (the row put the low byte of the length there) -- and into the
        // tape, if the row completed its group of four
        const u32 it_row = h.it - 1u;
        h.n_esc = w_uni(a2_escape<void>(esc, h.n_esc, it_row, hv.vm, (u32)h.big, (u32)(h.big >> 32)));
        const u32 where = (it_row & 2u) ? ((it_row & 1u) ? 0x04020100u : 0x03040100u)
                                        : ((it_row & 1u) ? 0x03020400u : 0x03020104u);
        hv.vacc = w_selu(h.big, hv.vacc, w_put_byte_sel(hv.vacc, (vu)255u, where));
        if ((h.it & 3u) == 0u)
            w_store32(cells, (((h.it >> 2) - 1u) & ((ring >> 2) - 1u)) * 64u + (vu)lane, hv.vacc);
        h.ev &= ~h.big;
    }
    {   // the rows of this stretch, per track
        const u32 dn = h.it - w_pack_get<A2_SC_ITS>(r.sc);
        if (PAIR || h.kb0 != A2_INVALID) w_pack_put<A2_SC_D0>(r.sc, w_pack_get<A2_SC_D0>(r.sc) + dn);
        if (PAIR || h.kb0 == A2_INVALID) w_pack_put<A2_SC_D1>(r.sc, w_pack_get<A2_SC_D1>(r.sc) + dn);
        if (PAIR) w_pack_put<A2_SC_IT_PAIR>(r.sc, w_pack_get<A2_SC_IT_PAIR>(r.sc) + dn);
        else      w_pack_put<A2_SC_IT_SINGLE>(r.sc, w_pack_get<A2_SC_IT_SINGLE>(r.sc) + dn);
        w_pack_put<A2_SC_ITS>(r.sc, h.it);
        w_pack_put<A2_SC_AGAIN>(r.sc, 0u);
        w_pack_put<A2_SC_MODE>(r.sc, PAIR ? 1u : 0u);
    }
    if (PAIR) {
        // the two bands no longer fit the wave (a2_replace declined): the narrower track waits
        if (h.ev != 0ull && h.fin == 0ull && h.big == 0ull && w_pack_get<A2_SC_PARKED>(r.sc) == 0u)
            (void)a2_park(h, hv, rc_k0, rc_k1, a2_tc(r.sc), r);
    } else {
        // the look-ahead found the running band narrow enough for the waiting track
        if (h.ev == 0ull && h.fin == 0ull && h.it != it_last && join_at > 0 && w_span(h.in) + 1 <= join_at)
            (void)a2_join(h, hv, rc_k0, rc_k1, a2_tc(r.sc), r);
    }
    r.vx = hv.vx; r.vnegk = hv.vnegk; r.vqlen = hv.vqlen; r.vtlen = hv.vtlen;
    r.vqb = hv.vqb; r.vtb = hv.vtb; r.vtop = hv.vtop;
    r.vacc = hv.vacc; r.rc_mlo = hv.rc_mlo; r.rc_mhi = hv.rc_mhi;
    r.rc_k0 = rc_k0; r.rc_k1 = rc_k1;
    w_pack_put<A2_SC_SPLIT>(r.sc, (u32)h.split);
    w_pack_put<A2_SC_KB0>(r.sc, h.kb0); w_pack_put<A2_SC_KB1>(r.sc, h.kb1);
    w_pack_put<A2_SC_NREPLACE>(r.sc, h.n_replace);
    w_pack_put<A2_SC_ACT_LO>(r.sc, (u32)h.act); w_pack_put<A2_SC_ACT_HI>(r.sc, (u32)(h.act >> 32));
    w_pack_put<A2_SC_IN_LO>(r.sc, (u32)h.in); w_pack_put<A2_SC_IN_HI>(r.sc, (u32)(h.in >> 32));
    w_pack_put<A2_SC_BEST0>(r.sc, (u32)h.best0); w_pack_put<A2_SC_BEST1>(r.sc, (u32)h.best1);
    w_pack_put<A2_SC_CELLS0>(r.sc, h.cells0); w_pack_put<A2_SC_CELLS1>(r.sc, h.cells1);
    w_pack_put<A2_SC_IT>(r.sc, h.it);
    w_pack_put<A2_SC_NESC>(r.sc, (u32)h.n_esc);
    w_pack_put<A2_SC_FIN_LO>(r.sc, (u32)h.fin); w_pack_put<A2_SC_FIN_HI>(r.sc, (u32)(h.fin >> 32));
    w_pack_put<A2_SC_EV_LO>(r.sc, (u32)h.ev); w_pack_put<A2_SC_EV_HI>(r.sc, (u32)(h.ev >> 32));
    w_pack_put<A2_SC_ROW_LO>(r.sc, (u32)h.act_row); w_pack_put<A2_SC_ROW_HI>(r.sc, (u32)(h.act_row >> 32));
    return r;
}

// What the event loop hands over besides the row loop's own state, and takes back
struct A2Drive {
    int pair;                  // in: the placement runs both tracks; out: the last stretch did
    int parked;                // 0, or 1 + the track that waits in vpark with rows of its own
    int p_kc, p_li, p_hin, p_best;
    u32 p_cells;
    int rem0, rem1;            // in: rows the tracks have left
    int tape_left;             // in: iterations the tape has left
    int d0, d1;                // out: rows the tracks computed
    u32 it_pair, it_single, n_park, n_join;  // out: statistics
};

// the call: marshal, run stretch after stretch until an event needs the event loop, take the
// state back.  (it_end / it_last / join_at of the first stretch come from the caller; the
// later ones are worked out by a2_park / a2_join.)
W_FN void a2_drive_call(const A2Args &A, A2Wave &w, A2Lanes &wl, A2Hot &h, A2HotV &hv, u32 it_end, u32 it_last,
                        int join_at, A2Drive &dr) {
    A2Regs r;
    r.vx = hv.vx; r.vnegk = hv.vnegk; r.vqlen = hv.vqlen; r.vtlen = hv.vtlen;
    r.vqb = hv.vqb; r.vtb = hv.vtb; r.vtop = hv.vtop; r.vacc = hv.vacc;
    r.rc_mlo = hv.rc_mlo; r.rc_mhi = hv.rc_mhi; r.rc_k0 = wl.rc_k0; r.rc_k1 = wl.rc_k1;
    r.vpark = wl.vpark;
    r.sc = w_undef();
    w_pack_put<A2_SC_ACT_LO>(r.sc, (u32)h.act); w_pack_put<A2_SC_ACT_HI>(r.sc, (u32)(h.act >> 32));
    w_pack_put<A2_SC_IN_LO>(r.sc, (u32)h.in); w_pack_put<A2_SC_IN_HI>(r.sc, (u32)(h.in >> 32));
    w_pack_put<A2_SC_BEST0>(r.sc, (u32)h.best0); w_pack_put<A2_SC_BEST1>(r.sc, (u32)h.best1);
    w_pack_put<A2_SC_CELLS0>(r.sc, h.cells0); w_pack_put<A2_SC_CELLS1>(r.sc, h.cells1);
    w_pack_put<A2_SC_SPLIT>(r.sc, (u32)h.split);
    w_pack_put<A2_SC_IT>(r.sc, h.it);
    w_pack_put<A2_SC_NESC>(r.sc, (u32)h.n_esc);
    w_pack_put<A2_SC_KB0>(r.sc, h.kb0); w_pack_put<A2_SC_KB1>(r.sc, h.kb1);
    w_pack_put<A2_SC_IT_END>(r.sc, it_end);
    w_pack_put<A2_SC_IT_LAST>(r.sc, it_last);
    w_pack_put<A2_SC_JOIN_AT>(r.sc, (u32)join_at);
    w_pack_put<A2_SC_QLEN0>(r.sc, (u32)w.T0.q_len); w_pack_put<A2_SC_TLEN0>(r.sc, (u32)w.T0.t_len);
    w_pack_put<A2_SC_QLEN1>(r.sc, (u32)w.T1.q_len); w_pack_put<A2_SC_TLEN1>(r.sc, (u32)w.T1.t_len);
    w_pack_put<A2_SC_QB0>(r.sc, w.T0.qb); w_pack_put<A2_SC_TB0>(r.sc, w.T0.tb);
    w_pack_put<A2_SC_QB1>(r.sc, w.T1.qb); w_pack_put<A2_SC_TB1>(r.sc, w.T1.tb);
    w_pack_put<A2_SC_WORDS_LO>(r.sc, (u32)(u64)A.words); w_pack_put<A2_SC_WORDS_HI>(r.sc, (u32)((u64)A.words >> 32));
    w_pack_put<A2_SC_CELLS_LO>(r.sc, (u32)(u64)w.cells); w_pack_put<A2_SC_CELLS_HI>(r.sc, (u32)((u64)w.cells >> 32));
    w_pack_put<A2_SC_RECS_LO>(r.sc, (u32)(u64)w.recs); w_pack_put<A2_SC_RECS_HI>(r.sc, (u32)((u64)w.recs >> 32));
    w_pack_put<A2_SC_ESC_LO>(r.sc, (u32)(u64)w.esc); w_pack_put<A2_SC_ESC_HI>(r.sc, (u32)((u64)w.esc >> 32));
    w_pack_put<A2_SC_RING>(r.sc, A.ring);
    w_pack_put<A2_SC_BAND>(r.sc, (u32)A.band);
    w_pack_put<A2_SC_PARKED>(r.sc, (u32)dr.parked);
    w_pack_put<A2_SC_P_KC>(r.sc, (u32)dr.p_kc); w_pack_put<A2_SC_P_LI>(r.sc, (u32)dr.p_li);
    w_pack_put<A2_SC_P_HIN>(r.sc, (u32)dr.p_hin); w_pack_put<A2_SC_P_BEST>(r.sc, (u32)dr.p_best);
    w_pack_put<A2_SC_P_CELLS>(r.sc, dr.p_cells);
    w_pack_put<A2_SC_D0>(r.sc, 0u); w_pack_put<A2_SC_D1>(r.sc, 0u);
    w_pack_put<A2_SC_REM0>(r.sc, (u32)dr.rem0); w_pack_put<A2_SC_REM1>(r.sc, (u32)dr.rem1);
    w_pack_put<A2_SC_TAPE_LAST>(r.sc, h.it + (u32)dr.tape_left);
    w_pack_put<A2_SC_ITS>(r.sc, h.it);
    w_pack_put<A2_SC_AGAIN>(r.sc, 0u);
    w_pack_put<A2_SC_MODE>(r.sc, dr.pair ? 1u : 0u);
    w_pack_put<A2_SC_IT_PAIR>(r.sc, 0u); w_pack_put<A2_SC_IT_SINGLE>(r.sc, 0u);
    w_pack_put<A2_SC_NPARK>(r.sc, 0u); w_pack_put<A2_SC_NJOIN>(r.sc, 0u);
    w_pack_put<A2_SC_NREPLACE>(r.sc, 0u);
    bool pair = dr.pair != 0;
    for (;;) {
        if (pair) r = a2_fast<true>(r); else r = a2_fast<false>(r);
        if (w_pack_get<A2_SC_AGAIN>(r.sc) == 0u) break;
        pair = w_pack_get<A2_SC_MODE>(r.sc) != 0u;
    }
    hv.vx = r.vx; hv.vnegk = r.vnegk; hv.vqlen = r.vqlen; hv.vtlen = r.vtlen;
    hv.vqb = r.vqb; hv.vtb = r.vtb; hv.vtop = r.vtop;
    hv.vacc = r.vacc; hv.rc_mlo = r.rc_mlo; hv.rc_mhi = r.rc_mhi;
    wl.rc_k0 = r.rc_k0; wl.rc_k1 = r.rc_k1;
    wl.vpark = r.vpark;
    h.split = (int)w_pack_get<A2_SC_SPLIT>(r.sc);
    w.split = h.split;
    h.kb0 = w_pack_get<A2_SC_KB0>(r.sc); h.kb1 = w_pack_get<A2_SC_KB1>(r.sc);
    w.st_recenter += w_pack_get<A2_SC_NREPLACE>(r.sc);
    h.act = ((u64)w_pack_get<A2_SC_ACT_HI>(r.sc) << 32) | w_pack_get<A2_SC_ACT_LO>(r.sc);
    h.in = ((u64)w_pack_get<A2_SC_IN_HI>(r.sc) << 32) | w_pack_get<A2_SC_IN_LO>(r.sc);
    h.best0 = (int)w_pack_get<A2_SC_BEST0>(r.sc); h.best1 = (int)w_pack_get<A2_SC_BEST1>(r.sc);
    h.cells0 = w_pack_get<A2_SC_CELLS0>(r.sc); h.cells1 = w_pack_get<A2_SC_CELLS1>(r.sc);
    h.it = w_pack_get<A2_SC_IT>(r.sc);
    h.n_esc = (int)w_pack_get<A2_SC_NESC>(r.sc);
    h.fin = ((u64)w_pack_get<A2_SC_FIN_HI>(r.sc) << 32) | w_pack_get<A2_SC_FIN_LO>(r.sc);
    h.ev = ((u64)w_pack_get<A2_SC_EV_HI>(r.sc) << 32) | w_pack_get<A2_SC_EV_LO>(r.sc);
    h.act_row = ((u64)w_pack_get<A2_SC_ROW_HI>(r.sc) << 32) | w_pack_get<A2_SC_ROW_LO>(r.sc);
    dr.pair = w_pack_get<A2_SC_MODE>(r.sc) != 0u ? 1 : 0;
    dr.parked = (int)w_pack_get<A2_SC_PARKED>(r.sc);
    dr.p_kc = (int)w_pack_get<A2_SC_P_KC>(r.sc); dr.p_li = (int)w_pack_get<A2_SC_P_LI>(r.sc);
    dr.p_hin = (int)w_pack_get<A2_SC_P_HIN>(r.sc); dr.p_best = (int)w_pack_get<A2_SC_P_BEST>(r.sc);
    dr.p_cells = w_pack_get<A2_SC_P_CELLS>(r.sc);
    dr.d0 = (int)w_pack_get<A2_SC_D0>(r.sc); dr.d1 = (int)w_pack_get<A2_SC_D1>(r.sc);
    dr.it_pair = w_pack_get<A2_SC_IT_PAIR>(r.sc); dr.it_single = w_pack_get<A2_SC_IT_SINGLE>(r.sc);
    dr.n_park = w_pack_get<A2_SC_NPARK>(r.sc); dr.n_join = w_pack_get<A2_SC_NJOIN>(r.sc);
}

W_FN u64 a2_zone(const A2Wave &w, int ti) {
    const u64 zone1 = w.split >= 64 ? 0ull : (w.split <= 0 ? ~0ull : (~0ull << w.split));
    return ti ? zone1 : ~zone1;
}

// a track whose row reached an end of one of its sequences: summary, trace-back, script
template <int TI>
W_FN void a2_finish(const A2Args &A, A2Wave &w, A2Lanes &wl, A2Track &t, u64 fin_t, u64 act_row) {
    // the finishing cell: the first such diagonal in ascending order (DW_banded.c:220-224)
    const int fl = w_lowest(fin_t);
    const int fin_x = w_readlane(wl.vx, fl);
    const int k_f = t.kc + 2 * fl;
    const int fin_y = fin_x - k_f;
    const int fin_d = t.d - 1;
    // cells evaluated: the reference stops its row at the finishing cell
    const u64 above = fl >= 63 ? 0ull : (~0ull << (fl + 1));
    const long long cells = (long long)t.cells - w_popc(act_row & a2_zone(w, TI) & above);
    a2_flush_partial(A, w, wl);
    const int n_ins = a2_trace<TI>(A, w, wl, t, w.it - 1u, k_f, fin_d);
    a2_result(A, t.g, 0, 1, fin_d, fin_x, fin_y, n_ins, cells);
    t.state = A2_IDLE;
}

// ---------------------------------------------------------------------------------------
// wide rows (61..191 diagonals; < 0.1 % of the rows, but a fifth of the alignments meet
// one): the track runs alone, one row at a time, 64 cells per pass, the previous row in an
// LDS ring indexed by diagonal -- the scheme of k_align.hip's ring mode.  On the tape a wide
// row is its first 64 cells as an ordinary iteration plus one continuation iteration
// (A2_CONT) per further 64 cells.
// ---------------------------------------------------------------------------------------
// one iteration's cells, bits and K fields onto the tape (any byte position)
W_FN void a2_emit(const A2Args &A, A2Wave &w, A2Lanes &wl, u64 fa, vu m, u32 k0, u32 k1) {
    const vi lane = w_lane();
    switch (w.it & 3u) {
        case 0: wl.vacc = w_put_byte<0>(wl.vacc, m); break;
        case 1: wl.vacc = w_put_byte<1>(wl.vacc, m); break;
        case 2: wl.vacc = w_put_byte<2>(wl.vacc, m); break;
        default: wl.vacc = w_put_byte<3>(wl.vacc, m); break;
    }
    const int slot = (int)(w.it & 63u);
    w_writelane2(wl.rc_mlo, wl.rc_mhi, (u32)fa, (u32)(fa >> 32), slot);
    w_writelane2(wl.rc_k0, wl.rc_k1, k0, k1, slot);
    w.it++;
    if ((w.it & 3u) == 0u)
        w_store32(w.cells, (((w.it >> 2) - 1u) & ((A.ring >> 2) - 1u)) * 64u + (vu)lane, wl.vacc);
    if ((w.it & 63u) == 0u)
        w_store_x4(w.recs, ((w.it - 64u) & (A.ring - 1u)) + (vu)lane, wl.rc_mlo, wl.rc_mhi, wl.rc_k0, wl.rc_k1);
}

// Runs track `t` (TI: its K field) through its wide rows, at most `row_budget` of them: with
// a neighbour waiting, a track that stays wide (reads that passed the window filter and then
// align badly widen the band for good) is handed back rather than holding the wave.  In: its last row in `from`
// (lane l: diagonal t.kc + 2 l, hull t.li .. t.hin).  Out: finished, dead (state A2_IDLE),
// handed back, or narrow again with its last row in wl.vx (state A2_RUN; the caller places it).
template <int TI>
W_FN void a2_wide(const A2Args &A, A2Wave &w, A2Lanes &wl, A2Track &t, vi from, int row_budget) {
    u32 *ring = w_lds() + 256;
    const vi lane = w_lane();
    int min_k = t.kc + 2 * t.li - 1, n = t.hin - t.li + 2;
    int d = t.d;
    {   // the last row into the ring of its parity
        const vi kp = t.kc + 2 * lane;
        w_lds_store(ring, (vu)(((d - 1) & 1) * A2_WIDE_RING) + ((vu)(kp >> 1) & (A2_WIDE_RING - 1u)), (vu)from);
        w_fence_block();
    }
    bool done = false, dead = false, back = false, back_tape = false;
    int fin_x = 0, fin_k = 0, fin_j = 0, n_chunk_fin = 0;
    u32 it_row = 0;
    while (!done && !dead && !back) {
        if (d >= t.max_d || n - 1 > A.band) { dead = true; break; }  // DW_banded.c:183-186
        if ((int)(w.it - t.it0) + 4 > (int)A.ring - 192) { back = back_tape = true; break; }
        if (n > 191 || row_budget-- <= 0) { back = true; break; }
        if (n <= A2_MAX_N - 12) break;
        const int par = d & 1, max_k = min_k + 2 * (n - 1);
        u32 *Vcur = ring + par * A2_WIDE_RING;
        const u32 *Vprev = ring + (par ^ 1) * A2_WIDE_RING;
        it_row = w.it;
        int best_row = -1;
        for (int c = 0; c * 64 < n && !done; c++) {
            const vi j = c * 64 + lane;
            const u64 act = w_ballot(j < n);
            const vi k = min_k + 2 * j;
            const vi a = (vi)w_lds_load(Vprev, (vu)((k - 1) >> 1) & (A2_WIDE_RING - 1u));
            const vi b = (vi)w_lds_load(Vprev, (vu)((k + 1) >> 1) & (A2_WIDE_RING - 1u));
            // from_above: k == min_k, or k != max_k and V[k-1] < V[k+1]   (:190)
            const u64 fa = (w_ballot(j == 0) | (w_ballot(a < b) & ~w_ballot(k == max_k))) & act;
            vi x = w_sel(fa, a + 1, b);
            vi y = x - k;
            vu m = 0u;
            W_WHERE(act) {
                // (the snake from its first base: "16 matched so far" makes the helper start)
                const A2Snake r = a2_snake_more<void>(A.words, (vu)t.qb, (vu)t.tb, (vi)t.q_len, (vi)t.t_len,
                                                      x, y, (vu)16u);
                m = r.m - 16u;
                x = r.x; y = r.y;
                w_lds_store(Vcur, (vu)(k >> 1) & (A2_WIDE_RING - 1u), (vu)x);
            }
            {
                const u64 big = w_ballot(m >= 255u) & act;
                if (big) {
                    w.n_esc = w_uni(a2_escape<void>(w.esc, w.n_esc, w.it, m, (u32)big, (u32)(big >> 32)));
                    m = w_minu(m, 255u);
                }
            }
            t.cells += (u32)w_popc(act);
            const u64 fin = (w_ballot(x >= t.q_len) | w_ballot(y >= t.t_len)) & act;  // :220
            const vi u = w_sel(act, -1, x + y);
            best_row = max(best_row, (int)w_readlaneu(w_prefix_max((vu)(u + 1)), 63) - 1);
            const u32 kfield = c == 0 ? (u32)(min_k + 1 - (int)(w.it & 1u)) : A2_CONT;
            a2_emit(A, w, wl, fa, m, TI == 0 ? kfield : A2_INVALID, TI == 0 ? A2_INVALID : kfield);
            if (fin) {  // the first finishing diagonal in ascending order
                const int fl = w_lowest(fin);
                fin_j = c * 64 + fl;
                fin_k = min_k + 2 * fin_j;
                fin_x = w_readlane(x, fl);
                t.cells -= (u32)w_popc(act & (fl >= 63 ? 0ull : (~0ull << (fl + 1))));
                n_chunk_fin = c + 1;
                done = true;
            }
        }
        d++;
        w.st_wide_rows++;
        if (done || w.n_esc > A2_ESC_CAP) break;
        w_fence_block();
        t.best = max(t.best, best_row);
        // band (:228-243) over the whole row
        int jlo = -1, jhi = -1;
        for (int c = 0; c * 64 < n; c++) {
            const vi j = c * 64 + lane;
            const vi k = min_k + 2 * j;
            const vi x = (vi)w_lds_load(Vcur, (vu)(k >> 1) & (A2_WIDE_RING - 1u));
            const u64 in = w_ballot(2 * x - k >= t.best - A.band) & w_ballot(j < n);
            if (in) {
                if (jlo < 0) jlo = c * 64 + w_lowest(in);
                jhi = c * 64 + w_highest(in);
            }
        }
        min_k = min_k + 2 * jlo - 1;  // (jlo >= 0: best_m is attained inside the row)
        n = jhi - jlo + 2;
    }
    t.d = d;
    if (w.n_esc > A2_ESC_CAP) return;  // (the caller hands everybody back)
    if (done) {
        (void)n_chunk_fin; (void)fin_j;
        a2_flush_partial(A, w, wl);
        const int n_ins = a2_trace<TI>(A, w, wl, t, it_row, fin_k, d - 1);
        a2_result(A, t.g, 0, 1, d - 1, fin_x, fin_x - fin_k, n_ins, (long long)t.cells);
        t.state = A2_IDLE;
        return;
    }
    if (dead) {
        a2_result(A, t.g, 0, 0, 0, 0, 0, 0, (long long)t.cells);
        t.state = A2_IDLE;
        return;
    }
    if (back) {
        a2_result(A, t.g, 2, 0, 0, 0, 0, 0, 0);
        t.state = A2_IDLE;
        w.st_bail++;
        if (back_tape) w.st_bail_tape++; else w.st_bail_wide++;
        return;
    }
    // narrow again: the last row back into registers, its hull around the middle of the wave
    {
        const int hull_n = n - 1;  // the hull's cells: diagonals min_k + 1, min_k + 3, ..
        const int li = max(2, (64 - hull_n) / 2);
        t.kc = (min_k + 1) - 2 * li;
        t.li = li;
        t.hin = li + hull_n - 1;
        const vi kp = t.kc + 2 * lane;
        wl.vx = w_sel(w_lanes(t.li, hull_n), A2_NEG,
                      (vi)w_lds_load(ring + ((d - 1) & 1) * A2_WIDE_RING, (vu)(kp >> 1) & (A2_WIDE_RING - 1u)));
        t.state = A2_RUN;
    }
}

// ---------------------------------------------------------------------------------------
// one wavefront: alignments from the work queue, two at a time, until the queue is empty
// ---------------------------------------------------------------------------------------
W_FN void a2_hand_back(const A2Args &A, A2Wave &w, A2Track &t) {
    a2_result(A, t.g, 2, 0, 0, 0, 0, 0, 0);  // err 2: the engine repeats it with the general kernel
    t.state = A2_IDLE;
    w.st_bail++;
}

W_FN void a2_wave(const A2Args &A, int slot) {
    A2Wave w;
    A2Lanes wl;
    w.T0.state = A2_IDLE; w.T1.state = A2_IDLE;
    w.T0.d = 0; w.T1.d = 0;
    w.T0.g = 0; w.T1.g = 0; w.T0.q_len = w.T0.t_len = w.T1.q_len = w.T1.t_len = 0;
    w.T0.max_d = w.T1.max_d = 0; w.T0.kc = w.T1.kc = 0; w.T0.li = w.T0.hin = w.T1.li = w.T1.hin = 0;
    w.T0.best = w.T1.best = -1; w.T0.it0 = w.T1.it0 = 0; w.T0.cells = w.T1.cells = 0;
    w.T0.qb = w.T0.tb = w.T1.qb = w.T1.tb = 0; w.T0.script_lo = w.T0.script_hi = w.T1.script_lo = w.T1.script_hi = 0;
    w.it = 0;
    w.pair = 0;
    w.split = 64;
    w.n_esc = 0;
    w.more = true;
    wl.vx = 0; wl.vpark = 0;
    wl.vacc = 0u; wl.rc_mlo = 0u; wl.rc_mhi = 0u; wl.rc_k0 = A2_INVALID; wl.rc_k1 = A2_INVALID;
    wl.vnegk = 0; wl.vqlen = 0; wl.vtlen = 0; wl.vqb = 0u; wl.vtb = 0u; wl.vtop = 0u;
    w.cells = A.cells + (u64)slot * A.slot_words;
    w.recs = w.cells + (u64)A.ring * 16u;
    w.esc = (u64 *)(w.recs + (u64)A.ring * 4u);
    w.st_pair = w.st_single = w.st_place = w.st_park = w.st_bail = w.st_tracks = w.st_wide = w.st_wide_rows = w.st_recenter = 0;
    w.st_bail_tape = w.st_bail_wide = w.st_bail_esc = 0;
    for (;;) {
        // ---- fill the free tracks
        if (w.T0.state == A2_IDLE && w.more) { w.more = a2_fetch(A, w.T0); if (w.more) { w.st_tracks++; a2_win_invalidate(0); } }
        if (w.T1.state == A2_IDLE && w.more) { w.more = a2_fetch(A, w.T1); if (w.more) { w.st_tracks++; a2_win_invalidate(1); } }
        if (w.T0.state == A2_IDLE && w.T1.state == A2_IDLE) break;
        // (escape entries of tracks long gone: start over when nobody can refer to them, and
        // drop the ones older than every track in flight when the list is half full -- they
        // are in tape order)
        if (w.n_esc > 0 && (w.T0.state == A2_IDLE || w.T0.d == 0) && (w.T1.state == A2_IDLE || w.T1.d == 0))
            w.n_esc = 0;
        if (w.n_esc > A.esc_cap / 2) {
            const bool live0 = w.T0.state != A2_IDLE && w.T0.d > 0, live1 = w.T1.state != A2_IDLE && w.T1.d > 0;
            u32 oldest = live0 ? w.T0.it0 : w.T1.it0;
            if (live0 && live1 && (int)(w.T1.it0 - w.T0.it0) < 0) oldest = w.T1.it0;
            const vi lane = w_lane();
            int kept = 0;
            for (int base = 0; base < w.n_esc; base += 64) {
                const u64 have = w_ballot(base + lane < w.n_esc);
                vu elo = 0u, ehi = 0u;
                W_WHERE(have) { w_load64(w.esc, (vu)(base + lane), elo, ehi); }
                // (26 bits of the iteration are kept in an entry: compare modulo 2^26)
                const u64 alive = have & w_ballot((vi)(((ehi >> 6) - oldest) << 6) >= 0);
                const vu to = (vu)kept + (vu)w_rank_in(alive);
                w_fence_block();
                W_WHERE(alive) { w_store64(w.esc, to, elo, ehi); }
                w_fence_block();
                kept += w_popc(alive);
            }
            w.n_esc = kept;
        }
        // ---- the tape must keep every row of the tracks in flight: whoever has been on it
        // for nearly a whole ring (parked for long while its neighbour ran, or longer than
        // the ring to begin with) is handed back; the rows below never run past what is left
        const int span0 = (w.T0.state != A2_IDLE && w.T0.d > 0) ? (int)(w.it - w.T0.it0) : 0;
        const int span1 = (w.T1.state != A2_IDLE && w.T1.d > 0) ? (int)(w.it - w.T1.it0) : 0;
        const int tape_left = (int)A.ring - 192 - max(span0, span1);
        if (tape_left < 64) {
            if (span0 >= span1) a2_hand_back(A, w, w.T0); else a2_hand_back(A, w, w.T1);
            w.st_bail_tape++;
            continue;
        }
#ifdef A2_HOOK_TRIP
        A2_HOOK_TRIP();
#endif
        const int rc = a2_place(w, wl);
        if (rc) {
            // a band too wide for the lanes: that track goes through its wide rows alone (its
            // neighbour, if it was running, waits in vpark)
            if (rc == 1) {
                const vi from = w.T0.state == A2_PARKED ? wl.vpark : wl.vx;
                if (w.T1.state == A2_RUN) { wl.vpark = wl.vx; w.T1.state = A2_PARKED; w.st_park++; }
                a2_wide<0>(A, w, wl, w.T0, from, w.T1.state == A2_IDLE ? 0x7fffffff : A.wide_patience);
            } else {
                const vi from = w.T1.state == A2_PARKED ? wl.vpark : wl.vx;
                if (w.T0.state == A2_RUN) { wl.vpark = wl.vx; w.T0.state = A2_PARKED; w.st_park++; }
                a2_wide<1>(A, w, wl, w.T1, from, w.T0.state == A2_IDLE ? 0x7fffffff : A.wide_patience);
            }
#ifdef A2_HOOK_WIDE
            A2_HOOK_WIDE(w, wl, rc);
#endif
            w.st_wide++;
            w.pair = 0;
            a2_win_invalidate(0);
            a2_win_invalidate(1);
            if (w.n_esc > A.esc_cap) {
                if (w.T0.state != A2_IDLE) { a2_hand_back(A, w, w.T0); w.st_bail_esc++; }
                if (w.T1.state != A2_IDLE) { a2_hand_back(A, w, w.T1); w.st_bail_esc++; }
                w.n_esc = 0;
            }
            continue;
        }
        // ---- rows until something happens
        const bool run0 = w.T0.state == A2_RUN, run1 = w.T1.state == A2_RUN;
        const int p = (int)(w.it & 1u);
        A2Hot h;
        A2HotV hv;
        hv.vx = wl.vx; hv.vnegk = wl.vnegk; hv.vqb = wl.vqb; hv.vtb = wl.vtb; hv.vqlen = wl.vqlen; hv.vtlen = wl.vtlen;
        hv.vtop = wl.vtop; hv.vacc = wl.vacc; hv.rc_mlo = wl.rc_mlo; hv.rc_mhi = wl.rc_mhi;
        h.split = w.split;
        h.zone1 = a2_zone(w, 1);
        h.forbid_to0 = h.forbid_to1 = 0ull;  // (a2_fast works them out of `split`)
        h.n_replace = 0u;
        h.it = w.it;
        h.n_esc = w.n_esc;
        h.fin = 0ull; h.ev = 0ull; h.act_row = 0ull;
        h.kb0 = run0 ? (u32)(w.T0.kc + p) : A2_INVALID;
        h.kb1 = run1 ? (u32)(w.T1.kc + p) : A2_INVALID;
        int budget, join_at = 0;
        if (w.pair) {
            a2_next_band(w.T0, p, h.lo0, h.hi0);
            a2_next_band(w.T1, p, h.lo1, h.hi1);
            if (w.T0.d == 0) h.lo0 = h.hi0 = -(w.T0.kc + (p ? 1 : -1)) / 2;  // diagonal 0's lane
            if (w.T1.d == 0) h.lo1 = h.hi1 = -(w.T1.kc + (p ? 1 : -1)) / 2;
            h.best0 = w.T0.best;
            h.best1 = (int)((u32)w.T1.best + A2_TOP);
            h.cells0 = w.T0.cells;
            h.cells1 = w.T1.cells;
            h.act = w_lanes(h.lo0, h.hi0 - h.lo0 + 1) | w_lanes(h.lo1, h.hi1 - h.lo1 + 1);
            h.in = 0ull;
            budget = min(min(w.T0.max_d - w.T0.d, w.T1.max_d - w.T1.d), tape_left);
        } else {
            // the running track plays "track 0" of the row loop, whichever it is.  (Its fields
            // are picked value by value: a reference chosen at run time would force both
            // track records into scratch memory, and everything read back from there counts
            // as divergent.)
            const int t_d = run0 ? w.T0.d : w.T1.d, t_kc = run0 ? w.T0.kc : w.T1.kc;
            const int t_li = run0 ? w.T0.li : w.T1.li, t_hin = run0 ? w.T0.hin : w.T1.hin;
            if (t_d == 0) {
                h.lo0 = h.hi0 = -(t_kc + (p ? 1 : -1)) / 2;
            } else if (p == 0) {
                h.lo0 = t_li; h.hi0 = t_hin + 1;
            } else {
                h.lo0 = t_li - 1; h.hi0 = t_hin;
            }
            h.lo1 = h.hi1 = 0;
            h.best0 = run0 ? w.T0.best : w.T1.best; h.best1 = 0;
            h.cells0 = run0 ? w.T0.cells : w.T1.cells; h.cells1 = 0;
            h.act = w_lanes(h.lo0, h.hi0 - h.lo0 + 1);
            h.in = 0ull;
            budget = min((run0 ? w.T0.max_d : w.T1.max_d) - t_d, tape_left);
            // (a parked neighbour joins again when the two bands fit with room to spare)
            if (w.T0.state == A2_PARKED || w.T1.state == A2_PARKED) {
                int plo, phi;
                if (w.T0.state == A2_PARKED) a2_next_band(w.T0, 0, plo, phi); else a2_next_band(w.T1, 0, plo, phi);
                join_at = max(1, 64 - A2_FREE_JOIN - (phi - plo + 1));
            }
        }
        // what the row loop's function needs to park and to join on its own (a waiting track
        // without rows of its own is placed by a2_place, not by a2_join: not announced)
        A2Drive dr;
        dr.pair = w.pair;
        dr.parked = (w.T0.state == A2_PARKED && w.T0.d > 0) ? 1 : (w.T1.state == A2_PARKED && w.T1.d > 0) ? 2 : 0;
        dr.p_kc = dr.parked == 1 ? w.T0.kc : w.T1.kc;
        dr.p_li = dr.parked == 1 ? w.T0.li : w.T1.li;
        dr.p_hin = dr.parked == 1 ? w.T0.hin : w.T1.hin;
        dr.p_best = dr.parked == 1 ? w.T0.best : w.T1.best;
        dr.p_cells = dr.parked == 1 ? w.T0.cells : w.T1.cells;
        dr.rem0 = w.T0.state != A2_IDLE ? w.T0.max_d - w.T0.d : 0;
        dr.rem1 = w.T1.state != A2_IDLE ? w.T1.max_d - w.T1.d : 0;
        dr.tape_left = tape_left;
        dr.d0 = dr.d1 = 0;
        dr.it_pair = dr.it_single = dr.n_park = dr.n_join = 0u;
        {
            const u32 it_last = h.it + (u32)budget;
            const u32 it_end = (!w.pair && join_at > 0 && budget > A2_LOOK_EVERY) ? h.it + A2_LOOK_EVERY : it_last;
            a2_drive_call(A, w, wl, h, hv, it_end, it_last, join_at, dr);
        }
        // ---- back from the row loop: at least one row was computed; the pair may have been
        // left and formed again any number of times, the state is the last stretch's
        w.st_pair += dr.it_pair; w.st_single += dr.it_single;
        w.st_park += dr.n_park; w.st_place += dr.n_join;
        const int done_it = dr.d0 + dr.d1;
        w.pair = dr.pair;
        wl.vx = hv.vx; wl.vacc = hv.vacc; wl.rc_mlo = hv.rc_mlo; wl.rc_mhi = hv.rc_mhi;
        w.it = h.it;
        w.n_esc = h.n_esc;
        const int p_last = (int)(w.it & 1u) ^ 1;
        const bool ran0 = h.kb0 != A2_INVALID, ran1 = h.kb1 != A2_INVALID;  // who ran the last stretch
        // the hulls of the last row: the lanes of it that passed the band filter
        if (w.pair) {
            const u64 in0 = h.in & a2_zone(w, 0), in1 = h.in & a2_zone(w, 1);
            w.T0.d += dr.d0; w.T1.d += dr.d1;
            w.T0.kc = (int)h.kb0 - 1 + p_last; w.T1.kc = (int)h.kb1 - 1 + p_last;
            w.T0.li = w_lowest(in0); w.T0.hin = w_highest(in0);
            w.T1.li = w_lowest(in1); w.T1.hin = w_highest(in1);
            w.T0.best = h.best0;
            w.T1.best = (int)((u32)h.best1 - A2_TOP);
            w.T0.cells = h.cells0;
            w.T1.cells = h.cells1;
            w.T0.state = A2_RUN; w.T1.state = A2_RUN;
        } else {
            const int n_li = w_lowest(h.in), n_hin = w_highest(h.in);
            const u32 n_cells = h.cells0;
            if (ran0) {
                w.T0.d += dr.d0; w.T0.kc = (int)h.kb0 - 1 + p_last; w.T0.li = n_li; w.T0.hin = n_hin;
                w.T0.best = h.best0; w.T0.cells = n_cells;
                w.T0.state = A2_RUN;
            } else {
                w.T1.d += dr.d1; w.T1.kc = (int)h.kb1 - 1 + p_last; w.T1.li = n_li; w.T1.hin = n_hin;
                w.T1.best = h.best0; w.T1.cells = n_cells;
                w.T1.state = A2_RUN;
            }
            // a track that was parked in there waits with what a2_park kept of it
            if (dr.parked == 1 && (dr.d0 > 0 || w.T0.state == A2_RUN)) {
                w.T0.d += dr.d0; w.T0.kc = dr.p_kc; w.T0.li = dr.p_li; w.T0.hin = dr.p_hin;
                w.T0.best = dr.p_best; w.T0.cells = dr.p_cells;
                w.T0.state = A2_PARKED;
            } else if (dr.parked == 2 && (dr.d1 > 0 || w.T1.state == A2_RUN)) {
                w.T1.d += dr.d1; w.T1.kc = dr.p_kc; w.T1.li = dr.p_li; w.T1.hin = dr.p_hin;
                w.T1.best = dr.p_best; w.T1.cells = dr.p_cells;
                w.T1.state = A2_PARKED;
            }
        }
#ifdef A2_HOOK_EXIT
        A2_HOOK_EXIT(w, h);
#endif
        if (w.n_esc > A.esc_cap) {  // the escape list is full: everybody on the tape goes back
            if (w.T0.state != A2_IDLE) { a2_hand_back(A, w, w.T0); w.st_bail_esc++; }
            if (w.T1.state != A2_IDLE) { a2_hand_back(A, w, w.T1); w.st_bail_esc++; }
            w.n_esc = 0;
            continue;
        }
        if (done_it > 0) {
            const u64 fin0 = h.fin & a2_zone(w, 0), fin1 = h.fin & a2_zone(w, 1);
            if (ran0 && fin0) a2_finish<0>(A, w, wl, w.T0, fin0, h.act_row);
            if (ran1 && fin1) a2_finish<1>(A, w, wl, w.T1, fin1, h.act_row);
        }
        // rows exhausted without reaching an end: unaligned (DW_banded.c:183, :171)
        if (w.T0.state == A2_RUN && w.T0.d >= w.T0.max_d) {
            a2_result(A, w.T0.g, 0, 0, 0, 0, 0, 0, (long long)w.T0.cells);
            w.T0.state = A2_IDLE;
        }
        if (w.T1.state == A2_RUN && w.T1.d >= w.T1.max_d) {
            a2_result(A, w.T1.g, 0, 0, 0, 0, 0, 0, (long long)w.T1.cells);
            w.T1.state = A2_IDLE;
        }
    }
    if (A.stats) {
        w_stat_add(A.stats + A2_STAT_PAIR_IT, w.st_pair);
        w_stat_add(A.stats + A2_STAT_SINGLE_IT, w.st_single);
        w_stat_add(A.stats + A2_STAT_PLACE, w.st_place);
        w_stat_add(A.stats + A2_STAT_PARK, w.st_park);
        w_stat_add(A.stats + A2_STAT_BAIL, w.st_bail);
        w_stat_add(A.stats + A2_STAT_TRACKS, w.st_recenter);
        w_stat_add(A.stats + A2_STAT_EXT, w.st_wide);
        w_stat_add(A.stats + A2_STAT_ESC, w.st_wide_rows);
        w_stat_add(A.stats + A2_STAT_BAIL_TAPE, w.st_bail_tape);
        w_stat_add(A.stats + A2_STAT_BAIL_WIDE, w.st_bail_wide);
        w_stat_add(A.stats + A2_STAT_BAIL_ESC, w.st_bail_esc);
    }
}
