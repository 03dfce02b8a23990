// k_align.hip -- banded O(ND) alignment with trace-back, one wavefront per alignment: the
// GENERAL kernel of the alignment stage.  The falcon_sense path runs k_align2 (two alignments
// per wavefront, k_align2.hip); this kernel takes what k_align2 hands back (FaAln.err == 2:
// repeated alone in worst-case slots), band tolerances below 64, and everything when
// FALCON_AMD_ALIGN1 is set (A/B runs, tests).
//
// Restates align() of the reference (src/c/DW_banded.c:115-330) for every
// (read window, seed window) pair of the work list:
//
//   forward  lane l owns diagonal kd + 2 l of the row, kd falling by one per row; the previous
//            row of the furthest-reaching table lives in ONE VGPR, so V[k+1] and V[k-1] are the
//            lane's own value and one DPP wave_shr:1 -- no parity case; the band climbs a lane
//            every two rows and is re-seated with a shuffle every few dozen rows.  Rows wider
//            than 60 diagonals (rare) fall back to a ring in LDS, up to 3 passes of 64 cells.
//            All cells of a row depend only on the previous row, so a row is
//            one data-parallel step:
//              - the snake compares 2-bit packed bases, 16 per step (v_alignbit of two packed
//                words, xor, v_ffbl), read through the vector L1 (FALCON_AMD_ALIGN_LDS=1: from
//                windows staged in LDS -- slower, it costs occupancy);
//              - lane sets (band, from_above, finished, next band) are scalar 64-bit masks:
//                every ballot is one v_cmp, the masks are combined by the scalar unit;
//              - best_m is a DPP wave maximum, the next band comes from s_ff1 / s_flbit, the
//                finishing diagonal ("first k in ascending order", DW_banded.c:220-224) from
//                a ballot;
//            each cell's reached x goes to the slot's arena (4 bytes, coalesced), one 16-byte
//            record per row (first diagonal, cell offset, from_above bits) is kept in a
//            register ring (v_writelane) and flushed 64 rows at a time.
//   trace    (DW_banded.c:264-319) the diagonal chain k_d is resolved 64 rows at a time from the
//            per-row from_above bits held in registers (v_readlane), then all 64 rows gather
//            their x in parallel and emit one u32 per row: (snake_length << 1) | from_above.
//            The gapped strings of the reference are never materialised; they are a pure
//            function of this edit script and the two sequences.
//
// The reference's qsort + bsearch over (d,k) records (:260,:267-277) and its
// O(max_d * band) calloc (:164) have no counterpart here.
//
// Bound by the CU's scalar and vector instruction streams; 4 B per DP
// cell to HBM; no MFMA.
#include "fa_device.h"
#include <cstdlib>
#include <type_traits>

#define RING 256  // V entries per parity; live band <= 191 diagonals

// FA_ALIGN_PROF: per-section s_memtime accounting of the row loop (debug builds only)
#ifdef FA_ALIGN_PROF
#define PROF_DECL u64 pf_t = __builtin_amdgcn_s_memtime(), pf_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PROF(i) do { const u64 n_ = __builtin_amdgcn_s_memtime(); pf_acc[i] += n_ - pf_t; pf_t = n_; } while (0)
#define PROF_FLUSH(p) do { if (lane == 0) for (int i_ = 0; i_ < 8; i_++) atomicAdd((unsigned long long *)(p) + i_, (unsigned long long)pf_acc[i_]); } while (0)
#else
#define PROF_DECL
#define PROF(i) do { } while (0)
#define PROF_FLUSH(p) do { } while (0)
#endif

#ifdef FA_TRACE_KERNEL
#define KTRACE(...) do { if (lane == 0) printf(__VA_ARGS__); } while (0)
#else
#define KTRACE(...) do { } while (0)
#endif

struct AlignArgs {
    const u32 *words;
    const FaSeq *seq;
    const FaPile *pile;
    const FaRange *range;
    const int *order;
    int n_work;
    int *counter;
    u32 *cells;
    FaRowRec *rows;
    FaRowExt *rowx;
    u64 cells_per_slot;
    u64 rows_per_slot;
    u32 *script;
    const u64 *script_off;
    FaAln *aln;
    int band;
    int lds_q_words;
    int lds_t_words;
    double max_diff;
    u64 *prof;  // FA_ALIGN_PROF builds: 8 section counters
};

// One alignment.  qL/tL: LDS windows (word aligned), qb/tb: base offset of
// window position 0 inside the first staged word.  Vring: 2 x RING ints.
// (kept out of line: inlined into the persistent work loop, hipcc 7.2 fuses the
// row loop with the work loop into one state machine and the per-row scalar
// overhead triples)
// LDS is addressed through the kernel's dynamic __shared__ array plus word
// offsets, never through generic pointers (those would compile to flat_load).
//
// SEQ_LDS = true : the read/seed windows were staged in LDS (q_off/t_off are
//                  word offsets into the dynamic __shared__ array);
// SEQ_LDS = false: the snake reads the packed words straight from HBM through
//                  the vector L1 (qg_/tg_ point at the first window word); LDS
//                  then only holds the 2 KB V ring, which is what lets 24+
//                  wavefronts share a CU -- the row loop is a dependent chain,
//                  so resident waves, not instruction count, set the pace.
template <bool SEQ_LDS>
__device__ __noinline__ void align_one(int band_v, u64 rows_per_slot_v, double max_diff_v,
                                       u32 *script_v, FaAln *aln_out_v, int q_off_v, int qb_v,
                                       int q_len_v, int t_off_v, int tb_v, int t_len_v, int v_off_v,
                                       u32 *cells_v, FaRowRec *rows_v, FaRowExt *rowx_v,
                                       const u32 *qg_v, const u32 *tg_v, u64 *prof_v) {
    // every argument is wave-uniform; pin them in SGPRs (see fa_uni)
    const int band = fa_uni(band_v), q_off = fa_uni(q_off_v), qb = fa_uni(qb_v);
    const int q_len = fa_uni(q_len_v), t_off = fa_uni(t_off_v), tb = fa_uni(tb_v);
    const int t_len = fa_uni(t_len_v), v_off = fa_uni(v_off_v);
    // (one argument for both slot dimensions: rows in the low word, cells in the high one)
    const u64 slot_dims = fa_uni(rows_per_slot_v);
    const u64 rows_per_slot = slot_dims & 0xffffffffull;
    const u32 cells_cap = (u32)(slot_dims >> 32);
    const double max_diff = fa_uni(max_diff_v);
    u32 *script_ = fa_uni(script_v);
    FaAln *aln_out_ = fa_uni(aln_out_v);
    u32 *cells_ = fa_uni(cells_v);
    FaRowRec *rows_ = fa_uni(rows_v);
    FaRowExt *rowx_ = fa_uni(rowx_v);
    const u32 *qg_ = fa_uni(qg_v), *tg_ = fa_uni(tg_v);
    u64 *prof_ = fa_uni(prof_v);
    (void)prof_;
    extern __shared__ __attribute__((aligned(16))) u32 smem[];
    typedef const __attribute__((address_space(1))) u32 *gcu32p;
    typedef typename std::conditional<SEQ_LDS, const u32 *, gcu32p>::type seqp;
    seqp qL, tL;
    if constexpr (SEQ_LDS) {
        qL = smem + q_off;
        tL = smem + t_off;
    } else {
        qL = (gcu32p)qg_;
        tL = (gcu32p)tg_;
    }
    int *Vring = (int *)(smem + v_off);
    // arena / output pointers are global memory: say so, or they become flat_*
    // (records are moved as clang vector types: struct assignment is only
    // defined for the generic address space)
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    typedef u32 u32x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(1))) u32 g_u32;
    typedef __attribute__((address_space(1))) u32x4 g_u32x4;
    typedef __attribute__((address_space(1))) u32x2 g_u32x2;
    static_assert(sizeof(FaRowRec) == 16 && sizeof(FaRowExt) == 16 && sizeof(FaAln) == 40, "layout");
    g_u32 *cells = (g_u32 *)cells_;
    g_u32 *script = (g_u32 *)script_;
    g_u32x4 *rows = (g_u32x4 *)rows_;    // {off, min_k, dlo, dhi}
    g_u32x4 *rowx = (g_u32x4 *)rowx_;    // {d1.lo, d1.hi, d2.lo, d2.hi}
    auto store_result = [&](const FaAln &r) {
        g_u32x4 *o4 = (g_u32x4 *)aln_out_;
        u32x4 a = {(u32)r.dist, (u32)r.q_e, (u32)r.t_e, (u32)r.size};
        u32x4 b = {(u32)r.accept, (u32)r.n_ins, (u32)r.aligned, (u32)r.err};
        u32x2 c = {(u32)(u64)r.cells, (u32)((u64)r.cells >> 32)};
        o4[0] = a;
        o4[1] = b;
        ((g_u32x2 *)aln_out_)[4] = c;
    };
    const int lane = fa_lane();
    FaAln res;
    res.dist = 0; res.q_e = 0; res.t_e = 0; res.size = 0; res.accept = 0; res.n_ins = 0;
    res.aligned = 0; res.err = 0; res.cells = 0;

    // (the double arithmetic runs on the VALU: pin the result, or every loop bound derived
    // from it -- the hot loop's row countdown -- lives in a VGPR)
    const int max_d = fa_uni((int)(0.3 * (double)(q_len + t_len)));  // DW_banded.c:149
    if ((u64)max_d > rows_per_slot) {  // host sized the slot from the same formula
        res.err = 1;
        store_result(res);
        return;
    }
    // zero the ring (reference: calloc'ed V, :153)
    for (int i = lane; i < 2 * RING; i += 64) Vring[i] = 0;
    KTRACE("align q_len=%d t_len=%d max_d=%d qb=%d tb=%d\n", q_len, t_len, max_d, qb, tb);

    int best_m = -1, min_k = 0, n = 1;  // n = cells of the row, max_k = min_k + 2(n-1)
    u32 row_off = 0;
    int fin_d = -1, fin_k = 0, fin_x = 0, fin_y = 0;
    // the last <= 64 row records live in registers, lane (d & 63) holds row d;
    // they are flushed to the arena 64 at a time with one coalesced 16-byte store
    // A record says where bit 0 of the row's from_above bits sits: its cell (`off`) and its
    // diagonal, NEGATED (`-k0`; bit i <-> diagonal k0 + 2 i <-> cell off + i).  Register-mode
    // rows store the lane mask as it is -- bit = lane, k0 = kd, off = row_off - lo -- so that
    // the hot loop does no shifting or min_k arithmetic on the scalar pipe; ring-mode rows
    // bit = cell index in the row, k0 = min_k, off = row_off.
    u32 rc_off = 0, rc_mink = 0, rc_dlo = 0, rc_dhi = 0;
    // WRITE_ROW_RECORD: row d's record into lane d & 63 (v_writelane; lane select in M0:
    // a VOP3 may read only one SGPR besides it).  FLUSH_ROW_RECORDS: rows d - (d & 63) .. d.
#define WRITE_ROW_RECORD(off_expr, dir0_, neg_k0_expr)                            \
    do {                                                                          \
        const u64 dw_ = (dir0_);                                                  \
        asm volatile("s_mov_b32 m0, %8\n\t"                                       \
                     "v_writelane_b32 %0, %4, m0\n\t"                             \
                     "v_writelane_b32 %1, %5, m0\n\t"                             \
                     "v_writelane_b32 %2, %6, m0\n\t"                             \
                     "v_writelane_b32 %3, %7, m0"                                 \
                     : "+v"(rc_off), "+v"(rc_mink), "+v"(rc_dlo), "+v"(rc_dhi)    \
                     : "s"((u32)(off_expr)), "s"((u32)(neg_k0_expr)), "s"((u32)dw_), \
                       "s"((u32)(dw_ >> 32)), "s"(d & 63));                       \
    } while (0)
#define FLUSH_ROW_RECORDS()                                                       \
    do {                                                                          \
        const int slot_ = d & 63;                                                 \
        if (lane <= slot_) {                                                      \
            const u32x4 rr_ = {rc_off, rc_mink, rc_dlo, rc_dhi};                  \
            rows[d - slot_ + lane] = rr_;                                         \
        }                                                                         \
    } while (0)
#define PUT_ROW_RECORD(dir0_, finished_)                                          \
    do {                                                                          \
        WRITE_ROW_RECORD(row_off, dir0_, -min_k);                                 \
        if ((d & 63) == 63 || (finished_)) FLUSH_ROW_RECORDS();                   \
    } while (0)

    // Register mode (rows of <= REG_MAX_N diagonals, i.e. nearly all of them):
    // lane l owns diagonal kd + 2l, kd falling by one per row, for as long as the
    // band stays inside the wave, and vreg holds the previous row's x on diagonal
    // (kd + 1) + 2l.  The two neighbours a = V[k-1], b = V[k+1] (DW_banded.c:190-196)
    // are then ONE DPP wave shift and the lane's own register -- no LDS traffic for V
    // at all, no parity case.  The band climbs one lane every two rows and is
    // re-seated with a shuffle every few dozen rows.  Wider rows fall back to the
    // LDS ring (ring mode) until the band narrows again.  The two modes are two
    // plain loops (the row loop must stay a simple loop: instruction issue, scalar
    // and vector alike, is the bottleneck of this kernel).
    const int m2lane = -2 * lane;
    const int REG_MAX_N = 60;
    const int nmax = min(REG_MAX_N, band + 1);
    int kd = -62;          // diagonal of lane 0 in row d (row 0: diagonal 0 sits on lane 31)
    int lo = 31;           // lane of min_k in register mode
    int vreg = 0;          // reference: calloc'ed V (:153)
    int d = 0;
    bool done = false, dead = false;
    // the slot holds fewer cells than the worst case rows x (band + 1) (the engine sizes it
    // for what alignments use, fa_internal.h): running out is reported, never written past
    bool overflow = false;
    PROF_DECL;
    while (!done && !dead) {
        // ================= register-mode rows =================
        // (the finishing row only leaves the loop; what it found is worked out after
        // it -- anything more inside and the compiler folds the row loop into the outer
        // mode loop, at a dozen scalar instructions per row)
        u64 fin = 0ull;
        int x = 0, y = 0;
        // The rare events -- rows exhausted (:183), band too wide (:184), row too wide
        // for register mode, band climbing out of the wave -- are tested when a
        // countdown runs out, not every row: each of their margins shrinks by at most
        // one per row (d and the band's top lane grow by one, n by at most one; the
        // band's bottom lane never moves down), so after a test that leaves a smallest
        // margin of s the next s rows cannot trip any of them.
        for (;;) {
            // a block of 64 row records is complete: store it (the hot loop below never
            // runs across such a boundary)
            if ((d & 63) == 0 && d > 0) { d--; FLUSH_ROW_RECORDS(); d++; }
            min_k = kd + 2 * lo;  // (register mode keeps min_k implicit)
            if (d >= max_d || n - 1 > band) { dead = true; break; }
            if (n > REG_MAX_N) break;
            // rows of <= 64 cells that still fit the slot, less one
            const int room = (int)((cells_cap - row_off) >> 6) - 1;
            if (room < 0) { overflow = true; dead = true; break; }
            if (63 - lo - n < 0 || lo < 1) {
                // re-seat the band low inside the wave (it climbs one lane every two rows)
                const int nlo = max(1, (64 - n) >> 2);
                const int sh = lo - nlo;        // new lane l takes old lane l + sh
                vreg = __shfl(vreg, lane + sh);
                kd += 2 * sh;
                lo = nlo;
            }
            // the hot loop: `safe` + 1 rows that cannot trip a rare event.  Two exits, one join
            // (a finishing row | the countdown): with a jump out of both loops the compiler
            // builds a state machine around the row, a dozen scalar instructions per row on
            // the CU's one scalar pipe -- which is what bounds this kernel.
            // (`safe` pinned: left to itself the compiler keeps this wave-uniform countdown
            // in a VGPR -- three VALU operations per row)
            int safe = fa_uni(min(min(min(min(max_d - 1 - d, nmax - n), 63 - lo - n), 63 - (d & 63)), room));
            int hi = lo + n - 1;
            int nkd = -kd;  // (the loop carries -kd: what the row adds)
            for (;;) {
            PROF(0);
            // Lane sets are kept twice: as a predicate (act: stores, selects) and as a
            // scalar mask (act_m: combined with single-compare ballots by the scalar unit).
            const u64 act_m = fa_lane_range(lo, n);
            // the same set as a predicate, straight from the mask: no per-row v_cmp pair
            const bool act = __builtin_amdgcn_inverse_ballot_w64(act_m);
            // Lane l owns diagonal kd + 2l with kd falling by one per row: the lane's own
            // register then always holds V[k+1] of the previous row and lane l-1 holds
            // V[k-1] -- one DPP shift, no parity case.  (The wave's edge lanes are never
            // inside the band: lo >= 1, hi <= 62.)
            const int a = __builtin_amdgcn_mov_dpp(vreg, 0x138, 0xf, 0xf, true);  // lane-1: V[k-1]
            const int b = vreg;                                                  //         V[k+1]
            // from_above: k == min_k, or k != max_k and V[k-1] < V[k+1]   (:190)
            const u64 fa_m = fa_mask_clr_set(fa_ballot(a < b), hi, lo) & act_m;
            x = fa_sel(fa_m, a + 1, b);
            y = fa_add3(x, m2lane, nkd);  // x - k, k = kd + 2 lane: one v_add3
            PROF(1);
            // snake and cell store under ONE exec mask, the band's: the loads carry 27
            // lanes instead of 64 (the vector memory pipe is the third thing this kernel
            // leans on: an all-lane store, even with the idle lanes dropped by a buffer
            // range check, was measured 72 -> 77 ms; idle lanes to a spare cell: 88 ms)
            const u32 off0 = row_off - (u32)lo;  // the cell lane 0 would have: record and store address
            if (act) {
                snake16_band(qL, tL, qb, tb, q_len, t_len, x, y);
                cells[off0 + (u32)lane] = fa_twice_plus(x, fa_m);  // x<<1 | from_above
            }
            PROF(2);
            vreg = x;
            fin = (fa_ballot(x >= q_len) | fa_ballot(y >= t_len)) & act_m;  // :220
            PROF(3);
            WRITE_ROW_RECORD(off0, fa_m, nkd);
            if (fin) break;  // (its records are flushed below)
            // (an LDS ds_max on one word instead of the DPP reduction was measured 2x
            // slower: 64 same-address atomics serialise)
            const int u = act ? x + y : -1;
            best_m = max(best_m, fa_wave_max_nonneg(u));  // some lane is active: max >= 0
            const u64 in = fa_ballot(u >= best_m - band) & act_m;  // :228-243
            PROF(4);
            // `in` is never empty: best_m is attained inside the row
            row_off += (u32)n;
            lo = __builtin_ctzll(in);            // absolute lanes: the new lowest diagonal
            hi = 64 - __builtin_clzll(in);       // (min_k = kd + 2 lo), one below, sits on the
            n = hi - lo + 1;                     // same lane one row on; the highest one lane up
            nkd++;
            d++;
            PROF(5);
            if (--safe < 0) break;
            }
            kd = -nkd;
            if (fin) break;
        }
        min_k = kd + 2 * lo;
        if (fin) {  // row d finished the alignment on its first (lowest) such diagonal
            FLUSH_ROW_RECORDS();
            const int fl = __builtin_ctzll(fin);
            fin_d = d;
            fin_k = kd + 2 * fl;
            fin_x = __builtin_amdgcn_readlane(x, fl);
            fin_y = __builtin_amdgcn_readlane(y, fl);
            res.cells = (long long)row_off + (fl - lo) + 1;
            done = true;
        }
        if (done || dead) break;
        // ================= ring-mode rows (61..191 diagonals) =================
        {   // spill the previous row into the ring: lane l holds diagonal kd + 1 + 2l of it
            const int par = d & 1;
            const int kp = kd + 1 + 2 * lane;
            Vring[(par ^ 1) * RING + ((kp >> 1) & (RING - 1))] = vreg;
        }
        for (;;) {
            if (d >= max_d || n - 1 > band) { dead = true; break; }
            if (n <= REG_MAX_N - 12) break;
            if (row_off + (u32)n > cells_cap) { overflow = true; dead = true; break; }
            const int par = d & 1;
            const int max_k = min_k + 2 * (n - 1);
            int *Vcur = Vring + par * RING;
            const int *Vprev = Vring + (par ^ 1) * RING;
            g_u32 *rowp = cells + row_off;
            u64 dirw[FA_ALIGN_MAXCH] = {0, 0, 0};
            int row_max = -1;
            bool finished = false;
#pragma unroll
            for (int c = 0; c < FA_ALIGN_MAXCH; c++) {
                if (c * 64 < n && !finished) {
                    const int j = c * 64 + lane;
                    const bool act = j < n;
                    const int k = min_k + 2 * j;
                    const int a = Vprev[((k - 1) >> 1) & (RING - 1)];
                    const int b = Vprev[((k + 1) >> 1) & (RING - 1)];
                    const bool from_above = (j == 0) || ((k != max_k) && (a < b));
                    int x = from_above ? b : a + 1;
                    int y = x - k;
                    snake16(qL, tL, qb, tb, q_len, t_len, act, x, y);
                    if (act) {
                        Vcur[(k >> 1) & (RING - 1)] = x;
                        rowp[j] = ((u32)x << 1) | (from_above ? 1u : 0u);
                        row_max = max(row_max, x + y);
                    }
                    dirw[c] = __ballot(act && from_above);
                    const u64 fin = __ballot(act && (x >= q_len || y >= t_len));
                    if (fin) {
                        const int fl = __ffsll((long long)fin) - 1;
                        fin_d = d;
                        fin_k = min_k + 2 * (c * 64 + fl);
                        fin_x = __builtin_amdgcn_readlane(x, fl);
                        fin_y = __builtin_amdgcn_readlane(y, fl);
                        res.cells = (long long)row_off + c * 64 + fl + 1;
                        finished = true;
                    }
                }
            }
            if (n > 64) {   // the two extra direction words of a wide row
                const u32x4 ex = {(u32)dirw[1], (u32)(dirw[1] >> 32), (u32)dirw[2],
                                  (u32)(dirw[2] >> 32)};
                rowx[d] = ex;  // every lane stores the same record
            }
            PUT_ROW_RECORD(dirw[0], finished);
            if (finished) { done = true; break; }
            best_m = max(best_m, fa_wave_max(row_max));
            int jlo = -1, jhi = -1;
#pragma unroll
            for (int c = 0; c < FA_ALIGN_MAXCH; c++) {
                if (c * 64 < n) {
                    const int j = c * 64 + lane;
                    const int k = min_k + 2 * j;
                    const int x = Vcur[(k >> 1) & (RING - 1)];
                    const u64 in = __ballot(j < n && (2 * x - k) >= best_m - band);
                    if (in) {
                        if (jlo < 0) jlo = c * 64 + __ffsll((long long)in) - 1;
                        jhi = c * 64 + 63 - __clzll((long long)in);
                    }
                }
            }
            row_off += (u32)n;
            min_k = min_k + 2 * jlo - 1;  // jlo >= 0: best_m is attained inside the row
            n = jhi - jlo + 2;
            d++;
        }
        if (done || dead) break;
        {   // reload into registers: put min_k on lane (64-n)/4
            const int par = d & 1;
            lo = max(1, (64 - n) >> 2);
            kd = min_k - 2 * lo;
            const int kp = kd + 1 + 2 * lane;
            vreg = Vring[(par ^ 1) * RING + ((kp >> 1) & (RING - 1))];
        }
    }
#undef PUT_ROW_RECORD
#undef WRITE_ROW_RECORD
#undef FLUSH_ROW_RECORDS

    if (!done) {  // unaligned: aln_str_size stays 0 (:171,:184-186)
        res.cells = row_off;
        res.err = overflow ? 2 : 0;  // 2: the engine repeats the launch with full-size slots
        store_result(res);
        return;
    }
    KTRACE(" forward done fin_d=%d fin_k=%d x=%d y=%d\n", fin_d, fin_k, fin_x, fin_y);
    res.aligned = 1;
    res.dist = fin_d;
    res.q_e = fin_x;
    res.t_e = fin_y;
    res.size = (fin_x + fin_y + fin_d) / 2;  // :248

    // ---- trace-back (DW_banded.c:264-319) -----------------------------------
    // 64 rows per block: lane l loads the record of row hi-l and parks
    // {min_k, dlo, dhi} in LDS; the diagonal chain k_d is then resolved on the
    // VALU from broadcast LDS reads (the scalar unit is the busy one), after
    // which all 64 rows gather their cell and emit one script word each.
    __threadfence_block();
    u32 *tb_rec = (u32 *)Vring;  // 64 x 4 words, the ring is free now
    int k_cur = fin_k;
    int n_ins = 0;
    for (int hi = fin_d; hi >= 0; hi -= 63) {
        // lanes 0..63 <-> rows hi, hi-1, ...; lane 63 (or the last row above 0)
        // is look-ahead only and is re-done by the next block.
        const int r = hi - lane;
        const bool have = r >= 0;
        u32x4 rv = {0u, 0u, 0u, 0u};
        if (have) rv = rows[r];
        tb_rec[4 * lane + 0] = 0u - rv.y;  // k0 (stored negated)
        tb_rec[4 * lane + 1] = rv.z;  // dlo
        tb_rec[4 * lane + 2] = rv.w;  // dhi
        const int n_rows = min(64, hi + 1);
        int my_k = 0, my_dir = 0;
        int kv = k_cur;  // the chain lives in a VGPR (same value in every lane)
        for (int l = 0; l < n_rows; l++) {
            const int mk = (int)tb_rec[4 * l + 0];
            const u32 dlo = tb_rec[4 * l + 1], dhi = tb_rec[4 * l + 2];
            const int j = (kv - mk) >> 1;
            u64 wbits = ((u64)dhi << 32) | dlo;
            if (j >= 64) {  // wide row: fetch the extra words (rare)
                const u32x4 ex = rowx[hi - l];
                wbits = (j < 128) ? (((u64)ex.y << 32) | ex.x) : (((u64)ex.w << 32) | ex.z);
            }
            const int bit = (int)((wbits >> (j & 63)) & 1ull);
            if (lane == l) {
                my_k = kv;
                my_dir = bit;
            }
            if (l < 63 || hi < 63) kv += 2 * bit - 1;  // lane 63's row is redone next block
        }
        k_cur = __builtin_amdgcn_readfirstlane(kv);
        // x2 of my row on the path
        int x2 = 0;
        if (have) x2 = (int)(cells[rv.x + (u32)((my_k + (int)rv.y) >> 1)] >> 1);
        int x2_prev = __shfl_down(x2, 1);  // row r-1 sits in lane+1
        const bool emit = have && (lane < 63 || hi < 63);
        if (emit) {
            int m;
            if (r == 0) {
                m = x2;  // row 0 starts at (0,0), no edit
                my_dir = 0;
            } else {
                const int x1 = my_dir ? x2_prev : x2_prev + 1;
                m = x2 - x1;
            }
            script[r] = ((u32)m << 1) | (u32)my_dir;
        }
        n_ins += __popcll(__ballot(emit && r > 0 && my_dir == 0));
        if (hi < 63) break;
        // k_cur now is the diagonal of row hi-63 (lane 63's row): restart there
    }
    PROF(6);
    PROF_FLUSH(prof_);
    KTRACE(" trace done n_ins=%d\n", n_ins);
    res.n_ins = n_ins;
    res.accept = (res.size > 500) &&
                 ((double)res.dist / (double)res.size < max_diff);  // falcon.c:629
    store_result(res);  // every lane stores the same record
}

template <bool SEQ_LDS>
__global__ __launch_bounds__(64, (SEQ_LDS ? 3 : 8)) void k_align(AlignArgs A) {
    extern __shared__ __attribute__((aligned(16))) u32 smem[];
    u32 *qL = smem;
    u32 *tL = qL + A.lds_q_words;
    const int lane = fa_lane();
    const int slot = blockIdx.x;
    u32 *cells = A.cells + (u64)slot * A.cells_per_slot;
    FaRowRec *rows = A.rows + (u64)slot * A.rows_per_slot;
    FaRowExt *rowx = A.rowx + (u64)slot * A.rows_per_slot;

    for (;;) {
        // One work item per trip.  Everything that steers control flow is forced
        // into SGPRs (readfirstlane) so the trip is provably wave-uniform.
        // (no `if (lane == 0)` around the atomic or the result stores of this
        // loop: hipcc 7.2 jump-threads such blocks across the back edge and
        // folds the readfirstlane on the lane != 0 path, which livelocks)
        int wi = atomicAdd(A.counter, lane == 0 ? 1 : 0);
        wi = __builtin_amdgcn_readfirstlane(wi);
        KTRACE("block %d got wi=%d of %d\n", (int)blockIdx.x, wi, A.n_work);
        if (wi >= A.n_work) break;
        const int g = __builtin_amdgcn_readfirstlane(A.order[wi]);
        const int q_idx = __builtin_amdgcn_readfirstlane(A.seq[g].idx);
        const int q_slen = __builtin_amdgcn_readfirstlane(A.seq[g].len);
        const u32 q_woff = (u32)__builtin_amdgcn_readfirstlane((int)A.seq[g].woff);
        const int pile_id = __builtin_amdgcn_readfirstlane(A.seq[g].pile);
        const int s1 = __builtin_amdgcn_readfirstlane(A.range[g].s1);
        const int e1 = __builtin_amdgcn_readfirstlane(A.range[g].e1);
        const int s2 = __builtin_amdgcn_readfirstlane(A.range[g].s2);
        const int e2 = __builtin_amdgcn_readfirstlane(A.range[g].e2);
        const int rg_ok = __builtin_amdgcn_readfirstlane(A.range[g].ok);
        const int seed_g = __builtin_amdgcn_readfirstlane(A.pile[pile_id].first);
        const int t_slen = __builtin_amdgcn_readfirstlane(A.seq[seed_g].len);
        const u32 t_woff = (u32)__builtin_amdgcn_readfirstlane((int)A.seq[seed_g].woff);

        const int q_len = e1 - s1, t_len = e2 - s2;  // falcon.c:626-627
        // stage the two windows: words [s/16, (s+len)/16 + 2]
        const int qw0 = s1 >> 4, tw0 = s2 >> 4;
        const int qn = ((s1 + q_len) >> 4) - qw0 + 3;
        const int tn = ((s2 + t_len) >> 4) - tw0 + 3;
        const bool skip = (q_idx == 0) || !rg_ok;
        const bool too_big = SEQ_LDS && !skip && (qn > A.lds_q_words || tn > A.lds_t_words);
        if (skip || too_big) {
            FaAln z;
            z.dist = 0; z.q_e = 0; z.t_e = 0; z.size = 0; z.accept = 0; z.n_ins = 0;
            z.aligned = 0; z.err = too_big ? 1 : 0; z.cells = 0;
            A.aln[g] = z;  // every lane stores the same record
        } else {
            const u32 *qg = A.words + q_woff + qw0;
            const u32 *tg = A.words + t_woff + tw0;
            // the sequence's own words end at ceil(len/16)+2 (zero padded by pack)
            const int qavail = ((q_slen + 15) >> 4) + 2 - qw0;
            const int tavail = ((t_slen + 15) >> 4) + 2 - tw0;
            if constexpr (SEQ_LDS) {
                for (int i = lane; i < qn; i += 64) qL[i] = (i < qavail) ? qg[i] : 0u;
                for (int i = lane; i < tn; i += 64) tL[i] = (i < tavail) ? tg[i] : 0u;
                __syncthreads();
                align_one<true>(A.band, A.rows_per_slot | (A.cells_per_slot << 32), A.max_diff, A.script + A.script_off[g],
                                A.aln + g, 0, s1 & 15, q_len, A.lds_q_words, s2 & 15, t_len,
                                A.lds_q_words + A.lds_t_words, cells, rows, rowx, qg, tg, A.prof);
            } else {
                (void)qavail; (void)tavail;
                align_one<false>(A.band, A.rows_per_slot | (A.cells_per_slot << 32), A.max_diff, A.script + A.script_off[g],
                                 A.aln + g, 0, s1 & 15, q_len, 0, s2 & 15, t_len, 0, cells, rows,
                                 rowx, qg, tg, A.prof);
            }
            __syncthreads();
        }
    }
}

static bool seq_in_lds() {
    static const bool v = getenv("FALCON_AMD_ALIGN_LDS") != nullptr;  // A/B switch; default: L1 path  // (one thread-safe initialisation: several threads ask)
    return v;
}

size_t fa_align_lds_bytes(int max_q_len, int max_t_len) {
    if (!seq_in_lds()) return (2 * RING + 4) * sizeof(u32);
    size_t qw = (size_t)(max_q_len >> 4) + 4, tw = (size_t)(max_t_len >> 4) + 4;
    qw = (qw + 3) & ~(size_t)3;
    tw = (tw + 3) & ~(size_t)3;
    return (qw + tw + 2 * RING + 4) * sizeof(u32);
}

int fa_align_blocks_per_cu(size_t lds_bytes) {
    int nb = 0;
    hipError_t e = seq_in_lds()
        ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_align<true>, 64, lds_bytes)
        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_align<false>, 64, lds_bytes);
    if (e != hipSuccess || nb <= 0) nb = 8;
    return nb;
}

void fa_launch_align_band(const FaBatchDev &b, const FaAlignArena &a, int max_q_len,
                          int max_t_len, double max_diff, int band, hipStream_t s) {
    fa_launch_align_list(b, a, max_q_len, max_t_len, max_diff, band, b.order, b.n_seq, s);
}

// the alignments of the sequences order[0 .. n_work) (device array)
void fa_launch_align_list(const FaBatchDev &b, const FaAlignArena &a, int max_q_len, int max_t_len,
                          double max_diff, int band, const int *order, int n_work, hipStream_t s) {
    if (n_work == 0) return;
    AlignArgs A;
    A.words = b.words; A.seq = b.seq; A.pile = b.pile; A.range = b.range; A.order = order;
    A.n_work = n_work;
    A.counter = a.counter;
    A.cells = a.cells; A.rows = a.rows; A.rowx = a.rowx;
    A.cells_per_slot = a.cells_per_slot; A.rows_per_slot = a.rows_per_slot;
    A.script = b.script; A.script_off = b.script_off; A.aln = b.aln;
    A.band = band;
    size_t qw = (size_t)(max_q_len >> 4) + 4, tw = (size_t)(max_t_len >> 4) + 4;
    qw = (qw + 3) & ~(size_t)3;
    tw = (tw + 3) & ~(size_t)3;
    A.prof = a.prof;
    A.lds_q_words = (int)qw;
    A.lds_t_words = (int)tw;
    A.max_diff = max_diff;
    size_t lds = fa_align_lds_bytes(max_q_len, max_t_len);
    (void)hipMemsetAsync(a.counter, 0, sizeof(int), s);
    int grid = a.n_slot < n_work ? a.n_slot : n_work;
    if (seq_in_lds()) {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute((const void *)k_align<true>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_align<true>, dim3(grid), dim3(64), lds, s, A);
    } else {
        hipLaunchKernelGGL(k_align<false>, dim3(grid), dim3(64), lds, s, A);
    }
}

void fa_launch_align(const FaBatchDev &b, const FaAlignArena &a, int max_q_len, int max_t_len,
                     double max_diff, hipStream_t s) {
    fa_launch_align_band(b, a, max_q_len, max_t_len, max_diff, FA_BAND, s);
}
