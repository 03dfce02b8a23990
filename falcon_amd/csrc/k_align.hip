// k_align.hip -- banded O(ND) alignment with trace-back, one wavefront per read.
//
// Restates align() of the reference (src/c/DW_banded.c:115-330) for every
// (read window, seed window) pair of the batch:
//
//   forward  row d of the furthest-reaching table lives on diagonals
//            min_k, min_k+2, .. max_k; lane j of the wave owns diagonal
//            min_k + 2j (rows wider than 64 diagonals take up to 3 passes).
//            All cells of a row depend only on the previous row,
//            so a row is one data-parallel step:
//              - V of the previous row comes from an LDS ring indexed by k>>1,
//                split by parity so consecutive lanes hit consecutive banks;
//              - the snake compares 2-bit packed bases, up to 32 per step, from
//                the read/seed windows staged in LDS (coalesced HBM loads);
//              - best_m is a DPP wave max, the next band comes from one
//                __ballot + ffs/clz per pass, the finishing diagonal ("first k
//                in ascending order", DW_banded.c:220-224) from a ballot;
//            each cell's reached x (+ from_above bit) is streamed to a per-slot
//            HBM arena, one 32-byte record per row keeps min_k, the row offset
//            and the from_above bits.
//   trace    (DW_banded.c:264-319) the diagonal chain k_d is resolved 64 rows
//            at a time from the per-row from_above bits held in registers
//            (v_readlane), then all 64 rows gather their x2 in parallel and emit
//            one u32 per row: (snake_length << 1) | from_above.  The gapped
//            strings of the reference are never materialised; they are a pure
//            function of this edit script and the two sequences.
//
// The reference's qsort + bsearch over (d,k) records (:260,:267-277) and its
// O(max_d * band) calloc (:164) have no counterpart here.
//
// Integer / LDS / HBM-write bound (4 B per DP cell); no MFMA.
#include "fa_device.h"

#define RING 256  // V entries per parity; live band <= 191 diagonals

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
    u64 cells_per_slot;
    u64 rows_per_slot;
    u32 *script;
    const u64 *script_off;
    FaAln *aln;
    int band;
    int lds_q_words;
    int lds_t_words;
    double max_diff;
};

// One alignment.  qL/tL: LDS windows (word aligned), qb/tb: base offset of
// window position 0 inside the first staged word.
__device__ void align_one(const AlignArgs &A, int g, const u32 *qL, int qb, int q_len,
                          const u32 *tL, int tb, int t_len, int *Vring, u32 *cells, FaRowRec *rows,
                          int lane) {
    FaAln res;
    res.dist = 0; res.q_e = 0; res.t_e = 0; res.size = 0; res.accept = 0; res.n_ins = 0;
    res.aligned = 0; res.err = 0; res.cells = 0;

    const int band = A.band;
    const int max_d = (int)(0.3 * (double)(q_len + t_len));  // DW_banded.c:149
    if ((u64)max_d > A.rows_per_slot) {  // host sized the slot from the same formula
        res.err = 1;
        A.aln[g] = res;
        return;
    }
    // zero the ring (reference: calloc'ed V, :153)
    for (int i = lane; i < 2 * RING; i += 64) Vring[i] = 0;

    KTRACE("align g=%d q_len=%d t_len=%d max_d=%d qb=%d tb=%d\n", g, q_len, t_len, max_d, qb, tb);
    int best_m = -1, min_k = 0, max_k = 0;
    u32 row_off = 0;
    int fin_d = -1, fin_k = 0, fin_x = 0, fin_y = 0;

    for (int d = 0; d < max_d; d++) {
        if (max_k - min_k > 2 * band) break;  // :184
        const int n = ((max_k - min_k) >> 1) + 1;
        int *Vcur = Vring + (d & 1) * RING;
        const int *Vprev = Vring + ((d & 1) ^ 1) * RING;
        int ureg[FA_ALIGN_MAXCH];
        u64 dirw[FA_ALIGN_MAXCH];
        int row_max = -1;
        bool finished = false;
#pragma unroll
        for (int c = 0; c < FA_ALIGN_MAXCH; c++) {
            ureg[c] = -1;
            dirw[c] = 0;
            if (c * 64 < n && !finished) {
                const int j = c * 64 + lane;
                const bool act = j < n;
                const int k = min_k + 2 * j;
                const int a = Vprev[((k - 1) >> 1) & (RING - 1)];
                const int b = Vprev[((k + 1) >> 1) & (RING - 1)];
                const bool from_above = (k == min_k) || ((k != max_k) && (a < b));  // :190
                int x = from_above ? b : a + 1;
                int y = x - k;
                if (act) {
                    // snake (:203-206), up to 32 bases per step
                    while (x < q_len && y < t_len) {
                        const int qa = qb + x, ta = tb + y;
                        const u64 diff = fa_window64(qL, qa) ^ fa_window64(tL, ta);
                        int lim = min(32 - (qa & 15), 32 - (ta & 15));
                        lim = min(lim, min(q_len - x, t_len - y));
                        int m = diff ? (__builtin_ctzll(diff) >> 1) : 32;
                        m = min(m, lim);
                        x += m;
                        y += m;
                        if (m < lim) break;
                    }
                    Vcur[(k >> 1) & (RING - 1)] = x;
                    cells[row_off + j] = ((u32)x << 1) | (from_above ? 1u : 0u);
                    ureg[c] = x + y;
                }
                dirw[c] = __ballot(act && from_above);
                const u64 fin = __ballot(act && (x >= q_len || y >= t_len));  // :220
                if (fin) {
                    const int fl = __ffsll((long long)fin) - 1;
                    fin_d = d;
                    fin_k = min_k + 2 * (c * 64 + fl);
                    fin_x = __builtin_amdgcn_readlane(x, fl);
                    fin_y = __builtin_amdgcn_readlane(y, fl);
                    res.cells = (long long)row_off + c * 64 + fl + 1;
                    finished = true;
                }
                row_max = max(row_max, ureg[c]);
            }
        }
        if (lane == 0) {
            FaRowRec rr;
            rr.off = row_off;
            rr.min_k = min_k;
            rr.dir[0] = dirw[0];
            rr.dir[1] = dirw[1];
            rr.dir[2] = dirw[2];
            rows[d] = rr;
        }
        KTRACE(" row d=%d min_k=%d max_k=%d n=%d fin=%d\n", d, min_k, max_k, n, (int)finished);
        if (finished) break;
        row_off += (u32)n;
        best_m = max(best_m, fa_wave_max(row_max));
        // band for the next row (:228-243)
        int jlo = 0x7fffffff, jhi = -1;
#pragma unroll
        for (int c = 0; c < FA_ALIGN_MAXCH; c++) {
            if (c * 64 < n) {
                const u64 in = __ballot(ureg[c] >= 0 && ureg[c] >= best_m - band);
                if (in) {
                    jlo = min(jlo, c * 64 + __ffsll((long long)in) - 1);
                    jhi = max(jhi, c * 64 + 63 - __clzll((long long)in));
                }
            }
        }
        int new_min, new_max;
        if (jhi < 0) {  // unreachable (best_m is attained in the row); reference init values
            new_min = max_k;
            new_max = min_k;
        } else {
            new_min = min_k + 2 * jlo;
            new_max = min_k + 2 * jhi;
        }
        max_k = new_max + 1;
        min_k = new_min - 1;
    }

    if (fin_d < 0) {  // unaligned: aln_str_size stays 0 (:171,:184-186)
        res.cells = row_off;
        A.aln[g] = res;
        return;
    }
    KTRACE(" forward done fin_d=%d fin_k=%d x=%d y=%d\n", fin_d, fin_k, fin_x, fin_y);
    res.aligned = 1;
    res.dist = fin_d;
    res.q_e = fin_x;
    res.t_e = fin_y;
    res.size = (fin_x + fin_y + fin_d) / 2;  // :248

    // ---- trace-back -------------------------------------------------------
    // make this wave's row records / cells visible to its own loads
    __threadfence_block();
    u32 *script = A.script + A.script_off[g];
    int k_cur = fin_k;
    int n_ins = 0;
    for (int hi = fin_d; hi >= 0; hi -= 63) {
        // lanes 0..63 <-> rows hi, hi-1, ...; lane 63 (or the last row above 0)
        // is look-ahead only and is re-done by the next block.
        const int r = hi - lane;
        const bool have = r >= 0;
        FaRowRec rr;
        rr.off = 0; rr.min_k = 0; rr.dir[0] = rr.dir[1] = rr.dir[2] = 0;
        if (have) rr = rows[r];
        const int n_rows = min(64, hi + 1);
        int my_k = 0, my_dir = 0;
        for (int l = 0; l < n_rows; l++) {
            const int mk = __builtin_amdgcn_readlane(rr.min_k, l);
            const int j = (k_cur - mk) >> 1;
            const int wsel = j >> 6;
            u32 lo32, hi32;
            if (wsel == 0) {
                lo32 = __builtin_amdgcn_readlane((int)(u32)rr.dir[0], l);
                hi32 = __builtin_amdgcn_readlane((int)(u32)(rr.dir[0] >> 32), l);
            } else if (wsel == 1) {
                lo32 = __builtin_amdgcn_readlane((int)(u32)rr.dir[1], l);
                hi32 = __builtin_amdgcn_readlane((int)(u32)(rr.dir[1] >> 32), l);
            } else {
                lo32 = __builtin_amdgcn_readlane((int)(u32)rr.dir[2], l);
                hi32 = __builtin_amdgcn_readlane((int)(u32)(rr.dir[2] >> 32), l);
            }
            const u64 wbits = ((u64)hi32 << 32) | lo32;
            const int bit = (int)((wbits >> (j & 63)) & 1ull);
            if (lane == l) {
                my_k = k_cur;
                my_dir = bit;
            }
            if (l < 63 || hi < 63) k_cur += bit ? 1 : -1;  // lane 63's row is redone next block
        }
        // x2 of my row on the path
        int x2 = 0;
        if (have) x2 = (int)(cells[rr.off + (u32)((my_k - rr.min_k) >> 1)] >> 1);
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
    KTRACE(" trace done n_ins=%d\n", n_ins);
    res.n_ins = n_ins;
    res.accept = (res.size > 500) &&
                 ((double)res.dist / (double)res.size < A.max_diff);  // falcon.c:629
    A.aln[g] = res;  // every lane stores the same record
}

__global__ __launch_bounds__(64) void k_align(AlignArgs A) {
    extern __shared__ __attribute__((aligned(16))) u32 smem[];
    u32 *qL = smem;
    u32 *tL = qL + A.lds_q_words;
    int *Vring = (int *)(tL + A.lds_t_words);
    const int lane = fa_lane();
    const int slot = blockIdx.x;
    u32 *cells = A.cells + (u64)slot * A.cells_per_slot;
    FaRowRec *rows = A.rows + (u64)slot * A.rows_per_slot;

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
        const bool too_big = !skip && (qn > A.lds_q_words || tn > A.lds_t_words);
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
            for (int i = lane; i < qn; i += 64) qL[i] = (i < qavail) ? qg[i] : 0u;
            for (int i = lane; i < tn; i += 64) tL[i] = (i < tavail) ? tg[i] : 0u;
            __syncthreads();
            align_one(A, g, qL, s1 & 15, q_len, tL, s2 & 15, t_len, Vring, cells, rows, lane);
            __syncthreads();
        }
    }
}

size_t fa_align_lds_bytes(int max_q_len, int max_t_len) {
    size_t qw = (size_t)(max_q_len >> 4) + 4, tw = (size_t)(max_t_len >> 4) + 4;
    qw = (qw + 3) & ~(size_t)3;
    tw = (tw + 3) & ~(size_t)3;
    return (qw + tw + 2 * RING) * sizeof(u32);
}

int fa_align_blocks_per_cu(size_t lds_bytes) {
    int nb = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_align, 64, lds_bytes);
    if (e != hipSuccess || nb <= 0) nb = 8;
    return nb;
}

void fa_launch_align_band(const FaBatchDev &b, const FaAlignArena &a, int max_q_len,
                          int max_t_len, double max_diff, int band, hipStream_t s) {
    if (b.n_seq == 0) return;
    AlignArgs A;
    A.words = b.words; A.seq = b.seq; A.pile = b.pile; A.range = b.range; A.order = b.order;
    A.n_work = b.n_seq;
    A.counter = a.counter;
    A.cells = a.cells; A.rows = a.rows;
    A.cells_per_slot = a.cells_per_slot; A.rows_per_slot = a.rows_per_slot;
    A.script = b.script; A.script_off = b.script_off; A.aln = b.aln;
    A.band = band;
    size_t qw = (size_t)(max_q_len >> 4) + 4, tw = (size_t)(max_t_len >> 4) + 4;
    qw = (qw + 3) & ~(size_t)3;
    tw = (tw + 3) & ~(size_t)3;
    A.lds_q_words = (int)qw;
    A.lds_t_words = (int)tw;
    A.max_diff = max_diff;
    size_t lds = fa_align_lds_bytes(max_q_len, max_t_len);
    (void)hipMemsetAsync(a.counter, 0, sizeof(int), s);
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute((const void *)k_align, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
    int grid = a.n_slot < b.n_seq ? a.n_slot : b.n_seq;
    hipLaunchKernelGGL(k_align, dim3(grid), dim3(64), lds, s, A);
}

void fa_launch_align(const FaBatchDev &b, const FaAlignArena &a, int max_q_len, int max_t_len,
                     double max_diff, hipStream_t s) {
    fa_launch_align_band(b, a, max_q_len, max_t_len, max_diff, FA_BAND, s);
}
