// k_align_wide.hip -- banded O(ND) alignment for band tolerances beyond the tuned
// kernel's 190: the `align(..., 1500, 1)` calls of contig layout
// (falcon_kit/mains/graph_to_contig.py:52-105, graph_to_utgs.py:19-57).
//
// Same algorithm and outputs as k_align.hip (DW_banded.c:115-330: furthest-reaching rows,
// band trimmed to band_tolerance around the best x+y, first finishing diagonal, trace
// back to an edit script), written for generality, not speed -- these calls are a
// handful per assembly: one wavefront per alignment, rows of up to band + 1 diagonals
// walked 64 at a time, the V table in a 2 x 2048 LDS ring, every cell's x (and its
// from_above bit) in the slot arena; the trace-back follows the path one row at a time
// reading the bit from the cell itself.
#include "fa_device.h"

#define WRING 2048  // V entries per parity: band + 1 <= 1501 live diagonals

struct WideArgs {
    const u32 *words;
    const FaSeq *seq;
    const FaPile *pile;
    const FaRange *range;
    int n_seq;
    u32 *cells;
    FaRowRec *rows;
    u64 cells_per_slot;
    u64 rows_per_slot;
    u32 *script;
    const u64 *script_off;
    FaAln *aln;
    int band;
    double max_diff;
};

__global__ __launch_bounds__(64) void k_align_wide(WideArgs A) {
    __shared__ int Vring[2 * WRING];
    const int lane = fa_lane();
    u32 *cells = A.cells + (u64)blockIdx.x * A.cells_per_slot;
    FaRowRec *rows = A.rows + (u64)blockIdx.x * A.rows_per_slot;
    for (int g = blockIdx.x; g < A.n_seq; g += gridDim.x) {
        const FaSeq sq = A.seq[g];
        const FaRange rg = A.range[g];
        FaAln res;
        res.dist = 0; res.q_e = 0; res.t_e = 0; res.size = 0; res.accept = 0; res.n_ins = 0;
        res.aligned = 0; res.err = 0; res.cells = 0;
        if (sq.idx == 0 || !rg.ok) {
            A.aln[g] = res;
            continue;
        }
        const FaSeq tsq = A.seq[A.pile[sq.pile].first];
        const int q_len = rg.e1 - rg.s1, t_len = rg.e2 - rg.s2;
        const u32 *qL = A.words + sq.woff + (rg.s1 >> 4);
        const u32 *tL = A.words + tsq.woff + (rg.s2 >> 4);
        const int qb = rg.s1 & 15, tb = rg.s2 & 15;
        const int band = A.band;
        const int max_d = (int)(0.3 * (double)(q_len + t_len));  // DW_banded.c:149
        if ((u64)max_d > A.rows_per_slot) {
            res.err = 1;
            A.aln[g] = res;
            continue;
        }
        __syncthreads();
        for (int i = lane; i < 2 * WRING; i += 64) Vring[i] = 0;  // calloc'ed V (:153)
        __syncthreads();

        int best_m = -1, min_k = 0, n = 1;
        u64 row_off = 0;
        int fin_d = -1, fin_k = 0, fin_x = 0, fin_y = 0;
        bool done = false, dead = false;
        int d = 0;
        for (; d < max_d; d++) {                       // :183
            if (n - 1 > band) { dead = true; break; }  // :184
            if (row_off + (u64)n > A.cells_per_slot) { res.err = 1; dead = true; break; }
            const int par = d & 1;
            const int max_k = min_k + 2 * (n - 1);
            int *Vcur = Vring + par * WRING;
            const int *Vprev = Vring + (par ^ 1) * WRING;
            int row_max = -1;
            for (int c0 = 0; c0 < n && !done; c0 += 64) {
                const int j = c0 + lane;
                const bool act = j < n;
                const int k = min_k + 2 * j;
                const int a = Vprev[((k - 1) >> 1) & (WRING - 1)];
                const int b = Vprev[((k + 1) >> 1) & (WRING - 1)];
                const bool from_above = (j == 0) || ((k != max_k) && (a < b));  // :190
                int x = from_above ? b : a + 1;
                int y = x - k;
                snake16(qL, tL, qb, tb, q_len, t_len, act, x, y);
                if (act) {
                    Vcur[(k >> 1) & (WRING - 1)] = x;
                    cells[row_off + (u64)j] = ((u32)x << 1) | (from_above ? 1u : 0u);
                    row_max = max(row_max, x + y);
                }
                const u64 fin = fa_ballot(x >= q_len || y >= t_len) & fa_ballot(j < n);  // :220
                if (fin) {
                    const int fl = __builtin_ctzll(fin);
                    fin_d = d;
                    fin_k = min_k + 2 * (c0 + fl);
                    fin_x = __shfl(x, fl);
                    fin_y = __shfl(y, fl);
                    res.cells = (long long)row_off + c0 + fl + 1;
                    done = true;
                }
            }
            if (lane == 0) {
                FaRowRec rr;
                rr.off = (u32)row_off; rr.min_k = min_k; rr.dlo = (u32)(row_off >> 32); rr.dhi = 0;
                rows[d] = rr;
            }
            if (done) break;
            best_m = max(best_m, fa_wave_max(row_max));
            // next band: the extreme diagonals within `band` of the best x+y (:228-243)
            __syncthreads();
            int jlo = -1, jhi = -1;
            for (int c0 = 0; c0 < n; c0 += 64) {
                const int j = c0 + lane;
                const int k = min_k + 2 * j;
                const int x = Vcur[(k >> 1) & (WRING - 1)];
                const u64 in = fa_ballot(2 * x - k >= best_m - band) & fa_ballot(j < n);
                if (in) {
                    if (jlo < 0) jlo = c0 + __builtin_ctzll(in);
                    jhi = c0 + 63 - __builtin_clzll(in);
                }
            }
            row_off += (u64)n;
            min_k = min_k + 2 * jlo - 1;  // jlo >= 0: best_m is attained inside the row
            n = jhi - jlo + 2;
        }
        if (!done) {  // unaligned: aln_str_size stays 0 (:171,:184-186)
            (void)dead;
            res.cells = (long long)row_off;
            A.aln[g] = res;
            continue;
        }
        res.aligned = 1;
        res.dist = fin_d;
        res.q_e = fin_x;
        res.t_e = fin_y;
        res.size = (fin_x + fin_y + fin_d) / 2;  // :248
        // ---- trace-back (:264-319): one row per step, the direction bit is in the cell
        __threadfence();
        __syncthreads();
        u32 *script = A.script + A.script_off[g];
        int k_cur = fin_k, n_ins = 0;
        int x2 = fin_x;  // x reached on the path in row r
        for (int r = fin_d; r >= 0; r--) {
            const FaRowRec rr = rows[r];
            const u64 off = (u64)rr.off | ((u64)rr.dlo << 32);
            const u32 cell = cells[off + (u64)((k_cur - rr.min_k) >> 1)];
            const int dir = (int)(cell & 1u);
            x2 = (int)(cell >> 1);
            int m;
            if (r == 0) {
                m = x2;  // row 0 starts at (0,0), no edit
                script[0] = (u32)m << 1;
            } else {
                const int k_prev = k_cur + (dir ? 1 : -1);
                const FaRowRec rp = rows[r - 1];
                const u64 offp = (u64)rp.off | ((u64)rp.dlo << 32);
                const int x2p = (int)(cells[offp + (u64)((k_prev - rp.min_k) >> 1)] >> 1);
                const int x1 = dir ? x2p : x2p + 1;
                m = x2 - x1;
                script[r] = ((u32)m << 1) | (u32)dir;
                if (dir == 0) n_ins++;
                k_cur = k_prev;
            }
        }
        res.n_ins = n_ins;
        res.accept = (res.size > 500) && ((double)res.dist / (double)res.size < A.max_diff);
        A.aln[g] = res;  // every lane stores the same record
    }
}

void fa_launch_align_wide(const FaBatchDev &b, const FaAlignArena &a, double max_diff, int band,
                          hipStream_t s) {
    if (b.n_seq == 0) return;
    WideArgs A;
    A.words = b.words; A.seq = b.seq; A.pile = b.pile; A.range = b.range; A.n_seq = b.n_seq;
    A.cells = a.cells; A.rows = a.rows;
    A.cells_per_slot = a.cells_per_slot; A.rows_per_slot = a.rows_per_slot;
    A.script = b.script; A.script_off = b.script_off; A.aln = b.aln;
    A.band = band; A.max_diff = max_diff;
    const int grid = a.n_slot < b.n_seq ? a.n_slot : b.n_seq;
    hipLaunchKernelGGL(k_align_wide, dim3(grid), dim3(64), 0, s, A);
}
