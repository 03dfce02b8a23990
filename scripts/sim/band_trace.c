/* band_trace.c -- DEVELOPMENT TOOL (scripts/sim): the band of every row of a banded O(ND)
 * alignment, as k_align2's lanes will see it.  The schedule is the one of the reference's
 * align() (src/c/DW_banded.c:183-243) restated as in oracle/falcon_oracle.c: row d holds the
 * diagonals min_k .. max_k (step 2); after the row the band is cut to the extreme diagonals
 * whose x + y is within `band` of best_m and widened by one on either side.
 *
 * Output per row d: lo[d], hi[d] = the extreme diagonals that passed the filter after row d
 * (the "hull" the kernel keeps).  Returns the number of rows with a hull (the row that ends
 * the alignment has none), or -rows-1 when the alignment gave up (band too wide / rows used up).
 *
 *   gcc -O2 -shared -fPIC -o scripts/sim/libband_trace.so scripts/sim/band_trace.c
 */
#include <limits.h>
#include <stdlib.h>

int band_trace(const char *q, int q_len, const char *t, int t_len, int band, int *lo, int *hi, int cap,
               long *cells_out) {
    int max_d = (int)(0.3 * (q_len + t_len));
    int *V = calloc((size_t)max_d * 2 + 3, sizeof(int));
    int *V0 = V + max_d + 1;
    int best_m = -1, min_k = 0, max_k = 0, done = 0, d;
    long cells = 0;
    for (d = 0; d < max_d && d < cap; d++) {
        if (max_k - min_k > 2 * band) break;
        for (int k = min_k; k <= max_k; k += 2) {
            int x;
            if (k == min_k || (k != max_k && V0[k - 1] < V0[k + 1])) x = V0[k + 1];
            else x = V0[k - 1] + 1;
            int y = x - k;
            while (x < q_len && y < t_len && q[x] == t[y]) { x++; y++; }
            V0[k] = x;
            cells++;
            if (x + y > best_m) best_m = x + y;
            if (x >= q_len || y >= t_len) { done = 1; break; }
        }
        if (done) break;
        int row_lo = INT_MAX, row_hi = INT_MIN;
        for (int k = min_k; k <= max_k; k += 2) {
            int u = 2 * V0[k] - k;
            if (u >= best_m - band) { if (k < row_lo) row_lo = k; if (k > row_hi) row_hi = k; }
        }
        lo[d] = row_lo; hi[d] = row_hi;
        max_k = row_hi + 1;
        min_k = row_lo - 1;
    }
    free(V);
    if (cells_out) *cells_out = cells;
    return done ? d : -d - 1;
}
