// k_chain.hip -- k-mer hits of every read against its pile's seed, and the
// best co-linear window (s1,e1,s2,e2) the banded alignment will be run on.
//
// Restates for the whole batch:
//   find_kmer_pos_for_seq  (src/c/kmer_lookup.c:207-286)  -- hits are never
//       materialised: every pass re-enumerates them from the CSR index in the
//       reference's order (query offset ascending, seed position ascending);
//   find_best_aln_range    (src/c/kmer_lookup.c:294-427)  -- diagonal binning,
//       first-fullest bin, +-5 bin filter, reset-at-negative running score;
//   the range sanity filter of generate_consensus (src/c/falcon.c:613-619).
//
// Round-1 mapping: one thread per read (reads are scheduled longest-first so
// the lanes of a wave have similar trip counts); the batch supplies >= 10^5
// independent reads, which is what hides the L2 latency of the CSR walks.
// Integer work, L2/HBM-latency bound; no MFMA.
#include "fa_device.h"

__global__ __launch_bounds__(256) void k_chain(const u32 *__restrict__ words,
                                               const FaSeq *__restrict__ seq,
                                               const FaPile *__restrict__ pile,
                                               const u32 *__restrict__ kidx,
                                               const u32 *__restrict__ kpos,
                                               const int *__restrict__ order,
                                               u32 *__restrict__ bins,
                                               const u64 *__restrict__ bin_off, int n_seq,
                                               FaRange *__restrict__ out) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_seq) return;
    const int g = order[slot];
    const FaSeq sq = seq[g];
    FaRange r;
    r.s1 = r.e1 = r.s2 = r.e2 = 0;
    r.ok = 0;
    r.n_hit = 0;
    r.score = 0;
    if (sq.idx == 0) {  // the seed itself is the target, not a query
        out[g] = r;
        return;
    }
    const FaPile pm = pile[sq.pile];
    const u32 *w = words + sq.woff;
    const u32 *T = kidx + pm.kidx_off;
    const u32 *P = kpos + pm.kpos_off;
    const int n_probe_end = sq.len - FA_K;  // probes at i = 0,4,8,.. while i < len-K (:251-252)
    const int BIN = FA_K * 6;               // falcon.c:602-604
    const int TH = 5;

    // pass 1: diagonal extent.  Buckets are ascending, so the extreme
    // diagonals of a probe come from its first and last entry.
    long long d_min = 0x7fffffffffffffffLL, d_max = -0x7fffffffffffffffLL - 1;
    int n_hit = 0;
    for (int i = 0; i < n_probe_end; i += FA_K / 2) {
        const u32 km = fa_kmer8(w, i);
        const u32 lo = T[km], hi = T[km + 1];
        if (lo == hi) continue;
        n_hit += (int)(hi - lo);
        const long long dl = (long long)i - (long long)P[hi - 1];
        const long long dh = (long long)i - (long long)P[lo];
        if (dl < d_min) d_min = dl;
        if (dh > d_max) d_max = dh;
    }
    r.n_hit = n_hit;
    if (n_hit == 0) {
        out[g] = r;
        return;
    }
    const int n_bin = (int)((d_max - d_min) / BIN) + 1;
    u32 *bc = bins + bin_off[g];
    for (int b = 0; b < n_bin; b++) bc[b] = 0;

    // pass 2: histogram of diagonals (:350-355)
    for (int i = 0; i < n_probe_end; i += FA_K / 2) {
        const u32 km = fa_kmer8(w, i);
        const u32 lo = T[km], hi = T[km + 1];
        for (u32 p = lo; p < hi; p++) {
            const long long d = (long long)i - (long long)P[p];
            bc[(int)((d - d_min) / BIN)]++;
        }
    }
    // pass 3: fullest bin, first maximum in hit order (:360-366)
    long long top_count = 0;
    int top_bin = -1;
    for (int i = 0; i < n_probe_end; i += FA_K / 2) {
        const u32 km = fa_kmer8(w, i);
        const u32 lo = T[km], hi = T[km + 1];
        for (u32 p = lo; p < hi; p++) {
            const long long d = (long long)i - (long long)P[p];
            const int b = (int)((d - d_min) / BIN);
            if ((long long)bc[b] > top_count) {
                top_count = bc[b];
                top_bin = b;
            }
        }
    }
    // pass 4: filter (:369-383) fused with the running-score scan (:385-411)
    int kept = 0;
    if (top_bin >= 0 && top_count > TH) {
        long long run = 0, best = 0;
        int prev_q = 0, start_q = 0, start_t = 0;
        for (int i = 0; i < n_probe_end; i += FA_K / 2) {
            const u32 km = fa_kmer8(w, i);
            const u32 lo = T[km], hi = T[km + 1];
            for (u32 p = lo; p < hi; p++) {
                const int t = (int)P[p];
                const long long d = (long long)i - (long long)t;
                const int b = (int)((d - d_min) / BIN);
                int db = b - top_bin;
                if (db < 0) db = -db;
                if (db > 5) continue;
                if ((int)bc[b] <= TH) continue;
                if (kept == 0) {
                    r.s1 = r.e1 = i;
                    r.s2 = r.e2 = t;
                    start_q = i;
                    start_t = t;
                } else {
                    run += 32 - (i - prev_q);
                    if (run < 0) {
                        run = 0;
                        start_q = i;
                        start_t = t;
                    } else if (run > best) {
                        best = run;
                        r.s1 = start_q;
                        r.s2 = start_t;
                        r.e1 = i;
                        r.e2 = t;
                        r.score = best;
                    }
                }
                prev_q = i;
                kept++;
            }
        }
    }
    if (kept <= 1) {  // :413-419
        r.s1 = r.e1 = r.s2 = r.e2 = 0;
        r.score = 0;
    }
    // sanity filter, falcon.c:613-619 (the 0.5*0.10 product is evaluated in
    // IEEE double exactly like the reference, Q8)
    const int dq = r.e1 - r.s1, dt = r.e2 - r.s2;
    int diff = dq - dt;
    if (diff < 0) diff = -diff;
    const int tol = (int)(0.5 * 0.10 * (double)(dq + dt));
    r.ok = !(dq < 100 || dt < 100 || diff > tol);
    out[g] = r;
}

void fa_launch_chain(const FaBatchDev &b, hipStream_t s) {
    if (b.n_seq == 0) return;
    unsigned grid = (unsigned)((b.n_seq + 255) / 256);
    hipLaunchKernelGGL(k_chain, dim3(grid), dim3(256), 0, s, b.words, b.seq, b.pile, b.kidx,
                       b.kpos, b.order, b.bins, b.bin_off, b.n_seq, b.range);
}
