/*
 * falcon_oracle.c -- CPU restatement of FALCON's falcon_sense hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see falcon_oracle.h).  Parity status: pinned by
 * tests/golden/ vectors generated from the compiled reference (oracle/_ref).
 *
 * Written from the behavioural specification, not from
 * the reference text; every function cites the reference lines it restates.
 * All reference paths are relative to /root/reference.
 *
 * Parity domain: sequences are ACGT only, every read longer
 * than K, fewer than 65535 reads per pile, insertion runs shorter than 248.
 */
#include "falcon_oracle.h"

#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void *xcalloc(size_t n, size_t sz) {
    void *p = calloc(n ? n : 1, sz ? sz : 1);
    if (!p) {
        fprintf(stderr, "falcon_oracle: out of memory (%zu x %zu)\n", n, sz);
        abort();
    }
    return p;
}

static void *xrealloc(void *p, size_t sz) {
    void *r = realloc(p, sz ? sz : 1);
    if (!r) {
        fprintf(stderr, "falcon_oracle: out of memory (%zu)\n", sz);
        abort();
    }
    return r;
}

/* A0 C1 G2 T3 (src/c/kmer_lookup.c:159-171); anything else is outside the
 * parity domain and is folded to 0 here. */
static inline unsigned base2(char c) {
    switch (c) {
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default:  return 0;
    }
}

/* ------------------------------------------------------------------------- */
/* Seed k-mer index.                                                          */
/* Reference: add_sequence (src/c/kmer_lookup.c:140-192) builds, per k-mer,  */
/* a start position plus a "next occurrence" chain in ascending position      */
/* order.  A CSR table (offset[kmer] .. offset[kmer+1]) of ascending          */
/* positions enumerates exactly the same list.  Only positions                */
/* 0 .. seed_len-K-1 are indexed (:174).                                      */
/* ------------------------------------------------------------------------- */
typedef struct {
    int K;
    uint32_t n_kmer;
    int *offset; /* n_kmer + 1 */
    int *pos;    /* indexed positions grouped by k-mer, ascending */
} seed_index;

static uint32_t kmer_at(const char *s, int i, int K) {
    uint32_t v = 0;
    for (int j = 0; j < K; j++) v = (v << 2) | base2(s[i + j]);
    return v;
}

static seed_index *seed_index_build(const char *seed, int seed_len, int K) {
    seed_index *ix = xcalloc(1, sizeof(*ix));
    ix->K = K;
    ix->n_kmer = 1u << (2 * K);
    ix->offset = xcalloc((size_t)ix->n_kmer + 1, sizeof(int));
    int n_pos = seed_len - K; /* number of indexed positions */
    if (n_pos < 0) n_pos = 0;
    ix->pos = xcalloc((size_t)n_pos, sizeof(int));
    for (int i = 0; i < n_pos; i++) ix->offset[kmer_at(seed, i, K) + 1]++;
    for (uint32_t m = 0; m < ix->n_kmer; m++) ix->offset[m + 1] += ix->offset[m];
    int *cursor = xcalloc(ix->n_kmer, sizeof(int));
    for (int i = 0; i < n_pos; i++) {
        uint32_t m = kmer_at(seed, i, K);
        ix->pos[ix->offset[m] + cursor[m]++] = i;
    }
    free(cursor);
    return ix;
}

static void seed_index_free(seed_index *ix) {
    if (!ix) return;
    free(ix->offset);
    free(ix->pos);
    free(ix);
}

/* find_kmer_pos_for_seq (src/c/kmer_lookup.c:207-286): probe the query every
 * K/2 bases while offset < len-K (:251-252); for each probe list every seed
 * occurrence in ascending seed position (:257-282).  mask_threshold >= 0
 * drops k-mers occurring more than that many times (mask_k_mer, :195-204). */
static fo_hits *hits_from_index(const seed_index *ix, const char *query, int query_len,
                                int mask_threshold) {
    fo_hits *h = xcalloc(1, sizeof(*h));
    int K = ix->K, step = K >> 1;
    int cap = 1024;
    h->query_pos = xcalloc(cap, sizeof(int));
    h->target_pos = xcalloc(cap, sizeof(int));
    for (int i = 0; i < query_len - K; i += step) {
        uint32_t m = kmer_at(query, i, K);
        int lo = ix->offset[m], hi = ix->offset[m + 1];
        if (lo == hi) continue;
        if (mask_threshold >= 0 && hi - lo > mask_threshold) continue;
        if (h->count + (hi - lo) > cap) {
            while (h->count + (hi - lo) > cap) cap *= 2;
            h->query_pos = xrealloc(h->query_pos, (size_t)cap * sizeof(int));
            h->target_pos = xrealloc(h->target_pos, (size_t)cap * sizeof(int));
        }
        for (int p = lo; p < hi; p++) {
            h->query_pos[h->count] = i;
            h->target_pos[h->count] = ix->pos[p];
            h->count++;
        }
    }
    return h;
}

fo_hits *fo_find_hits_masked(const char *seed, int seed_len, const char *query, int query_len,
                             int K, int mask_threshold) {
    seed_index *ix = seed_index_build(seed, seed_len, K);
    fo_hits *h = hits_from_index(ix, query, query_len, mask_threshold);
    seed_index_free(ix);
    return h;
}

fo_hits *fo_find_hits(const char *seed, int seed_len, const char *query, int query_len, int K) {
    return fo_find_hits_masked(seed, seed_len, query, query_len, K, -1);
}

void fo_free_hits(fo_hits *h) {
    if (!h) return;
    free(h->query_pos);
    free(h->target_pos);
    free(h);
}

/* find_best_aln_range (src/c/kmer_lookup.c:294-427).
 *  1. diagonal d = q - t, binned as (d - d_min) / bin_size (:350-355);
 *  2. the fullest bin, first maximum in hit order (:360-366);
 *  3. keep hits within +-5 bins of it whose own bin holds more than count_th
 *     hits, only if the fullest bin itself exceeds count_th (:369-383);
 *  4. over the kept hits run score += 32 - (q[i]-q[i-1]), reset below zero,
 *     and report the span of the first strict maximum (:385-411);
 *  5. fewer than two kept hits => all-zero range (:413-419). */
void fo_best_range(const fo_hits *h, int bin_size, int count_th, fo_range *out) {
    memset(out, 0, sizeof(*out));
    int n = h->count;
    if (n <= 0) return;
    long d_min = LONG_MAX, d_max = LONG_MIN;
    for (int i = 0; i < n; i++) {
        long d = (long)h->query_pos[i] - (long)h->target_pos[i];
        if (d < d_min) d_min = d;
        if (d > d_max) d_max = d;
    }
    long n_bin = (d_max - d_min) / bin_size + 1;
    int *bin_count = xcalloc((size_t)n_bin, sizeof(int));
    for (int i = 0; i < n; i++) {
        long d = (long)h->query_pos[i] - (long)h->target_pos[i];
        bin_count[(d - d_min) / bin_size]++;
    }
    long top_count = 0, top_bin = -1;
    for (int i = 0; i < n; i++) {
        long b = ((long)h->query_pos[i] - (long)h->target_pos[i] - d_min) / bin_size;
        if (bin_count[b] > top_count) {
            top_count = bin_count[b];
            top_bin = b;
        }
    }
    int *kq = xcalloc((size_t)n, sizeof(int));
    int *kt = xcalloc((size_t)n, sizeof(int));
    int kept = 0;
    if (top_bin >= 0 && top_count > count_th) {
        for (int i = 0; i < n; i++) {
            long b = ((long)h->query_pos[i] - (long)h->target_pos[i] - d_min) / bin_size;
            if (labs(b - top_bin) > 5) continue;
            if (bin_count[b] > count_th) {
                kq[kept] = h->query_pos[i];
                kt[kept] = h->target_pos[i];
                kept++;
            }
        }
    }
    if (kept > 1) {
        out->s1 = out->e1 = kq[0];
        out->s2 = out->e2 = kt[0];
        long run = 0, best = 0;
        int run_start = 0;
        for (int i = 1; i < kept; i++) {
            run += 32 - (kq[i] - kq[i - 1]);
            if (run < 0) {
                run = 0;
                run_start = i;
            } else if (run > best) {
                best = run;
                out->s1 = kq[run_start];
                out->s2 = kt[run_start];
                out->e1 = kq[i];
                out->e2 = kt[i];
                out->score = best;
            }
        }
    }
    free(bin_count);
    free(kq);
    free(kt);
}

static int cmp_int(const void *a, const void *b) {
    int x = *(const int *)a, y = *(const int *)b;
    return (x > y) - (x < y);
}

/* find_best_aln_range2 (src/c/kmer_lookup.c:429-585), used only by --trim.
 *  - widest window of sorted diagonals within delta = (int)(0.05*(max_q+max_t))
 *    (:462-488), where max_t carries the reference's quirk of being reset to
 *    max_q whenever it already exceeds the current hit (:458);
 *  - fewer than 32 hits in the window => zero range (:490-498);
 *  - chain hits inside the diagonal window: predecessor = closest earlier hit
 *    (Manhattan gap, strict <) with smaller t, q gap <= 320, t gap <= 320
 *    (:519-538); score += 64 - gap, floored at 0 (:539-551);
 *  - report the chain ending at the first strict maximum (:552-583). */
void fo_best_range2(const fo_hits *h, fo_range *out) {
    memset(out, 0, sizeof(*out));
    int n = h->count;
    if (n <= 0) return;
    int *ds = xcalloc((size_t)n, sizeof(int));
    int max_q = -1, max_t = -1;
    for (int i = 0; i < n; i++) {
        ds[i] = h->query_pos[i] - h->target_pos[i];
        if (h->query_pos[i] > max_q) max_q = h->query_pos[i];
        max_t = (max_t > h->target_pos[i]) ? max_q : h->target_pos[i];
    }
    qsort(ds, (size_t)n, sizeof(int), cmp_int);
    int delta = (int)(long)(0.05 * (max_q + max_t));
    int s = 0, e = 0, best_s = -1, best_e = -1, best_span = -1;
    for (;;) {
        int d_s = ds[s], d_e = ds[e];
        while (d_e < d_s + delta && e < n - 1) {
            e++;
            d_e = ds[e];
        }
        if (best_span == -1 || e - s > best_span) {
            best_span = e - s;
            best_s = s;
            best_e = e;
        }
        s++;
        if (s == n || e == n) break;
    }
    if (best_s == -1 || best_e == -1 || best_e - best_s < 32) {
        free(ds);
        return;
    }
    int lo = ds[best_s], hi = ds[best_e];
    free(ds);
    int *prev = xcalloc((size_t)n, sizeof(int));
    int *score = xcalloc((size_t)n, sizeof(int));
    int *links = xcalloc((size_t)n, sizeof(int));
    for (int i = 0; i < n; i++) prev[i] = -1;
    int top = -1, top_score = 0, top_links = 0;
    for (int i = 0; i < n; i++) {
        int cx = h->query_pos[i], cy = h->target_pos[i];
        int d = cx - cy;
        if (d < lo || d > hi) continue;
        int cand = -1, gap = 65535;
        for (int j = i - 1; j >= 0; j--) {
            int px = h->query_pos[j], py = h->target_pos[j];
            int pd = px - py;
            if (pd < lo || pd > hi) continue;
            if (cx - px > 320) break;
            if (cy > py && cx - px + cy - py < gap && cy - py <= 320) {
                gap = cx - px + cy - py;
                cand = j;
            }
        }
        if (cand != -1) {
            prev[i] = cand;
            score[i] = score[cand] + (64 - gap);
            links[i] = links[cand] + 1;
            if (score[i] < 0) {
                score[i] = 0;
                links[i] = 0;
            }
        }
        if (score[i] > top_score) {
            top_score = score[i];
            top_links = links[i];
            top = i;
        }
    }
    if (top != -1) {
        out->score = top_links + 1;
        out->e1 = h->query_pos[top];
        out->e2 = h->target_pos[top];
        int i = top;
        while (prev[i] != -1) i = prev[i];
        out->s1 = h->query_pos[i];
        out->s2 = h->target_pos[i];
    }
    free(prev);
    free(score);
    free(links);
}

/* ------------------------------------------------------------------------- */
/* Banded O(ND) alignment, src/c/DW_banded.c:115-330.                         */
/*                                                                            */
/* Row d of the furthest-reaching table is evaluated on diagonals             */
/* min_k, min_k+2, .. max_k (:188).  A cell takes its start from the diagonal */
/* above (k+1, a target-only step) when it is the lowest diagonal of the row, */
/* or when it is not the highest and V[k-1] < V[k+1]; otherwise from k-1 plus */
/* one (a query-only step) (:190-196).  The snake then runs while bases match */
/* (:203-206).  The first cell (ascending k) to touch either sequence end     */
/* finishes the alignment (:220-224).  After each row the band is trimmed to  */
/* the extreme diagonals whose x+y is within band_tolerance of the best x+y   */
/* seen so far, widened by one on each side (:228-243).  The search gives up  */
/* when the band grows past 2*band_tolerance (:184) or after                  */
/* max_d = (int)(0.3*(q_len+t_len)) rows (:149,183).                          */
/*                                                                            */
/* Instead of the reference's (d,k,x1,y1,x2,y2,pre_k) records + qsort/bsearch */
/* (:198-211,260-277) each row keeps its min_k and, per cell, the reached x   */
/* with one bit for the step direction; x1 is recovered from the previous     */
/* row.  The cell count equals the reference's d_path_idx.                    */
/* ------------------------------------------------------------------------- */
typedef struct {
    int min_k;
    int n;      /* cells in the row */
    long start; /* index of the first cell in the cell pool */
} trace_row;

fo_alignment *fo_align(const char *q, int q_len, const char *t, int t_len, int band_tolerance,
                       int want_str) {
    fo_alignment *a = xcalloc(1, sizeof(*a));
    a->q_aln_str = xcalloc((size_t)q_len + (size_t)t_len + 1, 1);
    a->t_aln_str = xcalloc((size_t)q_len + (size_t)t_len + 1, 1);

    int max_d = (int)(0.3 * (q_len + t_len));
    int band_size = band_tolerance * 2;
    if ((size_t)INT_MAX < (size_t)max_d * (size_t)(band_size + 1) * 2ULL) {
        /* the reference abort()s here (:158-161) */
        fprintf(stderr, "falcon_oracle: q_len=%d t_len=%d band=%d too big\n", q_len, t_len,
                band_size);
        abort();
    }
    int *V = xcalloc((size_t)max_d * 2 + 3, sizeof(int));
    int *V0 = V + max_d + 1; /* V0[k], k in [-max_d-1, max_d+1] */
    trace_row *rows = xcalloc((size_t)max_d + 1, sizeof(trace_row));
    long cell_cap = 4096, n_cell = 0;
    uint32_t *cells = xcalloc((size_t)cell_cap, sizeof(uint32_t));

    int best_m = -1, min_k = 0, max_k = 0;
    int done = 0, fin_d = 0, fin_k = 0, fin_x = 0, fin_y = 0;
    for (int d = 0; d < max_d && !done; d++) {
        if (max_k - min_k > band_size) break;
        int width = (max_k - min_k) / 2 + 1;
        if (n_cell + width > cell_cap) {
            while (n_cell + width > cell_cap) cell_cap *= 2;
            cells = xrealloc(cells, (size_t)cell_cap * sizeof(uint32_t));
        }
        rows[d].min_k = min_k;
        rows[d].start = n_cell;
        rows[d].n = 0;
        int row_lo = INT_MAX, row_hi = INT_MIN;
        /* U[k] (x+y per diagonal) of this row, kept beside V for the band */
        for (int k = min_k; k <= max_k; k += 2) {
            int x, from_above;
            if (k == min_k || (k != max_k && V0[k - 1] < V0[k + 1])) {
                x = V0[k + 1];
                from_above = 1;
            } else {
                x = V0[k - 1] + 1;
                from_above = 0;
            }
            int y = x - k;
            while (x < q_len && y < t_len && q[x] == t[y]) {
                x++;
                y++;
            }
            cells[n_cell++] = ((uint32_t)x << 1) | (uint32_t)from_above;
            rows[d].n++;
            V0[k] = x;
            if (x + y > best_m) best_m = x + y;
            if (x >= q_len || y >= t_len) {
                done = 1;
                fin_d = d;
                fin_k = k;
                fin_x = x;
                fin_y = y;
                break;
            }
        }
        if (done) break;
        /* band for the next row: every k of this row was just written */
        for (int k = min_k; k <= max_k; k += 2) {
            int u = 2 * V0[k] - k; /* x + y */
            if (u >= best_m - band_tolerance) {
                if (k < row_lo) row_lo = k;
                if (k > row_hi) row_hi = k;
            }
        }
        if (row_lo == INT_MAX) { /* cannot happen: best_m is attained in the row */
            row_lo = max_k;
            row_hi = min_k;
        }
        max_k = row_hi + 1;
        min_k = row_lo - 1;
    }
    a->cells = n_cell;

    if (done) {
        a->aln_q_e = fin_x;
        a->aln_t_e = fin_y;
        a->dist = fin_d;
        a->aln_str_size = (fin_x + fin_y + fin_d) / 2;
        if (want_str > 0) {
            /* walk back to row 0 collecting the diagonal of every row */
            int *path_k = xcalloc((size_t)fin_d + 1, sizeof(int));
            int k = fin_k;
            for (int d = fin_d; d >= 0; d--) {
                path_k[d] = k;
                uint32_t c = cells[rows[d].start + (k - rows[d].min_k) / 2];
                k = (c & 1u) ? k + 1 : k - 1;
            }
            int pos = 0, x = 0, y = 0;
            for (int d = 0; d <= fin_d; d++) {
                uint32_t c = cells[rows[d].start + (path_k[d] - rows[d].min_k) / 2];
                int x2 = (int)(c >> 1);
                if (d > 0) {
                    if (c & 1u) { /* target-only step */
                        a->q_aln_str[pos] = '-';
                        a->t_aln_str[pos] = t[y];
                        y++;
                    } else { /* query-only step */
                        a->q_aln_str[pos] = q[x];
                        a->t_aln_str[pos] = '-';
                        x++;
                    }
                    pos++;
                }
                while (x < x2) {
                    a->q_aln_str[pos] = q[x];
                    a->t_aln_str[pos] = t[y];
                    x++;
                    y++;
                    pos++;
                }
            }
            a->aln_str_size = pos;
            free(path_k);
        }
    }
    free(V);
    free(rows);
    free(cells);
    return a;
}

void fo_free_alignment(fo_alignment *a) {
    if (!a) return;
    free(a->q_aln_str);
    free(a->t_aln_str);
    free(a);
}

/* ------------------------------------------------------------------------- */
/* Alignment tags, src/c/falcon.c:106-162.                                    */
/* Column k of a gapped alignment becomes (t_pos, delta, q_base) plus the     */
/* same triple of the previous column; delta counts query bases since the     */
/* last target base.  Tagging stops at the first column whose delta (or whose */
/* predecessor's delta) reaches 255 (:138-152).                               */
/* ------------------------------------------------------------------------- */
typedef struct {
    int t_pos;
    int p_t_pos;
    uint8_t delta;
    uint8_t p_delta;
    char q_base;
    char p_q_base;
} tag_t;

typedef struct {
    int len;
    tag_t *tags;
} tag_list;

static tag_list make_tags(const char *q_aln, const char *t_aln, int aln_len, int s1, int s2,
                          int t_offset) {
    tag_list tl;
    tl.tags = xcalloc((size_t)aln_len + 1, sizeof(tag_t));
    int j = s2 - 1, jj = 0, p_j = -1, p_jj = 0;
    char p_base = '.';
    int k;
    (void)s1;
    for (k = 0; k < aln_len; k++) {
        if (q_aln[k] != '-') jj++;
        if (t_aln[k] != '-') {
            j++;
            jj = 0;
        }
        if (!(j + t_offset >= 0 && jj < 255 && p_jj < 255)) break;
        tag_t *g = &tl.tags[k];
        g->t_pos = j + t_offset;
        g->delta = (uint8_t)jj;
        g->q_base = q_aln[k];
        g->p_t_pos = p_j + t_offset;
        g->p_delta = (uint8_t)p_jj;
        g->p_q_base = p_base;
        p_j = j;
        p_jj = jj;
        p_base = q_aln[k];
    }
    tl.len = k;
    return tl;
}

static inline int base5(char c) {
    switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    case '-': return 4;
    default:  return -1;
    }
}

/* ------------------------------------------------------------------------- */
/* MSA graph + best path, src/c/falcon.c:308-558.                             */
/*                                                                            */
/* Nodes are (t_pos, delta, base in ACGT-).  Every tag adds one to the link   */
/* (its node <- previous column's node); links of a node are kept in the      */
/* order they first appear (update_col, :232-263).  coverage[t_pos] counts    */
/* delta==0 tags (:357-360).                                                  */
/* Scores (kept here in half units, exactly representable, Q6):               */
/*   link score = score(prev node) + link_count - coverage[t_pos]/2           */
/*   (prev node missing, p_t_pos == -1: just link_count - coverage/2)         */
/*   node score = max over links, strictly greater than -1 wins, first max in */
/*   link order (:420-462); a node with no link above -1 keeps -1 and the     */
/*   zero back pointer (0,0,A) (Q4).                                          */
/* Global best = first node in (t_pos, delta, base) order with a strictly     */
/* greater score (:464-469); it remembers the *link index* of its best link   */
/* (Q2), which the back-trace then uses as if it were a base code (:494-515). */
/* Back-trace (:494-528): emit the pending base (upper case iff               */
/* coverage[t_pos] > min_cov), unless the node's back pointer has t_pos -1    */
/* (so the first column is never emitted, Q1) or 2*t_len characters are out;  */
/* '-' nodes emit nothing; eqv = (int)score - (int)prev score.                */
/* ------------------------------------------------------------------------- */
typedef struct {
    int n_link;
    int link0;    /* first slot in the link pool */
    int best_t;   /* back pointer: t_pos, delta, base */
    int best_d;
    int best_b;
    int score_h;  /* node score in half units */
} node_t;

typedef struct {
    int p_t;
    int p_d;
    int p_b_char; /* raw previous base character */
    int count;
} link_t;

static fo_consensus *consensus_from_tags(tag_list *tls, int n_tl, int t_len, int min_cov) {
    unsigned *coverage = xcalloc((size_t)t_len + 1, sizeof(unsigned));
    int *max_delta = xcalloc((size_t)t_len + 1, sizeof(int));
    int *level0 = xcalloc((size_t)t_len + 2, sizeof(int));

    /* pass 1: coverage, deepest insertion level per position */
    int cur_t = 0; /* persists across alignments like the reference's t_pos (Q11) */
    for (int i = 0; i < n_tl; i++) {
        for (int j = 0; j < tls[i].len; j++) {
            const tag_t *g = &tls[i].tags[j];
            if (g->delta == 0) {
                cur_t = g->t_pos;
                coverage[cur_t]++;
            }
            if (g->delta > max_delta[cur_t]) max_delta[cur_t] = g->delta;
        }
    }
    for (int t = 0; t < t_len; t++) level0[t + 1] = level0[t] + max_delta[t] + 1;
    int n_level = level0[t_len];
    size_t n_node = (size_t)n_level * 5;
    node_t *nodes = xcalloc(n_node, sizeof(node_t));

    /* pass 2: tags per node = upper bound of its link slots */
    cur_t = 0;
    for (int i = 0; i < n_tl; i++) {
        for (int j = 0; j < tls[i].len; j++) {
            const tag_t *g = &tls[i].tags[j];
            if (g->delta == 0) cur_t = g->t_pos;
            int b = base5(g->q_base);
            if (b < 0) continue; /* outside the parity domain */
            nodes[(size_t)(level0[cur_t] + g->delta) * 5 + b].link0++;
        }
    }
    size_t n_slot = 0;
    for (size_t v = 0; v < n_node; v++) {
        int c = nodes[v].link0;
        nodes[v].link0 = (int)n_slot;
        n_slot += (size_t)c;
    }
    link_t *links = xcalloc(n_slot, sizeof(link_t));

    /* pass 3: accumulate links in first-seen order */
    cur_t = 0;
    for (int i = 0; i < n_tl; i++) {
        for (int j = 0; j < tls[i].len; j++) {
            const tag_t *g = &tls[i].tags[j];
            if (g->delta == 0) cur_t = g->t_pos;
            int b = base5(g->q_base);
            if (b < 0) continue;
            node_t *nd = &nodes[(size_t)(level0[cur_t] + g->delta) * 5 + b];
            link_t *lk = links + nd->link0;
            int kk;
            for (kk = 0; kk < nd->n_link; kk++) {
                if (lk[kk].p_t == g->p_t_pos && lk[kk].p_d == g->p_delta &&
                    lk[kk].p_b_char == g->p_q_base) {
                    lk[kk].count++;
                    break;
                }
            }
            if (kk == nd->n_link) {
                lk[kk].p_t = g->p_t_pos;
                lk[kk].p_d = g->p_delta;
                lk[kk].p_b_char = g->p_q_base;
                lk[kk].count = 1;
                nd->n_link++;
            }
        }
    }

    /* forward scoring */
    const int FLOOR_H = -2; /* -1.0 in half units */
    int g_best_h = FLOOR_H, g_best_ck = 0, g_best_t = 0, have_best = 0;
    size_t g_best_node = 0;
    int last_ck = -1; /* the reference does not reset best_ck per node */
    for (int t = 0; t < t_len; t++) {
        for (int dl = 0; dl <= max_delta[t]; dl++) {
            for (int b = 0; b < 5; b++) {
                size_t v = (size_t)(level0[t] + dl) * 5 + b;
                node_t *nd = &nodes[v];
                const link_t *lk = links + nd->link0;
                int best_h = FLOOR_H;
                for (int ck = 0; ck < nd->n_link; ck++) {
                    int pb = base5((char)lk[ck].p_b_char);
                    if (pb < 0) pb = 4;
                    int h;
                    if (lk[ck].p_t == -1) {
                        h = 2 * lk[ck].count - (int)coverage[t];
                    } else {
                        size_t pv = (size_t)(level0[lk[ck].p_t] + lk[ck].p_d) * 5 + pb;
                        h = nodes[pv].score_h + 2 * lk[ck].count - (int)coverage[t];
                    }
                    if (h > best_h) {
                        best_h = h;
                        nd->best_t = lk[ck].p_t;
                        nd->best_d = lk[ck].p_d;
                        nd->best_b = pb;
                        last_ck = ck;
                    }
                }
                nd->score_h = best_h;
                if (best_h > g_best_h) {
                    g_best_h = best_h;
                    g_best_node = v;
                    g_best_ck = last_ck;
                    g_best_t = t;
                    have_best = 1;
                }
            }
        }
    }

    fo_consensus *c = xcalloc(1, sizeof(*c));
    c->sequence = xcalloc((size_t)t_len * 2 + 1, 1);
    c->eqv = xcalloc((size_t)t_len * 2 + 1, sizeof(int));
    if (have_best) { /* the reference assert()s otherwise (:476) */
        unsigned index = 0;
        int ck = g_best_ck;
        int i = g_best_t;
        char bb = '$';
        const node_t *nd = &nodes[g_best_node];
        for (;;) {
            int upper = coverage[i] > (unsigned)min_cov;
            switch (ck) {
            case 0: bb = upper ? 'A' : 'a'; break;
            case 1: bb = upper ? 'C' : 'c'; break;
            case 2: bb = upper ? 'G' : 'g'; break;
            case 3: bb = upper ? 'T' : 't'; break;
            case 4: bb = '-'; break;
            default: break; /* keeps the previous character (Q2) */
            }
            int score0_h = nd->score_h;
            i = nd->best_t;
            if (i == -1 || index >= (unsigned)t_len * 2) break;
            ck = nd->best_b;
            nd = &nodes[(size_t)(level0[i] + nd->best_d) * 5 + ck];
            if (bb != '-') {
                c->sequence[index] = bb;
                c->eqv[index] = score0_h / 2 - nd->score_h / 2; /* (int) truncation, Q6 */
                index++;
            }
        }
        for (unsigned a = 0, z = index; a + 1 < z; a++) { /* reverse in place */
            z--;
            char tc = c->sequence[a];
            c->sequence[a] = c->sequence[z];
            c->sequence[z] = tc;
            int te = c->eqv[a];
            c->eqv[a] = c->eqv[z];
            c->eqv[z] = te;
        }
        c->sequence[index] = 0;
    }
    free(coverage);
    free(max_delta);
    free(level0);
    free(nodes);
    free(links);
    return c;
}

/* generate_consensus, src/c/falcon.c:562-666.
 * seqs[0] is the seed (target); every other string is aligned against it:
 * hits (K, probes every K/2) -> fo_best_range(bin K*6, threshold 5) (:602-604)
 * -> range sanity filter (:613-619) -> fo_align on the window, band 150
 * (:624-628) -> accepted when longer than 500 columns with
 * dist/columns < 1-min_idt (:580,629) -> tags with query id j (:630-634). */
fo_consensus *fo_generate_consensus(const char **seqs, int n_seq, int min_cov, int K,
                                    double min_idt) {
    double max_diff = 1.0 - min_idt;
    int t_len = (int)strlen(seqs[0]);
    seed_index *ix = seed_index_build(seqs[0], t_len, K);
    tag_list *tls = xcalloc((size_t)n_seq, sizeof(tag_list));
    int n_tl = 0;
    long stat_L = t_len, stat_C = 0, stat_D = 0, stat_A = 0;

    for (int j = 1; j < n_seq; j++) {
        int q_len = (int)strlen(seqs[j]);
        stat_L += q_len;
        fo_hits *h = hits_from_index(ix, seqs[j], q_len, -1);
        fo_range r;
        fo_best_range(h, K * 6, 5, &r);
        fo_free_hits(h);
        int dq = r.e1 - r.s1, dt = r.e2 - r.s2;
        if (dq < 100 || dt < 100 || abs(dq - dt) > (int)(0.5 * 0.10 * (dq + dt))) continue;
        fo_alignment *a = fo_align(seqs[j] + r.s1, dq, seqs[0] + r.s2, dt, 150, 1);
        stat_C += a->cells;
        if (a->aln_str_size > 500 && (double)a->dist / (double)a->aln_str_size < max_diff) {
            tls[n_tl++] = make_tags(a->q_aln_str, a->t_aln_str, a->aln_str_size, r.s1, r.s2, 0);
            stat_D += a->dist;
            stat_A += a->aln_str_size;
        }
        fo_free_alignment(a);
    }

    fo_consensus *c;
    if (n_tl > 0) {
        c = consensus_from_tags(tls, n_tl, t_len, min_cov);
    } else {
        c = xcalloc(1, sizeof(*c));
        c->sequence = xcalloc(1, 1);
        c->eqv = xcalloc(1, sizeof(int));
    }
    c->stat_L = stat_L;
    c->stat_C = stat_C;
    c->stat_D = stat_D;
    c->stat_A = stat_A;
    c->stat_T = t_len;
    c->stat_O = (long)strlen(c->sequence);
    c->n_aligned = n_tl;
    for (int i = 0; i < n_tl; i++) free(tls[i].tags);
    free(tls);
    seed_index_free(ix);
    return c;
}

void fo_free_consensus(fo_consensus *c) {
    if (!c) return;
    free(c->sequence);
    free(c->eqv);
    free(c);
}
