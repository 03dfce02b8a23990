/*
 * falcon_oracle.h -- CPU restatement of FALCON's pre-assembly consensus hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker.  The product path (falcon_amd/csrc) never
 * links, loads or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * below against golden vectors under tests/golden/ that were produced by the
 * compiled reference C (oracle/_ref/falcon_ref.so built from
 * /root/reference/src/c by oracle/Makefile; generator: oracle/gen_golden.py),
 * and tests/test_oracle_vs_ref.py cross-checks against oracle/_ref directly
 * whenever that build is present.
 *
 * The code is written from the behavioural specification in SURVEY.md
 * Appendix A; each function cites the reference file:line whose behaviour it
 * restates.  Data structures are deliberately different from the reference's
 * (CSR k-mer index instead of start/last/next chains, compact per-row trace
 * instead of a sorted (d,k) record array, flat node/link pools instead of a
 * pointer tree) so the restatement is an independent derivation.
 */
#ifndef FALCON_ORACLE_H
#define FALCON_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- k-mer seeding (reference: src/c/kmer_lookup.c) ---- */

typedef struct {
    int count;      /* number of hits */
    int *query_pos; /* ascending query offset (multiples of K/2) */
    int *target_pos;/* ascending seed position inside one query offset */
} fo_hits;

typedef struct {
    int s1, e1, s2, e2;
    long score;
} fo_range;

/* kmer_lookup.c:140-192 (index) + :207-286 (hits), K fixed by caller (8). */
fo_hits *fo_find_hits(const char *seed, int seed_len, const char *query, int query_len, int K);
void fo_free_hits(fo_hits *h);

/* kmer_lookup.c:294-427 */
void fo_best_range(const fo_hits *h, int bin_size, int count_th, fo_range *out);

/* kmer_lookup.c:195-204 (mask) + :429-585 (range2); --trim path only.
 * mask_threshold < 0 disables masking. */
fo_hits *fo_find_hits_masked(const char *seed, int seed_len, const char *query, int query_len,
                             int K, int mask_threshold);
void fo_best_range2(const fo_hits *h, fo_range *out);

/* ---- banded O(ND) alignment (reference: src/c/DW_banded.c:115-330) ---- */

typedef struct {
    int aln_str_size;
    int dist;
    int aln_q_s, aln_q_e, aln_t_s, aln_t_e;
    char *q_aln_str; /* NUL terminated, length aln_str_size (only if want_str) */
    char *t_aln_str;
    long cells;      /* (d,k) cells evaluated == reference d_path_idx at exit */
} fo_alignment;

fo_alignment *fo_align(const char *q, int q_len, const char *t, int t_len, int band_tolerance,
                       int want_str);
void fo_free_alignment(fo_alignment *a);

/* ---- per-pile consensus (reference: src/c/falcon.c:562-666, :308-558) ---- */

typedef struct {
    char *sequence; /* NUL terminated */
    int *eqv;       /* strlen(sequence) ints */
    /* work statistics for SURVEY.md section 8(d) B_alg = L/4+4C+8D+16A+12T+5O */
    long stat_L;    /* sum of input bases, all n_seq strings */
    long stat_C;    /* (d,k) cells evaluated over every align() call */
    long stat_D;    /* sum of dist over accepted alignments */
    long stat_A;    /* sum of aln_str_size over accepted alignments */
    long stat_T;    /* seed length */
    long stat_O;    /* consensus length */
    int n_aligned;  /* accepted alignments */
} fo_consensus;

fo_consensus *fo_generate_consensus(const char **seqs, int n_seq, int min_cov, int K,
                                    double min_idt);
void fo_free_consensus(fo_consensus *c);

#ifdef __cplusplus
}
#endif
#endif
