/*
 * falcon_amd.h -- C ABI of libfalcon_amd.so, the MI355X (gfx950) implementation of
 * FALCON's pre-assembly consensus ("falcon_sense") hot path.
 *
 * Two surfaces:
 *
 *  (1) LEGACY ABI -- exactly the symbols the reference binds through ctypes in
 *      falcon_kit/falcon_kit.py:54-122 (one CDLL for kup / DWA / falcon), with the
 *      struct layouts of src/c/common.h:57-126.  Loading this library in place of
 *      the reference's ext_falcon .so keeps `import falcon_kit` and
 *      falcon_kit/mains/consensus.py working unchanged.  generate_consensus and
 *      align execute on the GPU (a batch of one); see (2) for the fast path.
 *
 *  (2) BATCH ABI (additive) -- submit many piles at once; sequences are packed
 *      to 2 bits per base in HBM and every stage (k-mer index, k-mer chaining,
 *      banded O(ND) alignment + trace-back, MSA sweep + back-trace) runs as HIP
 *      kernels over the whole batch.  Results are bit-identical to the legacy
 *      entry points.
 *
 * All pointers are plain host pointers; there are no torch / HIP types in any
 * signature.  Functions returning int return 0 on success and a negative code
 * on failure; fa_last_error() describes the last failure of the calling thread.
 * If no usable HIP device is present every entry point fails loudly -- there is
 * no CPU fallback.
 */
#ifndef FALCON_AMD_H
#define FALCON_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------- */
/* (1) Legacy ABI -- replaces /root/reference/src/c/common.h:57-177           */
/* ------------------------------------------------------------------------- */
typedef int seq_coor_t;                       /* common.h:57 */

typedef struct {                              /* common.h:59-69 */
    seq_coor_t aln_str_size;
    seq_coor_t dist;
    seq_coor_t aln_q_s;
    seq_coor_t aln_q_e;
    seq_coor_t aln_t_s;
    seq_coor_t aln_t_e;
    char *q_aln_str;
    char *t_aln_str;
} alignment;

typedef struct {                              /* common.h:95-99 */
    seq_coor_t start;
    seq_coor_t last;
    seq_coor_t count;
} kmer_lookup;

typedef unsigned char base;                   /* common.h:101-104 */
typedef base *seq_array;
typedef seq_coor_t seq_addr;
typedef seq_addr *seq_addr_array;

typedef struct {                              /* common.h:107-111 */
    seq_coor_t count;
    seq_coor_t *query_pos;
    seq_coor_t *target_pos;
} kmer_match;

typedef struct {                              /* common.h:114-120 */
    seq_coor_t s1;
    seq_coor_t e1;
    seq_coor_t s2;
    seq_coor_t e2;
    long int score;
} aln_range;

typedef struct {                              /* common.h:123-126 */
    char *sequence;
    int *eqv;
} consensus_data;

/* kmer_lookup.c:71-119 -- table/array owners (falcon_kit.py:54-66) */
kmer_lookup *allocate_kmer_lookup(seq_coor_t size);
void init_kmer_lookup(kmer_lookup *kl, seq_coor_t size);
void free_kmer_lookup(kmer_lookup *kl);
seq_array allocate_seq(seq_coor_t size);
void init_seq_array(seq_array sa, seq_coor_t size);
void free_seq_array(seq_array sa);
seq_addr_array allocate_seq_addr(seq_coor_t size);
void free_seq_addr_array(seq_addr_array sda);

/* kmer_lookup.c:140-192, :195-204, :207-286, :288 (falcon_kit.py:68-75) */
void add_sequence(seq_coor_t start, unsigned int K, char *seq, seq_coor_t seq_len,
                  seq_addr_array sda, seq_array sa, kmer_lookup *lk);
void mask_k_mer(seq_coor_t size, kmer_lookup *kl, seq_coor_t threshold);
kmer_match *find_kmer_pos_for_seq(char *seq, seq_coor_t seq_len, unsigned int K,
                                  seq_addr_array sda, kmer_lookup *lk);
void free_kmer_match(kmer_match *ptr);

/* kmer_lookup.c:294-427, :429-585, :587 (falcon_kit.py:78-84) */
aln_range *find_best_aln_range(kmer_match *km, seq_coor_t K, seq_coor_t bin_size,
                               seq_coor_t count_th);
aln_range *find_best_aln_range2(kmer_match *km, seq_coor_t K, seq_coor_t bin_width,
                                seq_coor_t count_th);
void free_aln_range(aln_range *r);

/* DW_banded.c:115-337 (falcon_kit.py:111-114) */
alignment *align(char *query_seq, seq_coor_t q_len, char *target_seq, seq_coor_t t_len,
                 seq_coor_t band_tolerance, int get_aln_str);
void free_alignment(alignment *aln);

/* falcon.c:562-666, :776 (falcon_kit.py:119-122, consensus.py:20-23) */
consensus_data *generate_utg_consensus(char **input_seq, seq_coor_t *offset, unsigned int n_seq,
                                       unsigned min_cov, unsigned K, double min_idt); /* falcon.c:668 */
consensus_data *generate_consensus(char **input_seq, unsigned int n_seq, unsigned min_cov,
                                   unsigned K, double min_idt);
void free_consensus_data(consensus_data *c);
/* Additive: generate_consensus has no error channel (falcon.c:562-566 returns a pointer, nothing else), so a
 * pile this library cannot process (a byte outside ACGT, more than 65 534 reads) comes back as the EMPTY
 * consensus -- which the reference's callers drop without a word (consensus.py:286) -- with one line on stderr.
 * This counts those piles for a caller that wants to know; FALCON_AMD_LEGACY_ABORT=1 in the environment makes
 * generate_consensus abort() there instead, like the reference does on its own fatal errors (DW_banded.c:100-113). */
long fa_legacy_failed_piles(void);

/* ------------------------------------------------------------------------- */
/* (2) Batch ABI -- replaces the multiprocessing.Pool.imap over piles of       */
/*     falcon_kit/mains/consensus.py:264-274 and falcon_kit/multiproc.py:28-36 */
/* ------------------------------------------------------------------------- */
typedef struct fa_ctx fa_ctx;     /* one per process and GPU */
typedef struct fa_batch fa_batch; /* a set of piles resident in HBM */

typedef struct {
    /* B_alg = L/4 + 4C + 8D + 16A + 12T + 5O bytes */
    long long L;  /* input bases (all sequences)                          */
    long long C;  /* (d,k) cells evaluated by the banded alignments        */
    long long D;  /* sum of edit distance over accepted alignments         */
    long long A;  /* sum of alignment columns over accepted alignments     */
    long long T;  /* sum of seed lengths                                    */
    long long O;  /* sum of consensus lengths                               */
    long long n_piles, n_seqs, n_aligned;
    /* stage times of the last fa_batch_run, HIP events on the engine stream */
    float ms_index, ms_chain, ms_align, ms_consensus, ms_total;
    int align_slots; /* resident alignment work slots (wavefronts) */
    /* the kernels of the consensus stage (k_msa.hip), same clock */
    float ms_tags, ms_links, ms_score, ms_backtrace;
    /* alignment work slots hold `align_slot_cells` cells each; align_relaunched = the
     * alignments that outgrew them and were done again in worst-case slots */
    long long align_slot_cells;
    int align_relaunched;
    int n_piles_failed; /* piles of the last run without a consensus, see fa_batch_pile_error */
    /* the alignment arena the run used, in bytes (k_align2: tape rings, one-byte cells), and
     * what the two-alignments-per-wavefront kernel reports about itself: iterations with
     * both tracks / one track running, band placements, parkings, alignments handed back */
    long long align_arena_bytes;
    long long align_pair_iterations, align_single_iterations, align_placements, align_parkings,
        align_handed_back, align_wide_rows, align_replacements /* inside the row loop */;
    /* align_handed_back by cause: the alignment's rows did not fit its tape ring; its band stayed
     * wide (rows of more than 60 diagonals) for longer than a waiting neighbour has patience, or
     * grew beyond 191 diagonals; the slot's list of snakes of >= 255 bases was full */
    long long align_handed_back_tape, align_handed_back_wide, align_handed_back_escapes;
} fa_stats;

const char *fa_last_error(void);
int fa_device_count(void);
fa_ctx *fa_create(int device);
/* Optional: pay now what the context's first batch would pay for -- the pinned staging buffer for
 * batches of `batch_bases` bases, the code objects of the path's kernels, the first upload.  The
 * workers call it while their first batches are being read. */
int fa_warm(fa_ctx *ctx, long long batch_bases);
void fa_destroy(fa_ctx *ctx);

/* Stage a batch: n_pile piles, pile p owns pile_n_seq[p] consecutive entries of
 * seqs[]/seq_len[] (first entry = seed).  seq_len may be NULL (strlen is used).
 * Packs the bases to 2 bits each on the host, straight into pinned memory, and uploads them
 * (L/4 bytes over PCIe).  The host strings are not referenced after the call returns. */
fa_batch *fa_batch_create(fa_ctx *ctx, int n_pile, const int *pile_n_seq,
                          const char *const *seqs, const int *seq_len);
/* Run the whole path on the resident batch (may be called repeatedly). */
int fa_batch_run(fa_batch *b, unsigned min_cov, unsigned K, double min_idt);
/* The same in two calls, for callers that keep several (three is enough) batches of one
 * context in flight: fa_batch_submit queues the batch's seed index, chaining and alignment
 * kernels and returns at once -- it never waits for the device (0.06 ms); the rest of the run
 * (the host-side sizing of the consensus stage from the alignment summaries, then its kernels
 * on a stream of their own, beside the kernels of the batches submitted later) is taken care of
 * by a thread the context owns.  fa_batch_wait returns when the batch's results are ready, or
 * reports what went wrong in its consensus stage.  Same results as fa_batch_run, which is
 * submit + wait.  Calls on different batches of a context may come from different threads. */
int fa_batch_submit(fa_batch *b, unsigned min_cov, unsigned K, double min_idt);
int fa_batch_wait(fa_batch *b);
/* Copy results to the host (want_eqv != 0 also copies the eqv arrays). */
int fa_batch_fetch(fa_batch *b, int want_eqv);
/* Consensus of pile p: *seq is NUL terminated and owned by the batch. */
int fa_batch_result(fa_batch *b, int pile, const char **seq, int *len, const int **eqv);
int fa_batch_stats(fa_batch *b, fa_stats *out);
/* Piles fail alone.  The consensus stage handles up to 65534 usable reads per pile (16-bit
 * link counts like the reference's, falcon.c:86; its driver's default --max-n-read is 500); a
 * deeper pile -- or one whose consensus stage reports a device error -- does not fail its
 * batch: fa_batch_run succeeds, fa_batch_result gives that pile an empty consensus, and this
 * returns why (0: the pile is fine; 2: too many usable reads; 1: device error; 3: one of its
 * sequences holds a byte other than upper-case A, C, G, T -- the reference aligns raw
 * characters there, outside the parity domain), with a description in msg if msg != NULL. */
int fa_batch_pile_error(fa_batch *b, int pile, char *msg, int msg_cap);
/* The worker's output rules (falcon_kit/mains/consensus.py:275-299): the FASTA text the
 * reference prints for a consensus string -- nothing if it is shorter than 500; default:
 * ">seed_id" and the longest run of [ACGT] (the last of equally long ones); --output-multi:
 * every run >= 500, at most 10, ">prolog/<seed_id><i>/0_<len>", wrapped at 80 columns;
 * --output-full: ">seed_id_f" and the raw string. */
enum { FA_FASTA_DEFAULT = 0, FA_FASTA_MULTI = 1, FA_FASTA_FULL = 2 };
/* One string (host only, no device needed): up to cap bytes are written to out, the
 * return value is the length of the whole text (< 0: bad arguments). */
long long fa_fasta_records(const char *seed_id, const char *cns, long long len, int mode, char *out,
                           long long cap);
/* All piles of a fetched batch, in pile order, seed_ids[p] NUL-terminated; piles that
 * failed contribute nothing (fa_batch_pile_error / fa_stats.n_piles_failed tell).  *text is
 * owned by the batch and valid until the next call or fa_batch_free. */
int fa_batch_fasta(fa_batch *b, const char *const *seed_ids, int mode, const char **text, long long *len);
void fa_batch_free(fa_batch *b);

/* Diagnostics used by the parity tests: per-sequence stage outputs.
 * g = index into the flattened seqs[] of fa_batch_create. */
int fa_batch_range(fa_batch *b, int g, int *s1, int *e1, int *s2, int *e2, long long *score,
                   int *ok, int *n_hit);
int fa_batch_alignment(fa_batch *b, int g, int *dist, int *q_e, int *t_e, int *size, int *accept,
                       long long *cells);
/* ... and two intermediate lists of the last completed run, for a direct comparison with the
 * reference's (tests/test_gpu_parity.py); no caller of the path needs them.
 * fa_batch_debug_hits: the k-mer hits of sequence g against its pile's seed in the order
 * find_kmer_pos_for_seq reports them (src/c/kmer_lookup.c:207-286: query_pos[], target_pos[]),
 * as the chain stage enumerates them on the device.  Returns the number of hits (min(that,
 * cap) are stored), -1 on error.
 * fa_batch_debug_tags: what get_align_tags (src/c/falcon.c:106-162) yields for the accepted
 * alignment of sequence g, in the device's position-major form: words[t] for the t-th covered
 * seed position -- bit 31: the read deletes the seed base; bits 30..23: length n of the
 * insertion run behind it; bits 22..0: the run, n <= 11: 2 bits per base (0..3 = ACGT), first
 * base lowest, else the index of its first base in ins[] -- plus *lead_word, the word of the
 * position before the alignment's first (an alignment opening with an insertion run; 0: none),
 * and ins[], the alignment's inserted bases in read order.  Returns the number of covered
 * positions (0: the alignment was not accepted), -1 on error. */
/* fa_debug_pack: the host-side 2-bit packer batches are staged with (replaces the ASCII -> code
 * loops of src/c/kmer_lookup.c:159-171, :236-249) on one sequence, no device involved:
 * out[0 .. n_out) = 16 bases per word, base i at bits 2 (i mod 16), A C G T = 0 1 2 3, zero words
 * behind the sequence; returns the position of the first byte other than upper-case A, C, G,
 * T, or -1 (-2: bad arguments). */
int fa_debug_pack(const char *s, int len, unsigned *out, long long n_out);
int fa_batch_debug_hits(fa_batch *b, int g, int *q_pos, int *t_pos, int cap);
int fa_batch_debug_tags(fa_batch *b, int g, unsigned *words, int cap_words, unsigned *lead_word,
                        unsigned char *ins, int cap_ins, int *n_ins);

/* --trim (falcon_kit/mains/consensus.py:48-99 get_alignment): for every read of the
 * batch the window find_best_aln_range2 reports on its pile's seed, k-mers occurring
 * more than mask_threshold times in the seed masked (the driver passes 16).  Results
 * through fa_batch_range(): the raw aln_range (s1, e1, s2, e2, score), n_hit. */
int fa_batch_trim_windows(fa_batch *b, unsigned K, int mask_threshold);

/* Unitig consensus (src/c/falcon.c:668-773 generate_utg_consensus): seqs[0] is the unitig,
 * seqs[j] (NUL terminated) lies at offset[j] on it (negative: starts before it; such
 * entries are rewritten to 0 like the reference does).  Returns a batch holding one pile:
 * fa_batch_fetch / fa_batch_result(b, 0, ...) / fa_batch_free as usual; NULL on error. */
fa_batch *fa_utg_consensus(fa_ctx *ctx, int n_seq, const char *const *seqs, int *offset,
                           double min_idt);

/* Pairwise banded alignment of n independent (query, target) pairs in one
 * launch; out[i] receives a malloc'ed `alignment` (free with free_alignment). */
int fa_align_pairs(fa_ctx *ctx, int n, const char *const *q, const int *q_len,
                   const char *const *t, const int *t_len, int band_tolerance, int get_aln_str,
                   alignment **out);

/* ------------------------------------------------------------------------- */
/* (3) Ingest -- replaces get_seq_data + get_longest_reads of                   */
/*     falcon_kit/mains/consensus.py:161-209 and :26-45: the LA4Falcon "-fo"    */
/*     text stream is parsed, piles are admitted and their reads selected in   */
/*     native code, and handed to fa_batch_create() as pointer arrays.          */
/* ------------------------------------------------------------------------- */
typedef struct fa_reader fa_reader;

/* fd: the stream (e.g. 0); the five numbers are --min-n-read, --min-len-aln,
 * --min-cov-aln, --max-n-read, --max-cov-aln (consensus.py:225-236). */
fa_reader *fa_reader_open(int fd, int min_n_read, int min_len_aln, int min_cov_aln, int max_n_read,
                          int max_cov_aln);
/* Gather the next admitted piles, in stream order, until max_piles piles or max_bases
 * selected bases are reached (<= 0: no limit) or the stream ends.  Returns the number
 * of piles (0: end of stream, < 0: read error, see fa_reader_error).  On return
 *   pile_n_seq[p]            sequences of pile p (seed first), consecutive in
 *   seqs[] / seq_len[]       base pointers (NOT NUL-terminated) and lengths,
 *   seed_ids[p]              NUL-terminated name of the pile's seed (a name holding a
 *                            NUL byte therefore ends there);
 * exactly the arguments of fa_batch_create().  The arrays and what they point to stay
 * valid through ONE further fa_reader_next() on this reader (a thread may stage batch n
 * while another reads batch n + 1), until the call after that or fa_reader_close(). */
int fa_reader_next(fa_reader *r, int max_piles, long long max_bases, const int **pile_n_seq,
                   const char *const **seqs, const int **seq_len, const char *const **seed_ids);
/* How many batches stay valid at a time: what a fa_reader_next() hands out then lives through the
 * next n_batches - 1 calls (default 2, at most 64; only before the first fa_reader_next; 0 or -1).
 * A reader that runs ahead of the staging thread -- a block's text in a file, the devices still
 * being opened -- holds up to n_batches batches of text in memory. */
int fa_reader_keep(fa_reader *r, int n_batches);
const char *fa_reader_error(const fa_reader *r);
void fa_reader_close(fa_reader *r);

#ifdef __cplusplus
}
#endif
#endif
