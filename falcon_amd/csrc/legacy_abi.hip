// legacy_abi.hip -- the 19 symbols falcon_kit/falcon_kit.py:54-122 binds, so this
// library can stand in for the reference's ext_falcon shared object.
//
//  * generate_consensus / align run on the GPU through the batch engine
//    (a batch of one); there is no CPU implementation of either in this library
//    and both abort() loudly (like the reference's my_calloc/abort discipline,
//    DW_banded.c:100-113) when no HIP device is usable.
//  * the kup table functions (allocate_* / add_sequence / mask_k_mer /
//    find_kmer_pos_for_seq / find_best_aln_range[2]) operate on caller-owned
//    host structs whose layout the ABI fixes (common.h:95-120); they exist for
//    ABI completeness of the --trim and graph_to_contig callers (SURVEY.md 8f-1,
//    8f-3) and are NOT on the falcon_sense hot path: the hot path never builds
//    these host tables (its equivalents are k_seed_index / k_chain on the GPU).
#include "../../include/falcon_amd.h"

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <vector>

static fa_ctx *g_ctx = nullptr;
static std::mutex g_mu;

static fa_ctx *default_ctx() {
    if (!g_ctx) {
        int dev = 0;
        if (const char *e = getenv("FALCON_AMD_DEVICE")) dev = atoi(e);
        g_ctx = fa_create(dev);
        if (!g_ctx) {
            fprintf(stderr, "CRITICAL ERROR: falcon_amd: %s\n", fa_last_error());
            abort();
        }
    }
    return g_ctx;
}

static void die(const char *what) {
    fprintf(stderr, "CRITICAL ERROR: falcon_amd %s: %s\n", what, fa_last_error());
    abort();
}

// piles generate_consensus answered with the empty consensus because this library could not process them
static std::atomic<long> g_failed_piles{0};
extern "C" long fa_legacy_failed_piles(void) { return g_failed_piles.load(); }

// ---- falcon.c:562-666 ------------------------------------------------------
extern "C" consensus_data *generate_consensus(char **input_seq, unsigned int n_seq,
                                              unsigned min_cov, unsigned K, double min_idt) {
    std::lock_guard<std::mutex> lk(g_mu);
    fa_ctx *c = default_ctx();
    int n = (int)n_seq;
    fa_batch *b = fa_batch_create(c, 1, &n, (const char *const *)input_seq, nullptr);
    if (!b) die("generate_consensus(stage)");
    if (fa_batch_run(b, min_cov, K, min_idt)) die("generate_consensus(run)");
    if (fa_batch_fetch(b, 1)) die("generate_consensus(fetch)");
    const char *s = nullptr;
    const int *e = nullptr;
    int len = 0;
    if (fa_batch_result(b, 0, &s, &len, &e)) die("generate_consensus(result)");
    char why[256];
    if (fa_batch_pile_error(b, 0, why, (int)sizeof(why)) > 0) {
        // A pile this library cannot process (a byte outside ACGT, more than 65 534 reads): this
        // ABI has no error channel, and the reference has no such limit -- but it aborts only
        // when an allocation fails (DW_banded.c:100-113), and a ctypes caller
        // (falcon_kit/mains/consensus.py:102-120, graph_to_contig.py:58-104) must keep its
        // interpreter.  So: one loud line on stderr and the EMPTY consensus, which every caller
        // of the reference already handles (falcon.c:651-656; consensus.py:286 drops it).
        // (FALCON_AMD_LEGACY_ABORT=1: a caller that would rather stop than lose a read silently gets the
        // abort of rounds 1-4 back; fa_legacy_failed_piles() counts the piles answered this way)
        g_failed_piles++;
        if (getenv("FALCON_AMD_LEGACY_ABORT")) {
            fprintf(stderr, "CRITICAL ERROR: falcon_amd generate_consensus: %s\n", why);
            abort();
        }
        fprintf(stderr, "ERROR: falcon_amd generate_consensus: %s -- empty consensus returned\n", why);
        len = 0;
    }
    consensus_data *r = (consensus_data *)calloc(1, sizeof(consensus_data));
    r->sequence = (char *)calloc((size_t)len + 1, 1);
    r->eqv = (int *)calloc((size_t)len + 1, sizeof(int));
    if (len > 0) {
        memcpy(r->sequence, s, (size_t)len);
        memcpy(r->eqv, e, (size_t)len * sizeof(int));
    }
    fa_batch_free(b);
    return r;
}

// ---- falcon.c:668-773 (exported by the reference's library, bound by nobody) --------
extern "C" consensus_data *generate_utg_consensus(char **input_seq, seq_coor_t *offset,
                                                  unsigned int n_seq, unsigned min_cov, unsigned K,
                                                  double min_idt) {
    (void)min_cov;  // the reference ignores it too: get_cns_from_align_tags(..., 0), :755
    (void)K;        // no k-mer stage on this path
    std::lock_guard<std::mutex> lk(g_mu);
    fa_ctx *c = default_ctx();
    fa_batch *b = fa_utg_consensus(c, (int)n_seq, (const char *const *)input_seq, offset, min_idt);
    if (!b) die("generate_utg_consensus(run)");
    if (fa_batch_fetch(b, 1)) die("generate_utg_consensus(fetch)");
    const char *s = nullptr;
    const int *e = nullptr;
    int len = 0;
    if (fa_batch_result(b, 0, &s, &len, &e)) die("generate_utg_consensus(result)");
    consensus_data *r = (consensus_data *)calloc(1, sizeof(consensus_data));
    r->sequence = (char *)calloc((size_t)len + 1, 1);
    r->eqv = (int *)calloc((size_t)len + 1, sizeof(int));
    if (len > 0) {
        memcpy(r->sequence, s, (size_t)len);
        memcpy(r->eqv, e, (size_t)len * sizeof(int));
    }
    fa_batch_free(b);
    return r;
}

extern "C" void free_consensus_data(consensus_data *c) {  // falcon.c:776
    if (!c) return;
    free(c->sequence);
    free(c->eqv);
    free(c);
}

// ---- DW_banded.c:115-337 ---------------------------------------------------
extern "C" alignment *align(char *query_seq, seq_coor_t q_len, char *target_seq, seq_coor_t t_len,
                            seq_coor_t band_tolerance, int get_aln_str) {
    std::lock_guard<std::mutex> lk(g_mu);
    fa_ctx *c = default_ctx();
    const char *q = query_seq, *t = target_seq;
    alignment *out = nullptr;
    if (fa_align_pairs(c, 1, &q, &q_len, &t, &t_len, band_tolerance, get_aln_str, &out))
        die("align");
    return out;
}

extern "C" void free_alignment(alignment *a) {
    if (!a) return;
    free(a->q_aln_str);
    free(a->t_aln_str);
    free(a);
}

// ---- host-struct table functions (kmer_lookup.c) ---------------------------
extern "C" void init_kmer_lookup(kmer_lookup *kl, seq_coor_t size) {  // :80-88
    for (seq_coor_t i = 0; i < size; i++) {
        kl[i].start = INT_MAX;
        kl[i].last = INT_MAX;
        kl[i].count = 0;
    }
}

extern "C" kmer_lookup *allocate_kmer_lookup(seq_coor_t size) {  // :71-78
    kmer_lookup *kl = (kmer_lookup *)malloc((size_t)size * sizeof(kmer_lookup));
    init_kmer_lookup(kl, size);
    return kl;
}

extern "C" void free_kmer_lookup(kmer_lookup *kl) { free(kl); }

extern "C" void init_seq_array(seq_array sa, seq_coor_t size) { memset(sa, 0xff, (size_t)size); }

extern "C" seq_array allocate_seq(seq_coor_t size) {  // :95-107
    seq_array sa = (seq_array)malloc((size_t)size);
    init_seq_array(sa, size);
    return sa;
}

extern "C" void free_seq_array(seq_array sa) { free(sa); }

extern "C" seq_addr_array allocate_seq_addr(seq_coor_t size) {  // :113
    return (seq_addr_array)calloc((size_t)size, sizeof(seq_addr));
}

extern "C" void free_seq_addr_array(seq_addr_array sda) { free(sda); }

static inline int code_of(char c, int *ok) {
    switch (c) {
    case 'A': *ok = 1; return 0;
    case 'C': *ok = 1; return 1;
    case 'G': *ok = 1; return 2;
    case 'T': *ok = 1; return 3;
    default: *ok = 0; return 0;
    }
}

// Table form of the seed index (:140-192): first/last occurrence + count per
// k-mer, sda[] links each occurrence to the next one of the same k-mer.
extern "C" void add_sequence(seq_coor_t start, unsigned int K, char *seq, seq_coor_t seq_len,
                             seq_addr_array sda, seq_array sa, kmer_lookup *lk) {
    for (seq_coor_t i = 0; i < seq_len; i++) {
        int ok;
        int c = code_of(seq[i], &ok);
        if (ok) sa[start + i] = (base)c;
    }
    if (seq_len <= (seq_coor_t)K) return;
    const unsigned mask = (K >= 16) ? 0xffffffffu : ((1u << (2 * K)) - 1u);
    unsigned km = 0;
    for (unsigned j = 0; j < K; j++) km = (km << 2) | (sa[start + j] & 3u);
    for (seq_coor_t i = 0; i < seq_len - (seq_coor_t)K; i++) {
        kmer_lookup *e = &lk[km];
        const seq_coor_t pos = start + i;
        if (e->start == INT_MAX) {
            e->start = pos;
        } else {
            sda[e->last] = pos;
        }
        e->last = pos;
        e->count += 1;
        km = ((km << 2) | sa[start + i + K]) & mask;
    }
}

extern "C" void mask_k_mer(seq_coor_t size, kmer_lookup *kl, seq_coor_t threshold) {  // :195-204
    for (seq_coor_t i = 0; i < size; i++) {
        if (kl[i].count > threshold) kl[i].start = kl[i].last = INT_MAX;
    }
}

extern "C" kmer_match *find_kmer_pos_for_seq(char *seq, seq_coor_t seq_len, unsigned int K,
                                             seq_addr_array sda, kmer_lookup *lk) {  // :207-286
    std::vector<seq_coor_t> qv, tv;
    std::vector<unsigned char> code((size_t)std::max(seq_len, 1), 0);
    for (seq_coor_t i = 0; i < seq_len; i++) {
        int ok;
        code[i] = (unsigned char)code_of(seq[i], &ok);
    }
    const seq_coor_t step = (seq_coor_t)(K >> 1);
    for (seq_coor_t i = 0; i < seq_len - (seq_coor_t)K; i += step) {
        unsigned km = 0;
        for (unsigned j = 0; j < K; j++) km = (km << 2) | code[i + j];
        seq_coor_t pos = lk[km].start;
        if (pos == INT_MAX) continue;
        for (;;) {
            qv.push_back(i);
            tv.push_back(pos);
            const seq_coor_t nx = sda[pos];
            if (nx <= pos) break;
            pos = nx;
        }
    }
    kmer_match *m = (kmer_match *)malloc(sizeof(kmer_match));
    m->count = (seq_coor_t)qv.size();
    m->query_pos = (seq_coor_t *)calloc(qv.size() + 1, sizeof(seq_coor_t));
    m->target_pos = (seq_coor_t *)calloc(tv.size() + 1, sizeof(seq_coor_t));
    if (!qv.empty()) {
        memcpy(m->query_pos, qv.data(), qv.size() * sizeof(seq_coor_t));
        memcpy(m->target_pos, tv.data(), tv.size() * sizeof(seq_coor_t));
    }
    return m;
}

extern "C" void free_kmer_match(kmer_match *m) {
    if (!m) return;
    free(m->query_pos);
    free(m->target_pos);
    free(m);
}

extern "C" aln_range *find_best_aln_range(kmer_match *km, seq_coor_t K, seq_coor_t bin_size,
                                          seq_coor_t count_th) {  // :294-427
    (void)K;
    aln_range *r = (aln_range *)calloc(1, sizeof(aln_range));
    const int n = km->count;
    if (n <= 0) return r;
    long lo = LONG_MAX, hi = LONG_MIN;
    for (int i = 0; i < n; i++) {
        long d = (long)km->query_pos[i] - (long)km->target_pos[i];
        lo = std::min(lo, d);
        hi = std::max(hi, d);
    }
    std::vector<int> cnt((size_t)((hi - lo) / bin_size + 1), 0);
    auto bin_of = [&](int i) {
        return ((long)km->query_pos[i] - (long)km->target_pos[i] - lo) / (long)bin_size;
    };
    for (int i = 0; i < n; i++) cnt[bin_of(i)]++;
    long top = -1, top_n = 0;
    for (int i = 0; i < n; i++) {
        long b = bin_of(i);
        if (cnt[b] > top_n) {
            top_n = cnt[b];
            top = b;
        }
    }
    if (top < 0 || top_n <= count_th) return r;
    bool have_first = false, have_two = false;
    int prev_q = 0, start_q = 0, start_t = 0;
    long run = 0, best = 0;
    aln_range out = *r;
    for (int i = 0; i < n; i++) {
        long b = bin_of(i);
        if (labs(b - top) > 5 || cnt[b] <= count_th) continue;
        const int q = km->query_pos[i], t = km->target_pos[i];
        if (!have_first) {
            have_first = true;
            out.s1 = out.e1 = start_q = q;
            out.s2 = out.e2 = start_t = t;
        } else {
            have_two = true;
            run += 32 - (q - prev_q);
            if (run < 0) {
                run = 0;
                start_q = q;
                start_t = t;
            } else if (run > best) {
                best = run;
                out.s1 = start_q;
                out.s2 = start_t;
                out.e1 = q;
                out.e2 = t;
                out.score = best;
            }
        }
        prev_q = q;
    }
    if (have_two) *r = out;
    return r;
}

// find_best_aln_range2 (kmer_lookup.c:429-585), host form of what k_trimwin.hip does per read
// on the device, in the same four steps:
//   1. fold over the hits for max_q and the reference's `max_t` (which takes max_q when the
//      running value already exceeds the hit's target position -- its typo at :458, kept);
//   2. the diagonals q - t, sorted;
//   3. the densest diagonal window of width delta = 0.05 (max_q + max_t): the reference's
//      two-pointer sweep is monotone, so the window end of start s is a lower bound search
//      for ds[s] + delta (capped at the last hit); the first start with the widest span wins;
//   4. the hits inside that window, compacted in list order, chained: a hit's predecessor is
//      the earlier in-window hit with the smallest x + y gap among those within 320 bases
//      on both axes (looking back until the query distance exceeds 320; ties: the nearest
//      one), score 64 - gap per link with a floor at 0; the best-scoring hit ends the range,
//      its chain's head starts it.
extern "C" aln_range *find_best_aln_range2(kmer_match *km, seq_coor_t K, seq_coor_t bin_width,
                                           seq_coor_t count_th) {
    (void)K; (void)bin_width; (void)count_th;  // unused by the reference as well
    aln_range *out = (aln_range *)calloc(1, sizeof(aln_range));
    const int n = km->count;
    if (n <= 0) return out;
    const seq_coor_t *qp = km->query_pos, *tp = km->target_pos;
    // 1 + 2
    int max_q = -1, max_t = -1;
    std::vector<int> diag((size_t)n);
    for (int i = 0; i < n; i++) {
        max_q = std::max(max_q, (int)qp[i]);
        max_t = (max_t > (int)tp[i]) ? max_q : (int)tp[i];
        diag[i] = (int)qp[i] - (int)tp[i];
    }
    std::sort(diag.begin(), diag.end());
    const int delta = (int)(long)(0.05 * (max_q + max_t));
    // 3
    int win_s = 0, win_e = 0, widest = -1 - n;
    for (int s = 0; s < n; s++) {
        const int e = (int)std::min<ptrdiff_t>(
            n - 1, std::lower_bound(diag.begin(), diag.end(), diag[s] + delta) - diag.begin());
        if (s == 0 || e - s > widest) {
            widest = e - s;
            win_s = s;
            win_e = e;
        }
    }
    if (win_e - win_s < 32) return out;
    const int d_lo = diag[win_s], d_hi = diag[win_e];
    // 4
    struct Hit { int x, y, head, score, links; };
    std::vector<Hit> hit;
    hit.reserve((size_t)n);
    int best = -1, best_score = 0;
    for (int i = 0; i < n; i++) {
        const int x = qp[i], y = tp[i];
        if (x - y < d_lo || x - y > d_hi) continue;
        Hit h = {x, y, (int)hit.size(), 0, 0};
        int from = -1, gap = 65535;
        for (int j = (int)hit.size() - 1; j >= 0 && x - hit[j].x <= 320; j--) {
            const int dy = y - hit[j].y, g = x - hit[j].x + dy;
            if (dy > 0 && dy <= 320 && g < gap) {
                gap = g;
                from = j;
            }
        }
        if (from >= 0) {
            h.head = hit[from].head;
            h.score = hit[from].score + 64 - gap;
            h.links = hit[from].links + 1;
            // (a chain whose score falls below 0 keeps its head but starts counting afresh)
            if (h.score < 0) h.score = h.links = 0;
        }
        hit.push_back(h);
        if (h.score > best_score) {
            best_score = h.score;
            best = (int)hit.size() - 1;
        }
    }
    if (best < 0) return out;
    out->score = hit[best].links + 1;
    out->s1 = hit[hit[best].head].x;
    out->s2 = hit[hit[best].head].y;
    out->e1 = hit[best].x;
    out->e2 = hit[best].y;
    return out;
}

extern "C" void free_aln_range(aln_range *r) { free(r); }
