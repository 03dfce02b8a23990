// tests/san/host_san.cpp -- TEST INFRASTRUCTURE: the HIP-free host side of the library (reader.cpp: the
// LA4Falcon stream reader with its pread() helpers and its ring of kept batches; pack_host.cpp: ASCII -> 2 bits
// per base; fasta.cpp: the record printer) driven the way the worker drives it -- an ingest thread that runs
// ahead, a staging thread that packs batch n on several threads while batch n + 1 is being read, printers on
// several threads -- compiled with -fsanitize=thread or -fsanitize=address (falcon_amd/csrc/Makefile: tsan, asan).
// A report makes the sanitizer end the process with a status that is not 0; tests/test_host_sanitizers.py runs
// both builds.  What it checks itself: every way of reading (file / pipe with ragged writes, 2 or 8 batches
// kept, one or four read helpers) hands out the same piles, and what a batch points to is still what it was
// when the reader has moved on by as many batches as it promised.
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <unistd.h>

#include "../../falcon_amd/csrc/fa_host.h"
#include "../../include/falcon_amd.h"

static uint64_t fnv(uint64_t h, const void *p, size_t n) {
    const unsigned char *c = (const unsigned char *)p;
    for (size_t i = 0; i < n; i++) h = (h ^ c[i]) * 1099511628211ull;
    return h;
}

static std::string make_stream(int n_piles, unsigned seed) {
    std::mt19937 g(seed);
    std::string s;
    auto bases = [&](int n) {
        std::string b((size_t)n, 'A');
        for (char &c : b) c = "ACGT"[g() & 3];
        return b;
    };
    for (int p = 0; p < n_piles; p++) {
        const int seed_len = 2000 + (int)(g() % 4000), reads = 12 + (int)(g() % 30);
        char id[32];
        snprintf(id, sizeof(id), "%09d", p);
        s += id; s += ' '; s += bases(seed_len); s += '\n';
        for (int r = 0; r < reads; r++) {
            snprintf(id, sizeof(id), "%09d", 1000000 + p * 100 + r);
            s += id; s += ' '; s += bases(600 + (int)(g() % (unsigned)seed_len)); s += '\n';
            if (g() % 23 == 0) s += "a line of more than two tokens is skipped\n";
        }
        s += (g() % 17 == 0) ? "* *\n" : "+ +\n";
    }
    s += "- -\n";
    return s;
}

struct Batch {
    int n_piles;
    const int *pile_n_seq;
    const char *const *seqs;
    const int *seq_len;
    const char *const *seed_ids;
    uint64_t digest_at_handout;
};

static uint64_t digest(const Batch &b) {
    uint64_t h = 1469598103934665603ull;
    int k = 0;
    for (int p = 0; p < b.n_piles; p++) {
        h = fnv(h, b.seed_ids[p], strlen(b.seed_ids[p]));
        for (int i = 0; i < b.pile_n_seq[p]; i++, k++) h = fnv(h, b.seqs[k], (size_t)b.seq_len[k]);
    }
    return h;
}

// the staging thread's work on a batch: every sequence packed to 2 bits per base by `n_threads` threads into one
// buffer (as engine.hip does into its pinned staging buffer), then the digest of the words
static uint64_t pack_batch(const Batch &b, int n_threads) {
    int n_seq = 0;
    for (int p = 0; p < b.n_piles; p++) n_seq += b.pile_n_seq[p];
    std::vector<long long> off((size_t)n_seq + 1, 0);
    for (int i = 0; i < n_seq; i++) off[(size_t)i + 1] = off[(size_t)i] + ((b.seq_len[i] + 15) / 16 + 2 + 3) / 4 * 4;
    std::vector<unsigned> words((size_t)off[(size_t)n_seq], 0xdeadbeefu);
    std::atomic<int> next{0}, bad{0};
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++)
        th.emplace_back([&] {
            for (int i; (i = next.fetch_add(1)) < n_seq;)
                if (fa_pack_host(b.seqs[i], b.seq_len[i], words.data() + off[(size_t)i], off[(size_t)i + 1] - off[(size_t)i]) >= 0)
                    bad.fetch_add(1);
        });
    for (std::thread &t : th) t.join();
    if (bad.load()) { fprintf(stderr, "host_san: fa_pack_host refused a clean sequence\n"); exit(2); }
    return fnv(1469598103934665603ull, words.data(), words.size() * sizeof(unsigned));
}

// one pass over the stream: an ingest thread calls fa_reader_next and queues what it gets, this thread (the
// "stager") takes the batches `keep - 1` calls late, checks them and packs them
static uint64_t run(int fd, int keep, int max_piles, long *n_piles_out) {
    fa_reader *r = fa_reader_open(fd, 10, 500, 2, 40, 0);
    if (!r) { fprintf(stderr, "host_san: fa_reader_open failed\n"); exit(2); }
    if (fa_reader_keep(r, keep) != 0) { fprintf(stderr, "host_san: fa_reader_keep(%d) refused\n", keep); exit(2); }
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Batch> q;
    bool done = false;
    int taken = 0;  // batches the stager has finished with: the reader may only be keep - 1 calls ahead of it
    std::thread ingest([&] {
        for (int n_read = 0;; n_read++) {
            {   // (the worker's ingest thread waits the same way before it lets a batch's buffers go)
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return n_read - taken < keep - 1 || keep <= 1; });
            }
            Batch b{};
            b.n_piles = fa_reader_next(r, max_piles, 3000000, &b.pile_n_seq, &b.seqs, &b.seq_len, &b.seed_ids);
            if (b.n_piles < 0) { fprintf(stderr, "host_san: %s\n", fa_reader_error(r)); exit(2); }
            if (b.n_piles > 0) b.digest_at_handout = digest(b);
            std::lock_guard<std::mutex> lk(mu);
            if (b.n_piles == 0) { done = true; cv.notify_all(); return; }
            q.push_back(b);
            cv.notify_all();
        }
    });
    uint64_t all = 1469598103934665603ull;
    long n_piles = 0;
    for (;;) {
        Batch b;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !q.empty() || done; });
            if (q.empty()) break;
            b = q.front();
            q.pop_front();
        }
        if (digest(b) != b.digest_at_handout) { fprintf(stderr, "host_san: a kept batch changed under the reader\n"); exit(3); }
        const uint64_t w = pack_batch(b, 4);
        all = fnv(all, &w, sizeof(w));
        all = fnv(all, &b.digest_at_handout, sizeof(uint64_t));
        n_piles += b.n_piles;
        {
            std::lock_guard<std::mutex> lk(mu);
            taken++;
            cv.notify_all();
        }
    }
    ingest.join();
    fa_reader_close(r);
    *n_piles_out = n_piles;
    return all;
}

int main(int argc, char **argv) {
    const int n_piles = argc > 1 ? atoi(argv[1]) : 60;
    const std::string text = make_stream(n_piles, 7);
    char path[] = "/tmp/host_san_XXXXXX";
    const int tfd = mkstemp(path);
    if (tfd < 0 || write(tfd, text.data(), text.size()) != (ssize_t)text.size()) { perror("host_san: temp file"); return 2; }
    close(tfd);
    uint64_t first = 0;
    long first_n = -1;
    int variant = 0;
    for (int pipe_mode = 0; pipe_mode < 2; pipe_mode++)
        for (int keep : {2, 8})
            for (int one_slot = 0; one_slot < 3; one_slot++, variant++) {
                // (2: several helpers again, the scanner reading the bytes itself instead of walking their lists)
                if (one_slot == 1) setenv("FALCON_AMD_READER_SLOTS1", "1", 1); else unsetenv("FALCON_AMD_READER_SLOTS1");
                if (one_slot == 2) setenv("FALCON_AMD_READER_SCAN_INLINE", "1", 1); else unsetenv("FALCON_AMD_READER_SCAN_INLINE");
                int fd;
                std::thread writer;
                if (pipe_mode) {
                    int pf[2];
                    if (pipe(pf)) { perror("pipe"); return 2; }
                    fd = pf[0];
                    writer = std::thread([&text, w = pf[1], variant] {   // ragged writes, now and then a pause
                        std::mt19937 g(11u + (unsigned)variant);
                        for (size_t at = 0; at < text.size();) {
                            const size_t n = std::min(text.size() - at, (size_t)1 + g() % (g() % 5 ? 70000 : 900));
                            const ssize_t k = write(w, text.data() + at, n);
                            if (k <= 0) break;
                            at += (size_t)k;
                            if (g() % 61 == 0) usleep(300);
                        }
                        close(w);
                    });
                } else {
                    fd = open(path, O_RDONLY);
                }
                long got_n = 0;
                const uint64_t d = run(fd, keep, 5 + variant % 3, &got_n);
                if (writer.joinable()) writer.join();
                close(fd);
                // (the batches are cut by piles and bases, so the digest of the batch digests depends on max_piles:
                // what must agree is the pile count; the per-pile content is checked through the pack digest of
                // variants with the same max_piles)
                if (first_n < 0) first_n = got_n;
                if (got_n != first_n) { fprintf(stderr, "host_san: variant %d read %ld piles, the first one %ld\n", variant, got_n, first_n); return 3; }
                if (variant % 3 == 0) { if (!first) first = d; else if (d != first) { fprintf(stderr, "host_san: variant %d differs\n", variant); return 3; } }
            }
    unlink(path);
    // the record printer from several threads at once
    {
        std::string cns((size_t)5000, 'A');
        for (size_t i = 0; i < cns.size(); i++) cns[i] = (i % 977 == 0) ? 'a' : "ACGT"[(i * 7) & 3];
        std::vector<std::thread> th;
        std::atomic<long long> total{0};
        for (int t = 0; t < 8; t++)
            th.emplace_back([&, t] {
                std::vector<char> out(8192);
                for (int i = 0; i < 200; i++) total.fetch_add(fa_fasta_records("000000001", cns.data(), (long long)cns.size(), (t + i) % 3, out.data(), (long long)out.size()));
            });
        for (std::thread &t : th) t.join();
        if (total.load() <= 0) return 3;
    }
    printf("host_san: %ld piles read %d ways, no difference\n", first_n, variant);
    return 0;
}
