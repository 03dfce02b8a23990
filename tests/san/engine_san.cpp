// tests/san/engine_san.cpp -- TEST INFRASTRUCTURE: the batch engine's HOST logic (falcon_amd/csrc/engine.hip: pure host
// code -- staging, the front half under the context's lock, the planner thread, the back half, fetch, results, freeing,
// the block cache) under ThreadSanitizer, driven the way the worker drives it: several runner threads per context, each
// with batches between submit and wait, batches freed while others are in flight, two contexts at once.  The HIP runtime
// and the kernel launchers are stand-ins (tests/san/stub/hip/hip_runtime.h, engine_stub_kernels.cpp: no alignment is
// ever accepted, every consensus is empty) -- what runs is the engine's control flow on the threads it runs on.
// `make -C falcon_amd/csrc tsan_engine`; tests/test_host_sanitizers.py runs it.  A report ends it with a status != 0.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/falcon_amd.h"

static std::string bases(std::mt19937 &g, int n) {
    std::string s((size_t)n, 'A');
    for (char &c : s) c = "ACGT"[g() & 3];
    return s;
}

struct Piles {
    std::vector<std::string> store;
    std::vector<const char *> ptr;
    std::vector<int> n_seq;
};

static Piles make_piles(std::mt19937 &g, int n_pile) {
    Piles p;
    for (int i = 0; i < n_pile; i++) {
        const int reads = 3 + (int)(g() % 6), seed_len = 600 + (int)(g() % 900);
        p.n_seq.push_back(reads + 1);
        p.store.push_back(bases(g, seed_len));
        for (int r = 0; r < reads; r++) p.store.push_back(bases(g, 300 + (int)(g() % (unsigned)seed_len)));
    }
    for (const std::string &s : p.store) p.ptr.push_back(s.c_str());
    return p;
}

static std::atomic<long> n_done{0}, n_bad{0};

static void runner(fa_ctx *c, unsigned seed, int rounds) {
    std::mt19937 g(seed);
    for (int it = 0; it < rounds; it++) {
        // up to three batches between submit and wait, like the worker's runner threads together
        const int k = 1 + (int)(g() % 3);
        std::vector<Piles> in;
        std::vector<fa_batch *> bs;
        for (int j = 0; j < k; j++) {
            in.push_back(make_piles(g, 1 + (int)(g() % 5)));
            fa_batch *b = fa_batch_create(c, (int)in.back().n_seq.size(), in.back().n_seq.data(), in.back().ptr.data(), nullptr);
            if (!b) { fprintf(stderr, "engine_san: fa_batch_create: %s\n", fa_last_error()); n_bad++; return; }
            bs.push_back(b);
        }
        for (fa_batch *b : bs)
            if (fa_batch_submit(b, 4, 8, 0.70)) { fprintf(stderr, "engine_san: submit: %s\n", fa_last_error()); n_bad++; }
        for (size_t j = 0; j < bs.size(); j++) {
            fa_batch *b = bs[j];
            if (g() % 7 == 0) {          // a batch dropped while it is in flight (a failed job's batches are)
                fa_batch_free(b);
                continue;
            }
            const int want_eqv = (int)(g() & 1);
            if (fa_batch_wait(b) || fa_batch_fetch(b, want_eqv)) { fprintf(stderr, "engine_san: wait/fetch: %s\n", fa_last_error()); n_bad++; }
            for (int p = 0; p < (int)in[j].n_seq.size(); p++) {
                const char *s = nullptr; const int *e = nullptr; int len = -1;
                if (fa_batch_result(b, p, &s, &len, want_eqv ? &e : nullptr) || len != 0) {   // (no alignment is ever accepted here)
                    if (n_bad++ < 3) fprintf(stderr, "engine_san: result of pile %d: len %d (%s)\n", p, len, fa_last_error());
                }
            }
            fa_stats st;
            (void)fa_batch_stats(b, &st);
            if (g() % 3 == 0 && fa_batch_run(b, 4, 8, 0.70)) {                // a batch run again (bench.py does)
                if (n_bad++ < 3) fprintf(stderr, "engine_san: run again: %s\n", fa_last_error());
            }
            fa_batch_free(b);
            n_done++;
        }
    }
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 40;
    fa_ctx *c0 = fa_create(0), *c1 = fa_create(1);
    if (!c0 || !c1) { fprintf(stderr, "engine_san: fa_create: %s\n", fa_last_error()); return 2; }
    (void)fa_warm(c0, 1000000);
    std::vector<std::thread> th;
    for (int t = 0; t < 3; t++) th.emplace_back(runner, c0, 100u + (unsigned)t, rounds);
    for (int t = 0; t < 2; t++) th.emplace_back(runner, c1, 200u + (unsigned)t, rounds);
    for (std::thread &t : th) t.join();
    fa_destroy(c0);
    fa_destroy(c1);
    if (n_bad.load()) { fprintf(stderr, "engine_san: %ld calls failed\n", n_bad.load()); return 3; }
    printf("engine_san: %ld batches through two contexts on five threads, nothing failed\n", n_done.load());
    return 0;
}
