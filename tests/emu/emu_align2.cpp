// tests/emu/emu_align2.cpp -- TEST INFRASTRUCTURE: runs the source of the k_align2 kernel
// (falcon_amd/csrc/k_align2_core.h) on the host, one emulated wavefront after the other,
// through the lane emulator of tests/emu/fa_wave.h.  Built by tests/emu/Makefile into
// tests/emu/libemu_align2.so and driven by tests/test_emu_align2.py against the CPU oracle;
// the product never loads it.
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "fa_wave.h"  // (tests/emu's twin: this directory comes first on the include path)

namespace emu {
thread_local uint64_t exec = ~0ull;
thread_local Region regions[16];
thread_local int n_regions = 0;
thread_local u32 lds[1024];
// ---- accounting (scripts/a2_bytes.py): bytes and 32 / 64 / 128-byte units touched per wave-wide access, by what is
// accessed: 0 packed words, 1 tape cells, 2 tape records, 3 escape list, 4 edit scripts, 5 alignment records
static bool acct_on = false;
static int acct_depth = 0;
static bool acct_write = false;
static std::vector<std::pair<uintptr_t, size_t>> acct_lanes;
static double acct_tab[6][2][5];   // [what][read / write][instructions, bytes, 32 B units, 64 B units, 128 B units]
static const void *acct_arena = nullptr;
static size_t acct_arena_bytes = 0, acct_slot_words = 0;
static unsigned acct_ring = 0;
static int acct_what(uintptr_t a) {
    for (int i = 0; i < n_regions; i++) {
        const uintptr_t b = (uintptr_t)regions[i].p;
        if (a >= b && a < b + regions[i].bytes) {
            if (regions[i].p != acct_arena) return i == 0 ? 0 : i == 2 ? 4 : 5;
            const size_t w = ((a - b) / 4) % acct_slot_words;
            return w < (size_t)acct_ring * 16u ? 1 : w < (size_t)acct_ring * 20u ? 2 : 3;
        }
    }
    return 5;
}
Acct::Acct() { if (acct_on && acct_depth++ == 0) acct_lanes.clear(); }
Acct::~Acct() {
    if (!acct_on || --acct_depth != 0 || acct_lanes.empty()) return;
    const int what = acct_what(acct_lanes[0].first), rw = acct_write ? 1 : 0;
    double *t = acct_tab[what][rw];
    t[0] += 1;
    std::vector<uintptr_t> u[3];
    for (auto &ln : acct_lanes) {
        t[1] += (double)ln.second;
        for (int g = 0; g < 3; g++)
            for (uintptr_t x = ln.first >> (5 + g); x <= (ln.first + ln.second - 1) >> (5 + g); x++) u[g].push_back(x);
    }
    for (int g = 0; g < 3; g++) {
        std::sort(u[g].begin(), u[g].end());
        t[2 + g] += (double)(std::unique(u[g].begin(), u[g].end()) - u[g].begin());
    }
}
void check(const void *p, size_t bytes, bool write, const char *what) {
    if (acct_on && acct_depth > 0) { acct_write = write; acct_lanes.emplace_back((uintptr_t)p, bytes); }
    const char *c = (const char *)p;
    for (int i = 0; i < n_regions; i++) {
        const char *b = (const char *)regions[i].p;
        if (c >= b && c + bytes <= b + regions[i].bytes) {
            if (write && !regions[i].writable) break;
            return;
        }
    }
    fprintf(stderr, "emu: %s of %zu bytes at %p is outside every registered buffer\n", what, bytes, p);
    abort();
}
static void reg(const void *p, size_t bytes, bool writable) {
    regions[n_regions].p = p;
    regions[n_regions].bytes = bytes;
    regions[n_regions].writable = writable;
    n_regions++;
}
}  // namespace emu

#ifdef EMU_HIST
static long hist_sum[130], hist_one[70];
#define A2_HOOK_PLACE(h0, n0, h1, n1) do { if ((h0) && (h1)) hist_sum[std::min(129, (n0) + (n1))]++; if (h0) hist_one[std::min(69,(n0))]++; if (h1) hist_one[std::min(69,(n1))]++; } while (0)
extern "C" void emu_hist(long *sum, long *one) { memcpy(sum, hist_sum, sizeof(hist_sum)); memcpy(one, hist_one, sizeof(hist_one)); }
#endif
#ifdef EMU_CHECK_ROW
#define A2_HOOK_KEEP(w, keep, run0, run1, sh0, sh1, nl0, nl1, p) fprintf(stderr, "place it=%u p=%d keep=%016llx run %d %d | T0 st=%d d=%d li=%d hin=%d sh=%d nl=%d | T1 st=%d d=%d li=%d hin=%d sh=%d nl=%d\n", w.it, p, (unsigned long long)keep, (int)run0, (int)run1, w.T0.state, w.T0.d, w.T0.li, w.T0.hin, sh0, nl0, w.T1.state, w.T1.d, w.T1.li, w.T1.hin, sh1, nl1)
#define A2_HOOK_ROW(P, PAIR, h, hv, a1, b, x, y, act, fa) do { \
    for (int l_ = 0; l_ < 64; l_++) if (((act) >> l_) & 1) { \
        if (x.v[l_] < 0 || x.v[l_] > hv.vqlen.v[l_] || y.v[l_] < 0 || y.v[l_] > hv.vtlen.v[l_]) { \
            fprintf(stderr, "row P=%d PAIR=%d it=%u lane %d: x=%d y=%d a1=%d b=%d act=%016llx fa=%016llx split=%d in=%016llx\n", P, (int)PAIR, h.it, l_, x.v[l_], y.v[l_], a1.v[l_], b.v[l_], (unsigned long long)(act), (unsigned long long)(fa), h.split, (unsigned long long)h.in); \
            for (int k_ = 0; k_ < 64; k_++) fprintf(stderr, "%d:%d ", k_, hv.vx.v[k_]); fprintf(stderr, "\n"); abort(); } } } while (0)
#endif
#ifdef EMU_ROWS
#define A2_HOOK_ROW(P, PAIR, h, hv, a1, b, x, y, act, fa) do { if (h.it >= EMU_ROWS_FROM && h.it < EMU_ROWS_TO) \
    fprintf(stderr, "row it=%u P=%d pair=%d act0=%d act1=%d cells0=%u cells1=%u split=%d act=%016llx\n", h.it, P, (int)PAIR, \
            __builtin_popcountll((act) & ~h.zone1), __builtin_popcountll((act) & h.zone1), h.cells0, h.cells1, h.split, (unsigned long long)(act)); \
    if (h.it >= EMU_ROWS_FROM && h.it < EMU_ROWS_FROM + 40) { fprintf(stderr, "   best0=%d kb0=%d vx:", h.best0, (int)h.kb0); for (int l_ = 0; l_ < 40; l_++) fprintf(stderr, " %d", hv.vx.v[l_]); \
      fprintf(stderr, "\n   x/y pre-snake:"); for (int l_ = 0; l_ < 40; l_++) fprintf(stderr, " %d/%d", x.v[l_], y.v[l_]); fprintf(stderr, "\n"); } } while (0)
#endif
#ifdef EMU_TRACE_G
#define A2_HOOK_WIDE(w, wl, rc) do { fprintf(stderr, "wide episode over: rc=%d it=%u T0 st=%d g=%d d=%d li=%d hin=%d | T1 st=%d g=%d d=%d li=%d hin=%d\n   vx:", rc, (w).it, (w).T0.state, (w).T0.g, (w).T0.d, (w).T0.li, (w).T0.hin, (w).T1.state, (w).T1.g, (w).T1.d, (w).T1.li, (w).T1.hin); \
    for (int l_ = 0; l_ < 64; l_++) fprintf(stderr, " %d", (wl).vx.v[l_] < -1000 ? -1 : (wl).vx.v[l_]); fprintf(stderr, "\n   vpark:"); \
    for (int l_ = 0; l_ < 64; l_++) fprintf(stderr, " %d", (wl).vpark.v[l_] < -1000 ? -1 : (wl).vpark.v[l_]); fprintf(stderr, "\n"); } while (0)
#define A2_HOOK_ESC(t, w, long_one, want, itc, my_lane) do { if ((t).g == EMU_TRACE_G) { \
    fprintf(stderr, "esc lookup: n_esc=%d long_one=%016llx", (w).n_esc, (unsigned long long)(long_one)); \
    for (int l_ = 0; l_ < 64; l_++) if (((long_one) >> l_) & 1) fprintf(stderr, " [lane %d: it %u lane %d]", l_, itc.v[l_], my_lane.v[l_]); \
    fprintf(stderr, "\n  list:"); for (int e_ = 0; e_ < (w).n_esc; e_++) fprintf(stderr, " (%u,%u:%u)", (unsigned)((w).esc[e_] & 0xffffffffu), (unsigned)((w).esc[e_] >> 38), (unsigned)(((w).esc[e_] >> 32) & 63)); fprintf(stderr, "\n"); } } while (0)
#define A2_HOOK_EXIT(w, h) do { \
    if ((w).T0.state != A2_IDLE && (w).T0.g == EMU_TRACE_G) fprintf(stderr, "exit T0 it=%u pair=%d d=%d cells=%u li=%d hin=%d ev=%016llx fin=%016llx\n", (w).it, (w).pair, (w).T0.d, (w).T0.cells, (w).T0.li, (w).T0.hin, (unsigned long long)(h).ev, (unsigned long long)(h).fin); \
    if ((w).T1.state != A2_IDLE && (w).T1.g == EMU_TRACE_G) fprintf(stderr, "exit T1 it=%u pair=%d d=%d cells=%u li=%d hin=%d ev=%016llx fin=%016llx\n", (w).it, (w).pair, (w).T1.d, (w).T1.cells, (w).T1.li, (w).T1.hin, (unsigned long long)(h).ev, (unsigned long long)(h).fin); } while (0)
#define A2_HOOK_TRACE(t, ih, have, valid, n_rows, r_top, kv, kb) do { if ((t).g == EMU_TRACE_G) { \
    fprintf(stderr, "trace g=%d it0=%u ih=%u have=%016llx valid=%016llx n_rows=%d r_top=%d kv=%d kb:", (t).g, (t).it0, ih, \
            (unsigned long long)(have), (unsigned long long)(valid), n_rows, r_top, kv); \
    for (int l_ = 0; l_ < 64; l_++) fprintf(stderr, " %x", (unsigned)kb.v[l_]); fprintf(stderr, "\n"); } } while (0)
#endif
#ifdef EMU_DRIVE_LOG
#define A2_HOOK_DRIVE(i) do { if ((i) < 2 && h.it >= EMU_DRIVE_LOG_FROM && h.it < EMU_DRIVE_LOG_TO) fprintf(stderr, "%s it=%u kb0=%d kb1=%d split=%d in=%016llx act=%016llx parked=%u p_li=%d p_hin=%d p_kc=%d d0=%u d1=%u it_last=%u\n", (i) ? "JOIN" : "PARK", h.it, (int)h.kb0, (int)h.kb1, h.split, (unsigned long long)h.in, (unsigned long long)h.act, w_pack_get<A2_SC_PARKED>(r.sc), (int)w_pack_get<A2_SC_P_LI>(r.sc), (int)w_pack_get<A2_SC_P_HIN>(r.sc), (int)w_pack_get<A2_SC_P_KC>(r.sc), w_pack_get<A2_SC_D0>(r.sc), w_pack_get<A2_SC_D1>(r.sc), w_pack_get<A2_SC_IT_LAST>(r.sc)); } while (0)
#endif
#ifdef EMU_DRIVE
static long drive_n[3];
#define A2_HOOK_DRIVE(i) drive_n[i]++
#define A2_HOOK_TRIP() drive_n[2]++
extern "C" void emu_drive_counts(long *out) { out[0] = drive_n[0]; out[1] = drive_n[1]; out[2] = drive_n[2]; }
#endif
#include "k_align2_core.h"

// words per arena slot for a tape of `ring` iterations (must match the engine's sizing)
static u64 slot_words_for(u32 ring) { return (u64)ring * 16u + (u64)ring * 4u + (u64)A2_ESC_CAP * 2u; }

extern "C" void emu_acct_on(int on) { emu::acct_on = on != 0; if (on) memset(emu::acct_tab, 0, sizeof(emu::acct_tab)); }
extern "C" void emu_acct_get(double *out) { memcpy(out, emu::acct_tab, sizeof(emu::acct_tab)); }

extern "C" int emu_align2(const u32 *words, u64 n_words, const FaSeq *seq, int n_seq, const FaPile *pile,
                          int n_pile, const FaRange *range, const int *order, int n_work, u32 ring,
                          int n_wave, int band, double max_diff, u32 *script, u64 script_words,
                          const u64 *script_off, FaAln *aln, unsigned long long *stats) {
    if (ring < 256 || (ring & (ring - 1))) return -1;
    if (n_wave < 1) n_wave = 1;
    std::vector<u32> arena((size_t)slot_words_for(ring) * (size_t)n_wave, 0xA5A5A5A5u);
    if (const char *fill = getenv("EMU_FILL")) {  // what earlier launches may have left there
        u64 x = strtoull(fill, nullptr, 0) * 0x9E3779B97F4A7C15ull + 1;
        for (u32 &w : arena) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            // (mostly plausible cell bytes and record words, now and then anything)
            w = (x >> 60) ? (u32)(x >> 8) & 0x0f1f3f7fu : (u32)(x >> 16);
        }
    }
    int counter = 0;
    emu::n_regions = 0;
    emu::reg(words, n_words * 4, false);
    emu::reg(arena.data(), arena.size() * 4, true);
    emu::acct_arena = arena.data(); emu::acct_arena_bytes = arena.size() * 4;
    emu::acct_slot_words = (size_t)slot_words_for(ring); emu::acct_ring = ring;
    emu::reg(script, script_words * 4, true);
    emu::reg(aln, (size_t)n_seq * sizeof(FaAln), true);
    (void)n_pile;
    A2Args A;
    A.words = words; A.word_base = 0; A.seq = seq; A.pile = pile; A.range = range; A.order = order;
    A.n_work = n_work; A.counter = &counter;
    A.cells = arena.data(); A.recs = nullptr; A.esc = nullptr;
    A.slot_words = slot_words_for(ring);
    A.ring = ring;
    A.script = script; A.script_off = script_off; A.aln = aln;
    A.band = band; A.max_diff = max_diff;
    A.stats = stats;
    A.debug = 0;
    A.esc_cap = getenv("EMU_A2_ESC_CAP") ? atoi(getenv("EMU_A2_ESC_CAP")) : (int)A2_ESC_CAP;
    A.wide_patience = getenv("EMU_A2_WIDE_PATIENCE") ? atoi(getenv("EMU_A2_WIDE_PATIENCE")) : (int)A2_WIDE_PATIENCE;
    // the wavefronts of a launch run at the same time on the device and take work as they
    // go; here they run one after the other, wave w taking every n_wave-th chunk of the
    // queue is not needed for correctness -- any split of the queue is a legal schedule
    for (int w = 0; w < n_wave; w++) {
        emu::exec = ~0ull;
        // (wave w works until the queue is empty; with n_wave > 1 the first takes it all --
        // the parameter exists to exercise slot addressing)
        a2_wave(A, w);
    }
    return 0;
}
