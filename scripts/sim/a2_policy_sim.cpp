// a2_policy_sim.cpp -- DEVELOPMENT TOOL: how k_align2's lane placement behaves, counted on the host.
//
// Input: the per-row band hulls of a batch's alignments in queue order (scripts/sim/dump_bands.py).  A band's
// rows do not depend on where the band sits in the wave, so the events a placement policy raises -- bands laid
// out again, tracks parked and joined, iterations with one track -- can be counted without running a single
// DP cell.  Policies:
//   rigid   what k_align2 does (k_align2_core.h a2_place / a2_replace / a2_park / a2_join): two zones
//           [0, split) and [split, 64), two fence lanes around the boundary and the wave's ends as walls;
//           a hull on a forbidden lane lays BOTH bands out again (data movement).
//   soft    the same lanes, but the boundary between the zones is only a mask: a hull that reaches it while
//           the neighbour's hull is two lanes or more away moves the boundary, nothing else; the wave's ends
//           stay walls.
//   ring    lane = (diagonal / 2) mod 64: no walls, both boundaries between the two zones are masks; data
//           moves only when one of the two gaps between the bands closes while the other has room.
//
//   g++ -O2 -o /tmp/a2sim scripts/sim/a2_policy_sim.cpp && /tmp/a2sim /tmp/bands_ecoli.bin rigid
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct Aln { int n, ok, q_len, t_len; long long cells; std::vector<int> lo, hi; };
static std::vector<Aln> A;

static inline int fl2(int x) { return x >= 0 ? x / 2 : -((-x + 1) / 2); }

enum { IDLE = 0, RUN = 1, PARKED = 2 };
struct Trk {
    int st = IDLE, a = -1, d = 0, K0 = 0;
    int l = 0, h = 0;   // hull lanes of the last row (frame K0); ring: NOT reduced mod 64 (l <= h, any integers)
    bool wide = false;
};
struct Stats {
    long long it_pair = 0, it_single = 0, it_wide = 0, replace = 0, replace_next = 0, soft = 0, rotate = 0, park = 0,
              join = 0, place = 0, fin = 0, cells = 0, rows = 0, single_replace = 0;
};

static int FREE_MIN = 0, FREE_JOIN = 6, LOOK_EVERY = 8, MAX_N = 60;
static int RIGID_ASYM = 0;
static int SOFT_ASYM = 1;   // minimal (asymmetric) fences instead of the two-lane fence
static int RING_MINGAP = 1; // ring: a gap this small (or smaller) while the other has room: rotate; both: park
static int RING_JOIN_GAP = 3;

struct Wave {
    Trk T[2];
    unsigned it = 0;
    int pair = 0, split = 64;
    int bA = 0, bB = 0;  // ring: first lane of zone 1 / zone 0 (mod 64)
    unsigned last_replace_it = ~0u, single_since = 0;
    bool done = false;
};

static size_t next_work = 0;
static bool fetch(Trk &t) {
    if (next_work >= A.size()) return false;
    t = Trk();
    t.st = RUN; t.a = (int)next_work++; t.d = 0; t.K0 = 0; t.l = t.h = 0;
    return true;
}

// lanes of the hull after row index r of track t
static void hull_of(const Trk &t, int r, int &l, int &h) {
    const Aln &a = A[t.a];
    l = fl2(a.lo[r] - t.K0 + 1);
    h = fl2(a.hi[r] - t.K0 + 1);
}
// the band of the next row in lane coordinates, p = parity of the next iteration
static void next_band(const Trk &t, int p, int &lo, int &hi) {
    if (t.d == 0) { lo = hi = t.l; return; }
    if (p == 0) { lo = t.l; hi = t.h + 1; } else { lo = t.l - 1; hi = t.h; }
}
static void shift_to(Trk &t, int p, int nl) {  // the next band's lowest lane becomes nl
    int lo, hi;
    next_band(t, p, lo, hi);
    const int sh = lo - nl;
    t.K0 += 2 * sh; t.l -= sh; t.h -= sh;
}

// ------------------------------------------------------------------------------------------------
// rigid / soft: a2_place
// ------------------------------------------------------------------------------------------------
static void place_lin(Wave &w, Stats &s) {
    const int p = w.it & 1;
    bool have0 = w.T[0].st != IDLE, have1 = w.T[1].st != IDLE;
    int lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0;
    if (have0) next_band(w.T[0], p, lo0, hi0);
    if (have1) next_band(w.T[1], p, lo1, hi1);
    const int n0 = hi0 - lo0 + 1, n1 = hi1 - lo1 + 1;
    bool run0 = have0, run1 = have1;
    if (have0 && have1) {
        const int free_lanes = 64 - n0 - n1;
        const bool paired_now = w.pair && w.T[0].st == RUN && w.T[1].st == RUN && w.T[0].d > 0 && w.T[1].d > 0;
        if (free_lanes < (paired_now ? FREE_MIN : FREE_JOIN)) { if (n1 > n0) run0 = false; else run1 = false; }
    }
    w.T[0].wide = w.T[1].wide = false;
    if (run0 && !run1 && n0 > MAX_N) w.T[0].wide = true;
    if (run1 && !run0 && n1 > MAX_N) w.T[1].wide = true;
    const bool pair = run0 && run1;
    if (pair) {
        const int free_lanes = 64 - n0 - n1, g0 = free_lanes / 4, mid = free_lanes / 2;
        shift_to(w.T[0], p, g0);
        shift_to(w.T[1], p, g0 + n0 + mid);
        w.split = g0 + n0 + mid / 2;
    } else if (run0) { shift_to(w.T[0], p, std::max(0, (64 - n0) / 2)); w.split = 64; }
    else if (run1) { shift_to(w.T[1], p, std::max(0, (64 - n1) / 2)); w.split = 0; }
    if ((have0 && !run0 && w.T[0].st == RUN) || (have1 && !run1 && w.T[1].st == RUN)) s.park++;
    if (have0) w.T[0].st = run0 ? RUN : PARKED;
    if (have1) w.T[1].st = run1 ? RUN : PARKED;
    w.pair = pair;
    w.single_since = w.it;
    s.place++;
}

// ------------------------------------------------------------------------------------------------
// ring: place.  Lanes are integers; a track's lanes are taken mod 64.  Track 0's frame is kept, track 1 is
// laid into the middle of the free arc above it.
// ------------------------------------------------------------------------------------------------
static inline int m64(int x) { return ((x % 64) + 64) % 64; }
static void ring_bounds(Wave &w, int p) {
    // boundaries in the middle of the two gaps between the NEXT bands and the hulls
    int lo0, hi0, lo1, hi1;
    next_band(w.T[0], p, lo0, hi0);
    next_band(w.T[1], p, lo1, hi1);
    // gap A: above track 0, below track 1 (lanes hi0+1 .. lo1-1 in track-0-relative unrolled coordinates)
    const int gA = m64(lo1 - hi0 - 1), gB = m64(lo0 - hi1 - 1);
    // both parities: zone 0 needs lane h0+1 (grow up) -- next band already has what THIS parity needs; keep
    // a lane for the other parity where there is one: the boundary goes to the middle of the free lanes
    w.bA = m64(hi0 + 1 + (gA + (p == 0 ? 0 : 1)) / 2);
    w.bB = m64(hi1 + 1 + (gB + (p == 0 ? 0 : 1)) / 2);
}
static void place_ring(Wave &w, Stats &s) {
    const int p = w.it & 1;
    bool have0 = w.T[0].st != IDLE, have1 = w.T[1].st != IDLE;
    int lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0;
    if (have0) next_band(w.T[0], p, lo0, hi0);
    if (have1) next_band(w.T[1], p, lo1, hi1);
    const int n0 = hi0 - lo0 + 1, n1 = hi1 - lo1 + 1;
    bool run0 = have0, run1 = have1;
    if (have0 && have1) {
        const int free_lanes = 64 - n0 - n1;
        if (free_lanes < 2 * RING_JOIN_GAP) { if (n1 > n0) run0 = false; else run1 = false; }
    }
    w.T[0].wide = w.T[1].wide = false;
    if (run0 && !run1 && n0 > 62) w.T[0].wide = true;
    if (run1 && !run0 && n1 > 62) w.T[1].wide = true;
    const bool pair = run0 && run1;
    if (pair) {
        // whoever ran before keeps its lanes; the other goes to the middle of the free arc
        const int free_lanes = 64 - n0 - n1;
        const int keep = (w.T[0].st == RUN && w.T[0].d > 0) ? 0 : ((w.T[1].st == RUN && w.T[1].d > 0) ? 1 : 0);
        const int other = keep ^ 1;
        int klo, khi;
        next_band(w.T[keep], p, klo, khi);
        shift_to(w.T[other], p, khi + 1 + free_lanes / 2);
        ring_bounds(w, p);
    }
    if ((have0 && !run0 && w.T[0].st == RUN) || (have1 && !run1 && w.T[1].st == RUN)) s.park++;
    if (have0) w.T[0].st = run0 ? RUN : PARKED;
    if (have1) w.T[1].st = run1 ? RUN : PARKED;
    w.pair = pair;
    w.single_since = w.it;
    s.place++;
}

static int policy = 0;  // 0 rigid, 1 soft, 2 ring

static void place(Wave &w, Stats &s) { if (policy == 2) place_ring(w, s); else place_lin(w, s); }

// in-zone test of a lane range [lo, hi] (unrolled integers) for zone [zb, ze) on the ring
static bool in_arc(int lo, int hi, int zb, int ze_excl) {
    const int len = m64(ze_excl - zb) == 0 ? 64 : m64(ze_excl - zb);
    const int off = m64(lo - zb);
    return off + (hi - lo) < len;
}

static void step(Wave &w, Stats &s) {
    const int p = w.it & 1;
    int nrun = 0;
    bool fin[2] = {false, false};
    for (int t = 0; t < 2; t++) {
        Trk &T = w.T[t];
        if (T.st != RUN) continue;
        nrun++;
        const Aln &a = A[T.a];
        if (T.wide) {
            // a wide row: ceil(n / 64) passes, alone
            int lo, hi;
            next_band(T, p, lo, hi);
            const int n = hi - lo + 1;
            s.it_wide += (n + 63) / 64 - 1;
        }
        const int r = T.d++;
        s.rows++;
        if (r >= a.n) { fin[t] = true; continue; }
        hull_of(T, r, T.l, T.h);
    }
    if (nrun == 2) s.it_pair++; else if (nrun == 1) s.it_single++;
    w.it++;
    const int pn = w.it & 1;  // parity of the next row
    if (fin[0] || fin[1]) {
        for (int t = 0; t < 2; t++) if (fin[t]) { s.fin++; w.T[t].st = IDLE; fetch(w.T[t]); }
        if (w.T[0].st == IDLE && w.T[1].st == IDLE) { w.done = true; return; }
        // a parked neighbour gets its chance again
        place(w, s);
        return;
    }
    if (nrun == 0) { place(w, s); return; }
    // ------------------------------------------------------------------ a lone runner
    if (nrun == 1) {
        const int t = w.T[0].st == RUN ? 0 : 1;
        Trk &T = w.T[t], &O = w.T[t ^ 1];
        int lo, hi;
        next_band(T, pn, lo, hi);
        const int n = hi - lo + 1;
        if (T.wide) {
            if (n <= MAX_N) { T.wide = false; place(w, s); }
            return;
        }
        const int maxn = policy == 2 ? 62 : MAX_N;
        if (policy != 2) {
            // hull on lane 0 / 63: centre again
            if (T.l <= 0 || T.h >= 63 || lo < 0 || hi > 63) {
                if (n > maxn) { T.wide = true; return; }
                shift_to(T, pn, (64 - n) / 2);
                s.replace++; s.single_replace++;
            }
        } else if (n > maxn) { T.wide = true; return; }
        if (O.st == PARKED && (w.it - w.single_since) % LOOK_EVERY == 0) {
            int plo, phi;
            next_band(O, pn, plo, phi);
            const int np = O.d == 0 ? 1 : (O.h - O.l + 2);
            const int free_lanes = 64 - n - np;
            if (free_lanes >= (policy == 2 ? 2 * RING_JOIN_GAP : FREE_JOIN)) {
                O.st = RUN;
                s.join++;
                if (policy == 2) {
                    const int keep = t, other = t ^ 1;
                    int klo, khi;
                    next_band(w.T[keep], pn, klo, khi);
                    shift_to(w.T[other], pn, khi + 1 + free_lanes / 2);
                    w.pair = 1;
                    ring_bounds(w, pn);
                } else {
                    int lo0, hi0, lo1, hi1;
                    next_band(w.T[0], pn, lo0, hi0);
                    next_band(w.T[1], pn, lo1, hi1);
                    const int n0 = hi0 - lo0 + 1, n1 = hi1 - lo1 + 1;
                    const int g0 = free_lanes / 4, mid = free_lanes / 2;
                    shift_to(w.T[0], pn, g0);
                    shift_to(w.T[1], pn, g0 + n0 + mid);
                    w.split = g0 + n0 + mid / 2;
                    w.pair = 1;
                }
            }
        }
        return;
    }
    // ------------------------------------------------------------------ a pair
    Trk &T0 = w.T[0], &T1 = w.T[1];
    if (policy == 0 || policy == 1) {
        bool ev;
        const int sp = w.split;
        if ((policy == 0 && !RIGID_ASYM) || (policy == 1 && !SOFT_ASYM)) {
            // hull on a fence lane (split-1, split) or on the wall of the side the next row grows to
            ev = (T0.h >= sp - 1) || (T1.l <= sp) || (pn == 1 ? (T0.l <= 0) : (T1.h >= 63));
        } else {
            ev = pn == 1 ? (T0.l <= 0 || T1.l <= sp) : (T0.h >= sp - 1 || T1.h >= 63);
        }
        if (!ev) return;
        const int hull0 = T0.h - T0.l + 1, hull1 = T1.h - T1.l + 1, n0 = hull0 + 1, n1 = hull1 + 1;
        const int free_lanes = 64 - n0 - n1;
        if (policy == 1) {
            // the walls are fine and the gap between the hulls has room for a boundary: move the mask only
            const bool walls_ok = pn == 1 ? (T0.l >= 1) : (T1.h <= 62);
            const int gA = T1.l - T0.h - 1;
            if (walls_ok && gA >= 1) {
                // next row: grow down: b in [h0+1, l1-1]; grow up: b in [h0+2, l1]; centre it for both
                int blo = pn == 1 ? T0.h + 1 : T0.h + 2, bhi = pn == 1 ? T1.l - 1 : T1.l;
                // prefer a position that also serves the row after (the other parity)
                const int blo2 = std::max(blo, T0.h + 2), bhi2 = std::min(bhi, T1.l - 1);
                if (blo2 <= bhi2) { blo = blo2; bhi = bhi2; }
                w.split = (blo + bhi + 1) / 2;
                s.soft++;
                return;
            }
        }
        if (free_lanes < FREE_MIN || free_lanes < 0) {
            // park the narrower one (a2_park)
            const int pk = n1 > n0 ? 0 : 1;
            w.T[pk].st = PARKED;
            s.park++;
            Trk &R = w.T[pk ^ 1];
            int lo, hi;
            next_band(R, pn, lo, hi);
            const int n = hi - lo + 1;
            if (n > MAX_N) R.wide = true; else shift_to(R, pn, (64 - n) / 2);
            w.pair = 0;
            w.single_since = w.it;
            return;
        }
        const int g0 = free_lanes / 4, mid = free_lanes / 2;
        shift_to(T0, pn, g0);
        shift_to(T1, pn, g0 + n0 + mid);
        w.split = g0 + n0 + mid / 2;
        s.replace++;
        if (w.last_replace_it + 1 == w.it) s.replace_next++;
        w.last_replace_it = w.it;
        return;
    }
    // ring
    {
        int lo0, hi0, lo1, hi1;
        next_band(T0, pn, lo0, hi0);
        next_band(T1, pn, lo1, hi1);
        const bool ok0 = in_arc(lo0, hi0, w.bB, w.bA), ok1 = in_arc(lo1, hi1, w.bA, w.bB);
        if (ok0 && ok1) return;
        // free lanes between the HULLS on either side
        const int n0 = T0.h - T0.l + 1, n1 = T1.h - T1.l + 1;
        int gA = m64(T1.l - T0.h - 1), gB = m64(T0.l - T1.h - 1);
        if (n0 + n1 + gA + gB != 64) { gA = 64 - n0 - n1 - gB; }
        if (gA >= 1 + RING_MINGAP && gB >= 1 + RING_MINGAP) {
            ring_bounds(w, pn);
            s.soft++;
            return;
        }
        const int free_lanes = gA + gB;
        if (free_lanes >= 2 * (1 + RING_MINGAP)) {
            // rotate track 1 so that the gaps are equal again (data movement of one track)
            const int want = free_lanes / 2;
            // track 1's lanes move up by (want - gA)
            const int sh = want - gA;
            T1.K0 -= 2 * sh; T1.l += sh; T1.h += sh;
            ring_bounds(w, pn);
            s.rotate++;
            if (w.last_replace_it + 1 == w.it) s.replace_next++;
            w.last_replace_it = w.it;
            return;
        }
        const int pk = n1 > n0 ? 0 : 1;
        w.T[pk].st = PARKED;
        s.park++;
        w.pair = 0;
        w.single_since = w.it;
    }
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: a2sim bands.bin rigid|soft|ring [waves] [key=value ...]\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int n_aln = 0;
    if (fread(&n_aln, 4, 1, f) != 1) return 1;
    A.resize(n_aln);
    for (auto &a : A) {
        int hdr[4];
        if (fread(hdr, 4, 4, f) != 4 || fread(&a.cells, 8, 1, f) != 1) return 1;
        a.n = hdr[0]; a.ok = hdr[1]; a.q_len = hdr[2]; a.t_len = hdr[3];
        std::vector<int> buf(2 * (size_t)a.n);
        if (a.n && fread(buf.data(), 4, buf.size(), f) != buf.size()) return 1;
        a.lo.resize(a.n); a.hi.resize(a.n);
        for (int i = 0; i < a.n; i++) { a.lo[i] = buf[2 * i]; a.hi[i] = buf[2 * i + 1]; }
    }
    fclose(f);
    policy = !strcmp(argv[2], "rigid") ? 0 : !strcmp(argv[2], "soft") ? 1 : 2;
    int n_wave = argc > 3 ? atoi(argv[3]) : 0;
    for (int i = 4; i < argc; i++) {
        char *eq = strchr(argv[i], '=');
        if (!eq) continue;
        *eq = 0;
        const int v = atoi(eq + 1);
        if (!strcmp(argv[i], "free_min")) FREE_MIN = v;
        else if (!strcmp(argv[i], "free_join")) FREE_JOIN = v;
        else if (!strcmp(argv[i], "look")) LOOK_EVERY = v;
        else if (!strcmp(argv[i], "asym")) SOFT_ASYM = v;
        else if (!strcmp(argv[i], "rasym")) RIGID_ASYM = v;
        else if (!strcmp(argv[i], "mingap")) RING_MINGAP = v;
        else if (!strcmp(argv[i], "joingap")) RING_JOIN_GAP = v;
    }
    // the bench: 3072 piles (~304 k alignments) on 8192 resident wavefronts = 37 alignments per wavefront
    if (n_wave <= 0) n_wave = std::max(1, (int)(A.size() / 37));
    std::vector<Wave> W(n_wave);
    Stats s;
    for (auto &w : W) {
        fetch(w.T[0]); fetch(w.T[1]);
        if (w.T[0].st == IDLE) { w.done = true; continue; }
        place(w, s);
    }
    // every wavefront one iteration per round (they all issue at about the same rate)
    size_t live = W.size();
    long long rounds = 0;
    while (live) {
        live = 0;
        for (auto &w : W) if (!w.done) { step(w, s); live++; }
        rounds++;
    }
    const long long it = s.it_pair + s.it_single + s.it_wide;
    printf("%s: %d alignments on %d wavefronts: rows %lld, iterations %lld (pair %lld, single %lld = %.1f %%, wide passes %lld), rounds %lld\n",
           argv[2], n_aln, n_wave, s.rows, it, s.it_pair, s.it_single, 100.0 * s.it_single / it, s.it_wide, rounds);
    printf("  data moves (replace/rotate) %lld (%.1f per 1000 it; %lld single-track; %lld one iteration after the one before), "
           "mask moves %lld (%.1f per 1000 it), park %lld, join %lld, place %lld\n",
           s.replace + s.rotate, 1000.0 * (s.replace + s.rotate) / it, s.single_replace, s.replace_next, s.soft,
           1000.0 * s.soft / it, s.park, s.join, s.place);
    // instruction model: 63 per iteration, 250 per data move, 120 per mask move, 300 per park / join, 800 per place
    const double instr = 63.0 * it + 250.0 * (s.replace + s.rotate) + 120.0 * s.soft + 300.0 * (s.park + s.join) + 800.0 * s.place;
    printf("  instruction model: %.3f G (rows %.1f %%) = %.1f per row\n", instr / 1e9, 100.0 * 63.0 * it / instr, instr / s.rows);
    // ... and with round 6's costs: a data move inside the stream ~120, a look-up in the single stream's stride 12
    // per LOOK_EVERY single iterations (it was a round trip of ~150)
    const double instr6 = 63.0 * it + 120.0 * (s.replace + s.rotate) + 12.0 * s.soft + 12.0 * s.it_single / LOOK_EVERY +
                          300.0 * (s.park + s.join) + 800.0 * s.place;
    printf("  round-6 costs: %.3f G = %.1f per row\n", instr6 / 1e9, instr6 / s.rows);
    return 0;
}
