// k_trimwin.hip -- the --trim window of every read on its pile's seed.
//
// Restates, for the whole batch, what get_alignment() of the reference driver does per
// read on the host (falcon_kit/mains/consensus.py:48-99):
//   mask_k_mer             (src/c/kmer_lookup.c:195-204)  k-mers occurring more than
//                          `mask_th` times in the seed give no hits;
//   find_kmer_pos_for_seq  (:207-286)                      hits in probe order, seed
//                          positions ascending inside a probe;
//   find_best_aln_range2   (:429-585)                      widest window of sorted
//                          diagonals, then a chain over the hits inside it.
//
// One wavefront per read (persistent grid, work fetched with an atomic like k_align).
// The hit list IS materialised here (the window search needs the diagonals sorted):
// four u32 arrays of `cap` entries per resident wavefront, in LDS when the read has
// <= TW_LDS_N hits, else in the wavefront's HBM scratch slot -- the same code reaches
// both through generic pointers.
//
//   1  enumerate hits -> tq[i] = probe << 17 | seed position, ds[i] = q - t
//   2  max_q, max_t with the reference's typo (:458: max_t takes max_q, not itself,
//      when it was larger) -- a fold of threshold functions, composed with a wave scan
//   3  bitonic sort of the diagonals
//   4  per start s the reach e(s) of the window [ds[s], ds[s] + delta) by binary search
//      (the reference's two-pointer sweep is monotone, so e(s) = min(n-1, lower_bound));
//      first widest window, must span >= 32 hits (:490-498)
//   5  compact the hits inside the diagonal window, chain them in order: predecessor =
//      closest earlier hit (smallest Manhattan gap, ties to the nearest) with smaller t,
//      both gaps <= 320; score += 64 - gap floored at 0 (:519-551); the chain ending at
//      the first strict maximum is the range, its link count + 1 the score (:552-583).
// Integer work; one double multiply (0.05 * (max_q + max_t), as the reference).
#include "fa_device.h"

#define TW_LDS_N 2048          // hits per read held in LDS (4 arrays x 4 B)
#define TW_GAP 320
#define TW_INF 0x7fffffff

struct TrimArgs {
    const u32 *words;
    const FaSeq *seq;
    const FaPile *pile;
    const u32 *kidx;
    const u32 *kpos;
    const int *order;      // work list (-1 = padding)
    int n_work;
    int *counter;
    FaRange *out;
    u32 *scratch;          // n_slot x 4 x cap words
    u64 cap;               // entries per array and slot (power of two >= largest hit count)
    int mask_th;
    int count_only;        // 1: only count the hits of every read (sizes the scratch)
};

__device__ __forceinline__ int tw_excl_sum(int v, int lane, int &total) {
    int s = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(s, off);
        if (lane >= off) s += t;
    }
    total = __shfl(s, 63);
    return s - v;
}

__device__ __forceinline__ void trim_one(const TrimArgs &A, int g, u32 *a0, u32 *a1, u32 *a2, u32 *a3,
                                      u64 cap) {
    const int lane = fa_lane();
    const FaSeq sq = A.seq[g];
    FaRange r;
    r.s1 = r.e1 = r.s2 = r.e2 = 0;
    r.ok = 1;
    r.n_hit = 0;
    r.score = 0;
    if (sq.idx == 0) {  // the seed itself
        A.out[g] = r;
        return;
    }
    const FaPile pm = A.pile[sq.pile];
    const u32 *w = A.words + sq.woff;
    const u32 *T = A.kidx + pm.kidx_off;
    const u32 *P = A.kpos + pm.kpos_off;
    const int n_probe = (sq.len > FA_K) ? (sq.len - FA_K + 3) / 4 : 0;
    u32 *tq = a0;
    int *ds = (int *)a1;

    // ---- 1: hits
    int n = 0, last_p = -1;
    for (int p0 = 0; p0 < n_probe; p0 += 64) {
        const int p = p0 + lane;
        u32 lo = 0;
        int cnt = 0;
        if (p < n_probe) {
            const u32 km = fa_kmer8(w, 4 * p);
            lo = T[km];
            cnt = (int)(T[km + 1] - lo);
            if (cnt > A.mask_th) cnt = 0;
        }
        int tot;
        const int off = n + tw_excl_sum(cnt, lane, tot);
        if (!A.count_only) {
            if ((u64)(n + tot) > cap) {  // cannot happen: cap comes from the counting pass
                r.ok = -1;
                A.out[g] = r;
                return;
            }
            for (int j = 0; j < cnt; j++) {
                const int t = (int)P[lo + (u32)j];
                tq[off + j] = ((u32)p << 17) | (u32)t;
                ds[off + j] = 4 * p - t;
            }
        }
        const u64 hm = fa_ballot(cnt > 0);
        if (hm) last_p = p0 + 63 - __builtin_clzll(hm);
        n += tot;
    }
    r.n_hit = n;
    if (A.count_only || n == 0) {
        A.out[g] = r;
        return;
    }
    __threadfence_block();

    // ---- 2: max_q = q of the last hit (q never decreases); max_t: v <- (v > t ? q : t)
    const int max_q = 4 * last_p;
    int max_t = -1;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const int nv = min(64, n - i0);
        int fa = 0, fhi = 0, flo = 0;  // f(m) = m > fa ? fhi : flo
        if (i < n) {
            const u32 v = tq[i];
            fa = (int)(v & 0x1ffffu);
            fhi = 4 * (int)(v >> 17);
            flo = fa;
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int pa = __shfl_up(fa, off), phi = __shfl_up(fhi, off), plo = __shfl_up(flo, off);
            if (lane >= off) {  // earlier segment (p*) first, then mine
                const int nhi = (phi > fa) ? fhi : flo;
                const int nlo = (plo > fa) ? fhi : flo;
                fa = pa;
                fhi = nhi;
                flo = nlo;
            }
        }
        const int ca = __shfl(fa, nv - 1), chi = __shfl(fhi, nv - 1), clo = __shfl(flo, nv - 1);
        max_t = (max_t > ca) ? chi : clo;
    }
    const int delta = (int)(long)(0.05 * (double)(max_q + max_t));  // :470

    // ---- 3: sort the diagonals (bitonic, padded with +inf)
    int npad = 64;
    while (npad < n) npad <<= 1;
    for (int i = n + lane; i < npad; i += 64) ds[i] = TW_INF;
    __threadfence_block();
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i0 = 0; i0 < npad; i0 += 64) {
                const int i = i0 + lane;
                const int x = i ^ j;
                if (x > i) {
                    const int va = ds[i], vb = ds[x];
                    const bool up = (i & k) == 0;
                    if ((va > vb) == up) {
                        ds[i] = vb;
                        ds[x] = va;
                    }
                }
            }
            __threadfence_block();
        }
    }

    // ---- 4: widest window
    int b_span = -0x7fffffff, b_s = 0, b_e = 0;
    for (int s0 = 0; s0 < n; s0 += 64) {
        const int s = s0 + lane;
        if (s < n) {
            const long long target = (long long)ds[s] + delta;
            int lo = 0, hi = n;  // first index with ds[idx] >= target
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if ((long long)ds[mid] < target) lo = mid + 1; else hi = mid;
            }
            const int e = min(n - 1, lo);
            if (e - s > b_span) {  // strict: the lane's first
                b_span = e - s;
                b_s = s;
                b_e = e;
            }
        }
    }
    {
        const int wspan = fa_wave_max(b_span);
        const int cand = (b_span == wspan) ? b_s : TW_INF;
        const int ws = fa_wave_min(cand);  // the first start among the widest
        const u64 who = fa_ballot(cand == ws) & fa_ballot(b_span == wspan);
        const int src = __builtin_ctzll(who);
        b_s = ws;
        b_e = __shfl(b_e, src);
        b_span = wspan;
    }
    if (b_e - b_s < 32) {  // :490-498
        A.out[g] = r;
        return;
    }
    const int d_lo = ds[b_s], d_hi = ds[b_e];
    __threadfence_block();

    // ---- 5a: compact the hits inside the diagonal window (order kept)
    u32 *ctq = a1;  // the sorted diagonals are done with
    int m = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        u32 v = 0;
        bool in = false;
        if (i < n) {
            v = tq[i];
            const int d = 4 * (int)(v >> 17) - (int)(v & 0x1ffffu);
            in = d >= d_lo && d <= d_hi;
        }
        const u64 im = fa_ballot(in);
        if (in) ctq[m + __popcll(im & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))))] = v;
        m += __popcll(im);
    }
    __threadfence_block();

    // ---- 5b: chain
    int *sc = (int *)a2, *ln = (int *)a3, *rt = (int *)a0;
    int top = -1, top_score = 0, top_links = 0;
    for (int i = 0; i < m; i++) {
        const u32 cv = ctq[i];  // uniform address
        const int cx = 4 * (int)(cv >> 17), cy = (int)(cv & 0x1ffffu);
        int best = TW_INF;
        for (int base = i - 1; base >= 0; base -= 64) {
            const int j = base - lane;
            int key = TW_INF;
            int near_gap = TW_INF;
            if (j >= 0) {
                const u32 pv = ctq[j];
                const int px = 4 * (int)(pv >> 17), py = (int)(pv & 0x1ffffu);
                near_gap = cx - px;
                if (cx - px <= TW_GAP && cy > py && cy - py <= TW_GAP)
                    key = ((cx - px + cy - py) << 12) | (i - 1 - j);
            }
            best = min(best, fa_wave_min(key));
            // the farthest candidate of this chunk already lies beyond the q gap: stop
            const int far = __shfl(near_gap, min(63, base));
            if (far > TW_GAP) break;
            if (i - 1 - (base - 63) >= 4095) break;  // (cannot be reached: <= 81 probes x mask_th hits)
        }
        int s = 0, l = 0, root = i;
        if (best != TW_INF) {
            const int cand = i - 1 - (best & 0xfff);
            const int gap = best >> 12;
            s = sc[cand] + 64 - gap;
            l = ln[cand] + 1;
            root = rt[cand];
            if (s < 0) {
                s = 0;
                l = 0;
            }
        }
        sc[i] = s;
        ln[i] = l;
        rt[i] = root;
        if (s > top_score) {
            top_score = s;
            top_links = l;
            top = i;
        }
    }
    if (top >= 0) {
        const u32 ev = ctq[top], sv = ctq[rt[top]];
        r.score = top_links + 1;
        r.e1 = 4 * (int)(ev >> 17);
        r.e2 = (int)(ev & 0x1ffffu);
        r.s1 = 4 * (int)(sv >> 17);
        r.s2 = (int)(sv & 0x1ffffu);
    }
    A.out[g] = r;
}

__global__ __launch_bounds__(64) void k_trimwin(TrimArgs A) {
    __shared__ u32 lds[4 * TW_LDS_N];
    const int lane = fa_lane();
    u32 *slot = A.scratch ? A.scratch + (u64)blockIdx.x * 4ull * A.cap : nullptr;
    for (;;) {
        // (work fetch as in k_align: no lane-0 branch around the atomic)
        int wi = atomicAdd(A.counter, lane == 0 ? 1 : 0);
        wi = __builtin_amdgcn_readfirstlane(wi);
        if (wi >= A.n_work) break;
        const int g = __builtin_amdgcn_readfirstlane(A.order[wi]);
        if (g < 0) continue;
        // the counting pass stored the read's hit count: small reads work in LDS
        const int n_known = A.count_only ? 0 : __builtin_amdgcn_readfirstlane(A.out[g].n_hit);
        int npad = 64;
        while (npad < n_known) npad <<= 1;
        if (A.count_only || npad <= TW_LDS_N) {
            trim_one(A, g, lds, lds + TW_LDS_N, lds + 2 * TW_LDS_N, lds + 3 * TW_LDS_N, TW_LDS_N);
        } else {
            trim_one(A, g, slot, slot + A.cap, slot + 2 * A.cap, slot + 3 * A.cap, A.cap);
        }
        __syncthreads();
    }
}

void fa_launch_trimwin(const FaBatchDev &b, int n_slot, int *counter, u32 *scratch, u64 cap,
                       int mask_th, int count_only, hipStream_t s) {
    if (b.n_chain == 0) return;
    TrimArgs A;
    A.words = b.words; A.seq = b.seq; A.pile = b.pile; A.kidx = b.kidx; A.kpos = b.kpos;
    A.order = b.chain_order; A.n_work = b.n_chain; A.counter = counter; A.out = b.range;
    A.scratch = scratch; A.cap = cap; A.mask_th = mask_th; A.count_only = count_only;
    (void)hipMemsetAsync(counter, 0, sizeof(int), s);
    hipLaunchKernelGGL(k_trimwin, dim3(n_slot), dim3(64), 0, s, A);
}
