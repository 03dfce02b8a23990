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
// Mapping: one wavefront per read, lane <-> probe (query offsets 0,4,8,..).
//   pass A  bucket bounds of every probe -> diagonal extent (wave min/max), and a RECORD per probe
//           for the later passes: the probe's first and last hit themselves when it has at most
//           two (98 % of the probes of the bench piles), its bucket otherwise
//   pass B  diagonal histogram in LDS (ds atomics) + per bin the order key of
//           its first hit, so "first fullest bin in hit order" (:360-366) is a
//           reduction over bins instead of a second walk over the hits
//   pass C  wave reduction over the bins
//   pass D  the reference's sequential scan (:385-411) as a wave prefix scan:
//           all kept hits of one probe share q, so a probe is one element
//           r -> max(c, r + a + c)  with a = 32 - (q - q_prev), c = 32*(kept-1);
//           these maps are closed under composition, which gives every probe
//           its entering score; the first strict maximum and the last reset
//           before it are wave reductions.
// Integer work on L2-resident tables (the CSR index is 262 KB per pile);
// HBM traffic is the packed reads (L/4 bytes); no MFMA.
#include "fa_device.h"

#define CH_BIN (FA_K * 6)  // falcon.c:602-604
#define CH_TH 5

struct ChainArgs {
    const u32 *words;
    const FaSeq *seq;
    const FaPile *pile;
    const u32 *kidx;
    const u32 *kpos;
    const int *order;
    int n_seq;
    int lds_bins;
    FaRange *out;
    u64 *probe;              // per probe: the record of pass A (CH_REC_*)
    const u64 *probe_off;    // [n_seq]
};

typedef long long i64;

#define CH_U 4   // probe chunks (of 64) whose dependent loads are issued together
#define CH_UA 5  // ... in pass A, where three loads depend on one another (5: still 64 VGPRs, 8 wavefronts per SIMD)

// The chain stage is a string of dependent random reads (packed read -> bucket bounds
// -> seed positions); a wavefront that walks its probes 64 at a time spends its life
// waiting for them (measured: 74-85 % of wave cycles parked at an s_waitcnt).  Every pass
// therefore issues the loads of CH_U chunks back to back before using any of them, and only
// pass A walks the whole string: it already holds every probe's first and last hit (for the
// diagonal extent), and a probe with one or two hits has no others.  Rounds 1-3 kept the
// bucket bounds in the record, and passes B and D went back to P for every probe, one
// dependent load per hit: three to five exposed latencies per step of 256 probes in pass B,
// two to four per 64 probes in pass D.
//
// The record: bits 1..0 n = min(hits, 3);
//   n = 1, 2: bits 18..2 the first hit (seed position), bits 35..19 the last one
//   n = 3:    bits 18..2 the first bucket entry, bits 35..19 the bucket's size (>= 3)
// (seed positions and bucket sizes are < 100 000 < 2^17: fa_batch_create refuses longer seeds)
#define CH_REC(n, a, b) ((u64)(n) | ((u64)(a) << 2) | ((u64)(b) << 19))
#define CH_REC_N(r) ((u32)(r) & 3u)
#define CH_REC_A(r) ((u32)((r) >> 2) & 0x1ffffu)
#define CH_REC_B(r) ((u32)((r) >> 19) & 0x1ffffu)

// inclusive wave scan of the maps r -> max(u, r + v), composed left to right
// (64-bit, LDS crossbar: the fallback for reads with huge k-mer buckets)
__device__ __forceinline__ void scan_maps(i64 &u, i64 &v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const i64 pu = __shfl_up(u, off);
        const i64 pv = __shfl_up(v, off);
        if (lane >= off) {  // (pu,pv) then (u,v)
            u = max(u, pu + v);
            v = pv + v;
        }
    }
}

// the same scan on 32-bit values through DPP (row_shr 1,2,4,8, row_bcast 15 / 31);
// lanes without a source compose with the identity map (u = NEG32, v = 0)
#define NEG32 (-(1 << 30))
#define SCAN32_STEP(ctrl, rmask)                                                        \
    {                                                                                   \
        const int pu_ = __builtin_amdgcn_update_dpp(NEG32, u, ctrl, rmask, 0xf, false); \
        const int pv_ = __builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xf, false);     \
        u = max(u, pu_ + v);                                                            \
        v = pv_ + v;                                                                    \
    }
__device__ __forceinline__ void scan_maps32(int &u, int &v) {
    SCAN32_STEP(0x111, 0xf)
    SCAN32_STEP(0x112, 0xf)
    SCAN32_STEP(0x114, 0xf)
    SCAN32_STEP(0x118, 0xf)
    SCAN32_STEP(0x142, 0xa)
    SCAN32_STEP(0x143, 0xc)
}

__device__ __forceinline__ int wave_sum(int s) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    return s;
}

__global__ __launch_bounds__(64) void k_chain(ChainArgs A) {
#ifdef FA_EMU  // (tests/emu/simt: the launcher there hands the block its dynamic LDS)
    u32 *smem = (u32 *)simt::g_dyn_lds;
#else
    extern __shared__ __attribute__((aligned(16))) u32 smem[];
#endif
    u32 *bin_cnt = smem;
    u32 *bin_key = smem + A.lds_bins;
    const int lane = fa_lane();
    const int g = A.order[blockIdx.x];
    if (g < 0) return;  // padding of the interleaved work list
    const FaSeq sq = A.seq[g];
    FaRange r;
    r.s1 = r.e1 = r.s2 = r.e2 = 0;
    r.ok = 0;
    r.n_hit = 0;
    r.score = 0;
    if (sq.idx == 0) {  // the seed itself is the target, not a query
        A.out[g] = r;
        return;
    }
    const FaPile pm = A.pile[sq.pile];
    const u32 *w = A.words + sq.woff;
    const u32 *T = A.kidx + pm.kidx_off;
    const u32 *P = A.kpos + pm.kpos_off;
    const int n_probe = (sq.len > FA_K) ? (sq.len - FA_K + 3) / 4 : 0;  // i = 4p < len-K
    u64 *pr = A.probe + A.probe_off[g];  // the two later passes read the records coalesced
                                         // instead of repeating the random lookups

    // ---- pass A: diagonal extent and hit count
    int d_min = 0x7fffffff, d_max = -0x7fffffff;
    int n_hit = 0;
    u32 cnt_max = 0;  // largest bucket a probe of this read hits
    for (int p0 = 0; p0 < n_probe; p0 += 64 * CH_UA) {
        u32 km[CH_UA], lo[CH_UA], hi[CH_UA], tl[CH_UA], th[CH_UA];
        // (round 6: NO branch around the read -- `(p < n_probe) ? fa_kmer8(..) : 0` put each of the five reads into a
        // block of its own that ends with an s_waitcnt vmcnt(0) where the lanes join: five latencies in a row
        // where the comment above promises one.  A probe beyond the last reads the last one's bases; its
        // bucket is emptied below.)
#pragma unroll
        for (int u = 0; u < CH_UA; u++) {
            const int p = p0 + 64 * u + lane;
            km[u] = fa_kmer8(w, 4 * min(p, n_probe - 1));
        }
#pragma unroll
        for (int u = 0; u < CH_UA; u++) {
            lo[u] = T[km[u]];
            hi[u] = T[km[u] + 1];
        }
#pragma unroll
        for (int u = 0; u < CH_UA; u++) {
            const int p = p0 + 64 * u + lane;
            if (p >= n_probe) hi[u] = lo[u];
            const bool any = hi[u] > lo[u];
            tl[u] = P[any ? lo[u] : 0u];          // buckets are ascending
            th[u] = P[any ? hi[u] - 1u : 0u];
        }
#pragma unroll
        for (int u = 0; u < CH_UA; u++) {  // (selects, not branches: only the store is under a lane mask)
            const int p = p0 + 64 * u + lane;
            const int i = 4 * p;
            const u32 cnt = hi[u] - lo[u];  // (0 beyond the last probe)
            const u64 rec = cnt == 0 ? 0ull : cnt <= 2 ? CH_REC(cnt, tl[u], th[u]) : CH_REC(3u, lo[u], cnt);
            if (p < n_probe) pr[p] = rec;
            n_hit += (int)cnt;
            cnt_max = max(cnt_max, cnt);
            d_min = min(d_min, cnt ? i - (int)th[u] : 0x7fffffff);
            d_max = max(d_max, cnt ? i - (int)tl[u] : -0x7fffffff);
        }
    }
    d_min = fa_wave_min(d_min);
    d_max = fa_wave_max(d_max);
    cnt_max = (u32)fa_wave_max((int)min(cnt_max, 0x7fffffffu));
    n_hit = wave_sum(n_hit);
    r.n_hit = n_hit;
    if (n_hit == 0) {
        A.out[g] = r;
        return;
    }
    const int n_bin = (d_max - d_min) / CH_BIN + 1;
    if (n_bin > A.lds_bins) {  // host sizes the LDS from the same bound
        r.ok = -1;
        A.out[g] = r;
        return;
    }
    for (int b = lane; b < n_bin; b += 64) {
        bin_cnt[b] = 0;
        bin_key[b] = 0xffffffffu;
    }
    __syncthreads();

    // ---- pass B: histogram + first-hit order key per bin (:350-366)
    auto count_hit = [&](int p, u32 j, int t) {
        const int b = (4 * p - t - d_min) / CH_BIN;
        atomicAdd(&bin_cnt[b], 1u);
        atomicMin(&bin_key[b], ((u32)p << 17) | j);
    };
    for (int p0 = 0; p0 < n_probe; p0 += 64 * CH_U) {
        u64 v[CH_U];
#pragma unroll
        for (int u = 0; u < CH_U; u++) {
            const int p = p0 + 64 * u + lane;
            v[u] = pr[min(p, n_probe - 1)];
            if (p >= n_probe) v[u] = 0;
        }
#pragma unroll
        for (int u = 0; u < CH_U; u++) {
            const int p = p0 + 64 * u + lane;
            const u32 n = CH_REC_N(v[u]);
            if (n == 1 || n == 2) count_hit(p, 0, (int)CH_REC_A(v[u]));
            if (n == 2) count_hit(p, 1, (int)CH_REC_B(v[u]));
        }
#pragma unroll
        for (int u = 0; u < CH_U; u++) {  // the probes with larger buckets: four entries per round trip
            const int p = p0 + 64 * u + lane;
            const bool heavy = CH_REC_N(v[u]) == 3;
            if (!fa_ballot(heavy)) continue;
            const u32 lo = CH_REC_A(v[u]), cnt = heavy ? CH_REC_B(v[u]) : 0u;
            for (u32 j0 = 0; fa_ballot(j0 < cnt) != 0; j0 += 4) {
                if (j0 < cnt) {
                    u32 t[4];
#pragma unroll
                    for (u32 q = 0; q < 4; q++) t[q] = P[lo + min(j0 + q, cnt - 1u)];
#pragma unroll
                    for (u32 q = 0; q < 4; q++)
                        if (j0 + q < cnt) count_hit(p, j0 + q, (int)t[q]);
                }
            }
        }
    }
    __syncthreads();

    // ---- pass C: fullest bin, ties broken by the earliest first hit
    u32 best_cnt = 0, best_key = 0xffffffffu;
    int top_bin = -1;
    for (int b = lane; b < n_bin; b += 64) {
        const u32 c = bin_cnt[b], k = bin_key[b];
        if (c > best_cnt || (c == best_cnt && c > 0 && k < best_key)) {
            best_cnt = c;
            best_key = k;
            top_bin = b;
        }
    }
    {
        const u32 wc = (u32)fa_wave_max((int)best_cnt);
        u32 k = (best_cnt == wc && top_bin >= 0) ? best_key : 0xffffffffu;
        // min over lanes of an unsigned key: flip to a signed max
        const u32 wk = ~(u32)(fa_wave_max((int)((~k) ^ 0x80000000u)) ^ 0x80000000u);
        const u64 who = __ballot(best_cnt == wc && top_bin >= 0 && best_key == wk);
        const int src = who ? (__ffsll((long long)who) - 1) : 0;
        top_bin = __shfl(top_bin, src);
        best_cnt = wc;
        if (!who) top_bin = -1;
    }

    // ---- pass D: filter (:369-383) + running-score scan (:385-411)
    int kept_total = 0;
    if (top_bin >= 0 && (int)best_cnt > CH_TH) {
        // 32-bit scan arithmetic is exact while 64 groups of the largest bucket cannot
        // overflow it; reads that hit monster buckets take the 64-bit crossbar scan
        // (per group |v| < len + 32 * bucket < 2^22 + 2^21; 64 of them stay below 2^29, so
        // NEG32 + v cannot wrap)
        const bool small = cnt_max < (1u << 16) && sq.len < (1 << 22);
        i64 carry_run = 0, best = 0;
        int carry_prev_q = -1;             // q of the last kept group so far (-1: none yet)
        int carry_start_q = 0, carry_start_t = 0;
        // software pipeline: records two chunks ahead; one chunk ahead the first four bucket
        // entries of the probes that have more than two hits (the index clamped, so that every
        // chunk has the same loads in flight)
        auto record_at = [&](int p) {
            const u64 v = pr[min(p, n_probe - 1)];
            return p < n_probe ? v : 0ull;
        };
        auto entry = [&](u64 v, u32 j) {  // entry min(j, size - 1) of a large bucket, P[0] for the others
            const bool heavy = CH_REC_N(v) == 3;
            return P[heavy ? CH_REC_A(v) + min(j, CH_REC_B(v) - 1u) : 0u];
        };
        u64 v_nx = record_at(lane), v_n2 = record_at(lane + 64);
        u32 e_nx[4];
#pragma unroll
        for (u32 j = 0; j < 4; j++) e_nx[j] = entry(v_nx, j);
        for (int p0 = 0; p0 < n_probe; p0 += 64) {
            const int p = p0 + lane;
            const u64 v = v_nx;
            u32 e[4];
#pragma unroll
            for (u32 j = 0; j < 4; j++) e[j] = e_nx[j];
            v_nx = v_n2;
#pragma unroll
            for (u32 j = 0; j < 4; j++) e_nx[j] = entry(v_nx, j);
            v_n2 = record_at(p + 128);
            int m = 0, t_first = 0, t_last = 0;
            const int q = 4 * p;
            auto take = [&](bool there, int t) {  // one hit, in order: the +-5 bin filter (:369-383)
                const int b = there ? (q - t - d_min) / CH_BIN : 0;
                int db = b - top_bin;
                if (db < 0) db = -db;
                const bool keep = there && db <= 5 && (int)bin_cnt[b] > CH_TH;
                if (keep && m == 0) t_first = t;
                if (keep) t_last = t;
                m += keep ? 1 : 0;
            };
            const u32 n = CH_REC_N(v);
            const bool heavy = n == 3;
            const u32 cnt = heavy ? CH_REC_B(v) : n;  // (0 beyond the last probe)
            take(cnt >= 1, (int)(heavy ? e[0] : CH_REC_A(v)));
            take(cnt >= 2, (int)(heavy ? e[1] : CH_REC_B(v)));
            if (fa_ballot(heavy) != 0) {
                take(cnt >= 3, (int)e[2]);
                take(cnt >= 4, (int)e[3]);
                if (fa_ballot(cnt > 4) != 0)
                    for (u32 j = 4; j < cnt; j++) take(true, (int)P[CH_REC_A(v) + j]);
            }
            const bool has = m > 0;
            const u64 hm = fa_ballot(m > 0);
            if (!hm) continue;
            // q of the previous kept group
            const u64 below = hm & ((lane == 0) ? 0ull : (~0ull >> (64 - lane)));
            int prev_q = below ? (4 * (p0 + 63 - __clzll((long long)below))) : carry_prev_q;
            const bool first_ever = has && prev_q < 0;
            // map of this group: r -> max(c, r + a + c)
            const i64 c = 32ll * (m - 1);
            const i64 a = first_ever ? 0 : (i64)(32 - (q - prev_q));
            i64 r_out;  // score after my group
            if (small) {
                int u = has ? (int)c : NEG32;
                int vv = has ? (int)(a + c) : 0;
                scan_maps32(u, vv);
                // carry_run may be large: r_out = max(u, carry_run + vv) in 64 bits
                r_out = max((i64)u, carry_run + (i64)vv);
            } else {
                i64 u = has ? c : (i64)(-(1ll << 60));
                i64 vv = has ? (a + c) : 0;
                scan_maps(u, vv, lane);
                r_out = max(u, carry_run + vv);
            }
            // entering score of my group = leaving score of the previous kept group
            // (r_out of a lane without a group equals that of the last group before it)
            i64 r_prev = __shfl_up(r_out, 1);
            if (lane == 0) r_prev = carry_run;
            const bool reset = has && (first_ever || (r_prev + a < 0));
            // best so far: first strict maximum in order (:402)
            const i64 cand = has ? r_out : (i64)(-(1ll << 60));
            i64 wmax = cand;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) wmax = max(wmax, (i64)__shfl_xor(wmax, off));
            if (wmax > best) {
                const u64 at = fa_ballot(cand == wmax) & hm;
                const int L = __ffsll((long long)at) - 1;
                best = wmax;
                const u64 rs = __ballot(reset) & ((L == 63) ? ~0ull : ((1ull << (L + 1)) - 1));
                int sq_, st_;
                if (rs) {
                    const int R = 63 - __clzll((long long)rs);
                    sq_ = 4 * (p0 + R);
                    st_ = __shfl(t_first, R);
                } else {
                    sq_ = carry_start_q;
                    st_ = carry_start_t;
                }
                r.s1 = sq_;
                r.s2 = st_;
                r.e1 = 4 * (p0 + L);
                r.e2 = __shfl(t_last, L);
                r.score = best;
            }
            // carries for the next chunk
            const int last = 63 - __clzll((long long)hm);
            if (carry_prev_q < 0) {  // the very first kept hit initialises the range (:386-390)
                const int F = __ffsll((long long)hm) - 1;
                if (r.score == 0 && best == 0) {
                    r.s1 = r.e1 = 4 * (p0 + F);
                    r.s2 = r.e2 = __shfl(t_first, F);
                }
            }
            carry_run = __shfl(r_out, last);
            carry_prev_q = 4 * (p0 + last);
            const u64 rsall = __ballot(reset);
            if (rsall) {
                const int R = 63 - __clzll((long long)rsall);
                carry_start_q = 4 * (p0 + R);
                carry_start_t = __shfl(t_first, R);
            }
            kept_total += wave_sum(m);
        }
    }
    if (kept_total <= 1) {  // :413-419
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
    A.out[g] = r;
}

#ifndef FA_EMU
void fa_launch_chain(const FaBatchDev &b, int max_bins, hipStream_t s) {
    if (b.n_seq == 0 || b.n_chain == 0) return;
    ChainArgs A;
    A.words = b.words; A.seq = b.seq; A.pile = b.pile; A.kidx = b.kidx; A.kpos = b.kpos;
    A.order = b.chain_order; A.n_seq = b.n_seq; A.out = b.range;
    A.probe = b.probe; A.probe_off = b.probe_off;
    A.lds_bins = (max_bins + 3) & ~3;
    size_t lds = (size_t)A.lds_bins * 2 * sizeof(u32);
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute((const void *)k_chain, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
    hipLaunchKernelGGL(k_chain, dim3(b.n_chain), dim3(64), lds, s, A);
}

// (fa_warm: the code object of this file is loaded when one of its kernels is first looked at)
void fa_touch_chain() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(k_chain));
}
#endif  // FA_EMU
