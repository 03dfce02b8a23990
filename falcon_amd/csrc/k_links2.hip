// k_links2.hip -- the MSA graph's links (update_col, src/c/falcon.c:232-263), lanes = target POSITIONS.
//
// k_links (k_msa.hip, rounds 1-3) gives a lane to every alignment overlapping a segment and
// walks the positions one by one: per level a ballot-and-peel loop on the scalar unit finds the
// lanes holding the same (base, previous node) -- ~250 instructions per position, most of them
// scalar, with 45 of 64 lanes holding anything.  Here the segment's positions are the lanes
// and the alignments are walked one by one (64 tag words per load, one per lane, coalesced):
// what an alignment's column adds to a position is a handful of per-lane vector instructions,
// every lane busy with its own position, nothing scalar, nothing cross-lane:
//
//   * the tag word of the alignment at the position and at the position before it give the
//     column's key (node base, previous node's base and depth) for delta 0 and for every
//     inserted base -- the same keys k_links builds (falcon.c:126-160);
//   * the groups of a position live in lane-private LDS words.  The keys that make up nearly
//     all links have a slot of their own -- delta 0 after a plain column (4 keys), delta 0 after
//     a one-base insertion (8), delta 1 (8): a read-modify-write, no search, reached without the
//     general column code for every column but an alignment's first; the others (deeper
//     insertions, alignment starts, the unitig mode's first columns) go to a short linked list
//     (entries from a pool the 64 positions share: a seed that lost two bases sees a dozen
//     different two-base insertions at one position and none at its neighbours), searched
//     linearly;
//   * a group's rank among the links of its LEVEL is taken when it is created -- alignments
//     come in read order, so that is the reference's first-insertion order (Q5) -- and with the
//     groups per level of every lane known, a prefix sum over the lanes places every link word:
//     the segment's links leave back to back, level by level inside a position, exactly where
//     k_links puts them.
//
// 64 positions with more listed groups than the pool holds, or a level with more than 255
// groups, send their SEGMENT on through a to-do list -- to a second instance with a large pool,
// behind that to k_links -- so the tables stay small whatever the input: 4996 bytes of LDS per
// wavefront (three bytes per slot: u16 count + u8 rank; 128 pool entries with byte links), which is
// eight wavefronts per SIMD -- the kernel spent 41 % of its wave cycles parked at an s_waitcnt with
// the five that 7.7 KB allowed (10.65 -> 9.34 ms per 3072 piles, tests/test_kernel_resources.py).
#include <type_traits>

#include "k_msa.h"

#define L2_DIR 20    // groups with a slot of their own, per position
#define L2_POOL 128   // listed groups per 64 positions (entries of the lanes' linked lists): with 3-byte slots 5.1 KB of
                      // LDS per wavefront, 8 per SIMD
#define L2_POOL_BIG 4096  // ... of the second instance, which takes the segments the first cannot hold (piles of
                          // a thousand reads list a few dozen groups at every position)

// base `d` (1-based) of a tag's insertion run
__device__ __forceinline__ u32 l2_ins_base(const MsaArgs &A, u32 ins_off, u32 w, int d) {
    if (tag_nins(w) <= INL) return (w >> (2 * (d - 1))) & 3u;
    // (a run too long for the tag word -- rare.  Waited for on the spot: a load still pending
    // on one path into a join makes the compiler drain the memory queue at every later use,
    // and with it the tag words requested ahead)
    return fa_settled((u32)A.insb[ins_off + (w & TAG_PAY_MASK) + (u32)(d - 1)]);
}
// link word (without the count) of a group key: node base | previous base << 3 | previous depth << 6
// | start << 14 (| 1 << 15: the unitig mode's first column, whose previous node reads as '-')
__device__ __forceinline__ u32 l2_word(u32 key) {
    const u32 nb = key & 7u, pb = (key >> 3) & 7u, pd = (key >> 6) & 0xffu;
    return (nb << LW_NB_SHIFT) | (((key >> 14) & 1u) ? (1u << LW_START_BIT) : ((pd * 5u + pb) << LW_PIDX_SHIFT));
}

template <int POOL>
__device__ __forceinline__ void links2_segment(const MsaArgs &A, int sidx) {
    __shared__ u16 dcnt[L2_DIR * 64];              // the slots of their own: a group's count ...
    __shared__ uint8_t drnk[L2_DIR * 64];          // ... and its rank in its level (3 bytes, not 4: 8 wavefronts per SIMD fit)
    // the listed groups: a pool shared by the wavefront's 64 positions, a linked list per lane
    __shared__ u32 pk[POOL], pc[POOL];             // key | level << 16 ; count | rank << 16
    // (the next entry of the list: a byte for the small pool -- with it the tables are 4996 bytes)
    typedef typename std::conditional<(POOL <= 255), uint8_t, u16>::type pnx_t;
    constexpr u32 L2_NIL = POOL <= 255 ? 0xffu : 0xffffu;
    __shared__ pnx_t pnx[POOL];
    __shared__ u32 pool_n;
    const int lane = fa_lane();
    const int p = fa_uni(A.seg_pile[sidx]);
    const int t_lo = fa_uni(A.seg_t0[sidx]);
    const FaPile pm = A.pile[p];
    const int T = pm.seed_len;
    const int t_hi = min(T, t_lo + TSEG);
    if (A.score_out[p].err) return;
    const u32 *seedw = A.words + A.seq[pm.first].woff;
    FaTInfo *ti = A.tinfo + A.t_off[p];
    u32 *links = A.links + A.link_off[p];
    u16 *nlk = A.lvl_nlink16 + pm.node_off / 5;
    const u32 a0 = A.acc_first[p], a1 = A.acc_first[p + 1];
    // the segment's first link slot and first level slot (k_sscan)
    const u32 link0 = fa_uni(A.seg_base[2 * (size_t)sidx]);
    u32 out = link0;
    u32 lvl_next = fa_uni(A.seg_base[2 * (size_t)sidx + 1]);
    const bool unitig = A.first_links_back != 0;
    // (FALCON_AMD_LINKS1 -- k_links makes the links of every segment: only the position records here)
    bool overflow = false;
    bool any_overflow = A.links_old != 0;
    unsigned long long bound = 0;  // sum of coverage x levels over my positions

    for (int h0 = t_lo; h0 < t_hi; h0 += 64) {
        const int t = h0 + lane;
        const bool tin = t < t_hi;
        int cov = 0, maxn = 0;  // alignments with a column at my position, the longest insertion run behind one
        const u32 sb = tin ? fa_base_at(seedw, t) : 0u;
        const u32 sbp = (tin && t > 0) ? fa_base_at(seedw, t - 1) : 0u;
        fa_wave_sync();  // (the half before is done with the tables)
#pragma unroll
        for (int s = 0; s < L2_DIR; s++) dcnt[s * 64 + lane] = 0;
        pool_n = 0u;     // (every lane, the same word)
        fa_wave_sync();
        u32 head = L2_NIL;  // my list
        u32 lvln = 0;       // groups of levels 0..3, 8 bits each

        // rank of a new group among its level's links (levels 0..3 count in lvln, the deeper
        // ones only have listed groups)
        auto new_rank = [&](int dl) -> u32 {
            if (dl < 4) {
                const u32 r = (lvln >> (8 * dl)) & 255u;
                if (r == 255u) overflow = true;
                lvln += 1u << (8 * dl);
                return r;
            }
            u32 r = 0;
            for (u32 e = head; e != L2_NIL; e = pnx[e]) r += ((pk[e] >> 16) & 255u) == (u32)dl ? 1u : 0u;
            return r;
        };
        auto add_group = [&](int dl, u32 key, int slot) {
            if (slot >= 0) {
                const u32 c = dcnt[slot * 64 + lane];
                dcnt[slot * 64 + lane] = (u16)(c + 1u);
                if (c == 0u) drnk[slot * 64 + lane] = (uint8_t)new_rank(dl);
                return;
            }
            const u32 k = ((u32)dl << 16) | key;
            for (u32 e = head; e != L2_NIL; e = pnx[e])
                if (pk[e] == k) { pc[e] += 1u; return; }
            const u32 e = dl > 255 ? (u32)POOL : atomicAdd(&pool_n, 1u);
            if (e >= (u32)POOL) { overflow = true; return; }
            pc[e] = 1u | (new_rank(dl) << 16);
            pk[e] = k;
            pnx[e] = (pnx_t)head;
            head = e;
        };
        // what a column of an alignment adds to my position: its tag word w, the tag word wp of
        // the column before it (falcon.c:126-160)
        auto add_column = [&](int u, int ld, u32 insoff, u32 w, u32 wp, bool groups) {
            const int nins = tag_nins(w);
            const u32 base0 = (w & TAG_DEL) ? 4u : sb;
            const bool nocol = ld != 0 && u == 0;  // only the leading insertion run, no delta-0 column
            maxn = max(maxn, nins);
            if (!nocol) cov++;  // (coverage counts delta-0 tags, falcon.c:357-360)
            if (!groups) return;
            if (!nocol) {
                if (u == 0) {
                    // the alignment's first column: no previous node (p_t_pos == -1, falcon.c:434);
                    // in the unitig mode a read placed at t > 0 links to (t - 1, delta 0) with the
                    // '.' base, which the scorer reads as '-' (:140, :431)
                    if (t == 0 || !unitig) add_group(0, base0 | (5u << 3) | (1u << 14), -1);
                    else add_group(0, base0 | (4u << 3) | (1u << 15), -1);
                } else {
                    // the column before it: its last inserted base, or its base / '-'
                    const u32 pn = (u32)tag_nins(wp);
                    const u32 pb = pn > 0 ? l2_ins_base(A, insoff, wp, (int)pn) : ((wp & TAG_DEL) ? 4u : sbp);
                    const int del = base0 == 4u ? 1 : 0;
                    int slot = -1;
                    if (pn == 0) slot = del * 2 + (pb == 4u ? 1 : 0);
                    else if (pn == 1) slot = 4 + del * 4 + (int)pb;
                    add_group(0, base0 | (pb << 3) | (pn << 6), slot);
                }
            }
            for (int dl = 1; dl <= nins; dl++) {
                const u32 b = l2_ins_base(A, insoff, w, dl);
                if (nocol && dl == 1) {
                    // the alignment's very first tag (see above)
                    if (unitig) add_group(1, b | (4u << 3) | (1u << 15), -1);
                    else add_group(1, b | (5u << 3) | (1u << 14), -1);
                } else {
                    const u32 pbb = dl == 1 ? base0 : l2_ins_base(A, insoff, w, dl - 1);
                    const int slot = dl == 1 ? 12 + (base0 == 4u ? 4 : 0) + (int)b : -1;
                    add_group(dl, b | (pbb << 3) | ((u32)(dl - 1) << 6), slot);
                }
            }
        };

        // ---- the alignments, 64 at a time in the lanes; those overlapping the 64 positions one
        // by one, the tag words of the next one requested before this one is worked in
        for (u32 c0 = a0; c0 < a1; c0 += 64) {
            const u32 ai = c0 + (u32)lane;
            int v_s2 = 0, v_tc = 0, v_ld = 0;
            u32 v_dlo = 0, v_dhi = 0, v_ins = 0;
            bool ov = false;
            if (ai < a1) {
                const FaTagAln ta = A.ta[ai];
                const int tcw = A.tcov[ai];
                v_ld = (tcw & TCOV_LEAD) ? 1 : 0;  // (a leading insertion run sits at s2 - 1, k_tags)
                v_s2 = ta.s2 - v_ld;
                v_tc = (tcw & ~TCOV_LEAD) + v_ld;
                const u64 doff = ta.desc_off - (u64)v_ld;
                v_dlo = (u32)doff; v_dhi = (u32)(doff >> 32);
                v_ins = ta.ins_off;
                ov = v_tc > v_ld && v_s2 < min(h0 + 64, t_hi) && v_s2 + v_tc > h0;
            }
            // the overlapping alignments four at a time: the tag words of the next four are
            // under way while these four are worked in (one alignment ahead was not enough: a
            // round of ~150 instructions is shorter than a trip to HBM)
            struct Col4 { int u[4], ld[4]; u32 insoff[4], w[4], wp[4]; bool on[4], valid[4]; };
            int j_last = 0;
            auto request4 = [&](u64 &m) -> Col4 {
                Col4 c;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    c.valid[q] = m != 0ull;
                    if (c.valid[q]) j_last = (int)__builtin_ctzll(m);  // (else: the last one's words again)
                    m &= m - 1;
                    const int j = j_last;
                    const int s2 = __builtin_amdgcn_readlane(v_s2, j);
                    const int tc = __builtin_amdgcn_readlane(v_tc, j);
                    c.ld[q] = __builtin_amdgcn_readlane(v_ld, j);
                    c.insoff[q] = (u32)__builtin_amdgcn_readlane((int)v_ins, j);
                    const u64 doff = ((u64)(u32)__builtin_amdgcn_readlane((int)v_dhi, j) << 32) |
                                     (u64)(u32)__builtin_amdgcn_readlane((int)v_dlo, j);
                    const u32 *dptr = A.desc + doff;
                    c.u[q] = t - s2;
                    c.on[q] = c.valid[q] && tin && c.u[q] >= 0 && c.u[q] < tc;  // the alignment has a column at my position
                    // (both words from every lane, the index clamped into the alignment's words --
                    // the word before the first one is the slot in front of them: a fixed number
                    // of loads per request lets the wait for THESE words leave the next ones in flight)
                    const int iu = min(max(c.u[q], 0), tc - 1);
                    c.w[q] = dptr[iu];
                    c.wp[q] = dptr[iu - 1];
                }
                return c;
            };
            u64 m = fa_ballot(ov);
            if (m == 0ull) continue;
            Col4 cur = request4(m);
            for (;;) {
                const bool more = m != 0ull;
                const Col4 nxt = request4(m);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    // Every column but an alignment's first one (its start link, or the unitig mode's
                    // leading run: add_column) takes this path: the group of its delta-0 tag has a
                    // slot of its own unless the column before it carried two or more inserted bases,
                    // the group of its first inserted base has one too; only the deeper levels --
                    // and that one delta-0 case -- go to the list.  (The general add_column for every
                    // column that was not "plain" cost 7.5 of the kernel's 12.7 ms: a few lanes per
                    // alignment take it, so the wavefront did, nearly every time.)
                    const u32 w = cur.w[q], wp = cur.wp[q];
                    const u32 nins = (w >> TAG_NINS_SHIFT) & 0xffu, pn = (wp >> TAG_NINS_SHIFT) & 0xffu;
                    if (cur.on[q] && cur.u[q] > 0) {
                        const u32 del = w >> 31, base0 = del ? 4u : sb;
                        cov++;
                        maxn = max(maxn, (int)nins);
                        if (!any_overflow) {
                            // the column before it: its last inserted base, or its base / '-'
                            const u32 pb = pn == 0u ? ((wp >> 31) ? 4u : sbp)
                                                    : (pn == 1u ? (wp & 3u) : l2_ins_base(A, cur.insoff[q], wp, (int)pn));
                            if (pn <= 1u) {
                                const u32 slot0 = pn ? 4u + del * 4u + pb : del * 2u + (pb >> 2);
                                const u32 c = dcnt[slot0 * 64u + (u32)lane];
                                const bool fresh = c == 0u;
                                const u32 r = lvln & 255u;
                                overflow = overflow || (fresh && r == 255u);
                                dcnt[slot0 * 64u + (u32)lane] = (u16)(c + 1u);
                                if (fresh) drnk[slot0 * 64u + (u32)lane] = (uint8_t)r;
                                lvln += fresh ? 1u : 0u;
                            } else {
                                add_group(0, base0 | (pb << 3) | (pn << 6), -1);
                            }
                            if (nins) {
                                const u32 b1 = nins <= (u32)INL ? (w & 3u) : l2_ins_base(A, cur.insoff[q], w, 1);
                                const u32 slot1 = 12u + del * 4u + b1;
                                const u32 c = dcnt[slot1 * 64u + (u32)lane];
                                const bool fresh = c == 0u;
                                const u32 r = (lvln >> 8) & 255u;
                                overflow = overflow || (fresh && r == 255u);
                                dcnt[slot1 * 64u + (u32)lane] = (u16)(c + 1u);
                                if (fresh) drnk[slot1 * 64u + (u32)lane] = (uint8_t)r;
                                lvln += fresh ? 256u : 0u;
                                u32 pbb = b1;
                                for (int dl = 2; dl <= (int)nins; dl++) {
                                    const u32 b = l2_ins_base(A, cur.insoff[q], w, dl);
                                    add_group(dl, b | (pbb << 3) | ((u32)(dl - 1) << 6), -1);
                                    pbb = b;
                                }
                            }
                        }
                    } else if (cur.on[q]) {
                        add_column(cur.u[q], cur.ld[q], cur.insoff[q], w, wp, !any_overflow);
                    }
                }
                if (!more) break;
                cur = nxt;
            }
        }

        // ---- the position's record (k_score1/2, k_backtrace and k_links read it): coverage,
        // levels = 1 + its deepest insertion (slot 0 is always (t 0, delta 0); an uncovered
        // position has none, and keeps no links), level slots in position order
        const int nlev = cov > 0 ? 1 + maxn : (t == 0 && tin ? 1 : 0);
        const int lsum = wave_incl_sum(nlev, lane);
        FaTInfo x;
        x.lvl_start = lvl_next + (u32)(lsum - nlev);
        x.link_start = link0;
        x.cov = (u16)min(cov, 65535);
        x.nlev = (u16)nlev;
        if (tin) ti[t] = x;
        lvl_next += (u32)__builtin_amdgcn_readlane(lsum, 63);
        bound += (unsigned long long)(min(cov, 65535) * nlev);
        // ---- its links: level by level, inside a level by rank
        any_overflow = any_overflow || fa_ballot(overflow) != 0ull;
        if (any_overflow) continue;
        const u32 n0 = lvln & 255u, n1 = (lvln >> 8) & 255u, n2 = (lvln >> 16) & 255u, n3 = lvln >> 24;
        const u32 sum4 = n0 + n1 + n2 + n3;
        u32 deep = 0;  // groups of levels >= 4
        for (u32 e = head; e != L2_NIL; e = pnx[e]) deep += ((pk[e] >> 16) & 255u) >= 4u ? 1u : 0u;
        const u32 gtot = (tin && x.cov != 0) ? sum4 + deep : 0u;  // (an uncovered position keeps no links, k_tscan)
        const u32 gsum = (u32)wave_incl_sum((int)gtot, lane);
        const u32 base = out + gsum - gtot;
        auto level_off = [&](u32 dl) -> u32 {
            if (dl < 4u) return dl == 0u ? 0u : dl == 1u ? n0 : dl == 2u ? n0 + n1 : n0 + n1 + n2;
            u32 o = sum4;
            for (u32 e = head; e != L2_NIL; e = pnx[e]) {
                const u32 de = (pk[e] >> 16) & 255u;
                o += (de >= 4u && de < dl) ? 1u : 0u;
            }
            return o;
        };
        if (gtot != 0u) {
#pragma unroll
            for (int s = 0; s < L2_DIR; s++) {
                const u32 d = (u32)dcnt[s * 64 + lane] | ((u32)drnk[s * 64 + lane] << 16);
                if (d & 0xffffu) {
                    u32 key, dl;
                    if (s < 4) { key = ((s & 2) ? 4u : sb) | (((s & 1) ? 4u : sbp) << 3); dl = 0; }
                    else if (s < 12) { key = (((s - 4) & 4) ? 4u : sb) | ((u32)((s - 4) & 3) << 3) | (1u << 6); dl = 0; }
                    else { key = (u32)((s - 12) & 3) | ((((s - 12) & 4) ? 4u : sb) << 3); dl = 1; }
                    links[base + level_off(dl) + (d >> 16)] = l2_word(key) | (d & 0xffffu);
                }
            }
            for (u32 e = head; e != L2_NIL; e = pnx[e]) {
                const u32 k = pk[e], c = pc[e];
                links[base + level_off((k >> 16) & 255u) + (c >> 16)] = l2_word(k & 0xffffu) | (c & 0xffffu);
            }
            for (u32 dl = 0; dl < (u32)x.nlev; dl++) {  // links per level slot
                u32 n = dl == 0u ? n0 : dl == 1u ? n1 : dl == 2u ? n2 : dl == 3u ? n3 : 0u;
                if (dl >= 4u)
                    for (u32 e = head; e != L2_NIL; e = pnx[e]) n += ((pk[e] >> 16) & 255u) == dl ? 1u : 0u;
                nlk[x.lvl_start + dl] = (u16)n;
            }
        }
        out += (u32)__builtin_amdgcn_readlane((int)gsum, 63);
    }
    {   // what the pile's scores are bounded by.  (A segment that overflows this instance's tables
        // is walked again by the one behind it and adds its share a second time: the bound only
        // decides whether k_score2's 32-bit scores are safe, so too large a bound sends a pile to
        // k_score1 a little early and changes no answer.)
        unsigned long long b = bound;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const u32 lo = (u32)__shfl((int)(u32)b, lane ^ o), hi = (u32)__shfl((int)(u32)(b >> 32), lane ^ o);
            b += ((unsigned long long)hi << 32) | lo;
        }
        if (lane == 0) atomicAdd(&A.bound[p], b);
    }
    if (any_overflow) {
        // more groups than the tables hold: the instance with the large pool takes the segment,
        // and behind that one k_links
        if (lane == 0) {
            int *todo = A.wide_count + (POOL == L2_POOL ? 0 : 1) * (A.n_seg + 1);
            todo[1 + atomicAdd(todo, 1)] = sidx;
        }
    }
}

// one wavefront per segment
__global__ __launch_bounds__(64) void k_links2(MsaArgs A) {
    if ((int)blockIdx.x < A.n_seg) links2_segment<L2_POOL>(A, (int)blockIdx.x);
}
// a fixed grid looping over the first instance's to-do list (usually empty)
__global__ __launch_bounds__(64) void k_links2_big(MsaArgs A) {
    const int *in = A.wide_count;
    const int n_in = fa_uni(in[0]);
    for (int i = (int)blockIdx.x; i < n_in; i += (int)gridDim.x) {
        links2_segment<L2_POOL_BIG>(A, fa_uni(in[1 + i]));
        __syncthreads();  // (the next segment reuses the LDS)
    }
}

#ifndef FA_EMU
void fa_touch_links2() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(k_links2));
}
void fa_launch_links2(const MsaArgs &A, hipStream_t s) {
    hipLaunchKernelGGL(k_links2, dim3(A.n_seg), dim3(64), 0, s, A);
    hipLaunchKernelGGL(k_links2_big, dim3(A.n_seg < 2048 ? A.n_seg : 2048), dim3(64), 0, s, A);
}
#endif
