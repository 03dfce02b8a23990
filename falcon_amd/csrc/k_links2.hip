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
//     a one-base insertion (8), delta 1 (8): a read-modify-write, no search -- for all 64 lanes
//     as ONE predicated stream per column (a lane the case does not apply to adds 0 to a slot it
//     is not in; round 5); the others (deeper insertions, alignment starts, the unitig mode's
//     first columns) go to a short linked list (entries from a pool the 64 positions share: a
//     seed that lost two bases sees a dozen different two-base insertions at one position and
//     none at its neighbours), searched linearly, behind one wavefront-wide test per column;
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
// the five that 7.7 KB allowed (10.65 -> 9.34 ms per 3072 piles, tests/test_kernel_resources.py;
// 7.4 ms with the predicated column, at 0.93 of the chip's instruction-issue ceiling).
#include <type_traits>

#include "k_msa.h"

#define L2_DIR 20    // groups with a slot of their own, per position
#define L2_POOL 128   // listed groups per 64 positions (entries of the lanes' linked lists): with 3-byte slots 5.1 KB of
                      // LDS per wavefront, 8 per SIMD
#define L2_POOL_BIG 4096  // ... of the second instance, which takes the segments the first cannot hold (piles of
                          // a thousand reads list a few dozen groups at every position)

// does any lane hold p?  (The mask is pinned to a scalar register pair: a comparison of
// ballot(p) with 0 is otherwise rebuilt as a vector compare of a 0/1 select -- three vector
// instructions where s_cmp_lg_u64 does.)
__device__ __forceinline__ bool l2_any(bool p) {
    u64 m = fa_ballot(p);
#ifndef FA_EMU
    asm volatile("" : "+s"(m));
#endif
    return m != 0ull;
}
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
        // a listed group: found in my list, or a new entry of the pool at its head
        auto list_add = [&](u32 dl, u32 key) {
            const u32 k = (dl << 16) | key;
            for (u32 e = head; e != L2_NIL; e = pnx[e])
                if (pk[e] == k) { pc[e] += 1u; return; }
            const u32 e = dl > 255u ? (u32)POOL : atomicAdd(&pool_n, 1u);
            if (e >= (u32)POOL) { overflow = true; return; }
            pc[e] = 1u | (new_rank((int)dl) << 16);
            pk[e] = k;
            pnx[e] = (pnx_t)head;
            head = e;
        };
        auto add_group = [&](int dl, u32 key, int slot) {
            if (slot >= 0) {
                const u32 c = dcnt[slot * 64 + lane];
                dcnt[slot * 64 + lane] = (u16)(c + 1u);
                if (c == 0u) drnk[slot * 64 + lane] = (uint8_t)new_rank(dl);
                return;
            }
            list_add((u32)dl, key);
        };
        // what an alignment's FIRST column adds to my position (falcon.c:126-160): its tag word w;
        // `ld`: the alignment opens with an insertion run, which sits at the position before its
        // first column and comes without a delta-0 tag
        auto add_column_first = [&](int ld, u32 insoff, u32 w) {
            const int nins = tag_nins(w);
            const u32 base0 = (w & TAG_DEL) ? 4u : sb;
            const bool nocol = ld != 0;
            if (!nocol) {
                // no previous node (p_t_pos == -1, falcon.c:434); in the unitig mode a read placed at
                // t > 0 links to (t - 1, delta 0) with the '.' base, which the scorer reads as '-'
                // (:140, :431)
                if (t == 0 || !unitig) add_group(0, base0 | (5u << 3) | (1u << 14), -1);
                else add_group(0, base0 | (4u << 3) | (1u << 15), -1);
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
            struct Col4 { int u[4], j[4]; u32 w[4], wp[4]; bool on[4], valid[4]; };  // (j: the alignment's lane)
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
                    c.j[q] = j;
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
                    // One stream for every lane, nothing divergent on it: the delta-0 group of a column
                    // behind a column with at most one inserted base and the group of a column's first
                    // inserted base have slots of their own -- two reads, two writes, the lanes the case
                    // does not apply to adding 0 to a slot they are not in.  A slot's first alignment
                    // (its rank) and everything that goes to the list leave the stream through ONE
                    // wavefront-wide test each.  (Round 4 wrote this as nested per-lane branches: 125
                    // branch instructions per column in the code, ~200 instructions issued per alignment
                    // and 64 positions, more than half of them scalar bookkeeping of EXEC.)
                    const u32 w = cur.w[q], wp = cur.wp[q];
                    const int u = cur.u[q];
                    const bool on = cur.on[q], in = on && u > 0;
                    const u32 nins = (w >> TAG_NINS_SHIFT) & 0xffu, pn = (wp >> TAG_NINS_SHIFT) & 0xffu;
                    const u32 del = w >> 31;
                    // (coverage counts delta-0 tags, falcon.c:357-360: an alignment that opens with an
                    // insertion run has none at its first position and takes this back below)
                    cov += on ? 1 : 0;
                    maxn = max(maxn, on ? (int)nins : 0);
                    if (any_overflow) {
                        const int ld = __builtin_amdgcn_readlane(v_ld, cur.j[q]);
                        cov -= (on && u == 0 && ld != 0) ? 1 : 0;
                        continue;
                    }
                    const u32 pb1 = wp & 3u;  // the column before it carried one inserted base: this one
                    const bool f0 = in && pn <= 1u, f1 = in && nins >= 1u && nins <= (u32)INL;
                    const u32 slot0 = pn ? 4u + del * 4u + pb1 : del * 2u + (wp >> 31);  // (after a plain column: was it a '-'?)
                    const u32 slot1 = 12u + del * 4u + (w & 3u);
                    const u32 i0 = (f0 ? slot0 : 0u) * 64u + (u32)lane, i1 = (f1 ? slot1 : 12u) * 64u + (u32)lane;
                    const u32 c0 = dcnt[i0], c1 = dcnt[i1];
                    dcnt[i0] = (u16)(c0 + (f0 ? 1u : 0u));
                    dcnt[i1] = (u16)(c1 + (f1 ? 1u : 0u));
                    // (0 where a slot meets its first alignment)
                    const u32 z = min(f0 ? c0 : 1u, f1 ? c1 : 1u);
                    if (l2_any(z == 0u)) {
                        const bool fresh0 = f0 && c0 == 0u, fresh1 = f1 && c1 == 0u;
                        const u32 r0 = lvln & 255u, r1 = (lvln >> 8) & 255u;
                        overflow = overflow || (fresh0 && r0 == 255u) || (fresh1 && r1 == 255u);
                        if (fresh0) drnk[i0] = (uint8_t)r0;
                        if (fresh1) drnk[i1] = (uint8_t)r1;
                        lvln += (fresh0 ? 1u : 0u) + (fresh1 ? 256u : 0u);
                    }
                    // ---- the rest, ~1 % of the columns each and a lane or two of every other
                    // wavefront: a column behind two or more inserted bases (its delta-0 group is a
                    // listed one), the second and later inserted bases of this column; and, far
                    // rarer, a run too long for the tag word or an alignment's first column
                    const u32 deepest = max(pn, nins);
                    const u32 rest = on ? (u > 0 ? deepest : 255u) : 0u;  // (255: the first column)
                    if (!l2_any(rest >= 2u)) continue;
                    // items of a lane: 0 = the delta-0 group behind pn >= 2 inserted bases; k >= 1 = the
                    // group of inserted base k + 1
                    const bool deep0 = in && pn >= 2u, deepn = in && nins >= 2u;
                    const u32 it_end = deepn ? nins : (deep0 ? 1u : 0u);
                    auto items = [&](auto general, bool mine, u32 insoff) {
                        u32 it = mine ? (deep0 ? 0u : 1u) : it_end;
                        while (l2_any(it < it_end)) {
                            if (it < it_end) {
                                // (node base, base before it, depth before it): the last inserted base of the
                                // column before and this column's base -- or two neighbours of this column's run
                                const u32 src = it ? w : wp, d = it ? it : pn;  // `d`: 1-based index of the base before
                                u32 pbase, nb;
                                if (decltype(general)::value) {
                                    pbase = l2_ins_base(A, insoff, src, (int)d);
                                    nb = it ? l2_ins_base(A, insoff, w, (int)it + 1) : (del ? 4u : sb);
                                } else {  // (both runs inline in their tag words)
                                    pbase = (src >> (2u * d - 2u)) & 3u;
                                    nb = it ? (w >> (2u * it)) & 3u : (del ? 4u : sb);
                                }
                                list_add(it ? it + 1u : 0u, nb | (pbase << 3) | (d << 6));
                                it++;
                            }
                        }
                    };
                    if (l2_any(rest > (u32)INL)) {
                        // (what only these cases read of the alignment comes from its lane when they do)
                        const int ld = __builtin_amdgcn_readlane(v_ld, cur.j[q]);
                        const u32 insoff = (u32)__builtin_amdgcn_readlane((int)v_ins, cur.j[q]);
                        const bool first = on && u == 0;
                        if (first) {
                            cov -= ld != 0 ? 1 : 0;
                            add_column_first(ld, insoff, w);
                        }
                        if (in && nins > (u32)INL) {
                            const u32 b1 = l2_ins_base(A, insoff, w, 1);
                            add_group(1, b1 | ((del ? 4u : sb) << 3), (int)(12u + del * 4u + b1));
                        }
                        items(std::true_type(), in && deepest > (u32)INL, insoff);
                    }
                    items(std::false_type(), in && deepest <= (u32)INL, 0u);
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
__global__ __launch_bounds__(64, 8) void k_links2(MsaArgs A) {
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
