// k_score2.hip -- the score recurrence of the MSA graph (src/c/falcon.c:405-475), one workgroup of
// TWO wavefronts per pile, written so that the DEPENDENT chain of a level is as short as the
// hardware allows and nothing else stands in its way.
//
// A pile has ~2 levels (t, delta) per seed position and every level's scores depend on the
// level before it: 40 000 steps in sequence.  k_score1 (round 1-3) walked them with the whole
// work of a level on that chain -- link-word decoding, the node bookkeeping, the back pointers,
// the position loop's scalar control flow: ~180 instructions per level, and since a wavefront
// issues in order, ~1300 clocks per level whatever the occupancy.  Here a pile is walked in
// blocks of up to S2_NL levels, each block in three phases:
//
//   A  decode (data-parallel, lane = level, a loop over the level's links): every link word of
//      k_links becomes an 8-byte record in LDS -- the LDS address of its predecessor's score,
//      the LDS address of its node's score, and the addend `2 count - coverage`.  Nothing
//      here depends on the scores.
//   B  the chain (lane = link of the CURRENT level): read the record, read the predecessor's
//      score, add, and an LDS atomic maximum on the node's score slot, floored at the
//      reference's -1 beforehand.  Half a dozen instructions and ONE LDS round trip per level:
//      LDS operations of a wavefront execute in order, so the next level's reads see the
//      maxima without any wait.  Scores live in an LDS ring addressed by a running count of
//      levels, so every insertion depth takes the same path (no register-resident levels, no
//      deep-level scratch), and a node's links may come in any order.
//   C  resolve (data-parallel, lane = level again): with all scores of the block final, every
//      node finds its winning link -- the FIRST link in stored (= insertion, Q5) order that
//      reaches the node's score, and only if that score beat the -1 floor (strict '>',
//      falcon.c:420,447; Q4) -- and writes its 8-byte node record {score, (best predecessor
//      + 1) << 1 | upper}; the lane keeps its best node for the global maximum (first strict
//      maximum in (t, delta, base) order, falcon.c:464-469).
//
// Only B is a chain.  In the first half of round 4 one wavefront did A, B and C of a block one
// after the other and B was 40 % of its time; now wavefront 0 does nothing but B, and wavefront 1
// everything else, a block ahead and a block behind: in step k wavefront 0 runs the chain of
// block k - 1 while wavefront 1 resolves block k - 2 and then stages and decodes block k (into
// the set of records block k - 2 just left; there are two sets).  One workgroup barrier per step.
// The ring holds the block being decoded, the one in the chain, the one being resolved and the
// position before it at once: 1024 slots.
//
// What it does not hold goes to k_score1 through FaScoreOut.redo: the unitig mode (its links
// may name absent nodes, which k_score1's registers read as the floor), piles whose scores
// could outgrow 29 bits less the bias (never: `wide`), a position with more than S2_NL levels
// or S2_NK links, a level with more than 63 links.
#include "k_msa.h"

#define S2_NL 48              // levels per block
#define S2_NK 320             // link records per block
#define S2_RING 1024          // node score slots (u32): the block being decoded, the one in the chain, the one
                              // being resolved and the position before it are 960 consecutive slots at most
#define S2_BIAS (1u << 17)     // score + bias > 0: a link scores >= -2 - coverage (coverage <= 65535)
#define S2_FLOOR (S2_BIAS - 2u)   // the reference's -1, in half units
// LDS layout, byte offsets (everything a record addresses lies below 64 KB)
#define S2_DUMP_B (S2_RING * 4)           // 64 words: one per lane, for the lanes beyond a level's links
#define S2_ZERO_B (S2_DUMP_B + 64 * 4)    // one word holding score 0 (start links, falcon.c:434)
#define S2_REC_SZ ((S2_NK + 64) * 8)      // records of a block, 8 bytes per link: S2_NK + 64 of them
#define S2_REC_B(b) (S2_ZERO_B + 64 + (b) * S2_REC_SZ)   // two sets: the chain's and the other wavefront's
#define S2_MARK_B S2_REC_B(2)             // 64 words: level slot -> position of the block
#define S2_NLV_B(b) (S2_MARK_B + 64 * 4 + (b) * 64 * 4)  // links of the block's levels, two sets
#define S2_CTL_B(b) (S2_NLV_B(2) + (b) * 16)             // per set: levels of the block, what to do with it
#define S2_DONE_B S2_CTL_B(2)
#define S2_LDS_WORDS ((S2_DONE_B + 16) / 4)
#define S2_GO 1u
#define S2_NONE 0u

__device__ __forceinline__ u32 &s2_at(u32 *L, u32 byte) {
    return *reinterpret_cast<u32 *>(reinterpret_cast<char *>(L) + byte);
}

typedef u32 s2_u32x2 __attribute__((ext_vector_type(2)));

// what the decoding wavefront keeps of a block for its resolve phase, two blocks later
struct S2Block {
    int nl, off, maxn, n_l;
    u32 node5, rnode5, ring5_s, base5_s, upper_s;
};

// ---- B: the chain (wavefront 0).  Lanes 0 .. n_i - 1 hold the links of the level being scored:
// the predecessor's score, the link's own addend, and an LDS atomic maximum on the node's slot --
// the links of a node meet there in any order, and the next level's reads are behind it in the
// LDS queue.  (The record of the next level is requested by every lane, behind the gather; lanes
// beyond a level's links read records of later levels or the null records at the end, and do
// nothing with them.)
__device__ __forceinline__ void s2_chain(u32 *L, u32 rec_b, int n_l, int nl, int lane) {
    u32 ra = rec_b + 8u * (u32)lane;
    const u32 dump = (u32)(S2_DUMP_B + 4 * lane);
    s2_u32x2 r = *reinterpret_cast<const s2_u32x2 *>(&s2_at(L, ra));
    // (consumed here, so that the loop head need not wait for it -- a wait there is also a wait
    // for the atomic of the level before, on every level)
    r.x = fa_settled(r.x); r.y = fa_settled(r.y);
#if defined(S2_SETTLE_NL)
    // (round 6, a build switch, off until it has run on a GPU: the levels' link counts, which the caller read out of
    // the LDS a moment ago, consumed here too.  Left pending, their wait lands at the loop's head -- the compiler
    // cannot tell the first trip from the others -- where from the second trip on it is a wait for the ATOMIC of
    // the level before: the listing has `s_waitcnt lgkmcnt(0)` twice per level, i.e. the second LDS round trip this
    // loop was written to avoid.  With this line the one at the head is gone.)
    nl = (int)fa_settled((u32)nl);
#endif
    for (int i = 0; i < n_l; i++) {
        const int n_i = __builtin_amdgcn_readlane(nl, i);
        ra += 8u * (u32)n_i;
        const u64 here = fa_lane_range(0, n_i);  // (n_i <= 63)
        const s2_u32x2 rn = *reinterpret_cast<const s2_u32x2 *>(&s2_at(L, ra));
        // (no branch around the level's lanes: the others aim at their dump word -- a join here
        // would make the compiler wait for the atomic before the next level starts)
        const bool mine = __builtin_amdgcn_inverse_ballot_w64(here);
        const u32 ph = s2_at(L, mine ? (r.x & 0xffffu) : (u32)S2_ZERO_B);
        atomicMax(&s2_at(L, mine ? (r.x >> 16) : dump), ph + r.y);
        fa_wave_sync();
        r = rn;
    }
}

__global__ __launch_bounds__(128) void k_score2(MsaArgs A) {
    __shared__ __attribute__((aligned(8))) u32 L[S2_LDS_WORDS];
    const int lane = fa_lane();
    const int role = (int)(threadIdx.x >> 6);  // 0: the chain; 1: everything else
    const int p = blockIdx.x;
    if (p >= A.n_pile) return;
    const FaPile pm = A.pile[p];
    FaScoreOut so = A.score_out[p];
    so.err = fa_uni(so.err); so.wide = fa_uni(so.wide);
    so.n_levels = fa_uni(so.n_levels); so.n_links = fa_uni(so.n_links);
    if (so.err) return;
    so.redo = 1;
    // (no score exceeds the sum over the positions of coverage x levels, which k_links2 left)
    so.wide = fa_uni((u32)(A.bound[p] >= (unsigned long long)SC_FAST_SCORE_MAX ? 1 : 0));
    if (so.wide || A.first_links_back) {
        if (role == 1) A.score_out[p] = so;  // every lane stores the same record
        return;
    }
    if (role == 0) {
        // ======== wavefront 0: the chain of block k - 1 in step k, on the records the other
        // wavefront decoded in step k - 1
        for (u32 k = 0;; k++) {
            const u32 b = (k + 1u) & 1u;  // (block k - 1's set)
            if (k >= 1u && fa_uni(s2_at(L, S2_CTL_B(b) + 4u)) == S2_GO)
                s2_chain(L, S2_REC_B(b), (int)fa_uni(s2_at(L, S2_CTL_B(b))), (int)s2_at(L, S2_NLV_B(b) + 4u * (u32)lane), lane);
            __syncthreads();
            if (fa_uni(s2_at(L, S2_DONE_B)) != 0u) return;
        }
    }
    // ======== wavefront 1: in step k, resolve block k - 2 (its chain ran in step k - 1), then
    // stage and decode block k into the set that block's records just left
    const int T = pm.seed_len;
    const u32 *tiw = reinterpret_cast<const u32 *>(A.tinfo + A.t_off[p]);  // 3 words per position
    const u32 *links = A.links + A.link_off[p];
    const u16 *nlk = A.lvl_nlink16 + pm.node_off / 5;
    s2_u32x2 *nodes = reinterpret_cast<s2_u32x2 *>(A.nodes + pm.node_off);
    const u32 min_cov = A.min_cov;

    s2_at(L, S2_ZERO_B) = S2_BIAS;  // (every lane, the same word)
    s2_at(L, S2_DONE_B) = 0u;
    s2_at(L, S2_CTL_B(0) + 4u) = S2_NONE;
    s2_at(L, S2_CTL_B(1) + 4u) = S2_NONE;
    if (lane < 5) {
        // slot 0 = (t 0, delta 0): the target of the zero back pointer (Q4), whichever of its
        // nodes exist
        s2_u32x2 r;
        r.x = (u32)-2; r.y = (u32)(((0 + 1) << 1) | (((tiw[2] & 0xffffu) > min_cov) ? 1 : 0));
        nodes[lane] = r;
    }
    u32 best_s = S2_FLOOR;         // this lane's best node: biased score, node id, winning link
    int best_node = -1, best_ck = 0;
    u32 carry_plvl = 0;            // first level slot of the position before the block,
    u32 carry_nlev = 0;            // and its levels
    // The scores' ring is addressed by a running count of the levels scored so far (times five,
    // plus the base) -- NOT by the level slot: slots jump from segment to segment (every segment
    // has room to spare behind its levels), the count does not, so the previous position's nodes
    // and a block's own are 480 consecutive ring words wherever they lie in the node pool.
    u32 rn = 0;                    // ring index of the block's first node
    u32 lk_run = 0;                // the block's first link word: k_links writes the links of a segment of TSEG
                                   // positions back to back from the segment's first link slot
    const bool slot0_empty = (tiw[2] & 0xffffu) == 0u;  // position 0 uncovered: its level slot has no links, and no count
    const u32 link_end = (u32)A.link_cap[p];             // the pile's link slots
    // What a block reads from HBM -- the records of positions t0 + lane, the link counts of level
    // slots lvl0 + lane, S2_NK link words from lk_run on -- is requested while the block BEFORE
    // it is decoded: where it starts is known as soon as that block's extent is.
    struct Ahead { u32 x_lvl, x_link, x_cn, nl, lw[S2_NK / 64]; };
    auto request = [&](int t_first, u32 lvl_first, u32 lk_first) {
        Ahead a;
        const int tl = t_first + lane;
        a.x_lvl = 0; a.x_link = 0; a.x_cn = 0; a.nl = 0;
        if (tl < T) { a.x_lvl = tiw[3 * tl]; a.x_link = tiw[3 * tl + 1]; a.x_cn = tiw[3 * tl + 2]; }
        if (lvl_first + (u32)lane < (u32)so.n_levels && !(slot0_empty && lvl_first + (u32)lane == 0u))
            a.nl = (u32)nlk[lvl_first + (u32)lane];
#pragma unroll
        for (int q = 0; q < S2_NK / 64; q++) {
            const u32 i = lk_first + (u32)(64 * q + lane);
            a.lw[q] = i < link_end ? links[i] : 0u;
        }
        return a;
    };
    // ---- C: resolve a block.  Lane = level; its links in stored order.  A node's winner is the
    // first of its links that reaches the node's score, if that score beat the floor.  Four links
    // at a time: their records, then the two scores each of them names, then the four in order
    // (one link at a time was two LDS round trips per link with nothing to do meanwhile).
    auto resolve = [&](const S2Block &g, u32 rec_b) {
        u32 seen = 0;   // nodes (bit = base) met / resolved so far
        u32 won = 0;
        u32 cin = 0;    // links met per node, 6 bits each
        for (int k0 = 0; k0 < g.maxn; k0 += 4) {
            s2_u32x2 rec[4];
            u32 sn[4], hs[4];
#pragma unroll
            for (int q = 0; q < 4; q++)  // (a lane beyond its links reads some record of the set: never used)
                rec[q] = *reinterpret_cast<const s2_u32x2 *>(&s2_at(L, rec_b + 8u * (u32)(k0 + q < g.nl ? g.off + k0 + q : lane)));
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const bool on = k0 + q < g.nl;
                sn[q] = s2_at(L, on ? rec[q].x >> 16 : (u32)S2_ZERO_B);
                hs[q] = s2_at(L, on ? rec[q].x & 0xffffu : (u32)S2_ZERO_B);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (k0 + q < g.nl) {
                    const u32 w0 = rec[q].x, cv = rec[q].y;
                    const u32 src = w0 & 0xffffu, dst = w0 >> 16;
                    const u32 nbase = ((dst >> 2) - g.rnode5) & (S2_RING - 1u);
                    const u32 node = g.node5 + nbase;
                    const u32 bit = 1u << nbase;
                    const u32 h = hs[q] + cv;
                    const u32 ck = (cin >> (6u * nbase)) & 63u;
                    cin += 1u << (6u * nbase);
                    const bool wins = !(won & bit) && sn[q] > S2_FLOOR && h == sn[q];
                    const bool stays = !(seen & bit) && sn[q] <= S2_FLOOR;  // no link beats the floor: the zero back pointer (Q4)
                    seen |= bit;
                    if (wins || stays) {
                        won |= bit;
                        // (the predecessor's node id from its ring index: its index at its position is the difference)
                        const int pid = (wins && src != (u32)S2_ZERO_B) ? (int)(g.base5_s + (((src >> 2) - g.ring5_s) & (S2_RING - 1u)))
                                                                        : (wins ? -1 : 0);
                        s2_u32x2 r;
                        r.x = sn[q] - S2_BIAS;
                        r.y = (u32)((pid + 1) << 1) | g.upper_s;
                        nodes[node] = r;
                        // the first strict maximum in (t, delta, base) order: among equals the lowest node
                        if (sn[q] > best_s || (sn[q] == best_s && sn[q] > S2_FLOOR && (int)node < best_node)) {
                            best_s = sn[q];
                            best_node = (int)node;
                            best_ck = wins ? (int)ck : 0;
                        }
                    }
                }
            }
        }
    };
    int t0 = 0;
    Ahead cur = request(0, 0u, tiw[1]);
    S2Block g_prev, g_last;  // blocks k - 2 and k - 1 of step k
    g_prev.maxn = g_last.maxn = 0; g_prev.nl = g_last.nl = 0;
    bool have_prev = false, have_last = false;
    bool bail = false;
    for (u32 k = 0;; k++) {
        const u32 b = k & 1u;  // block k's set = block k - 2's
        if (have_prev) resolve(g_prev, S2_REC_B(b));
        S2Block g;
        g.maxn = 0; g.nl = 0; g.n_l = 0; g.off = 0; g.node5 = g.rnode5 = g.ring5_s = g.base5_s = g.upper_s = 0;
        bool have = false;
        if (t0 < T && !bail) {
            // ---- a block: as many positions (<= 63) of one k_links segment as fit S2_NL levels and
            // S2_NK links.  Lane j holds position t0 + j (a block of j positions ends where position
            // t0 + j starts) and, as a level lane, the link count of level slot lvl0 + j.
            const int tl = t0 + lane;
            const u32 x_lvl = cur.x_lvl, x_link = cur.x_link, x_cn = cur.x_cn;
            const u32 lvl0 = fa_uni(x_lvl);
            if ((t0 & (TSEG - 1)) == 0) lk_run = fa_uni(x_link);
            const int seg_end = (t0 / TSEG + 1) * TSEG;
            int nl = (int)cur.nl;  // links of level slot lvl0 + lane
            const int nl_sum = wave_incl_sum(nl, lane);
            // levels of the first `lane` positions: up to the end of the position before mine (the
            // level slots of a segment are contiguous; behind its last position lies a gap)
            const u32 end_j = x_lvl + (x_cn >> 16);
            const int lv_j = (int)((u32)__builtin_amdgcn_update_dpp((int)lvl0, (int)end_j, 0x138, 0xf, 0xf, false) - lvl0);  // wave_shr:1
            const int lk_j = __shfl(nl_sum, max(0, min(lv_j, 64) - 1));               // and their links
            const bool fits = lane >= 1 && tl <= T && tl <= seg_end && lv_j <= S2_NL && (lv_j == 0 || lk_j <= S2_NK);
            const int nb = __popcll(__ballot(fits));  // prefix sums are monotone, so is `fits`
            const int n_l = nb > 0 ? __builtin_amdgcn_readlane(lv_j, nb & 63) : 0;
            const int n_k = n_l > 0 ? __builtin_amdgcn_readlane(nl_sum, max(0, n_l - 1)) : 0;
            if (lane >= n_l) nl = 0;
            const int maxn = fa_wave_max(nl);
            if (nb == 0 || maxn > 63) {
                bail = true;  // one position too large for a block, a level with too many links: k_score1's
            } else {
                // ---- stage the link words (the high word of each record), null records behind them
                const u32 rec_b = S2_REC_B(b);
                fa_wave_sync();  // (the resolve phase above is done with this set of records)
#pragma unroll
                for (int q = 0; q < S2_NK / 64; q++)
                    if (64 * q + lane < n_k) s2_at(L, rec_b + 8u * (u32)(64 * q + lane) + 4u) = cur.lw[q];
                {   // the next block's extent is known: request its data
                    const int t_nx = t0 + nb;
                    // (inside a segment the next block's levels follow this one's; a new segment has its own first slot)
                    const u32 lvl_nx = t_nx < T ? (u32)__builtin_amdgcn_readlane((int)x_lvl, nb & 63) : lvl0 + (u32)n_l;
                    u32 lk_nx = lk_run + (u32)n_k;
                    if ((t_nx & (TSEG - 1)) == 0) lk_nx = (u32)__builtin_amdgcn_readlane((int)x_link, nb & 63);  // (a new segment: its own link slot)
                    cur = request(t_nx, lvl_nx, lk_nx);
                }
                s2_at(L, rec_b + 8u * (u32)(n_k + lane)) = (u32)S2_ZERO_B | ((u32)(S2_DUMP_B + 4 * lane) << 16);
                s2_at(L, rec_b + 8u * (u32)(n_k + lane) + 4u) = 0u;
                s2_at(L, S2_MARK_B + 4u * (u32)lane) = (u32)-1;
                s2_at(L, S2_NLV_B(b) + 4u * (u32)lane) = (u32)nl;
                s2_at(L, S2_CTL_B(b)) = (u32)n_l;
                fa_wave_sync();
                // ---- which position does level slot lvl0 + lane belong to: every position with levels
                // marks its first slot, the slots take the last mark at or below them
                const int nlev_j = (int)(x_cn >> 16);
                if (lane < nb && nlev_j > 0) s2_at(L, S2_MARK_B + 4u * (x_lvl - lvl0)) = (u32)lane;
                fa_wave_sync();
                const int js = max(0, wave_incl_max((int)s2_at(L, S2_MARK_B + 4u * (u32)lane), lane));
                const u32 plvl_j = (u32)__builtin_amdgcn_update_dpp((int)carry_plvl, (int)x_lvl, 0x138, 0xf, 0xf, false);  // wave_shr:1
                const u32 lvl_s = (u32)__shfl((int)x_lvl, js);
                const u32 cn_s = (u32)__shfl((int)x_cn, js);
                const u32 plvl_s = (u32)__shfl((int)plvl_j, js);
                const int cov_s = (int)(cn_s & 0xffffu);
                const int dl_s = (int)(lvl0 + (u32)lane - lvl_s);
                g.base5_s = (dl_s == 0 ? plvl_s : lvl_s) * 5u;  // node id of a link's predecessor = base5 + its index
                // ... and its ring index = ring5 + that index: the position before the block's first
                // one ends right below the block, whatever its slots are
                g.ring5_s = rn + 5u * (dl_s == 0 ? (js == 0 ? 0u - carry_nlev : plvl_s - lvl0) : lvl_s - lvl0);
                g.upper_s = (u32)cov_s > min_cov ? 1u : 0u;     // falcon.c:498 (Q7)
                g.node5 = (lvl0 + (u32)lane) * 5u;  // my level's first node, and its ring index
                g.rnode5 = rn + 5u * (u32)lane;
                g.off = nl_sum - nl;  // (lanes < n_l)
                g.nl = nl; g.maxn = maxn; g.n_l = n_l;
                carry_plvl = (u32)__builtin_amdgcn_readlane((int)x_lvl, nb - 1);
                carry_nlev = (u32)__builtin_amdgcn_readlane((int)x_cn, nb - 1) >> 16;
                // ---- A: decode.  Lane = level, k = its k-th link (any order of the nodes; insertion
                // order inside a node: k_links).  Every node of the block starts at the floor.
                if (lane < n_l) {
#pragma unroll
                    for (u32 bb = 0; bb < 5u; bb++) s2_at(L, ((g.rnode5 + bb) & (S2_RING - 1u)) << 2) = S2_FLOOR;
                }
                for (int k0 = 0; k0 < maxn; k0 += 4) {  // (four link words per LDS round trip)
                    u32 w4[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) w4[q] = s2_at(L, rec_b + 8u * (u32)(k0 + q < nl ? g.off + k0 + q : lane) + 4u);
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        if (k0 + q < nl) {
                            const u32 ra = rec_b + 8u * (u32)(g.off + k0 + q);
                            const u32 w = w4[q];
                            const int cnt = (int)(w & LW_CNT_MASK);
                            const u32 nbase = (w >> LW_NB_SHIFT) & 7u;
                            const u32 pidx = (w >> LW_PIDX_SHIFT) & 0x7ffu;
                            const bool start = (w >> LW_START_BIT) & 1u;
                            const u32 src = start ? (u32)S2_ZERO_B : (((g.ring5_s + pidx) & (S2_RING - 1u)) << 2);
                            const u32 dst = ((g.rnode5 + nbase) & (S2_RING - 1u)) << 2;
                            s2_u32x2 r;
                            r.x = src | (dst << 16);
                            r.y = (u32)(2 * cnt - cov_s);  // falcon.c:440-445, half units
                            *reinterpret_cast<s2_u32x2 *>(&s2_at(L, ra)) = r;
                        }
                    }
                }
                t0 += nb;
                lk_run += (u32)n_k;
                rn = (rn + 5u * (u32)n_l) & (S2_RING - 1u);
                have = true;
            }
        }
        s2_at(L, S2_CTL_B(b) + 4u) = have ? S2_GO : S2_NONE;
        // the last step: nothing decoded now or in the step before (its chain had nothing to run),
        // block k - 2 resolved just now -- or a block that does not fit this kernel
        const bool done = bail || (!have && !have_last);
        if (done) s2_at(L, S2_DONE_B) = 1u;
        __syncthreads();
        if (done) break;
        g_prev = g_last; have_prev = have_last;
        g_last = g; have_last = have;
    }
    if (bail) {
        A.score_out[p] = so;  // (redo: k_score1 scores the whole pile again)
        return;
    }
    // global best = first strict maximum in (t, delta, base) order (falcon.c:464-469): the
    // highest score, among equals the lowest node id
    const u32 top = (u32)fa_wave_max((int)best_s);
    const int cand = (best_s == top && best_node >= 0) ? best_node : 0x7fffffff;
    const int g_node = fa_wave_min(cand);
    const u64 who = fa_ballot(cand == g_node);
    so.redo = 0;
    if (g_node == 0x7fffffff) {
        so.g_h = -2; so.g_node = -1; so.g_ck = 0;
    } else {
        so.g_h = (int)(top - S2_BIAS);
        so.g_node = g_node;
        so.g_ck = __builtin_amdgcn_readlane(best_ck, (int)__builtin_ctzll(who));
    }
    A.score_out[p] = so;  // every lane stores the same record
}

#ifndef FA_EMU
void fa_touch_score2() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(k_score2));
}
void fa_launch_score2(const MsaArgs &A, hipStream_t s) {
    hipLaunchKernelGGL(k_score2, dim3(A.n_pile), dim3(128), 0, s, A);
}
#endif
