// k_consensus.hip -- alignment tags -> MSA graph -> best path -> consensus,
// one wavefront per pile, in a single sweep over the seed positions.
//
// Restates get_align_tags() (src/c/falcon.c:106-162) and
// get_cns_from_align_tags() (src/c/falcon.c:308-558) without ever materialising
// the tags (16 B per alignment column in the reference) or the 100000-position
// MSA workspace (:338-341):
//
//   * lane a of the wave owns accepted alignment a of the pile (accepted
//     alignments in read order; 64 per register "chunk", NCH chunks, the kernel
//     is instantiated for NCH = 1,2,4,8 and the host sorts piles into classes)
//     and walks its edit script ((snake << 1) | from_above per row, produced by
//     k_align) column by column.  The script and the read's packed bases are
//     per-lane streams: both are double-buffered in registers (16-byte script
//     quads, 16-base words) so no load sits on the critical path of a column;
//   * the sweep visits target positions t in ascending order and, inside one t,
//     the insertion levels delta = 0,1,2,..  -- the order of the reference's
//     scoring loops (:405-407).  At one (t, delta) level every participating
//     lane holds exactly one tag: its base and the node of its previous column.
//     Lanes with equal (base, previous node) are one *link* of the reference
//     (update_col, :232-263): they are grouped with __ballot, the group size is
//     the link count, and groups are visited in order of their lowest lane,
//     which is the reference's first-insertion order of links (Q5);
//   * node score = max over its links of score(prev) + count - coverage/2 with
//     the -1 floor and strict '>' (:420-462), kept in half units (Q6); lanes
//     0..4 accumulate the five nodes (A,C,G,T,-) of the level and write them as
//     8-byte records; each lane then carries its node's id and score forward,
//     which is all the "graph" the forward pass needs;
//   * the global best node (first strict maximum in (t,delta,base) order,
//     :464-469) and the link index of its best link (Q2) are tracked in scalars;
//   * the back-trace (:494-528) walks the 8-byte node records through a
//     64-level window staged in LDS (back pointers reach only a few levels
//     back) and writes the consensus right-aligned, 64 characters per store,
//     so no reversal pass (:531-539) is needed.
//
// Per-column reduction over aligned bases; integer only, no MFMA.  The forward
// sweep is a dependent chain over t (like the reference), so throughput comes
// from running one wave per pile over thousands of piles.
#include "fa_device.h"

struct CnsArgs {
    const u32 *words;
    const FaSeq *seq;
    const FaPile *pile;
    const FaRange *range;
    const FaAln *aln;
    const u32 *script;
    const u64 *script_off;
    FaNode *nodes;
    char *out_seq;
    int *out_eqv;
    FaPileOut *pile_out;
    const int *pile_list;  // piles of this NCH class
    int n_list;
    unsigned min_cov;
};

#define BT_WIN 64  // levels per back-trace window

// entry j of a lane's script through the two resident quads
__device__ __forceinline__ u32 quad_get(const uint4 &q, int k) {
    return k == 0 ? q.x : (k == 1 ? q.y : (k == 2 ? q.z : q.w));
}

template <int NCH>
__device__ void consensus_pile(const CnsArgs &A, const FaPile pm, int p, int n_acc,
                               const int *acc, u32 *win, int lane) {
    const int T = pm.seed_len;
    const u32 *seedw = A.words + A.seq[pm.first].woff;
    FaNode *nodes = A.nodes + pm.node_off;
    const long long node_cap = (long long)pm.node_cap;

    bool valid[NCH], started[NCH], finished[NCH];
    int s2[NCH], dist[NCH], sp[NCH], mrem[NCH], qpos[NCH], pnode[NCH], pscore[NCH];
    u32 nxt[NCH];
    // script stream: quads [qb, qb+4) and [qb+4, qb+8)
    const uint4 *scr4[NCH];
    uint4 q0[NCH], q1[NCH];
    int qb[NCH];
    // read stream: words wb and wb+1
    const u32 *rw[NCH];
    u32 r0[NCH], r1[NCH];
    int wb[NCH];

    int t0 = 0x7fffffff;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int a = c * 64 + lane;
        valid[c] = a < n_acc;
        started[c] = false;
        finished[c] = !valid[c];
        s2[c] = 0x7fffffff; dist[c] = 0; sp[c] = 0; mrem[c] = 0; qpos[c] = 0;
        pnode[c] = -1; pscore[c] = 0; nxt[c] = 0;
        scr4[c] = reinterpret_cast<const uint4 *>(A.script);
        rw[c] = A.words;
        q0[c] = make_uint4(0, 0, 0, 0); q1[c] = q0[c]; qb[c] = 0;
        r0[c] = r1[c] = 0; wb[c] = 0;
        if (valid[c]) {
            const int g = acc[a];
            const FaRange rg = A.range[g];
            const FaAln al = A.aln[g];
            scr4[c] = reinterpret_cast<const uint4 *>(A.script + A.script_off[g]);
            rw[c] = A.words + A.seq[g].woff;
            dist[c] = al.dist;
            q0[c] = scr4[c][0];
            if (al.dist >= 4) q1[c] = scr4[c][1];
            mrem[c] = (int)(q0[c].x >> 1);
            nxt[c] = (al.dist > 0) ? q0[c].y : 0u;
            qpos[c] = rg.s1;   // falcon.c:119 (i = s1-1 before the first column)
            s2[c] = rg.s2;     // falcon.c:120
            wb[c] = rg.s1 >> 4;
            r0[c] = rw[c][wb[c]];
            r1[c] = rw[c][wb[c] + 1];
            if (mrem[c] == 0 && al.dist == 0) finished[c] = true;
        }
        t0 = min(t0, s2[c]);
    }
    t0 = fa_wave_min(t0);

    // consume one script row for chunk c (the lane is at row sp, moves to sp+1)
#define ADVANCE_ROW(c)                                                            \
    do {                                                                          \
        sp[c]++;                                                                  \
        mrem[c] = (int)(nxt[c] >> 1);                                             \
        const int j_ = sp[c] + 1;                                                 \
        if (j_ <= dist[c]) {                                                      \
            if (j_ >= qb[c] + 4) { /* slide the window; prefetch the next quad */ \
                q0[c] = q1[c];                                                    \
                qb[c] += 4;                                                       \
                if (qb[c] + 4 <= dist[c]) q1[c] = scr4[c][(qb[c] >> 2) + 1];      \
            }                                                                     \
            nxt[c] = quad_get(q0[c], j_ - qb[c]);                                 \
        } else {                                                                  \
            nxt[c] = 0u;                                                          \
        }                                                                         \
        finished[c] = (mrem[c] == 0 && sp[c] == dist[c]);                         \
    } while (0)

    // consume one query base for chunk c
#define ADVANCE_QUERY(c)                                                          \
    do {                                                                          \
        qpos[c]++;                                                                \
        if ((qpos[c] >> 4) != wb[c]) {                                            \
            wb[c]++;                                                              \
            r0[c] = r1[c];                                                        \
            r1[c] = rw[c][wb[c] + 1];                                             \
        }                                                                         \
    } while (0)

    int lvl = 0;
    int g_h = -2, g_node = -1, g_ck = 0;
    bool overflow = false;
    if (t0 > 0) {  // node 0 must be (t_pos 0, delta 0, 'A'): the zero back pointer's target (Q4)
        if (lane < 5) {
            FaNode nd;
            nd.score_h = -2;
            nd.link = (0 + 1) << 1;
            nodes[lane] = nd;
        }
        lvl = 1;
    }

    int t = t0;
    u32 seed_word = 0;
    int seed_word_idx = -1;
    while (t < T && !overflow) {
        bool part[NCH];
        int key[NCH];  // (previous node + 1) * 8 + base: one link per distinct key
        int cov = 0;
        bool any_open = false;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            started[c] = started[c] || (valid[c] && s2[c] == t);
            part[c] = started[c] && !finished[c];
            cov += __popcll(__ballot(part[c]));
            any_open = any_open || (__ballot(valid[c] && !finished[c]) != 0ull);
        }
        if (!any_open) break;
        if (cov == 0) {  // coverage gap: jump to the next alignment start
            int nt = 0x7fffffff;
#pragma unroll
            for (int c = 0; c < NCH; c++)
                if (valid[c] && !started[c]) nt = min(nt, s2[c]);
            nt = fa_wave_min(nt);
            if (nt >= T) break;
            t = nt;
            continue;
        }
        if ((t >> 4) != seed_word_idx) {
            seed_word_idx = t >> 4;
            seed_word = seedw[seed_word_idx];
        }
        const int sbase = (int)((seed_word >> ((t & 15) * 2)) & 3u);
        const int upper = (cov > (int)A.min_cov) ? 1 : 0;  // falcon.c:498 (Q7)

        // ---- level delta = 0: every participating alignment consumes target base t
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            key[c] = 0;
            if (part[c]) {
                int base;
                if (mrem[c] > 0) {  // match column
                    base = sbase;
                    mrem[c]--;
                    ADVANCE_QUERY(c);
                    finished[c] = (mrem[c] == 0 && sp[c] == dist[c]);
                } else {            // target-only edit: query shows '-'
                    base = 4;
                    ADVANCE_ROW(c);
                }
                key[c] = ((pnode[c] + 1) << 3) | base;
            }
        }
        int delta = 0;
        for (;;) {
            // ---- one (t, delta) level: group lanes into links, score the 5 nodes
            if ((long long)(lvl + 1) * 5 > node_cap) {
                overflow = true;
                break;
            }
            int nb_h = -2, nb_prev = 0, nb_n = 0, nb_ck = 0;
            u64 rem[NCH];
#pragma unroll
            for (int c = 0; c < NCH; c++) rem[c] = __ballot(part[c]);
            for (;;) {
                int c0 = -1;
#pragma unroll
                for (int c = NCH - 1; c >= 0; c--)
                    if (rem[c]) c0 = c;
                if (c0 < 0) break;
                int kk = 0, ks = 0;
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    if (c == c0) {
                        const int l0 = __ffsll((long long)rem[c]) - 1;
                        kk = __builtin_amdgcn_readlane(key[c], l0);
                        ks = __builtin_amdgcn_readlane(pscore[c], l0);
                    }
                }
                int cnt = 0;
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    const u64 m = __ballot(part[c] && key[c] == kk);
                    cnt += __popcll(m);
                    rem[c] &= ~m;
                }
                const int kb = kk & 7, kp = (kk >> 3) - 1;
                const int h = ((kp < 0) ? 0 : ks) + 2 * cnt - cov;  // falcon.c:440-445
                if (lane == kb) {
                    if (h > nb_h) {  // strict, first maximum in link order (:447)
                        nb_h = h;
                        nb_prev = kp;
                        nb_ck = nb_n;
                    }
                    nb_n++;
                }
            }
            const int node0 = lvl * 5;
            int hs[5];
#pragma unroll
            for (int b = 0; b < 5; b++) {
                hs[b] = __builtin_amdgcn_readlane(nb_h, b);
                if (hs[b] > g_h) {  // :464-469
                    g_h = hs[b];
                    g_node = node0 + b;
                    g_ck = __builtin_amdgcn_readlane(nb_ck, b);
                }
            }
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                if (part[c]) {
                    const int b = key[c] & 7;
                    pscore[c] = b == 0 ? hs[0] : (b == 1 ? hs[1] : (b == 2 ? hs[2] : (b == 3 ? hs[3] : hs[4])));
                    pnode[c] = node0 + b;
                }
            }
            if (lane < 5) {
                FaNode nd;
                nd.score_h = nb_h;
                nd.link = ((nb_prev + 1) << 1) | upper;
                nodes[node0 + lane] = nd;
            }
            lvl++;

            // ---- next insertion level of the same t: query-only edits
            bool any_ins = false;
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                part[c] = started[c] && !finished[c] && mrem[c] == 0 && sp[c] < dist[c] &&
                          (nxt[c] & 1u) == 0u;
                any_ins = any_ins || (__ballot(part[c]) != 0ull);
            }
            if (!any_ins) break;
            delta++;
            if (delta >= 255) {  // tagging stops for these alignments (falcon.c:138-152)
#pragma unroll
                for (int c = 0; c < NCH; c++)
                    if (part[c]) finished[c] = true;
                break;
            }
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                if (part[c]) {
                    const int base = (int)((r0[c] >> ((qpos[c] & 15) * 2)) & 3u);
                    ADVANCE_QUERY(c);
                    ADVANCE_ROW(c);
                    key[c] = ((pnode[c] + 1) << 3) | base;
                }
            }
        }
        t++;
    }
#undef ADVANCE_ROW
#undef ADVANCE_QUERY

    // ---- back-trace (falcon.c:494-528), uniform across the wave -------------
    __threadfence_block();
    FaPileOut po;
    po.len = 0; po.start = 2 * T; po.n_aligned = n_acc; po.err = overflow ? 1 : 0;
    po.g_best_h = g_h;
    if (!overflow && g_node >= 0) {
        char *oseq = A.out_seq + pm.out_off;
        int *oeqv = A.out_eqv + pm.out_off;
        const unsigned lim = (unsigned)T * 2u;
        unsigned index = 0;
        int ck = g_ck;
        char bb = '$';
        int win_lo = -1, win_hi = -2;  // node-id range [win_lo, win_hi] resident in LDS
        auto fetch = [&](int node) -> FaNode {
            if (node < win_lo || node > win_hi) {
                const int lvl_hi = node / 5;
                const int lvl_lo = max(0, lvl_hi - (BT_WIN - 1));
                win_lo = lvl_lo * 5;
                win_hi = lvl_hi * 5 + 4;
                const int n_rec = win_hi - win_lo + 1;
                __syncthreads();
                const uint2 *src = reinterpret_cast<const uint2 *>(nodes + win_lo);
                for (int i = lane; i < n_rec; i += 64) {
                    const uint2 v = src[i];
                    win[2 * i] = v.x;
                    win[2 * i + 1] = v.y;
                }
                __syncthreads();
            }
            FaNode r;
            r.score_h = (int)win[2 * (node - win_lo)];
            r.link = (int)win[2 * (node - win_lo) + 1];
            return r;
        };
        FaNode rec = fetch(g_node);
        int out_c = 0, out_e = 0;  // lane (index & 63) holds character `index`
        for (;;) {
            const int up = rec.link & 1;
            switch (ck) {
            case 0: bb = up ? 'A' : 'a'; break;
            case 1: bb = up ? 'C' : 'c'; break;
            case 2: bb = up ? 'G' : 'g'; break;
            case 3: bb = up ? 'T' : 't'; break;
            case 4: bb = '-'; break;
            default: break;  // a link index >= 5 keeps the previous character (Q2)
            }
            const int score0 = rec.score_h;
            const int prev = (rec.link >> 1) - 1;
            if (prev == -1 || index >= lim) break;  // :517-519 (Q1)
            ck = prev % 5;
            rec = fetch(prev);
            if (bb != '-') {
                if (lane == (int)(index & 63u)) {
                    out_c = bb;
                    out_e = score0 / 2 - rec.score_h / 2;  // (int) truncations (Q6)
                }
                index++;
                if ((index & 63u) == 0u) {  // 64 characters ready: one coalesced store
                    const unsigned pos = lim - 1u - (index - 64u + (unsigned)lane);
                    oseq[pos] = (char)out_c;
                    oeqv[pos] = out_e;
                }
            }
        }
        if ((index & 63u) != 0u && lane < (int)(index & 63u)) {
            const unsigned pos = lim - 1u - ((index & ~63u) + (unsigned)lane);
            oseq[pos] = (char)out_c;
            oeqv[pos] = out_e;
        }
        po.len = (int)index;
        po.start = (int)(lim - index);
    }
    A.pile_out[p] = po;  // every lane stores the same record
}

// second launch-bound = waves per SIMD the register allocator must allow: the
// 1-chunk sweep fits 6, wider ones take what their state needs
template <int NCH>
__global__ __launch_bounds__(64, (NCH == 1 ? 6 : 1)) void k_consensus(CnsArgs A) {
    __shared__ int acc[NCH * 64];
    __shared__ u32 win[2 * 5 * BT_WIN];
    const int lane = fa_lane();
    const int p = __builtin_amdgcn_readfirstlane(A.pile_list[blockIdx.x]);
    const FaPile pm = A.pile[p];
    // accepted alignments in read order (falcon.c:597,:630-636)
    int n_acc = 0;
    for (int j0 = 1; j0 < pm.n_seq; j0 += 64) {
        const int j = j0 + lane;
        const bool ok = (j < pm.n_seq) && A.aln[pm.first + j].accept;
        const u64 m = __ballot(ok);
        const int rank = __popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
        if (ok && n_acc + rank < NCH * 64) acc[n_acc + rank] = pm.first + j;
        n_acc += __popcll(m);
    }
    __syncthreads();
    if (n_acc == 0 || n_acc > NCH * 64) {
        FaPileOut po;
        po.len = 0; po.start = 2 * pm.seed_len; po.n_aligned = n_acc;
        po.err = (n_acc > NCH * 64) ? 2 : 0;
        po.g_best_h = -2;
        A.pile_out[p] = po;
        return;
    }
    consensus_pile<NCH>(A, pm, p, n_acc, acc, win, lane);
}

void fa_launch_consensus(const FaBatchDev &b, unsigned min_cov, const int *const d_list[4],
                         const int n_list[4], hipStream_t s) {
    CnsArgs A;
    A.words = b.words; A.seq = b.seq; A.pile = b.pile; A.range = b.range; A.aln = b.aln;
    A.script = b.script; A.script_off = b.script_off; A.nodes = b.nodes;
    A.out_seq = b.out_seq; A.out_eqv = b.out_eqv; A.pile_out = b.pile_out;
    A.min_cov = min_cov;
    for (int k = 0; k < 4; k++) {
        if (n_list[k] == 0) continue;
        A.pile_list = d_list[k];
        A.n_list = n_list[k];
        switch (k) {
        case 0: hipLaunchKernelGGL(k_consensus<1>, dim3(n_list[k]), dim3(64), 0, s, A); break;
        case 1: hipLaunchKernelGGL(k_consensus<2>, dim3(n_list[k]), dim3(64), 0, s, A); break;
        case 2: hipLaunchKernelGGL(k_consensus<4>, dim3(n_list[k]), dim3(64), 0, s, A); break;
        default: hipLaunchKernelGGL(k_consensus<8>, dim3(n_list[k]), dim3(64), 0, s, A); break;
        }
    }
}
