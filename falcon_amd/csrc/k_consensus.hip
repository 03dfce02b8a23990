// k_consensus.hip -- alignment tags -> MSA graph -> best path -> consensus,
// one wavefront per pile, in a single sweep over the seed positions.
//
// Restates get_align_tags() (src/c/falcon.c:106-162) and
// get_cns_from_align_tags() (src/c/falcon.c:308-558) without ever materialising
// the tags (16 B per alignment column in the reference) or the 100000-position
// MSA workspace (:338-341):
//
//   * lane a of the wave owns accepted alignment a of the pile (accepted
//     alignments in read order; 64 per register "chunk", up to 8 chunks) and
//     walks its edit script ((snake << 1) | from_above per row, produced by
//     k_align) column by column;
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
//   * the back-trace (:494-528) walks the 8-byte node records and writes the
//     consensus right-aligned, so no reversal pass (:531-539) is needed.
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
    unsigned min_cov;
};

template <int NCH>
__device__ void consensus_pile(const CnsArgs &A, const FaPile pm, int p, int n_acc,
                               const int *acc, int lane) {
    const int T = pm.seed_len;
    const u32 *seedw = A.words + A.seq[pm.first].woff;
    FaNode *nodes = A.nodes + pm.node_off;
    const long long node_cap = (long long)pm.node_cap;

    bool valid[NCH], started[NCH], finished[NCH];
    int s2[NCH], dist[NCH], sp[NCH], mrem[NCH], qpos[NCH], pnode[NCH], pscore[NCH];
    u32 nxt[NCH];
    const u32 *scr[NCH];
    const u32 *rw[NCH];

    int t0 = 0x7fffffff;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int a = c * 64 + lane;
        valid[c] = a < n_acc;
        started[c] = false;
        finished[c] = !valid[c];
        s2[c] = 0x7fffffff; dist[c] = 0; sp[c] = 0; mrem[c] = 0; qpos[c] = 0;
        pnode[c] = -1; pscore[c] = 0; nxt[c] = 0; scr[c] = A.script; rw[c] = A.words;
        if (valid[c]) {
            const int g = acc[a];
            const FaRange rg = A.range[g];
            const FaAln al = A.aln[g];
            scr[c] = A.script + A.script_off[g];
            rw[c] = A.words + A.seq[g].woff;
            dist[c] = al.dist;
            mrem[c] = (int)(scr[c][0] >> 1);
            nxt[c] = (al.dist > 0) ? scr[c][1] : 0u;
            qpos[c] = rg.s1;   // falcon.c:119 (i = s1-1 before the first column)
            s2[c] = rg.s2;     // falcon.c:120
            if (mrem[c] == 0 && al.dist == 0) finished[c] = true;
        }
        t0 = min(t0, s2[c]);
    }
    t0 = fa_wave_min(t0);

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
        int base[NCH];
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
            base[c] = 0;
            if (part[c]) {
                if (mrem[c] > 0) {  // match column
                    base[c] = sbase;
                    mrem[c]--;
                    qpos[c]++;
                } else {            // target-only edit: query shows '-'
                    base[c] = 4;
                    sp[c]++;
                    mrem[c] = (int)(nxt[c] >> 1);
                    nxt[c] = (sp[c] < dist[c]) ? scr[c][sp[c] + 1] : 0u;
                }
                finished[c] = (mrem[c] == 0 && sp[c] == dist[c]);
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
                int kb = 0, kp = 0, ks = 0;
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    if (c == c0) {
                        const int l0 = __ffsll((long long)rem[c]) - 1;
                        kb = __builtin_amdgcn_readlane(base[c], l0);
                        kp = __builtin_amdgcn_readlane(pnode[c], l0);
                        ks = __builtin_amdgcn_readlane(pscore[c], l0);
                    }
                }
                int cnt = 0;
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    const u64 m = __ballot(part[c] && base[c] == kb && pnode[c] == kp);
                    cnt += __popcll(m);
                    rem[c] &= ~m;
                }
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
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const int sc = __shfl(nb_h, base[c]);
                if (part[c]) {
                    pscore[c] = sc;
                    pnode[c] = node0 + base[c];
                }
            }
            if (lane < 5) {
                FaNode nd;
                nd.score_h = nb_h;
                nd.link = ((nb_prev + 1) << 1) | upper;
                nodes[node0 + lane] = nd;
            }
#pragma unroll
            for (int b = 0; b < 5; b++) {
                const int hb = __builtin_amdgcn_readlane(nb_h, b);
                if (hb > g_h) {  // :464-469
                    g_h = hb;
                    g_node = node0 + b;
                    g_ck = __builtin_amdgcn_readlane(nb_ck, b);
                }
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
                    base[c] = (int)fa_base_at(rw[c], qpos[c]);
                    qpos[c]++;
                    sp[c]++;
                    mrem[c] = (int)(nxt[c] >> 1);
                    nxt[c] = (sp[c] < dist[c]) ? scr[c][sp[c] + 1] : 0u;
                    finished[c] = (mrem[c] == 0 && sp[c] == dist[c]);
                }
            }
        }
        t++;
    }

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
        FaNode rec = nodes[g_node];
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
            rec = nodes[prev];
            if (bb != '-') {
                const unsigned pos = lim - 1u - index;
                if (lane == 0) {
                    oseq[pos] = bb;
                    oeqv[pos] = score0 / 2 - rec.score_h / 2;  // (int) truncations (Q6)
                }
                index++;
            }
        }
        po.len = (int)index;
        po.start = (int)(lim - index);
    }
    if (lane == 0) A.pile_out[p] = po;
}

__global__ __launch_bounds__(64) void k_consensus(CnsArgs A, int n_pile) {
    __shared__ int acc[FA_CNS_MAX_ALN];
    const int p = blockIdx.x;
    if (p >= n_pile) return;
    const int lane = fa_lane();
    const FaPile pm = A.pile[p];
    // accepted alignments in read order (falcon.c:597,:630-636)
    int n_acc = 0;
    bool too_many = false;
    for (int j0 = 1; j0 < pm.n_seq; j0 += 64) {
        const int j = j0 + lane;
        const bool ok = (j < pm.n_seq) && A.aln[pm.first + j].accept;
        const u64 m = __ballot(ok);
        const int rank = __popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
        if (ok && n_acc + rank < FA_CNS_MAX_ALN) acc[n_acc + rank] = pm.first + j;
        n_acc += __popcll(m);
        if (n_acc > FA_CNS_MAX_ALN) too_many = true;
    }
    __syncthreads();
    if (too_many || n_acc == 0) {
        if (lane == 0) {
            FaPileOut po;
            po.len = 0; po.start = 2 * pm.seed_len; po.n_aligned = n_acc;
            po.err = too_many ? 2 : 0;
            po.g_best_h = -2;
            A.pile_out[p] = po;
        }
        return;
    }
    if (n_acc <= 64) consensus_pile<1>(A, pm, p, n_acc, acc, lane);
    else if (n_acc <= 128) consensus_pile<2>(A, pm, p, n_acc, acc, lane);
    else if (n_acc <= 256) consensus_pile<4>(A, pm, p, n_acc, acc, lane);
    else consensus_pile<8>(A, pm, p, n_acc, acc, lane);
}

void fa_launch_consensus(const FaBatchDev &b, unsigned min_cov, hipStream_t s) {
    if (b.n_pile == 0) return;
    CnsArgs A;
    A.words = b.words; A.seq = b.seq; A.pile = b.pile; A.range = b.range; A.aln = b.aln;
    A.script = b.script; A.script_off = b.script_off; A.nodes = b.nodes;
    A.out_seq = b.out_seq; A.out_eqv = b.out_eqv; A.pile_out = b.pile_out;
    A.min_cov = min_cov;
    hipLaunchKernelGGL(k_consensus, dim3(b.n_pile), dim3(64), 0, s, A, b.n_pile);
}
