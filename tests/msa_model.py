"""TEST INFRASTRUCTURE: a plain-python model of the MSA graph and its score recurrence
(src/c/falcon.c:106-162 get_align_tags, :232-263 update_col, :350-382 graph building, :405-475
scores), written from SURVEY.md Appendix A4/A5 with dictionaries -- slow, obvious, and
independent of both the kernels and the C oracle.  It yields the intermediate products the
kernels hand one another (coverage and levels per position, the link words of every level in
k_links' order, every node's score and back pointer) so that the emulated kernels are compared
stage by stage, not only through the consensus string."""
import collections

_B = {"A": 0, "C": 1, "G": 2, "T": 3, "-": 4, ".": 4}


def tags_of(q_aln, t_aln, s1, s2):
    """falcon.c:106-162: (t_pos, delta, q_base, p_t_pos, p_delta, p_q_base) per column."""
    out = []
    j, jj = s2 - 1, 0
    p_j, p_jj, p_b = -1, 0, "."
    for qc, tc in zip(q_aln, t_aln):
        if qc != "-":
            jj += 1
        if tc != "-":
            j += 1
            jj = 0
        if not (j >= 0 and jj < 255 and p_jj < 255):
            break
        out.append((j, jj, qc, p_j, p_jj, p_b))
        p_j, p_jj, p_b = j, jj, qc
    return out


class Graph:
    def __init__(self, T):
        self.T = T
        self.cov = [0] * T
        # (t, delta) -> base -> {(p_t, p_delta, p_base): count}, links in first-insertion order
        self.lv = collections.defaultdict(lambda: collections.defaultdict(collections.OrderedDict))
        self.order = collections.defaultdict(list)  # (t, delta) -> [(base, link key)] in first-insertion order
        self.max_delta = [0] * T
        self.ins_at = [0] * T   # inserted bases hanging off a position, over all alignments

    def add(self, tags):
        for (t, d, b, pt, pd, pb) in tags:
            if d == 0:
                self.cov[t] += 1
            else:
                self.ins_at[t] += 1
            self.max_delta[t] = max(self.max_delta[t], d)
            node = self.lv[(t, d)][_B[b]]
            key = (pt, pd, _B[pb])
            if key not in node:
                self.order[(t, d)].append((_B[b], key))
            node[key] = node.get(key, 0) + 1

    def layout(self, tseg=128):
        """Level slots and link slots the way k_sscan / k_links2 lay them out: every segment of
        `tseg` positions starts where the one before it would end if every tag were a link of
        its own (links) and every inserted base a level of its own (levels); inside a segment
        the levels follow one another in position order.  -> (first level slot per position,
        the segment's first link slot per position, levels per position, level slots, link slots)"""
        lvl_start, link_start, nlev = [], [], []
        seg_lvl = seg_link = 0
        for s0 in range(0, self.T, tseg):
            ts = range(s0, min(self.T, s0 + tseg))
            ls = seg_lvl
            for t in ts:
                n = (1 + self.max_delta[t]) if self.cov[t] > 0 else (1 if t == 0 else 0)
                lvl_start.append(ls); link_start.append(seg_link); nlev.append(n)
                ls += n
            ins = self._ins_tags(s0, min(self.T, s0 + tseg))
            seg_lvl += len(ts) + ins
            seg_link += sum(self.cov[t] for t in ts) + ins
        return lvl_start, link_start, nlev, seg_lvl, seg_link

    def _ins_tags(self, lo, hi):
        return sum(self.ins_at[t] for t in range(lo, hi))

    def link_words(self, t, d, lvl_start):
        """The level's link words in k_links' order: first-insertion order over the whole level
        (which is insertion order inside every node, falcon.c:245-262; the nodes interleave):
        count | base << 16 | (p_delta * 5 + p_base) << 19 | start << 30."""
        out = []
        for b, (pt, pd, pb) in self.order[(t, d)]:
            c = self.lv[(t, d)][b][(pt, pd, pb)]
            if pt == -1:
                out.append(c | (b << 16) | (1 << 30))
            else:
                out.append(c | (b << 16) | ((pd * 5 + pb) << 19))
        return out

    def scores(self, lvl_start):
        """falcon.c:405-475 in half units: node id -> (score, best predecessor node id or -1 / None
        when nothing beat the floor, index of the winning link), and the global best."""
        sc = {}
        best = (-2, -1, 0)
        for t in range(self.T):
            if self.cov[t] == 0:
                continue
            for d in range(self.max_delta[t] + 1):
                for b in sorted(self.lv[(t, d)]):
                    bs, bp, bk = -2, None, 0
                    for ck, ((pt, pd, pb), c) in enumerate(self.lv[(t, d)][b].items()):
                        if pt == -1:
                            h, pid = 2 * c - self.cov[t], -1
                        else:
                            pid = (lvl_start[pt] + pd) * 5 + pb
                            h = sc[pid][0] + 2 * c - self.cov[t]
                        if h > bs:
                            bs, bp, bk = h, pid, ck
                    nid = (lvl_start[t] + d) * 5 + b
                    sc[nid] = (bs, bp, bk)
                    if bs > best[0]:
                        best = (bs, nid, bk)
        return sc, best
