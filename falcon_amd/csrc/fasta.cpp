// fasta.cpp -- the worker's output rules in native code.
//
// Restates falcon_kit/mains/consensus.py:275-299 (and format_seq, :212-213) for one
// consensus string at a time, so that a batch's FASTA text is produced in one call without
// the interpreter (and its lock) in the loop:
//   * a consensus shorter than 500 is dropped (:276-277);
//   * --output-full: ">" + seed_id + "_f", then the raw string (:279-281);
//   * otherwise the runs of [ACGT] (good_region, :16) -- none: dropped (:283-285);
//   * --output-multi: every run >= 500, at most 10, header ">prolog/<seed_id><i>/0_<len>",
//     sequence wrapped at 80 columns (:286-295);
//   * default: ">" + seed_id and the longest run, the LAST of equally long ones (the stable
//     sort by length at :297 keeps input order among equals, :299 prints the last).
// No HIP in this file: it is usable (and tested) without a GPU.
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/falcon_amd.h"
#include "fa_host.h"

namespace {

inline bool solid(unsigned char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

}  // namespace

void fa_fasta_append(std::string &out, const char *seed_id, const char *cns, long long n, int mode) {
    if (n < 500) return;
    if (mode == FA_FASTA_FULL) {
        out += '>';
        out += seed_id;
        out += "_f\n";
        out.append(cns, (size_t)n);
        out += '\n';
        return;
    }
    long long best_at = -1, best_len = 0;
    int k = 0;
    for (long long i = 0; i < n;) {
        if (!solid((unsigned char)cns[i])) {
            i++;
            continue;
        }
        long long j = i + 1;
        while (j < n && solid((unsigned char)cns[j])) j++;
        const long long len = j - i;
        if (mode == FA_FASTA_MULTI) {
            if (len >= 500) {
                if (k == 10) break;
                char head[64];
                out += ">prolog/";
                out += seed_id;
                snprintf(head, sizeof head, "%d/0_%lld\n", k, len);
                out += head;
                for (long long a = i; a < j; a += 80) {
                    out.append(cns + a, (size_t)((j - a < 80) ? j - a : 80));
                    out += '\n';
                }
                k++;
            }
        } else if (len >= best_len) {
            best_at = i;
            best_len = len;
        }
        i = j;
    }
    if (mode != FA_FASTA_MULTI && best_at >= 0) {
        out += '>';
        out += seed_id;
        out += '\n';
        out.append(cns + best_at, (size_t)best_len);
        out += '\n';
    }
}

extern "C" long long fa_fasta_records(const char *seed_id, const char *cns, long long len, int mode,
                                      char *out, long long cap) {
    if (!seed_id || !cns || len < 0 || mode < 0 || mode > 2) return -1;
    std::string s;
    fa_fasta_append(s, seed_id, cns, len, mode);
    if (out && cap > 0) memcpy(out, s.data(), (size_t)((long long)s.size() < cap ? (long long)s.size() : cap));
    return (long long)s.size();
}
