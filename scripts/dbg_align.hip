// developer aid: exercises fa_align_pairs and the DPP reduction directly
#include "../include/falcon_amd.h"
#include "../falcon_amd/csrc/fa_device.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

__global__ void k_test_max(const int* in, int* out) {
    int v = in[threadIdx.x];
    int m = fa_wave_max(v);
    int mn = fa_wave_min(v);
    if (threadIdx.x == 0) { out[0] = m; out[1] = mn; }
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    // DPP check
    int h[64], *d_in, *d_out, ho[2];
    for (int t = 0; t < 5; t++) {
        int mx = -1000000, mn = 1000000;
        for (int i = 0; i < 64; i++) { h[i] = (rand() % 2000) - 1000; if (t == 0) h[i] = i; if (t==1) h[i] = 63 - i; mx = std::max(mx, h[i]); mn = std::min(mn, h[i]); }
        hipMalloc(&d_in, 256); hipMalloc(&d_out, 8);
        hipMemcpy(d_in, h, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_test_max, dim3(1), dim3(64), 0, 0, d_in, d_out);
        hipMemcpy(ho, d_out, 8, hipMemcpyDeviceToHost);
        printf("dpp max %d (exp %d) min %d (exp %d)\n", ho[0], mx, ho[1], mn);
    }
    fa_ctx* c = fa_create(0);
    if (!c) { printf("create failed %s\n", fa_last_error()); return 1; }
    srand(5);
    std::string s;
    for (int i = 0; i < 700; i++) s.push_back("ACGT"[rand() & 3]);
    std::string q300 = s.substr(0, 300);
    for (int rep = 0; rep < 6; rep++) {
        const char* q = (rep & 1) ? q300.c_str() : s.c_str();
        int ql = (rep & 1) ? 300 : 700;
        const char* t = s.c_str();
        int tl = 700;
        alignment* a = nullptr;
        int rc = fa_align_pairs(c, 1, &q, &ql, &t, &tl, 150, 1, &a);
        if (rc) { printf("rc %d %s\n", rc, fa_last_error()); continue; }
        printf("rep %d ql %d: size %d dist %d qe %d te %d strlen %zu %zu qmatch %d tmatch %d\n", rep, ql, a->aln_str_size, a->dist,
               a->aln_q_e, a->aln_t_e, strlen(a->q_aln_str), strlen(a->t_aln_str),
               (int)(strncmp(a->q_aln_str, q, ql) == 0), (int)(strncmp(a->t_aln_str, t, ql) == 0));
        free_alignment(a);
    }
    // full pipeline on a noisy two-sequence pile
    {
        std::string seed = s, qn;
        for (size_t i = 0; i < s.size(); i++) {
            int r = rand() % 100;
            if (r < 3) continue;                    // deletion
            if (r < 5) { qn.push_back("ACGT"[rand() & 3]); continue; }  // substitution
            qn.push_back(s[i]);
            if (r < 12) qn.push_back("ACGT"[rand() & 3]);  // insertion
        }
        const char* seqs[2] = {seed.c_str(), qn.c_str()};
        int n = 2;
        fa_batch* b = fa_batch_create(c, 1, &n, seqs, nullptr);
        if (!b) { printf("batch create failed %s\n", fa_last_error()); return 1; }
        int rc = fa_batch_run(b, 0, 8, 0.70);
        printf("run rc %d %s\n", rc, rc ? fa_last_error() : "");
        if (!rc) {
            int s1,e1,s2,e2,ok,nh; long long sc;
            fa_batch_range(b, 1, &s1,&e1,&s2,&e2,&sc,&ok,&nh);
            printf("range %d %d %d %d score %lld ok %d nhit %d\n", s1,e1,s2,e2,sc,ok,nh);
            int dist,qe,te,size,acc; long long cells;
            fa_batch_alignment(b, 1, &dist,&qe,&te,&size,&acc,&cells);
            printf("aln dist %d qe %d te %d size %d acc %d cells %lld\n", dist,qe,te,size,acc,cells);
            fa_batch_fetch(b, 1);
            const char* cs; int len; const int* eq;
            fa_batch_result(b, 0, &cs, &len, &eq);
            printf("cns len %d %.60s\n", len, cs);
        }
        fa_batch_free(b);
    }
    return 0;
}
