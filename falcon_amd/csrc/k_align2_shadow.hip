// k_align2_shadow.hip -- TESTS ONLY: k_align2 with both renderings of its row loop run side by
// side -- a2_rows_c, the statement, and a2_rows_asm, the hand-scheduled stream (k_align2_rows.h)
// -- from the same state, stretch by stretch.  Whatever differs between them goes to a log in
// device memory (FALCON_AMD_A2_SHADOW=1: fa_launch_align2 launches this kernel, waits, and
// prints the log to stderr); the statement's result is what the kernel goes on with, so a run
// completes and its answers are those of the compiler's rendering.
#include <cstdio>
#include <cstdlib>
#include "fa_wave.h"

#define A2_SHADOW 1
#define A2_SHADOW_CAP 64
__device__ u32 g_a2_shadow_n;
__device__ u32 g_a2_shadow_adopt;
__device__ u32 g_a2_shadow[A2_SHADOW_CAP * 32];

#include "k_align2_core.h"

W_FN void a2_shadow_log2(u32 what, u32 it_in, u32 it_c, u32 it_a, const u64 *v, u32 xc, u32 xa, u32 flags, u32 cc,
                         u32 ca) {
    const u32 i = (u32)w_uni((int)atomicAdd(&g_a2_shadow_n, fa_lane() == 0 ? 1u : 0u));
    if (i >= A2_SHADOW_CAP) return;
    if (fa_lane() == 0) {
        u32 *e = g_a2_shadow + i * 32;
        e[0] = what; e[1] = it_in; e[2] = it_c; e[3] = it_a;
        for (int k = 0; k < 8; k++) { e[4 + 2 * k] = (u32)v[k]; e[5 + 2 * k] = (u32)(v[k] >> 32); }
        e[20] = xc; e[21] = xa; e[22] = flags; e[23] = cc; e[24] = ca; e[25] = blockIdx.x;
    }
}

W_FN bool a2_shadow_adopt() { return w_uniu(g_a2_shadow_adopt) != 0u; }

__global__ __launch_bounds__(64, 4) void k_align2_shadow(A2Args A) {
    a2_wave(A, (int)blockIdx.x);
}

void fa_launch_align2_shadow(const A2Args &A, int grid, size_t lds, hipStream_t s) {
    u32 zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_a2_shadow_n), &zero, sizeof(zero));
    const char *mode = getenv("FALCON_AMD_A2_SHADOW");
    const u32 adopt = (mode && atoi(mode) == 2) ? 1u : 0u;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_a2_shadow_adopt), &adopt, sizeof(adopt));
    hipLaunchKernelGGL(k_align2_shadow, dim3(grid), dim3(64), lds, s, A);
    (void)hipStreamSynchronize(s);
    u32 n = 0;
    static u32 log[A2_SHADOW_CAP * 32];
    (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_a2_shadow_n), sizeof(n));
    (void)hipMemcpyFromSymbol(log, HIP_SYMBOL(g_a2_shadow), sizeof(log));
    fprintf(stderr, "a2_shadow: %u stretches differ\n", n);
    for (u32 i = 0; i < n && i < A2_SHADOW_CAP && i < 16; i++) {
        const u32 *e = log + i * 32;
        fprintf(stderr, "  [%u] what=%04x pair=%u head=%u split=%u wave=%u it_in=%u it_c=%u it_a=%u lane=%u vx c=%d a=%d cells0 c=%u a=%u\n",
                i, e[0], e[22] & 1u, (e[22] >> 1) & 1u, (e[22] >> 16) & 0xffu, e[25], e[1], e[2], e[3],
                (e[22] >> 8) & 0xffu, (int)e[20], (int)e[21], e[23], e[24]);
        fprintf(stderr, "      act c=%08x%08x a=%08x%08x  in c=%08x%08x a=%08x%08x\n", e[5], e[4], e[7], e[6], e[9], e[8],
                e[11], e[10]);
        fprintf(stderr, "      vx differs in %08x%08x  act at entry %08x%08x  fin c=%08x%08x a=%08x%08x\n", e[13], e[12],
                e[15], e[14], e[17], e[16], e[19], e[18]);
}
}
