// cndmask.hip -- does v_cndmask_b32 really issue 5.6x slower than other VALU ops (the 107 G/s
// line of profiles/r02_ubench_issue_rates.txt), and does the form the kernels use -- the mask
// in an SGPR pair, VOP3 -- share that?  Same harness as issue_rates.hip: 8 wavefronts per SIMD,
// 8 independent registers, 256 instructions per loop trip.
//
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/cndmask.bin scripts/ubench/cndmask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITER 2000
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define REP8(s) s s s s s s s s
#define REP32(s) REP8(s) REP8(s) REP8(s) REP8(s)

#define KERNEL(name, pre, body)                                                             \
    __global__ __launch_bounds__(64, 8) void name(unsigned *out, unsigned long long mask) { \
        unsigned a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = a + 5,      \
                 g = a + 6, h = a + 7, k = blockIdx.x | 1;                                    \
        asm volatile(pre : : "s"(mask) : "vcc");                                              \
        for (int i = 0; i < ITER; i++)                                                        \
            asm volatile(REP32(body)                                                          \
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) \
                         : "v"(k), "s"(mask) : "vcc");                                        \
        out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + e + f + g + h;                   \
    }

#define OP8(fmt_a, fmt_b) \
    fmt_a "%0, %0, %8" fmt_b "\n\t" fmt_a "%1, %1, %8" fmt_b "\n\t" fmt_a "%2, %2, %8" fmt_b "\n\t" fmt_a "%3, %3, %8" fmt_b "\n\t" \
    fmt_a "%4, %4, %8" fmt_b "\n\t" fmt_a "%5, %5, %8" fmt_b "\n\t" fmt_a "%6, %6, %8" fmt_b "\n\t" fmt_a "%7, %7, %8" fmt_b "\n\t"

KERNEL(k_and, "", OP8("v_and_b32 ", ""))
KERNEL(k_cnd_vcc_unset, "", OP8("v_cndmask_b32 ", ", vcc"))
KERNEL(k_cnd_vcc_set, "s_mov_b64 vcc, %0", OP8("v_cndmask_b32 ", ", vcc"))
KERNEL(k_cnd_sgpr, "", OP8("v_cndmask_b32_e64 ", ", %9"))
// the alternative a select could be written as: the move under a narrowed exec mask
#define X8 \
    "s_mov_b64 exec, %9\n\tv_mov_b32 %0, %8\n\ts_mov_b64 exec, -1\n\t" "s_mov_b64 exec, %9\n\tv_mov_b32 %1, %8\n\ts_mov_b64 exec, -1\n\t" \
    "s_mov_b64 exec, %9\n\tv_mov_b32 %2, %8\n\ts_mov_b64 exec, -1\n\t" "s_mov_b64 exec, %9\n\tv_mov_b32 %3, %8\n\ts_mov_b64 exec, -1\n\t" \
    "s_mov_b64 exec, %9\n\tv_mov_b32 %4, %8\n\ts_mov_b64 exec, -1\n\t" "s_mov_b64 exec, %9\n\tv_mov_b32 %5, %8\n\ts_mov_b64 exec, -1\n\t" \
    "s_mov_b64 exec, %9\n\tv_mov_b32 %6, %8\n\ts_mov_b64 exec, -1\n\t" "s_mov_b64 exec, %9\n\tv_mov_b32 %7, %8\n\ts_mov_b64 exec, -1\n\t"
KERNEL(k_exec_mov, "", X8)

typedef void (*kern_t)(unsigned *, unsigned long long);

static void run(const char *name, kern_t k, double per_iter, int n_cu, unsigned *out, unsigned long long mask) {
    const int grid = n_cu * 4 * 8;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, out, mask);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, out, mask);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double rate = (double)grid * ITER * per_iter / (ms * 1e-3);
    printf("%-44s mask %016llx  %8.3f ms  %7.1f G selects (or ops)/s\n", name, mask, ms, rate / 1e9);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    unsigned *out;
    CHECK(hipMalloc((void **)&out, (size_t)n_cu * 32 * 64 * sizeof(unsigned)));
    for (unsigned long long m : {0x5555555555555555ull, 0xffffffffffffffffull, 0x00000000ffff0000ull}) {
        run("v_and_b32 (reference)", k_and, 256, n_cu, out, m);
        run("v_cndmask_b32 .., vcc (vcc never written)", k_cnd_vcc_unset, 256, n_cu, out, m);
        run("v_cndmask_b32 .., vcc (vcc set once)", k_cnd_vcc_set, 256, n_cu, out, m);
        run("v_cndmask_b32_e64 .., s[n:n+1]", k_cnd_sgpr, 256, n_cu, out, m);
        run("s_mov exec / v_mov / s_mov exec (per select)", k_exec_mov, 256, n_cu, out, m);
    }
    return 0;
}
