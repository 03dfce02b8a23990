// scripts/ubench/coresident.hip -- does a kernel of few wavefronts get onto the chip while a persistent grid holds most
// of it?  A: `na` workgroups of one wavefront (64 VGPRs forced, `lds_a` bytes of LDS), each busy for ~30 ms; B, on another
// stream 2 ms later: 3072 workgroups of one wavefront (`vgpr_b` registers, `lds_b` bytes of LDS) of ~0.3 ms each.  Prints
// when B ended relative to A's start and end.  (k_align2 beside k_backtrace of the batch before: engine.hip, the gate.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int NV>
__global__ void __launch_bounds__(64) busy(unsigned long long ticks, int *sink) {
    extern __shared__ int lds[];
    int v[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = threadIdx.x * (i + 1);
    const unsigned long long t0 = wall_clock64();
    int acc = 0;
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < NV; i++) { v[i] = v[i] * 3 + acc; acc += v[i]; }
        lds[threadIdx.x] = acc;
    }
    if (acc == 12345) *sink = acc + lds[0];
}
int main(int argc, char **argv) {
    const int na = argc > 1 ? atoi(argv[1]) : 7168, lds_a = argc > 2 ? atoi(argv[2]) : 4096, lds_b = argc > 3 ? atoi(argv[3]) : 12800;
    int *sink;
    hipMalloc(&sink, 4);
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    hipEvent_t a0, a1, b0, b1;
    hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b0); hipEventCreate(&b1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a0, sa);
        hipLaunchKernelGGL(busy<28>, dim3(na), dim3(64), lds_a, sa, 3000000ull, sink);   // ~30 ms at 100 MHz
        hipEventRecord(a1, sa);
        hipLaunchKernelGGL(busy<4>, dim3(1), dim3(64), 0, sb, 200000ull, sink);          // B comes 2 ms later
        hipEventRecord(b0, sb);
        hipLaunchKernelGGL(busy<28>, dim3(3072), dim3(64), lds_b, sb, 30000ull, sink);
        hipEventRecord(b1, sb);
        hipDeviceSynchronize();
        float ta, tb0, tb1;
        hipEventElapsedTime(&ta, a0, a1); hipEventElapsedTime(&tb0, a0, b0); hipEventElapsedTime(&tb1, a0, b1);
        if (rep) printf("A: %d wavefronts, %d B of LDS each: %.1f ms.  B (3072 x 0.3 ms, %d B of LDS): from %.1f to %.1f ms after A began\n",
                        na, lds_a, ta, lds_b, tb0, tb1);
    }
    return 0;
}
