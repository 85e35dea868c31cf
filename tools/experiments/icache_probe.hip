// icache_probe.hip -- how much does STRAIGHT-LINE code cost on gfx950 when one wave per SIMD streams through more code than the
// instruction cache holds (64 KB shared by two CUs) and the waves of a CU are at different places of it?  The step kernel's lane
// programs are 1 MB of mostly unrolled code (collide_pair 184 KB, solve 172 KB, lane_step 182 KB) executed by ONE wave per SIMD.
//
// Kernel: NSEG segments of SEG_FMAS dependent-chain FMAs each (8 independent accumulators, distinct literal per instruction so
// nothing is merged), visited round-robin starting at segment (wave % NSEG).  Same FMA count for every configuration:
//   footprint = NSEG x SEG_FMAS x 8 bytes.   build: hipcc --offload-arch=gfx950 -O3 -o icache_probe icache_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int K> struct Seg {
    template <int I> static __device__ __forceinline__ void step(float (&a)[8], float x) {
        if constexpr (I > 0) {
            step<I - 1>(a, x);
            // v_fma_f32 with a literal: 8 bytes (12 with the 32-bit literal) per instruction
            a[I & 7] = __builtin_fmaf(a[I & 7], x, (float)(K * 4096 + I) * 1.0009765625f);
        }
    }
};
template <int NSEG, int SEG_FMAS, int K>
__device__ __forceinline__ void run_seg(int seg, float (&a)[8], float x) {
    if constexpr (K < NSEG) {
        if (seg == K) { Seg<K>::template step<SEG_FMAS>(a, x); return; }
        run_seg<NSEG, SEG_FMAS, K + 1>(seg, a, x);
    }
}
template <int NSEG, int SEG_FMAS>
__global__ void __launch_bounds__(64) probe(float* out, float x, int visits, int phase) {
    float a[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    int seg = phase ? (int)(blockIdx.x % NSEG) : 0;
    for (int v = 0; v < visits; v++) {
        run_seg<NSEG, SEG_FMAS, 0>(seg, a, x);
        seg = seg + 1 == NSEG ? 0 : seg + 1;
    }
    float s = 0;
    for (int k = 0; k < 8; k++) s += a[k];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int NSEG, int SEG_FMAS>
static void run(const char* what, float* d, int blocks, long total_fmas, int phase) {
    const int visits = (int)(total_fmas / SEG_FMAS);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NSEG, SEG_FMAS><<<blocks, 64>>>(d, 0.999f, visits, phase);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NSEG, SEG_FMAS><<<blocks, 64>>>(d, 0.999f, visits, phase);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)visits * SEG_FMAS);
    printf("%-44s footprint %7.1f KB  blocks %5d  phase %d  %8.3f ms  %6.2f cycles / FMA wave-instruction\n", what,
           NSEG * SEG_FMAS * 8.0 / 1024, blocks, phase, ms, cyc);
}
int main(int argc, char** argv) {
    float* d;
    hipMalloc(&d, 4096 * 64 * sizeof(float));
    const long total = 1L << 20;          // FMAs per wave
    for (int blocks : {1024, 2048}) {
        run<1, 512>("loop, 4 KB body", d, blocks, total, 0);
        run<8, 512>("8 segments x 4 KB = 32 KB", d, blocks, total, 1);
        run<16, 512>("16 segments x 4 KB = 64 KB", d, blocks, total, 1);
        run<32, 512>("32 segments x 4 KB = 128 KB", d, blocks, total, 1);
        run<64, 512>("64 segments x 4 KB = 256 KB", d, blocks, total, 1);
        run<64, 512>("64 segments x 4 KB = 256 KB, waves in step", d, blocks, total, 0);
        run<128, 512>("128 segments x 4 KB = 512 KB", d, blocks, total, 1);
    }
    return 0;
}
