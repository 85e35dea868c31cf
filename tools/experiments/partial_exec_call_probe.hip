// Does a value that the CALLER keeps in a vector register for a lane that SITS OUT a call survive a non-inlined callee that (a) uses
// every VGPR and (b) spills SGPRs into VGPR lanes (v_writelane ignores EXEC)?  This is the pattern DESIGN.md 5 "register hazard" names
// as the common factor of the round-2 garbage (`want` of collision() after a partially-masked call of collide_pair); the product
// removes the pattern structurally, this probe asks whether the compiler's calling convention alone is enough.
//   caller: NLIVE per-lane values live across the call (more than the caller-saved registers can hold without touching callee-saved
//           ones), lanes with (lane % 3 == 0) make the call, the others sit out;
//   callee: noinline, ~250 live vector values + ~120 live wave-uniform scalars (forces SGPR spills to VGPR lanes, look for v_writelane
//           in the -save-temps assembly), returns a checksum.
// Every lane then recomputes its NLIVE values and compares: a mismatch in a lane that sat out = the hazard.
// build: hipcc --offload-arch=gfx950 -O3 [-mllvm -enable-ipra=0] -o tools/experiments/_build/partial_exec_call_probe[_noipra] tools/experiments/partial_exec_call_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define NLIVE 96
#define NVEC 200
#define NSCAL 120

__device__ __attribute__((noinline)) float callee(const float* __restrict__ vin, const int* __restrict__ sin, int lane, int rounds) {
    float v[NVEC];
    int s[NSCAL];
#pragma unroll
    for (int i = 0; i < NVEC; i++) v[i] = vin[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < NSCAL; i++) s[i] = __builtin_amdgcn_readfirstlane(sin[i]);          // wave-uniform: lives in SGPRs
    for (int r = 0; r < rounds; r++) {
#pragma unroll
        for (int i = 0; i < NVEC; i++) v[i] = v[i] * 1.0001f + (float)(s[i % NSCAL] + r);
#pragma unroll
        for (int i = 0; i < NSCAL; i++) s[i] = __builtin_amdgcn_readfirstlane(s[i] * 3 + s[(i + 7) % NSCAL] + r);
    }
    float acc = 0;
#pragma unroll
    for (int i = 0; i < NVEC; i++) acc += v[i];
#pragma unroll
    for (int i = 0; i < NSCAL; i++) acc += (float)(s[i] & 255);
    return acc;
}

__global__ void __launch_bounds__(64) probe(const float* vin, const int* sin, int rounds, int* bad, float* sink) {
    const int lane = threadIdx.x;
    float live[NLIVE];
#pragma unroll
    for (int i = 0; i < NLIVE; i++) live[i] = __sinf(0.37f * (float)(lane * 131 + i * 17 + blockIdx.x));
    float r = 0;
    if (lane % 3 == 0) r = callee(vin, sin, lane, rounds);          // a partially-masked call: two thirds of the wave sit out
    int wrong = 0;
#pragma unroll
    for (int i = 0; i < NLIVE; i++) wrong += live[i] != __sinf(0.37f * (float)(lane * 131 + i * 17 + blockIdx.x));
    if (wrong) atomicAdd(bad + (lane % 3 == 0 ? 0 : 1), wrong);
    sink[blockIdx.x * 64 + lane] = r;
}

int main() {
    float* vin; int* sin; int* bad; float* sink;
    hipMalloc(&vin, NVEC * 64 * 4); hipMalloc(&sin, NSCAL * 4); hipMalloc(&bad, 8); hipMalloc(&sink, 1024 * 64 * 4);
    float hv[NVEC * 64]; int hs[NSCAL];
    for (int i = 0; i < NVEC * 64; i++) hv[i] = 0.001f * (i % 977);
    for (int i = 0; i < NSCAL; i++) hs[i] = i * 2654435761u % 1000;
    hipMemcpy(vin, hv, sizeof(hv), hipMemcpyHostToDevice); hipMemcpy(sin, hs, sizeof(hs), hipMemcpyHostToDevice);
    hipMemset(bad, 0, 8);
    probe<<<1024, 64>>>(vin, sin, 5, bad, sink);
    int hb[2]; hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
    printf("partial-EXEC call probe: corrupted caller values in lanes that made the call: %d, in lanes that sat out: %d (of %d lanes x %d values)\n",
           hb[0], hb[1], 1024 * 64, NLIVE);
    return hb[0] != 0 || hb[1] != 0;
}
