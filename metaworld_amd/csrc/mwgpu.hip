// mwgpu.hip -- libmwgpu.so: HIP (gfx950) backend of the batched Meta-World runtime + its C ABI (include/mwgpu.h).
//
// Kernel shape: one wavefront (64 threads) per workgroup carrying lpb <= 64 environments of one model group (the other
// threads are sub-lanes of those environments, mw_common.hpp), grid = sum over groups of ceil(n_env / lpb); dynamic LDS =
// the lanes' scratchpad (solver row scalars).  All per-env data is in the chunked column store described in
// mw_common.hpp, so each per-env load/store of a wave is one contiguous request.
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <stdexcept>
#include <string>

#define MW_LAMBDA __host__ __device__

namespace {
inline void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
}  // namespace
#include "mw_common.hpp"
namespace {
// one wavefront per workgroup; the dynamic LDS allocation is the lanes' scratchpad (solver row scalars, mw_phys.hpp)
template <class F>
__global__ void __launch_bounds__(64) k_lanes(F f, int block_words) {
    extern __shared__ float mw_scratchpad[];
    f((int)blockIdx.x, (int)threadIdx.x, mw::Scratchpad{(MW_LDS void*)mw_scratchpad, block_words, 0});
}

// one thread per environment, no scratchpad: the small per-env kernels around the step (scripted policies, accounting)
template <class F>
__global__ void __launch_bounds__(256) k_flat(F f, int n) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i < n) f(i);
}

struct Backend {
    static hipStream_t& stream() { static hipStream_t s = nullptr; return s; }
    static hipEvent_t* events() { static hipEvent_t ev[2] = {nullptr, nullptr}; return ev; }
    static void init(int device) {
        int n = 0;
        hip_check(hipGetDeviceCount(&n), "hipGetDeviceCount");
        if (n <= 0) throw std::runtime_error("libmwgpu: no HIP device visible (this library has no CPU fallback)");
        hip_check(hipSetDevice(device), "hipSetDevice");
        hip_check(hipDeviceGetAttribute(&max_lds(), hipDeviceAttributeMaxSharedMemoryPerBlock, device), "hipDeviceGetAttribute");
        hip_check(hipDeviceGetAttribute(&num_cu(), hipDeviceAttributeMultiprocessorCount, device), "hipDeviceGetAttribute");
        if (!stream()) {
            hip_check(hipStreamCreateWithFlags(&stream(), hipStreamNonBlocking), "hipStreamCreate");
            hip_check(hipEventCreate(&events()[0]), "hipEventCreate");
            hip_check(hipEventCreate(&events()[1]), "hipEventCreate");
        }
    }
    static void* alloc(size_t bytes) { void* p = nullptr; hip_check(hipMalloc(&p, bytes ? bytes : 16), "hipMalloc"); return p; }
    static void free(void* p) { if (p) (void)hipFree(p); }
    static void zero(void* p, size_t bytes) { hip_check(hipMemsetAsync(p, 0, bytes, stream()), "hipMemset"); }
    static void h2d(void* dst, const void* src, size_t bytes) {
        hip_check(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream()), "hipMemcpy H2D");
        hip_check(hipStreamSynchronize(stream()), "sync");
    }
    static void d2h(void* dst, const void* src, size_t bytes) {
        hip_check(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream()), "hipMemcpy D2H");
        hip_check(hipStreamSynchronize(stream()), "sync");
    }
    // LDS per workgroup: everything a CU has when the grid leaves one wave per CU, an equal share when several
    // workgroups must share a CU (the 512-VGPR lane programs allow at most one wave per SIMD, i.e. 4 per CU).
    // MW_LDS_BYTES overrides (experiments / tests of the column-store fallback rows).
    static int& max_lds() { static int v = 65536; return v; }
    static int& num_cu() { static int v = 256; return v; }
    static int compute_units() { return num_cu(); }
    static int lds_bytes(int nblocks) {
        static const char* ov = getenv("MW_LDS_BYTES");
        if (ov) return atoi(ov);
        static const char* wv = getenv("MW_WAVES_PER_CU");   // experiments: 8 = let two waves share a SIMD (256 VGPRs each)
        const int max_per_cu = wv ? atoi(wv) : 4;
        int per_cu = (nblocks + num_cu() - 1) / num_cu();
        per_cu = per_cu < 1 ? 1 : (per_cu > max_per_cu ? max_per_cu : per_cu);
        return (max_lds() / per_cu) & ~1023;
    }
    template <class F>
    static void launch(int nblocks, F f) {
        static int configured = 0;
        const int bytes = lds_bytes(nblocks);
        if (bytes > configured) {
            hip_check(hipFuncSetAttribute((const void*)k_lanes<F>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), "hipFuncSetAttribute(LDS)");
            configured = bytes;
        }
        hipLaunchKernelGGL(k_lanes<F>, dim3(nblocks), dim3(64), bytes, stream(), f, bytes / 4);
        hip_check(hipGetLastError(), "kernel launch");
    }
    template <class F>
    static void launch_flat(int n, F f) {
        hipLaunchKernelGGL(k_flat<F>, dim3((n + 255) / 256), dim3(256), 0, stream(), f, n);
        hip_check(hipGetLastError(), "kernel launch");
    }
    static void sync() { hip_check(hipStreamSynchronize(stream()), "hipStreamSynchronize"); }
    static void timed_begin() { hip_check(hipEventRecord(events()[0], stream()), "hipEventRecord"); }
    static float timed_end() {
        hip_check(hipEventRecord(events()[1], stream()), "hipEventRecord");
        hip_check(hipEventSynchronize(events()[1]), "hipEventSynchronize");
        float ms = 0;
        hip_check(hipEventElapsedTime(&ms, events()[0], events()[1]), "hipEventElapsedTime");
        return ms;
    }
};
}  // namespace

#include "mw_runtime.hpp"

#include "../../include/mwgpu.h"
#define MW_API(name) mw_##name
#include "mw_abi.inl"
