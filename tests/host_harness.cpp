// tests/host_harness.cpp -- TEST HARNESS ONLY.  Compiles the *same* lane programs that libmwgpu.so runs on
// the GPU (metaworld_amd/csrc/*.hpp) for the host, with a plain loop over (block, thread) standing in for
// the wavefronts, so that the kernel logic can be checked against the oracle in a container without a GPU.
// Exports the ABI of include/mwgpu.h under the prefix mwh_.  Never loaded by the product path.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#define MW_LAMBDA
#include "../metaworld_amd/csrc/mw_common.hpp"

namespace {
struct Backend {
    static void init(int) {}
    static void* alloc(size_t bytes) { return std::malloc(bytes ? bytes : 16); }
    static void free(void* p) { std::free(p); }
    static void zero(void* p, size_t bytes) { std::memset(p, 0, bytes); }
    static void h2d(void* d, const void* s, size_t n) { std::memcpy(d, s, n); }
    static void d2h(void* d, const void* s, size_t n) { std::memcpy(d, s, n); }
    // scratchpad rows per lane: MW_LDS_ROWS (default 24, so that typical scenes exercise both the scratchpad rows
    // and the column-store fallback rows of the solver)
    static int lds_rows() { const char* v = std::getenv("MW_LDS_ROWS"); return v ? std::atoi(v) : 24; }
    // emulated sub-lanes per environment (MW_NSUB = a power of two up to 64; default 8 = the small-batch device configuration)
    static int nsub() { const char* v = std::getenv("MW_NSUB"); return v ? std::atoi(v) : 8; }
    template <class F>
    static void launch(int nblocks, F f) {   // one call per environment (locate() rejects threads >= lanes per workgroup): the sub-lanes are emulated inside (MW_SUBS)
        const int rows = lds_rows(), ns = nsub();
        const int words = rows * mw::SR_N * 2;   // SR_N doubles per row
#pragma omp parallel
        {
            std::vector<double> pad((size_t)rows * mw::SR_N + 1, std::nan(""));   // LDS is not zero-initialised either
#pragma omp for schedule(dynamic)
            for (int b = 0; b < nblocks; b++)
                for (int t = 0; t < 64; t++) f(b, t, mw::Scratchpad{pad.data(), words, ns});
        }
    }
    template <class F>
    static void launch_flat(int n, F f) { for (int i = 0; i < n; i++) f(i); }
    static int compute_units() { return 256; }
    static void sync() {}
    static std::chrono::steady_clock::time_point& t0() { static std::chrono::steady_clock::time_point t; return t; }
    static void timed_begin() { t0() = std::chrono::steady_clock::now(); }
    static float timed_end() { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0()).count(); }
};
}  // namespace

#include "../metaworld_amd/csrc/mw_runtime.hpp"

#include "../include/mwgpu.h"
#define MW_API(name) mwh_##name
#include "../metaworld_amd/csrc/mw_abi.inl"

#ifdef MW_PROFILE
extern "C" void mwh_profile(double* out, int reset) {
    for (int i = 0; i < 8; i++) { out[i] = mw::mw_prof()[i]; if (reset) mw::mw_prof()[i] = 0; }
}
#endif
#ifdef MW_PROFILE
extern "C" void mwh_counters(long* out, int reset) {
    for (int i = 0; i < 8; i++) { out[i] = mw::mw_cnt()[i]; if (reset) mw::mw_cnt()[i] = 0; }
}
#endif
#ifdef MW_PROFILE
extern "C" void mwh_hist(long* out, int reset) {
    for (int i = 0; i < 256; i++) { out[i] = mw::mw_hist()[i]; if (reset) mw::mw_hist()[i] = 0; }
}
#endif
