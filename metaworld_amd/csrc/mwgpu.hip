// mwgpu.hip -- libmwgpu.so: HIP (gfx950) backend of the batched Meta-World runtime + its C ABI (include/mwgpu.h).
//
// Kernel shape: one wavefront (64 threads) per workgroup carrying lpb <= 64 environments of one model group (the other
// threads are sub-lanes of those environments, mw_common.hpp), grid = sum over groups of ceil(n_env / lpb); dynamic LDS =
// the lanes' scratchpad (solver row scalars).  All per-env data is in the chunked column store described in
// mw_common.hpp, so each per-env load/store of a wave is one contiguous request.
#include <hip/hip_runtime.h>
#include <mutex>

#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <rccl/rccl.h>   // types only: the entry points are resolved with dlsym (struct Rccl)

#include <stdexcept>
#include <string>
#include <vector>

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
__global__ void __launch_bounds__(64) k_lanes(F f, int block_words, int chain) {
    // (MW_SYNC, mw_common.hpp, is a wavefront-scope fence + wave barrier: correct only while a workgroup IS one wave64, and the lane
    //  roles of newton_direction_wave, mw_phys.hpp, assume 64 lanes: Backend::init refuses a device whose wavefront is not 64 wide)
    extern __shared__ float mw_scratchpad[];
    f((int)blockIdx.x, (int)threadIdx.x, mw::Scratchpad{(MW_LDS void*)mw_scratchpad, block_words, 0, 0, chain});
}

// flat wave kernels of the split collision (mw_split.inl): f(wave, lane, number of waves, LDS); no sub-lanes, no column-store chunking
template <class F>
__global__ void __launch_bounds__(64) k_waves(F f) {
    extern __shared__ float mw_wave_lds[];
    f((int)blockIdx.x, (int)threadIdx.x, (int)gridDim.x, mw::WaveLds{(MW_LDS void*)mw_wave_lds});
}

// one thread per environment, no scratchpad: the small per-env kernels around the step (scripted policies, accounting)
template <class F>
__global__ void __launch_bounds__(256) k_flat(F f, int n) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i < n) f(i);
}

// RCCL entry points, resolved at run time (mw_comm_init): no link-time dependency, and when torch is in the process its
// already loaded librccl.so.1 is the one that answers (same SONAME), so the library and torch.distributed share one RCCL.
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    static Rccl& get() {
        static Rccl r;
        if (!r.handle) {
            for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"})
                if ((r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
            if (!r.handle) throw std::runtime_error(std::string("libmwgpu: cannot load RCCL: ") + dlerror());
            auto sym = [&](const char* n) { void* p = dlsym(r.handle, n); if (!p) throw std::runtime_error(std::string("RCCL symbol missing: ") + n); return p; };
            r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
            r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
            r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
            r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
            r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
            r.CommUserRank = (decltype(r.CommUserRank))sym("ncclCommUserRank");
            r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        }
        return r;
    }
    void check(ncclResult_t e, const char* what) const {
        if (e != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + (GetErrorString ? GetErrorString(e) : "RCCL error"));
    }
};

// One stream / event pair / side stream per HIP device, created on first use; every ABI entry selects its context's
// device first (Backend::use), so contexts on different devices can live in one process.
struct Backend {
    struct Dev { std::mutex ev_mu;          /* the record + wait pairs on the shared events below are atomic per device (two contexts driven from two host threads) */
                 hipEvent_t cin = nullptr, cout = nullptr, done_ev = nullptr; std::vector<hipEvent_t> marks; int nmarks = 0; hipStream_t stream = nullptr, side = nullptr; hipEvent_t ev[2] = {nullptr, nullptr}; hipEvent_t xev[2] = {nullptr, nullptr}, gev[2] = {nullptr, nullptr}; bool gev_used[2] = {false, false}; int max_lds = 65536, num_cu = 256; bool ready = false; };
    static constexpr int MAX_DEV = 64;
    static Dev& dev() { static Dev d[MAX_DEV]; return d[cur()]; }
    static int& cur() { static thread_local int c = 0; return c; }
    static hipStream_t& stream() { return dev().stream; }
    static hipEvent_t* events() { return dev().ev; }
    static void use(int device) {
        if (device < 0 || device >= MAX_DEV) throw std::runtime_error("libmwgpu: bad device id");
        hip_check(hipSetDevice(device), "hipSetDevice");
        cur() = device;
    }
    static void init(int device) {
        int n = 0;
        hip_check(hipGetDeviceCount(&n), "hipGetDeviceCount");
        if (n <= 0) throw std::runtime_error("libmwgpu: no HIP device visible (this library has no CPU fallback)");
        if (device >= n) throw std::runtime_error("libmwgpu: device_id out of range");
        use(device);
        Dev& d = dev();
        if (d.ready) return;
        hip_check(hipDeviceGetAttribute(&d.max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device), "hipDeviceGetAttribute");
        hip_check(hipDeviceGetAttribute(&d.num_cu, hipDeviceAttributeMultiprocessorCount, device), "hipDeviceGetAttribute");
        int wave = 0;
        hip_check(hipDeviceGetAttribute(&wave, hipDeviceAttributeWarpSize, device), "hipDeviceGetAttribute");
        if (wave != 64) throw std::runtime_error("libmwgpu: the lane programs assume 64-wide wavefronts (one wave per workgroup, wave-local exchanges); this device reports " + std::to_string(wave));
        hip_check(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking), "hipStreamCreate");
        hip_check(hipStreamCreateWithFlags(&d.side, hipStreamNonBlocking), "hipStreamCreate");
        hip_check(hipEventCreateWithFlags(&d.cin, hipEventDisableTiming), "hipEventCreate");
        hip_check(hipEventCreateWithFlags(&d.cout, hipEventDisableTiming), "hipEventCreate");
        hip_check(hipEventCreateWithFlags(&d.done_ev, hipEventDisableTiming), "hipEventCreate");
        for (int k = 0; k < 2; k++) {
            hip_check(hipEventCreate(&d.ev[k]), "hipEventCreate");
            hip_check(hipEventCreateWithFlags(&d.xev[k], hipEventDisableTiming), "hipEventCreate");
            hip_check(hipEventCreateWithFlags(&d.gev[k], hipEventDisableTiming), "hipEventCreate");
        }
        d.ready = true;
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
    // queued copies (no sync): a batch of them is drained by one sync()
    static void d2h_async(void* dst, const void* src, size_t bytes) { hip_check(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream()), "hipMemcpy D2H"); }
    static void h2d_async(void* dst, const void* src, size_t bytes) { hip_check(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream()), "hipMemcpy H2D"); }
    static void* alloc_host(size_t bytes) { void* p = nullptr; hip_check(hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault), "hipHostMalloc"); return p; }
    static void free_host(void* p) { if (p) (void)hipHostFree(p); }
    // LDS per workgroup: everything a CU has when the grid leaves one wave per CU, an equal share when several
    // workgroups must share a CU (the 512-VGPR lane programs allow at most one wave per SIMD, i.e. 4 per CU).
    // MW_LDS_BYTES overrides (experiments / tests of the column-store fallback rows).
    static int compute_units() { return dev().num_cu; }
    static int lds_bytes(int nblocks) {
        static const char* ov = getenv("MW_LDS_BYTES");
        if (ov) return atoi(ov);
        static const char* wv = getenv("MW_WAVES_PER_CU");   // experiments: 8 = let two waves share a SIMD (256 VGPRs each)
        const int max_per_cu = wv ? atoi(wv) : 4;
        int per_cu = (nblocks + compute_units() - 1) / compute_units();
        per_cu = per_cu < 1 ? 1 : (per_cu > max_per_cu ? max_per_cu : per_cu);
        return (dev().max_lds / per_cu) & ~1023;
    }
    template <class F>
    static void launch(int nblocks, F f) {
        static int configured[MAX_DEV] = {0};
        const int bytes = lds_bytes(nblocks);
        if (bytes > configured[cur()]) {
            hip_check(hipFuncSetAttribute((const void*)k_lanes<F>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), "hipFuncSetAttribute(LDS)");
            configured[cur()] = bytes;
        }
        static const int chain = getenv("MW_CHAIN_LDS") ? atoi(getenv("MW_CHAIN_LDS")) : 2;   // experiments: cap on Env::chain_lds (0 = body-level chains through the column store, 1 = only the slots in front of the rows + composite inertias)
        hipLaunchKernelGGL(k_lanes<F>, dim3(nblocks), dim3(64), bytes, stream(), f, bytes / 4, chain);
        hip_check(hipGetLastError(), "kernel launch");
    }
    // persistent waves of the narrow-phase kernel: two per SIMD (its callee collide_pair uses 256 VGPRs)
    static int narrow_waves() { static const char* ov = getenv("MW_NARROW_WAVES"); return ov ? atoi(ov) : 8 * compute_units(); }
    template <class F>
    static void launch_waves(int nwaves, int lds_bytes, F f) {
        if (lds_bytes > 65536) hip_check(hipFuncSetAttribute((const void*)k_waves<F>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes), "hipFuncSetAttribute(LDS)");
        hipLaunchKernelGGL(k_waves<F>, dim3(nwaves), dim3(64), lds_bytes, stream(), f);
        hip_check(hipGetLastError(), "kernel launch");
    }
    template <class F>
    static void launch_flat(int n, F f) {
        hipLaunchKernelGGL(k_flat<F>, dim3((n + 255) / 256), dim3(256), 0, stream(), f, n);
        hip_check(hipGetLastError(), "kernel launch");
    }
    static void sync() { hip_check(hipStreamSynchronize(stream()), "hipStreamSynchronize"); }
    // ordering against the CALLER's stream (mw_step_device_on): our stream first waits for everything the caller has queued (its
    // writes to the action tensor), and afterwards the caller's stream waits for what we queued -- no host synchronisation at all
    static void wait_for_caller(void* s) {
        Dev& d = dev();
        std::lock_guard<std::mutex> g(d.ev_mu);
        hip_check(hipEventRecord(d.cin, (hipStream_t)s), "hipEventRecord(caller stream)");
        hip_check(hipStreamWaitEvent(d.stream, d.cin, 0), "hipStreamWaitEvent");
    }
    static void caller_waits_for_us(void* s) {
        Dev& d = dev();
        std::lock_guard<std::mutex> g(d.ev_mu);
        hip_check(hipEventRecord(d.cout, d.stream), "hipEventRecord");
        hip_check(hipStreamWaitEvent((hipStream_t)s, d.cout, 0), "hipStreamWaitEvent(caller stream)");
    }
    static void record_done() { hip_check(hipEventRecord(dev().done_ev, dev().stream), "hipEventRecord"); }
    static void wait_done() { hip_check(hipEventSynchronize(dev().done_ev), "hipEventSynchronize"); }
    static void timed_begin() { dev().nmarks = 0; hip_check(hipEventRecord(events()[0], stream()), "hipEventRecord"); }
    // per-launch times of a timed region (mw_launch_times): one event after every step of the resident loop, read after timed_end
    static constexpr int MAX_MARKS = 8192;
    static void timed_mark() {
        Dev& d = dev();
        if (d.nmarks >= MAX_MARKS) return;
        if ((int)d.marks.size() <= d.nmarks) { hipEvent_t e; hip_check(hipEventCreate(&e), "hipEventCreate"); d.marks.push_back(e); }
        hip_check(hipEventRecord(d.marks[d.nmarks++], d.stream), "hipEventRecord");
    }
    static int launch_times(float* out, int cap) {          // ms between consecutive marks of the last timed region (the first from timed_begin)
        Dev& d = dev();
        int n = 0;
        for (; n < d.nmarks && n < cap; n++)
            hip_check(hipEventElapsedTime(out + n, n == 0 ? d.ev[0] : d.marks[n - 1], d.marks[n]), "hipEventElapsedTime");
        return n;
    }
    static float timed_end() {
        hip_check(hipEventRecord(events()[1], stream()), "hipEventRecord");
        hip_check(hipEventSynchronize(events()[1]), "hipEventSynchronize");
        float ms = 0;
        hip_check(hipEventElapsedTime(&ms, events()[0], events()[1]), "hipEventElapsedTime");
        return ms;
    }
    // ---- cross-rank exchange (RCCL over xGMI): one communicator per context, collectives on the device's SIDE stream so
    //      that the gather of step k overlaps the kernel of step k+1 ----
    struct Comm { ncclComm_t comm = nullptr; int rank = 0, world = 1, count_seen = 0, rank_seen = -1; };
    static void comm_unique_id(void* out128) {
        ncclUniqueId id;
        Rccl& r = Rccl::get();
        r.check(r.GetUniqueId(&id), "ncclGetUniqueId");
        static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
        memcpy(out128, &id, sizeof(id));
    }
    static Comm* comm_init(const void* id128, int rank, int world) {
        Rccl& r = Rccl::get();
        ncclUniqueId id;
        memcpy(&id, id128, sizeof(id));
        Comm* c = new Comm();
        c->rank = rank; c->world = world;
        r.check(r.CommInitRank(&c->comm, world, id, rank), "ncclCommInitRank");
        // what RCCL itself reports for the new communicator (mw_comm_info): proof that it spans `world` ranks
        r.check(r.CommCount(c->comm, &c->count_seen), "ncclCommCount");
        r.check(r.CommUserRank(c->comm, &c->rank_seen), "ncclCommUserRank");
        if (getenv("MW_VERBOSE")) fprintf(stderr, "[mwgpu] RCCL communicator: ncclCommCount = %d, ncclCommUserRank = %d (asked for world %d, rank %d)\n", c->count_seen, c->rank_seen, world, rank);
        return c;
    }
    // [0] ranks in the communicator as the collective library reports them (ncclCommCount), [1] this rank (ncclCommUserRank),
    // [2] 1 = a real RCCL communicator, 0 = none (world size 1: the gather is a device copy), [3] HIP device of the context
    static void comm_info(const Comm* c, int* out) {
        out[0] = c ? c->count_seen : 1; out[1] = c ? c->rank_seen : 0; out[2] = c ? 1 : 0; out[3] = cur();
    }
    static void comm_free(Comm* c) { if (c) { if (c->comm) (void)Rccl::get().CommDestroy(c->comm); delete c; } }
    // side stream waits for everything queued on the main stream so far, then all-gathers `bytes` per rank; the "gather done"
    // event of the record slot is what the main stream waits for before a later kernel rewrites that slot (wait_gather_done)
    static void allgather_side(Comm* c, const void* send, void* recv, size_t bytes, int slot) {
        if (!c) { allgather_side_local(send, recv, bytes, slot); return; }
        Dev& d = dev();
        hip_check(hipEventRecord(d.xev[slot & 1], d.stream), "hipEventRecord");
        hip_check(hipStreamWaitEvent(d.side, d.xev[slot & 1], 0), "hipStreamWaitEvent");
        Rccl& r = Rccl::get();
        r.check(r.AllGather(send, recv, bytes, ncclInt8, c->comm, d.side), "ncclAllGather");
        hip_check(hipEventRecord(d.gev[slot & 1], d.side), "hipEventRecord");
        d.gev_used[slot & 1] = true;
    }
    // back-edge of the two-slot record ring: the main stream may not run the kernel that REWRITES record slot `slot` before the
    // side stream has finished the gather that READS it (a collective slower than one kernel: first-call RCCL setup, a slow rank)
    static void wait_gather_done(int slot) {
        Dev& d = dev();
        if (d.gev_used[slot & 1]) hip_check(hipStreamWaitEvent(d.stream, d.gev[slot & 1], 0), "hipStreamWaitEvent");
    }
    static void sync_side() { hip_check(hipStreamSynchronize(dev().side), "hipStreamSynchronize(side)"); }
    static void allgather_side_local(const void* send, void* recv, size_t bytes, int slot) {   // world size 1: a device copy
        Dev& d = dev();
        hip_check(hipEventRecord(d.xev[slot & 1], d.stream), "hipEventRecord");
        hip_check(hipStreamWaitEvent(d.side, d.xev[slot & 1], 0), "hipStreamWaitEvent");
        hip_check(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, d.side), "hipMemcpy D2D");
        hip_check(hipEventRecord(d.gev[slot & 1], d.side), "hipEventRecord");
        d.gev_used[slot & 1] = true;
    }
    static void copy_side(void* dst, const void* src, size_t bytes, bool dst_on_device) {
        hip_check(hipMemcpyAsync(dst, src, bytes, dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, dev().side), "hipMemcpy (side)");
    }
};
}  // namespace

#include "mw_runtime.hpp"

#include "../../include/mwgpu.h"
#define MW_API(name) mw_##name
#include "mw_abi.inl"
