// tests/host_harness.cpp -- TEST HARNESS ONLY.  Compiles the *same* lane programs that libmwgpu.so runs on
// the GPU (metaworld_amd/csrc/*.hpp) for the host, with a plain loop over (block, thread) standing in for
// the wavefronts, so that the kernel logic can be checked against the oracle in a container without a GPU.
// Exports the ABI of include/mwgpu.h under the prefix mwh_.  Never loaded by the product path.
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#define MW_LAMBDA
#include "../include/mwgpu.h"
#include "../metaworld_amd/csrc/mw_common.hpp"

namespace {
struct Backend {
    static void init(int) {}
    static void use(int) {}
    static void d2h_async(void* d, const void* s, size_t n) { std::memcpy(d, s, n); }
    static void h2d_async(void* d, const void* s, size_t n) { std::memcpy(d, s, n); }
    // Cross-rank exchange of the harness: the ranks are processes on one host, the "collective" is a POSIX shared-memory
    // segment named by the 128-byte id (tests/test_multirank_gloo.py hands the id from rank 0 to the others over gloo,
    // exactly as the RCCL unique id travels in the product path).
    struct Shm { std::atomic<int> count, gen; };
    struct Comm { int rank = 0, world = 1; char* base = nullptr; size_t cap = 1 << 20; std::string name; };
    static void comm_unique_id(void* out128) {
        static int counter = 0;
        std::memset(out128, 0, 128);
        std::snprintf((char*)out128, 128, "/mwh_%d_%d", (int)getpid(), counter++);
    }
    static Comm* comm_init(const void* id128, int rank, int world) {
        Comm* c = new Comm();
        c->rank = rank; c->world = world; c->name = (const char*)id128;
        const size_t total = sizeof(Shm) + c->cap * world;
        int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)total) != 0) throw std::runtime_error("host harness: shm_open failed");
        c->base = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (c->base == (char*)MAP_FAILED) throw std::runtime_error("host harness: mmap failed");
        return c;          // a fresh segment is zero-filled: count = gen = 0
    }
    static void comm_info(const Comm* c, int* out) { out[0] = c ? c->world : 1; out[1] = c ? c->rank : 0; out[2] = 0; out[3] = -1; }   // (no RCCL in the harness)
    static void comm_free(Comm* c) {
        if (!c) return;
        if (c->base) munmap(c->base, sizeof(Shm) + c->cap * c->world);
        if (c->rank == 0) shm_unlink(c->name.c_str());
        delete c;
    }
    static void barrier(Comm* c) {
        Shm* h = (Shm*)c->base;
        const int g = h->gen.load();
        if (h->count.fetch_add(1) + 1 == c->world) { h->count.store(0); h->gen.fetch_add(1); }
        else while (h->gen.load() == g) sched_yield();
    }
    // The product's SIDE STREAM is emulated by a worker thread with a FIFO of closures, so that the ordering rules of the
    // runtime (side waits for main at enqueue time -- trivially true here, the main "stream" is the calling thread --, main
    // waits for the "gather done" event of a record slot before rewriting it, sync_side) are exercised on the CPU.
    // MW_TEST_GATHER_DELAY_MS makes every gather slow (a slow rank / first-call RCCL setup); MW_TEST_NO_BACKEDGE=1 disables
    // wait_gather_done, which the test uses to show that it CAN see torn / late records.
    struct Side {
        std::thread th; std::mutex mu; std::condition_variable cv;
        std::deque<std::function<void()>> q;
        int pending[2] = {0, 0}, inflight = 0;
        bool stop = false;
        std::vector<int> log;          // episode_length of record 0 as seen by every gather, in order (mwh_test_gather_log)
        Side() { th = std::thread([this] { run(); }); }
        ~Side() { { std::lock_guard<std::mutex> l(mu); stop = true; } cv.notify_all(); th.join(); }
        void run() {
            std::unique_lock<std::mutex> l(mu);
            for (;;) {
                cv.wait(l, [this] { return stop || !q.empty(); });
                if (q.empty()) return;
                auto f = std::move(q.front()); q.pop_front();
                inflight = 1;
                l.unlock(); f(); l.lock();
                inflight = 0;
                cv.notify_all();
            }
        }
        void push(std::function<void()> f) { { std::lock_guard<std::mutex> l(mu); q.push_back(std::move(f)); } cv.notify_all(); }
        void drain() { std::unique_lock<std::mutex> l(mu); cv.wait(l, [this] { return q.empty() && !inflight; }); }
    };
    static Side& side() { static Side s; return s; }
    static void gather_now(Comm* c, const void* send, void* recv, size_t bytes) {
        if (const char* d = std::getenv("MW_TEST_GATHER_DELAY_MS")) std::this_thread::sleep_for(std::chrono::milliseconds(std::atoi(d)));
        if (!c) std::memcpy(recv, send, bytes);
        else {
            if (bytes > c->cap) throw std::runtime_error("host harness: all-gather block larger than the shared segment");
            char* data = c->base + sizeof(Shm);
            std::memcpy(data + c->cap * c->rank, send, bytes);
            barrier(c);
            for (int r = 0; r < c->world; r++) std::memcpy((char*)recv + bytes * r, data + c->cap * r, bytes);
            barrier(c);
        }
    }
    static void allgather_side(Comm* c, const void* send, void* recv, size_t bytes, int slot) {
        Side& s = side();
        { std::lock_guard<std::mutex> l(s.mu); s.pending[slot & 1]++; }
        s.push([c, send, recv, bytes, slot, &s] {
            gather_now(c, send, recv, bytes);
            std::lock_guard<std::mutex> l(s.mu);
            if (bytes >= sizeof(mw_bookkeeping)) s.log.push_back(((const mw_bookkeeping*)recv)[0].episode_length);
            s.pending[slot & 1]--;
        });
    }
    static void wait_gather_done(int slot) {
        if (std::getenv("MW_TEST_NO_BACKEDGE")) return;
        Side& s = side();
        std::unique_lock<std::mutex> l(s.mu);
        s.cv.wait(l, [&] { return s.pending[slot & 1] == 0; });
    }
    static void sync_side() { side().drain(); }
    static void copy_side(void* d, const void* s, size_t n, bool) { side().push([d, s, n] { std::memcpy(d, s, n); }); }
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
        // room for `rows` rows of the widest scene (Scratchpad::max_rows caps narrower ones) + the slots in front of the rows and the
        // chain transients of the widest scene (Env::chain_lds; MW_CHAIN_LDS=0: body-level chains through the column store)
        static const int chain = std::getenv("MW_CHAIN_LDS") ? std::atoi(std::getenv("MW_CHAIN_LDS")) : 2;
        const size_t slots = (size_t)rows * (mw::SR_N + mw::MAX_NV) + 8 * mw::MAX_NV + 2 + 18 * 64 + mw::TLS_SLOTS;
        const int words = (int)slots * 2;
#pragma omp parallel
        {
            std::vector<double> pad(slots + 1, std::nan(""));   // LDS is not zero-initialised either
#pragma omp for schedule(dynamic)
            for (int b = 0; b < nblocks; b++)
                for (int t = 0; t < 64; t++) f(b, t, mw::Scratchpad{pad.data(), words, ns, rows, chain});
        }
    }
    template <class F>
    static void launch_flat(int n, F f) { for (int i = 0; i < n; i++) f(i); }
    // flat wave kernels of the split collision: one call per "wave" (the host forms of mid_phase_env / narrow_wave do a whole
    // environment / all items per call), in order -- the work-item ids are then deterministic
    static int narrow_waves() { return 1; }
    template <class F>
    static void launch_waves(int nwaves, int lds_bytes, F f) {
        std::vector<double> pad((size_t)lds_bytes / 8 + 64, std::nan(""));
        for (int i = 0; i < nwaves; i++) f(i, 0, nwaves, mw::WaveLds{(void*)pad.data()});
    }
    static int compute_units() { return 256; }
    static void wait_for_caller(void*) {}
    static void caller_waits_for_us(void*) {}
    static void record_done() {}
    static void wait_done() {}
    static void* alloc_host(size_t bytes) { return std::malloc(bytes ? bytes : 16); }
    static void free_host(void* p) { std::free(p); }
    static void sync() {}
    static std::chrono::steady_clock::time_point& t0() { static std::chrono::steady_clock::time_point t; return t; }
    static std::vector<std::chrono::steady_clock::time_point>& marks() { static std::vector<std::chrono::steady_clock::time_point> m; return m; }
    static void timed_begin() { marks().clear(); t0() = std::chrono::steady_clock::now(); }
    static void timed_mark() { marks().push_back(std::chrono::steady_clock::now()); }
    static int launch_times(float* out, int cap) {
        int n = 0;
        for (; n < (int)marks().size() && n < cap; n++)
            out[n] = std::chrono::duration<float, std::milli>(marks()[n] - (n == 0 ? t0() : marks()[n - 1])).count();
        return n;
    }
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
extern "C" void mwh_pairstat(long* out, int reset) {
    for (int i = 0; i < 256; i++) { out[i] = mw::mw_pairstat()[i]; if (reset) mw::mw_pairstat()[i] = 0; }
}
#endif
#ifdef MW_PROFILE
extern "C" void mwh_hist(long* out, int reset) {
    for (int i = 0; i < 256; i++) { out[i] = mw::mw_hist()[i]; if (reset) mw::mw_hist()[i] = 0; }
}
#endif

// TEST HOOK (host harness only): the episode_length of record 0 as every all-gather of this process saw it, oldest first;
// returns the number of gathers logged and clears the log when out == nullptr
extern "C" int mwh_test_gather_log(int* out, int cap) {
    auto& s = Backend::side();
    s.drain();
    std::lock_guard<std::mutex> l(s.mu);
    const int n = (int)s.log.size();
    if (!out) { s.log.clear(); return n; }
    for (int i = 0; i < n && i < cap; i++) out[i] = s.log[i];
    return n;
}

