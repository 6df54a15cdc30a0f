// host_parallel.h -- the host side's thread helper: DUT_THREADS threads (default: what the box gives this process --
// its affinity mask cut by the cgroup's CPU quota, shared among the ranks of the node, at most 64) -- the caller and a
// pool of persistent workers --, work handed out in chunks through an atomic counter.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <exception>
#include <functional>
#include <mutex>
#include <new>
#include <utility>
#include <system_error>
#include <thread>
#include <vector>

#include <pthread.h>
#include <cstring>
#include <sched.h>
#include <cstdio>

namespace dut {

// CPUs this process may use: the affinity mask, cut by the cgroup's CFS quota (a box can show 256 CPUs and grant 16
// CPUs' worth of time: threads beyond the quota only add throttling)
inline int cpu_budget()
{
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = std::min(n, c); }
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                       // cgroup v2: "<quota|max> <period>"
        char q[32] = {0}; long long per = 0;
        if (fscanf(f, "%31s %lld", q, &per) == 2 && q[0] != 'm' && per > 0) {
            const long long quota = atoll(q);
            if (quota > 0) n = std::min<long long>(n, std::max<long long>(1, (quota + per - 1) / per));
        }
        fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
        long long quota = -1, per = 0;
        if (fscanf(g, "%lld", &quota) != 1) quota = -1;
        fclose(g);
        if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &per) != 1) per = 0; fclose(h); }
        if (quota > 0 && per > 0) n = std::min<long long>(n, std::max<long long>(1, (quota + per - 1) / per));
    }
    return n;
}

inline int worker_threads()
{
    static const int n = [] {
        const char *e = getenv("DUT_THREADS");
        int v = e ? atoi(e) : 0;
        if (v <= 0) {
            // one process per GPU: the ranks of a node share its CPUs (LOCAL_WORLD_SIZE is what torch.distributed.run sets)
            const char *lw = getenv("LOCAL_WORLD_SIZE");
            const int ranks = lw && atoi(lw) > 0 ? atoi(lw) : 1;
            v = std::max(1, std::min(64, cpu_budget() / ranks));
        }
        return v;
    }();
    return n;
}

// hand-out size for a loop over n records whose cost per record varies by orders of magnitude (short reads: millions
// of records of a few CIGAR operations; long reads: a few hundred thousand of thousands each): at most `most`, small
// enough that every thread gets several hand-outs
inline size_t grain_for(size_t n, size_t most)
{
    return std::min(most, std::max<size_t>(64, n / (16 * (size_t)worker_threads())));
}

// The workers: worker_threads() - 1 threads that live as long as the process and sleep between jobs (a contig passes
// through some twenty parallel loops; starting and joining fifteen threads for each was a tenth of its host time, and
// much more whenever another thread of the process was faulting pages in at the time -- thread stacks are mapped under
// the same lock).  A job is a counter of chunks; whoever is idle takes chunks of the first job that has some left, the
// caller included, so loops started from several threads at once, or from inside a chunk, share the workers and always
// make progress on the caller's own thread.
namespace detail {
struct Job {
    std::atomic<size_t> next{0};
    size_t chunks = 0;
    void (*run)(void *ctx, size_t chunk) = nullptr;   // must not throw
    void *ctx = nullptr;
    int active = 0;                                   // workers inside the job (guarded by Pool::mu)
};

struct Pool {
    std::mutex mu;
    std::condition_variable work_cv, done_cv;
    std::vector<Job *> jobs;                          // guarded by mu
    std::vector<std::thread> workers;
    bool stop = false;

    explicit Pool(int n)
    {
        workers.reserve((size_t)std::max(0, n));
        for (int i = 0; i < n; ++i) {
            try { workers.emplace_back([this] { loop(); }); } catch (const std::system_error &) { break; }   // fewer workers, same result
        }
    }
    ~Pool()
    {
        { std::lock_guard<std::mutex> g(mu); stop = true; }
        work_cv.notify_all();
        for (auto &t : workers) if (t.joinable()) t.join();
    }
    Pool(const Pool &) = delete;
    Pool &operator=(const Pool &) = delete;

    void loop()
    {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            if (stop) return;
            Job *j = nullptr;
            for (Job *c : jobs) if (c->next.load(std::memory_order_relaxed) < c->chunks) { j = c; break; }
            if (!j) { work_cv.wait(lk); continue; }
            j->active += 1;                           // under the lock: the caller cannot miss it
            lk.unlock();
            for (;;) {
                const size_t c = j->next.fetch_add(1);
                if (c >= j->chunks) break;
                j->run(j->ctx, c);
            }
            lk.lock();
            if (--j->active == 0) done_cv.notify_all();
        }
    }

    // runs the job to completion with the caller taking part; returns when no thread is inside it any more
    void run(Job &job)
    {
        if (!workers.empty() && job.chunks > 1) {
            { std::lock_guard<std::mutex> g(mu); jobs.push_back(&job); }
            const size_t wake = std::min(job.chunks - 1, workers.size());
            if (wake >= workers.size()) work_cv.notify_all(); else for (size_t i = 0; i < wake; ++i) work_cv.notify_one();
        }
        for (;;) {
            const size_t c = job.next.fetch_add(1);
            if (c >= job.chunks) break;
            job.run(job.ctx, c);
        }
        if (!workers.empty() && job.chunks > 1) {
            std::unique_lock<std::mutex> lk(mu);
            jobs.erase(std::find(jobs.begin(), jobs.end(), &job));      // no worker enters from here on
            done_cv.wait(lk, [&] { return job.active == 0; });
        }
    }
};

inline std::atomic<Pool *> &pool_ptr() { static std::atomic<Pool *> p{nullptr}; return p; }

// The process's pool, made on first use and never destroyed (its threads sleep on a condition variable that must
// outlive every static destructor).  A forked child starts without one: the parent's threads do not exist there.
inline Pool *pool()
{
    Pool *p = pool_ptr().load(std::memory_order_acquire);
    if (p) return p;
    static const int reg = pthread_atfork(nullptr, nullptr, [] { pool_ptr().store(nullptr, std::memory_order_release); });
    (void)reg;
    Pool *made = new Pool(worker_threads() - 1);
    if (pool_ptr().compare_exchange_strong(p, made, std::memory_order_acq_rel)) return made;
    delete made;                                      // another thread was first
    return p;
}
} // namespace detail

// fn(i) for i in [0, n), chunks of `grain` consecutive i per hand-out.  An exception thrown by fn (bad_alloc from a
// vector that grows, say) never ends a worker thread -- that would be std::terminate --: the first one is kept, the
// remaining chunks are skipped, every thread has left the loop, and it is rethrown on the calling thread, where the
// C ABI wrappers turn it into a status.
template <class F>
void parallel_for(size_t n, size_t grain, F fn)
{
    if (grain == 0) grain = 1;
    const size_t chunks = (n + grain - 1) / grain;
    if (chunks <= 1 || worker_threads() <= 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
    struct Ctx {
        F &fn; size_t n, grain, chunks;
        detail::Job job;
        std::exception_ptr first;
        std::mutex first_mu;
    } ctx{fn, n, grain, chunks, {}, nullptr, {}};
    ctx.job.chunks = chunks;
    ctx.job.ctx = &ctx;
    ctx.job.run = [](void *v, size_t c) {
        Ctx &x = *static_cast<Ctx *>(v);
        try {
            const size_t e = std::min(x.n, (c + 1) * x.grain);
            for (size_t i = c * x.grain; i < e; ++i) x.fn(i);
        } catch (...) {
            x.job.next.store(x.chunks);                           // nobody starts another chunk
            std::lock_guard<std::mutex> g(x.first_mu);
            if (!x.first) x.first = std::current_exception();
        }
    };
    detail::pool()->run(ctx.job);
    if (ctx.first) std::rethrow_exception(ctx.first);
}

// A thread that is joined when it goes out of scope (an exception unwinding past a joinable std::thread is
// std::terminate), and before it is assigned over.
class Thread {
    std::thread t_;
public:
    Thread() = default;
    Thread(std::thread &&t) : t_(std::move(t)) {}
    Thread(Thread &&) = default;
    Thread &operator=(Thread &&o) { if (t_.joinable()) t_.join(); t_ = std::move(o.t_); return *this; }
    ~Thread() { if (t_.joinable()) t_.join(); }
    bool joinable() const { return t_.joinable(); }
    void join() { t_.join(); }
};

// Starts fn on a new thread; when the system has no thread to give, runs it here and now instead.  fn must not
// throw: an exception on a helper thread has nowhere to go (the callers' lambdas only call C ABI functions, which
// catch everything, or catch themselves).  Returns a thread that is joinable only in the first case.
template <class F>
Thread spawn_or_run(F fn)
{
    auto guarded = [fn]() mutable noexcept { try { fn(); } catch (...) {} };
    try { return Thread(std::thread(guarded)); } catch (const std::system_error &) {}
    guarded();
    return Thread();
}

// A crew of threads that is started once and sleeps between jobs: start(n, fn) runs fn(t) for t in [0, n) on n of them
// and returns at once, wait() returns when all are done.  One job at a time (the pinned staging ring, which owns one,
// is held by one transfer at a time).
struct Crew {
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::vector<std::thread> th;
    std::function<void(int)> fn;
    int want = 0, running = 0;
    uint64_t gen = 0;
    bool quit = false;
    void loop(int id)
    {
        uint64_t seen = 0;
        for (;;) {
            std::function<void(int)> f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&] { return quit || (gen != seen && id < want); });
                if (quit) return;
                seen = gen; f = fn;
            }
            try { f(id); } catch (...) {}
            { std::lock_guard<std::mutex> g(mu); if (--running == 0) cv_done.notify_all(); }
        }
    }
    // fn(t) for t in [0, n) on the crew's threads; returns at once (wait() joins).  When the system has no thread to
    // give, the calls run here and now, one after the other.
    bool ensure_locked(int n)
    {
        while ((int)th.size() < n) {
            const int id = (int)th.size();
            try { th.emplace_back([this, id] { loop(id); }); } catch (const std::system_error &) { return false; }
        }
        return true;
    }
    void ensure(int n) { std::lock_guard<std::mutex> g(mu); (void)ensure_locked(n); }   // (at cl_create: none is made inside a transfer)
    void start(int n, std::function<void(int)> f)
    {
        bool have = true;
        {
            std::lock_guard<std::mutex> g(mu);
            have = ensure_locked(n);
            if (have) { fn = std::move(f); want = n; running = n; ++gen; }
        }
        if (have) { cv_job.notify_all(); return; }
        for (int t = 0; t < n; ++t) { try { f(t); } catch (...) {} }
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return running == 0; });
        fn = nullptr;
    }
    ~Crew()
    {
        { std::lock_guard<std::mutex> g(mu); quit = true; }
        cv_job.notify_all();
        for (std::thread &t : th) if (t.joinable()) t.join();
    }
};

// ---------------------------------------------------------------------------------------------
// Large temporaries of the per-contig host stages (reference spans, name hashes, the buckets of the name count: some
// 450 MB for a chr21-sized contig).  Taken from malloc, each would be mapped, faulted in page by page and unmapped
// again for every contig -- and an unmap of that size holds the address-space lock against every other thread that
// wants a page or a stack meanwhile.  A few blocks are kept here from one contig to the next instead (at most
// kScratchKeep blocks; the smallest one goes when a larger one comes back).  (Asking for transparent huge pages for
// these and the staging arrays instead -- MADV_HUGEPAGE, which this pool's boxes honour -- made a fresh process slower,
// not faster: the BAM reader's parse 70 -> 350 ms, profiles/r04_first_pass_stages.txt.)
// ---------------------------------------------------------------------------------------------
struct ScratchPool {
    static constexpr size_t kScratchKeep = 8;
    static constexpr size_t kMinPooled = 1u << 20;          // smaller blocks are plain malloc / free
    std::mutex mu;
    std::vector<std::pair<void *, size_t>> idle;
    void *take(size_t bytes, size_t *cap)
    {
        if (bytes >= kMinPooled) {
            std::lock_guard<std::mutex> g(mu);
            size_t best = idle.size();
            for (size_t i = 0; i < idle.size(); ++i)
                if (idle[i].second >= bytes && (best == idle.size() || idle[i].second < idle[best].second)) best = i;
            if (best != idle.size()) {
                void *p = idle[best].first; *cap = idle[best].second;
                idle.erase(idle.begin() + (long)best);
                return p;
            }
        }
        const size_t want = bytes >= kMinPooled ? bytes + bytes / 8 : bytes;      // (a little room: the next contig is rarely the same size)
        *cap = want;
        return malloc(want ? want : 1);
    }
    void give(void *p, size_t cap)
    {
        if (!p) return;
        if (cap >= kMinPooled) {
            std::lock_guard<std::mutex> g(mu);
            if (idle.size() < kScratchKeep) { idle.emplace_back(p, cap); return; }
            size_t small = 0;
            for (size_t i = 1; i < idle.size(); ++i) if (idle[i].second < idle[small].second) small = i;
            if (idle[small].second < cap) { void *q = idle[small].first; idle[small] = std::make_pair(p, cap); p = q; }
        }
        free(p);                                            // the block that was not kept
    }
};
inline ScratchPool &scratch_pool() { static ScratchPool *p = new ScratchPool(); return *p; }   // (never destroyed: threads may outlive main)

template <class T> class Scratch {
    T *p_ = nullptr;
    size_t cap_ = 0;
public:
    Scratch() = default;
    explicit Scratch(size_t n)
    {
        p_ = static_cast<T *>(scratch_pool().take((n ? n : 1) * sizeof(T), &cap_));
        if (!p_) throw std::bad_alloc();
    }
    Scratch(Scratch &&o) noexcept : p_(o.p_), cap_(o.cap_) { o.p_ = nullptr; o.cap_ = 0; }
    Scratch &operator=(Scratch &&o) noexcept { if (this != &o) { reset(); p_ = o.p_; cap_ = o.cap_; o.p_ = nullptr; o.cap_ = 0; } return *this; }
    Scratch(const Scratch &) = delete;
    Scratch &operator=(const Scratch &) = delete;
    ~Scratch() { reset(); }
    void reset() { if (p_) scratch_pool().give(p_, cap_); p_ = nullptr; cap_ = 0; }
    T *get() const { return p_; }
    explicit operator bool() const { return p_ != nullptr; }
};

} // namespace dut
