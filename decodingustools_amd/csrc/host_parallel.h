// host_parallel.h -- the host side's thread helper: DUT_THREADS (default: the machine's cores, at
// most 16) worker threads, work handed out in chunks through an atomic counter.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <exception>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

namespace dut {

inline int worker_threads()
{
    static const int n = [] {
        const char *e = getenv("DUT_THREADS");
        int v = e ? atoi(e) : 0;
        if (v <= 0) v = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        return v;
    }();
    return n;
}

// joins its threads when it goes out of scope, whichever way
struct JoinAll {
    std::vector<std::thread> &th;
    ~JoinAll() { for (auto &t : th) if (t.joinable()) t.join(); }
};

// fn(i) for i in [0, n), chunks of `grain` consecutive i per hand-out.  An exception thrown by fn (bad_alloc from a
// vector that grows, say) never ends a worker thread -- that would be std::terminate --: the first one is kept, the
// remaining chunks are skipped, every thread is joined, and it is rethrown on the calling thread, where the C ABI
// wrappers turn it into a status.
template <class F>
void parallel_for(size_t n, size_t grain, F fn)
{
    if (grain == 0) grain = 1;
    const size_t chunks = (n + grain - 1) / grain;
    const int nt = (int)std::min<size_t>((size_t)worker_threads(), chunks);
    if (nt <= 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
    std::atomic<size_t> next{0};
    std::exception_ptr first;
    std::mutex first_mu;
    auto body = [&]() noexcept {
        try {
            for (;;) {
                const size_t c = next.fetch_add(1);
                if (c >= chunks) break;
                const size_t e = std::min(n, (c + 1) * grain);
                for (size_t i = c * grain; i < e; ++i) fn(i);
            }
        } catch (...) {
            next.store(chunks);                                   // nobody starts another chunk
            std::lock_guard<std::mutex> g(first_mu);
            if (!first) first = std::current_exception();
        }
    };
    {
        std::vector<std::thread> th;
        JoinAll join{th};
        th.reserve((size_t)nt);
        for (int t = 1; t < nt; ++t) {
            try { th.emplace_back(body); } catch (const std::system_error &) { break; }   // no more threads: the ones we have do the work
        }
        body();
    }
    if (first) std::rethrow_exception(first);
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

} // namespace dut
