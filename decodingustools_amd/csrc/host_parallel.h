// host_parallel.h -- the host side's thread helper: DUT_THREADS (default: the machine's cores, at
// most 16) worker threads, work handed out in chunks through an atomic counter.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdlib>
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

// fn(i) for i in [0, n), chunks of `grain` consecutive i per hand-out
template <class F>
void parallel_for(size_t n, size_t grain, F fn)
{
    if (grain == 0) grain = 1;
    const size_t chunks = (n + grain - 1) / grain;
    const int nt = (int)std::min<size_t>((size_t)worker_threads(), chunks);
    if (nt <= 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
    std::atomic<size_t> next{0};
    auto body = [&]() {
        for (;;) {
            const size_t c = next.fetch_add(1);
            if (c >= chunks) break;
            const size_t e = std::min(n, (c + 1) * grain);
            for (size_t i = c * grain; i < e; ++i) fn(i);
        }
    };
    std::vector<std::thread> th;
    th.reserve((size_t)nt);
    for (int t = 1; t < nt; ++t) {
        try { th.emplace_back(body); } catch (const std::system_error &) { break; }   // no more threads: the ones we have do the work
    }
    body();
    for (auto &t : th) t.join();
}

// Starts fn on a new thread; when the system has no thread to give, runs it here and now instead (no exception
// leaves the library through the C ABI).  Returns a thread that is joinable only in the first case.
template <class F>
std::thread spawn_or_run(F fn)
{
    try { return std::thread(fn); } catch (const std::system_error &) {}
    fn();
    return std::thread();
}

} // namespace dut
