// bicg_parallel.h -- the set-up's loops over slices / groups / row ranges on several host threads.
// Every loop handed to parallel_ranges writes locations that belong to its own indices only, so the results do not depend on the
// number of threads (BICG_PLAN_THREADS; default: the hardware's threads divided by the ranks of the job, at most 32).
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <thread>
#include <vector>

namespace bicg {

inline int &plan_threads_setting()
{
    static int n = 0;                                  // 0: not set yet
    return n;
}
inline int plan_threads(int ranks_on_host = 1)
{
    int &n = plan_threads_setting();
    if (n > 0) return n;
    if (const char *sv = getenv("BICG_PLAN_THREADS")) { n = std::max(1, atoi(sv)); return n; }
    const unsigned hw = std::thread::hardware_concurrency();
    n = (int)std::min<unsigned>(32u, std::max<unsigned>(1u, (hw ? hw : 1u) / (unsigned)std::max(1, ranks_on_host)));
    return n;
}

// f(begin, end, part) over [0, n) cut into contiguous ranges, one per thread (part = 0 .. parts - 1; returns parts)
template <class F>
int parallel_ranges(size_t n, size_t min_per_thread, F f)
{
    int parts = plan_threads();
    if (min_per_thread > 0) parts = (int)std::min<size_t>((size_t)parts, std::max<size_t>(1, n / min_per_thread));
    if (parts <= 1 || n == 0) { f((size_t)0, n, 0); return 1; }
    std::vector<std::thread> th;
    th.reserve(parts - 1);
    for (int p = 1; p < parts; ++p) th.emplace_back([&, p] { f(n * (size_t)p / parts, n * (size_t)(p + 1) / parts, p); });
    f((size_t)0, n / parts, 0);
    for (auto &t : th) t.join();
    return parts;
}

}  // namespace bicg
