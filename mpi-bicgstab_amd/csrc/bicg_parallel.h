// bicg_parallel.h -- the set-up's loops over slices / groups / row ranges on several host threads.
// Every loop handed to parallel_ranges writes locations that belong to its own indices only, so the results do not depend on the
// number of threads (BICG_PLAN_THREADS; default: the hardware's threads divided by the ranks of the job, at most 32).
#pragma once

#include <sched.h>

#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <thread>
#include <vector>

namespace bicg {

inline int &plan_threads_setting()
{
    static int n = 0;                                  // 0: not set (bicg_set_plan_threads / BICG_PLAN_THREADS)
    return n;
}
// the ranks of the job as the communicator knows them (comm_set): the fallback when the launcher's environment does not say
// how many ranks share this host
inline int &plan_ranks_hint()
{
    static int n = 1;
    return n;
}
inline int ranks_on_this_host()
{
    // torchrun, Open MPI, MPICH / hydra, Slurm -- in that order; else every rank of the communicator (one node)
    for (const char *name : {"LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS", "SLURM_NTASKS_PER_NODE"})
        if (const char *sv = getenv(name)) { const int v = atoi(sv); if (v > 0) return v; }
    return std::max(1, plan_ranks_hint());
}
// hardware threads THIS process may run on (affinity mask / cpuset, what the loader's threads honour too), not the machine's
inline unsigned usable_hw_threads()
{
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) return (unsigned)c; }
    const unsigned hw = std::thread::hardware_concurrency();
    return hw ? hw : 1u;
}
inline int plan_threads()
{
    int &n = plan_threads_setting();
    if (n > 0) return n;
    if (const char *sv = getenv("BICG_PLAN_THREADS")) { n = std::max(1, atoi(sv)); return n; }
    return (int)std::min<unsigned>(32u, std::max<unsigned>(1u, usable_hw_threads() / (unsigned)ranks_on_this_host()));
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
