// bicg_stream.hip -- STREAM-style bandwidth probes of THIS GPU (bicg_stream_bench, include/bicgstab_hip.h).
//
// north_star prices the SpMV against "per-GPU STREAM-HBM bandwidth" and SURVEY.md section 8d asks for that number to
// be measured on the box rather than quoted: bench.py calls this before its timed legs and reports the SpMV's
// bandwidth as a fraction of the measured copy / triad / read rates next to the fraction of the 8 TB/s spec.
// Arrays are > 1 GB by default so that the 256 MiB Infinity Cache cannot serve them. gfx950 only.
#include "bicg_comm.h"
#include "bicg_knobs.h"
#include "../../include/bicgstab_hip.h"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

namespace {

typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int kThreads = 256;
constexpr int kUnroll = 4;

// kind 0: copy a = b (16-byte accesses)   kind 1: triad a = b + s c (16-byte accesses)
// kind 2: read-only, 8-byte loads         kind 3: read-only, 16-byte loads
// Two shapes, the faster one is reported: a grid-stride loop over 8192 resident-sized workgroups, and one workgroup per
// 16 KiB tile (the dispatcher balances ~100 k short workgroups); each with ordinary or non-temporal accesses.
template <bool NT> __device__ __forceinline__ f64x2 ld2(const f64x2 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st2(f64x2 *p, f64x2 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <bool NT> __device__ __forceinline__ double ld1(const double *p) { return NT ? __builtin_nontemporal_load(p) : *p; }

template <int KIND, bool NT, bool TILE>
__global__ void __launch_bounds__(kThreads) k_stream(double *__restrict__ a, const double *__restrict__ b, const double *__restrict__ c,
                                                     size_t n, double s, double *sink)
{
    // TILE: the workgroup's kUnroll * kThreads consecutive elements, once; otherwise grid-stride
    const size_t stride = TILE ? (size_t)kThreads : (size_t)gridDim.x * kThreads;
    const size_t first = TILE ? (size_t)blockIdx.x * kThreads * kUnroll + threadIdx.x : (size_t)blockIdx.x * kThreads + threadIdx.x;
    double acc = 0.0;
    if (KIND == 2) {
        for (size_t i = first; i < n; i += kUnroll * stride) {
            double v[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) v[u] = i + u * stride < n ? ld1<NT>(b + i + u * stride) : 0.0;
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) acc += v[u];
            if (TILE) break;
        }
    } else {
        const size_t n2 = n / 2;
        const f64x2 *b2 = reinterpret_cast<const f64x2 *>(b), *c2 = reinterpret_cast<const f64x2 *>(c);
        f64x2 *a2 = reinterpret_cast<f64x2 *>(a);
        for (size_t i = first; i < n2; i += kUnroll * stride) {
            f64x2 v[kUnroll], w[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const bool ok = i + u * stride < n2;
                v[u] = ok ? ld2<NT>(b2 + i + u * stride) : (f64x2)(0.0);
                if (KIND == 1) w[u] = ok ? ld2<NT>(c2 + i + u * stride) : (f64x2)(0.0);
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                if (KIND == 3) { acc += v[u].x + v[u].y; continue; }
                if (i + u * stride < n2) st2<NT>(a2 + i + u * stride, KIND == 1 ? v[u] + s * w[u] : v[u]);
            }
            if (TILE) break;
        }
    }
    if ((KIND == 2 || KIND == 3) && acc == 1.2345e-300) sink[blockIdx.x & 65535u] = acc;      // never true: keeps the loads alive
}

template <int KIND>
void launch_variant(int variant, double *a, const double *b, const double *c, size_t n, double s, double *sink, hipStream_t st)
{
    const size_t elems = KIND == 2 ? n : n / 2;                       // accesses of one thread-lane width
    const unsigned tiles = (unsigned)((elems + (size_t)kThreads * kUnroll - 1) / ((size_t)kThreads * kUnroll));
    const unsigned grid = 256u * 32u;
    switch (variant) {
    case 0: hipLaunchKernelGGL((k_stream<KIND, false, false>), dim3(grid), dim3(kThreads), 0, st, a, b, c, n, s, sink); break;
    case 1: hipLaunchKernelGGL((k_stream<KIND, true, false>), dim3(grid), dim3(kThreads), 0, st, a, b, c, n, s, sink); break;
    case 2: hipLaunchKernelGGL((k_stream<KIND, false, true>), dim3(tiles), dim3(kThreads), 0, st, a, b, c, n, s, sink); break;
    default: hipLaunchKernelGGL((k_stream<KIND, true, true>), dim3(tiles), dim3(kThreads), 0, st, a, b, c, n, s, sink); break;
    }
}

}  // namespace

extern "C" int bicg_stream_bench(int kind, unsigned long long bytes_per_array, int reps, double *gbps, double *ms_out)
{
    using namespace bicg;
    if (kind < 0 || kind > 3 || reps < 1) return 1;
    BICG_HIP(hipSetDevice(comm_get()->device));
    size_t n = (size_t)(bytes_per_array / 16) * 2;            // doubles, even
    if (n < 1024) return 1;
    double *a = nullptr, *b = nullptr, *c = nullptr, *sink = nullptr;
    BICG_HIP(hipMalloc((void **)&b, n * sizeof(double)));
    BICG_HIP(hipMemset(b, 0, n * sizeof(double)));
    if (kind <= 1) BICG_HIP(hipMalloc((void **)&a, n * sizeof(double)));
    if (kind == 1) { BICG_HIP(hipMalloc((void **)&c, n * sizeof(double))); BICG_HIP(hipMemset(c, 0, n * sizeof(double))); }
    BICG_HIP(hipMalloc((void **)&sink, 65536 * sizeof(double)));
    hipStream_t st;
    BICG_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    BICG_HIP(hipEventCreate(&e0)); BICG_HIP(hipEventCreate(&e1));
    BICG_HIP(hipDeviceSynchronize());
    auto go = [&](int variant) {
        switch (kind) {
        case 0: launch_variant<0>(variant, a, b, c, n, 0.0, sink, st); break;
        case 1: launch_variant<1>(variant, a, b, c, n, 1.0000001, sink, st); break;
        case 2: launch_variant<2>(variant, a, b, c, n, 0.0, sink, st); break;
        default: launch_variant<3>(variant, a, b, c, n, 0.0, sink, st); break;
        }
    };
    float ms = 0.f;
    for (int variant = 0; variant < 4; ++variant) {                   // the fastest shape counts
        for (int i = 0; i < 3; ++i) go(variant);
        BICG_HIP(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) go(variant);
        BICG_HIP(hipEventRecord(e1, st));
        BICG_HIP(hipEventSynchronize(e1));
        float t = 0.f;
        BICG_HIP(hipEventElapsedTime(&t, e0, e1));
        t /= (float)reps;
        if (knob_x("BICG_STREAM_VERBOSE")) fprintf(stderr, "bicg_stream_bench kind %d variant %d: %.4f ms\n", kind, variant, t);
        if (variant == 0 || t < ms) ms = t;
    }
    const double arrays = kind == 0 ? 2.0 : kind == 1 ? 3.0 : 1.0;     // bytes moved: read + written arrays
    if (gbps) *gbps = arrays * (double)n * 8.0 / ((double)ms * 1e-3) / 1e9;
    if (ms_out) *ms_out = ms;
    BICG_HIP(hipEventDestroy(e0)); BICG_HIP(hipEventDestroy(e1));
    BICG_HIP(hipStreamDestroy(st));
    for (double *p : {a, b, c, sink}) if (p) BICG_HIP(hipFree(p));
    return 0;
}
