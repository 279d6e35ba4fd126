// What does the x gather of the Transport-shaped SpMV cost on gfx950?  (timing experiment)
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/gather_cost.hip -o tools/micro/gather_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// MODE 0: stream val+col only. 1: + gather x[col]. 2: + products to LDS + barrier + thread-per-row sum (15/row) + y store
template <int MODE>
__global__ void __launch_bounds__(256) k(const double *val, const unsigned *col, const double *x, size_t n, double *y)
{
    __shared__ double prod[2048];
    const unsigned tid = threadIdx.x;
    size_t base = (size_t)blockIdx.x * 2040;     // 136 rows x 15
    double v[8]; unsigned c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { size_t j = base + tid + i * 256; bool ok = j < n && tid + i * 256 < 2040; v[i] = ok ? val[j] : 0.0; c[i] = ok ? col[j] : 0u; }
    double acc = 0.0;
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += v[i] * (double)c[i];
        if (acc == 1.2345e-300) y[blockIdx.x] = acc;
    } else if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += v[i] * x[c[i]];
        if (acc == 1.2345e-300) y[blockIdx.x] = acc;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) prod[tid + i * 256] = v[i] * x[c[i]];
        __syncthreads();
        if (tid < 136) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 15; ++k) s += prod[tid * 15 + k];
            y[(size_t)blockIdx.x * 136 + tid] = s;
        }
    }
}

int main(int argc, char **argv)
{
    const long nrows = 1602111;
    const long offs[15] = {-13807, -13806, -13690, -13689, -118, -117, -1, 0, 1, 117, 118, 13689, 13690, 13806, 13807};
    std::vector<unsigned> col((size_t)nrows * 15);
    for (long r = 0; r < nrows; ++r)
        for (int k = 0; k < 15; ++k) { long c = r + offs[k]; if (c < 0) c = 0; if (c >= nrows) c = nrows - 1; col[(size_t)r * 15 + k] = (unsigned)c; }
    const size_t n = col.size();
    std::vector<unsigned> col_seq(n);
    for (size_t j = 0; j < n; ++j) col_seq[j] = (unsigned)(j % nrows);       // perfectly coalesced "gather"
    double *val, *x, *y; unsigned *dcol, *dseq;
    CK(hipMalloc(&val, n * 8)); CK(hipMalloc(&x, nrows * 8)); CK(hipMalloc(&y, nrows * 8 + 4096)); CK(hipMalloc(&dcol, n * 4)); CK(hipMalloc(&dseq, n * 4));
    CK(hipMemset(val, 0, n * 8)); CK(hipMemset(x, 0, nrows * 8));
    CK(hipMemcpy(dcol, col.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dseq, col_seq.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const unsigned nb = (unsigned)((n + 2039) / 2040);
    auto timeit = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(a));
        for (int i = 0; i < 50; ++i) launch();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 50;
        printf("%-44s %7.1f us\n", name, ms * 1e3);
    };
    timeit("stream val+col", [&] { hipLaunchKernelGGL(k<0>, dim3(nb), dim3(256), 0, 0, val, dcol, x, n, y); });
    timeit("+ gather x[col] (banded, 15 diagonals)", [&] { hipLaunchKernelGGL(k<1>, dim3(nb), dim3(256), 0, 0, val, dcol, x, n, y); });
    timeit("+ gather x[col] (col = j % n, coalesced)", [&] { hipLaunchKernelGGL(k<1>, dim3(nb), dim3(256), 0, 0, val, dseq, x, n, y); });
    timeit("+ LDS + barrier + row sums + y (banded)", [&] { hipLaunchKernelGGL(k<2>, dim3(nb), dim3(256), 0, 0, val, dcol, x, n, y); });
    timeit("+ LDS + barrier + row sums + y (coalesced)", [&] { hipLaunchKernelGGL(k<2>, dim3(nb), dim3(256), 0, 0, val, dseq, x, n, y); });
    return 0;
}
