// Microbenchmark (round 4): what a plain copy reaches on this part, by how it is written -- the ceiling the read + write pass of
// the big-map update kernel is held against (profiles/record_stream_r04.txt).  Bytes counted: read + written.
// usage: copy_ceiling [GiB per buffer]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ void __launch_bounds__(256) copy_kernel(const v4 *__restrict__ a, v4 *__restrict__ b, size_t n4) {
    const size_t stride = (size_t) gridDim.x * 256;
    for (size_t k = (size_t) blockIdx.x * 256 + threadIdx.x; k < n4; k += stride * U) {
        v4 t[U];
#pragma unroll
        for (int u = 0; u < U; u++) t[u] = k + u * stride < n4 ? a[k + u * stride] : v4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; u++)
            if (k + u * stride < n4) {
                if (NT) __builtin_nontemporal_store(t[u], &b[k + u * stride]);
                else b[k + u * stride] = t[u];
            }
    }
}
// every block owns one contiguous slab (the update kernel's shape: a block streams its own particles' records)
template <int U, bool NT>
__global__ void __launch_bounds__(256) slab_kernel(const v4 *__restrict__ a, v4 *__restrict__ b, size_t n4) {
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
    for (size_t k = lo + threadIdx.x; k < hi; k += 256 * U) {
        v4 t[U];
#pragma unroll
        for (int u = 0; u < U; u++) t[u] = k + u * 256 < hi ? a[k + u * 256] : v4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; u++)
            if (k + u * 256 < hi) {
                if (NT) __builtin_nontemporal_store(t[u], &b[k + u * 256]);
                else b[k + u * 256] = t[u];
            }
    }
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 2.0;
    const size_t n4 = (size_t) (gib * (1ull << 30) / 16);
    v4 *a, *b;
    CK(hipMalloc(&a, n4 * 16));
    CK(hipMalloc(&b, n4 * 16));
    CK(hipMemset(a, 0, n4 * 16));
    CK(hipMemset(b, 0, n4 * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch) {
        for (int w = 0; w < 2; w++) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int reps = 5;
        for (int w = 0; w < reps; w++) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        printf("%-58s %.3f ms, %.2f TB/s (read + written)\n", name, ms, 2.0 * n4 * 16 / ms / 1e9);
    };
    run("hipMemcpyDtoD", [&] { CK(hipMemcpyAsync(b, a, n4 * 16, hipMemcpyDeviceToDevice, 0)); });
#define GRID(K, U, NT, G) run("grid-stride, " #G " blocks, " #U " float4 in flight, " #NT, [&] { hipLaunchKernelGGL((K<U, NT>), dim3(G), dim3(256), 0, 0, a, b, n4); })
    GRID(copy_kernel, 1, true, 2048);
    GRID(copy_kernel, 1, false, 2048);
    GRID(copy_kernel, 4, true, 2048);
    GRID(copy_kernel, 4, false, 2048);
    GRID(copy_kernel, 8, false, 1024);
    GRID(copy_kernel, 4, false, 4096);
    GRID(copy_kernel, 4, false, 8192);
#define SLAB(U, NT, G) run("slab per block, " #G " blocks, " #U " float4 in flight, " #NT, [&] { hipLaunchKernelGGL((slab_kernel<U, NT>), dim3(G), dim3(256), 0, 0, a, b, n4); })
    SLAB(4, true, 391);
    SLAB(4, false, 391);
    SLAB(4, false, 512);
    SLAB(4, false, 2048);
    SLAB(8, false, 2048);
    return 0;
}
