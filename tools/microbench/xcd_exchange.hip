// Microbenchmark (round 4): what one exchange between FOUR workgroups costs inside a running kernel -- the step of a persistent
// small-N loop -- against the kernel boundary it would replace.  Every iteration each workgroup writes
// 1 KB (256 floats), all four meet at a counter, each reads the 1 KB of its neighbour and checks it (a stale read is counted).
//   placement "one XCD":   grid of 32, the workgroups with blockIdx % 8 == 0 take part (the dispatcher deals workgroups round-robin
//                          to the 8 XCDs: those four share an L2), the others exit;
//   placement "four XCDs": grid of 4 (workgroups 0..3 land on XCDs 0..3).
//   protocol "model":  agent-scope release on the counter, agent-scope acquire on the poll (what the memory model asks for);
//   protocol "light":  stores drained (s_waitcnt vmcnt(0)), relaxed agent-scope counter, reads as agent-scope relaxed atomic loads
//                      (they bypass the CU's vector cache) -- enough only if the four share an L2; the check says whether it was.
// Every spin is bounded: a workgroup that waits too long gives up and says so.
// usage: xcd_exchange [iterations]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Shared {
    unsigned counter;
    unsigned gave_up;
    unsigned stale;
    unsigned pad;
};

template <bool MODEL>
__global__ void __launch_bounds__(256) exchange_kernel(float *data, Shared *sh, int iters, int stride, float *sink) {
    if (blockIdx.x % stride != 0) return;
    const int b = blockIdx.x / stride, t = threadIdx.x;
    if (b >= 4) return;
    float acc = 0.f;
    unsigned stale = 0;
    __shared__ int ok;
    for (int it = 0; it < iters; it++) {
        float *mine = data + ((size_t) (it & 1) * 4 + b) * 256;  // (two generations: a writer may run one iteration ahead of a reader)
        mine[t] = (float) (it * 4 + b) + 0.001f * t;
        if (MODEL) {
            __syncthreads();
            if (t == 0) {
                __hip_atomic_fetch_add(&sh->counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                ok = 1;
                while (__hip_atomic_load(&sh->counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < 4u * (unsigned) (it + 1))
                    if (++spins > 4000000u) {
                        ok = 0;
                        break;
                    }
            }
            __syncthreads();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) {
                __hip_atomic_fetch_add(&sh->counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                ok = 1;
                while (__hip_atomic_load(&sh->counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 4u * (unsigned) (it + 1))
                    if (++spins > 4000000u) {
                        ok = 0;
                        break;
                    }
            }
            __syncthreads();
        }
        if (!ok) {
            if (t == 0) atomicAdd(&sh->gave_up, 1u);
            break;
        }
        const int nb = (b + 1) & 3;
        const float *theirs = data + ((size_t) (it & 1) * 4 + nb) * 256;
        const float v = MODEL ? theirs[t] : __hip_atomic_load(theirs + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != (float) (it * 4 + nb) + 0.001f * t) stale++;
        acc += v;
    }
    if (stale) atomicAdd(&sh->stale, stale);
    sink[b * 256 + t] = acc;
}

__global__ void __launch_bounds__(256) boundary_kernel(float *data, int it, float *sink) {  // the same traffic, one launch per iteration
    const int b = blockIdx.x, t = threadIdx.x;
    const float *theirs = data + ((size_t) ((it + 1) & 1) * 4 + ((b + 1) & 3)) * 256;
    const float v = theirs[t];
    data[((size_t) (it & 1) * 4 + b) * 256 + t] = v + 1.0f;
    if (v == -1.f) sink[b * 256 + t] = v;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    float *data, *sink;
    Shared *sh;
    CK(hipMalloc(&data, sizeof(float) * 2 * 4 * 256));
    CK(hipMalloc(&sink, sizeof(float) * 4 * 256));
    CK(hipMalloc(&sh, sizeof(Shared)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch, bool persistent) {
        Shared h{};
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemset(sh, 0, sizeof(Shared)));
            CK(hipMemset(data, 0, sizeof(float) * 2 * 4 * 256));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
        }
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(&h, sh, sizeof(Shared), hipMemcpyDeviceToHost));
        if (persistent)
            printf("%-44s %.2f us per exchange; stale reads %u, workgroups that gave up %u\n", name, 1e3 * ms / iters, h.stale, h.gave_up);
        else
            printf("%-44s %.2f us per launch\n", name, 1e3 * ms / iters);
    };
    run("kernel boundary (4 workgroups per launch)", [&] { for (int it = 0; it < iters; it++) hipLaunchKernelGGL(boundary_kernel, dim3(4), dim3(256), 0, 0, data, it, sink); }, false);
    run("one XCD, protocol of the memory model", [&] { hipLaunchKernelGGL(exchange_kernel<true>, dim3(32), dim3(256), 0, 0, data, sh, iters, 8, sink); }, true);
    run("one XCD, light protocol", [&] { hipLaunchKernelGGL(exchange_kernel<false>, dim3(32), dim3(256), 0, 0, data, sh, iters, 8, sink); }, true);
    run("four XCDs, protocol of the memory model", [&] { hipLaunchKernelGGL(exchange_kernel<true>, dim3(4), dim3(256), 0, 0, data, sh, iters, 1, sink); }, true);
    run("four XCDs, light protocol", [&] { hipLaunchKernelGGL(exchange_kernel<false>, dim3(4), dim3(256), 0, 0, data, sh, iters, 1, sink); }, true);
    return 0;
}
