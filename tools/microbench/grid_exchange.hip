// Microbenchmark (round 5): what one step-to-step hand-over costs when ALL the tiles of a 10^5-particle set stay inside one
// running kernel -- a persistent step loop across the 8 XCDs -- against the kernel boundary the per-step launch pays.
// Every iteration each of G workgroups (256 threads) writes 96 bytes per thread (6 float4: the size of a pose + genealogy
// record), publishes a tagged total, waits until the totals of ALL G tiles carry this iteration's tag (each thread polls
// ceil(G / 256) entries; no counter: 392 arrivals on one word would serialise), then gathers the 96 bytes of a thread of
// another tile (another XCD, usually) and checks them: a stale read is counted.
//   protocol "fence":   streaming stores, agent-scope release fence (buffer_wbl2 sc1), publish; poll; agent-scope acquire
//                       fence (buffer_inv sc1); plain loads;
//   protocol "bypass":  streaming stores drained (s_waitcnt vmcnt(0)), publish; poll; gathers as agent-scope relaxed atomic
//                       loads (sc1: they miss in the reader's L1 and L2) -- correct only if streaming stores leave nothing
//                       dirty in the WRITER's L2; the check says whether they did;
//   protocol "through": stores as agent-scope relaxed atomic stores (sc1: written through), drained, publish; poll; gathers as
//                       in "bypass".
//   "boundary":         the same traffic as one launch per iteration.
// Every spin is bounded and an abort word ends everybody's loop: a kernel whose workgroups are not all resident gives up.
// usage: grid_exchange [G = 392] [iterations = 2000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int kB = 256, kRec = 6;

struct Status {
    unsigned abort_, gave_up, stale, pad;
};

__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int PROTO>
__global__ void __launch_bounds__(kB) exchange_kernel(float4 *data, unsigned long long *tot, Status *st, int G, int iters, int work, float *sink) {
    const int b = blockIdx.x, t = threadIdx.x;
    const size_t n = (size_t) G * kB;
    const int i = b * kB + t;
    unsigned stale = 0;
    float acc = 0.f;
    __shared__ int sh_ok;
    for (int it = 0; it < iters; it++) {
        const int gen = it & 1;
        float4 *dg = data + (size_t) gen * kRec * n;
        for (int k = 0; k < kRec; k++) {
            const float4 v = make_float4((float) it, (float) b, (float) t, (float) k + acc * 0.f);
            if (PROTO == 2) {
                unsigned long long *q = reinterpret_cast<unsigned long long *>(dg + (size_t) k * n + i);
                __hip_atomic_store(q, ((unsigned long long) __float_as_uint(v.y) << 32) | __float_as_uint(v.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(q + 1, ((unsigned long long) __float_as_uint(v.w) << 32) | __float_as_uint(v.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __builtin_nontemporal_store((v4f){v.x, v.y, v.z, v.w}, reinterpret_cast<v4f *>(dg + (size_t) k * n + i));
            }
        }
        if (PROTO == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned long long *tg = tot + (size_t) gen * G;
        if (t == 0) __hip_atomic_store(tg + b, ((unsigned long long) (unsigned) (it + 1) << 32) | __float_as_uint(1.0f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // poll the table
        unsigned spins = 0;
        float wsum = 0.f;
        for (;;) {
            bool ok = true;
            wsum = 0.f;
            for (int e = t; e < G; e += kB) {
                const unsigned long long v = ld_agent(tg + e);
                ok = ok && (unsigned) (v >> 32) == (unsigned) (it + 1);
                wsum += __uint_as_float((unsigned) v);
            }
            const int all = __syncthreads_and(ok ? 1 : 0);
            if (all) break;
            if (++spins > 200000u || __hip_atomic_load(&st->abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                spins = ~0u;
                break;
            }
        }
        if (spins == ~0u) {
            if (t == 0) {
                __hip_atomic_store(&st->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicAdd(&st->gave_up, 1u);
            }
            break;
        }
        if (PROTO == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const size_t a = ((size_t) i + (size_t) 12347 * (size_t) (it + 1) * 257) % n;
        bool bad = false;
        for (int k = 0; k < kRec; k++) {
            float4 v;
            const float4 *src = dg + (size_t) k * n + a;
            if (PROTO == 0) {
                v = *src;
            } else {
                const unsigned long long lo = ld_agent(reinterpret_cast<const unsigned long long *>(src)), hi = ld_agent(reinterpret_cast<const unsigned long long *>(src) + 1);
                v = make_float4(__uint_as_float((unsigned) lo), __uint_as_float((unsigned) (lo >> 32)), __uint_as_float((unsigned) hi), __uint_as_float((unsigned) (hi >> 32)));
            }
            bad = bad || v.x != (float) it || v.y != (float) (a / kB) || v.z != (float) (a % kB) || v.w != (float) k;
            acc += v.x;
        }
        if (bad) stale++;
        // stand-in for the step's arithmetic: `work` dependent multiply-adds
        for (int k = 0; k < work; k++) acc = acc * 1.0000001f + wsum * 1e-9f;
    }
    if (stale) atomicAdd(&st->stale, stale);
    sink[i] = acc;
}

__global__ void __launch_bounds__(kB) boundary_kernel(float4 *data, unsigned long long *tot, Status *st, int G, int it, int work, float *sink) {
    const int b = blockIdx.x, t = threadIdx.x;
    const size_t n = (size_t) G * kB;
    const int i = b * kB + t;
    float acc = 0.f;
    // what the previous launch left: the totals table and the records of another tile's thread
    const int pg = (it + 1) & 1;
    float wsum = 0.f;
    for (int e = t; e < G; e += kB) wsum += __uint_as_float((unsigned) tot[(size_t) pg * G + e]);
    const size_t a = ((size_t) i + (size_t) 12347 * (size_t) (it + 1) * 257) % n;
    bool bad = false;
    for (int k = 0; k < kRec; k++) {
        const float4 v = data[(size_t) pg * kRec * n + (size_t) k * n + a];
        bad = bad || (it > 0 && (v.x != (float) (it - 1) || v.y != (float) (a / kB) || v.z != (float) (a % kB) || v.w != (float) k));
        acc += v.x;
    }
    if (bad) atomicAdd(&st->stale, 1u);
    for (int k = 0; k < work; k++) acc = acc * 1.0000001f + wsum * 1e-9f;
    const int gen = it & 1;
    for (int k = 0; k < kRec; k++) {
        const float4 v = make_float4((float) it, (float) b, (float) t, (float) k + acc * 0.f);
        __builtin_nontemporal_store((v4f){v.x, v.y, v.z, v.w}, reinterpret_cast<v4f *>(data + (size_t) gen * kRec * n + (size_t) k * n + i));
    }
    if (t == 0) tot[(size_t) gen * G + b] = ((unsigned long long) (unsigned) (it + 1) << 32) | __float_as_uint(1.0f);
    if (acc == -1.f) sink[i] = acc;
}

int main(int argc, char **argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 392;
    const int iters = argc > 2 ? atoi(argv[2]) : 2000;
    if (G < 1 || G > 512) {  // every workgroup must be resident at once: 256 CUs x 2 is a safe ceiling for this kernel
        fprintf(stderr, "G out of range\n");
        return 1;
    }
    const size_t n = (size_t) G * kB;
    float4 *data;
    float *sink;
    unsigned long long *tot;
    Status *st;
    CK(hipMalloc(&data, sizeof(float4) * 2 * kRec * n));
    CK(hipMalloc(&sink, sizeof(float) * n));
    CK(hipMalloc(&tot, sizeof(unsigned long long) * 2 * G));
    CK(hipMalloc(&st, sizeof(Status)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("G = %d workgroups of %d threads, %d iterations, %d bytes written and %d gathered per thread and iteration\n", G, kB, iters, kRec * 16, kRec * 16);
    for (int work : {0, 2000}) {
        printf("-- %d dependent multiply-adds per iteration as the step's arithmetic\n", work);
        auto run = [&](const char *name, auto launch, bool persistent) {
            Status h{};
            float ms = 0.f;
            for (int rep = 0; rep < 2; rep++) {
                CK(hipMemset(st, 0, sizeof(Status)));
                CK(hipMemset(tot, 0, sizeof(unsigned long long) * 2 * G));
                CK(hipMemset(data, 0, sizeof(float4) * 2 * kRec * n));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                launch();
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            CK(hipMemcpy(&h, st, sizeof(Status), hipMemcpyDeviceToHost));
            printf("%-28s %6.2f us per %s; stale gathers %u, workgroups that gave up %u\n", name, 1e3 * ms / iters, persistent ? "iteration" : "launch", h.stale, h.gave_up);
            fflush(stdout);
        };
        run("kernel boundary", [&] { for (int it = 0; it < iters; it++) hipLaunchKernelGGL(boundary_kernel, dim3(G), dim3(kB), 0, 0, data, tot, st, G, it, work, sink); }, false);
        run("in-kernel, fence", [&] { hipLaunchKernelGGL(exchange_kernel<0>, dim3(G), dim3(kB), 0, 0, data, tot, st, G, iters, work, sink); }, true);
        run("in-kernel, bypass", [&] { hipLaunchKernelGGL(exchange_kernel<1>, dim3(G), dim3(kB), 0, 0, data, tot, st, G, iters, work, sink); }, true);
        run("in-kernel, through", [&] { hipLaunchKernelGGL(exchange_kernel<2>, dim3(G), dim3(kB), 0, 0, data, tot, st, G, iters, work, sink); }, true);
    }
    return 0;
}
