// Microbenchmark (round 4): what the landmark loop of the big-map update kernel can pull from HBM, by how its records are
// requested.  Layout and access shape of the product (kernels.h): recA float4[J][S], recB float[J][S], particle index fastest;
// one thread per particle, 256-thread blocks; every thread walks the SAME list of m landmark rows (random rows of J) and, per
// row, reads its 20-byte record, does `work` dependent FMAs on it and (pass 2) writes 20 bytes into the row's other buffer.
//   variant 0: registers, one chunk of 4 landmarks requested ahead (the product's pipeline), compiler-placed waits
//   variant D >= 1: LDS-DMA (global_load_lds_dwordx4 / _dword) into a ring of D + 1 chunks in LDS, D chunks in flight, counted
//              s_waitcnt vmcnt(N) placed by hand (loads, stores and LDS-DMA count together, in issue order)
// usage: record_stream <particles> <m> <work> <write 0|1>      (work < 0: |work| FMAs as four independent chains)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int CH = 4;

// `work` FMAs per record: one dependent chain (work > 0), or four independent chains of work / 4 each (work < 0): the same
// instruction count with instruction-level parallelism 4 -- what separates "bound by the bytes" from "bound by how fast ONE
// wave can issue a dependent chain" at 1.5 waves per SIMD
__device__ __forceinline__ float chew(float4 a, float b, int work, float acc) {
    float x = a.x + b, y = a.y, z = a.z, w = a.w;
    if (work >= 0) {
        for (int i = 0; i < work; i += 4) {
            x = __builtin_fmaf(x, 1.0001f, y);
            y = __builtin_fmaf(y, 0.9999f, z);
            z = __builtin_fmaf(z, 1.0002f, w);
            w = __builtin_fmaf(w, 0.9998f, x);
        }
    } else {
        for (int i = 0; i < -work; i += 4) {
            x = __builtin_fmaf(x, 1.0001f, 0.5f);
            y = __builtin_fmaf(y, 0.9999f, 0.25f);
            z = __builtin_fmaf(z, 1.0002f, 0.125f);
            w = __builtin_fmaf(w, 0.9998f, 0.0625f);
        }
    }
    return acc + x + y + z + w;
}

__global__ void __launch_bounds__(256) reg_kernel(const float4 *__restrict__ A, const float *__restrict__ B, float4 *__restrict__ A2, float *__restrict__ B2,
                                                   const int *__restrict__ rows, int m, size_t S, int n, int work, int wr, float *out) {
    extern __shared__ __align__(16) unsigned char lds[];
    float4 *shA = reinterpret_cast<float4 *>(lds);
    float *shB = reinterpret_cast<float *>(lds + sizeof(float4) * CH * 256);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const auto *r = (const __attribute__((address_space(4))) int *) reinterpret_cast<uintptr_t>(rows);
    float4 ta[CH];
    float tb[CH];
    float acc = 0.f;
    auto load = [&](int k0) {
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const size_t at = (size_t) r[min(k0 + k, m - 1)] * S + i;
            ta[k] = A[at];
            tb[k] = B[at];
        }
    };
    load(0);
    for (int k0 = 0; k0 < m; k0 += CH) {
#pragma unroll
        for (int k = 0; k < CH; k++) {
            shA[k * 256 + threadIdx.x] = ta[k];
            shB[k * 256 + threadIdx.x] = tb[k];
        }
        if (k0 + CH < m) load(k0 + CH);
        const int kn = min(CH, m - k0);
        for (int k = 0; k < kn; k++) {
            const float4 a = shA[k * 256 + threadIdx.x];
            const float b = shB[k * 256 + threadIdx.x];
            acc = chew(a, b, work, acc);
            if (wr) {
                const size_t at = (size_t) r[k0 + k] * S + i;
                __builtin_nontemporal_store(acc, &B2[at]);
                typedef float v4 __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store((v4){a.x, a.y, a.z, acc}, reinterpret_cast<v4 *>(&A2[at]));
            }
        }
    }
    out[i] = acc;
}

template <int D>
__global__ void __launch_bounds__(256) dma_kernel(const float4 *__restrict__ A, const float *__restrict__ B, float4 *__restrict__ A2, float *__restrict__ B2,
                                                   const int *__restrict__ rows, int m, size_t S, int n, int work, int wr, float *out) {
    // ring of D + 1 chunks: [slot][CH][256] float4, then the same of float
    extern __shared__ __align__(16) unsigned char lds[];
    constexpr int R = D + 1;
    float4 *shA = reinterpret_cast<float4 *>(lds);
    float *shB = reinterpret_cast<float *>(lds + sizeof(float4) * R * CH * 256);
    const int i0 = blockIdx.x * 256 + threadIdx.x;
    const int i = min(i0, n - 1);  // (every lane issues its DMA: the hand-counted waits assume it)
    const int wv = threadIdx.x / 64;
    const auto *r = (const __attribute__((address_space(4))) int *) reinterpret_cast<uintptr_t>(rows);
    float acc = 0.f;
    const int nch = (m + CH - 1) / CH;
    auto issue = [&](int c) {  // chunk c (clamped rows past the end: redundant reads, counts stay exact)
        const int slot = c % R;
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const size_t at = (size_t) r[min(c * CH + k, m - 1)] * S + i;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *) (A + at),
                                             (void __attribute__((address_space(3))) *) (shA + (slot * CH + k) * 256 + wv * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *) (B + at),
                                             (void __attribute__((address_space(3))) *) (shB + (slot * CH + k) * 256 + wv * 64), 4, 0, 0);
        }
    };
#pragma unroll
    for (int c = 0; c < D; c++) issue(c);
    for (int c = 0; c < nch; c++) {
        issue(c + D);  // (the slot it lands in was read out one chunk ago)
        // chunk c's 2 CH requests are the oldest; younger: the D chunks requested after it and, when writing, the 2 CH stores
        // of each of the D chunks computed since it was requested
        // (the first D chunks have fewer stores behind them: the smaller count, a lower bound, is the safe one there)
        constexpr int kLoadsOnly = D * 2 * CH, kWithStores = (2 * D * 2 * CH) < 63 ? (2 * D * 2 * CH) : 63;
        if (wr && c >= D) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kWithStores) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLoadsOnly) : "memory");
        const int slot = c % R;
#pragma unroll
        for (int k = 0; k < CH; k++) {
            // read through inline asm: a ds_read the compiler can see makes it wait vmcnt(0) first (LDS read after an LDS-DMA
            // write that may alias), which drains the chunks in flight and undoes the overlap
            float4 a;
            float b;
            {
                const unsigned oa = (unsigned) (size_t) (const __attribute__((address_space(3))) void *) (shA + (slot * CH + k) * 256 + threadIdx.x);
                const unsigned ob = (unsigned) (size_t) (const __attribute__((address_space(3))) void *) (shB + (slot * CH + k) * 256 + threadIdx.x);
                typedef float v4f __attribute__((ext_vector_type(4)));
                v4f av;
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(av), "=&v"(b) : "v"(oa), "v"(ob) : "memory");
                a = make_float4(av.x, av.y, av.z, av.w);
            }
            const bool on = c * CH + k < m;
            const float nacc = chew(a, b, work, acc);
            acc = on ? nacc : acc;
            if (wr) {  // unconditional (clamped: the last row is rewritten with the same bytes): counts stay exact
                const size_t at = (size_t) r[min(c * CH + k, m - 1)] * S + i;
                __builtin_nontemporal_store(on ? acc : b, &B2[at]);
                typedef float v4 __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store((v4){a.x, a.y, a.z, on ? acc : a.w}, reinterpret_cast<v4 *>(&A2[at]));
            }
        }
        asm volatile("" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (i0 < n) out[i0] = acc;
}

// the part's own ceiling for the same byte counts: a flat, perfectly coalesced read (sum) and copy (nontemporal stores)
__global__ void __launch_bounds__(256) flat_read_kernel(const float4 *__restrict__ a, size_t n4, float *out) {
    float acc = 0.f;
    for (size_t k = (size_t) blockIdx.x * 256 + threadIdx.x; k < n4; k += (size_t) gridDim.x * 256) {
        const float4 v = a[k];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void __launch_bounds__(256) flat_copy_kernel(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n4) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    for (size_t k = (size_t) blockIdx.x * 256 + threadIdx.x; k < n4; k += (size_t) gridDim.x * 256) {
        const float4 v = a[k];
        __builtin_nontemporal_store((v4){v.x, v.y, v.z, v.w}, reinterpret_cast<v4 *>(&b[k]));
    }
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 100000, m = argc > 2 ? atoi(argv[2]) : 1000, work = argc > 3 ? atoi(argv[3]) : 100, wr = argc > 4 ? atoi(argv[4]) : 0;
    const int J = 2400;
    const size_t S = (size_t) (n + 255) / 256 * 256;
    float4 *A, *A2;
    float *B, *B2, *out;
    int *rows;
    CK(hipMalloc(&A, sizeof(float4) * J * S));
    CK(hipMalloc(&B, sizeof(float) * J * S));
    CK(hipMalloc(&A2, sizeof(float4) * J * S));
    CK(hipMalloc(&B2, sizeof(float) * J * S));
    CK(hipMalloc(&out, sizeof(float) * S));
    CK(hipMemset(A, 0, sizeof(float4) * J * S));
    CK(hipMemset(B, 0, sizeof(float) * J * S));
    std::vector<int> h(m);
    unsigned long long x = 88172645463325252ull;
    for (int k = 0; k < m; k++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        h[k] = (int) (x % J);
    }
    CK(hipMalloc(&rows, sizeof(int) * m));
    CK(hipMemcpy(rows, h.data(), sizeof(int) * m, hipMemcpyHostToDevice));
    const int blocks = (int) (S / 256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double bytes = (double) n * m * 20.0 * (wr ? 2 : 1);
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
        printf("%-28s n %d m %d work %d write %d: %.3f ms per pass, %.2f TB/s algorithmic\n", name, n, m, work, wr, ms, bytes / ms / 1e9);
    };
    {
        const size_t n4 = (size_t) ((double) n * m * 20.0 / 16.0);  // the same bytes read (and, when writing, written) as one flat array
        if (wr) run("flat copy (ceiling)", [&] { hipLaunchKernelGGL(flat_copy_kernel, dim3(256 * 8), dim3(256), 0, 0, A, A2, n4); });
        else run("flat read (ceiling)", [&] { hipLaunchKernelGGL(flat_read_kernel, dim3(256 * 8), dim3(256), 0, 0, A, n4, out); });
    }
    run("registers, 1 chunk ahead", [&] { hipLaunchKernelGGL(reg_kernel, dim3(blocks), dim3(256), CH * 256 * 20, 0, A, B, A2, B2, rows, m, S, n, work, wr, out); });
    run("LDS-DMA, 1 chunk in flight", [&] { hipLaunchKernelGGL(dma_kernel<1>, dim3(blocks), dim3(256), 2 * CH * 256 * 20, 0, A, B, A2, B2, rows, m, S, n, work, wr, out); });
    run("LDS-DMA, 2 chunks in flight", [&] { hipLaunchKernelGGL(dma_kernel<2>, dim3(blocks), dim3(256), 3 * CH * 256 * 20, 0, A, B, A2, B2, rows, m, S, n, work, wr, out); });
    run("LDS-DMA, 3 chunks in flight", [&] { hipLaunchKernelGGL(dma_kernel<3>, dim3(blocks), dim3(256), 4 * CH * 256 * 20, 0, A, B, A2, B2, rows, m, S, n, work, wr, out); });
    run("LDS-DMA, 5 chunks in flight", [&] { hipLaunchKernelGGL(dma_kernel<5>, dim3(blocks), dim3(256), 6 * CH * 256 * 20, 0, A, B, A2, B2, rows, m, S, n, work, wr, out); });
    return 0;
}
