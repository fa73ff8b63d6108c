// Microbenchmark (round 4): the landmark loop of the big-map update kernel WITH its genealogy indirection -- every record's
// address comes from a look-up (gen[row][particle] -> slot) -- in two forms:
//   "product": the product's pipeline() (kernels.hip): look-ups one chunk ahead of the records, records one chunk ahead of the
//              arithmetic, through registers + ds_write, compiler-placed waits;
//   "ring":    look-ups AND records through LDS-DMA (global_load_lds) into rings in LDS, look-ups two chunks ahead of the
//              records, records two chunks ahead of the arithmetic, hand-counted s_waitcnt vmcnt(N) (loads return in order, so
//              a look-up that is waited for drains every record load issued before it: the lead of the look-ups over the
//              records must be as deep as the records' lead over the arithmetic), LDS read through inline asm (a ds_read the
//              compiler can see makes it wait vmcnt(0) behind an LDS-DMA).
// Every result is checked against the host (a wrong count reads LDS before the data has landed).
// usage: ring_stream <particles> <m> <work> <write 0|1>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int CH = 4;
typedef float v4 __attribute__((ext_vector_type(4)));
using G1 = const void __attribute__((address_space(1))) *;
using L3 = void __attribute__((address_space(3))) *;

__device__ __forceinline__ float chew(float4 a, float b, int work) {
    float x = a.x + b, y = a.y, z = a.z, w = a.w;
    for (int i = 0; i < work; i += 4) {
        x = __builtin_fmaf(x, 1.0001f, 0.5f);
        y = __builtin_fmaf(y, 0.9999f, 0.25f);
        z = __builtin_fmaf(z, 1.0002f, 0.125f);
        w = __builtin_fmaf(w, 0.9998f, 0.0625f);
    }
    return work ? x + y + z + w : x;
}

__global__ void __launch_bounds__(256) product_kernel(const float4 *__restrict__ A, const float *__restrict__ B, float4 *__restrict__ A2,
                                                       float *__restrict__ B2, const int *__restrict__ gen, const int *__restrict__ rows, int m, size_t S,
                                                       int n, int work, int wr, float *out) {
    extern __shared__ __align__(16) unsigned char lds[];
    float4 *shA = reinterpret_cast<float4 *>(lds);
    float *shB = reinterpret_cast<float *>(lds + sizeof(float4) * CH * 256);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const auto *r = (const __attribute__((address_space(4))) int *) reinterpret_cast<uintptr_t>(rows);
    int sl[CH];
    float4 ta[CH];
    float tb[CH];
    float acc = 0.f;
    auto load_slots = [&](int k0) {
#pragma unroll
        for (int k = 0; k < CH; k++) sl[k] = gen[(size_t) r[min(k0 + k, m - 1)] * S + i];
    };
    auto load_recs = [&](int k0) {
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const size_t at = (size_t) r[min(k0 + k, m - 1)] * S + sl[k];
            ta[k] = A[at];
            tb[k] = B[at];
        }
    };
    load_slots(0);
    load_recs(0);
    load_slots(CH);
    for (int k0 = 0; k0 < m; k0 += CH) {
#pragma unroll
        for (int k = 0; k < CH; k++) {
            shA[k * 256 + threadIdx.x] = ta[k];
            shB[k * 256 + threadIdx.x] = tb[k];
        }
        if (k0 + CH < m) {
            load_recs(k0 + CH);
            load_slots(k0 + 2 * CH);
        }
        const int kn = min(CH, m - k0);
        for (int k = 0; k < kn; k++) {
            const float4 a = shA[k * 256 + threadIdx.x];
            const float b = shB[k * 256 + threadIdx.x];
            const float c = chew(a, b, work);
            acc += c;
            if (wr) {
                const size_t at = (size_t) r[k0 + k] * S + i;
                __builtin_nontemporal_store(c, &B2[at]);
                __builtin_nontemporal_store((v4){a.x, a.y, a.z, c}, reinterpret_cast<v4 *>(&A2[at]));
            }
        }
    }
    out[i] = acc;
}

template <int WR>
__global__ void __launch_bounds__(256) ring_kernel(const float4 *__restrict__ A, const float *__restrict__ B, float4 *__restrict__ A2,
                                                    float *__restrict__ B2, const int *__restrict__ gen, const int *__restrict__ rows, int m, size_t S, int n,
                                                    int work, float *out) {
    constexpr int D = 2;  // records: chunks in flight ahead of the arithmetic; look-ups: the same lead over the records
    extern __shared__ __align__(16) unsigned char lds[];
    float4 *ringA = reinterpret_cast<float4 *>(lds);                                       // [D][CH][256]
    float *ringB = reinterpret_cast<float *>(lds + sizeof(float4) * D * CH * 256);         // [D][CH][256]
    int *ringS = reinterpret_cast<int *>(lds + (sizeof(float4) + sizeof(float)) * D * CH * 256);  // [D][CH][256]
    const int i0 = blockIdx.x * 256 + threadIdx.x;
    const int i = min(i0, n - 1);
    const int wv = threadIdx.x / 64;
    const auto *r = (const __attribute__((address_space(4))) int *) reinterpret_cast<uintptr_t>(rows);
    float acc = 0.f;
    const int nfull = m / CH;
    auto issue_slots = [&](int c) {  // CH operations
        const int pos = c % D;
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const size_t at = (size_t) r[min(c * CH + k, m - 1)] * S + i;
            __builtin_amdgcn_global_load_lds((G1) (gen + at), (L3) (ringS + (pos * CH + k) * 256 + wv * 64), 4, 0, 0);
        }
    };
    auto read_slots = [&](int c, int (&sl)[CH]) {
        const int pos = c % D;
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const unsigned o = (unsigned) (size_t) (const __attribute__((address_space(3))) void *) (ringS + (pos * CH + k) * 256 + threadIdx.x);
            asm volatile("ds_read_b32 %0, %1" : "=v"(sl[k]) : "v"(o) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto issue_recs = [&](int c, const int (&sl)[CH]) {  // 2 CH operations
        const int pos = c % D;
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const size_t at = (size_t) r[min(c * CH + k, m - 1)] * S + sl[k];
            __builtin_amdgcn_global_load_lds((G1) (A + at), (L3) (ringA + (pos * CH + k) * 256 + wv * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((G1) (B + at), (L3) (ringB + (pos * CH + k) * 256 + wv * 64), 4, 0, 0);
        }
    };
    auto read_recs = [&](int c, v4 (&ta)[CH], float (&tb)[CH]) {
        const int pos = c % D;
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const unsigned oa = (unsigned) (size_t) (const __attribute__((address_space(3))) void *) (ringA + (pos * CH + k) * 256 + threadIdx.x);
            const unsigned ob = (unsigned) (size_t) (const __attribute__((address_space(3))) void *) (ringB + (pos * CH + k) * 256 + threadIdx.x);
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b32 %1, %3" : "=&v"(ta[k]), "=&v"(tb[k]) : "v"(oa), "v"(ob) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // operations per iteration, in issue order: records of chunk c + D (2 CH), look-ups of chunk c + 2 D (CH), stores of chunk c (S)
    constexpr int S_ = WR ? 2 * CH : 0, kIter = 3 * CH + S_;
    // records of chunk c were issued D iterations ago, FIRST in their iteration: behind them that iteration's look-ups and
    // stores, D - 1 whole iterations, nothing of this one
    constexpr int kRecWait = CH + S_ + (D - 1) * kIter, kRecWait0 = CH + (D - 1) * 3 * CH;
    // look-ups of chunk c + D were issued D iterations ago, behind the records: behind them that iteration's stores, D - 1 whole
    // iterations, nothing of this one
    constexpr int kSlotWait = S_ + (D - 1) * kIter, kSlotWait0 = (D - 1) * 3 * CH;
    static_assert(kRecWait < 64 && D == 2, "vmcnt is six bits wide; the prologue below is written for D = 2");
    int sl[CH];
    if (nfull > 0) {
        issue_slots(0);
        issue_slots(1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CH) : "memory");  // look-ups of chunk 0 (behind them: chunk 1's)
        read_slots(0, sl);
        issue_recs(0, sl);
        issue_slots(2);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * CH) : "memory");  // look-ups of chunk 1 (behind them: records 0, look-ups 2)
        read_slots(1, sl);
        issue_recs(1, sl);
        issue_slots(3);
    }
    for (int c = 0; c < nfull; c++) {
        v4 ta[CH];
        float tb[CH];
        // (the first D iterations have no stores behind them yet: the smaller count -- a stricter wait -- is the exact one there)
        if (WR && c >= D) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kRecWait) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kRecWait0) : "memory");
        read_recs(c, ta, tb);  // ring position c % D is free from here on
        if (WR && c >= D) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kSlotWait) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kSlotWait0) : "memory");
        read_slots(c + D, sl);
        issue_recs(c + D, sl);
        issue_slots(c + 2 * D);
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const float4 a = make_float4(ta[k].x, ta[k].y, ta[k].z, ta[k].w);
            const float cc = chew(a, tb[k], work);
            acc += cc;
            if (WR) {
                const size_t at = (size_t) r[c * CH + k] * S + i;
                __builtin_nontemporal_store(cc, &B2[at]);
                __builtin_nontemporal_store((v4){a.x, a.y, a.z, cc}, reinterpret_cast<v4 *>(&A2[at]));
            }
        }
        asm volatile("" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int k = nfull * CH; k < m; k++) {  // the last, partial chunk: plain loads
        const size_t at = (size_t) r[k] * S + gen[(size_t) r[k] * S + i];
        const float4 a = A[at];
        const float cc = chew(a, B[at], work);
        acc += cc;
        if (WR) {
            const size_t ao = (size_t) r[k] * S + i;
            B2[ao] = cc;
            A2[ao] = make_float4(a.x, a.y, a.z, cc);
        }
    }
    if (i0 < n) out[i0] = acc;
}

__global__ void fill_kernel(float4 *A, float *B, int *gen, size_t S, int J) {
    const size_t k = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (k >= (size_t) J * S) return;
    const unsigned h = (unsigned) (k * 2654435761ull >> 7);
    A[k] = make_float4((float) (h & 1023) * (1.0f / 1024.0f), 1.f, 2.f, 3.f);
    B[k] = (float) ((h >> 10) & 255) * (1.0f / 256.0f);
    const size_t i = k % S;
    gen[k] = (int) (i ^ ((h >> 20) & 7));  // a neighbour within eight slots (stratified ancestors sit next to their offspring)
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 100000, m = argc > 2 ? atoi(argv[2]) : 1150, work = argc > 3 ? atoi(argv[3]) : 0, wr = argc > 4 ? atoi(argv[4]) : 0;
    const int J = 2400;
    const size_t S = (size_t) (n + 255) / 256 * 256;
    float4 *A, *A2;
    float *B, *B2, *out;
    int *gen, *rows;
    CK(hipMalloc(&A, sizeof(float4) * J * S));
    CK(hipMalloc(&B, sizeof(float) * J * S));
    CK(hipMalloc(&A2, sizeof(float4) * J * S));
    CK(hipMalloc(&B2, sizeof(float) * J * S));
    CK(hipMalloc(&gen, sizeof(int) * J * S));
    CK(hipMalloc(&out, sizeof(float) * S));
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned) (((size_t) J * S + 255) / 256)), dim3(256), 0, 0, A, B, gen, S, J);
    std::vector<int> h(m);
    unsigned long long x = 88172645463325252ull;
    for (int k = 0; k < m; k++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        h[k] = (int) (x % J);
    }
    CK(hipMalloc(&rows, sizeof(int) * m));
    CK(hipMemcpy(rows, h.data(), sizeof(int) * m, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    // expected sums for a few particles (work 0: acc = sum of a.x + b in landmark order, float)
    auto expect = [&](int i) {
        float acc = 0.f;
        for (int k = 0; k < m; k++) {
            const size_t g = (size_t) h[k] * S + i;
            const unsigned hg = (unsigned) (g * 2654435761ull >> 7);
            const size_t at = (size_t) h[k] * S + ((size_t) i ^ ((hg >> 20) & 7));
            const unsigned ha = (unsigned) (at * 2654435761ull >> 7);
            acc += (float) (ha & 1023) * (1.0f / 1024.0f) + (float) ((ha >> 10) & 255) * (1.0f / 256.0f);
        }
        return acc;
    };
    const int blocks = (int) (S / 256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double bytes = (double) n * m * 20.0 * (wr ? 2 : 1);
    auto run = [&](const char *name, auto launch) {
        CK(hipMemset(out, 0, sizeof(float) * S));
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
        int bad = 0;
        if (work == 0) {
            std::vector<float> ho(S);
            CK(hipMemcpy(ho.data(), out, sizeof(float) * S, hipMemcpyDeviceToHost));
            for (int i : {0, 1, 63, 64, 255, 256, 4097, n / 2, n - 1})
                if (i < n && std::fabs(ho[i] - expect(i)) > 1e-3f * std::fabs(expect(i))) bad++;
        }
        printf("%-34s n %d m %d work %d write %d: %.3f ms per pass, %.2f TB/s of records%s\n", name, n, m, work, wr, ms, bytes / ms / 1e9,
               work ? "" : (bad ? "  ** WRONG SUMS **" : "  (sums verified)"));
    };
    run("product pipeline (registers)", [&] { hipLaunchKernelGGL(product_kernel, dim3(blocks), dim3(256), CH * 256 * 20, 0, A, B, A2, B2, gen, rows, m, S, n, work, wr, out); });
    if (wr) run("ring (LDS-DMA, leads of 2 chunks)", [&] { hipLaunchKernelGGL(ring_kernel<1>, dim3(blocks), dim3(256), 2 * CH * 256 * 24, 0, A, B, A2, B2, gen, rows, m, S, n, work, out); });
    else run("ring (LDS-DMA, leads of 2 chunks)", [&] { hipLaunchKernelGGL(ring_kernel<0>, dim3(blocks), dim3(256), 2 * CH * 256 * 24, 0, A, B, A2, B2, gen, rows, m, S, n, work, out); });
    return 0;
}
