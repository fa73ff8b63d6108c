// Microbenchmark (round 4): issue cadence of ONE wave on a SIMD for dependent and independent VALU instructions (fp32 FMA,
// 32-bit integer multiply, packed fp32 FMA, transcendental), in shader cycles per instruction (s_memtime is a constant
// 100 MHz clock: converted with the measured cycles of a reference loop of s_nop... no: reported in ns per instruction
// and, with the 2.4 GHz shader clock of the part, in cycles).  usage: valu_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kIter = 4096;

template <int CHAINS, int KIND>
__global__ void __launch_bounds__(64) k(float *out, float a, float b) {
    float x[CHAINS];
    f2 p[CHAINS];
    unsigned u[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
        x[c] = a + c + threadIdx.x;
        p[c] = f2{a + c, b + threadIdx.x};
        u[c] = (unsigned) (threadIdx.x + c + 12345u);
    }
    for (int i = 0; i < kIter / 8; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (KIND == 0) x[c] = __builtin_fmaf(x[c], a, b);
                if (KIND == 1) u[c] = u[c] * 0xD2511F53u + (unsigned) r;
                if (KIND == 2) p[c] = __builtin_elementwise_fma(p[c], f2{a, a}, f2{b, b});
                if (KIND == 3) x[c] = __builtin_amdgcn_rcpf(x[c]) + b;
                if (KIND == 4) u[c] = __umulhi(u[c], 0xD2511F53u) ^ (unsigned) r;
                if (KIND == 5) {
                    const unsigned long long pr = (unsigned long long) 0xD2511F53u * u[c];
                    u[c] = (unsigned) (pr >> 32) ^ (unsigned) pr ^ (unsigned) r;
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += x[c] + p[c].x + p[c].y + (float) u[c];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

int main() {
    float *out;
    CK(hipMalloc(&out, 4 * 64 * 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch, int chains, int per) {
        launch(1);
        CK(hipDeviceSynchronize());
        // one wave per SIMD at most: 256 blocks of one wave = one wave per CU
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(e0));
            launch(256);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        const double ns = best * 1e6 / ((double) kIter * chains * per);
        printf("%-44s %d chain(s): %.2f ns per instruction and wave = %.1f cycles at 2.4 GHz\n", name, chains, ns, ns * 2.4);
    };
#define RUN(NAME, C, KIND, PER) run(NAME, [&](int g) { hipLaunchKernelGGL((k<C, KIND>), dim3(g), dim3(64), 0, 0, out, 1.0001f, 0.5f); }, C, PER)
    RUN("v_fma_f32", 1, 0, 1);
    RUN("v_fma_f32", 2, 0, 1);
    RUN("v_fma_f32", 4, 0, 1);
    RUN("v_fma_f32", 8, 0, 1);
    RUN("v_mul_lo_u32 + v_add (2 instr)", 1, 1, 2);
    RUN("v_mul_lo_u32 + v_add (2 instr)", 2, 1, 2);
    RUN("v_mul_lo_u32 + v_add (2 instr)", 4, 1, 2);
    RUN("v_mul_hi_u32 + v_xor (2 instr)", 1, 4, 2);
    RUN("v_mul_hi_u32 + v_xor (2 instr)", 2, 4, 2);
    RUN("v_mul_hi_u32 + v_xor (2 instr)", 4, 4, 2);
    RUN("v_mad_u64_u32 + v_xor3 (2 instr)", 1, 5, 2);
    RUN("v_mad_u64_u32 + v_xor3 (2 instr)", 2, 5, 2);
    RUN("v_mad_u64_u32 + v_xor3 (2 instr)", 4, 5, 2);
    RUN("v_mad_u64_u32 + v_xor3 (2 instr)", 8, 5, 2);
    RUN("v_pk_fma_f32", 1, 2, 1);
    RUN("v_pk_fma_f32", 2, 2, 1);
    RUN("v_pk_fma_f32", 4, 2, 1);
    RUN("v_rcp_f32 + v_add (2 instr)", 1, 3, 2);
    RUN("v_rcp_f32 + v_add (2 instr)", 2, 3, 2);
    RUN("v_rcp_f32 + v_add (2 instr)", 4, 3, 2);
    return 0;
}
