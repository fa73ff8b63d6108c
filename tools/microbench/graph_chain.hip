// Microbenchmark (round 5): does a HIP graph shorten the period of a chain of dependent small launches?  1 000 launches of a
// kernel of 392 workgroups x 256 threads that reads what the previous launch wrote (4 bytes per thread) -- as stream launches, and
// as one instantiated graph of 1 000 kernel nodes (captured from the same stream), launched once.
// With `work` = 0 the stream's figure is the HOST's launch rate; with work the host runs ahead and both are the device's period.
// usage: graph_chain [launches = 1000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) link_kernel(const float *in, float *out, int n, int work) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float v = in[(i + 4099) % n];
    for (int k = 0; k < work; k++) v = v * 1.0000001f + 1e-9f;  // (dependent multiply-adds: the launch's own duration)
    out[i] = v + 1.0f;
}

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 1000, G = 392, n = G * 256;
    float *a, *b;
    CK(hipMalloc(&a, sizeof(float) * n));
    CK(hipMalloc(&b, sizeof(float) * n));
    CK(hipMemset(a, 0, sizeof(float) * n));
    CK(hipMemset(b, 0, sizeof(float) * n));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
  for (int work : {0, 400, 1200}) {
    auto chain = [&]() {
        for (int k = 0; k < K; k++) hipLaunchKernelGGL(link_kernel, dim3(G), dim3(256), 0, st, (k & 1) ? b : a, (k & 1) ? a : b, n, work);
    };
    float ms = 0.f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        chain();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("work %4d: %d stream launches of %d workgroups:        %.2f us per launch\n", work, K, G, 1e3 * ms / K);
    hipGraph_t graph;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    chain();
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(exec, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("work %4d: one graph of %d kernel nodes, launched once: %.2f us per node\n", work, K, 1e3 * ms / K);
    CK(hipGraphExecDestroy(exec));
    CK(hipGraphDestroy(graph));
  }
    return 0;
}
