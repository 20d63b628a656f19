// What does a chain of dependent ~20 us kernels cost per link: plain launches on one stream against the same chain as a
// captured hipGraph (one hipGraphLaunch per chain)?   hipcc --offload-arch=gfx950 -O3 -o graph_gap graph_gap.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_work(float* p, int n, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = p[i];
    for (int k = 0; k < iters; ++k) v = fmaf(v, 1.0000001f, 0.5f);
    p[i] = v;
}

int main() {
    const int n = 1 << 20, links = 25, reps = 200;
    float* d;
    hipMalloc(&d, n * sizeof(float));
    hipMemset(d, 0, n * sizeof(float));
    hipStream_t s;
    hipStreamCreate(&s);
    for (int iters : {200, 2000, 6000}) {
        // time of one kernel alone
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, d, n, iters);
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, d, n, iters);
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        float one = 0.f;
        hipEventElapsedTime(&one, e0, e1);
        // plain chain
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r)
            for (int l = 0; l < links; ++l) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, d, n, iters);
        hipStreamSynchronize(s);
        const double plain = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        // captured chain
        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int l = 0; l < links; ++l) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, d, n, iters);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int i = 0; i < 5; ++i) hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        const double graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        printf("kernel %.1f us (event pair around one launch); chain of %d: plain %.1f us (%.2f per link over the kernel), graph %.1f us (%.2f per link)\n",
               one * 1e3, links, plain, plain / links - one * 1e3, graph, graph / links - one * 1e3);
        hipGraphExecDestroy(ge);
        hipGraphDestroy(g);
    }
    return 0;
}
