// Do a host-to-device copy on one stream and a CU-filling kernel on another overlap on this box?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
__global__ void spin(float* out, long long cycles) {
    const long long t0 = clock64();
    float x = threadIdx.x;
    while (clock64() - t0 < cycles) x = x * 1.0001f + 0.5f;
    if (x == 12345.f) out[0] = x;
}
int main() {
    const size_t n = 8600000;
    const int S = 3;
    void *d[S], *h[S]; float* o; (void)hipMalloc(&o, 64);
    hipStream_t st[S];
    for (int i = 0; i < S; ++i) { (void)hipMalloc(&d[i], n); (void)hipHostMalloc(&h[i], n, hipHostMallocDefault); memset(h[i], 1, n); (void)hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking); }
    const long long cyc = 480000;  // ~200 us of shader clock
    auto run = [&](int mode, int blocks) {
        // mode 0: copies only, 1: kernels only, 2: copy + kernel per stream, round-robin
        for (int i = 0; i < S; ++i) (void)hipStreamSynchronize(st[i]);
        auto t0 = std::chrono::steady_clock::now();
        const int iters = 60;
        for (int k = 0; k < iters; ++k) {
            const int i = k % S;
            if (mode != 1) (void)hipMemcpyAsync(d[i], h[i], n, hipMemcpyHostToDevice, st[i]);
            if (mode != 0) hipLaunchKernelGGL(spin, dim3(blocks), dim3(512), 0, st[i], o, cyc);
        }
        for (int i = 0; i < S; ++i) (void)hipStreamSynchronize(st[i]);
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / iters;
    };
    run(2, 512);
    printf("copies only            %.3f ms per batch\n", run(0, 0));
    printf("kernels only (512 WGs) %.3f ms per batch\n", run(1, 512));
    printf("copy + kernel, 3 streams round-robin (512 WGs) %.3f ms per batch\n", run(2, 512));
    printf("copy + kernel, 3 streams round-robin (256 WGs) %.3f ms per batch\n", run(2, 256));
    return 0;
}
