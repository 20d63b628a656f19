// Host-to-device copy rate of hipHostMalloc'd memory by allocation flag and stream kind (8.6 MB, the packed C3 batch).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
int main() {
    const size_t n = 8600000;
    void* d; (void)hipMalloc(&d, n);
    const unsigned flags[] = {hipHostMallocDefault, hipHostMallocNonCoherent, hipHostMallocPortable, hipHostMallocCoherent, hipHostMallocNumaUser};
    const char* names[] = {"default", "non-coherent", "portable", "coherent", "numa-user"};
    for (int sk = 0; sk < 2; ++sk) {
        hipStream_t st;
        (void)hipStreamCreateWithFlags(&st, sk ? hipStreamNonBlocking : hipStreamDefault);
        for (int f = 0; f < 5; ++f) {
            void* h = nullptr;
            if (hipHostMalloc(&h, n, flags[f]) != hipSuccess) { printf("%s: alloc failed\n", names[f]); continue; }
            memset(h, 1, n);
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            for (int i = 0; i < 3; ++i) (void)hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st);
            (void)hipStreamSynchronize(st);
            (void)hipEventRecord(e0, st);
            for (int i = 0; i < 20; ++i) (void)hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st);
            (void)hipEventRecord(e1, st);
            (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            auto t0 = std::chrono::steady_clock::now();
            (void)hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st);
            auto t1 = std::chrono::steady_clock::now();
            hipEvent_t e2; (void)hipEventCreateWithFlags(&e2, hipEventDisableTiming);
            (void)hipEventRecord(e2, st);
            (void)hipEventSynchronize(e2);
            auto t2 = std::chrono::steady_clock::now();
            printf("%s stream, %-13s %.3f ms per copy (%.1f GB/s); one copy: call %.3f ms, wait %.3f ms\n", sk ? "non-blocking" : "default     ",
                   names[f], ms / 20, n / (ms / 20) / 1e6, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                   std::chrono::duration<double, std::milli>(t2 - t1).count());
            (void)hipHostFree(h);
        }
    }
    return 0;
}
