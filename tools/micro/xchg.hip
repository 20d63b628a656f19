// Micro-benchmark: cost of a per-layer exchange between K workgroups of one graph (gfx950).
// Each workgroup writes its share of a 25.6 KB slice (N = 200 rows x 128 B) to a double-buffered global buffer,
// publishes a flag, waits for the other K - 1 flags and reads the other shares into LDS.  Variants:
//   placement 0: the K workgroups have consecutive block indices (different XCDs if dispatch is round-robin)
//   placement 1: block index = (g / 8) * 8K + c * 8 + g % 8 (same blockIdx % 8 = same XCD)
//   fence 0: agent-scope release / acquire (__threadfence / atomics at agent scope)
//   fence 1: s_waitcnt only on the writer, loads that bypass the L1 on the reader (valid on one XCD only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int FENCE>
__global__ __launch_bounds__(512) void k(float* xz, int* flags, int K, int placement, int layers, int nrows, unsigned long long* cyc,
                                         int* xcc, int* bad) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int g, c;
    if (placement) { const int grp = blockIdx.x / (8 * K), rem = blockIdx.x % (8 * K); c = rem / 8; g = grp * 8 + rem % 8; }
    else { g = blockIdx.x / K; c = blockIdx.x % K; }
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[g * K + c] = (int)(id & 0xf);
    }
    const int rows_per = (nrows + K - 1) / K;
    const int r0 = c * rows_per, r1 = min(nrows, r0 + rows_per);
    float* slice = xz + (size_t)g * 2 * nrows * 32;
    int* fl = flags + g * K;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < layers; ++l) {
        float* buf = slice + (size_t)(l & 1) * nrows * 32;
        // write own rows (value encodes layer and row so the reader can check)
        for (int i = threadIdx.x; i < (r1 - r0) * 8; i += blockDim.x) {
            const int row = r0 + i / 8, ch = i % 8;
            const float v = (float)(l * 1000 + row);
            f32x4 val = {v, v + 0.25f, v + 0.5f, (float)ch};
            *reinterpret_cast<f32x4*>(buf + row * 32 + ch * 4) = val;
        }
        if (FENCE == 0) __threadfence();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            if (FENCE == 0) __hip_atomic_store(&fl[c], l + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_store(&fl[c], l + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if ((int)threadIdx.x < K) {
            int spins = 0;
            while (__hip_atomic_load(&fl[threadIdx.x], FENCE == 0 ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < l + 1) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) { atomicOr(bad, 1); break; }
            }
        }
        __syncthreads();
        if (FENCE == 0) __threadfence();
        // read the other rows into LDS
        for (int i = threadIdx.x; i < nrows * 8; i += blockDim.x) {
            const int row = i / 8, ch = i % 8;
            if (row >= r0 && row < r1) continue;
            f32x4 val;
            if (FENCE == 0) val = *reinterpret_cast<const f32x4*>(buf + row * 32 + ch * 4);
            else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(val) : "v"(buf + row * 32 + ch * 4) : "memory");
            if (val[0] != (float)(l * 1000 + row) || val[3] != (float)ch) atomicOr(bad, 2);
            *reinterpret_cast<f32x4*>(lds + row * 32 + ch * 4) = val;
        }
        __syncthreads();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[g * K + c] = t1 - t0;
}

int main() {
    const int nrows = 200, layers = 200;
    float* xz; int *flags, *xcc, *bad; unsigned long long* cyc;
    const int maxB = 32, maxK = 8;
    (void)hipMalloc(&xz, (size_t)maxB * 2 * nrows * 128); (void)hipMalloc(&flags, maxB * maxK * 4); (void)hipMalloc(&xcc, maxB * maxK * 4);
    (void)hipMalloc(&bad, 4); (void)hipMalloc(&cyc, maxB * maxK * 8);
    for (int K : {2, 4, 8})
        for (int B : {1, 8, 32})
            for (int placement = 0; placement < 2; ++placement)
                for (int fence = 0; fence < 2; ++fence) {
                    if (placement && B % 8) continue;
                    (void)hipMemset(flags, 0, maxB * maxK * 4); (void)hipMemset(bad, 0, 4); (void)hipMemset(xz, 0, (size_t)maxB * 2 * nrows * 128);
                    if (fence == 0) hipLaunchKernelGGL(k<0>, dim3(B * K), dim3(512), nrows * 128, 0, xz, flags, K, placement, layers, nrows, cyc, xcc, bad);
                    else hipLaunchKernelGGL(k<1>, dim3(B * K), dim3(512), nrows * 128, 0, xz, flags, K, placement, layers, nrows, cyc, xcc, bad);
                    (void)hipDeviceSynchronize();
                    std::vector<unsigned long long> c(B * K); std::vector<int> x(B * K); int hb = 0;
                    (void)hipMemcpy(c.data(), cyc, B * K * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(x.data(), xcc, B * K * 4, hipMemcpyDeviceToHost);
                    (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
                    double avg = 0; for (auto v : c) avg += v; avg /= c.size();
                    int same = 1; for (int g = 0; g < B; ++g) for (int q = 1; q < K; ++q) same &= x[g * K + q] == x[g * K];
                    printf("K=%d B=%2d placement %d fence %d: %7.0f cycles per exchange (%.2f us at 100 MHz ticks?)  same XCD: %d  bad: %d\n", K, B, placement,
                           fence, avg / layers, avg / layers / 100.0, same, hb);
                }
    return 0;
}
