// Micro-benchmark: cost of ds_read_b128 under the bank-conflict patterns of the GCN aggregation (gfx950).
// Rows of 128 B in LDS; every lane reads one 16-byte chunk per instruction.
//   mode 0: 8 lanes per row, lanes rotated by 4 inside each 16-lane row (the round-2 8-lane map), random rows
//   mode 1: 4 lanes per row, upper/lower half chosen by slot ((s >> 1) & 1), random rows (the 16-row map)
//   mode 2: as 1, but the two slots of a bank group that read the same half get rows of different parity (conflict-free)
//   mode 3: as 1, every row of a wave has the same parity (two-way conflicts everywhere)
//   mode 4: as 1, all slots read the lower half, every row the same parity (four-way conflicts)
//   mode 5: 4 lanes per row, half chosen by slot, row parity FORCED equal to ((s ^ (s >> 2)) & 1) (an alternative conflict-free pattern)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) const float lds_f;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const f32x4 lds_cf4;

template <int MODE>
__global__ void k(const unsigned* __restrict__ idx, float* out, int iters, int nrows, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < nrows * 32; i += blockDim.x) lds[i] = (float)(i & 1023);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned off;   // byte offset inside the row
    int s;          // row slot of this lane
    if (MODE == 0) {
        const int rho = (((lane & 15) + 4) & 15) | (lane & 48);
        s = rho >> 3;
        off = (unsigned)(rho & 7) << 4;
    } else {
        s = lane >> 2;
        const int half = (MODE == 4) ? 0 : ((s >> 1) & 1);
        off = (unsigned)((lane & 3) | (half << 2)) << 4;
    }
    unsigned r[8];
    const int wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (int i = 0; i < 8; ++i) {
        unsigned u = idx[(wid * 16 * 8 + s * 8 + i) & ((1 << 20) - 1)] % (unsigned)nrows;
        if (MODE == 2) {
            // bank groups: slots {0,3,5,6} and {1,2,4,7} (and + 8).  Same-half pairs: (0,5),(3,6),(1,4),(2,7)
            const int odd = (s & 7) == 5 || (s & 7) == 6 || (s & 7) == 4 || (s & 7) == 7;
            u = (u & ~1u) | (unsigned)odd;
        }
        if (MODE == 3 || MODE == 4) u &= ~1u;
        if (MODE == 5) u = (u & ~1u) | (unsigned)((s ^ (s >> 2)) & 1);
        if (u >= (unsigned)nrows) u -= 2;
        r[i] = u;
    }
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    unsigned ad[8];
    for (int i = 0; i < 8; ++i) ad[i] = (r[i] & 127u) * 128u + off;   // rows 0..127; bit 7 of the address = row parity
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const unsigned flip = (unsigned)(it & 63) << 8;   // another row of the same parity, one VALU op per gather
        f32x4 z[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(z[i]) : "v"(ad[i] ^ flip));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; i += 4) { acc0 += z[i][0]; acc1 += z[i + 1][1]; acc2 += z[i + 2][2]; acc3 += z[i + 3][3]; }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0 + acc1 + acc2 + acc3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, unsigned* didx, float* dout, unsigned long long* dcyc) {
    const int iters = 2000, nrows = 128, waves = 16, threads = 1024, blocks = 256;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), nrows * 128, 0, didx, dout, iters, nrows, dcyc);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> c(blocks);
    (void)hipMemcpy(c.data(), dcyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto x : c) avg += x;
    avg /= blocks;
    printf("%-64s %6.2f CU-cycles per ds_read_b128 (16 waves per CU)\n", name, avg / (iters * 8.0 * waves));
}

int main() {
    unsigned* didx; float* dout; unsigned long long* dcyc;
    std::vector<unsigned> h(1 << 20);
    unsigned s = 12345;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = s >> 8; }
    (void)hipMalloc(&didx, h.size() * 4); (void)hipMemcpy(didx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&dout, 1 << 24); (void)hipMalloc(&dcyc, 1 << 16);
    run<0>("8 lanes/row, rotated map, random rows", didx, dout, dcyc);
    run<1>("4 lanes/row, half by slot, random rows", didx, dout, dcyc);
    run<2>("4 lanes/row, half by slot, same-half pairs of opposite parity", didx, dout, dcyc);
    run<3>("4 lanes/row, half by slot, all rows even", didx, dout, dcyc);
    run<4>("4 lanes/row, all lower halves, all rows even", didx, dout, dcyc);
    run<5>("4 lanes/row, half by slot, parity (s ^ s>>2) & 1", didx, dout, dcyc);
    return 0;
}
