// Micro test: wave-wide argmax under (value desc, index asc) by DPP moves + v_readlane against the __shfl_xor tree, on random
// data with absent lanes (v < 0) and ties.  hipcc --offload-arch=gfx950 -O3 -I distgcn_amd/csrc -I include tools/micro/wave_argmax.hip -o /tmp/wave_argmax && /tmp/wave_argmax
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "wave_reduce.h"
using namespace dgcn;

template <int CTRL>
__device__ __forceinline__ void step(double& p, int& v) {
    const double op = wave_dpp_f64<CTRL>(p);
    const int ov = __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
    if (ov >= 0 && (v < 0 || op > p || (op == p && ov < v))) { p = op; v = ov; }
}
__global__ void k(const double* p_in, const int* v_in, double* p_a, int* v_a, double* p_b, int* v_b, double* s_a, double* s_b) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    double p = p_in[i]; int v = v_in[i];
    {   // reference: shuffle tree
        double bp = p; int bv = v;
        for (int off = 1; off < 64; off <<= 1) {
            const double op = __shfl_xor(bp, off); const int ov = __shfl_xor(bv, off);
            if (ov >= 0 && (bv < 0 || op > bp || (op == bp && ov < bv))) { bp = op; bv = ov; }
        }
        p_a[i] = bp; v_a[i] = bv;
    }
    {
        double bp = p; int bv = v;
        step<0xB1>(bp, bv); step<0x4E>(bp, bv); step<0x141>(bp, bv); step<0x140>(bp, bv);
        double wp = 0.0; int wv = -1;
        for (int r = 0; r < 4; ++r) {
            const int rv = __builtin_amdgcn_readlane(bv, 16 * r);
            const double rp = wave_readlane_f64(bp, 16 * r);
            if (rv >= 0 && (wv < 0 || rp > wp || (rp == wp && rv < wv))) { wp = rp; wv = rv; }
        }
        p_b[i] = wp; v_b[i] = wv;
    }
    double s = p_in[i];
    double t = s;
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
    s_a[i] = t;
    s_b[i] = wave_sum_f64(s);
}
int main() {
    const int W = 4096, N = W * 64;
    double *p, *pa, *pb, *sa, *sb; int *v, *va, *vb;
    hipMallocManaged(&p, N * 8); hipMallocManaged(&pa, N * 8); hipMallocManaged(&pb, N * 8); hipMallocManaged(&sa, N * 8); hipMallocManaged(&sb, N * 8);
    hipMallocManaged(&v, N * 4); hipMallocManaged(&va, N * 4); hipMallocManaged(&vb, N * 4);
    srand(7);
    for (int w = 0; w < W; ++w) {
        const int dens = w % 5;  // 0: nobody, 1: sparse, ..
        for (int l = 0; l < 64; ++l) {
            const int i = w * 64 + l;
            p[i] = (rand() % 8) * 0.25 - 0.5;  // many ties, negatives
            v[i] = (dens == 0 || (dens < 4 && rand() % (dens == 1 ? 16 : 3))) ? -1 : w * 64 + l;
            if (w % 7 == 3 && l >= 10) v[i] = -1;  // only the first lanes (a small graph)
        }
    }
    hipLaunchKernelGGL(k, dim3(W), dim3(64), 0, 0, p, v, pa, va, pb, vb, sa, sb);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
    int bad = 0, bads = 0;
    for (int i = 0; i < N; ++i) {
        if (va[i] != vb[i] || (va[i] >= 0 && pa[i] != pb[i])) { if (bad < 5) printf("argmax differs at %d: (%g, %d) vs (%g, %d)\n", i, pa[i], va[i], pb[i], vb[i]); ++bad; }
        if (sa[i] != sb[i] && !(fabs(sa[i] - sb[i]) <= 1e-12 * fabs(sa[i]))) { if (bads < 5) printf("sum differs at %d: %g vs %g\n", i, sa[i], sb[i]); ++bads; }
    }
    printf("argmax mismatches %d of %d, sum mismatches %d\n", bad, N, bads);
    return bad || bads;
}
