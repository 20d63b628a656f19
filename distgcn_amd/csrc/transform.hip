// Feature transform Z = H . [W_0 | W_1 | ...] for every node of the batch.
// Replaces dot(x, W_i) of GraphConvolution._call (gcn/layers.py:202-203; tf.matmul / cuBLAS SGEMM
// in the reference, K2/K3 of SURVEY 2.2).
//
// The hidden 32x32 (c32) product is a dense contraction -> fp32 MFMA (v_mfma_f32_32x32x2_f32),
// which on gfx950 is bit-for-bit a k-ordered fmaf chain, so the VALU fallback (first layer,
// last layer, odd widths) and oracle/dgcn_oracle.c produce identical bits with a plain
// "acc = fmaf(h[k], w[k][n], acc)" loop.  No bf16/fp16: the 1e-5 score tolerance forbids it.
//
// Per launch the kernel is HBM-bound (reads rows*cin*4, writes rows*ctot*4 bytes); the MFMA work
// (rows*cin*ctot*2 flop) is ~3x below the fp32 matrix peak at that byte rate.
#include "common.h"

namespace dgcn {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// One wave = one 32-row tile x all CTOT columns.  256 threads = 4 waves = 128 rows per block.
template <int CIN, int CTOT>
__global__ __launch_bounds__(256) void k_transform_mfma(const float* __restrict__ H, int ldh, int rows,
                                                        const float* __restrict__ W, float* __restrict__ Z,
                                                        int ldz) {
    constexpr int KS = CIN / 2;      // MFMA k-steps (2 k per instruction)
    constexpr int CT = CTOT / 32;    // column tiles
    constexpr int LDS_LD = CIN + 1;  // +1 float: conflict-free column reads (ds_read_b32, 32 banks)
    __shared__ float tile[4][32 * LDS_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, idx = lane & 31;

    // B fragments for every (k-step, column tile): lane holds W[2s + half][ct*32 + idx]
    float bfrag[KS][CT];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) bfrag[s][ct] = W[(2 * s + half) * CTOT + ct * 32 + idx];

    float* my = tile[wave];
    for (int row0 = (blockIdx.x * 4 + wave) * 32; row0 < rows; row0 += gridDim.x * 128) {
        // ---- stage the 32 x CIN tile of H: coalesced float4 loads -> padded LDS
        constexpr int Q = CIN / 4;
#pragma unroll
        for (int t = 0; t < (32 * Q + 63) / 64; ++t) {
            const int i = lane + 64 * t;
            if (i < 32 * Q) {
                const int r = i / Q, q = i - r * Q;
                float4 h = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row0 + r < rows) h = *reinterpret_cast<const float4*>(H + (size_t)(row0 + r) * ldh + q * 4);
                float* dst = my + r * LDS_LD + q * 4;
                dst[0] = h.x; dst[1] = h.y; dst[2] = h.z; dst[3] = h.w;
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes have landed
        // A fragments: lane holds H[row0 + idx][2s + half]
        float afrag[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) afrag[s] = my[idx * LDS_LD + 2 * s + half];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[s], bfrag[s][ct], acc, 0, 0, 0);
            // C/D map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int r = row0 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                if (r < rows) Z[(size_t)r * ldz + ct * 32 + idx] = acc[reg];
            }
        }
        __builtin_amdgcn_wave_barrier();  // tile is overwritten by the next iteration
    }
}

// VALU fallback: one thread per output element, k-ordered fmaf chain (same bits as the MFMA path).
__global__ __launch_bounds__(256) void k_transform_valu(const float* __restrict__ H, int ldh, float h_const, int rows,
                                                        int cin, const float* __restrict__ W, int ctot,
                                                        float* __restrict__ Z, int ldz) {
    const long total = (long)rows * ctot;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / ctot), n = (int)(i - (long)r * ctot);
        float acc = 0.f;
        if (H) {
            const float* h = H + (size_t)r * ldh;
            for (int k = 0; k < cin; ++k) acc = fmaf(h[k], W[k * ctot + n], acc);
        } else {
            for (int k = 0; k < cin; ++k) acc = fmaf(h_const, W[k * ctot + n], acc);
        }
        Z[(size_t)r * ldz + n] = acc;
    }
}

// Narrow outputs (last layer: ctot = num_supports * 1): one thread per row keeps the H row in
// registers and produces all ctot chains, so H is read once.
template <int CTOT>
__global__ __launch_bounds__(256) void k_transform_narrow(const float* __restrict__ H, int ldh, int rows, int cin,
                                                          const float* __restrict__ W, float* __restrict__ Z,
                                                          int ldz) {
    for (int r = blockIdx.x * 256 + threadIdx.x; r < rows; r += gridDim.x * 256) {
        const float* h = H + (size_t)r * ldh;
        float acc[CTOT];
#pragma unroll
        for (int n = 0; n < CTOT; ++n) acc[n] = 0.f;
        for (int k = 0; k < cin; ++k) {
            const float hk = h[k];
#pragma unroll
            for (int n = 0; n < CTOT; ++n) acc[n] = fmaf(hk, W[k * CTOT + n], acc[n]);
        }
#pragma unroll
        for (int n = 0; n < CTOT; ++n) Z[(size_t)r * ldz + n] = acc[n];
    }
}

// The k chain carried in double and rounded once (the contract of layer index 1, DGCN_PRECISE in include/dgcn.h): one thread
// per output element.  Runs once per forward of the layer-by-layer path.
__global__ __launch_bounds__(256) void k_transform_f64acc(const float* __restrict__ H, int ldh, float h_const, int rows, int cin,
                                                          const float* __restrict__ W, int ctot, float* __restrict__ Z, int ldz) {
    const long total = (long)rows * ctot;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / ctot), n = (int)(i - (long)r * ctot);
        double acc = 0.0;
        if (H) {
            const float* h = H + (size_t)r * ldh;
            for (int k = 0; k < cin; ++k) acc = fma((double)h[k], (double)W[k * ctot + n], acc);
        } else {
            for (int k = 0; k < cin; ++k) acc = fma((double)h_const, (double)W[k * ctot + n], acc);
        }
        Z[(size_t)r * ldz + n] = (float)acc;
    }
}

// The hidden 32 -> (32 | 32) case of the same contract on the f64 matrix pipe: v_mfma_f64_16x16x4_f64 is exactly the
// ascending fma chain (tools/micro/mfma_f64.hip).  One wave = one 16-row tile x all 64 columns, eight chained MFMAs per
// 16-column block.  Operands swapped as in fused.hip's hidden_transform_f64 (A = weights, B = activations), and since the f64
// instruction returns row 4 * reg + (lane >> 4), lane r feeds weight column 4 * (r & 3) + (r >> 2) of the block: each lane
// ends with four CONSECUTIVE output features of one vertex = one 16-byte store.
using f64x4t = __attribute__((ext_vector_type(4))) double;
__global__ __launch_bounds__(256) void k_transform_mfma_f64_32x64(const float* __restrict__ H, int ldh, int rows,
                                                                  const float* __restrict__ W, float* __restrict__ Z, int ldz) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int rc = 4 * (r & 3) + (r >> 2);
    float b[8][4];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) b[s][ct] = W[(4 * s + kq) * 64 + ct * 16 + rc];
    const int tiles = (rows + 15) >> 4;
    for (int t = blockIdx.x * 4 + wave; t < tiles; t += gridDim.x * 4) {
        const int row = t * 16 + r;
        const int rl = row < rows ? row : rows - 1;  // (rows past the end feed the last row and store nothing)
        float av[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) av[s] = H[(size_t)rl * ldh + 4 * s + kq];
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
            f64x4t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                float b0 = b[s][2 * cp], b1 = b[s][2 * cp + 1], a0 = av[s];
                asm volatile("" : "+v"(b0), "+v"(b1), "+v"(a0));  // (keeps the conversions inside the loop: 64 fewer registers)
                const double ad = (double)a0;
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)b0, ad, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)b1, ad, acc1, 0, 0, 0);
            }
            if (row < rows) {
                float* dst = Z + (size_t)row * ldz + (2 * cp) * 16 + 4 * kq;
                *reinterpret_cast<float4*>(dst) = make_float4((float)acc0[0], (float)acc0[1], (float)acc0[2], (float)acc0[3]);
                *reinterpret_cast<float4*>(dst + 16) = make_float4((float)acc1[0], (float)acc1[1], (float)acc1[2], (float)acc1[3]);
            }
        }
    }
}

int transform_f64acc_dispatch(const float* H, int ldh, float h_const, int rows, int cin, const float* W, int ctot, float* Z,
                              int ldz, hipStream_t s) {
    if (rows <= 0) return DGCN_OK;
    TimedLaunch t("transform", s);
    if (H && cin == 32 && ctot == 64 && (ldz % 4 == 0) && ((uintptr_t)Z % 16 == 0)) {
        const int blocks = min(ceil_div(rows, 64), 256 * 8);
        DGCN_LAUNCH(t, k_transform_mfma_f64_32x64, dim3(blocks), dim3(256), 0, s, H, ldh, rows, W, Z, ldz);
        return check_launch("k_transform_mfma_f64_32x64");
    }
    const long total = (long)rows * ctot;
    const int blocks = (int)min((total + 255) / 256, (long)256 * 32);
    DGCN_LAUNCH(t, k_transform_f64acc, dim3(blocks), dim3(256), 0, s, H, ldh, h_const, rows, cin, W, ctot, Z, ldz);
    return check_launch("k_transform_f64acc");
}

int transform_dispatch(const float* H, int ldh, float h_const, int rows, int cin, const float* W, int ctot, float* Z,
                       int ldz, hipStream_t s) {
    if (rows <= 0) return DGCN_OK;
    TimedLaunch t("transform", s);
    const bool aligned = H && (ldh % 4 == 0) && ((uintptr_t)H % 16 == 0);
    const int mfma_blocks = min(ceil_div(rows, 128), 256 * 8);
#define DGCN_TF_CASE(CI, CO)                                                                                     \
    if (aligned && cin == CI && ctot == CO) {                                                                    \
        DGCN_LAUNCH(t, (k_transform_mfma<CI, CO>), dim3(mfma_blocks), dim3(256), 0, s, H, ldh, rows, W, Z, ldz); \
        return check_launch("k_transform_mfma");                                                                 \
    }
    DGCN_TF_CASE(32, 64)
    DGCN_TF_CASE(32, 32)
    DGCN_TF_CASE(32, 96)
    DGCN_TF_CASE(16, 32)
    DGCN_TF_CASE(16, 64)
    DGCN_TF_CASE(64, 128)
    DGCN_TF_CASE(64, 64)
    DGCN_TF_CASE(8, 32)
#undef DGCN_TF_CASE
    if (H && (ctot == 2 || ctot == 3)) {
        const dim3 grid(min(ceil_div(rows, 256), 4096));
        if (ctot == 2) DGCN_LAUNCH(t, (k_transform_narrow<2>), grid, dim3(256), 0, s, H, ldh, rows, cin, W, Z, ldz);
        else DGCN_LAUNCH(t, (k_transform_narrow<3>), grid, dim3(256), 0, s, H, ldh, rows, cin, W, Z, ldz);
        return check_launch("k_transform_narrow");
    }
    const long total = (long)rows * ctot;
    const int blocks = (int)min((total + 255) / 256, (long)256 * 16);
    DGCN_LAUNCH(t, k_transform_valu, dim3(blocks), dim3(256), 0, s, H, ldh, h_const, rows, cin, W, ctot, Z, ldz);
    return check_launch("k_transform_valu");
}

}  // namespace dgcn

using namespace dgcn;

extern "C" int dgcn_transform_batch(const float* H, int32_t ldh, float h_const, int32_t rows, int32_t cin,
                                    const float* W, int32_t ctot, float* Z, int32_t ldz, void* stream) {
    if (!W || !Z) return fail(DGCN_ERR_ARG, "dgcn_transform_batch: null argument");
    if (cin <= 0 || ctot <= 0 || ldz < ctot || (H && ldh < cin)) return fail(DGCN_ERR_ARG, "dgcn_transform_batch: bad sizes");
    return transform_dispatch(H, ldh, h_const, rows, cin, W, ctot, Z, ldz, (hipStream_t)stream);
}

extern "C" int dgcn_transform_f64acc_batch(const float* H, int32_t ldh, float h_const, int32_t rows, int32_t cin,
                                           const float* W, int32_t ctot, float* Z, int32_t ldz, void* stream) {
    if (!W || !Z) return fail(DGCN_ERR_ARG, "dgcn_transform_f64acc_batch: null argument");
    if (cin <= 0 || ctot <= 0 || ldz < ctot || (H && ldh < cin)) return fail(DGCN_ERR_ARG, "dgcn_transform_f64acc_batch: bad sizes");
    return transform_f64acc_dispatch(H, ldh, h_const, rows, cin, W, ctot, Z, ldz, (hipStream_t)stream);
}
