// A native caller of libdgcn.so: HIP runtime + include/dgcn.h only (no Python, no torch).
// Builds two small conflict graphs and a 3-layer GCN from a fixed pseudo-random stream, runs the whole
// path with dgcn_solve_batch (mwis_gdpg_call.py:200-235 for a batch) and prints, per graph, the selected
// set, its weight and the number of greedy rounds.  tests/test_gpu_api.py runs the binary and compares the
// output with the CPU twin on the same inputs.
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/solve_batch.cpp -Ldistgcn_amd -ldgcn \
//         -Wl,-rpath,'$ORIGIN/../distgcn_amd' -o examples/solve_batch
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "dgcn.h"

#define HIP_OK(x)                                                                  \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            return 2;                                                              \
        }                                                                          \
    } while (0)

// the same stream of numbers as tests/test_gpu_api.py::_example_inputs (64-bit LCG, top 24 bits)
struct Lcg {
    uint64_t s;
    double next() {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        return (double)(s >> 40) / 16777216.0;
    }
};

template <typename T>
static T* to_device(const std::vector<T>& v) {
    T* d = nullptr;
    if (hipMalloc(&d, (v.size() ? v.size() : 1) * sizeof(T)) != hipSuccess) return nullptr;
    if (!v.empty() && hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

int main() {
    Lcg rng{12345};
    // ---- two graphs: G(40, 0.15) and G(25, 0.3), undirected, no self-loops, block-diagonal CSR
    const int sizes[2] = {40, 25};
    const double dens[2] = {0.15, 0.3};
    std::vector<int32_t> graph_ptr{0}, row_ptr{0}, col_idx;
    std::vector<double> weights;
    int max_nodes = 0, max_graph_edges = 0, max_degree = 0;
    for (int g = 0; g < 2; ++g) {
        const int n = sizes[g], n0 = graph_ptr.back();
        std::vector<std::vector<int>> adj(n);
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j)
                if (rng.next() < dens[g]) { adj[i].push_back(j); adj[j].push_back(i); }
        int edges = 0;
        for (int i = 0; i < n; ++i) {
            for (int j : adj[i]) col_idx.push_back(n0 + j);  // rows are sorted by construction
            edges += (int)adj[i].size();
            max_degree = adj[i].size() > (size_t)max_degree ? (int)adj[i].size() : max_degree;
            row_ptr.push_back((int32_t)col_idx.size());
        }
        for (int i = 0; i < n; ++i) weights.push_back(0.05 + rng.next());
        graph_ptr.push_back(n0 + n);
        max_nodes = n > max_nodes ? n : max_nodes;
        max_graph_edges = edges > max_graph_edges ? edges : max_graph_edges;
    }
    const int N = graph_ptr.back(), B = 2;
    // ---- GCN_DQN 1 -> 32 -> 32 -> 1, weights [in][W0 | W1] row-major, leaky / leaky / linear
    const int dims[4] = {1, 32, 32, 1};
    std::vector<std::vector<float>> W(3);
    for (int l = 0; l < 3; ++l) {
        W[l].resize((size_t)dims[l] * 2 * dims[l + 1]);
        const double lim = std::sqrt(6.0 / (dims[l] + dims[l + 1]));
        for (float& x : W[l]) x = (float)((2.0 * rng.next() - 1.0) * lim);
    }
    // ---- float64 d^-1/2 table, as numpy.power(d, -0.5) with inf -> 0 (gcn/utils.py:124-125)
    std::vector<double> dinv(max_degree + 2);
    for (size_t d = 0; d < dinv.size(); ++d) dinv[d] = d ? std::pow((double)d, -0.5) : 0.0;

    int32_t *d_gp = to_device(graph_ptr), *d_rp = to_device(row_ptr), *d_ci = to_device(col_idx);
    double *d_w = to_device(weights), *d_dinv = to_device(dinv);
    float* d_W[3];
    DgcnLayer layers[3];
    for (int l = 0; l < 3; ++l) {
        d_W[l] = to_device(W[l]);
        layers[l] = DgcnLayer{dims[l], dims[l + 1], d_W[l], nullptr, l == 2 ? DGCN_ACT_LINEAR : DGCN_ACT_LEAKY_RELU};
    }
    if (!d_gp || !d_rp || !d_ci || !d_w || !d_dinv || !d_W[0] || !d_W[1] || !d_W[2]) return 2;
    DgcnBatch batch{B, N, (int32_t)col_idx.size(), max_nodes, max_graph_edges, d_gp, d_rp, d_ci};
    DgcnModel model{3, 2, layers};
    if (!dgcn_solve_supported(&batch, &model)) { fprintf(stderr, "shape outside the fused kernel\n"); return 3; }

    float* d_scores; uint8_t* d_state; int32_t *d_rounds, *d_status; double* d_totals; void* d_ws;
    const size_t ws_bytes = dgcn_solve_workspace(&batch, &model);
    HIP_OK(hipMalloc(&d_scores, N * sizeof(float)));
    HIP_OK(hipMalloc(&d_state, N));
    HIP_OK(hipMalloc(&d_rounds, B * sizeof(int32_t)));
    HIP_OK(hipMalloc(&d_totals, B * sizeof(double)));
    HIP_OK(hipMalloc(&d_status, sizeof(int32_t)));
    HIP_OK(hipMalloc(&d_ws, ws_bytes));
    HIP_OK(hipMemset(d_status, 0, sizeof(int32_t)));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    const int rc = dgcn_solve_batch(&batch, &model, d_dinv, (int32_t)dinv.size(), /*X=*/nullptr, /*x_const=*/1.0f, d_w,
                                    /*predict_mwis=*/1, d_scores, d_state, d_rounds, d_totals, d_status, d_ws, ws_bytes,
                                    stream);
    if (rc != DGCN_OK) { fprintf(stderr, "dgcn_solve_batch: %s\n", dgcn_last_error()); return 4; }
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<uint8_t> state(N);
    std::vector<float> scores(N);
    int32_t rounds[2], status;
    double totals[2];
    HIP_OK(hipMemcpy(state.data(), d_state, N, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(scores.data(), d_scores, N * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(rounds, d_rounds, sizeof(rounds), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(totals, d_totals, sizeof(totals), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(&status, d_status, sizeof(status), hipMemcpyDeviceToHost));
    if (status) { fprintf(stderr, "device-side fault bits %d\n", status); return 5; }
    printf("dgcn %d\n", dgcn_version());
    for (int g = 0; g < B; ++g) {
        printf("graph %d rounds %d total %.17g set", g, rounds[g], totals[g]);
        for (int v = graph_ptr[g]; v < graph_ptr[g + 1]; ++v)
            if (state[v] == 1) printf(" %d", v - graph_ptr[g]);
        printf("\n");
    }
    printf("scores");
    for (int v = 0; v < N; ++v) printf(" %.9g", scores[v]);
    printf("\n");
    return 0;
}
