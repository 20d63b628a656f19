/* Error-budget probe (analysis tool, not product, not oracle): the twin's forward with selectable
 * precision per phase, to see what each phase contributes to |score - float64 truth|.
 * mode bits per layer: 1 = aggregation sum (incl. Z0 + sum + bias) carried in double, rounded once;
 *                      2 = transform dot product carried in double, rounded once;
 *                      4 = aggregation as G=2 split chains ; 8 = transform as two half chains  */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
static float actf(float x, int a) { return a == 1 ? (x > 0 ? x : 0.2f * x) : x; }
int eb_forward(int n, const int32_t* rp, const int32_t* ci, const float* val, int L, const int32_t* dims,
               const float* const* W, const int32_t* acts, const int32_t* mode, float x_const, float* scores) {
    int maxd = 0;
    for (int l = 0; l <= L; ++l) if (dims[l] > maxd) maxd = dims[l];
    float* Z = malloc((size_t)n * 2 * maxd * 4 + 16);
    float* H = malloc((size_t)n * maxd * 4 + 16);
    float* H2 = malloc((size_t)n * maxd * 4 + 16);
    for (int l = 0; l < L; ++l) {
        int cin = dims[l], cout = dims[l + 1], ct = 2 * cout, m = mode[l];
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < ct; ++c) {
                if (m & 2) {
                    double a = 0;
                    for (int k = 0; k < cin; ++k) a = fma((double)(l ? H[(size_t)r * cin + k] : x_const), (double)W[l][k * ct + c], a);
                    Z[(size_t)r * ct + c] = (float)a;
                } else if (m & 8) {
                    float a = 0, b = 0;
                    for (int k = 0; k < cin / 2; ++k) a = fmaf(l ? H[(size_t)r * cin + k] : x_const, W[l][k * ct + c], a);
                    for (int k = cin / 2; k < cin; ++k) b = fmaf(l ? H[(size_t)r * cin + k] : x_const, W[l][k * ct + c], b);
                    Z[(size_t)r * ct + c] = a + b;
                } else {
                    float a = 0;
                    for (int k = 0; k < cin; ++k) a = fmaf(l ? H[(size_t)r * cin + k] : x_const, W[l][k * ct + c], a);
                    Z[(size_t)r * ct + c] = a;
                }
            }
        float* out = l == L - 1 ? scores : H2;
        for (int v = 0; v < n; ++v)
            for (int c = 0; c < cout; ++c) {
                float o;
                if (m & 1) {
                    double a = 0;
                    for (int j = rp[v]; j < rp[v + 1]; ++j) a = fma((double)val[j], (double)Z[(size_t)ci[j] * ct + cout + c], a);
                    o = (float)((double)Z[(size_t)v * ct + c] + a);
                } else if (m & 4) {
                    float a = 0, b = 0;
                    for (int j = rp[v]; j < rp[v + 1]; ++j) {
                        if ((j - rp[v]) & 1) b = fmaf(val[j], Z[(size_t)ci[j] * ct + cout + c], b);
                        else a = fmaf(val[j], Z[(size_t)ci[j] * ct + cout + c], a);
                    }
                    o = Z[(size_t)v * ct + c] + (a + b);
                } else {
                    float a = 0;
                    for (int j = rp[v]; j < rp[v + 1]; ++j) a = fmaf(val[j], Z[(size_t)ci[j] * ct + cout + c], a);
                    o = Z[(size_t)v * ct + c] + a;
                }
                out[(size_t)v * cout + c] = actf(o, acts[l]);
            }
        float* t = H; H = H2; H2 = t;
    }
    free(Z); free(H); free(H2);
    return 0;
}
