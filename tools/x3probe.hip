// x3probe: numerics of f32-accurate products on the bf16 MFMA pipe (three-way operand split).
// D[32][32] = sum_k A[32][k] * B[k][32], K = 288, on (a) v_mfma_f32_32x32x2_f32, (b..e) v_mfma_f32_32x32x16_bf16 with the
// operands split x = hi + mid + lo (bf16 each) and 3 / 5 / 6 / 9 of the nine cross products; error against float64.
// build: hipcc -O3 --offload-arch=gfx950 tools/x3probe.hip -o tools/x3probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int K = 288, NT = 512; // NT independent 32x32 problems

__device__ __forceinline__ float bf_hi_rn(float x, float &rest)
{
    const float h = (float)(__bf16)x; // RNE
    rest = x - h;
    return h;
}
__device__ __forceinline__ float bf_hi_tr(float x, float &rest)
{
    const float h = __uint_as_float(__float_as_uint(x) & 0xffff0000u);
    rest = x - h;
    return h;
}

template <bool RN>
__device__ __forceinline__ void split3(float x, __bf16 &h, __bf16 &m, __bf16 &l)
{
    float r1, r2, r3;
    const float fh = RN ? bf_hi_rn(x, r1) : bf_hi_tr(x, r1);
    const float fm = RN ? bf_hi_rn(r1, r2) : bf_hi_tr(r1, r2);
    const float fl = RN ? bf_hi_rn(r2, r3) : bf_hi_tr(r2, r3);
    h = (__bf16)fh; m = (__bf16)fm; l = (__bf16)fl;
}

// mode 0: f32 MFMA; 1: 3 terms (hh, hm, mh); 2: 5 terms (+ hl, lh); 3: 6 terms (+ mm); 4: 9 terms; 5: 6 terms, truncation split
// 6: 6 terms, small products first; 7: 6 terms, two accumulators of opposite sign (even k-steps +A, odd k-steps -A), D = P - N:
// the bf16 MFMA's accumulation has a tiny bias toward -infinity (mean error column); it cancels in P - N
// f16 pair: x * 2^-e = hi + lo' * 2^-11 (hi = f16(x'), lo' = f16((x' - hi) * 2^11), both RNE): 22..24 significand bits as long as
// x' sits in f16's normal range; e is a per-tensor power of two (exact)
__device__ __forceinline__ void split_h(float x, _Float16 &h, _Float16 &l)
{
    h = (_Float16)x;
    l = (_Float16)((x - (float)h) * 2048.0f);
}

// modes 8..11 (f16 pairs, 3 products hh + (hl' + l'h) 2^-11, two accumulators): 8 = no scaling, 9 = operands scaled by per-tensor
// powers of two (max -> [2^14, 2^15)), 10 = 9 with +/- main accumulators, 11 = 9 with all four products (+ l'l' 2^-22)
__global__ void k_probe_h(const float *A, const float *B, float *D, int mode, float sa, float sb)
{
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
    const float *a = A + (size_t)blockIdx.x * 32 * K, *b = B + (size_t)blockIdx.x * K * 32;
    f32x16 acc, accn, accx, accl;
    for (int r = 0; r < 16; ++r) acc[r] = accn[r] = accx[r] = accl[r] = 0.0f;
    for (int k = 0; k < K; k += 16) {
        f16x8 ah, al, bh, bl;
        for (int j = 0; j < 8; ++j) {
            _Float16 x, y;
            split_h(a[l31 * K + k + 8 * h + j] * sa, x, y);
            ah[j] = x; al[j] = y;
            split_h(b[(k + 8 * h + j) * 32 + l31] * sb, x, y);
            bh[j] = x; bl[j] = y;
        }
        const bool neg = mode == 10 && ((k >> 4) & 1);
        if (mode == 11) accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, accl, 0, 0, 0);
        accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accx, 0, 0, 0);
        accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accx, 0, 0, 0);
        if (neg) {
            f16x8 an;
            for (int j = 0; j < 8; ++j) an[j] = -ah[j];
            accn = __builtin_amdgcn_mfma_f32_32x32x16_f16(an, bh, accn, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        }
    }
    float *d = D + (size_t)blockIdx.x * 1024;
    const float inv = 1.0f / (sa * sb);
    for (int r = 0; r < 16; ++r) {
        const float cross = accx[r] * (1.0f / 2048.0f) + accl[r] * (1.0f / (2048.0f * 2048.0f));
        d[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = ((acc[r] - accn[r]) + cross) * inv;
    }
}

__global__ void k_probe(const float *A, const float *B, float *D, int mode)
{
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
    const float *a = A + (size_t)blockIdx.x * 32 * K, *b = B + (size_t)blockIdx.x * K * 32;
    f32x16 acc, accn;
    for (int r = 0; r < 16; ++r) acc[r] = accn[r] = 0.0f;
    if (mode == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[l31 * K + k + h], b[(k + h) * 32 + l31], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            bf16x8 ah, am, al, bh, bm, bl;
            for (int j = 0; j < 8; ++j) {
                __bf16 x, y, z;
                if (mode == 5) split3<false>(a[l31 * K + k + 8 * h + j], x, y, z); else split3<true>(a[l31 * K + k + 8 * h + j], x, y, z);
                ah[j] = x; am[j] = y; al[j] = z;
                if (mode == 5) split3<false>(b[(k + 8 * h + j) * 32 + l31], x, y, z); else split3<true>(b[(k + 8 * h + j) * 32 + l31], x, y, z);
                bh[j] = x; bm[j] = y; bl[j] = z;
            }
#define MM(x, y) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0)
            if (mode == 7) {
                if ((k >> 4) & 1) {
                    for (int j = 0; j < 8; ++j) { ah[j] = -ah[j]; am[j] = -am[j]; al[j] = -al[j]; }
#define MN(x, y) accn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, accn, 0, 0, 0)
                    MN(am, bm); MN(al, bh); MN(ah, bl); MN(am, bh); MN(ah, bm); MN(ah, bh);
                } else {
                    MM(am, bm); MM(al, bh); MM(ah, bl); MM(am, bh); MM(ah, bm); MM(ah, bh);
                }
            } else if (mode == 6) {
                MM(am, bm); MM(al, bh); MM(ah, bl); MM(am, bh); MM(ah, bm); MM(ah, bh);
            } else {
                MM(ah, bh); MM(ah, bm); MM(am, bh);
                if (mode >= 2) { MM(ah, bl); MM(al, bh); }
                if (mode >= 3) MM(am, bm);
                if (mode == 4) { MM(am, bl); MM(al, bm); MM(al, bl); }
            }
        }
    }
    float *d = D + (size_t)blockIdx.x * 1024;
    for (int r = 0; r < 16; ++r) d[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = acc[r] - accn[r];
}

int main()
{
    std::vector<float> A((size_t)NT * 32 * K), B((size_t)NT * K * 32), D((size_t)NT * 1024);
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
    const char *names[] = {"f32 mfma 32x32x2", "bf16x3: 3 terms", "bf16x3: 5 terms", "bf16x3: 6 terms", "bf16x3: 9 terms",
                           "bf16x3: 6 terms, truncation split", "bf16x3: 6 terms, small first",
                           "bf16x3: 6 terms, +/- accumulators", "f16x2: 3 terms, unscaled", "f16x2: 3 terms, scaled",
                           "f16x2: 3 terms, scaled, +/- main", "f16x2: 4 terms, scaled"};
    for (int dist = 0; dist < 4; ++dist) {
        srand(1234 + dist);
        auto rnd = []() { return (rand() + 0.5) / (RAND_MAX + 1.0); };
        auto gauss = [&]() { return sqrt(-2.0 * log(rnd())) * cos(6.283185307179586 * rnd()); };
        for (auto &v : A) v = (float)(gauss() * 0.06);                                       // weights
        for (auto &v : B) {
            const double g = gauss();
            v = dist == 0 ? (float)g                                                          // signed activations
                : dist == 1 ? (float)(g > 0 ? g : 0)                                          // post-ReLU
                : dist == 2 ? (float)(g * 1e-6)                                               // small gradients
                            : (float)(g * exp(8.0 * gauss()));                                // wide dynamic range
        }
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        const char *dn[] = {"gauss", "relu(gauss)", "1e-6*gauss", "gauss*exp(8*gauss)"};
        printf("== B operand: %s\n", dn[dist]);
        auto pow2_scale = [](const std::vector<float> &v) { // power of two that brings max |v| into [2^14, 2^15)
            float m = 0;
            for (float x : v) m = fmaxf(m, fabsf(x));
            int e;
            frexpf(m, &e); // m = f * 2^e, f in [0.5, 1)
            return ldexpf(1.0f, 15 - e);
        };
        const float sa = pow2_scale(A), sb = pow2_scale(B);
        for (int mode = 0; mode < 12; ++mode) {
            if (mode >= 8) hipLaunchKernelGGL(k_probe_h, dim3(NT), dim3(64), 0, 0, dA, dB, dD, mode, mode == 8 ? 1.0f : sa, mode == 8 ? 1.0f : sb);
            else hipLaunchKernelGGL(k_probe, dim3(NT), dim3(64), 0, 0, dA, dB, dD, mode);
            hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
            double num = 0, den = 0, maxrel = 0, bias = 0, sabs = 0;
            for (int t = 0; t < NT; ++t)
                for (int i = 0; i < 32; ++i)
                    for (int j = 0; j < 32; ++j) {
                        double s = 0, sa = 0;
                        for (int k = 0; k < K; ++k) {
                            const double p = (double)A[((size_t)t * 32 + i) * K + k] * (double)B[((size_t)t * K + k) * 32 + j];
                            s += p; sa += fabs(p);
                        }
                        const double e = D[(size_t)t * 1024 + i * 32 + j] - s;
                        num += e * e; den += s * s; bias += e; sabs += sa;
                        if (fabs(e) / sa > maxrel) maxrel = fabs(e) / sa;
                    }
            printf("  %-36s rel-L2 %.3e   max |e|/sum|ab| %.3e   mean e / mean sum|ab| %+.3e\n", names[mode], sqrt(num / den), maxrel,
                   bias / sabs);
        }
    }
    return 0;
}
