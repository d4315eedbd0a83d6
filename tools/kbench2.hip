// tools/kbench2.hip -- A/B of random_conv kernel variants (strip height / workgroup size) against the round-1 kernel (kept here, tools only).
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -Iinclude tools/kbench2.hip augcl_amd/csrc/capi_core.cpp -x hip -o tools/kbench2
#include "../augcl_amd/csrc/aug_kernels.hip"
#include <algorithm>
#include <functional>
#include <string>
#include <vector>
namespace augcl { namespace r1 {
// =========================================================================================
// random_conv (+ optional fused random_shift front end)
//   workgroup = one R-row strip of one RGB frame; the (R+2) x 3 x W input rows are staged
//   ONCE into LDS already divided by 255 (the reference's x/255.0, one IEEE division per
//   input element instead of 27), 3x3 windows are then read from LDS: one ds_read_b128 +
//   two edge ds_read_b32 per (channel,row), 81 FMAs per output pixel triple against
//   LDS-broadcast weights, sigmoid, *255, one float4 store per output channel.
// =========================================================================================
template <int H, int W, int R>
struct ConvSmem {
    float tile[3][R + 2][W]; // x/255, rows y0-1 .. y0+R (row-clamped = replicate pad)
    f32x2 w[84];             // 81 weights [o][i][ky][kx] (+3 pad), each stored twice: one aligned
                             // ds_read_b64 yields the (w,w) operand of v_pk_fma_f32 with no v_mov
};

// x / 255.0f, correctly rounded, in 3 VALU ops instead of the ~10 of the IEEE division
// sequence: q = x*r; q' = fma(fma(-q,255,x), r, q) with r = RN(1/255).  Verified EXHAUSTIVELY
// (all 2^31 non-negative finite f32, denormals included; tests/div255_check.c, run by tests/test_div255.py) to equal the IEEE
// quotient bit for bit; +-inf would turn into NaN in the residual, so it is passed through.
__device__ __forceinline__ float div255(float x)
{
    const float r = 1.0f / 255.0f;
    const float q = x * r;
    const float e = __builtin_fmaf(-q, 255.0f, x);
    const float y = __builtin_fmaf(e, r, q);
    return __builtin_isinf(x) ? x : y;
}

__device__ __forceinline__ float sigmoid255(float a)
{
    // torch.sigmoid = 1/(1+exp(-a)).  exp(-a) = 2^(-a*log2 e) on the v_exp_f32 unit: the rounding of
    // the product perturbs the result by <= |a| * 2^-24 relative, i.e. s(1-s)|a| * 6e-8 <= 1.4e-8 on
    // the sigmoid (x255: < 4e-6), and v_exp_f32 / v_rcp_f32 are 1 ulp: all far inside the 1e-5 budget
    // and an order of magnitude below the conv's own summation-order noise.
    const float e = __builtin_amdgcn_exp2f(a * -1.4426950408889634f);
    return __builtin_amdgcn_rcpf(1.0f + e) * 255.0f;
}

template <int H, int W, int R>
__device__ __forceinline__ void conv_stage_weights(ConvSmem<H, W, R> &sm, const float *__restrict__ weights,
                                                   uint64_t seed, uint64_t offset, int gsample, int b)
{
    const int tid = threadIdx.x;
    if (weights != nullptr) {
        if (tid < 81) {
            const float wv = weights[(size_t)b * 81 + tid];
            sm.w[tid] = f32x2{wv, wv};
        }
    } else if (tid < 21) {
        float w4[4];
        philox_conv_weights4(seed, offset, (uint32_t)gsample, tid, w4);
#pragma unroll
        for (int k = 0; k < 4; ++k) sm.w[4 * tid + k] = f32x2{w4[k], w4[k]};
    }
}

// one thread = 4 consecutive pixels of one row, all 3 output channels.  FMAs are issued as
// v_pk_fma_f32 on the pixel pairs (0,1) and (2,3): for tap kx the operand pairs are
// (v[kx],v[kx+1]) and (v[kx+2],v[kx+3]) of the 6-wide window v = [left, m0..m3, right]; the even
// pairs come from ds_read2_b32, the odd pairs are the two halves of one aligned ds_read_b128.
// Accumulation order per output: i, ky, kx ascending (same as oracle/aug_oracle.c).
template <int H, int W, int R, int ABL = 0> // ABL: developer ablation switch (tools/kbench), 0 = product
__device__ __forceinline__ void conv_compute_store(const ConvSmem<H, W, R> &sm, float *__restrict__ out_frame,
                                                   int y0)
{
    constexpr int W4 = W / 4;
    const int tid = threadIdx.x;
    if (tid >= R * W4) return;
    const int ry = tid / W4;
    const int x4 = (tid - ry * W4) * 4;
    const int xl = x4 > 0 ? x4 - 1 : 0;
    const int xr = x4 + 4 < W ? x4 + 4 : W - 1;

    f32x2 acc[3][2];
#pragma unroll
    for (int o = 0; o < 3; ++o) acc[o][0] = acc[o][1] = f32x2{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < (ABL == 2 ? 1 : 3); ++i) {
        f32x2 pe[3][3], po[3][2]; // even pairs (v0v1,v2v3,v4v5), odd pairs (v1v2,v3v4)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float *row = sm.tile[i][ry + ky];
            const float4 m = *reinterpret_cast<const float4 *>(row + x4);
            po[ky][0] = f32x2{m.x, m.y};
            po[ky][1] = f32x2{m.z, m.w};
            pe[ky][0] = f32x2{row[xl], m.x};
            pe[ky][1] = f32x2{m.y, m.z};
            pe[ky][2] = f32x2{m.w, row[xr]};
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    const f32x2 w2 = sm.w[((o * 3 + i) * 3 + ky) * 3 + kx];
                    const f32x2 in01 = (kx == 1) ? po[ky][0] : pe[ky][kx / 2];
                    const f32x2 in23 = (kx == 1) ? po[ky][1] : pe[ky][kx / 2 + 1];
                    acc[o][0] = __builtin_elementwise_fma(w2, in01, acc[o][0]);
                    acc[o][1] = __builtin_elementwise_fma(w2, in23, acc[o][1]);
                }
    }
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        float4 res = make_float4(sigmoid255(acc[o][0].x), sigmoid255(acc[o][0].y), sigmoid255(acc[o][1].x),
                                 sigmoid255(acc[o][1].y));
        if (ABL == 1) res = make_float4(acc[o][0].x, acc[o][0].y, acc[o][1].x, acc[o][1].y);
        store_nt(out_frame + ((size_t)o * H + (y0 + ry)) * W + x4, res);
    }
}

// =========================================================================================
// random_conv / fused random_shift -> (out_shift) -> random_conv -> (out_aug)
//   SHIFT = false: plain random_conv (in -> out_aug);  TIn = float | uint8_t
//   Every thread first issues ALL of its strip loads (NI quads in flight), then converts,
//   optionally stores the shifted rows, divides by 255 and writes the LDS tile.
// =========================================================================================
template <int H, int W, int R, int NT, typename TIn, bool SHIFT, int ABL = 0>
__global__ __launch_bounds__(NT) void k_conv_r1(const TIn *__restrict__ in, float *__restrict__ out_shift,
                                             float *__restrict__ out_aug, int n_frames, int frames, int pad,
                                             const int32_t *__restrict__ oy, const int32_t *__restrict__ ox,
                                             const float *__restrict__ weights, uint64_t seed, uint64_t offset,
                                             int sample0)
{
    constexpr int NS = H / R;
    constexpr int W4 = W / 4;
    constexpr int NQ = 3 * (R + 2) * W4;
    constexpr int NI = (NQ + NT - 1) / NT;
    static_assert(R * W4 <= NT, "one compute task per thread");
    __shared__ __attribute__((aligned(16))) ConvSmem<H, W, R> sm;
    // XCD-aware item map: workgroups g, g+8, g+16, ... share an XCD (and its L2), so the NS strips of
    // one frame are given to consecutive workgroups of ONE residue class: their halo rows are then
    // L2 hits instead of fabric re-reads (PMC: FETCH 1.19x -> ~1.0x of the algorithmic bytes).
    // Pure speed heuristic: any placement is correct.  Frames beyond a multiple of 8 map plainly.
    int strip, fidx;
    {
        const int nfr8 = (n_frames / 8) * 8;
        const int bid = blockIdx.x;
        if (bid < nfr8 * NS) {
            const int g = bid & 7, q = bid >> 3;
            fidx = (q / NS) * 8 + g;
            strip = q - (q / NS) * NS;
        } else {
            strip = bid % NS;
            fidx = bid / NS;
        }
    }
    const int b = fidx / frames; // fidx = b*frames + f
    const int y0 = strip * R;
    int dy = 0, dx = 0;
    if constexpr (SHIFT) sample_shift(oy, ox, seed, offset, sample0 + b, b, pad, dy, dx);
    if (out_aug != nullptr) conv_stage_weights<H, W, R>(sm, weights, seed, offset, sample0 + b, b);
    const TIn *__restrict__ src = in + (size_t)fidx * 3 * H * W;

    Quad<W, TIn> raw[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int j = threadIdx.x + k * NT;
        if (j < NQ) {
            const int ch = j / ((R + 2) * W4);
            const int rem = j - ch * ((R + 2) * W4);
            const int rr = rem / W4;
            const int x4 = (rem - rr * W4) * 4;
            const int yc = clampi(y0 - 1 + rr, 0, H - 1); // conv's replicate pad (on the shifted image)
            const int sy = clampi(yc + dy, 0, H - 1);     // shift's replicate pad on the source
            raw[k].load(src + ((size_t)ch * H + sy) * W, clampi(x4 + dx, 0, W - 4));
        }
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int j = threadIdx.x + k * NT;
        if (j < NQ) {
            const int ch = j / ((R + 2) * W4);
            const int rem = j - ch * ((R + 2) * W4);
            const int rr = rem / W4;
            const int x4 = (rem - rr * W4) * 4;
            const int ys = y0 - 1 + rr;
            float4 q = raw[k].get(SHIFT ? x4 + dx - clampi(x4 + dx, 0, W - 4) : 0);
            if (SHIFT && out_shift != nullptr && ys >= y0 && ys < y0 + R)
                store_nt(out_shift + ((size_t)(fidx * 3 + ch) * H + ys) * W + x4, q);
            q.x = div255(q.x);
            q.y = div255(q.y);
            q.z = div255(q.z);
            q.w = div255(q.w);
            *reinterpret_cast<float4 *>(&sm.tile[ch][rr][x4]) = q;
        }
    }
    if (out_aug == nullptr) return;
    __syncthreads();
    if (ABL == 3) { // staging only: copy the strip's rows back out of LDS
        const int tid = threadIdx.x;
        if (tid < R * W4) {
            const int ry = tid / W4, x4 = (tid - ry * W4) * 4;
            for (int o = 0; o < 3; ++o)
                *reinterpret_cast<float4 *>(out_aug + ((size_t)(fidx * 3 + o) * H + y0 + ry) * W + x4) =
                    *reinterpret_cast<const float4 *>(&sm.tile[o][ry + 1][x4]);
        }
        return;
    }
    conv_compute_store<H, W, R, ABL>(sm, out_aug + (size_t)fidx * 3 * H * W, y0);
}


} }
using namespace augcl;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
struct Variant { std::string name; double bytes; std::function<void(int)> launch; };
template <int R, int NT, typename TIn, bool SHIFT, int EXP = 0>
static void launch_new(const TIn *in, float *o_s, float *o_a, int B, const int32_t *oy, const int32_t *ox, const float *w)
{
    hipLaunchKernelGGL((k_conv<84, 84, R, NT, TIn, SHIFT, false, EXP>), dim3(B * 3 * (84 / R)), dim3(NT), 0, 0, in, o_s, o_a, B * 3, 3, 4, oy, ox, w, 0, 0, 0,
                       (const int32_t *)nullptr, (const int64_t *)nullptr, 0);
}
int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 256;
    const int RING = 8, RING8 = 40, ROUNDS = 5, ITERS = 100;
    const size_t n = (size_t)B * 9 * 84 * 84;
    std::vector<float *> fin(RING), fout(RING), fout2(2);
    std::vector<uint8_t *> uin(RING8);
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u >> 13) & 255);
    std::vector<uint8_t> hu(n);
    for (size_t i = 0; i < n; ++i) hu[i] = (uint8_t)h[i];
    for (int k = 0; k < RING; ++k) { CK(hipMalloc(&fin[k], n * 4)); CK(hipMalloc(&fout[k], n * 4)); CK(hipMemcpy(fin[k], h.data(), n * 4, hipMemcpyHostToDevice)); }
    for (int k = 0; k < RING8; ++k) { CK(hipMalloc(&uin[k], n)); CK(hipMemcpy(uin[k], hu.data(), n, hipMemcpyHostToDevice)); }
    for (int k = 0; k < 2; ++k) CK(hipMalloc(&fout2[k], n * 4));
    std::vector<int32_t> hoy(B), hox(B);
    for (int b = 0; b < B; ++b) { hoy[b] = (b * 7 + 3) % 9; hox[b] = (b * 5 + 1) % 9; }
    int32_t *oy, *ox; CK(hipMalloc(&oy, B * 4)); CK(hipMalloc(&ox, B * 4));
    CK(hipMemcpy(oy, hoy.data(), B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ox, hox.data(), B * 4, hipMemcpyHostToDevice));
    std::vector<float> hw(B * 81);
    for (int i = 0; i < B * 81; ++i) hw[i] = (float)((i * 40503u >> 7) & 1023) / 512.f - 1.f;
    float *w; CK(hipMalloc(&w, B * 81 * 4)); CK(hipMemcpy(w, hw.data(), B * 81 * 4, hipMemcpyHostToDevice));
    const double f2f = 2.0 * n * 4, u2 = (double)n * 9;
    std::vector<Variant> vs;
    vs.push_back({"r1 conv f32", f2f, [&](int k) { hipLaunchKernelGGL((r1::k_conv_r1<84, 84, 12, 256, float, false>), dim3(B * 3 * 7), dim3(256), 0, 0, fin[k % RING], (float *)nullptr, fout[k % RING], B * 3, 3, 0, (const int32_t *)nullptr, (const int32_t *)nullptr, w, 0, 0, 0); }});
    vs.push_back({"r1 shift_conv u8", u2, [&](int k) { hipLaunchKernelGGL((r1::k_conv_r1<84, 84, 12, 256, uint8_t, true>), dim3(B * 3 * 7), dim3(256), 0, 0, uin[k % RING8], fout[k % RING], fout2[k & 1], B * 3, 3, 4, oy, ox, w, 0, 0, 0); }});
#define NEWV(R, NT) \
    vs.push_back({"conv f32 R=" #R " NT=" #NT, f2f, [&](int k) { launch_new<R, NT, float, false>(fin[k % RING], nullptr, fout[k % RING], B, nullptr, nullptr, w); }}); \
    vs.push_back({"shift_conv u8 R=" #R " NT=" #NT, u2, [&](int k) { launch_new<R, NT, uint8_t, true>(uin[k % RING8], fout[k % RING], fout2[k & 1], B, oy, ox, w); }});
    NEWV(12, 256)
#define EXPV(E) \
    vs.push_back({"conv f32 EXP=" #E, f2f, [&](int k) { launch_new<12, 256, float, false, E>(fin[k % RING], nullptr, fout[k % RING], B, nullptr, nullptr, w); }}); \
    vs.push_back({"shift_conv u8 EXP=" #E, u2, [&](int k) { launch_new<12, 256, uint8_t, true, E>(uin[k % RING8], fout[k % RING], fout2[k & 1], B, oy, ox, w); }});
    EXPV(0) EXPV(2) EXPV(3) EXPV(6) EXPV(7)
    vs.push_back({"conv f32 R=12 NT=256 again", f2f, [&](int k) { launch_new<12, 256, float, false>(fin[k % RING], nullptr, fout[k % RING], B, nullptr, nullptr, w); }});
    vs.push_back({"conv f32 philox R=12", f2f, [&](int k) { launch_new<12, 256, float, false>(fin[k % RING], nullptr, fout[k % RING], B, nullptr, nullptr, nullptr); }});
    {
        static JitterParams jp; jp.order[0] = 1; jp.order[1] = 2; jp.order[2] = 3; jp.order[3] = 0;
        jp.c_r = 1.2f; jp.c_inv = -0.2f; jp.s_r = 0.8f; jp.s_inv = 0.2f; jp.hue = 0.1f; jp.bound = 1.0f; jp.scale = 255.0f;
        const int npx = 84 * 84;
        vs.push_back({"jitter product", f2f, [&](int k) { static int32_t ord[4] = {1, 2, 3, 0}; augcl_color_jitter_f32(fout[k % RING], B, 3, 84, 84, nullptr, 0, ord, 1.2, 0.8, 0.1, 1.0f, 255.0f, 0); }});
        vs.push_back({"jitter reg<1024,2,4>", f2f, [&, npx](int k) { hipLaunchKernelGGL((k_color_jitter_reg<1024, 2, 4>), dim3(B * 3), dim3(1024), 0, 0, fout[k % RING], npx, (const int32_t *)nullptr, 0, jp); }});
        vs.push_back({"jitter reg<896,2,7>", f2f, [&, npx](int k) { hipLaunchKernelGGL((k_color_jitter_reg<896, 2, 7, true>), dim3(B * 3), dim3(896), 0, 0, fout[k % RING], npx, (const int32_t *)nullptr, 0, jp); }});
        vs.push_back({"jitter reg<512,4,4>", f2f, [&, npx](int k) { hipLaunchKernelGGL((k_color_jitter_reg<512, 4, 4>), dim3(B * 3), dim3(512), 0, 0, fout[k % RING], npx, (const int32_t *)nullptr, 0, jp); }});
        vs.push_back({"jitter reg<256,7,2>", f2f, [&, npx](int k) { hipLaunchKernelGGL((k_color_jitter_reg<256, 7, 2>), dim3(B * 3), dim3(256), 0, 0, fout[k % RING], npx, (const int32_t *)nullptr, 0, jp); }});
    }
    vs.push_back({"product conv_f32 philox", f2f, [&](int k) { augcl_random_conv_f32(fin[k % RING], fout[k % RING], B, 3, 84, 84, nullptr, 1, 2, 0, 0); }});
    vs.push_back({"product shift_conv_u8 philox", u2, [&](int k) { augcl_shift_conv_u8_f32(uin[k % RING8], fout[k % RING], fout2[k & 1], B, 3, 84, 84, 4, nullptr, nullptr, nullptr, 1, 2, 0, 0); }});
    vs.push_back({"product conv_f32", f2f, [&](int k) { augcl_random_conv_f32(fin[k % RING], fout[k % RING], B, 3, 84, 84, w, 0, 0, 0, 0); }});
    vs.push_back({"product shift_conv_u8", u2, [&](int k) { augcl_shift_conv_u8_f32(uin[k % RING8], fout[k % RING], fout2[k & 1], B, 3, 84, 84, 4, oy, ox, w, 0, 0, 0, 0); }});
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<std::vector<double>> us(vs.size());
    for (int r = 0; r < ROUNDS; ++r)
        for (size_t v = 0; v < vs.size(); ++v) {
            for (int i = 0; i < 10; ++i) vs[v].launch(i);
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < ITERS; ++i) vs[v].launch(i);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            us[v].push_back(ms * 1000.0 / ITERS);
        }
    printf("B=%d\n%-36s %9s %9s %9s %8s\n", B, "variant", "min us", "med us", "GB/s(med)", "Mobs/s");
    for (size_t v = 0; v < vs.size(); ++v) {
        std::sort(us[v].begin(), us[v].end());
        const double med = us[v][us[v].size() / 2];
        printf("%-36s %9.2f %9.2f %9.0f %8.2f\n", vs[v].name.c_str(), us[v][0], med, vs[v].bytes / med / 1e3, B / med);
    }
    return 0;
}
