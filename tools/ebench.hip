// tools/ebench.hip -- developer microbenchmark for the encoder kernels (not part of the product).
#include "../augcl_amd/csrc/encoder_kernels.hip"
#include <algorithm>
#include <functional>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
using namespace augcl;
struct V { std::string name; double flop; std::function<void()> run; };
// single dependent accumulator chain, operands from LDS with 9-deep prefetch like k_conv32's loop
__global__ __launch_bounds__(256, 1) void k_mfma_chain1(float *out, int iters, float a0)
{
    __shared__ float sh[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) sh[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f;
    const float *base = sh + (threadIdx.x & 63);
    float xa[9], xb[9];
    for (int t = 0; t < 9; ++t) xa[t] = base[t * 64];
    for (int i = 0; i < iters; i += 2) {
        for (int t = 0; t < 9; ++t) xb[t] = base[t * 64 + ((i + 1) & 3) * 700];
        __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t < 9; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xa[t], acc, 0, 0, 0);
        for (int t = 0; t < 9; ++t) xa[t] = base[t * 64 + ((i + 2) & 3) * 700];
        __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t < 9; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xb[t], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the same operand pattern feeding TWO independent accumulator chains (two tiles at once, shared weights)
__global__ __launch_bounds__(512, 1) void k_mfma_chain2(float *out, int iters, float a0)
{
    __shared__ float sh[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sh[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float a = a0 + threadIdx.x * 1e-3f;
    const float *base = sh + (threadIdx.x & 63);
    float xa[9], xb[9];
    for (int t = 0; t < 9; ++t) xa[t] = base[t * 64];
    for (int i = 0; i < iters; i += 2) {
        for (int t = 0; t < 9; ++t) xb[t] = base[t * 64 + ((i + 1) & 3) * 700];
        __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t < 9; ++t) { if (t & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xa[t], acc1, 0, 0, 0); else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xa[t], acc0, 0, 0, 0); }
        for (int t = 0; t < 9; ++t) xa[t] = base[t * 64 + ((i + 2) & 3) * 700];
        __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t < 9; ++t) { if (t & 1) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xb[t], acc0, 0, 0, 0); else acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xb[t], acc1, 0, 0, 0); }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// one chain per wave, TWO waves per SIMD (512 threads per workgroup)
__global__ __launch_bounds__(512, 1) void k_mfma_chain1x2(float *out, int iters, float a0)
{
    __shared__ float sh[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) sh[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f;
    const float *base = sh + (threadIdx.x & 63);
    float xa[9], xb[9];
    for (int t = 0; t < 9; ++t) xa[t] = base[t * 64];
    for (int i = 0; i < iters; i += 2) {
        for (int t = 0; t < 9; ++t) xb[t] = base[t * 64 + ((i + 1) & 3) * 700];
        __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t < 9; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xa[t], acc, 0, 0, 0);
        for (int t = 0; t < 9; ++t) xa[t] = base[t * 64 + ((i + 2) & 3) * 700];
        __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t < 9; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xb[t], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
// 16x16x4 f32 MFMAs, one ds_read_b32 per MFMA, 4 independent accumulator tiles per wave; NW waves per workgroup
typedef float f32x4m __attribute__((ext_vector_type(4)));
template <int NW>
__global__ __launch_bounds__(NW * 64, 1) void k_mfma_16x4(float *out, int iters, float a0)
{
    __shared__ float sh[8192];
    for (int i = threadIdx.x; i < 8192; i += NW * 64) sh[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    f32x4m acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
    float w[9];
    for (int t = 0; t < 9; ++t) w[t] = a0 + threadIdx.x * 1e-3f + t;
    const int lane = threadIdx.x & 63;
    const float *base = sh + (lane & 15) + (lane >> 4) * 1040 + (threadIdx.x >> 6) * 16; // 4 k-groups 1040 floats apart (== 16 mod 64 banks)
    float xa[4][9], xb[4][9];
    for (int p = 0; p < 4; ++p) for (int t = 0; t < 9; ++t) xa[p][t] = base[p * 64 + (t / 3) * 88 + t % 3];
    for (int i = 0; i < iters; i += 2) {
        const float *b1 = base + ((i + 1) & 3) * 300;
        for (int p = 0; p < 4; ++p) for (int t = 0; t < 9; ++t) xb[p][t] = b1[p * 64 + (t / 3) * 88 + t % 3];
        __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t < 9; ++t) for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t], xa[p][t], acc[p], 0, 0, 0);
        const float *b2 = base + ((i + 2) & 3) * 300;
        for (int p = 0; p < 4; ++p) for (int t = 0; t < 9; ++t) xa[p][t] = b2[p * 64 + (t / 3) * 88 + t % 3];
        __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t < 9; ++t) for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t], xb[p][t], acc[p], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
    out[blockIdx.x * NW * 64 + threadIdx.x] = s;
}
// 9 accumulator tiles (k_wgrad32's shape): RUN consecutive MFMAs on the same accumulator before moving to the next one
template <int RUN>
__global__ __launch_bounds__(256, 1) void k_mfma_9acc(float *out, int iters, float a0)
{
    f32x16 acc[9];
    for (int t = 0; t < 9; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a[4], b[4];
    for (int u = 0; u < 4; ++u) { a[u] = a0 + threadIdx.x * 1e-3f + u; b[u] = 1.0f + threadIdx.x * 1e-4f + u; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int u = 0; u < RUN; ++u) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + t) & 3], acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < 9; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// k_wgrad32's pair loop in isolation: 9 accumulators, per pixel pair 9 MFMAs + (V >= 1) 8 LDS reads issued after the
// 2nd / 3rd MFMA + (V >= 2) a dozen address instructions after the 1st + (V >= 3) the select / bias-sum in front
template <int V>
__global__ __launch_bounds__(256, 1) void k_wg_pair(float *out, int pairs, float a0)
{
    __shared__ float sh[64 * 129 + 256];
    for (int i = threadIdx.x; i < 64 * 129 + 256; i += 256) sh[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    typedef float f2 __attribute__((ext_vector_type(2)));
    f32x16 acc[9];
    for (int t = 0; t < 9; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const float *pa = sh + l31 * 129 + h, *pz = sh + 64 * 129 + l31 + h;
    struct Ops { float z, s[3]; f2 d[3]; };
    Ops o0, o1;
    auto rd1 = [&](const float *a, const float *z, Ops &o) { o.z = *z; o.s[0] = a[0]; o.d[0] = f2{a[1], a[2]}; o.s[1] = a[37]; };
    auto rd2 = [&](const float *a, Ops &o) { o.d[1] = f2{a[38], a[39]}; o.s[2] = a[74]; o.d[2] = f2{a[75], a[76]}; };
    rd1(pa, pz, o0); rd2(pa, o0); o1 = o0;
    int xn = h, pn = 2 * (threadIdx.x >> 6) + h;
    float dbacc = 0.f;
    auto mf = [&](const Ops &o, float z, int t) { const float xv = t % 3 == 0 ? o.s[t / 3] : o.d[t / 3][t % 3 - 1]; acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(z, xv, acc[t], 0, 0, 0); };
    auto step = [&](const Ops &cur, Ops &nxt) {
        if (V >= 1) __builtin_amdgcn_s_waitcnt(0xC07F);
        float z = cur.z;
        if (V >= 3) { z = pn < pairs * 8 ? z : 0.0f; dbacc += z; }
        const float *na = pa, *nz = pz;
        mf(cur, z, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (V >= 2) { // the product's address walk
            const bool valid = pn < pairs * 8;
            nz = valid ? pz : sh + 64 * 129;
            na = valid ? pa : sh;
            pn += 8; xn += 8;
            const bool wrap = xn >= 35;
            pa += wrap ? 2 : 0; pa -= wrap ? 2 : 0; // keeps the walk's instructions, stays inside the array
            xn -= wrap ? 35 : 0;
        }
        __builtin_amdgcn_sched_barrier(0);
        mf(cur, z, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (V >= 1) rd1(na, nz, nxt);
        __builtin_amdgcn_sched_barrier(0);
        mf(cur, z, 2);
        __builtin_amdgcn_sched_barrier(0);
        if (V >= 1) rd2(na, nxt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 3; t < 9; ++t) mf(cur, z, t);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int i = 0; i < pairs; i += 2) { step(o0, o1); step(o1, o0); }
    float s = dbacc;
    for (int t = 0; t < 9; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// register-only f32 MFMA loop: the practical MFMA ceiling (clock under load) on this chip
__global__ __launch_bounds__(256, 1) void k_mfma_peak(float *out, int iters, float a0)
{
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, int ABL>
static void launch_conv(const float *in, const float *w, const float *bias, const float *mask, float *out, int B, int Hi, int Wi, int per_cu = 1)
{
    ConvArgs a{}; a.in = in; a.w = w; a.bias = bias; a.mask = mask; a.out = out; a.B = B; a.Hi = Hi; a.Wi = Wi; a.relu = 1;
    size_t lds;
    if (MODE == 0) { a.Ho = Hi - 2; a.Wo = Wi - 2; a.RT = pick_rt(a.Ho, 32 * Wi, 2, kBufLds, &a.S); const size_t bw = (size_t)32 * kConvPlane; lds = (std::max(2 * bw, bw + 32 * 292) + 32) * 4; }
    else { a.Ho = Hi + 2; a.Wo = Wi + 2; a.RT = pick_rt(a.Ho, 32 * (Wi + 2), 2, kBufLds - 32 * 8, &a.S); a.magic_wi = (uint32_t)(0x100000000ull / (unsigned)Wi) + 1u; const size_t bw = (size_t)32 * kConvPlane; lds = (std::max(2 * bw, bw + 32 * 292) + 32) * 4; }
    a.n_items = B * a.S;
    set_lds(k_conv32<MODE, ABL>, lds, "x");
    hipLaunchKernelGGL((k_conv32<MODE, ABL>), dim3(grid_for(a.n_items, per_cu)), dim3(512), lds, 0, a);
}
int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 256;
    const int H = argc > 2 ? atoi(argv[2]) : 41;
    const size_t nin = (size_t)B * 32 * H * H, nout = (size_t)B * 32 * (H + 2) * (H + 2);
    float *in, *out, *w, *bias, *mask, *ws, *dw, *db;
    CK(hipMalloc(&in, nout * 4)); CK(hipMalloc(&out, nout * 4)); CK(hipMalloc(&mask, nout * 4));
    CK(hipMalloc(&w, 9216 * 4)); CK(hipMalloc(&bias, 128)); CK(hipMalloc(&dw, 9216 * 4)); CK(hipMalloc(&db, 128));
    const size_t wsb = augcl_enc_wgrad_workspace_bytes(32); CK(hipMalloc(&ws, wsb + 8192)); // + room for the ABL = 4 stamps
    std::vector<float> h(nout); for (size_t i = 0; i < nout; ++i) h[i] = (float)((i * 2654435761u >> 9) & 1023) / 512.f - 0.7f;
    CK(hipMemcpy(in, h.data(), nout * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(mask, h.data(), nout * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, h.data(), 9216 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bias, h.data(), 128, hipMemcpyHostToDevice));
    const double f_fwd = 2.0 * 288 * 32 * B * (H - 2) * (H - 2), f_bwd = 2.0 * 288 * 32 * B * (double)(H + 2) * (H + 2);
    std::vector<V> vs;
    vs.push_back({"fwd product", f_fwd, [&] { launch_conv<0, 0>(in, w, bias, nullptr, out, B, H, H); }});
    vs.push_back({"fwd no staging", f_fwd, [&] { launch_conv<0, 1>(in, w, bias, nullptr, out, B, H, H); }});
    vs.push_back({"fwd 1/16 mfma", f_fwd, [&] { launch_conv<0, 2>(in, w, bias, nullptr, out, B, H, H); }});
    vs.push_back({"fwd no store", f_fwd, [&] { launch_conv<0, 3>(in, w, bias, nullptr, out, B, H, H); }});
    vs.push_back({"bwd-data product", f_bwd, [&] { launch_conv<1, 0>(in, w, nullptr, mask, out, B, H, H); }});
    vs.push_back({"bwd-data no staging", f_bwd, [&] { launch_conv<1, 1>(in, w, nullptr, mask, out, B, H, H); }});
    vs.push_back({"bwd-data 1/16 mfma", f_bwd, [&] { launch_conv<1, 2>(in, w, nullptr, mask, out, B, H, H); }});
    vs.push_back({"wgrad product", f_fwd, [&] { augcl_enc_conv32_backward_weights(in, out, dw, db, ws, wsb, B, H, H, 0); }});
    auto wg = [&](auto kern) {
        WgradArgs a{}; a.act = in; a.dz = out; a.partial = ws; a.B = B; a.Hi = H; a.Wi = H; a.Ho = H - 2; a.Wo = H - 2;
        int rt = (kBufLds / 4 - 64 * H - 64 - 32 - 600) / (32 * H + 32 * a.Wo); rt = std::min(rt, std::min(512 / H - 2, 256 / a.Wo)); if (rt > a.Ho) rt = a.Ho;
        a.S = (a.Ho + rt - 1) / rt; a.RT = (a.Ho + a.S - 1) / a.S; a.n_items = B * a.S;
        const int planeA = ((a.RT + 2) * H) | 1, planeZ = (a.RT * a.Wo + 1) | 1; size_t lds = 2 * (size_t)(32 * planeA + 32 * planeZ) * 4 + (size_t)2 * 2 * (((a.RT * a.Wo + 1) >> 1) + 8) * 8;
        set_lds(kern, lds, "x"); hipLaunchKernelGGL(kern, dim3(grid_for(a.n_items, 1)), dim3(256), lds, 0, a);
    };
    vs.push_back({"wgrad kernel only", f_fwd, [&] { wg(k_wgrad32<0>); }});
    vs.push_back({"wgrad no staging", f_fwd, [&] { wg(k_wgrad32<1>); }});
    vs.push_back({"wgrad 1/9 mfma", f_fwd, [&] { wg(k_wgrad32<2>); }});
    vs.push_back({"mfma f32 peak (regs only)", 2.0 * 32 * 32 * 2 * 16 * 4000.0 * 4 * 256, [&] { hipLaunchKernelGGL(k_mfma_peak, dim3(256), dim3(256), 0, 0, out, 4000, 0.5f); }});
    vs.push_back({"mfma 1 chain + LDS prefetch", 2.0 * 32 * 32 * 2 * 9 * 4000.0 * 4 * 256, [&] { hipLaunchKernelGGL(k_mfma_chain1, dim3(256), dim3(256), 0, 0, out, 4000, 0.5f); }});
    vs.push_back({"mfma 2 chains + LDS (1 wave)", 2.0 * 32 * 32 * 2 * 9 * 4000.0 * 4 * 256, [&] { hipLaunchKernelGGL(k_mfma_chain2, dim3(256), dim3(256), 0, 0, out, 4000, 0.5f); }});
    vs.push_back({"mfma 1 chain, 2 waves/SIMD", 2.0 * 32 * 32 * 2 * 9 * 4000.0 * 8 * 256, [&] { hipLaunchKernelGGL(k_mfma_chain1x2, dim3(256), dim3(512), 0, 0, out, 4000, 0.5f); }});
    vs.push_back({"mfma 16x16x4 + LDS, 4 waves", 2.0 * 16 * 16 * 4 * 36 * 1000.0 * 4 * 256, [&] { hipLaunchKernelGGL(k_mfma_16x4<4>, dim3(256), dim3(256), 0, 0, out, 1000, 0.5f); }});
    vs.push_back({"mfma 16x16x4 + LDS, 8 waves", 2.0 * 16 * 16 * 4 * 36 * 1000.0 * 8 * 256, [&] { hipLaunchKernelGGL(k_mfma_16x4<8>, dim3(256), dim3(512), 0, 0, out, 1000, 0.5f); }});
    vs.push_back({"mfma 9 accs round-robin", 2.0 * 32 * 32 * 2 * 9 * 4000.0 * 4 * 256, [&] { hipLaunchKernelGGL(k_mfma_9acc<1>, dim3(256), dim3(256), 0, 0, out, 4000, 0.5f); }});
    vs.push_back({"mfma 9 accs, runs of 2", 2.0 * 32 * 32 * 2 * 18 * 2000.0 * 4 * 256, [&] { hipLaunchKernelGGL(k_mfma_9acc<2>, dim3(256), dim3(256), 0, 0, out, 2000, 0.5f); }});
    vs.push_back({"mfma 9 accs, runs of 4", 2.0 * 32 * 32 * 2 * 36 * 1000.0 * 4 * 256, [&] { hipLaunchKernelGGL(k_mfma_9acc<4>, dim3(256), dim3(256), 0, 0, out, 1000, 0.5f); }});
    vs.push_back({"wgrad pair loop: MFMA only", 2.0 * 32 * 32 * 2 * 9 * 4000.0 * 4 * 256, [&] { hipLaunchKernelGGL(k_wg_pair<0>, dim3(256), dim3(256), 0, 0, out, 4000, 0.5f); }});
    vs.push_back({"wgrad pair loop: + LDS reads", 2.0 * 32 * 32 * 2 * 9 * 4000.0 * 4 * 256, [&] { hipLaunchKernelGGL(k_wg_pair<1>, dim3(256), dim3(256), 0, 0, out, 4000, 0.5f); }});
    vs.push_back({"wgrad pair loop: + addresses", 2.0 * 32 * 32 * 2 * 9 * 4000.0 * 4 * 256, [&] { hipLaunchKernelGGL(k_wg_pair<2>, dim3(256), dim3(256), 0, 0, out, 4000, 0.5f); }});
    vs.push_back({"wgrad pair loop: + select/sum", 2.0 * 32 * 32 * 2 * 9 * 4000.0 * 4 * 256, [&] { hipLaunchKernelGGL(k_wg_pair<3>, dim3(256), dim3(256), 0, 0, out, 4000, 0.5f); }});
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (argc > 3 && (atoi(argv[3]) == -3 || atoi(argv[3]) == -4)) { // -4: the same without the operand fetch (ABL 5) // per-strip shader-clock stamps of k_wgrad32<4, false, true>, workgroup 0 / wave 0
        WgradArgs a{}; a.act = in; a.dz = out; a.partial = ws; a.B = B; a.Hi = H; a.Wi = H; a.Ho = H - 2; a.Wo = H - 2;
        int rt = (kBufLds / 4 - 64 * H - 64 - 32 - 600) / (32 * H + 32 * a.Wo); rt = std::min(rt, std::min(512 / H - 2, 256 / a.Wo)); if (rt > a.Ho) rt = a.Ho;
        a.S = (a.Ho + rt - 1) / rt; a.RT = (a.Ho + a.S - 1) / a.S; a.n_items = B * a.S;
        const int planeA = ((a.RT + 2) * H) | 1, planeZ = (a.RT * a.Wo + 1) | 1; size_t lds = 2 * (size_t)(32 * planeA + 32 * planeZ) * 4 + (size_t)2 * 2 * (((a.RT * a.Wo + 1) >> 1) + 8) * 8;
        const int grid = grid_for(a.n_items, 1);
        const bool nofetch = atoi(argv[3]) == -4;
        if (nofetch) set_lds(k_wgrad32<5, false, true>, lds, "x"); else set_lds(k_wgrad32<4, false, true>, lds, "x");
        unsigned long long *dbg = reinterpret_cast<unsigned long long *>(ws + (size_t)grid * WG_SLAB + 8 * WG_SLAB);
        if ((size_t)(grid + 8) * WG_SLAB * 4 + 500 * 8 > wsb + 8192) { printf("workspace too small for stamps\n"); return 1; }
        CK(hipMemset(dbg, 0, 500 * 8));
        for (int rep = 0; rep < 3; ++rep) { if (nofetch) hipLaunchKernelGGL((k_wgrad32<5, false, true>), dim3(grid), dim3(512), lds, 0, a); else hipLaunchKernelGGL((k_wgrad32<4, false, true>), dim3(grid), dim3(512), lds, 0, a); }
        CK(hipDeviceSynchronize());
        printf("RT=%d S=%d npix=%d pairs/wave=%.1f items/WG=%.1f\n", a.RT, a.S, a.RT * a.Wo, a.RT * a.Wo / 8.0, (double)a.n_items / grid);
        std::vector<unsigned long long> hd(500); CK(hipMemcpy(hd.data(), dbg, 500 * 8, hipMemcpyDeviceToHost));
        unsigned long long prev = 0;
        for (int i = 0; i < 500 && hd[i]; ++i) {
            const unsigned long long t = hd[i] >> 4; const int id = (int)(hd[i] & 15);
            printf("%d:%llu ", id, prev ? t - prev : 0ull); prev = t;
            if (id == 5) printf("\n");
        }
        printf("\n");
        return 0;
    }
    if (argc > 3 && atoi(argv[3]) < 0) { // per-tile shader-clock stamps of workgroup 0's first compute wave (k_conv32<MODE, 4>)
        unsigned long long *dbg; CK(hipMalloc(&dbg, 2000 * 8)); CK(hipMemset(dbg, 0, 2000 * 8));
        const int mode = -atoi(argv[3]) - 1; // -1: forward, -2: backward-data
        auto go = [&](auto kern, int MODEV) {
            ConvArgs a{}; a.in = in; a.w = w; a.bias = MODEV ? nullptr : bias; a.mask = MODEV ? mask : nullptr; a.out = out; a.B = B; a.Hi = H; a.Wi = H; a.relu = 1;
            a.bias2 = reinterpret_cast<const float *>(dbg); a.dual = 0;
            size_t lds;
            if (MODEV == 0) { a.Ho = H - 2; a.Wo = H - 2; a.RT = pick_rt(a.Ho, 32 * H, 2, kBufLds, &a.S); const size_t bw = (size_t)32 * kConvPlane; lds = (std::max(2 * bw, bw + 32 * 292) + 32) * 4; }
            else { a.Ho = H + 2; a.Wo = H + 2; a.RT = pick_rt(a.Ho, 32 * (H + 2), 2, kBufLds - 32 * 8, &a.S); a.magic_wi = (uint32_t)(0x100000000ull / (unsigned)H) + 1u; const size_t bw = (size_t)32 * kConvPlane; lds = (std::max(2 * bw, bw + 32 * 292) + 32) * 4; }
            a.n_items = B * a.S;
            set_lds(kern, lds, "x");
            for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(grid_for(a.n_items, 1)), dim3(512), lds, 0, a);
            CK(hipDeviceSynchronize());
            printf("RT=%d S=%d npix=%d items/WG=%.1f\n", a.RT, a.S, a.RT * a.Wo, (double)a.n_items / grid_for(a.n_items, 1));
        };
        if (mode == 0) go(k_conv32<0, 4>, 0); else go(k_conv32<1, 4>, 1);
        std::vector<unsigned long long> hd(2000); CK(hipMemcpy(hd.data(), dbg, 2000 * 8, hipMemcpyDeviceToHost));
        unsigned long long prev = 0;
        for (int i = 0; i < 2000 && hd[i]; ++i) {
            const unsigned long long t = hd[i] >> 4; const int id = (int)(hd[i] & 15);
            printf("%d:%llu ", id, prev ? t - prev : 0ull); prev = t;
            if (id == 5) printf("\n");
        }
        printf("\n");
        return 0;
    }
    if (argc > 3) { // sustained run of ONE variant for argv[4] seconds (clock / power sampling from outside)
        const size_t v = (size_t)atoi(argv[3]) % vs.size();
        const double secs = argc > 4 ? atof(argv[4]) : 3.0;
        printf("sustained: %s\n", vs[v].name.c_str()); fflush(stdout);
        double total_us = 0; long n = 0;
        while (total_us < secs * 1e6) {
            CK(hipEventRecord(e0, 0)); for (int i = 0; i < 50; ++i) vs[v].run(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); total_us += ms * 1000.0; n += 50;
        }
        printf("%s: %.1f us per launch, %.1f TFLOP/s over %.1f s\n", vs[v].name.c_str(), total_us / n, vs[v].flop / (total_us / n) / 1e6, total_us / 1e6);
        return 0;
    }
    printf("B=%d H=%d\n%-26s %9s %9s\n", B, H, "variant", "med us", "TFLOP/s");
    std::vector<std::vector<double>> us(vs.size());
    for (int r = 0; r < 5; ++r) for (size_t v = 0; v < vs.size(); ++v) {
        for (int i = 0; i < 3; ++i) vs[v].run();
        CK(hipEventRecord(e0, 0)); for (int i = 0; i < 20; ++i) vs[v].run(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); us[v].push_back(ms * 1000.0 / 20);
    }
    for (size_t v = 0; v < vs.size(); ++v) { std::sort(us[v].begin(), us[v].end()); const double m = us[v][2]; printf("%-26s %9.1f %9.1f\n", vs[v].name.c_str(), m, vs[v].flop / m / 1e6); }
    return 0;
}
