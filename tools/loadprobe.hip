// loadprobe: what HBM rate do the conv kernels' LOADER waves reach by themselves, and what limits it?
// One persistent workgroup per CU reads strips of an f32 NCHW activation tensor (B,32,H,W) exactly as k_conv32_x2's loaders do --
// unit = 4 consecutive pixels x 8 channels = 8 buffer_load_dwordx4 in 8 channel planes, strips of RT + 2 rows, `depth` strips in
// flight per wave -- and throws the data away (one xor per register so that the loads stay).  Variants: loader waves per
// workgroup (4 / 8 / 16), strips in flight (1 / 2 / 3 / 4), and the same bytes as ONE contiguous run per strip (NHWC-like) instead
// of 32 plane segments.  More waves = more independent 4-wave loader groups per workgroup, each on its own strips.  build: hipcc -O3 --offload-arch=gfx950 tools/loadprobe.hip -o tools/loadprobe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t vu4 __attribute__((ext_vector_type(4)));

struct Args {
    const float *in;
    uint32_t *sink;
    int B, H, W, RT, S, n_items, contiguous;
};

template <int WAVES, int DEPTH>
__global__ __launch_bounds__(WAVES * 64) void k_load(Args a)
{
    const int tid = threadIdx.x;
    const int lt = tid & 255;                // 4 waves form one loader group (as in the real kernels); WAVES / 4 groups per workgroup
    const int grp = tid >> 8, NGRP = WAVES / 4;
    const int cg = lt & 3, q0 = lt >> 2;     // channel group (8 channels), first quad
    const int NQ = 64;                       // quads per round of one group
    const unsigned cs4 = (unsigned)(a.H * a.W) * 4u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in), 0, -1, 0x00020000);
    const int G = gridDim.x * NGRP; // every group walks its own strips
    uint32_t acc = 0;
    vu4 r[DEPTH][2][8];
    auto issue = [&](int item, vu4 (&R)[2][8]) {
        const int b = item / a.S, s = item - b * a.S;
        const int y0 = s * a.RT;
        const int nrow = min(a.RT, a.H - 2 - y0) + 2;
        const int npx = nrow * a.W;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p0 = min(4 * (q0 + NQ * u), npx - 4);
            if (a.contiguous) { // the strip as one run of nrow * W * 32 floats: unit -> 8 consecutive 16-byte chunks per lane group
                const unsigned base = (unsigned)((b * 32 * a.H + y0 * 32) * a.W) * 4u;
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    R[u][c] = __builtin_bit_cast(vu4, __builtin_amdgcn_raw_buffer_load_b128(
                                                          rs, (unsigned)(((q0 + NQ * u) * 32 + c * 4 + cg) * 16) % (unsigned)(npx * 128), base, 0));
            } else {
                const unsigned base = (unsigned)((b * 32 * a.H + y0) * a.W) * 4u;
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    R[u][c] = __builtin_bit_cast(vu4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)cg * 8u * cs4 + 4u * (unsigned)p0,
                                                                                             base + (unsigned)c * cs4, 0));
            }
        }
    };
    auto consume = [&](vu4 (&R)[2][8]) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc ^= R[u][c][0] ^ R[u][c][1] ^ R[u][c][2] ^ R[u][c][3];
    };
    const int first = blockIdx.x * NGRP + grp;
    // DEPTH strips in flight: slot j % DEPTH
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (first + d * G < a.n_items) issue(first + d * G, r[d]);
    int j = 0;
    for (int item = first; item < a.n_items; item += DEPTH * G) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int it = item + d * G;
            if (it < a.n_items) {
                consume(r[d]);
                if (it + DEPTH * G < a.n_items) issue(it + DEPTH * G, r[d]);
                __syncthreads(); // (the strip barrier of the real kernel)
            }
        }
        ++j;
    }
    if (acc == 0x12345678u) a.sink[blockIdx.x * blockDim.x + tid] = acc;
}

template <int WAVES, int DEPTH>
static void run(const Args &a, const char *tag, double bytes)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_load<WAVES, DEPTH>), dim3(256), dim3(WAVES * 64), 0, 0, a);
    hipEventRecord(e0);
    const int reps = 20;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k_load<WAVES, DEPTH>), dim3(256), dim3(WAVES * 64), 0, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("  %-44s %7.1f us  %5.2f TB/s\n", tag, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
}

int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 512;
    for (int H : {41, 31, 23}) {
        const size_t n = (size_t)B * 32 * H * H;
        float *in; uint32_t *sink;
        hipMalloc(&in, n * 4 + 4096); hipMalloc(&sink, 256 * 1024 * 4);
        hipMemset(in, 0, n * 4 + 4096);
        for (int RT : {9, 4}) {
            Args a{};
            a.in = in; a.sink = sink; a.B = B; a.H = H; a.W = H; a.RT = RT;
            a.S = (H - 2 + RT - 1) / RT; a.n_items = B * a.S;
            double bytes = 0; // bytes the loads name (incl. halo rows; the last quads repeat)
            for (int s = 0; s < a.S; ++s) bytes += (double)(std::min(RT, H - 2 - s * RT) + 2) * H * 128.0;
            bytes *= B;
            printf("== H = W = %d, B = %d, strips of %d + 2 rows (%d per image), %.0f MB named per launch\n", H, B, RT, a.S, bytes / 1e6);
            // a 4-wave round covers 256 quads = 1024 pixels x 8 channels... two units per lane: 512 pixels x 32 channels per 4 waves
            if ((RT + 2) * H <= 512) {
                a.contiguous = 0;
                run<4, 1>(a, "4 waves, 1 strip in flight, 32 planes", bytes);
                run<4, 2>(a, "4 waves, 2 strips in flight, 32 planes", bytes);
                run<4, 3>(a, "4 waves, 3 strips in flight, 32 planes", bytes);
                a.contiguous = 1;
                run<4, 2>(a, "4 waves, 2 strips in flight, ONE run", bytes);
            }
            a.contiguous = 0;
            if ((RT + 2) * H <= 512) {
                run<8, 1>(a, "8 waves (2 groups), 1 strip each in flight", bytes);
                run<8, 2>(a, "8 waves (2 groups), 2 strips each in flight", bytes);
                run<16, 1>(a, "16 waves (4 groups), 1 strip each in flight", bytes);
                run<16, 2>(a, "16 waves (4 groups), 2 strips each in flight", bytes);
            }
        }
        hipFree(in); hipFree(sink);
    }
    return 0;
}
