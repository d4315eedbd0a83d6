// replay_kernels.hip -- device-resident replay sampling.
//
// The reference keeps observations as Python lists of LazyFrames and, per sample(), walks B indices on the host,
// np.concatenate()s 2*B frame stacks and pushes 2 x B x 63.5 KB through PCIe (src/utils.py:206-214, 234-240,
// 283-322).  Here every RGB frame lives ONCE in a uint8 ring in HBM and a batch is materialised by a gather:
// out[b, 3k:3k+3] = frames[idx[b][k]].  One dword (4 pixels) per thread, float4 / dword stores: pure HBM copy
// work -- (1 + 4) bytes per pixel for the f32 batch the agents consume, 2 for a uint8 batch.
#include "augcl_common.h"

namespace augcl {

template <typename TOut>
__global__ __launch_bounds__(256) void k_replay_gather(const uint8_t *__restrict__ frames,
                                                       const int32_t *__restrict__ slots /* [B][stack] */,
                                                       TOut *__restrict__ out /* [B][stack*fq*4] */, int stack,
                                                       int fq /* dwords per RGB frame */, size_t total)
{
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
        const size_t fi = q / fq; // = b*stack + k
        const int off = (int)(q - fi * fq);
        const int slot = slots[fi];
        const uint32_t v = reinterpret_cast<const uint32_t *>(frames)[(size_t)slot * fq + off];
        if constexpr (sizeof(TOut) == 1)
            reinterpret_cast<uint32_t *>(out)[q] = v;
        else
            store_nt(reinterpret_cast<float *>(out) + 4 * q,
                     make_float4((float)(v & 0xffu), (float)((v >> 8) & 0xffu), (float)((v >> 16) & 0xffu), (float)(v >> 24)));
    }
}

// The whole ReplayBuffer.sample() in one launch: transition indices idx[B] pick rows of the device-resident tables
// (slot_tab[cap][2*stack] = ring slots of the obs / next_obs frame stacks, scal[cap][S] = action | reward | not_done)
// and both stacked batches plus the scalars are written.  Work item q < 2*B*stack*fq: one dword (4 pixels) in, one float4 out
// per lane: both sides of a wave's access are contiguous (256 B in, 1 KB out).  (Round 3 tried 16 pixels per lane -- one
// dwordx4 in, four float4 out: the four stores of a lane are then 64 B apart ACROSS lanes, every store instruction touches 64
// partial lines, WRITE_SIZE doubled and the launch took 111 instead of 28 us: profiles/r03_bench_kernel_stats.csv history.)
template <typename TOut>
__global__ __launch_bounds__(256) void k_replay_sample(const uint8_t *__restrict__ frames,
                                                       const int32_t *__restrict__ slot_tab,
                                                       const float *__restrict__ scal, const int64_t *__restrict__ idx,
                                                       TOut *__restrict__ obs, TOut *__restrict__ next_obs,
                                                       float *__restrict__ scal_out, int B, int stack, int fq, int S)
{
    // blockIdx.y = one frame of one stack: (which tensor, sample b, frame k) -- the table lookups are per workgroup and
    // on the scalar unit; blockIdx.x walks the frame's dwords (no per-element division).
    const int per = B * stack;
    const int first = obs == nullptr ? 1 : 0;      // obs == NULL: only the next_obs half (+ scalars) is wanted
    const int f = blockIdx.y + first * per;
    const int which = f >= per;
    const int fi = f - which * per; // = b*stack + k
    const int b = fi / stack, k = fi - b * stack;
    const int slot = slot_tab[idx[b] * (2 * stack) + which * stack + k];
    const uint32_t *__restrict__ src = reinterpret_cast<const uint32_t *>(frames) + (size_t)slot * fq;
    TOut *out = which ? next_obs : obs;
    for (int off = blockIdx.x * 256 + threadIdx.x; off < fq; off += gridDim.x * 256) {
        const uint32_t v = src[off];
        const size_t r = (size_t)fi * fq + off;
        if constexpr (sizeof(TOut) == 1)
            store_nt(reinterpret_cast<uint32_t *>(out) + r, v);
        else
            store_nt(reinterpret_cast<float *>(out) + 4 * r,
                     make_float4((float)(v & 0xffu), (float)((v >> 8) & 0xffu), (float)((v >> 16) & 0xffu), (float)(v >> 24)));
    }
    if (blockIdx.x == 0 && which == first && k == 0) { // one workgroup per sample copies its scalars
        const int A = S - 2; // outputs as three contiguous blocks: actions [B][S-2], rewards [B], not_dones [B]
        for (int j = threadIdx.x; j < S; j += 256)
            scal_out[j < A ? (size_t)b * A + j : (size_t)B * A + (size_t)(j - A) * B + b] = scal[idx[b] * S + j];
    }
}

} // namespace augcl

using namespace augcl;

extern "C" int augcl_replay_sample(const uint8_t *frames, const int32_t *slot_tab, const float *scal, const int64_t *idx,
                                   void *obs, void *next_obs, float *scal_out, int B, int stack, int frame_bytes, int S,
                                   int out_f32, augcl_stream_t stream)
{
    const char *name = "augcl_replay_sample";
    AUGCL_REQUIRE(B >= 0 && stack > 0 && frame_bytes > 0 && (frame_bytes & 3) == 0 && S >= 0, "%s: bad shape", name);
    if (B == 0) return AUGCL_OK;
    AUGCL_REQUIRE(frames && slot_tab && idx && next_obs && (S == 0 || (scal && scal_out)), "%s: null buffer", name);
    AUGCL_REQUIRE(aligned16(frames) && aligned16(obs) && aligned16(next_obs), "%s: buffers must be 16-byte aligned", name);
    const int fq = frame_bytes / 4;
    AUGCL_REQUIRE((size_t)2 * B * stack <= 65535, "%s: batch too large for one launch", name);
    const dim3 grid((fq + 255) / 256 < 8 ? (fq + 255) / 256 : 8, (obs ? 2 : 1) * B * stack);
    hipStream_t s = as_stream(stream);
    if (out_f32)
        hipLaunchKernelGGL(k_replay_sample<float>, dim3(grid), dim3(256), 0, s, frames, slot_tab, scal, idx, (float *)obs,
                           (float *)next_obs, scal_out, B, stack, fq, S);
    else
        hipLaunchKernelGGL(k_replay_sample<uint8_t>, dim3(grid), dim3(256), 0, s, frames, slot_tab, scal, idx,
                           (uint8_t *)obs, (uint8_t *)next_obs, scal_out, B, stack, fq, S);
    return augcl::check_launch(name);
}

extern "C" int augcl_replay_gather(const uint8_t *frames, const int32_t *slots, void *out, int B, int stack,
                                   int frame_bytes, int out_f32, augcl_stream_t stream)
{
    const char *name = "augcl_replay_gather";
    AUGCL_REQUIRE(B >= 0 && stack > 0 && frame_bytes > 0 && (frame_bytes & 3) == 0, "%s: bad shape", name);
    if (B == 0) return AUGCL_OK;
    AUGCL_REQUIRE(frames && slots && out, "%s: null buffer", name);
    AUGCL_REQUIRE(aligned16(frames) && aligned16(out), "%s: buffers must be 16-byte aligned", name);
    const int fq = frame_bytes / 4;
    const size_t total = (size_t)B * stack * fq;
    const size_t blocks = (total + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 16384 ? blocks : 16384);
    hipStream_t s = as_stream(stream);
    if (out_f32)
        hipLaunchKernelGGL(k_replay_gather<float>, dim3(grid), dim3(256), 0, s, frames, slots, (float *)out, stack, fq,
                           total);
    else
        hipLaunchKernelGGL(k_replay_gather<uint8_t>, dim3(grid), dim3(256), 0, s, frames, slots, (uint8_t *)out, stack,
                           fq, total);
    return augcl::check_launch(name);
}
