// encoder_reduce.hip -- fixed-order slab reduction for the backward-weights kernels, single or batched.
//   level 1: workgroup (x, y[, layer]) sums slabs y*nslab/8 .. of its layer for outputs x*256..;
//   level 2: adds the 8 partial rows and scatters into dW / db.
// The summation order depends only on (nslab, n), never on timing or on batching: gradients are bitwise
// reproducible, and the batched launch (whole-stack backward: 11 layers, 2 launches instead of 22) gives the same
// bits as the per-layer one.
#include "augcl_common.h"

#include <vector>

namespace augcl {

struct ReduceBatch {
    ReduceDesc d[kReduceMaxBatch];
};

__global__ __launch_bounds__(256) void k_reduce_slabs_l1(ReduceBatch b)
{
    const ReduceDesc &d = b.d[blockIdx.z];
    const int i = blockIdx.x * 256 + threadIdx.x, n = d.n;
    if (i >= n) return;
    const int per = (d.nslab + 7) / 8;
    const int g0 = blockIdx.y * per, g1 = min(d.nslab, g0 + per);
    const float *__restrict__ partial = d.partial;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int g = g0;
    for (; g + 3 < g1; g += 4) {
        s0 += partial[(size_t)g * n + i];
        s1 += partial[(size_t)(g + 1) * n + i];
        s2 += partial[(size_t)(g + 2) * n + i];
        s3 += partial[(size_t)(g + 3) * n + i];
    }
    for (; g < g1; ++g) s0 += partial[(size_t)g * n + i];
    d.part2[(size_t)blockIdx.y * n + i] = (s0 + s1) + (s2 + s3);
}

__global__ __launch_bounds__(256) void k_reduce_slabs_l2(ReduceBatch b)
{
    const ReduceDesc &d = b.d[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x, n = d.n;
    if (i >= n) return;
    float s = 0.f;
#pragma unroll
    for (int y = 0; y < 8; ++y) s += d.part2[(size_t)y * n + i];
    if (i < d.n_w) d.out_w[i] = s;
    else d.out_b[i - d.n_w] = s;
}

static thread_local std::vector<ReduceDesc> *tl_queue = nullptr;
static thread_local std::vector<ReduceDesc> tl_storage;

static int launch_batch(const ReduceDesc *d, int cnt, hipStream_t s, const char *what)
{
    for (int at = 0; at < cnt; at += kReduceMaxBatch) {
        const int m = cnt - at < kReduceMaxBatch ? cnt - at : kReduceMaxBatch;
        ReduceBatch b{};
        int nmax = 0;
        for (int j = 0; j < m; ++j) {
            b.d[j] = d[at + j];
            nmax = b.d[j].n > nmax ? b.d[j].n : nmax;
        }
        hipLaunchKernelGGL(k_reduce_slabs_l1, dim3((nmax + 255) / 256, 8, m), dim3(256), 0, s, b);
        hipLaunchKernelGGL(k_reduce_slabs_l2, dim3((nmax + 255) / 256, m), dim3(256), 0, s, b);
    }
    return check_launch(what);
}

int reduce_slabs(const ReduceDesc &d, hipStream_t s, const char *what)
{
    if (tl_queue) {
        tl_queue->push_back(d);
        return AUGCL_OK;
    }
    return launch_batch(&d, 1, s, what);
}

void reduce_defer_begin()
{
    tl_storage.clear();
    tl_queue = &tl_storage;
}

void reduce_defer_abort()
{
    tl_queue = nullptr;
    tl_storage.clear();
}

int reduce_defer_flush(hipStream_t s, const char *what)
{
    tl_queue = nullptr;
    if (tl_storage.empty()) return AUGCL_OK;
    const int rc = launch_batch(tl_storage.data(), (int)tl_storage.size(), s, what);
    tl_storage.clear();
    return rc;
}

} // namespace augcl
