// tools/fb1.hip -- run ONE f32 encoder kernel a few times (for rocprofv3 --pmc).  usage: fb1 which B H   (0 fwd, 1 bwd-data, 2 wgrad)
#include "../include/augcl.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
int main(int argc, char **argv)
{
    const int which = argc > 1 ? atoi(argv[1]) : 0, B = argc > 2 ? atoi(argv[2]) : 512, H = argc > 3 ? atoi(argv[3]) : 41;
    const size_t nbig = (size_t)B * 32 * (H + 2) * (H + 2);
    float *in, *out, *mask, *w, *bias, *ws, *dw, *db;
    CK(hipMalloc(&in, nbig * 4)); CK(hipMalloc(&out, nbig * 4)); CK(hipMalloc(&mask, nbig * 4));
    CK(hipMalloc(&w, 9216 * 4)); CK(hipMalloc(&bias, 128)); CK(hipMalloc(&dw, 9216 * 4)); CK(hipMalloc(&db, 128));
    const size_t wsb = augcl_enc_wgrad_workspace_bytes(32); CK(hipMalloc(&ws, wsb));
    std::vector<float> h(nbig); for (size_t i = 0; i < nbig; ++i) h[i] = (float)((i * 2654435761u >> 9) & 1023) / 512.f - 0.7f;
    CK(hipMemcpy(in, h.data(), nbig * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(mask, h.data(), nbig * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, h.data(), 9216 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bias, h.data(), 128, hipMemcpyHostToDevice));
    for (int i = 0; i < 6; ++i) {
        int rc;
        if (which == 0) rc = augcl_enc_conv32_forward(in, w, bias, out, B, H, H, 1, 0);
        else if (which == 1) rc = augcl_enc_conv32_backward_data(in, w, mask, out, B, H, H, 0);
        else rc = augcl_enc_conv32_backward_weights(in, out, dw, db, ws, wsb, B, H, H, 0);
        if (rc) { printf("rc=%d %s\n", rc, augcl_last_error()); return 1; }
    }
    CK(hipDeviceSynchronize());
    return 0;
}
