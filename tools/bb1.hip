// tools/bb1.hip -- run ONE bf16 encoder kernel a few times (for rocprofv3 --pmc).  usage: bb1 which B H   (which: 0 fwd, 1 bwd-data, 2 wgrad)
#include "../augcl_amd/csrc/encoder_bf16.hip"
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
int main(int argc, char **argv)
{
    const int which = argc > 1 ? atoi(argv[1]) : 0, B = argc > 2 ? atoi(argv[2]) : 512, H = argc > 3 ? atoi(argv[3]) : 41;
    const size_t nbig = (size_t)B * 32 * (H + 2) * (H + 2);
    void *in, *out, *mask; float *w, *bias, *ws, *dw, *db;
    CK(hipMalloc(&in, nbig * 2)); CK(hipMalloc(&out, nbig * 4)); CK(hipMalloc(&mask, nbig * 2));
    CK(hipMalloc(&w, 9216 * 4)); CK(hipMalloc(&bias, 128)); CK(hipMalloc(&dw, 9216 * 4)); CK(hipMalloc(&db, 128));
    const size_t wsb = augcl_enc_wgrad_workspace_bytes(32); CK(hipMalloc(&ws, wsb));
    std::vector<uint16_t> h(nbig); for (size_t i = 0; i < nbig; ++i) h[i] = (uint16_t)(0x3c00 + ((i * 2654435761u >> 9) & 0x1ff) + ((i & 4) ? 0x8000 : 0));
    CK(hipMemcpy(in, h.data(), nbig * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(mask, h.data(), nbig * 2, hipMemcpyHostToDevice));
    std::vector<float> hw(9216); for (int i = 0; i < 9216; ++i) hw[i] = ((i * 37) % 101) / 300.f - 0.15f;
    CK(hipMemcpy(w, hw.data(), 9216 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bias, hw.data(), 128, hipMemcpyHostToDevice));
    for (int i = 0; i < 6; ++i) {
        if (which == 0) augcl_enc_bf16_conv32_forward(in, w, bias, out, B, H, H, 1, 0, 0);
        else if (which == 1) augcl_enc_bf16_conv32_backward_data(in, w, mask, out, B, H, H, 0);
        else augcl_enc_bf16_conv32_backward_weights_mfma(in, out, dw, db, ws, wsb, B, H, H, 0);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
