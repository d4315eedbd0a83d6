// aug_extra_kernels.hip -- the rest of the reference's augmentation registry:
// random_crop / random_flip / random_rotation (one per-sample integer-affine gather), random_cutout_color,
// random_grayscale.  All are index / select operators on the pixel grid: HBM-bound, bit-exact.
#include "augcl_common.h"

namespace augcl {

// out[b,c,y,x] = in[b,c, ay*y + by*x + cy, ax*y + bx*x + cx]   (coef[b] = {ay,by,cy,ax,bx,cx})
//   crop (oy,ox): {1,0,oy, 0,1,ox};  hflip: {1,0,0, 0,-1,W-1};  rot90: {0,1,0, -1,0,W-1};
//   rot180: {-1,0,H-1, 0,-1,W-1};  rot270: {0,-1,H-1, 1,0,0};  identity: {1,0,0, 0,1,0}
__global__ __launch_bounds__(256) void k_affine_gather(const float *__restrict__ in, float *__restrict__ out, int C,
                                                       int Hin, int Win, int Hout, int Wout,
                                                       const int32_t *__restrict__ coef)
{
    const int plane = blockIdx.y; // b*C + c
    const int b = plane / C;
    const int ay = coef[b * 6 + 0], by = coef[b * 6 + 1], cy = coef[b * 6 + 2];
    const int ax = coef[b * 6 + 3], bx = coef[b * 6 + 4], cx = coef[b * 6 + 5];
    const float *__restrict__ src = in + (size_t)plane * Hin * Win;
    float *__restrict__ dst = out + (size_t)plane * Hout * Wout;
    const int n = Hout * Wout;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const int y = j / Wout, x = j - y * Wout;
        int sy = ay * y + by * x + cy, sx = ax * y + bx * x + cx;
        sy = sy < 0 ? 0 : (sy >= Hin ? Hin - 1 : sy); // host validates the coefficients; stay in bounds regardless
        sx = sx < 0 ? 0 : (sx >= Win ? Win - 1 : sx);
        __builtin_nontemporal_store(src[sy * Win + sx], &dst[j]);
    }
}

// random_cutout_color (src/augmentations.py:327-350): box rows h1..2*h1-1, cols w1..2*w1-1 <- colour[b][c]
__global__ __launch_bounds__(256) void k_cutout_color(const float *__restrict__ in, float *__restrict__ out, int C, int H,
                                                      int W, const int32_t *__restrict__ box,
                                                      const float *__restrict__ color)
{
    const int plane = blockIdx.y;
    const int b = plane / C;
    const int h1 = box[2 * b], w1 = box[2 * b + 1];
    const float col = color[plane];
    const int n = H * W;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const int y = j / W, x = j - y * W;
        const bool inside = y >= h1 && y < 2 * h1 && x >= w1 && x < 2 * w1;
        __builtin_nontemporal_store(inside ? col : in[(size_t)plane * n + j], &out[(size_t)plane * n + j]);
    }
}

// random_grayscale (src/augmentations.py:89-136), literal: t = uint8(x*255) (C-style wrap, what torch's CPU cast
// does), gray = uint8(0.2989 r + 0.587 g + 0.114 b) on t, out = (mask ? gray : t) / 255.
__device__ __forceinline__ float u8wrap(float v) { return (float)((int)v & 255); }

__global__ __launch_bounds__(256) void k_random_grayscale(const float *__restrict__ in, float *__restrict__ out,
                                                          int frames, int n /* H*W */, const uint8_t *__restrict__ mask)
{
    const int fidx = blockIdx.y; // b*frames + f
    const int b = fidx / frames;
    const bool m = mask[b] != 0;
    const float *__restrict__ src = in + (size_t)fidx * 3 * n;
    float *__restrict__ dst = out + (size_t)fidx * 3 * n;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const float r = u8wrap(src[j] * 255.0f), g = u8wrap(src[n + j] * 255.0f), bl = u8wrap(src[2 * n + j] * 255.0f);
        const float gray = u8wrap((r * 0.2989f + g * 0.587f) + bl * 0.114f);
        __builtin_nontemporal_store((m ? gray : r) / 255.0f, &dst[j]);
        __builtin_nontemporal_store((m ? gray : g) / 255.0f, &dst[n + j]);
        __builtin_nontemporal_store((m ? gray : bl) / 255.0f, &dst[2 * n + j]);
    }
}

// splice / CS_splice (src/augmentations.py:277-324).  Frame j = b*frames + f of the (B*frames,3,H,W) view pairs with
// overlay image j % B -- the reference tiles the image batch along dim 0 (`.repeat(frames,1,1,1)`), kept as is.
// HSV planes follow kornia.color.rgb_to_hsv (0.5.x): v = max, s = (max-min)/(v+eps), first-maximum channel selects
// h6 in {b'-g', 2d+r'-b', 4d+g'-r'} (x' = max-x, d = max-min or 1), h = 2*pi*((h6/d/6) mod 1).
// keep  = all(hsv > thres);  splice: out = keep ? rgb : overlay;  CS_splice: out = (keep2 | !keep) ? rgb : overlay
// with keep2 = all(hsv > (0, sat_thres, 0.6)).  Everything * 255 at the end, rgb = x / 255.
__global__ __launch_bounds__(256) void k_splice(const float *__restrict__ x, const float *__restrict__ imgs,
                                                float *__restrict__ out, int B, int n /* H*W */, int cs_mode, float ht,
                                                float st, float vt, float eps)
{
    const int j = blockIdx.y; // frame index in the (B*frames) view
    const float *__restrict__ src = x + (size_t)j * 3 * n;
    const float *__restrict__ ov = imgs + (size_t)(j % B) * 3 * n;
    float *__restrict__ dst = out + (size_t)j * 3 * n;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) {
        const float r = src[p] / 255.0f, g = src[n + p] / 255.0f, b = src[2 * n + p] / 255.0f;
        const float mx = fmaxf(fmaxf(r, g), b), mn = fminf(fminf(r, g), b);
        const float dc = mx - mn;
        const float sat = dc / (mx + eps);
        const float d = dc == 0.0f ? 1.0f : dc;
        const float rc = mx - r, gc = mx - g, bc = mx - b;
        float h6;
        if (r == mx) h6 = bc - gc;
        else if (g == mx) h6 = (2.0f * d + rc) - bc;
        else h6 = (4.0f * d + gc) - rc;
        float h = (h6 / d) / 6.0f;
        h = fmodf(h, 1.0f);
        if (h != 0.0f && h < 0.0f) h += 1.0f;
        h = 6.283185307179586f * h;
        const bool keep = h > ht && sat > st && mx > vt;
        bool use_rgb = keep;
        if (cs_mode) use_rgb = (h > 0.0f && sat > st && mx > 0.6f) || !keep;
        __builtin_nontemporal_store((use_rgb ? r : ov[p]) * 255.0f, &dst[p]);
        __builtin_nontemporal_store((use_rgb ? g : ov[n + p]) * 255.0f, &dst[n + p]);
        __builtin_nontemporal_store((use_rgb ? b : ov[2 * n + p]) * 255.0f, &dst[2 * n + p]);
    }
}

// VideoWrapper._interpolate_bg (src/env/wrappers.py:336-340): the video frame as float / 255, F.interpolate(bilinear,
// align_corners=False) to the observation size, * 255, .byte().  The trailing uint8 cast TRUNCATES, so matching the
// reference bit for bit needs ATen's float32 arithmetic operation by operation (pinned against the reference's own
// function run on this torch build, tests/golden/make_golden.py:make_eval): source coordinate = fma(scale, dst + 0.5,
// -0.5) clamped at 0 with scale = float(in) / float(out), weights (1 - l, l) with l = coordinate - floor, and
// w0 * a + w1 * b evaluated as fma(w0, a, w1 * b) on both axes, horizontal first.  This file is compiled with
// -ffp-contract=off: the fmas below are exactly the ones written.
struct BgAxis {
    int i0, i1;
    float l0, l1;
};
__device__ __forceinline__ BgAxis bg_axis(int d, int n_in, int n_out, float scale)
{
    BgAxis a;
    if (n_in == n_out) { // ATen: identical sizes copy through (index d, weights 1 / 0)
        a.i0 = a.i1 = d;
        a.l0 = 1.0f;
        a.l1 = 0.0f;
        return a;
    }
    float src = __builtin_fmaf(scale, (float)d + 0.5f, -0.5f);
    src = src < 0.0f ? 0.0f : src;
    a.i0 = min((int)floorf(src), n_in - 1);
    a.i1 = min(a.i0 + 1, n_in - 1);
    a.l1 = fminf(fmaxf(src - (float)a.i0, 0.0f), 1.0f);
    a.l0 = 1.0f - a.l1;
    return a;
}
__device__ __forceinline__ uint8_t bg_pixel(const uint8_t *__restrict__ plane, int Wb, const BgAxis &ay, const BgAxis &ax)
{
    const float p00 = (float)plane[ay.i0 * Wb + ax.i0] / 255.0f, p01 = (float)plane[ay.i0 * Wb + ax.i1] / 255.0f;
    const float p10 = (float)plane[ay.i1 * Wb + ax.i0] / 255.0f, p11 = (float)plane[ay.i1 * Wb + ax.i1] / 255.0f;
    const float top = __builtin_fmaf(ax.l0, p00, ax.l1 * p01);
    const float bot = __builtin_fmaf(ax.l0, p10, ax.l1 * p11);
    return (uint8_t)(__builtin_fmaf(ay.l0, top, ay.l1 * bot) * 255.0f);
}

__global__ __launch_bounds__(256) void k_interpolate_bg(const uint8_t *__restrict__ bg, uint8_t *__restrict__ out, int Hb,
                                                        int Wb, int H, int W)
{
    const size_t f = blockIdx.y;
    const float sh = (float)Hb / (float)H, sw = (float)Wb / (float)W;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < H * W; p += gridDim.x * 256) {
        const int y = p / W, x = p - y * W;
        const BgAxis ay = bg_axis(y, Hb, H, sh), ax = bg_axis(x, Wb, W, sw);
#pragma unroll
        for (int c = 0; c < 3; ++c) out[(f * 3 + c) * H * W + p] = bg_pixel(bg + (f * 3 + c) * (size_t)Hb * Wb, Wb, ay, ax);
    }
}

// do_green_screen (src/env/wrappers.py:209-260): a pixel whose HSV (python colorsys arithmetic in DOUBLE, as the
// reference's per-pixel Python loop computes it) falls in h*360 in [100,185], s*255 in [80,255], v*255 in [70,255]
// is replaced by the background pixel.  uint8 in, uint8 out: bit-exact.
// RESIZE: bg is the raw (3,Hb,Wb) video frame and the keyed-in pixel is _interpolate_bg's value computed on the spot
// (VideoWrapper._greenscreen = _interpolate_bg + do_green_screen, wrappers.py:342-348, in one launch).
template <bool RESIZE>
__global__ __launch_bounds__(256) void k_green_screen(const uint8_t *__restrict__ x, const uint8_t *__restrict__ bg,
                                                      uint8_t *__restrict__ out, int n /* H*W */, int W, int Hb, int Wb)
{
    const size_t f = blockIdx.y; // frame
    const uint8_t *px = x + f * 3 * n, *pb = bg + f * 3 * (RESIZE ? (size_t)Hb * Wb : (size_t)n);
    uint8_t *po = out + f * 3 * n;
    const int H = n / W;
    const float sh = RESIZE ? (float)Hb / (float)H : 1.0f, sw = RESIZE ? (float)Wb / (float)W : 1.0f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) {
        const uint8_t r8 = px[p], g8 = px[n + p], b8 = px[2 * n + p];
        const double r = r8 / 255.0, g = g8 / 255.0, b = b8 / 255.0;
        const double maxc = fmax(fmax(r, g), b), minc = fmin(fmin(r, g), b);
        double h = 0.0, s = 0.0;
        if (minc != maxc) {
            const double d = maxc - minc;
            s = d / maxc;
            const double rc = (maxc - r) / d, gc = (maxc - g) / d, bc = (maxc - b) / d;
            if (r == maxc) h = bc - gc;
            else if (g == maxc) h = 2.0 + rc - bc;
            else h = 4.0 + gc - rc;
            h = h / 6.0;
            h = h - floor(h); // python's  % 1.0
        }
        const double hh = h * 360, ss = s * 255, vv = maxc * 255;
        const bool key = 100 <= hh && hh <= 185 && 80 <= ss && ss <= 255 && 70 <= vv && vv <= 255;
        uint8_t k0 = r8, k1 = g8, k2 = b8;
        if (key) {
            if (RESIZE) {
                const int y = p / W, xx = p - y * W;
                const BgAxis ay = bg_axis(y, Hb, H, sh), ax = bg_axis(xx, Wb, W, sw);
                k0 = bg_pixel(pb, Wb, ay, ax);
                k1 = bg_pixel(pb + (size_t)Hb * Wb, Wb, ay, ax);
                k2 = bg_pixel(pb + 2 * (size_t)Hb * Wb, Wb, ay, ax);
            } else {
                k0 = pb[p]; k1 = pb[n + p]; k2 = pb[2 * n + p];
            }
        }
        po[p] = k0;
        po[n + p] = k1;
        po[2 * n + p] = k2;
    }
}

} // namespace augcl

using namespace augcl;

extern "C" {

int augcl_affine_gather_f32(const float *in, float *out, int B, int C, int Hin, int Win, int Hout, int Wout,
                            const int32_t *coef, augcl_stream_t stream)
{
    const char *name = "augcl_affine_gather_f32";
    AUGCL_REQUIRE(B >= 0 && C > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "%s: bad shape", name);
    if (B == 0) return AUGCL_OK;
    AUGCL_REQUIRE(in && out && coef && in != out, "%s: bad tensor", name);
    const int n = Hout * Wout;
    const int gx = (n + 255) / 256 < 32 ? (n + 255) / 256 : 32;
    hipLaunchKernelGGL(k_affine_gather, dim3(gx, B * C), dim3(256), 0, as_stream(stream), in, out, C, Hin, Win, Hout, Wout,
                       coef);
    return augcl::check_launch(name);
}

int augcl_cutout_color_f32(const float *in, float *out, int B, int C, int H, int W, const int32_t *box,
                           const float *color, augcl_stream_t stream)
{
    const char *name = "augcl_cutout_color_f32";
    AUGCL_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0, "%s: bad shape", name);
    if (B == 0) return AUGCL_OK;
    AUGCL_REQUIRE(in && out && box && color, "%s: null tensor", name);
    const int n = H * W;
    const int gx = (n + 255) / 256 < 32 ? (n + 255) / 256 : 32;
    hipLaunchKernelGGL(k_cutout_color, dim3(gx, B * C), dim3(256), 0, as_stream(stream), in, out, C, H, W, box, color);
    return augcl::check_launch(name);
}

int augcl_random_grayscale_f32(const float *in, float *out, int B, int frames, int H, int W, const uint8_t *mask,
                               augcl_stream_t stream)
{
    const char *name = "augcl_random_grayscale_f32";
    AUGCL_REQUIRE(B >= 0 && frames > 0 && H > 0 && W > 0, "%s: bad shape", name);
    if (B == 0) return AUGCL_OK;
    AUGCL_REQUIRE(in && out && mask, "%s: null tensor", name);
    const int n = H * W;
    const int gx = (n + 255) / 256 < 32 ? (n + 255) / 256 : 32;
    hipLaunchKernelGGL(k_random_grayscale, dim3(gx, B * frames), dim3(256), 0, as_stream(stream), in, out, frames, n, mask);
    return augcl::check_launch(name);
}

int augcl_splice_f32(const float *x, const float *imgs, float *out, int B, int frames, int H, int W, int cs_mode,
                     float hue_thres, float sat_thres, float val_thres, float eps, augcl_stream_t stream)
{
    const char *name = "augcl_splice_f32";
    AUGCL_REQUIRE(B >= 0 && frames > 0 && H > 0 && W > 0, "%s: bad shape", name);
    if (B == 0) return AUGCL_OK;
    AUGCL_REQUIRE(x && imgs && out && x != out, "%s: bad tensor", name);
    const int n = H * W;
    const int gx = (n + 255) / 256 < 32 ? (n + 255) / 256 : 32;
    hipLaunchKernelGGL(k_splice, dim3(gx, B * frames), dim3(256), 0, as_stream(stream), x, imgs, out, B, n, cs_mode,
                       hue_thres, sat_thres, val_thres, eps);
    return augcl::check_launch(name);
}

int augcl_green_screen_u8(const uint8_t *x, const uint8_t *bg, uint8_t *out, int N, int H, int W, augcl_stream_t stream)
{
    const char *name = "augcl_green_screen_u8";
    AUGCL_REQUIRE(N >= 0 && H > 0 && W > 0, "%s: bad shape", name);
    if (N == 0) return AUGCL_OK;
    AUGCL_REQUIRE(x && bg && out, "%s: null tensor", name);
    const int n = H * W;
    const int gx = (n + 255) / 256 < 64 ? (n + 255) / 256 : 64;
    hipLaunchKernelGGL(k_green_screen<false>, dim3(gx, N), dim3(256), 0, as_stream(stream), x, bg, out, n, W, 0, 0);
    return augcl::check_launch(name);
}

int augcl_interpolate_bg_u8(const uint8_t *bg, uint8_t *out, int N, int Hb, int Wb, int H, int W, augcl_stream_t stream)
{
    const char *name = "augcl_interpolate_bg_u8";
    AUGCL_REQUIRE(N >= 0 && H > 0 && W > 0 && Hb > 0 && Wb > 0, "%s: bad shape", name);
    if (N == 0) return AUGCL_OK;
    AUGCL_REQUIRE(bg && out, "%s: null tensor", name);
    const int n = H * W;
    const int gx = (n + 255) / 256 < 64 ? (n + 255) / 256 : 64;
    hipLaunchKernelGGL(k_interpolate_bg, dim3(gx, N), dim3(256), 0, as_stream(stream), bg, out, Hb, Wb, H, W);
    return augcl::check_launch(name);
}

int augcl_green_screen_resize_u8(const uint8_t *x, const uint8_t *bg_raw, uint8_t *out, int N, int H, int W, int Hb, int Wb,
                                 augcl_stream_t stream)
{
    const char *name = "augcl_green_screen_resize_u8";
    AUGCL_REQUIRE(N >= 0 && H > 0 && W > 0 && Hb > 0 && Wb > 0, "%s: bad shape", name);
    if (N == 0) return AUGCL_OK;
    AUGCL_REQUIRE(x && bg_raw && out, "%s: null tensor", name);
    const int n = H * W;
    const int gx = (n + 255) / 256 < 64 ? (n + 255) / 256 : 64;
    hipLaunchKernelGGL(k_green_screen<true>, dim3(gx, N), dim3(256), 0, as_stream(stream), x, bg_raw, out, n, W, Hb, Wb);
    return augcl::check_launch(name);
}

} // extern "C"
