// csplat_image.hip -- separable 11-tap Gaussian window of the SSIM loss (utils/loss_utils.py:30-58: five grouped
// 11x11 conv2d per SSIM evaluation; the window is the outer product of a 1-D Gaussian with itself).  "Next" row N2 of
// sits right after the rasterizer in every train step.  One kernel does both passes through an LDS tile:
// HBM-bound, 4 B read + 4 B written per pixel (the grouped-conv path measured 1.2-1.8 ms per call on 5x3x3x800x800;
// this is a 46 MB round trip).  The window is symmetric and the padding is zero, so the operator is self-adjoint: the
// backward of blur is the same kernel applied to the incoming gradient.
#include "csplat_common.h"

namespace {
constexpr int BW = 64, BH = 16, R5 = 5;
struct Taps { float w[11]; };

__global__ __launch_bounds__(256) void k_blur11(int H, int W, Taps taps, const float *__restrict__ in, float *__restrict__ out) {
    __shared__ float s_in[(BH + 2 * R5)][BW + 2 * R5 + 1];
    __shared__ float s_h[(BH + 2 * R5)][BW + 1];
    const size_t img = (size_t)blockIdx.z * H * W;
    const int x0 = blockIdx.x * BW, y0 = blockIdx.y * BH;
    for (int t = threadIdx.x; t < (BH + 2 * R5) * (BW + 2 * R5); t += 256) {
        const int ry = t / (BW + 2 * R5), rx = t - ry * (BW + 2 * R5);
        const int y = y0 + ry - R5, x = x0 + rx - R5;
        s_in[ry][rx] = (y >= 0 && y < H && x >= 0 && x < W) ? in[img + (size_t)y * W + x] : 0.f;   // zero padding
    }
    __syncthreads();
    for (int t = threadIdx.x; t < (BH + 2 * R5) * BW; t += 256) {
        const int ry = t / BW, rx = t - ry * BW;
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) a += taps.w[k] * s_in[ry][rx + k];
        s_h[ry][rx] = a;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < BH * BW; t += 256) {
        const int ry = t / BW, rx = t - ry * BW;
        const int y = y0 + ry, x = x0 + rx;
        if (y < H && x < W) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) a += taps.w[k] * s_h[ry + k][rx];
            out[img + (size_t)y * W + x] = a;
        }
    }
}
}  // namespace

extern "C" int csplat_blur11(void *stream, int64_t n_images, int H, int W, const float *taps11, const float *in, float *out) {
    CSPLAT_REQUIRE(n_images >= 0 && H > 0 && W > 0 && taps11 && in && out && in != out, "csplat_blur11: bad arguments");
    CSPLAT_REQUIRE(n_images < 65536, "csplat_blur11: too many images for one launch");
    if (n_images == 0) return 0;
    Taps t;
    memcpy(t.w, taps11, sizeof(t.w));
    dim3 grid(cdiv(W, BW), cdiv(H, BH), (unsigned)n_images);
    k_blur11<<<grid, 256, 0, (hipStream_t)stream>>>(H, W, t, in, out);
    LAUNCH_CHECK();
    return 0;
}
