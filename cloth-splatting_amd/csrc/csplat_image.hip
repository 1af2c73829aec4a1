// csplat_image.hip -- image-space losses: the separable 11-tap Gaussian window (csplat_blur11), the fused L1 loss
// (csplat_l1) and the fused SSIM (csplat_ssim_fwd / _bwd).  The window: SSIM loss (utils/loss_utils.py:30-58: five grouped
// 11x11 conv2d per SSIM evaluation; the window is the outer product of a 1-D Gaussian with itself).  "Next" row N2 of
// sits right after the rasterizer in every train step.  One kernel does both passes through an LDS tile:
// HBM-bound, 4 B read + 4 B written per pixel (the grouped-conv path measured 1.2-1.8 ms per call on 5x3x3x800x800;
// this is a 46 MB round trip).  The window is symmetric and the padding is zero, so the operator is self-adjoint: the
// backward of blur is the same kernel applied to the incoming gradient.
#include "csplat_common.h"
#include <math.h>
#include <stdlib.h>

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

// ---- fused SSIM (utils/loss_utils.py:40-70, size_average form).  Forward: ONE kernel loads an x / y tile with its 5-pixel
// halo, forms x^2, y^2, xy in LDS, runs the five separable 11x11 windows, evaluates the SSIM map and its three partial
// derivatives (w.r.t. mu1 = blur(x), s11 = blur(x^2), s12 = blur(xy)) and a per-workgroup sum of the map.  Backward: ONE
// kernel blurs the three partials (the window is self-adjoint) and combines g * (b1 + 2x b2 + y b3).  The composed form
// is a cat, a blur launch and ~25 elementwise launches forward, ~50 backward, each a full-image HBM round trip.
constexpr float SSIM_C1 = 0.01f * 0.01f, SSIM_C2 = 0.03f * 0.03f;

constexpr int SSIM_THREADS = 512;   // 8 waves share one tile's LDS: the load / horizontal / vertical phases of the 3 resident workgroups overlap better
__global__ __launch_bounds__(SSIM_THREADS) void k_ssim_fwd(int H, int W, Taps taps, const float *__restrict__ X, const float *__restrict__ Y,
                                                   float *__restrict__ P1, float *__restrict__ P2, float *__restrict__ P3,
                                                   float *__restrict__ map_out, float *__restrict__ partial,
                                                   const float *__restrict__ mask, int mask_channels, int channels) {
    __shared__ float s_x[(BH + 2 * R5)][BW + 2 * R5 + 1];
    __shared__ float s_y[(BH + 2 * R5)][BW + 2 * R5 + 1];
    __shared__ float s_h[5][(BH + 2 * R5)][BW + 1];
    __shared__ float s_red[SSIM_THREADS / 64];
    const size_t img = (size_t)blockIdx.z * H * W;
    const int x0 = blockIdx.x * BW, y0 = blockIdx.y * BH;
    for (int t = threadIdx.x; t < (BH + 2 * R5) * (BW + 2 * R5); t += SSIM_THREADS) {
        const int ry = t / (BW + 2 * R5), rx = t - ry * (BW + 2 * R5);
        const int y = y0 + ry - R5, x = x0 + rx - R5;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;       // zero padding (of the images AND of their products)
        s_x[ry][rx] = in ? X[img + (size_t)y * W + x] : 0.f;
        s_y[ry][rx] = in ? Y[img + (size_t)y * W + x] : 0.f;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < (BH + 2 * R5) * BW; t += SSIM_THREADS) {
        const int ry = t / BW, rx = t - ry * BW;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = taps.w[k], xv = s_x[ry][rx + k], yv = s_y[ry][rx + k];
            a0 += w * xv; a1 += w * yv; a2 += w * (xv * xv); a3 += w * (yv * yv); a4 += w * (xv * yv);
        }
        s_h[0][ry][rx] = a0; s_h[1][ry][rx] = a1; s_h[2][ry][rx] = a2; s_h[3][ry][rx] = a3; s_h[4][ry][rx] = a4;
    }
    __syncthreads();
    float acc = 0.f;
    for (int t = threadIdx.x; t < BH * BW; t += SSIM_THREADS) {
        const int ry = t / BW, rx = t - ry * BW;
        const int y = y0 + ry, x = x0 + rx;
        if (y < H && x < W) {
            float mu1 = 0.f, mu2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const float w = taps.w[k];
                mu1 += w * s_h[0][ry + k][rx]; mu2 += w * s_h[1][ry + k][rx]; s11 += w * s_h[2][ry + k][rx];
                s22 += w * s_h[3][ry + k][rx]; s12 += w * s_h[4][ry + k][rx];
            }
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
            const float A1 = 2.f * mu12 + SSIM_C1, A2 = 2.f * (s12 - mu12) + SSIM_C2;
            const float B1 = mu1_sq + mu2_sq + SSIM_C1, B2 = (s11 - mu1_sq) + (s22 - mu2_sq) + SSIM_C2;
            const float inv = 1.f / (B1 * B2);
            const float S = A1 * A2 * inv;
            const size_t o = img + (size_t)y * W + x;
            if (map_out) map_out[o] = S;
            // masked form (train_utils.py:64-67): the reduced quantity is sum((1 - S) * m) and the partials carry the weight m
            float m = 1.f;
            if (mask) {
                const size_t plane = mask_channels == 1 ? blockIdx.z / channels : blockIdx.z;
                m = mask[(plane * H + y) * W + x];
                acc += (1.f - S) * m;
            } else acc += S;
            if (P1) {
                P1[o] = m * (2.f * mu2 * (A2 - A1) * inv - S * 2.f * mu1 * (B2 - B1) * inv);   // dS/dmu1
                P2[o] = m * (-S / B2);                                                         // dS/ds11
                P3[o] = m * (2.f * A1 * inv);                                                  // dS/ds12
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t_ = 0.f;
        for (int k = 0; k < SSIM_THREADS / 64; k++) t_ += s_red[k];
        partial[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = t_;
    }
}

__global__ __launch_bounds__(SSIM_THREADS) void k_ssim_bwd(int H, int W, Taps taps, const float *__restrict__ X, const float *__restrict__ Y,
                                                   const float *__restrict__ P1, const float *__restrict__ P2,
                                                   const float *__restrict__ P3, const float *__restrict__ gscalar, float inv_n,
                                                   const float *__restrict__ addend, const float *__restrict__ add_scale,
                                                   float *__restrict__ dX) {
    __shared__ float s_p[3][(BH + 2 * R5)][BW + 2 * R5 + 1];
    __shared__ float s_h[3][(BH + 2 * R5)][BW + 1];
    const size_t img = (size_t)blockIdx.z * H * W;
    const int x0 = blockIdx.x * BW, y0 = blockIdx.y * BH;
    for (int t = threadIdx.x; t < (BH + 2 * R5) * (BW + 2 * R5); t += SSIM_THREADS) {
        const int ry = t / (BW + 2 * R5), rx = t - ry * (BW + 2 * R5);
        const int y = y0 + ry - R5, x = x0 + rx - R5;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        const size_t o = img + (size_t)y * W + x;
        s_p[0][ry][rx] = in ? P1[o] : 0.f; s_p[1][ry][rx] = in ? P2[o] : 0.f; s_p[2][ry][rx] = in ? P3[o] : 0.f;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < (BH + 2 * R5) * BW; t += SSIM_THREADS) {
        const int ry = t / BW, rx = t - ry * BW;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float w = taps.w[k];
            a0 += w * s_p[0][ry][rx + k]; a1 += w * s_p[1][ry][rx + k]; a2 += w * s_p[2][ry][rx + k];
        }
        s_h[0][ry][rx] = a0; s_h[1][ry][rx] = a1; s_h[2][ry][rx] = a2;
    }
    __syncthreads();
    const float g = gscalar[0] * inv_n;
    const float ga = addend ? add_scale[0] : 0.f;
    for (int t = threadIdx.x; t < BH * BW; t += SSIM_THREADS) {
        const int ry = t / BW, rx = t - ry * BW;
        const int y = y0 + ry, x = x0 + rx;
        if (y < H && x < W) {
            float b1 = 0.f, b2 = 0.f, b3 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; k++) {
                const float w = taps.w[k];
                b1 += w * s_h[0][ry + k][rx]; b2 += w * s_h[1][ry + k][rx]; b3 += w * s_h[2][ry + k][rx];
            }
            const size_t o = img + (size_t)y * W + x;
            const float d = g * (b1 + 2.f * X[o] * b2 + Y[o] * b3);
            dX[o] = addend ? d + ga * addend[o] : d;
        }
    }
}

// ---- the WHOLE image loss of a train step in one launch each way (train_utils.py:50-74 + the PSNR the reference logs every step,
// :262-283): Ll1 + lambda (1 - SSIM) [masked: mean|(x - y) m| + lambda mean((1 - S) m)], the per-camera PSNR, and -- optionally -- the
// sum with another device scalar (the regularisers), so that the loss the step differentiates leaves the library finished.  k_ssim_fwd's
// tile already holds x and y: the L1 term, the sign byte of its gradient and the squared error ride on the third phase.  Workgroup
// partials (3 floats) are summed in index order by k_image_loss_finish: bit-reproducible, no float atomics.
// Replaces k_l1 + k_psnr + k_ssim_fwd and eight stock reductions / elementwise launches.
// k_image_loss_fwd / _bwd, REGISTER-BLOCKED since round 5.  The round-4 forms (one output per thread and pass: every output re-read its
// eleven taps from LDS, re-formed x^2 / y^2 / xy per tap; counters: 428 lane-instructions and 79 LDS instructions per pixel, vector pipe
// ~45 % busy, 53 % of the wave-cycles waiting) measured 80.2 / 54.9 us at [3,3,800,800].  Here a thread owns a strip: the horizontal pass
// makes 4 consecutive outputs of a row from 14 loaded values per map (products formed once per loaded value), the vertical pass 2
// consecutive rows of a column from 12 values per map -- half the instructions -- for 76.1 / 48.1 us (same box, rocprofv3).  Not more,
// because what the kernel waits for is a tile's LIFE (load, barrier, pass, barrier, pass, stores) with three tiles resident per CU (51 KB of
// LDS each): the same scheme on 256 threads per tile (8 x 1 / 1 x 4 strips) was SLOWER than the round-4 kernels (97.6 / 61.3 us), and at 84
// VGPRs (5 waves per SIMD: two tiles per CU) the forward took 92.6 us -- hence __launch_bounds__(512, 6).  Same tile (64 x 16), same
// partial-sum layout, same arithmetic per output (tap order k = 0 .. 10): the finish kernel and the parity tests are unchanged.
constexpr int IL_T = 512, IL_HR = BH + 2 * R5, IL_XP = BW + 2 * R5 + 2, IL_HP = BW + 4;      // 26 halo rows; pitches 76 / 68 floats
constexpr int IL_HS = 4, IL_VS = 2;            // outputs per horizontal / vertical strip: 26 x 16 = 416 and 64 x 8 = 512 strips for 512 threads
static_assert((BW / IL_HS) * IL_HR <= IL_T && BW * (BH / IL_VS) == IL_T, "one strip per thread");
static_assert(BW == 64 && BH == 16, "the strip assignment below is written for 64 x 16 tiles");

// horizontal 11-tap pass of one row strip: out[o] = sum_k w[k] v[o + k], o = 0 .. IL_HS - 1
__device__ __forceinline__ void il_hrow(const Taps &taps, const float (&v)[IL_HS + 10], float *dst) {
    float a[IL_HS];
#pragma unroll
    for (int o = 0; o < IL_HS; o++) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) t += taps.w[k] * v[o + k];
        a[o] = t;
    }
    static_assert(IL_HS == 4, "one 16-byte store per strip");
    reinterpret_cast<float4 *>(dst)[0] = make_float4(a[0], a[1], a[2], a[3]);
}
// vertical 11-tap pass of one column strip: out[r] = sum_k w[k] s[(row0 + r + k) * IL_HP], r = 0 .. IL_VS - 1
__device__ __forceinline__ void il_vcol(const Taps &taps, const float *s, float (&out)[IL_VS]) {
    float v[IL_VS + 10];
#pragma unroll
    for (int j = 0; j < IL_VS + 10; j++) v[j] = s[j * IL_HP];
#pragma unroll
    for (int r = 0; r < IL_VS; r++) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) t += taps.w[k] * v[r + k];
        out[r] = t;
    }
}

__global__ __launch_bounds__(IL_T, 6) void k_image_loss_fwd(int H, int W, Taps taps, const float *__restrict__ X, const float *__restrict__ Y,
                                                           float *__restrict__ P1, float *__restrict__ P2, float *__restrict__ P3,
                                                           signed char *__restrict__ sign8, const float *__restrict__ mask, int mask_channels,
                                                           int channels, float *__restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float s_x[IL_HR][IL_XP];
    __shared__ __attribute__((aligned(16))) float s_y[IL_HR][IL_XP];
    __shared__ __attribute__((aligned(16))) float s_h[5][IL_HR][IL_HP];
    __shared__ float s_red[3][IL_T / 64];
    const size_t img = (size_t)blockIdx.z * H * W;
    const int x0 = blockIdx.x * BW, y0 = blockIdx.y * BH;
    for (int t = threadIdx.x; t < IL_HR * (BW + 2 * R5); t += IL_T) {
        const int ry = t / (BW + 2 * R5), rx = t - ry * (BW + 2 * R5);
        const int y = y0 + ry - R5, x = x0 + rx - R5;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        s_x[ry][rx] = in ? X[img + (size_t)y * W + x] : 0.f;
        s_y[ry][rx] = in ? Y[img + (size_t)y * W + x] : 0.f;
    }
    __syncthreads();
    constexpr int NS = BW / IL_HS, NV = IL_HS + 10;
    if (threadIdx.x < IL_HR * NS) {                    // strip = (halo row, IL_HS columns)
        const int ry = threadIdx.x / NS, c0 = (threadIdx.x % NS) * IL_HS;
        float xv[NV], yv[NV], v[NV];
#pragma unroll
        for (int k = 0; k < NV; k++) { xv[k] = s_x[ry][c0 + k]; yv[k] = s_y[ry][c0 + k]; }
        il_hrow(taps, xv, &s_h[0][ry][c0]);
        il_hrow(taps, yv, &s_h[1][ry][c0]);
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] = xv[k] * xv[k];
        il_hrow(taps, v, &s_h[2][ry][c0]);
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] = yv[k] * yv[k];
        il_hrow(taps, v, &s_h[3][ry][c0]);
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] = xv[k] * yv[k];
        il_hrow(taps, v, &s_h[4][ry][c0]);
    }
    __syncthreads();
    float acc_s = 0.f, acc_l = 0.f, acc_q = 0.f;
    {                                                   // strip = (column, IL_VS rows)
        const int rx = threadIdx.x & 63, r0 = (threadIdx.x >> 6) * IL_VS;
        float m1[IL_VS], m2[IL_VS], q11[IL_VS], q22[IL_VS], q12[IL_VS];
        il_vcol(taps, &s_h[0][r0][rx], m1);
        il_vcol(taps, &s_h[1][r0][rx], m2);
        il_vcol(taps, &s_h[2][r0][rx], q11);
        il_vcol(taps, &s_h[3][r0][rx], q22);
        il_vcol(taps, &s_h[4][r0][rx], q12);
        const int x = x0 + rx;
        const size_t plane = mask ? (mask_channels == 1 ? blockIdx.z / channels : blockIdx.z) : 0;
#pragma unroll
        for (int r = 0; r < IL_VS; r++) {
            const int y = y0 + r0 + r;
            if (y < H && x < W) {
                const float mu1 = m1[r], mu2 = m2[r], s11 = q11[r], s22 = q22[r], s12 = q12[r];
                const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
                const float A1 = 2.f * mu12 + SSIM_C1, A2 = 2.f * (s12 - mu12) + SSIM_C2;
                const float B1 = mu1_sq + mu2_sq + SSIM_C1, B2 = (s11 - mu1_sq) + (s22 - mu2_sq) + SSIM_C2;
                const float inv = 1.f / (B1 * B2);
                const float S = A1 * A2 * inv;
                const size_t o = img + (size_t)y * W + x;
                float m = 1.f;
                if (mask) {
                    m = mask[(plane * H + y) * W + x];
                    acc_s += (1.f - S) * m;
                } else acc_s += S;
                const float d0 = s_x[r0 + r + R5][rx + R5] - s_y[r0 + r + R5][rx + R5];
                const float d = d0 * m;                               // (utils/loss_utils.py:21-22: |(x - y) m|; the PSNR is unmasked)
                acc_l += fabsf(d);
                acc_q += d0 * d0;
                if (sign8) sign8[o] = (signed char)(d > 0.f ? 1 : (d < 0.f ? -1 : 0));
                if (P1) {
                    P1[o] = m * (2.f * mu2 * (A2 - A1) * inv - S * 2.f * mu1 * (B2 - B1) * inv);
                    P2[o] = m * (-S / B2);
                    P3[o] = m * (2.f * A1 * inv);
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { acc_s += __shfl_xor(acc_s, o, 64); acc_l += __shfl_xor(acc_l, o, 64); acc_q += __shfl_xor(acc_q, o, 64); }
    if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = acc_s; s_red[1][threadIdx.x >> 6] = acc_l; s_red[2][threadIdx.x >> 6] = acc_q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const size_t nwg = (size_t)gridDim.y * gridDim.x * gridDim.z;
        const size_t me = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        for (int q = 0; q < 3; q++) {
            float t_ = 0.f;
            for (int k = 0; k < IL_T / 64; k++) t_ += s_red[q][k];
            partial[q * nwg + me] = t_;
        }
    }
}

__global__ __launch_bounds__(IL_T, 6) void k_image_loss_bwd(int H, int W, Taps taps, const float *__restrict__ X, const float *__restrict__ Y,
                                                           const float *__restrict__ P1, const float *__restrict__ P2,
                                                           const float *__restrict__ P3, const signed char *__restrict__ sign8,
                                                           const float *__restrict__ mask, int mask_channels, int channels,
                                                           const float *__restrict__ gscalar, float g_l1, float g_ssim, float *__restrict__ dX) {
    __shared__ __attribute__((aligned(16))) float s_p[3][IL_HR][IL_XP];
    __shared__ __attribute__((aligned(16))) float s_h[3][IL_HR][IL_HP];
    const size_t img = (size_t)blockIdx.z * H * W;
    const int x0 = blockIdx.x * BW, y0 = blockIdx.y * BH;
    for (int t = threadIdx.x; t < IL_HR * (BW + 2 * R5); t += IL_T) {
        const int ry = t / (BW + 2 * R5), rx = t - ry * (BW + 2 * R5);
        const int y = y0 + ry - R5, x = x0 + rx - R5;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        const size_t o = img + (size_t)y * W + x;
        s_p[0][ry][rx] = in ? P1[o] : 0.f; s_p[1][ry][rx] = in ? P2[o] : 0.f; s_p[2][ry][rx] = in ? P3[o] : 0.f;
    }
    __syncthreads();
    constexpr int NS = BW / IL_HS, NV = IL_HS + 10;
    if (threadIdx.x < IL_HR * NS) {
        const int ry = threadIdx.x / NS, c0 = (threadIdx.x % NS) * IL_HS;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            float v[NV];
#pragma unroll
            for (int k = 0; k < NV; k++) v[k] = s_p[q][ry][c0 + k];
            il_hrow(taps, v, &s_h[q][ry][c0]);
        }
    }
    __syncthreads();
    const float gs = gscalar[0] * g_ssim, gl = gscalar[0] * g_l1;
    const int rx = threadIdx.x & 63, r0 = (threadIdx.x >> 6) * IL_VS;
    float b1[IL_VS], b2[IL_VS], b3[IL_VS];
    il_vcol(taps, &s_h[0][r0][rx], b1);
    il_vcol(taps, &s_h[1][r0][rx], b2);
    il_vcol(taps, &s_h[2][r0][rx], b3);
    const int x = x0 + rx;
    const size_t plane = mask ? (mask_channels == 1 ? blockIdx.z / channels : blockIdx.z) : 0;
#pragma unroll
    for (int r = 0; r < IL_VS; r++) {
        const int y = y0 + r0 + r;
        if (y < H && x < W) {
            const size_t o = img + (size_t)y * W + x;
            const float m = mask ? mask[(plane * H + y) * W + x] : 1.f;
            dX[o] = gs * (b1[r] + 2.f * X[o] * b2[r] + Y[o] * b3[r]) + gl * (float)sign8[o] * m;
        }
    }
}

// the second (one-workgroup) launch of the image loss: sums the workgroup partials in index order.  NOT a ticket in the kernel above:
// a device-scope release on gfx950 writes the XCD's dirty L2 lines back, and ~6000 workgroups each fencing while all of them stream the
// three partial-derivative images out cost 4x the kernel itself (measured 268-351 us against 75 + 4 for the two launches).
__global__ __launch_bounds__(SSIM_THREADS) void k_image_loss_finish(size_t per_plane, int planes, int channels, int H, int W, int masked, float lam,
                                                            float img_weight, const float *__restrict__ add, float add_weight,
                                                            float psnr_scale, const float *__restrict__ partial, float *__restrict__ out) {
    __shared__ float s_red[SSIM_THREADS / 64];
    const size_t nwg = per_plane * planes;
    // (ONE workgroup pulls ~135 KB of partials: 16-byte loads, four of them in flight per thread -- with 4-byte loads in a dependent
    //  `t += p[i]` loop the five sums of a 3-camera step took 12.8 us)
    auto block_sum = [&](const float *p, size_t lo, size_t hi) -> float {     // fixed-order sum of p[lo, hi)
        float t = 0.f;
        size_t a0 = lo + (((16u - (unsigned)((uintptr_t)(p + lo) & 15u)) & 15u) >> 2);      // first 16-byte-aligned element of p[lo ..
        if (a0 > hi) a0 = hi;
        const size_t n4 = (hi - a0) >> 2;
        const float4 *p4 = reinterpret_cast<const float4 *>(p + a0);
        size_t i = threadIdx.x;
        for (; i + 3 * SSIM_THREADS < n4; i += 4 * SSIM_THREADS) {
            const float4 u0 = p4[i], u1 = p4[i + SSIM_THREADS], u2 = p4[i + 2 * SSIM_THREADS], u3 = p4[i + 3 * SSIM_THREADS];
            t += ((u0.x + u0.y) + (u0.z + u0.w)) + ((u1.x + u1.y) + (u1.z + u1.w));
            t += ((u2.x + u2.y) + (u2.z + u2.w)) + ((u3.x + u3.y) + (u3.z + u3.w));
        }
        for (; i < n4; i += SSIM_THREADS) { const float4 u = p4[i]; t += (u.x + u.y) + (u.z + u.w); }
        for (size_t j = lo + threadIdx.x; j < a0; j += SSIM_THREADS) t += p[j];                    // <= 3 elements in front ...
        for (size_t j = a0 + (n4 << 2) + threadIdx.x; j < hi; j += SSIM_THREADS) t += p[j];        // ... and behind the float4 body
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = t;
        __syncthreads();
        float r = 0.f;
        for (int k = 0; k < SSIM_THREADS / 64; k++) r += s_red[k];
        return r;
    };
    const float ssim_sum = block_sum(partial, 0, nwg);
    const float l1_sum = block_sum(partial + nwg, 0, nwg);
    const int n_batch = planes / channels;
    float psnr_sum = 0.f;
    for (int b = 0; b < n_batch; b++) {
        const float q = block_sum(partial + 2 * nwg, (size_t)b * channels * per_plane, (size_t)(b + 1) * channels * per_plane);
        const float mse = q / ((float)channels * (float)H * (float)W);
        psnr_sum += 20.f * log10f(1.f / sqrtf(mse));
    }
    if (threadIdx.x == 0) {
        const float n = (float)planes * (float)H * (float)W;
        const float l1 = l1_sum / n;
        const float image_loss = masked ? l1 + lam * (ssim_sum / n) : l1 + lam * (1.f - ssim_sum / n);
        out[0] = img_weight * image_loss + (add ? add_weight * add[0] : 0.f);
        out[1] = psnr_scale * psnr_sum;
        out[2] = image_loss;
        out[3] = l1;
    }
}

// ---- fused L1 loss: mean |a - b| and its gradient sign(a - b) / n in ONE pass (utils/loss_utils.py:20-23 is three
// elementwise launches forward and three backward).  Deterministic: workgroup partials are summed in index order by
// whichever workgroup finishes last (ticket counter), not with float atomics.
constexpr int L1_BLOCKS = 256, L1_THREADS = 1024;  // one fat workgroup per CU; their partials are summed by k_l1_finish
__global__ __launch_bounds__(L1_THREADS) void k_l1(int64_t n, const float *__restrict__ a, const float *__restrict__ b,
                                                    float inv_n, float *__restrict__ partial, float *__restrict__ grad,
                                                    const float *__restrict__ mask, int64_t hw, int channels, int mask_channels,
                                                    signed char *__restrict__ sign8) {
    __shared__ float s_red[L1_THREADS / 64];
    float acc = 0.f;
    auto sg = [&](float d) { return d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f); };
    auto s8 = [&](float d) { return (signed char)(d > 0.f ? 1 : (d < 0.f ? -1 : 0)); };   // the backward's input when the caller asks for 1 byte per element
    // masked form (utils/loss_utils.py:21-22): mean |(a - b) * m|, m one plane per image (mask_channels == 1) or per channel
    auto moff = [&](int64_t e) {
        const int64_t plane = e / hw;
        return (mask_channels == 1 ? plane / channels : plane) * hw + (e - plane * hw);
    };
    const int64_t n4 = (mask && (hw & 3)) ? 0 : n >> 2;    // a 4-pixel group stays inside one image plane
    auto one = [&](int64_t i, const float4 x, const float4 y) {
        float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
        if (mask) {
            const float4 m = *reinterpret_cast<const float4 *>(mask + moff(i << 2));
            d0 *= m.x; d1 *= m.y; d2 *= m.z; d3 *= m.w;
            acc += (fabsf(d0) + fabsf(d1)) + (fabsf(d2) + fabsf(d3));
            if (grad) reinterpret_cast<float4 *>(grad)[i] = make_float4(sg(d0) * m.x, sg(d1) * m.y, sg(d2) * m.z, sg(d3) * m.w);
        } else {
            acc += (fabsf(d0) + fabsf(d1)) + (fabsf(d2) + fabsf(d3));
            if (grad) reinterpret_cast<float4 *>(grad)[i] = make_float4(sg(d0), sg(d1), sg(d2), sg(d3));
        }
        if (sign8) reinterpret_cast<char4 *>(sign8)[i] = make_char4(s8(d0), s8(d1), s8(d2), s8(d3));
    };
    const int64_t stride = (int64_t)gridDim.x * L1_THREADS;
    int64_t i = (int64_t)blockIdx.x * L1_THREADS + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {   // four independent pairs of 16-byte loads in flight (two: 24.6 us for 4 x 3 x 800^2)
        const float4 x0 = reinterpret_cast<const float4 *>(a)[i], y0 = reinterpret_cast<const float4 *>(b)[i];
        const float4 x1 = reinterpret_cast<const float4 *>(a)[i + stride], y1 = reinterpret_cast<const float4 *>(b)[i + stride];
        const float4 x2 = reinterpret_cast<const float4 *>(a)[i + 2 * stride], y2 = reinterpret_cast<const float4 *>(b)[i + 2 * stride];
        const float4 x3 = reinterpret_cast<const float4 *>(a)[i + 3 * stride], y3 = reinterpret_cast<const float4 *>(b)[i + 3 * stride];
        one(i, x0, y0);
        one(i + stride, x1, y1);
        one(i + 2 * stride, x2, y2);
        one(i + 3 * stride, x3, y3);
    }
    for (; i + stride < n4; i += 2 * stride) {   // two independent pairs of 16-byte loads in flight
        const float4 x0 = reinterpret_cast<const float4 *>(a)[i], y0 = reinterpret_cast<const float4 *>(b)[i];
        const float4 x1 = reinterpret_cast<const float4 *>(a)[i + stride], y1 = reinterpret_cast<const float4 *>(b)[i + stride];
        one(i, x0, y0);
        one(i + stride, x1, y1);
    }
    if (i < n4) one(i, reinterpret_cast<const float4 *>(a)[i], reinterpret_cast<const float4 *>(b)[i]);
    // tail: n not a multiple of 4 (<= 3 elements, all on workgroup 0), or every element when the planes are not 4-aligned
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * L1_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * L1_THREADS) {
        float d = a[i] - b[i];
        float m = 1.f;
        if (mask) { m = mask[moff(i)]; d *= m; }
        acc += fabsf(d);
        if (grad) grad[i] = sg(d) * m;
        if (sign8) sign8[i] = s8(d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    // one partial per workgroup; k_l1_finish (a second, one-workgroup launch) sums them in index order.  Rounds 1-4 had the LAST workgroup
    // to arrive do that behind a ticket: a device-scope release per workgroup, and on gfx950 a release writes the XCD's dirty L2 lines back
    // -- here the sign bytes this kernel has just stored and the image the compositing kernel wrote in front of it -- which made ~10 us of
    // fixed cost on a kernel that streams its 69 MB in ~14 us (24.8 us for four views, 15 us for ONE view: profiles/r04g, r05g).
    if (threadIdx.x == 0) { float t_ = 0.f; for (int k_ = 0; k_ < L1_THREADS / 64; k_++) t_ += s_red[k_]; partial[blockIdx.x] = t_; }
}
__global__ __launch_bounds__(L1_BLOCKS) void k_l1_finish(int nparts, const float *__restrict__ partial, float inv_n, float *__restrict__ loss) {
    __shared__ float s[L1_BLOCKS];
    s[threadIdx.x] = (int)threadIdx.x < nparts ? partial[threadIdx.x] : 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {       // groups of 16 in index order, then the group sums in index order: a fixed tree
        float tt = 0.f;
        for (int g = 0; g < L1_BLOCKS / 16; g++) {
            float t = 0.f;
            for (int k = 0; k < 16; k++) t += s[16 * g + k];
            tt += t;
        }
        *loss = tt * inv_n;
    }
}
// ---- PSNR per image (utils/image_utils.py psnr: mse over all channels and pixels of an image, 20 log10(1 / sqrt(mse))).
// The reference logs it every training step; as torch ops it is ~10 launches.  gridDim.y = image, PSNR_BLOCKS slices per
// image, last slice of an image (ticket) sums the slice partials in fixed order.
constexpr int PSNR_BLOCKS = 64;
__global__ __launch_bounds__(1024) void k_psnr(int64_t n, const float *__restrict__ a, const float *__restrict__ b,
                                               float *__restrict__ partial, unsigned *__restrict__ ticket, float *__restrict__ out) {
    __shared__ float s_red[16];
    __shared__ bool s_last;
    const float *pa = a + (size_t)blockIdx.y * n, *pb = b + (size_t)blockIdx.y * n;
    float acc = 0.f;
    if ((((uintptr_t)pa | (uintptr_t)pb) & 15u) == 0) {
        const int64_t n4 = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 1024) {
            const float4 x = reinterpret_cast<const float4 *>(pa)[i], y = reinterpret_cast<const float4 *>(pb)[i];
            const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
            acc += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
        if (blockIdx.x == 0)
            for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 1024) { const float d = pa[i] - pb[i]; acc += d * d; }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 1024) {
            const float d = pa[i] - pb[i];
            acc += d * d;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < 16; k++) t += s_red[k];
        partial[blockIdx.y * PSNR_BLOCKS + blockIdx.x] = t;
        __threadfence();
        s_last = atomicAdd(ticket + blockIdx.y, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (s_last && threadIdx.x == 0) {
        __threadfence();
        float t = 0.f;
        for (unsigned k = 0; k < gridDim.x; k++) t += __builtin_nontemporal_load(partial + blockIdx.y * PSNR_BLOCKS + k);
        const float mse = t / (float)n;
        out[blockIdx.y] = 20.f * log10f(1.f / sqrtf(mse));
        ticket[blockIdx.y] = 0u;
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

extern "C" size_t csplat_l1_scratch_bytes(void) { return (size_t)(L1_BLOCKS + 1) * 4; }   // the workgroup partials

// backward of the fused L1 when the forward kept one BYTE per element (csplat_l1_signs): out[i] = g[0] * sign[i] * m[i] / n -- the
// upstream gradient g is a device scalar, so the reference's three elementwise backward launches (and the multiply by the incoming
// gradient that a stored float gradient image needs) are this one pass: 1 (+4 masked) bytes read, 4 written per element
__global__ __launch_bounds__(256) void k_l1_bwd(int64_t n, const signed char *__restrict__ sign8, const float *__restrict__ g, float inv_n,
                                                const float *__restrict__ mask, int64_t hw, int channels, int mask_channels,
                                                float *__restrict__ out) {
    const float sc = g[0] * inv_n;
    auto moff = [&](int64_t e) {
        const int64_t plane = e / hw;
        return (mask_channels == 1 ? plane / channels : plane) * hw + (e - plane * hw);
    };
    const int64_t n4 = (mask && (hw & 3)) ? 0 : n >> 2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const char4 s = reinterpret_cast<const char4 *>(sign8)[i];
        float4 o = make_float4(sc * (float)s.x, sc * (float)s.y, sc * (float)s.z, sc * (float)s.w);
        if (mask) {
            const float4 m = *reinterpret_cast<const float4 *>(mask + moff(i << 2));
            o.x *= m.x; o.y *= m.y; o.z *= m.z; o.w *= m.w;
        }
        reinterpret_cast<float4 *>(out)[i] = o;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        out[i] = sc * (float)sign8[i] * (mask ? mask[moff(i)] : 1.f);
}

static int l1_launch(void *stream, int64_t n, const float *a, const float *b, void *scratch, float *loss, float *grad,
                     const float *mask, int64_t hw, int channels, int mask_channels, const char *who, signed char *sign8 = nullptr) {
    CSPLAT_REQUIRE(n > 0 && a && b && scratch && loss, "csplat_l1: bad arguments");
    CSPLAT_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)grad | (uintptr_t)mask) & 15u) == 0, "csplat_l1: operands must be 16-byte aligned");
    float *partial = (float *)scratch;
    const int64_t units = (mask && (hw & 3)) ? n : n / 4;
    const int64_t work = (units + L1_THREADS - 1) / L1_THREADS;
    const int grid = (int)(work < 1 ? 1 : (work > L1_BLOCKS ? L1_BLOCKS : work));
    CSPLAT_REQUIRE(((uintptr_t)sign8 & 3u) == 0, "csplat_l1: the sign buffer must be 4-byte aligned");
    k_l1<<<grid, L1_THREADS, 0, (hipStream_t)stream>>>(n, a, b, 1.0f / (float)n, partial, grad, mask, hw, channels, mask_channels, sign8);
    LAUNCH_CHECK();
    k_l1_finish<<<1, L1_BLOCKS, 0, (hipStream_t)stream>>>(grid, partial, 1.0f / (float)n, loss);
    LAUNCH_CHECK();
    (void)who;
    return 0;
}

extern "C" int csplat_l1(void *stream, int64_t n, const float *a, const float *b, void *scratch, float *loss, float *grad) {
    return l1_launch(stream, n, a, b, scratch, loss, grad, nullptr, 1, 1, 1, "csplat_l1");
}

extern "C" int csplat_l1_masked(void *stream, int64_t n_batch, int channels, int64_t hw, const float *a, const float *b,
                                const float *mask, int mask_channels, void *scratch, float *loss, float *grad) {
    CSPLAT_REQUIRE(n_batch > 0 && channels > 0 && hw > 0 && mask, "csplat_l1_masked: bad arguments");
    CSPLAT_REQUIRE(mask_channels == 1 || mask_channels == channels, "csplat_l1_masked: the mask has 1 plane per image or one per channel");
    return l1_launch(stream, n_batch * channels * hw, a, b, scratch, loss, grad, mask, hw, channels, mask_channels, "csplat_l1_masked");
}

extern "C" int csplat_l1_signs(void *stream, int64_t n_batch, int channels, int64_t hw, const float *a, const float *b, const float *mask,
                               int mask_channels, void *scratch, float *loss, signed char *sign8) {
    CSPLAT_REQUIRE(n_batch > 0 && channels > 0 && hw > 0 && sign8, "csplat_l1_signs: bad arguments");
    CSPLAT_REQUIRE(!mask || mask_channels == 1 || mask_channels == channels, "csplat_l1_signs: the mask has 1 plane per image or one per channel");
    return l1_launch(stream, n_batch * channels * hw, a, b, scratch, loss, nullptr, mask, hw, channels, mask ? mask_channels : 1, "csplat_l1_signs", sign8);
}
extern "C" int csplat_l1_signs_bwd(void *stream, int64_t n_batch, int channels, int64_t hw, const signed char *sign8, const float *mask,
                                   int mask_channels, const float *g_scalar, float *out) {
    CSPLAT_REQUIRE(n_batch > 0 && channels > 0 && hw > 0 && sign8 && g_scalar && out, "csplat_l1_signs_bwd: bad arguments");
    CSPLAT_REQUIRE((((uintptr_t)out | (uintptr_t)mask) & 15u) == 0 && ((uintptr_t)sign8 & 3u) == 0, "csplat_l1_signs_bwd: operands must be aligned");
    const int64_t n = n_batch * channels * hw;
    const int64_t work = (n / 4 + 255) / 256;
    const int grid = (int)(work < 1 ? 1 : (work > 4096 ? 4096 : work));
    k_l1_bwd<<<grid, 256, 0, (hipStream_t)stream>>>(n, sign8, g_scalar, 1.0f / (float)n, mask, hw, channels, mask ? mask_channels : 1, out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" size_t csplat_ssim_partial_count(int64_t n_images, int H, int W) { return (size_t)n_images * cdiv(H, BH) * cdiv(W, BW); }

static int ssim_fwd_launch(void *stream, int64_t n_images, int H, int W, const float *taps11, const float *x, const float *y,
                           float *p1, float *p2, float *p3, float *map_out, float *partial, const float *mask, int mask_channels,
                           int channels) {
    CSPLAT_REQUIRE(n_images >= 0 && n_images < 65536 && H > 0 && W > 0 && taps11 && x && y && partial, "csplat_ssim_fwd: bad arguments");
    CSPLAT_REQUIRE((p1 != nullptr) == (p2 != nullptr) && (p1 != nullptr) == (p3 != nullptr), "csplat_ssim_fwd: all three partials or none");
    if (n_images == 0) return 0;
    Taps t;
    memcpy(t.w, taps11, sizeof(t.w));
    dim3 grid(cdiv(W, BW), cdiv(H, BH), (unsigned)n_images);
    k_ssim_fwd<<<grid, SSIM_THREADS, 0, (hipStream_t)stream>>>(H, W, t, x, y, p1, p2, p3, map_out, partial, mask, mask_channels, channels);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int csplat_ssim_fwd(void *stream, int64_t n_images, int H, int W, const float *taps11, const float *x, const float *y,
                               float *p1, float *p2, float *p3, float *map_out, float *partial) {
    return ssim_fwd_launch(stream, n_images, H, W, taps11, x, y, p1, p2, p3, map_out, partial, nullptr, 1, 1);
}

// masked form: partial sums hold sum((1 - S) * m), the three partial-derivative images are pre-multiplied by m, so that
// csplat_ssim_bwd serves both forms unchanged.  n_images = batch * channels planes; mask [batch, mask_channels, H, W].
extern "C" int csplat_ssim_fwd_masked(void *stream, int64_t n_batch, int channels, int H, int W, const float *taps11, const float *x,
                                      const float *y, const float *mask, int mask_channels, float *p1, float *p2, float *p3,
                                      float *map_out, float *partial) {
    CSPLAT_REQUIRE(n_batch >= 0 && channels > 0 && mask, "csplat_ssim_fwd_masked: bad arguments");
    CSPLAT_REQUIRE(mask_channels == 1 || mask_channels == channels, "csplat_ssim_fwd_masked: the mask has 1 plane per image or one per channel");
    return ssim_fwd_launch(stream, n_batch * channels, H, W, taps11, x, y, p1, p2, p3, map_out, partial, mask, mask_channels, channels);
}

extern "C" int csplat_ssim_bwd(void *stream, int64_t n_images, int H, int W, const float *taps11, const float *x, const float *y,
                               const float *p1, const float *p2, const float *p3, const float *g_scalar, float inv_n,
                               const float *addend, const float *add_scale, float *dx) {
    CSPLAT_REQUIRE(n_images >= 0 && n_images < 65536 && H > 0 && W > 0 && taps11 && x && y && p1 && p2 && p3 && g_scalar && dx,
                   "csplat_ssim_bwd: bad arguments");
    if (n_images == 0) return 0;
    Taps t;
    memcpy(t.w, taps11, sizeof(t.w));
    dim3 grid(cdiv(W, BW), cdiv(H, BH), (unsigned)n_images);
    CSPLAT_REQUIRE((addend == nullptr) == (add_scale == nullptr), "csplat_ssim_bwd: addend and add_scale go together");
    k_ssim_bwd<<<grid, SSIM_THREADS, 0, (hipStream_t)stream>>>(H, W, t, x, y, p1, p2, p3, g_scalar, inv_n, addend, add_scale, dx);
    LAUNCH_CHECK();
    return 0;
}

// image loss of a train step (k_image_loss_fwd + k_image_loss_finish / k_image_loss_bwd).  out[4] = {img_weight * image_loss +
// add_weight * add[0], psnr_scale * sum_b PSNR_b, image_loss, Ll1}; scratch: csplat_image_loss_scratch_bytes (no initial state).
// p1..p3 / sign8 may be NULL when no gradient will be asked for.
extern "C" size_t csplat_image_loss_scratch_bytes(int64_t n_batch, int channels, int H, int W) {
    const size_t planes = (size_t)(n_batch > 0 ? n_batch : 1) * channels;
    return align256(3 * planes * cdiv(H, BH) * cdiv(W, BW) * 4);      // [3][workgroups] partial sums
}
extern "C" int csplat_image_loss_fwd(void *stream, int64_t n_batch, int channels, int H, int W, const float *taps11, const float *x,
                                     const float *y, const float *mask, int mask_channels, float lam, float img_weight, const float *add,
                                     float add_weight, float psnr_scale, float *p1, float *p2, float *p3, signed char *sign8, void *scratch,
                                     float *out) {
    CSPLAT_REQUIRE(n_batch > 0 && channels > 0 && n_batch * channels < 65536 && H > 0 && W > 0 && taps11 && x && y && scratch && out,
                   "csplat_image_loss_fwd: bad arguments");
    CSPLAT_REQUIRE((p1 != nullptr) == (p2 != nullptr) && (p1 != nullptr) == (p3 != nullptr) && (p1 != nullptr) == (sign8 != nullptr),
                   "csplat_image_loss_fwd: the three partials and the sign bytes go together");
    CSPLAT_REQUIRE(!mask || mask_channels == 1 || mask_channels == channels, "csplat_image_loss_fwd: the mask has 1 plane per image or one per channel");
    Taps t;
    memcpy(t.w, taps11, sizeof(t.w));
    dim3 grid(cdiv(W, BW), cdiv(H, BH), (unsigned)(n_batch * channels));
    float *partial = (float *)scratch;
    k_image_loss_fwd<<<grid, IL_T, 0, (hipStream_t)stream>>>(H, W, t, x, y, p1, p2, p3, sign8, mask, mask ? mask_channels : 1, channels, partial);
    LAUNCH_CHECK();
    k_image_loss_finish<<<1, SSIM_THREADS, 0, (hipStream_t)stream>>>((size_t)grid.x * grid.y, (int)grid.z, channels, H, W, mask ? 1 : 0, lam, img_weight,
                                                                     add, add_weight, psnr_scale, partial, out);
    LAUNCH_CHECK();
    return 0;
}
extern "C" int csplat_image_loss_bwd(void *stream, int64_t n_batch, int channels, int H, int W, const float *taps11, const float *x,
                                     const float *y, const float *p1, const float *p2, const float *p3, const signed char *sign8,
                                     const float *mask, int mask_channels, float lam, float img_weight, const float *g_scalar, float *dx) {
    CSPLAT_REQUIRE(n_batch > 0 && channels > 0 && n_batch * channels < 65536 && H > 0 && W > 0 && taps11 && x && y && p1 && p2 && p3 &&
                   sign8 && g_scalar && dx, "csplat_image_loss_bwd: bad arguments");
    Taps t;
    memcpy(t.w, taps11, sizeof(t.w));
    dim3 grid(cdiv(W, BW), cdiv(H, BH), (unsigned)(n_batch * channels));
    const float inv_n = 1.0f / ((float)(n_batch * channels) * (float)H * (float)W);
    k_image_loss_bwd<<<grid, IL_T, 0, (hipStream_t)stream>>>(H, W, t, x, y, p1, p2, p3, sign8, mask, mask ? mask_channels : 1, channels, g_scalar,
                                                             img_weight * inv_n, -lam * img_weight * inv_n, dx);
    LAUNCH_CHECK();
    return 0;
}

extern "C" size_t csplat_psnr_scratch_bytes(int64_t n_images) { return (size_t)(n_images > 0 ? n_images : 1) * (PSNR_BLOCKS + 1) * 4; }

extern "C" int csplat_psnr(void *stream, int64_t n_images, int64_t n_per_image, const float *a, const float *b, void *scratch,
                           float *out) {
    CSPLAT_REQUIRE(n_images >= 0 && n_images < 65536 && n_per_image > 0, "csplat_psnr: bad sizes");
    if (n_images == 0) return 0;
    CSPLAT_REQUIRE(a && b && scratch && out, "csplat_psnr: NULL");
    float *partial = (float *)scratch;
    unsigned *ticket = (unsigned *)scratch + n_images * PSNR_BLOCKS;
    HIP_TRY(hipMemsetAsync(ticket, 0, (size_t)n_images * 4, (hipStream_t)stream));
    const int64_t work = (n_per_image / 4 + 1023) / 1024;
    dim3 grid((unsigned)(work < 1 ? 1 : (work > PSNR_BLOCKS ? PSNR_BLOCKS : work)), (unsigned)n_images);
    k_psnr<<<grid, 1024, 0, (hipStream_t)stream>>>(n_per_image, a, b, partial, ticket, out);
    LAUNCH_CHECK();
    return 0;
}
