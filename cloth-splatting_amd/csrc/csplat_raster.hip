// csplat_raster.hip -- the depth-aware differentiable Gaussian rasterizer for gfx950 (MI355X).
//
// Replaces the CUDA extension behind GaussianRasterizer.forward / backward
// (/root/reference/gaussian_renderer/__init__.py:16,76,156-164; backward via
// scene_reconstruction/train_utils.py:288).  Kernel inventory = SURVEY.md 2.1 K1..K8:
//   K1 k_preprocess        per-Gaussian cull / projection / cov3D / cov2D / conic / radius / rect / SH->RGB
//   K2 (csplat_sort.hip)   inclusive scan of tiles_touched
//   K3 k_emit_keys         (tile<<32 | depth bits, id) per touched tile
//   K4 (csplat_sort.hip)   stable radix sort
//   K5 k_tile_ranges       [first,last) per tile; k_seg_plan (256-entry segments of every tile list);
//      k_block_masks       per list entry: which of the tile's sixteen 4x4 pixel blocks it can reach + tile-ordered records
//   K6 k_composite_fwd     front-to-back compositing of RGB + depth: one wavefront per 4x4 block, four survivors per step
//   K7 k_composite_bwd     per (segment, quadrant) workgroup, forward-ordered replay from the checkpoints, factored moment reduction
//                          reduction, LDS records, one atomic per (entry, quadrant)
//   K8 k_preprocess_bwd    conic->cov2D->cov3D/mean, mean2D(NDC)->mean3D, colour->SH, cov3D->(scale,quat)
//
// Index-deciding arithmetic (radius, tile rectangle, sort key) is compiled with FP contraction OFF and is
// written in the same association order as oracle/raster_ref.c, so tile/bin indices are bit-exact.
#include "csplat_common.h"

#include <atomic>
#include <chrono>
#include <mutex>

namespace {

constexpr float NEAR_Z = 0.2f;

__device__ constexpr float SH_C0 = 0.28209479177387814f;
__device__ constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                       -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                       0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                       -0.5900435899266435f};

struct Geom {
    float *depth;           // [P]
    float2 *xy;             // [P]
    float4 *conic_opacity;  // [P]
    float *rgb;             // [P][3]
    float *cov3D;           // [P][6]
    uint32_t *clamped;      // [P] bit c
    uint32_t *tiles_touched;// [P]
    uint32_t *offsets;      // [P] inclusive scan
    float *cut2;            // [P] squared cut-off distance for wave-level culling (see box_hit)
    void *scan_tmp;
    float4 *pack;           // [P][3] (x, y, conic a, conic b | conic c, opacity, r, g | b, depth, cut2, 0): what k_block_masks needs of a
                            // Gaussian in ONE 48-byte record -- it visits the Gaussians in list order (a random gather per field otherwise)
};

struct Cam {
    const float *view;    // device, 16 floats (transposed world->view)
    const float *proj;    // device, 16 floats (transposed full projection)
    const float *campos;  // device, 3 floats
    float tanfovx, tanfovy, fx, fy;
    int W, H, gx, gy;
};

struct ProjJac {
    float t0[3], t1[3];
    float tx, ty, tz;
    bool x_in, y_in;
};

__device__ __forceinline__ void quat_to_rot(const float *q, float R[3][3]) {
#pragma clang fp contract(off)
    float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = 1.f - 2.f * (y * y + z * z);
    R[0][1] = 2.f * (x * y - r * z);
    R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z);
    R[1][1] = 1.f - 2.f * (x * x + z * z);
    R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y);
    R[2][1] = 2.f * (y * z + r * x);
    R[2][2] = 1.f - 2.f * (x * x + y * y);
}

__device__ __forceinline__ void cov3d_from_scale_rot(const float *scale, float mod, const float *q, float *c6) {
#pragma clang fp contract(off)
    float R[3][3], m[3][3];
    quat_to_rot(q, R);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float s = mod * scale[k];
#pragma unroll
        for (int i = 0; i < 3; i++) m[k][i] = s * R[i][k];
    }
    c6[0] = m[0][0] * m[0][0] + m[1][0] * m[1][0] + m[2][0] * m[2][0];
    c6[1] = m[0][0] * m[0][1] + m[1][0] * m[1][1] + m[2][0] * m[2][1];
    c6[2] = m[0][0] * m[0][2] + m[1][0] * m[1][2] + m[2][0] * m[2][2];
    c6[3] = m[0][1] * m[0][1] + m[1][1] * m[1][1] + m[2][1] * m[2][1];
    c6[4] = m[0][1] * m[0][2] + m[1][1] * m[1][2] + m[2][1] * m[2][2];
    c6[5] = m[0][2] * m[0][2] + m[1][2] * m[1][2] + m[2][2] * m[2][2];
}

__device__ __forceinline__ void view_point(const float *p, const float *V, float *o) {
#pragma clang fp contract(off)
    o[0] = V[0] * p[0] + V[4] * p[1] + V[8] * p[2] + V[12];
    o[1] = V[1] * p[0] + V[5] * p[1] + V[9] * p[2] + V[13];
    o[2] = V[2] * p[0] + V[6] * p[1] + V[10] * p[2] + V[14];
}

__device__ __forceinline__ void proj_jacobian(const float *pv, const Cam &c, ProjJac &o) {
#pragma clang fp contract(off)
    const float limx = 1.3f * c.tanfovx, limy = 1.3f * c.tanfovy;
    const float tz = pv[2];
    const float txtz = pv[0] / tz, tytz = pv[1] / tz;
    o.x_in = !(txtz < -limx || txtz > limx);
    o.y_in = !(tytz < -limy || tytz > limy);
    const float tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
    const float ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
    const float J00 = c.fx / tz, J02 = -(c.fx * tx) / (tz * tz);
    const float J11 = c.fy / tz, J12 = -(c.fy * ty) / (tz * tz);
#pragma unroll
    for (int a = 0; a < 3; a++) {
        o.t0[a] = c.view[4 * a + 0] * J00 + c.view[4 * a + 2] * J02;
        o.t1[a] = c.view[4 * a + 1] * J11 + c.view[4 * a + 2] * J12;
    }
    o.tx = tx; o.ty = ty; o.tz = tz;
}

__device__ __forceinline__ void cov2d_from_cov3d(const float *c6, const ProjJac &pj, float &a, float &b, float &c) {
#pragma clang fp contract(off)
    const float *t0 = pj.t0, *t1 = pj.t1;
    const float Vm[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    float u0[3], u1[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        u0[j] = t0[0] * Vm[0][j] + t0[1] * Vm[1][j] + t0[2] * Vm[2][j];
        u1[j] = t1[0] * Vm[0][j] + t1[1] * Vm[1][j] + t1[2] * Vm[2][j];
    }
    a = (u0[0] * t0[0] + u0[1] * t0[1] + u0[2] * t0[2]) + 0.3f;
    b = u0[0] * t1[0] + u0[1] * t1[1] + u0[2] * t1[2];
    c = (u1[0] * t1[0] + u1[1] * t1[1] + u1[2] * t1[2]) + 0.3f;
}

__device__ __forceinline__ void tile_rect(float px, float py, int rad, const Cam &c, int &minx, int &miny, int &maxx,
                                          int &maxy) {
#pragma clang fp contract(off)
    minx = min(c.gx, max(0, (int)((px - (float)rad) / (float)CSPLAT_TILE)));
    miny = min(c.gy, max(0, (int)((py - (float)rad) / (float)CSPLAT_TILE)));
    maxx = min(c.gx, max(0, (int)((px + (float)rad + (float)(CSPLAT_TILE - 1)) / (float)CSPLAT_TILE)));
    maxy = min(c.gy, max(0, (int)((py + (float)rad + (float)(CSPLAT_TILE - 1)) / (float)CSPLAT_TILE)));
}

// SH coefficients are 48 floats (192 B) per Gaussian: read lane-per-Gaussian that is a 192-byte stride.  With STAGE the
// workgroup first copies its 256 x 48 contiguous floats into LDS with 16-byte coalesced loads (row stride 49 floats:
// conflict-free column reads) and the per-Gaussian code reads LDS instead.
constexpr int SH_ROW = 49;

template <int NT>
__device__ __forceinline__ void stage_sh_rows(const float *__restrict__ src, int rows, float *s_rows) {
    const float4 *src4 = reinterpret_cast<const float4 *>(src);
    for (int t = threadIdx.x; t < rows * 12; t += NT) {
        const float4 v = src4[t];
        const int row = t / 12, c = (t - row * 12) * 4;
        float *d = s_rows + row * SH_ROW + c;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}

// ------------------------------------------------------------------------------------------- K1
// s_shrows: the workgroup's SH rows in LDS when STAGE (filled by the caller: once per workgroup, also when it serves
// several views)
template <bool STAGE>
__device__ __forceinline__ void preprocess_body(int P, int D, int M, const float *__restrict__ means3D,
                                                const float *__restrict__ shs,
                                                const float *__restrict__ colors_precomp,
                                                const float *__restrict__ opacities,
                                                const float *__restrict__ scales, float scale_mod,
                                                const float *__restrict__ rotations,
                                                const float *__restrict__ cov3D_precomp, const Cam &cam, const Geom &g,
                                                int32_t *__restrict__ radii, int nocull, const float *s_shrows, int i, int srow) {
#pragma clang fp contract(off)
    if (i >= P) return;
    float depth = 0.f, px = 0.f, py = 0.f, cut = -1.f;
    float4 co = {0.f, 0.f, 0.f, 0.f};
    float rgb[3] = {0.f, 0.f, 0.f};
    float c6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint32_t clampbits = 0, touched = 0;
    int rad = 0;

    const float p[3] = {means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]};
    float pv[3];
    view_point(p, cam.view, pv);
    do {
        if (pv[2] <= NEAR_Z) break;
        const float *pr = cam.proj;
        const float hx = pr[0] * p[0] + pr[4] * p[1] + pr[8] * p[2] + pr[12];
        const float hy = pr[1] * p[0] + pr[5] * p[1] + pr[9] * p[2] + pr[13];
        const float hw = pr[3] * p[0] + pr[7] * p[1] + pr[11] * p[2] + pr[15];
        const float pw = 1.0f / (hw + 0.0000001f);
        const float ndcx = hx * pw, ndcy = hy * pw;
        if (cov3D_precomp) {
#pragma unroll
            for (int k = 0; k < 6; k++) c6[k] = cov3D_precomp[6 * i + k];
        } else {
            const float s[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
            const float q[4] = {rotations[4 * i], rotations[4 * i + 1], rotations[4 * i + 2], rotations[4 * i + 3]};
            cov3d_from_scale_rot(s, scale_mod, q, c6);
        }
        ProjJac pj;
        proj_jacobian(pv, cam, pj);
        float a, b, c;
        cov2d_from_cov3d(c6, pj, a, b, c);
        const float det = a * c - b * b;
        if (det == 0.0f) break;
        const float det_inv = 1.f / det;
        const float mid = 0.5f * (a + c);
        const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
        const float lam1 = mid + sq, lam2 = mid - sq;
        const float my_radius = ceilf(3.f * sqrtf(fmaxf(lam1, lam2)));
        const float ix = ((ndcx + 1.0f) * (float)cam.W - 1.0f) * 0.5f;
        const float iy = ((ndcy + 1.0f) * (float)cam.H - 1.0f) * 0.5f;
        const int r = (int)my_radius;
        int minx, miny, maxx, maxy;
        tile_rect(ix, iy, r, cam, minx, miny, maxx, maxy);
        if ((maxx - minx) * (maxy - miny) == 0) break;

        if (colors_precomp) {
#pragma unroll
            for (int k = 0; k < 3; k++) rgb[k] = colors_precomp[3 * i + k];
        } else {
            const float *sh = STAGE ? (const float *)(s_shrows + srow * SH_ROW) : shs + (size_t)i * M * 3;
            const float d0 = p[0] - cam.campos[0], d1 = p[1] - cam.campos[1], d2 = p[2] - cam.campos[2];
            const float len = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
            const float x = d0 / len, y = d1 / len, z = d2 / len;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
#define S(k) sh[(k) * 3 + ch]
                float res = SH_C0 * S(0);
                if (D > 0) {
                    res = res - SH_C1 * y * S(1) + SH_C1 * z * S(2) - SH_C1 * x * S(3);
                    if (D > 1) {
                        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        res = res + SH_C2[0] * xy * S(4) + SH_C2[1] * yz * S(5) + SH_C2[2] * (2.f * zz - xx - yy) * S(6) +
                              SH_C2[3] * xz * S(7) + SH_C2[4] * (xx - yy) * S(8);
                        if (D > 2) {
                            res = res + SH_C3[0] * y * (3.f * xx - yy) * S(9) + SH_C3[1] * xy * z * S(10) +
                                  SH_C3[2] * y * (4.f * zz - xx - yy) * S(11) +
                                  SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * S(12) +
                                  SH_C3[4] * x * (4.f * zz - xx - yy) * S(13) + SH_C3[5] * z * (xx - yy) * S(14) +
                                  SH_C3[6] * x * (xx - 3.f * yy) * S(15);
                        }
                    }
                }
#undef S
                res += 0.5f;
                if (res < 0.f) clampbits |= (1u << ch);
                rgb[ch] = fmaxf(res, 0.f);
            }
        }
        depth = pv[2];
        rad = r;
        px = ix; py = iy;
        const float op = opacities[i];
        co = make_float4(c * det_inv, -b * det_inv, a * det_inv, op);
        touched = (uint32_t)((maxy - miny) * (maxx - minx));
        // culling radius: alpha >= 1/255 needs d^2 <= 2*lambda_max*ln(255*opacity).  lam1 >= lambda_max (the max(0.1,.)
        // above only enlarges it); the margin covers the rounding of det (cancellation in a*c - b*b scales the stored
        // conic uniformly) and of the per-pixel power evaluation.
        const float cancel = 4e-7f * (a * c + b * b) / det;
        cut = cancel < 0.25f ? 2.f * lam1 * logf(255.f * op) * (1.0001f + 2.f * cancel) + 0.01f : 3.0e38f;
        if (nocull == 1) cut = 3.0e38f;
        if (nocull == 2) cut = cut * 4.f + 4.f;
    } while (0);

    g.depth[i] = depth;
    radii[i] = rad;
    g.xy[i] = make_float2(px, py);
    g.conic_opacity[i] = co;
#pragma unroll
    for (int k = 0; k < 3; k++) g.rgb[3 * i + k] = rgb[k];
#pragma unroll
    for (int k = 0; k < 6; k++) g.cov3D[6 * i + k] = c6[k];
    g.clamped[i] = clampbits;
    g.tiles_touched[i] = touched;
    g.cut2[i] = cut;
    g.pack[3 * (size_t)i] = make_float4(px, py, co.x, co.y);
    g.pack[3 * (size_t)i + 1] = make_float4(co.z, co.w, rgb[0], rgb[1]);
    g.pack[3 * (size_t)i + 2] = make_float4(rgb[2], depth, cut, 0.f);
}

template <bool STAGE>
__global__ __launch_bounds__(256) void k_preprocess(int P, int D, int M, const float *__restrict__ means3D,
                                                     const float *__restrict__ shs,
                                                     const float *__restrict__ colors_precomp,
                                                     const float *__restrict__ opacities,
                                                     const float *__restrict__ scales, float scale_mod,
                                                     const float *__restrict__ rotations,
                                                     const float *__restrict__ cov3D_precomp, Cam cam, Geom g,
                                                     int32_t *__restrict__ radii, int nocull) {
    __shared__ float s_shrows[STAGE ? 256 * SH_ROW : 1];
    if (STAGE) {
        const int base = blockIdx.x * 256;
        stage_sh_rows<256>(shs + (size_t)base * 48, min(256, P - base), s_shrows);
        __syncthreads();
    }
    preprocess_body<STAGE>(P, D, M, means3D, shs, colors_precomp, opacities, scales, scale_mod, rotations, cov3D_precomp, cam, g, radii,
                           nocull, s_shrows, (int)(blockIdx.x * blockDim.x + threadIdx.x), (int)threadIdx.x);
}

// The first phase of the forward (K1 + the three counting kernels) for ALL views of a step, one launch each (blockIdx.y =
// view): a 4-view step otherwise spends 16 launches (~10 us of host time each, the GPU idling in between) before its one
// host read.  Views share the Gaussians' view-independent inputs; means / rotations / cameras / outputs come per view.
constexpr int K1_MAX_VIEWS = 8;
struct K1View {
    const float *means3D, *rotations;
    Cam cam;
    Geom g;
    int32_t *radii;
    uint32_t *table, *info, *mailbox;
    int2 *ranges;
    uint32_t tag;
};
struct K1Table { int n; K1View v[K1_MAX_VIEWS]; };

// (the 192-byte SH row of a Gaussian is staged ONCE per workgroup and evaluated for every view's direction)
constexpr int K1V_G = 64;
template <bool STAGE>
__global__ __launch_bounds__(256) void k_preprocess_views(int P, int D, int M, const float *__restrict__ shs,
                                                           const float *__restrict__ opacities,
                                                           const float *__restrict__ scales, float scale_mod, K1Table tab,
                                                           int nocull) {
    // 64 Gaussians per workgroup, wave w takes the views w, w + 4, ...: the views of a Gaussian run side by side instead of one after the
    // other in one thread (P / 256 = 391 workgroups of four dependent load -> project -> SH rounds each: 1.2 waves per SIMD, 25 us for
    // the four views of the bench), the stores of a wave go to ONE view's arrays at consecutive indices
    __shared__ float s_shrows[STAGE ? K1V_G * SH_ROW : 1];
    const int base = blockIdx.x * K1V_G;
    if (STAGE) {
        stage_sh_rows<256>(shs + (size_t)base * 48, min(K1V_G, P - base), s_shrows);
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    for (int vi = threadIdx.x >> 6; vi < tab.n; vi += 4) {
        const K1View &w = tab.v[vi];
        preprocess_body<STAGE>(P, D, M, w.means3D, shs, nullptr, opacities, scales, scale_mod, w.rotations, nullptr, w.cam, w.g, w.radii, nocull,
                               s_shrows, base + lane, lane);
    }
}

// ------------------------------------------------------------------------------------------- K3
__global__ __launch_bounds__(256) void k_emit_keys(int P, const float2 *__restrict__ xy, const float *__restrict__ depth,
                                                    const uint32_t *__restrict__ offsets,
                                                    const int32_t *__restrict__ radii, Cam cam,
                                                    uint64_t *__restrict__ keys, uint32_t *__restrict__ ids) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int rad = radii[i];
    if (rad <= 0) return;
    uint32_t off = (i == 0) ? 0u : offsets[i - 1];
    const float2 p = xy[i];
    int minx, miny, maxx, maxy;
    tile_rect(p.x, p.y, rad, cam, minx, miny, maxx, maxy);
    const uint32_t dbits = __float_as_uint(depth[i]);
    for (int y = miny; y < maxy; y++)
        for (int x = minx; x < maxx; x++) {
            keys[off] = ((uint64_t)(uint32_t)(y * cam.gx + x) << 32) | dbits;
            ids[off] = (uint32_t)i;
            off++;
        }
}

// ------------------------------------------------------------------------------------------- K5
__global__ __launch_bounds__(256) void k_tile_ranges(int64_t R, const uint64_t *__restrict__ keys, int2 *__restrict__ ranges) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const uint32_t t = (uint32_t)(keys[i] >> 32);
    if (i == 0) ranges[t].x = 0;
    else {
        const uint32_t tp = (uint32_t)(keys[i - 1] >> 32);
        if (tp != t) { ranges[tp].y = (int)i; ranges[t].x = (int)i; }
    }
    if (i == R - 1) ranges[t].y = (int)R;
}

// ---- tile-bucketed binning (default path) ------------------------------------------------------------------------
// One MSD "radix" pass whose digit is the tile id, without any global atomic:
//   k_tile_count   workgroup b (1024 Gaussians) histograms its instances per tile in LDS -> table[b][tile]
//   k_tile_colscan per tile: exclusive scan over the workgroups, in place (each workgroup's offset inside the tile's list)
//   k_tile_scan    exclusive scan of the per-tile totals: tile ranges, R and the longest list
//   k_emit_bucket  workgroup b reloads its bases into LDS and drops every instance at base[tile]++ (LDS atomic)
//   k_tile_sort    each tile's list ordered by (depth bits, Gaussian id) with a stable LSD radix sort in LDS + registers.
//                  The composite key is unique, so the result is exactly the stable (tile | depth) radix order of the
//                  upstream pipeline.
constexpr int BUCKET_CAP = 8192;    // longest tile list the LDS sort takes (64 KB); longer lists -> global radix sort
constexpr int BUCKET_TILES = 12288; // most tiles the per-workgroup LDS histogram takes (48 KB)
constexpr int BUCKET_G = 1024;      // Gaussians per counting workgroup

__device__ __forceinline__ void tile_count_body(int P, int tiles, const float2 *__restrict__ xy,
                                                const int32_t *__restrict__ radii, const Cam &cam,
                                                uint32_t *__restrict__ table) {
    extern __shared__ uint32_t s_hist[];
    for (int t = threadIdx.x; t < tiles; t += BUCKET_G) s_hist[t] = 0u;
    __syncthreads();
    const int i = blockIdx.x * BUCKET_G + threadIdx.x;
    if (i < P) {
        const int rad = radii[i];
        if (rad > 0) {
            const float2 p = xy[i];
            int minx, miny, maxx, maxy;
            tile_rect(p.x, p.y, rad, cam, minx, miny, maxx, maxy);
            for (int y = miny; y < maxy; y++)
                for (int x = minx; x < maxx; x++) atomicAdd(&s_hist[y * cam.gx + x], 1u);
        }
    }
    __syncthreads();
    uint32_t *row = table + (size_t)blockIdx.x * tiles;
    for (int t = threadIdx.x; t < tiles; t += BUCKET_G) row[t] = s_hist[t];
}
__global__ __launch_bounds__(BUCKET_G) void k_tile_count(int P, int tiles, const float2 *__restrict__ xy,
                                                          const int32_t *__restrict__ radii, Cam cam,
                                                          uint32_t *__restrict__ table) {
    tile_count_body(P, tiles, xy, radii, cam, table);
}
__global__ __launch_bounds__(BUCKET_G) void k_tile_count_views(int P, int tiles, K1Table tab) {
    const K1View &w = tab.v[blockIdx.y];
    tile_count_body(P, tiles, w.g.xy, w.radii, w.cam, w.table);
}

// per tile (one lane each, coalesced across tiles): exclusive prefix over the nb counting workgroups, in place;
// the column total goes to cnt[tile]
__device__ __forceinline__ void tile_colscan_body(int tiles, int nb, uint32_t *__restrict__ table, uint32_t *__restrict__ cnt) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= tiles) return;
    uint32_t run = 0;
    int b = 0;
    for (; b + 8 <= nb; b += 8) {   // 8 independent loads in flight
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = table[(size_t)(b + u) * tiles + t];
#pragma unroll
        for (int u = 0; u < 8; u++) { table[(size_t)(b + u) * tiles + t] = run; run += v[u]; }
    }
    for (; b < nb; b++) { const uint32_t v = table[(size_t)b * tiles + t]; table[(size_t)b * tiles + t] = run; run += v; }
    cnt[t] = run;
}
__global__ __launch_bounds__(256) void k_tile_colscan(int tiles, int nb, uint32_t *__restrict__ table, uint32_t *__restrict__ cnt) {
    tile_colscan_body(tiles, nb, table, cnt);
}
__global__ __launch_bounds__(256) void k_tile_colscan_views(int tiles, int nb, K1Table tab) {
    uint32_t *table = tab.v[blockIdx.y].table;
    tile_colscan_body(tiles, nb, table, table + (size_t)nb * tiles);
}

// single workgroup: exclusive scan of the per-tile totals -> tile ranges, R, longest list
constexpr int INFO_BUSY = 64;   // word offset of the non-empty-tile list inside the info block: [count, tile ids ...]
constexpr int LPT_BINS = 512;   // bins of the longest-list-first order (list length / 16, BUCKET_CAP / 16 = 512)
__device__ __forceinline__ void tile_scan_body(int tiles, const uint32_t *__restrict__ cnt, int2 *__restrict__ ranges,
                                               uint32_t *__restrict__ info, volatile uint32_t *mailbox, uint32_t tag) {
    // ONE scan over the tiles of two running sums packed in 64 bits: low word = instances (the tile ranges), high word = number of
    // non-empty tiles (their compact list: the tile sort launches over it instead of over a grid that is ~90 % empty on scene_1)
    __shared__ unsigned long long s_w[17];
    __shared__ uint32_t s_max[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long carry = 0ull;
    uint32_t mx = 0;
    uint32_t *busy = info + INFO_BUSY;
    for (int base = 0; base < tiles; base += 1024) {
        const int t = base + threadIdx.x;
        const uint32_t c = t < tiles ? cnt[t] : 0u;
        mx = max(mx, c);
        const unsigned long long v0 = (unsigned long long)c | ((unsigned long long)(c ? 1u : 0u) << 32);
        unsigned long long inc = v0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned long long o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        if (w == 0) {
            unsigned long long v = lane < 16 ? s_w[lane] : 0ull, vi = v;
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) { const unsigned long long o = __shfl_up(vi, d, 64); if (lane >= d) vi += o; }
            if (lane < 16) s_w[lane] = vi - v;
            if (lane == 15) s_w[16] = vi;
        }
        __syncthreads();
        const unsigned long long exl = carry + s_w[w] + inc - v0;
        const uint32_t ex = (uint32_t)exl;
        if (t < tiles) {
            ranges[t] = c ? make_int2((int)ex, (int)(ex + c)) : make_int2(0, 0);
            if (c) busy[1 + (uint32_t)(exl >> 32)] = (uint32_t)t;
            else busy[tiles + 4 + tiles - 1 - (t - (int)(uint32_t)(exl >> 32))] = (uint32_t)t;   // launch-order list: empty tiles from the end
        }
        carry += s_w[16];
        __syncthreads();
    }
    // the non-empty tiles once more, LONGEST LIST FIRST (a counting sort on length / 16): the launch order of the compositing forward.
    // Its waves -- one per (tile, 4x4 block), ~14 k of them with work on scene_1 for 8192 wave slots, 40-75 us each -- are handed out in
    // grid order to whichever slot frees: in tile order the longest lists (the middle of the image) start in the middle of the launch and
    // the slots that draw three of them in a row set the kernel's duration while the others idle (4.5 of 8 waves resident on average);
    // longest first, what is still running at the end are the short lists.
    {
        __shared__ uint32_t s_bin[LPT_BINS + 1];
        const uint32_t nbusy = (uint32_t)(carry >> 32);
        uint32_t *lpt = busy + tiles + 4;          // (its tail, the empty tiles, was filled in the loop above)
        for (int i = threadIdx.x; i <= LPT_BINS; i += 1024) s_bin[i] = 0u;
        __syncthreads();                                   // (also: the busy list written above is visible to the whole workgroup)
        for (uint32_t i = threadIdx.x; i < nbusy; i += 1024) {
            const uint32_t c = cnt[busy[1 + i]];
            atomicAdd(&s_bin[LPT_BINS - 1 - min(c >> 4, (uint32_t)(LPT_BINS - 1))], 1u);
        }
        __syncthreads();
        if (w == 0) {                                      // exclusive scan of the bins by one wave: 8 consecutive bins per lane
            uint32_t v[LPT_BINS / 64], run = 0;
#pragma unroll
            for (int k = 0; k < LPT_BINS / 64; k++) { v[k] = s_bin[lane * (LPT_BINS / 64) + k]; run += v[k]; }
            uint32_t inc = run;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64); if (lane >= d) inc += o; }
            uint32_t base = inc - run;
#pragma unroll
            for (int k = 0; k < LPT_BINS / 64; k++) { s_bin[lane * (LPT_BINS / 64) + k] = base; base += v[k]; }
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nbusy; i += 1024) {
            const uint32_t t = busy[1 + i], c = cnt[t];
            lpt[atomicAdd(&s_bin[LPT_BINS - 1 - min(c >> 4, (uint32_t)(LPT_BINS - 1))], 1u)] = t;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
    if (lane == 0) s_max[w] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t m = 0;
        for (int k = 0; k < 16; k++) m = max(m, s_max[k]);
        info[0] = (uint32_t)carry;
        info[1] = m;
        info[2] = (uint32_t)(carry >> 32);               // non-empty tiles
        busy[0] = (uint32_t)(carry >> 32);
        if (mailbox) {   // host-mapped pinned memory: the host polls the tag instead of blocking in a stream synchronise
            mailbox[0] = (uint32_t)carry;
            mailbox[1] = m;
            mailbox[3] = (uint32_t)(carry >> 32);
            __threadfence_system();
            mailbox[2] = tag;
        }
    }
}
__global__ __launch_bounds__(1024) void k_tile_scan(int tiles, const uint32_t *__restrict__ cnt, int2 *__restrict__ ranges,
                                                     uint32_t *__restrict__ info, volatile uint32_t *mailbox, uint32_t tag) {
    tile_scan_body(tiles, cnt, ranges, info, mailbox, tag);
}
__global__ __launch_bounds__(1024) void k_tile_scan_views(int tiles, int nb, K1Table tab) {
    const K1View &w = tab.v[blockIdx.x];
    tile_scan_body(tiles, w.table + (size_t)nb * tiles, w.ranges, w.info, w.mailbox, w.tag);
}

__device__ __forceinline__ void emit_bucket_body(int P, int tiles, const float2 *__restrict__ xy,
                                                 const float *__restrict__ depth, const int32_t *__restrict__ radii,
                                                 const Cam &cam, const uint32_t *__restrict__ table,
                                                 const int2 *__restrict__ ranges, uint64_t *__restrict__ comp) {
    extern __shared__ uint32_t s_base[];
    const uint32_t *row = table + (size_t)blockIdx.x * tiles;
    for (int t = threadIdx.x; t < tiles; t += BUCKET_G) s_base[t] = (uint32_t)ranges[t].x + row[t];
    __syncthreads();
    const int i = blockIdx.x * BUCKET_G + threadIdx.x;
    if (i >= P) return;
    const int rad = radii[i];
    if (rad <= 0) return;
    const float2 p = xy[i];
    int minx, miny, maxx, maxy;
    tile_rect(p.x, p.y, rad, cam, minx, miny, maxx, maxy);
    const uint64_t v = ((uint64_t)__float_as_uint(depth[i]) << 32) | (uint32_t)i;
    for (int y = miny; y < maxy; y++)
        for (int x = minx; x < maxx; x++) comp[atomicAdd(&s_base[y * cam.gx + x], 1u)] = v;
}
__global__ __launch_bounds__(BUCKET_G) void k_emit_bucket(int P, int tiles, const float2 *__restrict__ xy,
                                                           const float *__restrict__ depth, const int32_t *__restrict__ radii,
                                                           Cam cam, const uint32_t *__restrict__ table,
                                                           const int2 *__restrict__ ranges, uint64_t *__restrict__ comp) {
    emit_bucket_body(P, tiles, xy, depth, radii, cam, table, ranges, comp);
}

// The second phase of the forward (after the one host read of the instance counts) for ALL views of a step, one launch per
// stage (blockIdx.y = view): the GPU's dispatcher packs the views' workgroups instead of 4 x 5 launches staggered by the
// host's launch rate.
constexpr int P2_MAX_VIEWS = 8;
struct P2View {
    Geom g;
    Cam cam;
    const int32_t *radii;
    const uint32_t *table;
    int2 *ranges;
    uint64_t *keys_u, *keys_sorted;
    uint32_t *ids_sorted;
    int *seg_offset, *slot_tile;
    float4 *ckpt;
    uint16_t *mask16;
    unsigned long long *bbits;  // [slots][16 blocks][4]: which entries of a segment each block blended (K6 -> K7)
    unsigned long long *bmask;  // [chunks][16 blocks]: which entries of a 64-entry list chunk reach each block (K5b -> K6)
    float4 *recA, *recB;
    float2 *recC;
    const float *bg;
    float *final_T;
    uint32_t *n_contrib;
    float *out_color, *out_depth;
    uint32_t R;                 // list capacity the binning chunk was laid out for (the null record sits at index R)
    // speculative launch (finish_views_batched): the host has not read the counts yet and sized the chunks from the previous call;
    // every kernel of the second phase checks the counts the scan left in `info` against those capacities and leaves the view
    // alone when they do not fit (the host notices the same way and repeats the phase with exact sizes)
    const uint32_t *info;       // [0] tile instances, [1] longest tile list, [2] non-empty tiles
    uint32_t Lcap;              // longest tile list the sort's LDS was sized for
    uint32_t Bcap;              // non-empty tiles the compositing forward's grid was sized for
    int spec;
};
struct P2Table { P2View v[P2_MAX_VIEWS]; uint32_t *valid; int nviews; };     // valid: see csplat_forward_views_faith (NULL otherwise)
__device__ __forceinline__ bool p2_live(const P2View &w) { return !w.spec || (w.info[0] - 1u < w.R && w.info[1] <= w.Lcap && w.info[2] <= w.Bcap); }
__device__ __forceinline__ void seg_plan_body(int tiles, const int2 *__restrict__ ranges, int *__restrict__ seg_offset,
                                              int *__restrict__ slot_tile);
// (the LAST workgroup of every view does not emit: it lays out the view's 256-entry segments -- the former k_seg_plan launch; both only
// need the tile ranges)
__global__ __launch_bounds__(BUCKET_G) void k_emit_bucket_views(int P, int tiles, P2Table tab) {
    const P2View &w = tab.v[blockIdx.y];
    if (!p2_live(w)) return;
    if (blockIdx.x == gridDim.x - 1) { seg_plan_body(tiles, w.ranges, w.seg_offset, w.slot_tile); return; }
    emit_bucket_body(P, tiles, w.g.xy, w.g.depth, w.radii, w.cam, w.table, w.ranges, w.keys_u);
}

constexpr int TSORT_THREADS = 1024;
constexpr int TSORT_WAVES = TSORT_THREADS / 64;
constexpr int TSORT_ITEMS = BUCKET_CAP / TSORT_THREADS;   // 8 keys per lane at most
constexpr int TSORT_NB = 4 * TSORT_THREADS;               // interpolation buckets: every thread owns 4 consecutive ones
constexpr int TSORT_GRID = 512;                           // workgroups per view striding over the non-empty tiles
constexpr int TSORT_LONG = 64;                            // most keys in one bucket before the tile takes the radix fallback
constexpr int TSORT_WORDS = TSORT_NB + 4 + 256 + 8 + 2 + 2 + TSORT_WAVES + 2;   // u32 words of LDS behind the keys
__host__ __device__ inline size_t tsort_lds_bytes(int longest) {
    const int items = (longest > 0 ? longest : 1) + TSORT_THREADS - 1;
    return (size_t)(items / TSORT_THREADS) * TSORT_THREADS * 8 + (size_t)TSORT_WORDS * 4;
}

// One tile's (depth bits << 32 | Gaussian id) keys put into ascending order entirely in LDS + registers.  The result is the
// unique sorted order of the (unique) composite keys, i.e. exactly the stable (tile | depth) radix order of the upstream
// pipeline -- however it is reached:
//   * default (round 3): an INTERPOLATION BUCKET sort.  A tile holds 1 .. 8192 keys whose depths span a narrow range; the
//     depth bits (positive floats: unsigned order = numeric order) are mapped monotonically onto 4096 buckets between the tile's
//     own minimum and maximum, every key takes a slot in its bucket with ONE LDS atomic (the order inside a bucket is whatever
//     the atomics made it), an exclusive scan of the bucket counts gives the bucket starts, the keys are dropped at start +
//     slot, and every key then counts the keys of its bucket that are smaller (its rank: ~1 independent LDS read per key) and
//     moves to start + rank.  Buckets are ordered and complete and the composite keys unique, so the outcome does not depend
//     on the atomic order: bit-identical to the radix sort, in 8 barriers instead of ~6 per radix pass x 3 passes.
//   * fallback (a bucket with more than TSORT_LONG keys: depths piled onto a few buckets by an outlier; all depths equal;
//     csplat_debug_flags bit 11): the round-2 stable LSD radix sort -- keys live in registers between passes (lane l of wave w
//     owns positions w*64*items + i*64 + l), every pass ranks the 8-bit digit with 8 ballots per key and per-wave LDS counters.
// mode: bit 0 = ids < 2^24 (radix: skip byte 3), bit 1 = radix only, bit 2 = bucket limit 1 (tests the fallback path)
__device__ __forceinline__ void tile_sort_body(const int2 *__restrict__ ranges, const uint64_t *__restrict__ comp,
                                               uint64_t *__restrict__ keys_sorted,
                                               uint32_t *__restrict__ ids_sorted, int mode, int tile) {
    extern __shared__ uint64_t s_key[];                 // [m] keys, then the counters
    const int2 r = ranges[tile];
    const int n = r.y - r.x;
    if (n <= 0) return;
    const int skip_byte3 = mode & 1;
    const int items = (n + TSORT_THREADS - 1) / TSORT_THREADS;
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_key + (size_t)items * TSORT_THREADS);   // [TSORT_NB] buckets / [TSORT_WAVES][256]
    uint32_t *s_dig = s_cnt + TSORT_NB + 4;                                                   // [256] + [4] (+ 4 spare)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wbase = w * items * 64;
    const uint64_t lt = (1ull << lane) - 1ull;
    uint64_t key[TSORT_ITEMS];
    uint32_t rank[TSORT_ITEMS];
#pragma unroll
    for (int i = 0; i < TSORT_ITEMS; i++) {
        const int idx = wbase + i * 64 + lane;
        key[i] = (i < items && idx < n) ? comp[r.x + idx] : ~0ull;
    }
    uint32_t *s_diff = s_dig + 264;                                                           // [2]
    uint32_t *s_mm = s_diff + 2;                                                              // [2] min, max of the depth bits
    uint32_t *s_wtot = s_mm + 2;                                                              // [TSORT_WAVES]
    const uint64_t hi = (uint64_t)(uint32_t)tile << 32;
    if (!(mode & 2)) {
        // ---- interpolation bucket sort
        uint32_t dmin = ~0u, dmax = 0u;
#pragma unroll
        for (int i = 0; i < TSORT_ITEMS; i++) {
            const int idx = wbase + i * 64 + lane;
            if (i < items && idx < n) { const uint32_t d = (uint32_t)(key[i] >> 32); dmin = min(dmin, d); dmax = max(dmax, d); }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, o, 64));
            dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, o, 64));
        }
        if (threadIdx.x == 0) { s_mm[0] = ~0u; s_mm[1] = 0u; }
        reinterpret_cast<uint4 *>(s_cnt)[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        if (lane == 0) { atomicMin(&s_mm[0], dmin); atomicMax(&s_mm[1], dmax); }
        __syncthreads();
        dmin = s_mm[0]; dmax = s_mm[1];
        if (dmax != dmin) {                                   // (workgroup-uniform)
            // monotone map of the depth bits onto [0, TSORT_NB): uint -> float conversion, a positive scale and truncation
            // are all non-decreasing, so bucket order never contradicts depth order
            const float scale = (float)TSORT_NB / ((float)(dmax - dmin) + 1.0f);
            uint32_t bs[TSORT_ITEMS];                         // bucket << 16 | slot inside the bucket
#pragma unroll
            for (int i = 0; i < TSORT_ITEMS; i++) {
                const int idx = wbase + i * 64 + lane;
                if (i < items && idx < n) {
                    const uint32_t b = min((uint32_t)(TSORT_NB - 1), (uint32_t)((float)((uint32_t)(key[i] >> 32) - dmin) * scale));
                    bs[i] = (b << 16) | atomicAdd(&s_cnt[b], 1u);
                }
            }
            __syncthreads();
            // exclusive scan of the bucket counts: 4 buckets per thread, a wave scan, the wave totals
            const uint4 c = reinterpret_cast<const uint4 *>(s_cnt)[threadIdx.x];
            const uint32_t tot = c.x + c.y + c.z + c.w;
            uint32_t inc = tot;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t o = __shfl_up(inc, dd, 64); if (lane >= dd) inc += o; }
            if (lane == 63) s_wtot[w] = inc;
            __syncthreads();
            uint32_t ex = inc - tot;
            for (int k = 0; k < w; k++) ex += s_wtot[k];
            reinterpret_cast<uint4 *>(s_cnt)[threadIdx.x] = make_uint4(ex, ex + c.x, ex + c.x + c.y, ex + c.x + c.y + c.z);
            if (threadIdx.x == 0) s_cnt[TSORT_NB] = (uint32_t)n;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < TSORT_ITEMS; i++) {
                const int idx = wbase + i * 64 + lane;
                if (i < items && idx < n) s_key[s_cnt[bs[i] >> 16] + (bs[i] & 0xFFFFu)] = key[i];
            }
            __syncthreads();
            // inside a bucket the order is whatever the atomics made it: every key finds its RANK among the keys of its bucket
            // (independent LDS reads, a bucket holds ~1 key on average; the composite keys are unique) ...
            const int limit = (mode & 4) ? 1 : TSORT_LONG;
            bool long_run = false;
            uint32_t dst[TSORT_ITEMS];
#pragma unroll
            for (int i = 0; i < TSORT_ITEMS; i++) {
                const int idx = wbase + i * 64 + lane;
                if (i < items && idx < n) {
                    const uint32_t b = bs[i] >> 16;
                    const uint32_t lo = s_cnt[b], cb = s_cnt[b + 1] - lo;     // (s_cnt[TSORT_NB] = n)
                    uint32_t rk = 0;
                    if (cb > (uint32_t)limit) long_run = true;
                    else
                        for (uint32_t j = 0; j < cb; j++) rk += s_key[lo + j] < key[i] ? 1u : 0u;
                    dst[i] = lo + rk;
                }
            }
            if (!__syncthreads_or(long_run)) {
                // ... and moves there (the keys are still in registers: in place, behind a barrier)
#pragma unroll
                for (int i = 0; i < TSORT_ITEMS; i++) {
                    const int idx = wbase + i * 64 + lane;
                    if (i < items && idx < n) s_key[dst[i]] = key[i];
                }
                __syncthreads();
#pragma unroll
                for (int i = 0; i < TSORT_ITEMS; i++) {
                    const int idx = wbase + i * 64 + lane;
                    if (i < items && idx < n) {
                        const uint64_t k = s_key[idx];
                        keys_sorted[r.x + idx] = hi | (k >> 32);
                        ids_sorted[r.x + idx] = (uint32_t)k;
                    }
                }
                return;
            }
            // (fallback: key[] still holds the tile's keys; the composite key is unique, so the radix sort below gives the
            // same order whatever order they are in)
        }
    }
    // ---- stable LSD radix sort (fallback)
    // digits on which every key of the tile agrees need no pass (a stable pass over a constant digit is the identity):
    // typically the exponent byte of the depth, and more on short lists.  One OR-reduction of (key ^ first key).
    __syncthreads();
    if (threadIdx.x < 2) s_diff[threadIdx.x] = 0u;
    __syncthreads();
    {
        const uint64_t k0 = comp[r.x];
        uint64_t dv = 0ull;
#pragma unroll
        for (int i = 0; i < TSORT_ITEMS; i++) {
            const int idx = wbase + i * 64 + lane;
            if (i < items && idx < n) dv |= key[i] ^ k0;
        }
        uint32_t lo = (uint32_t)dv, hi32 = (uint32_t)(dv >> 32);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo |= (uint32_t)__shfl_xor((int)lo, o, 64); hi32 |= (uint32_t)__shfl_xor((int)hi32, o, 64); }
        if (lane == 0) { atomicOr(&s_diff[0], lo); atomicOr(&s_diff[1], hi32); }
    }
    __syncthreads();
    const uint64_t diffbits = (uint64_t)s_diff[0] | ((uint64_t)s_diff[1] << 32);
    auto pass = [&](int shift) {
        for (int t = threadIdx.x; t < TSORT_WAVES * 256; t += TSORT_THREADS) s_cnt[t] = 0u;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TSORT_ITEMS; i++) {
            if (i < items) {   // workgroup-uniform
                const int idx = wbase + i * 64 + lane;
                const bool valid = idx < n;
                const uint32_t d = (uint32_t)(key[i] >> shift) & 0xFF;
                unsigned long long peers = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
                for (int b = 0; b < 8; b++) {
                    const unsigned long long mb = __builtin_amdgcn_ballot_w64(valid && ((d >> b) & 1));
                    peers &= ((d >> b) & 1) ? mb : ~mb;
                }
                const uint32_t prev = s_cnt[w * 256 + d];
                rank[i] = prev + (uint32_t)__popcll(peers & lt);
                __builtin_amdgcn_wave_barrier();
                if (valid && (peers & lt) == 0ull) s_cnt[w * 256 + d] = prev + (uint32_t)__popcll(peers);
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        uint32_t c[TSORT_WAVES];
        uint32_t tot = 0;
        if (threadIdx.x < 256) {
#pragma unroll
            for (int k = 0; k < TSORT_WAVES; k++) { c[k] = s_cnt[k * 256 + threadIdx.x]; tot += c[k]; }
            uint32_t inc = tot;   // inclusive scan of the digit totals over 4 waves of 64 digits
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t o = __shfl_up(inc, dd, 64); if (lane >= dd) inc += o; }
            if (lane == 63) s_dig[256 + w] = inc;
            s_dig[threadIdx.x] = inc - tot;
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            uint32_t run = s_dig[threadIdx.x];
            for (int k = 0; k < w; k++) run += s_dig[256 + k];
#pragma unroll
            for (int k = 0; k < TSORT_WAVES; k++) { s_cnt[k * 256 + threadIdx.x] = run; run += c[k]; }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TSORT_ITEMS; i++) {
            if (i < items) {
                const int idx = wbase + i * 64 + lane;
                if (idx < n) {
                    const uint32_t d = (uint32_t)(key[i] >> shift) & 0xFF;
                    s_key[s_cnt[w * 256 + d] + rank[i]] = key[i];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TSORT_ITEMS; i++) {
            const int idx = wbase + i * 64 + lane;
            if (i < items && idx < n) key[i] = s_key[idx];
        }
        __syncthreads();
    };
    auto varies = [&](int byte) { return ((diffbits >> (byte * 8)) & 0xFFull) != 0ull && !(byte == 3 && skip_byte3); };   // workgroup-uniform
    // The key is (depth bits, Gaussian id) and the id only breaks ties between EQUAL depths, which are rare (exact clones right
    // after a densification step, coplanar centres): sort on the depth bytes alone (3 passes on a typical tile instead of 6),
    // then put the runs of equal depth into id order.  Short runs are fixed in place by the lane that owns the run's first
    // element; a run longer than TIE_RUN (or a tile whose depths are all equal) falls back to the full LSD sort over every
    // varying byte -- the composite key is unique, so the result is the same whatever order the keys are in by then.
    constexpr int TIE_RUN = 8;
    bool full = (diffbits >> 32) == 0ull;        // no depth byte varies: nothing but the ids to sort on
    if (!full) {
        for (int byte = 4; byte < 8; byte++)
            if (varies(byte)) pass(byte * 8);
        // (s_key now holds the keys in depth order, key[] = this lane's elements of it)
        bool long_run = false;
#pragma unroll
        for (int i = 0; i < TSORT_ITEMS; i++) {
            const int idx = wbase + i * 64 + lane;
            if (i < items && idx + 1 < n) {
                const uint32_t d = (uint32_t)(key[i] >> 32);
                const bool starts = (idx == 0 || (uint32_t)(s_key[idx - 1] >> 32) != d) && (uint32_t)(s_key[idx + 1] >> 32) == d;
                if (starts) {
                    int j = idx + 2;
                    while (j < n && j - idx <= TIE_RUN && (uint32_t)(s_key[j] >> 32) == d) j++;
                    if (j - idx > TIE_RUN) long_run = true;
                    else
                        for (int a2 = idx + 1; a2 < j; a2++) {          // insertion sort of the run by id (low word)
                            const uint64_t v = s_key[a2];
                            int b2 = a2 - 1;
                            while (b2 >= idx && s_key[b2] > v) { s_key[b2 + 1] = s_key[b2]; b2--; }
                            s_key[b2 + 1] = v;
                        }
                }
            }
        }
        full = __syncthreads_or(long_run);
        if (!full) {
#pragma unroll
            for (int i = 0; i < TSORT_ITEMS; i++) {
                const int idx = wbase + i * 64 + lane;
                if (i < items && idx < n) key[i] = s_key[idx];
            }
        }
    }
    if (full)
        for (int byte = 0; byte < 8; byte++)
            if (varies(byte)) pass(byte * 8);
#pragma unroll
    for (int i = 0; i < TSORT_ITEMS; i++) {
        const int idx = wbase + i * 64 + lane;
        if (i < items && idx < n) {
            keys_sorted[r.x + idx] = hi | (key[i] >> 32);
            ids_sorted[r.x + idx] = (uint32_t)key[i];
        }
    }
}
__global__ __launch_bounds__(TSORT_THREADS) void k_tile_sort(const int2 *__restrict__ ranges, const uint64_t *__restrict__ comp,
                                                              uint64_t *__restrict__ keys_sorted,
                                                              uint32_t *__restrict__ ids_sorted, const uint32_t *__restrict__ info, int mode) {
    // the workgroups stride over the compact list of non-empty tiles the tile scan left behind the counts
    const uint32_t *busy = info + INFO_BUSY;
    const int nbusy = (int)busy[0];
    for (int b = blockIdx.x; b < nbusy; b += gridDim.x) {
        tile_sort_body(ranges, comp, keys_sorted, ids_sorted, mode, (int)busy[1 + b]);
        __syncthreads();
    }
}
__global__ __launch_bounds__(TSORT_THREADS) void k_tile_sort_views(P2Table tab, int mode) {
    const P2View &w = tab.v[blockIdx.y];
    if (!p2_live(w)) return;
    const uint32_t *busy = w.info + INFO_BUSY;
    const int nbusy = (int)busy[0];
    for (int b = blockIdx.x; b < nbusy; b += gridDim.x) {
        tile_sort_body(w.ranges, w.keys_u, w.keys_sorted, w.ids_sorted, mode, (int)busy[1 + b]);
        __syncthreads();
    }
}

template <int CTRL, int RMASK>
__device__ __forceinline__ float dpp_mov(float v, float old) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, RMASK, 0xF, false));
}
// min / max over the 64 lanes, result broadcast to every lane (through an SGPR)
__device__ __forceinline__ float wave_min(float v) {
    v = fminf(v, dpp_mov<0xB1, 0xF>(v, v)); v = fminf(v, dpp_mov<0x4E, 0xF>(v, v));
    v = fminf(v, dpp_mov<0x141, 0xF>(v, v)); v = fminf(v, dpp_mov<0x140, 0xF>(v, v));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fminf(fminf(r0, r1), fminf(r2, r3));
}
__device__ __forceinline__ float wave_max(float v) { return -wave_min(-v); }

// Wave-level culling.  A list entry can only change a pixel if alpha = opacity*exp(power) >= 1/255, and
// power <= -0.5*d^2/lambda_max(cov2D); so it is irrelevant to EVERY pixel of an axis-aligned box whose distance to
// the centre satisfies d^2 > cut2 = 2*lambda_max*ln(255*opacity) (K1 stores cut2 with a safety margin).  Lane l
// tests entry l of a 64-entry group against the wave's box of still-live pixels; the ballot is the work list.
//
// Second, exact stage (scene_1: a quarter of the circle test's survivors reach no pixel -- the projected Gaussians are
// anisotropic and the circle of radius sqrt(cut2) over-covers their ellipse): alpha >= 1/255 <=> q(d) = A dx^2 + 2B dx dy
// + C dy^2 <= tau = 2 ln(255 opacity), so the entry is irrelevant to the whole box if the MINIMUM of q over the box
// exceeds tau.  q is convex: the minimum is 0 if the centre is inside, else it lies on one of the (at most two) edges
// facing the centre, where q is a 1-D quadratic whose minimiser is clamped to the edge.  ~35 VALU per entry per chunk,
// against ~26 per survivor and pixel row saved.  Margins: 1e-3 relative + 1e-3 absolute on tau, 1e-5 of the sum of the
// absolute terms of q (cancellation); culling must stay exact (tests compare against the un-culled run bit for bit).
__device__ __forceinline__ bool box_hit(float2 c, float cut2, float4 co, float bx0, float bx1, float by0, float by1, bool exact) {
    const float lx = bx0 - c.x, hx = bx1 - c.x, ly = by0 - c.y, hy = by1 - c.y;     // the box relative to the centre
    const float ex = fmaxf(lx, fminf(0.f, hx)), ey = fmaxf(ly, fminf(0.f, hy));     // nearest point of the box, per axis
    if (!(ex * ex + ey * ey <= cut2)) return false;
    if (!exact || cut2 > 1.0e30f) return true;
    const float A = co.x, B = co.y, C = co.z;
    float qmin = 0.f, sabs = 0.f;
    if (ex != 0.f || ey != 0.f) {
        float q1 = 3.0e38f, s1 = 0.f, q2 = 3.0e38f, s2 = 0.f;
        if (ex != 0.f) {
            const float dy = fminf(fmaxf(-B * ex * __builtin_amdgcn_rcpf(fmaxf(C, 1e-30f)), ly), hy);
            const float t0 = A * ex * ex, t1 = 2.f * B * ex * dy, t2 = C * dy * dy;
            q1 = t0 + t1 + t2; s1 = t0 + fabsf(t1) + t2;
        }
        if (ey != 0.f) {
            const float dx = fminf(fmaxf(-B * ey * __builtin_amdgcn_rcpf(fmaxf(A, 1e-30f)), lx), hx);
            const float t0 = C * ey * ey, t1 = 2.f * B * ey * dx, t2 = A * dx * dx;
            q2 = t0 + t1 + t2; s2 = t0 + fabsf(t1) + t2;
        }
        const bool first = q1 <= q2;
        qmin = first ? q1 : q2; sabs = first ? s1 : s2;
    }
    const float tau = 2.f * __logf(255.f * co.w);
    return qmin - 1e-5f * sabs <= tau * 1.001f + 1e-3f;
}

#ifndef CSPLAT_SEG
#define CSPLAT_SEG 256
#endif
constexpr int SEG = CSPLAT_SEG;   // tile-list entries per backward segment (multiple of 64)

// per-tile segment plan: seg_offset[t] = first segment slot of tile t (exclusive scan of ceil(n_t / SEG)),
// slot_tile[slot] = owning tile.  One workgroup; tiles are few (2500 at 800x800).
__device__ __forceinline__ void seg_plan_body(int tiles, const int2 *__restrict__ ranges, int *__restrict__ seg_offset,
                                              int *__restrict__ slot_tile) {
    __shared__ int s_w[17];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int carry = 0;
    for (int base = 0; base < tiles; base += 1024) {
        const int t = base + threadIdx.x;
        int ns = 0;
        if (t < tiles) { const int2 r = ranges[t]; ns = (r.y - r.x + SEG - 1) / SEG; }
        int inc = ns;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        if (w == 0) {
            int v = lane < 16 ? s_w[lane] : 0, vi = v;
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) { const int o = __shfl_up(vi, d, 64); if (lane >= d) vi += o; }
            if (lane < 16) s_w[lane] = vi - v;
            if (lane == 15) s_w[16] = vi;
        }
        __syncthreads();
        const int ex = carry + s_w[w] + inc - ns;
        if (t < tiles) {
            seg_offset[t] = ex;
            for (int k = 0; k < ns; k++) slot_tile[ex + k] = t;
        }
        carry += s_w[16];
        __syncthreads();
    }
    if (threadIdx.x == 0) seg_offset[tiles] = carry;
}
__global__ __launch_bounds__(1024) void k_seg_plan(int tiles, const int2 *__restrict__ ranges, int *__restrict__ seg_offset,
                                                    int *__restrict__ slot_tile) {
    seg_plan_body(tiles, ranges, seg_offset, slot_tile);
}
__global__ __launch_bounds__(1024) void k_seg_plan_views(int tiles, P2Table tab) {
    const P2View &w = tab.v[blockIdx.x];
    if (!p2_live(w)) return;
    seg_plan_body(tiles, w.ranges, w.seg_offset, w.slot_tile);
}

// =================================================================================================== K5b / K6 / K7, block form
// The compositing kernels work on 4x4 PIXEL BLOCKS (16 per tile) instead of 8x8 quadrants: a wavefront owns ONE block and
// advances through the block's survivors FOUR AT A TIME -- DPP row r (16 lanes = the 16 pixels of the block) evaluates
// survivor r of the group.  On scene_1 a projected Gaussian covers ~16 of the 64 pixels of a quadrant (26 % of the lanes
// did useful work per survivor); it covers ~8 of the 16 pixels of the blocks it reaches, and a quadrant's survivor reaches
// 2.2 of the 4 blocks: ~1.8x fewer wave-instructions per (pixel, Gaussian) pair, 4x more waves, 4x shorter serial chains.
//   * the per-pixel transmittance chain crosses the four rows: every lane all-gathers the four (1 - alpha) factors of its
//     pixel (three v_permlane{16,32}_swap) and forms the running products in the sequential order T*F0*F1*F2*F3 -- the
//     same association as a one-entry-at-a-time walk, so skipping culled entries (factor 1) cannot change a bit of T;
//   * which entries reach which block is decided ONCE per view by k_block_masks (exact ellipse-vs-box test, one lane per
//     tile-list entry, 16 boxes): a 16-bit mask per entry plus a tile-ordered copy of what compositing reads (40 B, so the
//     walkers read contiguous records instead of gathering five arrays by Gaussian id).  Since round 4 the same masks also leave
//     TRANSPOSED (bmask[chunk][block]: the sixteen ballots of a wave's 64 entries): K6's wave takes its block's word of a chunk with one
//     scalar load; K7 no longer looks at the masks at all -- it walks what K6 found BLENDED (bbits);
//   * K7 runs FORWARD through a 256-entry segment: with S_k = sum_{j<=k} (c_j . dL/dC) alpha_j T_j (restarted from the
//     forward's checkpoint) the upstream back-to-front recurrence collapses to
//         dL/dalpha_k = T_k (c_k . dL/dC) - (out_colour . dL/dC - S_k) / (1 - alpha_k),
//     the same identity the depth-split restart already used once per segment;
//   * what a survivor's 16-lane row sums over its pixels are the MOMENTS of m = G dL/dalpha about the Gaussian's centre + three colour
//     sums (round 6; rounds 2-5: the nine gradient values, by a 4-level butterfly): they factor over the 4 x 4 block, 19 DPP adds + 3
//     selects per survivor row (processN), and go, nine lanes at once, to the Gaussian's 64-byte record as ONE global float-atomic
//     request per (entry, block) -- requests are priced per 64 bytes at the memory side (MI355X_MICROARCH.md "Global float atomics"),
//     and their rate is what binds the kernel (profiles/r06_k7_elimination.txt); K8 turns the moments into gradients.
constexpr float T_EPS = 0.0001f;
constexpr float ALPHA_MIN = 1.f / 255.f;

struct Row4 { float v0, v1, v2, v3; };
// every lane receives the values its pixel position holds in rows 0..3 (rows = 16-lane groups)
__device__ __forceinline__ Row4 rows_allgather(float f) {
    const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(f), __float_as_uint(f), false, false);   // [f0 f0 f2 f2], [f1 f1 f3 f3]
    const auto ev = __builtin_amdgcn_permlane32_swap(s16[0], s16[0], false, false);                           // f0 x4, f2 x4
    const auto od = __builtin_amdgcn_permlane32_swap(s16[1], s16[1], false, false);                           // f1 x4, f3 x4
    return {__uint_as_float(ev[0]), __uint_as_float(od[0]), __uint_as_float(ev[1]), __uint_as_float(od[1])};
}
__device__ __forceinline__ float rows_sum(float f) { const Row4 g = rows_allgather(f); return ((g.v0 + g.v1) + g.v2) + g.v3; }
// row r of the wave takes the r-th argument: three DPP moves with a row mask (lanes of the other rows keep the old value)
__device__ __forceinline__ float rowsel(int, float a, float b, float c, float d) {
    int x = __float_as_int(a);
    x = __builtin_amdgcn_update_dpp(x, __float_as_int(b), 0xE4, 0x2, 0xF, false);
    x = __builtin_amdgcn_update_dpp(x, __float_as_int(c), 0xE4, 0x4, 0xF, false);
    x = __builtin_amdgcn_update_dpp(x, __float_as_int(d), 0xE4, 0x8, 0xF, false);
    return __int_as_float(x);
}

// ------------------------------------------------------------------------------------------- K5b
__device__ __forceinline__ void block_masks_body(int64_t R, int64_t null_at, int gx, const uint64_t *__restrict__ keys_sorted,
                                                 const uint32_t *__restrict__ ids_sorted, const float4 *__restrict__ pack,
                                                 uint16_t *__restrict__ mask16, float4 *__restrict__ recA,
                                                 float4 *__restrict__ recB, float2 *__restrict__ recC, int exact, int64_t block,
                                                 unsigned long long *__restrict__ bmask) {
    const int64_t i = block * 256 + threadIdx.x;
    // (workgroup-uniform: nothing of this workgroup's range is in use -- no list entry, not the list's last chunk, not the null record)
    if (block * 256 > (R | 63) && !(block * 256 <= null_at && null_at < block * 256 + 256)) return;
    if (i == null_at) {   // the null record behind the list (at the list's CAPACITY): opacity 0, pads incomplete groups of four
        mask16[i] = 0;
        recA[i] = make_float4(0.f, 0.f, 0.f, 0.f); recB[i] = make_float4(0.f, 0.f, 0.f, 0.f); recC[i] = make_float2(0.f, 0.f);
    }
    uint32_t m = 0;
    if (i < R) {
    const uint32_t tile = (uint32_t)(keys_sorted[i] >> 32), id = ids_sorted[i];
    const float4 pa = pack[3 * (size_t)id], pb = pack[3 * (size_t)id + 1], pc = pack[3 * (size_t)id + 2];
    const float2 c = make_float2(pa.x, pa.y);
    const float4 co = make_float4(pa.z, pa.w, pb.x, pb.y);
    const float cut = pc.z;
    const float x0 = (float)((tile % (uint32_t)gx) * CSPLAT_TILE), y0 = (float)((tile / (uint32_t)gx) * CSPLAT_TILE);
#pragma unroll
    for (int by = 0; by < 4; by++)
#pragma unroll
        for (int bx = 0; bx < 4; bx++)
            if (box_hit(c, cut, co, x0 + 4.f * bx, x0 + 4.f * bx + 3.f, y0 + 4.f * by, y0 + 4.f * by + 3.f, exact)) m |= 1u << (by * 4 + bx);
    mask16[i] = (uint16_t)m;
    recA[i] = pa;
    recB[i] = pb;
    recC[i] = make_float2(pc.x, pc.y);
    }
    // the wave's 64 entries are list chunk i >> 6: the sixteen ballots ARE the chunk's per-block words; lane b of the wave stores block b's
    // (entries at or behind the list's end contribute 0; K6 masks its last chunk by the list length anyway)
    if (bmask && (i >> 6) <= (R >> 6)) {
        uint32_t lo = 0u, hi = 0u;
        // (v_writelane_b32 with a literal lane: block b's ballot -- an SGPR pair -- lands in lane b of (lo, hi))
#define CSPLAT_WORD_TO_LANE(b)                                                                                                      \
        {                                                                                                                           \
            const unsigned long long wb_ = __builtin_amdgcn_ballot_w64((m >> b) & 1u);                                              \
            asm("v_writelane_b32 %0, %2, " #b "\n\tv_writelane_b32 %1, %3, " #b                                                     \
                : "+v"(lo), "+v"(hi) : "s"((uint32_t)wb_), "s"((uint32_t)(wb_ >> 32)));                                             \
        }
        CSPLAT_WORD_TO_LANE(0) CSPLAT_WORD_TO_LANE(1) CSPLAT_WORD_TO_LANE(2) CSPLAT_WORD_TO_LANE(3)
        CSPLAT_WORD_TO_LANE(4) CSPLAT_WORD_TO_LANE(5) CSPLAT_WORD_TO_LANE(6) CSPLAT_WORD_TO_LANE(7)
        CSPLAT_WORD_TO_LANE(8) CSPLAT_WORD_TO_LANE(9) CSPLAT_WORD_TO_LANE(10) CSPLAT_WORD_TO_LANE(11)
        CSPLAT_WORD_TO_LANE(12) CSPLAT_WORD_TO_LANE(13) CSPLAT_WORD_TO_LANE(14) CSPLAT_WORD_TO_LANE(15)
#undef CSPLAT_WORD_TO_LANE
        const int lane = threadIdx.x & 63;
        if (lane < 16) bmask[(size_t)(i >> 6) * 16 + lane] = ((unsigned long long)hi << 32) | lo;
    }
}
__global__ __launch_bounds__(256) void k_block_masks(int64_t R, int gx, const uint64_t *__restrict__ keys_sorted,
                                                      const uint32_t *__restrict__ ids_sorted, const float4 *__restrict__ pack,
                                                      uint16_t *__restrict__ mask16, float4 *__restrict__ recA,
                                                      float4 *__restrict__ recB, float2 *__restrict__ recC, int exact,
                                                      unsigned long long *__restrict__ bmask) {
    block_masks_body(R, R, gx, keys_sorted, ids_sorted, pack, mask16, recA, recB, recC, exact, blockIdx.x, bmask);
}
// xcd_views = V (1, 2, 4 or 8) on a 1-D grid: a workgroup's XCD is blockIdx.x % 8 and XCD x serves ONLY view x % V.  The list entries of
// a tile gather their Gaussians' 48-byte records in depth order (random), and a Gaussian recurs in the tiles next to and below it -- one
// tile row later, ~2 MB of gathers per view: inside one XCD's 4 MB L2 when that L2 sees one view, outside it when the workgroups of all
// the step's views interleave on every XCD.  xcd_views = 0: blockIdx.y = view.
__global__ __launch_bounds__(256) void k_block_masks_views(P2Table tab, int exact, int xcd_views) {
    int view = blockIdx.y;
    int64_t block = blockIdx.x;
    if (xcd_views > 0) {
        const int xcd = blockIdx.x & 7;
        view = xcd % xcd_views;
        block = (int64_t)(blockIdx.x >> 3) * (8 / xcd_views) + xcd / xcd_views;
    }
    const P2View &w = tab.v[view];
    if (!p2_live(w)) return;
    block_masks_body(w.spec ? (int64_t)w.info[0] : (int64_t)w.R, (int64_t)w.R, w.cam.gx, w.keys_sorted, w.ids_sorted, w.g.pack, w.mask16,
                     w.recA, w.recB, w.recC, exact, block, w.bmask);
}

// The survivors of block `blk` among list positions [lo, hi) of one tile, as a stream of GROUPS OF FOUR that never cross a
// SEG boundary (incomplete groups are padded with -1).  A 64-entry chunk of masks is turned into list positions with one
// ballot + mbcnt and appended to a small ring in LDS (wave-private); group k is ring[4k .. 4k+3], so row r of the wave reads
// its survivor with one ds_read_b32.  All counters are wave-uniform (SGPRs); the mask of the next chunk is prefetched.
constexpr int RING = 128;   // >= 3 groups in flight (12; K7 keeps 2) + one chunk (64) + padding (3)
constexpr int RING16 = 256; // groups of sixteen: 3 x 16 in flight + one chunk + padding (15)
template <int G, int RN>
struct BlockStreamT {
    const uint16_t *m16;     // the tile's masks (already offset by range.x)
    int *ring;
    int cbase, hi, blk, lane, tail;
    uint32_t m_next;
    __device__ __forceinline__ uint32_t load(int base) const { const int e = base + lane; return e < hi ? (uint32_t)m16[e] : 0u; }
    __device__ __forceinline__ void start(const uint16_t *masks, int lo, int hi_, int blk_, int lane_, int *ring_) {
        m16 = masks; hi = hi_; blk = blk_; lane = lane_; ring = ring_; cbase = lo; tail = 0;
        m_next = load(lo);
    }
    // (the caller has already requested the first chunk: first = load(lo))
    __device__ __forceinline__ void start(const uint16_t *masks, int lo, int hi_, int blk_, int lane_, int *ring_, uint32_t first) {
        m16 = masks; hi = hi_; blk = blk_; lane = lane_; ring = ring_; cbase = lo; tail = 0;
        m_next = first;
    }
    __device__ __forceinline__ void ingest() {   // one chunk
        const uint32_t m = m_next;
        m_next = load(cbase + 64);
        const bool hit = (m >> blk) & 1u;
        const unsigned long long cur = __builtin_amdgcn_ballot_w64(hit);
        if (hit) {
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(cur >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cur, 0u));
            ring[(tail + rank) & (RN - 1)] = cbase + lane;
        }
        tail += (int)__popcll(cur);
        cbase += 64;
        if ((cbase & (SEG - 1)) == 0 || cbase >= hi) {   // the segment (or the list) ends here: complete the group
            const int pad = (-tail) & (G - 1);
            if (lane < pad) ring[(tail + lane) & (RN - 1)] = -1;
            tail += pad;
        }
    }
    // list position of survivor `slot` of group k (-1 = padding); false when the stream ends before group k
    __device__ __forceinline__ bool group(int k, int slot, int &pos) {
        while (G * k + G > tail && cbase < hi) ingest();
        if (G * k >= tail) return false;
        pos = ring[(G * k + slot) & (RN - 1)];
        return true;
    }
};
typedef BlockStreamT<4, RING> BlockStream;

// The same stream fed from K5b's TRANSPOSED masks (round 4): bmask[chunk][block] is the ballot a wave of BlockStreamT forms from 64 mask
// loads -- here it arrives by ONE scalar load per chunk (two chunks ahead), so the stream issues no vector-memory instruction at all and
// the only loads of K6's loop are the step's records.  Chunks of bmask are aligned to the GLOBAL list index; a tile's list starts at any
// rx, so tile-relative chunk j is bits o.. of word g0 + j joined with bits ..o-1 of word g0 + j + 1 (o = rx & 63: a funnel shift on the
// scalar unit) -- segment boundaries (multiples of 256 tile-relative entries) then fall between chunks as before.
// (the words are read through a CONSTANT-address-space pointer: nothing writes bmask while K6 runs, and only then does the compiler keep
//  the loads on the scalar unit inside the loop -- behind the loop's stores a plain global pointer gets a vector load + v_readfirstlane
//  and an s_waitcnt vmcnt(0) on the spot)
typedef const __attribute__((address_space(4))) unsigned long long *const_u64_ptr;
template <int G, int RN>
struct WordStreamT {
    const_u64_ptr bw;                // word of global chunk g0 for this block; + 16 per chunk
    int *ring;
    int cbase, hi, lane, tail, o, j;
    unsigned long long wa, wb, wc;
    __device__ __forceinline__ void start(const unsigned long long *bmask, uint32_t rx, int hi_, int blk, int lane_, int *ring_) {
        bw = (const_u64_ptr)(bmask + ((size_t)(rx >> 6) * 16 + (size_t)blk));
        o = (int)(rx & 63u); hi = hi_; lane = lane_; ring = ring_; cbase = 0; tail = 0; j = 0;
        wa = bw[0]; wb = bw[16]; wc = bw[32];
    }
    __device__ __forceinline__ void ingest() {   // one tile-relative chunk
        unsigned long long cur = o ? (wa >> o) | (wb << (64 - o)) : wa;
        const int rem = hi - cbase;
        if (rem < 64) cur &= (1ull << rem) - 1ull;
        wa = wb; wb = wc; j++;
        wc = bw[(size_t)(j + 2) * 16];
        if (__builtin_amdgcn_inverse_ballot_w64(cur)) {
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(cur >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cur, 0u));
            ring[(tail + rank) & (RN - 1)] = cbase + lane;
        }
        tail += (int)__popcll(cur);
        cbase += 64;
        if ((cbase & (SEG - 1)) == 0 || cbase >= hi) {   // the segment (or the list) ends here: complete the group
            const int pad = (-tail) & (G - 1);
            if (lane < pad) ring[(tail + lane) & (RN - 1)] = -1;
            tail += pad;
        }
    }
    // list position of survivor `slot` of group k (-1 = padding, and -1 with `false` when the stream ends before group k)
    __device__ __forceinline__ bool group(int k, int slot, int &pos) {
        while (G * k + G > tail && cbase < hi) ingest();
        const bool ok = G * k < tail;
        pos = ok ? ring[(G * k + slot) & (RN - 1)] : -1;
        return ok;
    }
};

struct Trip { float4 a, b; float2 c; int pos; uint32_t id; };   // the lane's survivor of a group (row r's), pos = list position or -1

// ---- which entries a block BLENDED (round 4).  K5b's masks say which entries can REACH a 4x4 block (ellipse vs box); K6 finds out which
// of them any pixel of the block actually blends -- alpha >= 1/255 at some pixel centre that is still open -- and K7 only ever does
// arithmetic for those: a survivor that no pixel blended has factor 1 and addend 0 at all sixteen pixels (bit for bit: same exp, same
// tests), so dropping it changes no bit of T, S or any gradient.  K6 marks a blended survivor with ONE BIT in a 256-bit LDS strip (the
// segment's list positions); when its walk leaves a segment the strip is stored as four 64-bit words bbits[slot][block][0..3] -- the
// TRANSPOSE of mask16 restricted to what was blended -- and K7's waves read their segment's survivor set with one scalar 32-byte load
// instead of four vector loads of masks + ballots.  Segments a block's walk skipped (no survivor) get zero words; K7 never looks behind
// the block's last blended entry (blk_hi).
// (the strip is kept as 8 x 32 BITS, set with ds_or_b32: the flush is then one LDS read and one 4-byte store by eight lanes -- ballots
// over a byte strip, four 64-bit selects and their addresses cost the survivor-column K6 22 VGPRs at the flush point, i.e. its fifth wave)
__device__ __forceinline__ void bbits_mark(uint32_t *s_bits, int pos) { atomicOr(&s_bits[(pos & (SEG - 1)) >> 5], 1u << (pos & 31)); }
__device__ __forceinline__ void bbits_flush(uint32_t *s_bits, unsigned long long *__restrict__ bbits, size_t slot, int blk, int lane) {
    if (lane < SEG / 32) {
        reinterpret_cast<uint32_t *>(bbits)[(slot * 16 + (size_t)blk) * (SEG / 32) + lane] = s_bits[lane];
        s_bits[lane] = 0u;
    }
}
__device__ __forceinline__ void bbits_zero(unsigned long long *__restrict__ bbits, size_t slot, int blk, int lane) {
    if (lane < SEG / 32) reinterpret_cast<uint32_t *>(bbits)[(slot * 16 + (size_t)blk) * (SEG / 32) + lane] = 0u;
}

// ------------------------------------------------------------------------------------------- K6
// grid: 16 single-wave workgroups per tile; the 16 blocks of a tile have the same blockIdx % 8 (same XCD, shared L2 lines)
__device__ __forceinline__ void composite_fwd_body(int tiles, int W, int H, int gx, const int2 *__restrict__ ranges,
                                                   const uint16_t *__restrict__ mask16, const float4 *__restrict__ recA,
                                                   const float4 *__restrict__ recB, const float2 *__restrict__ recC,
                                                   uint32_t null_rec, const float *__restrict__ bg,
                                                   int *seg_offset, float4 *__restrict__ ckpt,
                                                   float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,
                                                   float *__restrict__ out_color, float *__restrict__ out_depth, int wg,
                                                   unsigned long long *__restrict__ bbits,
                                                   const uint32_t *__restrict__ order = nullptr) {
    __shared__ int s_ring[RING];
    __shared__ uint32_t s_hit[SEG / 32];
    // item (wg >> 7) * 8 + (wg & 7), block (wg >> 3) & 15: the 16 blocks of an item share blockIdx % 8 (one XCD).  order: a permutation
    // of the tiles, longest list first, the empty tiles (background only) last (k_tile_scan)
    const int item = ((wg >> 7) << 3) + (wg & 7), blk = (wg >> 3) & 15;
    if (item >= tiles) return;
    const int tile = order ? (int)order[item] : item;
    const int lane = threadIdx.x, r = lane >> 4, l16 = lane & 15;
    const int px = (tile % gx) * CSPLAT_TILE + (blk & 3) * 4 + (l16 & 3);
    const int py = (tile / gx) * CSPLAT_TILE + (blk >> 2) * 4 + (l16 >> 2);
    const bool inside = px < W && py < H;
    const int pix = py * W + px;
    const float fx = (float)px, fy = (float)py;
    const int2 range = ranges[tile];
    const int n = range.y - range.x;
    const uint32_t rx = (uint32_t)range.x;
    bool done = !inside;
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f;   // T: the pixel's (same in its 4 lanes); C*, Dp: this row's share
    uint32_t last = 0;
    if (n > 0 && __builtin_amdgcn_ballot_w64(!done) != 0ull) {
        const int seg0 = seg_offset[tile];
        BlockStream st;
        st.start(mask16 + rx, 0, n, blk, lane, s_ring);
        int seg_written = -1;
        if (lane < SEG / 32) s_hit[lane] = 0u;
        auto fetch = [&](Trip &t, int k) -> bool {
            if (!st.group(k, r, t.pos)) return false;
            const uint32_t ri = t.pos >= 0 ? rx + (uint32_t)t.pos : null_rec;
            t.a = recA[ri]; t.b = recB[ri]; t.c = recC[ri];
            return true;
        };
        auto process = [&](const Trip &t) {
            const int seg = __builtin_amdgcn_readfirstlane(t.pos) / SEG;   // (a group's first entry is never padding)
            if (seg != seg_written) {
                // entering a new 256-entry segment: checkpoint (T, colour so far) for the depth-split backward, for every
                // segment start passed since the last one (segments without a survivor of this block get the same state)
                const float t0 = rows_sum(C0), t1 = rows_sum(C1), t2 = rows_sum(C2);
                if (r == 0)
                    for (int s = seg_written + 1; s <= seg; s++)
                        ckpt[(size_t)(seg0 + s) * 256 + blk * 16 + l16] = make_float4(T, t0, t1, t2);
                C0 = r == 0 ? t0 : 0.f; C1 = r == 0 ? t1 : 0.f; C2 = r == 0 ? t2 : 0.f;
                if (seg_written >= 0) bbits_flush(s_hit, bbits, (size_t)(seg0 + seg_written), blk, lane);
                for (int s = seg_written + 1; s < seg; s++) bbits_zero(bbits, (size_t)(seg0 + s), blk, lane);
                seg_written = seg;
            }
            const float dx = t.a.x - fx, dy = t.a.y - fy;
            const float power = -0.5f * (t.a.z * dx * dx + t.b.x * dy * dy) - t.a.w * dx * dy;
            const float a = fminf(0.99f, t.b.y * __expf(power));
            const float al = (!done && power <= 0.f && a >= ALPHA_MIN) ? a : 0.f;
            const float F = 1.f - al;
            const Row4 g = rows_allgather(F);
            const float P1 = T * g.v0, P2 = P1 * g.v1, P3 = P2 * g.v2, P4 = P3 * g.v3;
            const float Tr = rowsel(r, T, P1, P2, P3);
            const bool blend = al > 0.f && Tr * F >= T_EPS;     // (Tr * F is this row's P_{r+1}, bit for bit)
            const float wgt = blend ? al * Tr : 0.f;
            C0 += t.b.z * wgt; C1 += t.b.w * wgt; C2 += t.c.x * wgt; Dp += t.c.y * wgt;
            last = blend ? (uint32_t)(t.pos + 1) : last;
            {   // the row's survivor was blended at one of its 16 pixels: its byte in the segment's strip
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(blend);
                if (l16 == 0 && ((bal >> (lane & 48)) & 0xFFFFull) != 0ull) bbits_mark(s_hit, t.pos);
            }
            // the products only decrease: the pixel's T after the group is the last one still above the threshold
            T = P4 >= T_EPS ? P4 : (P3 >= T_EPS ? P3 : (P2 >= T_EPS ? P2 : (P1 >= T_EPS ? P1 : T)));
            done = done || !(P4 >= T_EPS);
        };
        // software pipeline, three groups in flight: the records of group k+3 are requested when group k has been composited
        Trip ta, tb, tc;
        bool va = fetch(ta, 0), vb = fetch(tb, 1), vc = fetch(tc, 2);
        int k = 3;
        while (va) {
            process(ta);
            if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
            va = fetch(ta, k++);
            if (!vb) break;
            process(tb);
            if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
            vb = fetch(tb, k++);
            if (!vc) break;
            process(tc);
            if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
            vc = fetch(tc, k++);
        }
        if (seg_written >= 0) bbits_flush(s_hit, bbits, (size_t)(seg0 + seg_written), blk, lane);
    }
    C0 = rows_sum(C0); C1 = rows_sum(C1); C2 = rows_sum(C2); Dp = rows_sum(Dp);
    {
        const Row4 g = rows_allgather(__uint_as_float(last));
        last = max(max(__float_as_uint(g.v0), __float_as_uint(g.v1)), max(__float_as_uint(g.v2), __float_as_uint(g.v3)));
    }
    {   // the block's largest n_contrib, for K7's workgroups (seg_offset[tiles + 1 ...] = blk_hi[tile][blk])
        uint32_t m = inside ? last : 0u;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
        if (lane == 0) reinterpret_cast<uint32_t *>(seg_offset)[tiles + 1 + tile * 16 + blk] = m;
    }
    if (inside && r == 0) {
        final_T[pix] = T;
        n_contrib[pix] = last;
        const size_t HW = (size_t)H * W;
        out_color[pix] = C0 + T * bg[0];
        out_color[HW + pix] = C1 + T * bg[1];
        out_color[2 * HW + pix] = C2 + T * bg[2];
        out_depth[pix] = Dp;
    }
}
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// One level of a row scan (x[lane] op= x[lane - N] inside every DPP row of 16; lanes without a source keep their value) for FOUR
// independent registers at once: the four instructions are independent, so three of them cover the two wait states a DPP read
// needs after a VALU write of the same register (FIRST: the registers were last written by ordinary VALU code -> s_nop 1).
#define CSPLAT_ROW_SCAN4(OP, N, FIRST, a, b, c, d)                                                                                  \
    asm(FIRST "v_" OP "_f32_dpp %0, %0, %0 row_shr:" #N " row_mask:0xf bank_mask:0xf\n\t"                                          \
              "v_" OP "_f32_dpp %1, %1, %1 row_shr:" #N " row_mask:0xf bank_mask:0xf\n\t"                                          \
              "v_" OP "_f32_dpp %2, %2, %2 row_shr:" #N " row_mask:0xf bank_mask:0xf\n\t"                                          \
              "v_" OP "_f32_dpp %3, %3, %3 row_shr:" #N " row_mask:0xf bank_mask:0xf"                                               \
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
__device__ __forceinline__ void row_scan4_mul(float (&x)[4]) {
    CSPLAT_ROW_SCAN4("mul", 1, "s_nop 1\n\t", x[0], x[1], x[2], x[3]);
    CSPLAT_ROW_SCAN4("mul", 2, "", x[0], x[1], x[2], x[3]);
    CSPLAT_ROW_SCAN4("mul", 4, "", x[0], x[1], x[2], x[3]);
    CSPLAT_ROW_SCAN4("mul", 8, "", x[0], x[1], x[2], x[3]);
}
__device__ __forceinline__ void row_scan4_add(float (&x)[4]) {
    CSPLAT_ROW_SCAN4("add", 1, "s_nop 1\n\t", x[0], x[1], x[2], x[3]);
    CSPLAT_ROW_SCAN4("add", 2, "", x[0], x[1], x[2], x[3]);
    CSPLAT_ROW_SCAN4("add", 4, "", x[0], x[1], x[2], x[3]);
    CSPLAT_ROW_SCAN4("add", 8, "", x[0], x[1], x[2], x[3]);
}
// ------------------------------------------------------------------------------------------- K6, survivor-column form (round 3)
// The lane mapping of composite_bwd16_body for the forward: a step takes SIXTEEN consecutive survivors of the block, lane l holds
// survivor l & 15 and the four pixels of block row l >> 4.  The transmittance of a pixel in front of every survivor is a 4-level
// row_shr product scan along its DPP row (+ one shift, one row_newbcast) instead of an all-gather of four factors + a row select per
// group of four, a lane accumulates colour and depth for ITS survivor only (summed over the row's lanes once per tile and at the
// segment checkpoints), and the serial chain a wave walks -- what bounds this kernel: one wave per block goes through the whole
// tile list -- is a quarter as many steps long.  The products of a step associate as a scan tree, not front to back: final_T and the
// alpha / transmittance decisions can differ from a sequential walk in the last bit (the tests hold n_contrib to the oracle up to
// counted threshold ties and final_T to 1e-4, as they do for v_exp_f32 against expf).  Measured (profiles/r03*): 45 %
// fewer VALU instructions than the row form (composite_fwd_body) but 96-106 VGPRs against 62, i.e. 4-5 waves per SIMD against 8.  While
// the launch still handed 16 waves to every empty tile it lost (188-197 us against 182 us for the four views of a step); launched for
// the non-empty tiles only -- ~17 k long waves for 8192 slots, where the length of a wave is what counts and not how many fit -- it
// wins: 137 against 158 us.  It is the DEFAULT; csplat_debug_flags bit 15 selects the row form.
__device__ __forceinline__ void row_scan4_min(float (&x)[4]) {
    CSPLAT_ROW_SCAN4("min", 1, "s_nop 1\n\t", x[0], x[1], x[2], x[3]);
    CSPLAT_ROW_SCAN4("min", 2, "", x[0], x[1], x[2], x[3]);
    CSPLAT_ROW_SCAN4("min", 4, "", x[0], x[1], x[2], x[3]);
    CSPLAT_ROW_SCAN4("min", 8, "", x[0], x[1], x[2], x[3]);
}
// sum over the 16 lanes of every DPP row, result in all of them
__device__ __forceinline__ float row_total(float v) {
    v = dpp_add<0xB1>(v); v = dpp_add<0x4E>(v); v = dpp_add<0x141>(v); v = dpp_add<0x140>(v);
    return v;
}
__device__ __forceinline__ void composite_fwd16_body(int tiles, int W, int H, int gx, const int2 *__restrict__ ranges,
                                                     const uint16_t *__restrict__ mask16, const float4 *__restrict__ recA,
                                                     const float4 *__restrict__ recB, const float2 *__restrict__ recC,
                                                     uint32_t null_rec, const float *__restrict__ bg,
                                                     int *seg_offset, float4 *__restrict__ ckpt,
                                                     float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,
                                                     float *__restrict__ out_color, float *__restrict__ out_depth,
                                                     unsigned long long *__restrict__ bbits,
                                                     const unsigned long long *__restrict__ bmask,
                                                     const uint32_t *__restrict__ order = nullptr) {
    __shared__ int s_ring[RING16];
    __shared__ uint32_t s_hit[SEG / 32];
    const int wg = blockIdx.x;
    const int item = ((wg >> 7) << 3) + (wg & 7), blk = (wg >> 3) & 15;
    if (item >= tiles) return;
    const int tile = order ? (int)order[item] : item;
    const int lane = threadIdx.x, sv = lane & 15, q = lane >> 4;
    const int px0 = (tile % gx) * CSPLAT_TILE + (blk & 3) * 4;
    const int py = (tile / gx) * CSPLAT_TILE + (blk >> 2) * 4 + q;
    const float fy = (float)py;
    const int2 range = ranges[tile];
    const int n = range.y - range.x;
    const uint32_t rx = (uint32_t)range.x;
    // A pixel that is DONE (outside the image, or its walk has ended: T fell below 1e-4) keeps working transmittance 0 -- every weight it
    // forms is 0 by arithmetic, no select -- and its final T waits in s_Tend; which pixels are done is a LANE MASK per pixel column
    // (m_done[j], an SGPR pair: the decisions of a step are scalar-unit logic on compare results, selects take the masks directly).
    __shared__ float s_Tend[16];
    bool inside[4];
    unsigned long long m_done[4];
    float T[4], C0[4], C1[4], C2[4], Dp[4], fx[4];
    uint32_t last[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        inside[j] = px0 + j < W && py < H;
        m_done[j] = __builtin_amdgcn_ballot_w64(!inside[j]);
        T[j] = inside[j] ? 1.f : 0.f; C0[j] = C1[j] = C2[j] = Dp[j] = 0.f;      // T: the pixel's (same in its 16 lanes); C*, Dp: this lane's survivors' share
        last[j] = 0u;
        fx[j] = (float)(px0 + j);
    }
    int seg0 = 0, seg_written = -1;
    if (n > 0 && (m_done[0] & m_done[1] & m_done[2] & m_done[3]) != ~0ull) {
        seg0 = seg_offset[tile];
        WordStreamT<16, RING16> st;
        st.start(bmask, rx, n, blk, lane, s_ring);
        if (lane < SEG / 32) s_hit[lane] = 0u;
        // (the three loads are issued whether or not the stream still has a step: s_waitcnt vmcnt counts in order, and a load the
        //  compiler must assume was NOT issued makes it wait for the youngest ones -- see K7's loop)
        auto fetch = [&](Trip &t, int k) -> bool {
            const bool ok = st.group(k, sv, t.pos);
            const uint32_t ri = t.pos >= 0 ? rx + (uint32_t)t.pos : null_rec;
            t.a = recA[ri]; t.b = recB[ri]; t.c = recC[ri];
            return ok;
        };
        auto process = [&](const Trip &t) {
            const int seg = __builtin_amdgcn_readfirstlane(t.pos) / SEG;   // (a group's first entry is never padding)
            if (seg != seg_written) {
                // entering a new 256-entry segment: checkpoint (T, colour so far) of every pixel for the depth-split backward, for
                // every segment start passed since the last one.  The colour so far is spread over the row's lanes: sum it, keep
                // the total in lane 0 of the row
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float t0 = row_total(C0[j]), t1 = row_total(C1[j]), t2 = row_total(C2[j]);
                    if (sv == 0)
                        for (int s_ = seg_written + 1; s_ <= seg; s_++)
                            ckpt[(size_t)(seg0 + s_) * 256 + blk * 16 + q * 4 + j] = make_float4(T[j], t0, t1, t2);
                    C0[j] = sv == 0 ? t0 : 0.f; C1[j] = sv == 0 ? t1 : 0.f; C2[j] = sv == 0 ? t2 : 0.f;
                }
                if (seg_written >= 0) bbits_flush(s_hit, bbits, (size_t)(seg0 + seg_written), blk, lane);
                for (int s_ = seg_written + 1; s_ < seg; s_++) bbits_zero(bbits, (size_t)(seg0 + s_), blk, lane);
                seg_written = seg;
            }
            const float dy = t.a.y - fy;
            float al[4], inc[4];
            unsigned long long m_live[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float dx = t.a.x - fx[j];
                const float power = -0.5f * (t.a.z * dx * dx + t.b.x * dy * dy) - t.a.w * dx * dy;
                const float a = fminf(0.99f, t.b.y * __expf(power));
                m_live[j] = __builtin_amdgcn_ballot_w64(power <= 0.f) & __builtin_amdgcn_ballot_w64(a >= ALPHA_MIN);
                al[j] = __builtin_amdgcn_inverse_ballot_w64(m_live[j]) ? a : 0.f;
                inc[j] = 1.f - al[j];
            }
            row_scan4_mul(inc);                                          // the pixel's factor up to and including every survivor
            float Tr[4], P[4], Pend[4];
#pragma unroll
            for (int j = 0; j < 4; j++) { Tr[j] = T[j]; P[j] = T[j] * inc[j]; }
            // transmittance in front of the lane's survivor (T x the scan of the lane to the left; survivor 0 of the row keeps T) and
            // behind the step's last survivor (row_newbcast:15)
            asm("s_nop 1\n\t"
                "v_mul_f32_dpp %0, %4, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_mul_f32_dpp %1, %5, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_mul_f32_dpp %2, %6, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_mul_f32_dpp %3, %7, %3 row_shr:1 row_mask:0xf bank_mask:0xf"
                : "+v"(Tr[0]), "+v"(Tr[1]), "+v"(Tr[2]), "+v"(Tr[3]) : "v"(inc[0]), "v"(inc[1]), "v"(inc[2]), "v"(inc[3]));
            asm("s_nop 1\n\t"
                "v_mov_b32_dpp %0, %4 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                "v_mov_b32_dpp %1, %5 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                "v_mov_b32_dpp %2, %6 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                "v_mov_b32_dpp %3, %7 row_newbcast:15 row_mask:0xf bank_mask:0xf"
                : "=&v"(Pend[0]), "=&v"(Pend[1]), "=&v"(Pend[2]), "=&v"(Pend[3]) : "v"(P[0]), "v"(P[1]), "v"(P[2]), "v"(P[3]));
            unsigned long long m_end[4], any_end = 0ull, m_bl = 0ull;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                m_end[j] = __builtin_amdgcn_ballot_w64(!(Pend[j] >= T_EPS)) & ~m_done[j];       // the pixel's walk ends inside this step
                any_end |= m_end[j];
            }
            if (any_end == 0ull) {
                // no pixel of the block ends in this step: every product of an open pixel is above the threshold (they only decrease
                // along the row), so a survivor is blended exactly where its alpha passed -- and a done pixel's weight is 0 x anything
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float wgt = al[j] * Tr[j];
                    C0[j] += t.b.z * wgt; C1[j] += t.b.w * wgt; C2[j] += t.c.x * wgt; Dp[j] += t.c.y * wgt;
                    const unsigned long long mb = m_live[j] & ~m_done[j];
                    m_bl |= mb;
                    last[j] = __builtin_amdgcn_inverse_ballot_w64(mb) ? (uint32_t)(t.pos + 1) : last[j];
                    T[j] = Pend[j];
                }
            } else {
                float cand[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const unsigned long long mb = m_live[j] & ~m_done[j] & __builtin_amdgcn_ballot_w64(P[j] >= T_EPS);
                    m_bl |= mb;
                    const float wgt = __builtin_amdgcn_inverse_ballot_w64(mb) ? al[j] * Tr[j] : 0.f;
                    C0[j] += t.b.z * wgt; C1[j] += t.b.w * wgt; C2[j] += t.c.x * wgt; Dp[j] += t.c.y * wgt;
                    last[j] = __builtin_amdgcn_inverse_ballot_w64(mb) ? (uint32_t)(t.pos + 1) : last[j];
                    // the T an ending pixel keeps: the last product above the threshold (the products only decrease) = the row's smallest candidate
                    cand[j] = __builtin_amdgcn_inverse_ballot_w64(m_end[j]) ? (P[j] >= T_EPS ? P[j] : T[j]) : 3.0e38f;
                }
                row_scan4_min(cand);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float mend = dpp_mov<0x15F, 0xF>(cand[j], cand[j]);
                    const bool ended = __builtin_amdgcn_inverse_ballot_w64(m_end[j]);
                    if (ended && sv == 0) s_Tend[q * 4 + j] = mend;
                    m_done[j] |= m_end[j];
                    T[j] = __builtin_amdgcn_inverse_ballot_w64(m_done[j]) ? 0.f : Pend[j];
                }
            }
            {   // survivor sv was blended at one of the block's 16 pixels (its four lanes, four pixels each): its bit in the strip
                const uint32_t any16 = (uint32_t)(m_bl | (m_bl >> 16) | (m_bl >> 32) | (m_bl >> 48)) & 0xFFFFu;
                if (lane < 16 && ((any16 >> lane) & 1u)) bbits_mark(s_hit, t.pos);
            }
        };
        // software pipeline, two steps in flight (a step is ~16 survivors x 4 pixels of arithmetic: one step ahead covers the fetch)
        Trip ta, tb;
        bool va = fetch(ta, 0), vb = fetch(tb, 1);
        int k = 2;
        auto all_done = [&]() { return (m_done[0] & m_done[1] & m_done[2] & m_done[3]) == ~0ull; };
        while (va) {
            process(ta);
            if (all_done()) break;
            va = fetch(ta, k++);
            if (!vb) break;
            process(tb);
            if (all_done()) break;
            vb = fetch(tb, k++);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++)         // the pixels whose walk ended: the transmittance they kept
        if (inside[j] && __builtin_amdgcn_inverse_ballot_w64(m_done[j])) T[j] = s_Tend[q * 4 + j];
    uint32_t hi_ = 0u;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        C0[j] = row_total(C0[j]); C1[j] = row_total(C1[j]); C2[j] = row_total(C2[j]); Dp[j] = row_total(Dp[j]);
        uint32_t m = last[j];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
        last[j] = m;
        hi_ = max(hi_, inside[j] ? m : 0u);
    }
    {   // the block's largest n_contrib, for K7's workgroups (seg_offset[tiles + 1 ...] = blk_hi[tile][blk])
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) hi_ = max(hi_, (uint32_t)__shfl_xor((int)hi_, o, 64));
        if (lane == 0) reinterpret_cast<uint32_t *>(seg_offset)[tiles + 1 + tile * 16 + blk] = hi_;
    }
    if (sv < 4) {   // lane j of every row writes pixel j of that row
        const size_t HW = (size_t)H * W;
        float t_ = T[0], c0 = C0[0], c1 = C1[0], c2 = C2[0], dp = Dp[0];
        uint32_t la = last[0];
        bool in_ = inside[0];
#pragma unroll
        for (int j = 1; j < 4; j++)
            if (sv == j) { t_ = T[j]; c0 = C0[j]; c1 = C1[j]; c2 = C2[j]; dp = Dp[j]; la = last[j]; in_ = inside[j]; }
        if (in_) {
            const int pix = py * W + px0 + sv;
            final_T[pix] = t_;
            n_contrib[pix] = la;
            out_color[pix] = c0 + t_ * bg[0];
            out_color[HW + pix] = c1 + t_ * bg[1];
            out_color[2 * HW + pix] = c2 + t_ * bg[2];
            out_depth[pix] = dp;
        }
    }
    // (the last segment's strip leaves here, where nothing else is live: flushed right behind the loop it cost the kernel 18 VGPRs)
    if (seg_written >= 0) bbits_flush(s_hit, bbits, (size_t)(seg0 + seg_written), blk, lane);
}
template <bool ROWS>
__global__ __launch_bounds__(64) void k_composite_fwd(int tiles, int W, int H, int gx, const int2 *__restrict__ ranges,
                                                       const uint16_t *__restrict__ mask16, const float4 *__restrict__ recA,
                                                       const float4 *__restrict__ recB, const float2 *__restrict__ recC,
                                                       uint32_t null_rec, const float *__restrict__ bg,
                                                       int *seg_offset, float4 *__restrict__ ckpt,
                                                       float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,
                                                       float *__restrict__ out_color, float *__restrict__ out_depth,
                                                       unsigned long long *__restrict__ bbits,
                                                       const unsigned long long *__restrict__ bmask) {
    if (ROWS)
        composite_fwd_body(tiles, W, H, gx, ranges, mask16, recA, recB, recC, null_rec, bg, seg_offset, ckpt, final_T, n_contrib, out_color,
                           out_depth, (int)blockIdx.x, bbits);
    else
        composite_fwd16_body(tiles, W, H, gx, ranges, mask16, recA, recB, recC, null_rec, bg, seg_offset, ckpt, final_T, n_contrib, out_color,
                             out_depth, bbits, bmask);
}
// the waves behind the first busy_grid of a K6 launch: the tiles of the launch-order list that got no waves of their own -- the empty
// ones -- receive what K6 writes for a tile without a list (background colour, T = 1, no contributor, blk_hi = 0), 256 pixels a pass
constexpr int K6_EXTRA = 256;
__device__ __forceinline__ void paint_empty_tiles(int tiles, int W, int H, int gx, const uint32_t *__restrict__ order, int busy_grid,
                                                  const float *__restrict__ bg, int *seg_offset, float *__restrict__ final_T,
                                                  uint32_t *__restrict__ n_contrib, float *__restrict__ out_color,
                                                  float *__restrict__ out_depth) {
    const int lane = threadIdx.x;
    const size_t HW = (size_t)H * W;
    const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
    uint32_t *blk_hi = reinterpret_cast<uint32_t *>(seg_offset) + tiles + 1;
    for (int pos = (busy_grid >> 7 << 3) + ((int)blockIdx.x - busy_grid); pos < tiles; pos += (int)gridDim.x - busy_grid) {
        const int tile = (int)order[pos];
        if (lane < 16) blk_hi[tile * 16 + lane] = 0u;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int px = (tile % gx) * CSPLAT_TILE + (lane & 15), py = (tile / gx) * CSPLAT_TILE + 4 * q + (lane >> 4);
            if (px < W && py < H) {
                const int pix = py * W + px;
                final_T[pix] = 1.f;
                n_contrib[pix] = 0u;
                out_color[pix] = b0; out_color[HW + pix] = b1; out_color[2 * HW + pix] = b2;
                out_depth[pix] = 0.f;
            }
        }
    }
}
// blockIdx.x < busy_grid: one wave per (item, block) of the first busy_grid / 16 entries of the launch-order list (the non-empty tiles,
// longest list first: every one of them is among the entries, p2_live: info[2] <= Bcap); the waves behind paint what is left of the list,
// the empty tiles.  (Round 3 measured the alternatives that left the library in round 4: 16 waves for every tile in tile order, and
// 1024 n persistent waves per view walking the items -- DESIGN section 6.)
template <bool ROWS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5))) void k_composite_fwd_views(int tiles, int W, int H, P2Table tab, int busy_grid) {
    if (tab.valid && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        // the verdict of a launch on faith (csplat_forward_views_faith): every view's counts fitted the capacities its second phase was
        // laid out for -- what the backward's kernels and the optimizer step read before they touch anything
        bool ok = true;
        for (int i = 0; i < tab.nviews; i++) ok = ok && p2_live(tab.v[i]);
        *tab.valid = ok ? 1u : 0u;
    }
    const P2View &w = tab.v[blockIdx.y];
    if (!p2_live(w)) return;
    const uint32_t *order = w.info + INFO_BUSY + tiles + 4;
    if ((int)blockIdx.x >= busy_grid) {
        paint_empty_tiles(tiles, W, H, w.cam.gx, order, busy_grid, w.bg, w.seg_offset, w.final_T, w.n_contrib, w.out_color, w.out_depth);
        return;
    }
    if (ROWS)
        composite_fwd_body(tiles, W, H, w.cam.gx, w.ranges, w.mask16, w.recA, w.recB, w.recC, w.R, w.bg, w.seg_offset, w.ckpt, w.final_T,
                           w.n_contrib, w.out_color, w.out_depth, (int)blockIdx.x, w.bbits, order);
    else
        composite_fwd16_body(tiles, W, H, w.cam.gx, w.ranges, w.mask16, w.recA, w.recB, w.recC, w.R, w.bg, w.seg_offset, w.ckpt, w.final_T,
                             w.n_contrib, w.out_color, w.out_depth, w.bbits, w.bmask, order);
}

// ------------------------------------------------------------------------------------------- K7
// per-Gaussian gradient accumulator filled by K7 and consumed by K8 (one 64-byte record per Gaussian):
//   rounds 1-5: 0 dmean2D.x  1 dmean2D.y  2 dconic.a  3 dconic.b  4 dconic.c  5 dopacity  6..8 dcolour  9..15 pad
//   round 6:    0 Mx  1 My  2 Mxx  3 Mxy  4 Myy  5 M0 (= dopacity)  6..8 dcolour -- moments of G dL/dalpha (moments_to_gradients, K8)
constexpr int ACC_STRIDE = 16;

// ---- K7's chains across the four survivor rows of a group (round 4).  v_permlane16_swap(x, x) hands every lane the two values of its
// row PAIR (even row's, odd row's); one v_permlane32_swap of the pair's product / sum hands it the totals of both pairs.  The prefix a row
// needs is then two row-masked DPP operations away -- 2 swaps + 2 masked ops per chain instead of an all-gather of the four values
// (3 swaps), four dependent operations and three row selects.  The product / sum of a group associates as (f0 f1)(f2 f3) instead of
// front to back: K7's T and S differ from a sequential walk in the last bit (K7 takes every blend decision from n_contrib, not from T).
#define CSPLAT_ROWMASK_OP(OP, dst, src0, src1, RM)                                                                                  \
    asm("s_nop 1\n\tv_" OP "_f32_dpp %0, %1, %2 quad_perm:[0,1,2,3] row_mask:" RM " bank_mask:0xf" : "+v"(dst) : "v"(src0), "v"(src1))
// Tin: the pixel's transmittance in front of the group (same in the pixel's four lanes) -> in front of the lane's own survivor, behind the group
__device__ __forceinline__ void rows_scan_mul(float f, float Tin, float &Tr, float &Tout) {
    const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(f), __float_as_uint(f), false, false);
    const float ev = __uint_as_float(s16[0]), od = __uint_as_float(s16[1]);       // the pair's even / odd row
    const float pr = ev * od;
    const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(pr), __float_as_uint(pr), false, false);
    const float lo = __uint_as_float(s32[0]), hi = __uint_as_float(s32[1]);       // rows 0-1, rows 2-3
    float base = Tin;
    CSPLAT_ROWMASK_OP("mul", base, lo, Tin, "0xc");                               // rows 2, 3: behind the first pair
    Tr = base;
    CSPLAT_ROWMASK_OP("mul", Tr, ev, base, "0xa");                                // odd rows: behind the pair's even row
    Tout = (Tin * lo) * hi;
}
// inclusive: Sr = Sin + the addends up to and including the lane's own survivor
__device__ __forceinline__ void rows_scan_add(float g, float Sin, float &Sr, float &Sout) {
    const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(g), __float_as_uint(g), false, false);
    const float ev = __uint_as_float(s16[0]), od = __uint_as_float(s16[1]);
    const float pr = ev + od;
    const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(pr), __float_as_uint(pr), false, false);
    const float lo = __uint_as_float(s32[0]), hi = __uint_as_float(s32[1]);
    float base = Sin;
    CSPLAT_ROWMASK_OP("add", base, lo, Sin, "0xc");
    float incl = ev;
    CSPLAT_ROWMASK_OP("add", incl, od, ev, "0xa");
    Sr = base + incl;
    Sout = (Sin + lo) + hi;
}
// grid: one 4-wave workgroup per (segment slot, group of four live blocks); wave w = one 4x4 block.  The four waves of a workgroup
// share nothing but the launch geometry: no LDS records, no barrier.
// Round 4, first step: a wave's survivors are the entries of the segment its block BLENDED (bbits, written by K6 -- see bbits_flush), not
// the entries that reach the block: the four ballot words arrive with one scalar load, the whole segment's survivor list is laid out in
// the wave's LDS ring before the first group (no mask loads, no ingest inside the loop, a counted loop), and every group of four does
// arithmetic that lands in a gradient: 280 -> 259 us for the four views of a step.
// Second step: the nine row sums of a survivor go STRAIGHT to the Gaussian's 64-byte record (one global float atomic request per (entry,
// block): the nine lanes that hold the sums address one record).  Rounds 1-3 added them into an LDS record per list entry first (shared
// by the workgroup's four blocks, ds_add_f32), zeroed before and flushed behind a barrier -- a quarter of the global requests, but: an
// LDS float atomic per group on a unit the CU's 32 waves share, 9 KB of zeroing and ten flush rounds per workgroup, and every wave waiting
// at the barrier for the slowest of its four blocks (8.7 k of a live wave's 46 k cycles, tools/k7_stamps.py).  259 -> 241 us (same-box
// A/B, three alternations).  DET (csplat_debug_flags bit 8, bit-reproducible): the sums are STORED, one 9-float record per (list entry,
// block) -- each pair is visited exactly once -- and k_det_reduce adds every Gaussian's records in emission order.
#ifndef CSPLAT_K7X
#define CSPLAT_K7X 0
#endif
constexpr int RING7 = SEG;     // a segment's survivors of one block, padded to a multiple of four: at most SEG
template <bool DET>
__device__ __forceinline__ void composite_bwd_body(int tiles, int W, int H, int gx, const int2 *__restrict__ ranges,
                                                   const uint32_t *__restrict__ ids_sorted,
                                                   const unsigned long long *__restrict__ bbits, const float4 *__restrict__ recA,
                                                   const float4 *__restrict__ recB, const float2 *__restrict__ recC,
                                                   uint32_t null_rec, const int *__restrict__ seg_offset,
                                                   const int *__restrict__ slot_tile, const float4 *__restrict__ ckpt,
                                                   const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
                                                   const float *__restrict__ out_color, const float *__restrict__ dL_dpix,
                                                   float *__restrict__ acc, float *__restrict__ det,
                                                   unsigned long long *stamp = nullptr, int wg = (int)blockIdx.x) {
    __shared__ int s_ring[4][RING7];
    __shared__ float4 s_ra[4][64], s_rb[4][64];                     // one batch of 64 survivors' records per wave (see `stage`)
    __shared__ float s_rc[4][64];
    __shared__ uint32_t s_rid[DET ? 1 : 4][DET ? 1 : 64];
    // in-kernel stamps (csplat_debug_stamps; tools/k7_stamps.py): wave 0 of every workgroup leaves s_memtime at the phase boundaries
    unsigned long long *my_stamp = stamp ? stamp + ((size_t)blockIdx.y * gridDim.x + (size_t)wg) * 12 : nullptr;
    auto mark = [&](int k) {
        if (my_stamp && threadIdx.x == 0) {
            asm volatile("" ::: "memory");
            my_stamp[k] = (unsigned long long)__builtin_amdgcn_s_memtime();
            asm volatile("" ::: "memory");
        }
    };
    mark(0);
    const int slot = ((wg >> 5) << 3) + (wg & 7), quad = (wg >> 3) & 3;     // the 4 workgroups of a slot share blockIdx % 8
    if (slot >= seg_offset[tiles]) return;
    const int tile = slot_tile[slot];
    const int seg = slot - seg_offset[tile];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane >> 4, l16 = lane & 15;
    const int seg_lo = seg * SEG;
    // (round 3) K6 leaves every block's largest n_contrib (blk_hi).  The blocks of the tile that still blend something at or behind this
    // segment -- blk_hi > seg_lo -- are PACKED four to a workgroup in index order: workgroup `quad` of the slot takes the live blocks
    // 4 quad .. 4 quad + 3, whichever quadrant of the tile they lie in (they only share the segment's LDS records).
    const uint32_t *blk_hi = reinterpret_cast<const uint32_t *>(seg_offset) + tiles + 1 + tile * 16;
    uint32_t livemask = 0u;
#pragma unroll
    for (int b = 0; b < 16; b++) livemask |= ((int)blk_hi[b] > seg_lo ? 1u : 0u) << b;
    const int n_live = __builtin_popcount(livemask);
    if (4 * quad >= n_live) return;                                      // (workgroup-uniform)
    const int kth = 4 * quad + __builtin_amdgcn_readfirstlane(w);
    uint32_t m_ = livemask;
    for (int i = 0; i < kth && m_; i++) m_ &= m_ - 1u;
    const bool has_block = kth < n_live;
    const int blk = has_block ? __builtin_ctz(m_) : 0;
    // the entries of this segment the block blended: four 64-bit words, one scalar load
    constexpr int NW = SEG / 64;
    const unsigned long long *bw = bbits + ((size_t)slot * 16 + (size_t)blk) * NW;
    unsigned long long sw[NW];
#pragma unroll
    for (int c = 0; c < NW; c++) sw[c] = has_block ? bw[c] : 0ull;
    const int px = (tile % gx) * CSPLAT_TILE + (blk & 3) * 4 + (l16 & 3);
    const int py = (tile / gx) * CSPLAT_TILE + (blk >> 2) * 4 + (l16 >> 2);
    const bool inside = px < W && py < H;
    const int pix = py * W + px;
    const float fx = (float)px, fy = (float)py;
    const int2 range = ranges[tile];
    const uint32_t rx = (uint32_t)range.x;
    mark(1);                                                            // the scalar chain (slot -> tile -> range, blk_hi, bbits) has returned
    // ONE memory round trip for everything a wave needs before its first group: the pixel's constants, its checkpoint and (below, `stage`)
    // the records + ids of its first 64 survivors are requested together
    unsigned long long any_ = 0ull;
#pragma unroll
    for (int c = 0; c < NW; c++) any_ |= sw[c];
    const bool live = any_ != 0ull;
    const size_t HW = (size_t)H * W;
    int ncontrib = 0;
    float dp0 = 0.f, dp1 = 0.f, dp2 = 0.f, oc0 = 0.f, oc1 = 0.f, oc2 = 0.f;
    float4 ck = make_float4(1.f, 0.f, 0.f, 0.f);
    if (live) {
        if (inside) {
            ncontrib = (int)n_contrib[pix];
            dp0 = dL_dpix[pix]; dp1 = dL_dpix[HW + pix]; dp2 = dL_dpix[2 * HW + pix];
            oc0 = out_color[pix]; oc1 = out_color[HW + pix]; oc2 = out_color[2 * HW + pix];
        }
        ck = ckpt[(size_t)slot * 256 + blk * 16 + l16];
    }
    // the wave's survivor list: list positions of the set bits, in order, padded with -1 to a multiple of four (wave-private LDS)
    int *ring = s_ring[w];
    int total = 0;
    if (live) {
#pragma unroll
        for (int c = 0; c < NW; c++) {
            const unsigned long long cur = sw[c];
            if ((cur >> lane) & 1ull) {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(cur >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cur, 0u));
                ring[total + rank] = seg_lo + 64 * c + lane;
            }
            total += (int)__popcll(cur);
        }
        const int pad = (-total) & 3;
        if (lane < pad) ring[total + lane] = -1;
        total += pad;
    }
    mark(2);
    if (live) {
        // ---- the survivors' RECORDS go through LDS, 64 survivors (16 groups) a batch: lane i requests survivor i's record (x, y, conic,
        // opacity, colour -- 36 bytes -- and the Gaussian's id) and parks it in the wave's strip; the group loop then contains NO vector
        // load, only LDS reads (row r reads slot 4k + r: a broadcast) and the atomics.  Why: vmcnt counts loads and atomics together, IN
        // ORDER -- a record requested behind an atomic cannot be used before that atomic has retired, and under load a float atomic stays
        // counted for ~3,000 cycles (MI355X_MICROARCH.md, cycle constants).  With the records fetched from global memory two groups
        // ahead, every group waited for the atomic of the group before the last: 1,670 cycles per group for ~350 of arithmetic
        // (tools/k7_stamps.py).  The first batch's requests travel with the pixel constants and the checkpoint: one round trip in all
        // before the first group; later batches (a block that blended more than 64 of the segment's 256 entries) wait once per batch.
        float4 *ra = s_ra[w], *rb = s_rb[w];
        float *rc = s_rc[w];
        uint32_t *rid = s_rid[DET ? 0 : w];
        auto stage = [&](int b0) {      // survivors b0 .. b0 + 63 -> the strip
            const int idx = b0 + lane;
            const int pos = idx < total ? ring[idx] : -1;
            const uint32_t ri = pos >= 0 ? rx + (uint32_t)pos : null_rec;
            const float4 A = recA[ri], B = recB[ri];
            const float C = reinterpret_cast<const float *>(recC)[2 * (size_t)ri];      // (c.y, the depth, is K6's)
            uint32_t id = 0u;
            if (!DET) id = ids_sorted[pos >= 0 ? rx + (uint32_t)pos : rx];
            ra[lane] = A; rb[lane] = B; rc[lane] = C;
            if (!DET) rid[lane] = id;
        };
        stage(0);
        const float OD = oc0 * dp0 + oc1 * dp1 + oc2 * dp2;
        float T = 1.f, S = 0.f;
        if (ncontrib > seg_lo) {
            T = ck.x;
            S = ck.y * dp0 + ck.z * dp1 + ck.w * dp2;
        }
        if (my_stamp && threadIdx.x == 0) { asm volatile("" :: "v"(S), "v"(T)); }
        mark(3);                                                        // the pixel constants, the checkpoint and the first batch have arrived
        // after the moment reduction (processN) nine lanes of a row hold a record entry each: lane 4 j + t of the row = block pixel
        // (column t, row j).  Record: 0 Mx  1 My  2 Mxx  3 Mxy  4 Myy  5 M0  6..8 colour (K8: moments -> dL/dmean2D, dL/dconic)
        const bool lq0 = (l16 & 3) == 0, lq1 = (l16 & 3) == 1, lq2 = (l16 & 3) == 2;
        constexpr int RED_T[16] = {5, 0, 2, 7, 4, -1, -1, -1, 1, 3, -1, 8, 6, -1, -1, -1};
        int red_t_ = -1;
#pragma unroll
        for (int q = 0; q < 16; q++) red_t_ = l16 == q ? RED_T[q] : red_t_;
        const bool red_active = red_t_ >= 0;
        const int red_t = red_active ? red_t_ : 0;
        int base = 0;                   // first survivor of the batch in the strip
        // (every LDS read of the loop is unconditional, with a clamped slot: the compiler's lgkmcnt bookkeeping assumes the path on which
        //  a conditional read was NOT issued, and then waits for the youngest ones)
        auto fetch = [&](Trip &t, int k) {
            const int sl = 4 * k + r;
            t.pos = ring[base + sl];
            t.a = ra[sl]; t.b = rb[sl]; t.c.x = rc[sl];
            if (!DET) t.id = rid[sl];
        };
        // N groups at once, statement by statement: a group is one dependent chain of ~110 vector instructions (~10 cycles from one to the
        // next: ~1,200 cycles a group for a wave on its own, tools/k7_stamps.py -- the same with the atomics removed); groups k and k + 1
        // only meet where T and S pass from one to the other, so written side by side the two chains fill each other's waits.
        auto processN = [&](auto NC, const Trip *const *t) {
            constexpr int N = decltype(NC)::value;
            float dx[N], dy[N], G[N], al[N], F[N], gdot[N], Tr[N], Sr[N], dcc[N], tot[N];
            bool act[N];
#pragma unroll
            for (int u = 0; u < N; u++) {
                dx[u] = t[u]->a.x - fx; dy[u] = t[u]->a.y - fy;
                const float power = -0.5f * (t[u]->a.z * dx[u] * dx[u] + t[u]->b.x * dy[u] * dy[u]) - t[u]->a.w * dx[u] * dy[u];
                G[u] = __expf(power);
                const float a = fminf(0.99f, t[u]->b.y * G[u]);
                act[u] = t[u]->pos < ncontrib && power <= 0.f && a >= ALPHA_MIN;   // (padding: pos = -1, opacity 0 -> a = 0)
                al[u] = act[u] ? a : 0.f;
                F[u] = 1.f - al[u];
                gdot[u] = t[u]->b.z * dp0 + t[u]->b.w * dp1 + t[u]->c.x * dp2;
            }
#pragma unroll
            for (int u = 0; u < N; u++) rows_scan_mul(F[u], T, Tr[u], T);
#pragma unroll
            for (int u = 0; u < N; u++) dcc[u] = al[u] * Tr[u];
#pragma unroll
            for (int u = 0; u < N; u++) rows_scan_add(gdot[u] * dcc[u], S, Sr[u], S);
#pragma unroll
            for (int u = 0; u < N; u++) {
                const float dL_dalpha = act[u] ? Tr[u] * gdot[u] - (OD - Sr[u]) * __builtin_amdgcn_rcpf(F[u]) : 0.f;
                // Round 6: the row's lanes no longer form the nine GRADIENT values and reduce each over the 16 pixels (14 multiplications + a
                // 4-level butterfly of ~24 DPP operations); they reduce the MOMENTS of m = G dL/dalpha about the Gaussian's centre,
                //   M0 = sum m, Mx = sum m dx, My = sum m dy, Mxx = sum m dx^2, Mxy = sum m dx dy, Myy = sum m dy^2,
                // which factor over the 4 x 4 block (dx depends on the pixel column only, dy on the row only): first over the rows j at a
                // fixed column (values m, m dy, m dy^2 and colour 0 -- a two-level transposing fold on the 4-lane banks), then over the
                // columns with the weights 1, dx, dx^2 (two quad_perm levels); colours 1 and 2 take a five-step reduction of their own.
                // 7 multiplications + 19 DPP adds + 3 selects; K8 turns the summed moments into dL/dmean2D and dL/dconic once per
                // Gaussian: dL/dmean2D = -0.5 o (a Mx + b My, c My + b Mx), dL/dconic = -0.5 o (Mxx, Mxy, Myy), dL/dopacity = M0.
                const float gda = G[u] * dL_dalpha;
                const float c0v = dcc[u] * dp0, c1v = dcc[u] * dp1, c2v = dcc[u] * dp2;
                float a1, a2, P, Q, R, X, b1, b2;
                asm("v_mul_f32 %0, %8, %10\n\t"                                                               // a1 = m dy
                    "v_mul_f32 %1, %0, %10\n\t"                                                               // a2 = m dy^2
                    "v_add_f32_dpp %2, %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"                       // P (rows 0, 1) = m      + partner row's
                    "v_add_f32_dpp %5, %12, %12 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"                     // X (rows 0, 1) = colour 1
                    "v_add_f32_dpp %5, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"                     // X (rows 2, 3) = colour 2
                    "v_add_f32_dpp %2, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"                       // P (rows 2, 3) = m dy
                    "v_add_f32_dpp %3, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"                       // Q (rows 0, 1) = m dy^2
                    "v_add_f32_dpp %3, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"                     // Q (rows 2, 3) = colour 0
                    "v_add_f32_dpp %4, %2, %2 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"                      // R (rows 0, 2) = P over all four rows
                    "v_add_f32_dpp %5, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"                 // X: the half's pairs
                    "v_add_f32_dpp %4, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"                       // R (rows 1, 3) = Q over all four rows
                    "v_mul_f32 %6, %4, %9\n\t"                                                                // b1 = R dx
                    "v_mul_f32 %7, %6, %9\n\t"                                                                // b2 = R dx^2
                    "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                    "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                    "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                    "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                    "v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                    "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                    "v_add_f32_dpp %6, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                    "v_add_f32_dpp %7, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                    : "=&v"(a1), "=&v"(a2), "=&v"(P), "=&v"(Q), "=&v"(R), "=&v"(X), "=&v"(b1), "=&v"(b2)
                    : "v"(gda), "v"(dx[u]), "v"(dy[u]), "v"(c0v), "v"(c1v), "v"(c2v));
                // column t of the block's lane grid keeps: t = 0 the plain sums, t = 1 the dx-weighted, t = 2 the dx^2-weighted, t = 3 colours 1 / 2
                tot[u] = lq0 ? R : (lq1 ? b1 : (lq2 ? b2 : X));
            }
            // (every survivor of the list was blended at one of the block's pixels: the row always has something to add, padding aside)
#pragma unroll
            for (int u = 0; u < N; u++)
                if (red_active && t[u]->pos >= 0) {
                    if (DET) det[((size_t)(rx + (uint32_t)t[u]->pos) * 16 + (size_t)blk) * 9 + red_t] = tot[u];   // one (entry, block) pair is visited exactly once
                    else if (CSPLAT_K7X != 1) atomicAdd(acc + (size_t)t[u]->id * ACC_STRIDE + red_t, tot[u]);                      // nine lanes, one 64-byte record
                    else asm volatile("" :: "v"(tot[u]));        // (elimination build CSPLAT_K7X=1: the sums are formed, nothing is sent)
                }
        };
        auto process = [&](const Trip &t0) { const Trip *t[1] = {&t0}; processN(std::integral_constant<int, 1>{}, t); };
        auto process2 = [&](const Trip &t0, const Trip &t1) { const Trip *t[2] = {&t0, &t1}; processN(std::integral_constant<int, 2>{}, t); };
        bool first = true;
        // (round 6, tried and dropped: requesting the NEXT batch's records before this batch's atomics are issued -- vmcnt retires in order and
        //  a load queued behind a float atomic waits for it -- costs ten registers across the batch (44 bytes of scratch at the 64-register
        //  bound) and changed nothing: 214-215 against 210-217 us, profiles/r06_k7_elimination.txt)
        for (; base < total && CSPLAT_K7X != 4; base += 64) {
            if (!first) stage(base);
            const int ngroups = min(64, total - base) >> 2, last_g = ngroups - 1;
            // two groups in flight: group k + 2 is read from the strip when group k has been composited
            Trip ta, tb;
            fetch(ta, 0);
            fetch(tb, min(1, last_g));
            if (first) {
                if (my_stamp && threadIdx.x == 0) { asm volatile("" :: "v"(ta.a.x), "v"(ta.c.x)); }
                mark(4);                                                // strip -> the first group's records have arrived
            }
            first = false;
            int k = 0;
            for (; k + 1 < ngroups; k += 2) {       // (a fetch past the end re-reads the last group: never processed)
                process2(ta, tb);
                fetch(ta, min(k + 2, last_g));
                fetch(tb, min(k + 3, last_g));
            }
            if (k < ngroups) process(ta);
        }
    }
    mark(5);                                                            // this wave's groups are done
    if (my_stamp && threadIdx.x == 0) { my_stamp[8] = (unsigned long long)total; my_stamp[9] = 1ull; }
}

// (four survivors x 16 pixels per step.  The survivor-column forms of round 3 -- 16 survivors per step, DPP row scans, MFMA reduction:
// 29 % fewer VALU instructions but 112 VGPRs = 4 waves per SIMD, and slower -- left the library in round 4; they are in the history at
// commit 809fd4b and described in DESIGN section 6)
template <bool DET>
__global__ __launch_bounds__(256) void k_composite_bwd_rows(int tiles, int W, int H, int gx, const int2 *__restrict__ ranges,
                                                             const uint32_t *__restrict__ ids_sorted,
                                                             const unsigned long long *__restrict__ bbits, const float4 *__restrict__ recA,
                                                             const float4 *__restrict__ recB, const float2 *__restrict__ recC,
                                                             uint32_t null_rec, const int *__restrict__ seg_offset,
                                                             const int *__restrict__ slot_tile, const float4 *__restrict__ ckpt,
                                                             const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
                                                             const float *__restrict__ out_color, const float *__restrict__ dL_dpix,
                                                             float *__restrict__ acc, float *__restrict__ det) {
    composite_bwd_body<DET>(tiles, W, H, gx, ranges, ids_sorted, bbits, recA, recB, recC, null_rec, seg_offset, slot_tile, ckpt, final_T,
                            n_contrib, out_color, dL_dpix, acc, det, nullptr);
}

// K7 for ALL views of a step in one launch (blockIdx.y = view), preceded by one launch that clears every view's records
constexpr int B2_MAX_VIEWS = 8;
struct B2View {
    const int2 *ranges;
    const uint32_t *ids_sorted;
    const unsigned long long *bbits;
    const float4 *recA, *recB;
    const float2 *recC;
    const int *seg_offset, *slot_tile;
    const float4 *ckpt;
    const float *final_T;
    const uint32_t *n_contrib;
    const float *out_color, *dL_dpix;
    float *acc;
    uint32_t R;
    float *det;        // bit-reproducible mode: the (entry, block) records behind the per-Gaussian ones, else NULL
};
struct B2Table { B2View v[B2_MAX_VIEWS]; unsigned long long *stamp; const uint32_t *valid; };
__global__ __launch_bounds__(256, 8) void k_composite_bwd_rows_views(int tiles, int W, int H, int gx, B2Table tab) {
    if (tab.valid && *tab.valid == 0u) return;     // (a forward launched on faith whose counts did not fit: its chunks hold nothing)
    const B2View &w = tab.v[blockIdx.y];
    composite_bwd_body<false>(tiles, W, H, gx, w.ranges, w.ids_sorted, w.bbits, w.recA, w.recB, w.recC, w.R, w.seg_offset, w.slot_tile, w.ckpt,
                              w.final_T, w.n_contrib, w.out_color, w.dL_dpix, w.acc, nullptr, tab.stamp);
}
// the same launch in the bit-reproducible mode (round 6: the batched and the recorded step take the mode too, so that an eager and a
// replayed step can be compared bit for bit)
__global__ __launch_bounds__(256) void k_composite_bwd_rows_views_det(int tiles, int W, int H, int gx, B2Table tab) {
    if (tab.valid && *tab.valid == 0u) return;
    const B2View &w = tab.v[blockIdx.y];
    composite_bwd_body<true>(tiles, W, H, gx, w.ranges, w.ids_sorted, w.bbits, w.recA, w.recB, w.recC, w.R, w.seg_offset, w.slot_tile, w.ckpt,
                             w.final_T, w.n_contrib, w.out_color, w.dL_dpix, w.acc, w.det, nullptr);
}
__global__ __launch_bounds__(256) void k_zero_det_views(B2Table tab) {
    if (tab.valid && *tab.valid == 0u) return;
    const B2View &w = tab.v[blockIdx.y];
    float4 *p = reinterpret_cast<float4 *>(w.det);
    const int64_t n4 = ((int64_t)(w.R > 0 ? w.R : 1) * 16 * 9 + 3) / 4;        // (det_bytes() is a multiple of 256: the tail is ours)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ __launch_bounds__(256) void k_zero_acc_views(int64_t n4, B2Table tab) {
    float4 *p = reinterpret_cast<float4 *>(tab.v[blockIdx.y].acc);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// DET mode, second half: Gaussian i sums the records of its tile instances in emission order (tiles y-major, x; quadrants
// 0..3): a fixed order, whatever the scheduling of K7.  The instance of Gaussian i in a tile's sorted list is found by
// binary search on the unique (depth bits, id) key.
__global__ __launch_bounds__(256) void k_det_reduce(int P, Cam cam, const float2 *__restrict__ xy, const float *__restrict__ depth,
                                                     const int32_t *__restrict__ radii, const int2 *__restrict__ ranges,
                                                     const uint64_t *__restrict__ keys_sorted, const uint32_t *__restrict__ ids_sorted,
                                                     const float *__restrict__ det, float *__restrict__ acc) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float s[9];
#pragma unroll
    for (int t = 0; t < 9; t++) s[t] = 0.f;
    const int rad = radii[i];
    if (rad > 0) {
        const float2 p = xy[i];
        int minx, miny, maxx, maxy;
        tile_rect(p.x, p.y, rad, cam, minx, miny, maxx, maxy);
        const uint64_t want = ((uint64_t)__float_as_uint(depth[i]) << 32) | (uint32_t)i;
        for (int y = miny; y < maxy; y++)
            for (int x = minx; x < maxx; x++) {
                const int2 rg = ranges[y * cam.gx + x];
                int lo = rg.x, hi = rg.y;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    const uint64_t k = ((keys_sorted[mid] & 0xFFFFFFFFull) << 32) | ids_sorted[mid];
                    if (k < want) lo = mid + 1; else hi = mid;
                }
                for (int q = 0; q < 16; q++)
#pragma unroll
                    for (int t = 0; t < 9; t++) s[t] += det[((size_t)lo * 16 + q) * 9 + t];
            }
    }
#pragma unroll
    for (int t = 0; t < 9; t++) acc[(size_t)i * ACC_STRIDE + t] = s[t];
}

struct DetView { Cam cam; const float2 *xy; const float *depth; const int32_t *radii; const int2 *ranges; const uint64_t *keys_sorted;
                 const uint32_t *ids_sorted; const float *det; float *acc; };
struct DetTable { DetView v[B2_MAX_VIEWS]; const uint32_t *valid; };
__global__ __launch_bounds__(256) void k_det_reduce_views(int P, DetTable tab) {
    if (tab.valid && *tab.valid == 0u) return;
    const DetView &w = tab.v[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float s[9];
#pragma unroll
    for (int t = 0; t < 9; t++) s[t] = 0.f;
    const int rad = w.radii[i];
    if (rad > 0) {
        const float2 p = w.xy[i];
        int minx, miny, maxx, maxy;
        tile_rect(p.x, p.y, rad, w.cam, minx, miny, maxx, maxy);
        const uint64_t want = ((uint64_t)__float_as_uint(w.depth[i]) << 32) | (uint32_t)i;
        for (int y = miny; y < maxy; y++)
            for (int x = minx; x < maxx; x++) {
                const int2 rg = w.ranges[y * w.cam.gx + x];
                int lo = rg.x, hi = rg.y;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    const uint64_t k = ((w.keys_sorted[mid] & 0xFFFFFFFFull) << 32) | w.ids_sorted[mid];
                    if (k < want) lo = mid + 1; else hi = mid;
                }
                for (int q = 0; q < 16; q++)
#pragma unroll
                    for (int t = 0; t < 9; t++) s[t] += w.det[((size_t)lo * 16 + q) * 9 + t];
            }
    }
#pragma unroll
    for (int t = 0; t < 9; t++) w.acc[(size_t)i * ACC_STRIDE + t] = s[t];
}

// ------------------------------------------------------------------------------------------- K8
// K7's per-Gaussian record (round 6) holds the MOMENTS of m = G dL/dalpha over the pixels the Gaussian was blended at, about its centre:
// 0 Mx  1 My  2 Mxx  3 Mxy  4 Myy  5 M0  6..8 dL/dcolour.  With conic (a, b, c) and opacity o the pixel-level gradients upstream sums
// pixel by pixel are linear in them:  dL/dmean2D = -0.5 o (a Mx + b My, c My + b Mx)  (pixel units),  dL/dconic = -0.5 o (Mxx, Mxy, Myy),
// dL/dopacity = M0.  In place: a9[0..4] become (dmean2D.x, dmean2D.y, dconic.a, dconic.b, dconic.c), a9[5..8] stay.
__device__ __forceinline__ void moments_to_gradients(float (&a9)[9], const float4 co) {
    const float h = -0.5f * co.w;
    const float mx = a9[0], my = a9[1];
    a9[0] = h * (co.x * mx + co.y * my);
    a9[1] = h * (co.z * my + co.y * mx);
    a9[2] *= h; a9[3] *= h; a9[4] *= h;
}
// CSPLAT_SCRATCH_ZEROED: the record K8 has just read goes back to zero (12 of its 16 floats: the 9 in use, as three 16-byte stores)
__device__ __forceinline__ void clear_record(const float *acc, int i) {
    float4 *p = reinterpret_cast<float4 *>(const_cast<float *>(acc) + (size_t)i * ACC_STRIDE);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    p[0] = z; p[1] = z; p[2] = z;
}
template <bool STAGE, int NT>
__global__ __launch_bounds__(NT) void k_preprocess_bwd(int P, int D, int M, const float *__restrict__ means3D,
                                                         const float *__restrict__ shs, const float *__restrict__ scales,
                                                         float scale_mod, const float *__restrict__ rotations,
                                                         int use_precomp_cov, Cam cam, Geom g,
                                                         const int32_t *__restrict__ radii, const float *__restrict__ acc,
                                                         float *__restrict__ dL_dmean2D, float *__restrict__ dL_dconic,
                                                         float *__restrict__ dL_dopacity, float *__restrict__ dL_dcolor,
                                                         float *__restrict__ dL_dmean3D, float *__restrict__ dL_dcov3D,
                                                         float *__restrict__ dL_dsh, float *__restrict__ dL_dscale,
                                                         float *__restrict__ dL_drot, unsigned accmask) {
    // accmask (CSPLAT_ACC_*): outputs that are ADDED to instead of written -- several views of one step share the
    // gradient buffer of a shared parameter (csplat_backward_views), which replaces autograd's per-view temporaries
    // and its V-1 summation launches per parameter.
#define PUT(ptr, idx, val, bit) do { float *p_ = (ptr) + (idx); *p_ = (accmask & (bit)) ? *p_ + (val) : (val); } while (0)
    // STAGE: SH coefficients in / SH gradients out go through LDS so that HBM sees contiguous 16-byte accesses (the
    // lane-per-Gaussian 4-byte stores at a 192-byte stride wrote 2.7x the algorithmic bytes)
    __shared__ float s_in[STAGE ? NT * SH_ROW : 1];
    __shared__ float s_out[STAGE ? NT * SH_ROW : 1];
    const int i = blockIdx.x * NT + threadIdx.x;
    const int rows = min(NT, P - blockIdx.x * NT);
    if (STAGE) {
        stage_sh_rows<NT>(shs + (size_t)blockIdx.x * NT * 48, rows, s_in);
        for (int k = 0; k < 48; k++) s_out[threadIdx.x * SH_ROW + k] = 0.f;
        __syncthreads();
    }
    if (i < P) {
    const bool vis = radii[i] > 0;
    float a9[9];
#pragma unroll
    for (int k = 0; k < 9; k++) a9[k] = vis ? acc[(size_t)i * ACC_STRIDE + k] : 0.f;
    if (vis && (accmask & CSPLAT_SCRATCH_ZEROED)) clear_record(acc, i);      // (consumed: the caller's buffer is all zero again for its next step)
    moments_to_gradients(a9, vis ? g.conic_opacity[i] : make_float4(0.f, 0.f, 0.f, 0.f));
    a9[0] *= (float)cam.W; a9[1] *= (float)cam.H;      // (K7 leaves dL/dmean2D without the pixel <- NDC factors 2 * 0.5 W, 2 * 0.5 H)
    dL_dmean2D[3 * i] = a9[0]; dL_dmean2D[3 * i + 1] = a9[1]; dL_dmean2D[3 * i + 2] = 0.f;
    dL_dconic[4 * i] = a9[2]; dL_dconic[4 * i + 1] = a9[3]; dL_dconic[4 * i + 2] = 0.f; dL_dconic[4 * i + 3] = a9[4];
    PUT(dL_dopacity, i, a9[5], CSPLAT_ACC_OPACITY);
    PUT(dL_dcolor, 3 * i, a9[6], CSPLAT_ACC_COLOR); PUT(dL_dcolor, 3 * i + 1, a9[7], CSPLAT_ACC_COLOR);
    PUT(dL_dcolor, 3 * i + 2, a9[8], CSPLAT_ACC_COLOR);

    float dmean[3] = {0.f, 0.f, 0.f};
    float g6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!vis) {
#pragma unroll
        for (int k = 0; k < 3; k++) PUT(dL_dmean3D, 3 * i + k, 0.f, CSPLAT_ACC_MEAN3D);
#pragma unroll
        for (int k = 0; k < 6; k++) PUT(dL_dcov3D, 6 * i + k, 0.f, CSPLAT_ACC_COV3D);
        if (dL_dsh && !STAGE) for (int k = 0; k < M * 3; k++) dL_dsh[(size_t)i * M * 3 + k] = 0.f;
        if (dL_dscale) for (int k = 0; k < 3; k++) PUT(dL_dscale, 3 * i + k, 0.f, CSPLAT_ACC_SCALE);
        if (dL_drot) for (int k = 0; k < 4; k++) PUT(dL_drot, 4 * i + k, 0.f, CSPLAT_ACC_ROT);
    } else {
    const float p[3] = {means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]};
    const float *view = cam.view, *proj = cam.proj;

    // ---- conic -> cov2D -> cov3D and view-space mean
    {
        float pv[3];
        view_point(p, view, pv);
        ProjJac pj;
        proj_jacobian(pv, cam, pj);
        float c6[6];
#pragma unroll
        for (int k = 0; k < 6; k++) c6[k] = g.cov3D[6 * i + k];
        float a, b, c;
        cov2d_from_cov3d(c6, pj, a, b, c);
        const float denom = a * c - b * b;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        const float gcx = a9[2], gcy = a9[3], gcz = a9[4];
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        const float *t0 = pj.t0, *t1 = pj.t1;
        if (denom2inv != 0.f) {
            dL_da = denom2inv * (-c * c * gcx + 2.f * b * c * gcy + (denom - a * c) * gcz);
            dL_dc = denom2inv * (-a * a * gcz + 2.f * a * b * gcy + (denom - a * c) * gcx);
            dL_db = denom2inv * 2.f * (b * c * gcx - (denom + 2.f * b * b) * gcy + a * b * gcz);
            g6[0] = t0[0] * t0[0] * dL_da + t0[0] * t1[0] * dL_db + t1[0] * t1[0] * dL_dc;
            g6[3] = t0[1] * t0[1] * dL_da + t0[1] * t1[1] * dL_db + t1[1] * t1[1] * dL_dc;
            g6[5] = t0[2] * t0[2] * dL_da + t0[2] * t1[2] * dL_db + t1[2] * t1[2] * dL_dc;
            g6[1] = 2.f * t0[0] * t0[1] * dL_da + (t0[0] * t1[1] + t0[1] * t1[0]) * dL_db + 2.f * t1[0] * t1[1] * dL_dc;
            g6[2] = 2.f * t0[0] * t0[2] * dL_da + (t0[0] * t1[2] + t0[2] * t1[0]) * dL_db + 2.f * t1[0] * t1[2] * dL_dc;
            g6[4] = 2.f * t0[2] * t0[1] * dL_da + (t0[1] * t1[2] + t0[2] * t1[1]) * dL_db + 2.f * t1[1] * t1[2] * dL_dc;
        }
        const float Vm[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
        float dT0[3], dT1[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const float Vt0 = Vm[r][0] * t0[0] + Vm[r][1] * t0[1] + Vm[r][2] * t0[2];
            const float Vt1 = Vm[r][0] * t1[0] + Vm[r][1] * t1[1] + Vm[r][2] * t1[2];
            dT0[r] = 2.f * Vt0 * dL_da + Vt1 * dL_db;
            dT1[r] = 2.f * Vt1 * dL_dc + Vt0 * dL_db;
        }
        const float dJ00 = view[0] * dT0[0] + view[4] * dT0[1] + view[8] * dT0[2];
        const float dJ02 = view[2] * dT0[0] + view[6] * dT0[1] + view[10] * dT0[2];
        const float dJ11 = view[1] * dT1[0] + view[5] * dT1[1] + view[9] * dT1[2];
        const float dJ12 = view[2] * dT1[0] + view[6] * dT1[1] + view[10] * dT1[2];
        const float tz = 1.f / pj.tz, tz2 = tz * tz, tz3 = tz2 * tz;
        const float xg = pj.x_in ? 1.f : 0.f, yg = pj.y_in ? 1.f : 0.f;
        const float dtx = xg * -cam.fx * tz2 * dJ02;
        const float dty = yg * -cam.fy * tz2 * dJ12;
        const float dtz = -cam.fx * tz2 * dJ00 - cam.fy * tz2 * dJ11 + (2.f * cam.fx * pj.tx) * tz3 * dJ02 +
                          (2.f * cam.fy * pj.ty) * tz3 * dJ12;
        dmean[0] += view[0] * dtx + view[1] * dty + view[2] * dtz;
        dmean[1] += view[4] * dtx + view[5] * dty + view[6] * dtz;
        dmean[2] += view[8] * dtx + view[9] * dty + view[10] * dtz;
    }
    // ---- mean2D (NDC) -> mean3D
    {
        const float hw = proj[3] * p[0] + proj[7] * p[1] + proj[11] * p[2] + proj[15];
        const float m_w = 1.0f / (hw + 0.0000001f);
        const float mul1 = (proj[0] * p[0] + proj[4] * p[1] + proj[8] * p[2] + proj[12]) * m_w * m_w;
        const float mul2 = (proj[1] * p[0] + proj[5] * p[1] + proj[9] * p[2] + proj[13]) * m_w * m_w;
        const float gx2 = a9[0], gy2 = a9[1];
        dmean[0] += (proj[0] * m_w - proj[3] * mul1) * gx2 + (proj[1] * m_w - proj[3] * mul2) * gy2;
        dmean[1] += (proj[4] * m_w - proj[7] * mul1) * gx2 + (proj[5] * m_w - proj[7] * mul2) * gy2;
        dmean[2] += (proj[8] * m_w - proj[11] * mul1) * gx2 + (proj[9] * m_w - proj[11] * mul2) * gy2;
    }
    // ---- colour -> SH (+ view direction -> mean3D)
    if (shs && dL_dsh) {
        const float *sh = STAGE ? (const float *)(s_in + threadIdx.x * SH_ROW) : shs + (size_t)i * M * 3;
        float *gsh = STAGE ? s_out + threadIdx.x * SH_ROW : dL_dsh + (size_t)i * M * 3;
        const uint32_t cl = g.clamped[i];
        const float vx = p[0] - cam.campos[0], vy = p[1] - cam.campos[1], vz = p[2] - cam.campos[2];
        const float sum2 = vx * vx + vy * vy + vz * vz;
        const float len = sqrtf(sum2);
        const float x = vx / len, y = vy / len, z = vz / len;
        float ddx = 0.f, ddy = 0.f, ddz = 0.f;
        for (int k = (D + 1) * (D + 1) * 3; k < M * 3; k++) gsh[k] = 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float dRGB = ((cl >> ch) & 1u) ? 0.f : a9[6 + ch];
            float dx_ = 0.f, dy_ = 0.f, dz_ = 0.f;
#define S(k) sh[(k) * 3 + ch]
#define GS(k) gsh[(k) * 3 + ch]
            GS(0) = SH_C0 * dRGB;
            if (D > 0) {
                GS(1) = -SH_C1 * y * dRGB;
                GS(2) = SH_C1 * z * dRGB;
                GS(3) = -SH_C1 * x * dRGB;
                dx_ = -SH_C1 * S(3); dy_ = -SH_C1 * S(1); dz_ = SH_C1 * S(2);
                if (D > 1) {
                    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    GS(4) = SH_C2[0] * xy * dRGB;
                    GS(5) = SH_C2[1] * yz * dRGB;
                    GS(6) = SH_C2[2] * (2.f * zz - xx - yy) * dRGB;
                    GS(7) = SH_C2[3] * xz * dRGB;
                    GS(8) = SH_C2[4] * (xx - yy) * dRGB;
                    dx_ += SH_C2[0] * y * S(4) + SH_C2[2] * 2.f * -x * S(6) + SH_C2[3] * z * S(7) + SH_C2[4] * 2.f * x * S(8);
                    dy_ += SH_C2[0] * x * S(4) + SH_C2[1] * z * S(5) + SH_C2[2] * 2.f * -y * S(6) + SH_C2[4] * 2.f * -y * S(8);
                    dz_ += SH_C2[1] * y * S(5) + SH_C2[2] * 4.f * z * S(6) + SH_C2[3] * x * S(7);
                    if (D > 2) {
                        GS(9) = SH_C3[0] * y * (3.f * xx - yy) * dRGB;
                        GS(10) = SH_C3[1] * xy * z * dRGB;
                        GS(11) = SH_C3[2] * y * (4.f * zz - xx - yy) * dRGB;
                        GS(12) = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * dRGB;
                        GS(13) = SH_C3[4] * x * (4.f * zz - xx - yy) * dRGB;
                        GS(14) = SH_C3[5] * z * (xx - yy) * dRGB;
                        GS(15) = SH_C3[6] * x * (xx - 3.f * yy) * dRGB;
                        dx_ += SH_C3[0] * S(9) * 6.f * xy + SH_C3[1] * S(10) * yz + SH_C3[2] * S(11) * -2.f * xy +
                               SH_C3[3] * S(12) * -6.f * xz + SH_C3[4] * S(13) * (-3.f * xx + 4.f * zz - yy) +
                               SH_C3[5] * S(14) * 2.f * xz + SH_C3[6] * S(15) * 3.f * (xx - yy);
                        dy_ += SH_C3[0] * S(9) * 3.f * (xx - yy) + SH_C3[1] * S(10) * xz +
                               SH_C3[2] * S(11) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * S(12) * -6.f * yz +
                               SH_C3[4] * S(13) * -2.f * xy + SH_C3[5] * S(14) * -2.f * yz + SH_C3[6] * S(15) * -6.f * xy;
                        dz_ += SH_C3[1] * S(10) * xy + SH_C3[2] * S(11) * 8.f * yz +
                               SH_C3[3] * S(12) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * S(13) * 8.f * xz +
                               SH_C3[5] * S(14) * (xx - yy);
                    }
                }
            }
#undef S
#undef GS
            ddx += dx_ * dRGB; ddy += dy_ * dRGB; ddz += dz_ * dRGB;
        }
        const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
        dmean[0] += ((sum2 - vx * vx) * ddx - vy * vx * ddy - vz * vx * ddz) * invsum32;
        dmean[1] += (-vx * vy * ddx + (sum2 - vy * vy) * ddy - vz * vy * ddz) * invsum32;
        dmean[2] += (-vx * vz * ddx - vy * vz * ddy + (sum2 - vz * vz) * ddz) * invsum32;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) PUT(dL_dmean3D, 3 * i + k, dmean[k], CSPLAT_ACC_MEAN3D);
#pragma unroll
    for (int k = 0; k < 6; k++) PUT(dL_dcov3D, 6 * i + k, g6[k], CSPLAT_ACC_COV3D);

    // ---- cov3D -> scale, quaternion
    if (!use_precomp_cov && dL_dscale && dL_drot) {
        const float q[4] = {rotations[4 * i], rotations[4 * i + 1], rotations[4 * i + 2], rotations[4 * i + 3]};
        float R[3][3];
        quat_to_rot(q, R);
        const float s[3] = {scale_mod * scales[3 * i], scale_mod * scales[3 * i + 1], scale_mod * scales[3 * i + 2]};
        const float dS[3][3] = {{g6[0], 0.5f * g6[1], 0.5f * g6[2]},
                                {0.5f * g6[1], g6[3], 0.5f * g6[4]},
                                {0.5f * g6[2], 0.5f * g6[4], g6[5]}};
        float dA[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int k = 0; k < 3; k++)
                dA[r][k] = 2.f * (dS[r][0] * R[0][k] * s[k] + dS[r][1] * R[1][k] * s[k] + dS[r][2] * R[2][k] * s[k]);
#pragma unroll
        for (int k = 0; k < 3; k++) PUT(dL_dscale, 3 * i + k, dA[0][k] * R[0][k] + dA[1][k] * R[1][k] + dA[2][k] * R[2][k], CSPLAT_ACC_SCALE);
        float dR[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int k = 0; k < 3; k++) dR[r][k] = dA[r][k] * s[k];
        const float qr = q[0], qx = q[1], qy = q[2], qz = q[3];
        const float dq0 = 2.f * (-qz * dR[0][1] + qy * dR[0][2] + qz * dR[1][0] - qx * dR[1][2] - qy * dR[2][0] + qx * dR[2][1]);
        const float dq1 = 2.f * (qy * dR[0][1] + qz * dR[0][2] + qy * dR[1][0] - 2.f * qx * dR[1][1] - qr * dR[1][2] +
                                    qz * dR[2][0] + qr * dR[2][1] - 2.f * qx * dR[2][2]);
        const float dq2 = 2.f * (-2.f * qy * dR[0][0] + qx * dR[0][1] + qr * dR[0][2] + qx * dR[1][0] + qz * dR[1][2] -
                                    qr * dR[2][0] + qz * dR[2][1] - 2.f * qy * dR[2][2]);
        const float dq3 = 2.f * (-2.f * qz * dR[0][0] - qr * dR[0][1] + qx * dR[0][2] + qr * dR[1][0] - 2.f * qz * dR[1][1] +
                                    qy * dR[1][2] + qx * dR[2][0] + qy * dR[2][1]);
        PUT(dL_drot, 4 * i, dq0, CSPLAT_ACC_ROT); PUT(dL_drot, 4 * i + 1, dq1, CSPLAT_ACC_ROT);
        PUT(dL_drot, 4 * i + 2, dq2, CSPLAT_ACC_ROT); PUT(dL_drot, 4 * i + 3, dq3, CSPLAT_ACC_ROT);
    }
    }   // visible
    }   // i < P
    if (STAGE) {   // coalesced 16-byte stores of the workgroup's SH gradients
        __syncthreads();
        float4 *dst4 = reinterpret_cast<float4 *>(dL_dsh + (size_t)blockIdx.x * NT * 48);
        for (int t = threadIdx.x; t < rows * 12; t += NT) {
            const int row = t / 12, c = (t - row * 12) * 4;
            const float *sp = s_out + row * SH_ROW + c;
            float4 o = make_float4(sp[0], sp[1], sp[2], sp[3]);
            if (accmask & CSPLAT_ACC_SH) { const float4 u = dst4[t]; o.x += u.x; o.y += u.y; o.z += u.z; o.w += u.w; }
            dst4[t] = o;
        }
    }
#undef PUT
}

// K8 for ALL views of a step in one launch (csplat_backward_views).  The per-view kernels above add into shared gradient
// buffers and therefore run one after the other behind the concurrent K7s (a tail of ~27 us per view).  Here a thread
// keeps its Gaussian and loops over the views: inputs and SH rows are read once, gradients of parameters that all views
// share are summed in registers (SH: in the LDS rows) and written once, per-view outputs (mean2D, conic, and mean3D /
// rotation when every view has its own deformed copy) are written per view.  Same arithmetic per view as k_preprocess_bwd.
constexpr int K8_MAX_VIEWS = 8;
struct K8View {
    Cam cam;
    Geom g;
    const int32_t *radii;
    const float *acc, *means3D, *rotations;
    float *dL_dmean2D, *dL_dconic, *dL_dopacity, *dL_dcolor, *dL_dmean3D, *dL_dcov3D, *dL_dscale, *dL_drot;
    unsigned accmask;
};
struct K8Table {
    int n;
    unsigned sharedmask;   // CSPLAT_ACC_* bits of the outputs whose buffer is the same in every view
    const uint32_t *valid; // (csplat_forward_views_faith) 0 there: the forward left the views untouched -- nothing to differentiate
    K8View v[K8_MAX_VIEWS];
};

// VL lanes per Gaussian, lane vl takes the views vl, vl + VL, ...: with one lane per Gaussian the launch has P / 64 = 1564 waves (1.5 per
// SIMD) that each walk V dependent load -> compute rounds; with VL = 4 it has four times the waves and (V <= 4) one round each.  The
// sums over the views of the shared-parameter gradients cross the VL lanes with quad DPP adds (fixed association), the SH rows in LDS.
template <int NT, int VL>
__global__ __launch_bounds__(NT) void k_preprocess_bwd_views(int P, int D, int M, const float *__restrict__ shs,
                                                               const float *__restrict__ scales, float scale_mod,
                                                               int use_precomp_cov, float *__restrict__ dL_dsh, K8Table tab, int block0) {
    constexpr bool STAGE = true;
    if (tab.valid && *tab.valid == 0u) return;
    // (block0: first workgroup of a Gaussian-range SLICE of the launch -- csplat_backward_views_parts: the gradient rows of a finished slice
    //  can leave for the other ranks while the next slice computes)
    const int bx = (int)blockIdx.x + block0;
    const unsigned smask = tab.sharedmask;
    // shared output: add to the thread's running sum; per-view output: write (or add, by that view's accmask)
#define PUTL(local, ptr, idx, val, bit)                                                    \
    do {                                                                                   \
        if (smask & (bit)) (local) += (val);                                               \
        else { float *p_ = (ptr) + (idx); *p_ = (accmask & (bit)) ? *p_ + (val) : (val); } \
    } while (0)
    static_assert(VL == 1 || VL == 4, "one lane or one quad per Gaussian");
    constexpr int NG = NT / VL;                   // Gaussians per workgroup
    __shared__ float s_in[STAGE ? NG * SH_ROW : 1];
    __shared__ float s_out[STAGE ? NT * SH_ROW : 1];
    const int gi = threadIdx.x / VL, vl = threadIdx.x % VL;
    const int i = bx * NG + gi;
    const int rows = min(NG, P - bx * NG);
    if (STAGE) {
        stage_sh_rows<NT>(shs + (size_t)bx * NG * 48, rows, s_in);
        for (int k = 0; k < 48; k++) s_out[threadIdx.x * SH_ROW + k] = 0.f;
        __syncthreads();
    }
    if (i < P) {
    float L_op = 0.f, L_col[3] = {0.f, 0.f, 0.f}, L_m3[3] = {0.f, 0.f, 0.f}, L_c6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float L_sc[3] = {0.f, 0.f, 0.f}, L_rt[4] = {0.f, 0.f, 0.f, 0.f};
    // the per-view record (radius, the nine accumulated pixel-level gradients) of view vi+1 is requested while view vi is
    // processed: otherwise the thread walks V dependent load -> compute rounds
    bool vis_n = false;
    float a9_n[9];
    float4 co_n = make_float4(0.f, 0.f, 0.f, 0.f);       // (the view's conic + opacity travel with its record: moments_to_gradients)
    if (vl < tab.n) {
        vis_n = tab.v[vl].radii[i] > 0;
        co_n = tab.v[vl].g.conic_opacity[i];
#pragma unroll
        for (int k = 0; k < 9; k++) a9_n[k] = tab.v[vl].acc[(size_t)i * ACC_STRIDE + k];
        if (vis_n && (tab.v[vl].accmask & CSPLAT_SCRATCH_ZEROED)) clear_record(tab.v[vl].acc, i);
    }
    for (int vi = vl; vi < tab.n; vi += VL) {
    const K8View &w = tab.v[vi];
    const Cam cam = w.cam;
    const Geom g = w.g;
    const int32_t *radii = w.radii;
    const float *acc = w.acc, *means3D = w.means3D, *rotations = w.rotations;
    float *dL_dmean2D = w.dL_dmean2D, *dL_dconic = w.dL_dconic, *dL_dopacity = w.dL_dopacity, *dL_dcolor = w.dL_dcolor;
    float *dL_dmean3D = w.dL_dmean3D, *dL_dcov3D = w.dL_dcov3D, *dL_dscale = w.dL_dscale, *dL_drot = w.dL_drot;
    const unsigned accmask = w.accmask;
    (void)radii; (void)acc;
    const bool vis = vis_n;
    float a9[9];
#pragma unroll
    for (int k = 0; k < 9; k++) a9[k] = vis ? a9_n[k] : 0.f;
    moments_to_gradients(a9, vis ? co_n : make_float4(0.f, 0.f, 0.f, 0.f));
    a9[0] *= (float)cam.W; a9[1] *= (float)cam.H;      // (see k_preprocess_bwd)
    if (vi + VL < tab.n) {
        const K8View &wn = tab.v[vi + VL];
        vis_n = wn.radii[i] > 0;
#pragma unroll
        for (int k = 0; k < 9; k++) a9_n[k] = wn.acc[(size_t)i * ACC_STRIDE + k];
        co_n = wn.g.conic_opacity[i];
        if (vis_n && (wn.accmask & CSPLAT_SCRATCH_ZEROED)) clear_record(wn.acc, i);
    }
    dL_dmean2D[3 * i] = a9[0]; dL_dmean2D[3 * i + 1] = a9[1]; dL_dmean2D[3 * i + 2] = 0.f;
    dL_dconic[4 * i] = a9[2]; dL_dconic[4 * i + 1] = a9[3]; dL_dconic[4 * i + 2] = 0.f; dL_dconic[4 * i + 3] = a9[4];
    PUTL(L_op, dL_dopacity, i, a9[5], CSPLAT_ACC_OPACITY);
    PUTL(L_col[0], dL_dcolor, 3 * i, a9[6], CSPLAT_ACC_COLOR); PUTL(L_col[1], dL_dcolor, 3 * i + 1, a9[7], CSPLAT_ACC_COLOR);
    PUTL(L_col[2], dL_dcolor, 3 * i + 2, a9[8], CSPLAT_ACC_COLOR);

    float dmean[3] = {0.f, 0.f, 0.f};
    float g6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!vis) {
#pragma unroll
        for (int k = 0; k < 3; k++) PUTL(L_m3[k], dL_dmean3D, 3 * i + k, 0.f, CSPLAT_ACC_MEAN3D);
#pragma unroll
        for (int k = 0; k < 6; k++) PUTL(L_c6[k], dL_dcov3D, 6 * i + k, 0.f, CSPLAT_ACC_COV3D);
        if (dL_dscale)
#pragma unroll
            for (int k = 0; k < 3; k++) PUTL(L_sc[k], dL_dscale, 3 * i + k, 0.f, CSPLAT_ACC_SCALE);
        if (dL_drot)
#pragma unroll
            for (int k = 0; k < 4; k++) PUTL(L_rt[k], dL_drot, 4 * i + k, 0.f, CSPLAT_ACC_ROT);
    } else {
    const float p[3] = {means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]};
    const float *view = cam.view, *proj = cam.proj;

    // ---- conic -> cov2D -> cov3D and view-space mean
    {
        float pv[3];
        view_point(p, view, pv);
        ProjJac pj;
        proj_jacobian(pv, cam, pj);
        float c6[6];
#pragma unroll
        for (int k = 0; k < 6; k++) c6[k] = g.cov3D[6 * i + k];
        float a, b, c;
        cov2d_from_cov3d(c6, pj, a, b, c);
        const float denom = a * c - b * b;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        const float gcx = a9[2], gcy = a9[3], gcz = a9[4];
        float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
        const float *t0 = pj.t0, *t1 = pj.t1;
        if (denom2inv != 0.f) {
            dL_da = denom2inv * (-c * c * gcx + 2.f * b * c * gcy + (denom - a * c) * gcz);
            dL_dc = denom2inv * (-a * a * gcz + 2.f * a * b * gcy + (denom - a * c) * gcx);
            dL_db = denom2inv * 2.f * (b * c * gcx - (denom + 2.f * b * b) * gcy + a * b * gcz);
            g6[0] = t0[0] * t0[0] * dL_da + t0[0] * t1[0] * dL_db + t1[0] * t1[0] * dL_dc;
            g6[3] = t0[1] * t0[1] * dL_da + t0[1] * t1[1] * dL_db + t1[1] * t1[1] * dL_dc;
            g6[5] = t0[2] * t0[2] * dL_da + t0[2] * t1[2] * dL_db + t1[2] * t1[2] * dL_dc;
            g6[1] = 2.f * t0[0] * t0[1] * dL_da + (t0[0] * t1[1] + t0[1] * t1[0]) * dL_db + 2.f * t1[0] * t1[1] * dL_dc;
            g6[2] = 2.f * t0[0] * t0[2] * dL_da + (t0[0] * t1[2] + t0[2] * t1[0]) * dL_db + 2.f * t1[0] * t1[2] * dL_dc;
            g6[4] = 2.f * t0[2] * t0[1] * dL_da + (t0[1] * t1[2] + t0[2] * t1[1]) * dL_db + 2.f * t1[1] * t1[2] * dL_dc;
        }
        const float Vm[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
        float dT0[3], dT1[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const float Vt0 = Vm[r][0] * t0[0] + Vm[r][1] * t0[1] + Vm[r][2] * t0[2];
            const float Vt1 = Vm[r][0] * t1[0] + Vm[r][1] * t1[1] + Vm[r][2] * t1[2];
            dT0[r] = 2.f * Vt0 * dL_da + Vt1 * dL_db;
            dT1[r] = 2.f * Vt1 * dL_dc + Vt0 * dL_db;
        }
        const float dJ00 = view[0] * dT0[0] + view[4] * dT0[1] + view[8] * dT0[2];
        const float dJ02 = view[2] * dT0[0] + view[6] * dT0[1] + view[10] * dT0[2];
        const float dJ11 = view[1] * dT1[0] + view[5] * dT1[1] + view[9] * dT1[2];
        const float dJ12 = view[2] * dT1[0] + view[6] * dT1[1] + view[10] * dT1[2];
        const float tz = 1.f / pj.tz, tz2 = tz * tz, tz3 = tz2 * tz;
        const float xg = pj.x_in ? 1.f : 0.f, yg = pj.y_in ? 1.f : 0.f;
        const float dtx = xg * -cam.fx * tz2 * dJ02;
        const float dty = yg * -cam.fy * tz2 * dJ12;
        const float dtz = -cam.fx * tz2 * dJ00 - cam.fy * tz2 * dJ11 + (2.f * cam.fx * pj.tx) * tz3 * dJ02 +
                          (2.f * cam.fy * pj.ty) * tz3 * dJ12;
        dmean[0] += view[0] * dtx + view[1] * dty + view[2] * dtz;
        dmean[1] += view[4] * dtx + view[5] * dty + view[6] * dtz;
        dmean[2] += view[8] * dtx + view[9] * dty + view[10] * dtz;
    }
    // ---- mean2D (NDC) -> mean3D
    {
        const float hw = proj[3] * p[0] + proj[7] * p[1] + proj[11] * p[2] + proj[15];
        const float m_w = 1.0f / (hw + 0.0000001f);
        const float mul1 = (proj[0] * p[0] + proj[4] * p[1] + proj[8] * p[2] + proj[12]) * m_w * m_w;
        const float mul2 = (proj[1] * p[0] + proj[5] * p[1] + proj[9] * p[2] + proj[13]) * m_w * m_w;
        const float gx2 = a9[0], gy2 = a9[1];
        dmean[0] += (proj[0] * m_w - proj[3] * mul1) * gx2 + (proj[1] * m_w - proj[3] * mul2) * gy2;
        dmean[1] += (proj[4] * m_w - proj[7] * mul1) * gx2 + (proj[5] * m_w - proj[7] * mul2) * gy2;
        dmean[2] += (proj[8] * m_w - proj[11] * mul1) * gx2 + (proj[9] * m_w - proj[11] * mul2) * gy2;
    }
    // ---- colour -> SH (+ view direction -> mean3D)
    if (shs && dL_dsh) {
        const float *sh = (const float *)(s_in + gi * SH_ROW);
        float *gsh = s_out + threadIdx.x * SH_ROW;
        const uint32_t cl = g.clamped[i];
        const float vx = p[0] - cam.campos[0], vy = p[1] - cam.campos[1], vz = p[2] - cam.campos[2];
        const float sum2 = vx * vx + vy * vy + vz * vz;
        const float len = sqrtf(sum2);
        const float x = vx / len, y = vy / len, z = vz / len;
        float ddx = 0.f, ddy = 0.f, ddz = 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float dRGB = ((cl >> ch) & 1u) ? 0.f : a9[6 + ch];
            float dx_ = 0.f, dy_ = 0.f, dz_ = 0.f;
#define S(k) sh[(k) * 3 + ch]
#define GS(k) gsh[(k) * 3 + ch]
            GS(0) += SH_C0 * dRGB;
            if (D > 0) {
                GS(1) += -SH_C1 * y * dRGB;
                GS(2) += SH_C1 * z * dRGB;
                GS(3) += -SH_C1 * x * dRGB;
                dx_ = -SH_C1 * S(3); dy_ = -SH_C1 * S(1); dz_ = SH_C1 * S(2);
                if (D > 1) {
                    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    GS(4) += SH_C2[0] * xy * dRGB;
                    GS(5) += SH_C2[1] * yz * dRGB;
                    GS(6) += SH_C2[2] * (2.f * zz - xx - yy) * dRGB;
                    GS(7) += SH_C2[3] * xz * dRGB;
                    GS(8) += SH_C2[4] * (xx - yy) * dRGB;
                    dx_ += SH_C2[0] * y * S(4) + SH_C2[2] * 2.f * -x * S(6) + SH_C2[3] * z * S(7) + SH_C2[4] * 2.f * x * S(8);
                    dy_ += SH_C2[0] * x * S(4) + SH_C2[1] * z * S(5) + SH_C2[2] * 2.f * -y * S(6) + SH_C2[4] * 2.f * -y * S(8);
                    dz_ += SH_C2[1] * y * S(5) + SH_C2[2] * 4.f * z * S(6) + SH_C2[3] * x * S(7);
                    if (D > 2) {
                        GS(9) += SH_C3[0] * y * (3.f * xx - yy) * dRGB;
                        GS(10) += SH_C3[1] * xy * z * dRGB;
                        GS(11) += SH_C3[2] * y * (4.f * zz - xx - yy) * dRGB;
                        GS(12) += SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * dRGB;
                        GS(13) += SH_C3[4] * x * (4.f * zz - xx - yy) * dRGB;
                        GS(14) += SH_C3[5] * z * (xx - yy) * dRGB;
                        GS(15) += SH_C3[6] * x * (xx - 3.f * yy) * dRGB;
                        dx_ += SH_C3[0] * S(9) * 6.f * xy + SH_C3[1] * S(10) * yz + SH_C3[2] * S(11) * -2.f * xy +
                               SH_C3[3] * S(12) * -6.f * xz + SH_C3[4] * S(13) * (-3.f * xx + 4.f * zz - yy) +
                               SH_C3[5] * S(14) * 2.f * xz + SH_C3[6] * S(15) * 3.f * (xx - yy);
                        dy_ += SH_C3[0] * S(9) * 3.f * (xx - yy) + SH_C3[1] * S(10) * xz +
                               SH_C3[2] * S(11) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * S(12) * -6.f * yz +
                               SH_C3[4] * S(13) * -2.f * xy + SH_C3[5] * S(14) * -2.f * yz + SH_C3[6] * S(15) * -6.f * xy;
                        dz_ += SH_C3[1] * S(10) * xy + SH_C3[2] * S(11) * 8.f * yz +
                               SH_C3[3] * S(12) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * S(13) * 8.f * xz +
                               SH_C3[5] * S(14) * (xx - yy);
                    }
                }
            }
#undef S
#undef GS
            ddx += dx_ * dRGB; ddy += dy_ * dRGB; ddz += dz_ * dRGB;
        }
        const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
        dmean[0] += ((sum2 - vx * vx) * ddx - vy * vx * ddy - vz * vx * ddz) * invsum32;
        dmean[1] += (-vx * vy * ddx + (sum2 - vy * vy) * ddy - vz * vy * ddz) * invsum32;
        dmean[2] += (-vx * vz * ddx - vy * vz * ddy + (sum2 - vz * vz) * ddz) * invsum32;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) PUTL(L_m3[k], dL_dmean3D, 3 * i + k, dmean[k], CSPLAT_ACC_MEAN3D);
#pragma unroll
    for (int k = 0; k < 6; k++) PUTL(L_c6[k], dL_dcov3D, 6 * i + k, g6[k], CSPLAT_ACC_COV3D);

    // ---- cov3D -> scale, quaternion
    if (!use_precomp_cov && dL_dscale && dL_drot) {
        const float q[4] = {rotations[4 * i], rotations[4 * i + 1], rotations[4 * i + 2], rotations[4 * i + 3]};
        float R[3][3];
        quat_to_rot(q, R);
        const float s[3] = {scale_mod * scales[3 * i], scale_mod * scales[3 * i + 1], scale_mod * scales[3 * i + 2]};
        const float dS[3][3] = {{g6[0], 0.5f * g6[1], 0.5f * g6[2]},
                                {0.5f * g6[1], g6[3], 0.5f * g6[4]},
                                {0.5f * g6[2], 0.5f * g6[4], g6[5]}};
        float dA[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int k = 0; k < 3; k++)
                dA[r][k] = 2.f * (dS[r][0] * R[0][k] * s[k] + dS[r][1] * R[1][k] * s[k] + dS[r][2] * R[2][k] * s[k]);
#pragma unroll
        for (int k = 0; k < 3; k++) PUTL(L_sc[k], dL_dscale, 3 * i + k, dA[0][k] * R[0][k] + dA[1][k] * R[1][k] + dA[2][k] * R[2][k], CSPLAT_ACC_SCALE);
        float dR[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int k = 0; k < 3; k++) dR[r][k] = dA[r][k] * s[k];
        const float qr = q[0], qx = q[1], qy = q[2], qz = q[3];
        const float dq0 = 2.f * (-qz * dR[0][1] + qy * dR[0][2] + qz * dR[1][0] - qx * dR[1][2] - qy * dR[2][0] + qx * dR[2][1]);
        const float dq1 = 2.f * (qy * dR[0][1] + qz * dR[0][2] + qy * dR[1][0] - 2.f * qx * dR[1][1] - qr * dR[1][2] +
                                    qz * dR[2][0] + qr * dR[2][1] - 2.f * qx * dR[2][2]);
        const float dq2 = 2.f * (-2.f * qy * dR[0][0] + qx * dR[0][1] + qr * dR[0][2] + qx * dR[1][0] + qz * dR[1][2] -
                                    qr * dR[2][0] + qz * dR[2][1] - 2.f * qy * dR[2][2]);
        const float dq3 = 2.f * (-2.f * qz * dR[0][0] - qr * dR[0][1] + qx * dR[0][2] + qr * dR[1][0] - 2.f * qz * dR[1][1] +
                                    qy * dR[1][2] + qx * dR[2][0] + qy * dR[2][1]);
        PUTL(L_rt[0], dL_drot, 4 * i, dq0, CSPLAT_ACC_ROT); PUTL(L_rt[1], dL_drot, 4 * i + 1, dq1, CSPLAT_ACC_ROT);
        PUTL(L_rt[2], dL_drot, 4 * i + 2, dq2, CSPLAT_ACC_ROT); PUTL(L_rt[3], dL_drot, 4 * i + 3, dq3, CSPLAT_ACC_ROT);
    }
    }   // visible
    }   // views
    if (VL == 4) {   // the four lanes' sums over their views: ((v0 + v1) + (v2 + v3)) in every lane
        auto quad_sum = [](float v) {
            v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
            v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
            return v;
        };
        L_op = quad_sum(L_op);
#pragma unroll
        for (int k = 0; k < 3; k++) { L_col[k] = quad_sum(L_col[k]); L_m3[k] = quad_sum(L_m3[k]); L_sc[k] = quad_sum(L_sc[k]); }
#pragma unroll
        for (int k = 0; k < 6; k++) L_c6[k] = quad_sum(L_c6[k]);
#pragma unroll
        for (int k = 0; k < 4; k++) L_rt[k] = quad_sum(L_rt[k]);
    }
    if (vl == 0) {   // gradients of the parameters every view shares: one write (added to the buffer only if the first view was asked to)
        const unsigned accmask = tab.v[0].accmask;
        const K8View &w = tab.v[0];
#define PUTS(ptr, idx, val, bit) do { if (smask & (bit)) { float *p_ = (ptr) + (idx); *p_ = (accmask & (bit)) ? *p_ + (val) : (val); } } while (0)
        PUTS(w.dL_dopacity, i, L_op, CSPLAT_ACC_OPACITY);
#pragma unroll
        for (int k = 0; k < 3; k++) PUTS(w.dL_dcolor, 3 * i + k, L_col[k], CSPLAT_ACC_COLOR);
#pragma unroll
        for (int k = 0; k < 3; k++) PUTS(w.dL_dmean3D, 3 * i + k, L_m3[k], CSPLAT_ACC_MEAN3D);
#pragma unroll
        for (int k = 0; k < 6; k++) PUTS(w.dL_dcov3D, 6 * i + k, L_c6[k], CSPLAT_ACC_COV3D);
        if (w.dL_dscale)
#pragma unroll
            for (int k = 0; k < 3; k++) PUTS(w.dL_dscale, 3 * i + k, L_sc[k], CSPLAT_ACC_SCALE);
        if (w.dL_drot)
#pragma unroll
            for (int k = 0; k < 4; k++) PUTS(w.dL_drot, 4 * i + k, L_rt[k], CSPLAT_ACC_ROT);
#undef PUTS
    }
    }   // i < P
    if (STAGE) {   // coalesced 16-byte stores of the workgroup's SH gradients
        __syncthreads();
        float4 *dst4 = reinterpret_cast<float4 *>(dL_dsh + (size_t)bx * NG * 48);
        for (int t = threadIdx.x; t < rows * 12; t += NT) {
            const int row = t / 12, c = (t - row * 12) * 4;
            const float *sp = s_out + row * VL * SH_ROW + c;
            float4 o = make_float4(sp[0], sp[1], sp[2], sp[3]);
#pragma unroll
            for (int v2 = 1; v2 < VL; v2++) {            // the rows of the Gaussian's other view lanes, in lane order
                const float *sq = sp + v2 * SH_ROW;
                o.x += sq[0]; o.y += sq[1]; o.z += sq[2]; o.w += sq[3];
            }
            if (tab.v[0].accmask & CSPLAT_ACC_SH) { const float4 u = dst4[t]; o.x += u.x; o.y += u.y; o.z += u.z; o.w += u.w; }
            dst4[t] = o;
        }
    }
#undef PUTL
}

// ------------------------------------------------------------------------------------------- layouts
enum { G_DEPTH, G_XY, G_CONIC, G_RGB, G_COV3D, G_CLAMPED, G_TOUCHED, G_OFFSETS, G_CUT2, G_SCANTMP, G_PACK, G_NFIELDS };

int64_t max_slots(int64_t R, int tiles) { return R / SEG + tiles + 1; }
size_t geom_offsets(int P, size_t *off) {
    size_t o = 0;
    const size_t n = (size_t)(P > 0 ? P : 1);
    const size_t sz[G_NFIELDS] = {n * 4, n * 8, n * 16, n * 12, n * 24, n * 4, n * 4, n * 4, n * 4,
                                  csplat_scan_temp_bytes(P), n * 48};
    for (int k = 0; k < G_NFIELDS; k++) { off[k] = o; o += align256(sz[k]); }
    return o;
}
Geom geom_view(void *base, int P) {
    size_t off[G_NFIELDS];
    geom_offsets(P, off);
    char *b = (char *)base;
    Geom g;
    g.depth = (float *)(b + off[G_DEPTH]); g.xy = (float2 *)(b + off[G_XY]);
    g.conic_opacity = (float4 *)(b + off[G_CONIC]); g.rgb = (float *)(b + off[G_RGB]);
    g.cov3D = (float *)(b + off[G_COV3D]); g.clamped = (uint32_t *)(b + off[G_CLAMPED]);
    g.tiles_touched = (uint32_t *)(b + off[G_TOUCHED]); g.offsets = (uint32_t *)(b + off[G_OFFSETS]);
    g.cut2 = (float *)(b + off[G_CUT2]);
    g.scan_tmp = (void *)(b + off[G_SCANTMP]);
    g.pack = (float4 *)(b + off[G_PACK]);
    return g;
}
// image: 0 ranges | 1 n_contrib | 2 final_T | 3 info u32[4] (R, longest tile list)
size_t image_offsets(int W, int H, size_t *off) {
    const size_t tiles = (size_t)cdiv(W, CSPLAT_TILE) * cdiv(H, CSPLAT_TILE), X = (size_t)W * H;
    off[0] = 0;
    off[1] = align256(tiles * 8);
    off[2] = off[1] + align256(X * 4);
    off[3] = off[2] + align256(X * 4);
    off[4] = off[3] + align256(256 + 2 * (tiles + 4) * 4);   // info: [0] R, [1] longest list; word 64: number of non-empty tiles, their ids in
                                                             // tile order; word 64 + tiles + 4: the same ids, longest list first (K6's launch order)
    return off[4];
}
// per-(counting workgroup, tile) table of the bucketed binning path; requested as its own TEMP-class chunk
size_t bucket_table_bytes(int P, int tiles) { return align256(((size_t)cdiv(P > 0 ? P : 1, BUCKET_G) + 1) * tiles * 4); }
// binning: 0 keys_sorted u64[R] | 1 ids_sorted u32[R] | 2 seg_offset i32[tiles+1] + blk_hi u32[tiles][16] (largest n_contrib of every
//          4x4 pixel block, written by K6: K7 drops the (segment, quadrant) workgroups behind it on ONE scalar load) | 3 slot_tile i32[slots]
//          | 4 ckpt float4[slots][16 blocks][16 pixels]   (slots = R/SEG + tiles + 1 bounds sum_t ceil(n_t/SEG))
//          | 5 mask16 u16[R+1] | 6 recA float4[R+1] | 7 recB float4[R+1] | 8 recC float2[R+1]   (entry R = the null record)
//          | 9 bbits u64[slots][16 blocks][SEG / 64]: per segment and block, which of the segment's 256 entries the block BLENDED (K6 -> K7)
//          | 10 bmask u64[R / 64 + 4][16 blocks]: mask16 TRANSPOSED -- per 64 list entries (global index >> 6) and block, which entries
//            reach the block (K5b -> K6, one scalar 8-byte load per chunk instead of 64 mask loads and a ballot)
constexpr int B_NFIELDS = 11;
size_t binning_offsets(int64_t R, int tiles, size_t *off) {
    const size_t n = (size_t)(R > 0 ? R : 1);
    const size_t slots = (size_t)max_slots(R, tiles);
    off[0] = 0;
    off[1] = align256(n * 8);
    off[2] = off[1] + align256(n * 4);
    off[3] = off[2] + align256((size_t)(tiles + 1) * 4 + (size_t)tiles * 16 * 4);   // seg_offset[tiles + 1], then blk_hi[tiles][16]
    off[4] = off[3] + align256(slots * 4);
    off[5] = off[4] + align256(slots * 256 * 16);
    off[6] = off[5] + align256((n + 1) * 2);
    off[7] = off[6] + align256((n + 1) * 16);
    off[8] = off[7] + align256((n + 1) * 16);
    off[9] = off[8] + align256((n + 1) * 8);
    off[10] = off[9] + align256(slots * 16 * (SEG / 8));
    return off[10] + align256(((n + 63) / 64 + 4) * 16 * 8);
}
// temp: 0 keys_unsorted | 1 ids_unsorted | 2 keys_tmp | 3 ids_tmp | 4 sort table
size_t temp_offsets(int64_t R, size_t *off) {
    const size_t n = (size_t)(R > 0 ? R : 1);
    off[0] = 0;
    off[1] = off[0] + align256(n * 8);
    off[2] = off[1] + align256(n * 4);
    off[3] = off[2] + align256(n * 8);
    off[4] = off[3] + align256(n * 4);
    return off[4] + csplat_sort_temp_bytes(R);
}

// camera constants stay in HBM (80 bytes, read through the scalar cache by every wave): no host round trip
int make_cam(Cam &c, const float *view, const float *proj, const float *campos, float tanfovx, float tanfovy, int W, int H) {
    c.view = view; c.proj = proj; c.campos = campos;
    c.tanfovx = tanfovx; c.tanfovy = tanfovy;
    c.fx = (float)W / (2.0f * tanfovx); c.fy = (float)H / (2.0f * tanfovy);
    c.W = W; c.H = H; c.gx = cdiv(W, CSPLAT_TILE); c.gy = cdiv(H, CSPLAT_TILE);
    return 0;
}

// Mailboxes for the one host read of the forward (R and the longest tile list): 64 slots of host-pinned, device-mapped
// memory.  The scan kernel stores the two words, fences at system scope and stores a per-call tag; the host spins on the
// tag.  A blocking hipStreamSynchronize costs tens of microseconds of wake-up latency during which the GPU idles.
struct Mailboxes {
    volatile uint32_t *host = nullptr;
    uint32_t *dev = nullptr;
    std::atomic<uint32_t> next{1};
    bool tried = false;
};
Mailboxes g_mail;
std::mutex g_mail_mu;
constexpr int MAIL_SLOTS = 64, MAIL_WORDS = 16;

bool mail_init() {
    std::lock_guard<std::mutex> lk(g_mail_mu);
    if (!g_mail.tried) {
        g_mail.tried = true;
        void *h = nullptr, *d = nullptr;
        if (hipHostMalloc(&h, MAIL_SLOTS * MAIL_WORDS * 4, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess &&
            hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
            memset(h, 0, MAIL_SLOTS * MAIL_WORDS * 4);
            g_mail.host = (volatile uint32_t *)h;
            g_mail.dev = (uint32_t *)d;
        }
        (void)hipGetLastError();
    }
    return g_mail.host != nullptr;
}

// csplat_debug_flags: bit 0 no culling; bit 1 force the global radix sort; bit 2 no mailbox; bit 4 culling radius x4;
// bit 5 circle test only; bit 7 per-view K8 launches; bit 8 bit-reproducible backward (ordered sums instead of float atomics);
// bit 9 per-view launches on per-view streams; bit 10 no speculative second phase; bit 11 tile sort = the LSD radix sort only;
// bit 12 tile sort: a tile with any multi-key bucket takes the radix fallback (test hook);
// bit 15 K6 in the ROW form (four survivors per step; default: the survivor-column form, 16 per step, DPP row scans).
// (bits 13, 14, 16-21 selected the shelved kernel forms of round 3; they left the library in round 4 and are ignored)
unsigned g_debug_flags = 0;
unsigned long long *g_stamp_buf = nullptr;     // csplat_debug_stamps: 12 u64 per K7 workgroup (rows form, batched launch)
size_t g_stamp_words = 0;

// `mode` argument of the tile sort kernels: bit 0 ids < 2^24, bit 1 radix only, bit 2 fallback limit 1
int tsort_mode(int P) { return (P < (1 << 24) ? 1 : 0) | ((g_debug_flags & 2048u) ? 2 : 0) | ((g_debug_flags & 4096u) ? 4 : 0); }

int higher_msb(uint32_t n) {  // number of bits needed to represent tile ids < n (upstream getHigherMsb)
    int b = 0;
    while ((1u << b) < n && b < 31) b++;
    return b == 0 ? 1 : b;
}

// events for cross-stream ordering (no timing): a ring; a wait captures the record that precedes it, so reuse is safe
hipEvent_t pooled_event() {
    constexpr int N = 256;
    static hipEvent_t ring[N];
    static std::atomic<unsigned> next{0};
    static std::mutex mu;
    const unsigned k = next.fetch_add(1) % N;
    std::lock_guard<std::mutex> lk(mu);
    if (!ring[k] && hipEventCreateWithFlags(&ring[k], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return ring[k];
}

}  // namespace

extern "C" {

int csplat_abi_version(void) { return CSPLAT_ABI_VERSION; }
int csplat_debug_flags(unsigned flags) { g_debug_flags = flags; return 0; }
unsigned csplat_debug_flags_query(void) { return g_debug_flags; }
// measurement hook: a device buffer that the batched row-form K7 fills with s_memtime stamps (12 u64 per workgroup, launch order
// [view][workgroup]); NULL switches it off.  Not part of the operator interface.
int csplat_debug_stamps(void *buf, size_t bytes) { g_stamp_buf = (unsigned long long *)buf; g_stamp_words = bytes / 8; return 0; }
}  // extern "C"
// (the stamp buffer for the library's other translation units: nullptr unless one of at least `need_words` words was handed over)
unsigned long long *csplat_stamp_buffer(size_t need_words) { return (g_stamp_buf && g_stamp_words >= need_words) ? g_stamp_buf : nullptr; }
extern "C" {
const char *csplat_last_error(void) { return g_csplat_err; }

size_t csplat_geom_bytes(int P) { size_t off[G_NFIELDS]; return geom_offsets(P, off); }
size_t csplat_image_bytes(int W, int H) { size_t off[5]; return image_offsets(W, H, off); }
size_t csplat_binning_bytes(int64_t R, int W, int H) { size_t off[B_NFIELDS]; return binning_offsets(R, cdiv(W, CSPLAT_TILE) * cdiv(H, CSPLAT_TILE), off); }
size_t csplat_temp_bytes(int P, int64_t R, int W, int H) { (void)P; (void)W; (void)H; size_t off[5]; return temp_offsets(R, off); }
// backward scratch: the per-Gaussian records; in the bit-reproducible mode (csplat_debug_flags bit 8) also one 9-float record
// per (list entry, quadrant)
static size_t det_bytes(int64_t R) { return align256((size_t)(R > 0 ? R : 1) * 16 * 9 * 4); }
size_t csplat_backward_scratch_bytes(int P, int64_t R) {
    return align256((size_t)(P > 0 ? P : 1) * ACC_STRIDE * 4) + ((g_debug_flags & 256u) ? det_bytes(R) : 0);
}
int csplat_geom_layout(int P, size_t *o8) { size_t off[G_NFIELDS]; geom_offsets(P, off); for (int k = 0; k < 8; k++) o8[k] = off[k]; return 0; }
// every sub-buffer of the BINNING chunk (csplat.h: csplat_binning_fields): 0 keys 1 ids 2 seg_offset + blk_hi 3 slot_tile 4 checkpoints
// 5 masks 6-8 records A / B / C 9 bbits 10 bmask; o11[11] = byte offsets for a chunk laid out for R list entries (diagnostics: tools/,
// bench.py's count of K7's atomic requests)
int csplat_binning_fields(int64_t R, int W, int H, size_t *o11) { binning_offsets(R, cdiv(W, CSPLAT_TILE) * cdiv(H, CSPLAT_TILE), o11); return 0; }
int csplat_binning_layout(int64_t R, int W, int H, size_t *o2) { size_t off[B_NFIELDS]; binning_offsets(R, cdiv(W, CSPLAT_TILE) * cdiv(H, CSPLAT_TILE), off); o2[0] = off[0]; o2[1] = off[1]; return 0; }
int csplat_image_layout(int W, int H, size_t *o3) { size_t off[5]; image_offsets(W, H, off); o3[0] = off[0]; o3[1] = off[1]; o3[2] = off[2]; return 0; }

// ---- two-phase forward.  begin: K1 + the counting half of the binning, everything that does not need num_rendered;
// finish: reads num_rendered (mailbox poll), allocates the R-sized chunks, K3..K6.  A caller with several independent
// views issues every begin (each on its own stream) before the first finish, so the one host round trip per view and
// the under-filled compositing kernels of different views overlap.  csplat_forward = begin + finish.
struct FwdTicket {
    bool used = false;
    hipStream_t s = nullptr;
    int P = 0, W = 0, H = 0, tiles = 0, nb = 0;
    bool can_bucket = false, use_mail = false;
    uint32_t tag = 0;
    volatile uint32_t *mb_host = nullptr;
    Cam cam;
    Geom g;
    void *gbase = nullptr, *ibase = nullptr;
    int2 *ranges = nullptr;
    uint32_t *n_contrib = nullptr, *info = nullptr, *table = nullptr;
    float *final_T = nullptr;
    const float *bg = nullptr;
    int32_t *radii = nullptr;
    csplat_alloc_fn alloc = nullptr;
    void *alloc_ctx = nullptr;
    uint32_t *mb_dev = nullptr;
    int D = 0, M = 0;
    float scale_modifier = 1.f;
    const float *means3D = nullptr, *shs = nullptr, *colors_precomp = nullptr, *opacities = nullptr, *scales = nullptr, *rotations = nullptr,
                *cov3D_precomp = nullptr;
};
constexpr int MAX_TICKETS = 64;
static FwdTicket g_tickets[MAX_TICKETS];
static std::mutex g_ticket_mu;

// allocation + bookkeeping of a forward: nothing is launched
static int begin_prepare(void *stream, int P, int D, int M, const float *bg, int W, int H, const float *means3D,
                         const float *shs, const float *colors_precomp, const float *opacities, const float *scales,
                         float scale_modifier, const float *rotations, const float *cov3D_precomp, const float *view,
                         const float *proj, const float *campos, float tanfovx, float tanfovy, int prefiltered,
                         csplat_alloc_fn alloc, void *alloc_ctx, int32_t *radii, int *ticket_out) {
    hipStream_t s = (hipStream_t)stream;
    (void)prefiltered;
    CSPLAT_REQUIRE(P >= 0 && W > 0 && H > 0, "csplat_forward: bad sizes");
    CSPLAT_REQUIRE(ticket_out != nullptr, "csplat_forward_begin: ticket_out missing");
    // (an empty input, P == 0, legitimately arrives with NULL data pointers)
    CSPLAT_REQUIRE(P == 0 || (shs != nullptr) != (colors_precomp != nullptr), "provide exactly one of shs / colors_precomp");
    CSPLAT_REQUIRE(P == 0 || (cov3D_precomp != nullptr) != (scales != nullptr && rotations != nullptr),
                   "provide exactly one of (scales, rotations) / cov3D_precomp");
    CSPLAT_REQUIRE(shs == nullptr || (D >= 0 && D <= 3 && M >= (D + 1) * (D + 1)), "SH degree / coefficient count mismatch");
    CSPLAT_REQUIRE(alloc != nullptr, "allocator callback missing");
    Cam cam;
    make_cam(cam, view, proj, campos, tanfovx, tanfovy, W, H);

    void *gbase = alloc(alloc_ctx, CSPLAT_CHUNK_GEOM, csplat_geom_bytes(P));
    void *ibase = alloc(alloc_ctx, CSPLAT_CHUNK_IMAGE, csplat_image_bytes(W, H));
    CSPLAT_REQUIRE(gbase && ibase, "allocator returned NULL");
    size_t ioff[5];
    image_offsets(W, H, ioff);
    const int tiles = cam.gx * cam.gy;
    const bool can_bucket = tiles <= BUCKET_TILES && !(g_debug_flags & 2u);
    uint32_t *table = nullptr;
    if (can_bucket) {
        table = (uint32_t *)alloc(alloc_ctx, CSPLAT_CHUNK_TABLE, bucket_table_bytes(P, tiles));
        CSPLAT_REQUIRE(table, "allocator returned NULL");
    }
    int tk = -1;
    {
        std::lock_guard<std::mutex> lk(g_ticket_mu);
        for (int i = 0; i < MAX_TICKETS && tk < 0; i++)
            if (!g_tickets[i].used) tk = i;
        if (tk >= 0) g_tickets[tk].used = true;
    }
    CSPLAT_REQUIRE(tk >= 0, "csplat_forward_begin: more than 64 forwards begun and not finished");
    FwdTicket &t = g_tickets[tk];
    t.s = s; t.P = P; t.W = W; t.H = H; t.tiles = tiles; t.nb = cdiv(P > 0 ? P : 1, BUCKET_G); t.can_bucket = can_bucket;
    t.use_mail = false; t.tag = 0; t.mb_host = nullptr; t.mb_dev = nullptr;
    if (can_bucket && !(g_debug_flags & 4u) && mail_init()) {
        t.use_mail = true;
        t.tag = g_mail.next.fetch_add(1);
        if (t.tag == 0) t.tag = g_mail.next.fetch_add(1);
        const int slot = (int)(t.tag % MAIL_SLOTS);
        t.mb_host = g_mail.host + slot * MAIL_WORDS;
        t.mb_dev = g_mail.dev + slot * MAIL_WORDS;
    }
    t.cam = cam; t.g = geom_view(gbase, P); t.gbase = gbase; t.ibase = ibase;
    t.ranges = (int2 *)((char *)ibase + ioff[0]);
    t.n_contrib = (uint32_t *)((char *)ibase + ioff[1]);
    t.final_T = (float *)((char *)ibase + ioff[2]);
    t.info = (uint32_t *)((char *)ibase + ioff[3]);
    t.table = table; t.bg = bg; t.radii = radii; t.alloc = alloc; t.alloc_ctx = alloc_ctx;
    t.D = D; t.M = M; t.means3D = means3D; t.shs = shs; t.colors_precomp = colors_precomp; t.opacities = opacities; t.scales = scales;
    t.scale_modifier = scale_modifier; t.rotations = rotations; t.cov3D_precomp = cov3D_precomp;
    *ticket_out = tk;
    return 0;
}

static int nocull_mode() { return (int)((g_debug_flags & 1u) ? 1 : ((g_debug_flags & 16u) ? 2 : 0)); }

// K1 + the counting half of the binning of ONE view, on its stream
static int begin_launch(const FwdTicket &t) {
    hipStream_t s = t.s;
    const int P = t.P, tiles = t.tiles, nb = t.nb;
    if (!t.can_bucket) HIP_TRY(hipMemsetAsync(t.ranges, 0, (size_t)tiles * 8, s));
    if (P > 0) {
        ProfScope ps(PROF_K1, s);
        const bool stage = t.shs != nullptr && t.M == 16 && ((uintptr_t)t.shs & 15u) == 0;
        if (stage)
            k_preprocess<true><<<cdiv(P, 256), 256, 0, s>>>(P, t.D, t.M, t.means3D, t.shs, t.colors_precomp, t.opacities, t.scales,
                                                             t.scale_modifier, t.rotations, t.cov3D_precomp, t.cam, t.g, t.radii, nocull_mode());
        else
            k_preprocess<false><<<cdiv(P, 256), 256, 0, s>>>(P, t.D, t.M, t.means3D, t.shs, t.colors_precomp, t.opacities, t.scales,
                                                              t.scale_modifier, t.rotations, t.cov3D_precomp, t.cam, t.g, t.radii, nocull_mode());
        LAUNCH_CHECK();
    }
    if (t.can_bucket) {
        ProfScope ps(PROF_K2, s);
        if (P > 0) {
            k_tile_count<<<nb, BUCKET_G, (size_t)tiles * 4, s>>>(P, tiles, t.g.xy, t.radii, t.cam, t.table);
            LAUNCH_CHECK();
        } else {
            HIP_TRY(hipMemsetAsync(t.table, 0, (size_t)nb * tiles * 4, s));
        }
        uint32_t *tile_cnt = t.table + (size_t)nb * tiles;   // last row of the chunk: per-tile totals
        k_tile_colscan<<<cdiv(tiles, 256), 256, 0, s>>>(tiles, nb, t.table, tile_cnt);
        LAUNCH_CHECK();
        k_tile_scan<<<1, 1024, 0, s>>>(tiles, tile_cnt, t.ranges, t.info, t.mb_dev, t.tag);
        LAUNCH_CHECK();
    }
    return 0;
}

// the same for V views that share the Gaussians (P, SH, opacities, scales; own means / rotations / cameras): four launches on
// `join` in all, then every view's stream waits for them.  false = the views do not qualify (the caller launches per view).
static bool begin_views_compatible(int V, const int *tk) {
    // (V == 1 qualifies too since round 5: a camera-by-camera caller -- the reference's own loop, train_utils.py:259-272 -- then gets the
    //  speculative second phase as well instead of a blocking read of its counts per camera)
    if (V < 1 || V > K1_MAX_VIEWS || (g_debug_flags & 512u)) return false;
    const FwdTicket &a = g_tickets[tk[0]];
    if (a.P <= 0 || !a.can_bucket || !a.shs || a.colors_precomp || a.cov3D_precomp || !a.scales || !a.rotations) return false;
    for (int i = 1; i < V; i++) {
        const FwdTicket &w = g_tickets[tk[i]];
        if (w.P != a.P || w.D != a.D || w.M != a.M || w.W != a.W || w.H != a.H || w.shs != a.shs || w.opacities != a.opacities ||
            w.scales != a.scales || w.scale_modifier != a.scale_modifier || w.colors_precomp || w.cov3D_precomp || !w.rotations ||
            !w.can_bucket)
            return false;
    }
    return true;
}
static int begin_launch_views(int V, const int *tk, hipStream_t join) {
    const FwdTicket &a = g_tickets[tk[0]];
    const int P = a.P, tiles = a.tiles, nb = a.nb;
    K1Table tab;
    tab.n = V;
    for (int i = 0; i < V; i++) {
        const FwdTicket &t = g_tickets[tk[i]];
        K1View &k = tab.v[i];
        k.means3D = t.means3D; k.rotations = t.rotations; k.cam = t.cam; k.g = t.g; k.radii = t.radii; k.table = t.table;
        k.info = t.info; k.mailbox = t.mb_dev; k.ranges = t.ranges; k.tag = t.tag;
    }
    {
        ProfScope ps(PROF_K1, join);
        const bool stage = a.M == 16 && ((uintptr_t)a.shs & 15u) == 0;
        if (stage)
            k_preprocess_views<true><<<cdiv(P, K1V_G), 256, 0, join>>>(P, a.D, a.M, a.shs, a.opacities, a.scales, a.scale_modifier, tab,
                                                                     nocull_mode());
        else
            k_preprocess_views<false><<<cdiv(P, K1V_G), 256, 0, join>>>(P, a.D, a.M, a.shs, a.opacities, a.scales, a.scale_modifier, tab,
                                                                      nocull_mode());
        LAUNCH_CHECK();
    }
    {
        ProfScope ps(PROF_K2, join);
        k_tile_count_views<<<dim3(nb, V), BUCKET_G, (size_t)tiles * 4, join>>>(P, tiles, tab);
        LAUNCH_CHECK();
        k_tile_colscan_views<<<dim3(cdiv(tiles, 256), V), 256, 0, join>>>(tiles, nb, tab);
        LAUNCH_CHECK();
        k_tile_scan_views<<<V, 1024, 0, join>>>(tiles, nb, tab);
        LAUNCH_CHECK();
    }
    return 0;
}

int csplat_forward_begin(void *stream, int P, int D, int M, const float *bg, int W, int H, const float *means3D,
                         const float *shs, const float *colors_precomp, const float *opacities, const float *scales,
                         float scale_modifier, const float *rotations, const float *cov3D_precomp, const float *view,
                         const float *proj, const float *campos, float tanfovx, float tanfovy, int prefiltered,
                         csplat_alloc_fn alloc, void *alloc_ctx, int32_t *radii, int *ticket_out) {
    if (int rc = begin_prepare(stream, P, D, M, bg, W, H, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations,
                               cov3D_precomp, view, proj, campos, tanfovx, tanfovy, prefiltered, alloc, alloc_ctx, radii, ticket_out))
        return rc;
    const int rc = begin_launch(g_tickets[*ticket_out]);
    if (rc) {
        std::lock_guard<std::mutex> lk(g_ticket_mu);
        g_tickets[*ticket_out].used = false;
    }
    return rc;
}

// the one host read of a forward: R and the longest tile list of a view whose first phase has been launched
static int finish_read(const FwdTicket &t, uint32_t host_info[3], hipStream_t launched_on = nullptr) {
    hipStream_t s = launched_on ? launched_on : t.s;     // the stream the first phase was launched on
    host_info[0] = host_info[1] = host_info[2] = 0;
    if (t.can_bucket) {
        bool got = false;
        if (t.use_mail) {   // spin on the tag (bounded: fall back to a stream synchronise after 2 s)
            const auto t0 = std::chrono::steady_clock::now();
            unsigned spins = 0;
            while (!(got = (t.mb_host[2] == t.tag))) {
                if ((++spins & 0x3FFu) == 0 &&
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) break;
            }
            if (got) { host_info[0] = t.mb_host[0]; host_info[1] = t.mb_host[1]; host_info[2] = t.mb_host[3]; }
        }
        if (!got) {
            HIP_TRY(hipMemcpyAsync(host_info, t.info, 12, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
        }
    } else if (t.P > 0) {
        ProfScope ps(PROF_K2, s);
        if (int rc = csplat_inclusive_scan_u32(s, t.g.tiles_touched, t.g.offsets, t.P, t.g.scan_tmp)) return rc;
        HIP_TRY(hipMemcpyAsync(host_info, t.g.offsets + (t.P - 1), 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        host_info[1] = 0xFFFFFFFFu;
        host_info[2] = 0xFFFFFFFFu;
    }
    // list positions, tile ranges and the int `num_rendered` of the ABI are 32-bit (as upstream's): refuse instead of wrapping
    CSPLAT_REQUIRE(host_info[0] <= 0x7FFFFFFFu, "csplat_forward: more than 2^31 - 1 tile instances (Gaussian x tile pairs) in one view");
    return 0;
}
// longest tile list the in-LDS sort takes (64 KB of keys + 17 KB of counters when the device grants 96 KB per workgroup)
static uint32_t tile_sort_cap() {
    static int s_lds_big = -1;
    if (s_lds_big < 0) {
        s_lds_big = hipFuncSetAttribute((const void *)k_tile_sort, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess &&
                    hipFuncSetAttribute((const void *)k_tile_sort_views, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) == hipSuccess;
        (void)hipGetLastError();   // a refusal must not poison the launch checks below
    }
    return s_lds_big ? (uint32_t)BUCKET_CAP : 5120u;
}

// Second phase of ALL views in one launch per stage on `join` (see P2Table).  *done = false (and nothing launched or
// allocated) when the views do not qualify -- a tile list too long for the LDS sort, different sizes -- and the caller
// finishes view by view.
// What the last batched call saw, per image size: the capacities the next one is launched with BEFORE its counts are read.
// (a short ring: a process that alternates between scenes of different density -- bench.py --mode scenes -- is served by the largest
// of its recent calls with the same shape instead of failing the speculation at every switch)
struct SpecHist { int W = 0, H = 0, P = 0; uint32_t R = 0, longest = 0, busy = 0; };
constexpr int SPEC_RING = 8;
static SpecHist g_spec_ring[SPEC_RING];
static int g_spec_next = 0;
static std::mutex g_spec_mu;

// A call whose speculative second phase has been launched but whose counts have not been read yet (csplat_forward_views_deferred):
// what csplat_forward_views_settle needs to finish it.  Keyed by the caller's view array.
struct PendingViews {
    bool used = false;
    const csplat_view *key = nullptr;
    int V = 0, tk[P2_MAX_VIEWS];
    uint32_t Rcap[P2_MAX_VIEWS], Lcap = 0, Bcap = 0;
};
constexpr int MAX_PENDING = 8;
static PendingViews g_pending[MAX_PENDING];
static std::mutex g_pending_mu;

// lays the chunks of every view out for `Rcap[i]` list entries and launches the five stages of the second phase on `join`
// (valid != nullptr: K6's first wave leaves 1 there when every view's counts fitted the capacities, else 0 -- csplat_forward_views_faith)
static int p2_launch(int V, const int *tk, csplat_view *v, hipStream_t join, const uint32_t *Rcap, uint32_t Lcap, int spec, uint32_t Bcap,
                     uint32_t *valid = nullptr) {
    const FwdTicket &a = g_tickets[tk[0]];
    const int P = a.P, W = a.W, H = a.H, tiles = a.tiles, nb = a.nb;
    {
        P2Table tab;
        tab.valid = valid; tab.nviews = V;
        uint32_t maxR = 0;
        for (int i = 0; i < V; i++) {
            const FwdTicket &t = g_tickets[tk[i]];
            const uint32_t R = Rcap[i];
            maxR = R > maxR ? R : maxR;
            void *bbase = t.alloc(t.alloc_ctx, CSPLAT_CHUNK_BINNING, csplat_binning_bytes(R, W, H));
            void *tbase = t.alloc(t.alloc_ctx, CSPLAT_CHUNK_TEMP, csplat_temp_bytes(P, R, W, H));
            CSPLAT_REQUIRE(bbase && tbase, "allocator returned NULL");
            size_t boff[B_NFIELDS], toff[5];
            binning_offsets(R, tiles, boff);
            temp_offsets(R, toff);
            P2View &k = tab.v[i];
            k.g = t.g; k.cam = t.cam; k.radii = t.radii; k.table = t.table; k.ranges = t.ranges;
            k.keys_u = (uint64_t *)((char *)tbase + toff[0]);
            k.keys_sorted = (uint64_t *)((char *)bbase + boff[0]);
            k.ids_sorted = (uint32_t *)((char *)bbase + boff[1]);
            k.seg_offset = (int *)((char *)bbase + boff[2]);
            k.slot_tile = (int *)((char *)bbase + boff[3]);
            k.ckpt = (float4 *)((char *)bbase + boff[4]);
            k.mask16 = (uint16_t *)((char *)bbase + boff[5]);
            k.bbits = (unsigned long long *)((char *)bbase + boff[9]);
            k.bmask = (unsigned long long *)((char *)bbase + boff[10]);
            k.recA = (float4 *)((char *)bbase + boff[6]); k.recB = (float4 *)((char *)bbase + boff[7]);
            k.recC = (float2 *)((char *)bbase + boff[8]);
            k.bg = t.bg; k.final_T = t.final_T; k.n_contrib = t.n_contrib; k.out_color = v[i].out_color; k.out_depth = v[i].out_depth;
            k.R = R; k.info = t.info; k.Lcap = Lcap; k.Bcap = Bcap; k.spec = spec;
            v[i].layout_rendered = (int)R; v[i].geom = t.gbase; v[i].binning = bbase; v[i].image = t.ibase;
        }
        {
            ProfScope ps(PROF_K3, join);
            k_emit_bucket_views<<<dim3(nb + 1, V), BUCKET_G, (size_t)tiles * 4, join>>>(P, tiles, tab);
            LAUNCH_CHECK();
        }
        {
            const size_t lds = tsort_lds_bytes((int)Lcap);
            ProfScope ps(PROF_K4, join);
            k_tile_sort_views<<<dim3(tiles < TSORT_GRID ? tiles : TSORT_GRID, V), TSORT_THREADS, lds, join>>>(tab, tsort_mode(P));
            LAUNCH_CHECK();
        }
        {
            ProfScope ps(PROF_K5, join);
            const int exact = (g_debug_flags & (1u | 16u | 32u)) ? 0 : 1, nbm = cdiv((int64_t)maxR + 1, 256);
            if (V == 1 || V == 2 || V == 4 || V == 8)         // one view per XCD (see k_block_masks_views)
                k_block_masks_views<<<dim3((unsigned)(((int64_t)nbm * V + 7) / 8 * 8)), 256, 0, join>>>(tab, exact, V);
            else
                k_block_masks_views<<<dim3(nbm, V), 256, 0, join>>>(tab, exact, 0);
            LAUNCH_CHECK();
        }
        {
            ProfScope ps(PROF_K6, join);
            // Waves for the NON-EMPTY tiles only (Bcap of them, longest list first) + K6_EXTRA waves that paint the empty tiles' pixels: on
            // scene_1 nine tiles in ten are empty, and launching 16 waves for each of them cost ~50 us of the launch (the same launch on a
            // scene of 500 Gaussians: 51 us).  What is left is ~17 k waves with work for 8192 wave slots, 40-75 us each in the row form
            // (four survivors a step): the launch is as long as a few of those in a row on the unluckiest slot.  Hence the SURVIVOR-COLUMN
            // form by default (sixteen survivors a step: a quarter of the steps, each longer; 96 VGPRs, which no longer matters with two
            // waves per slot to place): 158 -> 137 us for four views, step 0.685 -> 0.665 ms (same box, three alternations).  Bit 15 of
            // csplat_debug_flags selects the row form.
            const int total = cdiv(tiles, 8) * 128, bg_ = cdiv((int)Bcap, 8) * 128, busy_grid = bg_ < total ? bg_ : total;
            if (!(g_debug_flags & 32768u))
                k_composite_fwd_views<false><<<dim3(busy_grid + K6_EXTRA, V), 64, 0, join>>>(tiles, W, H, tab, busy_grid);
            else
                k_composite_fwd_views<true><<<dim3(busy_grid + K6_EXTRA, V), 64, 0, join>>>(tiles, W, H, tab, busy_grid);
            LAUNCH_CHECK();
        }
        return 0;
    }
}

// mode 0: the whole second phase (launch, read the counts, repeat with exact sizes if the speculation missed);
// mode 1: as 0, but when the speculative launch is possible return right after it with *pend filled (pend->used) -- nothing is read;
// mode 2: finish a mode-1 call: read the counts, accept or repeat (*relaunched)
static int finish_views_batched(int V, const int *tk, csplat_view *v, hipStream_t join, bool *done, int mode = 0,
                                PendingViews *pend = nullptr, int *relaunched = nullptr) {
    *done = false;
    if (V < 1 || V > P2_MAX_VIEWS || (g_debug_flags & 512u)) return 0;
    const FwdTicket &a = g_tickets[tk[0]];
    for (int i = 0; i < V; i++) {
        const FwdTicket &t = g_tickets[tk[i]];
        if (!t.can_bucket || t.P != a.P || t.W != a.W || t.H != a.H || t.P <= 0) return 0;
    }
    const int P = a.P, W = a.W, H = a.H, tiles = a.tiles;
    const uint32_t cap = tile_sort_cap();
    uint32_t info[P2_MAX_VIEWS][3];
    bool have_info = false;
    auto read_counts = [&]() -> int {   // the one host read of the call: instances and longest tile list of every view
        for (int i = 0; i < V && !have_info; i++)
            if (int rc = finish_read(g_tickets[tk[i]], info[i], join)) return rc;
        have_info = true;
        return 0;
    };
    auto launch = [&](const uint32_t *Rcap, uint32_t Lcap, int spec, uint32_t Bcap) -> int { return p2_launch(V, tk, v, join, Rcap, Lcap, spec, Bcap); };
    auto remember = [&]() {
        std::lock_guard<std::mutex> lk(g_spec_mu);
        SpecHist &e = g_spec_ring[g_spec_next];
        g_spec_next = (g_spec_next + 1) % SPEC_RING;
        e.W = W; e.H = H; e.P = P; e.R = 0; e.longest = 0; e.busy = 0;
        for (int i = 0; i < V; i++) {
            e.R = info[i][0] > e.R ? info[i][0] : e.R;
            e.longest = info[i][1] > e.longest ? info[i][1] : e.longest;
            e.busy = info[i][2] > e.busy ? info[i][2] : e.busy;
        }
    };
    auto release = [&]() {
        std::lock_guard<std::mutex> lk(g_ticket_mu);
        for (int i = 0; i < V; i++) g_tickets[tk[i]].used = false;
    };
    // ---- speculative launch: in a training loop the counts of consecutive steps differ by a few per cent, and waiting for them
    // leaves the GPU idle for a host round trip (~40 us of a 1 ms step).  With the previous call's counts + 1/8 as capacities the
    // second phase is queued straight behind the first; the counts are read AFTER that (the GPU is busy with the phase by then),
    // and a view whose counts do not fit was left untouched by the kernels (p2_live) -- the phase is then repeated with exact
    // sizes.  csplat_debug_flags bit 10 switches the speculation off.
    SpecHist hist;
    {
        std::lock_guard<std::mutex> lk(g_spec_mu);
        for (const SpecHist &e : g_spec_ring)
            if (e.R > 0 && e.W == W && e.H == H && e.P == P) {
                hist.W = W; hist.H = H; hist.P = P;
                hist.R = e.R > hist.R ? e.R : hist.R;
                hist.longest = e.longest > hist.longest ? e.longest : hist.longest;
                hist.busy = e.busy > hist.busy ? e.busy : hist.busy;
            }
    }
    if (mode == 2 || (!(g_debug_flags & 1024u) && hist.R > 0)) {
        uint32_t Rcap[P2_MAX_VIEWS];
        uint32_t Lcap, Bcap;
        if (mode == 2) {        // the capacities the pending call was launched with
            for (int i = 0; i < V; i++) Rcap[i] = pend->Rcap[i];
            Lcap = pend->Lcap;
            Bcap = pend->Bcap;
        } else {
            const uint64_t want = (uint64_t)hist.R + hist.R / 8 + 4096;
            const uint32_t rc32 = (uint32_t)(want > 0x7FFFFF00ull ? 0x7FFFFF00ull : want);
            for (int i = 0; i < V; i++) Rcap[i] = rc32;
            Lcap = hist.longest + hist.longest / 4 + 64;
            Lcap = Lcap > cap ? cap : Lcap;
            Bcap = hist.busy + hist.busy / 8 + 16;          // non-empty tiles: sizes the compositing forward's grid
            Bcap = Bcap > (uint32_t)tiles ? (uint32_t)tiles : Bcap;
            if (int rc = launch(Rcap, Lcap, 1, Bcap)) return rc;
            if (mode == 1) {    // deferred: the caller reads the counts later (csplat_forward_views_settle), the GPU has its work
                pend->used = true; pend->key = v; pend->V = V; pend->Lcap = Lcap; pend->Bcap = Bcap;
                for (int i = 0; i < V; i++) { pend->tk[i] = tk[i]; pend->Rcap[i] = Rcap[i]; v[i].num_rendered = -1; v[i].busy_tiles = 0; }
                *done = true;
                return 0;
            }
        }
        if (int rc = read_counts()) return rc;
        bool fits = true;
        for (int i = 0; i < V; i++) fits = fits && info[i][0] - 1u < Rcap[i] && info[i][1] <= Lcap && info[i][2] <= Bcap;
        remember();
        if (fits) {
            for (int i = 0; i < V; i++) { v[i].num_rendered = (int)info[i][0]; v[i].busy_tiles = (int)info[i][2]; }
            release();
            *done = true;
            return 0;
        }
    }
    if (int rc = read_counts()) return rc;
    uint32_t longest = 0, busiest = 0, Rex[P2_MAX_VIEWS];
    for (int i = 0; i < V; i++) {
        if (info[i][1] > cap || info[i][0] == 0) return 0;      // a list too long for the in-LDS sort, an empty view: view by view
        longest = info[i][1] > longest ? info[i][1] : longest;
        busiest = info[i][2] > busiest ? info[i][2] : busiest;
        Rex[i] = info[i][0];
    }
    remember();
    if (int rc = launch(Rex, longest, 0, busiest)) return rc;
    if (relaunched) *relaunched = 1;
    for (int i = 0; i < V; i++) { v[i].num_rendered = (int)info[i][0]; v[i].busy_tiles = (int)info[i][2]; }
    release();
    *done = true;
    return 0;
}

int csplat_forward_finish(int ticket, float *out_color, float *out_depth, int *num_rendered, void **geom_out,
                          void **binning_out, void **image_out) {
    CSPLAT_REQUIRE(ticket >= 0 && ticket < MAX_TICKETS && g_tickets[ticket].used, "csplat_forward_finish: unknown ticket");
    const FwdTicket t = g_tickets[ticket];
    {
        std::lock_guard<std::mutex> lk(g_ticket_mu);
        g_tickets[ticket].used = false;   // released whatever happens below
    }
    hipStream_t s = t.s;
    const int P = t.P, W = t.W, H = t.H, tiles = t.tiles, nb = t.nb;
    const bool can_bucket = t.can_bucket;
    const Cam cam = t.cam;
    const Geom g = t.g;
    void *gbase = t.gbase, *ibase = t.ibase;
    int2 *ranges = t.ranges;
    uint32_t *n_contrib = t.n_contrib, *table = t.table;
    float *final_T = t.final_T;
    const float *bg = t.bg;
    int32_t *radii = t.radii;
    csplat_alloc_fn alloc = t.alloc;
    void *alloc_ctx = t.alloc_ctx;
    uint32_t host_info[3] = {0, 0, 0};   // R, longest tile list, non-empty tiles
    if (int rc = finish_read(t, host_info)) return rc;
    const uint32_t R = host_info[0];
    const uint32_t cap = tile_sort_cap();
    const bool bucketed = can_bucket && host_info[1] <= cap;
    *num_rendered = (int)R;
    void *bbase = alloc(alloc_ctx, CSPLAT_CHUNK_BINNING, csplat_binning_bytes(R, W, H));
    CSPLAT_REQUIRE(bbase, "allocator returned NULL");
    size_t boff[B_NFIELDS];
    binning_offsets(R, tiles, boff);
    uint64_t *keys_sorted = (uint64_t *)((char *)bbase + boff[0]);
    uint32_t *ids_sorted = (uint32_t *)((char *)bbase + boff[1]);
    int *seg_offset = (int *)((char *)bbase + boff[2]);
    int *slot_tile = (int *)((char *)bbase + boff[3]);
    float4 *ckpt = (float4 *)((char *)bbase + boff[4]);
    uint16_t *mask16 = (uint16_t *)((char *)bbase + boff[5]);
    float4 *recA = (float4 *)((char *)bbase + boff[6]), *recB = (float4 *)((char *)bbase + boff[7]);
    float2 *recC = (float2 *)((char *)bbase + boff[8]);
    unsigned long long *bbits = (unsigned long long *)((char *)bbase + boff[9]);
    unsigned long long *bmask = (unsigned long long *)((char *)bbase + boff[10]);
    if (R > 0) {
        void *tbase = alloc(alloc_ctx, CSPLAT_CHUNK_TEMP, csplat_temp_bytes(P, R, W, H));
        CSPLAT_REQUIRE(tbase, "allocator returned NULL");
        size_t toff[5];
        temp_offsets(R, toff);
        uint64_t *keys_u = (uint64_t *)((char *)tbase + toff[0]);
        if (bucketed) {
            {
                ProfScope ps(PROF_K3, s);
                k_emit_bucket<<<nb, BUCKET_G, (size_t)tiles * 4, s>>>(P, tiles, g.xy, g.depth, radii, cam, table, ranges, keys_u);
                LAUNCH_CHECK();
            }
            const size_t lds = tsort_lds_bytes((int)host_info[1]);
            ProfScope ps(PROF_K4, s);
            k_tile_sort<<<tiles < TSORT_GRID ? tiles : TSORT_GRID, TSORT_THREADS, lds, s>>>(ranges, keys_u, keys_sorted, ids_sorted, t.info, tsort_mode(P));
            LAUNCH_CHECK();
        } else {
            // a tile list longer than the LDS sort takes: the global stable radix sort (upstream's pipeline shape)
            uint32_t *ids_u = (uint32_t *)((char *)tbase + toff[1]);
            uint64_t *keys_t = (uint64_t *)((char *)tbase + toff[2]);
            uint32_t *ids_t = (uint32_t *)((char *)tbase + toff[3]);
            void *stab = (char *)tbase + toff[4];
            if (can_bucket) {   // (the bucket path was attempted: the P-scan has not run yet)
                ProfScope ps(PROF_K2, s);
                if (int rc = csplat_inclusive_scan_u32(s, g.tiles_touched, g.offsets, P, g.scan_tmp)) return rc;
            }
            {
                ProfScope ps(PROF_K3, s);
                k_emit_keys<<<cdiv(P, 256), 256, 0, s>>>(P, g.xy, g.depth, g.offsets, radii, cam, keys_u, ids_u);
                LAUNCH_CHECK();
            }
            const int end_bit = 32 + higher_msb((uint32_t)tiles);
            {
                ProfScope ps(PROF_K4, s);
                if (int rc = csplat_sort_pairs(s, keys_u, ids_u, keys_sorted, ids_sorted, keys_t, ids_t, R, end_bit, stab)) return rc;
            }
            {
                ProfScope ps(PROF_K5, s);
                HIP_TRY(hipMemsetAsync(ranges, 0, (size_t)tiles * 8, s));
                k_tile_ranges<<<cdiv(R, 256), 256, 0, s>>>(R, keys_sorted, ranges);
                LAUNCH_CHECK();
            }
        }
    }
    {
        ProfScope ps(PROF_K5, s);
        k_seg_plan<<<1, 1024, 0, s>>>(tiles, ranges, seg_offset, slot_tile);
        LAUNCH_CHECK();
    }
    {
        ProfScope ps(PROF_K5, s);
        k_block_masks<<<cdiv((int64_t)R + 1, 256), 256, 0, s>>>((int64_t)R, cam.gx, keys_sorted, ids_sorted, g.pack, mask16, recA, recB, recC,
                                                                 (g_debug_flags & (1u | 16u | 32u)) ? 0 : 1, bmask);
        LAUNCH_CHECK();
    }
    {
        ProfScope ps(PROF_K6, s);
        if (g_debug_flags & 32768u)       // (bit 15: the row form, four survivors a step; default: the survivor-column form, 74 -> 67 us alone)
            k_composite_fwd<true><<<cdiv(tiles, 8) * 128, 64, 0, s>>>(tiles, W, H, cam.gx, ranges, mask16, recA, recB, recC, R, bg, seg_offset, ckpt,
                                                                      final_T, n_contrib, out_color, out_depth, bbits, bmask);
        else
            k_composite_fwd<false><<<cdiv(tiles, 8) * 128, 64, 0, s>>>(tiles, W, H, cam.gx, ranges, mask16, recA, recB, recC, R, bg, seg_offset, ckpt,
                                                                       final_T, n_contrib, out_color, out_depth, bbits, bmask);
        LAUNCH_CHECK();
    }
    *geom_out = gbase; *binning_out = bbase; *image_out = ibase;
    return 0;
}

int csplat_forward(void *stream, int P, int D, int M, const float *bg, int W, int H, const float *means3D,
                   const float *shs, const float *colors_precomp, const float *opacities, const float *scales,
                   float scale_modifier, const float *rotations, const float *cov3D_precomp, const float *view,
                   const float *proj, const float *campos, float tanfovx, float tanfovy, int prefiltered,
                   csplat_alloc_fn alloc, void *alloc_ctx, float *out_color, float *out_depth, int32_t *radii,
                   int *num_rendered, void **geom_out, void **binning_out, void **image_out) {
    int ticket = -1;
    if (int rc = csplat_forward_begin(stream, P, D, M, bg, W, H, means3D, shs, colors_precomp, opacities, scales, scale_modifier,
                                      rotations, cov3D_precomp, view, proj, campos, tanfovx, tanfovy, prefiltered, alloc,
                                      alloc_ctx, radii, &ticket))
        return rc;
    return csplat_forward_finish(ticket, out_color, out_depth, num_rendered, geom_out, binning_out, image_out);
}

// K7 on `stream`; K8 on `k8_stream` (after an event wait when it differs); accmask see k_preprocess_bwd
static int backward_impl(hipStream_t s, hipStream_t k8s, bool with_k7, bool with_k8, unsigned accmask, int P, int D, int M, int R, const float *bg, int W, int H,
                         const float *means3D, const float *shs, const float *scales, float scale_modifier,
                         const float *rotations, const float *cov3D_precomp, const float *view, const float *proj,
                         const float *campos, float tanfovx, float tanfovy, const int32_t *radii, const void *geom,
                         const void *binning, const void *image, const float *out_color, const float *dL_dpix, void *scratch,
                         float *dL_dmean2D, float *dL_dconic, float *dL_dopacity, float *dL_dcolor, float *dL_dmean3D,
                         float *dL_dcov3D, float *dL_dsh, float *dL_dscale, float *dL_drot) {
    CSPLAT_REQUIRE(geom && binning && image && out_color, "csplat_backward: missing saved state");
    CSPLAT_REQUIRE(dL_dmean2D && dL_dconic && dL_dopacity && dL_dcolor && dL_dmean3D && dL_dcov3D, "missing gradient outputs");
    CSPLAT_REQUIRE(scratch != nullptr, "csplat_backward: scratch (csplat_backward_scratch_bytes) missing");
    if (P <= 0) return 0;
    Cam cam;
    make_cam(cam, view, proj, campos, tanfovx, tanfovy, W, H);
    Geom g = geom_view((void *)geom, P);
    const int tiles = cam.gx * cam.gy;
    size_t ioff[5], boff[B_NFIELDS];
    image_offsets(W, H, ioff);
    binning_offsets(R, tiles, boff);
    const int2 *ranges = (const int2 *)((const char *)image + ioff[0]);
    const uint32_t *n_contrib = (const uint32_t *)((const char *)image + ioff[1]);
    const float *final_T = (const float *)((const char *)image + ioff[2]);
    const uint32_t *ids_sorted = (const uint32_t *)((const char *)binning + boff[1]);
    const int *seg_offset = (const int *)((const char *)binning + boff[2]);
    const int *slot_tile = (const int *)((const char *)binning + boff[3]);
    const float4 *ckpt = (const float4 *)((const char *)binning + boff[4]);
    const uint64_t *keys_sorted = (const uint64_t *)((const char *)binning + boff[0]);
    const unsigned long long *bbits = (const unsigned long long *)((const char *)binning + boff[9]);
    const float4 *recA = (const float4 *)((const char *)binning + boff[6]), *recB = (const float4 *)((const char *)binning + boff[7]);
    const float2 *recC = (const float2 *)((const char *)binning + boff[8]);
    float *acc = (float *)scratch;
    // (with_k7 = false: K7 of all views was launched as one batch by the caller)
    const bool det_mode = (g_debug_flags & 256u) != 0;
    float *det = det_mode ? (float *)((char *)scratch + align256((size_t)P * ACC_STRIDE * 4)) : nullptr;
    if (with_k7 && det_mode) HIP_TRY(hipMemsetAsync(det, 0, (size_t)(R > 0 ? R : 1) * 16 * 9 * 4, s));
    else if (with_k7 && !(accmask & CSPLAT_SCRATCH_ZEROED)) HIP_TRY(hipMemsetAsync(acc, 0, (size_t)P * ACC_STRIDE * 4, s));
    if (with_k7) {
        ProfScope ps(PROF_K7, s);
        if (R > 0) {
            const unsigned grid = (unsigned)cdiv(max_slots(R, tiles), 8) * 32u;
            if (det_mode)
                k_composite_bwd_rows<true><<<grid, 256, 0, s>>>(tiles, W, H, cam.gx, ranges, ids_sorted, bbits, recA, recB, recC, (uint32_t)R,
                                                                seg_offset, slot_tile, ckpt, final_T, n_contrib, out_color, dL_dpix, acc, det);
            else
                k_composite_bwd_rows<false><<<grid, 256, 0, s>>>(tiles, W, H, cam.gx, ranges, ids_sorted, bbits, recA, recB, recC, (uint32_t)R,
                                                                 seg_offset, slot_tile, ckpt, final_T, n_contrib, out_color, dL_dpix, acc, det);
            LAUNCH_CHECK();
        }
        if (det_mode) {   // fixed-order sum of every Gaussian's instance records (writes all of acc)
            k_det_reduce<<<cdiv(P, 256), 256, 0, s>>>(P, cam, g.xy, g.depth, radii, ranges, keys_sorted, ids_sorted, det, acc);
            LAUNCH_CHECK();
        }
    }
    if (!with_k8) return 0;   // (csplat_backward_views runs one K8 over all views afterwards)
    if (with_k7 && k8s != s) {
        hipEvent_t ev = pooled_event();
        CSPLAT_REQUIRE(ev != nullptr, "csplat_backward_views: no event");
        HIP_TRY(hipEventRecord(ev, s));
        HIP_TRY(hipStreamWaitEvent(k8s, ev, 0));
    }
    {
        ProfScope ps(PROF_K8, k8s);
        const bool stage = shs != nullptr && dL_dsh != nullptr && M == 16 && (((uintptr_t)shs | (uintptr_t)dL_dsh) & 15u) == 0;
        CSPLAT_REQUIRE(stage || !(accmask & CSPLAT_ACC_SH), "accumulating dL_dsh needs M == 16 and 16-byte aligned buffers");
        if (stage)
            k_preprocess_bwd<true, 128><<<cdiv(P, 128), 128, 0, k8s>>>(P, D, M, means3D, shs, scales, scale_modifier, rotations,
                                                                        cov3D_precomp != nullptr, cam, g, radii, acc, dL_dmean2D,
                                                                        dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D,
                                                                        dL_dsh, dL_dscale, dL_drot, accmask);
        else
            k_preprocess_bwd<false, 256><<<cdiv(P, 256), 256, 0, k8s>>>(P, D, M, means3D, shs, scales, scale_modifier, rotations,
                                                                         cov3D_precomp != nullptr, cam, g, radii, acc, dL_dmean2D,
                                                                         dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D,
                                                                         dL_dsh, dL_dscale, dL_drot, accmask);
        LAUNCH_CHECK();
    }
    return 0;
}

int csplat_backward(void *stream, int P, int D, int M, int R, const float *bg, int W, int H, const float *means3D,
                    const float *shs, const float *colors_precomp, const float *scales, float scale_modifier,
                    const float *rotations, const float *cov3D_precomp, const float *view, const float *proj,
                    const float *campos, float tanfovx, float tanfovy, const int32_t *radii, const void *geom,
                    const void *binning, const void *image, const float *out_color, const float *dL_dpix, void *scratch,
                    float *dL_dmean2D,
                    float *dL_dconic, float *dL_dopacity, float *dL_dcolor, float *dL_dmean3D, float *dL_dcov3D,
                    float *dL_dsh, float *dL_dscale, float *dL_drot) {
    (void)colors_precomp;
    return backward_impl((hipStream_t)stream, (hipStream_t)stream, true, true, 0u, P, D, M, R, bg, W, H, means3D, shs, scales, scale_modifier,
                         rotations, cov3D_precomp, view, proj, campos, tanfovx, tanfovy, radii, geom, binning, image, out_color,
                         dL_dpix, scratch, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale,
                         dL_drot);
}

// ---- batched entry points: V independent views, one stream each, fenced against `join_stream`
static int fence_in(int V, const csplat_view *v, hipStream_t join) {
    hipEvent_t ev = pooled_event();
    CSPLAT_REQUIRE(ev != nullptr, "csplat_*_views: no event");
    HIP_TRY(hipEventRecord(ev, join));
    for (int i = 0; i < V; i++) {
        bool seen = (hipStream_t)v[i].stream == join;
        for (int j = 0; j < i && !seen; j++) seen = v[j].stream == v[i].stream;
        if (!seen) HIP_TRY(hipStreamWaitEvent((hipStream_t)v[i].stream, ev, 0));
    }
    return 0;
}
static int fence_out(int V, const csplat_view *v, hipStream_t join) {
    for (int i = 0; i < V; i++) {
        bool seen = (hipStream_t)v[i].stream == join;
        for (int j = 0; j < i && !seen; j++) seen = v[j].stream == v[i].stream;
        if (seen) continue;
        hipEvent_t ev = pooled_event();
        CSPLAT_REQUIRE(ev != nullptr, "csplat_*_views: no event");
        HIP_TRY(hipEventRecord(ev, (hipStream_t)v[i].stream));
        HIP_TRY(hipStreamWaitEvent(join, ev, 0));
    }
    return 0;
}

// the per-view tail of a forward call: every prepared ticket is finished (= released) even after an error
static int finish_views_one_by_one(int V, csplat_view *v, const int *tickets, int begun, hipStream_t join, bool fenced, int rc) {
    for (int i = 0; i < begun; i++) {
        csplat_view &w = v[i];
        if (rc == 0) {
            rc = csplat_forward_finish(tickets[i], w.out_color, w.out_depth, &w.num_rendered, &w.geom, &w.binning, &w.image);
            w.layout_rendered = w.num_rendered;
        } else {
            std::lock_guard<std::mutex> lk(g_ticket_mu);
            g_tickets[tickets[i]].used = false;
        }
    }
    // (also on the error path: the side streams may hold work on torch-owned chunks that the caller frees on its own stream as
    // soon as the error propagates)
    if (fenced) { const int r2 = fence_out(V, v, join); if (rc == 0) rc = r2; }
    return rc;
}

static int forward_views_impl(int V, csplat_view *v, csplat_alloc_fn alloc, void *join_stream, int *pending) {
    CSPLAT_REQUIRE(V >= 0 && V <= MAX_TICKETS && (V == 0 || v != nullptr), "csplat_forward_views: bad view count");
    hipStream_t join = (hipStream_t)join_stream;
    int tickets[MAX_TICKETS];
    int rc = 0, begun = 0;
    if (pending) *pending = 0;
    for (; begun < V; begun++) {
        csplat_view &w = v[begun];
        rc = begin_prepare(w.stream, w.P, w.D, w.M, w.bg, w.W, w.H, w.means3D, w.shs, w.colors_precomp, w.opacities, w.scales,
                           w.scale_modifier, w.rotations, w.cov3D_precomp, w.view, w.proj, w.campos, w.tanfovx, w.tanfovy,
                           w.prefiltered, alloc, w.alloc_ctx, w.radii, &tickets[begun]);
        if (rc) break;
    }
    bool fenced = false;
    if (rc == 0 && begin_views_compatible(V, tickets)) {
        // first phase of all views in four launches on the join stream ...
        rc = begin_launch_views(V, tickets, join);
        // ... and, when every tile list fits the in-LDS sort, the second phase in five more, also on the join stream: no
        // side stream is involved at all
        bool done = false;
        PendingViews pend;
        if (rc == 0) rc = finish_views_batched(V, tickets, v, join, &done, pending ? 1 : 0, &pend);
        if (rc == 0 && done && pend.used) {      // deferred: park the call until csplat_forward_views_settle
            std::lock_guard<std::mutex> lk(g_pending_mu);
            int slot = -1;
            for (int i = 0; i < MAX_PENDING && slot < 0; i++)
                if (!g_pending[i].used) slot = i;
            if (slot >= 0) {
                g_pending[slot] = pend;
                *pending = 1;
                return 0;
            }
        }
        if (rc == 0 && done && pend.used) {      // (no free slot: settle right here)
            int rl = 0;
            done = false;
            rc = finish_views_batched(V, tickets, v, join, &done, 2, &pend, &rl);
        }
        if (done || rc) {
            if (rc) {
                std::lock_guard<std::mutex> lk(g_ticket_mu);
                for (int i = 0; i < begun; i++) g_tickets[tickets[i]].used = false;
            }
            return rc;
        }
        rc = fence_in(V, v, join);     // otherwise the views' streams branch off here
        fenced = rc == 0;
    } else if (rc == 0) {
        rc = fence_in(V, v, join);
        fenced = rc == 0;
        for (int i = 0; i < V && rc == 0; i++) rc = begin_launch(g_tickets[tickets[i]]);
    }
    return finish_views_one_by_one(V, v, tickets, begun, join, fenced, rc);
}

int csplat_forward_views(int V, csplat_view *v, csplat_alloc_fn alloc, void *join_stream) {
    return forward_views_impl(V, v, alloc, join_stream, nullptr);
}

// csplat_forward_views WITHOUT any host read: both phases are launched with the caller's capacities (caps[0] list entries per view,
// caps[1] longest tile list, caps[2] non-empty tiles) and *valid (device) receives 1 when every view's counts fitted them, else 0 -- in
// which case the second phase left the views untouched and csplat_backward_views (views[i].valid = valid) does nothing either.  Nothing
// here waits for the GPU or reads from it: the call can be recorded into a hipGraph (stream capture) and replayed.  The views must
// qualify for the one-launch-per-stage path (2..8 views sharing P, SH, opacities, scales and the image size), else an error.
int csplat_forward_views_faith(int V, csplat_view *v, csplat_alloc_fn alloc, void *join_stream, const uint32_t *caps, uint32_t *valid) {
    CSPLAT_REQUIRE(V >= 2 && V <= P2_MAX_VIEWS && v != nullptr && caps != nullptr && valid != nullptr, "csplat_forward_views_faith: bad arguments");
    CSPLAT_REQUIRE(caps[0] > 0 && caps[0] <= 0x7FFFFF00u && caps[1] > 0 && caps[1] <= tile_sort_cap() && caps[2] > 0,
                   "csplat_forward_views_faith: capacities out of range");
    hipStream_t join = (hipStream_t)join_stream;
    int tickets[P2_MAX_VIEWS];
    int rc = 0, begun = 0;
    for (; begun < V; begun++) {
        csplat_view &w = v[begun];
        rc = begin_prepare(w.stream, w.P, w.D, w.M, w.bg, w.W, w.H, w.means3D, w.shs, w.colors_precomp, w.opacities, w.scales,
                           w.scale_modifier, w.rotations, w.cov3D_precomp, w.view, w.proj, w.campos, w.tanfovx, w.tanfovy,
                           w.prefiltered, alloc, w.alloc_ctx, w.radii, &tickets[begun]);
        if (rc) break;
    }
    auto release = [&]() {
        std::lock_guard<std::mutex> lk(g_ticket_mu);
        for (int i = 0; i < begun; i++) g_tickets[tickets[i]].used = false;
    };
    if (rc == 0 && !begin_views_compatible(V, tickets)) {
        release();
        CSPLAT_REQUIRE(false, "csplat_forward_views_faith: the views do not qualify for the one-launch-per-stage path");
    }
    if (rc == 0) rc = begin_launch_views(V, tickets, join);
    if (rc == 0) {
        uint32_t Rcap[P2_MAX_VIEWS];
        for (int i = 0; i < V; i++) Rcap[i] = caps[0];
        const uint32_t tiles = (uint32_t)g_tickets[tickets[0]].tiles;
        rc = p2_launch(V, tickets, v, join, Rcap, caps[1], 1, caps[2] > tiles ? tiles : caps[2], valid);
    }
    for (int i = 0; i < V && rc == 0; i++) { v[i].num_rendered = (int)caps[0]; v[i].busy_tiles = 0; v[i].valid = valid; }
    release();
    return rc;
}
// byte offset, inside the IMAGE chunk, of the three counts a view's first phase leaves (u32: tile instances, longest tile list, non-empty
// tiles) -- what a caller that launched on faith reads, at a time of its choosing, to size the next launch
size_t csplat_image_info_offset(int W, int H) { size_t off[5]; image_offsets(W, H, off); return off[3]; }

// csplat_forward_views with the one host read DEFERRED.  When the second phase can be launched speculatively (capacities from the
// previous call of the same shape) the call returns right behind that launch with *pending = 1: views[i].layout_rendered is the capacity,
// views[i].num_rendered is -1, and the caller does whatever host work it has (the GPU is busy with K1..K6) before it calls
// csplat_forward_views_settle with the SAME array.  *pending = 0: the call was complete (first call of a shape, views that do not qualify).
int csplat_forward_views_deferred(int V, csplat_view *v, csplat_alloc_fn alloc, void *join_stream, int *pending) {
    CSPLAT_REQUIRE(pending != nullptr, "csplat_forward_views_deferred: pending missing");
    return forward_views_impl(V, v, alloc, join_stream, pending);
}

// Reads the counts of a pending call.  They fit the capacities: num_rendered is filled in, nothing else changes (*relaunched = 0).
// They do not: the second phase is repeated with exact sizes -- new BINNING chunks through the allocator of the call, layout_rendered /
// binning updated -- and *relaunched = 1: whatever the caller derived from layout_rendered must be rebuilt.
int csplat_forward_views_settle(int V, csplat_view *v, void *join_stream, int *relaunched) {
    CSPLAT_REQUIRE(v != nullptr && relaunched != nullptr, "csplat_forward_views_settle: bad arguments");
    *relaunched = 0;
    PendingViews pend;
    {
        std::lock_guard<std::mutex> lk(g_pending_mu);
        int slot = -1;
        for (int i = 0; i < MAX_PENDING && slot < 0; i++)
            if (g_pending[i].used && g_pending[i].key == v && g_pending[i].V == V) slot = i;
        CSPLAT_REQUIRE(slot >= 0, "csplat_forward_views_settle: no pending call for this view array");
        pend = g_pending[slot];
        g_pending[slot].used = false;
    }
    hipStream_t join = (hipStream_t)join_stream;
    bool done = false;
    int rc = finish_views_batched(V, pend.tk, v, join, &done, 2, &pend, relaunched);
    if (done || rc) {
        if (rc) {
            std::lock_guard<std::mutex> lk(g_ticket_mu);
            for (int i = 0; i < V; i++) g_tickets[pend.tk[i]].used = false;
        }
        return rc;
    }
    // the exact counts do not qualify for the batched phase (a list too long for the in-LDS sort, an empty view): view by view
    *relaunched = 1;
    rc = fence_in(V, v, join);
    return finish_views_one_by_one(V, v, pend.tk, V, join, rc == 0, rc);
}

// Can ONE K8 serve all views?  Same Gaussians (P, D, M, scale modifier, SH and scale tensors), SH staging applicable, and every
// gradient output either the SAME buffer in all views (then views after the first must have been asked to add into it) or
// a DIFFERENT buffer in every view.  Fills the table and returns true; anything else keeps the per-view launches.
static bool k8_views_table(int V, const csplat_view *v, K8Table &tab) {
    if (V < 2 || V > K8_MAX_VIEWS || (g_debug_flags & 128u)) return false;
    const csplat_view &a = v[0];
    if (a.P <= 0 || a.cov3D_precomp || !a.shs || !a.dL_dsh || a.M != 16 || !a.scales || !a.rotations || !a.dL_dscale || !a.dL_drot ||
        ((((uintptr_t)a.shs | (uintptr_t)a.dL_dsh) & 15u) != 0))
        return false;
    for (int i = 1; i < V; i++) {
        const csplat_view &w = v[i];
        if (w.P != a.P || w.D != a.D || w.M != a.M || w.scale_modifier != a.scale_modifier || w.shs != a.shs || w.dL_dsh != a.dL_dsh ||
            w.scales != a.scales || w.cov3D_precomp || !w.rotations || !w.dL_dscale || !w.dL_drot || !(w.accmask & CSPLAT_ACC_SH))
            return false;
    }
    unsigned sharedmask = 0;
    auto classify = [&](auto get, unsigned bit) {      // -> false when the buffers are neither all equal nor all different
        bool all_same = true, all_diff = true;
        for (int i = 0; i < V; i++)
            for (int j = i + 1; j < V; j++) {
                if (get(v[i]) == get(v[j])) all_diff = false; else all_same = false;
            }
        if (all_same) {
            for (int i = 1; i < V; i++)
                if (bit && !(v[i].accmask & bit)) return false;
            sharedmask |= bit;
            return true;
        }
        return all_diff;
    };
    if (!classify([](const csplat_view &w) { return (const void *)w.dL_dopacity; }, CSPLAT_ACC_OPACITY)) return false;
    if (!classify([](const csplat_view &w) { return (const void *)w.dL_dcolor; }, CSPLAT_ACC_COLOR)) return false;
    if (!classify([](const csplat_view &w) { return (const void *)w.dL_dmean3D; }, CSPLAT_ACC_MEAN3D)) return false;
    if (!classify([](const csplat_view &w) { return (const void *)w.dL_dcov3D; }, CSPLAT_ACC_COV3D)) return false;
    if (!classify([](const csplat_view &w) { return (const void *)w.dL_dscale; }, CSPLAT_ACC_SCALE)) return false;
    if (!classify([](const csplat_view &w) { return (const void *)w.dL_drot; }, CSPLAT_ACC_ROT)) return false;
    for (int i = 0; i < V; i++)      // mean2D / conic are per-view by construction
        for (int j = i + 1; j < V; j++)
            if (v[i].dL_dmean2D == v[j].dL_dmean2D || v[i].dL_dconic == v[j].dL_dconic || v[i].scratch == v[j].scratch) return false;
    tab.n = V;
    tab.sharedmask = sharedmask;
    tab.valid = v[0].valid;
    for (int i = 0; i < V; i++) {
        const csplat_view &w = v[i];
        if (!w.geom || !w.scratch || !w.dL_dmean2D || !w.dL_dconic || !w.dL_dopacity || !w.dL_dcolor || !w.dL_dmean3D || !w.dL_dcov3D ||
            !w.means3D || !w.radii)
            return false;
        K8View &k = tab.v[i];
        make_cam(k.cam, w.view, w.proj, w.campos, w.tanfovx, w.tanfovy, w.W, w.H);
        k.g = geom_view((void *)w.geom, w.P);
        k.radii = w.radii; k.acc = (const float *)w.scratch; k.means3D = w.means3D; k.rotations = w.rotations;
        k.dL_dmean2D = w.dL_dmean2D; k.dL_dconic = w.dL_dconic; k.dL_dopacity = w.dL_dopacity; k.dL_dcolor = w.dL_dcolor;
        k.dL_dmean3D = w.dL_dmean3D; k.dL_dcov3D = w.dL_dcov3D; k.dL_dscale = w.dL_dscale; k.dL_drot = w.dL_drot;
        k.accmask = w.accmask;
    }
    return true;
}

// csplat_backward_views cut into parts (round 6: the gradient exchange of a view-parallel step starts before the backward has ended).
// parts bit 0: the compositing backward (K7) of all views; bit 1: the per-Gaussian backward (K8) for SLICE `slice` of `nslices` equal
// ranges of Gaussians (boundaries at multiples of 32: csplat_backward_slice_rows) -- a caller launches K7 once, then the K8 slices one by
// one, and may hand the gradient rows of slice g to its collective while slice g + 1 computes.  Only the one-launch-per-stage path can be
// cut (the views share P, SH, scales and the image size, as csplat_forward_views_faith requires); parts == 3 with one slice is
// csplat_backward_views.  The sum of the parts is the whole call bit for bit: every Gaussian's arithmetic is the same in any slicing.
static int backward_views_impl(int V, csplat_view *v, void *join_stream, unsigned parts, int slice, int nslices) {
    CSPLAT_REQUIRE(V >= 0 && (V == 0 || v != nullptr), "csplat_backward_views: bad view count");
    CSPLAT_REQUIRE(parts >= 1 && parts <= 3 && nslices >= 1 && slice >= 0 && slice < nslices, "csplat_backward_views_parts: bad parts / slice");
    const bool want_k7 = (parts & 1u) != 0, want_k8 = (parts & 2u) != 0, whole = parts == 3u && nslices == 1;
    hipStream_t join = (hipStream_t)join_stream;
    bool shared = false;   // any view adding into another view's buffers: all K8 run on the join stream, in view order
    for (int i = 0; i < V; i++) shared |= (v[i].accmask & ~(unsigned)CSPLAT_SCRATCH_ZEROED) != 0u;
    K8Table tab;
    const bool one_k8 = shared && k8_views_table(V, v, tab);
    // K7 of all views in ONE launch on the join stream (plus one launch clearing the records) when the views are alike
    const bool det_mode = (g_debug_flags & 256u) != 0;
    bool batch_k7 = V >= 2 && V <= B2_MAX_VIEWS && !(g_debug_flags & 512u);
    for (int i = 0; i < V && batch_k7; i++)
        batch_k7 = v[i].P == v[0].P && v[i].P > 0 && v[i].W == v[0].W && v[i].H == v[0].H && v[i].num_rendered > 0 && v[i].geom &&
                   v[i].binning && v[i].image && v[i].out_color && v[i].scratch && v[i].dL_dpix;
    // one launch per stage for all views: everything runs on the join stream, the views' own streams are not involved and need
    // neither the entry nor the exit fence (six event / wait calls, ~25 us of host time per step)
    const bool lone = V == 1 && (hipStream_t)v[0].stream == join;      // one view on the caller's stream: nothing to fence
    const bool side_streams = !(batch_k7 && one_k8) && !lone;
    CSPLAT_REQUIRE(whole || (batch_k7 && one_k8), "csplat_backward_views_parts: only the one-launch-per-stage path can be cut into parts");
    CSPLAT_REQUIRE(!(V > 0 && v[0].valid) || !side_streams, "csplat_backward_views: views launched on faith need the one-launch-per-stage path");
    if (side_streams)
        if (int rc = fence_in(V, v, join)) return rc;
    // from here on side streams may hold work on caller-owned buffers: whatever fails, the exit fence is still issued
    auto body = [&]() -> int {
        if (batch_k7 && want_k7) {
            const int W = v[0].W, H = v[0].H, P = v[0].P, gx = cdiv(W, CSPLAT_TILE), tiles = gx * cdiv(H, CSPLAT_TILE);
            B2Table bt;
            DetTable dt;
            dt.valid = v[0].valid;
            int64_t slots = 0;
            size_t ioff[5];
            image_offsets(W, H, ioff);
            for (int i = 0; i < V; i++) {
                const csplat_view &w = v[i];
                size_t boff[B_NFIELDS];
                const int Rl = w.layout_rendered > 0 ? w.layout_rendered : w.num_rendered;   // what the chunk was laid out for
                binning_offsets(Rl, tiles, boff);
                const char *b = (const char *)w.binning, *im = (const char *)w.image;
                B2View &k = bt.v[i];
                k.ranges = (const int2 *)(im + ioff[0]); k.n_contrib = (const uint32_t *)(im + ioff[1]); k.final_T = (const float *)(im + ioff[2]);
                k.ids_sorted = (const uint32_t *)(b + boff[1]); k.seg_offset = (const int *)(b + boff[2]); k.slot_tile = (const int *)(b + boff[3]);
                k.ckpt = (const float4 *)(b + boff[4]); k.bbits = (const unsigned long long *)(b + boff[9]);
                k.recA = (const float4 *)(b + boff[6]); k.recB = (const float4 *)(b + boff[7]); k.recC = (const float2 *)(b + boff[8]);
                k.out_color = w.out_color; k.dL_dpix = w.dL_dpix; k.acc = (float *)w.scratch; k.R = (uint32_t)Rl;
                k.det = det_mode ? (float *)((char *)w.scratch + align256((size_t)P * ACC_STRIDE * 4)) : nullptr;
                if (det_mode) {
                    DetView &d = dt.v[i];
                    make_cam(d.cam, w.view, w.proj, w.campos, w.tanfovx, w.tanfovy, w.W, w.H);
                    const Geom g = geom_view((void *)w.geom, P);
                    d.xy = g.xy; d.depth = g.depth; d.radii = w.radii; d.ranges = k.ranges; d.keys_sorted = (const uint64_t *)(b + boff[0]);
                    d.ids_sorted = k.ids_sorted; d.det = k.det; d.acc = k.acc;
                }
                // segments of the view: <= R / SEG + (non-empty tiles) + 1 with the EXACT counts the forward read -- the layout's bound
                // (capacity / SEG + all tiles + 1) launches twice as many workgroups that find no segment
                const int64_t sl = (w.busy_tiles > 0 && w.num_rendered > 0) ? (int64_t)w.num_rendered / SEG + w.busy_tiles + 1 : max_slots(Rl, tiles);
                slots = sl > slots ? sl : slots;
            }
            {   // (measurement hook, off unless csplat_debug_stamps handed over a buffer large enough for this launch)
                const size_t need = (size_t)V * ((size_t)cdiv(slots, 8) * 32u) * 12;
                bt.stamp = (g_stamp_buf && g_stamp_words >= need) ? g_stamp_buf : nullptr;
                bt.valid = v[0].valid;
            }
            bool zeroed = true;      // every view's records are zero already (CSPLAT_SCRATCH_ZEROED) and K8 will leave them so: no clearing launch
            for (int i = 0; i < V; i++) zeroed = zeroed && (v[i].accmask & CSPLAT_SCRATCH_ZEROED);
            if (det_mode) {          // (k_det_reduce_views writes every record: nothing to clear but the (entry, block) records)
                k_zero_det_views<<<dim3(1024, V), 256, 0, join>>>(bt);
                LAUNCH_CHECK();
            } else if (!zeroed) {
                const int64_t n4 = (int64_t)P * ACC_STRIDE / 4;
                k_zero_acc_views<<<dim3((unsigned)(cdiv(n4, 256) > 1024 ? 1024 : cdiv(n4, 256)), V), 256, 0, join>>>(n4, bt);
                LAUNCH_CHECK();
            }
            ProfScope ps(PROF_K7, join);       // (the bracket bench.py's roofline reads: K7's launch alone, not the record clearing in front of it)
            const unsigned items = (unsigned)cdiv(slots, 8) * 32u;
            if (det_mode) {
                k_composite_bwd_rows_views_det<<<dim3(items, V), 256, 0, join>>>(tiles, W, H, gx, bt);
                LAUNCH_CHECK();
                k_det_reduce_views<<<dim3((unsigned)cdiv(P, 256), V), 256, 0, join>>>(P, dt);
            } else
                k_composite_bwd_rows_views<<<dim3(items, V), 256, 0, join>>>(tiles, W, H, gx, bt);
            LAUNCH_CHECK();
        }
        for (int i = 0; i < V && !(batch_k7 && one_k8); i++) {
            const csplat_view &w = v[i];
            if (int rc = backward_impl((hipStream_t)w.stream, (shared || batch_k7) ? join : (hipStream_t)w.stream, !batch_k7, !one_k8,
                                       w.accmask, w.P, w.D, w.M,
                                       w.layout_rendered > 0 ? w.layout_rendered : w.num_rendered, w.bg, w.W, w.H, w.means3D, w.shs, w.scales, w.scale_modifier, w.rotations,
                                       w.cov3D_precomp, w.view, w.proj, w.campos, w.tanfovx, w.tanfovy, w.radii, w.geom, w.binning,
                                       w.image, w.out_color, w.dL_dpix, w.scratch, w.dL_dmean2D, w.dL_dconic, w.dL_dopacity,
                                       w.dL_dcolor, w.dL_dmean3D, w.dL_dcov3D, w.dL_dsh, w.dL_dscale, w.dL_drot))
                return rc;
        }
        if (one_k8 && want_k8) {   // every view's K7 is queued on its own stream: the join stream waits for all of them, then ONE K8
            for (int i = 0; i < V && !batch_k7; i++) {
                if ((hipStream_t)v[i].stream == join) continue;
                hipEvent_t ev = pooled_event();
                CSPLAT_REQUIRE(ev != nullptr, "csplat_backward_views: no event");
                HIP_TRY(hipEventRecord(ev, (hipStream_t)v[i].stream));
                HIP_TRY(hipStreamWaitEvent(join, ev, 0));
            }
            ProfScope ps(PROF_K8, join);
            const csplat_view &a = v[0];
            const int nb = cdiv(a.P, 32);
            const int b_lo = (int)((int64_t)nb * slice / nslices), b_hi = (int)((int64_t)nb * (slice + 1) / nslices);
            if (b_hi > b_lo) {
                k_preprocess_bwd_views<128, 4><<<b_hi - b_lo, 128, 0, join>>>(a.P, a.D, a.M, a.shs, a.scales, a.scale_modifier, 0, a.dL_dsh, tab, b_lo);
                LAUNCH_CHECK();
            }
        }
        return 0;
    };
    const int rc = body();
    const int r2 = side_streams ? fence_out(V, v, join) : 0;
    return rc ? rc : r2;
}
int csplat_backward_views(int V, csplat_view *v, void *join_stream) { return backward_views_impl(V, v, join_stream, 3u, 0, 1); }
int csplat_backward_views_parts(int V, csplat_view *v, void *join_stream, unsigned parts, int slice, int nslices) {
    return backward_views_impl(V, v, join_stream, parts, slice, nslices);
}
// rows [*row_lo, *row_hi) of the P Gaussians that K8 slice `slice` of `nslices` finishes
int csplat_backward_slice_rows(int P, int slice, int nslices, int64_t *row_lo, int64_t *row_hi) {
    CSPLAT_REQUIRE(P >= 0 && nslices >= 1 && slice >= 0 && slice < nslices && row_lo && row_hi, "csplat_backward_slice_rows: bad arguments");
    const int64_t nb = cdiv(P, 32);
    const int64_t lo = nb * slice / nslices * 32, hi = nb * (slice + 1) / nslices * 32;
    *row_lo = lo < P ? lo : P; *row_hi = hi < P ? hi : P;
    return 0;
}

}  // extern "C"
