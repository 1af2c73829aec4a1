/*
 * oracle/raster_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, scalar, one Gaussian / one pixel at a time) of the
 * depth-aware differentiable Gaussian rasterizer that cloth-splatting calls at
 *   /root/reference/gaussian_renderer/__init__.py:156-164  (forward)
 *   /root/reference/scene_reconstruction/train_utils.py:288 (backward, via autograd)
 *
 * PARITY UNPINNED: the arithmetic lives in the third-party CUDA extension
 * `ingra14m/depth-diff-gaussian-rasterization` (a fork of
 * `graphdeco-inria/diff-gaussian-rasterization`), an un-vendored, un-pinned git
 * submodule (/root/reference/.gitmodules:7-9; directory empty).  Neither source nor
 * golden vectors exist in the reference, so this file restates the PUBLISHED
 * algorithm of that project and is anchored on the
 * reference's call site, argument conventions (row-vector 4x4 matrices,
 * gaussian_renderer/__init__.py:61-74; scene_reconstruction/cameras.py:63-67) and the
 * importable pieces (utils/sh_utils.py:57-112 for SH->RGB, utils/general_utils.py:81-102
 * for quaternion->rotation), which tests/golden pins.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Build: see oracle/Makefile (REAL=float -> liboracle_f32.so, REAL=double -> liboracle_f64.so),
 * compiled with -ffp-contract=off so that every index-deciding expression (radius,
 * tile rectangle, sort key) rounds exactly as written; the HIP preprocess kernel is
 * compiled the same way and written in the same association order.
 *
 * Conventions reproduced (upstream behaviour):
 *   - tile 16x16, near plane 0.2, +0.3 low-pass on cov2D diagonal, 1.3*tanfov clamp,
 *     alpha cap 0.99 (straight-through in backward), alpha skip < 1/255,
 *     stop when T*(1-alpha) < 1e-4 (that Gaussian is not blended),
 *     1/(w + 1e-7) homogeneous divide, 1/(det^2 + 1e-7) guard in backward,
 *   - dL/dmeans2D is returned in NDC units (x 0.5*W, 0.5*H),
 *   - dL/dscale omits the scale_modifier factor (upstream quirk; exact for modifier 1),
 *   - the depth image carries no gradient,
 *   - quaternion (r,x,y,z) is used as passed (not re-normalised).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif

#define TILE 16
#define NEAR_Z ((REAL)0.2)

#ifdef _OPENMP
#include <omp.h>
#endif

static inline REAL rsqrt_(REAL x) { return (sizeof(REAL) == 4) ? (REAL)sqrtf((float)x) : (REAL)sqrt((double)x); }
static inline REAL rexp_(REAL x) { return (sizeof(REAL) == 4) ? (REAL)expf((float)x) : (REAL)exp((double)x); }
static inline REAL rceil_(REAL x) { return (sizeof(REAL) == 4) ? (REAL)ceilf((float)x) : (REAL)ceil((double)x); }
static inline REAL rmax_(REAL a, REAL b) { return a > b ? a : b; }
static inline REAL rmin_(REAL a, REAL b) { return a < b ? a : b; }
static inline int imin_(int a, int b) { return a < b ? a : b; }
static inline int imax_(int a, int b) { return a > b ? a : b; }

/* SH basis constants: utils/sh_utils.py:26-54 */
static const REAL SH_C0 = (REAL)0.28209479177387814;
static const REAL SH_C1 = (REAL)0.4886025119029199;
static const REAL SH_C2[5] = {(REAL)1.0925484305920792, (REAL)-1.0925484305920792, (REAL)0.31539156525252005,
                              (REAL)-1.0925484305920792, (REAL)0.5462742152960396};
static const REAL SH_C3[7] = {(REAL)-0.5900435899266435, (REAL)2.890611442640554, (REAL)-0.4570457994644658,
                              (REAL)0.3731763325901154, (REAL)-0.4570457994644658, (REAL)1.445305721320277,
                              (REAL)-0.5900435899266435};

int oracle_real_bytes(void) { return (int)sizeof(REAL); }

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* small problems (a 200x200 image, a few thousand Gaussians) run slower on 100+ threads than on a few */
void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* quaternion (r,x,y,z) -> rotation rows; utils/general_utils.py:81-102 (without the normalise) */
static void quat_to_rot(const REAL *q, REAL R[3][3]) {
    REAL r = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = (REAL)1 - (REAL)2 * (y * y + z * z);
    R[0][1] = (REAL)2 * (x * y - r * z);
    R[0][2] = (REAL)2 * (x * z + r * y);
    R[1][0] = (REAL)2 * (x * y + r * z);
    R[1][1] = (REAL)1 - (REAL)2 * (x * x + z * z);
    R[1][2] = (REAL)2 * (y * z - r * x);
    R[2][0] = (REAL)2 * (x * z - r * y);
    R[2][1] = (REAL)2 * (y * z + r * x);
    R[2][2] = (REAL)1 - (REAL)2 * (x * x + y * y);
}

/* Sigma = R diag(s)^2 R^T stored as (xx,xy,xz,yy,yz,zz); s = mod*scale.
 * m[k][i] = s_k * R[i][k]  (upstream M = S * R_glm), Sigma_ij = sum_k m[k][i]*m[k][j]. */
static void cov3d_from_scale_rot(const REAL *scale, REAL mod, const REAL *q, REAL *cov6) {
    REAL R[3][3], m[3][3];
    quat_to_rot(q, R);
    for (int k = 0; k < 3; k++) {
        REAL s = mod * scale[k];
        for (int i = 0; i < 3; i++) m[k][i] = s * R[i][k];
    }
    cov6[0] = m[0][0] * m[0][0] + m[1][0] * m[1][0] + m[2][0] * m[2][0];
    cov6[1] = m[0][0] * m[0][1] + m[1][0] * m[1][1] + m[2][0] * m[2][1];
    cov6[2] = m[0][0] * m[0][2] + m[1][0] * m[1][2] + m[2][0] * m[2][2];
    cov6[3] = m[0][1] * m[0][1] + m[1][1] * m[1][1] + m[2][1] * m[2][1];
    cov6[4] = m[0][1] * m[0][2] + m[1][1] * m[1][2] + m[2][1] * m[2][2];
    cov6[5] = m[0][2] * m[0][2] + m[1][2] * m[1][2] + m[2][2] * m[2][2];
}

/* rows of (J * Rw): t0[a], t1[a]; also returns clamped view coords and clamp flags. */
typedef struct {
    REAL t0[3], t1[3];
    REAL tx, ty, tz; /* tx,ty after the 1.3*tanfov clamp */
    int x_in, y_in;
} proj_jac_t;

static void view_point(const REAL *p, const REAL *V, REAL *out) {
    out[0] = V[0] * p[0] + V[4] * p[1] + V[8] * p[2] + V[12];
    out[1] = V[1] * p[0] + V[5] * p[1] + V[9] * p[2] + V[13];
    out[2] = V[2] * p[0] + V[6] * p[1] + V[10] * p[2] + V[14];
}

static void proj_jacobian(const REAL *pv, const REAL *V, REAL fx, REAL fy, REAL tanfovx, REAL tanfovy,
                          proj_jac_t *o) {
    REAL limx = (REAL)1.3 * tanfovx, limy = (REAL)1.3 * tanfovy;
    REAL tz = pv[2];
    REAL txtz = pv[0] / tz, tytz = pv[1] / tz;
    o->x_in = !(txtz < -limx || txtz > limx);
    o->y_in = !(tytz < -limy || tytz > limy);
    REAL tx = rmin_(limx, rmax_(-limx, txtz)) * tz;
    REAL ty = rmin_(limy, rmax_(-limy, tytz)) * tz;
    REAL J00 = fx / tz, J02 = -(fx * tx) / (tz * tz);
    REAL J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
    /* Rw[k][a] = V[4*a + k] (V is the transposed, i.e. column-major, world->view matrix) */
    for (int a = 0; a < 3; a++) {
        o->t0[a] = V[4 * a + 0] * J00 + V[4 * a + 2] * J02;
        o->t1[a] = V[4 * a + 1] * J11 + V[4 * a + 2] * J12;
    }
    o->tx = tx; o->ty = ty; o->tz = tz;
}

static void cov2d_from_cov3d(const REAL *c6, const proj_jac_t *pj, REAL *a, REAL *b, REAL *c) {
    const REAL *t0 = pj->t0, *t1 = pj->t1;
    REAL Vm[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    REAL u0[3], u1[3];
    for (int j = 0; j < 3; j++) {
        u0[j] = t0[0] * Vm[0][j] + t0[1] * Vm[1][j] + t0[2] * Vm[2][j];
        u1[j] = t1[0] * Vm[0][j] + t1[1] * Vm[1][j] + t1[2] * Vm[2][j];
    }
    *a = (u0[0] * t0[0] + u0[1] * t0[1] + u0[2] * t0[2]) + (REAL)0.3;
    *b = u0[0] * t1[0] + u0[1] * t1[1] + u0[2] * t1[2];
    *c = (u1[0] * t1[0] + u1[1] * t1[1] + u1[2] * t1[2]) + (REAL)0.3;
}

static void sh_to_rgb(int deg, int M, const REAL *p, const REAL *campos, const REAL *sh /* [M][3] */, REAL *rgb,
                      uint8_t *clamped) {
    REAL d[3] = {p[0] - campos[0], p[1] - campos[1], p[2] - campos[2]};
    REAL len = rsqrt_(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    REAL x = d[0] / len, y = d[1] / len, z = d[2] / len;
    (void)M;
    for (int ch = 0; ch < 3; ch++) {
#define S(i) sh[(i) * 3 + ch]
        REAL r = SH_C0 * S(0);
        if (deg > 0) {
            r = r - SH_C1 * y * S(1) + SH_C1 * z * S(2) - SH_C1 * x * S(3);
            if (deg > 1) {
                REAL xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + SH_C2[0] * xy * S(4) + SH_C2[1] * yz * S(5) + SH_C2[2] * ((REAL)2 * zz - xx - yy) * S(6) +
                    SH_C2[3] * xz * S(7) + SH_C2[4] * (xx - yy) * S(8);
                if (deg > 2) {
                    r = r + SH_C3[0] * y * ((REAL)3 * xx - yy) * S(9) + SH_C3[1] * xy * z * S(10) +
                        SH_C3[2] * y * ((REAL)4 * zz - xx - yy) * S(11) +
                        SH_C3[3] * z * ((REAL)2 * zz - (REAL)3 * xx - (REAL)3 * yy) * S(12) +
                        SH_C3[4] * x * ((REAL)4 * zz - xx - yy) * S(13) + SH_C3[5] * z * (xx - yy) * S(14) +
                        SH_C3[6] * x * (xx - (REAL)3 * yy) * S(15);
                }
            }
        }
#undef S
        r += (REAL)0.5;
        clamped[ch] = (uint8_t)(r < (REAL)0);
        rgb[ch] = rmax_(r, (REAL)0);
    }
}

/* ------------------------------------------------------------------------------------------
 * K1: per-Gaussian preprocess.  Outputs (all length-P arrays, zero where culled):
 *   depth[P], radii[P] (int32), xy[P][2], conic_opacity[P][4], rgb[P][3], clamped[P][3] (u8),
 *   cov3D[P][6], tiles_touched[P] (u32), rect[P][4] (minx,miny,maxx,maxy; diagnostic).
 * Returns sum(tiles_touched) = number of tile instances R.
 * ------------------------------------------------------------------------------------------ */
int64_t oracle_preprocess(int P, int D, int M, int W, int H, const REAL *means3D, const REAL *shs,
                          const REAL *colors_precomp, const REAL *opacities, const REAL *scales, REAL scale_mod,
                          const REAL *rotations, const REAL *cov3D_precomp, const REAL *view, const REAL *proj,
                          const REAL *campos, REAL tanfovx, REAL tanfovy, REAL *depth, int32_t *radii, REAL *xy,
                          REAL *conic_opacity, REAL *rgb, uint8_t *clamped, REAL *cov3D, uint32_t *tiles_touched,
                          int32_t *rect) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const REAL fx = (REAL)W / ((REAL)2 * tanfovx), fy = (REAL)H / ((REAL)2 * tanfovy);
    int64_t total = 0;
#pragma omp parallel for schedule(static) reduction(+ : total)
    for (int i = 0; i < P; i++) {
        depth[i] = 0; radii[i] = 0; tiles_touched[i] = 0;
        xy[2 * i] = xy[2 * i + 1] = 0;
        for (int k = 0; k < 4; k++) conic_opacity[4 * i + k] = 0;
        for (int k = 0; k < 3; k++) { rgb[3 * i + k] = 0; clamped[3 * i + k] = 0; }
        for (int k = 0; k < 6; k++) cov3D[6 * i + k] = 0;
        if (rect) for (int k = 0; k < 4; k++) rect[4 * i + k] = 0;

        const REAL *p = means3D + 3 * i;
        REAL pv[3];
        view_point(p, view, pv);
        if (pv[2] <= NEAR_Z) continue;

        REAL hx = proj[0] * p[0] + proj[4] * p[1] + proj[8] * p[2] + proj[12];
        REAL hy = proj[1] * p[0] + proj[5] * p[1] + proj[9] * p[2] + proj[13];
        REAL hw = proj[3] * p[0] + proj[7] * p[1] + proj[11] * p[2] + proj[15];
        REAL pw = (REAL)1 / (hw + (REAL)0.0000001);
        REAL ndcx = hx * pw, ndcy = hy * pw;

        REAL c6[6];
        if (cov3D_precomp) memcpy(c6, cov3D_precomp + 6 * i, sizeof(c6));
        else cov3d_from_scale_rot(scales + 3 * i, scale_mod, rotations + 4 * i, c6);
        for (int k = 0; k < 6; k++) cov3D[6 * i + k] = c6[k];

        proj_jac_t pj;
        proj_jacobian(pv, view, fx, fy, tanfovx, tanfovy, &pj);
        REAL a, b, c;
        cov2d_from_cov3d(c6, &pj, &a, &b, &c);
        REAL det = a * c - b * b;
        if (det == (REAL)0) continue;
        REAL det_inv = (REAL)1 / det;
        REAL mid = (REAL)0.5 * (a + c);
        REAL sq = rsqrt_(rmax_((REAL)0.1, mid * mid - det));
        REAL lam1 = mid + sq, lam2 = mid - sq;
        REAL my_radius = rceil_((REAL)3 * rsqrt_(rmax_(lam1, lam2)));
        REAL px = ((ndcx + (REAL)1) * (REAL)W - (REAL)1) * (REAL)0.5;
        REAL py = ((ndcy + (REAL)1) * (REAL)H - (REAL)1) * (REAL)0.5;
        int rad = (int)my_radius;
        int minx = imin_(gx, imax_(0, (int)((px - (REAL)rad) / (REAL)TILE)));
        int miny = imin_(gy, imax_(0, (int)((py - (REAL)rad) / (REAL)TILE)));
        int maxx = imin_(gx, imax_(0, (int)((px + (REAL)rad + (REAL)(TILE - 1)) / (REAL)TILE)));
        int maxy = imin_(gy, imax_(0, (int)((py + (REAL)rad + (REAL)(TILE - 1)) / (REAL)TILE)));
        if ((maxx - minx) * (maxy - miny) == 0) continue;

        if (colors_precomp) {
            for (int k = 0; k < 3; k++) rgb[3 * i + k] = colors_precomp[3 * i + k];
        } else {
            sh_to_rgb(D, M, p, campos, shs + (size_t)i * M * 3, rgb + 3 * i, clamped + 3 * i);
        }
        depth[i] = pv[2];
        radii[i] = rad;
        xy[2 * i] = px; xy[2 * i + 1] = py;
        conic_opacity[4 * i + 0] = c * det_inv;
        conic_opacity[4 * i + 1] = -b * det_inv;
        conic_opacity[4 * i + 2] = a * det_inv;
        conic_opacity[4 * i + 3] = opacities[i];
        tiles_touched[i] = (uint32_t)((maxy - miny) * (maxx - minx));
        if (rect) { rect[4 * i] = minx; rect[4 * i + 1] = miny; rect[4 * i + 2] = maxx; rect[4 * i + 3] = maxy; }
        total += tiles_touched[i];
    }
    return total;
}

/* ------------------------------------------------------------------------------------------
 * K2-K5: binning.  keys = (tile_id << 32) | float_bits(depth); stable sort; tile ranges.
 * The stable sort is an LSD radix sort over all 64 key bits (sorting more bits than upstream's
 * 32+ceil(log2 tiles) cannot change the order: the extra bits are zero).
 * ------------------------------------------------------------------------------------------ */
static void radix_sort_pairs(uint64_t *k, uint32_t *v, uint64_t *k2, uint32_t *v2, int64_t n) {
    for (int pass = 0; pass < 8; pass++) {
        int64_t cnt[257];
        memset(cnt, 0, sizeof(cnt));
        int sh = pass * 8;
        for (int64_t i = 0; i < n; i++) cnt[((k[i] >> sh) & 0xFF) + 1]++;
        int single = 0;
        for (int d = 1; d <= 256; d++) if (cnt[d] == n) single = 1;
        if (single) continue; /* digit constant: pass is the identity */
        for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
        for (int64_t i = 0; i < n; i++) {
            int64_t dst = cnt[(k[i] >> sh) & 0xFF]++;
            k2[dst] = k[i]; v2[dst] = v[i];
        }
        memcpy(k, k2, (size_t)n * sizeof(uint64_t));
        memcpy(v, v2, (size_t)n * sizeof(uint32_t));
    }
}

int oracle_bin(int P, int W, int H, const REAL *depth, const int32_t *radii, const REAL *xy,
               const uint32_t *tiles_touched, int64_t R, uint64_t *keys_unsorted /* may be NULL */,
               uint64_t *keys_sorted, uint32_t *ids_sorted, int32_t *ranges /* [tiles][2] */) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    uint64_t *k2 = (uint64_t *)malloc((size_t)(R > 0 ? R : 1) * sizeof(uint64_t));
    uint32_t *v2 = (uint32_t *)malloc((size_t)(R > 0 ? R : 1) * sizeof(uint32_t));
    if (!k2 || !v2) return -1;
    int64_t off = 0;
    for (int i = 0; i < P; i++) {
        if (radii[i] <= 0) continue;
        REAL px = xy[2 * i], py = xy[2 * i + 1];
        int rad = radii[i];
        int minx = imin_(gx, imax_(0, (int)((px - (REAL)rad) / (REAL)TILE)));
        int miny = imin_(gy, imax_(0, (int)((py - (REAL)rad) / (REAL)TILE)));
        int maxx = imin_(gx, imax_(0, (int)((px + (REAL)rad + (REAL)(TILE - 1)) / (REAL)TILE)));
        int maxy = imin_(gy, imax_(0, (int)((py + (REAL)rad + (REAL)(TILE - 1)) / (REAL)TILE)));
        float df = (float)depth[i];
        uint32_t dbits;
        memcpy(&dbits, &df, 4);
        for (int y = miny; y < maxy; y++)
            for (int x = minx; x < maxx; x++) {
                uint64_t key = ((uint64_t)(uint32_t)(y * gx + x) << 32) | dbits;
                keys_sorted[off] = key;
                ids_sorted[off] = (uint32_t)i;
                off++;
            }
        (void)tiles_touched;
    }
    if (off != R) { free(k2); free(v2); return -2; }
    if (keys_unsorted) memcpy(keys_unsorted, keys_sorted, (size_t)R * sizeof(uint64_t));
    radix_sort_pairs(keys_sorted, ids_sorted, k2, v2, R);
    free(k2); free(v2);
    for (int t = 0; t < gx * gy; t++) ranges[2 * t] = ranges[2 * t + 1] = 0;
    for (int64_t i = 0; i < R; i++) {
        uint32_t t = (uint32_t)(keys_sorted[i] >> 32);
        if (i == 0 || (uint32_t)(keys_sorted[i - 1] >> 32) != t) ranges[2 * t] = (int32_t)i;
        if (i == R - 1 || (uint32_t)(keys_sorted[i + 1] >> 32) != t) ranges[2 * t + 1] = (int32_t)(i + 1);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * K6: front-to-back compositing of colour + depth per pixel.
 * ------------------------------------------------------------------------------------------ */
void oracle_render_fwd(int W, int H, const int32_t *ranges, const uint32_t *ids_sorted, const REAL *xy,
                       const REAL *conic_opacity, const REAL *rgb, const REAL *depth, const REAL *bg,
                       REAL *out_color /* [3][H][W] */, REAL *out_depth /* [H][W] */, REAL *final_T,
                       uint32_t *n_contrib) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++)
    for (int lp = 0; lp < TILE * TILE; lp++) {
        int px = (tile % gx) * TILE + lp % TILE, py = (tile / gx) * TILE + lp / TILE;
        if (px >= W || py >= H) continue;
        int pix = py * W + px;
        int s = ranges[2 * tile], e = ranges[2 * tile + 1];
        REAL T = 1, C[3] = {0, 0, 0}, Dp = 0;
        uint32_t contributor = 0, last = 0;
        for (int j = s; j < e; j++) {
            contributor++;
            uint32_t g = ids_sorted[j];
            REAL dx = xy[2 * g] - (REAL)px, dy = xy[2 * g + 1] - (REAL)py;
            const REAL *co = conic_opacity + 4 * g;
            REAL power = (REAL)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
            if (power > (REAL)0) continue;
            REAL alpha = rmin_((REAL)0.99, co[3] * rexp_(power));
            if (alpha < (REAL)1 / (REAL)255) continue;
            REAL test_T = T * ((REAL)1 - alpha);
            if (test_T < (REAL)0.0001) break;
            for (int ch = 0; ch < 3; ch++) C[ch] += rgb[3 * g + ch] * alpha * T;
            Dp += depth[g] * alpha * T;
            T = test_T;
            last = contributor;
        }
        final_T[pix] = T;
        n_contrib[pix] = last;
        for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * H * W + pix] = C[ch] + T * bg[ch];
        out_depth[pix] = Dp;
    }
}

/* ------------------------------------------------------------------------------------------
 * K7: back-to-front replay; accumulates per-Gaussian dL/d(mean2D, conic, opacity, colour).
 * Sequential over pixels within a tile so that the per-Gaussian sums have ONE defined order
 * (pixel-major inside a tile, tiles in parallel into private buffers is avoided: we parallelise
 * over tiles and use per-thread accumulation buffers reduced in thread order).
 * dL_dconic is stored as upstream does: [P][4] with (x,y,_,w) = (a, b/2-weighted, unused, c).
 * ------------------------------------------------------------------------------------------ */
void oracle_render_bwd(int P, int W, int H, const int32_t *ranges, const uint32_t *ids_sorted, const REAL *xy,
                       const REAL *conic_opacity, const REAL *rgb, const REAL *bg, const REAL *final_T,
                       const uint32_t *n_contrib, const REAL *dL_dpix /* [3][H][W] */, REAL *dL_dmean2D /* [P][3] */,
                       REAL *dL_dconic /* [P][4] */, REAL *dL_dopacity /* [P] */, REAL *dL_dcolor /* [P][3] */) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    memset(dL_dmean2D, 0, (size_t)P * 3 * sizeof(REAL));
    memset(dL_dconic, 0, (size_t)P * 4 * sizeof(REAL));
    memset(dL_dopacity, 0, (size_t)P * sizeof(REAL));
    memset(dL_dcolor, 0, (size_t)P * 3 * sizeof(REAL));
    const REAL ddelx_dx = (REAL)0.5 * (REAL)W, ddely_dy = (REAL)0.5 * (REAL)H;
    /* one accumulator row per LIST ENTRY (a tile's entries are only touched by the thread that owns the tile: no races, no
     * per-thread copies), in double; afterwards every Gaussian's rows are summed in list order by one thread -- the result does
     * not depend on the number of threads or on which thread took which tile */
    size_t Rtot = 0;
    for (int t = 0; t < gx * gy; t++)
        if ((size_t)ranges[2 * t + 1] > Rtot) Rtot = (size_t)ranges[2 * t + 1];
    double *inst = (double *)calloc((Rtot ? Rtot : 1) * 9, sizeof(double));
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        int s = ranges[2 * tile], e = ranges[2 * tile + 1];
        if (e <= s) continue;
        int tx0 = (tile % gx) * TILE, ty0 = (tile / gx) * TILE;
        for (int ly = 0; ly < TILE; ly++)
            for (int lx = 0; lx < TILE; lx++) {
                int px = tx0 + lx, py = ty0 + ly;
                if (px >= W || py >= H) continue;
                int pix = py * W + px;
                REAL T_final = final_T[pix], T = T_final;
                int last = (int)n_contrib[pix];
                REAL accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, last_alpha = 0;
                REAL dpx[3];
                for (int ch = 0; ch < 3; ch++) dpx[ch] = dL_dpix[(size_t)ch * H * W + pix];
                REAL bg_dot = bg[0] * dpx[0] + bg[1] * dpx[1] + bg[2] * dpx[2];
                for (int j = s + last - 1; j >= s; j--) {
                    uint32_t g = ids_sorted[j];
                    double *A = inst + (size_t)j * 9;
                    REAL dx = xy[2 * g] - (REAL)px, dy = xy[2 * g + 1] - (REAL)py;
                    const REAL *co = conic_opacity + 4 * g;
                    REAL power = (REAL)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > (REAL)0) continue;
                    REAL G = rexp_(power);
                    REAL alpha = rmin_((REAL)0.99, co[3] * G);
                    if (alpha < (REAL)1 / (REAL)255) continue;
                    T = T / ((REAL)1 - alpha);
                    REAL dchannel_dcolor = alpha * T;
                    REAL dL_dalpha = 0;
                    for (int ch = 0; ch < 3; ch++) {
                        REAL c = rgb[3 * g + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + ((REAL)1 - last_alpha) * accum_rec[ch];
                        last_color[ch] = c;
                        dL_dalpha += (c - accum_rec[ch]) * dpx[ch];
                        A[6 + ch] += (double)(dchannel_dcolor * dpx[ch]);
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final / ((REAL)1 - alpha)) * bg_dot;
                    REAL dL_dG = co[3] * dL_dalpha;
                    REAL gdx = G * dx, gdy = G * dy;
                    REAL dG_ddelx = -gdx * co[0] - gdy * co[1];
                    REAL dG_ddely = -gdy * co[2] - gdx * co[1];
                    A[0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
                    A[1] += (double)(dL_dG * dG_ddely * ddely_dy);
                    A[2] += (double)((REAL)-0.5 * gdx * dx * dL_dG);
                    A[3] += (double)((REAL)-0.5 * gdx * dy * dL_dG);
                    A[4] += (double)((REAL)-0.5 * gdy * dy * dL_dG);
                    A[5] += (double)(G * dL_dalpha);
                }
            }
    }
    double *acc = (double *)calloc((size_t)(P ? P : 1) * 9, sizeof(double));
    for (size_t j = 0; j < Rtot; j++) {
        const double *A = inst + j * 9;
        double *D = acc + (size_t)ids_sorted[j] * 9;
        for (int k = 0; k < 9; k++) D[k] += A[k];
    }
#pragma omp parallel for schedule(static)
    for (int g = 0; g < P; g++) {
        const double *s9 = acc + (size_t)g * 9;
        dL_dmean2D[3 * g + 0] = (REAL)s9[0];
        dL_dmean2D[3 * g + 1] = (REAL)s9[1];
        dL_dconic[4 * g + 0] = (REAL)s9[2];
        dL_dconic[4 * g + 1] = (REAL)s9[3];
        dL_dconic[4 * g + 3] = (REAL)s9[4];
        dL_dopacity[g] = (REAL)s9[5];
        for (int ch = 0; ch < 3; ch++) dL_dcolor[3 * g + ch] = (REAL)s9[6 + ch];
    }
    free(acc);
    free(inst);
}

/* ------------------------------------------------------------------------------------------
 * K8: per-Gaussian backward: conic -> cov2D -> (cov3D, view-space mean) ; mean2D(NDC) -> mean3D ;
 * colour -> SH (+ direction term into mean3D) ; cov3D -> (scale, quaternion).
 * Outputs are fully overwritten.  dL_dcov3D is always produced; dL_dscale/dL_drot only when
 * scales/rotations were used (cov3D_precomp == NULL); dL_dsh only when shs were used.
 * ------------------------------------------------------------------------------------------ */
void oracle_preprocess_bwd(int P, int D, int M, int W, int H, const REAL *means3D, const REAL *shs,
                           const uint8_t *clamped, const REAL *scales, REAL scale_mod, const REAL *rotations,
                           const REAL *cov3D /* as saved by forward */, int use_precomp_cov, const REAL *view,
                           const REAL *proj, const REAL *campos, REAL tanfovx, REAL tanfovy, const int32_t *radii,
                           const REAL *dL_dmean2D /* [P][3] NDC */, const REAL *dL_dconic /* [P][4] */,
                           const REAL *dL_dcolor /* [P][3] */, REAL *dL_dmean3D /* [P][3] */,
                           REAL *dL_dcov3D /* [P][6] */, REAL *dL_dsh /* [P][M][3] or NULL */,
                           REAL *dL_dscale /* [P][3] or NULL */, REAL *dL_drot /* [P][4] or NULL */) {
    const REAL fx = (REAL)W / ((REAL)2 * tanfovx), fy = (REAL)H / ((REAL)2 * tanfovy);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        for (int k = 0; k < 3; k++) dL_dmean3D[3 * i + k] = 0;
        for (int k = 0; k < 6; k++) dL_dcov3D[6 * i + k] = 0;
        if (dL_dsh) for (int k = 0; k < M * 3; k++) dL_dsh[(size_t)i * M * 3 + k] = 0;
        if (dL_dscale) for (int k = 0; k < 3; k++) dL_dscale[3 * i + k] = 0;
        if (dL_drot) for (int k = 0; k < 4; k++) dL_drot[4 * i + k] = 0;
        if (radii[i] <= 0) continue;

        const REAL *p = means3D + 3 * i;
        REAL dmean[3] = {0, 0, 0};

        /* ---- cov2D backward (upstream computeCov2DCUDA) ---- */
        REAL pv[3];
        view_point(p, view, pv);
        proj_jac_t pj;
        proj_jacobian(pv, view, fx, fy, tanfovx, tanfovy, &pj);
        const REAL *c6 = cov3D + 6 * i;
        REAL a, b, c;
        cov2d_from_cov3d(c6, &pj, &a, &b, &c);
        REAL denom = a * c - b * b;
        REAL denom2inv = (REAL)1 / ((denom * denom) + (REAL)0.0000001);
        REAL gcx = dL_dconic[4 * i + 0], gcy = dL_dconic[4 * i + 1], gcz = dL_dconic[4 * i + 3];
        REAL dL_da = 0, dL_db = 0, dL_dc = 0;
        const REAL *t0 = pj.t0, *t1 = pj.t1;
        if (denom2inv != (REAL)0) {
            dL_da = denom2inv * (-c * c * gcx + (REAL)2 * b * c * gcy + (denom - a * c) * gcz);
            dL_dc = denom2inv * (-a * a * gcz + (REAL)2 * a * b * gcy + (denom - a * c) * gcx);
            dL_db = denom2inv * (REAL)2 * (b * c * gcx - (denom + (REAL)2 * b * b) * gcy + a * b * gcz);
            REAL *g6 = dL_dcov3D + 6 * i;
            g6[0] = t0[0] * t0[0] * dL_da + t0[0] * t1[0] * dL_db + t1[0] * t1[0] * dL_dc;
            g6[3] = t0[1] * t0[1] * dL_da + t0[1] * t1[1] * dL_db + t1[1] * t1[1] * dL_dc;
            g6[5] = t0[2] * t0[2] * dL_da + t0[2] * t1[2] * dL_db + t1[2] * t1[2] * dL_dc;
            g6[1] = (REAL)2 * t0[0] * t0[1] * dL_da + (t0[0] * t1[1] + t0[1] * t1[0]) * dL_db +
                    (REAL)2 * t1[0] * t1[1] * dL_dc;
            g6[2] = (REAL)2 * t0[0] * t0[2] * dL_da + (t0[0] * t1[2] + t0[2] * t1[0]) * dL_db +
                    (REAL)2 * t1[0] * t1[2] * dL_dc;
            g6[4] = (REAL)2 * t0[2] * t0[1] * dL_da + (t0[1] * t1[2] + t0[2] * t1[1]) * dL_db +
                    (REAL)2 * t1[1] * t1[2] * dL_dc;
        }
        {
            REAL Vm[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
            REAL Vt0[3], Vt1[3], dT0[3], dT1[3];
            for (int r = 0; r < 3; r++) {
                Vt0[r] = Vm[r][0] * t0[0] + Vm[r][1] * t0[1] + Vm[r][2] * t0[2];
                Vt1[r] = Vm[r][0] * t1[0] + Vm[r][1] * t1[1] + Vm[r][2] * t1[2];
            }
            for (int r = 0; r < 3; r++) {
                dT0[r] = (REAL)2 * Vt0[r] * dL_da + Vt1[r] * dL_db;
                dT1[r] = (REAL)2 * Vt1[r] * dL_dc + Vt0[r] * dL_db;
            }
            /* dL/dJ[i][k] = sum_a Rw[k][a] dT_i[a], Rw[k][a] = view[4a+k] */
            REAL dJ00 = view[0] * dT0[0] + view[4] * dT0[1] + view[8] * dT0[2];
            REAL dJ02 = view[2] * dT0[0] + view[6] * dT0[1] + view[10] * dT0[2];
            REAL dJ11 = view[1] * dT1[0] + view[5] * dT1[1] + view[9] * dT1[2];
            REAL dJ12 = view[2] * dT1[0] + view[6] * dT1[1] + view[10] * dT1[2];
            REAL tz = (REAL)1 / pj.tz, tz2 = tz * tz, tz3 = tz2 * tz;
            REAL xg = pj.x_in ? (REAL)1 : (REAL)0, yg = pj.y_in ? (REAL)1 : (REAL)0;
            REAL dtx = xg * -fx * tz2 * dJ02;
            REAL dty = yg * -fy * tz2 * dJ12;
            REAL dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + ((REAL)2 * fx * pj.tx) * tz3 * dJ02 +
                       ((REAL)2 * fy * pj.ty) * tz3 * dJ12;
            dmean[0] += view[0] * dtx + view[1] * dty + view[2] * dtz;
            dmean[1] += view[4] * dtx + view[5] * dty + view[6] * dtz;
            dmean[2] += view[8] * dtx + view[9] * dty + view[10] * dtz;
        }

        /* ---- mean2D (NDC) -> mean3D through the perspective divide ---- */
        {
            REAL hw = proj[3] * p[0] + proj[7] * p[1] + proj[11] * p[2] + proj[15];
            REAL m_w = (REAL)1 / (hw + (REAL)0.0000001);
            REAL mul1 = (proj[0] * p[0] + proj[4] * p[1] + proj[8] * p[2] + proj[12]) * m_w * m_w;
            REAL mul2 = (proj[1] * p[0] + proj[5] * p[1] + proj[9] * p[2] + proj[13]) * m_w * m_w;
            REAL gx2 = dL_dmean2D[3 * i + 0], gy2 = dL_dmean2D[3 * i + 1];
            dmean[0] += (proj[0] * m_w - proj[3] * mul1) * gx2 + (proj[1] * m_w - proj[3] * mul2) * gy2;
            dmean[1] += (proj[4] * m_w - proj[7] * mul1) * gx2 + (proj[5] * m_w - proj[7] * mul2) * gy2;
            dmean[2] += (proj[8] * m_w - proj[11] * mul1) * gx2 + (proj[9] * m_w - proj[11] * mul2) * gy2;
        }

        /* ---- colour -> SH, and direction -> mean3D ---- */
        if (shs && dL_dsh) {
            const REAL *sh = shs + (size_t)i * M * 3;
            REAL *gsh = dL_dsh + (size_t)i * M * 3;
            REAL dir0[3] = {p[0] - campos[0], p[1] - campos[1], p[2] - campos[2]};
            REAL len = rsqrt_(dir0[0] * dir0[0] + dir0[1] * dir0[1] + dir0[2] * dir0[2]);
            REAL x = dir0[0] / len, y = dir0[1] / len, z = dir0[2] / len;
            REAL dRGB[3];
            for (int ch = 0; ch < 3; ch++) dRGB[ch] = clamped[3 * i + ch] ? (REAL)0 : dL_dcolor[3 * i + ch];
            REAL dRGBdx[3] = {0, 0, 0}, dRGBdy[3] = {0, 0, 0}, dRGBdz[3] = {0, 0, 0};
#define S(k) sh[(k) * 3 + ch]
#define GS(k) gsh[(k) * 3 + ch]
            for (int ch = 0; ch < 3; ch++) {
                GS(0) = SH_C0 * dRGB[ch];
                if (D > 0) {
                    GS(1) = -SH_C1 * y * dRGB[ch];
                    GS(2) = SH_C1 * z * dRGB[ch];
                    GS(3) = -SH_C1 * x * dRGB[ch];
                    dRGBdx[ch] = -SH_C1 * S(3);
                    dRGBdy[ch] = -SH_C1 * S(1);
                    dRGBdz[ch] = SH_C1 * S(2);
                    if (D > 1) {
                        REAL xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        GS(4) = SH_C2[0] * xy * dRGB[ch];
                        GS(5) = SH_C2[1] * yz * dRGB[ch];
                        GS(6) = SH_C2[2] * ((REAL)2 * zz - xx - yy) * dRGB[ch];
                        GS(7) = SH_C2[3] * xz * dRGB[ch];
                        GS(8) = SH_C2[4] * (xx - yy) * dRGB[ch];
                        dRGBdx[ch] += SH_C2[0] * y * S(4) + SH_C2[2] * (REAL)2 * -x * S(6) + SH_C2[3] * z * S(7) +
                                      SH_C2[4] * (REAL)2 * x * S(8);
                        dRGBdy[ch] += SH_C2[0] * x * S(4) + SH_C2[1] * z * S(5) + SH_C2[2] * (REAL)2 * -y * S(6) +
                                      SH_C2[4] * (REAL)2 * -y * S(8);
                        dRGBdz[ch] += SH_C2[1] * y * S(5) + SH_C2[2] * (REAL)2 * (REAL)2 * z * S(6) +
                                      SH_C2[3] * x * S(7);
                        if (D > 2) {
                            GS(9) = SH_C3[0] * y * ((REAL)3 * xx - yy) * dRGB[ch];
                            GS(10) = SH_C3[1] * xy * z * dRGB[ch];
                            GS(11) = SH_C3[2] * y * ((REAL)4 * zz - xx - yy) * dRGB[ch];
                            GS(12) = SH_C3[3] * z * ((REAL)2 * zz - (REAL)3 * xx - (REAL)3 * yy) * dRGB[ch];
                            GS(13) = SH_C3[4] * x * ((REAL)4 * zz - xx - yy) * dRGB[ch];
                            GS(14) = SH_C3[5] * z * (xx - yy) * dRGB[ch];
                            GS(15) = SH_C3[6] * x * (xx - (REAL)3 * yy) * dRGB[ch];
                            dRGBdx[ch] += SH_C3[0] * S(9) * (REAL)3 * (REAL)2 * xy + SH_C3[1] * S(10) * yz +
                                          SH_C3[2] * S(11) * -(REAL)2 * xy +
                                          SH_C3[3] * S(12) * -(REAL)3 * (REAL)2 * xz +
                                          SH_C3[4] * S(13) * (-(REAL)3 * xx + (REAL)4 * zz - yy) +
                                          SH_C3[5] * S(14) * (REAL)2 * xz +
                                          SH_C3[6] * S(15) * (REAL)3 * (xx - yy);
                            dRGBdy[ch] += SH_C3[0] * S(9) * (REAL)3 * (xx - yy) + SH_C3[1] * S(10) * xz +
                                          SH_C3[2] * S(11) * (-(REAL)3 * yy + (REAL)4 * zz - xx) +
                                          SH_C3[3] * S(12) * -(REAL)3 * (REAL)2 * yz +
                                          SH_C3[4] * S(13) * -(REAL)2 * xy +
                                          SH_C3[5] * S(14) * -(REAL)2 * yz +
                                          SH_C3[6] * S(15) * -(REAL)3 * (REAL)2 * xy;
                            dRGBdz[ch] += SH_C3[1] * S(10) * xy + SH_C3[2] * S(11) * (REAL)4 * (REAL)2 * yz +
                                          SH_C3[3] * S(12) * (REAL)3 * ((REAL)2 * zz - xx - yy) +
                                          SH_C3[4] * S(13) * (REAL)4 * (REAL)2 * xz +
                                          SH_C3[5] * S(14) * (xx - yy);
                        }
                    }
                }
            }
#undef S
#undef GS
            REAL ddir[3] = {dRGBdx[0] * dRGB[0] + dRGBdx[1] * dRGB[1] + dRGBdx[2] * dRGB[2],
                            dRGBdy[0] * dRGB[0] + dRGBdy[1] * dRGB[1] + dRGBdy[2] * dRGB[2],
                            dRGBdz[0] * dRGB[0] + dRGBdz[1] * dRGB[1] + dRGBdz[2] * dRGB[2]};
            /* d normalize(v)/dv applied to ddir */
            REAL sum2 = dir0[0] * dir0[0] + dir0[1] * dir0[1] + dir0[2] * dir0[2];
            REAL invsum32 = (REAL)1 / rsqrt_(sum2 * sum2 * sum2);
            REAL vx = dir0[0], vy = dir0[1], vz = dir0[2];
            dmean[0] += ((+sum2 - vx * vx) * ddir[0] - vy * vx * ddir[1] - vz * vx * ddir[2]) * invsum32;
            dmean[1] += (-vx * vy * ddir[0] + (sum2 - vy * vy) * ddir[1] - vz * vy * ddir[2]) * invsum32;
            dmean[2] += (-vx * vz * ddir[0] - vy * vz * ddir[1] + (sum2 - vz * vz) * ddir[2]) * invsum32;
        }

        for (int k = 0; k < 3; k++) dL_dmean3D[3 * i + k] = dmean[k];

        /* ---- cov3D -> scale, quaternion ---- */
        if (!use_precomp_cov && dL_dscale && dL_drot) {
            const REAL *q = rotations + 4 * i;
            REAL R[3][3];
            quat_to_rot(q, R);
            REAL s[3] = {scale_mod * scales[3 * i], scale_mod * scales[3 * i + 1], scale_mod * scales[3 * i + 2]};
            const REAL *g6 = dL_dcov3D + 6 * i;
            REAL dS[3][3] = {{g6[0], (REAL)0.5 * g6[1], (REAL)0.5 * g6[2]},
                             {(REAL)0.5 * g6[1], g6[3], (REAL)0.5 * g6[4]},
                             {(REAL)0.5 * g6[2], (REAL)0.5 * g6[4], g6[5]}};
            /* Sigma = A A^T with A[i][k] = R[i][k]*s_k ;  dL/dA = 2 dS A */
            REAL dA[3][3];
            for (int r = 0; r < 3; r++)
                for (int k = 0; k < 3; k++)
                    dA[r][k] = (REAL)2 * (dS[r][0] * R[0][k] * s[k] + dS[r][1] * R[1][k] * s[k] + dS[r][2] * R[2][k] * s[k]);
            /* dL/d(mod*scale_k) = sum_r dA[r][k] R[r][k]   (upstream omits the modifier factor) */
            for (int k = 0; k < 3; k++)
                dL_dscale[3 * i + k] = dA[0][k] * R[0][k] + dA[1][k] * R[1][k] + dA[2][k] * R[2][k];
            /* dL/dR[r][k] = dA[r][k] * s_k */
            REAL dR[3][3];
            for (int r = 0; r < 3; r++)
                for (int k = 0; k < 3; k++) dR[r][k] = dA[r][k] * s[k];
            REAL qr = q[0], qx = q[1], qy = q[2], qz = q[3];
            dL_drot[4 * i + 0] = (REAL)2 * (-qz * dR[0][1] + qy * dR[0][2] + qz * dR[1][0] - qx * dR[1][2] -
                                            qy * dR[2][0] + qx * dR[2][1]);
            dL_drot[4 * i + 1] = (REAL)2 * (qy * dR[0][1] + qz * dR[0][2] + qy * dR[1][0] - (REAL)2 * qx * dR[1][1] -
                                            qr * dR[1][2] + qz * dR[2][0] + qr * dR[2][1] - (REAL)2 * qx * dR[2][2]);
            dL_drot[4 * i + 2] = (REAL)2 * (-(REAL)2 * qy * dR[0][0] + qx * dR[0][1] + qr * dR[0][2] + qx * dR[1][0] +
                                            qz * dR[1][2] - qr * dR[2][0] + qz * dR[2][1] - (REAL)2 * qy * dR[2][2]);
            dL_drot[4 * i + 3] = (REAL)2 * (-(REAL)2 * qz * dR[0][0] - qr * dR[0][1] + qx * dR[0][2] + qr * dR[1][0] -
                                            (REAL)2 * qz * dR[1][1] + qy * dR[1][2] + qx * dR[2][0] + qy * dR[2][1]);
        }
    }
}
