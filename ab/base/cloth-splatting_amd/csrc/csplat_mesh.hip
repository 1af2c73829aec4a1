// csplat_mesh.hip -- fused mesh -> Gaussian transform (SURVEY.md 8(f) "next" row N1), forward and backward.
//
// Replaces, per render() call, MultiGaussianMesh.get_xyz + get_rotation
// (/root/reference/scene_reconstruction/gaussian_mesh.py:151-188): gather the 3 vertices of every Gaussian's face,
// barycentric centre, Kabsch rotation of the rest face onto the deformed face (roma.rigid_points_registration ->
// batched 3x3 SVD in the reference), rotation matrix -> unit quaternion (roma, XYZW), composition with the Gaussian's
// own normalised rotation (roma.quat_composition; the reference feeds a WXYZ-initialised parameter through roma's XYZW
// product -- SURVEY F8 -- reproduced as is).  As ~80 torch ops this costs ~500 tiny kernels per camera (forward +
// autograd) plus two sort-based index_put backward passes; here it is ONE kernel each way.
//
// The Kabsch solution for 3 points is closed form (no SVD): see csplat/rotations.py::kabsch_triangles, which the tests
// pin against the SVD formulation.  The backward is the hand-written adjoint of the same sequence of steps
// (transform_one_bwd, step numbers matching transform_one): ~1/14 of the instructions of the forward-mode sweep over the 16
// inputs it replaced (7800 -> ~550 per Gaussian) and a third of its registers.  Vertex gradients are scattered with float
// atomics.  Both directions take the T cameras of a training step in one launch (vertices [T][V][3]).
#include "csplat_common.h"

namespace {

__device__ __forceinline__ float rsqrt_(float a) { return 1.f / sqrtf(a); }
__device__ __forceinline__ float val(float a) { return a; }
template <typename T> __device__ __forceinline__ T lift(float v);
template <> __device__ __forceinline__ float lift<float>(float v) { return v; }

// rest-face constants of one Gaussian (independent of the deformation): orthonormal in-plane basis + normal of the
// centred rest triangle and the in-plane coordinates of its 3 points
struct RestFace {
    float ux[3], vx[3], nx[3], xu[3], xv[3];
};

template <typename T>
__device__ __forceinline__ void plane_basis(const T p0[3], const T p1[3], T u[3], T v[3], T n[3]) {
    T inv = rsqrt_(p0[0] * p0[0] + p0[1] * p0[1] + p0[2] * p0[2]);
    for (int k = 0; k < 3; k++) u[k] = p0[k] * inv;
    T dt = p1[0] * u[0] + p1[1] * u[1] + p1[2] * u[2];
    T t[3];
    for (int k = 0; k < 3; k++) t[k] = p1[k] - dt * u[k];
    inv = rsqrt_(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    for (int k = 0; k < 3; k++) v[k] = t[k] * inv;
    n[0] = u[1] * v[2] - u[2] * v[1];
    n[1] = u[2] * v[0] - u[0] * v[2];
    n[2] = u[0] * v[1] - u[1] * v[0];
}

// y[3][3] deformed face vertices, bary[3], q0[4] raw rotation parameter -> pos[3], quat[4]
template <typename T>
__device__ __forceinline__ void transform_one(const T y[3][3], const T bary[3], const T q0[4], const RestFace &rf, T pos[3],
                                              T quat[4]) {
    // ---- barycentric centre (gaussian_mesh.py:166-168)
    const T bs = bary[0] + bary[1] + bary[2];
    for (int c = 0; c < 3; c++) pos[c] = (bary[0] * y[0][c] + bary[1] * y[1][c] + bary[2] * y[2][c]) / bs;
    // ---- closed-form Kabsch of the rest triangle onto the deformed one
    T yh[3][3];
    for (int c = 0; c < 3; c++) {
        const T m = (y[0][c] + y[1][c] + y[2][c]) * (1.f / 3.f);
        for (int k = 0; k < 3; k++) yh[k][c] = y[k][c] - m;
    }
    T uy[3], vy[3], ny[3];
    plane_basis(yh[0], yh[1], uy, vy, ny);
    T a = lift<T>(0.f), b = lift<T>(0.f), c_ = lift<T>(0.f), d = lift<T>(0.f);
    for (int k = 0; k < 3; k++) {
        const T yu = yh[k][0] * uy[0] + yh[k][1] * uy[1] + yh[k][2] * uy[2];
        const T yv = yh[k][0] * vy[0] + yh[k][1] * vy[1] + yh[k][2] * vy[2];
        a = a + yu * rf.xu[k]; b = b + yu * rf.xv[k];
        c_ = c_ + yv * rf.xu[k]; d = d + yv * rf.xv[k];
    }
    const bool pos_det = val(a) * val(d) - val(b) * val(c_) > 0.f;
    T q00 = pos_det ? a + d : a - d;
    T q01 = pos_det ? b - c_ : b + c_;
    const T nrm = rsqrt_(q00 * q00 + q01 * q01);
    q00 = q00 * nrm; q01 = q01 * nrm;
    const T q10 = pos_det ? -q01 : q01;
    const T q11 = pos_det ? q00 : -q00;
    const float sgn = pos_det ? 1.f : -1.f;
    T R[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R[i][j] = q00 * (uy[i] * rf.ux[j]) + q01 * (uy[i] * rf.vx[j]) + q10 * (vy[i] * rf.ux[j]) + q11 * (vy[i] * rf.vx[j]) +
                      ny[i] * (sgn * rf.nx[j]);
    // ---- rotation matrix -> unit quaternion XYZW (roma.rotmat_to_unitquat: largest of (diagonal, trace))
    const T tr = R[0][0] + R[1][1] + R[2][2];
    int choice = 0;   // argmax over [R00, R11, R22, trace], first maximum wins (torch.argmax)
    float best = val(R[0][0]);
    if (val(R[1][1]) > best) { best = val(R[1][1]); choice = 1; }
    if (val(R[2][2]) > best) { best = val(R[2][2]); choice = 2; }
    if (val(tr) > best) { best = val(tr); choice = 3; }
    T qr[4];
    // (static indices in every branch: runtime-indexed register arrays would go to scratch)
#define QBRANCH(I, J, K)                     \
    {                                        \
        qr[I] = (1.f - tr) + 2.f * R[I][I];  \
        qr[J] = R[J][I] + R[I][J];           \
        qr[K] = R[K][I] + R[I][K];           \
        qr[3] = R[K][J] - R[J][K];           \
    }
    if (choice == 3) {
        qr[0] = R[2][1] - R[1][2]; qr[1] = R[0][2] - R[2][0]; qr[2] = R[1][0] - R[0][1]; qr[3] = tr + 1.f;
    } else if (choice == 0) QBRANCH(0, 1, 2)
    else if (choice == 1) QBRANCH(1, 2, 0)
    else QBRANCH(2, 0, 1)
#undef QBRANCH
    T inv = rsqrt_(qr[0] * qr[0] + qr[1] * qr[1] + qr[2] * qr[2] + qr[3] * qr[3]);
    for (int i = 0; i < 4; i++) qr[i] = qr[i] * inv;
    // ---- own rotation: F.normalize(_rotation), then roma.quat_composition([rotation, relative]) in XYZW convention
    inv = rsqrt_(q0[0] * q0[0] + q0[1] * q0[1] + q0[2] * q0[2] + q0[3] * q0[3]);
    T p[4];
    for (int i = 0; i < 4; i++) p[i] = q0[i] * inv;
    // Hamilton product p * qr, vector part first (x, y, z), scalar last (w)
    quat[0] = p[3] * qr[0] + qr[3] * p[0] + (p[1] * qr[2] - p[2] * qr[1]);
    quat[1] = p[3] * qr[1] + qr[3] * p[1] + (p[2] * qr[0] - p[0] * qr[2]);
    quat[2] = p[3] * qr[2] + qr[3] * p[2] + (p[0] * qr[1] - p[1] * qr[0]);
    quat[3] = p[3] * qr[3] - (p[0] * qr[0] + p[1] * qr[1] + p[2] * qr[2]);
}

__global__ __launch_bounds__(256) void k_rest_faces(int P, const int64_t *__restrict__ vid, const float *__restrict__ rest,
                                                     RestFace *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float x[3][3], xh[3][3];
    for (int k = 0; k < 3; k++)
        for (int c = 0; c < 3; c++) x[k][c] = rest[3 * vid[3 * (size_t)i + k] + c];
    for (int c = 0; c < 3; c++) {
        const float m = (x[0][c] + x[1][c] + x[2][c]) * (1.f / 3.f);
        for (int k = 0; k < 3; k++) xh[k][c] = x[k][c] - m;
    }
    RestFace rf;
    plane_basis(xh[0], xh[1], rf.ux, rf.vx, rf.nx);
    for (int k = 0; k < 3; k++) {
        rf.xu[k] = xh[k][0] * rf.ux[0] + xh[k][1] * rf.ux[1] + xh[k][2] * rf.ux[2];
        rf.xv[k] = xh[k][0] * rf.vx[0] + xh[k][1] * rf.vx[1] + xh[k][2] * rf.vx[2];
    }
    out[i] = rf;
}

// blockIdx.y = camera t of the step: vertices [T][V][3] -> out_pos [T][P][3], out_quat [T][P][4]
__global__ __launch_bounds__(256) void k_mesh_fwd(int P, int V, const int64_t *__restrict__ vid, const float *__restrict__ verts_all,
                                                   const float *__restrict__ bary, const float *__restrict__ rot,
                                                   const RestFace *__restrict__ rest, float *__restrict__ out_pos,
                                                   float *__restrict__ out_quat) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float *verts = verts_all + (size_t)blockIdx.y * V * 3;
    const size_t o = (size_t)blockIdx.y * P + i;
    float y[3][3], b[3], q0[4], pos[3], quat[4];
    for (int k = 0; k < 3; k++)
        for (int c = 0; c < 3; c++) y[k][c] = verts[3 * vid[3 * (size_t)i + k] + c];
    for (int k = 0; k < 3; k++) b[k] = bary[3 * (size_t)i + k];
    for (int k = 0; k < 4; k++) q0[k] = rot[4 * (size_t)i + k];
    transform_one<float>(y, b, q0, rest[i], pos, quat);
    for (int c = 0; c < 3; c++) out_pos[3 * o + c] = pos[c];
    for (int c = 0; c < 4; c++) out_quat[4 * o + c] = quat[c];
}

__device__ __forceinline__ float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// adjoint of transform_one: (gpos, gquat) -> (gy, gb, gq0).  The forward values are recomputed (same expressions, same
// branches); the steps are undone in reverse order.
__device__ __forceinline__ void transform_one_bwd(const float y[3][3], const float bary[3], const float q0[4], const RestFace &rf,
                                                  const float gpos[3], const float gquat[4], float gy[3][3], float gb[3],
                                                  float gq0[4]) {
    // ================= forward recomputation
    const float bs = bary[0] + bary[1] + bary[2], ibs = 1.f / bs;
    float pos[3];
    for (int c = 0; c < 3; c++) pos[c] = (bary[0] * y[0][c] + bary[1] * y[1][c] + bary[2] * y[2][c]) * ibs;
    float yh[3][3];
    for (int c = 0; c < 3; c++) {
        const float m = (y[0][c] + y[1][c] + y[2][c]) * (1.f / 3.f);
        for (int k = 0; k < 3; k++) yh[k][c] = y[k][c] - m;
    }
    // plane_basis(yh[0], yh[1]) with its intermediates kept
    const float inv0 = rsqrt_(dot3(yh[0], yh[0]));
    float u[3], t[3], v[3], n[3];
    for (int k = 0; k < 3; k++) u[k] = yh[0][k] * inv0;
    const float dt = dot3(yh[1], u);
    for (int k = 0; k < 3; k++) t[k] = yh[1][k] - dt * u[k];
    const float inv1 = rsqrt_(dot3(t, t));
    for (int k = 0; k < 3; k++) v[k] = t[k] * inv1;
    n[0] = u[1] * v[2] - u[2] * v[1]; n[1] = u[2] * v[0] - u[0] * v[2]; n[2] = u[0] * v[1] - u[1] * v[0];
    float a = 0.f, b = 0.f, c_ = 0.f, d = 0.f;
    for (int k = 0; k < 3; k++) {
        const float yu = dot3(yh[k], u), yv = dot3(yh[k], v);
        a += yu * rf.xu[k]; b += yu * rf.xv[k]; c_ += yv * rf.xu[k]; d += yv * rf.xv[k];
    }
    const bool pos_det = a * d - b * c_ > 0.f;
    const float sgn = pos_det ? 1.f : -1.f;
    const float r00 = a + sgn * d, r01 = b - sgn * c_;          // raw (q00, q01)
    const float nrm = rsqrt_(r00 * r00 + r01 * r01);
    const float q00 = r00 * nrm, q01 = r01 * nrm;
    const float q10 = -sgn * q01, q11 = sgn * q00;
    float R[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R[i][j] = q00 * (u[i] * rf.ux[j]) + q01 * (u[i] * rf.vx[j]) + q10 * (v[i] * rf.ux[j]) + q11 * (v[i] * rf.vx[j]) +
                      n[i] * (sgn * rf.nx[j]);
    const float tr = R[0][0] + R[1][1] + R[2][2];
    int choice = 0;
    float best = R[0][0];
    if (R[1][1] > best) { best = R[1][1]; choice = 1; }
    if (R[2][2] > best) { best = R[2][2]; choice = 2; }
    if (tr > best) { best = tr; choice = 3; }
    float qr[4];
#define QBRANCH(I, J, K)                     \
    {                                        \
        qr[I] = (1.f - tr) + 2.f * R[I][I];  \
        qr[J] = R[J][I] + R[I][J];           \
        qr[K] = R[K][I] + R[I][K];           \
        qr[3] = R[K][J] - R[J][K];           \
    }
    if (choice == 3) {
        qr[0] = R[2][1] - R[1][2]; qr[1] = R[0][2] - R[2][0]; qr[2] = R[1][0] - R[0][1]; qr[3] = tr + 1.f;
    } else if (choice == 0) QBRANCH(0, 1, 2)
    else if (choice == 1) QBRANCH(1, 2, 0)
    else QBRANCH(2, 0, 1)
#undef QBRANCH
    const float invq = rsqrt_(qr[0] * qr[0] + qr[1] * qr[1] + qr[2] * qr[2] + qr[3] * qr[3]);
    for (int i = 0; i < 4; i++) qr[i] *= invq;
    const float invp = rsqrt_(q0[0] * q0[0] + q0[1] * q0[1] + q0[2] * q0[2] + q0[3] * q0[3]);
    float p[4];
    for (int i = 0; i < 4; i++) p[i] = q0[i] * invp;

    // ================= adjoint, last step first
    // 9. quat = Hamilton(p, qr) (bilinear)
    const float *g = gquat;
    float gp[4], gqr[4];
    gp[0] = g[0] * qr[3] - g[1] * qr[2] + g[2] * qr[1] - g[3] * qr[0];
    gp[1] = g[0] * qr[2] + g[1] * qr[3] - g[2] * qr[0] - g[3] * qr[1];
    gp[2] = -g[0] * qr[1] + g[1] * qr[0] + g[2] * qr[3] - g[3] * qr[2];
    gp[3] = g[0] * qr[0] + g[1] * qr[1] + g[2] * qr[2] + g[3] * qr[3];
    gqr[0] = g[0] * p[3] + g[1] * p[2] - g[2] * p[1] - g[3] * p[0];
    gqr[1] = -g[0] * p[2] + g[1] * p[3] + g[2] * p[0] - g[3] * p[1];
    gqr[2] = g[0] * p[1] - g[1] * p[0] + g[2] * p[3] - g[3] * p[2];
    gqr[3] = g[0] * p[0] + g[1] * p[1] + g[2] * p[2] + g[3] * p[3];
    // 8. p = q0 / |q0|
    {
        const float pg = p[0] * gp[0] + p[1] * gp[1] + p[2] * gp[2] + p[3] * gp[3];
        for (int i = 0; i < 4; i++) gq0[i] = invp * (gp[i] - p[i] * pg);
    }
    // 7. qr = raw / |raw|, raw linear in R (by branch)
    float graw[4];
    {
        const float qg = qr[0] * gqr[0] + qr[1] * gqr[1] + qr[2] * gqr[2] + qr[3] * gqr[3];
        for (int i = 0; i < 4; i++) graw[i] = invq * (gqr[i] - qr[i] * qg);
    }
    float gR[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) gR[i][j] = 0.f;
#define QBRANCH_BWD(I, J, K)                                                        \
    {                                                                               \
        gR[0][0] -= graw[I]; gR[1][1] -= graw[I]; gR[2][2] -= graw[I];              \
        gR[I][I] += 2.f * graw[I];                                                  \
        gR[J][I] += graw[J]; gR[I][J] += graw[J];                                   \
        gR[K][I] += graw[K]; gR[I][K] += graw[K];                                   \
        gR[K][J] += graw[3]; gR[J][K] -= graw[3];                                   \
    }
    if (choice == 3) {
        gR[2][1] += graw[0]; gR[1][2] -= graw[0];
        gR[0][2] += graw[1]; gR[2][0] -= graw[1];
        gR[1][0] += graw[2]; gR[0][1] -= graw[2];
        gR[0][0] += graw[3]; gR[1][1] += graw[3]; gR[2][2] += graw[3];
    } else if (choice == 0) QBRANCH_BWD(0, 1, 2)
    else if (choice == 1) QBRANCH_BWD(1, 2, 0)
    else QBRANCH_BWD(2, 0, 1)
#undef QBRANCH_BWD
    // 6. R = q00 u ux^T + q01 u vx^T + q10 v ux^T + q11 v vx^T + sgn n nx^T
    float A[3], B[3], gn[3], gu[3], gv[3];
    for (int i = 0; i < 3; i++) {
        A[i] = gR[i][0] * rf.ux[0] + gR[i][1] * rf.ux[1] + gR[i][2] * rf.ux[2];
        B[i] = gR[i][0] * rf.vx[0] + gR[i][1] * rf.vx[1] + gR[i][2] * rf.vx[2];
        gn[i] = sgn * (gR[i][0] * rf.nx[0] + gR[i][1] * rf.nx[1] + gR[i][2] * rf.nx[2]);
    }
    const float gq00 = dot3(u, A), gq01 = dot3(u, B), gq10 = dot3(v, A), gq11 = dot3(v, B);
    for (int i = 0; i < 3; i++) {
        gu[i] = q00 * A[i] + q01 * B[i];
        gv[i] = q10 * A[i] + q11 * B[i];
    }
    // 5. q10 = -sgn q01, q11 = sgn q00; (q00, q01) = (r00, r01) / |(r00, r01)|; r00 = a + sgn d, r01 = b - sgn c
    const float gn00 = gq00 + sgn * gq11, gn01 = gq01 - sgn * gq10;
    const float qg2 = q00 * gn00 + q01 * gn01;
    const float gr00 = nrm * (gn00 - q00 * qg2), gr01 = nrm * (gn01 - q01 * qg2);
    const float ga = gr00, gd = sgn * gr00, gbb = gr01, gc = -sgn * gr01;
    // 4. a, b, c, d from yu_k = yh_k . u, yv_k = yh_k . v
    float gyh[3][3];
    for (int k = 0; k < 3; k++) {
        const float gyu = ga * rf.xu[k] + gbb * rf.xv[k], gyv = gc * rf.xu[k] + gd * rf.xv[k];
        for (int c = 0; c < 3; c++) {
            gyh[k][c] = gyu * u[c] + gyv * v[c];
            gu[c] += gyu * yh[k][c];
            gv[c] += gyv * yh[k][c];
        }
    }
    // 3. plane basis: n = u x v; v = t / |t|; t = p1 - dt u; dt = p1 . u; u = p0 / |p0|
    gu[0] += v[1] * gn[2] - v[2] * gn[1]; gu[1] += v[2] * gn[0] - v[0] * gn[2]; gu[2] += v[0] * gn[1] - v[1] * gn[0];
    gv[0] += gn[1] * u[2] - gn[2] * u[1]; gv[1] += gn[2] * u[0] - gn[0] * u[2]; gv[2] += gn[0] * u[1] - gn[1] * u[0];
    float gt[3];
    {
        const float vg = dot3(v, gv);
        for (int k = 0; k < 3; k++) gt[k] = inv1 * (gv[k] - v[k] * vg);
    }
    const float gdt = -dot3(gt, u);
    for (int k = 0; k < 3; k++) {
        gyh[1][k] += gt[k] + gdt * u[k];
        gu[k] += -dt * gt[k] + gdt * yh[1][k];
    }
    {
        const float ug = dot3(u, gu);
        for (int k = 0; k < 3; k++) gyh[0][k] += inv0 * (gu[k] - u[k] * ug);
    }
    // 2. yh = y - mean;  1. pos = sum_k bary_k y_k / bs
    for (int c = 0; c < 3; c++) {
        const float m = (gyh[0][c] + gyh[1][c] + gyh[2][c]) * (1.f / 3.f);
        for (int k = 0; k < 3; k++) gy[k][c] = gyh[k][c] - m + gpos[c] * bary[k] * ibs;
    }
    for (int k = 0; k < 3; k++)
        gb[k] = (gpos[0] * (y[k][0] - pos[0]) + gpos[1] * (y[k][1] - pos[1]) + gpos[2] * (y[k][2] - pos[2])) * ibs;
}

// one thread per Gaussian, looping over the T cameras: d_bary / d_rot are summed over the cameras in registers (fixed
// order), the vertex gradients of camera t go to d_verts[t] with atomics
__global__ __launch_bounds__(256) void k_mesh_bwd(int T, int P, int V, const int64_t *__restrict__ vid,
                                                   const float *__restrict__ verts_all, const float *__restrict__ bary,
                                                   const float *__restrict__ rot, const RestFace *__restrict__ rest,
                                                   const float *__restrict__ g_pos, const float *__restrict__ g_quat,
                                                   float *__restrict__ d_verts, float *__restrict__ d_bary, float *__restrict__ d_rot,
                                                   float *__restrict__ corner_grads) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    int64_t v3[3];
    float b[3], q0[4];
    for (int k = 0; k < 3; k++) v3[k] = vid[3 * (size_t)i + k];
    for (int k = 0; k < 3; k++) b[k] = bary[3 * (size_t)i + k];
    for (int k = 0; k < 4; k++) q0[k] = rot[4 * (size_t)i + k];
    const RestFace rf = rest[i];
    float sb[3] = {0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < T; t++) {
        const float *verts = verts_all + (size_t)t * V * 3;
        const size_t o = (size_t)t * P + i;
        float y[3][3], gp[3], gq[4], gy[3][3], gb[3], gq0[4];
        for (int k = 0; k < 3; k++)
            for (int c = 0; c < 3; c++) y[k][c] = verts[3 * v3[k] + c];
        for (int c = 0; c < 3; c++) gp[c] = g_pos ? g_pos[3 * o + c] : 0.f;
        for (int c = 0; c < 4; c++) gq[c] = g_quat ? g_quat[4 * o + c] : 0.f;
        transform_one_bwd(y, b, q0, rf, gp, gq, gy, gb, gq0);
        if (corner_grads) {   // [T][P][3 corners][3]: summed per vertex by k_vertex_gather (no atomics, fixed order)
            float *o9 = corner_grads + o * 9;
            for (int k = 0; k < 3; k++)
                for (int c = 0; c < 3; c++) o9[3 * k + c] = gy[k][c];
        } else {
            float *dv = d_verts + (size_t)t * V * 3;
            for (int k = 0; k < 3; k++)
                for (int c = 0; c < 3; c++) atomicAdd(dv + 3 * v3[k] + c, gy[k][c]);
        }
        for (int k = 0; k < 3; k++) sb[k] += gb[k];
        for (int k = 0; k < 4; k++) sq[k] += gq0[k];
    }
    for (int k = 0; k < 3; k++) d_bary[3 * (size_t)i + k] = sb[k];
    for (int k = 0; k < 4; k++) d_rot[4 * (size_t)i + k] = sq[k];
}

// d_verts[t][v] = sum of the corner gradients of the (Gaussian, corner) pairs incident to vertex v, in the fixed order of the
// incidence list (corners[rowptr[v] .. rowptr[v+1]) = 3 * gaussian + corner, ascending).  100k Gaussians on a 10k-vertex mesh
// put ~30 contributions on every vertex coordinate: as float atomics that is 2.7 M serialised L2 operations per step (150 us);
// as a gather over the static incidence it is 10 MB of reads.
__global__ __launch_bounds__(256) void k_vertex_gather(int P, int V, const int *__restrict__ rowptr, const int *__restrict__ corners,
                                                        const float *__restrict__ corner_grads, float *__restrict__ d_verts) {
    // 8 lanes per vertex stride through its incidence list (one lane per vertex is a chain of ~30 dependent-latency
    // gathers); the 8 partial sums meet in a fixed butterfly
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int v = gid >> 3, sub = gid & 7;
    const float *cg = corner_grads + (size_t)blockIdx.y * P * 9;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if (v < V) {
        for (int e = rowptr[v] + sub, end = rowptr[v + 1]; e < end; e += 8) {
            const float *g = cg + 3 * (size_t)corners[e];
            s0 += g[0]; s1 += g[1]; s2 += g[2];
        }
    }
#pragma unroll
    for (int m = 4; m >= 1; m >>= 1) {
        s0 += __shfl_xor(s0, m, 64); s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64);
    }
    if (v < V && sub == 0) {
        float *o = d_verts + ((size_t)blockIdx.y * V + v) * 3;
        o[0] = s0; o[1] = s1; o[2] = s2;
    }
}

// pixel coordinates of world points: [p, 1] @ full_proj (row-vector convention), perspective divide, ndc -> pixel
// (reference gaussian_renderer/__init__.py:166-179).  `full` is the device-resident 4x4 as torch stores it (row-major).
__global__ __launch_bounds__(256) void k_project_points(int64_t n, const float *__restrict__ full, float W, float H,
                                                        const float *__restrict__ pts, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    const float hx = x * full[0] + y * full[4] + z * full[8] + full[12];
    const float hy = x * full[1] + y * full[5] + z * full[9] + full[13];
    const float hw = x * full[3] + y * full[7] + z * full[11] + full[15];
    out[2 * i] = ((hx / hw + 1.f) * W - 1.f) * 0.5f;
    out[2 * i + 1] = ((hy / hw + 1.f) * H - 1.f) * 0.5f;
}

}  // namespace

extern "C" {

size_t csplat_mesh_rest_bytes(int P) { return align256((size_t)(P > 0 ? P : 1) * sizeof(RestFace)); }

int csplat_mesh_rest(void *stream, int P, const int64_t *face_vertex_ids, const float *rest_vertices, void *rest_out) {
    CSPLAT_REQUIRE(P >= 0 && (P == 0 || (face_vertex_ids && rest_vertices && rest_out)), "csplat_mesh_rest: bad arguments");
    if (P == 0) return 0;
    k_rest_faces<<<cdiv(P, 256), 256, 0, (hipStream_t)stream>>>(P, face_vertex_ids, rest_vertices, (RestFace *)rest_out);
    LAUNCH_CHECK();
    return 0;
}

int csplat_mesh_transform_fwd_views(void *stream, int T, int P, int V, const int64_t *face_vertex_ids, const float *vertices,
                                    const float *bary, const float *rotation, const void *rest, float *out_xyz, float *out_quat) {
    CSPLAT_REQUIRE(T >= 0 && T < 65536 && P >= 0 && V >= 0, "csplat_mesh_transform_fwd_views: bad sizes");
    if (P == 0 || T == 0) return 0;
    CSPLAT_REQUIRE(face_vertex_ids && vertices && bary && rotation && rest && out_xyz && out_quat, "csplat_mesh_transform_fwd_views: NULL");
    k_mesh_fwd<<<dim3(cdiv(P, 256), T), 256, 0, (hipStream_t)stream>>>(P, V, face_vertex_ids, vertices, bary, rotation,
                                                                       (const RestFace *)rest, out_xyz, out_quat);
    LAUNCH_CHECK();
    return 0;
}

int csplat_mesh_transform_bwd_views(void *stream, int T, int P, int V, const int64_t *face_vertex_ids, const float *vertices,
                                    const float *bary, const float *rotation, const void *rest, const float *g_xyz,
                                    const float *g_quat, float *d_vertices, float *d_bary, float *d_rotation,
                                    const int *vertex_rowptr, const int *vertex_corners, float *corner_scratch) {
    CSPLAT_REQUIRE(T >= 0 && T < 65536 && P >= 0 && V >= 0, "csplat_mesh_transform_bwd_views: bad sizes");
    CSPLAT_REQUIRE((V == 0 || T == 0 || d_vertices) && (P == 0 || (d_bary && d_rotation)), "csplat_mesh_transform_bwd_views: NULL outputs");
    const bool gather = vertex_rowptr != nullptr;
    CSPLAT_REQUIRE(!gather || (vertex_corners && corner_scratch), "csplat_mesh_transform_bwd_views: incidence without corners / scratch");
    hipStream_t s = (hipStream_t)stream;
    if (V > 0 && T > 0 && (!gather || P == 0)) HIP_TRY(hipMemsetAsync(d_vertices, 0, (size_t)T * V * 3 * 4, s));
    if (P == 0) return 0;   // (every Gaussian pruned: the vertex gradient is zero)
    CSPLAT_REQUIRE(T == 0 || (face_vertex_ids && vertices && bary && rotation && rest), "csplat_mesh_transform_bwd_views: NULL inputs");
    k_mesh_bwd<<<cdiv(P, 256), 256, 0, s>>>(T, P, V, face_vertex_ids, vertices, bary, rotation, (const RestFace *)rest, g_xyz, g_quat,
                                            d_vertices, d_bary, d_rotation, gather ? corner_scratch : nullptr);
    LAUNCH_CHECK();
    if (gather && V > 0 && T > 0) {
        k_vertex_gather<<<dim3(cdiv(8 * (int64_t)V, 256), T), 256, 0, s>>>(P, V, vertex_rowptr, vertex_corners, corner_scratch, d_vertices);
        LAUNCH_CHECK();
    }
    return 0;
}

int csplat_mesh_transform_fwd(void *stream, int P, const int64_t *face_vertex_ids, const float *vertices, const float *bary,
                              const float *rotation, const void *rest, float *out_xyz, float *out_quat) {
    return csplat_mesh_transform_fwd_views(stream, 1, P, 0, face_vertex_ids, vertices, bary, rotation, rest, out_xyz, out_quat);
}

int csplat_mesh_transform_bwd(void *stream, int P, int V, const int64_t *face_vertex_ids, const float *vertices,
                              const float *bary, const float *rotation, const void *rest, const float *g_xyz,
                              const float *g_quat, float *d_vertices, float *d_bary, float *d_rotation) {
    return csplat_mesh_transform_bwd_views(stream, 1, P, V, face_vertex_ids, vertices, bary, rotation, rest, g_xyz, g_quat, d_vertices,
                                           d_bary, d_rotation, nullptr, nullptr, nullptr);
}

int csplat_project_points(void *stream, int64_t n, const float *full_proj, int W, int H, const float *points, float *out_pixels) {
    CSPLAT_REQUIRE(n >= 0 && W > 0 && H > 0, "csplat_project_points: bad sizes");
    if (n == 0) return 0;
    CSPLAT_REQUIRE(full_proj && points && out_pixels, "csplat_project_points: NULL");
    k_project_points<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(n, full_proj, (float)W, (float)H, points, out_pixels);
    LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
