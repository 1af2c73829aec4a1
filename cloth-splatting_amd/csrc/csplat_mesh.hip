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
// pin against the SVD formulation.  The backward is forward-mode automatic differentiation inside the kernel: the SAME
// templated function is instantiated on a dual number carrying 4 tangent directions, swept 4 times over the 16 inputs of
// a Gaussian (9 vertex coordinates, 3 barycentric weights, 4 quaternion components) and contracted with the incoming
// gradient -- the backward cannot drift from the forward.  Vertex gradients are scattered with float atomics.
#include "csplat_common.h"

namespace {

template <int N>
struct Dual {
    float v;
    float d[N];
};
template <int N> __device__ __forceinline__ Dual<N> mk(float v) { Dual<N> r; r.v = v; for (int i = 0; i < N; i++) r.d[i] = 0.f; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator+(Dual<N> a, Dual<N> b) { Dual<N> r; r.v = a.v + b.v; for (int i = 0; i < N; i++) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator-(Dual<N> a, Dual<N> b) { Dual<N> r; r.v = a.v - b.v; for (int i = 0; i < N; i++) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator-(Dual<N> a) { Dual<N> r; r.v = -a.v; for (int i = 0; i < N; i++) r.d[i] = -a.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator*(Dual<N> a, Dual<N> b) { Dual<N> r; r.v = a.v * b.v; for (int i = 0; i < N; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator*(Dual<N> a, float b) { Dual<N> r; r.v = a.v * b; for (int i = 0; i < N; i++) r.d[i] = a.d[i] * b; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator*(float b, Dual<N> a) { return a * b; }
template <int N> __device__ __forceinline__ Dual<N> operator+(Dual<N> a, float b) { a.v += b; return a; }
template <int N> __device__ __forceinline__ Dual<N> operator-(float b, Dual<N> a) { Dual<N> r = -a; r.v += b; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator/(Dual<N> a, Dual<N> b) {
    Dual<N> r; const float inv = 1.f / b.v; r.v = a.v * inv;
    for (int i = 0; i < N; i++) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
    return r;
}
template <int N> __device__ __forceinline__ Dual<N> rsqrt_(Dual<N> a) {
    Dual<N> r; r.v = 1.f / sqrtf(a.v); const float k = -0.5f * r.v / a.v;
    for (int i = 0; i < N; i++) r.d[i] = k * a.d[i];
    return r;
}
__device__ __forceinline__ float rsqrt_(float a) { return 1.f / sqrtf(a); }
__device__ __forceinline__ float val(float a) { return a; }
template <int N> __device__ __forceinline__ float val(Dual<N> a) { return a.v; }
template <typename T> __device__ __forceinline__ T lift(float v);
template <> __device__ __forceinline__ float lift<float>(float v) { return v; }
template <> __device__ __forceinline__ Dual<4> lift<Dual<4>>(float v) { return mk<4>(v); }

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

__global__ __launch_bounds__(256) void k_mesh_fwd(int P, const int64_t *__restrict__ vid, const float *__restrict__ verts,
                                                   const float *__restrict__ bary, const float *__restrict__ rot,
                                                   const RestFace *__restrict__ rest, float *__restrict__ out_pos,
                                                   float *__restrict__ out_quat) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float y[3][3], b[3], q0[4], pos[3], quat[4];
    for (int k = 0; k < 3; k++)
        for (int c = 0; c < 3; c++) y[k][c] = verts[3 * vid[3 * (size_t)i + k] + c];
    for (int k = 0; k < 3; k++) b[k] = bary[3 * (size_t)i + k];
    for (int k = 0; k < 4; k++) q0[k] = rot[4 * (size_t)i + k];
    transform_one<float>(y, b, q0, rest[i], pos, quat);
    for (int c = 0; c < 3; c++) out_pos[3 * (size_t)i + c] = pos[c];
    for (int c = 0; c < 4; c++) out_quat[4 * (size_t)i + c] = quat[c];
}

// four threads per Gaussian, one per sweep of four tangent directions (inputs 4*sweep .. 4*sweep+3): 4x the wavefronts of a
// thread-per-Gaussian launch (100k Gaussians are only 1.5 waves per SIMD, each a long dependent chain)
__global__ __launch_bounds__(256) void k_mesh_bwd(int P, const int64_t *__restrict__ vid, const float *__restrict__ verts,
                                                   const float *__restrict__ bary, const float *__restrict__ rot,
                                                   const RestFace *__restrict__ rest, const float *__restrict__ g_pos,
                                                   const float *__restrict__ g_quat, float *__restrict__ d_verts,
                                                   float *__restrict__ d_bary, float *__restrict__ d_rot) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int i = gid >> 2, sweep = gid & 3;
    if (i >= P) return;
    typedef Dual<4> D;
    float in[16];
    int64_t v3[3];
    for (int k = 0; k < 3; k++) {
        v3[k] = vid[3 * (size_t)i + k];
        for (int c = 0; c < 3; c++) in[3 * k + c] = verts[3 * v3[k] + c];
    }
    for (int k = 0; k < 3; k++) in[9 + k] = bary[3 * (size_t)i + k];
    for (int k = 0; k < 4; k++) in[12 + k] = rot[4 * (size_t)i + k];
    float g[7];
    for (int c = 0; c < 3; c++) g[c] = g_pos ? g_pos[3 * (size_t)i + c] : 0.f;
    for (int c = 0; c < 4; c++) g[3 + c] = g_quat ? g_quat[4 * (size_t)i + c] : 0.f;
    const RestFace rf = rest[i];
    D y[3][3], b[3], q0[4], pos[3], quat[4];
#pragma unroll
    for (int t = 0; t < 16; t++) {
        D x = mk<4>(in[t]);
        x.d[t & 3] = (t >> 2) == sweep ? 1.f : 0.f;
        if (t < 9) y[t / 3][t % 3] = x;
        else if (t < 12) b[t - 9] = x;
        else q0[t - 12] = x;
    }
    transform_one<D>(y, b, q0, rf, pos, quat);
#pragma unroll
    for (int u = 0; u < 4; u++) {
        float s = 0.f;
        for (int c = 0; c < 3; c++) s += g[c] * pos[c].d[u];
        for (int c = 0; c < 4; c++) s += g[3 + c] * quat[c].d[u];
        const int t = 4 * sweep + u;   // input this derivative belongs to: 0-8 vertices, 9-11 barycentrics, 12-15 rotation
        if (t < 9) atomicAdd(d_verts + 3 * (t < 3 ? v3[0] : (t < 6 ? v3[1] : v3[2])) + t % 3, s);
        else if (t < 12) d_bary[3 * (size_t)i + (t - 9)] = s;
        else d_rot[4 * (size_t)i + (t - 12)] = s;
    }
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

int csplat_mesh_transform_fwd(void *stream, int P, const int64_t *face_vertex_ids, const float *vertices, const float *bary,
                              const float *rotation, const void *rest, float *out_xyz, float *out_quat) {
    CSPLAT_REQUIRE(P >= 0, "csplat_mesh_transform_fwd: bad P");
    if (P == 0) return 0;
    CSPLAT_REQUIRE(face_vertex_ids && vertices && bary && rotation && rest && out_xyz && out_quat, "csplat_mesh_transform_fwd: NULL");
    k_mesh_fwd<<<cdiv(P, 256), 256, 0, (hipStream_t)stream>>>(P, face_vertex_ids, vertices, bary, rotation, (const RestFace *)rest,
                                                              out_xyz, out_quat);
    LAUNCH_CHECK();
    return 0;
}

int csplat_mesh_transform_bwd(void *stream, int P, int V, const int64_t *face_vertex_ids, const float *vertices,
                              const float *bary, const float *rotation, const void *rest, const float *g_xyz,
                              const float *g_quat, float *d_vertices, float *d_bary, float *d_rotation) {
    CSPLAT_REQUIRE(P >= 0 && V >= 0, "csplat_mesh_transform_bwd: bad sizes");
    CSPLAT_REQUIRE((V == 0 || d_vertices) && (P == 0 || (d_bary && d_rotation)), "csplat_mesh_transform_bwd: NULL outputs");
    hipStream_t s = (hipStream_t)stream;
    if (V > 0) HIP_TRY(hipMemsetAsync(d_vertices, 0, (size_t)V * 3 * 4, s));
    if (P == 0) return 0;   // (every Gaussian pruned: the vertex gradient is zero)
    k_mesh_bwd<<<cdiv(4 * (int64_t)P, 256), 256, 0, s>>>(P, face_vertex_ids, vertices, bary, rotation, (const RestFace *)rest, g_xyz, g_quat,
                                            d_vertices, d_bary, d_rotation);
    LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
