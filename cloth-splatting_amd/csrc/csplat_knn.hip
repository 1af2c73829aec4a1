// csplat_knn.hip -- simple_knn._C.distCUDA2 for gfx950: exact 3-nearest-neighbour mean squared distance.
//
// Replaces the CUDA extension called at /root/reference/scene_reconstruction/gaussian_mesh.py:250 and
// gaussian_model.py:134.  A single k-select kernel: one query point per lane, candidate
// points streamed through LDS in 1024-point slabs (every lane reads the same LDS word -> broadcast, no bank
// conflicts), the three best squared distances kept in registers.  Self is excluded by index, so coincident
// points contribute 0 exactly as upstream.  The distance is evaluated as dx*dx + dy*dy + dz*dz with FP
// contraction off, the association order of oracle/knn_ref.c: the result is bit-identical to the oracle.
#include "csplat_common.h"

namespace {
constexpr int KNN_THREADS = 256;
constexpr int KNN_SLAB = 1024;

__global__ __launch_bounds__(KNN_THREADS) void k_dist2(int P, const float *__restrict__ pts, float *__restrict__ out) {
#pragma clang fp contract(off)
    __shared__ float s_p[KNN_SLAB * 3];
    const int i = blockIdx.x * KNN_THREADS + threadIdx.x;
    const bool live = i < P;
    const float x = live ? pts[3 * i] : 0.f, y = live ? pts[3 * i + 1] : 0.f, z = live ? pts[3 * i + 2] : 0.f;
    float b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;
    for (int base = 0; base < P; base += KNN_SLAB) {
        const int cnt = min(KNN_SLAB, P - base);
        __syncthreads();
        for (int k = threadIdx.x; k < cnt * 3; k += KNN_THREADS) s_p[k] = pts[(size_t)base * 3 + k];
        __syncthreads();
        const int self = i - base;  // index of this lane's own point inside the slab, if any
#pragma unroll 4
        for (int j = 0; j < cnt; j++) {
            const float dx = s_p[3 * j] - x, dy = s_p[3 * j + 1] - y, dz = s_p[3 * j + 2] - z;
            float d = dx * dx + dy * dy + dz * dz;
            d = (j == self) ? INFINITY : d;
            // insert into the sorted triple (b0 <= b1 <= b2)
            const float n2 = fminf(b2, fmaxf(b1, d));
            const float n1 = fminf(b1, fmaxf(b0, d));
            const float n0 = fminf(b0, d);
            b0 = n0; b1 = n1; b2 = n2;
        }
    }
    if (live) out[i] = (b0 + b1 + b2) / 3.0f;
}
}  // namespace

extern "C" int csplat_dist2(void *stream, int P, const float *xyz, float *out) {
    CSPLAT_REQUIRE(P >= 0, "csplat_dist2: bad P");
    if (P == 0) return 0;
    ProfScope ps(PROF_KNN, (hipStream_t)stream);
    k_dist2<<<cdiv(P, KNN_THREADS), KNN_THREADS, 0, (hipStream_t)stream>>>(P, xyz, out);
    LAUNCH_CHECK();
    return 0;
}
