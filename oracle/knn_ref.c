/*
 * oracle/knn_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of simple_knn._C.distCUDA2, called at
 *   /root/reference/scene_reconstruction/gaussian_mesh.py:250 and gaussian_model.py:134.
 * PARITY UNPINNED: `bkerbl/simple-knn` is an un-vendored, un-pinned submodule
 * (/root/reference/.gitmodules:1-3, directory empty).  Published behaviour:
 * for every point the mean of the squared distances to its 3 nearest OTHER points (self excluded
 * by index, so coincident points contribute 0).  Upstream's Morton-box pruning is conservative,
 * i.e. the result equals brute force; this file IS the brute force, O(P^2), fp32 or fp64 by REAL,
 * distance evaluated as dx*dx + dy*dy + dz*dz in that order.
 * tests/ additionally pins this against scipy.spatial.cKDTree.
 */
#include <stdint.h>
#include <math.h>

#ifndef REAL
#define REAL float
#endif

void oracle_dist2(int P, const REAL *pts, REAL *out) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        REAL b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;
        REAL x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        for (int j = 0; j < P; j++) {
            if (j == i) continue;
            REAL dx = pts[3 * j] - x, dy = pts[3 * j + 1] - y, dz = pts[3 * j + 2] - z;
            REAL d = dx * dx + dy * dy + dz * dz;
            if (d < b2) {
                if (d < b1) {
                    b2 = b1;
                    if (d < b0) { b1 = b0; b0 = d; } else b1 = d;
                } else b2 = d;
            }
        }
        out[i] = (b0 + b1 + b2) / (REAL)3;
    }
}
