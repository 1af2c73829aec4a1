// csplat_common.h -- shared host-side helpers for libcsplat.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/csplat.h"

#define CSPLAT_TILE 16
#define CSPLAT_TILE_PIX 256
#define CSPLAT_WAVE 64

extern thread_local char g_csplat_err[512];

static inline int csplat_fail(const char *what, const char *file, int line, hipError_t e) {
    snprintf(g_csplat_err, sizeof(g_csplat_err), "%s failed at %s:%d: %s", what, file, line,
             e == hipSuccess ? "" : hipGetErrorString(e));
    return 1;
}

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) return csplat_fail(#expr, __FILE__, __LINE__, _e);       \
    } while (0)

#define LAUNCH_CHECK() HIP_TRY(hipGetLastError())

#define CSPLAT_REQUIRE(cond, msg)                                                      \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            snprintf(g_csplat_err, sizeof(g_csplat_err), "%s (%s:%d)", msg, __FILE__, __LINE__); \
            return 2;                                                                  \
        }                                                                              \
    } while (0)

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- device-wide primitives implemented in csplat_sort.hip -------------------------------------
// inclusive scan of n uint32 (in -> out), total also written to *total_dev (device u32).  temp >= scan_temp_bytes(n)
size_t csplat_scan_temp_bytes(int64_t n);
int csplat_inclusive_scan_u32(hipStream_t s, const uint32_t *in, uint32_t *out, int64_t n, void *temp);

// stable LSD radix sort of (u64 key, u32 value) pairs on key bits [0, end_bit).  Result ends in keys_out/vals_out.
size_t csplat_sort_temp_bytes(int64_t n);
int csplat_sort_pairs(hipStream_t s, const uint64_t *keys_in, const uint32_t *vals_in, uint64_t *keys_out,
                      uint32_t *vals_out, uint64_t *keys_tmp, uint32_t *vals_tmp, int64_t n, int end_bit, void *temp);

// measurement hook (csplat_debug_stamps): the caller-provided device buffer, or nullptr when none of `need_words` u64 words is registered
unsigned long long *csplat_stamp_buffer(size_t need_words);

// ---- optional event bracketing (csplat_prof_*), implemented in csplat_sort.hip --------------------
enum { PROF_K1 = 0, PROF_K2, PROF_K3, PROF_K4, PROF_K5, PROF_K6, PROF_K7, PROF_K8, PROF_KNN, PROF_GNN, PROF_NCLASSES };
extern unsigned g_csplat_prof_mask;
void csplat_prof_mark(int cls, hipStream_t s, bool begin);
struct ProfScope {
    int cls; hipStream_t s; bool on;
    ProfScope(int c, hipStream_t st) : cls(c), s(st), on((g_csplat_prof_mask >> c) & 1u) { if (on) csplat_prof_mark(cls, s, true); }
    ~ProfScope() { if (on) csplat_prof_mark(cls, s, false); }
};

// ReLU as torch applies it: a NaN stays a NaN.  fmaxf(x, 0) (v_max_f32) returns 0 for a NaN -- and an overflow of the fp16-piece kernels
// upstream (csplat_edge_mlp.hip) would be laundered into finite garbage by the next layer's ReLU; round 6: every ReLU of the library lets
// it through, so that it reaches the output where it is detected (meshnet/graph_network.py, meshnet/rollout.py).
__device__ __forceinline__ float relu_keep_nan(float x) { return x < 0.f ? 0.f : x; }
