// csplat_optim.hip -- optimizer step on the device side of the training loop.
#include "csplat_common.h"
#include <math.h>

// ---- multi-tensor Adam (first half): torch.optim.Adam walks its parameter groups one at a time, and
// the reference gives every Gaussian attribute its own group (scene_reconstruction/gaussian_mesh.py:126-136): 7 groups x
// ~8 elementwise launches per step for ~25 us of memory traffic.  One launch here: blockIdx.y = tensor, same arithmetic
// order as torch's foreach implementation (lerp, mul+addcmul, sqrt / bias_correction2_sqrt + eps, addcdiv).
namespace {
// one Adam update; the same arithmetic order as torch's foreach implementation
__device__ __forceinline__ void adam1(float &p, float g, float &m, float &v, float w1, float w2, float b2, float eps, float bc2_sqrt, float step_size) {
    m = m + w1 * (g - m);                       // exp_avg.lerp_(grad, 1 - beta1)
    v = v * b2 + (w2 * g) * g;                  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);            // param.addcdiv_(exp_avg, denom, value=-step_size)
}
// the tensor's elements from workgroup `bx` of `nbx`: 16 bytes per lane and array while the four arrays are 16-byte aligned (28 bytes move
// per element; 4-byte accesses held the launch at 2.9 TB/s), the <= 3 elements behind the last whole float4 by workgroup 0
__device__ __forceinline__ void adam_span(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                                          long long n, int bx, int nbx, float w1, float w2, float b2, float eps, float bc2_sqrt, float step_size) {
    const bool al = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15u) == 0);
    const long long n4 = al ? n >> 2 : 0;
    for (long long i = (long long)bx * 256 + threadIdx.x; i < n4; i += (long long)nbx * 256) {
        const float4 g4 = reinterpret_cast<const float4 *>(g)[i];
        float4 m4 = reinterpret_cast<float4 *>(m)[i], v4 = reinterpret_cast<float4 *>(v)[i], p4 = reinterpret_cast<float4 *>(p)[i];
        adam1(p4.x, g4.x, m4.x, v4.x, w1, w2, b2, eps, bc2_sqrt, step_size);
        adam1(p4.y, g4.y, m4.y, v4.y, w1, w2, b2, eps, bc2_sqrt, step_size);
        adam1(p4.z, g4.z, m4.z, v4.z, w1, w2, b2, eps, bc2_sqrt, step_size);
        adam1(p4.w, g4.w, m4.w, v4.w, w1, w2, b2, eps, bc2_sqrt, step_size);
        reinterpret_cast<float4 *>(m)[i] = m4; reinterpret_cast<float4 *>(v)[i] = v4; reinterpret_cast<float4 *>(p)[i] = p4;
    }
    for (long long i = (n4 << 2) + (long long)bx * 256 + threadIdx.x; i < n; i += (long long)nbx * 256) {
        float pm = p[i], mm = m[i], vm = v[i];
        adam1(pm, g[i], mm, vm, w1, w2, b2, eps, bc2_sqrt, step_size);
        m[i] = mm; v[i] = vm; p[i] = pm;
    }
}
struct AdamDesc { float *p; const float *g; float *m, *v; long long n; float step_size; int pad; };
struct AdamTable { AdamDesc d[CSPLAT_ADAM_MAX_TENSORS]; };
__global__ __launch_bounds__(256) void k_adam(AdamTable tab, float beta2, float w1, float w2, float eps, float bc2_sqrt) {
    const AdamDesc d = tab.d[blockIdx.y];
    adam_span(d.p, d.g, d.m, d.v, d.n, (int)blockIdx.x, (int)gridDim.x, w1, w2, beta2, eps, bc2_sqrt, d.step_size);
}
}  // namespace

// (hyper-parameters arrive as doubles and are combined in double before the cast to fp32, as torch does with its Python
// floats: 1 - 0.999 must be 1.0e-3, not 1 - 0.999f)
extern "C" int csplat_adam_step(void *stream, int n_tensors, float *const *params, const float *const *grads, float *const *exp_avg,
                                float *const *exp_avg_sq, const int64_t *numel, const double *lr, double beta1, double beta2,
                                double eps, int64_t step) {
    CSPLAT_REQUIRE(n_tensors >= 0 && step >= 1 && (n_tensors == 0 || (params && grads && exp_avg && exp_avg_sq && numel && lr)),
                   "csplat_adam_step: bad arguments");
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const float bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    for (int base = 0; base < n_tensors; base += CSPLAT_ADAM_MAX_TENSORS) {
        AdamTable tab;
        memset(&tab, 0, sizeof(tab));
        const int cnt = n_tensors - base < CSPLAT_ADAM_MAX_TENSORS ? n_tensors - base : CSPLAT_ADAM_MAX_TENSORS;
        int64_t longest = 0;
        for (int i = 0; i < cnt; i++) {
            CSPLAT_REQUIRE(numel[base + i] == 0 || (params[base + i] && grads[base + i] && exp_avg[base + i] && exp_avg_sq[base + i]),
                           "csplat_adam_step: NULL tensor");
            tab.d[i] = AdamDesc{params[base + i], grads[base + i], exp_avg[base + i], exp_avg_sq[base + i], (long long)numel[base + i],
                                (float)(lr[base + i] / bc1), 0};
            longest = numel[base + i] > longest ? numel[base + i] : longest;
        }
        if (longest == 0) continue;
        const int64_t want = (longest + 4095) / 4096;
        dim3 grid((unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want)), (unsigned)cnt);
        k_adam<<<grid, 256, 0, (hipStream_t)stream>>>(tab, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, bc2_sqrt);
        LAUNCH_CHECK();
    }
    return 0;
}

// ---- the same step with its step count, learning rates and go / no-go word ON THE DEVICE: nothing in the launch depends on a value
// the host would have to compute per step, so the launch can be recorded into a hipGraph and replayed (csplat.train.CapturedStep).
//   state[0] (int32): steps taken so far; the kernels use state[0] + 1 and k_adam_tick advances it -- unless *valid == 0
//   lr (double[n_tensors], device): the groups' learning rates (the host rewrites them when a schedule moves them)
//   valid (device word or NULL): 0 = the step's gradients are not to be applied (a forward launched on faith that did not fit)
// Arithmetic as csplat_adam_step: the bias corrections are formed in double from the count, per workgroup.
namespace {
struct AdamDevDesc { float *p; const float *g; float *m, *v; long long n; };
struct AdamDevTable { AdamDevDesc d[CSPLAT_ADAM_MAX_TENSORS]; };
__global__ __launch_bounds__(256) void k_adam_dev(AdamDevTable tab, const double *__restrict__ lr, double beta1, double beta2, float eps,
                                                  const int *__restrict__ state, const uint32_t *__restrict__ valid) {
    if (valid && *valid == 0u) return;
    const AdamDevDesc d = tab.d[blockIdx.y];
    {   // a short tensor in a launch sized for the longest: leave before the two pow() (units as adam_span walks them)
        const bool al = ((((uintptr_t)d.p | (uintptr_t)d.g | (uintptr_t)d.m | (uintptr_t)d.v) & 15u) == 0);
        const long long units = al ? (d.n + 3) / 4 : d.n;
        if ((long long)blockIdx.x * 256 >= units) return;
    }
    __shared__ float s_c[2];
    if (threadIdx.x == 0) {
        const double step = (double)(state[0] + 1);
        const double bc1 = 1.0 - pow(beta1, step);
        s_c[0] = (float)(lr[blockIdx.y] / bc1);
        s_c[1] = (float)sqrt(1.0 - pow(beta2, step));
    }
    __syncthreads();
    const float step_size = s_c[0], bc2_sqrt = s_c[1];
    const float w1 = (float)(1.0 - beta1), w2 = (float)(1.0 - beta2), b2 = (float)beta2;
    adam_span(d.p, d.g, d.m, d.v, d.n, (int)blockIdx.x, (int)gridDim.x, w1, w2, b2, eps, bc2_sqrt, step_size);
}
__global__ void k_adam_tick(int *state, const uint32_t *__restrict__ valid) {
    if (threadIdx.x == 0 && !(valid && *valid == 0u)) state[0] += 1;
}
}  // namespace

extern "C" int csplat_adam_step_dev(void *stream, int n_tensors, float *const *params, const float *const *grads, float *const *exp_avg,
                                    float *const *exp_avg_sq, const int64_t *numel, const double *lr_dev, double beta1, double beta2,
                                    double eps, int *state_dev, const uint32_t *valid_dev) {
    CSPLAT_REQUIRE(n_tensors >= 1 && n_tensors <= CSPLAT_ADAM_MAX_TENSORS && params && grads && exp_avg && exp_avg_sq && numel && lr_dev &&
                   state_dev, "csplat_adam_step_dev: bad arguments (1..48 tensors)");
    AdamDevTable tab;
    memset(&tab, 0, sizeof(tab));
    int64_t longest = 0;
    for (int i = 0; i < n_tensors; i++) {
        CSPLAT_REQUIRE(numel[i] == 0 || (params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i]), "csplat_adam_step_dev: NULL tensor");
        tab.d[i] = AdamDevDesc{params[i], grads[i], exp_avg[i], exp_avg_sq[i], (long long)numel[i]};
        longest = numel[i] > longest ? numel[i] : longest;
    }
    if (longest > 0) {
        const int64_t want = (longest + 4095) / 4096;
        dim3 grid((unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want)), (unsigned)n_tensors);
        k_adam_dev<<<grid, 256, 0, (hipStream_t)stream>>>(tab, lr_dev, beta1, beta2, (float)eps, state_dev, valid_dev);
        LAUNCH_CHECK();
    }
    k_adam_tick<<<1, 64, 0, (hipStream_t)stream>>>(state_dev, valid_dev);
    LAUNCH_CHECK();
    return 0;
}

// ---- the log line of a recorded step: up to 32 scalars scattered over device tensors (a step count, a go / no-go word, PSNR, loss, the
// views' instance counts ...) gathered into ONE float array by one launch, so that one copy node carries them to pinned host memory
namespace {
struct WordsTable { const void *src[32]; int kind[32]; int count[32]; int n; };      // kind: 0 float, 1 int32 -> float (exact below 2^24), 2 int32 bits
__global__ void k_gather_words(WordsTable tab, float *__restrict__ dst) {
    int o = 0;
    for (int i = 0; i < tab.n; i++) {
        for (int k = threadIdx.x; k < tab.count[i]; k += blockDim.x)
            dst[o + k] = tab.kind[i] == 1 ? (float)reinterpret_cast<const int *>(tab.src[i])[k] : reinterpret_cast<const float *>(tab.src[i])[k];
        o += tab.count[i];
    }
}
}  // namespace
extern "C" int csplat_gather_words(void *stream, int n, const void *const *src, const int *kind, const int *count, float *dst) {
    CSPLAT_REQUIRE(n >= 1 && n <= 32 && src && kind && count && dst, "csplat_gather_words: 1..32 sources");
    WordsTable tab;
    memset(&tab, 0, sizeof(tab));
    tab.n = n;
    for (int i = 0; i < n; i++) {
        CSPLAT_REQUIRE(src[i] && count[i] >= 1 && kind[i] >= 0 && kind[i] <= 2, "csplat_gather_words: bad source");
        tab.src[i] = src[i]; tab.kind[i] = kind[i]; tab.count[i] = count[i];
    }
    k_gather_words<<<1, 64, 0, (hipStream_t)stream>>>(tab, dst);
    LAUNCH_CHECK();
    return 0;
}

// ---- capacity-based densify / prune (second half).  The reference re-creates every nn.Parameter and both
// Adam moments of all 7 attribute groups with boolean-mask indexing / torch.cat whenever the number of Gaussians changes
// (scene_reconstruction/gaussian_model.py:266-341, gaussian_mesh.py:336-431): ~60 allocations and gathers per surgery.  Here
// the attributes, their moments and the per-Gaussian statistics live in capacity buffers (csplat/store.py); a surgery is
//   csplat_mask_to_map   keep / select mask -> destination row of every source row (stable: order is preserved), count
//   csplat_rows_scatter  ONE launch moving (or zero-filling) the rows of up to CSPLAT_ROWS_MAX_TENSORS tensors
// and the Parameter objects are re-pointed at the first `count` rows of the destination buffers.
namespace {
__global__ __launch_bounds__(256) void k_mask_flags(int64_t n, const uint8_t *__restrict__ mask, uint32_t *__restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) flags[i] = mask[i] ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_mask_map(int64_t n, const uint8_t *__restrict__ mask, const uint32_t *__restrict__ scan,
                                                   int32_t base, int32_t *__restrict__ map, int32_t *__restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) map[i] = mask[i] ? base + (int32_t)scan[i] - 1 : -1;
    if (i == n - 1) *count = (int32_t)scan[i];
}
struct RowsDesc { const uint32_t *src; uint32_t *dst; long long words; };      // words per row (4-byte units); src NULL = zero fill
struct RowsTable { RowsDesc d[CSPLAT_ROWS_MAX_TENSORS]; };
__global__ __launch_bounds__(256) void k_rows_scatter(RowsTable tab, int64_t n_rows, const int32_t *__restrict__ map) {
    const RowsDesc d = tab.d[blockIdx.y];
    const long long total = d.words * n_rows;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long row = e / d.words;
        const int32_t to = map[row];
        if (to >= 0) d.dst[(long long)to * d.words + (e - row * d.words)] = d.src ? d.src[e] : 0u;
    }
}
}  // namespace

extern "C" size_t csplat_mask_to_map_temp_bytes(int64_t n) { return align256((size_t)(n > 0 ? n : 1) * 4) * 2 + csplat_scan_temp_bytes(n) + 256; }

extern "C" int csplat_mask_to_map(void *stream, int64_t n, const uint8_t *mask, int32_t base, int32_t *map, int32_t *count_dev, void *temp) {
    CSPLAT_REQUIRE(n >= 0 && (n == 0 || (mask && map && temp)) && count_dev, "csplat_mask_to_map: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) { HIP_TRY(hipMemsetAsync(count_dev, 0, 4, s)); return 0; }
    uint32_t *flags = (uint32_t *)temp;
    uint32_t *scan = (uint32_t *)((char *)temp + align256((size_t)n * 4));
    void *stmp = (char *)temp + 2 * align256((size_t)n * 4);
    k_mask_flags<<<cdiv(n, 256), 256, 0, s>>>(n, mask, flags);
    LAUNCH_CHECK();
    if (int rc = csplat_inclusive_scan_u32(s, flags, scan, n, stmp)) return rc;
    k_mask_map<<<cdiv(n, 256), 256, 0, s>>>(n, mask, scan, base, map, count_dev);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int csplat_rows_scatter(void *stream, int n_tensors, const void *const *src, void *const *dst, const int64_t *row_bytes,
                                   int64_t n_rows, const int32_t *map) {
    CSPLAT_REQUIRE(n_tensors >= 0 && n_tensors <= CSPLAT_ROWS_MAX_TENSORS && n_rows >= 0, "csplat_rows_scatter: bad sizes");
    if (n_tensors == 0 || n_rows == 0) return 0;
    CSPLAT_REQUIRE(src && dst && row_bytes && map, "csplat_rows_scatter: NULL");
    RowsTable tab;
    memset(&tab, 0, sizeof(tab));
    long long longest = 0;
    for (int i = 0; i < n_tensors; i++) {
        CSPLAT_REQUIRE(dst[i] && row_bytes[i] > 0 && row_bytes[i] % 4 == 0, "csplat_rows_scatter: rows are whole 4-byte words");
        tab.d[i] = RowsDesc{(const uint32_t *)src[i], (uint32_t *)dst[i], (long long)(row_bytes[i] / 4)};
        longest = tab.d[i].words > longest ? tab.d[i].words : longest;
    }
    const long long want = (longest * n_rows + 1023) / 1024;
    dim3 grid((unsigned)(want < 1 ? 1 : (want > 4096 ? 4096 : want)), (unsigned)n_tensors);
    k_rows_scatter<<<grid, 256, 0, (hipStream_t)stream>>>(tab, n_rows, map);
    LAUNCH_CHECK();
    return 0;
}

// ---- the per-step activations of the Gaussian parameters, gaussian_model.py:96-121 as render() uses them
// (gaussian_renderer/__init__.py:92-118): opacity = sigmoid(_opacity), scaling = exp(_scaling), features = cat(_features_dc,
// _features_rest) -- three launches forward and five backward (sigmoid / exp backward, two slice copies, an add) for 21 MB of traffic,
// in a training step that is bound by the host's launch rate.  One launch each way: element e of the [P][52] row (opacity | 3 scales |
// 48 SH values, dc first).
namespace {
// one grid-stride pass over the P * 48 SH values (coalesced writes, reads in two nearly contiguous streams), with the opacity and the
// three scales of Gaussian i handled by the thread that copies its first SH value
__global__ __launch_bounds__(256) void k_gauss_act_fwd(int64_t P, const float *__restrict__ op_raw, const float *__restrict__ sc_raw,
                                                        const float *__restrict__ f_dc, const float *__restrict__ f_rest,
                                                        float *__restrict__ opacity, float *__restrict__ scales, float *__restrict__ shs) {
    const uint32_t n = (uint32_t)(P * 48);            // (the host refuses P * 48 >= 2^31: 32-bit index arithmetic, division by a constant)
    for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < n; e += gridDim.x * 256u) {
        const uint32_t i = e / 48u;
        const int k = (int)(e - i * 48u);
        shs[e] = k < 3 ? f_dc[3 * i + k] : f_rest[45 * i + k - 3];
        if (k == 0) opacity[i] = 1.f / (1.f + expf(-op_raw[i]));
        else if (k < 4) scales[3 * i + k - 1] = expf(sc_raw[3 * i + k - 1]);
    }
}
__global__ __launch_bounds__(256) void k_gauss_act_bwd(int64_t P, const float *__restrict__ opacity, const float *__restrict__ scales,
                                                        const float *__restrict__ g_op, const float *__restrict__ g_sc,
                                                        const float *__restrict__ g_shs, float *__restrict__ d_op_raw,
                                                        float *__restrict__ d_sc_raw, float *__restrict__ d_dc, float *__restrict__ d_rest) {
    const uint32_t n = (uint32_t)(P * 48);            // (the host refuses P * 48 >= 2^31: 32-bit index arithmetic, division by a constant)
    for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < n; e += gridDim.x * 256u) {
        const uint32_t i = e / 48u;
        const int k = (int)(e - i * 48u);
        const float g = g_shs ? g_shs[e] : 0.f;
        if (k < 3) d_dc[3 * i + k] = g; else d_rest[45 * i + k - 3] = g;
        if (k == 0) { const float o = opacity[i]; d_op_raw[i] = g_op ? g_op[i] * ((1.f - o) * o) : 0.f; }     // torch: grad * (1 - y) * y
        else if (k < 4) d_sc_raw[3 * i + k - 1] = g_sc ? g_sc[3 * i + k - 1] * scales[3 * i + k - 1] : 0.f;
    }
}
}  // namespace

extern "C" int csplat_gauss_act_fwd(void *stream, int64_t P, const float *opacity_raw, const float *scaling_raw, const float *features_dc,
                                    const float *features_rest, float *opacity, float *scales, float *shs) {
    CSPLAT_REQUIRE(P >= 0 && P < (int64_t)0x7FFFFFFF / 48 && (P == 0 || (opacity_raw && scaling_raw && features_dc && features_rest && opacity && scales && shs)),
                   "csplat_gauss_act_fwd: bad arguments");
    if (P == 0) return 0;
    const int64_t want = cdiv(P * 48, 256);
    k_gauss_act_fwd<<<(unsigned)(want > 8192 ? 8192 : want), 256, 0, (hipStream_t)stream>>>(P, opacity_raw, scaling_raw, features_dc,
                                                                                            features_rest, opacity, scales, shs);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int csplat_gauss_act_bwd(void *stream, int64_t P, const float *opacity, const float *scales, const float *g_opacity,
                                    const float *g_scales, const float *g_shs, float *d_opacity_raw, float *d_scaling_raw,
                                    float *d_features_dc, float *d_features_rest) {
    CSPLAT_REQUIRE(P >= 0 && P < (int64_t)0x7FFFFFFF / 48 && (P == 0 || (opacity && scales && d_opacity_raw && d_scaling_raw && d_features_dc && d_features_rest)),
                   "csplat_gauss_act_bwd: bad arguments");
    if (P == 0) return 0;
    const int64_t want = cdiv(P * 48, 256);
    k_gauss_act_bwd<<<(unsigned)(want > 8192 ? 8192 : want), 256, 0, (hipStream_t)stream>>>(P, opacity, scales, g_opacity, g_scales, g_shs,
                                                                                            d_opacity_raw, d_scaling_raw, d_features_dc,
                                                                                            d_features_rest);
    LAUNCH_CHECK();
    return 0;
}

// ---- the densification statistics of a training step in ONE launch (scene_reconstruction/train_utils.py:276-285: radii =
// cat(radii_list).max(0), visibility = cat(filters).any(0), viewspace gradient = sum over the step's cameras): as torch ops two cats,
// two reductions and a compare.  V <= CSPLAT_STATS_MAX_VIEWS views: their screen-space gradients [P][3] are summed in view order,
// the radii [P] take the maximum, visible = max > 0.
namespace {
constexpr int STATS_MAX_VIEWS = 16;
struct StatsTable { const float *g[STATS_MAX_VIEWS]; const int *r[STATS_MAX_VIEWS]; };
__global__ __launch_bounds__(256) void k_step_stats(int64_t P, int V, StatsTable tab, float *__restrict__ grad_sum, int *__restrict__ radii_max,
                                                    uint8_t *__restrict__ visible) {
    const int64_t n = 3 * P;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (grad_sum) {
            float s = tab.g[0] ? tab.g[0][i] : 0.f;
            for (int v = 1; v < V; v++) s += tab.g[v] ? tab.g[v][i] : 0.f;
            grad_sum[i] = s;
        }
        if (i < P && radii_max) {
            int m = tab.r[0][i];
            for (int v = 1; v < V; v++) m = max(m, tab.r[v][i]);
            radii_max[i] = m;
            if (visible) visible[i] = m > 0 ? 1 : 0;
        }
    }
}
}  // namespace

extern "C" int csplat_step_stats(void *stream, int64_t P, int V, const float *const *mean2d_grads, const int *const *radii, float *grad_sum,
                                 int *radii_max, uint8_t *visible) {
    CSPLAT_REQUIRE(P >= 0 && V >= 1 && V <= STATS_MAX_VIEWS, "csplat_step_stats: 1 <= V <= 16 views");
    CSPLAT_REQUIRE((grad_sum == nullptr || mean2d_grads) && (radii_max == nullptr || radii) && (visible == nullptr || radii_max),
                   "csplat_step_stats: bad arguments");
    if (P == 0) return 0;
    StatsTable tab;
    memset(&tab, 0, sizeof(tab));
    for (int v = 0; v < V; v++) {
        if (grad_sum) tab.g[v] = mean2d_grads[v];          // (a NULL entry = a view without a gradient: counts as zero)
        if (radii_max) { CSPLAT_REQUIRE(radii[v], "csplat_step_stats: NULL radii"); tab.r[v] = radii[v]; }
    }
    const int64_t work = (3 * P + 255) / 256;
    k_step_stats<<<(int)(work > 4096 ? 4096 : work), 256, 0, (hipStream_t)stream>>>(P, V, tab, grad_sum, radii_max, visible);
    LAUNCH_CHECK();
    return 0;
}
