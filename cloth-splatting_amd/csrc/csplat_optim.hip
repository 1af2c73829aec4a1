// csplat_optim.hip -- optimizer step on the device side of the training loop.
#include "csplat_common.h"
#include <math.h>

// ---- multi-tensor Adam (first half): torch.optim.Adam walks its parameter groups one at a time, and
// the reference gives every Gaussian attribute its own group (scene_reconstruction/gaussian_mesh.py:126-136): 7 groups x
// ~8 elementwise launches per step for ~25 us of memory traffic.  One launch here: blockIdx.y = tensor, same arithmetic
// order as torch's foreach implementation (lerp, mul+addcmul, sqrt / bias_correction2_sqrt + eps, addcdiv).
namespace {
struct AdamDesc { float *p; const float *g; float *m, *v; long long n; float step_size; int pad; };
struct AdamTable { AdamDesc d[CSPLAT_ADAM_MAX_TENSORS]; };
__global__ __launch_bounds__(256) void k_adam(AdamTable tab, float beta2, float w1, float w2, float eps, float bc2_sqrt) {
    const AdamDesc d = tab.d[blockIdx.y];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < d.n; i += (long long)gridDim.x * 256) {
        const float g = d.g[i];
        float m = d.m[i], v = d.v[i];
        m = m + w1 * (g - m);                       // exp_avg.lerp_(grad, 1 - beta1)
        v = v * beta2 + (w2 * g) * g;               // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        d.m[i] = m; d.v[i] = v;
        d.p[i] = d.p[i] - d.step_size * (m / denom);  // param.addcdiv_(exp_avg, denom, value=-step_size)
    }
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
        const int64_t want = (longest + 1023) / 1024;
        dim3 grid((unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want)), (unsigned)cnt);
        k_adam<<<grid, 256, 0, (hipStream_t)stream>>>(tab, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, bc2_sqrt);
        LAUNCH_CHECK();
    }
    return 0;
}
