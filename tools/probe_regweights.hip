// tools/probe_regweights.hip -- feasibility probe for a three-layer edge MLP whose WEIGHTS live in registers (round 5).
// One 4-wave workgroup per CU, one wave per SIMD (512 registers each).  Wave j owns output features 32j .. 32j + 31 of all three
// 128 x 128 layers: 3 layers x 3 bf16 pieces x 8 k-steps = 72 MFMA A operands = 288 VGPRs, loaded ONCE.  Activations travel between
// the layers through LDS as bf16-piece images (B operands: one ds_read_b128 per piece and step).  The probe runs the MFMA skeleton of
// that design -- per 64-row super-tile and layer: 2 tiles x 8 steps x 6 MFMAs per wave, operands from LDS, a token epilogue that
// writes the next layer's pieces, one barrier -- and reports cycles per super-tile against the 9,216-cycle MFMA floor.
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/probe_regweights.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int STRIDE = 136;                  // bf16 per activation row in LDS (272 B)
constexpr int TILE_ELEMS = 32 * STRIDE;      // one piece of one 32-row tile
constexpr int BUF_ELEMS = 2 * 3 * TILE_ELEMS;   // [tile 2][piece 3]

// token epilogue: ReLU, cut into three pieces, write this wave's 32 features of the tile's rows (4 x 8 bytes per piece)
__device__ __forceinline__ void epilogue(const f32x16 &acc, __bf16 *dst) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        __bf16 p1[4], p2[4], p3[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const float x = fmaxf(acc[4 * q + t], 0.f) * 1e-3f;
            p1[t] = (__bf16)x;
            const float r1 = x - (float)p1[t];
            p2[t] = (__bf16)r1;
            p3[t] = (__bf16)(r1 - (float)p2[t]);
        }
        *reinterpret_cast<uint2 *>(dst + 8 * q) = *reinterpret_cast<uint2 *>(p1);
        *reinterpret_cast<uint2 *>(dst + 8 * q + TILE_ELEMS) = *reinterpret_cast<uint2 *>(p2);
        *reinterpret_cast<uint2 *>(dst + 8 * q + 2 * TILE_ELEMS) = *reinterpret_cast<uint2 *>(p3);
    }
}

template <bool TWO, bool EPI, bool BAR>
__global__ __launch_bounds__(256) void k_probe(int iters, const bf16x8 *__restrict__ wimg, float *out, long long *clk) {
    extern __shared__ __bf16 s_act[];           // two buffers (layer in / layer out): 2 x BUF_ELEMS
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r32 = lane & 31, h = lane >> 5;
    bf16x8 W[3][3][8];
#pragma unroll
    for (int l = 0; l < 3; l++)
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int st = 0; st < 8; st++) W[l][p][st] = wimg[(((l * 4 + w) * 3 + p) * 8 + st) * 64 + lane];
    for (int t = threadIdx.x; t < 2 * BUF_ELEMS; t += 256) s_act[t] = (__bf16)(0.001f * (t & 255));
    __syncthreads();
    float sink = 0.f;
    const long long t0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int l = 0; l < 3; l++) {
            const __bf16 *in = s_act + (l & 1) * BUF_ELEMS;
            __bf16 *nxt = s_act + ((l + 1) & 1) * BUF_ELEMS;
            if constexpr (!TWO) {
#pragma unroll
            for (int tile = 0; tile < 2; tile++) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; r++) acc[r] = 0.f;
                const __bf16 *row = in + (size_t)tile * 3 * TILE_ELEMS + (size_t)r32 * STRIDE + 64 * h;
                bf16x8 b[3], bn[3];
#pragma unroll
                for (int p = 0; p < 3; p++) b[p] = *reinterpret_cast<const bf16x8 *>(row + p * TILE_ELEMS);
#pragma unroll
                for (int st = 0; st < 8; st++) {
                    if (st < 7) {
#pragma unroll
                        for (int p = 0; p < 3; p++) bn[p] = *reinterpret_cast<const bf16x8 *>(row + p * TILE_ELEMS + 8 * (st + 1));
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[l][0][st], b[2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[l][2][st], b[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[l][1][st], b[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[l][0][st], b[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[l][1][st], b[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[l][0][st], b[0], acc, 0, 0, 0);
#pragma unroll
                    for (int p = 0; p < 3; p++) b[p] = bn[p];
                }
                if constexpr (EPI) epilogue(acc, nxt + (size_t)tile * 3 * TILE_ELEMS + (size_t)r32 * STRIDE + 32 * w + 4 * h);
                sink += acc[0];
            }
            } else {
                f32x16 acc0, acc1;
#pragma unroll
                for (int r = 0; r < 16; r++) acc0[r] = acc1[r] = 0.f;
                const __bf16 *row = in + (size_t)r32 * STRIDE + 64 * h;
                bf16x8 b[6], bn[6];
#pragma unroll
                for (int p = 0; p < 6; p++) b[p] = *reinterpret_cast<const bf16x8 *>(row + p * TILE_ELEMS);
#pragma unroll
                for (int st = 0; st < 8; st++) {
                    if (st < 7) {
#pragma unroll
                        for (int p = 0; p < 6; p++) bn[p] = *reinterpret_cast<const bf16x8 *>(row + p * TILE_ELEMS + 8 * (st + 1));
                    }
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[l][0][st], b[2], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[l][0][st], b[5], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[l][2][st], b[0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[l][2][st], b[3], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[l][1][st], b[1], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[l][1][st], b[4], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[l][0][st], b[1], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[l][0][st], b[4], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[l][1][st], b[0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[l][1][st], b[3], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[l][0][st], b[0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[l][0][st], b[3], acc1, 0, 0, 0);
#pragma unroll
                    for (int p = 0; p < 6; p++) b[p] = bn[p];
                }
                if constexpr (EPI) {
                    epilogue(acc0, nxt + (size_t)r32 * STRIDE + 32 * w + 4 * h);
                    epilogue(acc1, nxt + (size_t)3 * TILE_ELEMS + (size_t)r32 * STRIDE + 32 * w + 4 * h);
                }
                sink += acc0[0] + acc1[0];
            }
            if constexpr (BAR) __syncthreads();
        }
    }
    const long long t1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = sink;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main() {
    const int iters = 200, grid = 256;
    bf16x8 *wimg; float *out; long long *clk;
    hipMalloc(&wimg, 3 * 4 * 3 * 8 * 64 * 16); hipMemset(wimg, 0x3c, 3 * 4 * 3 * 8 * 64 * 16);
    hipMalloc(&out, grid * 256 * 4); hipMalloc(&clk, grid * 8);
    const size_t lds = (size_t)2 * BUF_ELEMS * 2;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char *name, void (*kern)(int, const bf16x8 *, float *, long long *)) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            kern<<<grid, 256, lds>>>(iters, wimg, out, clk);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            best = ms < best ? ms : best;
        }
        printf("%-28s %.2f us per 64-row super-tile = %.0f cycles at 2.4 GHz (MFMA floor 9216)  err=%s\n", name, best * 1e3 / iters,
               best * 1e3 / iters * 2400.0, hipGetErrorString(hipGetLastError()));
    };
    run("one chain, epi, barrier", k_probe<false, true, true>);
    run("one chain, no epi, barrier", k_probe<false, false, true>);
    run("one chain, no epi, no bar", k_probe<false, false, false>);
    run("two chains, epi, barrier", k_probe<true, true, true>);
    run("two chains, no epi, barrier", k_probe<true, false, true>);
    run("two chains, no epi, no bar", k_probe<true, false, false>);
    return 0;
}
