// tools/mfma_rate.hip -- measures the sustained issue rate of v_mfma_f32_32x32x2_f32 on this device (and the shader clock
// while it runs), to anchor the fp32-MFMA roofline used for csplat_linear128 in DESIGN.md.
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(int iters, float *out, long long *clk) {
    f32x16 acc[NACC];
    for (int c = 0; c < NACC; c++) for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int c = 0; c < NACC; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    long long t1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int c = 0; c < NACC; c++) for (int r = 0; r < 16; r++) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

// the inner loop of csplat_linear128 in isolation: B operand from LDS (k-major W^T, stride 129), A operand from registers
template <int GROUP, bool BARRIER>
__global__ __launch_bounds__(256, 2) void k2(int tiles, float *out) {
    extern __shared__ float s_wt[];
    for (int t = threadIdx.x; t < 128 * 129; t += 256) s_wt[t] = t * 1e-5f;
    __syncthreads();
    const int lane = threadIdx.x & 63, r32 = lane & 31, h = lane >> 5;
    const float *wrow = s_wt + (64 * h) * 129 + r32;
    float x[64];
    for (int i = 0; i < 64; i++) x[i] = (threadIdx.x + i) * 1e-3f;
    float s = 0.f;
    for (int tile = 0; tile < tiles; tile++) {
        f32x16 acc[4];
        for (int c = 0; c < 4; c++) for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
        float bc[4 * GROUP], bn[4 * GROUP];
#pragma unroll
        for (int u = 0; u < GROUP; u++)
#pragma unroll
            for (int c = 0; c < 4; c++) bc[4 * u + c] = wrow[u * 129 + 32 * c];
#pragma unroll
        for (int g = 0; g < 64 / GROUP; g++) {
            if (g < 64 / GROUP - 1) {
#pragma unroll
                for (int u = 0; u < GROUP; u++)
#pragma unroll
                    for (int c = 0; c < 4; c++) bn[4 * u + c] = wrow[(GROUP * (g + 1) + u) * 129 + 32 * c];
            }
            if (BARRIER) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < GROUP; u++)
#pragma unroll
                for (int c = 0; c < 4; c++)
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[GROUP * g + u], bc[4 * u + c], acc[c], 0, 0, 0);
            if (BARRIER) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4 * GROUP; i++) bc[i] = bn[i];
        }
        for (int c = 0; c < 4; c++) for (int r = 0; r < 16; r++) s += acc[c][r];
        for (int i = 0; i < 64; i++) x[i] += s * 1e-9f;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int GROUP, bool BARRIER>
void run2(int wgs, int tiles) {
    float *out; hipMalloc(&out, (size_t)wgs * 256 * 4);
    const size_t lds = 128 * 129 * 4;
    hipFuncSetAttribute((const void *)k2<GROUP, BARRIER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k2<GROUP, BARRIER><<<wgs, 256, lds>>>(tiles, out);
    hipEventRecord(e0);
    k2<GROUP, BARRIER><<<wgs, 256, lds>>>(tiles, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double nmfma = (double)wgs * 4 * tiles * 256.0;
    printf("LDS-fed loop GROUP=%d barrier=%d wgs=%d: %.3f ms  %.1f TFLOP/s  (%s)\n", GROUP, (int)BARRIER, wgs, ms,
           nmfma * 4096.0 / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
    hipFree(out);
}

// memory-pattern probes for csplat_linear128: MODE 0 = coalesced float4 copy; MODE 1 = the kernel's pattern (16 x 16-byte
// loads per lane at 512-byte lane stride; 64 coalesced dword stores per lane); MODE 2 = loads only; MODE 3 = stores only
template <int MODE>
__global__ __launch_bounds__(256, 2) void k3(long long M, const float *A, float *out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r32 = lane & 31, h = lane >> 5;
    if (MODE == 0) {
        const long long n4 = M * 32;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
            reinterpret_cast<float4 *>(out)[i] = reinterpret_cast<const float4 *>(A)[i];
        return;
    }
    const long long ntile = M / 32;
    float acc = 0.f;
    for (long long tile = (long long)blockIdx.x * 4 + w; tile < ntile; tile += (long long)gridDim.x * 4) {
        float4 x[16];
        if (MODE != 3) {
            const float4 *ap = reinterpret_cast<const float4 *>(A + (tile * 32 + r32) * 128 + 64 * h);
#pragma unroll
            for (int q = 0; q < 16; q++) x[q] = ap[q];
        } else {
#pragma unroll
            for (int q = 0; q < 16; q++) x[q] = make_float4(tile, q, lane, 0.f);
        }
        if (MODE != 2) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const long long orow = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                out[orow * 128 + r32] = x[r].x; out[orow * 128 + 32 + r32] = x[r].y;
                out[orow * 128 + 64 + r32] = x[r].z; out[orow * 128 + 96 + r32] = x[r].w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; q++) acc += x[q].x + x[q].y + x[q].z + x[q].w;
        }
    }
    if (MODE == 2 && acc == 1.2345f) out[0] = acc;
}
template <int MODE>
void run3(long long M, int wgs) {
    float *A, *out; hipMalloc(&A, M * 512); hipMalloc(&out, M * 512); hipMemset(A, 0, M * 512);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k3<MODE><<<wgs, 256>>>(M, A, out);
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) k3<MODE><<<wgs, 256>>>(M, A, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    const double bytes = (MODE == 2 || MODE == 3) ? M * 512.0 : M * 1024.0;
    printf("mem probe MODE=%d wgs=%d: %.1f us  %.2f TB/s\n", MODE, wgs, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
    hipFree(A); hipFree(out);
}

// ---- bf16 32x32x16 rate (the 3-way bf16 split that emulates fp32 products needs 6 of these per fp32 32x32x16 block)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256) void kb(int iters, float *out) {
    f32x16 acc[NACC];
    for (int c = 0; c < NACC; c++) for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(threadIdx.x * 1e-3f + i); b[i] = (__bf16)(blockIdx.x * 1e-3f - i); }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int c = 0; c < NACC; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < NACC; c++) for (int r = 0; r < 16; r++) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void runb(int wgs, int iters) {
    float *out; hipMalloc(&out, (size_t)wgs * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kb<NACC><<<wgs, 256>>>(iters, out);
    hipEventRecord(e0);
    kb<NACC><<<wgs, 256>>>(iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double nmfma = (double)wgs * 4 * iters * 8.0 * NACC;
    printf("bf16 32x32x16 NACC=%d wgs=%d: %.3f ms  %.1f TFLOP/s  %.1f cycles/MFMA/SIMD at 2.4 GHz\n", NACC, wgs, ms,
           nmfma * 32768.0 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (nmfma / 1024.0));
    hipFree(out);
}
template <int NACC>
void run(int wgs, int threads, int iters) {
    float *out; long long *clk, h[2];
    hipMalloc(&out, (size_t)wgs * threads * 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<wgs, threads>>>(iters, out, clk);
    hipEventRecord(e0);
    k<NACC><<<wgs, threads>>>(iters, out, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double nmfma = (double)wgs * (threads / 64) * iters * 8.0 * NACC;
    double tf = nmfma * 4096.0 / (ms * 1e-3) / 1e12;
    printf("NACC=%d wgs=%d waves/wg=%d: %.3f ms  %.1f TFLOP/s  s_memtime %.0f ticks, wall %.0f ticks (100 MHz) -> %.0f MHz memtime; %.1f cycles/MFMA/SIMD at 2.4 GHz\n",
           NACC, wgs, threads / 64, ms, tf, (double)h[0], (double)h[1], (double)h[0] / ((double)h[1] / 100.0),
           ms * 1e-3 * 2.4e9 / (nmfma / 1024.0));
    hipFree(out); hipFree(clk);
}
int main() {
    run<4>(256, 256, 4096);    // 1 wave per SIMD, 4 independent accumulators
    run<4>(512, 256, 4096);    // 2 waves per SIMD
    run<1>(256, 256, 8192);    // dependent chain
    run<2>(256, 256, 8192);
    runb<4>(512, 4096); runb<1>(256, 8192);
    run3<0>(300000, 2048); run3<0>(300000, 512);
    run3<1>(300000, 512); run3<2>(300000, 512); run3<3>(300000, 512);
    run3<1>(300000, 2048); run3<2>(300000, 2048); run3<3>(300000, 2048);
    run2<2, true>(512, 64);
    run2<2, false>(512, 64);
    run2<4, true>(512, 64);
    run2<1, true>(512, 64);
    run2<2, true>(256, 64);
    return 0;
}
