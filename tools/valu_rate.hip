// tools/valu_rate.hip -- cycles per wave64 VALU instruction on this device, as a function of the waves resident per SIMD.
// Settles the constant bench.py prices K7's `issue_frac` with (4 cycles vs the guide's 2).
//
// One workgroup per CU (grid = #CUs), blockDim = 64 * 4 * W  => W waves on each of the CU's 4 SIMDs (W = 1, 2, 4).  Every wave
// runs a loop of 16 x CHAINS instructions of one kind on CHAINS independent register chains (CHAINS = 8: no dependent-issue
// stalls; CHAINS = 1: the dependent-chain latency).  Timed with s_memtime (= shader cycles) inside the kernel; the figure
// printed is   cycles of the slowest wave / (instructions one wave issued x W)
// -- what one wave64 instruction costs the SIMD.  2.0 = the SIMD-32 pipe saturated; ~4 with W = 1 is the single-wave issue
// limit of MI355X_MICROARCH.md.
// build: hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

enum Op { FMA, MUL, CNDMASK, DPP_ADD, EXP, PK_FMA, PK_MUL, MOV, PERMLANE32_SWAP, ADD_DPP_ROW_SHR, LDS_READ,
          CNDMASK_SGPR, CNDMASK_NODEP, CMP_VCC, CMP_CNDMASK, MAX, ADD_SGPR, MOV_DPP, READLANE, BPERMUTE, MED3, FMA_CONST, PERMLANE16_SWAP,
          ADD_F32, SUB_F32, FMAC, MUL_DPP, FMA_NEG, ADD_U32, AND_B32, LSHL, CVT_F32_U32, MAD_U32_U24, BFE, CMP_E64, RCP, LDEXP, PK_ADD, MIN, MBCNT, ADD3, LSHL_ADD, CNDMASK_VCC_SMOV, N_OPS };
static const char *OP_NAME[N_OPS] = {"v_fma_f32", "v_mul_f32", "v_cndmask_b32 (vcc)", "v_add_f32 dpp quad_perm", "v_exp_f32",
                                     "v_pk_fma_f32 (2 fp32 results)", "v_pk_mul_f32 (2 fp32 results)", "v_mov_b32",
                                     "v_permlane32_swap", "v_add_f32 dpp row_shr:1", "ds_read_b32 (for scale)",
                                     "v_cndmask_b32 e64 (sgpr pair)", "v_cndmask_b32 vcc, dst != src", "v_cmp_lt_f32 -> vcc",
                                     "v_cmp_lt_f32 + v_cndmask (pair)", "v_max_f32", "v_add_f32 with sgpr operand",
                                     "v_mov_b32 dpp row_shr:1", "v_readlane_b32", "ds_bpermute_b32", "v_med3_f32",
                                     "v_fma_f32 inline const", "v_permlane16_swap",
                                     "v_add_f32", "v_sub_f32", "v_fmac_f32", "v_mul_f32 dpp row_shr:1", "v_fma_f32 neg/abs modifiers", "v_add_u32", "v_and_b32",
                                     "v_lshlrev_b32", "v_cvt_f32_u32", "v_mad_u32_u24", "v_bfe_u32", "v_cmp_lt_f32 e64 -> sgpr pair", "v_rcp_f32",
                                     "v_ldexp_f32", "v_pk_add_f32 (2 results)", "v_min_f32", "v_mbcnt_lo_u32_b32", "v_add3_u32", "v_lshl_add_u32",
                                     "v_cndmask vcc after s_mov vcc"};

template <int OP, int CHAINS>
__global__ void k_rate(int iters, float *out, long long *cyc) {
    __shared__ float lds[4096];
    for (int t = threadIdx.x; t < 4096; t += blockDim.x) lds[t] = t * 1e-6f;
    __syncthreads();
    float r[CHAINS], q[CHAINS];
    for (int c = 0; c < CHAINS; c++) { r[c] = threadIdx.x * 1e-3f + c; q[c] = 1.0f + c * 1e-3f; }
    const float a = 1.0000001f, b = 1e-9f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 pa = {a, a}, pb = {b, b};
    unsigned long long smask = 0x5555aaaa5555aaaaull ^ (unsigned long long)iters;
    float sval = 1e-9f * iters;
    int baddr = ((threadIdx.x + 1) & 63) * 4;
    if (OP == CNDMASK_VCC_SMOV) asm volatile("s_mov_b64 vcc, %0" :: "s"(smask) : "vcc");
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[c]) : "v"(a), "v"(b));
                if (OP == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[c]) : "v"(a));
                if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[c]) : "v"(q[c]) : );
                if (OP == DPP_ADD) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[c]));
                if (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(r[c]));
                if (OP == PK_FMA) {
                    f2 v = {r[c], q[c]};
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(pa), "v"(pb));
                    r[c] = v.x; q[c] = v.y;
                }
                if (OP == PK_MUL) {
                    f2 v = {r[c], q[c]};
                    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(pa));
                    r[c] = v.x; q[c] = v.y;
                }
                if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(r[c]) : "v"(q[c]));
                if (OP == PERMLANE32_SWAP) {
                    auto pr = __builtin_amdgcn_permlane32_swap(__float_as_uint(r[c]), __float_as_uint(q[c]), false, false);
                    r[c] = __uint_as_float(pr[0]); q[c] = __uint_as_float(pr[1]);
                    asm volatile("" : "+v"(r[c]), "+v"(q[c]));
                }
                if (OP == ADD_DPP_ROW_SHR) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[c]));
                if (OP == CNDMASK_SGPR) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[c]) : "v"(q[c]), "s"(smask));
                if (OP == CNDMASK_NODEP) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r[c]) : "v"(q[c]), "v"(a));
                if (OP == CMP_VCC) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(r[c]), "v"(q[c]) : "vcc");
                if (OP == CMP_CNDMASK) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[c]) : "v"(q[c]) : "vcc");
                if (OP == MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[c]) : "v"(q[c]));
                if (OP == ADD_SGPR) asm volatile("v_add_f32 %0, %1, %0" : "+v"(r[c]) : "s"(sval));
                if (OP == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[c]) : "v"(q[c]));
                if (OP == READLANE) { int sv; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sv) : "v"(r[c])); asm volatile("" :: "s"(sv)); }
                if (OP == BPERMUTE) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(r[c]) : "v"(baddr));
                if (OP == MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r[c]) : "v"(q[c]), "v"(a));
                if (OP == FMA_CONST) asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(r[c]));
                if (OP == PERMLANE16_SWAP) {
                    auto pr = __builtin_amdgcn_permlane16_swap(__float_as_uint(r[c]), __float_as_uint(q[c]), false, false);
                    r[c] = __uint_as_float(pr[0]); q[c] = __uint_as_float(pr[1]);
                    asm volatile("" : "+v"(r[c]), "+v"(q[c]));
                }
                if (OP == ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[c]) : "v"(q[c]));
                if (OP == SUB_F32) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[c]) : "v"(q[c]));
                if (OP == FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r[c]) : "v"(q[c]), "v"(a));
                if (OP == MUL_DPP) asm volatile("v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[c]));
                if (OP == FMA_NEG) asm volatile("v_fma_f32 %0, -%0, |%1|, -%2" : "+v"(r[c]) : "v"(a), "v"(b));
                if (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[c]) : "v"(q[c]));
                if (OP == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[c]) : "v"(q[c]));
                if (OP == LSHL) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r[c]));
                if (OP == CVT_F32_U32) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(r[c]));
                if (OP == MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r[c]) : "v"(q[c]), "v"(a));
                if (OP == BFE) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(r[c]));
                if (OP == CMP_E64) { unsigned long long m; asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(r[c]), "v"(q[c])); asm volatile("" :: "s"(m)); }
                if (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[c]));
                if (OP == LDEXP) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(r[c]) : "v"(baddr));
                if (OP == PK_ADD) {
                    f2 v = {r[c], q[c]};
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"(pb));
                    r[c] = v.x; q[c] = v.y;
                }
                if (OP == MIN) asm volatile("v_min_f32 %0, %0, %1" : "+v"(r[c]) : "v"(q[c]));
                if (OP == MBCNT) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(r[c]) : "s"((int)smask));
                if (OP == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r[c]) : "v"(q[c]), "v"(a));
                if (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(r[c]) : "v"(q[c]));
                if (OP == CNDMASK_VCC_SMOV) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[c]) : "v"(q[c]));
                if (OP == LDS_READ) {
                    int addr = (__float_as_int(r[c]) & 0x3ffc);
                    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r[c]) : "v"(addr));
                }
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < CHAINS; c++) s += r[c] + q[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        // slowest wave of the workgroup
        atomicMax((unsigned long long *)&cyc[blockIdx.x], (unsigned long long)(t1 - t0));
    }
}

template <int OP, int CHAINS>
static double run(int waves_per_simd, int cus, int iters, float *d_out, long long *d_cyc) {
    // W <= 4: one workgroup of 64*4*W threads per CU; W = 8: two 1024-thread workgroups per CU (a CU holds 32 waves)
    int wgs = waves_per_simd > 4 ? 2 * cus : cus;
    int threads = 64 * 4 * (waves_per_simd > 4 ? waves_per_simd / 2 : waves_per_simd);
    hipMemset(d_cyc, 0, sizeof(long long) * wgs);
    k_rate<OP, CHAINS><<<wgs, threads>>>(iters / 8, d_out, d_cyc);      // warm-up
    hipMemset(d_cyc, 0, sizeof(long long) * wgs);
    k_rate<OP, CHAINS><<<wgs, threads>>>(iters, d_out, d_cyc);
    hipDeviceSynchronize();
    std::vector<long long> h(wgs);
    hipMemcpy(h.data(), d_cyc, sizeof(long long) * wgs, hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < wgs; i++) avg += (double)h[i];
    avg /= wgs;
    // s_memtime counts at a fixed 100 MHz reference on some parts; calibrated below against clock64
    double insts = (double)iters * 16 * CHAINS;
    return avg / (insts * waves_per_simd);
}

__global__ void k_cal(long long *o) {
    long long m0 = __builtin_amdgcn_s_memtime(), c0 = clock64(), w0 = wall_clock64();
    float x = threadIdx.x;
    for (int i = 0; i < 200000; i++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
    long long m1 = __builtin_amdgcn_s_memtime(), c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { o[0] = m1 - m0; o[1] = c1 - c0; o[2] = w1 - w0; o[3] = (long long)x; }
}

template <int OP>
static void row(int cus, float *d_out, long long *d_cyc, double scale) {
    const int iters = 2000;
    printf("| %-32s |", OP_NAME[OP]);
    for (int w : {1, 2, 4}) printf(" %5.2f |", scale * run<OP, 8>(w, cus, iters, d_out, d_cyc));
    printf(" %5.2f |\n", scale * run<OP, 1>(1, cus, iters, d_out, d_cyc));
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    float *d_out; long long *d_cyc;
    hipMalloc(&d_out, sizeof(float) * cus * 2048);
    hipMalloc(&d_cyc, sizeof(long long) * (2 * cus + 8));
    k_cal<<<1, 64>>>(d_cyc); hipDeviceSynchronize();
    long long c[4]; hipMemcpy(c, d_cyc, sizeof(c), hipMemcpyDeviceToHost);
    int wall_khz = 0; hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    double secs = (double)c[2] / (wall_khz * 1e3);
    printf("device %s, %d CUs; calibration (one wave, 200k dependent v_fma): s_memtime ticks %lld, clock64 %lld, wall %lld ticks @ %d kHz = %.3f ms\n",
           p.gcnArchName, cus, c[0], c[1], c[2], wall_khz, secs * 1e3);
    double mem_hz = c[0] / secs, clk_hz = c[1] / secs;
    printf("s_memtime rate %.1f MHz, clock64 rate %.1f MHz; dependent v_fma: %.2f ns each\n", mem_hz / 1e6, clk_hz / 1e6, secs / 200000 * 1e9);
    // s_memtime ticks ARE shader cycles on gfx950 (MI355X_MICROARCH.md, cycle-constants table); the rate above is the clock it ran at
    double scale = 1.0;
    printf("cycles per wave64 instruction per SIMD; columns = waves resident per SIMD (8 independent chains per wave); last column = ONE wave, ONE dependent chain (latency)\n");
    printf("| instruction                      |  W=1  |  W=2  |  W=4  | dep.  |\n|---|---|---|---|---|\n");
    row<FMA>(cus, d_out, d_cyc, scale);
    row<MUL>(cus, d_out, d_cyc, scale);
    row<MOV>(cus, d_out, d_cyc, scale);
    row<CNDMASK>(cus, d_out, d_cyc, scale);
    row<DPP_ADD>(cus, d_out, d_cyc, scale);
    row<ADD_DPP_ROW_SHR>(cus, d_out, d_cyc, scale);
    row<PERMLANE32_SWAP>(cus, d_out, d_cyc, scale);
    row<EXP>(cus, d_out, d_cyc, scale);
    row<PK_FMA>(cus, d_out, d_cyc, scale);
    row<PK_MUL>(cus, d_out, d_cyc, scale);
    row<LDS_READ>(cus, d_out, d_cyc, scale);
    row<CNDMASK_SGPR>(cus, d_out, d_cyc, scale);
    row<CNDMASK_NODEP>(cus, d_out, d_cyc, scale);
    row<CMP_VCC>(cus, d_out, d_cyc, scale);
    row<CMP_CNDMASK>(cus, d_out, d_cyc, scale);
    row<MAX>(cus, d_out, d_cyc, scale);
    row<MED3>(cus, d_out, d_cyc, scale);
    row<ADD_SGPR>(cus, d_out, d_cyc, scale);
    row<FMA_CONST>(cus, d_out, d_cyc, scale);
    row<MOV_DPP>(cus, d_out, d_cyc, scale);
    row<PERMLANE16_SWAP>(cus, d_out, d_cyc, scale);
    row<READLANE>(cus, d_out, d_cyc, scale);
    row<BPERMUTE>(cus, d_out, d_cyc, scale);
    printf("(v_cmp + v_cndmask row: cycles per PAIR)\n");
    row<ADD_F32>(cus, d_out, d_cyc, scale); row<SUB_F32>(cus, d_out, d_cyc, scale); row<FMAC>(cus, d_out, d_cyc, scale);
    row<MUL_DPP>(cus, d_out, d_cyc, scale); row<FMA_NEG>(cus, d_out, d_cyc, scale); row<MIN>(cus, d_out, d_cyc, scale);
    row<PK_ADD>(cus, d_out, d_cyc, scale); row<RCP>(cus, d_out, d_cyc, scale); row<LDEXP>(cus, d_out, d_cyc, scale);
    row<ADD_U32>(cus, d_out, d_cyc, scale); row<AND_B32>(cus, d_out, d_cyc, scale); row<LSHL>(cus, d_out, d_cyc, scale);
    row<CVT_F32_U32>(cus, d_out, d_cyc, scale); row<MAD_U32_U24>(cus, d_out, d_cyc, scale); row<BFE>(cus, d_out, d_cyc, scale);
    row<ADD3>(cus, d_out, d_cyc, scale); row<LSHL_ADD>(cus, d_out, d_cyc, scale); row<MBCNT>(cus, d_out, d_cyc, scale);
    row<CMP_E64>(cus, d_out, d_cyc, scale); row<CNDMASK_VCC_SMOV>(cus, d_out, d_cyc, scale);
    return 0;
}
