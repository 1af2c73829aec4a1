// csplat_edge_mlp.hip -- the WHOLE edge MLP of an InteractionNetwork layer in one launch (rollout / inference, BASELINE configs[3]):
//
//   msg[e] = LayerNorm( W3 relu( W2 relu( alpha * We e0[e] + b0 + xa[dst[e]] + xb[src[e]] ) + b2 ) + b3 )
//
// = /root/reference/meshnet/graph_network.py:178-199 (`message`: edge_fn(cat[x_i, x_j, e]) with the first Linear cut into its three
// column blocks, the x_i / x_j blocks applied at node level: xa = x Wi^T (+ bias rides in b0 here), xb = x Wj^T) for edge features
// alpha * e0 (every layer doubles its edge features, SURVEY F7: alpha = 2^l).
//
// Rounds 1-4 ran this as three csplat_linear128 launches, each a full [E,128] HBM round trip (307 MB per launch at E = 300k: 75-97 us
// each, memory-paced).  Here the two inner [rows,128] activations never leave the registers: algorithmic traffic per layer drops from
// 3 x 307 MB to 154 MB in + 154 MB out (+ the L2-resident gathers), and the kernel is paced by its MFMAs.
//
// Design (gfx950).  One persistent 8-wave workgroup per CU; a wave owns 32 edge rows of a 256-row round and carries them through the
// three layers.  The products run on v_mfma_f32_32x32x16_bf16 with both operands cut into three bf16 pieces (the six partial products
// that matter, fp32 accumulation: fp32-level accuracy at 6/16 of the fp32-MFMA time -- csplat_gemm.hip, B3) and are formed TRANSPOSED:
// MFMA A operand = weight rows (output features), B operand = the lane's edge row.  A lane (row n = lane & 31, half h = lane >> 5)
// then ends a layer holding 64 features of ITS OWN row -- accumulator register r of column tile c <-> feature 32c + 8(r >> 2) + 4h +
// (r & 3) -- which is exactly the B operand of the next layer once that layer's contraction index is permuted to match: step st of the
// next layer contracts over the eight features held in registers 8(st & 1) .. +7 of tile st >> 1.  The permutation is applied to the
// WEIGHTS, once: csplat_gnn_edge_mlp3_pack writes, per layer, the three bf16 pieces of the weight matrix as the exact byte image the
// kernel wants in LDS (rows padded to 272 B: conflict-free 16-byte operand reads), 104,448 B per layer.
// What sank round 2's attempt at this kernel ("k_edge_mlp3": 349 us against 253-276 for three launches) was
// re-staging: cutting fp32 weights into bf16 pieces 3 x per round cost 17 k cycles per layer.  With the image pre-cut, re-staging a
// layer is a straight 102 KiB copy L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: 13 wave-instructions per wave, no VGPRs, ~2.5 k
// cycles) between two barriers; the next round's edge rows are requested under layer 3's MFMAs.
// Layer 1's gathers and bias are loaded straight into the accumulators (alpha is a power of two: start at (b0 + xa + xb) / alpha and
// scale the finished sum); LayerNorm statistics cross the lane pair (n, 0) / (n, 1) with one v_permlane32_swap each.
#include "csplat_common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int EM_N = 128;                 // layer width
constexpr int EM_STRIDE = 136;            // bf16 elements per image row (272 B)
constexpr int EM_PIECE = EM_N * EM_STRIDE;
constexpr size_t EM_LAYER_BYTES = (size_t)3 * EM_PIECE * 2;      // 104,448 = 102 KiB: three bf16 pieces of one 128 x 128 weight
constexpr int EM_CHUNKS = (int)(EM_LAYER_BYTES / 1024);          // 1 KiB LDS-DMA pieces per layer
static_assert(EM_LAYER_BYTES % 1024 == 0, "the layer image is copied in whole 1 KiB wave-instructions");
constexpr int EM_WAVES = 8, EM_ROWS = 32 * EM_WAVES;             // rows per round
constexpr int EM_SCR = 32 * EM_N * 4;                            // a wave's transposition scratch: its 32 x 128 fp32 tile
constexpr size_t EM_LDS_BYTES = (size_t)EM_WAVES * EM_SCR > EM_LAYER_BYTES ? (size_t)EM_WAVES * EM_SCR : EM_LAYER_BYTES;

// position pos = 64h + 8st + i of an image row (the element lane-half h feeds into step st as operand element i) <-> source column
__host__ __device__ inline int em_src_col(int layer, int pos) {
    if (layer == 0) return pos;           // layer 1 contracts over the edge row as it lies in memory: half h = columns 64h .. 64h + 63
    const int h = pos >> 6, st = (pos >> 3) & 7, i = pos & 7;
    return 32 * (st >> 1) + 16 * (st & 1) + 8 * (i >> 2) + 4 * h + (i & 3);
}

__global__ __launch_bounds__(EM_STRIDE) void k_edge_mlp3_pack(const float *__restrict__ W0, int ld0, const float *__restrict__ W1, int ld1,
                                                               const float *__restrict__ W2, int ld2, __bf16 *__restrict__ img) {
    const int l = blockIdx.y, j = blockIdx.x, pos = threadIdx.x;
    const float *W = l == 0 ? W0 : (l == 1 ? W1 : W2);
    const int ld = l == 0 ? ld0 : (l == 1 ? ld1 : ld2);
    const float x = pos < EM_N ? W[(size_t)j * ld + em_src_col(l, pos)] : 0.f;      // (pad columns: zeros)
    const __bf16 p1 = (__bf16)x;
    const float r1 = x - (float)p1;
    const __bf16 p2 = (__bf16)r1;
    const __bf16 p3 = (__bf16)(r1 - (float)p2);
    __bf16 *row = img + (size_t)l * 3 * EM_PIECE + (size_t)j * EM_STRIDE + pos;
    row[0] = p1; row[EM_PIECE] = p2; row[2 * EM_PIECE] = p3;
}

// a ^ b formed where it is written: a volatile statement is not hoisted out of the round loop (the compiler otherwise precomputes the ~100
// round-invariant lane addresses of the transposition scratch and spills them all)
__device__ __forceinline__ int xor_here(int a, int b) {
    int r;
    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float pair_sum(float v) {      // v of lane (n, 0) + v of lane (n, 1), in both lanes
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
    return __int_as_float(sw[0]) + __int_as_float(sw[1]);
}

__global__ __launch_bounds__(64 * EM_WAVES) void k_edge_mlp3(int64_t M, const float *__restrict__ e0, float alpha, float inv_alpha,
                                                             const float *__restrict__ xa, const int64_t *__restrict__ ia,
                                                             const float *__restrict__ xb, const int64_t *__restrict__ ib,
                                                             const char *__restrict__ img, const float *__restrict__ b0,
                                                             const float *__restrict__ b1, const float *__restrict__ b2,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                             float *__restrict__ out, int dbg, unsigned long long *__restrict__ stamps) {
    extern __shared__ char s_img[];       // ONE LDS object: the current layer's image, or (between rounds) the waves' transposition scratch
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r32 = lane & 31, h = lane >> 5;
    const int64_t nround = (M + EM_ROWS - 1) / EM_ROWS;
    int zs = 0, zv = 0, r32v = r32, lanev = lane, hv = h;      // (opaque zeros and the lane ids formed with them per round, see the round loop)
    // (measurement hook, csplat_debug_stamps / tools/edge_mlp3_stamps.py: wave 0 leaves s_memtime at the phase boundaries of its first rounds)
    int stamp_at = 0;
    auto stamp = [&]() {
        if (stamps && w == 0 && stamp_at < 64) {
            const unsigned long long t = __builtin_readcyclecounter();
            if (lane == 0) stamps[(size_t)blockIdx.x * 64 + stamp_at] = t;
            stamp_at++;
        }
    };

    // layer `layer`'s image L2 -> LDS: chunk c (1 KiB) by wave c % 8, lane l moving bytes 16 l .. 16 l + 15 (the LDS image is
    // byte-identical to the global one, padding included, so the lane-linear destination of an LDS-DMA instruction is the layout)
    const unsigned voff = lane * 16;
    bool staged_once = false;
    auto stage = [&](int layer) {
        if ((dbg & 1) && staged_once) return;        // (timing experiment: no re-staging -- results wrong)
        staged_once = true;
        // (the base stays on the scalar unit and is laundered per call: left alone, the compiler forms the 39 per-lane 64-bit source
        //  addresses of the three layers once, outside the round loop, and spills them -- 78 registers)
        const char *sb = img + (size_t)layer * EM_LAYER_BYTES + (size_t)w * 1024;
        asm volatile("" : "+s"(sb));
#pragma unroll
        for (int k = 0; k < (EM_CHUNKS + EM_WAVES - 1) / EM_WAVES; k++) {
            const int c = w + EM_WAVES * k;
            if (c < EM_CHUNKS)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sb + (size_t)k * (EM_WAVES * 1024) + voff),
                                                 (__attribute__((address_space(3))) void *)(s_img + c * 1024), 16, 0, 0);
        }
    };
    // the lane's 64 columns of a [.][128] row, as 16 float4: group g = 4c + q <-> columns 32c + 8q + 4h .. + 3
    auto col_of = [&](int g) { return 32 * (g >> 2) + 8 * (g & 3) + 4 * hv; };

    // one layer's products: acc[c] += W_c (pieces, from LDS) x X (this lane's 64 contraction values, cut into pieces on the fly)
    // refill != nullptr: X[8st .. 8st + 7] is re-loaded from refill[2st], refill[2st + 1] as soon as step st has cut its pieces (the next
    // round's row, fetched into the registers this round frees: no second register set)
    auto products = [&](float (&X)[64], f32x16 (&acc)[4], const float4 *refill) __attribute__((always_inline)) {
        const __bf16 *wl = reinterpret_cast<const __bf16 *>(s_img) + (size_t)r32 * EM_STRIDE + 64 * h;
        auto ldw = [&](int p, int c, int st) {
            return *reinterpret_cast<const bf16x8 *>(wl + ((size_t)p * EM_N + 32 * c) * EM_STRIDE + 8 * st);
        };
        if (dbg & 2) return;                          // (timing experiment: no products)
        bf16x8 wc[3], wn[3];
#pragma unroll
        for (int p = 0; p < 3; p++) wc[p] = ldw(p, 0, 0);
#pragma unroll
        for (int st = 0; st < 8; st++) {
            bf16x8 a1, a2, a3;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float x = X[8 * st + j];
                a1[j] = (__bf16)x;
                const float r1 = x - (float)a1[j];
                a2[j] = (__bf16)r1;
                a3[j] = (__bf16)(r1 - (float)a2[j]);
            }
            if (refill) {
                const float4 t0 = refill[2 * st], t1 = refill[2 * st + 1];
                X[8 * st] = t0.x; X[8 * st + 1] = t0.y; X[8 * st + 2] = t0.z; X[8 * st + 3] = t0.w;
                X[8 * st + 4] = t1.x; X[8 * st + 5] = t1.y; X[8 * st + 6] = t1.z; X[8 * st + 7] = t1.w;
            }
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int cn = (c + 1) & 3, sn = c == 3 ? st + 1 : st;
                if (sn < 8) {
#pragma unroll
                    for (int p = 0; p < 3; p++) wn[p] = ldw(p, cn, sn);
                }
                __builtin_amdgcn_sched_barrier(0);      // next operands requested before this group's 6 MFMAs (192 cycles)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[0], a3, acc[c], 0, 0, 0);      // small terms first
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[2], a1, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[1], a2, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[0], a2, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[1], a1, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[0], a1, acc[c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < 3; p++) wc[p] = wn[p];
            }
        }
    };
    auto init_bias = [&](const float *__restrict__ b, f32x16 (&acc)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const float4 t = *reinterpret_cast<const float4 *>(b + zs + col_of(g));
            acc[g >> 2][4 * (g & 3)] = t.x; acc[g >> 2][4 * (g & 3) + 1] = t.y; acc[g >> 2][4 * (g & 3) + 2] = t.z; acc[g >> 2][4 * (g & 3) + 3] = t.w;
        }
    };
    // ---- the wave's 16 KB transposition scratch (its slice of the LDS the images occupy during the layers; used only between a round's
    // last products and the next round's first image copy).  Global memory is touched in whole rows -- an instruction moves tile rows
    // 2i and 2i + 1, lane (r32, h) the 16 bytes at granule r32 of row 2i + h: 8 cache lines per instruction -- while the MFMA operands
    // want lane = row: a lane that reads its own row's 16 bytes touches 64 lines per instruction, and the texture-address unit prices a
    // memory instruction per line (measured, tools/edge_mlp3_stamps.py: the gathers, row loads and row stores of the first version took
    // 52 k of a round's 131 k cycles).  Logical granule g (4 floats) of tile row n lives at physical granule g ^ n of the row: both
    // access patterns are then conflict-free (16 lanes of an LDS lane group = 16 distinct n mod 16, or 16 distinct granules of one row).
    char *scr = s_img + w * EM_SCR;
    auto scr_at = [&](int g) { return reinterpret_cast<float4 *>(scr + ((r32v * 32 + xor_here(g, r32v)) << 4)); };      // lane = row view
    // rows -> scratch by LDS-DMA; grow(i) = the global row this lane's half of instruction i reads (tile row 2i + h)
    auto dma_rows = [&](const float *__restrict__ base, auto grow) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int trow = 2 * i + hv;
            const float *src = base + (int64_t)grow(i) * EM_N + 4 * xor_here(trow, r32v);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(scr + i * 1024), 16, 0, 0);
        }
    };

    float X[64];                              // the lane's contraction values of the coming layer
    bool have_x = false;
    f32x16 acc[4];
    // a round's inputs: the 32 edge rows of the wave (-> X), and the layer-1 accumulators' start (b0 + xa[dst] + xb[src]) / alpha
    auto load_inputs = [&](int64_t round) __attribute__((always_inline)) {
        const int64_t base_row = (round * EM_WAVES + w) * 32;
        const int64_t row = base_row + r32 < M ? base_row + r32 : M - 1;       // rows past M: clamped loads, masked stores
        const int ja = (int)ia[row], jb = (int)ib[row];
        if (!have_x) {                        // (first round of the workgroup; later rounds find X prefetched under the previous round's layer 3)
            dma_rows(e0, [&](int i) { const int64_t r = base_row + 2 * i + hv; return r < M ? r : M - 1; });
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 16; q++) { const float4 t = *scr_at(16 * hv + q); X[4 * q] = t.x; X[4 * q + 1] = t.y; X[4 * q + 2] = t.z; X[4 * q + 3] = t.w; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        // the x_i block's rows: destination nodes, scattered -> through the scratch as whole rows
        dma_rows(xa, [&](int i) { const int lo = __builtin_amdgcn_readlane(ja, 2 * i), hi = __builtin_amdgcn_readlane(ja, 2 * i + 1); return hv ? hi : lo; });
        // bias + the x_j block's rows, straight into registers: with the edge list in PyG's coalesced (source-major) order the 32 rows of a
        // tile share 1-3 source nodes, i.e. 1-3 cache lines per instruction (any other order is as correct and slower)
        {   // (four batches of four column groups: 8 float4 in flight each -- all 32 at once would cost 128 registers beside X and acc)
            const float *pb = xb + (size_t)jb * EM_N;
#pragma unroll
            for (int bq = 0; bq < 4; bq++) {
                float4 tb[4], tc[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    tb[u] = *reinterpret_cast<const float4 *>(b0 + zs + col_of(4 * bq + u));
                    tc[u] = *reinterpret_cast<const float4 *>(pb + col_of(4 * bq + u));
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    acc[bq][4 * u] = (tb[u].x + tc[u].x) * inv_alpha; acc[bq][4 * u + 1] = (tb[u].y + tc[u].y) * inv_alpha;
                    acc[bq][4 * u + 2] = (tb[u].z + tc[u].z) * inv_alpha; acc[bq][4 * u + 3] = (tb[u].w + tc[u].w) * inv_alpha;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const float4 t = *scr_at(8 * (g >> 2) + 2 * (g & 3) + hv);
            const int c = g >> 2, r = 4 * (g & 3);
            acc[c][r] += t.x * inv_alpha; acc[c][r + 1] += t.y * inv_alpha; acc[c][r + 2] += t.z * inv_alpha; acc[c][r + 3] += t.w * inv_alpha;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    for (int64_t round = blockIdx.x; round < nround; round += gridDim.x) {
        // opaque zeros, renewed per round: everything addressed through them stays INSIDE the loop.  Left alone, the compiler hoists the
        // round-invariant loads (bias, gamma, beta: 80 float4) and the ~100 per-lane addresses out of the loop and spills them all
        asm volatile("s_mov_b32 %0, 0" : "=s"(zs));
        asm volatile("v_mov_b32 %0, 0" : "=v"(zv));
        r32v = r32 + zv; lanev = lane + zv; hv = h + zv;
        stamp();                              // 0: round start
        load_inputs(round);                   // (the scratch is free: first round, or the previous round's stores have read it back)
        stamp();                              // 1: inputs in registers
        // ---------------- layer 1: alpha * We e0 + b0 + xa[dst] + xb[src], ReLU
        __syncthreads();                      // (every wave is done with its scratch: the image may land)
        stage(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                      // (every wave's share of the image has landed)
        stamp();                              // 2: layer-1 image in
        products(X, acc, nullptr);
        stamp();                              // (layer-1 products done)
#pragma unroll
        for (int k = 0; k < 64; k++) X[k] = fmaxf(alpha * acc[k >> 4][k & 15], 0.f);
        // ---------------- layer 2
        __syncthreads();
        stamp();                              // 3: every wave's layer-1 products done
        stage(1);
        init_bias(b1, acc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stamp();                              // 4: layer-2 image in
        products(X, acc, nullptr);
        stamp();                              // 5
#pragma unroll
        for (int k = 0; k < 64; k++) X[k] = fmaxf(acc[k >> 4][k & 15], 0.f);
        // ---------------- layer 3 + LayerNorm
        __syncthreads();
        stamp();                              // 6: every wave's layer-2 products done
        stage(2);
        init_bias(b2, acc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stamp();                              // 7: layer-3 image in
        // the NEXT round's edge row is fetched during these products, chunk by chunk into the registers the steps free (lane = row: 64
        // cache lines per instruction, which the texture-address unit has all of this layer's MFMAs to work off)
        const bool more = round + gridDim.x < nround;
        {
            const int64_t nrow = ((round + gridDim.x) * EM_WAVES + w) * 32 + r32v;
            const float4 *ap = reinterpret_cast<const float4 *>(e0 + (nrow < M ? nrow : M - 1) * EM_N + 64 * hv);
            products(X, acc, ap);             // (the last round re-reads a clamped row for nothing: no second copy of the loop)
        }
        stamp();                              // 8
        have_x = more;
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 64; k++) sum += acc[k >> 4][k & 15];
        const float mean = pair_sum(sum) * (1.f / EM_N);
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < 64; k++) { const float d = acc[k >> 4][k & 15] - mean; acc[k >> 4][k & 15] = d; sq += d * d; }
        const float rstd = rsqrtf(pair_sum(sq) * (1.f / EM_N) + eps);
        __syncthreads();                      // (every wave is done with the layer-3 image: the scratch may be written)
        stamp();                              // 9: every wave's layer-3 products + LayerNorm sums done
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const int col = col_of(g);
            const float4 ga = *reinterpret_cast<const float4 *>(gamma + zs + col), be = *reinterpret_cast<const float4 *>(beta + zs + col);
            const int c = g >> 2, r = 4 * (g & 3);
            *scr_at(8 * c + 2 * (g & 3) + hv) = make_float4(acc[c][r] * rstd * ga.x + be.x, acc[c][r + 1] * rstd * ga.y + be.y,
                                                           acc[c][r + 2] * rstd * ga.z + be.z, acc[c][r + 3] * rstd * ga.w + be.w);
        }
        {   // whole rows back out: instruction i stores tile rows 2i, 2i + 1
            const int64_t base_row = (round * EM_WAVES + w) * 32;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int trow = 2 * i + hv;
                const float4 v = *reinterpret_cast<const float4 *>(scr + i * 1024 + lanev * 16);
                if (base_row + trow < M) *reinterpret_cast<float4 *>(out + (base_row + trow) * EM_N + 4 * xor_here(trow, r32v)) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp();                              // 10: rows out issued
    }
}

}  // namespace

extern "C" size_t csplat_gnn_edge_mlp3_image_bytes(void) { return 3 * EM_LAYER_BYTES; }

extern "C" int csplat_gnn_edge_mlp3_pack(void *stream, const float *W0, int ld0, const float *W1, int ld1, const float *W2, int ld2, void *image) {
    CSPLAT_REQUIRE(W0 && W1 && W2 && image && ld0 >= EM_N && ld1 >= EM_N && ld2 >= EM_N, "csplat_gnn_edge_mlp3_pack: bad arguments");
    CSPLAT_REQUIRE(((uintptr_t)image & 15u) == 0, "csplat_gnn_edge_mlp3_pack: the image must be 16-byte aligned");
    k_edge_mlp3_pack<<<dim3(EM_N, 3), EM_STRIDE, 0, (hipStream_t)stream>>>(W0, ld0, W1, ld1, W2, ld2, (__bf16 *)image);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int csplat_gnn_edge_mlp3(void *stream, int64_t E, const float *e0, float alpha, const float *xa, const int64_t *index_a,
                                    const float *xb, const int64_t *index_b, const void *image, const float *b0, const float *b1,
                                    const float *b2, const float *ln_gamma, const float *ln_beta, float ln_eps, float *out) {
    CSPLAT_REQUIRE(E >= 0 && (E == 0 || (e0 && xa && index_a && xb && index_b && image && b0 && b1 && b2 && ln_gamma && ln_beta && out)),
                   "csplat_gnn_edge_mlp3: bad arguments");
    if (E == 0) return 0;
    const uintptr_t al = (uintptr_t)e0 | (uintptr_t)xa | (uintptr_t)xb | (uintptr_t)image | (uintptr_t)b0 | (uintptr_t)b1 | (uintptr_t)b2 |
                         (uintptr_t)ln_gamma | (uintptr_t)ln_beta | (uintptr_t)out;
    CSPLAT_REQUIRE((al & 15u) == 0, "csplat_gnn_edge_mlp3: operands must be 16-byte aligned");
    CSPLAT_REQUIRE(out != e0, "csplat_gnn_edge_mlp3: out must not alias e0 (rows are re-read by later rounds' prefetch)");
    int ex = 0;
    const float m = frexpf(alpha, &ex);
    CSPLAT_REQUIRE(alpha > 0.f && m == 0.5f, "csplat_gnn_edge_mlp3: alpha must be a power of two (the edge scale 2^l)");
    static int s_ok = -1;
    if (s_ok < 0) {
        s_ok = hipFuncSetAttribute((const void *)k_edge_mlp3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)EM_LDS_BYTES) == hipSuccess;
        (void)hipGetLastError();
    }
    CSPLAT_REQUIRE(s_ok, "csplat_gnn_edge_mlp3: 128 KB of dynamic LDS refused by the runtime");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_GNN, s);
    const int64_t nround = (E + EM_ROWS - 1) / EM_ROWS;
    const int grid = (int)(nround < 256 ? nround : 256);      // persistent: one 8-wave workgroup per CU
    static const int dbg = getenv("CSPLAT_EM_DEBUG") ? atoi(getenv("CSPLAT_EM_DEBUG")) : 0;      // (timing experiments only)
    k_edge_mlp3<<<grid, 64 * EM_WAVES, EM_LDS_BYTES, s>>>(E, e0, alpha, 1.0f / alpha, xa, index_a, xb, index_b, (const char *)image, b0, b1, b2,
                                                            ln_gamma, ln_beta, ln_eps, out, dbg,
                                                            csplat_stamp_buffer((size_t)grid * 64));
    LAUNCH_CHECK();
    return 0;
}
