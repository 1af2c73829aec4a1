// csplat_sort.hip -- device-wide inclusive scan (K2) and stable LSD radix sort of (u64,u32) pairs (K4).
//
// Replaces cub::DeviceScan::InclusiveSum / cub::DeviceRadixSort::SortPairs as used by the upstream
// rasterizer behind gaussian_renderer/__init__.py:156.  Hand-written for
// gfx950: 64-wide wavefronts, 64-bit ballots for the in-wave digit match, LDS digit counters.
//
// Sort pass = 3 launches: per-block digit histogram -> single-block exclusive scan of the
// [digit][block] table -> stable scatter (rank = table prefix + preceding waves + in-wave match rank).
// Stability is what makes equal (tile, depth) keys keep their emission order (ascending Gaussian id).
#include "csplat_common.h"

#include <mutex>
#include <vector>

thread_local char g_csplat_err[512] = {0};
unsigned g_csplat_prof_mask = 0;

namespace {
struct ProfClass {
    std::vector<hipEvent_t> begin, end, pool;
};
ProfClass g_prof[PROF_NCLASSES];
std::mutex g_prof_mu;
hipEvent_t prof_event(ProfClass &c) {
    hipEvent_t e = nullptr;
    if (!c.pool.empty()) { e = c.pool.back(); c.pool.pop_back(); }
    else (void)hipEventCreate(&e);
    return e;
}
}  // namespace

void csplat_prof_mark(int cls, hipStream_t s, bool begin) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfClass &c = g_prof[cls];
    // (never inside a stream capture: an event recorded by a graph node cannot be timed on this ROCm -- external event-record nodes
    //  were tried in round 4: hipEventElapsedTime refuses them with "invalid resource handle")
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusActive) return;
    hipEvent_t e = prof_event(c);
    (void)hipEventRecord(e, s);
    (begin ? c.begin : c.end).push_back(e);
}

extern "C" int csplat_prof_enable(unsigned mask) { g_csplat_prof_mask = mask; return 0; }

extern "C" int csplat_prof_read(int cls, double *ms_total, int64_t *launches) {
    CSPLAT_REQUIRE(cls >= 0 && cls < PROF_NCLASSES, "csplat_prof_read: bad kernel class");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfClass &c = g_prof[cls];
    double tot = 0;
    const size_t n = c.begin.size() < c.end.size() ? c.begin.size() : c.end.size();
    for (size_t i = 0; i < n; i++) {
        float ms = 0;
        HIP_TRY(hipEventSynchronize(c.end[i]));
        HIP_TRY(hipEventElapsedTime(&ms, c.begin[i], c.end[i]));
        tot += ms;
    }
    for (hipEvent_t e : c.begin) c.pool.push_back(e);
    for (hipEvent_t e : c.end) c.pool.push_back(e);
    c.begin.clear(); c.end.clear();
    if (ms_total) *ms_total = tot;
    if (launches) *launches = (int64_t)n;
    return 0;
}

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 2048

constexpr int SORT_THREADS = 256;
constexpr int SORT_ITEMS = 16;
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;  // 4096 keys per workgroup
constexpr int SORT_WAVES = SORT_THREADS / 64;
constexpr int RADIX = 256;

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// block-wide exclusive scan of one value per thread; returns exclusive prefix, *total = block sum
template <int THREADS>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *lds /* THREADS/64 + 1 */, uint32_t *total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int NW = THREADS / 64;
    uint32_t inc = wave_inclusive_scan(v, lane);
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    if (w == 0) {
        uint32_t t = lane < NW ? lds[lane] : 0;
        uint32_t ti = wave_inclusive_scan(t, lane);
        if (lane < NW) lds[lane] = ti - t;  // exclusive wave offsets
        if (lane == NW - 1) lds[NW] = ti;
    }
    __syncthreads();
    uint32_t r = lds[w] + inc - v;
    *total = lds[NW];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_block_sums(const uint32_t *__restrict__ in, uint32_t *__restrict__ sums,
                                                                   int64_t n) {
    __shared__ uint32_t lds[SCAN_THREADS / 64 + 1];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
    if (base + SCAN_ITEMS <= n) {
        const uint4 *p = reinterpret_cast<const uint4 *>(in + base);
        uint4 a = p[0], b = p[1];
        s = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
    } else {
        for (int i = 0; i < SCAN_ITEMS; i++)
            if (base + i < n) s += in[base + i];
    }
    uint32_t tot;
    block_exclusive_scan<SCAN_THREADS>(s, lds, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// single workgroup: in-place exclusive scan of m values; coalesced 16-byte accesses, 4096 values per sweep
__global__ __launch_bounds__(1024) void k_scan_single(uint32_t *__restrict__ data, int64_t m, uint32_t *__restrict__ total_out) {
    __shared__ uint32_t lds[1024 / 64 + 1];
    uint32_t carry = 0;
    for (int64_t base = 0; base < m; base += 4096) {
        const int64_t i0 = base + (int64_t)threadIdx.x * 4;
        uint32_t v[4];
        const bool full = i0 + 4 <= m;
        if (full) {
            const uint4 q = *reinterpret_cast<const uint4 *>(data + i0);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = (i0 + k < m) ? data[i0 + k] : 0u;
        }
        const uint32_t s4 = v[0] + v[1] + v[2] + v[3];
        uint32_t tot;
        uint32_t ex = block_exclusive_scan<1024>(s4, lds, &tot) + carry;
        uint32_t o[4];
        o[0] = ex; o[1] = o[0] + v[0]; o[2] = o[1] + v[1]; o[3] = o[2] + v[2];
        if (full) {
            *reinterpret_cast<uint4 *>(data + i0) = make_uint4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (i0 + k < m) data[i0 + k] = o[k];
        }
        carry += tot;
    }
    if (total_out && threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_final(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                              const uint32_t *__restrict__ block_prefix, int64_t n) {
    __shared__ uint32_t lds[SCAN_THREADS / 64 + 1];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    const bool full = base + SCAN_ITEMS <= n;
    if (full) {
        const uint4 *p = reinterpret_cast<const uint4 *>(in + base);
        uint4 a = p[0], b = p[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) v[i] = (base + i < n) ? in[base + i] : 0;
    }
#pragma unroll
    for (int i = 1; i < SCAN_ITEMS; i++) v[i] += v[i - 1];
    uint32_t tot;
    uint32_t ex = block_exclusive_scan<SCAN_THREADS>(v[SCAN_ITEMS - 1], lds, &tot) + block_prefix[blockIdx.x];
    if (full) {
        uint4 a = {v[0] + ex, v[1] + ex, v[2] + ex, v[3] + ex}, b = {v[4] + ex, v[5] + ex, v[6] + ex, v[7] + ex};
        uint4 *q = reinterpret_cast<uint4 *>(out + base);
        q[0] = a; q[1] = b;
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++)
            if (base + i < n) out[base + i] = v[i] + ex;
    }
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SORT_THREADS) void k_sort_hist(const uint64_t *__restrict__ keys, uint32_t *__restrict__ table,
                                                             int64_t n, int shift, int nb) {
    __shared__ uint32_t h[RADIX];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * SORT_TILE;
#pragma unroll 4
    for (int i = 0; i < SORT_ITEMS; i++) {
        int64_t idx = base + (int64_t)i * SORT_THREADS + threadIdx.x;
        if (idx < n) atomicAdd(&h[(uint32_t)(keys[idx] >> shift) & 0xFF], 1u);
    }
    __syncthreads();
    table[(int64_t)threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(SORT_THREADS) void k_sort_scatter(const uint64_t *__restrict__ keys_in,
                                                                const uint32_t *__restrict__ vals_in,
                                                                uint64_t *__restrict__ keys_out,
                                                                uint32_t *__restrict__ vals_out,
                                                                const uint32_t *__restrict__ table_excl, int64_t n,
                                                                int shift, int nb) {
    __shared__ uint32_t cnt[SORT_WAVES][RADIX];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < SORT_WAVES; i++) cnt[i][threadIdx.x] = 0;
    __syncthreads();

    const int64_t wbase = (int64_t)blockIdx.x * SORT_TILE + (int64_t)w * (SORT_ITEMS * 64);
    uint64_t key[SORT_ITEMS];
    uint32_t val[SORT_ITEMS];
    uint32_t rank[SORT_ITEMS];
    const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; i++) {
        int64_t idx = wbase + (int64_t)i * 64 + lane;
        bool valid = idx < n;
        key[i] = valid ? keys_in[idx] : 0ull;
        val[i] = valid ? vals_in[idx] : 0u;
    }
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; i++) {
        int64_t idx = wbase + (int64_t)i * 64 + lane;
        bool valid = idx < n;
        uint32_t d = (uint32_t)(key[i] >> shift) & 0xFF;
        uint64_t peers = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            uint64_t m = __builtin_amdgcn_ballot_w64(valid && ((d >> b) & 1));
            peers &= ((d >> b) & 1) ? m : ~m;
        }
        uint32_t prev = cnt[w][d];
        rank[i] = prev + (uint32_t)__popcll(peers & lt);
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & lt) == 0) cnt[w][d] = prev + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {   // thread t owns digit t: turn per-wave counts into absolute output offsets
        const int d = threadIdx.x;
        uint32_t run = table_excl[(int64_t)d * nb + blockIdx.x];
#pragma unroll
        for (int i = 0; i < SORT_WAVES; i++) {
            uint32_t c = cnt[i][d];
            cnt[i][d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; i++) {
        int64_t idx = wbase + (int64_t)i * 64 + lane;
        if (idx < n) {
            uint32_t d = (uint32_t)(key[i] >> shift) & 0xFF;
            uint32_t dst = cnt[w][d] + rank[i];
            keys_out[dst] = key[i];
            vals_out[dst] = val[i];
        }
    }
}

}  // namespace

size_t csplat_scan_temp_bytes(int64_t n) {
    int64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    return align256((size_t)(nb + 1) * sizeof(uint32_t)) + 256;
}

int csplat_inclusive_scan_u32(hipStream_t s, const uint32_t *in, uint32_t *out, int64_t n, void *temp) {
    if (n <= 0) return 0;
    int nb = cdiv(n, SCAN_TILE);
    uint32_t *sums = (uint32_t *)temp;
    k_scan_block_sums<<<nb, SCAN_THREADS, 0, s>>>(in, sums, n);
    LAUNCH_CHECK();
    k_scan_single<<<1, 1024, 0, s>>>(sums, nb, nullptr);
    LAUNCH_CHECK();
    k_scan_final<<<nb, SCAN_THREADS, 0, s>>>(in, out, sums, n);
    LAUNCH_CHECK();
    return 0;
}

size_t csplat_sort_temp_bytes(int64_t n) {
    int64_t nb = (n + SORT_TILE - 1) / SORT_TILE;
    if (nb < 1) nb = 1;
    return align256((size_t)RADIX * nb * sizeof(uint32_t));
}

int csplat_sort_pairs(hipStream_t s, const uint64_t *keys_in, const uint32_t *vals_in, uint64_t *keys_out,
                      uint32_t *vals_out, uint64_t *keys_tmp, uint32_t *vals_tmp, int64_t n, int end_bit, void *temp) {
    if (n <= 0) return 0;
    const int passes = (end_bit + 7) / 8;
    const int nb = cdiv(n, SORT_TILE);
    uint32_t *table = (uint32_t *)temp;
    const uint64_t *ksrc = keys_in;
    const uint32_t *vsrc = vals_in;
    for (int p = 0; p < passes; p++) {
        // destination alternates so that the LAST pass lands in keys_out / vals_out
        const bool to_out = ((passes - 1 - p) % 2) == 0;
        uint64_t *kdst = to_out ? keys_out : keys_tmp;
        uint32_t *vdst = to_out ? vals_out : vals_tmp;
        k_sort_hist<<<nb, SORT_THREADS, 0, s>>>(ksrc, table, n, p * 8, nb);
        LAUNCH_CHECK();
        k_scan_single<<<1, 1024, 0, s>>>(table, (int64_t)RADIX * nb, nullptr);
        LAUNCH_CHECK();
        k_sort_scatter<<<nb, SORT_THREADS, 0, s>>>(ksrc, vsrc, kdst, vdst, table, n, p * 8, nb);
        LAUNCH_CHECK();
        ksrc = kdst;
        vsrc = vdst;
    }
    return 0;
}
