// knn_cosine.hip -- feature-space cosine KNN on fp32 MFMA for gfx950.
//
// Replaces cosine_distance + torch.topk (pointconv_util.py:111-153): the reference normalises both
// feature sets, materialises 1 - Q^ . R^T as a (B,Q,N) matrix with bmm and takes the 16 smallest per
// row (39 calls per forward).  Here the correlation tile never leaves registers:
//   * rows are normalised once (x / sqrt(sum x^2 + 1e-8)) into a caller-provided workspace;
//   * a wave owns 32 queries: they sit on the MFMA column (lane & 31) and their features are the
//     B operand, resident in VGPRs (C/2 registers per lane);
//   * reference features stream through a double-buffered LDS tile of 32 rows (row stride C+1 floats:
//     the per-k-step A-operand read is bank-conflict-free) shared by the 4 waves of a workgroup;
//   * per tile, C/2 v_mfma_f32_32x32x2_f32 accumulate the 32x32 correlation block as an exact fp32,
//     k-ordered fma chain (the oracle's definition); lane (query j, half h) then holds 16 of the 32
//     reference rows, filters them against its running 16-th best into an LDS queue and merges with
//     the register bitonic network shared with knn.hip; the two lane halves of a query are merged at
//     the end.
// Result: the K smallest of d = 1 - dot under the lexicographic order (d, index), ascending.
#include "common.h"
#include "topk.h"

namespace {

typedef mcp_key u64;  // 64-bit (distance, index) key, see topk.h
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define KEY_INF MCP_KEY_INF
constexpr int K = 16, QS = 16, CHK = 4, RT = 32, WAVES = 4;

__device__ __forceinline__ int chan_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// x / sqrt(sum(x^2) + 1e-8): sequential sum of rounded squares (oracle canon).  A wave stages RW consecutive rows
// through LDS with coalesced float4 traffic (rows are contiguous in memory), lane r < RW sums row r in channel order
// from a padded (conflict-free) tile, and the scaled rows go back out coalesced.
template <int C>
__global__ __launch_bounds__(256) void normalize_rows_kernel(long long rows_a, const float *__restrict__ xa, float *__restrict__ ya,
                                                             long long rows_b, const float *__restrict__ xb, float *__restrict__ yb) {
    constexpr int RW = C <= 64 ? 64 : (C == 128 ? 32 : 16), S = C + 1, V = RW * C / 4;  // float4s per tile
    __shared__ float tile[4][RW * S];
    __shared__ float den[4][RW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *t = tile[wave];
    // both feature sets of a search in ONE launch: tiles [0, ta) belong to (xa, ya), the rest to (xb, yb)
    const long long ta = (rows_a + RW - 1) / RW, tb = (rows_b + RW - 1) / RW;
    for (long long g = (long long)blockIdx.x * 4 + wave; g < ta + tb; g += (long long)gridDim.x * 4) {
        const bool second = g >= ta;
        const float *x = second ? xb : xa;
        float *y = second ? yb : ya;
        const long long rows = second ? rows_b : rows_a, r0 = (second ? g - ta : g) * RW;
        const int live = (int)min((long long)RW, rows - r0);
        const float4 *src = reinterpret_cast<const float4 *>(x + r0 * C);
        __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < V; e += 64) {
            const int row = (e * 4) / C, col = (e * 4) % C;
            if (row < live) {
                const float4 v = src[e];
                float *d = t + row * S + col;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < live) {
            const float *xr = t + lane * S;
            float s = 0.f;
            for (int j = 0; j < C; ++j) s = s + xr[j] * xr[j];
            den[wave][lane] = sqrtf(s + 1e-8f);
        }
        __builtin_amdgcn_wave_barrier();
        float4 *dst = reinterpret_cast<float4 *>(y + r0 * C);
        for (int e = lane; e < V; e += 64) {
            const int row = (e * 4) / C, col = (e * 4) % C;
            if (row < live) {
                const float *d = t + row * S + col;
                const float dn = den[wave][row];
                dst[e] = make_float4(d[0] / dn, d[1] / dn, d[2] / dn, d[3] / dn);
            }
        }
    }
}

template <int C>
void launch_normalize(long long rows_a, const float *xa, float *ya, long long rows_b, const float *xb, float *yb, hipStream_t s) {
    constexpr int RW = C <= 64 ? 64 : (C == 128 ? 32 : 16);
    const long long tiles = (rows_a + RW - 1) / RW + (rows_b + RW - 1) / RW, blocks = (tiles + 3) / 4;
    hipLaunchKernelGGL(normalize_rows_kernel<C>, dim3((unsigned)min(blocks, 4096LL)), dim3(256), 0, s, rows_a, xa, ya, rows_b, xb, yb);
}

template <int C>
__global__ __launch_bounds__(64 * WAVES) void knn_cosine_kernel(int q, int n, int kout, const float *__restrict__ nq,
                                                                const float *__restrict__ nr, int *__restrict__ idx,
                                                                float *__restrict__ dist) {
    constexpr int S = C + 1;                     // padded LDS row stride (floats)
    constexpr int PER_THREAD = RT * C / (64 * WAVES);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *tiles = smem;                         // [2][RT * S]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    uint2(*queue)[64] = reinterpret_cast<uint2(*)[64]>(smem + 2 * RT * S) + (size_t)wave * QS;

    const int b = blockIdx.y;
    const int qi = blockIdx.x * (32 * WAVES) + wave * 32 + col;
    const bool live = qi < q;
    const float *qrow = nq + ((size_t)b * q + (live ? qi : 0)) * C;
    float bq[C / 2];
#pragma unroll
    for (int s = 0; s < C / 2; ++s) bq[s] = qrow[2 * s + h];
    nr += (size_t)b * n * C;

    u64 a[K];
#pragma unroll
    for (int j = 0; j < K; ++j) a[j] = KEY_INF;
    float tau = INFINITY;
    int cnt = 0;
    auto flush = [&]() {
        mcp_flush_queue<K, QS>(a, queue, lane, cnt);
        tau = mcp_tau_of(a[K - 1]);
        cnt = 0;
    };

    const int ntiles = (n + RT - 1) / RT;
    float pre[PER_THREAD];
    auto fetch = [&](int t) {
#pragma unroll
        for (int u = 0; u < PER_THREAD; ++u) {
            const int e = tid + u * 64 * WAVES, row = e / C, cc = e % C;
            const int gr = t * RT + row;
            pre[u] = gr < n ? nr[(size_t)gr * C + cc] : 0.f;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < PER_THREAD; ++u) {
            const int e = tid + u * 64 * WAVES, row = e / C, cc = e % C;
            tiles[buf * RT * S + row * S + cc] = pre[u];
        }
    };
    fetch(0);
    stash(0);
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < ntiles) fetch(t + 1);
        __syncthreads();
        const float *ta = tiles + cur * RT * S + col * S + h;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < C / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[2 * s], bq[s], acc, 0, 0, 0);
        const int base = t * RT;
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += CHK) {
#pragma unroll
            for (int r = r0; r < r0 + CHK; ++r) {
                const int ri = base + chan_of(r, h);
                const float d = 1.0f - acc[r];
                if (ri < n && d < tau) {
                    queue[cnt][lane] = make_uint2(__float_as_uint(d), (uint32_t)ri);
                    ++cnt;
                }
            }
            if (__builtin_amdgcn_ballot_w64(cnt > QS - CHK)) flush();
        }
        if (t + 1 < ntiles) stash(cur ^ 1);
    }
    flush();
    // merge the two lane halves of each query
    u64 o[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const uint32_t lo = __shfl_xor(mcp_key_lo(a[j]), 32), hi = __shfl_xor(mcp_key_hi(a[j]), 32);
        o[j] = mcp_key_words(hi, lo);
    }
    mcp_merge_sorted<K, K>(a, o);
    if (!live || h) return;
    int *oi = idx + ((size_t)b * q + qi) * kout;
    float *od = dist ? dist + ((size_t)b * q + qi) * kout : nullptr;
    mcp_store_list<K>(a, kout, oi, od);
}

// Round 4: the same search with the reference range split over the RS waves of a workgroup and NO shared tile.  Round 3's kernel gave a
// workgroup 128 queries and walked every reference with them: 1024 waves at level 1 (one per SIMD, nothing to overlap the selection
// with), 256 and 128 waves at levels 2 / 3 -- a quarter and an eighth of the chip, 83 / 71 us for 7 / 3 us of matrix work.  Here a
// workgroup owns 32 queries; wave r takes the reference tiles r, r + RS, ... and reads its A operand straight from global memory
// (a lane needs the channels of ITS reference row with its own parity: eight 16-byte loads per 32 channels, both lane halves hit the
// same lines, the next 32 channels are in flight while the 16 MFMAs of these run) -- no LDS tile, no workgroup barrier in the loop;
// the RS partial lists meet in LDS at the end.  The k-ordered fp32 fma chain per (query, reference) and the (distance, index) order
// are unchanged, so the indices are the same bits.
template <int C, int RS>
__global__ __launch_bounds__(64 * RS) void knn_cosine_split_kernel(int q, int n, int kout, const float *__restrict__ nq,
                                                                   const float *__restrict__ nr, int *__restrict__ idx,
                                                                   float *__restrict__ dist) {
    constexpr int CH = C / 32;                   // chunks of 32 channels (16 k-steps) per tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    uint2(*queue)[64] = reinterpret_cast<uint2(*)[64]>(smem) + (size_t)wave * QS;

    const int b = blockIdx.y;
    const int qi = blockIdx.x * 32 + col;
    const bool live = qi < q;
    const float *qrow = nq + ((size_t)b * q + (live ? qi : 0)) * C;
    // channel parity h of the query's row (the B operand of k-step i is channel 2i + h).  C <= 128: in registers (whole 16-byte loads,
    // two of the four floats kept).  C = 256 would be 128 registers per lane -- with the staging sets and the selection lists that
    // spilled 74 of them (round 4) -- so there the workgroup's 32 queries sit in LDS once, [k-step][lane], shared by its RS waves:
    // one conflict-free ds_read_b32 per 64-cycle MFMA.
    constexpr bool QL = C > 128;
    float bq[QL ? 1 : C / 2];
    float *ql = smem + (size_t)RS * QS * 64 * 2;   // behind the queues (uint2 = two floats)
    if (QL) {
        for (int i = wave; i < C / 2; i += RS) ql[i * 64 + lane] = qrow[2 * i + h];
        __syncthreads();
    } else {
#pragma unroll
        for (int j = 0; j < (QL ? 0 : C / 4); ++j) {
            const float4 v = reinterpret_cast<const float4 *>(qrow)[j];
            bq[QL ? 0 : 2 * j] = h ? v.y : v.x;
            bq[QL ? 0 : 2 * j + 1] = h ? v.w : v.z;
        }
    }
    nr += (size_t)b * n * C;

    u64 a[K];
#pragma unroll
    for (int j = 0; j < K; ++j) a[j] = KEY_INF;
    float tau = INFINITY;
    int cnt = 0;
    auto flush = [&]() {
        mcp_flush_queue<K, QS>(a, queue, lane, cnt);
        tau = mcp_tau_of(a[K - 1]);
        cnt = 0;
    };

    const int ntiles = (n + RT - 1) / RT;
    const int mine = wave < ntiles ? (ntiles - wave + RS - 1) / RS : 0;   // tiles wave, wave + RS, ...
    // a lane reads its own reference row (clamped at the end of the cloud: those rows are skipped in the selection)
    auto load = [&](int t, int c, float4 (&v)[8]) {
        const int row = min(t * RT + col, n - 1);
        const float4 *p = reinterpret_cast<const float4 *>(nr + (size_t)row * C + c * 32);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[j];
    };
    f32x16 acc;
    auto mfma16 = [&](int c, const float4 (&v)[8]) {   // c is a compile-time constant at every call (unrolled chunk loop): bq stays in registers
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // channels 4j .. 4j+3 of the chunk = k-steps 2j (channels 4j, 4j+1) and 2j+1 (4j+2, 4j+3); lane half h holds channel parity h
            const float q0 = QL ? ql[(c * 16 + 2 * j) * 64 + lane] : bq[QL ? 0 : c * 16 + 2 * j];
            const float q1 = QL ? ql[(c * 16 + 2 * j + 1) * 64 + lane] : bq[QL ? 0 : c * 16 + 2 * j + 1];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? v[j].y : v[j].x, q0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? v[j].w : v[j].z, q1, acc, 0, 0, 0);
        }
    };
    float4 va[8], vb[8];
    if (mine > 0) load(wave, 0, va);
    for (int ti = 0; ti < mine; ++ti) {
        const int t = wave + ti * RS;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int c = 0; c < CH; c += 2) {   // two register sets take turns: the next 32 channels are in flight under these 16 MFMAs
            load(t, c + 1, vb);
            mfma16(c, va);
            if (c + 2 < CH) load(t, c + 2, va);
            else if (ti + 1 < mine) load(t + RS, 0, va);
            mfma16(c + 1, vb);
        }
        const int base = t * RT;
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += CHK) {
#pragma unroll
            for (int r = r0; r < r0 + CHK; ++r) {
                const int ri = base + chan_of(r, h);
                const float d = 1.0f - acc[r];
                if (ri < n && d < tau) {
                    queue[cnt][lane] = make_uint2(__float_as_uint(d), (uint32_t)ri);
                    ++cnt;
                }
            }
            if (__builtin_amdgcn_ballot_w64(cnt > QS - CHK)) flush();
        }
    }
    flush();
    // merge the two lane halves of each query
    u64 o[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const uint32_t lo = __shfl_xor(mcp_key_lo(a[j]), 32), hi = __shfl_xor(mcp_key_hi(a[j]), 32);
        o[j] = mcp_key_words(hi, lo);
    }
    mcp_merge_sorted<K, K>(a, o);
    // ... and the RS reference parts: waves 1.. park their lists in LDS (their own queue space: K = QS slots of 64 lanes), wave 0 merges
    __syncthreads();
    if (wave > 0 && h == 0) {
#pragma unroll
        for (int j = 0; j < K; ++j) queue[j][col] = make_uint2(mcp_key_lo(a[j]), mcp_key_hi(a[j]));
    }
    __syncthreads();
    if (wave != 0 || h) return;
    for (int w = 1; w < RS; ++w) {
        uint2(*other)[64] = reinterpret_cast<uint2(*)[64]>(smem) + (size_t)w * QS;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const uint2 e = other[j][col];
            o[j] = mcp_key_words(e.y, e.x);
        }
        mcp_merge_sorted<K, K>(a, o);
    }
    if (!live) return;
    int *oi = idx + ((size_t)b * q + qi) * kout;
    float *od = dist ? dist + ((size_t)b * q + qi) * kout : nullptr;
    mcp_store_list<K>(a, kout, oi, od);
}

template <int C, int RS>
int launch_cosine_split(int b, int q, int n, int k, const float *nq, const float *nr, int *idx, float *dist, hipStream_t s) {
    static_assert(K == QS && (C / 32) % 2 == 0, "the partial lists reuse the queues; chunks are consumed in pairs");
    const size_t lds = (size_t)RS * QS * 64 * sizeof(uint2) + (C > 128 ? (size_t)(C / 2) * 64 * sizeof(float) : 0);
    auto kern = knn_cosine_split_kernel<C, RS>;
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        { const hipError_t attr_e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (attr_e_ != hipSuccess) return (int)attr_e_; }
        attr_once.done();
    }
    hipLaunchKernelGGL(kern, dim3(mcp_divup(q, 32), b), dim3(64 * RS), lds, s, q, n, k, nq, nr, idx, dist);
    return mcp_launch_status();
}

// reference parts per workgroup: enough waves to cover the chip's 1024 SIMDs twice where the problem allows it
template <int C>
int launch_cosine_auto(int b, int q, int n, int k, const float *nq, const float *nr, int *idx, float *dist, hipStream_t s) {
    const long long groups = (long long)b * ((q + 31) / 32);
    const int tiles = (n + RT - 1) / RT;
    if (groups * 2 >= 2048 || tiles < 4) return launch_cosine_split<C, 2>(b, q, n, k, nq, nr, idx, dist, s);
    if (groups * 4 >= 2048 || tiles < 8) return launch_cosine_split<C, 4>(b, q, n, k, nq, nr, idx, dist, s);
    return launch_cosine_split<C, 8>(b, q, n, k, nq, nr, idx, dist, s);
}

template <int C>
int launch_cosine(int b, int q, int n, int k, const float *nq, const float *nr, int *idx, float *dist, hipStream_t s) {
    const size_t lds = (size_t)2 * RT * (C + 1) * sizeof(float) + (size_t)WAVES * QS * 64 * sizeof(uint2);
    auto kern = knn_cosine_kernel<C>;
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        { const hipError_t attr_e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (attr_e_ != hipSuccess) return (int)attr_e_; }
        attr_once.done();
    }
    hipLaunchKernelGGL(kern, dim3(mcp_divup(q, 32 * WAVES), b), dim3(64 * WAVES), lds, s, q, n, k, nq, nr, idx, dist);
    return mcp_launch_status();
}

}  // namespace

#ifdef MCP_AB
static int g_cosine_use_old = 0;   // A/B builds only: 1 = round 3's shared-tile kernel
extern "C" __attribute__((visibility("default"))) void mcp_knn_cosine_use_old(int on) { g_cosine_use_old = on; }
#endif

MCP_EXPORT int mcp_knn_cosine(int b, int q, int n, int c, int k, const float *qfeat, const float *rfeat, int *idx, float *dist,
                              float *workspace, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && q > 0 && n > 0 && c > 0 && k > 0 && qfeat && rfeat && idx && workspace);
    if (k > K || (c != 64 && c != 128 && c != 256)) return MCP_ERR_UNSUPPORTED;
    if ((((uintptr_t)qfeat) | ((uintptr_t)rfeat) | ((uintptr_t)workspace)) & 15) return MCP_ERR_BAD_ARG;  // float4 row traffic
    hipStream_t s = (hipStream_t)stream;
    float *nq = workspace, *nr = workspace + (size_t)b * q * c;
    mcp_prof_begin(MCP_KERNEL_KNN_COSINE, s);
    const long long rq = (long long)b * q, rr = (long long)b * n;
    if (c == 64) launch_normalize<64>(rq, qfeat, nq, rr, rfeat, nr, s);
    else if (c == 128) launch_normalize<128>(rq, qfeat, nq, rr, rfeat, nr, s);
    else launch_normalize<256>(rq, qfeat, nq, rr, rfeat, nr, s);
    int rc = mcp_launch_status();
    if (rc == MCP_OK) {
#ifdef MCP_AB
        if (g_cosine_use_old)
            rc = c == 64    ? launch_cosine<64>(b, q, n, k, nq, nr, idx, dist, s)
                 : c == 128 ? launch_cosine<128>(b, q, n, k, nq, nr, idx, dist, s)
                            : launch_cosine<256>(b, q, n, k, nq, nr, idx, dist, s);
        else
#endif
        rc = c == 64    ? launch_cosine_auto<64>(b, q, n, k, nq, nr, idx, dist, s)
             : c == 128 ? launch_cosine_auto<128>(b, q, n, k, nq, nr, idx, dist, s)
                        : launch_cosine_auto<256>(b, q, n, k, nq, nr, idx, dist, s);
    }
    mcp_prof_end(MCP_KERNEL_KNN_COSINE, s);
    return rc;
}
