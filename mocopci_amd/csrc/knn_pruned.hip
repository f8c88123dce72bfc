// knn_pruned.hip -- exact KNN with spatial pruning for large reference sets (gfx950).
//
// Same result definition as knn.hip / the oracle (K smallest under (distance, index), ascending,
// distances in the shared fp32 canon), but the reference cloud is visited selectively:
//   * queries and references are Morton-sorted once per cloud (mcp_build_cloud: one launch per cloud up to 16384 points;
//     mcp_morton_codes / mcp_tile_boxes for larger ones, with the sort left to the caller), with one axis-aligned box per
//     tile of PT consecutive sorted references;
//   * a wave owns 16 consecutive SORTED queries (a compact region) x 4 lanes each, computes the lower bound of the
//     squared distance from its query box to every tile box, and visits tiles in ascending bound order; it stops as soon
//     as the smallest unvisited bound exceeds the largest per-lane K-th distance plus a slack that covers the rounding of
//     the distance expression, and skips a tile no query of the wave can use (point-to-box test) -- every skipped
//     reference would have failed the per-lane "d <= tau" test anyway, so the output is bit-identical to the exhaustive
//     scan;
//   * within a visited tile the scan / threshold queue / register bitonic merge are those of knn.hip; candidates carry
//     the ORIGINAL reference index (tie order is defined on it), and "d <= tau" (not "<") is used because tiles are no
//     longer visited in index order;
//   * results are written to the query's original row (qperm).
// At N=8192 a wave scans ~900 of the 8192 references (64-point tiles measured 5-7 % faster than 128/256).
#include <stdlib.h>

#include <rocprim/block/block_radix_sort.hpp>

#include "common.h"
#include "topk.h"

namespace {

typedef mcp_key u64;  // 64-bit (distance, index) key, see topk.h
#define KEY_INF MCP_KEY_INF
#ifndef MCP_PRUNED_PT
#define MCP_PRUNED_PT 64
#endif
constexpr int PT = MCP_PRUNED_PT;  // references per tile
constexpr int MAX_TPL = 16;  // tiles per lane -> up to 1024 tiles (N <= 65536 at 64 references per tile)

__device__ __forceinline__ float wave_minf(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_maxf(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ---- preprocessing kernels ------------------------------------------------------------------
__device__ __forceinline__ uint32_t spread10(uint32_t v) {  // 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
// 30-bit Morton code on the per-batch box [lo, hi] (box (B,6): lo xyz, hi xyz)
__global__ __launch_bounds__(256) void morton_kernel(int n, const float *__restrict__ xyz, const float *__restrict__ box,
                                                     int *__restrict__ codes) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *bx = box + b * 6;
    const float *p = xyz + ((size_t)b * n + i) * 3;
    // isotropic cells (one scale for the three axes): LiDAR clouds are flat, per-axis scaling would slice them
    // into thin slabs with huge x/y extent
    const float ext = fmaxf(fmaxf(bx[3] - bx[0], bx[4] - bx[1]), bx[5] - bx[2]);
    uint32_t c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float t = ext > 0.f ? (p[a] - bx[a]) / ext : 0.f;
        t = fminf(fmaxf(t * 1024.f, 0.f), 1023.f);
        c[a] = (uint32_t)t;
    }
    codes[(size_t)b * n + i] = (int)(spread10(c[0]) | (spread10(c[1]) << 1) | (spread10(c[2]) << 2));
}
// one wave per tile of PT sorted points: (lo xyz, hi xyz)
__global__ __launch_bounds__(64) void tile_box_kernel(int n, int tiles, const float *__restrict__ sorted_xyz, float *__restrict__ boxes) {
    const int b = blockIdx.y, t = blockIdx.x, lane = threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = t * PT + lane; i < min(n, (t + 1) * PT); i += 64) {
        const float *p = sorted_xyz + ((size_t)b * n + i) * 3;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(lo[a], p[a]);
            hi[a] = fmaxf(hi[a], p[a]);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = wave_minf(lo[a]);
        hi[a] = wave_maxf(hi[a]);
    }
    if (lane == 0) {
        float *o = boxes + ((size_t)b * tiles + t) * 6;
        o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2];
    }
}

// ---- fused cloud builder: bbox -> Morton keys -> block radix sort -> gather -> tile boxes -----------------------
// One workgroup per batch element (N <= 16384).  The sort is rocPRIM's block_radix_sort on (30-bit code, index) pairs
// held in registers (IPT per thread): stable, so points of one cell stay in index order -- the same permutation as
// sorting (code << 32 | index).
constexpr int BT = 1024;
#ifndef MCP_CLOUD_CELL_BITS
#define MCP_CLOUD_CELL_BITS 7
#endif
constexpr int CELL_BITS = MCP_CLOUD_CELL_BITS;
template <int IPT>
struct CloudSort {
    using Sort = rocprim::block_radix_sort<uint32_t, BT, IPT, uint32_t>;
    union Lds {
        typename Sort::storage_type sort;
        uint32_t order[BT * IPT];  // sorted original indices, for the tile boxes
    };
};

template <int IPT>
__global__ __launch_bounds__(BT) void build_cloud_kernel(int n, int tiles, const float *__restrict__ xyz,
                                                         float *__restrict__ sorted_xyz, int *__restrict__ perm,
                                                         float *__restrict__ boxes) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long dyn[];
    // dynamic LDS: [0,512) reduction scratch + bbox, then the sort storage / index list
    float(*red)[BT / 64] = reinterpret_cast<float(*)[BT / 64]>(dyn);     // [6][16] floats = 384 B
    float *bbox = reinterpret_cast<float *>(dyn) + 6 * (BT / 64);          // [6]
    typename CloudSort<IPT>::Lds &lds = *reinterpret_cast<typename CloudSort<IPT>::Lds *>(dyn + 64);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    xyz += (size_t)b * n * 3;
    sorted_xyz += (size_t)b * n * 3;
    perm += (size_t)b * n;
    boxes += (size_t)b * tiles * 6;
    // 1. bounding box of the cloud
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < n; i += BT) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = xyz[(size_t)i * 3 + a];
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = wave_minf(lo[a]);
        hi[a] = wave_maxf(hi[a]);
        if (lane == 0) { red[a][wave] = lo[a]; red[3 + a][wave] = hi[a]; }
    }
    __syncthreads();
    if (tid < 6) {
        float v = red[tid][0];
        for (int w = 1; w < BT / 64; ++w) v = tid < 3 ? fminf(v, red[tid][w]) : fmaxf(v, red[tid][w]);
        bbox[tid] = v;
    }
    __syncthreads();
    const float ext = fmaxf(fmaxf(bbox[3] - bbox[0], bbox[4] - bbox[1]), bbox[5] - bbox[2]);
    // 2. keys: isotropic 10-bit cells (LiDAR clouds are flat: per-axis scaling would slice them into thin slabs);
    //    thread t holds points t*IPT .. t*IPT+IPT-1, padding sorts last (bit 30)
    uint32_t keys[IPT], vals[IPT];
#pragma unroll
    for (int u = 0; u < IPT; ++u) {
        const int i = tid * IPT + u;
        // CELL_BITS per axis: the order only steers the pruning (any order gives the same neighbours), and a <= 16384-point cloud
        // has no use for 2^30 cells -- 7 bits per axis (cells of 1/128 of the extent) sort in 6 radix passes instead of 8
        uint32_t k = 1u << (3 * CELL_BITS);
        if (i < n) {
            uint32_t c[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float t = ext > 0.f ? (xyz[(size_t)i * 3 + a] - bbox[a]) / ext : 0.f;
                t = fminf(fmaxf(t * (float)(1 << CELL_BITS), 0.f), (float)((1 << CELL_BITS) - 1));
                c[a] = (uint32_t)t;
            }
            k = spread10(c[0]) | (spread10(c[1]) << 1) | (spread10(c[2]) << 2);
        }
        keys[u] = k;
        vals[u] = (uint32_t)i;
    }
    // 3. stable radix sort of the (code, index) pairs over the code's bits (padding: the bit above them)
    typename CloudSort<IPT>::Sort().sort(keys, vals, lds.sort, 0, 3 * CELL_BITS + 1);
    __syncthreads();
    // 4. permutation + sorted coordinates (sorted position s = tid*IPT + u)
#pragma unroll
    for (int u = 0; u < IPT; ++u) {
        const int s = tid * IPT + u;
        lds.order[s] = vals[u];
        if (s < n) {
            const int src = (int)vals[u];
            perm[s] = src;
            sorted_xyz[(size_t)s * 3 + 0] = xyz[(size_t)src * 3 + 0];
            sorted_xyz[(size_t)s * 3 + 1] = xyz[(size_t)src * 3 + 1];
            sorted_xyz[(size_t)s * 3 + 2] = xyz[(size_t)src * 3 + 2];
        }
    }
    __syncthreads();
    // 5. one box per tile of PT sorted points (a wave per tile)
    for (int t = wave; t < tiles; t += BT / 64) {
        float tl[3] = {INFINITY, INFINITY, INFINITY}, th[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = t * PT + lane; i < min(n, (t + 1) * PT); i += 64) {
            const int src = (int)lds.order[i];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float v = xyz[(size_t)src * 3 + a];
                tl[a] = fminf(tl[a], v);
                th[a] = fmaxf(th[a], v);
            }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            tl[a] = wave_minf(tl[a]);
            th[a] = wave_maxf(th[a]);
        }
        if (lane == 0) {
            float *o = boxes + (size_t)t * 6;
            o[0] = tl[0]; o[1] = tl[1]; o[2] = tl[2]; o[3] = th[0]; o[4] = th[1]; o[5] = th[2];
        }
    }
}

template <int IPT>
int launch_build_cloud(int b, int n, int tiles, const float *xyz, float *sorted_xyz, int *perm, float *boxes, hipStream_t s) {
    auto kern = build_cloud_kernel<IPT>;
    const size_t lds = 512 + sizeof(typename CloudSort<IPT>::Lds);
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_once.done();
    }
    hipLaunchKernelGGL(kern, dim3(b), dim3(BT), lds, s, n, tiles, xyz, sorted_xyz, perm, boxes);
    return mcp_launch_status();
}

// ---- the search ------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ float pair_dist(float qx, float qy, float qz, float qn, const float4 r) {
    if (MODE == MCP_DIST_EXPANSION) return mcp_expdist(qx, qy, qz, qn, r.x, r.y, r.z, r.w);
    return mcp_sqdist3(qx, qy, qz, r.x, r.y, r.z);
}

template <int K>
struct PrunedLds {
#ifndef MCP_PRUNED_QS
#define MCP_PRUNED_QS 16
#endif
    static constexpr int QS = MCP_PRUNED_QS < K ? MCP_PRUNED_QS : K;  // the merge network needs QS <= K
    static constexpr int TILE_BYTES = PT * 16;      // float4 (x,y,z,|r|^2)
    static constexpr int PERM_BYTES = PT * 4;       // original indices, stored [sub][PT/SUB]
    static constexpr int QUEUE_BYTES = QS * 64 * 8;
    static constexpr int WAVE_BYTES = TILE_BYTES + PERM_BYTES + QUEUE_BYTES;
};

// min over the SUB adjacent lanes that share a query
template <int SUB>
__device__ __forceinline__ float sub_min(float v) {
    if (SUB >= 2) v = fminf(v, __uint_as_float(mcp_dpp<0xB1>(__float_as_uint(v))));  // quad_perm [1,0,3,2]
    if (SUB >= 4) v = fminf(v, __uint_as_float(mcp_dpp<0x4E>(__float_as_uint(v))));  // quad_perm [2,3,0,1]
    return v;
}

template <int SUB>
__device__ __forceinline__ float sub_max(float v) {
    if (SUB >= 2) v = fmaxf(v, __uint_as_float(mcp_dpp<0xB1>(__float_as_uint(v))));
    if (SUB >= 4) v = fmaxf(v, __uint_as_float(mcp_dpp<0x4E>(__float_as_uint(v))));
    return v;
}

// SUB lanes cooperate on one query (64/SUB queries per wave): lane sub = lane % SUB scans references
// r = sub (mod SUB) of every visited tile into its own K-list; the push threshold is the minimum of the
// SUB K-th distances (any one list already holds K references below it, so nothing above can reach the
// final K); the SUB lists are merged through DPP exchanges at the end.  More, shorter waves with smaller
// query boxes: better pruning, latency and load balance than one lane per query.
#ifdef MCP_KNN_DIAG
// diagnostic build only (never in the product library): [0] waves, [1] tiles visited, [2] flushes, [3] pushes (all lanes)
__device__ unsigned long long g_knn_diag[4];
#define KNN_COUNT(slot, v) do { if (lane == 0) atomicAdd(&g_knn_diag[slot], (unsigned long long)(v)); } while (0)
#else
#define KNN_COUNT(slot, v)
#endif

template <int K, int MODE, int SUB, int TPL>
__global__ __launch_bounds__(64) void knn_pruned_kernel(int q, int n, int tiles, int kout, const float *__restrict__ query,
                                                        const int *__restrict__ qperm, const float *__restrict__ ref,
                                                        const int *__restrict__ rperm, const float *__restrict__ boxes,
                                                        int *__restrict__ idx, float *__restrict__ dist) {
    using L = PrunedLds<K>;
    constexpr int QS = L::QS, CHK = 4, QPW = 64 / SUB, RPL = PT / SUB;  // queries per wave, references per lane per tile
    extern __shared__ float4 smem_f4[];
    const int lane = threadIdx.x;
    char *wbase = reinterpret_cast<char *>(smem_f4);
    float4 *tile = reinterpret_cast<float4 *>(wbase);
    int *tperm = reinterpret_cast<int *>(wbase + L::TILE_BYTES);
    uint2(*queue)[64] = reinterpret_cast<uint2(*)[64]>(wbase + L::TILE_BYTES + L::PERM_BYTES);

    const int b = blockIdx.y;
    const int sub = lane % SUB;
    const int q0 = blockIdx.x * QPW;            // first query of the wave: always < q (grid is sized from q)
    const int qi = q0 + lane / SUB;
    const bool live = qi < q;
    // dead lanes replicate the wave's first query so they do not inflate the query box
    const float *qp = query + ((size_t)b * q + (live ? qi : q0)) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const float qn = mcp_sqnorm3(qx, qy, qz);
    ref += (size_t)b * n * 3;
    rperm += (size_t)b * n;
    boxes += (size_t)b * tiles * 6;

    const float bl0 = wave_minf(qx), bl1 = wave_minf(qy), bl2 = wave_minf(qz);
    const float bh0 = wave_maxf(qx), bh1 = wave_maxf(qy), bh2 = wave_maxf(qz);

    // lower bound of the squared distance to each tile box; lane l owns tiles l, l+64, ... (TPL per lane)
    float lb[TPL];
    float m2 = fmaxf(fmaxf(fabsf(bl0), fabsf(bh0)), fmaxf(fmaxf(fabsf(bl1), fabsf(bh1)), fmaxf(fabsf(bl2), fabsf(bh2))));
#pragma unroll
    for (int u = 0; u < TPL; ++u) {
        const int t = lane + 64 * u;
        lb[u] = INFINITY;
        if (t < tiles) {
            const float *bx = boxes + t * 6;
            const float g0 = fmaxf(0.f, fmaxf(bx[0] - bh0, bl0 - bx[3]));
            const float g1 = fmaxf(0.f, fmaxf(bx[1] - bh1, bl1 - bx[4]));
            const float g2 = fmaxf(0.f, fmaxf(bx[2] - bh2, bl2 - bx[5]));
            lb[u] = g0 * g0 + g1 * g1 + g2 * g2;
            m2 = fmaxf(m2, fmaxf(fmaxf(fabsf(bx[0]), fabsf(bx[3])), fmaxf(fmaxf(fabsf(bx[1]), fabsf(bx[4])), fmaxf(fabsf(bx[2]), fabsf(bx[5])))));
        }
    }
    m2 = wave_maxf(m2);
    // |computed - exact| of either distance form is below ~40 * 2^-24 * M^2 = 2.4e-6 M^2 (M = largest |coordinate|);
    // the bound arithmetic errs by a few ulp of the bound: 3e-5*M^2 absolute + 1e-6 relative covers both.
    const float slack_abs = 3e-5f * m2 * m2, slack_rel = 1e-6f;

    u64 a[K];
#pragma unroll
    for (int j = 0; j < K; ++j) a[j] = KEY_INF;
    float tau_own = live ? INFINITY : -INFINITY;  // dead lanes never push and never hold the wave back
    float tau = tau_own;                          // shared push threshold of the query's SUB lanes
    float taumax = INFINITY;                      // largest tau in the wave: what the next tile must beat
    int cnt = 0;
    auto flush = [&]() {
        KNN_COUNT(2, 1);
#ifdef MCP_KNN_DIAG
        atomicAdd(&g_knn_diag[3], (unsigned long long)cnt);
#endif
        mcp_flush_queue<K, QS>(a, queue, lane, cnt);
        if (live) tau_own = mcp_tau_of(a[K - 1]);
        // Push threshold shared by the query's SUB lanes.  Two valid upper bounds of the K-th distance of the union:
        // (a) any single lane's K-th entry; (b) the largest of the lanes' (K/SUB)-th entries -- SUB * K/SUB = K
        // candidates lie at or below it.  (b) is the tight one: a lane's own K-th entry is about the SUB*K-th overall.
        tau = sub_min<SUB>(tau_own);
        if (SUB > 1) tau = fminf(tau, sub_max<SUB>(live ? mcp_tau_of(a[K / SUB - 1]) : -INFINITY));
        taumax = mcp_unord(mcp_wave_max_u32(mcp_ord(tau)));
        cnt = 0;
    };

    // Next unvisited tile by ascending bound, one wave reduction: the key packs the bound's ordered bits truncated to 22
    // (rounded DOWN, so the decoded value is still a lower bound) above the inverted 10-bit tile id.  Returns -1 when
    // every tile has been visited; marks the returned tile visited.
    constexpr uint32_t LBQ_INF = 0xFF800000u >> 10;
    auto next_tile = [&](float &bound) -> int {
        float mylb = lb[0];
        int myu = 0;
#pragma unroll
        for (int u = 1; u < TPL; ++u) {
            if (lb[u] < mylb) { mylb = lb[u]; myu = u; }
        }
        const uint32_t myt = (uint32_t)(lane + 64 * myu);
        const uint32_t wkey = mcp_wave_max_u32(((~(mcp_ord(mylb) >> 10)) << 10) | (~myt & 0x3FFu));
        const uint32_t lbq = ~(wkey >> 10) & 0x3FFFFFu;
        if (lbq == LBQ_INF) return -1;
        const int t = (int)(~wkey & 0x3FFu);
        bound = mcp_unord(lbq << 10);
        if ((int)myt == t) {
#pragma unroll
            for (int u = 0; u < TPL; ++u)
                if (u == myu) lb[u] = INFINITY;
        }
        return t;
    };
    // a tile's coordinates and original indices travel through registers, so the NEXT tile's global loads are in
    // flight while the current tile is scanned (PT / 64 references per lane)
    struct TileRegs { float x[PT / 64], y[PT / 64], z[PT / 64]; int pi[PT / 64]; };
    auto fetch = [&](int t, TileRegs &g) {
#pragma unroll
        for (int u = 0; u < PT / 64; ++u) {
            const int gi = t * PT + lane + 64 * u;
            const bool ok = gi < n;
            const int gg = ok ? gi : 0;
            const float x = ref[(size_t)gg * 3 + 0], y = ref[(size_t)gg * 3 + 1], z = ref[(size_t)gg * 3 + 2];
            g.pi[u] = ok ? rperm[gg] : 0;
            // padding: distance evaluates to +inf in both forms
            g.x[u] = ok ? x : (MODE == MCP_DIST_EXPANSION ? 0.f : INFINITY);
            g.y[u] = ok ? y : 0.f;
            g.z[u] = ok ? z : NAN;  // marks padding for the norm below
        }
    };

    // Second, per-query filter.  Tiles are ordered (and the walk is ended) by their distance to the wave's QUERY BOX, which
    // under-estimates badly when the 16 queries of a wave are spread out or far from the references.  A tile that passes
    // the box test is scanned only if, for at least one query, the squared distance from the query POINT to the tile box
    // is within that query's own threshold (same rounding slack as the box test).
    auto wanted = [&](int tt) -> bool {
        const float *bx = boxes + tt * 6;  // wave-uniform address
        const float g0 = fmaxf(0.f, fmaxf(bx[0] - qx, qx - bx[3]));
        const float g1 = fmaxf(0.f, fmaxf(bx[1] - qy, qy - bx[4]));
        const float g2 = fmaxf(0.f, fmaxf(bx[2] - qz, qz - bx[5]));
        const float g = g0 * g0 + g1 * g1 + g2 * g2;
        return __builtin_amdgcn_ballot_w64(g <= tau + (tau * slack_rel + slack_abs)) != 0;  // dead lanes: tau = -inf
    };
    // next tile in bound order that passes both tests; -1 ends the walk (every later tile has a larger box bound)
    auto next_wanted = [&]() -> int {
        for (;;) {
            float bd = 0.f;
            const int tt = next_tile(bd);
            if (tt < 0 || !(bd <= taumax + (taumax * slack_rel + slack_abs))) return -1;
            if (wanted(tt)) return tt;
        }
    };

    float bound = 0.f;
    int t = next_tile(bound);  // the first tile is always visited (tau = +inf)
    TileRegs cur, nxt;
    if (t >= 0) fetch(t, cur);
    KNN_COUNT(0, 1);
    while (t >= 0) {
        KNN_COUNT(1, 1);
        int t2 = next_wanted();  // chosen with the thresholds as they stand BEFORE this tile's scan ...
        if (t2 >= 0) fetch(t2, nxt);
        // stage the tile: coordinates + squared norm in reference order, original indices grouped per sub-lane
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < PT / 64; ++u) {
            const int r = lane + 64 * u;
            const bool pad = cur.z[u] != cur.z[u];
            float4 v = make_float4(cur.x[u], cur.y[u], pad ? 0.f : cur.z[u], 0.f);
            v.w = pad ? (MODE == MCP_DIST_EXPANSION ? INFINITY : 0.f) : mcp_sqnorm3(v.x, v.y, v.z);
            tile[r] = v;
            tperm[(r % SUB) * RPL + r / SUB] = cur.pi[u];
        }
        __builtin_amdgcn_wave_barrier();
        // lane scans references sub, sub+SUB, ... ; j-th reference of the lane is tile[j*SUB + sub]
        const int4 *myperm = reinterpret_cast<const int4 *>(tperm + sub * RPL);
        float4 rc[CHK];
        int4 pc = myperm[0];
#pragma unroll
        for (int u = 0; u < CHK; ++u) rc[u] = tile[u * SUB + sub];
        for (int j0 = 0; j0 < RPL; j0 += CHK) {
            float4 rn[CHK];
            const int jn = j0 + CHK < RPL ? j0 + CHK : j0;
#pragma unroll
            for (int u = 0; u < CHK; ++u) rn[u] = tile[(jn + u) * SUB + sub];
            const int4 pn = myperm[jn >> 2];
            float d[CHK];
#pragma unroll
            for (int u = 0; u < CHK; ++u) d[u] = pair_dist<MODE>(qx, qy, qz, qn, rc[u]);
            const int pidx[CHK] = {pc.x, pc.y, pc.z, pc.w};
#pragma unroll
            for (int u = 0; u < CHK; ++u) {
                if (d[u] <= tau) {
                    queue[cnt][lane] = make_uint2(__float_as_uint(d[u]), (uint32_t)pidx[u]);
                    ++cnt;
                }
            }
            if (__builtin_amdgcn_ballot_w64(cnt > QS - CHK)) flush();
#pragma unroll
            for (int u = 0; u < CHK; ++u) rc[u] = rn[u];
            pc = pn;
        }
        // tighten tau before the next pruning decision, but only when a queue is at least half full: a stale
        // (larger) tau is still a valid bound, it just prunes a little less
        if (__builtin_amdgcn_ballot_w64(cnt >= QS / 2)) flush();
        // ... and re-examined with the tightened ones: the prefetched tile may have become useless (its loads are then
        // wasted and the next candidate is fetched without overlap)
        if (t2 >= 0 && !wanted(t2)) {
            t2 = next_wanted();
            if (t2 >= 0) fetch(t2, nxt);
        }
        t = t2;
        cur = nxt;
    }
    flush();
    // merge the SUB lists of each query (after each round both partners hold the union's K smallest)
    if (SUB >= 2) {
        u64 o[K];
#pragma unroll
        for (int j = 0; j < K; ++j)
            o[j] = mcp_key_words(mcp_dpp<0xB1>(mcp_key_hi(a[j])), mcp_dpp<0xB1>(mcp_key_lo(a[j])));
        mcp_merge_sorted<K, K>(a, o);
    }
    if (SUB >= 4) {
        u64 o[K];
#pragma unroll
        for (int j = 0; j < K; ++j)
            o[j] = mcp_key_words(mcp_dpp<0x4E>(mcp_key_hi(a[j])), mcp_dpp<0x4E>(mcp_key_lo(a[j])));
        mcp_merge_sorted<K, K>(a, o);
    }
    if (!live || sub != 0) return;
    const int row = qperm ? qperm[(size_t)b * q + qi] : qi;
    int *oi = idx + ((size_t)b * q + row) * kout;
    float *od = dist ? dist + ((size_t)b * q + row) * kout : nullptr;
    mcp_store_list<K>(a, kout, oi, od);
}

template <int K, int MODE, int SUB, int TPL>
int launch_pruned_tpl(int b, int q, int n, int tiles, int k, const float *query, const int *qperm, const float *ref, const int *rperm,
                      const float *boxes, int *idx, float *dist, hipStream_t s) {
    const size_t lds = (size_t)PrunedLds<K>::WAVE_BYTES;
    auto kern = knn_pruned_kernel<K, MODE, SUB, TPL>;
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {  // lets the CU's whole 160 KB LDS count towards residency (default budget: 64 KB)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_once.done();
    }
    hipLaunchKernelGGL(kern, dim3(mcp_divup(q, 64 / SUB), b), dim3(64), lds, s, q, n, tiles, k, query, qperm,
                       ref, rperm, boxes, idx, dist);
    return mcp_launch_status();
}

template <int K, int MODE, int SUB>
int launch_pruned_sub(int b, int q, int n, int tiles, int k, const float *query, const int *qperm, const float *ref, const int *rperm,
                      const float *boxes, int *idx, float *dist, hipStream_t s) {
    // tile bounds held per lane: sized to the cloud so the per-visit argmin stays short
    if (tiles <= 64) return launch_pruned_tpl<K, MODE, SUB, 1>(b, q, n, tiles, k, query, qperm, ref, rperm, boxes, idx, dist, s);
    if (tiles <= 128) return launch_pruned_tpl<K, MODE, SUB, 2>(b, q, n, tiles, k, query, qperm, ref, rperm, boxes, idx, dist, s);
    if (tiles <= 256) return launch_pruned_tpl<K, MODE, SUB, 4>(b, q, n, tiles, k, query, qperm, ref, rperm, boxes, idx, dist, s);
    return launch_pruned_tpl<K, MODE, SUB, MAX_TPL>(b, q, n, tiles, k, query, qperm, ref, rperm, boxes, idx, dist, s);
}

template <int K, int MODE>
int launch_pruned(int b, int q, int n, int tiles, int k, const float *query, const int *qperm, const float *ref, const int *rperm,
                  const float *boxes, int *idx, float *dist, hipStream_t s) {
    // 4 lanes per query (1 and 2 were measured slower at every shape of the pipeline)
    return launch_pruned_sub<K, MODE, 4>(b, q, n, tiles, k, query, qperm, ref, rperm, boxes, idx, dist, s);
}

template <int MODE>
int launch_pruned_k(int b, int q, int n, int tiles, int k, const float *query, const int *qperm, const float *ref, const int *rperm,
                    const float *boxes, int *idx, float *dist, hipStream_t s) {
    if (k <= 4) return launch_pruned<4, MODE>(b, q, n, tiles, k, query, qperm, ref, rperm, boxes, idx, dist, s);
    if (k <= 16) return launch_pruned<16, MODE>(b, q, n, tiles, k, query, qperm, ref, rperm, boxes, idx, dist, s);
    return launch_pruned<32, MODE>(b, q, n, tiles, k, query, qperm, ref, rperm, boxes, idx, dist, s);
}

}  // namespace

#ifdef MCP_KNN_DIAG
extern "C" __attribute__((visibility("default"))) int mcp_knn_diag_read(unsigned long long *out4) {
    hipError_t e = hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_knn_diag), sizeof(unsigned long long) * 4);
    unsigned long long z[4] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_knn_diag), z, sizeof(z));
    return (int)e;
}
#endif

MCP_EXPORT int mcp_knn_tile_size(void) { return PT; }

MCP_EXPORT int mcp_morton_codes(int b, int n, const float *xyz, const float *box, int *codes, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && xyz && box && codes);
    hipLaunchKernelGGL(morton_kernel, dim3(mcp_divup(n, 256), b), dim3(256), 0, (hipStream_t)stream, n, xyz, box, codes);
    return mcp_launch_status();
}

MCP_EXPORT int mcp_tile_boxes(int b, int n, const float *sorted_xyz, float *boxes, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && sorted_xyz && boxes);
    const int tiles = (n + PT - 1) / PT;
    hipLaunchKernelGGL(tile_box_kernel, dim3(tiles, b), dim3(64), 0, (hipStream_t)stream, n, tiles, sorted_xyz, boxes);
    return mcp_launch_status();
}

MCP_EXPORT int mcp_build_cloud(int b, int n, const float *xyz, float *sorted_xyz, int *perm, float *boxes, mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && n > 0 && xyz && sorted_xyz && perm && boxes);
    if (n > 16384) return MCP_ERR_UNSUPPORTED;  // larger clouds: mcp_morton_codes + an external sort + mcp_tile_boxes
    const int tiles = (n + PT - 1) / PT;
    hipStream_t s = (hipStream_t)stream;
    if (n <= BT) return launch_build_cloud<1>(b, n, tiles, xyz, sorted_xyz, perm, boxes, s);
    if (n <= 2 * BT) return launch_build_cloud<2>(b, n, tiles, xyz, sorted_xyz, perm, boxes, s);
    if (n <= 4 * BT) return launch_build_cloud<4>(b, n, tiles, xyz, sorted_xyz, perm, boxes, s);
    if (n <= 8 * BT) return launch_build_cloud<8>(b, n, tiles, xyz, sorted_xyz, perm, boxes, s);
    return launch_build_cloud<16>(b, n, tiles, xyz, sorted_xyz, perm, boxes, s);
}

MCP_EXPORT int mcp_knn_pruned(int b, int q, int n, int k, int dist_form, const float *query_sorted, const int *qperm,
                              const float *ref_sorted, const int *rperm, const float *boxes, int *idx, float *dist,
                              mcp_stream_t stream) {
    MCP_CHECK_ARGS(b > 0 && q > 0 && n > 0 && k > 0 && query_sorted && ref_sorted && rperm && boxes && idx);
    MCP_CHECK_ARGS(dist_form == MCP_DIST_EXPANSION || dist_form == MCP_DIST_DIRECT);
    const int tiles = (n + PT - 1) / PT;
    if (k > 32 || tiles > 64 * MAX_TPL) return MCP_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    mcp_prof_begin(MCP_KERNEL_KNN, s);
    int rc;
    if (dist_form == MCP_DIST_EXPANSION)
        rc = launch_pruned_k<MCP_DIST_EXPANSION>(b, q, n, tiles, k, query_sorted, qperm, ref_sorted, rperm, boxes, idx, dist, s);
    else
        rc = launch_pruned_k<MCP_DIST_DIRECT>(b, q, n, tiles, k, query_sorted, qperm, ref_sorted, rperm, boxes, idx, dist, s);
    mcp_prof_end(MCP_KERNEL_KNN, s);
    return rc;
}
