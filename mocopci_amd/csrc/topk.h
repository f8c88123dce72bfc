// topk.h -- per-lane top-K selection shared by knn.hip, knn_pruned.hip and knn_cosine.hip.
// Candidates are uint64 keys (ord(distance) << 32 | index): one v_cmp_lt_u64 orders (distance, index)
// lexicographically, so every kernel returns the same, fully defined result.  The K-list lives in VGPRs,
// sorted ascending; candidates that pass the per-lane threshold tau are parked in an LDS queue
// ([slot][lane], conflict-free) and merged in batches with register bitonic networks (static indices only).
#pragma once
#include "common.h"

typedef unsigned long long mcp_u64;
constexpr mcp_u64 MCP_KEY_INF = ~0ull;

__device__ __forceinline__ void mcp_ce_asc(mcp_u64 &a, mcp_u64 &b) {
    const bool sw = b < a;
    const mcp_u64 lo = sw ? b : a, hi = sw ? a : b;
    a = lo;
    b = hi;
}
__device__ __forceinline__ void mcp_ce_dir(mcp_u64 &a, mcp_u64 &b, bool up) {
    const bool sw = up ? (b < a) : (a < b);
    const mcp_u64 x = sw ? b : a, y = sw ? a : b;
    a = x;
    b = y;
}
// full bitonic sort, ascending
template <int N>
__device__ __forceinline__ void mcp_bitonic_sort(mcp_u64 (&v)[N]) {
#pragma unroll
    for (int k = 2; k <= N; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int l = i ^ j;
                if (l > i) mcp_ce_dir(v[i], v[l], (i & k) == 0);
            }
}
// v bitonic -> ascending
template <int N>
__device__ __forceinline__ void mcp_bitonic_merge_asc(mcp_u64 (&v)[N]) {
#pragma unroll
    for (int j = N >> 1; j > 0; j >>= 1)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int l = i ^ j;
            if (l > i) mcp_ce_asc(v[i], v[l]);
        }
}
// a (K ascending) <- the K smallest of a U q (QS ascending, QS <= K), ascending: element-wise min against the
// reversed q makes the tail bitonic, one bitonic merge restores the order
template <int K, int QS>
__device__ __forceinline__ void mcp_merge_sorted(mcp_u64 (&a)[K], const mcp_u64 (&q)[QS]) {
#pragma unroll
    for (int i = K - QS; i < K; ++i) {
        const mcp_u64 o = q[K - 1 - i];
        a[i] = o < a[i] ? o : a[i];
    }
    mcp_bitonic_merge_asc<K>(a);
}
__device__ __forceinline__ mcp_u64 mcp_make_key(float d, uint32_t idx) { return ((mcp_u64)mcp_ord(d) << 32) | idx; }
// threshold = distance of the K-th entry (+inf while the list is not full)
__device__ __forceinline__ float mcp_tau_of(mcp_u64 kth) {
    const uint32_t hi = (uint32_t)(kth >> 32);
    return hi == 0xFFFFFFFFu ? INFINITY : mcp_unord(hi);
}
// drain a lane's queue ([slot][lane] of (distance bits, index)) into its K-list
template <int K, int QS>
__device__ __forceinline__ void mcp_flush_queue(mcp_u64 (&a)[K], const uint2 (*queue)[64], int lane, int cnt) {
    mcp_u64 qk[QS];
#pragma unroll
    for (int s = 0; s < QS; ++s) {
        const uint2 e = queue[s][lane];
        qk[s] = s < cnt ? mcp_make_key(__uint_as_float(e.x), e.y) : MCP_KEY_INF;
    }
    mcp_bitonic_sort<QS>(qk);
    mcp_merge_sorted<K, QS>(a, qk);
}
// write the first kout entries of a sorted K-list; missing entries (fewer than kout candidates) repeat the last valid one
template <int K>
__device__ __forceinline__ void mcp_store_list(const mcp_u64 (&a)[K], int kout, int *oi, float *od) {
    mcp_u64 last = a[0];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        if (j < kout) {
            const mcp_u64 kk = a[j] == MCP_KEY_INF ? last : a[j];
            last = kk;
            oi[j] = kk == MCP_KEY_INF ? 0 : (int)(uint32_t)kk;
            if (od) od[j] = kk == MCP_KEY_INF ? 0.f : mcp_unord((uint32_t)(kk >> 32));
        }
    }
}
