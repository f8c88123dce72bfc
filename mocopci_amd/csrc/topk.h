// topk.h -- per-lane top-K selection shared by knn.hip, knn_pruned.hip and knn_cosine.hip.
// Candidates are 64-bit keys that order (distance, index) lexicographically (see mcp_key below), so every kernel
// returns the same, fully defined result.  The K-list lives in VGPRs,
// sorted ascending; candidates that pass the per-lane threshold tau are parked in an LDS queue
// ([slot][lane], conflict-free) and merged in batches with register bitonic networks (static indices only).
#pragma once
#include "common.h"

// Key representation.  A key orders candidates by (distance, index).  It is held as an IEEE double whose HIGH word
// is the float's own bit pattern and whose LOW word is the index (complemented when the distance is negative, so
// that the sign-magnitude order of doubles is ascending in the index on both sides of zero).  Non-NaN floats give
// finite doubles (a float's exponent never fills the double's 11-bit field), so one v_min_f64 / v_max_f64 pair is
// a full compare-exchange -- against v_cmp_lt_u64 + four v_cndmask for an integer key.  -0.0 sorts before +0.0,
// exactly as in the oracle's ord() order.  The empty slot is +inf (as a double).
typedef double mcp_key;
#define MCP_KEY_INF (__hiloint2double(0x7FF00000, 0))

__device__ __forceinline__ uint32_t mcp_key_hi(mcp_key k) { return (uint32_t)__double2hiint(k); }
__device__ __forceinline__ uint32_t mcp_key_lo(mcp_key k) { return (uint32_t)__double2loint(k); }
__device__ __forceinline__ mcp_key mcp_key_words(uint32_t hi, uint32_t lo) { return __hiloint2double((int)hi, (int)lo); }
__device__ __forceinline__ mcp_key mcp_make_key(float d, uint32_t idx) {
    const uint32_t hi = __float_as_uint(d);
    return mcp_key_words(hi, (int)hi < 0 ? ~idx : idx);
}
__device__ __forceinline__ bool mcp_key_is_inf(mcp_key k) { return mcp_key_hi(k) == 0x7FF00000u; }
__device__ __forceinline__ float mcp_key_dist(mcp_key k) { return __uint_as_float(mcp_key_hi(k)); }
__device__ __forceinline__ uint32_t mcp_key_index(mcp_key k) {
    const uint32_t hi = mcp_key_hi(k), lo = mcp_key_lo(k);
    return (int)hi < 0 ? ~lo : lo;
}
// keys are never NaN and never signalling: the raw instructions, without the canonicalisation fmin()/fmax() add
__device__ __forceinline__ mcp_key mcp_key_min(mcp_key a, mcp_key b) {
    mcp_key r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ mcp_key mcp_key_max(mcp_key a, mcp_key b) {
    mcp_key r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ void mcp_ce_asc(mcp_key &a, mcp_key &b) {
    const mcp_key lo = mcp_key_min(a, b), hi = mcp_key_max(a, b);
    a = lo;
    b = hi;
}
__device__ __forceinline__ void mcp_ce_dir(mcp_key &a, mcp_key &b, bool up) {
    const mcp_key lo = mcp_key_min(a, b), hi = mcp_key_max(a, b);
    a = up ? lo : hi;  // 'up' is a compile-time constant at every call site (fully unrolled networks)
    b = up ? hi : lo;
}
// full bitonic sort, ascending
template <int N>
__device__ __forceinline__ void mcp_bitonic_sort(mcp_key (&v)[N]) {
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
__device__ __forceinline__ void mcp_bitonic_merge_asc(mcp_key (&v)[N]) {
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
__device__ __forceinline__ void mcp_merge_sorted(mcp_key (&a)[K], const mcp_key (&q)[QS]) {
#pragma unroll
    for (int i = K - QS; i < K; ++i) {
        a[i] = mcp_key_min(a[i], q[K - 1 - i]);
    }
    mcp_bitonic_merge_asc<K>(a);
}
// threshold = distance of the K-th entry (+inf while the list is not full)
__device__ __forceinline__ float mcp_tau_of(mcp_key kth) { return mcp_key_is_inf(kth) ? INFINITY : mcp_key_dist(kth); }
// drain a lane's queue ([slot][lane] of (distance bits, index)) into its K-list
template <int K, int QS>
__device__ __forceinline__ void mcp_flush_queue(mcp_key (&a)[K], const uint2 (*queue)[64], int lane, int cnt) {
    // All QS reads are issued back to back and pinned above the selects by one empty asm statement.  Left alone the compiler
    // sinks every read under its own `s < cnt` branch: QS divergent branches, each with its own round trip to LDS.
    static_assert(QS == 4 || QS == 8 || QS == 16, "the pin below lists its operands");
    unsigned long long e[QS];
#pragma unroll
    for (int s = 0; s < QS; ++s) e[s] = *reinterpret_cast<const unsigned long long *>(&queue[s][lane]);
    if constexpr (QS == 16)
        asm volatile("" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]), "+v"(e[8]),
                     "+v"(e[9]), "+v"(e[10]), "+v"(e[11]), "+v"(e[12]), "+v"(e[13]), "+v"(e[14]), "+v"(e[15]));
    else if constexpr (QS == 8)
        asm volatile("" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]));
    else
        asm volatile("" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]));
    mcp_key qk[QS];
#pragma unroll
    for (int s = 0; s < QS; ++s)
        qk[s] = s < cnt ? mcp_make_key(__uint_as_float((uint32_t)e[s]), (uint32_t)(e[s] >> 32)) : MCP_KEY_INF;
    mcp_bitonic_sort<QS>(qk);
    mcp_merge_sorted<K, QS>(a, qk);
}
// write the first kout entries of a sorted K-list; missing entries (fewer than kout candidates) repeat the last valid one
template <int K>
__device__ __forceinline__ void mcp_store_list(const mcp_key (&a)[K], int kout, int *oi, float *od) {
    mcp_key last = a[0];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        if (j < kout) {
            const mcp_key kk = mcp_key_is_inf(a[j]) ? last : a[j];
            last = kk;
            oi[j] = mcp_key_is_inf(kk) ? 0 : (int)mcp_key_index(kk);
            if (od) od[j] = mcp_key_is_inf(kk) ? 0.f : mcp_key_dist(kk);
        }
    }
}
