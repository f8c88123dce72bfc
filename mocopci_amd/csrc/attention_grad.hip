// attention_grad.hip -- backward of the fp32 flash-style attention for tiny head dims (mcp_attention_small, head_dim 8 / 16) on gfx950.
// Callers in a training graph: InterFrameAttentionInterpretation (mocopci.py:650-667: [frames x B, 8 heads, N <= 2048 tokens]) and the
// CrossAttention of the EI cross-formers (mocopci.py:72-86).  The reference differentiates the materialised (heads, N, N) softmax;
// library flash kernels pad head_dim 8 to their MFMA K (the three of them take 11.4 ms per training step at B = 8, N = 8192).
// Three kernels in the forward's arrangement -- a wave owns 32 rows on the MFMA column, the other side streams through LDS tiles,
// products over the head dim run as HD/2 v_mfma_f32_32x32x2_f32 (exact fp32), products with HD output columns as fma chains with
// LDS broadcast reads; nothing of size Nq x Nk is written:
//   stats  per query: L = log2-sum-exp of its score row (the forward's online softmax without V), D = dO . O;
//   dq     query-stationary: p = exp2(s - L), dp = dO . V_j (second MFMA set), ds = p (dp - D), dQ += ds K_j;
//   dkv    key-stationary:   the same p, dp, ds per (query, key), dV += p dO_i, dK += ds Q_i.
// Every sum has a fixed order (lane-local chains, two lane halves added once): results repeat bit for bit.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int WAVES = 4, KT = 64;  // rows of the streamed side per LDS stage (two 32-row MFMA tiles)

__device__ __forceinline__ int chan_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// An ordering point for the unrolled per-row loops: the accumulators pass through an empty volatile asm, so the row's fma chains
// end before it and (volatile asms keep their order) the next row's LDS reads, whose address passes through one too, start after it.
// Left alone, the scheduler issues the reads of all 32 unrolled rows first: 256-500 registers, hundreds of spills.
template <int N>
__device__ __forceinline__ void pin(float (&a)[N]) {
#pragma unroll
    for (int i = 0; i < N; i += 8)
        asm volatile("" : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]), "+v"(a[i + 3]), "+v"(a[i + 4]), "+v"(a[i + 5]), "+v"(a[i + 6]), "+v"(a[i + 7]));
}

// One LDS stage of the streamed side: `rows` rows of HD floats from up to two row-major sources, each stored twice --
// padded (stride HD + 1: conflict-free MFMA A-operand reads) and plain (16-byte rows for broadcast float4 reads).
template <int HD>
struct Stage {
    static constexpr int KS = HD + 1;
    static constexpr int PAD = KT * KS, PLAIN = KT * HD;
};

// Attention dropout (net.train(): mocopci.py:660-662 drops entries of the softmax matrix at rate 0.05).  The keep / drop decision of
// entry (row, key) is a counter-based hash of (seed, row, key) -- row = the query's index over (batch, head, query) -- so the forward
// and both backward kernels regenerate the same mask without storing it.  m = 1 / (1 - p) for a kept entry, 0 for a dropped one.
__device__ __forceinline__ float drop_scale(uint32_t seed, uint32_t row, uint32_t key, uint32_t threshold, float inv_keep) {
    uint32_t x = seed ^ (row * 0x9E3779B1u) ^ (key * 0x85EBCA77u);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x >= threshold ? inv_keep : 0.f;
}

// ---- the forward with dropout: mcp_attention_small's kernel (attention.hip) with the mask applied to P in P.V only (the row sums
// that normalise the softmax are taken before the mask, as softmax -> dropout -> matmul does) ----
// lse != NULL: the row's log-sum-exp (log2 domain, the statistic attention_stats_kernel recomputes) is written for the backward --
// a training forward keeps it (mcp_attention_small_lse) and the backward skips the statistics pass.  DROP = false: no mask.
template <int HD, bool DROP>
__global__ __launch_bounds__(64 * WAVES) void attention_small_drop_kernel(int nq, int nk, int heads, const float *__restrict__ q, int qs,
                                                                          const float *__restrict__ k, int ks, const float *__restrict__ v, int vs,
                                                                          float scale_log2e, uint32_t seed, uint32_t threshold, float inv_keep,
                                                                          float *__restrict__ out, float *__restrict__ lse) {
    constexpr int KS = HD + 1;
    __shared__ float kt[2][KT * KS];
    __shared__ __attribute__((aligned(16))) float vt[2][KT * HD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    const int head = blockIdx.y, bf = blockIdx.z;
    const int qi = blockIdx.x * (32 * WAVES) + wave * 32 + col;
    const bool live = qi < nq;
    const size_t qrow = (size_t)bf * nq + (live ? qi : 0);
    const uint32_t row = (uint32_t)(((size_t)bf * heads + head) * nq + qi);
    q += qrow * qs + head * HD;
    k += (size_t)bf * nk * ks + head * HD;
    v += (size_t)bf * nk * vs + head * HD;
    float qf[HD / 2];
#pragma unroll
    for (int s = 0; s < HD / 2; ++s) qf[s] = q[2 * s + h] * scale_log2e;
    float m = -INFINITY, l = 0.f, o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
    constexpr int F4 = KT * HD / 4, LOADS = (2 * F4 + 64 * WAVES - 1) / (64 * WAVES);
    float4 pre[LOADS];
    auto fetch = [&](int t) {
#pragma unroll
        for (int u = 0; u < LOADS; ++u) {
            const int e = tid + u * 64 * WAVES;
            const bool isv = e >= F4;
            const int f = isv ? e - F4 : e, r_ = f / (HD / 4), c4 = f % (HD / 4), key = t * KT + r_;
            pre[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < 2 * F4 && key < nk) pre[u] = *reinterpret_cast<const float4 *>((isv ? v + (size_t)key * vs : k + (size_t)key * ks) + c4 * 4);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < LOADS; ++u) {
            const int e = tid + u * 64 * WAVES;
            if (e >= 2 * F4) continue;
            const bool isv = e >= F4;
            const int f = isv ? e - F4 : e, r_ = f / (HD / 4), c4 = f % (HD / 4);
            if (isv) {
                *reinterpret_cast<float4 *>(&vt[buf][r_ * HD + c4 * 4]) = pre[u];
            } else {
                float *dst = &kt[buf][r_ * KS + c4 * 4];
                dst[0] = pre[u].x; dst[1] = pre[u].y; dst[2] = pre[u].z; dst[3] = pre[u].w;
            }
        }
    };
    const int stages = (nk + KT - 1) / KT;
    fetch(0);
    stash(0);
    for (int t = 0; t < stages; ++t) {
        const int cur = t & 1;
        if (t + 1 < stages) fetch(t + 1);
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < KT / 32; ++sub) {
            const float *ka = &kt[cur][(sub * 32 + col) * KS + h];
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < HD / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[2 * s], qf[s], acc, 0, 0, 0);
            const int kbase = t * KT + sub * 32;
            if (kbase + 32 > nk) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kbase + chan_of(r, h) >= nk) acc[r] = -INFINITY;
            }
            float mt = acc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mt = fmaxf(mt, acc[r]);
            const float mn = fmaxf(m, mt);
            if (mn == -INFINITY) continue;
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            m = mn;
            l *= alpha;
#pragma unroll
            for (int d = 0; d < HD; ++d) o[d] *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(acc[r] - mn);
                l += p;
                float pm = DROP ? p * drop_scale(seed, row, (uint32_t)(kbase + chan_of(r, h)), threshold, inv_keep) : p;
                int off = (sub * 32 + chan_of(r, h)) * HD;
                asm volatile("" : "+v"(off), "+v"(pm));
                const float *vr = &vt[cur][off];
#pragma unroll
                for (int d = 0; d < HD; d += 4) {
                    const float4 vv = *reinterpret_cast<const float4 *>(vr + d);
                    o[d + 0] = __builtin_fmaf(pm, vv.x, o[d + 0]);
                    o[d + 1] = __builtin_fmaf(pm, vv.y, o[d + 1]);
                    o[d + 2] = __builtin_fmaf(pm, vv.z, o[d + 2]);
                    o[d + 3] = __builtin_fmaf(pm, vv.w, o[d + 3]);
                }
                pin(o);
            }
        }
        if (t + 1 < stages) stash(cur ^ 1);
    }
    const float mo = __shfl_xor(m, 32), lo = __shfl_xor(l, 32);
    const float mm = fmaxf(m, mo);
    const float a0 = m == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m - mm), a1 = mo == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mo - mm);
    const float lsum = l * a0 + lo * a1;
    const float inv = 1.0f / lsum;
    float res[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) res[d] = (o[d] * a0 + __shfl_xor(o[d], 32) * a1) * inv;
    if (live && h == 0) {
        float *dst = out + qrow * (size_t)(heads * HD) + head * HD;
#pragma unroll
        for (int d = 0; d < HD; d += 4) *reinterpret_cast<float4 *>(dst + d) = make_float4(res[d], res[d + 1], res[d + 2], res[d + 3]);
        if (lse) lse[((size_t)bf * heads + head) * nq + qi] = mm + __builtin_amdgcn_logf(lsum);  // v_log_f32 is log2: as attention_stats_kernel
    }
}

// D = dO . O per (batch, head, query): what is left of the statistics pass when the forward kept the log-sum-exp
template <int HD>
__global__ __launch_bounds__(256) void attention_dsum_kernel(long long rows, int heads, const float *__restrict__ out, const float *__restrict__ gout,
                                                            int nq, float *__restrict__ dsum) {
    // thread = (batch, query, head) in the memory order of out / gout; dsum is (batch, head, query)
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= rows) return;
    const float4 *o = reinterpret_cast<const float4 *>(out + e * HD), *g = reinterpret_cast<const float4 *>(gout + e * HD);
    float d = 0.f;
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
        const float4 a = g[c], b = o[c];
        d = __builtin_fmaf(a.x, b.x, d); d = __builtin_fmaf(a.y, b.y, d); d = __builtin_fmaf(a.z, b.z, d); d = __builtin_fmaf(a.w, b.w, d);
    }
    const long long bq = e / heads;
    const int head = (int)(e - bq * heads);
    const long long bfi = bq / nq;
    const int qi = (int)(bq - bfi * nq);
    dsum[(bfi * heads + head) * nq + qi] = d;
}

// ---- stats: L (log2 domain) and D = dO . O per (batch, head, query) ----
template <int HD>
__global__ __launch_bounds__(64 * WAVES) void attention_stats_kernel(int nq, int nk, int heads, const float *__restrict__ q, int qs, const float *__restrict__ k,
                                                                     int ks, float scale_log2e, const float *__restrict__ out, const float *__restrict__ gout,
                                                                     float *__restrict__ lse, float *__restrict__ dsum) {
    constexpr int KS = HD + 1;
    __shared__ float kt[2][KT * KS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    const int head = blockIdx.y, bf = blockIdx.z;
    const int qi = blockIdx.x * (32 * WAVES) + wave * 32 + col;
    const bool live = qi < nq;
    const size_t qrow = (size_t)bf * nq + (live ? qi : 0);
    q += qrow * qs + head * HD;
    k += (size_t)bf * nk * ks + head * HD;
    float qf[HD / 2];
#pragma unroll
    for (int s = 0; s < HD / 2; ++s) qf[s] = q[2 * s + h] * scale_log2e;
    float m = -INFINITY, l = 0.f;
    constexpr int F4 = KT * HD / 4, LOADS = (F4 + 64 * WAVES - 1) / (64 * WAVES);
    float4 pre[LOADS];
    auto fetch = [&](int t) {
#pragma unroll
        for (int u = 0; u < LOADS; ++u) {
            const int e = tid + u * 64 * WAVES, row = e / (HD / 4), c4 = e % (HD / 4), key = t * KT + row;
            pre[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < F4 && key < nk) pre[u] = *reinterpret_cast<const float4 *>(k + (size_t)key * ks + c4 * 4);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < LOADS; ++u) {
            const int e = tid + u * 64 * WAVES, row = e / (HD / 4), c4 = e % (HD / 4);
            if (e >= F4) continue;
            float *dst = &kt[buf][row * KS + c4 * 4];
            dst[0] = pre[u].x; dst[1] = pre[u].y; dst[2] = pre[u].z; dst[3] = pre[u].w;
        }
    };
    const int stages = (nk + KT - 1) / KT;
    fetch(0);
    stash(0);
    for (int t = 0; t < stages; ++t) {
        const int cur = t & 1;
        if (t + 1 < stages) fetch(t + 1);
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < KT / 32; ++sub) {
            const float *ka = &kt[cur][(sub * 32 + col) * KS + h];
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < HD / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[2 * s], qf[s], acc, 0, 0, 0);
            const int kbase = t * KT + sub * 32;
            if (kbase + 32 > nk) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kbase + chan_of(r, h) >= nk) acc[r] = -INFINITY;
            }
            float mt = acc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mt = fmaxf(mt, acc[r]);
            const float mn = fmaxf(m, mt);
            if (mn == -INFINITY) continue;
            l *= __builtin_amdgcn_exp2f(m - mn);
            m = mn;
#pragma unroll
            for (int r = 0; r < 16; ++r) l += __builtin_amdgcn_exp2f(acc[r] - mn);
        }
        if (t + 1 < stages) stash(cur ^ 1);
    }
    const float mo = __shfl_xor(m, 32), lo = __shfl_xor(l, 32);
    const float mm = fmaxf(m, mo);
    const float a0 = m == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m - mm), a1 = mo == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mo - mm);
    const float lsum = l * a0 + lo * a1;
    if (live && h == 0) {
        const float *orow = out + qrow * (size_t)(heads * HD) + head * HD, *grow = gout + qrow * (size_t)(heads * HD) + head * HD;
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < HD; ++c) d = __builtin_fmaf(grow[c], orow[c], d);
        const size_t o = ((size_t)bf * heads + head) * nq + qi;
        lse[o] = mm + __builtin_amdgcn_logf(lsum);  // v_log_f32 is log2
        dsum[o] = d;
    }
}

// ---- dq: query-stationary ----
template <int HD, bool DROP>
__global__ __launch_bounds__(64 * WAVES, 2) void attention_dq_kernel(int nq, int nk, int heads, const float *__restrict__ q, int qs, const float *__restrict__ k,
                                                                  int ks, const float *__restrict__ v, int vs, float scale_log2e, float scale,
                                                                  uint32_t seed, uint32_t threshold, float inv_keep,
                                                                  const float *__restrict__ gout, const float *__restrict__ lse,
                                                                  const float *__restrict__ dsum, float *__restrict__ dq) {
    constexpr int KS = HD + 1;
    __shared__ float kt[2][KT * KS];                                  // K, padded: A operand of S
    __shared__ float vt[2][KT * KS];                                  // V, padded: A operand of dP
    __shared__ __attribute__((aligned(16))) float kp[2][KT * HD];     // K, plain: broadcast rows of dQ += ds K
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    const int head = blockIdx.y, bf = blockIdx.z;
    const int qi = blockIdx.x * (32 * WAVES) + wave * 32 + col;
    const bool live = qi < nq;
    const size_t qrow = (size_t)bf * nq + (live ? qi : 0);
    q += qrow * qs + head * HD;
    k += (size_t)bf * nk * ks + head * HD;
    v += (size_t)bf * nk * vs + head * HD;
    const float *grow = gout + qrow * (size_t)(heads * HD) + head * HD;
    const size_t so = ((size_t)bf * heads + head) * nq + (live ? qi : 0);
    const float L = lse[so], D = dsum[so];
    const uint32_t drow = (uint32_t)(((size_t)bf * heads + head) * nq + qi);
    float qf[HD / 2], gf[HD / 2];
#pragma unroll
    for (int s = 0; s < HD / 2; ++s) {
        qf[s] = q[2 * s + h] * scale_log2e;
        gf[s] = grow[2 * s + h];
    }
    float acc_q[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) acc_q[d] = 0.f;
    constexpr int F4 = KT * HD / 4, LOADS = (2 * F4 + 64 * WAVES - 1) / (64 * WAVES);
    float4 pre[LOADS];
    auto fetch = [&](int t) {
#pragma unroll
        for (int u = 0; u < LOADS; ++u) {
            const int e = tid + u * 64 * WAVES;
            const bool isv = e >= F4;
            const int f = isv ? e - F4 : e, row = f / (HD / 4), c4 = f % (HD / 4), key = t * KT + row;
            pre[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < 2 * F4 && key < nk) pre[u] = *reinterpret_cast<const float4 *>((isv ? v + (size_t)key * vs : k + (size_t)key * ks) + c4 * 4);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < LOADS; ++u) {
            const int e = tid + u * 64 * WAVES;
            if (e >= 2 * F4) continue;
            const bool isv = e >= F4;
            const int f = isv ? e - F4 : e, row = f / (HD / 4), c4 = f % (HD / 4);
            float *dst = isv ? &vt[buf][row * KS + c4 * 4] : &kt[buf][row * KS + c4 * 4];
            dst[0] = pre[u].x; dst[1] = pre[u].y; dst[2] = pre[u].z; dst[3] = pre[u].w;
            if (!isv) *reinterpret_cast<float4 *>(&kp[buf][row * HD + c4 * 4]) = pre[u];
        }
    };
    const int stages = (nk + KT - 1) / KT;
    fetch(0);
    stash(0);
    for (int t = 0; t < stages; ++t) {
        const int cur = t & 1;
        if (t + 1 < stages) fetch(t + 1);
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < KT / 32; ++sub) {
            const float *ka = &kt[cur][(sub * 32 + col) * KS + h], *va = &vt[cur][(sub * 32 + col) * KS + h];
            f32x16 acc, accp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accp[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < HD / 2; ++s) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[2 * s], qf[s], acc, 0, 0, 0);
                accp = __builtin_amdgcn_mfma_f32_32x32x2f32(va[2 * s], gf[s], accp, 0, 0, 0);
            }
            const int kbase = t * KT + sub * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = sub * 32 + chan_of(r, h);
                float p = __builtin_amdgcn_exp2f(acc[r] - L);
                if (kbase + chan_of(r, h) >= nk) p = 0.f;
                const float mk = DROP ? drop_scale(seed, drow, (uint32_t)(kbase + chan_of(r, h)), threshold, inv_keep) : 1.0f;
                float ds = p * (mk * accp[r] - D);
                int off = key * HD;
                asm volatile("" : "+v"(off), "+v"(ds));
                const float *kr = &kp[cur][off];
#pragma unroll
                for (int d = 0; d < HD; d += 4) {
                    const float4 kk = *reinterpret_cast<const float4 *>(kr + d);
                    acc_q[d + 0] = __builtin_fmaf(ds, kk.x, acc_q[d + 0]);
                    acc_q[d + 1] = __builtin_fmaf(ds, kk.y, acc_q[d + 1]);
                    acc_q[d + 2] = __builtin_fmaf(ds, kk.z, acc_q[d + 2]);
                    acc_q[d + 3] = __builtin_fmaf(ds, kk.w, acc_q[d + 3]);
                }
                pin(acc_q);
            }
        }
        if (t + 1 < stages) stash(cur ^ 1);
    }
    float res[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) res[d] = (acc_q[d] + __shfl_xor(acc_q[d], 32)) * scale;
    if (live && h == 0) {
        float *dst = dq + qrow * (size_t)(heads * HD) + head * HD;
#pragma unroll
        for (int d = 0; d < HD; d += 4) *reinterpret_cast<float4 *>(dst + d) = make_float4(res[d], res[d + 1], res[d + 2], res[d + 3]);
    }
}

// ---- dkv: key-stationary; writes dK | dV into a (BF, Nk, 2 heads HD) tensor laid out like the forward's kv ----
template <int HD, bool DROP>
__global__ __launch_bounds__(64 * WAVES, 2) void attention_dkv_kernel(int nq, int nk, int heads, const float *__restrict__ q, int qs, const float *__restrict__ k,
                                                                   int ks, const float *__restrict__ v, int vs, float scale_log2e, float scale,
                                                                   uint32_t seed, uint32_t threshold, float inv_keep,
                                                                   const float *__restrict__ gout, const float *__restrict__ lse,
                                                                   const float *__restrict__ dsum, float *__restrict__ dkv) {
    constexpr int KS = HD + 1;
    __shared__ float qt[2][KT * KS];                                  // Q, padded: A operand of S
    __shared__ float gt[2][KT * KS];                                  // dO, padded: A operand of dP
    __shared__ __attribute__((aligned(16))) float qp[2][KT * HD];     // Q, plain: broadcast rows of dK += ds Q
    __shared__ __attribute__((aligned(16))) float gp[2][KT * HD];     // dO, plain: broadcast rows of dV += p dO
    __shared__ float lt[2][KT], dt[2][KT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    const int head = blockIdx.y, bf = blockIdx.z;
    const int ki = blockIdx.x * (32 * WAVES) + wave * 32 + col;
    const bool live = ki < nk;
    const size_t krow = (size_t)bf * nk + (live ? ki : 0);
    const float *kr_ = k + krow * ks + head * HD, *vr_ = v + krow * vs + head * HD;
    q += (size_t)bf * nq * qs + head * HD;
    gout += (size_t)bf * nq * (size_t)(heads * HD) + head * HD;
    lse += ((size_t)bf * heads + head) * nq;
    dsum += ((size_t)bf * heads + head) * nq;
    const uint32_t rbase = (uint32_t)(((size_t)bf * heads + head) * nq);
    float kf[HD / 2], vf[HD / 2];
#pragma unroll
    for (int s = 0; s < HD / 2; ++s) {
        kf[s] = kr_[2 * s + h] * scale_log2e;
        vf[s] = vr_[2 * s + h];
    }
    float acc_k[HD], acc_v[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { acc_k[d] = 0.f; acc_v[d] = 0.f; }
    constexpr int F4 = KT * HD / 4, LOADS = (2 * F4 + 64 * WAVES - 1) / (64 * WAVES);
    float4 pre[LOADS];
    float pl = 0.f, pd = 0.f;
    auto fetch = [&](int t) {
#pragma unroll
        for (int u = 0; u < LOADS; ++u) {
            const int e = tid + u * 64 * WAVES;
            const bool isg = e >= F4;
            const int f = isg ? e - F4 : e, row = f / (HD / 4), c4 = f % (HD / 4), qq = t * KT + row;
            pre[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < 2 * F4 && qq < nq) pre[u] = *reinterpret_cast<const float4 *>((isg ? gout + (size_t)qq * (heads * HD) : q + (size_t)qq * qs) + c4 * 4);
        }
        if (tid < KT) {
            const int qq = t * KT + tid;
            pl = qq < nq ? lse[qq] : INFINITY;  // a query beyond nq contributes p = exp2(s - inf) = 0
            pd = qq < nq ? dsum[qq] : 0.f;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < LOADS; ++u) {
            const int e = tid + u * 64 * WAVES;
            if (e >= 2 * F4) continue;
            const bool isg = e >= F4;
            const int f = isg ? e - F4 : e, row = f / (HD / 4), c4 = f % (HD / 4);
            float *dst = isg ? &gt[buf][row * KS + c4 * 4] : &qt[buf][row * KS + c4 * 4];
            dst[0] = pre[u].x; dst[1] = pre[u].y; dst[2] = pre[u].z; dst[3] = pre[u].w;
            *reinterpret_cast<float4 *>(isg ? &gp[buf][row * HD + c4 * 4] : &qp[buf][row * HD + c4 * 4]) = pre[u];
        }
        if (tid < KT) {
            lt[buf][tid] = pl;
            dt[buf][tid] = pd;
        }
    };
    const int stages = (nq + KT - 1) / KT;
    fetch(0);
    stash(0);
    for (int t = 0; t < stages; ++t) {
        const int cur = t & 1;
        if (t + 1 < stages) fetch(t + 1);
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < KT / 32; ++sub) {
            const float *qa = &qt[cur][(sub * 32 + col) * KS + h], *ga = &gt[cur][(sub * 32 + col) * KS + h];
            f32x16 acc, accp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accp[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < HD / 2; ++s) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[2 * s], kf[s], acc, 0, 0, 0);    // rows = queries of the tile, column = this lane's key
                accp = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[2 * s], vf[s], accp, 0, 0, 0);  // dP[query][key] = dO_query . V_key
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qq = sub * 32 + chan_of(r, h);
                const float p = __builtin_amdgcn_exp2f(acc[r] - lt[cur][qq]);
                const float mk = DROP ? drop_scale(seed, rbase + (uint32_t)(t * KT + qq), (uint32_t)ki, threshold, inv_keep) : 1.0f;
                const float pm = DROP ? p * mk : p;
                float ds = p * (mk * accp[r] - dt[cur][qq]);
                int off = qq * HD;
                asm volatile("" : "+v"(off), "+v"(ds));
                const float *qr = &qp[cur][off], *gr = &gp[cur][off];
#pragma unroll
                for (int d = 0; d < HD; d += 4) {
                    const float4 qv = *reinterpret_cast<const float4 *>(qr + d), gv = *reinterpret_cast<const float4 *>(gr + d);
                    acc_k[d + 0] = __builtin_fmaf(ds, qv.x, acc_k[d + 0]);
                    acc_k[d + 1] = __builtin_fmaf(ds, qv.y, acc_k[d + 1]);
                    acc_k[d + 2] = __builtin_fmaf(ds, qv.z, acc_k[d + 2]);
                    acc_k[d + 3] = __builtin_fmaf(ds, qv.w, acc_k[d + 3]);
                    acc_v[d + 0] = __builtin_fmaf(pm, gv.x, acc_v[d + 0]);
                    acc_v[d + 1] = __builtin_fmaf(pm, gv.y, acc_v[d + 1]);
                    acc_v[d + 2] = __builtin_fmaf(pm, gv.z, acc_v[d + 2]);
                    acc_v[d + 3] = __builtin_fmaf(pm, gv.w, acc_v[d + 3]);
                }
                pin(acc_k);
                pin(acc_v);
            }
        }
        if (t + 1 < stages) stash(cur ^ 1);
    }
    float rk[HD], rv[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        rk[d] = (acc_k[d] + __shfl_xor(acc_k[d], 32)) * scale;
        rv[d] = acc_v[d] + __shfl_xor(acc_v[d], 32);
    }
    if (live && h == 0) {
        float *dst = dkv + krow * (size_t)(2 * heads * HD) + head * HD;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
            *reinterpret_cast<float4 *>(dst + d) = make_float4(rk[d], rk[d + 1], rk[d + 2], rk[d + 3]);
            *reinterpret_cast<float4 *>(dst + heads * HD + d) = make_float4(rv[d], rv[d + 1], rv[d + 2], rv[d + 3]);
        }
    }
}

// saved_lse != NULL: the forward's log-sum-exp (mcp_attention_small_lse): only D = dO . O is computed here; else the statistics pass
template <int HD, bool DROP>
int launch_all(int bf, int nq, int nk, int heads, const float *q, int qs, const float *k, int ks, const float *v, int vs, float scale, uint32_t seed,
               uint32_t threshold, float inv_keep, const float *out, const float *gout, float *dq, float *dkv, float *lse, float *dsum,
               const float *saved_lse, hipStream_t s) {
    const float sl2 = scale * 1.44269504088896340736f;
    const dim3 gq(mcp_divup(nq, 32 * WAVES), heads, bf), gk(mcp_divup(nk, 32 * WAVES), heads, bf);
    if (saved_lse) {
        const long long rows = (long long)bf * nq * heads;
        hipLaunchKernelGGL(attention_dsum_kernel<HD>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, rows, heads, out, gout, nq, dsum);
        lse = const_cast<float *>(saved_lse);   // read only below
    } else {
        hipLaunchKernelGGL(attention_stats_kernel<HD>, gq, dim3(64 * WAVES), 0, s, nq, nk, heads, q, qs, k, ks, sl2, out, gout, lse, dsum);
    }
    hipLaunchKernelGGL((attention_dq_kernel<HD, DROP>), gq, dim3(64 * WAVES), 0, s, nq, nk, heads, q, qs, k, ks, v, vs, sl2, scale, seed, threshold, inv_keep,
                       gout, lse, dsum, dq);
    hipLaunchKernelGGL((attention_dkv_kernel<HD, DROP>), gk, dim3(64 * WAVES), 0, s, nq, nk, heads, q, qs, k, ks, v, vs, sl2, scale, seed, threshold, inv_keep,
                       gout, lse, dsum, dkv);
    return mcp_launch_status();
}

// drop probability -> (threshold of the 32-bit hash below which an entry is dropped, 1 / (1 - p))
bool drop_params(float drop_p, uint32_t *threshold, float *inv_keep) {
    if (!(drop_p >= 0.f && drop_p < 1.f)) return false;
    *threshold = (uint32_t)((double)drop_p * 4294967296.0);
    *inv_keep = (float)(1.0 / (1.0 - (double)drop_p));
    return true;
}

}  // namespace

MCP_EXPORT size_t mcp_attention_small_grad_workspace_bytes(int bf, int nq, int heads) {
    if (bf <= 0 || nq <= 0 || heads <= 0) return 0;
    return (size_t)2 * bf * heads * nq * sizeof(float);
}

namespace {
int forward_with(int bf, int nq, int nk, int heads, int hd, const float *q, int q_stride, const float *k, int k_stride, const float *v, int v_stride,
                 float scale, float drop_p, unsigned seed, float *out, float *lse, mcp_stream_t stream);
}

MCP_EXPORT int mcp_attention_small_dropout(int bf, int nq, int nk, int heads, int hd, const float *q, int q_stride, const float *k, int k_stride,
                                           const float *v, int v_stride, float scale, float drop_p, unsigned seed, float *out, mcp_stream_t stream) {
    return forward_with(bf, nq, nk, heads, hd, q, q_stride, k, k_stride, v, v_stride, scale, drop_p, seed, out, nullptr, stream);
}

MCP_EXPORT int mcp_attention_small_lse(int bf, int nq, int nk, int heads, int hd, const float *q, int q_stride, const float *k, int k_stride,
                                       const float *v, int v_stride, float scale, float drop_p, unsigned seed, float *out, float *lse,
                                       mcp_stream_t stream) {
    MCP_CHECK_ARGS(lse);
    return forward_with(bf, nq, nk, heads, hd, q, q_stride, k, k_stride, v, v_stride, scale, drop_p, seed, out, lse, stream);
}

namespace {
int forward_with(int bf, int nq, int nk, int heads, int hd, const float *q, int q_stride, const float *k, int k_stride, const float *v, int v_stride,
                 float scale, float drop_p, unsigned seed, float *out, float *lse, mcp_stream_t stream) {
    MCP_CHECK_ARGS(bf > 0 && nq > 0 && nk > 0 && heads > 0 && q && k && v && out);
    if (hd != 8 && hd != 16) return MCP_ERR_UNSUPPORTED;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) return MCP_ERR_BAD_ARG;
    if ((q_stride | k_stride | v_stride) & 3) return MCP_ERR_BAD_ARG;
    uint32_t threshold;
    float inv_keep;
    if (!drop_params(drop_p, &threshold, &inv_keep)) return MCP_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const float sl2 = scale * 1.44269504088896340736f;
    const dim3 grid(mcp_divup(nq, 32 * WAVES), heads, bf);
    mcp_prof_begin(MCP_KERNEL_ATTENTION, s);
#define MCP_ATT_FWD(HD_, DROP_)                                                                                                                     \
    hipLaunchKernelGGL((attention_small_drop_kernel<HD_, DROP_>), grid, dim3(64 * WAVES), 0, s, nq, nk, heads, q, q_stride, k, k_stride, v, v_stride, sl2, \
                       seed, threshold, inv_keep, out, lse)
    if (drop_p > 0.f) {
        if (hd == 8) MCP_ATT_FWD(8, true); else MCP_ATT_FWD(16, true);
    } else {
        if (hd == 8) MCP_ATT_FWD(8, false); else MCP_ATT_FWD(16, false);
    }
#undef MCP_ATT_FWD
    mcp_prof_end(MCP_KERNEL_ATTENTION, s);
    return mcp_launch_status();
}
}  // namespace

namespace {
int grad_with(int bf, int nq, int nk, int heads, int hd, const float *q, int q_stride, const float *k, int k_stride, const float *v, int v_stride, float scale,
              float drop_p, unsigned seed, const float *out, const float *grad_out, const float *saved_lse, float *grad_q, float *grad_kv, void *workspace,
              size_t workspace_bytes, mcp_stream_t stream);
}

MCP_EXPORT int mcp_attention_small_grad(int bf, int nq, int nk, int heads, int hd, const float *q, int q_stride, const float *k, int k_stride, const float *v,
                                        int v_stride, float scale, float drop_p, unsigned seed, const float *out, const float *grad_out, float *grad_q,
                                        float *grad_kv, void *workspace, size_t workspace_bytes, mcp_stream_t stream) {
    return grad_with(bf, nq, nk, heads, hd, q, q_stride, k, k_stride, v, v_stride, scale, drop_p, seed, out, grad_out, nullptr, grad_q, grad_kv, workspace,
                     workspace_bytes, stream);
}

MCP_EXPORT int mcp_attention_small_grad_lse(int bf, int nq, int nk, int heads, int hd, const float *q, int q_stride, const float *k, int k_stride,
                                            const float *v, int v_stride, float scale, float drop_p, unsigned seed, const float *out, const float *grad_out,
                                            const float *lse, float *grad_q, float *grad_kv, void *workspace, size_t workspace_bytes, mcp_stream_t stream) {
    MCP_CHECK_ARGS(lse);
    return grad_with(bf, nq, nk, heads, hd, q, q_stride, k, k_stride, v, v_stride, scale, drop_p, seed, out, grad_out, lse, grad_q, grad_kv, workspace,
                     workspace_bytes, stream);
}

namespace {
int grad_with(int bf, int nq, int nk, int heads, int hd, const float *q, int q_stride, const float *k, int k_stride, const float *v, int v_stride, float scale,
              float drop_p, unsigned seed, const float *out, const float *grad_out, const float *saved_lse, float *grad_q, float *grad_kv, void *workspace,
              size_t workspace_bytes, mcp_stream_t stream) {
    MCP_CHECK_ARGS(bf > 0 && nq > 0 && nk > 0 && heads > 0 && q && k && v && out && grad_out && grad_q && grad_kv && workspace);
    if (hd != 8 && hd != 16) return MCP_ERR_UNSUPPORTED;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out | (uintptr_t)grad_out | (uintptr_t)grad_q | (uintptr_t)grad_kv) & 15) return MCP_ERR_BAD_ARG;
    if ((q_stride | k_stride | v_stride) & 3) return MCP_ERR_BAD_ARG;
    if (workspace_bytes < mcp_attention_small_grad_workspace_bytes(bf, nq, heads)) return MCP_ERR_BAD_ARG;
    uint32_t threshold;
    float inv_keep;
    if (!drop_params(drop_p, &threshold, &inv_keep)) return MCP_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    float *lse = static_cast<float *>(workspace), *dsum = lse + (size_t)bf * heads * nq;
    mcp_prof_begin(MCP_KERNEL_ATTENTION, s);
    int rc;
#define MCP_ATT_ARGS bf, nq, nk, heads, q, q_stride, k, k_stride, v, v_stride, scale, seed, threshold, inv_keep, out, grad_out, grad_q, grad_kv, lse, dsum, saved_lse, s
    if (drop_p > 0.f) rc = hd == 8 ? launch_all<8, true>(MCP_ATT_ARGS) : launch_all<16, true>(MCP_ATT_ARGS);
    else rc = hd == 8 ? launch_all<8, false>(MCP_ATT_ARGS) : launch_all<16, false>(MCP_ATT_ARGS);
#undef MCP_ATT_ARGS
    mcp_prof_end(MCP_KERNEL_ATTENTION, s);
    return rc;
}
}  // namespace
