// attention.hip -- fp32 flash-style attention for tiny head dims (8, 16) on gfx950.
//
// Caller-side block of the hot path: InterFrameAttentionInterpretation
// (mocopci.py:650-667: [5 frames x B, 8 heads, N<=2048 tokens, head_dim 8/16]) and CrossAttention of the
// EI cross-formers (mocopci.py:72-86).  The reference materialises the (heads, N, N) score tensor
// (640 MiB per sample at N=2048).  Library flash kernels pad head_dim 8 to their MFMA K and run far from
// the exp/FMA floor, so:
//   * a wave owns 32 queries (MFMA column = lane & 31); S^T = K . Q^T per 32-key tile is hd/2
//     v_mfma_f32_32x32x2_f32 (exact fp32) with Q (pre-scaled by scale*log2 e) resident in VGPRs and the
//     K tile read from a padded LDS image (bank-conflict-free);
//   * lane-half h ends up with 16 of the tile's 32 keys for its query and runs its OWN online softmax
//     stream (max, sum, O[hd]) over them -- no cross-lane traffic per tile; the two halves are merged
//     once at the end;
//   * P.V (N = hd = 8/16 columns) would waste 3/4 of an MFMA, so it runs on packed fp32 FMAs with V rows
//     read as LDS broadcasts;
//   * q, k, v are read in place from the projection outputs (row strides), out is written token-major,
//     so no permute/contiguous copies surround the call.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef mcp_f2 f2;  // scalar pair: see common.h (no packed-fp32 instructions)
constexpr int WAVES = 4, KT = 64;  // keys per LDS stage (two 32-key MFMA tiles)

__device__ __forceinline__ int chan_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int HD>
__global__ __launch_bounds__(64 * WAVES) void attention_small_kernel(int nq, int nk, int heads, const float *__restrict__ q, int qs,
                                                                     const float *__restrict__ k, int ks, const float *__restrict__ v,
                                                                     int vs, float scale_log2e, float *__restrict__ out, int os, int kv_shift) {
    constexpr int KS = HD + 1;  // padded K row stride (floats): A-operand reads are conflict-free
    __shared__ float kt[2][KT * KS];
    __shared__ __attribute__((aligned(16))) float vt[2][KT * HD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    const int head = blockIdx.y, bf = blockIdx.z;
    const int qi = blockIdx.x * (32 * WAVES) + wave * 32 + col;
    const bool live = qi < nq;
    q += ((size_t)bf * nq + (live ? qi : 0)) * qs + head * HD;
    int bkv = bf + kv_shift;  // keys / values of batch element (bf + kv_shift) mod BF (0 <= kv_shift < BF)
    if (bkv >= (int)gridDim.z) bkv -= (int)gridDim.z;
    k += (size_t)bkv * nk * ks + head * HD;
    v += (size_t)bkv * nk * vs + head * HD;

    // B operand: Q[query][2s + h], pre-scaled so that p = exp2(s - m)
    float qf[HD / 2];
#pragma unroll
    for (int s = 0; s < HD / 2; ++s) qf[s] = q[2 * s + h] * scale_log2e;

    float m = -INFINITY, l = 0.f;
    // output accumulators as float pairs: P.V runs on v_pk_fma_f32 (two fma per lane and instruction)
    f2 o[HD / 2];
#pragma unroll
    for (int d = 0; d < HD / 2; ++d) o[d] = f2{0.f, 0.f};

    // stage loader: thread t loads one float4 of K or V
    constexpr int F4_PER_TILE = KT * HD / 4;                 // float4s per K (or V) stage
    constexpr int LOADS = (2 * F4_PER_TILE + 64 * WAVES - 1) / (64 * WAVES);
    float4 pre[LOADS];
    auto fetch = [&](int t) {
#pragma unroll
        for (int u = 0; u < LOADS; ++u) {
            const int e = tid + u * 64 * WAVES;
            const bool isv = e >= F4_PER_TILE;
            const int f = isv ? e - F4_PER_TILE : e;
            const int row = f / (HD / 4), c4 = f % (HD / 4);
            const int key = t * KT + row;
            pre[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < 2 * F4_PER_TILE && key < nk) {
                const float *src = (isv ? v + (size_t)key * vs : k + (size_t)key * ks) + c4 * 4;
                pre[u] = *reinterpret_cast<const float4 *>(src);
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < LOADS; ++u) {
            const int e = tid + u * 64 * WAVES;
            if (e >= 2 * F4_PER_TILE) continue;
            const bool isv = e >= F4_PER_TILE;
            const int f = isv ? e - F4_PER_TILE : e;
            const int row = f / (HD / 4), c4 = f % (HD / 4);
            if (isv) {
                *reinterpret_cast<float4 *>(&vt[buf][row * HD + c4 * 4]) = pre[u];
            } else {
                float *dst = &kt[buf][row * KS + c4 * 4];
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
            // keys beyond nk must not contribute
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
            if (mn == -INFINITY) continue;  // this half has seen no valid key yet
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            m = mn;
            l *= alpha;
            const f2 alpha2 = {alpha, alpha};
#pragma unroll
            for (int d = 0; d < HD / 2; ++d) o[d] = o[d] * alpha2;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(acc[r] - mn);
                l += p;
                const float *vr = &vt[cur][(sub * 32 + chan_of(r, h)) * HD];
                const f2 p2 = {p, p};
#pragma unroll
                for (int d = 0; d < HD; d += 4) {
                    const float4 vv = *reinterpret_cast<const float4 *>(vr + d);
                    o[d / 2 + 0] = mcp_f2_fma(p2, f2{vv.x, vv.y}, o[d / 2 + 0]);
                    o[d / 2 + 1] = mcp_f2_fma(p2, f2{vv.z, vv.w}, o[d / 2 + 1]);
                }
            }
        }
        if (t + 1 < stages) stash(cur ^ 1);
    }
    // merge the two lane halves of each query (log-sum-exp combine)
    const float mo = __shfl_xor(m, 32), lo = __shfl_xor(l, 32);
    const float mm = fmaxf(m, mo);
    const float a0 = m == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m - mm), a1 = mo == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mo - mm);
    const float lsum = l * a0 + lo * a1;
    const float inv = 1.0f / lsum;
    float res[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        const float od = (d & 1) ? o[d / 2].y : o[d / 2].x;
        res[d] = (od * a0 + __shfl_xor(od, 32) * a1) * inv;
    }
    if (live && h == 0) {
        float *dst = out + ((size_t)bf * nq + qi) * os + head * HD;
#pragma unroll
        for (int d = 0; d < HD; d += 4) *reinterpret_cast<float4 *>(dst + d) = make_float4(res[d], res[d + 1], res[d + 2], res[d + 3]);
    }
}


// ---- wide heads (32: EI cross-former of level 3, mocopci.py:72-86 with dim 256 / 8 heads; 256: Cross_Frame_Att, whose 4 "heads"
// are C = 256 wide, mocopci.py:499-522) --------------------------------------------------------------------------------
// Both products on fp32 MFMA.  A wave owns 32 queries (MFMA column); per 32-key tile
//   S^T = K . Q^T      : HD/2 v_mfma_f32_32x32x2_f32, A = K tile from LDS (padded rows), B = Q resident in VGPRs (pre-scaled);
//   O^T += V^T . P     : per 32-channel tile of the head, 16 MFMAs whose B operand is the P tile exactly as the softmax left
//                        it in the accumulator layout (k-step r <-> keys chan_of(r, half)), A = V rows read from LDS in that
//                        same key order; O^T stays in HD/32 accumulator tiles, every register of a lane belongs to that
//                        lane's query, so the online-softmax rescale is lane-local.
// The running maximum is shared by the two lane halves of a query (one cross-half exchange per tile) because both halves feed
// the same MFMA sum; the row sums stay per half and are added once at the end.
template <int HD>
struct WideCfg {
    static constexpr int KT = 32, KS = HD + 1, TD = HD / 32;
    static constexpr size_t LDS_BYTES = 2 * (size_t)KT * (KS + HD) * sizeof(float);
};

template <int HD>
__global__ __launch_bounds__(64 * WAVES, 1) void attention_wide_kernel(int nq, int nk, const float *__restrict__ q, int qs,
                                                                       const float *__restrict__ k, int ks, const float *__restrict__ v,
                                                                       int vs, float scale_log2e, float *__restrict__ out, int os, int kv_shift) {
    using C = WideCfg<HD>;
    constexpr int KT = C::KT, KS = C::KS, TD = C::TD;
    extern __shared__ __attribute__((aligned(16))) float lds_w[];
    float *kt = lds_w;                   // [2][KT][KS]
    float *vt = lds_w + 2 * KT * KS;     // [2][KT][HD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    const int head = blockIdx.y, bf = blockIdx.z;
    const int qi = blockIdx.x * (32 * WAVES) + wave * 32 + col;
    const bool live = qi < nq;
    q += ((size_t)bf * nq + (live ? qi : 0)) * qs + head * HD;
    int bkv = bf + kv_shift;  // keys / values of batch element (bf + kv_shift) mod BF (0 <= kv_shift < BF)
    if (bkv >= (int)gridDim.z) bkv -= (int)gridDim.z;
    k += (size_t)bkv * nk * ks + head * HD;
    v += (size_t)bkv * nk * vs + head * HD;

    float qf[HD / 2];
#pragma unroll
    for (int s4 = 0; s4 < HD / 4; ++s4) {  // Q[query][2s + h]: one float4 holds the operands of two k-steps for both halves
        const float4 t = *reinterpret_cast<const float4 *>(q + 4 * s4);
        qf[2 * s4 + 0] = (h ? t.y : t.x) * scale_log2e;
        qf[2 * s4 + 1] = (h ? t.w : t.z) * scale_log2e;
    }
    f32x16 o[TD];
#pragma unroll
    for (int d = 0; d < TD; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    // Staging: K and V tiles of the NEXT stage are fetched separately (K under the S MFMAs, V under the P.V MFMAs), so only one
    // tile's worth of registers (HD/8 float4 per thread) is ever in flight -- at HD = 256 both at once would spill.
    constexpr int F4_ROW = HD / 4, F4_TILE = KT * F4_ROW;              // float4s per K (or V) tile
    constexpr int LOADS = (F4_TILE + 64 * WAVES - 1) / (64 * WAVES);
    float4 pre[LOADS];
    auto fetch = [&](int t, const float *src, int stride) {
#pragma unroll
        for (int u = 0; u < LOADS; ++u) {
            const int e = tid + u * 64 * WAVES;
            const int row = e / F4_ROW, c4 = e % F4_ROW, key = t * KT + row;
            pre[u] = make_float4(0.f, 0.f, 0.f, 0.f);  // keys past nk: zero rows (their scores are masked, 0 * 0 stays 0)
            if (e < F4_TILE && key < nk) pre[u] = *reinterpret_cast<const float4 *>(src + (size_t)key * stride + c4 * 4);
        }
    };
    auto stash_k = [&](int buf) {
#pragma unroll
        for (int u = 0; u < LOADS; ++u) {
            const int e = tid + u * 64 * WAVES;
            if (e >= F4_TILE) continue;
            float *dst = &kt[(buf * KT + e / F4_ROW) * KS + (e % F4_ROW) * 4];
            dst[0] = pre[u].x; dst[1] = pre[u].y; dst[2] = pre[u].z; dst[3] = pre[u].w;
        }
    };
    auto stash_v = [&](int buf) {
#pragma unroll
        for (int u = 0; u < LOADS; ++u) {
            const int e = tid + u * 64 * WAVES;
            if (e < F4_TILE) *reinterpret_cast<float4 *>(&vt[(buf * KT + e / F4_ROW) * HD + (e % F4_ROW) * 4]) = pre[u];
        }
    };

    const int stages = (nk + KT - 1) / KT;
    fetch(0, k, ks);
    stash_k(0);
    fetch(0, v, vs);
    stash_v(0);
    for (int t = 0; t < stages; ++t) {
        const int cur = t & 1;
        const bool more = t + 1 < stages;
        __syncthreads();  // stage `cur` is complete; every wave has finished reading stage cur^1 (previous iteration)
        if (more) fetch(t + 1, k, ks);
        const float *ka = &kt[(cur * KT + col) * KS + h];
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < HD / 2; ++s) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[2 * s], qf[s], acc, 0, 0, 0);
            if (HD > 64 && (s & 15) == 15) __builtin_amdgcn_sched_barrier(0);  // keep the LDS operand reads from being hoisted en bloc (registers)
        }
        if (more) {
            stash_k(cur ^ 1);
            fetch(t + 1, v, vs);
        } else {
            // Last stage: nothing sits between the S MFMAs and the first vector read of their result when the tile is also full (the
            // masking below is skipped).  With the accumulators in ordinary VGPRs (-amdgpu-mfma-vgpr-form) this compiler's hazard
            // recogniser left 5 of the 18 wait states a 16-pass MFMA result needs on that path (tools/isa_lint.py found it; every
            // other consumer of an MFMA result in the library has its wait states) -- so they are spelled out here, once per launch
            // and wave.
            asm volatile("s_nop 15\n\ts_nop 1" ::: "memory");
        }
        const int kbase = t * KT;
        if (kbase + KT > nk) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kbase + chan_of(r, h) >= nk) acc[r] = -INFINITY;
        }
        float mt = acc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, acc[r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32));          // both halves of a query agree on the maximum (tile 0 always has key 0)
        const float mn = fmaxf(m, mt);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        l *= alpha;
#pragma unroll
        for (int d = 0; d < TD; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = __builtin_amdgcn_exp2f(acc[r] - mn);
            l += p[r];
        }
#pragma unroll
        for (int d = 0; d < TD; ++d) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float a = vt[(cur * KT + chan_of(r, h)) * HD + 32 * d + col];
                o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, p[r], o[d], 0, 0, 0);
            }
            if (HD > 64) __builtin_amdgcn_sched_barrier(0);
        }
        if (more) stash_v(cur ^ 1);
    }
    const float inv = 1.0f / (l + __shfl_xor(l, 32));
    if (live) {
        float *dst = out + ((size_t)bf * nq + qi) * os + head * HD;
#pragma unroll
        for (int d = 0; d < TD; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g)  // registers 4g..4g+3 = channels 32d + 8g + 4h .. +3
                *reinterpret_cast<float4 *>(dst + 32 * d + 8 * g + 4 * h) =
                    make_float4(o[d][4 * g] * inv, o[d][4 * g + 1] * inv, o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv);
    }
}

// ---- the same attention with the KEYS split over the waves of a workgroup (round 5) --------------------------------------------------
// Cross_Frame_Att (mocopci.py:499-522) is 16 x 3 problems of 256 queries x 256 keys at head width 256: in the kernel above that is 96
// workgroups = 384 waves of 8 key stages each, 55 us of f32-MFMA work per wave on a third of the chip's SIMDs (107 us per launch).  Here a
// workgroup owns 32 queries and its four waves take the 32-key stages w, w + 4, ...: four times the waves, a quarter of the serial MFMA
// chain each.  A wave stages its K tile, then (in the same LDS buffer, once the S MFMAs have read K) its V tile -- the tiles are not
// shared between waves any more, so there is no workgroup barrier in the loop; the four partial results (running maximum, row sums,
// O^T tiles) meet in LDS at the end and wave w finishes the output channels 64 w .. 64 w + 63:
//     M = max_i m_i,  O = sum_i exp2(m_i - M) O_i / sum_i exp2(m_i - M) l_i        (the usual log-sum-exp merge, waves in order).
// Same MFMA sequence per (query, key stage) as attention_wide_kernel; the merge changes the order in which the stages' contributions
// are added, so results agree to rounding (tests: same tolerance against float64), not bit for bit.
template <int HD>
struct WideSplitCfg {
    static constexpr int KT = 32, KS = HD + 1, TD = HD / 32;
    static constexpr int BUF = KT * KS;                                  // floats per wave: K tile (padded rows) or V tile or the wave's O^T
    static constexpr size_t LDS_BYTES = (size_t)WAVES * (BUF + 2 * 64) * sizeof(float);
    static_assert(TD * 16 * 64 <= BUF, "a wave's O^T tiles fit its staging buffer");
    static_assert(TD % WAVES == 0, "output tiles shared out evenly");
};

template <int HD>
__global__ __launch_bounds__(64 * WAVES, 1) void attention_wide_ksplit_kernel(int nq, int nk, const float *__restrict__ q, int qs,
                                                                              const float *__restrict__ k, int ks, const float *__restrict__ v,
                                                                              int vs, float scale_log2e, float *__restrict__ out, int os, int kv_shift) {
    using C = WideSplitCfg<HD>;
    constexpr int KT = C::KT, KS = C::KS, TD = C::TD, BUF = C::BUF;
    extern __shared__ __attribute__((aligned(16))) float lds_ws[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    float *buf = lds_ws + wave * BUF;                       // this wave's staging buffer
    float *ml = lds_ws + WAVES * BUF;                       // [WAVES][2][64]: running maximum, row sum
    const int head = blockIdx.y, bf = blockIdx.z;
    const int qi = blockIdx.x * 32 + col;
    const bool live = qi < nq;
    q += ((size_t)bf * nq + (live ? qi : 0)) * qs + head * HD;
    int bkv = bf + kv_shift;
    if (bkv >= (int)gridDim.z) bkv -= (int)gridDim.z;
    k += (size_t)bkv * nk * ks + head * HD;
    v += (size_t)bkv * nk * vs + head * HD;

    float qf[HD / 2];
#pragma unroll
    for (int s4 = 0; s4 < HD / 4; ++s4) {
        const float4 t = *reinterpret_cast<const float4 *>(q + 4 * s4);
        qf[2 * s4 + 0] = (h ? t.y : t.x) * scale_log2e;
        qf[2 * s4 + 1] = (h ? t.w : t.z) * scale_log2e;
    }
    f32x16 o[TD];
#pragma unroll
    for (int d = 0; d < TD; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    // A wave stages a 32-key tile by itself: 32 x HD/4 float4 = 32 per lane, all in flight at once and UNDER the MFMA phase in front of
    // their use: the V tile is requested before the S MFMAs, the next stage's K tile before the P.V MFMAs (one round trip per tile,
    // hidden; four dependent round trips of eight loads each, exposed, made the first version of this kernel 85 us)
    constexpr int F4_ROW = HD / 4, PER_LANE = KT * F4_ROW / 64;
    float4 pre[PER_LANE];
    auto issue = [&](int t, const float *src, int stride) {
#pragma unroll
        for (int u = 0; u < PER_LANE; ++u) {
            const int e = u * 64 + lane;
            const int row = e / F4_ROW, c4 = e % F4_ROW, key = t * KT + row;
            pre[u] = key < nk ? *reinterpret_cast<const float4 *>(src + (size_t)key * stride + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&](bool padded) {
#pragma unroll
        for (int u = 0; u < PER_LANE; ++u) {
            const int e = u * 64 + lane;
            const int row = e / F4_ROW, c4 = e % F4_ROW;
            if (padded) {
                float *dst = &buf[row * KS + c4 * 4];
                dst[0] = pre[u].x; dst[1] = pre[u].y; dst[2] = pre[u].z; dst[3] = pre[u].w;
            } else {
                *reinterpret_cast<float4 *>(&buf[row * HD + c4 * 4]) = pre[u];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    const int stages = (nk + KT - 1) / KT;
    if (wave < stages) issue(wave, k, ks);
    for (int t = wave; t < stages; t += WAVES) {
        commit(true);                      // this stage's K tile
        issue(t, v, vs);                   // its V tile: in flight under the S MFMAs
        const float *ka = &buf[col * KS + h];
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < HD / 2; ++s) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[2 * s], qf[s], acc, 0, 0, 0);
            if ((s & 15) == 15) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_wave_barrier();   // every lane's K reads are issued; LDS serves a wave's accesses in order: V may overwrite the buffer
        commit(false);
        if (t + WAVES < stages) issue(t + WAVES, k, ks);   // the next stage's K tile: in flight under the softmax and the P.V MFMAs
        const int kbase = t * KT;
        if (kbase + KT > nk) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kbase + chan_of(r, h) >= nk) acc[r] = -INFINITY;
        }
        float mt = acc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, acc[r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float mn = fmaxf(m, mt);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        l *= alpha;
#pragma unroll
        for (int d = 0; d < TD; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = __builtin_amdgcn_exp2f(acc[r] - mn);
            l += p[r];
        }
#pragma unroll
        for (int d = 0; d < TD; ++d) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float a = buf[chan_of(r, h) * HD + 32 * d + col];
                o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, p[r], o[d], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_wave_barrier();   // V reads issued before the next stage's K tile is written
    }
    // ---- the four partial results meet in LDS ----
    const float lq = l + __shfl_xor(l, 32);   // the query's row sum over this wave's stages (both lane halves)
#pragma unroll
    for (int d = 0; d < TD; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) buf[(d * 16 + r) * 64 + lane] = o[d][r];
    ml[(wave * 2 + 0) * 64 + lane] = m;
    ml[(wave * 2 + 1) * 64 + lane] = lq;
    __syncthreads();
    float mm = -INFINITY;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) mm = fmaxf(mm, ml[(w * 2 + 0) * 64 + lane]);
    float sc[WAVES], lsum = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const float mw = ml[(w * 2 + 0) * 64 + lane];
        sc[w] = mw == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mw - mm);   // a wave without stages (nk < 32 * WAVES) contributes nothing
        lsum += sc[w] * ml[(w * 2 + 1) * 64 + lane];
    }
    const float inv = 1.0f / lsum;
    if (live) {
        float *dst = out + ((size_t)bf * nq + qi) * os + head * HD;
        constexpr int TPW = TD / WAVES;
#pragma unroll
        for (int dd = 0; dd < TPW; ++dd) {
            const int d = wave * TPW + dd;
            float res[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < WAVES; ++w) a += sc[w] * lds_ws[w * BUF + (d * 16 + r) * 64 + lane];
                res[r] = a * inv;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4 *>(dst + 32 * d + 8 * g + 4 * h) = make_float4(res[4 * g], res[4 * g + 1], res[4 * g + 2], res[4 * g + 3]);
        }
    }
}

template <int HD>
int launch_wide_ksplit(int bf, int nq, int nk, int heads, const float *q, int qs, const float *k, int ks, const float *v, int vs, float sl2,
                       float *out, int os, int kv_shift, hipStream_t s) {
    auto kern = attention_wide_ksplit_kernel<HD>;
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        { const hipError_t attr_e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (attr_e_ != hipSuccess) return (int)attr_e_; }
        attr_once.done();
    }
    hipLaunchKernelGGL(kern, dim3(mcp_divup(nq, 32), heads, bf), dim3(64 * WAVES), WideSplitCfg<HD>::LDS_BYTES, s, nq, nk, q, qs, k, ks, v, vs, sl2,
                       out, os, kv_shift);
    return mcp_launch_status();
}

template <int HD>
int launch_wide(int bf, int nq, int nk, int heads, const float *q, int qs, const float *k, int ks, const float *v, int vs, float sl2,
                float *out, int os, int kv_shift, hipStream_t s) {
    auto kern = attention_wide_kernel<HD>;
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        { const hipError_t attr_e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (attr_e_ != hipSuccess) return (int)attr_e_; }
        attr_once.done();
    }
    hipLaunchKernelGGL(kern, dim3(mcp_divup(nq, 32 * WAVES), heads, bf), dim3(64 * WAVES), WideCfg<HD>::LDS_BYTES, s, nq, nk, q, qs, k, ks, v,
                       vs, sl2, out, os, kv_shift);
    return mcp_launch_status();
}

}  // namespace

namespace {
int attention_any(int bf, int nq, int nk, int heads, int hd, const float *q, int q_stride, const float *k, int k_stride, const float *v,
                  int v_stride, int kv_shift, float scale, float *out, int out_stride, hipStream_t s) {
    // float4 accesses: every row start and head offset must be 16-byte aligned
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) return MCP_ERR_BAD_ARG;
    if ((q_stride | k_stride | v_stride | out_stride) & 3) return MCP_ERR_BAD_ARG;
    if (kv_shift < 0 || kv_shift >= bf) return MCP_ERR_BAD_ARG;
    const float sl2 = scale * 1.44269504088896340736f;
    const dim3 grid(mcp_divup(nq, 32 * WAVES), heads, bf);
    int rc;
    mcp_prof_begin(MCP_KERNEL_ATTENTION, s);
    if (hd == 8 || hd == 16) {
        if (hd == 8)
            hipLaunchKernelGGL(attention_small_kernel<8>, grid, dim3(64 * WAVES), 0, s, nq, nk, heads, q, q_stride, k, k_stride, v, v_stride, sl2,
                               out, out_stride, kv_shift);
        else
            hipLaunchKernelGGL(attention_small_kernel<16>, grid, dim3(64 * WAVES), 0, s, nq, nk, heads, q, q_stride, k, k_stride, v, v_stride,
                               sl2, out, out_stride, kv_shift);
        rc = mcp_launch_status();
    } else {
        rc = hd == 32   ? launch_wide<32>(bf, nq, nk, heads, q, q_stride, k, k_stride, v, v_stride, sl2, out, out_stride, kv_shift, s)
             : hd == 64 ? launch_wide<64>(bf, nq, nk, heads, q, q_stride, k, k_stride, v, v_stride, sl2, out, out_stride, kv_shift, s)
                        // head width 256: few, long problems -- when the query-stationary form would not cover the chip's 1024 SIMDs and there
                        // are key stages to share out, the waves of a workgroup split the keys instead
                        : ((long long)mcp_divup(nq, 32 * WAVES) * heads * bf * WAVES < 1024 && nk >= 32 * WAVES)
                              ? launch_wide_ksplit<256>(bf, nq, nk, heads, q, q_stride, k, k_stride, v, v_stride, sl2, out, out_stride, kv_shift, s)
                              : launch_wide<256>(bf, nq, nk, heads, q, q_stride, k, k_stride, v, v_stride, sl2, out, out_stride, kv_shift, s);
    }
    mcp_prof_end(MCP_KERNEL_ATTENTION, s);
    return rc;
}
}  // namespace

MCP_EXPORT int mcp_attention_small(int bf, int nq, int nk, int heads, int hd, const float *q, int q_stride, const float *k,
                                   int k_stride, const float *v, int v_stride, float scale, float *out, int out_stride,
                                   mcp_stream_t stream) {
    MCP_CHECK_ARGS(bf > 0 && nq > 0 && nk > 0 && heads > 0 && q && k && v && out);
    if (hd != 8 && hd != 16) return MCP_ERR_UNSUPPORTED;
    return attention_any(bf, nq, nk, heads, hd, q, q_stride, k, k_stride, v, v_stride, 0, scale, out, out_stride, (hipStream_t)stream);
}

MCP_EXPORT int mcp_attention_wide(int bf, int nq, int nk, int heads, int hd, const float *q, int q_stride, const float *k,
                                  int k_stride, const float *v, int v_stride, float scale, float *out, int out_stride,
                                  mcp_stream_t stream) {
    MCP_CHECK_ARGS(bf > 0 && nq > 0 && nk > 0 && heads > 0 && q && k && v && out);
    if (hd != 32 && hd != 64 && hd != 256) return MCP_ERR_UNSUPPORTED;
    return attention_any(bf, nq, nk, heads, hd, q, q_stride, k, k_stride, v, v_stride, 0, scale, out, out_stride, (hipStream_t)stream);
}

MCP_EXPORT int mcp_attention(int bf, int nq, int nk, int heads, int hd, const float *q, int q_stride, const float *k, int k_stride,
                             const float *v, int v_stride, int kv_batch_shift, float scale, float *out, int out_stride,
                             mcp_stream_t stream) {
    MCP_CHECK_ARGS(bf > 0 && nq > 0 && nk > 0 && heads > 0 && q && k && v && out);
    if (hd != 8 && hd != 16 && hd != 32 && hd != 64 && hd != 256) return MCP_ERR_UNSUPPORTED;
    return attention_any(bf, nq, nk, heads, hd, q, q_stride, k, k_stride, v, v_stride, kv_batch_shift, scale, out, out_stride,
                         (hipStream_t)stream);
}
