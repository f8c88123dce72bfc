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
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int WAVES = 4, KT = 64;  // keys per LDS stage (two 32-key MFMA tiles)

__device__ __forceinline__ int chan_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int HD>
__global__ __launch_bounds__(64 * WAVES) void attention_small_kernel(int nq, int nk, int heads, const float *__restrict__ q, int qs,
                                                                     const float *__restrict__ k, int ks, const float *__restrict__ v,
                                                                     int vs, float scale_log2e, float *__restrict__ out, int os) {
    constexpr int KS = HD + 1;  // padded K row stride (floats): A-operand reads are conflict-free
    __shared__ float kt[2][KT * KS];
    __shared__ __attribute__((aligned(16))) float vt[2][KT * HD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    const int head = blockIdx.y, bf = blockIdx.z;
    const int qi = blockIdx.x * (32 * WAVES) + wave * 32 + col;
    const bool live = qi < nq;
    q += ((size_t)bf * nq + (live ? qi : 0)) * qs + head * HD;
    k += (size_t)bf * nk * ks + head * HD;
    v += (size_t)bf * nk * vs + head * HD;

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
            for (int d = 0; d < HD / 2; ++d) o[d] *= alpha2;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(acc[r] - mn);
                l += p;
                const float *vr = &vt[cur][(sub * 32 + chan_of(r, h)) * HD];
                const f2 p2 = {p, p};
#pragma unroll
                for (int d = 0; d < HD; d += 4) {
                    const float4 vv = *reinterpret_cast<const float4 *>(vr + d);
                    o[d / 2 + 0] = __builtin_elementwise_fma(p2, f2{vv.x, vv.y}, o[d / 2 + 0]);
                    o[d / 2 + 1] = __builtin_elementwise_fma(p2, f2{vv.z, vv.w}, o[d / 2 + 1]);
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

}  // namespace

MCP_EXPORT int mcp_attention_small(int bf, int nq, int nk, int heads, int hd, const float *q, int q_stride, const float *k,
                                   int k_stride, const float *v, int v_stride, float scale, float *out, int out_stride,
                                   mcp_stream_t stream) {
    MCP_CHECK_ARGS(bf > 0 && nq > 0 && nk > 0 && heads > 0 && q && k && v && out);
    if (hd != 8 && hd != 16) return MCP_ERR_UNSUPPORTED;
    // float4 accesses: every row start and head offset must be 16-byte aligned
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) return MCP_ERR_BAD_ARG;
    if ((q_stride | k_stride | v_stride | out_stride) & 3) return MCP_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const float sl2 = scale * 1.44269504088896340736f;
    const dim3 grid(mcp_divup(nq, 32 * WAVES), heads, bf);
    mcp_prof_begin(MCP_KERNEL_ATTENTION, s);
    if (hd == 8)
        hipLaunchKernelGGL(attention_small_kernel<8>, grid, dim3(64 * WAVES), 0, s, nq, nk, heads, q, q_stride, k, k_stride, v, v_stride, sl2,
                           out, out_stride);
    else
        hipLaunchKernelGGL(attention_small_kernel<16>, grid, dim3(64 * WAVES), 0, s, nq, nk, heads, q, q_stride, k, k_stride, v, v_stride,
                           sl2, out, out_stride);
    mcp_prof_end(MCP_KERNEL_ATTENTION, s);
    return mcp_launch_status();
}
