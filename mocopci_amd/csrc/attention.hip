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
