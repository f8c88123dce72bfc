// mlp.hip -- fused two-layer per-point MLP for gfx950:   out = [res +] W2 . act(W1 . x + b1) + b2,   act(v) = v > 0 ? v : slope * v.
//
// Caller-side blocks of the hot path: Mlp_T of Multi_Frame_Att (mocopci.py:1558-1565 inside :551-575: fc1,
// depthwise k=1 conv, PReLU, fc2, residual) and the flow heads trans_block / trans_block_2 -> mapping_xyz (:566-567, :510-511).
// The reference runs each as Linear, (conv), PReLU, Linear launches over (rows x 4C) activations written to and re-read from
// memory.  Here a wave owns 32 rows (MFMA column); x^T sits in registers as the B operand; the hidden layer is produced 32 units
// at a time -- h = act(W1[chunk] . x + b1[chunk]) in one accumulator tile -- and immediately consumed as the B operand of
// out += W2[:, chunk] . h, so the (rows x hidden) activation never exists.  Both layers run on the bf16 matrix pipe through the
// exact three-way operand split of mfma_split.h.  The weights do not fit LDS (C = 128: 512 KB of fp32), so they are streamed:
// mcp_mlp2_pack lays them out chunk by chunk ([W1 rows of the chunk | W2 columns of the chunk | b1 of the chunk], already split),
// and the workgroup double-buffers one chunk in LDS while the previous one is multiplied.
#include "common.h"
#include "mfma_split.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// floats per chunk image: W1 part (cin/16 k-steps), W2 part (cot out tiles x 2 k-steps), b1 (32, as [half][reg])
__host__ __device__ constexpr int chunk_floats(int cin, int cot) { return ((cin / 16) * 3 * 64 + cot * 2 * 3 * 64) * 4 + 32; }

__global__ __launch_bounds__(256) void mlp2_pack_kernel(int cin, int hidden, int cout, int cot, const float *__restrict__ w1,
                                                        const float *__restrict__ b1, const float *__restrict__ w2,
                                                        const float *__restrict__ b2, float *__restrict__ packed) {
    const int chunks = hidden / 32, cf = chunk_floats(cin, cot), k1 = cin / 16;
    const int first = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    // W1 part of chunk c = out tile c of the (hidden x cin) matrix
    for (int e = first; e < chunks * k1 * 64; e += stride) {
        const int lane = e & 63, s = (e >> 6) % k1, c = (e >> 6) / k1;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = w1[(size_t)(32 * c + (lane & 31)) * cin + 32 * (s >> 1) + mcp_chan_of(8 * (s & 1) + i, lane >> 5)];
        const McpSplit3 sp = mcp_split8(v);
        uint4 *o = reinterpret_cast<uint4 *>(packed + (size_t)c * cf) + (size_t)s * 3 * 64 + lane;
        o[0] = sp.p1; o[64] = sp.p2; o[128] = sp.p3;
    }
    // W2 part of chunk c: for every out tile t, k-steps 2c and 2c+1 of the (cout_padded x hidden) matrix (rows >= cout are zero)
    for (int e = first; e < chunks * cot * 2 * 64; e += stride) {
        const int lane = e & 63, s = (e >> 6) & 1, t = ((e >> 7) % cot), c = (e >> 7) / cot;
        const int row = 32 * t + (lane & 31);
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = row < cout ? w2[(size_t)row * hidden + 32 * c + mcp_chan_of(8 * s + i, lane >> 5)] : 0.f;
        const McpSplit3 sp = mcp_split8(v);
        uint4 *o = reinterpret_cast<uint4 *>(packed + (size_t)c * cf) + (size_t)k1 * 3 * 64 + (size_t)(t * 2 + s) * 3 * 64 + lane;
        o[0] = sp.p1; o[64] = sp.p2; o[128] = sp.p3;
    }
    for (int e = first; e < chunks * 32; e += stride) {  // b1 of chunk c as [half][reg]
        const int r = e & 15, h = (e >> 4) & 1, c = e >> 5;
        packed[(size_t)c * cf + cf - 32 + (h * 16 + r)] = b1[32 * c + mcp_chan_of(r, h)];
    }
    for (int e = first; e < cot * 32; e += stride) {  // b2 image after the chunks: [tile][half][reg]
        const int r = e & 15, h = (e >> 4) & 1, t = e >> 5;
        const int ch = 32 * t + mcp_chan_of(r, h);
        packed[(size_t)chunks * cf + e] = ch < cout ? b2[ch] : 0.f;
    }
}

template <int CIN, int COT, int NW>
__global__ __launch_bounds__(64 * NW, 1) void mlp2_kernel(long long rows, int hidden, int cout, float slope, const float *__restrict__ x,
                                                           int xs_, const float *__restrict__ res, int rs_, const float *__restrict__ packed,
                                                           float *__restrict__ out, int os_) {
    constexpr int K1 = CIN / 16, CF = chunk_floats(CIN, COT), CH4 = CF / 4;
    constexpr int LOADS = (CH4 + 64 * NW - 1) / (64 * NW);
    constexpr int BUF = LOADS * 64 * NW * 4;  // floats per LDS buffer: one chunk image rounded up to whole passes of the workgroup
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [2][BUF]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    const int chunks = hidden / 32;
    const long long row = ((long long)blockIdx.x * NW + wave) * 32 + col;
    const bool live = row < rows;
    const long long rr = live ? row : rows - 1;

    // x^T as the B operand: rows loaded straight into accumulator layout, split once
    McpSplit3 xs[K1];
    {
        const float4 *xr = reinterpret_cast<const float4 *>(x + rr * xs_);
#pragma unroll
        for (int t = 0; t < CIN / 32; ++t) {
            f32x16 a;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 v = xr[(32 * t + 8 * g + 4 * h) >> 2];
                a[4 * g + 0] = v.x; a[4 * g + 1] = v.y; a[4 * g + 2] = v.z; a[4 * g + 3] = v.w;
            }
            xs[2 * t + 0] = mcp_split_kstep(a, 0);
            xs[2 * t + 1] = mcp_split_kstep(a, 1);
        }
    }
    f32x16 acc_o[COT];
    {
        const float *b2i = packed + (size_t)chunks * CF;
#pragma unroll
        for (int t = 0; t < COT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[t][r] = b2i[(t * 2 + h) * 16 + r];
    }

    // Staging registers of the next chunk's image.  Native vector type and no guards: as an array of HIP's uint4 structs written
    // under `if (e < CH4)` the array lived in scratch memory and every load was followed by a full s_waitcnt vmcnt(0) -- the chunk
    // hand-off then cost a memory round trip per chunk.  The LDS buffers are padded to whole passes of the workgroup, the source
    // index is clamped (entries past the chunk re-read its last one and are never used).
    u32x4 pre[LOADS];
#define MLP_FETCH(c_)                                                                                                    \
    {                                                                                                                    \
        const u32x4 *src_ = reinterpret_cast<const u32x4 *>(packed + (size_t)(c_) * CF);                                 \
        _Pragma("unroll") for (int u = 0; u < LOADS; ++u) pre[u] = src_[min(tid + u * 64 * NW, CH4 - 1)];                \
    }
#define MLP_STASH(buf_)                                                                                                  \
    {                                                                                                                    \
        u32x4 *dst_ = reinterpret_cast<u32x4 *>(lds + (size_t)(buf_) * BUF);                                             \
        _Pragma("unroll") for (int u = 0; u < LOADS; ++u) dst_[tid + u * 64 * NW] = pre[u];                              \
    }
    MLP_FETCH(0)
    MLP_STASH(0)
    for (int c = 0; c < chunks; ++c) {
        const int cur = c & 1;
        MLP_FETCH(min(c + 1, chunks - 1))  // the last chunk fetches itself again: no branch in the loop body
        __syncthreads();  // chunk c is complete in buffer cur; nobody reads buffer cur^1 any more
        const float *img = lds + (size_t)cur * BUF;
        const uint4 *w1c = reinterpret_cast<const uint4 *>(img) + lane;
        const uint4 *w2c = w1c + (size_t)K1 * 3 * 64;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = img[CF - 32 + h * 16 + r];
        acc = mcp_tile_split<K1>(w1c, xs, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = acc[r] > 0.f ? acc[r] : acc[r] * slope;
        McpSplit3 hs[2];
        hs[0] = mcp_split_kstep(acc, 0);
        hs[1] = mcp_split_kstep(acc, 1);
#pragma unroll
        for (int t = 0; t < COT; ++t) acc_o[t] = mcp_tile_split<2>(w2c + (size_t)t * 2 * 3 * 64, hs, acc_o[t]);
        MLP_STASH(cur ^ 1)
    }
#undef MLP_FETCH
#undef MLP_STASH
    if (!live) return;
    float *orow = out + row * os_;
    const float *rrow = res ? res + row * rs_ : nullptr;
#pragma unroll
    for (int t = 0; t < COT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = 32 * t + 8 * g + 4 * h;
            if (ch + 3 < cout && !(os_ & 3) && !(rrow && (rs_ & 3))) {
                float4 v = make_float4(acc_o[t][4 * g], acc_o[t][4 * g + 1], acc_o[t][4 * g + 2], acc_o[t][4 * g + 3]);
                if (rrow) {
                    const float4 r4 = *reinterpret_cast<const float4 *>(rrow + ch);
                    v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
                }
                *reinterpret_cast<float4 *>(orow + ch) = v;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ch + u < cout) orow[ch + u] = acc_o[t][4 * g + u] + (rrow ? rrow[ch + u] : 0.f);
            }
        }
}

template <int CIN, int COT, int NW>
int launch_mlp2(long long rows, int hidden, int cout, float slope, const float *x, int xs_, const float *res, int rs_, const float *packed,
                float *out, int os_, hipStream_t s) {
    auto kern = mlp2_kernel<CIN, COT, NW>;
    const size_t lds = 2 * (size_t)((chunk_floats(CIN, COT) / 4 + 64 * NW - 1) / (64 * NW)) * (64 * NW) * 4 * sizeof(float);
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        { const hipError_t attr_e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (attr_e_ != hipSuccess) return (int)attr_e_; }
        attr_once.done();
    }
    const long long groups = (rows + 32LL * NW - 1) / (32LL * NW);
    hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(64 * NW), lds, s, rows, hidden, cout, slope, x, xs_, res, rs_, packed, out, os_);
    return mcp_launch_status();
}

bool supported(int cin, int hidden, int cout) {
    const int cot = (cout + 31) / 32;
    if (hidden <= 0 || hidden % 32) return false;
    // The shapes where the fused kernel beats the library chain (tools/mlp_ab.py, MI355X): C = 64 both heads (52 vs 88 us, 40 vs 74 us
    // at 49152 rows), C = 128 with a narrow output (81 vs 85 us at 24576 rows).  Measured and left to the library: 128 -> 512 -> 128
    // (118 vs 111 us) and 256 -> 1024 -> 3 (228 vs 116 us: 12288 rows are 48 workgroups' worth of work for this tiling).
    return (cin == 64 && (cot == 1 || cot == 2)) || (cin == 128 && (cot == 1 || cot == 4));
}

}  // namespace

MCP_EXPORT int mcp_mlp2_packed_floats(int cin, int hidden, int cout) {
    if (!supported(cin, hidden, cout)) return 0;
    const int cot = (cout + 31) / 32;
    return (hidden / 32) * chunk_floats(cin, cot) + cot * 32;
}

MCP_EXPORT int mcp_mlp2_pack(int cin, int hidden, int cout, const float *w1, const float *b1, const float *w2, const float *b2, float *packed,
                             mcp_stream_t stream) {
    MCP_CHECK_ARGS(w1 && b1 && w2 && b2 && packed);
    if (!supported(cin, hidden, cout)) return MCP_ERR_UNSUPPORTED;
    if (((uintptr_t)packed) & 15) return MCP_ERR_BAD_ARG;
    hipLaunchKernelGGL(mlp2_pack_kernel, dim3(128), dim3(256), 0, (hipStream_t)stream, cin, hidden, cout, (cout + 31) / 32, w1, b1, w2, b2, packed);
    return mcp_launch_status();
}

MCP_EXPORT int mcp_mlp2(long long rows, int cin, int hidden, int cout, float slope, const float *x, int x_stride, const float *res,
                        int res_stride, const float *packed, float *out, int out_stride, mcp_stream_t stream) {
    MCP_CHECK_ARGS(rows > 0 && x && packed && out);
    if (!supported(cin, hidden, cout)) return MCP_ERR_UNSUPPORTED;
    if ((((uintptr_t)x) | ((uintptr_t)packed)) & 15 || (x_stride & 3) || x_stride < cin || out_stride < cout || (res && res_stride < cout))
        return MCP_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int cot = (cout + 31) / 32;
    int rc;
    mcp_prof_begin(MCP_KERNEL_MLP, s);
    // 8 waves per workgroup share one streamed weight image; below 32768 rows 4-wave workgroups (128 rows each) cover the chip
    // better (24576 rows: 82 -> 65 us at 128-512-128, 53 -> 42 us at 128-512-3; equal at 49152 rows)
#define MLP2_GO(CIN_, COT_)                                                                                                              \
    (rows < 32768 ? launch_mlp2<CIN_, COT_, 4>(rows, hidden, cout, slope, x, x_stride, res, res_stride, packed, out, out_stride, s)       \
                  : launch_mlp2<CIN_, COT_, 8>(rows, hidden, cout, slope, x, x_stride, res, res_stride, packed, out, out_stride, s))
    if (cin == 64 && cot == 1) rc = MLP2_GO(64, 1);
    else if (cin == 64) rc = MLP2_GO(64, 2);
    else if (cot == 1) rc = MLP2_GO(128, 1);
    else rc = MLP2_GO(128, 4);
#undef MLP2_GO
    mcp_prof_end(MCP_KERNEL_MLP, s);
    return rc;
}
