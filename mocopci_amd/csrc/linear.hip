// linear.hip -- per-point Linear (1x1 convolution) with fused epilogue for gfx950, for the tall-skinny shapes of the caller graph:
//
//     out[r, 0:n] = act( sum_i W_i . x_i[r] + b ) [+ res[r]],     act(v) = v > 0 ? v : slope * v   (slope 1 = none, 0 = ReLU)
//
// rows 10^4 .. 2*10^5, K = sum K_i <= ~2000, n <= 256.  The reference runs these as Conv1d / Linear launches followed by
// LeakyReLU / PReLU / residual launches, and concatenates the inputs first (torch.cat of encoder, fusion and upsampled features,
// mocopci.py:186-187, :846-847; Conv1d wrapper :1111-1127).  Here
//   * a wave owns 32 rows on the MFMA column; the K loop runs in chunks of 32 input channels: the wave loads its own 32 x 32 tile
//     of x straight from memory in accumulator layout (four float4 per lane), splits it (mfma_split.h) and multiplies it into
//     n/32 accumulator tiles on the bf16 matrix pipe;
//   * the inputs may be up to three separate tensors with their own row strides (the pieces of a concatenation, or column
//     slices of wider tensors): chunk c simply reads from the piece it falls into, nothing is concatenated;
//   * the weight image ([chunk][out tile][k-step][piece][lane] x 16 B, split once by mcp_linear_pack) is streamed through LDS,
//     one chunk double-buffered, shared by the 8 waves of the workgroup;
//   * bias is the accumulators' initial value; activation, residual and the store happen in registers.
#include "common.h"
#include "mfma_split.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NW = 8, MAXSEG = 3;

struct Segs {
    const float *x[MAXSEG];
    int stride[MAXSEG];
    int chunks[MAXSEG];  // 32-channel chunks of each piece (K_i / 32, K_i rounded up)
    int k[MAXSEG];       // valid channels of each piece (a multiple of 4)
};

// floats of one chunk image: NT out tiles x 2 k-steps x 3 pieces x 64 lanes x uint4
__host__ __device__ constexpr int chunk_floats(int nt) { return nt * 2 * 3 * 64 * 4; }

// W (n x ktot) row-major; the K axis is the concatenation of the pieces, each padded to a multiple of 32 in the image
__global__ __launch_bounds__(256) void linear_pack_kernel(int n, int ktot, int nseg, int k0, int k1, int k2, const float *__restrict__ w,
                                                          const float *__restrict__ b, float *__restrict__ packed) {
    const int ks[3] = {k0, k1, k2};
    const int nt = (n + 31) / 32;
    int total_chunks = 0;
    for (int i = 0; i < nseg; ++i) total_chunks += (ks[i] + 31) / 32;
    const int cf = chunk_floats(nt);
    const int first = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    for (int e = first; e < total_chunks * nt * 2 * 64; e += stride) {
        const int lane = e & 63, s = (e >> 6) & 1, t = (e >> 7) % nt, c = (e >> 7) / nt;
        // chunk c -> (piece, chunk within the piece) -> column offset in W
        int seg = 0, cc = c, col0 = 0;
        while (seg < nseg - 1 && cc >= (ks[seg] + 31) / 32) { cc -= (ks[seg] + 31) / 32; col0 += ks[seg]; ++seg; }
        const int row = 32 * t + (lane & 31);
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ch = 32 * cc + mcp_chan_of(8 * s + i, lane >> 5);  // channel within the piece
            v[i] = (row < n && ch < ks[seg]) ? w[(size_t)row * ktot + col0 + ch] : 0.f;
        }
        const McpSplit3 sp = mcp_split8(v);
        uint4 *o = reinterpret_cast<uint4 *>(packed + (size_t)c * cf) + (size_t)(t * 2 + s) * 3 * 64 + lane;
        o[0] = sp.p1; o[64] = sp.p2; o[128] = sp.p3;
    }
    for (int e = first; e < nt * 32; e += stride) {  // bias image after the chunks: [tile][half][reg]
        const int r = e & 15, h = (e >> 4) & 1, t = e >> 5;
        const int ch = 32 * t + mcp_chan_of(r, h);
        packed[(size_t)total_chunks * cf + e] = (b && ch < n) ? b[ch] : 0.f;
    }
}

template <int NT>
__global__ __launch_bounds__(64 * NW, 1) void linear_kernel(long long rows, int n, Segs sg, int nseg, int total_chunks, float slope,
                                                             const float *__restrict__ packed, const float *__restrict__ res, int rs_,
                                                             float *__restrict__ out, int os_) {
    constexpr int CF = chunk_floats(NT), CH4 = CF / 4;
    constexpr int LOADS = (CH4 + 64 * NW - 1) / (64 * NW);
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [2][CF]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    const long long row = ((long long)blockIdx.x * NW + wave) * 32 + col;
    const bool live = row < rows;
    const long long rr = live ? row : rows - 1;

    f32x16 acc[NT];
    {
        const float *bi = packed + (size_t)total_chunks * CF;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = bi[(t * 2 + h) * 16 + r];
    }
    uint4 pre[LOADS];
    auto fetch = [&](int c) {
        const uint4 *src = reinterpret_cast<const uint4 *>(packed + (size_t)c * CF);
#pragma unroll
        for (int u = 0; u < LOADS; ++u) {
            const int e = tid + u * 64 * NW;
            if (e < CH4) pre[u] = src[e];
        }
    };
    auto stash = [&](int buf) {
        uint4 *dst = reinterpret_cast<uint4 *>(lds + (size_t)buf * CF);
#pragma unroll
        for (int u = 0; u < LOADS; ++u) {
            const int e = tid + u * 64 * NW;
            if (e < CH4) dst[e] = pre[u];
        }
    };
    // this wave's 32 x 32 tile of x for chunk (seg, cc): registers 4g..4g+3 <- channels 32cc + 8g + 4h .. +3 (zero past the piece)
    auto load_x = [&](int seg, int cc, f32x16 &a) {
        const float *xr = sg.x[seg] + rr * sg.stride[seg] + 32 * cc;
        const int kleft = sg.k[seg] - 32 * cc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = 8 * g + 4 * h;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ch < kleft) v = *reinterpret_cast<const float4 *>(xr + ch);
            a[4 * g + 0] = v.x; a[4 * g + 1] = v.y; a[4 * g + 2] = v.z; a[4 * g + 3] = v.w;
        }
    };
    int seg = 0, cc = 0;
    f32x16 xa, xn;
    load_x(0, 0, xa);
    fetch(0);
    stash(0);
    for (int c = 0; c < total_chunks; ++c) {
        const int cur = c & 1;
        // next chunk's coordinates, weights and x tile are requested before this chunk is multiplied
        int nseg_ = seg, ncc = cc + 1;
        if (ncc >= sg.chunks[seg]) { nseg_ = seg + 1; ncc = 0; }
        const bool more = c + 1 < total_chunks;
        if (more) {
            fetch(c + 1);
            load_x(nseg_, ncc, xn);
        }
        __syncthreads();  // chunk c is complete in buffer cur; nobody reads buffer cur^1 any more
        McpSplit3 xs[2];
        xs[0] = mcp_split_kstep(xa, 0);
        xs[1] = mcp_split_kstep(xa, 1);
        const uint4 *wc = reinterpret_cast<const uint4 *>(lds + (size_t)cur * CF) + lane;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = mcp_tile_split<2>(wc + (size_t)t * 2 * 3 * 64, xs, acc[t]);
        if (more) {
            stash(cur ^ 1);
            xa = xn;
            seg = nseg_;
            cc = ncc;
        }
    }
    if (!live) return;
    float *orow = out + row * os_;
    const float *rrow = res ? res + row * rs_ : nullptr;
    const bool vec = !(os_ & 3) && !(((uintptr_t)out) & 15) && !(rrow && ((rs_ & 3) || (((uintptr_t)res) & 15)));
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = 32 * t + 8 * g + 4 * h;
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float a = acc[t][4 * g + u];
                v[u] = a > 0.f ? a : a * slope;
            }
            if (vec && ch + 3 < n) {
                if (rrow) {
                    const float4 r4 = *reinterpret_cast<const float4 *>(rrow + ch);
                    v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                }
                *reinterpret_cast<float4 *>(orow + ch) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ch + u < n) orow[ch + u] = v[u] + (rrow ? rrow[ch + u] : 0.f);
            }
        }
}

template <int NT>
int launch_linear(long long rows, int n, const Segs &sg, int nseg, int total_chunks, float slope, const float *packed, const float *res, int rs_,
                  float *out, int os_, hipStream_t s) {
    auto kern = linear_kernel<NT>;
    const size_t lds = 2 * (size_t)chunk_floats(NT) * sizeof(float);
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    const long long groups = (rows + 32LL * NW - 1) / (32LL * NW);
    hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(64 * NW), lds, s, rows, n, sg, nseg, total_chunks, slope, packed, res, rs_, out, os_);
    return mcp_launch_status();
}

int count_chunks(int nseg, const int *k) {
    int c = 0;
    for (int i = 0; i < nseg; ++i) c += (k[i] + 31) / 32;
    return c;
}

}  // namespace

MCP_EXPORT int mcp_linear_packed_floats(int n, int nseg, const int *k_seg) {
    if (n <= 0 || n > 256 || nseg < 1 || nseg > MAXSEG || !k_seg) return 0;
    for (int i = 0; i < nseg; ++i)
        if (k_seg[i] <= 0 || (k_seg[i] & 3)) return 0;
    const int nt = (n + 31) / 32;
    if (nt == 5 || nt == 7) return 0;  // output widths the kernel is instantiated for: up to 128, 192, 256
    return count_chunks(nseg, k_seg) * chunk_floats(nt) + nt * 32;
}

MCP_EXPORT int mcp_linear_pack(int n, int nseg, const int *k_seg, const float *w, const float *b, float *packed, mcp_stream_t stream) {
    MCP_CHECK_ARGS(w && packed && k_seg);
    if (mcp_linear_packed_floats(n, nseg, k_seg) == 0) return MCP_ERR_UNSUPPORTED;
    if (((uintptr_t)packed) & 15) return MCP_ERR_BAD_ARG;
    int ktot = 0, ks[3] = {0, 0, 0};
    for (int i = 0; i < nseg; ++i) { ks[i] = k_seg[i]; ktot += k_seg[i]; }
    hipLaunchKernelGGL(linear_pack_kernel, dim3(128), dim3(256), 0, (hipStream_t)stream, n, ktot, nseg, ks[0], ks[1], ks[2], w, b, packed);
    return mcp_launch_status();
}

MCP_EXPORT int mcp_linear(long long rows, int n, int nseg, const float *const *x, const int *x_stride, const int *k_seg, float slope,
                          const float *packed, const float *res, int res_stride, float *out, int out_stride, mcp_stream_t stream) {
    MCP_CHECK_ARGS(rows > 0 && x && x_stride && k_seg && packed && out);
    if (mcp_linear_packed_floats(n, nseg, k_seg) == 0) return MCP_ERR_UNSUPPORTED;
    Segs sg{};
    for (int i = 0; i < nseg; ++i) {
        if (!x[i] || (((uintptr_t)x[i]) & 15) || (x_stride[i] & 3) || x_stride[i] < k_seg[i]) return MCP_ERR_BAD_ARG;
        sg.x[i] = x[i];
        sg.stride[i] = x_stride[i];
        sg.k[i] = k_seg[i];
        sg.chunks[i] = (k_seg[i] + 31) / 32;
    }
    if (out_stride < n || (res && res_stride < n) || (((uintptr_t)packed) & 15)) return MCP_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int nt = (n + 31) / 32, total = count_chunks(nseg, k_seg);
    int rc;
    mcp_prof_begin(MCP_KERNEL_LINEAR, s);
    switch (nt) {
        case 1: rc = launch_linear<1>(rows, n, sg, nseg, total, slope, packed, res, res_stride, out, out_stride, s); break;
        case 2: rc = launch_linear<2>(rows, n, sg, nseg, total, slope, packed, res, res_stride, out, out_stride, s); break;
        case 3: rc = launch_linear<3>(rows, n, sg, nseg, total, slope, packed, res, res_stride, out, out_stride, s); break;
        case 4: rc = launch_linear<4>(rows, n, sg, nseg, total, slope, packed, res, res_stride, out, out_stride, s); break;
        case 6: rc = launch_linear<6>(rows, n, sg, nseg, total, slope, packed, res, res_stride, out, out_stride, s); break;
        case 8: rc = launch_linear<8>(rows, n, sg, nseg, total, slope, packed, res, res_stride, out, out_stride, s); break;
        default: rc = MCP_ERR_UNSUPPORTED;
    }
    mcp_prof_end(MCP_KERNEL_LINEAR, s);
    return rc;
}
