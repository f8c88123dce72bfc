// linear.hip -- per-point Linear (1x1 convolution) with fused epilogue for gfx950, for the tall-skinny shapes of the caller graph:
//
//     out[r, 0:n] = act( sum_i W_i . x_i[r] + b ) [+ res[r]],     act(v) = v > 0 ? v : slope * v   (slope 1 = none, 0 = ReLU)
//
// rows 10^4 .. 2*10^5, K = sum K_i <= ~2000, n <= 256.  The reference runs these as Conv1d / Linear launches followed by
// LeakyReLU / PReLU / residual launches, and concatenates the inputs first (torch.cat of encoder, fusion and upsampled features,
// mocopci.py:186-187, :846-847; Conv1d wrapper :1111-1127).  Here
//   * a wave owns 32 rows on the MFMA column; the K loop runs in chunks of 32 input channels: the wave loads its own 32 x 32 tile
//     of x with coalesced float4 reads, turns it into accumulator layout through a private LDS tile, splits it (mfma_split.h)
//     and multiplies it into n/32 accumulator tiles on the bf16 matrix pipe;
//   * the inputs may be up to three separate tensors with their own row strides (the pieces of a concatenation, or column
//     slices of wider tensors): chunk c simply reads from the piece it falls into, nothing is concatenated;
//   * the weight image ([chunk][out tile][k-step][piece][lane] x 16 B, split once by mcp_linear_pack) is streamed through LDS,
//     one chunk double-buffered, shared by the 8 waves of the workgroup;
//   * bias is the accumulators' initial value; activation, residual and the store happen in registers.
#include "common.h"
#include "mfma_split.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int MAXSEG = 3;
constexpr long long MCP_LINEAR_SPLITK_ROWS = 16384;  // below this the split-K kernel (see linear_splitk_kernel)

struct Segs {
    const float *x[MAXSEG];
    int stride[MAXSEG];
    int chunks[MAXSEG];  // 32-channel chunks of each piece (K_i / 32, K_i rounded up)
    int k[MAXSEG];       // valid channels of each piece (a multiple of 4)
};

// floats of one chunk image: NT out tiles x 2 k-steps x 3 pieces x 64 lanes x uint4
__host__ __device__ constexpr int chunk_floats(int nt) { return nt * 2 * 3 * 64 * 4; }
// floats of one LDS stage buffer: kc chunks rounded up to whole passes of the workgroup of nw waves (uint4 per thread per pass)
__host__ __device__ constexpr size_t stage_floats(int nt, int kc, int nw) { return (size_t)((kc * chunk_floats(nt) / 4 + 64 * nw - 1) / (64 * nw)) * (64 * nw) * 4; }

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

// KC: 32-channel chunks of the weight image per LDS stage (one barrier per stage).  With KC = 1 a long K at a narrow output is
// bound by the hand-off (K = 536 -> 64: 17 barriers of 24 MFMAs each); KC = 2..4 divides that.
// Rows sit on the MFMA column, so in operand layout a lane owns a K-slice of ONE row: read or written straight from memory
// every lane of a load touches a different cache line (16 B of it), the lines of ~100 KB of tiles in flight per CU do not
// survive in the 32 KB L1 until their other seven pieces are asked for, and the kernel moved 2.3 TB/s.  Both the x tiles and
// the output tiles therefore go through a wave-private, padded LDS tile: global accesses are coalesced (8 lanes cover a row's
// 128 B, 8 rows per instruction), the transposition to / from operand layout is four ds_write_b128 + four ds_read_b128 per tile.
// No barrier is involved: a wave's LDS instructions execute in order.  x tiles are requested three chunks ahead.
constexpr int XP = 36;              // floats per padded tile row (32 + 4: conflict-free for both access patterns)
constexpr int XT = 32 * XP;         // one 32 x 32 tile

template <int NT, int KC, int NW>
__global__ __launch_bounds__(64 * NW, 1) void linear_kernel(long long rows, int n, Segs sg, int nseg, int total_chunks, float slope,
                                                             const float *__restrict__ packed, int cf_total, const float *__restrict__ res, int rs_,
                                                             float *__restrict__ out, int os_) {
    // A workgroup owns NT output tiles (32 columns each) starting at tile tb = blockIdx.y * NT of the image's cf_total / 1536 tiles
    // per chunk: outputs wider than 256 run as grid.y column blocks of 128, each re-reading (and re-splitting) its rows of x.
    constexpr int CF = chunk_floats(NT), CH4 = CF / 4;  // floats / uint4 per chunk of this workgroup's NT tiles
    const int tb = blockIdx.y * NT;
    constexpr int LOADS = (KC * CH4 + 64 * NW - 1) / (64 * NW);
    constexpr int SF = LOADS * 64 * NW * 4;  // floats per stage buffer: KC chunks, rounded up to whole passes of the workgroup
    extern __shared__ __attribute__((aligned(16))) float lds[];  // weights [2][SF] | tiles [NW][XT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    const int cr = lane >> 3, cq = lane & 7;  // coalesced coordinates: row cr (+8j) of the tile, float4 number cq of its 32 channels
    float *tile = lds + 2 * SF + wave * XT;  // one tile per wave is enough: a wave's LDS instructions execute in order, so the
                                             // read of chunk c is done before the write of chunk c+1 lands on the same addresses
    const long long row0 = ((long long)blockIdx.x * NW + wave) * 32;

    f32x16 acc[NT];
    {
        const float *bi = packed + (size_t)total_chunks * cf_total + tb * 32;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = bi[(t * 2 + h) * 16 + r];
    }
    // Weight stage st = chunks KC*st .. KC*st + KC - 1 (the last one may be short): contiguous in the image.  Branch-free: a load
    // under a branch makes the compiler wait for ALL outstanding loads (s_waitcnt vmcnt(0)) at every use, which serialises the
    // whole prefetch pipeline; entries past the stage's end re-read its last one (never used).  Macros rather than lambdas: with
    // the staging registers captured by reference the array ended up in scratch memory.
    u32x4 pre[LOADS];  // native vector type: an array of HIP's uint4 structs here ended up in scratch memory
#define LIN_FETCH(st_)                                                                                          \
    {                                                                                                           \
        const u32x4 *src_ = reinterpret_cast<const u32x4 *>(packed + (size_t)(st_) * KC * cf_total + tb * (CF / NT));  \
        const int valid_ = min(KC, total_chunks - (st_) * KC);                                                  \
        _Pragma("unroll") for (int u = 0; u < LOADS; ++u) {                                                     \
            const int e_ = tid + u * 64 * NW, kk_ = min(e_ / CH4, valid_ - 1), w_ = e_ - (e_ / CH4) * CH4;      \
            pre[u] = src_[(size_t)kk_ * (cf_total / 4) + w_];                                                   \
        }                                                                                                       \
    }
#define LIN_STASH(buf_)                                                                                         \
    {                                                                                                           \
        u32x4 *dst_ = reinterpret_cast<u32x4 *>(lds + (size_t)(buf_) * SF);                                    \
        _Pragma("unroll") for (int u = 0; u < LOADS; ++u) dst_[tid + u * 64 * NW] = pre[u];                     \
    }
    struct Tile { float4 v[4]; };
    // rows past the end re-read the last row (their results are never stored)
    long long xrow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xrow[j] = min(row0 + 8 * j + cr, rows - 1);
    // chunk `issued` of the concatenated K axis -> (piece, chunk within the piece) by comparisons with the pieces' first chunks:
    // scalar selects, no branch, so the loop body stays one basic block (see fetch)
    const int c1 = sg.chunks[0], c2 = c1 + (nseg > 1 ? sg.chunks[1] : 0);
    int issued = 0;
    auto request = [&](Tile &g) {
        // Past the last chunk (a short last stage is padded to KC chunks) and past a piece's width the tile is zero: the load
        // then re-reads the row's first float4 and the value is masked off -- an AND rather than a select, so that the compiler
        // cannot turn it back into a load under a branch.
        const bool live_chunk = issued < total_chunks;
        const int ci = live_chunk ? issued : 0;
        const int sgi = (ci >= c1) + (ci >= c2);
        const int cci = ci - (sgi == 0 ? 0 : sgi == 1 ? c1 : c2);
        const float *xb = sgi == 0 ? sg.x[0] : sgi == 1 ? sg.x[1] : sg.x[2];
        const int stride = sgi == 0 ? sg.stride[0] : sgi == 1 ? sg.stride[1] : sg.stride[2];
        const int kk = sgi == 0 ? sg.k[0] : sgi == 1 ? sg.k[1] : sg.k[2];
        const bool ok = live_chunk && 32 * cci + 4 * cq < kk;  // channels of a piece are a multiple of 4
        const uint32_t mask = ok ? 0xFFFFFFFFu : 0u;
        const float *base = xb + (ok ? 32 * cci + 4 * cq : 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint4 v = *reinterpret_cast<const uint4 *>(base + xrow[j] * stride);
            g.v[j] = make_float4(__uint_as_float(v.x & mask), __uint_as_float(v.y & mask), __uint_as_float(v.z & mask), __uint_as_float(v.w & mask));
        }
        ++issued;
    };
    auto to_lds = [&](const Tile &g) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4 *>(tile + (8 * j + cr) * XP + 4 * cq) = g.v[j];
    };
    // operand layout: registers 4g..4g+3 <- channels 8g + 4h .. +3 of row `col`
    auto from_lds = [&](f32x16 &a) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *reinterpret_cast<const float4 *>(tile + col * XP + 8 * g + 4 * h);
            a[4 * g + 0] = v.x; a[4 * g + 1] = v.y; a[4 * g + 2] = v.z; a[4 * g + 3] = v.w;
        }
    };
    Tile ga, gb, gc;  // chunks c+1, c+2, c+3 while chunk c is multiplied
    request(ga);
    to_lds(ga);
    request(ga);
    request(gb);
    LIN_FETCH(0)
    LIN_STASH(0)
    const int stages = (total_chunks + KC - 1) / KC;
    for (int st = 0; st < stages; ++st) {
        const int cur = st & 1;
        // No branch in the loop body: the number of loads in flight at every wait is then a compile-time constant and the waits
        // are exact counts.  The last stage fetches itself again (nobody reads the other buffer afterwards), and a short last
        // stage runs all KC chunks: zero x tiles (request) against re-read, finite weights (fetch) add nothing.
        LIN_FETCH(min(st + 1, stages - 1))
        __syncthreads();  // stage st is complete in buffer cur; nobody reads buffer cur^1 any more
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            request(gc);
            f32x16 xa;
            from_lds(xa);
            to_lds(ga);  // the next chunk's tile (after the last chunk: zeros nobody uses)
            ga = gb;
            gb = gc;
            McpSplit3 xs[2];
            xs[0] = mcp_split_kstep(xa, 0);
            xs[1] = mcp_split_kstep(xa, 1);
            const uint4 *wc = reinterpret_cast<const uint4 *>(lds + (size_t)cur * SF + (size_t)k * CF) + lane;
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mcp_tile_split<2>(wc + (size_t)t * 2 * 3 * 64, xs, acc[t]);
        }
        LIN_STASH(cur ^ 1)
    }
#undef LIN_FETCH
#undef LIN_STASH
    // epilogue: activation in registers, one 32-channel tile at a time through the wave's LDS tile, coalesced residual + store
    const bool vec = !(os_ & 3) && !(((uintptr_t)out) & 15) && !(res && ((rs_ & 3) || (((uintptr_t)res) & 15)));
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float a = acc[t][4 * g + u];
                v[u] = a > 0.f ? a : a * slope;
            }
            *reinterpret_cast<float4 *>(tile + col * XP + 8 * g + 4 * h) = make_float4(v[0], v[1], v[2], v[3]);
        }
        const int ch = 32 * (tb + t) + 4 * cq;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long row = row0 + 8 * j + cr;
            const float4 v4 = *reinterpret_cast<const float4 *>(tile + (8 * j + cr) * XP + 4 * cq);
            if (row >= rows || ch >= n) continue;
            float v[4] = {v4.x, v4.y, v4.z, v4.w};
            float *orow = out + row * os_;
            const float *rrow = res ? res + row * rs_ : nullptr;
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
}

// ---- few rows, long K: split-K (r3) -----------------------------------------------------------------------------------------
// The layers of the lower pyramid levels have 1024 .. 8192 rows and K up to 2072 (PointConv projections (3+D)*8 -> C, the q / kv
// projections of cross_block3, ...): linear_kernel gives a wave 32 rows and the whole K loop, i.e. 32 .. 256 waves on 1024 SIMDs,
// and the library's f32 GEMMs take 15 .. 65 us for a few GFLOP there (one 63 us call at level 3).  Here the FOUR waves of a
// workgroup share one 32-row tile and split its K chunks between them (wave w takes chunks w, w+4, ...), a workgroup covers
// NT <= 2 output tiles (grid.y runs over the rest), so 4096 rows x 256 columns are 512 workgroups; the weight pieces come
// straight from the packed image in L2 into registers (every wave reads different chunks: nothing to share through LDS, no
// barrier in the K loop), one chunk ahead, like the x tile; the four partial accumulators are summed in a fixed order through
// LDS by wave 0, which then runs the same epilogue as linear_kernel.  Same split-bf16 arithmetic, deterministic.
template <int NT>
__global__ __launch_bounds__(256) void linear_splitk_kernel(long long rows, int n, Segs sg, int nseg, int total_chunks, float slope,
                                                            const float *__restrict__ packed, int cf_total, const float *__restrict__ res, int rs_,
                                                            float *__restrict__ out, int os_) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // x tiles [4][XT] | partial accumulators [3][NT][16][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, col = lane & 31;
    const int cr = lane >> 3, cq = lane & 7;
    float *tile = lds + wave * XT;
    float *part = lds + 4 * XT;
    const int tb = blockIdx.y * NT;
    const long long row0 = (long long)blockIdx.x * 32;
    f32x16 acc[NT];
    {
        const float *bi = packed + (size_t)total_chunks * cf_total + tb * 32;  // wave 0 starts from the bias, the others from zero
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = wave == 0 ? bi[(t * 2 + h) * 16 + r] : 0.f;
    }
    long long xrow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xrow[j] = min(row0 + 8 * j + cr, rows - 1);
    const int c1 = sg.chunks[0], c2 = c1 + (nseg > 1 ? sg.chunks[1] : 0);
    struct Tile { float4 v[4]; };
    struct Wts { u32x4 w[NT][2][3]; };
    // chunk ci of the concatenated K axis: x tile (masked past a piece's width) and the weight pieces of this workgroup's tiles;
    // `live` false: a dummy re-read of chunk 0 whose x tile is zero (keeps the loop body branch-free: exact wait counts)
    auto request = [&](int ci_, bool live, Tile &g, Wts &wt) {
        const int ci = live ? ci_ : 0;
        const int sgi = (ci >= c1) + (ci >= c2);
        const int cci = ci - (sgi == 0 ? 0 : sgi == 1 ? c1 : c2);
        const float *xb = sgi == 0 ? sg.x[0] : sgi == 1 ? sg.x[1] : sg.x[2];
        const int stride = sgi == 0 ? sg.stride[0] : sgi == 1 ? sg.stride[1] : sg.stride[2];
        const int kk = sgi == 0 ? sg.k[0] : sgi == 1 ? sg.k[1] : sg.k[2];
        const bool ok = live && 32 * cci + 4 * cq < kk;
        const uint32_t mask = ok ? 0xFFFFFFFFu : 0u;
        const float *base = xb + (ok ? 32 * cci + 4 * cq : 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint4 v = *reinterpret_cast<const uint4 *>(base + xrow[j] * stride);
            g.v[j] = make_float4(__uint_as_float(v.x & mask), __uint_as_float(v.y & mask), __uint_as_float(v.z & mask), __uint_as_float(v.w & mask));
        }
        const u32x4 *wc = reinterpret_cast<const u32x4 *>(packed + (size_t)ci * cf_total) + (size_t)tb * 2 * 3 * 64 + lane;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) wt.w[t][s2][pc] = wc[(size_t)((t * 2 + s2) * 3 + pc) * 64];
    };
    const int mine = total_chunks > wave ? (total_chunks - wave + 3) / 4 : 0;  // chunks wave, wave + 4, ...
    Tile xa_, xb_;
    Wts wa_, wb_;
    request(wave, mine > 0, xa_, wa_);
    for (int i = 0; i < mine; ++i) {
        request(wave + 4 * (i + 1), i + 1 < mine, xb_, wb_);
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4 *>(tile + (8 * j + cr) * XP + 4 * cq) = xa_.v[j];
        f32x16 xa;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *reinterpret_cast<const float4 *>(tile + col * XP + 8 * g + 4 * h);
            xa[4 * g + 0] = v.x; xa[4 * g + 1] = v.y; xa[4 * g + 2] = v.z; xa[4 * g + 3] = v.w;
        }
        McpSplit3 xs[2];
        xs[0] = mcp_split_kstep(xa, 0);
        xs[1] = mcp_split_kstep(xa, 1);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const uint4 w1 = __builtin_bit_cast(uint4, wa_.w[t][s2][0]), w2 = __builtin_bit_cast(uint4, wa_.w[t][s2][1]),
                            w3 = __builtin_bit_cast(uint4, wa_.w[t][s2][2]);
                acc[t] = mcp_mfma_bf16(w3, xs[s2].p1, acc[t]);  // small terms first, as mcp_mfma_split
                acc[t] = mcp_mfma_bf16(w1, xs[s2].p3, acc[t]);
                acc[t] = mcp_mfma_bf16(w2, xs[s2].p2, acc[t]);
                acc[t] = mcp_mfma_bf16(w2, xs[s2].p1, acc[t]);
                acc[t] = mcp_mfma_bf16(w1, xs[s2].p2, acc[t]);
                acc[t] = mcp_mfma_bf16(w1, xs[s2].p1, acc[t]);
            }
        xa_ = xb_;
        wa_ = wb_;
    }
    // fixed-order sum of the four partial accumulators: waves 1..3 publish, wave 0 adds 1, 2, 3
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) part[(((wave - 1) * NT + t) * 16 + r) * 64 + lane] = acc[t][r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] += part[((w * NT + t) * 16 + r) * 64 + lane];
    const bool vec = !(os_ & 3) && !(((uintptr_t)out) & 15) && !(res && ((rs_ & 3) || (((uintptr_t)res) & 15)));
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float a = acc[t][4 * g + u];
                v[u] = a > 0.f ? a : a * slope;
            }
            *reinterpret_cast<float4 *>(tile + col * XP + 8 * g + 4 * h) = make_float4(v[0], v[1], v[2], v[3]);
        }
        const int ch = 32 * (tb + t) + 4 * cq;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long row = row0 + 8 * j + cr;
            const float4 v4 = *reinterpret_cast<const float4 *>(tile + (8 * j + cr) * XP + 4 * cq);
            if (row >= rows || ch >= n) continue;
            float v[4] = {v4.x, v4.y, v4.z, v4.w};
            float *orow = out + row * os_;
            const float *rrow = res ? res + row * rs_ : nullptr;
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
}

template <int NT>
int launch_linear_splitk(long long rows, int n, int nt_total, const Segs &sg, int nseg, int total_chunks, float slope, const float *packed, const float *res,
                         int rs_, float *out, int os_, hipStream_t s) {
    const size_t lds = (4 * (size_t)XT + 3 * (size_t)NT * 16 * 64) * sizeof(float);
    hipLaunchKernelGGL(linear_splitk_kernel<NT>, dim3((unsigned)((rows + 31) / 32), nt_total / NT), dim3(256), lds, s, rows, n, sg, nseg, total_chunks, slope,
                       packed, chunk_floats(nt_total), res, rs_, out, os_);
    return mcp_launch_status();
}

template <int NT, int KC, int NW>
int launch_linear_kc(long long rows, int n, const Segs &sg, int nseg, int total_chunks, float slope, const float *packed, const float *res, int rs_,
                     float *out, int os_, hipStream_t s, int col_blocks = 1) {
    auto kern = linear_kernel<NT, KC, NW>;
    const size_t lds = (2 * stage_floats(NT, KC, NW) + (size_t)NW * XT) * sizeof(float);
    static_assert((2 * stage_floats(NT, KC, NW) + (size_t)NW * XT) * sizeof(float) <= 160 * 1024, "LDS budget");
    static McpPerDeviceOnce attr_once;
    if (attr_once.need()) {
        { const hipError_t attr_e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (attr_e_ != hipSuccess) return (int)attr_e_; }
        attr_once.done();
    }
    const long long groups = (rows + 32LL * NW - 1) / (32LL * NW);
    hipLaunchKernelGGL(kern, dim3((unsigned)groups, col_blocks), dim3(64 * NW), lds, s, rows, n, sg, nseg, total_chunks, slope, packed,
                       chunk_floats(NT * col_blocks), res, rs_, out, os_);
    return mcp_launch_status();
}

// stage depth by shape: long K gets KC chunks per barrier (two stages of 12 KB per chunk per 32 outputs beside the tiles).
// Workgroup size by rows: 8 waves (256 rows) share one weight stream where there are rows enough to fill the chip several times
// over; below that 4-wave workgroups (128 rows) -- 32768 rows are 128 eight-wave workgroups on 256 CUs.
template <int NT, int NW>
int launch_linear_nw(long long rows, int n, const Segs &sg, int nseg, int total_chunks, float slope, const float *packed, const float *res, int rs_,
                     float *out, int os_, hipStream_t s, int col_blocks = 1) {
    constexpr int KC = NT <= 2 ? 4 : NT <= 4 ? 2 : 1;
    if (KC > 1 && total_chunks >= 2 * KC)
        return launch_linear_kc<NT, KC, NW>(rows, n, sg, nseg, total_chunks, slope, packed, res, rs_, out, os_, s, col_blocks);
    return launch_linear_kc<NT, 1, NW>(rows, n, sg, nseg, total_chunks, slope, packed, res, rs_, out, os_, s, col_blocks);
}
// outputs wider than 256 columns (a multiple of 128): column blocks of four tiles over grid.y
int launch_linear_blocked(long long rows, int n, const Segs &sg, int nseg, int total_chunks, float slope, const float *packed, const float *res,
                          int rs_, float *out, int os_, hipStream_t s) {
    const int col_blocks = ((n + 31) / 32) / 4;
    if (rows >= 131072) return launch_linear_nw<4, 8>(rows, n, sg, nseg, total_chunks, slope, packed, res, rs_, out, os_, s, col_blocks);
    return launch_linear_nw<4, 4>(rows, n, sg, nseg, total_chunks, slope, packed, res, rs_, out, os_, s, col_blocks);
}
template <int NT>
int launch_linear(long long rows, int n, const Segs &sg, int nseg, int total_chunks, float slope, const float *packed, const float *res, int rs_,
                  float *out, int os_, hipStream_t s) {
    // NT > 4 (outputs wider than 128 columns in one pass): 4-wave workgroups whatever the row count -- an 8-wave workgroup caps a
    // wave at 256 registers and NT = 8 then ran with 10 of its accumulators in scratch (round 4's ISA); at 4 waves it has 512.
    if constexpr (NT > 4) return launch_linear_nw<NT, 4>(rows, n, sg, nseg, total_chunks, slope, packed, res, rs_, out, os_, s);
    else {
        if (rows >= 131072) return launch_linear_nw<NT, 8>(rows, n, sg, nseg, total_chunks, slope, packed, res, rs_, out, os_, s);
        return launch_linear_nw<NT, 4>(rows, n, sg, nseg, total_chunks, slope, packed, res, rs_, out, os_, s);
    }
}

// Narrow-output Linear with the activation on its INPUT:  out[r, j] = b[j] + sum_k W[j, k] * act(x[r, k]),  j < n <= 4
// (act(v) = v > 0 ? v : slope v).  The tail of Mlp_T where only a flow is read: PReLU, then the (4C -> 3) map that fc2 and
// mapping_xyz fold into (mocopci.py:1561-1565, :566-567).  The library takes 67 us for 12288 x 1024 -> 3 (48 workgroups) after a
// separate PReLU launch; this is a streaming reduction: a wave owns RW rows at a time, lanes run along K with float4 loads, W sits
// in registers (K <= 1024: 3 x 16 floats per lane), three DPP wave sums per row.
constexpr int NARROW_RW = 4;
template <int KQ>  // float4 per lane per row: K = 256 * KQ
__global__ __launch_bounds__(256) void linear_narrow_kernel(long long rows, int n, const float *__restrict__ x, int xs_, const float *__restrict__ w,
                                                            const float *__restrict__ b, float slope, float *__restrict__ out, int os_) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int K = 256 * KQ;
    float4 wr[4][KQ];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < KQ; ++u) wr[j][u] = j < n ? *reinterpret_cast<const float4 *>(w + (size_t)j * K + 4 * (lane + 64 * u)) : make_float4(0.f, 0.f, 0.f, 0.f);
    const long long nblk = (rows + NARROW_RW - 1) / NARROW_RW;
    for (long long blk = (long long)blockIdx.x * 4 + wave; blk < nblk; blk += (long long)gridDim.x * 4) {
        float4 xv[NARROW_RW][KQ];
#pragma unroll
        for (int r = 0; r < NARROW_RW; ++r) {
            const long long row = min(blk * NARROW_RW + r, rows - 1);
#pragma unroll
            for (int u = 0; u < KQ; ++u) xv[r][u] = *reinterpret_cast<const float4 *>(x + row * xs_ + 4 * (lane + 64 * u));
        }
#pragma unroll
        for (int r = 0; r < NARROW_RW; ++r) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < KQ; ++u) {
                const float v[4] = {xv[r][u].x, xv[r][u].y, xv[r][u].z, xv[r][u].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float a = v[c] > 0.f ? v[c] : v[c] * slope;
                    const float wj[4] = {c == 0 ? wr[0][u].x : c == 1 ? wr[0][u].y : c == 2 ? wr[0][u].z : wr[0][u].w,
                                         c == 0 ? wr[1][u].x : c == 1 ? wr[1][u].y : c == 2 ? wr[1][u].z : wr[1][u].w,
                                         c == 0 ? wr[2][u].x : c == 1 ? wr[2][u].y : c == 2 ? wr[2][u].z : wr[2][u].w,
                                         c == 0 ? wr[3][u].x : c == 1 ? wr[3][u].y : c == 2 ? wr[3][u].z : wr[3][u].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(wj[j], a, acc[j]);
                }
            }
            const long long row = blk * NARROW_RW + r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float sum = acc[j];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
                if (lane == 0 && j < n && row < rows) out[row * os_ + j] = sum + (b ? b[j] : 0.f);
            }
        }
    }
}

int count_chunks(int nseg, const int *k) {
    int c = 0;
    for (int i = 0; i < nseg; ++i) c += (k[i] + 31) / 32;
    return c;
}

}  // namespace

MCP_EXPORT int mcp_linear_narrow(long long rows, int k, int n, const float *x, int x_stride, const float *w, const float *b, float in_slope,
                                 float *out, int out_stride, mcp_stream_t stream) {
    MCP_CHECK_ARGS(rows > 0 && x && w && out);
    if (n < 1 || n > 4 || (k != 256 && k != 512 && k != 1024)) return MCP_ERR_UNSUPPORTED;
    if (((((uintptr_t)x) | ((uintptr_t)w)) & 15) || (x_stride & 3) || x_stride < k || out_stride < n) return MCP_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const long long nblk = (rows + NARROW_RW - 1) / NARROW_RW;
    const unsigned grid = (unsigned)min((nblk + 3) / 4, 4096LL);
    mcp_prof_begin(MCP_KERNEL_LINEAR, s);
    if (k == 256) hipLaunchKernelGGL(linear_narrow_kernel<1>, dim3(grid), dim3(256), 0, s, rows, n, x, x_stride, w, b, in_slope, out, out_stride);
    else if (k == 512) hipLaunchKernelGGL(linear_narrow_kernel<2>, dim3(grid), dim3(256), 0, s, rows, n, x, x_stride, w, b, in_slope, out, out_stride);
    else hipLaunchKernelGGL(linear_narrow_kernel<4>, dim3(grid), dim3(256), 0, s, rows, n, x, x_stride, w, b, in_slope, out, out_stride);
    mcp_prof_end(MCP_KERNEL_LINEAR, s);
    return mcp_launch_status();
}

MCP_EXPORT int mcp_linear_packed_floats(int n, int nseg, const int *k_seg) {
    if (n <= 0 || n > 2048 || nseg < 1 || nseg > MAXSEG || !k_seg) return 0;
    for (int i = 0; i < nseg; ++i)
        if (k_seg[i] <= 0 || (k_seg[i] & 3)) return 0;
    const int nt = (n + 31) / 32;
    if (nt == 5 || nt == 7) return 0;  // output widths the kernel is instantiated for: up to 128, 192, 256 ...
    if (nt > 8 && (nt & 3)) return 0;  // ... and wider ones in column blocks of 128: a multiple of 128 (the last 31 columns may be missing)
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

MCP_EXPORT int mcp_linear_as(long long rows, long long policy_rows, int n, int nseg, const float *const *x, const int *x_stride, const int *k_seg,
                             float slope, const float *packed, const float *res, int res_stride, float *out, int out_stride, mcp_stream_t stream) {
    MCP_CHECK_ARGS(rows > 0 && policy_rows > 0 && x && x_stride && k_seg && packed && out);
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
    if (policy_rows < MCP_LINEAR_SPLITK_ROWS && total >= 4) {  // few rows: four waves split the K chunks of one 32-row tile
        rc = (nt & 1) ? launch_linear_splitk<1>(rows, n, nt, sg, nseg, total, slope, packed, res, res_stride, out, out_stride, s)
                      : launch_linear_splitk<2>(rows, n, nt, sg, nseg, total, slope, packed, res, res_stride, out, out_stride, s);
        mcp_prof_end(MCP_KERNEL_LINEAR, s);
        return rc;
    }
    switch (nt) {
        case 1: rc = launch_linear<1>(rows, n, sg, nseg, total, slope, packed, res, res_stride, out, out_stride, s); break;
        case 2: rc = launch_linear<2>(rows, n, sg, nseg, total, slope, packed, res, res_stride, out, out_stride, s); break;
        case 3: rc = launch_linear<3>(rows, n, sg, nseg, total, slope, packed, res, res_stride, out, out_stride, s); break;
        case 4: rc = launch_linear<4>(rows, n, sg, nseg, total, slope, packed, res, res_stride, out, out_stride, s); break;
        case 6: rc = launch_linear<6>(rows, n, sg, nseg, total, slope, packed, res, res_stride, out, out_stride, s); break;
        case 8: rc = launch_linear<8>(rows, n, sg, nseg, total, slope, packed, res, res_stride, out, out_stride, s); break;
        default: rc = nt > 8 ? launch_linear_blocked(rows, n, sg, nseg, total, slope, packed, res, res_stride, out, out_stride, s) : MCP_ERR_UNSUPPORTED;
    }
    mcp_prof_end(MCP_KERNEL_LINEAR, s);
    return rc;
}

MCP_EXPORT int mcp_linear(long long rows, int n, int nseg, const float *const *x, const int *x_stride, const int *k_seg, float slope,
                          const float *packed, const float *res, int res_stride, float *out, int out_stride, mcp_stream_t stream) {
    return mcp_linear_as(rows, rows, n, nseg, x, x_stride, k_seg, slope, packed, res, res_stride, out, out_stride, stream);
}
