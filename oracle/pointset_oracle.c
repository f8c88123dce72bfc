/*
 * oracle/pointset_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement (OpenMP over independent queries in the KNN loops only) of the reference's point-set operators,
 * written from the text of the reference's CUDA kernels and Python layers
 * (citations are file:line under the reference tree).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (mocopci_amd/) never does.
 *
 * Parity pinning: the reference ships no tests or golden vectors for these
 * kernels and its CUDA sources cannot be built here
 * (nvcc absent, THC/THC.h gone).  K1..K9 below are therefore pinned only by
 * this restatement of the .cu text ("parity unpinned" against a CUDA run);
 * the Python-level layers (knn_point, cosine knn, PointConv, cross layers,
 * UpsampleFlow, fusion, TransformerBlock, full forward) are pinned by golden
 * fixtures generated from the reference's own Python (tests/golden/).
 *
 * Floating-point canon.  Every sum of three products "a*a + b*b + c*c" in the
 * .cu files is evaluated here as fmaf(c,c, fmaf(b,b, a*a)): a left-to-right
 * parse with each add contracted into an FMA, which is what nvcc -O2 emits
 * with its default --fmad=true.  The HIP kernels use the same explicit
 * sequence; both sides are built with -ffp-contract=off so nothing else is
 * contracted.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

#include <omp.h>
ORC_API void orc_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
ORC_API int orc_get_threads(void) { return omp_get_max_threads(); }

static inline float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
    /* (a-b)^2 summed, canon order; a is the "second" operand in the .cu text only by name */
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

/* pointnet2/src/cuda_utils.h:10-14  opt_n_threads */
ORC_API int orc_opt_n_threads(int work_size) {
    const int pow_2 = (int)(log((double)work_size) / log(2.0));
    int v = 1 << pow_2;
    if (v > 1024) v = 1024;
    if (v < 1) v = 1;
    return v;
}

/* K1: pointnet2/src/sampling_gpu.cu:93-209 (kernel), :211-253 (launcher).
 * Literal simulation: block_size virtual threads, per-thread strided scan with
 * strict '>' (:136-137), then the LDS halving tree where __update (:86-91)
 * keeps slot idx1 unless v2 > v1.  temp is read and written exactly as the
 * kernel does (caller pre-fills 1e10, pointnet2_utils.py:26). */
ORC_API int orc_fps(const float *xyz, float *temp, int *idxs, int b, int n, int m) {
    if (m <= 0) return 0;
    int bs = orc_opt_n_threads(n);
    /* launcher switch (:217-243): any value not in the list -> <512> with n_threads threads;
       opt_n_threads only returns powers of two <= 1024 so the default never fires. */
    float *dists = (float *)malloc(sizeof(float) * bs);
    int *dists_i = (int *)malloc(sizeof(int) * bs);
    for (int bi = 0; bi < b; ++bi) {
        const float *d = xyz + (size_t)bi * n * 3;
        float *t = temp + (size_t)bi * n;
        int *o = idxs + (size_t)bi * m;
        int old = 0;
        o[0] = 0;
        for (int j = 1; j < m; ++j) {
            float x1 = d[old * 3 + 0], y1 = d[old * 3 + 1], z1 = d[old * 3 + 2];
            for (int tid = 0; tid < bs; ++tid) {
                int besti = 0;
                float best = -1.f;
                for (int k = tid; k < n; k += bs) {
                    float dd = sqdist3(d[k * 3 + 0], d[k * 3 + 1], d[k * 3 + 2], x1, y1, z1);
                    float d2 = fminf(dd, t[k]);
                    t[k] = d2;
                    besti = d2 > best ? k : besti;
                    best = d2 > best ? d2 : best;
                }
                dists[tid] = best;
                dists_i[tid] = besti;
            }
            for (int h = bs / 2; h >= 1; h >>= 1) {
                for (int tid = 0; tid < h; ++tid) {
                    float v1 = dists[tid], v2 = dists[tid + h];
                    int i1 = dists_i[tid], i2 = dists_i[tid + h];
                    dists[tid] = fmaxf(v1, v2);
                    dists_i[tid] = v2 > v1 ? i2 : i1;
                }
            }
            old = dists_i[0];
            o[j] = old;
        }
    }
    free(dists);
    free(dists_i);
    return 0;
}

/* K2: sampling_gpu.cu:8-24   out[b,c,m] = points[b,c,idx[b,m]] */
ORC_API int orc_gather(const float *points, const int *idx, float *out, int b, int c, int n, int m) {
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci)
            for (int p = 0; p < m; ++p)
                out[((size_t)bi * c + ci) * m + p] = points[((size_t)bi * c + ci) * n + idx[(size_t)bi * m + p]];
    return 0;
}

/* K3: sampling_gpu.cu:46-63   scatter-add (sequential order here; the CUDA atomics are unordered) */
ORC_API int orc_gather_grad(const float *grad_out, const int *idx, float *grad_points, int b, int c, int n, int m) {
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci)
            for (int p = 0; p < m; ++p)
                grad_points[((size_t)bi * c + ci) * n + idx[(size_t)bi * m + p]] += grad_out[((size_t)bi * c + ci) * m + p];
    return 0;
}

/* K5: group_points_gpu.cu:47-66   out[b,c,s,k] = points[b,c,idx[b,s,k]] */
ORC_API int orc_group(const float *points, const int *idx, float *out, int b, int c, int n, int npoints, int nsample) {
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci)
            for (int s = 0; s < npoints; ++s)
                for (int k = 0; k < nsample; ++k)
                    out[(((size_t)bi * c + ci) * npoints + s) * nsample + k] =
                        points[((size_t)bi * c + ci) * n + idx[((size_t)bi * npoints + s) * nsample + k]];
    return 0;
}

/* K6: group_points_gpu.cu:8-25 */
ORC_API int orc_group_grad(const float *grad_out, const int *idx, float *grad_points, int b, int c, int n, int npoints, int nsample) {
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci)
            for (int s = 0; s < npoints; ++s)
                for (int k = 0; k < nsample; ++k)
                    grad_points[((size_t)bi * c + ci) * n + idx[((size_t)bi * npoints + s) * nsample + k]] +=
                        grad_out[(((size_t)bi * c + ci) * npoints + s) * nsample + k];
    return 0;
}

/* K4: ball_query_gpu.cu:9-45.  idx must be pre-zeroed by the caller (pointnet2_utils.py:218). */
ORC_API int orc_ball_query(const float *new_xyz, const float *xyz, int *idx, int b, int n, int m, float radius, int nsample) {
    float radius2 = radius * radius;
    for (int bi = 0; bi < b; ++bi)
        for (int p = 0; p < m; ++p) {
            const float *q = new_xyz + ((size_t)bi * m + p) * 3;
            const float *r = xyz + (size_t)bi * n * 3;
            int *o = idx + ((size_t)bi * m + p) * nsample;
            int cnt = 0;
            for (int k = 0; k < n; ++k) {
                float d2 = sqdist3(q[0], q[1], q[2], r[k * 3 + 0], r[k * 3 + 1], r[k * 3 + 2]);
                if (d2 < radius2) {
                    if (cnt == 0)
                        for (int l = 0; l < nsample; ++l) o[l] = k;
                    o[cnt] = k;
                    ++cnt;
                    if (cnt >= nsample) break;
                }
            }
        }
    return 0;
}

/* K7: interpolate_gpu.cu:9-52.  double accumulators holding float d; strict '<' cascade. */
ORC_API int orc_three_nn(const float *unknown, const float *known, float *dist2, int *idx, int b, int n, int m) {
    for (int bi = 0; bi < b; ++bi)
        for (int p = 0; p < n; ++p) {
            const float *u = unknown + ((size_t)bi * n + p) * 3;
            const float *kn = known + (size_t)bi * m * 3;
            double best1 = 1e40, best2 = 1e40, best3 = 1e40;
            int besti1 = 0, besti2 = 0, besti3 = 0;
            for (int k = 0; k < m; ++k) {
                float d = sqdist3(u[0], u[1], u[2], kn[k * 3 + 0], kn[k * 3 + 1], kn[k * 3 + 2]);
                if (d < best1) {
                    best3 = best2; besti3 = besti2;
                    best2 = best1; besti2 = besti1;
                    best1 = d; besti1 = k;
                } else if (d < best2) {
                    best3 = best2; besti3 = besti2;
                    best2 = d; besti2 = k;
                } else if (d < best3) {
                    best3 = d; besti3 = k;
                }
            }
            float *od = dist2 + ((size_t)bi * n + p) * 3;
            int *oi = idx + ((size_t)bi * n + p) * 3;
            od[0] = (float)best1; od[1] = (float)best2; od[2] = (float)best3;
            oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
        }
    return 0;
}

/* K8: interpolate_gpu.cu:77-97   out = w0*p0 + w1*p1 + w2*p2 (canon: fma(w2,p2, fma(w1,p1, w0*p0))) */
ORC_API int orc_three_interpolate(const float *points, const int *idx, const float *weight, float *out, int b, int c, int m, int n) {
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            const float *pt = points + ((size_t)bi * c + ci) * m;
            for (int p = 0; p < n; ++p) {
                const float *w = weight + ((size_t)bi * n + p) * 3;
                const int *id = idx + ((size_t)bi * n + p) * 3;
                out[((size_t)bi * c + ci) * n + p] = fmaf(w[2], pt[id[2]], fmaf(w[1], pt[id[1]], w[0] * pt[id[0]]));
            }
        }
    return 0;
}

/* K9: interpolate_gpu.cu:120-142 */
ORC_API int orc_three_interpolate_grad(const float *grad_out, const int *idx, const float *weight, float *grad_points, int b, int c, int n, int m) {
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci) {
            float *gp = grad_points + ((size_t)bi * c + ci) * m;
            for (int p = 0; p < n; ++p) {
                const float *w = weight + ((size_t)bi * n + p) * 3;
                const int *id = idx + ((size_t)bi * n + p) * 3;
                float g = grad_out[((size_t)bi * c + ci) * n + p];
                gp[id[0]] += g * w[0];
                gp[id[1]] += g * w[1];
                gp[id[2]] += g * w[2];
            }
        }
    return 0;
}

/* ------------------------------------------------------------------------
 * F1: brute-force KNN.
 * mode 0 ("expansion"): models/m_models/mocopci.py:1130-1169 (= models/pointconv_util.py:67-140)
 *     dist = -2*q.r ; dist += |q|^2 ; dist += |r|^2 ; topk(largest=False, sorted=False)
 *     canon: dot = fma(qz,rz, fma(qy,ry, qx*rx)); d = (fma(-2,dot,|q|^2)) + |r|^2,
 *     |p|^2 = (x*x + y*y) + z*z with each square rounded (torch.sum(src ** 2, -1)).
 * mode 1 ("direct"): pytorch3d.ops.knn_points as called at models/pointconv_util.py:910
 *     (pytorch3d 0.7.5, absent from the tree: squared L2 by direct differences, sorted ascending).
 * torch.topk(sorted=False) leaves order and tie choice unspecified; this oracle
 * DEFINES them: the K smallest under the lexicographic order (d, index), written
 * ascending.  Consumers in the reference are permutation-invariant over K.
 * ------------------------------------------------------------------------ */
#define ORC_MAX_K 128
typedef struct { float d; int i; } orc_cand;

static inline int cand_less(float d1, int i1, float d2, int i2) { return d1 < d2 || (d1 == d2 && i1 < i2); }

static void topk_insert(orc_cand *best, int k, int *cnt, float d, int i) {
    /* best[0..cnt) ascending under cand_less */
    if (*cnt == k && !cand_less(d, i, best[k - 1].d, best[k - 1].i)) return;
    int pos = (*cnt < k) ? (*cnt)++ : k - 1;
    while (pos > 0 && cand_less(d, i, best[pos - 1].d, best[pos - 1].i)) {
        best[pos] = best[pos - 1];
        --pos;
    }
    best[pos].d = d;
    best[pos].i = i;
}

static inline float sqnorm3(const float *p) { return (p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]; }

ORC_API float orc_pair_dist(const float *q, const float *r, int mode) {
    if (mode == 0) {
        float dot = fmaf(q[2], r[2], fmaf(q[1], r[1], q[0] * r[0]));
        return fmaf(-2.f, dot, sqnorm3(q)) + sqnorm3(r);
    }
    return sqdist3(q[0], q[1], q[2], r[0], r[1], r[2]);
}

ORC_API int orc_knn(const float *query, const float *ref, int *idx, float *dist, int b, int q, int n, int k, int mode) {
    if (k < 1 || k > ORC_MAX_K) return 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi)
        for (int p = 0; p < q; ++p) {
            orc_cand best[ORC_MAX_K];
            const float *qp = query + ((size_t)bi * q + p) * 3;
            const float *r = ref + (size_t)bi * n * 3;
            int cnt = 0;
            for (int j = 0; j < n; ++j) topk_insert(best, k, &cnt, orc_pair_dist(qp, r + (size_t)j * 3, mode), j);
            for (int j = 0; j < k; ++j) {
                /* fewer refs than k: repeat the last valid entry (product kernels do the same; the
                   reference would raise in torch.topk) */
                int src = j < cnt ? j : cnt - 1;
                idx[((size_t)bi * q + p) * k + j] = cnt ? best[src].i : 0;
                if (dist) dist[((size_t)bi * q + p) * k + j] = cnt ? best[src].d : 0.f;
            }
        }
    return 0;
}

/* F2: feature-space cosine KNN, models/pointconv_util.py:111-153.
 *   src/sqrt(sum(src^2)+1e-8), 1 - bmm, topk smallest.  Canon: squared norm is a
 *   sequential sum of rounded squares; the dot product is a sequential-k fmaf chain
 *   starting from 0 (the order v_mfma_f32_32x32x2_f32 produces); d = 1 - dot.
 *   feats are channel-last [B,N,C]. nq/nr are the normalised copies (scratch, may be NULL). */
ORC_API int orc_normalize_rows(const float *x, float *y, int rows, int c) {
    for (int r = 0; r < rows; ++r) {
        float s = 0.f;
        for (int j = 0; j < c; ++j) s = s + x[(size_t)r * c + j] * x[(size_t)r * c + j];
        float den = sqrtf(s + 1e-8f);
        for (int j = 0; j < c; ++j) y[(size_t)r * c + j] = x[(size_t)r * c + j] / den;
    }
    return 0;
}

ORC_API int orc_knn_cosine(const float *qfeat, const float *rfeat, int *idx, float *dist, int b, int q, int n, int c, int k) {
    float *nq = (float *)malloc(sizeof(float) * (size_t)b * q * c);
    float *nr = (float *)malloc(sizeof(float) * (size_t)b * n * c);
    orc_normalize_rows(qfeat, nq, b * q, c);
    orc_normalize_rows(rfeat, nr, b * n, c);
    if (k < 1 || k > ORC_MAX_K) return 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi)
        for (int p = 0; p < q; ++p) {
            orc_cand best[ORC_MAX_K];
            const float *a = nq + ((size_t)bi * q + p) * c;
            int cnt = 0;
            for (int j = 0; j < n; ++j) {
                const float *r = nr + ((size_t)bi * n + j) * c;
                float acc = 0.f;
                for (int t = 0; t < c; ++t) acc = fmaf(a[t], r[t], acc);
                topk_insert(best, k, &cnt, 1.0f - acc, j);
            }
            for (int j = 0; j < k; ++j) {
                int src = j < cnt ? j : cnt - 1;
                idx[((size_t)bi * q + p) * k + j] = cnt ? best[src].i : 0;
                if (dist) dist[((size_t)bi * q + p) * k + j] = cnt ? best[src].d : 0.f;
            }
        }
    free(nq);
    free(nr);
    return 0;
}

/* Row gather on channel-last tensors: out[b,s,k,:] = points[b, idx[b,s,k], :]
 * (= index_points_group, mocopci.py:1204-1215, without its two permute copies). */
ORC_API int orc_group_rows(const float *points, const int *idx, float *out, int b, int n, int c, int total_per_batch) {
    for (int bi = 0; bi < b; ++bi)
        for (int s = 0; s < total_per_batch; ++s)
            memcpy(out + ((size_t)bi * total_per_batch + s) * c, points + ((size_t)bi * n + idx[(size_t)bi * total_per_batch + s]) * c,
                   sizeof(float) * c);
    return 0;
}

/* F5: UpsampleFlow, mocopci.py:1485-1502 (PointWarping :1472-1479 uses the same weights).
 * knn_point(3, sparse, dense) in expansion mode; weights from the DISTANCE (not squared) of the
 * gathered differences: dist = clamp(||sparse[idx]-dense||, 1e-10); w = (1/dist)/sum(1/dist);
 * out[b,n,:] = sum_j w_j * feat[b, idx_j, :].  Channel-last feats [B,S,C] -> [B,N,C].
 * Canon: norm = sqrt((dx*dx + dy*dy) + dz*dz) (torch.norm over 3 elements: rounded squares,
 * sequential sum); inv_j = 1/dist_j; nrm = (inv0+inv1)+inv2; w_j = inv_j/nrm;
 * out = (w0*f0 + w1*f1) + w2*f2 with rounded products (torch.sum over dim=2 of the product). */
ORC_API int orc_interp3_weights(const float *dense, const float *sparse, const int *idx3, float *w3, int b, int n, int s) {
    for (int bi = 0; bi < b; ++bi)
        for (int p = 0; p < n; ++p) {
            const float *x = dense + ((size_t)bi * n + p) * 3;
            float inv[3];
            for (int j = 0; j < 3; ++j) {
                const float *y = sparse + ((size_t)bi * s + idx3[((size_t)bi * n + p) * 3 + j]) * 3;
                float dx = y[0] - x[0], dy = y[1] - x[1], dz = y[2] - x[2];
                float nr = sqrtf((dx * dx + dy * dy) + dz * dz);
                if (nr < 1e-10f) nr = 1e-10f;
                inv[j] = 1.0f / nr;
            }
            float nrm = (inv[0] + inv[1]) + inv[2];
            for (int j = 0; j < 3; ++j) w3[((size_t)bi * n + p) * 3 + j] = inv[j] / nrm;
        }
    return 0;
}

ORC_API int orc_interp3_apply(const float *feat, const int *idx3, const float *w3, float *out, int b, int n, int s, int c) {
    for (int bi = 0; bi < b; ++bi)
        for (int p = 0; p < n; ++p) {
            const int *id = idx3 + ((size_t)bi * n + p) * 3;
            const float *w = w3 + ((size_t)bi * n + p) * 3;
            const float *f0 = feat + ((size_t)bi * s + id[0]) * c;
            const float *f1 = feat + ((size_t)bi * s + id[1]) * c;
            const float *f2 = feat + ((size_t)bi * s + id[2]) * c;
            float *o = out + ((size_t)bi * n + p) * c;
            for (int t = 0; t < c; ++t) o[t] = (w[0] * f0[t] + w[1] * f1[t]) + w[2] * f2[t];
        }
    return 0;
}

/* Chamfer distance, models/utils.py:36-45 -> pytorch3d.loss.chamfer_distance defaults
 * (pytorch3d 0.7.5, absent): squared L2 by direct differences, mean over points, sum of
 * both directions, mean over batch.  Accumulated in double here; the product kernel sums
 * in fp32 and is compared with a stated tolerance. */
ORC_API double orc_chamfer(const float *x, const float *y, int b, int n, int m) {
    double total = 0.0;
    for (int bi = 0; bi < b; ++bi) {
        const float *xp = x + (size_t)bi * n * 3, *yp = y + (size_t)bi * m * 3;
        double sx = 0.0, sy = 0.0;
        for (int i = 0; i < n; ++i) {
            float best = INFINITY;
            for (int j = 0; j < m; ++j) {
                float d = sqdist3(xp[i * 3], xp[i * 3 + 1], xp[i * 3 + 2], yp[j * 3], yp[j * 3 + 1], yp[j * 3 + 2]);
                if (d < best) best = d;
            }
            sx += best;
        }
        for (int j = 0; j < m; ++j) {
            float best = INFINITY;
            for (int i = 0; i < n; ++i) {
                float d = sqdist3(yp[j * 3], yp[j * 3 + 1], yp[j * 3 + 2], xp[i * 3], xp[i * 3 + 1], xp[i * 3 + 2]);
                if (d < best) best = d;
            }
            sy += best;
        }
        total += sx / n + sy / m;
    }
    return total / b;
}

/* ------------------------------------------------------------------------
 * EMD metric: approxmatch + matchcost,
 * models/EMD/cuda/emd_kernel.cu:29-162 and :204-247 (Fan et al. soft auction, 10 levels
 * j = 7..-2, level = -4^j, last level 0).  Sequential restatement of one block's work:
 * same pass order, same per-thread summation order (l ascending / k ascending), expf in
 * place of the device's __expf.  match is (B, m, n) as in the reference; temp is internal.
 * Returns cost (B) = sum_{l,k} match[l][k] * |xyz2[l]-xyz1[k]|^2.
 * ------------------------------------------------------------------------ */
static inline float emd_d2(const float *a, const float *b) {
    /* (x2-x1)*(x2-x1)+(y2-y1)*(y2-y1)+(z2-z1)*(z2-z1): canon fma(dz,dz, fma(dy,dy, dx*dx)) */
    return sqdist3(b[0], b[1], b[2], a[0], a[1], a[2]);
}

ORC_API int orc_emd(const float *xyz1, const float *xyz2, float *match, float *cost, int b, int n, int m) {
    float *remainL = (float *)malloc(sizeof(float) * (size_t)(n + m) * 2);
    float *remainR = remainL + n, *ratioL = remainL + n + m, *ratioR = remainL + n + m + n;
    float multiL, multiR;
    if (n >= m) { multiL = 1; multiR = (float)(n / m); } else { multiL = (float)(m / n); multiR = 1; }
    for (int i = 0; i < b; ++i) {
        const float *p1 = xyz1 + (size_t)i * n * 3, *p2 = xyz2 + (size_t)i * m * 3;
        float *mt = match + (size_t)i * n * m;
        for (size_t j = 0; j < (size_t)n * m; ++j) mt[j] = 0;
        for (int j = 0; j < n; ++j) remainL[j] = multiL;
        for (int j = 0; j < m; ++j) remainR[j] = multiR;
        for (int j = 7; j >= -2; --j) {
            float level = -powf(4.0f, (float)j);
            if (j == -2) level = 0;
            for (int k = 0; k < n; ++k) {
                float suml = 1e-9f;
                for (int l = 0; l < m; ++l) suml += expf(level * emd_d2(p1 + k * 3, p2 + l * 3)) * remainR[l];
                ratioL[k] = remainL[k] / suml;
            }
            for (int l = 0; l < m; ++l) {
                float sumr = 0;
                for (int k = 0; k < n; ++k) sumr += expf(level * emd_d2(p1 + k * 3, p2 + l * 3)) * ratioL[k];
                sumr *= remainR[l];
                float consumption = fminf(remainR[l] / (sumr + 1e-9f), 1.0f);
                ratioR[l] = consumption * remainR[l];
                remainR[l] = fmaxf(0.0f, remainR[l] - sumr);
            }
            for (int k = 0; k < n; ++k) {
                float suml = 0, rl = ratioL[k];
                for (int l = 0; l < m; ++l) {
                    float w = expf(level * emd_d2(p1 + k * 3, p2 + l * 3)) * rl * ratioR[l];
                    mt[(size_t)l * n + k] += w;
                    suml += w;
                }
                remainL[k] = fmaxf(0.0f, remainL[k] - suml);
            }
        }
        /* matchcost, emd_kernel.cu:204-247 (double accumulation here; the device sums in fp32) */
        double s = 0.0;
        for (int k = 0; k < n; ++k)
            for (int l = 0; l < m; ++l) s += (double)emd_d2(p1 + k * 3, p2 + l * 3) * (double)mt[(size_t)l * n + k];
        cost[i] = (float)s;
    }
    free(remainL);
    return 0;
}
