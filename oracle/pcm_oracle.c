/*
 * pcm_oracle.c -- CPU ORACLE (test infrastructure, never shipped in the product path).
 * See pcm_oracle.h for scope and the parity-pinning statement.
 *
 * Arithmetic contract: every fp32 expression is evaluated un-contracted,
 * left-to-right exactly as written in the reference .cu source.  Build with
 * -ffp-contract=off (oracle/Makefile does).  OpenMP is used only ACROSS independent
 * units (clouds / queries / output elements), never inside one, so results are identical
 * with any thread count.
 *
 * All file:line citations are relative to /root/reference/libs/pointops/src/.
 */
#include "pcm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* One accumulation step of the squared distance.  Default (the arithmetic contract of SURVEY F9, and what every parity test uses):
 * the product and the sum rounded separately.  -DPCM_ORACLE_FMAD builds the SENSITIVITY variant only (libpcm_oracle_fmad.so,
 * tools/fmad_sensitivity.py): the fused form a CUDA compiler's default -fmad=true makes of the same source expression. */
#ifdef PCM_ORACLE_FMAD
#define PCM_ACC(d, a) fmaf((a), (a), (d))
#else
#define PCM_ACC(d, a) ((d) + (a) * (a))
#endif

#define PCM_TPB_MAX 1024
#define PCM_KNN_MAX 128        /* knn_query_cuda_kernel.cu:82-83: float best_dist[128]          */
#define PCM_BALL_CAND_MAX 2048 /* ball_query_cuda_kernel.cu:86-87: float candi_dist[2048]      */

/* cuda_utils.h:11-14.  Same double-precision log ratio, same truncation, same clamps. */
int pcm_opt_n_threads_cpu(int work_size)
{
    const int pow_2 = (int)(log((double)work_size) / log(2.0));
    int v = 1 << pow_2;
    if (v > PCM_TPB_MAX) v = PCM_TPB_MAX;
    if (v < 1) v = 1;
    return v;
}

/* ------------------------------------------------------------------------------------ */
/* K1  farthest point sampling  (sampling/sampling_cuda_kernel.cu:15-129)                */
/* ------------------------------------------------------------------------------------ */
/*
 * One CUDA block per cloud, BS = opt_n_threads(n_max) threads.  Thread t owns points
 * start_n+t, +BS, ... and keeps (best, besti) with strict '>' (:57-58); the shared-memory
 * tree (:64-123) halves the stride from BS/2 to 1 and __update (:5-10) keeps the LOWER slot
 * on ties (value by max, index by strict v2 > v1).  We walk the points once in ascending k
 * (each thread still sees its own points in ascending order, so per-thread results are
 * identical) and then run the tree literally.
 */
int pcm_farthest_point_sampling_cpu(int b, int n_max, const float *xyz, const int *offset,
                                    const int *new_offset, float *tmp, int *idx)
{
    if (b <= 0) return 0;
    if (n_max < 1) return 1;
    const int BS = pcm_opt_n_threads_cpu(n_max);
    int bid;
#pragma omp parallel for schedule(dynamic, 1)
    for (bid = 0; bid < b; bid++) {
        float dists[PCM_TPB_MAX];
        int dists_i[PCM_TPB_MAX];
        const int start_n = bid == 0 ? 0 : offset[bid - 1];
        const int end_n = offset[bid];
        const int start_m = bid == 0 ? 0 : new_offset[bid - 1];
        const int end_m = new_offset[bid];
        int old = start_n;
        int j, k, t, s;
        /* :39 writes idx[start_m] unconditionally; for an empty output range that store
         * lands in the next cloud's slot (a race in the reference) -- skipped here. */
        if (end_m <= start_m) continue;
        idx[start_m] = start_n;
        for (j = start_m + 1; j < end_m; j++) {
            const float x1 = xyz[old * 3 + 0];
            const float y1 = xyz[old * 3 + 1];
            const float z1 = xyz[old * 3 + 2];
            for (t = 0; t < BS; t++) { /* :43-44 */
                dists[t] = -1.0f;
                dists_i[t] = start_n;
            }
            t = 0;
            for (k = start_n; k < end_n; k++) {
                const float x2 = xyz[k * 3 + 0];
                const float y2 = xyz[k * 3 + 1];
                const float z2 = xyz[k * 3 + 2];
                const float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
                float d = dx * dx;
                d = PCM_ACC(d, dy);
                d = PCM_ACC(d, dz); /* :54, left-to-right, no FMA */
                const float tk = tmp[k];
                const float d2 = d < tk ? d : tk; /* min(d, tmp[k]) :55 */
                tmp[k] = d2;
                if (d2 > dists[t]) { /* :57-58 strict > */
                    dists[t] = d2;
                    dists_i[t] = k;
                }
                if (++t == BS) t = 0;
            }
            for (s = BS >> 1; s >= 1; s >>= 1) { /* :64-123 */
                for (t = 0; t < s; t++) {
                    const float v1 = dists[t], v2 = dists[t + s];
                    const int i1 = dists_i[t], i2 = dists_i[t + s];
                    dists[t] = v1 > v2 ? v1 : v2; /* max(v1, v2) :8 */
                    dists_i[t] = v2 > v1 ? i2 : i1; /* :9 */
                }
            }
            old = dists_i[0]; /* :125 */
            idx[j] = old;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* heap helpers shared by K2/K3 (knn_query_cuda_kernel.cu:15-42, ball_query...:15-42)    */
/* ------------------------------------------------------------------------------------ */
static void pcm_reheap(float *dist, int *idx, int k)
{
    int root = 0;
    int child = root * 2 + 1;
    while (child < k) {
        if (child + 1 < k && dist[child + 1] > dist[child]) child++;
        if (dist[root] > dist[child]) return; /* equal keys DO swap */
        {
            const float td = dist[root];
            const int ti = idx[root];
            dist[root] = dist[child];
            idx[root] = idx[child];
            dist[child] = td;
            idx[child] = ti;
        }
        root = child;
        child = root * 2 + 1;
    }
}

static void pcm_heap_sort(float *dist, int *idx, int k)
{
    int i;
    for (i = k - 1; i > 0; i--) {
        const float td = dist[0];
        const int ti = idx[0];
        dist[0] = dist[i];
        idx[0] = idx[i];
        dist[i] = td;
        idx[i] = ti;
        pcm_reheap(dist, idx, i);
    }
}

/* get_bt_idx (knn_query_cuda_kernel.cu:45-56): first i with q < new_offset[i]. */
static int pcm_get_bt_idx(int q, const int *off)
{
    int i = 0;
    while (!(q < off[i])) i++;
    return i;
}

/* ------------------------------------------------------------------------------------ */
/* K2  kNN query  (knn_query/knn_query_cuda_kernel.cu:60-104)                            */
/* ------------------------------------------------------------------------------------ */
int pcm_knn_query_cpu(int m, int nsample, const float *xyz, const float *new_xyz,
                      const int *offset, const int *new_offset, int *idx, float *dist2)
{
    if (nsample < 1 || nsample > PCM_KNN_MAX) return 1;
    int q;
#pragma omp parallel for schedule(static, 64)
    for (q = 0; q < m; q++) {
        float best_dist[PCM_KNN_MAX];
        int best_idx[PCM_KNN_MAX];
        const int bt = pcm_get_bt_idx(q, new_offset);
        const int start = bt == 0 ? 0 : offset[bt - 1];
        const int end = offset[bt];
        const float new_x = new_xyz[q * 3 + 0];
        const float new_y = new_xyz[q * 3 + 1];
        const float new_z = new_xyz[q * 3 + 2];
        int i;
        for (i = 0; i < nsample; i++) {
            best_dist[i] = 1e10f;
            best_idx[i] = -1;
        }
        for (i = start; i < end; i++) {
            const float dx = new_x - xyz[i * 3 + 0];
            const float dy = new_y - xyz[i * 3 + 1];
            const float dz = new_z - xyz[i * 3 + 2];
            float d2 = dx * dx;
            d2 = PCM_ACC(d2, dy);
            d2 = PCM_ACC(d2, dz); /* :91 */
            if (d2 < best_dist[0]) { /* :92 strict < */
                best_dist[0] = d2;
                best_idx[0] = i;
                pcm_reheap(best_dist, best_idx, nsample);
            }
        }
        pcm_heap_sort(best_dist, best_idx, nsample); /* :99 */
        for (i = 0; i < nsample; i++) {
            idx[q * nsample + i] = best_idx[i];
            dist2[q * nsample + i] = best_dist[i];
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* K3  ball query  (ball_query/ball_query_cuda_kernel.cu:58-123)                         */
/* ------------------------------------------------------------------------------------ */
/*
 * Quirks kept on purpose: (a) `d2 <= 1e-5` compares in double (:91); (b) heap_sort runs on
 * an array that was never heapified (:103) so the result is a deterministic permutation,
 * generally NOT sorted; (c) in the subsample branch dist2 receives the candidate INDEX
 * (:120).  More than 2048 candidates overflow the reference's stack arrays (UB): the oracle
 * returns 2 instead.
 */
int pcm_ball_query_cpu(int m, int nsample, float min_radius, float max_radius,
                       const float *xyz, const float *new_xyz, const int *offset,
                       const int *new_offset, int *idx, float *dist2)
{
    if (nsample < 1) return 1;
    const float max_radius2 = max_radius * max_radius;
    const float min_radius2 = min_radius * min_radius;
    int overflow = 0;
    int q;
#pragma omp parallel for schedule(static, 16)
    for (q = 0; q < m; q++) {
        float candi_dist[PCM_BALL_CAND_MAX];
        int candi_idx[PCM_BALL_CAND_MAX];
        int candi_num = 0;
        const int bt = pcm_get_bt_idx(q, new_offset);
        const int start = bt == 0 ? 0 : offset[bt - 1];
        const int end = offset[bt];
        const float new_x = new_xyz[q * 3 + 0];
        const float new_y = new_xyz[q * 3 + 1];
        const float new_z = new_xyz[q * 3 + 2];
        int i, bad = 0;
        for (i = start; i < end; i++) {
            const float dx = new_x - xyz[i * 3 + 0];
            const float dy = new_y - xyz[i * 3 + 1];
            const float dz = new_z - xyz[i * 3 + 2];
            float d2 = dx * dx;
            d2 = PCM_ACC(d2, dy);
            d2 = PCM_ACC(d2, dz);
            if ((double)d2 <= 1e-5 || (d2 >= min_radius2 && d2 < max_radius2)) {
                if (candi_num >= PCM_BALL_CAND_MAX) {
                    bad = 1;
                    break;
                }
                candi_dist[candi_num] = d2;
                candi_idx[candi_num] = i;
                candi_num += 1;
            }
        }
        if (bad) {
#pragma omp atomic write
            overflow = 1;
            for (i = 0; i < nsample; i++) {
                idx[q * nsample + i] = -1;
                dist2[q * nsample + i] = 1e10f;
            }
            continue;
        }
        pcm_heap_sort(candi_dist, candi_idx, candi_num); /* :103, no heapify */
        if (candi_num <= nsample) {
            for (i = 0; i < candi_num; i++) {
                idx[q * nsample + i] = candi_idx[i];
                dist2[q * nsample + i] = candi_dist[i];
            }
            for (i = candi_num; i < nsample; i++) {
                idx[q * nsample + i] = -1;
                dist2[q * nsample + i] = 1e10f;
            }
        } else {
            const float sep = (float)candi_num / nsample; /* :115 */
            for (i = 0; i < nsample; i++) {
                const int index = (int)(sep * i); /* :118 float*int -> float -> trunc */
                idx[q * nsample + i] = candi_idx[index];
                dist2[q * nsample + i] = (float)candi_idx[index]; /* :120 (sic) */
            }
        }
    }
    return overflow ? 2 : 0;
}

/* ------------------------------------------------------------------------------------ */
/* K4  random ball query  (random_ball_query/random_ball_query_cuda_kernel.cu:58-108)    */
/* ------------------------------------------------------------------------------------ */
int pcm_random_ball_query_cpu(int m, int nsample, float min_radius, float max_radius,
                              const int *order, const float *xyz, const float *new_xyz,
                              const int *offset, const int *new_offset, int *idx, float *dist2)
{
    if (nsample < 1) return 1;
    const float max_radius2 = max_radius * max_radius;
    const float min_radius2 = min_radius * min_radius;
    int q;
#pragma omp parallel for schedule(static, 64)
    for (q = 0; q < m; q++) {
        const int bt = pcm_get_bt_idx(q, new_offset);
        const int start = bt == 0 ? 0 : offset[bt - 1];
        const int end = offset[bt];
        const float new_x = new_xyz[q * 3 + 0];
        const float new_y = new_xyz[q * 3 + 1];
        const float new_z = new_xyz[q * 3 + 2];
        int cnt = 0, i;
        for (i = start; i < end; i++) {
            const int o = order[i];
            const float dx = new_x - xyz[o * 3 + 0];
            const float dy = new_y - xyz[o * 3 + 1];
            const float dz = new_z - xyz[o * 3 + 2];
            float d2 = dx * dx;
            d2 = PCM_ACC(d2, dy);
            d2 = PCM_ACC(d2, dz);
            if ((double)d2 <= 1e-5 || (d2 >= min_radius2 && d2 < max_radius2)) {
                dist2[q * nsample + cnt] = d2;
                idx[q * nsample + cnt] = o;
                cnt += 1;
                if (cnt >= nsample) break;
            }
        }
        for (i = cnt; i < nsample; i++) {
            idx[q * nsample + i] = -1;
            dist2[q * nsample + i] = 1e10f;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* K5  grouping  (grouping/grouping_cuda_kernel.cu:5-25)                                 */
/* ------------------------------------------------------------------------------------ */
int pcm_grouping_forward_cpu(int m, int nsample, int c, const float *input, const int *idx,
                             float *output)
{
    long r;
    const long rows = (long)m * nsample;
#pragma omp parallel for schedule(static)
    for (r = 0; r < rows; r++) {
        const float *src = input + (long)idx[r] * c; /* no -1 handling in the reference */
        memcpy(output + r * c, src, (size_t)c * sizeof(float));
    }
    return 0;
}

int pcm_grouping_backward_cpu(int m, int nsample, int c, const float *grad_output,
                              const int *idx, float *grad_input)
{
    /* atomicAdd in the reference (:24): order-free on the GPU; sequential in thread order here. */
    long r;
    int ch;
    const long rows = (long)m * nsample;
    for (r = 0; r < rows; r++) {
        float *dst = grad_input + (long)idx[r] * c;
        const float *g = grad_output + r * c;
        for (ch = 0; ch < c; ch++) dst[ch] = dst[ch] + g[ch];
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* K6  interpolation  (interpolation/interpolation_cuda_kernel.cu:5-33)                  */
/* ------------------------------------------------------------------------------------ */
int pcm_interpolation_forward_cpu(int n, int c, int k, const float *input, const int *idx,
                                  const float *weight, float *output)
{
    long e;
    const long total = (long)n * c;
#pragma omp parallel for schedule(static)
    for (e = 0; e < total; e++) {
        const int c_idx = (int)(e % c);
        const long n_idx = e / c;
        int i;
        for (i = 0; i < k; i++) { /* :12-17: output += input * weight, k ascending */
            const long ii = n_idx * k + i;
            const float p = input[(long)idx[ii] * c + c_idx] * weight[ii];
            output[e] = output[e] + p;
        }
    }
    return 0;
}

int pcm_interpolation_backward_cpu(int n, int c, int k, const float *grad_output, const int *idx,
                                   const float *weight, float *grad_input)
{
    long e;
    const long total = (long)n * c;
    for (e = 0; e < total; e++) {
        const int c_idx = (int)(e % c);
        const long n_idx = e / c;
        int i;
        for (i = 0; i < k; i++) {
            const long ii = n_idx * k + i;
            const long dst = (long)idx[ii] * c + c_idx;
            const float p = grad_output[e] * weight[ii];
            grad_input[dst] = grad_input[dst] + p;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* K7  subtraction  (subtraction/subtraction_cuda_kernel.cu:5-30)                        */
/* ------------------------------------------------------------------------------------ */
int pcm_subtraction_forward_cpu(int n, int nsample, int c, const float *input1,
                                const float *input2, const int *idx, float *output)
{
    long r;
    const long rows = (long)n * nsample;
#pragma omp parallel for schedule(static)
    for (r = 0; r < rows; r++) {
        const long n_idx = r / nsample;
        const float *a = input1 + n_idx * c;
        const float *bsrc = input2 + (long)idx[r] * c;
        float *o = output + r * c;
        int ch;
        for (ch = 0; ch < c; ch++) o[ch] = a[ch] - bsrc[ch];
    }
    return 0;
}

int pcm_subtraction_backward_cpu(int n, int nsample, int c, const int *idx,
                                 const float *grad_output, float *grad_input1, float *grad_input2)
{
    long r;
    const long rows = (long)n * nsample;
    for (r = 0; r < rows; r++) {
        const long n_idx = r / nsample;
        float *g1 = grad_input1 + n_idx * c;
        float *g2 = grad_input2 + (long)idx[r] * c;
        const float *g = grad_output + r * c;
        int ch;
        for (ch = 0; ch < c; ch++) {
            g1[ch] = g1[ch] + g[ch];
            g2[ch] = g2[ch] + (-g[ch]);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* K8  aggregation  (aggregation/aggregation_cuda_kernel.cu:5-39)                        */
/* ------------------------------------------------------------------------------------ */
int pcm_aggregation_forward_cpu(int n, int nsample, int c, int w_c, const float *input,
                                const float *position, const float *weight, const int *idx,
                                float *output)
{
    long e;
    const long total = (long)n * c;
#pragma omp parallel for schedule(static)
    for (e = 0; e < total; e++) {
        const int c_idx = (int)(e % c);
        const long n_idx = e / c;
        const int w_c_idx = c_idx % w_c;
        int s;
        for (s = 0; s < nsample; s++) {
            const long ii = n_idx * nsample + s;
            const float in = input[(long)idx[ii] * c + c_idx];
            const float pos = position[n_idx * nsample * c + (long)s * c + c_idx];
            const float w = weight[n_idx * nsample * w_c + (long)s * w_c + w_c_idx];
            const float sum = in + pos;
            const float p = sum * w;
            output[e] = output[e] + p;
        }
    }
    return 0;
}

int pcm_aggregation_backward_cpu(int n, int nsample, int c, int w_c, const float *input,
                                 const float *position, const float *weight, const int *idx,
                                 const float *grad_output, float *grad_input, float *grad_position,
                                 float *grad_weight)
{
    long e;
    const long total = (long)n * c;
    for (e = 0; e < total; e++) {
        const int c_idx = (int)(e % c);
        const long n_idx = e / c;
        const int w_c_idx = c_idx % w_c;
        int s;
        for (s = 0; s < nsample; s++) {
            const long ii = n_idx * nsample + s;
            const long in_i = (long)idx[ii] * c + c_idx;
            const long pos_i = n_idx * nsample * c + (long)s * c + c_idx;
            const long w_i = n_idx * nsample * w_c + (long)s * w_c + w_c_idx;
            const float gw = grad_output[e] * weight[w_i];
            grad_input[in_i] = grad_input[in_i] + gw;
            grad_position[pos_i] = gw;
            {
                const float sum = input[in_i] + position[pos_i];
                const float p = grad_output[e] * sum;
                grad_weight[w_i] = grad_weight[w_i] + p;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* K9  attention relation / fusion steps  (attention/attention_cuda_kernel.cu:9-86)      */
/* ------------------------------------------------------------------------------------ */
int pcm_attention_relation_step_forward_cpu(int m, int g, int c, const float *query,
                                            const float *key, const float *weight,
                                            const int *index_target, const int *index_refer,
                                            float *output)
{
    long r;
#pragma omp parallel for schedule(static)
    for (r = 0; r < m; r++) {
        int gi, ci;
        for (gi = 0; gi < g; gi++) {
            /* atomicAdd over c in the reference; ascending c here */
            for (ci = 0; ci < c; ci++) {
                const long q_i = (long)index_target[r] * g * c + (long)gi * c + ci;
                const long k_i = (long)index_refer[r] * g * c + (long)gi * c + ci;
                float v = query[q_i] * key[k_i];
                v = v * weight[ci];
                output[r * g + gi] = output[r * g + gi] + v;
            }
        }
    }
    return 0;
}

int pcm_attention_relation_step_backward_cpu(int m, int g, int c, const float *query,
                                             float *grad_query, const float *key, float *grad_key,
                                             const float *weight, float *grad_weight,
                                             const int *index_target, const int *index_refer,
                                             const float *grad_output)
{
    long r;
    int gi, ci;
    for (r = 0; r < m; r++)
        for (gi = 0; gi < g; gi++)
            for (ci = 0; ci < c; ci++) {
                const long q_i = (long)index_target[r] * g * c + (long)gi * c + ci;
                const long k_i = (long)index_refer[r] * g * c + (long)gi * c + ci;
                const float grad_r = grad_output[r * g + gi];
                float a = grad_r * key[k_i];
                a = a * weight[ci];
                grad_query[q_i] = grad_query[q_i] + a;
                a = grad_r * query[q_i];
                a = a * weight[ci];
                grad_key[k_i] = grad_key[k_i] + a;
                a = grad_r * key[k_i];
                a = a * query[q_i];
                grad_weight[ci] = grad_weight[ci] + a;
            }
    return 0;
}

int pcm_attention_fusion_step_forward_cpu(int m, int g, int c, const float *weight,
                                          const float *value, const int *index_target,
                                          const int *index_refer, float *output)
{
    long r;
    int gi, ci;
    for (r = 0; r < m; r++)
        for (gi = 0; gi < g; gi++)
            for (ci = 0; ci < c; ci++) {
                const long o_i = (long)index_target[r] * g * c + (long)gi * c + ci;
                const long v_i = (long)index_refer[r] * g * c + (long)gi * c + ci;
                const float f = weight[r * g + gi] * value[v_i];
                output[o_i] = output[o_i] + f;
            }
    return 0;
}

int pcm_attention_fusion_step_backward_cpu(int m, int g, int c, const float *weight,
                                           float *grad_weight, const float *value,
                                           float *grad_value, const int *index_target,
                                           const int *index_refer, const float *grad_output)
{
    long r;
    int gi, ci;
    for (r = 0; r < m; r++)
        for (gi = 0; gi < g; gi++)
            for (ci = 0; ci < c; ci++) {
                const long o_i = (long)index_target[r] * g * c + (long)gi * c + ci;
                const long v_i = (long)index_refer[r] * g * c + (long)gi * c + ci;
                const long w_i = r * g + gi;
                const float grad = grad_output[o_i];
                const float a = grad * value[v_i];
                grad_weight[w_i] = grad_weight[w_i] + a;
                {
                    const float bq = grad * weight[w_i];
                    grad_value[v_i] = grad_value[v_i] + bq;
                }
            }
    return 0;
}

/* Thread count of the oracle's own OpenMP team (bench.py's cpu_baseline leg: the host path's framework ops and the oracle
 * must not both spin a full machine's worth of threads). */
int pcm_oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}
