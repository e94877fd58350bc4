/*
 * pcm_pointops.h -- C ABI of libpcm_pointops.so, the MI355X (gfx950) pointops library.
 *
 * This is the drop-in boundary for the reference's native layer
 *   /root/reference/libs/pointops/src/<op>/<op>_cuda_kernel.h   (extern "C" *_cuda_launcher)
 *   /root/reference/libs/pointops/src/pointops_api.cpp:15-32     (the 16 pybind entry points)
 * Each pcm_*_hip below takes EXACTLY the reference launcher's argument list (cited per function)
 * plus a trailing stream, and returns an int status instead of void:
 *     0                success (kernel(s) enqueued on `stream`; asynchronous like the reference)
 *     PCM_ERR_*        argument rejected before any launch (cases that are UB in the reference)
 *     >= 1000          1000 + hipError_t from the launch
 * Arguments are checked before the runtime is touched: negative sizes are rejected, and an empty call (no rows / no jobs)
 * returns PCM_OK without a launch or a status query (tests/test_capi.py sweeps every entry point below for both, on a host
 * without a device).
 * Plain pointers and sizes only: no torch types, no C++ types.  All pointers are DEVICE pointers
 * (HBM), row-major, fp32 data / int32 indices, packed "(n,3) + cumulative offset" layout
 * (cloud i = [offset[i-1], offset[i]), offset[-1] := 0).
 * `stream` is a hipStream_t passed as void* (NULL = the legacy default stream the reference uses).
 * The library keeps no global state: every call is independent and thread-safe.
 *
 * Arithmetic contract: distances are un-contracted IEEE fp32, (a-b)*(a-b) summed x,y,z left to
 * right (the library is built with -ffp-contract=off); FPS indices and kNN/ball-query neighbour
 * lists are bit-exact against oracle/pcm_oracle.c, including tie cases.
 */
#ifndef PCM_POINTOPS_H
#define PCM_POINTOPS_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCM_OK 0
#define PCM_ERR_BAD_ARG 1      /* negative sizes, nsample out of range, n_max < 1 ...           */
#define PCM_ERR_UNSUPPORTED 2  /* shape outside what the kernels cover (see each function)      */
#define PCM_ERR_HIP_BASE 1000  /* + hipError_t                                                  */

#define PCM_KNN_MAX_NSAMPLE 128   /* knn_query_cuda_kernel.cu:82-83  float best_dist[128]       */
#define PCM_BALL_MAX_CAND 2048    /* ball_query_cuda_kernel.cu:86-87 float candi_dist[2048]     */

/* Library / build identification (e.g. "pcm_pointops 0.1 gfx950 fp-contract=off"). */
const char *pcm_version(void);

/* cuda_utils.h:11-14 opt_n_threads(): the reference block size that defines FPS tie order. */
int pcm_opt_n_threads(int work_size);

/* ---- K1 farthest point sampling -------------------------------------------------------------
 * replaces farthest_point_sampling_cuda_launcher   sampling/sampling_cuda_kernel.h:9-17
 *          (kernel: sampling/sampling_cuda_kernel.cu:15-171; wrapper: functions/sampling.py:6-26)
 * b clouds; n = n_max = max_i N_i (the reference passes this as `n`); xyz (sum N_i, 3);
 * offset (b), new_offset (b); tmp (sum N_i) pre-filled with 1e10f by the caller (used only by the
 * large-cloud path, N_i > 16384); idx (new_offset[b-1]) receives GLOBAL point indices. */
int pcm_farthest_point_sampling_hip(int b, int n, const float *xyz, const int *offset,
                                    const int *new_offset, float *tmp, int *idx, void *stream);

/* ---- K2 kNN query ---------------------------------------------------------------------------
 * replaces knn_query_cuda_launcher   knn_query/knn_query_cuda_kernel.h
 *          (kernel: knn_query/knn_query_cuda_kernel.cu:60-112; wrapper: functions/query.py:6-23)
 * idx (m, nsample) ascending by dist2, -1 / 1e10f padding when the cloud has < nsample points;
 * dist2 (m, nsample) SQUARED distances (the Python wrapper takes the sqrt).  1 <= nsample <= 128. */
int pcm_knn_query_hip(int m, int nsample, const float *xyz, const float *new_xyz,
                      const int *offset, const int *new_offset, int *idx, float *dist2,
                      void *stream);

/* ---- K3 ball query --------------------------------------------------------------------------
 * replaces ball_query_cuda_launcher   ball_query/ball_query_cuda_kernel.h
 *          (kernel: ball_query/ball_query_cuda_kernel.cu:58-190; wrapper: functions/query.py:72-107)
 * Reproduces the reference's candidate order (heap_sort without heapify), the -1 / 1e10f padding
 * and the dist2 := index quirk of the subsample branch (:120).  A query with more than 2048
 * in-range candidates (stack overflow / UB in the reference) yields an all -1 / 1e10f row. */
int pcm_ball_query_hip(int m, int nsample, float min_radius, float max_radius, const float *xyz,
                       const float *new_xyz, const int *offset, const int *new_offset, int *idx,
                       float *dist2, void *stream);
/* Same, with the number of clouds b (offset / new_offset length): the owning cloud of a query is found by bisection
 * instead of the reference's linear scan of new_offset (:64-71 get_bt_idx), which costs up to b dependent loads. */
int pcm_ball_query_b_hip(int b, int m, int nsample, float min_radius, float max_radius, const float *xyz,
                         const float *new_xyz, const int *offset, const int *new_offset, int *idx,
                         float *dist2, void *stream);
/* same results through two kernels and a caller-provided workspace (candidate collection at full occupancy, then the heap replay;
 * csrc/ball.hip).  pcm_ball_query_ws_bytes(m) bytes of device memory, contents irrelevant; 0 = use pcm_ball_query_b_hip. */
size_t pcm_ball_query_ws_bytes(int m);
int pcm_ball_query_ws_hip(int b, int m, int nsample, float min_radius, float max_radius, const float *xyz,
                          const float *new_xyz, const int *offset, const int *new_offset, int *idx, float *dist2,
                          void *ws, size_t ws_bytes, void *stream);

/* ---- K4 random ball query -------------------------------------------------------------------
 * replaces random_ball_query_cuda_launcher   random_ball_query/random_ball_query_cuda_kernel.h
 *          (kernel: ..._kernel.cu:58-123; wrapper: functions/query.py:26-69)
 * order (sum N_i): per-cloud permutation of global indices (the wrapper's torch.randperm). */
int pcm_random_ball_query_hip(int m, int nsample, float min_radius, float max_radius,
                              const int *order, const float *xyz, const float *new_xyz,
                              const int *offset, const int *new_offset, int *idx, float *dist2,
                              void *stream);
int pcm_random_ball_query_b_hip(int b, int m, int nsample, float min_radius, float max_radius,
                                const int *order, const float *xyz, const float *new_xyz,
                                const int *offset, const int *new_offset, int *idx, float *dist2,
                                void *stream);

/* ---- K5 grouping ----------------------------------------------------------------------------
 * replaces grouping_{forward,backward}_cuda_launcher   grouping/grouping_cuda_kernel.h
 *          (kernels: grouping/grouping_cuda_kernel.cu:5-40; wrapper: functions/grouping.py:6-32)
 * forward: output(m,nsample,c) = input[idx];  backward: grad_input(n,c) += scatter(grad_output)
 * (grad_input pre-zeroed by the caller).  No -1 handling, like the reference. */
int pcm_grouping_forward_hip(int m, int nsample, int c, const float *input, const int *idx,
                             float *output, void *stream);
int pcm_grouping_backward_hip(int m, int nsample, int c, const float *grad_output, const int *idx,
                              float *grad_input, void *stream);

/* ---- K6 interpolation -----------------------------------------------------------------------
 * replaces interpolation_{forward,backward}_cuda_launcher   interpolation/interpolation_cuda_kernel.h
 *          (kernels: interpolation_cuda_kernel.cu:5-47; wrapper: functions/interpolation.py:25-59)
 * forward: output(n,c) = sum_k input[idx[n,k]] * weight[n,k] (k ascending; written, not accumulated -- the
 *          reference adds into the caller's zeroed buffer, same result).
 * backward: one atomic per element (the reference's form; grad_input pre-zeroed).  The atomic-free form is
 *          pcm_scatter_plan_hip(n*k, m, idx) + pcm_segment_sum_hip(rowdiv = k, scale = weight), below. */
int pcm_interpolation_forward_hip(int n, int c, int k, const float *input, const int *idx,
                                  const float *weight, float *output, void *stream);
int pcm_interpolation_backward_hip(int n, int c, int k, const float *grad_output, const int *idx,
                                   const float *weight, float *grad_input, void *stream);

/* ---- K7 subtraction -------------------------------------------------------------------------
 * replaces subtraction_{forward,backward}_cuda_launcher   subtraction/subtraction_cuda_kernel.h
 * backward: grad_input1 is WRITTEN (segmented sum over the nsample rows of each query, no atomics); grad_input2
 *          receives one atomic per element (pre-zeroed), or pass grad_input2 = NULL and scatter through a plan. */
int pcm_subtraction_forward_hip(int n, int nsample, int c, const float *input1,
                                const float *input2, const int *idx, float *output, void *stream);
int pcm_subtraction_backward_hip(int n, int nsample, int c, const int *idx,
                                 const float *grad_output, float *grad_input1, float *grad_input2,
                                 void *stream);

/* ---- K8 aggregation -------------------------------------------------------------------------
 * replaces aggregation_{forward,backward}_cuda_launcher   aggregation/aggregation_cuda_kernel.h
 * backward: grad_position written, grad_weight accumulated (pre-zeroed) by one thread per (row, w_c) in a fixed
 *          order; grad_input receives one atomic per element (pre-zeroed), or pass grad_input = NULL and scatter
 *          through a plan (pcm_segment_sum_hip with scale_mode 2, rowdiv = nsample). */
int pcm_aggregation_forward_hip(int n, int nsample, int c, int w_c, const float *input,
                                const float *position, const float *weight, const int *idx,
                                float *output, void *stream);
int pcm_aggregation_backward_hip(int n, int nsample, int c, int w_c, const float *input,
                                 const float *position, const float *weight, const int *idx,
                                 const float *grad_output, float *grad_input, float *grad_position,
                                 float *grad_weight, void *stream);

/* ---- segmented gather-sum: the atomic-free form of the scatter-adds above (csrc/segsum.hip) ----------------
 * No reference counterpart: the reference scatters with atomicAdd (grouping_cuda_kernel.cu:24,
 * interpolation_cuda_kernel.cu:35-40, subtraction_cuda_kernel.cu:36-38, aggregation_cuda_kernel.cu:42-46).
 * pcm_scatter_plan_hip inverts idx (rows entries with values in [0, n_dst), negatives skipped) into a CSR held in
 * `ws` (pcm_scatter_plan_ws_ints(rows, n_dst) ints): *start_out (n_dst + 1) and *list_out (the entries of each
 * destination row).  pcm_segment_sum_hip then writes
 *     dst[j, col] = sign * sum_{t in segment j} src[s(e_t) * src_stride + src_off + col] * scale(e_t, col)
 * with segment j = [start[j], start[j+1]) (start == NULL: [j*seglen, (j+1)*seglen)), e_t = list ? list[t] : t,
 * s(e) = map ? map[e] : e / rowdiv (negative: skipped), scale_mode 0 none | 1 scale[e] | 2 scale[e*w_c + col % w_c]. */
long pcm_scatter_plan_ws_ints(long rows, int n_dst);
int pcm_scatter_plan_hip(long rows, int n_dst, const int *idx, int *ws, const int **start_out,
                         const int **list_out, void *stream);
/* The same CSR with every segment sorted ascending -- a function of idx alone, so a sum taken in list order is
 * reproducible from run to run (the unsorted plan fills in the order its atomics retire, like the reference's scatter).
 * start (n_dst + 1) and list (rows) are caller-owned; scratch: pcm_scatter_plan_sorted_scratch_ints(n_dst) ints.
 * rows + 3 * n_dst must stay below 2^31 (PCM_ERR_UNSUPPORTED otherwise; the same limit applies to pcm_scatter_plan_hip). */
long pcm_scatter_plan_sorted_scratch_ints(int n_dst);
int pcm_scatter_plan_sorted_hip(long rows, int n_dst, const int *idx, int *scratch, int *start, int *list,
                                void *stream);
int pcm_segment_sum_hip(long n_dst, int c, const int *start, int seglen, const int *list, const int *map,
                        int rowdiv, const float *scale, int scale_mode, int w_c, float sign,
                        const float *src, int src_stride, int src_off, float *dst, void *stream);

/* ---- K9 attention steps ---------------------------------------------------------------------
 * replaces attention_{relation,fusion}_step_{forward,backward}_cuda_launcher
 *          attention/attention_cuda_kernel.h (kernels: attention_cuda_kernel.cu:9-147) */
int pcm_attention_relation_step_forward_hip(int m, int g, int c, const float *query,
                                            const float *key, const float *weight,
                                            const int *index_target, const int *index_refer,
                                            float *output, void *stream);
int pcm_attention_relation_step_backward_hip(int m, int g, int c, const float *query,
                                             float *grad_query, const float *key, float *grad_key,
                                             const float *weight, float *grad_weight,
                                             const int *index_target, const int *index_refer,
                                             const float *grad_output, void *stream);
int pcm_attention_fusion_step_forward_hip(int m, int g, int c, const float *weight,
                                          const float *value, const int *index_target,
                                          const int *index_refer, float *output, void *stream);
int pcm_attention_fusion_step_backward_hip(int m, int g, int c, const float *weight,
                                           float *grad_weight, const float *value,
                                           float *grad_value, const int *index_target,
                                           const int *index_refer, const float *grad_output,
                                           void *stream);

/* =============================================================================================
 * Fused entry points with no native counterpart in the reference: they replace pure-PyTorch code
 * on the hot path (same results, fewer HBM passes).
 * ============================================================================================= */

/* pcm_knn_query_hip with the number of clouds b made explicit (b = offset length): the cloud of a
 * query is found by bisection instead of the reference's linear get_bt_idx scan.  Same outputs. */
int pcm_knn_query_b_hip(int b, int m, int nsample, const float *xyz, const float *new_xyz,
                        const int *offset, const int *new_offset, int *idx, float *dist2,
                        void *stream);
/* ... and with the size of the largest cloud, n_max (0 = unknown), when the caller knows it on the host (reserved for
 * size-specialised variants; the results never depend on it). */
int pcm_knn_query_n_hip(int b, int n_max, int m, int nsample, const float *xyz, const float *new_xyz,
                        const int *offset, const int *new_offset, int *idx, float *dist2, void *stream);

/* grouping(idx, feat, xyz, new_xyz, with_xyz)   functions/grouping.py:35-59
 * with_xyz (xyz and new_xyz non-null): output (m, nsample, 3+c) =
 *     [ (xyz[idx]-new_xyz[row]) * (idx != -1), feat[idx] or 0 ];
 * features only (xyz == new_xyz == NULL): output (m, nsample, c) = feat[idx] or 0 for idx == -1. */
int pcm_group_xyz_feat_forward_hip(int m, int nsample, int c, const float *xyz,
                                   const float *new_xyz, const float *feat, const int *idx,
                                   float *output, void *stream);
/* backward of the above w.r.t. feat: grad_feat(n,c) += grad_output[..., xc:] scattered by idx,
 * xc = with_xyz ? 3 : 0 (rows with idx == -1 skipped; grad_feat pre-zeroed by the caller). */
int pcm_group_xyz_feat_backward_hip(int m, int nsample, int c, int with_xyz,
                                    const float *grad_output, const int *idx, float *grad_feat,
                                    void *stream);

/* ---- fused set-abstraction layer ------------------------------------------------------------------
 * replaces, after FPS + kNN, the framework ops of ACTPCD.pcd_sampling / PCDObsEncoder.pcd_sampling
 * (src/models/components/act/act.py:446-460; diffusion_policy/vision/pcd_obs_encoder.py:174-190):
 * grouping(with_xyz) -> Linear(3+C->H, no bias) -> BatchNorm1d(H) (training statistics over the m*K
 * rows) -> ReLU -> max over K.  The caller supplies Gf = feat @ Wf^T (n,H) (fp32 or bf16, one GEMM on
 * the n points); the xyz part Wp (p_j - q_i) is added in fp32 inside the gather.  Nothing of size
 * m*K*H is ever written.  sign(a) = sign(gamma) is known before the statistics, so the gather keeps ONE extremum per
 * (query, channel): sel = (gamma >= 0 ? max_s y_s : min_s y_s) and its slot asel.  Workspaces are caller-allocated
 * (sizes in policy/sa_fused.py):
 *   forward : sel (m,H) f32; asel (m,H) u8; partial (slots,5,H); sums (2,H); stat (4,H) = {mean, invstd, a, b};
 *             z (m,H) the tokens.  running_mean/var updated in place (or NULL).
 *   index   : pcm_sa_index_hip -> ent (m,K) 16-byte records (j, p_j - q_i) [j = -1: (-1,0,0,0)], cnt (n), S (n,3),
 *             RM (12): occurrence count / summed relative coordinates of every point and the 3 + 9 global moments.
 *             A function of (idx, p, q) only, so it runs next to the kNN query; cnt / S / RM zeroed by the caller.
 *             The gather and the backward read `ent` instead of chasing idx -> p / q.
 *   backward: D (n,H) scratch (zeroed by the caller only for the global-atomic fallback); red1 (5,H), red2 (3,H);
 *             outputs dGf (n,H) in Gf's dtype, dWp (H,3), dgamma (H), dbeta (H).
 * offset / new_offset (b) + n_max (largest cloud) select the LDS-staged scatter (one workgroup per cloud
 * and channel chunk, ds_add_f32, no global atomics, D need not be zeroed); pass NULL / b = 0 for the
 * global-atomic fallback (D zeroed by the caller).
 * stage_mask <= 0 runs every kernel of the call; a bit mask runs only the selected kernels (forward:
 * 1 gather+stats, 2 reduce, 4 affine, 8 apply; backward: 2 bwd1, 4 reduce, 8 bwd2, 16 reduce,
 * 32 bwd3) -- used by bench.py to time one kernel at a time, and by synchronised BatchNorm: forward 1|2 yields the local
 * sums, the caller combines the statistics of all ranks into `stat` and runs stage 8; backward 2|4 yields red1, the
 * caller all-reduces its first two rows into red_global (2,H) = {sum delta, sum delta * yhat} and runs 8|16|32 with
 * `count` = m*K of the global batch (red_global NULL / count <= 0: single rank).
 * pcm_sa_fused_slots(rows, H, bf16, K): partial-row slots a row-streaming kernel over `rows` rows writes. */
int pcm_sa_fused_slots(int units, int H, int bf16, int K);
/* the partial-row buffer must hold pcm_sa_fused_reduce_scratch_rows() rows beyond the slots (second level of the reductions) */
int pcm_sa_fused_reduce_scratch_rows(void);
int pcm_sa_reduce_rows_hip(int nslots, int VH, const float *partial, float *scratch, float *out, void *stream);
int pcm_sa_fused_bwd1_lds_channels(int H, int n_max);
int pcm_sa_fused_forward_hip(int m, int K, int H, int gf_is_bf16, const void *Gf, const void *ent,
                             const float *Wp, const float *gamma, const float *beta, float eps,
                             float momentum, float *running_mean, float *running_var, float *sel,
                             unsigned char *asel, float *partial, float *sums, float *stat, float *z,
                             int stage_mask, void *stream);
int pcm_sa_index_hip(int m, int K, const float *p, const float *q, const int *idx, const int *offset,
                     const int *new_offset, int b, int n_max, void *ent, float *cnt, float *S, float *RM,
                     void *stream);
int pcm_sa_fused_backward_hip(int m, int n, int K, int H, int gf_is_bf16, const void *Gf,
                              const void *ent, const float *Wp,
                              const float *stat, const float *dz, const float *sel,
                              const unsigned char *asel, float *D, const float *cnt, const float *S,
                              const float *RM, float *partial, float *red1, float *red2, void *dGf,
                              float *dWp, float *dgamma, float *dbeta, const int *offset,
                              const int *new_offset, int b, int n_max, const float *red_global, double count,
                              int stage_mask, void *stream);

/* Reproducible (atomic-free) forms of the index pass and of backward pass 1 (csrc/sa_scatter.hip; SURVEY.md section 5
 * "deterministic-mode option (sorted segmented reduce)").  The neighbour lists are inverted once per batch into a CSR with
 * sorted segments; cnt / S are segment sums in list order, RM a fixed-order two-level sum; the m*H deltas are bucketed per
 * query by arg-extremum slot ("pack": runs of (channel, delta)) and every D row is summed run by run in list order
 * ("gather").  Same results as the atomic kernels up to fp32 re-association, bit-identical from run to run.
 *   pcm_sa_index_det_hip : writes ent (m,K,4), csr = start (n+1) | list (m*K) ints, cnt (n), S (n,3), RM (12); nothing is
 *                          zeroed by the caller; scratch: pcm_sa_index_det_scratch_ints(n) ints.
 *   pcm_sa_bwd1_det_hip  : replaces stages 2|4 of pcm_sa_fused_backward_hip (run that with stage_mask 8|16|32 afterwards):
 *                          D (n,H) written entirely, red1 (5,H); partial: pcm_sa_bwd1_det_slots(m) * 5 * H floats;
 *                          ws: pcm_sa_bwd1_det_ws_bytes(m,K,H) bytes.  stage_mask 1 pack | 2 gather | 4 reduce (<= 0 all).
 *   pcm_sa_det_supported : K <= 63, H <= 1024. */
int pcm_sa_det_supported(int K, int H);
long pcm_sa_index_det_scratch_ints(int n);
int pcm_sa_index_entries_hip(int m, int K, const float *p, const float *q, const int *idx, void *ent, void *stream);
int pcm_sa_index_det_hip(int m, int K, int n, const float *p, const float *q, const int *idx, void *ent, int *csr,
                         int *scratch, float *cnt, float *S, float *RM, void *stream);
int pcm_sa_bwd1_det_slots(int m);
long pcm_sa_bwd1_det_ws_bytes(int m, int K, int H);
int pcm_sa_bwd1_det_hip(int m, int n, int K, int H, const float *dz, const float *sel, const unsigned char *asel,
                        const float *stat, const void *ent, const int *csr, void *ws, float *D, float *partial,
                        float *red1, int stage_mask, void *stream);

/* ---- attention for long query sets (csrc/attn_flash.hip) ----------------------------------------------------
 * Same contract as pcm_attn_small_*_hip (head_dim 64, bf16, key_padding_mask (B,S) bytes, counter-hash dropout keyed by
 * (seed, site, b, h, q, key), out / dout (B, L, H*64) contiguous, lse (B, H, L) fp32), organised for hundreds to
 * thousands of queries: a workgroup keeps 128 rows resident and streams 64-row tiles of the other side through
 * swizzled LDS (ds_read_b64_tr_b16 for the transposed operands).  Replaces the framework's flash kernels on the encoder
 * self-attention (reference: src/models/components/act/transformer.py:221,244-262 nn.MultiheadAttention).  Backward is
 * three launches (delta, dK/dV, dQ), atomic-free; `delta` is a (B, H, L) fp32 workspace. */
int pcm_attn_flash_supported(int L, int S, int head_dim);
int pcm_attn_flash_forward_hip(int B, int H, int L, int S, const void *q, long q_bs, long q_ls, const void *k, long k_bs,
                               long k_ls, const void *v, long v_bs, long v_ls, const unsigned char *key_padding_mask,
                               float scale, float p_drop, const long *seed, unsigned site, void *out, float *lse,
                               void *stream);
int pcm_attn_flash_backward_hip(int B, int H, int L, int S, const void *q, long q_bs, long q_ls, const void *k, long k_bs,
                                long k_ls, const void *v, long v_bs, long v_ls, const unsigned char *key_padding_mask,
                                float scale, float p_drop, const long *seed, unsigned site, const void *out,
                                const void *dout, const float *lse, float *delta, void *dq, long dq_bs, long dq_ls, void *dk,
                                long dk_bs, long dk_ls, void *dv, long dv_bs, long dv_ls, void *stream);
/* the same with a stage mask (1 delta | 2 dK, dV | 4 dQ; <= 0 all): lets bench.py time one kernel at a time */
int pcm_attn_flash_backward_stages_hip(int B, int H, int L, int S, const void *q, long q_bs, long q_ls, const void *k,
                                       long k_bs, long k_ls, const void *v, long v_bs, long v_ls,
                                       const unsigned char *key_padding_mask, float scale, float p_drop, const long *seed,
                                       unsigned site, const void *out, const void *dout, const float *lse, float *delta,
                                       void *dq, long dq_bs, long dq_ls, void *dk, long dk_bs, long dk_ls, void *dv, long dv_bs,
                                       long dv_ls, int stage_mask, void *stream);

/* ---- hipGraph surgery -----------------------------------------------------------------------------------
 * Replace every MEMSET node of a captured, not yet instantiated hipGraph_t by a fill-kernel node with the same
 * destination, value, extent and dependencies (csrc/graph_fix.hip: memset nodes created by stream capture replay
 * with a garbage pattern on ROCm 7.2; ATen reductions zero their semaphores with one).  *n_replaced (may be NULL)
 * receives the number of nodes replaced. */
int pcm_graph_replace_memsets(void *graph, int *n_replaced);
/* HIP runtime version the library is bound to (HIP_VERSION encoding) and a plain asynchronous memset issued through that same
 * binding (d32 == 0: `count` bytes, else `count` dwords): what the host side uses to decide whether the rewrite above is needed
 * (pointcloudmatters_amd/_graphs.py: always on runtimes <= 7.2.x, where the defect was found; a captured-memset self-test
 * decides on newer ones). */
int pcm_hip_runtime_version(int *version);
int pcm_memset_async(void *dst, int value, long count, int d32, void *stream);

/* ---- fused  out = LayerNorm(x + dropout(y))  ------------------------------------------------------------
 * replaces the `x = x + dropout(y); x = norm(x)` tail of every post-norm transformer sub-layer
 * (src/models/components/act/transformer.py:250-256, 330-345) -- 4 framework launches forward, 6+
 * backward -- by one kernel each way.  x, out, s, dout, dx: (R,E) fp32; y, dy: (R,E) bf16 (y_is_bf16) or
 * fp32; E % 256 == 0, E <= 1024 (else PCM_ERR_UNSUPPORTED).  The dropout mask is a counter-based hash of
 * (*seed, site, element) recomputed in backward; `seed` is a DEVICE int64 so hipGraph replays draw new
 * masks; p_drop = 0 disables dropout (seed may be NULL).  backward also reduces dgamma | dbeta | dysum (3,E)
 * from `partial` (pcm_drln_blocks(R) x 3 x E floats of scratch); dysum = column sums of dy, i.e. the bias
 * gradient of the projection that produced y; dysum_bf16 (E, may be NULL) receives the same sums rounded to bf16. */
int pcm_drln_blocks(long R);
int pcm_drln_forward_hip(long R, int E, int y_is_bf16, const float *x, const void *y, const float *gamma,
                         const float *beta, float eps, float p_drop, const long *seed, unsigned site,
                         float *s, float *out, float *mean, float *rstd, void *stream);
/* same, and the consumer's bf16 operands emitted by the same launch (replaces its pcm_add_cast2_hip): sum_bf16 (nullable) =
 * bf16(out + pos) with pos (pos_n elements, a multiple of E that divides R*E) broadcast over the leading rows -- the
 * decoder's cross-attention query input tgt + query_pos, transformer.py:332 -- and out_bf16 (nullable) = bf16(out) */
int pcm_drln_forward2_hip(long R, int E, int y_is_bf16, const float *x, const void *y, const float *gamma,
                          const float *beta, float eps, float p_drop, const long *seed, unsigned site, float *s,
                          float *out, float *mean, float *rstd, const float *pos, long pos_n, void *sum_bf16,
                          void *out_bf16, void *stream);
int pcm_drln_backward_hip(long R, int E, int y_is_bf16, const float *dout, const float *s,
                          const float *mean, const float *rstd, const float *gamma, float p_drop,
                          const long *seed, unsigned site, float *dx, void *dy, float *partial,
                          float *dgamma_dbeta, void *dysum_bf16, void *stream);
/* same with the output gradient given as TWO addends (dout2 nullable): the layer output had two consumers in the graph (the
 * decoder's norm1 output feeds the cross-attention query AND the residual, transformer.py:330-338) and their gradients
 * are summed while loading instead of by the autograd engine's add launch */
int pcm_drln_backward2_hip(long R, int E, int y_is_bf16, const float *dout, const float *dout2, const float *s,
                           const float *mean, const float *rstd, const float *gamma, float p_drop, const long *seed,
                           unsigned site, float *dx, void *dy, float *partial, float *dgamma_dbeta, void *dysum_bf16,
                           void *stream);

/* ---- the attention output projection, the residual add and the norm as ONE kernel on the matrix cores (csrc/proj_ln.hip) ----------
 *   out = LayerNorm(x + dropout(a W^T + bias))      transformer.py:244-256 (encoder), :296-346 (decoder): out_proj of
 *   nn.MultiheadAttention followed by `src = src + self.dropout1(src2); src = self.norm1(src)`
 * a (R, K) bf16 with row stride a_ls elements (a multiple of 8), W (E, K) bf16 row-major (nn.Linear.weight under autocast), bias (E)
 * bf16 or fp32 (nullable); the product is rounded to bf16 like the library GEMM's output under autocast, everything else and every
 * other argument as pcm_drln_forward2_hip -- same dropout counter hash, so pcm_drln_backward2_hip is the backward of both.
 * E in {256, 512, 768, 1024}, K a multiple of 32 up to 1024 (pcm_proj_drln_mfma_supported), 16-byte aligned a and W. */
int pcm_proj_drln_mfma_supported(int E, int K);
int pcm_proj_drln_mfma_forward_hip(long R, int E, int K, const void *a_bf16, long a_ls, const void *w_bf16, const void *bias,
                                   int bias_is_bf16, const float *x, const float *gamma, const float *beta, float eps, float p_drop,
                                   const long *seed, unsigned site, float *s, float *out, float *mean, float *rstd, const float *pos,
                                   long pos_n, void *sum_bf16, void *out_bf16, void *stream);

/* The BACKWARD of that chain for the short sites as ONE kernel (csrc/proj_ln.hip, round 6; opt-in, PCM_PROJ_MFMA_BWD): pcm_drln_backward2_hip's
 * row code -- dx, dy (bf16), the partial rows of dgamma | dbeta | column sums of dy: same arguments, same bits per row -- AND the input
 * gradient of the projection, da (R, K) bf16 with row stride da_ls = dy W, W (E, K) bf16 row-major (what autograd computes as
 * `grad_output @ weight` for nn.Linear, transformer.py:244-256, 296-346).  partial: pcm_proj_drln_mfma_backward_blocks(R) rows of 3 E
 * floats; dgamma_dbeta (3, E) fp32 nullable (NULL: the rows are left for pcm_reduce_batch_hip), dysum_bf16 (E) nullable.
 * E in {256, 512, 768, 1024}, K in {256, 512, 1024}; 16-byte aligned W and da, da_ls a multiple of 8. */
int pcm_proj_drln_mfma_backward_supported(int E, int K);
int pcm_proj_drln_mfma_backward_blocks(long R);
int pcm_proj_drln_mfma_backward_hip(long R, int E, int K, const float *dout, const float *dout2, const float *s, const float *mean,
                                    const float *rstd, const float *gamma, float p_drop, const long *seed, unsigned site,
                                    const void *w_bf16, float *dx, void *dy_bf16, void *da_bf16, long da_ls, float *partial,
                                    float *dgamma_dbeta, void *dysum_bf16, void *stream);

/* The INPUT gradient of the short in-projections as one kernel (csrc/proj_ln.hip, round 6; opt-in, PCM_LINEAR_MFMA_BWD):
 *   dx (R, K) fp32 = dy (R, N) W (N, K) [+ dres (R, K) fp32, nullable];   dpos (R, K) fp32, nullable = dy[:, :pos_cols] W[:pos_cols]
 * dy bf16 with row stride dy_ls (a multiple of 8), W (N, K) bf16 row-major: autograd's `grad_output @ weight` for the packed
 * self-attention in-projection of nn.MultiheadAttention (N = 3 E: dq | dk | dv side by side, pos_cols = 2 E: q = k = x + pos, v = x;
 * transformer.py:244-262, 296-346) and for the cross-attention query projection (N = E, pos_cols >= N).
 * N % 32 == 0 (<= 3072), K in {256, 512, 1024}, pos_cols % 32 == 0 or >= N; 16-byte aligned pointers. */
int pcm_linear_mfma_backward_supported(int N, int K, int pos_cols);
int pcm_linear_mfma_backward_hip(long R, int N, int K, const void *dy_bf16, long dy_ls, const void *w_bf16, const float *dres, float *dx,
                                 float *dpos, int pos_cols, void *stream);

/* out = A W^T + bias for short activations on the matrix cores, operand preparation fused in (csrc/proj_ln.hip): the in-projections of
 * nn.MultiheadAttention and the cross-attention query projection (transformer.py:244-262, 296-346: `q = k = with_pos_embed(x, pos)`).
 * A: a_is_f32 == 0: bf16 (R, K), row stride a_ls elements, for the output columns [0, pos_cols), and a_alt_bf16 (nullable, same layout)
 * for the others; a_is_f32 == 1: fp32 x (R, K), row stride a_ls, converted to bf16 on the way in, with pos (nullable; pos_n fp32
 * elements, a multiple of K, broadcast over the leading rows) ADDED first for the output columns [0, pos_cols) -- q and k of the packed
 * in-projection -- and not for the others (v); emit_pos_bf16 / emit_x_bf16 (nullable, (R, K) bf16 contiguous) receive bf16(x + pos) /
 * bf16(x), the operands of the backward's weight-gradient products.  W (N, K) bf16 row-major; bias (N) bf16 / fp32 / NULL; out (R, N)
 * bf16 or fp32 with row stride out_ls.  K % 32 == 0 (<= 1024), N % 8 == 0, pos_cols a multiple of 256 or >= N. */
int pcm_linear_mfma_supported(int N, int K, int pos_cols);
int pcm_linear_mfma_forward_hip(long R, int N, int K, const void *a, int a_is_f32, long a_ls, const void *a_alt_bf16, const float *pos,
                                long pos_n, int pos_cols, const void *w_bf16, const void *bias, int bias_is_bf16, void *out,
                                int out_is_bf16, long out_ls, void *emit_pos_bf16, void *emit_x_bf16, void *stream);

/* ---- fused feed-forward sub-layer  out = LayerNorm(x + dropout(W2 dropout(relu(W1 x + b1)) + b2)) ------------
 * replaces linear1 -> relu -> dropout -> linear2 -> dropout -> add -> norm of every transformer layer
 * (src/models/components/act/transformer.py:253-256, 342-345) for the shipped dim_feedforward = 32
 * (F == 32, E in {256, 512}; pcm_ffn_ln_supported).  All tensors fp32: x, s, out, dout, dx, dy (R,E);
 * hd, dh (R,F); W1 (F,E), W2 (E,F).  backward writes dy and dh so the caller forms dW2 = dy^T hd and
 * dW1 = dh^T x with two GEMMs, and reduces `partial` (pcm_ffn_ln_blocks(R) x (3E+F)) into
 * sums = [dgamma(E) | dbeta(E) | db2(E) | db1(F)].  Dropout masks as in pcm_drln_* (device seed + site). */
int pcm_ffn_ln_supported(int E, int F);
int pcm_ffn_ln_blocks(long R);
int pcm_ffn_ln_forward_hip(long R, int E, int F, const float *x, const float *W1, const float *b1,
                           const float *W2, const float *b2, const float *gamma, const float *beta,
                           float eps, float p_hidden, float p_out, const long *seed, unsigned site_a,
                           unsigned site_b, float *hd, float *s, float *out, float *mean, float *rstd,
                           void *stream);
/* same, emitting the NEXT layer's in-projection operands bf16(out + pos) / bf16(out) (see pcm_drln_forward2_hip;
 * q = k = src + pos, v = src of the next layer, transformer.py:244-249) */
int pcm_ffn_ln_forward2_hip(long R, int E, int F, const float *x, const float *W1, const float *b1, const float *W2,
                            const float *b2, const float *gamma, const float *beta, float eps, float p_hidden,
                            float p_out, const long *seed, unsigned site_a, unsigned site_b, float *hd, float *s,
                            float *out, float *mean, float *rstd, const float *pos, long pos_n, void *sum_bf16,
                            void *out_bf16, void *stream);
int pcm_ffn_ln_backward_hip(long R, int E, int F, const float *dout, const float *x, const float *s,
                            const float *mean, const float *rstd, const float *hd, const float *W1,
                            const float *W2, const float *gamma, float p_hidden, float p_out,
                            const long *seed, unsigned site_b, float *dx, float *dy, float *dh,
                            float *partial, float *sums, void *stream);
/* same, output gradient as two addends (dout2 nullable), see pcm_drln_backward2_hip (here: a decoder layer's output feeds
 * the next layer AND the stack of intermediate outputs, transformer.py:185-199) */
int pcm_ffn_ln_backward2_hip(long R, int E, int F, const float *dout, const float *dout2, const float *x, const float *s,
                             const float *mean, const float *rstd, const float *hd, const float *W1, const float *W2,
                             const float *gamma, float p_hidden, float p_out, const long *seed, unsigned site_b,
                             float *dx, float *dy, float *dh, float *partial, float *sums, void *stream);

/* ---- the same sub-layer on the matrix cores (csrc/ffn_mfma.hip), the bf16-autocast path ---------------------
 * Same argument lists as pcm_ffn_ln_forward2_hip / pcm_ffn_ln_backward2_hip and the same dropout hash; the two products run as
 * v_mfma_f32_16x16x32_bf16 (h and y leave their GEMMs as bf16, like F.linear under autocast; transformer.py:253-256, 342-345),
 * everything stored stays fp32.  Rows are processed in 16-row tiles: `partial` has pcm_ffn_ln_mfma_blocks(R) = ceil(R / 16) rows of
 * 3E + F -- always size it from that function, never from a tile height assumed by the caller. */
int pcm_ffn_ln_mfma_supported(int E, int F);
int pcm_ffn_ln_mfma_blocks(long R);
int pcm_ffn_ln_mfma_forward_hip(long R, int E, int F, const float *x, const float *W1, const float *b1, const float *W2,
                                const float *b2, const float *gamma, const float *beta, float eps, float p_hidden,
                                float p_out, const long *seed, unsigned site_a, unsigned site_b, float *hd, float *s,
                                float *out, float *mean, float *rstd, const float *pos, long pos_n, void *sum_bf16,
                                void *out_bf16, void *stream);
int pcm_ffn_ln_mfma_backward_hip(long R, int E, int F, const float *dout, const float *dout2, const float *x, const float *s,
                                 const float *mean, const float *rstd, const float *hd, const float *W1, const float *W2,
                                 const float *gamma, float p_hidden, float p_out, const long *seed, unsigned site_b,
                                 float *dx, float *dy, float *dh, float *partial, float *sums, void *stream);
/* out[e] = sum over the nslots rows of partial (nslots x VH), fp64 accumulation in a fixed order */
int pcm_ffn_reduce_rows_hip(int nslots, int VH, const float *partial, float *out, void *stream);

/* ---- Diffusion-Policy sampler: one fused DDPM reverse step x_t -> x_{t-1} ------------------------------
 * replaces diffusers' DDPMScheduler.step + the conditioning re-imposition inside `conditional_sample`
 * (src/models/components/diffusion_policy/diffusion_unet_image_policy.py:130-141): epsilon prediction,
 * fixed_small variance, clip_sample (configs/model/maniskill2_diffusion_policy_model.yaml:30-38).
 *   x0 = (xt - sqrt_one_minus_abar*eps)/sqrt_abar; clamp to +-clip (clip > 0);
 *   prev = coef_x0*x0 + coef_xt*xt (+ sigma*noise when noise != NULL and sigma != 0); prev = cond where cond_mask.
 * n elements; eps bf16 (eps_is_bf16) or fp32; xt, noise, cond, prev fp32; cond_mask bytes (NULL: none).
 * prev may alias xt.  fp32, un-contracted, in that order. */
int pcm_ddpm_step_hip(long n, int eps_is_bf16, const void *eps, const float *xt, const float *noise,
                      const unsigned char *cond_mask, const float *cond, float sqrt_abar,
                      float sqrt_one_minus_abar, float coef_x0, float coef_xt, float sigma, float clip,
                      float *prev, void *stream);

/* ---- Diffusion-Policy U-Net blocks, channels-last --------------------------------------------------------
 * replace Conv1dBlock = Conv1d -> GroupNorm -> Mish (+ FiLM, + residual add) of
 * src/models/components/diffusion_policy/diffusion/conv1d_components.py:25-45 and conditional_unet1d.py:56-75.
 * Activations are (B, T, C) row-major ("channels-last"); x / res / film may be bf16 (flags), y, dy are fp32.
 *   pcm_im2col_cl_hip : cols (B*L_out, C*K), column c*K+k = x[b, l*stride+k-pad, c] or 0; L_out = (T+2*pad-K)/stride+1;
 *                       the GEMM with nn.Conv1d's weight viewed (C_out, C_in*K) then IS the convolution.
 *   pcm_col2im_cl_hip : adjoint (dx from dcols).
 *   pcm_gn_mish_*     : y = mish(GroupNorm_G(x)); film_mode 1: y = film[b,0,c]*y + film[b,1,c] (film (B,2,C));
 *                       film_mode 2: y += film[b,c]; res != NULL: y += res.  mean/rstd: (B*G).  backward writes dx
 *                       (x's dtype), dgb_partial (B,3,C) = per-sample dgamma | dbeta | dconv_bias (sum over B is the
 *                       gradient) and dfilm (fp32, film's shape).  conv_bias (C, or NULL) is added to x on load: the
 *                       bias of the convolution that produced x, whose gradient (sum of dx) then comes for free.
 *                       Supported when T*C/G <= 6656 and C/G <= 1024
 *                       (pcm_gn_mish_supported; PCM_ERR_UNSUPPORTED otherwise). */
int pcm_gn_mish_supported(int T, int C, int G);
int pcm_gn_mish_forward_hip(int B, int T, int C, int G, int x_is_bf16, const void *x, const float *gamma,
                            const float *beta, float eps, int film_mode, int film_is_bf16, const void *film,
                            int res_is_bf16, const void *res, const float *conv_bias, float *y, float *mean,
                            float *rstd, void *stream);
int pcm_gn_mish_backward_hip(int B, int T, int C, int G, int x_is_bf16, const void *x, const float *gamma,
                             const float *beta, const float *mean, const float *rstd, int film_mode,
                             int film_is_bf16, const void *film, const float *conv_bias, const float *dy, void *dx,
                             float *dgb_partial, float *dfilm, void *stream);
int pcm_im2col_cl_hip(int B, int T, int C, int K, int stride, int pad, int x_is_bf16, const void *x,
                      int out_is_bf16, void *cols, void *stream);
int pcm_col2im_cl_hip(int B, int T, int C, int K, int stride, int pad, int cols_is_bf16, const void *dcols,
                      int dx_is_bf16, void *dx, void *stream);

/* ---- PointNet layer tail: BatchNorm1d (batch statistics) + ReLU over packed point features (n, C) ----------
 * replaces the BatchNorm1d + ReLU of every PointNet layer (src/models/components/pcd_encoder/pointnet.py:25-56).
 * y, z, dz, dy: (n, C) row-major, all bf16 (is_bf16) or all fp32; C % 4 == 0 and (C <= 1024 or C % 1024 == 0)
 * (pcm_bn_relu_supported).  partial: pcm_bn_relu_slots(n, C) x 2 x C floats of scratch; sums: 2 x C; stat: 4 x C =
 * { mean, invstd, a = gamma*invstd, b = beta - a*mean }.  forward updates running_mean / running_var (momentum,
 * unbiased variance) unless they are NULL; use_given_stat = 1 skips the statistics and applies `stat` as given
 * (eval mode: the caller fills it from the running statistics; synchronised BatchNorm: from the statistics of all ranks),
 * use_given_stat = 2 stops after the local sums (sums[0] = sum (y - y[0]), sums[1] = sum (y - y[0])^2).
 * backward leaves sums = [dbeta | dgamma]; phase 1 stops after those local sums, phase 2 only applies given (all-reduced)
 * sums with `count` = rows of the global batch (<= 0: n). */
/* synchronised BatchNorm (torch.nn.SyncBatchNorm semantics, configs/trainer/ddp.yaml:9 `sync_batchnorm: true`): the host-side
 * statistics exchange around the collective as two launches instead of ~28 framework launches per layer.
 * pack: (mean, M2, count) of this rank from the forward kernels' shifted sums (sums (2, C): sum (y - shift), sum (y - shift)^2;
 * shift = row *row_index of the fp32 / bf16 (rows, C) matrix src -- row 0 when row_index is NULL, zeros when it is negative:
 * the row the forward kernel accumulated around) -> pack (2C + 1).  combine: the gathered packs of all W ranks (W, 2C + 1) -> stat (4, C) of the global batch
 * (fp64, rank order), running statistics (nullable pair), ratio[0] = count_local / N. */
int pcm_bn_sync_pack_hip(int C, double count, const float *sums, const void *src, int src_is_bf16, const int *row_index, float *pack,
                         void *stream);
int pcm_bn_sync_combine_hip(int W, int C, const float *gathered, const float *gamma, const float *beta, float eps, float momentum,
                            float *running_mean, float *running_var, double count_local, float *stat, float *ratio, void *stream);
int pcm_bn_relu_supported(long n, int C);
int pcm_bn_relu_slots(long n, int C);
int pcm_bn_relu_forward_hip(long n, int C, int is_bf16, const void *y, const float *gamma, const float *beta,
                            float eps, float momentum, float *running_mean, float *running_var,
                            int use_given_stat, float *partial, float *sums, float *stat, void *z, void *stream);
int pcm_bn_relu_backward_hip(long n, int C, int is_bf16, const void *y, const void *dz, const float *stat,
                             float *partial, float *sums, void *dy, int phase, double count, void *stream);
/* the same pair with the activation as an argument: relu != 0 is pcm_bn_relu_*; relu == 0 is BatchNorm1d alone -- the last layer of the
 * Diffusion Policy's projector (pcd_obs_encoder.py:100-120) -- so that every BatchNorm of PCDObsEncoder is owned by these kernels
 * (synchronised statistics inside them, no torch.nn.SyncBatchNorm module left: the data-parallel step can be captured as a chain) */
int pcm_bn_act_forward_hip(long n, int C, int is_bf16, int relu, const void *y, const float *gamma, const float *beta, float eps,
                           float momentum, float *running_mean, float *running_var, int use_given_stat, float *partial, float *sums,
                           float *stat, void *z, void *stream);
int pcm_bn_act_backward_hip(long n, int C, int is_bf16, int relu, const void *y, const void *dz, const float *stat, float *partial,
                            float *sums, void *dy, int phase, double count, void *stream);

/* ---- GPU-side GridSamplePCD keys (the data-path step before the hot path) --------------------------------
 * replaces the per-cloud NumPy arithmetic of src/data/components/transformpcd.py:684-701, 776-790 for a packed
 * batch: gmin (b,3) = per-cloud min of floor(coord / grid_size) (float64 division, as NumPy >= 2 promotes it);
 * grid_coord (n,3) int64 = floor(...) - gmin[cloud]; key (n) = FNV-1a 64 of grid_coord (uint64 bit pattern);
 * cloud (n) = cloud index of each point.  Bit-exact with the reference's functions. */
int pcm_voxel_keys_hip(int n, int b, const float *coord, const int *offset, double grid_size, int *gmin,
                       long *grid_coord, long *key, int *cloud, void *stream);

/* ---- glue around the attention in-projection (src/models/components/act/transformer.py:244-249, 318-323) ------
 * pcm_add_cast2_hip : sum_bf16 = bf16(x + pos), x_bf16 = bf16(x) (x_bf16 may be NULL); x has n floats, pos has pos_n
 *                     floats and is repeated (n % pos_n == 0: a position table broadcast over the batch);
 * pcm_add2_cast_hip : out = f32(a) + f32(b) for two bf16 arrays of n elements;
 * pcm_colsum_hip    : out[t*C + c] = sum over rows of g_t[row*ld_t + c] for ntensors <= 3 matrices of the same dtype
 *                     (bias gradients of the q / k / v projections); `partial` = pcm_colsum_slots(rows, C) * ntensors * C
 *                     floats; out is fp32 or bf16 (out_is_bf16).  C % 4 == 0, C <= 1024; n, pos_n multiples of 4. */
int pcm_add_cast2_hip(long n, long pos_n, const float *x, const float *pos, void *sum_bf16, void *x_bf16, void *stream);
int pcm_add2_cast_hip(long n, const void *a_bf16, const void *b_bf16, float *out, void *stream);
/* same, and a_f32 (nullable) = f32(a): the first addend alone, widened in the same pass */
int pcm_add2_cast2_hip(long n, const void *a_bf16, const void *b_bf16, float *out, float *a_f32, void *stream);
/* same with a third, fp32 addend c (nullable): out = f32(a) + f32(b) + c */
int pcm_add3_cast2_hip(long n, const void *a_bf16, const void *b_bf16, const float *c_f32, float *out, float *a_f32,
                       void *stream);
/* same with the first addend in two parts: out = (f32(a) + f32(a2)) + f32(b) [+ c], a_f32 (nullable) = f32(a) + f32(a2) -- the
 * three input gradients dq W_q, dk W_k, dv W_v of a self-attention in-projection out of ONE batched product */
int pcm_add4_cast2_hip(long n, const void *a_bf16, const void *a2_bf16, const void *b_bf16, const float *c_f32, float *out,
                       float *a_f32, void *stream);
int pcm_colsum_slots(long rows, int C);
/* out[e] = sum over nslabs of partial[s*n + e] (fp64, fixed order), written as fp32 or bf16: closes a split-K product */
int pcm_slab_sum_hip(int nslabs, long n, const float *partial, int out_is_bf16, void *out, void *stream);
int pcm_colsum_hip(long rows, int C, int ntensors, int in_is_bf16, const void *g0, long ld0, const void *g1, long ld1,
                   const void *g2, long ld2, float *partial, int out_is_bf16, void *out, void *stream);
/* n closing reductions in ONE launch per 24: out[e] = sum_{s < nslots[i]} partial[i][s * width[i] + e], fp64 in the fixed
 * order of pcm_slab_sum_hip (bit-identical results).  out_f32[i] (nullable) takes all columns, out_bf16[i] (nullable) the
 * columns bf16_from[i]..width[i]-1 stored from index 0; at least one of the two per reduction.  Host arrays of length n.
 * The first stages leave their partial rows for it when called with a NULL result pointer: pcm_drln_backward_hip
 * (dgamma_dbeta = NULL: pcm_drln_blocks(R) rows of 3E), pcm_ffn_ln_backward_hip (sums = NULL: pcm_ffn_ln_blocks(R) rows of
 * 3E + F), pcm_colsum_hip (out = NULL: pcm_colsum_slots(rows, C) rows of ntensors*C).  The backward pass of the reference
 * runs one reduction kernel per bias / norm gradient (torch autograd, transformer.py:296-346); here a whole backward
 * stage closes all of them together (policy/deferred.py). */
int pcm_reduce_batch_hip(int n, const void *const *partial, const int *nslots, const int *width, void *const *out_f32,
                         void *const *out_bf16, const int *bf16_from, void *stream);
/* first stages of n column sums in one launch per 16: per job the arguments of pcm_colsum_hip as host arrays (g and ld
 * hold 3 entries per job); partial[i] receives pcm_colsum_slots(rows[i], C[i]) rows of ntensors[i]*C[i] sums, closed by
 * pcm_reduce_batch_hip.  Same arithmetic as pcm_colsum_hip. */
/* n device-to-device copies dst[i][0..nbytes[i]) = src[i][...] in one launch per 32 (host arrays; pairs must not overlap):
 * the input / index staging of a training step that replays captured graphs (what Lightning's batch transfer + the
 * reference's per-tensor `.to(device)` do one tensor at a time, maniskill2_act_bc_module.py:64-86) */
int pcm_copy_batch_hip(int n, void *const *dst, const void *const *src, const long *nbytes, void *stream);
/* *counters[i] += 1 for n distinct device int64 counters in one launch per 64 (BatchNorm's num_batches_tracked) */
int pcm_incr_i64_batch_hip(int n, void *const *counters, void *stream);
int pcm_colsum_batch_hip(int n, const long *rows, const int *C, const int *ntensors, const int *in_is_bf16, const void *const *g,
                         const long *ld, void *const *partial, void *stream);

/* ---- the CVAE latent head (src/models/components/act/act.py:175-181, act/utils.py:36-39) in one launch each way ------------
 * latent_info (B, 2D) fp32 or bf16 = [mu | logvar]; z (B, D) fp32 = mu + exp(logvar / 2) * eps with the framework's roundings
 * under bf16 autocast (the halving in the input dtype, exp / product / sum in fp32); eps_in (B, D) fp32 when the caller
 * supplies the noise (parity tests), else NULL: drawn from the counter hash of the dropout masks (seed: device int64, site).
 * Also writes contiguous copies mu, logvar (B, D, input dtype) for the KL term and eps_out / std_out (fp32) for backward.
 * backward: d_latent_info[:, :D] = T(dz) + dmu, [:, D:] = T((dz*eps)*std)/2 + dlogvar (any of dz / dmu / dlogvar may be NULL). */
int pcm_cvae_latent_forward_hip(int B, int D, int is_bf16, const void *latent_info, const float *eps_in, const long *seed,
                                unsigned site, float *z, void *mu, void *logvar, float *eps_out, float *std_out, void *stream);
int pcm_cvae_latent_backward_hip(int B, int D, int is_bf16, const float *dz, const void *dmu, const void *dlogvar,
                                 const float *eps, const float *std_, void *d_latent_info, void *stream);

/* ---- the ACT training loss (src/models/components/act/act.py:281-291, loss/misc.py:10-26) in one launch each way -----------
 * a_hat (B, Q, A) fp32 or bf16, actions (B, Q, A) fp32, is_pad (B, Q) bytes (non-zero = padded), mu / logvar (B, D) fp32 or
 * bf16; n = B*Q*A, bd = B*D.  action = mean over ALL n elements of (a_hat - actions)^2 * !is_pad (MSELoss(reduction="none")
 * then .mean()), kl = mean_b sum_d -0.5 (1 + logvar - mu^2 - exp(logvar)), loss = action + kl_weight * kl.
 * forward : stats[3] = {loss, action, kl}; ga (n), gmu (bd), glv (bd) fp32 = gradients per unit of upstream gradient.
 * backward: upstream gradients of (loss, action, kl) as device scalars (NULL = 0); da / dmu / dlv in the inputs' dtypes. */
int pcm_act_loss_forward_hip(int n, int A, int bd, int B, int a_is_bf16, const void *a_hat, const float *actions,
                             const unsigned char *is_pad, int l_is_bf16, const void *mu, const void *logvar, float kl_weight,
                             float *stats, float *ga, float *gmu, float *glv, void *stream);
int pcm_act_loss_backward_hip(int n, int bd, const float *g_loss, const float *g_action, const float *g_kl, float kl_weight,
                              const float *ga, const float *gmu, const float *glv, int a_is_bf16, void *da, int l_is_bf16,
                              void *dmu, void *dlv, void *stream);

/* ---- sine position embedding of the sampled centres (src/models/components/act/act.py:467-506, default arguments) ------
 * out (m, H) fp32: for axis a = x, y, z and npf = H / 3 (even), k = npf / 2:
 *     out[r][a*npf + j]     = sin(coord[r][a] / dim_t[2j]),      j < k
 *     out[r][a*npf + k + j] = cos(coord[r][a] / dim_t[2j + 1]),  j < k
 * (the reference's x_embed is (m, 1), so its stack(..., dim=2).flatten(1) yields the sine BLOCK followed by the cosine
 * block per axis, not DETR's interleaving), columns 3*npf .. H-1 = 0.  dim_t (npf) = temperature ** (2 * (i // 2) / npf) is
 * passed in (computed once by the caller with the reference's own expression).  One launch instead of ~18 framework kernels. */
int pcm_coord_embed_sine_hip(long m, int H, int npf, const float *coord, const float *dim_t, float *out, void *stream);

/* ---- multi-head attention for short query sequences (L <= 128 queries, head_dim 64), MFMA -----------------------
 * replaces nn.MultiheadAttention's scaled-dot-product core for the CVAE encoder and the decoder of
 * src/models/components/act/transformer.py:225-262, 296-346 (100-102 queries; 18 of 22 attention calls per step).
 * q (B,L,.), k, v (B,S,.): bf16, element strides *_bs (batch) / *_ls (row), head h occupies columns [64h, 64h+64);
 * strides multiples of 8.  key_padding_mask (B,S) bytes, non-zero = ignore, or NULL.  out (B,L,H*64) bf16 contiguous,
 * lse (B,H,L) fp32.  Dropout on the attention weights: p_drop with a DEVICE seed + call site (as pcm_drln_*).
 * backward: dout (B,L,H*64) bf16 contiguous; dq/dk/dv bf16 with their own strides (multiples of 8), written fully; a (batch, head) whose dout is all zero
 * takes a fast path that writes exact zeros. */
int pcm_attn_small_supported(int L, int S, int head_dim);
int pcm_attn_small_forward_hip(int B, int H, int L, int S, const void *q, long q_bs, long q_ls, const void *k,
                               long k_bs, long k_ls, const void *v, long v_bs, long v_ls,
                               const unsigned char *key_padding_mask, float scale, float p_drop, const long *seed,
                               unsigned site, void *out, float *lse, void *stream);
int pcm_attn_small_backward_hip(int B, int H, int L, int S, const void *q, long q_bs, long q_ls, const void *k,
                                long k_bs, long k_ls, const void *v, long v_bs, long v_ls,
                                const unsigned char *key_padding_mask, float scale, float p_drop, const long *seed,
                                unsigned site, const void *out, const void *dout, const float *lse, void *dq,
                                long dq_bs, long dq_ls, void *dk, long dk_bs, long dk_ls, void *dv, long dv_bs,
                                long dv_ls, void *stream);

/* ---- training-step tail: global-norm clip + AdamW over one flat fp32 buffer ---------------------
 * replaces the torch passes the reference runs per step: clip_grad_norm_ (configs/trainer/ddp.yaml:12)
 * and AdamW.step (src/models/maniskill2_act_bc_module.py:347-367).  p, g, m, v: n floats each,
 * 16-byte aligned.  hyper: 9 floats ON THE DEVICE = { lr, beta1, beta2, eps, weight_decay,
 * 1-beta1^t, sqrt(1-beta2^t), max_norm (<=0: no clipping), grad_scale } so a captured hipGraph can
 * replay the launches while the host rewrites only this array.
 *   pcm_grad_sumsq_hip  writes <= pcm_optim_partials_capacity() per-block sums of g^2 into
 *                       `partials` (fixed grid: deterministic) and their count into *npartials_out;
 *   pcm_adamw_flat_hip  reduces the partials, derives clip_coef = min(1, max_norm/(norm+1e-6)) and
 *                       applies torch.optim.AdamW's update; norm_out (optional) receives the norm;
 *                       p_bf16 (optional, n bf16) receives a bf16 mirror of the updated weights. */
int pcm_optim_partials_capacity(void);
int pcm_grad_sumsq_hip(long n, const float *g, float *partials, int *npartials_out, void *stream);
int pcm_adamw_flat_hip(long n, float *p, const float *g, float *m, float *v, const float *hyper,
                       const float *partials, int npartials, float *norm_out, void *p_bf16,
                       void *stream);

/* Gradient hand-off into the flat fp32 gradient buffer (and the packed-gradient assembly copies), n jobs per call, 96 per launch, job
 * tables passed by value (capturable).  Replaces the per-tensor accumulate / bucket copies of the reference's loop
 * (src/models/maniskill2_act_bc_module.py:64-86 under Lightning + DDP).  kind[i]:
 *   PCM_XFER_ZERO      dst[i][0..numel) = 0                  (fp32 destination, src ignored)
 *   PCM_XFER_SET_BF16  dst (fp32) = src (bf16)               PCM_XFER_SET_F32   dst (fp32) = src (fp32)
 *   PCM_XFER_ADD_BF16  dst (fp32) += src (bf16, exact widen) PCM_XFER_ADD_F32   dst (fp32) += src (fp32)
 *   PCM_XFER_COPY_2B   raw copy of numel 2-byte elements
 * Jobs must not overlap; numel < 2^31 each.  Any alignment is accepted (16-byte aligned pairs take the vector path). */
#define PCM_XFER_ZERO 0
#define PCM_XFER_SET_BF16 1
#define PCM_XFER_SET_F32 2
#define PCM_XFER_ADD_BF16 3
#define PCM_XFER_ADD_F32 4
#define PCM_XFER_COPY_2B 5
int pcm_xfer_batch_hip(int n, void *const *dst, const void *const *src, const long *numel, const int *kind, void *stream);

#ifdef __cplusplus
}
#endif
#endif
