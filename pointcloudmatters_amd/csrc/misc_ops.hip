// misc_ops.hip -- interpolation (K6), subtraction (K7), aggregation (K8) and the scatter-attention
// steps (K9) for gfx950.  API-completeness kernels: none has a call site in the reference's src/
// (SURVEY.md F3); they are HBM-bound gathers / atomics and keep the reference's per-element
// arithmetic order (un-contracted fp32, `out += a * b` with k / nsample ascending).
//
// Replaces, under /root/reference/libs/pointops/src/:
//   interpolation/interpolation_cuda_kernel.cu:5-47, subtraction/subtraction_cuda_kernel.cu:5-44,
//   aggregation/aggregation_cuda_kernel.cu:5-53, attention/attention_cuda_kernel.cu:9-147.
// Thread mapping: channel index fastest (coalesced along c), 64-bit element indices (the
// reference's int32 index math overflows past 2^31 elements).  The k-neighbour weighted sums (interpolation forward,
// the per-query half of subtraction backward) run on the segmented gather-sum of segsum.hip: one group of lanes per
// output row, 16-byte loads, the (idx, weight) pairs of the row fetched once per group.  The scatter halves have two
// forms: the reference-ABI entry points below keep one atomic per element (no workspace in that ABI); the planned
// form (pcm_scatter_plan_hip + pcm_segment_sum_hip, what pointops/*.py calls) inverts idx once and sums each
// destination row without atomics.
#include "pcm_common.hpp"

namespace {

constexpr int kBlock = 256;

inline int grid_for(long total)
{
    long blocks = (total + kBlock - 1) / kBlock;
    const long cap = 256L * 32;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

#define PCM_GRID_STRIDE(e, total) \
    for (long e = (long)blockIdx.x * kBlock + threadIdx.x; e < (total); e += (long)gridDim.x * kBlock)

// ---- K6 ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void interp_bwd(long total, int c, int k, const float *__restrict__ grad_output,
                                                      const int *__restrict__ idx, const float *__restrict__ weight,
                                                      float *__restrict__ grad_input)
{
    PCM_GRID_STRIDE(e, total)
    {
        const int c_idx = (int)(e % c);
        const long n_idx = e / c;
        const float g = grad_output[e];
        for (int i = 0; i < k; ++i) {
            const long ii = n_idx * k + i;
            unsafeAtomicAdd(grad_input + (long)idx[ii] * c + c_idx, g * weight[ii]);
        }
    }
}

// ---- K7 ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void sub_fwd(long total, int nsample, int c, const float *__restrict__ input1,
                                                   const float *__restrict__ input2, const int *__restrict__ idx,
                                                   float *__restrict__ output)
{
    PCM_GRID_STRIDE(e, total)
    {
        const int c_idx = (int)(e % c);
        const long row = e / c;  // n_idx * nsample + nsample_idx
        const long n_idx = row / nsample;
        output[e] = input1[n_idx * c + c_idx] - input2[(long)idx[row] * c + c_idx];
    }
}

// grad_input2 scatter of the reference ABI; grad_input1 (a sum over the nsample rows of one query) is a segmented sum.
__global__ __launch_bounds__(kBlock) void sub_bwd_scatter(long total, int c, const int *__restrict__ idx,
                                                           const float *__restrict__ grad_output, float *__restrict__ grad_input2)
{
    PCM_GRID_STRIDE(e, total)
    {
        const int c_idx = (int)(e % c);
        const long row = e / c;
        unsafeAtomicAdd(grad_input2 + (long)idx[row] * c + c_idx, -grad_output[e]);
    }
}

// ---- K8 ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void agg_fwd(long total, int nsample, int c, int w_c, const float *__restrict__ input,
                                                   const float *__restrict__ position, const float *__restrict__ weight,
                                                   const int *__restrict__ idx, float *__restrict__ output)
{
    PCM_GRID_STRIDE(e, total)
    {
        const int c_idx = (int)(e % c);
        const long n_idx = e / c;
        const int w_c_idx = c_idx % w_c;
        float acc = output[e];
        for (int s = 0; s < nsample; ++s) {
            const long ii = n_idx * nsample + s;
            const float in = input[(long)idx[ii] * c + c_idx];
            const float pos = position[ii * c + c_idx];
            const float w = weight[ii * w_c + w_c_idx];
            const float sum = in + pos;
            acc = acc + sum * w;
        }
        output[e] = acc;
    }
}

// float4 form of agg_fwd for c % 4 == 0 and w_c % 4 == 0: same per-element arithmetic, a quarter of the loads.
__global__ __launch_bounds__(kBlock) void agg_fwd4(long total4, int nsample, int c, int w_c, const float *__restrict__ input,
                                                    const float *__restrict__ position, const float *__restrict__ weight,
                                                    const int *__restrict__ idx, float *__restrict__ output)
{
    const int c4 = c >> 2;
    PCM_GRID_STRIDE(e4, total4)
    {
        const int c_idx = (int)(e4 % c4) << 2;
        const long n_idx = e4 / c4;
        const int w_c_idx = c_idx % w_c;
        float4 acc = *reinterpret_cast<const float4 *>(output + n_idx * c + c_idx);
        for (int s = 0; s < nsample; ++s) {
            const long ii = n_idx * nsample + s;
            const float4 in = *reinterpret_cast<const float4 *>(input + (long)idx[ii] * c + c_idx);
            const float4 pos = *reinterpret_cast<const float4 *>(position + ii * c + c_idx);
            const float4 w = *reinterpret_cast<const float4 *>(weight + ii * w_c + w_c_idx);
            acc.x = acc.x + (in.x + pos.x) * w.x;
            acc.y = acc.y + (in.y + pos.y) * w.y;
            acc.z = acc.z + (in.z + pos.z) * w.z;
            acc.w = acc.w + (in.w + pos.w) * w.w;
        }
        *reinterpret_cast<float4 *>(output + n_idx * c + c_idx) = acc;
    }
}

// The two gradients of aggregation that need no scatter: grad_position[ii, c] = g[n, c] * w[ii, c % w_c] and
// grad_weight[ii, wc] = sum over the c / w_c channels with c % w_c == wc of g[n, c] * (in[idx[ii], c] + pos[ii, c]).
// One thread per (ii, wc) walks its channels in ascending order: coalesced along wc, a fixed order, no atomics
// (the reference issues one atomicAdd per (ii, c): aggregation_cuda_kernel.cu:44).
__global__ __launch_bounds__(kBlock) void agg_bwd_local(long total, int nsample, int c, int w_c, const float *__restrict__ input,
                                                         const float *__restrict__ position, const float *__restrict__ weight,
                                                         const int *__restrict__ idx, const float *__restrict__ grad_output,
                                                         float *__restrict__ grad_position, float *__restrict__ grad_weight,
                                                         int accumulate_weight)
{
    PCM_GRID_STRIDE(e, total)
    {
        const int wc = (int)(e % w_c);
        const long ii = e / w_c;
        const long n_idx = ii / nsample;
        const float w = weight[e];
        const long src = (long)idx[ii] * c;
        float acc = accumulate_weight ? grad_weight[e] : 0.f;
        for (int c_idx = wc; c_idx < c; c_idx += w_c) {
            const float g = grad_output[n_idx * c + c_idx];
            grad_position[ii * c + c_idx] = g * w;
            const float sum = input[src + c_idx] + position[ii * c + c_idx];
            acc = acc + g * sum;
        }
        grad_weight[e] = acc;
    }
}

// grad_input scatter of the reference ABI (one atomic per (ii, c)); the planned form is pcm_segment_sum_hip, scale mode 2.
__global__ __launch_bounds__(kBlock) void agg_bwd_scatter(long total, int nsample, int c, int w_c, const float *__restrict__ weight,
                                                           const int *__restrict__ idx, const float *__restrict__ grad_output,
                                                           float *__restrict__ grad_input)
{
    PCM_GRID_STRIDE(e, total)
    {
        const int c_idx = (int)(e % c);
        const long n_idx = e / c;
        const int w_c_idx = c_idx % w_c;
        const float g = grad_output[e];
        for (int s = 0; s < nsample; ++s) {
            const long ii = n_idx * nsample + s;
            unsafeAtomicAdd(grad_input + (long)idx[ii] * c + c_idx, g * weight[ii * w_c + w_c_idx]);
        }
    }
}

// ---- K9 ---------------------------------------------------------------------------------------
// relation forward: out[r,g] = sum_c q[t_r,g,c] * k[f_r,g,c] * w[c].  The reference issues one
// atomicAdd per (r,g,c); here one thread owns (r,g) and sums c ascending -- same value set, a
// fixed order, no atomics.
__global__ __launch_bounds__(kBlock) void attn_rel_fwd(long total, int g, int c, const float *__restrict__ query,
                                                        const float *__restrict__ key, const float *__restrict__ weight,
                                                        const int *__restrict__ it, const int *__restrict__ ir,
                                                        float *__restrict__ output)
{
    PCM_GRID_STRIDE(e, total)
    {
        const int g_idx = (int)(e % g);
        const long r = e / g;
        const float *q = query + ((long)it[r] * g + g_idx) * c;
        const float *k = key + ((long)ir[r] * g + g_idx) * c;
        float acc = output[e];
        for (int ci = 0; ci < c; ++ci) {
            float v = q[ci] * k[ci];
            v = v * weight[ci];
            acc = acc + v;
        }
        output[e] = acc;
    }
}

__global__ __launch_bounds__(kBlock) void attn_rel_bwd(long total, int g, int c, const float *__restrict__ query,
                                                        float *__restrict__ grad_query, const float *__restrict__ key,
                                                        float *__restrict__ grad_key, const float *__restrict__ weight,
                                                        float *__restrict__ grad_weight, const int *__restrict__ it,
                                                        const int *__restrict__ ir, const float *__restrict__ grad_output)
{
    PCM_GRID_STRIDE(e, total)  // e over (r, g, c), c fastest
    {
        const int c_idx = (int)(e % c);
        const long rg = e / c;
        const int g_idx = (int)(rg % g);
        const long r = rg / g;
        const long q_i = ((long)it[r] * g + g_idx) * c + c_idx;
        const long k_i = ((long)ir[r] * g + g_idx) * c + c_idx;
        const float grad_r = grad_output[rg];
        float a = grad_r * key[k_i];
        a = a * weight[c_idx];
        unsafeAtomicAdd(grad_query + q_i, a);
        a = grad_r * query[q_i];
        a = a * weight[c_idx];
        unsafeAtomicAdd(grad_key + k_i, a);
        a = grad_r * key[k_i];
        a = a * query[q_i];
        unsafeAtomicAdd(grad_weight + c_idx, a);
    }
}

__global__ __launch_bounds__(kBlock) void attn_fus_fwd(long total, int g, int c, const float *__restrict__ weight,
                                                        const float *__restrict__ value, const int *__restrict__ it,
                                                        const int *__restrict__ ir, float *__restrict__ output)
{
    PCM_GRID_STRIDE(e, total)
    {
        const int c_idx = (int)(e % c);
        const long rg = e / c;
        const int g_idx = (int)(rg % g);
        const long r = rg / g;
        const long o_i = ((long)it[r] * g + g_idx) * c + c_idx;
        const long v_i = ((long)ir[r] * g + g_idx) * c + c_idx;
        unsafeAtomicAdd(output + o_i, weight[rg] * value[v_i]);
    }
}

__global__ __launch_bounds__(kBlock) void attn_fus_bwd(long total, int g, int c, const float *__restrict__ weight,
                                                        float *__restrict__ grad_weight, const float *__restrict__ value,
                                                        float *__restrict__ grad_value, const int *__restrict__ it,
                                                        const int *__restrict__ ir, const float *__restrict__ grad_output)
{
    PCM_GRID_STRIDE(e, total)
    {
        const int c_idx = (int)(e % c);
        const long rg = e / c;
        const int g_idx = (int)(rg % g);
        const long r = rg / g;
        const long o_i = ((long)it[r] * g + g_idx) * c + c_idx;
        const long v_i = ((long)ir[r] * g + g_idx) * c + c_idx;
        const float grad = grad_output[o_i];
        unsafeAtomicAdd(grad_weight + rg, grad * value[v_i]);
        unsafeAtomicAdd(grad_value + v_i, grad * weight[rg]);
    }
}

}  // namespace

#define PCM_ST ((hipStream_t)stream)

extern "C" int pcm_interpolation_forward_hip(int n, int c, int k, const float *input, const int *idx,
                                             const float *weight, float *output, void *stream)
{
    if (n < 0 || c < 0 || k < 0) return PCM_ERR_BAD_ARG;
    const long total = (long)n * c;
    if (total == 0) return PCM_OK;
    // output[n, :] = sum_i input[idx[n, i], :] * weight[n, i], i ascending (written, not accumulated)
    return pcm_segment_sum_hip(n, c, nullptr, k, nullptr, idx, 1, weight, 1, 1, 1.f, input, c, 0, output, stream);
}

extern "C" int pcm_interpolation_backward_hip(int n, int c, int k, const float *grad_output, const int *idx,
                                              const float *weight, float *grad_input, void *stream)
{
    if (n < 0 || c < 0 || k < 0) return PCM_ERR_BAD_ARG;
    const long total = (long)n * c;
    if (total == 0) return PCM_OK;
    hipLaunchKernelGGL(interp_bwd, dim3(grid_for(total)), dim3(kBlock), 0, PCM_ST, total, c, k, grad_output, idx, weight, grad_input);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_subtraction_forward_hip(int n, int nsample, int c, const float *input1, const float *input2,
                                           const int *idx, float *output, void *stream)
{
    if (n < 0 || c < 0 || nsample < 0) return PCM_ERR_BAD_ARG;
    const long total = (long)n * nsample * c;
    if (total == 0) return PCM_OK;
    hipLaunchKernelGGL(sub_fwd, dim3(grid_for(total)), dim3(kBlock), 0, PCM_ST, total, nsample, c, input1, input2, idx, output);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_subtraction_backward_hip(int n, int nsample, int c, const int *idx, const float *grad_output,
                                            float *grad_input1, float *grad_input2, void *stream)
{
    if (n < 0 || c < 0 || nsample < 0) return PCM_ERR_BAD_ARG;
    const long total = (long)n * nsample * c;
    if (total == 0) return PCM_OK;
    const int rc = pcm_segment_sum_hip(n, c, nullptr, nsample, nullptr, nullptr, 1, nullptr, 0, 1, 1.f, grad_output, c, 0, grad_input1, stream);
    if (rc != PCM_OK || grad_input2 == nullptr) return rc;  // grad_input2 == nullptr: the caller scatters through a plan
    hipLaunchKernelGGL(sub_bwd_scatter, dim3(grid_for(total)), dim3(kBlock), 0, PCM_ST, total, c, idx, grad_output, grad_input2);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_aggregation_forward_hip(int n, int nsample, int c, int w_c, const float *input, const float *position,
                                           const float *weight, const int *idx, float *output, void *stream)
{
    if (n < 0 || c < 0 || nsample < 0 || w_c < 1) return PCM_ERR_BAD_ARG;
    const long total = (long)n * c;
    if (total == 0) return PCM_OK;
    const bool vec4 = c % 4 == 0 && w_c % 4 == 0 &&
                      (((uintptr_t)input | (uintptr_t)position | (uintptr_t)weight | (uintptr_t)output) % 16 == 0);
    if (vec4)
        hipLaunchKernelGGL(agg_fwd4, dim3(grid_for(total / 4)), dim3(kBlock), 0, PCM_ST, total / 4, nsample, c, w_c, input, position,
                           weight, idx, output);
    else
        hipLaunchKernelGGL(agg_fwd, dim3(grid_for(total)), dim3(kBlock), 0, PCM_ST, total, nsample, c, w_c, input, position, weight,
                           idx, output);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_aggregation_backward_hip(int n, int nsample, int c, int w_c, const float *input, const float *position,
                                            const float *weight, const int *idx, const float *grad_output,
                                            float *grad_input, float *grad_position, float *grad_weight, void *stream)
{
    if (n < 0 || c < 0 || nsample < 0 || w_c < 1) return PCM_ERR_BAD_ARG;
    const long total = (long)n * c;
    if (total == 0) return PCM_OK;
    // grad_weight is accumulated into (the reference adds into the caller's zeroed buffer), grad_position written
    const long locals = (long)n * nsample * w_c;
    if (locals > 0)
        hipLaunchKernelGGL(agg_bwd_local, dim3(grid_for(locals)), dim3(kBlock), 0, PCM_ST, locals, nsample, c, w_c, input, position,
                           weight, idx, grad_output, grad_position, grad_weight, 1);
    if (grad_input != nullptr && nsample > 0)  // nullptr: the caller scatters through a plan (pcm_segment_sum_hip, scale mode 2)
        hipLaunchKernelGGL(agg_bwd_scatter, dim3(grid_for(total)), dim3(kBlock), 0, PCM_ST, total, nsample, c, w_c, weight, idx,
                           grad_output, grad_input);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_attention_relation_step_forward_hip(int m, int g, int c, const float *query, const float *key,
                                                       const float *weight, const int *index_target,
                                                       const int *index_refer, float *output, void *stream)
{
    if (m < 0 || g < 0 || c < 0) return PCM_ERR_BAD_ARG;
    const long total = (long)m * g;
    if (total == 0) return PCM_OK;
    hipLaunchKernelGGL(attn_rel_fwd, dim3(grid_for(total)), dim3(kBlock), 0, PCM_ST, total, g, c, query, key, weight, index_target,
                       index_refer, output);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_attention_relation_step_backward_hip(int m, int g, int c, const float *query, float *grad_query,
                                                        const float *key, float *grad_key, const float *weight,
                                                        float *grad_weight, const int *index_target,
                                                        const int *index_refer, const float *grad_output, void *stream)
{
    if (m < 0 || g < 0 || c < 0) return PCM_ERR_BAD_ARG;
    const long total = (long)m * g * c;
    if (total == 0) return PCM_OK;
    hipLaunchKernelGGL(attn_rel_bwd, dim3(grid_for(total)), dim3(kBlock), 0, PCM_ST, total, g, c, query, grad_query, key, grad_key,
                       weight, grad_weight, index_target, index_refer, grad_output);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_attention_fusion_step_forward_hip(int m, int g, int c, const float *weight, const float *value,
                                                     const int *index_target, const int *index_refer, float *output,
                                                     void *stream)
{
    if (m < 0 || g < 0 || c < 0) return PCM_ERR_BAD_ARG;
    const long total = (long)m * g * c;
    if (total == 0) return PCM_OK;
    hipLaunchKernelGGL(attn_fus_fwd, dim3(grid_for(total)), dim3(kBlock), 0, PCM_ST, total, g, c, weight, value, index_target,
                       index_refer, output);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_attention_fusion_step_backward_hip(int m, int g, int c, const float *weight, float *grad_weight,
                                                      const float *value, float *grad_value, const int *index_target,
                                                      const int *index_refer, const float *grad_output, void *stream)
{
    if (m < 0 || g < 0 || c < 0) return PCM_ERR_BAD_ARG;
    const long total = (long)m * g * c;
    if (total == 0) return PCM_OK;
    hipLaunchKernelGGL(attn_fus_bwd, dim3(grid_for(total)), dim3(kBlock), 0, PCM_ST, total, g, c, weight, grad_weight, value,
                       grad_value, index_target, index_refer, grad_output);
    return PCM_LAUNCH_STATUS();
}

extern "C" const char *pcm_version(void) { return "pcm_pointops 0.1 gfx950 fp-contract=off"; }
