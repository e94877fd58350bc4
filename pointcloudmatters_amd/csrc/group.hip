// group.hip -- neighbourhood gather / scatter kernels for gfx950 (MI355X).  HBM-bound.
//
// Replaces grouping_{forward,backward}_cuda_kernel
//   (/root/reference/libs/pointops/src/grouping/grouping_cuda_kernel.cu:5-40)
// and the pure-PyTorch grouping() of functions/grouping.py:35-59 (fused xyz+feat variant).
//
// Layout: input (n,c) row-major, idx (m,nsample), output (m,nsample,c).  Algorithmic bytes per
// launch: forward  read 4*R (idx) + gathered rows (<= R*c*4, mostly L2 hits since every source
// row is reused ~nsample*m/n times), write R*c*4 with R = m*nsample;  backward mirrors it.
// One 64-lane wave walks one output row with 16-byte accesses when c % 4 == 0 (row-contiguous,
// fully coalesced stores); the row index is wave-uniform so idx is one scalar load per row.
#include "pcm_common.hpp"

namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / PCM_WAVE;

__device__ __forceinline__ long wave_id() { return (long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); }
__device__ __forceinline__ long wave_count() { return (long)gridDim.x * kWavesPerBlock; }

inline int grid_for_rows(long rows)
{
    long blocks = (rows + kWavesPerBlock - 1) / kWavesPerBlock;
    const long cap = 256L * 16;  // 16 workgroups per CU, grid-stride beyond that
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

template <bool VEC4>
__global__ __launch_bounds__(kBlock) void pcm_grouping_fwd_kernel(long rows, int c, const float *__restrict__ input,
                                                                  const int *__restrict__ idx, float *__restrict__ output)
{
    const int lane = threadIdx.x & 63;
    const PcmXcdSplit sp = pcm_xcd_split(rows, kWavesPerBlock);  // see pcm_common.hpp
    for (long r = sp.first + (threadIdx.x >> 6); r < sp.hi; r += sp.step) {
        const long src = (long)idx[r];
        if (VEC4) {
            const float4 *in4 = reinterpret_cast<const float4 *>(input + src * c);
            float4 *out4 = reinterpret_cast<float4 *>(output + r * c);
            for (int v = lane; v < (c >> 2); v += 64) out4[v] = in4[v];
        } else {
            const float *in = input + src * c;
            float *out = output + r * c;
            for (int v = lane; v < c; v += 64) out[v] = in[v];
        }
    }
}

__global__ __launch_bounds__(kBlock) void pcm_grouping_bwd_kernel(long rows, int c, const float *__restrict__ grad_output,
                                                                  const int *__restrict__ idx, float *__restrict__ grad_input)
{
    const int lane = threadIdx.x & 63;
    for (long r = wave_id(); r < rows; r += wave_count()) {
        const long dst = (long)idx[r];
        const float *g = grad_output + r * c;
        float *o = grad_input + dst * c;
        for (int v = lane; v < c; v += 64) unsafeAtomicAdd(o + v, g[v]);  // global_atomic_add_f32, like :24
    }
}

// out[r, 0:3] = (xyz[idx[r]] - new_xyz[r / nsample]) * (idx[r] != -1);  out[r, 3:] = feat[idx[r]] or 0
// xc = 3 (with_xyz) or 0 (features only: xyz / new_xyz may be null).
__global__ __launch_bounds__(kBlock) void pcm_group_xyz_feat_fwd_kernel(long rows, int nsample, int c, int xc,
                                                                        const float *__restrict__ xyz,
                                                                        const float *__restrict__ new_xyz,
                                                                        const float *__restrict__ feat,
                                                                        const int *__restrict__ idx, float *__restrict__ output)
{
    const int lane = threadIdx.x & 63;
    const int w = c + xc;
    // XCD x gathers for the x-th eighth of the output rows = whole clouds: a feature row is fetched by one L2, not by eight
    const PcmXcdSplit sp = pcm_xcd_split(rows, kWavesPerBlock);
    for (long r = sp.first + (threadIdx.x >> 6); r < sp.hi; r += sp.step) {
        const int src = idx[r];
        const long q = r / nsample;
        float *out = output + r * w;
        if (src >= 0) {
            const float *f = feat + (long)src * c;
            if (lane < xc) out[lane] = xyz[(long)src * 3 + lane] - new_xyz[q * 3 + lane];
            for (int v = lane; v < c; v += 64) out[xc + v] = f[v];
        } else {
            // appended zero row (grouping.py:40-41) and the sign(idx+1) mask (:49-56):
            // (0 - new_xyz) * 0 keeps the reference's signed zero; feature columns are +0
            for (int v = lane; v < w; v += 64) out[v] = v < xc ? (0.f - new_xyz[q * 3 + v]) * 0.f : 0.f;
        }
    }
}

__global__ __launch_bounds__(kBlock) void pcm_group_xyz_feat_bwd_kernel(long rows, int c, int xc, const float *__restrict__ grad_output,
                                                                        const int *__restrict__ idx, float *__restrict__ grad_feat)
{
    const int lane = threadIdx.x & 63;
    const int w = c + xc;
    for (long r = wave_id(); r < rows; r += wave_count()) {
        const int dst = idx[r];
        if (dst < 0) continue;
        const float *g = grad_output + r * w + xc;
        float *o = grad_feat + (long)dst * c;
        for (int v = lane; v < c; v += 64) unsafeAtomicAdd(o + v, g[v]);
    }
}

}  // namespace

extern "C" int pcm_grouping_forward_hip(int m, int nsample, int c, const float *input, const int *idx,
                                        float *output, void *stream)
{
    if (m < 0 || nsample < 0 || c < 0) return PCM_ERR_BAD_ARG;
    const long rows = (long)m * nsample;
    if (rows == 0 || c == 0) return PCM_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool vec4 = (c % 4 == 0) && (((uintptr_t)input | (uintptr_t)output) % 16 == 0);
    if (vec4)
        hipLaunchKernelGGL(pcm_grouping_fwd_kernel<true>, dim3(pcm_xcd_grid(grid_for_rows(rows))), dim3(kBlock), 0, st, rows, c, input, idx, output);
    else
        hipLaunchKernelGGL(pcm_grouping_fwd_kernel<false>, dim3(pcm_xcd_grid(grid_for_rows(rows))), dim3(kBlock), 0, st, rows, c, input, idx, output);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_grouping_backward_hip(int m, int nsample, int c, const float *grad_output, const int *idx,
                                         float *grad_input, void *stream)
{
    if (m < 0 || nsample < 0 || c < 0) return PCM_ERR_BAD_ARG;
    const long rows = (long)m * nsample;
    if (rows == 0 || c == 0) return PCM_OK;
    hipLaunchKernelGGL(pcm_grouping_bwd_kernel, dim3(grid_for_rows(rows)), dim3(kBlock), 0, (hipStream_t)stream, rows, c,
                       grad_output, idx, grad_input);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_group_xyz_feat_forward_hip(int m, int nsample, int c, const float *xyz, const float *new_xyz,
                                              const float *feat, const int *idx, float *output, void *stream)
{
    if (m < 0 || nsample < 1 || c < 0) return m == 0 ? PCM_OK : PCM_ERR_BAD_ARG;
    const long rows = (long)m * nsample;
    if (rows == 0) return PCM_OK;
    const int xc = (xyz != nullptr && new_xyz != nullptr) ? 3 : 0;
    hipLaunchKernelGGL(pcm_group_xyz_feat_fwd_kernel, dim3(pcm_xcd_grid(grid_for_rows(rows))), dim3(kBlock), 0, (hipStream_t)stream, rows,
                       nsample, c, xc, xyz, new_xyz, feat, idx, output);
    return PCM_LAUNCH_STATUS();
}

extern "C" int pcm_group_xyz_feat_backward_hip(int m, int nsample, int c, int with_xyz, const float *grad_output,
                                               const int *idx, float *grad_feat, void *stream)
{
    if (m < 0 || nsample < 1 || c < 0) return m == 0 ? PCM_OK : PCM_ERR_BAD_ARG;
    const long rows = (long)m * nsample;
    if (rows == 0 || c == 0) return PCM_OK;
    hipLaunchKernelGGL(pcm_group_xyz_feat_bwd_kernel, dim3(grid_for_rows(rows)), dim3(kBlock), 0, (hipStream_t)stream, rows, c,
                       with_xyz ? 3 : 0, grad_output, idx, grad_feat);
    return PCM_LAUNCH_STATUS();
}
