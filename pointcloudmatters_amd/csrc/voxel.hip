// voxel.hip -- voxel-grid keys for the GPU-side GridSamplePCD (the step before the BC hot path) for gfx950.
//
// /root/reference/src/data/components/transformpcd.py:684-701 does, per cloud and on a CPU worker,
//   scaled = coord / grid_size (float64 under NumPy >= 2);  grid = floor(scaled).astype(int);  grid -= grid.min(0);
//   key = fnv_hash_vec(grid)  (FNV-1a over the three coordinates as uint64, transformpcd.py:776-790)
// before sort / unique / pick.  Here the same arithmetic runs for a whole packed batch (n points, b clouds):
//   pcm_voxel_min  : per-cloud minimum of floor(coord / grid_size)   (wave-reduced, then one atomicMin per wave)
//   pcm_voxel_hash : grid_coord (n,3) int64 relative to that minimum, key (n) = FNV-1a 64 (bit pattern in an int64)
// Sorting / unique / gather are rocPRIM-backed framework ops (pointcloudmatters_amd/bc/gpu_transforms.py).
// Integer / byte work: bit-exact against the reference's own functions (tests/golden/gridsample_ref.npz).
// Bytes per point: 12 read (x2 passes) + 24 (grid_coord) + 8 (key) written.
#include "pcm_common.hpp"

#include <limits.h>

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ int floor_div(float c, double grid)
{
    return (int)floor((double)c / grid);
}

__device__ __forceinline__ int wave_min_i32(int v)
{
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(v, off);
        v = o < v ? o : v;
    }
    return v;
}

__global__ __launch_bounds__(kBlock) void pcm_voxel_min_kernel(int n, int b, const float *__restrict__ coord,
                                                               const int *__restrict__ offset, double grid, int *__restrict__ gmin)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < n;
    const int cloud = live ? pcm_cloud_of(i, offset, b) : -1;
    int g[3] = {INT_MAX, INT_MAX, INT_MAX};
    if (live)
        for (int d = 0; d < 3; ++d) g[d] = floor_div(coord[(size_t)i * 3 + d], grid);
    // the common case: the whole wave lies in one cloud -> one atomic per coordinate per wave
    const int first = __shfl(cloud, 0);
    const bool uniform = __all(cloud == first || !live) && first >= 0;
    if (uniform) {
        for (int d = 0; d < 3; ++d) {
            const int m = wave_min_i32(g[d]);
            if ((threadIdx.x & 63) == 0) atomicMin(&gmin[first * 3 + d], m);
        }
    } else if (live) {
        for (int d = 0; d < 3; ++d) atomicMin(&gmin[cloud * 3 + d], g[d]);
    }
}

__global__ __launch_bounds__(kBlock) void pcm_voxel_hash_kernel(int n, int b, const float *__restrict__ coord,
                                                                const int *__restrict__ offset, double grid,
                                                                const int *__restrict__ gmin, long *__restrict__ grid_coord,
                                                                long *__restrict__ key, int *__restrict__ cloud_out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int cloud = pcm_cloud_of(i, offset, b);
    unsigned long long h = 14695981039346656037ull;
    for (int d = 0; d < 3; ++d) {
        const long g = (long)floor_div(coord[(size_t)i * 3 + d], grid) - (long)gmin[cloud * 3 + d];
        grid_coord[(size_t)i * 3 + d] = g;
        h *= 1099511628211ull;
        h ^= (unsigned long long)g;
    }
    key[i] = (long)h;
    cloud_out[i] = cloud;
}

}  // namespace

extern "C" int pcm_voxel_keys_hip(int n, int b, const float *coord, const int *offset, double grid_size, int *gmin,
                                  long *grid_coord, long *key, int *cloud, void *stream)
{
    if (n < 0 || b < 0 || !(grid_size > 0.0)) return PCM_ERR_BAD_ARG;
    if (n == 0 || b == 0) return PCM_OK;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetD32Async((hipDeviceptr_t)gmin, INT_MAX, (size_t)b * 3, s);
    if (e != hipSuccess) return pcm_status(e);
    const int blocks = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(pcm_voxel_min_kernel, dim3(blocks), dim3(kBlock), 0, s, n, b, coord, offset, grid_size, gmin);
    hipLaunchKernelGGL(pcm_voxel_hash_kernel, dim3(blocks), dim3(kBlock), 0, s, n, b, coord, offset, grid_size, gmin, grid_coord, key,
                       cloud);
    return PCM_LAUNCH_STATUS();
}
