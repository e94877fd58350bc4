// segsum.hip -- segmented gather-sum for gfx950: the atomic-free form of every "scatter-add by idx"
// in pointops (grouping / interpolation / subtraction / aggregation backward) and, with implicit
// segments, of the k-neighbour weighted sums of their forwards.  HBM-bound.
//
// The reference scatters with one atomicAdd per (row, channel):
//   /root/reference/libs/pointops/src/grouping/grouping_cuda_kernel.cu:24,
//   interpolation/interpolation_cuda_kernel.cu:35-40, subtraction/subtraction_cuda_kernel.cu:36-38,
//   aggregation/aggregation_cuda_kernel.cu:42-46.
// Here the index list is inverted once (pcm_scatter_plan_hip: count -> scan -> fill = a CSR of "which source
// rows land on destination row j") and each destination row is then SUMMED by one group of lanes with 16-byte
// loads and written exactly once: no read-modify-write traffic, no contention on hub rows, and the destination
// needs no zero-fill.  The per-term arithmetic is the reference's (`acc = acc + g * w`, un-contracted fp32);
// only the order of the terms of one destination row differs from run to run (the fill order), exactly as the
// order of the reference's atomics does -- unless the plan is built by pcm_scatter_plan_sorted_hip, whose segments are
// sorted and therefore reproducible.
//
//   segment j  = entries t in [start[j], start[j+1])            (start == nullptr: [j*seglen, (j+1)*seglen))
//   entry id   e = list ? list[t] : t
//   source row s = map ? map[e] : e / rowdiv                     (map[e] < 0: the entry is skipped)
//   term         = src[s*src_stride + src_off + col] * scale     scale: none | scale[e] | scale[e*w_c + col % w_c]
//   dst[j*c + col] = sign * sum of terms
#include "pcm_common.hpp"

namespace {

constexpr int kBlock = 256;
constexpr int kScanTile = 4096;  // entries per scan workgroup: 256 threads x 16

__global__ __launch_bounds__(kBlock) void pcm_plan_zero_kernel(int n, int *__restrict__ a)
{
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) a[i] = 0;
}

__global__ __launch_bounds__(kBlock) void pcm_plan_count_kernel(long rows, int n_dst, const int *__restrict__ idx,
                                                                 int *__restrict__ cnt)
{
    for (long r = (long)blockIdx.x * kBlock + threadIdx.x; r < rows; r += (long)gridDim.x * kBlock) {
        const int j = idx[r];
        if (j >= 0 && j < n_dst) atomicAdd(cnt + j, 1);
    }
}

__device__ __forceinline__ int block_sum(int v, int *sh)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    int t = 0;
    for (int i = 0; i < kBlock / 64; ++i) t += sh[i];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(kBlock) void pcm_plan_tilesum_kernel(int n_dst, const int *__restrict__ cnt, int *__restrict__ tsum)
{
    __shared__ int sh[kBlock / 64];
    const int base = blockIdx.x * kScanTile;
    int v = 0;
    for (int i = threadIdx.x; i < kScanTile; i += kBlock)
        if (base + i < n_dst) v += cnt[base + i];
    const int t = block_sum(v, sh);
    if (threadIdx.x == 0) tsum[blockIdx.x] = t;
}

// start[j] = exclusive prefix of cnt; cursor[j] = start[j] (consumed by the fill pass); start[n_dst] = total.
__global__ __launch_bounds__(kBlock) void pcm_plan_scan_kernel(int n_dst, const int *__restrict__ cnt,
                                                                const int *__restrict__ tsum, int *__restrict__ start,
                                                                int *__restrict__ cursor)
{
    __shared__ int sh[kBlock / 64];
    __shared__ int wsum[kBlock / 64];
    int pre = 0;
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += kBlock) pre += tsum[i];
    int carry = block_sum(pre, sh);
    const int base = blockIdx.x * kScanTile;
    constexpr int PER = kScanTile / kBlock;  // 16 consecutive entries per thread
    const int lo = base + threadIdx.x * PER;
    int loc[PER];
    int s = 0;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        loc[u] = lo + u < n_dst ? cnt[lo + u] : 0;
        s += loc[u];
    }
    // exclusive scan of the per-thread sums across the workgroup
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = s;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int woff = 0;
    for (int i = 0; i < w; ++i) woff += wsum[i];
    int run = carry + woff + inc - s;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        if (lo + u < n_dst) {
            start[lo + u] = run;
            cursor[lo + u] = run;
        }
        run += loc[u];
    }
    if (lo <= n_dst - 1 && n_dst - 1 < lo + PER) start[n_dst] = run;  // the thread owning the last entry
}

__global__ __launch_bounds__(kBlock) void pcm_plan_fill_kernel(long rows, int n_dst, const int *__restrict__ idx,
                                                                int *__restrict__ cursor, int *__restrict__ list)
{
    for (long r = (long)blockIdx.x * kBlock + threadIdx.x; r < rows; r += (long)gridDim.x * kBlock) {
        const int j = idx[r];
        if (j >= 0 && j < n_dst) list[atomicAdd(cursor + j, 1)] = (int)r;
    }
}

// ---- deterministic order: sort the entries of every segment ascending ------------------------------------------
// pcm_plan_fill_kernel places the entries of one destination row in the order its atomics happen to retire.  After
// these two passes list[start[j] .. start[j+1]) is ascending, i.e. the CSR -- and every sum taken over it in list
// order -- is a function of idx alone.  Segments of <= 64 entries (every kNN / ball-query neighbourhood in practice)
// are rank-sorted in the registers of one wave; longer ones by a bitonic network in LDS (one workgroup each, up to
// kSortCap entries); anything longer by one thread's in-place heap sort (correct, slow, never seen outside fuzzing).
constexpr int kSortCap = 16384;  // ints of LDS for one long segment (64 KiB)

__global__ __launch_bounds__(kBlock) void pcm_plan_sort_small_kernel(int n_dst, const int *__restrict__ start, int *__restrict__ list)
{
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * kBlock + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * kBlock) >> 6;
    for (long j = wave; j < n_dst; j += nwaves) {
        const int t0 = start[j], len = start[j + 1] - t0;  // wave-uniform
        if (len < 2 || len > 64) continue;
        const int v = lane < len ? list[t0 + lane] : 0x7FFFFFFF;
        int rank = 0;
        for (int u = 0; u < len; ++u) rank += __shfl(v, u) < v ? 1 : 0;  // entries are distinct row ids
        if (lane < len) list[t0 + rank] = v;
    }
}

__global__ __launch_bounds__(kBlock) void pcm_plan_sort_large_kernel(int n_dst, int per_block, const int *__restrict__ start,
                                                                      int *__restrict__ list)
{
    extern __shared__ int sbuf[];  // kSortCap ints, then a queue of up to kBlock long segments
    __shared__ int queue[kBlock];
    __shared__ int nqueue;
    const int j_begin = blockIdx.x * per_block, j_end = min(n_dst, j_begin + per_block);
    for (int base = j_begin; base < j_end; base += kBlock) {
        if (threadIdx.x == 0) nqueue = 0;
        __syncthreads();
        const int j = base + threadIdx.x;
        if (j < j_end && start[j + 1] - start[j] > 64) queue[atomicAdd(&nqueue, 1)] = j;
        __syncthreads();
        const int nq = nqueue;
        for (int qi = 0; qi < nq; ++qi) {
            const int jj = queue[qi];
            const int t0 = start[jj], len = start[jj + 1] - t0;
            if (len <= kSortCap) {
                int P = 128;
                while (P < len) P <<= 1;
                for (int e = threadIdx.x; e < P; e += kBlock) sbuf[e] = e < len ? list[t0 + e] : 0x7FFFFFFF;
                __syncthreads();
                for (int k = 2; k <= P; k <<= 1)
                    for (int s = k >> 1; s > 0; s >>= 1) {
                        for (int e = threadIdx.x; e < P; e += kBlock) {
                            const int o = e ^ s;
                            if (o > e) {
                                const int a = sbuf[e], b = sbuf[o];
                                if ((a > b) == ((e & k) == 0)) sbuf[e] = b, sbuf[o] = a;
                            }
                        }
                        __syncthreads();
                    }
                for (int e = threadIdx.x; e < len; e += kBlock) list[t0 + e] = sbuf[e];
                __syncthreads();
            } else if (threadIdx.x == 0) {  // heap sort in place
                int *a = list + t0;
                auto sift = [&](int root, int end) {
                    for (;;) {
                        int child = 2 * root + 1;
                        if (child >= end) break;
                        if (child + 1 < end && a[child + 1] > a[child]) ++child;
                        if (a[root] >= a[child]) break;
                        const int t = a[root];
                        a[root] = a[child], a[child] = t, root = child;
                    }
                };
                for (int r = len / 2 - 1; r >= 0; --r) sift(r, len);
                for (int e = len - 1; e > 0; --e) {
                    const int t = a[0];
                    a[0] = a[e], a[e] = t;
                    sift(0, e);
                }
            }
        }
        __syncthreads();
    }
}

template <int VEC> struct VecT;
template <> struct VecT<1> { typedef float T; };
template <> struct VecT<4> { typedef float4 T; };

template <int VEC>
__device__ __forceinline__ void acc_term(float (&a)[VEC], const typename VecT<VEC>::T &v, const float (&s)[VEC]);
template <>
__device__ __forceinline__ void acc_term<1>(float (&a)[1], const float &v, const float (&s)[1]) { a[0] = a[0] + v * s[0]; }
template <>
__device__ __forceinline__ void acc_term<4>(float (&a)[4], const float4 &v, const float (&s)[4])
{
    a[0] = a[0] + v.x * s[0];
    a[1] = a[1] + v.y * s[1];
    a[2] = a[2] + v.z * s[2];
    a[3] = a[3] + v.w * s[3];
}

struct SegArgs {
    long n_dst;
    int c, seglen, rowdiv, src_stride, src_off, w_c, lpr_log2;
    const int *start, *list, *map;
    const float *scale, *src;
    float *dst;
    float sign;
};

// SCALE: 0 none, 1 per entry, 2 per (entry, col % w_c).  One group of L = 2^lpr_log2 lanes per destination row.
// The entries of the row are fetched L at a time, one per lane (a coalesced read of list / map / scale), and handed
// round the group with ds_bpermute: the address chain list[t] -> map[e] -> src row is walked once per L entries, not
// once per entry, and four source rows are in flight per lane before the first accumulate.  Terms are added in
// ascending t (for the implicit segments of the forwards: the reference's k-ascending order, bit for bit).
template <int VEC, int SCALE>
__global__ __launch_bounds__(kBlock) void pcm_segment_sum_kernel(SegArgs a)
{
    typedef typename VecT<VEC>::T V;
    constexpr int NC = 2;  // column chunks per pass: c <= 2 * L * VEC needs one pass over the entries
    constexpr int UN = 4;  // entries in flight
    const int lpr = 1 << a.lpr_log2;
    const int sub = threadIdx.x & (lpr - 1);
    // XCD x sums the x-th eighth of the destination rows (pcm_common.hpp): the source rows of a cloud go through one L2
    const PcmXcdSplit sp = pcm_xcd_split(a.n_dst, kBlock >> a.lpr_log2);
    for (long j = sp.first + (threadIdx.x >> a.lpr_log2); j < sp.hi; j += sp.step) {
        long t0, t1;
        if (a.start) {
            t0 = a.start[j];
            t1 = a.start[j + 1];
        } else {
            t0 = j * a.seglen;
            t1 = t0 + a.seglen;
        }
        for (int col0 = 0; col0 < a.c; col0 += NC * lpr * VEC) {
            float acc[NC][VEC];
#pragma unroll
            for (int n = 0; n < NC; ++n)
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[n][v] = 0.f;
            for (long tb = t0; tb < t1; tb += lpr) {
                // one entry per lane
                long my_e = -1, my_s = -1;
                float my_w = 1.f;
                if (tb + sub < t1) {
                    my_e = a.list ? (long)a.list[tb + sub] : tb + sub;
                    my_s = a.map ? (long)a.map[my_e] : (a.rowdiv == 1 ? my_e : my_e / a.rowdiv);
                    if (SCALE == 1) my_w = a.scale[my_e];
                }
                const int nb = (int)(t1 - tb < lpr ? t1 - tb : lpr);
                for (int i = 0; i < nb; i += UN) {
                    long s[UN], e[UN];
                    float w1[UN];
                    V val[UN][NC];
#pragma unroll
                    for (int u = 0; u < UN; ++u) {
                        // lanes past nb carry s = -1: skipped below
                        s[u] = __shfl(my_s, i + u, lpr);
                        if (i + u >= nb) s[u] = -1;
                        if (SCALE == 2) e[u] = __shfl(my_e, i + u, lpr);
                        if (SCALE == 1) w1[u] = __shfl(my_w, i + u, lpr);
#pragma unroll
                        for (int n = 0; n < NC; ++n) {
                            const int col = col0 + (n * lpr + sub) * VEC;
                            val[u][n] = V();
                            if (s[u] >= 0 && col < a.c) val[u][n] = *reinterpret_cast<const V *>(a.src + s[u] * a.src_stride + a.src_off + col);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < UN; ++u) {
                        if (s[u] < 0) continue;
#pragma unroll
                        for (int n = 0; n < NC; ++n) {
                            const int col = col0 + (n * lpr + sub) * VEC;
                            if (col >= a.c) continue;
                            float w[VEC];
#pragma unroll
                            for (int v = 0; v < VEC; ++v)
                                w[v] = SCALE == 0 ? 1.f : SCALE == 1 ? w1[u] : a.scale[e[u] * a.w_c + (col + v) % a.w_c];
                            acc_term<VEC>(acc[n], val[u][n], w);
                        }
                    }
                }
            }
#pragma unroll
            for (int n = 0; n < NC; ++n) {
                const int col = col0 + (n * lpr + sub) * VEC;
                if (col >= a.c) continue;
                float *o = a.dst + j * a.c + col;
                if (VEC == 4) {
                    *reinterpret_cast<float4 *>(o) = make_float4(a.sign * acc[n][0], a.sign * acc[n][VEC > 1 ? 1 : 0],
                                                                  a.sign * acc[n][VEC > 2 ? 2 : 0], a.sign * acc[n][VEC > 3 ? 3 : 0]);
                } else {
                    o[0] = a.sign * acc[n][0];
                }
            }
        }
    }
}

inline int ilog2_ceil(int v)
{
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

inline int blocks_for(long items, long per_block, long cap)
{
    long b = (items + per_block - 1) / per_block;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

static int plan_build(long rows, int n_dst, const int *idx, int *cnt, int *start, int *cursor, int *tsum, int *list, int sorted,
                      hipStream_t st)
{
    const int tiles = (n_dst + kScanTile - 1) / kScanTile;
    if (n_dst == 0) {  // start = { 0 }
        hipLaunchKernelGGL(pcm_plan_zero_kernel, dim3(1), dim3(kBlock), 0, st, 1, start);
        return PCM_LAUNCH_STATUS();
    }
    hipLaunchKernelGGL(pcm_plan_zero_kernel, dim3(blocks_for(n_dst, kBlock, 2048)), dim3(kBlock), 0, st, n_dst, cnt);
    if (rows > 0)
        hipLaunchKernelGGL(pcm_plan_count_kernel, dim3(blocks_for(rows, kBlock, 4096)), dim3(kBlock), 0, st, rows, n_dst, idx, cnt);
    hipLaunchKernelGGL(pcm_plan_tilesum_kernel, dim3(tiles), dim3(kBlock), 0, st, n_dst, cnt, tsum);
    hipLaunchKernelGGL(pcm_plan_scan_kernel, dim3(tiles), dim3(kBlock), 0, st, n_dst, cnt, tsum, start, cursor);
    if (rows > 0)
        hipLaunchKernelGGL(pcm_plan_fill_kernel, dim3(blocks_for(rows, kBlock, 4096)), dim3(kBlock), 0, st, rows, n_dst, idx, cursor, list);
    if (sorted && rows > 1) {
        hipLaunchKernelGGL(pcm_plan_sort_small_kernel, dim3(blocks_for(n_dst, kBlock / 64, 2048)), dim3(kBlock), 0, st, n_dst, start, list);
        const int grid = blocks_for(n_dst, kBlock, 1024);
        const int per_block = (n_dst + grid - 1) / grid;
        const size_t lds = (size_t)kSortCap * sizeof(int);
        const int rc = pcm_status(hipFuncSetAttribute(reinterpret_cast<const void *>(pcm_plan_sort_large_kernel),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (rc) return rc;
        hipLaunchKernelGGL(pcm_plan_sort_large_kernel, dim3(grid), dim3(kBlock), lds, st, n_dst, per_block, start, list);
    }
    return PCM_LAUNCH_STATUS();
}

static bool plan_fits(long rows, int n_dst) { return rows >= 0 && n_dst >= 0 && rows + 3L * n_dst + 4096 < 2147483647L; }

extern "C" long pcm_scatter_plan_ws_ints(long rows, int n_dst)
{
    const long tiles = ((long)n_dst + kScanTile - 1) / kScanTile;
    return 3L * n_dst + 2 + tiles + rows;  // cnt | start (n_dst+1) | cursor | tile sums | list (rows)
}

// ws: pcm_scatter_plan_ws_ints(rows, n_dst) ints.  On return *start_out / *list_out point into ws.
extern "C" int pcm_scatter_plan_hip(long rows, int n_dst, const int *idx, int *ws, const int **start_out,
                                    const int **list_out, void *stream)
{
    if (rows < 0 || n_dst < 0 || (rows > 0 && idx == nullptr) || ws == nullptr) return PCM_ERR_BAD_ARG;
    if (!plan_fits(rows, n_dst)) return PCM_ERR_UNSUPPORTED;  // the scan and the list use 32-bit offsets
    const int tiles = (n_dst + kScanTile - 1) / kScanTile;
    int *cnt = ws, *start = cnt + n_dst, *cursor = start + n_dst + 1, *tsum = cursor + n_dst, *list = tsum + (tiles > 0 ? tiles : 0);
    if (start_out) *start_out = start;
    if (list_out) *list_out = list;
    return plan_build(rows, n_dst, idx, cnt, start, cursor, tsum, list, 0, (hipStream_t)stream);
}

// The same CSR with every segment sorted ascending (a function of idx alone: sums taken in list order are reproducible
// from run to run), written to caller-owned arrays: start (n_dst + 1), list (rows); scratch:
// pcm_scatter_plan_sorted_scratch_ints(n_dst) ints.
extern "C" long pcm_scatter_plan_sorted_scratch_ints(int n_dst)
{
    return 2L * n_dst + ((long)n_dst + kScanTile - 1) / kScanTile + 1;  // cnt | cursor | tile sums
}

extern "C" int pcm_scatter_plan_sorted_hip(long rows, int n_dst, const int *idx, int *scratch, int *start, int *list, void *stream)
{
    if (rows < 0 || n_dst < 0 || (rows > 0 && (idx == nullptr || list == nullptr)) || scratch == nullptr || start == nullptr)
        return PCM_ERR_BAD_ARG;
    if (!plan_fits(rows, n_dst)) return PCM_ERR_UNSUPPORTED;
    int *cnt = scratch, *cursor = cnt + n_dst, *tsum = cursor + n_dst;
    return plan_build(rows, n_dst, idx, cnt, start, cursor, tsum, list, 1, (hipStream_t)stream);
}

extern "C" int pcm_segment_sum_hip(long n_dst, int c, const int *start, int seglen, const int *list, const int *map,
                                   int rowdiv, const float *scale, int scale_mode, int w_c, float sign, const float *src,
                                   int src_stride, int src_off, float *dst, void *stream)
{
    if (n_dst < 0 || c < 0 || (start == nullptr && seglen < 0) || rowdiv < 1 || scale_mode < 0 || scale_mode > 2 ||
        (scale_mode != 0 && scale == nullptr) || (scale_mode == 2 && w_c < 1))
        return PCM_ERR_BAD_ARG;
    if (n_dst == 0 || c == 0) return PCM_OK;
    SegArgs a;
    a.n_dst = n_dst; a.c = c; a.seglen = seglen; a.rowdiv = rowdiv; a.src_stride = src_stride; a.src_off = src_off;
    a.w_c = w_c > 0 ? w_c : 1; a.start = start; a.list = list; a.map = map; a.scale = scale; a.src = src; a.dst = dst; a.sign = sign;
    const bool vec4 = c % 4 == 0 && src_stride % 4 == 0 && src_off % 4 == 0 && (((uintptr_t)src | (uintptr_t)dst) % 16 == 0);
    const int vec = vec4 ? 4 : 1;
    int l2 = ilog2_ceil((c + 2 * vec - 1) / (2 * vec));  // two column chunks per lane
    if (l2 > 6) l2 = 6;
    if (l2 < 2) l2 = 2;
    a.lpr_log2 = l2;
    const long per_block = kBlock >> l2;
    const int grid = pcm_xcd_grid(blocks_for(n_dst, per_block, 256L * 32));
    hipStream_t st = (hipStream_t)stream;
#define PCM_SEG_LAUNCH(V, S) hipLaunchKernelGGL((pcm_segment_sum_kernel<V, S>), dim3(grid), dim3(kBlock), 0, st, a)
    if (vec4) {
        if (scale_mode == 0) PCM_SEG_LAUNCH(4, 0); else if (scale_mode == 1) PCM_SEG_LAUNCH(4, 1); else PCM_SEG_LAUNCH(4, 2);
    } else {
        if (scale_mode == 0) PCM_SEG_LAUNCH(1, 0); else if (scale_mode == 1) PCM_SEG_LAUNCH(1, 1); else PCM_SEG_LAUNCH(1, 2);
    }
#undef PCM_SEG_LAUNCH
    return PCM_LAUNCH_STATUS();
}
