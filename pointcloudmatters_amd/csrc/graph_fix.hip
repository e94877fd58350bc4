// graph_fix.hip -- hipGraph surgery: replace every memset node of a captured graph by a fill-kernel node.
//
// Why: on ROCm 7.2 / gfx950 a MEMSET node created by stream capture (hipMemsetAsync / hipMemsetD32Async) fills
// correctly on the first launch of the instantiated graph and writes a stale 16-byte host pattern on later launches
// (measured on MI355X, tools/dbg/graph_memset2.py: replay 0 -> 0, replay >= 1 -> two pointer-sized garbage words,
// for 4 B ... 4 KiB fills).  PyTorch's column-sum / full reductions zero their inter-block semaphores with exactly such
// a captured memset (ATen/native/cuda/Reduce.cuh); once the semaphore holds garbage no block is ever "the last one"
// and the reduction's output silently keeps stale memory -- bias gradients of a replayed training step froze or
// turned NaN after a few dozen steps.  Kernel nodes replay correctly, so the step graphs are patched between
// capture and instantiation: same destination, value, extent and dependencies, but executed as a kernel.
#include "pcm_common.hpp"

#include <vector>

namespace {

// 2-D fill of `width` elements of `esize` bytes per row, `height` rows `pitch` bytes apart.
__global__ __launch_bounds__(256) void pcm_graph_fill_kernel(unsigned char *dst, unsigned int value, unsigned int esize,
                                                             size_t width, size_t height, size_t pitch)
{
    const size_t total = width * height;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / width, col = e % width;
        unsigned char *p = dst + row * pitch + col * esize;
        if (esize == 4)
            *reinterpret_cast<unsigned int *>(p) = value;
        else if (esize == 2)
            *reinterpret_cast<unsigned short *>(p) = (unsigned short)value;
        else
            *p = (unsigned char)value;
    }
}

}  // namespace

// Returns PCM_OK and the number of replaced nodes in *n_replaced (may be NULL).  `graph` is a hipGraph_t that has not
// been instantiated yet (torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()).
extern "C" int pcm_graph_replace_memsets(void *graph_, int *n_replaced)
{
    hipGraph_t graph = (hipGraph_t)graph_;
    if (n_replaced) *n_replaced = 0;
    if (!graph) return PCM_ERR_BAD_ARG;
    size_t n = 0;
    hipError_t e = hipGraphGetNodes(graph, nullptr, &n);
    if (e != hipSuccess) return pcm_status(e);
    std::vector<hipGraphNode_t> nodes(n);
    if (n) {
        e = hipGraphGetNodes(graph, nodes.data(), &n);
        if (e != hipSuccess) return pcm_status(e);
    }
    int replaced = 0;
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType type;
        e = hipGraphNodeGetType(nodes[i], &type);
        if (e != hipSuccess) return pcm_status(e);
        if (type != hipGraphNodeTypeMemset) continue;
        hipMemsetParams mp;
        e = hipGraphMemsetNodeGetParams(nodes[i], &mp);
        if (e != hipSuccess) return pcm_status(e);
        if (mp.elementSize != 1 && mp.elementSize != 2 && mp.elementSize != 4) return PCM_ERR_UNSUPPORTED;
        size_t nd = 0, no = 0;
        e = hipGraphNodeGetDependencies(nodes[i], nullptr, &nd);
        if (e != hipSuccess) return pcm_status(e);
        std::vector<hipGraphNode_t> deps(nd);
        if (nd) {
            e = hipGraphNodeGetDependencies(nodes[i], deps.data(), &nd);
            if (e != hipSuccess) return pcm_status(e);
        }
        e = hipGraphNodeGetDependentNodes(nodes[i], nullptr, &no);
        if (e != hipSuccess) return pcm_status(e);
        std::vector<hipGraphNode_t> outs(no);
        if (no) {
            e = hipGraphNodeGetDependentNodes(nodes[i], outs.data(), &no);
            if (e != hipSuccess) return pcm_status(e);
        }
        unsigned char *dst = (unsigned char *)mp.dst;
        unsigned int value = mp.value, esize = mp.elementSize;
        size_t width = mp.width, height = mp.height ? mp.height : 1, pitch = mp.pitch;
        if (height == 1) pitch = width * esize;
        void *args[] = {&dst, &value, &esize, &width, &height, &pitch};
        hipKernelNodeParams kp = {};
        const size_t total = width * height;
        size_t blocks = (total + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        if (blocks < 1) blocks = 1;
        kp.blockDim = dim3(256);
        kp.gridDim = dim3((unsigned)blocks);
        kp.func = reinterpret_cast<void *>(pcm_graph_fill_kernel);
        kp.kernelParams = args;
        kp.extra = nullptr;
        kp.sharedMemBytes = 0;
        hipGraphNode_t knode;
        e = hipGraphAddKernelNode(&knode, graph, nd ? deps.data() : nullptr, nd, &kp);
        if (e != hipSuccess) return pcm_status(e);
        for (size_t j = 0; j < no; ++j) {
            e = hipGraphAddDependencies(graph, &knode, &outs[j], 1);
            if (e != hipSuccess) return pcm_status(e);
        }
        e = hipGraphDestroyNode(nodes[i]);
        if (e != hipSuccess) return pcm_status(e);
        ++replaced;
    }
    if (n_replaced) *n_replaced = replaced;
    return PCM_OK;
}

// The HIP runtime this library is bound to (HIP_VERSION encoding: major * 10000000 + minor * 100000 + patch).  _graphs.py keeps the
// rewrite unconditionally on runtimes up to the one it was found on (7.2.x) and lets a self-test retire it only on newer ones.
extern "C" int pcm_hip_runtime_version(int *version)
{
    if (!version) return PCM_ERR_BAD_ARG;
    return pcm_status(hipRuntimeGetVersion(version));
}

// hipMemsetAsync (d32 == 0: `count` bytes of the low byte of `value`) / hipMemsetD32Async (d32 != 0: `count` dwords) issued through
// THIS library's runtime binding -- what the self-test of _graphs.memset_fix_needed captures: a second, separately dlopen()ed copy
// of the runtime must never be handed torch's stream handle.
extern "C" int pcm_memset_async(void *dst, int value, long count, int d32, void *stream)
{
    if (count < 0 || (count > 0 && !dst)) return PCM_ERR_BAD_ARG;
    if (count == 0) return PCM_OK;
    return pcm_status(d32 ? hipMemsetD32Async((hipDeviceptr_t)dst, value, (size_t)count, (hipStream_t)stream)
                          : hipMemsetAsync(dst, value, (size_t)count, (hipStream_t)stream));
}
