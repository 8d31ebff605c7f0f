// HIP-graph memset nodes replay a wrong byte value from the second launch on (ROCm 7.2 / gfx950: tools/diagnostics/memset_node_probe.py),
// which breaks every library kernel that relies on a memset inside a captured region -- torch's multi-block reductions zero their
// semaphores that way.  p4c_graph_replace_memsets rewrites a captured, not yet instantiated graph: every 1-D memset node
// becomes a kernel node (a fill kernel with the same destination, value and dependencies).  trainer.GraphedTrainingStep calls it between
// capture_end and instantiate (torch.cuda.CUDAGraph(keep_graph=True)).
#include "common.hpp"

#include <vector>

namespace p4c {
namespace {

// element size 1, 2 or 4 bytes; `count` elements of `value` (low bits)
__global__ void graph_fill_kernel(void* dst, unsigned int value, int elem, long long count) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
        if (elem == 4) reinterpret_cast<unsigned int*>(dst)[i] = value;
        else if (elem == 2) reinterpret_cast<unsigned short*>(dst)[i] = (unsigned short)value;
        else reinterpret_cast<unsigned char*>(dst)[i] = (unsigned char)value;
    }
}

}  // namespace
}  // namespace p4c

using namespace p4c;

extern "C" int p4c_graph_replace_memsets(void* graph_handle, int* replaced, int* left) {
    P4C_CHECK_ARG(graph_handle && replaced && left, "p4c_graph_replace_memsets: NULL argument");
    hipGraph_t graph = reinterpret_cast<hipGraph_t>(graph_handle);
    *replaced = *left = 0;
    size_t n = 0;
    P4C_CHECK_HIP(hipGraphGetNodes(graph, nullptr, &n));
    std::vector<hipGraphNode_t> nodes(n);
    if (n) P4C_CHECK_HIP(hipGraphGetNodes(graph, nodes.data(), &n));
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType type;
        P4C_CHECK_HIP(hipGraphNodeGetType(nodes[i], &type));
        if (type != hipGraphNodeTypeMemset) continue;
        hipMemsetParams mp;
        P4C_CHECK_HIP(hipGraphMemsetNodeGetParams(nodes[i], &mp));
        if (mp.height > 1 || (mp.elementSize != 1 && mp.elementSize != 2 && mp.elementSize != 4) || mp.width == 0) {
            ++*left;   // 2-D memsets are left alone (none seen from torch)
            continue;
        }
        size_t nd = 0, no = 0;
        P4C_CHECK_HIP(hipGraphNodeGetDependencies(nodes[i], nullptr, &nd));
        std::vector<hipGraphNode_t> deps(nd);
        if (nd) P4C_CHECK_HIP(hipGraphNodeGetDependencies(nodes[i], deps.data(), &nd));
        P4C_CHECK_HIP(hipGraphNodeGetDependentNodes(nodes[i], nullptr, &no));
        std::vector<hipGraphNode_t> outs(no);
        if (no) P4C_CHECK_HIP(hipGraphNodeGetDependentNodes(nodes[i], outs.data(), &no));

        void* dst = mp.dst;
        unsigned int value = mp.value;
        int elem = (int)mp.elementSize;
        long long count = (long long)mp.width;
        void* args[4] = {&dst, &value, &elem, &count};
        hipKernelNodeParams kp;
        memset(&kp, 0, sizeof(kp));
        long long blocks = (count + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        kp.func = reinterpret_cast<void*>(graph_fill_kernel);
        kp.gridDim = dim3((unsigned)blocks);
        kp.blockDim = dim3(256);
        kp.sharedMemBytes = 0;
        kp.kernelParams = args;
        kp.extra = nullptr;
        hipGraphNode_t fill;
        P4C_CHECK_HIP(hipGraphAddKernelNode(&fill, graph, nd ? deps.data() : nullptr, nd, &kp));
        for (size_t k = 0; k < no; ++k) P4C_CHECK_HIP(hipGraphAddDependencies(graph, &fill, &outs[k], 1));
        P4C_CHECK_HIP(hipGraphDestroyNode(nodes[i]));
        ++*replaced;
    }
    return P4C_OK;
}
