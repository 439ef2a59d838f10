// Library-level entry points of libstylerenderer_hip.so (error strings, ABI version).
#include <vector>

#include "common.h"

extern "C" int sr_abi_version(void) { return 9; }   // 9: sr_rasterize_grad_* take adj_slot; sr_adam_flat_guarded rolls *step back

extern "C" const char* sr_error_string(int code) {
    if (code == SR_OK) return "ok";
    if (code == SR_EINVAL) return "stylerenderer_amd: invalid argument (size, null pointer or unsupported combination)";
    if (code == SR_ERANGE) return "stylerenderer_amd: size exceeds the kernel's index range";
    if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
    return "stylerenderer_amd: unknown error";
}

// ---- in-graph signalling for the overlapped gradient reduction (include/stylerenderer_amd.h) ----------------------
// A bucket of the flat gradient buffer that is complete in the MIDDLE of a replayed phase graph has to release its
// all-reduce on another stream while the rest of the graph keeps running.  The API for that is an event-record node
// (hipEventRecordWithFlags(..., hipEventRecordExternal)); it works on the ROCm 7.2 runtime but returns
// hipErrorInvalidValue on the HIP 7.0.51831 runtime PyTorch-ROCm 2.10 bundles, and hipStreamWaitValue32 does not
// release early on either (scripts/event_graph_probe.cpp, event_graph_probe_torch.py).  Plain kernels do: k_signal_set
// is a node of the graph that publishes the replay's EPOCH (a device word the host writes, in stream order, in front
// of the replay) into the bucket's word once everything captured before it has finished; k_signal_wait is the first
// thing in the communication stream and spins (one lane, s_sleep) until the bucket's word reaches the epoch of the
// replay the host has just launched.  A replay the host did not announce (a bare graph.replay() in a test or probe)
// republishes the epoch already reached: it cannot run ahead of the host's count the way an incrementing counter
// does.  Measured inside a replay: the consumer starts 0.5 us after the producer node.
namespace {

__global__ void k_signal_bump(unsigned* counter) {
    __threadfence();
    atomicAdd(counter, 1u);
}

__global__ void k_signal_set(unsigned* word, const unsigned* epoch) {
    __threadfence();
    __hip_atomic_store(word, __hip_atomic_load(epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// The word lives in pinned HOST memory (the host-released mode of the bucketed reducer polls it from the CPU instead
// of spinning a kernel on the communication stream).  The node runs after everything captured before it (graph
// dependency: those kernels' end-of-kernel release has made the bucket visible device-wide), so the store itself needs
// no fence — a release at system scope would write back this XCD's whole L2 for nothing.
__global__ void k_signal_set_host(unsigned* word, const unsigned* epoch) {
    __hip_atomic_store(word, __hip_atomic_load(epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

// timeout_ticks in 100 MHz wall-clock ticks (0: wait for ever).  On expiry the kernel stores `code` into *status (pinned
// host memory, system scope) and RETURNS: what is queued behind it runs on unfinished data, but the host finds the code
// at its next check and raises instead of the job sitting in a silent device-side spin until an outer limit kills it.
// `poison` (sr_signal_wait_poison): a float of the data the wait guards (the first element of the all-reduce bucket);
// on expiry it is overwritten with NaN BEFORE the kernel returns, so the collective queued behind the wait spreads the
// NaN to every rank and the guarded Adam step (sr_adam_flat_guarded) refuses the update everywhere.
__global__ void k_signal_wait(const unsigned* counter, unsigned at_least, unsigned long long timeout_ticks, int* status,
                              int code, float* poison) {
    const unsigned long long t0 = wall_clock64();
    // signed distance: correct across the 2^32 wrap of a word that only ever grows
    while ((int)(__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - at_least) < 0) {
        __builtin_amdgcn_s_sleep(16);
        if (timeout_ticks && wall_clock64() - t0 > timeout_ticks) {
            if (poison) {
                *poison = __builtin_nanf("");
                __threadfence();
            }
            if (status) __hip_atomic_store(status, code, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
    }
}

}  // namespace

extern "C" int sr_signal_bump(uint32_t* counter, sr_stream_t stream) {
    if (!counter) return SR_EINVAL;
    hipLaunchKernelGGL(k_signal_bump, dim3(1), dim3(1), 0, sr_stream(stream), counter);
    return sr_launch_status();
}

extern "C" int sr_signal_set(uint32_t* word, const uint32_t* epoch, sr_stream_t stream) {
    if (!word || !epoch) return SR_EINVAL;
    hipLaunchKernelGGL(k_signal_set, dim3(1), dim3(1), 0, sr_stream(stream), word, epoch);
    return sr_launch_status();
}

extern "C" int sr_signal_set_host(uint32_t* word_host, const uint32_t* epoch, sr_stream_t stream) {
    if (!word_host || !epoch) return SR_EINVAL;
    hipLaunchKernelGGL(k_signal_set_host, dim3(1), dim3(1), 0, sr_stream(stream), word_host, epoch);
    return sr_launch_status();
}

extern "C" int sr_signal_wait(const uint32_t* counter, uint32_t at_least, sr_stream_t stream) {
    if (!counter) return SR_EINVAL;
    hipLaunchKernelGGL(k_signal_wait, dim3(1), dim3(1), 0, sr_stream(stream), counter, at_least, 0ull, (int*)nullptr, 0,
                       (float*)nullptr);
    return sr_launch_status();
}

extern "C" int sr_signal_wait_timeout(const uint32_t* counter, uint32_t at_least, uint64_t timeout_us,
                                      int32_t* status_host, int32_t code, sr_stream_t stream) {
    if (!counter) return SR_EINVAL;
    hipLaunchKernelGGL(k_signal_wait, dim3(1), dim3(1), 0, sr_stream(stream), counter, at_least,
                       (unsigned long long)timeout_us * 100ull, status_host, code, (float*)nullptr);
    return sr_launch_status();
}

extern "C" int sr_signal_wait_poison(const uint32_t* counter, uint32_t at_least, uint64_t timeout_us,
                                     int32_t* status_host, int32_t code, float* poison, sr_stream_t stream) {
    if (!counter) return SR_EINVAL;
    hipLaunchKernelGGL(k_signal_wait, dim3(1), dim3(1), 0, sr_stream(stream), counter, at_least,
                       (unsigned long long)timeout_us * 100ull, status_host, code, poison);
    return sr_launch_status();
}

// ---- hipGraph repair: memset nodes -> kernel nodes -------------------------------------------------------------
// The HIP runtime bundled with PyTorch 2.10+rocm7.0 (7.0.51831) replays a captured MEMSET node correctly once; from the
// second launch of the executable graph on the node writes a corrupted value (0x10 bytes instead of 0:
// scripts/memset_graph_probe_torch.py; the same program on the ROCm 7.2 runtime is fine).  torch's multi-block
// reductions zero their semaphores with hipMemsetAsync, so every captured `sum` / `mean` that needs more than one
// workgroup per output returned garbage from the second replay on (scripts/graph_reduce_probe.py) — the path-length
// phase and the inversion loss among them.  This entry point rewrites a captured graph BEFORE it is instantiated:
// every memset node becomes a kernel node (k_graph_fill) with the same predecessors and successors.
namespace {

__global__ __launch_bounds__(256) void k_graph_fill(unsigned char* __restrict__ dst, unsigned value, unsigned elem,
                                                    unsigned long long row_bytes, unsigned long long height,
                                                    unsigned long long pitch, int vec16) {
    const unsigned long long stride = (unsigned long long)gridDim.x * 256;
    unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (vec16) {        // one row, 16-byte aligned, a multiple of 16 bytes, value pattern replicated over 32 bits
        const uint4 v = make_uint4(value, value, value, value);
        for (const unsigned long long n16 = row_bytes / 16; i < n16; i += stride) reinterpret_cast<uint4*>(dst)[i] = v;
        return;
    }
    for (const unsigned long long total = row_bytes * height; i < total; i += stride) {
        const unsigned long long row = i / row_bytes, col = i - row * row_bytes;
        dst[row * pitch + col] = (unsigned char)(value >> (8 * (col % elem)));
    }
}

}  // namespace

extern "C" int sr_graph_node_count(void* graph_handle, int* kernel_nodes, int* all_nodes) {
    if (!graph_handle) return SR_EINVAL;
    hipGraph_t graph = static_cast<hipGraph_t>(graph_handle);
    size_t n = 0;
    hipError_t e = hipGraphGetNodes(graph, nullptr, &n);
    if (e != hipSuccess) return static_cast<int>(e);
    std::vector<hipGraphNode_t> nodes(n);
    if (n && (e = hipGraphGetNodes(graph, nodes.data(), &n)) != hipSuccess) return static_cast<int>(e);
    int kernels = 0;
    for (size_t k = 0; k < n; ++k) {
        hipGraphNodeType type;
        if ((e = hipGraphNodeGetType(nodes[k], &type)) != hipSuccess) return static_cast<int>(e);
        kernels += type == hipGraphNodeTypeKernel;
    }
    if (kernel_nodes) *kernel_nodes = kernels;
    if (all_nodes) *all_nodes = (int)n;
    return SR_OK;
}

extern "C" int sr_graph_replace_memset_nodes(void* graph_handle, int* replaced) {
    if (!graph_handle) return SR_EINVAL;
    hipGraph_t graph = static_cast<hipGraph_t>(graph_handle);
    size_t n = 0;
    hipError_t e = hipGraphGetNodes(graph, nullptr, &n);
    if (e != hipSuccess) return static_cast<int>(e);
    std::vector<hipGraphNode_t> nodes(n);
    if (n && (e = hipGraphGetNodes(graph, nodes.data(), &n)) != hipSuccess) return static_cast<int>(e);
    int count = 0;
    for (size_t k = 0; k < n; ++k) {
        hipGraphNodeType type;
        if ((e = hipGraphNodeGetType(nodes[k], &type)) != hipSuccess) return static_cast<int>(e);
        if (type != hipGraphNodeTypeMemset) continue;
        hipMemsetParams mp;
        if ((e = hipGraphMemsetNodeGetParams(nodes[k], &mp)) != hipSuccess) return static_cast<int>(e);
        size_t nd = 0, ns = 0;
        if ((e = hipGraphNodeGetDependencies(nodes[k], nullptr, &nd)) != hipSuccess) return static_cast<int>(e);
        std::vector<hipGraphNode_t> deps(nd);
        if (nd && (e = hipGraphNodeGetDependencies(nodes[k], deps.data(), &nd)) != hipSuccess) return static_cast<int>(e);
        if ((e = hipGraphNodeGetDependentNodes(nodes[k], nullptr, &ns)) != hipSuccess) return static_cast<int>(e);
        std::vector<hipGraphNode_t> succ(ns);
        if (ns && (e = hipGraphNodeGetDependentNodes(nodes[k], succ.data(), &ns)) != hipSuccess) return static_cast<int>(e);
        unsigned char* dst = static_cast<unsigned char*>(mp.dst);
        unsigned elem = mp.elementSize ? mp.elementSize : 1;
        unsigned value = mp.value;
        if (elem == 1) value = (value & 0xFFu) * 0x01010101u;
        else if (elem == 2) value = (value & 0xFFFFu) * 0x00010001u;
        unsigned long long row_bytes = (unsigned long long)mp.width * elem, height = mp.height ? mp.height : 1;
        unsigned long long pitch = mp.pitch ? mp.pitch : row_bytes;
        int vec16 = (height == 1 && row_bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) ? 1 : 0;
        const unsigned long long items = vec16 ? row_bytes / 16 : row_bytes * height;
        void* args[] = {&dst, &value, &elem, &row_bytes, &height, &pitch, &vec16};
        hipKernelNodeParams kp;
        kp.func = reinterpret_cast<void*>(k_graph_fill);
        kp.gridDim = dim3((unsigned)sr_stream_grid((int64_t)(items ? items : 1), 256));
        kp.blockDim = dim3(256);
        kp.sharedMemBytes = 0;
        kp.kernelParams = args;
        kp.extra = nullptr;
        hipGraphNode_t fill;
        if ((e = hipGraphAddKernelNode(&fill, graph, nd ? deps.data() : nullptr, nd, &kp)) != hipSuccess)
            return static_cast<int>(e);
        for (size_t j = 0; j < ns; ++j)
            if ((e = hipGraphAddDependencies(graph, &fill, &succ[j], 1)) != hipSuccess) return static_cast<int>(e);
        if ((e = hipGraphDestroyNode(nodes[k])) != hipSuccess) return static_cast<int>(e);
        ++count;
    }
    if (replaced) *replaced = count;
    return SR_OK;
}
