// Probe: does a memset node inside a captured hipGraph execute, in order, on every replay?  (gfx950 / ROCm 7)
//   hipcc --offload-arch=gfx950 -O2 scripts/memset_graph_probe.cpp -o build/msp && build/msp
// graph: k_count (adds 1 to every word of `sem`) -> hipMemsetAsync(sem, 0, bytes) -> k_count -> k_copy(out <- sem).
// Correct result after every replay: out[i] == 1.  A memset that is skipped or reordered leaves 2, 3, 4 ...
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void k_count(int* sem, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(sem + i, 1);
}
__global__ void k_copy(int* out, const int* sem, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = sem[i];
}
__global__ void k_busy(float* x, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float v = x[i];
        for (int k = 0; k < 2000; ++k) v = v * 1.0001f + 0.5f;
        x[i] = v;
    }
}

static void run(size_t bytes, int replays) {
    const int n = (int)(bytes / 4) > 0 ? (int)(bytes / 4) : 1;
    int *sem, *out;
    float* busy;
    hipMalloc(&sem, n * 4 + 64);
    hipMalloc(&out, n * 4 + 64);
    hipMalloc(&busy, 1 << 24);
    hipMemset(sem, 0, n * 4);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipGraph_t g;
    hipGraphExec_t e;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    k_busy<<<(1 << 22) / 256, 256, 0, s>>>(busy, 1 << 22);
    k_count<<<(n + 255) / 256, 256, 0, s>>>(sem, n);
    hipMemsetAsync(sem, 0, bytes, s);
    k_count<<<(n + 255) / 256, 256, 0, s>>>(sem, n);
    k_copy<<<(n + 255) / 256, 256, 0, s>>>(out, sem, n);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
    std::vector<int> h(n);
    printf("memset of %7zu bytes:", bytes);
    for (int r = 0; r < replays; ++r) {
        hipGraphLaunch(e, s);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost);
        int mx = 0, mn = 1 << 30;
        for (int v : h) {
            mx = v > mx ? v : mx;
            mn = v < mn ? v : mn;
        }
        printf("  replay %d: [%d, %d]%s", r, mn, mx, (mn == 1 && mx == 1) ? " ok" : " WRONG");
    }
    printf("\n");
    hipGraphExecDestroy(e);
    hipGraphDestroy(g);
    hipFree(sem);
    hipFree(out);
    hipFree(busy);
}

int main() {
    for (size_t bytes : {4, 8, 12, 16, 64, 256, 1024, 4096, 65536, 1 << 20}) run(bytes, 4);
    return 0;
}
