// Probe: can a second stream start work in the MIDDLE of a replayed hipGraph on gfx950 / ROCm 7?
//   hipcc --offload-arch=gfx950 -O2 scripts/event_graph_probe.cpp -o /tmp/egp && /tmp/egp
// The graph is  A (spin ~2 ms) -> signal -> B (spin ~2 ms).  A consumer kernel C on another stream must start after A
// ended and before B ended.  Three signalling mechanisms are tried:
//   1. external event-record node  (hipEventRecordWithFlags(..., hipEventRecordExternal) during capture)
//      + hipStreamWaitEvent on the consumer stream after hipGraphLaunch;
//   2. an in-graph kernel that bumps a counter + hipStreamWaitValue32 on the consumer stream;
//   3. an in-graph kernel that bumps a counter + a one-lane polling kernel on the consumer stream.
// Prints, per mechanism, the wall-clock stamps (us, relative to A's start) and OVERLAP yes/no.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("  %s -> %s\n", #x, hipGetErrorString(e_));                 \
            return false;                                                      \
        }                                                                      \
    } while (0)

__global__ void k_spin(long long* stamp, long long ticks) {
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) stamp[0] = t0;
    while (wall_clock64() - t0 < ticks) {
    }
    if (threadIdx.x == 0) stamp[1] = wall_clock64();
}
__global__ void k_bump(unsigned* flag) {
    __threadfence();
    atomicAdd(flag, 1u);
}
__global__ void k_poll(const unsigned* flag, unsigned target) {
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) < target) __builtin_amdgcn_s_sleep(8);
}
__global__ void k_stamp(long long* stamp) { stamp[0] = wall_clock64(); }

static bool run(int mech, int replays) {
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    long long* st;
    CK(hipHostMalloc(&st, 8 * sizeof(long long)));
    unsigned* flag = nullptr;
    if (mech == 2) {
        hipError_t e = hipExtMallocWithFlags((void**)&flag, 64, hipMallocSignalMemory);
        if (e != hipSuccess) {
            printf("  hipMallocSignalMemory -> %s; trying plain hipMalloc\n", hipGetErrorString(e));
            CK(hipMalloc(&flag, 64));
        }
    } else {
        CK(hipMalloc(&flag, 64));
    }
    CK(hipMemset(flag, 0, 64));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    int rate_khz = 0;
    CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    const long long ticks = (long long)rate_khz * 2;   // 2 ms

    hipGraph_t graph;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
    k_spin<<<1, 64, 0, s1>>>(st + 0, ticks);
    if (mech == 1) {
        CK(hipEventRecordWithFlags(ev, s1, hipEventRecordExternal));
    } else {
        k_bump<<<1, 1, 0, s1>>>(flag);
    }
    k_spin<<<1, 64, 0, s1>>>(st + 2, ticks);
    CK(hipStreamEndCapture(s1, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));

    bool all = true;
    for (int r = 1; r <= replays; ++r) {
        CK(hipGraphLaunch(exec, s1));
        if (mech == 1) {
            CK(hipStreamWaitEvent(s2, ev, 0));
        } else if (mech == 2) {
            CK(hipStreamWaitValue32(s2, flag, (unsigned)r, hipStreamWaitValueGte, 0xFFFFFFFFu));
        } else {
            k_poll<<<1, 1, 0, s2>>>(flag, (unsigned)r);
        }
        k_stamp<<<1, 1, 0, s2>>>(st + 4);
        CK(hipStreamSynchronize(s2));
        CK(hipStreamSynchronize(s1));
        const double us = 1e3 / rate_khz;
        const double a0 = 0, a1 = (st[1] - st[0]) * us, b0 = (st[2] - st[0]) * us, b1 = (st[3] - st[0]) * us,
                     c = (st[4] - st[0]) * us;
        const bool ok = c >= a1 - 1.0 && c < b1 - 500.0;
        all = all && ok;
        printf("  replay %d: A [%.0f, %.0f]  B [%.0f, %.0f]  consumer at %.0f us  -> %s\n", r, a0, a1, b0, b1, c,
               ok ? "OVERLAP (after A, inside B)" : (c < a1 ? "TOO EARLY" : "NO OVERLAP"));
    }
    hipGraphExecDestroy(exec);
    hipGraphDestroy(graph);
    return all;
}

int main() {
    const char* names[4] = {"", "external event-record node + hipStreamWaitEvent", "k_bump + hipStreamWaitValue32",
                            "k_bump + polling kernel"};
    int can = 0;
    hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    for (int mech = 1; mech <= 3; ++mech) {
        printf("mechanism %d: %s\n", mech, names[mech]);
        const bool ok = run(mech, 3);
        printf("  => %s\n", ok ? "WORKS" : "does not work");
        (void)hipGetLastError();
    }
    return 0;
}
