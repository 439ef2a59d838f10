// Library-level entry points of libstylerenderer_hip.so (error strings, ABI version).
#include "common.h"

extern "C" int sr_abi_version(void) { return 3; }

extern "C" const char* sr_error_string(int code) {
    if (code == SR_OK) return "ok";
    if (code == SR_EINVAL) return "stylerenderer_amd: invalid argument (size, null pointer or unsupported combination)";
    if (code == SR_ERANGE) return "stylerenderer_amd: size exceeds the kernel's index range";
    if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
    return "stylerenderer_amd: unknown error";
}

// ---- in-graph signalling for the overlapped gradient reduction (include/stylerenderer_amd.h) ----------------------
// An event recorded on a CAPTURING stream with hipEventRecordExternal becomes an event-record NODE of the hipGraph:
// every replay records it at that point of the graph, and a hipStreamWaitEvent issued on another stream after
// hipGraphLaunch waits for exactly that point — the rest of the graph keeps running (scripts/event_graph_probe.cpp:
// consumer starts 12 us after the producer node inside a replay on gfx950 / ROCm 7).  On a stream that is not
// capturing this is a plain hipEventRecord.  torch.cuda.Event(external=True) refuses to do this on ROCm builds.
extern "C" int sr_event_create(void** event) {
    if (!event) return SR_EINVAL;
    hipEvent_t ev = nullptr;
    const hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) return static_cast<int>(e);
    *event = ev;
    return SR_OK;
}

extern "C" int sr_event_destroy(void* event) {
    if (!event) return SR_OK;
    return static_cast<int>(hipEventDestroy(static_cast<hipEvent_t>(event)));
}

extern "C" int sr_event_record(void* event, sr_stream_t stream) {
    if (!event) return SR_EINVAL;
    hipStream_t s = sr_stream(stream);
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    hipError_t e = hipStreamIsCapturing(s, &st);
    if (e != hipSuccess) return static_cast<int>(e);
    e = hipEventRecordWithFlags(static_cast<hipEvent_t>(event), s,
                                st == hipStreamCaptureStatusActive ? hipEventRecordExternal : hipEventRecordDefault);
    return static_cast<int>(e);
}

extern "C" int sr_stream_wait_event(sr_stream_t stream, void* event) {
    if (!event) return SR_EINVAL;
    return static_cast<int>(hipStreamWaitEvent(sr_stream(stream), static_cast<hipEvent_t>(event), 0));
}
