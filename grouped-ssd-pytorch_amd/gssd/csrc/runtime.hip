// Error string + ABI identification for libgssd_hip.so.
#include <stdarg.h>
#include <string.h>
#include "common.h"

static thread_local char g_err[512] = "";

void gssd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int gssd_abi_version(void) { return 8; }
extern "C" int gssd_conv_desc_size(void) { return (int)sizeof(gssd_conv_desc); }
extern "C" const char* gssd_last_error(void) { return g_err; }
extern "C" const char* gssd_build_arch(void) { return "gfx950"; }

// ---- timing events that become event-record nodes under stream capture (include/gssd_hip.h) ----
extern "C" int gssd_event_create(gssd_event_t* ev) {
    GSSD_CHECK_ARG(ev);
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) {
        gssd_set_error("gssd_event_create: hipEventCreate failed");
        return GSSD_ELAUNCH;
    }
    *ev = e;
    return GSSD_OK;
}
extern "C" int gssd_event_destroy(gssd_event_t ev) {
    GSSD_CHECK_ARG(ev);
    return hipEventDestroy(reinterpret_cast<hipEvent_t>(ev)) == hipSuccess ? GSSD_OK : GSSD_ELAUNCH;
}
extern "C" int gssd_event_record_node(gssd_event_t ev, gssd_stream_t stream) {
    GSSD_CHECK_ARG(ev);
    hipStream_t s = as_stream(stream);
    hipEvent_t e = reinterpret_cast<hipEvent_t>(ev);
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    hipGraph_t graph = nullptr;
    const hipGraphNode_t* deps = nullptr;
    size_t ndeps = 0;
    hipError_t rc = hipStreamGetCaptureInfo_v2(s, &status, &id, &graph, &deps, &ndeps);
    if (rc == hipSuccess && status == hipStreamCaptureStatusActive) {
        // an event-record NODE behind everything the stream has captured so far; what the stream captures next depends on it
        // (hipEventRecordWithFlags(hipEventRecordExternal) does the same where the runtime accepts it: this image's returns invalid argument)
        hipGraphNode_t node = nullptr;
        rc = hipGraphAddEventRecordNode(&node, graph, deps, ndeps, e);
        if (rc == hipSuccess) rc = hipStreamUpdateCaptureDependencies(s, &node, 1, hipStreamSetCaptureDependencies);
    } else {
        rc = hipEventRecord(e, s);
    }
    if (rc != hipSuccess) {
        gssd_set_error("gssd_event_record_node: %s", hipGetErrorString(rc));
        return GSSD_ELAUNCH;
    }
    return GSSD_OK;
}
extern "C" int gssd_event_elapsed_ms(gssd_event_t start, gssd_event_t stop, float* ms) {
    GSSD_CHECK_ARG(start && stop && ms);
    const hipError_t e = hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop));
    if (e != hipSuccess) {
        gssd_set_error("gssd_event_elapsed_ms: %s", hipGetErrorString(e));
        return GSSD_ELAUNCH;
    }
    return GSSD_OK;
}
