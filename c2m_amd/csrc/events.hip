// events.hip -- HIP timing events for the measurement side (bench.py's roofline object, ops.ConvProfiler).
// torch.cuda.Event records with the default flags: every record carries a system-scope release fence, i.e. an L2
// write-back between two kernels of the timed region (600 of them per bench step: +2 % step time, and the packed weights
// of the next launch leave the L2).  These events are created with hipEventDisableSystemFence: the time stamps are the
// same, the stream is not fenced.  Nothing on the product path uses them.
#include "common.h"

C2M_API int c2m_event_create(void** out) {
    hipEvent_t e = nullptr;
    const int rc = (int)hipEventCreateWithFlags(&e, hipEventDisableSystemFence);
    *out = (void*)e;
    return rc;
}

C2M_API int c2m_event_record(void* ev, void* stream) {
    return (int)hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
}

// milliseconds between two recorded events (both must have completed: synchronise the stream / device first)
C2M_API int c2m_event_elapsed_ms(void* start, void* stop, float* ms) {
    return (int)hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
}

C2M_API int c2m_event_destroy(void* ev) {
    return (int)hipEventDestroy((hipEvent_t)ev);
}

// ---- ABI self-description (include/c2m_geom.h): the ctypes host parses the header it ships with and refuses a library that was
// built from another one (c2m_amd/_lib.py)
C2M_API int c2m_abi_version(void) { return C2M_ABI_VERSION; }
C2M_API int c2m_geom_len(void) { return C2M_G_LEN; }
C2M_API int c2m_wino_geom_len(void) { return C2M_WG_LEN; }
