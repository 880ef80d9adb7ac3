// Device/stream/memory plumbing behind the vx_* C ABI (include/visp_hip_kernels.h).
// Replaces what the reference gets from ggml_backend_* (src/visp/ml.cpp:59-95, 479-502, 708-741).
#include "vx_common.h"

#include <cstring>
#include <mutex>
#include <vector>

static thread_local char g_vx_error[512];

void vx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_vx_error, sizeof g_vx_error, fmt, ap);
    va_end(ap);
}

hipError_t vx_ensure_dynamic_lds(const void* kernel, int bytes) {
    // (kernel, device) pairs already raised to `bytes`; a handful of kernels x up to 8 devices
    struct entry { const void* kernel; int device; int bytes; };
    static std::mutex mu;
    static std::vector<entry> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    for (entry& d : done)
        if (d.kernel == kernel && d.device == dev) {
            if (d.bytes >= bytes) return hipSuccess;
            e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e == hipSuccess) d.bytes = bytes;
            return e;
        }
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.push_back({kernel, dev, bytes});
    return e;
}

extern "C" {

const char* vx_last_error(void) { return g_vx_error; }

int vx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int vx_set_device(int index) {
    VX_CHECK(hipSetDevice(index));
    return 1;
}

int vx_device_info(int index, char* name, int name_cap, char* arch, int arch_cap, size_t* total_mem,
                   size_t* free_mem, int* n_cu) {
    hipDeviceProp_t prop;
    VX_CHECK(hipGetDeviceProperties(&prop, index));
    if (name && name_cap > 0) snprintf(name, name_cap, "%s", prop.name);
    if (arch && arch_cap > 0) snprintf(arch, arch_cap, "%s", prop.gcnArchName);
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (total_mem) *total_mem = prop.totalGlobalMem;
    if (free_mem) {
        size_t f = 0, t = 0;
        int cur = 0;
        VX_CHECK(hipGetDevice(&cur));
        VX_CHECK(hipSetDevice(index));
        VX_CHECK(hipMemGetInfo(&f, &t));
        VX_CHECK(hipSetDevice(cur));
        *free_mem = f;
    }
    return 1;
}

int vx_malloc(void** ptr, size_t bytes) {
    VX_CHECK(hipMalloc(ptr, bytes ? bytes : 16));
    return 1;
}
int vx_free(void* ptr) {
    if (ptr) VX_CHECK(hipFree(ptr));
    return 1;
}
int vx_memset(void* ptr, int value, size_t bytes, void* stream) {
    VX_CHECK(hipMemsetAsync(ptr, value, bytes, as_stream(stream)));
    return 1;
}
int vx_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
    VX_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return 1;
}
int vx_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
    VX_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    VX_CHECK(hipStreamSynchronize(as_stream(stream)));
    return 1;
}
int vx_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
    VX_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return 1;
}
int vx_malloc_host(void** ptr, size_t bytes) {
    VX_CHECK(hipHostMalloc(ptr, bytes ? bytes : 16, hipHostMallocDefault));
    return 1;
}
int vx_free_host(void* ptr) {
    if (ptr) VX_CHECK(hipHostFree(ptr));
    return 1;
}
int vx_memcpy_h2d_async(void* dst, const void* src, size_t bytes, void* stream) {
    VX_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return 1;
}
int vx_memcpy_d2h_async(void* dst, const void* src, size_t bytes, void* stream) {
    VX_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    return 1;
}
int vx_event_sync(void* ev) {
    VX_CHECK(hipEventSynchronize(reinterpret_cast<hipEvent_t>(ev)));
    return 1;
}
int vx_stream_create(void** stream) {
    hipStream_t s;
    VX_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return 1;
}
int vx_stream_destroy(void* stream) {
    if (stream) VX_CHECK(hipStreamDestroy(as_stream(stream)));
    return 1;
}
int vx_stream_sync(void* stream) {
    VX_CHECK(hipStreamSynchronize(as_stream(stream)));
    return 1;
}
int vx_event_create(void** ev) {
    hipEvent_t e;
    VX_CHECK(hipEventCreate(&e));
    *ev = e;
    return 1;
}
int vx_event_destroy(void* ev) {
    if (ev) VX_CHECK(hipEventDestroy(reinterpret_cast<hipEvent_t>(ev)));
    return 1;
}
int vx_event_record(void* ev, void* stream) {
    VX_CHECK(hipEventRecord(reinterpret_cast<hipEvent_t>(ev), as_stream(stream)));
    return 1;
}
int vx_event_elapsed_ms(void* start, void* stop, float* ms) {
    VX_CHECK(hipEventSynchronize(reinterpret_cast<hipEvent_t>(stop)));
    VX_CHECK(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
    return 1;
}

int vx_stream_wait_event(void* stream, void* ev) {
    VX_CHECK(hipStreamWaitEvent(as_stream(stream), reinterpret_cast<hipEvent_t>(ev), 0));
    return 1;
}

int vx_graph_begin_capture(void* stream) {
    VX_CHECK(hipStreamBeginCapture(as_stream(stream), hipStreamCaptureModeThreadLocal));
    return 1;
}
int vx_graph_end_capture(void* stream, void** graph_exec) {
    hipGraph_t graph = nullptr;
    VX_CHECK(hipStreamEndCapture(as_stream(stream), &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    VX_CHECK(e);
    *graph_exec = exec;
    return 1;
}
int vx_graph_launch(void* graph_exec, void* stream) {
    VX_CHECK(hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(graph_exec), as_stream(stream)));
    return 1;
}
int vx_graph_destroy(void* graph_exec) {
    if (graph_exec) VX_CHECK(hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(graph_exec)));
    return 1;
}

} // extern "C"
