// runtime.hip -- error strings, grow-only workspace, library entry points.
#include <stdarg.h>
#include <string.h>

#include <mutex>

#include "common.h"

namespace nnhip {

static thread_local char g_err[512] = "";

void set_last_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_status(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    set_last_error("%s: %s (%d)", what, hipGetErrorString(e), (int)e);
    return (int)e;
}

static std::mutex g_ws_mu;
// Two grow-only blocks: 0 = the general workspace (split-K slabs, column-sum partials, ...), 1 = the slabs of the GROUPED parameter-
// gradient launch (gemm.hip: gemm_f32_wgrad_group) -- its own block because that launch may run on a side stream next to kernels of
// the main stream that use block 0 (neunet_hip/_lib.py: NNHIP_WGRAD_STREAM; round 6).
static void* g_ws[2] = {nullptr, nullptr};
static size_t g_ws_bytes[2] = {0, 0};
static bool g_ws_locked = false;   // nnhipWorkspaceLock: a captured hipGraph holds the blocks' addresses

void* workspace_arena(int which, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    if (bytes <= g_ws_bytes[which]) return g_ws[which];
    if (g_ws_locked) {
        set_last_error("workspace is locked at %zu bytes (a captured hipGraph uses it) but %zu bytes were requested; "
                       "run the new shape eagerly before capturing, or nnhipWorkspaceLock(0)", g_ws_bytes[which], bytes);
        return nullptr;
    }
    if (g_ws[which]) {
        (void)hipDeviceSynchronize();  // kernels may still read the old block
        (void)hipFree(g_ws[which]);
        g_ws[which] = nullptr;
        g_ws_bytes[which] = 0;
    }
    size_t want = bytes + (bytes >> 2);  // 25 % head-room: fewer regrowths
    want = (want + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    if (hipMalloc(&g_ws[which], want) != hipSuccess) {
        g_ws[which] = nullptr;
        if (hipMalloc(&g_ws[which], bytes) != hipSuccess) {
            g_ws[which] = nullptr;
            return nullptr;
        }
        want = bytes;
    }
    g_ws_bytes[which] = want;
    return g_ws[which];
}
void* workspace(size_t bytes) { return workspace_arena(0, bytes); }

bool workspace_locked() { return g_ws_locked; }

const float* zero_block() {
    static float* z = nullptr;
    static std::once_flag once;
    std::call_once(once, []() {
        float* p = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&p), 256) == hipSuccess && hipMemset(p, 0, 256) == hipSuccess) z = p;
    });
    return z;
}

// Library-owned device words (arrival tickets, the CE denominator scratch) and the workspace partials behind them are ONE set
// per process: two launches that use them must not overlap.  On one stream that is program order.  A launch that arrives on a
// DIFFERENT stream than the previous user is ordered behind everything the device has been given so far (hipDeviceSynchronize):
// the single-stream assumption is enforced instead of assumed (round-3 review).  The previous caller's stream handle is only ever
// COMPARED, never used -- it is the caller's object and may be gone by now (CuPy destroys its streams on garbage collection;
// round-4 advisor: an event recorded on a destroyed stream is undefined behaviour and left the guard failing for the rest of the
// process).  Costs a pointer compare when the stream is the same; a stream change is rare (a few per process) and pays a device
// sync.  While `st` is being captured into a hipGraph nothing is inserted (a sync is illegal there; a captured step is
// single-stream by construction, neunet_hip/graph.py), and a sync refused because ANOTHER stream is capturing is skipped likewise.
static std::mutex g_ss_mu;
static hipStream_t g_ss_last = nullptr;
static bool g_ss_have_last = false;
static hipEvent_t g_ss_ev = nullptr;        // recorded behind the last user's launches (shared_state_done), outside graph captures
static bool g_ss_ev_live = false;
static bool ss_capturing(hipStream_t st) {
    hipStreamCaptureStatus a = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &a) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a != hipStreamCaptureStatusNone;
}
int serialize_shared_state(hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_ss_mu);
    if (g_ss_have_last && g_ss_last != st && !ss_capturing(st)) {
        // round 6 (advisor): the previous user left an EVENT behind its launches (a library-owned object; its stream handle is never
        // touched again) -- the new stream waits for that event instead of the whole device.  No event (the previous user ran inside a
        // graph capture, or event creation failed): the device sync of round 5.
        bool ordered = false;
        if (g_ss_ev_live) {
            if (hipStreamWaitEvent(st, g_ss_ev, 0) == hipSuccess) ordered = true;
            else (void)hipGetLastError();
        }
        if (!ordered) {
            const hipError_t e = hipDeviceSynchronize();
            if (e != hipSuccess) {
                (void)hipGetLastError();          // e.g. a capture in progress elsewhere: nothing to order against, carry on
                if (e != hipErrorStreamCaptureUnsupported && e != hipErrorStreamCaptureImplicit && e != hipErrorStreamCaptureInvalidated)
                    return hip_status(e, "shared-state guard (device synchronize)");
            }
        }
    }
    g_ss_last = st;
    g_ss_have_last = true;
    return 0;
}
// called (through SharedStateUse's destructor) after a user of the shared words has enqueued its launches on `st`
void shared_state_done(hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_ss_mu);
    if (ss_capturing(st)) { g_ss_ev_live = false; return; }      // (an event recorded into a capture is not one a later stream can wait on)
    if (!g_ss_ev && hipEventCreateWithFlags(&g_ss_ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); g_ss_ev = nullptr; }
    if (g_ss_ev && hipEventRecord(g_ss_ev, st) == hipSuccess) g_ss_ev_live = true;
    else { (void)hipGetLastError(); g_ss_ev_live = false; }
}

static unsigned* g_deverr_host = nullptr;
static unsigned* g_deverr_dev = nullptr;
static void device_error_init() {
    static std::once_flag once;
    std::call_once(once, []() {
        void* h = nullptr;
        void* d = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return; }
        memset(h, 0, 64);
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(h); return; }
        g_deverr_host = static_cast<unsigned*>(h);
        g_deverr_dev = static_cast<unsigned*>(d);
    });
}
unsigned* device_error_word() {
    device_error_init();
    return g_deverr_dev;
}
static const char* device_error_text(unsigned code) {
    switch (code) {
        case NNHIP_DEVERR_MLP_BARRIER:
            return "the optimizer-in-backward launch (nnhipLinearReLULinearBackwardAdam) waited 20 s for its blocks to check in and gave up: "
                   "the W2 / b2 update of that step was skipped, the optimizer state is half-stepped";
        default: return "unknown device error code";
    }
}
int device_error_status(const char* who) {
    if (!g_deverr_host) return 0;
    const unsigned code = *const_cast<volatile unsigned*>(g_deverr_host);
    if (code == NNHIP_DEVERR_NONE) return 0;
    set_last_error("%s: device error %u raised by an earlier kernel -- %s (nnhipClearDeviceError() resets it)", who, code, device_error_text(code));
    return NNHIP_EDEVICE;
}

unsigned* sync_words() {
    static unsigned* w = nullptr;
    static std::once_flag once;
    std::call_once(once, []() {
        unsigned* p = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&p), kSyncWords * sizeof(unsigned)) == hipSuccess &&
            hipMemset(p, 0, kSyncWords * sizeof(unsigned)) == hipSuccess)
            w = p;
    });
    return w;
}

}  // namespace nnhip

extern "C" int nnhipVersion(void) { return 210; }

extern "C" int nnhipDeviceError(void) { return nnhip::device_error_status("nnhipDeviceError"); }

extern "C" int nnhipClearDeviceError(void) {
    if (nnhip::g_deverr_host) *const_cast<volatile unsigned*>(nnhip::g_deverr_host) = 0u;
    return 0;
}

// test hook: raise a device error code the way a kernel would (a system-scope store from the device), on `stream`
__global__ void raise_device_error_kernel(unsigned* err, unsigned code) {
    if (err) __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
extern "C" int nnhipRaiseDeviceErrorForTest(int32_t code, nnhipStream_t stream) {
    unsigned* w = nnhip::device_error_word();
    if (!w) { nnhip::set_last_error("nnhipRaiseDeviceErrorForTest: no device error word (pinned allocation failed)"); return NNHIP_ENOMEM; }
    hipLaunchKernelGGL(raise_device_error_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, w, (unsigned)code);
    NNHIP_LAUNCH_CHECK("raise_device_error_kernel");
    return 0;
}

extern "C" int nnhipWorkspaceReserve(int64_t bytes) {
    if (bytes < 0) { nnhip::set_last_error("nnhipWorkspaceReserve: negative size"); return NNHIP_EINVAL; }
    if (bytes > 0 && !nnhip::workspace((size_t)bytes)) return NNHIP_ENOMEM;
    return 0;
}

extern "C" int nnhipWorkspaceLock(int locked) {
    std::lock_guard<std::mutex> lk(nnhip::g_ws_mu);
    nnhip::g_ws_locked = locked != 0;
    return 0;
}

extern "C" const char* nnhipGetLastErrorString(void) { return nnhip::g_err; }

extern "C" int nnhipCleanup(void) {
    nnhip::conv_reduce_cleanup();
    nnhip::colsum_cleanup();
    std::lock_guard<std::mutex> lk(nnhip::g_ws_mu);
    nnhip::g_ws_locked = false;
    int rc = 0;
    bool synced = false;
    for (int w = 0; w < 2; ++w) {
        if (!nnhip::g_ws[w]) continue;
        if (!synced) { (void)hipDeviceSynchronize(); synced = true; }
        hipError_t e = hipFree(nnhip::g_ws[w]);
        nnhip::g_ws[w] = nullptr;
        nnhip::g_ws_bytes[w] = 0;
        if (e != hipSuccess && !rc) rc = nnhip::hip_status(e, "hipFree(workspace)");
    }
    if (rc) return rc;
    return 0;
}
