// common.cpp -- error reporting and library identity for libsvdx.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../../include/svdx.h"

static thread_local char g_err[512] = "";

void svdx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int svdx_version(void) { return SVDX_VERSION; }

extern "C" int svdx_last_error(char* buf, size_t n) {
    if (!buf || n == 0) return -1;
    strncpy(buf, g_err, n - 1);
    buf[n - 1] = 0;
    return 0;
}

extern "C" int svdx_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        svdx_set_error("no HIP device");
        return 0;
    }
    int dev = 0;
    hipGetDevice(&dev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        svdx_set_error("hipGetDeviceProperties failed");
        return 0;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        svdx_set_error("device arch %s is not gfx950 (libsvdx is MI355X-only)", prop.gcnArchName);
        return 0;
    }
    return 1;
}

// ---- launch plans -------------------------------------------------------------------------------------------------------------------------
// What SURVEY.md 8(b) calls svdx_unet_forward / svdx_unet_backward / svdx_workspace_bytes, in the form this library can honour: the step's
// ~1,500 launches are issued by the host-side operators (Python), which also own every buffer; a PLAN is the list of launches of one such
// pass -- kernel, grid, block, LDS bytes and the argument bytes (pointers into the caller's buffers included) -- recorded while the pass runs
// or is being captured, and replayed from C alone.  The caller keeps the buffers the recorded pointers refer to alive and in place (the
// memory pool of the captured step does exactly that).
namespace {
struct PlanNode {
    const void* fn; dim3 grid, block; unsigned lds;
    std::vector<size_t> off;          // byte offset of every argument inside `blob` (16-byte aligned)
    std::vector<char> blob;
};
struct Plan { std::vector<PlanNode> nodes; size_t arg_bytes = 0; };
thread_local Plan* g_rec = nullptr;
}  // namespace

bool svdx_plan_recording() { return g_rec != nullptr; }

void svdx_plan_record(const void* fn, dim3 grid, dim3 block, unsigned lds, void* const* args, const size_t* sizes, int nargs) {
    PlanNode n;
    n.fn = fn; n.grid = grid; n.block = block; n.lds = lds;
    size_t total = 0;
    for (int i = 0; i < nargs; ++i) { n.off.push_back(total); total += (sizes[i] + 15) & ~(size_t)15; }
    n.blob.resize(total + 16);
    char* base = n.blob.data();
    for (int i = 0; i < nargs; ++i) memcpy(base + n.off[i], args[i], sizes[i]);
    g_rec->arg_bytes += total;
    g_rec->nodes.push_back(std::move(n));
}

extern "C" int svdx_plan_begin(void) {
    if (g_rec) { svdx_set_error("svdx_plan_begin: this thread is already recording a plan"); return -2; }
    g_rec = new Plan();
    return 0;
}

extern "C" int svdx_plan_end(void** plan) {
    if (!g_rec || !plan) { svdx_set_error("svdx_plan_end: no plan is being recorded on this thread"); return -2; }
    *plan = g_rec;
    g_rec = nullptr;
    return 0;
}

extern "C" int64_t svdx_plan_launches(const void* plan) { return plan ? (int64_t)static_cast<const Plan*>(plan)->nodes.size() : -1; }

extern "C" int64_t svdx_plan_bytes(const void* plan) {
    if (!plan) return -1;
    const Plan* p = static_cast<const Plan*>(plan);
    return (int64_t)(p->arg_bytes + p->nodes.size() * sizeof(PlanNode));
}

extern "C" int svdx_plan_replay(const void* plan, void* stream) {
    if (!plan) { svdx_set_error("svdx_plan_replay: null plan"); return -2; }
    const Plan* p = static_cast<const Plan*>(plan);
#ifdef SVDX_SIM
    if (!p->nodes.empty()) { svdx_set_error("svdx_plan_replay: the simulator build records no launches"); return -1; }
#else
    std::vector<void*> ptrs;
    for (const PlanNode& n : p->nodes) {
        ptrs.resize(n.off.size());
        char* base = const_cast<char*>(n.blob.data());
        for (size_t i = 0; i < n.off.size(); ++i) ptrs[i] = base + n.off[i];
        hipError_t e = hipLaunchKernel(n.fn, n.grid, n.block, ptrs.data(), n.lds, (hipStream_t)stream);
        if (e != hipSuccess) { svdx_set_error("svdx_plan_replay: launch %zu failed: %s", (size_t)(&n - p->nodes.data()), hipGetErrorString(e)); return -1; }
    }
#endif
    return 0;
}

extern "C" int svdx_plan_free(void* plan) {
    delete static_cast<Plan*>(plan);
    return 0;
}
