/*
 * rt_reduce.cpp — RCCL reduce of per-GPU accumulation buffers (the exchange step of the multi-GPU render; replaces the
 * host-side stitch of the two half images, reference main.cu:711-730).
 *
 * librccl.so is opened on first use (a single-GPU process never pays for it); types come from <rccl/rccl.h>.
 * The exchange logic (argument checks, communicator cache, the grouped reduces) works on a table of function
 * pointers and on plain device / stream / buffer lists, so that it can be driven without GPUs: the real table is
 * filled from librccl + hipSetDevice; a build with -DRT_TEST_HOOKS (make testhooks -> build/librt_reduce_test.so, loaded
 * by tests/test_reduce_cpu.py only; the product library is built without it) adds rt_debug_reduce_with_fake, which
 * injects a recording table (no machine here or in the one-GPU box has two devices to run the real one on).
 */
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/rt_api.h"
#include "rt_host.h"
#include "rt_reduce.h"

static bool frame_ok(int w, int h) { return w > 0 && h > 0 && (long long)w * h <= (1ll << 30); }

namespace {
struct RcclOps {
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    hipError_t (*SetDevice)(int) = nullptr;
};
struct RcclState {
    void* handle = nullptr;
    RcclOps ops;
    std::vector<int> devices;          /* the device list the cached communicators were made for */
    std::vector<ncclComm_t> comms;
};
RcclState g_rccl;
std::mutex g_rccl_mutex;

int rccl_load()
{
    if (g_rccl.handle) return RT_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) { rt_set_error(std::string("rt_reduce_accum: cannot load librccl: ") + dlerror()); return RT_ERR_RCCL; }
    bool ok = true;
    auto sym = [&](const char* name) { void* p = dlsym(h, name); if (!p) ok = false; return p; };
    RcclOps& o = g_rccl.ops;
    o.CommInitAll = (decltype(o.CommInitAll))sym("ncclCommInitAll");
    o.CommDestroy = (decltype(o.CommDestroy))sym("ncclCommDestroy");
    o.GroupStart = (decltype(o.GroupStart))sym("ncclGroupStart");
    o.GroupEnd = (decltype(o.GroupEnd))sym("ncclGroupEnd");
    o.Reduce = (decltype(o.Reduce))sym("ncclReduce");
    o.GetErrorString = (decltype(o.GetErrorString))sym("ncclGetErrorString");
    o.SetDevice = &hipSetDevice;
    if (!ok) { dlclose(h); rt_set_error("rt_reduce_accum: librccl lacks an expected symbol"); return RT_ERR_RCCL; }
    g_rccl.handle = h;
    return RT_OK;
}
#define RCCLCHK(st, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { \
    rt_set_error(std::string(#call) + ": " + (st).ops.GetErrorString(r_)); return RT_ERR_RCCL; } } while (0)

/* checks the lists and makes sure `st` holds communicators for exactly this device list (made on first use, kept
 * for the next call, replaced when the list changes) */
int rccl_comms(RcclState& st, const char* who, const int* devices, float* const* accum, int n)
{
    for (int i = 0; i < n; ++i) {
        if (!accum[i]) { rt_set_error(std::string(who) + ": null context or buffer"); return RT_ERR_ARG; }
        for (int j = 0; j < i; ++j)
            if (devices[j] == devices[i]) { rt_set_error(std::string(who) + ": two contexts on one device"); return RT_ERR_ARG; }
    }
    const std::vector<int> want(devices, devices + n);
    if (st.devices != want) {
        for (ncclComm_t c : st.comms) if (c) (void)st.ops.CommDestroy(c);
        st.comms.assign(n, nullptr);
        st.devices.clear();
        RCCLCHK(st, st.ops.CommInitAll(st.comms.data(), n, want.data()));
        st.devices = want;
    }
    return RT_OK;
}

/* rank i = position i in the lists; in place: the root's buffer receives the sum */
int reduce_accum_impl(RcclState& st, const int* devices, const hipStream_t* streams, float* const* accum, int n, uint64_t count, int root)
{
    if (n < 1 || root < 0 || root >= n || count == 0 || !devices || !streams || !accum) { rt_set_error("rt_reduce_accum: bad arguments"); return RT_ERR_ARG; }
    int rc = rccl_comms(st, "rt_reduce_accum", devices, accum, n);
    if (rc) return rc;
    RCCLCHK(st, st.ops.GroupStart());
    for (int i = 0; i < n; ++i) {
        if (st.ops.SetDevice(devices[i]) != hipSuccess) { (void)st.ops.GroupEnd(); rt_set_error("rt_reduce_accum: hipSetDevice failed"); return RT_ERR_HIP; }
        ncclResult_t r = st.ops.Reduce(accum[i], accum[i], (size_t)count, ncclFloat, ncclSum, root, st.comms[i], streams[i]);
        if (r != ncclSuccess) { (void)st.ops.GroupEnd(); rt_set_error(std::string("ncclReduce: ") + st.ops.GetErrorString(r)); return RT_ERR_RCCL; }
    }
    RCCLCHK(st, st.ops.GroupEnd());
    return RT_OK;
}

/* one reduce per band, rooted at the band's owner, all in one group: every GPU receives 1/n of the frame
 * instead of GPU 0 receiving all of it (bands may differ in height, which ncclReduceScatter cannot express) */
int reduce_scatter_impl(RcclState& st, const int* devices, const hipStream_t* streams, float* const* accum, int n, int w, int h, int* rows)
{
    if (n < 1 || !frame_ok(w, h) || !rows || !devices || !streams || !accum) { rt_set_error("rt_reduce_scatter_rows: bad arguments"); return RT_ERR_ARG; }
    int rc = rccl_comms(st, "rt_reduce_scatter_rows", devices, accum, n);
    if (rc) return rc;
    for (int i = 0; i <= n; ++i) rows[i] = (int)((long long)h * i / n);
    RCCLCHK(st, st.ops.GroupStart());
    for (int band = 0; band < n; ++band) {
        const size_t off = (size_t)rows[band] * w * 3, cnt = (size_t)(rows[band + 1] - rows[band]) * w * 3;
        if (cnt == 0) continue;
        for (int i = 0; i < n; ++i) {
            if (st.ops.SetDevice(devices[i]) != hipSuccess) { (void)st.ops.GroupEnd(); rt_set_error("rt_reduce_scatter_rows: hipSetDevice failed"); return RT_ERR_HIP; }
            ncclResult_t r = st.ops.Reduce(accum[i] + off, accum[i] + off, cnt, ncclFloat, ncclSum, band, st.comms[i], streams[i]);
            if (r != ncclSuccess) { (void)st.ops.GroupEnd(); rt_set_error(std::string("ncclReduce: ") + st.ops.GetErrorString(r)); return RT_ERR_RCCL; }
        }
    }
    RCCLCHK(st, st.ops.GroupEnd());
    return RT_OK;
}
}  // namespace

int rt_reduce_accum_lists(const int* devices, const hipStream_t* streams, float* const* accum, int n, uint64_t count, int root)
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    int rc = rccl_load();
    if (rc) return rc;
    return reduce_accum_impl(g_rccl, devices, streams, accum, n, count, root);
}

int rt_reduce_scatter_lists(const int* devices, const hipStream_t* streams, float* const* accum, int n, int w, int h, int* rows)
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    int rc = rccl_load();
    if (rc) return rc;
    return reduce_scatter_impl(g_rccl, devices, streams, accum, n, w, h, rows);
}

#ifdef RT_TEST_HOOKS
/* ---- test hook (tests/test_reduce_cpu.py; declared nowhere but here and in the test; -DRT_TEST_HOOKS builds only) ----
 * Runs the exchange logic above over a FAKE RCCL table that writes every call into `log` as one line of text,
 * with no GPU and no librccl: mode 0 = reduce_accum_impl, 1 = reduce_scatter_impl.  The fake state (communicator
 * cache) lives across calls until `reset` is non-zero, so that a test can see communicators being reused for the
 * same device list and replaced for another.  fail_at >= 0 makes the fail_at-th ncclReduce of the call fail.
 * Buffers are fake too: rank i's buffer is the address base + i * 2^32, only ever printed as offsets. */
namespace {
std::string* g_fake_log = nullptr;
int g_fake_reduce_calls = 0, g_fake_fail_at = -1, g_fake_next_comm = 0, g_fake_device = -1;
void fake_say(const std::string& s) { if (g_fake_log) { *g_fake_log += s; *g_fake_log += '\n'; } }
ncclResult_t fake_init(ncclComm_t* comms, int n, const int* devs)
{
    std::string s = "init";
    for (int i = 0; i < n; ++i) { comms[i] = (ncclComm_t)(uintptr_t)(0x1000 + ++g_fake_next_comm); s += " dev" + std::to_string(devs[i]) + "=comm" + std::to_string(g_fake_next_comm); }
    fake_say(s);
    return ncclSuccess;
}
ncclResult_t fake_destroy(ncclComm_t c) { fake_say("destroy comm" + std::to_string((int)((uintptr_t)c - 0x1000))); return ncclSuccess; }
ncclResult_t fake_group_start() { fake_say("group_start"); return ncclSuccess; }
ncclResult_t fake_group_end() { fake_say("group_end"); return ncclSuccess; }
ncclResult_t fake_reduce(const void* src, void* dst, size_t count, ncclDataType_t dt, ncclRedOp_t op, int root, ncclComm_t c, hipStream_t st)
{
    const uintptr_t a = (uintptr_t)src;
    fake_say("reduce dev" + std::to_string(g_fake_device) + " comm" + std::to_string((int)((uintptr_t)c - 0x1000)) + " rank_buf" + std::to_string((int)(a >> 32) - 1) +
             " off" + std::to_string((unsigned long long)(a & 0xffffffffull) / sizeof(float)) + " count" + std::to_string((unsigned long long)count) +
             " root" + std::to_string(root) + " stream" + std::to_string((int)(uintptr_t)st) + (src == dst ? " inplace" : " COPY") +
             (dt == ncclFloat && op == ncclSum ? " f32sum" : " WRONGTYPE"));
    return g_fake_reduce_calls++ == g_fake_fail_at ? ncclInternalError : ncclSuccess;
}
const char* fake_error_string(ncclResult_t) { return "fake failure"; }
hipError_t fake_set_device(int d) { g_fake_device = d; return hipSuccess; }
RcclState g_fake_state;
}  // namespace

extern "C" int rt_debug_reduce_with_fake(int mode, const int* devices, int n, unsigned long long count, int root, int w, int h, int* rows,
                                         int null_buffer, int fail_at, int reset, char* log, int log_cap)
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (reset) { g_fake_state = RcclState(); g_fake_next_comm = 0; }
    RcclOps& o = g_fake_state.ops;
    o.CommInitAll = fake_init; o.CommDestroy = fake_destroy; o.GroupStart = fake_group_start; o.GroupEnd = fake_group_end;
    o.Reduce = fake_reduce; o.GetErrorString = fake_error_string; o.SetDevice = fake_set_device;
    std::string text;
    g_fake_log = &text;
    g_fake_reduce_calls = 0;
    g_fake_fail_at = fail_at;
    std::vector<float*> bufs((size_t)(n > 0 ? n : 0));
    std::vector<hipStream_t> streams(bufs.size());
    for (size_t i = 0; i < bufs.size(); ++i) {
        bufs[i] = (int)i == null_buffer ? nullptr : (float*)(uintptr_t)((i + 1) << 32);
        streams[i] = (hipStream_t)(uintptr_t)(100 + i);
    }
    const int rc = mode == 0 ? reduce_accum_impl(g_fake_state, devices, streams.data(), bufs.data(), n, count, root)
                             : reduce_scatter_impl(g_fake_state, devices, streams.data(), bufs.data(), n, w, h, rows);
    g_fake_log = nullptr;
    if (log && log_cap > 0) { std::strncpy(log, text.c_str(), (size_t)log_cap - 1); log[log_cap - 1] = 0; }
    return rc;
}
#endif
