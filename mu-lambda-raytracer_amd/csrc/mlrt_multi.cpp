// csrc/mlrt_multi.cpp - one frame over several GPUs of one node from ONE process (include/mlrt.h, mlrt_multi_*).
//
// The reference parallelises Renderer::render over rows, one task per row (/root/reference/src/raytrace.rs:172-186).  Here
// the rows are dealt to the devices round-robin (rows differ in cost: contiguous bands would unbalance final_scene), the
// scene is replicated (<= 2 MB), every device renders its rows on its own stream with its own context, and ONE gather
// brings the finished rows - one byte per channel - to the first device: RCCL over xGMI (ncclGroupStart { ncclSend per
// peer | ncclRecv per peer on the root } ncclGroupEnd; RCCL has no gather primitive), then one assemble kernel and one
// device-to-host copy.  Every sample's Philox stream is keyed by (seed, pixel, sample) and accumulation is integer, so the
// frame is bit-identical for any number of devices (tested).
//
// librccl.so is loaded on first use (dlopen): a single-GPU caller of libmlrt_hip.so never pays for it.  A device list with
// repeats (a rehearsal of the N-device code path on fewer GPUs) cannot form an RCCL communicator: its rows travel by
// hipMemcpyPeerAsync instead (MLRT_GATHER_COPY), which is also selectable explicitly.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include "../../include/mlrt.h"

extern "C" int mlrt_multi_rows_to_bytes(const int32_t* d_rgb, uint8_t* d_bytes, int n, hipStream_t stream);
extern "C" int mlrt_multi_assemble(const uint8_t* d_gathered, size_t stride, int n_devices, int width, int height, int32_t* d_frame, hipStream_t stream);

namespace mlrt {
void set_last_error(const std::string& msg);
}

namespace {

struct Fail {
    int32_t code;
    std::string msg;
};
#define MULTI_REQUIRE(cond, text) \
    do { if (!(cond)) throw Fail{MLRT_ERR_INVALID, std::string(text)}; } while (0)
#define MULTI_HIP(call) \
    do { hipError_t e_ = (call); if (e_ != hipSuccess) throw Fail{MLRT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)}; } while (0)
#define MULTI_MLRT(call) \
    do { const int32_t rc_ = (call); if (rc_ != MLRT_OK) throw Fail{rc_, std::string(mlrt_last_error())}; } while (0)

// The handful of RCCL entry points the gather needs, resolved from librccl.so at first use.
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl& rccl() {
    static Rccl r;
    if (r.lib) return r;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) throw Fail{MLRT_ERR_UNSUPPORTED, std::string("cannot load librccl.so for the multi-device gather: ") + dlerror()};
    auto sym = [&](const char* n) {
        void* p = dlsym(r.lib, n);
        if (!p) throw Fail{MLRT_ERR_UNSUPPORTED, std::string("librccl.so lacks ") + n};
        return p;
    };
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    return r;
}
// The caller's current device, restored on every way out of an entry point: the calls below switch devices as they go, and
// an embedding application (or a torch process) does not expect libmlrt_hip to leave another device current.
struct DeviceGuard {
    int saved = -1;
    DeviceGuard() { if (hipGetDevice(&saved) != hipSuccess) saved = -1; }
    ~DeviceGuard() { if (saved >= 0) (void)hipSetDevice(saved); }
};

// The frame's plan, pure host arithmetic (CPU-tested through mlrt_debug_multi_plan): which gather the device list gets, the rows of
// every device (mlrt_rows_for_device: round-robin), the stride of a device's slice of the root's gather area - device 0 holds the most
// rows - and so the offset k * stride a peer's rows are received at.
struct MultiPlan {
    int32_t gather = MLRT_GATHER_COPY;
    size_t stride = 0;
    std::vector<int32_t> n_rows;
};
void check_device_list(const int32_t* devices, int32_t n_devices, int32_t gather_mode, int32_t& gather_out) {
    MULTI_REQUIRE(devices, "devices is NULL");
    MULTI_REQUIRE(n_devices >= 1 && n_devices <= 64, "n_devices out of range");
    MULTI_REQUIRE(gather_mode == MLRT_GATHER_AUTO || gather_mode == MLRT_GATHER_RCCL || gather_mode == MLRT_GATHER_COPY, "unknown gather mode");
    for (int k = 0; k < n_devices; k++) MULTI_REQUIRE(devices[k] >= 0, "negative device ordinal");
    const bool distinct = std::set<int32_t>(devices, devices + n_devices).size() == (size_t)n_devices;
    MULTI_REQUIRE(gather_mode != MLRT_GATHER_RCCL || distinct, "MLRT_GATHER_RCCL needs distinct devices (an RCCL communicator has one rank per GPU)");
    gather_out = gather_mode == MLRT_GATHER_AUTO ? ((distinct && n_devices > 1) ? MLRT_GATHER_RCCL : MLRT_GATHER_COPY) : gather_mode;
}
void plan_frame(int32_t n_devices, int32_t width, int32_t height, MultiPlan& P) {
    MULTI_REQUIRE(width >= 2 && height >= 2, "image must be at least 2x2");
    const size_t row_elems = (size_t)width * 3;
    P.n_rows.assign((size_t)n_devices, 0);
    for (int k = 0; k < n_devices; k++) P.n_rows[(size_t)k] = mlrt_rows_for_device(height, k, n_devices, nullptr);
    P.stride = (size_t)P.n_rows[0] * row_elems;  // device 0 has the most rows
}

#define MULTI_RCCL(call) \
    do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) throw Fail{MLRT_ERR_HIP, std::string(#call) + ": " + rccl().GetErrorString(r_)}; } while (0)

}  // namespace

struct mlrt_multi {
    int32_t n = 0, precision = 0, gather = MLRT_GATHER_COPY;
    std::vector<int32_t> devices;
    std::vector<mlrt_context*> ctx;
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> done;           // device k's bytes are ready / have arrived (copy gather)
    std::vector<int32_t*> d_rows, d_rgb;    // per device: its row list and its finished rows
    std::vector<uint8_t*> d_bytes;          // per device: the rows as bytes (device 0: a slice of d_gathered)
    std::vector<int32_t> n_rows;
    uint8_t* d_gathered = nullptr;          // root: [n][stride]
    int32_t* d_frame = nullptr;             // root: [H][W][3]
    size_t stride = 0;
    int32_t width = 0, height = 0;          // geometry the buffers are sized for
    std::vector<ncclComm_t> comms;
    double frame_ms = -1.0;
};

namespace {

void release_frame_buffers(mlrt_multi* m) {
    for (int k = 0; k < m->n; k++) {
        (void)hipSetDevice(m->devices[(size_t)k]);
        if (m->d_rows[(size_t)k]) (void)hipFree(m->d_rows[(size_t)k]);
        if (m->d_rgb[(size_t)k]) (void)hipFree(m->d_rgb[(size_t)k]);
        // (device 0's byte buffer is a slice of d_gathered unless an RCCL self-transfer was forced: then it is an allocation of its own)
        if (m->d_bytes[(size_t)k] && m->d_bytes[(size_t)k] != m->d_gathered) (void)hipFree(m->d_bytes[(size_t)k]);
        m->d_rows[(size_t)k] = nullptr; m->d_rgb[(size_t)k] = nullptr; m->d_bytes[(size_t)k] = nullptr;
    }
    (void)hipSetDevice(m->devices[0]);
    if (m->d_gathered) (void)hipFree(m->d_gathered);
    if (m->d_frame) (void)hipFree(m->d_frame);
    m->d_gathered = nullptr; m->d_frame = nullptr;
    m->width = m->height = 0;
}

void destroy(mlrt_multi* m) {
    if (!m) return;
    release_frame_buffers(m);
    for (ncclComm_t c : m->comms) if (c) (void)rccl().CommDestroy(c);
    for (int k = 0; k < m->n; k++) {
        (void)hipSetDevice(m->devices[(size_t)k]);
        if (m->done[(size_t)k]) (void)hipEventDestroy(m->done[(size_t)k]);
        if (m->streams[(size_t)k]) (void)hipStreamDestroy(m->streams[(size_t)k]);
        if (m->ctx[(size_t)k]) mlrt_context_destroy(m->ctx[(size_t)k]);
    }
    delete m;
}

// Buffers for a W x H frame: row lists (uploaded once), row and byte buffers per device, the gather area and the frame on the root.
void size_for(mlrt_multi* m, int32_t width, int32_t height) {
    if (m->width == width && m->height == height) return;
    release_frame_buffers(m);
    const size_t row_elems = (size_t)width * 3;
    MultiPlan plan;
    plan_frame(m->n, width, height, plan);
    m->stride = plan.stride;
    MULTI_HIP(hipSetDevice(m->devices[0]));
    MULTI_HIP(hipMalloc(&m->d_gathered, (size_t)m->n * m->stride));
    MULTI_HIP(hipMalloc(&m->d_frame, (size_t)height * row_elems * sizeof(int32_t)));
    for (int k = 0; k < m->n; k++) {
        std::vector<int32_t> rows((size_t)height);
        const int32_t count = mlrt_rows_for_device(height, k, m->n, rows.data());
        MULTI_REQUIRE(count == plan.n_rows[(size_t)k], "internal error: row plan");
        m->n_rows[(size_t)k] = count;
        if (count == 0) continue;
        MULTI_HIP(hipSetDevice(m->devices[(size_t)k]));
        MULTI_HIP(hipMalloc(&m->d_rows[(size_t)k], (size_t)count * sizeof(int32_t)));
        MULTI_HIP(hipMemcpy(m->d_rows[(size_t)k], rows.data(), (size_t)count * sizeof(int32_t), hipMemcpyHostToDevice));
        MULTI_HIP(hipMalloc(&m->d_rgb[(size_t)k], (size_t)count * row_elems * sizeof(int32_t)));
        // the root's own rows are written straight into its slice of the gather area; with an RCCL self-transfer forced
        // (MLRT_GATHER_RCCL on one device) they take the wire like everybody else's
        if (k == 0 && !(m->gather == MLRT_GATHER_RCCL && m->n == 1)) m->d_bytes[0] = m->d_gathered;
        else MULTI_HIP(hipMalloc(&m->d_bytes[(size_t)k], (size_t)count * row_elems));
    }
    m->width = width; m->height = height;
}

template <typename F>
int32_t guarded(F&& f) {
    try {
        f();
        return MLRT_OK;
    } catch (const Fail& e) {
        mlrt::set_last_error(e.msg);
        return e.code;
    } catch (const std::exception& e) {
        mlrt::set_last_error(std::string("internal error: ") + e.what());
        return MLRT_ERR_INVALID;
    }
}

}  // namespace

extern "C" {

int32_t mlrt_rows_for_device(int32_t image_height, int32_t k, int32_t n_devices, int32_t* rows_out) {
    if (image_height < 0 || n_devices < 1 || k < 0 || k >= n_devices) return 0;
    int32_t count = 0;
    for (int32_t j = k; j < image_height; j += n_devices) {
        if (rows_out) rows_out[count] = j;
        count++;
    }
    return count;
}

int32_t mlrt_multi_create(const mlrt_scene* scene, const int32_t* devices, int32_t n_devices, int32_t precision, const mlrt_options* options,
                          int32_t gather_mode, mlrt_multi** out) {
    mlrt_multi* m = nullptr;
    DeviceGuard guard;
    const int32_t rc = guarded([&] {
        MULTI_REQUIRE(scene && devices && out, "scene/devices/out is NULL");
        *out = nullptr;
        int32_t gather = MLRT_GATHER_COPY;
        check_device_list(devices, n_devices, gather_mode, gather);
        mlrt_options opts;
        if (options) opts = *options; else mlrt_options_from_env(&opts);
        m = new mlrt_multi();
        m->n = n_devices; m->precision = precision;
        m->devices.assign(devices, devices + n_devices);
        m->gather = gather;
        m->ctx.assign((size_t)n_devices, nullptr); m->streams.assign((size_t)n_devices, nullptr); m->done.assign((size_t)n_devices, nullptr);
        m->d_rows.assign((size_t)n_devices, nullptr); m->d_rgb.assign((size_t)n_devices, nullptr); m->d_bytes.assign((size_t)n_devices, nullptr);
        m->n_rows.assign((size_t)n_devices, 0);
        for (int k = 0; k < n_devices; k++) {
            MULTI_MLRT(mlrt_context_create_opts(scene, devices[k], precision, &opts, &m->ctx[(size_t)k]));  // (validates the ordinal)
            MULTI_HIP(hipSetDevice(devices[k]));
            MULTI_HIP(hipStreamCreateWithFlags(&m->streams[(size_t)k], hipStreamNonBlocking));
            MULTI_HIP(hipEventCreateWithFlags(&m->done[(size_t)k], hipEventDisableTiming));
        }
        if (m->gather == MLRT_GATHER_RCCL) {
            m->comms.assign((size_t)n_devices, nullptr);
            MULTI_RCCL(rccl().CommInitAll(m->comms.data(), n_devices, m->devices.data()));
        } else {
            // peer copies: let the root address its peers' memory directly where the fabric allows (xGMI); otherwise the
            // runtime stages the copy
            for (int k = 1; k < n_devices; k++) {
                if (devices[k] == devices[0]) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, devices[0], devices[k]) == hipSuccess && can) {
                    (void)hipSetDevice(devices[0]);
                    const hipError_t e = hipDeviceEnablePeerAccess(devices[k], 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) throw Fail{MLRT_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e)};
                    (void)hipGetLastError();
                }
            }
        }
        *out = m;
    });
    if (rc != MLRT_OK && m) destroy(m);
    return rc;
}

void mlrt_multi_destroy(mlrt_multi* m) {
    DeviceGuard guard;
    destroy(m);
}

// Test hook (not part of include/mlrt.h): the plan mlrt_multi_create + mlrt_multi_render follow for a device list and a frame, with no
// device call - argument checks, the gather chosen, rows per device, the stride of the gather area.  out = gather, stride in bytes,
// then n_devices row counts, then n_devices receive offsets in bytes (k * stride).  Returns what mlrt_multi_create would for the arguments.
int32_t mlrt_debug_multi_plan(const int32_t* devices, int32_t n_devices, int32_t gather_mode, int32_t width, int32_t height, int64_t* out) {
    return guarded([&] {
        MULTI_REQUIRE(out, "out is NULL");
        int32_t gather = MLRT_GATHER_COPY;
        check_device_list(devices, n_devices, gather_mode, gather);
        MultiPlan plan;
        plan_frame(n_devices, width, height, plan);
        out[0] = gather; out[1] = (int64_t)plan.stride;
        for (int k = 0; k < n_devices; k++) { out[2 + k] = plan.n_rows[(size_t)k]; out[2 + n_devices + k] = (int64_t)((size_t)k * plan.stride); }
    });
}

int32_t mlrt_multi_render(mlrt_multi* m, const mlrt_camera* camera, const mlrt_params* params, int32_t* rgb_out, mlrt_progress_fn progress, void* user) {
    DeviceGuard guard;
    return guarded([&] {
        MULTI_REQUIRE(m && camera && params && rgb_out, "multi/camera/params/rgb_out is NULL");
        MULTI_REQUIRE(params->image_width >= 2 && params->image_height >= 2, "image must be at least 2x2");
        const auto t0 = std::chrono::steady_clock::now();
        size_for(m, params->image_width, params->image_height);
        const size_t row_elems = (size_t)params->image_width * 3;
        // every device: its rows, then the rows as bytes, all on its own stream; nothing here waits for a device
        for (int k = 0; k < m->n; k++) {
            const int32_t count = m->n_rows[(size_t)k];
            if (count == 0) continue;
            MULTI_MLRT(mlrt_context_render_device(m->ctx[(size_t)k], camera, params, m->d_rows[(size_t)k], count, m->d_rgb[(size_t)k], nullptr, m->streams[(size_t)k]));
            MULTI_HIP(hipSetDevice(m->devices[(size_t)k]));
            const int rc = mlrt_multi_rows_to_bytes(m->d_rgb[(size_t)k], m->d_bytes[(size_t)k], (int)((size_t)count * row_elems), m->streams[(size_t)k]);
            if (rc != 0) throw Fail{MLRT_ERR_HIP, std::string("rows_to_bytes launch: ") + hipGetErrorString((hipError_t)rc)};
        }
        // the one gather of the frame
        if (m->gather == MLRT_GATHER_RCCL) {
            const bool self = m->n == 1;  // (forced RCCL on one device: the root's rows take the wire too)
            MULTI_RCCL(rccl().GroupStart());
            for (int k = self ? 0 : 1; k < m->n; k++) {
                const size_t bytes = (size_t)m->n_rows[(size_t)k] * row_elems;
                if (bytes == 0) continue;
                MULTI_RCCL(rccl().Send(m->d_bytes[(size_t)k], bytes, ncclUint8, 0, m->comms[(size_t)k], m->streams[(size_t)k]));
                MULTI_RCCL(rccl().Recv(m->d_gathered + (size_t)k * m->stride, bytes, ncclUint8, k, m->comms[0], m->streams[0]));
            }
            MULTI_RCCL(rccl().GroupEnd());
        } else {
            for (int k = 1; k < m->n; k++) {
                const size_t bytes = (size_t)m->n_rows[(size_t)k] * row_elems;
                if (bytes == 0) continue;
                MULTI_HIP(hipSetDevice(m->devices[(size_t)k]));
                if (m->devices[(size_t)k] == m->devices[0]) MULTI_HIP(hipMemcpyAsync(m->d_gathered + (size_t)k * m->stride, m->d_bytes[(size_t)k], bytes, hipMemcpyDeviceToDevice, m->streams[(size_t)k]));
                else MULTI_HIP(hipMemcpyPeerAsync(m->d_gathered + (size_t)k * m->stride, m->devices[0], m->d_bytes[(size_t)k], m->devices[(size_t)k], bytes, m->streams[(size_t)k]));
                MULTI_HIP(hipEventRecord(m->done[(size_t)k], m->streams[(size_t)k]));
                MULTI_HIP(hipStreamWaitEvent(m->streams[0], m->done[(size_t)k], 0));
            }
        }
        MULTI_HIP(hipSetDevice(m->devices[0]));
        const int rc = mlrt_multi_assemble(m->d_gathered, m->stride, m->n, params->image_width, params->image_height, m->d_frame, m->streams[0]);
        if (rc != 0) throw Fail{MLRT_ERR_HIP, std::string("assemble launch: ") + hipGetErrorString((hipError_t)rc)};
        MULTI_HIP(hipMemcpyAsync(rgb_out, m->d_frame, (size_t)params->image_height * row_elems * sizeof(int32_t), hipMemcpyDeviceToHost, m->streams[0]));
        // the peers' streams finish before the root's (the gather orders them); report progress as they do
        if (progress) {
            int32_t rows_done = 0;
            for (int k = m->n - 1; k >= 1; k--) {
                MULTI_HIP(hipSetDevice(m->devices[(size_t)k]));
                MULTI_HIP(hipStreamSynchronize(m->streams[(size_t)k]));
                rows_done += m->n_rows[(size_t)k];
                if (m->n_rows[(size_t)k] > 0) progress(rows_done, params->image_height, user);
            }
        }
        MULTI_HIP(hipSetDevice(m->devices[0]));
        MULTI_HIP(hipStreamSynchronize(m->streams[0]));
        for (int k = 1; k < m->n; k++) { MULTI_HIP(hipSetDevice(m->devices[(size_t)k])); MULTI_HIP(hipStreamSynchronize(m->streams[(size_t)k])); }
        for (int k = 0; k < m->n; k++) if (m->n_rows[(size_t)k] > 0) MULTI_MLRT(mlrt_context_check(m->ctx[(size_t)k]));
        if (progress) progress(params->image_height, params->image_height, user);
        m->frame_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    });
}

int32_t mlrt_multi_info(mlrt_multi* m, double* frame_ms, double* kernel_ms_per_device, int32_t* gather_used) {
    if (!m) { mlrt::set_last_error("mlrt_multi_info: NULL argument"); return MLRT_ERR_INVALID; }
    if (frame_ms) *frame_ms = m->frame_ms;
    if (kernel_ms_per_device) for (int k = 0; k < m->n; k++) kernel_ms_per_device[k] = m->n_rows[(size_t)k] > 0 ? mlrt_context_last_kernel_ms(m->ctx[(size_t)k]) : 0.0;
    if (gather_used) *gather_used = m->gather;
    return MLRT_OK;
}

int32_t mlrt_render_multi(const mlrt_scene* scene, const mlrt_camera* camera, const mlrt_params* params, int32_t precision, const int32_t* devices,
                          int32_t n_devices, int32_t* rgb_out, mlrt_progress_fn progress, void* user) {
    mlrt_multi* m = nullptr;
    mlrt_options o;
    mlrt_options_from_env(&o);
    if (params && params->tracer == MLRT_TRACER_SINGLE_LIGHT) o.single_light_tracer = 1;
    int32_t rc = mlrt_multi_create(scene, devices, n_devices, precision, &o, MLRT_GATHER_AUTO, &m);
    if (rc != MLRT_OK) return rc;
    rc = mlrt_multi_render(m, camera, params, rgb_out, progress, user);
    mlrt_multi_destroy(m);
    return rc;
}

}  // extern "C"
