// gather_rccl.hip -- the one exchange step of the multi-GPU mode: after the last iteration every
// rank's packed band buffer is all-gathered over RCCL (xGMI) and un-interleaved into the full frame, behind the C ABI, so
// that a C++ engine needs nothing but libcrt.so + librccl.  The reference has no counterpart (one GPU, one queue:
// engine/src/vk_engine.cpp:156-157).
//
// RCCL is bound at run time (dlopen of its soname): libcrt.so then loads on a box without RCCL, and in a process
// that already holds a copy (PyTorch ships its own under the same soname) both users share that one copy instead of
// ending up with two runtimes.  Only five entry points are used: ncclGetUniqueId, ncclCommInitRank, ncclAllGather,
// ncclCommDestroy, ncclGetErrorString.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdlib>

#include <mutex>
#include <string>

#include "build_internal.h"
#include "gather_internal.h"

namespace crt {

namespace {

// the slice of rccl.h this file needs (declared here so that building libcrt.so does not need the RCCL headers)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;               // 0 = ncclSuccess
constexpr int kNcclFloat = 7;           // ncclFloat32 / ncclFloat
static_assert(sizeof(ncclUniqueId) == CRT_COMM_ID_BYTES, "ncclUniqueId is 128 bytes (NCCL_UNIQUE_ID_BYTES)");

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
    // CRT_RCCL_SONAME: one soname or path to bind instead of the defaults (a site-specific RCCL build; the CPU suite uses it
    // to check that a box without RCCL gets an error code and a message, not a crash)
    const char *override_name = getenv("CRT_RCCL_SONAME");
    const char *defaults[] = {"librccl.so.1", "librccl.so"};
    const char *one[] = {override_name};
    const char *const *names = override_name && *override_name ? one : defaults;
    const int n_names = override_name && *override_name ? 1 : 2;
    for (int i = 0; i < n_names; i++) if (!g_rccl.handle) g_rccl.handle = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);     // a copy the process already holds
    for (int i = 0; i < n_names; i++) if (!g_rccl.handle) g_rccl.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!g_rccl.handle) {
        const char *e = dlerror();      // ONE call: dlerror() clears the message it returns
        g_rccl.error = std::string("RCCL is not loadable: ") + (e ? e : "dlopen failed");
        return;
    }
    auto sym = [&](const char *s) { void *p = dlsym(g_rccl.handle, s); if (!p && g_rccl.error.empty()) g_rccl.error = std::string("RCCL lacks ") + s; return p; };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
}

bool rccl_ready(std::string &err) {
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.error.empty()) { err = g_rccl.error; return false; }
    return true;
}

std::string nccl_error(const char *what, ncclResult_t r) {
    return std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error") + " (" + std::to_string(r) + ")";
}

}  // namespace

// Rank-major concatenation of the packed band buffers (every rank padded to max_rows rows) -> the frame.  Row y lives in
// rank (y / 8) % world at packed row (y / 8 / world) * 8 + y % 8 (crt_band_rows; computeraytracer_amd/dist.py: row_index).
__global__ __launch_bounds__(256) void k_bands_assemble(const float4 *gathered, float4 *full, uint32_t width, uint32_t height, uint32_t world, uint32_t max_rows) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)width * height) return;
    const uint32_t y = (uint32_t)(i / width), x = (uint32_t)(i - (size_t)y * width);
    const uint32_t band = y / 8u, rank = band % world, prow = (band / world) * 8u + (y % 8u);
    full[i] = gathered[((size_t)rank * max_rows + prow) * width + x];
}

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return false; } } while (0)

static bool ensure_g(DeviceBuffer &b, size_t bytes, std::string &err) {
    if (b.bytes >= bytes && b.ptr) return true;
    if (b.ptr) { hipError_t e = hipFree(b.ptr); (void)e; b.ptr = nullptr; b.bytes = 0; }
    HIP_TRY(hipMalloc(&b.ptr, bytes < 256 ? 256 : bytes));
    b.bytes = bytes < 256 ? 256 : bytes;
    return true;
}

uint32_t gather_max_rows(uint32_t height, uint32_t world) {      // rows of rank 0's packed band buffer: no rank owns more
    const uint32_t nb = (height + 7u) / 8u, w = world > 1 ? world : 1;
    return ((nb + w - 1u) / w) * 8u;
}

bool comm_unique_id(void *id128, std::string &err) {
    if (!rccl_ready(err)) return false;
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != 0) { err = nccl_error("ncclGetUniqueId", r); return false; }
    memcpy(id128, id.internal, sizeof id.internal);
    return true;
}

bool comm_init(GatherState &g, const void *id128, uint32_t rank, uint32_t world, std::string &err) {
    if (!rccl_ready(err)) return false;
    if (g.comm) { err = "a communicator is already initialised for this context"; return false; }
    if (world == 0 || rank >= world) { err = "rank must be below world"; return false; }
    ncclUniqueId id;
    memcpy(id.internal, id128, sizeof id.internal);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = g_rccl.CommInitRank(&comm, (int)world, id, (int)rank);      // collective: every rank of `world` calls it
    if (r != 0) { err = nccl_error("ncclCommInitRank", r); return false; }
    g.comm = comm; g.rank = rank; g.world = world;
    return true;
}

void comm_destroy(GatherState &g) {
    if (g.comm && g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t)g.comm);
    g.comm = nullptr; g.world = 0; g.rank = 0;
    DeviceBuffer *all[] = {&g.gathered, &g.full};
    for (DeviceBuffer *b : all) if (b->ptr) { hipError_t e = hipFree(b->ptr); (void)e; b->ptr = nullptr; b->bytes = 0; }
}

bool bands_assemble(hipStream_t stream, const float *gathered_dev, float *full_dev, uint32_t width, uint32_t height, uint32_t world, std::string &err) {
    if (width == 0 || height == 0 || world == 0) { err = "empty frame"; return false; }
    const size_t px = (size_t)width * height;
    hipLaunchKernelGGL(k_bands_assemble, dim3((uint32_t)((px + 255) / 256)), dim3(256), 0, stream, (const float4 *)gathered_dev, (float4 *)full_dev,
                       width, height, world, gather_max_rows(height, world));
    HIP_TRY(hipGetLastError());
    return true;
}

// band = this rank's packed band buffer, padded to gather_max_rows rows (context.hip allocates it that way).
bool gather_image(hipStream_t stream, GatherState &g, const float *band_dev, uint32_t width, uint32_t height, float *full_dev, std::string &err) {
    if (!g.comm) { err = "crt_comm_init has not been called"; return false; }
    const uint32_t max_rows = gather_max_rows(height, g.world);
    const size_t send_floats = (size_t)max_rows * width * 4;
    if (!ensure_g(g.gathered, send_floats * 4 * g.world, err)) return false;
    const ncclResult_t r = g_rccl.AllGather(band_dev, g.gathered.ptr, send_floats, kNcclFloat, (ncclComm_t)g.comm, stream);
    if (r != 0) { err = nccl_error("ncclAllGather", r); return false; }
    return bands_assemble(stream, (const float *)g.gathered.ptr, full_dev, width, height, g.world, err);
}

bool gather_full_buffer(GatherState &g, uint32_t width, uint32_t height, float **out, std::string &err) {
    if (!ensure_g(g.full, (size_t)width * height * 16, err)) return false;
    *out = (float *)g.full.ptr;
    return true;
}

}  // namespace crt
