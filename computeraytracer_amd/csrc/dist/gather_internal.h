// gather_internal.h -- interface between context.hip and dist/gather_rccl.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>
#include <string>

#include "build_internal.h"

namespace crt {

struct GatherState {
    void *comm = nullptr;          // ncclComm_t
    uint32_t rank = 0, world = 0;
    DeviceBuffer gathered, full;   // rank-major concatenation of the band buffers; the assembled frame (when the caller passes no buffer)
};

uint32_t gather_max_rows(uint32_t height, uint32_t world);
bool comm_unique_id(void *id128, std::string &err);
bool comm_init(GatherState &g, const void *id128, uint32_t rank, uint32_t world, std::string &err);
void comm_destroy(GatherState &g);
bool gather_image(hipStream_t stream, GatherState &g, const float *band_dev, uint32_t width, uint32_t height, float *full_dev, std::string &err);
bool gather_full_buffer(GatherState &g, uint32_t width, uint32_t height, float **out, std::string &err);
bool bands_assemble(hipStream_t stream, const float *gathered_dev, float *full_dev, uint32_t width, uint32_t height, uint32_t world, std::string &err);

}  // namespace crt
