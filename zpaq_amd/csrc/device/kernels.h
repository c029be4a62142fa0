// Launch prototypes for device/kernels.hip (host-callable).
#pragma once
#include <hip/hip_runtime_api.h>

#include "layout.h"

namespace zpq {
hipError_t launch_init_arena(const BlockJob* d_jobs, uint32_t nblocks, const DeviceTables* d_tb,
                             uint32_t chunks, hipStream_t st);
hipError_t launch_code_serial(bool decode, const BlockJob* d_jobs, BlockResult* d_res, uint32_t nblocks,
                              const DeviceTables* d_tb, hipStream_t st);
hipError_t launch_code_wave(bool decode, const BlockJob* d_jobs, BlockResult* d_res, uint32_t nblocks,
                            const DeviceTables* d_tb, hipStream_t st);
hipError_t launch_sha1(const Sha1Job* d_jobs, uint32_t n, uint8_t* d_digests, hipStream_t st);
hipError_t launch_selftest(int32_t* d_out, hipStream_t st);
}  // namespace zpq
