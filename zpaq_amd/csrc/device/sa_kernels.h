// Launch prototypes of device/sa_kernels.hip: suffix arrays of many blocks at once (prefix doubling, rocPRIM sorts).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

namespace zpq {
// device memory build_suffix_arrays needs for `total` bytes of input in `nblocks` blocks
size_t sa_workspace_bytes(uint64_t total, uint32_t nblocks);
// d_in[b] -> block b's bytes on the device; d_off[0..nblocks] = exclusive prefix sums of the lengths; d_sa: the arrays back
// to back (block b's at d_sa + off[b]); max_len < 2^24, nblocks < 65536, total < 2^32.  Synchronises `st` once per round.
hipError_t build_suffix_arrays(const uint8_t* const* d_in, const uint64_t* d_off, uint32_t nblocks, uint64_t total, uint32_t max_len,
                               uint32_t* d_sa, void* ws, size_t ws_bytes, hipStream_t st, uint32_t* rounds_out);
}  // namespace zpq
