// Launch prototypes of device/sa_kernels.hip: suffix arrays of many blocks at once (prefix doubling, rocPRIM sorts).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

#include "layout.h"

namespace zpq {
// what the sorter leaves in its workspace besides the arrays: rank[i] - 1 = position of suffix i in its block's suffix array (the
// inverse array), blk[i] = block of element i
struct SaSideArrays { uint32_t* rank = nullptr; uint16_t* blk = nullptr; };
// device memory build_suffix_arrays needs for `total` bytes of input in `nblocks` blocks
size_t sa_workspace_bytes(uint64_t total, uint32_t nblocks);
// d_in[b] -> block b's bytes on the device; d_off[0..nblocks] = exclusive prefix sums of the lengths; d_sa: the arrays back
// to back (block b's at d_sa + off[b]); max_len < 2^24, nblocks < 65536, total < 2^32.  Synchronises `st` once per round.
hipError_t build_suffix_arrays(const uint8_t* const* d_in, const uint64_t* d_off, uint32_t nblocks, uint64_t total, uint32_t max_len,
                               uint32_t* d_sa, void* ws, size_t ws_bytes, hipStream_t st, uint32_t* rounds_out, SaSideArrays* side = nullptr);
// Behind the sort, for the same batch (device/lz77_kernel.h): the LZ77 parse of the blocks of kind 1 / 2 -- 16 bytes of decisions
// per element in `res`, then the matches taken in toks[blocks[b].tok_off ..) and their number in counts[b] -- and the BWT of
// the blocks of kind 3 (n + 1 bytes at bwt_out + off + b, the index of the whole string in bwt_idx[b]).  in_all: the blocks'
// bytes back to back like the arrays.
hipError_t launch_sort_preprocessors(const uint8_t* in_all, const uint32_t* sa_all, const SaSideArrays& side, const LzBlock* blocks, uint32_t nblocks,
                                     uint64_t total, bool any_lz, bool any_bwt, void* res, LzTok* toks, uint32_t* counts, uint8_t* bwt_out,
                                     uint32_t* bwt_idx, hipStream_t st);
}  // namespace zpq
