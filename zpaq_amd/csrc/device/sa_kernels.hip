// Suffix arrays of many blocks at once on the MI355X -- the sort behind compressBlock's byte-aligned LZ77 and BWT
// pre-processors (LZBuffer with a suffix array, libzpaq.cpp:6463-6883; divsufsort / divbwt, 4658-6434).  A suffix array
// is canonical, so any correct builder gives the reference's parse and the reference's BWT; divsufsort's induced sorting
// is a serial algorithm, this is prefix doubling (Manber-Myers / Larsson-Sadakane) laid out for a GPU:
//
//   every suffix of every block is one element of ONE array (block b owns [off_b, off_b + n_b)); round h sorts all
//   elements by the 64-bit key  block << 48 | rank_h[i] << 24 | rank_h[i + h]  (rank 0 = past the end of the block: the
//   end of the string sorts before every byte, as in the reference) with one radix sort, then renames: equal neighbours
//   keep a rank, a flag + prefix sum gives the others theirs, relative to the block's first element.  After the round
//   ranks order the suffixes by their first 2h bytes; when every key of a round is distinct the ranks are the inverse
//   suffix array.  log2(longest repeat) rounds: 3-4 for random data, 6-8 for text, log2(n) for a block of zeros.
//
// Each round streams the arrays a few times at HBM rate (radix sort of 64-bit keys + 32-bit values, one gather, one
// scan, one scatter): bandwidth work, no MFMA.  Blocks of up to 2^24 bytes and 65 535 blocks per call; the caller
// (host/blocks.cpp) keeps the host's SA-IS for anything else and for small batches.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <cstdint>

#include "lz77_kernel.h"
#include "sa_kernels.h"

namespace zpq {

namespace {

// rank of round 0: byte + 1 (1..256); block id and position of every element
__global__ __launch_bounds__(256) void sa_init_kernel(const uint8_t* const* in, const uint64_t* off, uint32_t nblocks, uint64_t total,
                                                      uint32_t* rank, uint16_t* blk) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  // block of element i: binary search in off[0..nblocks]
  uint32_t lo = 0, hi = nblocks;
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (off[mid] <= i) lo = mid; else hi = mid; }
  blk[i] = (uint16_t)lo;
  rank[i] = (uint32_t)in[lo][i - off[lo]] + 1u;
}

__global__ __launch_bounds__(256) void sa_keys_kernel(const uint32_t* rank, const uint16_t* blk, const uint64_t* off, uint64_t total, uint32_t h,
                                                      uint64_t* keys, uint32_t* vals) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t b = blk[i];
  const uint64_t end = off[b + 1];
  const uint32_t r2 = i + h < end ? rank[i + h] : 0u;
  keys[i] = (uint64_t)b << 48 | (uint64_t)rank[i] << 24 | r2;
  vals[i] = (uint32_t)i;
}

// 1 where a sorted key differs from its left neighbour (the first element of the array counts as different)
__global__ __launch_bounds__(256) void sa_flags_kernel(const uint64_t* keys, uint64_t total, uint32_t* flags) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= total) return;
  flags[j] = (j == 0 || keys[j] != keys[j - 1]) ? 1u : 0u;
}

// new rank of the element at sorted position j: names counted from the block's first sorted position (= off[block]: the
// block id is the major key), starting at 1
__global__ __launch_bounds__(256) void sa_rename_kernel(const uint64_t* keys, const uint32_t* vals, const uint32_t* scan, const uint64_t* off,
                                                        uint64_t total, uint32_t* rank) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= total) return;
  const uint32_t b = (uint32_t)(keys[j] >> 48);
  rank[vals[j]] = scan[j] - scan[off[b]] + 1u;
}

// ranks are a permutation of 1..n_b inside every block now: sa[off_b + rank - 1] = position in the block
__global__ __launch_bounds__(256) void sa_invert_kernel(const uint32_t* rank, const uint16_t* blk, const uint64_t* off, uint64_t total, uint32_t* sa) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint64_t o = off[blk[i]];
  sa[o + rank[i] - 1u] = (uint32_t)(i - o);
}

inline unsigned grid_for(uint64_t n) { return (unsigned)((n + 255) / 256); }

__global__ __launch_bounds__(256) void lz77_search_kernel(const uint8_t* in_all, const uint32_t* sa_all, const uint32_t* rank_all, const uint16_t* blk,
                                                          const LzBlock* blocks, uint64_t total, uint4* res) {
  lz77_search_body(in_all, sa_all, rank_all, blk, blocks, total, res);
}
__global__ __launch_bounds__(64) void lz77_walk_kernel(const LzBlock* blocks, const uint4* res, LzTok* toks, uint32_t* counts) {
  lz77_walk_body(blocks, res, toks, counts);
}
__global__ __launch_bounds__(256) void bwt_emit_kernel(const uint8_t* in_all, const uint32_t* sa_all, const uint16_t* blk, const LzBlock* blocks,
                                                       uint64_t total, uint8_t* out_all, uint32_t* idx) {
  bwt_emit_body(in_all, sa_all, blk, blocks, total, out_all, idx);
}

}  // namespace

size_t sa_workspace_bytes(uint64_t total, uint32_t nblocks) {
  size_t sort_tmp = 0, scan_tmp = 0;
  (void)rocprim::radix_sort_pairs(nullptr, sort_tmp, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)total, 0, 64);
  (void)rocprim::inclusive_scan(nullptr, scan_tmp, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)total, rocprim::plus<uint32_t>());
  const size_t a = (size_t)((total + 63) & ~63ull);
  // keys x 2, vals x 2, rank, flags / scan, blk, off, pointers, library scratch
  return a * (8 + 8 + 4 + 4 + 4 + 4 + 2) + ((size_t)nblocks + 2) * 16 + (sort_tmp > scan_tmp ? sort_tmp : scan_tmp) + 4096;
}

// d_in[b] -> bytes of block b ON THE DEVICE, d_off[0..nblocks] = exclusive prefix sums of the lengths (device), total = d_off[nblocks];
// d_sa receives the suffix arrays back to back (d_sa + off[b] = block b's).  `ws` = sa_workspace_bytes(total, nblocks) bytes of device memory.
hipError_t build_suffix_arrays(const uint8_t* const* d_in, const uint64_t* d_off, uint32_t nblocks, uint64_t total, uint32_t max_len,
                               uint32_t* d_sa, void* ws, size_t ws_bytes, hipStream_t st, uint32_t* rounds_out, SaSideArrays* side) {
  if (!total) return hipSuccess;
  if (nblocks > 65535u || max_len >= (1u << 24) || total >= (1ull << 32)) return hipErrorInvalidValue;
  const size_t a = (size_t)((total + 63) & ~63ull);
  uint8_t* p = (uint8_t*)ws;
  uint64_t* keys = (uint64_t*)p; p += a * 8;
  uint64_t* keys2 = (uint64_t*)p; p += a * 8;
  uint32_t* vals = (uint32_t*)p; p += a * 4;
  uint32_t* vals2 = (uint32_t*)p; p += a * 4;
  uint32_t* rank = (uint32_t*)p; p += a * 4;
  uint32_t* flags = (uint32_t*)p; p += a * 4;
  uint16_t* blk = (uint16_t*)p; p += a * 2;
  p = (uint8_t*)(((uintptr_t)p + 255) & ~(uintptr_t)255);
  void* tmp = p;
  const size_t tmp_bytes = ws_bytes - (size_t)(p - (uint8_t*)ws);
  const unsigned g = grid_for(total);
  hipLaunchKernelGGL(sa_init_kernel, dim3(g), dim3(256), 0, st, d_in, d_off, nblocks, total, rank, blk);
  // bits of the key that matter: block id on top of two 24-bit ranks
  unsigned blk_bits = 1;
  while ((1u << blk_bits) < nblocks) ++blk_bits;
  uint32_t h = 1, rounds = 0;
  for (;; h <<= 1) {
    hipLaunchKernelGGL(sa_keys_kernel, dim3(g), dim3(256), 0, st, rank, blk, d_off, total, h, keys, vals);
    size_t need = tmp_bytes;
    hipError_t e = rocprim::radix_sort_pairs(tmp, need, keys, keys2, vals, vals2, (size_t)total, 0, 48 + blk_bits, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sa_flags_kernel, dim3(g), dim3(256), 0, st, keys2, total, flags);
    need = tmp_bytes;
    e = rocprim::inclusive_scan(tmp, need, flags, flags, (size_t)total, rocprim::plus<uint32_t>(), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sa_rename_kernel, dim3(g), dim3(256), 0, st, keys2, vals2, flags, d_off, total, rank);
    ++rounds;
    // every key distinct <=> the last prefix sum equals the number of elements
    uint32_t names = 0;
    e = hipMemcpyAsync(&names, flags + (total - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    if ((uint64_t)names == total || h >= max_len) break;
  }
  hipLaunchKernelGGL(sa_invert_kernel, dim3(g), dim3(256), 0, st, rank, blk, d_off, total, d_sa);
  if (rounds_out) *rounds_out = rounds;
  if (side) { side->rank = rank; side->blk = blk; }
  return hipGetLastError();
}

hipError_t launch_sort_preprocessors(const uint8_t* in_all, const uint32_t* sa_all, const SaSideArrays& side, const LzBlock* blocks, uint32_t nblocks,
                                     uint64_t total, bool any_lz, bool any_bwt, void* res, LzTok* toks, uint32_t* counts, uint8_t* bwt_out,
                                     uint32_t* bwt_idx, hipStream_t st) {
  if (!total || !nblocks) return hipSuccess;
  const unsigned g = grid_for(total);
  if (any_lz) {
    hipLaunchKernelGGL(lz77_search_kernel, dim3(g), dim3(256), 0, st, in_all, sa_all, (const uint32_t*)side.rank, (const uint16_t*)side.blk, blocks, total, (uint4*)res);
    hipLaunchKernelGGL(lz77_walk_kernel, dim3(nblocks), dim3(64), 0, st, blocks, (const uint4*)res, toks, counts);
  }
  if (any_bwt)
    hipLaunchKernelGGL(bwt_emit_kernel, dim3(g), dim3(256), 0, st, in_all, sa_all, (const uint16_t*)side.blk, blocks, total, bwt_out, bwt_idx);
  return hipGetLastError();
}

}  // namespace zpq
