// HIP kernels for gfx950 (MI355X).  No CUDA compatibility paths.
//
//   init_arena_kernel   : Predictor::init (libzpaq.cpp:1776-1846) for every block
//                         of a batch, streaming 16-B stores (HBM-write bound).
//   code_serial_kernel  : generic one-lane coder, any header (fallback/baseline).
//   code_wave_kernel    : wave-parallel coder, one ZPAQ block per wavefront,
//                         components spread over lanes (see model_wave.h).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "layout.h"
#include "model_serial.h"
#include "model_wave.h"

namespace zpq {

// ------------------------------------------------------------------ init
__global__ __launch_bounds__(256) void init_arena_kernel(const BlockJob* jobs, const DeviceTables* tb) {
  const BlockJob job = jobs[blockIdx.y];
  const PlanHeader* ph = (const PlanHeader*)job.plan;
  const Segment* segs = (const Segment*)(job.plan + ph->off_seg);
  const uint32_t nseg = ph->nseg;
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
  for (uint32_t s = 0; s < nseg; ++s) {
    const Segment sg = segs[s];
    uint4* dst = (uint4*)(job.arena + sg.off);
    const uint64_t n16 = sg.bytes >> 4;
    switch (sg.kind) {
      case F_ZERO:
      case F_U32: {
        const uint32_t v = sg.kind == F_ZERO ? 0u : sg.value;
        const uint4 q = make_uint4(v, v, v, v);
        for (uint64_t i = tid; i < n16; i += nthreads) dst[i] = q;
        break;
      }
      case F_SSE:
        for (uint64_t i = tid; i < n16; i += nthreads) {
          const uint32_t j = (uint32_t)(i & 7) * 4;
          dst[i] = make_uint4(tb->sse_row[j] | sg.value, tb->sse_row[j + 1] | sg.value,
                              tb->sse_row[j + 2] | sg.value, tb->sse_row[j + 3] | sg.value);
        }
        break;
      case F_ICM:
        for (uint64_t i = tid; i < n16; i += nthreads) dst[i] = ((const uint4*)tb->icm_init)[i];
        break;
      case F_ISSE:
        for (uint64_t i = tid; i < n16; i += nthreads) dst[i] = ((const uint4*)tb->isse_init)[i];
        break;
      case F_MATCHBUF:
        for (uint64_t i = tid; i < n16; i += nthreads) dst[i] = make_uint4(i == 0 ? 1u : 0u, 0, 0, 0);
        break;
    }
  }
}

// ------------------------------------------------------------- serial coder
struct RangeCoder {
  uint32_t low, high;
};

template <bool DEC>
__global__ void code_serial_kernel(const BlockJob* jobs, BlockResult* res, uint32_t nblocks,
                                   const DeviceTables* tb) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  const BlockJob job = jobs[b];
  BlockResult& out_res = res[job.res_slot];
  SerialCtx s;
  serial_open(s, job, tb);
  uint32_t low = 1, high = 0xFFFFFFFFu;
  uint32_t steps = 0;
  int status = 0;
  if (!DEC) {
    // Encoder::compress / encode (libzpaq.cpp:2402-2447)
    uint32_t n = 0;   // bytes produced
    auto put = [&](uint32_t c) { if (n < job.out_cap) job.out[n] = (uint8_t)c; ++n; };
    auto encode = [&](int y, uint32_t p) {
      const uint32_t mid = low + (uint32_t)(((uint64_t)(high - low) * p) >> 16);
      if (y) high = mid; else low = mid + 1;
      while ((high ^ low) < 0x1000000u) {
        put(high >> 24);
        high = high << 8 | 255u;
        low = low << 8;
        low += (low == 0);
      }
    };
    for (uint32_t k = 0; k < job.in_len && !status; ++k) {
      const int c = job.in[k];
      encode(0, 0);
      for (int i = 7; i >= 0; --i) {
        const int pr = serial_predict(s);
        const int y = (c >> i) & 1;
        encode(y, (uint32_t)pr * 2 + 1);
        status = serial_update(s, y);
        ++steps;
        if (status) break;
      }
    }
    if (!status) encode(1, 0);
    if (!status && n > job.out_cap) status = 3;
    out_res.out_len = n;
    out_res.consumed = job.in_len;
  } else {
    // Decoder::decompress / decode (libzpaq.cpp:2104-2155)
    uint32_t rp = 0, n = 0, curr = 0;
    bool eos = false;
    for (int i = 0; i < 4; ++i) {
      if (rp >= job.in_len) { status = 6; break; }
      curr = curr << 8 | job.in[rp++];
    }
    while (!status && !eos && n < job.out_cap) {
      int c = 1;
      for (int bit = -1; bit < 8; ++bit) {
        uint32_t p = 0;
        if (bit >= 0) p = (uint32_t)serial_predict(s) * 2 + 1;
        if (curr < low || curr > high) { status = 2; break; }
        const uint32_t mid = low + (uint32_t)(((uint64_t)(high - low) * p) >> 16);
        int y;
        if (curr <= mid) { y = 1; high = mid; } else { y = 0; low = mid + 1; }
        while ((high ^ low) < 0x1000000u) {
          high = high << 8 | 255u;
          low = low << 8;
          low += (low == 0);
          if (rp >= job.in_len) { status = 6; break; }
          curr = curr << 8 | job.in[rp++];
        }
        if (status) break;
        if (bit < 0) {
          if (y) { eos = true; if (curr != 0) status = 2; break; }
        } else {
          c += c + y;
          status = serial_update(s, y);
          ++steps;
          if (status) break;
        }
      }
      if (status || eos) break;
      job.out[n++] = (uint8_t)(c - 256);
    }
    out_res.out_len = n;
    out_res.consumed = eos ? rp : 0;
  }
  out_res.status = status;
  out_res.steps = steps;
}

// ------------------------------------------------------------------- SHA-1
// FIPS 180-1, one lane per block (the hash is a serial chain over the block's 64-byte groups; the batch supplies the
// parallelism).  Replaces libzpaq::SHA1 (libzpaq.cpp:106-177) for the digest compressBlock stores in the segment
// trailer when the block is already on the device unchanged (methods without pre-processing).
__device__ __forceinline__ uint32_t sha_rol(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

__device__ __forceinline__ void sha1_rounds(uint32_t h[5], uint32_t w[16]) {
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4];
#pragma unroll
  for (int i = 0; i < 80; ++i) {
    uint32_t x;
    if (i < 16) x = w[i];
    else x = w[i & 15] = sha_rol(w[(i + 13) & 15] ^ w[(i + 8) & 15] ^ w[(i + 2) & 15] ^ w[i & 15], 1);
    uint32_t f, k;
    if (i < 20) { f = (b & c) | (~b & d); k = 0x5A827999u; }
    else if (i < 40) { f = b ^ c ^ d; k = 0x6ED9EBA1u; }
    else if (i < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8F1BBCDCu; }
    else { f = b ^ c ^ d; k = 0xCA62C1D6u; }
    const uint32_t t = sha_rol(a, 5) + f + e + k + x;
    e = d; d = c; c = sha_rol(b, 30); b = a; a = t;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e;
}

__global__ __launch_bounds__(64) void sha1_blocks_kernel(const Sha1Job* jobs, uint32_t n, uint8_t* digests) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  const Sha1Job job = jobs[b];
  typedef __attribute__((address_space(1))) const uint8_t g8;
  typedef uint32_t __attribute__((aligned(1))) u32u;
  g8* p = (g8*)job.p;
  uint32_t h[5] = {0x67452301u, 0xEFCDAB89u, 0x98BADCFEu, 0x10325476u, 0xC3D2E1F0u};
  uint32_t w[16];
  const uint32_t full = job.len >> 6;
  for (uint32_t k = 0; k < full; ++k) {
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = __builtin_bswap32(*(__attribute__((address_space(1))) const u32u*)(p + 64u * k + 4u * i));
    sha1_rounds(h, w);
  }
  // the tail: remaining bytes, 0x80, zeros, the bit length as 64 bits big-endian -- one or two more groups
  const uint32_t rem = job.len & 63u;
  g8* q = p + 64u * full;
  const int groups = rem < 56 ? 1 : 2;
  for (int gi = 0; gi < groups; ++gi) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      uint32_t v = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t at = (uint32_t)gi * 64u + 4u * i + j;
        const uint32_t byte = at < rem ? (uint32_t)q[at] : (at == rem ? 0x80u : 0u);
        v = v << 8 | byte;
      }
      w[i] = v;
    }
    if (gi == groups - 1) { w[14] = job.len >> 29; w[15] = job.len << 3; }
    sha1_rounds(h, w);
  }
  uint8_t* out = digests + 20u * job.slot;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16);
    out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i];
  }
}

// ---------------------------------------------------------------- selftest
// Checks the cross-lane idioms the wave kernel relies on (DPP reduction,
// readlane, bpermute shuffles).  out[0]=wave_sum(lane) (2016), out[1]=wave_sum
// of (lane*lane - 1000) (21344), out[2]=readlane(lane*3, 41) (123),
// out[3]=sum of __shfl(lane, (lane+5)&63) over lanes (2016), out[4]=wave_sum
// with only lanes < 19 contributing 7 each (133), out[5]=sum of wave_shr:1 of lane*7 (13671).
__global__ void selftest_kernel(int32_t* out) {
  const int lane = threadIdx.x & 63;
  const int a = wave_sum(lane);
  const int b = wave_sum(lane * lane - 1000);
  const int c = rl(lane * 3, 41);
  const int d = wave_sum(__shfl(lane, (lane + 5) & 63));
  int x = 0;
  if (lane < 19) x = 7;
  const int e = wave_sum(x);
  // wave_shr:1 (used by the specialised kernel's ISSE fast path): lane i reads lane i-1, lane 0 reads 0
  const int f = wave_sum(__builtin_amdgcn_update_dpp(0, lane * 7, 0x138, 0xF, 0xF, false));
  if (lane == 0) { out[0] = a; out[1] = b; out[2] = c; out[3] = d; out[4] = e; out[5] = f; }
}

// ------------------------------------------------------------ launch glue
static inline hipError_t last() { return hipGetLastError(); }

hipError_t launch_init_arena(const BlockJob* d_jobs, uint32_t nblocks, const DeviceTables* d_tb,
                             uint32_t chunks, hipStream_t st) {
  if (!nblocks) return hipSuccess;
  hipLaunchKernelGGL(init_arena_kernel, dim3(chunks, nblocks), dim3(256), 0, st, d_jobs, d_tb);
  return last();
}

hipError_t launch_sha1(const Sha1Job* d_jobs, uint32_t n, uint8_t* d_digests, hipStream_t st) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(sha1_blocks_kernel, dim3((n + 63) / 64), dim3(64), 0, st, d_jobs, n, d_digests);
  return last();
}

hipError_t launch_selftest(int32_t* d_out, hipStream_t st) {
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, st, d_out);
  return last();
}

hipError_t launch_code_serial(bool decode, const BlockJob* d_jobs, BlockResult* d_res, uint32_t nblocks,
                              const DeviceTables* d_tb, hipStream_t st) {
  if (!nblocks) return hipSuccess;
  // one wavefront per block, one active lane: <<<nblocks, 1>>>
  if (decode) hipLaunchKernelGGL(code_serial_kernel<true>, dim3(nblocks), dim3(1), 0, st, d_jobs, d_res, nblocks, d_tb);
  else hipLaunchKernelGGL(code_serial_kernel<false>, dim3(nblocks), dim3(1), 0, st, d_jobs, d_res, nblocks, d_tb);
  return last();
}

hipError_t launch_code_wave(bool decode, const BlockJob* d_jobs, BlockResult* d_res, uint32_t nblocks,
                            const DeviceTables* d_tb, hipStream_t st) {
  if (!nblocks) return hipSuccess;
  const uint32_t wg = (nblocks + kWavesPerGroup - 1) / kWavesPerGroup;
  if (decode) hipLaunchKernelGGL(code_wave_kernel<true>, dim3(wg), dim3(64 * kWavesPerGroup), 0, st, d_jobs, d_res, nblocks, d_tb);
  else hipLaunchKernelGGL(code_wave_kernel<false>, dim3(wg), dim3(64 * kWavesPerGroup), 0, st, d_jobs, d_res, nblocks, d_tb);
  return last();
}

}  // namespace zpq
