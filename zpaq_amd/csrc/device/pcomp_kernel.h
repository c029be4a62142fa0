// PCOMP on the device: PostProcessor::write in its PROG state (libzpaq.cpp:2231-2239) -- every decoded byte of a
// segment goes through the block's post-processing program (ZPAQL::run with OUT, 1027-1262), then the EOS call.
// The program is translated to straight-line HIP by host/codegen.cpp like HCOMP is (the archive carries it, so this
// is the MI355X analogue of the reference JIT-compiling PCOMP, 3231-3811); one lane per segment, its H / M / R
// arrays in HBM.  The inverse transforms (LZ77 copy loops, inverse BWT list walking, E8E9) are serial per segment;
// the batch supplies the parallelism.
#pragma once
#ifndef ZPQ_LANE_VM
#define ZPQ_LANE_VM 1
#endif
#include "spec_kernel.h"

namespace zpq {

struct PcompOut {
  g_u8* p;
  unsigned cap, n;
  __device__ __forceinline__ void operator()(unsigned a) {
    if (n < cap) p[n] = (unsigned char)a;
    ++n;
  }
};

template <class Post>
__device__ __forceinline__ void pcomp_body(const PcompJob* jobs, unsigned n) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const PcompJob job = jobs[i];
  PcompOut out{(g_u8*)job.out, job.out_cap, 0u};
  g_u8* const M = (g_u8*)job.M;
  g_u32* const H = (g_u32*)job.H;
  g_u32* const R = (g_u32*)job.R;
  const g_u8* const in = (const g_u8*)job.in;
  unsigned b = 0, c = 0, d = 0, f = 0;
  int status = 0;
  for (unsigned k = 0; k < job.in_len && !status; ++k) status = Post::pcomp(in[k], b, c, d, f, M, H, R, out);
  if (!status) status = Post::pcomp(0xFFFFFFFFu, b, c, d, f, M, H, R, out);
  job.result[0] = out.n;
  job.result[1] = (unsigned)status;
}

}  // namespace zpq
