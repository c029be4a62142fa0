// The LZ77 parse through a suffix array and the BWT's last column on the MI355X -- compressBlock's pre-processors behind the
// sort (LZBuffer::fill with a suffix array, libzpaq.cpp:6693-6757; the byte-aligned / bit-packed codes of 6759-6883 stay on
// the host: host/preproc.cpp lz77_serialize; divbwt's output, 4658-6434 / 7709-7716).
//
// The reference walks a block greedily: at position i it looks at the neighbours of suffix i (and of i + 1 .. i + lookahead)
// in the suffix array, scores them, takes the best match or a literal, and moves on by what it took.  Only the walk is serial:
// WHAT the search at a position finds depends on the position alone -- and on one bit of history, whether literals are
// pending, which costs a look-ahead match 4 points.  So:
//
//   lz77_search_kernel   one lane per position of every block of the batch: the reference's candidate loop statement for
//                        statement (both directions, `bucket` neighbours each, its break rules), run for both values of
//                        that bit at once over the same candidates; 16 bytes per position: the decision for either value
//   lz77_walk_kernel     one wavefront per block: 64 decisions per load, then a wave-uniform loop (v_readlane) that follows
//                        the chain i -> i + length and lists the matches taken
//   bwt_emit_kernel      one lane per suffix: the byte in front of it (255 for the whole string; its index on the side)
//
// The inverse suffix array is the rank array the sorter ends with (device/sa_kernels.hip).  Bit-exact by construction; the
// emulator runs this file against the host's parse (tests/emu/lz77_emu_main.cpp), the GPU tests against the reference's archives.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#include "layout.h"

namespace zpq {

static const uint32_t kLzMaxMatch = (1u << 14) * 3, kLzMaxLiteral = (1u << 14) / 4;
// the decision word of a position packs (match length - literals) in bits 0..15, the literals in front in bits 16..23 and the
// "take it" flag in bit 31 (lz77_search_body / lz77_walk_body): the kernel's contract, checked where the words are made
static_assert(kLzMaxMatch < 65536u, "an LZ77 match length must fit 16 bits of the decision word");
static const uint32_t kLzMaxLookahead = 255u;      // literals in front of a match: 8 bits

typedef unsigned long long __attribute__((aligned(1))) lz_u64u;

__device__ __forceinline__ int lz_bit_length(uint32_t x) { return x ? 32 - __builtin_clz(x) : 0; }

// first l >= from with in[p + l] != in[i + l], at most `limit` (p < i, i + limit <= n)
__device__ __forceinline__ uint32_t lz_match_end(const uint8_t* in, uint32_t p, uint32_t i, uint32_t from, uint32_t limit) {
  uint32_t l = from;
  while (l + 8 <= limit) {
    const unsigned long long a = *(const lz_u64u*)(in + p + l), b = *(const lz_u64u*)(in + i + l);
    if (a != b) return l + (uint32_t)(__builtin_ctzll(a ^ b) >> 3);
    l += 8;
  }
  while (l < limit && in[p + l] == in[i + l]) ++l;
  return l;
}

// The search LZBuffer::fill makes at position i, for "no literals pending" (r1) and "literals pending" (r0):
// x = offset, y = length | literals in front << 16 | 1 << 31 when a match is taken, 0 when the position is a literal.
__device__ __forceinline__ void lz_search(const uint8_t* in, uint32_t n, const uint32_t* sa, const uint32_t* rank, uint32_t i,
                                          const LzBlock& B, uint2& r0, uint2& r1) {
  const uint32_t mask = (1u << B.checkbits) - 1u;
  const uint32_t lim = n - i < kLzMaxMatch ? n - i : kLzMaxMatch;
  uint32_t blen[2], bp[2], blit[2];
  int bscore[2];
  bool alive[2];
  for (int v = 0; v < 2; ++v) { blen[v] = B.min_match - 1u; bp[v] = 0; blit[v] = 0; bscore[v] = 0; alive[v] = true; }
  for (uint32_t h = 0; h <= B.lookahead && (alive[0] || alive[1]); ++h) {
    // (the reference keeps the inverse array for one aligned window of 2^checkbits positions: a look-ahead that leaves the
    // window of i finds nothing)
    if (h + i >= n || ((h + i) & ~mask) != (i & ~mask)) continue;
    const uint32_t q = rank[h + i] - 1u;
    for (int dir = -1; dir <= 1; dir += 2) {
      bool on[2] = {alive[0], alive[1]};
      for (uint32_t k = 1; k <= B.bucket && (on[0] || on[1]); ++k) {
        const uint32_t at_q = q + (uint32_t)(dir * (int)k);
        if (at_q >= n) break;                                    // (past either end of the array: so is every later k)
        const uint32_t p = sa[at_q] - h;
        if (p >= i) continue;                                    // a later suffix (or one that starts inside the look-ahead)
        const uint32_t l = h < lim ? lz_match_end(in, p, i, h, lim) : h;
        uint32_t l1 = h;
        while (l1 > 0 && in[p + l1 - 1] == in[i + l1 - 1]) --l1;
        const int base = (int)(l - l1) * 8 - lz_bit_length(i - p) - 11;
        for (int v = 0; v < 2; ++v) {
          if (!on[v]) continue;
          int score = base - ((v == 1 && l1 > 0) ? 4 : 0);
          for (uint32_t a = 0; a < h; ++a) score = score * 5 / 8;
          if (score > bscore[v]) { blen[v] = l; bp[v] = p; blit[v] = l1; bscore[v] = score; }
          if (l < blen[v] || l < B.min_match || l > 255u) on[v] = false;
        }
      }
    }
    for (int v = 0; v < 2; ++v)
      if (alive[v] && (bscore[v] <= 0 || blen[v] < B.min_match)) alive[v] = false;
  }
  uint2 r[2];
  for (int v = 0; v < 2; ++v) {
    const uint32_t off = i - bp[v];
    const uint32_t need = B.min_match + (B.kind == 2 ? (uint32_t)(off >= (1u << 16)) + (uint32_t)(off >= (1u << 24)) : 0u);
    const bool take = off > 0 && bscore[v] > 0 && blen[v] - blit[v] >= need;
    r[v].x = take ? off : 0u;
    r[v].y = take ? ((blen[v] - blit[v]) | (blit[v] & kLzMaxLookahead) << 16 | 1u << 31) : 0u;     // (the engine refuses a larger look-ahead: engine_sort_preprocess)
  }
  r0 = r[0];
  r1 = r[1];
}

__device__ __forceinline__ void lz77_search_body(const uint8_t* in_all, const uint32_t* sa_all, const uint32_t* rank_all, const uint16_t* blk,
                                                 const LzBlock* blocks, uint64_t total, uint4* res) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const LzBlock B = blocks[blk[e]];
  if (B.kind != 1 && B.kind != 2) return;
  uint2 r0, r1;
  lz_search(in_all + B.off, B.n, sa_all + B.off, rank_all + B.off, (uint32_t)(e - B.off), B, r0, r1);
  uint4 r;
  r.x = r0.x; r.y = r0.y; r.z = r1.x; r.w = r1.y;
  res[e] = r;
}

// One wavefront per block (64 threads per workgroup).  counts[b] may exceed tok_cap: the list is incomplete then.
__device__ __forceinline__ void lz77_walk_body(const LzBlock* blocks, const uint4* res, LzTok* toks, uint32_t* counts) {
  const uint32_t b = blockIdx.x;
  const int lane = (int)(threadIdx.x & 63u);
  const LzBlock B = blocks[b];
  if (B.kind != 1 && B.kind != 2) { if (lane == 0) counts[b] = 0; return; }
  const uint4* r = res + B.off;
  uint32_t i = 0, lit = 0, nt = 0;
  while (i < B.n) {
    const uint32_t base = i;
    uint4 e;
    e.x = e.y = e.z = e.w = 0;
    if (base + (uint32_t)lane < B.n) e = r[base + (uint32_t)lane];
    // positions of this chunk where a match is taken although literals are pending
    const unsigned long long pending_takes = __builtin_amdgcn_ballot_w64((e.y >> 31) != 0u);
    while (i < B.n && i - base < 64u) {
      const int src = (int)(i - base);
      if (lit != 0) {
        // a run of literals in one step: up to the next such position, the end of the chunk or of the block, or the flush
        const unsigned long long ahead = pending_takes >> src;
        uint32_t d = ahead ? (uint32_t)__builtin_ctzll(ahead) : 64u - (uint32_t)src;
        d = d < B.n - i ? d : B.n - i;
        d = d < kLzMaxLiteral - lit ? d : kLzMaxLiteral - lit;
        if (d) {
          i += d;
          lit += d;
          if (lit >= kLzMaxLiteral) lit = 0;
          continue;
        }
      }
      const bool none = lit == 0;                                    // (the same in every lane)
      const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)(none ? e.z : e.x), src);
      const uint32_t y = (uint32_t)__builtin_amdgcn_readlane((int)(none ? e.w : e.y), src);
      if (y >> 31) {
        const uint32_t len = y & 0xFFFFu, front = (y >> 16) & 0xFFu;
        if (lane == 0 && nt < B.tok_cap) {
          LzTok t;
          t.i = i; t.off = x; t.len = len; t.blit = front;
          toks[B.tok_off + nt] = t;
        }
        ++nt;
        lit = 0;
        i += front + len;
      } else {
        ++lit;
        ++i;
        if (lit >= kLzMaxLiteral) lit = 0;                           // (flushed: LZBuffer::fill)
      }
    }
  }
  if (lane == 0) counts[b] = nt;
}

// BWT as preprocess_block lays it out: out[0] = last byte, out[j + 1] = the byte in front of suffix sa[j] (255 for the whole
// string, whose 1-based index goes to idx[b]); block b's n + 1 bytes start at out_all + off + b.
__device__ __forceinline__ void bwt_emit_body(const uint8_t* in_all, const uint32_t* sa_all, const uint16_t* blk, const LzBlock* blocks,
                                              uint64_t total, uint8_t* out_all, uint32_t* idx) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const uint32_t b = blk[e];
  const LzBlock B = blocks[b];
  if (B.kind != 3) return;
  const uint32_t j = (uint32_t)(e - B.off);
  const uint8_t* in = in_all + B.off;
  uint8_t* out = out_all + B.off + b;
  const uint32_t s = sa_all[e];
  out[j + 1] = s ? in[s - 1] : (uint8_t)255;
  if (!s) idx[b] = j + 1;
  if (j == 0) out[0] = in[B.n - 1];
}

}  // namespace zpq
