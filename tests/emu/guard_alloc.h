// TEST INFRASTRUCTURE: buffers for the emulated kernels with an inaccessible page in front and behind, placed so that the
// buffer ENDS where the rear page starts (up to `align - 1` bytes of slack).  A load or store outside a block's arena, its
// input, its output or the stream buffer stops the emulator with SIGSEGV instead of reading a neighbour -- on the GPU the
// same access lands in another block's state, or past the allocation (a memory fault that takes the queue down).
// ZPQ_EMU_GUARD=0: plain allocations, the arenas of the blocks back to back like the engine's pool.
#pragma once
#include <sys/mman.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace emu {

inline bool guard_on() {
  const char* e = getenv("ZPQ_EMU_GUARD");
  return !(e && e[0] == '0');
}

// n bytes, start aligned to `align` (a power of two <= 4096), filled with `fill`
inline uint8_t* guard_alloc(size_t n, size_t align, int fill) {
  const size_t pg = 4096;
  const size_t span = (n + align - 1) / align * align;
  const size_t body = (span + pg - 1) / pg * pg + (span == 0 ? pg : 0);
  uint8_t* m = (uint8_t*)mmap(nullptr, body + 2 * pg, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (m == (uint8_t*)MAP_FAILED) { perror("mmap"); exit(2); }
  if (mprotect(m + pg, body, PROT_READ | PROT_WRITE) != 0) { perror("mprotect"); exit(2); }
  uint8_t* p = m + pg + body - span;
  if (fill) memset(p, fill, span);
  return p;
}

}  // namespace emu
