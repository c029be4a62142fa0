// TEST INFRASTRUCTURE -- driver of the host-side wavefront emulator (see wave_emu.h).
//
//   emu_run enc|dec <waves> <header.bin> <out_cap> <out_prefix> <input> [<input> ...]
//
// <waves> = wavefronts (= blocks) per workgroup the kernel was generated for.
//
// Runs the generated specialised kernel (compiled into this executable from the text
// zpq_plan_spec_source returns) over the given inputs, one ZPAQ block per wavefront,
// `waves` blocks per workgroup, and writes <out_prefix>.<k> for block k.  The plan, the
// arena layout and the constant tables come from libzpaq_amd.so's host code -- the same
// objects the engine uploads to the GPU; the arena is initialised here the way
// init_arena_kernel does it (device/kernels.hip).
#include "guard_alloc.h"
#include "wave_emu.h"

#include <cstring>
#include <string>
#include <vector>

#include "layout.h"
#include "zpaq_amd.h"

extern "C" void zpq_spec_encode(const zpq::BlockJob* jobs, zpq::BlockResult* res, unsigned nblocks,
                                const zpq::DeviceTables* tb);
extern "C" void zpq_spec_decode(const zpq::BlockJob* jobs, zpq::BlockResult* res, unsigned nblocks,
                                const zpq::DeviceTables* tb);
#ifdef ZPQ_EMU_DUAL
extern "C" void zpq_spec_decode2(const zpq::BlockJob* jobs, zpq::BlockResult* res, unsigned nblocks,
                                 const zpq::DeviceTables* tb);
#endif
#ifdef ZPQ_EMU_TEAM
extern "C" void zpq_spec_decode3(const zpq::BlockJob* jobs, zpq::BlockResult* res, unsigned nblocks,
                                 const zpq::DeviceTables* tb);
#endif

namespace {

std::vector<uint8_t> slurp(const char* path) {
  std::vector<uint8_t> v;
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  uint8_t buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
  fclose(f);
  return v;
}

struct Launch {
  bool dec;
  bool dual;
  const zpq::BlockJob* jobs;
  zpq::BlockResult* res;
  unsigned nblocks;
  const zpq::DeviceTables* tb;
};

void kernel_thunk(void* p) {
  Launch* l = (Launch*)p;
#ifdef ZPQ_EMU_DUAL
  if (l->dual) { zpq_spec_decode2(l->jobs, l->res, l->nblocks, l->tb); return; }
#endif
#ifdef ZPQ_EMU_TEAM
  if (l->dual) { zpq_spec_decode3(l->jobs, l->res, l->nblocks, l->tb); return; }
#endif
  if (l->dec) zpq_spec_decode(l->jobs, l->res, l->nblocks, l->tb);
  else zpq_spec_encode(l->jobs, l->res, l->nblocks, l->tb);
}

// init_arena_kernel's statement (kernels.hip), over an arena that starts DIRTY: the engine's arenas hold the previous batch's
// state, so whatever the fill leaves out is garbage there -- 16-byte stores over sg.bytes >> 4 units, like the kernel
// (round 6: a 4-byte H array -- hh = 0 -- fell through the fill and the GPU coded with the previous batch's H[0])
void init_arena(uint8_t* arena, const uint8_t* blob, const zpq::DeviceTables& tb) {
  const zpq::PlanHeader* ph = (const zpq::PlanHeader*)blob;
  const zpq::Segment* segs = (const zpq::Segment*)(blob + ph->off_seg);
  memset(arena, 0xC3, ph->arena_bytes);
  for (uint32_t s = 0; s < ph->nseg; ++s) {
    const zpq::Segment& sg = segs[s];
    uint32_t* dst = (uint32_t*)(arena + sg.off);
    const uint64_t n = (sg.bytes >> 4) * 4;
    switch (sg.kind) {
      case zpq::F_ZERO: for (uint64_t i = 0; i < n; ++i) dst[i] = 0; break;
      case zpq::F_U32: for (uint64_t i = 0; i < n; ++i) dst[i] = sg.value; break;
      case zpq::F_SSE: for (uint64_t i = 0; i < n; ++i) dst[i] = tb.sse_row[i & 31] | sg.value; break;
      case zpq::F_ICM: for (uint64_t i = 0; i < n; ++i) dst[i] = tb.icm_init[i]; break;
      case zpq::F_ISSE: for (uint64_t i = 0; i < n; ++i) dst[i] = tb.isse_init[i]; break;
      case zpq::F_MATCHBUF: for (uint64_t i = 0; i < n; ++i) dst[i] = i == 0 ? 1u : 0u; break;
      default: fprintf(stderr, "emu_run: unknown segment kind %u\n", sg.kind); exit(2);
    }
  }
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 7) { fprintf(stderr, "usage: emu_run enc|dec <waves> <header.bin> <out_cap> <out_prefix> <input>...\n"); return 2; }
  // "dec2": two blocks per wavefront (spec_dual_kernel.h); "dec3": the lockstep decoder (spec_team_kernel.h), <waves> = its
  // threads per workgroup / 64; both address the blocks of a workgroup relative to its first arena: one contiguous pool
  const bool team = !strcmp(argv[1], "dec3");
  const bool dual = team || !strcmp(argv[1], "dec2");
  const bool dec = dual || !strcmp(argv[1], "dec");
  const unsigned waves = (unsigned)atoi(argv[2]);
  const std::vector<uint8_t> header = slurp(argv[3]);
  const uint32_t out_cap = (uint32_t)strtoul(argv[4], nullptr, 10);
  const std::string prefix = argv[5];
  const unsigned nb = (unsigned)(argc - 6);

  zpq_plan* plan = nullptr;
  if (zpq_plan_create(header.data(), header.size(), &plan) != 0) { fprintf(stderr, "plan: %s\n", zpq_last_error()); return 2; }
  size_t blob_len = 0;
  const uint8_t* blob = zpq_plan_blob(plan, &blob_len);
  const zpq::PlanHeader* ph = (const zpq::PlanHeader*)blob;

  static zpq::DeviceTables tb;
  int32_t dt2k[256];
  if (!zpq_table(1, tb.stretch, sizeof tb.stretch) || !zpq_table(0, tb.squash, sizeof tb.squash) ||
      !zpq_table(2, tb.dt, sizeof tb.dt) || !zpq_table(3, dt2k, sizeof dt2k) || !zpq_table(4, tb.ns, sizeof tb.ns) ||
      !zpq_table(5, tb.icm_init, sizeof tb.icm_init) || !zpq_table(6, tb.isse_init, sizeof tb.isse_init) ||
      !zpq_table(7, tb.sse_row, sizeof tb.sse_row) || !zpq_table(8, tb.stretch_cb, sizeof tb.stretch_cb) ||
      !zpq_table(9, tb.stretch_top, sizeof tb.stretch_top)) { fprintf(stderr, "tables unavailable\n"); return 2; }
  memcpy(tb.dt2k, dt2k, sizeof dt2k);

  const unsigned njobs = nb;
  std::vector<std::vector<uint8_t>> ins(njobs), outs(njobs);
  std::vector<zpq::BlockJob> jobs(njobs);
  std::vector<zpq::BlockResult> res(njobs);
  // every block's arena and input between inaccessible pages (guard_alloc.h); ZPQ_EMU_GUARD=0: one contiguous arena pool
  // like the engine's, so that neighbouring blocks' arenas touch
  // (the two-blocks-per-wavefront kernel addresses the second block's arena relative to the first's: back to back, as the
  // engine lays them out; the pool as a whole sits between guard pages then)
  const bool guard = emu::guard_on() && !dual;
  uint8_t* pool = guard ? nullptr : (dual && emu::guard_on() ? emu::guard_alloc((size_t)njobs * ph->arena_bytes, 256, 0)
                                                             : (uint8_t*)calloc((size_t)njobs, ph->arena_bytes));
  if (!guard && !pool) { fprintf(stderr, "arena pool: out of memory\n"); return 2; }
  for (unsigned b = 0; b < njobs; ++b) {
    if (b < nb) ins[b] = slurp(argv[6 + b]);
    outs[b].assign((size_t)out_cap + 64, 0xEE);
    uint8_t* const arena = guard ? emu::guard_alloc(ph->arena_bytes, 256, 0) : pool + (size_t)b * ph->arena_bytes;
    init_arena(arena, blob, tb);
    memset(&jobs[b], 0, sizeof(jobs[b]));
    jobs[b].plan = blob;
    jobs[b].arena = arena;
    jobs[b].in = ins[b].data();
    if (guard) {      // the engine pads every input to a multiple of 64 bytes: that much may be read, not more
      uint8_t* gin = emu::guard_alloc(ins[b].size(), 64, 0);
      if (!ins[b].empty()) memcpy(gin, ins[b].data(), ins[b].size());
      jobs[b].in = gin;
    }
    jobs[b].out = outs[b].data();
    jobs[b].in_len = (uint32_t)ins[b].size();
    jobs[b].out_cap = b < nb ? out_cap : 0;
    jobs[b].res_slot = b;
    res[b] = zpq::BlockResult{0, 0, -1, 0};
  }
  Launch l{dec, dual, jobs.data(), res.data(), nb, &tb};
  const unsigned per_wg = dual ? 8 : waves;      // blocks per workgroup
  for (unsigned wg = 0; wg < (nb + per_wg - 1) / per_wg; ++wg) emu::run_workgroup(kernel_thunk, &l, team ? 64 * waves : (dual ? 256 : 64 * waves), wg);
  for (unsigned b = 0; b < nb; ++b) {
    // guard bytes past the capacity must be untouched
    for (unsigned k = 0; k < 64; ++k)
      if (outs[b][(size_t)out_cap + k] != 0xEE) { fprintf(stderr, "block %u wrote past its output capacity\n", b); return 3; }
    const std::string path = prefix + "." + std::to_string(b);
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); return 2; }
    const uint32_t n = res[b].out_len < out_cap ? res[b].out_len : out_cap;
    fwrite(outs[b].data(), 1, n, f);
    fclose(f);
    printf("block %u status %d out_len %u consumed %u steps %u\n", b, res[b].status, res[b].out_len, res[b].consumed,
           res[b].steps);
  }
  printf("cross_lane_ops %lu\n", emu::cross_lane_ops());
  if (!(dual && emu::guard_on())) free(pool);
  zpq_plan_destroy(plan);
  return 0;
}
