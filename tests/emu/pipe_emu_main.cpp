// TEST INFRASTRUCTURE -- host-side run of the PIPELINED ENCODER (zpaq_amd/csrc/device/pipe_kernel.h) under the
// wavefront emulator (wave_emu.h).
//
//   pipe_emu_run <header.bin> <out_cap> <out_prefix> <input> [<input> ...]
//
// Runs the six generated kernels step by step over the inputs (one ZPAQ block each) exactly as the engine
// launches them on the GPU -- same grids, same stream buffer, same arenas -- except that inside a step the
// workgroups run in REVERSE dataflow order (consumers before producers): a unit that wrongly depended on data
// produced in the same step would read a stale slot here and fail the comparison with the oracle.
#include "guard_alloc.h"
#include "wave_emu.h"

#include <cstring>
#include <string>
#include <vector>

#include "layout.h"
#include "zpaq_amd.h"

extern "C" void zpq_pipe_hcomp(zpq::PipeArgs a);
extern "C" void zpq_pipe_rows(zpq::PipeArgs a);
extern "C" void zpq_pipe_light(zpq::PipeArgs a);
extern "C" void zpq_pipe_icm(zpq::PipeArgs a);
extern "C" void zpq_pipe_isse(zpq::PipeArgs a);
extern "C" void zpq_pipe_mix(zpq::PipeArgs a);
extern "C" void zpq_pipe_persist(zpq::PipeArgs a);
extern "C" void zpq_pipe_repack(zpq::PipeArgs a);

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

struct Launch { int which; zpq::PipeArgs a; };
void kernel_thunk(void* p) {
  Launch* l = (Launch*)p;
  switch (l->which) {
    case 0: zpq_pipe_hcomp(l->a); break;
    case 1: zpq_pipe_rows(l->a); break;
    case 2: zpq_pipe_light(l->a); break;
    case 3: zpq_pipe_icm(l->a); break;
    case 4: zpq_pipe_isse(l->a); break;
    case 6: zpq_pipe_persist(l->a); break;
    case 7: zpq_pipe_repack(l->a); break;
    default: zpq_pipe_mix(l->a); break;
  }
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
      default: fprintf(stderr, "pipe_emu_run: unknown segment kind %u\n", sg.kind); exit(2);
    }
  }
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 8) { fprintf(stderr, "usage: pipe_emu_run <header.bin> <out_cap> <out_prefix> <mode> <chunk> <group> <input>...\n"); return 2; }
  const std::vector<uint8_t> header = slurp(argv[1]);
  const uint32_t out_cap = (uint32_t)strtoul(argv[2], nullptr, 10);
  const std::string prefix = argv[3];
  const int mode_arg = atoi(argv[4]), chunk = atoi(argv[5]), group = atoi(argv[6]);
  const int mode = mode_arg & 15;
  const bool persist = (mode_arg & 16) != 0;       // the persistent launch (device/pipe_persist.h): ONE grid of live workgroups
  argv += 3; argc -= 3;                    // the inputs follow
  const unsigned nb = (unsigned)(argc - 4);

  zpq_plan* plan = nullptr;
  if (zpq_plan_create(header.data(), header.size(), &plan) != 0) { fprintf(stderr, "plan: %s\n", zpq_last_error()); return 2; }
  size_t blob_len = 0;
  const uint8_t* blob = zpq_plan_blob(plan, &blob_len);
  const zpq::PlanHeader* ph = (const zpq::PlanHeader*)blob;
  uint64_t lay[16];
  if (zpq_plan_pipe_layout_opts(plan, mode, chunk, group, lay) != 0) { fprintf(stderr, "layout: %s\n", zpq_last_error()); return 2; }
  const uint64_t group_bytes = lay[0];
  const unsigned C = (unsigned)lay[2], nlight = (unsigned)lay[3], nicm = (unsigned)lay[4], nisse = (unsigned)lay[5],
                 mixw = (unsigned)lay[6], hl = (unsigned)lay[7], maxlevel = (unsigned)lay[8], G = (unsigned)lay[9], nrows = (unsigned)lay[10],
                 mixt = (unsigned)lay[11], rowt = (unsigned)lay[12], lightt = (unsigned)lay[13];

  static zpq::DeviceTables tb;
  int32_t dt2k[256];
  if (!zpq_table(1, tb.stretch, sizeof tb.stretch) || !zpq_table(0, tb.squash, sizeof tb.squash) ||
      !zpq_table(2, tb.dt, sizeof tb.dt) || !zpq_table(3, dt2k, sizeof dt2k) || !zpq_table(4, tb.ns, sizeof tb.ns) ||
      !zpq_table(5, tb.icm_init, sizeof tb.icm_init) || !zpq_table(6, tb.isse_init, sizeof tb.isse_init) ||
      !zpq_table(7, tb.sse_row, sizeof tb.sse_row) || !zpq_table(8, tb.stretch_cb, sizeof tb.stretch_cb) ||
      !zpq_table(9, tb.stretch_top, sizeof tb.stretch_top)) { fprintf(stderr, "tables unavailable\n"); return 2; }
  memcpy(tb.dt2k, dt2k, sizeof dt2k);

  std::vector<std::vector<uint8_t>> ins(nb), outs(nb);
  std::vector<zpq::BlockJob> jobs(nb);
  std::vector<zpq::BlockResult> res(nb);
  // arenas, inputs and the stream buffer between inaccessible pages (guard_alloc.h); ZPQ_EMU_GUARD=0: plain allocations
  const bool guard = emu::guard_on();
  uint8_t* pool = guard ? nullptr : (uint8_t*)calloc((size_t)nb, ph->arena_bytes);
  const unsigned ngroups = (nb + G - 1) / G;
  // 0xA5 everywhere: a unit reading a stream slot nobody wrote gets garbage, not zeros
  uint8_t* pipe = guard ? emu::guard_alloc((size_t)ngroups * group_bytes, 256, 0xA5) : (uint8_t*)malloc((size_t)ngroups * group_bytes);
  if ((!guard && !pool) || !pipe) { fprintf(stderr, "out of memory\n"); return 2; }
  if (!guard) memset(pipe, 0xA5, (size_t)ngroups * group_bytes);
  unsigned maxlen = 0;
  for (unsigned b = 0; b < nb; ++b) {
    ins[b] = slurp(argv[4 + b]);
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
    jobs[b].out_cap = out_cap;
    jobs[b].res_slot = b;
    res[b] = zpq::BlockResult{0, 0, -1, 0};
    if (ins[b].size() > maxlen) maxlen = (unsigned)ins[b].size();
  }
  const unsigned nchunks = maxlen ? (maxlen + C - 1) / C : 1;
  {
    // MIX tables the chain keeps as packed rows are rewritten from Predictor::init's pattern, as the engine does before the first launch
    Launch l{7, zpq::PipeArgs{jobs.data(), res.data(), nb, &tb, pipe, 0, 0u}};
    for (unsigned b = 0; b < nb; ++b) emu::run_workgroup(kernel_thunk, &l, 256, b);
  }
  if (persist) {
    const unsigned wpg = (unsigned)lay[14], waves = (unsigned)(lay[15] & 0xFFFF), nunit = (unsigned)(lay[15] >> 16);
    if (!wpg) { fprintf(stderr, "pipe_emu_run: this chain has no persistent launch\n"); return 2; }
    std::vector<uint32_t> prog((size_t)ngroups * nunit, 0), gchunks(ngroups, 1), ctl(4, 0);
    for (unsigned g = 0; g < ngroups; ++g) {
      unsigned ml = 0;
      for (unsigned b = g * G; b < nb && b < (g + 1) * G; ++b) if (ins[b].size() > ml) ml = (unsigned)ins[b].size();
      gchunks[g] = ml ? (ml + C - 1) / C : 1;
    }
    Launch l{6, zpq::PipeArgs{jobs.data(), res.data(), nb, &tb, pipe, 0, 0u}};
    l.a.prog = prog.data(); l.a.group_chunks = gchunks.data(); l.a.ctl = ctl.data();
    l.a.group0 = 0; l.a.ngroups_here = ngroups; l.a.timeout_ticks = 200000;
    l.a.arrive_need = ngroups * wpg; l.a.arrive_ticks = 200000;      // (the arrival handshake: every workgroup of the grid is alive here)
    l.a.spread = ngroups > 1 ? 2 : 1;          // (the device spreads over its 8 XCDs: here two, so that the mapping is exercised)
    emu::run_grid(kernel_thunk, &l, 64 * waves, ngroups * wpg, 0);
    if (ctl[0]) { fprintf(stderr, "pipe_emu_run: the persistent launch aborted (slot %u, chunk %u)\n", ctl[1], ctl[2]); return 4; }
  } else {
  const unsigned grids[6] = {(nb + hl - 1) / hl, nrows * ngroups, nlight * ngroups, nicm * ngroups, nisse * ngroups, mixw * ngroups};
  for (unsigned step = 0; step < nchunks + maxlevel; ++step) {
    Launch l{0, zpq::PipeArgs{jobs.data(), res.data(), nb, &tb, pipe, (int)step, 0u}};
    for (int which = 5; which >= 0; --which) {
      l.which = which;
      for (unsigned wg = grids[which]; wg-- > 0;) emu::run_workgroup(kernel_thunk, &l, which == 0 ? 64 : (which == 5 ? mixt : (which == 1 ? rowt : (which == 2 ? lightt : G))), wg);
    }
  }
  }
  for (unsigned b = 0; b < nb; ++b) {
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
  free(pool);
  if (!guard) free(pipe);
  zpq_plan_destroy(plan);
  return 0;
}
