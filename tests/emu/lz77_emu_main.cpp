// TEST INFRASTRUCTURE -- the pre-processor kernels behind the suffix sort (zpaq_amd/csrc/device/lz77_kernel.h) on the
// host-side wavefront emulator (wave_emu.h), against the host's own parse.
//
//   lz77_emu <kind> <min_match> <lookahead> <bucket> <checkbits> <out_prefix> <input> [<input> ...]
//
// kind 1 / 2: LZ77 (bit-packed / byte-aligned codes) -> <out_prefix>.<k> = block k's token list (16 bytes per match);
// kind 3: BWT -> <out_prefix>.<k> = the n + 5 bytes preprocess_block makes.  The suffix array comes from the library's host
// sorter (zpq_suffix_array_host), the rank array is its inverse + 1: what device/sa_kernels.hip leaves behind.
#include "wave_emu.h"

#include <string>
#include <vector>

#include "lz77_kernel.h"
#include "zpaq_amd.h"

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

struct Args {
  const uint8_t* in_all;
  const uint32_t *sa_all, *rank_all;
  const uint16_t* blk;
  const zpq::LzBlock* blocks;
  uint64_t total;
  uint4* res;
  zpq::LzTok* toks;
  uint32_t* counts;
  uint8_t* bwt_out;
  uint32_t* idx;
};

void search_thunk(void* p) { Args* a = (Args*)p; zpq::lz77_search_body(a->in_all, a->sa_all, a->rank_all, a->blk, a->blocks, a->total, a->res); }
void walk_thunk(void* p) { Args* a = (Args*)p; zpq::lz77_walk_body(a->blocks, a->res, a->toks, a->counts); }
void bwt_thunk(void* p) { Args* a = (Args*)p; zpq::bwt_emit_body(a->in_all, a->sa_all, a->blk, a->blocks, a->total, a->bwt_out, a->idx); }

}  // namespace

int main(int argc, char** argv) {
  if (argc < 8) { fprintf(stderr, "usage: lz77_emu <kind> <min_match> <lookahead> <bucket> <checkbits> <out_prefix> <input>...\n"); return 2; }
  const uint32_t kind = (uint32_t)atoi(argv[1]), min_match = (uint32_t)atoi(argv[2]), lookahead = (uint32_t)atoi(argv[3]),
                 bucket = (uint32_t)atoi(argv[4]), checkbits = (uint32_t)atoi(argv[5]);
  const std::string prefix = argv[6];
  const unsigned nb = (unsigned)(argc - 7);
  std::vector<zpq::LzBlock> blocks(nb);
  std::vector<uint8_t> in_all;
  std::vector<uint32_t> sa_all, rank_all;
  std::vector<uint16_t> blk;
  uint64_t ntok = 0;
  for (unsigned b = 0; b < nb; ++b) {
    const std::vector<uint8_t> in = slurp(argv[7 + b]);
    zpq::LzBlock& B = blocks[b];
    memset(&B, 0, sizeof B);
    B.off = in_all.size();
    B.n = (uint32_t)in.size();
    B.kind = in.empty() ? 0u : kind;
    B.min_match = min_match; B.lookahead = lookahead; B.bucket = bucket; B.checkbits = checkbits;
    B.tok_off = ntok;
    B.tok_cap = kind == 3 ? 0u : B.n / (min_match ? min_match : 1u) + 2u;
    ntok += B.tok_cap;
    std::vector<uint32_t> sa(in.size() + 1), rank(in.size() + 1);
    if (zpq_suffix_array_host(in.data(), B.n, sa.data()) != 0) { fprintf(stderr, "suffix array: %s\n", zpq_last_error()); return 2; }
    for (uint32_t j = 0; j < B.n; ++j) rank[sa[j]] = j + 1;
    in_all.insert(in_all.end(), in.begin(), in.end());
    sa_all.insert(sa_all.end(), sa.begin(), sa.begin() + B.n);
    rank_all.insert(rank_all.end(), rank.begin(), rank.begin() + B.n);
    blk.insert(blk.end(), B.n, (uint16_t)b);
  }
  const uint64_t total = in_all.size();
  in_all.resize(total + 64);                 // (the engine's input buffer is padded as well; nothing may read it)
  std::vector<uint4> res(total + 1);
  std::vector<zpq::LzTok> toks(ntok + 1);
  std::vector<uint32_t> counts(nb, 0), idx(nb, 0);
  std::vector<uint8_t> bwt(total + nb + 1, 0);
  Args a{in_all.data(), sa_all.data(), rank_all.data(), blk.data(), blocks.data(), total, res.data(), toks.data(), counts.data(), bwt.data(), idx.data()};
  const unsigned wgs = (unsigned)((total + 255) / 256);
  if (kind == 1 || kind == 2) {
    for (unsigned wg = 0; wg < wgs; ++wg) emu::run_workgroup(search_thunk, &a, 256, wg);
    for (unsigned b = 0; b < nb; ++b) emu::run_workgroup(walk_thunk, &a, 64, b);
  } else {
    for (unsigned wg = 0; wg < wgs; ++wg) emu::run_workgroup(bwt_thunk, &a, 256, wg);
  }
  for (unsigned b = 0; b < nb; ++b) {
    const std::string path = prefix + "." + std::to_string(b);
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); return 2; }
    const zpq::LzBlock& B = blocks[b];
    if (kind == 3) {
      if (B.n) {
        fwrite(bwt.data() + B.off + b, 1, (size_t)B.n + 1, f);
        uint32_t x = idx[b];
        for (int k = 0; k < 4; ++k) { fputc((int)(x & 255u), f); x >>= 8; }
      }
    } else {
      if (counts[b] > B.tok_cap) { fprintf(stderr, "block %u: %u tokens for %u slots\n", b, counts[b], B.tok_cap); return 3; }
      fwrite(toks.data() + B.tok_off, 16, counts[b], f);
    }
    fclose(f);
    printf("block %u n %u tokens %u\n", b, B.n, counts[b]);
  }
  return 0;
}
