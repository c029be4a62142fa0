/* include/zpaq_amd.h -- the drop-in boundary: a thin C ABI over the MI355X engine.
 *
 * Plain C types, caller-owned buffers, integer status codes, no exceptions and
 * no torch types across this line.  Everything above it (the libzpaq-compatible
 * C++ classes in include/libzpaq.h, the Python mirror in zpaq_amd/) is host
 * plumbing; everything below it is HIP for gfx950.
 *
 * Each entry point names the reference interface it replaces (file:line into
 * zpaq 7.15's libzpaq.h / libzpaq.cpp).  The library is thread-safe: concurrent
 * callers are serialised on the device queue, mirroring how zpaq.cpp:1947 calls
 * compressBlock() from many threads.
 *
 * There is NO CPU fallback for the modelled path: if no gfx950 device/kernels
 * are available these functions return ZPQ_E_DEVICE, they never silently code
 * on the host.
 */
#ifndef ZPAQ_AMD_H
#define ZPAQ_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (per call and per block) ---- */
enum {
  ZPQ_OK = 0,
  ZPQ_E_NOMEM = 1,     /* "Out of memory" (libzpaq.h:925)                         */
  ZPQ_E_CORRUPT = 2,   /* "archive corrupted" (libzpaq.cpp:2108)                 */
  ZPQ_E_OVERFLOW = 3,  /* caller's output buffer too small                        */
  ZPQ_E_HEADER = 4,    /* bad COMP/HCOMP header (libzpaq.cpp:887-931, 1776-1846) */
  ZPQ_E_VM = 5,        /* "ZPAQL execution error" (libzpaq.cpp:1265)             */
  ZPQ_E_EOF = 6,       /* "unexpected end of file" (libzpaq.cpp:2120)            */
  ZPQ_E_DEVICE = 7,    /* no usable gfx950 device / HIP runtime error            */
  ZPQ_E_UNSUPPORTED = 8, /* valid ZPAQ feature outside this build's hot-path scope */
  ZPQ_E_ARG = 9
};

/* Thread-local text for the last failing call on this thread. */
const char* zpq_last_error(void);
const char* zpq_version(void);

/* ---- device ---- */
/* device >= 0: bind the engine to that HIP device (one process per GPU; without a call LOCAL_RANK picks it).
 * device == -1: drive EVERY visible GPU from this process (or the list in ZPAQ_AMD_DEVICES = "all" | "0,2,3"):
 * one engine per device, host-buffer batches (zpq_encode_batch, zpq_decode_batch, zpq_compress_blocks,
 * zpq_decompress, the libzpaq C++ classes) are sharded over them; the device-resident entry points keep using
 * the first one.  Idempotent.  Uploads the predictor's constant tables (squash/stretch/dt/
 * dt2k/state table; Predictor::init libzpaq.cpp:1731-1761) after verifying the
 * reference's two table checksums (1759-1760) on the host. */
int zpq_init(int device);
int zpq_device_count(void);
/* Engines zpq_init() configured: 1, or one per device named by zpq_init(-1) / ZPAQ_AMD_DEVICES (a device named twice gets two). */
int zpq_engine_count(void);
/* How a host-buffer batch of n blocks is split when the engine drives several GPUs (zpq_init(-1) or
 * ZPAQ_AMD_DEVICES=all|0,1,..): shard k of `parts` codes blocks [lo, hi) -- contiguous ranges, so archive order is
 * kept; one engine and one host thread per device, no collective (blocks are independent, libzpaq.h:57-59). */
void zpq_shard_range(uint64_t n, uint32_t parts, uint32_t k, uint64_t* lo, uint64_t* hi);
void zpq_shutdown(void);
/* Cap on device bytes the engine may hold for model state (default: 85% of
 * free HBM at init).  Batches needing more are run in several residency waves. */
int zpq_set_state_budget(uint64_t bytes);
/* Select the coding kernel: 0 = auto (best available for each plan: the pipelined encoder for compression, the
 * per-header specialised wavefront kernel for decompression), 1 = generic one-lane kernel, 2 = generic
 * wave-parallel kernel, 3 = per-header specialised wavefront kernel in both directions (one block per wavefront),
 * 4 = pipelined encoder (decompression as with 3), 5 = the decoder with two blocks per wavefront (compression as with 0),
 * 6 = the lockstep decoder (row / mixer wavefronts, the 8 blocks of a workgroup bit by bit together: what 0 picks for a
 * launch of more than 4 x CUs blocks whose chain it takes; compression as with 0).  3, 4, 5 and 6 fail with
 * ZPQ_E_UNSUPPORTED when the kernel cannot be built. */
int zpq_set_kernel(int which);

/* ---- model plan: a parsed block header + device arena layout ---- */
typedef struct zpq_plan zpq_plan;
/* header = hsize_lo hsize_hi hh hm ph pm n COMP.. 0 HCOMP.. 0 exactly as stored
 * in the archive.  Replaces ZPAQL::read (libzpaq.cpp:887) + the sizing half of
 * Predictor::init (1776-1846). */
int zpq_plan_create(const uint8_t* header, size_t hlen, zpq_plan** out);
void zpq_plan_destroy(zpq_plan*);
int zpq_plan_ncomp(const zpq_plan*);
/* ZPAQL::memory() (libzpaq.cpp:986-1006): what findBlock(&mem) reports. */
double zpq_plan_memory(const zpq_plan*);
/* Device bytes one in-flight block of this plan occupies. */
uint64_t zpq_plan_state_bytes(const zpq_plan*);
/* ALGORITHMIC model-state bytes moved per coded input byte,
 * excluding init and I/O: the roofline numerator. */
double zpq_plan_algo_bytes_per_byte(const zpq_plan*);

/* Per-header kernel specialisation (the GPU analogue of the reference's x86
 * JIT, libzpaq.cpp:3824-4583 / 3231-3811): HIP source generated for this
 * header, instantiating device/spec_kernel.h.  zpq_plan_spec_source returns the
 * text and its cache key (40 hex chars + NUL) so that a build step can
 * precompile it to <cache dir>/<key>.hsaco; at run time the engine loads that
 * file or falls back to hipRTC.  Needs no GPU. */
int zpq_plan_spec_source(const zpq_plan*, char* src, size_t cap, size_t* len, char key41[41]);
/* The same for the pipelined ENCODER of this header (device/pipe_kernel.h: one lane per block and component,
 * components connected by streams in HBM; used for compression whenever the chain supports it), and its
 * dataflow plan: out[0] bytes of stream buffer per group of blocks, [1] ring slots, [2] chunk bytes,
 * [3] units in the light kernel, [4] ICM maps, [5] ISSE maps, [6] MIX wavefronts per group, [7] blocks per HCOMP
 * workgroup, [8] highest dataflow level (a batch of L-byte blocks takes ceil(L / chunk) + out[8] steps),
 * [9] blocks per group (= threads per workgroup of every kernel but hcomp, which has 64), [10] ROW units,
 * [11] threads per workgroup of the mix kernel, [12] of the rows kernel, [13] of the light kernel.
 * The encoder has two shapes per chain: mode 0 "throughput" (a lane per block; batches that fill the GPU) and mode 1
 * "latency" (MIX / CM / MIX2 with a lane per bit position as well; the engine uses it for chains with at most 640 blocks
 * in the batch), mode 2 = mode 1 with 2048-byte steps (blocks of 128 KiB and more).  The plain calls give mode 0; the
 * _opts calls take the mode and, for tests, the chunk (bytes per step, 0 = the mode's own) and the group (blocks per
 * wavefront, 0 = 32). */
/* The decoder with two blocks per wavefront (device/spec_dual_kernel.h): chains of up to 32 components. */
int zpq_plan_spec_dual_source(const zpq_plan*, char* src, size_t cap, size_t* len, char key41[41]);
/* The lockstep decoder (device/spec_team_kernel.h): the 8 blocks of a workgroup advance bit by bit together, their ICM /
 * ISSE components on row wavefronts (16 lanes per block), everything else on mixer wavefronts (two blocks each).  Chains of
 * up to 32 components whose ISSEs are fed by the ICM / ISSE right before them. */
int zpq_plan_spec_team_source(const zpq_plan*, char* src, size_t cap, size_t* len, char key41[41]);
int zpq_plan_pipe_source(const zpq_plan*, char* src, size_t cap, size_t* len, char key41[41]);
int zpq_plan_pipe_layout(const zpq_plan*, uint64_t out[16]);
int zpq_plan_pipe_source_opts(const zpq_plan*, int mode, int chunk, int group, char* src, size_t cap, size_t* len, char key41[41]);
int zpq_plan_pipe_layout_opts(const zpq_plan*, int mode, int chunk, int group, uint64_t out[16]);
/* The same for a block's PCOMP post-processing program (device/pcomp_kernel.h: LZ77 / BWT / E8E9 inverses run one
 * lane per segment on the device when a batch has enough of them): code = the PCOMP bytes without their 2-byte
 * length, ph / pm = header bytes 4 and 5. */
int zpq_pcomp_source(const uint8_t* code, size_t codelen, int ph, int pm, char* src, size_t cap, size_t* len, char key41[41]);
/* Host post-processing (PostProcessor::write, libzpaq.cpp:2195-2241): 1 when this PCOMP program (bytes as for
 * zpq_pcomp_source) is one of those compressBlock's own methods generate, which the host runs as C++ translated at build time
 * (the counterpart of the reference's x86 JIT, libzpaq.cpp:3231-3811); 0 when it will be interpreted. */
int zpq_pcomp_is_translated(const uint8_t* code, size_t codelen, int ph, int pm);
/* Runs only the hipRTC compilation of that source (needs no GPU; nothing is loaded or cached):
 * returns the size of the gfx950 code object, or 0 with the compiler log in `log`. */
size_t zpq_plan_spec_jit(const zpq_plan*, char* log, size_t cap);
size_t zpq_plan_spec_dual_jit(const zpq_plan*, char* log, size_t cap);      /* the decoder with two blocks per wavefront */
size_t zpq_plan_spec_team_jit(const zpq_plan*, char* log, size_t cap);      /* the lockstep decoder */
/* Headers nobody prebuilt (level-5 chains whose periodic models depend on the data): compile the kernels of `n` plans
 * with hipRTC on up to `threads` host threads at once (0 = as many as the process may use, at most 16) -- the pipelined
 * encoder when decode == 0, the wavefront kernel otherwise -- into the code-object cache.  The engine does the same at
 * the start of every batch that brings more than one unseen header (at most ZPAQ_AMD_MAX_JIT = 64 per call); calling
 * it ahead of time (a caller that knows its methods) moves the cost out of the first batch.  Needs no GPU.
 * Returns the number of code objects compiled, or < 0. */
int zpq_precompile(const zpq_plan* const* plans, size_t n, int decode, int threads);
/* Introspection for tools and tests: the plan as the kernels see it (device/layout.h: PlanHeader,
 * CompDesc[n], Segment[nseg], HCOMP bytes).  The pointer stays valid until zpq_plan_destroy. */
const uint8_t* zpq_plan_blob(const zpq_plan*, size_t* len);
/* Which kernel will code this plan on the current device: 4 pipelined encoder (compression only), 3 specialised,
 * 2 generic wave, 1 generic one-lane.  note (optional) receives where the
 * specialised kernel came from ("cache:<key>" / "hiprtc") or why it is not used. */
int zpq_plan_kernel_kind(zpq_plan*, char* note, size_t cap);            /* compression */
int zpq_plan_kernel_kind2(zpq_plan*, int decode, char* note, size_t cap);
/* ... for a batch that holds nblocks blocks of this plan (the encoder's mode and the decoder's workgroup shape depend on it) */
int zpq_plan_kernel_kind3(zpq_plan*, int decode, uint32_t nblocks, char* note, size_t cap);
/* ... whose longest block has block_bytes bytes (latency shape: 2048-byte steps for blocks of 128 KiB and more) */
int zpq_plan_kernel_kind4(zpq_plan*, int decode, uint32_t nblocks, uint32_t block_bytes, char* note, size_t cap);
/* Directories used by the specialisation cache / hipRTC include path. */
const char* zpq_spec_cache_dir(void);
const char* zpq_spec_include_dir(void);

/* ---- the hot path: batches of independent blocks ---- */
/* Encoder::compress over in[b][0..in_len[b]) then EOS, for every block
 * (libzpaq.cpp:2419-2447 driving Predictor::predict/update 1854-2066 and
 * ZPAQL::run 1027).  `in[b]` is what the Compressor feeds the Encoder: PP header
 * byte(s) then data.  out[b] receives the coded bytes including the 4 EOS
 * flush bytes (not the 00 00 00 00 terminator).  status[b] is per block.
 * Host pointers; the call copies in, runs, copies out. */
int zpq_encode_batch(const zpq_plan* const* plans, const uint8_t* const* in,
                     const uint32_t* in_len, uint32_t nblocks,
                     uint8_t* const* out, const uint32_t* out_cap,
                     uint32_t* out_len, int32_t* status);

/* Decoder::decompress until EOS for every block (libzpaq.cpp:2127-2155).
 * in[b] = coded bytes starting at the first coded byte and including at least
 * the 4-zero terminator.  out[b] gets the decoded bytes (PP header included);
 * consumed[b] = coded bytes read (terminator included).  max_out[b] bounds the
 * decode ("decode first k bytes", Decompresser::decompress(n) 2315): decoding
 * stops early with status ZPQ_OK and consumed = 0 when max_out is reached
 * before EOS. */
int zpq_decode_batch(const zpq_plan* const* plans, const uint8_t* const* in,
                     const uint32_t* in_len, uint32_t nblocks,
                     uint8_t* const* out, const uint32_t* max_out,
                     uint32_t* out_len, uint32_t* consumed, int32_t* status);

/* Device-resident variants (inputs already in HBM; used by bench.py and by
 * callers that keep data on the GPU).  d_in/d_out are DEVICE pointers; block b
 * reads d_in + in_off[b] and writes d_out + out_off[b] (in_off/in_len/out_off/
 * out_cap are HOST arrays).  Per-block results land in the DEVICE array d_res.
 * `stream` is a hipStream_t (NULL = the engine's own stream).  One plan for the
 * whole batch, which must fit the state budget.  Of block b's output region
 * [out_off[b], out_off[b] + out_cap[b]) the first out_len bytes are the result;
 * what the rest holds afterwards is undefined (the coder of small batches stores
 * four bytes per coded bit and lets the next store overwrite what was not yet
 * final), nothing outside the region is written.  The call enqueues
 * init_arena + the coding kernel on `stream`; with timed!=0 it also brackets
 * them with hipEvents on that stream, waits, and makes the durations available
 * through zpq_last_timing(). */
typedef struct zpq_block_result {
  uint32_t out_len;    /* bytes produced                                   */
  uint32_t consumed;   /* decode: coded bytes read incl. terminator, 0 if stopped at max_out */
  int32_t status;      /* ZPQ_* per block                                  */
  uint32_t steps;      /* predict/update steps executed (coded bits)       */
} zpq_block_result;

int zpq_encode_device(const zpq_plan* plan, const void* d_in, const uint64_t* in_off,
                      const uint32_t* in_len, uint32_t nblocks, void* d_out,
                      const uint64_t* out_off, const uint32_t* out_cap,
                      zpq_block_result* d_res, void* stream, int timed);
int zpq_decode_device(const zpq_plan* plan, const void* d_in, const uint64_t* in_off,
                      const uint32_t* in_len, uint32_t nblocks, void* d_out,
                      const uint64_t* out_off, const uint32_t* max_out,
                      zpq_block_result* d_res, void* stream, int timed);
/* Same, with one plan PER BLOCK (plans[b]): blocks are grouped by plan internally and the groups
 * run concurrently on side streams; results keep the caller's block order.  decode = 0 / 1. */
int zpq_code_device_multi(int decode, const zpq_plan* const* plans, const void* d_in,
                          const uint64_t* in_off, const uint32_t* in_len, uint32_t nblocks,
                          void* d_out, const uint64_t* out_off, const uint32_t* cap,
                          zpq_block_result* d_res, void* stream, int timed);
/* Durations (ms, hipEvent) of the last timed call on this process: Predictor
 * init kernel and coding kernel(s); blocks = blocks they covered. */
int zpq_last_timing(float* init_ms, float* code_ms, uint32_t* blocks);
/* 1 when the pipelined encoder of the last timed call ran as persistent launches (one launch per chain for the whole
 * sequence, device/pipe_persist.h), 0 when it ran step by step (or the call had no pipelined group). */
int zpq_last_persistent(void);
/* When a persistent launch of the last timed call was given up (its workgroups did not become resident together -- something
 * else held compute units -- or a unit's watchdog fired) and the step kernels coded the batch instead: milliseconds from the
 * launch until the engine knew; 0 when none was given up.  The arrival handshake bounds it to ~2 x 20 ms
 * (ZPAQ_AMD_PERSIST_ARRIVE_MS) and leaves the model state untouched. */
double zpq_last_persist_abort_ms(void);
/* SHA-1 (libzpaq::SHA1, libzpaq.cpp:106-177) of n buffers ON THE DEVICE, one lane per buffer, 20 bytes each into
 * out20n.  zpq_compress_blocks uses the same kernel for blocks that reach the device unchanged (methods without
 * pre-processing): the digest for the segment trailer is computed beside the coder instead of on a host thread. */
int zpq_sha1_batch_device(const uint8_t* const* in, const uint32_t* len, uint32_t n, uint8_t* out20n);
/* Wall-clock phases (ms) of this process's last zpq_compress_blocks call -- the end-to-end path through the
 * drop-in API: out[0] total, [1] host front half (SHA-1, method expansion, header assembly), [2] device call
 * (staging + H2D + kernels + D2H), [3] archive stitching, [4] / [5] the Predictor-init and coding kernels inside
 * [2] (hipEvents), [6] blocks. */
int zpq_last_api_timing(double out[8]);
/* Runs a tiny kernel exercising the cross-lane idioms (DPP reduction, readlane,
 * bpermute); out8[0..5] must equal {2016, 21344, 123, 2016, 133, 13671}. */
int zpq_selftest(int32_t out8[8]);

/* ---- block-level drop-ins (host side + hot path) ---- */
/* Batched libzpaq::compressBlock (libzpaq.h:1505, libzpaq.cpp:7543): each
 * in[b] becomes one ZPAQ block with one segment, bit-identical to the
 * reference's output for the same (method, filename, comment, dosha1).
 * in[b] may be modified in place exactly where the reference would (E8E9).
 * out_len[b] always receives the needed size; ZPQ_E_OVERFLOW if cap too small. */
int zpq_compress_blocks(const char* method, uint8_t* const* in, const uint32_t* in_len,
                        uint32_t nblocks, const char* const* filename,
                        const char* const* comment, int dosha1,
                        uint8_t* const* out, const uint64_t* out_cap, uint64_t* out_len);

/* libzpaq::decompress (libzpaq.h:1268, libzpaq.cpp:2378) over a whole archive
 * (any number of blocks/segments): all blocks are located on the host, decoded
 * on the device as one batch, and concatenated in order.  Verifies segment
 * SHA-1 trailers when present (status ZPQ_E_CORRUPT on mismatch). */
int zpq_decompress(const uint8_t* archive, uint64_t n, uint8_t* out, uint64_t cap,
                   uint64_t* out_len);
/* Work bound for archives from untrusted sources: the ZPAQL steps ONE call of a block's own (non-standard) PCOMP
 * post-processor may take before the block is rejected with ZPQ_E_VM.  The reference (PostProcessor / ZPAQL::run,
 * libzpaq.cpp:1310-1330, 2236-2330) has no bound at all and a damaged program loops for good; the default here is 2^34
 * (about a minute), which no useful program comes near.  0 restores the default.  Process-wide. */
void zpq_set_pcomp_step_limit(uint64_t steps);

/* ---- host-side pieces of the boundary, exposed for reuse and for tests ---- */
/* SHA1 (libzpaq.h:934-954). */
void zpq_sha1(const uint8_t* in, uint64_t n, uint8_t out20[20]);
/* tests: 1 = hash with the portable compression function instead of the x86 SHA extensions (0 = back to automatic) */
void zpq_sha1_force_portable(int yes);
/* Suffix arrays (the sort inside the byte-aligned LZ77 and BWT pre-processors of methods 2-4; reference: divsufsort,
 * libzpaq.cpp:4658-6434, called from LZBuffer 6463-6883 and compressBlock 7709-7716) of n host buffers in ONE device
 * call: out[i] receives len[i] positions, the end of the string ordered before every byte.  zpq_compress_blocks uses
 * this by itself for batches with 4 or more such blocks; ZPQ_E_UNSUPPORTED when the device declines (a buffer of 16 MiB
 * or more, more than 65 535 buffers, not enough memory) -- the library then sorts on the host. */
int zpq_suffix_arrays_device(const uint8_t* const* in, const uint32_t* len, uint32_t n, uint32_t* const* out);
int zpq_suffix_array_host(const uint8_t* in, uint32_t n, uint32_t* out);      /* the host's sorter (SA-IS), one buffer */
/* compressBlock's level->method expansion (libzpaq.cpp:7579-7691), including
 * level-5 period detection on the data.  Writes a NUL-terminated "x..." /
 * "0..." string. */
int zpq_expand_method(const char* method, const uint8_t* data, uint32_t n, char* out, size_t cap);
/* makeConfig + Compiler (libzpaq.cpp:6887, 2698): "x..." method -> block header
 * bytes as stored (hcomp) and the PCOMP bytes the Compressor codes through the
 * model (len16 + code, empty if none); args9 receives $1..$9. */
int zpq_method_to_header(const char* xmethod, int* args9, uint8_t* hcomp, size_t hcap,
                         size_t* hlen, uint8_t* pcomp, size_t pcap, size_t* plen);
/* The stored block header of built-in model `level` (1 min.cfg, 2 mid.cfg, 3 max.cfg): what
 * Compressor::startBlock(int level) writes (libzpaq.cpp:2793-2839).  Feed it to zpq_plan_create to code blocks
 * with the legacy models through the batch entry points (the coder's input is then a 0 byte + the data:
 * Compressor::postProcess(NULL), libzpaq.cpp:2853-2870). */
int zpq_builtin_model_header(int level, uint8_t* hcomp, size_t hcap, size_t* hlen);
/* The pre-processing half of compressBlock for an explicit "x.." method (libzpaq.cpp:7709-7716; LZ77 / BWT /
 * E8E9, host/preproc.cpp): writes the stream the coder will see (the input itself when the method does not
 * transform it).  E8E9 methods rewrite `data` in place, as the reference rewrites its input buffer. */
int zpq_preprocess_block(const char* xmethod, uint8_t* data, uint32_t n, uint8_t* out, size_t cap, size_t* len);
/* The same with the suffix array of the block supplied (zpq_suffix_arrays_device / zpq_suffix_array_host): what
 * zpq_compress_blocks does for a batch.  For a method with E8E9 (x.,5 / x.,6 / x.,7) the caller filters first with
 * zpq_e8e9 -- the suffixes sorted are those of the filtered bytes -- and this call then does not filter again. */
int zpq_preprocess_block_sa(const char* xmethod, uint8_t* data, uint32_t n, const uint32_t* sa, uint8_t* out, size_t cap, size_t* len);
/* The pre-processors behind the sort for n buffers in one device call: the suffix sort, the LZ77 parse (LZBuffer::fill with a
   suffix array, libzpaq.cpp:6693-6757) and the BWT's last column run on the GPU (device/lz77_kernel.h), the parse comes back
   as a list of matches (4 x uint32: position of the search, offset, length, literals in front) and the host writes LZBuffer's
   codes from it (6759-6883).  zpq_preprocess_blocks_device returns what zpq_preprocess_block does, buffer by buffer;
   zpq_lz77_tokens_host is the host's list, zpq_lz77_serialize the coder. */
int zpq_preprocess_blocks_device(const char* xmethod, uint8_t* const* data, const uint32_t* len, uint32_t n, uint8_t* const* out, const size_t* cap,
                                 size_t* outlen);
int zpq_lz77_tokens_host(const char* xmethod, uint8_t* data, uint32_t n, uint32_t* tokens4, size_t cap, size_t* count);
int zpq_lz77_serialize(const char* xmethod, const uint8_t* data, uint32_t n, const uint32_t* tokens4, size_t ntok, uint8_t* out, size_t cap, size_t* len);
void zpq_e8e9(uint8_t* data, uint32_t n);      /* e8e9 (libzpaq.cpp:6450-6459), in place */
/* Compiler alone (libzpaq.cpp:2698): ZPAQL source text -> header / PCOMP bytes. */
int zpq_assemble(const char* config, const int* args9, uint8_t* hcomp, size_t hcap,
                 size_t* hlen, uint8_t* pcomp, size_t pcap, size_t* plen);
/* Host copies of the predictor's constant tables (for tests): which = 0 squash
 * u16[4096], 1 stretch i16[32768], 2 dt i32[1024], 3 dt2k i32[256], 4 state
 * table u8[1024], 5 ICM initial side table u32[256], 6 ISSE initial side table
 * u32[512], 7 SSE initial row u32[32] (Predictor::init, libzpaq.cpp:1776-1846), 8 / 9 the compact form of
 * stretch the pipelined encoder keeps in LDS (u32[2016] groups of 8 + i16[256] top end).
 * Returns bytes written. */
size_t zpq_table(int which, void* out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
