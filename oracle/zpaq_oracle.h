/* oracle/zpaq_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the libzpaq 7.15 hot path (context-mixing
 * predictor + HCOMP VM + binary arithmetic coder).  It exists to CHECK the HIP
 * path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it; nothing under zpaq_amd/ links, imports or executes it.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this restatement against
 * (a) the table checksums the reference asserts (libzpaq.cpp:1752-1761),
 * (b) the known-answer archives of BASELINE.md §2 / tests/golden/, and
 * (c) live differential runs against oracle/_ref (the compiled reference) where
 *     that is present.
 */
#ifndef ZPAQ_ORACLE_H
#define ZPAQ_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zo_model zo_model;

/* Error codes (negative returns). */
enum { ZO_OK = 0, ZO_EHEADER = -1, ZO_ENOMEM = -2, ZO_ECORRUPT = -3, ZO_EOVERFLOW = -4,
       ZO_EVM = -5, ZO_EEOF = -6 };

/* Constant tables (generated from closed forms). */
const uint16_t* zo_squash_table(void);  /* [4096]  libzpaq.cpp:1737-1741 */
const int16_t*  zo_stretch_table(void); /* [32768] libzpaq.cpp:1743-1750 */
const int32_t*  zo_dt_table(void);      /* [1024]  libzpaq.cpp:1307      */
const int32_t*  zo_dt2k_table(void);    /* [256]   libzpaq.cpp:1271      */
const uint8_t*  zo_state_table(void);   /* [1024]  libzpaq.cpp:726-855   */
/* 1 iff the two checksums of libzpaq.cpp:1759-1760 hold for our tables. */
int zo_tables_ok(void);

/* Model over a block header exactly as stored in the archive:
 * hsize_lo hsize_hi hh hm ph pm n COMP... 0 HCOMP... 0  (ZPAQL::read, libzpaq.cpp:887). */
zo_model* zo_model_new(const uint8_t* header, size_t hlen, int* err);
void      zo_model_free(zo_model*);
double    zo_model_memory(const zo_model*);      /* ZPAQL::memory(), libzpaq.cpp:986 */
int       zo_model_ncomp(const zo_model*);
int       zo_predict(zo_model*);                 /* Predictor::predict0, 1854 -> 0..32767 */
int       zo_update(zo_model*, int y);           /* Predictor::update0, 1954; <0 on VM error */
void      zo_model_p(const zo_model*, int32_t* p_out); /* p[0..n-1] after predict */
uint32_t  zo_model_h(const zo_model*, int i);

/* Encoder::compress over in[0..n) followed by EOS (libzpaq.cpp:2419-2447).
 * `in` is the byte sequence the Compressor feeds the Encoder: the PP header
 * byte(s) then the (pre-processed) data.  Output: coded bytes including the 4
 * flush bytes of encode(1,0); for n_comp==0 the stored framing incl. nothing
 * more (the caller appends the 00 00 00 00 terminator in both modes).
 * If trace!=NULL, trace[k] receives predict() (0..32767) of coded bit k for
 * k<ntrace.  Returns bytes produced (may exceed cap: output truncated) or <0. */
long long zo_encode(const uint8_t* header, size_t hlen, const uint8_t* in, size_t n,
                    uint8_t* out, size_t cap, uint16_t* trace, size_t ntrace);

/* Decoder::decompress until EOS (libzpaq.cpp:2127-2155).  Decodes from
 * coded[0..ncoded), writes decoded bytes (PP header bytes included) to out.
 * *consumed = bytes of `coded` read.  Returns decoded byte count or <0. */
long long zo_decode(const uint8_t* header, size_t hlen, const uint8_t* coded, size_t ncoded,
                    uint8_t* out, size_t cap, size_t* consumed);

/* Run only the HCOMP program over a byte sequence; writes H[0..n_comp) after
 * each byte into hout[k*n_comp + i].  Returns 0 or <0. */
int zo_hcomp_trace(const uint8_t* header, size_t hlen, const uint8_t* in, size_t n, uint32_t* hout);

#ifdef __cplusplus
}
#endif
#endif
