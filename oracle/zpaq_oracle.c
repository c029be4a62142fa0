/* oracle/zpaq_oracle.c -- TEST INFRASTRUCTURE ONLY (see zpaq_oracle.h).
 *
 * CPU restatement of libzpaq 7.15's hot path, written from the arithmetic
 * specification rather than from the reference's
 * control flow; every function cites the reference lines it restates.  All
 * arithmetic is 32-bit two's complement / unsigned wrap unless noted.
 *
 * Limits (documented deviations): tables larger than 2^31 bytes are refused
 * (the reference allows sizebits up to 32, i.e. multi-GiB tables, where its
 * own index arithmetic already wraps at 32 bits).
 */
#include "zpaq_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef uint8_t U8;
typedef uint16_t U16;
typedef uint32_t U32;
typedef uint64_t U64;

enum { T_NONE = 0, T_CONS, T_CM, T_ICM, T_MATCH, T_AVG, T_MIX2, T_MIX, T_ISSE, T_SSE };
/* bytes per COMP entry, libzpaq.cpp:714 */
static const int comp_len[10] = {0, 2, 3, 2, 3, 4, 6, 6, 3, 5};

/* ------------------------------------------------------------------ tables */
static U16 g_squash[4096];
static int16_t g_stretch[32768];
static int32_t g_dt[1024];
static int32_t g_dt2k[256];
static U8 g_ns[1024];
static int g_tables_built = 0;

/* Bit-history state table, built by the public ZPAQ-spec construction
 * (SURVEY App. A.5); equals sns[] at libzpaq.cpp:726-855 (tested). */
static int st_num_states(int n0, int n1) {
  static const int bound[6] = {20, 48, 15, 8, 6, 5};
  if (n0 < n1) { int t = n0; n0 = n1; n1 = t; }
  if (n0 < 0 || n1 < 0 || n1 >= 6 || n0 > bound[n1]) return 0;
  return 1 + (n1 > 0 && n0 + n1 <= 17);
}
static int st_discount(int n) {
  return (n >= 1) + (n >= 2) + (n >= 3) + (n >= 4) + (n >= 5) + (n >= 7) + (n >= 8);
}
static void st_next(int* n0, int* n1, int y) {
  if (*n0 < *n1) { st_next(n1, n0, 1 - y); return; }
  if (y) { ++*n1; *n0 = st_discount(*n0); }
  else   { ++*n0; *n1 = st_discount(*n1); }
  while (!st_num_states(*n0, *n1)) {
    if (*n1 < 2) --*n0;
    else { *n0 = (*n0 * (*n1 - 1) + (*n1 / 2)) / *n1; --*n1; }
  }
}
static void build_state_table(void) {
  enum { N = 50 };
  static U8 t[N][N][2];
  int state = 0;
  memset(t, 0, sizeof(t));
  for (int i = 0; i < N; ++i)
    for (int n1 = 0; n1 <= i; ++n1) {
      int n0 = i - n1, n = st_num_states(n0, n1);
      if (n) { t[n0][n1][0] = (U8)state; t[n0][n1][1] = (U8)(state + n - 1); state += n; }
    }
  memset(g_ns, 0, sizeof(g_ns));
  for (int n0 = 0; n0 < N; ++n0)
    for (int n1 = 0; n1 < N; ++n1)
      for (int y = 0; y < st_num_states(n0, n1); ++y) {
        int s = t[n0][n1][y];
        int a0 = n0, a1 = n1, b0 = n0, b1 = n1;
        st_next(&a0, &a1, 0);
        st_next(&b0, &b1, 1);
        g_ns[s * 4 + 0] = t[a0][a1][0];
        g_ns[s * 4 + 1] = t[b0][b1][1];
        g_ns[s * 4 + 2] = (U8)n0;
        g_ns[s * 4 + 3] = (U8)n1;
      }
}

static void build_tables(void) {
  if (g_tables_built) return;
  /* dt2k[i]=2048/i, dt[i]=(1<<17)/(2i+3)*2  (libzpaq.cpp:1271, 1307) */
  g_dt2k[0] = 0;
  for (int i = 1; i < 256; ++i) g_dt2k[i] = 2048 / i;
  for (int i = 0; i < 1024; ++i) g_dt[i] = (1 << 17) / (i * 2 + 3) * 2;
  /* squash: comment at libzpaq.cpp:1737; clipped outside [1376,2720) (1739-1741) */
  for (int i = 0; i < 4096; ++i) {
    if (i < 1376) g_squash[i] = 0;
    else if (i >= 2720) g_squash[i] = 32767;
    else g_squash[i] = (U16)(int)(32768.0 / (1 + exp((i - 2048) * (-1.0 / 64))));
  }
  /* stretch: comment at libzpaq.cpp:1743; upper half from the closed form,
   * lower half mirrored exactly as the reference does (1749-1750). */
  for (int i = 16384; i < 32768; ++i)
    g_stretch[i] = (int16_t)((int)(log((i + 0.5) / (32767.5 - i)) * 64 + 0.5 + 100000) - 100000);
  for (int i = 0; i < 16384; ++i) g_stretch[i] = (int16_t)(-g_stretch[32767 - i]);
  build_state_table();
  g_tables_built = 1;
}

const uint16_t* zo_squash_table(void) { build_tables(); return g_squash; }
const int16_t* zo_stretch_table(void) { build_tables(); return g_stretch; }
const int32_t* zo_dt_table(void) { build_tables(); return g_dt; }
const int32_t* zo_dt2k_table(void) { build_tables(); return g_dt2k; }
const uint8_t* zo_state_table(void) { build_tables(); return g_ns; }

/* The two known-answer checksums the reference asserts in DEBUG builds
 * (libzpaq.cpp:1752-1761). */
int zo_tables_ok(void) {
  build_tables();
  U32 sqsum = 0, stsum = 0;
  for (int i = 32767; i >= 0; --i) stsum = stsum * 3 + (U32)(int32_t)g_stretch[i];
  for (int i = 4095; i >= 0; --i) sqsum = sqsum * 3 + g_squash[i];
  return stsum == 3887533746u && sqsum == 2278286169u;
}

static inline int squash(int x) { return g_squash[x + 2048]; }    /* libzpaq.h:1160 */
static inline int stretch(int x) { return g_stretch[x]; }          /* libzpaq.h:1167 */
static inline int clamp2k(int x) { return x < -2048 ? -2048 : x > 2047 ? 2047 : x; }
static inline int clamp512k(int x) {
  return x < -(1 << 19) ? -(1 << 19) : x >= (1 << 19) ? (1 << 19) - 1 : x;
}
static inline int ns_next(int s, int y) { return g_ns[s * 4 + y]; }   /* libzpaq.h:1101 */
static inline int ns_cminit(int s) {                                    /* libzpaq.h:1106 */
  return ((g_ns[s * 4 + 3] * 2 + 1) << 22) / (g_ns[s * 4 + 2] + g_ns[s * 4 + 3] + 1);
}

/* ------------------------------------------------------------------- model */
typedef struct {
  int type;
  const U8* cp;      /* COMP bytes of this component */
  U32* cm; size_t cm_n;   /* Component::cm  (U32 / int view) */
  U8* ht;  size_t ht_n;   /* Component::ht  */
  U16* a16; size_t a16_n; /* Component::a16 */
  U32 limit, cxt, a, b, c;
} comp_t;

struct zo_model {
  U8* header; size_t hlen;      /* stored form */
  int n;                        /* components */
  const U8* prog; int prog_len; /* HCOMP bytes incl. trailing 0 */
  /* HCOMP machine (ZPAQL private state, libzpaq.h:1052-1062) */
  U8* M; U32 msize; U32* H; U32 hsize; U32 R[256];
  U32 A, B, C, D; int F;
  /* predictor (libzpaq.h:1131-1137) */
  int c8, hmap4;
  int32_t p[256];
  U32 h[256];
  comp_t comp[256];
  double memory;
};

static void* xcalloc(size_t n, size_t sz) { return calloc(n ? n : 1, sz); }

void zo_model_free(zo_model* m) {
  if (!m) return;
  for (int i = 0; i < 256; ++i) { free(m->comp[i].cm); free(m->comp[i].ht); free(m->comp[i].a16); }
  free(m->M); free(m->H); free(m->header); free(m);
}

/* ZPAQL::read (libzpaq.cpp:887-931) + Predictor::init (1723-1851). */
zo_model* zo_model_new(const uint8_t* header, size_t hlen, int* err) {
  int e = ZO_EHEADER;
  build_tables();
  zo_model* m = (zo_model*)calloc(1, sizeof(zo_model));
  if (!m) { if (err) *err = ZO_ENOMEM; return 0; }
  if (hlen < 8) goto fail;
  m->header = (U8*)malloc(hlen);
  if (!m->header) { e = ZO_ENOMEM; goto fail; }
  memcpy(m->header, header, hlen);
  m->hlen = hlen;
  {
    size_t hsize = header[0] + 256u * header[1];
    if (hsize + 2 != hlen) goto fail;
  }
  const U8* hd = m->header;
  int hh = hd[2], hm = hd[3];
  m->n = hd[6];
  if (hh > 24 || hm > 28) goto fail; /* sanity: reference allows 32 (libzpaq.cpp:1018) */
  /* walk COMP */
  size_t pos = 7;
  for (int i = 0; i < m->n; ++i) {
    if (pos >= hlen) goto fail;
    int type = hd[pos];
    if (type < 1 || type > 9) goto fail;              /* "Invalid component type" */
    if (pos + comp_len[type] > hlen) goto fail;
    m->comp[i].type = type;
    m->comp[i].cp = hd + pos;
    pos += comp_len[type];
  }
  if (pos >= hlen || hd[pos] != 0) goto fail;          /* "missing COMP END" */
  ++pos;
  if (pos >= hlen || hd[hlen - 1] != 0) goto fail;     /* "missing HCOMP END" */
  m->prog = hd + pos;
  m->prog_len = (int)(hlen - pos);
  /* memory(): libzpaq.cpp:986-1006 (header.size() there = hsize+300) */
  m->memory = ldexp(1.0, hh + 2) + ldexp(1.0, hm) + ldexp(1.0, hd[4] + 2) + ldexp(1.0, hd[5])
            + (double)(hlen - 2 + 300);
  /* machine state: ZPAQL::init (1012-1024) */
  m->hsize = 1u << hh; m->msize = 1u << hm;
  m->H = (U32*)xcalloc(m->hsize, 4);
  m->M = (U8*)xcalloc(m->msize, 1);
  if (!m->H || !m->M) { e = ZO_ENOMEM; goto fail; }
  m->c8 = 1; m->hmap4 = 1;   /* Predictor ctor, libzpaq.cpp:1703 */
  /* components: Predictor::init 1776-1846 */
  for (int i = 0; i < m->n; ++i) {
    comp_t* cr = &m->comp[i];
    const U8* cp = cr->cp;
    double size = ldexp(1.0, cp[1]);
    switch (cr->type) {
      case T_CONS: m->p[i] = (cp[1] - 128) * 4; break;
      case T_CM:
        if (cp[1] > 28) goto fail;
        cr->cm_n = (size_t)1 << cp[1];
        cr->cm = (U32*)xcalloc(cr->cm_n, 4);
        if (!cr->cm) { e = ZO_ENOMEM; goto fail; }
        cr->limit = cp[2] * 4u;
        for (size_t j = 0; j < cr->cm_n; ++j) cr->cm[j] = 0x80000000u;
        m->memory += 4 * size;
        break;
      case T_ICM:
        if (cp[1] > 24) goto fail;
        cr->limit = 1023;
        cr->cm_n = 256; cr->cm = (U32*)xcalloc(256, 4);
        cr->ht_n = (size_t)64 << cp[1]; cr->ht = (U8*)xcalloc(cr->ht_n, 1);
        if (!cr->cm || !cr->ht) { e = ZO_ENOMEM; goto fail; }
        for (int j = 0; j < 256; ++j) cr->cm[j] = (U32)ns_cminit(j);
        m->memory += 64 * size + 1024;
        break;
      case T_MATCH:
        if (cp[1] > 28 || cp[2] > 30) goto fail;
        cr->cm_n = (size_t)1 << cp[1]; cr->cm = (U32*)xcalloc(cr->cm_n, 4);
        cr->ht_n = (size_t)1 << cp[2]; cr->ht = (U8*)xcalloc(cr->ht_n, 1);
        if (!cr->cm || !cr->ht) { e = ZO_ENOMEM; goto fail; }
        cr->ht[0] = 1;
        m->memory += 4 * size + ldexp(1.0, cp[2]);
        break;
      case T_AVG:
        if (cp[1] >= i || cp[2] >= i) goto fail;
        break;
      case T_MIX2:
        if (cp[1] > 29 || cp[2] >= i || cp[3] >= i) goto fail;
        cr->c = 1u << cp[1];
        cr->a16_n = (size_t)1 << cp[1]; cr->a16 = (U16*)xcalloc(cr->a16_n, 2);
        if (!cr->a16) { e = ZO_ENOMEM; goto fail; }
        for (size_t j = 0; j < cr->a16_n; ++j) cr->a16[j] = 32768;
        m->memory += 2 * size;
        break;
      case T_MIX: {
        if (cp[1] > 24 || cp[2] >= i || cp[3] < 1 || cp[3] > i - cp[2]) goto fail;
        int mm = cp[3];
        cr->c = 1u << cp[1];
        cr->cm_n = (size_t)mm << cp[1]; cr->cm = (U32*)xcalloc(cr->cm_n, 4);
        if (!cr->cm) { e = ZO_ENOMEM; goto fail; }
        for (size_t j = 0; j < cr->cm_n; ++j) cr->cm[j] = (U32)(65536 / mm);
        m->memory += 4 * size * mm;
        break;
      }
      case T_ISSE:
        if (cp[1] > 24 || cp[2] >= i) goto fail;
        cr->ht_n = (size_t)64 << cp[1]; cr->ht = (U8*)xcalloc(cr->ht_n, 1);
        cr->cm_n = 512; cr->cm = (U32*)xcalloc(512, 4);
        if (!cr->cm || !cr->ht) { e = ZO_ENOMEM; goto fail; }
        for (int j = 0; j < 256; ++j) {
          cr->cm[j * 2] = 1 << 15;
          cr->cm[j * 2 + 1] = (U32)clamp512k(stretch(ns_cminit(j) >> 8) * 1024);
        }
        m->memory += 64 * size + 2048;
        break;
      case T_SSE:
        if (cp[1] > 24 || cp[2] >= i || cp[3] > cp[4] * 4) goto fail;
        cr->cm_n = (size_t)32 << cp[1]; cr->cm = (U32*)xcalloc(cr->cm_n, 4);
        if (!cr->cm) { e = ZO_ENOMEM; goto fail; }
        cr->limit = cp[4] * 4u;
        for (size_t j = 0; j < cr->cm_n; ++j)
          cr->cm[j] = ((U32)squash((int)(j & 31) * 64 - 992) << 17) | cp[3];
        m->memory += 128 * size;
        break;
    }
  }
  if (err) *err = 0;
  return m;
fail:
  zo_model_free(m);
  if (err) *err = e;
  return 0;
}

double zo_model_memory(const zo_model* m) { return m->memory; }
int zo_model_ncomp(const zo_model* m) { return m->n; }
void zo_model_p(const zo_model* m, int32_t* p_out) { memcpy(p_out, m->p, sizeof(int32_t) * (size_t)m->n); }
uint32_t zo_model_h(const zo_model* m, int i) { return m->h[i]; }

/* ----------------------------------------------------------------- HCOMP VM
 * ZPAQL::run0 / execute (libzpaq.cpp:1027-1262), decoded by operand group
 *  instead of the reference's 256-way switch. */
static inline U32 vm_get(zo_model* m, int k, int* pc) {
  switch (k) {
    case 0: return m->A;
    case 1: return m->B;
    case 2: return m->C;
    case 3: return m->D;
    case 4: return m->M[m->B & (m->msize - 1)];
    case 5: return m->M[m->C & (m->msize - 1)];
    case 6: return m->H[m->D & (m->hsize - 1)];
    default: return m->prog[(*pc)++];
  }
}
static inline void vm_set(zo_model* m, int g, U32 v) {
  switch (g) {
    case 0: m->A = v; break;
    case 1: m->B = v; break;
    case 2: m->C = v; break;
    case 3: m->D = v; break;
    case 4: m->M[m->B & (m->msize - 1)] = (U8)v; break;
    case 5: m->M[m->C & (m->msize - 1)] = (U8)v; break;
    case 6: m->H[m->D & (m->hsize - 1)] = v; break;
  }
}

static int vm_run(zo_model* m, U32 input) {
  int pc = 0;
  const int len = m->prog_len;
  m->A = input;
  for (;;) {
    if (pc < 0 || pc >= len) return ZO_EVM;  /* reference would hit opcode 0 in the guard */
    int op = m->prog[pc++];
    int g = op >> 3, k = op & 7;
    if (op < 64) {
      if (g == 7) {
        switch (op) {
          case 56: return 0;                                                     /* HALT */
          case 57: break;                                                        /* OUT: no sink in HCOMP */
          case 59: m->A = (m->A + m->M[m->B & (m->msize - 1)] + 512) * 773; break;  /* HASH */
          case 60: { U32* d = &m->H[m->D & (m->hsize - 1)]; *d = (*d + m->A + 512) * 773; break; } /* HASHD */
          case 63: pc += 1 + (int8_t)m->prog[pc]; break;                          /* JMP */
          default: return ZO_EVM;
        }
      } else if (k == 7) {
        if (pc >= len) return ZO_EVM;
        switch (g) {
          case 0: case 1: case 2: case 3: vm_set(m, g, m->R[m->prog[pc++]]); break;  /* X=R N */
          case 4: if (m->F) pc += 1 + (int8_t)m->prog[pc]; else ++pc; break;          /* JT */
          case 5: if (!m->F) pc += 1 + (int8_t)m->prog[pc]; else ++pc; break;         /* JF */
          case 6: m->R[m->prog[pc++]] = m->A; break;                                  /* R=A N */
        }
      } else {
        U32 x = vm_get(m, g, &pc);
        switch (k) {
          case 0:  /* X<>A ; byte operands exchange only the low 8 bits (libzpaq.h:1073) */
            if (op == 0) return ZO_EVM;
            if (g == 4 || g == 5) { U32 a = m->A; vm_set(m, g, a & 255); m->A = (a & 0xFFFFFF00u) | x; }
            else { U32 a = m->A; vm_set(m, g, a); m->A = x; }
            break;
          case 1: vm_set(m, g, x + 1); break;
          case 2: vm_set(m, g, x - 1); break;
          case 3: vm_set(m, g, ~x); break;
          case 4: vm_set(m, g, 0); break;
          default: return ZO_EVM;   /* k==5,6 */
        }
      }
    } else if (op < 120) {            /* dest(g-8) = src(k) */
      U32 v = vm_get(m, k, &pc);
      vm_set(m, g - 8, v);
    } else if (op < 128) {
      return ZO_EVM;
    } else if (op < 240) {            /* A op= src(k) */
      U32 v = vm_get(m, k, &pc);
      switch (g - 16) {
        case 0: m->A += v; break;
        case 1: m->A -= v; break;
        case 2: m->A *= v; break;
        case 3: m->A = v ? m->A / v : 0; break;
        case 4: m->A = v ? m->A % v : 0; break;
        case 5: m->A &= v; break;
        case 6: m->A &= ~v; break;
        case 7: m->A |= v; break;
        case 8: m->A ^= v; break;
        case 9: m->A <<= (v & 31); break;
        case 10: m->A >>= (v & 31); break;
        case 11: m->F = (m->A == v); break;
        case 12: m->F = (m->A < v); break;
        case 13: m->F = (m->A > v); break;
      }
    } else if (op == 255) {           /* LJ */
      if (pc + 1 >= len) return ZO_EVM;
      int t = m->prog[pc] + 256 * m->prog[pc + 1];
      if (t >= len) return ZO_EVM;    /* hbegin+t >= hend (libzpaq.cpp:1258) */
      pc = t;
    } else return ZO_EVM;
  }
}

/* ---------------------------------------------------------------- find()
 * Predictor::find (libzpaq.cpp:2072-2088). */
static size_t find_row(U8* ht, size_t ht_n, int sizebits, U32 cxt) {
  int chk = (cxt >> sizebits) & 255;
  size_t h0 = ((size_t)(U32)(cxt * 16u)) & (ht_n - 16);
  if (ht[h0] == chk) return h0;
  size_t h1 = h0 ^ 16;
  if (ht[h1] == chk) return h1;
  size_t h2 = h0 ^ 32;
  if (ht[h2] == chk) return h2;
  size_t v;
  if (ht[h0 + 1] <= ht[h1 + 1] && ht[h0 + 1] <= ht[h2 + 1]) v = h0;
  else if (ht[h1 + 1] < ht[h2 + 1]) v = h1;
  else v = h2;
  memset(ht + v, 0, 16);
  ht[v] = (U8)chk;
  return v;
}

/* ---------------------------------------------------------------- predict
 * Predictor::predict0 (libzpaq.cpp:1854-1951). */
int zo_predict(zo_model* m) {
  const int c8 = m->c8, hmap4 = m->hmap4;
  int32_t* p = m->p;
  for (int i = 0; i < m->n; ++i) {
    comp_t* cr = &m->comp[i];
    const U8* cp = cr->cp;
    switch (cr->type) {
      case T_CONS: break;
      case T_CM:
        cr->cxt = m->h[i] ^ (U32)hmap4;
        p[i] = stretch(cr->cm[cr->cxt & (cr->cm_n - 1)] >> 17);
        break;
      case T_ICM:
        if (c8 == 1 || (c8 & 0xf0) == 16)
          cr->c = (U32)find_row(cr->ht, cr->ht_n, cp[1] + 2, m->h[i] + 16u * (U32)c8);
        cr->cxt = cr->ht[cr->c + (hmap4 & 15)];
        p[i] = stretch(cr->cm[cr->cxt & 255] >> 8);
        break;
      case T_MATCH:
        if (cr->a == 0) p[i] = 0;
        else {
          cr->c = (cr->ht[(cr->limit - cr->b) & (cr->ht_n - 1)] >> (7 - cr->cxt)) & 1;
          int d = g_dt2k[cr->a];
          p[i] = stretch((cr->c ? -d : d) & 32767);
        }
        break;
      case T_AVG:
        p[i] = (p[cp[1]] * cp[3] + p[cp[2]] * (256 - cp[3])) >> 8;
        break;
      case T_MIX2: {
        cr->cxt = (m->h[i] + (U32)(c8 & cp[5])) & (cr->c - 1);
        int w = cr->a16[cr->cxt];
        p[i] = (w * p[cp[2]] + (65536 - w) * p[cp[3]]) >> 16;
        break;
      }
      case T_MIX: {
        int mm = cp[3];
        cr->cxt = ((m->h[i] + (U32)(c8 & cp[5])) & (cr->c - 1)) * (U32)mm;
        const int32_t* wt = (const int32_t*)&cr->cm[cr->cxt];
        int s = 0;
        for (int j = 0; j < mm; ++j) s += (wt[j] >> 8) * p[cp[2] + j];
        p[i] = clamp2k(s >> 8);
        break;
      }
      case T_ISSE: {
        if (c8 == 1 || (c8 & 0xf0) == 16)
          cr->c = (U32)find_row(cr->ht, cr->ht_n, cp[1] + 2, m->h[i] + 16u * (U32)c8);
        cr->cxt = cr->ht[cr->c + (hmap4 & 15)];
        const int32_t* wt = (const int32_t*)&cr->cm[cr->cxt * 2];
        p[i] = clamp2k((wt[0] * p[cp[2]] + wt[1] * 64) >> 16);
        break;
      }
      case T_SSE: {
        cr->cxt = (m->h[i] + (U32)c8) * 32u;
        int pq = p[cp[2]] + 992;
        if (pq < 0) pq = 0;
        if (pq > 1983) pq = 1983;
        int wt = pq & 63;
        pq >>= 6;
        cr->cxt += (U32)pq;
        size_t mask = cr->cm_n - 1;
        p[i] = stretch((int)(((cr->cm[cr->cxt & mask] >> 10) * (U32)(64 - wt)
                            + (cr->cm[(cr->cxt + 1) & mask] >> 10) * (U32)wt) >> 13));
        cr->cxt += (U32)(wt >> 5);
        break;
      }
    }
  }
  return squash(p[m->n - 1]);
}

/* Predictor::train (libzpaq.h:1151-1157): error*dt[count] is a wrapping
 * 32-bit product. */
static inline void train(comp_t* cr, int y) {
  U32* pn = &cr->cm[cr->cxt & (cr->cm_n - 1)];
  U32 count = *pn & 0x3ff;
  int32_t error = y * 32767 - (int32_t)(*pn >> 17);
  U32 prod = (U32)error * (U32)g_dt[count];
  *pn += (prod & 0xFFFFFC00u) + (count < cr->limit);
}

/* ---------------------------------------------------------------- update
 * Predictor::update0 (libzpaq.cpp:1954-2066). */
int zo_update(zo_model* m, int y) {
  const int hmap4 = m->hmap4;
  int32_t* p = m->p;
  for (int i = 0; i < m->n; ++i) {
    comp_t* cr = &m->comp[i];
    const U8* cp = cr->cp;
    switch (cr->type) {
      case T_CM: train(cr, y); break;
      case T_ICM: {
        U8* s = &cr->ht[cr->c + (hmap4 & 15)];
        *s = (U8)ns_next(*s, y);
        U32* pn = &cr->cm[cr->cxt & 255];
        *pn += (U32)((int32_t)((U32)(y * 32767) - (*pn >> 8)) >> 2);
        break;
      }
      case T_MATCH: {
        size_t mask = cr->ht_n - 1;
        if ((int)cr->c != y) cr->a = 0;
        cr->ht[cr->limit & mask] = (U8)(cr->ht[cr->limit & mask] * 2 + y);
        if (++cr->cxt == 8) {
          cr->cxt = 0;
          cr->limit = (cr->limit + 1) & (U32)mask;
          U32* idx = &cr->cm[m->h[i] & (cr->cm_n - 1)];
          if (cr->a == 0) {
            cr->b = cr->limit - *idx;
            if (cr->b & (U32)mask)
              while (cr->a < 255 &&
                     cr->ht[(cr->limit - cr->a - 1) & mask] == cr->ht[(cr->limit - cr->a - cr->b - 1) & mask])
                ++cr->a;
          } else cr->a += cr->a < 255;
          *idx = cr->limit;
        }
        break;
      }
      case T_MIX2: {
        int err = ((y * 32767 - squash(p[i])) * cp[4]) >> 5;
        int w = cr->a16[cr->cxt];
        w += (err * (p[cp[2]] - p[cp[3]]) + (1 << 12)) >> 13;
        if (w < 0) w = 0;
        if (w > 65535) w = 65535;
        cr->a16[cr->cxt] = (U16)w;
        break;
      }
      case T_MIX: {
        int mm = cp[3];
        int err = ((y * 32767 - squash(p[i])) * cp[4]) >> 4;
        int32_t* wt = (int32_t*)&cr->cm[cr->cxt];
        for (int j = 0; j < mm; ++j)
          wt[j] = clamp512k(wt[j] + ((err * p[cp[2] + j] + (1 << 12)) >> 13));
        break;
      }
      case T_ISSE: {
        int err = y * 32767 - squash(p[i]);
        int32_t* wt = (int32_t*)&cr->cm[cr->cxt * 2];
        wt[0] = clamp512k(wt[0] + ((err * p[cp[2]] + (1 << 12)) >> 13));
        wt[1] = clamp512k(wt[1] + ((err + 16) >> 5));
        cr->ht[cr->c + (hmap4 & 15)] = (U8)ns_next((int)cr->cxt, y);
        break;
      }
      case T_SSE: train(cr, y); break;
      default: break;   /* CONS, AVG */
    }
  }
  /* c8 / hmap4 bookkeeping (libzpaq.cpp:2055-2065) */
  m->c8 += m->c8 + y;
  if (m->c8 >= 256) {
    int r = vm_run(m, (U32)(m->c8 - 256));
    if (r < 0) return r;
    m->hmap4 = 1;
    m->c8 = 1;
    for (int i = 0; i < m->n; ++i) m->h[i] = m->H[(U32)i & (m->hsize - 1)];
  } else if (m->c8 >= 16 && m->c8 < 32)
    m->hmap4 = (m->hmap4 & 0xf) << 5 | y << 4 | 1;
  else
    m->hmap4 = (m->hmap4 & 0x1f0) | (((m->hmap4 & 0xf) * 2 + y) & 0xf);
  return 0;
}

/* ------------------------------------------------------------------ coder */
typedef struct { U8* out; size_t cap, n; U32 low, high; } enc_t;
static inline void put(enc_t* e, int c) { if (e->n < e->cap) e->out[e->n] = (U8)c; ++e->n; }

/* Encoder::encode (libzpaq.cpp:2402-2416) */
static inline void encode_bit(enc_t* e, int y, U32 p) {
  U32 mid = e->low + (U32)(((U64)(e->high - e->low) * p) >> 16);
  if (y) e->high = mid; else e->low = mid + 1;
  while ((e->high ^ e->low) < 0x1000000u) {
    put(e, (int)(e->high >> 24));
    e->high = e->high << 8 | 255;
    e->low = e->low << 8;
    e->low += (e->low == 0);
  }
}

long long zo_encode(const uint8_t* header, size_t hlen, const uint8_t* in, size_t n,
                    uint8_t* out, size_t cap, uint16_t* trace, size_t ntrace) {
  int err = 0;
  if (hlen >= 7 && header[6] == 0) {
    /* stored mode: Encoder::compress n==0 branch (libzpaq.cpp:2436-2446) */
    enc_t e = {out, cap, 0, 0, 0};
    size_t pos = 0;
    while (pos < n) {
      size_t k = n - pos; if (k > 65536) k = 65536;
      put(&e, (int)(k >> 24) & 255); put(&e, (int)(k >> 16) & 255);
      put(&e, (int)(k >> 8) & 255);  put(&e, (int)k & 255);
      for (size_t i = 0; i < k; ++i) put(&e, in[pos + i]);
      pos += k;
    }
    return (long long)e.n;
  }
  zo_model* m = zo_model_new(header, hlen, &err);
  if (!m) return err;
  enc_t e = {out, cap, 0, 1, 0xFFFFFFFFu};      /* Encoder::init 2394-2399 */
  size_t bit = 0;
  for (size_t k = 0; k < n; ++k) {
    int c = in[k];
    encode_bit(&e, 0, 0);                         /* not-EOS flag (2426) */
    for (int i = 7; i >= 0; --i) {
      int pr = zo_predict(m);
      if (trace && bit < ntrace) trace[bit] = (uint16_t)pr;
      ++bit;
      int y = (c >> i) & 1;
      encode_bit(&e, y, (U32)pr * 2 + 1);
      int r = zo_update(m, y);
      if (r < 0) { zo_model_free(m); return r; }
    }
  }
  encode_bit(&e, 1, 0);                            /* EOS: flushes 4 bytes (2424) */
  zo_model_free(m);
  return (long long)e.n;
}

/* Decoder::decode / decompress (libzpaq.cpp:2104-2155). */
long long zo_decode(const uint8_t* header, size_t hlen, const uint8_t* coded, size_t ncoded,
                    uint8_t* out, size_t cap, size_t* consumed) {
  int err = 0;
  size_t rp = 0, n = 0;
  if (hlen >= 7 && header[6] == 0) {               /* stored (2146-2154) */
    for (;;) {
      if (rp + 4 > ncoded) return ZO_EEOF;
      U32 len = (U32)coded[rp] << 24 | (U32)coded[rp + 1] << 16 | (U32)coded[rp + 2] << 8 | coded[rp + 3];
      rp += 4;
      if (len == 0) break;
      if (rp + len > ncoded) return ZO_EEOF;
      for (U32 i = 0; i < len; ++i) { if (n < cap) out[n] = coded[rp + i]; ++n; }
      rp += len;
    }
    if (consumed) *consumed = rp;
    return (long long)n;
  }
  zo_model* m = zo_model_new(header, hlen, &err);
  if (!m) return err;
  U32 low = 1, high = 0xFFFFFFFFu, curr = 0;
  for (int i = 0; i < 4; ++i) { if (rp >= ncoded) { zo_model_free(m); return ZO_EEOF; } curr = curr << 8 | coded[rp++]; }
  long long ret = 0;
  for (;;) {
    /* decode(0): EOS flag */
    int c = 1;
    for (int b = -1; b < 8; ++b) {
      U32 p = 0;
      if (b >= 0) p = (U32)zo_predict(m) * 2 + 1;
      if (curr < low || curr > high) { ret = ZO_ECORRUPT; goto done; }
      U32 mid = low + (U32)(((U64)(high - low) * p) >> 16);
      int y;
      if (curr <= mid) { y = 1; high = mid; } else { y = 0; low = mid + 1; }
      while ((high ^ low) < 0x1000000u) {
        high = high << 8 | 255;
        low = low << 8;
        low += (low == 0);
        if (rp >= ncoded) { ret = ZO_EEOF; goto done; }
        curr = curr << 8 | coded[rp++];
      }
      if (b < 0) {
        if (y) { if (curr != 0) ret = ZO_ECORRUPT; else ret = (long long)n; goto done; }
      } else {
        c += c + y;
        int r = zo_update(m, y);
        if (r < 0) { ret = r; goto done; }
      }
    }
    if (n < cap) out[n] = (U8)(c - 256);
    ++n;
  }
done:
  zo_model_free(m);
  if (consumed) *consumed = rp;
  return ret;
}

int zo_hcomp_trace(const uint8_t* header, size_t hlen, const uint8_t* in, size_t n, uint32_t* hout) {
  int err = 0;
  zo_model* m = zo_model_new(header, hlen, &err);
  if (!m) return err;
  for (size_t k = 0; k < n; ++k) {
    int r = vm_run(m, in[k]);
    if (r < 0) { zo_model_free(m); return r; }
    for (int i = 0; i < m->n; ++i) hout[k * (size_t)m->n + i] = m->H[(U32)i & (m->hsize - 1)];
  }
  zo_model_free(m);
  return 0;
}
