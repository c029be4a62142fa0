// Compression-side pre-processors of libzpaq::compressBlock (host, per block): the E8E9 filter, the two LZ77
// encodings and the BWT (reference: e8e9 libzpaq.cpp:6450-6459, LZBuffer 6463-6883, dispatch 7709-7716).  Their
// inverses are the PCOMP programs host/method.cpp emits, which travel inside the archive; the bytes produced here
// are what the context model then codes (or, with no model, what is stored), so they have to be the reference's
// byte for byte.  tests/test_host.py compares every branch with the compiled reference.
//
//   args[0] log2 of the block size in MiB        args[1] 1 bit-packed LZ77, 2 byte-aligned LZ77, 3 BWT, +4 E8E9
//   args[2] minimum match length                 args[3] second (longer) context order searched first, or 0
//   args[4] log2 searches per hash bucket        args[5] log2 hash table size; args[0] + 21 selects a suffix array
//   args[6] look-ahead of the second context
#include <algorithm>
#include <cstring>

#include "common.hpp"

namespace zpq {

namespace {

int bit_length(unsigned x) { return x ? 32 - __builtin_clz(x) : 0; }   // lg() of the reference

// ---------------------------------------------------------------------------------------------------------
// Suffix array by induced sorting (SA-IS, Nong / Zhang / Chan).  The reference calls divsufsort; any correct
// suffix sorter yields the same array.  `s` ends with a unique smallest symbol 0.
template <class T>
void sais(const T* s, int* sa, int n, int K) {
  if (n == 1) { sa[0] = 0; return; }
  if (n == 2) { sa[0] = 1; sa[1] = 0; return; }
  // suffix types: bit 0 = S-type, bit 1 = leftmost S-type (LMS)
  std::vector<unsigned char> ty((size_t)n);
  ty[(size_t)n - 1] = 1;
  for (int i = n - 2; i >= 0; --i) ty[(size_t)i] = (unsigned char)(s[i] < s[i + 1] || (s[i] == s[i + 1] && (ty[(size_t)i + 1] & 1)));
  for (int i = 1; i < n; ++i) if ((ty[(size_t)i] & 1) && !(ty[(size_t)i - 1] & 1)) ty[(size_t)i] |= 2;
  std::vector<int> cnt((size_t)K, 0), bkt((size_t)K);
  for (int i = 0; i < n; ++i) ++cnt[(size_t)s[i]];
  auto buckets = [&](bool ends) {
    int sum = 0;
    for (int c = 0; c < K; ++c) { sum += cnt[(size_t)c]; bkt[(size_t)c] = ends ? sum : sum - cnt[(size_t)c]; }
  };
  auto induce = [&]() {
    buckets(false);
    for (int i = 0; i < n; ++i) {
      const int j = sa[i] - 1;
      if (j >= 0 && !(ty[(size_t)j] & 1)) sa[bkt[(size_t)s[j]]++] = j;
    }
    buckets(true);
    for (int i = n - 1; i >= 0; --i) {
      const int j = sa[i] - 1;
      if (j >= 0 && (ty[(size_t)j] & 1)) sa[--bkt[(size_t)s[j]]] = j;
    }
  };
  // 1. sort the LMS substrings
  std::fill(sa, sa + n, -1);
  buckets(true);
  for (int i = 1; i < n; ++i) if (ty[(size_t)i] & 2) sa[--bkt[(size_t)s[i]]] = i;
  induce();
  // 2. name them
  int n1 = 0;
  for (int i = 0; i < n; ++i) if (sa[i] > 0 && (ty[(size_t)sa[i]] & 2)) sa[n1++] = sa[i];
  std::fill(sa + n1, sa + n, -1);
  int names = 0, prev = -1;
  for (int i = 0; i < n1; ++i) {
    const int pos = sa[i];
    bool diff = prev < 0;
    for (int d = 0; !diff; ++d) {
      if (s[pos + d] != s[prev + d] || ty[(size_t)pos + d] != ty[(size_t)prev + d]) diff = true;
      else if (d > 0 && ((ty[(size_t)pos + d] & 2) || (ty[(size_t)prev + d] & 2))) break;
    }
    if (diff) { ++names; prev = pos; }
    sa[n1 + pos / 2] = names - 1;
  }
  std::vector<int> s1((size_t)n1), sa1((size_t)n1);
  for (int i = n1, j = 0; i < n; ++i) if (sa[i] >= 0) s1[(size_t)j++] = sa[i];
  // 3. order of the LMS suffixes: direct when all names differ, else recurse
  if (names < n1) sais(s1.data(), sa1.data(), n1, names);
  else for (int i = 0; i < n1; ++i) sa1[(size_t)s1[(size_t)i]] = i;
  // 4. induce the full array from the sorted LMS suffixes
  for (int i = 1, j = 0; i < n; ++i) if (ty[(size_t)i] & 2) s1[(size_t)j++] = i;          // LMS positions in text order
  for (int i = 0; i < n1; ++i) sa1[(size_t)i] = s1[(size_t)sa1[(size_t)i]];
  std::fill(sa, sa + n, -1);
  buckets(true);
  for (int i = n1 - 1; i >= 0; --i) { const int j = sa1[(size_t)i]; sa[--bkt[(size_t)s[j]]] = j; }
  induce();
}

// The top level on bytes: symbol (byte + 1, 0 for the end of the string) in the low 9 bits of a 16-bit word, the suffix
// type in bit 15 (S-type) and bit 14 (leftmost S-type) -- the induction loops, whose text accesses are random, read one
// word per suffix instead of a symbol and a type.
void sais_bytes(unsigned short* s, int* sa, int n) {
  constexpr int K = 257;
  constexpr unsigned short kS = 0x8000u, kLms = 0x4000u, kSym = 0x01FFu;
  if (n == 1) { sa[0] = 0; return; }
  if (n == 2) { sa[0] = 1; sa[1] = 0; return; }
  s[n - 1] |= kS;
  for (int i = n - 2; i >= 0; --i) {
    const unsigned a = s[i] & kSym, b = s[i + 1] & kSym;
    if (a < b || (a == b && (s[i + 1] & kS))) s[i] |= kS;
  }
  for (int i = 1; i < n; ++i) if ((s[i] & kS) && !(s[i - 1] & kS)) s[i] |= kLms;
  int cnt[K] = {0}, bkt[K];
  for (int i = 0; i < n; ++i) ++cnt[s[i] & kSym];
  auto buckets = [&](bool ends) {
    int sum = 0;
    for (int c = 0; c < K; ++c) { sum += cnt[c]; bkt[c] = ends ? sum : sum - cnt[c]; }
  };
  auto induce = [&]() {
    buckets(false);
    for (int i = 0; i < n; ++i) {
      const int j = sa[i] - 1;
      if (j >= 0) { const unsigned v = s[j]; if (!(v & kS)) sa[bkt[v & kSym]++] = j; }
    }
    buckets(true);
    for (int i = n - 1; i >= 0; --i) {
      const int j = sa[i] - 1;
      if (j >= 0) { const unsigned v = s[j]; if (v & kS) sa[--bkt[v & kSym]] = j; }
    }
  };
  std::fill(sa, sa + n, -1);
  buckets(true);
  for (int i = 1; i < n; ++i) if (s[i] & kLms) sa[--bkt[s[i] & kSym]] = i;
  induce();
  int n1 = 0;
  for (int i = 0; i < n; ++i) if (sa[i] > 0 && (s[sa[i]] & kLms)) sa[n1++] = sa[i];
  std::fill(sa + n1, sa + n, -1);
  int names = 0, prev = -1;
  for (int i = 0; i < n1; ++i) {
    const int pos = sa[i];
    bool diff = prev < 0;
    for (int d = 0; !diff; ++d) {
      // (symbol and S bit compared together; the LMS bit ends a substring)
      if (((s[pos + d] ^ s[prev + d]) & (kSym | kS)) != 0) diff = true;
      else if (d > 0 && ((s[pos + d] | s[prev + d]) & kLms)) break;
    }
    if (diff) { ++names; prev = pos; }
    sa[n1 + pos / 2] = names - 1;
  }
  std::vector<int> s1((size_t)n1), sa1((size_t)n1);
  for (int i = n1, j = 0; i < n; ++i) if (sa[i] >= 0) s1[(size_t)j++] = sa[i];
  if (names < n1) sais(s1.data(), sa1.data(), n1, names);
  else for (int i = 0; i < n1; ++i) sa1[(size_t)s1[(size_t)i]] = i;
  for (int i = 1, j = 0; i < n; ++i) if (s[i] & kLms) s1[(size_t)j++] = i;
  for (int i = 0; i < n1; ++i) sa1[(size_t)i] = s1[(size_t)sa1[(size_t)i]];
  std::fill(sa, sa + n, -1);
  buckets(true);
  for (int i = n1 - 1; i >= 0; --i) { const int j = sa1[(size_t)i]; sa[--bkt[s[j] & kSym]] = j; }
  induce();
}

}  // namespace

// Suffix array of in[0..n) with the end of the string ordered before every byte.
std::vector<U32> suffix_array(const U8* in, U32 n) {
  std::vector<U32> out;
  if (!n) return out;
  std::vector<unsigned short> s((size_t)n + 1);           // byte + 1, the end of the string as the unique smallest symbol
  for (U32 i = 0; i < n; ++i) s[i] = (unsigned short)(in[i] + 1u);
  s[n] = 0;
  out.resize((size_t)n + 1);                              // sorted in place (entries are < 2^31), then the sentinel's
  sais_bytes(s.data(), (int*)out.data(), (int)n + 1);     // entry at the front is dropped
  out.erase(out.begin());
  return out;
}

// E8E9: x86 CALL/JMP targets made absolute (libzpaq.cpp:6450-6459).  Scans backward, in place.
void e8e9_forward(U8* buf, U32 n) {
  for (long i = (long)n - 5; i >= 0; --i) {
    if ((buf[i] & 254) == 0xe8 && ((buf[i + 4] + 1) & 254) == 0) {
      const unsigned a = (buf[i + 1] | buf[i + 2] << 8 | buf[i + 3] << 16) + (unsigned)i;
      buf[i + 1] = (U8)a;
      buf[i + 2] = (U8)(a >> 8);
      buf[i + 3] = (U8)(a >> 16);
    }
  }
}

// ... and back (what the E8E9 post-processor does, libzpaq.cpp:7302-7324): scans forward, in place.  Used where a filtered
// caller buffer has to be handed back as it came (zpq_preprocess_blocks_device when the device declines).
void e8e9_inverse(U8* buf, U32 n) {
  for (long i = 0; i + 4 < (long)n; ++i) {
    if ((buf[i] & 254) == 0xe8 && ((buf[i + 4] + 1) & 254) == 0) {
      const unsigned a = (buf[i + 1] | buf[i + 2] << 8 | buf[i + 3] << 16) - (unsigned)i;
      buf[i + 1] = (U8)a;
      buf[i + 2] = (U8)(a >> 8);
      buf[i + 3] = (U8)(a >> 16);
    }
  }
}

namespace {

// One block through LZ77 (level 1: bit-packed codes, level 2: byte-aligned codes).
class Lz77 {
 public:
  Lz77(const U8* in, U32 n, const int args[9], std::vector<U8>& out, const U32* sa = nullptr, bool emit_only = false)
      : in_(in), n_(n), out_(out), level_(args[1] & 3),
        use_sa_(args[5] - args[0] >= 21),
        checkbits_(use_sa_ ? 17 + args[0] : 12 - args[0]),
        min_match_((unsigned)args[2]), min_match2_((unsigned)args[3]), lookahead_((unsigned)args[6]),
        bucket_((1u << args[4]) - 1u),
        shift1_(args[2] > 0 ? (unsigned)((args[5] - 1) / args[2] + 1) : 1u),
        shift2_(args[3] > 0 ? (unsigned)((args[5] - 1) / args[3] + 1) : 0u),
        min_both_((int)std::max<unsigned>(min_match_, min_match2_ + lookahead_) + 4),
        rb_(args[0] > 4 ? (unsigned)(args[0] - 4) : 0u) {
    if ((min_match_ < 4 && level_ == 1) || (min_match_ < 1 && level_ == 2)) fail(ZPQ_E_ARG, "match length $3 too small");
    if (emit_only) return;                            // (emit_tokens only: nothing is searched)
    if (use_sa_) {
      if (sa) sa_.assign(sa, sa + n);                 // built on the device for the whole batch (device/sa_kernels.hip)
      else sa_ = suffix_array(in, n);
      // the inverse array for one aligned window of 2^checkbits positions at a time (like the reference: 8 MB that stay
      // in cache instead of 4 n bytes of scattered writes)
      isa_.assign((size_t)1 << checkbits_, 0);
      isa_window_ = 0xFFFFFFFFu;
    } else {
      if (args[5] < 1 || args[5] > 30) fail(ZPQ_E_ARG, "LZ77 hash table size out of range");
      ht_.assign((size_t)1 << args[5], 0);
    }
  }

  // The parse through a suffix array as a list of matches (what device/lz77_kernel.h produces for a whole batch): the
  // position the search stood at, the literals in front of the match that belong to it, its length and offset.
  void tokens(std::vector<LzToken>& toks) { toks_ = &toks; run(); toks_ = nullptr; }

  // The coded stream from such a list: literal runs between the matches, flushed after kMaxLiteral of them as run() does.
  void emit_tokens(const LzToken* toks, size_t ntok) {
    static const unsigned kMaxLiteral = (1u << 14) / 4;
    unsigned pos = 0, lit = 0;
    for (size_t t = 0; t <= ntok; ++t) {
      const unsigned stop = t < ntok ? toks[t].i : n_;
      if (stop < pos || stop > n_) fail(ZPQ_E_DEVICE, "LZ77 token list out of order");
      while (pos < stop) {                                       // literal steps: one byte each
        const unsigned take = std::min(stop - pos, kMaxLiteral - lit);
        lit += take;
        pos += take;
        if (lit >= kMaxLiteral) literals(pos, lit);
      }
      if (t == ntok) break;
      const LzToken& k = toks[t];
      if (k.off == 0 || k.off > k.i || k.len == 0 || (U64)k.i + k.blit + k.len > n_) fail(ZPQ_E_DEVICE, "LZ77 token out of range");
      lit += k.blit;
      literals(k.i + k.blit, lit);
      match(k.len, k.off);
      pos = k.i + k.blit + k.len;
    }
    literals(n_, lit);
    if (nbits_ > 0) out_.push_back((U8)bits_);
    bits_ = nbits_ = 0;
  }

  void run() {
    static const unsigned kMaxMatch = (1u << 14) * 3, kMaxLiteral = (1u << 14) / 4;
    const unsigned mask = (1u << checkbits_) - 1u;
    const unsigned htmask = use_sa_ ? 0u : (unsigned)ht_.size() - 1u;
    unsigned i = 0, lit = 0, h1 = 0, h2 = 0;
    auto at = [&](unsigned k) -> unsigned { return k < n_ ? in_[k] : 0u; };   // reads past the end see zeros
    while (i < n_) {
      unsigned blen = min_match_ - 1, bp = 0, blit = 0;
      int bscore = 0;
      if (use_sa_) {
        // Every step of the search is a dependent random access (inverse array -> suffix array -> text): the positions
        // a few steps ahead are sent on their way now -- their suffix array entries first, then, one stage later, the
        // text behind their nearest neighbours (8-10 % on a 1 MiB block; nothing on one that fits the cache).
        if ((i & ~mask) == isa_window_) {
          const unsigned ia = i + 8, ib = i + 4;
          if (ia < n_ && (ia & ~mask) == isa_window_) __builtin_prefetch(sa_.data() + isa_[ia & mask]);
          if (ib < n_ && (ib & ~mask) == isa_window_) {
            const unsigned qb = isa_[ib & mask];
            if (qb + 1 < n_) __builtin_prefetch(in_ + sa_[qb + 1]);
            if (qb >= 1) __builtin_prefetch(in_ + sa_[qb - 1]);
          }
        }
        // neighbours of position h + i in the suffix array are the longest matches of the text following it
        for (unsigned h = 0; h <= lookahead_; ++h) {
          // the reference keeps the inverse array for one aligned window of 2^checkbits positions at a time:
          // a look-ahead that leaves the window of i finds nothing
          if (h + i >= n_ || ((h + i) & ~mask) != (i & ~mask)) continue;
          if ((i & ~mask) != isa_window_) {
            isa_window_ = i & ~mask;
            for (U32 j = 0; j < n_; ++j) if ((sa_[j] & ~mask) == isa_window_) isa_[sa_[j] & mask] = j;
          }
          const unsigned q = isa_[(h + i) & mask];
          for (int dir = -1; dir <= 1; dir += 2) {
            for (unsigned k = 1; k <= bucket_; ++k) {
              const unsigned at_q = q + (unsigned)(dir * (int)k);
              unsigned p;
              // (the text behind a candidate is a random access: ask for the one six entries on now)
              { const unsigned nq = at_q + (unsigned)(dir * 6); if (nq < n_) __builtin_prefetch(in_ + sa_[nq]); }
              if (at_q < n_ && (p = sa_[at_q] - h) < i) {
                unsigned l, l1;
                l = h < std::min(n_ - i, kMaxMatch) ? match_end(p, i, h, std::min(n_ - i, kMaxMatch)) : h;
                for (l1 = h; l1 > 0 && in_[p + l1 - 1] == in_[i + l1 - 1]; --l1) {}
                int score = (int)(l - l1) * 8 - bit_length(i - p) - 4 * (lit == 0 && l1 > 0) - 11;
                for (unsigned a = 0; a < h; ++a) score = score * 5 / 8;
                if (score > bscore) { blen = l; bp = p; blit = l1; bscore = score; }
                if (l < blen || l < min_match_ || l > 255) break;
              }
            }
          }
          if (bscore <= 0 || blen < min_match_) break;
        }
      } else if (level_ == 1 || min_match_ <= 64) {
        if (min_match2_ > 0) {                                   // the longer context first
          for (unsigned k = 0; k <= bucket_; ++k) {
            unsigned p = ht_[h2 ^ k];
            if (p && (p & mask) == (at(i + 3) & mask)) {
              p >>= checkbits_;
              if (p < i && i + blen <= n_ && in_[p + blen - 1] == in_[i + blen - 1]) {
                unsigned l;
                l = lookahead_ < std::min(n_ - i, kMaxMatch) ? match_end(p, i, lookahead_, std::min(n_ - i, kMaxMatch)) : lookahead_;
                if (l >= min_match2_ + lookahead_) {
                  int l1;
                  for (l1 = (int)lookahead_; l1 > 0 && in_[p + l1 - 1] == in_[i + l1 - 1]; --l1) {}
                  const int score = (int)(l - (unsigned)l1) * 8 - bit_length(i - p) - 8 * (lit == 0 && l1 > 0) - 11;
                  if (score > bscore) { blen = l; bp = p; blit = (unsigned)l1; bscore = score; }
                }
              }
            }
            if (blen >= 128) break;
          }
        }
        if (!min_match2_ || blen < min_match2_) {
          for (unsigned k = 0; k <= bucket_; ++k) {
            unsigned p = ht_[h1 ^ k];
            if (p && i + 3 < n_ && (p & mask) == (in_[i + 3] & mask)) {
              p >>= checkbits_;
              if (p < i && i + blen <= n_ && in_[p + blen - 1] == in_[i + blen - 1]) {
                unsigned l;
                l = match_end(p, i, 0, std::min(n_ - i, kMaxMatch));
                const int score = (int)l * 8 - bit_length(i - p) - 2 * (lit > 0) - 11;
                if (score > bscore) { blen = l; bp = p; blit = 0; bscore = score; }
              }
            }
            if (blen >= 128) break;
          }
        }
      }
      // a match that pays for its code goes out after the literals that precede it
      const unsigned off = i - bp;
      if (off > 0 && bscore > 0 &&
          blen - blit >= min_match_ + (level_ == 2) * ((off >= (1u << 16)) + (off >= (1u << 24)))) {
        if (toks_) { toks_->push_back(LzToken{i, off, blen - blit, blit}); lit = 0; }
        else {
          lit += blit;
          literals(i + blit, lit);
          match(blen - blit, off);
        }
      } else {
        blen = 1;
        ++lit;
      }
      if (use_sa_) i += blen;
      else {
        while (blen--) {                                          // index the bytes stepped over
          if (i + (unsigned)min_both_ < n_) {
            const unsigned ih = ((i * 1234547u) >> 19) & bucket_;
            const unsigned p = (i << checkbits_) | (in_[i + 3] & mask);
            if (min_match2_) {
              ht_[h2 ^ ih] = p;
              h2 = (((h2 * 9) << shift2_) + (in_[i + min_match2_ + lookahead_] + 1u) * 23456789u) & htmask;
            }
            ht_[h1 ^ ih] = p;
            h1 = (((h1 * 5) << shift1_) + (in_[i + min_match_] + 1u) * 123456791u) & htmask;
          }
          ++i;
        }
      }
      if (lit >= kMaxLiteral) { if (toks_) lit = 0; else literals(i, lit); }
    }
    if (toks_) return;
    literals(n_, lit);
    if (nbits_ > 0) out_.push_back((U8)bits_);
    bits_ = nbits_ = 0;
  }

 private:
  // first l >= from with in[p + l] != in[i + l], at most `limit` (p < i, i + limit <= n): 8 bytes per step
  unsigned match_end(unsigned p, unsigned i, unsigned from, unsigned limit) const {
    unsigned l = from;
    while (l + 8 <= limit) {
      unsigned long long a, b;
      memcpy(&a, in_ + p + l, 8);
      memcpy(&b, in_ + i + l, 8);
      if (a != b) return l + (unsigned)(__builtin_ctzll(a ^ b) >> 3);
      l += 8;
    }
    while (l < limit && in_[p + l] == in_[i + l]) ++l;
    return l;
  }
  void putb(unsigned x, int k) {                                  // k bits of x, least significant first
    x &= (1u << k) - 1u;
    bits_ |= x << nbits_;
    nbits_ += (unsigned)k;
    while (nbits_ > 7) { out_.push_back((U8)bits_); bits_ >>= 8; nbits_ -= 8; }
  }
  // in[i - lit .. i - 1] as literals
  void literals(unsigned i, unsigned& lit) {
    if (level_ == 1) {
      if (lit < 1) return;
      int ll = bit_length(lit);
      putb(0, 2);
      --ll;
      while (--ll >= 0) { putb(1, 1); putb((lit >> ll) & 1u, 1); }   // interleaved Elias gamma, leading 1 implied
      putb(0, 1);
      while (lit) putb(in_[i - lit--], 8);
    } else {
      while (lit > 0) {
        const unsigned run = std::min(lit, 64u);
        out_.push_back((U8)(run - 1));
        for (unsigned j = i - lit; j < i - lit + run; ++j) out_.push_back(in_[j]);
        lit -= run;
      }
    }
  }
  void match(unsigned len, unsigned off) {
    if (level_ == 1) {             // mm,mmm,n,ll,r,q: length 4n+ll at offset ((q-1) << rb) + r + 1
      int ll = bit_length(len) - 1;
      off += (1u << rb_) - 1u;
      const int lo = bit_length(off) - 1 - (int)rb_;
      putb((unsigned)(lo + 8) >> 3, 2);
      putb((unsigned)lo & 7u, 3);
      while (--ll >= 2) { putb(1, 1); putb((len >> ll) & 1u, 1); }
      putb(0, 1);
      putb(len & 3u, 2);
      putb(off, (int)rb_);
      putb(off >> rb_, lo);
    } else {                       // yyxxxxxx + y+1 offset bytes, length x + minimum match; long matches are split
      --off;
      while (len > 0) {
        const unsigned len1 = len > min_match_ * 2 + 63 ? min_match_ + 63 : (len > min_match_ + 63 ? len - min_match_ : len);
        if (off < (1u << 16)) {
          out_.push_back((U8)(64 + len1 - min_match_));
          out_.push_back((U8)(off >> 8));
          out_.push_back((U8)off);
        } else if (off < (1u << 24)) {
          out_.push_back((U8)(128 + len1 - min_match_));
          out_.push_back((U8)(off >> 16));
          out_.push_back((U8)(off >> 8));
          out_.push_back((U8)off);
        } else {
          out_.push_back((U8)(192 + len1 - min_match_));
          out_.push_back((U8)(off >> 24));
          out_.push_back((U8)(off >> 16));
          out_.push_back((U8)(off >> 8));
          out_.push_back((U8)off);
        }
        len -= len1;
      }
    }
  }

  const U8* in_;
  const U32 n_;
  std::vector<U8>& out_;
  const int level_;
  const bool use_sa_;
  const int checkbits_;
  const unsigned min_match_, min_match2_, lookahead_, bucket_, shift1_, shift2_;
  const int min_both_;
  const unsigned rb_;
  unsigned bits_ = 0, nbits_ = 0;
  std::vector<U32> ht_, sa_, isa_;
  unsigned isa_window_ = 0xFFFFFFFFu;     // first position of the window isa_ holds
  std::vector<LzToken>* toks_ = nullptr;  // tokens(): matches are listed instead of coded
};

}  // namespace

// What compressBlock feeds the coder for a method with args[1] != 0 (libzpaq.cpp:7709-7716).  `data` is modified in
// place where the reference modifies its input buffer (E8E9).  Returns true when `out` holds the stream to code,
// false when the (possibly E8E9-filtered) input itself is coded.
bool preprocess_needs_suffix_array(const int args[9]) {
  const int kind = args[1];
  if (kind < 1 || kind > 7 || kind == 4) return false;
  const int level = kind & 3;
  return level == 3 || ((level == 1 || level == 2) && args[5] - args[0] >= 21);
}

bool preprocess_block(U8* data, U32 n, const int args[9], std::vector<U8>& out, const U32* sa_in, bool e8e9_done) {
  out.clear();
  const int kind = args[1];
  if (kind < 1 || kind > 7) return false;
  if (kind == 4) { if (!e8e9_done) e8e9_forward(data, n); return false; }
  if (kind > 4 && !e8e9_done) e8e9_forward(data, n);
  const int level = kind & 3;
  if (level == 3) {                                  // BWT: last column, end-of-string as 255, its index in 4 bytes
    std::vector<U32> own;
    if (!sa_in) own = suffix_array(data, n);
    const U32* sa = sa_in ? sa_in : own.data();
    out.reserve((size_t)n + 5);
    U32 idx = 0;
    out.push_back(n > 0 ? data[n - 1] : 255);
    for (U32 i = 1; i <= n; ++i) {
      if (sa[i - 1] == 0) { idx = i; out.push_back(255); }
      else out.push_back(data[sa[i - 1] - 1]);
    }
    for (int k = 0; k < 4; ++k) { out.push_back((U8)idx); idx >>= 8; }
    return true;
  }
  out.reserve((size_t)n / 2 + 64);
  Lz77 lz(data, n, args, out, sa_in);
  if (args[5] - args[0] >= 21) {                     // through a suffix array: the parse as a token list, then the coder --
    std::vector<LzToken> toks;                       // the same coder takes the device's list (lz77_serialize)
    lz.tokens(toks);
    lz.emit_tokens(toks.data(), toks.size());
  } else lz.run();
  return true;
}

void lz77_host_tokens(const U8* data, U32 n, const int args[9], const U32* sa, std::vector<LzToken>& toks) {
  toks.clear();
  if ((args[1] & 3) < 1 || (args[1] & 3) > 2 || args[5] - args[0] < 21) fail(ZPQ_E_ARG, "not an LZ77 method that searches a suffix array");
  std::vector<U8> unused;
  Lz77 lz(data, n, args, unused, sa);
  lz.tokens(toks);
}

void lz77_serialize(const U8* data, U32 n, const int args[9], const LzToken* toks, size_t ntok, std::vector<U8>& out) {
  out.clear();
  if ((args[1] & 3) < 1 || (args[1] & 3) > 2) fail(ZPQ_E_ARG, "not an LZ77 method");
  out.reserve((size_t)n / 2 + 64);
  Lz77 lz(data, n, args, out, nullptr, true);
  lz.emit_tokens(toks, ntok);
}

}  // namespace zpq
