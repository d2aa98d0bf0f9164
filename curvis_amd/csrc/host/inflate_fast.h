/* inflate_fast.h -- a DEFLATE (RFC 1951) / zlib (RFC 1950) decoder for the PNG backgrounds of `curvis image|video`.
 *
 * Why not zlib's inflate(): once the per-pixel path had left the CPU, loading the two 8192x4096 sky textures was most of what
 * a user waits for -- and most of THAT was zlib inflating 134 MB per texture at 150-600 MB/s (a byte-wise bit reader, a
 * 9-bit primary table, a copy of every output byte into its private window between calls).  This decoder works the way the
 * fast modern ones do (libdeflate's design; written here from RFC 1951): a 64-bit bit buffer refilled a word at a time, an
 * 11-bit litlen table whose entries carry base value and extra-bit count, the output buffer itself as the window, matches
 * copied a word at a time.  The whole stream must be in memory, contiguous, with kInputPadding readable bytes behind it.
 *
 * The reference decodes PNG through the `png` crate (image 0.25.2 -> png 0.17.13 -> fdeflate / miniz_oxide); any conforming
 * inflater yields the same bytes.  tests/test_inflate_host.py compares this one with zlib on thousands of streams (every
 * compression level and strategy, stored / fixed / dynamic blocks, corrupted and truncated input) through the host twin.
 *
 * Untrusted input: every read of the output window and every write is bounds-checked against the caller's buffer; reads of
 * the input run at most kInputPadding bytes past its end (zeros), which is detected and reported as truncation. */
#ifndef CURVIS_INFLATE_FAST_H
#define CURVIS_INFLATE_FAST_H

#include <cstddef>
#include <cstdint>
#include <cstring>

namespace cvinflate {

enum : int { OK = 0, E_DATA = -1, E_TRUNCATED = -2, E_OUTPUT_FULL = -3, E_HEADER = -4 };
enum : size_t { kInputPadding = 64 }; /* the bit reader looks ahead: at most 8 + 2 x 7 + 8 bytes past the end, see the checks below */

namespace detail {

enum : unsigned {
  kLitlenBits = 11, /* primary table: 2048 entries (8 KiB) -- every literal code of a typical PNG block fits */
  kDistBits = 8,
  kLitlenEnough = 2048 + 1024, /* primary + sub-tables: codes of up to 15 bits behind an 11-bit prefix need at most 2^4 entries
                                  each; at most 286 symbols => far fewer than 1024 sub-table entries in total (checked at build) */
  kDistEnough = 256 + 512,
  /* entry layout: bits 0-7 bits to consume, bits 8-11 extra-bit count (or sub-table index bits), bits 12-15 kind, bits 16-31 payload */
  kLiteral = 0x1000u,
  kEndOfBlock = 0x2000u,
  kSubtable = 0x4000u,
  kBase = 0x8000u, /* a length / distance symbol: payload = base value */
};

struct Tables {
  uint32_t litlen[kLitlenEnough];
  uint32_t dist[kDistEnough];
};

inline unsigned reverse_bits(unsigned code, unsigned len) {
  unsigned r = 0;
  for (unsigned i = 0; i < len; ++i) r |= ((code >> i) & 1u) << (len - 1 - i);
  return r;
}

/* canonical Huffman code (RFC 1951 3.2.2) -> decode table indexed by the next `primary` bits of the stream (LSB first).
 * payload[sym] / kind[sym] describe what a symbol decodes to.  Returns false for an over-subscribed code, or an incomplete one
 * (but for the two cases the format allows: no code at all, and a single code of one bit -- the unused patterns stay invalid). */
inline bool build_table(uint32_t *table, unsigned table_cap, unsigned primary, const uint8_t *lens, unsigned n_syms,
                        const uint32_t *sym_entry /* entry of each symbol without its length field */) {
  unsigned count[16] = {0};
  for (unsigned s = 0; s < n_syms; ++s) count[lens[s]]++;
  unsigned used = n_syms - count[0];
  /* Kraft sum in units of 2^-15 */
  unsigned long kraft = 0;
  for (unsigned l = 1; l <= 15; ++l) kraft += (unsigned long)count[l] << (15 - l);
  if (kraft > (1ul << 15)) return false;
  if (kraft < (1ul << 15) && !(used == 0 || (used == 1 && count[1] == 1))) return false;
  unsigned next_code[16];
  {
    unsigned code = 0;
    count[0] = 0;
    for (unsigned l = 1; l <= 15; ++l) {
      code = (code + count[l - 1]) << 1;
      next_code[l] = code;
    }
  }
  for (unsigned i = 0; i < (1u << primary); ++i) table[i] = 0; /* invalid */
  /* pass 1: codes no longer than the primary index -- replicated; and, per primary prefix, the longest code behind it */
  uint8_t sub_bits[1u << kLitlenBits]; /* primary <= kLitlenBits */
  std::memset(sub_bits, 0, (size_t)1 << primary);
  unsigned code_of[288 + 32];
  for (unsigned s = 0; s < n_syms; ++s) {
    const unsigned l = lens[s];
    if (!l) continue;
    const unsigned rev = reverse_bits(next_code[l]++, l);
    code_of[s] = rev;
    if (l <= primary) {
      const uint32_t e = sym_entry[s] | l;
      for (unsigned i = rev; i < (1u << primary); i += 1u << l) table[i] = e;
    } else {
      const unsigned prefix = rev & ((1u << primary) - 1);
      if (l - primary > sub_bits[prefix]) sub_bits[prefix] = (uint8_t)(l - primary);
    }
  }
  /* pass 2: sub-tables */
  unsigned next_free = 1u << primary;
  for (unsigned p = 0; p < (1u << primary); ++p) {
    if (!sub_bits[p]) continue;
    const unsigned size = 1u << sub_bits[p];
    if (next_free + size > table_cap) return false;
    table[p] = ((uint32_t)next_free << 16) | kSubtable | ((uint32_t)sub_bits[p] << 8) | primary;
    for (unsigned i = 0; i < size; ++i) table[next_free + i] = 0;
    next_free += size;
  }
  for (unsigned s = 0; s < n_syms; ++s) {
    const unsigned l = lens[s];
    if (l <= primary) continue;
    const unsigned rev = code_of[s], prefix = rev & ((1u << primary) - 1);
    const unsigned start = table[prefix] >> 16, bits = sub_bits[prefix], rest = l - primary;
    const uint32_t e = sym_entry[s] | rest;
    for (unsigned i = rev >> primary; i < (1u << bits); i += 1u << rest) table[start + i] = e;
  }
  return true;
}

/* what the symbols decode to (RFC 1951 3.2.5) */
struct SymbolEntries {
  uint32_t litlen[288];
  uint32_t dist[32];
  SymbolEntries() {
    static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (unsigned s = 0; s < 256; ++s) litlen[s] = ((uint32_t)s << 16) | kLiteral;
    litlen[256] = kEndOfBlock;
    for (unsigned s = 257; s < 286; ++s) litlen[s] = ((uint32_t)len_base[s - 257] << 16) | kBase | ((uint32_t)len_extra[s - 257] << 8);
    litlen[286] = litlen[287] = 0; /* never valid in a stream: an entry without a kind is a data error when it is met */
    for (unsigned s = 0; s < 30; ++s) dist[s] = ((uint32_t)dist_base[s] << 16) | kBase | ((uint32_t)dist_extra[s] << 8);
    dist[30] = dist[31] = 0;
  }
};

inline uint64_t load_le64(const uint8_t *p) {
  uint64_t v;
  std::memcpy(&v, p, 8);
#if defined(__BYTE_ORDER__) && __BYTE_ORDER__ == __ORDER_BIG_ENDIAN__
  v = __builtin_bswap64(v);
#endif
  return v;
}

}  // namespace detail

/* Progress: called with the number of output bytes that are FINAL AND NO LONGER READ by the decoder (everything more than
 * 32 KiB behind the write position: a match reaches back at most that far), roughly every MiB and at the end -- a caller may
 * rewrite those bytes in place while decoding goes on. */
struct Progress {
  void (*fn)(void *user, size_t settled_bytes) = nullptr;
  void *user = nullptr;
};

/* raw DEFLATE stream in[0, in_len) (+ kInputPadding readable bytes) -> out[0, out_cap).  *out_len = bytes produced,
 * *in_used = bytes of the stream consumed (whole bytes: the stream ends on the byte holding the last bit). */
inline int inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *out_len, size_t *in_used, const Progress &progress = Progress()) {
  using namespace detail;
  static const SymbolEntries syms;
  Tables *T = new Tables; /* 20 KiB: not on a helper thread's stack */
  struct Free {
    Tables *t;
    ~Free() { delete t; }
  } free_tables{T};
  const uint8_t *const in_end = in + in_len;
  const uint8_t *ip = in;
  uint8_t *op = out, *const out_end = out + out_cap;
  uint64_t bitbuf = 0;
  unsigned bitcnt = 0;
  size_t next_report = (size_t)1 << 20;
  auto report = [&](bool final) {
    if (!progress.fn) return;
    const size_t have = (size_t)(op - out);
    progress.fn(progress.user, final ? have : (have > 32768 ? have - 32768 : 0));
  };
#define CV_REFILL()                                    \
  do {                                                 \
    bitbuf |= load_le64(ip) << bitcnt;                 \
    ip += (63 - bitcnt) >> 3;                          \
    bitcnt |= 56;                                      \
  } while (0)
#define CV_TAKE(n) (bitbuf >>= (n), bitcnt -= (n))
  int last = 0;
  bool have_fixed = false;
  do {
    if (ip > in_end + 8) return E_TRUNCATED; /* running on padding */
    CV_REFILL();
    last = (int)(bitbuf & 1);
    const unsigned type = (unsigned)(bitbuf >> 1) & 3;
    CV_TAKE(3);
    if (type == 0) { /* stored: to the byte boundary, LEN, ~LEN, bytes */
      CV_TAKE(bitcnt & 7);
      ip -= bitcnt >> 3; /* give the whole bytes in the bit buffer back */
      bitbuf = 0;
      bitcnt = 0;
      if (ip > in_end || in_end - ip < 4) return E_TRUNCATED;
      const unsigned len = ip[0] | (ip[1] << 8), nlen = ip[2] | (ip[3] << 8);
      ip += 4;
      if ((len ^ nlen) != 0xFFFFu) return E_DATA;
      if ((size_t)(in_end - ip) < len) return E_TRUNCATED;
      if ((size_t)(out_end - op) < len) return E_OUTPUT_FULL;
      if (len) std::memcpy(op, ip, len); /* an empty stored block (a sync flush) may meet an empty output buffer: no null to memcpy */
      ip += len;
      op += len;
      have_fixed = false; /* (the tables are untouched; nothing to do) */
      if ((size_t)(op - out) >= next_report) {
        report(false);
        next_report = (size_t)(op - out) + ((size_t)1 << 20);
      }
      continue;
    }
    if (type == 3) return E_DATA;
    if (type == 1) {
      if (!have_fixed) {
        uint8_t lens[288 + 32];
        for (unsigned i = 0; i < 144; ++i) lens[i] = 8;
        for (unsigned i = 144; i < 256; ++i) lens[i] = 9;
        for (unsigned i = 256; i < 280; ++i) lens[i] = 7;
        for (unsigned i = 280; i < 288; ++i) lens[i] = 8;
        for (unsigned i = 0; i < 32; ++i) lens[288 + i] = 5;
        if (!build_table(T->litlen, kLitlenEnough, kLitlenBits, lens, 288, syms.litlen)) return E_DATA;
        if (!build_table(T->dist, kDistEnough, kDistBits, lens + 288, 32, syms.dist)) return E_DATA;
        have_fixed = true;
      }
    } else { /* dynamic: the code lengths, themselves Huffman coded (RFC 1951 3.2.7) */
      have_fixed = false;
      const unsigned hlit = (unsigned)(bitbuf & 31) + 257, hdist = (unsigned)((bitbuf >> 5) & 31) + 1, hclen = (unsigned)((bitbuf >> 10) & 15) + 4;
      CV_TAKE(14);
      if (hlit > 286 || hdist > 30) return E_DATA;
      static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      uint8_t cl_lens[19] = {0};
      for (unsigned i = 0; i < hclen; ++i) {
        if (bitcnt < 3) CV_REFILL();
        cl_lens[order[i]] = (uint8_t)(bitbuf & 7);
        CV_TAKE(3);
      }
      uint32_t cl_entry[19], cl_table[128];
      for (unsigned s = 0; s < 19; ++s) cl_entry[s] = ((uint32_t)s << 16) | kLiteral;
      if (!build_table(cl_table, 128, 7, cl_lens, 19, cl_entry)) return E_DATA;
      uint8_t lens[288 + 32];
      unsigned n = 0;
      const unsigned total = hlit + hdist;
      while (n < total) {
        if (ip > in_end + 8) return E_TRUNCATED;
        CV_REFILL();
        const uint32_t e = cl_table[bitbuf & 127];
        if (!e) return E_DATA;
        CV_TAKE(e & 0xFF);
        const unsigned sym = e >> 16;
        if (sym < 16) {
          lens[n++] = (uint8_t)sym;
          continue;
        }
        unsigned rep, val = 0;
        if (sym == 16) {
          if (!n) return E_DATA;
          val = lens[n - 1];
          rep = 3 + (unsigned)(bitbuf & 3);
          CV_TAKE(2);
        } else if (sym == 17) {
          rep = 3 + (unsigned)(bitbuf & 7);
          CV_TAKE(3);
        } else {
          rep = 11 + (unsigned)(bitbuf & 127);
          CV_TAKE(7);
        }
        if (n + rep > total) return E_DATA;
        std::memset(lens + n, (int)val, rep);
        n += rep;
      }
      if (!lens[256]) return E_DATA; /* a block without an end-of-block code cannot end */
      uint8_t ll[288], dl[32];
      std::memset(ll, 0, sizeof ll);
      std::memset(dl, 0, sizeof dl);
      std::memcpy(ll, lens, hlit);
      std::memcpy(dl, lens + hlit, hdist);
      if (!build_table(T->litlen, kLitlenEnough, kLitlenBits, ll, 288, syms.litlen)) return E_DATA;
      if (!build_table(T->dist, kDistEnough, kDistBits, dl, 32, syms.dist)) return E_DATA;
    }
    /* the symbols of the block */
    const uint32_t *const LT = T->litlen, *const DT = T->dist;
    for (;;) {
      if (ip > in_end + 8) return E_TRUNCATED;
      CV_REFILL(); /* >= 56 bits: a litlen code (15) + its extra bits (5) + a distance code (15) + its extra bits (13) = 48 */
      uint32_t e = LT[bitbuf & ((1u << kLitlenBits) - 1)];
      if (e & kLiteral) { /* the common case first; up to three literals per refill (3 x 15 < 56) */
        if (op >= out_end) return E_OUTPUT_FULL;
        CV_TAKE(e & 0xFF);
        *op++ = (uint8_t)(e >> 16);
        e = LT[bitbuf & ((1u << kLitlenBits) - 1)];
        if (e & kLiteral) {
          if (op >= out_end) return E_OUTPUT_FULL;
          CV_TAKE(e & 0xFF);
          *op++ = (uint8_t)(e >> 16);
          e = LT[bitbuf & ((1u << kLitlenBits) - 1)];
          if (e & kLiteral) {
            if (op >= out_end) return E_OUTPUT_FULL;
            CV_TAKE(e & 0xFF);
            *op++ = (uint8_t)(e >> 16);
            continue;
          }
        }
        /* `e` is the entry of the NEXT symbol, not yet consumed; bits left: >= 56 - 30 = 26: a length / end-of-block symbol with a
         * distance behind it may need 48 */
        if (bitcnt < 48) CV_REFILL();
      }
      if (e & kSubtable) {
        CV_TAKE(kLitlenBits);
        e = LT[(e >> 16) + (bitbuf & ((1u << ((e >> 8) & 0xF)) - 1))];
        if (e & kLiteral) {
          if (op >= out_end) return E_OUTPUT_FULL;
          CV_TAKE(e & 0xFF);
          *op++ = (uint8_t)(e >> 16);
          continue;
        }
      }
      if (e & kEndOfBlock) {
        CV_TAKE(e & 0xFF);
        break;
      }
      if (!(e & kBase)) return E_DATA; /* an unused pattern of an incomplete code, or symbol 286 / 287 */
      CV_TAKE(e & 0xFF);
      unsigned xb = (e >> 8) & 0xF;
      const unsigned len = (e >> 16) + (unsigned)(bitbuf & ((1u << xb) - 1));
      CV_TAKE(xb);
      uint32_t d = DT[bitbuf & ((1u << kDistBits) - 1)];
      if (d & kSubtable) {
        CV_TAKE(kDistBits);
        d = DT[(d >> 16) + (bitbuf & ((1u << ((d >> 8) & 0xF)) - 1))];
      }
      if (!(d & kBase)) return E_DATA;
      CV_TAKE(d & 0xFF);
      xb = (d >> 8) & 0xF;
      const size_t dist = (d >> 16) + (size_t)(bitbuf & ((1u << xb) - 1));
      CV_TAKE(xb);
      if (dist > (size_t)(op - out)) return E_DATA; /* before the start of the output */
      if ((size_t)(out_end - op) < len) return E_OUTPUT_FULL;
      const uint8_t *src = op - dist;
      if (dist >= 8 && (size_t)(out_end - op) >= (size_t)len + 8) { /* whole words; may write up to 7 bytes past the match, inside the buffer */
        uint8_t *dst = op;
        const uint8_t *const stop = op + len;
        do {
          std::memcpy(dst, src, 8);
          dst += 8;
          src += 8;
        } while (dst < stop);
        op += len;
      } else if (dist == 1) {
        std::memset(op, *src, len);
        op += len;
      } else {
        for (unsigned k = 0; k < len; ++k) op[k] = src[k]; /* overlapping, bytewise: the pattern repeats */
        op += len;
      }
      if ((size_t)(op - out) >= next_report) {
        report(false);
        next_report = (size_t)(op - out) + ((size_t)1 << 20);
      }
    }
  } while (!last);
#undef CV_REFILL
#undef CV_TAKE
  /* bytes really consumed: what the pointer has passed minus the whole bytes still unread in the bit buffer */
  const size_t passed = (size_t)(ip - in), unread = bitcnt >> 3;
  if (passed < unread) return E_DATA;
  const size_t used = passed - unread;
  if (used > in_len) return E_TRUNCATED;
  if (out_len) *out_len = (size_t)(op - out);
  if (in_used) *in_used = used;
  report(true);
  return OK;
}

/* zlib wrapper (RFC 1950): 2-byte header, DEFLATE data, Adler-32 of the output (big endian).  The checksum is RETURNED
 * (*adler_stored), not verified: the caller may be rewriting the output in place as it settles and sums it on the way. */
inline int inflate_zlib(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *out_len, uint32_t *adler_stored,
                        const Progress &progress = Progress()) {
  if (in_len < 6) return E_TRUNCATED;
  const unsigned cmf = in[0], flg = in[1];
  if ((cmf & 15) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20)) return E_HEADER; /* deflate, window <= 32 KiB, no preset dictionary */
  size_t used = 0;
  const int rc = inflate_raw(in + 2, in_len - 2, out, out_cap, out_len, &used, progress);
  if (rc != OK) return rc;
  if (in_len - 2 - used < 4) return E_TRUNCATED;
  const uint8_t *a = in + 2 + used;
  if (adler_stored) *adler_stored = ((uint32_t)a[0] << 24) | ((uint32_t)a[1] << 16) | ((uint32_t)a[2] << 8) | a[3];
  return OK;
}

}  // namespace cvinflate

#endif /* CURVIS_INFLATE_FAST_H */
