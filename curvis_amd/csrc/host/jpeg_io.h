/* jpeg_io.h -- JPEG backgrounds for the `curvis` host binary (README.md of the reference suggests `.jpg` star maps;
 * src/images.rs:7-9 load_image = image::open -> image 0.25.2 -> zune-jpeg 0.4.13, neither under /root/reference).
 *
 * A from-scratch decoder for what such files are in practice: 8-bit Huffman JPEG, baseline / extended sequential
 * (SOF0, SOF1) and progressive (SOF2), 1 component (grey) or 3 (YCbCr, or RGB when the Adobe APP14 segment says
 * so), sampling factors 1 or 2 in each direction (4:4:4, 4:2:2, 4:4:0, 4:2:0), restart intervals.  Arithmetic
 * coding, 12-bit precision, lossless, hierarchical and 4-component (CMYK) files are rejected with a message.
 *
 * The arithmetic follows the published design zune-jpeg's scalar path shares with stb_image (public domain): the
 * 12-bit fixed-point 8x8 IDCT (constants 2217, -7567, 3135, ... = f2f(0.5411961) ...; +512 >> 10 after the column
 * pass, 65536 + (128 << 17) >> 17 after the row pass), triangle-filter chroma upsampling ((3 near + far + 2) >> 2,
 * and (9 a + 3 b + 3 c + d + 8) >> 4 for 2x2), and zune-jpeg's integer colour conversion
 * (r = y + (45 cr >> 5), g = y - ((11 cb + 23 cr) >> 5), b = y + (113 cb >> 6)).  That is a restatement FROM
 * KNOWLEDGE of those crates; it cannot be checked against them in this image, so JPEG input is outside the pixel-
 * parity claims (SURVEY.md 8f N2 recommends PNG skies for parity).  tests/test_cli_host.py compares it with
 * libjpeg (Pillow) within the tolerance two conforming decoders differ by.
 */
#ifndef CURVIS_JPEG_IO_H
#define CURVIS_JPEG_IO_H

#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>
#include <thread>
#include <system_error>

#include <new>

#include "png_io.h"

namespace jpegio {

static const uint8_t kZigzag[64 + 15] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                         6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                         39, 46, 53, 60, 61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huff {
  bool present = false;
  uint8_t lookup_len[512];  /* 9-bit fast table: code length (0 = longer than 9 bits) */
  uint8_t lookup_sym[512];
  int32_t maxcode[18];      /* left-aligned to 16 bits, +1; maxcode[17] sentinel */
  int32_t delta[17];
  uint16_t code[256];
  uint8_t size[257], values[256];
  bool build(const uint8_t counts[16], const uint8_t *vals, int n) {
    int k = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < counts[i]; ++j) size[k++] = (uint8_t)(i + 1);
    size[k] = 0;
    if (k != n || n > 256) return false;
    std::memcpy(values, vals, (size_t)n);
    int c = 0;
    k = 0;
    for (int j = 1; j <= 16; ++j) {
      delta[j] = k - c;
      if (size[k] == j) {
        while (size[k] == j) code[k++] = (uint16_t)c++;
        if (c - 1 >= (1 << j)) return false;
      }
      maxcode[j] = c << (16 - j);
      c <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    std::memset(lookup_len, 0, sizeof lookup_len);
    for (int i = 0; i < k; ++i) {
      const int s = size[i];
      if (s <= 9) {
        const int first = code[i] << (9 - s), cnt = 1 << (9 - s);
        for (int j = 0; j < cnt; ++j) {
          lookup_len[first + j] = (uint8_t)s;
          lookup_sym[first + j] = values[i];
        }
      }
    }
    present = true;
    return true;
  }
};

struct Component {
  int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0, dc_pred = 0;
  int blocks_w = 0, blocks_h = 0;          /* allocated size in blocks (whole MCUs) */
  int w = 0, h_px = 0;                     /* true size in pixels of this component */
  struct Zeroed { /* blocks_w * blocks_h * 64 coefficients, zero to begin with: calloc -- zero pages straight from the kernel, no pass
                     over 100-200 MB that the entropy decoder then writes once */
    int16_t *p = nullptr;
    size_t n = 0;
    Zeroed() = default;
    Zeroed(const Zeroed &) = delete;
    Zeroed &operator=(const Zeroed &) = delete;
    void alloc(size_t count) {
      release();
      if (!count) return;
      p = (int16_t *)std::calloc(count, sizeof(int16_t));
      if (!p) throw std::bad_alloc();
      n = count;
    }
    void release() {
      std::free(p);
      p = nullptr;
      n = 0;
    }
    int16_t &operator[](size_t i) { return p[i]; }
    const int16_t &operator[](size_t i) const { return p[i]; }
    ~Zeroed() { release(); }
  } coef;
  std::vector<uint8_t> plane;              /* blocks_w*8 x blocks_h*8 */
};

struct Decoder {
  const uint8_t *p, *end;
  size_t file_bytes = 0;
  std::string *err;
  uint16_t qt[4][64];
  bool qt_present[4] = {false, false, false, false};
  Huff hdc[4], hac[4];
  Component comp[3];
  int ncomp = 0, W = 0, H = 0, hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
  bool progressive = false, have_sof = false;
  int restart_interval = 0, adobe_transform = -1;
  bool jfif = false;
  /* bit reader */
  uint32_t bitbuf = 0;
  int bitcnt = 0;
  bool hit_marker = false;
  uint8_t marker = 0;
  int eobrun = 0;

  bool fail(const char *m) {
    if (err->empty()) *err = m;
    return false;
  }
  void fill() {
    while (bitcnt <= 24) {
      uint32_t b = 0;
      if (!hit_marker && p < end) {
        b = *p++;
        if (b == 0xFF) {
          uint8_t c = p < end ? *p : 0xD9;
          while (c == 0xFF && p + 1 < end) c = *++p; /* fill bytes */
          if (c == 0) {
            ++p; /* stuffed zero */
          } else {
            marker = c;
            hit_marker = true;
            if (p < end) ++p;
            b = 0;
          }
        }
      }
      bitbuf |= b << (24 - bitcnt);
      bitcnt += 8;
    }
  }
  int get_bits(int n) {
    if (n == 0) return 0;
    if (bitcnt < n) fill();
    const int v = (int)(bitbuf >> (32 - n));
    bitbuf <<= n;
    bitcnt -= n;
    return v;
  }
  int get_bit() { return get_bits(1); }
  static int extend(int v, int n) { return (n && v < (1 << (n - 1))) ? v - (1 << n) + 1 : v; } /* T.81 F.12 */
  int receive_extend(int n) { return extend(get_bits(n), n); }
  int decode_huff(const Huff &h) {
    if (bitcnt < 16) fill();
    const int c = (int)(bitbuf >> 23);
    int k = h.lookup_len[c];
    if (k) {
      bitbuf <<= k;
      bitcnt -= k;
      return h.lookup_sym[c];
    }
    const int32_t temp = (int32_t)(bitbuf >> 16);
    for (k = 10; k <= 16; ++k)
      if (temp < h.maxcode[k]) break;
    if (k == 17) return -1;
    const int idx = (int)((bitbuf >> (32 - k)) & ((1u << k) - 1)) + h.delta[k];
    if (idx < 0 || idx >= 256) return -1;
    bitbuf <<= k;
    bitcnt -= k;
    return h.values[idx];
  }
  void reset_entropy() {
    bitbuf = 0;
    bitcnt = 0;
    hit_marker = false;
    marker = 0;
    eobrun = 0;
    for (int i = 0; i < ncomp; ++i) comp[i].dc_pred = 0;
  }

  /* ---- block decoders: coefficients are stored UN-dequantised, natural order ---- */
  bool block_baseline(int16_t *b, Component &c) {
    const int t = decode_huff(hdc[c.hd]);
    if (t < 0 || t > 15) return fail("bad Huffman code (DC)");
    c.dc_pred += t ? receive_extend(t) : 0;
    b[0] = (int16_t)c.dc_pred;
    for (int k = 1; k < 64;) {
      const int rs = decode_huff(hac[c.ha]);
      if (rs < 0) return fail("bad Huffman code (AC)");
      const int s = rs & 15, r = rs >> 4;
      if (s == 0) {
        if (r != 15) break;
        k += 16;
      } else {
        k += r;
        if (k > 63) return fail("AC coefficient index out of range");
        b[kZigzag[k++]] = (int16_t)receive_extend(s);
      }
    }
    return true;
  }
  bool block_prog_dc(int16_t *b, Component &c, int ah, int al) {
    if (ah == 0) {
      const int t = decode_huff(hdc[c.hd]);
      if (t < 0 || t > 15) return fail("bad Huffman code (progressive DC)");
      c.dc_pred += t ? receive_extend(t) : 0;
      b[0] = (int16_t)(c.dc_pred * (1 << al));
    } else if (get_bit()) {
      b[0] = (int16_t)(b[0] + (1 << al));
    }
    return true;
  }
  bool block_prog_ac(int16_t *b, Component &c, int ss, int se, int ah, int al) {
    const Huff &h = hac[c.ha];
    if (ah == 0) {
      if (eobrun) {
        --eobrun;
        return true;
      }
      for (int k = ss; k <= se;) {
        const int rs = decode_huff(h);
        if (rs < 0) return fail("bad Huffman code (progressive AC)");
        const int s = rs & 15, r = rs >> 4;
        if (s == 0) {
          if (r < 15) {
            eobrun = (1 << r) - 1;
            if (r) eobrun += get_bits(r);
            break;
          }
          k += 16;
        } else {
          k += r;
          if (k > 63) return fail("AC coefficient index out of range");
          b[kZigzag[k++]] = (int16_t)(receive_extend(s) * (1 << al));
        }
      }
      return true;
    }
    /* refinement (T.81 G.1.2.3) */
    const int bit = 1 << al;
    if (eobrun) {
      --eobrun;
      for (int k = ss; k <= se; ++k) {
        int16_t *q = &b[kZigzag[k]];
        if (*q != 0 && get_bit() && (*q & bit) == 0) *q = (int16_t)(*q > 0 ? *q + bit : *q - bit);
      }
      return true;
    }
    int k = ss;
    do {
      const int rs = decode_huff(h);
      if (rs < 0) return fail("bad Huffman code (AC refinement)");
      int s = rs & 15, r = rs >> 4;
      if (s == 0) {
        if (r < 15) {
          eobrun = (1 << r) - 1;
          if (r) eobrun += get_bits(r);
          r = 64; /* force the end of the block */
        }
      } else {
        if (s != 1) return fail("bad refinement magnitude");
        s = get_bit() ? bit : -bit;
      }
      while (k <= se) {
        int16_t *q = &b[kZigzag[k++]];
        if (*q != 0) {
          if (get_bit() && (*q & bit) == 0) *q = (int16_t)(*q > 0 ? *q + bit : *q - bit);
        } else {
          if (r == 0) {
            *q = (int16_t)s;
            break;
          }
          --r;
        }
      }
    } while (k <= se);
    return true;
  }

  /* ---- markers ---- */
  bool read_dqt(const uint8_t *d, size_t len) {
    while (len) {
      const int pq = d[0] >> 4, tq = d[0] & 15;
      if (tq > 3 || pq > 1) return fail("bad DQT");
      const size_t need = 1 + (pq ? 128 : 64);
      if (len < need) return fail("truncated DQT");
      for (int i = 0; i < 64; ++i) qt[tq][kZigzag[i]] = pq ? (uint16_t)((d[1 + 2 * i] << 8) | d[2 + 2 * i]) : d[1 + i];
      qt_present[tq] = true;
      d += need;
      len -= need;
    }
    return true;
  }
  bool read_dht(const uint8_t *d, size_t len) {
    while (len) {
      if (len < 17) return fail("truncated DHT");
      const int tc = d[0] >> 4, th = d[0] & 15;
      if (tc > 1 || th > 3) return fail("bad DHT");
      int n = 0;
      for (int i = 0; i < 16; ++i) n += d[1 + i];
      if (n > 256 || len < (size_t)17 + n) return fail("truncated DHT");
      if (!(tc ? hac[th] : hdc[th]).build(d + 1, d + 17, n)) return fail("bad Huffman table");
      d += 17 + n;
      len -= 17 + (size_t)n;
    }
    return true;
  }
  bool read_sof(const uint8_t *d, size_t len, bool prog) {
    if (have_sof) return fail("more than one frame header");
    if (len < 6 || d[0] != 8) return fail("only 8-bit JPEG is supported");
    H = (d[1] << 8) | d[2];
    W = (d[3] << 8) | d[4];
    ncomp = d[5];
    if (W == 0 || H == 0) return fail("empty JPEG frame");
    if (ncomp != 1 && ncomp != 3) return fail("only grey and three-component JPEG files are supported (no CMYK)");
    if (len < (size_t)6 + 3 * ncomp) return fail("truncated frame header");
    /* untrusted header: 2^28 pixels at most (a 16k x 16k sky; the coefficient arrays below cost 2 B per sample), and no
     * more pixels than the file could possibly code (a progressive all-grey image still spends > 1 bit per 8x8 block
     * and component): a few forged header bytes must not drive multi-GiB allocations */
    if ((uint64_t)W * H > ((uint64_t)1 << 28)) return fail("JPEG dimensions out of range (more than 2^28 pixels)");
    if ((uint64_t)W * H > (uint64_t)file_bytes * 4096u) return fail("JPEG dimensions are not plausible for a file of this size");
    for (int i = 0; i < ncomp; ++i) {
      Component &c = comp[i];
      c.id = d[6 + 3 * i];
      c.h = d[7 + 3 * i] >> 4;
      c.v = d[7 + 3 * i] & 15;
      c.tq = d[8 + 3 * i];
      if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) return fail("unsupported sampling factors");
      hmax = c.h > hmax ? c.h : hmax;
      vmax = c.v > vmax ? c.v : vmax;
    }
    if (ncomp == 1) comp[0].h = comp[0].v = hmax = vmax = 1; /* a single component is never interleaved */
    mcus_x = (W + 8 * hmax - 1) / (8 * hmax);
    mcus_y = (H + 8 * vmax - 1) / (8 * vmax);
    for (int i = 0; i < ncomp; ++i) {
      Component &c = comp[i];
      c.blocks_w = mcus_x * c.h;
      c.blocks_h = mcus_y * c.v;
      c.w = (W * c.h + hmax - 1) / hmax;
      c.h_px = (H * c.v + vmax - 1) / vmax;
      c.coef.alloc((size_t)c.blocks_w * c.blocks_h * 64);
    }
    /* a large image: helper threads have the kernel populate the coefficient arrays while the entropy decoder starts at their
     * front (png_io.h populate_pages: the first-touch faults of 100-200 MB leave its critical path) */
    join_helpers();
    if ((size_t)W * (size_t)H >= ((size_t)1 << 22) && !std::getenv("CURVIS_NO_POPULATE"))
      for (int i = 0; i < ncomp; ++i) {
        int16_t *ptr = comp[i].coef.p;
        const size_t bytes = comp[i].coef.n * sizeof(int16_t);
        try {
          helpers.emplace_back([ptr, bytes] { pngio::populate_pages((uint8_t *)ptr, bytes); });
        } catch (const std::exception &) { /* then the pages arrive one fault at a time */
        }
      }
    progressive = prog;
    have_sof = true;
    return true;
  }
  bool read_scan(const uint8_t *d, size_t len) {
    if (!have_sof) return fail("scan before frame header");
    if (len < 1) return fail("truncated SOS");
    const int ns = d[0];
    if (ns < 1 || ns > ncomp || len < (size_t)4 + 2 * ns) return fail("bad SOS");
    int order[3];
    for (int i = 0; i < ns; ++i) {
      int which = -1;
      for (int k = 0; k < ncomp; ++k)
        if (comp[k].id == d[1 + 2 * i]) which = k;
      if (which < 0) return fail("scan refers to an unknown component");
      comp[which].hd = d[2 + 2 * i] >> 4;
      comp[which].ha = d[2 + 2 * i] & 15;
      if (comp[which].hd > 3 || comp[which].ha > 3) return fail("bad table selector");
      order[i] = which;
    }
    const int ss = d[1 + 2 * ns], se = d[2 + 2 * ns], ah = d[3 + 2 * ns] >> 4, al = d[3 + 2 * ns] & 15;
    if (progressive) {
      if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13 || (ss == 0 && se != 0) || (ss != 0 && ns != 1)) return fail("bad progressive scan");
    } else if (ss != 0 || se != 63 || ah != 0 || al != 0) {
      return fail("bad sequential scan parameters");
    }
    for (int i = 0; i < ns; ++i) {
      const Component &c = comp[order[i]];
      if ((!progressive || ss == 0) && ah == 0 && !hdc[c.hd].present) return fail("missing DC Huffman table");
      if ((!progressive || ss != 0) && !hac[c.ha].present) return fail("missing AC Huffman table");
    }
    reset_entropy();
    int todo = restart_interval ? restart_interval : 0x7fffffff;
    auto one = [&](Component &c, int bx, int by) -> bool {
      int16_t *b = &c.coef[((size_t)by * c.blocks_w + bx) * 64];
      if (!progressive) return block_baseline(b, c);
      return ss == 0 ? block_prog_dc(b, c, ah, al) : block_prog_ac(b, c, ss, se, ah, al);
    };
    auto restart = [&]() -> bool {
      if (--todo > 0) return true;
      if (bitcnt < 24) fill();
      if (!hit_marker || marker < 0xD0 || marker > 0xD7) return true; /* no RSTn here: the next marker ends the scan */
      reset_entropy();
      todo = restart_interval;
      return true;
    };
    if (ns == 1) { /* non-interleaved: the component's own block grid, cropped to its true size */
      Component &c = comp[order[0]];
      const int bw = (c.w + 7) / 8, bh = (c.h_px + 7) / 8;
      for (int by = 0; by < bh; ++by)
        for (int bx = 0; bx < bw; ++bx) {
          if (!one(c, bx, by)) return false;
          if (!restart()) return false;
        }
    } else {
      for (int my = 0; my < mcus_y; ++my)
        for (int mx = 0; mx < mcus_x; ++mx) {
          for (int i = 0; i < ns; ++i) {
            Component &c = comp[order[i]];
            for (int v = 0; v < c.v; ++v)
              for (int h = 0; h < c.h; ++h)
                if (!one(c, mx * c.h + h, my * c.v + v)) return false;
          }
          if (!restart()) return false;
        }
    }
    /* leave p at the marker that ended the entropy-coded segment */
    if (hit_marker) {
      p -= 2;
    } else {
      while (p + 1 < end && !(p[0] == 0xFF && p[1] != 0 && !(p[1] >= 0xD0 && p[1] <= 0xD7))) ++p;
    }
    return true;
  }

  /* ---- reconstruction ---- */
  static uint8_t clamp8(long long v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
  static void idct_block(uint8_t *out, int stride, const int16_t *coef, const uint16_t *q) {
    /* 64-bit intermediates and dequantised coefficients held to +-2^15 (an 8-bit JPEG never exceeds that): a forged
     * 16-bit quantisation table with extreme coefficients cannot overflow the butterflies (signed overflow is UB) */
    typedef long long idct_t;
    { /* a block without AC coefficients -- most of a star map -- is one value: what the two passes below make of it, exactly
       * ((dc q 16384 + 65536 + (128 << 17)) >> 17 in every position), without making them */
      uint64_t ac[16];
      std::memcpy(ac, coef, sizeof ac);
#if !defined(__BYTE_ORDER__) || __BYTE_ORDER__ != __ORDER_LITTLE_ENDIAN__
#error "the DC-only test below takes coef[0] for the LOW 16 bits of the first 64-bit word: little-endian hosts only"
#endif
      uint64_t any = ac[0] & ~(uint64_t)0xFFFF; /* little endian (checked above): coef[0] is the low 16 bits */
      for (int i = 1; i < 16; ++i) any |= ac[i];
      if (!any) {
        idct_t x = (idct_t)coef[0] * (idct_t)q[0];
        x = x > 32767 ? 32767 : (x < -32768 ? -32768 : x);
        const uint8_t px = clamp8((x * 16384 + 65536 + (128 << 17)) >> 17);
        for (int i = 0; i < 8; ++i, out += stride) std::memset(out, px, 8);
        return;
      }
    }
    idct_t val[64], *v = val;
    idct_t d[64];
    for (int i = 0; i < 64; ++i) {
      const idct_t x = (idct_t)coef[i] * (idct_t)q[i];
      d[i] = x > 32767 ? 32767 : (x < -32768 ? -32768 : x);
    }
    const idct_t *D = d;
#define CV_F2F(x) ((int)((x)*4096 + 0.5))
#define CV_IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                                          \
  idct_t t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;                                  \
  p2 = s2;                                                                                    \
  p3 = s6;                                                                                    \
  p1 = (p2 + p3) * CV_F2F(0.5411961f);                                                        \
  t2 = p1 + p3 * CV_F2F(-1.847759065f);                                                       \
  t3 = p1 + p2 * CV_F2F(0.765366865f);                                                        \
  p2 = s0;                                                                                    \
  p3 = s4;                                                                                    \
  t0 = (p2 + p3) * 4096;                                                                      \
  t1 = (p2 - p3) * 4096;                                                                      \
  x0 = t0 + t3;                                                                               \
  x3 = t0 - t3;                                                                               \
  x1 = t1 + t2;                                                                               \
  x2 = t1 - t2;                                                                               \
  t0 = s7;                                                                                    \
  t1 = s5;                                                                                    \
  t2 = s3;                                                                                    \
  t3 = s1;                                                                                    \
  p3 = t0 + t2;                                                                               \
  p4 = t1 + t3;                                                                               \
  p1 = t0 + t3;                                                                               \
  p2 = t1 + t2;                                                                               \
  p5 = (p3 + p4) * CV_F2F(1.175875602f);                                                      \
  t0 = t0 * CV_F2F(0.298631336f);                                                             \
  t1 = t1 * CV_F2F(2.053119869f);                                                             \
  t2 = t2 * CV_F2F(3.072711026f);                                                             \
  t3 = t3 * CV_F2F(1.501321110f);                                                             \
  p1 = p5 + p1 * CV_F2F(-0.899976223f);                                                       \
  p2 = p5 + p2 * CV_F2F(-2.562915447f);                                                       \
  p3 = p3 * CV_F2F(-1.961570560f);                                                            \
  p4 = p4 * CV_F2F(-0.390180644f);                                                            \
  t3 += p1 + p4;                                                                              \
  t2 += p2 + p3;                                                                              \
  t1 += p2 + p4;                                                                              \
  t0 += p1 + p3;
    for (int i = 0; i < 8; ++i, ++D, ++v) { /* columns */
      if (D[8] == 0 && D[16] == 0 && D[24] == 0 && D[32] == 0 && D[40] == 0 && D[48] == 0 && D[56] == 0) {
        const idct_t dc = D[0] * 4;
        v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dc;
      } else {
        CV_IDCT_1D(D[0], D[8], D[16], D[24], D[32], D[40], D[48], D[56])
        x0 += 512;
        x1 += 512;
        x2 += 512;
        x3 += 512;
        v[0] = (x0 + t3) >> 10;
        v[56] = (x0 - t3) >> 10;
        v[8] = (x1 + t2) >> 10;
        v[48] = (x1 - t2) >> 10;
        v[16] = (x2 + t1) >> 10;
        v[40] = (x2 - t1) >> 10;
        v[24] = (x3 + t0) >> 10;
        v[32] = (x3 - t0) >> 10;
      }
    }
    v = val;
    for (int i = 0; i < 8; ++i, v += 8, out += stride) { /* rows */
      CV_IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
      x0 += 65536 + (128 << 17);
      x1 += 65536 + (128 << 17);
      x2 += 65536 + (128 << 17);
      x3 += 65536 + (128 << 17);
      out[0] = clamp8((x0 + t3) >> 17);
      out[7] = clamp8((x0 - t3) >> 17);
      out[1] = clamp8((x1 + t2) >> 17);
      out[6] = clamp8((x1 - t2) >> 17);
      out[2] = clamp8((x2 + t1) >> 17);
      out[5] = clamp8((x2 - t1) >> 17);
      out[3] = clamp8((x3 + t0) >> 17);
      out[4] = clamp8((x3 - t0) >> 17);
    }
#undef CV_IDCT_1D
#undef CV_F2F
  }
  /* [0, n) in contiguous pieces over up to `threads` threads (the caller's included); fn(begin, end) */
  template <class F>
  static void parallel_ranges(int n, int threads, F fn) {
    threads = std::max(1, std::min(threads, n));
    if (threads == 1) {
      fn(0, n);
      return;
    }
    std::vector<std::thread> th;
    const int per = (n + threads - 1) / threads;
    bool oom = false; /* an exception must not leave a helper thread (std::terminate): it is passed on by the caller's */
    auto guarded = [&fn, &oom](int b, int e) {
      try {
        fn(b, e);
      } catch (const std::bad_alloc &) {
        oom = true;
      }
    };
    int next = per; /* the caller's own piece is [0, per); pieces no thread could be made for are done here as well */
    try {
      th.reserve((size_t)threads);
      for (; next < n; next += per) th.emplace_back(guarded, next, std::min(n, next + per));
    } catch (const std::exception &) { /* no more threads to be had (or no memory for one) */
    }
    guarded(0, std::min(n, per));
    for (; next < n; next += per) guarded(next, std::min(n, next + per));
    for (auto &x : th) x.join();
    if (oom) throw std::bad_alloc();
  }
  /* Reconstruction is independent per block and per pixel row (the entropy decoding before it is not): block rows of the
   * inverse DCT and pixel rows of upsampling + colour conversion are shared out over a few threads -- a large image only
   * (an 8192x4096 star map: 0.7-1.5 s of one thread, about half of it here) */
  int reconstruct_threads() const {
    if ((size_t)W * (size_t)H < ((size_t)1 << 20)) return 1;
    if (const char *e = std::getenv("CURVIS_DECODE_THREADS")) return std::max(1, std::min(64, std::atoi(e)));
    return (int)std::max(1u, std::min(4u, std::thread::hardware_concurrency() / 2u));
  }
  std::vector<std::thread> helpers; /* page-populating helpers of the coefficient arrays (read_sof) */
  void join_helpers() {
    for (auto &t : helpers)
      if (t.joinable()) t.join();
    helpers.clear();
  }
  ~Decoder() { join_helpers(); }
  bool reconstruct(pngio::Image &img) {
    join_helpers();
    const int threads = reconstruct_threads();
    const bool timing = std::getenv("CURVIS_DEBUG_TIMING") != nullptr;
    double t_last = timing ? pngio::now_s() : 0.0;
    auto mark = [&](const char *what) {
      if (!timing) return;
      const double t = pngio::now_s();
      std::fprintf(stderr, "[curvis timing]     jpeg decode: %-22s %7.1f ms\n", what, (t - t_last) * 1e3);
      t_last = t;
    };
    for (int i = 0; i < ncomp; ++i) {
      Component &c = comp[i];
      if (!qt_present[c.tq]) return fail("missing quantisation table");
      const int stride = c.blocks_w * 8;
      c.plane.assign((size_t)stride * c.blocks_h * 8, 0);
      const uint16_t *q = qt[c.tq];
      parallel_ranges(c.blocks_h, threads, [&c, stride, q](int b0, int b1) {
        for (int by = b0; by < b1; ++by)
          for (int bx = 0; bx < c.blocks_w; ++bx)
            idct_block(&c.plane[((size_t)by * 8) * stride + (size_t)bx * 8], stride, &c.coef[((size_t)by * c.blocks_w + bx) * 64], q);
      });
      c.coef.release();
    }
    mark("inverse DCT");
    img.w = (uint32_t)W;
    img.h = (uint32_t)H;
    { /* the RGBA image: its pages populated by the threads that will write them, then sized (convert_rows writes all four bytes) */
      const size_t row_bytes = (size_t)W * 4;
      img.rgba.clear();
      img.rgba.reserve(row_bytes * (size_t)H);
      if (threads > 1 && !std::getenv("CURVIS_NO_POPULATE")) {
        uint8_t *base = img.rgba.data();
        parallel_ranges(H, threads, [base, row_bytes](int y0, int y1) { pngio::populate_pages(base + (size_t)y0 * row_bytes, (size_t)(y1 - y0) * row_bytes); });
      }
      img.rgba.resize(row_bytes * (size_t)H);
    }
    const bool rgb_direct = ncomp == 3 && (adobe_transform == 0 || (adobe_transform < 0 && !jfif && comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B'));
    parallel_ranges(H, threads, [this, &img, rgb_direct](int y0, int y1) { convert_rows(img, rgb_direct, y0, y1); });
    mark("upsampling + colour");
    return true;
  }
  /* pixel rows [y0, y1): full-resolution rows of each component (triangle-filter upsampling, edge samples replicated), then
   * the colour transform; reads the component planes, writes its own rows of img.rgba */
  void convert_rows(pngio::Image &img, bool rgb_direct, int y0, int y1) const {
    std::vector<uint8_t> rows[3];
    for (int i = 0; i < ncomp; ++i) rows[i].resize((size_t)W + 16);
    for (int y = y0; y < y1; ++y) {
      for (int i = 0; i < ncomp; ++i) {
        const Component &c = comp[i];
        const int stride = c.blocks_w * 8;
        const int hs = hmax / c.h, vs = vmax / c.v;
        uint8_t *out = rows[i].data();
        if (hs == 1 && vs == 1) {
          std::memcpy(out, &c.plane[(size_t)y * stride], (size_t)W);
          continue;
        }
        /* vertical neighbours of the chroma row nearest to luma row y */
        const int cy = vs == 2 ? y >> 1 : y;
        int cfar = vs == 2 ? ((y & 1) ? cy + 1 : cy - 1) : cy;
        if (cfar < 0) cfar = 0;
        if (cfar > c.h_px - 1) cfar = c.h_px - 1;
        const uint8_t *near = &c.plane[(size_t)cy * stride], *far = &c.plane[(size_t)cfar * stride];
        const int cw = c.w;
        if (hs == 1) { /* v2 only: (3 near + far + 2) >> 2 */
          for (int x = 0; x < W; ++x) out[x] = (uint8_t)((3 * near[x] + far[x] + 2) >> 2);
        } else if (vs == 1) { /* h2 only */
          if (cw == 1) {
            out[0] = out[1] = near[0];
          } else {
            out[0] = near[0];
            out[1] = (uint8_t)((near[0] * 3 + near[1] + 2) >> 2);
            int x;
            for (x = 1; x < cw - 1; ++x) {
              const int n = 3 * near[x] + 2;
              out[2 * x] = (uint8_t)((n + near[x - 1]) >> 2);
              out[2 * x + 1] = (uint8_t)((n + near[x + 1]) >> 2);
            }
            out[2 * x] = (uint8_t)((near[cw - 2] * 3 + near[cw - 1] + 2) >> 2);
            out[2 * x + 1] = near[cw - 1];
          }
        } else { /* h2 v2 */
          if (cw == 1) {
            out[0] = out[1] = (uint8_t)((3 * near[0] + far[0] + 2) >> 2);
          } else {
            int t0 = 3 * near[0] + far[0], t1;
            out[0] = (uint8_t)((t0 + 2) >> 2);
            for (int x = 1; x < cw; ++x) {
              t1 = t0;
              t0 = 3 * near[x] + far[x];
              out[2 * x - 1] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
              out[2 * x] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
            }
            out[2 * cw - 1] = (uint8_t)((t0 + 2) >> 2);
          }
        }
      }
      uint8_t *o = &img.rgba[(size_t)y * W * 4];
      if (ncomp == 1) {
        for (int x = 0; x < W; ++x) {
          o[4 * x] = o[4 * x + 1] = o[4 * x + 2] = rows[0][x];
          o[4 * x + 3] = 255;
        }
      } else if (rgb_direct) {
        for (int x = 0; x < W; ++x) {
          o[4 * x] = rows[0][x];
          o[4 * x + 1] = rows[1][x];
          o[4 * x + 2] = rows[2][x];
          o[4 * x + 3] = 255;
        }
      } else {
        for (int x = 0; x < W; ++x) {
          const int yy = rows[0][x], cb = (int)rows[1][x] - 128, cr = (int)rows[2][x] - 128;
          o[4 * x] = clamp8(yy + ((45 * cr) >> 5));
          o[4 * x + 1] = clamp8(yy - ((11 * cb + 23 * cr) >> 5));
          o[4 * x + 2] = clamp8(yy + ((113 * cb) >> 6));
          o[4 * x + 3] = 255;
        }
      }
    }
  }

  bool run(pngio::Image &img) {
    const double t_run0 = pngio::now_s();
    if (end - p < 4 || p[0] != 0xFF || p[1] != 0xD8) return fail("not a JPEG file");
    p += 2;
    bool seen_scan = false;
    for (;;) {
      while (p < end && *p != 0xFF) ++p; /* tolerate garbage between segments */
      while (p < end && *p == 0xFF) ++p;
      if (p >= end) break;
      const uint8_t m = *p++;
      if (m == 0xD9) break;                         /* EOI */
      if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue; /* TEM, stray RSTn */
      if (end - p < 2) return fail("truncated JPEG segment");
      const size_t len = ((size_t)p[0] << 8) | p[1];
      if (len < 2 || (size_t)(end - p) < len) return fail("truncated JPEG segment");
      const uint8_t *d = p + 2;
      const size_t n = len - 2;
      p += len;
      switch (m) {
        case 0xDB: if (!read_dqt(d, n)) return false; break;
        case 0xC4: if (!read_dht(d, n)) return false; break;
        case 0xC0: case 0xC1: if (!read_sof(d, n, false)) return false; break;
        case 0xC2: if (!read_sof(d, n, true)) return false; break;
        case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
          return fail("unsupported JPEG process (lossless, hierarchical or arithmetic coding)");
        case 0xDD:
          if (n < 2) return fail("bad DRI");
          restart_interval = (d[0] << 8) | d[1];
          break;
        case 0xE0: if (n >= 5 && !std::memcmp(d, "JFIF", 5)) jfif = true; break;
        case 0xEE: if (n >= 12 && !std::memcmp(d, "Adobe", 5)) adobe_transform = d[11]; break;
        case 0xDA:
          if (!read_scan(d, n)) return false;
          seen_scan = true;
          break;
        default: break; /* APPn, COM, DNL ...: skipped */
      }
    }
    if (!have_sof || !seen_scan) return fail("JPEG file without image data");
    if (std::getenv("CURVIS_DEBUG_TIMING"))
      std::fprintf(stderr, "[curvis timing]     jpeg decode: %-22s %7.1f ms\n", "segments + entropy", (pngio::now_s() - t_run0) * 1e3);
    return reconstruct(img);
  }
};

inline bool decode(const std::vector<uint8_t> &file, pngio::Image &img, std::string &err) {
  Decoder *d = new Decoder();
  d->p = file.data();
  d->end = file.data() + file.size();
  d->file_bytes = file.size();
  d->err = &err;
  std::memset(d->qt, 0, sizeof d->qt);
  bool ok = false;
  try {
    ok = d->run(img);
  } catch (const std::bad_alloc &) { /* extern "C" callers and decoder threads must see an error, not std::terminate */
    err = "out of memory while decoding the JPEG file";
  } catch (const std::system_error &e) { /* a helper thread could not be started */
    err = std::string("could not decode the JPEG file: ") + e.what();
  }
  delete d;
  if (!ok && err.empty()) err = "corrupt JPEG";
  return ok;
}

/* PNG or JPEG by signature: image::open guesses the format from the content too (image 0.25.2 ImageReader) */
inline bool load_image(const std::string &path, pngio::Image &img, std::string &err) {
  std::vector<uint8_t> file;
  if (!pngio::read_file(path, file)) {
    err = "could not read " + path;
    return false;
  }
  if (file.size() >= 2 && file[0] == 0xFF && file[1] == 0xD8) return jpegio::decode(file, img, err);
  try {
    return pngio::decode(file, img, err);
  } catch (const std::bad_alloc &) {
    err = "out of memory while decoding the PNG file";
    return false;
  } catch (const std::system_error &e) { /* a helper thread (follower, CRC) could not be started */
    err = std::string("could not decode the PNG file: ") + e.what();
    return false;
  }
}

}  // namespace jpegio
#endif
