/* png_io.h -- minimal PNG reader/writer over zlib for the curvis host binary.
 *
 * Reader: 8/16-bit grey, grey+alpha, RGB, RGBA, palette (with tRNS), non-interlaced and Adam7,
 * expanded to RGBA8 the way the reference obtains texels (image 0.25.2: DynamicImage::get_pixel ->
 * Rgba<u8>; 16-bit samples are reduced with (v + 128) / 257, grey is replicated, missing alpha = 255)
 * -- src/images.rs:8, :107-111.
 * Writer: RGB8, one IDAT (src/rendering.rs:110, :311 save an ImageRgb8 as PNG; the compressed bytes differ from
 * the image crate's encoder, the decoded pixels are identical): zlib at a chosen level with filter type 0, or the
 * fast PNG-specific deflate below (the default of the curvis binary).
 */
#ifndef CURVIS_PNG_IO_H
#define CURVIS_PNG_IO_H
#include <zlib.h>
#include "inflate_fast.h"
#include "../png_codes.h"

#include <time.h>
#if defined(__linux__)
#include <sys/mman.h>
#endif
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include <memory>
#include <new>
#include <condition_variable>
#include <mutex>
#include <thread>

namespace pngio {

struct Image {
  uint32_t w = 0, h = 0;
  std::vector<uint8_t> rgba; /* h*w*4 */
};

inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline void put_be32(std::vector<uint8_t> &v, uint32_t x) {
  v.push_back((uint8_t)(x >> 24));
  v.push_back((uint8_t)(x >> 16));
  v.push_back((uint8_t)(x >> 8));
  v.push_back((uint8_t)x);
}

inline bool read_file(const std::string &path, std::vector<uint8_t> &out) {
  FILE *f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  std::fseek(f, 0, SEEK_END);
  long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  out.resize(n > 0 ? (size_t)n : 0);
  size_t got = n > 0 ? std::fread(out.data(), 1, (size_t)n, f) : 0;
  std::fclose(f);
  return got == out.size();
}

inline uint8_t paeth(uint8_t a, uint8_t b, uint8_t c) {
  int p = (int)a + (int)b - (int)c;
  int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
  if (pa <= pb && pa <= pc) return a;
  if (pb <= pc) return b;
  return c;
}

/* un-filter `rows` scanlines of `stride` bytes (bpp = bytes per complete pixel, >= 1) in place;
 * src points at filter byte of the first line; returns bytes consumed or 0 on error */
inline size_t unfilter(uint8_t *src, size_t avail, size_t rows, size_t stride, size_t bpp, std::vector<uint8_t> &out) {
  out.assign(rows * stride, 0);
  if (avail < rows * (stride + 1)) return 0;
  for (size_t y = 0; y < rows; ++y) {
    const uint8_t ft = src[y * (stride + 1)];
    const uint8_t *in = src + y * (stride + 1) + 1;
    uint8_t *cur = out.data() + y * stride;
    const uint8_t *up = y ? out.data() + (y - 1) * stride : nullptr;
    for (size_t x = 0; x < stride; ++x) {
      const uint8_t a = x >= bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
      uint8_t v = in[x];
      switch (ft) {
        case 0: break;
        case 1: v = (uint8_t)(v + a); break;
        case 2: v = (uint8_t)(v + b); break;
        case 3: v = (uint8_t)(v + (uint8_t)(((int)a + (int)b) >> 1)); break;
        case 4: v = (uint8_t)(v + paeth(a, b, c)); break;
        default: return 0;
      }
      cur[x] = v;
    }
  }
  return rows * (stride + 1);
}

/* First-touch page faults are a fifth of a large decode (128 MiB = 32 768 faults per buffer, on the critical path of whichever
 * stage writes first).  A helper thread asks the kernel to populate a buffer in one go (MADV_POPULATE_WRITE, Linux 5.14: ordinary
 * 4 KiB pages, no compaction) while the stages start at its front; where the call is unknown nothing changes. */
inline void populate_pages(uint8_t *p, size_t bytes) {
#if defined(__linux__) && defined(MADV_POPULATE_WRITE)
  const uintptr_t lo = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, hi = ((uintptr_t)p + bytes) & ~(uintptr_t)4095;
  for (uintptr_t a = lo; a < hi; a += (uintptr_t)8 << 20) /* 8 MiB at a time: the front is ready early */
    if (::madvise((void *)a, (size_t)std::min<uintptr_t>(hi - a, (uintptr_t)8 << 20), MADV_POPULATE_WRITE) != 0) return;
#else
  (void)p;
  (void)bytes;
#endif
}

inline uint8_t reduce16(uint32_t v) { return (uint8_t)((v + 128u) / 257u); } /* image crate u16 -> u8 */

/* The same reconstruction IN PLACE and with one loop per filter type, BPP a compile-time constant (3 or 4: 8-bit RGB / RGBA,
 * what a sky texture is): the generic routine above decides per byte what the row's filter is and zero-fills a second copy of
 * the image first -- 0.35 s of the 0.8 s an 8192x4096 background took to load.  `row` points at the first data byte of the
 * scanline (after its filter byte), `up` at the previous RECONSTRUCTED scanline or nullptr. */
template <size_t BPP>
inline bool unfilter_row_inplace(uint8_t ft, uint8_t *row, const uint8_t *up, size_t stride) {
  switch (ft) {
    case 0:
      return true;
    case 1: /* Sub */
      for (size_t x = BPP; x < stride; ++x) row[x] = (uint8_t)(row[x] + row[x - BPP]);
      return true;
    case 2: /* Up */
      if (up)
        for (size_t x = 0; x < stride; ++x) row[x] = (uint8_t)(row[x] + up[x]);
      return true;
    case 3: /* Average */
      if (!up) {
        for (size_t x = BPP; x < stride; ++x) row[x] = (uint8_t)(row[x] + (row[x - BPP] >> 1));
        return true;
      }
      for (size_t x = 0; x < BPP; ++x) row[x] = (uint8_t)(row[x] + (up[x] >> 1));
      for (size_t x = BPP; x < stride; ++x) row[x] = (uint8_t)(row[x] + (uint8_t)(((unsigned)row[x - BPP] + (unsigned)up[x]) >> 1));
      return true;
    case 4: { /* Paeth: the predictor of the first pixel is `up` (a = c = 0), without a previous line it is `a` (b = c = 0) */
      if (!up) {
        for (size_t x = BPP; x < stride; ++x) row[x] = (uint8_t)(row[x] + row[x - BPP]);
        return true;
      }
#if defined(__SSE2__)
      { /* one pixel per step, its channels side by side in 16-bit lanes; no branches (the scalar loop below mispredicts on noisy
         * rows).  Loads take 4 bytes also for BPP 3: the byte behind a row is the next row's filter byte / the buffer's slack. */
        const __m128i zero = _mm_setzero_si128(), low = _mm_set1_epi16(0x00FF);
        __m128i a = zero, c = zero;
        for (size_t x = 0; x < stride; x += BPP) {
          int32_t wb, wd;
          std::memcpy(&wb, up + x, 4);
          std::memcpy(&wd, row + x, 4);
          const __m128i b = _mm_unpacklo_epi8(_mm_cvtsi32_si128(wb), zero);
          __m128i d = _mm_unpacklo_epi8(_mm_cvtsi32_si128(wd), zero);
          __m128i pa = _mm_sub_epi16(b, c), pb = _mm_sub_epi16(a, c); /* p - a, p - b with p = a + b - c */
          __m128i pc = _mm_add_epi16(pa, pb);
          pa = _mm_max_epi16(pa, _mm_sub_epi16(zero, pa));
          pb = _mm_max_epi16(pb, _mm_sub_epi16(zero, pb));
          pc = _mm_max_epi16(pc, _mm_sub_epi16(zero, pc));
          const __m128i smallest = _mm_min_epi16(pc, _mm_min_epi16(pa, pb));
          const __m128i is_a = _mm_cmpeq_epi16(smallest, pa), is_b = _mm_cmpeq_epi16(smallest, pb); /* ties: a, then b, then c */
          const __m128i bc = _mm_or_si128(_mm_and_si128(is_b, b), _mm_andnot_si128(is_b, c));
          const __m128i pred = _mm_or_si128(_mm_and_si128(is_a, a), _mm_andnot_si128(is_a, bc));
          d = _mm_and_si128(_mm_add_epi16(d, pred), low);
          const int32_t out = _mm_cvtsi128_si32(_mm_packus_epi16(d, d));
          std::memcpy(row + x, &out, BPP); /* BPP bytes only: what follows is still filtered input */
          a = d;
          c = b;
        }
        return true;
      }
#endif
      int a[BPP], c[BPP];
      for (size_t k = 0; k < BPP; ++k) {
        row[k] = (uint8_t)(row[k] + up[k]);
        a[k] = row[k];
        c[k] = up[k];
      }
      for (size_t x = BPP; x < stride; x += BPP) /* stride is a multiple of BPP */
        for (size_t k = 0; k < BPP; ++k) {
          const int b = up[x + k];
          const int pa0 = b - c[k], pb0 = a[k] - c[k]; /* p - a, p - b with p = a + b - c */
          const int pa = pa0 < 0 ? -pa0 : pa0, pb = pb0 < 0 ? -pb0 : pb0, pc0 = pa0 + pb0, pc = pc0 < 0 ? -pc0 : pc0;
          const int pred = (pa <= pb && pa <= pc) ? a[k] : (pb <= pc ? b : c[k]);
          const int v = (row[x + k] + pred) & 0xFF;
          row[x + k] = (uint8_t)v;
          a[k] = v;
          c[k] = b;
        }
      return true;
    }
    default:
      return false;
  }
}

inline double now_s();
inline bool decode(const std::vector<uint8_t> &file, Image &img, std::string &err) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  const bool timing = std::getenv("CURVIS_DEBUG_TIMING") != nullptr;
  double t_last = timing ? now_s() : 0.0;
  auto mark = [&](const char *what) {
    if (!timing) return;
    const double t = now_s();
    std::fprintf(stderr, "[curvis timing]     png decode: %-22s %7.1f ms\n", what, (t - t_last) * 1e3);
    t_last = t;
  };
  if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) {
    err = "not a PNG file (only PNG backgrounds are supported by this build; convert JPEG skies to PNG)";
    return false;
  }
  size_t pos = 8;
  uint32_t W = 0, H = 0;
  int depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> plte, trns;
  /* the IDAT chunks are first only NOTED where they are in the file: (data, length, CRC field); before inflating they are copied
   * into ONE contiguous, padded buffer (inflate_fast.h reads a few bytes ahead), and their CRC-32s are checked by a helper
   * thread meanwhile (a 36 MB star map: 95 ms of checksumming before the first byte was inflated when it ran first) */
  struct Span { const uint8_t *data; uint32_t len; const uint8_t *type; };
  std::vector<Span> idat;
  bool have_ihdr = false;
  while (pos + 12 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    const char *type = (const char *)&file[pos + 4];
    if (pos + 12 + (size_t)len > file.size()) {
      err = "truncated PNG chunk";
      return false;
    }
    const uint8_t *data = &file[pos + 8];
    /* every chunk carries a CRC-32 of type + data.  Like the reference's png crate, a mismatch is fatal for CRITICAL
     * chunks only (upper-case first letter: IHDR, PLTE, IDAT, IEND); an ancillary chunk with a damaged CRC (tEXt, iCCP,
     * ... and tRNS) is skipped, so a star map with a broken metadata chunk loads here as it does there */
    if (!std::memcmp(type, "IDAT", 4)) {
      idat.push_back(Span{data, len, &file[pos + 4]});
      pos += 12 + (size_t)len;
      continue;
    }
    if ((uint32_t)crc32(0L, &file[pos + 4], (uInt)(4 + len)) != be32(&file[pos + 8 + len])) {
      if (!(type[0] & 0x20)) {
        err = "PNG chunk CRC mismatch";
        return false;
      }
      pos += 12 + (size_t)len;
      continue;
    }
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len != 13) { err = "bad IHDR"; return false; }
      W = be32(data); H = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
      have_ihdr = true;
    } else if (!std::memcmp(type, "PLTE", 4)) {
      plte.assign(data, data + len);
    } else if (!std::memcmp(type, "tRNS", 4)) {
      trns.assign(data, data + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      break;
    }
    pos += 12 + (size_t)len;
  }
  mark("chunks");
  bool idat_crc_bad = false;
  std::thread crc_thread([&] {
    for (const Span &sp : idat)
      if ((uint32_t)crc32(0L, sp.type, (uInt)(4 + sp.len)) != be32(sp.data + sp.len)) idat_crc_bad = true;
  });
  struct Joiner { /* whichever way decode() is left, the helper has finished with `file` first */
    std::thread &t;
    ~Joiner() { if (t.joinable()) t.join(); }
  } crc_joiner{crc_thread};
  auto idat_crc_ok = [&] { /* a damaged critical chunk is THE error, whatever the inflater made of its bytes */
    if (crc_thread.joinable()) crc_thread.join();
    if (idat_crc_bad) err = "PNG chunk CRC mismatch";
    return !idat_crc_bad;
  };
  if (!have_ihdr || W == 0 || H == 0) { err = "missing IHDR"; return false; }
  int channels;
  switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: err = "unsupported PNG colour type"; return false;
  }
  /* allowed bit depths per colour type (PNG specification, table 11.1): grey 1/2/4/8/16, palette 1/2/4/8,
   * everything else 8/16.  Anything else (0, 3, 5, 6, 7, ...) is a forged header: depth 0 would divide by zero. */
  const bool small_depth = depth == 1 || depth == 2 || depth == 4;
  const bool depth_ok = ctype == 0 ? (small_depth || depth == 8 || depth == 16)
                        : ctype == 3 ? (small_depth || depth == 8)
                                     : (depth == 8 || depth == 16);
  if (!depth_ok) { err = "unsupported PNG bit depth"; return false; }
  if (interlace != 0 && interlace != 1) { err = "unsupported PNG interlace method"; return false; }
  /* untrusted dimensions go straight into allocations: bound them (65536 x 32768 is four times the largest sky
   * that makes sense here; RGBA8 of that is 8 GiB) */
  if (W > 65536u || H > 65536u || (uint64_t)W * H > ((uint64_t)1 << 31)) { err = "PNG dimensions out of range"; return false; }
  const size_t bits_pp = (size_t)channels * depth;
  const size_t bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
  /* fast path: 8-bit RGB / RGBA, not interlaced, no transparent colour -- every sky texture in practice.  Scanlines are
   * reconstructed IN PLACE in the inflated buffer and go to the RGBA image a row at a time (cache-hot), by a second thread
   * that follows the inflater.  What makes rewriting earlier rows safe is inflate_fast.h's Progress contract: the output
   * buffer IS the window, the inflater never reads or writes below op - 32768, and Progress publishes settled = have - 32 KiB
   * to the follower, which touches nothing above that bound.
   * An 8192x4096 background: 0.80 s -> 0.23 s (smooth), 1.3 s -> 0.7 s (noisy, inflate-bound). */
  const bool fast = !interlace && depth == 8 && (ctype == 6 || (ctype == 2 && trns.size() < 6));
  const size_t fstride = (size_t)W * (size_t)channels;
  std::mutex fmu;
  std::condition_variable fcv;
  size_t f_have = 0;      /* inflated bytes the follower may touch (guarded by fmu) */
  bool f_done = false;    /* the inflater has stopped (end of stream or error) */
  bool f_bad_filter = false, f_oom = false, f_no_thread = false;
  /* declared BEFORE the follower and its guard: locals die in reverse order, and the follower / converter threads read and write
   * raw.data() until the guard has joined them */
  struct RawBuffer { /* the inflated scanlines: NOT zero-filled (a vector's resize would write 134 MB that the inflater overwrites) */
    std::unique_ptr<uint8_t[]> p;
    size_t n = 0;
    uint8_t *data() const { return p.get(); }
    size_t size() const { return n; }
    void resize(size_t bytes) { /* grows once, from empty; afterwards only shrinks */
      if (!p) p.reset(new uint8_t[bytes]);
      n = bytes;
    }
  } raw;
  uint32_t follower_adler = 1; /* Adler-32 of the scanlines the follower has taken (it sums a row before it rewrites it) */
  std::thread follower;
  struct FollowerGuard { /* however decode() is left -- an exception included -- the follower is told to stop and joined first */
    std::thread &t;
    std::mutex &mu;
    std::condition_variable &cv;
    bool &done;
    ~FollowerGuard() {
      if (!t.joinable()) return;
      {
        std::lock_guard<std::mutex> g(mu);
        done = true;
      }
      cv.notify_all();
      t.join();
    }
  } follower_guard{follower, fmu, fcv, f_done};
  img.w = W;
  img.h = H;
  /* inflate (inflate_fast.h: the stream contiguous in memory, the output buffer its own window) */
  {
    size_t zbytes = 0;
    for (const Span &sp : idat) zbytes += sp.len;
    std::unique_ptr<uint8_t[]> zin(new uint8_t[zbytes + cvinflate::kInputPadding]);
    {
      size_t o = 0;
      for (const Span &sp : idat) {
        std::memcpy(zin.get() + o, sp.data, sp.len);
        o += sp.len;
      }
      std::memset(zin.get() + zbytes, 0, cvinflate::kInputPadding);
    }
    /* the inflated stream of a valid file is exactly the filtered scanlines; an interlaced image adds at most one
     * filter byte per pass row (< 2 H rows in total) plus rounding: the buffer never grows beyond that, so a
     * decompression bomb fails instead of exhausting memory */
    const size_t cap = (((size_t)W * bits_pp + 7) / 8 + 1) * ((size_t)H + 8) * (interlace ? 2 : 1) + 64;
    raw.resize(cap);
    std::thread populate_raw;
    struct JoinPopulate {
      std::thread &t;
      ~JoinPopulate() { if (t.joinable()) t.join(); }
    } join_populate{populate_raw};
    if (fast && cap >= ((size_t)16 << 20) && !std::getenv("CURVIS_NO_POPULATE")) {
      try {
        populate_raw = std::thread([&raw, cap] { populate_pages(raw.data(), cap); });
      } catch (const std::exception &) { /* without it the pages arrive one fault at a time */
      }
    }
    if (fast)
      follower = std::thread([&] {
        /* a third stage behind this one: reconstructed rows -> the RGBA image (its 128 MiB of first-touch page faults included),
         * so that checksum + reconstruction, the serial part, has this thread to itself */
        std::mutex cmu;
        std::condition_variable ccv;
        size_t rows_ready = 0;
        bool rows_done = false, c_oom = false;
        std::thread converter;
        try {
          converter = std::thread([&] {
            try {
              img.rgba.reserve((size_t)W * H * 4);
              if (!std::getenv("CURVIS_NO_POPULATE")) populate_pages(img.rgba.data(), (size_t)W * H * 4); /* this stage has nothing to do yet */
              img.rgba.resize((size_t)W * H * 4);
            } catch (const std::bad_alloc &) {
              c_oom = true;
              return;
            }
            size_t y = 0;
            for (;;) {
              size_t avail;
              {
                std::unique_lock<std::mutex> g(cmu);
                ccv.wait(g, [&] { return rows_ready > y || rows_done; });
                avail = rows_ready;
                if (avail <= y) return;
              }
              for (; y < avail; ++y) {
                const uint8_t *src = raw.data() + y * (fstride + 1) + 1;
                uint8_t *o = &img.rgba[y * (size_t)W * 4];
                if (channels == 4) {
                  std::memcpy(o, src, fstride);
                } else {
                  for (size_t x = 0; x < W; ++x) {
                    o[4 * x + 0] = src[3 * x + 0];
                    o[4 * x + 1] = src[3 * x + 1];
                    o[4 * x + 2] = src[3 * x + 2];
                    o[4 * x + 3] = 255;
                  }
                }
              }
            }
          });
        } catch (const std::exception &) { /* no thread to be had: this stage cannot run -- said as such, not as "out of memory" */
          f_no_thread = true;
          return;
        }
        auto hand_over = [&](size_t rows, bool last) {
          {
            std::lock_guard<std::mutex> g(cmu);
            rows_ready = rows;
            rows_done = rows_done || last;
          }
          ccv.notify_one();
        };
        const uint8_t *up = nullptr;
        size_t seen = 0, y = 0;
        uLong adler = 1;
        for (; y < H; ++y) {
          const size_t need = (y + 1) * (fstride + 1);
          if (seen < need) {
            hand_over(y, false); /* about to wait: what is reconstructed can go on meanwhile */
            std::unique_lock<std::mutex> g(fmu);
            fcv.wait(g, [&] { return f_have >= need || f_done; });
            seen = f_have;
            if (seen < need) break; /* the stream ended short: the inflater's caller reports it */
          }
          uint8_t *line = raw.data() + y * (fstride + 1);
          adler = adler32(adler, line, (uInt)(fstride + 1)); /* before the row is rewritten */
          const bool ok = channels == 4 ? unfilter_row_inplace<4>(line[0], line + 1, up, fstride) : unfilter_row_inplace<3>(line[0], line + 1, up, fstride);
          if (!ok) {
            f_bad_filter = true;
            break;
          }
          up = line + 1;
          if ((y & 31) == 31) hand_over(y + 1, false);
        }
        hand_over(y, true);
        converter.join();
        if (c_oom) f_oom = true;
        follower_adler = (uint32_t)adler;
      });
    auto finish_follower = [&] {
      if (!follower.joinable()) return;
      {
        std::lock_guard<std::mutex> g(fmu);
        f_done = true;
      }
      fcv.notify_all();
      follower.join();
    };
    struct Publish { /* inflate_fast's progress: bytes the decoder no longer reads back may be rewritten by the follower */
      std::mutex &mu;
      std::condition_variable &cv;
      size_t &have;
      static void call(void *u, size_t settled) {
        Publish *p = (Publish *)u;
        {
          std::lock_guard<std::mutex> g(p->mu);
          p->have = settled;
        }
        p->cv.notify_one();
      }
    } publish{fmu, fcv, f_have};
    cvinflate::Progress prog;
    if (fast) {
      prog.fn = &Publish::call;
      prog.user = &publish;
    }
    size_t have = 0;
    uint32_t adler_stored = 0;
    const int rc = cvinflate::inflate_zlib(zin.get(), zbytes, raw.data(), raw.size(), &have, &adler_stored, prog);
    finish_follower();
    if (!idat_crc_ok()) return false;
    if (rc == cvinflate::E_OUTPUT_FULL) { err = "PNG data stream larger than its header allows"; return false; }
    if (rc != cvinflate::OK) { err = "corrupt PNG data stream"; return false; }
    raw.resize(have);
    /* Adler-32 of the inflated data (RFC 1950), as zlib checks it: the follower has summed the rows it took before it rewrote
     * them; whatever it did not take is still as inflated */
    if (!f_bad_filter && !f_oom && !f_no_thread) {
      uLong adler = 1;
      size_t from = 0;
      if (fast) {
        const size_t rows_taken = std::min((size_t)H, have / (fstride + 1));
        adler = follower_adler;
        from = rows_taken * (fstride + 1);
      }
      for (size_t o = from; o < have;) { /* zlib's adler32 takes a 32-bit length */
        const size_t n = std::min<size_t>(have - o, (size_t)1 << 30);
        adler = adler32(adler, raw.data() + o, (uInt)n);
        o += n;
      }
      if ((uint32_t)adler != adler_stored) { err = "corrupt PNG data stream"; return false; }
    }
  }
  mark(fast ? "inflate (+ unfilter, RGBA)" : "inflate");
  if (fast) {
    if (f_oom) throw std::bad_alloc(); /* in THIS thread, where the callers expect it */
    if (f_no_thread) { err = "could not start a PNG decoder thread (std::system_error: resource unavailable)"; return false; }
    if (f_bad_filter || raw.size() < (size_t)H * (fstride + 1)) { err = "corrupt PNG scanlines"; return false; }
    return true;
  }
  img.rgba.assign((size_t)W * H * 4, 255);
  auto sample = [&](const uint8_t *line, size_t x, int ch) -> uint32_t { /* raw sample value */
    if (depth == 8) return line[x * channels + ch];
    if (depth == 16) return ((uint32_t)line[(x * channels + ch) * 2] << 8) | line[(x * channels + ch) * 2 + 1];
    const size_t bit = x * depth;
    return (line[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1);
  };
  auto put = [&](size_t px, size_t py, const uint8_t *line, size_t x) {
    uint8_t *o = &img.rgba[(py * W + px) * 4];
    auto to8 = [&](uint32_t v) -> uint8_t {
      if (depth == 16) return reduce16(v);
      if (depth == 8) return (uint8_t)v;
      return (uint8_t)(v * 255u / ((1u << depth) - 1)); /* png crate EXPAND of 1/2/4-bit grey */
    };
    switch (ctype) {
      case 0: {
        const uint32_t v = sample(line, x, 0);
        o[0] = o[1] = o[2] = to8(v);
        o[3] = 255;
        if (trns.size() >= 2 && v == (((uint32_t)trns[0] << 8) | trns[1])) o[3] = 0;
        break;
      }
      case 2: {
        uint32_t v[3];
        for (int c = 0; c < 3; ++c) { v[c] = sample(line, x, c); o[c] = to8(v[c]); }
        o[3] = 255;
        if (trns.size() >= 6) {
          bool eq = true;
          for (int c = 0; c < 3; ++c) eq = eq && v[c] == (((uint32_t)trns[2 * c] << 8) | trns[2 * c + 1]);
          if (eq) o[3] = 0;
        }
        break;
      }
      case 3: {
        const uint32_t idx = sample(line, x, 0);
        if ((size_t)idx * 3 + 2 < plte.size()) { o[0] = plte[idx * 3]; o[1] = plte[idx * 3 + 1]; o[2] = plte[idx * 3 + 2]; }
        else { o[0] = o[1] = o[2] = 0; }
        o[3] = idx < trns.size() ? trns[idx] : 255;
        break;
      }
      case 4:
        o[0] = o[1] = o[2] = to8(sample(line, x, 0));
        o[3] = to8(sample(line, x, 1));
        break;
      default:
        for (int c = 0; c < 4; ++c) o[c] = to8(sample(line, x, c));
        break;
    }
  };
  std::vector<uint8_t> lines;
  if (!interlace) {
    const size_t stride = ((size_t)W * bits_pp + 7) / 8;
    if (!unfilter(raw.data(), raw.size(), H, stride, bpp, lines)) { err = "corrupt PNG scanlines"; return false; }
    for (size_t y = 0; y < H; ++y)
      for (size_t x = 0; x < W; ++x) put(x, y, lines.data() + y * stride, x);
  } else { /* Adam7 */
    static const int xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1},
                     dy[7] = {8, 8, 8, 4, 4, 2, 2};
    size_t off = 0;
    for (int p = 0; p < 7; ++p) {
      const size_t pw = (W + dx[p] - 1 - xs[p]) / dx[p], ph = (H + dy[p] - 1 - ys[p]) / dy[p];
      if (W <= (uint32_t)xs[p] || H <= (uint32_t)ys[p] || pw == 0 || ph == 0) continue;
      const size_t stride = (pw * bits_pp + 7) / 8;
      const size_t used = unfilter(raw.data() + off, raw.size() - off, ph, stride, bpp, lines);
      if (!used) { err = "corrupt interlaced PNG"; return false; }
      off += used;
      for (size_t y = 0; y < ph; ++y)
        for (size_t x = 0; x < pw; ++x) put(xs[p] + x * dx[p], ys[p] + y * dy[p], lines.data() + y * stride, x);
    }
  }
  return true;
}

inline bool load(const std::string &path, Image &img, std::string &err) {
  std::vector<uint8_t> file;
  if (!read_file(path, file)) {
    err = "could not read " + path;
    return false;
  }
  return decode(file, img, err);
}

inline void chunk(std::vector<uint8_t> &out, const char *type, const std::vector<uint8_t> &data) {
  put_be32(out, (uint32_t)data.size());
  const size_t start = out.size();
  out.insert(out.end(), type, type + 4);
  out.insert(out.end(), data.begin(), data.end());
  const uint32_t crc = (uint32_t)crc32(0L, out.data() + start, (uInt)(out.size() - start));
  put_be32(out, crc);
}

/* ------------------------------------------------------------------------------------------------
 * Fast PNG writer for the frame files of `curvis video` (src/rendering.rs:291-316 saves one PNG per frame).
 * The reference's encoder (image 0.25 -> png 0.17, CompressionType::Fast) is a PNG-specific fast deflate; so is
 * this one, written from the formats (RFC 1950/1951, PNG 1.2), not from that crate: filter Up (Sub on the first
 * row), then ONE dynamic-Huffman deflate block over the filtered bytes in which the only matches are runs of zero
 * bytes (distance 1, length 3..258) -- no hash chains, no lazy matching.  Two passes over the filtered frame
 * (token histogram, then bit emission with a 64-bit accumulator): 0.3-1.5 GB/s per host thread depending on how
 * much of the frame is zero runs, where zlib level 1 manages 0.1-0.25 GB/s (profiles/round3_cli_video.txt).
 * The compressed bytes differ from the reference's; the decoded pixels are identical (tests decode the files with
 * this header's reader AND with Python's zlib). */
struct EncodeTimes { /* seconds, accumulated by the caller across frames (per-stage profile of `curvis video`) */
  double filter = 0, deflate = 0, checksum = 0, write = 0;
  size_t raw_bytes = 0, file_bytes = 0, frames = 0;
};
inline double now_s() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* code lengths (<= maxlen) of a Huffman code for freq[0..n), n <= 288; symbols with a zero count get no code.  The method is
 * png_codes.h's (shared with the device's png_codes_kernel): package-merge over the symbols in (count, index) order -- the optimal
 * code under the limit. */
inline void huffman_lengths(const uint32_t *freq, int n, int maxlen, uint8_t *len) {
  if (n < 0 || n > pngcodes::kMaxLeaves || maxlen < 1 || maxlen > pngcodes::kMaxLimit || (n > 1 && (1 << maxlen) < n))
    throw std::length_error("huffman_lengths: not an alphabet / code length of the deflate format"); /* (the callers pass 286 and 12) */
  uint64_t key[pngcodes::kMaxLeaves];
  int m = 0;
  for (int i = 0; i < n; ++i) {
    len[i] = 0;
    if (freq[i]) key[m++] = ((uint64_t)freq[i] << 9) | (uint64_t)i; /* n <= 288 < 512 */
  }
  if (m == 0) return;
  std::sort(key, key + m);
  uint32_t w[pngcodes::kMaxLeaves];
  uint8_t ls[pngcodes::kMaxLeaves];
  for (int k = 0; k < m; ++k) w[k] = (uint32_t)(key[k] >> 9);
  pngcodes::pm_lengths_sorted(w, m, maxlen, ls);
  for (int k = 0; k < m; ++k) len[key[k] & 511u] = ls[k];
}

/* canonical codes (RFC 1951 3.2.2), bit-reversed for an LSB-first bit writer; entry = code | len << 16 */
inline void canonical_codes(const uint8_t *len, int n, uint32_t *entry) {
  uint32_t count[16] = {0}, next[16] = {0};
  for (int i = 0; i < n; ++i) count[len[i]]++;
  count[0] = 0;
  uint32_t code = 0;
  for (int b = 1; b < 16; ++b) {
    code = (code + count[b - 1]) << 1;
    next[b] = code;
  }
  for (int i = 0; i < n; ++i) {
    const int l = len[i];
    uint32_t c = l ? next[l]++ : 0, r = 0;
    for (int k = 0; k < l; ++k) r |= ((c >> k) & 1u) << (l - 1 - k);
    entry[i] = r | ((uint32_t)l << 16);
  }
}

struct BitWriter { /* LSB-first, branch-free: an unaligned 8-byte store per put, the pointer advances by whole bytes */
  uint8_t *p;
  uint64_t acc = 0;
  unsigned nb = 0;
  explicit BitWriter(uint8_t *dst) : p(dst) {}
  inline void put(uint64_t bits, unsigned n) { /* nb <= 7 on entry, n <= 48 */
    acc |= bits << nb;
    nb += n;
    std::memcpy(p, &acc, 8); /* little-endian host (x86-64); the buffer carries 8 spare bytes */
    const unsigned adv = nb >> 3;
    p += adv;
    acc >>= adv * 8u;
    nb &= 7u;
  }
  inline uint8_t *finish() {
    if (nb) *p++ = (uint8_t)acc;
    nb = 0;
    acc = 0;
    return p;
  }
};

/* length -> (symbol 257.., extra bits, extra value) of RFC 1951 3.2.5 */
inline void length_symbol(int length, int &sym, int &ebits, int &eval) { pngcodes::length_symbol(length, sym, ebits, eval); }

inline uint32_t load32(const uint8_t *p) {
  uint32_t v;
  std::memcpy(&v, p, 4);
  return v;
}
inline uint64_t load64(const uint8_t *p) {
  uint64_t v;
  std::memcpy(&v, p, 8);
  return v;
}
/* end of the run of zero bytes that starts at d[i] (d[i..i+3] are known to be zero) */
inline size_t zero_run_end(const uint8_t *d, size_t i, size_t n) {
  size_t j = i + 4;
  while (j + 8 <= n && load64(d + j) == 0) j += 8;
  while (j < n && d[j] == 0) ++j;
  return j;
}
/* tokens of the filtered stream: every byte a literal, except that a run of r >= 4 zero bytes is one literal 0
 * followed by distance-1 matches covering the other r - 1 (pieces of 3..258).  `lit4` takes four literals at once
 * (one store of the bit writer: the code is limited to 12 bits per symbol), the test for a run is one 32-bit load. */
template <class Lit, class Lit4, class Match>
inline void for_each_token(const uint8_t *d, size_t n, Lit lit, Lit4 lit4, Match match) {
  size_t i = 0;
  while (i + 4 <= n) {
    const uint32_t w4 = load32(d + i);
    if (w4 != 0) { /* no run of four starts here: four literals (a run that starts inside them is picked up next time round,
                      up to three of its zeros having gone out as literals) */
      lit4(w4);
      i += 4;
      continue;
    }
    const size_t j = zero_run_end(d, i, n);
    size_t rem = j - i - 1; /* zeros after the first: >= 3 */
    lit(0);
    while (rem > 260) {
      match(258);
      rem -= 258;
    }
    if (rem > 258) {
      match((int)rem - 3);
      rem = 3;
    }
    match((int)rem);
    i = j;
  }
  for (; i < n; ++i) lit(d[i]);
}

/* header of the ONE dynamic-Huffman block a frame is written as: final block, 286 literal/length codes of the given lengths
 * (all non-zero), one distance code of length 1; the code lengths themselves go out raw, as 4-bit code-length codes.
 * 1222 bits. */
inline void put_dynamic_block_header(BitWriter &bw, const uint8_t ll_len[286]) {
  bw.put(1, 1);  /* BFINAL */
  bw.put(2, 2);  /* BTYPE = 10: dynamic Huffman */
  bw.put(29, 5); /* HLIT: 286 literal/length codes */
  bw.put(0, 5);  /* HDIST: 1 distance code */
  bw.put(15, 4); /* HCLEN: all 19 code-length codes */
  static const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  for (int k = 0; k < 19; ++k) bw.put(order[k] < 16 ? 4u : 0u, 3u); /* lengths 0..15 written raw as 4-bit codes, no repeats */
  auto rev4 = [](uint32_t v) { return ((v & 1u) << 3) | ((v & 2u) << 1) | ((v & 4u) >> 1) | ((v & 8u) >> 3); };
  for (int i = 0; i < 286; ++i) bw.put(rev4(ll_len[i]), 4);
  bw.put(rev4(1), 4); /* the one distance code: length 1 */
}

/* the whole PNG file of an RGB8 image into `out` */
/* Returns the length of the PNG file now at the start of `out`.  `out` is scratch that is only ever GROWN (to the worst-case
 * size of this frame): a caller that reuses it across frames pays the allocation and its zero fill once, not per frame. */
inline size_t encode_rgb8_fast(const uint8_t *rgb, uint32_t w, uint32_t h, std::vector<uint8_t> &out, EncodeTimes *tm = nullptr) {
  const size_t stride = (size_t)w * 3, line = stride + 1, n = line * h;
  double t0 = now_s();
  /* per-thread scratch kept across frames: a fresh std::vector would zero-fill (and page-fault) 3x the frame size each time */
  static thread_local std::vector<uint8_t> flt_buf;
  if (flt_buf.size() < n) flt_buf.resize(n);
  uint8_t *const fl = flt_buf.data();
  for (uint32_t y = 0; y < h; ++y) {
    const uint8_t *cur = rgb + (size_t)y * stride;
    uint8_t *dst = fl + (size_t)y * line;
    if (y == 0) { /* Sub: left neighbour (3 bytes back) */
      dst[0] = 1;
      for (size_t i = 0; i < stride; ++i) dst[1 + i] = (uint8_t)(cur[i] - (i >= 3 ? cur[i - 3] : 0));
    } else { /* Up, eight bytes per operation: per-byte a - b = ((a | H) - (b & ~H)) ^ ((a ^ ~b) & H), H = 0x80 in every byte */
      const uint8_t *prev = cur - stride;
      dst[0] = 2;
      const uint64_t H = 0x8080808080808080ULL;
      size_t i = 0;
      for (; i + 8 <= stride; i += 8) {
        const uint64_t a = load64(cur + i), b = load64(prev + i);
        const uint64_t r = ((a | H) - (b & ~H)) ^ ((a ^ ~b) & H);
        std::memcpy(dst + 1 + i, &r, 8);
      }
      for (; i < stride; ++i) dst[1 + i] = (uint8_t)(cur[i] - prev[i]);
    }
  }
  double t1 = now_s();
  uLong ad = adler32(0L, Z_NULL, 0);
  for (size_t off = 0; off < n;) { /* uInt-sized pieces */
    const size_t piece = std::min<size_t>(n - off, (size_t)1 << 30);
    ad = adler32(ad, fl + off, (uInt)piece);
    off += piece;
  }
  double t2 = now_s();
  /* pass 1: token histogram of every 4th row (each taken as a stream of its own); every symbol gets a count of at
   * least one so that the code covers whatever the other rows hold -- a quarter of the work for ~1 % of file size */
  uint32_t hist[286], hist2[256] = {0}, hist3[256] = {0}, hist4[256] = {0}, hist_len[259] = {0};
  for (int i = 0; i < 286; ++i) hist[i] = 1;
  for (uint32_t y = 0; y < h; y += 4)
    for_each_token(fl + (size_t)y * line, line, [&](uint8_t b) { hist[b]++; },
                   [&](uint32_t w4) { /* four tables: no store-to-load stall when neighbours are equal */
                     hist[w4 & 0xffu]++;
                     hist2[(w4 >> 8) & 0xffu]++;
                     hist3[(w4 >> 16) & 0xffu]++;
                     hist4[w4 >> 24]++;
                   },
                   [&](int len) { hist_len[len]++; });
  for (int i = 0; i < 256; ++i) hist[i] += hist2[i] + hist3[i] + hist4[i];
  for (int len = 3; len <= 258; ++len)
    if (hist_len[len]) {
      int sym, eb, ev;
      length_symbol(len, sym, eb, ev);
      hist[sym] += hist_len[len];
    }
  uint8_t ll_len[286];
  uint32_t ll[286];
  huffman_lengths(hist, 286, 12, ll_len); /* <= 12 bits per code: four literals fit one 64-bit put (48 + 7 pending bits) */
  canonical_codes(ll_len, 286, ll);
  /* matches of length 3..258: code, extra bits and the 1-bit distance code ('0') folded into one entry */
  uint32_t m_bits[259];
  uint8_t m_n[259];
  for (int len = 3; len <= 258; ++len) {
    int sym, eb, ev;
    length_symbol(len, sym, eb, ev);
    const int cl = (int)(ll[sym] >> 16);
    m_bits[len] = (ll[sym] & 0xffffu) | ((uint32_t)ev << cl); /* + distance code 0 (one zero bit) */
    m_n[len] = (uint8_t)(cl + eb + 1);
  }
  /* worst case: 12 bits per byte; the buffer is sized for 16 */
  const size_t cap = 8 + 25 + 12 + 2 + 512 + n * 2 + 16 + 4 + 12;
  if (out.size() < cap) out.resize(cap);
  uint8_t *o = out.data();
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  std::memcpy(o, sig, 8);
  o += 8;
  auto be = [](uint8_t *q, uint32_t v) {
    q[0] = (uint8_t)(v >> 24);
    q[1] = (uint8_t)(v >> 16);
    q[2] = (uint8_t)(v >> 8);
    q[3] = (uint8_t)v;
  };
  be(o, 13);
  std::memcpy(o + 4, "IHDR", 4);
  be(o + 8, w);
  be(o + 12, h);
  o[16] = 8;
  o[17] = 2;
  o[18] = o[19] = o[20] = 0;
  be(o + 21, (uint32_t)crc32(0L, o + 4, 17));
  o += 25;
  uint8_t *idat = o; /* length filled in afterwards */
  std::memcpy(o + 4, "IDAT", 4);
  o += 8;
  *o++ = 0x78;
  *o++ = 0x01;
  BitWriter bw(o);
  put_dynamic_block_header(bw, ll_len);
  for_each_token(fl, n, [&](uint8_t b) { bw.put(ll[b] & 0xffffu, ll[b] >> 16); },
                 [&](uint32_t w4) { /* four codes (<= 12 bits each) in one store */
                   const uint32_t e0 = ll[w4 & 0xffu], e1 = ll[(w4 >> 8) & 0xffu], e2 = ll[(w4 >> 16) & 0xffu], e3 = ll[w4 >> 24];
                   const unsigned l0 = e0 >> 16, l1 = e1 >> 16, l2 = e2 >> 16, l3 = e3 >> 16;
                   const uint64_t lo = (uint64_t)(e0 & 0xffffu) | ((uint64_t)(e1 & 0xffffu) << l0);
                   const uint64_t hi = (uint64_t)(e2 & 0xffffu) | ((uint64_t)(e3 & 0xffffu) << l2);
                   bw.put(lo | (hi << (l0 + l1)), l0 + l1 + l2 + l3);
                 },
                 [&](int len) { bw.put(m_bits[len], m_n[len]); });
  bw.put(ll[256] & 0xffffu, ll[256] >> 16);
  o = bw.finish();
  be(o, (uint32_t)ad);
  o += 4;
  const size_t idat_len = (size_t)(o - (idat + 8));
  be(idat, (uint32_t)idat_len);
  double t3 = now_s();
  uLong crc = crc32(0L, Z_NULL, 0);
  for (size_t off = 0; off < idat_len + 4;) {
    const size_t piece = std::min<size_t>(idat_len + 4 - off, (size_t)1 << 30);
    crc = crc32(crc, idat + 4 + off, (uInt)piece);
    off += piece;
  }
  be(o, (uint32_t)crc);
  o += 4;
  be(o, 0);
  std::memcpy(o + 4, "IEND", 4);
  be(o + 8, (uint32_t)crc32(0L, o + 4, 4));
  o += 12;
  const size_t file_len = (size_t)(o - out.data());
  double t4 = now_s();
  if (tm) {
    tm->filter += t1 - t0;
    tm->checksum += (t2 - t1) + (t4 - t3);
    tm->deflate += t3 - t2;
    tm->raw_bytes += stride * h;
    tm->file_bytes += file_len;
    tm->frames += 1;
  }
  return file_len;
}

inline bool write_file(const std::string &path, const uint8_t *data, size_t n, std::string &err) {
  FILE *f = std::fopen(path.c_str(), "wb");
  if (!f) {
    err = "could not open " + path + " for writing";
    return false;
  }
  const bool ok = std::fwrite(data, 1, n, f) == n;
  std::fclose(f);
  if (!ok) err = "short write to " + path;
  return ok;
}

/* A PNG file around a finished zlib stream of the filtered scanlines (what curvis_ctx_deflate_frames hands back: filtering,
 * Huffman coding and Adler-32 were done on the GPU): signature, IHDR, IDAT chunk(s) with their CRC-32, IEND. */
inline bool save_zlib_stream_rgb8(const std::string &path, const uint8_t *z, size_t len, uint32_t w, uint32_t h, std::string &err,
                                  EncodeTimes *tm = nullptr, const uint32_t *idat_crc = nullptr) {
  const double t0 = now_s();
  FILE *f = std::fopen(path.c_str(), "wb");
  if (!f) {
    err = "could not open " + path + " for writing";
    return false;
  }
  auto be = [](uint8_t *q, uint32_t v) {
    q[0] = (uint8_t)(v >> 24);
    q[1] = (uint8_t)(v >> 16);
    q[2] = (uint8_t)(v >> 8);
    q[3] = (uint8_t)v;
  };
  uint8_t head[8 + 25 + 8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  uint8_t *o = head + 8;
  be(o, 13);
  std::memcpy(o + 4, "IHDR", 4);
  be(o + 8, w);
  be(o + 12, h);
  o[16] = 8;
  o[17] = 2;
  o[18] = o[19] = o[20] = 0;
  be(o + 21, (uint32_t)crc32(0L, o + 4, 17));
  bool ok = std::fwrite(head, 1, 8 + 25, f) == 8 + 25;
  double t_crc = 0.0;
  const size_t kIdatMax = (size_t)1 << 30; /* a chunk stays below 2^31 bytes */
  for (size_t off = 0; ok && (off < len || off == 0); off += kIdatMax) {
    const size_t piece = std::min(kIdatMax, len - off);
    uint8_t ch[8], tail[4];
    be(ch, (uint32_t)piece);
    std::memcpy(ch + 4, "IDAT", 4);
    const double tc = now_s();
    uLong crc;
    if (idat_crc && len <= kIdatMax) { /* computed on the device (curvis_ctx_deflate_frames_crc): one chunk, nothing to do here */
      crc = *idat_crc;
    } else {
      crc = crc32(0L, ch + 4, 4);
      crc = crc32(crc, z + off, (uInt)piece);
    }
    t_crc += now_s() - tc;
    be(tail, (uint32_t)crc);
    ok = std::fwrite(ch, 1, 8, f) == 8 && std::fwrite(z + off, 1, piece, f) == piece && std::fwrite(tail, 1, 4, f) == 4;
    if (len == 0) break;
  }
  uint8_t end[12];
  be(end, 0);
  std::memcpy(end + 4, "IEND", 4);
  be(end + 8, (uint32_t)crc32(0L, end + 4, 4));
  ok = ok && std::fwrite(end, 1, 12, f) == 12;
  ok = (std::fclose(f) == 0) && ok;
  if (!ok) err = "short write to " + path;
  if (tm) {
    tm->checksum += t_crc;
    tm->write += (now_s() - t0) - t_crc;
    tm->raw_bytes += (size_t)w * 3 * h;
    tm->file_bytes += 8 + 25 + 12 + len + 12 * ((len + kIdatMax - 1) / kIdatMax);
    tm->frames += 1;
  }
  return ok;
}

/* level < 0: the fast writer above; 0..9: zlib at that level, filter type 0 */
inline bool save_rgb8(const std::string &path, const uint8_t *rgb, uint32_t w, uint32_t h, std::string &err, int level = 6,
                      EncodeTimes *tm = nullptr) {
  if (level < 0) {
    /* A chunk length must stay below 2^31 (PNG spec; libpng and the `png` crate reject longer ones) and the fast deflate's
     * worst case is 12 bits per byte: frames whose WORST-CASE stream would not fit one IDAT take the zlib route, which
     * splits its stream into several IDAT chunks below. */
    if (((uint64_t)w * 3 * h + h) * 3 / 2 + 1024 >= ((uint64_t)1 << 31)) {
      level = 1;
    } else {
      static thread_local std::vector<uint8_t> file;
      const size_t file_len = encode_rgb8_fast(rgb, w, h, file, tm);
      const double t0 = now_s();
      const bool ok = write_file(path, file.data(), file_len, err);
      if (tm) tm->write += now_s() - t0;
      return ok;
    }
  }
  const double ta = now_s();

  std::vector<uint8_t> raw((size_t)h * ((size_t)w * 3 + 1));
  for (uint32_t y = 0; y < h; ++y) {
    raw[(size_t)y * (w * 3 + 1)] = 0;
    std::memcpy(&raw[(size_t)y * (w * 3 + 1) + 1], rgb + (size_t)y * w * 3, (size_t)w * 3);
  }
  uLongf clen = compressBound((uLong)raw.size());
  std::vector<uint8_t> comp(clen);
  if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), level) != Z_OK) {
    err = "zlib compress failed";
    return false;
  }
  comp.resize(clen);
  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  std::vector<uint8_t> ihdr;
  put_be32(ihdr, w);
  put_be32(ihdr, h);
  ihdr.push_back(8);
  ihdr.push_back(2);
  ihdr.push_back(0);
  ihdr.push_back(0);
  ihdr.push_back(0);
  chunk(out, "IHDR", ihdr);
  /* several IDAT chunks when the stream is long: a single chunk must stay below 2^31 bytes */
  const size_t kIdatMax = (size_t)1 << 30;
  for (size_t off = 0; off < comp.size() || off == 0; off += kIdatMax) {
    const size_t piece = std::min(kIdatMax, comp.size() - off);
    chunk(out, "IDAT", std::vector<uint8_t>(comp.begin() + (ptrdiff_t)off, comp.begin() + (ptrdiff_t)(off + piece)));
    if (comp.empty()) break;
  }
  chunk(out, "IEND", {});
  const double tb = now_s();
  const bool ok = write_file(path, out.data(), out.size(), err);
  if (tm) {
    tm->deflate += tb - ta;
    tm->write += now_s() - tb;
    tm->raw_bytes += (size_t)w * 3 * h;
    tm->file_bytes += out.size();
    tm->frames += 1;
  }
  return ok;
}

}  // namespace pngio
#endif
