/* png_io.h -- minimal PNG reader/writer over zlib for the curvis host binary.
 *
 * Reader: 8/16-bit grey, grey+alpha, RGB, RGBA, palette (with tRNS), non-interlaced and Adam7,
 * expanded to RGBA8 the way the reference obtains texels (image 0.25.2: DynamicImage::get_pixel ->
 * Rgba<u8>; 16-bit samples are reduced with (v + 128) / 257, grey is replicated, missing alpha = 255)
 * -- src/images.rs:8, :107-111.
 * Writer: RGB8, one IDAT, filter type 0 (src/rendering.rs:110, :311 save an ImageRgb8 as PNG; the
 * compressed bytes differ from the image crate's encoder, the decoded pixels are identical).
 */
#ifndef CURVIS_PNG_IO_H
#define CURVIS_PNG_IO_H
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

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

inline uint8_t reduce16(uint32_t v) { return (uint8_t)((v + 128u) / 257u); } /* image crate u16 -> u8 */

inline bool decode(const std::vector<uint8_t> &file, Image &img, std::string &err) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) {
    err = "not a PNG file (only PNG backgrounds are supported by this build; convert JPEG skies to PNG)";
    return false;
  }
  size_t pos = 8;
  uint32_t W = 0, H = 0;
  int depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, plte, trns;
  bool have_ihdr = false;
  while (pos + 12 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    const char *type = (const char *)&file[pos + 4];
    if (pos + 12 + (size_t)len > file.size()) {
      err = "truncated PNG chunk";
      return false;
    }
    const uint8_t *data = &file[pos + 8];
    /* every chunk carries a CRC-32 of type + data (the png crate rejects a mismatch for critical chunks) */
    if ((uint32_t)crc32(0L, &file[pos + 4], (uInt)(4 + len)) != be32(&file[pos + 8 + len])) {
      err = "PNG chunk CRC mismatch";
      return false;
    }
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len != 13) { err = "bad IHDR"; return false; }
      W = be32(data); H = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
      have_ihdr = true;
    } else if (!std::memcmp(type, "PLTE", 4)) {
      plte.assign(data, data + len);
    } else if (!std::memcmp(type, "tRNS", 4)) {
      trns.assign(data, data + len);
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      break;
    }
    pos += 12 + (size_t)len;
  }
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
  /* inflate */
  std::vector<uint8_t> raw;
  {
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (inflateInit(&zs) != Z_OK) { err = "zlib init failed"; return false; }
    zs.next_in = idat.data();
    zs.avail_in = (uInt)idat.size();
    /* the inflated stream of a valid file is exactly the filtered scanlines; an interlaced image adds at most one
     * filter byte per pass row (< 2 H rows in total) plus rounding: the buffer never grows beyond that, so a
     * decompression bomb fails instead of exhausting memory */
    const size_t cap = (((size_t)W * bits_pp + 7) / 8 + 1) * ((size_t)H + 8) * (interlace ? 2 : 1) + 64;
    raw.resize(cap);
    size_t have = 0;
    int rc;
    do {
      if (have == raw.size()) { inflateEnd(&zs); err = "PNG data stream larger than its header allows"; return false; }
      zs.next_out = raw.data() + have;
      zs.avail_out = (uInt)std::min<size_t>(raw.size() - have, 1u << 30);
      rc = inflate(&zs, Z_NO_FLUSH);
      have = zs.total_out;
    } while (rc == Z_OK);
    inflateEnd(&zs);
    if (rc != Z_STREAM_END) { err = "corrupt PNG data stream"; return false; }
    raw.resize(have);
  }
  img.w = W;
  img.h = H;
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

inline bool save_rgb8(const std::string &path, const uint8_t *rgb, uint32_t w, uint32_t h, std::string &err, int level = 6) {
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
  chunk(out, "IDAT", comp);
  chunk(out, "IEND", {});
  FILE *f = std::fopen(path.c_str(), "wb");
  if (!f) {
    err = "could not open " + path + " for writing";
    return false;
  }
  const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
  std::fclose(f);
  if (!ok) err = "short write to " + path;
  return ok;
}

}  // namespace pngio
#endif
