// san_images.cpp -- TEST INFRASTRUCTURE: the host binary's PNG and JPEG decoders (curvis_amd/csrc/host/png_io.h,
// jpeg_io.h) under AddressSanitizer + UndefinedBehaviorSanitizer on hostile input.  Decodes every file named on
// the command line; prints one line per file.  The sanitizers abort on the first finding.
#include <cstdio>
#include <string>
#include <memory>
#include <vector>

#include "../../curvis_amd/csrc/host/jpeg_io.h"

// `san_images --encode`: the fast PNG writer (pngio::encode_rgb8_fast) on a few hundred small images of awkward shapes
// and contents, decoded again by pngio::decode and compared.
static int encode_round_trips() {
  uint32_t state = 12345u;
  auto rnd = [&]() {
    state ^= state << 13;
    state ^= state >> 17;
    state ^= state << 5;
    return state;
  };
  int n = 0;
  std::vector<uint8_t> file;
  for (int it = 0; it < 400; ++it) {
    const uint32_t w = 1 + rnd() % (it % 7 == 0 ? 400 : 40), h = 1 + rnd() % 40;
    std::vector<uint8_t> rgb((size_t)w * h * 3);
    const int kind = it % 5;
    for (size_t i = 0; i < rgb.size(); ++i) {
      switch (kind) {
        case 0: rgb[i] = 0; break;                                        // one zero run over everything
        case 1: rgb[i] = (uint8_t)rnd(); break;                           // noise
        case 2: rgb[i] = (rnd() % 97 == 0) ? (uint8_t)rnd() : 7; break;    // long runs with sparse breaks
        case 3: rgb[i] = (uint8_t)((i / 3) % w * 255 / w); break;          // horizontal gradient (Up residual zero)
        default: rgb[i] = (uint8_t)(rnd() & 3); break;                    // short runs and literals
      }
    }
    file.clear();
    const size_t file_len = pngio::encode_rgb8_fast(rgb.data(), w, h, file); /* `file` is scratch, kept at capacity across frames */
    const std::vector<uint8_t> png(file.begin(), file.begin() + (ptrdiff_t)file_len);
    pngio::Image img;
    std::string err;
    if (!pngio::decode(png, img, err) || img.w != w || img.h != h) {
      std::printf("encode round trip %d (%ux%u kind %d): decode failed: %s\n", it, w, h, kind, err.c_str());
      return 1;
    }
    for (size_t px = 0; px < (size_t)w * h; ++px)
      if (img.rgba[px * 4] != rgb[px * 3] || img.rgba[px * 4 + 1] != rgb[px * 3 + 1] || img.rgba[px * 4 + 2] != rgb[px * 3 + 2]) {
        std::printf("encode round trip %d (%ux%u kind %d): pixel %zu differs\n", it, w, h, kind, px);
        return 1;
      }
    ++n;
  }
  std::printf("encoded and decoded %d images\n", n);
  return 0;
}

// `san_images --inflate`: inflate_fast.h on streams zlib made (levels 0 / 1 / 6 / 9, fixed and dynamic blocks) from contents of
// several kinds -- intact (must give the input back) and then damaged a few thousand times (bit flips, cuts, junk): heap buffers
// of exactly the size the decoder is told, so that AddressSanitizer sees a single byte too far; any status is fine, a finding is not.
static int inflate_fuzz() {
  uint32_t state = 777u;
  auto rnd = [&]() {
    state ^= state << 13;
    state ^= state >> 17;
    state ^= state << 5;
    return state;
  };
  int intact = 0, damaged_ok = 0, damaged_err = 0;
  for (int it = 0; it < 60; ++it) {
    const size_t n = it % 6 == 0 ? 0 : 1 + rnd() % (it % 5 == 0 ? 300000 : 20000);
    std::vector<uint8_t> data(n);
    for (size_t i = 0; i < n; ++i) {
      switch (it % 4) {
        case 0: data[i] = (uint8_t)rnd(); break;
        case 1: data[i] = (uint8_t)(rnd() % 97 == 0 ? rnd() : 0); break;
        case 2: data[i] = (uint8_t)(i % (1 + it)); break;
        default: data[i] = (uint8_t)(rnd() & 7); break;
      }
    }
    static const int levels[4] = {0, 1, 6, 9};
    for (int lv = 0; lv < 5; ++lv) {
      z_stream zs;
      std::memset(&zs, 0, sizeof zs);
      if (deflateInit2(&zs, lv < 4 ? levels[lv] : 6, Z_DEFLATED, 15, 8, lv < 4 ? Z_DEFAULT_STRATEGY : Z_FIXED) != Z_OK) return 1;
      std::vector<uint8_t> z(deflateBound(&zs, (uLong)n) + 16);
      zs.next_in = data.data();
      zs.avail_in = (uInt)n;
      zs.next_out = z.data();
      zs.avail_out = (uInt)z.size();
      if (deflate(&zs, Z_FINISH) != Z_STREAM_END) return 1;
      z.resize(zs.total_out);
      deflateEnd(&zs);
      auto run = [&](const std::vector<uint8_t> &stream, size_t cap, std::vector<uint8_t> &out, size_t &got) {
        std::unique_ptr<uint8_t[]> in(new uint8_t[stream.size() + cvinflate::kInputPadding]); /* exactly the padding asked for */
        std::memcpy(in.get(), stream.data(), stream.size());
        std::memset(in.get() + stream.size(), 0, cvinflate::kInputPadding);
        out.assign(cap, 0);
        uint32_t ad = 0;
        got = 0;
        return cvinflate::inflate_zlib(in.get(), stream.size(), out.data(), cap, &got, &ad);
      };
      std::vector<uint8_t> out;
      size_t got = 0;
      if (run(z, n, out, got) != cvinflate::OK || got != n || (n && std::memcmp(out.data(), data.data(), n) != 0)) {
        std::printf("inflate: intact stream %d level %d did not round-trip\n", it, lv);
        return 1;
      }
      ++intact;
      for (int d = 0; d < 12; ++d) {
        std::vector<uint8_t> b = z;
        if (d % 3 == 0 && b.size() > 3)
          for (int k = 0; k < 1 + (int)(rnd() % 3); ++k) b[2 + rnd() % (b.size() - 2)] ^= (uint8_t)(1u << (rnd() % 8));
        else if (d % 3 == 1)
          b.resize(rnd() % (b.size() + 1));
        else if (b.size() > 6)
          for (int k = 0; k < 4; ++k) b[2 + rnd() % (b.size() - 2)] = (uint8_t)rnd();
        const size_t cap = d % 4 == 0 ? n / 2 : n + rnd() % 64; /* sometimes too little room */
        (run(b, cap, out, got) == cvinflate::OK ? damaged_ok : damaged_err)++;
      }
    }
  }
  std::printf("inflated %d intact streams, %d damaged ones accepted, %d rejected\n", intact, damaged_ok, damaged_err);
  return 0;
}

int main(int argc, char **argv) {
  if (argc == 2 && std::string(argv[1]) == "--encode") return encode_round_trips();
  if (argc == 2 && std::string(argv[1]) == "--inflate") return inflate_fuzz();
  int ok = 0, bad = 0;
  for (int i = 1; i < argc; ++i) {
    pngio::Image img;
    std::string err;
    if (jpegio::load_image(argv[i], img, err) && img.rgba.size() == (size_t)img.w * img.h * 4)
      ++ok;
    else
      ++bad;
  }
  std::printf("decoded %d, rejected %d\n", ok, bad);
  return 0;
}
