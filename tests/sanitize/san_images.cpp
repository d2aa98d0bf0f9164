// san_images.cpp -- TEST INFRASTRUCTURE: the host binary's PNG and JPEG decoders (curvis_amd/csrc/host/png_io.h,
// jpeg_io.h) under AddressSanitizer + UndefinedBehaviorSanitizer on hostile input.  Decodes every file named on
// the command line; prints one line per file.  The sanitizers abort on the first finding.
#include <cstdio>
#include <string>
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

int main(int argc, char **argv) {
  if (argc == 2 && std::string(argv[1]) == "--encode") return encode_round_trips();
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
