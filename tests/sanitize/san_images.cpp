// san_images.cpp -- TEST INFRASTRUCTURE: the host binary's PNG and JPEG decoders (curvis_amd/csrc/host/png_io.h,
// jpeg_io.h) under AddressSanitizer + UndefinedBehaviorSanitizer on hostile input.  Decodes every file named on
// the command line; prints one line per file.  The sanitizers abort on the first finding.
#include <cstdio>
#include <string>

#include "../../curvis_amd/csrc/host/jpeg_io.h"

int main(int argc, char **argv) {
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
