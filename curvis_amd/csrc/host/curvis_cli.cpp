/* curvis_cli.cpp -- `curvis image | video | custom`: the reference's command line (src/cli.rs:35-122,
 * src/main.rs:135-235) over libcurvis_hip.so.
 *
 *   curvis image <IMAGE FILE 1> <IMAGE FILE 2> [OUTPUT FOLDER] [-i toml] [-m toml] [-c toml] [-s toml]
 *   curvis video <IMAGE FILE 1> <IMAGE FILE 2> [OUTPUT FOLDER] [-v toml] [-m toml] [-c toml] [-s toml]
 *   curvis custom
 *
 * Same positional arguments, flags, TOML keys (including `ray_integration_max_itarations`,
 * src/settings.rs:121), defaults (settings/defaults/\*.toml), validation messages, output names
 * (`<out>/<image_name>.png`, `<out>/tmp/frame_{k}.png`, src/rendering.rs:108, :296) and the reference's
 * observable quirks: max_iterations_sampling is wired to sampling_initial_nums (src/main.rs:47, :107),
 * the video path passes sampling_convergence_threshold_1 twice (src/rendering.rs:305-306), and the
 * camera-path interpolator reads one row too far in the last CSV segment (src/interpolation.rs:76-90:
 * the reference panics there; this binary reports the same condition and exits with status 101).
 *
 * Extensions (do not exist in the reference): --mode efficient|brute (default efficient = what the
 * reference renders; brute = RelativisticSystem::render_image, the per-pixel integrator),
 * --devices N (image --mode brute: rows of the frame split over N GPUs; video: frames k mod N across N GPUs, one host thread + one context per GPU, skies uploaded to
 * each or broadcast from GPU 0 with RCCL: --sky-broadcast rccl|upload), --batch B (frames per kernel launch),
 * --writers T (PNG encoder threads; default: a quarter of the host threads, 4..64), --stats FILE (JSON lines, one per
 * frame, with that frame's own early-termination counters: rays, executed Euler steps, escaped +l / -l, capped),
 * --resume (video: keep <out>/tmp and skip the frames whose frame_{k}.png is already there; default off = the
 * reference's behaviour of deleting and recreating tmp, src/rendering.rs:276-287), --gpu-png auto|on|off (video: filter + Huffman
 * coding + Adler-32 of every frame on the GPU, curvis_ctx_deflate_frames -- the frames never cross PCIe as pixels and a writer
 * thread only wraps the stream into a PNG file; auto = with the fast writer, in modes brute and efficient),
 * --contexts-per-device C (video: C worker threads with a context each per GPU, frames k mod (N*C); default 2 in --mode
 * efficient, whose render call is half host work, else 1).  A batch of frames whose render
 * call fails is re-queued on another GPU (frames are independent) before the run is declared failed.
 * Backgrounds: PNG (any colour type / bit depth) or JPEG (8-bit Huffman, baseline / progressive, grey or YCbCr; own
 * decoder in jpeg_io.h -- JPEG input is outside the pixel-parity claims, see there).
 */
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <filesystem>
#include <functional>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <sstream>
#include <string>
#include <future>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "../../../include/curvis_hip.h"
#include "png_io.h"
#include "jpeg_io.h"
/* the binary's parts, in dependency order (each opens its own anonymous namespace) */
#include "cli_settings.h"
#include "cli_args.h"
#include "cli_image.h"
#include "cli_writers.h"
#include "cli_video.h"

int main(int argc, char **argv) {
  const Args a = parse_args(argc, argv);
  if (a.sub == "image" || a.sub == "video") {
    const int rc = a.sub == "image" ? image_main(a) : video_main(a);
    /* every output file has been written and closed, every context destroyed: leave without the HIP runtime's exit handlers
     * (code objects, queues, its worker threads -- ~0.1 s of a 0.4 s `curvis image`); CURVIS_SLOW_EXIT=1 takes the long way */
    std::fflush(stdout);
    std::fflush(stderr);
    if (!std::getenv("CURVIS_SLOW_EXIT")) std::_Exit(rc);
    return rc;
  }
  if (a.sub == "selftest-png") { /* hidden: decode <in.png> the way skies are decoded, dump RGBA8 to <out> */
    if (argc != 4) die("usage: curvis selftest-png <in.png> <out.rgba>", 2);
    pngio::Image img;
    std::string err;
    if (!jpegio::load_image(argv[2], img, err)) die("selftest-png: " + err);
    FILE *f = std::fopen(argv[3], "wb");
    if (!f) die("selftest-png: cannot write output");
    const uint32_t hdr[2] = {img.w, img.h};
    std::fwrite(hdr, sizeof hdr, 1, f);
    std::fwrite(img.rgba.data(), 1, img.rgba.size(), f);
    std::fclose(f);
    return 0;
  }
  if (a.sub == "custom") { /* src/custom.rs: not implemented in the reference either */
    std::printf("Custom script\n");
    die("Error in curstom script: not implemented");
  }
  die("Unrecognized subcommand \"" + a.sub + "\"");
}
