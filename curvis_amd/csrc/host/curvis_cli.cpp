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
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "../../../include/curvis_hip.h"
#include "png_io.h"
#include "jpeg_io.h"

namespace {

/* ------------------------------------------------------------------ tiny TOML subset (key = value) */
struct TomlValue {
  enum Kind { STRING, INTEGER, FLOAT, BOOLEAN } kind = STRING;
  std::string s;
  long long i = 0;
  double f = 0.0;
};
typedef std::map<std::string, TomlValue> TomlTable;

bool parse_toml(const std::string &text, TomlTable &out, std::string &err) {
  std::istringstream in(text);
  std::string line;
  int ln = 0;
  while (std::getline(in, line)) {
    ++ln;
    /* strip comments outside strings */
    bool in_str = false;
    size_t cut = std::string::npos;
    for (size_t k = 0; k < line.size(); ++k) {
      if (line[k] == '"' && (k == 0 || line[k - 1] != '\\')) in_str = !in_str;
      if (line[k] == '#' && !in_str) {
        cut = k;
        break;
      }
    }
    if (cut != std::string::npos) line.resize(cut);
    auto trim = [](std::string &s) {
      size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
      s = (a == std::string::npos) ? std::string() : s.substr(a, b - a + 1);
    };
    trim(line);
    if (line.empty()) continue;
    if (line[0] == '[') continue; /* tables are not used by the reference's settings */
    const size_t eq = line.find('=');
    if (eq == std::string::npos) {
      err = "line " + std::to_string(ln) + ": expected key = value";
      return false;
    }
    std::string key = line.substr(0, eq), val = line.substr(eq + 1);
    trim(key);
    trim(val);
    if (key.empty() || val.empty()) {
      err = "line " + std::to_string(ln) + ": empty key or value";
      return false;
    }
    TomlValue v;
    if (val[0] == '"' || val[0] == '\'') {
      const char q = val[0];
      const size_t end = val.find_last_of(q);
      if (end == 0) {
        err = "line " + std::to_string(ln) + ": unterminated string";
        return false;
      }
      v.kind = TomlValue::STRING;
      v.s = val.substr(1, end - 1);
    } else if (val == "true" || val == "false") {
      v.kind = TomlValue::BOOLEAN;
      v.i = val == "true";
    } else {
      std::string num;
      for (char ch : val)
        if (ch != '_') num.push_back(ch);
      const bool is_float = num.find_first_of(".eE") != std::string::npos || num == "inf" || num == "nan" ||
                            num == "+inf" || num == "-inf";
      char *endp = nullptr;
      errno = 0;
      if (is_float) {
        v.kind = TomlValue::FLOAT;
        v.f = std::strtod(num.c_str(), &endp);
      } else {
        v.kind = TomlValue::INTEGER;
        v.i = std::strtoll(num.c_str(), &endp, 10);
        v.f = (double)v.i;
      }
      if (!endp || *endp != 0 || errno == ERANGE) {
        err = "line " + std::to_string(ln) + ": invalid value `" + val + "`";
        return false;
      }
    }
    out[key] = v;
  }
  return true;
}

/* serde semantics: missing field = error; an integer deserialises into f64; a float does not into u32 */
bool get_f64(const TomlTable &t, const char *k, double &out, std::string &err) {
  auto it = t.find(k);
  if (it == t.end()) {
    err = std::string("missing field `") + k + "`";
    return false;
  }
  if (it->second.kind != TomlValue::FLOAT && it->second.kind != TomlValue::INTEGER) {
    err = std::string("invalid type for `") + k + "`, expected f64";
    return false;
  }
  out = it->second.f;
  return true;
}
bool get_u32(const TomlTable &t, const char *k, uint32_t &out, std::string &err) {
  auto it = t.find(k);
  if (it == t.end()) {
    err = std::string("missing field `") + k + "`";
    return false;
  }
  if (it->second.kind != TomlValue::INTEGER || it->second.i < 0 || it->second.i > 4294967295LL) {
    err = std::string("invalid type or range for `") + k + "`, expected u32";
    return false;
  }
  out = (uint32_t)it->second.i;
  return true;
}
bool get_str(const TomlTable &t, const char *k, std::string &out, std::string &err) {
  auto it = t.find(k);
  if (it == t.end()) {
    err = std::string("missing field `") + k + "`";
    return false;
  }
  if (it->second.kind != TomlValue::STRING) {
    err = std::string("invalid type for `") + k + "`, expected a string";
    return false;
  }
  out = it->second.s;
  return true;
}

bool read_text(const std::string &path, std::string &out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::ostringstream ss;
  ss << f.rdbuf();
  out = ss.str();
  return true;
}
bool path_exists(const std::string &p) {
  struct stat st;
  return ::stat(p.c_str(), &st) == 0;
}
bool is_dir(const std::string &p) {
  struct stat st;
  return ::stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}

/* ------------------------------------------------------------------ settings (src/settings.rs:22-217) */
struct VideoSettings {
  std::string video_name = "output_video";
  double frame_rate = 30.0;
  std::string filepath_to_camera_path = "paths/path_through.csv";
};
struct ImageSettings {
  std::string image_name = "output_image";
  double t = 0.0, l = 5.0, theta = 1.5707963267948966192313216916398, phi = 0.0;
  double forward_x = -1.0, forward_y = 0.0, forward_z = 0.0, up_x = 0.0, up_y = 0.0, up_z = 1.0;
};
struct CameraSettings {
  uint32_t resolution_x = 960, resolution_y = 540;
  double diagonal = 43.0, focal_length = 15.0;
};
struct SimulationSettings {
  double escape_radius = 100.0;
  uint32_t ray_integration_max_itarations = 40000;
  double ray_integration_step = 0.05;
  uint32_t sampling_initial_nums = 100, sampling_max_iterations = 50;
  double sampling_convergence_threshold_1 = 1e-5, sampling_convergence_threshold_2 = 1e-5;
};

std::string package_root() { /* CURVIS_HOME, else the directory two levels above the executable */
  if (const char *h = std::getenv("CURVIS_HOME")) return h;
  char buf[4096];
  ssize_t n = ::readlink("/proc/self/exe", buf, sizeof buf - 1);
  if (n <= 0) return ".";
  buf[n] = 0;
  std::string p(buf);
  for (int k = 0; k < 2; ++k) {
    const size_t s = p.find_last_of('/');
    if (s == std::string::npos) return ".";
    p.resize(s);
  }
  return p; /* .../curvis_amd */
}
std::string resolve_path(const std::string &p) { /* src/filepaths.rs:42-47: relative paths are package-relative */
  if (!p.empty() && p[0] == '/') return p;
  if (path_exists(p)) return p;
  const std::string root = package_root();
  for (const std::string &cand : {root + "/" + p, root + "/data/" + p})
    if (path_exists(cand)) return cand;
  return root + "/" + p;
}
bool has_extension(const std::string &p, const char *ext) {
  const size_t d = p.find_last_of('.');
  return d != std::string::npos && p.substr(d + 1) == ext;
}

bool load_table(const std::string &file, TomlTable &t, std::string &err) {
  if (!has_extension(file, "toml")) {
    err = "The file \"" + file + "\" is not a toml file.";
    return false;
  }
  std::string text;
  if (!read_text(file, text)) {
    err = "Could not read file \"" + file + "\"";
    return false;
  }
  return parse_toml(text, t, err);
}

bool from_toml(const std::string &file, VideoSettings &s, std::string &err) {
  TomlTable t;
  return load_table(file, t, err) && get_str(t, "video_name", s.video_name, err) &&
         get_f64(t, "frame_rate", s.frame_rate, err) && get_str(t, "filepath_to_camera_path", s.filepath_to_camera_path, err);
}
bool from_toml(const std::string &file, ImageSettings &s, std::string &err) {
  TomlTable t;
  return load_table(file, t, err) && get_str(t, "image_name", s.image_name, err) && get_f64(t, "t", s.t, err) &&
         get_f64(t, "l", s.l, err) && get_f64(t, "theta", s.theta, err) && get_f64(t, "phi", s.phi, err) &&
         get_f64(t, "forward_x", s.forward_x, err) && get_f64(t, "forward_y", s.forward_y, err) &&
         get_f64(t, "forward_z", s.forward_z, err) && get_f64(t, "up_x", s.up_x, err) &&
         get_f64(t, "up_y", s.up_y, err) && get_f64(t, "up_z", s.up_z, err);
}
bool from_toml(const std::string &file, CameraSettings &s, std::string &err) {
  TomlTable t;
  return load_table(file, t, err) && get_u32(t, "resolution_x", s.resolution_x, err) &&
         get_u32(t, "resolution_y", s.resolution_y, err) && get_f64(t, "diagonal", s.diagonal, err) &&
         get_f64(t, "focal_length", s.focal_length, err);
}
bool from_toml(const std::string &file, SimulationSettings &s, std::string &err) {
  TomlTable t;
  return load_table(file, t, err) && get_f64(t, "escape_radius", s.escape_radius, err) &&
         get_u32(t, "ray_integration_max_itarations", s.ray_integration_max_itarations, err) &&
         get_f64(t, "ray_integration_step", s.ray_integration_step, err) &&
         get_u32(t, "sampling_initial_nums", s.sampling_initial_nums, err) &&
         get_u32(t, "sampling_max_iterations", s.sampling_max_iterations, err) &&
         get_f64(t, "sampling_convergence_threshold_1", s.sampling_convergence_threshold_1, err) &&
         get_f64(t, "sampling_convergence_threshold_2", s.sampling_convergence_threshold_2, err);
}
/* metric file: tried as Interstellar (m, a, rho) first, then Ellis (rho) -- src/cli.rs:233-261 */
bool metric_from_toml(const std::string &file, curvis_metric &m, std::string &err) {
  TomlTable t;
  if (!load_table(file, t, err)) {
    err = "Could not read the metric configuration file.";
    return false;
  }
  std::string e;
  double mm, aa, rho;
  if (get_f64(t, "m", mm, e) && get_f64(t, "a", aa, e) && get_f64(t, "rho", rho, e)) {
    m.kind = CURVIS_METRIC_INTERSTELLAR;
    m.m = mm;
    m.a = aa;
    m.rho = rho;
    return true;
  }
  if (get_f64(t, "rho", rho, e)) {
    m.kind = CURVIS_METRIC_ELLIS;
    m.rho = rho;
    m.m = m.a = 0.0;
    return true;
  }
  err = "Could not read the metric configuration file.";
  return false;
}

bool validate(const CameraSettings &c, std::string &err) { /* src/settings.rs:98-124 */
  if (c.resolution_x == 0) return err = "The resolution in the x direction must be larger than zero.", false;
  if (c.resolution_y == 0) return err = "The resolution in the y direction must be larger than zero.", false;
  if (c.diagonal <= 0.0) return err = "The diagonal of the camera must be larger than zero.", false;
  if (c.focal_length <= 0.0) return err = "The focal length of the camera must be larger than zero.", false;
  return true;
}
bool validate(const SimulationSettings &s, std::string &err) { /* src/settings.rs:137-174 */
  if (s.escape_radius <= 0.0) return err = "The escape radius must be larger than zero.", false;
  if (s.ray_integration_max_itarations == 0)
    return err = "The maximum number of iterations for the ray integration must be larger than zero.", false;
  if (s.ray_integration_step <= 0.0) return err = "The step for the ray integration must be larger than zero.", false;
  if (s.sampling_initial_nums <= 1) return err = "The initial number of samples must be larger than two.", false;
  if (s.sampling_max_iterations == 0)
    return err = "The maximum number of iterations for the sampling must be larger than zero.", false;
  if (s.sampling_convergence_threshold_1 <= 0.0)
    return err = "The first convergence threshold for the sampling must be larger than zero.", false;
  if (s.sampling_convergence_threshold_2 <= 0.0)
    return err = "The second convergence threshold for the sampling must be larger than zero.", false;
  return true;
}

/* ------------------------------------------------------------------ camera path (src/csv.rs, src/interpolation.rs) */
struct CameraPath {
  std::vector<double> pos, fwd, up; /* n*4, n*3, n*3 */
  size_t n = 0;
};
bool load_path(const std::string &file, CameraPath &p, std::string &err) {
  std::string text;
  if (!read_text(file, text)) return err = "Could not open file", false;
  size_t start = 0, index = 0;
  while (start <= text.size()) {
    size_t end = text.find('\n', start);
    const bool last = end == std::string::npos;
    if (last) end = text.size();
    std::string line = text.substr(start, end - start);
    start = end + 1;
    if (last && line.empty()) break;
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (index++ == 0) {
      if (last) break;
      continue; /* header */
    }
    double v[10];
    size_t a = 0;
    int k = 0;
    for (size_t i = 0; i <= line.size() && k < 10; ++i) {
      if (i == line.size() || line[i] == ',') {
        const std::string tok = line.substr(a, i - a);
        char *endp = nullptr;
        if (tok.empty() || std::isspace((unsigned char)tok.front()) || std::isspace((unsigned char)tok.back()))
          return err = "Could not parse float", false;
        v[k] = std::strtod(tok.c_str(), &endp);
        if (!endp || *endp) return err = "Could not parse float", false;
        ++k;
        a = i + 1;
      }
    }
    if (k < 10) return err = "Could not read all ten columns of the camera path", false;
    p.pos.insert(p.pos.end(), v, v + 4);
    p.fwd.insert(p.fwd.end(), v + 4, v + 7);
    p.up.insert(p.up.end(), v + 7, v + 10);
    p.n++;
    if (last) break;
  }
  return p.n > 0 ? true : (err = "empty camera path", false);
}
/* 0 ok, 1 = panic "time outside range", 2 = index out of bounds (the off-by-one) */
int path_camera(const CameraPath &p, double t, double pos[4], double fwd[3], double up[3]) {
  const double min_time = p.pos[0], max_time = p.pos[4 * (p.n - 1)];
  if (t < min_time || t > max_time) return 1;
  double t1 = min_time, t2 = max_time;
  size_t i = 0;
  while (t > p.pos[4 * i]) {
    t1 = p.pos[4 * i];
    t2 = p.pos[4 * (i + 1)];
    i += 1;
  }
  const double frac = (t - t1) / (t2 - t1);
  const size_t i1 = i, i2 = i + 1;
  if (i2 >= p.n) return 2;
  if (!(frac >= 0.0 && frac <= 1.0)) return 1;
  for (int k = 0; k < 4; ++k) pos[k] = p.pos[4 * i1 + k] + frac * (p.pos[4 * i2 + k] - p.pos[4 * i1 + k]);
  for (int k = 0; k < 3; ++k) {
    fwd[k] = p.fwd[3 * i1 + k] + frac * (p.fwd[3 * i2 + k] - p.fwd[3 * i1 + k]);
    up[k] = p.up[3 * i1 + k] + frac * (p.up[3 * i2 + k] - p.up[3 * i1 + k]);
  }
  return 0;
}

/* ------------------------------------------------------------------ command line */
struct Args {
  std::string sub, bg1, bg2, out, image_toml, video_toml, metric_toml, camera_toml, sim_toml, mode = "efficient", stats,
      sky_broadcast = "rccl";
  bool sky_broadcast_explicit = false, resume = false;
  int contexts = 0; /* 0 = automatic (2 in --mode efficient, else 1); video: contexts (= host worker threads) per device: while one waits on the host-side sampler or the D2H copy another's kernels run */
  int devices = 1, device = 0, batch = 8, writers = 0; /* writers 0 = automatic: a quarter of the host's threads, 4..64 */
  int png_level = -1; /* -1 = the fast PNG writer (png_io.h; the reference's image crate also saves with its fast setting), 0..9 = zlib */
  int encode_bench = 0; /* video, diagnostics: every rendered frame is encoded this many extra times into a scratch file */
  std::string gpu_png = "auto"; /* video: PNG front end on the device (curvis_ctx_deflate_frames): auto = with the fast writer, on, off */
};
[[noreturn]] void die(const std::string &msg, int code = 1) {
  std::fprintf(stderr, "%s\n", msg.c_str());
  std::exit(code);
}
void usage() {
  std::printf(
      "Usage: curvis <COMMAND>\n\nCommands:\n  image   renders a single image frame\n  video   renders a video\n"
      "  custom  runs the custom script\n\n"
      "curvis image <IMAGE FILE 1> <IMAGE FILE 2> [OUTPUT FOLDER] [-i|--image-settings <TOML FILE>]\n"
      "curvis video <IMAGE FILE 1> <IMAGE FILE 2> [OUTPUT FOLDER] [-v|--video-settings <TOML FILE>]\n"
      "  common: [-m|--metric-settings <TOML FILE>] [-c|--camera-settings <TOML FILE>] [-s|--simulation-settings <TOML FILE>]\n"
      "  extensions: [--mode efficient|brute|direct] [--device N] [--devices N] [--batch B] [--stats FILE]\n"
      "              [--sky-broadcast rccl|upload] [--writers T] [--resume] [--png-level -1..9] [--gpu-png auto|on|off]\n"
      "              [--contexts-per-device C]\n");
}
Args parse_args(int argc, char **argv) {
  Args a;
  if (argc < 2) {
    usage();
    die("Subcommand not found");
  }
  a.sub = argv[1];
  if (a.sub == "selftest-png") return a;
  if (a.sub == "-h" || a.sub == "--help") {
    usage();
    std::exit(0);
  }
  std::vector<std::string> pos;
  for (int i = 2; i < argc; ++i) {
    std::string s = argv[i], val;
    auto take = [&](std::string &dst) {
      const size_t eq = s.find('=');
      if (s.rfind("--", 0) == 0 && eq != std::string::npos)
        dst = s.substr(eq + 1);
      else if (i + 1 < argc)
        dst = argv[++i];
      else
        die("error: a value is required for '" + s + "' but none was supplied", 2);
    };
    const std::string key = s.substr(0, s.find('='));
    if (key == "-i" || key == "--image-settings") take(a.image_toml);
    else if (key == "-v" || key == "--video-settings") take(a.video_toml);
    else if (key == "-m" || key == "--metric-settings") take(a.metric_toml);
    else if (key == "-c" || key == "--camera-settings") take(a.camera_toml);
    else if (key == "-s" || key == "--simulation-settings") take(a.sim_toml);
    else if (key == "--mode") take(a.mode);
    else if (key == "--stats") take(a.stats);
    else if (key == "--sky-broadcast") { take(a.sky_broadcast); a.sky_broadcast_explicit = true; }
    else if (key == "--resume") a.resume = true;
    else if (key == "--devices") { take(val); a.devices = std::atoi(val.c_str()); }
    else if (key == "--contexts-per-device") { take(val); a.contexts = std::max(0, std::min(8, std::atoi(val.c_str()))); }
    else if (key == "--device") { take(val); a.device = std::atoi(val.c_str()); }
    else if (key == "--batch") { take(val); a.batch = std::atoi(val.c_str()); }
    else if (key == "--writers") { take(val); a.writers = std::atoi(val.c_str()); }
    else if (key == "--png-level") { take(val); a.png_level = std::max(-1, std::min(9, std::atoi(val.c_str()))); }
    else if (key == "--encode-bench") { take(val); a.encode_bench = std::max(0, std::atoi(val.c_str())); }
    else if (key == "--gpu-png") take(a.gpu_png);
    else if (key == "-h" || key == "--help") { usage(); std::exit(0); }
    else if (!s.empty() && s[0] == '-') die("error: unexpected argument '" + s + "' found", 2);
    else pos.push_back(s);
  }
  if (a.sub == "image" || a.sub == "video") {
    if (pos.size() < 2) die("error: the following required arguments were not provided:\n  <IMAGE FILE 1>\n  <IMAGE FILE 2>", 2);
    if (pos.size() > 3) die("error: unexpected argument '" + pos[3] + "' found", 2);
    a.bg1 = pos[0];
    a.bg2 = pos[1];
    if (pos.size() == 3) a.out = pos[2];
    if (a.sub == "image" && !a.video_toml.empty()) die("error: unexpected argument '-v' found", 2);
    if (a.sub == "video" && !a.image_toml.empty()) die("error: unexpected argument '-i' found", 2);
  }
  if (a.mode != "efficient" && a.mode != "brute" && a.mode != "direct") die("error: --mode must be efficient, brute or direct", 2);
  if (a.sky_broadcast != "rccl" && a.sky_broadcast != "upload") die("error: --sky-broadcast must be rccl or upload", 2);
  if (a.gpu_png != "auto" && a.gpu_png != "on" && a.gpu_png != "off") die("error: --gpu-png must be auto, on or off", 2);
  if (a.devices < 1) a.devices = 1;
  /* --mode efficient spends about half of a frame's render call on the host (the adaptive sampler between its launches of
   * lone waves): two contexts per GPU overlap that with each other's kernels (measured: 1830 -> 2500 1080p frames/s on one
   * MI355X, three or four contexts are slower again).  The per-pixel modes keep the GPU busy by themselves. */
  if (a.contexts == 0) a.contexts = a.mode == "efficient" ? 2 : 1;
  if (a.batch < 1) a.batch = 1;
  if (a.writers < 1) { /* encoding a 1080p frame costs 5-25 ms of a host thread (zlib: 20-230): the GPU renders one in 0.4-10 ms */
    unsigned hw = std::thread::hardware_concurrency();
    /* a container may see every CPU of the host behind a much smaller cgroup quota ("1600000 100000" = 16 CPUs) */
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
      double quota = 0.0, period = 0.0;
      if (std::fscanf(f, "%lf %lf", &quota, &period) == 2 && quota > 0.0 && period > 0.0)
        hw = std::min(hw, (unsigned)(quota / period + 0.5) * 4u); /* the quota itself: this many writers at most */
      std::fclose(f);
    }
    a.writers = (int)std::min(64u, std::max(4u, hw / 4u));
  }
  return a;
}

struct Common {
  curvis_metric metric{CURVIS_METRIC_ELLIS, 0, 1.0, 0.0, 0.0}; /* default: Ellis rho = 1 (ellis_metric_settings.toml) */
  CameraSettings cam;
  SimulationSettings sim;
  pngio::Image sky1, sky2;
  std::string out;
};

void load_common(const Args &a, Common &c, const char *what) {
  std::string err;
  auto need = [&](const std::string &f, const char *label) {
    if (!path_exists(f)) die(std::string("Error with ") + label + ": File \"" + f + "\" not found.");
  };
  need(a.bg1, "background image 1");
  need(a.bg2, "background image 2");
  if (a.out.empty()) {
    char cwd[4096];
    if (!::getcwd(cwd, sizeof cwd)) die("Error with output folder: Could not get current working directory.");
    c.out = cwd;
  } else {
    if (!path_exists(a.out)) die("Error with output folder: File \"" + a.out + "\" not found.");
    if (!is_dir(a.out)) die("Error with output folder: \"" + a.out + "\" is not a folder.");
    c.out = a.out;
  }
  if (!a.metric_toml.empty()) {
    need(a.metric_toml, "metric settings");
    if (!metric_from_toml(a.metric_toml, c.metric, err)) die("Error with metric settings: " + err);
  }
  if (!a.camera_toml.empty()) {
    need(a.camera_toml, "camera settings");
    if (!from_toml(a.camera_toml, c.cam, err)) die("Error with camera settings: " + err);
  }
  if (!a.sim_toml.empty()) {
    need(a.sim_toml, "simulation settings");
    if (!from_toml(a.sim_toml, c.sim, err)) die("Error with simulation settings: " + err);
  }
  /* instantiate_metric (src/main.rs:114-132): constructor panics */
  if (curvis_metric_validate(&c.metric) != CURVIS_OK)
    die(std::string("Error in rendering ") + what + ": metric parameters must be positive (src/metrics.rs:409-456)", 101);
  if (!validate(c.cam, err) || !validate(c.sim, err)) die(std::string("Error in rendering ") + what + ": " + err);
  /* both backgrounds are decoded at the same time (an 8192x4096 PNG takes the better part of a second to inflate and
   * unfilter); errors are reported in the reference's order, image 1 first */
  std::string err2;
  bool ok2 = false;
  std::thread second([&] { ok2 = jpegio::load_image(a.bg2, c.sky2, err2); });
  const bool ok1 = jpegio::load_image(a.bg1, c.sky1, err);
  second.join();
  if (!ok1) die(std::string("Error in rendering ") + what + ": background image 1: " + err);
  if (!ok2) die(std::string("Error in rendering ") + what + ": background image 2: " + err2);
}

void check(int rc, curvis_ctx *ctx, const char *what) {
  if (rc == CURVIS_OK) return;
  const char *msg = curvis_last_error(ctx);
  die(std::string("Error in rendering ") + what + ": " + (msg ? msg : "") + " (code " + std::to_string(rc) + ")",
      (rc == CURVIS_E_CAMERA_OUTSIDE || rc == CURVIS_E_PARALLEL || rc == CURVIS_E_SAMPLING) ? 101 : 1);
}

curvis_ctx *make_ctx_bare(int device, const char *what) {
  curvis_ctx *ctx = nullptr;
  int rc = curvis_ctx_create(device, &ctx);
  if (rc != CURVIS_OK) die(std::string("Error in rendering ") + what + ": " + curvis_last_error(nullptr));
  return ctx;
}
void upload_skies(curvis_ctx *ctx, const Common &c, const char *what) {
  check(curvis_ctx_set_sky(ctx, 0, c.sky1.rgba.data(), c.sky1.w, c.sky1.h), ctx, what);
  check(curvis_ctx_set_sky(ctx, 1, c.sky2.rgba.data(), c.sky2.w, c.sky2.h), ctx, what);
}
curvis_ctx *make_ctx(int device, const Common &c, const char *what) {
  curvis_ctx *ctx = make_ctx_bare(device, what);
  upload_skies(ctx, c, what);
  return ctx;
}

/* per-frame statistics of the last render_frames call of this thread in "direct" mode (one render call per frame there;
 * the batch calls of the other modes keep theirs inside the context: curvis_ctx_frame_stats) */
thread_local std::vector<curvis_stats> g_direct_frame_stats;

int render_frames(curvis_ctx *ctx, const Args &a, const Common &c, const curvis_camera *cams, uint32_t n, double thr2,
                  uint8_t *rgb, curvis_stats *st) {
  if (a.mode == "brute")
    return curvis_render_brute_batch(ctx, &c.metric, cams, n, c.sim.ray_integration_max_itarations, c.sim.escape_radius,
                                     c.sim.ray_integration_step, rgb, st);
  if (a.mode == "direct") { /* extension: compute_escape_angle for every pixel, no sampling / interpolation; frame by frame */
    curvis_stats tot;
    std::memset(&tot, 0, sizeof tot);
    const size_t fbytes = (size_t)cams[0].res_x * cams[0].res_y * 3;
    g_direct_frame_stats.clear();
    for (uint32_t f = 0; f < n; ++f) {
      curvis_stats one;
      const int rc = curvis_render_direct(ctx, &c.metric, cams + f, c.sim.ray_integration_max_itarations, c.sim.escape_radius,
                                          c.sim.ray_integration_step, rgb + (size_t)f * fbytes, &one);
      if (rc != CURVIS_OK) return rc;
      g_direct_frame_stats.push_back(one);
      tot.rays += one.rays;
      tot.steps += one.steps;
      tot.n_pos += one.n_pos;
      tot.n_neg += one.n_neg;
      tot.n_none += one.n_none;
      tot.n_oob += one.n_oob;
      tot.kernel_ms += one.kernel_ms;
      tot.integrate_ms += one.integrate_ms;
      tot.total_ms += one.total_ms;
    }
    if (st) *st = tot;
    return CURVIS_OK;
  }
  /* src/main.rs:46-47 / :106-107: alphas_num AND max_iterations_sampling both take sampling_initial_nums */
  return curvis_render_efficient_batch(ctx, &c.metric, cams, n, c.sim.ray_integration_max_itarations, c.sim.escape_radius,
                                       c.sim.ray_integration_step, c.sim.sampling_initial_nums, c.sim.sampling_initial_nums,
                                       c.sim.sampling_convergence_threshold_1, thr2, rgb, st);
}

int image_main(const Args &a) {
  std::printf("Image rendering\n");
  Common c;
  ImageSettings is;
  std::string err;
  if (!a.image_toml.empty()) {
    if (!path_exists(a.image_toml)) die("Error with image settings: File \"" + a.image_toml + "\" not found.");
    if (!from_toml(a.image_toml, is, err)) die("Error with image settings: " + err);
  }
  load_common(a, c, "image");
  if (is.image_name.empty()) die("Error in rendering image: Image name cannot be an empty string.");
  const double pos[4] = {is.t, is.l, is.theta, is.phi}, fwd[3] = {is.forward_x, is.forward_y, is.forward_z},
               up[3] = {is.up_x, is.up_y, is.up_z};
  curvis_camera cam;
  int rc = curvis_camera_init(&cam, pos, fwd, up, c.cam.focal_length, c.cam.diagonal, c.cam.resolution_x, c.cam.resolution_y);
  if (rc == CURVIS_E_PARALLEL) die("Error in rendering image: Forward and up vectors must not be parallel", 101);
  if (rc != CURVIS_OK) die("Error in rendering image: invalid camera settings");
  curvis_ctx *ctx = make_ctx(a.device, c, "image");
  if (!path_exists(c.out) && ::mkdir(c.out.c_str(), 0777) != 0)
    die("Error in rendering image: Could not create video output folder \"" + c.out + "\"");
  std::vector<uint8_t> rgb((size_t)cam.res_x * cam.res_y * 3);
  curvis_stats st;
  std::memset(&st, 0, sizeof st);
  if (a.mode == "brute" && a.devices > 1) {
    /* --mode brute --devices N: the rows of the ONE frame are split over N GPUs (rays are independent,
     * src/systems.rs:316-326); one host thread + one context per GPU, the bands land in place in `rgb` */
    std::vector<curvis_ctx *> ctxs((size_t)a.devices, nullptr);
    std::vector<curvis_stats> sts((size_t)a.devices);
    std::vector<int> rcs((size_t)a.devices, CURVIS_OK);
    ctxs[0] = ctx;
    const bool share = std::getenv("CURVIS_TEST_SHARE_DEVICE") != nullptr; /* test hook: all bands on one GPU */
    for (int r = 1; r < a.devices; ++r) ctxs[(size_t)r] = make_ctx(share ? a.device : a.device + r, c, "image");
    const uint32_t H = cam.res_y, base = H / (uint32_t)a.devices, extra = H % (uint32_t)a.devices;
    std::vector<std::thread> th;
    for (int r = 0; r < a.devices; ++r)
      th.emplace_back([&, r] {
        const uint32_t begin = (uint32_t)r * base + std::min<uint32_t>((uint32_t)r, extra);
        const uint32_t count = base + ((uint32_t)r < extra ? 1u : 0u);
        std::memset(&sts[(size_t)r], 0, sizeof(curvis_stats));
        if (count == 0) return;
        rcs[(size_t)r] = curvis_render_brute_rows(ctxs[(size_t)r], &c.metric, &cam, begin, count,
                                                  c.sim.ray_integration_max_itarations, c.sim.escape_radius,
                                                  c.sim.ray_integration_step, rgb.data() + (size_t)begin * cam.res_x * 3,
                                                  &sts[(size_t)r]);
      });
    for (auto &t : th) t.join();
    for (int r = 0; r < a.devices; ++r) {
      check(rcs[(size_t)r], ctxs[(size_t)r], "image");
      st.rays += sts[(size_t)r].rays;
      st.steps += sts[(size_t)r].steps;
      st.n_pos += sts[(size_t)r].n_pos;
      st.n_neg += sts[(size_t)r].n_neg;
      st.n_none += sts[(size_t)r].n_none;
      st.n_oob += sts[(size_t)r].n_oob;
      st.kernel_ms = std::max(st.kernel_ms, sts[(size_t)r].kernel_ms);
      if (r > 0) curvis_ctx_destroy(ctxs[(size_t)r]);
    }
  } else {
    check(render_frames(ctx, a, c, &cam, 1, c.sim.sampling_convergence_threshold_2, rgb.data(), &st), ctx, "image");
  }
  /* PathBuf::join(image_name).with_extension("png") (src/rendering.rs:108): an existing extension is REPLACED */
  const std::string file = (std::filesystem::path(c.out) / std::filesystem::path(is.image_name).replace_extension("png")).string();
  if (!pngio::save_rgb8(file, rgb.data(), cam.res_x, cam.res_y, err, a.png_level))
    die("Error in rendering image: Could not save image frame \"" + file + "\" due to error: " + err);
  if (!a.stats.empty()) {
    FILE *f = std::fopen(a.stats.c_str(), "w");
    if (f) {
      std::fprintf(f, "{\"frame\": 0, \"mode\": \"%s\", \"rays\": %llu, \"steps\": %llu, \"n_pos\": %llu, \"n_neg\": %llu, \"n_none\": %llu, \"n_oob\": %llu, \"kernel_ms\": %.4f, \"mray_steps_per_s\": %.1f}\n",
                   a.mode.c_str(), (unsigned long long)st.rays, (unsigned long long)st.steps, (unsigned long long)st.n_pos,
                   (unsigned long long)st.n_neg, (unsigned long long)st.n_none, (unsigned long long)st.n_oob, st.kernel_ms,
                   st.kernel_ms > 0.0 ? (double)st.steps / st.kernel_ms / 1e3 : 0.0);
      std::fclose(f);
    }
  }
  curvis_ctx_destroy(ctx);
  return 0;
}

/* std::fs::remove_dir_all (src/rendering.rs:278): no shell involved, the path is never interpreted */
int rm_rf(const std::string &dir) {
  std::error_code ec;
  std::filesystem::remove_all(std::filesystem::path(dir), ec);
  return ec ? 1 : 0;
}

/* frame writers: PNG encoding (zlib) costs more host time per frame than the GPU needs to render it, so
 * frames are compressed and written by a small pool of host threads while the GPU renders the next batch. */
class WriterPool {
 public:
  explicit WriterPool(int n) {
    for (int i = 0; i < n; ++i) th_.emplace_back([this] { run(); });
  }
  ~WriterPool() { finish(); }
  void submit(std::function<void()> job) {
    std::unique_lock<std::mutex> g(mu_);
    cv_space_.wait(g, [this] { return q_.size() < 64; }); /* bound the frames held in host memory */
    q_.push_back(std::move(job));
    cv_work_.notify_one();
  }
  void finish() {
    {
      std::lock_guard<std::mutex> g(mu_);
      if (done_) return;
      done_ = true;
    }
    cv_work_.notify_all();
    for (auto &t : th_) t.join();
  }

 private:
  void run() {
    for (;;) {
      std::function<void()> job;
      {
        std::unique_lock<std::mutex> g(mu_);
        cv_work_.wait(g, [this] { return done_ || !q_.empty(); });
        if (q_.empty()) return;
        job = std::move(q_.front());
        q_.pop_front();
        cv_space_.notify_one();
      }
      job();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_work_, cv_space_;
  std::deque<std::function<void()>> q_;
  std::vector<std::thread> th_;
  bool done_ = false;
};

/* Page-locked batch buffers of the video workers: the render call copies a batch of frames into one of them by DMA, the
 * writer threads encode straight out of it, and the last frame written gives it back -- no pageable bounce copy (2 GB/s)
 * and no per-frame memcpy between the render call and the encoder.  A worker that finds the pool empty waits: that is
 * the back-pressure of a host that cannot keep up. */
class PinnedPool {
 public:
  PinnedPool(size_t bytes_each, int n) : bytes_(bytes_each) {
    for (int i = 0; i < n; ++i) {
      void *p = nullptr;
      if (curvis_host_alloc(bytes_each, &p) != CURVIS_OK || !p) break;
      free_.push_back((uint8_t *)p);
      all_.push_back((uint8_t *)p);
    }
  }
  ~PinnedPool() {
    for (uint8_t *p : all_) curvis_host_free(p);
  }
  size_t buffers() const { return all_.size(); }
  /* a buffer that returns to the pool when the last holder lets go of it */
  std::shared_ptr<uint8_t> take(double *waited_s) {
    std::unique_lock<std::mutex> g(mu_);
    const double t0 = pngio::now_s();
    cv_.wait(g, [this] { return !free_.empty(); });
    if (waited_s) *waited_s += pngio::now_s() - t0;
    uint8_t *p = free_.back();
    free_.pop_back();
    return std::shared_ptr<uint8_t>(p, [this](uint8_t *q) {
      {
        std::lock_guard<std::mutex> g2(mu_);
        free_.push_back(q);
      }
      cv_.notify_one();
    });
  }

 private:
  size_t bytes_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<uint8_t *> free_, all_;
};

int video_main(const Args &a) {
  std::printf("Video rendering\n");
  Common c;
  VideoSettings vs;
  std::string err;
  if (!a.video_toml.empty()) {
    if (!path_exists(a.video_toml)) die("Error with video settings: File \"" + a.video_toml + "\" not found.");
    if (!from_toml(a.video_toml, vs, err)) die("Error with video settings: " + err);
  }
  load_common(a, c, "video");
  vs.filepath_to_camera_path = resolve_path(vs.filepath_to_camera_path); /* normalize() */
  if (vs.video_name.empty()) die("Error in rendering video: Video name cannot be an empty string.");
  if (!has_extension(vs.filepath_to_camera_path, "csv"))
    die("Error in rendering video: The camera path \"" + vs.filepath_to_camera_path + "\" is not a csv file.");
  if (!path_exists(vs.filepath_to_camera_path))
    die("Error in rendering video: The camera path \"" + vs.filepath_to_camera_path + "\" does not exist.");
  CameraPath path;
  if (!load_path(vs.filepath_to_camera_path, path, err)) die("Error in rendering video: " + err, 101);
  /* times_of_frames (src/rendering.rs:224-238) */
  std::vector<double> times;
  {
    const double min_time = path.pos[0], max_time = path.pos[4 * (path.n - 1)], dt = 1.0 / vs.frame_rate;
    for (double t = min_time; t < max_time; t += dt) times.push_back(t);
  }
  if (!path_exists(c.out) && ::mkdir(c.out.c_str(), 0777) != 0)
    die("Error in rendering video: Could not create video output folder \"" + c.out + "\"");
  const std::string tmp = c.out + "/tmp";
  if (a.resume) { /* opt-in: keep the frames a previous (interrupted) run has written */
    if (!path_exists(tmp) && ::mkdir(tmp.c_str(), 0777) != 0)
      die("Error in rendering video: Could not create tmp output folder \"" + tmp + "\"");
  } else {
    if (path_exists(tmp) && rm_rf(tmp) != 0)
      die("Error in rendering video: Could not remove pre-existing tmp folder \"" + tmp + "\"");
    if (::mkdir(tmp.c_str(), 0777) != 0) die("Error in rendering video: Could not create tmp output folder \"" + tmp + "\"");
  }
  std::printf("Rendering %zu frames...\n", times.size());

  /* cameras of all frames; the reference panics when it reaches the broken last segment, after having
   * written the frames before it: frames up to the first failing one are rendered, then exit 101. */
  std::vector<curvis_camera> cams;
  std::string panic_msg;
  for (size_t k = 0; k < times.size(); ++k) {
    double pos[4], fwd[3], up[3];
    const int prc = path_camera(path, times[k], pos, fwd, up);
    if (prc != 0) {
      panic_msg = prc == 2 ? "index out of bounds in the camera-path interpolation (src/interpolation.rs:76-90)"
                           : "Interpolation time outside the camera path";
      break;
    }
    curvis_camera cam;
    const int rc = curvis_camera_init(&cam, pos, fwd, up, c.cam.focal_length, c.cam.diagonal, c.cam.resolution_x, c.cam.resolution_y);
    if (rc != CURVIS_OK) {
      panic_msg = "Forward and up vectors must not be parallel";
      break;
    }
    cams.push_back(cam);
  }
  const size_t n_frames = cams.size();
  const size_t fbytes = (size_t)c.cam.resolution_x * c.cam.resolution_y * 3;
  std::mutex io_mu;
  std::atomic<int> failed{0};
  FILE *stats_f = a.stats.empty() ? nullptr : std::fopen(a.stats.c_str(), "w");
  WriterPool writers(a.writers);
  /* per-stage host profile (--stats): what the writer threads spent encoding and writing, what each device worker
   * spent inside the render call (GPU kernels + D2H of the batch), waiting for work and handing frames over */
  pngio::EncodeTimes enc_total;
  pngio::EncodeTimes bench_total; /* --encode-bench: the extra encodes, kept apart */
  struct DeviceSummary {
    std::string pci_bus_id;
    int sclk_mhz = -1, power_w = -1;
    size_t frames = 0, batches = 0;
    double render_s = 0, kernel_ms = 0, submit_s = 0, wait_s = 0, busy_s = 0, pool_wait_s = 0;
    double png_ms = 0; /* device PNG front end: HIP-event time of its kernels */
    size_t png_frames = 0, png_fallback_frames = 0;
    double sky_s = 0, sky_bcast_s = 0; /* skies into this device's HBM: all of it / the curvis_ctx_bcast_skies call alone */
    unsigned long long steps = 0;
  };
  /* workers = devices x contexts-per-device; worker r drives device r / contexts with a context of its own (`--mode efficient`
   * spends half of a frame's render call on the host -- the adaptive sampler between its launches --, so a second context
   * on the same GPU fills the gaps) */
  const int n_workers = a.devices * a.contexts;
  auto device_of = [&](int rank) { return a.device + rank / a.contexts; };
  std::vector<DeviceSummary> dev_sum((size_t)n_workers);
  std::vector<std::unique_ptr<PinnedPool>> pools((size_t)n_workers); /* destroyed after writers.finish() below */
  const double t_video0 = pngio::now_s();
  /* sky distribution: rank 0 uploads the two textures once; with --sky-broadcast rccl (default for
   * --devices > 1) the other GPUs receive them with ncclBroadcast over xGMI (curvis_ctx_bcast_skies),
   * otherwise every GPU uploads from host memory. */
  std::vector<ncclComm_t> comms;
  const bool share_device = std::getenv("CURVIS_TEST_SHARE_DEVICE") != nullptr; /* test hook: every worker on GPU a.device */
  bool use_rccl = a.sky_broadcast == "rccl" && !share_device && (a.devices > 1 || std::getenv("CURVIS_FORCE_RCCL"));
  if (use_rccl) {
    std::vector<int> devs;
    for (int r = 0; r < a.devices; ++r) devs.push_back(a.device + r);
    comms.resize(a.devices);
    /* one node, one process: RCCL's bootstrap needs no network.  Left to itself it picks the first "real" interface, and on
     * hosts where that one is slow or unroutable communicator set-up was seen to take 6 s (lo: 2.7 s) up to ~80 s */
    ::setenv("NCCL_SOCKET_IFNAME", "lo", 0 /* a value the user has set stays */);
    const ncclResult_t nrc = ncclCommInitAll(comms.data(), a.devices, devs.data());
    if (nrc != ncclSuccess) {
      /* asked for explicitly: a broken xGMI broadcast must not hide behind a silent fallback */
      if (a.sky_broadcast_explicit)
        die(std::string("Error in rendering video: --sky-broadcast rccl: ncclCommInitAll failed (") + ncclGetErrorString(nrc) + ")");
      std::fprintf(stderr, "warning: ncclCommInitAll failed (%s), uploading the skies to every device instead\n", ncclGetErrorString(nrc));
      comms.clear();
      use_rccl = false;
    }
  }
  /* Work list: frame k belongs to device k mod N (src/rendering.rs:291-316 has no cross-frame state), in batches of
   * --batch frames per launch.  With --resume the frames already on disk are dropped first.  A batch whose render
   * call fails goes to a shared retry queue and is taken by a DIFFERENT device (by the same one when there is only
   * one); after max(2, N) failed attempts the run fails. */
  struct Batch {
    std::vector<size_t> frames;
    int attempts = 0, last_device = -1;
  };
  std::vector<std::deque<Batch>> own((size_t)n_workers);
  size_t n_skipped = 0, n_batches = 0;
  for (int r = 0; r < n_workers; ++r) {
    Batch cur;
    for (size_t k = (size_t)r; k < n_frames; k += (size_t)n_workers) {
      if (a.resume) {
        struct stat sb;
        const std::string file = tmp + "/frame_" + std::to_string(k) + ".png";
        if (::stat(file.c_str(), &sb) == 0 && sb.st_size > 0) {
          ++n_skipped;
          continue;
        }
      }
      cur.frames.push_back(k);
      if (cur.frames.size() == (size_t)a.batch) {
        own[(size_t)r].push_back(cur);
        cur.frames.clear();
      }
    }
    if (!cur.frames.empty()) own[(size_t)r].push_back(cur);
    n_batches += own[(size_t)r].size();
  }
  if (a.resume) std::printf("Resuming: %zu of %zu frames already present in \"%s\"\n", n_skipped, n_frames, tmp.c_str());
  /* PNG front end on the device: with the fast writer (the default) in the modes whose frames of a batch sit together in the
   * context's framebuffer; --encode-bench measures the HOST encoder and therefore keeps it */
  const bool gpu_png = a.gpu_png == "on" ? (a.mode != "direct")
                       : a.gpu_png == "auto" ? (a.png_level < 0 && a.mode != "direct" && a.encode_bench == 0) : false;
  std::mutex q_mu;
  std::condition_variable q_cv;
  std::deque<Batch> retry;
  size_t batches_done = 0;
  const int max_attempts = std::max(2, n_workers);
  /* fault injection for the tests: "rank:n" makes the n-th render call of that worker fail once */
  int fail_rank = -1, fail_call = -1;
  if (const char *fi = std::getenv("CURVIS_TEST_FAIL_BATCH")) std::sscanf(fi, "%d:%d", &fail_rank, &fail_call);
  auto worker = [&](int rank) {
    curvis_ctx *ctx = make_ctx_bare(share_device ? a.device : device_of(rank), "video");
    DeviceSummary &ds = dev_sum[(size_t)rank];
    {
      char id[64] = {0};
      (void)curvis_ctx_device_status(ctx, id, sizeof id, nullptr, nullptr);
      ds.pci_bus_id = id;
    }
    const double t_worker0 = pngio::now_s();
    if (use_rccl && rank % a.contexts == 0) { /* one context per device takes part in the broadcast; its siblings upload */
      if (rank == 0) upload_skies(ctx, c, "video");
      const double t_b0 = pngio::now_s();
      check(curvis_ctx_bcast_skies(ctx, comms[rank / a.contexts], 0), ctx, "video");
      ds.sky_bcast_s = pngio::now_s() - t_b0;
      /* every GPU checks what arrived over xGMI against the decoded files (head, middle and tail of both textures):
       * a broken broadcast must stop the run, not colour its frames */
      const pngio::Image *sk[2] = {&c.sky1, &c.sky2};
      for (int w = 0; w < 2; ++w) {
        const size_t total = sk[w]->rgba.size(), piece = std::min<size_t>(total, (size_t)1 << 16);
        std::vector<uint8_t> got(piece);
        for (size_t off : {(size_t)0, (total - piece) / 2, total - piece}) {
          check(curvis_ctx_read_sky(ctx, w, off, piece, got.data()), ctx, "video");
          if (std::getenv("CURVIS_TEST_CORRUPT_BCAST")) got[piece / 2] ^= 0x10;  /* test hook: pretend a flipped bit */
          if (std::memcmp(got.data(), sk[w]->rgba.data() + off, piece) != 0)
            die("Error in rendering video: background " + std::to_string(w + 1) + " arrived corrupted on device " +
                std::to_string(device_of(rank)) + " after the RCCL broadcast");
        }
      }
    } else {
      upload_skies(ctx, c, "video");
    }
    ds.sky_s = pngio::now_s() - t_worker0;
    std::vector<curvis_camera> bc;
    std::vector<uint8_t> rgb_pageable; /* only if page-locked memory could not be had */
    /* one being filled, up to two with the writers.  The pool belongs to video_main's scope: writer jobs hold its
     * buffers (and its mutex, through the deleter) after this worker has returned */
    pools[(size_t)rank].reset(new PinnedPool((size_t)a.batch * fbytes, 3));
    PinnedPool &pool = *pools[(size_t)rank];
    if (pool.buffers() < 2) {
      std::lock_guard<std::mutex> gi(io_mu);
      std::fprintf(stderr, "warning: device %d: no page-locked host memory for the frame buffers, using pageable memory\n", device_of(rank));
    }
    int calls = 0;
    for (;;) {
      Batch b;
      {
        std::unique_lock<std::mutex> g(q_mu);
        for (;;) {
          if (failed || batches_done == n_batches) {
            g.unlock();
            q_cv.notify_all(); /* nobody may sleep on while the others leave */
            ds.busy_s = pngio::now_s() - t_worker0;
            curvis_ctx_destroy(ctx);
            return;
          }
          auto it = retry.begin();
          while (it != retry.end() && it->last_device == rank && n_workers > 1) ++it;
          if (it != retry.end()) {
            b = *it;
            retry.erase(it);
            break;
          }
          if (!own[(size_t)rank].empty()) {
            b = own[(size_t)rank].front();
            own[(size_t)rank].pop_front();
            break;
          }
          const double tw = pngio::now_s();
          q_cv.wait_for(g, std::chrono::milliseconds(200)); /* re-checks `failed`: a writer thread sets it without this lock */
          ds.wait_s += pngio::now_s() - tw;
        }
      }
      const size_t nb = b.frames.size();
      bc.clear();
      for (size_t j = 0; j < nb; ++j) bc.push_back(cams[b.frames[j]]);
      std::shared_ptr<uint8_t> batch_buf;
      uint8_t *rgb_ptr = nullptr;
      if (pool.buffers() >= 2) {
        batch_buf = pool.take(&ds.pool_wait_s);
        rgb_ptr = batch_buf.get();
      } else {
        rgb_pageable.resize(nb * fbytes);
        rgb_ptr = rgb_pageable.data();
      }
      curvis_stats st;
      /* src/rendering.rs:305-306: threshold_1 is passed for both thresholds */
      const double t_r0 = pngio::now_s();
      /* with the device PNG front end the pixels stay in HBM (rgb_out = NULL) and the batch buffer receives the frames' zlib
       * streams instead; should they not fit (frames that do not compress: > 1 byte per byte) the raw frames are fetched after
       * all and the host encoder takes them */
      std::vector<size_t> zoff;
      bool streams = false;
      int rc = render_frames(ctx, a, c, bc.data(), (uint32_t)nb, c.sim.sampling_convergence_threshold_1, gpu_png ? nullptr : rgb_ptr, &st);
      if (rc == CURVIS_OK && gpu_png) {
        zoff.resize(nb + 1);
        double pms = 0.0;
        /* test hook: pretend the streams do not fit (frames that do not compress), so that the fall-back below runs */
        const size_t zcap = std::getenv("CURVIS_TEST_SMALL_PNG_BUFFER") ? (size_t)64 : batch_buf ? (size_t)a.batch * fbytes : nb * fbytes;
        const int zrc = curvis_ctx_deflate_frames(ctx, c.cam.resolution_x, c.cam.resolution_y, (uint32_t)nb, rgb_ptr, zcap, zoff.data(), &pms);
        if (zrc == CURVIS_OK) {
          streams = true;
          ds.png_ms += pms;
          ds.png_frames += nb;
        } else {
          rc = curvis_ctx_download(ctx, rgb_ptr, nb * fbytes);
          ds.png_fallback_frames += nb;
        }
      }
      const double batch_call_ms = (pngio::now_s() - t_r0) * 1e3;
      ds.render_s += batch_call_ms * 1e-3;
      const bool injected = rank == fail_rank && calls == fail_call;
      if (injected) rc = CURVIS_E_HIP;
      ++calls;
      if (rc != CURVIS_OK) {
        std::lock_guard<std::mutex> g(q_mu);
        {
          std::lock_guard<std::mutex> gi(io_mu);
          std::fprintf(stderr, "warning: device %d: rendering frames %zu.. failed: %s (code %d), attempt %d of %d%s\n",
                       device_of(rank), b.frames[0], injected ? "injected test fault" : curvis_last_error(ctx), rc,
                       b.attempts + 1, max_attempts, b.attempts + 1 < max_attempts ? "; re-queued" : "");
        }
        b.attempts++;
        b.last_device = rank;
        if (b.attempts >= max_attempts) {
          std::lock_guard<std::mutex> gi(io_mu);
          std::fprintf(stderr, "Error in rendering video: frames %zu.. could not be rendered on any device\n", b.frames[0]);
          failed = 1;
        } else {
          retry.push_back(b);
        }
        q_cv.notify_all();
        continue;
      }
      /* hand the frames of this batch to the writer pool (each job owns a copy of its frame and ITS statistics:
       * the kernels keep one set of counters per frame of a launch) */
      ds.frames += nb;
      ds.batches += 1;
      ds.kernel_ms += st.kernel_ms;
      ds.steps += st.steps;
      if (ds.batches % 8 == 1) { /* clock and power while the device is under load */
        int sclk = -1, pw = -1;
        (void)curvis_ctx_device_status(ctx, nullptr, 0, &sclk, &pw);
        if (sclk > 0) ds.sclk_mhz = sclk;
        if (pw > 0) ds.power_w = pw;
      }
      const double t_s0 = pngio::now_s();
      for (size_t j = 0; j < nb; ++j) {
        const size_t k = b.frames[j];
        /* the writer job keeps the batch buffer alive and reads its frame in place; with pageable memory it owns a copy */
        std::shared_ptr<std::vector<uint8_t>> copy;
        const size_t f_off = streams ? zoff[j] : j * fbytes, f_len = streams ? zoff[j + 1] - zoff[j] : fbytes;
        if (!batch_buf) copy = std::make_shared<std::vector<uint8_t>>(rgb_ptr + f_off, rgb_ptr + f_off + f_len);
        const uint8_t *frame = batch_buf ? batch_buf.get() + f_off : copy->data();
        curvis_stats fs;
        std::memset(&fs, 0, sizeof fs);
        if (a.mode == "direct" && j < g_direct_frame_stats.size())
          fs = g_direct_frame_stats[j];
        else
          (void)curvis_ctx_frame_stats(ctx, (uint32_t)j, &fs);
        const double batch_ms = st.kernel_ms;
        writers.submit([&, k, frame, f_len, streams, batch_buf, copy, fs, nb, rank, batch_ms, batch_call_ms] {
          const std::string file = tmp + "/frame_" + std::to_string(k) + ".png";
          const std::string part = file + ".part"; /* written under another name, then renamed: --resume never sees half a file */
          std::string e;
          pngio::EncodeTimes tm, tb;
          bool ok = streams ? pngio::save_zlib_stream_rgb8(part, frame, f_len, c.cam.resolution_x, c.cam.resolution_y, e, &tm)
                            : pngio::save_rgb8(part, frame, c.cam.resolution_x, c.cam.resolution_y, e, a.png_level, &tm);
          if (ok && std::rename(part.c_str(), file.c_str()) != 0) {
            ok = false;
            e = std::strerror(errno);
          }
          for (int rep = 0; ok && rep < a.encode_bench; ++rep) { /* diagnostics: the host's encode capacity with one GPU feeding it */
            std::string e2;
            if (streams)
              (void)pngio::save_zlib_stream_rgb8(part + ".bench", frame, f_len, c.cam.resolution_x, c.cam.resolution_y, e2, &tb);
            else
              (void)pngio::save_rgb8(part + ".bench", frame, c.cam.resolution_x, c.cam.resolution_y, e2, a.png_level, &tb);
          }
          if (a.encode_bench) std::remove((part + ".bench").c_str());
          std::lock_guard<std::mutex> g(io_mu);
          if (!ok) {
            std::fprintf(stderr, "Error in rendering video: Could not save image frame \"%s\" due to error: %s\n", file.c_str(), e.c_str());
            failed = 1;
            q_cv.notify_all(); /* device workers waiting for work must see it */
            return;
          }
          for (auto pr : {std::make_pair(&enc_total, &tm), std::make_pair(&bench_total, &tb)}) {
            pr.first->filter += pr.second->filter;
            pr.first->deflate += pr.second->deflate;
            pr.first->checksum += pr.second->checksum;
            pr.first->write += pr.second->write;
            pr.first->raw_bytes += pr.second->raw_bytes;
            pr.first->file_bytes += pr.second->file_bytes;
            pr.first->frames += pr.second->frames;
          }
          std::printf("Rendering frame %zu/%zu...\n", k + 1, times.size());
          if (stats_f)
            std::fprintf(stats_f, "{\"frame\": %zu, \"time\": %.17g, \"device\": %d, \"mode\": \"%s\", \"rays\": %llu, \"steps\": %llu, \"n_pos\": %llu, \"n_neg\": %llu, \"n_none\": %llu, \"n_oob\": %llu, \"kernel_ms\": %.4f, \"mray_steps_per_s\": %.1f, \"batch_frames\": %zu, \"batch_kernel_ms\": %.4f, \"batch_call_ms\": %.4f}\n",
                         k, times[k], device_of(rank), a.mode.c_str(), (unsigned long long)fs.rays, (unsigned long long)fs.steps,
                         (unsigned long long)fs.n_pos, (unsigned long long)fs.n_neg, (unsigned long long)fs.n_none,
                         (unsigned long long)fs.n_oob, fs.kernel_ms, fs.kernel_ms > 0.0 ? (double)fs.steps / fs.kernel_ms / 1e3 : 0.0, nb,
                         batch_ms, batch_call_ms);
        });
      }
      ds.submit_s += pngio::now_s() - t_s0; /* frame copies + time blocked on a full writer queue */
      {
        std::lock_guard<std::mutex> g(q_mu);
        ++batches_done;
      }
      q_cv.notify_all();
    }
  };
  std::vector<std::thread> th;
  for (int r = 0; r < n_workers; ++r) th.emplace_back(worker, r);
  for (auto &t : th) t.join();
  for (ncclComm_t cm : comms) ncclCommDestroy(cm);
  const double t_workers_done = pngio::now_s();
  writers.finish();
  pools.clear(); /* every writer job is done: the page-locked buffers can go */
  const double t_video1 = pngio::now_s();
  if (stats_f) std::fclose(stats_f);
  if (!a.stats.empty()) { /* <stats>.summary.json + a table: who rendered what at which clock, where the host's time went */
    const double wall = t_video1 - t_video0;
    size_t total_frames = 0;
    for (const DeviceSummary &d : dev_sum) total_frames += d.frames;
    std::string js = "{\"frames\": " + std::to_string(total_frames) + ", \"wall_s\": " + std::to_string(wall) +
                     ", \"frames_per_s\": " + std::to_string(wall > 0 ? total_frames / wall : 0.0) +
                     ", \"writers\": " + std::to_string(a.writers) + ", \"png_level\": " + std::to_string(a.png_level) +
                     ", \"gpu_png\": " + (gpu_png ? "true" : "false") +
                     ", \"writer_drain_s\": " + std::to_string(t_video1 - t_workers_done);
    { /* how the two textures reached the devices: the slowest device's time; for RCCL the broadcast call alone as well
       * (root: upload first, then header + 2 x ncclBroadcast; the first collective of a communicator carries its set-up) */
      double sky_max = 0, bcast_max = 0;
      for (const DeviceSummary &d : dev_sum) {
        sky_max = std::max(sky_max, d.sky_s);
        bcast_max = std::max(bcast_max, d.sky_bcast_s);
      }
      const double sky_bytes = (double)c.sky1.rgba.size() + (double)c.sky2.rgba.size();
      char buf[384];
      std::snprintf(buf, sizeof buf,
                    ", \"sky_distribution\": {\"via\": \"%s\", \"bytes\": %.0f, \"seconds\": %.4f, \"broadcast_call_s\": %.4f, "
                    "\"sky_broadcast_gbps\": %.2f}",
                    use_rccl ? "rccl: ncclCommInitAll + curvis_ctx_bcast_skies" : "upload to every device", sky_bytes, sky_max, bcast_max,
                    bcast_max > 0 ? sky_bytes / bcast_max / 1e9 : 0.0);
      js += buf;
    }
    js += ", \"devices\": [";
    std::printf("device  pci_bus_id     frames  kernel ms/frame  render-call ms/frame  fps    sclk MHz  power W  wait s  hand-over s\n");
    for (size_t r = 0; r < dev_sum.size(); ++r) {
      const DeviceSummary &d = dev_sum[r];
      const double kf = d.frames ? d.kernel_ms / d.frames : 0.0, rf = d.frames ? d.render_s * 1e3 / d.frames : 0.0;
      std::printf("%-7zu %-14s %-7zu %-16.3f %-21.3f %-6.1f %-9d %-8d %-7.2f %.2f\n", (size_t)device_of((int)r), d.pci_bus_id.c_str(), d.frames, kf, rf,
                  d.busy_s > 0 ? d.frames / d.busy_s : 0.0, d.sclk_mhz, d.power_w, d.wait_s, d.submit_s);
      char buf[768];
      std::snprintf(buf, sizeof buf,
                    "%s{\"device\": %zu, \"pci_bus_id\": \"%s\", \"frames\": %zu, \"batches\": %zu, \"kernel_ms_per_frame\": %.4f, "
                    "\"render_call_ms_per_frame\": %.4f, \"frames_per_s\": %.2f, \"mray_steps_per_s\": %.1f, \"sclk_mhz\": %d, \"power_w\": %d, "
                    "\"wait_s\": %.3f, \"hand_over_s\": %.3f, \"buffer_wait_s\": %.3f, \"busy_s\": %.3f, \"gpu_png_frames\": %zu, "
                    "\"gpu_png_kernel_ms_per_frame\": %.4f, \"gpu_png_fallback_frames\": %zu}",
                    r ? ", " : "", (size_t)device_of((int)r), d.pci_bus_id.c_str(), d.frames, d.batches, kf, rf, d.busy_s > 0 ? d.frames / d.busy_s : 0.0,
                    d.kernel_ms > 0 ? (double)d.steps / d.kernel_ms / 1e3 : 0.0, d.sclk_mhz, d.power_w, d.wait_s, d.submit_s, d.pool_wait_s, d.busy_s,
                    d.png_frames, d.png_frames ? d.png_ms / d.png_frames : 0.0, d.png_fallback_frames);
      js += buf;
    }
    js += "]";
    for (auto pr : {std::make_pair("encode", &enc_total), std::make_pair("encode_bench", &bench_total)}) {
      const pngio::EncodeTimes &t = *pr.second;
      if (!t.frames) continue;
      const double per = 1e3 / (double)t.frames, cpu = t.filter + t.deflate + t.checksum + t.write;
      char buf[640];
      std::snprintf(buf, sizeof buf,
                    ", \"%s\": {\"frames\": %zu, \"filter_ms\": %.3f, \"deflate_ms\": %.3f, \"checksum_ms\": %.3f, \"write_ms\": %.3f, "
                    "\"thread_ms_per_frame\": %.3f, \"raw_mb_per_frame\": %.3f, \"file_mb_per_frame\": %.3f, \"mb_per_s_per_thread\": %.1f, "
                    "\"frames_per_s_per_thread\": %.1f}",
                    pr.first, t.frames, t.filter * per, t.deflate * per, t.checksum * per, t.write * per, cpu * per, t.raw_bytes / 1e6 / t.frames,
                    t.file_bytes / 1e6 / t.frames, cpu > 0 ? t.raw_bytes / 1e6 / cpu : 0.0, cpu > 0 ? t.frames / cpu : 0.0);
      js += buf;
      std::printf("%s: %zu frames, per frame and writer thread: filter %.2f + deflate %.2f + checksums %.2f + file write %.2f = %.2f ms "
                  "(%.0f MB/s, %.1f frames/s per thread), %.2f -> %.2f MB\n",
                  pr.first, t.frames, t.filter * per, t.deflate * per, t.checksum * per, t.write * per, cpu * per, cpu > 0 ? t.raw_bytes / 1e6 / cpu : 0.0,
                  cpu > 0 ? t.frames / cpu : 0.0, t.raw_bytes / 1e6 / t.frames, t.file_bytes / 1e6 / t.frames);
    }
    js += "}\n";
    std::printf("video: %zu frames in %.2f s wall = %.1f frames/s (%d writer threads, png level %d; writers still busy %.2f s after the last render)\n",
                total_frames, wall, wall > 0 ? total_frames / wall : 0.0, a.writers, a.png_level, t_video1 - t_workers_done);
    if (FILE *sf = std::fopen((a.stats + ".summary.json").c_str(), "w")) {
      std::fputs(js.c_str(), sf);
      std::fclose(sf);
    }
  }
  if (failed) return 1;
  if (!panic_msg.empty()) {
    std::fprintf(stderr, "thread 'main' panicked: %s (frame %zu of %zu)\n", panic_msg.c_str(), n_frames, times.size());
    return 101;
  }
  return 0;
}

}  // namespace

int main(int argc, char **argv) {
  const Args a = parse_args(argc, argv);
  if (a.sub == "image") return image_main(a);
  if (a.sub == "video") return video_main(a);
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
