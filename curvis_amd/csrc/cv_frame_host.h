/* cv_frame_host.h -- the two ONCE-PER-FRAME host values of render_image_efficient, taken over the PLATFORM libm.
 *
 *   cam_bg = vector3_from_theta_phi(theta_cam, phi_cam)              src/systems.rs:393-397, src/algebra.rs:118-126
 *   rot_bg = rotation_from_two_vectors(x^, cam_bg)                    src/systems.rs:411,     src/algebra.rs:92-101
 *
 * The reference computes them on the host with Rust's f64::{sin, cos, acos}, i.e. glibc on x86-64 Linux; so does this
 * file -- they are host values of the orchestration, not device math, and the pixel ON the optical axis depends on their
 * last bit: its rotation axis cam_bg x (rot_bg x^) is an exact zero (-> NaN -> texel (0, 0)) or rounding noise
 * (DESIGN.md section 3).  With cv_math.h here (rounds 1-5) that pixel differed from every glibc flavour of the oracle in
 * 2 of the 240 orbit frames.  sin and cos of one value are ONE glibc sincos() call: what LLVM emits for llvm.sin.f64 +
 * llvm.cos.f64 of one operand in one block on *-linux-gnu (oracle/llvm_sincos_probe.c), the oracle's CVO_LIBM_SINCOS*
 * flavours.  Everything per ray and per pixel stays cv_math.h (CvMath), identical on x86-64 and gfx950.
 * Host only: included by curvis_hip.hip (through efficient_host.h) and by the x86 twin of the tests. */
#ifndef CURVIS_CV_FRAME_HOST_H
#define CURVIS_CV_FRAME_HOST_H

#include <math.h>

#include "cv_efficient.h"

#if !defined(__GLIBC__)
#error "cv_frame_host.h takes sin/cos/acos from glibc (sincos() is a GNU extension; g++ and clang++ define _GNU_SOURCE)"
#endif

namespace cvk {

struct PlatformLibm {
  static void sincos(double x, double *s, double *c) { ::sincos(x, s, c); }
  static double acos(double x) { return ::acos(x); }
};

/* false == the reference's panic "v1 and v2 must not be parallel" (camera on the x axis of the background space) */
inline bool efficient_frame_pose(double theta_cam, double phi_cam, EfficientFrame &F) {
  vector3_from_theta_phi<PlatformLibm>(theta_cam, phi_cam, F.cam_bg);
  const double ex[3] = {1.0, 0.0, 0.0};
  return rotation_from_two_vectors<PlatformLibm>(ex, F.cam_bg, F.rot_bg);
}

}  // namespace cvk
#endif
