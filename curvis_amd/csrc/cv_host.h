/* cv_host.h -- host-side (C++) restatement of the small amount of linear algebra that sits
 * between the reference's settings and the kernels: Orientation::new (src/algebra.rs:16-38),
 * rotation_matrix_from_forward_up_pairs (:64-74), Camera::new (src/cameras.rs:79-122).
 * nalgebra 0.33.0 evaluation order (dot = a0b0+a1b1+a2b2 left to right, normalize = v/norm,
 * M*v column-accumulated, face_towards columns [x y z], inverse = transpose).
 * Compile with -ffp-contract=off.
 */
#ifndef CURVIS_CV_HOST_H
#define CURVIS_CV_HOST_H
#include <cmath>
#include <cstring>

namespace cvh {

struct Vec3 {
  double x, y, z;
};
inline double dot(const Vec3 &a, const Vec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double norm(const Vec3 &a) { return std::sqrt(dot(a, a)); }
inline Vec3 cross(const Vec3 &a, const Vec3 &b) {
  return Vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline Vec3 normalize(const Vec3 &a) {
  const double n = norm(a);
  return Vec3{a.x / n, a.y / n, a.z / n};
}

/* 3x3 row-major */
struct Mat3 {
  double m[9];
};
inline Vec3 mul(const Mat3 &A, const Vec3 &v) {
  return Vec3{(A.m[0] * v.x + A.m[1] * v.y) + A.m[2] * v.z, (A.m[3] * v.x + A.m[4] * v.y) + A.m[5] * v.z,
              (A.m[6] * v.x + A.m[7] * v.y) + A.m[8] * v.z};
}
inline Mat3 mul(const Mat3 &A, const Mat3 &B) {
  Mat3 C;
  for (int j = 0; j < 3; ++j) {
    Vec3 col = mul(A, Vec3{B.m[j], B.m[3 + j], B.m[6 + j]});
    C.m[j] = col.x;
    C.m[3 + j] = col.y;
    C.m[6 + j] = col.z;
  }
  return C;
}
inline Mat3 transpose(const Mat3 &A) {
  return Mat3{{A.m[0], A.m[3], A.m[6], A.m[1], A.m[4], A.m[7], A.m[2], A.m[5], A.m[8]}};
}
inline Mat3 identity() { return Mat3{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }

/* nalgebra Rotation3::face_towards(dir, up) */
inline Mat3 face_towards(const Vec3 &dir, const Vec3 &up) {
  const Vec3 z = normalize(dir);
  const Vec3 x = normalize(cross(up, z));
  const Vec3 y = normalize(cross(z, x));
  return Mat3{{x.x, y.x, z.x, x.y, y.y, z.y, x.z, y.z, z.z}};
}

struct Orientation {
  Vec3 forward, up;
  Mat3 rotation, inverse_rotation;
};

/* src/algebra.rs:16-38; false == the reference's panic (forward, up parallel) */
inline bool orientation_new(const Vec3 &forward, const Vec3 &up, Orientation &o) {
  if (norm(cross(forward, up)) == 0.0) return false;
  const Mat3 r1 = face_towards(Vec3{1, 0, 0}, Vec3{0, 0, 1});
  const Mat3 r2 = face_towards(forward, up);
  o.rotation = mul(r2, transpose(r1));
  o.inverse_rotation = transpose(o.rotation);
  o.forward = forward;
  o.up = mul(o.rotation, Vec3{0, 0, 1});
  return true;
}

}  // namespace cvh
#endif
