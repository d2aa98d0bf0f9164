"""A SECOND, independent restatement of the reference's per-pixel path (R1-R10 of SURVEY 8a) in plain Python,
written from the reference text (file:line below), not from oracle/curvis_oracle.c.  Python floats are IEEE
doubles, `math` calls the platform libm -- the functions a Linux build of the Rust reference calls -- and no
operation is fused, so this must agree BIT FOR BIT with the oracle's libm flavour.  Test infrastructure only
(tests/test_ref_python.py); pure-Python loops, small frames.

Third-party arithmetic (nalgebra 0.33.0), restated from its documented behaviour: dot = (a0*b0 + a1*b1) + a2*b2,
normalize = v / sqrt(dot(v, v)), cross, matrix * vector accumulating left to right, 3x3 product likewise,
Rotation3::face_towards(dir, up) = columns [normalize(up x z), z x x (normalised), z = normalize(dir)],
inverse = transpose.
"""
import math

PI = math.pi


# ---- nalgebra pieces --------------------------------------------------------------------------------
def dot(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def norm(a):
    return math.sqrt(dot(a, a))


def normalize(a):
    n = norm(a)
    return [a[0] / n, a[1] / n, a[2] / n]


def cross(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def mat_vec(m, v):  # m[i][j]
    return [(m[i][0] * v[0] + m[i][1] * v[1]) + m[i][2] * v[2] for i in range(3)]


def mat_mul(a, b):
    return [[(a[i][0] * b[0][j] + a[i][1] * b[1][j]) + a[i][2] * b[2][j] for j in range(3)] for i in range(3)]


def transpose(m):
    return [[m[j][i] for j in range(3)] for i in range(3)]


def face_towards(direction, up):
    z = normalize(direction)
    x = normalize(cross(up, z))
    y = normalize(cross(z, x))
    return [[x[0], y[0], z[0]], [x[1], y[1], z[1]], [x[2], y[2], z[2]]]


# ---- src/algebra.rs ---------------------------------------------------------------------------------
def orientation(forward, up):
    """Orientation::new (:16-38) + rotation_matrix_from_forward_up_pairs (:64-74): (rotation, inverse)"""
    if norm(cross(forward, up)) == 0.0:
        raise ValueError("Forward and up vectors must not be parallel")
    r1 = face_towards([1.0, 0.0, 0.0], [0.0, 0.0, 1.0])
    r2 = face_towards(forward, up)
    rot = mat_mul(r2, transpose(r1))
    return rot, transpose(rot)


def rem_euclid(a, b):
    r = math.fmod(a, b)
    return r + b if r < 0.0 else r


def normalize_theta_phi(theta, phi):  # :106-116
    if theta < 0.0:
        theta, phi = abs(theta), phi + PI
    return theta, rem_euclid(phi, 2.0 * PI)


def theta_phi_from_vector3(v):  # :128-134
    r = norm(v)
    return normalize_theta_phi(math.acos(v[2] / r), math.atan2(v[1], v[0]))


# ---- src/metrics.rs ---------------------------------------------------------------------------------
class Ellis:  # :417-421
    def __init__(self, rho):
        self.rho = rho

    def r(self, l):
        return math.sqrt(self.rho * self.rho + l * l)

    def r_squared(self, l):
        return self.rho * self.rho + l * l

    def r_derivative(self, l):
        return l / self.r(l)


class Interstellar:  # :461-485
    def __init__(self, m, a, rho):
        self.m, self.a, self.rho = m, a, rho

    def scaled_distance(self, l):
        return 2.0 * (abs(l) - self.a) / (PI * self.m)

    def r(self, l):
        if abs(l) > self.a:
            x = self.scaled_distance(l)
            return self.rho + self.m * (x * math.atan(x) - math.log(1.0 + x * x) / 2.0)
        return self.rho

    def r_squared(self, l):
        r = self.r(l)
        return r * r

    def r_derivative(self, l):
        if abs(l) > self.a:
            x = self.scaled_distance(l)
            return (2.0 / PI) * math.copysign(1.0, l) * math.atan(x)
        return 0.0


class Flat:  # :501-505
    def r(self, l):
        return l

    def r_squared(self, l):
        return l * l

    def r_derivative(self, l):
        return 1.0


def new_photon(metric, position, direction):  # :301-334 -> (x contravariant, p covariant)
    d = normalize(direction)
    r = metric.r(position[1])
    return list(position), [1.0, d[0], d[1] * r, d[2] * r * math.sin(position[2])]


def update_relativistic_object(metric, x, p, delta):  # :283-297 with :223-244 and :247-270
    l, theta = x[1], x[2]
    g00c = 1.0 / -1.0
    g11c = 1.0 / 1.0
    g22c = 1.0 / metric.r_squared(l)
    s = math.sin(theta)
    g33c = 1.0 / (metric.r_squared(l) * (s * s))
    dx = [p[0] * g00c, p[1] * g11c, p[2] * g22c, p[3] * g33c]
    b_squared = p[2] * p[2] + (p[3] * p[3]) / (s * s)
    r = metric.r(l)
    dp = [0.0,
          b_squared * metric.r_derivative(l) / (r * r * r),
          (p[3] * p[3]) * (math.cos(theta) / (metric.r_squared(l) * (s * s * s))),
          0.0]
    for i in range(4):
        x[i] = x[i] + dx[i] * delta
        p[i] = p[i] + dp[i] * delta


def relativistic_vector_to_direction(metric, p, x):  # :339-349 via to_contravariant :190-203
    l, theta = x[1], x[2]
    s = math.sin(theta)
    v = [p[0] * (1.0 / -1.0), p[1] * (1.0 / 1.0), p[2] * (1.0 / metric.r_squared(l)),
         p[3] * (1.0 / (metric.r_squared(l) * (s * s)))]
    r = metric.r(l)
    return [v[1] * 1.0, v[2] * r, v[3] * r]  # the third component uses frame_field_22 as well (:347)


# ---- src/systems.rs, src/cameras.rs, src/images.rs ---------------------------------------------------
def escape_photon(metric, x, p, delta, max_iterations, max_radius):  # :115-139 -> (code, steps)
    if abs(x[1]) > max_radius:
        raise ValueError("Photon already beyond the maximum radius. Cannot evaluate escape.")
    for k in range(max_iterations):
        update_relativistic_object(metric, x, p, delta)
        if x[1] > max_radius:
            return 1, k + 1
        if x[1] < -max_radius:
            return -1, k + 1
    return 0, max_iterations


class Camera:  # src/cameras.rs:79-172
    def __init__(self, position, forward, up, focal_length, sensor_diagonal, res_x, res_y):
        self.position = list(position)
        self.rot, _ = orientation(forward, up)
        self.focal = focal_length
        aspect = float(res_x) / float(res_y)
        self.sensor_h = math.sqrt((sensor_diagonal * sensor_diagonal) / (aspect * aspect + 1.0))
        self.sensor_w = aspect * self.sensor_h
        self.res_x, self.res_y = res_x, res_y

    def outward_world(self, px, py):
        h = 0.5 - (float(py) / float(self.res_y))
        w = (float(px) / float(self.res_x)) - 0.5
        v = normalize([self.focal * 1.0, -self.sensor_w * w, self.sensor_h * h])
        return mat_vec(self.rot, v)


def as_u32(v):  # Rust `as u32`: saturating, NaN -> 0
    if v != v or v <= 0.0:
        return 0
    return 4294967295 if v >= 4294967295.0 else int(v)


def sky_indices(direction, width, height, inverse_rotation=None):  # src/images.rs:115-174, default orientation
    w = direction if inverse_rotation is None else mat_vec(inverse_rotation, direction)
    theta, phi = theta_phi_from_vector3(w)
    theta, phi = normalize_theta_phi(theta, phi)
    y = as_u32((theta / PI) * float(height))
    x = as_u32(rem_euclid(0.5 - phi / (2.0 * PI), 1.0) * float(width))
    return x, y


def render_pixel(metric, camera, px, py, max_iterations, max_radius, delta):
    """-> (x[4], p[4], steps, code, tx, ty) with raw (unclamped) texel indices for a width x height sky given later"""
    x, p = new_photon(metric, camera.position, camera.outward_world(px, py))
    code, steps = escape_photon(metric, x, p, delta, max_iterations, max_radius)
    d = relativistic_vector_to_direction(metric, p, x) if code != 0 else None
    return x, p, steps, code, d


# =====================================================================================================
# CLI variant: render_image_efficient (src/systems.rs:333-527), compute_escape_angle (:203-261),
# doubly_sample_function (src/sampling.rs:46-245); interp 1.0.3 interp_slice and nalgebra's
# rotation_between / from_axis_angle restated from their documented behaviour.
# =====================================================================================================
EPS = 2.220446049250313e-16


def vector3_from_theta_phi(theta, phi):  # src/algebra.rs:118-126
    theta, phi = normalize_theta_phi(theta, phi)
    return [math.sin(theta) * math.cos(phi), math.sin(theta) * math.sin(phi), math.cos(theta)]


def identity():
    return [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]


def from_axis_angle(u, angle):
    """nalgebra Rotation3::from_axis_angle(&Unit(u), angle): identity iff angle == 0 (NaN != 0)"""
    if not (angle != 0.0):
        return identity()
    ux, uy, uz = u
    sqx, sqy, sqz = ux * ux, uy * uy, uz * uz
    sn, cs = math.sin(angle), math.cos(angle)
    omc = 1.0 - cs
    return [[sqx + (1.0 - sqx) * cs, ux * uy * omc - uz * sn, ux * uz * omc + uy * sn],
            [ux * uy * omc + uz * sn, sqy + (1.0 - sqy) * cs, uy * uz * omc - ux * sn],
            [ux * uz * omc - uy * sn, uy * uz * omc + ux * sn, sqz + (1.0 - sqz) * cs]]


def rotation_between(a, b):
    """nalgebra Rotation3::rotation_between(a, b) -> matrix or None (antiparallel)"""
    an, bn = norm(a), norm(b)
    if not (an <= 0.0) and not (bn <= 0.0):  # try_normalize(0.0)
        na_ = [a[0] / an, a[1] / an, a[2] / an]
        nb_ = [b[0] / bn, b[1] / bn, b[2] / bn]
        c = cross(na_, nb_)
        sq = dot(c, c)
        if sq > EPS * EPS:  # Unit::try_new(c, f64::EPSILON)
            n = math.sqrt(sq)
            return from_axis_angle([c[0] / n, c[1] / n, c[2] / n], math.acos(dot(na_, nb_)) * 1.0)
        if dot(na_, nb_) < 0.0:
            return None
    return identity()


def rotation_from_two_vectors(v1, v2):  # src/algebra.rs:92-101
    if norm(cross(v1, v2)) == 0.0:
        raise ValueError("v1 and v2 must not be parallel")
    r = rotation_between(v1, v2)
    if r is None:
        raise ValueError("rotation_between returned None")
    return r


def compute_escape_angle(metric, l, alpha, delta, max_iterations, max_radius):
    """-> (angle, space in {+1.0, -1.0}) or (nan, nan); also returns the number of Euler steps"""
    x, p = new_photon(metric, [0.0, l, PI / 2.0, 0.0], [math.cos(alpha), 0.0, math.sin(alpha)])
    code, steps = escape_photon(metric, x, p, delta, max_iterations, max_radius)
    if code == 0:
        return float("nan"), float("nan"), steps
    tangent = relativistic_vector_to_direction(metric, p, x)
    world_position = vector3_from_theta_phi(x[2], x[3])
    rot = rotation_from_two_vectors([1.0, 0.0, 0.0], world_position)
    wd = normalize(mat_vec(rot, tangent))
    vx, vy = dot(wd, [1.0, 0.0, 0.0]), dot(wd, [0.0, 1.0, 0.0])
    angle = math.acos(vx) if vy >= 0.0 else 2.0 * PI - math.acos(vx)
    return angle, (1.0 if code > 0 else -1.0), steps


def _finite(b):
    return all(math.isfinite(v) for v in b)


def doubly_sample_function(a_min, a_max, n0, max_iterations, thr1, thr2, f):
    """src/sampling.rs:46-124 (with :129-140, :144-195, :198-245); f(alpha) -> (e, s)"""
    step = (a_max - a_min) / float(n0 - 1)
    pts = [[a_min + float(i) * step] for i in range(n0)]
    pts = [[a[0], *f(a[0])] for a in pts]
    pts = [b for b in pts if _finite(b)]
    iteration = 0
    while iteration < max_iterations:
        prev = len(pts)
        pts = [b for b in pts if _finite(b)]
        if len(pts) < 3:
            raise ValueError("bipoints list has length < 3")
        new, i, n = [], 0, len(pts)
        while i < n - 2:
            b1, b2, b3 = pts[i], pts[(i + 1) % n], pts[(i + 2) % n]
            area1 = abs((b1[0] * b2[1] + b2[0] * b3[1] + b3[0] * b1[1]) - (b1[1] * b2[0] + b2[1] * b3[0] + b3[1] * b1[0]))
            area2 = abs((b1[0] * b2[2] + b2[0] * b3[2] + b3[0] * b1[2]) - (b1[2] * b2[0] + b2[2] * b3[0] + b3[2] * b1[0]))
            if not (area1 > thr1 or area2 > thr2):
                new.append(list(b1))
                i += 1
                continue
            a1, a2 = (b1[0] + b2[0]) / 2.0, (b2[0] + b3[0]) / 2.0
            e1 = f(a1)
            e2 = f(a2)
            new += [list(b1), [a1, *e1], list(b2), [a2, *e2]]
            i += 2
        pts = [b for b in new if _finite(b)]
        if len(pts) < prev or len(pts) == prev:
            break
        iteration += 1
    return [b[0] for b in pts], [b[1] for b in pts], [b[2] for b in pts]


def interp_slice(x, y, xp):
    """interp 1.0.3: per-segment slope m = dy/dx and intercept c = y - x*m; index = (number of leading x < xp) - 1,
    saturating, clamped to n-2; result m*xp + c (linear extrapolation outside)"""
    n = len(x)
    if n == 0:
        return [0.0 for _ in xp]
    if n == 1:
        return [y[0] for _ in xp]
    m = [(y[i + 1] - y[i]) / (x[i + 1] - x[i]) for i in range(n - 1)]
    c = [y[i] - x[i] * m[i] for i in range(n - 1)]
    out = []
    for q in xp:
        k = 0
        while k < n and x[k] < q:
            k += 1
        i = min(max(k - 1, 0), n - 2)
        out.append(m[i] * q + c[i])
    return out


def render_image_efficient(metric, camera, sky_pos, sky_neg, max_iter, max_radius, delta, alpha_nums, max_iter_sampling,
                           thr1, thr2):
    """-> (rgb[H][W] tuples, sample table (a, e, s), total Euler steps, calls)"""
    W, H = camera.res_x, camera.res_y
    cam_bg = vector3_from_theta_phi(camera.position[2], camera.position[3])
    tan_dirs, axes = [], []
    for i in range(W):
        for j in range(H):
            out_tan = camera.outward_world(i, j)
            out_bg = mat_vec(rotation_from_two_vectors([1.0, 0.0, 0.0], cam_bg), out_tan)
            tan_dirs.append(out_tan)
            axes.append(cross(cam_bg, out_bg))
    img_alphas = [math.acos(dot(d, [1.0, 0.0, 0.0])) for d in tan_dirs]
    stats = {"steps": 0, "calls": 0}

    def f(alpha):
        a, s, st = compute_escape_angle(metric, camera.position[1], alpha, delta, max_iter, max_radius)
        stats["steps"] += st
        stats["calls"] += 1
        return a, s

    sa, se, ss = doubly_sample_function(-0.1 * PI, 1.1 * PI, alpha_nums, max_iter_sampling, thr1, thr2, f)
    esc = interp_slice(sa, se, img_alphas)
    spc = interp_slice(sa, ss, img_alphas)
    rgb = [[(0, 0, 0)] * W for _ in range(H)]
    for index, (axis, e, s) in enumerate(zip(axes, esc, spc)):
        i, j = index // H, index % H
        an = norm(axis)
        u = [axis[0] / an, axis[1] / an, axis[2] / an] if an != 0.0 else [float("nan")] * 3
        final = mat_vec(from_axis_angle(u, e), cam_bg)
        if s == 1.0 or s == -1.0:
            sky = sky_pos if s == 1.0 else sky_neg
            tx, ty = sky_indices(final, sky.shape[1], sky.shape[0])
            tx, ty = min(tx, sky.shape[1] - 1), min(ty, sky.shape[0] - 1)
            rgb[j][i] = tuple(int(v) for v in sky[ty, tx, :3])
    return rgb, (sa, se, ss), stats["steps"], stats["calls"]
