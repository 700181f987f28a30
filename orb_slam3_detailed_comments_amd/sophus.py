"""Host-side mirror of the pose types the reference's matchers take: Sophus::SE3f and Sophus::Sim3f.

A C++ caller hands the drop-in ORBmatcher its own Sophus objects (include/orb_slam3_amd/ORBmatcher.h reads unit_quaternion() / translation()
from them).  A Python caller has no Sophus, so these two classes restate the reference's VENDORED Sophus in float32, statement by statement
(reference: Thirdparty/Sophus/sophus/so3.hpp, se3.hpp, rxso3.hpp, sim3.hpp; the Eigen quaternion operations underneath:
Eigen/src/Geometry/Quaternion.h of Eigen >= 3.3, which the vendored Sophus requires, Thirdparty/Sophus/CMakeLists.txt:35).  Why this matters:
Sophus keeps a rotation as a unit quaternion; `SE3f(R, t)` converts the matrix once (Shoemake's branch on the trace), `rotationMatrix()`
converts back (`toRotationMatrix`), and `T * p` rotates by the quaternion without forming the matrix - each rounds differently from `R @ p + t`
in the last bit, and the searches downstream are decided by last bits.  Host logic only: nothing here runs on the GPU.
"""
import numpy as np

f32 = np.float32
_ONE, _TWO, _HALF = f32(1.0), f32(2.0), f32(0.5)


def _cross(a, b):
    """MatrixBase::cross for 3-vectors (Eigen/src/Geometry/OrthoMethods.h)."""
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], f32)


def quaternion_from_matrix(R):
    """Eigen::Quaternionf(Matrix3f): quaternionbase_assign_impl<Other,3,3>::run.  Returns coeffs() = (x, y, z, w)."""
    m = np.asarray(R, f32).reshape(3, 3)
    q = np.zeros(4, f32)
    t = m[0, 0] + (m[1, 1] + m[2, 2])                 # trace(): Eigen's unrolled 3-term reduction, a0 + (a1 + a2)
    if t > 0:
        t = np.sqrt(t + _ONE, dtype=f32)
        q[3] = _HALF * t
        t = _HALF / t
        q[0] = (m[2, 1] - m[1, 2]) * t
        q[1] = (m[0, 2] - m[2, 0]) * t
        q[2] = (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + _ONE, dtype=f32)
        q[i] = _HALF * t
        t = _HALF / t
        q[3] = (m[k, j] - m[j, k]) * t
        q[j] = (m[j, i] + m[i, j]) * t
        q[k] = (m[k, i] + m[i, k]) * t
    return q


def quaternion_to_matrix(q):
    """QuaternionBase::toRotationMatrix for coeffs (x, y, z, w)."""
    x, y, z, w = [f32(v) for v in q]
    tx, ty, tz = _TWO * x, _TWO * y, _TWO * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[_ONE - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, _ONE - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, _ONE - (txx + tyy)]], f32)


def _squared_norm(q):
    """coeffs().squaredNorm() of a Quaternionf: one SSE packet on the reference's x86-64 target, reduced by predux<Packet4f> =
    (x*x + z*z) + (y*y + w*w).  Only the HOST mirror evaluates this (normalising constructors, scale()); the device never does."""
    return (q[0] * q[0] + q[2] * q[2]) + (q[1] * q[1] + q[3] * q[3])


def _normalized(q):
    n = np.sqrt(_squared_norm(q), dtype=f32)
    return (q / n).astype(f32)


def _rotate(q, p):
    """SO3::operator*(point), so3.hpp:357-367."""
    v = q[:3]
    uv = _cross(v, p)
    uv = uv + uv
    return (p + q[3] * uv) + _cross(v, uv)


class SE3f:
    """Sophus::SE3f.  SE3f(R, t): from a rotation matrix (se3.hpp:480-482, no normalisation); SE3f(q=(x, y, z, w), t=...): from a quaternion
    (se3.hpp:488-490, normalised by the SO3 constructor, so3.hpp:480-487); SE3f(): identity."""

    def __init__(self, R=None, t=None, q=None, _raw=None):
        self.t = np.zeros(3, f32) if t is None else np.ascontiguousarray(t, f32).reshape(3).copy()
        if _raw is not None:
            self.q = np.ascontiguousarray(_raw, f32).copy()
        elif q is not None:
            self.q = _normalized(np.ascontiguousarray(q, f32).reshape(4))
        elif R is not None:
            self.q = quaternion_from_matrix(R)
        else:
            self.q = np.array([0, 0, 0, 1], f32)

    def unit_quaternion(self):
        return self.q

    def translation(self):
        return self.t

    def rotationMatrix(self):
        return quaternion_to_matrix(self.q)

    def inverse(self):
        """se3.hpp:208-211: invR = so3().inverse() (conjugate, re-normalised by the quaternion constructor); SE3(invR, invR * (t * -1))."""
        qi = _normalized(np.array([-self.q[0], -self.q[1], -self.q[2], self.q[3]], f32))
        return SE3f(t=_rotate(qi, self.t * f32(-1.0)), _raw=qi)

    def __mul__(self, other):
        if isinstance(other, SE3f):                   # se3.hpp:304-308 over so3.hpp:325-340 (the product quaternion is normalised)
            a, b = self.q, other.q
            ax, ay, az, aw = a
            bx, by, bz, bw = b
            q = np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx,
                          aw * bw - ax * bx - ay * by - az * bz], f32)
            return SE3f(t=self.t + _rotate(a, other.t), _raw=_normalized(q))
        p = np.ascontiguousarray(other, f32).reshape(3)   # se3.hpp:321-324
        return _rotate(self.q, p) + self.t


class Sim3f:
    """Sophus::Sim3f as the reference builds it: Sim3f(scale, R, t) = Sim3f(RxSO3f(scale, R), t) (rxso3.hpp:460-466: quaternion(R) * sqrt(scale);
    src/Converter.cc:410-413).  Sim3f(q=..., t=...) takes the scaled quaternion itself (sim3.hpp:402-409)."""

    def __init__(self, scale=None, R=None, t=None, q=None):
        self.t = np.zeros(3, f32) if t is None else np.ascontiguousarray(t, f32).reshape(3).copy()
        if q is not None:
            self.q = np.ascontiguousarray(q, f32).reshape(4).copy()
        else:
            self.q = (quaternion_from_matrix(np.eye(3, dtype=f32) if R is None else R) * np.sqrt(f32(1.0 if scale is None else scale), dtype=f32)).astype(f32)

    def quaternion(self):
        return self.q

    def translation(self):
        return self.t

    def scale(self):
        return _squared_norm(self.q)                  # rxso3.hpp:350

    def rotationMatrix(self):
        return quaternion_to_matrix(_normalized(self.q))    # rxso3.hpp:341-345

    def _act(self, p):
        """RxSO3::operator*(point), rxso3.hpp:265-273."""
        v = self.q[:3]
        tv = _cross(v, p)
        tv = tv + tv
        return self.scale() * p + (self.q[3] * tv + _cross(v, tv))

    def inverse(self):
        """sim3.hpp:129-132 over rxso3.hpp:156-158 (quaternion().inverse() = conjugate().coeffs() / squaredNorm())."""
        n2 = _squared_norm(self.q)
        qi = (np.array([-self.q[0], -self.q[1], -self.q[2], self.q[3]], f32) / n2).astype(f32)
        inv = Sim3f(q=qi)
        inv.t = inv._act(self.t * f32(-1.0))
        return inv

    def __mul__(self, p):
        return self._act(np.ascontiguousarray(p, f32).reshape(3)) + self.t     # sim3.hpp:226-229


def as_se3(pose):
    """An SE3f, or (R, t) taken through Sophus' SE3(R, t) constructor - the way a pose held as matrices reaches the reference (Frame::SetPose)."""
    return pose if isinstance(pose, SE3f) else SE3f(pose[0], pose[1])
