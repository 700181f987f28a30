// eigen_small.h — TEST INFRASTRUCTURE (part of oracle/; never shipped).  The part of Eigen that the reference's camera models
// (src/CameraModels/Pinhole.cpp, src/CameraModels/KannalaBrandt8.cpp) need beyond the Vector2f / Vector3f / Matrix3f stand-ins of slam_types.h, so
// that both files compile UNMODIFIED and in place (oracle/Makefile): Vector2d / Vector3d / Matrix<double,2,3> (the double overloads, never executed
// by the tests), Matrix<float,3,4> with block comma initialisers, Matrix4f with row assignment, Vector4f::head, and Eigen::JacobiSVD<Matrix4f>.
//
// Eigen is an external dependency of the reference (CMakeLists.txt:46 find_package(Eigen3 3.1.0 REQUIRED); the vendored Sophus needs >= 3.3,
// Thirdparty/Sophus/CMakeLists.txt:35) and is not installed in this image.  What follows restates the PUBLISHED algorithm of Eigen 3.3.7:
//   * Eigen/src/SVD/JacobiSVD.h, JacobiSVD::compute for a square real matrix (no QR preconditioner): scale by the largest |coefficient|, sweeps
//     over (p, q), p = 1..n-1, q = 0..p-1, threshold max(numeric_limits::min, 2 eps * maxDiagEntry), two-sided rotations, singular values made
//     positive and sorted in descending order (columns of V swapped along);
//   * Eigen/src/misc/RealSvd2x2.h, real_2x2_jacobi_svd;
//   * Eigen/src/Jacobi/Jacobi.h, JacobiRotation::makeJacobi / operator* / transpose, apply_rotation_in_the_plane (x <- c x + s y, y <- -s x + c y).
// All in the scalar type of the matrix (fp32 for Matrix4f), as Eigen does.  PARITY UNPINNED for this header alone (no Eigen here to run against);
// every line of the camera models above it is the reference's own.
#ifndef ORBX_EIGEN_SMALL_H
#define ORBX_EIGEN_SMALL_H
#include <cmath>
#include <limits>

#include <type_traits>
namespace Eigen {

enum { ComputeFullU = 0x04, ComputeThinU = 0x08, ComputeFullV = 0x10, ComputeThinV = 0x20 };

template <typename T, int R, int C> struct SmallMatrix;

// writable view of one row (A.row(i) = ...)
template <typename T, int R, int C> struct RowRef {
    SmallMatrix<T, R, C>* M; int r;
    RowRef& operator=(const SmallMatrix<T, 1, C>& v) { for (int j = 0; j < C; j++) M->m[r][j] = v.m[0][j]; return *this; }
    operator SmallMatrix<T, 1, C>() const { SmallMatrix<T, 1, C> v; for (int j = 0; j < C; j++) v.m[0][j] = M->m[r][j]; return v; }
};

template <typename T, int R, int C> struct SmallMatrix {
    T m[R][C];
    SmallMatrix() { for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) m[i][j] = T(0); }
    T& operator()(int r, int c) { return m[r][c]; }
    const T& operator()(int r, int c) const { return m[r][c]; }
    // vectors (one row or one column)
    T& operator()(int i) { return C == 1 ? m[i][0] : m[0][i]; }
    const T& operator()(int i) const { return C == 1 ? m[i][0] : m[0][i]; }
    T& operator[](int i) { return (*this)(i); }
    const T& operator[](int i) const { return (*this)(i); }
    SmallMatrix<T, 1, C> row(int r) const { SmallMatrix<T, 1, C> v; for (int j = 0; j < C; j++) v.m[0][j] = m[r][j]; return v; }
    RowRef<T, R, C> row(int r) { return RowRef<T, R, C>{this, r}; }
    static SmallMatrix Identity() { SmallMatrix I; for (int i = 0; i < R && i < C; i++) I.m[i][i] = T(1); return I; }
#ifdef ORBX_LOOPCLOSING_WORLD      // (loopclosing_world.h: compile check of LoopClosing.cc; declared only, nothing of it is linked)
    template <class U> SmallMatrix<U, R, C> castTo() const;
    template <class U> typename std::conditional<std::is_same<U, float>::value && R == 3 && C == 1, Vector3f, SmallMatrix<U, R, C> >::type cast() const;
    struct CommaInitSM { CommaInitSM& operator,(T v); };
    CommaInitSM operator<<(T v);
    SmallMatrix operator/(T s) const; SmallMatrix operator*(T s) const; SmallMatrix operator+(const SmallMatrix& o) const; SmallMatrix operator-(const SmallMatrix& o) const;
    SmallMatrix& operator*=(T s); T norm() const; void setZero(); static SmallMatrix Zero(); SmallMatrix<T, C, R> transpose() const;
    friend std::ostream& operator<<(std::ostream& os, const SmallMatrix& M) { return os; }
#endif
};
template <typename T, int C> SmallMatrix<T, 1, C> operator*(T s, const SmallMatrix<T, 1, C>& v) { SmallMatrix<T, 1, C> r; for (int j = 0; j < C; j++) r.m[0][j] = s * v.m[0][j]; return r; }
template <typename T, int C> SmallMatrix<T, 1, C> operator-(const SmallMatrix<T, 1, C>& a, const SmallMatrix<T, 1, C>& b) {
    SmallMatrix<T, 1, C> r; for (int j = 0; j < C; j++) r.m[0][j] = a.m[0][j] - b.m[0][j]; return r;
}

typedef SmallMatrix<double, 2, 1> Vector2d;
typedef SmallMatrix<double, 3, 1> Vector3d;

// Vector4f: head(3) and the division of the head by the last coefficient (Eigen's operator/ is a true division, scalar_quotient_op)
struct Vector4f : SmallMatrix<float, 4, 1> {
    Vector3f head(int n) const { (void)n; return Vector3f(m[0][0], m[1][0], m[2][0]); }
};

// Matrix<float,3,4>: [R | t]
struct Matrix34f : SmallMatrix<float, 3, 4> {
    Vector3f col(int j) const { return Vector3f(m[0][j], m[1][j], m[2][j]); }
    template <int BR, int BC> Matrix3f block(int r0, int c0) const {
        static_assert(BR == 3 && BC == 3, "only 3x3 blocks");
        Matrix3f B; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) B(i, j) = m[r0 + i][c0 + j]; return B;
    }
};
struct CommaInit34 {
    Matrix34f* M; int c;
    CommaInit34& operator,(const Vector3f& v) { for (int i = 0; i < 3; i++) M->m[i][c] = v.d[i]; c++; return *this; }
    CommaInit34& operator,(const Matrix3f& B) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M->m[i][c + j] = B(i, j); c += 3; return *this; }
};
inline CommaInit34 operator<<(Matrix34f& M, const Matrix3f& B) { CommaInit34 ci{&M, 0}; ci, B; return ci; }

// Matrix3f << a, b, c, ... (row by row)
struct CommaInit3 {
    Matrix3f* M; int k;
    CommaInit3& operator,(float v) { (*M)(k / 3, k % 3) = v; k++; return *this; }
};
inline CommaInit3 operator<<(Matrix3f& M, float v) { CommaInit3 ci{&M, 0}; ci, v; return ci; }

typedef SmallMatrix<float, 4, 4> Matrix4f;

// Eigen/src/Jacobi/Jacobi.h
template <typename Scalar> struct JacobiRotation {
    Scalar m_c, m_s;
    JacobiRotation() : m_c(0), m_s(0) {}
    JacobiRotation(Scalar c, Scalar s) : m_c(c), m_s(s) {}
    Scalar& c() { return m_c; }
    Scalar& s() { return m_s; }
    Scalar c() const { return m_c; }
    Scalar s() const { return m_s; }
    JacobiRotation operator*(const JacobiRotation& other) const {            // real scalars: conj is the identity
        return JacobiRotation(m_c * other.m_c - m_s * other.m_s, m_c * other.m_s + m_s * other.m_c);
    }
    JacobiRotation transpose() const { return JacobiRotation(m_c, -m_s); }
    // the rotation that diagonalises the symmetric 2x2 matrix [x y; y z]
    bool makeJacobi(const Scalar& x, const Scalar& y, const Scalar& z) {
        Scalar deno = Scalar(2) * std::abs(y);
        if (deno < (std::numeric_limits<Scalar>::min)()) { m_c = Scalar(1); m_s = Scalar(0); return false; }
        Scalar tau = (x - z) / deno;
        Scalar w = std::sqrt(tau * tau + Scalar(1));
        Scalar t;
        if (tau > Scalar(0)) t = Scalar(1) / (tau + w);
        else t = Scalar(1) / (tau - w);
        Scalar sign_t = t > Scalar(0) ? Scalar(1) : Scalar(-1);
        Scalar n = Scalar(1) / std::sqrt(t * t + Scalar(1));
        m_s = -sign_t * (y / std::abs(y)) * std::abs(t) * n;
        m_c = n;
        return true;
    }
};
// apply_rotation_in_the_plane over n strided coefficients
template <typename Scalar> inline void apply_rotation_in_the_plane(Scalar* x, int incx, Scalar* y, int incy, int n, const JacobiRotation<Scalar>& j) {
    const Scalar c = j.c(), s = j.s();
    if (c == Scalar(1) && s == Scalar(0)) return;
    for (int i = 0; i < n; i++) {
        const Scalar xi = *x, yi = *y;
        *x = c * xi + s * yi;
        *y = -s * xi + c * yi;
        x += incx; y += incy;
    }
}
template <typename Scalar, int N> inline void applyOnTheLeft(SmallMatrix<Scalar, N, N>& M, int p, int q, const JacobiRotation<Scalar>& j) {
    apply_rotation_in_the_plane(&M.m[p][0], 1, &M.m[q][0], 1, N, j);                       // rows p and q
}
template <typename Scalar, int N> inline void applyOnTheRight(SmallMatrix<Scalar, N, N>& M, int p, int q, const JacobiRotation<Scalar>& j) {
    apply_rotation_in_the_plane(&M.m[0][p], N, &M.m[0][q], N, N, j.transpose());           // columns p and q
}
// Eigen/src/misc/RealSvd2x2.h
template <typename Scalar, int N> inline void real_2x2_jacobi_svd(const SmallMatrix<Scalar, N, N>& matrix, int p, int q, JacobiRotation<Scalar>* j_left, JacobiRotation<Scalar>* j_right) {
    SmallMatrix<Scalar, 2, 2> m;
    m(0, 0) = matrix(p, p); m(0, 1) = matrix(p, q); m(1, 0) = matrix(q, p); m(1, 1) = matrix(q, q);
    JacobiRotation<Scalar> rot1;
    Scalar t = m(0, 0) + m(1, 1);
    Scalar d = m(1, 0) - m(0, 1);
    if (std::abs(d) < (std::numeric_limits<Scalar>::min)()) { rot1.s() = Scalar(0); rot1.c() = Scalar(1); }
    else {
        Scalar u = t / d;
        Scalar tmp = std::sqrt(Scalar(1) + u * u);
        rot1.s() = Scalar(1) / tmp;
        rot1.c() = u / tmp;
    }
    applyOnTheLeft(m, 0, 1, rot1);
    j_right->makeJacobi(m(0, 0), m(0, 1), m(1, 1));
    *j_left = rot1 * j_right->transpose();
}

template <typename MatrixType> class JacobiSVD;
template <> class JacobiSVD<Matrix4f> {
    struct MatV : Matrix4f { Vector4f col(int j) const { Vector4f v; for (int i = 0; i < 4; i++) v.m[i][0] = m[i][j]; return v; } };
    MatV m_matrixV;
    float m_singularValues[4];
public:
    JacobiSVD(const Matrix4f& matrix, unsigned int computationOptions) { (void)computationOptions; compute(matrix); }
    const MatV& matrixV() const { return m_matrixV; }
    float singularValue(int i) const { return m_singularValues[i]; }
    void compute(const Matrix4f& matrix) {
        typedef float RealScalar;
        const int n = 4;
        const RealScalar precision = RealScalar(2) * std::numeric_limits<RealScalar>::epsilon();
        const RealScalar considerAsZero = (std::numeric_limits<RealScalar>::min)();
        RealScalar scale = 0;
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) scale = std::max(scale, std::abs(matrix(i, j)));
        if (scale == RealScalar(0)) scale = RealScalar(1);
        Matrix4f W;
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) W(i, j) = matrix(i, j) / scale;
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) m_matrixV(i, j) = i == j ? 1.f : 0.f;
        RealScalar maxDiagEntry = 0;
        for (int i = 0; i < n; i++) maxDiagEntry = std::max(maxDiagEntry, std::abs(W(i, i)));
        bool finished = false;
        while (!finished) {
            finished = true;
            for (int p = 1; p < n; ++p)
                for (int q = 0; q < p; ++q) {
                    RealScalar threshold = std::max(considerAsZero, precision * maxDiagEntry);
                    if (std::abs(W(p, q)) > threshold || std::abs(W(q, p)) > threshold) {
                        finished = false;
                        JacobiRotation<RealScalar> j_left, j_right;
                        real_2x2_jacobi_svd(W, p, q, &j_left, &j_right);
                        applyOnTheLeft(W, p, q, j_left);
                        applyOnTheRight(W, p, q, j_right);
                        applyOnTheRight(static_cast<Matrix4f&>(m_matrixV), p, q, j_right);
                        maxDiagEntry = std::max(maxDiagEntry, std::max(std::abs(W(p, p)), std::abs(W(q, q))));
                    }
                }
        }
        for (int i = 0; i < n; ++i) m_singularValues[i] = std::abs(W(i, i));           // (the sign goes into U, which is not computed)
        for (int i = 0; i < n; ++i) m_singularValues[i] *= scale;
        for (int i = 0; i < n; i++) {
            int pos = 0; RealScalar maxRemainingSingularValue = m_singularValues[i];
            for (int k = 1; k < n - i; k++) if (m_singularValues[i + k] > maxRemainingSingularValue) { maxRemainingSingularValue = m_singularValues[i + k]; pos = k; }
            if (maxRemainingSingularValue == RealScalar(0)) break;
            if (pos) {
                pos += i;
                std::swap(m_singularValues[i], m_singularValues[pos]);
                for (int k = 0; k < n; k++) std::swap(m_matrixV(k, pos), m_matrixV(k, i));
            }
        }
    }
};

// Eigen::Matrix<T, R, C> for the combinations the reference's Frame.h / camera models name
template <typename T, int R, int C> struct MatrixSel { typedef SmallMatrix<T, R, C> type; };
template <> struct MatrixSel<float, 3, 1> { typedef Vector3f type; };
template <> struct MatrixSel<float, 3, 3> { typedef Matrix3f type; };
template <> struct MatrixSel<float, 2, 1> { typedef Vector2f type; };
template <> struct MatrixSel<float, 1, 3> { typedef Vector3f type; };
template <> struct MatrixSel<float, 3, 4> { typedef Matrix34f type; };
template <> struct MatrixSel<float, 4, 1> { typedef Vector4f type; };
#ifdef ORBX_TRACKING_WORLD         // (tracking_world.h: Eigen::Matrix<float, 4, 4, Eigen::RowMajor> m(ptr) of Tracking.cc:1370, declared only)
enum { ColMajor = 0, RowMajor = 1 };
struct RowMajorMatrix4f { explicit RowMajorMatrix4f(const float* data); };
template <typename T, int R, int C, int Opt> struct MatrixSelOpt { typedef typename MatrixSel<T, R, C>::type type; };
template <> struct MatrixSelOpt<float, 4, 4, 1> { typedef RowMajorMatrix4f type; };
template <typename T, int R, int C, int Opt = 0> using Matrix = typename MatrixSelOpt<T, R, C, Opt>::type;
#else
template <typename T, int R, int C> using Matrix = typename MatrixSel<T, R, C>::type;
#endif

}  // namespace Eigen
#endif
