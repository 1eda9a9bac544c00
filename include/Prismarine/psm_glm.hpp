// psm_glm.hpp -- the few glm types/functions the Prismarine host API mentions in its signatures, and the ones the reference's
// viewer builds its node transforms with (Source/Examples/Viewer.cpp:240-258: make_mat4 / make_vec3 / make_quat, translate,
// scale, mat4_cast on doubles).
// If a real glm is on the include path it is used; otherwise this minimal stand-in (column-major,
// glm conventions: m[col][row], perspective/lookAt right-handed, depth -1..1) keeps the headers
// self-contained.  Host-side convenience only: kernels never see these types.
#pragma once
#if defined(__has_include)
#if __has_include(<glm/glm.hpp>) && !defined(PSM_NO_SYSTEM_GLM)
#include <glm/glm.hpp>
#include <glm/gtc/matrix_transform.hpp>
#include <glm/gtc/type_ptr.hpp>
#include <glm/gtc/quaternion.hpp>
#define PSM_HAVE_GLM 1
#endif
#endif

#ifndef PSM_HAVE_GLM
#include <cmath>
#include <cstddef>

namespace glm {

template <typename T> struct tvec2 { T x, y; tvec2() : x(0), y(0) {} tvec2(T a, T b) : x(a), y(b) {} };
template <typename T> struct tvec3 {
    T x, y, z;
    tvec3() : x(0), y(0), z(0) {}
    explicit tvec3(T s) : x(s), y(s), z(s) {}
    tvec3(T a, T b, T c) : x(a), y(b), z(c) {}
    template <typename U> tvec3(const tvec3<U>& o) : x(T(o.x)), y(T(o.y)), z(T(o.z)) {}
    T& operator[](int i) { return (&x)[i]; }
    const T& operator[](int i) const { return (&x)[i]; }
};
template <typename T> struct tvec4 {
    T x, y, z, w;
    tvec4() : x(0), y(0), z(0), w(0) {}
    explicit tvec4(T s) : x(s), y(s), z(s), w(s) {}
    tvec4(T a, T b, T c, T d) : x(a), y(b), z(c), w(d) {}
    tvec4(const tvec3<T>& v, T d) : x(v.x), y(v.y), z(v.z), w(d) {}
    T& operator[](int i) { return (&x)[i]; }
    const T& operator[](int i) const { return (&x)[i]; }
};
template <typename T> inline tvec3<T> operator+(const tvec3<T>& a, const tvec3<T>& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> inline tvec3<T> operator-(const tvec3<T>& a, const tvec3<T>& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> inline tvec3<T> operator*(const tvec3<T>& a, T s) { return {a.x * s, a.y * s, a.z * s}; }
template <typename T> inline tvec3<T> operator/(const tvec3<T>& a, T s) { return {a.x / s, a.y / s, a.z / s}; }
template <typename T> inline T dot(const tvec3<T>& a, const tvec3<T>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> inline tvec3<T> cross(const tvec3<T>& a, const tvec3<T>& b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
template <typename T> inline tvec3<T> normalize(const tvec3<T>& a) { return a * (T(1) / std::sqrt(dot(a, a))); }

template <typename T> struct tmat4 {
    tvec4<T> c[4];
    tmat4() : tmat4(T(1)) {}
    explicit tmat4(T d) { for (int i = 0; i < 4; i++) c[i] = tvec4<T>(T(0)); c[0].x = c[1].y = c[2].z = c[3].w = d; }
    template <typename U> explicit tmat4(const tmat4<U>& o) { for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) c[i][j] = T(o.c[i][j]); }
    tvec4<T>& operator[](int i) { return c[i]; }
    const tvec4<T>& operator[](int i) const { return c[i]; }
};
template <typename T> inline tmat4<T> operator*(const tmat4<T>& a, const tmat4<T>& b) {
    tmat4<T> r(T(0));
    for (int col = 0; col < 4; col++) for (int row = 0; row < 4; row++) { T s = 0; for (int k = 0; k < 4; k++) s += a[k][row] * b[col][k]; r[col][row] = s; }
    return r;
}
template <typename T> inline tmat4<T>& operator*=(tmat4<T>& a, const tmat4<T>& b) { a = a * b; return a; }
template <typename T> inline tmat4<T> transpose(const tmat4<T>& m) { tmat4<T> r(T(0)); for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r[i][j] = m[j][i]; return r; }
template <typename T> inline tvec4<T> operator*(const tvec4<T>& a, const tvec4<T>& b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
template <typename T> inline tvec4<T> operator+(const tvec4<T>& a, const tvec4<T>& b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
template <typename T> inline tvec4<T> operator-(const tvec4<T>& a, const tvec4<T>& b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
template <typename T> inline tvec4<T> operator*(const tvec4<T>& a, T s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
// the operation order of glm's own 4x4 inverse (cofactors in six 2x2 groups, determinant from column 0), so that the
// matrices the host layer uploads are bit for bit glm's (tests/golden/glm_host_formulas.npz)
template <typename T> inline tmat4<T> inverse(const tmat4<T>& m) {
    T c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3], c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3], c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    T c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3], c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3], c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    T c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2], c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2], c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    T c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3], c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3], c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    T c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2], c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2], c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    T c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1], c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1], c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    tvec4<T> f0(c00, c00, c02, c03), f1(c04, c04, c06, c07), f2(c08, c08, c10, c11), f3(c12, c12, c14, c15), f4(c16, c16, c18, c19), f5(c20, c20, c22, c23);
    tvec4<T> v0(m[1][0], m[0][0], m[0][0], m[0][0]), v1(m[1][1], m[0][1], m[0][1], m[0][1]), v2(m[1][2], m[0][2], m[0][2], m[0][2]), v3(m[1][3], m[0][3], m[0][3], m[0][3]);
    tvec4<T> i0(v1 * f0 - v2 * f1 + v3 * f2), i1(v0 * f0 - v2 * f3 + v3 * f4), i2(v0 * f1 - v1 * f3 + v3 * f5), i3(v0 * f2 - v1 * f4 + v2 * f5);
    tvec4<T> sa(T(1), T(-1), T(1), T(-1)), sb(T(-1), T(1), T(-1), T(1));
    tmat4<T> inv(T(0));
    inv[0] = i0 * sa; inv[1] = i1 * sb; inv[2] = i2 * sa; inv[3] = i3 * sb;
    tvec4<T> row0(inv[0][0], inv[1][0], inv[2][0], inv[3][0]);
    tvec4<T> d0(m[0] * row0);
    T one_over_det = T(1) / ((d0.x + d0.y) + (d0.z + d0.w));
    for (int c = 0; c < 4; c++) inv[c] = inv[c] * one_over_det;
    return inv;
}
template <typename T> inline tmat4<T> translate(const tvec3<T>& v) { tmat4<T> r(T(1)); r[3].x = v.x; r[3].y = v.y; r[3].z = v.z; return r; }
template <typename T> inline tmat4<T> scale(const tvec3<T>& v) { tmat4<T> r(T(1)); r[0].x = v.x; r[1].y = v.y; r[2].z = v.z; return r; }
template <typename T> inline tmat4<T> lookAt(const tvec3<T>& eye, const tvec3<T>& center, const tvec3<T>& up) {
    tvec3<T> f = normalize(center - eye), s = normalize(cross(f, up)), u = cross(s, f);
    tmat4<T> r(T(1));
    r[0][0] = s.x; r[1][0] = s.y; r[2][0] = s.z;
    r[0][1] = u.x; r[1][1] = u.y; r[2][1] = u.z;
    r[0][2] = -f.x; r[1][2] = -f.y; r[2][2] = -f.z;
    r[3][0] = -dot(s, eye); r[3][1] = -dot(u, eye); r[3][2] = dot(f, eye);
    return r;
}
template <typename T> inline tmat4<T> perspective(T fovy, T aspect, T zn, T zf) {
    T t = std::tan(fovy / T(2));
    tmat4<T> r(T(0));
    r[0][0] = T(1) / (aspect * t); r[1][1] = T(1) / t; r[2][2] = -(zf + zn) / (zf - zn); r[2][3] = -T(1);
    r[3][2] = -(T(2) * zf * zn) / (zf - zn);
    return r;
}
// glm/gtc/type_ptr.hpp make_*: copies of the memory (a matrix column by column, a quaternion as x, y, z, w)
template <typename T> inline tmat4<T> make_mat4(const T* p) { tmat4<T> r(T(0)); for (int c = 0; c < 4; c++) for (int j = 0; j < 4; j++) r[c][j] = p[4 * c + j]; return r; }
template <typename T> inline tvec3<T> make_vec3(const T* p) { return tvec3<T>(p[0], p[1], p[2]); }
template <typename T> struct tquat { T x, y, z, w; };
template <typename T> inline tquat<T> make_quat(const T* p) { return tquat<T>{p[0], p[1], p[2], p[3]}; }
// glm/gtc/quaternion.inl mat3_cast / mat4_cast, in its operation order (tests/golden/glm_gltf_transforms.npz)
template <typename T> inline tmat4<T> mat4_cast(const tquat<T>& q) {
    tmat4<T> r(T(1));
    T qxx(q.x * q.x), qyy(q.y * q.y), qzz(q.z * q.z), qxz(q.x * q.z), qxy(q.x * q.y), qyz(q.y * q.z), qwx(q.w * q.x), qwy(q.w * q.y), qwz(q.w * q.z);
    r[0][0] = T(1) - T(2) * (qyy + qzz); r[0][1] = T(2) * (qxy + qwz); r[0][2] = T(2) * (qxz - qwy);
    r[1][0] = T(2) * (qxy - qwz); r[1][1] = T(1) - T(2) * (qxx + qzz); r[1][2] = T(2) * (qyz + qwx);
    r[2][0] = T(2) * (qxz + qwy); r[2][1] = T(2) * (qyz - qwx); r[2][2] = T(1) - T(2) * (qxx + qyy);
    return r;
}
template <typename T> inline T pi() { return T(3.14159265358979323846264338327950288); }
template <typename T> inline const T* value_ptr(const tmat4<T>& m) { return &m.c[0].x; }
template <typename T> inline const T* value_ptr(const tvec4<T>& v) { return &v.x; }

typedef tvec2<float> vec2; typedef tvec3<float> vec3; typedef tvec4<float> vec4;
typedef tvec2<int> ivec2; typedef tvec4<int> ivec4; typedef tvec4<unsigned> uvec4;
typedef tvec3<double> dvec3; typedef tmat4<float> mat4; typedef tmat4<double> dmat4; typedef tquat<float> quat; typedef tquat<double> dquat;

}  // namespace glm
#endif  // !PSM_HAVE_GLM
