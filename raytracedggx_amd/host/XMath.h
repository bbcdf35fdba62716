// Minimal fp32 row-vector matrix math for the host side of the path: the handful of DirectXMath
// calls the reference makes in RayTracedGGX.cpp:262-277 and RayTracer.cpp:250-305
// (XMMatrixLookAtLH, XMMatrixPerspectiveFovLH, XMMatrixRotationY, XMMatrixScaling,
// XMMatrixTranslation, XMMatrixMultiply, XMMatrixInverse, XMMatrixTranspose, XMStoreFloat3x4).
// DirectXMath itself is not part of the reference tree; semantics follow its public definitions:
// row-major storage, v' = v * M, left-handed.  Trigonometry is evaluated in double and rounded once;
// the inverse is evaluated in double and rounded once.
#pragma once
#include <cmath>
#include <cstring>

namespace xm {

struct Float3 { float x, y, z; };
struct Float4 { float x, y, z, w; };
struct Matrix { float r[4][4]; };

inline Matrix Identity() { Matrix m{}; m.r[0][0] = m.r[1][1] = m.r[2][2] = m.r[3][3] = 1.0f; return m; }
inline Matrix Multiply(const Matrix& A, const Matrix& B) {
  Matrix C;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      float s = A.r[i][0] * B.r[0][j];
      s = s + A.r[i][1] * B.r[1][j];
      s = s + A.r[i][2] * B.r[2][j];
      s = s + A.r[i][3] * B.r[3][j];
      C.r[i][j] = s;
    }
  return C;
}
inline Matrix operator*(const Matrix& A, const Matrix& B) { return Multiply(A, B); }
inline Matrix Transpose(const Matrix& A) { Matrix T; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) T.r[i][j] = A.r[j][i]; return T; }
inline Matrix Scaling(float x, float y, float z) { Matrix m = Identity(); m.r[0][0] = x; m.r[1][1] = y; m.r[2][2] = z; return m; }
inline Matrix Translation(float x, float y, float z) { Matrix m = Identity(); m.r[3][0] = x; m.r[3][1] = y; m.r[3][2] = z; return m; }
inline Matrix RotationY(float angle) {
  const float s = (float)std::sin((double)angle), c = (float)std::cos((double)angle);
  Matrix m = Identity();
  m.r[0][0] = c; m.r[0][2] = -s; m.r[2][0] = s; m.r[2][2] = c;
  return m;
}
// XMMatrixRotationRollPitchYaw(pitch, yaw, roll): roll about z first, then pitch about x, then yaw about y (row vectors)
inline Matrix RotationRollPitchYaw(float pitch, float yaw, float roll) {
  const float cp = (float)std::cos((double)pitch), sp = (float)std::sin((double)pitch);
  const float cy = (float)std::cos((double)yaw), sy = (float)std::sin((double)yaw);
  const float cr = (float)std::cos((double)roll), sr = (float)std::sin((double)roll);
  Matrix m = Identity();
  m.r[0][0] = cr * cy + sr * sp * sy; m.r[0][1] = sr * cp; m.r[0][2] = sr * sp * cy - cr * sy;
  m.r[1][0] = cr * sp * sy - sr * cy; m.r[1][1] = cr * cp; m.r[1][2] = sr * sy + cr * sp * cy;
  m.r[2][0] = cp * sy; m.r[2][1] = -sp; m.r[2][2] = cp * cy;
  return m;
}
inline float Dot(Float3 a, Float3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline Float3 Cross(Float3 a, Float3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline Float3 Sub(Float3 a, Float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Float3 Normalize(Float3 v) { const float inv = 1.0f / std::sqrt(Dot(v, v)); return {v.x * inv, v.y * inv, v.z * inv}; }
inline Matrix LookAtLH(Float3 eye, Float3 focus, Float3 up) {
  const Float3 zAxis = Normalize(Sub(focus, eye));
  const Float3 xAxis = Normalize(Cross(up, zAxis));
  const Float3 yAxis = Cross(zAxis, xAxis);
  Matrix m = Identity();
  m.r[0][0] = xAxis.x; m.r[1][0] = xAxis.y; m.r[2][0] = xAxis.z; m.r[3][0] = -Dot(xAxis, eye);
  m.r[0][1] = yAxis.x; m.r[1][1] = yAxis.y; m.r[2][1] = yAxis.z; m.r[3][1] = -Dot(yAxis, eye);
  m.r[0][2] = zAxis.x; m.r[1][2] = zAxis.y; m.r[2][2] = zAxis.z; m.r[3][2] = -Dot(zAxis, eye);
  return m;
}
inline Matrix PerspectiveFovLH(float fovY, float aspect, float zn, float zf) {
  const double half = 0.5 * (double)fovY;
  const float sinFov = (float)std::sin(half), cosFov = (float)std::cos(half);
  const float height = cosFov / sinFov, width = height / aspect, range = zf / (zf - zn);
  Matrix m{};
  m.r[0][0] = width; m.r[1][1] = height; m.r[2][2] = range; m.r[2][3] = 1.0f; m.r[3][2] = -range * zn;
  return m;
}
// General inverse: adjugate / determinant in double precision, one rounding to fp32 per element.
inline Matrix Inverse(const Matrix& A) {
  double m[16], inv[16];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m[i * 4 + j] = (double)A.r[i][j];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  const double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  const double rdet = 1.0 / det;
  Matrix R;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) R.r[i][j] = (float)(inv[i * 4 + j] * rdet);
  return R;
}
// XMStoreFloat4x4(dst, M): row-major copy.
inline void StoreFloat4x4(float* dst, const Matrix& M) { std::memcpy(dst, M.r, 64); }
// XMStoreFloat3x4(dst, M): the first three rows of the transpose (count floats, 12 normally).
inline void StoreFloat3x4(float* dst, const Matrix& M, int count = 12) { for (int k = 0; k < count; ++k) dst[k] = M.r[k % 4][k / 4]; }

}  // namespace xm
