// Small dense vector / matrix value types for the host-side mirror of the reference API.
//
// The reference spells its containers Eigen::VectorXf / MatrixXf / Vector2f
// (include/ilqgames/utils/types.h:62-74).  Eigen is not part of this product: the numerics run on
// the GPU through the C ABI (include/ilqg.h); the host only needs containers with the same storage
// order (column-major, so a block can be memcpy'd into the trajectory-major device buffers) and the
// handful of accessors that problem-definition code uses (Zero, operator(), size, head/segment,
// norm, x()/y()).  Nothing here is a linear-algebra solver.
#ifndef ILQGAMES_HOST_LINALG_HPP_
#define ILQGAMES_HOST_LINALG_HPP_

#include <cmath>
#include <cstddef>
#include <initializer_list>
#include <ostream>
#include <vector>

namespace ilqgames {
namespace host {

template <typename S>
class Matrix;

template <typename S>
class Vector {
 public:
  Vector() {}
  explicit Vector(std::ptrdiff_t n) : v_(static_cast<size_t>(n)) {}
  Vector(std::initializer_list<S> init) : v_(init) {}

  static Vector Zero(std::ptrdiff_t n) { return Constant(n, S(0)); }
  static Vector Constant(std::ptrdiff_t n, S value) {
    Vector out(n);
    for (auto& e : out.v_) e = value;
    return out;
  }

  std::ptrdiff_t size() const { return static_cast<std::ptrdiff_t>(v_.size()); }
  std::ptrdiff_t rows() const { return size(); }
  std::ptrdiff_t cols() const { return 1; }
  void resize(std::ptrdiff_t n) { v_.resize(static_cast<size_t>(n)); }
  void setZero() { for (auto& e : v_) e = S(0); }
  void setConstant(S value) { for (auto& e : v_) e = value; }

  S& operator()(std::ptrdiff_t i) { return v_[static_cast<size_t>(i)]; }
  const S& operator()(std::ptrdiff_t i) const { return v_[static_cast<size_t>(i)]; }
  S& operator[](std::ptrdiff_t i) { return v_[static_cast<size_t>(i)]; }
  const S& operator[](std::ptrdiff_t i) const { return v_[static_cast<size_t>(i)]; }
  S* data() { return v_.data(); }
  const S* data() const { return v_.data(); }

  // Copies (problem-definition code only reads these).
  Vector segment(std::ptrdiff_t start, std::ptrdiff_t n) const {
    Vector out(n);
    for (std::ptrdiff_t i = 0; i < n; i++) out(i) = (*this)(start + i);
    return out;
  }
  Vector head(std::ptrdiff_t n) const { return segment(0, n); }
  Vector tail(std::ptrdiff_t n) const { return segment(size() - n, n); }
  void set_segment(std::ptrdiff_t start, const Vector& src) {
    for (std::ptrdiff_t i = 0; i < src.size(); i++) (*this)(start + i) = src(i);
  }

  S dot(const Vector& o) const {
    S acc = S(0);
    for (std::ptrdiff_t i = 0; i < size(); i++) acc += (*this)(i) * o(i);
    return acc;
  }
  S squaredNorm() const { return dot(*this); }
  S norm() const { return std::sqrt(squaredNorm()); }
  S sum() const {
    S acc = S(0);
    for (const auto& e : v_) acc += e;
    return acc;
  }
  S maxCoeff() const {
    S best = v_.empty() ? S(0) : v_[0];
    for (const auto& e : v_) best = e > best ? e : best;
    return best;
  }
  bool isApprox(const Vector& o, S tol = S(1e-5)) const {
    if (size() != o.size()) return false;
    return (*this - o).squaredNorm() <= tol * tol * std::fmin(squaredNorm(), o.squaredNorm());
  }

  Vector& operator+=(const Vector& o) {
    for (std::ptrdiff_t i = 0; i < size(); i++) (*this)(i) += o(i);
    return *this;
  }
  Vector& operator-=(const Vector& o) {
    for (std::ptrdiff_t i = 0; i < size(); i++) (*this)(i) -= o(i);
    return *this;
  }
  Vector& operator*=(S s) {
    for (auto& e : v_) e *= s;
    return *this;
  }
  Vector& operator/=(S s) {
    for (auto& e : v_) e /= s;
    return *this;
  }
  friend Vector operator+(Vector a, const Vector& b) { return a += b; }
  friend Vector operator-(Vector a, const Vector& b) { return a -= b; }
  friend Vector operator-(Vector a) { return a *= S(-1); }
  friend Vector operator*(Vector a, S s) { return a *= s; }
  friend Vector operator*(S s, Vector a) { return a *= s; }
  friend Vector operator/(Vector a, S s) { return a /= s; }
  friend bool operator==(const Vector& a, const Vector& b) { return a.v_ == b.v_; }
  friend bool operator!=(const Vector& a, const Vector& b) { return !(a == b); }
  friend std::ostream& operator<<(std::ostream& os, const Vector& a) {
    for (std::ptrdiff_t i = 0; i < a.size(); i++) os << (i ? " " : "") << a(i);
    return os;
  }

 private:
  std::vector<S> v_;
};

// Column-major, like Eigen's default, so data() can be copied straight into the [n*n] / [m*n]
// blocks of the device layouts in include/ilqg.h.
template <typename S>
class Matrix {
 public:
  Matrix() : r_(0), c_(0) {}
  Matrix(std::ptrdiff_t rows, std::ptrdiff_t cols)
      : r_(rows), c_(cols), v_(static_cast<size_t>(rows * cols)) {}

  static Matrix Zero(std::ptrdiff_t rows, std::ptrdiff_t cols) {
    Matrix out(rows, cols);
    out.setZero();
    return out;
  }
  static Matrix Identity(std::ptrdiff_t rows, std::ptrdiff_t cols) {
    Matrix out = Zero(rows, cols);
    for (std::ptrdiff_t i = 0; i < rows && i < cols; i++) out(i, i) = S(1);
    return out;
  }

  std::ptrdiff_t rows() const { return r_; }
  std::ptrdiff_t cols() const { return c_; }
  std::ptrdiff_t size() const { return r_ * c_; }
  void resize(std::ptrdiff_t rows, std::ptrdiff_t cols) {
    r_ = rows;
    c_ = cols;
    v_.assign(static_cast<size_t>(rows * cols), S(0));
  }
  void setZero() { for (auto& e : v_) e = S(0); }
  void setIdentity() { *this = Identity(r_, c_); }

  S& operator()(std::ptrdiff_t i, std::ptrdiff_t j) { return v_[static_cast<size_t>(j * r_ + i)]; }
  const S& operator()(std::ptrdiff_t i, std::ptrdiff_t j) const { return v_[static_cast<size_t>(j * r_ + i)]; }
  S* data() { return v_.data(); }
  const S* data() const { return v_.data(); }

  Matrix block(std::ptrdiff_t i0, std::ptrdiff_t j0, std::ptrdiff_t rows, std::ptrdiff_t cols) const {
    Matrix out(rows, cols);
    for (std::ptrdiff_t j = 0; j < cols; j++)
      for (std::ptrdiff_t i = 0; i < rows; i++) out(i, j) = (*this)(i0 + i, j0 + j);
    return out;
  }
  void set_block(std::ptrdiff_t i0, std::ptrdiff_t j0, const Matrix& src) {
    for (std::ptrdiff_t j = 0; j < src.cols(); j++)
      for (std::ptrdiff_t i = 0; i < src.rows(); i++) (*this)(i0 + i, j0 + j) = src(i, j);
  }
  Matrix transpose() const {
    Matrix out(c_, r_);
    for (std::ptrdiff_t j = 0; j < c_; j++)
      for (std::ptrdiff_t i = 0; i < r_; i++) out(j, i) = (*this)(i, j);
    return out;
  }
  S norm() const {
    S acc = S(0);
    for (const auto& e : v_) acc += e * e;
    return std::sqrt(acc);
  }

  Matrix& operator+=(const Matrix& o) {
    for (size_t i = 0; i < v_.size(); i++) v_[i] += o.v_[i];
    return *this;
  }
  Matrix& operator-=(const Matrix& o) {
    for (size_t i = 0; i < v_.size(); i++) v_[i] -= o.v_[i];
    return *this;
  }
  Matrix& operator*=(S s) {
    for (auto& e : v_) e *= s;
    return *this;
  }
  friend Matrix operator+(Matrix a, const Matrix& b) { return a += b; }
  friend Matrix operator-(Matrix a, const Matrix& b) { return a -= b; }
  friend Matrix operator*(Matrix a, S s) { return a *= s; }
  friend Matrix operator*(S s, Matrix a) { return a *= s; }
  friend Matrix operator*(const Matrix& a, const Matrix& b) {
    Matrix out = Zero(a.rows(), b.cols());
    for (std::ptrdiff_t j = 0; j < b.cols(); j++)
      for (std::ptrdiff_t k = 0; k < a.cols(); k++)
        for (std::ptrdiff_t i = 0; i < a.rows(); i++) out(i, j) += a(i, k) * b(k, j);
    return out;
  }
  friend Vector<S> operator*(const Matrix& a, const Vector<S>& x) {
    Vector<S> out = Vector<S>::Zero(a.rows());
    for (std::ptrdiff_t k = 0; k < a.cols(); k++)
      for (std::ptrdiff_t i = 0; i < a.rows(); i++) out(i) += a(i, k) * x(k);
    return out;
  }
  friend bool operator==(const Matrix& a, const Matrix& b) {
    return a.r_ == b.r_ && a.c_ == b.c_ && a.v_ == b.v_;
  }
  friend std::ostream& operator<<(std::ostream& os, const Matrix& a) {
    for (std::ptrdiff_t i = 0; i < a.rows(); i++) {
      for (std::ptrdiff_t j = 0; j < a.cols(); j++) os << (j ? " " : "") << a(i, j);
      os << "\n";
    }
    return os;
  }

 private:
  std::ptrdiff_t r_, c_;
  std::vector<S> v_;
};

// Planar point (the reference's Point2 = Eigen::Vector2f, types.h:68).
class Point2f {
 public:
  Point2f() : x_(0.0f), y_(0.0f) {}
  Point2f(float x, float y) : x_(x), y_(y) {}
  static Point2f Zero() { return Point2f(); }

  float& x() { return x_; }
  float& y() { return y_; }
  float x() const { return x_; }
  float y() const { return y_; }
  float operator()(int i) const { return i == 0 ? x_ : y_; }
  float& operator()(int i) { return i == 0 ? x_ : y_; }

  float dot(const Point2f& o) const { return x_ * o.x_ + y_ * o.y_; }
  float squaredNorm() const { return dot(*this); }
  float norm() const { return std::sqrt(squaredNorm()); }

  Point2f& operator+=(const Point2f& o) { x_ += o.x_; y_ += o.y_; return *this; }
  Point2f& operator-=(const Point2f& o) { x_ -= o.x_; y_ -= o.y_; return *this; }
  Point2f& operator*=(float s) { x_ *= s; y_ *= s; return *this; }
  Point2f& operator/=(float s) { x_ /= s; y_ /= s; return *this; }
  friend Point2f operator+(Point2f a, const Point2f& b) { return a += b; }
  friend Point2f operator-(Point2f a, const Point2f& b) { return a -= b; }
  friend Point2f operator-(Point2f a) { return a *= -1.0f; }
  friend Point2f operator*(Point2f a, float s) { return a *= s; }
  friend Point2f operator*(float s, Point2f a) { return a *= s; }
  friend Point2f operator/(Point2f a, float s) { return a /= s; }
  friend bool operator==(const Point2f& a, const Point2f& b) { return a.x_ == b.x_ && a.y_ == b.y_; }
  friend std::ostream& operator<<(std::ostream& os, const Point2f& p) { return os << p.x_ << " " << p.y_; }

 private:
  float x_, y_;
};

}  // namespace host
}  // namespace ilqgames

#endif  // ILQGAMES_HOST_LINALG_HPP_
