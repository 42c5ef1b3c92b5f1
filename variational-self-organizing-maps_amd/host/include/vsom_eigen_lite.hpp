// vsom_eigen_lite.hpp -- the small subset of Eigen's dense-vector interface that the public
// signatures of the reference's SOM.hpp / Transformation.hpp / DataSet.hpp use
// (Eigen::VectorXf, VectorXi, ArrayXf, ArrayXi).  Used only when the real <Eigen/Dense> is not
// installed (it is not in the build image); with real Eigen present SOM.hpp includes that instead.
// This is an API shim for the host-side boundary, not an arithmetic engine: all training math
// runs in libvsom_hip.so.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>   // real Eigen pulls it in; apps/main.cpp:43 relies on std::strcmp through SOM.hpp
#include <initializer_list>
#include <ostream>
#include <vector>

namespace Eigen {

using Index = std::ptrdiff_t;

template <typename T>
class Vec {
    std::vector<T> v_;

public:
    using Scalar = T;
    Vec() = default;
    explicit Vec(Index n) : v_((size_t)n) {}
    Vec(Index n, Index /*cols*/) : v_((size_t)n) {}
    Vec(std::initializer_list<T> l) : v_(l) {}

    Index size() const { return (Index)v_.size(); }
    Index rows() const { return (Index)v_.size(); }
    Index cols() const { return 1; }
    T *data() { return v_.data(); }
    const T *data() const { return v_.data(); }
    void resize(Index n) { v_.resize((size_t)n); }
    T &operator()(Index i) { return v_[(size_t)i]; }
    const T &operator()(Index i) const { return v_[(size_t)i]; }
    T &operator[](Index i) { return v_[(size_t)i]; }
    const T &operator[](Index i) const { return v_[(size_t)i]; }

    Vec &setZero() { for (auto &x : v_) x = T(0); return *this; }
    Vec &setOnes() { for (auto &x : v_) x = T(1); return *this; }
    Vec &setConstant(T c) { for (auto &x : v_) x = c; return *this; }
    static Vec Zero(Index n) { Vec r(n); r.setZero(); return r; }
    static Vec Ones(Index n) { Vec r(n); r.setOnes(); return r; }
    static Vec Constant(Index n, T c) { Vec r(n); r.setConstant(c); return r; }
    static Vec Random(Index n)
    {
        Vec r(n);
        for (auto &x : r.v_) x = T(2.0 * std::rand() / RAND_MAX - 1.0);
        return r;
    }

    // Eigen's array()/matrix() views are the identity here
    Vec &array() { return *this; }
    const Vec &array() const { return *this; }
    Vec &matrix() { return *this; }
    const Vec &matrix() const { return *this; }
    const Vec &transpose() const { return *this; }

    template <typename U>
    Vec<U> cast() const
    {
        Vec<U> r(size());
        for (Index i = 0; i < size(); ++i) r[i] = (U)v_[(size_t)i];
        return r;
    }
    Vec head(Index n) const { Vec r(n); for (Index i = 0; i < n; ++i) r[i] = v_[(size_t)i]; return r; }
    Vec tail(Index n) const { Vec r(n); for (Index i = 0; i < n; ++i) r[i] = v_[v_.size() - (size_t)n + (size_t)i]; return r; }

    T dot(const Vec &o) const { T s = T(0); for (Index i = 0; i < size(); ++i) s += v_[(size_t)i] * o[i]; return s; }
    T squaredNorm() const { return dot(*this); }
    T sum() const { T s = T(0); for (auto x : v_) s += x; return s; }
    Vec sign() const { Vec r(size()); for (Index i = 0; i < size(); ++i) { T a = v_[(size_t)i]; r[i] = (a != a) ? a : T((a > T(0)) - (a < T(0))); } return r; }
    Vec sqrt() const { Vec r(size()); for (Index i = 0; i < size(); ++i) r[i] = (T)std::sqrt((double)v_[(size_t)i]); return r; }
    Vec abs() const { Vec r(size()); for (Index i = 0; i < size(); ++i) r[i] = v_[(size_t)i] < T(0) ? -v_[(size_t)i] : v_[(size_t)i]; return r; }

    Vec &operator+=(const Vec &o) { for (Index i = 0; i < size(); ++i) v_[(size_t)i] += o[i]; return *this; }
    Vec &operator-=(const Vec &o) { for (Index i = 0; i < size(); ++i) v_[(size_t)i] -= o[i]; return *this; }
    Vec &operator*=(T s) { for (auto &x : v_) x *= s; return *this; }
    bool operator==(const Vec &o) const { return v_ == o.v_; }

    // comma initialiser: v << a, b, c;
    struct Comma {
        Vec &v; Index i;
        Comma &operator,(T x) { v[i++] = x; return *this; }
    };
    Comma operator<<(T x) { v_[0] = x; return Comma{*this, 1}; }
};

template <typename T> Vec<T> operator+(Vec<T> a, const Vec<T> &b) { a += b; return a; }
template <typename T> Vec<T> operator-(Vec<T> a, const Vec<T> &b) { a -= b; return a; }
template <typename T> Vec<T> operator*(Vec<T> a, T s) { a *= s; return a; }
template <typename T> Vec<T> operator*(T s, Vec<T> a) { a *= s; return a; }
template <typename T> Vec<T> operator*(const Vec<T> &a, const Vec<T> &b) { Vec<T> r(a.size()); for (Index i = 0; i < a.size(); ++i) r[i] = a[i] * b[i]; return r; }
template <typename T> Vec<T> operator/(Vec<T> a, T s) { for (Index i = 0; i < a.size(); ++i) a[i] /= s; return a; }
template <typename T> std::ostream &operator<<(std::ostream &os, const Vec<T> &v)
{
    for (Index i = 0; i < v.size(); ++i) os << (i ? " " : "") << v[i];
    return os;
}

using VectorXf = Vec<float>;
using ArrayXf = Vec<float>;
using VectorXi = Vec<int>;
using ArrayXi = Vec<int>;
using VectorXd = Vec<double>;

}   // namespace Eigen
