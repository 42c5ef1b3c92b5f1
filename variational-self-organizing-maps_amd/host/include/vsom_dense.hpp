// Picks the dense-vector types of the public API: real Eigen when installed, else the shim.
#pragma once
#if __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
#elif __has_include("Eigen/Dense")
#include "Eigen/Dense"
#else
#include "vsom_eigen_lite.hpp"
#endif
