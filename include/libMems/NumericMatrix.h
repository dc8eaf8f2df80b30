// libMems/NumericMatrix.h -- NumericMatrix<T>: Matrix<T> with element-wise arithmetic (mauveAligner.cpp:617,798;
// calculateBackboneCoverage.cpp:106-127).
#ifndef MAUVE_HIP_NUMERICMATRIX_H
#define MAUVE_HIP_NUMERICMATRIX_H
#include "Matrix.h"
namespace mems {
template <class T>
class NumericMatrix : public Matrix<T> {
public:
    NumericMatrix() {}
    NumericMatrix(unsigned rows, unsigned cols) : Matrix<T>(rows, cols) {}
    NumericMatrix &operator+=(const NumericMatrix &o) { for (size_t i = 0; i < this->d_.size() && i < o.d_.size(); i++) this->d_[i] += o.d_[i]; return *this; }
    NumericMatrix &operator-=(const NumericMatrix &o) { for (size_t i = 0; i < this->d_.size() && i < o.d_.size(); i++) this->d_[i] -= o.d_[i]; return *this; }
    NumericMatrix &operator*=(const T &k) { for (T &v : this->d_) v *= k; return *this; }
    NumericMatrix &operator/=(const T &k) { for (T &v : this->d_) v /= k; return *this; }
    void identity() { for (unsigned r = 0; r < this->rows_; r++) for (unsigned c = 0; c < this->cols_; c++) (*this)(r, c) = r == c ? T(1) : T(0); }
};
}  // namespace mems
#endif
