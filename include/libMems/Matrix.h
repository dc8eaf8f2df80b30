// libMems/Matrix.h -- Matrix<T>: the dense row-major table behind NumericMatrix (mauveAligner.cpp:617,798).
#ifndef MAUVE_HIP_MATRIX_H
#define MAUVE_HIP_MATRIX_H
#include <iostream>
#include <sstream>
#include <string>
#include <vector>
namespace mems {
template <class T>
class Matrix {
public:
    Matrix() : rows_(0), cols_(0) {}
    Matrix(unsigned rows, unsigned cols) { init(rows, cols); }
    void init(unsigned rows, unsigned cols) { rows_ = rows; cols_ = cols; d_.assign((size_t)rows * cols, T()); }
    void init(unsigned rows, unsigned cols, const T &v) { rows_ = rows; cols_ = cols; d_.assign((size_t)rows * cols, v); }
    T &operator()(unsigned r, unsigned c) { return d_.at((size_t)r * cols_ + c); }
    const T &operator()(unsigned r, unsigned c) const { return d_.at((size_t)r * cols_ + c); }
    unsigned rows() const { return rows_; }
    unsigned cols() const { return cols_; }
    // one row per line, tab separated (the layout --lcb-stats / the identity matrix output prints)
    void print(std::ostream &os) const
    {
        for (unsigned r = 0; r < rows_; r++) {
            for (unsigned c = 0; c < cols_; c++) os << (c ? "\t" : "") << (*this)(r, c);
            os << '\n';
        }
    }
    void read(std::istream &is)
    {
        std::vector<std::vector<T>> rows; std::string line;
        while (std::getline(is, line)) {
            if (line.empty()) continue;
            std::istringstream ls(line); std::vector<T> r; T v;
            while (ls >> v) r.push_back(v);
            rows.push_back(r);
        }
        init((unsigned)rows.size(), rows.empty() ? 0u : (unsigned)rows[0].size());
        for (unsigned r = 0; r < rows_; r++) for (unsigned c = 0; c < cols_ && c < rows[r].size(); c++) (*this)(r, c) = rows[r][c];
    }
protected:
    unsigned rows_, cols_;
    std::vector<T> d_;
};
}  // namespace mems
#endif
