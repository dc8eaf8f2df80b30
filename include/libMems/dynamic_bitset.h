// libMems/dynamic_bitset.h -- mems::bitset_t, the run-time sized bit vector libMems takes from boost::dynamic_bitset<>
// (CompactGappedAlignment's rows, scoreProcrastAlignment.cpp:292-298; the segment filters of bbFilter.cpp:117-140 and
// bbAnalyze.cpp:283,343): the operations the in-tree sources use, on a std::vector<bool>.
#ifndef MAUVE_HIP_DYNAMIC_BITSET_H
#define MAUVE_HIP_DYNAMIC_BITSET_H
#include <cstddef>
#include <vector>
namespace mems {
class bitset_t {
public:
    typedef std::vector<bool>::reference reference;
    typedef size_t size_type;
    static const size_t npos = (size_t)-1;
    bitset_t() {}
    explicit bitset_t(size_t n, bool v = false) : b_(n, v) {}
    size_t size() const { return b_.size(); }
    void resize(size_t n, bool v = false) { b_.resize(n, v); }
    bool empty() const { return b_.empty(); }
    reference operator[](size_t i) { return b_[i]; }
    bool operator[](size_t i) const { return b_[i]; }
    bool test(size_t i) const { return b_.at(i); }
    bitset_t &set(size_t i, bool v = true) { b_.at(i) = v; return *this; }
    bitset_t &set() { b_.assign(b_.size(), true); return *this; }
    bitset_t &reset(size_t i) { b_.at(i) = false; return *this; }
    bitset_t &reset() { b_.assign(b_.size(), false); return *this; }
    bitset_t &flip(size_t i) { b_.at(i) = !b_[i]; return *this; }
    bitset_t &flip() { b_.flip(); return *this; }
    size_t count() const { size_t n = 0; for (bool x : b_) n += x; return n; }
    bool any() const { for (bool x : b_) if (x) return true; return false; }
    bool none() const { return !any(); }
    size_t find_first() const { return find_from(0); }
    size_t find_next(size_t i) const { return find_from(i + 1); }
    void push_back(bool v) { b_.push_back(v); }
    // (not in boost::dynamic_bitset: what CompactGappedAlignment needs to reverse and crop its rows)
    void reverse() { std::vector<bool> r(b_.rbegin(), b_.rend()); b_.swap(r); }
    void erase_range(size_t a, size_t b) { b_.erase(b_.begin() + (std::ptrdiff_t)a, b_.begin() + (std::ptrdiff_t)b); }
    bitset_t &operator|=(const bitset_t &o) { for (size_t i = 0; i < b_.size() && i < o.b_.size(); i++) b_[i] = b_[i] || o.b_[i]; return *this; }
    bitset_t &operator&=(const bitset_t &o) { for (size_t i = 0; i < b_.size(); i++) b_[i] = b_[i] && i < o.b_.size() && o.b_[i]; return *this; }
    bitset_t &operator^=(const bitset_t &o) { for (size_t i = 0; i < b_.size() && i < o.b_.size(); i++) b_[i] = b_[i] != o.b_[i]; return *this; }
    bitset_t &operator-=(const bitset_t &o) { for (size_t i = 0; i < b_.size() && i < o.b_.size(); i++) b_[i] = b_[i] && !o.b_[i]; return *this; }
    bitset_t operator~() const { bitset_t r(*this); r.flip(); return r; }
    bool operator==(const bitset_t &o) const { return b_ == o.b_; }
    bool operator!=(const bitset_t &o) const { return b_ != o.b_; }
    bool operator<(const bitset_t &o) const { return b_ < o.b_; }
private:
    size_t find_from(size_t i) const { for (; i < b_.size(); i++) if (b_[i]) return i; return npos; }
    std::vector<bool> b_;
};
inline bitset_t operator|(bitset_t a, const bitset_t &b) { return a |= b; }
inline bitset_t operator&(bitset_t a, const bitset_t &b) { return a &= b; }
inline bitset_t operator^(bitset_t a, const bitset_t &b) { return a ^= b; }
inline bitset_t operator-(bitset_t a, const bitset_t &b) { return a -= b; }
}  // namespace mems
#endif
