// TEST-ONLY stand-in for lms/math/lookup_table.h (see ../../README.md).
// linearSearch is not pinned by any reference test: THIS build defines it as
// piecewise-linear interpolation over ascending vx, clamped to the end values; an empty table
// returns its argument unchanged (so an unconfigured module uses the measured speed as it is).
#pragma once
#include <vector>
namespace lms { namespace math {
enum class LookupTableOrder { ASC, DESC };
template <typename T, LookupTableOrder ORDER> struct LookupTable {
    std::vector<T> vx, vy;
    T linearSearch(T x) const {
        if (vx.empty() || vx.size() != vy.size()) return x;
        if (x <= vx.front()) return vy.front();
        for (size_t i = 1; i < vx.size(); ++i)
            if (x <= vx[i]) {
                const T t = (x - vx[i - 1]) / (vx[i] - vx[i - 1]);
                return vy[i - 1] + t * (vy[i] - vy[i - 1]);
            }
        return vy.back();
    }
};
}}  // namespace lms::math
