// Build-recipe stub (NOT reference source): stands in for the header the reference's
// cmake generates from src/include/k_range.hpp.in with its default cache values
// SPADES_MIN_K=1, SPADES_MAX_K=128 (src/cmake/options.cmake:55-56).
#ifndef K_RANGE_HPP_
#define K_RANGE_HPP_
#include <cstdlib>
namespace runtime_k {
const size_t MIN_K = 1;
const size_t MAX_K = 128;
}
#endif
