// Build-recipe stub (NOT reference source): stands in for the cmake-generated
// src/include/config.hpp.in with every optional allocator / debug switch left undefined.
#ifndef __SPADES_CONFIG_HPP__
#define __SPADES_CONFIG_HPP__
#endif
