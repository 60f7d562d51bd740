// tests/host_shims/spill_split_shim.cpp — C entry for tests/test_spill_split_cpu.py: the out-of-core merge's key-range planner as it is
// (spades_amd/csrc/smx_spill_split.hpp, host-only), compiled with g++.
#include "../../spades_amd/csrc/smx_spill_split.hpp"

extern "C" int spill_split_plan(const char *const *ptrs, const uint64_t *ns, unsigned nruns, unsigned nw, uint64_t max_part, uint64_t *cuts_out,
                                unsigned cuts_cap /* rows */) {
    std::vector<smx_split::Slice> runs(nruns);
    for (unsigned r = 0; r < nruns; ++r) runs[r] = {ptrs[r], ns[r]};
    const auto cuts = smx_split::plan(runs, nw, max_part);
    if (cuts.size() > cuts_cap) return -(int)cuts.size();
    for (size_t p = 0; p < cuts.size(); ++p)
        for (unsigned r = 0; r < nruns; ++r) cuts_out[p * nruns + r] = cuts[p][r];
    return (int)cuts.size();
}
