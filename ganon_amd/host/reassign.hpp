// reassign.hpp -- `ganon reassign`: the file side of /root/reference/src/ganon/reassign.py; the EM itself
// runs on the device behind gn_reassign_* (include/ganon_hip.h).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace gnhost
{

struct ReassignConfig
{
    std::vector<std::string> input_prefix;
    std::string              output_prefix;
    uint32_t                 max_iter  = 10; // /root/reference/src/ganon/config.py:770-786
    double                   threshold = 0;
    bool                     remove_all = false, skip_one = false, skip_rep = false, verbose = false, quiet = false;
    int                      device = 0;
};

// reassign.py:8-223; false where the reference returns False.  Throws std::runtime_error where the reference's Python
// raises (a line of .all that is not three fields, a count that is no integer, an unwritable output).
bool run_reassign(const ReassignConfig& cfg);

// str(round(x, 6)) of Python 3, the form the iteration lines of the log use (reassign.py:131-138)
std::string py_round6_repr(double x);

} // namespace gnhost
