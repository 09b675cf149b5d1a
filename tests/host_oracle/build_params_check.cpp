// build_params_check.cpp -- test harness: runs the PRODUCT's sizing code (ganon_amd/host/build_params.cpp) on counts given
// on stdin and prints what it chose, so that tests/test_build_cpu.py can compare it with oracle/build_params.py.
// stdin:  max_fp filter_size hash_functions mode n  c_0 ... c_{n-1}      (one case per line)
// stdout: n_bins max_hashes_bin hash_functions bin_size_bits max_fp(hex) true_max_fp(hex) true_avg_fp(hex) layout-digest
#include "../../ganon_amd/host/build_params.hpp"

#include <cstdio>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

int main()
{
    std::string line;
    while (std::getline(std::cin, line))
    {
        std::istringstream    in(line);
        double                max_fp, filter_size;
        unsigned              h;
        std::string           mode;
        size_t                n;
        in >> max_fp >> filter_size >> h >> mode >> n;
        std::vector<uint64_t> counts(n);
        for (auto& c : counts)
            in >> c;
        gnbuild::IbfParams p;
        gnbuild::choose_capacity(max_fp, filter_size, counts, (uint8_t)h, mode, p);
        uint64_t digest = 1469598103934665603ull;
        size_t   n_spans = 0;
        if (p.n_bins)
        {
            gnbuild::true_fp(counts, p);
            for (const auto& b : gnbuild::lay_out_bins(p, counts))
            {
                for (uint64_t v : { (uint64_t)b.target, b.first, b.last })
                {
                    digest ^= v;
                    digest *= 1099511628211ull;
                }
                ++n_spans;
            }
        }
        std::printf("%llu %llu %u %llu %a %a %a %zu %llu\n", (unsigned long long)p.n_bins, (unsigned long long)p.max_hashes_bin,
                    (unsigned)p.hash_functions, (unsigned long long)p.bin_size_bits, p.max_fp, p.true_max_fp, p.true_avg_fp, n_spans,
                    (unsigned long long)digest);
    }
    return 0;
}
