// reassign_main.cpp -- `ganon-reassign`: the command line of `ganon reassign` (/root/reference/src/ganon/config.py:746-808)
// in front of run_reassign (reassign.cpp).  Exit code 0 iff the run returned true (/root/reference/src/ganon/ganon.py:52-58).
#include "reassign.hpp"

#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

namespace
{

const char* kHelp =
    "usage: ganon-reassign -i [...] [-o OUTPUT_PREFIX] [-e MAX_ITER] [-s THRESHOLD] [--remove-all] [--skip-one] [--skip-rep]\n"
    "                      [--device N] [--verbose] [--quiet]\n\n"
    "required arguments:\n"
    "  -i [ ...], --input-prefix [ ...]   Input prefix to find files from ganon classify (.rep and .all)\n"
    "  -o, --output-prefix                Alternative output prefix for reassigned files. If not provided, will use same path of\n"
    "                                     input files (will overwrite .rep). In case of multiple files, the output will be the\n"
    "                                     suffix. Example: {output_prefix}{filename}.one\n\n"
    "EM arguments:\n"
    "  -e, --max-iter                     Max. number of iterations for the EM algorithm. If 0, will run until convergence (check\n"
    "                                     --threshold) (default: 10)\n"
    "  -s, --threshold                    Convergence threshold limit to stop the EM algorithm. (default: 0)\n\n"
    "other arguments:\n"
    "  --remove-all                       Remove input file (.all) after processing.\n"
    "  --skip-one                         Do not write output file (.one) after processing.\n"
    "  --skip-rep                         Do not write report file (.rep) after processing.\n"
    "  --device N                         HIP device the EM runs on (default: 0; there is no CPU path)\n"
    "  --verbose                          Verbose output mode\n"
    "  --quiet                            Quiet output mode\n";

bool is_flag(const char* a)
{
    return a[0] == '-' && a[1] != '\0' && !(a[1] >= '0' && a[1] <= '9') && a[1] != '.';
}

} // namespace

int main(int argc, char** argv)
{
    gnhost::ReassignConfig cfg;
    bool                   have_input = false;
    auto                   fail       = [](const std::string& m) {
        std::cerr << "ganon-reassign: error: " << m << '\n';
        return 2; // argparse's exit code for a bad command line
    };
    for (int i = 1; i < argc; ++i)
    {
        std::string a = argv[i], v;
        bool        has_v = false;
        const auto  eq    = a.find('=');
        if (a.rfind("--", 0) == 0 && eq != std::string::npos)
        {
            v     = a.substr(eq + 1);
            a     = a.substr(0, eq);
            has_v = true;
        }
        auto value = [&](std::string& out) {
            if (has_v)
                out = v;
            else if (i + 1 < argc)
                out = argv[++i];
            else
                return false;
            return true;
        };
        if (a == "-h" || a == "--help")
        {
            std::cerr << kHelp;
            return 0;
        }
        else if (a == "-i" || a == "--input-prefix")
        {
            have_input = true;
            if (has_v)
                cfg.input_prefix.push_back(v);
            while (i + 1 < argc && !is_flag(argv[i + 1])) // nargs="*"
                cfg.input_prefix.push_back(argv[++i]);
        }
        else if (a == "-o" || a == "--output-prefix")
        {
            if (!value(cfg.output_prefix))
                return fail("argument -o/--output-prefix: expected one argument");
        }
        else if (a == "-e" || a == "--max-iter")
        {
            std::string s;
            char*       end = nullptr;
            if (!value(s))
                return fail("argument -e/--max-iter: expected one argument");
            const long long x = std::strtoll(s.c_str(), &end, 10);
            if (s.empty() || *end || x < 0 || x > 0xffffffffll)
                return fail("argument -e/--max-iter: invalid value: '" + s + "'");
            cfg.max_iter = (uint32_t)x;
        }
        else if (a == "-s" || a == "--threshold")
        {
            std::string s;
            char*       end = nullptr;
            if (!value(s))
                return fail("argument -s/--threshold: expected one argument");
            const double x = std::strtod(s.c_str(), &end);
            if (s.empty() || *end || x != x) // (any float, as argparse: a negative threshold runs to --max-iter)
                return fail("argument -s/--threshold: invalid value: '" + s + "'");
            cfg.threshold = x;
        }
        else if (a == "--device")
        {
            std::string s;
            if (!value(s))
                return fail("argument --device: expected one argument");
            cfg.device = std::atoi(s.c_str());
        }
        else if (a == "--remove-all")
            cfg.remove_all = true;
        else if (a == "--skip-one")
            cfg.skip_one = true;
        else if (a == "--skip-rep")
            cfg.skip_rep = true;
        else if (a == "--verbose")
            cfg.verbose = true;
        else if (a == "--quiet")
            cfg.quiet = true;
        else
            return fail("unrecognized arguments: " + a);
    }
    if (!have_input)
        return fail("the following arguments are required: -i/--input-prefix");
    try
    {
        return gnhost::run_reassign(cfg) ? EXIT_SUCCESS : EXIT_FAILURE;
    }
    catch (const std::exception& e)
    {
        std::cerr << "ganon-reassign: " << e.what() << '\n';
        return EXIT_FAILURE;
    }
}
