// main.cpp -- ganon-classify entry point; exit codes as /root/reference/src/ganon-classify/main.cpp:7-17.
#include "config.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "filter_io.hpp"
#include "startup.hpp"
#include "tunables.hpp"

#include <iostream>

namespace gnhost
{
bool run(Config config);                  // classify.cpp
bool verify_filter(const Config& config); // verify.cpp
} // namespace gnhost

int main(int argc, char** argv)
{
    gnhost::StartupLog::get(); // (time zero of the [startup] lines)
    gnhost::HostTunables::init(); // the environment is read here, once (tunables.hpp); --verbose lists what was set
    int  exit_code = 0;
    auto config    = gnhost::parse_command_line(argc, argv, exit_code);
    if (!config.has_value())
        return exit_code;
    if (!config->inspect_filter.empty())
    {
        std::cout.precision(17);
        return gnhost::inspect_filter_file(config->inspect_filter, config->hibf, std::cout) ? EXIT_SUCCESS : EXIT_FAILURE;
    }
    if (!config->verify_filter.empty())
        return gnhost::verify_filter(config.value()) ? EXIT_SUCCESS : EXIT_FAILURE;
    const bool   verbose = config->verbose;
    const bool   ok      = gnhost::run(std::move(config.value()));
    const double t_done  = gnhost::StartupLog::now();
    if (verbose)
        gnhost::StartupLog::get().span("main() returns: every output file is closed (what follows is the runtime's teardown, skipped)", t_done);
    if (verbose)
        gnhost::StartupLog::get().print(std::cerr);
    // Every output is written and closed inside run().  What a normal return would do next -- unlock gigabytes of page-locked
    // buffers, free the filters, take the HIP runtime down handle by handle -- took 0.6 s after 0.14 s of classification
    // (profiles/r05_e2e_startup1.json); the kernel driver reclaims all of it at once when the process ends.
    std::cout.flush();
    std::cerr.flush();
    std::fflush(nullptr);
    // (a profiler that writes its trace from an exit handler needs the normal return: rocprofv3 preloads its tool library)
    if (gnhost::fast_exit())
        std::_Exit(ok ? EXIT_SUCCESS : EXIT_FAILURE);
    return ok ? EXIT_SUCCESS : EXIT_FAILURE;
}
