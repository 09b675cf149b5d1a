// main.cpp -- ganon-classify entry point; exit codes as /root/reference/src/ganon-classify/main.cpp:7-17.
#include "config.hpp"

#include <cstdlib>

#include "filter_io.hpp"
#include "startup.hpp"

#include <iostream>

namespace gnhost
{
bool run(Config config);                  // classify.cpp
bool verify_filter(const Config& config); // verify.cpp
} // namespace gnhost

int main(int argc, char** argv)
{
    gnhost::StartupLog::get(); // (time zero of the [startup] lines)
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
    return gnhost::run(std::move(config.value())) ? EXIT_SUCCESS : EXIT_FAILURE;
}
