// main.cpp -- ganon-classify entry point; exit codes as /root/reference/src/ganon-classify/main.cpp:7-17.
#include "config.hpp"

#include <cstdlib>

namespace gnhost
{
bool run(Config config); // classify.cpp
}

int main(int argc, char** argv)
{
    int  exit_code = 0;
    auto config    = gnhost::parse_command_line(argc, argv, exit_code);
    if (!config.has_value())
        return exit_code;
    return gnhost::run(std::move(config.value())) ? EXIT_SUCCESS : EXIT_FAILURE;
}
