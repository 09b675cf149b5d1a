// TEST-ONLY: taps of ganon_amd/host/robin_order.hpp for tests/test_reference_order.py.
//   H <text>            -> rh_hash, std_hash (restated), std::hash<std::string> (the real libstdc++), all in hex
//   T <n> then n lines "<id> <hash hex>"  -> the ids in slot order on one line
#include "../../ganon_amd/host/robin_order.hpp"

#include <functional>
#include <iostream>
#include <sstream>
#include <string>

int main()
{
    std::string line;
    while (std::getline(std::cin, line))
    {
        if (line.rfind("H ", 0) == 0 || line == "H")
        {
            const std::string s = line.size() > 2 ? line.substr(2) : std::string();
            std::cout << std::hex << gnhost::rh_hash(s) << ' ' << gnhost::std_hash_bytes(s.data(), s.size()) << ' '
                      << (uint64_t)std::hash<std::string>{}(s) << std::dec << '\n';
        }
        else if (line.rfind("P ", 0) == 0)
        {
            std::istringstream is(line.substr(2));
            std::string        a, b;
            is >> a >> b;
            const uint64_t real = (uint64_t)std::hash<std::string>{}(a) ^ ((uint64_t)std::hash<std::string>{}(b) << 1);
            std::cout << std::hex << gnhost::pair_hash(a, b) << ' ' << real << std::dec << '\n';
        }
        else if (line.rfind("T ", 0) == 0)
        {
            const size_t       n = std::stoul(line.substr(2));
            gnhost::RobinSlots t;
            t.clear();
            for (size_t i = 0; i < n; ++i)
            {
                std::getline(std::cin, line);
                std::istringstream is(line);
                uint32_t           id;
                std::string        hx;
                is >> id >> hx;
                t.insert(id, std::stoull(hx, nullptr, 16));
            }
            std::vector<uint32_t> order;
            t.order(order);
            for (size_t i = 0; i < order.size(); ++i)
                std::cout << (i ? " " : "") << order[i];
            std::cout << '\n';
        }
    }
    return 0;
}
