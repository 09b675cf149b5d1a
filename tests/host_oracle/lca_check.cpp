// lca_check.cpp -- TEST-ONLY driver of the product's ganon_amd/host/lca.hpp:
//   lca_check <file.tax> <root> <node,node,...> [<node,node,...> ...]   -> one LCA per query line on stdout
// (the .tax rows are "node <tab> parent <tab> ..."; edges are added parent -> node like classify.cpp does)
#include "../../ganon_amd/host/lca.hpp"

#include <fstream>
#include <iostream>
#include <sstream>

int main(int argc, char** argv)
{
    if (argc < 4)
        return 2;
    gnhost::LCA   lca;
    std::ifstream in(argv[1]);
    std::string   line;
    while (std::getline(in, line))
    {
        std::istringstream       ss(line);
        std::vector<std::string> f;
        std::string              x;
        while (std::getline(ss, x, '\t'))
            f.push_back(x);
        if (f.size() >= 2)
            lca.addEdge(f[1], f[0]);
    }
    lca.doEulerWalk(argv[2]);
    for (int i = 3; i < argc; ++i)
    {
        std::istringstream       ss(argv[i]);
        std::vector<std::string> q;
        std::string              x;
        while (std::getline(ss, x, ','))
            q.push_back(x);
        std::cout << lca.getLCA(q) << "\n";
    }
    return 0;
}
