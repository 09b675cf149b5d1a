// TEST-ONLY: ganon_amd/host/pgzip.cpp against zlib's own gzread on the same file.
//   pgzip_check <file.gz> <threads> <chunk bytes> [read unit]   -> "OK <bytes> chunks=.. redone=.. members=.. markers=.." | "DIFF ..." | "ERROR <msg> after <bytes>"
#include "../../ganon_amd/host/pgzip.hpp"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>

int main(int argc, char** argv)
{
    if (argc < 4)
        return 2;
    const unsigned threads = (unsigned)std::atoi(argv[2]);
    const size_t   chunk   = (size_t)std::atoll(argv[3]);
    const size_t   unit    = argc > 4 ? (size_t)std::atoll(argv[4]) : (1u << 20);
    auto           pg      = gnhost::ParallelGzip::open(argv[1], threads, 0, chunk);
    if (!pg)
    {
        std::cout << "NOTGZIP\n";
        return 0;
    }
    if (std::getenv("PGZIP_BENCH")) // timing only: read through, no comparison
    {
        std::vector<char> buf(unit);
        uint64_t          o = 0;
        for (;;)
        {
            const size_t na = pg->pread(buf.data(), unit, o);
            o += na;
            pg->release_below(o);
            if (na < unit)
                break;
        }
        std::cout << "READ " << o << "\n";
        return 0;
    }
    gzFile gz = gzopen(argv[1], "rb");
    gzbuffer(gz, 1 << 20);
    std::vector<char> a(unit), b(unit);
    uint64_t          off = 0;
    for (;;)
    {
        size_t na = 0;
        try
        {
            na = pg->pread(a.data(), unit, off);
        }
        catch (std::exception const& e)
        {
            std::cout << "ERROR " << e.what() << " after " << off << "\n";
            return 0;
        }
        // zlib's view of the same bytes
        size_t nb = 0;
        while (nb < na)
        {
            const int r = gzread(gz, b.data() + nb, (unsigned)(na - nb));
            if (r <= 0)
                break;
            nb += (size_t)r;
        }
        if (nb != na || std::memcmp(a.data(), b.data(), na) != 0)
        {
            std::cout << "DIFF at " << off << " (" << na << " vs " << nb << " bytes)\n";
            return 0;
        }
        off += na;
        pg->release_below(off);
        if (na < unit)
            break;
    }
    char      extra;
    const int more = gzread(gz, &extra, 1);
    if (more > 0)
    {
        std::cout << "DIFF zlib has more data after " << off << "\n";
        return 0;
    }
    const auto st = pg->stats();
    std::cout << "OK " << off << " chunks=" << st.chunks << " redone=" << st.redone << " members=" << st.members << " markers=" << st.markers << "\n";
    return 0;
}
