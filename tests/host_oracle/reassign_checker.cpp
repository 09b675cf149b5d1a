// reassign_checker.cpp -- TEST-ONLY: the gn_reassign_* calls of include/ganon_hip.h on the CPU, in plain loops that follow
// /root/reference/src/ganon/reassign.py:96-145,226-241 statement by statement.  Linked with ganon_amd/host/reassign.cpp into
// tests/host_oracle/ganon-reassign-oracle so that the CPU suite can run the text side of `ganon-reassign` (tables, .one,
// .rep, log) against the vectors the reference's own reassign.py produced; the product binary links libganon_hip.so.
#include "../../include/ganon_hip.h"

#include <cmath>
#include <vector>

struct gn_reassign
{
    std::vector<uint64_t> off;
    std::vector<uint32_t> target;
    uint32_t              n_targets = 0;
    std::vector<uint64_t> uniq, counts, choice;
    std::vector<double>   prob, diffs;
    uint64_t              n_unique = 0, n_multi = 0;
};

extern "C" const char* gn_last_error(void)
{
    return "checker";
}

static uint64_t top(const gn_reassign* g, uint64_t r) // get_top_match
{
    uint64_t at = g->off[r];
    double   max_p = 0;
    for (uint64_t i = g->off[r]; i < g->off[r + 1]; ++i)
        if (g->prob[g->target[i]] > max_p)
        {
            max_p = g->prob[g->target[i]];
            at    = i;
        }
    return at;
}

extern "C" int gn_reassign_create(int, uint64_t n_reads, uint64_t n_entries, uint32_t n_targets, const uint64_t* off, const uint32_t* target,
                                  gn_reassign** out)
{
    auto* g = new gn_reassign();
    g->off.assign(off, off + n_reads + 1);
    g->target.assign(target, target + n_entries);
    g->n_targets = n_targets;
    g->uniq.assign(n_targets, 0);
    for (uint64_t r = 0; r < n_reads; ++r)
    {
        const uint64_t d = off[r + 1] - off[r];
        if (d == 1)
        {
            ++g->uniq[target[off[r]]];
            ++g->n_unique;
        }
        else if (d > 1)
            ++g->n_multi;
    }
    *out = g;
    return GN_OK;
}

extern "C" int gn_reassign_run(gn_reassign* g, uint32_t max_iter, double threshold, uint32_t* iterations)
{
    const uint64_t n_reads = g->off.size() - 1;
    const double   tiw     = g->n_unique ? (double)g->n_unique : 1.0;
    g->prob.resize(g->n_targets);
    for (uint32_t t = 0; t < g->n_targets; ++t)
        g->prob[t] = (double)g->uniq[t] / tiw;
    g->diffs.clear();
    uint32_t it = 0;
    for (;;)
    {
        g->counts = g->uniq;
        for (uint64_t r = 0; r < n_reads; ++r)
            if (g->off[r + 1] - g->off[r] > 1)
                ++g->counts[g->target[top(g, r)]];
        double diff = 0;
        for (uint32_t t = 0; t < g->n_targets; ++t)
        {
            const double np = (double)g->counts[t] / (double)n_reads;
            diff += std::fabs(g->prob[t] - np);
            g->prob[t] = np;
        }
        g->diffs.push_back(diff);
        if (diff <= threshold)
            break;
        if (max_iter > 0 && it == max_iter - 1)
            break;
        ++it;
    }
    g->choice.resize(n_reads);
    for (uint64_t r = 0; r < n_reads; ++r)
        g->choice[r] = g->off[r + 1] - g->off[r] == 1 ? g->off[r] : top(g, r);
    if (iterations)
        *iterations = it + 1;
    return GN_OK;
}

extern "C" int gn_reassign_diffs(const gn_reassign* g, double* diffs, uint32_t cap)
{
    for (uint32_t i = 0; i < cap && i < g->diffs.size(); ++i)
        diffs[i] = g->diffs[i];
    return GN_OK;
}

extern "C" int gn_reassign_fetch(gn_reassign* g, uint64_t* counts, uint64_t* unique, double* prob, uint64_t* choice)
{
    for (uint32_t t = 0; t < g->n_targets; ++t)
    {
        if (counts)
            counts[t] = g->counts[t];
        if (unique)
            unique[t] = g->uniq[t];
        if (prob)
            prob[t] = g->prob[t];
    }
    if (choice)
        for (size_t r = 0; r < g->choice.size(); ++r)
            choice[r] = g->choice[r];
    return GN_OK;
}

extern "C" int gn_reassign_info(const gn_reassign* g, uint64_t* nu, uint64_t* nm, uint64_t* nw, float* ms, uint64_t* bytes)
{
    if (nu)
        *nu = g->n_unique;
    if (nm)
        *nm = g->n_multi;
    if (nw)
        *nw = 0;
    if (ms)
        *ms = 0.f;
    if (bytes)
        *bytes = 0;
    return GN_OK;
}

extern "C" int gn_reassign_free(gn_reassign* g)
{
    delete g;
    return GN_OK;
}
