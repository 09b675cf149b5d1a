#include "build_params.hpp"

#include <cmath>

namespace gnbuild
{

uint64_t bits_for(double max_fp, uint64_t elements)
{
    return std::ceil((elements * std::log(max_fp)) / std::log(1.0 / std::pow(2, std::log(2))));
}

uint64_t bits_for(double max_fp, uint64_t elements, uint8_t hash_functions)
{
    return std::ceil(elements * (-hash_functions / std::log(1 - std::exp(std::log(max_fp) / hash_functions))));
}

uint8_t hash_functions_for(uint64_t bin_size_bits, uint64_t elements, uint8_t requested)
{
    uint8_t h = requested;
    if (h == 0) // ln2 * bits per element, truncated to a byte (:308-314)
        h = static_cast<uint8_t>(static_cast<int64_t>(std::log(2) * (bin_size_bits / static_cast<double>(elements))));
    if (h > kMaxHashFunctions || h == 0)
        h = kMaxHashFunctions;
    return h;
}

double bloom_fp(uint64_t bin_size_bits, uint8_t hash_functions, uint64_t elements)
{
    return std::pow(1 - std::exp(-hash_functions / (bin_size_bits / static_cast<double>(elements))), hash_functions);
}

uint64_t padded_bins(uint64_t n_bins)
{
    return std::ceil(n_bins / 64.0) * 64;
}

namespace
{

uint64_t bins_needed(const std::vector<uint64_t>& counts, uint64_t capacity) // number_of_bins (:336-347)
{
    uint64_t n = 0;
    for (uint64_t c : counts)
        n += std::ceil(c / static_cast<double>(capacity)); // (uint64 += double: summed in double, as the reference does)
    return n;
}

// by how much a bin has to grow so that a target split over `splits` bins still meets max_fp (:350-362)
double split_correction(uint64_t splits, double max_fp, uint8_t hash_functions, uint64_t capacity)
{
    const double per_bin = 1.0 - std::exp(std::log(1.0 - max_fp) / splits);
    const size_t grown   = bits_for(per_bin, capacity, hash_functions);
    const size_t plain   = bits_for(max_fp, capacity, hash_functions);
    return static_cast<double>(grown) / plain;
}

struct Candidate
{
    uint64_t capacity, n_bins, filter_bits;
    double   fp;
};

} // namespace

void choose_capacity(double max_fp, double filter_size_mb, const std::vector<uint64_t>& counts, uint8_t hash_functions,
                     const std::string& mode, IbfParams& out)
{
    uint64_t largest = 0;
    for (uint64_t c : counts)
        if (c > largest)
            largest = c;
    if (largest == 0)
        return; // nothing to build (the reference's loop has no sensible result here either; n_bins stays 0)
    const bool by_size = filter_size_mb != 0;

    std::vector<Candidate> candidates;
    uint64_t               least_bits = 0, least_bins = 0;
    double                 least_fp   = 1;
    const size_t           stride     = largest < 100 ? largest : 100;
    for (size_t n = largest + 1; n > stride; n -= stride)
    {
        const uint64_t capacity = n - 1;
        const uint64_t n_bins   = bins_needed(counts, capacity);
        int64_t        bin_bits = 0;
        uint8_t        h        = 0;
        if (by_size)
        {
            bin_bits = (filter_size_mb / static_cast<double>(padded_bins(n_bins))) * 8388608u;
            h        = hash_functions_for(bin_bits, capacity, hash_functions);
        }
        else if (hash_functions == 0)
        {
            bin_bits = bits_for(max_fp, capacity);
            h        = hash_functions_for(bin_bits, capacity, hash_functions);
        }
        else
        {
            h        = hash_functions_for(bin_bits, capacity, hash_functions);
            bin_bits = bits_for(max_fp, capacity, h);
        }
        const uint64_t splits = std::ceil(largest / static_cast<double>(capacity)); // of the largest target
        double         fp     = 0;
        uint64_t       bits   = 0;
        if (by_size)
        {
            fp = 1 - std::pow(1.0 - bloom_fp(bin_bits, h, capacity), splits);
            if (fp < least_fp)
                least_fp = fp;
        }
        else
        {
            const uint64_t filled = std::ceil(largest / static_cast<double>(splits)); // what a split bin really holds
            double         approx = bloom_fp(bin_bits, h, filled);
            if (approx > max_fp)
                approx = max_fp;
            const double rate = split_correction(splits, approx, h, capacity);
            if (std::isinf(rate) || std::isnan(rate))
                break;
            bin_bits = bin_bits * rate;
            bits     = bin_bits * padded_bins(n_bins);
            if (bits == 0)
                break;
            if (bits < least_bits || least_bits == 0)
                least_bits = bits;
        }
        candidates.push_back(Candidate{ capacity, n_bins, bits, fp });
        if (n_bins < least_bins || least_bins == 0)
            least_bins = n_bins;
    }

    // weighted harmonic mean of (size or fp) ratio and bin-count ratio, each relative to the best seen (:560-616)
    double tilt = 1;
    if (mode == "smaller" || mode == "faster")
        tilt = 0.5;
    else if (mode == "smallest" || mode == "fastest")
        tilt = 0;
    double w_var = 1, w_bins = 1;
    if (mode == "smaller" || mode == "smallest")
        w_var = tilt;
    else if (mode == "faster" || mode == "fastest")
        w_bins = tilt;

    double best = 0;
    for (const Candidate& c : candidates)
    {
        const double var_ratio  = by_size ? c.fp / least_fp : c.filter_bits / static_cast<double>(least_bits);
        const double bins_ratio = c.n_bins / static_cast<double>(least_bins);
        const double score      = (1 + std::pow(tilt, 2)) * ((var_ratio * bins_ratio) / ((w_var * var_ratio) + (w_bins * bins_ratio)));
        if (score < best || best == 0)
        {
            best = score;
            if (by_size)
            {
                out.bin_size_bits = (filter_size_mb / static_cast<double>(padded_bins(c.n_bins))) * 8388608u;
                out.max_fp        = c.fp;
            }
            else
            {
                out.bin_size_bits = c.filter_bits / padded_bins(c.n_bins);
                out.max_fp        = max_fp;
            }
            out.max_hashes_bin = c.capacity;
            out.n_bins         = c.n_bins;
            out.hash_functions = hash_functions_for(out.bin_size_bits, c.capacity, hash_functions);
        }
    }
}

void true_fp(const std::vector<uint64_t>& counts, IbfParams& p)
{
    double highest = 0, sum = 0;
    for (uint64_t c : counts)
    {
        if (c == 0) // no minimisers, no bins: contributes 0 (what the reference's 0/0 arithmetic ends in on x86-64), still averaged over
            continue;
        const uint64_t bins    = std::ceil(c / static_cast<double>(p.max_hashes_bin));
        const uint64_t per_bin = std::ceil(c / static_cast<double>(bins));
        const double   fp      = 1.0 - std::pow(1.0 - bloom_fp(p.bin_size_bits, p.hash_functions, per_bin), bins);
        if (fp > highest)
            highest = fp;
        sum += fp;
    }
    p.true_max_fp = highest;
    p.true_avg_fp = sum / static_cast<double>(counts.size());
}

std::vector<BinSpan> lay_out_bins(const IbfParams& p, const std::vector<uint64_t>& counts, std::vector<uint64_t>* shares)
{
    std::vector<BinSpan> bins;
    if (shares)
        shares->assign(counts.size(), 0);
    for (uint32_t t = 0; t < counts.size(); ++t)
    {
        const uint64_t c = counts[t];
        if (c == 0)
            continue;
        const uint64_t n_bins = std::ceil(c / static_cast<double>(p.max_hashes_bin));
        uint64_t       share  = std::ceil(c / static_cast<double>(n_bins));
        if (share > p.max_hashes_bin)
            share = p.max_hashes_bin;
        if (shares)
            (*shares)[t] = share;
        for (uint64_t i = 0; i < n_bins; ++i)
        {
            const uint64_t first = i * share;
            if (first >= c)
                break;
            uint64_t last = first + share - 1;
            if (last >= c)
                last = c - 1;
            bins.push_back(BinSpan{ t, first, last });
        }
    }
    return bins;
}

} // namespace gnbuild
