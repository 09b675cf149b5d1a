// filter_io.cpp -- see filter_io.hpp.  cereal's portable binary conventions restated: arithmetic types are raw
// little-endian, std::string = u64 length + bytes, std::vector<T> = u64 size + elements (raw block for arithmetic
// T), std::tuple = elements in index order, bool = 1 byte.
#include "filter_io.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <thread>

#include <fcntl.h>
#include <unistd.h>

namespace gnhost
{

namespace
{

struct Reader
{
    std::ifstream is;
    std::string   path;
    uint64_t      size = 0, pos = 0;

    explicit Reader(const std::string& p) : is(p, std::ios::binary), path(p)
    {
        if (!is)
            throw std::runtime_error("cannot open filter file " + p);
        size = std::filesystem::file_size(p);
    }
    void raw(void* dst, uint64_t n)
    {
        if (pos + n > size)
            throw std::runtime_error(path + ": truncated (wanted " + std::to_string(n) + " bytes at offset "
                                     + std::to_string(pos) + ", file has " + std::to_string(size) + ")");
        is.read(reinterpret_cast<char*>(dst), (std::streamsize)n);
        if (!is)
            throw std::runtime_error(path + ": read error at offset " + std::to_string(pos));
        pos += n;
    }
    void skip(uint64_t n)
    {
        if (pos + n > size)
            throw std::runtime_error(path + ": truncated (payload of " + std::to_string(n) + " bytes at offset "
                                     + std::to_string(pos) + ", file has " + std::to_string(size) + ")");
        pos += n;
        is.seekg((std::streamoff)pos);
    }
    void seek(uint64_t to)
    {
        pos = to;
        is.clear();
        is.seekg((std::streamoff)pos);
    }
    template <typename T>
    T get()
    {
        T v;
        raw(&v, sizeof(T));
        return v;
    }
    std::string str()
    {
        const uint64_t n = get<uint64_t>();
        if (n > size - pos)
            throw std::runtime_error(path + ": implausible string length " + std::to_string(n) + " at offset "
                                     + std::to_string(pos));
        std::string s(n, '\0');
        if (n)
            raw(s.data(), n);
        return s;
    }
    uint64_t count(uint64_t min_elem_bytes)
    {
        const uint64_t n = get<uint64_t>();
        if (min_elem_bytes && n > (size - pos) / min_elem_bytes)
            throw std::runtime_error(path + ": implausible container size " + std::to_string(n) + " at offset "
                                     + std::to_string(pos));
        return n;
    }
};

// seqan3::interleaved_bloom_filter<uncompressed>: bins, technical_bins, bin_size_, hash_shift, bin_words, hash_funs
// (all size_t), then the sdsl::bit_vector `data`.
void read_ibf_fields(Reader& r, IbfShape& m)
{
    m.bins           = r.get<uint64_t>();
    m.technical_bins = r.get<uint64_t>();
    m.bin_size       = r.get<uint64_t>();
    m.hash_shift     = r.get<uint64_t>();
    m.bin_words      = r.get<uint64_t>();
    m.hash_funs      = r.get<uint64_t>();
    std::ostringstream why;
    if (m.bins == 0 || m.bin_size == 0)
        why << "empty filter (bins=" << m.bins << ", bin_size=" << m.bin_size << ")";
    else if (m.bin_words != ((m.bins + 63) >> 6))
        why << "bin_words " << m.bin_words << " != ceil(bins/64) for bins " << m.bins;
    else if (m.technical_bins != m.bin_words * 64)
        why << "technical_bins " << m.technical_bins << " != 64*bin_words";
    else if (m.hash_shift != (uint64_t)__builtin_clzll(m.bin_size))
        why << "hash_shift " << m.hash_shift << " != countl_zero(bin_size " << m.bin_size << ")";
    else if (m.hash_funs < 1 || m.hash_funs > 5)
        why << "hash_funs " << m.hash_funs << " outside 1..5";
    if (!why.str().empty())
        throw std::runtime_error(r.path + ": not a SeqAn3 IBF at offset " + std::to_string(r.pos) + ": " + why.str());
}

// The header sdsl-lite writes in front of a bit_vector's words.  Recalled layout (sdsl-lite v3 cereal support): the
// width as a one-byte size tag (1), the growth factor (float), the size in BITS (u64), then ceil(bits/64) words.  No
// real file pins that in the reference tree, so the variants that differ in exactly the parts a version could change
// are accepted too -- without the width byte and/or the growth factor, and with the size counted in 64-bit words --
// each still has to agree with technical_bins * bin_size, and for a flat .ibf (payload last in the file) the header
// length is additionally dictated by the file size.
// Returns with r positioned at the first payload byte.  `exact_len` = -1 when the file size cannot decide.
void read_bitvector_header(Reader& r, const IbfShape& m, int64_t exact_len)
{
    const uint64_t bits  = m.technical_bins * m.bin_size;
    const uint64_t start = r.pos;
    struct Variant
    {
        bool width, growth;
    };
    static const Variant variants[] = { { true, true }, { false, false }, { true, false }, { false, true } };
    std::string          seen;
    for (const auto& v : variants)
    {
        const int64_t len = (v.width ? 1 : 0) + (v.growth ? 4 : 0) + 8;
        if (exact_len >= 0 && len != exact_len)
            continue;
        if (start + (uint64_t)len > r.size)
            continue;
        r.seek(start);
        uint8_t width = 1;
        if (v.width)
            width = r.get<uint8_t>();
        if (v.growth)
            (void)r.get<float>();
        const uint64_t size = r.get<uint64_t>();
        if (width == 1 && (size == bits || size * 64 == bits))
            return;
        seen += " [width " + std::to_string(width) + ", size " + std::to_string(size) + "]";
    }
    r.seek(start);
    throw std::runtime_error(r.path + ": unexpected sdsl bit_vector header at offset " + std::to_string(start) + " (expected "
                             + std::to_string(bits) + " bits; candidates read:" + seen + ")");
}

void replace_all(std::string& str, const std::string& from, const std::string& to)
{
    size_t start_pos = 0;
    while ((start_pos = str.find(from, start_pos)) != std::string::npos)
    {
        str.replace(start_pos, from.length(), to);
        start_pos += to.length();
    }
}

// ---- streaming the payload -----------------------------------------------------------------------------------
// pread with a few threads: the page cache / a tmpfs copies at 3-6 GB/s per thread, one thread would be the limit
void parallel_pread(int fd, uint8_t* dst, uint64_t bytes, uint64_t offset, const std::string& path)
{
    const unsigned    hw     = std::max(1u, std::thread::hardware_concurrency());
    const unsigned    nth    = (unsigned)std::min<uint64_t>(std::min(16u, hw), std::max<uint64_t>(1, bytes >> 23)); // >= 8 MiB each
    std::atomic<bool> failed{ false };
    auto              work = [&](uint64_t lo, uint64_t hi) {
        while (lo < hi && !failed)
        {
            const ssize_t got = ::pread(fd, dst + lo, (size_t)std::min<uint64_t>(hi - lo, 1ull << 30), (off_t)(offset + lo));
            if (got <= 0)
            {
                failed = true;
                return;
            }
            lo += (uint64_t)got;
        }
    };
    if (nth <= 1)
        work(0, bytes);
    else
    {
        std::vector<std::thread> pool;
        const uint64_t           per = ((bytes + nth - 1) / nth + 4095) & ~4095ull;
        for (unsigned t = 0; t < nth; ++t)
        {
            const uint64_t lo = std::min<uint64_t>(bytes, (uint64_t)t * per), hi = std::min<uint64_t>(bytes, lo + per);
            if (lo < hi)
                pool.emplace_back(work, lo, hi);
        }
        for (auto& t : pool)
            t.join();
    }
    if (failed)
        throw std::runtime_error(path + ": read error / unexpected end of file in the filter payload");
}

struct Fd
{
    int fd = -1;
    explicit Fd(const std::string& p) : fd(::open(p.c_str(), O_RDONLY)) {}
    ~Fd()
    {
        if (fd >= 0)
            ::close(fd);
    }
};

// rows of IBF `ibf` from file offset `offset` into the sink: fill one staging buffer while the previous one is copied
void stream_matrix(const std::string& path, int fd, uint64_t offset, const IbfShape& m, uint32_t ibf, FilterSink& sink)
{
    const uint64_t row_bytes = m.bin_words * 8;
    const uint64_t total     = m.payload_bytes();
    const uint64_t want      = std::min<uint64_t>(total, 256ull << 20);
    const uint64_t per       = std::max<uint64_t>(1, want / row_bytes); // rows per chunk
    const uint64_t cap       = per * row_bytes;
    uint64_t*      stage[2]  = { sink.staging(0, cap), sink.staging(1, cap) };
    std::unique_ptr<uint64_t[]> own[2];
    for (int i = 0; i < 2; ++i)
        if (!stage[i])
        {
            own[i].reset(new uint64_t[cap / 8]);
            stage[i] = own[i].get();
        }
    std::string err;
    uint64_t    chunk = 0;
    for (uint64_t row = 0; row < m.bin_size; row += per, ++chunk)
    {
        const uint64_t n   = std::min(per, m.bin_size - row);
        uint64_t*      buf = stage[chunk & 1];
        parallel_pread(fd, reinterpret_cast<uint8_t*>(buf), n * row_bytes, offset + row * row_bytes, path);
        // the copy of the previous chunk ran while this one was read; it used the OTHER buffer, but the buffer filled
        // next is that one again, so it has to be finished before the next round
        if (!sink.drain(err) || !sink.rows(ibf, row, n, buf, err))
            throw std::runtime_error(path + ": " + err);
    }
    if (!sink.drain(err))
        throw std::runtime_error(path + ": " + err);
}

void load_ibf(const std::string& path, FilterMeta& out, FilterSink& sink)
{
    Reader r(path);
    out.is_hibf = false;
    int version[3];
    r.raw(version, sizeof(version)); // std::tuple<int,int,int>
    IBFConfig& c     = out.ibf_config;
    c.n_bins         = r.get<uint64_t>();
    c.max_hashes_bin = r.get<uint64_t>();
    c.hash_functions = r.get<uint8_t>();
    c.kmer_size      = r.get<uint8_t>();
    c.window_size    = r.get<uint16_t>();
    c.bin_size_bits  = r.get<uint64_t>();
    c.max_fp         = r.get<double>();
    c.true_max_fp    = r.get<double>();
    c.true_avg_fp    = r.get<double>();

    std::vector<std::pair<std::string, uint64_t>> hashes_count;
    const uint64_t                                nhc = r.count(16);
    for (uint64_t i = 0; i < nhc; ++i)
    {
        std::string t = r.str();
        uint64_t    n = r.get<uint64_t>();
        hashes_count.emplace_back(std::move(t), n);
    }
    std::vector<std::pair<uint64_t, std::string>> bin_map;
    const uint64_t                                nbm = r.count(16);
    for (uint64_t i = 0; i < nbm; ++i)
    {
        uint64_t    b = r.get<uint64_t>();
        std::string t = r.str();
        bin_map.emplace_back(b, std::move(t));
    }
    IbfShape m;
    read_ibf_fields(r, m);
    // the payload is the last thing in the file: whatever precedes it is the bit_vector header
    const uint64_t payload = m.payload_bytes();
    if (r.size < r.pos + 8 + payload)
        throw std::runtime_error(path + ": truncated (the IBF payload needs " + std::to_string(payload) + " bytes, the file has "
                                 + std::to_string(r.size - r.pos) + " left)");
    read_bitvector_header(r, m, (int64_t)(r.size - r.pos - payload));
    const uint64_t payload_at = r.pos;
    if (m.bins != c.n_bins || m.bin_size != c.bin_size_bits || m.hash_funs != c.hash_functions)
        throw std::runtime_error(path + ": IBFConfig (n_bins/bin_size_bits/hash_functions) disagrees with the stored IBF");
    if (c.kmer_size == 0 || c.kmer_size > 32 || c.window_size < c.kmer_size)
        throw std::runtime_error(path + ": invalid k/w in IBFConfig");

    // target order: first appearance scanning bins in ascending order (deterministic; the reference iterates a
    // robin_hood map, GanonClassify.cpp:1021-1025)
    std::sort(bin_map.begin(), bin_map.end(), [](auto const& a, auto const& b) { return a.first < b.first; });
    std::map<std::string, size_t> idx;
    for (auto const& [binno, target] : bin_map)
    {
        if (binno >= m.bins)
            throw std::runtime_error(path + ": bin_map references bin " + std::to_string(binno) + " >= bins");
        auto it = idx.find(target);
        if (it == idx.end())
        {
            it = idx.emplace(target, out.targets.size()).first;
            out.targets.push_back(target);
            out.target_bins.emplace_back();
        }
        out.target_bins[it->second].push_back(binno);
    }
    // per-target fpr, GanonClassify.cpp:968-982
    std::map<std::string, double> fpr;
    for (auto const& [target, count] : hashes_count)
    {
        uint64_t n_bins_target = std::ceil(count / static_cast<double>(c.max_hashes_bin));
        uint64_t n_hashes_bin  = std::ceil(count / static_cast<double>(n_bins_target));
        fpr[target] = 1.0 - std::pow(1.0 - false_positive(c.bin_size_bits, c.hash_functions, n_hashes_bin), n_bins_target);
    }
    out.target_fpr.resize(out.targets.size(), 0.0);
    for (size_t t = 0; t < out.targets.size(); ++t)
    {
        auto it = fpr.find(out.targets[t]);
        if (it != fpr.end())
            out.target_fpr[t] = it->second; // operator[] default (0.0) otherwise, like target_fpr[target] at :533
    }
    out.bin_count = m.bins;
    out.shapes.assign(1, m);

    std::string err;
    if (!sink.begin(out, err))
        throw std::runtime_error(path + ": " + err);
    Fd fd(path);
    if (fd.fd < 0)
        throw std::runtime_error("cannot open filter file " + path);
    stream_matrix(path, fd.fd, payload_at, m, 0, sink);
    if (!sink.end(err))
        throw std::runtime_error(path + ": " + err);
}

void load_hibf(const std::string& path, FilterMeta& out, FilterSink& sink)
{
    Reader r(path);
    out.is_hibf = true;
    (void)r.get<uint32_t>();                     // parsed_version
    const uint64_t window_size = r.get<uint64_t>();
    const uint64_t shape_size  = r.get<uint64_t>(); // seqan3::shape (dynamic_bitset<58>): size, bits
    const uint64_t shape_bits  = r.get<uint64_t>();
    (void)shape_size;
    (void)r.get<uint8_t>(); // parts
    (void)r.get<uint8_t>(); // compressed
    std::vector<std::vector<std::string>> bin_path;
    const uint64_t                        nbp = r.count(8);
    bin_path.resize(nbp);
    for (auto& lst : bin_path)
    {
        const uint64_t n = r.count(8);
        lst.resize(n);
        for (auto& s : lst)
            s = r.str();
    }
    const double fpr = r.get<double>();
    (void)r.get<uint8_t>(); // is_hibf
    // hierarchical_interleaved_bloom_filter: ibf_vector, next_ibf_id, user_bins{user_bin_filenames, ibf_bin_to_filename_position}.
    // The tables FOLLOW the matrices: first pass over the IBF headers only (payloads skipped), matrices in a second pass.
    const uint64_t nibf = r.count(48);
    if (nibf == 0)
        throw std::runtime_error(path + ": HIBF without IBFs");
    out.shapes.resize(nibf);
    std::vector<uint64_t> payload_at(nibf);
    for (uint64_t i = 0; i < nibf; ++i)
    {
        read_ibf_fields(r, out.shapes[i]);
        read_bitvector_header(r, out.shapes[i], -1);
        payload_at[i] = r.pos;
        r.skip(out.shapes[i].payload_bytes());
    }
    auto read_vv = [&](std::vector<std::vector<int64_t>>& vv) {
        const uint64_t n = r.count(8);
        vv.resize(n);
        for (auto& v : vv)
        {
            const uint64_t m = r.count(8);
            v.resize(m);
            if (m)
                r.raw(v.data(), m * 8);
        }
    };
    read_vv(out.next_ibf_id);
    const uint64_t           nub = r.count(8);
    std::vector<std::string> user_bin_filenames(nub);
    for (auto& s : user_bin_filenames)
        s = r.str();
    read_vv(out.bin_to_user);
    if (r.pos != r.size)
        throw std::runtime_error(path + ": " + std::to_string(r.size - r.pos) + " trailing bytes after the HIBF");
    if (out.next_ibf_id.size() != nibf || out.bin_to_user.size() != nibf)
        throw std::runtime_error(path + ": next_ibf_id / ibf_bin_to_filename_position do not cover every IBF");
    for (uint64_t i = 0; i < nibf; ++i)
        if (out.next_ibf_id[i].size() < out.shapes[i].bins || out.bin_to_user[i].size() < out.shapes[i].bins)
            throw std::runtime_error(path + ": per-bin tables shorter than the IBF's bin count");
    out.n_user_bins = nub;

    // ibf_config from raptor params, GanonClassify.cpp:903-906
    out.ibf_config.window_size = (uint16_t)window_size;
    out.ibf_config.kmer_size   = (uint8_t)__builtin_popcountll(shape_bits);
    out.ibf_config.max_fp      = fpr;
    if (out.ibf_config.kmer_size == 0 || out.ibf_config.kmer_size > 32 || window_size < out.ibf_config.kmer_size)
        throw std::runtime_error(path + ": invalid shape/window in the raptor index header");

    // targets from bin_path, GanonClassify.cpp:908-935
    std::map<std::string, size_t> idx;
    uint64_t                      binno = 0;
    for (auto const& file_list : bin_path)
    {
        for (auto const& filename : file_list)
        {
            std::string f     = std::filesystem::path(filename).filename().string();
            size_t      found = f.find(".minimiser");
            if (found != std::string::npos)
                f = f.substr(0, found);
            replace_all(f, "|||", ".");
            replace_all(f, "---", " ");
            auto it = idx.find(f);
            if (it == idx.end())
            {
                it = idx.emplace(f, out.targets.size()).first;
                out.targets.push_back(f);
                out.target_bins.emplace_back();
                out.target_fpr.push_back(fpr);
            }
            out.target_bins[it->second].push_back(binno);
        }
        ++binno;
    }
    for (auto const& b : out.target_bins)
        if (b[0] >= nub)
            throw std::runtime_error(path + ": bin_path has more entries than user bins");
    out.bin_count = nub;

    std::string err;
    if (!sink.begin(out, err))
        throw std::runtime_error(path + ": " + err);
    Fd fd(path);
    if (fd.fd < 0)
        throw std::runtime_error("cannot open filter file " + path);
    for (uint64_t i = 0; i < nibf; ++i)
        stream_matrix(path, fd.fd, payload_at[i], out.shapes[i], (uint32_t)i, sink);
    if (!sink.end(err))
        throw std::runtime_error(path + ": " + err);
}

} // namespace

double false_positive(uint64_t bin_size_bits, uint8_t hash_functions, uint64_t n_hashes)
{
    return std::pow(1 - std::exp(-hash_functions / (bin_size_bits / static_cast<double>(n_hashes))), hash_functions);
}

void load_filter_file(const std::string& path, bool hibf, FilterMeta& meta, FilterSink& sink)
{
    meta = FilterMeta();
    if (hibf)
        load_hibf(path, meta, sink);
    else
        load_ibf(path, meta, sink);
}

std::map<std::string, TaxNode> load_tax(const std::string& path)
{
    std::map<std::string, TaxNode> tax;
    std::ifstream                  infile(path);
    std::string                    line;
    while (std::getline(infile, line, '\n'))
    {
        std::istringstream       stream_line(line);
        std::vector<std::string> fields;
        std::string              field;
        while (std::getline(stream_line, field, '\t'))
            fields.push_back(field);
        if (fields.size() < 4)
            continue;
        tax[fields[0]] = TaxNode{ fields[1], fields[2], fields[3] };
    }
    return tax;
}

} // namespace gnhost
