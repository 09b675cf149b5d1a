// filter_io.cpp -- see filter_io.hpp.  cereal's portable binary conventions restated: arithmetic types are raw
// little-endian, std::string = u64 length + bytes, std::vector<T> = u64 size + elements (raw block for arithmetic
// T), std::tuple = elements in index order, bool = 1 byte.
#include "filter_io.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
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
    std::ostream* trace = nullptr; // --inspect-filter: every named field is printed with its offset as it is read

    template <typename V>
    void note(uint64_t at, const char* name, const V& value, const char* what = "")
    {
        if (!trace)
            return;
        std::ostringstream o;
        o << "@" << at;
        std::string left = o.str();
        left.resize(std::max<size_t>(left.size(), 12), ' ');
        left += name;
        left.resize(std::max<size_t>(left.size(), 48), ' ');
        *trace << left << " = " << value << (what[0] ? "   " : "") << what << "\n";
    }
    void check(const char* what, bool ok, const std::string& detail = std::string())
    {
        if (trace)
        {
            std::string left = std::string("check       ") + what;
            left.resize(std::max<size_t>(left.size(), 72), ' ');
            *trace << left << (ok ? " ok" : " MISMATCH") << (detail.empty() ? "" : "   ") << detail << "\n";
        }
    }

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
    T get(const char* name = nullptr)
    {
        const uint64_t at = pos;
        T              v;
        raw(&v, sizeof(T));
        if (name)
        {
            if constexpr (sizeof(T) == 1)
                note(at, name, (unsigned)v);
            else
                note(at, name, v);
        }
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
    m.bins           = r.get<uint64_t>("ibf.bins");
    m.technical_bins = r.get<uint64_t>("ibf.technical_bins");
    m.bin_size       = r.get<uint64_t>("ibf.bin_size (rows S)");
    m.hash_shift     = r.get<uint64_t>("ibf.hash_shift");
    m.bin_words      = r.get<uint64_t>("ibf.bin_words (W)");
    m.hash_funs      = r.get<uint64_t>("ibf.hash_funs");
    r.check("bin_words == ceil(bins / 64)", m.bin_words == ((m.bins + 63) >> 6), std::to_string((m.bins + 63) >> 6));
    r.check("technical_bins == 64 * bin_words", m.technical_bins == m.bin_words * 64, std::to_string(m.bin_words * 64));
    r.check("hash_shift == countl_zero(bin_size)", m.bin_size != 0 && m.hash_shift == (uint64_t)__builtin_clzll(m.bin_size),
            m.bin_size ? std::to_string(__builtin_clzll(m.bin_size)) : std::string("bin_size 0"));
    r.check("hash_funs in 1..5", m.hash_funs >= 1 && m.hash_funs <= 5);
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
    if (r.trace && exact_len >= 0)
        r.note(r.pos, "bytes between the IBF fields and the payload", exact_len, "(file size - S*W*8 - offset: the bit_vector header must be this long)");
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
        {
            if (r.trace)
            {
                std::string how = std::string(v.width ? "width byte, " : "") + (v.growth ? "growth factor (float), " : "") + "size (u64) in "
                                  + (size == bits ? "bits" : "64-bit words");
                r.note(start, "sdsl bit_vector header", how, "");
                r.note(start + (uint64_t)len - 8, "bit_vector.size", size);
                r.check("bit_vector size == technical_bins * bin_size", true, std::to_string(bits) + " bits");
            }
            return;
        }
        seen += " [width " + std::to_string(width) + ", size " + std::to_string(size) + "]";
    }
    r.seek(start);
    r.check("bit_vector size == technical_bins * bin_size", false, "candidates read:" + seen);
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
    const unsigned    nth    = (unsigned)std::min<uint64_t>(std::min(16u, hw), std::max<uint64_t>(1, bytes >> 21)); // >= 2 MiB each
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

thread_local LoadTiming g_load_timing;
double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
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
    // two staging buffers of 1/32 of the matrix each, 32 .. 256 MiB: page-locking them is what a small filter's load consists of
    // (2 x 256 MiB took 0.21 s of the 0.41 s a 1 GiB filter needed, profiles/r05_e2e_startup1.json); a 128 GiB filter still gets 256 MiB pieces
    const uint64_t want      = std::min<uint64_t>(total, std::min<uint64_t>(256ull << 20, std::max<uint64_t>(32ull << 20, total >> 5)));
    const uint64_t per       = std::max<uint64_t>(1, want / row_bytes); // rows per chunk
    const uint64_t cap       = per * row_bytes;
    const double   t_stage   = now_s();
    uint64_t*      stage[2]  = { sink.staging(0, cap), sink.staging(1, cap) };
    g_load_timing.staging_s += now_s() - t_stage;
    g_load_timing.payload_bytes += total;
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
        const double t0 = now_s();
        parallel_pread(fd, reinterpret_cast<uint8_t*>(buf), n * row_bytes, offset + row * row_bytes, path);
        const double t1 = now_s();
        g_load_timing.pread_s += t1 - t0;
        // the copy of the previous chunk ran while this one was read; it used the OTHER buffer, but the buffer filled
        // next is that one again, so it has to be finished before the next round
        if (!sink.drain(err) || !sink.rows(ibf, row, n, buf, err))
            throw std::runtime_error(path + ": " + err);
        g_load_timing.sink_s += now_s() - t1;
    }
    const double t2 = now_s();
    if (!sink.drain(err))
        throw std::runtime_error(path + ": " + err);
    g_load_timing.sink_s += now_s() - t2;
}

// everything of a .ibf but the bits (GanonClassify.cpp:949-986); returns the offset of the first payload byte
uint64_t parse_ibf(Reader& r, FilterMeta& out)
{
    const std::string& path = r.path;
    out.is_hibf = false;
    int version[3];
    const uint64_t at0 = r.pos;
    r.raw(version, sizeof(version)); // std::tuple<int,int,int>
    r.note(at0, "version (tuple<int,int,int>)", std::to_string(version[0]) + "." + std::to_string(version[1]) + "." + std::to_string(version[2]));
    IBFConfig& c     = out.ibf_config;
    c.n_bins         = r.get<uint64_t>("IBFConfig.n_bins");
    c.max_hashes_bin = r.get<uint64_t>("IBFConfig.max_hashes_bin");
    c.hash_functions = r.get<uint8_t>("IBFConfig.hash_functions");
    c.kmer_size      = r.get<uint8_t>("IBFConfig.kmer_size");
    c.window_size    = r.get<uint16_t>("IBFConfig.window_size");
    c.bin_size_bits  = r.get<uint64_t>("IBFConfig.bin_size_bits");
    c.max_fp         = r.get<double>("IBFConfig.max_fp");
    c.true_max_fp    = r.get<double>("IBFConfig.true_max_fp");
    c.true_avg_fp    = r.get<double>("IBFConfig.true_avg_fp");

    std::vector<std::pair<std::string, uint64_t>> hashes_count;
    const uint64_t                                at_hc = r.pos;
    const uint64_t                                nhc   = r.count(16);
    for (uint64_t i = 0; i < nhc; ++i)
    {
        std::string t = r.str();
        uint64_t    n = r.get<uint64_t>();
        hashes_count.emplace_back(std::move(t), n);
    }
    r.note(at_hc, "hashes_count (vector<tuple<string,u64>>)", nhc,
           nhc ? ("entries, first: \"" + hashes_count.front().first + "\" -> " + std::to_string(hashes_count.front().second)).c_str() : "entries");
    std::vector<std::pair<uint64_t, std::string>> bin_map;
    const uint64_t                                at_bm = r.pos;
    const uint64_t                                nbm   = r.count(16);
    for (uint64_t i = 0; i < nbm; ++i)
    {
        uint64_t    b = r.get<uint64_t>();
        std::string t = r.str();
        bin_map.emplace_back(b, std::move(t));
    }
    r.note(at_bm, "bin_map (vector<tuple<u64,string>>)", nbm,
           nbm ? ("entries, first: bin " + std::to_string(bin_map.front().first) + " -> \"" + bin_map.front().second + "\"").c_str() : "entries");
    IbfShape m;
    read_ibf_fields(r, m);
    // the payload is the last thing in the file: whatever precedes it is the bit_vector header
    const uint64_t payload = m.payload_bytes();
    r.check("payload S*W*8 (+ an 8..13 byte header) fits the bytes left", r.size >= r.pos + 8 + payload,
            std::to_string(payload) + " payload bytes, " + std::to_string(r.size - r.pos) + " bytes left");
    if (r.size < r.pos + 8 + payload)
        throw std::runtime_error(path + ": truncated (the IBF payload needs " + std::to_string(payload) + " bytes, the file has "
                                 + std::to_string(r.size - r.pos) + " left)");
    read_bitvector_header(r, m, (int64_t)(r.size - r.pos - payload));
    const uint64_t payload_at = r.pos;
    r.note(payload_at, "payload (S rows of W little-endian words)", payload, "bytes, to the end of the file");
    r.check("IBFConfig n_bins / bin_size_bits / hash_functions == the stored IBF's", m.bins == c.n_bins && m.bin_size == c.bin_size_bits && m.hash_funs == c.hash_functions);
    r.check("1 <= kmer_size <= 32, window_size >= kmer_size", !(c.kmer_size == 0 || c.kmer_size > 32 || c.window_size < c.kmer_size));
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
            r.check("every bin_map entry names a bin < bins", false, "bin " + std::to_string(binno));
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
    if (r.trace)
    {
        r.check("every bin_map entry names a bin < bins", true);
        uint64_t mapped = 0, no_count = 0;
        for (auto const& b : out.target_bins)
            mapped += b.size();
        for (auto const& t : out.targets)
            no_count += fpr.find(t) == fpr.end();
        r.note(at_bm, "targets (distinct names in bin_map)", out.targets.size(),
               (std::to_string(mapped) + " of " + std::to_string(m.bins) + " bins mapped; " + std::to_string(no_count)
                + " target(s) without a hashes_count entry (fpr 0 then, GanonClassify.cpp:533)").c_str());
    }
    return payload_at;
}

void load_ibf(const std::string& path, FilterMeta& out, FilterSink& sink)
{
    const double   t0 = now_s();
    Reader         r(path);
    const uint64_t payload_at = parse_ibf(r, out);
    const IbfShape m          = out.shapes.at(0);
    const double   t1 = now_s();
    g_load_timing.parse_s += t1 - t0;

    std::string err;
    if (!sink.begin(out, err))
        throw std::runtime_error(path + ": " + err);
    g_load_timing.begin_s += now_s() - t1;
    Fd fd(path);
    if (fd.fd < 0)
        throw std::runtime_error("cannot open filter file " + path);
    stream_matrix(path, fd.fd, payload_at, m, 0, sink);
    const double t_end = now_s();
    if (!sink.end(err))
        throw std::runtime_error(path + ": " + err);
    g_load_timing.end_s += now_s() - t_end;
}

// everything of a raptor .hibf but the bits (GanonClassify.cpp:875-938); payload_at[i] = first payload byte of IBF i
void parse_hibf(Reader& r, FilterMeta& out, std::vector<uint64_t>& payload_at)
{
    const std::string& path = r.path;
    out.is_hibf = true;
    (void)r.get<uint32_t>("raptor index: parsed_version");
    const uint64_t window_size = r.get<uint64_t>("window_size");
    const uint64_t shape_size  = r.get<uint64_t>("shape.size (dynamic_bitset<58>)"); // seqan3::shape: size, bits
    const uint64_t shape_bits  = r.get<uint64_t>("shape.bits");
    (void)shape_size;
    (void)r.get<uint8_t>("parts");
    (void)r.get<uint8_t>("compressed");
    std::vector<std::vector<std::string>> bin_path;
    const uint64_t                        at_bp = r.pos;
    const uint64_t                        nbp = r.count(8);
    bin_path.resize(nbp);
    for (auto& lst : bin_path)
    {
        const uint64_t n = r.count(8);
        lst.resize(n);
        for (auto& s : lst)
            s = r.str();
    }
    r.note(at_bp, "bin_path (vector<vector<string>>)", nbp, nbp && !bin_path[0].empty() ? ("lists, first: \"" + bin_path[0][0] + "\"").c_str() : "lists");
    const double fpr = r.get<double>("fpr");
    (void)r.get<uint8_t>("is_hibf");
    // hierarchical_interleaved_bloom_filter: ibf_vector, next_ibf_id, user_bins{user_bin_filenames, ibf_bin_to_filename_position}.
    // The tables FOLLOW the matrices: first pass over the IBF headers only (payloads skipped), matrices in a second pass.
    const uint64_t nibf = r.get<uint64_t>("ibf_vector.size");
    if (nibf > (r.size - r.pos) / 48)
        throw std::runtime_error(path + ": implausible container size " + std::to_string(nibf) + " at offset " + std::to_string(r.pos));
    if (nibf == 0)
        throw std::runtime_error(path + ": HIBF without IBFs");
    out.shapes.resize(nibf);
    payload_at.assign(nibf, 0);
    std::ostream* const trace = r.trace;
    for (uint64_t i = 0; i < nibf; ++i)
    {
        if (trace)
        {
            r.trace = i < 2 ? trace : nullptr; // (the first two IBFs in full, the rest summarised below)
            if (r.trace)
                *trace << "--- ibf_vector[" << i << "]\n";
        }
        read_ibf_fields(r, out.shapes[i]);
        read_bitvector_header(r, out.shapes[i], -1);
        payload_at[i] = r.pos;
        r.note(r.pos, "payload", out.shapes[i].payload_bytes(), "bytes");
        r.skip(out.shapes[i].payload_bytes());
    }
    r.trace = trace;
    if (trace)
    {
        uint64_t bytes = 0, bmin = UINT64_MAX, bmax = 0;
        for (auto const& m : out.shapes)
            bytes += m.payload_bytes(), bmin = std::min(bmin, m.bins), bmax = std::max(bmax, m.bins);
        *trace << "--- " << nibf << " IBFs parsed, " << bytes << " payload bytes, " << bmin << ".." << bmax << " bins each\n";
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
    const uint64_t           nub = r.get<uint64_t>("user_bin_filenames.size (user bins)");
    if (nub > (r.size - r.pos) / 8)
        throw std::runtime_error(path + ": implausible container size " + std::to_string(nub) + " at offset " + std::to_string(r.pos));
    std::vector<std::string> user_bin_filenames(nub);
    for (auto& s : user_bin_filenames)
        s = r.str();
    read_vv(out.bin_to_user);
    r.note(r.pos, "end of the archive", r.pos, "");
    r.check("no bytes left after user_bins", r.pos == r.size, std::to_string(r.size - r.pos) + " left");
    if (r.pos != r.size)
        throw std::runtime_error(path + ": " + std::to_string(r.size - r.pos) + " trailing bytes after the HIBF");
    r.check("next_ibf_id and ibf_bin_to_filename_position have one entry per IBF", out.next_ibf_id.size() == nibf && out.bin_to_user.size() == nibf);
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
    if (r.trace)
    {
        r.check("per-bin tables cover every IBF's bins; bin_path entries name user bins", true);
        r.note(at_bp, "targets (distinct names in bin_path)", out.targets.size(), ("k = popcount(shape) = " + std::to_string((unsigned)out.ibf_config.kmer_size)).c_str());
    }
}

void load_hibf(const std::string& path, FilterMeta& out, FilterSink& sink)
{
    Reader                r(path);
    std::vector<uint64_t> payload_at;
    const double t0 = now_s();
    parse_hibf(r, out, payload_at);
    const uint64_t nibf = out.shapes.size();
    const double   t1   = now_s();
    g_load_timing.parse_s += t1 - t0;

    std::string err;
    if (!sink.begin(out, err))
        throw std::runtime_error(path + ": " + err);
    g_load_timing.begin_s += now_s() - t1;
    Fd fd(path);
    if (fd.fd < 0)
        throw std::runtime_error("cannot open filter file " + path);
    for (uint64_t i = 0; i < nibf; ++i)
        stream_matrix(path, fd.fd, payload_at[i], out.shapes[i], (uint32_t)i, sink);
    const double t_end = now_s();
    if (!sink.end(err))
        throw std::runtime_error(path + ": " + err);
    g_load_timing.end_s += now_s() - t_end;
}

} // namespace

double false_positive(uint64_t bin_size_bits, uint8_t hash_functions, uint64_t n_hashes)
{
    return std::pow(1 - std::exp(-hash_functions / (bin_size_bits / static_cast<double>(n_hashes))), hash_functions);
}

const LoadTiming& last_load_timing()
{
    return g_load_timing;
}

void load_filter_file(const std::string& path, bool hibf, FilterMeta& meta, FilterSink& sink)
{
    g_load_timing = LoadTiming();
    meta = FilterMeta();
    if (hibf)
        load_hibf(path, meta, sink);
    else
        load_ibf(path, meta, sink);
}

bool inspect_filter_file(const std::string& path, bool hibf, std::ostream& out)
{
    FilterMeta meta;
    try
    {
        Reader r(path);
        r.trace = &out;
        out << "file        " << path << " (" << r.size << " bytes), read as " << (hibf ? "a raptor .hibf (--hibf)" : "a ganon-build .ibf")
            << "; cereal binary archive, little-endian\n";
        if (hibf)
        {
            std::vector<uint64_t> payload_at;
            parse_hibf(r, meta, payload_at);
        }
        else
            parse_ibf(r, meta);
    }
    catch (const std::exception& e)
    {
        out << "result      INCONSISTENT -- first inconsistency: " << e.what() << "\n";
        return false;
    }
    uint64_t bytes = 0;
    for (auto const& m : meta.shapes)
        bytes += m.payload_bytes();
    out << "result      CONSISTENT: " << (hibf ? "HIBF" : "IBF") << ", k=" << (unsigned)meta.ibf_config.kmer_size << " w=" << meta.ibf_config.window_size << ", "
        << meta.shapes.size() << " IBF(s), " << meta.bin_count << (hibf ? " user bins, " : " bins, ") << meta.targets.size() << " targets, " << bytes
        << " bytes of bits (not read)\n";
    return true;
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
