// reassign.cpp -- `ganon reassign`: what /root/reference/src/ganon/reassign.py does with a classification's .rep and .all
// files, with the EM (:96-145) and the final choice (:153-181) on the device (csrc/gn_reassign.hip, gn_reassign_*).
//
// The host's part is the text: which tables a .rep names (:35-59), a table's reads / targets / entries in the order the
// reference's dicts would hold them (:76-92), the .one lines (:153-181), the rewritten .rep (:189-219), the log lines
// (print_log -> stderr).  No CPU fallback: without a device gn_reassign_create fails and so does the run.
//
// Differences a user cannot see unless the input is damaged: Python opens the files in text mode (a lone '\r' ends a line
// there, not here) and raises a traceback where this throws a one-line error; rep files found for a prefix are visited in
// sorted order (the reference takes the directory's own order).
#include "reassign.hpp"

#include "ganon_hip.h"

#include <algorithm>
#include <charconv>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string_view>
#include <thread>
#include <unordered_map>

namespace fs = std::filesystem;

namespace gnhost
{

namespace
{

// ---- Python's string forms ---------------------------------------------------------------------------------------------
bool py_space(unsigned char c) // str.rstrip() without arguments, the one-byte part of it
{
    return c == ' ' || (c >= 0x09 && c <= 0x0d) || (c >= 0x1c && c <= 0x1f);
}

std::string_view rstrip(std::string_view s)
{
    while (!s.empty() && py_space((unsigned char)s.back()))
        s.remove_suffix(1);
    return s;
}

std::vector<std::string_view> split_tabs(std::string_view s)
{
    std::vector<std::string_view> out;
    size_t                        at = 0;
    for (;;)
    {
        const size_t t = s.find('\t', at);
        if (t == std::string_view::npos)
        {
            out.push_back(s.substr(at));
            return out;
        }
        out.push_back(s.substr(at, t - at));
        at = t + 1;
    }
}

// int(text): optional blanks, optional sign, decimal digits
bool py_int(std::string_view s, long long& v)
{
    while (!s.empty() && py_space((unsigned char)s.front()))
        s.remove_prefix(1);
    s = rstrip(s);
    bool neg = false;
    if (!s.empty() && (s.front() == '+' || s.front() == '-'))
    {
        neg = s.front() == '-';
        s.remove_prefix(1);
    }
    if (s.empty() || s.size() > 18)
        return false;
    long long x = 0;
    for (char c : s)
    {
        if (c < '0' || c > '9')
            return false;
        x = x * 10 + (c - '0');
    }
    v = neg ? -x : x;
    return true;
}

// str(PurePosixPath(p)): '.' components and repeated slashes go, '..' stays
std::string py_path(const std::string& p)
{
    if (p.empty())
        return ".";
    std::string out;
    const bool  abs = p[0] == '/';
    const bool  two = p.size() >= 2 && p[1] == '/' && (p.size() == 2 || p[2] != '/'); // exactly two leading slashes are kept
    size_t      at  = 0;
    while (at <= p.size())
    {
        size_t e = p.find('/', at);
        if (e == std::string::npos)
            e = p.size();
        const std::string_view part(p.data() + at, e - at);
        if (!part.empty() && part != ".")
        {
            if (!out.empty())
                out += '/';
            out.append(part);
        }
        at = e + 1;
    }
    if (abs)
        out = (two ? "//" : "/") + out;
    return out.empty() ? "." : out;
}

bool check_file(const std::string& p) // util.py:118-125
{
    std::error_code ec;
    return fs::is_regular_file(p, ec) && fs::file_size(p, ec) > 0 && !ec;
}

std::string slurp(const std::string& path)
{
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f)
        throw std::runtime_error("cannot open " + path);
    std::string s;
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (n > 0)
    {
        s.resize((size_t)n);
        if (std::fread(s.data(), 1, (size_t)n, f) != (size_t)n)
        {
            std::fclose(f);
            throw std::runtime_error("short read of " + path);
        }
    }
    std::fclose(f);
    return s;
}

template <typename F>
void each_line(const std::string& text, F&& f) // the line WITH its '\n', as Python's file iteration hands it out
{
    size_t at = 0;
    while (at < text.size())
    {
        size_t e = text.find('\n', at);
        e        = e == std::string::npos ? text.size() : e + 1;
        f(std::string_view(text.data() + at, e - at));
        at = e;
    }
}

struct Table // :76-92
{
    std::string                                    text;
    std::vector<std::string_view>                  read_ids, target_names;
    std::unordered_map<std::string_view, uint32_t> target_index;
    std::vector<uint64_t>                          off;
    std::vector<uint32_t>                          target;
    std::vector<long long>                         count;
};

// The table of one .all file (:76-92).  The lines are parsed by several threads -- chunks of the text cut at line ends; per chunk the
// fields, the count, the target's number among the CHUNK's targets, and the runs of consecutive lines with one read id -- and put
// together in chunk order by one: targets numbered by first appearance over the chunks in order, one hash-map insert per RUN (a read's
// lines follow each other in what ganon-classify writes, so that is one per read, the same the one-thread version paid; a run whose id
// was seen before makes the table "not grouped" and its entries are gathered behind the read's earlier ones, as Python's dict of lists
// would hold them).
void read_table(const std::string& path, Table& tb)
{
    tb.text = slurp(path);
    const std::string& text = tb.text;
    struct Run
    {
        std::string_view id;
        size_t           lines;
    };
    struct Chunk
    {
        size_t                                         begin = 0, end = 0, n_lines = 0;
        std::vector<std::string_view>                  targets;
        std::unordered_map<std::string_view, uint32_t> index;
        std::vector<uint32_t>                          tgt; // per line: number among this chunk's targets
        std::vector<long long>                         cnt;
        std::vector<Run>                               runs;
        std::string                                    error; // first line of the chunk that is no entry
        size_t                                         error_line = 0;
    };
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t   nt = std::max<size_t>(1, std::min<size_t>({ (size_t)hw, 16, text.size() / (1u << 20) + 1 }));
    std::vector<Chunk> ch(nt);
    for (size_t c = 0; c < nt; ++c)
    {
        size_t b = c == 0 ? 0 : ch[c - 1].end;
        size_t e = c + 1 == nt ? text.size() : std::max(b, text.size() / nt * (c + 1));
        if (e < text.size())
        {
            const size_t nl = text.find('\n', e);
            e               = nl == std::string::npos ? text.size() : nl + 1;
        }
        ch[c].begin = b;
        ch[c].end   = e;
    }
    auto parse = [&](Chunk& k) {
        const size_t guess = (k.end - k.begin) / 24 + 16;
        k.tgt.reserve(guess);
        k.cnt.reserve(guess);
        k.runs.reserve(guess / 2);
        size_t at = k.begin;
        while (at < k.end)
        {
            size_t e = text.find('\n', at);
            e        = e == std::string::npos || e >= k.end ? k.end : e + 1;
            const std::string_view raw(text.data() + at, e - at);
            at = e;
            ++k.n_lines;
            const std::string_view line = rstrip(raw);
            const size_t           t1   = line.find('\t');
            const size_t           t2   = t1 == std::string_view::npos ? t1 : line.find('\t', t1 + 1);
            long long              c    = 0;
            if (t2 == std::string_view::npos || line.find('\t', t2 + 1) != std::string_view::npos)
                k.error = " is not `read <tab> target <tab> count`";
            else if (!py_int(line.substr(t2 + 1), c))
                k.error = ": the count is no integer";
            if (!k.error.empty())
            {
                k.error_line = k.n_lines;
                return;
            }
            const std::string_view rid = line.substr(0, t1), tname = line.substr(t1 + 1, t2 - t1 - 1);
            if (!k.runs.empty() && k.runs.back().id == rid)
                ++k.runs.back().lines;
            else
                k.runs.push_back(Run{ rid, 1 });
            auto [tt, fresh] = k.index.try_emplace(tname, (uint32_t)k.targets.size());
            if (fresh)
                k.targets.push_back(tname);
            k.tgt.push_back(tt->second);
            k.cnt.push_back(c);
        }
    };
    {
        std::vector<std::thread> th;
        for (size_t c = 1; c < nt; ++c)
            th.emplace_back([&, c] { parse(ch[c]); });
        parse(ch[0]);
        for (auto& t : th)
            t.join();
    }
    size_t lines_before = 0, n = 0;
    for (auto& k : ch)
    {
        if (!k.error.empty())
            throw std::runtime_error(path + ": line " + std::to_string(lines_before + k.error_line) + k.error);
        lines_before += k.n_lines;
        n += k.tgt.size();
    }
    // targets in first-appearance order; the chunks' numbers become the table's
    std::vector<std::vector<uint32_t>> remap(nt);
    for (size_t c = 0; c < nt; ++c)
    {
        remap[c].resize(ch[c].targets.size());
        for (size_t t = 0; t < ch[c].targets.size(); ++t)
        {
            auto [it, fresh] = tb.target_index.try_emplace(ch[c].targets[t], (uint32_t)tb.target_names.size());
            if (fresh)
                tb.target_names.push_back(ch[c].targets[t]);
            remap[c][t] = it->second;
        }
    }
    // reads: one insert per run
    std::unordered_map<std::string_view, uint32_t> reads;
    size_t                                         n_runs = 0;
    for (auto& k : ch)
        n_runs += k.runs.size();
    reads.reserve(n_runs);
    tb.read_ids.reserve(n_runs);
    std::vector<uint32_t> line_read(n);
    bool                  grouped = true; // every read's lines follow each other: the table is in CSR order as it stands
    size_t                at      = 0;
    std::string_view      last_id;
    uint32_t              last_read = 0;
    for (auto& k : ch)
        for (const Run& run : k.runs)
        {
            uint32_t r;
            if (!tb.read_ids.empty() && run.id == last_id) // (a read whose lines straddle two chunks)
                r = last_read;
            else
            {
                auto [it, fresh] = reads.try_emplace(run.id, (uint32_t)tb.read_ids.size());
                if (fresh)
                {
                    if (tb.read_ids.size() >= 0xffffffffull)
                        throw std::runtime_error(path + ": more than 2^32 - 1 reads");
                    tb.read_ids.push_back(run.id);
                }
                else
                    grouped = false;
                r = it->second;
            }
            last_id   = run.id;
            last_read = r;
            std::fill(line_read.begin() + at, line_read.begin() + at + run.lines, r);
            at += run.lines;
        }
    // the entries, in file order
    tb.target.resize(n);
    tb.count.resize(n);
    {
        std::vector<size_t> first(nt, 0);
        for (size_t c = 1; c < nt; ++c)
            first[c] = first[c - 1] + ch[c - 1].tgt.size();
        auto place = [&](size_t c) {
            for (size_t i = 0; i < ch[c].tgt.size(); ++i)
            {
                tb.target[first[c] + i] = remap[c][ch[c].tgt[i]];
                tb.count[first[c] + i]  = ch[c].cnt[i];
            }
        };
        std::vector<std::thread> th;
        for (size_t c = 1; c < nt; ++c)
            th.emplace_back(place, c);
        place(0);
        for (auto& t : th)
            t.join();
    }
    const size_t n_reads = tb.read_ids.size();
    tb.off.assign(n_reads + 1, 0);
    for (uint32_t r : line_read)
        ++tb.off[r + 1];
    for (size_t r = 0; r < n_reads; ++r)
        tb.off[r + 1] += tb.off[r];
    if (!grouped) // a read listed in two places is one read (:83-85): its entries in file order, stable
    {
        std::vector<uint64_t>  cur(tb.off.begin(), tb.off.end() - 1);
        std::vector<uint32_t>  t2(n);
        std::vector<long long> c2(n);
        for (size_t i = 0; i < n; ++i)
        {
            const uint64_t o = cur[line_read[i]]++;
            t2[o]            = tb.target[i];
            c2[o]            = tb.count[i];
        }
        tb.target.swap(t2);
        tb.count.swap(c2);
    }
}

struct GnError : std::runtime_error
{
    using std::runtime_error::runtime_error;
};

void gn(int rc, const char* what)
{
    if (rc != GN_OK)
        throw GnError(std::string(what) + ": " + gn_last_error());
}

struct Em
{
    std::vector<uint64_t> counts, choice;
    std::vector<double>   diffs;
    uint64_t              n_multi = 0;
    float                 ms_device = 0.f;
    uint64_t              bytes_per_iteration = 0;
};

void run_em(const ReassignConfig& cfg, const Table& tb, Em& em)
{
    gn_reassign* g = nullptr;
    gn(gn_reassign_create(cfg.device, tb.read_ids.size(), tb.target.size(), (uint32_t)tb.target_names.size(), tb.off.data(),
                          tb.target.data(), &g),
       "gn_reassign_create");
    try
    {
        uint32_t iterations = 0;
        gn(gn_reassign_run(g, cfg.max_iter, cfg.threshold, &iterations), "gn_reassign_run");
        em.diffs.resize(iterations);
        em.counts.resize(tb.target_names.size());
        em.choice.resize(tb.read_ids.size());
        gn(gn_reassign_fetch(g, em.counts.data(), nullptr, nullptr, em.choice.data()), "gn_reassign_fetch");
        gn(gn_reassign_diffs(g, em.diffs.data(), iterations), "gn_reassign_diffs");
        gn(gn_reassign_info(g, nullptr, &em.n_multi, nullptr, &em.ms_device, &em.bytes_per_iteration), "gn_reassign_info");
    }
    catch (...)
    {
        gn_reassign_free(g);
        throw;
    }
    gn_reassign_free(g);
}

void log(const ReassignConfig& cfg, const std::string& s) // util.py:52-55
{
    if (!cfg.quiet)
    {
        std::cerr << s << '\n';
        std::cerr.flush();
    }
}

std::vector<std::string> find_rep_files(const std::string& ip) // util.py:174-179
{
    std::vector<std::string> out;
    std::error_code          ec;
    fs::path                 dir;
    std::string              name;
    if (fs::is_directory(ip, ec))
        dir = ip;
    else
    {
        const fs::path p(ip);
        dir  = p.parent_path().empty() ? fs::path(".") : p.parent_path();
        name = p.filename().string();
    }
    if (!fs::is_directory(dir, ec))
        return out;
    for (const auto& e : fs::directory_iterator(dir, ec))
    {
        const std::string fn = e.path().filename().string();
        if (fn.size() >= name.size() + 4 && fn.compare(0, name.size(), name) == 0 && fn.compare(fn.size() - 4, 4, ".rep") == 0)
            out.push_back(py_path((dir / fn).string()));
    }
    std::sort(out.begin(), out.end());
    return out;
}

void write_file(const std::string& path, const std::string& text)
{
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (!f)
        throw std::runtime_error("cannot write " + path);
    const bool ok = text.empty() || std::fwrite(text.data(), 1, text.size(), f) == text.size();
    if (std::fclose(f) != 0 || !ok)
        throw std::runtime_error("cannot write " + path);
}

} // namespace

std::string py_round6_repr(double x)
{
    // round(x, 6): the correctly rounded six-decimal form read back as a double (CPython's double_round), then repr():
    // shortest digits that read back, fixed notation for decimal exponents -4 .. 15, else d.ddde-XX
    char buf[64];
    std::snprintf(buf, sizeof(buf), "%.6f", x);
    const double r = std::strtod(buf, nullptr);
    if (r == 0.0)
        return std::signbit(r) ? "-0.0" : "0.0";
    char       sci[64];
    const auto res = std::to_chars(sci, sci + sizeof(sci), r, std::chars_format::scientific);
    std::string_view s(sci, (size_t)(res.ptr - sci));
    std::string out;
    if (s.front() == '-')
    {
        out = "-";
        s.remove_prefix(1);
    }
    const size_t     e = s.find('e');
    std::string      digits;
    for (char c : s.substr(0, e))
        if (c != '.')
            digits += c;
    const int exp10  = std::atoi(std::string(s.substr(e + 1)).c_str());
    const int decpt  = exp10 + 1;
    const int nd     = (int)digits.size();
    if (decpt <= -4 || decpt > 16)
    {
        out += digits[0];
        if (nd > 1)
            out += "." + digits.substr(1);
        char eb[16];
        std::snprintf(eb, sizeof(eb), "e%c%02d", exp10 < 0 ? '-' : '+', std::abs(exp10));
        return out + eb;
    }
    if (decpt <= 0)
        return out + "0." + std::string((size_t)-decpt, '0') + digits;
    if (decpt >= nd)
        return out + digits + std::string((size_t)(decpt - nd), '0') + ".0";
    return out + digits.substr(0, (size_t)decpt) + "." + digits.substr((size_t)decpt);
}

bool run_reassign(const ReassignConfig& cfg)
{
    using clk = std::chrono::steady_clock;
    log(cfg, "Reassigning reads");
    log(cfg, "");
    std::vector<std::string> rep_files;
    for (const auto& ip : cfg.input_prefix)
    {
        auto found = find_rep_files(ip);
        rep_files.insert(rep_files.end(), found.begin(), found.end());
    }
    for (const std::string& rep_file : rep_files)
    {
        const fs::path    rp(rep_file);
        const std::string stem            = rp.stem().string();
        const std::string rep_file_prefix = py_path((rp.parent_path() / stem).string());
        const std::string out_file_prefix =
            cfg.output_prefix.empty() ? rep_file_prefix : (rep_files.size() == 1 ? cfg.output_prefix : cfg.output_prefix + stem);
        const std::string rep_file_out = cfg.skip_rep ? "" : out_file_prefix + ".rep";

        // :35-59 -- the hierarchies of the report and their tables
        std::vector<std::pair<std::string, std::string>> all_files; // (hierarchy, .all path) in first-appearance order
        std::vector<std::string>                         info;
        if (!check_file(rep_file))
        {
            log(cfg, "No .rep/.all file(s) found with prefix --input-prefix " + rep_file_prefix);
            return false;
        }
        log(cfg, "Ganon report output found: " + rep_file);
        const std::string rep_text = slurp(rep_file);
        each_line(rep_text, [&](std::string_view line) {
            if (line[0] != '#')
            {
                const std::string h(line.substr(0, line.find('\t')));
                if (std::none_of(all_files.begin(), all_files.end(), [&](const auto& x) { return x.first == h; }))
                    all_files.emplace_back(h, "");
            }
            else
                info.emplace_back(rstrip(line));
        });
        for (auto& [h, af] : all_files)
        {
            if (check_file(rep_file_prefix + "." + h + ".all"))
                af = rep_file_prefix + "." + h + ".all";
            else if (check_file(rep_file_prefix + ".all"))
            {
                all_files.assign(1, { std::string(), rep_file_prefix + ".all" }); // --output-single: one table for every row
                break;
            }
            else
            {
                log(cfg, "No matching files for given .rep [" + rep_file_prefix + "*.all]");
                return false;
            }
        }

        std::string new_rep;
        for (const auto& [hierarchy, af] : all_files)
        {
            log(cfg, af + (hierarchy.empty() ? "" : " [" + hierarchy + "]"));
            const auto t0 = clk::now();
            Table      tb;
            read_table(af, tb);
            const auto t1 = clk::now();
            Em         em;
            run_em(cfg, tb, em);
            const auto t2 = clk::now();
            for (size_t i = 0; i < em.diffs.size(); ++i)
                log(cfg, " - Iteration " + std::to_string(i + 1) + " (" + py_round6_repr(em.diffs[i]) + ")");

            if (!cfg.skip_one) // :148-187
            {
                const std::string one_file_out = all_files.size() == 1 ? out_file_prefix + ".one" : out_file_prefix + "." + hierarchy + ".one";
                std::string       text;
                text.reserve(tb.text.size() / 2 + 64);
                char num[32];
                for (size_t r = 0; r < tb.read_ids.size(); ++r)
                {
                    const uint64_t e = em.choice[r];
                    text.append(tb.read_ids[r]);
                    text += '\t';
                    text.append(tb.target_names[tb.target[e]]);
                    text += '\t';
                    const auto res = std::to_chars(num, num + sizeof(num), tb.count[e]);
                    text.append(num, (size_t)(res.ptr - num));
                    text += '\n';
                }
                write_file(one_file_out, text);
                log(cfg, " - " + std::to_string(em.n_multi) + " reassigned reads to " + one_file_out);
            }
            const auto t3 = clk::now();
            if (cfg.verbose && !cfg.quiet)
            {
                auto s = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
                char line[320];
                std::snprintf(line, sizeof(line),
                              "[reassign] %zu reads, %zu entries, %zu targets; read table %.3f s, device (upload + EM + fetch) %.3f s of which "
                              "kernels %.3f ms (%.1f GB/s over %zu iterations), write .one %.3f s",
                              tb.read_ids.size(), tb.target.size(), tb.target_names.size(), s(t0, t1), s(t1, t2), (double)em.ms_device,
                              em.ms_device > 0 ? (double)em.bytes_per_iteration * (double)(em.diffs.size() + 1) / ((double)em.ms_device * 1e6) : 0.0,
                              em.diffs.size(), s(t2, t3));
                std::cerr << line << '\n';
            }

            if (!rep_file_out.empty()) // :189-219
            {
                each_line(rep_text, [&](std::string_view raw) {
                    if (raw[0] == '#')
                        return;
                    const auto fld = split_tabs(rstrip(raw));
                    if (fld.size() < 4)
                        throw std::runtime_error(rep_file + ": a data row of fewer than four fields");
                    long long unique = 0;
                    if (!py_int(fld[3], unique))
                        throw std::runtime_error(rep_file + ": the unique column is no integer");
                    if (!hierarchy.empty() && fld[0] != hierarchy)
                        return;
                    const auto it = tb.target_index.find(fld[1]);
                    if (it == tb.target_index.end())
                        return;
                    new_rep.append(fld[0]).append("\t").append(fld[1]).append("\t").append(fld[2]).append("\t");
                    new_rep += std::to_string(unique) + "\t" + std::to_string((long long)em.counts[it->second] - unique) + "\t";
                    if (fld.size() >= 6)
                        new_rep.append(fld[5]);
                    new_rep += '\t';
                    if (fld.size() >= 7)
                        new_rep.append(fld[6]);
                    new_rep += '\n';
                });
            }
        }
        if (!rep_file_out.empty())
        {
            for (const auto& i : info)
                new_rep += i + "\n";
            write_file(rep_file_out, new_rep);
            log(cfg, "New .rep file: " + rep_file_out);
        }
        if (cfg.remove_all)
            for (const auto& x : all_files)
                std::remove(x.second.c_str());
    }
    return true;
}

} // namespace gnhost
