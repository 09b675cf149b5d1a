// backend_oracle.cpp -- TEST-ONLY Backend built on the CPU oracle (oracle/ganon_oracle.c).
// Lets the CPU test-suite run the host pipeline (CLI, file formats, readers, post-processing, writers) of
// ganon_amd/host end to end on the reference's known-answer scenarios without a GPU.  It is compiled only into
// tests/host_oracle/ganon-classify-oracle; the product binary links ganon_amd/host/backend_hip.cpp instead.
#include "../../ganon_amd/host/backend.hpp"
#include "../../oracle/ganon_oracle.h"
#include "../../ganon_amd/host/config.hpp"

#include <iostream>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>

namespace gnhost
{
namespace
{
class OracleBackend final : public Backend
{
public:
    // FilterSink: the matrices are assembled in host memory, chunk by chunk
    bool begin(const FilterMeta& f, std::string&) override
    {
        Held h;
        h.is_hibf = f.is_hibf;
        for (auto const& m : f.shapes)
        {
            h.mats.emplace_back((size_t)(m.bin_size * m.bin_words), 0ull);
            h.shapes.push_back(m);
        }
        if (f.is_hibf)
        {
            h.next = f.next_ibf_id;
            h.b2u  = f.bin_to_user;
            h.n_user_bins = f.n_user_bins;
        }
        h.off.push_back(0);
        for (auto const& b : f.target_bins)
        {
            for (auto x : b)
                h.bins.push_back((uint32_t)x);
            h.off.push_back((uint32_t)h.bins.size());
        }
        h.n_targets = (uint32_t)f.targets.size();
        held_.push_back(std::move(h));
        return true;
    }
    uint64_t* staging(int, size_t) override { return nullptr; }
    bool      rows(uint32_t ibf, uint64_t row_begin, uint64_t n_rows, const uint64_t* src, std::string&) override
    {
        Held& h = held_.back();
        std::copy(src, src + n_rows * h.shapes[ibf].bin_words, h.mats[ibf].begin() + row_begin * h.shapes[ibf].bin_words);
        return true;
    }
    bool drain(std::string&) override { return true; }
    bool end(std::string&) override
    {
        Held& h = held_.back();
        for (size_t i = 0; i < h.mats.size(); ++i)
        {
            const IbfShape& m = h.shapes[i];
            if (m.bins & 63) // padding bins are never reported (counting_vector has `bins` entries)
                for (uint64_t r = 0; r < m.bin_size; ++r)
                    h.mats[i][r * m.bin_words + m.bin_words - 1] &= (1ull << (m.bins & 63)) - 1ull;
            h.ibfs.push_back(gno_ibf{ h.mats[i].data(), m.bins, m.bin_size, m.bin_words, m.hash_shift, (uint32_t)m.hash_funs });
        }
        return true;
    }
    void clear_filters() override
    {
        if (!origin_)
            held_.clear();
    }

    // The checker can stand in for the product backend's optional features, so that the host pipeline's code for them (raw FASTQ
    // pieces accepted in file order, several batch contexts per worker thread) runs in the CPU tests too:
    //   $GANON_HOST_DEVICE_FASTQ=1   raw pieces, records found here by the slab parser's rule (scalar restatement)
    //   twin()                        a second context on the same filters
    bool tokenises_fastq() const override
    {
        const char* e = std::getenv("GANON_HOST_DEVICE_FASTQ");
        return e && e[0] == '1';
    }
    std::unique_ptr<Backend> twin() override
    {
        auto t     = std::make_unique<OracleBackend>();
        t->origin_ = origin_ ? origin_ : this;
        return t;
    }
    // the records of one text by the slab parser's rule (four-line FASTQ) or the two-line FASTA rule; -> first byte behind them
    static size_t find_records(const uint8_t* t, size_t n, int format, std::vector<uint32_t>& rec, std::vector<uint32_t>& seq,
                               std::vector<uint32_t>& len_out)
    {
        const bool    fasta_text = format != 0;      // two lines per record
        const uint8_t hdr        = format == 1 ? '>' : '@';
        static const LegalLetters legal;
        rec.clear();
        seq.clear();
        len_out.clear();
        size_t pos = 0;
        auto line_end = [&](size_t p) -> size_t { // index of the '\n' that ends the line at p, or n
            const void* q = p < n ? std::memchr(t + p, '\n', n - p) : nullptr;
            return q ? (size_t)((const uint8_t*)q - t) : n;
        };
        while (pos < n && fasta_text) // two-line records: >id / letters, then a '>' or the end of the text
        {
            const size_t a = line_end(pos);
            if (a == n || a == pos || t[pos] != hdr)
                break;
            const size_t bnl = line_end(a + 1);
            if (bnl == n)
                break;
            size_t len = bnl - a - 1;
            if (len && t[bnl - 1] == '\r')
                --len;
            bool ok = !(bnl > a + 1 && (t[a + 1] == '>' || t[a + 1] == ';')) && (bnl + 1 >= n || t[bnl + 1] == hdr);
            for (size_t i = 0; i < len && ok; ++i)
                ok = legal.ok[t[a + 1 + i]];
            if (!ok)
                break;
            rec.push_back((uint32_t)pos);
            seq.push_back((uint32_t)(a + 1));
            len_out.push_back((uint32_t)len);
            pos = bnl + 1;
        }
        while (pos < n && !fasta_text)
        {
            const size_t a = line_end(pos);
            if (a == n || a == pos || t[pos] != '@')
                break;
            const size_t bnl = line_end(a + 1);
            if (bnl == n)
                break;
            size_t len = bnl - a - 1;
            if (len && t[bnl - 1] == '\r')
                --len;
            bool ok = true;
            for (size_t i = 0; i < len && ok; ++i)
                ok = legal.ok[t[a + 1 + i]];
            if (!ok)
                break;
            const size_t c = line_end(bnl + 1);
            if (c == n || c == bnl + 1 || t[bnl + 1] != '+')
                break;
            const size_t d = line_end(c + 1);
            if (d == n || d - c - 1 != len)
                break;
            rec.push_back((uint32_t)pos);
            seq.push_back((uint32_t)(a + 1));
            len_out.push_back((uint32_t)len);
            pos = d + 1;
        }
        return pos;
    }

    bool tokenise(ReadBatch& b, uint32_t& n_reads, uint64_t& parsed_bytes, std::string&) override
    {
        const int fmt = b.raw_fasta ? 1 : 0;
        size_t    pos = find_records(b.text.data(), b.text.size(), fmt, tok_rec_, tok_seq_, tok_len_);
        if (b.paired) // the pairs BOTH texts hold before either's first non-record
        {
            std::vector<uint32_t> rec2;
            size_t                pos2 = find_records(b.text2.data(), b.text2.size(), fmt, rec2, tok_seq2_, tok_len2_);
            const size_t          v    = std::min(tok_rec_.size(), rec2.size());
            if (v < tok_rec_.size())
                pos = tok_rec_[v];
            if (v < rec2.size())
                pos2 = rec2[v];
            tok_rec_.resize(v);
            tok_seq_.resize(v);
            tok_len_.resize(v);
            tok_seq2_.resize(v);
            tok_len2_.resize(v);
            b.raw_parsed2 = pos2;
        }
        n_reads      = (uint32_t)tok_rec_.size();
        parsed_bytes = pos;
        return true;
    }

    bool classify(ReadBatch& b, uint32_t k, uint32_t w, const std::vector<double>& rel_cutoff, BatchResult& out,
                  std::string& err) override
    {
        // $GANON_TEST_FAIL_AT_BATCH=n: the n-th batch of the run fails (the pipeline's way down: no hang, the message on stderr)
        static std::atomic<long> seen{ 0 };
        if (const char* e = std::getenv("GANON_TEST_FAIL_AT_BATCH"))
            if (++seen == std::atol(e))
            {
                err = "injected failure (GANON_TEST_FAIL_AT_BATCH)";
                return false;
            }
        std::vector<Held>& held_ = origin_ ? origin_->held_ : this->held_;
        if (b.raw) // the first raw_keep records found by tokenise(): described for the pipeline, then classified like parsed reads
        {
            b.rec_at.assign(tok_rec_.begin(), tok_rec_.begin() + b.raw_keep);
            b.seq_at.assign(tok_seq_.begin(), tok_seq_.begin() + b.raw_keep);
            b.seq_len.assign(tok_len_.begin(), tok_len_.begin() + b.raw_keep);
            if (b.paired)
            {
                b.seq_at2.assign(tok_seq2_.begin(), tok_seq2_.begin() + b.raw_keep);
                b.seq_len2.assign(tok_len2_.begin(), tok_len2_.begin() + b.raw_keep);
            }
        }
        const size_t n = b.size();
        out.n_hashes.assign(n, 0);
        out.status.assign(n, 0);
        out.per_filter.assign(held_.size(), FilterResult{});
        for (auto& fr : out.per_filter)
            fr.match_off.assign(n + 1, 0);
        std::vector<uint8_t>  r1, r2;
        std::vector<uint64_t> hashes;
        std::vector<uint16_t> counts;
        for (size_t r = 0; r < n; ++r)
        {
            auto ranks = [&](const uint8_t* src, uint64_t len, std::vector<uint8_t>& dst) {
                dst.resize(len);
                for (uint64_t i = 0; i < len; ++i)
                    dst[i] = gno_char_to_rank(src[i], nullptr);
            };
            ranks(b.seq1(r), b.len1(r), r1);
            if (b.paired)
                ranks(b.seq2(r), b.len2(r), r2);
            else
                r2.clear();
            size_t nh = 0;
            if (r1.size() < w)
                out.status[r] = 1;
            else
            {
                hashes.resize(r1.size() + r2.size() + 1);
                nh = gno_minimiser_hash(r1.data(), r1.size(), k, w, hashes.data(), hashes.size());
                if (r2.size() >= w)
                    nh += gno_minimiser_hash(r2.data(), r2.size(), k, w, hashes.data() + nh, hashes.size() - nh);
                if (nh > 65535)
                    out.status[r] = 2;
            }
            out.n_hashes[r] = (uint32_t)nh;
            for (size_t i = 0; i < held_.size(); ++i)
            {
                FilterResult& fr = out.per_filter[i];
                if (out.status[r] == 0)
                {
                    Held&          h   = held_[i];
                    const uint64_t thr = gno_threshold_cutoff(nh, rel_cutoff[i]);
                    if (!h.is_hibf)
                    {
                        counts.assign(h.ibfs[0].bins, 0);
                        gno_ibf_bulk_count(&h.ibfs[0], hashes.data(), nh, counts.data());
                        for (uint32_t t = 0; t < h.n_targets; ++t)
                        {
                            uint64_t s = 0;
                            for (uint32_t x = h.off[t]; x < h.off[t + 1]; ++x)
                                s += counts[h.bins[x]];
                            if (s > nh)
                                s = nh;
                            if (s >= thr)
                                fr.matches.push_back(Match{ (uint32_t)r, t, (uint32_t)s });
                        }
                    }
                    else
                    {
                        std::vector<const int64_t*> nx, bu;
                        for (size_t j = 0; j < h.ibfs.size(); ++j)
                        {
                            nx.push_back(h.next[j].data());
                            bu.push_back(h.b2u[j].data());
                        }
                        gno_hibf hb{ (uint32_t)h.ibfs.size(), h.ibfs.data(), nx.data(), bu.data(), h.n_user_bins };
                        counts.assign(h.n_user_bins ? h.n_user_bins : 1, 0);
                        gno_hibf_bulk_count(&hb, hashes.data(), nh, thr, counts.data());
                        for (uint32_t t = 0; t < h.n_targets; ++t)
                        {
                            uint64_t s = counts[h.bins[h.off[t]]];
                            if (s > 0)
                            {
                                if (s > nh)
                                    s = nh;
                                fr.matches.push_back(Match{ (uint32_t)r, t, (uint32_t)s });
                            }
                        }
                    }
                }
                fr.match_off[r + 1] = fr.matches.size();
            }
        }
        return true;
    }
    std::string describe() const override { return "CPU oracle (test-only checker backend)"; }

private:
    struct Held
    {
        bool                              is_hibf = false;
        std::vector<std::vector<uint64_t>> mats;
        std::vector<IbfShape>             shapes;
        std::vector<gno_ibf>              ibfs;
        std::vector<std::vector<int64_t>> next, b2u;
        uint64_t                          n_user_bins = 0;
        std::vector<uint32_t>             off, bins;
        uint32_t                          n_targets = 0;
    };
    struct LegalLetters // dna15, either case (host/seq_io.cpp LegalTable)
    {
        bool ok[256];
        LegalLetters()
        {
            std::fill(ok, ok + 256, false);
            for (const char* p = "ACGTURYSWKMBDHVNacgturyswkmbdhvn"; *p; ++p)
                ok[(unsigned char)*p] = true;
        }
    };
    std::vector<Held>     held_;
    OracleBackend*        origin_ = nullptr; // a twin classifies against its origin's filters
    std::vector<uint32_t> tok_rec_, tok_seq_, tok_len_, tok_seq2_, tok_len2_;
};
} // namespace

// one checker instance per requested "device" (the multi-worker pipeline is exercised with --device 0,0,..)
std::vector<std::unique_ptr<Backend>> make_backends(const std::vector<int>& devices, std::string&)
{
    std::vector<std::unique_ptr<Backend>> out;
    for (size_t i = 0; i < std::max<size_t>(devices.size(), 1); ++i)
        out.emplace_back(new OracleBackend());
    return out;
}
} // namespace gnhost

namespace gnhost
{
// main.cpp's --verify-filter lives in verify.cpp (device lookups through libganon_hip.so); the checker binary has no device
bool verify_filter(const Config&)
{
    std::cerr << "--verify-filter needs the HIP backend (ganon-classify), not the test checker" << std::endl;
    return false;
}
} // namespace gnhost
