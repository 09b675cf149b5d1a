// seq_io.cpp -- see seq_io.hpp
#include "seq_io.hpp"

#include <zlib.h>

#include <algorithm>
#include <cstring>

namespace gnhost
{

namespace
{
bool ends_with(const std::string& s, const char* suf)
{
    const size_t n = std::strlen(suf);
    return s.size() >= n && std::equal(s.end() - n, s.end(), suf, [](char a, char b) { return std::tolower(a) == b; });
}

// dna15 legal letters
struct LegalTable
{
    bool ok[256];
    LegalTable()
    {
        std::fill(ok, ok + 256, false);
        for (const char* p = "ACGTURYSWKMBDHVN"; *p; ++p)
        {
            ok[(unsigned char)*p]               = true;
            ok[(unsigned char)std::tolower(*p)] = true;
        }
    }
};
const LegalTable kLegal;
} // namespace

struct SeqReader::Impl
{
    gzFile      gz = nullptr;
    std::string path;
    bool        fastq = false;
    std::vector<char> buf;
    size_t      pos = 0, len = 0;
    bool        eof = false;
    std::string line;
    bool        have_pending = false; // FASTA: header of the next record already read into `line`

    bool fill()
    {
        if (eof)
            return false;
        const int n = gzread(gz, buf.data(), (unsigned)buf.size());
        if (n < 0)
        {
            int         err = 0;
            const char* msg = gzerror(gz, &err);
            throw ParseError(" " + std::string(msg ? msg : "decompression error"));
        }
        pos = 0;
        len = (size_t)n;
        if (n == 0)
            eof = true;
        return n > 0;
    }
    // reads one line (without '\n', trailing '\r' stripped); false at EOF with nothing read
    bool getline(std::string& out)
    {
        out.clear();
        bool any = false;
        while (true)
        {
            if (pos == len && !fill())
                break;
            any              = true;
            const char* b    = buf.data() + pos;
            const char* nl   = static_cast<const char*>(std::memchr(b, '\n', len - pos));
            if (nl)
            {
                out.append(b, nl - b);
                pos += (size_t)(nl - b) + 1;
                break;
            }
            out.append(b, len - pos);
            pos = len;
        }
        if (!out.empty() && out.back() == '\r')
            out.pop_back();
        return any;
    }
};

SeqReader::SeqReader(const std::string& path) : impl_(new Impl)
{
    impl_->path = path;
    std::string base = path;
    for (const char* z : { ".gz", ".bgzf", ".bgz" })
        if (ends_with(base, z))
        {
            base = base.substr(0, base.size() - std::strlen(z));
            break;
        }
    if (ends_with(base, ".bz2"))
        throw ParseError(" bzip2 input is not supported by this build (no bzlib in the image)");
    bool known = false;
    for (const char* e : { ".fq", ".fastq" })
        if (ends_with(base, e))
        {
            impl_->fastq = true;
            known        = true;
        }
    for (const char* e : { ".fa", ".fasta", ".fna", ".ffn", ".faa", ".frn", ".fas" })
        if (ends_with(base, e))
            known = true;
    if (!known)
        throw ParseError(" unknown sequence file extension (expected FASTA or FASTQ, optionally gzipped)");
    impl_->gz = gzopen(path.c_str(), "rb"); // transparently reads uncompressed files too
    if (!impl_->gz)
        throw ParseError(" cannot open file");
    gzbuffer(impl_->gz, 1 << 20);
    impl_->buf.resize(1 << 20);
}

SeqReader::~SeqReader()
{
    if (impl_ && impl_->gz)
        gzclose(impl_->gz);
}

bool SeqReader::next(std::string& id, std::string& seq)
{
    Impl& s = *impl_;
    id.clear();
    seq.clear();
    if (s.fastq)
    {
        // @id / sequence line(s) / +[id] / quality line(s)
        do
        {
            if (!s.getline(s.line))
                return false;
        } while (s.line.empty());
        if (s.line[0] != '@')
            throw ParseError(" FASTQ record does not start with '@'");
        id = s.line.substr(1);
        while (true)
        {
            if (!s.getline(s.line))
                throw ParseError(" unexpected end of FASTQ record");
            if (!s.line.empty() && s.line[0] == '+')
                break;
            // validate the whole line, then append it in one go
            bool bad = false;
            for (unsigned char c : s.line)
                bad |= !kLegal.ok[c];
            if (bad)
                for (char c : s.line)
                    if (!kLegal.ok[(unsigned char)c])
                        throw ParseError(std::string(" Encountered an unexpected letter: char_is_valid_for<dna15> evaluated to false on '")
                                         + c + "'");
            seq.append(s.line);
        }
        size_t q = 0;
        while (q < seq.size())
        {
            if (!s.getline(s.line))
                throw ParseError(" unexpected end of FASTQ qualities");
            q += s.line.size();
        }
        if (q != seq.size())
            throw ParseError(" sequence and quality lengths differ");
        return true;
    }
    // FASTA
    if (!s.have_pending)
    {
        do
        {
            if (!s.getline(s.line))
                return false;
        } while (s.line.empty());
    }
    s.have_pending = false;
    if (s.line[0] != '>' && s.line[0] != ';')
        throw ParseError(" FASTA record does not start with '>'");
    id = s.line.substr(1);
    while (s.getline(s.line))
    {
        if (!s.line.empty() && (s.line[0] == '>' || s.line[0] == ';'))
        {
            s.have_pending = true;
            break;
        }
        for (char c : s.line)
        {
            if (std::isspace((unsigned char)c) || std::isdigit((unsigned char)c))
                continue;
            if (!kLegal.ok[(unsigned char)c])
                throw ParseError(std::string(" Encountered an unexpected letter: char_is_valid_for<dna15> evaluated to false on '") + c
                                 + "'");
            seq.push_back(c);
        }
    }
    return true;
}

} // namespace gnhost
