// seq_io.hpp -- FASTA/FASTQ (plain or gzip) reader with the input semantics of the reference's
// seqan3::sequence_file_input<raptor::dna4_traits, fields<id, seq>> (GanonClassify.cpp:1220-1287;
// /root/reference/src/utils/include/utils/dna4_traits.hpp:15-18):
//   * format by extension (.fa .fasta .fna .ffn .faa .frn .fas / .fq .fastq, optional .gz/.bgzf/.bz2 suffix)
//   * id = the full header line after '>' / '@' (SeqAn3 default truncate_ids = false)
//   * sequence letters must be legal dna15 (ACGTU + IUPAC, any case); anything else is a parse error, which
//     makes the caller skip the rest of the file exactly like the reference's catch (:1278-1283)
//   * the bases are kept as ASCII: the conversion to dna4 ranks (IUPAC collapse) happens on the device
//   * FASTA: whitespace and digits inside sequences are ignored; FASTQ: qualities are read and dropped
#pragma once

#include "hostmem.hpp"

#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace gnhost
{

struct ParseError : std::runtime_error
{
    using std::runtime_error::runtime_error;
};

class SeqReader
{
public:
    // start_offset: begin at this byte of an UNCOMPRESSED file (must be the first byte of a record)
    explicit SeqReader(const std::string& path, uint64_t start_offset = 0); // throws ParseError / runtime_error
    ~SeqReader();
    // next record: its id is appended to `ids` and its ASCII sequence to `bases` (the caller keeps the offsets);
    // returns false at end of file.  A ParseError leaves both containers as they were before the call.
    bool next(std::string& ids, ByteBuf& bases);

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

// Parallel parser for uncompressed FASTQ files whose records have the common four-line form (@id / bases / + / qualities):
// the file is mapped, cut into byte slabs, and a few threads parse slabs ahead of the consumer, which receives them in
// file order.  A slab starts at the first record boundary at or after its first byte; a line starts a record iff it begins
// with '@' and the line after the next begins with '+' (a quality line may begin with '@', but then the line after the
// next is a base line, and '+' is no legal base).  The records of a slab are validated strictly; anything the four-line
// form does not cover (wrapped sequences, blank lines) ends the slab as `irregular` at that record, and the caller
// continues there with the sequential SeqReader -- so the result is the sequential parser's, record for record.
class ParallelFastq
{
public:
    struct Slab
    {
        std::string           ids;         // ids back to back
        std::vector<uint64_t> id_off{ 0 }; // n+1
        ByteBuf               bases;       // ASCII, back to back (device-bound: page-locked under the HIP backend)
        std::vector<uint64_t> off{ 0 };    // n+1
        std::vector<uint64_t> rec_at;      // n+1: byte offset of every record in the file (last = end of the parsed part)
        // raw mode (open(..., raw = true)): the bytes of the slab's records as they lie in the file, nothing parsed -- whoever
        // takes the slab finds the records itself (the HIP backend does, on the device: csrc/gn_fastq.hip)
        ByteBuf               text;
        uint64_t              text_at = 0; // offset of text[0] in the file (the first byte of a record, by the slab rule below)
        uint64_t              text_lines = 0; // raw mode: number of '\n' in `text`
        std::string           error;       // a ParseError ended the file after the records above
        bool                  irregular = false;
        uint64_t              resume_at = 0; // irregular: byte offset of the first record that was not parsed here
        bool                  resume_after_previous = false; // (inside the reader: resume_at = where the slab before this one ended)
        size_t                size() const { return off.size() - 1; }
    };
    // nullptr when the file is not eligible (compressed, not FASTQ by extension, smaller than min_bytes, cannot be mapped)
    // mate_room: the slabs' base buffers are reserved with room for as many bases again (the first file of a pair: the
    // mates are appended behind them when the slab becomes a batch -- without the room that append re-locks pages)
    // raw: uncompressed FASTQ or FASTA only -- the slabs are cut by the same rule but delivered as text (Slab::text), unparsed
    static std::unique_ptr<ParallelFastq> open(const std::string& path, unsigned threads, size_t slab_bytes, size_t min_bytes,
                                               bool mate_room = false, bool raw = false);
    ~ParallelFastq();
    bool fasta() const;   // the file is FASTA by its name (raw slabs: two-line records, >id / letters)
    bool next(Slab& out); // slabs in file order; false at the end of the file (or after an error / irregular slab)
    void recycle(Slab&& used); // hands a consumed slab's buffers back to the parser threads

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
    explicit ParallelFastq(Impl* i);
};

// number of '\n' in [p, p + n)
uint64_t count_newlines(const uint8_t* p, size_t n);


// Where the lines of a plain (uncompressed) text file begin: threads count the newlines of fixed chunks ahead of the caller, and
// line_begin(L) answers with the byte offset of line L's first byte (line 0 begins at byte 0) as soon as the chunks up to it are
// counted.  The mate file of a pair whose pieces travel as text is cut with it: file 1's piece holds n records, the mates are the
// next 4 n lines of file 2 (GanonClassify.cpp:1240-1252: file 2 is consumed with take(n_reads)).
class LineIndex
{
public:
    // nullptr: not a plain FASTQ/FASTA file by name and magic bytes, or smaller than min_bytes
    static std::unique_ptr<LineIndex> open(const std::string& path, unsigned threads, size_t min_bytes);
    ~LineIndex();
    static constexpr uint64_t kNoSuchLine = ~0ull;
    // byte offset of line L's first byte; the file's size when L is exactly the number of lines it has and its last byte is a
    // newline (the line that would begin behind the end); kNoSuchLine when the file has fewer lines
    uint64_t line_begin(uint64_t line);
    uint64_t size() const;
    // bytes [begin, end) of the file into dst (page-locked under the HIP backend)
    bool read(uint64_t begin, uint64_t end, ByteBuf& dst, size_t reserve) const;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
    explicit LineIndex(Impl* i);
};

} // namespace gnhost
