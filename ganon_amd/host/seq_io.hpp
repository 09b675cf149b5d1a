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

#include <cstdint>
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
    explicit SeqReader(const std::string& path); // throws ParseError / runtime_error
    ~SeqReader();
    // next record: its id is appended to `ids` and its ASCII sequence to `bases` (the caller keeps the offsets);
    // returns false at end of file.  A ParseError leaves both containers as they were before the call.
    bool next(std::string& ids, std::vector<uint8_t>& bases);

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

} // namespace gnhost
