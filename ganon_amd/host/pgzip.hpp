// pgzip.hpp -- parallel reader for ordinary (non-blocked) gzip files: the input form most read sets come in
// (`reads.fq.gz`; the reference reads it through one seqan3/zlib stream, GanonClassify.cpp:1220-1287,1433).
//
// A DEFLATE stream has no index, and every block may refer to the 32 KiB of output before it, so a second thread cannot
// simply start in the middle.  Two passes make it parallel all the same (the idea of pugz / rapidgzip):
//   1. the compressed file is cut into chunks; for every chunk but the first a thread SEARCHES the first position (bit
//      granular) at which a dynamic-Huffman block header parses, its code is complete and a whole block decodes to plausible
//      text, and decodes from there to the next chunk's start with a decoder that emits 16-bit symbols: a byte, or a
//      MARKER "byte i of the 32 KiB window before this chunk" wherever a match reaches back over the chunk's start;
//   2. in file order the windows are settled (the last 32 KiB of chunk k-1 resolve the markers in the last 32 KiB of chunk
//      k -- a 32 K-symbol step per chunk, the only sequential part), then every chunk's markers are replaced in parallel,
//      the members' CRC-32 and length are checked, and the bytes are published.
// A chunk whose predecessor does not end exactly where it began (a false start) is decoded again from the true position.
// Consumers read the decompressed stream by offset (pread) inside a sliding window: the slab parsers of seq_io.cpp run on
// it unchanged, as on an uncompressed file.
//
// Errors (damaged or truncated data, wrong CRC) are not reported from here with zlib's exact timing; the caller treats
// them as "not parseable in parallel from this point" and lets the sequential zlib reader produce the records and the
// message from there, so that the outcome is the sequential reader's byte for byte (seq_io.cpp, ParallelFastq).
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>

namespace gnhost
{

class ParallelGzip
{
public:
    // nullptr when the file is no gzip file, is a BGZF file smaller than min_bytes ... or cannot be mapped.
    // chunk_bytes = compressed bytes per chunk (0 = default 2 MiB; tests use small ones)
    static std::unique_ptr<ParallelGzip> open(const std::string& path, unsigned threads, size_t min_bytes, size_t chunk_bytes = 0);
    ~ParallelGzip();

    // Up to n decompressed bytes at offset off (>= the last release_below()): waits until they exist; fewer than n only at
    // the end of the stream, 0 beyond it.  Throws std::runtime_error when the stream is damaged at or before that point.
    size_t pread(char* dst, size_t n, uint64_t off);
    // the caller is done with everything below `off`
    void release_below(uint64_t off);
    // how much decompressed data may wait for release_below (the producer pauses beyond it); at least two of the caller's read units
    void set_retain_limit(uint64_t bytes);
    // decompressed size once the end has been reached by pread (UINT64_MAX before)
    uint64_t known_size() const;

    struct Stats
    {
        uint64_t chunks = 0, redone = 0, members = 0, markers = 0;
    };
    Stats stats() const;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
    explicit ParallelGzip(Impl* i);
};

} // namespace gnhost
