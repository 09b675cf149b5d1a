// gn_scan.h -- exclusive scan of 32-bit counts into 64-bit offsets.
// hipcub::DeviceScan::ExclusiveSum adds in the INPUT's type: with uint32 counts the offsets wrap once a batch holds more
// than 2^32 matches (seen with 6.2 G matches of 2 M reads against 16 384 split-bin targets at --rel-cutoff 0.2: read segments
// aliased, totals were off by 0.04 %).  With a 64-bit initial value the accumulator is 64 bits wide.
#pragma once
#include <hipcub/hipcub.hpp>

static inline hipError_t gn_scan_counts(void* tmp, size_t& tmp_bytes, const uint32_t* counts, uint64_t* offsets, int n, hipStream_t st)
{
    return hipcub::DeviceScan::ExclusiveScan(tmp, tmp_bytes, counts, offsets, hipcub::Sum(), (uint64_t)0, n, st);
}

// ... the same with offsets that start at `first` (the mates' letters lie behind the first file's in the stream's buffer)
static inline hipError_t gn_scan_counts_from(void* tmp, size_t& tmp_bytes, const uint32_t* counts, uint64_t* offsets, uint64_t first, int n,
                                             hipStream_t st)
{
    return hipcub::DeviceScan::ExclusiveScan(tmp, tmp_bytes, counts, offsets, hipcub::Sum(), first, n, st);
}
