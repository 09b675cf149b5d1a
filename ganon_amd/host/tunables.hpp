// tunables.hpp -- every environment knob of the host binaries in ONE table, read ONCE.
//
// Up to round 5 the host read its environment where it needed it: 36 getenv() sites, 15 of them in classify.cpp.  Now main() calls
// HostTunables::init() (a library user that never does gets the same on first use), every site asks this object, and `--verbose`
// lists what was set -- so that a run's stderr says under which knobs it ran.  None of these has a counterpart in the reference
// (its knobs are the command line, CommandLineParser.cpp:14-46, which cli.cpp mirrors); they are measurement and test hooks:
// defaults are the product.  The LIBRARY's switches are a separate list ($GANON_HIP_ABLATE, parsed once in libganon_hip.so).
#pragma once

#include <algorithm>
#include <array>
#include <cstdlib>
#include <mutex>
#include <optional>
#include <ostream>
#include <string>

namespace gnhost
{

enum class Knob
{
    timing,
    batch_reads,
    lanes,
    post_threads,
    mate_threads,
    parse_threads,
    slab_bytes,
    parallel_min,
    pair_text,
    no_prefilter,
    device_fastq,
    device_inflate,
    device_inflate_min,
    device_inflate_room,
    device_inflate_step,
    device_inflate_chunk,
    device_inflate_turns,
    no_shared_hashes,
    no_warm_up,
    pageable,
    prelock_mib,
    no_bgzf,
    no_pgzip,
    inflate_threads,
    inflate_chunk,
    partition_workers,
    device_budget,
    device,
    hip_ablate,
    full_teardown,
    COUNT
};

struct KnobInfo
{
    const char* env;
    const char* what;
};

inline const KnobInfo& knob_info(Knob k)
{
    static const KnobInfo table[(size_t)Knob::COUNT] = {
        { "GANON_HOST_TIMING", "print [host ...] timing, stall and CPU lines on stderr" },
        { "GANON_HOST_BATCH_READS", "reads per device batch (default 1048576; tests of the multi-worker pipeline use small ones)" },
        { "GANON_HOST_LANES", "sets of device streams a worker thread drives in turn (default 2)" },
        { "GANON_HOST_POST_THREADS", "threads of the post pool (default: 3, or two per distinct device within half of the usable cores)" },
        { "GANON_HOST_MATE_THREADS", "threads appending the second mates of a batch of pairs (default 3)" },
        { "GANON_HOST_PARSE_THREADS", "slab parser threads (default: half of the usable cores, 4..12)" },
        { "GANON_HOST_SLAB_BYTES", "bytes of input text per slab / device piece (default 48 MiB)" },
        { "GANON_HOST_PARALLEL_MIN", "files smaller than this are read sequentially (default 32 MiB)" },
        { "GANON_HOST_PAIR_TEXT", "1 / 0: mate files always / never travel as text pieces (default: files of 4 GiB and more)" },
        { "GANON_HOST_NO_PREFILTER", "no filter_matches pre-pass on the device: the host judges every match" },
        { "GANON_HOST_DEVICE_FASTQ", "0: the host parses FASTQ records (default: the device finds them in the raw text)" },
        { "GANON_HOST_DEVICE_INFLATE", "0: gzip input is inflated by the host's threads (default: on the device)" },
        { "GANON_HOST_DEVICE_INFLATE_MIN", "gzip files smaller than this are inflated by the host (default 1 MiB)" },
        { "GANON_HOST_DEVICE_INFLATE_ROOM", "device memory that must be free beside filters and batch buffers for the device inflater (default 40 GiB)" },
        { "GANON_HOST_DEVICE_INFLATE_STEP", "compressed bytes per device inflate step (default 128 / 256 MiB by file size)" },
        { "GANON_HOST_DEVICE_INFLATE_CHUNK", "bytes of compressed data per device chunk (default: the library's 32 KiB)" },
        { "GANON_HOST_DEVICE_INFLATE_TURNS", "inflaters that take a .gz file's steps in turn (default: one per distinct device; more than the devices: several on one -- tests)" },
        { "GANON_HOST_NO_SHARED_HASHES", "every filter's stream hashes the batch itself" },
        { "GANON_HOST_NO_WARM_UP", "no warm-up batch through the worker contexts" },
        { "GANON_HOST_PAGEABLE", "device-bound host buffers from the heap, not from the page-locked pool" },
        { "GANON_HOST_PRELOCK_MIB", "MiB of page-locked blocks prepared while the filters load (default 512; 0: none)" },
        { "GANON_HOST_NO_BGZF", "blocked gzip is read like any gzip (one member after the other)" },
        { "GANON_HOST_NO_PGZIP", "gzip input the device does not take is inflated by zlib on one thread" },
        { "GANON_HOST_INFLATE_THREADS", "threads of the host's parallel inflater (default: twice the parser threads)" },
        { "GANON_HOST_INFLATE_CHUNK", "compressed bytes per chunk of the host's parallel inflater" },
        { "GANON_PARTITION_WORKERS", "workers that drive a level with a partitioned filter (default 2)" },
        { "GANON_DEVICE_BUDGET", "bytes every --device entry may hold (tests: forces a partition on one GPU; e.g. 6G)" },
        { "GANON_DEVICE", "default for --device" },
        { "GANON_HIP_ABLATE", "the library's switch list (include/ganon_hip.h); the host adds sync=block when it names no sync mode" },
        { "GANON_HOST_FULL_TEARDOWN", "return from main normally (destructors, atexit handlers) instead of _Exit after the outputs are closed" },
    };
    return table[(size_t)k];
}

class HostTunables
{
public:
    static HostTunables& mut()
    {
        static HostTunables t;
        return t;
    }
    // reads the environment; main() calls it first thing.  Idempotent: the first call wins (get() makes it when nobody did).
    static void init()
    {
        HostTunables& t = mut();
        std::call_once(t.once_, [&t] {
            for (size_t i = 0; i < (size_t)Knob::COUNT; ++i)
                if (const char* v = std::getenv(knob_info((Knob)i).env))
                    t.raw_[i] = std::string(v);
        });
    }
    static const HostTunables& get()
    {
        init();
        return mut();
    }

    bool is_set(Knob k) const { return raw_[(size_t)k].has_value(); }
    // "set and not switched off": NAME=0 is off, as for the knobs that default to on
    bool off(Knob k) const { return is_set(k) && !raw_[(size_t)k]->empty() && (*raw_[(size_t)k])[0] == '0'; }
    const std::string* str(Knob k) const { return is_set(k) ? &*raw_[(size_t)k] : nullptr; }
    size_t size(Knob k, size_t dflt) const
    {
        return is_set(k) ? (size_t)std::max(0LL, std::atoll(raw_[(size_t)k]->c_str())) : dflt;
    }

    // `--verbose`: the knobs this run was started with (one line; "none" when the environment names none)
    void list_set(std::ostream& os) const
    {
        os << "[host tunables] ";
        bool any = false;
        for (size_t i = 0; i < (size_t)Knob::COUNT; ++i)
            if (raw_[i])
            {
                os << (any ? " " : "") << knob_info((Knob)i).env << '=' << *raw_[i];
                any = true;
            }
        os << (any ? "" : "none set (defaults)") << '\n';
    }
    // `--list-tunables`-style table: every knob, its value and what it does
    void list_all(std::ostream& os) const
    {
        for (size_t i = 0; i < (size_t)Knob::COUNT; ++i)
            os << "  " << knob_info((Knob)i).env << (raw_[i] ? "=" + *raw_[i] : std::string(" (unset)")) << "  -- " << knob_info((Knob)i).what << '\n';
    }

private:
    HostTunables() = default;
    std::array<std::optional<std::string>, (size_t)Knob::COUNT> raw_;
    std::once_flag                                               once_;
};

inline const HostTunables& tun()
{
    return HostTunables::get();
}

} // namespace gnhost
