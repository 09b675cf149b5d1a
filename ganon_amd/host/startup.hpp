// startup.hpp -- where a run's time goes before the first read is classified (`ganon-classify --verbose`: the [startup] lines).
// A 16 M-read file classifies in 0.13 s; the process needed 1.6 s (round 4): runtime initialisation, code objects, page-locking,
// the filter's way into HBM and the warm-up batch are the rest, so each is timed where it happens.  Marks are cheap (a clock
// read and a push under a mutex) and always taken; printing is --verbose only.  The reference's counterpart is its
// "loading filter(s) elapsed" line (timeLoadFilters, GanonClassify.cpp:1470-1477).
#pragma once
#include "tunables.hpp"
#include <cstdlib>
#include <cstring>

#include <chrono>
#include <cstdio>
#include <fstream>
#include <mutex>
#include <ostream>
#include <string>
#include <unistd.h>
#include <vector>

namespace gnhost
{

class StartupLog
{
public:
    static StartupLog& get()
    {
        static StartupLog s;
        return s;
    }
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    // a phase that ran from t0 to now (phases may overlap: some run in threads of their own)
    void span(const std::string& what, double t0, const std::string& note = std::string())
    {
        std::lock_guard<std::mutex> lk(m_);
        spans_.push_back({ what, t0 - origin_, now() - origin_, note });
    }
    // seconds between exec() and main(): dynamic linking (libamdhip64 and friends), static constructors
    double before_main() const { return before_main_; }
    void   print(std::ostream& os)
    {
        std::lock_guard<std::mutex> lk(m_);
        char                        buf[512];
        std::snprintf(buf, sizeof(buf), "[startup] exec -> main %.3f s (dynamic linking, static constructors); times below count from main\n", before_main_);
        os << buf;
        for (auto const& s : spans_)
        {
            std::snprintf(buf, sizeof(buf), "[startup] %7.3f .. %7.3f s  %6.3f s  %s%s%s\n", s.a, s.b, s.b - s.a, s.what.c_str(), s.note.empty() ? "" : "  -- ",
                          s.note.c_str());
            os << buf;
        }
        spans_.clear();
    }

private:
    StartupLog() : origin_(now())
    {
        // process start from /proc/self/stat (field 22, clock ticks since boot) against /proc/uptime
        std::ifstream st("/proc/self/stat"), up("/proc/uptime");
        std::string   line;
        double        uptime = 0;
        if (std::getline(st, line) && (up >> uptime))
        {
            const size_t rp = line.rfind(')');
            size_t       at = rp == std::string::npos ? 0 : rp + 2;
            for (int f = 3; f < 22 && at < line.size(); ++f)
                at = line.find(' ', at) + 1;
            const double start = std::atof(line.c_str() + at) / (double)sysconf(_SC_CLK_TCK);
            before_main_       = uptime > start ? uptime - start : 0;
        }
    }
    struct Span
    {
        std::string what;
        double      a, b;
        std::string note;
    };
    std::mutex        m_;
    std::vector<Span> spans_;
    double            origin_, before_main_ = 0;
};

// main() leaves with _Exit once every output is closed (see main.cpp) unless a profiler's exit handler or $GANON_HOST_FULL_TEARDOWN wants
// the normal return: whoever would only free memory on the way out asks here and does not bother
// Anything that does its work in an exit handler or a destructor keeps the normal return: a preloaded tool of ANY kind ($LD_PRELOAD:
// rocprofv3, heap checkers, ...), a profiler registered with the ROCm runtime, coverage runs ($GCOV_PREFIX, $LLVM_PROFILE_FILE), sanitizer
// builds (their leak reports run at exit) and sanitizer option variables.  (These are other tools' variables, read here and only here;
// the host's own knobs are in tunables.hpp.)
inline bool fast_exit()
{
#if defined(__SANITIZE_ADDRESS__) || defined(__SANITIZE_THREAD__)
    return false;
#else
#if defined(__has_feature)
#if __has_feature(address_sanitizer) || __has_feature(thread_sanitizer) || __has_feature(memory_sanitizer)
    return false;
#endif
#endif
    const char* preload = std::getenv("LD_PRELOAD");
    const bool  tooled  = (preload && *preload) || std::getenv("ROCP_TOOL_LIBRARIES") || std::getenv("ROCPROFILER_REGISTER_LIBRARY") ||
                        std::getenv("GCOV_PREFIX") || std::getenv("LLVM_PROFILE_FILE") || std::getenv("ASAN_OPTIONS") || std::getenv("LSAN_OPTIONS") ||
                        std::getenv("TSAN_OPTIONS");
    return !tun().is_set(Knob::full_teardown) && !tooled;
#endif
}

} // namespace gnhost
