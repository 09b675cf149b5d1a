// cpu_tally.hpp -- CPU seconds per group of threads (parsers, device workers, post pool, ...), for $GANON_HOST_TIMING.
// On a host with fewer cores than threads the sum of these, divided by the cores, is the floor of the run's wall time:
// which stage to make cheaper is read off this table, not off the stages' wall times.
#pragma once

#include <sys/resource.h>

#include <atomic>
#include <cstdint>
#include <ostream>

namespace gnhost
{

struct CpuTally
{
    std::atomic<uint64_t> user_us{ 0 }, sys_us{ 0 };
    // adds what the calling thread has used so far (call once, when the thread is about to end)
    void add_this_thread()
    {
        rusage r;
        if (getrusage(RUSAGE_THREAD, &r) == 0)
        {
            user_us += (uint64_t)r.ru_utime.tv_sec * 1000000u + (uint64_t)r.ru_utime.tv_usec;
            sys_us += (uint64_t)r.ru_stime.tv_sec * 1000000u + (uint64_t)r.ru_stime.tv_usec;
        }
    }
    // ... or what it used between two points (a thread that does other things too: the main thread's merge + write loop)
    static void thread_now(uint64_t& user, uint64_t& sys)
    {
        rusage r;
        user = sys = 0;
        if (getrusage(RUSAGE_THREAD, &r) == 0)
        {
            user = (uint64_t)r.ru_utime.tv_sec * 1000000u + (uint64_t)r.ru_utime.tv_usec;
            sys  = (uint64_t)r.ru_stime.tv_sec * 1000000u + (uint64_t)r.ru_stime.tv_usec;
        }
    }
    void add_since(uint64_t user0, uint64_t sys0)
    {
        uint64_t u, s;
        thread_now(u, s);
        user_us += u - user0;
        sys_us += s - sys0;
    }
    void print(std::ostream& os, const char* name) const
    {
        os << name << ' ' << user_us.load() * 1e-6 << " + " << sys_us.load() * 1e-6;
    }
};

// first and last time something happened, in seconds since a common start (the pipeline's ramp-up and tail, $GANON_HOST_TIMING)
struct EventSpan
{
    std::atomic<int64_t> first_us{ -1 }, last_us{ -1 };
    std::atomic<uint64_t> count{ 0 };
    void mark(int64_t us)
    {
        int64_t expect = -1;
        first_us.compare_exchange_strong(expect, us);
        int64_t prev = last_us.load();
        while (prev < us && !last_us.compare_exchange_weak(prev, us)) {}
        ++count;
    }
    void reset()
    {
        first_us = -1;
        last_us  = -1;
        count    = 0;
    }
    void print(std::ostream& os, const char* name) const
    {
        os << name << ' ' << count.load() << "x " << first_us.load() * 1e-6 << " .. " << last_us.load() * 1e-6;
    }
};

struct CpuTallies
{
    CpuTally parse, inflate, reader, mate, worker, post, merge, phase;
};
inline CpuTallies g_cpu;

} // namespace gnhost
