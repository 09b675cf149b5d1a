// pipeline.hpp -- what the reader, the device workers and the post stage of ganon-classify pass between each other (classify.cpp was one
// file of 2 300 lines up to round 5; the seams are the reference's own: one parser thread, GanonClassify.cpp:1220-1287,1436-1441, in front
// of the classify threads, :1579-1597).   reader.cpp: files -> numbered batches;   classify.cpp: workers, post stage, merge + write.
#pragma once

#include "backend.hpp"
#include "plan.hpp"
#include "report.hpp"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace gnhost
{

// ---- queues --------------------------------------------------------------------------------------------------
// bounded producer/consumer queue; consumed items can be handed back so that their (already faulted-in) buffers
// are reused by the producer
template <typename T>
class BoundedQueue
{
public:
    explicit BoundedQueue(size_t cap) : cap_(cap) {}
    void push(T&& b)
    {
        std::unique_lock<std::mutex> lk(m_);
        const auto t0 = std::chrono::steady_clock::now();
        not_full_.wait(lk, [&] { return q_.size() < cap_; });
        blocked_push_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        q_.push_back(std::move(b));
        not_empty_.notify_one();
    }
    // seconds producers spent waiting for room / consumers waiting for an item ($GANON_HOST_TIMING)
    double blocked_push() const { return blocked_push_; }
    double blocked_pop() const { return blocked_pop_; }
    void done()
    {
        std::lock_guard<std::mutex> lk(m_);
        done_ = true;
        not_empty_.notify_all();
    }
    bool pop(T& b)
    {
        std::unique_lock<std::mutex> lk(m_);
        const auto t0 = std::chrono::steady_clock::now();
        not_empty_.wait(lk, [&] { return !q_.empty() || done_; });
        blocked_pop_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (q_.empty())
            return false;
        b = std::move(q_.front());
        q_.pop_front();
        not_full_.notify_one();
        return true;
    }
    size_t size()
    {
        std::lock_guard<std::mutex> lk(m_);
        return q_.size();
    }
    size_t capacity() const { return cap_; }
    // 1 = got one, 0 = nothing there right now, -1 = the producer is done and nothing is left
    int try_pop(T& b)
    {
        std::lock_guard<std::mutex> lk(m_);
        if (q_.empty())
            return done_ ? -1 : 0;
        b = std::move(q_.front());
        q_.pop_front();
        not_full_.notify_one();
        return 1;
    }
    void recycle(T&& b)
    {
        std::lock_guard<std::mutex> lk(m_);
        if (free_.size() < cap_ + 2)
            free_.push_back(std::move(b));
    }
    bool take_free(T& b)
    {
        std::lock_guard<std::mutex> lk(m_);
        if (free_.empty())
            return false;
        b = std::move(free_.back());
        free_.pop_back();
        return true;
    }

private:
    std::mutex              m_;
    std::condition_variable not_full_, not_empty_;
    std::deque<T>           q_;
    std::vector<T>          free_;
    size_t                  cap_;
    bool                    done_ = false;
    double                  blocked_push_ = 0, blocked_pop_ = 0;
};
using BatchQueue = BoundedQueue<ReadBatch>;

// a batch together with what the device said about it
// what the post stage makes of one batch; merged into the level's tallies and files in input order
struct PostOutput
{
    std::string              all, lca, unc; // text for the .all / .one / .unc files
    ReadSetTally             reads;         // this batch's share of the per-prefix read tally
    std::vector<TargetTally> targets;       // ... and of the per-target tallies (dense by node id)
    ReadBatch                left;          // reads that stay unclassified on a level that is not the last
    bool                     has_left = false;
    std::vector<uint32_t>    touched;       // --reference-order: node ids in the order this batch first touched their report rows
};

struct ClassifiedBatch
{
    ReadBatch   rb;
    BatchResult res;
    PostOutput  post;
};

// Results of several device workers, handed to the post stage in input order.  A worker may not run ahead of the
// post stage by more than `window` batches (bounds the memory held by finished batches).
class InOrder
{
public:
    explicit InOrder(size_t window) : window_(window) {}
    void wait_turn(uint64_t seq) // before a worker starts on batch `seq`
    {
        std::unique_lock<std::mutex> lk(m_);
        const auto t0 = std::chrono::steady_clock::now();
        cv_.wait(lk, [&] { return seq < next_ + window_ || aborted_; });
        blocked_turn_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    bool try_turn(uint64_t seq) // the same without waiting
    {
        std::lock_guard<std::mutex> lk(m_);
        return seq < next_ + window_ || aborted_;
    }
    double blocked_turn() const { return blocked_turn_; }
    double blocked_take() const { return blocked_take_; }
    void put(uint64_t seq, ClassifiedBatch&& cb)
    {
        std::lock_guard<std::mutex> lk(m_);
        ready_.emplace(seq, std::move(cb));
        cv_.notify_all();
    }
    void producer_done() // one worker has run out of input
    {
        std::lock_guard<std::mutex> lk(m_);
        ++finished_;
        cv_.notify_all();
    }
    void abort()
    {
        std::lock_guard<std::mutex> lk(m_);
        aborted_ = true;
        cv_.notify_all();
    }
    // next batch in input order; false when every worker is done and nothing is left (or after abort())
    bool take(ClassifiedBatch& cb, size_t n_workers)
    {
        std::unique_lock<std::mutex> lk(m_);
        const auto t0 = std::chrono::steady_clock::now();
        cv_.wait(lk, [&] { return aborted_ || ready_.count(next_) || (finished_ == n_workers && ready_.empty()); });
        blocked_take_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (aborted_)
            return false;
        auto it = ready_.find(next_);
        if (it == ready_.end())
            return false;
        cb = std::move(it->second);
        ready_.erase(it);
        ++next_;
        cv_.notify_all();
        return true;
    }
    void recycle(ClassifiedBatch&& cb)
    {
        std::lock_guard<std::mutex> lk(m_);
        if (free_.size() < window_)
            free_.push_back(std::move(cb));
    }
    bool take_free(ClassifiedBatch& cb)
    {
        std::lock_guard<std::mutex> lk(m_);
        if (free_.empty())
            return false;
        cb = std::move(free_.back());
        free_.pop_back();
        return true;
    }

private:
    std::mutex                          m_;
    std::condition_variable             cv_;
    std::map<uint64_t, ClassifiedBatch> ready_;
    std::vector<ClassifiedBatch>        free_;
    uint64_t                            next_ = 0;
    size_t                              window_, finished_ = 0;
    bool                                aborted_ = false;
    double                              blocked_turn_ = 0, blocked_take_ = 0;
};

// reads / bases per device batch ($GANON_HOST_BATCH_READS: smaller batches for tests of the multi-worker pipeline)
size_t           batch_reads();
constexpr size_t kBatchBases = 1ull << 28;

// cores this process may really use: the affinity mask, capped by the cgroup's CPU quota
unsigned usable_cores();

// Ends of device text sources (threads joined, gigabytes of device buffers freed: some twenty milliseconds): they run beside the last
// batches and are waited for when ganon_classify leaves, not by the reader.
struct Cleanups
{
    std::mutex               m;
    std::vector<std::thread> th;
    void add(std::thread t);
    void join_all();
};
extern Cleanups g_cleanups;

// The device inflaters decode from the moment their file is open -- beside the filters being read and uploaded -- but begin no new step
// while a level's worker contexts are set up: every device allocation of those waits for the decode kernels then (six contexts 0.03 ->
// 0.4-0.9 s, profiles/r05_e2e_device_inflate_run8: the whole process got slower although the timed part got faster).
struct DeviceGate
{
    std::atomic<bool> run{ true };
};
extern DeviceGate g_devices_ready;

// appends the mates-2 region behind the mates-1 region and rebases its offsets (the reader's pairs; the post stage's carried reads)
void finalize_batch(ReadBatch& rb, ByteBuf& bases2);

// distinct GPUs of the run (set by ganon_classify before the reader starts): the reader sizes its helper threads by it -- one link takes
// what ~8 slab readers copy out of the page cache (5 GB/s each, scripts/host_ceiling.py), and what fed one GPU does not feed eight
extern std::atomic<unsigned> g_distinct_devices;

// the reader thread (reader.cpp)
void parse_reads(BatchQueue& queue, RunReport& report, std::mutex& report_mutex, const ReadPlan& plan, bool raw_fastq, Backend* device_text,
                 bool further_levels);

} // namespace gnhost
