// post.hpp -- the post stage of ganon-classify: what one classified batch becomes (split from classify.cpp in round 6; pipeline.hpp
// holds what travels between the stages).  The reference does this per read inside its classify threads
// (/root/reference/src/ganon-classify/GanonClassify.cpp:753-830: filter_matches :579-613, lca_matches :615-627, the writers' lines
// :1289-1322, the carry to the next hierarchy level :811-820); here a pool of threads does it per batch, and the ordered merge
// (classify.cpp) writes the batches' texts in input order.
#pragma once

#include "config.hpp"
#include "filter_io.hpp"
#include "lca.hpp"
#include "pipeline.hpp"
#include "plan.hpp"
#include "robin_order.hpp"

#include <atomic>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace gnhost
{

struct MatchEntry
{
    uint32_t gid;
    size_t   count;
    double   fpr;
    bool     fpr_ok; // the backend already verified the --fpr-query rule for this match
    uint64_t ins;    // --reference-order: (filter, rank in the filter's TMap order) of the insertion into the read's map
};
// the post stage runs on a small pool of threads, one batch each; this is what a thread keeps between batches
struct PostScratch
{
    std::vector<MatchEntry>  matches;
    // several filters of a level can report the same target: its slot in `matches`, valid while the stamp is the read's
    std::vector<uint32_t>    slot_of, stamp_of;
    uint32_t                 stamp = 0;
    std::vector<uint32_t>    kept_gids;
    std::vector<std::string> kept_targets;
    // --reference-order
    RobinSlots               slots;
    std::vector<uint32_t>    slot_order, pos_of, touch_stamp;
    std::vector<MatchEntry>  reordered;
    uint32_t                 touch_epoch = 0;
};

// What a level's post stage reads (all of it set up before the first batch; the threads of the pool share it) and the few
// things it adds to (atomics, or under the mutex).
struct PostContext
{
    const Config&                                    config;
    const Level&                                     level;
    const std::vector<FilterMeta>&                   filters;
    const std::vector<std::vector<uint32_t>>&        target_gid; // per filter: target index -> node id of the level
    const std::vector<std::string>&                  node_names;
    const std::unordered_map<std::string, uint32_t>& node_ids;
    const LCA&                                       lca;
    const std::vector<uint64_t>&                     name_hash;  // --reference-order (robin_order.hpp)
    const std::vector<std::vector<uint32_t>>&        map_rank;
    bool                                             first_level, last_level, shared_targets;
    std::atomic<uint64_t>&                           diag_unmerged;
    std::atomic<uint64_t>&                           diag_fpr_evals;
    std::mutex&                                      timing_mutex;
    double&                                          sec_post;
};

// one batch's reads -> text, tallies, carried reads.  Touches nothing shared but the context's read-only tables, so several
// threads run it at once.
void post_stage(const PostContext& cx, ReadBatch& rb, const BatchResult& res, PostScratch& sc, PostOutput& po);

// `readID \t target \t count \n` (write_classified, :1289-1306)
void append_line(std::string& dst, std::string_view id, std::string_view target, size_t count);

} // namespace gnhost
