// post.cpp -- see post.hpp.
#include "post.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>

namespace gnhost
{

namespace
{

inline size_t ceil_share(size_t n, double p) // ceil(n * p) in double, as threshold_rel (:492-495)
{
    return std::ceil(n * p);
}
inline double binomial_coefficient(double n, double k) noexcept // :498-501
{
    // lgamma_r: the values of std::lgamma without its write to the global `signgam` (the post stage runs on several threads)
    int sign = 0;
    return std::exp(lgamma_r(n + 1, &sign) - lgamma_r(n - k + 1, &sign) - lgamma_r(k + 1, &sign));
}

void append_number(std::string& dst, size_t v)
{
    char  tmp[24];
    char* e = tmp + sizeof(tmp);
    char* q = e;
    do
    {
        *--q = char('0' + v % 10);
        v /= 10;
    } while (v);
    dst.append(q, e - q);
}

TargetTally& tally_at(std::vector<TargetTally>& v, uint32_t gid)
{
    if (gid >= v.size())
        v.resize((size_t)gid + 1);
    return v[gid];
}

} // namespace

void append_line(std::string& dst, std::string_view id, std::string_view target, size_t count)
{
    dst.append(id);
    dst += '\t';
    dst.append(target);
    dst += '\t';
    append_number(dst, count);
    dst += '\n';
}

void post_stage(const PostContext& cx, ReadBatch& rb, const BatchResult& res, PostScratch& sc, PostOutput& po)
{
    // (the names the body has always used)
    const Config& config      = cx.config;
    const Level&  level       = cx.level;
    const auto&   filters     = cx.filters;
    const auto&   target_gid  = cx.target_gid;
    const auto&   node_names  = cx.node_names;
    const auto&   name_hash   = cx.name_hash;
    const auto&   map_rank    = cx.map_rank;
    const LCA&    lca         = cx.lca;
    const bool    first_level = cx.first_level, last_level = cx.last_level, shared_targets = cx.shared_targets;
    auto          known_nid   = [&](const std::string& s) -> uint32_t { return cx.node_ids.at(s); };
    const auto    t0    = std::chrono::steady_clock::now();
    po.reads = ReadSetTally();
    po.targets.clear();
    po.has_left = false;
    ReadSetTally& total = po.reads;
    auto&         per_target = po.targets;
    std::string & buf_all = po.all, &buf_lca = po.lca, &buf_unc = po.unc;
    auto&         matches = sc.matches;
    auto &        slot_of = sc.slot_of, &stamp_of = sc.stamp_of;
    uint32_t&     stamp = sc.stamp;
    auto&         kept_gids = sc.kept_gids;
    auto&         kept_targets = sc.kept_targets;
    buf_all.clear();
    buf_lca.clear();
    buf_unc.clear();
    ReadBatch left;
    left.paired = rb.paired;
    left.prefix = rb.prefix;
    ByteBuf left2;
    left.off1.assign(1, 0);
    if (left.paired)
        left.off2.assign(1, 0);
    const bool     o_all = config.output_all, o_lca = config.output_lca && !config.skip_lca, o_unc = config.output_unclassified; // which texts are wanted
    const bool     one_filter = filters.size() == 1;
    uint64_t       n_unmerged = 0, n_fpr_evals = 0;
    po.touched.clear();
    if (config.reference_order)
    {
        if (sc.touch_stamp.size() < node_names.size())
            sc.touch_stamp.resize(node_names.size(), 0u);
        if (++sc.touch_epoch == 0)
        {
            std::fill(sc.touch_stamp.begin(), sc.touch_stamp.end(), 0u);
            sc.touch_epoch = 1;
        }
    }
    auto touch = [&](uint32_t gid) { // rep[{prefix, target}] of the reference: the row exists from here on
        if (config.reference_order && sc.touch_stamp[gid] != sc.touch_epoch)
        {
            sc.touch_stamp[gid] = sc.touch_epoch;
            po.touched.push_back(gid);
        }
    };

    for (size_t r = 0; r < rb.size(); ++r)
    {
        const size_t read1_len = rb.len1(r), read2_len = rb.len2(r);
        const size_t n_hashes  = res.n_hashes[r];
        size_t       max_count_read = 0, min_count_read = n_hashes; // :686-688,704
        matches.clear();
        if (res.status[r] == 1) // :743-747
        {
            if (first_level)
                total.too_short++;
        }
        else if (res.status[r] == 2) // :737-741
        {
            if (first_level)
                total.too_many_minimisers++;
        }
        else
        {
            if (first_level) // :709-714
            {
                total.reads_seen++;
                total.bases_seen += read1_len + read2_len;
                total.minimisers_seen += n_hashes;
            }
            if (++stamp == 0) // (wrapped: forget every slot)
            {
                std::fill(stamp_of.begin(), stamp_of.end(), 0u);
                stamp = 1;
            }
            for (size_t i = 0; i < filters.size(); ++i) // select_matches insert rule (:531-537)
            {
                const FilterResult& fr = res.per_filter[i];
                for (uint64_t x = fr.match_off[r]; x < fr.match_off[r + 1]; ++x)
                {
                    Match          m   = fr.matches[x];
                    const uint32_t gid = target_gid[i][m.target];
                    bool           ok  = !fr.fpr_ok.empty() && fr.fpr_ok[x] != 0;
                    if (fr.flag_in_count)
                    {
                        ok = (m.count & FilterResult::kMatchFprOk) != 0;
                        m.count &= ~FilterResult::kMatchFprOk;
                    }
                    MatchEntry*    e   = nullptr;
                    if (!one_filter) // (one filter reports a target once)
                    {
                        if (gid >= stamp_of.size())
                        {
                            stamp_of.resize(node_names.size(), 0u);
                            slot_of.resize(node_names.size(), 0u);
                        }
                        if (stamp_of[gid] == stamp)
                            e = &matches[slot_of[gid]];
                    }
                    const size_t existing = e ? e->count : 0;
                    if (m.count > existing)
                    {
                        if (e)
                        {
                            e->count  = m.count;
                            e->fpr    = filters[i].target_fpr[m.target];
                            e->fpr_ok = ok;
                        }
                        else
                        {
                            if (!one_filter)
                            {
                                stamp_of[gid] = stamp;
                                slot_of[gid]  = (uint32_t)matches.size();
                            }
                            matches.push_back(MatchEntry{ gid, m.count, filters[i].target_fpr[m.target], ok,
                                                          config.reference_order ? ((uint64_t)i << 40) | map_rank[i][m.target] : 0ull });
                        }
                        if (m.count > max_count_read)
                            max_count_read = m.count;
                        if (m.count < min_count_read)
                            min_count_read = m.count;
                    }
                }
            }
        }

        // `matches` are the survivors of the --rel-filter rule and the read's maximum comes with them -- unless the
        // backend left this read alone (bit 31: more matches over the level's filters than its merge takes)
        const bool prefiltered = res.prefiltered && !(res.max_count[r] & 0x80000000u);
        if (prefiltered)
            max_count_read = res.max_count[r];
        else if (res.prefiltered)
            ++n_unmerged;
        if (config.reference_order && !matches.empty())
        {
            // the read's TMatches: keys arrive filter after filter in each filter's TMap order; what filter_matches
            // walks is the map's slot order (:583)
            if (matches.size() > 1)
            {
                std::sort(matches.begin(), matches.end(), [](const MatchEntry& a, const MatchEntry& b) { return a.ins < b.ins; });
                sc.slots.clear();
                if (sc.pos_of.size() < node_names.size())
                    sc.pos_of.resize(node_names.size());
                for (size_t x = 0; x < matches.size(); ++x)
                {
                    sc.slots.insert(matches[x].gid, name_hash[matches[x].gid]);
                    sc.pos_of[matches[x].gid] = (uint32_t)x;
                }
                sc.slots.order(sc.slot_order);
                sc.reordered.clear();
                for (uint32_t g : sc.slot_order)
                    sc.reordered.push_back(matches[sc.pos_of[g]]);
                matches.swap(sc.reordered);
            }
        }
        else if (shared_targets && matches.size() > 1) // one order whoever did the merge: by the target's id in the level
            std::sort(matches.begin(), matches.end(), [](const MatchEntry& a, const MatchEntry& b) { return a.gid < b.gid; });
        bool classified = false;
        if (max_count_read > 0) // :753-808
        {
            const size_t threshold_filter =
                prefiltered ? 0 : max_count_read - ceil_share(max_count_read - min_count_read, level.rel_filter);
            // filter_matches (:579-613)
            size_t       kept = 0;
            uint32_t     first_kept = 0;
            size_t       first_kept_count = 0;
            const size_t all_mark = buf_all.size(); // lines of a read that ends up unclassified are dropped
            kept_gids.clear();
            for (auto const& me : matches)
            {
                touch(me.gid);
                if (me.count >= (double)threshold_filter)
                {
                    if (level.fpr_query < 1.0 && !me.fpr_ok)
                    {
                        double q = 1;
                        ++n_fpr_evals;
                        for (size_t i = 0; i <= me.count; i++)
                            q -= binomial_coefficient(n_hashes, i) * pow(me.fpr, i) * pow(1 - me.fpr, n_hashes - i);
                        if (q > level.fpr_query)
                        {
                            total.dropped_by_fpr_query++; // (only the level's total is ever reported: no per-target row to touch)
                            continue;
                        }
                    }
                    tally_at(per_target, me.gid).matches++;
                    if (kept == 0)
                    {
                        first_kept       = me.gid;
                        first_kept_count = me.count;
                    }
                    ++kept;
                    kept_gids.push_back(me.gid);
                    if (o_all)
                        append_line(buf_all, rb.id(r), node_names[me.gid], me.count);
                }
                else
                    total.dropped_by_rel_filter++;
            }
            if (kept > 0)
            {
                classified = true;
                total.reads_classified++;
                total.minimisers_of_classified += n_hashes;
                total.best_match_minimisers += max_count_read;
                if (kept == 1) // :773-778 / :790-793: the single kept match with its own count
                {
                    tally_at(per_target, first_kept).unique_reads++;
                    if (o_lca)
                        append_line(buf_lca, rb.id(r), node_names[first_kept], first_kept_count);
                }
                else if (!config.skip_lca) // lca_matches :615-627
                {
                    kept_targets.clear();
                    for (uint32_t g : kept_gids)
                        kept_targets.push_back(node_names[g]);
                    const std::string target_lca = lca.getLCA(kept_targets);
                    touch(known_nid(target_lca));
                    tally_at(per_target, known_nid(target_lca)).lca_reads++;
                    if (o_lca)
                        append_line(buf_lca, rb.id(r), target_lca, max_count_read);
                }
                else // :794-799
                {
                    touch(known_nid(config.tax_root_node));
                    tally_at(per_target, known_nid(config.tax_root_node)).lca_reads++;
                }
            }
            else
                buf_all.resize(all_mark);
        }
        if (classified)
            continue;
        if (!last_level) // :811-820
        {
            const std::string_view id = rb.id(r);
            left.id_buf.append(id);
            left.id_off.push_back(left.id_buf.size());
            left.bases.insert(left.bases.end(), rb.seq1(r), rb.seq1(r) + read1_len);
            left.off1.push_back(left.bases.size());
            if (rb.paired)
            {
                left2.insert(left2.end(), rb.seq2(r), rb.seq2(r) + read2_len);
                left.off2.push_back(left2.size());
            }
        }
        else if (o_unc) // :821-825
        {
            buf_unc.append(rb.id(r));
            buf_unc += '\n';
        }
    }
    total.dropped_by_rel_filter += res.dropped_rel_filter;
    total.dropped_by_fpr_query += res.dropped_fpr_query;
    cx.diag_unmerged.fetch_add(n_unmerged, std::memory_order_relaxed);
    cx.diag_fpr_evals.fetch_add(n_fpr_evals, std::memory_order_relaxed);
    if (!last_level && left.size() != 0)
    {
        finalize_batch(left, left2);
        po.left     = std::move(left);
        po.has_left = true;
    }
    std::lock_guard<std::mutex> lk(cx.timing_mutex);
    cx.sec_post += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

} // namespace gnhost
