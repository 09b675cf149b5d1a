// lca.hpp -- lowest common ancestor over string node ids: Euler tour + sparse-table RMQ, the same construction and
// query semantics as /root/reference/src/utils/include/utils/LCA.hpp:22-174 (addEdge / doEulerWalk / getLCA),
// written iteratively so that deep taxonomies cannot overflow the stack.
#pragma once

#include <cassert>
#include <string>
#include <unordered_map>
#include <vector>

namespace gnhost
{

class LCA
{
public:
    void addEdge(const std::string& father, const std::string& son)
    {
        const int f = encode(father);
        const int s = encode(son);
        if ((int)children_.size() < (int)decode_.size())
            children_.resize(decode_.size());
        children_[f].push_back(s);
    }

    void doEulerWalk(const std::string& root_node)
    {
        const int n = (int)decode_.size();
        children_.resize(n);
        first_.assign(n, -1);
        euler_.clear();
        depth_.clear();
        auto it = ids_.find(root_node);
        if (it == ids_.end())
            return;
        root_ = it->second;
        std::vector<std::pair<int, size_t>> stack;
        stack.emplace_back(it->second, 0);
        first_[it->second] = 0;
        euler_.push_back(it->second);
        depth_.push_back(0);
        while (!stack.empty())
        {
            auto& [u, idx] = stack.back();
            if (idx < children_[u].size())
            {
                const int c = children_[u][idx++];
                if (first_[c] != -1) // a node that is its own ancestor (e.g. a "1 <tab> 1" row) or has two parents:
                    continue;        // visit it once, or the walk would never end
                first_[c] = (int)euler_.size();
                euler_.push_back(c);
                depth_.push_back((int)stack.size());
                stack.emplace_back(c, 0);
            }
            else
            {
                stack.pop_back();
                if (!stack.empty())
                {
                    euler_.push_back(stack.back().first);
                    depth_.push_back((int)stack.size() - 1);
                }
            }
        }
        // sparse table of argmin(depth) (LCA.hpp:105-130)
        const int len = (int)depth_.size();
        log_          = 1;
        while ((1 << log_) <= len)
            ++log_;
        table_.assign((size_t)len * log_, 0);
        for (int i = 0; i < len; ++i)
            table_[(size_t)i * log_] = i;
        for (int j = 1; (1 << j) <= len; ++j)
            for (int i = 0; i + (1 << j) - 1 < len; ++i)
            {
                const int a = table_[(size_t)i * log_ + j - 1];
                const int b = table_[(size_t)(i + (1 << (j - 1))) * log_ + j - 1];
                table_[(size_t)i * log_ + j] = depth_[a] < depth_[b] ? a : b;
            }
    }

    // LCA.hpp:165-174 (taxIds.size() > 1)
    // A node that the walk did not reach (a subtree that is not connected to the root) has no position in the tour:
    // the only ancestor it can share with anything is taken to be the root.
    std::string getLCA(const std::vector<std::string>& taxIds) const
    {
        int lca = pair(id_of(taxIds[0]), id_of(taxIds[1]));
        for (size_t i = 2; i < taxIds.size(); ++i)
            lca = pair(lca, id_of(taxIds[i]));
        return decode_.at(lca);
    }

private:
    int encode(const std::string& s)
    {
        auto it = ids_.find(s);
        if (it != ids_.end())
            return it->second;
        const int id = (int)decode_.size();
        ids_.emplace(s, id);
        decode_.push_back(s);
        return id;
    }
    int id_of(const std::string& s) const
    {
        auto it = ids_.find(s);
        return it == ids_.end() ? root_ : it->second;
    }
    int pair(int u, int v) const
    {
        if (u == v)
            return u;
        int i = first_[u], j = first_[v];
        if (i < 0 || j < 0)
            return root_;
        if (i > j)
            std::swap(i, j);
        int k = 0;
        while ((1 << (k + 1)) <= j - i + 1)
            ++k;
        const int a = table_[(size_t)i * log_ + k];
        const int b = table_[(size_t)(j - (1 << k) + 1) * log_ + k];
        return euler_[depth_[a] <= depth_[b] ? a : b];
    }

    std::unordered_map<std::string, int> ids_;
    std::vector<std::string>             decode_;
    std::vector<std::vector<int>>        children_;
    std::vector<int>                     euler_, depth_, first_, table_;
    int                                  log_ = 1, root_ = 0;
};

} // namespace gnhost
