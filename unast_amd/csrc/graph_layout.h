// Laying a dependency DAG out on a handful of in-order streams: the planning half of the stream-replay executor (graph_exec.cpp),
// free of any HIP call so that it builds and runs on the CPU under -fsanitize=address (tests/native/test_graph_layout.cpp).
//
// Input: n nodes, deps[i] = the nodes i waits for.  Output: the nodes in a topological order (smallest index first among the ready
// ones), each on a stream, with a WAIT in front of a node for every dependency that stream order does not already cover and a RECORD
// behind every node some other stream waits for.  A node continues the stream of a dependency that is still that stream's tail (the
// newest such), otherwise it opens a branch on a stream that is empty, whose branch has ended (its tail has no unplaced successors),
// or -- failing both -- the least recently used one.
// keep_chains (round 4): at a fork the stream goes to the successor with the LONGEST way still ahead of it (in nodes, to a sink); a
// shorter-lived successor that happens to come first in the order -- a weight gradient issued in front of the input gradient that
// continues the backward chain -- opens the branch instead.  Without it the long dependent chain of a step hops streams at every such
// fork (one event hand-off each) and the side work takes over the stream its producer ran on.
// label (round 4, optional): label[v] = the stream node v was CAPTURED on (-1: unknown), as the capture noted it launch by launch
// (graph_exec.cpp, unast_capture_note).  Then the replay keeps the program's own stream structure -- the one the eager step runs with and
// was tuned on -- instead of inventing one from the DAG: node v goes to stream label[v]; an unlabelled node (a launch that did not pass
// through the C ABI) joins the stream of its first labelled successor, else of its first labelled dependency, else stream 0.
#pragma once
#include <algorithm>
#include <functional>
#include <string>
#include <vector>

namespace unast_layout {

enum { L_NODE = 0, L_RECORD = 3, L_WAIT = 4 };

struct LayoutOp {
    int kind;       // L_NODE / L_RECORD / L_WAIT
    int stream;
    int id;         // L_NODE: node index; L_RECORD / L_WAIT: event index
};

struct Layout {
    std::vector<LayoutOp> ops;
    std::vector<int> event_node;        // event index -> the node it is recorded behind
    int cross_edges = 0;
};

inline bool plan_layout(int n, const std::vector<std::vector<int>>& deps, int nstreams, Layout& out, std::string& err, bool keep_chains = false,
                        const std::vector<int>* label = nullptr) {
    out = Layout();
    if (n <= 0 || nstreams < 1 || (int)deps.size() != n) { err = "bad arguments"; return false; }
    std::vector<std::vector<int>> succ(n);
    for (int i = 0; i < n; ++i)
        for (int d : deps[i]) {
            if (d < 0 || d >= n || d == i) { err = "dependency outside the graph"; return false; }
            succ[d].push_back(i);
        }
    std::vector<int> indeg(n), order;
    order.reserve(n);
    {
        std::vector<int> ready;
        for (int i = 0; i < n; ++i) { indeg[i] = (int)deps[i].size(); if (!indeg[i]) ready.push_back(i); }
        std::make_heap(ready.begin(), ready.end(), std::greater<int>());
        while (!ready.empty()) {
            std::pop_heap(ready.begin(), ready.end(), std::greater<int>());
            const int v = ready.back(); ready.pop_back();
            order.push_back(v);
            for (int s : succ[v]) if (--indeg[s] == 0) { ready.push_back(s); std::push_heap(ready.begin(), ready.end(), std::greater<int>()); }
        }
        if ((int)order.size() != n) { err = "graph has a cycle"; return false; }
    }
    // way[v]: nodes on the longest path from v to a sink (v included)
    std::vector<int> way(n, 1);
    if (keep_chains)
        for (int oi = n - 1; oi >= 0; --oi) {
            const int v = order[oi];
            for (int s : succ[v]) way[v] = std::max(way[v], way[s] + 1);
        }
    std::vector<int> fixed;                     // the stream of every node when labels are given
    if (label) {
        if ((int)label->size() != n) { err = "label array of the wrong size"; return false; }
        fixed = *label;
        for (int v = 0; v < n; ++v) if (fixed[v] >= nstreams) { err = "label beyond the stream count"; return false; }
        for (int oi = n - 1; oi >= 0; --oi) {   // successors first
            const int v = order[oi];
            if (fixed[v] >= 0) continue;
            int best = -1;
            for (int u : succ[v]) if (fixed[u] >= 0 && (best < 0 || u < best)) best = u;
            if (best >= 0) fixed[v] = fixed[best];
        }
        for (int oi = 0; oi < n; ++oi) {
            const int v = order[oi];
            if (fixed[v] >= 0) continue;
            for (int d : deps[v]) if (fixed[d] >= 0) { fixed[v] = fixed[d]; break; }
            if (fixed[v] < 0) fixed[v] = 0;
        }
    }
    std::vector<int> stream_of(n, -1), tail(nstreams, -1), remaining(n), event_of(n, -1), pos_in_order(n, 0);
    std::vector<long long> last_use(nstreams, -1);
    for (int i = 0; i < n; ++i) remaining[i] = (int)succ[i].size();
    std::vector<std::vector<int>> waited(nstreams, std::vector<int>(nstreams, -1));   // waited[s][t]: position of the newest node of stream t that s is behind
    std::vector<LayoutOp> ops;
    ops.reserve(3 * (size_t)n);
    long long tick = 0;
    for (int oi = 0; oi < n; ++oi) {
        const int v = order[oi];
        pos_in_order[v] = oi;
        int s = label ? fixed[v] : -1, best = -1;
        if (!label) for (int d : deps[v]) {
            const int sd = stream_of[d];
            if (tail[sd] != d || pos_in_order[d] <= best) continue;
            if (keep_chains) {            // does a later successor of d have more of the graph behind it?  Then the stream is kept for that one.
                bool outlived = false;
                for (int u : succ[d]) if (u != v && stream_of[u] < 0 && way[u] > way[v]) { outlived = true; break; }
                if (outlived) continue;
            }
            best = pos_in_order[d]; s = sd;
        }
        if (s < 0) {
            for (int t = 0; t < nstreams && s < 0; ++t) if (tail[t] < 0) s = t;
            if (s < 0) {
                long long lru = -1;
                for (int t = 0; t < nstreams; ++t)
                    if (remaining[tail[t]] == 0 && (s < 0 || last_use[t] < lru)) { s = t; lru = last_use[t]; }
            }
            if (s < 0) {
                long long lru = 0;
                for (int t = 0; t < nstreams; ++t) if (s < 0 || last_use[t] < lru) { s = t; lru = last_use[t]; }
            }
        }
        for (int d : deps[v]) {
            --remaining[d];
            const int sd = stream_of[d];
            if (sd == s) continue;                                   // stream order covers it (d was placed earlier on the same stream)
            if (waited[s][sd] >= pos_in_order[d]) continue;          // already behind a later node of that stream
            if (event_of[d] < 0) { event_of[d] = (int)out.event_node.size(); out.event_node.push_back(d); }
            ops.push_back(LayoutOp{L_WAIT, s, event_of[d]});
            waited[s][sd] = pos_in_order[d];
            ++out.cross_edges;
        }
        ops.push_back(LayoutOp{L_NODE, s, v});
        // the record for cross-stream consumers sits right behind the node; whether it is needed is known only later, so every node gets
        // a slot (id = node index for now) and unused ones are dropped below
        ops.push_back(LayoutOp{L_RECORD, s, v});
        stream_of[v] = s; tail[s] = v; last_use[s] = ++tick;
    }
    out.ops.reserve(ops.size());
    for (const LayoutOp& op : ops) {
        if (op.kind == L_RECORD) {
            if (event_of[op.id] < 0) continue;
            out.ops.push_back(LayoutOp{L_RECORD, op.stream, event_of[op.id]});
        } else {
            out.ops.push_back(op);
        }
    }
    return true;
}

}  // namespace unast_layout
