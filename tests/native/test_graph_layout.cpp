// CPU unit run of the stream-replay planner (unast_amd/csrc/graph_layout.h), built with -fsanitize=address,undefined by
// tests/test_cpu_host.py.  Synthetic DAGs: chains, diamonds, wide fork/joins, the four-branch shape of the captured train step, random
// DAGs; every layout is checked by simulating in-order streams with events: a node may only run when all its dependencies have run.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>
#include "../../unast_amd/csrc/graph_layout.h"

using namespace unast_layout;

static int failures = 0;
#define CHECK(c, ...) do { if (!(c)) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); ++failures; } } while (0)

// Replays the layout the way the executor does (ops issued in list order onto in-order streams; WAIT blocks the stream until the event's
// RECORD has executed) with an adversarial scheduler: among the streams whose head op can run, always take the LAST one, so that nothing
// is ordered by accident of issue order.
static bool simulate(int n, const std::vector<std::vector<int>>& deps, int nstreams, const Layout& lay, std::string& why) {
    std::vector<std::vector<LayoutOp>> q(nstreams);
    for (const LayoutOp& op : lay.ops) {
        if (op.stream < 0 || op.stream >= nstreams) { why = "stream index out of range"; return false; }
        q[op.stream].push_back(op);
    }
    std::vector<size_t> head(nstreams, 0);
    std::vector<char> done(n, 0), fired(lay.event_node.size(), 0);
    std::vector<int> seen(n, 0);
    size_t left = lay.ops.size();
    while (left) {
        bool progressed = false;
        for (int t = nstreams - 1; t >= 0; --t) {
            if (head[t] >= q[t].size()) continue;
            const LayoutOp& op = q[t][head[t]];
            if (op.kind == L_WAIT) {
                if (op.id < 0 || op.id >= (int)fired.size()) { why = "event index out of range"; return false; }
                if (!fired[op.id]) continue;
            } else if (op.kind == L_RECORD) {
                if (op.id < 0 || op.id >= (int)fired.size()) { why = "event index out of range"; return false; }
                if (!done[lay.event_node[op.id]]) { why = "record ahead of its node"; return false; }
                fired[op.id] = 1;
            } else {
                if (op.id < 0 || op.id >= n) { why = "node index out of range"; return false; }
                for (int d : deps[op.id]) if (!done[d]) { why = "node " + std::to_string(op.id) + " ran before dependency " + std::to_string(d); return false; }
                done[op.id] = 1; ++seen[op.id];
            }
            ++head[t]; --left; progressed = true;
            break;
        }
        if (!progressed) { why = "deadlock"; return false; }
    }
    for (int i = 0; i < n; ++i) if (seen[i] != 1) { why = "node " + std::to_string(i) + " placed " + std::to_string(seen[i]) + " times"; return false; }
    return true;
}

static void run_case_mode(const char* name, int n, const std::vector<std::vector<int>>& deps, int nstreams, int max_cross, bool keep_chains);
static void run_case(const char* name, int n, const std::vector<std::vector<int>>& deps, int nstreams, int max_cross = -1) {
    run_case_mode(name, n, deps, nstreams, max_cross, false);
    run_case_mode(name, n, deps, nstreams, max_cross, true);          // the shipped policy (graph_exec.cpp): streams kept for the longer chain at a fork
}

static void run_case_mode(const char* name, int n, const std::vector<std::vector<int>>& deps, int nstreams, int max_cross, bool keep_chains) {
    Layout lay; std::string err, why;
    const bool ok = plan_layout(n, deps, nstreams, lay, err, keep_chains);
    CHECK(ok, "%s: plan_layout failed: %s", name, err.c_str());
    if (!ok) return;
    CHECK(simulate(n, deps, nstreams, lay, why), "%s (%d streams): %s", name, nstreams, why.c_str());
    int waits = 0, records = 0;
    // hipStreamWaitEvent on an event whose record has not been ISSUED yet does not wait: every record must precede its waits in issue order
    std::vector<char> issued(lay.event_node.size(), 0);
    for (const LayoutOp& op : lay.ops) {
        waits += op.kind == L_WAIT; records += op.kind == L_RECORD;
        if (op.kind == L_RECORD) { CHECK(!issued[op.id], "%s: event %d recorded twice", name, op.id); issued[op.id] = 1; }
        if (op.kind == L_WAIT) CHECK(issued[op.id], "%s: wait on event %d issued before its record", name, op.id);
    }
    CHECK(waits == lay.cross_edges, "%s: cross_edges %d != waits %d", name, lay.cross_edges, waits);
    CHECK(records == (int)lay.event_node.size(), "%s: %d records for %d events", name, records, (int)lay.event_node.size());
    if (max_cross >= 0) CHECK(lay.cross_edges <= max_cross, "%s: %d cross-stream edges, expected <= %d", name, lay.cross_edges, max_cross);
}

int main() {
    // a chain stays on one stream: no events at all
    { const int n = 50; std::vector<std::vector<int>> d(n); for (int i = 1; i < n; ++i) d[i] = {i - 1}; for (int s = 1; s <= 4; ++s) run_case("chain", n, d, s, 0); }
    // diamond: one fork, one join
    { std::vector<std::vector<int>> d = {{}, {0}, {0}, {1, 2}}; run_case("diamond", 4, d, 1, 0); run_case("diamond", 4, d, 4, 2); }
    // wide fork / join, more branches than streams
    { const int w = 40; std::vector<std::vector<int>> d(w + 2); for (int i = 1; i <= w; ++i) d[i] = {0}; for (int i = 1; i <= w; ++i) d[w + 1].push_back(i);
      for (int s = 1; s <= 16; s *= 2) run_case("fork-join", w + 2, d, s); }
    // the captured train step's shape: four long branches that exchange a few edges, a join, then an optimizer tail
    { const int len = 300, nb = 4, n = 1 + nb * len + 20; std::vector<std::vector<int>> d(n);
      for (int b = 0; b < nb; ++b) for (int i = 0; i < len; ++i) { const int v = 1 + b * len + i; d[v] = {i ? v - 1 : 0}; if (b && i % 37 == 5) d[v].push_back(1 + (b - 1) * len + i - 3); }
      const int j = 1 + nb * len; for (int b = 0; b < nb; ++b) d[j].push_back(b * len + len);
      for (int i = 1; i < 20; ++i) d[j + i] = {j + i - 1};
      run_case("train-step", n, d, 4); run_case("train-step", n, d, 2); run_case("train-step", n, d, 1, 0); }
    // keep_chains: a backward-like chain whose every node forks a short side branch (a weight gradient) issued BEFORE the chain's next
    // node: the chain must stay on ONE stream (no event on it), the side work goes elsewhere
    { const int len = 60, n = 1 + 2 * len + 1; std::vector<std::vector<int>> d(n);       // node 0, then per step: side (2i+1), chain (2i+2); sink joins everything
      for (int i = 0; i < len; ++i) { const int prev = i ? 2 * i : 0; d[2 * i + 1] = {prev}; d[2 * i + 2] = {prev}; }
      for (int i = 0; i < len; ++i) d[n - 1].push_back(2 * i + 1);
      d[n - 1].push_back(2 * len);
      Layout lay; std::string err; CHECK(plan_layout(n, d, 4, lay, err, true), "fork-chain: %s", err.c_str());
      int chain_stream = -1; bool one = true;
      for (const LayoutOp& op : lay.ops) if (op.kind == L_NODE && op.id > 0 && op.id < n - 3 /* the last step's two nodes are equally far from the sink: a tie */ && op.id % 2 == 0) { if (chain_stream < 0) chain_stream = op.stream; one = one && op.stream == chain_stream; }
      CHECK(one, "fork-chain: the chain hops streams under keep_chains");
      run_case("fork-chain", n, d, 4); run_case("fork-chain", n, d, 2); }
    // independent nodes only
    { std::vector<std::vector<int>> d(9); run_case("independent", 9, d, 3, 0); }
    // random DAGs (duplicate dependencies included: the capture can list one twice)
    { std::mt19937 rng(1234);
      for (int rep = 0; rep < 300; ++rep) {
          const int n = 1 + (int)(rng() % 200), s = 1 + (int)(rng() % 8); std::vector<std::vector<int>> d(n);
          for (int i = 1; i < n; ++i) { const int k = (int)(rng() % 4); for (int j = 0; j < k; ++j) d[i].push_back((int)(rng() % i)); }
          // relabel so that creation order is NOT a topological order
          std::vector<int> perm(n); for (int i = 0; i < n; ++i) perm[i] = i; std::shuffle(perm.begin(), perm.end(), rng);
          std::vector<std::vector<int>> dp(n); for (int i = 0; i < n; ++i) for (int x : d[i]) dp[perm[i]].push_back(perm[x]);
          run_case("random", n, dp, s);
      } }
    // labels (the streams the nodes were captured on): every labelled node lands on its stream, unlabelled ones join a labelled neighbour,
    // and the layout is still a correct schedule under the adversarial simulator
    { std::mt19937 rng(99);
      for (int rep = 0; rep < 200; ++rep) {
          const int n = 1 + (int)(rng() % 300), s = 1 + (int)(rng() % 6); std::vector<std::vector<int>> d(n);
          for (int i = 1; i < n; ++i) { const int k = (int)(rng() % 3); for (int j = 0; j < k; ++j) d[i].push_back((int)(rng() % i)); }
          std::vector<int> label(n);
          for (int i = 0; i < n; ++i) label[i] = rng() % 5 == 0 ? -1 : (int)(rng() % s);
          Layout lay; std::string err, why;
          const bool ok = plan_layout(n, d, s, lay, err, true, &label);
          CHECK(ok, "labels: plan_layout failed: %s", err.c_str());
          if (!ok) continue;
          CHECK(simulate(n, d, s, lay, why), "labels (%d streams): %s", s, why.c_str());
          std::vector<char> issued(lay.event_node.size(), 0);
          for (const LayoutOp& op : lay.ops) {
              if (op.kind == L_NODE && label[op.id] >= 0) CHECK(op.stream == label[op.id], "labels: node %d on stream %d, captured on %d", op.id, op.stream, label[op.id]);
              if (op.kind == L_RECORD) issued[op.id] = 1;
              if (op.kind == L_WAIT) CHECK(issued[op.id], "labels: wait on event %d issued before its record", op.id);
          }
      }
      // two captured streams that hand over twice: exactly the two waits the program made, whatever the stream count says
      std::vector<std::vector<int>> d = {{}, {0}, {1}, {0}, {3, 2}, {4}, {2}, {6, 5}};
      std::vector<int> label = {0, 0, 0, 1, 1, 1, 0, 0};
      Layout lay; std::string err; CHECK(plan_layout(8, d, 2, lay, err, true, &label), "labels/two-streams: %s", err.c_str());
      CHECK(lay.cross_edges == 3, "labels/two-streams: %d cross-stream edges, expected 3", lay.cross_edges);
      std::vector<int> bad = {0, 0, 5, 1, 1, 1, 0, 0};
      CHECK(!plan_layout(8, d, 2, lay, err, true, &bad), "label beyond the stream count accepted"); }
    // rejected inputs
    { Layout lay; std::string err;
      std::vector<std::vector<int>> cyc = {{2}, {0}, {1}}; CHECK(!plan_layout(3, cyc, 2, lay, err) && err.find("cycle") != std::string::npos, "cycle not detected");
      std::vector<std::vector<int>> self = {{0}}; CHECK(!plan_layout(1, self, 2, lay, err), "self-dependency accepted");
      std::vector<std::vector<int>> out = {{}, {7}}; CHECK(!plan_layout(2, out, 2, lay, err), "dependency outside the graph accepted");
      std::vector<std::vector<int>> one = {{}}; CHECK(!plan_layout(1, one, 0, lay, err), "zero streams accepted"); CHECK(!plan_layout(0, {}, 1, lay, err), "empty graph accepted"); }
    if (failures) { std::printf("%d failure(s)\n", failures); return 1; }
    std::printf("graph_layout: all cases passed\n");
    return 0;
}
