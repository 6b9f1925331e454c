// Stream replay of a captured HIP graph.
//
// The train step is captured once (torch.cuda.CUDAGraph, unast_amd/graphed.py): ~1 700 kernel / memset / memcpy nodes on four
// forked streams.  hipGraphLaunch replays such a multi-branch graph correctly but slowly on ROCm 7.2 -- 39-41 ms per config-3
// step against 33 ms for the same launches issued eagerly -- while issuing them from Python costs 27 ms of host time per step.
// This executor keeps the capture (static addresses, arguments frozen in the nodes) and replaces the launch: it reads the
// nodes and their dependency edges back (hipGraphGetNodes / hipGraphNodeGetDependencies / hipGraph*NodeGetParams), lays the DAG
// out on a handful of ordinary HIP streams -- a node continues the stream of a dependency that is still that stream's tail,
// otherwise it opens a branch on another stream; every cross-stream edge becomes an event record + stream wait -- and replays
// that plan with plain hipLaunchKernel / hipMemsetAsync / hipMemcpyAsync calls from C++ (3-4 us per node, no Python).
// The kernels then run under the normal stream scheduler, exactly as when they were launched one by one.
//
// Anything the executor does not understand (host nodes, child graphs, 2-D/3-D copies, kernels without a kernelParams array)
// makes plan creation fail with an error string; the caller then falls back to hipGraphLaunch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>
#include "common.h"
#include "graph_layout.h"
#include "../../include/unast_hip.h"

namespace {

enum OpKind { OP_KERNEL = 0, OP_MEMSET = 1, OP_MEMCPY = 2, OP_RECORD = 3, OP_WAIT = 4, OP_ALLREDUCE = 5 };

struct PlanOp {
    int kind;
    int stream;                 // executor stream index
    int event;                  // OP_RECORD / OP_WAIT
    hipKernelNodeParams kp;     // OP_KERNEL
    hipMemsetParams ms;         // OP_MEMSET
    void* cdst; const void* csrc; size_t cbytes; hipMemcpyKind ckind;   // OP_MEMCPY
    size_t crows, cdpitch, cspitch;                                     // rows > 1: a pitched (2-D) copy of `crows` rows of `cbytes` bytes
    float* rbuf; long long rcount;                                      // OP_ALLREDUCE (a captured unast_allreduce_marker)
};

struct Plan {
    std::vector<PlanOp> ops;
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> events;
    hipEvent_t begin = nullptr;
    std::vector<hipEvent_t> ends;
    std::vector<hipEvent_t> ar_done;   // one per OP_ALLREDUCE, in plan order: the next collective waits for the previous one (one communicator, one collective at a time)
    int kernels = 0, memsets = 0, memcpys = 0, cross_edges = 0, allreduces = 0;
    int64_t comm = 0;           // RCCL communicator (unast_comm_init) the plan's collectives are issued on
};

}  // namespace

extern "C" const void* unast_allreduce_marker_func(void);       // comm.cpp
extern "C" int unast_graph_plan_destroy(int64_t handle);

// Which stream was a node captured on?  The graph does not say; the capture does: right after a launch the capturing stream's dependency
// set (hipStreamGetCaptureInfo_v2) is exactly the node that launch created.  The Python side calls unast_capture_note(stream) after every
// call of this library while a capture is open (unast_amd/_lib.py check()), unast_capture_reset() when one begins.
static std::mutex g_label_mutex;
static std::unordered_map<hipGraphNode_t, hipStream_t> g_labels;

extern "C" int unast_capture_reset(void) {
    std::lock_guard<std::mutex> lock(g_label_mutex);
    g_labels.clear();
    return UNAST_OK;
}

extern "C" int unast_capture_note(hipStream_t stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    hipGraph_t g = nullptr;
    const hipGraphNode_t* dep = nullptr;
    size_t nd = 0;
    if (hipStreamGetCaptureInfo_v2(stream, &st, &id, &g, &dep, &nd) != hipSuccess || st != hipStreamCaptureStatusActive) return UNAST_OK;    // not capturing: nothing to note
    std::lock_guard<std::mutex> lock(g_label_mutex);
    for (size_t i = 0; i < nd; ++i) g_labels.emplace(dep[i], stream);       // (a node keeps the stream that noted it first: the one that launched it)
    return UNAST_OK;
}

// Inherited dependencies of a side stream, pruned (engine._Segment._backward).  Inside a capture every hand-off between two side streams is
// relayed by the origin stream, whose dependency set therefore accumulates the producers of ALL hand-offs so far -- and a side stream that
// syncs with the origin inherits them all (1.6 ms of the replayed step).  At the start of a backward segment the only foreign
// producers it can need are those of its incoming gradients (`keep`: their producers' tail nodes, noted when they were produced): every
// other node in the stream's dependency set that was captured on ANOTHER side stream is dropped.  Nodes of the stream itself, of the
// origin, and nodes nobody noted (launches that did not pass through this library) always stay.
extern "C" int unast_capture_prune(hipStream_t stream, hipStream_t origin, void** keep, int nkeep) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    hipGraph_t g = nullptr;
    const hipGraphNode_t* dep = nullptr;
    size_t nd = 0;
    if (hipStreamGetCaptureInfo_v2(stream, &st, &id, &g, &dep, &nd) != hipSuccess || st != hipStreamCaptureStatusActive || nd == 0) return 0;
    std::vector<hipGraphNode_t> kept;
    {
        std::lock_guard<std::mutex> lock(g_label_mutex);
        for (size_t i = 0; i < nd; ++i) {
            auto it = g_labels.find(dep[i]);
            bool stay = it == g_labels.end() || it->second == stream || it->second == origin;
            for (int k = 0; k < nkeep && !stay; ++k) stay = keep[k] == (void*)dep[i];
            if (stay) kept.push_back(dep[i]);
        }
    }
    if (kept.empty() || kept.size() == nd) return 0;
    const hipError_t e = hipStreamUpdateCaptureDependencies(stream, kept.data(), kept.size(), hipStreamSetCaptureDependencies);
    if (e != hipSuccess) { unast_set_error(UNAST_ERR_LAUNCH, "unast_capture_prune: hipStreamUpdateCaptureDependencies failed (%d: %s)", (int)e, hipGetErrorString(e)); return -1; }
    return (int)(nd - kept.size());
}

// The capturing stream's current dependency set (the nodes its next captured node will depend on); -1 when the stream is not capturing.
extern "C" int unast_capture_get_deps(hipStream_t stream, void** out, int cap) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    hipGraph_t g = nullptr;
    const hipGraphNode_t* dep = nullptr;
    size_t nd = 0;
    if (hipStreamGetCaptureInfo_v2(stream, &st, &id, &g, &dep, &nd) != hipSuccess || st != hipStreamCaptureStatusActive) return -1;
    if ((int)nd > cap) return -2;
    for (size_t i = 0; i < nd; ++i) out[i] = (void*)dep[i];
    return (int)nd;
}

extern "C" int64_t unast_graph_plan_create(void* graph_handle, int nstreams) {
    hipGraph_t graph = (hipGraph_t)graph_handle;
    if (!graph || nstreams < 1 || nstreams > 16) { unast_set_error(UNAST_ERR_ARG, "unast_graph_plan_create: bad arguments"); return 0; }
    size_t n = 0;
    if (hipGraphGetNodes(graph, nullptr, &n) != hipSuccess || n == 0) { unast_set_error(UNAST_ERR_ARG, "unast_graph_plan_create: hipGraphGetNodes failed or empty graph"); return 0; }
    std::vector<hipGraphNode_t> nodes(n);
    if (hipGraphGetNodes(graph, nodes.data(), &n) != hipSuccess) { unast_set_error(UNAST_ERR_ARG, "unast_graph_plan_create: hipGraphGetNodes failed"); return 0; }
    std::map<hipGraphNode_t, int> index;
    for (size_t i = 0; i < n; ++i) index[nodes[i]] = (int)i;
    std::vector<std::vector<int>> deps(n);
    for (size_t i = 0; i < n; ++i) {
        size_t nd = 0;
        if (hipGraphNodeGetDependencies(nodes[i], nullptr, &nd) != hipSuccess) { unast_set_error(UNAST_ERR_ARG, "unast_graph_plan_create: hipGraphNodeGetDependencies failed"); return 0; }
        if (nd) {
            std::vector<hipGraphNode_t> d(nd);
            if (hipGraphNodeGetDependencies(nodes[i], d.data(), &nd) != hipSuccess) { unast_set_error(UNAST_ERR_ARG, "unast_graph_plan_create: hipGraphNodeGetDependencies failed"); return 0; }
            for (size_t j = 0; j < nd; ++j) {
                auto it = index.find(d[j]);
                if (it == index.end()) { unast_set_error(UNAST_ERR_ARG, "unast_graph_plan_create: dependency outside the graph"); return 0; }
                deps[i].push_back(it->second);
            }
        }
    }
    // ---- lay the DAG out on streams (graph_layout.h: pure host logic, unit-tested on the CPU under AddressSanitizer) -----------------
    unast_layout::Layout lay;
    std::vector<int> prio;
    {
        std::string err;
        static const bool keep_chains = [] { const char* e = getenv("UNAST_REPLAY_KEEP_CHAINS"); return !(e && e[0] == '0'); }();
        // Off by default: with the eager step's exact stream structure (5 streams, 96 hand-offs) the replayed step measured 29.7 ms against
        // 28.5 ms from the DAG layout below and 27.3 ms eager (tools/ab_labels.sh): what separates replay from eager is not the
        // layout but the capture itself, which has to route side-stream hand-offs through its origin stream (engine._ViaOrigin).
        static const bool use_labels = [] { const char* e = getenv("UNAST_REPLAY_LABELS"); return e && e[0] == '1'; }();
        std::vector<int> label;
        if (use_labels) {                                            // the streams the nodes were captured on, numbered by first appearance
            std::lock_guard<std::mutex> lock(g_label_mutex);
            std::map<hipStream_t, int> number;
            size_t known = 0;
            label.assign(n, -1);
            for (size_t i = 0; i < n; ++i) {
                auto it = g_labels.find(nodes[i]);
                if (it == g_labels.end()) continue;
                auto ins = number.emplace(it->second, (int)number.size());
                label[i] = ins.first->second;
                ++known;
            }
            if (known * 2 < n || number.size() > 16) label.clear();  // a capture nobody noted (or a foreign graph): lay it out from the DAG
            else {
                nstreams = (int)number.size();
                prio.assign(nstreams, 0);                            // a plan stream inherits the priority of the stream it stands for
                for (auto& kv : number) { int p = 0; if (hipStreamGetPriority(kv.first, &p) == hipSuccess) prio[kv.second] = p; }
            }
        }
        if (!unast_layout::plan_layout((int)n, deps, nstreams, lay, err, keep_chains, label.empty() ? nullptr : &label)) { unast_set_error(UNAST_ERR_ARG, "unast_graph_plan_create: %s", err.c_str()); return 0; }
    }
    Plan* plan = new Plan();
    plan->events.assign(lay.event_node.size(), nullptr);
    plan->cross_edges = lay.cross_edges;
    std::vector<PlanOp> ops;
    ops.reserve(lay.ops.size());
    for (const unast_layout::LayoutOp& lo : lay.ops) {
        if (lo.kind == unast_layout::L_WAIT || lo.kind == unast_layout::L_RECORD) {
            PlanOp w{}; w.kind = lo.kind == unast_layout::L_WAIT ? OP_WAIT : OP_RECORD; w.stream = lo.stream; w.event = lo.id;
            ops.push_back(w);
            continue;
        }
        const int v = lo.id, s = lo.stream;
        hipGraphNodeType type;
        if (hipGraphNodeGetType(nodes[v], &type) != hipSuccess) { unast_graph_plan_destroy((int64_t)(intptr_t)plan); unast_set_error(UNAST_ERR_ARG, "unast_graph_plan_create: hipGraphNodeGetType failed"); return 0; }
        PlanOp op{};
        op.stream = s;
        if (type == hipGraphNodeTypeKernel) {
            op.kind = OP_KERNEL;
            if (hipGraphKernelNodeGetParams(nodes[v], &op.kp) != hipSuccess || !op.kp.func || !op.kp.kernelParams) {
                unast_graph_plan_destroy((int64_t)(intptr_t)plan); unast_set_error(UNAST_ERR_ARG, "unast_graph_plan_create: kernel node %d without a kernelParams array", v); return 0;
            }
            if (op.kp.func == unast_allreduce_marker_func()) {       // a gradient exchange: RCCL is called in its place at replay time
                op.kind = OP_ALLREDUCE;
                op.rbuf = *reinterpret_cast<float**>(op.kp.kernelParams[0]);
                op.rcount = *reinterpret_cast<long long*>(op.kp.kernelParams[1]);
                if (!op.rbuf || op.rcount <= 0) { unast_graph_plan_destroy((int64_t)(intptr_t)plan); unast_set_error(UNAST_ERR_ARG, "unast_graph_plan_create: bad all-reduce marker (node %d)", v); return 0; }
                ++plan->allreduces;
            } else {
                ++plan->kernels;
            }
        } else if (type == hipGraphNodeTypeMemset) {
            op.kind = OP_MEMSET;
            if (hipGraphMemsetNodeGetParams(nodes[v], &op.ms) != hipSuccess || op.ms.height > 1 || (op.ms.elementSize != 1 && op.ms.elementSize != 2 && op.ms.elementSize != 4)) {
                unast_graph_plan_destroy((int64_t)(intptr_t)plan); unast_set_error(UNAST_ERR_ARG, "unast_graph_plan_create: unsupported memset node %d", v); return 0;
            }
            ++plan->memsets;
        } else if (type == hipGraphNodeTypeMemcpy) {
            hipMemcpy3DParms cp;
            memset(&cp, 0, sizeof(cp));
            const hipError_t ge = hipGraphMemcpyNodeGetParams(nodes[v], &cp);
            if (ge != hipSuccess || cp.extent.depth > 1 || cp.srcArray || cp.dstArray ||
                cp.srcPos.x || cp.srcPos.y || cp.srcPos.z || cp.dstPos.x || cp.dstPos.y || cp.dstPos.z) {
                unast_graph_plan_destroy((int64_t)(intptr_t)plan);
                unast_set_error(UNAST_ERR_ARG, "unast_graph_plan_create: unsupported memcpy node %d (only linear and pitched 2-D copies; get-params rc %d, extent %zu x %zu x %zu, "
                                "src pos %zu %zu %zu, dst pos %zu %zu %zu, arrays %d %d, kind %d)", v, (int)ge, cp.extent.width, cp.extent.height, cp.extent.depth,
                                cp.srcPos.x, cp.srcPos.y, cp.srcPos.z, cp.dstPos.x, cp.dstPos.y, cp.dstPos.z, cp.srcArray != nullptr, cp.dstArray != nullptr, (int)cp.kind);
                return 0;
            }
            op.kind = OP_MEMCPY; op.cdst = cp.dstPtr.ptr; op.csrc = cp.srcPtr.ptr; op.cbytes = cp.extent.width; op.ckind = cp.kind;
            op.crows = cp.extent.height > 1 ? cp.extent.height : 1; op.cdpitch = cp.dstPtr.pitch; op.cspitch = cp.srcPtr.pitch;
            if (op.crows > 1 && (op.cdpitch < op.cbytes || op.cspitch < op.cbytes)) {
                unast_graph_plan_destroy((int64_t)(intptr_t)plan); unast_set_error(UNAST_ERR_ARG, "unast_graph_plan_create: memcpy node %d: pitch below the row width", v); return 0;
            }
            if (!op.cdst || !op.csrc || op.cbytes == 0) {            // (a getter that returns zeroed parameters must not become a silent 0-byte copy)
                unast_graph_plan_destroy((int64_t)(intptr_t)plan); unast_set_error(UNAST_ERR_ARG, "unast_graph_plan_create: memcpy node %d without a pointer / extent", v); return 0;
            }
            ++plan->memcpys;
        } else if (type == hipGraphNodeTypeEmpty || type == hipGraphNodeTypeEventRecord || type == hipGraphNodeTypeWaitEvent) {
            op.kind = -1;                                            // ordering only: keeps its place on the stream, launches nothing
        } else {
            unast_graph_plan_destroy((int64_t)(intptr_t)plan); unast_set_error(UNAST_ERR_ARG, "unast_graph_plan_create: unsupported node type %d (node %d)", (int)type, v); return 0;
        }
        ops.push_back(op);
    }
    for (auto& op : ops) if (op.kind >= 0) plan->ops.push_back(op);
    // streams, events
    plan->streams.assign(nstreams, nullptr);
    plan->ends.assign(nstreams, nullptr);
    bool ok = true;
    for (int t = 0; t < nstreams && ok; ++t)
        ok = (t < (int)prio.size() && prio[t] != 0 ? hipStreamCreateWithPriority(&plan->streams[t], hipStreamNonBlocking, prio[t])
                                                   : hipStreamCreateWithFlags(&plan->streams[t], hipStreamNonBlocking)) == hipSuccess;
    for (size_t i = 0; i < plan->events.size() && ok; ++i) ok = hipEventCreateWithFlags(&plan->events[i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&plan->begin, hipEventDisableTiming) == hipSuccess;
    for (int t = 0; t < nstreams && ok; ++t) ok = hipEventCreateWithFlags(&plan->ends[t], hipEventDisableTiming) == hipSuccess;
    plan->ar_done.assign((size_t)plan->allreduces, nullptr);
    for (size_t i = 0; i < plan->ar_done.size() && ok; ++i) ok = hipEventCreateWithFlags(&plan->ar_done[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) { unast_graph_plan_destroy((int64_t)(intptr_t)plan); unast_set_error(UNAST_ERR_LAUNCH, "unast_graph_plan_create: stream / event creation failed"); return 0; }
    return (int64_t)(intptr_t)plan;
}

extern "C" int unast_graph_plan_info(int64_t handle, int* out4) {
    Plan* plan = (Plan*)(intptr_t)handle;
    UNAST_REQUIRE(plan && out4, "unast_graph_plan_info: bad arguments");
    out4[0] = plan->kernels; out4[1] = plan->memsets; out4[2] = plan->memcpys; out4[3] = plan->cross_edges;
    return UNAST_OK;
}

extern "C" int unast_graph_plan_streams(int64_t handle) {
    Plan* plan = (Plan*)(intptr_t)handle;
    return plan ? (int)plan->streams.size() : -1;
}

extern "C" int unast_graph_plan_allreduces(int64_t handle) {
    Plan* plan = (Plan*)(intptr_t)handle;
    return plan ? plan->allreduces : -1;
}

extern "C" int unast_graph_plan_allreduce_counts(int64_t handle, long long* out, int n) {
    Plan* plan = (Plan*)(intptr_t)handle;
    UNAST_REQUIRE(plan && (out || n == 0), "unast_graph_plan_allreduce_counts: bad arguments");
    int i = 0;
    for (const PlanOp& op : plan->ops)
        if (op.kind == OP_ALLREDUCE) { if (i < n) out[i] = op.rcount; ++i; }
    return i;
}

extern "C" int unast_graph_plan_set_comm(int64_t handle, int64_t comm) {
    Plan* plan = (Plan*)(intptr_t)handle;
    UNAST_REQUIRE(plan, "unast_graph_plan_set_comm: null plan");
    plan->comm = comm;
    return UNAST_OK;
}

extern "C" int unast_graph_plan_replay(int64_t handle, hipStream_t origin) {
    Plan* plan = (Plan*)(intptr_t)handle;
    UNAST_REQUIRE(plan, "unast_graph_plan_replay: null plan");
    UNAST_REQUIRE(plan->allreduces == 0 || plan->comm, "unast_graph_plan_replay: the plan holds %d gradient exchanges but no communicator (unast_graph_plan_set_comm)", plan->allreduces);
    // everything enqueued on the caller's stream so far happens before the plan; the caller's stream waits for all of it at the end
    if (hipEventRecord(plan->begin, origin) != hipSuccess) return unast_set_error(UNAST_ERR_LAUNCH, "unast_graph_plan_replay: hipEventRecord failed");
    for (auto s : plan->streams) hipStreamWaitEvent(s, plan->begin, 0);
    size_t ar_i = 0;
    int ar_prev_stream = -1;
    int rc = UNAST_OK;
    for (const PlanOp& op : plan->ops) {
        hipStream_t s = plan->streams[op.stream];
        hipError_t e = hipSuccess;
        switch (op.kind) {
            case OP_KERNEL: e = hipLaunchKernel(op.kp.func, op.kp.gridDim, op.kp.blockDim, op.kp.kernelParams, op.kp.sharedMemBytes, s); break;
            case OP_MEMSET:
                if (op.ms.elementSize == 1) e = hipMemsetAsync(op.ms.dst, (int)op.ms.value, op.ms.width, s);
                else if (op.ms.elementSize == 2) e = hipMemsetD16Async((hipDeviceptr_t)op.ms.dst, (unsigned short)op.ms.value, op.ms.width, s);
                else e = hipMemsetD32Async((hipDeviceptr_t)op.ms.dst, (int)op.ms.value, op.ms.width, s);
                break;
            case OP_MEMCPY:
                e = op.crows > 1 ? hipMemcpy2DAsync(op.cdst, op.cdpitch, op.csrc, op.cspitch, op.cbytes, op.crows, op.ckind, s)
                                 : hipMemcpyAsync(op.cdst, op.csrc, op.cbytes, op.ckind, s);
                break;
            case OP_RECORD: e = hipEventRecord(plan->events[op.event], s); break;
            case OP_WAIT: e = hipStreamWaitEvent(s, plan->events[op.event], 0); break;
            case OP_ALLREDUCE: {
                // Collectives of one RCCL communicator must not run concurrently nor in an order that differs between ranks: whatever streams
                // the layout put them on, each waits for the one issued before it (plan order = capture order: graph_layout.h takes the
                // smallest node index among the ready ones, and hipGraphGetNodes lists nodes as they were captured -- the order in which the
                // eager step's ddp._issue calls come, so a rank that replays and a rank that runs the same step eagerly pair up).
                if (ar_i > 0 && ar_prev_stream != op.stream) e = hipStreamWaitEvent(s, plan->ar_done[ar_i - 1], 0);
                if (e != hipSuccess) break;
                rc = unast_allreduce(plan->comm, op.rbuf, op.rcount, s);
                if (rc) break;
                e = hipEventRecord(plan->ar_done[ar_i], s);
                ar_prev_stream = op.stream; ++ar_i;
                break;
            }
            default: break;
        }
        if (rc) break;
        if (e != hipSuccess) { rc = unast_set_error(UNAST_ERR_LAUNCH, "unast_graph_plan_replay: op kind %d failed: %s", op.kind, hipGetErrorString(e)); break; }
    }
    // Also after a failure: the caller's stream must not run ahead of what the plan streams already hold (the caller raises; kernels of
    // the failed replay may still be running against the buffers it is about to free or reuse).
    for (size_t t = 0; t < plan->streams.size(); ++t) {
        (void)hipEventRecord(plan->ends[t], plan->streams[t]);
        (void)hipStreamWaitEvent(origin, plan->ends[t], 0);
    }
    if (rc) return rc;
    return unast_check_launch("unast_graph_plan_replay");
}

extern "C" int unast_graph_plan_destroy(int64_t handle) {
    Plan* plan = (Plan*)(intptr_t)handle;
    if (!plan) return UNAST_OK;
    for (auto s : plan->streams) if (s) { hipStreamSynchronize(s); hipStreamDestroy(s); }
    for (auto e : plan->events) if (e) hipEventDestroy(e);
    if (plan->begin) hipEventDestroy(plan->begin);
    for (auto e : plan->ends) if (e) hipEventDestroy(e);
    for (auto e : plan->ar_done) if (e) hipEventDestroy(e);
    delete plan;
    return UNAST_OK;
}
