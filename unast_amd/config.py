"""Run-time knobs of the HIP path."""
import os

# MFMA operand precision: 3 = split-bf16 (hi/lo pairs, three MFMAs per product, ~fp32-faithful; meets the 1e-3
# parity target of BASELINE.json), 1 = plain bf16 operands (fast mode; ~1e-2 relative error end to end).
NSPLIT = 1 if os.environ.get("UNAST_PREC", "bf16x3").lower() in ("bf16", "1") else 3


def set_precision(name):
    global NSPLIT
    NSPLIT = {"bf16": 1, "bf16x3": 3}[name]


def precision_name():
    return "bf16" if NSPLIT == 1 else "bf16x3"


# Weights as GEMM B operands are read from a pre-split copy kept next to the flat parameter store (engine.FlatStore);
# off = the kernels re-split the fp32 weights in every row panel (bit-identical results; for A/B timing only).
PRESPLIT_WEIGHTS = os.environ.get("UNAST_PRESPLIT", "1") != "0"

# Inside the train-step functions the text side, the speech side and the discriminator run on three HIP streams (their
# launches under-fill the chip one at a time); 0 = everything on the caller's stream.  Direct calls of model methods outside
# those functions are always single-stream.
SIDE_STREAMS = os.environ.get("UNAST_SIDE_STREAMS", "1") != "0"

# infer_sequence replays ONE captured HIP graph per decoded position (the step is ~35 small launches and otherwise bound by
# host launch overhead); 0 = launch every kernel from Python.
DECODE_GRAPH = os.environ.get("UNAST_DECODE_GRAPH", "1") != "0"

# Weight-gradient GEMMs (and their split-K reductions) are issued on a companion stream of the side stream they come from:
# they are off the backward pass's critical chain (their results are read by the optimizer only) and bandwidth-bound, while
# the chain they leave behind is attention / dgrad work.  Needs SIDE_STREAMS.
WGRAD_STREAMS = os.environ.get("UNAST_WGRAD_STREAMS", "1") != "0"
# ... only for weight gradients that reduce over at least this many tokens: the hand-off costs ~15 us of host time per launch,
# which small (launch-bound) configurations cannot hide (config 2: 22.7 -> 27.7 ms/step without this gate).
WGRAD_STREAM_MIN_TOKENS = int(os.environ.get("UNAST_WGRAD_MIN_TOKENS", "8192"))
# Only the speech side gets a companion: text + speech + discriminator + speech companion = exactly four real streams, one per
# hardware queue of HIP's default (see unast_amd/__init__.py); the text side's weight gradients are small and the
# discriminator's run fine on its own stream (same-box A/B: 37.9 ms/step either way, tools/stream_groups_ab.sh).
WGRAD_COMPANION_OF = set(x for x in os.environ.get("UNAST_WGRAD_COMPANION_OF", "speech").split(",") if x)
# The reduction of LayerNorm's gamma / beta gradient partials is off the backward chain too (same companion stream, same gate).
LN_FINALIZE_OFFLOAD = os.environ.get("UNAST_LN_FINALIZE_INLINE", "0") != "1"

# Which logical streams share a real HIP stream ("a:x,b:x" puts a and b on the stream named x).  Experiment switch.
STREAM_GROUPS = dict(kv.split(":") for kv in os.environ.get("UNAST_STREAM_GROUPS", "").split(",") if ":" in kv)

# HIP stream priority per logical stream ("speech:-1,text:0"; lower = more urgent, as in torch.cuda.Stream(priority=)): the step's critical
# chain is the speech side; what runs beside it on the other streams takes CUs from its kernels.  The stream replay creates its streams
# with the priorities of the streams the nodes were captured on (csrc/graph_exec.cpp).
STREAM_PRIORITY = {k: int(v) for k, v in (kv.split(":") for kv in os.environ.get("UNAST_STREAM_PRIO", "").split(",") if ":" in kv)}

# Fixed summation order of every fp32 sum (ops.deterministic_sums; utils.set_deterministic(True, fixed_sums=True) sets it).
DETERMINISTIC_SUMS = os.environ.get("UNAST_DETERMINISTIC_SUMS", "0") == "1"

# While a HIP graph is being captured, a backward segment drops the dependencies its stream inherited from the origin stream's relay that
# are not the producers of its own incoming gradients (engine._Segment._backward; include/unast_hip.h unast_capture_prune).  0 = round-2 form.
CAPTURE_PRUNE = os.environ.get("UNAST_CAPTURE_PRUNE", "1") != "0"

# LayerNorm backward in the epilogue of the input-gradient GEMM that produces its dy (ops.linear_dgrad_lnbwd; inside the encoder / decoder
# stacks, where a sub-layer's output has the next sub-layer as its only consumer).  0 = GEMM + stand-alone LayerNorm backward.
PANEL_LNBWD = os.environ.get("UNAST_PANEL_LNBWD", "1") != "0"

# Weight gradients of one backward closure (an attention sub-layer's out-proj + in-proj, an FFN's two linears, ...) go out as
# ONE grouped launch (csrc/gemm.hip gemm_group_kernel) instead of one split-K launch + one reduction each; 0 = one by one.
WGRAD_GROUP = os.environ.get("UNAST_WGRAD_GROUP", "1") != "0"

# Attention backward as ONE pass (dK, dV and dQ; csrc/attention.hip attn_dkv_kernel<*,1>) instead of a dQ kernel + a dK/dV kernel.
ATTN_FUSED_BWD = os.environ.get("UNAST_ATTN_FUSED_BWD", "1") != "0"
# Terms per dV / dK / dQ product of the one-pass backward: 3 = a_hi b_hi + a_hi b_lo + a_lo b_hi as everywhere else; 2 = the probabilities P
# and the score gradients dS (made by the kernel itself) enter as ONE bf16 part: 12 instead of 15 MFMAs per tile product and no low-part
# split of P / dS.  tools/oracle_emu_attn_terms.py prices it at the model level (gradient norms and the median gradient error against
# fp64 do not move: the per-element 2^-9 is zero-mean and a weight gradient sums 25 600 rows), but a single attention call's dQ / dK / dV
# move from ~2e-5 to ~2e-3 of their largest element, which tests/test_gpu_kernels.py::test_attention_fwd_bwd holds at 5e-5: opt-in.
ATTN_BWD_TERMS = int(os.environ.get("UNAST_ATTN_BWD_TERMS", "3"))

# Input-gradient GEMMs (dX = dY W) can read W^T from a transposed pre-split copy (K-contiguous operand, as the forward GEMMs read
# W) instead of the untransposed weights through transposed LDS reads.  Measured on MI355X at config 3: GEMM family 24.08 vs
# 24.46 ms/step, step 32.9 vs 33.1 ms -- within noise (these launches are bound by their C / gate streams, not by the B path), so
# it is OFF by default (it costs a second 68 MB weight copy and a refresh launch per optimizer phase); 1 = on.
DGRAD_TRANSPOSED = os.environ.get("UNAST_DGRAD_T", "0") == "1"

# The in-projections write Q / K / V -- and the out-projection's input-gradient GEMM writes dO -- in the pre-split operand format;
# the attention kernels then stage K/V (forward) and Q/dO (backward) tiles without fp32 -> hi/lo conversions.  0 = fp32 (A/B).
ATTN_PRESPLIT = os.environ.get("UNAST_ATTN_PRESPLIT", "1") != "0"

# BatchNorm batch statistics of the conv stacks taken in the conv GEMM's epilogue (unast_gemm colstats) instead of by a column-sum
# pass over the conv output.  0 = the separate pass (A/B).
CONV_BN_STATS = os.environ.get("UNAST_CONV_BN_STATS", "1") != "0"

# Row-panel GEMM (csrc/panel.hip) for K <= 256 contractions over at least PANEL_MIN_ROWS rows: the activation panel stays in registers,
# the weights stream through LDS by LDS-DMA from tiled bf16 planes kept next to the flat parameter store (engine.FlatStore).  Measured on
# MI355X against the tile GEMM at M = 25 600 (tools/bench_panel.py): linear1 51-54 vs 67-70 us, in-projection 42-45 vs 47-52, out-projection
# 17.6 vs 21, K/V projection 29 vs 36, prenet fc1 11 vs 14.7; at M = 5 760 (text side) the tile kernel is faster.  0 = tile GEMM everywhere.
PANEL_GEMM = os.environ.get("UNAST_PANEL", "1") != "0"
PANEL_MIN_ROWS = int(os.environ.get("UNAST_PANEL_MIN_ROWS", "16384"))
# 1128 = 128-row panels as 16 waves x 16 rows (4 waves / SIMD); 128 = 8 waves x 32 rows; 64 = 8 waves x 16 rows
PANEL_ROWS = int(os.environ.get("UNAST_PANEL_ROWS", "1128"))
# LayerNorm in the epilogue of the out-projection GEMMs (post-LN sub-layers): z, y, mean, rstd come out of one launch.
PANEL_LN = os.environ.get("UNAST_PANEL_LN", "1") != "0"
# K > 256 contractions into 256 columns (FFN linear2 with its LayerNorm, the input gradients of linear1 / the in-projections) on the
# K-streamed form of the panel kernel
PANEL_KSTREAM = os.environ.get("UNAST_PANEL_KSTREAM", "1") != "0"
# linear1 writes one keep bit per hidden element (relu > 0 and not dropped); linear2's input-gradient GEMM gates with those bits
# (scalar loads) instead of re-reading the 105 MB hidden activation.
PANEL_GATE_BITS = os.environ.get("UNAST_PANEL_GATE_BITS", "1") != "0"

# Data-parallel gradient exchange through the C ABI's own RCCL communicator (csrc/comm.cpp: unast_comm_init / unast_allreduce) instead of
# torch.distributed calls: stream-ordered, one ctypes call per bucket, and -- because the stream-replay executor can issue it from C++ --
# the captured train step stays usable under a process group.  0 = torch.distributed all_reduce (eager step only).
NATIVE_COMM = os.environ.get("UNAST_NATIVE_COMM", "1") != "0"

# Opt-in (default off): the backward of an ENCODER's self-attention does not visit query tiles past the sequence length.  Exact in the
# train step -- nothing downstream of an encoder reads a padded position (cross-attention and the discriminator mask by length), so dO is
# zero there -- but not for a caller who puts a loss on padded encoder outputs, which the reference would differentiate.
ENC_SKIP_PAD_GRADS = os.environ.get("UNAST_ENC_SKIP_PAD_GRADS", "0") == "1"

# Test infrastructure: > 0 = every side-stream call (forward, backward segment, weight-gradient companion) starts with a spin kernel of
# random length up to this many microseconds, which shifts the streams against each other; results must not change (tests/test_gpu_streams.py).
STREAM_JITTER = int(os.environ.get("UNAST_STREAM_JITTER", "0"))
STREAM_JITTER_SEED = int(os.environ.get("UNAST_STREAM_JITTER_SEED", "0"))

# Test / debugging aid: loss kernels keep zero-on-entry / zero-on-exit workspaces (ops.masked_mse, train._loss_ws); 1 = check a workspace is
# zero when it is handed out (one host synchronisation per call).
DEBUG_WORKSPACES = os.environ.get("UNAST_DEBUG_WORKSPACES", "0") == "1"

# The generator phase of an outer step with one auto-encoder and one supervised sub-step (ae_steps = sp_steps = 1, no cross-model sub-step,
# both batches in one shape) as ONE forward and ONE backward (train.train_gen_joint_step): each encoder's stack runs once over both
# sub-steps' batches, the frozen discriminator once over both sub-steps' encoder outputs.  0 = the two sub-steps one after the other.
JOINT_GEN = os.environ.get("UNAST_JOINT_GEN", "1") != "0"

# Heads and losses in one launch each (north_star): the text head's GEMM computes the cross-entropy and its gradient from its accumulators
# (csrc/loss.hip text_head_loss_kernel) when the step tells the decoder call what the loss will be (decode_sequence(..., loss_hint=));
# text_loss() then only hands the results over.  0 = head GEMM, loss forward and loss backward as three launches.
FUSED_HEAD_LOSS = os.environ.get("UNAST_FUSED_HEAD_LOSS", "1") != "0"

# ... and the two speech decoder calls of the joint generator step as one (SpeechTransformer.decode_pair): self-attention and feed-forward
# over both sub-steps' targets, cross-attention per call.  0 = two decode_sequence calls.
JOINT_DECODERS = os.environ.get("UNAST_JOINT_DECODERS", "1") != "0"
