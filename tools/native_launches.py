"""Where the torch-native kernel launches of one train step come from: the step run eagerly under torch.profiler (CPU activity, Python
stacks), aten ops that launch a kernel grouped by the innermost unast_amd / bench frame.  usage: python tools/native_launches.py [c3|small]"""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import train, utils
from unast_amd.configs import make_args
from unast_amd.engine import join_streams
from unast_amd.portable import synth_batch
D = torch.device("cuda:0"); train.DEVICE = D
small = len(sys.argv) > 1 and sys.argv[1] == "small"
B, Tt, Tm, L = (4, 28, 96, 2) if small else (32, 180, 800, 4)
utils.set_seed(0)
args = make_args(num_layers=L, ae_steps=1, sp_steps=1, d_steps=1, cm_steps=0, train_batch_size=B)
_, _, model, opt, sched = train.initialize_model(args)
mk = lambda s: tuple(torch.from_numpy(x).to(D) for x in synth_batch(B, Tt, Tm, seed=s, ragged=True))
batches = dict(unsup=[mk(1)], sup=[mk(2)], disc=[mk(3)], cm=[])
losses = collections.defaultdict(list)
train.SYNC_LOSSES = False
for i in range(3):
    train.train_step(losses, model, opt, None, batches, i, args, defer_d_phase=True)
join_streams(); torch.cuda.synchronize()
LAUNCHING = ("aten::zero_", "aten::fill_", "aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::mul_", "aten::div", "aten::div_", "aten::sub", "aten::neg",
             "aten::sum", "aten::mean", "aten::stack", "aten::cat", "aten::clone", "aten::index", "aten::index_put_", "aten::masked_fill_", "aten::where", "aten::sqrt",
             "aten::_to_copy", "aten::eq", "aten::ne", "aten::lt", "aten::ge", "aten::gt", "aten::le", "aten::any", "aten::all", "aten::arange", "aten::sigmoid", "aten::exp",
             "aten::_foreach_add_", "aten::addcmul_", "aten::clamp", "aten::clamp_", "aten::reciprocal", "aten::rsqrt", "aten::pow", "aten::abs", "aten::max", "aten::min")
from torch.profiler import profile, ProfilerActivity
try:
    cfg = torch._C._profiler._ExperimentalConfig(verbose=True)
except Exception:
    cfg = None
with profile(activities=[ProfilerActivity.CPU], with_stack=True, experimental_config=cfg) as prof:
    train.train_step(losses, model, opt, None, batches, 3, args, defer_d_phase=True)
    join_streams(); torch.cuda.synchronize()
agg = collections.Counter()
evs = prof.events()
print("events %d, with a stack %d" % (len(evs), sum(1 for e in evs if e.stack)))
# only top-level launching ops: an op nested inside another launching op (zero_ -> fill_, clone -> copy_) is the same launch
spans = sorted(((e.time_range.start, e.time_range.end, e.thread, e) for e in evs if e.name in LAUNCHING), key=lambda t: (t[2], t[0], -t[1]))
last_end = {}
for st, en, th, e in spans:
    if th in last_end and st < last_end[th]:
        continue
    last_end[th] = en
    where = "?"
    for fr in (e.stack or []):
        if ("unast_amd" in fr or "bench.py" in fr) and "_lib.py" not in fr:
            where = fr.split("/root/repo/")[-1] if "/root/repo/" in fr else fr
            break
    shp = ""
    agg[(e.name, where)] += 1
print("top-level launching aten ops in one eager step: %d" % sum(agg.values()))
for (name, where), c in sorted(agg.items(), key=lambda kv: -kv[1]):
    print("%4d  %-18s %s" % (c, name, where))
