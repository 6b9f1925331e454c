"""Timeline of one config-3 train step by side stream (forward segments only; backward runs inside autograd): where does the
speech stream wait?  Diagnostic: uses the optional trace hook of engine.on_stream."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collections import defaultdict
from unast_amd import train, utils, engine
from unast_amd.configs import make_args
from unast_amd.portable import synth_batch
dev = torch.device("cuda:0"); train.DEVICE = dev
args = make_args(num_layers=4, ae_steps=1, sp_steps=1, d_steps=1, cm_steps=0)
utils.set_seed(0); utils.set_deterministic(False)
_, _, model, opt, sched = train.initialize_model(args)
batch = tuple(torch.from_numpy(x).to(dev) for x in synth_batch(32, 180, 800, seed=0))
batches = dict(unsup=[batch], sup=[batch], disc=[batch], cm=[]); losses = defaultdict(list)
for i in range(int(os.environ.get("WARM", "4"))): train.train_step(losses, model, opt, sched, batches, i, args, defer_d_phase=True)
torch.cuda.synchronize()
base = torch.cuda.Event(enable_timing=True); base.record()
engine._Streams.trace = []
marks = []
def mark(tag):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((tag, e))
orig_ae, orig_sp, orig_d, orig_opt = train.train_ae_step, train.train_sp_step, train.train_discriminator_step, train.optimizer_step
def wrap(fn, tag):
    def w(*a, **k):
        mark(tag + " begin (main)"); r = fn(*a, **k); mark(tag + " end (main)"); return r
    return w
train.train_ae_step, train.train_sp_step, train.train_discriminator_step, train.optimizer_step = wrap(orig_ae, "AE"), wrap(orig_sp, "SP"), wrap(orig_d, "D"), wrap(orig_opt, "OPT")
for i in range(2): train.train_step(losses, model, opt, sched, batches, 10 + i, args, defer_d_phase=True)
end = torch.cuda.Event(enable_timing=True); end.record()
torch.cuda.synchronize()
tr = engine._Streams.trace; engine._Streams.trace = None
print("two steps: %.1f ms" % base.elapsed_time(end))
rows = [(base.elapsed_time(e0), base.elapsed_time(e1), name, call) for name, call, e0, e1 in tr] + [(base.elapsed_time(e), base.elapsed_time(e), "main", tag) for tag, e in marks]
for t0, t1, name, call in sorted(rows):
    print("%8.2f -> %8.2f  (%6.2f ms)  %-7s %s" % (t0, t1, t1 - t0, name, call))

# ---- which waits does each decorated call issue?  (second pass, host-side log)
import time
names = {}
def nm(s):
    for (d, n), st in engine._Streams.pool.items():
        if st == s: return n
    return "main" if s == torch.cuda.default_stream() else "?"
log = []
ow, oe = torch.cuda.Stream.wait_stream, torch.cuda.Stream.wait_event
def ws(self, other): log.append(("wait_stream", nm(self), nm(other))); return ow(self, other)
def we(self, ev): log.append(("wait_event", nm(self), getattr(ev, "_tag", "?"))); return oe(self, ev)
torch.cuda.Stream.wait_stream, torch.cuda.Stream.wait_event = ws, we
orr = torch.cuda.Stream.record_event
def rr(self, event=None):
    e = orr(self, event); 
    try: e._tag = "event@" + nm(self)
    except Exception: pass
    return e
torch.cuda.Stream.record_event = rr
train.train_ae_step, train.train_sp_step, train.train_discriminator_step, train.optimizer_step = orig_ae, orig_sp, orig_d, orig_opt
log.clear()
train.train_step(losses, model, opt, sched, batches, 30, args, defer_d_phase=True)
log.append(("---- next step", "", ""))
train.train_step(losses, model, opt, sched, batches, 31, args, defer_d_phase=True)
torch.cuda.synchronize()
seen = 0
for kind, a, b in log:
    if kind.startswith("----"): seen = 1; print(kind); continue
    if seen and a in ("text", "main"): print("   ", kind, a, "<-", b)
