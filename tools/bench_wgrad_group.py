"""Grouped weight-gradient launch vs one split-K launch per gradient, config-3 shapes (tokens 25600)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd import ops, config
D = torch.device("cuda:0")
groups = {"attn (Wo + Wqkv)": [(25600, 256, 256), (25600, 768, 256)], "ffn (W2 + W1)": [(25600, 256, 1024), (25600, 1024, 256)],
          "cross (Wo + Wq + Wkv@5760)": [(25600, 256, 256), (25600, 256, 256), (5760, 512, 256)],
          "layer (all four)": [(25600, 256, 1024), (25600, 1024, 256), (25600, 256, 256), (25600, 768, 256)]}
for tgt in (256, 384, 512, 768):
    ops.WGRAD_GROUP_TARGET = tgt
    for name, probs in groups.items():
        data = [(torch.randn(M, N, device=D), torch.randn(M, K, device=D), torch.zeros(N, K, device=D), torch.zeros(N, device=D)) for M, N, K in probs]
        res = {}
        for grouped in (True, False):
            config.WGRAD_GROUP = grouped
            def run():
                with ops.wgrad_batch():
                    for dy, x, dW, db in data:
                        ops.linear_wgrad(dy, x, dW, db=db)
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                run()
            e1.record(); torch.cuda.synchronize()
            res[grouped] = e0.elapsed_time(e1) / 50 * 1e3
        print("target %4d  %-28s grouped %7.1f us   one-by-one %7.1f us" % (tgt, name, res[True], res[False]), flush=True)
