"""Stream-topology probes for HIP-graph capture on this ROCm: which wait patterns between side streams survive
hipStreamEndCapture?  Each case in its own process."""
import os
import subprocess
import sys

CASES = ["T1_cycle", "T2_fresh_stream", "T3_via_origin", "T5_cycle_prealloc_events", "T6_cycle_three", "T7_autograd_two_streams", "T8_autograd_cycle",
         "T9_one_way", "T10_cycle_waitstream"]


def run(name):
    import torch
    dev = torch.device("cuda:0")
    x = torch.zeros(1 << 20, device=dev)
    A, B, C, A2 = (torch.cuda.Stream() for _ in range(4))
    pre = [torch.cuda.Event() for _ in range(8)]

    def k(t):
        t.add_(1.0)

    def body():
        O = torch.cuda.current_stream()
        if name in ("T1_cycle", "T5_cycle_prealloc_events", "T10_cycle_waitstream"):
            A.wait_stream(O); B.wait_stream(O)
            with torch.cuda.stream(A):
                k(x)
                e1 = pre[0] if "prealloc" in name else torch.cuda.Event()
                e1.record(A)
            if name == "T10_cycle_waitstream":
                B.wait_stream(A)
            else:
                B.wait_event(e1)
            with torch.cuda.stream(B):
                k(x)
                e2 = pre[1] if "prealloc" in name else torch.cuda.Event()
                e2.record(B)
            if name == "T10_cycle_waitstream":
                A.wait_stream(B)
            else:
                A.wait_event(e2)
            with torch.cuda.stream(A):
                k(x)
            O.wait_stream(A); O.wait_stream(B)
        elif name == "T9_one_way":
            A.wait_stream(O); B.wait_stream(O)
            with torch.cuda.stream(A):
                k(x); e1 = torch.cuda.Event(); e1.record(A)
            B.wait_event(e1)
            with torch.cuda.stream(B):
                k(x)
            with torch.cuda.stream(A):
                k(x)
            O.wait_stream(A); O.wait_stream(B)
        elif name == "T2_fresh_stream":
            A.wait_stream(O); B.wait_stream(O)
            with torch.cuda.stream(A):
                k(x); e1 = torch.cuda.Event(); e1.record(A)
            B.wait_event(e1)
            with torch.cuda.stream(B):
                k(x); e2 = torch.cuda.Event(); e2.record(B)
            A2.wait_stream(O); A2.wait_event(e2); A2.wait_event(e1)
            with torch.cuda.stream(A2):
                k(x)
            O.wait_stream(A); O.wait_stream(B); O.wait_stream(A2)
        elif name == "T3_via_origin":
            A.wait_stream(O); B.wait_stream(O)
            with torch.cuda.stream(A):
                k(x)
            O.wait_stream(A); B.wait_stream(O)
            with torch.cuda.stream(B):
                k(x)
            O.wait_stream(B); A.wait_stream(O)
            with torch.cuda.stream(A):
                k(x)
            O.wait_stream(A); O.wait_stream(B)
        elif name == "T6_cycle_three":
            for s in (A, B, C):
                s.wait_stream(O)
            with torch.cuda.stream(A):
                k(x); e1 = torch.cuda.Event(); e1.record(A)
            with torch.cuda.stream(B):
                y = torch.ones(8, device=dev); e2 = torch.cuda.Event(); e2.record(B)
            C.wait_event(e1); C.wait_event(e2)
            with torch.cuda.stream(C):
                k(x); e3 = torch.cuda.Event(); e3.record(C)
            A.wait_event(e3); B.wait_event(e3)
            with torch.cuda.stream(A):
                k(x)
            with torch.cuda.stream(B):
                y.add_(1)
            for s in (A, B, C):
                O.wait_stream(s)
        elif name in ("T7_autograd_two_streams", "T8_autograd_cycle"):
            w = torch.ones(1024, device=dev, requires_grad=True)
            A.wait_stream(O)
            with torch.cuda.stream(A):
                y = (w * 2).tanh()
            if name == "T8_autograd_cycle":
                B.wait_stream(O); B.wait_stream(A)
                with torch.cuda.stream(B):
                    z = (y * 3).sin()
                A.wait_stream(B)
                with torch.cuda.stream(A):
                    u = (z * y).sum()
                O.wait_stream(A); O.wait_stream(B)
                u.backward()
            else:
                O.wait_stream(A)
                z = (y * 3).sum()
                z.backward()
            O.wait_stream(A); O.wait_stream(B)

    body(); torch.cuda.synchronize()
    cs = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    cs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cs):
        g.capture_begin()
        body()
        g.capture_end()
    torch.cuda.current_stream().wait_stream(cs)
    g.replay(); torch.cuda.synchronize()
    print("CASE", name, "OK", float(x[0]))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for c in CASES:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), c], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            tail = (r.stdout.decode().strip().splitlines() or [""])[-1]
            err = [l for l in r.stderr.decode().splitlines() if ("Error" in l or "error" in l)][-3:]
            print("%-28s rc=%4d %s %s" % (c, r.returncode, tail, " | ".join(e.strip() for e in err)), flush=True)
