"""Which parts of the train step survive HIP-graph capture?  Each case runs in its own process (a crash stays contained).
Usage: python tools/debug_capture.py            # runs every case
       python tools/debug_capture.py CASE       # runs one case in this process"""
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES2 = ["ae_step_ms", "ae_nowgradstream_ms", "ae_nolnoffload_ms", "ae_nodisc_ms", "textonly_ms", "speechonly_ms", "speechonly_nowgradstream_ms",
          "ae_keepevents_ms", "ae_relaxed_ms", "ae_threadlocal_ms", "speechfwd_ms", "textonly_nobwd_ms"]
CASES3 = ["discfwd_ms", "discdetach_ms", "aedisc_singlethread_ms", "aedisc_waitstream_ms", "aedisc_grouptext_ms", "aedisc_nofreeze_ms", "aedisc_norecord_ms",
          "aedisc_sync_producer_ms"]
CASES = ["fwd_text_1s", "fwd_speech_1s", "fwdbwd_text_1s", "ae_step_1s", "ae_step_ms", "sp_step_ms", "gen_phase_1s", "gen_phase_ms", "d_phase_1s", "d_phase_ms",
         "opt_only", "body_1s", "body_ms", "memset_only", "losses_only"]


def run_case(name):
    import torch
    from unast_amd import config, ops, train, utils
    from unast_amd.configs import make_args
    from unast_amd.engine import join_streams
    from unast_amd.inference import _capture
    from unast_amd.portable import synth_batch
    dev = torch.device("cuda:0")
    train.DEVICE = dev
    config.SIDE_STREAMS = name.endswith("_ms")
    if "nowgradstream" in name:
        config.WGRAD_STREAMS = False
    if "nolnoffload" in name:
        config.LN_FINALIZE_OFFLOAD = False
    keep = []
    if "grouptext" in name:
        config.STREAM_GROUPS = {"disc": "text"}
    if "singlethread" in name:
        torch.autograd.set_multithreading_enabled(False)
    if "norecord" in name:
        torch.Tensor.record_stream = lambda self, s: None
    if "waitstream" in name or "sync_producer" in name:
        from unast_amd import engine
        _we = torch.cuda.Stream.wait_event
        def we(self, ev):
            # coarser: wait for the whole producing stream(s) instead of one recorded event
            for (d, nm), st in engine._Streams.pool.items():
                if st != self:
                    self.wait_stream(st)
        torch.cuda.Stream.wait_event = we
    if "keepevents" in name:                      # never destroy an event while the capture is open
        _rec = torch.cuda.Stream.record_event
        def rec(self, event=None):
            e = _rec(self, event)
            keep.append(e)
            return e
        torch.cuda.Stream.record_event = rec
    args = make_args(num_layers=2, ae_steps=1, sp_steps=1, d_steps=1, cm_steps=0, lr=1e-4)
    utils.set_seed(0)
    utils.set_deterministic(False)
    _, _, model, opt, _ = train.initialize_model(args)
    batch = tuple(torch.from_numpy(x).to(dev) for x in synth_batch(4, 28, 96, seed=1, ragged=True))
    ops.step_state()
    losses = defaultdict(list)
    model.train()

    def gen():
        train.freeze_model_parameters(model.discriminator)
        train.train_ae_step(losses, model, batch, 0, 2, args)
        train.train_sp_step(losses, model, batch, 0, 2, args)
        train.optimizer_step(model, opt, args)

    def dph(defer):
        train.unfreeze_model_parameters(model.discriminator)
        train.train_discriminator_step(losses, model, batch, 0, 1, args, defer=defer)
        train.optimizer_step(model, opt, args, defer=defer)

    def fn():
        (text, mel, tl, ml), _ = train.process_batch(batch)
        if name == "fwd_text_1s":
            with torch.no_grad():
                model.text_ae(text, tl)
        elif name == "fwd_speech_1s":
            with torch.no_grad():
                model.speech_ae(mel, ml)
        elif name == "fwdbwd_text_1s":
            out = model.text_ae(text, tl)
            out.sum().backward()
        elif name.startswith("discfwd") or name.startswith("discdetach"):
            from unast_amd.engine import side_streams
            train.freeze_model_parameters(model.discriminator)
            with side_streams():
                if name.startswith("discfwd"):
                    with torch.no_grad():
                        tp, th = model.text_ae(text, tl, ret_enc_hid=True)
                        pre, post, stop, sh = model.speech_ae(mel, ml, ret_enc_hid=True)
                        db = train.discriminator_shuffle_batch(th, tl, sh, ml, "transformer")
                        dl, _ = train.discriminator_hidden_to_loss(model, db, freeze_discriminator=True)
                    join_streams()
                else:
                    tp, th = model.text_ae(text, tl, ret_enc_hid=True)
                    pre, post, stop, sh = model.speech_ae(mel, ml, ret_enc_hid=True)
                    db = train.discriminator_shuffle_batch(th.detach().requires_grad_(True), tl, sh.detach().requires_grad_(True), ml, "transformer")
                    dl, _ = train.discriminator_hidden_to_loss(model, db, freeze_discriminator=True)
                    sl = train.speech_loss(mel, None, pre, post, ml, stop, 5.0)
                    tl_ = train.text_loss(text, tp.permute(0, 2, 1), 1.0)
                    join_streams()
                    ((dl + sl + tl_) / 2).backward()
        elif name.startswith("aedisc_"):
            if "nofreeze" in name:
                train.unfreeze_model_parameters(model.discriminator)
            else:
                train.freeze_model_parameters(model.discriminator)
            train.train_ae_step(losses, model, batch, 0, 2, args)
        elif name.startswith("ae_nodisc"):
            args.use_discriminator = False
            train.train_ae_step(losses, model, batch, 0, 2, args)
        elif name.startswith("textonly") or name.startswith("speechonly") or name.startswith("speechfwd"):
            from unast_amd.engine import side_streams
            with side_streams():
                if name.startswith("textonly"):
                    out = model.text_ae(text, tl)
                    l = train.text_loss(text, out.permute(0, 2, 1), 1.0)
                elif name.startswith("speechfwd"):
                    with torch.no_grad():
                        model.speech_ae(mel, ml)
                    l = None
                else:
                    pre, post, stop = model.speech_ae(mel, ml)
                    l = train.speech_loss(mel, None, pre, post, ml, stop, 5.0)
                join_streams()
                if l is not None and "nobwd" not in name:
                    (l / 2).backward()
        elif name.startswith("ae_"):
            train.freeze_model_parameters(model.discriminator)
            train.train_ae_step(losses, model, batch, 0, 2, args)
        elif name.startswith("sp_step"):
            train.freeze_model_parameters(model.discriminator)
            train.train_sp_step(losses, model, batch, 0, 2, args)
        elif name.startswith("gen_phase"):
            gen()
        elif name.startswith("d_phase"):
            dph(False)
        elif name == "opt_only":
            model._store().touched.add("gen")
            train.optimizer_step(model, opt, args)
        elif name.startswith("body"):
            dph(name.endswith("_ms"))
            gen()
        elif name == "memset_only":
            x = torch.empty(4, 300, 80, device=dev)
            ops.specaugment(mel, utils.lens_i32(ml, dev), torch.empty_like(mel), 1, 1)
        elif name == "losses_only":
            lg = torch.randn(4, 46, 28, device=dev, requires_grad=True)
            l = train.text_loss(text, lg, 1.0)
            l.backward()
        join_streams()
    # warm-up eagerly twice (lazy initialisations, streams), then capture, then replay twice
    fn(); fn()
    join_streams(); torch.cuda.synchronize()
    if "relaxed" in name or "threadlocal" in name:
        mode = "relaxed" if "relaxed" in name else "thread_local"
        cs = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        cs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cs):
            g.capture_begin(capture_error_mode=mode)
            fn()
            g.capture_end()
        torch.cuda.current_stream().wait_stream(cs)
    else:
        g = _capture(fn)
    torch.cuda.synchronize()
    ops.set_step_state(1, {0: [1e-4, 0.1, 0.03], 1: [1e-4, 0.1, 0.03]})
    g.replay(); g.replay()
    torch.cuda.synchronize()
    print("CASE", name, "OK")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run_case(sys.argv[1])
    else:
        for c in {"1": CASES, "2": CASES2, "3": CASES3}[os.environ.get("DBG_SET", "1")]:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), c], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            tail = (r.stdout.decode().strip().splitlines() or [""])[-1]
            err = [l for l in r.stderr.decode().splitlines() if ("Error" in l or "error" in l or "Fatal" in l or "File \"/" in l)][-6:]
            print("%-16s rc=%4d %s %s" % (c, r.returncode, tail, " | ".join(e.strip() for e in err)), flush=True)
