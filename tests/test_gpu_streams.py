"""Side-stream bookkeeping of unast_amd.engine (on_stream / side_streams / join_streams)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
D = torch.device("cuda:0")


def test_outputs_are_tagged_but_handed_back_arguments_are_not():
    """A call's products carry (stream, event) so that consumers wait for that call only; an argument the call hands back
    (decode_sequence returns tgt_lens) must keep its own producer, or every later user of the batch's lengths waits for it."""
    from unast_amd import engine

    @engine.on_stream("speech")
    def produce(x, lens):
        return x * 2.0, lens                      # second result is the argument itself

    @engine.on_stream("disc")
    def consume(y, lens):
        return y.sum() + lens.sum()

    x = torch.ones(1024, device=D)
    lens = torch.arange(4, device=D)
    with engine.side_streams():
        y, l2 = produce(x, lens)
        reg = engine._Streams.producer
        assert reg[y.untyped_storage().data_ptr()][0] == "speech"
        assert lens.untyped_storage().data_ptr() not in reg, "a handed-back argument was re-tagged as a product"
        out = consume(y, l2)
        engine.join_streams()
        assert float(out) == 2048.0 + 6.0
    assert not engine._Streams.enabled and not engine._Streams.producer


def test_calls_outside_the_context_stay_on_the_callers_stream():
    from unast_amd import engine
    seen = []

    @engine.on_stream("text")
    def f(x):
        seen.append(torch.cuda.current_stream())
        return x + 1

    x = torch.zeros(8, device=D)
    f(x)
    assert seen[-1] == torch.cuda.current_stream()
    with engine.side_streams():
        f(x)
        assert seen[-1] != torch.cuda.current_stream()
        f2 = engine.on_stream("text")(lambda t: (seen.append(torch.cuda.current_stream()), t)[1])
        f2(x)
        assert seen[-1] == seen[-2]               # same logical stream -> same real stream


def test_weight_gradient_stream_choice_is_sticky_within_a_phase():
    """The first weight-gradient launch into a buffer decides whether it goes to the companion stream; later launches into the
    same buffer follow even if their token count is on the other side of the gate (their accumulations must stay ordered)."""
    from unast_amd import engine, ops, config
    g = torch.Generator().manual_seed(0)
    dW = torch.zeros(256, 256, device=D)
    big = (torch.randn(config.WGRAD_STREAM_MIN_TOKENS, 256, generator=g).to(D), torch.randn(config.WGRAD_STREAM_MIN_TOKENS, 256, generator=g).to(D))
    small = (torch.randn(512, 256, generator=g).to(D), torch.randn(512, 256, generator=g).to(D))
    ops.reset_wgrad_choices()

    @engine.on_stream("speech")
    def run():
        ops.linear_wgrad(big[0], big[1], dW)
        first = ops._WGRAD_CHOICE[dW.data_ptr()]
        ops.linear_wgrad(small[0], small[1], dW)
        return first, ops._WGRAD_CHOICE[dW.data_ptr()]

    with engine.side_streams():
        first, second = run()
        engine.join_streams()
    assert first is True and second is True
    ref = big[0].double().cpu().t() @ big[1].double().cpu() + small[0].double().cpu().t() @ small[1].double().cpu()
    torch.cuda.synchronize()
    assert float((dW.double().cpu() - ref).abs().max() / ref.abs().max()) < 3e-5
    ops.reset_wgrad_choices()
    assert not ops._WGRAD_CHOICE
