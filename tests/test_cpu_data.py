"""The batch contract (src/preprocess.py:82-118 collate, src/train.py:32-78 BatchGetter) on features read from disk."""
import types

import numpy as np
import pytest
import torch


def _write_corpus(root, lens):
    rng = np.random.default_rng(0)
    rows = []
    for i, (tl, ml) in enumerate(lens):
        uid = "LJ%03d" % i
        ids = rng.integers(3, 46, size=tl).astype(np.int32)
        ids[-1] = 2
        np.save(root / (uid + ".ids.npy"), ids)
        np.save(root / (uid + ".pt.npy"), rng.random((ml, 80), dtype=np.float32))
        rows.append("%s|some text %d|some text %d" % (uid, i, i))
    (root / "metadata.csv").write_text("\n".join(rows) + "\n")
    return str(root / "metadata.csv")


def test_collate_sorts_by_text_length_and_zero_pads(tmp_path):
    from unast_amd.data import NpyFeatureDataset, collate_fn_transformer
    lens = [(5, 30), (9, 12), (5, 40), (7, 7)]
    ds = NpyFeatureDataset(_write_corpus(tmp_path, lens), str(tmp_path), ret_file_names=True)
    assert len(ds) == 4
    (text, mel, tl, ml), names = collate_fn_transformer([ds[i] for i in range(4)])
    assert names == ["LJ001", "LJ003", "LJ000", "LJ002"], "longest text first, ties in their original order"
    assert text.dtype == torch.int64 and mel.dtype == torch.float32 and tl.dtype == torch.int64 and ml.dtype == torch.int64
    assert text.shape == (4, 9) and mel.shape == (4, 40, 80)
    assert tl.tolist() == [9, 7, 5, 5] and ml.tolist() == [12, 7, 30, 40]
    for row, uid in enumerate(names):
        ids = np.load(tmp_path / (uid + ".ids.npy"))
        m = np.load(tmp_path / (uid + ".pt.npy"))
        assert np.array_equal(text[row, :len(ids)].numpy(), ids) and (text[row, len(ids):] == 0).all()
        assert np.array_equal(mel[row, :len(m)].numpy(), m) and (mel[row, len(m):] == 0).all()
    plain = collate_fn_transformer([{k: v for k, v in ds[i].items() if k != "fname"} for i in range(4)])
    assert len(plain) == 4 and torch.equal(plain[0], text)
    with pytest.raises(TypeError):
        collate_fn_transformer([np.zeros(3)])


def test_text_front_end_callable_and_missing_ids(tmp_path):
    from unast_amd.data import NpyFeatureDataset
    csv = _write_corpus(tmp_path, [(4, 6)])
    (tmp_path / "LJ000.ids.npy").unlink()
    with pytest.raises(FileNotFoundError):
        NpyFeatureDataset(csv, str(tmp_path))[0]
    s = NpyFeatureDataset(csv, str(tmp_path), text_to_ids=lambda t: [3 + (ord(c) % 40) for c in t] + [2])[0]
    assert s["text_length"] == len("some text 0") + 1 and s["text"][-1] == 2 and s["mel_length"] == 6


def test_batch_getter_cycles_and_feeds_process_batch(tmp_path):
    from unast_amd import train
    from unast_amd.data import NpyFeatureDataset, BatchGetter
    ds = NpyFeatureDataset(_write_corpus(tmp_path, [(5 + i, 20 + 3 * i) for i in range(5)]), str(tmp_path))
    args = types.SimpleNamespace(train_batch_size=2, num_workers=0, use_discriminator=True)
    bg = BatchGetter(args, ds, ds, ds)
    seen = 0
    for _ in range(7):                                          # 5 utterances, batch 2, drop_last: two batches per epoch
        for get in (bg.get_supervised_batch, bg.get_unsupervised_batch, bg.get_discriminator_batch):
            text, mel, tl, ml = get()
            assert text.shape[0] == 2 and tl[0] >= tl[1] and mel.shape[1] == int(ml.max()) and text.shape[1] == int(tl.max())
            seen += 1
    assert seen == 21
    train.DEVICE = torch.device("cpu")
    (text, mel, tl, ml), (gold_char, gold_mel, gold_stop) = train.process_batch(bg.get_supervised_batch())
    assert gold_mel.shape == mel.shape and gold_char.shape == text.shape
