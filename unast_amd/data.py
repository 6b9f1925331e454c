"""The data contract either side of the train step: what a batch looks like when it reaches
process_batch, for real pre-computed features on disk as well as for synthetic ones.

  * `collate_fn_transformer` -- the reference's collate (src/preprocess.py:82-118): samples ordered by text length, longest
    first (a stable sort: equal lengths keep their order), text and mel zero-padded to the longest of the batch, returned as
    (text int64 [B,Tt], mel float32 [B,Tm,M], text_length int64 [B], mel_length int64 [B]); with file names in the samples,
    ((...), fnames).  Everything downstream relies on this layout: the packed LSTM wants lengths sorted, the padded tail must
    be zero (BatchNorm statistics and the stop-token loss see it).
  * `NpyFeatureDataset` -- the reference's LJDatasets (src/preprocess.py:14-52) for features that are already on disk: one
    `<root>/<id>.pt.npy` mel [T,80] per utterance listed in a `id|text|...` metadata file.  Phoneme ids come from
    `<root>/<id>.ids.npy` when present, else from the `text_to_ids` callable (the reference's g2p front end -- cmudict, number and
    abbreviation expansion, src/data/ -- is data preparation and stays outside this package).
  * `BatchGetter` -- src/train.py:32-78: three endlessly cycling shuffled loaders (supervised / unsupervised / discriminator).

The loaders hand over host tensors (pinned when a GPU is present); process_batch moves them to the device."""
import os

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset


def _pad_rows(arrays, dtype):
    """Stack arrays that differ in their first dimension, zero-padding it to the longest."""
    n = max(a.shape[0] for a in arrays)
    out = np.zeros((len(arrays), n) + tuple(arrays[0].shape[1:]), dtype=dtype)
    for i, a in enumerate(arrays):
        out[i, :a.shape[0]] = a
    return out


def collate_fn_transformer(batch):
    """src/preprocess.py:82-118."""
    if not (len(batch) > 0 and isinstance(batch[0], dict)):
        raise TypeError("batch must contain dicts; found {}".format(type(batch[0]) if len(batch) else "an empty batch"))
    order = sorted(range(len(batch)), key=lambda i: -int(batch[i]["text_length"]))       # stable: ties keep their order
    text = _pad_rows([np.asarray(batch[i]["text"]) for i in order], np.int64)
    mel = _pad_rows([np.asarray(batch[i]["mel"], dtype=np.float32) for i in order], np.float32)
    out = (torch.from_numpy(text), torch.from_numpy(mel),
           torch.tensor([int(batch[i]["text_length"]) for i in order], dtype=torch.long),
           torch.tensor([int(batch[i]["mel_length"]) for i in order], dtype=torch.long))
    if "fname" in batch[0]:
        return out, [batch[i]["fname"] for i in order]
    return out


class NpyFeatureDataset(Dataset):
    """Utterances listed in a `|`-separated metadata file (first column = id, second = transcript), features under root_dir."""

    def __init__(self, csv_file, root_dir, ret_file_names=False, text_to_ids=None):
        with open(csv_file, encoding="utf-8") as f:
            self.rows = [line.rstrip("\n").split("|") for line in f if line.strip()]
        self.root_dir, self.ret_file_names, self.text_to_ids = root_dir, ret_file_names, text_to_ids

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, idx):
        uid = self.rows[idx][0]
        base = os.path.join(self.root_dir, uid)
        if os.path.exists(base + ".ids.npy"):
            text = np.load(base + ".ids.npy").astype(np.int32)
        elif self.text_to_ids is not None:
            text = np.asarray(self.text_to_ids(self.rows[idx][1]), dtype=np.int32)
        else:
            raise FileNotFoundError("%s.ids.npy is missing and no text_to_ids callable was given" % base)
        mel = np.load(base + ".pt.npy")
        sample = {"text": text, "mel": mel, "text_length": len(text), "mel_length": mel.shape[0]}
        if self.ret_file_names:
            sample["fname"] = uid
        return sample


class BatchGetter:
    """src/train.py:32-78."""

    def __init__(self, args, supervised_dataset, unsupervised_dataset, full_dataset):
        kw = dict(batch_size=args.train_batch_size, shuffle=True, collate_fn=collate_fn_transformer, drop_last=True,
                  num_workers=getattr(args, "num_workers", 0), pin_memory=torch.cuda.is_available())
        self._loaders = {"supervised": DataLoader(supervised_dataset, **kw), "unsupervised": DataLoader(unsupervised_dataset, **kw)}
        if args.use_discriminator:
            self._loaders["discriminator"] = DataLoader(full_dataset, **kw)
        self._iters = {k: iter(v) for k, v in self._loaders.items()}

    def _next(self, which):
        try:
            return next(self._iters[which])
        except StopIteration:                                   # an epoch of this loader is over: start the next one
            self._iters[which] = iter(self._loaders[which])
            return next(self._iters[which])

    def get_supervised_batch(self):
        return self._next("supervised")

    def get_unsupervised_batch(self):
        return self._next("unsupervised")

    def get_discriminator_batch(self):
        return self._next("discriminator")
