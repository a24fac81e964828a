"""Data side of the hot path: collation of ragged items and the dataset
protocol the samplers / trainer rely on.

Reference: brever/data.py:389-491 (``BreverDataLoader``) and the attribute /
method set of ``BreverDataset`` that ``BreverTrainer`` and the samplers touch
(brever/data.py:225-326; the minimal spec is the reference's own
tests/utils.py:9-42 ``DummyDataset``). FLAC-in-tar reading and segmentation are
out of scope for this round: the benchmark and tests run
on synthetic mixtures generated in memory by ``SyntheticMixtureDataset``.
"""
import random

import torch
import torch.nn.functional as F


class BreverDataLoader(torch.utils.data.DataLoader):
    """DataLoader that right-zero-pads ragged items to the batch maximum."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.collate_fn = self._collate_fn

    def set_epoch(self, epoch):
        self.batch_sampler.set_epoch(epoch)
        dataset = self.dataset
        if isinstance(dataset, torch.utils.data.Subset):
            dataset = dataset.dataset
        dataset.set_epoch(epoch)

    @staticmethod
    def _collate_fn(unbatched):
        """Collate a list of items (tensors, or tuples of tensors).

        Each model input is padded with zeros on the right of its last
        dimension up to the longest example in the batch, then stacked. Returns
        ``(batched, lengths)`` where ``lengths`` holds the original last-dim
        sizes, shape ``(batch,)`` for tensor items and ``(batch, n_inputs)``
        for tuple items, on the device of the first item
        (brever/data.py:407-491).
        """
        single = isinstance(unbatched[0], torch.Tensor)
        rows = [(item,) if single else tuple(item) for item in unbatched]
        lengths = torch.tensor(
            [[x.shape[-1] for x in row] for row in rows],
            device=rows[0][0].device,
        )
        longest = lengths.amax(dim=0).tolist()
        batched = []
        for column, target in zip(zip(*rows), longest):
            batched.append(torch.stack(
                [F.pad(x, (0, target - x.shape[-1])) for x in column]
            ))
        if single:
            return batched[0], lengths.squeeze(-1)
        return batched, lengths


class SyntheticMixtureDataset(torch.utils.data.Dataset):
    """In-memory synthetic noisy/clean pairs implementing the dataset protocol.

    Item ``i`` is a float32 tensor ``(2 sources [mixture, foreground],
    2 channels [left = right], length_i)`` (the layout returned by
    ``BreverDataset.__getitem__``, brever/data.py:244-257) built as in
    SURVEY.md §8(d): clean ``0.1*randn`` (seed ``1234+i``), noise
    ``0.1*randn`` (seed ``5678+i``) scaled to an SNR drawn from U(-5, 10) dB
    with ``random.Random(0)``.
    """

    def __init__(self, n_items, length, fs=16000, min_length=None,
                 transform=None, seed=0):
        self.fs = fs
        self.transform = transform
        self.preloaded_data = None
        self.segment_strategy = 'pass'
        self.rmm_dset = None
        rng = random.Random(seed)
        self._snrs = [rng.uniform(-5.0, 10.0) for _ in range(n_items)]
        if min_length is None:
            self._lengths = [int(length)]*n_items
        else:
            lrng = random.Random(seed + 1)
            self._lengths = [lrng.randint(int(min_length), int(length))
                             for _ in range(n_items)]
        self._segment_info = [(i, (0, n)) for i, n in enumerate(self._lengths)]
        self._duration = sum(self._lengths)/fs
        self._effective_duration = self._duration

    def make_item(self, i):
        n = self._lengths[i]
        g = torch.Generator().manual_seed(1234 + i)
        clean = 0.1*torch.randn(n, generator=g)
        g = torch.Generator().manual_seed(5678 + i)
        noise = 0.1*torch.randn(n, generator=g)
        gain = 10.0**(-self._snrs[i]/20.0)*clean.norm()/noise.norm()
        mixture = clean + gain*noise
        item = torch.stack([mixture, clean])            # (sources, n)
        return item.unsqueeze(1).repeat(1, 2, 1)        # (sources, 2, n)

    def __len__(self):
        return len(self._lengths)

    def __getitem__(self, index):
        if self.preloaded_data is not None:
            return self.preloaded_data[index]
        item = self.make_item(index)
        if self.transform is not None:
            item = self.transform(item)
        return item

    def get_segment_length(self, i):
        return self._lengths[i]

    def get_max_segment_length(self):
        return max(self._lengths)

    def preload(self, device, tqdm_desc=None):
        data = []
        for i in range(len(self)):
            item = self[i]
            if isinstance(item, torch.Tensor):
                item = item.to(device)
            else:
                item = [x.to(device) for x in item]
            data.append(item)
        self.preloaded_data = data

    def set_epoch(self, epoch):
        pass
