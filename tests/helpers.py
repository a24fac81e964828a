"""Test doubles shared by the CPU and GPU suites (mirrors the seam the reference
tests use: tests/utils.py:9-86 -- an in-memory dataset and a 1x1-conv model)."""
import random

import torch

from brever_amd.models.base import BreverBaseModel
from oracle import criterion as ref_criterion


class DummyDataset(torch.utils.data.Dataset):
    """Seeded random-length items; same construction recipe (hence the same
    lengths and samples) as the reference's test dataset."""

    def __init__(self, n_examples, n_sources, n_channels, min_length, max_length,
                 transform=None):
        rng = random.Random(42)
        self._segment_info = [(i, (0, rng.randint(min_length, max_length)))
                              for i in range(n_examples)]
        g = torch.Generator().manual_seed(42)
        self.sources = [
            torch.randn((n_sources, n_channels, self._segment_info[i][1][1]),
                        generator=g)
            for i in range(n_examples)
        ]
        self.n_examples = n_examples
        self.transform = transform
        self.preloaded_data = None
        self._duration = sum(x[1][1] for x in self._segment_info)/16000
        self._effective_duration = self._duration
        self.segment_strategy = 'pass'
        self.rmm_dset = None

    def __getitem__(self, index):
        if self.preloaded_data is not None:
            return self.preloaded_data[index]
        item = self.sources[index]
        if self.transform is not None:
            item = self.transform(item)
        return item

    def __len__(self):
        return self.n_examples

    def get_segment_length(self, i):
        return self._segment_info[i][1][1]

    def get_max_segment_length(self):
        return max(end - start for _, (start, end) in self._segment_info)

    def preload(self, device, tqdm_desc=None):
        self.preloaded_data = [self[i].to(device) for i in range(len(self))]

    def set_epoch(self, epoch):
        pass


class DummyModel(BreverBaseModel):
    """1x1 conv over channels, oracle SNR criterion (CPU-capable)."""

    def __init__(self, channels=2, output_sources=1, criterion='snr'):
        super().__init__(criterion=ref_criterion.CRITERIA[criterion])
        self.conv = torch.nn.Conv1d(channels, channels*output_sources, 1)
        self.output_sources = output_sources
        self.channels = channels
        self.optimizer = torch.optim.Adam(self.parameters())

    def forward(self, x):
        x = self.conv(x)
        return x.reshape(x.shape[0], self.output_sources, self.channels,
                         x.shape[-1])

    def loss(self, batch, lengths, use_amp):
        inputs, labels = batch[:, 0], batch[:, 1:]
        return self.criterion(self(inputs), labels, lengths).mean()

    def _enhance(self, x, use_amp):
        return x.mean(-2)
