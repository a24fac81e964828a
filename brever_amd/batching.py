"""Variable-length batch samplers (host side, pure integer logic).

Drop-in for the sampler surface of the reference (brever/batching.py:13-290):
registry keys ``random`` / ``sorted`` / ``bucket`` and the
``DistributedBatchSamplerWrapper``. The batch compositions must be *bit-exact*
with the reference for the same dataset lengths, seed and epoch, so the random
streams are consumed in exactly the same order:

* ``_seed = random.Random(seed).randrange(2**32)``          (batching.py:84)
* item order: ``random.Random(_seed + epoch)`` -> ``shuffle`` of the index list,
  or one ``random()`` tie-break draw per item in dataset order when sorting
  (batching.py:106-128)
* batch order: a *fresh* ``random.Random(_seed + epoch)`` -> ``shuffle``
  (batching.py:149-151)

Parity is pinned by ``tests/golden/batching.json`` (generated from the imported
reference by ``tests/golden/make_golden.py``).

The dynamic budget is expressed in seconds of audio per batch; on a 288 GB
MI355X it is a throughput knob rather than a memory limit (see
``hbm_batch_seconds``).
"""
import logging
import random

import numpy as np
import torch

from .registry import Registry

BatchSamplerRegistry = Registry('batch_sampler')


def hbm_batch_seconds(bytes_per_second_of_audio, hbm_bytes=288e9, reserve=0.25):
    """Largest dynamic batch size (seconds) whose saved activations fit HBM.

    ``bytes_per_second_of_audio`` is what the model keeps resident per second
    of input for its backward pass (``ConvTasNet.activation_bytes_per_second``).
    ``reserve`` is the fraction of HBM kept free for weights, optimizer state,
    RCCL buffers and the allocator.
    """
    return float(hbm_bytes) * (1.0 - reserve) / float(bytes_per_second_of_audio)


class BreverBatchSampler(torch.utils.data.Sampler):
    """Base sampler. Subclasses implement ``_generate_batches(indices)`` which
    returns a list of batches, each a list of ``(segment_idx, segment_length)``.

    ``batch_size`` is a number of segments (``dynamic=False``) or a number of
    seconds of padded audio (``dynamic=True``), see brever/batching.py:31-57.
    """

    def __init__(self, dataset, batch_size, drop_last=False, shuffle=True,
                 seed=0, dynamic=False, sort=False, fs=16000, reverse=False):
        self.dataset = dataset
        if dynamic:
            self.batch_size = round(fs*batch_size)
        else:
            if isinstance(batch_size, float):
                logging.warning('Got float batch_size even though dynamic is '
                                'False. Casting batch_size to int.')
            self.batch_size = int(batch_size)
        self.drop_last = drop_last
        self.shuffle = shuffle
        self.dynamic = dynamic
        self.sort = sort
        self.reverse = reverse
        self._seed = random.Random(seed).randrange(2**32)
        self._epoch = 0
        self._previous_epoch = -1
        self._segment_lengths = None
        self._batches = None

    # -- epoch handling ------------------------------------------------------
    def set_epoch(self, epoch):
        self._epoch = epoch

    def _rng(self):
        return random.Random(self._seed + self._epoch)

    def __iter__(self):
        if self.shuffle:
            if self._epoch == self._previous_epoch:
                raise ValueError(
                    'the set_epoch method must be called before iterating '
                    'over the dataloader in order to regenerate the batches '
                    'with the correct seed'
                )
            self.generate_batches()
            self.shuffle_batches()
            self._previous_epoch = self._epoch
        elif self._batches is None:
            self.generate_batches()
        for batch in self._batches:
            yield [idx for idx, _ in batch]

    def __len__(self):
        if self._batches is None:
            self.generate_batches()
        return len(self._batches)

    # -- batch generation ----------------------------------------------------
    def generate_batches(self):
        self._batches = self._generate_batches(self._generate_indices())

    def shuffle_batches(self):
        self._rng().shuffle(self._batches)

    def get_segment_lengths(self):
        if isinstance(self.dataset, torch.utils.data.Subset):
            dataset, indices = self.dataset.dataset, self.dataset.indices
        else:
            dataset, indices = self.dataset, range(len(self.dataset))
        # lengths are cached unless the dataset is re-mixed every epoch
        if self._segment_lengths is None or dataset.rmm_dset is not None:
            self._segment_lengths = [
                (i, dataset.get_segment_length(j))
                for i, j in enumerate(indices)
            ]

    def _generate_indices(self):
        self.get_segment_lengths()
        n = len(self._segment_lengths)
        if not self.sort:
            order = list(range(n))
            if self.shuffle:
                self._rng().shuffle(order)
            return order
        if self.shuffle:
            # one tie-break draw per item, in dataset order
            rng = self._rng()
            keyed = [(length, rng.random(), i)
                     for i, length in self._segment_lengths]
            keyed.sort(key=lambda k: (k[0], k[1]), reverse=self.reverse)
            return [i for _, _, i in keyed]
        ranked = sorted(self._segment_lengths, key=lambda x: x[1],
                        reverse=self.reverse)
        return [i for i, _ in ranked]

    def _generate_batches(self, indices):
        raise NotImplementedError

    # -- statistics ----------------------------------------------------------
    def calc_batch_stats(self, transform_length=None):
        if transform_length is None:
            def transform_length(x):
                return x
        sizes, pads = [], []
        for batch in self._batches:
            lens = [transform_length(length) for _, length in batch]
            longest = max(lens)
            sizes.append(len(batch)*longest)
            pads.append(sum(longest - length for length in lens))
        return sizes, pads


class _SequentialFillSampler(BreverBatchSampler):
    """Greedy fill in index order: close the batch when the next segment
    would overflow it (brever/batching.py:173-204)."""

    def _generate_batches(self, indices):
        batches, current = [], []
        for i in indices:
            item = self._segment_lengths[i]
            if self._overflows(current, item[1]):
                batches.append(current)
                current = []
            current.append(item)
        if current and not self.drop_last:
            batches.append(current)
        return batches

    def _overflows(self, batch, segment_length):
        if not self.dynamic:
            return len(batch) + 1 > self.batch_size
        if segment_length > self.batch_size:
            raise ValueError(
                'got a segment that is longer than the dynamic batch size'
            )
        longest = max((length for _, length in batch), default=0)
        return (len(batch) + 1)*max(segment_length, longest) > self.batch_size


@BatchSamplerRegistry.register('random')
class RandomBatchSampler(_SequentialFillSampler):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, sort=False, **kwargs)


@BatchSamplerRegistry.register('sorted')
class SortedBatchSampler(_SequentialFillSampler):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, sort=True, **kwargs)


@BatchSamplerRegistry.register('bucket')
class BucketBatchSampler(BreverBatchSampler):
    """Length buckets with uniformly spaced right limits; a bucket is emitted
    as a batch as soon as it is full (brever/batching.py:219-276)."""

    def __init__(self, *args, num_buckets=10, **kwargs):
        super().__init__(*args, **kwargs)
        self.num_buckets = num_buckets

    def _generate_batches(self, indices):
        longest = max(length for _, length in self._segment_lengths)
        limits = np.linspace(longest/self.num_buckets, longest,
                             self.num_buckets)
        self.right_bucket_limits = limits
        if self.dynamic:
            capacity = self.batch_size//limits
        else:
            capacity = [self.batch_size]*self.num_buckets
        batches = []
        buckets = [[] for _ in range(self.num_buckets)]
        for i in indices:
            item = self._segment_lengths[i]
            b = np.searchsorted(limits, item[1])
            if not 0 <= b < self.num_buckets:
                raise ValueError(
                    'attempted to assign a segment to a non-existent bucket'
                )
            buckets[b].append(item)
            if len(buckets[b]) == capacity[b]:
                batches.append(buckets[b])
                buckets[b] = []
            elif len(buckets[b]) > capacity[b]:
                raise ValueError(
                    'maximum number of segments allowed in bucket exceeded'
                )
        if not self.drop_last:
            batches.extend(bucket for bucket in buckets if bucket)
        return batches


class DistributedBatchSamplerWrapper(torch.utils.data.DistributedSampler):
    """Shards *batches* (not items) over ranks: rank ``r`` takes every
    ``world``-th entry of a seeded permutation of the batch list, padded by
    repetition so every rank runs the same number of steps
    (brever/batching.py:279-290).

    Faithful to the reference, the wrapped sampler's batch list is composed
    once (through ``len(sampler)``) and only the batch -> rank assignment is
    reshuffled per epoch.
    """

    def __init__(self, sampler, *args, **kwargs):
        super().__init__(dataset=sampler, *args, **kwargs)
        self.sampler = sampler

    def __iter__(self):
        for j in super().__iter__():
            yield [idx for idx, _ in self.sampler._batches[j]]

    def set_epoch(self, epoch):
        super().set_epoch(epoch)
        self.sampler.set_epoch(epoch)
