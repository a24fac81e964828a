"""Data side of the hot path: collation of ragged items and the dataset
protocol the samplers / trainer rely on.

Reference: brever/data.py:389-491 (``BreverDataLoader``) and the attribute /
method set of ``BreverDataset`` that ``BreverTrainer`` and the samplers touch
(brever/data.py:225-326; the minimal spec is the reference's own
tests/utils.py:9-42 ``DummyDataset``). ``BreverDataset`` reads the
reference's dataset layout -- ``audio/NNNNN_<source>.flac`` in a directory or in
``audio.tar`` -- with the reference's segmentation strategies (brever/data.py:112-210,
integer arithmetic, bit-exact against fixtures); WAV files are decoded here, FLAC by the
native decoder of the library (``csrc/flac.hip``, RFC 9639; the ``soundfile`` wheel the reference
uses is absent). The benchmark and most tests run on synthetic mixtures generated in memory by
``SyntheticMixtureDataset``.
"""
import io
import logging
import os
import random
import re
import struct
import tarfile

import numpy as np
import torch
import torch.nn.functional as F

from .inspect import NoParse


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


class DevicePrefetcher:
    """Pinned, double-buffered host -> HBM staging of the batches of a loader.

    The reference moves every batch with a synchronous ``.to(device)`` inside the step
    loop (brever/training.py:310-314) after an optional ``pin_memory`` DataLoader
    (brever/data.py:494-530). Here batch ``i + 1`` is copied on a side HIP stream from
    one of ``depth`` reusable pinned host buffers while batch ``i`` computes; the
    consumer's stream waits on the copy's event only. A 16 x 2 x 64 000 fp32 batch is
    8.2 MB (~0.15 ms over PCIe Gen5): fully hidden behind a multi-millisecond step.
    Iterating yields the same ``(batch, lengths)`` pairs as the loader, on ``device``.
    CPU devices pass through unchanged.
    """

    def __init__(self, loader, device, depth=2):
        self.loader = loader
        self.device = torch.device(device)
        self.depth = max(1, int(depth))
        self._pinned = {}

    def __len__(self):
        return len(self.loader)

    def _pin(self, slot, key, t):
        """Copy ``t`` into this slot's pinned buffer (grown on demand)."""
        buf = self._pinned.get((slot, key))
        n = t.numel()
        if buf is None or buf.dtype != t.dtype or buf.numel() < n:
            buf = torch.empty(max(n, 1), dtype=t.dtype).pin_memory()
            self._pinned[(slot, key)] = buf
        view = buf[:n].view(t.shape)
        view.copy_(t)
        return view

    def _stage(self, item, slot, stream, fence):
        if fence is not None:
            fence.synchronize()          # the copy that last used this slot's buffers is done
        batch, lengths = item
        with torch.cuda.stream(stream):
            def up(key, t):
                if t.is_cuda:
                    return t
                return self._pin(slot, key, t.contiguous()).to(self.device, non_blocking=True)
            if isinstance(batch, (list, tuple)):
                dev_batch = [up(('b', i), x) for i, x in enumerate(batch)]
            else:
                dev_batch = up('b', batch)
            dev_lengths = up('l', lengths)
            ev = torch.cuda.Event()
            ev.record(stream)
        return dev_batch, dev_lengths, ev

    def __iter__(self):
        if self.device.type != 'cuda':
            yield from self.loader
            return
        stream = torch.cuda.Stream(self.device)
        it = iter(self.loader)
        queue, fences, slot = [], [None]*self.depth, 0

        def fill():
            nonlocal slot
            try:
                item = next(it)
            except StopIteration:
                return False
            staged = self._stage(item, slot, stream, fences[slot])
            fences[slot] = staged[2]
            queue.append(staged)
            slot = (slot + 1) % self.depth
            return True

        for _ in range(self.depth):
            if not fill():
                break
        while queue:
            batch, lengths, ev = queue.pop(0)
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for t in (batch if isinstance(batch, list) else [batch]) + [lengths]:
                t.record_stream(cur)
            fill()                       # next copy runs while this batch computes
            yield batch, lengths


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

    def preload_indices(self, indices, device='cpu'):
        """Keep only ``indices`` in memory (one rank's share of a dataset every rank indexes the same
        way: ``bench.py --gpus N``); other items are still synthesised on demand."""
        class _Cache(dict):
            def __missing__(cache, i):
                item = self.make_item(i)
                return self.transform(item) if self.transform is not None else item
        self.preloaded_data = None
        data = _Cache()
        for i in indices:
            data[i] = self[i].to(device)
        self.preloaded_data = data

    def set_epoch(self, epoch):
        pass


# ---------------------------------------------------------------------------------------------
# audio files
# ---------------------------------------------------------------------------------------------
def _wav_header(f):
    """(sample rate, channels, frames, format tag, bits, data offset) of a RIFF/WAVE stream."""
    head = f.read(12)
    if len(head) < 12 or head[:4] != b'RIFF' or head[8:12] != b'WAVE':
        raise ValueError('not a RIFF/WAVE file')
    fmt = None
    while True:
        chunk = f.read(8)
        if len(chunk) < 8:
            raise ValueError('WAVE file without a data chunk')
        tag, size = chunk[:4], struct.unpack('<I', chunk[4:])[0]
        if tag == b'fmt ':
            body = f.read(size + (size & 1))
            code, channels, rate, _, _, bits = struct.unpack('<HHIIHH', body[:16])
            if code == 0xFFFE and size >= 26:                   # WAVE_FORMAT_EXTENSIBLE
                code = struct.unpack('<H', body[24:26])[0]
            fmt = (rate, channels, code, bits)
        elif tag == b'data':
            if fmt is None:
                raise ValueError('WAVE data chunk before fmt chunk')
            rate, channels, code, bits = fmt
            return rate, channels, size//(channels*bits//8), code, bits, f.tell()
        else:
            f.seek(size + (size & 1), io.SEEK_CUR)


def _flac_info(data, name):
    import ctypes
    from . import hip
    frames, rate = ctypes.c_int64(0), ctypes.c_int32(0)
    channels, bits = ctypes.c_int32(0), ctypes.c_int32(0)
    status = hip.lib().brv_flac_info(data, len(data), ctypes.byref(frames), ctypes.byref(rate),
                                     ctypes.byref(channels), ctypes.byref(bits))
    if status != 0:
        raise ValueError(f'{name}: not a FLAC stream this decoder reads (status {status})')
    return frames.value, rate.value, channels.value, bits.value


def audio_info(f, name):
    """(frames, sample rate) of an open audio file (``torchaudio.info`` in data.py:143). WAV
    and FLAC headers are parsed here (FLAC: ``brv_flac_info`` of the native library)."""
    if name.lower().endswith('.wav'):
        rate, _, frames, _, _, _ = _wav_header(f)
        return frames, rate
    head = f.read(1 << 16)                 # STREAMINFO is the first metadata block
    try:
        frames, rate, _, _ = _flac_info(head, name)
    except ValueError:                     # metadata (pictures, tags) longer than the head
        head += f.read()
        frames, rate, _, _ = _flac_info(head, name)
    if frames == 0:                        # unknown length in the header: decode to count
        return len(audio_read(io.BytesIO(head + f.read()), name)[0]), rate
    return frames, rate


def write_flac(path, x, fs):
    """Mono 16-bit FLAC file of the float signal ``x`` (what ``torchaudio.save(<name>.flac, x, fs)`` of the
    reference's ``scripts/test_model.py:201-209`` produces for a float tensor: samples clipped to [-1, 1),
    scaled by 2^15 and rounded) through the native encoder ``brv_flac_encode16`` (csrc/flac.hip)."""
    import ctypes

    import numpy as np

    from . import hip
    pcm = np.clip(np.rint(np.asarray(x, dtype=np.float64).reshape(-1)*32768.0), -32768, 32767).astype(np.int16)
    lib = hip.lib()
    src = pcm.ctypes.data_as(ctypes.c_void_p)
    size = lib.brv_flac_encode16(src, pcm.size, int(fs), None, 0)
    if size < 0:
        raise RuntimeError(f'brv_flac_encode16 failed ({size})')
    buf = (ctypes.c_uint8*size)()
    got = lib.brv_flac_encode16(src, pcm.size, int(fs), buf, size)
    if got != size:
        raise RuntimeError(f'brv_flac_encode16 failed ({got})')
    with open(path, 'wb') as f:
        f.write(bytes(buf))


def audio_read(f, name):
    """float32 array (frames,) or (frames, channels) and the sample rate (``sf.read`` in
    data.py:265). FLAC goes through the native decoder ``brv_flac_decode`` (csrc/flac.hip)."""
    if not name.lower().endswith('.wav'):
        import ctypes
        from . import hip
        data = f.read()
        frames, rate, channels, _ = _flac_info(data, name)
        capacity = frames if frames > 0 else 8*len(data)
        out = np.empty((capacity, channels), dtype=np.float32)
        got = hip.lib().brv_flac_decode(data, len(data),
                                        out.ctypes.data_as(ctypes.c_void_p), capacity)
        if got > capacity:
            # the decoder counts the frames of the whole stream and stores only what fits: streams
            # without a length in the header whose frames compress below a byte each (silence:
            # CONSTANT subframes) need a second pass with the exact size (they used to be truncated)
            capacity = int(got)
            out = np.empty((capacity, channels), dtype=np.float32)
            got = hip.lib().brv_flac_decode(data, len(data),
                                            out.ctypes.data_as(ctypes.c_void_p), capacity)
        if got < 0:
            raise ValueError(f'{name}: FLAC decoding failed (status {got})')
        if got > capacity:
            raise ValueError(f'{name}: FLAC stream longer than its decoded size ({got} > {capacity})')
        out = out[:got]
        return (out if channels > 1 else out[:, 0]), rate
    rate, channels, frames, code, bits, _ = _wav_header(f)
    raw = f.read(frames*channels*bits//8)
    if code == 3 and bits == 32:
        x = np.frombuffer(raw, dtype='<f4').astype(np.float32)
    elif code == 1 and bits == 16:
        x = np.frombuffer(raw, dtype='<i2').astype(np.float32)/32768.0
    elif code == 1 and bits == 32:
        x = (np.frombuffer(raw, dtype='<i4').astype(np.float64)/2147483648.0).astype(np.float32)
    elif code == 1 and bits == 24:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = (np.where(v >= 1 << 23, v - (1 << 24), v)/8388608.0).astype(np.float32)
    else:
        raise ValueError(f'unsupported WAVE encoding (format {code}, {bits} bits)')
    return (x.reshape(frames, channels) if channels > 1 else x), rate


class TarArchive:
    """Member lookup in ``audio.tar`` with one file handle per DataLoader worker
    (``tarfile`` objects cannot be shared across processes; brever/data.py:374-386)."""

    def __init__(self, archive):
        self.archive = archive
        self._handles = {}
        self.members = {m.name: m for m in self._handle().getmembers()}

    def _handle(self):
        info = torch.utils.data.get_worker_info()
        key = info.id if info is not None else None
        if key not in self._handles:
            self._handles[key] = tarfile.open(self.archive)
        return self._handles[key]

    def get_file(self, name):
        return self._handle().extractfile(self.members[name])


_mixture_maker = None


def set_mixture_maker(factory):
    """Install the mixture maker behind ``BreverDataset(dynamic_mixing=True)``.

    The reference hard-wires ``RandomMixtureMakerDataset`` (brever/data.py:494-532), which
    synthesises mixtures on the fly from raw speech / noise corpora and BRIRs through
    ``brever.mixture.RandomMixtureMaker`` -- the dataset-synthesis subsystem, out of scope of
    this build. The dataset side of dynamic mixing IS built: any class with that object's
    protocol plugs in here --

        factory(path, sources=[...], size=N)   # N mixtures per epoch
        .set_epoch(epoch)                      # redraw the epoch's mixtures
        .file_lengths                          # list of N lengths in samples
        [i] -> list of float32 arrays (frames, 2), one per source

    ``SyntheticMixtureMaker`` below is such a class (the synthetic recipe of SURVEY.md 8d).
    ``None`` uninstalls."""
    global _mixture_maker
    _mixture_maker = factory


class SyntheticMixtureMaker:
    """Mixture maker in the ``RandomMixtureMakerDataset`` protocol drawing the synthetic noisy /
    clean pairs of ``SyntheticMixtureDataset`` with a new seed and new lengths every epoch."""

    def __init__(self, path, sources, size, fs=16000, min_seconds=1.0, max_seconds=4.0):
        self.sources, self.size, self.fs = list(sources), size, fs
        self.bounds = (int(min_seconds*fs), int(max_seconds*fs))
        self.set_epoch(0)

    def set_epoch(self, epoch):
        rng = random.Random(epoch)
        self._lengths = [rng.randint(*self.bounds) for _ in range(self.size)]
        self._seed = 7919*epoch

    @property
    def file_lengths(self):
        return list(self._lengths)

    def __getitem__(self, i):
        L = self._lengths[i]
        g = torch.Generator().manual_seed(self._seed + i)
        clean = 0.1*torch.randn(L, generator=g)
        noise = 0.1*torch.randn(L, generator=g)
        snr = -5 + 15*float(torch.rand(1, generator=g))
        mix = clean + noise*(clean.norm()/noise.norm())*10**(-snr/20)
        named = {'mixture': mix, 'foreground': clean, 'background': mix - clean}
        return [named[s].unsqueeze(1).repeat(1, 2).numpy().astype('float32')
                for s in self.sources]


class BreverDataset(torch.utils.data.Dataset):
    """Reads a dataset made by the reference's ``scripts/create_dataset.py``: same constructor,
    segmentation and item contract as brever/data.py:23-326 (items are
    ``(n_sources, 2, n_samples)`` float32 tensors, or whatever ``transform`` makes of them).
    ``dynamic_mixing=True`` draws every epoch's mixtures from the installed mixture maker
    (``set_mixture_maker``) instead of the audio files."""

    def __init__(
        self,
        path: NoParse[str],
        segment_length: float = 0.0,
        overlap_length: float = 0.0,
        fs: int = 16000,
        sources: list[str] = ['mixture', 'foreground'],
        segment_strategy: str = 'pass',
        max_segment_length: float = 0.0,
        tar: bool = True,
        transform: NoParse[object] = None,
        dynamic_mixing: bool = False,
        dynamic_mixtures_per_epoch: int = 1000,
    ):
        if dynamic_mixing and _mixture_maker is None:
            raise NotImplementedError(
                'dynamic_mixing=True needs a mixture maker: the reference synthesises mixtures '
                'from raw corpora with brever.mixture.RandomMixtureMaker, which is outside this '
                'build; install a class with the same protocol through '
                'brever_amd.data.set_mixture_maker (e.g. SyntheticMixtureMaker)')
        self.path = path
        self.segment_length = round(segment_length*fs)
        self.overlap_length = round(overlap_length*fs)
        self.fs = fs
        self.sources = list(sources)
        self.segment_strategy = segment_strategy
        self.max_segment_length = round(max_segment_length*fs)
        self.archive = TarArchive(os.path.join(path, 'audio.tar')) \
            if tar and not dynamic_mixing else None
        self.rmm_dset = _mixture_maker(path, sources=self.sources,
                                       size=dynamic_mixtures_per_epoch) \
            if dynamic_mixing else None
        self.transform = transform
        self.preloaded_data = None
        self._ext = None
        self.get_segment_info()

    # -- files ---------------------------------------------------------------------------------
    def _names(self):
        if self.archive is None:
            return [f'audio/{f}' for f in os.listdir(os.path.join(self.path, 'audio'))]
        return list(self.archive.members)

    def count_files(self):
        found = [re.match(r'audio/(\d+)_.+\.(flac|wav)$', n) for n in self._names()]
        found = [m for m in found if m]
        if not found:
            raise FileNotFoundError(f'no audio/NNNNN_<source>.flac|wav files under {self.path}')
        self._ext = found[0].group(2)
        return max(int(m.group(1)) for m in found) + 1

    def build_paths(self, file_idx):
        return [os.path.join('audio', f'{file_idx:05d}_{source}.{self._ext}')
                for source in self.sources]

    def get_file(self, name):
        if self.archive is None:
            return open(os.path.join(self.path, name), 'rb')
        return self.archive.get_file(name.replace('\\', '/'))

    def get_file_lengths(self):
        if self.rmm_dset is not None:                  # data.py:155-158
            self._duration = float('inf')
            return list(self.rmm_dset.file_lengths)
        lengths = []
        for file_idx in range(self.count_files()):
            per_source = []
            for p in self.build_paths(file_idx):
                with self.get_file(p) as f:
                    per_source.append(audio_info(f, p)[0])
            if any(n != per_source[0] for n in per_source):
                raise ValueError(f'sources {file_idx} do not all have the same length')
            lengths.append(per_source[0])
        self._duration = sum(lengths)/self.fs
        return lengths

    # -- segmentation (integer arithmetic) -----------------------------------------------------
    def get_segment_info(self):
        self._segment_info = segment_table(
            self.get_file_lengths(), self.segment_length, self.overlap_length,
            self.segment_strategy, self.max_segment_length, owner=self)
        self._effective_duration = float('inf') if self.rmm_dset is not None else \
            sum(e - s for _, (s, e) in self._segment_info)/self.fs

    def __len__(self):
        return len(self._segment_info)

    def get_segment_length(self, i):
        if self.segment_strategy == 'random':
            return self.segment_length
        _, (start, end) = self._segment_info[i]
        return end - start

    def get_max_segment_length(self):
        if self.segment_strategy == 'random':
            return self.segment_length
        return max(end - start for _, (start, end) in self._segment_info)

    # -- items -------------------------------------------------------------------------------
    def load_file(self, path):
        with self.get_file(path) as f:
            x, fs = audio_read(f, path)
        if fs != self.fs:
            raise ValueError('file sampling rate does not match dataset fs attribute, got '
                             f'{fs} and {self.fs}')
        return x

    def load_segment(self, index):
        file_idx, (start, end) = self._segment_info[index]
        if self.segment_strategy == 'random' and self.segment_length != 0.0:
            start = random.randint(start, end - self.segment_length)
            end = start + self.segment_length
        if self.rmm_dset is None:
            loaded = [self.load_file(p) for p in self.build_paths(file_idx)]
        else:
            loaded = self.rmm_dset[file_idx]
        sources = torch.from_numpy(np.stack(loaded))
        sources = sources.unsqueeze(1) if sources.ndim == 2 else sources.transpose(1, 2)
        if end > sources.shape[-1]:
            if self.segment_strategy not in ('pad', 'random'):
                raise ValueError("attempting to load a segment outside of file range but segment "
                                 f"strategy is not in ['pad', 'random'], got "
                                 f'{self.segment_strategy}')
            sources = F.pad(sources, (0, end - sources.shape[-1]))
        return sources[..., start:end]

    def __getitem__(self, index):
        if self.preloaded_data is not None:
            return self.preloaded_data[index]
        sources = self.load_segment(index)
        if self.transform is not None:
            sources = self.transform(sources)
        return sources

    def preload(self, device, tqdm_desc=None):
        if self.segment_strategy == 'random':
            raise ValueError("can't preload when segment_strategy is 'random'")
        if self.rmm_dset is not None:
            raise ValueError("can't preload when using dynamic mixing")
        data = []
        for i in range(len(self)):
            item = self[i]
            data.append(item.to(device) if isinstance(item, torch.Tensor)
                        else [t.to(device) for t in item])
        self.preloaded_data = data      # only now: __getitem__ must not see a partial list

    def set_epoch(self, epoch):
        if self.rmm_dset is not None:                  # new mixtures, new lengths (data.py:323-326)
            self.rmm_dset.set_epoch(epoch)
            self.get_segment_info()


def segment_table(file_lengths, segment_length, overlap_length, strategy, max_segment_length=0,
                  owner=None):
    """``[(file index, (start, end)), ...]`` in samples for the five trailing-segment
    strategies (brever/data.py:112-210). With ``segment_length == 0`` files are taken whole
    unless one exceeds ``max_segment_length``, which then becomes the segment length (and is
    written back to ``owner.segment_length`` as the reference does)."""
    if segment_length == 0 and max_segment_length != 0 and max(file_lengths) > max_segment_length:
        logging.warning('Found a file longer than max_segment_length. Setting segment_length '
                        f'to max_segment_length ({max_segment_length}).')
        segment_length = max_segment_length
        if owner is not None:
            owner.segment_length = segment_length
    if segment_length == 0:
        return [(i, (0, n)) for i, n in enumerate(file_lengths)]
    if strategy not in ('drop', 'pass', 'pad', 'overlap', 'random'):
        raise ValueError(f'unrecognized segment strategy, got {strategy}')
    table = []
    hop = segment_length - overlap_length
    for i, n in enumerate(file_lengths):
        if strategy == 'random':
            table.append((i, (0, max(n, segment_length))))
            continue
        whole = (n - segment_length)//hop + 1
        table.extend((i, (k*hop, k*hop + segment_length)) for k in range(whole))
        covered = (whole - 1)*hop + segment_length if whole > 0 else 0
        if covered == n or strategy == 'drop':
            continue
        # files shorter than a segment with overlap give whole < 0 and a negative start, exactly
        # as the reference computes it (data.py:186-203); kept for identical segment tables
        tail_start = whole*hop
        if strategy == 'pass':
            table.append((i, (tail_start, n)))
        elif strategy == 'pad':
            table.append((i, (tail_start, tail_start + segment_length)))
        else:                                                     # overlap
            table.append((i, (n - segment_length, n)))
    return table
