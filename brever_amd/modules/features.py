"""Filterbank features on the HIP path.

Reference: ``FeatureExtractor`` (brever/modules/features.py:13-140), its ``fbe`` family
(:142-220: squared magnitude averaged over the two channels, mel filterbank, optional
normalisation over the filters ('pdf'), log / cubic compression, DCT-II + delta /
double-delta rows ('mfcc' ...)) and the binaural level / phase differences ``ild`` / ``ipd``
(:222-262). Every step is a kernel of ``libbrever_hip.so`` (``brv_fbe_power``,
``brv_binaural``, the mel and DCT products ``brv_matmul_f32``, ``brv_col_normalize``,
``brv_compress``, ``brv_deltas``). Not built: ``ic`` (interaural coherence), whose recursive
smoothing is ``torchaudio.functional.lfilter`` in the reference (absent wheel, parity
unpinned): asking for it raises ``NotImplementedError``.
"""
import math

import numpy as np
import torch

from .. import hip

eps = torch.finfo().eps          # features.py:10


class FeatureExtractor:
    # name -> (normalize, compression mode, dct); compression 0 none, 1 log, 2 cubic
    _fbe_family = {'fbe': (False, 0, False), 'logfbe': (False, 1, False),
                   'cubicfbe': (False, 2, False), 'pdf': (True, 0, False),
                   'logpdf': (True, 1, False), 'cubicpdf': (True, 2, False),
                   'mfcc': (False, 1, True), 'cubicmfcc': (False, 2, True),
                   'pdfcc': (True, 1, True)}
    _known = set(_fbe_family) | {'ild', 'ipd', 'ic'}
    _n_dct = 14                   # DCT coefficients kept, the DC term is dropped (features.py:143)

    def __init__(self, features, mel_fb, hop_length=256, fs=16e3):
        self.features = sorted(features)
        self.mel_fb = mel_fb
        self.hop_length = hop_length
        self.fs = fs
        self.indices = None
        self._dct = None
        for f in self.features:
            if f not in self._known:
                raise ValueError(f'unrecognized feature, got {f}')

    def _n(self, feature):
        # the reference's table lists 13 for the DCT features although they come with their
        # delta and double-delta rows (39 rows); kept for an identical n_features
        return 13 if self._fbe_family.get(feature, (0, 0, False))[2] else self.mel_fb.n_filters

    @property
    def n_features(self):
        return sum(self._n(f) for f in self.features)

    def __call__(self, x):
        output = []
        self.indices = {}
        i_start = 0
        for feature in self.features:
            data = self.calc_feature(x, feature)
            output.append(data)
            i_end = i_start + data.shape[-2]
            self.indices[feature] = (i_start, i_end)
            i_start = i_end
        return torch.cat(output, dim=-2)

    def calc_feature(self, x, feature):
        unbatched = x.ndim == 3
        if unbatched:
            x = x.unsqueeze(0)
        elif x.ndim != 4:
            raise ValueError(f'input must be 3 or 4 dimensional, got {x.ndim}')
        if feature in ('ild', 'ipd'):
            out = self.binaural(x, 0 if feature == 'ild' else 1)
        elif feature in self._fbe_family:
            out = self.fbe(x, *self._fbe_family[feature])
        elif feature == 'ic':
            out = self.ic(x)
        else:
            raise ValueError(f'unrecognized feature, got {feature}')
        return out.squeeze(0) if unbatched else out

    def binaural(self, x, mode):
        """ILD / IPD of (B, 2, bins, frames) complex -> (B, n_filters, frames)."""
        hip.require_device(x)
        B, C, bins, F = x.shape
        if C != 2:
            raise ValueError(f'binaural features need 2 channels, got {C}')
        spec = torch.view_as_real(x.to(torch.complex64).contiguous())
        cue = torch.empty(B, bins, F, dtype=torch.float32, device=x.device)
        hip.check(hip.lib().brv_binaural(hip.ptr(spec), hip.ptr(cue), B, bins*F, mode, float(eps),
                                         hip.stream()), 'brv_binaural')
        return self.mel_fb(cue)

    def ic(self, x, tau=10e-3):
        """Interaural coherence (features.py:263-293) of (B, 2, bins, frames) complex ->
        (B, n_filters, frames)."""
        hip.require_device(x)
        lib = hip.lib()
        B, C, bins, F = x.shape
        if C != 2:
            raise ValueError(f'binaural features need 2 channels, got {C}')
        alpha = math.exp(-self.hop_length/(tau*self.fs))
        spec = torch.view_as_real(x.to(torch.complex64).contiguous())
        coh = torch.empty(B, bins, F, dtype=torch.float32, device=x.device)
        hip.check(lib.brv_interaural_coherence(hip.ptr(spec), hip.ptr(coh), B, bins, F, alpha,
                                               hip.stream()), 'brv_interaural_coherence')
        out = self.mel_fb(coh).contiguous()
        hip.check(lib.brv_compress(hip.ptr(out), hip.ptr(out), out.numel(), 3, 0.0, hip.stream()),
                  'brv_compress')
        return out

    def _dct_matrix(self, M, device):
        """Rows 1..n_dct-1 of the orthonormal DCT-II of length M (scipy.fft.dct(type=2,
        norm='ortho'), features.py:200-205), built in float64."""
        if self._dct is None or self._dct.shape[1] != M or self._dct.device != device:
            k = np.arange(1, self._n_dct)[:, None]
            n = np.arange(M)[None, :]
            D = math.sqrt(2.0/M)*np.cos(math.pi*(2*n + 1)*k/(2*M))
            self._dct = torch.from_numpy(D).float().to(device).contiguous()
        return self._dct

    def fbe(self, x, normalize=False, compression=0, dct=False):
        """(B, channels, bins, frames) complex -> (B, n_filters, frames), or (B, 39, frames) with
        the DCT (13 cepstral rows + their first and second differences)."""
        hip.require_device(x)
        lib = hip.lib()
        B, C, bins, F = x.shape
        spec = torch.view_as_real(x.to(torch.complex64).contiguous())
        power = torch.empty(B, bins, F, dtype=torch.float32, device=x.device)
        hip.check(lib.brv_fbe_power(hip.ptr(spec), hip.ptr(power), B, C, bins*F, hip.stream()),
                  'brv_fbe_power')
        out = self.mel_fb(power).contiguous()
        M = out.shape[1]
        if normalize:
            hip.check(lib.brv_col_normalize(hip.ptr(out), B, M, F, float(eps), hip.stream()),
                      'brv_col_normalize')
        if compression:
            hip.check(lib.brv_compress(hip.ptr(out), hip.ptr(out), out.numel(), compression,
                                       float(eps), hip.stream()), 'brv_compress')
        if dct:
            D = self._dct_matrix(M, x.device)
            K = D.shape[0]
            cep = torch.empty(B, K, F, dtype=torch.float32, device=x.device)
            hip.check(lib.brv_matmul_f32(hip.ptr(D), hip.ptr(out), hip.ptr(cep), B, K, F, M, 0,
                                         hip.stream()), 'brv_matmul_f32')
            out = torch.empty(B, 3*K, F, dtype=torch.float32, device=x.device)
            hip.check(lib.brv_deltas(hip.ptr(cep), hip.ptr(out), B, K, F, hip.stream()),
                      'brv_deltas')
        return out
