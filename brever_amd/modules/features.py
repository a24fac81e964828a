"""Log-mel filterbank features on the HIP path.

Reference: ``FeatureExtractor`` (brever/modules/features.py:13-140) and its ``fbe``
family (:142-205): squared magnitude averaged over the two channels, mel filterbank,
optional normalisation / compression. Built here: ``fbe``, ``logfbe``, ``cubicfbe``
(what the FFNN model uses by default is ``logfbe``). The binaural features (``ild``,
``ipd``, ``ic``) and the DCT-based ones (``mfcc`` ...) need ``torchaudio.lfilter`` /
``scipy.fft`` host code in the reference and are not built: asking for them raises
``NotImplementedError`` (same names, so configurations fail loudly instead of silently
computing something else).
"""
import torch

from .. import hip

eps = torch.finfo().eps          # features.py:10


class FeatureExtractor:
    _built = {'fbe': 0, 'logfbe': 1, 'cubicfbe': 2}
    _known = {'ild', 'ipd', 'ic', 'fbe', 'logfbe', 'cubicfbe', 'pdf', 'logpdf', 'cubicpdf',
              'mfcc', 'cubicmfcc', 'pdfcc'}

    def __init__(self, features, mel_fb, hop_length=256, fs=16e3):
        self.features = sorted(features)
        self.mel_fb = mel_fb
        self.hop_length = hop_length
        self.fs = fs
        self.indices = None
        for f in self.features:
            if f not in self._known:
                raise ValueError(f'unrecognized feature, got {f}')
            if f not in self._built:
                raise NotImplementedError(f'feature {f} is not built yet on the HIP path')

    @property
    def n_features(self):
        return self.mel_fb.n_filters*len(self.features)

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
        out = self.fbe(x, mode=self._built[feature])
        return out.squeeze(0) if unbatched else out

    def fbe(self, x, mode=0):
        """(B, channels, bins, frames) complex -> (B, n_filters, frames)."""
        hip.require_device(x)
        lib = hip.lib()
        B, C, bins, F = x.shape
        spec = torch.view_as_real(x.to(torch.complex64).contiguous())
        power = torch.empty(B, bins, F, dtype=torch.float32, device=x.device)
        hip.check(lib.brv_fbe_power(hip.ptr(spec), hip.ptr(power), B, C, bins*F, hip.stream()),
                  'brv_fbe_power')
        out = self.mel_fb(power)
        if mode:
            hip.check(lib.brv_compress(hip.ptr(out), hip.ptr(out), out.numel(), mode,
                                       float(eps), hip.stream()), 'brv_compress')
        return out
