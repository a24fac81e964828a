"""FIR up / down sampling by 2 on the HIP path (reference: brever/modules/resampling.py:8-77,
``Resample`` and its ``Upsample`` / ``Downsample`` shorthands): a depthwise strided FIR
(``brv_fir_resample2d``) with the separable kernel ``fir_kernel x fir_kernel`` normalised to
unit sum; up-sampling is the transposed operator with the kernel scaled by 4."""
import math

import torch
import torch.nn as nn

from .. import hip


class Resample(nn.Module):
    """FIR up / down sampling by 2 (brever/modules/resampling.py:8-61); the paddings of the
    down-sampling calls are stacked and reused by the matching up-sampling calls."""

    def __init__(self, fir_kernel, buffer_padding=False):
        super().__init__()
        kernel = torch.as_tensor(fir_kernel, dtype=torch.float32)
        kernel = kernel.outer(kernel).unsqueeze(0).unsqueeze(1)
        kernel /= kernel.sum()
        self.register_buffer('kernel', kernel)
        self._paddings = [] if buffer_padding else None

    def plan(self, shape, up_or_down):
        """(padding, output (H, W), is_up) of one call; 'down' calls push their padding on the
        stack that the matching 'up' calls pop (resampling.py:30-55)."""
        H, W = shape[-2:]
        K = self.kernel.shape[-1]
        if up_or_down == 'down':
            padding = tuple(math.ceil(K/2) - 1 if dim % 2 == 0 else math.ceil((K + 1)/2) - 1
                            for dim in (H, W))
            if self._paddings is not None:
                out_pad = tuple(0 if (dim + 2*pad - K) % 2 == 0 else 1
                                for dim, pad in zip((H, W), padding))
                self._paddings.append((padding, out_pad))
            return padding, ((H + 2*padding[0] - K)//2 + 1, (W + 2*padding[1] - K)//2 + 1), False
        if up_or_down == 'up':
            if self._paddings is not None:
                padding, out_pad = self._paddings.pop()
            else:
                padding, out_pad = ((K - 1)//2,)*2, (0, 0)
            return padding, ((H - 1)*2 - 2*padding[0] + K + out_pad[0],
                             (W - 1)*2 - 2*padding[1] + K + out_pad[1]), True
        raise ValueError(f'up_or_down must be up or down, got {up_or_down}')

    def forward(self, x, up_or_down):
        x = x.contiguous()
        B, C, H, W = x.shape
        K = self.kernel.shape[-1]
        padding, (Ho, Wo), up = self.plan(x.shape, up_or_down)
        y = torch.empty(B, C, Ho, Wo, dtype=torch.float32, device=x.device)
        hip.check(hip.lib().brv_fir_resample2d(
            hip.ptr(x), hip.ptr(self.kernel.float().contiguous()), hip.ptr(y), B*C, H, W, Ho, Wo, K,
            padding[0], padding[1], int(up), 4.0 if up else 1.0, hip.stream()),
            'brv_fir_resample2d')
        return y


class Upsample(Resample):
    def __init__(self, fir_kernel):
        super().__init__(fir_kernel=fir_kernel, buffer_padding=False)

    def forward(self, x):
        return super().forward(x, 'up')


class Downsample(Resample):
    def __init__(self, fir_kernel):
        super().__init__(fir_kernel=fir_kernel, buffer_padding=False)

    def forward(self, x):
        return super().forward(x, 'down')
