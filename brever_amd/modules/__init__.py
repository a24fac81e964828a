"""DSP modules of the HIP path (reference: brever/modules/)."""
from .stft import STFT, ConvSTFT, MelFilterbank  # noqa: F401
from .features import FeatureExtractor  # noqa: F401
from .ema import EMA, EMAKarras  # noqa: F401
from .normalization import CausalGroupNorm, CausalInstanceNorm, CausalLayerNorm  # noqa: F401
from .resampling import Downsample, Resample, Upsample  # noqa: F401
