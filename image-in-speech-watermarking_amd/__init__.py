"""MI355X-native image-in-speech watermark train step (encode -> attack -> decode -> loss).

Hand-written gfx950 HIP kernels (csrc/, C ABI in include/wm_kernels.h) behind the reference's own Python
surface.  Import with ``importlib.import_module("image-in-speech-watermarking_amd")``.
"""
from ._lib import lib, WmKernelError, EXPORTS, LIB_PATH  # noqa: F401  (raises ImportError if the .so is missing)
from . import ops, attacks, dataprep, evaluate, checkpoint, torch_ops, losses, noise_layers, hidden_nets  # noqa: F401
from .model import UformerAudio, get_arch  # noqa: F401,E402
from .train import FusedAdamW, TrainStep, GradSync  # noqa: F401,E402
from .attacks import AttackConfig  # noqa: F401,E402
from .small_models import ModelA, HiddenDecoder, ConvBNRelu, decoded_ber, mse_loss  # noqa: F401,E402
from .losses import CharbonnierLoss, TVLoss, tv_loss  # noqa: F401,E402
from .noise_layers import Noiser, Identity, Crop, Cropout, Dropout, Resize, Quantization, JpegCompression, parse_noise  # noqa: F401,E402
from .hidden_nets import HiddenEncoder, HiddenDiscriminator, EncoderDecoder, Hidden  # noqa: F401,E402
