"""gemm_gan_amd - MI355X-native WGAN-GP engine for GeMM-GAN's conditional generator/critic hot path.

Public surface = the reference's own names (see model.py) + the low-level Engine wrapper.
Importing this package does not need a GPU; constructing an Engine / building a WGAN_GP does, and
fails loudly otherwise (no CPU fallback on the product path).
"""
from ._lib import LIB_PATH, build, load  # noqa: F401
from .engine import Engine  # noqa: F401
from .model import WGAN_GP, WGAN_GP_model, discriminator, generator, rccl_process_group_options  # noqa: F401
from . import film  # noqa: F401  (FiLM-only sibling, src/conditional_gan_film.py: gemm_gan_amd.film.WGAN_GP ...)
from . import img_transformer  # noqa: F401  (src/conditional_gan_img_transformer.py)
from . import vanilla  # noqa: F401  (src/vanilla_gan_unconditional.py)

__all__ = ["Engine", "WGAN_GP", "WGAN_GP_model", "generator", "discriminator", "rccl_process_group_options", "build", "load", "LIB_PATH"]
