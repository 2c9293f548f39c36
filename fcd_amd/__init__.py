"""fcd_amd: the FCD Fixed-Point GAN hot path on MI355X (gfx950) HIP kernels.

Drop-in for the reference's ``models.fixed_point_gan.Generator`` /
``Discriminator`` and the train / generate_masks part of ``fcd_solver.FCDSolver``.
There is no CPU fallback: everything here launches into ``libfcd_hip.so``.
"""
from .generator import Generator, ResidualBlock  # noqa: F401
from .discriminator import Discriminator  # noqa: F401
from .solver import FCDSolver, FusedAdam, default_config  # noqa: F401
