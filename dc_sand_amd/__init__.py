"""dc_sand_amd -- MI355X (gfx950) steering-coefficient generator.

A from-scratch, AMD-native implementation of the one hot path of
ska-sa/dc_sand's ``beamformer_coefficient_generator``: per-(antenna, beam,
channel, time) delay evaluation -> sincos -> packed complex weights.

Layers (top to bottom):

* :mod:`dc_sand_amd.beamformer_coeff_test` / :mod:`dc_sand_amd.unit_test` --
  Python mirror of the reference's ``BeamformerCoeffTest`` / ``UnitTest``
  five-phase harness (the host side, in the ``pycuda_example`` shape);
* :mod:`dc_sand_amd.generator` -- thin object wrapper over the C-ABI;
* :mod:`dc_sand_amd._lib` -- ctypes binding of ``include/dcs_beamformer.h``;
* ``csrc/`` -- hand-written HIP kernels + the C-ABI (``libdcs_beamformer.so``).

There is no CPU fallback: importing :mod:`dc_sand_amd._lib` without the built
library raises, and every compute entry point needs a HIP device.
"""

from .parameters import BeamformerParameters, delay_vals_dtype  # noqa: F401

__all__ = ["BeamformerParameters", "delay_vals_dtype"]
__version__ = "0.1.0"
