"""tempestmodel_amd -- MI355X-native spectral-element dynamics engine behind Tempest's
HorizontalDynamics / VerticalDynamics / TimestepScheme plugin surface.

The compute path is the HIP library ``libtempest_mi355x.so`` (C ABI in
``include/tempest_mi355x.h``); this package holds the ctypes binding, the host-side mirror of
the reference plugin classes and the host grid / test-case synthesiser.  There is no CPU
fallback: importing :mod:`tempestmodel_amd.engine` fails loudly if the library is missing.
"""
__version__ = "0.1.0"
