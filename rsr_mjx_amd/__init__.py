"""rsr_mjx_amd — MI355X-native batched stepper for the RSR-MJX Airbot / Go2 environments.

Only the env hot path lives here: the MJCF-subset model compiler, the
C-ABI loader for the HIP stepper and the Python mirror of the reference env interface.
"""
__version__ = "0.1.0"
