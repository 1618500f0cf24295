"""clavaddpm_amd: MI355X-native (gfx950 HIP) implementation of ClavaDDPM's diffusion hot path.

Drop-in modules mirror the reference layout: `clavaddpm_amd.tab_ddpm`,
`clavaddpm_amd.scripts`; kernels live in `clavaddpm_amd/csrc` behind the C ABI of
`include/clava_hip.h`.
"""
__version__ = "0.1.0"
