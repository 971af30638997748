"""MI355X-native hot path of the A-Frame `gaussian_splatting` component.

The product is the C-ABI library `csrc/libgs_splat_hip.so` (include/gs_splat.h): hand-written HIP kernels
for gfx950.  This Python package is plumbing around it: `build` compiles it, `capi` binds it with ctypes,
`synth` generates the synthetic workloads; the reference-language host side (N-API addon + JS component
shim) lives in `js/`.

The directory name carries the reference's hyphenated name, so import it with
    importlib.import_module("aframe-gaussian-splatting_amd")
"""
__version__ = "0.1.0"
