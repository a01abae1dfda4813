"""synthpy_amd — MI355X-native engine for synthPy's ray-propagation → detector hot path.

Host code mirrors the reference's two API generations and calls hand-written HIP
kernels (synthpy_amd/csrc) through the C ABI in include/synthray.h:

    synthpy_amd.solvers_legacy.full_solver   ScalarDomain(x, y, z, extent) / init_beam / solve
    synthpy_amd.solvers_legacy.rtm_solver    Shadowgraphy / Schlieren / Refractometry / Interferometry
    synthpy_amd.simulator.{domain,beam,propagator,diagnostics}   the JAX-generation surface
    synthpy_amd.field_generator.gaussian3D   benchmark volumes (host NumPy, an input of the path)
    synthpy_amd.engine                       device volumes, ray bundles, detector images
    synthpy_amd.distributed                  ray sharding over GPUs + the image sum

There is no CPU implementation of the path in this package: without libsynthray.so
the compute modules fail to import, without a GPU the first device call raises.
"""
__version__ = "0.1.0"
