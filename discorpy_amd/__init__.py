"""discorpy_amd -- MI355X (gfx950) implementation of discorpy's backward-unwarp path.

Drop-in for the four remap functions of ``discorpy.post.postprocessing`` (reference file
``discorpy/post/postprocessing.py``): same names, argument order, defaults and error messages,
executed by hand-written HIP kernels through the C ABI in ``include/discorpy_hip.h``.
"""
__version__ = "0.1.0"
