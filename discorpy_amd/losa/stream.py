"""Out-of-core correction of a whole tomography stack.

``correct_stack(src, dst, xcenter, ycenter, list_fact)`` writes the distortion-corrected stack of ``src`` --
any ``(depth, height, width)`` dataset with ``shape`` / ``dtype`` / slicing: an h5py dataset
(``discorpy.losa.loadersaver.load_hdf_object``), a ``numpy.memmap``, an array -- into ``dst`` (an h5py dataset
from ``open_hdf_stream``, a memmap, an array) without ever holding more than one pass in memory.  It is the
loop a user of the reference writes around ``unwarp_chunk_slices_backward`` (``examples/example_04.py:92-96``),
with the reads of the next pass overlapped with the GPU work of the current one by the stack path itself
(``discorpy_amd.post.postprocessing._stack_rows_lazy``).
"""
import numpy as np

from ..post import postprocessing as _pp

__all__ = ["correct_stack"]


def correct_stack(src, dst, xcenter, ycenter, list_fact, rows_per_pass=None, row_range=None, blend=None):
    """
    Parameters
    ----------
    src : array_like, (depth, height, width)
        Projections; read lazily, band by band.
    dst : array_like, (depth, height, width) or (depth, stop - start, width) when ``row_range`` is given
        Receives the corrected rows: ``dst[:, r, :]`` = sinogram ``r`` of the corrected stack (the layout of the
        reference's ``unwarp_chunk_slices_backward`` output, written pass by pass).
    rows_per_pass : int, optional
        Rows produced per pass (default: as many as keep one pass's output near 512 MiB).
    row_range : (start, stop), optional
        Half-open range of rows to produce (default: all).

    Returns
    -------
    int
        Number of passes made.
    """
    if len(src.shape) != 3:
        raise ValueError("Input must be a 3D data")
    (depth, height, width) = src.shape
    start, stop = (0, height) if row_range is None else (int(row_range[0]), int(row_range[1]))
    if not 0 <= start < stop <= height:
        raise ValueError("Selected index is out of the range")
    if tuple(dst.shape) != (depth, stop - start, width):
        raise ValueError("dst must have shape %s" % ((depth, stop - start, width),))
    itemsize = np.dtype(src.dtype).itemsize
    if rows_per_pass is None:
        rows_per_pass = max(1, min(stop - start, (512 << 20) // max(1, depth * width * itemsize)))
    rows_per_pass = int(max(1, min(rows_per_pass, 65535)))
    passes = 0
    for r0 in range(start, stop, rows_per_pass):
        r1 = min(stop, r0 + rows_per_pass)
        block = _pp.unwarp_chunk_slices_backward(src, xcenter, ycenter, list_fact, r0, r1 - 1, blend=blend)
        if hasattr(block, "cpu"):
            block = block.cpu().numpy()
        dst[:, r0 - start:r1 - start, :] = block
        passes += 1
    return passes
