"""Calibration-file and stack I/O adjacent to the unwarp path.

File formats and behaviour of the reference's ``discorpy/losa/loadersaver.py`` for what feeds, or is fed by, the unwarp
functions -- so that a script swaps ``import discorpy.losa.loadersaver as losa`` for this module and keeps running:

* the four metadata functions (``save_metadata_txt`` :713-751, ``load_metadata_txt`` :754-776, ``save_metadata_json``
  :791-826, ``load_metadata_json`` :829-848): ``(xcenter, ycenter, list_fact)`` written by discorpy is read here and vice versa;
* ``load_image`` :84-106 / ``save_image`` :413-451 (PIL, imported when first used);
* the HDF5 entries ``load_hdf_file`` :248-329, ``load_hdf_object`` :332-355, ``save_hdf_file`` :560-605 and
  ``open_hdf_stream`` :608-656 (h5py, imported when first used): ``load_hdf_object`` hands back the dataset itself, which
  ``post.unwarp_slice_backward`` / ``unwarp_chunk_slices_backward`` / ``losa.stream.correct_stack`` read band by band, and
  ``open_hdf_stream`` the dataset they write into.

Pure Python, no GPU involved.  Plots, pickles and the HDF tree browsers of ``losa`` are not on the path and not here.
"""
import json
from pathlib import Path

import numpy as np

__all__ = ["save_metadata_txt", "load_metadata_txt", "save_metadata_json", "load_metadata_json", "load_image", "save_image",
           "load_hdf_file", "load_hdf_object", "save_hdf_file", "open_hdf_stream"]

_HDF_SUFFIXES = {".hdf", ".h5", ".nxs", ".hdf5"}


def _existing(file_path):
    path = Path(file_path)
    if not path.exists():
        raise ValueError(f"No such file: {path}")
    return path


def _free_name(path):
    """`name_0000.ext`, `name_0001.ext`, ... : the first one that does not exist yet."""
    if not path.exists():
        return path
    n = 0
    while True:
        cand = path.parent / f"{path.stem}_{n:04d}{path.suffix}"
        if not cand.exists():
            return cand
        n += 1


def _prepare(file_path, suffixes, default, overwrite):
    path = Path(file_path).resolve()
    if path.suffix.lower() not in suffixes:
        path = path.with_suffix(default)
    path.parent.mkdir(parents=True, exist_ok=True)
    return path if overwrite else _free_name(path)


def save_metadata_txt(file_path, xcenter, ycenter, list_fact, overwrite=True):
    """
    Write metadata to a text file: one ``key = value`` line for xcenter, ycenter, factor0, ...

    Returns the (possibly renamed) file path; ``overwrite=False`` appends ``_0000``, ``_0001``, ...
    A suffix other than .txt / .dat is replaced by .txt.
    """
    path = _prepare(file_path, {".txt", ".dat"}, ".txt", overwrite)
    lines = ["xcenter = " + str(xcenter), "ycenter = " + str(ycenter)]
    lines += ["factor" + str(i) + " = " + str(fact) for i, fact in enumerate(list_fact)]
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return path


def load_metadata_txt(file_path):
    """
    Load distortion coefficients from a text file: the last whitespace-separated token of every
    line is a number (so ``key = value`` and the ``key : value`` of ``data/coef_dot_05.txt`` both
    parse).  Returns ``(xcenter, ycenter, list_fact)``.
    """
    with open(_existing(file_path), "r") as f:
        data = [float(line.split()[-1]) for line in f.read().splitlines()]
    return data[0], data[1], data[2:]


def _jsonable(obj):
    if isinstance(obj, np.integer):
        return int(obj)
    if isinstance(obj, np.floating):
        return float(obj)
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    raise TypeError(f"Object of type '{type(obj).__name__}' is not JSON serializable")


def save_metadata_json(file_path, xcenter, ycenter, list_fact, overwrite=True):
    """Write ``{"xcenter", "ycenter", "list_fact"}`` as JSON (suffix forced to .json)."""
    path = _prepare(file_path, {".json"}, ".json", overwrite)
    meta = {"xcenter": float(xcenter), "ycenter": float(ycenter), "list_fact": list_fact}
    with open(path, "w") as f:
        json.dump(meta, f, indent=4, default=_jsonable)
    return path


def load_metadata_json(file_path):
    """Returns ``(xcenter, ycenter, list_fact)`` from a JSON file written by ``save_metadata_json``."""
    with open(_existing(file_path), "r") as f:
        meta = json.load(f)
    return meta["xcenter"], meta["ycenter"], meta["list_fact"]


# ------------------------------------------------------------------------------------------------ images

def _pil():
    try:
        from PIL import Image
    except ImportError as e:          # pragma: no cover -- PIL is a dependency of the reference too
        raise ImportError("load_image / save_image need Pillow (PIL), as discorpy's do") from e
    return Image


def load_image(file_path, average=True):
    """
    An image file as a float32 array (reference ``loadersaver.py:84-106``); a multichannel image is averaged over its
    shortest axis when ``average`` is True.  Any failure to read surfaces as ``ValueError``, as there.
    """
    path = _existing(file_path)
    try:
        mat = np.array(_pil().open(path), dtype=np.float32)
    except Exception as error:      # noqa: BLE001 -- the reference's contract: ValueError whatever went wrong
        raise ValueError(error)
    if mat.ndim > 2 and average is True:
        mat = np.mean(mat, axis=int(np.argmin(mat.shape)))
    return mat


def save_image(file_path, mat, overwrite=True):
    """
    Save 2D data to an image (reference ``loadersaver.py:413-451``): TIFF keeps the values (a multichannel array is
    averaged over its shortest axis first); any other format gets the data stretched to 0..255 ``uint8`` unless it already
    is ``uint8``.  Returns the path written (``_0000``, ``_0001`` ... appended when ``overwrite`` is False).
    """
    path = Path(file_path).resolve()
    mat = np.asarray(mat)
    if path.suffix in (".tif", ".tiff"):
        if mat.ndim > 2:
            mat = np.mean(mat, axis=int(np.argmin(mat.shape)))
    elif mat.dtype != np.uint8:
        lo, hi = np.min(mat), np.max(mat)
        mat = np.uint8(255.0 * (mat - lo) / (hi - lo)) if hi != lo else np.uint8(mat)
    path.parent.mkdir(parents=True, exist_ok=True)
    if not overwrite:
        path = _free_name(path)
    try:
        _pil().fromarray(mat).save(path)
    except Exception as error:      # noqa: BLE001
        raise ValueError(f"Couldn't write to file: {path}. Error {error}")
    return path


# ------------------------------------------------------------------------------------------------ HDF5

def _h5py():
    try:
        import h5py
    except ImportError as e:
        raise ImportError("the HDF5 entry points need h5py (as discorpy.losa.loadersaver does); every other function of this "
                          "module works without it") from e
    return h5py


def _open_for_reading(file_path):
    path = _existing(file_path)
    try:
        return _h5py().File(path, "r")
    except ImportError:
        raise
    except Exception as error:      # noqa: BLE001
        raise ValueError(f"Error: {error}")


def _first_data_key(h5py, ifile):
    """The first group that holds a dataset called ``data`` -- NeXus files keep the stack there (reference ``_get_key``
    :237-245, used when ``key_path`` is None)."""
    def visit(name, obj):
        if isinstance(obj, h5py.Group):
            for key, val in obj.items():
                if key == "data" and isinstance(val, h5py.Dataset):
                    return f"{obj.name}/{key}"
        return None
    return ifile.visititems(visit)


def load_hdf_object(file_path, key_path):
    """
    The dataset at ``key_path`` of an hdf / nxs file AS AN OBJECT (reference ``loadersaver.py:332-355``): nothing is read
    until it is sliced.  Pass it to ``post.unwarp_slice_backward`` / ``unwarp_chunk_slices_backward`` or
    ``losa.stream.correct_stack``: they read only the row band a request needs, chunk by chunk.
    """
    ifile = _open_for_reading(file_path)
    if key_path not in ifile:
        raise ValueError(f"Couldn't open object with the key: {key_path}")
    return ifile[key_path]


def load_hdf_file(file_path, key_path=None, index=None, axis=0):
    """
    Load a 2D dataset, or a 3D dataset or slices of it as float32 (reference ``loadersaver.py:248-329``).

    ``index``: an int (one slice, returned 2D), ``(start, stop)``, ``(start, stop, step)`` or any longer sequence of
    indices, taken along ``axis`` (clipped to 0..2); None loads the whole stack.  ``key_path`` None searches the file for
    the first ``.../data`` dataset.
    """
    ifile = _open_for_reading(file_path)
    if key_path is None:
        key_path = _first_data_key(_h5py(), ifile)
        if key_path is None:
            raise ValueError("Please provide the key path to the dataset!")
    if key_path not in ifile:
        raise ValueError("Couldn't open object with the key path: {}".format(key_path))
    idata = ifile[key_path]
    ndim = len(idata.shape)
    if ndim < 2 or ndim > 3:
        raise ValueError("Require a 2D or 3D dataset!")
    if ndim == 2:
        return np.asarray(idata)
    axis = int(np.clip(axis, 0, 2))
    if index is None:
        return np.float32(idata[:, :, :])

    def take(sel):
        key = [slice(None)] * 3
        key[axis] = sel
        return np.float32(idata[tuple(key)])
    if isinstance(index, (int, np.integer)) and not isinstance(index, bool):
        return take(int(index))                       # h5py drops the indexed axis: a 2D slice
    if not isinstance(index, (tuple, list)):
        raise ValueError("index must be an int, a tuple or a list")
    picks = list(range(*index)) if len(index) in (2, 3) else list(index)
    mat = take(picks)
    if mat.shape[axis] == 0:
        raise ValueError("Empty indices!")
    if mat.shape[axis] == 1:
        mat = np.swapaxes(mat, axis, 0)[0]
    return mat


def _hdf_target(file_path, overwrite):
    path = Path(file_path).resolve()
    if path.suffix.lower() not in _HDF_SUFFIXES:
        path = path.with_suffix(".hdf")
    path.parent.mkdir(parents=True, exist_ok=True)
    if not overwrite:
        path = _free_name(path)
    try:
        return path, _h5py().File(path, "w")
    except ImportError:
        raise
    except Exception as error:      # noqa: BLE001
        raise ValueError(f"Couldn't write to file: {path}. Error {error}")


def save_hdf_file(file_path, idata, key_path="entry", overwrite=True):
    """
    Write ``idata`` to ``<key_path>/data`` of a new hdf file (reference ``loadersaver.py:560-605``); a suffix other than
    .hdf / .h5 / .nxs / .hdf5 becomes .hdf.  Returns the path written.
    """
    path, ofile = _hdf_target(file_path, overwrite)
    ofile.create_group(key_path).create_dataset("data", data=idata)
    ofile.close()
    return path


def open_hdf_stream(file_path, data_shape, key_path="entry/data", data_type="float32", overwrite=True, **options):
    """
    Open a new hdf file and return the (empty) dataset ``key_path`` of shape ``data_shape`` to write into, slice by slice
    -- e.g. as ``dst`` of ``losa.stream.correct_stack`` (reference ``loadersaver.py:608-656``).  Keyword options are dicts
    of ``{key: value}`` metadata written beside it, e.g. ``options={"entry/angles": angles, "entry/energy": 53}``; a
    metadata key that contains ``key_path`` is refused, as there.
    """
    path, ofile = _hdf_target(file_path, overwrite)
    for opts in options.values():
        for key in opts:
            if key_path in key:
                raise ValueError("!!! Selected key path, '{0}', can not be a child key-path of '{1}' !!!\n!!! Change to make "
                                 "sure they are at the same level !!!".format(key, key_path))
            ofile.create_dataset(key, data=opts[key])
    return ofile.create_dataset(key_path, data_shape, dtype=data_type)
