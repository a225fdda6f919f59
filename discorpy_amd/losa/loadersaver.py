"""Calibration-file helpers adjacent to the unwarp path.

File formats and behaviour of the four metadata functions of the reference
(``/root/reference/discorpy/losa/loadersaver.py``: ``save_metadata_txt`` :713-751,
``load_metadata_txt`` :754-776, ``save_metadata_json`` :791-826, ``load_metadata_json`` :829-848), so
that ``(xcenter, ycenter, list_fact)`` written by discorpy is read here and vice versa.  Pure
Python, no GPU involved.  Nothing else of ``losa`` (images, HDF, plots) is in scope.
"""
import json
from pathlib import Path

import numpy as np

__all__ = ["save_metadata_txt", "load_metadata_txt", "save_metadata_json", "load_metadata_json"]


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
