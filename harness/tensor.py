"""utils.tensor.append / remove / steal (utils/tensor.lua:142-193) for the small host-side sets
(pending, observed).  The big candidate set is edited on the device (Context.grid_remove)."""
import numpy as np


def append(tnsr, subtnsr, axis=0):
    """utils/tensor.lua:142-153."""
    sub = np.array(subtnsr, dtype=np.float64, copy=True)
    if tnsr is not None and np.ndim(tnsr) > 0 and np.size(tnsr) > 0:
        return np.concatenate([tnsr, sub.reshape((-1,) + np.shape(tnsr)[1:]) if axis == 0 else sub], axis=axis)
    if axis == 0 and (sub.ndim == 1 or sub.shape[0] == sub.size):
        sub = sub.reshape(1, sub.size)
    return sub


def remove(tnsr, idx1, axis=0):
    """utils/tensor.lua:158-170: stable deletion of the 1-based indices idx1; None when nothing is left."""
    idx1 = np.atleast_1d(np.asarray(idx1, dtype=np.int64))
    keep = np.ones(np.shape(tnsr)[axis], dtype=bool)
    keep[idx1 - 1] = False
    if not keep.any():
        return None
    return np.compress(keep, tnsr, axis=axis)


def steal(res, src, idx1, axis_r=0, axis_s=0):
    """utils/tensor.lua:175-193: move slices idx1 (1-based) from src to the end of res."""
    idx1 = np.atleast_1d(np.asarray(idx1, dtype=np.int64)).ravel()
    res = append(res, np.take(src, idx1 - 1, axis=axis_s), axis_r)
    src = remove(src, idx1, axis_s)
    return res, src
