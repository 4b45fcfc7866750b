"""Host-side logic of the u-space read/write hook (reference: libs/dissection.py:21-34,55-70,115-186).

The arithmetic of the hook (x += delta * scale, broadcast over the batch) runs on the GPU
(uspace_add_broadcast / the forward's mid_delta argument).  What lives here is only the
control plane the reference keys on the formatted timestep: which file, which rows, whether
this step edits at all.  Direction tables are read from disk once and kept resident on the
device instead of being re-loaded at every ODE step.
"""
import numbers
import os

import numpy as np
import torch


def should_edit(timestep_digit, t_edit):
    """Same truth table as the reference (libs/dissection.py:21-34), including the "0.00" skip."""
    if timestep_digit == "0.00":
        return False
    if isinstance(t_edit, bool):
        raise ValueError(t_edit)
    if isinstance(t_edit, (float, int)):
        return float(timestep_digit) <= t_edit
    if isinstance(t_edit, str) and t_edit.startswith("every_"):
        return float(timestep_digit) % float(t_edit[len("every_"):]) == 0.0
    raise ValueError(f"unsupported t_edit {t_edit!r}")


def select_rows(table, ith):
    """int -> that row; "31_39_20" -> mean of the rows (libs/dissection.py:55-70).  fp32, host."""
    table = np.asarray(table, dtype=np.float32)
    if isinstance(ith, (int, np.integer)):
        return np.ascontiguousarray(table[int(ith)])
    if isinstance(ith, str):
        ids = [int(s) for s in ith.split("_")]
        acc = np.zeros_like(table[0])
        for i in ids:
            acc = acc + table[i]
        return np.ascontiguousarray(acc / np.float32(len(ids)))
    raise TypeError(f"ith element must be int or 'a_b_c' string, got {ith!r}")


class HookPlan:
    """What one forward has to do at the hooked location.  ``scale`` is a float, or -- extension for the
    batched write_scales sweep -- a per-sample sequence (``row_scales``, with ``scale`` = 1)."""

    __slots__ = ("kind", "path", "ith", "scale", "row_scales")

    def __init__(self, kind, path=None, ith=None, scale=0.0):
        self.kind, self.path, self.ith = kind, path, ith
        # the reference multiplies by whatever `write_scale` is (libs/dissection.py:157): Python numbers, numpy scalars
        # (an element of np.linspace), 0-dim arrays and 0-dim tensors are all ONE factor for the whole batch
        if torch.is_tensor(scale) and scale.dim() == 0:
            scale = float(scale.item())
        elif isinstance(scale, np.ndarray) and scale.ndim == 0:
            scale = float(scale)
        if isinstance(scale, numbers.Real) or isinstance(scale, (np.floating, np.integer)):
            self.scale, self.row_scales = float(scale), None
        else:
            self.scale = 1.0
            self.row_scales = np.asarray(
                scale.detach().cpu().numpy() if torch.is_tensor(scale) else scale, dtype=np.float32).reshape(-1)


def plan_uspace_hook(timestep_digit, kwargs):
    """Decide the hook action for this step; returns None (no-op) or a HookPlan.

    Raises ValueError for an unknown dissect_name exactly where the reference does
    (libs/dissection.py:182); missing keys are tolerated."""
    if kwargs.get("dissect_task") != "uspace_uvit":
        return None
    name = kwargs.get("dissect_name")
    if name == "read":
        root = kwargs.get("read_path_root")
        return HookPlan("read", os.path.join(root, f"{kwargs['batch_id']}_{timestep_digit}"))
    if name == "write_attr":
        if not should_edit(timestep_digit, kwargs.get("t_edit")):
            return None
        return HookPlan("write", os.path.join(kwargs.get("write_path_root"), f"delta_{timestep_digit}.npy"),
                        kwargs.get("ith_attr"), kwargs.get("write_scale"))
    if name == "write_pca":
        if not should_edit(timestep_digit, kwargs.get("t_edit")):
            return None
        return HookPlan("write", os.path.join(kwargs.get("write_path_root"),
                                              f"pca{kwargs.get('pca_n')}_{timestep_digit}.npy"),
                        kwargs.get("ith_component"), kwargs.get("write_scale"))
    raise ValueError(f"dissect_name should be read or write, here is {name}")


class DeltaCache:
    """Device-resident direction vectors keyed by (file, mtime, selection)."""

    def __init__(self, max_entries=256):
        self._store = {}
        self._max = max_entries

    def get(self, path, ith, device, expected_numel):
        st = os.stat(path)
        key = (path, st.st_mtime_ns, st.st_size, str(ith), str(device))
        hit = self._store.get(key)
        if hit is None:
            row = select_rows(np.load(path), ith)
            if row.size != expected_numel:
                raise ValueError(f"{path}: direction has {row.size} elements per sample, activation has {expected_numel}")
            hit = torch.from_numpy(row.reshape(-1)).to(device)
            if len(self._store) >= self._max:
                self._store.pop(next(iter(self._store)))
            self._store[key] = hit
        return hit


def save_activation(path, tensor, kwargs=None):
    """read mode: np.save(f"{batch_id}_{t:.2f}", x) (libs/dissection.py:126-136).

    Extension: with ``direction_accumulator=`` (tools.utils_attr.DirectionAccumulator) and ``attrs=``
    ([B, attr_dim]) in the kwargs the activation is folded into device-resident attribute sums instead of
    being written to disk (the file name's timestep part is the accumulator key)."""
    acc = kwargs.get("direction_accumulator") if kwargs else None
    if acc is not None:      # DirectionAccumulator (needs attrs) or tools.utils_pca.PcaAccumulator (does not)
        acc.update(os.path.basename(path).split("_")[-1], tensor, kwargs.get("attrs"))
        return
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.save(path, tensor.detach().cpu().numpy())
