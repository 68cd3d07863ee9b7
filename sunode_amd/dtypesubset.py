"""Flat layout of nested state / parameter records.

Restates the *semantics* of the reference's ``DTypeSubset``
(/root/reference/sunode/dtypesubset.py:71-288) for the batched engine:

* a nested ``{name: shape | {…}}`` declaration is laid out in dict insertion
  order, arrays C-order ravelled, every leaf ``float64``;
* the "subset" (= parameters we differentiate with respect to) keeps the
  declaration order (reference ``:201``) and is reachable through a *view*
  dtype with explicit byte offsets into the full record (reference
  ``:185-190``); the complement is the ``remainder`` (reference ``:283-288``).

The implementation is organised around one table of leaves (path, shape,
item offset, byte offset, dim names) from which every dtype / slice / index
map the device kernels need (``subset_index`` / ``remainder_index``: positions
of the differentiated and fixed parameters inside the flat record) is derived.
"""
from __future__ import annotations

import dataclasses
from typing import Any, Callable, Dict, List, Optional, Tuple

import numpy as np
import pandas as pd

Path = Tuple[str, ...]
Shape = Tuple[int, ...]


# --------------------------------------------------------------------------
# dict helpers (reference dtypesubset.py:10-64)
# --------------------------------------------------------------------------
def as_flattened(vals: Dict[str, Any], base: Optional[Path] = None) -> Dict[Path, Any]:
    """``{'a': {'b': 1}} -> {('a', 'b'): 1}`` keeping insertion order."""
    prefix: Path = tuple(base) if base else ()
    flat: Dict[Path, Any] = {}

    def walk(pre: Path, node: Dict[str, Any]) -> None:
        for key, item in node.items():
            if isinstance(item, dict):
                walk(pre + (key,), item)
            else:
                flat[pre + (key,)] = item
    walk(prefix, vals)
    return flat


def as_nested(vals: Dict[Path, Any]) -> Dict[str, Any]:
    """Inverse of :func:`as_flattened`."""
    root: Dict[str, Any] = {}
    for path, item in vals.items():
        if len(path) == 0:
            raise ValueError("empty path")
        node = root
        for key in path[:-1]:
            node = node.setdefault(key, {})
        if path[-1] in node:
            raise ValueError("duplicate path %r" % (path,))
        node[path[-1]] = item
    return root


def count_items(dtype: np.dtype) -> int:
    """Number of scalar items in a (possibly nested, possibly sub-array) dtype."""
    dtype = np.dtype(dtype)
    if dtype.fields is None:
        return int(np.prod(dtype.shape, dtype=np.int64)) if dtype.shape else 1
    return sum(count_items(sub) for sub, *_ in dtype.fields.values())


def _as_dict(data: np.ndarray) -> Any:
    if data.dtype.fields is None:
        return data
    return {name: _as_dict(data[name]) for name in data.dtype.names}


def _from_dict(data: np.ndarray, vals: Any) -> None:
    if data.dtype.fields is None:
        data[...] = vals
        return
    for name in data.dtype.names:
        sub = data.dtype.fields[name][0]
        if sub.fields is not None:
            _from_dict(data[name], vals[name])
        else:
            data[name] = vals[name]


@dataclasses.dataclass
class _Leaf:
    path: Path
    shape: Shape
    dim_names: Tuple[str, ...]
    start: int          # first flat item index inside the full record
    size: int           # number of items
    in_subset: bool


class DTypeSubset:
    """See module docstring.  Public attribute names follow the reference."""

    def __init__(
        self,
        dims: Dict[str, Any],
        subset_paths: List[Path],
        fixed_dtype: Optional[np.dtype] = None,
        coords: Optional[Dict[str, Any]] = None,
        dim_basename: str = "",
    ) -> None:
        self._decl = dims
        self._fixed_dtype = fixed_dtype
        self.coords: Dict[str, pd.Index] = (
            {} if coords is None else {k: pd.Index(v) for k, v in coords.items()}
        )
        wanted = {tuple(p) for p in subset_paths}
        self._auto_dims: set = set()

        self._leaves: List[_Leaf] = []
        self.dims: Dict[str, Any] = {}
        self.dtype, self.subset_dtype, self.subset_view_dtype = self._build(
            dims, (), wanted, self.dims, dim_basename, 0
        )[:3]

        self.paths: List[Path] = [leaf.path for leaf in self._leaves]
        # declaration order wins over the order the caller listed the paths in
        self.subset_paths: List[Path] = [l.path for l in self._leaves if l.in_subset]
        self.flat_slices: Dict[Path, slice] = {
            l.path: slice(l.start, l.start + l.size) for l in self._leaves
        }
        self.flat_shapes: Dict[Path, Shape] = {l.path: l.shape for l in self._leaves}
        self.item_count: int = sum(l.size for l in self._leaves)
        self._remainder: Optional["DTypeSubset"] = None

    # -- construction ------------------------------------------------------
    def _build(self, decl, prefix, wanted, dims_out, dim_basename, item_base):
        """Walk one nesting level; returns (dtype, subset_dtype, view_dtype, n_items)."""
        full, sub, view_names, view_formats, view_offsets = [], [], [], [], []
        byte_off = 0
        n_items = 0
        for name, spec in decl.items():
            path = prefix + (name,)
            if isinstance(spec, dict):
                child_dims: Dict[str, Any] = {}
                cd, cs, cv, cn = self._build(
                    spec, path, wanted, child_dims,
                    "dim_basename_%s" % name, item_base + n_items)
                dims_out[name] = child_dims
                full.append((name, cd, ()))
                if cs.itemsize > 0:
                    sub.append((name, cs, ()))
                    view_names.append(name)
                    view_formats.append(cv)
                    view_offsets.append(byte_off)
                byte_off += cd.itemsize
                n_items += cn
                continue

            if self._fixed_dtype is None:
                leaf_dtype, spec = spec
            else:
                leaf_dtype = self._fixed_dtype
            if isinstance(spec, (int, str)):
                spec = (spec,)
            shape: List[int] = []
            names: List[str] = []
            for axis, entry in enumerate(spec):
                if isinstance(entry, str):
                    if entry not in self.coords:
                        raise KeyError("Unknown dimension name: %s" % entry)
                    shape.append(len(self.coords[entry]))
                    names.append(entry)
                else:
                    label = "%s_%s_dim%s__" % (dim_basename, name, axis)
                    if label in self.coords:
                        raise ValueError(
                            "Can not create two different dimensions with the same name: %s." % label)
                    self.coords[label] = pd.RangeIndex(int(entry), name=label)
                    self._auto_dims.add(label)
                    shape.append(int(entry))
                    names.append(label)
            shape_t = tuple(shape)
            size = int(np.prod(shape_t, dtype=np.int64)) if shape_t else 1
            in_sub = path in wanted
            self._leaves.append(_Leaf(path, shape_t, tuple(names), item_base + n_items, size, in_sub))
            dims_out[name] = (leaf_dtype, names)
            full.append((name, leaf_dtype, shape_t))
            if in_sub:
                sub.append((name, leaf_dtype, shape_t))
                view_names.append(name)
                view_formats.append((leaf_dtype, shape_t))
                view_offsets.append(byte_off)
            byte_off += np.dtype([(name, leaf_dtype, shape_t)]).itemsize
            n_items += size
        dtype = np.dtype(full)
        view = np.dtype({
            "names": view_names, "formats": view_formats,
            "offsets": view_offsets, "itemsize": dtype.itemsize,
        })
        return dtype, np.dtype(sub), view, n_items

    # -- counts -------------------------------------------------------------
    @property
    def n_subset(self) -> int:
        return count_items(self.subset_dtype)

    @property
    def n_items(self) -> int:
        return count_items(self.dtype)

    # -- index maps used by the batched kernels -----------------------------
    @property
    def subset_index(self) -> np.ndarray:
        """Flat positions (in the full record) of the subset items, subset order."""
        idx = [np.arange(l.start, l.start + l.size) for l in self._leaves if l.in_subset]
        return np.concatenate(idx).astype(np.int64) if idx else np.zeros(0, np.int64)

    @property
    def remainder_index(self) -> np.ndarray:
        """Flat positions of the non-subset items, declaration order."""
        idx = [np.arange(l.start, l.start + l.size) for l in self._leaves if not l.in_subset]
        return np.concatenate(idx).astype(np.int64) if idx else np.zeros(0, np.int64)

    # -- value plumbing ------------------------------------------------------
    def set_from_subset(self, value_buffer: np.ndarray, subset_buffer: np.ndarray) -> None:
        value_buffer.view(self.subset_dtype).fill(subset_buffer)

    def as_dataclass(
        self,
        dataclass_name: str,
        flat_subset: np.ndarray,
        flat_remainder: np.ndarray,
        item_map: Optional[Callable[[np.ndarray], Any]] = None,
    ) -> Any:
        """Nested dataclass instance ``p.alpha`` / ``p.rates.sigma`` whose leaves
        are taken consecutively from ``flat_subset`` (differentiated leaves) and
        ``flat_remainder`` (all other leaves).  Reference ``:215-259``."""
        conv = item_map if item_map is not None else (lambda x: x)
        cursors = {True: 0, False: 0}
        sources = {True: np.asarray(flat_subset, dtype=object) if len(flat_subset) else np.zeros(0, object),
                   False: np.asarray(flat_remainder, dtype=object) if len(flat_remainder) else np.zeros(0, object)}
        by_path = {l.path: l for l in self._leaves}

        def take(leaf: _Leaf) -> Any:
            src = sources[leaf.in_subset]
            lo = cursors[leaf.in_subset]
            if lo + leaf.size > len(src):
                raise AssertionError("not enough items for %r" % (leaf.path,))
            cursors[leaf.in_subset] = lo + leaf.size
            return conv(src[lo:lo + leaf.size].reshape(leaf.shape))

        def make(name: str, dtype: np.dtype, prefix: Path) -> Any:
            values = []
            for field in dtype.names:
                sub = dtype.fields[field][0]
                if sub.fields is None:
                    values.append(take(by_path[prefix + (field,)]))
                else:
                    values.append(make(field, sub, prefix + (field,)))
            cls = dataclasses.make_dataclass(name, list(dtype.names))
            return cls(*values)

        if self.dtype.names is None:
            out = dataclasses.make_dataclass(dataclass_name, [])()
        else:
            out = make(dataclass_name, self.dtype, ())
        if cursors[True] != len(sources[True]) or cursors[False] != len(sources[False]):
            raise AssertionError("unused items left over")
        return out

    def from_dict(self, vals: Dict[str, Any], out: Optional[np.ndarray] = None) -> np.ndarray:
        if out is None:
            out = np.zeros((1,), dtype=self.dtype)[0]
        _from_dict(out, vals)
        return out

    def subset_from_dict(self, vals: Dict[str, Any], out: Optional[np.ndarray] = None) -> np.ndarray:
        if out is None:
            out = np.zeros((1,), dtype=self.subset_dtype)[0]
        _from_dict(out, vals)
        return out

    def as_dict(self, vals: np.ndarray) -> Dict[str, Any]:
        if vals.dtype != self.dtype:
            raise ValueError("Invalid dtype.")
        return _as_dict(vals)

    def subset_as_dict(self, vals: np.ndarray) -> Dict[str, Any]:
        if vals.dtype != self.subset_dtype:
            raise ValueError("Invalid dtype.")
        return _as_dict(vals)

    @property
    def remainder(self) -> "DTypeSubset":
        if self._remainder is None:
            rest = [l.path for l in self._leaves if not l.in_subset]
            self._remainder = DTypeSubset(
                self._decl, rest, fixed_dtype=self._fixed_dtype,
                coords={k: v for k, v in self.coords.items() if k not in self._auto_dims},
            )
        return self._remainder
