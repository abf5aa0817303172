"""Gallery persistence in the reference's on-disk format, read without executing anything.

The reference stores its gallery with ``pickle.dump`` (`/root/reference/src/app.py:67-91`) as a
``list[{'name': str, 'embedding_numpy': ndarray(1,512) float32, 'image_path': str}]`` and reads it
back with ``pickle.load`` (`src/app.py:104-123`).  ``pickle.load`` runs whatever callables the file
names; this module instead *parses* the pickle opcode stream (``pickletools.genops`` — a pure
disassembler) with a tiny data-only stack machine.  Globals are kept as inert symbols, never
imported or called; the only "objects" it will materialise are ``numpy.ndarray`` /
``numpy.dtype`` reconstructions, rebuilt here from their raw bytes with ``numpy.frombuffer``.
Anything else in the stream raises ``UnsafeGalleryError``.
"""
from __future__ import annotations

import os
import pickle
import pickletools
from typing import Any, List

import numpy as np


class UnsafeGalleryError(ValueError):
    pass


class _Sym:
    __slots__ = ("module", "name")

    def __init__(self, module: str, name: str):
        self.module, self.name = module, name

    def key(self):
        return (self.module, self.name)


class _Reduced:
    __slots__ = ("sym", "args", "state")

    def __init__(self, sym: _Sym, args: tuple):
        self.sym, self.args, self.state = sym, args, None


_MARK = object()

_NDARRAY_RECONSTRUCT = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct")}
_NDARRAY = ("numpy", "ndarray")
_DTYPE = ("numpy", "dtype")
_ALLOWED_DTYPES = {"f4", "f8", "f2", "i4", "i8", "u1", "i1", "i2", "u2", "u4", "u8", "b1"}


def _materialise(obj: Any) -> Any:
    """Turn the inert parse tree into plain Python / numpy values."""
    if isinstance(obj, _Reduced):
        k = obj.sym.key()
        if k in _NDARRAY_RECONSTRUCT:
            if not (len(obj.args) == 3 and isinstance(obj.args[0], _Sym) and obj.args[0].key() == _NDARRAY):
                raise UnsafeGalleryError("unexpected ndarray reconstruct arguments")
            st = obj.state
            if not (isinstance(st, tuple) and len(st) == 5):
                raise UnsafeGalleryError("unexpected ndarray state")
            _ver, shape, dt, fortran, raw = st
            dt = _materialise(dt)
            if not isinstance(dt, np.dtype) or not isinstance(raw, (bytes, bytearray)):
                raise UnsafeGalleryError("ndarray state is not (dtype, raw bytes)")
            shape = tuple(int(s) for s in shape)
            n = int(np.prod(shape)) if shape else 1
            if n * dt.itemsize != len(raw):
                raise UnsafeGalleryError("ndarray byte count does not match its shape")
            arr = np.frombuffer(bytes(raw), dtype=dt, count=n)
            return arr.reshape(shape, order="F" if fortran else "C").copy()
        if k == _DTYPE:
            code = obj.args[0] if obj.args else None
            if not (isinstance(code, str) and code in _ALLOWED_DTYPES):
                raise UnsafeGalleryError(f"dtype {code!r} not allowed")
            dt = np.dtype(code)
            st = obj.state
            if isinstance(st, tuple) and len(st) >= 2 and st[1] in ("<", ">", "=", "|"):
                dt = dt.newbyteorder(st[1]) if st[1] in ("<", ">") else dt
            return dt
        raise UnsafeGalleryError(f"refusing to construct {obj.sym.module}.{obj.sym.name}")
    if isinstance(obj, _Sym):
        raise UnsafeGalleryError(f"bare global {obj.module}.{obj.name} in gallery file")
    if isinstance(obj, list):
        return [_materialise(o) for o in obj]
    if isinstance(obj, tuple):
        return tuple(_materialise(o) for o in obj)
    if isinstance(obj, dict):
        return {_materialise(k): _materialise(v) for k, v in obj.items()}
    return obj


def safe_load_pickle(data: bytes) -> Any:
    """Parse a pickle byte string holding only list/dict/str/int/float/bool/None/bytes/tuple and
    numpy arrays.  Executes nothing named by the file."""
    stack: List[Any] = []
    memo: dict = {}

    def pop_mark():
        items = []
        while True:
            if not stack:
                raise UnsafeGalleryError("MARK underflow")
            x = stack.pop()
            if x is _MARK:
                break
            items.append(x)
        items.reverse()
        return items

    for op, arg, _pos in pickletools.genops(data):
        n = op.name
        if n in ("PROTO", "FRAME"):
            continue
        elif n == "STOP":
            break
        elif n == "MARK":
            stack.append(_MARK)
        elif n == "EMPTY_LIST":
            stack.append([])
        elif n == "EMPTY_DICT":
            stack.append({})
        elif n == "EMPTY_TUPLE":
            stack.append(())
        elif n in ("MEMOIZE",):
            memo[len(memo)] = stack[-1]
        elif n in ("BINPUT", "LONG_BINPUT", "PUT"):
            memo[int(arg)] = stack[-1]
        elif n in ("BINGET", "LONG_BINGET", "GET"):
            stack.append(memo[int(arg)])
        elif n in ("SHORT_BINUNICODE", "BINUNICODE", "BINUNICODE8", "UNICODE",
                   "BININT", "BININT1", "BININT2", "LONG1", "LONG4", "INT", "LONG",
                   "BINFLOAT", "FLOAT", "SHORT_BINBYTES", "BINBYTES", "BINBYTES8", "BYTEARRAY8"):
            stack.append(arg)
        elif n == "NONE":
            stack.append(None)
        elif n == "NEWTRUE":
            stack.append(True)
        elif n == "NEWFALSE":
            stack.append(False)
        elif n == "TUPLE1":
            a = stack.pop(); stack.append((a,))
        elif n == "TUPLE2":
            b = stack.pop(); a = stack.pop(); stack.append((a, b))
        elif n == "TUPLE3":
            c = stack.pop(); b = stack.pop(); a = stack.pop(); stack.append((a, b, c))
        elif n == "TUPLE":
            stack.append(tuple(pop_mark()))
        elif n == "LIST":
            stack.append(list(pop_mark()))
        elif n == "APPEND":
            v = stack.pop(); stack[-1].append(v)
        elif n == "APPENDS":
            items = pop_mark(); stack[-1].extend(items)
        elif n == "SETITEM":
            v = stack.pop(); k = stack.pop(); stack[-1][k] = v
        elif n == "SETITEMS":
            items = pop_mark()
            d = stack[-1]
            if not isinstance(d, dict):
                raise UnsafeGalleryError("SETITEMS on non-dict")
            for i in range(0, len(items), 2):
                d[items[i]] = items[i + 1]
        elif n == "DICT":
            items = pop_mark()
            stack.append({items[i]: items[i + 1] for i in range(0, len(items), 2)})
        elif n == "STACK_GLOBAL":
            name = stack.pop(); module = stack.pop()
            if not (isinstance(name, str) and isinstance(module, str)):
                raise UnsafeGalleryError("malformed STACK_GLOBAL")
            stack.append(_Sym(module, name))
        elif n == "GLOBAL":
            module, name = str(arg).split(" ", 1)
            stack.append(_Sym(module, name))
        elif n == "REDUCE":
            args = stack.pop(); fn = stack.pop()
            if not isinstance(fn, _Sym) or not isinstance(args, tuple):
                raise UnsafeGalleryError("REDUCE on non-global")
            if fn.key() not in _NDARRAY_RECONSTRUCT and fn.key() != _DTYPE:
                raise UnsafeGalleryError(f"refusing to call {fn.module}.{fn.name}")
            stack.append(_Reduced(fn, args))
        elif n == "BUILD":
            state = stack.pop()
            tgt = stack[-1]
            if not isinstance(tgt, _Reduced):
                raise UnsafeGalleryError("BUILD on a non-numpy object")
            tgt.state = state
        else:
            raise UnsafeGalleryError(f"pickle opcode {n} not allowed in a gallery file")
    if len(stack) != 1:
        raise UnsafeGalleryError("malformed gallery pickle")
    return _materialise(stack[0])


def read_gallery_file(path: str) -> List[dict]:
    """Return the raw saved records ``[{name, embedding_numpy, image_path}, ...]`` of a gallery file
    written by the reference's ``save_refs`` (`src/app.py:79-87`) or by ``write_gallery_file``."""
    with open(path, "rb") as f:
        obj = safe_load_pickle(f.read())
    if not isinstance(obj, list):
        raise UnsafeGalleryError("gallery file is not a list")
    out = []
    for rec in obj:
        if not (isinstance(rec, dict) and "name" in rec and "embedding_numpy" in rec):
            raise UnsafeGalleryError("gallery record lacks name / embedding_numpy")
        emb = rec["embedding_numpy"]
        if not isinstance(emb, np.ndarray):
            raise UnsafeGalleryError("embedding_numpy is not an ndarray")
        out.append({"name": str(rec["name"]), "embedding_numpy": emb,
                    "image_path": rec.get("image_path")})
    return out


def write_gallery_file(path: str, records: List[dict]) -> None:
    """Write ``[{name, embedding_numpy, image_path}]`` exactly as `src/app.py:79-87` does
    (pickle protocol 4 list of dicts holding numpy arrays) so the reference can read it back."""
    clean = [{"name": str(r["name"]),
              "embedding_numpy": np.ascontiguousarray(r["embedding_numpy"], dtype=np.float32),
              "image_path": r.get("image_path")} for r in records]
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        pickle.dump(clean, f, protocol=4)
