"""Non-executing reader for the reference's committed run log.

The reference ships one data artefact, ``src/simulation_log.pkl`` (written by its logger at tick
1000).  It is a pickle, and pickles from an untrusted tree must never be unpickled.  This module does
NOT unpickle: it walks the opcode stream with ``pickletools.genops`` (a disassembler; it imports and
calls nothing named in the file) and interprets the two dozen opcodes the file uses as *inert data*:

* containers / scalars (dict, list, tuple, int, float, bool, None, str, bytes) are rebuilt as-is;
* a ``STACK_GLOBAL`` becomes an inert ``GlobalRef(module, name)`` marker, never an import;
* a ``REDUCE``/``BUILD`` is honoured only for the four numpy markers the file contains
  (``_reconstruct``, ``ndarray``, ``dtype``, ``scalar``) and is re-expressed as
  ``numpy.frombuffer`` on the raw bytes with a whitelisted dtype string.  Anything else raises.

Only used by ``make_fixtures.py`` in the build container (the reference tree does not travel).
"""
from __future__ import annotations

import pickletools
from dataclasses import dataclass, field

import numpy as np

_ALLOWED_GLOBALS = {
    ("numpy.core.multiarray", "_reconstruct"),
    ("numpy._core.multiarray", "_reconstruct"),
    ("numpy", "ndarray"),
    ("numpy", "dtype"),
    ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "scalar"),
}
_ALLOWED_DTYPES = {"f8", "f4", "i8", "i4", "u1", "b1"}


@dataclass(frozen=True)
class GlobalRef:
    module: str
    name: str


@dataclass
class _DtypeStub:
    code: str
    byteorder: str = "="


@dataclass
class _ArrayStub:
    value: np.ndarray | None = None
    meta: tuple = field(default_factory=tuple)


class _Mark:
    pass


_MARK = _Mark()


def _np_dtype(stub: _DtypeStub) -> np.dtype:
    if stub.code not in _ALLOWED_DTYPES:
        raise ValueError(f"dtype {stub.code!r} not whitelisted")
    bo = stub.byteorder if stub.byteorder in "<>=|" else "="
    if bo == "|":
        bo = "="
    return np.dtype(bo + stub.code if stub.code not in ("u1", "b1") else stub.code)


def _reduce(fn, args):
    if not isinstance(fn, GlobalRef):
        raise ValueError("REDUCE on a non-global")
    key = (fn.module, fn.name)
    if key not in _ALLOWED_GLOBALS:
        raise ValueError(f"global {key} not whitelisted")
    if fn.name == "_reconstruct":
        return _ArrayStub(meta=args)
    if fn.name == "dtype":
        code = args[0]
        if not isinstance(code, str):
            raise ValueError("dtype code must be str")
        return _DtypeStub(code=code)
    if fn.name == "scalar":
        dt, raw = args
        return np.frombuffer(raw, dtype=_np_dtype(dt))[0].item()
    raise ValueError(f"unsupported reduce {key}")


def _build(obj, state):
    if isinstance(obj, _DtypeStub):
        # numpy dtype __setstate__ tuple: (version, byteorder, ...)
        obj.byteorder = state[1]
        return obj
    if isinstance(obj, _ArrayStub):
        _version, shape, dt, fortran, raw = state
        if fortran:
            raise ValueError("fortran-ordered arrays not expected")
        if not isinstance(raw, (bytes, bytearray)):
            raise ValueError("object arrays not supported")
        arr = np.frombuffer(raw, dtype=_np_dtype(dt)).reshape(shape).copy()
        obj.value = arr
        return obj
    raise ValueError("BUILD on unsupported object")


def _finish(x):
    """Replace stubs by plain numpy arrays, recursively."""
    if isinstance(x, _ArrayStub):
        if x.value is None:
            raise ValueError("array never built")
        return x.value
    if isinstance(x, dict):
        return {_finish(k): _finish(v) for k, v in x.items()}
    if isinstance(x, list):
        return [_finish(v) for v in x]
    if isinstance(x, tuple):
        return tuple(_finish(v) for v in x)
    if isinstance(x, (_DtypeStub, GlobalRef)):
        raise ValueError("dangling marker in output")
    return x


def read_log(path: str):
    data = open(path, "rb").read()
    stack: list = []
    memo: list = []

    def pop_to_mark():
        items = []
        while True:
            v = stack.pop()
            if v is _MARK:
                break
            items.append(v)
        items.reverse()
        return items

    for op, arg, _pos in pickletools.genops(data):
        n = op.name
        if n in ("PROTO", "FRAME"):
            continue
        if n == "STOP":
            break
        if n == "MEMOIZE":
            memo.append(stack[-1])
        elif n in ("BINGET", "LONG_BINGET"):
            stack.append(memo[arg])
        elif n in ("BININT", "BININT1", "BININT2", "BINFLOAT", "SHORT_BINUNICODE", "BINUNICODE",
                   "SHORT_BINBYTES", "BINBYTES"):
            stack.append(arg)
        elif n == "NONE":
            stack.append(None)
        elif n == "NEWTRUE":
            stack.append(True)
        elif n == "NEWFALSE":
            stack.append(False)
        elif n == "EMPTY_LIST":
            stack.append([])
        elif n == "EMPTY_DICT":
            stack.append({})
        elif n == "MARK":
            stack.append(_MARK)
        elif n == "TUPLE1":
            a = stack.pop()
            stack.append((a,))
        elif n == "TUPLE2":
            b = stack.pop(); a = stack.pop()
            stack.append((a, b))
        elif n == "TUPLE3":
            c = stack.pop(); b = stack.pop(); a = stack.pop()
            stack.append((a, b, c))
        elif n == "TUPLE":
            stack.append(tuple(pop_to_mark()))
        elif n == "APPENDS":
            items = pop_to_mark()
            stack[-1].extend(items)
        elif n == "APPEND":
            v = stack.pop()
            stack[-1].append(v)
        elif n == "SETITEMS":
            items = pop_to_mark()
            d = stack[-1]
            for i in range(0, len(items), 2):
                k = items[i]
                d[k if not isinstance(k, _ArrayStub) else id(k)] = items[i + 1]
        elif n == "SETITEM":
            v = stack.pop(); k = stack.pop()
            stack[-1][k] = v
        elif n == "STACK_GLOBAL":
            name = stack.pop(); module = stack.pop()
            stack.append(GlobalRef(module, name))
        elif n == "REDUCE":
            args = stack.pop(); fn = stack.pop()
            stack.append(_reduce(fn, args))
        elif n == "BUILD":
            state = stack.pop()
            _build(stack[-1], state)
        else:
            raise ValueError(f"opcode {n} not supported by the inert reader")
    if len(stack) != 1:
        raise ValueError("malformed stream")
    return _finish(stack[0])


if __name__ == "__main__":
    import sys
    log = read_log(sys.argv[1])
    for k, v in log.items():
        print(k, type(v).__name__, (len(v) if hasattr(v, "__len__") else v))
