"""CPU: the generated CPython binding (layout2img_amd/fastcall.py) hands the C ABI exactly what ctypes hands it.

An echo library with the 69 prototypes of include/l2i.h is generated and compiled here; every entry point is called through ctypes (with the
package's `argtypes`) and through the generated wrapper with the same Python arguments -- random ones per parameter kind, the edge values, and
the 433 calls of a dry-run training iteration (tests/dryrun.py) -- and the words the callee received must be identical."""
import ctypes
import os
import random
import subprocess

import pytest
import torch

from tests import dryrun

_KIND = {ctypes.c_void_p: "p", ctypes.c_int: "i", ctypes.c_longlong: "l", ctypes.c_float: "f"}
_CT = {"p": "void*", "i": "int", "l": "long long", "f": "float"}


def _echo_library(tmp_path):
    """Every prototype of the table, each storing its arguments as 64-bit words (pointer value, sign-extended int, float bit pattern) + the count."""
    from layout2img_amd import _lib
    src = ["#include <stdint.h>\n#include <string.h>\nuint64_t echo_words[64]; int echo_count;\n"]
    for name, sig in sorted(_lib.SIGNATURES.items()):
        kinds = [_KIND[t] for t in sig]
        params = ", ".join(f"{_CT[k]} a{i}" for i, k in enumerate(kinds)) or "void"
        body = []
        for i, k in enumerate(kinds):
            if k == "p":
                body.append(f"echo_words[{i}] = (uint64_t)(uintptr_t)a{i};")
            elif k == "f":
                body.append(f"{{ uint32_t b; memcpy(&b, &a{i}, 4); echo_words[{i}] = b; }}")
            else:
                body.append(f"echo_words[{i}] = (uint64_t)(int64_t)a{i};")
        src.append(f"int {name}({params}) {{ {' '.join(body)} echo_count = {len(kinds)}; return 0; }}\n")
    c = tmp_path / "echo.c"
    c.write_text("".join(src))
    so = tmp_path / "libecho.so"
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", str(c), "-o", str(so)])
    lib = ctypes.CDLL(str(so))
    for name, sig in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = sig, ctypes.c_int
    return lib


@pytest.fixture()
def bindings(tmp_path):
    from layout2img_amd import _lib, fastcall
    fastcall.build()
    echo = _echo_library(tmp_path)
    mod = fastcall.load(lib=echo)
    words = (ctypes.c_uint64 * 64).in_dll(echo, "echo_words")
    count = ctypes.c_int.in_dll(echo, "echo_count")

    def received():
        out = tuple(words[:count.value])
        count.value = 0
        return out
    try:
        yield echo, mod, received
    finally:
        fastcall.load()      # (one module object per process: bind the wrappers back to libl2i_hip.so)


def _random_args(kinds, rng):
    out = []
    for k in kinds:
        if k == "p":
            out.append(rng.choice([None, 0, rng.randrange(1, 2 ** 48) * 16, 2 ** 63 + 4096, 2 ** 64 - 8]))
        elif k == "i":
            out.append(rng.choice([0, 1, -1, True, False, 2 ** 31 - 1, -2 ** 31, rng.randrange(-10 ** 6, 10 ** 6)]))
        elif k == "l":
            out.append(rng.choice([0, -1, 2 ** 40 + 3, 2 ** 63 - 1, -2 ** 63, rng.randrange(0, 2 ** 34)]))
        else:
            out.append(rng.choice([0.0, 1.0, -2.5, 1e-8, 3, 0.1, float("inf"), 1e-45, rng.random()]))
    return out


def test_every_entry_point_receives_the_same_words_as_through_ctypes(bindings):
    from layout2img_amd import _lib
    echo, mod, received = bindings
    rng = random.Random(7)
    for name, sig in sorted(_lib.SIGNATURES.items()):
        kinds = [_KIND[t] for t in sig]
        for _ in range(12):
            args = _random_args(kinds, rng)
            assert getattr(echo, name)(*args) == 0
            a = received()
            assert getattr(mod, name)(*args) == 0
            b = received()
            assert a == b and len(a) == len(kinds), (name, args, a, b)


def test_the_calls_of_a_training_iteration_arrive_unchanged(bindings):
    """The real argument lists: one dry-run iteration's 433 calls, ctypes arrays (l2i_psp_stages_*) included."""
    echo, mod, received = bindings
    real_calls = []
    with dryrun.dry_run() as trace:
        from layout2img_amd import _lib
        record = _lib.call

        def tee(name, *args):
            real_calls.append((name, args))     # (the recorder normalises arrays: keep the objects themselves here)
            record(name, *args)
        _lib.call = tee
        tr, (real, label, bbox, z, z_im) = dryrun.build("coco", torch.bfloat16)
        tr.step(real, label, bbox, z, None)
        del real_calls[:]
        tr.step(real, label, bbox, z, None)
    assert len(real_calls) == 433
    arrays = 0
    for name, args in real_calls:
        arrays += sum(isinstance(a, ctypes.Array) for a in args)
        assert getattr(echo, name)(*args) == 0
        a = received()
        assert getattr(mod, name)(*args) == 0
        assert a == received(), (name, args)
    assert arrays > 0


def test_the_binding_refuses_what_ctypes_refuses_and_what_ctypes_would_truncate(bindings):
    echo, mod, received = bindings
    ok = [None, 1, 1, 1, None, None, 0, 0, None, 0, None]      # l2i_channel_stats(x, rows, C, rows_per_group, sums, sqsums, raw, dtype, scratch, scratch_floats, stream)
    assert mod.l2i_channel_stats(*ok) == 0
    for k, bad, exc in ((2, 0.5, TypeError), (2, 2 ** 31, OverflowError), (2, "3", TypeError), (1, 1.0, TypeError), (0, "x", TypeError), (0, 1.5, TypeError)):
        args = list(ok)
        args[k] = bad
        with pytest.raises(exc):
            mod.l2i_channel_stats(*args)
    with pytest.raises(TypeError):
        mod.l2i_channel_stats(*ok[:-1])
    with pytest.raises(TypeError):
        mod.l2i_adam_step(None, None, None, None, 4, True, 0.0, 0.999, 1e-8, 1, 1.0, None, None)   # a bool is not a learning rate


def test_call_dispatches_to_the_binding_when_asked(monkeypatch):
    """L2I_FASTCALL=1: `_lib.call` goes through the generated wrappers (same error behaviour: a non-zero return code raises)."""
    from layout2img_amd import _lib, fastcall
    fastcall.build()
    monkeypatch.setenv("L2I_FASTCALL", "1")
    monkeypatch.setattr(_lib, "_FN", {})
    monkeypatch.setattr(_lib, "_lib", None)
    _lib.load()
    assert _lib._FN["l2i_version"].__class__.__name__ == "builtin_function_or_method"
    with pytest.raises(RuntimeError, match="bad argument"):
        _lib.call("l2i_channel_stats", None, 0, 0, 0, None, None, 0, 0, None, 0, None)
    monkeypatch.setenv("L2I_FASTCALL", "0")
    monkeypatch.setattr(_lib, "_FN", {})
    monkeypatch.setattr(_lib, "_lib", None)
    _lib.load()
    assert "ctypes" in type(_lib._FN["l2i_version"]).__module__ or "CDLL" in repr(type(_lib._FN["l2i_version"]))
