"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/l2i.h declares
(no compute calls: there is no GPU here)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "l2i.h")).read()
    return sorted(set(re.findall(r"^int\s+(l2i_\w+)\s*\(", txt, flags=re.M)))


def test_library_exports_every_declared_symbol():
    from layout2img_amd import _lib, build
    build.build()
    lib = _lib.load()
    names = _header_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SIGNATURES) == names  # the ctypes table binds exactly the header's entry points
    assert lib.l2i_version() == 1


def test_header_cites_reference_for_every_entry_point():
    txt = open(os.path.join(ROOT, "include", "l2i.h")).read()
    for ref in ("rcnn_discriminator_app.py", "resnet_generator_app_v2.py", "norm_module.py", "train_context_app_v2.py",
                "sync_batchnorm/batchnorm.py", "setup.py"):
        assert ref in txt


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "layout2img_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert "oracle" not in src.replace("no oracle", ""), f


def _params(arglist):
    """parameter declarations of a C parameter list: (kind, ...) with kind in p (pointer), i (int), l (long long), f (float)"""
    out = []
    for a in [x.strip() for x in arglist.replace("\n", " ").split(",") if x.strip() and x.strip() != "void"]:
        if "*" in a:
            out.append("p")
        elif re.search(r"\blong long\b", a):
            out.append("l")
        elif re.search(r"\bfloat\b", a):
            out.append("f")
        elif re.search(r"\b(int|unsigned)\b", a):
            out.append("i")
        else:
            raise AssertionError(a)
    return out


def test_header_definitions_and_ctypes_table_agree_argument_by_argument():
    """Three places state every entry point's signature: include/l2i.h (what a maintainer binds), the `extern "C"` definitions in csrc/*.hip (what
    is exported) and layout2img_amd/_lib.py::SIGNATURES (what the package calls through ctypes). A drifted prototype is a silent stack-garbage bug
    -- and round 6 changed a dozen argument lists: compare the three, type class by type class (pointer / int / long long / float)."""
    import ctypes
    from layout2img_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "l2i.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    protos = {m.group(1): _params(m.group(2)) for m in re.finditer(r"^int\s+(l2i_\w+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.M | re.S)}
    defs = {}
    csrc = os.path.join(ROOT, "layout2img_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith(".hip"):
            src = re.sub(r"//[^\n]*", "", open(os.path.join(csrc, f)).read())
            for m in re.finditer(r'extern\s+"C"\s+int\s+(l2i_\w+)\s*\(([^{;]*?)\)\s*\{', src, flags=re.S):
                defs[m.group(1)] = _params(m.group(2))
    kind = {ctypes.c_void_p: "p", ctypes.c_int: "i", ctypes.c_longlong: "l", ctypes.c_float: "f"}
    assert sorted(protos) == sorted(_lib.SIGNATURES)
    for name, sig in _lib.SIGNATURES.items():
        table = [kind[t] for t in sig]
        assert protos[name] == table, (name, "header", protos[name], "ctypes", table)
        if name in defs:   # (debug / trace readers are defined through macros)
            assert defs[name] == table, (name, "definition", defs[name], "ctypes", table)
    assert len([n for n in _lib.SIGNATURES if n in defs]) >= 60


def test_integration_document_stubs_bind_the_current_signatures():
    """INTEGRATION.md shows the reference-side ctypes stubs a maintainer would add; their `argtypes` lists are evaluated here and compared with
    the package's own table, so the document cannot drift behind a changed prototype."""
    import ctypes
    from layout2img_amd import _lib
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stubs = re.findall(r"^_l2i\.(l2i_\w+)\.argtypes\s*=\s*(.+?)(?=^\S)", txt, flags=re.M | re.S)
    assert len(stubs) >= 2
    for name, expr in stubs:
        assert eval(expr, {"ctypes": ctypes}) == _lib.SIGNATURES[name], name
    for name in set(re.findall(r"\bl2i_\w+", txt)):   # every entry point the document names exists
        assert name in _lib.SIGNATURES or (name.endswith("_") and any(k.startswith(name) for k in _lib.SIGNATURES)), name   # (`l2i_psp_pool_*`)
