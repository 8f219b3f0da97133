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
