"""Build libl2i_hip.so (gfx950 only) with hipcc, in-tree.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libl2i_hip.so")
SOURCES = ["conv_igemm.hip", "conv_wgrad.hip", "weights.hip", "norm.hip", "roi_align.hip", "attention.hip", "misc.hip", "psp.hip", "layout.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"] + os.environ.get("L2I_EXTRA_FLAGS", "").split()


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    try:   # the generated CPython binding of the C ABI (gcc, a second or two; no device code). Opt-in at run time (L2I_FASTCALL=1), so a host
        from . import fastcall   # without Python.h / gcc still builds the library; fastcall.load() then says what is missing
        fastcall.build(force=force)
    except Exception as e:   # noqa: BLE001
        sys.stderr.write(f"layout2img_amd.build: the optional CPython binding was not built ({e})\n")
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
